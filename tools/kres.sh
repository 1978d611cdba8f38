#!/bin/bash
# per-kernel register / spill / LDS / occupancy summary of one .hip source (cross-compiles for gfx950; no GPU needed)
# usage: tools/kres.sh cylindertag_amd/csrc/k_quad.hip [extra hipcc flags]
src=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -fPIC -c "$src" -o /dev/null -Rpass-analysis=kernel-resource-usage "$@" 2>&1 |
  python3 -c '
import re, sys, subprocess
cur = None
rows = []
for l in sys.stdin:
    m = re.search(r"remark:\s+([A-Za-z][\w ]*?)(?: \[[\w/]+\])?: (.+?) \[-Rpass", l)
    if not m: continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k in ("Function Name", "Name"):
        cur = {"name": v}; rows.append(cur)
    elif cur is not None:
        cur[k] = v
for r in rows:
    try:
        name = subprocess.check_output(["c++filt", r["name"]]).decode().strip()
    except Exception:
        name = r["name"]
    name = re.sub(r"\(.*", "", name)
    print("%-58s VGPR %4s AGPR %3s SGPR %4s  spill V %3s S %3s  scratch %4s  LDS %6s  occ %s" % (name[:58], r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"),
          r.get("VGPRs Spill"), r.get("SGPRs Spill"), r.get("ScratchSize"), r.get("LDS Size"), r.get("Occupancy")))
'
