#!/bin/bash
# A/B bench of library variants built with `make -C cylindertag_amd OUT=_var/<name> EXTRA=-D... _var/<name>/libctag_hip.so`.
# usage (on the GPU box): tools/ab.sh w0 w32 ...   -> one line per variant: fps, sweep roofline fraction, stage ms
for v in "$@"; do
  CTAG_HIP_LIB=$PWD/cylindertag_amd/_var/$v/libctag_hip.so timeout 300 python bench.py --cpu-frames 0 --host-frames 0 --pose-frames 0 --latency-calls 0 2>/dev/null | tail -1 > /tmp/ab_$v.json
  python - "$v" <<PY
import json, sys
d = json.load(open("/tmp/ab_%s.json" % sys.argv[1]))
print(sys.argv[1], d["value"], d["roofline"]["frac"], d["stage_ms_per_step"])
PY
done
