#!/bin/bash
# A/B of library builds on the GPU box: tools/ab_lib.sh <variant dir under cylindertag_amd/_var | "base"> ...   (BENCH_ARGS as in ab_env.sh)
for v in "$@"; do
  lib=$PWD/cylindertag_amd/_var/$v/libctag_hip.so; [ "$v" = base ] && lib=$PWD/cylindertag_amd/_build/libctag_hip.so
  CTAG_HIP_LIB=$lib timeout 300 python bench.py --cpu-frames 0 --host-frames 0 --pose-frames 0 --latency-calls 0 --pipelined-steps 0 ${BENCH_ARGS} 2>/dev/null | tail -1 > /tmp/ablib_$v.json
  python - "$v" <<PY
import json, sys
d = json.load(open("/tmp/ablib_%s.json" % sys.argv[1]))
s = d["stage_ms_per_step"]
print("[%s]" % sys.argv[1], d["value"], "fps (one stream %s)  sweep frac" % (d.get("one_stream") or {}).get("value"), d["roofline"]["frac"], " ".join("%s=%.3f" % (k, v) for k, v in s.items()))
PY
done
