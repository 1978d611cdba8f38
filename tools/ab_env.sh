#!/bin/bash
# A/B bench of environment settings on the GPU box: tools/ab_env.sh "NAME=VALUE ..." "NAME=VALUE ..." ...  (one headline-only bench run per argument; "" = defaults)
i=0
for v in "$@"; do
  i=$((i+1))
  env $v timeout 300 python bench.py --cpu-frames 0 --host-frames 0 --pose-frames 0 --latency-calls 0 --pipelined-steps 0 ${BENCH_ARGS} 2>/tmp/abenv_$i.err | tail -1 > /tmp/abenv_$i.json
  python - "$v" $i <<PY
import json, sys
d = json.load(open("/tmp/abenv_%s.json" % sys.argv[2]))
s = d["stage_ms_per_step"]
print("[%s]" % sys.argv[1], d["value"], "fps (one stream %s)  sweep frac" % (d.get("one_stream") or {}).get("value"), d["roofline"]["frac"], " ".join("%s=%.3f" % (k, v) for k, v in s.items()))
PY
done
