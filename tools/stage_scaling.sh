#!/bin/bash
# per-kernel times of bench steps of N frames, scaled to 4096 frames: where small batches lose (kernel tails)
for n in "$@"; do
  python3 bench.py --frames $n --steps 20 --warmup 3 --cpu-frames 0 --host-frames 0 --pose-frames 0 --latency-calls 0 2>/dev/null | tail -1 > /tmp/ss_$n.json
  python3 - $n <<'PY'
import json, sys
n = int(sys.argv[1]); d = json.load(open("/tmp/ss_%d.json" % n)); s = d["stage_ms_per_step"]
print(n, d["value"], d["ms_per_step"], {k: round(v * 4096 / n, 2) for k, v in s.items()})
PY
done
