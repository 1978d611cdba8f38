#!/bin/bash
for v in "$@"; do
  for fr in 4096 512; do
  CTAG_HIP_LIB=$PWD/cylindertag_amd/_var/$v/libctag_hip.so timeout 300 python bench.py --frames $fr --steps 10 --warmup 3 --cpu-frames 0 --host-frames 0 --pose-frames 0 --latency-calls 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('$v', $fr, d['value'], 'welsch', d['stage_ms_per_step']['welsch'], 'packed', d['stage_ms_per_step']['quad_edges'])"
  done
done
