#!/bin/bash
# Collects several rocprofv3 --pmc passes of one small bench run (GPU box) and prints per-kernel means.
# usage: tools/pmc_run.sh "<counters pass 1>" "<counters pass 2>" ...
export TMPDIR=/tmp
B="python3 bench.py --frames 1024 --chunk 1024 --steps 1 --warmup 0 --cpu-frames 0 --host-frames 0 --pose-frames 0 --latency-calls 0"
rm -rf gpurun_out/pmc_*
i=0
for set in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $set -d gpurun_out/pmc_$i -- $B > gpurun_out/pmc_$i.log 2>&1 || tail -3 gpurun_out/pmc_$i.log
done
python3 tools/pmc_summary.py gpurun_out/pmc_*
