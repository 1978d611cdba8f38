#!/bin/bash
# HBM bytes of the threshold+label sweep (GPU box): FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes (own runs, --kernel-trace
# only) of one 1024-frame bench pass, summarised by tools/pmc_traffic.py into profiles/<tag>_pmc_traffic.json.  usage: tools/pmc_traffic.sh r02
TAG=${1:-r02}
export TMPDIR=/tmp
B="python3 bench.py --frames 1024 --chunk 1024 --steps 1 --warmup 1 --streams 1 --pipelined-steps 0 --cpu-frames 0 --host-frames 0 --pose-frames 0 --latency-calls 0"
rm -rf gpurun_out/tf_fetch gpurun_out/tf_write
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d gpurun_out/tf_fetch -- $B > gpurun_out/tf_fetch.log 2>&1 || tail -3 gpurun_out/tf_fetch.log
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d gpurun_out/tf_write -- $B > gpurun_out/tf_write.log 2>&1 || tail -3 gpurun_out/tf_write.log
python3 tools/pmc_traffic.py gpurun_out/tf_fetch gpurun_out/tf_write 1024 $TAG | tail -12
cp profiles/${TAG:0:3}_pmc_traffic.json gpurun_out/
rm -rf gpurun_out/tf_fetch gpurun_out/tf_write
