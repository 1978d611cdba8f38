#!/bin/bash
# HBM bytes of the threshold+label sweep (GPU box): FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes (own runs, --kernel-trace
# only) of one bench pass, summarised by tools/pmc_traffic.py into profiles/<tag>_pmc_traffic.json.
# usage: tools/pmc_traffic.sh r05                      (1024 frames of 1920x1080)
#        tools/pmc_traffic.sh r05_4k 3840x2160 256     (tag, frame size, frames per launch)
TAG=${1:-r05}
SIZE=${2:-1920x1080}
N=${3:-1024}
export TMPDIR=/tmp
B="python3 bench.py --size $SIZE --frames $N --chunk $N --steps 1 --warmup 1 --streams 1 --pipelined-steps 0 --cpu-frames 0 --host-frames 0 --pose-frames 0 --latency-calls 0"
rm -rf gpurun_out/tf_fetch gpurun_out/tf_write
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d gpurun_out/tf_fetch -- $B > gpurun_out/tf_fetch.log 2>&1 || tail -3 gpurun_out/tf_fetch.log
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d gpurun_out/tf_write -- $B > gpurun_out/tf_write.log 2>&1 || tail -3 gpurun_out/tf_write.log
python3 tools/pmc_traffic.py gpurun_out/tf_fetch gpurun_out/tf_write $N $TAG $SIZE | tail -12
cp profiles/${TAG}_pmc_traffic.json gpurun_out/
rm -rf gpurun_out/tf_fetch gpurun_out/tf_write
