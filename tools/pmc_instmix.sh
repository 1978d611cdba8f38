#!/bin/bash
# Instruction-mix / stall / TA / LDS counters per kernel (GPU box): four rocprofv3 --pmc passes (<= 8 SQ counters each, own
# runs with --kernel-trace only) of one 1024-frame bench pass, summarised by tools/pmc_instmix.py into profiles/<tag>_pmc_instmix.json
# usage: tools/pmc_instmix.sh r02a
TAG=${1:-r02}
SIZE=${2:-1920x1080}   # tools/pmc_instmix.sh r04_4k 3840x2160 -> profiles/r04_4k_pmc_instmix.json
export TMPDIR=/tmp
B="python3 bench.py --size $SIZE --frames 1024 --chunk 1024 --steps 1 --warmup 1 --streams 1 --pipelined-steps 0 --cpu-frames 0 --host-frames 0 --pose-frames 0 --latency-calls 0"
P1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P2="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_CVT"
P3="SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY"
P4="TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS_F32"
rm -rf gpurun_out/im_*
i=0
for set in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $set -d gpurun_out/im_$i -- $B > gpurun_out/im_$i.log 2>&1 || tail -3 gpurun_out/im_$i.log
done
python3 tools/pmc_instmix.py $TAG gpurun_out/im_*/
