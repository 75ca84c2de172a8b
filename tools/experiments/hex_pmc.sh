#!/bin/bash
# Instruction and cache counters of the hexahedral lane kernels:  bash tools/experiments/hex_pmc.sh <tag> <P> <N>
set -u
TAG=${1:-hexpmc}; P=${2:-1}; N=${3:-96}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export SEIGEN_HIP_PATH=${SEIGEN_HIP_PATH:-lane}
CMD="python3 tools/experiments/hex_throughput.py $P $N quadrilateral"
rocprofv3 --output-format csv --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/a -o run -- $CMD > $OUT/a.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAVES GRBM_GUI_ACTIVE -d $OUT/b -o run -- $CMD > $OUT/b.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum -d $OUT/c -o run -- $CMD > $OUT/c.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum -d $OUT/d -o run -- $CMD > $OUT/d.log 2>&1
python3 tools/pmc_summary.py $OUT/a $OUT/b $OUT/c $OUT/d 2>&1 | grep -v "rocclr\|layout"
tail -2 $OUT/d.log
