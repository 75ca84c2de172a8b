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
# (a pass with TCP_* counters made the run crawl for minutes on this pool: not collected)
python3 tools/pmc_summary.py $OUT/a $OUT/b $OUT/c 2>&1 | grep -v "rocclr\|layout"
