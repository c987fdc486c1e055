#!/bin/bash
# SQ counters of one GEMM / attention shape in two rocprofv3 --pmc passes (8 SQ slots each); run on the GPU box.
# usage: tools/pmc_gemm.sh <out.txt> <target script> SHAPE...
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=$1; TARGET=$2; shift 2
for S in "$@"; do
  SHAPE=$S rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d /tmp/p1_$S -o a --output-format csv -- python3 $TARGET > /tmp/p1_$S.log 2>&1
  SHAPE=$S rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_SCA --kernel-trace -d /tmp/p2_$S -o b --output-format csv -- python3 $TARGET > /tmp/p2_$S.log 2>&1
  echo "== $S" >> $OUT
  python3 tools/pmc_sum.py /tmp/p1_$S /tmp/p2_$S >> $OUT
done
