#!/bin/bash
# Counter evidence for the UNet forward at the benchmark batch (run on the GPU box via gpurun, from the repo root):
#   tools/refresh_counters.sh <tag> [bf16|f16|bf16x3]  -> gpurun_out/<tag>_{traffic,mfma_util}.json (+ the kernel stats of the same workload)
# Copy them to profiles/roundN_{traffic,mfma_util}.json: bench.py reports roofline.traffic / roofline.mfma_busy only from records
# whose libsdn.so sha256 equals the running library's.  Counters are collected in their own passes (--pmc + --kernel-trace only).
set -eo pipefail
TAG=${1:-counters}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
RAW=/tmp/sdn_ctr_$TAG
rm -rf $RAW && mkdir -p $RAW
trap 'rc=$?; if [ $rc -ne 0 ]; then echo "refresh_counters: FAILED (rc $rc)"; tail -20 $OUT/${TAG}_ctr.err; fi; rm -rf $RAW' EXIT
cd /tmp && export TMPDIR=/tmp
MODE=${2:-bf16}                                   # second argument: the plan (bf16 | f16 | bf16x3); bench.py reads the bf16 records only
W="python3 $ROOT/tools/unet_forward.py 2 192 $MODE"
rocprofv3 --kernel-trace --stats -d $RAW/st -o s --output-format csv -- $W > /dev/null 2> $OUT/${TAG}_ctr.err
cp $RAW/st/s_kernel_stats.csv $OUT/${TAG}_unet_kernel_stats.csv
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $RAW/p1 -o a --output-format csv -- $W > /dev/null 2>> $OUT/${TAG}_ctr.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace -d $RAW/p2 -o b --output-format csv -- $W > /dev/null 2>> $OUT/${TAG}_ctr.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $RAW/pf -o f --output-format csv -- $W > /dev/null 2>> $OUT/${TAG}_ctr.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $RAW/pw -o w --output-format csv -- $W > /dev/null 2>> $OUT/${TAG}_ctr.err
cd $ROOT/tools
python3 pmc_mfma.py $RAW/p1/a_counter_collection.csv $RAW/p2/b_counter_collection.csv $OUT/${TAG}_mfma_util.json 10
python3 pmc_traffic.py $RAW/pf/f_counter_collection.csv $RAW/pw/w_counter_collection.csv $OUT/${TAG}_traffic.json "3 x 64 (UNet only, tools/unet_forward.py, $MODE plan)"
