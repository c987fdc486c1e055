#!/bin/bash
# Re-collects the evidence bench.py's roofline block cites (run on the GPU box via gpurun, from the repo root):
#   tools/refresh_profiles.sh <tag>      -> gpurun_out/<tag>_{kernel_stats.csv,bench_under_rocprof.json,bench.json,traffic.json}
# Copy the four files into profiles/ (traffic.json as profiles/roundN_traffic.json: bench.py reads the newest one and reports it only
# when its libsdn.so sha256 equals the running library) to have them judged.  Pass SDN_GIT_HEAD=$(git rev-parse HEAD) through gpurun.
set -eo pipefail
TAG=${1:-refresh}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
# the raw traces are far over gpurun's 64 MiB return limit: keep them out of gpurun_out/ whatever happens
RAW=/tmp/sdn_prof_$TAG
rm -rf $RAW && mkdir -p $RAW
trap 'rc=$?; if [ $rc -ne 0 ]; then echo "refresh_profiles: FAILED (rc $rc)"; tail -20 $OUT/${TAG}_rocprof.err; fi; rm -rf $RAW' EXIT
cd /tmp && export TMPDIR=/tmp
# 1. kernel trace + stats (timing): the dominant kernel's AverageNs must agree with bench.py's live HIP-event average
rocprofv3 --kernel-trace --stats -d $RAW/stats -o s --output-format csv -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_rocprof.err
cp $RAW/stats/s_kernel_stats.csv $OUT/${TAG}_kernel_stats.csv
# 2. PMC passes, counters only (separate runs: TCC has few slots; never combined with other trace domains).  SKIP_PMC=1: the
#    counter records come from tools/refresh_counters.sh (UNet-only workload) instead
if [ "${SKIP_PMC:-0}" != "1" ]; then
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $RAW/pmc_f -o f --output-format csv -- python3 $ROOT/bench.py --steps 1 --warmup 0 --inference-steps 2 --no-cpu-baseline --no-extras > /dev/null 2>> $OUT/${TAG}_rocprof.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $RAW/pmc_w -o w --output-format csv -- python3 $ROOT/bench.py --steps 1 --warmup 0 --inference-steps 2 --no-cpu-baseline --no-extras > /dev/null 2>> $OUT/${TAG}_rocprof.err
python3 $ROOT/tools/pmc_traffic.py $RAW/pmc_f/f_counter_collection.csv $RAW/pmc_w/w_counter_collection.csv $OUT/${TAG}_traffic.json "3 x 64 (e2e default)"
fi
# 3. the plain default bench line
cd $ROOT && python3 bench.py > $OUT/${TAG}_bench.json 2>> $OUT/${TAG}_rocprof.err
tail -c 300 $OUT/${TAG}_bench.json
