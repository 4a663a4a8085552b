#!/bin/bash
# rocprofv3 passes over bench.py on the GPU box; summaries land in profiles/ (tracked).  Usage: profile_bench.sh <tag>
#   1. --kernel-trace --stats, default mode (weight gradients co-scheduled on the side stream)
#   2. --kernel-trace --stats with DKD_ASYNC_WGRAD=0 (one stream: un-stretched per-kernel durations)
#   3. --pmc FETCH_SIZE and 4. --pmc WRITE_SIZE (separate passes, counters only)
set -e
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="$R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-roofline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/async -- python3 $ARGS > $O/async.log 2>&1
echo "pass 1 done"
DKD_ASYNC_WGRAD=0 DKD_TEACHER_STREAM=0 DKD_CHAIN_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial -- python3 $ARGS > $O/serial.log 2>&1
echo "pass 2 done"
ARGS2="$R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-roofline"
DKD_ASYNC_WGRAD=0 DKD_TEACHER_STREAM=0 DKD_CHAIN_STREAMS=0 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $ARGS2 > $O/fetch.log 2>&1
echo "pass 3 done"
DKD_ASYNC_WGRAD=0 DKD_TEACHER_STREAM=0 DKD_CHAIN_STREAMS=0 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $ARGS2 > $O/write.log 2>&1
echo "pass 4 done"
cd $R
python3 scripts/summarize_rocprof.py $O/async ${TAG}_bench_bs64 4 "\`rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-roofline\` (default mode: weight-gradient GEMMs co-scheduled on a side stream, so per-kernel durations of overlapping kernels are stretched)."
python3 scripts/summarize_rocprof.py $O/serial ${TAG}_bench_bs64_serial 4 "Same command with \`DKD_ASYNC_WGRAD=0 DKD_TEACHER_STREAM=0 DKD_CHAIN_STREAMS=0\` (every kernel on one stream: per-kernel durations are not stretched by co-scheduling; this is the view bench.py's roofline.achieved corresponds to)."
python3 scripts/pmc_summarize.py $O/fetch $O/write ${TAG}_pmc_traffic
mkdir -p gpurun_out/profiles_$TAG && cp profiles/${TAG}_* gpurun_out/profiles_$TAG/
ls profiles/
