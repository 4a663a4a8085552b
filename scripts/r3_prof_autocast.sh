#!/bin/bash
# rocprofv3 kernel trace of the mixed-precision step (default streams, then one stream); summaries -> profiles/r03_autocast_*
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r03ac
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/async -- python3 $R/scripts/step_autocast.py 4 > $O/async.log 2>&1 && echo pass1
DKD_ASYNC_WGRAD=0 DKD_TEACHER_STREAM=0 DKD_CHAIN_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial -- python3 $R/scripts/step_autocast.py 4 > $O/serial.log 2>&1 && echo pass2
cd $R
python3 scripts/summarize_rocprof.py $O/async r03_autocast 5 "\`rocprofv3 --kernel-trace --stats -- python3 scripts/step_autocast.py 4\`: the whole config-[1] step (bs 64) under the bf16 autocast policy, default mode (teacher / frequency chain / weight gradients on side streams: durations of overlapping kernels are stretched); 5 steps incl. warm-up." > /dev/null
python3 scripts/summarize_rocprof.py $O/serial r03_autocast_serial 5 "Same command with \`DKD_ASYNC_WGRAD=0 DKD_TEACHER_STREAM=0 DKD_CHAIN_STREAMS=0\` (every kernel on one stream: un-stretched per-kernel durations; weight gradients then return through autograd's AccumulateGrad, hence the ATen add launches)." > /dev/null
mkdir -p gpurun_out/profiles_r03 && cp profiles/r03_autocast* gpurun_out/profiles_r03/
tail -2 $O/async.log; tail -2 $O/serial.log
