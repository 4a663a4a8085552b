#!/bin/bash
# rocprofv3 kernel trace of BASELINE configs[3] (vitl14 -> swin_tiny, 518 px, bs 16) under the autocast policy, one stream;
# summary -> profiles/r03_config3_policy_*
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r03c3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
DKD_ASYNC_WGRAD=0 DKD_TEACHER_STREAM=0 DKD_CHAIN_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial -- python3 $R/scripts/time_configs.py --configs 3p --steps 2 --warmup 1 > $O/serial.log 2>&1 && echo pass
cd $R
python3 scripts/summarize_rocprof.py $O/serial r03_config3_policy 3 "\`DKD_ASYNC_WGRAD=0 DKD_TEACHER_STREAM=0 DKD_CHAIN_STREAMS=0 rocprofv3 --kernel-trace --stats -- python3 scripts/time_configs.py --configs 3p --steps 2 --warmup 1\`: BASELINE configs[3] (dinov2_vitl14 -> swin_tiny, 518 px, bs 16) with the whole step under the bf16 autocast policy, every kernel on one stream; 3 steps." > /dev/null
mkdir -p gpurun_out/profiles_r03 && cp profiles/r03_config3_policy* gpurun_out/profiles_r03/
tail -2 $O/serial.log
