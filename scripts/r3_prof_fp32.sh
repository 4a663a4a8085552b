#!/bin/bash
# rocprofv3 kernel trace of the fp32 bench step (default streams) + exposed-time analysis; fp32 step and the autocast step
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r03x
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/f32 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-roofline > $O/f32.log 2>&1 && echo pass1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ac -- python3 $R/scripts/step_autocast.py 4 > $O/ac.log 2>&1 && echo pass2
cd $R
python3 scripts/exposed_time.py $(ls $O/f32/*/*kernel_trace.csv | head -1) > $O/exposed_f32.log 2>&1
python3 scripts/exposed_time.py $(ls $O/ac/*/*kernel_trace.csv | head -1) > $O/exposed_ac.log 2>&1
rm -f $O/*/*/*kernel_trace.csv
cat $O/exposed_f32.log | head -40
