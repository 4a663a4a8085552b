#!/bin/bash
# Round-3 tracked evidence, all in one box visit (re-run after the last code change): fp32 bench profiles + PMC traffic + config [2]
# fp32 (r3_profiles.sh), the mixed-precision step (r3_prof_autocast.sh), configs [2] / [3] under the policy, step times of all configs.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/profiles_r03
bash scripts/r3_profiles.sh > gpurun_out/r3_final_profiles.log 2>&1; tail -3 gpurun_out/r3_final_profiles.log
bash scripts/r3_prof_autocast.sh > gpurun_out/r3_final_autocast.log 2>&1; tail -2 gpurun_out/r3_final_autocast.log
bash scripts/r3_prof_config3.sh > gpurun_out/r3_final_c3.log 2>&1; tail -1 gpurun_out/r3_final_c3.log
bash scripts/r3_prof_config2p.sh > gpurun_out/r3_final_c2p.log 2>&1; tail -1 gpurun_out/r3_final_c2p.log
python3 scripts/time_configs.py --configs 0,1,2,3,4 2> /dev/null | grep '"config"' > profiles/r03_config_times.jsonl
cp profiles/r03_* gpurun_out/profiles_r03/
cat profiles/r03_config_times.jsonl | cut -c1-110
