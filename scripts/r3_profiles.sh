#!/bin/bash
# Round-3 tracked evidence in one box visit: fp32 bench profiles (async / serial / PMC traffic), config [2] kernel stats,
# PMC traffic of the bf16 ring GEMM inside the autocast step.  Summaries land in profiles/ (copied back through gpurun_out/).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash scripts/profile_bench.sh r03 > gpurun_out/r03_profile_bench.log 2>&1 && echo bench-profiles
# BASELINE configs[2] (dinov2_vitb14 -> resnet_50, bs 128) is id "3" of scripts/bench_configs.py
bash scripts/profile_config.sh 3 > gpurun_out/r03_config2.log 2>&1 && python3 scripts/summarize_rocprof.py gpurun_out/prof_cfg3/run r03_config2 0 "\`rocprofv3 --kernel-trace --stats -- python3 scripts/bench_configs.py 3\`: BASELINE configs[2], dinov2_vitb14 -> resnet_50, 224x224, bs 128, fp32 (warm-up + timed steps of that script); im2col_nhwc / col2im rows are the materialised 3x3 patches (VERDICT r02 item 5)." > /dev/null && echo config2
O=$R/gpurun_out/prof_r03ac
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
DKD_ASYNC_WGRAD=0 DKD_TEACHER_STREAM=0 DKD_CHAIN_STREAMS=0 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/scripts/step_autocast.py 2 > $O/fetch.log 2>&1 && echo fetch
DKD_ASYNC_WGRAD=0 DKD_TEACHER_STREAM=0 DKD_CHAIN_STREAMS=0 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/scripts/step_autocast.py 2 > $O/write.log 2>&1 && echo write
cd $R
python3 scripts/pmc_summarize.py $O/fetch $O/write r03_autocast_pmc_traffic > /dev/null && echo pmc16
sed -i 's/of `bench.py --steps 2 --warmup 1`/of `scripts\/step_autocast.py 2` (the mixed-precision step, one stream)/' profiles/r03_autocast_pmc_traffic.md
mkdir -p gpurun_out/profiles_r03 && cp profiles/r03_* gpurun_out/profiles_r03/
rm -rf gpurun_out/prof_r03/*/*/*_counter_collection.csv gpurun_out/prof_r03ac/*/*/*_counter_collection.csv gpurun_out/prof_r03/*/*/*kernel_trace.csv
ls profiles | grep r03
