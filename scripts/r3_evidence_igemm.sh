#!/bin/bash
# Tracked evidence after the implicit-GEMM convolution / MobileNetV3 / MiT work, one box visit: BASELINE configs[2] kernel stats in fp32
# and under the policy (the im2col / col2im rows are gone from the forward and the stride-1 input gradients), the A/B against
# DKD_CONV_IGEMM=0, step times of all five configs, and the bench line.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/profiles_r03
bash scripts/profile_config.sh 3 > gpurun_out/r03_config2.log 2>&1 && python3 scripts/summarize_rocprof.py gpurun_out/prof_cfg3/run r03_config2 0 "\`rocprofv3 --kernel-trace --stats -- python3 scripts/bench_configs.py 3\`: BASELINE configs[2], dinov2_vitb14 -> resnet_50, 224x224, bs 128, fp32 (warm-up + timed steps of that script), with the implicit-GEMM convolution (conv_igemm_f32_kernel: forward, stride-1 input gradient, weight gradient); the remaining im2col_nhwc / col2im rows are the three stride-2 3x3 input gradients and the RGB stem." > /dev/null && echo config2-fp32
bash scripts/r3_prof_config2p.sh > gpurun_out/r3_final_c2p.log 2>&1; tail -1 gpurun_out/r3_final_c2p.log
DKD_CONV_IGEMM=0 python3 scripts/time_configs.py --configs 2 --steps 8 2> /dev/null | grep '"config"' | sed 's/^{/{"DKD_CONV_IGEMM": 0, /' > profiles/r03_config2_igemm_ab.jsonl
python3 scripts/time_configs.py --configs 2 --steps 8 2> /dev/null | grep '"config"' | sed 's/^{/{"DKD_CONV_IGEMM": 1, /' >> profiles/r03_config2_igemm_ab.jsonl
python3 scripts/time_configs.py --configs 0,1,2,3,4 2> /dev/null | grep '"config"' > profiles/r03_config_times.jsonl
python3 bench.py > gpurun_out/r03_bench_final.log 2> gpurun_out/r03_bench_final.err && tail -1 gpurun_out/r03_bench_final.log > profiles/r03_bench_bs64.json
cp profiles/r03_* gpurun_out/profiles_r03/
rm -rf gpurun_out/prof_cfg3/run/*/*kernel_trace.csv gpurun_out/prof_r03c2p/serial/*/*kernel_trace.csv
cut -c1-120 profiles/r03_config2_igemm_ab.jsonl profiles/r03_config_times.jsonl
cut -c1-200 profiles/r03_bench_bs64.json
