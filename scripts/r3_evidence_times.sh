#!/bin/bash
# step times only (after the host run-ahead bound of DistillationStepper): configs[2] A/B against DKD_CONV_IGEMM=0, all five configs, bench line
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/profiles_r03
DKD_CONV_IGEMM=0 python3 scripts/time_configs.py --configs 2 --steps 10 2> /dev/null | grep '"config"' | sed 's/^{/{"DKD_CONV_IGEMM": 0, /' > profiles/r03_config2_igemm_ab.jsonl
python3 scripts/time_configs.py --configs 2 --steps 10 2> /dev/null | grep '"config"' | sed 's/^{/{"DKD_CONV_IGEMM": 1, /' >> profiles/r03_config2_igemm_ab.jsonl
python3 scripts/time_configs.py --configs 0,1,2,3,4 2> /dev/null | grep '"config"' > profiles/r03_config_times.jsonl
python3 bench.py > gpurun_out/r03_bench_final.log 2> gpurun_out/r03_bench_final.err && tail -1 gpurun_out/r03_bench_final.log > profiles/r03_bench_bs64.json
cp profiles/r03_config2_igemm_ab.jsonl profiles/r03_config_times.jsonl profiles/r03_bench_bs64.json gpurun_out/profiles_r03/
python3 - <<'PY'
import json
for f in ("profiles/r03_config2_igemm_ab.jsonl", "profiles/r03_config_times.jsonl"):
    for l in open(f):
        d = json.loads(l)
        print(d.get("DKD_CONV_IGEMM", "-"), d["config"], d["precision"], d["ms_per_step"], d["images_per_s"], d["peak_gib"], d["device_allocs_in_timed_loop"], d["host_ms_per_step_max"])
b = json.loads(open("profiles/r03_bench_bs64.json").read())
print("bench", b["value"], b["ms_per_step"], b["roofline"]["frac"], {k: v["ms_per_step"] for k, v in b.items() if k.startswith("alt_")})
PY
