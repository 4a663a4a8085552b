#!/bin/bash
# SQ / TCC counter passes over scripts/one_gemm_bf16.py (one counter group per run); prints per-kernel averages.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc_b16
i=0
for C in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_ANY" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_REQ_sum"; do
  i=$((i+1)); D=$R/gpurun_out/pmc_b16/g$i
  timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -- python3 $R/scripts/one_gemm_bf16.py "$@" > $D.log 2>&1 || { tail -3 $D.log; echo "group $i failed: $C"; }
done
cd $R && python3 - <<'PY'
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_b16/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "gemm_bf16" in k: k="ours"
        elif "Cijk" in k: k="lib "+k[k.find("MT"):k.find("MT")+16]
        else: continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,d in agg.items():
    print(k, {c: round(sum(v)/len(v)) for c,v in sorted(d.items())})
PY
