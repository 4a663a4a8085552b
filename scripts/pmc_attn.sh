#!/bin/bash
# SQ counter passes over scripts/one_attn.py (one counter group per run); prints per-kernel averages.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_attn; mkdir -p $R/gpurun_out/pmc_attn
i=0
for C in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_MISC"; do
  i=$((i+1)); D=$R/gpurun_out/pmc_attn/g$i
  timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $D -- python3 $R/scripts/one_attn.py "$@" > $D.log 2>&1 || { tail -5 $D.log; echo "group $i failed: $C"; }
done
cd $R && python3 - <<'PY'
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_attn/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "attn_fwd" not in k: continue
        agg["attn_fwd"][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,d in agg.items():
    for c,v in sorted(d.items()): print(k, c, round(sum(v)/len(v)))
PY
