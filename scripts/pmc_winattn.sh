#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc_wa
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU"; do
  i=$((i+1)); D=$R/gpurun_out/pmc_wa/g$i
  timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $D -- python3 $R/scripts/bench_winattn.py > $D.log 2>&1 || { tail -5 $D.log; echo "group $i failed"; }
done
cd $R && python3 - <<'PY'
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_wa/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "winattn" not in k: continue
        k="fwd" if "fwd" in k else ("bwd" if "bwd_kernel" in k else "dbias")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,d in agg.items():
    print(k, {c: round(sum(v)/len(v)) for c,v in sorted(d.items())})
PY
