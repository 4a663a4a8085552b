#!/bin/bash
# SQ counter passes over scripts/one_jpeg_decode.py (one counter group per run); prints per-launch averages of the two JPEG kernels.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_jpeg; mkdir -p $R/gpurun_out/pmc_jpeg
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_SMEM"; do
  i=$((i+1)); D=$R/gpurun_out/pmc_jpeg/g$i
  timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -- python3 $R/scripts/one_jpeg_decode.py > $D.log 2>&1 || { tail -3 $D.log; echo "group $i failed: $C"; }
done
cd $R && python3 - <<'PY'
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_jpeg/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "jpeg" not in k: continue
        import re; agg[re.search(r"jpeg_\w+", k).group(0)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,d in agg.items():
    print(k)
    for c,v in sorted(d.items()): print("    %-32s %14.0f" % (c, sum(v)/len(v)))
PY
