#!/bin/bash
# HBM-side fetch traffic and duration of the fp32 LDS-DMA GEMM under different tile orders (DKD_TILE_ORDER, csrc/gemm_f32.hip)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_order
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for SHAPE in "16448 384 1536" "50176 192 768" "12544 384 1536"; do
for ORD in 0 1 2; do
  T=$(echo $SHAPE | tr ' ' x)_o$ORD
  DKD_TILE_ORDER=$ORD timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f_$T -- python3 $R/scripts/one_gemm.py $SHAPE 0 > $O/f_$T.log 2>&1 || echo "fetch pass failed $T"
  DKD_TILE_ORDER=$ORD timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_$T -- python3 $R/scripts/one_gemm.py $SHAPE 0 > $O/t_$T.log 2>&1 || echo "trace pass failed $T"
done
done
cd $R && python3 - <<'PY'
import csv, glob, collections, os
for d in sorted(glob.glob("gpurun_out/pmc_order/f_*")):
    if not os.path.isdir(d): continue
    tag = os.path.basename(d)[2:]
    per = collections.defaultdict(float)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm_f32" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
                per[r["Dispatch_Id"]] += float(r["Counter_Value"])
    vals = list(per.values())
    durs = []
    for f in glob.glob("gpurun_out/pmc_order/t_" + tag + "/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm_f32" in r["Name"]:
                durs.append((r["Name"][:40], float(r["AverageNs"]) / 1e3))
    # FETCH_SIZE is in KB on this image with the gfx950 half count (MI355X_MICROARCH.md): MB = 2 * KB / 1024
    print(tag, "fetch MB/launch", [round(2 * v / 1024, 1) for v in vals], "avg us", [(n, round(u, 1)) for n, u in durs])
PY
