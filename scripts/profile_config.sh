#!/bin/bash
# rocprofv3 kernel stats of one step-loop of another BASELINE config: profile_config.sh <config id>
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_cfg$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/run -- python3 $R/scripts/bench_configs.py $1 > $O/run.log 2>&1
cd $R && python3 - <<PY
import csv,glob
f=glob.glob("$O/run/*/*kernel_stats.csv")[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:22]:
    print("%-110s %6s %9.2f ms %8.1f us %5.1f%%" % (r["Name"][:110], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot))
PY
