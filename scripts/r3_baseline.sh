#!/bin/bash
# round-3 starting point on one box: bench line, host overhead of the autocast step, launch census of it, GEMM yardsticks
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3a
mkdir -p $O
cd $R
python3 bench.py --steps 20 --warmup 5 --no-cpu-bs64 > $O/bench.json 2> $O/bench.err && echo bench done
python3 scripts/host_overhead_autocast.py > $O/ho.log 2>&1 && echo ho done
python3 scripts/bench_gemm_bf16_vendor.py > $O/gemm16.log 2>&1 && echo gemm16 done
python3 scripts/bench_gemm_shapes.py > $O/gemm32.log 2>&1 && echo gemm32 done
cd /tmp && export TMPDIR=/tmp
PROJECTOR=bf16_autocast rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ac -- python3 $R/scripts/step_autocast.py 4 > $O/prof_ac.log 2>&1 && echo prof done
