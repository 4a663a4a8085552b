"""Register / LDS / occupancy of every kernel of one csrc file as hipcc sees it (-Rpass-analysis=kernel-resource-usage), one line per kernel.
Runs without a GPU.      python scripts/kernel_resources.py gemm_x3.hip [substring of the kernel name]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dinov2_distillation_amd import build  # noqa: E402

src = os.path.join(build.CSRC, sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = [build._hipcc(), *build.FLAGS, *build.FILE_FLAGS.get(os.path.basename(src), []), "-Rpass-analysis=kernel-resource-usage", "-c", src,
       "-o", "/tmp/_kres.o"]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = {}
for line in err.splitlines():
    m = re.search(r"remark: +([A-Za-z \[\]/]+): (.*?) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()}
    else:
        cur[k] = v
    if k.startswith("LDS Size"):
        if flt in cur["name"]:
            nm = re.sub(r"\(anonymous namespace\)::", "", cur["name"]).split("(")[0]
            print("%-90s VGPR %3s AGPR %3s SGPR %3s spill %s/%s occ %s LDS %s" % (
                nm[:90], cur.get("VGPRs"), cur.get("AGPRs"), cur.get("TotalSGPRs"), cur.get("VGPR Spill", cur.get("VGPRs Spill")),
                cur.get("SGPR Spill", cur.get("SGPRs Spill")), cur.get("Occupancy [waves/SIMD]"), cur.get("LDS Size [bytes/block]")))
