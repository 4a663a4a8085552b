#!/usr/bin/env python3
"""Turn a rocprofv3 --kernel-trace --stats output dir into a committed summary under profiles/ (csv + markdown)."""
import csv, glob, os, shutil, sys

src, name = sys.argv[1], sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 0
note = sys.argv[4] if len(sys.argv) > 4 else ""
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
stats = sorted(glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True))[0]
os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
shutil.copy(stats, os.path.join(root, "profiles", name + "_kernel_stats.csv"))
rows = list(csv.DictReader(open(stats)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(os.path.join(root, "profiles", name + "_kernel_stats.md"), "w") as f:
    f.write(f"# {name}_kernel_stats: rocprofv3 --kernel-trace --stats\n\n{note}\n\n")
    f.write(f"Total kernel time {tot/1e6:.2f} ms over the whole run" + (f" = {tot/1e6/steps:.2f} ms per step ({steps} steps incl. warm-up)" if steps else "") + ".\n\n")
    f.write("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
    for r in rows[:40]:
        f.write("| `%s` | %s | %.2f | %.1f | %.1f |\n" % (r["Name"][:110].replace("|", "/"), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                   float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
print(open(os.path.join(root, "profiles", name + "_kernel_stats.md")).read()[:3000])
