#!/usr/bin/env python3
"""Fold rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs) into profiles/<name>.json/.md.
HBM-side bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB * 1024: FETCH_SIZE reads exactly half of the bytes on gfx950
for 16-B/lane loads (MI355X_MICROARCH.md section HBM) - re-checked here on known byte counts (calibration dirs)."""
import csv, glob, json, os, sys, collections

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
fetch_dir, write_dir, name = sys.argv[1], sys.argv[2], sys.argv[3]
cal_f, cal_w = (sys.argv[4], sys.argv[5]) if len(sys.argv) > 5 else (None, None)


def per_kernel(d, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    f = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))[0]
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        k = k[k.find("::") + 2:] if "(anonymous namespace)::" in k else k
        k = k.split("<")[0].split("(")[0].strip()
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    return agg


F, W = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
out = {"unit": "bytes per launch (HBM/fabric side)", "correction": "FETCH_SIZE x 2 (gfx950), WRITE_SIZE exact", "kernels": {}}
for k in sorted(F, key=lambda k: -(2 * F[k][1] + W.get(k, [0, 0])[1])):
    n = F[k][0]
    fb, wb = 2 * F[k][1] * 1024 / n, W.get(k, [1, 0.0])[1] * 1024 / max(W.get(k, [1, 0])[0], 1)
    out["kernels"][k] = {"launches": n, "fetch_bytes": round(fb), "write_bytes": round(wb), "traffic_bytes": round(fb + wb)}
cal = []
if cal_f:
    for c, d in (("FETCH_SIZE", cal_f), ("WRITE_SIZE", cal_w)):
        f = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))[0]
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c and ("axpby" in r["Kernel_Name"] or "gemm_f32" in r["Kernel_Name"]):
                cal.append((c, "axpby 2x512MiB in / 512MiB out" if "axpby" in r["Kernel_Name"] else "gemm 200704x128x384 (A 308.3 MB read once, C 102.8 MB)",
                            float(r["Counter_Value"]) * 1024 / 1e6))
    out["calibration_MB_reported"] = cal
os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
json.dump(out, open(os.path.join(root, "profiles", name + ".json"), "w"), indent=1)
with open(os.path.join(root, "profiles", name + ".md"), "w") as f:
    f.write(f"# {name}: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) of `bench.py --steps 2 --warmup 1`\n\n")
    f.write("HBM-side traffic per launch = 2 x FETCH_SIZE (gfx950 half-count, verified below) + WRITE_SIZE.\n\n| kernel | launches | fetch MB | write MB | traffic MB |\n|---|---|---|---|---|\n")
    for k, v in list(out["kernels"].items())[:25]:
        f.write(f"| `{k}` | {v['launches']} | {v['fetch_bytes']/1e6:.1f} | {v['write_bytes']/1e6:.1f} | {v['traffic_bytes']/1e6:.1f} |\n")
    if cal:
        f.write("\nCalibration on known byte counts (raw counter x 1024, MB):\n\n")
        for c, what, mb in cal:
            f.write(f"* {c}: {what}: reported {mb:.1f} MB\n")
print(json.dumps({k: v for k, v in list(out["kernels"].items())[:6]}, indent=1))
