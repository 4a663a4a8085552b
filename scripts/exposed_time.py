"""From a rocprofv3 kernel trace of bench.py (default, multi-stream mode): how much of the last step's wall time has NO
matrix-core kernel (GEMM / attention) in flight, and which kernels occupy that exposed time.
    python scripts/exposed_time.py <kernel_trace.csv> [steps_in_trace]"""
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]))
rows.sort()
# the last step: from the last-but-one adamw group to the last one
adam = [i for i, r in enumerate(rows) if "adamw_kernel" in r[2]]
ends = [rows[i][1] for i in adam]
groups = []
for e in ends:  # adamw launches come 3 per step, close together
    if not groups or e - groups[-1][-1] > 5_000_000: groups.append([e])
    else: groups[-1].append(e)
t0, t1 = groups[-2][-1], groups[-1][-1]
step = [r for r in rows if r[0] >= t0 and r[1] <= t1]
print("step wall %.2f ms, %d kernels, queues %s" % ((t1 - t0) / 1e6, len(step), sorted({r[3] for r in step})))
is_mfma = lambda n: "gemm_" in n or "attn_" in n or "winattn" in n
ev = []
for s, e, n, q in step:
    ev.append((s, 1, is_mfma(n), n)); ev.append((e, -1, is_mfma(n), n))
ev.sort(key=lambda x: (x[0], x[1]))
mf = 0; live = collections.Counter(); last = t0
exposed = collections.Counter(); idle = 0; mfma_time = 0
for t, d, m, n in ev:
    dt = t - last
    if dt > 0:
        if mf > 0: mfma_time += dt
        elif live:
            for k in live: exposed[k] += dt / len(live)
        else: idle += dt
    last = t
    if m: mf += d
    else:
        live[n] += d
        if live[n] == 0: del live[n]
tot = t1 - t0
print("matrix-core kernel in flight: %.2f ms (%.1f %%)   none, other kernels running: %.2f ms   nothing running: %.2f ms" % (
    mfma_time / 1e6, 100 * mfma_time / tot, sum(exposed.values()) / 1e6, idle / 1e6))
for k, v in exposed.most_common(18):
    print("  %7.3f ms  %s" % (v / 1e6, k[:110]))
# the largest gaps with nothing running at all: what ends before / starts after them
gaps = []
cur_end, prev_n = t0, '(step start)'
for s_, e_, n_, q_ in sorted(step):
    if s_ > cur_end: gaps.append((s_ - cur_end, cur_end - t0, prev_n, n_))
    if e_ > cur_end: cur_end, prev_n = e_, n_
print("largest idle gaps:")
print("  %d gaps, %.2f ms in total; %d gaps over 20 us = %.2f ms" % (len(gaps), sum(g[0] for g in gaps) / 1e6, sum(g[0] > 20000 for g in gaps), sum(g[0] for g in gaps if g[0] > 20000) / 1e6))
for d, at, a, b in sorted(gaps, reverse=True)[:25]:
    print("  %6.1f us at +%.2f ms   after %s   before %s" % (d / 1e3, at / 1e6, a[:50], b[:50]))
