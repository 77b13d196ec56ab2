#!/usr/bin/env python3
"""GPU busy / idle analysis of a rocprofv3 kernel-trace CSV over its last `frac` of wall time (steady state):
busy time, idle time, the distribution of inter-kernel gaps, and the kernels with the largest total time.
usage: trace_gaps.py kernel_trace.csv [frac=0.5]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
t0, t1 = ev[0][0], max(e[1] for e in ev)
cut = t1 - (t1 - t0) * frac
ev = [e for e in ev if e[0] >= cut]
wall = (max(e[1] for e in ev) - ev[0][0]) / 1e6
busy = 0.0; cur_end = ev[0][0]; gaps = []
for s, e, _ in ev:
    if s > cur_end:
        gaps.append((s - cur_end) / 1e3)
    busy += max(0, e - max(s, cur_end)) / 1e6
    cur_end = max(cur_end, e)
print(f"window {wall:.2f} ms, {len(ev)} kernels, busy {busy:.2f} ms ({100 * busy / wall:.1f}%), idle {wall - busy:.2f} ms")
gaps.sort()
if gaps:
    n = len(gaps)
    print(f"gaps: n={n} median {gaps[n // 2]:.1f} us  p90 {gaps[int(n * .9)]:.1f}  p99 {gaps[int(n * .99)]:.1f}  max {gaps[-1]:.1f}  sum {sum(gaps) / 1e3:.2f} ms")
    for lo, hi in [(0, 2), (2, 5), (5, 10), (10, 20), (20, 50), (50, 1e9)]:
        g = [x for x in gaps if lo <= x < hi]
        print(f"  gaps {lo:>3}-{hi if hi < 1e9 else 'inf':>4} us: n={len(g):5d} sum {sum(g) / 1e3:7.2f} ms")
agg = collections.defaultdict(lambda: [0, 0.0])
for s, e, n in ev:
    a = agg[n]; a[0] += 1; a[1] += (e - s) / 1e6
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"{t:8.2f} ms  n={c:5d}  {n[:110]}")

# ---- exclusive time: how long each kernel is the ONLY one running (a proxy for the critical path when several streams
# overlap), how long >= 2 kernels overlap, per-queue busy time
pts = []
for i, (s, e, n) in enumerate(ev):
    pts.append((s, 1, i)); pts.append((e, 0, i))
pts.sort()
active = set(); last = pts[0][0]; excl = collections.defaultdict(float); multi = 0.0
for t, kind, i in pts:
    if t > last and active:
        if len(active) == 1:
            excl[ev[next(iter(active))][2]] += (t - last) / 1e6
        else:
            multi += (t - last) / 1e6
    last = t
    if kind: active.add(i)
    else: active.discard(i)
print(f"exclusive (one kernel running) {sum(excl.values()):.2f} ms, overlapped (>=2 running) {multi:.2f} ms")
for n, t in sorted(excl.items(), key=lambda kv: -kv[1])[:25]:
    print(f"{t:8.2f} ms exclusive  (total {agg[n][1]:6.2f}, n={agg[n][0]:4d})  {n[:100]}")
qcol = next((c for c in ("Queue_Id", "Stream_Id") if c in rows[0]), None)
if qcol:
    q = collections.defaultdict(lambda: [0, 0.0])
    for r in rows:
        if int(r["Start_Timestamp"]) >= cut:
            a = q[r[qcol]]; a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    for k, (c, t) in sorted(q.items(), key=lambda kv: -kv[1][1]):
        print(f"{qcol} {k}: {c} kernels, {t:.2f} ms busy")
