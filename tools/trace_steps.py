#!/usr/bin/env python3
"""Per-step GPU timeline of the LAST n steps of a rocprofv3 kernel trace of bench.py (steps delimited by the two adam_multi launches of a step):
wall, kernel-time sum, idle / one-kernel / overlapped time and the gap histogram, plus the kernels after which the GPU idles longest.
usage: trace_steps.py kernel_trace.csv nsteps"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
nsteps = int(sys.argv[2])
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
adam = [i for i, e in enumerate(ev) if e[2].startswith("adam_multi")]
a0, a1 = adam[-(2 * nsteps + 1)], adam[-1]
tstart, tend = ev[a0][1], ev[a1][1]
sel = [e for e in ev if e[0] >= tstart and e[0] < tend]
pts = []
for i, (s, e, n) in enumerate(sel):
    pts.append((s, 1, i)); pts.append((e, -1, i))
pts.sort()
act = 0; last = tstart; idle = one = multi = 0; gaps = []; lastend = None
after = collections.defaultdict(lambda: [0, 0.0])
for t, k, i in pts:
    dt = t - last
    if dt > 0:
        if act == 0:
            idle += dt; gaps.append(dt / 1e3)
            if lastend is not None:
                a = after[(sel[lastend][2][:60], sel[i][2][:60])]; a[0] += 1; a[1] += dt / 1e3
        elif act == 1: one += dt
        else: multi += dt
    last = t; act += k
    if k < 0: lastend = i
ksum = sum(e[1] - e[0] for e in sel) / 1e6
print(f"{nsteps} steps: wall/step {(tend - tstart) / 1e6 / nsteps:.3f} ms, kernels/step {len(sel) / nsteps:.1f}, kernel-time sum/step {ksum / nsteps:.3f} ms, "
      f"idle {idle / 1e6 / nsteps:.3f}, one kernel running {one / 1e6 / nsteps:.3f}, >= 2 running {multi / 1e6 / nsteps:.3f}")
for lo, hi in [(0, 1), (1, 2), (2, 5), (5, 10), (10, 20), (20, 50), (50, 1e9)]:
    g = [x for x in gaps if lo <= x < hi]
    print(f"   gaps {lo:>3}-{hi if hi < 1e9 else 'inf':>4} us: per step n {len(g) / nsteps:6.1f} sum {sum(g) / 1e3 / nsteps:.3f} ms")
print("largest idle by (kernel before, kernel after), us per step:")
for (a, b), (n, t) in sorted(after.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"  {t / nsteps:7.1f} us  n/step {n / nsteps:5.1f}  {a}  ->  {b}")
