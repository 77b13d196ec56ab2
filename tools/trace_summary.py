#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel-trace CSV: per (kernel, grid) call count, mean / min duration."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
d = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"]
    if pat and pat not in n:
        continue
    k = (n.replace("void ", "").replace("unsigned short", "bf16").replace("(ConvArgs)", "").replace("(WgradArgs)", ""),
         r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
    d.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    v2 = sorted(v)
    print(f"{k[0][:60]:60s} grid {k[1]:>8s},{k[2]:>3s},{k[3]} n={len(v):3d} mean {sum(v)/len(v):8.1f} us  median {v2[len(v2)//2]:8.1f}  min {v2[0]:8.1f}")
