#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: mean counter value per (kernel, grid)."""
import collections
import csv
import sys

pat = sys.argv[2] if len(sys.argv) > 2 else ""
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.OrderedDict()
for r in rows:
    if pat and pat not in r["Kernel_Name"]:
        continue
    key = (r["Kernel_Name"].replace("void ", "").replace("unsigned short", "bf16").replace("(ConvArgs)", "")[:48], r["Grid_Size"],
           r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"])
    d.setdefault(key, collections.OrderedDict()).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for key, c in d.items():
    print(key)
    print("    " + "  ".join(f"{k}={sum(v)/len(v):.3g}" for k, v in c.items()))
