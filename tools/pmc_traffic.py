#!/usr/bin/env python3
"""HBM traffic per launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) over the same command.

usage: pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> [steady_fraction=0.5]

Per kernel name: mean FETCH_SIZE and WRITE_SIZE per dispatch over the last `steady_fraction` of dispatches (steady state),
in bytes.  rocprofv3 reports both in KiB-sized units of 1024 bytes... no: in kilobytes as documented (x1024 here); on gfx950
FETCH_SIZE counts 64 bytes per 128-byte request of a wide coalesced streaming read, so the corrected read traffic is
2 x FETCH_SIZE (MI355X_MICROARCH.md, "HBM"); WRITE_SIZE is taken as reported (uncalibrated there).  Both raw values are
kept in the output next to the corrected total."""
import collections
import csv
import json
import sys


def per_kernel(path, counter, frac):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0)))
    rows = rows[int(len(rows) * (1 - frac)):]
    d = collections.defaultdict(list)
    for r in rows:
        d[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in d.items()}


def main():
    frac = float(sys.argv[4]) if len(sys.argv) > 4 else 0.5
    f = per_kernel(sys.argv[1], "FETCH_SIZE", frac)
    w = per_kernel(sys.argv[2], "WRITE_SIZE", frac)
    out = {}
    for k in sorted(set(f) | set(w), key=lambda k: -(f.get(k, (0, 0))[0] * f.get(k, (0, 0))[1])):
        fk, nf = f.get(k, (0.0, 0)); wk, nw = w.get(k, (0.0, 0))
        out[k] = {"fetch_size_raw_bytes": fk * 1024.0, "write_size_raw_bytes": wk * 1024.0,
                  "hbm_bytes_per_launch": 2.0 * fk * 1024.0 + wk * 1024.0, "dispatches_averaged": min(nf, nw) if nf and nw else max(nf, nw)}
    json.dump({"method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, kernel trace only); KB -> bytes x1024; "
                         "gfx950 correction: read bytes = 2 x FETCH_SIZE; per launch = mean over steady-state dispatches",
               "kernels": out}, open(sys.argv[3], "w"), indent=1)
    for k in list(out)[:14]:
        v = out[k]
        print(f"{v['hbm_bytes_per_launch'] / 1e6:9.1f} MB/launch (fetch raw {v['fetch_size_raw_bytes'] / 1e6:7.1f}, write raw {v['write_size_raw_bytes'] / 1e6:7.1f}) n={v['dispatches_averaged']:4d}  {k[:90]}")


if __name__ == "__main__":
    main()
