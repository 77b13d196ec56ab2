#!/usr/bin/env python3
"""Per-kernel MFMA / LDS counters from rocprofv3 --pmc passes (SQ block) of the bench step.

usage: pmc_mfma.py <counter_collection.csv> <out.json> [<second pass csv> ...]

Per kernel name (steady-state half of its dispatches): mean per dispatch of every collected counter, the dispatch duration
from the CSV's own timestamps (kernels are serialised under counter collection, so this is the kernel ALONE), and
  mfma_busy_frac  = SQ_VALU_MFMA_BUSY_CYCLES / (duration_ns * 2.4 GHz * 256 CU * 4 SIMD)      [cycles, MI355X_MICROARCH.md]
  mfma_flops_est  = SQ_INSTS_VALU_MFMA_MOPS_BF16 * 512                                         [MOPS unit = 512 FLOP]
  lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE                                 [extra cycles / all LDS cycles]
  lds_issue_stall_frac = SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES                                     [both in quad-cycles]
ROCm 7.2 ships no gfx950 derived-counter formulas (guide, "rocprofv3 PMC slots"), hence the explicit arithmetic."""
import collections
import csv
import json
import sys

CLK_GHZ, CUS, SIMDS = 2.4, 256, 4


def read(path, acc):
    rows = list(csv.DictReader(open(path)))
    if not rows:
        return
    cols = rows[0].keys()
    t0 = next((c for c in cols if c.lower().startswith("start_timestamp")), None)
    t1 = next((c for c in cols if c.lower().startswith("end_timestamp")), None)
    by_kernel = collections.defaultdict(lambda: collections.defaultdict(dict))
    for r in rows:
        d = by_kernel[r["Kernel_Name"]][int(r["Dispatch_Id"])]
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        if t0 and t1:
            d["_ns"] = float(r[t1]) - float(r[t0])
        d["_regs"] = (r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("LDS_Block_Size"), r.get("Scratch_Size"))
    for k, disp in by_kernel.items():
        ids = sorted(disp)
        ids = ids[len(ids) // 2:] or ids
        a = acc[k]
        names = set()
        for i in ids:
            names |= set(disp[i])
        for n in names:
            if n == "_regs":
                a[n] = disp[ids[-1]][n]
                continue
            vals = [disp[i][n] for i in ids if n in disp[i]]
            a[n] = sum(vals) / len(vals)
        a["_dispatches"] = len(ids)


def main():
    acc = collections.defaultdict(dict)
    read(sys.argv[1], acc)
    for p in sys.argv[3:]:
        read(p, acc)
    out = {}
    for k, a in acc.items():
        ns = a.get("_ns", 0.0)
        e = {"dispatches_averaged": a.get("_dispatches"), "alone_us": round(ns / 1e3, 2), "vgpr_agpr_lds_scratch": a.get("_regs"),
             "counters": {n: v for n, v in a.items() if not n.startswith("_")}}
        c = e["counters"]
        if ns and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            e["mfma_busy_frac"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (ns * CLK_GHZ * CUS * SIMDS), 4)
        if "SQ_INSTS_VALU_MFMA_MOPS_BF16" in c:
            e["mfma_flops_est"] = c["SQ_INSTS_VALU_MFMA_MOPS_BF16"] * 512.0
            if ns:
                e["mfma_tflops_alone"] = round(e["mfma_flops_est"] / ns / 1e3, 1)
        if c.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_conflict_frac"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 4)
        if c.get("SQ_WAVE_CYCLES"):
            e["lds_issue_stall_frac"] = round(c.get("SQ_WAIT_INST_LDS", 0.0) / c["SQ_WAVE_CYCLES"], 4)
        out[k] = e
    order = sorted(out, key=lambda k: -(out[k]["alone_us"] * (out[k]["dispatches_averaged"] or 0)))
    json.dump({"method": __doc__.split("usage")[0].strip() + " Formulas in tools/pmc_mfma.py.",
               "kernels": {k: out[k] for k in order}}, open(sys.argv[2], "w"), indent=1)
    for k in order[:24]:
        e = out[k]
        print(f"{e['alone_us']:9.1f} us x{e['dispatches_averaged']:3d}  mfma_busy {e.get('mfma_busy_frac', '-')}  "
              f"TF(alone) {e.get('mfma_tflops_alone', '-')}  lds_conf {e.get('lds_conflict_frac', '-')}  "
              f"lds_stall {e.get('lds_issue_stall_frac', '-')}  {k.replace('void ', '')[:70]}")


if __name__ == "__main__":
    main()
