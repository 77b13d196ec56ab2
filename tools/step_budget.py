#!/usr/bin/env python3
"""Where one G+D iteration's GPU time goes, from a layer table written by ``bench.py --layer-table``.

Three cuts of the same rows (per (layer, kernel) of one surveyed eager step, every launch bracketed by HIP events):
by kernel family, by resolution, and by regime -- launches whose duration is within 2x of the event-bracketed launch
floor (the shortest launches of the table) are latency-bound whatever their arithmetic; launches above 2.5 TB/s of
algorithmic bytes are bandwidth-bound; the rest is where kernel quality (MFMA efficiency, occupancy) decides.

    python tools/step_budget.py profiles/r02_f_step_bf16_b4_layer_table.tsv
"""
import collections
import csv
import re
import sys

FAMILIES = [("conv", r"conv[23]?_kernel"), ("wgrad", r"wgrad|prereduce|colsum|rgb_wgrad"), ("epilogue", r"gepi"),
            ("blur", r"blur"), ("rgb/fade/act", r"rgb_|axpby|lrelu|up2|pool2|bias_act|fade_"), ("optimizer", r"adam|ema|sumsq|scale_dev|clip"),
            ("pack", r"pack_weight|pack_upblur|rgbconv_pack"), ("linear/mapping", r"gemm|style_|pixelnorm|linear|mbstd")]


def family(kernel):
    for name, pat in FAMILIES:
        if re.search(pat, kernel):
            return name
    return "other"


BATCH = [1]                                                    # images per step, from the first "B<n>" of the table


def resolution(layer):
    m = re.search(r"(\d+)x(\d+)", layer)
    if m and int(m.group(1)) > 4096:                           # "rgb_in <pixels>x<channels>": pixels = B*H*W
        return int(round((int(m.group(1)) / BATCH[0]) ** 0.5))
    if m:
        return int(m.group(1))
    m = re.search(r"HW(\d+)", layer)
    if m:
        return int(round(int(m.group(1)) ** 0.5))
    return 0


def main(path):
    rows = list(csv.DictReader(open(path), delimiter="\t"))
    for r in rows:
        m = re.search(r" B(\d+) ", r["layer"])
        if m:
            BATCH[0] = int(m.group(1)); break
    for r in rows:
        r["n"] = float(r["calls_per_step"]); r["us"] = float(r["avg_us"]); r["ms"] = float(r["ms_per_step"])
        r["gbs"] = float(r["GB/s(algorithmic)"]); r["tf"] = float(r["TFLOP/s"])
    total = sum(r["ms"] for r in rows); launches = sum(r["n"] for r in rows)
    floor = sorted(r["us"] for r in rows)[max(0, len(rows) // 20)]          # 5th percentile of the average durations
    print(f"{path}\n{total:.2f} ms of kernel time in {launches:.0f} launches per step; launch floor (5th percentile) {floor:.1f} us\n")

    def table(title, key, order=None):
        ms = collections.Counter(); n = collections.Counter()
        for r in rows:
            ms[key(r)] += r["ms"]; n[key(r)] += r["n"]
        print(f"{title:28s} {'ms/step':>8s} {'share':>7s} {'launches':>9s} {'avg us':>7s}")
        for k in (order or [k for k, _ in ms.most_common()]):
            if k in ms:
                print(f"{str(k):28s} {ms[k]:8.3f} {ms[k] / total:7.1%} {n[k]:9.0f} {ms[k] * 1e3 / n[k]:7.1f}")
        print()

    table("by kernel family", lambda r: family(r["kernel"]))
    table("by resolution (0 = none)", lambda r: resolution(r["layer"]), sorted({resolution(r["layer"]) for r in rows}))

    def regime(r):
        if r["us"] <= 2.0 * floor:
            return "latency-bound (<= 2x floor)"
        if r["gbs"] >= 2500:
            return "bandwidth-bound (>= 2.5 TB/s)"
        if r["tf"] >= 400:
            return "MFMA >= 400 TFLOP/s"
        return "in between"
    table("by regime", regime)
    lat = [r for r in rows if regime(r).startswith("latency")]
    print(f"latency-bound launches: {sum(r['n'] for r in lat):.0f} per step; at the floor they would take "
          f"{sum(r['n'] for r in lat) * floor / 1e3:.2f} ms, they take {sum(r['ms'] for r in lat):.2f} ms")
    bw = [r for r in rows if regime(r).startswith("bandwidth")]
    nbytes = sum(r["gbs"] * r["us"] * r["n"] for r in bw) * 1e3             # GB/s * us = kB
    print(f"bandwidth-bound launches move {nbytes / 1e9:.2f} GB (algorithmic) per step in {sum(r['ms'] for r in bw):.2f} ms "
          f"= {nbytes / 1e9 / max(sum(r['ms'] for r in bw), 1e-9):.2f} TB/s")


if __name__ == "__main__":
    main(sys.argv[1])
