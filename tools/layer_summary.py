#!/usr/bin/env python3
"""Summarise a bench.py --layer-table TSV: per-kernel totals, then the top per-layer rows.   usage: layer_summary.py FILE [N] [filter]"""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1]), delimiter='\t'))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
flt = sys.argv[3] if len(sys.argv) > 3 else ""
short = lambda k: k.split('(')[0].replace('unsigned short', 'bf16').replace('void ', '')
agg = collections.OrderedDict()
for r in rows:
    a = agg.setdefault(short(r['kernel']), [0, 0.0])
    a[0] += int(float(r['calls_per_step'])); a[1] += float(r['ms_per_step'])
print('total ms', round(sum(v[1] for v in agg.values()), 3), 'launches', sum(v[0] for v in agg.values()))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:n]:
    print(f"{k:58s} n={v[0]:4d} ms={v[1]:.3f} avg_us={v[1]*1e3/v[0]:.1f}")
print()
for r in [r for r in rows if flt in r['kernel'] or flt in r['layer']][:n]:
    print(f"{r['layer']:30s} {short(r['kernel'])[:44]:44s} n={r['calls_per_step']:>3s} us={r['avg_us']:>6s} TF={r['TFLOP/s']:>6s} GB/s={r['GB/s(algorithmic)']:>5s} ms={r['ms_per_step']}")
