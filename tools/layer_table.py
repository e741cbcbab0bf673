#!/usr/bin/env python3
"""Summarise a bench.py --layer-report CSV by GEMM shape (dev tool)."""
import csv
import sys
from collections import OrderedDict

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
agg = OrderedDict()
for r in rows:
    key = (r["Cin"], r["Cout"], r["k"], r["stride"], r["M"])
    a = agg.setdefault(key, [0, 0.0, 0.0])
    a[0] += 1
    a[1] += float(r["gflop"])
    a[2] += float(r["ms"])
tot = sum(a[2] for a in agg.values())
totf = sum(a[1] for a in agg.values())
print(f"{'Cin':>5} {'Cout':>5} k s {'M':>8} {'n':>2} {'gflop':>7} {'ms':>7} {'TF/s':>6} {'%t':>5}")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][2])[:top]:
    print(f"{k[0]:>5} {k[1]:>5} {k[2]} {k[3]} {k[4]:>8} {a[0]:>2} {a[1]:7.1f} {a[2]:7.3f} {a[1] / a[2]:6.1f} {100 * a[2] / tot:5.1f}")
print(f"total {tot:.2f} ms, {totf:.0f} GFLOP, {totf / tot:.1f} TF/s")
