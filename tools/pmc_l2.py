#!/usr/bin/env python3
"""Fold a rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum pass (and optionally a TCC_EA0_RDREQ_sum one) into an L2 hit rate per
kernel family: separates "the fabric-side read counter is high because L2 misses are served by the Infinity Cache" from
"the kernel re-reads HBM".  usage: pmc_l2.py <counter_collection.csv>"""
import collections
import csv
import sys

FAMILIES = ("conv_igemm", "conv_wino", "bottleneck_link", "stem_pool", "flash_attention64", "attention_kernel", "layernorm", "mano_heads")
acc = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    fam = next((f for f in FAMILIES if f in r["Kernel_Name"]), "other")
    acc[fam][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[fam].add(r["Dispatch_Id"])
for fam, c in sorted(acc.items()):
    h, m = c.get("TCC_HIT_sum", 0.0), c.get("TCC_MISS_sum", 0.0)
    if h + m <= 0:
        continue
    n = len(disp[fam])
    print(f"{fam}: {n} dispatches, L2 hit rate {100 * h / (h + m):.1f} % (TCC_HIT_sum {h:.4g}, TCC_MISS_sum {m:.4g}); "
          f"misses x 128 B = {m * 128 / n / 1e9:.3f} GB per dispatch leave the L2 towards Infinity Cache / HBM")
