#!/usr/bin/env python3
"""Fold a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass into an MFMA-pipe utilisation figure per
kernel family.  usage: pmc_mfma.py <counter_collection.csv>
MFMA busy fraction = (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs): the SQ counter is summed
over all SIMDs of the chip, GRBM_GUI_ACTIVE over the 8 XCDs.  Families: every kernel of libhands_hip.so that issues
MFMAs (conv_igemm*, stem_pool*, flash_attention64, attention_kernel, mano_heads); the rest is folded into "other"."""
import collections
import csv
import sys

FAMILIES = ("conv_igemm", "conv_wino", "bottleneck_link", "stem_pool", "flash_attention64", "attention_kernel", "cross_attention", "mano_heads")
acc = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    fam = next((f for f in FAMILIES if f in n), "other")
    acc[fam][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[fam].add(r["Dispatch_Id"])
for fam, c in sorted(acc.items()):
    if c["GRBM_GUI_ACTIVE"] <= 0:
        continue
    busy = (c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0) / (c["GRBM_GUI_ACTIVE"] / 8.0)
    print(f"{fam}: {len(disp[fam])} dispatches, MFMA pipe busy {100 * busy:.1f} % of GPU-active cycles "
          f"(SQ_VALU_MFMA_BUSY_CYCLES {c['SQ_VALU_MFMA_BUSY_CYCLES']:.4g}, GRBM_GUI_ACTIVE {c['GRBM_GUI_ACTIVE']:.4g})")
