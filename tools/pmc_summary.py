#!/usr/bin/env python3
"""Fold rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/r01_pmc_<workload>.json (dev tool).
usage: pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <workload> <out.json> <bz>
HBM bytes = 2 * FETCH_SIZE (KB; gfx950 reports half of a wide coalesced read stream, MI355X_MICROARCH.md
section HBM) + WRITE_SIZE (KB), summed over every conv_igemm* / conv_wino dispatch of the run."""
import csv
import json
import sys

fetch_csv, write_csv, workload, out = sys.argv[1:5]
bz = int(sys.argv[5])          # samples per step the passes were taken at: bench.py refuses the summary at any other


def total(fn, counter):
    tot, disp = 0.0, set()
    for r in csv.DictReader(open(fn)):
        if ("conv_igemm" in r["Kernel_Name"] or "conv_wino" in r["Kernel_Name"]) and r["Counter_Name"] == counter:
            tot += float(r["Counter_Value"])
            disp.add(r["Dispatch_Id"])
    return tot, len(disp)


def by_grid(fn, counter):
    """per (template instantiation, grid size): [dispatches, KB] -- which launch shapes carry the traffic"""
    g = {}
    for r in csv.DictReader(open(fn)):
        if ("conv_igemm" in r["Kernel_Name"] or "conv_wino" in r["Kernel_Name"]) and r["Counter_Name"] == counter:
            inst = r["Kernel_Name"].split("<")[1].split(">")[0].replace(" ", "") if "<" in r["Kernel_Name"] else "?"
            key = ("sk" if "_sk_" in r["Kernel_Name"] else ("wino" if "conv_wino" in r["Kernel_Name"] else "")) + f"<{inst}> grid={r.get('Grid_Size', '?')}"
            d = g.setdefault(key, [set(), 0.0])
            d[0].add(r["Dispatch_Id"])
            d[1] += float(r["Counter_Value"])
    return {k: [len(v[0]), v[1]] for k, v in g.items()}


f, nf = total(fetch_csv, "FETCH_SIZE")
w, nw = total(write_csv, "WRITE_SIZE")
assert nf == nw and nf > 0, (nf, nw)
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_sha16      # hash of the kernel sources: bench.py uses a stored summary only for the binary it was taken with

res = {"workload": workload, "bz": bz, "csrc_sha16": csrc_sha16(), "kernel": "conv_igemm_f32_kernel + conv_igemm_group_f32_kernel + conv_igemm_sk_f32_kernel + conv_wino_f32_kernel + conv_wino4_f32_kernel (every dispatch whose name contains conv_igemm or conv_wino)", "dispatches": nf,
       "fetch_size_kb_sum": f, "write_size_kb_sum": w, "fetch_correction": 2.0,
       "hbm_gb_per_launch": (2.0 * f + w) * 1024 / nf / 1e9,
       "read_gb_per_launch": 2.0 * f * 1024 / nf / 1e9, "write_gb_per_launch": w * 1024 / nf / 1e9,
       "mode": sys.argv[6] if len(sys.argv) > 6 else "serial",
       "command": f"rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 bench.py --workload {workload} "
                  f"--bz {bz} {'--serial ' if len(sys.argv) <= 6 or sys.argv[6] == 'serial' else ''}--steps 1 --warmup 1 --no-cpu-baseline --no-also"
                  + (" (HANDS_BENCH_SHIPPED_ONLY=1: only shipped-mode forwards in the process)" if len(sys.argv) > 6 and sys.argv[6] != "serial" else "")}
gf, gw = by_grid(fetch_csv, "FETCH_SIZE"), by_grid(write_csv, "WRITE_SIZE")
res["by_launch_shape_gb_per_launch"] = {
    k: {"dispatches": gf[k][0], "read": round(2.0 * gf[k][1] * 1024 / gf[k][0] / 1e9, 4),
        "write": round(gw.get(k, [1, 0.0])[1] * 1024 / gf[k][0] / 1e9, 4)} for k in sorted(gf)}


def by_position(fn, counter):
    """dispatches of the conv kernels in launch order -> [(kernel family, grid, KB)]"""
    rows = {}
    for r in csv.DictReader(open(fn)):
        if ("conv_igemm" in r["Kernel_Name"] or "conv_wino" in r["Kernel_Name"]) and r["Counter_Name"] == counter:
            d = rows.setdefault(int(r["Dispatch_Id"]), [r["Kernel_Name"].split("(")[0].split("::")[-1][:40], r.get("Grid_Size", "?"), 0.0])
            d[2] += float(r["Counter_Value"])
    return [rows[k] for k in sorted(rows)]


# per launch POSITION within a forward (the serial pass runs whole forwards: the launch sequence repeats with the period found
# below), averaged over the forwards: joins with bench.py --layer-report's rows (same order, stem rows excluded)
pf, pw = by_position(fetch_csv, "FETCH_SIZE"), by_position(write_csv, "WRITE_SIZE")
sig = [(a, b) for a, b, _ in pf]
period = next((p_ for p_ in range(16, len(sig) // 2 + 1) if len(sig) % p_ == 0 and all(sig[i] == sig[i % p_] for i in range(len(sig)))), None)
if period and len(pw) == len(pf):
    reps = len(pf) // period
    res["launch_period"] = period
    res["forwards"] = reps
    res["gb_per_forward"] = res["hbm_gb_per_launch"] * period
    res["by_position_gb"] = [[pf[i][0], pf[i][1], round(2.0 * sum(pf[i + r * period][2] for r in range(reps)) * 1024 / reps / 1e9, 4),
                              round(sum(pw[i + r * period][2] for r in range(reps)) * 1024 / reps / 1e9, 4)] for i in range(period)]
json.dump(res, open(out, "w"), indent=1)
print({k: v for k, v in res.items() if k != 'by_launch_shape_gb_per_launch'})
