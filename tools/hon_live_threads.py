#!/usr/bin/env python3
"""HandOccNet shipped default against the fp32 oracle run LIVE on this machine at several ATen thread counts (GPU box: its host CPU
and its 16-core quota differ from the dev container's, where the stored references of tools/hon_parity_ab.py are made -- ATen's
blocked sums depend on both).  Per seed (bz = 2): max vertex error of the HIP forward against each live reference, and how far the
references are from each other.
usage: python tools/hon_live_threads.py [--seeds N] [--first S] [--threads 1,8,16] [--out gpurun_out/hon_live_threads.json]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HANDS_SYNTHETIC_MANO", "1")
import numpy as np
import torch

import hands_amd
from hands_amd.weights import synthetic_inputs
from oracle import handoccnet_oracle as HO

ap = argparse.ArgumentParser()
ap.add_argument("--seeds", type=int, default=200)
ap.add_argument("--first", type=int, default=3000)
ap.add_argument("--threads", default="1,8,16")
ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "hon_live_threads.json"))
a = ap.parse_args()
counts = [int(t) for t in a.threads.split(",")]
model = hands_amd.apply_recipe(hands_amd.HandOccNet())
sd = {k: v.clone() for k, v in model.state_dict().items()}
ar, al = hands_amd.synthetic_mano_asset(True), hands_amd.synthetic_mano_asset(False)
model = model.to("cuda").eval()
model.async_forward = False
vs = lambda o: torch.stack([o[f"mano.vertices.{h}"] for h in "rl"]).double().cpu()
res = {"threads": counts, "cpu_count": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)), "seeds": [],
       "hip_vs": {str(t): [] for t in counts}, "ref_vs_first": {str(t): [] for t in counts[1:]}}
for i, seed in enumerate(range(a.first, a.first + a.seeds)):
    ci, cm = synthetic_inputs(2, seed)
    out = vs(model({k: v.to("cuda") for k, v in ci.items()}, {k: v.to("cuda") for k, v in cm.items()}))
    refs = {}
    for t in counts:
        torch.set_num_threads(t)
        refs[t] = vs(HO.handoccnet_forward(sd, ar, al, ci, cm))
        res["hip_vs"][str(t)].append((out - refs[t]).abs().max().item())
    for t in counts[1:]:
        res["ref_vs_first"][str(t)].append((refs[t] - refs[counts[0]]).abs().max().item())
    res["seeds"].append(seed)
    if (i + 1) % 50 == 0:
        print(i + 1, flush=True)
summ = {"n": len(res["seeds"]), "cpu_count": res["cpu_count"], "affinity": res["affinity"]}
for t in counts:
    e = np.array(res["hip_vs"][str(t)])
    summ[f"hip_vs_t{t}"] = {"exceed": int((e > 1e-6).sum()), "median": float(np.median(e)), "p99": float(np.percentile(e, 99)), "max": float(e.max())}
    print(f"HIP vs the live {t}-thread reference: > 1e-6 m {int((e > 1e-6).sum())}/{len(e)}, median {np.median(e):.3e}, p99 {np.percentile(e, 99):.3e}, max {e.max():.3e}")
for t in counts[1:]:
    r = np.array(res["ref_vs_first"][str(t)])
    summ[f"ref_t{t}_vs_t{counts[0]}"] = {"median": float(np.median(r)), "max": float(r.max())}
    print(f"reference at {t} threads vs at {counts[0]}: median {np.median(r):.3e}, max {r.max():.3e}")
res["summary"] = summ
json.dump(res, open(a.out, "w"))
json.dump(summ, open(a.out.replace(".json", "_summary.json"), "w"), indent=1)
