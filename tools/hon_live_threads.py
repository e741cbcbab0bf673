#!/usr/bin/env python3
"""HandOccNet shipped default against the fp32 oracle run LIVE on this machine at several ATen thread counts (GPU box: its host CPU
and its 16-core quota differ from the dev container's, where the stored references of tools/hon_parity_ab.py are made -- ATen's
blocked sums depend on both).  Per seed (bz = 2): max vertex error of the HIP forward against each live reference, and how far the
references are from each other.
usage: python tools/hon_live_threads.py [--seeds N] [--first S] [--threads 1,8,16] [--out gpurun_out/hon_live_threads.json]
       [--procs 8,1,1]   (worker processes per thread count, same order as --threads: the references are then computed by spawned
                          pools created before this process touches the GPU -- ATen splits a sum by the thread COUNT, not by which
                          cores are free, so the bits are those of the serial run; default: in this process, one after the other)"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HANDS_SYNTHETIC_MANO", "1")

_W = {}


def _init(threads):
    import torch
    import hands_amd
    torch.set_num_threads(threads)
    m = hands_amd.apply_recipe(hands_amd.HandOccNet())
    _W["sd"] = {k: v.clone() for k, v in m.state_dict().items()}
    _W["ar"], _W["al"] = hands_amd.synthetic_mano_asset(True), hands_amd.synthetic_mano_asset(False)


def _ref(seed):
    import torch
    from hands_amd.weights import synthetic_inputs
    from oracle import handoccnet_oracle as HO
    ci, cm = synthetic_inputs(2, seed)
    o = HO.handoccnet_forward(_W["sd"], _W["ar"], _W["al"], ci, cm)
    return seed, torch.stack([o[f"mano.vertices.{h}"] for h in "rl"]).double().numpy()


def main():
    import multiprocessing as mp
    import numpy as np
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=200)
    ap.add_argument("--first", type=int, default=3000)
    ap.add_argument("--threads", default="1,8,16")
    ap.add_argument("--procs", default="")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "hon_live_threads.json"))
    a = ap.parse_args()
    counts = [int(t) for t in a.threads.split(",")]
    seeds = list(range(a.first, a.first + a.seeds))
    procs = [int(p) for p in a.procs.split(",")] if a.procs else []
    pending = {}
    if procs:                                   # pools first: nothing in this process has touched the GPU yet
        assert len(procs) == len(counts)
        ctx = mp.get_context("spawn")
        pools = {t: ctx.Pool(p, initializer=_init, initargs=(t,)) for t, p in zip(counts, procs)}
        pending = {t: pools[t].imap_unordered(_ref, seeds, chunksize=4) for t in counts}

    import torch
    import hands_amd
    from hands_amd.weights import synthetic_inputs
    from oracle import handoccnet_oracle as HO
    model = hands_amd.apply_recipe(hands_amd.HandOccNet())
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    ar, al = hands_amd.synthetic_mano_asset(True), hands_amd.synthetic_mano_asset(False)
    model = model.to("cuda").eval()
    model.async_forward = False
    vs = lambda o: torch.stack([o[f"mano.vertices.{h}"] for h in "rl"]).double().cpu()
    res = {"threads": counts, "procs": procs, "cpu_count": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)), "seeds": seeds,
           "hip_vs": {str(t): [] for t in counts}, "ref_vs_first": {str(t): [] for t in counts[1:]}}
    try:
        res["cpu_max"] = open("/sys/fs/cgroup/cpu.max").read().strip()
    except OSError:
        pass
    outs, refs = {}, {t: {} for t in counts}
    for i, seed in enumerate(seeds):
        ci, cm = synthetic_inputs(2, seed)
        outs[seed] = vs(model({k: v.to("cuda") for k, v in ci.items()}, {k: v.to("cuda") for k, v in cm.items()})).numpy()
        if not procs:
            for t in counts:
                torch.set_num_threads(t)
                refs[t][seed] = vs(HO.handoccnet_forward(sd, ar, al, ci, cm)).numpy()
            if (i + 1) % 50 == 0:
                print(i + 1, flush=True)
    for t, it in pending.items():
        for n, (seed, v) in enumerate(it):
            refs[t][seed] = v
            if (n + 1) % 100 == 0:
                print(f"{t} threads: {n + 1}", flush=True)
    for seed in seeds:
        for t in counts:
            res["hip_vs"][str(t)].append(float(np.abs(outs[seed] - refs[t][seed]).max()))
        for t in counts[1:]:
            res["ref_vs_first"][str(t)].append(float(np.abs(refs[t][seed] - refs[counts[0]][seed]).max()))
    summ = {"n": len(seeds), "first_seed": a.first, "cpu_count": res["cpu_count"], "affinity": res["affinity"], "cpu_max": res.get("cpu_max"),
            "procs": procs}
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


if __name__ == "__main__":
    main()
