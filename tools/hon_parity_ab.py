#!/usr/bin/env python3
"""HandOccNet parity A/B over many input seeds (dev tool, GPU box).

Per seed (bz = 2 -> 4 hands) and per arm: max vertex error of the HIP forward against the fp32 oracle (= the reference's
arithmetic, 8 ATen threads as everywhere in tests/) AND against an fp64 evaluation of the same network; plus the fp32 oracle's
own error against fp64 (the floor an exact evaluator would sit at from the reference).  Arms = winograd_scope x engine.chain_limit
(blocked fp32 summation through the deterministic split-K form), one model instance per arm.  The CPU forwards run in worker
processes (8 threads each) created before the parent touches the GPU.

usage: python tools/hon_parity_ab.py [--seeds N] [--first S] [--arms a,b,...] [--workers W] [--out gpurun_out/hon_ab.json]
       [--speed]   (hands/s of every arm at bz = 32 and 256, shipped pipelined mode)
arm syntax: <scope>[+c...][+f:stage.stage...][+b:stages][+w:stages] in THIS order (f: stages accumulated in fp64: resnet fpn fit set hourglass reghead encoder mlp, or ALL)
            <scope>[+c<chain_limit>[k<chain_min_k>][p<chain_max_pix>][t = token GEMMs unblocked][i = blocks summed inside the launch]], scope in direct | trunk | backbone | backbone+fit | all
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HANDS_SYNTHETIC_MANO", "1")

_W = {}
_NO_F64_3X3 = False
_W4_STAGES = frozenset()


def _worker_init():
    import torch
    import hands_amd
    torch.set_num_threads(8)
    m = hands_amd.apply_recipe(hands_amd.HandOccNet())
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    _W["sd"] = sd
    _W["sd64"] = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    _W["ar"], _W["al"] = hands_amd.synthetic_mano_asset(True), hands_amd.synthetic_mano_asset(False)


def _worker(seed):
    import torch
    from hands_amd.weights import synthetic_inputs
    from oracle import handoccnet_oracle as HO
    c64 = lambda d: {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in d.items()}
    ci, cm = synthetic_inputs(2, seed)
    r32 = HO.handoccnet_forward(_W["sd"], _W["ar"], _W["al"], ci, cm)
    r64 = HO.handoccnet_forward(_W["sd64"], _W["ar"], _W["al"], c64(ci), c64(cm))
    v32 = torch.stack([r32[f"mano.vertices.{h}"] for h in "rl"]).numpy()
    v64 = torch.stack([r64[f"mano.vertices.{h}"] for h in "rl"]).numpy()
    return seed, v32, v64


def wilson(k, n, z=1.96):
    if n == 0:
        return (0.0, 1.0)
    p = k / n
    d = 1 + z * z / n
    c = (p + z * z / (2 * n)) / d
    h = z * math.sqrt(p * (1 - p) / n + z * z / (4 * n * n)) / d
    return (max(0.0, c - h), min(1.0, c + h))


def parse_arm(name):
    global _NO_F64_3X3
    _NO_F64_3X3 = name.endswith("+nf3")             # "+nf3": the 3x3 layers of the fp64 stages stay fp32
    name = name[:-4] if _NO_F64_3X3 else name
    global _W4_STAGES
    name, _, w4 = name.partition("+w4:")           # "+w4:resnet": F(4x4,3x3) for the Winograd layers of these stages
    _W4_STAGES = frozenset(x for x in w4.split(".") if x)
    name, _, wst = name.partition("+w:")           # "+w:resnet.fpn": Winograd F(2x2) in these stages only (overrides the scope)
    name, _, blk = name.partition("+b:")           # "+b:fpn.fit.set": blocked summation in these stages only
    name, _, f64 = name.partition("+f:")           # "+f:reghead.encoder.mlp": stages accumulated in fp64
    scope, _, rest = name.partition("+c")
    limit = min_k = max_pix = 0
    skip_tok, in_kernel = "t" in rest, "i" in rest
    rest = rest.rstrip("ti")
    if rest:
        rest, _, mp = rest.partition("p")
        lim, _, mk = rest.partition("k")
        limit, min_k, max_pix = int(lim), int(mk or 0), int(mp or 0)
    fs = lambda t: frozenset(x for x in t.split(".") if x)
    return scope, limit, min_k, max_pix, skip_tok, in_kernel, fs(f64), (fs(blk) if blk else None), (fs(wst) if wst else None)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=1000)
    ap.add_argument("--first", type=int, default=1000)
    ap.add_argument("--arms", default="backbone,all,all+c256k512,all+c128i,all+c64i")
    ap.add_argument("--workers", type=int, default=2)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "hon_ab.json"))
    ap.add_argument("--speed", action="store_true")
    ap.add_argument("--refs", default=None, help="npz from tools/hon_refs_precompute.py (first, v32, v64): no CPU forwards "
                    "here except --check of them, recomputed live and compared")
    ap.add_argument("--check", type=int, default=4)
    ap.add_argument("--refs-threads", default="", help="more REFERENCE arms: t1=path,t16=path (npz from tools/hon_refs_threads.py: the "
                    "fp32 oracle at another ATen thread count, same seeds); every HIP arm is compared with each")
    a = ap.parse_args()
    arms = a.arms.split(",")
    import multiprocessing as mp
    pool = mp.get_context("spawn").Pool(a.workers, initializer=_worker_init)
    seeds = list(range(a.first, a.first + a.seeds))
    if a.refs:
        import numpy as np
        z = np.load(a.refs)
        first, n_have = int(z["first"]), len(z["v32"])
        seeds = [s for s in seeds if first <= s < first + n_have]
        live = pool.map(_worker, seeds[:a.check])          # the stored forwards were made on another machine: check a few
        for s, v32, v64 in live:
            d32 = float(np.abs(z["v32"][s - first] - v32).max())
            d64 = float(np.abs(z["v64"][s - first] - v64).max())
            print(f"refs check seed {s}: stored vs live fp32 {d32:.3e}, fp64 {d64:.3e}", flush=True)
        it = ((s, z["v32"][s - first], z["v64"][s - first]) for s in seeds)
    else:
        it = pool.imap(_worker, seeds)            # starts computing while the models are built

    import numpy as np
    import torch
    import hands_amd
    from hands_amd.weights import synthetic_inputs
    extra = {}                                     # reference arm name -> (first seed, v32 array)
    for item in filter(None, a.refs_threads.split(",")):
        nm, _, path = item.partition("=")
        zz = np.load(path)
        extra[nm] = (int(zz["first"]), zz["v32"])
    models = {}
    for name in arms:
        scope, limit, min_k, max_pix, skip_tok, in_kernel, f64, blk, wst = parse_arm(name)
        m = hands_amd.apply_recipe(hands_amd.HandOccNet()).to("cuda").eval()
        m.acc64_stages = f64 if "ALL" not in f64 else frozenset(hands_amd.handoccnet.STAGES)
        m.acc64_3x3 = not _NO_F64_3X3
        m.wino4_stages, m.engine.winograd4 = _W4_STAGES, bool(_W4_STAGES)
        if blk is not None:
            m.block_stages = blk
        if wst is not None:
            m.wino_stages = wst
        m.engine.winograd = scope != "direct"
        m.winograd_scope = scope if scope != "direct" else "backbone"
        m.engine.chain_limit, m.engine.chain_min_k, m.engine.chain_max_pix = limit, min_k, max_pix
        m.engine.chain_skip_tokens, m.engine.chain_in_kernel = skip_tok, in_kernel
        m.invalidate_packed()
        m.async_forward = False
        models[name] = m
    res = {"arms": arms, "seeds": [], "ref32_vs_64": [], "vs32": {n: [] for n in arms}, "vs64": {n: [] for n in arms},
           "vs32_other": {nm: {n: [] for n in arms} for nm in extra}, "ref_other_vs_64": {nm: [] for nm in extra},
           "ref_other_vs_ref": {nm: [] for nm in extra}}
    t0 = time.time()
    for i, (seed, v32, v64) in enumerate(it):
        ci, cm = synthetic_inputs(2, seed)
        gi, gm = {k: v.to("cuda") for k, v in ci.items()}, {k: v.to("cuda") for k, v in cm.items()}
        res["seeds"].append(seed)
        res["ref32_vs_64"].append(float(np.abs(v32.astype(np.float64) - v64).max()))
        others = {nm: arr[seed - f0] for nm, (f0, arr) in extra.items() if 0 <= seed - f0 < len(arr)}
        for nm, vo in others.items():
            res["ref_other_vs_64"][nm].append(float(np.abs(vo.astype(np.float64) - v64).max()))
            res["ref_other_vs_ref"][nm].append(float(np.abs(vo - v32).max()))
        for name, m in models.items():
            out = m(gi, gm)
            torch.cuda.synchronize()
            v = torch.stack([out[f"mano.vertices.{h}"] for h in "rl"]).cpu().numpy()
            res["vs32"][name].append(float(np.abs(v - v32).max()))
            res["vs64"][name].append(float(np.abs(v.astype(np.float64) - v64).max()))
            for nm, vo in others.items():
                res["vs32_other"][nm][name].append(float(np.abs(v - vo).max()))
        if (i + 1) % 100 == 0 or i + 1 == len(seeds):
            json.dump(res, open(a.out, "w"))
            print(f"[{time.time() - t0:6.0f} s] {i + 1} seeds", flush=True)
    pool.close()
    n = len(res["seeds"])
    r = np.array(res["ref32_vs_64"])
    summary = {"n": n, "ref32_vs_64": {"median": float(np.median(r)), "p99": float(np.percentile(r, 99)), "max": float(r.max()),
                                      "above_1e-6": int((r > 1e-6).sum())}, "arms": {}}
    print(f"oracle fp32 vs fp64 (the floor): median {np.median(r):.3e} p99 {np.percentile(r, 99):.3e} max {r.max():.3e}; "
          f"> 1e-6: {(r > 1e-6).sum()} of {n}")
    for name in arms:
        e, e64 = np.array(res["vs32"][name]), np.array(res["vs64"][name])
        k = int((e > 1e-6).sum())
        lo, hi = wilson(k, n)
        ratio = e64 / np.maximum(r, 1e-12)
        summary["arms"][name] = {"exceed": k, "exceed_rate": k / n, "wilson95": [lo, hi], "median": float(np.median(e)),
                                 "p90": float(np.percentile(e, 90)), "p99": float(np.percentile(e, 99)), "max": float(e.max()),
                                 "above_1.2e-6": int((e > 1.2e-6).sum()),
                                 "vs64_median": float(np.median(e64)), "vs64_p99": float(np.percentile(e64, 99)), "vs64_max": float(e64.max()),
                                 "median_ratio_hip64_over_ref64": float(np.median(ratio))}
        print(f"{name:18s} vs ref32: > 1e-6 {k:3d}/{n} = {100 * k / n:.2f} % [{100 * lo:.2f}, {100 * hi:.2f}]  median {np.median(e):.3e} "
              f"p90 {np.percentile(e, 90):.3e} p99 {np.percentile(e, 99):.3e} max {e.max():.3e} (> 1.2e-6: {(e > 1.2e-6).sum()}) | "
              f"vs fp64: median {np.median(e64):.3e} p99 {np.percentile(e64, 99):.3e} max {e64.max():.3e}, "
              f"median err(HIP,64) / err(ref32,64) = {np.median(ratio):.2f}")
    for nm in extra:          # the reference at other thread counts: its own distance from fp64 / from the 8-thread run, and every arm against it
        ro, rr = np.array(res["ref_other_vs_64"][nm]), np.array(res["ref_other_vs_ref"][nm])
        summary.setdefault("refs", {})[nm] = {"n": len(ro), "vs64_median": float(np.median(ro)), "vs64_max": float(ro.max()),
                                              "vs_ref8_median": float(np.median(rr)), "vs_ref8_max": float(rr.max())}
        print(f"reference {nm}: {len(ro)} seeds, vs fp64 median {np.median(ro):.3e} max {ro.max():.3e}; vs the 8-thread reference median "
              f"{np.median(rr):.3e} max {rr.max():.3e}")
        for name in arms:
            e = np.array(res["vs32_other"][nm][name])
            k = int((e > 1e-6).sum())
            lo, hi = wilson(k, len(e))
            summary["arms"][name][f"vs_{nm}"] = {"n": len(e), "exceed": k, "wilson95": [lo, hi], "median": float(np.median(e)),
                                                  "p99": float(np.percentile(e, 99)), "max": float(e.max())}
            print(f"  {name:18s} vs {nm}: > 1e-6 {k:3d}/{len(e)} [{100 * lo:.2f}, {100 * hi:.2f}] %  median {np.median(e):.3e} p99 "
                  f"{np.percentile(e, 99):.3e} max {e.max():.3e}")
    if a.speed:
        for name, m in models.items():
            m.async_forward = True
            for bz in (32, 256):
                gi, gm = synthetic_inputs(bz, 0, device="cuda")
                for _ in range(3):
                    out = m(gi, gm)
                torch.cuda.synchronize()
                reps = 12 if bz == 32 else 5
                best = 1e9
                for _ in range(3):
                    t = time.perf_counter()
                    for _ in range(reps):
                        out = m(gi, gm)
                    torch.cuda.synchronize()
                    best = min(best, (time.perf_counter() - t) / reps)
                summary["arms"][name][f"hands_per_s_bz{bz}"] = 2 * bz / best
                print(f"{name:18s} bz {bz:3d}: {2 * bz / best:8.1f} hands/s", flush=True)
                del out
            torch.cuda.empty_cache()
    res["summary"] = summary
    json.dump(res, open(a.out, "w"))
    json.dump(summary, open(a.out.replace(".json", "_summary.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
