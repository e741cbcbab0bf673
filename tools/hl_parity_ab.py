#!/usr/bin/env python3
"""hands_light (the headline configuration; --workload hamer_light: config 3 at bz = 1, one arm): many-seed parity of the HIP forward
per 3x3 route, with statistics.

Per seed (bz = 2 -> 4 hands) and per arm: max vertex error against the fp32 oracle (= the reference's arithmetic) AND against an
fp64 evaluation of the same network, so "the HIP path is as accurate as the reference" is a number: the median of
err(HIP, fp64) / err(oracle fp32, fp64).  Exceedance of 1e-6 m is reported with a Wilson 95 % interval.  The HandOccNet twin of
this tool is hon_parity_ab.py.

usage: python tools/hl_parity_ab.py --precompute [--first S] [--seeds N]      (dev container, CPU: build_ab/hl_refs_<S>_<N>.npz)
       python tools/hl_parity_ab.py --refs build_ab/hl_refs_<S>_<N>.npz [--arms direct,f2x2,f4x4] [--check 4] [--out ...]   (GPU box)
arms: direct = the direct 3x3 kernel, f2x2 = Winograd F(2x2,3x3) in every 3x3 / stride-1 layer, f4x4 = the shipped default
(F(4x4,3x3) in all four ResNet stages); "+c<limit>[k<min_k>][p<max_pix>]" adds blocked summation inside the direct launches
(K >= min_k, maps of at most max_pix output pixels)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("HANDS_SYNTHETIC_MANO", "1")

_W = {}


def _init(workload="hands_light"):
    import torch
    import hands_amd
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    _W["workload"] = workload
    m = hands_amd.apply_recipe(hands_amd.HandsLight() if workload == "hands_light" else hands_amd.HAMER())
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    _W["sd"] = sd
    _W["sd64"] = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    _W["ar"], _W["al"] = hands_amd.synthetic_mano_asset(True), hands_amd.synthetic_mano_asset(False)


def _refs(seed):
    import torch
    from hands_amd.weights import synthetic_inputs
    if _W["workload"] == "hands_light":
        from oracle import hands_oracle as O
        fwd, bz = O.hands_light_forward, 2
    else:
        from oracle import hamer_oracle as HM
        fwd, bz = HM.hamer_forward, 1
    c64 = lambda d: {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in d.items()}
    ci, cm = synthetic_inputs(bz, seed)
    r32 = fwd(_W["sd"], _W["ar"], _W["al"], ci, cm)
    r64 = fwd(_W["sd64"], _W["ar"], _W["al"], c64(ci), c64(cm))
    v32 = torch.stack([r32[f"mano.vertices.{h}"] for h in "rl"]).numpy()
    v64 = torch.stack([r64[f"mano.vertices.{h}"] for h in "rl"]).numpy()
    return v32, v64


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precompute", action="store_true")
    ap.add_argument("--workload", default="hands_light", choices=("hands_light", "hamer_light"))
    ap.add_argument("--seeds", type=int, default=1000)
    ap.add_argument("--first", type=int, default=1000)
    ap.add_argument("--refs", default=None)
    ap.add_argument("--arms", default="direct,f2x2,f4x4")
    ap.add_argument("--check", type=int, default=4, help="stored CPU forwards recomputed live on this machine and compared")
    ap.add_argument("--speed", action="store_true", help="hands/s of every arm at bz = 256 (shipped multi-stream mode)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "hl_ab.json"))
    a = ap.parse_args()
    import numpy as np
    _init(a.workload)
    hamer = a.workload == "hamer_light"
    if a.precompute:
        out = os.path.join(ROOT, "build_ab", f"{'hm' if hamer else 'hl'}_refs_{a.first}_{a.seeds}.npz")
        os.makedirs(os.path.dirname(out), exist_ok=True)
        v32s, v64s = [], []
        t0 = time.time()
        for i, seed in enumerate(range(a.first, a.first + a.seeds)):
            v32, v64 = _refs(seed)
            v32s.append(v32); v64s.append(v64)
            if (i + 1) % 50 == 0 or i + 1 == a.seeds:
                np.savez(out, first=a.first, v32=np.stack(v32s), v64=np.stack(v64s))
                print(f"[{time.time() - t0:6.0f} s] {i + 1}", flush=True)
        return
    import torch
    import hands_amd
    from hands_amd.weights import synthetic_inputs
    from hon_parity_ab import wilson
    z = np.load(a.refs)
    first, n_have = int(z["first"]), len(z["v32"])
    seeds = [s for s in range(a.first, a.first + a.seeds) if first <= s < first + n_have]
    for s in seeds[:a.check]:
        v32, v64 = _refs(s)
        print(f"refs check seed {s}: stored vs live fp32 {np.abs(z['v32'][s - first] - v32).max():.3e}, "
              f"fp64 {np.abs(z['v64'][s - first] - v64).max():.3e}", flush=True)
    arms = ["default"] if hamer else a.arms.split(",")
    models = {}
    for name in arms:
        route, _, rest = name.partition("+c")          # "+c<limit>[k<min_k>]": blocked summation inside the launch
        if hamer:
            models[name] = hands_amd.apply_recipe(hands_amd.HAMER()).to("cuda").eval()
            continue
        m = hands_amd.apply_recipe(hands_amd.HandsLight())
        if route != "f4x4":
            m.winograd4_stages = ()
        m = m.to("cuda").eval()
        m.engine.winograd = route != "direct"
        if rest:
            rest, _, mp = rest.partition("p")
            lim, _, mk = rest.partition("k")
            m.engine.chain_limit, m.engine.chain_min_k, m.engine.chain_in_kernel = int(lim), int(mk or 0), True
            m.engine.chain_max_pix = int(mp or 0)
        m.async_tail = False
        models[name] = m
    res = {"arms": arms, "seeds": [], "ref32_vs_64": [], "vs32": {n: [] for n in arms}, "vs64": {n: [] for n in arms}}
    t0 = time.time()
    for i, seed in enumerate(seeds):
        v32, v64 = z["v32"][seed - first], z["v64"][seed - first]
        ci, cm = synthetic_inputs(1 if hamer else 2, seed)
        gi, gm = {k: v.to("cuda") for k, v in ci.items()}, {k: v.to("cuda") for k, v in cm.items()}
        res["seeds"].append(seed)
        res["ref32_vs_64"].append(float(np.abs(v32.astype(np.float64) - v64).max()))
        for name, m in models.items():
            out = m(gi, gm)
            torch.cuda.synchronize()
            v = torch.stack([out[f"mano.vertices.{h}"] for h in "rl"]).cpu().numpy()
            res["vs32"][name].append(float(np.abs(v - v32).max()))
            res["vs64"][name].append(float(np.abs(v.astype(np.float64) - v64).max()))
        if (i + 1) % 100 == 0 or i + 1 == len(seeds):
            print(f"[{time.time() - t0:6.0f} s] {i + 1} seeds", flush=True)
    n = len(res["seeds"])
    r = np.array(res["ref32_vs_64"])
    summary = {"n": n, "ref32_vs_64": {"median": float(np.median(r)), "p99": float(np.percentile(r, 99)), "max": float(r.max())}, "arms": {}}
    print(f"oracle fp32 vs fp64 (the floor): median {np.median(r):.3e} p99 {np.percentile(r, 99):.3e} max {r.max():.3e}")
    for name in arms:
        e, e64 = np.array(res["vs32"][name]), np.array(res["vs64"][name])
        k = int((e > 1e-6).sum())
        lo, hi = wilson(k, n)
        ratio = e64 / np.maximum(r, 1e-12)
        summary["arms"][name] = {"exceed": k, "exceed_rate": k / n, "wilson95": [lo, hi], "median": float(np.median(e)),
                                 "p90": float(np.percentile(e, 90)), "p99": float(np.percentile(e, 99)), "max": float(e.max()),
                                 "vs64_median": float(np.median(e64)), "vs64_p99": float(np.percentile(e64, 99)), "vs64_max": float(e64.max()),
                                 "median_ratio_hip64_over_ref64": float(np.median(ratio))}
        print(f"{name:8s} vs ref32: > 1e-6 {k:3d}/{n} = {100 * k / n:.2f} % [{100 * lo:.2f}, {100 * hi:.2f}]  median {np.median(e):.3e} "
              f"p90 {np.percentile(e, 90):.3e} p99 {np.percentile(e, 99):.3e} max {e.max():.3e} | vs fp64: median {np.median(e64):.3e} "
              f"p99 {np.percentile(e64, 99):.3e} max {e64.max():.3e}, median err(HIP,64) / err(ref32,64) = {np.median(ratio):.2f}")
    if a.speed and not hamer:        # arms alternate (clocks drift over a call): four rounds of six forwards each, best and median per arm
        gi, gm = synthetic_inputs(256, 0, device="cuda")
        times = {name: [] for name in models}
        for name, m in models.items():
            m.async_tail = True
            for _ in range(2):
                out = m(gi, gm)
            torch.cuda.synchronize()
        for _ in range(4):
            for name, m in models.items():
                out = m(gi, gm)
                torch.cuda.synchronize()
                t = time.perf_counter()
                for _ in range(6):
                    out = m(gi, gm)
                torch.cuda.synchronize()
                times[name].append((time.perf_counter() - t) / 6)
        for name, ts in times.items():
            summary["arms"][name]["hands_per_s_bz256"] = 512 / min(ts)
            summary["arms"][name]["hands_per_s_bz256_median"] = 512 / float(np.median(ts))
            print(f"{name:18s} bz 256: best {512 / min(ts):8.1f}  median {512 / float(np.median(ts)):8.1f} hands/s", flush=True)
    res["summary"] = summary
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(res, open(a.out, "w"))
    json.dump(summary, open(a.out.replace(".json", "_summary.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
