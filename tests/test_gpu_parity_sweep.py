"""Multi-seed parity of the SHIPPED default routes against the oracle.

Winograd is the default 3x3 route of hands_light (F(4x4,3x3) in all four ResNet stages since round 5, every stride-1 3x3) and
of handoccnet_light (F(2x2,3x3) in every 3x3 / stride-1 layer outside the fp64 stages; src/models/handoccnet_light/backbone.py:44-65,68-119).  handoccnet_light amplifies any fp32 re-association, so one golden
seed is not evidence of the margin: here >= 8 input seeds per model run through the HIP default path and must stay within
the north-star bar -- max vertex error <= 1e-6 m (= 1e-3 mm) and root-aligned MPJPE <= 1e-3 mm -- against the oracle, the
two golden seeds also against the reference-generated fixtures.  The worst seed is printed and written to
gpurun_out/parity_sweep_<model>.json (profiles/README.md quotes it)."""
import json
import os

import numpy as np
import pytest
import torch

import hands_amd
from hands_amd.weights import synthetic_inputs
from oracle import handoccnet_oracle as HO
from oracle import hands_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
BAR_M, BAR_MPJPE_MM = 1e-6, 1e-3
SEEDS = range(10, 20)          # disjoint from the golden seeds (0-2) and the bench sweep (1-8)


def _sweep(model, sd, oracle_fwd, name):
    ar, al = hands_amd.synthetic_mano_asset(True), hands_amd.synthetic_mano_asset(False)
    rows = []
    for seed in SEEDS:
        ci, cm = synthetic_inputs(2, seed)
        ref = oracle_fwd(sd, ar, al, ci, cm)
        out = model({k: v.to(DEV) for k, v in ci.items()}, {k: v.to(DEV) for k, v in cm.items()})
        torch.cuda.synchronize()
        verr = max((out[f"mano.vertices.{h}"].cpu() - ref[f"mano.vertices.{h}"]).abs().max().item() for h in "rl")
        mp = max(O.mpjpe_ra_mm(out[f"mano.joints3d.{h}"].cpu(), ref[f"mano.joints3d.{h}"]) for h in "rl")
        rows.append({"seed": seed, "max_vertex_err_m": verr, "mpjpe_mm": mp})
    worst = max(rows, key=lambda r: r["max_vertex_err_m"])
    print(f"{name}: worst of {len(rows)} seeds: seed {worst['seed']} max vertex err {worst['max_vertex_err_m']:.3e} m "
          f"(bar {BAR_M:.0e}), MPJPE {max(r['mpjpe_mm'] for r in rows):.3e} mm; all: "
          + " ".join(f"{r['max_vertex_err_m']:.2e}" for r in rows))
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        json.dump({"model": name, "rows": rows, "worst": worst, "bar_m": BAR_M},
                  open(os.path.join(ROOT, "gpurun_out", f"parity_sweep_{name}.json"), "w"), indent=1)
    except OSError:
        pass
    return rows, worst


def test_hands_light_default_route_multi_seed_parity():
    model = hands_amd.apply_recipe(hands_amd.HandsLight())
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.to(DEV).eval()
    assert model.engine.winograd, "the shipped default of HandsLight is the Winograd route"
    rows, worst = _sweep(model, sd, O.hands_light_forward, "hands_light")
    assert len(rows) >= 8
    assert worst["max_vertex_err_m"] <= BAR_M and max(r["mpjpe_mm"] for r in rows) <= BAR_MPJPE_MM, worst


def test_handoccnet_default_scope_multi_seed_parity(golden_dir):
    model = hands_amd.apply_recipe(hands_amd.HandOccNet())
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.to(DEV).eval()
    assert model.engine.winograd and model.winograd_scope == "all" and model.engine.chain_limit == 64 and model.engine.chain_in_kernel \
        and model.acc64_stages == {"reghead", "mlp"} and model.engine.acc64, \
        "shipped default (round 6): Winograd in every 3x3 / stride-1 layer, direct chains <= 64 floats, heat-map head + MLPs in fp64"
    rows, worst = _sweep(model, sd, HO.handoccnet_forward, "handoccnet_light")
    assert len(rows) >= 8
    assert worst["max_vertex_err_m"] <= BAR_M and max(r["mpjpe_mm"] for r in rows) <= BAR_MPJPE_MM, worst
    # the two golden seeds against what the REFERENCE ITSELF produced (tests/golden/make_golden_handoccnet.py)
    for seed in (0, 1):
        d = np.load(os.path.join(golden_dir, f"handoccnet_light_bz2_seed{seed}.npz"))
        inputs, meta_info = synthetic_inputs(2, seed, device=DEV)
        out = model(inputs, meta_info)
        torch.cuda.synchronize()
        verr = max(np.abs(out[f"mano.vertices.{h}"].cpu().numpy() - d[f"out/mano.vertices.{h}"]).max() for h in "rl")
        print(f"handoccnet_light golden seed {seed}: max vertex err vs the reference's own output {verr:.3e} m")
        assert verr <= BAR_M, (seed, verr)


def test_handoccnet_error_distribution_guard():
    """48 more input seeds (200-247) through the shipped default route (round 6: Winograd everywhere + direct chains <= 64 floats +
    the heat-map head and the MLPs accumulated in fp64 + 32-key P V blocks + fp64 spatial softmax).  The 1000-seed A/B
    (tools/hon_parity_ab.py, profiles/r06_hon_parity_ab_1000seeds_summary.json) has 0 of 1000 inputs above 1e-6 m against the
    reference run with 1, 8 AND 16 ATen threads; 1 250 more inputs live on a GPU box: 0 against the 8-thread oracle, 1 against the
    1-thread one (1.02e-6) -- the rate at which the 8-thread oracle misses the 1-thread oracle (1 of 1 250, 1.07e-6).  Guard: NONE of the 48 above 1e-6 m (the oracle here is live, 8 threads, and also
    re-run with 1 thread: the reference's own vertices move by 2-5e-7 m with the thread count, the HIP path must stay inside the
    bar against both), median <= 5e-7, p90 <= 6.5e-7.

    And the reason it holds: the HIP path is CLOSER to an fp64 evaluation of the network than the reference's own fp32 arithmetic.
    Over the first 12 seeds the median of err(HIP, fp64) / err(fp32 oracle, fp64) is <= 0.95 (measured 0.74 over 1000 seeds; 1.02 in
    round 5) and the largest err(HIP, fp64) <= 4.5e-7 m."""
    model = hands_amd.apply_recipe(hands_amd.HandOccNet())
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    model = model.to(DEV).eval()
    ar, al = hands_amd.synthetic_mano_asset(True), hands_amd.synthetic_mano_asset(False)
    c64 = lambda d: {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in d.items()}
    vstack = lambda o: torch.stack([o[f"mano.vertices.{h}"] for h in "rl"]).double().cpu()
    errs, errs1, ratios, e64s = [], [], [], []
    threads = torch.get_num_threads()
    for seed in range(200, 248):
        ci, cm = synthetic_inputs(2, seed)
        ref = vstack(HO.handoccnet_forward(sd, ar, al, ci, cm))
        out = vstack(model({k: v.to(DEV) for k, v in ci.items()}, {k: v.to(DEV) for k, v in cm.items()}))
        errs.append((out - ref).abs().max().item())
        if seed < 224:                        # the same reference arithmetic with ANOTHER blocking of ATen's sums
            torch.set_num_threads(1)
            try:
                ref1 = vstack(HO.handoccnet_forward(sd, ar, al, ci, cm))
            finally:
                torch.set_num_threads(threads)
            errs1.append((out - ref1).abs().max().item())
        if seed < 212:
            r64 = vstack(HO.handoccnet_forward(sd64, ar, al, c64(ci), c64(cm)))
            e64s.append((out - r64).abs().max().item())
            ratios.append(e64s[-1] / max((ref - r64).abs().max().item(), 1e-12))
    e, e1 = np.sort(np.array(errs)), np.sort(np.array(errs1))
    print(f"handoccnet_light, 48 seeds vs the {threads}-thread oracle: median {np.median(e):.3e}, p90 {np.percentile(e, 90):.3e}, max {e[-1]:.3e}, "
          f"above 1e-6: {(e > 1e-6).sum()}; 24 seeds vs the 1-thread oracle: median {np.median(e1):.3e}, max {e1[-1]:.3e}; "
          f"err(HIP, fp64) / err(oracle fp32, fp64) over 12 seeds: median {np.median(ratios):.2f}, max {max(ratios):.2f}, max err(HIP, fp64) {max(e64s):.3e}")
    assert (e > 1e-6).sum() == 0 and (e1 > 1e-6).sum() == 0, (e[-4:], e1[-4:])
    assert np.median(e) <= 5e-7 and np.percentile(e, 90) <= 6.5e-7, e[-6:]
    assert np.median(ratios) <= 0.95 and max(e64s) <= 4.5e-7, (ratios, e64s)
