#!/usr/bin/env python3
"""bench.py -- hands/sec of the hands_light forward path (BASELINE.json config 2) on N MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--bz 256]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one HandsLight.forward over bz=256 synthetic samples per GPU (1 global image + right crop
+ left crop each = 2 hands; 768 ResNet-50 trunk passes) followed, when N > 1, by one RCCL
all-gather of the packed predictions.  Inputs are resident in HBM before the timed region; outputs
stay on the device.  Weak scaling: per-GPU work is fixed, value = N * 2 * bz * K / t.

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (conv_igemm_f32, every
convolution / linear layer of the path): achieved = algorithmic FLOPs of all its launches in one
step / the summed duration of those launches, each bracketed by HIP events on the launch stream in
a separate instrumented step.  `cpu_baseline` times the CPU oracle (a torch-CPU port of the
reference path) on the box's host cores on a bounded sample and doubles as the MPJPE checker.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HANDS_SYNTHETIC_MANO", "1")   # synthetic MANO asset (data: "synthetic"); the licensed files are not here

import torch
import torch.distributed as dist

FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD


def host_cores():
    """Cores this process may really use: min(affinity, cgroup cpu.max quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


def pmc_traffic_per_launch(workload):
    """(GB per conv_igemm launch, source) from the committed rocprofv3 --pmc summary, or (None, None)."""
    fn = os.path.join(ROOT, "profiles", f"r01_pmc_{workload}.json")
    try:
        d = json.load(open(fn))
        return round(d["hbm_gb_per_launch"], 4), os.path.relpath(fn, ROOT)
    except (OSError, ValueError, KeyError):
        return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--bz", type=int, default=0, help="samples per GPU per step (2 hands each); "
                    "default 256 (hands_light) / 64 (hamer_light)")
    ap.add_argument("--workload", default="hands_light", choices=["hands_light", "hamer_light", "handoccnet_light"],
                    help="hands_light = BASELINE.json configs[1] (the headline metric); hamer_light = configs[2]; "
                         "handoccnet_light = configs[3] (bz = 256/8 per GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--serial", action="store_true",
                    help="one HIP stream for everything (every launch alone on the chip): the mode the "
                         "per-kernel rocprofv3 averages in profiles/ are taken in")
    ap.add_argument("--latency-mode", action="store_true",
                    help="HandsLight.latency_mode: split-K on every launch with <= 128 output tiles (small-batch serving)")
    ap.add_argument("--cpu-bz", type=int, default=16)
    ap.add_argument("--layer-report", default="", help="write a per-launch CSV of the GEMM kernel here")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # HANDS_BENCH_SHARE_GPU=1 + HANDS_BENCH_BACKEND=gloo: dry-run of the N>1 code path on a 1-GPU box
    # (every rank on cuda:0, gloo collectives); never set by the driver.
    share = os.environ.get("HANDS_BENCH_SHARE_GPU") == "1"
    backend = os.environ.get("HANDS_BENCH_BACKEND", "nccl")
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import hands_amd
    from hands_amd.dist import gather_predictions
    from hands_amd.hands_light import HandsLight

    hamer = args.workload == "hamer_light"
    handocc = args.workload == "handoccnet_light"
    if not args.bz:
        args.bz = {"hamer_light": 64, "handoccnet_light": 32}.get(args.workload, 256)
    ctor = {"hamer_light": hands_amd.HAMER, "handoccnet_light": hands_amd.HandOccNet}.get(args.workload, hands_amd.HandsLight)
    model = hands_amd.apply_recipe(ctor())
    flop_per_hand = {"hamer_light": 251e9, "handoccnet_light": 36.2e9}.get(args.workload, 12.77e9)
    sd_cpu = {k: v.clone() for k, v in model.state_dict().items()} if rank == 0 else None
    model = model.to(dev).eval()
    if args.latency_mode:
        HandsLight.latency_mode = True
    if args.serial:
        HandsLight.overlap_trunks = False
    if os.environ.get("HANDS_CHUNKS"):
        HandsLight.trunk_chunks = tuple(int(v) for v in os.environ["HANDS_CHUNKS"].split(","))
    bz = args.bz
    inputs, meta = hands_amd.synthetic_inputs(bz, seed=rank, device=dev)

    def step():
        out = model(inputs, meta)
        if world > 1 and backend != "nccl":      # dry-run only: gloo gathers host tensors
            torch.cuda.synchronize(dev)
            return gather_predictions({k: v.cpu() for k, v in out.items()})
        return gather_predictions(out) if world > 1 else out

    for _ in range(args.warmup):
        step()

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    cpu_coll = world > 1 and backend != "nccl"

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if backend != "nccl" else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    n_gpus = world
    ms_per_step = elapsed / args.steps * 1e3
    hands_per_s = n_gpus * 2 * bz * args.steps / elapsed

    # ---- roofline of the dominant kernel: instrumented step, HIP events around every launch ----
    events = []
    macs = [0]

    launch_info = []
    conv_bytes = [0.0]

    main_stream = torch.cuda.current_stream(dev)

    def hook(phase, pc, npix, stream_handle, has_res):
        # the instrumented forward runs on ONE stream (overlap_trunks=False): the stream handed to
        # the C ABI is torch's current stream, so the events bracket exactly this launch
        assert stream_handle == main_stream.cuda_stream
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(main_stream)
        events.append(ev)
        if phase == "begin":
            macs[0] += pc.macs_per_pixel * npix
            # algorithmic bytes: input read once + output written once + weights once (fp32)
            conv_bytes[0] += 4.0 * (npix * pc.Cout * (2 if has_res else 1) + npix * pc.stride * pc.stride * pc.Cin +
                                    pc.w.numel())
            launch_info.append((pc.Cin, pc.Cout, pc.KH, pc.stride, npix, pc.macs_per_pixel * npix))

    n_prof = 3
    HandsLight.conv_hook = staticmethod(hook)
    HandsLight.overlap_trunks = False     # one stream: every launch is timed alone on the chip
    for _ in range(n_prof):
        model(inputs, meta)
    torch.cuda.synchronize(dev)
    HandsLight.conv_hook = None
    HandsLight.overlap_trunks = not args.serial
    durs_ms = [events[i].elapsed_time(events[i + 1]) for i in range(0, len(events), 2)]
    launches = len(durs_ms) // n_prof
    conv_ms = sum(durs_ms) / n_prof
    conv_flops = 2.0 * macs[0] / n_prof
    achieved = conv_flops / (conv_ms * 1e-3) / 1e12
    if args.layer_report:
        with open(args.layer_report, "w") as fh:
            fh.write("idx,Cin,Cout,k,stride,M,gflop,ms,tflops\n")
            for i in range(launches):
                ms = sum(durs_ms[i + r * launches] for r in range(n_prof)) / n_prof
                cin, cout, k, st, npix, mc = launch_info[i]
                fh.write(f"{i},{cin},{cout},{k},{st},{npix},{2 * mc / 1e9:.3f},{ms:.4f},{2 * mc / ms / 1e9:.2f}\n")
    # HBM traffic per launch of this kernel from the committed PMC passes (FETCH_SIZE doubled as
    # MI355X_MICROARCH.md prescribes for gfx950, + WRITE_SIZE; separate --pmc runs, profiles/README.md)
    traffic, traffic_src = pmc_traffic_per_launch(args.workload)
    roofline = {"bound": "mfma", "kernel": "conv_igemm_f32_kernel", "achieved": round(achieved, 2),
                "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
                "traffic": traffic, "traffic_unit": "GB/launch (PMC)", "traffic_source": traffic_src,
                "algorithmic_gb_per_launch": round(conv_bytes[0] / n_prof / launches / 1e9, 4),
                "launches_per_step": launches,
                "avg_launch_us": round(conv_ms * 1e3 / launches, 2),
                "kernel_ms_per_step": round(conv_ms, 3),
                "algorithmic_gflop_per_sample": round(conv_flops / bz / 1e9, 3)}

    # ---- CPU baseline (oracle = torch-CPU port of the reference path) + MPJPE checker ----------
    cpu_baseline = None
    parity = None
    if not args.no_cpu_baseline and world == 1:      # rank 0 at N=1 only (bounded sample, host cores)
        from oracle import hands_oracle as O
        cb = 2 if hamer else (8 if handocc else args.cpu_bz)
        ci, cm = hands_amd.synthetic_inputs(cb, seed=0)
        ar, al = hands_amd.synthetic_mano_asset(True), hands_amd.synthetic_mano_asset(False)
        cores = host_cores()
        best = None
        ref = None
        if hamer:
            from oracle import hamer_oracle as HO
            oracle_fwd = HO.hamer_forward
        elif handocc:
            from oracle import handoccnet_oracle as HOC
            oracle_fwd = HOC.handoccnet_forward
        else:
            oracle_fwd = O.hands_light_forward
        for nthreads in sorted({max(1, cores // 2), cores}):
            torch.set_num_threads(nthreads)
            ref = oracle_fwd(sd_cpu, ar, al, ci, cm)      # warm-up + checker output
            times = []
            t_budget = time.perf_counter()
            while len(times) < 5 and (time.perf_counter() - t_budget) < 12.0:
                t1 = time.perf_counter()
                oracle_fwd(sd_cpu, ar, al, ci, cm)
                times.append(time.perf_counter() - t1)
            med = sorted(times)[len(times) // 2]
            if best is None or med < best[0]:
                best = (med, nthreads, len(times))
        med, nthreads, nruns = best
        cpu_baseline = {"value": round(2 * cb / med, 2), "unit": "hands/s", "cores": nthreads, "kind": "port",
                        "sample": f"oracle (torch-CPU port of the reference path) {args.workload} forward, bz={cb} "
                        f"({2 * cb} hands), median of {nruns} runs, fp32; host allows {cores} cores "
                        f"(cgroup quota / affinity), best of {{cores/2, cores}} threads"}
        got = model({k: v.to(dev) for k, v in ci.items()}, {k: v.to(dev) for k, v in cm.items()})
        verr = max((got[f"mano.vertices.{h}"].cpu() - ref[f"mano.vertices.{h}"]).abs().max().item() for h in "rl")
        mp = max(O.mpjpe_ra_mm(got[f"mano.joints3d.{h}"].cpu(), ref[f"mano.joints3d.{h}"]) for h in "rl")
        parity = {"mpjpe_vs_ref_mm": round(mp, 7), "max_vertex_err_m": float(f"{verr:.3e}"), "checked_hands": 2 * cb}

    line = {
        "metric": "hands/sec", "value": round(hands_per_s, 1), "unit": "hands/s", "n_gpus": n_gpus,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": (f"hamer_light ViT-H/16 (256x192) + decoder head + MANO, bz={bz} samples/GPU "
                                f"({2 * bz} crops = hands)" if hamer else
                                f"handoccnet_light LeakyReLU ResNet-50 + FPN + FIT/SET + hourglass regressor + MANO, "
                                f"256x256, bz={bz} samples/GPU ({2 * bz} crops = hands)" if handocc else
                                "hands_light ResNet-50 x3 + feature_conv + HMR + MANO, 224x224, "
                                f"bz={bz} samples/GPU ({2 * bz} hands, {3 * bz} trunk passes)"),
                   "per_gpu_batch": bz, "global_batch": bz * n_gpus, "img_res": 224,
                   "parallelism": f"dp{n_gpus}" + ("+allgather" if n_gpus > 1 else ""),
                   "latency_mode": bool(args.latency_mode)},
        "hands_per_sec_per_gpu": round(hands_per_s / n_gpus, 1),
        "path_tflops": round(hands_per_s * flop_per_hand / 1e12 / n_gpus, 2),
        "roofline": roofline, "cpu_baseline": cpu_baseline, "parity": parity,
    }
    print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
