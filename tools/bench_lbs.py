#!/usr/bin/env python3
"""BASELINE.json configs[4]: two-hand (left + right MANO) batched LBS, bs=1024 crops per node,
all-gather of the 778x3 vertices when N > 1.  `python tools/bench_lbs.py [--bz 1024] [--steps 50]`
or under torchrun (one rank per GPU, bz = 1024 / N per rank).  A step = MANOHead.forward for both
hands of every crop (rotmat -> axis-angle -> MANO LBS -> camera -> projection) = 3 launches per
side.  Prints one JSON line in bench.py's format.  Algorithmic work: 1.17 MFLOP and
10.2 KB of mandatory HBM traffic per hand."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HANDS_SYNTHETIC_MANO", "1")
import torch
import torch.distributed as dist


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--bz", type=int, default=0)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=300)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    import hands_amd
    from hands_amd import _lib
    from hands_amd.dist import gather_predictions
    from hands_amd.hands_light import run_mano_heads
    from oracle import hands_oracle as O   # input generator (6D -> R) + checker / CPU baseline only

    bz = args.bz or 1024 // world
    model = hands_amd.HandsLight().to(dev).eval()
    P = model.packed(dev)
    L = _lib.lib()
    g = torch.Generator().manual_seed(100 + rank)
    rot = O.rotation_6d_to_matrix(torch.randn(2 * bz * 16, 6, generator=g)).view(2 * bz, 16, 3, 3).contiguous()
    shape = torch.randn(2 * bz, 10, generator=g)
    cam = torch.tensor([1.0, 0, 0]) + 0.1 * torch.randn(2 * bz, 3, generator=g)
    K = torch.tensor([[1000.0, 0, 112], [0, 1000.0, 112], [0, 0, 1]]).repeat(bz, 1, 1)
    d_rot, d_shape, d_cam, d_K = rot.to(dev), shape.to(dev), cam.to(dev), K.to(dev)
    bufs = {}

    def buf(name, n):
        t = bufs.get(name)
        if t is None or t.numel() < n:
            t = bufs[name] = torch.empty(n, device=dev)
        return t

    def step():
        out = run_mano_heads(L, P["mano_r"], P["mano_l"], d_rot, d_shape, d_cam, d_cam, d_K, 224.0, bz,
                             torch.cuda.current_stream(dev).cuda_stream, buf)
        if world > 1:
            return gather_predictions({k: out[k] for k in ("mano.vertices.r", "mano.vertices.l")})
        return out

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        out = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    if rank != 0:
        dist.destroy_process_group()
        return
    hands = world * 2 * bz * args.steps / el
    # device-only time of the 6 launches (HIP events on the launch stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.steps):
        run_mano_heads(L, P["mano_r"], P["mano_l"], d_rot, d_shape, d_cam, d_cam, d_K, 224.0, bz,
                       torch.cuda.current_stream(dev).cuda_stream, buf)
    e1.record()
    torch.cuda.synchronize(dev)
    dev_ms = e0.elapsed_time(e1) / args.steps
    gbs = 2 * bz * 10.2e3 / (dev_ms * 1e-3) / 1e9
    tfl = 2 * bz * 1.17e6 / (dev_ms * 1e-3) / 1e12
    cpu_baseline = parity = None
    if not args.no_cpu_baseline and world == 1:
        cb = min(bz, 256)
        ar, al = hands_amd.synthetic_mano_asset(True), hands_amd.synthetic_mano_asset(False)
        cores = len(os.sched_getaffinity(0))
        try:
            q, p = open("/sys/fs/cgroup/cpu.max").read().split()
            cores = min(cores, max(1, int(q) // int(p))) if q != "max" else cores
        except (OSError, ValueError):
            pass
        torch.set_num_threads(cores)

        def cpu():
            r = O.mano_head(rot[:cb], shape[:cb], cam[:cb], K[:cb], ar, 224.0, ".r")
            l = O.mano_head(rot[bz:bz + cb], shape[bz:bz + cb], cam[bz:bz + cb], K[:cb], al, 224.0, ".l")
            return r, l
        ref = cpu()
        ts = []
        t_b = time.perf_counter()
        while len(ts) < 20 and time.perf_counter() - t_b < 15:
            t1 = time.perf_counter()
            cpu()
            ts.append(time.perf_counter() - t1)
        med = sorted(ts)[len(ts) // 2]
        cpu_baseline = {"value": round(2 * cb / med, 1), "unit": "hands/s", "cores": cores, "kind": "port",
                        "sample": f"oracle MANOHead (torch-CPU port) on {cb} right + {cb} left hands, median of {len(ts)} runs"}
        verr = max((out["mano.vertices.r"][:cb].cpu() - ref[0]["vertices.r"]).abs().max().item(),
                   (out["mano.vertices.l"][:cb].cpu() - ref[1]["vertices.l"]).abs().max().item())
        parity = {"max_vertex_err_m": float(f"{verr:.3e}"), "checked_hands": 2 * cb}
    print(json.dumps({
        "metric": "hands/sec", "value": round(hands, 1), "unit": "hands/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(el / args.steps * 1e3, 4), "higher_is_better": True,
        "scaling": "strong" if not args.bz else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"two-hand MANO LBS (MANOHead.forward x2), {bz} crops/GPU, {world} GPU(s)"
                               + (", all-gather of 778x3 vertices" if world > 1 else ""),
                   "per_gpu_batch": bz, "global_batch": bz * world},
        "roofline": {"bound": "hbm", "kernel": "mano_pose + blend GEMM + mano_skin (6 launches/step)",
                     "achieved": round(gbs, 2), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 5),
                     "traffic": None, "device_ms_per_step": round(dev_ms, 4), "us_per_launch": round(dev_ms * 1e3 / 6, 2),
                     "mfma_tflops": round(tfl, 3), "mfma_frac_of_fp32_peak": round(tfl / 157.3, 5),
                     "note": "latency-bound: 2.4 GFLOP and 21 MB of mandatory traffic per step at bz=1024"},
        "cpu_baseline": cpu_baseline, "parity": parity}), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
