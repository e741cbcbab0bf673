#!/usr/bin/env python3
"""hands_light throughput with N replicas (HandsLight.replica(): same packed weights, own workspaces / streams / engine) taking
alternate batches on their own torch streams -- two whole forwards in flight (dev tool, GPU box).  usage: python tools/replica_speed.py [bz]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HANDS_SYNTHETIC_MANO", "1")
import torch

import hands_amd

bz = int(sys.argv[1]) if len(sys.argv) > 1 else 256
model = hands_amd.apply_recipe(hands_amd.HandsLight()).to("cuda").eval()
inputs, meta = hands_amd.synthetic_inputs(bz, seed=0, device="cuda")
for n in (1, 2, 3, 1, 2):
    reps = [model] + [model.replica() for _ in range(n - 1)]
    streams = [torch.cuda.Stream() for _ in reps]

    def run(steps):
        for i in range(steps):
            with torch.cuda.stream(streams[i % n]):
                reps[i % n](inputs, meta)
        torch.cuda.synchronize()

    run(2 * n)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        run(12)
        best = min(best, (time.perf_counter() - t0) / 12)
    print(f"{n} replica(s): {2 * bz / best:8.1f} hands/s ({best * 1e3:.2f} ms per forward)", flush=True)
    del reps
    torch.cuda.empty_cache()
