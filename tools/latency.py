import os, sys, time, torch
sys.path.insert(0, ".")
os.environ.setdefault("HANDS_SYNTHETIC_MANO", "1")
import hands_amd
from hands_amd.weights import synthetic_inputs
dev = torch.device("cuda:0")
m = hands_amd.apply_recipe(hands_amd.HandsLight()).to(dev).eval()
import numpy as np
for mode in (False, True):
  m.latency_mode = mode
  print("latency_mode", mode)
  for bz in (1, 2, 8, 32):
      inputs, meta = synthetic_inputs(bz, seed=0)
      inputs = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inputs.items()}
      meta = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in meta.items()}
      for _ in range(5): out = m(inputs, meta)
      torch.cuda.synchronize()
      v = out["mano.vertices.r"].clone()
      if not mode: ref_v = globals().setdefault("REFV", {}); ref_v[bz] = v
      else: print(f"   max |vertices - default mode| = {(v - REFV[bz]).abs().max().item():.2e} m")
      t0 = time.perf_counter()
      n = 30
      for _ in range(n): m(inputs, meta)
      torch.cuda.synchronize()
      dt = (time.perf_counter() - t0) / n
      # host enqueue time only
      t0 = time.perf_counter()
      for _ in range(n): m(inputs, meta)
      t_enq = (time.perf_counter() - t0) / n
      torch.cuda.synchronize()
      gf = hands_amd.GraphedForward(m, inputs, meta)
      og = gf(inputs, meta); torch.cuda.synchronize()
      same = all(torch.equal(og[k], out[k]) for k in out)
      t0 = time.perf_counter()
      for _ in range(n): gf(inputs, meta)
      torch.cuda.synchronize()
      dg = (time.perf_counter() - t0) / n
      print(f"   hipGraph replay {dg*1e3:.2f} ms/forward ({2*bz/dg:.0f} hands/s), bit-identical to eager: {same}")
      print(f"bz={bz}: {dt*1e3:.2f} ms/forward ({2*bz/dt:.0f} hands/s), host enqueue {t_enq*1e3:.2f} ms")
