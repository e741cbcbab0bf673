import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); os.environ.setdefault("HANDS_SYNTHETIC_MANO", "1")
import torch, hands_amd
m = hands_amd.apply_recipe(hands_amd.HandOccNet()).to("cuda").eval()
for bz, steps in ((32, 30), (256, 6)):
    inputs, meta = hands_amd.synthetic_inputs(bz, 0, device="cuda")
    for scope, cl in (("backbone", 0), ("backbone", 512), ("backbone", 256), ("all", 0), ("all", 512), ("all", 256), ("backbone", 0)):
        m.winograd_scope = scope; m.engine.chain_limit = cl; m.invalidate_packed()
        for _ in range(3): m(inputs, meta)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps): m(inputs, meta)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
        print(f"bz={bz} scope={scope} chain_limit={cl}: {dt*1e3:.2f} ms/step, {2*bz/dt:.1f} hands/s")
