"""hipGraph capture of a forward pass for fixed shapes (small-batch serving).

At bz <= 8 one `HandsLight.forward` is ~250 kernel launches of a few microseconds each on five HIP
streams; the host needs ~2 ms to enqueue them, which is as long as the GPU needs to run them in
`latency_mode`.  `GraphedForward` captures the whole forward (all streams, fork/join events included)
into one hipGraph and replays it with a single launch.  The reference has no counterpart (it runs
eager PyTorch, `src/models/generic/wrapper.py:68-75`); outputs are bit-identical to the eager call
of this package with the same flags.
"""
from __future__ import annotations

import torch


class GraphedForward:
    def __init__(self, model, inputs, meta_info, warmup: int = 2):
        dev = inputs["img"].device
        if dev.type != "cuda":
            raise RuntimeError("GraphedForward needs a HIP device (no CPU fallback)")
        own = lambda d: {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in d.items()}
        self.model = model
        self.static_in, self.static_meta = own(inputs), own(meta_info)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):             # warm-up: every workspace / packed weight exists before capture
            for _ in range(max(1, warmup)):
                model(self.static_in, self.static_meta)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.static_out = model(self.static_in, self.static_meta)

    def _load(self, dst, src):
        for k, v in src.items():
            if torch.is_tensor(v):
                if k not in dst or dst[k].shape != v.shape:
                    raise ValueError(f"GraphedForward was captured for other shapes: {k} {tuple(v.shape)}")
                dst[k].copy_(v, non_blocking=True)

    def __call__(self, inputs, meta_info):
        """Copies the tensors into the captured buffers, replays, returns the (static) output xdict --
        clone what must survive the next call."""
        self._load(self.static_in, inputs)
        self._load(self.static_meta, meta_info)
        self.graph.replay()
        return self.static_out
