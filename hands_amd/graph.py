"""hipGraph capture of a forward pass for fixed shapes (small-batch serving, and the 8-GPU shard sizes).

At bz <= 8 one `HandsLight.forward` is ~250 kernel launches of a few microseconds each on five HIP
streams; the host needs ~2 ms to enqueue them, which is as long as the GPU needs to run them in
`latency_mode`.  `GraphedForward` captures the whole forward (all streams, fork/join events included)
into one hipGraph and replays it with a single launch.  It takes any of the three models
(`HandsLight`, `HAMER`, `HandOccNet`): under capture each of them runs its synchronous form (no
asynchronous tail, no pipelined call), every workspace comes from the graph's private memory pool.

``depth=2`` keeps TWO forwards in flight, as `HandOccNet.async_forward` does for eager calls: two
captured instances with their own static buffers, call i replayed on pipeline stream i & 1 and
joined at the first use of its result (`stream_xdict`).  At the 8-GPU shard size of handoccnet_light
(32 samples per GPU) a launch is 1-4 tiles per CU, so two forwards in flight fill each other's
tails; the graph removes the host enqueue of the ~230 launches of each.

The reference has no counterpart (it runs eager PyTorch, `src/models/generic/wrapper.py:68-75`);
outputs are bit-identical to the eager call of this package with the same flags.
"""
from __future__ import annotations

import torch

from .xdict import stream_xdict


class GraphedForward:
    def __init__(self, model, inputs, meta_info, warmup: int = 2, depth: int = 1):
        dev = next(v for v in inputs.values() if torch.is_tensor(v)).device
        if dev.type != "cuda":
            raise RuntimeError("GraphedForward needs a HIP device (no CPU fallback)")
        if depth < 1:
            raise ValueError("depth >= 1")
        if depth > 1 and not getattr(model, "graph_private_buffers", False):
            # HandsLight / HAMER keep persistent per-model workspaces: two captured instances in flight would share them
            raise ValueError(f"GraphedForward(depth={depth}) needs a model whose forward allocates every buffer per call "
                             "(HandOccNet); use depth=1, or one GraphedForward per HandsLight.replica()")
        own = lambda d: {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in d.items()}
        self.model, self.dev, self.depth = model, dev, int(depth)
        self._calls = 0
        self._tokens = []
        self.static_in, self.static_meta, self.static_out, self.graphs = [], [], [], []
        self._done = [None] * self.depth          # event of the last replay of instance i (pipelined mode)
        self._hold = [None] * self.depth          # event behind which instance i's static outputs are still being read
        self._last = 0
        self._streams = [torch.cuda.Stream(device=dev) for _ in range(self.depth)] if self.depth > 1 else []
        side = torch.cuda.Stream(device=dev)
        for i in range(self.depth):
            s_in, s_meta = own(inputs), own(meta_info)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):             # warm-up: every workspace / packed weight exists before capture
                for _ in range(max(1, warmup) if i == 0 else 1):
                    dict(model(s_in, s_meta).items())       # (.items() joins an asynchronous result on `side`)
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            eng = getattr(model, "engine", None)
            if eng is not None:                   # split-K workspaces of THIS capture come from its own memory pool; the tables
                self._tokens.append((eng, eng.begin_capture()))     # of other live captures on a shared engine stay
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = model(s_in, s_meta)
            self.static_in.append(s_in)
            self.static_meta.append(s_meta)
            self.static_out.append(out)
            self.graphs.append(g)
        self.graph = self.graphs[0]               # depth == 1 names (kept: callers and tests use them)

    @staticmethod
    def _load(dst, src):
        for k, v in src.items():
            if torch.is_tensor(v):
                if k not in dst or dst[k].shape != v.shape:
                    raise ValueError(f"GraphedForward was captured for other shapes: {k} {tuple(v.shape)}")
                dst[k].copy_(v, non_blocking=True)

    def __call__(self, inputs, meta_info):
        """Copies the tensors into the captured buffers, replays, returns the (static) output xdict --
        clone what must survive the next ``depth`` calls.  depth > 1: the result is a `stream_xdict` joined at
        its first use; the caller may overwrite its inputs as soon as this returns."""
        i = self._calls % self.depth
        if self._hold[i] is not None:             # a consumer registered with hold_until() still reads this instance's outputs
            torch.cuda.current_stream(self.dev).wait_event(self._hold[i])     # (the wait may stay even if the loads below raise)
        if self.depth == 1:
            self._load(self.static_in[0], inputs)
            self._load(self.static_meta[0], meta_info)
            # bookkeeping only after both loads went through: a shape error leaves the slot, its hold and `_last` where they were
            self._calls += 1
            self._hold[i], self._last = None, i
            self.graphs[0].replay()
            return self.static_out[0]
        main = torch.cuda.current_stream(self.dev)
        if self._done[i] is not None:             # instance i's previous replay still reads its static inputs
            main.wait_event(self._done[i])
        self._load(self.static_in[i], inputs)     # on the CALLER's stream: its tensors are free when we return
        self._load(self.static_meta[i], meta_info)
        self._calls += 1                          # (a shape error above leaves the pipeline slot, its hold and `_last` where they were)
        self._hold[i], self._last = None, i
        ev = torch.cuda.Event()
        ev.record(main)
        st = self._streams[i]
        st.wait_event(ev)
        with torch.cuda.stream(st):
            self.graphs[i].replay()
            ready = torch.cuda.Event()
            ready.record(st)
        self._done[i] = ready
        return stream_xdict(self.static_out[i], ready, self.dev)

    def hold_until(self, event):
        """The (static) outputs returned by the LAST call are read asynchronously -- e.g. packed and all-gathered on a side stream
        (`hands_amd.dist.gather_predictions` of a pending result) -- until ``event``: the next replay of that captured instance,
        ``depth`` calls from now, waits for it instead of overwriting them under the reader."""
        self._hold[self._last] = event

    def synchronize(self):
        for ev in self._done:
            if ev is not None:
                ev.synchronize()

    def __del__(self):
        for eng, token in getattr(self, "_tokens", ()):
            eng.drop_capture(token)
