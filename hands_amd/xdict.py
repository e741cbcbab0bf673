"""Strict dictionary used as the output container of ``HandsLight.forward``.

Mirrors the behaviour of the reference container (reference: common/xdict.py:26-258 and
common/ld_utils.py:12-14) that callers of the hot path rely on:

* ``d[k] = v`` raises ``AssertionError`` when ``k`` is already present (xdict.py:50-55);
* ``merge`` raises ``AssertionError`` on overlapping keys (xdict.py:89-104);
* ``prefix`` / ``postfix`` / ``replace_keys`` / ``search`` / ``rm`` / ``subset`` return new containers;
* ``detach`` moves every tensor to the CPU, recursively through lists/tuples/dicts
  (xdict.py:233-241 -> common/thing.py:57-66);
* ``has_invalid`` only *reports* NaN/Inf (xdict.py:243-258).

Written from the behaviour, not from the text, of the reference.
"""
from __future__ import annotations

import numpy as np
import torch


def _to_cpu(obj):
    if isinstance(obj, torch.Tensor):
        return obj.detach().cpu()
    if isinstance(obj, tuple):
        return tuple(_to_cpu(o) for o in obj)
    if isinstance(obj, list):
        return [_to_cpu(o) for o in obj]
    if isinstance(obj, dict):
        return {k: _to_cpu(v) for k, v in obj.items()}
    return obj


def _to_dev(obj, dev):
    if isinstance(obj, torch.Tensor):
        return obj.to(dev)
    if isinstance(obj, list):
        return [_to_dev(o, dev) for o in obj]
    if isinstance(obj, dict):
        return {k: _to_dev(v, dev) for k, v in obj.items()}
    return obj


def _to_np(obj):
    if isinstance(obj, torch.Tensor):
        return obj.detach().cpu().numpy()
    if isinstance(obj, list):
        return np.array(obj)
    if isinstance(obj, dict):
        return {k: _to_np(v) for k, v in obj.items()}
    return obj


def _to_torch(obj):
    if isinstance(obj, np.ndarray):
        return torch.from_numpy(obj)
    if isinstance(obj, list):
        return torch.tensor(np.array(obj))
    if isinstance(obj, dict):
        return {k: _to_torch(v) for k, v in obj.items()}
    return obj


class xdict(dict):
    def __init__(self, mydict=None):
        super().__init__()
        if mydict is not None:
            for k, v in mydict.items():
                dict.__setitem__(self, k, v)

    # -- strict assignment ---------------------------------------------------------------------
    def __setitem__(self, key, val):
        assert key not in self, f"Key already exists {key}"
        dict.__setitem__(self, key, val)

    def overwrite(self, k, v):
        dict.__setitem__(self, k, v)

    def merge(self, dict2):
        assert isinstance(dict2, dict)
        dup = set(self.keys()) & set(dict2.keys())
        assert len(dup) == 0, f"Merge failed: duplicate keys ({dup})"
        self.update(dict2)

    # -- key transforms ------------------------------------------------------------------------
    def prefix(self, text):
        return xdict({text + k: v for k, v in self.items()})

    def postfix(self, text):
        return xdict({k + text: v for k, v in self.items()})

    def replace_keys(self, str_src, str_tar):
        return xdict({k.replace(str_src, str_tar): v for k, v in self.items()})

    def subset(self, keys):
        return xdict({k: self[k] for k in keys})

    def search(self, keyword, replace_to=None):
        out = {}
        for k, v in self.items():
            if keyword in k:
                out[k if replace_to is None else k.replace(keyword, replace_to)] = v
        return xdict(out)

    def rm(self, keyword, keep_list=(), verbose=False):
        out = {}
        for k, v in self.items():
            if keyword not in k or k in keep_list:
                out[k] = v
            elif verbose:
                print(f"Removing: {k}")
        return xdict(out)

    def sorted_keys(self):
        return sorted(self.keys())

    # -- value transforms ----------------------------------------------------------------------
    def mul(self, scalar):
        if isinstance(scalar, int):
            scalar = float(scalar)
        assert isinstance(scalar, float)
        out = {}
        for k, v in self.items():
            out[k] = [x * scalar for x in v] if isinstance(v, list) else v * scalar
        return xdict(out)

    def apply(self, operation, criterion=None):
        return xdict({k: operation(v) for k, v in self.items() if criterion is None or criterion(k, v)})

    def to(self, dev):
        if dev is None:
            return self
        return xdict(_to_dev(dict(self), dev))

    def to_np(self):
        return xdict(_to_np(dict(self)))

    def to_torch(self):
        return xdict(_to_torch(dict(self)))

    def detach(self):
        return xdict(_to_cpu(dict(self)))

    def has_invalid(self):
        for k, v in self.items():
            if isinstance(v, torch.Tensor):
                if torch.isnan(v).any():
                    print(f"{k} contains nan values")
                    return True
                if torch.isinf(v).any():
                    print(f"{k} contains inf values")
                    return True
        return False

    def print_stat(self):
        for k, v in self.items():
            if isinstance(v, (torch.Tensor, np.ndarray)):
                print(f"{k:<20}: {tuple(v.shape)}\t{type(v).__name__}")
            elif isinstance(v, (list, tuple)):
                print(f"{k:<20}: len {len(v)}\t{type(v).__name__}")
            else:
                print(f"{k:<20}: {type(v).__name__}")

    def save(self, path, dev=None, verbose=True):
        if verbose:
            print(f"Saving to {path}")
        torch.save(self.to(dev), path)


class stream_xdict(xdict):
    """An xdict whose tensors are being produced on a side HIP stream (the tail of ``HandsLight.forward``:
    feature_conv, HMR heads, MANO, grasp), so that the caller's stream is free to start the next forward's
    trunks meanwhile.  Plain PyTorch stream semantics are kept at the point of use: the first access of any
    kind (``d[k]``, iteration, ``len``, ``dict(d)``, ``.items()``, ``.detach()`` ...) makes the CURRENT stream
    wait for the producer and registers the tensors with it; from then on it is an ordinary xdict.  A caller
    that never looks at the result (a throughput loop) never waits; ``torch.cuda.synchronize()`` covers the
    side stream as it covers every stream."""

    def __init__(self, pending: dict, ready_event, device):
        super().__init__()
        self.__dict__["_pending"] = pending
        self.__dict__["_ready"] = ready_event
        self.__dict__["_device"] = device

    def _join(self):
        p = self.__dict__.get("_pending")
        if p is not None:
            self.__dict__["_pending"] = None
            cur = torch.cuda.current_stream(self.__dict__["_device"])
            cur.wait_event(self.__dict__["_ready"])
            for v in p.values():
                if isinstance(v, torch.Tensor) and v.is_cuda:
                    v.record_stream(cur)
            dict.update(self, p)

    @property
    def is_pending(self):
        return self.__dict__.get("_pending") is not None


def _joined(name):
    base = getattr(dict, name)

    def method(self, *a, **k):
        self._join()
        return base(self, *a, **k)
    method.__name__ = name
    return method


# every Python-level entry point joins first; overriding __iter__ / keys also takes dict(d), {**d} and
# dict.update(other, d) off CPython's exact-dict fast path, so they go through these methods too
for _n in ("__getitem__", "__iter__", "__len__", "__contains__", "keys", "values", "items", "get", "__repr__",
           "__eq__", "__ne__", "pop", "popitem", "copy", "update", "setdefault", "__delitem__", "__reversed__",
           "__or__", "__ror__", "__ior__", "clear", "__reduce_ex__", "__sizeof__"):
    setattr(stream_xdict, _n, _joined(_n))


def _xd_setitem(self, key, val):
    self._join()
    xdict.__setitem__(self, key, val)


def _xd_overwrite(self, k, v):
    self._join()
    dict.__setitem__(self, k, v)


stream_xdict.__setitem__ = _xd_setitem
stream_xdict.overwrite = _xd_overwrite
stream_xdict.__hash__ = None


def prefix_dict(mydict, prefix):
    """reference: common/ld_utils.py:12-14 (returns a plain dict)."""
    return {prefix + k: v for k, v in mydict.items()}
