"""CPU affinity of a rank process: the cores of the NUMA node its GPU hangs off.

One process per GPU enqueues ~130-250 launches per forward; on an 8-GPU node (two sockets) a rank whose host thread runs
on the far socket pays a cross-socket hop on every doorbell write and pinned-memory copy.  ``pin_rank_to_gpu_node`` reads the
topology from sysfs only (KFD topology + the PCI device's ``local_cpulist`` / ``numa_node``) and calls
``os.sched_setaffinity`` -- no GPU call, no ``numactl``, no re-exec -- so a rank can (and must) do it BEFORE its first HIP
call.  The reference has no counterpart (its only multi-GPU code is Lightning DDP for training,
``scripts_method/train.py:57-73``).
"""
from __future__ import annotations

import glob
import os


def parse_cpulist(text: str) -> set:
    """'0-3,8,10-11' -> {0,1,2,3,8,10,11}"""
    cpus = set()
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            lo, hi = part.split("-")
            cpus.update(range(int(lo), int(hi) + 1))
        else:
            cpus.add(int(part))
    return cpus


def kfd_gpu_pci_addresses(sysfs: str = "/sys") -> list:
    """PCI addresses ('dddd:bb:dd.f') of the GPUs in KFD node order (= HIP's default device order)."""
    nodes = []
    for fn in glob.glob(os.path.join(sysfs, "class/kfd/kfd/topology/nodes/*/properties")):
        try:
            props = dict(line.split()[:2] for line in open(fn) if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) <= 0:
                continue                                   # a CPU node
            loc, dom = int(props["location_id"]), int(props.get("domain", "0"))
            idx = int(os.path.basename(os.path.dirname(fn)))
        except (OSError, ValueError, KeyError):
            continue
        nodes.append((idx, f"{dom:04x}:{(loc >> 8) & 0xff:02x}:{(loc >> 3) & 0x1f:02x}.{loc & 7:x}"))
    return [a for _, a in sorted(nodes)]


def visible_index(local_rank: int, env=None):
    """Index into the KFD GPU list of the device this rank will open as ``cuda:local_rank``, honouring
    ROCR_/HIP_/CUDA_VISIBLE_DEVICES lists of plain integers (UUID lists: unknown -> None)."""
    env = os.environ if env is None else env
    idx = local_rank
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):   # HIP's list indexes ROCR's
        v = env.get(var)
        if v is None or (var == "CUDA_VISIBLE_DEVICES" and env.get("HIP_VISIBLE_DEVICES") is not None):
            continue
        toks = [t.strip() for t in v.split(",") if t.strip() != ""]
        if idx >= len(toks) or not toks[idx].isdigit():
            return None
        idx = int(toks[idx])
    return idx


def gpu_node_cpus(local_rank: int, sysfs: str = "/sys", env=None):
    """(cpus of the GPU's NUMA node, numa node id, pci address) or (None, None, None) when sysfs does not say."""
    idx = visible_index(local_rank, env)
    gpus = kfd_gpu_pci_addresses(sysfs)
    if idx is None or idx >= len(gpus):
        return None, None, None
    dev = os.path.join(sysfs, "bus/pci/devices", gpus[idx])
    try:
        node = int(open(os.path.join(dev, "numa_node")).read().strip())
    except (OSError, ValueError):
        node = -1
    cpus = None
    try:
        if node >= 0:
            cpus = parse_cpulist(open(os.path.join(sysfs, f"devices/system/node/node{node}/cpulist")).read())
        else:
            cpus = parse_cpulist(open(os.path.join(dev, "local_cpulist")).read())
    except (OSError, ValueError):
        cpus = None
    return (cpus or None), node, gpus[idx]


def pin_rank_to_gpu_node(local_rank: int, sysfs: str = "/sys", env=None) -> dict:
    """Restrict this process to (its current affinity) & (the cores of its GPU's NUMA node).  Never widens the
    affinity, never leaves it empty; returns what it did (for the bench line)."""
    if not hasattr(os, "sched_setaffinity"):
        return {"pinned": False, "why": "no sched_setaffinity"}
    cpus, node, pci = gpu_node_cpus(local_rank, sysfs, env)
    if not cpus:
        return {"pinned": False, "why": "topology not in sysfs"}
    have = os.sched_getaffinity(0)
    want = have & cpus
    if not want:
        return {"pinned": False, "why": "node cores outside this process's cpuset", "numa_node": node, "pci": pci}
    if want != have:
        os.sched_setaffinity(0, want)
    return {"pinned": True, "numa_node": node, "pci": pci, "cpus": len(want)}
