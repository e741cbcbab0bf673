#!/usr/bin/env python3
"""bench.py -- hands/sec of the hand-mesh forward paths on N MI355X (BASELINE.json configs).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload hands_light] [--bz B]

* ``--gpus N`` with no torchrun environment (WORLD_SIZE unset): this process only LAUNCHES -- it starts N
  fresh child processes (one per GPU, RCCL = torch's "nccl" backend, rendezvous on 127.0.0.1) before
  touching the GPU itself, waits for them and passes rank 0's JSON line through.  Under
  ``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`` every process is a rank.
* Default workload ``hands_light`` = BASELINE.json configs[1], the headline metric: a step = one
  HandsLight.forward over bz=256 synthetic samples per GPU (1 global image + right crop + left crop each
  = 2 hands; 768 ResNet-50 trunk passes) followed, when N > 1, by one RCCL all-gather of the packed
  predictions.  Weak scaling: per-GPU work is fixed, value = N * 2 * bz * K / t.
  ``hamer_light`` = configs[2] (bz=64), ``handoccnet_light`` = configs[3] (bz=256 GLOBAL, 256/N per GPU),
  ``mano_lbs`` = configs[4] (1024 crops GLOBAL, 1024/N per GPU, all-gather of the 778x3 vertices):
  strong scaling.
* Inputs are resident in HBM before the timed region; outputs stay on the device.

Output (rank 0): stdout carries exactly ONE line, the headline: a JSON object below 4 KB (``compact_headline``).  At N=1 the
default run also writes one short JSON line per extra measurement to STDERR (``{"also": "<name>", ...}``: BASELINE configs
3-5 and the separately reported arithmetic modes, each with roofline + cpu_baseline + parity; the headline carries their
values as ``also: {name: hands/s}``); the per-kernel / per-launch-shape tables of everything go to
``gpurun_out/bench_details.json``.  What the keys mean is DESIGN.md section 5:
``roofline`` covers the MFMA kernels (conv_igemm_f32 family + conv_wino_f32 + the fused stem): achieved = algorithmic
FLOPs of all their launches in one step / the summed duration of those launches, each bracketed by HIP events on the
launch stream in ``mode: "serial"`` (one HIP stream, every launch alone on the chip -- the mode the rocprofv3 per-kernel
averages in profiles/ are taken in); ``roofline.step_ms_same_mode`` = ``serial.ms_per_step`` is the wall clock of that mode;
``value`` / ``ms_per_step`` are the shipped multi-stream mode; ``with_d2h_ms_per_step`` adds the predictions' copy to the
host.  ``cpu_baseline`` times the CPU oracle on the box's host cores on a bounded sample and
doubles as the parity checker (``parity``: that sample + a multi-seed sweep).
"""
import argparse
import json
import os
import re
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HANDS_SYNTHETIC_MANO", "1")   # synthetic MANO asset (data: "synthetic"); the licensed files are not here

FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA (the 5 PFLOP/s headline includes 2:1 sparsity)
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
FLOP_PER_HAND = {"hands_light": 12.77e9, "hamer_light": 251e9, "handoccnet_light": 36.2e9, "mano_lbs": 1.17e6}
GLOBAL_BATCH = {"handoccnet_light": 256, "mano_lbs": 1024}   # strong-scaling configs: fixed global batch
WORKLOADS = ("hands_light", "hamer_light", "handoccnet_light", "mano_lbs")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--bz", type=int, default=0, help="samples per GPU per step (2 hands each); default 256 "
                    "(hands_light), 64 (hamer_light), 256/N (handoccnet_light), 1024/N (mano_lbs)")
    ap.add_argument("--workload", default="hands_light", choices=WORKLOADS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the configs 3-5 measurements on the hands_light line")
    ap.add_argument("--no-sweep", action="store_true", help="skip the multi-seed parity sweep against the oracle")
    ap.add_argument("--serial", action="store_true",
                    help="time the one-stream mode as the headline too (every launch alone on the chip)")
    ap.add_argument("--latency-mode", action="store_true",
                    help="small-batch serving mode: split-K on every launch with <= 128 output tiles")
    ap.add_argument("--graph", type=int, default=-1, metavar="DEPTH",
                    help="replay the forward as a hipGraph (hands_amd.GraphedForward); DEPTH > 1 keeps that many captured instances "
                         "in flight (handoccnet_light only).  Default: 4 for --workload handoccnet_light at N > 1 (its fastest "
                         "bit-identical mode at the 8-GPU shard size; config.timed_mode says so), eager otherwise; 0 forces eager")
    ap.add_argument("--layer-report", default="", help="write a per-launch CSV of the MFMA kernels here")
    ap.add_argument("--no-pmc", action="store_true",
                    help="do not take the same-run rocprofv3 --pmc passes (two child runs of two one-stream forwards each BEFORE the first GPU call "
                         "of this process: 30-90 s per workload, up to 2 x 150 s); roofline.traffic then comes from a stored summary "
                         "whose csrc_sha16 equals the loaded library's, or is null")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)   # the profiled child: two one-stream forwards
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------------
# launcher: N fresh rank processes, started BEFORE this process makes any GPU call
# ------------------------------------------------------------------------------------------------------
def visible_gpu_count():
    """GPUs the rank processes will see, found WITHOUT loading torch or touching the HIP runtime / /dev/kfd (a process that
    has opened the GPU must not be followed by exec'ing children on this pool, and the launcher has no use for a GPU):
    the *_VISIBLE_DEVICES lists, else the KFD topology in sysfs (nodes with simd_count > 0 are GPUs), else the DRM render
    nodes.  None = unknown (the ranks then find out themselves)."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([t for t in v.split(",") if t.strip() != ""])
    import glob
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if nodes:
        n = 0
        for fn in nodes:
            try:
                props = dict(line.split()[:2] for line in open(fn) if len(line.split()) >= 2)
                n += int(props.get("simd_count", "0")) > 0
            except (OSError, ValueError):
                return None
        return n
    rnd = glob.glob("/dev/dri/renderD*")
    return len(rnd) if rnd else None


def launch_ranks(args):
    """Pure launcher: never imports torch, never opens the GPU.  Starts one fresh process per rank, keeps every rank's
    stderr in a file, and on failure prints the tail of the FIRST rank that failed."""
    share = os.environ.get("HANDS_BENCH_SHARE_GPU") == "1"
    ngpu = visible_gpu_count()
    if ngpu is not None and ngpu < args.gpus and not share:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but only {ngpu} GPU(s) are visible\n")
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    import tempfile
    logdir = os.path.join(ROOT, "gpurun_out", "bench_ranks") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else \
        tempfile.mkdtemp(prefix="hands_bench_ranks_")
    os.makedirs(logdir, exist_ok=True)
    procs, logs = [], []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                   HANDS_BENCH_LAUNCHED="1")
        log = open(os.path.join(logdir, f"rank{r}.stderr"), "w+")
        logs.append(log)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL, stderr=log))
    deadline = time.time() + float(os.environ.get("HANDS_BENCH_TIMEOUT", "1500"))
    rc, first_bad = 0, None
    alive = list(procs)
    while alive:
        for p in list(alive):
            code = p.poll()
            if code is not None:
                alive.remove(p)
                if code and first_bad is None:
                    first_bad = procs.index(p)
                rc = rc or code
        if (rc or time.time() > deadline) and alive:      # one rank failed / timed out: stop the others (exact PIDs)
            for p in alive:
                p.kill()
            for p in alive:
                p.wait()
            rc = rc or 124
            break
        time.sleep(0.05)
    if rc:
        which = first_bad if first_bad is not None else 0
        why = f"rank {which} exited with code {procs[which].returncode}" if first_bad is not None else "timed out"
        sys.stderr.write(f"bench.py launcher: {why} (rank logs: {logdir})\n")
        # the first failing rank in full, then a short tail of every other rank (a hang of rank 3 shows up as rank 0's
        # watchdog message: the cause is in another log)
        for r in [which] + [r for r in range(args.gpus) if r != which]:
            logs[r].flush()
            logs[r].seek(0)
            tail = logs[r].read()[-(4000 if r == which else 600):]
            sys.stderr.write(f"--- rank {r} stderr (tail), exit code {procs[r].returncode} ---\n{tail}\n")
    for log in logs:
        log.close()
    return rc


# ------------------------------------------------------------------------------------------------------
# same-run HBM traffic: two rocprofv3 --pmc child passes, started BEFORE this process touches the GPU
# ------------------------------------------------------------------------------------------------------
SAME_RUN_PMC = {}


def csrc_tree_sha16():
    """sha256 (16 hex digits) over the kernel sources in the tree: the same bytes in the same order as csrc/Makefile hashes
    into the library (csrc_sha.inc)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "hands_amd", "csrc")
    files = sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")) + glob.glob(os.path.join(csrc, "*.cpp")),
                   key=os.path.basename) + sorted(glob.glob(os.path.join(ROOT, "include", "*.h")))
    for fn in files:
        h.update(open(fn, "rb").read())
    return h.hexdigest()[:16]


def csrc_sha16():
    """Hash of the kernel sources the LOADED library was built from (hands_csrc_sha16, embedded at build time; ADVICE r5: a hash of
    the tree would also validate a stale .so or an A/B variant loaded through $HANDS_HIP_LIB).  Loading the library makes no GPU
    call.  Falls back to the tree hash only if the library cannot be asked."""
    try:
        import ctypes
        from hands_amd import _lib
        h = ctypes.CDLL(_lib.LIB_PATH)
        h.hands_csrc_sha16.restype = ctypes.c_char_p
        return h.hands_csrc_sha16().decode()
    except (OSError, AttributeError, ImportError):
        return csrc_tree_sha16()


def pmc_sum_conv_dispatches(csv_path, counter):
    """(sum of `counter` over the conv_igemm* / conv_wino* dispatches, number of such dispatches) of a rocprofv3
    counter_collection.csv -- the same selection as tools/pmc_summary.py."""
    import csv
    tot, disp = 0.0, set()
    with open(csv_path) as fh:
        for r in csv.DictReader(fh):
            if ("conv_igemm" in r["Kernel_Name"] or "conv_wino" in r["Kernel_Name"]) and r["Counter_Name"] == counter:
                tot += float(r["Counter_Value"])
                disp.add(r["Dispatch_Id"])
    return tot, len(disp)


def same_run_pmc(args, workload, bz):
    """HBM bytes of the GEMM / convolution launches measured IN THIS RUN: `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`
    (separate passes, no trace domains) over a fresh child `python3 bench.py --pmc-child` that runs two one-stream forwards
    (the mode `roofline` is defined in).  The children are started and finished before this process makes its first GPU
    call (a process that has opened the GPU must not be followed by an exec on this pool).  Bytes = 2 x FETCH_SIZE + WRITE_SIZE
    (KB; MI355X_MICROARCH.md, HBM section: gfx950 reports half of a wide read stream), per launch.  Returns a dict or None."""
    import glob
    import shutil
    import tempfile
    rocprof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if rocprof is None:
        return {"error": "rocprofv3 not found"}
    t0 = time.time()
    tmp = tempfile.mkdtemp(prefix="hands_pmc_", dir="/tmp" if os.path.isdir("/tmp") else None)
    env = dict(os.environ, TMPDIR=tmp, HANDS_BENCH_PMC_CHILD="1")
    sums = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [rocprof, "--pmc", counter, "--output-format", "csv", "-d", out, "-o", "pmc", "--", sys.executable,
                   os.path.abspath(__file__), "--pmc-child", "--workload", workload, "--bz", str(bz)]
            try:
                p = subprocess.run(cmd, env=env, cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE,
                                   timeout=float(os.environ.get("HANDS_PMC_TIMEOUT", "150")))
            except subprocess.TimeoutExpired:
                return {"error": f"{counter} pass timed out"}
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if p.returncode != 0 or not files:
                return {"error": f"{counter} pass rc {p.returncode}: " + p.stderr.decode(errors="replace")[-200:]}
            sums[counter] = pmc_sum_conv_dispatches(files[0], counter)
        (f_kb, nf), (w_kb, nw) = sums["FETCH_SIZE"], sums["WRITE_SIZE"]
        if nf == 0 or nf != nw:
            return {"error": f"dispatch counts differ: {nf} vs {nw}"}
        return {"gb_per_launch": (2.0 * f_kb + w_kb) * 1024 / nf / 1e9, "read_gb_per_launch": 2.0 * f_kb * 1024 / nf / 1e9,
                "write_gb_per_launch": w_kb * 1024 / nf / 1e9, "dispatches": nf, "bz": bz, "workload": workload,
                "seconds": round(time.time() - t0, 1),
                "source": "same-run: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE child passes, 2 one-stream forwards each"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def pmc_child(args):
    """The process the counter passes profile: the model of the headline, two forwards in the one-stream mode."""
    import torch
    import hands_amd
    wl = args.workload
    ctor = {"hamer_light": hands_amd.HAMER, "handoccnet_light": hands_amd.HandOccNet}.get(wl, hands_amd.HandsLight)
    model = hands_amd.apply_recipe(ctor()).to("cuda").eval()
    model.overlap_trunks = False
    inputs, meta = hands_amd.synthetic_inputs(args.bz or default_bz(wl, 1), seed=0, device="cuda")
    for _ in range(int(os.environ.get("HANDS_PMC_CHILD_FORWARDS", "2"))):     # (tools/launches_per_forward.sh diffs two counts)
        dict(model(inputs, meta).items())
        torch.cuda.synchronize()


# ------------------------------------------------------------------------------------------------------
# helpers (rank processes only below this line)
# ------------------------------------------------------------------------------------------------------
CPU_PIN = {}


def pin_to_gpu_numa_node():
    """N > 1: each rank restricts itself to the cores of its GPU's NUMA node (sysfs only, os.sched_setaffinity) BEFORE its
    first GPU call.  N = 1 keeps every core (the cpu_baseline leg uses them)."""
    if int(os.environ.get("WORLD_SIZE", "1")) <= 1 or os.environ.get("HANDS_BENCH_SHARE_GPU") == "1":
        CPU_PIN["note"] = "unpinned (single rank)"
        return
    from hands_amd.affinity import pin_rank_to_gpu_node
    r = pin_rank_to_gpu_node(int(os.environ.get("LOCAL_RANK", "0")))
    CPU_PIN.update(r)
    CPU_PIN["note"] = (f"numa{r.get('numa_node')}:{r.get('cpus')}cpus" if r.get("pinned") else f"unpinned ({r.get('why')})")


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


PMC_LEGACY_BZ = {"hands_light": 256, "hamer_light": 64, "handoccnet_light": 256}   # r01 / r02 summaries carry no "bz"


def pmc_traffic_per_launch(workload, bz):
    """(GB per conv_igemm launch, source file) from the newest committed rocprofv3 --pmc summary TAKEN AT THIS BATCH SIZE,
    or (None, None): a per-launch traffic figure of another batch size must never be divided by this run's algorithmic
    bytes (round 2 printed 8.68x for handoccnet_light that way)."""
    for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
        for fn in (os.path.join(ROOT, "profiles", f"{rnd}_pmc_{workload}_bz{bz}.json"),
                   os.path.join(ROOT, "profiles", f"{rnd}_pmc_{workload}.json")):
            try:
                d = json.load(open(fn))
                if int(d.get("bz", PMC_LEGACY_BZ.get(workload, -1))) != int(bz):
                    continue
                # a stored figure describes the binary it was taken with: summaries carry the hash of the kernel sources
                # (tools/pmc_summary.py); one without it, or of other sources, is reported as stale and NOT used
                if d.get("csrc_sha16") != csrc_sha16():
                    return None, "stale (kernel sources changed since): " + os.path.relpath(fn, ROOT)
                return round(d["hbm_gb_per_launch"], 4), "stored: " + os.path.relpath(fn, ROOT)
            except (OSError, ValueError, KeyError):
                continue
    return None, None


def pmc_shipped_gb_per_step(workload, bz):
    """(GB per forward over the conv launches of the SHIPPED mode, source file) from the newest committed shipped-mode counter
    summary taken at this batch size (tools/profile_round.sh: HANDS_BENCH_SHIPPED_ONLY passes), or (None, None)."""
    for rnd in ("r06", "r05", "r04"):
        fn = os.path.join(ROOT, "profiles", f"{rnd}_pmc_{workload}_bz{bz}_shipped.json")
        try:
            d = json.load(open(fn))
            if int(d.get("bz", -1)) == int(bz) and d.get("gb_per_forward"):
                if d.get("csrc_sha16") != csrc_sha16():        # taken with other kernel sources: not this binary's figure
                    return None, "stale (kernel sources changed since): " + os.path.relpath(fn, ROOT)
                return round(d["gb_per_forward"], 3), "stored: " + os.path.relpath(fn, ROOT)
        except (OSError, ValueError, KeyError):
            continue
    return None, None


class Ctx:
    """Rank context: torch.distributed state + device."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        assert self.world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={self.world}"
        # HANDS_BENCH_SHARE_GPU=1 + HANDS_BENCH_BACKEND=gloo: dry-run of the N>1 code path on a 1-GPU box
        # (every rank on cuda:0, gloo collectives on host copies); never set by the driver.
        share = os.environ.get("HANDS_BENCH_SHARE_GPU") == "1"
        self.backend = os.environ.get("HANDS_BENCH_BACKEND", "nccl")
        idx = 0 if share else local_rank
        torch.cuda.set_device(idx)
        self.dev = torch.device("cuda", idx)
        if self.world > 1:
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=self.dev)
            else:
                dist.init_process_group(self.backend)
        self.rccl_ranks = dist.get_world_size() if self.world > 1 else 1
        assert self.rccl_ranks == args.gpus, f"--gpus {args.gpus} but the process group has {self.rccl_ranks} ranks"
        # gloo dry runs gather host copies unless HANDS_BENCH_GLOO_DEVICE=1 (gloo stages device tensors itself:
        # that exercises the same stream-ordered gather path RCCL takes)
        self.host_collective = (self.world > 1 and self.backend != "nccl" and
                                os.environ.get("HANDS_BENCH_GLOO_DEVICE") != "1")

    def fence(self):
        self.torch.cuda.synchronize(self.dev)
        if self.world > 1:
            self.dist.barrier()
            self.torch.cuda.synchronize(self.dev)

    def max_over_ranks(self, seconds):
        if self.world == 1:
            return seconds
        t = self.torch.tensor([seconds], dtype=self.torch.float64, device="cpu" if self.backend != "nccl" else self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather_floats(self, x):
        """[x of rank 0, x of rank 1, ...] on every rank."""
        if self.world == 1:
            return [float(x)]
        t = self.torch.tensor([float(x)], dtype=self.torch.float64, device="cpu" if self.backend != "nccl" else self.dev)
        parts = [self.torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(parts, t)
        return [float(p.item()) for p in parts]

    def timed(self, step, steps, warmup):
        """W untimed steps, then exactly K steps between barrier + synchronize pairs; max over ranks."""
        for _ in range(warmup):
            step()
        self.fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        self.fence()
        self.last_local_seconds = time.perf_counter() - t0      # this rank's own clock (per-rank rates)
        return self.max_over_ranks(self.last_local_seconds)

    def close(self):
        if self.world > 1:
            self.dist.destroy_process_group()


CEILINGS = {}


def measure_ceilings(ctx):
    """The chip's own ceilings on THIS box (report against the datasheet AND the measured figures): the bare
    fp32-MFMA loop of csrc/ceilings.hip (constant operands = the matrix pipe's ceiling; full-mantissa operands = what it clocks at on
    real data) and a 2 GiB streaming read, each the best of three launches timed with events on the launch stream."""
    if CEILINGS:
        return CEILINGS
    torch = ctx.torch
    from hands_amd import _lib
    L = _lib.lib()
    st = torch.cuda.current_stream(ctx.dev)
    out = torch.empty(512 * 256, device=ctx.dev)

    def best_of(launch, n=3):
        best = 0.0
        for i in range(n + 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            work = launch()
            e1.record(st)
            e1.synchronize()
            if i:                                  # the first launch warms the clocks
                best = max(best, work / (e0.elapsed_time(e1) * 1e-3))
        return best

    def mfma(rnd):
        def go():
            fl = L.hands_ceiling_mfma_f32(_lib.ptr(out), out.numel(), 3000, rnd, st.cuda_stream)
            if fl <= 0:
                raise RuntimeError(f"hands_ceiling_mfma_f32 failed: {fl}")
            return float(fl)
        return go

    CEILINGS["mfma_f32_tflops"] = round(best_of(mfma(0)) / 1e12, 1)
    CEILINGS["mfma_f32_random_operands_tflops"] = round(best_of(mfma(1)) / 1e12, 1)
    buf = torch.empty((2 << 30) // 4, device=ctx.dev)
    buf.fill_(1.0)

    def read():
        _lib.check(L.hands_ceiling_hbm_read_f32(_lib.ptr(buf), buf.numel(), _lib.ptr(out), st.cuda_stream), "hands_ceiling_hbm_read_f32")
        return float(buf.numel() * 4)

    CEILINGS["hbm_read_tbs"] = round(best_of(read) / 1e12, 2)
    del buf
    torch.cuda.empty_cache()
    CEILINGS["source"] = "same-run: hands_ceiling_mfma_f32 (3000 x 32 MFMA per wave, 2 waves / SIMD) and hands_ceiling_hbm_read_f32 (2 GiB), best of 3"
    return CEILINGS


def default_bz(workload, world):
    if workload in GLOBAL_BATCH:
        return max(1, GLOBAL_BATCH[workload] // world)
    return {"hamer_light": 64}.get(workload, 256)


def workload_text(workload, bz, world):
    if workload == "hamer_light":
        return f"hamer_light ViT-H/16 (256x192) + decoder head + MANO, bz={bz} samples/GPU ({2 * bz} crops = hands)"
    if workload == "handoccnet_light":
        return ("handoccnet_light LeakyReLU ResNet-50 + FPN + FIT/SET + hourglass regressor + MANO, 256x256, "
                f"bz={bz} samples/GPU ({2 * bz} crops = hands), global batch {bz * world}")
    if workload == "mano_lbs":
        return (f"two-hand MANO LBS (MANOHead.forward x2), {bz} crops/GPU ({2 * bz} hands), global {bz * world} crops"
                + (", all-gather of 778x3 vertices" if world > 1 else ""))
    return f"hands_light ResNet-50 x3 + feature_conv + HMR + MANO, 224x224, bz={bz} samples/GPU ({2 * bz} hands, {3 * bz} trunk passes)"


# ------------------------------------------------------------------------------------------------------
# model workloads (hands_light / hamer_light / handoccnet_light)
# ------------------------------------------------------------------------------------------------------
def host_enqueue_ms(ctx, step, n=3):
    """Host time to ENQUEUE one step (no synchronisation inside): what the CPU side of a rank costs per step."""
    torch = ctx.torch
    torch.cuda.synchronize(ctx.dev)
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        step()
        ts.append(time.perf_counter() - t0)
        torch.cuda.synchronize(ctx.dev)
    if ctx.world > 1:
        ctx.dist.barrier()
    return round(sorted(ts)[len(ts) // 2] * 1e3, 3)


def allgather_us(ctx, make_out, n=5):
    """Device time of the prediction all-gather alone (pack + one RCCL collective + unpack views), HIP events on the
    stream it runs on: the forward is joined first so the events bracket only the collective."""
    torch = ctx.torch
    from hands_amd.dist import gather_predictions
    from hands_amd.xdict import xdict
    out = xdict(dict(make_out().items()))       # joined, plain dict: the gather runs on the current stream
    torch.cuda.synchronize(ctx.dev)
    st = torch.cuda.current_stream(ctx.dev)
    ts = []
    for _ in range(n):
        ctx.dist.barrier()
        torch.cuda.synchronize(ctx.dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        gather_predictions(out)
        e1.record(st)
        torch.cuda.synchronize(ctx.dev)
        ts.append(e0.elapsed_time(e1) * 1e3)
    return round(sorted(ts)[len(ts) // 2], 1)


def forward_with_d2h_ms(ctx, model, inputs, meta, n):
    """BASELINE.md section 5's second figure: forward + the predictions on the host (the reference's eval loop moves the
    xdict to the CPU after inference_pose, src/models/generic/wrapper.py:68-75): one packed (bz, D) buffer, one D2H copy
    into pinned memory, synchronised every step as a caller that consumes the result would."""
    torch = ctx.torch
    from hands_amd.dist import pack_predictions
    host = [None]

    def step():
        flat, _ = pack_predictions(model(inputs, meta))
        if host[0] is None:
            host[0] = torch.empty(flat.shape, dtype=flat.dtype, pin_memory=True)
        host[0].copy_(flat, non_blocking=True)
        torch.cuda.current_stream(ctx.dev).synchronize()

    step()
    torch.cuda.synchronize(ctx.dev)
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    return round((time.perf_counter() - t0) / n * 1e3, 3)


def measure_model(ctx, workload, bz, steps, warmup, args, serial_headline=False, parity_bz=0, layer_report="", math="fp32",
                  winograd=None, winograd_scope=None, graph=0):
    """Returns (result dict, model, cpu state_dict) -- result holds value / ms_per_step / roofline / serial /
    overlapped (+ parity vs the oracle on ``parity_bz`` samples when > 0, rank 0 only)."""
    torch = ctx.torch
    import hands_amd
    from hands_amd.dist import gather_predictions
    ctor = {"hamer_light": hands_amd.HAMER, "handoccnet_light": hands_amd.HandOccNet}.get(workload, hands_amd.HandsLight)
    model = hands_amd.apply_recipe(ctor())
    sd_cpu = {k: v.clone() for k, v in model.state_dict().items()} if (ctx.rank == 0 and parity_bz) else None
    model = model.to(ctx.dev).eval()
    model.latency_mode = bool(args.latency_mode)
    model.engine.math = math
    if os.environ.get("HANDS_STREAMK"):            # developer A/B switch
        model.engine.stream_k = {"1": True, "0": False}.get(os.environ["HANDS_STREAMK"], "auto")
    if os.environ.get("HANDS_MATH"):               # developer switch; the default line is always exact fp32
        model.engine.math = os.environ["HANDS_MATH"]
    if os.environ.get("HANDS_ASYNC_TAIL") and hasattr(model, "async_tail"):
        model.async_tail = os.environ["HANDS_ASYNC_TAIL"] == "1"
    if os.environ.get("HANDS_SMALL_MAP_SPLITK") and hasattr(model, "small_map_splitk"):      # developer A/B switch
        model.small_map_splitk = os.environ["HANDS_SMALL_MAP_SPLITK"] == "1"
    if os.environ.get("HANDS_FUSE_PRE"):           # developer A/B switch
        model.engine.fuse_pre = os.environ["HANDS_FUSE_PRE"] == "1"
    if winograd is not None:                       # the model's own default otherwise
        model.engine.winograd = bool(winograd)
    if winograd_scope is not None and hasattr(model, "winograd_scope"):    # HandOccNet: default "all" + chains <= 128 floats
        model.winograd_scope = winograd_scope
        if winograd_scope == "backbone":           # the comparison line: rounds 3-4's default, single fp32 chains, small-map split-K
            model.engine.chain_limit = 0
            model.small_map_splitk = True
        model.invalidate_packed()
    if os.environ.get("HANDS_WINO4_STAGES") is not None and hasattr(model, "winograd4_stages"):    # developer A/B switch: "", "4", "1234"
        model.winograd4_stages = tuple(int(c) for c in os.environ["HANDS_WINO4_STAGES"])
        model.invalidate_packed()
    if os.environ.get("HANDS_WINOGRAD"):           # developer A/B switch
        model.engine.winograd = os.environ["HANDS_WINOGRAD"] == "1"
    if os.environ.get("HANDS_WINOGRAD_SCOPE") and hasattr(model, "winograd_scope"):   # developer A/B switch: all | trunk
        model.winograd_scope = os.environ["HANDS_WINOGRAD_SCOPE"]
        model.invalidate_packed()
    if os.environ.get("HANDS_PIPE_DEPTH") and hasattr(model, "pipeline_depth"):        # developer A/B switch
        model.pipeline_depth = int(os.environ["HANDS_PIPE_DEPTH"])
    if os.environ.get("HANDS_ASYNC_FORWARD") and hasattr(model, "async_forward"):      # developer A/B switch
        model.async_forward = os.environ["HANDS_ASYNC_FORWARD"] == "1"
    model.overlap_trunks = not serial_headline
    if workload == "hands_light" and os.environ.get("HANDS_CHUNKS"):
        model.trunk_chunks = tuple(int(v) for v in os.environ["HANDS_CHUNKS"].split(","))
    inputs, meta = hands_amd.synthetic_inputs(bz, seed=ctx.rank, device=ctx.dev)
    fwd = hands_amd.GraphedForward(model, inputs, meta, depth=graph) if graph else model

    def step():
        out = fwd(inputs, meta)
        if ctx.host_collective:                   # dry-run only: gloo gathers host tensors
            torch.cuda.synchronize(ctx.dev)
            return gather_predictions({k: v.cpu() for k, v in out.items()})
        if ctx.world == 1:
            return out
        g = gather_predictions(out)
        if graph and getattr(g, "is_pending", False):
            # the gather packs the captured instance's STATIC outputs on a side stream: its next replay waits for that
            fwd.hold_until(g.__dict__["_ready"])
        return g

    elapsed = ctx.timed(step, steps, warmup)
    enqueue_ms = host_enqueue_ms(ctx, step)
    rank_rates = ctx.gather_floats(2 * bz * steps / ctx.last_local_seconds)
    # (both through `fwd`, the execution mode `value` was measured in: the hipGraph replay when --graph is set)
    gather_us = allgather_us(ctx, lambda: fwd(inputs, meta)) if ctx.world > 1 and not ctx.host_collective else None
    d2h_ms = forward_with_d2h_ms(ctx, fwd, inputs, meta, max(2, min(steps, 5))) if ctx.world == 1 else None
    if ctx.rank != 0:
        return None, model, None
    flop_per_hand = FLOP_PER_HAND[workload]
    hands_per_s = ctx.world * 2 * bz * steps / elapsed
    res = {"value": round(hands_per_s, 1), "ms_per_step": round(elapsed / steps * 1e3, 3),
           "hands_per_sec_per_gpu": round(hands_per_s / ctx.world, 1), "host_enqueue_ms_per_step": enqueue_ms,
           "with_d2h_ms_per_step": d2h_ms}
    if ctx.world > 1:
        # each rank's hands over ITS OWN clock between the two fences (value uses the max-over-ranks time)
        res["per_rank_hands_per_sec"] = {"min": round(min(rank_rates), 1), "max": round(max(rank_rates), 1)}
        res["allgather_us"] = gather_us
    path_tf = hands_per_s * flop_per_hand / 1e12 / ctx.world
    res["overlapped" if not serial_headline else "serial"] = {
        "mode": "serial" if serial_headline else "multi-stream",
        "ms_per_step": res["ms_per_step"], "path_tflops": round(path_tf, 2),
        "path_frac_of_fp32_mfma_peak": round(path_tf / FP32_MFMA_PEAK_TFLOPS, 4)}

    if os.environ.get("HANDS_BENCH_SHIPPED_ONLY") == "1":
        # developer switch for counter passes (tools/pmc_positions.sh): only the shipped-mode forwards run in this process, so
        # every dispatch rocprofv3 sees is a launch of the mode `value` is measured in
        res["roofline"] = None
        return res, model, sd_cpu
    # ---- the same model in one-stream mode: wall clock + every MFMA launch bracketed by events -------
    model.overlap_trunks = False
    n_ser = max(2, min(steps, 5))
    for _ in range(1):
        model(inputs, meta)
    torch.cuda.synchronize(ctx.dev)
    t0 = time.perf_counter()
    for _ in range(n_ser):
        model(inputs, meta)
    torch.cuda.synchronize(ctx.dev)
    ser_ms = (time.perf_counter() - t0) / n_ser * 1e3
    if not serial_headline:
        ser_tf = 2 * bz / (ser_ms * 1e-3) * flop_per_hand / 1e12
        res["serial"] = {"mode": "serial", "ms_per_step": round(ser_ms, 3),
                         "hands_per_sec": round(2 * bz / (ser_ms * 1e-3), 1), "path_tflops": round(ser_tf, 2),
                         "path_frac_of_fp32_mfma_peak": round(ser_tf / FP32_MFMA_PEAK_TFLOPS, 4), "steps": n_ser}

    main_stream = torch.cuda.current_stream(ctx.dev)
    events, info = [], []

    def hook(phase, pc, npix, stream_handle, has_res, kernel):
        # one-stream mode: the stream handed to the C ABI is torch's current stream, so the two events
        # bracket exactly this launch
        assert stream_handle == main_stream.cuda_stream
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(main_stream)
        events.append(ev)
        if phase == "begin":
            macs = getattr(pc, "group_macs", None) or pc.macs_per_pixel * npix      # a grouped launch carries its members' sums
            # algorithmic bytes: input read once + output written once (+ residual read) + weights once
            if kernel.startswith("stem_pool"):    # image in (RGB0 NHWC4, or the 3 NCHW planes), pooled 64-channel map out
                nbytes = 4.0 * (npix * 4 * (3 if "planar" in kernel else 4) + (npix // 4) * 64 + pc.w.numel())
            elif hasattr(pc, "group_bytes"):
                nbytes = pc.group_bytes
            else:
                nbytes = 4.0 * (npix * pc.Cout * (2 if has_res else 1) + npix * pc.stride * pc.stride * pc.Cin + pc.w.numel())
            # Winograd launches execute 16 / 36 of the layer's algorithmic multiplications (+ idle tile lanes): keep both counts
            xmacs = model.engine.last_wino_macs if kernel.startswith("conv_wino") else macs
            info.append((kernel, pc.Cin, pc.Cout, pc.KH, pc.stride, npix, macs, nbytes, xmacs))

    n_prof = 3
    model.conv_hook = hook
    for _ in range(n_prof):
        model(inputs, meta)
    torch.cuda.synchronize(ctx.dev)
    model.conv_hook = None
    model.overlap_trunks = not serial_headline
    # an event pair with nothing between its two records still reads a few microseconds: calibrate that
    # bracket overhead on the same stream and take it off every launch (rocprofv3's per-kernel averages in
    # profiles/ are the cross-check)
    cal = [torch.cuda.Event(enable_timing=True) for _ in range(202)]
    for ev in cal:
        ev.record(main_stream)
    torch.cuda.synchronize(ctx.dev)
    gaps = sorted(cal[i].elapsed_time(cal[i + 1]) for i in range(0, 202, 2))
    ev_overhead_ms = gaps[len(gaps) // 2]
    durs_ms = [max(events[i].elapsed_time(events[i + 1]) - ev_overhead_ms, 0.0) for i in range(0, len(events), 2)]
    launches = len(durs_ms) // n_prof
    # per launch the MEDIAN of the n_prof instrumented forwards (the launch sequence of a forward is deterministic): a box now
    # and then stretches one launch of one pass from 0.5 to 4 ms (seen in A/B runs of every variant), which a mean would keep
    if all(info[i][0] == info[i + r * launches][0] for i in range(launches) for r in range(n_prof)):
        med = [sorted(durs_ms[i + r * launches] for r in range(n_prof))[n_prof // 2] for i in range(launches)]
        durs_ms = med * n_prof
    per = {}
    for i, ms in enumerate(durs_ms):
        k = info[i][0]
        d = per.setdefault(k, {"launches": 0, "ms": 0.0, "flop": 0.0, "bytes": 0.0, "xflop": 0.0})
        d["launches"] += 1
        d["ms"] += ms
        d["flop"] += 2.0 * info[i][6]
        d["bytes"] += info[i][7]
        d["xflop"] += 2.0 * info[i][8]
    kernels = {k: {"launches_per_step": d["launches"] // n_prof, "ms_per_step": round(d["ms"] / n_prof, 3),
                   "avg_launch_us": round(d["ms"] * 1e3 / d["launches"], 2),
                   "tflops": round(d["flop"] / (d["ms"] * 1e-3) / 1e12, 2),
                   "algorithmic_gb_per_launch": round(d["bytes"] / d["launches"] / 1e9, 4)} for k, d in per.items()}
    for k, d in per.items():
        if d["xflop"] != d["flop"]:      # Winograd F(2x2,3x3): fewer executed than algorithmic FLOPs
            kernels[k]["executed_tflops"] = round(d["xflop"] / (d["ms"] * 1e-3) / 1e12, 2)
            kernels[k]["executed_over_algorithmic"] = round(d["xflop"] / d["flop"], 4)
    x_flop = sum(d["xflop"] for d in per.values()) / n_prof
    k_ms = sum(d["ms"] for d in per.values()) / n_prof
    k_flop = sum(d["flop"] for d in per.values()) / n_prof
    achieved = k_flop / (k_ms * 1e-3) / 1e12
    # the GEMM / convolution family (plain, stream-K and split-K launches of conv_igemm): what the PMC summaries cover
    fam = [d for k, d in per.items() if k.startswith(("conv_igemm", "conv_wino"))]
    fam_launches = sum(d["launches"] for d in fam)
    fam_alg_gb = sum(d["bytes"] for d in fam) / max(fam_launches, 1) / 1e9
    igemm = [d for k, d in per.items() if k.startswith("conv_igemm")]
    ig_ms = sum(d["ms"] for d in igemm) / n_prof
    ig_flop = sum(d["flop"] for d in igemm) / n_prof
    sr = SAME_RUN_PMC.get((workload, bz))
    if sr and "gb_per_launch" in sr:
        traffic, traffic_src = round(sr["gb_per_launch"], 4), sr["source"]
    else:
        traffic, traffic_src = pmc_traffic_per_launch(workload, bz)
        if sr and "error" in sr:
            traffic_src = f"{traffic_src} (same-run pass failed: {sr['error'][:80]})"
    ship_gb, ship_src = pmc_shipped_gb_per_step(workload, bz)
    fam_alg_gb_step = sum(d["bytes"] for d in fam) / n_prof / 1e9
    if math == "bf16x3":
        # this mode runs six v_mfma_f32_32x32x16_bf16 per k-16 step where the exact path runs eight fp32 MFMAs: its
        # ceiling in fp32-EQUIVALENT FLOPs is the dense bf16 peak / 6, not the fp32-MFMA peak
        peak, peak_note = BF16_MFMA_PEAK_TFLOPS / 6.0, "bf16_dense_peak/6"
    else:
        peak, peak_note = FP32_MFMA_PEAK_TFLOPS, "fp32_mfma_peak"
    # every launch priced on the roof that bounds IT: time at the MFMA peak for its algorithmic FLOPs
    # against time at 8 TB/s for its algorithmic bytes; a launch is "hbm" when the second is the longer one
    shapes, by_bound = {}, {"mfma": [0, 0.0, 0.0], "hbm": [0, 0.0, 0.0]}
    for i in range(launches):
        kern, cin, cout, kk, st, npix, mc, nb, _x = info[i]
        ms = durs_ms[i]
        t_m, t_h = 2.0 * mc / (peak * 1e12), nb / (HBM_PEAK_GBS * 1e9)
        bound = "hbm" if t_h > t_m else "mfma"
        b = by_bound[bound]
        b[0] += 1
        b[1] += ms
        b[2] += max(t_m, t_h) * 1e3
        sh = shapes.setdefault((kern, cin, cout, kk, st, npix), {"n": 0, "ms": 0.0, "flop": 2.0 * mc, "bytes": nb, "bound": bound})
        sh["n"] += 1
        sh["ms"] += ms
    shape_rows = [{"kernel": k[0], "Cin": k[1], "Cout": k[2], "k": k[3], "stride": k[4], "M": k[5], "launches": v["n"],
                   "us_per_launch": round(v["ms"] / v["n"] * 1e3, 1), "bound": v["bound"],
                   "tflops": round(v["flop"] * v["n"] / (v["ms"] * 1e-3) / 1e12, 1),
                   "alg_tbs": round(v["bytes"] * v["n"] / (v["ms"] * 1e-3) / 1e12, 2),
                   "frac": round(max(v["flop"] / (peak * 1e12), v["bytes"] / (HBM_PEAK_GBS * 1e9)) * v["n"] / (v["ms"] * 1e-3), 3)}
                  for k, v in shapes.items()]
    sk = model.engine.stream_k
    ceil = measure_ceilings(ctx) if math == "fp32" else {}
    peak_m, hbm_m = ceil.get("mfma_f32_tflops"), ceil.get("hbm_read_tbs")
    own_roof_m = None
    if peak_m and hbm_m and k_ms:      # every launch on ITS roof again, priced with the measured ceilings instead of the datasheet's
        own_roof_m = round(sum(max(2.0 * info[i][6] / (peak_m * 1e12), info[i][7] / (hbm_m * 1e12)) * 1e3 for i in range(launches)) / k_ms, 4)
    res["roofline"] = {
        "bound": "mfma", "mode": "serial", "kernel": "+".join(sorted(k.replace("_kernel", "") for k in per)),
        "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s", "peak_is": peak_note,
        "frac": round(achieved / peak, 4),
        # the same against the ceilings measured on THIS box in this run (section 8d): the bare MFMA loop, a streaming read
        "peak_measured": peak_m, "frac_measured": round(achieved / peak_m, 4) if peak_m else None,
        "peak_measured_random_operands": ceil.get("mfma_f32_random_operands_tflops"), "hbm_measured_tbs": hbm_m,
        "frac_on_own_roof_measured": own_roof_m, "ceilings_source": ceil.get("source"),
        "executed_tflops": round(x_flop / (k_ms * 1e-3) / 1e12, 2), "executed_frac": round(x_flop / (k_ms * 1e-3) / 1e12 / peak, 4),
        "dominant": {"kernel": "conv_igemm_f32", "share_of_kernel_ms": round(ig_ms / k_ms, 3) if k_ms else None,
                     "frac": round(ig_flop / (ig_ms * 1e-3) / 1e12 / peak, 4) if ig_ms else None},
        # per-launch roof: frac_on_own_roof = sum over launches of (time at the roof that bounds the launch) / measured time
        "by_bound": {b: {"launches": v[0], "ms": round(v[1], 3), "frac_of_own_roof": round(v[2] / v[1], 4) if v[1] else None}
                     for b, v in by_bound.items()},
        "frac_on_own_roof": round(sum(v[2] for v in by_bound.values()) / k_ms, 4) if k_ms else None,
        "traffic": traffic, "traffic_unit": "GB/launch", "traffic_source": traffic_src,
        "traffic_over_algorithmic": round(traffic / fam_alg_gb, 3) if traffic and fam_alg_gb > 0 else None,
        "algorithmic_gb_per_conv_launch": round(fam_alg_gb, 4),
        # the same counters over the launches of the SHIPPED mode (plain kernels on several streams; the one-stream mode above
        # runs stream-K launches, whose sequential n-tiles re-read their activation panel): GB per forward over all conv launches
        "traffic_shipped_gb_per_step": ship_gb, "traffic_shipped_source": ship_src,
        "algorithmic_gb_per_step": round(fam_alg_gb_step, 3),
        "traffic_shipped_over_algorithmic": round(ship_gb / fam_alg_gb_step, 3) if ship_gb and fam_alg_gb_step > 0 else None,
        "stream_k": {"engine_setting": sk, "used_in_this_serial_pass": any(k == "conv_igemm_sk_f32_kernel" for k in per),
                     "launches_per_step": per.get("conv_igemm_sk_f32_kernel", {"launches": 0})["launches"] // n_prof},
        "timing": f"hip_events_per_launch_median_of_{n_prof}",
        "launches_per_step": launches, "kernel_ms_per_step": round(k_ms, 3),
        "step_ms_same_mode": round(ser_ms, 3), "event_bracket_overhead_us": round(ev_overhead_ms * 1e3, 2),
        "algorithmic_gflop_per_sample": round(k_flop / bz / 1e9, 3), "kernels": kernels, "shapes": shape_rows}
    if math == "bf16x3":
        for blk in ("overlapped", "serial"):
            if blk in res:
                res[blk]["path_frac_of_bf16x3_ceiling"] = round(res[blk]["path_tflops"] / peak, 4)
                res[blk].pop("path_frac_of_fp32_mfma_peak", None)
    if layer_report:
        with open(layer_report, "w") as fh:
            fh.write("idx,kernel,Cin,Cout,k,stride,M,gflop,alg_mb,ms,tflops,alg_tbs,bound,frac_of_own_roof\n")
            for i in range(launches):
                ms = sum(durs_ms[i + r * launches] for r in range(n_prof)) / n_prof
                kern, cin, cout, k, st, npix, mc, nb, _x = info[i]
                t_m, t_h = 2.0 * mc / (peak * 1e12), nb / (HBM_PEAK_GBS * 1e9)
                fh.write(f"{i},{kern},{cin},{cout},{k},{st},{npix},{2 * mc / 1e9:.3f},{nb / 1e6:.2f},{ms:.4f},{2 * mc / ms / 1e9:.2f},"
                         f"{nb / ms / 1e9:.3f},{'hbm' if t_h > t_m else 'mfma'},{max(t_m, t_h) * 1e3 / ms:.3f}\n")
    return res, model, sd_cpu


def oracle_forward_fn(workload):
    from oracle import hands_oracle as O
    if workload == "hamer_light":
        from oracle import hamer_oracle as HO
        return HO.hamer_forward
    if workload == "handoccnet_light":
        from oracle import handoccnet_oracle as HOC
        return HOC.handoccnet_forward
    return O.hands_light_forward


ORACLE_THREADS = 8      # the checker's ATen thread count: the one the golden fixtures were generated with (tests/conftest.py).  ATen's
                        # blocked sums depend on it -- handoccnet_light's REFERENCE vertices move by 2-5e-7 m between 1, 8 and 16 threads
                        #  -- so every parity figure names the count; cpu_baseline's TIMING uses all host cores


def oracle_at(ctx, threads, fn):
    """fn() with ATen at `threads` threads (restored afterwards)."""
    torch = ctx.torch
    before = torch.get_num_threads()
    torch.set_num_threads(max(1, int(threads)))
    try:
        return fn()
    finally:
        torch.set_num_threads(before)


def parity_vs_oracle(ctx, workload, model, sd_cpu, cb, ref=None, sample=None, threads=ORACLE_THREADS):
    """MPJPE (root-aligned, mm) and max vertex error of the HIP path against the oracle on ``cb`` samples; the oracle runs at
    ``threads`` ATen threads unless ``ref`` (its output, computed by the caller at that count) is given."""
    import hands_amd
    from oracle import hands_oracle as O
    torch = ctx.torch
    ci, cm = sample if sample is not None else hands_amd.synthetic_inputs(cb, seed=0)
    if ref is None:
        ar, al = hands_amd.synthetic_mano_asset(True), hands_amd.synthetic_mano_asset(False)
        ref = oracle_at(ctx, threads, lambda: oracle_forward_fn(workload)(sd_cpu, ar, al, ci, cm))
    got = model({k: v.to(ctx.dev) for k, v in ci.items()}, {k: v.to(ctx.dev) for k, v in cm.items()})
    torch.cuda.synchronize(ctx.dev)
    verr = max((got[f"mano.vertices.{h}"].cpu() - ref[f"mano.vertices.{h}"]).abs().max().item() for h in "rl")
    mp = max(O.mpjpe_ra_mm(got[f"mano.joints3d.{h}"].cpu(), ref[f"mano.joints3d.{h}"]) for h in "rl")
    return {"mpjpe_vs_ref_mm": round(mp, 7), "max_vertex_err_m": float(f"{verr:.3e}"), "checked_hands": 2 * cb,
            "checker": "oracle", "oracle_threads": int(threads)}


HON_STAGES = ("resnet", "fpn", "fit", "set", "hourglass", "reghead", "encoder", "mlp")
SWEEP = {"hands_light": (2, range(1, 9)), "handoccnet_light": (2, range(1, 9)), "hamer_light": (1, range(1, 4))}


def parity_sweep(ctx, workload, model, sd_cpu):
    """Worst max-vertex error of the HIP path against the oracle over several input seeds, LIVE, at ORACLE_THREADS threads; for
    handoccnet_light (whose reference moves with ATen's thread count) the same seeds also against the oracle at 1 thread and at
    every host core -- the bar has to hold however the reference is run."""
    import hands_amd
    cb, seeds = SWEEP[workload]
    counts = [ORACLE_THREADS]
    if workload == "handoccnet_light":
        counts += [t for t in (1, host_cores()) if t not in counts]
    worst = {t: (0.0, None) for t in counts}
    for seed in seeds:
        sample = hands_amd.synthetic_inputs(cb, seed=seed)
        for t in counts:
            e = parity_vs_oracle(ctx, workload, model, sd_cpu, cb, sample=sample, threads=t)["max_vertex_err_m"]
            if e >= worst[t][0]:
                worst[t] = (e, seed)
    w, ws = worst[ORACLE_THREADS]
    res = {"worst_vertex_err_m": float(f"{w:.3e}"), "worst_seed": ws, "sweep_seeds": len(seeds), "bar_m": 1e-6,
           "oracle_threads": ORACLE_THREADS,
           "live_sweep_inside_bar": bool(max(v[0] for v in worst.values()) <= 1e-6)}
    if len(counts) > 1:
        res["worst_by_oracle_threads"] = {str(t): float(f"{worst[t][0]:.3e}") for t in counts}
    res.update(stored_exceed_rate(workload, model))
    if not res["live_sweep_inside_bar"] and "exceed_source" in res:
        res["exceed_source"] += " -- NOTE: this run's LIVE sweep has an input above the bar, the stored rate does not describe it"
    return res


def stored_exceed_rate(workload, model):
    """handoccnet_light: the share of random inputs whose max vertex error against the reference exceeds 1e-6 m, from the committed
    1000-seed A/B of the setting this model runs (profiles/r05_hon_parity_ab_1000seeds*_summary.json, tools/hon_parity_ab.py) -- a
    STORED figure (the run takes 2 minutes of CPU forwards), named as such; absent for any other setting.  hands_light: the same from
    profiles/r05_hl_parity_ab_1000seeds_summary.json (tools/hl_parity_ab.py), arm = the 3x3 route the model runs."""
    def pick(files, arm):
        for fn in files:
            try:
                d = json.load(open(os.path.join(ROOT, "profiles", fn)))
                a = d["arms"].get(arm)
                if a:
                    return {"exceed_rate": round(a["exceed_rate"], 4), "exceed_wilson95": [round(v, 4) for v in a["wilson95"]],
                            "exceed_n": d["n"], "median_err_ratio_vs_fp64": round(a["median_ratio_hip64_over_ref64"], 2),
                            "exceed_source": f"stored: profiles/{fn} arm {arm}"}
            except (OSError, ValueError, KeyError):
                continue
        return {}

    if workload == "hands_light" and hasattr(model, "winograd4_stages") and getattr(model.engine, "math", "fp32") == "fp32":
        if not model.engine.winograd:
            arm = "direct"
        elif model.engine.winograd4 and tuple(model.winograd4_stages) == (1, 2, 3, 4):
            arm = "f4x4"
        elif not (model.engine.winograd4 and model.winograd4_stages):
            arm = "f2x2"
        else:
            return {}
        return pick(("r05_hl_parity_ab_1000seeds_summary.json",), arm)
    if workload == "hamer_light" and getattr(model.engine, "math", "fp32") == "fp32":
        return pick(("r05_hm_parity_ab_400seeds_summary.json",), "default")
    if workload != "handoccnet_light" or not hasattr(model, "winograd_scope"):
        return {}
    e = model.engine
    arm = model.winograd_scope if e.winograd else "direct"
    if e.chain_limit:
        arm += f"+c{e.chain_limit}"
        if e.chain_min_k and e.chain_min_k != 2 * e.chain_limit:
            arm += f"k{e.chain_min_k}"
        if getattr(e, "chain_in_kernel", False):
            arm += "i"
    f64 = getattr(model, "acc64_stages", frozenset()) if getattr(e, "acc64", False) else frozenset()
    if f64 or getattr(model, "wino_stages", None) is not None or getattr(model, "block_stages", None) != frozenset(HON_STAGES):
        if getattr(model, "wino_stages", None) is not None or model.block_stages != frozenset(HON_STAGES):
            return {}                                        # a per-stage plan nobody ran the statistics for
        arm += "+f:" + ".".join(s_ for s_ in ("reghead", "encoder", "mlp", "fit", "set", "fpn", "resnet", "hourglass") if s_ in f64)
        res = pick(("r06_hon_parity_ab_1000seeds_summary.json",), arm)
        if res and arm == "all+c64i+f:reghead.mlp":
            res.update(live_thread_counts())
        return res
    if not getattr(model, "small_map_splitk", False):       # the runs before `_e_` had the small-map split-K rules on
        return {}                                            # round-5 figures describe round-5 kernels (attention, softmax changed since)
    return {}


def live_thread_counts():
    """handoccnet_light, shipped default: inputs above 1e-6 m against the oracle run LIVE on a GPU box's host at 8 and at 1 ATen
    threads, and the oracle at 8 threads against ITSELF at 1 (tools/hon_live_threads.py, seeds 3000-3249 and 4000-4999; STORED:
    profiles/r06_hon_live_threads_*_summary.json).  The reference is not inside the bar against itself on every input."""
    k = {"8": 0, "1": 0, "n": 0}
    self_max = 0.0
    for fn in ("r06_hon_live_threads_250seeds_summary.json", "r06_hon_live_threads_1000seeds_summary.json"):
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", fn)))
            k["8"] += d["hip_vs_t8"]["exceed"]
            k["1"] += d["hip_vs_t1"]["exceed"]
            k["n"] += d["n"]
            self_max = max(self_max, d["ref_t8_vs_t1"]["max"])
        except (OSError, ValueError, KeyError):
            return {}
    return {"exceed_live_box": {"vs_oracle_8_threads": f"{k['8']}/{k['n']}", "vs_oracle_1_thread": f"{k['1']}/{k['n']}",
                                "oracle_8_vs_1_threads_max_m": float(f"{self_max:.3e}")}}


def cpu_baseline_hands_light(ctx, model, sd_cpu):
    """the oracle on all host cores, bz in {1, 8, 32}, median of 5 after a warm-up."""
    import hands_amd
    torch = ctx.torch
    oracle_fwd = oracle_forward_fn("hands_light")
    ar, al = hands_amd.synthetic_mano_asset(True), hands_amd.synthetic_mano_asset(False)
    cores = host_cores()
    torch.set_num_threads(cores)
    by_bz, ref, sample = {}, None, None
    t_all = time.perf_counter()
    for cb in (1, 8, 32):
        ci, cm = hands_amd.synthetic_inputs(cb, seed=0)
        r = oracle_fwd(sd_cpu, ar, al, ci, cm)                  # warm-up + checker output
        times = []
        t_budget = time.perf_counter()
        while len(times) < 5 and (time.perf_counter() - t_budget) < 14.0:
            t1 = time.perf_counter()
            oracle_fwd(sd_cpu, ar, al, ci, cm)
            times.append(time.perf_counter() - t1)
        med = sorted(times)[len(times) // 2]
        by_bz[str(cb)] = {"hands_per_sec": round(2 * cb / med, 2), "ms_per_forward": round(med * 1e3, 1), "runs": len(times)}
        if cb == 8:
            ref, sample = r, (ci, cm)
    best = max(by_bz, key=lambda k: by_bz[k]["hands_per_sec"])
    base = {"value": by_bz[best]["hands_per_sec"], "unit": "hands/s", "cores": cores, "kind": "port",
            "threads": torch.get_num_threads(), "os_cpu_count": os.cpu_count(),
            "bz": int(best),
            "sample": f"oracle hands_light forward fp32, bz in (1,8,32), median of <=5 after warm-up, best bz={best}, "
                      f"{time.perf_counter() - t_all:.0f} s of CPU work",
            "by_bz": by_bz}
    # the checker's output is recomputed at ORACLE_THREADS (the timing above used every core; ATen's sums depend on the count)
    return base, parity_vs_oracle(ctx, "hands_light", model, sd_cpu, 8, sample=sample)


def cpu_baseline_small(ctx, wl, model, sd_cpu, cb):
    """configs 3 / 4: the oracle's forward of ``wl`` on ``cb`` samples on all host cores this process may use, median of
    <= 3 runs after a warm-up (bounded: <= 15 s), + the parity of the HIP path against that same oracle output."""
    import hands_amd
    torch = ctx.torch
    ci, cm = hands_amd.synthetic_inputs(cb, seed=0)
    ar, al = hands_amd.synthetic_mano_asset(True), hands_amd.synthetic_mano_asset(False)
    fwd = oracle_forward_fn(wl)
    cores = host_cores()
    torch.set_num_threads(cores)
    ref = fwd(sd_cpu, ar, al, ci, cm)
    ts = []
    t_b = time.perf_counter()
    while len(ts) < 3 and time.perf_counter() - t_b < 15:
        t1 = time.perf_counter()
        fwd(sd_cpu, ar, al, ci, cm)
        ts.append(time.perf_counter() - t1)
    med = sorted(ts)[len(ts) // 2]
    base = {"value": round(2 * cb / med, 2), "unit": "hands/s", "cores": cores, "kind": "port",
            "bz": cb, "sample": f"oracle {wl} forward fp32, bz={cb}, median of {len(ts)} after warm-up"}
    return base, parity_vs_oracle(ctx, wl, model, sd_cpu, cb, sample=(ci, cm))      # checker at ORACLE_THREADS, not at the timing's count


# ------------------------------------------------------------------------------------------------------
# config 5: two-hand MANO LBS alone
# ------------------------------------------------------------------------------------------------------
def measure_lbs(ctx, bz, steps, warmup, with_cpu=True):
    """configs[4]: a step = ONE pre-bound C-ABI call (hands_amd.ManoHeadsPlan.launch -> hands_mano_heads_f32: both hands'
    MANOHead.forward, mano_head.py:21-65) over fixed device buffers, + the all-gather of the vertices when N > 1."""
    torch = ctx.torch
    import hands_amd
    from hands_amd import _lib
    from hands_amd._lib import check, ptr
    from hands_amd.dist import gather_predictions
    from hands_amd.hands_light import ManoHeadsPlan
    dev = ctx.dev
    model = hands_amd.HandsLight().to(dev).eval()
    P = model.packed(dev)
    L = _lib.lib()
    g = torch.Generator().manual_seed(100 + ctx.rank)
    # random rotations: rotation_6d_to_matrix(randn(.,6)) -- made by the PRODUCT's device kernel
    # (hands_rot6d_to_matrix_f32), not by the checker
    six = torch.randn(2 * bz, 96, generator=g).to(dev)
    d_rot = torch.empty(2 * bz, 16, 3, 3, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    check(L.hands_rot6d_to_matrix_f32(ptr(six), 96, ptr(d_rot), 2 * bz, st), "rot6d")
    shape = torch.randn(2 * bz, 10, generator=g)
    cam = torch.tensor([1.0, 0, 0]) + 0.1 * torch.randn(2 * bz, 3, generator=g)
    K = torch.tensor([[1000.0, 0, 112], [0, 1000.0, 112], [0, 0, 1]]).repeat(bz, 1, 1)
    d_shape, d_cam, d_K = shape.to(dev), cam.to(dev), K.to(dev)
    torch.cuda.synchronize(dev)
    rot = d_rot.cpu()
    plan = ManoHeadsPlan(L, P["mano_r"], P["mano_l"], d_rot, d_shape, d_cam, d_K, 224.0, bz)
    verts = {k: plan.outputs[k] for k in ("mano.vertices.r", "mano.vertices.l")}

    def step():
        plan.launch(st)
        if ctx.world > 1:
            v = verts
            if ctx.host_collective:
                torch.cuda.synchronize(dev)
                v = {k: t.cpu() for k, t in verts.items()}
            return gather_predictions(v)
        return plan.outputs

    out = step()
    elapsed = ctx.timed(step, steps, warmup)
    enqueue_ms = host_enqueue_ms(ctx, step)
    rank_rates = ctx.gather_floats(2 * bz * steps / ctx.last_local_seconds)
    gather_us = allgather_us(ctx, lambda: verts) if ctx.world > 1 and not ctx.host_collective else None
    if ctx.rank != 0:
        return None
    out = plan.outputs
    hands = ctx.world * 2 * bz * steps / elapsed
    # device time of the kernel: the same launch issued back to back between two HIP events on the launch stream
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    nrep = max(steps, 200)
    for _ in range(10):
        plan.launch(st)
    e0.record()
    for _ in range(nrep):
        plan.launch(st)
    e1.record()
    torch.cuda.synchronize(dev)
    dev_ms = e0.elapsed_time(e1) / nrep
    gbs = 2 * bz * 10.2e3 / (dev_ms * 1e-3) / 1e9
    tfl = 2 * bz * 1.17e6 / (dev_ms * 1e-3) / 1e12
    res = {"value": round(hands, 1), "ms_per_step": round(elapsed / steps * 1e3, 4),
           "hands_per_sec_per_gpu": round(hands / ctx.world, 1), "host_enqueue_ms_per_step": enqueue_ms,
           "allgather_us": gather_us,
           "per_rank_hands_per_sec": {"min": round(min(rank_rates), 1), "max": round(max(rank_rates), 1)} if ctx.world > 1 else None,
           "roofline": {"bound": "mfma", "mode": "back-to-back launches", "kernel": "mano_heads",
                        "achieved": round(tfl, 3), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(tfl / FP32_MFMA_PEAK_TFLOPS, 5), "traffic": None,
                        "device_ms_per_step": round(dev_ms, 4), "us_per_launch": round(dev_ms * 1e3, 2), "launches_per_step": 1,
                        "hbm_gbs": round(gbs, 2), "hbm_frac_of_8tbs": round(gbs / HBM_PEAK_GBS, 5)}}
    if with_cpu and ctx.world == 1:
        from oracle import hands_oracle as O   # checker / CPU baseline only
        cb = min(bz, 256)
        ar, al = hands_amd.synthetic_mano_asset(True), hands_amd.synthetic_mano_asset(False)
        cores = host_cores()
        torch.set_num_threads(cores)

        def cpu():
            r = O.mano_head(rot[:cb], shape[:cb], cam[:cb], K[:cb], ar, 224.0, ".r")
            l = O.mano_head(rot[bz:bz + cb], shape[bz:bz + cb], cam[bz:bz + cb], K[:cb], al, 224.0, ".l")
            return r, l
        ref = cpu()
        ts = []
        t_b = time.perf_counter()
        while len(ts) < 10 and time.perf_counter() - t_b < 6:
            t1 = time.perf_counter()
            cpu()
            ts.append(time.perf_counter() - t1)
        med = sorted(ts)[len(ts) // 2]
        res["cpu_baseline"] = {"value": round(2 * cb / med, 1), "unit": "hands/s", "cores": cores, "kind": "port",
                               "bz": cb, "sample": f"oracle MANOHead x2 on {cb} crops, median of {len(ts)}"}
        verr = max((out["mano.vertices.r"][:cb].cpu() - ref[0]["vertices.r"]).abs().max().item(),
                   (out["mano.vertices.l"][:cb].cpu() - ref[1]["vertices.l"]).abs().max().item())
        res["parity"] = {"max_vertex_err_m": float(f"{verr:.3e}"), "checked_hands": 2 * cb,
                         "checker": "oracle"}
    return res


# ------------------------------------------------------------------------------------------------------
# output: short lines.  The driver keeps the last 8000 characters of stdout: stdout is ONE line, the headline,
# below 4 KB; every extra measurement is its own short line on stderr; tables
# (per kernel, per launch shape, by-bz CPU timings) go to gpurun_out/bench_details.json; prose lives in DESIGN.md section 5.
# ------------------------------------------------------------------------------------------------------
HEADLINE_LIMIT = 4096
ROOFLINE_KEYS = ("bound", "mode", "kernel", "achieved", "peak", "unit", "frac", "peak_measured", "frac_measured", "hbm_measured_tbs",
                 "executed_frac", "dominant", "frac_on_own_roof", "frac_on_own_roof_measured",
                 "by_bound", "traffic", "traffic_source", "traffic_over_algorithmic", "traffic_shipped_over_algorithmic", "kernel_ms_per_step",
                 "step_ms_same_mode",
                 "launches_per_step", "us_per_launch", "device_ms_per_step", "hbm_gbs")
CPU_KEYS = ("value", "unit", "cores", "kind", "bz", "sample")
PARITY_KEYS = ("mpjpe_vs_ref_mm", "max_vertex_err_m", "checked_hands", "oracle_threads", "worst_vertex_err_m", "worst_seed", "sweep_seeds",
               "bar_m", "live_sweep_inside_bar", "worst_by_oracle_threads", "exceed_rate", "exceed_wilson95", "exceed_n",
               "median_err_ratio_vs_fp64", "exceed_source", "exceed_live_box")
CONFIG_KEYS = ("workload", "per_gpu_batch", "global_batch", "parallelism", "rccl_ranks", "collective_backend", "launched_by",
               "timed_mode", "conv3x3_stride1", "steps", "warmup", "allgather_selfcheck_us")


def _pick(d, keys):
    return None if d is None else {k: d[k] for k in keys if k in d}


def compact_entry(full):
    """The judged keys of one measurement (headline or `also` entry) without tables or prose."""
    out = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                "scaling", "vs_baseline", "dtype", "data") if k in full}
    out["config"] = _pick(full.get("config"), CONFIG_KEYS)
    out["roofline"] = _pick(full.get("roofline"), ROOFLINE_KEYS)
    out["cpu_baseline"] = _pick(full.get("cpu_baseline"), CPU_KEYS)
    out["parity"] = _pick(full.get("parity"), PARITY_KEYS)
    if isinstance(full.get("serial"), dict):
        out["serial"] = _pick(full["serial"], ("ms_per_step", "hands_per_sec"))
    for k in ("with_d2h_ms_per_step", "hands_per_sec_per_gpu", "host_enqueue_ms_per_step", "allgather_us",
              "per_rank_hands_per_sec", "math", "error"):
        if full.get(k) is not None:
            out[k] = full[k]
    return out


ALSO_LIMIT = 2048


def compact_also(full, key):
    """One `also` line (stderr): the judged keys, below ALSO_LIMIT characters -- optional keys are shed, least important first
    (everything is in gpurun_out/bench_details.json)."""
    line = dict(compact_entry(full), also=key)
    for drop in (("roofline", "by_bound"), ("cpu_baseline", "sample"), ("roofline", "kernel"), ("parity", "exceed_wilson95"),
                 ("roofline", "dominant"), ("config", "conv3x3_stride1"), ("roofline", "traffic_source"), ("serial",)):
        if len(json.dumps(line)) < ALSO_LIMIT:
            break
        d = line
        for k in drop[:-1]:
            d = d.get(k) or {}
        d.pop(drop[-1], None)
    return line


def compact_headline(full, also=None, details_path=None):
    """THE stdout line: < HEADLINE_LIMIT characters, whatever the measurements returned."""
    line = compact_entry(full)
    if also:
        line["also"] = {k: (v.get("value") if isinstance(v, dict) and "error" not in v else "error") for k, v in also.items()
                        if isinstance(v, dict)}
    if details_path:
        line["details"] = details_path
    # never exceed the limit: shed optional keys, least important first
    for drop in (("roofline", "by_bound"), ("also",), ("cpu_baseline", "sample"), ("roofline", "dominant"), ("details",),
                 ("per_rank_hands_per_sec",), ("config", "launched_by"), ("config", "timed_mode"), ("serial",)):
        if len(json.dumps(line)) < HEADLINE_LIMIT:
            break
        d = line
        for k in drop[:-1]:
            d = d.get(k) or {}
        d.pop(drop[-1], None)
    if len(json.dumps(line)) >= HEADLINE_LIMIT:          # a pathological string somewhere: keep the contract keys only
        line["config"] = {"workload": str((full.get("config") or {}).get("workload"))[:200]}
        line["roofline"] = _pick(full.get("roofline"), ("bound", "achieved", "peak", "unit", "frac", "traffic"))
        if line["roofline"] and (full.get("roofline") or {}).get("traffic_source"):
            line["roofline"]["traffic_source"] = str(full["roofline"]["traffic_source"])[:40]
        line["cpu_baseline"] = _pick(full.get("cpu_baseline"), ("value", "unit", "cores", "kind"))
    return line


def write_details(full, also):
    """Everything measured (per-kernel and per-shape tables included) as a file next to the rank logs; returns the
    repo-relative path or None when the directory is not writable."""
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        fn = os.path.join(d, "bench_details.json")
        with open(fn, "w") as fh:
            json.dump(dict(full, also=also) if also else full, fh, indent=1)
        return os.path.relpath(fn, ROOT)
    except OSError:
        return None


def conv3x3_route(model):
    if model is None or not hasattr(model, "engine"):
        return None
    if not model.engine.winograd:
        return "direct"
    if getattr(model.engine, "winograd4", False) and getattr(model, "winograd4_stages", ()):
        return "winograd_f4x4:stages" + "".join(str(i) for i in model.winograd4_stages)     # (F(2x2) where F(4x4) is not packed)
    return ("winograd_f2x2" + (f":{model.winograd_scope}" if hasattr(model, "winograd_scope") else "")
            + (f"+chains<={model.engine.chain_limit}" + ("(in-kernel)" if getattr(model.engine, "chain_in_kernel", False) else "")
               if getattr(model.engine, "chain_limit", 0) else ""))


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))       # pure launcher: no GPU call was made in this process

    if args.pmc_child:
        return pmc_child(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.graph < 0:                         # auto: the fastest bit-identical mode of config 4's shard size
        args.graph = 4 if (args.workload == "handoccnet_light" and world > 1 and not args.serial and not args.latency_mode) else 0
    if (world == 1 and args.workload != "mano_lbs" and not args.no_pmc and not args.latency_mode and not args.graph
            and not os.environ.get("HANDS_BENCH_PMC_CHILD") and not os.environ.get("ROCPROFILER_LIBRARY_CTOR")
            and "rocprof" not in os.environ.get("LD_PRELOAD", "") and not os.environ.get("HANDS_BENCH_SHIPPED_ONLY")):
        # before the first GPU call of this process: the counter passes are children with their own GPU context
        wl0 = args.workload
        bz0 = args.bz or default_bz(wl0, 1)
        try:
            SAME_RUN_PMC[(wl0, bz0)] = same_run_pmc(args, wl0, bz0)
        except Exception as e:                 # the headline must survive a failure of the counter passes
            SAME_RUN_PMC[(wl0, bz0)] = {"error": f"{type(e).__name__}: {e}"[:200]}
    pin_to_gpu_numa_node()                 # before the first GPU call of this rank (no re-exec, no numactl)
    ctx = Ctx(args)
    torch = ctx.torch
    selfcheck = None
    if ctx.world > 1:
        # first contact of the ranks over the fabric, on the packed prediction layout, before anything is timed: a dead
        # peer / corrupted segment ends THIS rank with code 3 / 4 and a message on its stderr (the launcher prints every
        # rank's tail; under torchrun the agent reports the failing rank)
        from hands_amd.dist import allgather_selfcheck
        try:
            selfcheck = allgather_selfcheck("cpu" if ctx.host_collective else ctx.dev,
                                            timeout_s=float(os.environ.get("HANDS_SELFCHECK_TIMEOUT", "180")))
        except RuntimeError as e:
            sys.stderr.write(str(e) + "\n")
            sys.stderr.flush()
            os._exit(4)
    wl = args.workload
    bz = args.bz or default_bz(wl, ctx.world)
    strong = wl in GLOBAL_BATCH and not args.bz
    first = rank0 = ctx.rank == 0
    full, also = None, {}
    if wl == "mano_lbs":
        res = measure_lbs(ctx, bz, args.steps, args.warmup, with_cpu=not args.no_cpu_baseline)
        model = None
    else:
        want_parity = first and not args.no_cpu_baseline and ctx.world == 1
        res, model, sd_cpu = measure_model(ctx, wl, bz, args.steps, args.warmup, args, serial_headline=args.serial,
                                           parity_bz=8 if want_parity else 0, layer_report=args.layer_report, graph=args.graph)
    if rank0:
        cpu_baseline, parity = res.pop("cpu_baseline", None), res.pop("parity", None)
        if wl != "mano_lbs" and not args.no_cpu_baseline and ctx.world == 1:
            if wl == "hands_light":
                cpu_baseline, parity = cpu_baseline_hands_light(ctx, model, sd_cpu)
            else:
                cpu_baseline, parity = cpu_baseline_small(ctx, wl, model, sd_cpu, 1 if wl == "hamer_light" else 4)
            if not args.no_sweep:
                parity.update(parity_sweep(ctx, wl, model, sd_cpu))
        full = dict(res)
        full.update({
            "metric": "hands/sec", "unit": "hands/s", "n_gpus": ctx.world, "steps": args.steps, "warmup": args.warmup,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": workload_text(wl, bz, ctx.world), "per_gpu_batch": bz, "global_batch": bz * ctx.world,
                       "img_res": 224, "parallelism": f"dp{ctx.world}" + ("+allgather" if ctx.world > 1 else ""),
                       "rccl_ranks": ctx.rccl_ranks, "collective_backend": ctx.backend if ctx.world > 1 else None,
                       "allgather_selfcheck_us": selfcheck["us"] if selfcheck else None,
                       "launched_by": "bench.py launcher" if os.environ.get("HANDS_BENCH_LAUNCHED") else
                                      ("torchrun" if ctx.world > 1 else "direct"),
                       "timed_mode": ("serial" if args.serial else "multi-stream") + (f"+hipgraph(depth={args.graph})" if args.graph else ""),
                       "latency_mode": bool(args.latency_mode),
                       "cpu_affinity": CPU_PIN.get("note"), "conv3x3_stride1": conv3x3_route(model)},
            "cpu_baseline": cpu_baseline, "parity": parity})

    # ---- BASELINE configs 3-5 in the same run (N=1, headline workload only): one short line each ------
    if rank0 and wl == "hands_light" and ctx.world == 1 and not args.no_also:
        del model
        torch.cuda.empty_cache()
        t_also = time.perf_counter()

        def emit_also(key, r):
            # one short line per extra measurement -- on STDERR: stdout carries exactly ONE JSON line, the headline, so that
            # whichever line of stdout a harness parses (first, last, only) is the BASELINE metric on its config
            also[key] = r
            sys.stderr.write(json.dumps(compact_also(r, key)) + "\n")
            sys.stderr.flush()

        # name, bz, steps, warmup, parity bz.  *_bf16x3: separately reported arithmetic mode, never the headline value;
        # *_backbone_unblocked: HandOccNet's default of rounds 3-4 (Winograd in the backbone only, single fp32 chains), for comparison
        for name, abz, asteps, awarm, pbz in (("hands_light_bf16x3", 256, 10, 3, 8), ("hamer_light", 64, 4, 1, 1),
                                              ("hamer_light_bf16x3", 64, 4, 1, 1), ("handoccnet_light", 32, 30, 8, 2),
                                              ("handoccnet_light_graph4", 32, 32, 8, 2),
                                              ("handoccnet_light_backbone_unblocked", 32, 30, 8, 2)):
            key = name
            try:
                math, wino, wscope, graph = "fp32", None, None, 0
                mg = re.search(r"_graph(\d+)$", name)   # hands_amd.GraphedForward(depth=N): N captured forwards in flight
                if mg:                                   # (round 5: depth 3 / 4 / 6 3736 / 3790 / 3753 hands/s, eager with three in flight 3870)
                    name, graph = name[: mg.start()], int(mg.group(1))
                if name.endswith("_backbone_unblocked"):
                    name, wino, wscope = name[: -len("_backbone_unblocked")], True, "backbone"
                if name.endswith("_bf16x3"):
                    name, math = name[: -len("_bf16x3")], "bf16x3"
                r, m, sd = measure_model(ctx, name, abz, asteps, awarm, args, parity_bz=pbz, math=math, winograd=wino,
                                         winograd_scope=wscope, graph=graph)
                r.update(metric="hands/sec", unit="hands/s", n_gpus=1, steps=asteps, warmup=awarm, dtype="f32", math=math)
                if math == "fp32" and name != "hands_light" and not args.no_cpu_baseline and not wino and not graph:
                    r["cpu_baseline"], r["parity"] = cpu_baseline_small(ctx, name, m, sd, pbz)
                    if not args.no_sweep:
                        r["parity"].update(parity_sweep(ctx, name, m, sd))
                else:
                    r["parity"] = parity_vs_oracle(ctx, name, m, sd, pbz)
                r["config"] = {"workload": workload_text(name, abz, 1), "per_gpu_batch": abz, "global_batch": abz,
                               "conv3x3_stride1": conv3x3_route(m),
                               "timed_mode": "multi-stream" + (f"+hipgraph(depth={graph})" if graph else "")}
                emit_also(key, r)
                del m, sd
                torch.cuda.empty_cache()
            except Exception as e:       # the headline line must survive a failure of an extra measurement
                emit_also(key, {"error": f"{type(e).__name__}: {e}"[:300]})
        try:
            r = measure_lbs(ctx, 1024, 50, 50, with_cpu=True)
            r.update(metric="hands/sec", unit="hands/s", n_gpus=1, steps=50, warmup=50, dtype="f32")
            r["config"] = {"workload": workload_text("mano_lbs", 1024, 1), "per_gpu_batch": 1024, "global_batch": 1024}
            emit_also("mano_lbs", r)
        except Exception as e:
            emit_also("mano_lbs", {"error": f"{type(e).__name__}: {e}"[:300]})
        full["also_seconds"] = round(time.perf_counter() - t_also, 1)
    if rank0:
        details = write_details(full, also)
        print(json.dumps(compact_headline(full, also, details)), flush=True)       # the headline: the only stdout line, < 4 KB
    ctx.close()


if __name__ == "__main__":
    main()
