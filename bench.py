#!/usr/bin/env python3
"""Benchmark of the TSDF hot path: depth frames/s -> 32^3 TSDF (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = LAUNCHES_PER_STEP (16) back-to-back fused launches of the voxelizer, each over the rank's batch of
1024 synthetic 320x240 full-frame depth crops (BASELINE.json configs[1]), inputs already resident in HBM: 16,384
frames per GPU and step, ~2.2 ms.  (One launch is 0.135 ms; a timed region of 20 single launches would be 2.7 ms,
and one 0.3 ms scheduling hiccup in one of eight processes would cost the aggregate 10 %.)
Frames are independent, so N GPUs = N ranks each voxelizing its own shard (weak scaling, no data-path
collective).  RCCL carries the process group, the rendezvous and the small gathers of timing numbers; the barriers
that bracket the timed region are node-local (NodeBarrier: /dev/shm, microseconds) because a collective barrier
costs 0.2-0.4 ms and would be charged to N>1 only.  Every rank pins itself to the cores of its GPU's NUMA node
before its first GPU call (the scaling risk of this path is host-side).  Rank 0 prints ONE JSON line.

Extra objects on that line:
  roofline     the fused kernel against the HBM roofline: algorithmic bytes per launch
               (24 + 4*N_px + 12*R^3 + 16 (+8 offsets) per frame) divided by
               the mean launch duration over the timed region, measured with a HIP event pair on
               the launch stream (it includes the ~2 us launch-to-launch gap; the spread of single
               launches comes from a separate pass with per-launch event pairs).
  cpu_baseline the oracle (C restatement of the reference math, oracle/tsdf_oracle.c) timed on
               this box's host cores on a bounded sample of the same frames (rank 0, N=1 only).
  per_rank     event-timed ms per step, frames/s and CPU mask of every rank.
  extras       "configs[3]_sharded" at every N: the 76,500-crop set split by pixel-balanced contiguous shards, every
               rank voxelizing its own (per-rank and aggregate frames/s); the other BASELINE configs, the loaders and
               the latency table at N=1.

TSDF_BENCH_REHEARSAL=1  dry run of the N>1 path on a box with fewer GPUs than ranks (ranks share devices, gloo
                        instead of RCCL).  TSDF_BENCH_DRYRUN=1: no GPU at all — the launches are a host stub; this
                        exists so that a CPU test can drive this file's own multi-rank code (tests/test_bench_ranks.py).
                        Neither is ever set by the driver; the line then says so and its numbers mean nothing.
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FRAMES_PER_LAUNCH = 1024
LAUNCHES_PER_STEP = 16
FRAMES_PER_GPU = FRAMES_PER_LAUNCH          # (name kept: frames per GPU and launch)
ROTATION = 6                                # distinct resident batches (+ output buffers) a rank's launches rotate through
RES = 32
ALL_SUBJECTS = 76500   # BASELINE configs[3]: all nine MSRA subjects
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md chip table)
HBM_COPY_GBS = 6290.0  # what a plain copy kernel reaches on this part (same guide, "measured copy ceiling")
# SURVEY.md section 6 [probe]: the reference's own Python loop (pre/tsdf_for.py::tsdf_f), measured once in the survey
# container (1 core of an 8-core Xeon @ 2.1 GHz).  The reference's Python does not travel to the GPU box, so this is
# quoted, never re-measured; it is what "the reference on a CPU" means — the oracle below is a C port, ~400x faster.
SURVEY_PY_LOOP = {"tsdf_f_crop_120x140_s_per_frame": 0.104, "tsdf_f_full_320x240_s_per_frame": 0.159,
                  "DataProcess.process_crop_s_per_frame": 0.185,
                  "where": "SURVEY.md section 6: survey container, 1 core, Python 3.10 / numpy 2.2 — quoted, not re-measured"}


def algorithmic_bytes(offsets: np.ndarray, n: int, R: int) -> int:
    """header 24 + depth 4*N (read once) + tsdf 12*R^3 + max_l/mid_p 16 + offset 8."""
    n_px = int(offsets[n] - offsets[0])
    return 4 * n_px + n * (24 + 8 + 12 * R ** 3 + 16)


# ---- CPU affinity: each rank on the cores next to its GPU --------------------------------------------------------
def _parse_cpulist(txt: str):
    cpus = set()
    for part in txt.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus.update(range(int(a), int(b or a) + 1))
    return cpus


def _format_cpulist(cpus) -> str:
    cpus = sorted(cpus)
    out, i = [], 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        out.append(str(cpus[i]) if i == j else f"{cpus[i]}-{cpus[j]}")
        i = j + 1
    return ",".join(out)


def _visible_devices(env) -> "list[int] | None":
    """HIP device ordinal -> index among the node's GPUs, from ROCR_VISIBLE_DEVICES (applied by the runtime below HIP)
    and then HIP_VISIBLE_DEVICES (or its alias CUDA_VISIBLE_DEVICES when that one is unset), when they are plain index
    lists; None = identity; [] = unreadable (UUIDs or the like)."""
    order = None
    hip = env.get("HIP_VISIBLE_DEVICES")
    for v in (env.get("ROCR_VISIBLE_DEVICES"), hip if hip is not None else env.get("CUDA_VISIBLE_DEVICES")):
        if v is None or v.strip() == "":
            continue
        try:
            idx = [int(x) for x in v.split(",") if x.strip() != ""]
        except ValueError:
            return []
        if order is None:
            order = idx
        elif any(i >= len(order) for i in idx):
            return []
        else:
            order = [order[i] for i in idx]
    return order


def gpu_topology(sysfs: str = "/sys", env=None):
    """[(pci address, NUMA node, set of local CPUs)] of the GPUs in HIP enumeration order, read from the KFD topology
    and the PCI devices' sysfs entries WITHOUT touching the GPU (so that it can run before the first HIP call).
    Empty when anything is unreadable."""
    env = os.environ if env is None else env
    base = os.path.join(sysfs, "class", "kfd", "kfd", "topology", "nodes")
    try:
        nodes = sorted(int(d) for d in os.listdir(base) if d.isdigit())
    except OSError:
        return []
    gpus = []
    for nd in nodes:
        try:
            props = dict(ln.split()[:2] for ln in open(os.path.join(base, str(nd), "properties")) if len(ln.split()) >= 2)
        except OSError:
            continue      # a GPU of the host that this container may not open (device cgroup): HIP does not enumerate it either
        if int(props.get("simd_count", "0")) <= 0:
            continue      # a CPU node
        loc, dom = int(props.get("location_id", "0")), int(props.get("domain", "0"))
        bdf = f"{dom:04x}:{(loc >> 8) & 0xff:02x}:{(loc >> 3) & 0x1f:02x}.{loc & 7}"
        pdir = os.path.join(sysfs, "bus", "pci", "devices", bdf)
        try:
            cpus = _parse_cpulist(open(os.path.join(pdir, "local_cpulist")).read())
            numa = int(open(os.path.join(pdir, "numa_node")).read().strip())
        except (OSError, ValueError):
            return []
        gpus.append((bdf, numa, cpus))
    vis = _visible_devices(env)
    if vis is not None:
        if any(i >= len(gpus) for i in vis):
            return []
        gpus = [gpus[i] for i in vis]
    return gpus


def plan_affinity(topo, device_of_rank, allowed):
    """CPU set per local rank: the allowed CPUs local to the rank's GPU; ranks whose GPUs share a NUMA node split that
    node's CPUs into contiguous equal parts (a part of fewer than 2 CPUs is not worth having: they then share).
    None for a rank whose GPU is unknown or has no allowed local CPU (the caller leaves its mask alone)."""
    by_node = {}
    for r, d in enumerate(device_of_rank):
        if 0 <= d < len(topo):
            by_node.setdefault(topo[d][1], []).append(r)
    plan = [None] * len(device_of_rank)
    for node, ranks in by_node.items():
        cpus = sorted(set().union(*[topo[device_of_rank[r]][2] for r in ranks]) & set(allowed))
        if not cpus:
            continue
        per = len(cpus) // len(ranks)
        for k, r in enumerate(ranks):
            plan[r] = set(cpus[k * per:(k + 1) * per]) if per >= 2 else set(cpus)
    return plan


def pin_rank(local_rank: int, local_world: int, n_devices_hint: "int | None", rehearsal: bool, sysfs: str = "/sys"):
    """Pin this process (and every thread it starts later: the HIP runtime's included) to its GPU's cores.  Returns a
    small report for the JSON line; never raises — an unreadable topology leaves the mask as it is."""
    try:
        allowed = os.sched_getaffinity(0)
    except AttributeError:
        return {"pinned": False, "why": "no sched_getaffinity"}
    topo = gpu_topology(sysfs)
    rep = {"pinned": False, "cpus": _format_cpulist(allowed), "gpus_seen": len(topo)}
    if not topo:
        rep["why"] = "GPU topology unreadable"
        return rep
    ndev = len(topo) if not n_devices_hint else min(len(topo), n_devices_hint)
    dev_of = [(r % ndev) if rehearsal else r for r in range(local_world)]
    plan = plan_affinity(topo, dev_of, allowed)
    mine = plan[local_rank] if local_rank < len(plan) else None
    if not mine:
        rep["why"] = "no allowed CPU local to the GPU"
        return rep
    try:
        os.sched_setaffinity(0, mine)
    except OSError as e:
        rep["why"] = f"sched_setaffinity: {e}"
        return rep
    d = dev_of[local_rank]
    rep.update(pinned=True, cpus=_format_cpulist(mine), numa_node=topo[d][1], pci=topo[d][0])
    return rep


def allowed_cpus():
    """The CPUs this process was allowed when it started (captured at import, before any rank pins itself next to its GPU)."""
    return set(_ALLOWED_AT_START) if _ALLOWED_AT_START else set(range(os.cpu_count() or 1))


try:
    _ALLOWED_AT_START = frozenset(os.sched_getaffinity(0))
except AttributeError:   # pragma: no cover
    _ALLOWED_AT_START = frozenset()


def cgroup_cpu_quota(root: str = "/sys/fs/cgroup", proc_cgroup: str = "/proc/self/cgroup"):
    """CPUs' worth of run time the cgroup CPU controller grants this process: quota / period of the tightest limit on the
    path from the process's cgroup up to the root (cgroup v2 `cpu.max`, v1 `cpu.cfs_quota_us` / `cpu.cfs_period_us`).
    Returns (cpus as a float or None when unlimited or unreadable, where it was read)."""
    paths = {}
    try:
        for ln in open(proc_cgroup):
            _, ctrl, path = ln.rstrip("\n").split(":", 2)
            paths[ctrl] = path
    except (OSError, ValueError):
        pass
    best, where, seen = None, "no cgroup CPU controller file readable", []

    def walk(base, rel):
        parts = [p for p in rel.split("/") if p]
        for k in range(len(parts), -1, -1):
            yield os.path.join(base, *parts[:k])

    # v2 (unified hierarchy, either at the root or under "unified")
    for base in (root, os.path.join(root, "unified")):
        for d in walk(base, paths.get("", "/")):
            try:
                q, per = open(os.path.join(d, "cpu.max")).read().split()[:2]
            except (OSError, ValueError):
                continue
            seen.append(os.path.join(d, "cpu.max"))
            if q != "max" and float(per) > 0 and (best is None or float(q) / float(per) < best):
                best, where = float(q) / float(per), os.path.join(d, "cpu.max")
    # v1
    rel = next((v for k, v in paths.items() if "cpu" in k.split(",")), "/")
    for base in (os.path.join(root, "cpu"), os.path.join(root, "cpu,cpuacct")):
        for d in walk(base, rel):
            try:
                q = float(open(os.path.join(d, "cpu.cfs_quota_us")).read())
                per = float(open(os.path.join(d, "cpu.cfs_period_us")).read())
            except (OSError, ValueError):
                continue
            seen.append(os.path.join(d, "cpu.cfs_quota_us"))
            if q > 0 and per > 0 and (best is None or q / per < best):
                best, where = q / per, os.path.join(d, "cpu.cfs_quota_us")
    if best is None and seen:
        where = "no limit set in " + ", ".join(seen)
    return best, where


def plan_cpu_legs(n_allowed: int, quota):
    """Thread counts the CPU baseline times, and the one the box is believed to be able to run in parallel.  1 thread
    always; the quota's count when the cgroup states one (never more than the affinity mask allows); with no quota the
    affinity count — and, when that is above 16, 16 as well: a one-GPU box of the pool shows all 256 CPUs of its host in
    the mask and no cgroup quota, while its real share is 16 (256 OpenMP threads then run 5x SLOWER than 16: round 4)."""
    if quota is not None:
        usable = max(1, min(n_allowed, int(quota + 0.999)))
        why = "cgroup quota"
    else:
        usable = max(1, n_allowed)
        why = "affinity mask (no cgroup quota)"
    legs = [1]
    if quota is None and n_allowed > 16:
        legs.append(16)
    if usable not in legs:
        legs.append(usable)
    return legs, usable, why


def cpu_baseline(depth, offsets, headers, single_s=4.0, multi_s=2.0, legs_override=None):
    """Time the oracle on host cores over the same frames (BASELINE.md section 4 / SURVEY.md 8(d)): one thread, and the
    number of threads the box can really run — the cgroup CPU quota when there is one, else the affinity mask (plus a
    16-thread leg when that mask is wider than 16: see plan_cpu_legs).  `value` / `cores` are the FASTEST leg; every leg
    is listed with its thread count.  Whole passes over the 1024-frame batch are repeated until a leg's budget is used, so
    the sample is always a multiple of the bench workload.  The rank pinned itself to its GPU's NUMA node at start-up: for
    this measurement the mask the process STARTED with is put back (OpenMP workers inherit the mask of the thread that
    creates them) and the pinned one restored afterwards."""
    import oracle  # test infrastructure; used here only as the reported CPU baseline

    oracle.lib()  # build/load outside the timed region
    n = FRAMES_PER_GPU
    allowed = allowed_cpus()
    pinned = None
    try:
        pinned = os.sched_getaffinity(0)
        os.sched_setaffinity(0, allowed)
    except (AttributeError, OSError):
        pinned = None
    n_all = len(allowed)
    quota, quota_where = cgroup_cpu_quota()
    legs, usable, why = plan_cpu_legs(n_all, quota)
    if legs_override:
        legs = list(legs_override)

    def leg(nthreads, budget):
        oracle.voxelize(depth[: offsets[16]], offsets[:17], headers[:16], R=RES, n_threads=nthreads)  # warm
        frames, used, t0 = 0, nthreads, time.perf_counter()
        while True:
            r = oracle.voxelize(depth, offsets, headers, R=RES, n_threads=nthreads)
            frames += n
            used = r["threads"]
            dt = time.perf_counter() - t0
            if dt >= budget:
                return {"threads": int(used), "frames_per_s": round(frames / dt, 1), "frames": frames, "seconds": round(dt, 2)}

    try:
        done = [leg(t, single_s if t == 1 else multi_s) for t in legs]
    finally:
        if pinned is not None:
            try:
                os.sched_setaffinity(0, pinned)
            except OSError:
                pass
    best = max(done, key=lambda r: r["frames_per_s"])
    single = done[0]
    return {
        "value": best["frames_per_s"], "unit": "frames/s", "cores": best["threads"], "kind": "port",
        "sample": f"oracle/tsdf_oracle.c (C restatement of the reference math; the numba path itself is not "
                  f"runnable: no usable numba, no params.py) over the same 1024 synthetic frames; value = the fastest of "
                  f"{len(done)} legs: {best['frames']} frames in {best['seconds']:.2f} s on {best['threads']} OpenMP threads.  "
                  f"Thread counts timed: {[r['threads'] for r in done]}; parallel count {usable} chosen by the {why}"
                  + (f" ({quota:.2f} CPUs, {quota_where})" if quota is not None else f" of {n_all} CPUs ({quota_where})"),
        "legs": done,
        "single_thread_value": single["frames_per_s"],
        "usable_cpus": usable, "usable_cpus_from": why, "cgroup_quota_cpus": quota,
        "allowed_cpus": n_all, "allowed_cpulist": _format_cpulist(allowed), "os_cpu_count": os.cpu_count(),
        "provenance": "a C/OpenMP PORT written for this project, orders of magnitude faster than anything the reference "
                      "itself can run on a CPU; the reference's own implementation is the Python loop below",
        "reference_python_loop_frames_per_s_per_core": round(1.0 / SURVEY_PY_LOOP["tsdf_f_full_320x240_s_per_frame"], 2),
        "reference_python_loop": SURVEY_PY_LOOP,
    }


class NodeBarrier:
    """Barrier for the ranks of ONE node through a small file in /dev/shm: every rank publishes the number of the
    barrier it has reached in its own 64-byte slot and spins until all slots show it (a few microseconds, against
    0.2-0.4 ms for a collective barrier).  The frames shard with no exchange, so the only thing a barrier does here is
    bracket the timed region; RCCL still carries the process group, the one-time rendezvous below and the gathers of
    timing numbers.  `ok` is False (and the caller keeps using the collective barrier) when the ranks do not see
    each other's writes within the timeout, e.g. ranks in different /dev/shm namespaces."""

    def __init__(self, dist, rank, world, collective_barrier, flag_device, timeout_s=5.0):
        self.rank, self.world, self.n, self.ok = rank, world, 0, False
        run = os.environ.get("TORCHELASTIC_RUN_ID", "x") + "_" + os.environ.get("MASTER_PORT", "0")
        self.path = f"/dev/shm/tsdf_bench_barrier_{run}"
        try:
            if rank == 0:
                with open(self.path, "wb") as f:
                    f.write(b"\0" * (64 * world))
            collective_barrier()  # the file exists and is zeroed before anyone maps it
            self.slots = np.memmap(self.path, dtype=np.int64, mode="r+", shape=(world, 8))
            good = self._wait(timeout_s)
        except OSError:
            good = False
        flag = torch.tensor([1 if good else 0], dtype=torch.int32, device=flag_device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        self.ok = bool(flag.item())
        collective_barrier()
        if rank == 0:
            try:
                os.unlink(self.path)  # the mappings stay valid
            except OSError:
                pass

    def _wait(self, timeout_s=None):
        self.n += 1
        self.slots[self.rank, 0] = self.n
        col = self.slots[:, 0]
        t0 = time.perf_counter()
        while int(col.min()) < self.n:
            if timeout_s is not None and time.perf_counter() - t0 > timeout_s:
                return False
        return True

    def __call__(self):
        if not self._wait(600.0):   # a rank that died: fail instead of spinning for ever
            raise RuntimeError(f"rank {self.rank}: node barrier {self.n} not reached by every rank within 600 s")


# ---- the device under test, or its host stand-in ------------------------------------------------------------------
class HipBackend:
    """The product path: torch for memory / events / streams, the voxelizer through its C ABI."""

    name = "hip"

    def __init__(self, dev_index):
        torch.cuda.set_device(dev_index)
        self.dev = torch.device("cuda", dev_index)
        self.pkg = importlib.import_module("handposeestimation-with-3d-cnns_amd")

    def upload(self, depth, offsets, headers):
        return tuple(torch.from_numpy(np.ascontiguousarray(a)).to(self.dev) for a in (depth, offsets, headers))

    def alloc_out(self, d, o, h, res=RES):
        out = self.pkg.voxelize(d, o, h, res=res)
        torch.cuda.synchronize()
        assert bool((out.status == 0).all())
        return out

    def launch(self, d, o, h, out, res=RES):
        self.pkg.voxelize(d, o, h, res=res, out=out)

    def sync(self):
        torch.cuda.synchronize()

    def event(self):
        return torch.cuda.Event(enable_timing=True)

    def record(self, ev):
        ev.record()

    def elapsed_ms(self, a, b):
        return a.elapsed_time(b)

    def kernel_name(self, n, res=RES):
        """The instantiation the library launches for this batch, from the library itself (tsdf_describe_launch)."""
        import ctypes
        L = self.pkg._lib.load()
        buf = ctypes.create_string_buffer(160)
        with torch.cuda.device(self.dev):
            rc = L.tsdf_describe_launch(int(n), int(res), 0, 0, buf, 160)
        return buf.value.decode() if rc == 0 else f"unknown (tsdf_describe_launch returned {rc})"

    def pci(self):
        p = torch.cuda.get_device_properties(self.dev)
        try:
            return f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        except AttributeError:
            return None


class HostStub:
    """TSDF_BENCH_DRYRUN=1: no GPU, a launch is a 20 us busy wait.  Exists ONLY so that the multi-rank orchestration of
    this file (pinning, shard planning, barriers, gathers, the JSON line) can be driven by a CPU test.  It computes
    nothing: it is not a CPU path of the voxelizer, and the line it produces is marked."""

    name = "host-stub"

    def __init__(self, dev_index):
        self.dev = torch.device("cpu")
        self.pkg = None

    def upload(self, depth, offsets, headers):
        return depth, offsets, headers

    def alloc_out(self, d, o, h, res=RES):
        return None

    def launch(self, d, o, h, out, res=RES):
        t = time.perf_counter() + 20e-6
        while time.perf_counter() < t:
            pass

    def sync(self):
        pass

    def event(self):
        return [0.0]

    def record(self, ev):
        ev[0] = time.perf_counter()

    def elapsed_ms(self, a, b):
        return (b[0] - a[0]) * 1e3

    def kernel_name(self, n, res=RES):
        return "host stub (no kernel)"

    def pci(self):
        return None


def _time_launches(fn, k, warm=3, warm_ms=40.0, cold=None):
    """Mean time of k back-to-back launches (us), HIP events on the current stream, at STEADY STATE: after the `warm`
    launches the workload keeps being launched until `warm_ms` of GPU time have passed.  Round 4 found that a kernel that
    is co-bound by instruction issue (the augmented 64^3 one) runs 5-15 % slower for its first ~30 launches (~25 ms) after
    the GPU did something else — 829, 717, 750, 776, 800, 816, ... 700, 690 us launch by launch
    (profiles/r04/warmup.log) — while store-bound kernels do not care; rounds 1-3 timed 10 launches after 3 warm-up ones
    and so reported the transient for that kernel (747 us where the steady state was ~705).  `cold` (a dict) receives that
    old-style figure as well, so that the two can be told apart."""
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a.record()
    for _ in range(k):
        fn()
    b.record()
    torch.cuda.synchronize()
    first = a.elapsed_time(b) / k * 1e3
    if cold is not None:
        cold["us_per_launch_first_%d_after_%d_warmup" % (k, warm)] = round(first, 1)
    spent = first * k * 1e-3
    while spent < warm_ms:
        n = max(1, min(64, int((warm_ms - spent) / max(first * 1e-3, 1e-3)) + 1))
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        spent += a.elapsed_time(b)
    a.record()
    for _ in range(k):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / k * 1e3


def live_traffic(timeout_s=60.0):
    """HBM bytes per launch of the headline kernel from PMC counters collected BY THIS RUN: three rocprofv3 child passes
    (FETCH_SIZE of the full entry, WRITE_SIZE of the full entry, FETCH_SIZE of the phase-1-only entry — counters never
    share a pass with anything but --kernel-trace, FETCH and WRITE need separate passes: MI355X_MICROARCH.md) over
    tools/exp_pmc.py, which launches the same 1024 seeded frames the timed region uses.  Corrections as the guide
    prescribes and tools/collect_profiles.py documents: both counters are KiB; FETCH_SIZE reports a wide (16 B/lane)
    streaming read at 1/2, so the depth stream — isolated by the phase-1-only pass — counts twice; the staging re-read
    (4 B/lane LDS-DMA, uncalibrated) is priced at its known byte count; WRITE_SIZE is exact.  The children launch over
    ROTATION buffer sets in rotation, like the timed region.  Returns a dict, None (no rocprofv3) or {"failed": why} (a pass
    failed or timed out): the caller then falls back to the tracked profiles/pmc_traffic.json and says so.  Children are separate processes (this process has initialised the GPU: it must not exec)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None
    tmp = tempfile.mkdtemp(prefix="tsdf_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", PMC_LAUNCHES="12", PMC_ROTATE=str(ROTATION))
    kernel = "tsdf_fused_kernel<32, 0, false, false"

    failed = []

    def one(counter, mode):
        d = os.path.join(tmp, counter + "_" + mode)
        e = dict(env, PMC_MODE=mode)
        # its own session: on a timeout the whole group goes (rocprofv3 runs the program as a child — killing the launcher
        # alone would leave a python on the GPU under the extras that follow)
        try:
            p = subprocess.Popen([rocprof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--",
                                  sys.executable, os.path.join(ROOT, "tools", "exp_pmc.py")],
                                 env=e, cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
        except OSError as err:
            failed.append(f"{counter}/{mode}: {err}")
            return None
        try:
            rc = p.wait(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            import signal
            try:
                os.killpg(p.pid, signal.SIGKILL)
            except OSError:
                pass
            p.wait()
            failed.append(f"{counter}/{mode}: live pass timed out after {timeout_s:.0f} s (process group killed)")
            return None
        if rc != 0:
            failed.append(f"{counter}/{mode}: rocprofv3 exit code {rc}")
            return None
        vals = []
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if kernel in row["Kernel_Name"] and row["Counter_Name"] == counter:
                    vals.append(float(row["Counter_Value"]))
        if not vals:
            failed.append(f"{counter}/{mode}: no row of {kernel}")
        return float(np.median(vals)) if vals else None

    try:
        fetch = one("FETCH_SIZE", "full")
        write = one("WRITE_SIZE", "full") if fetch is not None else None
        fetch_p1 = one("FETCH_SIZE", "aabb") if write is not None else None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    if fetch is None or write is None or fetch_p1 is None:
        return {"failed": "; ".join(failed) or "unknown"}
    synth = importlib.import_module("handposeestimation-with-3d-cnns_amd.synth")
    stage = 0
    for i in range(FRAMES_PER_LAUNCH):     # the staged rectangle of every frame: 4 bytes x its bounding rectangle of valid pixels
        h, d = synth.synth_frame(i, "full")
        ys, xs = np.nonzero(np.abs(d.reshape(h[5] - h[3], h[4] - h[2])) >= 1)
        stage += int((ys.max() - ys.min() + 1) * (xs.max() - xs.min() + 1)) * 4
    wide = fetch_p1 * 1024 * 2
    return {"hbm_bytes_per_launch": wide + stage + write * 1024, "depth_stream_bytes": wide, "staging_bytes_known": stage,
            "staging_bytes_reported": (fetch - fetch_p1) * 1024, "write_bytes": write * 1024,
            "FETCH_SIZE_KiB": fetch, "FETCH_SIZE_KiB_phase1_only": fetch_p1, "WRITE_SIZE_KiB": write}


def stream_ceilings():
    """What plain stream kernels reach on THIS box (tools/probes/hbm_probe.hip, built by __graft_entry__.build(); a child
    process, ~2 s): read-only, write-only, copy and the voxelizer's 307:393 read:write mix, best grid of four.  Context
    for roofline.frac — the spec peak is not what a stream gets, and reads and writes differ.  None if not built."""
    exe = os.path.join(ROOT, "tools", "probes", "hbm_probe.bin")
    if not os.path.exists(exe):
        return None
    if "rocprof" in os.environ.get("LD_PRELOAD", "") or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return None   # under a profiler: no second GPU program from inside the profiled one
    import subprocess
    try:
        txt = subprocess.run([exe, "384"], capture_output=True, text=True, timeout=120).stdout
    except Exception:
        return None
    import re
    best = {}
    for line in txt.splitlines():
        m = re.match(r"grid\s+\d+ x1024\s+(.*?)\s+[\d.]+ us\s+([\d.]+) GB/s", line)
        if m:
            best[m.group(1)] = max(best.get(m.group(1), 0.0), float(m.group(2)))
    keys = {"read": "read_only", "write nt": "write_only_nt", "copy": "copy", "copy nt": "copy_nt",
            "same, nt stores": "read_then_write_307_393_nt", "read then write 307:393": "read_then_write_307_393"}
    res = {v: best[k] for k, v in keys.items() if k in best}
    if not res:
        return None
    res["unit"] = "GB/s"
    res["what"] = ("plain persistent stream kernels, 16 B per lane, 384 MiB buffers, best of 1/2/4/8 workgroups per CU, on this "
                   "box.  copy = reads and writes interleaved all the time (what a kernel that overlaps its phases produces); "
                   "read_then_write = every thread reads its share first and writes afterwards (the chip alternates "
                   "between two one-directional streams)")
    return res


# ---- BASELINE configs[3]: all nine subjects, frame-sharded across the ranks ------------------------------------------
N_DISTINCT_CROPS = 2048


def crop_lengths(synth):
    """Pixels per frame of the 76,500-crop set WITHOUT generating a frame: the set is N_DISTINCT_CROPS seeded MSRA-like
    crops repeated; a crop's size is the first thing its generator draws (synth.synth_frame)."""
    lens = np.empty(N_DISTINCT_CROPS, np.int64)
    for i in range(N_DISTINCT_CROPS):
        rng = np.random.default_rng(1234 + 100000 + i)
        bw = int(rng.integers(90, 161))
        bh = int(rng.integers(90, 161))
        lens[i] = bw * bh
    reps = (ALL_SUBJECTS + N_DISTINCT_CROPS - 1) // N_DISTINCT_CROPS
    return np.tile(lens, reps)[:ALL_SUBJECTS]


def shard_frames(synth, packing, a, b):
    """Frames [a, b) of the 76,500-crop set as one packed batch (only the distinct crops the shard needs are generated)."""
    ids = np.arange(a, b) % N_DISTINCT_CROPS
    uniq = np.unique(ids)
    made = {int(u): synth.synth_frame(100000 + int(u), "crop") for u in uniq}
    lens = np.array([made[int(i)][1].size for i in ids], np.int64)
    off = np.zeros(b - a + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    depth = np.empty(int(off[-1]), np.float32)
    hdr = np.empty((b - a, 6), np.int32)
    for k, i in enumerate(ids):
        h, d = made[int(i)]
        depth[off[k]:off[k + 1]] = d
        hdr[k] = h
    return packing.PackedFrames(depth, off, hdr)


def config3_sharded(be, synth, shard, packing, rank, world, launches=3):
    """This rank's pixel-balanced contiguous shard of the 76,500-crop set, resident, voxelized by `launches` launches
    (HIP events).  Replaces the subject x gesture x frame loop of pre/read_MSRA.py:51,78,98-106.  Returns
    (frames, pixels, seconds per launch, index of the shard's first frame)."""
    lens = crop_lengths(synth)
    a, b = shard.shard_bounds(ALL_SUBJECTS, world, weights=lens)[rank]
    if be.name == "host-stub":
        off = np.concatenate([[0], np.cumsum(lens[a:b])])
        d = o = h = out = None
    else:
        pk = shard_frames(synth, packing, a, b)
        assert np.array_equal(np.diff(pk.offsets), lens[a:b]), "crop_lengths() no longer mirrors synth.synth_frame()"
        off = pk.offsets
        d, o, h = be.upload(pk.depth, pk.offsets, pk.headers)
        out = be.alloc_out(d, o, h)
    be.launch(d, o, h, out)   # warm
    e0, e1 = be.event(), be.event()
    be.sync()
    be.record(e0)
    for _ in range(launches):
        be.launch(d, o, h, out)
    be.record(e1)
    be.sync()
    sec = be.elapsed_ms(e0, e1) / launches * 1e-3
    del d, o, h, out
    if be.name == "hip":
        torch.cuda.empty_cache()
    return b - a, int(off[-1]), sec, a


def one_frame_report(pkg, synth, dev):
    """BASELINE configs[0] the way pre/time_test.py:24-30 would report it: ONE MSRA-like frame through each
    implementation.  (a) the cal_tsdf_cuda shim end to end — upload, one launch, volume back on the host, as
    pre/tsdf_numba.py:133-158 does with four copies and two launches; (b) the same frame resident, one launch, result left
    on the GPU; (c) the oracle (C port) on one core; (d) the reference's Python loop, quoted from the survey."""
    import oracle
    h, d = synth.synth_frame(100007, "crop")
    s = {"header": h, "data": d}
    for _ in range(5):
        r = pkg.cal_tsdf_cuda(s)
    assert r is not None
    ts = []
    for _ in range(60):
        t0 = time.perf_counter()
        pkg.cal_tsdf_cuda(s)
        ts.append(time.perf_counter() - t0)
    pc2 = np.array([[-60.0, -70.0, -480.0], [70.0, 60.0, -380.0]], np.float32)   # the loop entry takes its grid from a cloud
    for _ in range(5):
        pkg.tsdf_f({"header": h, "depth": d}, pc2)
    tf = []
    for _ in range(60):
        t0 = time.perf_counter()
        pkg.tsdf_f({"header": h, "depth": d}, pc2)
        tf.append(time.perf_counter() - t0)
    td = torch.from_numpy(d).to(dev)
    to = torch.tensor([0, d.size], dtype=torch.int64, device=dev)
    th = torch.from_numpy(h[None]).to(dev)
    out = pkg.voxelize(td, to, th)
    singles = []
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(60):
        torch.cuda.synchronize()
        a.record()
        pkg.voxelize(td, to, th, out=out)
        b.record()
        torch.cuda.synchronize()
        singles.append(a.elapsed_time(b) * 1e-3)
    off = np.array([0, d.size], np.int64)
    oracle.voxelize(d, off, h[None], R=RES, n_threads=1)
    tc = []
    for _ in range(30):
        t0 = time.perf_counter()
        oracle.voxelize(d, off, h[None], R=RES, n_threads=1)
        tc.append(time.perf_counter() - t0)
    return {
        "frame": f"one MSRA-like crop, bbox {int(h[4] - h[2])}x{int(h[5] - h[3])}, -> 32^3",
        "cal_tsdf_cuda_shim_s": round(float(np.median(ts)), 6),
        "cal_tsdf_cuda_shim_what": "handposeestimation-with-3d-cnns_amd.tsdf_numba.cal_tsdf_cuda(s): offsets + header + crop up as ONE "
                                   "page-locked block, ONE launch, volume + max_l + mid_p + status back as one block, one "
                                   "synchronisation, numpy copies out (host wall, median of 60; the reference: 4 copies, 2 launches)",
        "tsdf_f_shim_s": round(float(np.median(tf)), 6),
        "tsdf_f_shim_what": "handposeestimation-with-3d-cnns_amd.tsdf_for.tsdf_f(data, point_cloud): the CPU-loop entry's signature "
                            "(pre/tsdf_for.py:6-20; float64 [c,x,y,z] result) served by the GPU the same way",
        "voxelize_resident_s": round(float(np.median(singles)), 7),
        "voxelize_resident_what": "the same frame already on the GPU, one launch, result left there (HIP events, median of 60)",
        "oracle_one_core_s": round(float(np.median(tc)), 6),
        "reference_python_loop_s": SURVEY_PY_LOOP["tsdf_f_crop_120x140_s_per_frame"],
        "reference_python_loop_what": SURVEY_PY_LOOP["where"] + " (tsdf_for.tsdf_f, bbox 120x140)",
        "what": "BASELINE configs[0] / pre/time_test.py:24-30 (the reference's only benchmark: one frame, CPU loop vs GPU path)",
    }


def extras(pkg, synth, dev, batches, offsets):
    """The other BASELINE.json configs and the small-batch latencies, measured OUTSIDE the timed region (rank 0,
    N=1).  Every entry says what it ran; rates are device-resident unless the entry says "streamed".  `batches` = the
    timed region's resident (depth, offsets, headers, outputs) sets; everything but the rotation entry uses the first."""
    import torch.utils.data as tdata
    td, to, th = batches[0][:3]
    ex = {}
    ex["configs[0]_one_frame"] = one_frame_report(pkg, synth, dev)
    # ---- small batches: launch latency as a training step sees it (reference batch size: 16, 3D_CNN/train.py:36)
    lat = {}
    for n in (1, 16, 64, 256):
        d, o, h = td[: int(offsets[n])], to[: n + 1].contiguous(), th[:n].contiguous()
        out = pkg.voxelize(d, o, h, res=RES)
        b2b = _time_launches(lambda: pkg.voxelize(d, o, h, res=RES, out=out), 200, 20)
        singles = []
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(40):
            torch.cuda.synchronize()
            a.record()
            pkg.voxelize(d, o, h, res=RES, out=out)
            b.record()
            torch.cuda.synchronize()
            singles.append(a.elapsed_time(b) * 1e3)
        lat[str(n)] = {"back_to_back_us": round(b2b, 2), "single_launch_us_median": round(float(np.median(singles)), 2)}
    ex["latency_full_frames"] = lat
    # ---- other resolutions / sizes of the same fused kernel
    def resident(name, d, o, h, R, k, offs_np, fn=None, note=""):
        n = h.shape[0]
        out = fn(None) if fn else pkg.voxelize(d, o, h, res=R)
        cold = {}
        us = _time_launches((lambda: fn(out)) if fn else (lambda: pkg.voxelize(d, o, h, res=R, out=out)), k, cold=cold)
        ab = algorithmic_bytes(offs_np, n, R)
        ex[name] = {"frames": n, "res": R, "us_per_launch": round(us, 1), "frames_per_s": round(n / us * 1e6),
                    "algorithmic_GBps": round(ab / us / 1e3, 1), "frac_of_hbm_peak": round(ab / us / 1e3 / HBM_PEAK_GBS, 4)}
        ex[name].update(cold)      # what rounds 1-3 reported under us_per_launch: the first k launches after 3 warm-up ones
        if note:
            ex[name]["what"] = note
        del out
    # The headline's two figures once more, paired in one pass at steady state (_time_launches: 120 launches each after
    # 40 ms of the same workload): the timed region's batches in rotation, and the first batch re-launched.  With one
    # batch part of its 315 MB of depth is still in the 256 MiB Infinity Cache when the next launch reads it
    # (tools/exp_mall.py: inputs alone in rotation cost +5 %, outputs alone +1 %).
    state = {"i": 0}

    def launch_rot():
        a_ = batches[state["i"] % len(batches)]
        state["i"] += 1
        pkg.voxelize(a_[0], a_[1], a_[2], res=RES, out=a_[3])
    us_rot = _time_launches(launch_rot, 120, warm=12)
    us_same = _time_launches(lambda: pkg.voxelize(td, to, th, res=RES, out=batches[0][3]), 120, warm=12)
    ab1 = algorithmic_bytes(offsets, FRAMES_PER_LAUNCH, RES)
    ex["full_1024_six_batches_in_rotation"] = {
        "frames": FRAMES_PER_LAUNCH, "res": RES, "us_per_launch": round(us_rot, 1), "us_per_launch_same_batch": round(us_same, 1),
        "frames_per_s": round(FRAMES_PER_LAUNCH / us_rot * 1e6), "frac_of_hbm_peak": round(ab1 / us_rot / 1e3 / HBM_PEAK_GBS, 4),
        "frac_of_hbm_peak_same_batch": round(ab1 / us_same / 1e3 / HBM_PEAK_GBS, 4),
        "what": f"the headline workload ({len(batches)} resident batches and output buffer sets in rotation) and one batch "
                "re-launched, 120 launches each, back to back in one pass"}
    # The same rotation issued alternately on TWO HIP streams: consecutive launches then overlap on the GPU — one launch's
    # workgroups take the CUs the previous launch's tail frees — which is what a consumer that voxelizes batch k+1 while it
    # still works on batch k gets (the entry is re-entrant and thread-safe for distinct streams; every stream has its own
    # work-queue word).  Whole-region throughput, NOT a kernel duration: the overlapped kernels each run longer, so this
    # figure is no roofline fraction of "the kernel" and never the headline.
    s_a, s_b = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def two_stream_region(k):
        cur = torch.cuda.current_stream(dev)
        e0.record(cur)
        for st in (s_a, s_b):
            st.wait_event(e0)
        for i in range(k):
            with torch.cuda.stream(s_a if i % 2 == 0 else s_b):
                launch_rot()
        for st in (s_a, s_b):
            cur.wait_stream(st)
        e1.record(cur)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / k * 1e3
    two_stream_region(48)
    us_two = min(two_stream_region(120) for _ in range(3))
    ex["full_1024_rotation_two_streams"] = {
        "frames": FRAMES_PER_LAUNCH, "res": RES, "us_per_launch_throughput": round(us_two, 1),
        "frames_per_s": round(FRAMES_PER_LAUNCH / us_two * 1e6),
        "algorithmic_GBps": round(ab1 / us_two / 1e3, 1), "frac_of_hbm_peak_throughput": round(ab1 / us_two / 1e3 / HBM_PEAK_GBS, 4),
        "what": "the headline's rotation with consecutive launches issued alternately on two HIP streams (120 launches, best of "
                "3 regions): launches overlap, the tail of one fills with the head of the next.  Throughput of the region; the "
                "single-stream figures above are the kernel's own duration"}
    # the same kernel on 4096 frames (the 1024 frames four times over): eight frames per half-workgroup instead of two, i.e.
    # what the launch's tail costs at the BASELINE batch size ("Where the headline launch's time goes")
    d4 = td.repeat(4)
    o4 = torch.from_numpy(np.concatenate([offsets[:-1] + k * int(offsets[-1]) for k in range(4)] + [[4 * int(offsets[-1])]])).to(dev)
    h4 = th.repeat(4, 1)
    resident("full_4096", d4, o4, h4, RES, 10, o4.cpu().numpy(), note="4096 full 320x240 frames -> 32^3 in one launch")
    del d4, o4, h4
    resident("full_1024_r64", td, to, th, 64, 10, offsets, note="1024 full 320x240 frames -> 64^3, plain")
    # configs[4]: 64^3 with the fused 3-D augmentation (reference distributions, augment.random_affines)
    mid = pkg.voxelize(td, to, th).mid_p.cpu().numpy()
    xf = torch.from_numpy(pkg.augment.random_affines(mid, rng=np.random.RandomState(2026))[0]).to(dev)
    resident("configs[4]_aug_r64", td, to, th, 64, 10, offsets,
             fn=lambda out: pkg.voxelize_aug(td, to, th, xf, res=64, out=out),
             note="BASELINE configs[4]: 1024 full frames -> 64^3 with the 3-D augmentation fused into the kernel")
    # MSRA-like crops: 2,048 distinct seeded crops, repeated to the stated counts
    crops = [synth.synth_frame(100000 + i, "crop") for i in range(N_DISTINCT_CROPS)]
    base = pkg.packing.pack_frames(crops)

    def tiled(n):
        reps = (n + N_DISTINCT_CROPS - 1) // N_DISTINCT_CROPS
        lens = np.tile(np.diff(base.offsets), reps)[:n]
        off = np.zeros(n + 1, np.int64)
        np.cumsum(lens, out=off[1:])
        return pkg.packing.PackedFrames(np.ascontiguousarray(np.tile(base.depth, reps)[: off[-1]]), off,
                                        np.ascontiguousarray(np.tile(base.headers, (reps, 1))[:n]))

    pk1 = tiled(1024)
    c1 = pk1.to_torch(dev)
    resident("crops_1024", c1[0], c1[1], c1[2], RES, 30, pk1.offsets, note="1024 MSRA-like crops (bbox side 90-160 px)")
    del c1
    # configs[2]: one subject (8,500 crops) streamed from a pinned pack through VoxelLoader: H2D on a copy stream
    # overlapped with the voxelizer (PCIe-inclusive: never the headline value)
    pk2 = tiled(8500)
    pk2.gt = np.zeros((8500, 63), np.float32)
    ds = pkg.MSRADepthDataset.from_packs([pk2])
    loader = pkg.VoxelLoader(ds, batch_size=1024, device=dev, max_pixels=1024 * 160 * 160)
    rates = []
    for _ in range(7):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nseen = 0
        for batch in loader:
            nseen += batch.tsdf.shape[0]
        torch.cuda.synchronize()
        rates.append(nseen / (time.perf_counter() - t0))
    in_bytes = pk2.depth.size * 4
    ex["configs[2]_streamed"] = {
        "frames": 8500, "batch": 1024, "crops_per_s": round(max(rates[2:])),
        "h2d_GBps": round(in_bytes * max(rates[2:]) / 8500 / 1e9, 2),
        "epochs_crops_per_s": [round(r) for r in rates],
        "what": "BASELINE configs[2]: 8,500 MSRA-like crops from a page-locked pack through dataset.VoxelLoader "
                "(ONE hipMemcpyAsync per batch on a copy stream, overlapped with the fused voxelizer + labels, which reads "
                "offsets / headers / labels from page-locked host memory); PCIe-bound"}
    # configs[2] with the subject RESIDENT on the GPU (ResidentLoader / tsdf_voxelize_indexed_hip): the pack is uploaded
    # once, shuffled batches are drawn by index on the device; next to it what shuffling costs the host-fed loader
    def epochs(ld, k):
        r = []
        for _ in range(k):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            seen = 0
            for batch in ld:
                seen += batch[0].shape[0]
            torch.cuda.synchronize()
            r.append(seen / (time.perf_counter() - t0))
        return r
    rl = pkg.ResidentLoader(ds, batch_size=1024, device=dev, shuffle=True)
    r1024 = epochs(rl, 5)
    r16 = epochs(pkg.ResidentLoader(ds, batch_size=16, device=dev, shuffle=True), 3)
    r16p = epochs(pkg.ResidentLoader(ds, batch_size=16, device=dev, shuffle=True, prefetch=64), 5)
    h1024 = epochs(pkg.VoxelLoader(ds, batch_size=1024, device=dev, shuffle=True, max_pixels=1024 * 160 * 160), 3)
    ex["configs[2]_resident_shuffled"] = {
        "frames": 8500, "resident_bytes": rl.resident_bytes(),
        "batch_1024_crops_per_s": round(max(r1024[1:])),
        # (key meanings as in round 2: batch_16 = one launch per batch; the prefetch ring has its own key.  Round 3's line
        # had reported the prefetch variant under "batch_16_crops_per_s".)
        "batch_16_crops_per_s": round(max(r16[1:])),
        "batch_16_prefetch64_crops_per_s": round(max(r16p[1:])),
        "host_fed_shuffled_batch_1024_crops_per_s": round(max(h1024[1:])),
        "what": "BASELINE configs[2] with the subject's pack resident in HBM (uploaded once; all of MSRA is 4.8 GB): shuffled "
                "batches drawn by index on the device, labels included (dataset.ResidentLoader).  batch_16 = the reference's "
                "training batch size (3D_CNN/train.py:36), one launch per batch; batch_16_prefetch64 = the same with "
                "prefetch=64: the epoch's permutation is uploaded once, one launch voxelizes 64 batches into a ring and the "
                "loader yields 16-frame views (bit-identical batches); host_fed = the same shuffled batches through VoxelLoader, "
                "whose host gathers the crops before uploading them"}
    # the literal north-star consumer: the reference's own loader call (3D_CNN/train.py:36,86-91) over the on-the-fly dataset
    mds = pkg.MSRA_Dataset.from_raw(ds, device=dev)
    dl = tdata.DataLoader(mds, batch_size=16, shuffle=True)
    rdl = epochs(dl, 4)

    class _Floor(tdata.Dataset):      # what torch's loader machinery costs by itself: a dataset that does nothing
        def __init__(self, n, item):
            self.n, self.item = n, item
        def __len__(self):
            return self.n
        def __getitems__(self, idx):
            return self.item
    first = next(iter(dl))
    floor = epochs(tdata.DataLoader(_Floor(8500, [pkg.dataset.PreBatched(tuple(first))]), batch_size=16, shuffle=True), 3)
    ex["north_star_dataloader_batch_16"] = {
        "frames": 8500, "batch": 16, "crops_per_s": round(max(rdl[1:])),
        "loader_floor_crops_per_s": round(max(floor[1:])),
        "what": "DataLoader(MSRA_Dataset(...), batch_size=16, shuffle=True) — the reference's training loader call "
                "(3D_CNN/train.py:36,86-91) — on the resident on-the-fly dataset: every batch is ONE launch issued from "
                "__getitems__ (indices by value inside the kernel arguments, outputs into a recycled ring, no per-item "
                "tensors, collation by a registered type) and is GPU-bound (tools/exp_dataloader16.py).  loader_floor = the same DataLoader call over a dataset whose __getitems__ "
                "returns a constant: the cost of torch's sampler / fetcher / iterator machinery on this host, which bounds any "
                "dataset behind that call"}
    ra = epochs(pkg.ResidentLoader(ds, batch_size=1024, device=dev, shuffle=True, res=64, augment=True), 3)
    ex["configs[4]_resident_shuffled_aug_r64"] = {
        "frames": 8500, "batch": 1024, "crops_per_s": round(max(ra[1:])),
        "what": "BASELINE configs[4] as a training loop would run it: the subject resident in HBM, shuffled batches by index, a "
                "fresh reference-distribution 3-D augmentation per frame drawn on the host (numpy) and fused into the 64^3 "
                "voxelizer, mapped joints + labels from the same launch (dataset.ResidentLoader(augment=True, res=64))"}
    del rl, dl, mds
    # the same pipeline over four subjects' worth of frames: the 8,500-frame number above carries the fixed cost of
    # starting and draining a 9-batch epoch
    del loader, ds
    pk4 = tiled(34000)
    pk4.gt = np.zeros((34000, 63), np.float32)
    loader = pkg.VoxelLoader(pkg.MSRADepthDataset.from_packs([pk4]), batch_size=1024, device=dev, max_pixels=1024 * 160 * 160)
    rates = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nseen = 0
        for batch in loader:
            nseen += batch.tsdf.shape[0]
        torch.cuda.synchronize()
        rates.append(nseen / (time.perf_counter() - t0))
    # what the link gives on this box: the same page-locked bytes in the same 1024-frame pieces, copies only
    pinned, o4 = pk4._pinned, pk4.offsets
    dbuf = [torch.empty(1024 * 160 * 160, dtype=torch.float32, device=dev) for _ in range(2)]
    cs = torch.cuda.Stream(dev)
    raw = 0.0
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(cs):
            for k, a in enumerate(range(0, 34000, 1024)):
                src = pinned[int(o4[a]):int(o4[min(34000, a + 1024)])]
                dbuf[k & 1][: src.numel()].copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        raw = max(raw, 34000 / (time.perf_counter() - t0))
    ex["streamed_34000_crops"] = {"frames": 34000, "batch": 1024, "crops_per_s": round(max(rates[1:])),
                                  "h2d_GBps": round(pk4.depth.size * 4 * max(rates[1:]) / 34000 / 1e9, 2),
                                  "raw_h2d_crops_per_s": round(raw),
                                  "raw_h2d_GBps": round(pk4.depth.size * 4 * raw / 34000 / 1e9, 2),
                                  "what": "the configs[2] pipeline over 34,000 crops (2.1 GB page-locked), and the bare "
                                          "hipMemcpyAsync rate of the same bytes in the same pieces on this box"}
    return ex


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the other configs / latency table (rank 0, N=1)")
    ap.add_argument("--no-config3", action="store_true", help="skip configs[3]_sharded (every rank, every N)")
    ap.add_argument("--no-same-batch", action="store_true",
                    help="skip the one-batch-re-launched pass behind roofline.frac_same_batch (profile runs: every traced "
                         "launch of the kernel is then a launch of the rotation, and the trace's average is roofline's)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not run the three rocprofv3 PMC child passes that measure roofline.traffic (rank 0, N=1)")
    args = ap.parse_args()

    # stdout carries ONE line, the JSON: everything else that writes to file descriptor 1 during the run — RCCL prints a
    # five-line version banner there when the first communicator comes up — is sent to stderr instead
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    dry = os.environ.get("TSDF_BENCH_DRYRUN") == "1"
    # TSDF_BENCH_REHEARSAL=1: dry run of the N>1 code path on a box with fewer GPUs than ranks (ranks share
    # devices, the barrier / gathers go over gloo instead of RCCL, the line says "rehearsal": true).
    rehearsal = os.environ.get("TSDF_BENCH_REHEARSAL") == "1" or dry

    # ---- pin to the GPU's cores BEFORE the first GPU call (the runtime's helper threads inherit the mask); counting
    # devices does not initialise the GPU on this image
    ndev_hint = None if dry else (torch.cuda.device_count() or None)
    affinity = pin_rank(local_rank, local_world, ndev_hint, rehearsal, os.environ.get("TSDF_BENCH_SYSFS", "/sys"))

    if not dry and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the voxelizer has no CPU path")
    dev_index = 0 if dry else (local_rank % torch.cuda.device_count() if rehearsal else local_rank)
    be = (HostStub if dry else HipBackend)(dev_index)
    got_pci = be.pci()
    if got_pci and affinity.get("pci"):
        affinity["pci_matches_runtime"] = got_pci[:10] == affinity["pci"][:10]   # domain:bus:device
        if not affinity["pci_matches_runtime"]:
            # the runtime enumerates the GPUs in another order than the KFD topology lists them: pin again, late, to the
            # cores of the GPU this rank really has (its helper threads keep the first mask; better than the wrong node)
            topo = gpu_topology(os.environ.get("TSDF_BENCH_SYSFS", "/sys"))
            hit = [i for i, t in enumerate(topo) if t[0][:10] == got_pci[:10]]
            if hit:
                try:
                    cpus = topo[hit[0]][2] & set(os.sched_getaffinity(0)) or topo[hit[0]][2]
                    os.sched_setaffinity(0, cpus)
                    affinity.update(repinned=True, cpus=_format_cpulist(os.sched_getaffinity(0)), numa_node=topo[hit[0]][1],
                                    pci=topo[hit[0]][0])
                except OSError:
                    pass

    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:
        # launched by torch.distributed.run: RCCL for rendezvous + small gathers only
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=be.dev)
    comm_dev = "cpu" if (rehearsal or dist is None) else be.dev

    def gather(vals):
        """[world][len(vals)] float64 on every rank (N small numbers: host-side bookkeeping, not the data path)."""
        t = torch.tensor([vals], dtype=torch.float64, device=comm_dev)
        if dist is None:
            return t.cpu().numpy()
        out = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        return torch.cat(out).cpu().numpy()

    synth = importlib.import_module("handposeestimation-with-3d-cnns_amd.synth")
    shard = importlib.import_module("handposeestimation-with-3d-cnns_amd.shard")
    packing = importlib.import_module("handposeestimation-with-3d-cnns_amd.packing")

    # This rank's ROTATION batches of 1024 seeded synthetic frames, all resident, each with its own output buffers: launch
    # i voxelizes batch i % ROTATION.  One batch re-launched would keep part of its 315 MB of depth in the 256 MiB
    # Infinity Cache from launch to launch (7-8 % faster: profiles/r04/mall.log) — a consumer never sees that: in
    # pre/tsdf_numba.py:119-161 every call is a new frame.  Six batches = 4.3 GB of distinct memory per GPU.
    # Batch k of rank r = frames [(k*world + r)*1024, +1024) of the seeded set (k = 0: what rounds 1-4 timed).
    batches = []
    seeds = [(k * world + rank) * FRAMES_PER_GPU for k in range(ROTATION)]
    for k in range(ROTATION):
        if dry:
            depth = np.zeros(FRAMES_PER_GPU * 76800, np.float32)
            offsets = np.arange(FRAMES_PER_GPU + 1, dtype=np.int64) * 76800
            headers = np.tile(np.array([320, 240, 0, 0, 320, 240], np.int32), (FRAMES_PER_GPU, 1))
        else:
            depth, offsets, headers = synth.synth_batch(FRAMES_PER_GPU, "full", seed0=seeds[k],
                                                        threads=min(8, len(os.sched_getaffinity(0))))
        if k == 0:
            depth0, offsets0, headers0 = depth, offsets, headers     # the CPU baseline's sample, the traffic bookkeeping
        d_, o_, h_ = be.upload(depth, offsets, headers)
        batches.append((d_, o_, h_, be.alloc_out(d_, o_, h_)))     # outputs allocated once
    depth, offsets, headers = depth0, offsets0, headers0
    td, to, th, out = batches[0]
    abytes = algorithmic_bytes(offsets, FRAMES_PER_GPU, RES)     # the same for every batch: full frames, 76,800 px each
    working_set = ROTATION * abytes

    def collective_barrier():
        dist.barrier() if rehearsal else dist.barrier(device_ids=[dev_index])

    node_barrier = NodeBarrier(dist, rank, world, collective_barrier, comm_dev) if dist is not None else None

    def barrier():
        be.sync()
        if dist is not None:
            node_barrier() if node_barrier.ok else collective_barrier()
        be.sync()

    turn = [0]

    def step():
        for _ in range(LAUNCHES_PER_STEP):
            d_, o_, h_, out_ = batches[turn[0] % ROTATION]
            turn[0] += 1
            be.launch(d_, o_, h_, out_)

    for _ in range(args.warmup):
        step()
    # HIP events on the launch stream (torch's current stream): one pair around the K timed steps.
    # (Per-launch pairs were dropped from the timed region: every timestamped record costs ~3 us of
    # stream idle time between two 135 us kernels; they are taken in a separate pass below.)
    ev0, ev1 = be.event(), be.event()

    barrier()
    t0 = time.perf_counter()
    be.record(ev0)
    for k in range(args.steps):
        step()
    be.record(ev1)
    barrier()
    elapsed = time.perf_counter() - t0

    n_launch = args.steps * LAUNCHES_PER_STEP
    my_event_ms = be.elapsed_ms(ev0, ev1)

    # the same number of launches over ONE batch (what rounds 1-4 reported as the headline): outside the timed region
    same_ms = float("nan")
    if not args.no_same_batch:
        ev2, ev3 = be.event(), be.event()
        for _ in range(LAUNCHES_PER_STEP):
            be.launch(td, to, th, out)
        be.sync()
        be.record(ev2)
        for _ in range(n_launch):
            be.launch(td, to, th, out)
        be.record(ev3)
        be.sync()
        same_ms = be.elapsed_ms(ev2, ev3) / n_launch

    timing = gather([elapsed, my_event_ms, same_ms] + seeds)          # [world][3 + ROTATION]
    elapsed_max = float(timing[:, 0].max())

    # diagnostic pass (outside the timed region): per-launch event pairs -> spread of single launches (in rotation)
    ev = [(be.event(), be.event()) for _ in range(48)]
    for i, (a, b) in enumerate(ev):
        d_, o_, h_, out_ = batches[i % ROTATION]
        be.record(a)
        be.launch(d_, o_, h_, out_)
        be.record(b)
    be.sync()
    kern_ms = np.array([be.elapsed_ms(a, b) for a, b in ev])
    launch_ms = my_event_ms / n_launch  # mean launch-to-launch time over the timed region

    # ---- BASELINE configs[3]: every rank voxelizes its shard of the 76,500-crop set (no collective on the data path)
    c3 = None
    if not args.no_config3:
        f3, px3, s3, a3 = config3_sharded(be, synth, shard, packing, rank, world)
        g3 = gather([f3, px3, s3, a3])               # [world][4]
        fr, px, sec = g3[:, 0], g3[:, 1], g3[:, 2]
        c3 = {
            "frames": int(fr.sum()), "per_rank_frames": [int(v) for v in fr],
            "per_rank_first_frame": [int(v) for v in g3[:, 3]],
            "per_rank_pixels": [int(v) for v in px],
            "per_rank_ms_per_launch": [round(float(v) * 1e3, 3) for v in sec],
            "per_rank_fps": [round(float(f / s)) for f, s in zip(fr, sec)],
            "aggregate_fps": round(float(fr.sum() / sec.max())),
            "imbalance": round(float(sec.max() / sec.mean()), 4),
            "what": "BASELINE configs[3]: all nine subjects' worth of MSRA-like crops (76,500), contiguous shards balanced by "
                    "pixels (shard.shard_bounds), every rank voxelizes its own shard resident in its GPU's HBM in one launch "
                    "(HIP events, mean of 3); aggregate = all frames / the slowest rank's launch; imbalance = slowest / mean"}

    if rank == 0:
        total_frames = world * FRAMES_PER_GPU * n_launch
        mean_ms = float(launch_ms)
        achieved = abytes / (mean_ms * 1e-3) / 1e9
        # HBM bytes per launch from the rocprofv3 PMC passes of this same command (separate FETCH_SIZE and
        # WRITE_SIZE runs, gfx950 corrections applied; tools/make_profiles.sh writes the file)
        traffic = None
        traffic_source = None
        traffic_parts = None
        under_profiler = any(k.startswith("ROCPROF") or k.startswith("ROCP_") for k in os.environ)
        live_note = None
        if world == 1 and not dry and not rehearsal and not args.no_live_traffic and not under_profiler:
            traffic_parts = live_traffic()
            if traffic_parts and "hbm_bytes_per_launch" in traffic_parts:
                traffic = traffic_parts["hbm_bytes_per_launch"]
                traffic_source = (f"measured by this run: three rocprofv3 --pmc child passes (FETCH_SIZE, WRITE_SIZE, FETCH_SIZE of "
                                  f"the phase-1-only entry) over the first batch's 1024 frames held in {ROTATION} buffer sets "
                                  "launched in rotation like the timed region; FETCH x2 for the wide depth stream, the staging "
                                  "re-read at its known byte count, WRITE exact.  FETCH_SIZE counts what L2 requests from the "
                                  "fabric, Infinity Cache hits included: it is the same with one batch re-launched")
            elif traffic_parts:
                live_note = traffic_parts.get("failed")
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if traffic is None and os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
                traffic_source = ("profiles/pmc_traffic.json: rocprofv3 FETCH_SIZE/WRITE_SIZE passes of this command from the "
                                  "tracked profile run, NOT measured in this run (the live passes were skipped or failed"
                                  + (": " + live_note if live_note else "") + ")")
                traffic_parts = None
            except Exception:
                traffic = None
        line = {
            "metric": "depth frames/sec to 32^3 TSDF",
            "value": round(total_frames / elapsed_max, 1),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed_max / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "barrier": (("node (/dev/shm)" if node_barrier.ok else "collective") if dist is not None else "none (one process)"),
            # self-describing for an N-rank run: what carried the process group ("nccl" IS RCCL on ROCm) and how many ranks
            "dist_backend": (dist.get_backend() if dist is not None else None),
            "dist_world_size": (dist.get_world_size() if dist is not None else 1),
            "config": {
                "workload": "BASELINE configs[1]: batch 1024 synthetic 320x240 full-frame depth crops -> 32^3 "
                            "3-channel TSDF per GPU and launch, inputs resident in HBM; one step = "
                            f"{LAUNCHES_PER_STEP} back-to-back fused launches, each over the next of {ROTATION} resident "
                            "batches of 1024 frames (no batch is re-read from a cache)",
                "frames_per_launch": FRAMES_PER_LAUNCH, "launches_per_step": LAUNCHES_PER_STEP, "batches_in_rotation": ROTATION,
                "frames_per_gpu_per_step": FRAMES_PER_LAUNCH * LAUNCHES_PER_STEP, "res": RES, "layout": "czyx",
                "parallelism": f"frame-sharded x{world}, no collective",
            },
            "per_rank": {
                "ms_per_step_events": [round(float(v) / args.steps, 4) for v in timing[:, 1]],
                "ms_per_step_host_wall": [round(float(v) / args.steps * 1e3, 4) for v in timing[:, 0]],
                "frames_per_s_events": [round(FRAMES_PER_GPU * n_launch / (float(v) * 1e-3)) for v in timing[:, 1]],
                # every rank's own roofline fraction: its algorithmic bytes per second over the 8 TB/s of ITS GPU
                "frac_of_hbm_peak_events": [round(abytes * n_launch / (float(v) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                                            for v in timing[:, 1]],
                "frac_of_hbm_peak_same_batch": [(round(abytes / (float(v) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if v == v else None)
                                                for v in timing[:, 2]],
                "batch_seed0": [[int(v) for v in row[3:]] for row in timing],
                "what": "every rank's own timed region: HIP events on its launch stream, and host wall between the two barriers; "
                        "batch_seed0 = the first seed of each of the rank's resident batches (1024 consecutive seeds each)",
            },
            "roofline": {
                "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "traffic_source": traffic_source, "traffic_parts": traffic_parts,
                "what": f"the timed region itself: every launch takes the next of {ROTATION} resident batches and output buffer sets "
                        f"({working_set / 1e9:.2f} GB of distinct memory >> the 256 MiB Infinity Cache), so no launch finds its input "
                        "in a cache; frac_same_batch = the same number of launches over ONE batch (rounds 1-4's headline)",
                "frac_same_batch": (round(abytes / (same_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if same_ms == same_ms else None),
                "launch_ms_mean_same_batch": (round(float(same_ms), 4) if same_ms == same_ms else None),
                "rotation": ROTATION, "working_set_bytes": working_set,
                "frac_of_measured_copy": round(achieved / HBM_COPY_GBS, 4),
                "kernel": be.kernel_name(FRAMES_PER_LAUNCH, RES), "algorithmic_bytes_per_launch": abytes,
                "launch_ms_mean": round(mean_ms, 4),
                "single_launch_ms_median": round(float(np.median(kern_ms)), 4),
                "single_launch_ms_min": round(float(kern_ms.min()), 4),
            },
        }
        if rehearsal:
            line["rehearsal"] = True
        if dry:
            line["dry_run"] = "TSDF_BENCH_DRYRUN=1: host stub, no GPU work — orchestration test only, the numbers mean nothing"
        ex = {}
        if c3 is not None:
            ex["configs[3]_sharded"] = c3
        if world == 1 and not args.no_cpu_baseline:
            # (a dry run times the oracle on its all-background frames for a moment: the object's shape, not a number)
            line["cpu_baseline"] = (cpu_baseline(depth, offsets, headers, 0.05, 0.05, legs_override=[1, 2]) if dry
                                    else cpu_baseline(depth, offsets, headers))
        if world == 1 and not args.no_extras and not rehearsal:
            ex.update(extras(be.pkg, synth, be.dev, batches, offsets))
            sc = stream_ceilings()
            if sc:
                ex["stream_ceilings"] = sc
                cp = max(sc.get("copy_nt", 0.0), sc.get("copy", 0.0))
                if cp > 0:   # interleaved reads and writes are the fair comparison for this kernel
                    line["roofline"]["frac_of_copy_stream_this_box"] = round(line["roofline"]["achieved"] / cp, 4)
                    if traffic:
                        line["roofline"]["traffic_rate_GBps"] = round(traffic / (mean_ms * 1e-3) / 1e9, 1)
                        line["roofline"]["traffic_rate_over_copy_stream"] = round(traffic / (mean_ms * 1e-3) / 1e9 / cp, 4)
        if ex:
            line["extras"] = ex

    # every rank's CPU mask, on rank 0's line (small python objects: bookkeeping, not the data path)
    if dist is not None:
        masks = [None] * world
        dist.all_gather_object(masks, affinity)
    else:
        masks = [affinity]
    if rank == 0:
        line["per_rank"]["affinity"] = masks
        json_out.write(json.dumps(line) + "\n")
        json_out.flush()

    if dist is not None:
        collective_barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
