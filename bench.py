#!/usr/bin/env python3
"""Benchmark of the TSDF hot path: depth frames/s -> 32^3 TSDF (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one fused launch of the voxelizer over the rank's batch of 1024 synthetic
320x240 full-frame depth crops (BASELINE.json configs[1]), inputs already resident in HBM.
Frames are independent, so N GPUs = N ranks each voxelizing its own 1024-frame shard
(weak scaling, no data-path collective).  RCCL carries the process group, the rendezvous and the
max-over-ranks of the elapsed time; the barriers that bracket the timed region are node-local
(NodeBarrier: /dev/shm, microseconds) because a collective barrier costs 0.2-0.4 ms — a tenth of the
driver's 20-step timed region — and would be charged to N>1 only.  Rank 0 prints ONE JSON line.

Extra objects on that line:
  roofline     the fused kernel against the HBM roofline: algorithmic bytes per launch
               (24 + 4*N_px + 12*R^3 + 16 (+8 offsets) per frame) divided by
               the mean launch duration over the timed region, measured with a HIP event pair on
               the launch stream (it includes the ~2 us launch-to-launch gap; the spread of single
               launches comes from a separate pass with per-launch event pairs).
  cpu_baseline the oracle (C restatement of the reference math, oracle/tsdf_oracle.c) timed on
               this box's host cores on a bounded sample of the same frames (rank 0, N=1 only).
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

FRAMES_PER_GPU = 1024
RES = 32
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md chip table)
HBM_COPY_GBS = 6290.0  # what a plain copy kernel reaches on this part (same guide, "measured copy ceiling")


def algorithmic_bytes(offsets: np.ndarray, n: int, R: int) -> int:
    """header 24 + depth 4*N (read once) + tsdf 12*R^3 + max_l/mid_p 16 + offset 8."""
    n_px = int(offsets[n] - offsets[0])
    return 4 * n_px + n * (24 + 8 + 12 * R ** 3 + 16)


def host_threads() -> int:
    """Threads for the CPU baseline: the cores this process may use, capped at the 1-GPU box's CPU
    share (16) so that the baseline does not oversubscribe a shared host."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    return max(1, min(avail, 16))


def cpu_baseline(depth, offsets, headers, single_s=4.0, multi_s=1.5):
    """Time the oracle on host cores over the same frames: ~4 s on one thread plus ~1.5 s wall on all
    threads (16 on the GPU box: about 28 s of CPU work in total).  Whole passes over the 1024-frame batch are repeated until
    the leg's budget is used, so the sample is always a multiple of the bench workload."""
    import oracle  # test infrastructure; used here only as the reported CPU baseline

    oracle.lib()  # build/load outside the timed region
    n = FRAMES_PER_GPU
    threads = host_threads()

    def leg(nthreads, budget):
        oracle.voxelize(depth[: offsets[16]], offsets[:17], headers[:16], R=RES, n_threads=nthreads)  # warm
        frames, used, t0 = 0, nthreads, time.perf_counter()
        while True:
            r = oracle.voxelize(depth, offsets, headers, R=RES, n_threads=nthreads)
            frames += n
            used = r["threads"]
            dt = time.perf_counter() - t0
            if dt >= budget:
                return frames / dt, frames, dt, used

    fps1, n1, t1, _ = leg(1, single_s)
    fpsN, nN, tN, used = leg(threads, multi_s)
    return {
        "value": round(fpsN, 1), "unit": "frames/s", "cores": int(used), "kind": "port",
        "sample": f"oracle/tsdf_oracle.c (C restatement of the reference math; the numba path itself is not "
                  f"runnable: no usable numba, no params.py) over the same 1024 synthetic frames: "
                  f"{nN} frames in {tN:.2f} s on {used} OpenMP threads; single thread {n1} frames in {t1:.2f} s",
        "single_thread_value": round(fps1, 1),
    }


class NodeBarrier:
    """Barrier for the ranks of ONE node through a small file in /dev/shm: every rank publishes the number of the
    barrier it has reached in its own 64-byte slot and spins until all slots show it (a few microseconds, against
    0.2-0.4 ms for a collective barrier — 10 % of the driver's 20-step timed region).  The frames shard with no
    exchange, so the only thing a barrier does here is bracket the timed region; RCCL still carries the process
    group, the one-time rendezvous below and the max-over-ranks of the elapsed time.  `ok` is False (and the caller
    keeps using the collective barrier) when the ranks do not see each other's writes within the timeout, e.g. ranks
    in different /dev/shm namespaces."""

    def __init__(self, dist, rank, world, collective_barrier, timeout_s=5.0):
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
        flag = torch.tensor([1 if good else 0], dtype=torch.int32, device="cuda" if dist.get_backend() == "nccl" else "cpu")
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
        t0 = time.perf_counter()
        while int(self.slots[:, 0].min()) < self.n:
            if timeout_s is not None and time.perf_counter() - t0 > timeout_s:
                return False
        return True

    def __call__(self):
        self._wait()


def _time_launches(fn, k, warm=3):
    """Mean time of k back-to-back launches (us), HIP events on the current stream."""
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(k):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / k * 1e3


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


def extras(pkg, synth, dev, td, to, th, offsets):
    """The other BASELINE.json configs and the small-batch latencies, measured OUTSIDE the timed region (rank 0,
    N=1).  Every entry says what it ran; rates are device-resident unless the entry says "streamed"."""
    ex = {}
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
        us = _time_launches((lambda: fn(out)) if fn else (lambda: pkg.voxelize(d, o, h, res=R, out=out)), k)
        ab = algorithmic_bytes(offs_np, n, R)
        ex[name] = {"frames": n, "res": R, "us_per_launch": round(us, 1), "frames_per_s": round(n / us * 1e6),
                    "algorithmic_GBps": round(ab / us / 1e3, 1), "frac_of_hbm_peak": round(ab / us / 1e3 / HBM_PEAK_GBS, 4)}
        if note:
            ex[name]["what"] = note
        del out
    resident("full_1024_r64", td, to, th, 64, 10, offsets, note="1024 full 320x240 frames -> 64^3, plain")
    # configs[4]: 64^3 with the fused 3-D augmentation (reference distributions, augment.random_affines)
    mid = pkg.voxelize(td, to, th).mid_p.cpu().numpy()
    xf = torch.from_numpy(pkg.augment.random_affines(mid, rng=np.random.RandomState(2026))[0]).to(dev)
    resident("configs[4]_aug_r64", td, to, th, 64, 10, offsets,
             fn=lambda out: pkg.voxelize_aug(td, to, th, xf, res=64, out=out),
             note="BASELINE configs[4]: 1024 full frames -> 64^3 with the 3-D augmentation fused into the kernel")
    # MSRA-like crops: 2,048 distinct seeded crops, repeated to the stated counts
    crops = [synth.synth_frame(100000 + i, "crop") for i in range(2048)]
    base = pkg.packing.pack_frames(crops)

    def tiled(n):
        reps = (n + 2047) // 2048
        lens = np.tile(np.diff(base.offsets), reps)[:n]
        off = np.zeros(n + 1, np.int64)
        np.cumsum(lens, out=off[1:])
        return pkg.packing.PackedFrames(np.ascontiguousarray(np.tile(base.depth, reps)[: off[-1]]), off,
                                        np.ascontiguousarray(np.tile(base.headers, (reps, 1))[:n]))

    pk1 = tiled(1024)
    c1 = pk1.to_torch(dev)
    resident("crops_1024", c1[0], c1[1], c1[2], RES, 30, pk1.offsets, note="1024 MSRA-like crops (bbox side 90-160 px)")
    del c1
    # configs[3], one-GPU form: all nine subjects' worth of crops resident, one launch
    pk3 = tiled(76500)
    c3 = pk3.to_torch(dev)
    resident("configs[3]_one_gpu", c3[0], c3[1], c3[2], RES, 3, pk3.offsets,
             note="BASELINE configs[3] on ONE GPU: 76,500 MSRA-like crops resident, one launch (the 8-GPU form is "
                  "bench.py --gpus 8: frames shard by rank, no collective)")
    del c3, pk3
    torch.cuda.empty_cache()
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
                seen += batch.tsdf.shape[0]
            torch.cuda.synchronize()
            r.append(seen / (time.perf_counter() - t0))
        return r
    rl = pkg.ResidentLoader(ds, batch_size=1024, device=dev, shuffle=True)
    r1024 = epochs(rl, 5)
    r16 = epochs(pkg.ResidentLoader(ds, batch_size=16, device=dev, shuffle=True), 3)
    h1024 = epochs(pkg.VoxelLoader(ds, batch_size=1024, device=dev, shuffle=True, max_pixels=1024 * 160 * 160), 3)
    ex["configs[2]_resident_shuffled"] = {
        "frames": 8500, "resident_bytes": rl.resident_bytes(),
        "batch_1024_crops_per_s": round(max(r1024[1:])), "batch_16_crops_per_s": round(max(r16[1:])),
        "host_fed_shuffled_batch_1024_crops_per_s": round(max(h1024[1:])),
        "what": "BASELINE configs[2] with the subject's pack resident in HBM (uploaded once; all of MSRA is 4.8 GB): shuffled "
                "batches drawn by index on the device, labels included (dataset.ResidentLoader); and the same shuffled "
                "batches through VoxelLoader, whose host gathers the crops before uploading them"}
    ra = epochs(pkg.ResidentLoader(ds, batch_size=1024, device=dev, shuffle=True, res=64, augment=True), 3)
    ex["configs[4]_resident_shuffled_aug_r64"] = {
        "frames": 8500, "batch": 1024, "crops_per_s": round(max(ra[1:])),
        "what": "BASELINE configs[4] as a training loop would run it: the subject resident in HBM, shuffled batches by index, a "
                "fresh reference-distribution 3-D augmentation per frame drawn on the host (numpy) and fused into the 64^3 "
                "voxelizer, mapped joints + labels from the same launch (dataset.ResidentLoader(augment=True, res=64))"}
    del rl
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
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the other configs / latency table (rank 0, N=1)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the voxelizer has no CPU path")
    # TSDF_BENCH_REHEARSAL=1: dry run of the N>1 code path on a box with fewer GPUs than ranks (ranks share
    # devices, the barrier / max-of-elapsed go over gloo instead of RCCL, the line says "rehearsal": true).
    # Never set by the driver; a rehearsal number is not a measurement.
    rehearsal = os.environ.get("TSDF_BENCH_REHEARSAL") == "1"
    dev_index = local_rank % torch.cuda.device_count() if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:
        # launched by torch.distributed.run: RCCL for the start/stop barrier + max of elapsed only
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    pkg = importlib.import_module("handposeestimation-with-3d-cnns_amd")
    synth = importlib.import_module("handposeestimation-with-3d-cnns_amd.synth")

    # this rank's shard: frames [rank*1024, (rank+1)*1024) of the seeded synthetic set
    depth, offsets, headers = synth.synth_batch(FRAMES_PER_GPU, "full", seed0=rank * FRAMES_PER_GPU)
    td, to, th = (torch.from_numpy(a).to(dev) for a in (depth, offsets, headers))
    out = pkg.voxelize(td, to, th, res=RES)  # allocates the outputs once
    torch.cuda.synchronize()
    assert bool((out.status == 0).all())

    def collective_barrier():
        dist.barrier() if rehearsal else dist.barrier(device_ids=[local_rank])

    node_barrier = NodeBarrier(dist, rank, world, collective_barrier) if dist is not None else None

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            node_barrier() if node_barrier.ok else collective_barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        pkg.voxelize(td, to, th, res=RES, out=out)
    # HIP events on the launch stream (torch's current stream): one pair around the K timed launches.
    # (Per-launch pairs were dropped from the timed region: every timestamped record costs ~3 us of
    # stream idle time between two 150 us kernels; they are taken in a separate pass below.)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    barrier()
    t0 = time.perf_counter()
    ev0.record()
    for k in range(args.steps):
        pkg.voxelize(td, to, th, res=RES, out=out)
    ev1.record()
    barrier()
    elapsed = time.perf_counter() - t0

    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # diagnostic pass (outside the timed region): per-launch event pairs -> spread of single launches
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(min(args.steps, 50))]
    for a, b in ev:
        a.record()
        pkg.voxelize(td, to, th, res=RES, out=out)
        b.record()
    torch.cuda.synchronize()

    kern_ms = np.array([a.elapsed_time(b) for a, b in ev])
    launch_ms = ev0.elapsed_time(ev1) / args.steps  # mean launch-to-launch time over the timed region
    abytes = algorithmic_bytes(offsets, FRAMES_PER_GPU, RES)

    if rank == 0:
        total_frames = world * FRAMES_PER_GPU * args.steps
        mean_ms = float(launch_ms)
        achieved = abytes / (mean_ms * 1e-3) / 1e9
        # HBM bytes per launch from the rocprofv3 PMC passes of this same command (separate FETCH_SIZE and
        # WRITE_SIZE runs, gfx950 corrections applied; tools/make_profiles.sh writes the file)
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "depth frames/sec to 32^3 TSDF",
            "value": round(total_frames / elapsed, 1),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "barrier": (("node (/dev/shm)" if node_barrier.ok else "collective") if dist is not None else "none (one process)"),
            "config": {
                "workload": "BASELINE configs[1]: batch 1024 synthetic 320x240 full-frame depth crops -> 32^3 "
                            "3-channel TSDF per GPU, inputs resident in HBM, one fused launch per step",
                "frames_per_gpu": FRAMES_PER_GPU, "res": RES, "layout": "czyx",
                "parallelism": f"frame-sharded x{world}, no collective",
            },
            "roofline": {
                "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "traffic_source": "profiles/pmc_traffic.json: rocprofv3 FETCH_SIZE/WRITE_SIZE passes of this command, "
                                  "not measured in this run",
                "frac_of_measured_copy": round(achieved / HBM_COPY_GBS, 4),
                "working_set_bytes": abytes,
                "kernel": "tsdf_fused_kernel<32, 0, false, false>", "algorithmic_bytes_per_launch": abytes,
                "launch_ms_mean": round(mean_ms, 4),
                "single_launch_ms_median": round(float(np.median(kern_ms)), 4),
                "single_launch_ms_min": round(float(kern_ms.min()), 4),
            },
        }
        if rehearsal:
            line["rehearsal"] = True
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(depth, offsets, headers)
        if world == 1 and not args.no_extras and not rehearsal:
            line["extras"] = extras(pkg, synth, dev, td, to, th, offsets)
            sc = stream_ceilings()
            if sc:
                line["extras"]["stream_ceilings"] = sc
                cp = max(sc.get("copy_nt", 0.0), sc.get("copy", 0.0))
                if cp > 0:   # interleaved reads and writes are the fair comparison for this kernel
                    line["roofline"]["frac_of_copy_stream_this_box"] = round(line["roofline"]["achieved"] / cp, 4)
                    if traffic:
                        line["roofline"]["traffic_rate_GBps"] = round(traffic / (mean_ms * 1e-3) / 1e9, 1)
                        line["roofline"]["traffic_rate_over_copy_stream"] = round(traffic / (mean_ms * 1e-3) / 1e9 / cp, 4)
        print(json.dumps(line), flush=True)

    if dist is not None:
        dist.barrier() if rehearsal else dist.barrier(device_ids=[local_rank])
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
