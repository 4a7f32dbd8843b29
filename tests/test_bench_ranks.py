"""bench.py's OWN multi-rank code path, driven on the CPU: torch.distributed.run with two
ranks over gloo executes bench.py itself — pinning, shard planning of BASELINE configs[3], the node barrier, the gathers
and the JSON line — with TSDF_BENCH_DRYRUN=1, which swaps the GPU launches for a host stub (nothing is computed; the line
is marked).  Plus the CPU-affinity planner on a fake sysfs tree of an 8-GPU, 2-socket node.
"""
import importlib.util
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _fake_sysfs(root, gpus, cpu_nodes=2):
    """KFD topology + PCI entries: `cpu_nodes` CPU nodes first (simd_count 0), then one node per GPU
    (bus, numa node, local cpulist)."""
    nodes = os.path.join(root, "class", "kfd", "kfd", "topology", "nodes")
    k = 0
    for _ in range(cpu_nodes):
        os.makedirs(os.path.join(nodes, str(k)))
        open(os.path.join(nodes, str(k), "properties"), "w").write("cpu_cores_count 64\nsimd_count 0\nlocation_id 0\ndomain 0\n")
        k += 1
    for bus, numa, cpus in gpus:
        os.makedirs(os.path.join(nodes, str(k)))
        open(os.path.join(nodes, str(k), "properties"), "w").write(
            f"cpu_cores_count 0\nsimd_count 1024\nlocation_id {bus << 8}\ndomain 0\n")
        pdir = os.path.join(root, "bus", "pci", "devices", f"0000:{bus:02x}:00.0")
        os.makedirs(pdir)
        open(os.path.join(pdir, "local_cpulist"), "w").write(cpus + "\n")
        open(os.path.join(pdir, "numa_node"), "w").write(f"{numa}\n")
        k += 1


def test_affinity_plan_on_an_eight_gpu_two_socket_node(tmp_path):
    b = _bench()
    gpus = [(0x05 + 0x10 * i, 0 if i < 4 else 1, "0-47,96-143" if i < 4 else "48-95,144-191") for i in range(8)]
    _fake_sysfs(str(tmp_path), gpus)
    topo = b.gpu_topology(str(tmp_path), env={})
    assert [t[1] for t in topo] == [0, 0, 0, 0, 1, 1, 1, 1] and topo[0][0] == "0000:05:00.0"
    assert topo[0][2] == set(range(0, 48)) | set(range(96, 144))
    allowed = set(range(192))
    plan = b.plan_affinity(topo, list(range(8)), allowed)
    # four ranks per socket, disjoint equal parts of that socket's cores, nothing from the other socket
    for r in range(8):
        assert len(plan[r]) == 24 and plan[r] <= topo[r][2]
        for q in range(r):
            assert not (plan[r] & plan[q])
    # a container that may only use a few cores: parts of < 2 cores are not split further
    plan = b.plan_affinity(topo, list(range(8)), {0, 1, 2, 50})
    assert plan[0] == {0, 1, 2} and plan[3] == {0, 1, 2} and plan[4] == {50}
    # no allowed core next to the GPU: the mask is left alone
    assert b.plan_affinity(topo, [0, 4], {1, 2, 3})[1] is None
    # visible-device lists reorder; anything unreadable gives up
    assert [t[0] for t in b.gpu_topology(str(tmp_path), env={"HIP_VISIBLE_DEVICES": "7,0"})] == ["0000:75:00.0", "0000:05:00.0"]
    assert [t[0] for t in b.gpu_topology(str(tmp_path), env={"ROCR_VISIBLE_DEVICES": "4,5,6", "HIP_VISIBLE_DEVICES": "2"})] \
        == ["0000:65:00.0"]
    assert b.gpu_topology(str(tmp_path), env={"HIP_VISIBLE_DEVICES": "GPU-deadbeef"}) == []
    assert b.gpu_topology(str(tmp_path / "nothing"), env={}) == []
    # a one-GPU container on that host: the other GPUs' nodes are listed but may not be opened (seen on the GPU box:
    # "Operation not permitted") — they are skipped, as HIP skips them
    import shutil
    for k in (2, 3, 4, 6, 7, 8, 9):
        f = tmp_path / "class" / "kfd" / "kfd" / "topology" / "nodes" / str(k) / "properties"
        os.remove(f)
        os.mkdir(f)          # opening it now fails with an OSError, whoever runs the test
    one = b.gpu_topology(str(tmp_path), env={})
    assert [t[0] for t in one] == ["0000:35:00.0"] and one[0][1] == 0
    assert b._format_cpulist({0, 1, 2, 5, 7, 8}) == "0-2,5,7-8" and b._parse_cpulist("0-2,5,7-8\n") == {0, 1, 2, 5, 7, 8}


def test_crop_lengths_mirror_the_generator(synth):
    """config3's shard planner derives every crop's size from the seed alone; it must stay in step with synth_frame."""
    b = _bench()
    lens = b.crop_lengths(synth)
    assert lens.size == b.ALL_SUBJECTS
    for i in (0, 1, 777, 2047):
        assert lens[i] == synth.synth_frame(100000 + i, "crop")[1].size == lens[i + 2048]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_bench(tmp_path, world, through_torchrun=True, extra_env=None, args=("--steps", "3", "--warmup", "1")):
    """bench.py with the GPU work stubbed (TSDF_BENCH_DRYRUN=1) over a fake sysfs tree with `max(world, 2)` GPUs on two
    sockets; returns (the parsed JSON line, the fake topology's (bus, numa, cpulist) rows)."""
    cpus = sorted(os.sched_getaffinity(0))
    half = max(1, len(cpus) // 2)
    b = _bench()
    ngpu = max(world, 2)
    gpus = [(0x05 + 0x10 * i, 0 if i < ngpu // 2 else 1,
             b._format_cpulist(cpus[:half] if i < ngpu // 2 else (cpus[half:] or cpus[:half]))) for i in range(ngpu)]
    sysfs = tmp_path / f"sys_{world}_{int(through_torchrun)}"
    _fake_sysfs(str(sysfs), gpus)
    env = dict(os.environ, TSDF_BENCH_DRYRUN="1", TSDF_BENCH_SYSFS=str(sysfs), OMP_NUM_THREADS="1")
    for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "TORCHELASTIC_RUN_ID", "RANK", "WORLD_SIZE",
              "LOCAL_RANK", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    env.update(extra_env or {})
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", str(world), *args]
    if through_torchrun:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
               "127.0.0.1", "--master-port", str(_free_port())] + tail
    else:
        cmd = [sys.executable] + tail
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                     # rank 0 prints ONE line
    return json.loads(lines[0]), gpus


@pytest.mark.parametrize("world", [2, 8])
def test_bench_multi_rank_path_runs_under_torchrun_with_gloo(tmp_path, world):
    """`python -m torch.distributed.run ... bench.py --gpus N` exactly as the driver launches it, GPU work stubbed: N = 2,
    and N = 8 — the size of the node the scaling curve will be taken on (this container has 8 cores)."""
    b = _bench()
    j, gpus = _run_bench(tmp_path, world)
    assert j["n_gpus"] == world and j["steps"] == 3 and j["warmup"] == 1 and j["scaling"] == "weak"
    assert j["rehearsal"] is True and "dry_run" in j and j["barrier"].startswith("node")
    assert j["unit"] == "frames/s" and j["metric"].startswith("depth frames/sec")
    assert j["dist_backend"] == "gloo" and j["dist_world_size"] == world   # (an RCCL run says "nccl")
    cfg = j["config"]
    assert cfg["frames_per_launch"] == 1024 and cfg["launches_per_step"] == 16 and cfg["batches_in_rotation"] == b.ROTATION
    assert cfg["parallelism"] == f"frame-sharded x{world}, no collective"
    # value is total frames over the slowest rank's wall time; ms_per_step * steps is that wall time
    total = world * 1024 * 16 * 3
    assert abs(j["value"] - total / (j["ms_per_step"] * 3e-3)) / j["value"] < 1e-3
    pr = j["per_rank"]
    for key in ("ms_per_step_events", "ms_per_step_host_wall", "frames_per_s_events", "frac_of_hbm_peak_events"):
        assert len(pr[key]) == world and all(v > 0 for v in pr[key])
    assert max(pr["ms_per_step_host_wall"]) == pytest.approx(j["ms_per_step"], rel=1e-3)
    # every rank draws its own frames: `world` distinct first seeds, ROTATION batches each, no seed shared by two ranks
    seeds = pr["batch_seed0"]
    assert len(seeds) == world and all(len(s) == b.ROTATION for s in seeds)
    flat = [v for s in seeds for v in s]
    assert len(set(flat)) == world * b.ROTATION and min(np.diff(sorted(flat))) >= 1024
    assert [s[0] for s in seeds] == [r * 1024 for r in range(world)]
    aff = pr["affinity"]
    assert len(aff) == world and all(a["pinned"] for a in aff), aff
    assert [a["numa_node"] for a in aff] == [g[1] for g in gpus[:world]] and aff[0]["pci"] == "0000:05:00.0"
    assert len({a["pci"] for a in aff}) == world                # one GPU each
    if len(os.sched_getaffinity(0)) >= 2:
        assert b._parse_cpulist(aff[0]["cpus"]).isdisjoint(b._parse_cpulist(aff[-1]["cpus"]))
    rf = j["roofline"]
    assert rf["rotation"] == b.ROTATION and rf["working_set_bytes"] >= 4e9 and rf["frac_same_batch"] > 0
    c3 = j["extras"]["configs[3]_sharded"]
    assert c3["frames"] == 76500 and sum(c3["per_rank_frames"]) == 76500 and len(c3["per_rank_fps"]) == world
    assert len(set(c3["per_rank_first_frame"])) == world and c3["per_rank_first_frame"][0] == 0   # distinct shards
    px = np.array(c3["per_rank_pixels"], float)
    assert px.max() / px.mean() < 1.01                           # pixel-balanced shards
    assert c3["aggregate_fps"] > 0 and c3["imbalance"] >= 1.0
    assert "cpu_baseline" not in j                               # rank 0, N=1 only


def _keys(o, depth=2):
    """Key tree of a JSON object down to `depth` levels."""
    if not isinstance(o, dict) or depth == 0:
        return None
    return {k: _keys(v, depth - 1) for k, v in o.items()}


def test_one_rank_through_torchrun_prints_the_plain_line(tmp_path):
    """A SCALE run's N=1 point is launched through torch.distributed.run (world 1, TORCHELASTIC_RUN_ID set), the BENCH
    run as plain `python bench.py`: both must print the same line — same keys, `roofline` and `cpu_baseline` included —
    or the two cannot be compared."""
    plain, _ = _run_bench(tmp_path, 1, through_torchrun=False)
    tr, _ = _run_bench(tmp_path, 1, through_torchrun=True)
    assert plain["dist_backend"] is None and plain["barrier"].startswith("none")
    assert tr["dist_backend"] == "gloo" and tr["dist_world_size"] == 1 and tr["barrier"].startswith("node")
    for j in (plain, tr):
        assert j["n_gpus"] == 1 and "roofline" in j and "cpu_baseline" in j and "configs[3]_sharded" in j["extras"]
        assert j["cpu_baseline"]["kind"] == "port" and j["cpu_baseline"]["cores"] >= 1
        assert j["cpu_baseline"]["value"] == max(r["frames_per_s"] for r in j["cpu_baseline"]["legs"])
        assert j["extras"]["configs[3]_sharded"]["per_rank_frames"] == [76500]
    kp, kt = _keys(plain), _keys(tr)
    assert kp == kt, (kp, kt)
    assert _keys(plain["roofline"], 1) == _keys(tr["roofline"], 1)


def test_cpu_baseline_thread_plan(tmp_path):
    """The CPU baseline's thread counts: the cgroup quota when there is one, the affinity mask otherwise, 16 beside a mask
    wider than 16 (round 4: 256 threads on a 16-CPU share ran 5x slower than 16 and were reported as the baseline)."""
    b = _bench()
    assert b.plan_cpu_legs(8, None) == ([1, 8], 8, "affinity mask (no cgroup quota)")
    assert b.plan_cpu_legs(256, None)[0] == [1, 16, 256]
    assert b.plan_cpu_legs(256, 16.0) == ([1, 16], 16, "cgroup quota")
    assert b.plan_cpu_legs(4, 2.5)[0] == [1, 3] and b.plan_cpu_legs(1, None)[0] == [1]
    assert b.plan_cpu_legs(8, 64.0)[1] == 8                      # never more threads than the mask allows
    # cgroup v2, nested: the tightest limit on the way up counts
    root = tmp_path / "cg2"
    (root / "a" / "b").mkdir(parents=True)
    (root / "cpu.max").write_text("max 100000\n")
    (root / "a" / "cpu.max").write_text("1600000 100000\n")
    (root / "a" / "b" / "cpu.max").write_text("max 100000\n")
    pc = tmp_path / "proc_cgroup2"
    pc.write_text("0::/a/b\n")
    q, where = b.cgroup_cpu_quota(str(root), str(pc))
    assert q == 16.0 and where.endswith("a/cpu.max")
    # cgroup v1
    root = tmp_path / "cg1"
    (root / "cpu" / "job").mkdir(parents=True)
    (root / "cpu" / "cpu.cfs_quota_us").write_text("-1\n")
    (root / "cpu" / "cpu.cfs_period_us").write_text("100000\n")
    (root / "cpu" / "job" / "cpu.cfs_quota_us").write_text("250000\n")
    (root / "cpu" / "job" / "cpu.cfs_period_us").write_text("100000\n")
    pc = tmp_path / "proc_cgroup1"
    pc.write_text("3:cpuset:/jobs\n1:cpu,cpuacct:/job\n0::/\n")
    assert b.cgroup_cpu_quota(str(root), str(pc))[0] == 2.5
    pc.write_text("1:cpu,cpuacct:/\n")
    q, where = b.cgroup_cpu_quota(str(root), str(pc))
    assert q is None and where.startswith("no limit set")
    assert b.cgroup_cpu_quota(str(tmp_path / "none"), str(tmp_path / "nope")) == (None, "no cgroup CPU controller file readable")
