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


@pytest.mark.parametrize("world", [2])
def test_bench_multi_rank_path_runs_under_torchrun_with_gloo(tmp_path, world):
    """`python -m torch.distributed.run ... bench.py --gpus 2` exactly as the driver launches it, GPU work stubbed."""
    cpus = sorted(os.sched_getaffinity(0))
    half = max(1, len(cpus) // 2)
    b = _bench()
    gpus = [(0x05, 0, b._format_cpulist(cpus[:half])), (0x15, 1, b._format_cpulist(cpus[half:] or cpus[:half]))]
    _fake_sysfs(str(tmp_path / "sys"), gpus)
    env = dict(os.environ, TSDF_BENCH_DRYRUN="1", TSDF_BENCH_SYSFS=str(tmp_path / "sys"), OMP_NUM_THREADS="1")
    for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world),
           "--steps", "3", "--warmup", "1"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                     # rank 0 prints ONE line
    j = json.loads(lines[0])
    assert j["n_gpus"] == world and j["steps"] == 3 and j["warmup"] == 1 and j["scaling"] == "weak"
    assert j["rehearsal"] is True and "dry_run" in j and j["barrier"].startswith("node")
    assert j["unit"] == "frames/s" and j["metric"].startswith("depth frames/sec")
    assert j["dist_backend"] == "gloo" and j["dist_world_size"] == world   # (an RCCL run says "nccl")
    cfg = j["config"]
    assert cfg["frames_per_launch"] == 1024 and cfg["launches_per_step"] == 16
    # value is total frames over the slowest rank's wall time; ms_per_step * steps is that wall time
    total = world * 1024 * 16 * 3
    assert abs(j["value"] - total / (j["ms_per_step"] * 3e-3)) / j["value"] < 1e-3
    pr = j["per_rank"]
    for key in ("ms_per_step_events", "ms_per_step_host_wall", "frames_per_s_events", "frac_of_hbm_peak_events"):
        assert len(pr[key]) == world and all(v > 0 for v in pr[key])
    assert max(pr["ms_per_step_host_wall"]) == pytest.approx(j["ms_per_step"], rel=1e-3)
    aff = pr["affinity"]
    assert len(aff) == world and all(a["pinned"] for a in aff), aff
    assert aff[0]["numa_node"] == 0 and aff[1]["numa_node"] == 1 and aff[0]["pci"] == "0000:05:00.0"
    if len(cpus) >= 2:
        assert b._parse_cpulist(aff[0]["cpus"]).isdisjoint(b._parse_cpulist(aff[1]["cpus"]))
    c3 = j["extras"]["configs[3]_sharded"]
    assert c3["frames"] == 76500 and sum(c3["per_rank_frames"]) == 76500 and len(c3["per_rank_fps"]) == world
    px = np.array(c3["per_rank_pixels"], float)
    assert px.max() / px.mean() < 1.01                           # pixel-balanced shards
    assert c3["aggregate_fps"] > 0 and c3["imbalance"] >= 1.0
    assert "cpu_baseline" not in j                               # rank 0, N=1 only
