"""Frame-parallel sharding: frames of a stream go round-robin to ranks, no
data-path collective.  The N>1 bookkeeping of bench.py is exercised here with gloo on CPU
(world_size 2): every frame is decoded by exactly one rank, the per-rank results together
equal the single-rank result, and the throughput reduction is a MAX over ranks."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, zlib
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, os.path.join(%(root)r, "tests")); sys.path.insert(0, %(root)r)
import bench, oracle, vecgen
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
nframes = 6
mine = bench.shard_frames(nframes, rank, world)
orc = oracle.OracleDecoder()
crcs = torch.zeros(nframes, dtype=torch.int64)
for i in mine:
    data = vecgen.encode(vecgen.synth_image(96, 64, 3, seed=100 + i), mct=1, nlevels=3)
    info, planes, _ = orc.decode(data)          # stand-in for the device decode on this CPU-only box
    crcs[i] = oracle.framecrc(planes)
dist.all_reduce(crcs, op=dist.ReduceOp.SUM)       # test-only gather; the data path has no collective
t = torch.tensor([0.5 + rank], dtype=torch.float64)
tmax = bench.max_over_ranks(float(t.item()))
per_rank = bench.all_ranks(0.5 + rank)
if rank == 0:
    print("RESULT", ",".join(str(int(c)) for c in crcs), tmax, sorted(mine), "PER_RANK", per_rank)
dist.destroy_process_group()
'''


def test_round_robin_two_ranks_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29611", str(script)],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")][0].split()
    crcs = [int(x) for x in line[1].split(",")]
    # single-process reference
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    import vecgen
    orc = oracle.OracleDecoder()
    want = []
    for i in range(6):
        data = vecgen.encode(vecgen.synth_image(96, 64, 3, seed=100 + i), mct=1, nlevels=3)
        want.append(oracle.framecrc(orc.decode(data)[1]))
    assert crcs == want
    assert float(line[2]) == 1.5          # MAX over ranks of (0.5, 1.5)
    assert out.stdout.split("PER_RANK")[1].strip().startswith("[0.5, 1.5]")      # every rank's seconds, in rank order


def test_shard_frames_partition():
    sys.path.insert(0, ROOT)
    import bench
    for world in (1, 2, 4, 8):
        seen = []
        for r in range(world):
            seen += bench.shard_frames(240, r, world)
        assert sorted(seen) == list(range(240))
        assert bench.shard_frames(240, 0, world)[:2] == ([0, world] if world > 1 else [0, 1])


def test_bench_refuses_a_world_size_that_is_not_gpus():
    """bench.py under a launcher: one rank per GPU or nothing (checked before torch or the GPU is touched)"""
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29613")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode != 0
    assert "WORLD_SIZE is 3" in out.stderr
    assert not out.stdout.strip()                       # no JSON line from a run that was refused


def test_bench_self_launch_builds_a_one_rank_per_gpu_command(monkeypatch):
    """`bench.py --gpus N` without a launcher starts torch.distributed.run with N ranks on 127.0.0.1 and passes its own
    arguments on; its exit status is the launcher's"""
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    def fake_call(cmd, **kw):
        seen["cmd"] = cmd
        return 7
    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    try:
        bench.main()
        raise AssertionError("main() returned")
    except SystemExit as e:
        assert e.code == 7
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]


def _fake_sysfs(tmp_path, numa_of_gpu, cpus_of_node):
    """a /sys tree with one CPU node and len(numa_of_gpu) GPU nodes in the KFD topology"""
    nodes = tmp_path / "class/kfd/kfd/topology/nodes"
    (nodes / "0").mkdir(parents=True)
    (nodes / "0" / "properties").write_text("cpu_cores_count 64\nsimd_count 0\ndrm_render_minor 0\n")
    for g, numa in enumerate(numa_of_gpu):
        d = nodes / str(g + 1)
        d.mkdir()
        d.joinpath("properties").write_text("cpu_cores_count 0\nsimd_count 1024\ndrm_render_minor %d\n" % (128 + g))
        dev = tmp_path / ("class/drm/renderD%d/device" % (128 + g))
        dev.mkdir(parents=True)
        dev.joinpath("numa_node").write_text("%d\n" % numa)
    for n, cl in cpus_of_node.items():
        d = tmp_path / ("devices/system/node/node%d" % n)
        d.mkdir(parents=True)
        d.joinpath("cpulist").write_text(cl + "\n")
    return str(tmp_path)


def test_rank_is_bound_to_the_numa_node_of_its_gpu(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    have = sorted(os.sched_getaffinity(0))
    if len(have) < 8:
        import pytest
        pytest.skip("needs 8 CPUs in the affinity mask")
    lo, hi = have[:len(have) // 2], have[len(have) // 2:]
    as_list = lambda xs: ",".join(str(x) for x in xs)
    root = _fake_sysfs(tmp_path, [0, 0, 1, 1], {0: as_list(lo), 1: as_list(hi)})
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    assert [bench.gpu_numa_node(i, root) for i in range(4)] == [0, 0, 1, 1]
    assert bench.gpu_numa_node(4, root) is None
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "2,3")
    assert bench.gpu_numa_node(0, root) == 1
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    try:
        got = bench.bind_near_gpu(3, root)
        assert got == {"numa_node": 1, "bound": True, "cpus": len(hi)} and sorted(os.sched_getaffinity(0)) == hi
        os.sched_setaffinity(0, have)
        assert bench.bind_near_gpu(0, str(tmp_path / "nothing"))["bound"] is False      # no topology: leave the mask alone
        assert sorted(os.sched_getaffinity(0)) == have
    finally:
        os.sched_setaffinity(0, have)
