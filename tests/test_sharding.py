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
if rank == 0:
    print("RESULT", ",".join(str(int(c)) for c in crcs), tmax, sorted(mine))
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


def test_shard_frames_partition():
    sys.path.insert(0, ROOT)
    import bench
    for world in (1, 2, 4, 8):
        seen = []
        for r in range(world):
            seen += bench.shard_frames(240, r, world)
        assert sorted(seen) == list(range(240))
        assert bench.shard_frames(240, 0, world)[:2] == ([0, world] if world > 1 else [0, 1])
