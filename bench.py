#!/usr/bin/env python3
"""bench.py -- Mpixel/s of the MI355X-native HTJ2K decode hot path on BASELINE.json configs[1]
(3840x2160 RGB 8-bit, lossless 5/3 + RCT, 64x64 codeblocks, 5 levels), device-resident.

  python bench.py --gpus N --steps K --warmup W
  (N > 1 without a launcher: the script starts `python -m torch.distributed.run --nproc-per-node N` itself, one rank
   per GPU; under a launcher it reads RANK / LOCAL_RANK / WORLD_SIZE and checks WORLD_SIZE == N.)

A "step" is one pass of the hot path -- HT block decode + dequantisation, inverse DWT, inverse
RCT + level shift + clip + rgb24 pack (tile_codeblocks() + the tail of jpeg2000_decode_tile(),
libavcodec/jpeg2000dec.c:2212-2395) -- over one batch of BATCH synthetic 4K frames whose
compressed codeblock bytes and descriptors are already resident in HBM when the timed region
starts; decoded frames stay in HBM.  Host marker/Tier-2 parsing and PCIe are outside the
timed region (their rates are reported as extra fields, never as `value`).  After the timed region
the first and the last frame of the batch are downloaded and compared with their source images and with
the oracle's framecrc (`parity_checked`), and no block may have been rejected.

Frames of a stream are independent, so ranks shard them round-robin with no data-path
collective ("weak" scaling: every rank decodes its own batch per step).

The JSON line also carries
  roofline      the IDWT launches of the timed region (the HBM-bound stage BASELINE.json's north_star names), timed
                live with HIP events on the kernels' own stream:
                  achieved / frac   bytes that have to cross HBM (what the kernels read and write at the least: 16-bit
                                    sub-bands and LL bands where the job keeps them so, 1 byte per sample out of the
                                    final level that is fused with RCT + rgb24 store) / launch time, against 8 TB/s
                  algorithmic_GBps  SURVEY 8(d)'s figure (4 B read + 4 B written per sample and level) / launch time --
                                    a rate for comparison with the CPU figures, not a fraction of the bus
                  traffic           HBM bytes per launch from the PMC passes committed under profiles/
  cpu_baseline           the CPU oracle (single-thread C restatement of the reference decoder incl. its host parsing:
                         kind "port") on a bounded sample of the same frames
  cpu_baseline_hot_path  the same without host parsing: block decode + IDWT + MCT/write only -- like for like with `value`
  cpu_baseline_all_cores frame-parallel oracle decodes on all host cores (the reference's frame threads,
                         libavcodec/pthread_frame.c:856-889)
  stream240     BASELINE configs[4] end to end: 240 distinct 4K 10-bit RGB frames, packets in host memory -> rgb48
                frames in page-locked host memory through the asynchronous pipeline, sharded round-robin over the
                ranks; host parsing and PCIe included (never `value`)
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

WIDTH, HEIGHT, NCOMP = 3840, 2160, 3
NLEVELS, CB = 5, (6, 6)
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
COPY_CEILING_GBS = 6290.0      # same guide, line 36: float4 copy, 79 % of the spec


def committed_traffic(frames_per_step, launches_per_step=None):
    """HBM bytes per IDWT launch from the PMC passes committed under profiles/ (rocprofv3 cannot run inside this
    process): (2 * FETCH_SIZE + WRITE_SIZE) * 1024 averaged over the IDWT launches of one step of this same
    workload (tools/prof_round.sh + tools/make_profiles.py); None when the batch, or the number of IDWT launches a
    step makes, differs from the profiled one.  The newest matching file wins."""
    best = None
    pdir = os.path.join(ROOT, "profiles")
    if os.path.isdir(pdir):
        for name in sorted(os.listdir(pdir)):
            if name.endswith("_idwt_traffic.json"):
                try:
                    d = json.load(open(os.path.join(pdir, name)))
                    lps = d.get("idwt_launches_per_step")
                    if d.get("frames_per_step") == frames_per_step and (
                            launches_per_step is None or lps is None or abs(lps - launches_per_step) < 1e-6):
                        best = (d["hbm_bytes_per_launch"], name)
                except (OSError, ValueError, KeyError):
                    pass
    return best


def shard_frames(nframes, rank, world):
    """frame i -> rank i mod world"""
    return list(range(rank, nframes, world))


def max_over_ranks(seconds):
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return seconds
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([seconds], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_ranks(value):
    """[value of rank 0, value of rank 1, ...] on every rank (a SUM all-reduce of one-hot vectors: the timing
    bookkeeping, not a data-path collective)"""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [float(value)]
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.zeros(dist.get_world_size(), dtype=torch.float64, device=dev)
    t[dist.get_rank()] = float(value)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(x) for x in t.tolist()]


def _cpulist(text):
    out = set()
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.update(range(int(a), int(b or a) + 1))
    return out


def gpu_numa_node(device_index, sysfs="/sys"):
    """NUMA node of HIP device `device_index`, read from sysfs without touching the GPU: the KFD topology lists the
    GPU nodes in the order the runtime enumerates them (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES index lists are
    honoured), each with its DRM render minor, whose PCI device has a numa_node.  None when any of it is missing."""
    try:
        base = os.path.join(sysfs, "class/kfd/kfd/topology/nodes")
        gpus = []
        for name in sorted(os.listdir(base), key=int):
            props = dict(l.split()[:2] for l in open(os.path.join(base, name, "properties")) if len(l.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                gpus.append(int(props["drm_render_minor"]))
        for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
            vis = os.environ.get(var)
            if vis:
                gpus = [gpus[int(x)] for x in vis.split(",") if x.strip().isdigit() and int(x) < len(gpus)]
        minor = gpus[device_index]
        node = int(open(os.path.join(sysfs, "class/drm/renderD%d/device/numa_node" % minor)).read())
        return node if node >= 0 else None
    except (OSError, ValueError, IndexError, KeyError):
        return None


def bind_near_gpu(device_index, sysfs="/sys"):
    """Before the first GPU call: restrict this process (and the threads it will start -- parser workers, staging
    copies, the pipeline) to the CPUs of the NUMA node its GPU hangs off, so that page-locked buffers are first
    touched there.  A plain sched_setaffinity call, nothing is re-executed.  Returns what was done, for the JSON line."""
    try:
        have = os.sched_getaffinity(0)
    except AttributeError:
        return {"numa_node": None, "bound": False}
    node = gpu_numa_node(device_index, sysfs)
    if node is None:
        return {"numa_node": None, "bound": False, "cpus": len(have)}
    try:
        near = _cpulist(open(os.path.join(sysfs, "devices/system/node/node%d/cpulist" % node)).read()) & have
    except (OSError, ValueError):
        near = set()
    if len(near) < 4 or near == have:                     # nothing to gain, or a cgroup that put us elsewhere
        return {"numa_node": node, "bound": False, "cpus": len(have)}
    try:
        os.sched_setaffinity(0, near)
    except OSError as e:                                  # a container that does not allow it: run unbound, say so
        return {"numa_node": node, "bound": False, "cpus": len(have), "error": str(e)}
    return {"numa_node": node, "bound": True, "cpus": len(near)}


def make_streams(nstreams, rank, with_images=False):
    """distinct synthetic 4K RGB frames (BASELINE.md section 3 generator) -> HTJ2K codestreams"""
    import vecgen
    out, imgs = [], []
    for i in range(nstreams):
        img = vecgen.synth_image(WIDTH, HEIGHT, NCOMP, depth=8, seed=2 + i + 16 * rank, noise=8)
        out.append(vecgen.encode(img, mct=1, nlevels=NLEVELS, cb=CB, transform=1))
        imgs.append(img)
    return (out, imgs) if with_images else out


def host_cores():
    """cores this process may really use: the affinity mask, cut down to the cgroup's CPU quota where there is one
    (a GPU box gives a job a share of the host, not the whole machine)"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period) + 0.5)))
    except (OSError, ValueError):
        pass
    try:                                                   # this pool's boxes give a job 16 host cores per GPU
        import torch
        n = min(n, 16 * max(1, torch.cuda.device_count()))
    except Exception:                                      # noqa: BLE001
        pass
    return n


def cpu_baselines(streams, budget_s=10.0):
    """the CPU oracle on a bounded sample of the bench's frames, on this box's host cores: whole decodes on one
    thread, the hot path alone (host parsing excluded) on one thread, whole decodes on all cores"""
    import threading
    import oracle
    orc = oracle.OracleDecoder()
    orc.decode(streams[0])                      # warm caches / page in
    n, t0 = 0, time.perf_counter()
    while True:
        orc.decode(streams[n % len(streams)])
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 64:
            break
    one = {"value": round(n * WIDTH * HEIGHT / el / 1e6, 3), "unit": "Mpixel/s", "cores": 1, "kind": "port",
           "sample": "%d decodes of the bench's 4K RGB lossless frames in %.1f s, single thread, "
                     "oracle/ (C restatement of the reference decoder incl. its host parsing)" % (n, el)}
    # hot path only: parse outside the clock, then blocks + IDWT + MCT/write_frame
    n2, hot, t_parse = 0, 0.0, 0.0
    while hot < budget_s * 0.6 and n2 < 32:
        t0 = time.perf_counter()
        orc.parse(streams[n2 % len(streams)])
        t1 = time.perf_counter()
        orc.decode_parsed()
        orc.idwt()
        orc.write()
        hot += time.perf_counter() - t1
        t_parse += t1 - t0
        n2 += 1
    hot_path = {"value": round(n2 * WIDTH * HEIGHT / hot / 1e6, 3), "unit": "Mpixel/s", "cores": 1, "kind": "port",
                "sample": "%d frames in %.1f s: tile_codeblocks() (HT block decode + dequantisation + IDWT) + mct_decode + "
                          "write_frame of the oracle, host parsing (%.1f ms per frame) not counted -- the stages `value` times"
                          % (n2, hot, t_parse / max(n2, 1) * 1e3)}
    orc.close()
    # all cores: one oracle decoder per thread (ctypes releases the GIL inside the library), frames handed out round-robin
    cores = host_cores()
    counts, stop_at = [0] * cores, time.perf_counter() + budget_s * 0.8

    def worker(k):
        d = oracle.OracleDecoder()
        i = k
        while time.perf_counter() < stop_at:
            d.decode(streams[i % len(streams)])
            counts[k] += 1
            i += cores
        d.close()
    t0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(k,)) for k in range(cores)]
    for t in th: t.start()
    for t in th: t.join()
    el = time.perf_counter() - t0
    allc = {"value": round(sum(counts) * WIDTH * HEIGHT / el / 1e6, 3), "unit": "Mpixel/s", "cores": cores, "kind": "port",
            "sample": "%d whole decodes in %.1f s on %d threads, one frame per thread at a time (frame threading)" % (sum(counts), el, cores)}
    return one, hot_path, allc


def part1_leg(dec, nframes, with_cpu):
    """the bench's 4K RGB frames coded with Part-1 (MQ) codeblocks, default mode switches, device-resident like `value`"""
    import ffmpeg_ht_amd
    import vecgen
    srcs = [vecgen.encode(vecgen.synth_image(WIDTH, HEIGHT, NCOMP, depth=8, seed=2 + i, noise=8), mct=1, nlevels=NLEVELS,
                          cb=CB, transform=1, part1=True) for i in range(2)]
    pk = [ffmpeg_ht_amd.packet(d) for d in srcs]
    job = dec.job()
    job.parse_batch([pk[i % 2] for i in range(nframes)]).upload().run().wait()
    nblocks = job.num_blocks()
    steps = 3
    t0 = time.perf_counter()
    for _ in range(steps):
        job.run()
    job.wait()
    dt = (time.perf_counter() - t0) / steps
    ht_ms = job.stage_ms()[0]
    job.free()
    out = {"Mpixel_s": round(nframes * WIDTH * HEIGHT / dt / 1e6, 1), "frames_per_step": nframes,
           "ms_per_step": round(dt * 1e3, 2), "k_mq_decode_ms": round(ht_ms, 2),
           "codeblocks_per_s": round(nblocks / dt), "code_MB_per_s": round(sum(len(srcs[i % 2]) for i in range(nframes)) / dt / 1e6, 1),
           "workload": "configs[1] geometry with Part-1 blocks: 3840x2160 RGB 8-bit lossless 5/3 + RCT, 64x64, 5 levels, "
                       "one quality layer, no mode switches"}
    if with_cpu:
        import oracle
        orc = oracle.OracleDecoder()
        t0 = time.perf_counter()
        orc.decode(srcs[0])
        out["cpu_oracle_Mpixel_s_1core"] = round(WIDTH * HEIGHT / (time.perf_counter() - t0) / 1e6, 2)
        orc.close()
    return out


def stream240_leg(dec, m, rank, world, nframes, barrier, min_frames_per_rank=1200, nwarm_distinct=6):
    """BASELINE configs[4]: a stream of `nframes` distinct 4K 10-bit lossless RGB frames (-> rgb48, samples << 6), frame
    i on rank i mod world, every rank's share through the asynchronous pipeline: packets in pageable host memory ->
    frames in page-locked host memory.  Host parsing, staging, PCIe both ways and the kernels are all inside the clock.
    A rank cycles through its share until it has decoded at least `min_frames_per_rank` frames (30 frames per rank at
    N = 8 would be 30 ms of clock: thread wake-ups, not throughput), and the pipeline is warmed up on OTHER frames than
    the timed ones.  Returns (seconds on this rank, frames this rank decoded inside the clock, parity, ...)."""
    import concurrent.futures
    import numpy as np
    import vecgen
    mine = shard_frames(nframes, rank, world)
    workers = max(1, min(16, host_cores() // (1 if os.environ.get("HTJ2K_BENCH_BOUND") == "1" else max(world, 1))))

    def make(i):                                           # ctypes releases the GIL inside the encoder
        img = vecgen.synth_image(WIDTH, HEIGHT, NCOMP, depth=10, seed=1000 + i, noise=20)
        data = vecgen.encode(img, depth=10, mct=1, nlevels=NLEVELS, cb=CB)
        return data, (img if i in (mine[0], mine[-1]) else None)
    t0 = time.perf_counter()
    with concurrent.futures.ThreadPoolExecutor(workers) as ex:
        made = list(ex.map(make, mine))
        warm_made = list(ex.map(make, [100000 + nwarm_distinct * rank + k for k in range(nwarm_distinct)]))
    t_enc = time.perf_counter() - t0
    keep = {mine[k]: im for k, (_, im) in enumerate(made) if im is not None}
    nbytes = sum(len(d) for d, _ in made)
    info0 = dec.probe(made[0][0])
    pkts = [m.packet(d) for d, _ in made]                  # padded buffers, sent by reference (htj2k_pipe_send_ref)
    warm_pkts = [m.packet(d) for d, _ in warm_made]
    del made, warm_made
    reps = max(1, -(-min_frames_per_rank // len(pkts)))
    total = reps * len(pkts)
    pinned, ptrs = dec.alloc_frame_pinned(info0)
    first = last = None
    pipe = dec.pipe(batch=8, depth=3)
    # warm-up outside the clock on frames that are not part of the stream: buffers of all jobs get allocated
    warm = 24
    sent = got = 0
    while got < warm:
        while sent < warm and pipe.send(warm_pkts[sent % len(warm_pkts)]):
            sent += 1
        if sent == warm:
            pipe.flush()
        pipe.receive(into=pinned)
        got += 1
    barrier()
    t0 = time.perf_counter()
    sent = got = 0
    while got < total:
        while sent < total and pipe.send(pkts[sent % len(pkts)]):
            sent += 1
        if sent == total:
            pipe.flush()
        if pipe.receive(into=pinned) is None:
            break
        if got == 0:
            first = pinned[0][0].copy()
        if got == total - 1:
            last = pinned[0][0].copy()
        got += 1
    dt = time.perf_counter() - t0
    barrier()
    pipe.close()
    ok = got == total
    for fr, idx in ((first, mine[0]), (last, mine[-1])):
        img = keep.get(idx)
        ok = ok and fr is not None and img is not None and \
            np.array_equal(fr.view(np.uint16).reshape(HEIGHT, WIDTH, 3) >> 6, np.stack(img, -1))
    dec.free_frame_pinned(ptrs)
    return dt, total, ok, t_enc, nbytes / len(pkts), reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=128, help="4K frames per step and per GPU (about 48 GB of HBM at 128 with the one-job copy of the roofline pass; "
                    "measured on one box, two jobs: 48 frames 148, 64 156, 96 162, 128 167 Gpixel/s -- the block decoder's launches balance "
                    "better over the SIMDs the more waves they have; the IDWT does not care)")
    ap.add_argument("--jobs", type=int, default=2, help="the batch is split over this many jobs on HIP streams of their own, as the "
                    "frame pipeline keeps several jobs in flight: the instruction-bound HT kernels of one job run beside the "
                    "bandwidth-bound IDWT launches of the other.  `roofline` always comes from a separate pass of ONE job "
                    "holding the whole batch, where every IDWT launch has the chip to itself")
    ap.add_argument("--distinct", type=int, default=4, help="distinct synthetic frames per rank (cycled to fill the batch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--part1", type=int, default=0, metavar="FRAMES",
                    help="also time the same 4K frames coded with Part-1 (MQ) blocks: FRAMES per step through k_mq_decode "
                         "(rank 0, reported as the extra object \"part1\"; SURVEY 8f rank 3, not part of `value`)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the host-side legs after the timed region (profiling runs: "
                    "every kernel launch in the trace is then a batch launch)")
    ap.add_argument("--stream240", type=int, default=240, metavar="FRAMES",
                    help="frames of the configs[4] end-to-end leg (all ranks; 0 = skip; skipped with --no-e2e)")
    ap.add_argument("--stream240-min-frames", type=int, default=1200, metavar="N", dest="stream240_min_frames",
                    help="every rank cycles through its share of the stream240 frames until it has decoded this many "
                         "(>= 1 s of clock per rank)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: become one.  Nothing has touched the GPU yet (not even `import torch`), the ranks are children
        # of this process and its exit status is theirs.
        port = 29500 + os.getpid() % 2000
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE is %d: launch one rank per GPU" % (args.gpus, world))
    # rehearsal knobs (never set by the driver): several ranks on the one GPU of a test box, over gloo
    backend = os.environ.get("HTJ2K_BENCH_BACKEND", "nccl")
    device = int(os.environ["HTJ2K_BENCH_DEVICE"]) if "HTJ2K_BENCH_DEVICE" in os.environ else (local_rank if world > 1 else 0)
    # before anything touches the GPU (or allocates page-locked memory): stay on the CPUs next to this rank's GPU
    affinity = bind_near_gpu(device) if world > 1 and os.environ.get("HTJ2K_BENCH_BIND", "1") != "0" else {"bound": False}
    if affinity.get("bound"):
        os.environ["HTJ2K_BENCH_BOUND"] = "1"
    import numpy as np
    import torch
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(device)
        dist.init_process_group(backend)
    import ffmpeg_ht_amd as m
    import oracle

    streams, images = make_streams(min(args.distinct, args.batch), rank, with_images=True)
    batch = [streams[i % len(streams)] for i in range(args.batch)]
    njobs = max(1, min(args.jobs, args.batch))
    per_job = [batch[i::njobs] for i in range(njobs)]

    dec = m.Decoder(device_id=device)
    per_job = [[m.packet(x) for x in b] for b in per_job]     # padded packet buffers, built once
    jobs = [dec.job().parse_batch(b) for b in per_job]         # cold: allocates the pinned staging buffers
    t0 = time.perf_counter()
    for job, b in zip(jobs, per_job):
        job.parse_batch(b)
    t_parse = time.perf_counter() - t0
    host_ms = [job.host_ms() for job in jobs]
    for job in jobs:                                       # cold: allocates the device buffers
        job.upload()
    for job in jobs:
        job.wait()
    t0 = time.perf_counter()
    for job in jobs:
        job.upload()
    for job in jobs:
        job.wait()
    t_upload = time.perf_counter() - t0
    nblocks = sum(job.num_blocks() for job in jobs)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        for job in jobs:
            job.run(7)
    for job in jobs:
        job.wait()
    barrier()
    follow = os.environ.get("HTJ2K_BENCH_FOLLOW", "1") != "0"
    t0 = time.perf_counter()
    for _ in range(args.steps):
        for job in jobs:
            job.run(7)
        if follow:
            for job in jobs:
                job.stage_ms()          # the host follows the streams step by step (reads the step's events)
    for job in jobs:
        job.wait()
    barrier()
    elapsed_here = time.perf_counter() - t0
    elapsed = max_over_ranks(elapsed_here)
    rank_seconds = all_ranks(elapsed_here)

    frames_total = args.steps * args.batch * world
    value = frames_total * WIDTH * HEIGHT / elapsed / 1e6

    # parity of what was timed: the first and the last frame of the batch, as the last timed step left them in HBM,
    # against their source images (the streams are lossless) and against the oracle's framecrc; no block rejected
    parity = True
    block_errors = sum(job.block_errors() for job in jobs)
    orc = oracle.OracleDecoder()
    crc_of_stream = {}
    for jix, fix in ((0, 0), (njobs - 1, len(per_job[-1]) - 1)):
        k = (fix * njobs + jix) % len(streams)              # which distinct stream that frame is
        _, planes = jobs[jix].download_frame(fix)
        got = planes[0].reshape(HEIGHT, WIDTH, NCOMP)
        if k not in crc_of_stream:
            crc_of_stream[k] = oracle.framecrc(orc.decode(streams[k])[1])
        parity = parity and bool(np.array_equal(got, np.stack(images[k], -1))) and oracle.framecrc(planes) == crc_of_stream[k]
    orc.close()
    parity = parity and block_errors == 0
    if not parity:
        sys.exit("bench.py: the timed frames do not match their sources / the oracle (block errors: %d)" % block_errors)

    # `roofline` pass: the whole batch as ONE job, nothing else on the chip, per-launch HIP events on the job's stream
    def measure_one_job(job, steps):
        acc = dict(ht=0.0, idwt=0.0, pack=0.0, lms=0.0, lby=0.0, lhb=0.0, n=0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            job.run(7)
            a, b, c = job.stage_ms()
            acc["ht"] += a; acc["idwt"] += b; acc["pack"] += c
            for (ms, by), hb in zip(job.idwt_launches(), job.idwt_hbm_bytes()):
                acc["lms"] += ms; acc["lby"] += by; acc["lhb"] += hb; acc["n"] += 1
        job.wait()
        torch.cuda.synchronize()
        acc["seconds"] = time.perf_counter() - t0
        acc["steps"] = steps
        return acc
    if njobs == 1:
        rjob, own_rjob = jobs[0], False
    else:
        rjob, own_rjob = dec.job().parse_batch([m.packet(x) for x in batch]).upload(), True
        for _ in range(2):
            rjob.run(7)
        rjob.wait()
    rf = measure_one_job(rjob, max(5, args.steps // 2))
    one_job = {"value": round(rf["steps"] * args.batch * WIDTH * HEIGHT / rf["seconds"] / 1e6, 2), "unit": "Mpixel/s",
               "steps": rf["steps"], "ms_per_step": round(rf["seconds"] / rf["steps"] * 1e3, 4),
               "note": "the batch as one job, stages strictly one after the other: the pass `roofline` and the stage times come from"}
    c16, ll16 = bool(rjob.coef16()), rjob.ll16() == 1
    if own_rjob:
        rjob.free()
    copy_measured = dec.copy_bench(512, 10) if rank == 0 else 0.0
    ht_ms, idwt_ms, pack_ms = rf["ht"], rf["idwt"], rf["pack"]
    idwt_launch_ms, idwt_launch_bytes, idwt_launch_hbm, nlaunch = rf["lms"], rf["lby"], rf["lhb"], rf["n"]
    rsteps = rf["steps"]

    # end-to-end rate of one frame through the plain htj2k_decode() entry (parse + H2D + kernels + D2H)
    n_e2e = 0 if (args.no_e2e or rank != 0 or world > 1) else 8       # the host-side legs: rank 0 of the N = 1 run only
    e2e = 0.0
    if n_e2e:
        pk1 = [m.packet(x) for x in streams]
        buf1 = m.alloc_frame(dec.probe(streams[0]))
        dec.decode_into(pk1[0], buf1)                     # first call allocates the context's own job
        t0 = time.perf_counter()
        for i in range(n_e2e):
            dec.decode_into(pk1[i % len(pk1)], buf1)
        e2e = n_e2e * WIDTH * HEIGHT / (time.perf_counter() - t0) / 1e6

    # the asynchronous pipeline (htj2k_pipe_*): packets in host memory -> frames in host memory, with host
    # parsing (several threads), H2D, kernels and D2H of different batches overlapping
    pipe_rate = pipe_rate_pinned = pipe_rate_device = 0.0
    if not args.no_e2e and rank == 0 and world == 1:
        info0 = dec.probe(streams[0])
        pk = [m.packet(x) for x in streams]                # zero-copy sends (htj2k_pipe_send_ref): the staging copy is the workers'

        def run_pipe(receive, nwarm=48, nfr=96):        # 2 * depth - 1 = 5 jobs of 8 frames exist: all are allocated by frame 40
            pipe = dec.pipe(batch=8, depth=3)
            sent = got = 0
            t0 = None
            while got < nwarm + nfr:
                while sent < nwarm + nfr and pipe.send(pk[sent % len(pk)]):
                    sent += 1
                if sent == nwarm + nfr:
                    pipe.flush()
                if receive(pipe) is None:
                    break
                got += 1
                if got == nwarm:
                    t0 = time.perf_counter()               # buffers of all jobs are allocated by now
            rate = (got - nwarm) * WIDTH * HEIGHT / (time.perf_counter() - t0) / 1e6 if t0 and got > nwarm else 0.0
            pipe.close()
            return rate
        pipe_rate_device = run_pipe(lambda p: p.receive_device(), nfr=192)
        buf = m.alloc_frame(info0)
        pipe_rate = run_pipe(lambda p: p.receive(into=buf))
        pinned, ptrs = dec.alloc_frame_pinned(info0)
        pipe_rate_pinned = run_pipe(lambda p: p.receive(into=pinned))
        del pinned
        dec.free_frame_pinned(ptrs)

    # BASELINE configs[4], end to end, on every rank
    stream240 = None
    if args.stream240 > 0 and not args.no_e2e:
        n240 = max(args.stream240, world)
        dt, mine, ok, t_enc, bytes_per_frame, reps = stream240_leg(dec, m, rank, world, n240, barrier,
                                                                   min_frames_per_rank=args.stream240_min_frames)
        dt_all = max_over_ranks(dt)
        per_rank_s = all_ranks(dt)
        decoded = sum(all_ranks(mine))
        if min(all_ranks(1.0 if ok else 0.0)) < 1.0:
            sys.exit("bench.py: stream240 frames do not match their sources")
        stream240 = {"value": round(decoded * WIDTH * HEIGHT / dt_all / 1e6, 1), "unit": "Mpixel/s", "frames": n240,
                     "frames_decoded": int(decoded), "passes_over_own_share_rank0": reps,
                     "frames_this_rank": mine, "seconds": round(dt_all, 4),
                     "seconds_per_rank": [round(x, 4) for x in per_rank_s], "parity_checked": True,
                     "workload": "configs[4]: %d distinct 3840x2160 10-bit RGB lossless 5/3 + RCT frames (rgb48 out), frame i on "
                                 "rank i mod %d, every rank cycling through its share until it has decoded >= %d frames (warm-up "
                                 "on other frames); pageable packets in -> page-locked frames out through htj2k_pipe (batches of "
                                 "8, 3 in flight); host parsing + staging + PCIe + kernels inside the clock, max over ranks"
                                 % (n240, world, args.stream240_min_frames),
                     "compressed_MB_per_frame": round(bytes_per_frame / 1e6, 2),
                     "encode_s_outside_the_clock": round(t_enc, 1)}

    if rank == 0:
        algorithmic = idwt_launch_bytes / (idwt_launch_ms * 1e-3) / 1e9 if idwt_launch_ms > 0 else 0.0
        achieved = idwt_launch_hbm / (idwt_launch_ms * 1e-3) / 1e9 if idwt_launch_ms > 0 else 0.0
        traffic = committed_traffic(args.batch, nlaunch / max(rsteps, 1))
        res = {
            "metric": "Mpixels/s HTJ2K decode (4K lossless 5/3) at 1/2/4/8 GPU; IDWT HBM GB/s vs peak",
            "value": round(value, 2),
            "unit": "Mpixel/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32 (int16 storage)" if c16 else "int32",
            "data": "synthetic",
            "parity_checked": True,
            "seconds_per_rank": [round(x, 5) for x in rank_seconds],
            "affinity": affinity,
            "config": {"workload": "configs[1]: 3840x2160 RGB 8-bit lossless 5/3 + RCT, 64x64 codeblocks, 5 levels, "
                                   "single tile, HT cleanup pass only; %d frames per step per GPU as %d concurrent "
                                   "jobs (HIP streams; the frame pipeline keeps that many in flight), device-resident input "
                                   "(codeblock bytes + descriptors) and output (rgb24)" % (args.batch, njobs),
                       "frames_per_step": args.batch, "jobs": njobs, "codeblocks_per_step": nblocks,
                       "distinct_frames": len(streams), "block_errors": block_errors,
                       "sharding": "frames round-robin over ranks, no collective"},
            "roofline": {"bound": "hbm",
                         "kernel": "k_idwt_stream_ll16_x3 (levels 1-3 in one launch) + k_idwt_stream_ll16 (level 4; 16-bit sub-bands "
                                   "and LL bands) + k_idwt_stream_pack<5/3, 3> (level 5 fused with RCT + rgb24 store)"
                                   if ll16 and nlaunch == 3 * rsteps else
                                   "k_idwt_stream_ll16 (levels 1-4: 16-bit sub-bands and LL bands) + k_idwt_stream_pack<5/3, 3> "
                                   "(level 5 fused with RCT + rgb24 store)" if ll16 else
                                   "k_idwt_stream<5/3> (levels 1-4) + k_idwt_stream_pack<5/3, 3> (level 5 fused with RCT + rgb24 store)",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic[0] if traffic else None,
                         "traffic_source": ("profiles/" + traffic[1]) if traffic else None,
                         "launches": nlaunch, "avg_launch_us": round(idwt_launch_ms / max(nlaunch, 1) * 1e3, 2),
                         "hbm_MB_per_launch": round(idwt_launch_hbm / max(nlaunch, 1) / 1e6, 3),
                         "algorithmic_GBps": round(algorithmic, 1),
                         "algorithmic_MB_per_launch": round(idwt_launch_bytes / max(nlaunch, 1) / 1e6, 3),
                         "copy_ceiling": COPY_CEILING_GBS,
                         "copy_ceiling_measured": round(copy_measured, 1),
                         "frac_of_copy_ceiling_measured": round(achieved / copy_measured, 4) if copy_measured > 0 else None,
                         "copy_ceiling_note": "copy_ceiling: the guide's float4-copy figure; copy_ceiling_measured: htj2k_copy_bench "
                                              "in this process on this device right after the roofline pass -- a kernel that only copies "
                                              "512 MB to another 512 MB (about the bytes of one IDWT launch), best of three grid sizes",
                         "sub_bands_16bit": c16, "ll_bands_16bit": ll16,
                         "measured": "separate pass after the timed region: the same batch as one job, %d steps, every IDWT launch alone on the chip" % rsteps,
                         "note": "achieved / frac count the bytes the launches have to move through HBM (per sample: sub-bands "
                                 "read as 2 B where sub_bands_16bit, LL bands read and written as 2 B where ll_bands_16bit -- "
                                 "checked on the device, the transform runs again with int32 if one does not fit -- 4 B otherwise; "
                                 "the final level writes the rgb24 frame, 1 B per sample); algorithmic_GBps is SURVEY 8(d)'s "
                                 "4 B + 4 B per sample and level over the same time: a rate, not a share of the bus; `traffic` "
                                 "is what the PMC counters saw per launch"},
            "stage_ms_per_step_one_job": {"ht_decode_dequant": round(ht_ms / rsteps, 4),
                                          "idwt": round(idwt_ms / rsteps, 4),
                                          "mct_pack": round(pack_ms / rsteps, 4)},
            "one_job": one_job,
            "host": {"parse_ms_per_frame_one_core": round(sum(h[0] for h in host_ms) / len(host_ms), 3),
                     "staging_copy_ms_per_frame_one_core": round(sum(h[1] for h in host_ms) / len(host_ms), 3),
                     "parse_batch_wall_ms_per_frame": round(t_parse / args.batch * 1e3, 3),
                     "upload_ms_per_frame": round(t_upload / args.batch * 1e3, 3),
                     "end_to_end_Mpixel_s_single_frame_calls": round(e2e, 1),
                     "end_to_end_Mpixel_s_pipeline": round(pipe_rate, 1),
                     "end_to_end_Mpixel_s_pipeline_pinned_frames": round(pipe_rate_pinned, 1),
                     "packets_to_device_frames_Mpixel_s_pipeline": round(pipe_rate_device, 1),
                     "pipeline": "htj2k_pipe: 96 (to the device: 192) frames after 48 warm-up, batches of 8, depth 3, pageable packets in, "
                                 "frames out into pageable / page-locked (htj2k_host_alloc) planes",
                     "note": "parse = marker + Tier-2 parse of one frame on one core (no code-block byte is read: the "
                             "packets are uploaded as they are and k_gather builds the byte pool on the device); "
                             "staging copy = the packet's copy into page-locked memory"},
        }
        if not args.no_cpu_baseline and world == 1:          # reported baselines of the N = 1 run only
            one, hot, allc = cpu_baselines(streams)
            res["cpu_baseline"] = one
            res["cpu_baseline_hot_path"] = hot
            res["cpu_baseline_all_cores"] = allc
            res["speedup"] = {"device_resident_vs_hot_path_1_core": round(value / hot["value"], 1),
                              "pipeline_vs_whole_decode_1_core": round(pipe_rate_pinned / one["value"], 1) if pipe_rate_pinned else None,
                              "pipeline_vs_whole_decode_all_cores": round(pipe_rate_pinned / allc["value"], 2) if pipe_rate_pinned else None,
                              "note": "like with like: `value` (device-resident stages) against the oracle's same stages; the "
                                      "packets-to-frames pipeline against whole oracle decodes"}
        if args.part1 > 0:
            res["part1"] = part1_leg(dec, args.part1, not args.no_cpu_baseline)
        if stream240:
            res["stream240"] = stream240
        print(json.dumps(res), flush=True)
    for job in jobs:
        job.free()
    dec.close()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
