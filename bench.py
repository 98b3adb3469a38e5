#!/usr/bin/env python3
"""bench.py -- Mpixel/s of the MI355X-native HTJ2K decode hot path on BASELINE.json configs[1]
(3840x2160 RGB 8-bit, lossless 5/3 + RCT, 64x64 codeblocks, 5 levels), device-resident.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path -- HT block decode + dequantisation, inverse DWT, inverse
RCT + level shift + clip + rgb24 pack (tile_codeblocks() + the tail of jpeg2000_decode_tile(),
libavcodec/jpeg2000dec.c:2212-2395) -- over one batch of BATCH synthetic 4K frames whose
compressed codeblock bytes and descriptors are already resident in HBM when the timed region
starts; decoded frames stay in HBM.  Host marker/Tier-2 parsing and PCIe are outside the
timed region (their rates are reported as extra fields and in DESIGN.md, never as `value`).

Frames of a stream are independent, so ranks shard them round-robin with no data-path
collective ("weak" scaling: every rank decodes its own batch per step).

The JSON line also carries
  roofline      the IDWT kernels (the HBM-bound stage BASELINE.json's north_star names):
                algorithmic bytes (sum over levels of 2*4*lh*lv per plane) / their
                launch durations, measured live with HIP events around every IDWT launch of the
                timed region on the kernels' own stream, against the 8 TB/s HBM3E peak.  The final
                level runs fused with the inverse MCT + rgb24 store, so it physically writes 1
                byte per sample where the algorithmic figure counts 4: `achieved_min_hbm_traffic`
                and `frac_min_hbm_traffic` restate the rate with that launch counted at
                4*lh*lv read + frame bytes written (what has to cross HBM at the least), and
                `copy_ceiling` is the float4-copy rate the microarchitecture guide quotes
  cpu_baseline  the CPU oracle (a single-thread C restatement of the reference decoder:
                kind "port") timed on this box's host cores on a bounded sample of the same
                workload -- test infrastructure used here only as the measured baseline.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

WIDTH, HEIGHT, NCOMP = 3840, 2160, 3
NLEVELS, CB = 5, (6, 6)
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
COPY_CEILING_GBS = 6290.0      # same guide, line 36: float4 copy, 79 % of the spec


def committed_traffic(frames_per_step):
    """HBM bytes per IDWT launch from the PMC passes committed under profiles/ (rocprofv3 cannot run inside this
    process): (2 * FETCH_SIZE + WRITE_SIZE) * 1024 averaged over the IDWT launches of one step of this same
    workload (tools/prof_r01.sh + tools/make_profiles.py); None when the batch differs from the profiled one."""
    best = None
    pdir = os.path.join(ROOT, "profiles")
    if os.path.isdir(pdir):
        for name in sorted(os.listdir(pdir)):
            if name.endswith("_idwt_traffic.json"):
                try:
                    d = json.load(open(os.path.join(pdir, name)))
                    if d.get("frames_per_step") == frames_per_step:
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


def make_streams(nstreams, rank):
    """distinct synthetic 4K RGB frames (BASELINE.md section 3 generator) -> HTJ2K codestreams"""
    import vecgen
    out = []
    for i in range(nstreams):
        img = vecgen.synth_image(WIDTH, HEIGHT, NCOMP, depth=8, seed=2 + i + 16 * rank, noise=8)
        out.append(vecgen.encode(img, mct=1, nlevels=NLEVELS, cb=CB, transform=1))
    return out


def cpu_baseline(streams, budget_s=12.0):
    """single-thread CPU oracle on a bounded sample of the same frames"""
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
    orc.close()
    return {"value": round(n * WIDTH * HEIGHT / el / 1e6, 3), "unit": "Mpixel/s", "cores": 1, "kind": "port",
            "sample": "%d decodes of the bench's 4K RGB lossless frames in %.1f s, single thread, "
                      "oracle/j2k_oracle.c (C restatement of the reference decoder incl. host parsing)" % (n, el)}


def two_jobs_leg(dec, packets, steps):
    """The same frames as two jobs on two HIP streams (not part of `value`, N=1 only): the instruction-bound HT kernels
    of one job run beside the bandwidth-bound IDWT launches of the other.  It is reported beside `value` and not as
    `value` because the IDWT launches then share the chip, and `roofline` is defined per launch (bench.py --jobs 2
    measures everything that way)."""
    import torch
    jobs = [dec.job().parse_batch(packets[i::2]) for i in range(2)]
    for job in jobs:
        job.upload()
    for job in jobs:
        job.wait()
    for _ in range(2):
        for job in jobs:
            job.run(7)
    for job in jobs:
        job.wait()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for job in jobs:
            job.run(7)
        for job in jobs:
            job.stage_ms()                                 # as in the timed region: the step's events are read
    for job in jobs:
        job.wait()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for job in jobs:
        job.free()
    return {"value": round(steps * len(packets) * WIDTH * HEIGHT / dt / 1e6, 2), "unit": "Mpixel/s", "jobs": 2,
            "steps": steps, "ms_per_step": round(dt / steps * 1e3, 4)}


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=48, help="4K frames per step and per GPU (16 GB of HBM at 48; the rate is flat from 16: 66 -> 69 Gpixel/s)")
    ap.add_argument("--jobs", type=int, default=1, help="the batch is split over this many jobs (HIP streams): the "
                    "latency-bound VLC kernel of one job overlaps the bandwidth-bound kernels of the other")
    ap.add_argument("--distinct", type=int, default=4, help="distinct synthetic frames per rank (cycled to fill the batch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--part1", type=int, default=0, metavar="FRAMES",
                    help="also time the same 4K frames coded with Part-1 (MQ) blocks: FRAMES per step through k_mq_decode "
                         "(rank 0, reported as the extra object \"part1\"; SURVEY 8f rank 3, not part of `value`)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the three single-frame htj2k_decode() calls after the "
                    "timed region (profiling runs: every kernel launch in the trace is then a batch launch)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    # rehearsal knobs (never set by the driver): several ranks on the one GPU of a test box, over gloo
    backend = os.environ.get("HTJ2K_BENCH_BACKEND", "nccl")
    device = int(os.environ["HTJ2K_BENCH_DEVICE"]) if "HTJ2K_BENCH_DEVICE" in os.environ else (local_rank if world > 1 else 0)
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(device)
        dist.init_process_group(backend)
    import ffmpeg_ht_amd as m

    streams = make_streams(min(args.distinct, args.batch), rank)
    batch = [streams[i % len(streams)] for i in range(args.batch)]
    njobs = max(1, min(args.jobs, args.batch))
    per_job = [batch[i::njobs] for i in range(njobs)]

    dec = m.Decoder(device_id=device)
    per_job = [[m.packet(x) for x in b] for b in per_job]     # padded packet buffers, built once
    jobs = [dec.job().parse_batch(b) for b in per_job]         # cold: allocates the pinned byte pools
    t0 = time.perf_counter()
    for job, b in zip(jobs, per_job):
        job.parse_batch(b)
    t_parse = time.perf_counter() - t0
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
    # parity spot-check of what is being timed: frame 0 of the batch is lossless vs its source
    barrier()
    t0 = time.perf_counter()
    ht_ms = idwt_ms = pack_ms = 0.0
    idwt_launch_ms, idwt_launch_bytes, idwt_launch_hbm = 0.0, 0.0, 0.0
    nlaunch = 0
    for _ in range(args.steps):
        for job in jobs:
            job.run(7)
        # the per-stage / per-launch HIP events are read after the step's stream work is done
        for job in jobs:
            a, b, c = job.stage_ms()
            ht_ms += a
            idwt_ms += b
            pack_ms += c
            for (ms, by), hb in zip(job.idwt_launches(), job.idwt_hbm_bytes()):
                idwt_launch_ms += ms
                idwt_launch_bytes += by
                idwt_launch_hbm += hb
                nlaunch += 1
    for job in jobs:
        job.wait()
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)

    frames_total = args.steps * args.batch * world
    value = frames_total * WIDTH * HEIGHT / elapsed / 1e6

    # (before the host legs create and destroy their streams: the two jobs' streams should sit on hardware queues of their own)
    two_jobs = None
    if njobs == 1 and not args.no_e2e and world == 1 and args.batch >= 2:
        two_jobs = two_jobs_leg(dec, per_job[0], max(5, args.steps // 2))

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
        pk = [m.packet(x) for x in streams]

        def run_pipe(buf, nwarm=24, nfr=96):
            pipe = dec.pipe(batch=8, depth=3)
            sent = got = 0
            t0 = None
            while got < nwarm + nfr:
                while sent < nwarm + nfr and pipe.send(pk[sent % len(pk)]):
                    sent += 1
                if sent == nwarm + nfr:
                    pipe.flush()
                if pipe.receive(into=buf) is None:
                    break
                got += 1
                if got == nwarm:
                    t0 = time.perf_counter()               # buffers of all three jobs are allocated by now
            rate = (got - nwarm) * WIDTH * HEIGHT / (time.perf_counter() - t0) / 1e6 if t0 and got > nwarm else 0.0
            pipe.close()
            return rate
        def run_pipe_device(nwarm=24, nfr=192):
            pipe = dec.pipe(batch=8, depth=3)
            sent = got = 0
            t0 = None
            while got < nwarm + nfr:
                while sent < nwarm + nfr and pipe.send(pk[sent % len(pk)]):
                    sent += 1
                if sent == nwarm + nfr:
                    pipe.flush()
                if pipe.receive_device() is None:
                    break
                got += 1
                if got == nwarm:
                    t0 = time.perf_counter()
            rate = (got - nwarm) * WIDTH * HEIGHT / (time.perf_counter() - t0) / 1e6 if t0 and got > nwarm else 0.0
            pipe.close()
            return rate
        pipe_rate_device = run_pipe_device()
        pipe_rate = run_pipe(m.alloc_frame(info0))
        pinned, ptrs = dec.alloc_frame_pinned(info0)
        pipe_rate_pinned = run_pipe(pinned)
        del pinned
        dec.free_frame_pinned(ptrs)

    if rank == 0:
        achieved = idwt_launch_bytes / (idwt_launch_ms * 1e-3) / 1e9 if idwt_launch_ms > 0 else 0.0
        achieved_hbm = idwt_launch_hbm / (idwt_launch_ms * 1e-3) / 1e9 if idwt_launch_ms > 0 else 0.0
        traffic = committed_traffic(args.batch) if njobs == 1 else None
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
            "dtype": "int32",
            "data": "synthetic",
            "config": {"workload": "configs[1]: 3840x2160 RGB 8-bit lossless 5/3 + RCT, 64x64 codeblocks, 5 levels, "
                                   "single tile, HT cleanup pass only; %d frames per step per GPU in %d concurrent "
                                   "jobs (HIP streams), device-resident input (codeblock bytes + descriptors) and "
                                   "output (rgb24)" % (args.batch, njobs),
                       "frames_per_step": args.batch, "jobs": njobs, "codeblocks_per_step": nblocks,
                       "sharding": "frames round-robin over ranks, no collective"},
            "roofline": {"bound": "hbm",
                         "kernel": "k_idwt_stream<5/3> (levels 1-4) + k_idwt_stream_pack<5/3,3> (level 5 fused with RCT + rgb24 store)",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic[0] if traffic else None,
                         "traffic_source": ("profiles/" + traffic[1]) if traffic else None,
                         "launches": nlaunch, "avg_launch_us": round(idwt_launch_ms / max(nlaunch, 1) * 1e3, 2),
                         "algorithmic_MB_per_launch": round(idwt_launch_bytes / max(nlaunch, 1) / 1e6, 3),
                         "achieved_min_hbm_traffic": round(achieved_hbm, 1),
                         "frac_min_hbm_traffic": round(achieved_hbm / HBM_PEAK_GBS, 4),
                         "min_hbm_MB_per_launch": round(idwt_launch_hbm / max(nlaunch, 1) / 1e6, 3),
                         "copy_ceiling": COPY_CEILING_GBS,
                         "sub_bands_16bit": bool(jobs[0].coef16()),
                         "ll_bands_16bit": jobs[0].ll16() == 1,
                         "note": "achieved/frac count SURVEY 8(d)'s algorithmic bytes (4 B per sample read and written "
                                 "per level); when sub_bands_16bit is true the sub-bands really move as 2 B samples (exact "
                                 "for this workload: every band has M_b <= 15), with ll_bands_16bit the LL bands between the "
                                 "levels too (checked on the device, second run with int32 if one does not fit), so frac can "
                                 "exceed the share of the bus that is busy -- *_min_hbm_traffic and `traffic` count the "
                                 "bytes that really move"},
            "stage_ms_per_step_sum_over_jobs": {"ht_decode_dequant": round(ht_ms / args.steps, 4),
                                                "idwt": round(idwt_ms / args.steps, 4),
                                                "mct_pack": round(pack_ms / args.steps, 4)},
            "host": {"parse_ms_per_frame": round(t_parse / args.batch * 1e3, 3),
                     "upload_ms_per_frame": round(t_upload / args.batch * 1e3, 3),
                     "end_to_end_Mpixel_s_single_frame_calls": round(e2e, 1),
                     "end_to_end_Mpixel_s_pipeline": round(pipe_rate, 1),
                     "end_to_end_Mpixel_s_pipeline_pinned_frames": round(pipe_rate_pinned, 1),
                     "packets_to_device_frames_Mpixel_s_pipeline": round(pipe_rate_device, 1),
                     "pipeline": "htj2k_pipe: 96 frames after 24 warm-up, batches of 8, 3 in flight, pageable packets in, "
                                 "frames out into pageable / page-locked (htj2k_host_alloc) planes"},
        }
        if not args.no_cpu_baseline and world == 1:          # a reported baseline of the N = 1 run only
            res["cpu_baseline"] = cpu_baseline(streams)
        if args.part1 > 0:
            res["part1"] = part1_leg(dec, args.part1, not args.no_cpu_baseline)
        if two_jobs:
            res["two_jobs"] = two_jobs
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
