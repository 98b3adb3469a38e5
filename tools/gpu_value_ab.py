"""A/B of settings on the bench's two-job pass (what `value` measures) inside ONE process, interleaved round by round
(see gpu_idwt_ab.py for why).  usage: python tools/gpu_value_ab.py "ENV=val,knob=val" ... ("-" = defaults);
BATCH (default 128), JOBS (2), STEPS (6), ROUNDS (5) from the environment.  Prints the median Gpixel/s per setting."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ffmpeg_ht_amd as m
import vecgen

BATCH, JOBS = int(os.environ.get("BATCH", "128")), int(os.environ.get("JOBS", "2"))
STEPS, ROUNDS = int(os.environ.get("STEPS", "6")), int(os.environ.get("ROUNDS", "5"))
W, H = 3840, 2160
ENVS = ("HTJ2K_WPB", "HTJ2K_PK_LDS", "HTJ2K_OCC_LDS", "HTJ2K_STRIP", "HTJ2K_TW16", "HTJ2K_TWF", "HTJ2K_X3_TH")
KNOBS = {"idwt_pk": 1, "idwt_x3": 1, "ll16": 1, "coef16": 1, "ht_pair": 1}
streams = [vecgen.encode(list(vecgen.synth_image(W, H, 3, seed=2 + i, noise=8)), mct=1, nlevels=5, cb=(6, 6), transform=1) for i in range(2)]
dec = m.Decoder()
pk = [m.packet(x) for x in streams]
jobs = []
for j in range(JOBS):
    job = dec.job().parse_batch([pk[i % 2] for i in range(BATCH // JOBS)]); job.upload(); job.wait(); jobs.append(job)

def apply(spec):
    for e in ENVS: os.environ.pop(e, None)
    for k, v in KNOBS.items(): dec.set_int(k, v)
    for kv in spec.split(","):
        if "=" not in kv: continue
        k, v = kv.split("=")
        if k.startswith("HTJ2K_"): os.environ[k] = v
        else: dec.set_int(k, int(v))

MODE = os.environ.get("MODE", "follow")     # follow: as bench.py (the host reads every step's events); free: everything queued at once
# (Tried with a build that let the block decoder and the IDWT of a job be started by separate calls: job 0 half a step ahead of
# job 1, so that the IDWT of one always runs beside the block decoder of the other -- 162.3 Gpixel/s against 165.9 with both
# jobs in step: the kernel timeline shows the jobs of this pass running the same stage side by side, and that is the better
# arrangement -- two latency-bound k_ht_vlc2 launches fill each other's issue gaps.)
import threading
def timed():
    for job in jobs: job.run(7)
    for job in jobs: job.wait()
    t0 = time.perf_counter()
    if MODE == "threads":                   # a host thread per job, as htj2k_pipe's workers: run, wait for the step, run again
        def work(job):
            for _ in range(STEPS):
                job.run(7); job.stage_ms()
        th = [threading.Thread(target=work, args=(job,)) for job in jobs]
        for t in th: t.start()
        for t in th: t.join()
    else:
        for _ in range(STEPS):
            for job in jobs: job.run(7)
            if MODE == "follow":
                for job in jobs: job.stage_ms()
    for job in jobs: job.wait()
    return STEPS * (BATCH // JOBS) * JOBS * W * H / (time.perf_counter() - t0) / 1e9

settings = sys.argv[1:] or ["-"]
res = {i: [] for i in range(len(settings))}
for r in range(ROUNDS + 1):
    for i, s in enumerate(settings):
        apply(s)
        v = timed()
        if r: res[i].append(v)
for i, s in enumerate(settings):
    print("%-48s median %.1f  (min %.1f max %.1f) Gpixel/s" % (s, np.median(res[i]), min(res[i]), max(res[i])), flush=True)
