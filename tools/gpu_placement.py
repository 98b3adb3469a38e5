import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ffmpeg_ht_amd as m
import vecgen
data = vecgen.encode(list(vecgen.synth_image(3840, 2160, 3, seed=2, noise=8)), mct=1, nlevels=5, cb=(6, 6), transform=1)
dec = m.Decoder()
pk = m.packet(data)
jobs = []
for k in range(5):
    job = dec.job().parse_batch([pk] * 48); job.upload(); job.wait(); jobs.append(job)
for rnd in range(3):
    for k, job in enumerate(jobs):
        acc = []
        for _ in range(5):
            job.run(); job.wait()
            acc.append([ms * 1e3 for ms, _ in job.idwt_launches()])
        a = np.median(np.array(acc), axis=0)
        print("round %d job %d: %s sum %.1f us  ht %.3f ms" % (rnd, k, " ".join("%7.1f" % x for x in a), a.sum(), job.stage_ms()[0]), flush=True)
print("copy ceiling", dec.copy_bench(512, 10))
