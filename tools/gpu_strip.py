"""per-launch IDWT times of the 4K bench batch for several strip heights (HTJ2K_STRIP)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ffmpeg_ht_amd as m
import bench

streams = bench.make_streams(2, 0)
batch = [streams[i % 2] for i in range(8)]
dec = m.Decoder()
job = dec.job().parse_batch(batch); job.upload(); job.wait()
for strip in sys.argv[1:] or ["0", "16", "32", "64", "128", "256"]:
    if strip == "0": os.environ.pop("HTJ2K_STRIP", None)
    else: os.environ["HTJ2K_STRIP"] = strip
    for _ in range(3): job.run(7)
    job.wait()
    acc = None
    for _ in range(10):
        job.run(7); job.wait()
        l = job.idwt_launches()
        t = np.array([x[0] for x in l]); by = np.array([x[1] for x in l])
        acc = t if acc is None else acc + t
    acc /= 10
    print("strip=%-4s" % strip, " ".join("%7.1f" % (x * 1e3) for x in acc), " total %.1f us  %.0f GB/s" % (acc.sum() * 1e3, by.sum() / acc.sum() / 1e6))
job.free(); dec.close()
