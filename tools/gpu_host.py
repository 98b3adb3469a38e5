"""where does the host-side time of a batch go?  (warm numbers; batch of 8 4K frames)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ffmpeg_ht_amd as m
import bench
streams = bench.make_streams(4, 0)
pk = [m.packet(x) for x in streams]
batch = [pk[i % 4] for i in range(8)]
dec = m.Decoder()
for gather, thr in ((1, 1), (1, 4), (1, 16), (0, 1), (0, 16)):
    dec.set_int("device_gather", gather)
    dec.set_int("parse_threads", thr)
    job = dec.job()
    for it in range(3):
        t0 = time.perf_counter(); job.parse_batch(batch); t1 = time.perf_counter()
        job.upload(); t2 = time.perf_counter(); job.wait(); t3 = time.perf_counter()
        job.run(); job.wait(); t4 = time.perf_counter()
        info = job.frame_info(0)
        buf = m.alloc_frame(info)
        t5 = time.perf_counter()
        for f in range(8): job.download_frame(f) if False else dec.L.htj2k_job_download_frame(dec.h, job.h, f, __import__("ctypes").byref(buf[1]))
        t6 = time.perf_counter()
    hp, hs = job.host_ms()
    print("device_gather %d threads %2d: parse_batch(8) %.2f ms [per frame and thread: parse %.3f, staging copy %.3f]  upload(call) %.2f  upload(wait) %.2f  run+wait %.2f  d2h(8, pageable) %.2f" % (
        gather, thr, (t1 - t0) * 1e3, hp, hs, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t6 - t5) * 1e3), flush=True)
    job.free()
