import os, sys, time
import numpy as np
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ffmpeg_ht_amd as m
import oracle, vecgen
dec = m.Decoder()
orc = oracle.OracleDecoder()
img = vecgen.synth_image(3840, 2160, 3, depth=12, seed=2, noise=20)
for qstep in (1.0, 4.0, 16.0):
    data = vecgen.encode(img, depth=12, mct=1, part1=True, transform=0, qstep=qstep, cb=(5, 5))
    bpp = len(data) * 8 / (3840 * 2160)
    info, planes, n, st = dec.decode(data)
    info_o, planes_o, _ = orc.decode(data)
    ok = all(np.array_equal(a, b) for a, b in zip(planes, planes_o))
    nb = 40
    job = dec.job()
    job.parse_batch([m.packet(data)] * nb); job.upload(); job.run(); job.wait()
    t0 = time.perf_counter()
    for _ in range(3):
        job.run()
    job.wait()
    dt = (time.perf_counter() - t0) / 3
    print("qstep %.4f: %.2f bit/pixel, parity %s, batch %d: %.2f ms/step %.1f Mpixel/s stages %s blocks %d" % (qstep, bpp, ok, nb, dt * 1e3, nb * 3840 * 2160 / dt / 1e6, job.stage_ms(), job.num_blocks() // nb), flush=True)
    job.free()
