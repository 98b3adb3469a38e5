"""device-resident decode rate of other 8-bit 4K formats (gray, 4:2:0 planar) beside the bench's rgb24, and which
path they take (16-bit sub-bands or not).  usage: python tools/gpu_formats.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ffmpeg_ht_amd as m
import oracle, vecgen

dec = m.Decoder()
orc = oracle.OracleDecoder()
W, H = 3840, 2160
cases = {
    "rgb24 (bench)": dict(img=vecgen.synth_image(W, H, 3, seed=2), kw=dict(mct=1)),
    "gray8": dict(img=vecgen.synth_image(W, H, 1, seed=2), kw=dict()),
    "yuv420p8": dict(img=vecgen.synth_image(W, H, 3, seed=2, dx=[1, 2, 2], dy=[1, 2, 2]), kw=dict(dx=[1, 2, 2], dy=[1, 2, 2], width=W, height=H)),
    "rgb48 10-bit": dict(img=vecgen.synth_image(W, H, 3, depth=10, seed=2, noise=20), kw=dict(mct=1, depth=10)),
}
for name, c in cases.items():
    data = vecgen.encode(c["img"], nlevels=5, **c["kw"])
    info_o, planes_o, _ = orc.decode(data)
    nb = 24
    job = dec.job().parse_batch([m.packet(data)] * nb).upload().run().wait()
    info, planes = job.download_frame(nb - 1)
    ok = all(np.array_equal(a, b) for a, b in zip(planes, planes_o))
    t0 = time.perf_counter()
    for _ in range(5):
        job.run()
    job.wait()
    dt = (time.perf_counter() - t0) / 5
    print("%-14s parity %s  coef16 %s  %.2f ms per %d frames = %.1f Gpixel/s  stages %s" % (
        name, ok, job.coef16(), dt * 1e3, nb, nb * W * H / dt / 1e9, tuple(round(v, 2) for v in job.stage_ms())), flush=True)
    job.free()
