"""stage times of the other BASELINE/SURVEY configurations (parity cases, not bench lines): C3 4K 4:2:2 12-bit 9/7
32x32 (cleanup only and with SigProp+MagRef), C4 8K 16-bit 5/3 gray and rgb48, and the bench's C2 for reference"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ffmpeg_ht_amd as m
import vecgen

def img(w, h, nc, depth, seed, dx=None):
    out = vecgen.synth_image(w, h, nc, depth=depth, seed=seed, noise=8)
    comps = list(out) if isinstance(out, (list, tuple)) else ([out[..., i] for i in range(nc)] if out.ndim == 3 else [out])
    if dx:
        comps = [c[:, ::dx[i]] for i, c in enumerate(comps)]
    return comps

cfgs = {
    "C2 4K rgb8 5/3 mct cb64":       lambda: vecgen.encode(img(3840, 2160, 3, 8, 2), mct=1, nlevels=5, cb=(6, 6), transform=1),
    "C3 4K 422 12b 9/7 cb32":        lambda: vecgen.encode(img(3840, 2160, 3, 12, 3, dx=[1, 2, 2]), depth=12, dx=[1, 2, 2], dy=[1, 1, 1], nlevels=5, cb=(5, 5), transform=0, qstep=1.0 / 16),
    "C3 + SigProp/MagRef":           lambda: vecgen.encode(img(3840, 2160, 3, 12, 3, dx=[1, 2, 2]), depth=12, dx=[1, 2, 2], dy=[1, 1, 1], nlevels=5, cb=(5, 5), transform=0, qstep=1.0 / 16, passes=3),
    "C4 8K gray16 5/3 L6":           lambda: vecgen.encode(img(7680, 4320, 1, 16, 4), depth=16, nlevels=6, cb=(6, 6), transform=1),
    "C4 8K rgb48 5/3 L6 mct":        lambda: vecgen.encode(img(7680, 4320, 3, 16, 5), depth=16, mct=1, nlevels=6, cb=(6, 6), transform=1),
}
dec = m.Decoder()
for name, mk in cfgs.items():
    t0 = time.time(); data = mk(); te = time.time() - t0
    pk = m.packet(data)
    nb = 4
    job = dec.job().parse_batch([pk] * nb); job.upload(); job.wait()
    for _ in range(2): job.run()
    job.wait()
    acc = np.zeros(3)
    for _ in range(5):
        job.run(); job.wait(); acc += np.array(job.stage_ms())
    acc /= 5
    info = job.frame_info(0)
    px = info.width * info.height * nb
    print("%-28s %5.1f MB  ht %.3f  idwt %.3f  pack %.3f ms per %d frames -> %.1f Gpixel/s  blocks %d errs %d (enc %.1fs)" % (
        name, len(data) / 1e6, acc[0], acc[1], acc[2], nb, px / acc.sum() / 1e6, job.num_blocks(), job.block_errors(), te), flush=True)
    job.free()
