"""stage times of the other BASELINE/SURVEY configurations (parity cases, not bench lines): C3 4K 4:2:2 12-bit 9/7
32x32 (cleanup only and with SigProp+MagRef), C4 8K 16-bit 5/3 gray and rgb48, and the bench's C2 for reference"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ffmpeg_ht_amd as m
import vecgen

def img(w, h, nc, depth, seed, dx=None):
    out = vecgen.synth_image(w, h, nc, depth=depth, seed=seed, noise=int(os.environ.get("NOISE", "8")))
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
import json
dec = m.Decoder()
for kv in os.environ.get("KNOBS", "").split(","):
    if "=" in kv: dec.set_int(kv.split("=")[0], int(kv.split("=")[1]))
# frames per job: the bench's 128 for C2 and as many samples for the others (rounds 1 and 2 ran 48 / 32 / 16 / 8: NBDIV=3 comes
# close; the launches of every stage balance better over the chip the more waves they have -- C3 108 -> 120 Gpixel/s)
NB = {"C2": 128, "C3": 96, "C4 8K gray16": 48, "C4 8K rgb48": 24}
NB = {k: max(1, v // int(os.environ.get("NBDIV", "1"))) for k, v in NB.items()}
out = {}
for name, mk in cfgs.items():
    if len(sys.argv) > 1 and not any(a in name for a in sys.argv[1:]): continue
    t0 = time.time(); data = mk(); te = time.time() - t0
    pk = m.packet(data)
    nb = next(v for k, v in NB.items() if name.startswith(k))
    job = dec.job().parse_batch([pk] * nb); job.upload(); job.wait()
    for _ in range(2): job.run()
    job.wait()
    acc = np.zeros(3); lms = lhb = lalg = 0.0; per = {}
    R = 5
    for _ in range(R):
        job.run(); job.wait(); acc += np.array(job.stage_ms())
        for i, ((ms, by), hb) in enumerate(zip(job.idwt_launches(), job.idwt_hbm_bytes())):
            lms += ms; lalg += by; lhb += hb
            e = per.setdefault(i, [0.0, 0.0]); e[0] += ms; e[1] += hb
    acc /= R
    info = job.frame_info(0)
    px = info.width * info.height * nb
    frac = lhb / (lms * 1e-3) / 8e12
    print("%-28s %5.1f MB  ht %.3f  idwt %.3f  pack %.3f ms per %d frames -> %.1f Gpixel/s  blocks %d errs %d  IDWT %.0f GB/s of bytes that move = %.3f of 8 TB/s (algorithmic %.0f GB/s) coef16 %d ll16 %d bpw %d" % (
        name, len(data) / 1e6, acc[0], acc[1], acc[2], nb, px / acc.sum() / 1e6, job.num_blocks(), job.block_errors(),
        lhb / (lms * 1e-3) / 1e9, frac, lalg / (lms * 1e-3) / 1e9, job.coef16(), job.ll16(), job.ht_blocks_per_wave()), flush=True)
    print("      launches (us, MB, TB/s):", "  ".join("%.0f/%.0f/%.2f" % (v[0] / R * 1e3, v[1] / R / 1e6, v[1] / v[0] / 1e9) for v in per.values()), flush=True)
    out[name] = dict(frames_per_job=nb, ht_ms=round(float(acc[0]), 4), idwt_ms=round(float(acc[1]), 4), Gpixel_s=round(px / acc.sum() / 1e6, 1),
                     idwt_hbm_GBps=round(lhb / (lms * 1e-3) / 1e9, 1), idwt_frac_of_8TBps=round(frac, 4), idwt_algorithmic_GBps=round(lalg / (lms * 1e-3) / 1e9, 1),
                     blocks=job.num_blocks(), coef16=bool(job.coef16()), ll16=job.ll16())
    job.free()
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "configs.json"), "w"), indent=1)
