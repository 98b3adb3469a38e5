"""A/B of IDWT launch settings inside ONE process on ONE box, interleaved: the boxes of the pool -- and one box over a few
minutes -- differ by more than most of the effects looked for (same binary, same settings: 660 / 711 / 722 us for C2's final
level on three occasions), so settings are only ever compared round by round in the same run.
usage: python tools/gpu_idwt_ab.py [C2|C3|C4g|C4c] "ENV=val,ENV=val;knob=val" ...   (one argument per setting; "-" = defaults)
prints the median per-launch times (us) of every setting and the sum"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ffmpeg_ht_amd as m
import vecgen

def img(w, h, nc, depth, seed, dx=None):
    comps = list(vecgen.synth_image(w, h, nc, depth=depth, seed=seed, noise=8))
    return [c[:, ::dx[i]] for i, c in enumerate(comps)] if dx else comps

WORK = {
    "C2": (48, lambda: vecgen.encode(img(3840, 2160, 3, 8, 2), mct=1, nlevels=5, cb=(6, 6), transform=1)),
    "C3": (32, lambda: vecgen.encode(img(3840, 2160, 3, 12, 3, dx=[1, 2, 2]), depth=12, dx=[1, 2, 2], dy=[1, 1, 1], nlevels=5, cb=(5, 5), transform=0, qstep=1.0 / 16)),
    "C4g": (16, lambda: vecgen.encode(img(7680, 4320, 1, 16, 4), depth=16, nlevels=6, cb=(6, 6), transform=1)),
    "C4c": (8, lambda: vecgen.encode(img(7680, 4320, 3, 16, 5), depth=16, mct=1, nlevels=6, cb=(6, 6), transform=1)),
}
ENVS = ("HTJ2K_WPB", "HTJ2K_PK_LDS", "HTJ2K_OCC_LDS", "HTJ2K_STRIP", "HTJ2K_TW16", "HTJ2K_TW32", "HTJ2K_TWF", "HTJ2K_X3_TH")
KNOBS = {"idwt_pk": 1, "idwt_x3": 1, "ll16": 1, "coef16": 1}

args = sys.argv[1:]
work = args.pop(0) if args and args[0] in WORK else "C2"
settings = args or ["-"]
nb, mk = WORK[work]
dec = m.Decoder()
job = dec.job().parse_batch([m.packet(mk())] * nb); job.upload(); job.wait()

def apply(spec):
    for e in ENVS: os.environ.pop(e, None)
    for k, v in KNOBS.items(): dec.set_int(k, v)
    for kv in spec.replace(";", ",").split(","):
        if "=" not in kv: continue
        k, v = kv.split("=")
        if k.startswith("HTJ2K_"): os.environ[k] = v
        else: dec.set_int(k, int(v))

ROUNDS = int(os.environ.get("ROUNDS", "7"))
res = {s: [] for s in settings}
for r in range(ROUNDS + 1):
    for s in settings:
        apply(s)
        job.run(); job.wait()
        job.run(); job.wait()
        if r: res[s].append([ms * 1e3 for ms, _ in job.idwt_launches()])
for s in settings:
    a = np.median(np.array(res[s]), axis=0)
    print("%-44s %s  sum %.1f us" % (s, " ".join("%7.1f" % x for x in a), a.sum()), flush=True)
