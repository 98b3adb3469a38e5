"""frames with damaged codeblock bodies through jobs with 16-bit and with 32-bit LL bands (knob "ll16"): same frames,
whatever the coefficients add up to; counts how often an LL band really left the 16-bit range.
usage: python tools/gpu_ll16_corrupt.py [iterations]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ffmpeg_ht_amd as m
import vecgen

dec = m.Decoder()
rng = np.random.default_rng(17)
srcs = [vecgen.encode(vecgen.synth_image(256, 192, 3, depth=12, seed=9, noise=1500), depth=12, mct=0, nlevels=4),
        vecgen.encode(vecgen.synth_image(512, 256, 3, depth=10, seed=4, noise=300), depth=10, mct=1, nlevels=5),
        vecgen.encode(vecgen.synth_image(256, 192, 3, seed=5, noise=60), mct=1, nlevels=4)]
stat = dict(compared=0, fell_back=0, skipped=0, bad=0)
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 300):
    clean = srcs[it % len(srcs)]
    start = clean.index(b"\xff\x93") + 2
    b = bytearray(clean)
    mode = it % 3
    for _ in range(int(rng.integers(5, 300))):
        pos = int(rng.integers(start, len(b) - 70))
        if mode == 0: b[pos:pos + 64] = b"\xff" * 64
        elif mode == 1: b[pos] = int(rng.integers(0, 256))
        else: b[pos:pos + 16] = bytes(rng.integers(0, 256, 16, dtype=np.uint8))
    out = {}
    for ll16 in (1, 0):
        dec.set_int("ll16", ll16)
        try:
            job = dec.job().parse_batch([bytes(b)]).upload().run().wait()
        except m.Htj2kError:
            out = None
            break
        if not job.coef16():
            job.free(); out = None
            break
        out[ll16] = (job.ll16(), job.download_frame(0)[1])
        job.free()
    if not out:
        stat["skipped"] += 1
        continue
    stat["compared"] += 1
    stat["fell_back"] += out[1][0] == 2
    if not all(np.array_equal(a, c) for a, c in zip(out[1][1], out[0][1])):
        stat["bad"] += 1
        print("MISMATCH at iteration", it)
dec.set_int("ll16", 1)
print(stat)
sys.exit(1 if stat["bad"] else 0)
