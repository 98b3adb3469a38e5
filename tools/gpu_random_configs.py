"""random small codestreams (size, levels, block shape, depth, components, subsampling, transform, passes, HT / Part-1 /
MIXED, tiles, offsets, lowres) through the GPU path and the oracle: pixels, error counts and error codes must agree.
Geometry is biased towards multiples of 4 and 32 so that the fast stores and the 16-bit sub-band path get their share.
usage: python tools/gpu_random_configs.py [count] [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ffmpeg_ht_amd as m
import oracle, vecgen

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dec = m.Decoder()
orc = oracle.OracleDecoder()
stat = dict(ok=0, c16=0, enc_fail=0, frame_err=0, bad=0)
ONLY = set(int(v) for v in os.environ["ONLY"].split(",")) if os.environ.get("ONLY") else None
t0 = time.time()
for it in range(N):
    c16bias = os.environ.get("C16BIAS") == "1"         # mostly jobs that qualify for 16-bit sub-bands and the fast stores
    even = rng.random() < 0.6
    w = int(rng.integers(1, 12)) * 32 if even else int(rng.integers(1, 400))
    h = int(rng.integers(1, 10)) * 16 if even else int(rng.integers(1, 300))
    nc = int(rng.choice([1, 3, 3, 4]))
    depth = int(rng.choice([8, 8, 8, 10, 12, 16]))
    nl = int(rng.integers(0, 6))
    cbw = int(rng.integers(2, 8)); cbh = int(rng.integers(2, min(10, 12 - cbw) + 1))
    kw = dict(nlevels=nl, cb=(cbw, cbh), depth=depth)
    mode = int(rng.integers(0, 10))
    if mode <= 5: pass                                              # HT
    elif mode <= 7: kw.update(part1=True, cblk_style=int(rng.choice([0, 0, 1, 4, 8, 0x20, 5, 0x2F])))
    else: kw.update(mixed=True)
    if rng.random() < 0.25 and not kw.get("part1"): kw["passes"] = int(rng.choice([2, 3]))
    if rng.random() < 0.3: kw.update(transform=0, qstep=float(rng.choice([0.25, 1.0, 4.0])))
    sub = nc == 3 and rng.random() < 0.25
    dx = [1, 2, 2] if sub else None
    dy = [1, int(rng.choice([1, 2])), 0] if sub else None
    if sub: dy[2] = dy[1]
    if nc >= 3 and not sub and rng.random() < 0.6: kw["mct"] = 1
    if c16bias and rng.random() < 0.85:
        nl = int(rng.integers(1, 5))
        w = int(rng.integers(1, 9)) * (8 << nl)
        h = int(rng.integers(2, 200))
        nc = int(rng.choice([1, 3, 3]))
        depth = int(rng.choice([8, 8, 10]))
        cbw = int(rng.integers(2, 7)); cbh = int(rng.integers(2, min(10, 12 - cbw) + 1))
        kw = dict(nlevels=nl, cb=(cbw, cbh), depth=depth)
        sub = nc == 3 and rng.random() < 0.3
        dx = [1, 2, 2] if sub else None
        dy = [1, 1, 1] if sub else None
        if sub and (w // 2) % (8 << nl): sub, dx, dy = False, None, None
        if nc == 3 and not sub and rng.random() < 0.7: kw["mct"] = 1
        if rng.random() < 0.3: kw["tile"] = (int(rng.integers(1, 4)) * (8 << nl), int(rng.choice([32, 64, 96, 128])))
    if rng.random() < 0.2 and "tile" not in kw: kw["tile"] = (int(rng.choice([32, 64, 96, 100])), int(rng.choice([32, 48, 64, 70])))
    if rng.random() < 0.15: kw["offset"] = (int(rng.integers(0, 9)), int(rng.integers(0, 9)))
    if rng.random() < 0.2: kw["prog"] = int(rng.integers(0, 5))
    if rng.random() < 0.15: kw["prec"] = [(int(rng.integers(5, 9)), int(rng.integers(5, 9))), (int(rng.integers(4, 8)), int(rng.integers(4, 8)))]
    if rng.random() < 0.1: kw.update(sop=True, eph=bool(rng.integers(0, 2)))
    if rng.random() < 0.1 and not kw.get("part1") and not kw.get("mixed"): kw["placeholder_sets"] = int(rng.integers(1, 3))
    if rng.random() < 0.1: kw["guard_bits"] = int(rng.integers(1, 5))
    if rng.random() < 0.1 and kw.get("part1"): kw["drop_passes"] = int(rng.integers(1, 6))
    if rng.random() < 0.05: kw["force_include"] = True
    opts = {}
    if rng.random() < 0.1 and nl > 0: opts["reduction_factor"] = int(rng.integers(1, nl + 1))
    if kw.get("transform") == 0 and rng.random() < 0.3: opts["bitexact"] = 1
    if ONLY is not None and it not in ONLY:
        continue
    try:
        img = vecgen.synth_image(w, h, nc, depth=depth, seed=it + 7, noise=int(rng.choice([0, 4, 20])), dx=dx, dy=dy)
        if sub: kw.update(dx=dx, dy=dy, width=w, height=h)
        data = vecgen.encode(img, **kw)
        # now and then inside a JP2 file: enumerated colourspace (16 sRGB, 17 grey, 18 sYCC -> planar YUV) and, for
        # three components, a channel definition box that permutes them (write_frame's plane choice, jpeg2000dec.c:2326)
        if rng.random() < 0.15 and nc in (1, 3):
            cs = 17 if nc == 1 else int(rng.choice([16, 18]))
            cdef = None
            if nc == 3 and rng.random() < 0.5:
                perm = rng.permutation(3)
                cdef = [(c, 0, int(perm[c]) + 1) for c in range(3)]
            data = vecgen.jp2_wrap(data, w, h, nc, depth, colourspace=cs, cdef=cdef)
    except Exception as e:
        stat["enc_fail"] += 1
        continue
    dec.set_int("bitexact", opts.get("bitexact", 0)); dec.set_int("reduction_factor", opts.get("reduction_factor", 0))
    try:
        info_o, planes_o, _ = orc.decode(data, **opts); eo = 0
    except oracle.DecodeError as e:
        eo = e.code
    try:
        job = dec.job().parse_batch([data, data]).upload().run().wait()
        res = [job.download_frame(f)[1] for f in range(2)]
        c16 = job.coef16(); nerr = job.block_errors(); job.free(); eg = 0
    except m.Htj2kError as e:
        eg = e.code
    if eo or eg:
        stat["frame_err"] += 1
        if eo != eg:
            stat["bad"] += 1; print("ERROR CODE MISMATCH", it, eo, eg, (w, h, nc, depth), kw, opts, flush=True)
        continue
    good = nerr == 2 * orc.block_errors() and all(np.array_equal(a, b) for f in range(2) for a, b in zip(res[f], planes_o))
    stat["ok" if good else "bad"] += 1
    stat["c16"] += bool(c16)
    if not good:
        print("MISMATCH", it, (w, h, nc, depth), kw, opts, "coef16", c16, flush=True)
        # who is wrong: the long-lived oracle parser / device context, or a fresh one?
        orc2 = oracle.OracleDecoder(); dec2 = m.Decoder()
        dec2.set_int("bitexact", opts.get("bitexact", 0)); dec2.set_int("reduction_factor", opts.get("reduction_factor", 0))
        _, po2, _ = orc2.decode(data, **opts)
        _, pg2, _, _ = dec2.decode(data)
        eq = lambda A, B: all(np.array_equal(a, b) for a, b in zip(A, B))
        print("   old oracle == fresh oracle:", eq(planes_o, po2), " old gpu == fresh gpu:", eq(res[0], pg2),
              " fresh gpu == fresh oracle:", eq(pg2, po2), " frame 0 == frame 1:", eq(res[0], res[1]), flush=True)
        orc2.close(); dec2.close()
        if ONLY is not None:
            print("  block errors gpu/oracle", nerr, orc.block_errors(), "pix_fmt", info_o.pix_fmt)
            for pi, (a, b) in enumerate(zip(res[0], planes_o)):
                d = a.astype(np.int64) - b.astype(np.int64)
                ys, xs = np.nonzero(d)
                print("  plane", pi, a.shape, "ndiff", len(ys), "maxabs", int(np.abs(d).max()) if d.size else 0,
                      "first", (int(ys[0]), int(xs[0]), int(a[ys[0], xs[0]]), int(b[ys[0], xs[0]])) if len(ys) else None,
                      "rows", (int(ys.min()), int(ys.max())) if len(ys) else None, "cols", (int(xs.min()), int(xs.max())) if len(ys) else None)
            for knob in (("idwt_mode", 0), ("idwt_mode", 1), ("ht_mode", 0), ("coef16", 0)):
                dec.set_int(*knob)
                _, pl, _, st2 = dec.decode(data)
                print("   with", knob, "equal to oracle:", all(np.array_equal(a, b) for a, b in zip(pl, planes_o)), flush=True)
                dec.set_int(knob[0], 3 if knob[0] == "idwt_mode" else 1)
    if it % 50 == 49: print(it + 1, stat, "%.0fs" % (time.time() - t0), flush=True)
print("done", stat, flush=True)
sys.exit(1 if stat["bad"] else 0)
