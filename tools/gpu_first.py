"""First contact with the GPU: KATs, a few generated streams, unit kernels; prints diffs vs the oracle."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ffmpeg_ht_amd as m
import oracle, vecgen

dec = m.Decoder()
print("device", dec.device_name(), flush=True)
orc = oracle.OracleDecoder()
mode = int(os.environ.get("MODE", "1"))
dec.set_int("idwt_mode", mode)

def cmp_stream(name, data, **kw):
    info_o, planes_o, n_o = orc.decode(data, **kw)
    t = time.time()
    info, planes, n, st = dec.decode(data)
    dt = time.time() - t
    ok = all(np.array_equal(a, b) for a, b in zip(planes, planes_o)) and n == n_o
    md = max(int(np.abs(a.astype(np.int64) - b.astype(np.int64)).max()) for a, b in zip(planes, planes_o))
    print(f"{name:34s} {'OK ' if ok else 'BAD'} maxdiff={md} consumed={n}/{n_o} blocks={st.n_codeblocks} errs={st.n_block_errors}/{orc.block_errors()} "
          f"ms: parse={st.ms_parse:.2f} ht={st.ms_ht:.3f} idwt={st.ms_idwt:.3f} pack={st.ms_pack:.3f} wall={dt*1e3:.1f}", flush=True)
    return ok

kats = json.load(open(os.path.join(ROOT, "tests", "golden", "kats.json")))
for k in kats:
    cmp_stream(k["name"], bytes.fromhex(k["hex"]))

g = vecgen.synth_image(200, 150, 1, seed=3)
rgb = vecgen.synth_image(190, 131, 3, seed=5)
cmp_stream("gray 200x150 L5", vecgen.encode(g))
cmp_stream("gray cb32", vecgen.encode(g, cb=(5, 5)))
cmp_stream("gray cb16x64 L3", vecgen.encode(g, cb=(4, 6), nlevels=3))
cmp_stream("gray L0", vecgen.encode(g, nlevels=0))
cmp_stream("gray off(3,5) L2", vecgen.encode(vecgen.synth_image(201, 149, 1, seed=4), nlevels=2, offset=(3, 5)))
cmp_stream("rgb mct", vecgen.encode(rgb, mct=1))
cmp_stream("rgb tiles", vecgen.encode(rgb, mct=1, tile=(64, 64), nlevels=3))
cmp_stream("gray16", vecgen.encode(vecgen.synth_image(160, 120, 1, depth=16, seed=8, noise=400), depth=16, nlevels=4))
cmp_stream("gray 97 q2", vecgen.encode(g, transform=0, qstep=2))
cmp_stream("rgb 97 ict", vecgen.encode(rgb, transform=0, mct=1, qstep=1))
cmp_stream("gray 3 passes", vecgen.encode(g, passes=3))
cmp_stream("gray 2 passes vsc", vecgen.encode(g, passes=2, vsc=True))
cmp_stream("gray 97 3 passes", vecgen.encode(g, passes=3, transform=0, qstep=2))
cmp_stream("gray plhd", vecgen.encode(g, placeholder_sets=1))
cmp_stream("gray 7x5 L1", vecgen.encode(vecgen.synth_image(7, 5, 1, seed=9), nlevels=1))
cmp_stream("gray 3x1 L2", vecgen.encode([np.array([[77, 3, 250]])], nlevels=2))
cmp_stream("noise", vecgen.encode([np.random.default_rng(2).integers(0, 256, (130, 130))], nlevels=3))
cmp_stream("zeros", vecgen.encode([np.full((100, 100), 128)], nlevels=3))
ycc = vecgen.synth_image(192, 128, 3, depth=12, seed=11, noise=30, dx=[1, 2, 2], dy=[1, 1, 1])
cmp_stream("yuv422p12 97", vecgen.encode(ycc, depth=12, dx=[1, 2, 2], dy=[1, 1, 1], transform=0, qstep=1, cb=(5, 5), width=192, height=128))

# IDWT unit
rng = np.random.default_rng(7)
for typ, name in ((1, "53"), (0, "97f"), (2, "97i")):
    bad = 0
    for it in range(12):
        x0, y0 = int(rng.integers(0, 9)), int(rng.integers(0, 9))
        w, h = int(rng.integers(1, 200)), int(rng.integers(1, 150))
        lev = int(rng.integers(1, 8))
        border = [[x0, x0 + w], [y0, y0 + h]]
        if typ == 0:
            p = (rng.standard_normal((h, w)) * 100).astype(np.float32)
        else:
            p = rng.integers(-2000, 2000, (h, w)).astype(np.int32) * (256 if typ == 2 else 1)
        a = oracle.idwt(p, border, lev, typ)
        for md in (0, 1):
            dec.set_int("idwt_mode", md)
            b = dec.idwt(p, border, lev, typ)
            if not np.array_equal(a.view(np.uint32), b.view(np.uint32)):
                bad += 1
                print("  idwt mismatch", name, "mode", md, border, lev, "ndiff", int((a.view(np.uint32) != b.view(np.uint32)).sum()))
    print("idwt", name, "bad", bad, flush=True)
dec.set_int("idwt_mode", mode)

big = vecgen.synth_image(1920, 1080, 3, seed=2)
data = vecgen.encode(big, mct=1)
cmp_stream("1080p rgb", data)
cmp_stream("1080p rgb again", data)
for md in (0, 1):
    dec.set_int("idwt_mode", md)
    for typ in (1, 0):
        ms = dec.idwt_bench(3840, 2160, 5, typ, 3, 5)
        print(f"idwt bench 4K x3 type {typ} mode {md}: {ms:.3f} ms -> {265.2e6/ms/1e9*1e3:.1f} GB/s algorithmic", flush=True)
