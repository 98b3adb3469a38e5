"""damaged HT bodies through the device path and the oracle (the loop of tests/test_gpu_parity.py::
test_damaged_ht_bodies_match_the_oracle): for every frame that differs, which code-blocks differ after the block-decode
stage, with their descriptors; the damaged stream is written to gpurun_out/damage_<name>_<it>.j2c for a closer look.
usage: python tools/gpu_damage_diff.py [iterations per stream]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ffmpeg_ht_amd as m
import oracle, streams, vecgen

N = int(sys.argv[1]) if len(sys.argv) > 1 else 24
dec, orc = m.Decoder(), oracle.OracleDecoder()
rng = np.random.default_rng(11)
names = ["gray_l5_cb64", "rgb_mct", "gray_3passes", "rgb_3passes_cb32", "gray_97_q2", "placeholder_2_3p", "noise_max",
         "gray_l3_cb256x16", "gray_3passes_vsc"]
extra = {"c16_256x192": vecgen.encode(vecgen.synth_image(256, 192, 3, seed=5, noise=10), mct=1, nlevels=4),
         "c16_512x256_cb32": vecgen.encode(vecgen.synth_image(512, 256, 3, seed=6, noise=30), mct=1, nlevels=5, cb=(5, 5))}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
nbad = 0
for name in names + sorted(extra):
    data, kw = (extra[name], {}) if name in extra else streams.get(name)
    start = data.index(b"\xff\x93") + 2
    for it in range(N):
        b = bytearray(data)
        mode = it % 4
        for _ in range([1, 8, 64, 400][mode]):
            pos = int(rng.integers(start, len(b) - 2))
            if mode == 0: b[pos] ^= 1 << int(rng.integers(0, 8))
            elif mode == 3: b[pos] = 0xFF
            else: b[pos] = int(rng.integers(0, 256))
        b = bytes(b)
        try:
            orc.decode_blocks(b, **kw)
        except oracle.DecodeError:
            continue
        dec.set_int("bitexact", kw.get("bitexact", 0))
        try:
            job = dec.job().parse(b).upload().run(1).wait()
        except m.Htj2kError:
            continue
        nb = orc.num_blocks()
        po, w, h, st = ctypes.c_uint32(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        blocks = []
        for i in range(nb):
            note = orc.L.orc_frame_block_note(orc.h, i, ctypes.byref(po), ctypes.byref(w), ctypes.byref(h), ctypes.byref(st))
            blocks.append((po.value, w.value, h.value, st.value, note))
        bad = []
        for tc in range(job.num_tilecomps()):
            a, o = job.plane(tc).view(np.uint32).reshape(-1), orc.plane(tc).view(np.uint32).reshape(-1)
            off = orc.plane_offset(tc)
            if np.array_equal(a, o):
                continue
            for i, (bp, bw, bh, bst, note) in enumerate(blocks):
                if not (off <= bp < off + a.size):
                    continue
                rel = bp - off
                idx = (rel + np.arange(bh)[:, None] * bst + np.arange(bw)[None, :]).reshape(-1)
                d = a[idx] != o[idx]
                if d.any():
                    k = int(np.argmax(d))
                    bad.append((i, tc, bw, bh, note, int(d.sum()), (k // bw, k % bw), hex(int(a[idx][k])), hex(int(o[idx][k]))))
        if bad:
            # the single-kernel HT decoder (ht_mode 0: MEL / VLC / MagSgn serial on lane 0) on the same frame
            dec.set_int("ht_mode", 0)
            j0 = dec.job().parse(b).upload().run(1).wait()
            same0 = all(np.array_equal(j0.plane(tc).view(np.uint32), orc.plane(tc).view(np.uint32)) for tc in range(j0.num_tilecomps()))
            print("   ht_mode 0 agrees with the oracle:", same0, "block errors", j0.block_errors())
            j0.free()
            dec.set_int("ht_mode", 1)
            nbad += 1
            gerr, oerr = job.block_errors(), orc.block_errors()
            print("DIFF", name, it, "mode", mode, "block errors gpu/oracle", gerr, oerr, "under-run blocks", orc.underrun_blocks())
            for r in bad[:8]:
                print("   block %d tc %d %dx%d note %d: %d samples differ, first at (row, col) %s gpu %s oracle %s" % r)
            open(os.path.join(ROOT, "gpurun_out", "damage_%s_%d.j2c" % (name, it)), "wb").write(b)
        job.free()
print("frames with differing blocks:", nbad)
