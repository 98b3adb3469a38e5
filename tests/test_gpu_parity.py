"""GPU parity tests proper: everything goes through the C ABI (libhtj2k_amd.so) on a real
MI355X and is compared with the CPU oracle on the same inputs.
Bars: bit-exact for every integer path (5/3, 9/7 fixed point, HT block decode, MCT,
pack); for the 9/7 float path the coefficient planes must be within 1 ULP and the final
pixels identical (in practice both are bit-identical: same operation order, no FMA)."""
import ctypes
import json
import os

import numpy as np
import pytest

import oracle
import streams
import vecgen

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def dec():
    import ffmpeg_ht_amd as m
    d = m.Decoder()
    assert d.device_name().startswith("gfx950"), d.device_name()
    yield d
    d.close()


def _ulp_diff(a, b):
    ai = a.view(np.int32).astype(np.int64)
    bi = b.view(np.int32).astype(np.int64)
    ai = np.where(ai < 0, -(ai & 0x7FFFFFFF), ai)
    bi = np.where(bi < 0, -(bi & 0x7FFFFFFF), bi)
    return int(np.abs(ai - bi).max()) if a.size else 0


KATS = json.load(open(os.path.join(HERE, "golden", "kats.json")))


@pytest.mark.parametrize("kat", KATS, ids=[k["name"] for k in KATS])
def test_kats_on_gpu(dec, kat):
    info, planes, consumed, st = dec.decode(bytes.fromhex(kat["hex"]))
    assert oracle.framecrc(planes) == int(kat["framecrc"], 16)
    assert st.n_block_errors == 0


@pytest.mark.parametrize("mode", [0, 1, 3, 4], ids=["idwt_generic", "idwt_tile", "idwt_stream_fused",
                                                         "idwt_stream_unfused"])
@pytest.mark.parametrize("name", sorted(streams.CASES))
def test_frames_match_oracle(dec, orc, name, mode):
    data, kw = streams.get(name)
    dec.set_int("idwt_mode", min(mode, 3))
    dec.set_int("fuse_pack", 0 if mode == 4 else 1)
    dec.set_int("bitexact", kw.get("bitexact", 0))
    dec.set_int("reduction_factor", kw.get("reduction_factor", 0))
    try:
        info_o, planes_o, consumed_o = orc.decode(data, **kw)
        info, planes, consumed, st = dec.decode(data)
    finally:
        dec.set_int("bitexact", 0)
        dec.set_int("reduction_factor", 0)
        dec.set_int("idwt_mode", 3)
        dec.set_int("fuse_pack", 1)
    assert (info.width, info.height, info.pix_fmt, info.bits_per_raw_sample) == \
           (info_o.width, info_o.height, info_o.pix_fmt, info_o.bits_per_raw_sample)
    assert consumed == consumed_o
    assert st.n_block_errors == orc.block_errors() == 0
    for a, b in zip(planes, planes_o):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("name", ["gray_l5_cb64", "rgb_mct", "gray_97_q2", "yuv422p12_97", "gray_3passes",
                                  "gray_97_bitexact", "placeholder_2_3p", "gray_l3_cb256x16", "gray_l2_cb4x1024"])
def test_stage_planes_match_oracle(dec, orc, name):
    """coefficient planes after HT decode + dequantisation, and after the IDWT"""
    data, kw = streams.get(name)
    dec.set_int("bitexact", kw.get("bitexact", 0))
    try:
        orc.decode_blocks(data, **kw)
        job = dec.job().parse(data).upload().run(1).wait()
        ntc = job.num_tilecomps()
        assert ntc == orc.num_tilecomps()
        for tc in range(ntc):
            a, b = job.plane(tc), orc.plane(tc)
            assert a.dtype == b.dtype
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "dequantised plane %d" % tc
        orc.idwt()
        for mode in (0, 1, 3):
            dec.set_int("idwt_mode", mode)
            job.run(1).run(2).wait()
            for tc in range(ntc):
                a, b = job.plane(tc), orc.plane(tc)
                if a.dtype == np.float32:
                    assert _ulp_diff(a, b) <= 1            # north-star tolerance for 9/7 float
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "idwt plane %d mode %d" % (tc, mode)
        job.free()
    finally:
        dec.set_int("bitexact", 0)
        dec.set_int("idwt_mode", 3)


@pytest.mark.parametrize("fuse", [1, 0], ids=["fused", "unfused"])
def test_batch_job_of_mixed_frames(dec, orc, fuse):
    """one job = a batch of frames of different formats: every stage is one launch over merged
    descriptor tables; fusable tiles (rgb24, gray, 4:4:4 planar) and unfusable ones (4:2:0, 4:2:2
    12-bit 9/7, 16-bit, tiles with odd origins) share the launches of a level"""
    names = ["rgb_mct", "gray_l5_cb64", "yuv420p8", "rgb_97_ict", "yuv422p12_97", "gray16", "rgb_tiles_offsets",
             "rgb10_mct", "rgba8", "gray_3passes", "tiny_3x1_l2", "rgb_mct"]
    names = [n for n in names if not streams.get(n)[1]]           # default decoder options only
    pkts = [streams.get(n)[0] for n in names]
    dec.set_int("idwt_mode", 3)
    dec.set_int("fuse_pack", fuse)
    try:
        job = dec.job().parse_batch(pkts).upload().run().wait()
        assert job.block_errors() == 0
        launches = job.idwt_launches()
        hbm = job.idwt_hbm_bytes()
        assert len(launches) == len(hbm) > 0
        for (ms, by), hb in zip(launches, hbm):
            assert ms >= 0 and 0 < hb <= by                     # a fused final level moves less than 2*4*lh*lv
        if fuse:
            assert any(hb < by for (ms, by), hb in zip(launches, hbm))
        else:
            assert all(hb == by for (ms, by), hb in zip(launches, hbm))
        import ffmpeg_ht_amd as m
        fr0 = m.Frame()
        assert dec.L.htj2k_job_device_frame(dec.h, job.h, 0, ctypes.byref(fr0)) == 0
        ls = ctypes.c_int(0)
        assert fr0.data[0] == dec.L.htj2k_job_device_plane(job.h, 0, ctypes.byref(ls)) and fr0.linesize[0] == ls.value
        for f, name in enumerate(names):
            info_o, planes_o, _ = orc.decode(pkts[f])
            info, planes = job.download_frame(f)
            assert (info.width, info.height, info.pix_fmt) == (info_o.width, info_o.height, info_o.pix_fmt), name
            for a, b in zip(planes, planes_o):
                assert np.array_equal(a, b), name
        job.free()
    finally:
        dec.set_int("fuse_pack", 1)


def test_batch_of_many_tiles(dec, orc):
    """a job may hold any number of tile-components (42 tiles x 3 components x 3 frames here)"""
    data, kw = streams.get("yuv420_42_tiles")
    info_o, planes_o, _ = orc.decode(data)
    job = dec.job().parse_batch([data, data, data]).upload().run().wait()
    assert job.num_tilecomps() == 3 * 126
    for f in range(3):
        info, planes = job.download_frame(f)
        for a, b in zip(planes, planes_o):
            assert np.array_equal(a, b)
    job.free()


def test_random_configurations(dec, orc):
    """tools/gpu_random_configs.py as a regression test: 300 random small codestreams (sizes, levels, block shapes,
    depths, subsampling, 5/3 / 9/7 / 9/7 fixed point, HT / Part-1 / MIXED, tiles, offsets, lowres), each decoded
    twice in one job, against the oracle"""
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(HERE), "tools", "gpu_random_configs.py")
    for env, seed in ({}, "5"), ({"C16BIAS": "1"}, "6"):           # the second draw: mostly jobs that take the 16-bit sub-band path
        r = subprocess.run([sys.executable, tool, "300", seed], capture_output=True, text=True, timeout=900, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
        assert "'bad': 0" in r.stdout.splitlines()[-1]


def test_pipeline_in_order_with_bad_packets(dec, orc):
    """htj2k_pipe_*: frames come back in send order, bit-identical to the oracle; a packet that does
    not parse costs its own frame only (the rest of its batch is decoded one by one)"""
    import ffmpeg_ht_amd as m
    names = [n for n in sorted(streams.CASES) if not streams.get(n)[1]]
    pkts = [streams.get(n)[0] for n in names]
    good = pkts[0]
    items = []                                           # (packet, expect_ok)
    for i, p in enumerate(pkts):
        items.append((p, True))
        if i == 5:
            items.append((b"\x00\x01garbage" * 9, False))
        if i == 9:
            items.append((good[:len(good) // 3], False))  # truncated body
    pipe = dec.pipe(batch=4, depth=2)
    try:
        got, sent = [], 0
        def drain_one():
            try:
                r = pipe.receive()
            except m.Htj2kError as e:
                got.append(("err", e.code))
                return True
            if r is None:
                return False
            got.append(r)
            return True
        keep = [m.packet(pkt) for pkt, _ in items]        # odd packets go in by reference (htj2k_pipe_send_ref)
        for i, (pkt, _) in enumerate(items):
            while not pipe.send(keep[i] if i & 1 else pkt):
                assert drain_one()
            sent += 1
        pipe.flush()
        while len(got) < sent:
            assert drain_one()
        assert pipe.receive() is None
    finally:
        pipe.close()
    assert len(got) == len(items)
    for (pkt, ok), g in zip(items, got):
        if not ok:
            assert g[0] == "err" and g[1] < 0
            continue
        info, planes = g
        info_o, planes_o, _ = orc.decode(pkt)
        assert (info.width, info.height, info.pix_fmt) == (info_o.width, info_o.height, info_o.pix_fmt)
        for a, b in zip(planes, planes_o):
            assert np.array_equal(a, b)


def test_idwt_random_borders(dec):
    """the reference's own DWT unit test shape (libavcodec/tests/jpeg2000dwt.c): random
    borders incl. odd origins, 1..3 sample lines, levels deeper than the size allows"""
    rng = np.random.default_rng(1234)
    cases = [([[151, 170], [140, 183]], 15), ([[1, 4], [1, 3]], 2), ([[5, 6], [3, 20]], 2), ([[0, 1], [0, 1]], 3),
             ([[1, 2], [1, 2]], 1), ([[0, 2], [0, 2]], 1), ([[3, 6], [2, 4]], 4), ([[0, 257], [1, 130]], 6),
             ([[3, 1003], [5, 705]], 3), ([[0, 976], [1, 300]], 2), ([[7, 495], [0, 2]], 1), ([[0, 2], [3, 400]], 2)]
    for _ in range(24):
        x0, y0 = int(rng.integers(0, 40)), int(rng.integers(0, 40))
        cases.append(([[x0, x0 + int(rng.integers(1, 260))], [y0, y0 + int(rng.integers(1, 200))]], int(rng.integers(1, 12))))
    for border, lev in cases:
        w, h = border[0][1] - border[0][0], border[1][1] - border[1][0]
        for typ in (1, 0, 2):
            if typ == 0:
                p = (rng.standard_normal((h, w)) * 300).astype(np.float32)
            else:
                p = rng.integers(-3000, 3000, (h, w)).astype(np.int32) * (256 if typ == 2 else 1)
            want = oracle.idwt(p, border, lev, typ)
            for mode in (0, 1, 3):
                dec.set_int("idwt_mode", mode)
                got = dec.idwt(p, border, lev, typ)
                assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (border, lev, typ, mode)
    dec.set_int("idwt_mode", 3)


def test_idwt_53_wraparound(dec):
    """unsigned wrap-around adds / arithmetic shifts of sr_1d53 (jpeg2000dwt.c:321-324)"""
    rng = np.random.default_rng(5)
    p = rng.integers(-2**31, 2**31 - 1, (70, 90), dtype=np.int64).astype(np.int32)
    border = [[1, 91], [0, 70]]
    want = oracle.idwt(p, border, 3, 1)
    for mode in (0, 1, 3):
        dec.set_int("idwt_mode", mode)
        assert np.array_equal(dec.idwt(p, border, 3, 1), want)
    dec.set_int("idwt_mode", 3)


def test_mct_kernels(dec):
    """Jpeg2000DSPContext.mct_decode[] (checkasm jpeg2000dsp: rct_int memcmp, ict_float)"""
    rng = np.random.default_rng(9)
    n = 512 * 37 + 5
    ints = [rng.integers(-2**20, 2**20, n).astype(np.int32) for _ in range(3)]
    for typ in (1, 2):
        want = oracle.mct(typ, *ints)
        got = dec.mct(typ, *ints)
        for a, b in zip(got, want):
            assert np.array_equal(a, b)
    fl = [(rng.standard_normal(n) * 500).astype(np.float32) for _ in range(3)]
    want = oracle.mct(0, *fl)
    got = dec.mct(0, *fl)
    for a, b in zip(got, want):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def _block_case(rng, w, h, passes, amp, causal=False, M_b=None, density=1.0):
    vals = rng.integers(-amp, amp + 1, (h, w))
    if density < 1.0:
        vals = np.where(rng.random((h, w)) < density, vals, 0)
    data, lcup, lref, maxU = vecgen.encode_block(vals, passes, causal)
    p = 1 if passes > 1 else 0
    M_b = M_b or max(maxU + p, 1) + 1
    return data, lcup, lref, passes, M_b - 1 - p, M_b, causal


def test_ht_block_decoder_unit(dec):
    """ff_jpeg2000_decode_htj2k + dequantization_int on raw cleanup/refinement segments:
    block shapes incl. odd sizes, 1-wide, 1024x4, sparse and dense, every pass count"""
    import ffmpeg_ht_amd as m
    rng = np.random.default_rng(77)
    shapes = [(64, 64), (32, 32), (63, 61), (1, 1), (2, 2), (3, 5), (1, 40), (40, 1), (1024, 4), (4, 1024),
              (128, 32), (17, 200), (64, 3)]
    descs, pool, expect, off, soff = [], b"", [], 0, 0
    for (w, h) in shapes:
        for passes in (1, 2, 3):
            for amp, density in ((1, 0.05), (3, 0.5), (200, 1.0), (30000, 1.0)):
                data, lcup, lref, npasses, zbp, M_b, causal = _block_case(rng, w, h, passes, amp, causal=(passes == 3 and w % 2 == 0),
                                                                          density=density)
                ret, t1 = oracle.ht_decode_block(data, lcup, lref, npasses, zbp, w, h, M_b, vsc=causal)
                assert ret == 1
                want = np.zeros((h, w), dtype=np.int32)
                oracle.lib().orc_dequant_int(t1.ctypes.data_as(ctypes.c_void_p), w, want.ctypes.data_as(ctypes.c_void_p), w, w, h, M_b, 32768)
                d = m.BlockDesc()
                d.data_off, d.plane_off, d.lcup, d.lref, d.w, d.h, d.stride = off, soff, lcup, lref, w, h, w
                d.npasses, d.zbp, d.M_b, d.flags, d.roi_shift, d.f_step, d.i_step = npasses, zbp, M_b, (8 if causal else 0) | 1, 0, 1.0, 32768
                descs.append(d)
                padded = data + b"\0" * ((-len(data)) % 16)
                pool += padded
                off += len(padded)
                expect.append((soff, want))
                soff += w * h
    # ... through the un-stuffing kernel with one, two and four blocks per wavefront (the jobs' choice by block width;
    # the unit entry runs one per wave unless told otherwise): streams of every length against the groups' pass sizes
    try:
        for g in ("1", "2", "4"):
            os.environ["HTJ2K_UNSTUFF_G"] = g
            got, status = dec.ht_blocks(descs, pool, soff)
            assert not status.any(), g
            for (o, want) in expect:
                assert np.array_equal(got[o:o + want.size].reshape(want.shape), want), g
    finally:
        os.environ.pop("HTJ2K_UNSTUFF_G", None)


def test_ht_block_errors_zero_the_block(dec):
    """Lcup < 2, bad Scup, exponent bound above maxbp (jpeg2000htdec.c:1252,1268,715)"""
    import ffmpeg_ht_amd as m
    rng = np.random.default_rng(3)
    vals = rng.integers(-200, 201, (32, 32))
    data, lcup, lref, maxU = vecgen.encode_block(vals, 1)
    cases = []
    cases.append((data, 1, 0, 1, 9, 10))                       # Lcup < 2
    bad = bytearray(data); bad[lcup - 1] = 0xFF                 # Scup > Lcup
    cases.append((bytes(bad), lcup, 0, 1, 9, 10))
    cases.append((data, lcup, 0, 1, 2, 3))                     # zbp too small for the coded magnitudes: U > maxbp
    descs, pool, off = [], b"", 0
    for i, (dat, lc, lr, npass, zbp, M_b) in enumerate(cases):
        ret, _ = oracle.ht_decode_block(dat, lc, lr, npass, zbp, 32, 32, M_b)
        assert ret < 0
        d = m.BlockDesc()
        d.data_off, d.plane_off, d.lcup, d.lref, d.w, d.h, d.stride = off, i * 1024, lc, lr, 32, 32, 32
        d.npasses, d.zbp, d.M_b, d.flags, d.f_step, d.i_step = npass, zbp, M_b, 1, 1.0, 32768
        descs.append(d)
        padded = dat + b"\0" * ((-len(dat)) % 16)
        pool += padded
        off += len(padded)
    got, status = dec.ht_blocks(descs, pool, 1024 * len(cases))
    assert status.all()
    assert not got.any()


def test_corrupt_frame_block_is_zeroed(dec, orc):
    data = bytearray(vecgen.encode(streams._img(64, 64, 1, 8, 3), nlevels=0))
    data[-3] = 0xFF
    info_o, planes_o, _ = orc.decode(bytes(data))
    info, planes, _, st = dec.decode(bytes(data))
    assert st.n_block_errors == 1 == orc.block_errors()
    assert np.array_equal(planes[0], planes_o[0])


def test_truncated_and_garbage_streams_do_not_fault(dec, orc):
    """frame-fatal errors come back as the reference's AVERROR codes; nothing hangs or faults"""
    import ffmpeg_ht_amd as m
    data, _ = streams.get("rgb_mct")
    for cut in (10, 60, 200, len(data) // 3, len(data) - 7):
        try:
            orc.decode(data[:cut])
            code_o = 0
        except oracle.DecodeError as e:
            code_o = e.code
        try:
            dec.decode(data[:cut])
            code = 0
        except m.Htj2kError as e:
            code = e.code
        assert code == code_o
    # bit flips inside the body: both decoders must agree on a frame (blocks may be rejected)
    rng = np.random.default_rng(11)
    for _ in range(6):
        bad = bytearray(data)
        for pos in rng.integers(300, len(bad) - 2, 8):
            bad[pos] ^= 1 << int(rng.integers(0, 8))
        try:
            info_o, planes_o, _ = orc.decode(bytes(bad))
        except oracle.DecodeError as e:
            with pytest.raises(m.Htj2kError):
                dec.decode(bytes(bad))
            continue
        info, planes, _, st = dec.decode(bytes(bad))      # must return; pixel parity on garbage is not required
        assert planes[0].shape == planes_o[0].shape


# ---------------------------------------------------------------- 16-bit sub-bands between block decoder and IDWT
def _coef16_streams():
    """8-bit RGB + RCT frames whose every IDWT level has the streaming kernels' fast geometry (even origin, widths a
    multiple of 4): the jobs that qualify for 16-bit sub-bands"""
    out = []
    for (w, h, nl, kw) in [(256, 192, 4, {}), (512, 256, 5, {}), (64, 48, 2, {}), (384, 160, 3, dict(cb=(5, 5))),
                           (256, 128, 3, dict(tile=(128, 64))), (320, 200, 1, {})]:
        img = vecgen.synth_image(w, h, 3, seed=w + h, noise=10)
        out.append(((w, h, nl), img, vecgen.encode(img, mct=1, nlevels=nl, **kw)))
    # flat areas: blocks without any pass are zero-filled by the block decoder
    img = [np.where(np.add.outer(np.arange(192), np.arange(256)) < 200, 90, c).astype(np.int32) for c in vecgen.synth_image(256, 192, 3, seed=3)]
    out.append(((256, 192, 4), img, vecgen.encode(img, mct=1, nlevels=4)))
    # the other fast stores: one 8-bit plane per component (gray, 4:2:0) and interleaved 16-bit (rgb48 from 10 bits)
    img = vecgen.synth_image(256, 192, 1, seed=11)
    out.append((("gray8", 256, 192), img, vecgen.encode(img, nlevels=4)))
    img = vecgen.synth_image(512, 256, 3, seed=12, dx=[1, 2, 2], dy=[1, 2, 2])
    out.append((("yuv420p8", 512, 256), img, vecgen.encode(img, nlevels=3, dx=[1, 2, 2], dy=[1, 2, 2], width=512, height=256)))
    img = vecgen.synth_image(256, 128, 3, depth=10, seed=13, noise=20)
    out.append((("rgb48", 256, 128), img, vecgen.encode(img, depth=10, mct=1, nlevels=3)))
    return out


def test_coef16_jobs_match_int32_jobs_and_the_oracle(dec, orc):
    """reversible jobs whose every band has M_b <= 15 (so every coefficient fits int16) and whose levels and frame
    qualify for a fast store (rgb24, rgb48, one 8-bit plane per component) keep their sub-bands as int16 between
    the block decoder and the IDWT: same pixels as the int32 layout, as the oracle, and as the source"""
    for key, img, data in _coef16_streams():
        info_o, planes_o, _ = orc.decode(data)
        res = {}
        # (coef16, ht_pair): 16-bit sub-bands with k_ht_decode_pair (two blocks per wave, the default where it applies),
        # with k_ht_decode<true>, and the int32 layout
        for knob in ((1, 1), (1, 0), (0, 1)):
            dec.set_int("coef16", knob[0])
            dec.set_int("ht_pair", knob[1])
            job = dec.job().parse_batch([data, data, data]).upload().run().wait()
            assert job.coef16() == bool(knob[0]), (key, knob)
            assert job.block_errors() == 0
            res[knob] = [job.download_frame(f)[1] for f in range(3)]
            job.free()
        dec.set_int("coef16", 1)
        dec.set_int("ht_pair", 1)
        for f in range(3):
            for a, b, c, d in zip(res[(1, 1)][f], res[(1, 0)][f], res[(0, 1)][f], planes_o):
                assert np.array_equal(a, b) and np.array_equal(a, c) and np.array_equal(a, d), key
        if isinstance(key[0], int):                                  # 8-bit RGB: lossless against the source as well
            assert np.array_equal(res[(1, 1)][0][0].reshape(info_o.height, info_o.width, 3), np.stack(img, -1)), key


def test_multi_block_kernel_matches_the_column_kernel_and_the_oracle(dec, orc):
    """jobs with 32-bit sub-bands whose HT blocks all are at most 64 columns wide, without ROI shift and of one transform
    (with or without SigProp / MagRef passes) decode 2 or 4 blocks per wavefront with a lane per quad (k_ht_decode_multi); the same streams with a
    block per wavefront and a lane per sample column (k_ht_decode<true>, knob ht_multi 0) and the oracle give the same
    frames.  The catalogue covers 5/3, 9/7 float and fixed point, 8 to 16 bits, odd widths and heights, tiles, offsets,
    placeholder passes, 32- and 64-column blocks; streams that do not qualify must not take the kernel."""
    seen = set()
    dec.set_int("coef16", 0)                                   # so that the 8-bit streams qualify as well
    try:
        for name in sorted(streams.CASES):
            data, kw = streams.get(name)
            if kw.get("reduction_factor") or kw.get("bitexact"):
                continue
            info_o, planes_o, _ = orc.decode(data, **kw)
            res = {}
            for knob in (1, 0):
                dec.set_int("ht_multi", knob)
                job = dec.job().parse_batch([data, data]).upload().run().wait()
                res[knob] = ([job.download_frame(f)[1] for f in range(2)], job.ht_blocks_per_wave(), job.block_errors())
                job.free()
            assert res[0][1] in (0, 1), name                     # 0: no HT block in the stream
            seen.add(res[1][1])
            assert res[1][2] == res[0][2] == orc.block_errors(), name
            for f in range(2):
                for a, b, c in zip(res[1][0][f], res[0][0][f], planes_o):
                    assert np.array_equal(a, b) and np.array_equal(a, c), (name, res[1][1])
            if "roi" in name:
                assert res[1][1] in (0, 1), name
    finally:
        dec.set_int("ht_multi", 1)
        dec.set_int("coef16", 1)
    assert {1, 2, 4} <= seen, seen


def test_plt_streams_decode_the_same_with_packet_threads(dec, orc):
    """knob packet_threads: a frame parsed on its own has the packets of tiles with a PLT list read by several threads;
    same frames as without, as the oracle's, for every container variant of a stream with many packets"""
    import cs_rewrite
    img = streams._img(640, 480, 3, 8, 31)
    cs = vecgen.encode(img, sop=True, eph=True, mct=1, nlevels=4, prec=[(7, 7), (6, 6)], cb=(5, 5))
    try:
        for vn, data in cs_rewrite.variants(cs, True):
            data = bytes(data)
            if "ppm_tp3" in vn:                              # (a reference quirk decodes this one differently: tests/test_plan_equality.py)
                continue
            try:
                info_o, planes_o, _ = orc.decode(data)
            except oracle.DecodeError:                       # (more tile-parts than the reference takes: both must refuse)
                planes_o = None
            for th in (1, 4):
                dec.set_int("packet_threads", th)
                if planes_o is None:
                    with pytest.raises(Exception):
                        dec.decode(data)
                    continue
                info, planes, consumed, st = dec.decode(data)
                assert st.n_block_errors == 0, vn
                for a, b in zip(planes, planes_o):
                    assert np.array_equal(a, b), (vn, th)
    finally:
        dec.set_int("packet_threads", 1)


def test_coef16_is_not_used_where_it_does_not_apply(dec, orc):
    """odd geometry, 16-bit samples (M_b > 15), 9/7, refinement passes, Part-1 blocks, staged runs: int32 sub-bands as before"""
    for name in ("rgb_mct", "gray16", "rgb_97_ict", "rgb_3passes_cb32", "p1_rgb_mct", "gray_l5_cb64"):
        data, kw = streams.get(name)
        job = dec.job().parse_batch([data]).upload().run().wait()
        assert not job.coef16(), name
        job.free()
    key, img, data = _coef16_streams()[0]
    job = dec.job().parse_batch([data]).upload()
    job.run(1)
    job.run(6)
    job.wait()
    assert not job.coef16()
    info, planes = job.download_frame(0)
    assert np.array_equal(planes[0].reshape(192, 256, 3), np.stack(img, -1))
    job.free()


def test_coef16_rejected_block_is_zeroed(dec, orc):
    key, img, data = _coef16_streams()[0]
    bad = bytearray(data)
    bad[-3] = 0xFF                                               # the last block's Scup becomes invalid
    info_o, planes_o, _ = orc.decode(bytes(bad))
    assert orc.block_errors() == 1
    job = dec.job().parse_batch([bytes(bad)]).upload().run().wait()
    assert job.coef16() and job.block_errors() == 1
    info, planes = job.download_frame(0)
    assert np.array_equal(planes[0], planes_o[0])
    job.free()


def test_ll16_jobs_match_int32_ll_bands_and_the_oracle(dec, orc):
    """knob "ll16" (default on): jobs with 16-bit sub-bands also write the LL bands between the IDWT levels as 16-bit samples: same
    pixels as with int32 LL bands, as the oracle, as the source; the pipeline entry points take the same path"""
    try:
        for key, img, data in _coef16_streams():
            info_o, planes_o, _ = orc.decode(data)
            res = {}
            for ll16 in (1, 0):
                dec.set_int("ll16", ll16)
                job = dec.job().parse_batch([data, data]).upload().run().wait()
                assert job.coef16() and job.block_errors() == 0
                assert job.ll16() == ll16, (key, ll16, job.ll16())
                res[ll16] = [job.download_frame(f)[1] for f in range(2)]
                job.free()
            for f in range(2):
                for a, b, d in zip(res[1][f], res[0][f], planes_o):
                    assert np.array_equal(a, b) and np.array_equal(a, d), key
            dec.set_int("ll16", 1)
            info, planes, _, st = dec.decode(data)                    # htj2k_decode: run + download, no wait in between
            assert all(np.array_equal(a, d) for a, d in zip(planes, planes_o)), key
    finally:
        dec.set_int("ll16", 1)


def test_first_three_idwt_levels_in_one_launch(dec, orc):
    """knob "idwt_x3" (default on): jobs with 16-bit LL bands run the first three 5/3 levels of every plane as ONE launch
    whose intermediate LL bands stay in LDS (k_idwt_stream_ll16_x3).  Same frames as with one launch per level, as the
    oracle and as the source -- over heights that leave odd row counts and short last bands at every level, band heights
    from 8 rows up (HTJ2K_X3_TH), several frames per job, mixed plane sizes (4:2:0), and with the overflow check of the
    16-bit LL bands firing (ll16_test_bits)"""
    cases = [(256, 192, 3, 4, {}), (512, 256, 3, 5, {}), (1920, 1080, 3, 5, {}), (1024, 542, 1, 5, {}), (256, 46, 1, 4, {}),
             (1024, 70, 3, 5, dict(cb=(5, 5))), (512, 258, 1, 4, {})]
    try:
        for (w, h, nc, nl, kw) in cases:
            img = vecgen.synth_image(w, h, nc, seed=w + h + nl, noise=10)
            data = vecgen.encode(img, mct=1 if nc == 3 else 0, nlevels=nl, **kw)
            info_o, planes_o, _ = orc.decode(data)
            got = {}
            for x3, th in ((0, 36), (1, 36), (1, 8), (1, 20)):
                dec.set_int("idwt_x3", x3)
                os.environ["HTJ2K_X3_TH"] = str(th)
                job = dec.job().parse_batch([data, data, data]).upload().run().wait()
                assert job.coef16() and job.ll16() == 1 and job.block_errors() == 0, (w, h, x3)
                n_launch = len(job.idwt_launches())
                assert n_launch == (nl if x3 == 0 else nl - 2), (w, h, nl, x3, n_launch)     # levels 0-2 as one launch
                got[(x3, th)] = [job.download_frame(f)[1] for f in range(3)]
                job.free()
            for k, frames in got.items():
                for f in range(3):
                    assert all(np.array_equal(a, d) for a, d in zip(frames[f], planes_o)), (w, h, k, f)
            if nc == 3:
                assert np.array_equal(got[(1, 36)][0][0].reshape(h, w, 3), np.stack(img, -1))
        # 4:2:0: planes of two sizes in one launch
        img = vecgen.synth_image(1024, 512, 3, seed=12, dx=[1, 2, 2], dy=[1, 2, 2])
        data = vecgen.encode(img, nlevels=4, dx=[1, 2, 2], dy=[1, 2, 2], width=1024, height=512)
        info_o, planes_o, _ = orc.decode(data)
        for x3 in (0, 1):
            dec.set_int("idwt_x3", x3)
            job = dec.job().parse_batch([data, data]).upload().run().wait()
            assert job.ll16() == 1
            for f in range(2):
                assert all(np.array_equal(a, d) for a, d in zip(job.download_frame(f)[1], planes_o)), ("420", x3, f)
            job.free()
        # the range check of what a level stores also guards the bands that stay in LDS
        dec.set_int("idwt_x3", 1)
        dec.set_int("ll16_test_bits", 6)
        img = vecgen.synth_image(512, 256, 3, seed=5, noise=10)
        data = vecgen.encode(img, mct=1, nlevels=5)
        info_o, planes_o, _ = orc.decode(data)
        job = dec.job().parse_batch([data]).upload().run().wait()
        assert job.ll16() == 2                                           # overflowed: the transform ran again with int32 LL bands
        assert all(np.array_equal(a, d) for a, d in zip(job.download_frame(0)[1], planes_o))
        job.free()
    finally:
        dec.set_int("idwt_x3", 1)
        dec.set_int("ll16_test_bits", 16)
        os.environ.pop("HTJ2K_X3_TH", None)


def test_final_level_on_pairs_of_16_bit_samples(dec, orc):
    """knob "idwt_pk" (default on): the final 5/3 level of 8-bit pictures -- lifting, inverse RCT, clip, rgb24 / 8-bit plane
    store -- runs on pairs of 16-bit samples (v_pk_* instructions) where interval arithmetic over the bands' M_b and the
    checked range of the LL band shows that no intermediate can leave 16 bits.  Same frames as with 32-bit arithmetic, as
    the oracle and as the source: RGB with and without the RCT, gray, 4:2:0 planes, one level (the LL band is the block
    decoder's, unchecked) up to five, line ends inside a wave (mirror lanes), several frames per job; deeper pictures do
    not qualify; and with the tightened range check firing the job falls back to the 32-bit kernels."""
    cases = [(256, 192, 3, 1, 1), (512, 256, 3, 1, 5), (1920, 1080, 3, 1, 5), (1024, 542, 1, 0, 5), (256, 46, 1, 0, 2), (512, 258, 3, 0, 3),
             (64, 64, 3, 1, 1), (3840, 2160, 3, 1, 5)]
    try:
        for (w, h, nc, mct, nl) in cases:
            img = vecgen.synth_image(w, h, nc, seed=w + h + nl, noise=10)
            data = vecgen.encode(img, mct=mct, nlevels=nl)
            info_o, planes_o, _ = orc.decode(data)
            for pk in (1, 0):
                dec.set_int("idwt_pk", pk)
                job = dec.job().parse_batch([data, data]).upload().run().wait()
                assert job.coef16() and job.block_errors() == 0, (w, h, pk)
                used = job.idwt_packed()
                assert (10 <= used <= 16) if pk else used == 0, (w, h, nc, mct, nl, pk, used)
                for f in range(2):
                    assert all(np.array_equal(a, d) for a, d in zip(job.download_frame(f)[1], planes_o)), (w, h, nc, mct, nl, pk, f)
                job.free()
            if nc == 3:
                dec.set_int("idwt_pk", 1)
                info, planes, _, st = dec.decode(data)
                assert np.array_equal(planes[0].reshape(h, w, 3), np.stack(img, -1))
        dec.set_int("idwt_pk", 1)
        # 4:2:0: three single-plane groups of two sizes
        img = vecgen.synth_image(1024, 512, 3, seed=12, dx=[1, 2, 2], dy=[1, 2, 2])
        data = vecgen.encode(img, nlevels=4, dx=[1, 2, 2], dy=[1, 2, 2], width=1024, height=512)
        info_o, planes_o, _ = orc.decode(data)
        job = dec.job().parse_batch([data, data]).upload().run().wait()
        assert job.idwt_packed() >= 10
        for f in range(2):
            assert all(np.array_equal(a, d) for a, d in zip(job.download_frame(f)[1], planes_o)), ("420", f)
        job.free()
        # 10-bit components: 16-bit sub-bands, but not this kernel
        img = vecgen.synth_image(512, 256, 3, depth=10, seed=3, noise=10)
        data = vecgen.encode(img, mct=1, nlevels=4, depth=10)
        info_o, planes_o, _ = orc.decode(data)
        job = dec.job().parse_batch([data]).upload().run().wait()
        assert job.coef16() and job.idwt_packed() == 0
        assert all(np.array_equal(a, d) for a, d in zip(job.download_frame(0)[1], planes_o))
        job.free()
        # the LL bands are checked against the bits the packed kernel needs, and a job that fails runs again in 32 bits
        img = vecgen.synth_image(512, 256, 3, seed=5, noise=10)
        data = vecgen.encode(img, mct=1, nlevels=5)
        info_o, planes_o, _ = orc.decode(data)
        dec.set_int("ll16_test_bits", 6)
        job = dec.job().parse_batch([data]).upload().run().wait()
        assert job.ll16() == 2 and job.idwt_packed() == 0
        assert all(np.array_equal(a, d) for a, d in zip(job.download_frame(0)[1], planes_o))
        job.free()
    finally:
        dec.set_int("idwt_pk", 1)
        dec.set_int("ll16_test_bits", 16)


def test_packed_range_check_leaves_extreme_pictures_alone(dec):
    """The packed final level asks the LL bands of an 8-bit RGB picture to stay within 11 bits (+-1024); a picture that does
    not pays with a second run of the whole transform.  No picture should: full-amplitude noise, saturated colour bars,
    one- and two-pixel checkerboards and stripes in complementary colours, random saturated blocks -- the worst a forward
    transform can make of 8-bit samples -- all decode on the packed kernels at the first attempt, losslessly."""
    W, H = 1024, 512
    rng = np.random.default_rng(1)
    yy, xx = np.mgrid[0:H, 0:W]
    pats = {
        "noise": [rng.integers(0, 256, (H, W)).astype(np.int32) for _ in range(3)],
        "bars": [(((xx // 128) >> k) & 1).astype(np.int32) * 255 for k in range(3)],
        "checker": [(((xx + yy) & 1) * 255).astype(np.int32), (((xx + yy + 1) & 1) * 255).astype(np.int32), ((xx & 1) * 255).astype(np.int32)],
        "stripes": [(((xx >> 1) & 1) * 255).astype(np.int32), ((((xx >> 1) + 1) & 1) * 255).astype(np.int32), (((xx >> 1) & 1) * 255).astype(np.int32)],
        "blocks8": [np.kron(rng.integers(0, 2, (H // 8, W // 8)), np.ones((8, 8), int)).astype(np.int32) * 255 for _ in range(3)],
        "blocks32": [np.kron(rng.integers(0, 2, (H // 32, W // 32)), np.ones((32, 32), int)).astype(np.int32) * 255 for _ in range(3)],
    }
    for name, img in pats.items():
        data = vecgen.encode(img, mct=1, nlevels=5)
        job = dec.job().parse_batch([data]).upload().run().wait()
        assert job.coef16() and job.ll16() == 1 and job.idwt_packed() >= 10, (name, job.ll16(), job.idwt_packed())
        assert np.array_equal(job.download_frame(0)[1][0].reshape(H, W, 3), np.stack(img, -1)), name
        job.free()


def test_ll16_overflow_runs_the_transform_again(dec, orc):
    """Nothing bounds the LL bands of crafted or corrupt coefficient data, so the level kernels flag a sample that does
    not fit and the job repeats the IDWT with int32 LL bands before it hands out frames.  A stream produced by a forward
    transform cannot get there (every intermediate LL band is a low-pass of the picture), so the knob "ll16_test_bits"
    lowers the limit: with 6 bits ordinary frames overflow, and must come out exactly as before -- from wait + download,
    from download alone (htj2k_decode) and from the pipeline."""
    try:
        dec.set_int("ll16", 1)
        for key, img, data in _coef16_streams()[:4]:
            info_o, planes_o, _ = orc.decode(data)
            for bits, want in ((16, 1), (6, 2)):
                dec.set_int("ll16_test_bits", bits)
                job = dec.job().parse_batch([data, data]).upload().run().wait()
                assert job.coef16() and job.ll16() == want, (key, bits, job.ll16())
                for f in range(2):
                    assert all(np.array_equal(a, d) for a, d in zip(job.download_frame(f)[1], planes_o)), (key, bits)
                job.run().wait()                                  # and again: the job is back on 16-bit LL bands, overflows again
                assert job.ll16() == want
                assert all(np.array_equal(a, d) for a, d in zip(job.download_frame(1)[1], planes_o)), (key, bits)
                job.free()
                info, planes, _, st = dec.decode(data)
                assert all(np.array_equal(a, d) for a, d in zip(planes, planes_o)), (key, bits)
        dec.set_int("ll16_test_bits", 6)
        key, img, data = _coef16_streams()[1]
        info_o, planes_o, _ = orc.decode(data)
        pipe = dec.pipe(batch=3, depth=3)
        got = 0
        for i in range(7):
            assert pipe.send(data)
        pipe.flush()
        while True:
            r = pipe.receive()
            if r is None:
                break
            assert all(np.array_equal(a, d) for a, d in zip(r[1], planes_o))
            got += 1
        pipe.close()
        assert got == 7
    finally:
        dec.set_int("ll16_test_bits", 16)
        dec.set_int("ll16", 1)


# ---------------------------------------------------------------- Part-1 (MQ-coded) blocks: k_mq_decode
OPJ = np.load(os.path.join(HERE, "golden", "opj_part1.npz"))


@pytest.mark.parametrize("name", sorted(k[:-4] for k in OPJ.files if k.endswith(".j2k")))
def test_openjpeg_encoded_part1_streams(dec, orc, name):
    """streams written by a third-party encoder (OpenJPEG, tests/golden/make_openjpeg_part1.py): the device path
    equals the oracle bit for bit (9/7 included) and both equal the pixels stored with the stream"""
    data = OPJ[name + ".j2k"].tobytes()
    pix = OPJ[name + ".pix"]
    info_o, planes_o, _ = orc.decode(data)
    info, planes, _, st = dec.decode(data)
    assert st.n_block_errors == 0
    for a, b in zip(planes, planes_o):
        assert np.array_equal(a, b)
    tol = 0 if OPJ[name + ".lossless"][0] else 1
    assert np.abs(planes[0].reshape(pix.shape).astype(np.int64) - pix.astype(np.int64)).max() <= tol


def test_part1_block_decoder_unit(dec):
    """htj2k_mq_blocks = decode_cblk + dequantization_int on single blocks: shapes incl. 1-wide, 1024 x 4, 4 x 1024 and
    odd sizes, every mode switch, ROI up-shift, and the two ways decode_cblk() fails part-way (the passes decoded up to
    there stay and are dequantised, jpeg2000dec.c:2275-2290)"""
    import ffmpeg_ht_amd as m
    rng = np.random.default_rng(78)
    descs, pool, expect, soff = [], b"", [], 0

    def add(vals, seg, lens, passes, K, npasses, style, band, M_b, roi=0, npasses_sig=None, drop_starts=0):
        nonlocal pool, soff
        h, w = vals.shape
        data, length, starts = oracle.mq_block_layout(seg, lens, passes, style)
        npz = npasses if npasses_sig is None else npasses_sig
        if npz > npasses:
            starts = starts + [length] * (npz - npasses)
        if drop_starts:
            starts = starts[:-drop_starts]
        ret, t1 = oracle.mq_decode_block(data, length, npz, K, w, h, M_b, style, band, starts, roi_shift=roi)
        want = np.zeros((h, w), dtype=np.int32)
        oracle.lib().orc_dequant_int(t1.ctypes.data_as(ctypes.c_void_p), w, want.ctypes.data_as(ctypes.c_void_p), w, w, h, M_b, 32768)
        d = m.BlockDesc()
        d.data_off, d.plane_off, d.lcup, d.lref, d.w, d.h, d.stride = len(pool), soff, length, len(starts), w, h, w
        d.npasses, d.zbp, d.M_b, d.flags, d.roi_shift, d.f_step, d.i_step = npz, K, M_b, 4 | 1, roi, 1.0, 32768
        descs.append(d)
        pool += oracle.mq_block_region(data, length, style, band, starts)
        expect.append((soff, want, ret))
        soff += w * h

    import vecgen
    for style in (0, 0x01, 0x04, 0x08, 0x2F):
        for (w, h) in [(64, 64), (32, 32), (63, 61), (1, 1), (3, 5), (1, 40), (40, 1), (1024, 4), (4, 1024), (128, 32), (17, 200)]:
            for amp, density in ((1, 0.05), (200, 1.0), (30000, 1.0)):
                band = int(rng.integers(0, 4))
                vals = rng.integers(-amp, amp + 1, (h, w)) * (rng.random((h, w)) < density)
                vals[0, 0] = amp
                seg, lens, passes, K, npasses = vecgen.encode_block_p1(vals, band=band, style=style)
                add(vals, seg, lens, passes, K, npasses, style, band, K + 2)
    # ROI up-shift of the samples below the threshold (jpeg2000dec.c:2071-2086)
    vals = rng.integers(-9, 10, (32, 32))
    vals[0, 0] = 9
    seg, lens, passes, K, npasses = vecgen.encode_block_p1(vals, band=2, style=0)
    add(vals, seg, lens, passes, K + 3, npasses, 0, 2, K + 6, roi=3)
    # "bpno became invalid" part-way, "Missing needed termination" part-way, "bpno invalid" before the first pass
    vals = rng.integers(-50, 51, (32, 32))
    vals[0, 0] = 50
    seg, lens, passes, K, npasses = vecgen.encode_block_p1(vals, band=1, style=0x04)
    add(vals, seg, lens, passes, K, npasses, 0x04, 1, 28, npasses_sig=npasses + 9)
    add(vals, seg, lens, passes, K, npasses, 0x04, 1, K + 2, drop_starts=3)
    add(vals, seg, lens, passes, 31, npasses, 0x04, 1, 9)
    got, status = dec.mq_blocks(descs, pool, soff)
    nerr = 0
    for i, (o, want, ret) in enumerate(expect):
        assert (status[i] != 0) == (ret < 0), i
        nerr += ret < 0
        assert np.array_equal(got[o:o + want.size].reshape(want.shape), want), i
    assert nerr == 3


def test_part1_and_ht_frames_in_one_batch(dec, orc):
    """HT and Part-1 blocks share a job: the HT kernels take the front of the block table, k_mq_decode the rest"""
    names = ["p1_rgb_mct", "rgb_mct", "p1_bypass_termall", "gray_3passes", "p1_gray_cb4x1024", "p1_97", "yuv422p12_97",
             "p1_all_switches", "p1_truncated_2", "mixed_3passes_vsc", "mixed_rgb_cb32"]
    pkts = [streams.get(n)[0] for n in names]
    job = dec.job().parse_batch(pkts).upload().run().wait()
    assert job.block_errors() == 0
    for f, name in enumerate(names):
        info_o, planes_o, _ = orc.decode(pkts[f])
        info, planes = job.download_frame(f)
        assert (info.width, info.height, info.pix_fmt) == (info_o.width, info_o.height, info_o.pix_fmt), name
        for a, b in zip(planes, planes_o):
            assert np.array_equal(a, b), name
    job.free()


@pytest.mark.parametrize("name", ["p1_gray_cb32", "p1_bypass_termall", "p1_all_switches", "p1_rgb_mct", "p1_cb256x16_modes"])
def test_part1_corrupt_bodies_match_oracle(dec, orc, name):
    """the MQ decoder is total: on damaged code bytes (spurious markers, broken terminations, invalid bit-plane
    counts) it still produces samples, and the reference dequantises whatever decode_cblk() left behind
    (jpeg2000dec.c:2275-2290).  The device path must leave exactly the same pixels and error count."""
    import ffmpeg_ht_amd as m
    data, _ = streams.get(name)
    rng = np.random.default_rng(len(name))
    compared = 0
    for _ in range(12):
        bad = bytearray(data)
        for pos in rng.integers(200, len(bad) - 2, 10):
            bad[pos] = int(rng.integers(0, 256)) if rng.integers(0, 2) else 0xFF
        try:
            info_o, planes_o, _ = orc.decode(bytes(bad))
        except oracle.DecodeError as e:
            with pytest.raises(m.Htj2kError):
                dec.decode(bytes(bad))
            continue
        info, planes, _, st = dec.decode(bytes(bad))
        assert st.n_block_errors == orc.block_errors()
        for a, b in zip(planes, planes_o):
            assert np.array_equal(a, b)
        compared += 1
    assert compared > 0


def test_linesize_padding_is_respected(dec, orc):
    data, _ = streams.get("rgb_mct")
    info_o, planes_o, _ = orc.decode(data)
    info, planes, _, _ = dec.decode(data, align=64)
    assert np.array_equal(planes[0], planes_o[0])


def test_full_size_4k_properties(dec, orc):
    """BASELINE config 2 at full size (3840x2160 RGB 8-bit, 5/3 + RCT, 5 levels, 64x64):
    lossless round trip on the GPU, and a checksum of the frame against the oracle."""
    img = vecgen.synth_image(3840, 2160, 3, seed=2)
    data = vecgen.encode(img, mct=1)
    info, planes, _, st = dec.decode(data)
    assert st.n_codeblocks == 6321 and st.n_block_errors == 0
    got = planes[0].reshape(2160, 3840, 3)
    assert np.array_equal(got, np.stack(img, -1))
    info_o, planes_o, _ = orc.decode(data)
    assert oracle.framecrc(planes) == oracle.framecrc(planes_o)
    # single-frame calls on packets of this size stage and upload the packet in pieces on helper threads while the caller's
    # copy is parsed: other frames through the same buffers, a packet that is cut short (the parser's error, no hang), then
    # the first frame again
    img2 = vecgen.synth_image(3840, 2160, 3, seed=9, noise=3)
    data2 = vecgen.encode(img2, mct=1)
    for d, im in ((data2, img2), (data, img), (data2, img2)):
        info, planes, _, st = dec.decode(d)
        assert st.n_block_errors == 0 and np.array_equal(planes[0].reshape(2160, 3840, 3), np.stack(im, -1))
    import ffmpeg_ht_amd as m
    try:
        dec.decode(data[:len(data) // 2])
    except m.Htj2kError:
        pass
    info, planes, _, st = dec.decode(data)
    assert np.array_equal(planes[0].reshape(2160, 3840, 3), np.stack(img, -1))


def test_full_size_4k_422_irreversible(dec, orc):
    """BASELINE config 3: 4K 12-bit 4:2:2 9/7, 32x32 codeblocks (16 473 blocks)"""
    img = vecgen.synth_image(3840, 2160, 3, depth=12, seed=3, noise=30, dx=[1, 2, 2], dy=[1, 1, 1])
    data = vecgen.encode(img, depth=12, dx=[1, 2, 2], dy=[1, 1, 1], transform=0, qstep=1.0, cb=(5, 5), width=3840, height=2160)
    info, planes, _, st = dec.decode(data)
    assert st.n_codeblocks == 16473 and st.n_block_errors == 0
    info_o, planes_o, _ = orc.decode(data)
    for a, b in zip(planes, planes_o):
        assert np.array_equal(a, b)


def test_full_size_8k_16bit(dec, orc):
    """BASELINE config 4: 7680x4320 16-bit gray, 5/3, 6 levels (8 227 blocks, IDWT stress)"""
    img = vecgen.synth_image(7680, 4320, 1, depth=16, seed=4, noise=300)
    data = vecgen.encode(img, depth=16, nlevels=6)
    info, planes, _, st = dec.decode(data)
    assert st.n_codeblocks == 8227 and st.n_block_errors == 0
    assert np.array_equal(planes[0], img[0])


def test_full_size_4k_10bit_stream_sharded(dec, orc):
    """BASELINE config 5 in miniature: a stream of 4K 10-bit lossless RGB frames (rgb48, samples << 6), distinct
    seeds, sharded round-robin over ranks (bench.shard_frames) -- here every "rank" is the same GPU -- each rank's
    frames going through the pipeline; lossless round trip and a checksum against the oracle for one frame."""
    import bench
    nframes, world = 6, 2
    imgs = [vecgen.synth_image(3840, 2160, 3, depth=10, seed=50 + i, noise=20) for i in range(nframes)]
    pkts = [vecgen.encode(im, depth=10, mct=1) for im in imgs]
    seen = set()
    for rank in range(world):
        mine = bench.shard_frames(nframes, rank, world)
        pipe = dec.pipe(batch=2, depth=2)
        try:
            out, sent = [], 0
            while len(out) < len(mine):
                while sent < len(mine) and pipe.send(pkts[mine[sent]]):
                    sent += 1
                if sent == len(mine):
                    pipe.flush()
                out.append(pipe.receive())
        finally:
            pipe.close()
        for f, (info, planes) in zip(mine, out):
            seen.add(f)
            got = planes[0].reshape(2160, 3840, 3).astype(np.int64) >> 6       # write_frame_16: << (16 - 10)
            assert np.array_equal(got, np.stack(imgs[f], -1)), f
            if f == 1:
                _, planes_o, _ = orc.decode(pkts[f])
                assert oracle.framecrc(planes) == oracle.framecrc(planes_o)
    assert seen == set(range(nframes))


def test_c_example_program(orc, tmp_path):
    """examples/htj2k_decode.c: the C ABI driven from plain C (no Python in the loop), one-shot and pipelined"""
    import subprocess
    root = os.path.dirname(HERE)
    r = subprocess.run(["make", "-C", root, "examples"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    exe = os.path.join(root, "examples", "htj2k_decode")
    for name in ("rgb_mct", "yuv422p12_97", "gray_3passes"):
        data, kw = streams.get(name)
        src = tmp_path / (name + ".j2c")
        src.write_bytes(data)
        _, planes_o, _ = orc.decode(data, **kw)
        want = b"".join(np.ascontiguousarray(p).tobytes() for p in planes_o)
        for extra in ([], ["-p", "20"]):
            out = tmp_path / (name + ".raw")
            r = subprocess.run([exe] + extra + [str(src), str(out)], capture_output=True, text=True, timeout=300)
            assert r.returncode == 0, (r.stdout, r.stderr)
            assert out.read_bytes() == want, (name, extra)
    # -s: a sequence of different frames in one file, cut apart by htj2k_splitter_* in 64 KB reads
    names = ["rgb_mct", "p1_bypass_termall", "gray16", "mixed_rgb_cb32", "rgb_mct"]
    seq = tmp_path / "sequence.j2k"
    seq.write_bytes(b"".join(streams.get(n)[0] for n in names))
    want = b""
    for n in names:
        _, planes_o, _ = orc.decode(streams.get(n)[0])
        want += b"".join(np.ascontiguousarray(p).tobytes() for p in planes_o)
    out = tmp_path / "sequence.raw"
    r = subprocess.run([exe, "-s", str(seq), str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert r.stdout.count("frame ") == len(names), r.stdout
    assert out.read_bytes() == want
    # -x: the same frames as frame-wrapped picture elements of an MXF file, beside sound elements and fill
    import test_mxf
    mxf = tmp_path / "sequence.mxf"
    mxf.write_bytes(test_mxf._mxf([streams.get(n)[0] for n in names], run_in=b"\x00" * 37))
    out = tmp_path / "mxf.raw"
    r = subprocess.run([exe, "-x", str(mxf), str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert r.stdout.count("frame ") == len(names), r.stdout
    assert out.read_bytes() == want
    # ... and as one clip-wrapped element, cut apart by the splitter
    mxf.write_bytes(test_mxf._mxf([b"".join(streams.get(n)[0] for n in names)], picture_key=test_mxf.PICT_J2K_CLIP, forms=("b8",)))
    r = subprocess.run([exe, "-x", str(mxf), str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert r.stdout.count("frame ") == len(names), r.stdout
    assert out.read_bytes() == want


def test_device_gather_matches_host_gather_and_the_oracle(dec, orc):
    """"device_gather" (default): the packets are uploaded as they are and k_gather builds the byte pool from the parser's
    gather table; 0: the parser copies the code-block bytes on the host.  Same frames either way, and the oracle's --
    on single- and multi-piece blocks: HT, Part-1 with terminated segments (0xFF 0xFF + trailers), MIXED, quality layers
    (OpenJPEG fixtures), tile-parts / PPM / PPT variants (tests/cs_rewrite.py), and all of them in one batch."""
    import cs_rewrite
    z = np.load(os.path.join(HERE, "golden", "opj_part1.npz"))
    items = [(n, streams.get(n)[0]) for n in ("gray_l5_cb64", "rgb_tiles_offsets", "gray_3passes", "p1_bypass_termall", "p1_all_switches",
                                              "p1_rgb_tiles", "mixed_rgb_cb32", "tiny_1x1", "all_zero", "yuv420p8", "placeholder_2_3p")]
    items += [(k, z[k].tobytes()) for k in z.files if k.endswith(".j2k") and z[k[:-4] + ".lossless"][0]]
    base = vecgen.encode(streams._img(190, 131, 3, 8, 6), part1=True, cblk_style=0x05, tile=(64, 64), nlevels=3, sop=True, eph=True)
    items += [("p1_termall_tiles." + n, d) for n, d in cs_rewrite.variants(base, False) if n in ("tp3_tlm_plt", "ppm", "ppt_tp3", "coc_qcc_tile")]
    base = vecgen.encode(streams._img(190, 131, 3, 8, 5), mct=1, prog=2, prec=[(7, 7), (6, 6)], nlevels=3, sop=True, eph=True, cap_extra_bits=0x1800)
    items += [("ht_rgb_rpcl." + n, d) for n, d in cs_rewrite.variants(base, True) if n in ("tp_each_packet", "ppt", "ppm", "tp3_interleaved")]
    dec.set_int("bitexact", 0); dec.set_int("reduction_factor", 0)
    want = {}
    for name, data in items:
        want[name] = orc.decode(data)[1]
    try:
        for gather in (1, 0):
            dec.set_int("device_gather", gather)
            for name, data in items:
                info, planes, consumed, st = dec.decode(data)
                assert all(np.array_equal(a, b) for a, b in zip(planes, want[name])), (name, gather)
            job = dec.job().parse_batch([d for _, d in items]).upload().run().wait()
            for f, (name, _) in enumerate(items):
                assert all(np.array_equal(a, b) for a, b in zip(job.download_frame(f)[1], want[name])), (name, gather, "batch")
            ms_parse, ms_stage = job.host_ms()
            assert ms_parse > 0 and (ms_stage > 1e-3) == bool(gather)     # the staging copy exists only with device gather
            job.free()
    finally:
        dec.set_int("device_gather", 1)


# ---------------------------------------------------------------- damaged HT code-block bodies
def test_damaged_ht_bodies_match_the_oracle(dec, orc):
    """Bit flips, random bytes and runs of 0xFF inside the code-block bytes (headers intact): the same error code, or
    the same pixels and the same number of rejected blocks.  Two documented corners are carved out and must be DETECTED,
    not assumed -- the oracle's instrumented block decoder reports the blocks (oracle.underrun_windows):
      1. a corrupt block whose backward VLC (or MagRef) reader consumes more bits than its stream holds gets zeros on
         the device, where the reference hands out its first byte again and again (jpeg2000htdec.c:145-201 pins `pos`
         to 0);
      2. a corrupt cleanup pass of an odd-sized block with refinement passes can mark the lower right sample of a quad
         significant although it lies outside the block; the reference keeps it on one border (x3 = x1 | x2,
         jpeg2000htdec.c:985) and SigProp sees it as a neighbour, the device masks every sample outside the block.
    For such frames the dequantised coefficient planes must still agree everywhere outside the windows of exactly
    those blocks, and block-error counts may differ by at most the number of reported blocks."""
    import ffmpeg_ht_amd as m
    rng = np.random.default_rng(11)
    names = ["gray_l5_cb64", "rgb_mct", "gray_3passes", "rgb_3passes_cb32", "gray_97_q2", "placeholder_2_3p", "noise_max",
             "gray_l3_cb256x16", "gray_3passes_vsc"]
    extra = {"c16_256x192": vecgen.encode(vecgen.synth_image(256, 192, 3, seed=5, noise=10), mct=1, nlevels=4),
             "c16_512x256_cb32": vecgen.encode(vecgen.synth_image(512, 256, 3, seed=6, noise=30), mct=1, nlevels=5, cb=(5, 5))}
    same = carved = rejected = 0
    for name in names + sorted(extra):
        data, kw = (extra[name], {}) if name in extra else streams.get(name)
        start = data.index(b"\xff\x93") + 2
        for it in range(24):
            b = bytearray(data)
            mode = it % 4
            for _ in range([1, 8, 64, 400][mode]):
                pos = int(rng.integers(start, len(b) - 2))
                if mode == 0: b[pos] ^= 1 << int(rng.integers(0, 8))
                elif mode == 3: b[pos] = 0xFF
                else: b[pos] = int(rng.integers(0, 256))
            b = bytes(b)
            try:
                info_o, planes_o, _ = orc.decode(b, **kw); eo = 0
            except oracle.DecodeError as e:
                eo = e.code
            try:
                info, planes, _, st = dec.decode(b); eg = 0
            except m.Htj2kError as e:
                eg = e.code
            assert eo == eg, (name, it)
            if eo:
                rejected += 1
                continue
            if orc.underrun_blocks() == 0:
                assert st.n_block_errors == orc.block_errors(), (name, it)
                assert all(np.array_equal(x, y) for x, y in zip(planes, planes_o)), (name, it)
                same += 1
                continue
            # the carved-out corner: compare the planes the block decoder wrote, outside the reported blocks
            carved += 1
            orc.decode_blocks(b, **kw)                         # (orc.decode went on to transform the planes in place)
            wins = orc.underrun_windows()
            dec.set_int("bitexact", kw.get("bitexact", 0))
            job = dec.job().parse(b).upload().run(1).wait()
            for tc in range(job.num_tilecomps()):
                a, o = job.plane(tc).view(np.uint32), orc.plane(tc).view(np.uint32)
                h, w = a.shape
                off = orc.plane_offset(tc)
                mask = np.zeros(h * w, dtype=bool)
                for (po, bw, bh, stride) in wins:
                    if off <= po < off + h * w:
                        rel = po - off
                        for r in range(bh):
                            mask[rel + r * stride: rel + r * stride + bw] = True
                diff = (a.reshape(-1) != o.reshape(-1)) & ~mask
                assert not diff.any(), (name, it, tc, int(diff.sum()))
            assert abs(job.block_errors() - orc.block_errors()) <= len(wins), (name, it)
            job.free()
    assert same > 150 and carved < same // 4, (same, carved, rejected)


# ---------------------------------------------------------------- device-resident frames out of the pipeline
def test_pipe_device_frames_stay_valid_while_the_pipe_runs_on(dec, orc):
    """htj2k_pipe_receive_device: the planes of a frame stay untouched until all frames of depth - 1 further batches
    have been handed out -- the consumer keeps sending and receiving, copies the frames out late, and still finds the
    oracle's pixels.  Includes a batch with a bad packet (per-frame retry jobs) and mixed formats."""
    import ffmpeg_ht_amd as m
    names = ["rgb_mct", "gray_l5_cb64", "yuv420p8", "gray16", "rgb_tiles", "p1_gray_cb32", "gray_97_q2", "rgb10_mct"]
    pkts, want = [], []
    for i in range(30):
        data = streams.get(names[i % len(names)])[0]
        if i == 13:
            data = data[:40]                                  # fails to parse: its batch is retried frame by frame
            want.append(None)
        else:
            want.append(orc.decode(data)[1])
        pkts.append(data)
    batch, depth = 4, 3
    pipe = dec.pipe(batch=batch, depth=depth)
    held = []                                                 # (index, info, Frame) not copied out yet
    try:
        sent = got = 0
        while got < len(pkts):
            while sent < len(pkts) and pipe.send(pkts[sent]):
                sent += 1
            if sent == len(pkts):
                pipe.flush()
            info = m.Info()
            r = dec.L.htj2k_pipe_info(pipe.h, ctypes.byref(info))
            if r < 0:
                assert want[got] is None and r != m.EAGAIN
                dec.L.htj2k_pipe_skip(pipe.h)
                got += 1
                continue
            fr = pipe.receive_device()
            assert fr is not None
            held.append((got, info, fr))
            got += 1
            # frames older than (depth - 1) batches may be overwritten: copy those out now, as late as is allowed
            while held and (got - 1) // batch - held[0][0] // batch >= depth - 1:
                i, inf, f = held.pop(0)
                planes = dec.fetch_device_frame(inf, f)
                assert all(np.array_equal(a, b) for a, b in zip(planes, want[i])), i
        for i, inf, f in held:
            planes = dec.fetch_device_frame(inf, f)
            assert all(np.array_equal(a, b) for a, b in zip(planes, want[i])), i
    finally:
        pipe.close()


def test_unit_entry_points_reject_bad_descriptors(dec):
    """htj2k_ht_blocks / htj2k_mq_blocks take caller-built tables: a window outside the coefficient buffer, an over-sized
    block, a trailer that points outside the block's bytes must come back as EINVAL, not as a device fault"""
    import ffmpeg_ht_amd as m
    vals = (np.arange(64 * 64).reshape(64, 64) % 17 - 8).astype(np.int32)
    seg, lcup, lref, maxU = vecgen.encode_block(vals, passes=1)
    good = dict(data_off=0, plane_off=0, lcup=lcup, lref=lref, w=64, h=64, stride=64, npasses=1, zbp=9,
                M_b=10, flags=1, roi_shift=0, tcomp=0, f_step=1.0, i_step=32768)
    nsamples = 64 * 64

    def run(entry, **over):
        d = m.BlockDesc(**dict(good, **over))
        coef = np.zeros(nsamples, dtype=np.int32)
        status = (ctypes.c_int * 1)()
        buf = (ctypes.c_uint8 * (len(seg) + 64)).from_buffer_copy(bytes(seg) + bytes(64))
        return getattr(dec.L, entry)(dec.h, ctypes.byref(d), 1, buf, ctypes.c_size_t(len(seg) + 64),
                                     coef.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(nsamples), status)
    assert run("htj2k_ht_blocks") == 0
    for over in (dict(plane_off=1), dict(plane_off=nsamples), dict(w=128, stride=128), dict(w=0), dict(h=0), dict(stride=32),
                 dict(w=1024, h=8, stride=1024), dict(M_b=31), dict(roi_shift=31), dict(npasses=100), dict(data_off=1 << 20),
                 dict(flags=1 | 4)):
        assert run("htj2k_ht_blocks", **over) == -22, over
    assert run("htj2k_mq_blocks") == -22                      # not a Part-1 descriptor
    assert run("htj2k_mq_blocks", flags=1 | 4, lref=3000) == -22      # trailer outside the pool


# ---------------------------------------------------------------- the remaining full-size BASELINE configurations
def test_config1_512_gray(dec, orc):
    """BASELINE config 1: 512x512 gray 8-bit lossless 5/3, 5 levels, 64x64 (70 code-blocks)"""
    img = vecgen.synth_image(512, 512, 1, seed=1)
    data = vecgen.encode(img)
    info, planes, _, st = dec.decode(data)
    assert st.n_codeblocks == 70 and st.n_block_errors == 0
    assert np.array_equal(planes[0], img[0])
    assert oracle.framecrc(planes) == oracle.framecrc(orc.decode(data)[1])


def test_full_size_4k_422_irreversible_three_passes(dec, orc):
    """BASELINE config 3 with SigProp + MagRef passes in every block (k_ht_refine at full size)"""
    img = vecgen.synth_image(3840, 2160, 3, depth=12, seed=3, noise=30, dx=[1, 2, 2], dy=[1, 1, 1])
    data = vecgen.encode(img, depth=12, dx=[1, 2, 2], dy=[1, 1, 1], transform=0, qstep=1.0, cb=(5, 5), width=3840, height=2160, passes=3)
    info, planes, _, st = dec.decode(data)
    assert st.n_codeblocks == 16473 and st.n_block_errors == 0
    info_o, planes_o, _ = orc.decode(data)
    for a, b in zip(planes, planes_o):
        assert np.array_equal(a, b)


def test_full_size_8k_gray16_and_rgb48_against_the_oracle(dec, orc):
    """BASELINE config 4, both variants: 7680x4320 16-bit, 5/3, 6 levels -- gray16 and RGB -> rgb48 with RCT; lossless
    round trip and the oracle's framecrc"""
    img = vecgen.synth_image(7680, 4320, 1, depth=16, seed=4, noise=300)
    data = vecgen.encode(img, depth=16, nlevels=6)
    info, planes, _, st = dec.decode(data)
    assert st.n_codeblocks == 8227 and st.n_block_errors == 0
    assert np.array_equal(planes[0], img[0])
    assert oracle.framecrc(planes) == oracle.framecrc(orc.decode(data)[1])
    img = vecgen.synth_image(7680, 4320, 3, depth=16, seed=5, noise=300)
    data = vecgen.encode(img, depth=16, nlevels=6, mct=1, expn_bias=1)
    info, planes, _, st = dec.decode(data)
    assert st.n_codeblocks == 3 * 8227 and st.n_block_errors == 0 and m_pix(info) == "rgb48le"
    assert np.array_equal(planes[0].reshape(4320, 7680, 3), np.stack(img, -1))
    assert oracle.framecrc(planes) == oracle.framecrc(orc.decode(data)[1])


def m_pix(info):
    return oracle.PIX_NAMES[info.pix_fmt]


def test_two_devices_in_one_process(orc):
    """one host thread and one context per device, frames round-robin, no collective.  Runs where two
    GPUs are visible (the driver's multi-GPU node); a one-GPU box skips it."""
    import threading
    import torch
    import bench
    import ffmpeg_ht_amd as m
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible")
    names = ["rgb_mct", "gray_l5_cb64", "yuv420p8", "gray16", "rgb_tiles", "p1_gray_cb32"]
    pkts = [streams.get(names[i % len(names)])[0] for i in range(24)]
    want = [orc.decode(p)[1] for p in pkts]
    results, errors = {}, []

    def worker(rank, world):
        try:
            d = m.Decoder(device_id=rank)
            pipe = d.pipe(batch=4, depth=2)
            mine = bench.shard_frames(len(pkts), rank, world)
            sent = 0
            for k in range(len(mine)):
                while sent < len(mine) and pipe.send(pkts[mine[sent]]):
                    sent += 1
                if sent == len(mine):
                    pipe.flush()
                results[mine[k]] = pipe.receive()[1]
            pipe.close()
            d.close()
        except Exception as e:                                   # noqa: BLE001 - reported by the main thread
            errors.append((rank, repr(e)))
    threads = [threading.Thread(target=worker, args=(r, 2)) for r in range(2)]
    for t in threads: t.start()
    for t in threads: t.join()
    assert not errors, errors
    for i in range(len(pkts)):
        assert all(np.array_equal(a, b) for a, b in zip(results[i], want[i])), i


def test_frames_in_flight_is_the_default_pipe_depth(orc):
    """htj2k_opts.frames_in_flight: a pipe opened with depth 0 keeps that many batches in flight"""
    import ffmpeg_ht_amd as m
    data = streams.get("gray_l5_cb64")[0]
    want = orc.decode(data)[1]
    for fif, expect in ((1, 1), (4, 4), (0, 3)):
        d = m.Decoder(frames_in_flight=fif)
        pipe = d.pipe(batch=1, depth=0)
        try:
            accepted = 0
            while accepted < 20 and pipe.send(data):          # nothing is received: send stops when `depth` batches wait
                accepted += 1
            assert accepted == expect, (fif, accepted)
            for _ in range(accepted):
                info, planes = pipe.receive()
                assert all(np.array_equal(a, b) for a, b in zip(planes, want))
        finally:
            pipe.close()
            d.close()


def test_device_frames_outlive_the_pipe_and_the_decoder(orc):
    """reference-counted device frames (htj2k_pipe_receive_device_ref) may outlive the decoder, as AVFrames may outlive
    avcodec_free_context: htj2k_pipe_close with frames out defers freeing their jobs, and the pipe's reference keeps the
    context behind htj2k_close; the last htj2k_pipe_release_device frees everything; a token works once"""
    import ffmpeg_ht_amd as m
    names = ["rgb_mct", "gray_l5_cb64", "yuv420p8"]
    pkts = [streams.get(n)[0] for n in names] * 2
    want = [orc.decode(p)[1] for p in pkts[:len(names)]]
    d1 = m.Decoder()
    pipe = d1.pipe(batch=2, depth=2)
    held = []
    sent = 0
    while len(held) < len(pkts):
        while sent < len(pkts) and pipe.send(pkts[sent]):
            sent += 1
        if sent == len(pkts):
            pipe.flush()
        info = m.Info()
        r = d1.L.htj2k_pipe_info(pipe.h, ctypes.byref(info))
        assert r >= 0                                          # 2 * depth - 1 = 3 jobs of 2 frames: all six can be out
        fr, tok = pipe.receive_device_ref()
        held.append((len(held), info, fr, tok))
    L, ph = d1.L, pipe.h
    pipe.close()                                              # frames are out: deferred
    d1.close()                                                # the caller's reference; the pipe still holds one
    d2 = m.Decoder()                                          # (a device pointer is good in any context of the process)
    try:
        for k, (i, inf, fr, tok) in enumerate(held):
            got = d2.fetch_device_frame(inf, fr)
            assert all(np.array_equal(a, b) for a, b in zip(got, want[k % len(names)])), k
        for k, (i, inf, fr, tok) in enumerate(held):
            assert L.htj2k_pipe_release_device(ph, ctypes.c_uint64(tok)) == 0
            if k + 1 < len(held):
                assert L.htj2k_pipe_release_device(ph, ctypes.c_uint64(tok)) == -22   # a token releases its frame once
        # (the last release freed the pipe: the handle must not be used again)
        info, planes, _, st = d2.decode(pkts[0])             # the device is fine
        assert st.n_block_errors == 0
    finally:
        d2.close()


def test_pipe_device_frames_with_explicit_release(dec, orc):
    """htj2k_pipe_receive_device_ref / htj2k_pipe_release_device (what a reference-counted AV_PIX_FMT_HIP frame needs):
    frames held across many later batches keep their pixels; a consumer sitting on frames of every batch gets EAGAIN
    from send until it releases one; stale tokens are refused"""
    names = ["rgb_mct", "gray_l5_cb64", "yuv420p8", "gray16"]
    pkts = [streams.get(names[i % len(names)])[0] for i in range(40)]
    want = [orc.decode(p)[1] for p in pkts[:len(names)]]
    pipe = dec.pipe(batch=2, depth=3)
    try:
        sent = got = 0
        held = []                                             # (index, info, frame, token)
        starved = False
        while got < len(pkts):
            progressed = False
            while sent < len(pkts) and pipe.send(pkts[sent]):
                sent += 1; progressed = True
            if sent == len(pkts):
                pipe.flush()
            import ffmpeg_ht_amd as m
            info = m.Info()
            r = dec.L.htj2k_pipe_info(pipe.h, ctypes.byref(info))
            if r == m.EAGAIN:
                # nothing in flight and send refused: every slot is pinned by frames we hold
                assert len(held) >= 1 and not progressed
                starved = True
                i, inf, fr, tok = held.pop(0)
                assert all(np.array_equal(a, b) for a, b in zip(dec.fetch_device_frame(inf, fr), want[i % len(names)])), i
                pipe.release_device(tok)
                continue
            assert r >= 0
            fr, tok = pipe.receive_device_ref()
            held.append((got, info, fr, tok))
            got += 1
            if len(held) > 9:                                  # keep the nine newest, check and release the oldest
                i, inf, f0, t0 = held.pop(0)
                assert all(np.array_equal(a, b) for a, b in zip(dec.fetch_device_frame(inf, f0), want[i % len(names)])), i
                pipe.release_device(t0)
                assert dec.L.htj2k_pipe_release_device(pipe.h, ctypes.c_uint64(t0)) == -22      # released twice
        assert starved                                          # 9 held frames > 2 * 3 in flight: the pipe did run dry
        for i, inf, fr, tok in held:
            assert all(np.array_equal(a, b) for a, b in zip(dec.fetch_device_frame(inf, fr), want[i % len(names)])), i
            pipe.release_device(tok)
    finally:
        pipe.close()
