"""The oracle (and the shared host parser) against the reference's known answers:
KAT-1..6 of SURVEY.md 8(c) -- hand-assembled HTJ2K codestreams whose framecrc and pixel
values were produced by the reference decoder itself."""
import json
import os

import numpy as np
import pytest

import oracle

HERE = os.path.dirname(os.path.abspath(__file__))
KATS = json.load(open(os.path.join(HERE, "golden", "kats.json")))


@pytest.mark.parametrize("kat", KATS, ids=[k["name"] for k in KATS])
def test_kat_framecrc_and_pixels(orc, kat):
    data = bytes.fromhex(kat["hex"])
    info, planes, consumed = orc.decode(data)
    assert oracle.PIX_NAMES[info.pix_fmt] == kat["pix_fmt"]
    assert (info.width, info.height) == (kat["width"], kat["height"])
    assert orc.block_errors() == 0
    assert oracle.framecrc(planes) == int(kat["framecrc"], 16)
    nc = 3 if kat["pix_fmt"] == "rgb24" else 1
    img = planes[0].reshape(kat["height"], kat["width"], nc)
    if "pixels" in kat:
        expect = np.full_like(img, kat["others"])
        for key, v in kat["pixels"].items():
            r, c = map(int, key.split(","))
            expect[r, c] = v
        assert np.array_equal(img, expect)
    if "top_left_4x4" in kat:
        assert img[:4, :4, 0].tolist() == kat["top_left_4x4"]
    if "rows0_3_cols0_2" in kat:
        assert img[:4, :3, 0].tolist() == kat["rows0_3_cols0_2"]
    if "n_not_128" in kat:
        assert int((img != 128).sum()) == kat["n_not_128"]


def test_kat_info_fields(orc):
    info = orc.probe(bytes.fromhex(KATS[0]["hex"]))
    assert info.is_ht == 1 and info.lossless == 1 and info.bits_per_raw_sample == 8 and info.ncomponents == 1
    info5 = orc.probe(bytes.fromhex(KATS[4]["hex"]))
    assert info5.lossless == 0


def test_cxtvlc_rows_match_the_reference_luts():
    """csrc/ht_cxtvlc_rows.h (what the kernels, the oracle and the vector factory expand their CxtVLC tables from) against
    its source: all 2 x 1024 entries of dec_cxt_vlc_table0/1 (libavcodec/jpeg2000htdec.c:1342-1502).  Runs where the
    reference tree is present (the build container); the GPU box only has the committed header."""
    import subprocess
    import sys
    if not os.path.exists("/root/reference/libavcodec/jpeg2000htdec.c"):
        pytest.skip("reference tree not present")
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "tools", "derive_cxtvlc.py"), "--check"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count(" 0 differ") == 2


def _expected_steps(bits, nlevels_coded, nres_decoded, guard, entries, irreversible):
    """band step sizes by an independent script, in the arithmetic init_band_stepsize() uses (jpeg2000.c:214-272;
    SURVEY Appendix E.9): float products rounded where the reference assigns to its float, double where it multiplies by
    a double.  entries = [(expn, mant)] in QCD order.  -> {(resolution, orientation): (f_step, i_step, M_b)}"""
    f32 = np.float32
    X, K = f32(0.812893066115961), f32(1.230174104914001)
    out = {}
    g = 0
    for r in range(nlevels_coded + 1):
        for orient in ([0] if r == 0 else [1, 2, 3]):
            e, m = entries[g]
            g += 1
            step = f32(2.0) ** f32(bits - e)                                   # exact: a power of two
            step = f32(np.float64(step) * (m / 2048.0 + 1.0))
            if irreversible:
                lowpass = 0
                if orient in (1, 2):
                    step = f32(step * f32(X * f32(2)))
                    lowpass = 1
                elif orient == 3:
                    step = f32(step * f32(f32(X * X) * f32(4)))
                step = f32(np.float64(step) * np.float64(K) ** (2 * (nres_decoded - r) + lowpass - 2))
            out[(r, orient)] = (float(step), int(np.floor(f32(step * f32(32768.0)))), e + guard - 1)
    return out


@pytest.mark.parametrize("nlevels,qstep,depth", [(2, 1.0, 8), (3, 0.5, 8), (5, 1.0 / 16, 12), (5, 2.0, 10), (4, 1.0 / 3, 16)])
def test_irreversible_step_sizes_beyond_one_level(orc, nlevels, qstep, depth):
    """A8 (init_band_stepsize) at NL > 1: KAT-5 pins the float / double operation order at one level only.  Here the
    (exponent, mantissa) pairs are read back from the stream's QCD segment by this test and every band's f_stepsize /
    i_stepsize recomputed by an independent numpy script in the documented order; the oracle's parser (and with it the
    product's, tests/test_plan_equality.py) must carry exactly those values in its block table.  Also at lowres 1 and 2
    (the K exponent counts the levels that are decoded, not the ones that are coded)."""
    import struct
    import vecgen
    img = vecgen.synth_image(96, 80, 1, depth=depth, seed=nlevels, noise=20)
    data = vecgen.encode(img, depth=depth, nlevels=nlevels, transform=0, qstep=qstep)
    q = data.index(b"\xff\x5c")
    lq, sq = struct.unpack_from(">HB", data, q + 2)
    assert sq & 31 == 2                                                       # scalar expounded
    guard = sq >> 5
    entries = [(v >> 11, v & 0x7FF) for v in struct.unpack_from(">%dH" % ((lq - 3) // 2), data, q + 5)]
    assert len(entries) == 3 * nlevels + 1
    for lowres in (0, 1, 2):
        if lowres >= nlevels:
            continue
        want = _expected_steps(depth, nlevels, nlevels + 1 - lowres, guard, entries, True)
        want = {k: v for k, v in want.items() if k[0] <= nlevels - lowres}
        blocks = orc.plan_blocks(data, reduction_factor=lowres)
        got = sorted({(float(np.float32(b["f_step"])), int(b["i_step"]), int(b["M_b"])) for b in blocks})
        assert got == sorted(set(want.values())), (lowres, got, sorted(set(want.values())))
