"""Part-1 (MQ-coded) codeblocks on the CPU: the oracle's restatement of decode_cblk() (oracle/j2k_oracle_mq.c)
and the shared host parser, pinned by a third party both ways:
 * streams written by OpenJPEG's ENCODER (tests/golden/opj_part1.npz, made by make_openjpeg_part1.py) must
   decode to the pixels stored with them (the source image for the lossless ones);
 * streams written by the test-vector factory, with every mode switch, must decode to the source image and to
   what OpenJPEG's DECODER makes of them.
No golden vector of the reference covers this path offline: "pinned by OpenJPEG", not by a
reference run."""
import io
import os

import numpy as np
import pytest

import oracle
import streams

try:
    from PIL import Image, features
    HAVE_OPJ = bool(features.check("jpg_2000"))
except Exception:  # pragma: no cover
    HAVE_OPJ = False

FIX = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "opj_part1.npz"))
FIX_NAMES = sorted(k[:-4] for k in FIX.files if k.endswith(".j2k"))

# name -> (synthetic image arguments, depth): lossless streams whose decode must equal the source
LOSSLESS = {
    "p1_gray": ((200, 150, 1, 8, 3), 8), "p1_gray_cb32": ((200, 150, 1, 8, 3), 8), "p1_gray_cb16x64": ((200, 150, 1, 8, 3), 8),
    "p1_gray_cb64x4": ((200, 150, 1, 8, 3), 8), "p1_gray_cb4x1024": ((40, 1100, 1, 8, 14), 8),
    "p1_bypass": ((200, 150, 1, 12, 3, 60), 12), "p1_reset": ((200, 150, 1, 8, 3), 8), "p1_termall": ((200, 150, 1, 8, 3), 8),
    "p1_vsc": ((200, 150, 1, 8, 3), 8), "p1_segsym": ((200, 150, 1, 8, 3), 8), "p1_bypass_termall": ((200, 150, 1, 12, 3, 60), 12),
    "p1_all_switches": ((160, 120, 1, 16, 8, 400), 16), "p1_rgb_mct": ((190, 131, 3, 8, 5), 8),
    "p1_rgb_tiles": ((190, 131, 3, 8, 6), 8),
    "p1_gray_cb128x32": ((300, 90, 1, 8, 13), 8), "p1_cb256x16_modes": ((600, 70, 1, 12, 13, 60), 12), "p1_gray_cb1024x4": ((1100, 40, 1, 8, 14), 8),
    "mixed_gray": ((200, 150, 1, 8, 3), 8), "mixed_rgb_cb32": ((190, 131, 3, 8, 5), 8), "mixed_gray16_tiles": ((160, 120, 1, 16, 8, 400), 16),
}


@pytest.mark.parametrize("name", FIX_NAMES)
def test_openjpeg_encoded_streams(orc, name):
    data = FIX[name + ".j2k"].tobytes()
    pix = FIX[name + ".pix"]
    info, planes, consumed = orc.decode(data)
    assert info.is_ht == 0 and orc.block_errors() == 0
    got = planes[0].reshape(pix.shape).astype(np.int64)
    tol = 0 if FIX[name + ".lossless"][0] else 1          # 9/7: float evaluation order differs between decoders
    assert np.abs(got - pix.astype(np.int64)).max() <= tol


@pytest.mark.parametrize("name", sorted(LOSSLESS))
def test_part1_lossless_round_trip(orc, name):
    args, depth = LOSSLESS[name]
    img = streams._img(*args)
    data, kw = streams.get(name)
    info, planes, _ = orc.decode(data, **kw)
    fmt = oracle.PIX_NAMES[info.pix_fmt]
    shift = 16 - depth if fmt in ("rgb48le", "gray16le") else 0
    got = planes[0].reshape(info.height, info.width, -1).astype(np.int64) >> shift
    assert np.array_equal(got, np.stack(img, -1))


@pytest.mark.skipif(not HAVE_OPJ, reason="Pillow/OpenJPEG not importable")
@pytest.mark.parametrize("name", ["p1_gray", "p1_gray_cb16x64", "p1_bypass", "p1_reset", "p1_termall", "p1_vsc", "p1_segsym",
                                  "p1_bypass_termall", "p1_rgb_mct", "p1_rgb_tiles", "p1_noise_max", "p1_tiny_3x1"])
def test_openjpeg_decoder_agrees(orc, name):
    data, kw = streams.get(name)
    info, planes, _ = orc.decode(data, **kw)
    im = Image.open(io.BytesIO(data))
    im.load()
    a = np.array(im)
    got = planes[0].reshape(a.shape).astype(np.int64)          # Pillow scales 12-bit samples up to 16 bits, like gray16le
    assert np.array_equal(got, a.astype(np.int64))


@pytest.mark.skipif(not HAVE_OPJ, reason="Pillow/OpenJPEG not importable")
def test_openjpeg_decoder_agrees_irreversible(orc):
    data, kw = streams.get("p1_97")
    info, planes, _ = orc.decode(data, **kw)
    a = np.array(Image.open(io.BytesIO(data))).astype(np.int64)
    assert np.abs(planes[0].reshape(a.shape).astype(np.int64) - a).max() <= 1


def test_mq_state_table_checksum(orc):
    """the 47-row probability table expanded to the reference's 2 x 47 layout (mqc.c:32-71): spot values"""
    import ctypes
    qe = (ctypes.c_uint16 * 94)()
    nm = (ctypes.c_uint8 * 94)()
    nl = (ctypes.c_uint8 * 94)()
    orc.L.orc_mq_tables(qe, nm, nl)
    assert qe[0] == 0x5601 and qe[92] == 0x5601 and qe[90] == 0x0001
    assert list(nm[:12]) == [2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 76, 77]
    assert list(nl[:12]) == [3, 2, 12, 13, 18, 19, 24, 25, 58, 59, 66, 67]
    assert (nm[92], nm[93], nl[92], nl[93]) == (92, 93, 92, 93)


def _p1_block_case(rng, w, h, amp, density, style, band):
    import vecgen
    vals = rng.integers(-amp, amp + 1, (h, w)) * (rng.random((h, w)) < density)
    vals[0, 0] = amp                                   # never all zero
    seg, lens, passes, K, npasses = vecgen.encode_block_p1(vals, band=band, style=style)
    return vals, seg, lens, passes, K, npasses


@pytest.mark.parametrize("style", [0, 0x01, 0x02, 0x04, 0x08, 0x20, 0x05, 0x2F])
def test_block_level_round_trip(style):
    """decode_cblk() on single blocks from the factory's EBCOT encoder: shapes incl. 1-wide, 1024 x 4, odd sizes"""
    rng = np.random.default_rng(100 + style)
    for (w, h) in [(64, 64), (32, 32), (63, 61), (1, 1), (3, 5), (1, 40), (40, 1), (1024, 4), (4, 1024), (128, 32), (17, 200)]:
        for amp, density in ((1, 0.05), (3, 0.5), (200, 1.0), (30000, 1.0)):
            for band in (0, 1, 3):
                vals, seg, lens, passes, K, npasses = _p1_block_case(rng, w, h, amp, density, style, band)
                M_b = K + 2
                data, length, starts = oracle.mq_block_layout(seg, lens, passes, style)
                ret, t1 = oracle.mq_decode_block(data, length, npasses, K, w, h, M_b, style, band, starts)
                assert ret == 1
                mag = (t1 & 0x7FFFFFFF) >> (31 - M_b)
                assert np.array_equal(np.where(t1 < 0, -mag, mag), vals), (w, h, amp, style, band)


def test_block_level_errors_keep_the_decoded_passes():
    """"bpno became invalid" and "Missing needed termination": decode_cblk() fails part-way and the reference goes
    on to dequantise what is there (jpeg2000dec.c:2019-2022, 2040-2043, 2275-2290)"""
    rng = np.random.default_rng(5)
    vals, seg, lens, passes, K, npasses = _p1_block_case(rng, 32, 32, 50, 1.0, 0x04, 1)
    data, length, starts = oracle.mq_block_layout(seg, lens, passes, 0x04)
    ret_ok, good = oracle.mq_decode_block(data, length, npasses, K, 32, 32, K + 2, 0x04, 1, starts)
    assert ret_ok == 1
    # more passes signalled than bit-planes exist: the bit-plane counter runs below zero
    ret, part = oracle.mq_decode_block(data, length, npasses + 9, K, 32, 32, 28, 0x04, 1, starts + [length] * 9)
    assert ret < 0 and part.any()
    # a TERMALL block that lost its last three segment starts
    ret, part = oracle.mq_decode_block(data, length, npasses, K, 32, 32, K + 2, 0x04, 1, starts[:-3])
    assert ret < 0 and part.any() and not np.array_equal(part, good)
