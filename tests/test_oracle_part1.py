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
