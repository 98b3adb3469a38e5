"""Oracle + shared host parser on build-generated HTJ2K streams (CPU only):
 * lossless round trip: decode(encode(image)) == image
 * third opinion: OpenJPEG (via Pillow, when importable) decodes the same pixels
 * header facts: pix_fmt selection, dimensions, error paths of the parser."""
import io

import numpy as np
import pytest

import oracle
import streams
import vecgen

try:
    from PIL import Image, features
    HAVE_OPJ = bool(features.check("jpg_2000"))
except Exception:  # pragma: no cover
    HAVE_OPJ = False


@pytest.mark.parametrize("name", sorted(streams.CASES))
def test_oracle_decodes(orc, name):
    data, kw = streams.get(name)
    info, planes, consumed = orc.decode(data, **kw)
    assert orc.block_errors() == 0
    assert consumed > 0 and info.is_ht == (0 if name.startswith("p1_") else 1)     # p1_*: Part-1 streams, no CAP marker (MIXED ones have it)
    assert len(planes) == info.nplanes


ROUNDTRIP = {
    "gray_l5_cb64": ((200, 150, 1, 8, 3), 8), "gray_l5_cb32": ((200, 150, 1, 8, 3), 8),
    "gray_offset": ((201, 149, 1, 8, 4), 8), "rgb_mct": ((190, 131, 3, 8, 5), 8),
    "rgb_tiles_offsets": ((190, 131, 3, 8, 6), 8), "rgb_cprl_prec": ((190, 131, 3, 8, 5), 8),
    "gray16": ((160, 120, 1, 16, 8, 400), 16), "gray12": ((160, 120, 1, 12, 7, 40), 12),
    "rgb10_mct": ((160, 120, 3, 10, 8, 20), 10), "placeholder_1": ((200, 150, 1, 8, 3), 8),
    "gray_deep_levels": ((37, 23, 1, 8, 15), 8), "gray_l2_cb4x1024": ((40, 1100, 1, 8, 14), 8),
}


@pytest.mark.parametrize("name", sorted(ROUNDTRIP))
def test_lossless_round_trip(orc, name):
    args, depth = ROUNDTRIP[name]
    img = streams._img(*args)
    data, kw = streams.get(name)
    info, planes, _ = orc.decode(data, **kw)
    fmt = oracle.PIX_NAMES[info.pix_fmt]
    shift = 16 - depth if fmt in ("rgb48le", "gray16le") else 0     # write_frame's << (precision - cbps)
    got = planes[0].reshape(info.height, info.width, -1).astype(np.int64) >> shift
    want = np.stack(img, -1)
    assert np.array_equal(got, want)


@pytest.mark.skipif(not HAVE_OPJ, reason="Pillow/OpenJPEG not importable")
@pytest.mark.parametrize("name", ["gray_l5_cb64", "gray_l3_cb16x64", "rgb_mct", "rgb_rpcl_prec", "rgb_tiles",
                                  "gray_sop_eph", "noise_max", "tiny_3x1_l2", "gray_3passes", "gray_2passes",
                                  "gray_3passes_vsc", "gray_l3_cb256x16"])
def test_openjpeg_agrees_bit_exact(orc, name):
    """reversible streams: every conforming decoder must produce the same pixels"""
    data, kw = streams.get(name)
    info, planes, _ = orc.decode(data, **kw)
    im = Image.open(io.BytesIO(data))
    im.load()
    a = np.array(im)
    got = planes[0].reshape(a.shape)
    assert np.array_equal(got, a)


@pytest.mark.skipif(not HAVE_OPJ, reason="Pillow/OpenJPEG not importable")
@pytest.mark.parametrize("name", ["gray_97_q2", "rgb_97_ict", "gray_97_3passes"])
def test_openjpeg_agrees_irreversible(orc, name):
    """9/7: implementations differ in float evaluation order; at most 1 LSB apart"""
    data, kw = streams.get(name)
    info, planes, _ = orc.decode(data, **kw)
    im = Image.open(io.BytesIO(data))
    im.load()
    a = np.array(im).astype(np.int64)
    assert np.abs(planes[0].reshape(a.shape).astype(np.int64) - a).max() <= 1


def test_pix_fmt_selection(orc):
    expect = {"gray_l5_cb64": "gray", "rgb_mct": "rgb24", "gray16": "gray16le", "gray12": "gray16le",
              "rgb10_mct": "rgb48le", "yuv420p8": "yuv420p", "yuv422p12_97": "yuv422p12le", "rgba8": "rgba"}
    for name, fmt in expect.items():
        data, kw = streams.get(name)
        info = orc.probe(data, **kw)
        assert oracle.PIX_NAMES[info.pix_fmt] == fmt, name


def test_jp2_wrapper_colourspace_and_sar(orc):
    img = streams._img(96, 64, 3, 10, 21)
    cs = vecgen.encode(img, depth=10, nlevels=3)
    jp2 = vecgen.jp2_wrap(cs, 96, 64, 3, 10, colourspace=18, res=(300, 1, 150, 1, 0, 0))
    info, planes, _ = orc.decode(jp2)
    assert oracle.PIX_NAMES[info.pix_fmt] == "yuv444p10le"      # colr enumerated colourspace 18 (jpeg2000dec.c:344-347)
    assert (info.sar_num, info.sar_den) == (2, 1)
    for c in range(3):
        assert np.array_equal(planes[c], img[c])
    info2, planes2, _ = orc.decode(vecgen.jp2_wrap(cs, 96, 64, 3, 10, colourspace=16))
    assert oracle.PIX_NAMES[info2.pix_fmt] == "rgb48le"


def test_palettised_jp2(orc):
    """pclr / cmap boxes select pal8: plane 0 holds the indices, plane 1 the 256 x 0xAARRGGBB palette"""
    data, kw = streams.get("pal8_jp2")
    info, planes, _ = orc.decode(data)
    assert oracle.PIX_NAMES[info.pix_fmt] == "pal8" and info.has_palette == 1 and info.nplanes == 2
    assert np.array_equal(planes[0], (np.add.outer(np.arange(40), np.arange(56)) * 3 % 200))
    pal = planes[1].reshape(-1).view("<u4")
    assert pal.shape == (256,)
    for i in (0, 1, 57, 199):
        assert pal[i] == (0xFF << 24 | ((i * 7) & 255) << 16 | ((255 - i) & 255) << 8 | ((i * 3 + 1) & 255))
    assert not pal[200:].any()


def test_lowres_dimensions(orc):
    data, _ = streams.get("gray_l5_cb64")
    info = orc.probe(data, reduction_factor=2)
    assert (info.width, info.height) == (50, 38)
    with pytest.raises(oracle.DecodeError) as e:
        orc.probe(data, reduction_factor=6)          # only 6 resolution levels (jpeg2000dec.c:509-517)
    assert e.value.code == -22


def test_parser_error_paths(orc):
    data, _ = streams.get("gray_l5_cb64")
    with pytest.raises(oracle.DecodeError) as e:
        orc.decode(b"\x00")
    assert e.value.code == -0x41444E49
    with pytest.raises(oracle.DecodeError):
        orc.decode(b"\xff\x4f\xff\x52\x00\x03\x00")      # COD before SIZ, truncated
    # truncated body: the packet reader runs out of bytes
    with pytest.raises(oracle.DecodeError) as e:
        orc.decode(data[:len(data) // 2])
    assert e.value.code == -0x41444E49
    # MULTIHT bit in Ccap15 is rejected (jpeg2000dec.c:462-465)
    bad = vecgen.encode(streams._img(64, 64, 1, 8, 3), nlevels=1, cap_extra_bits=0x2000)
    with pytest.raises(oracle.DecodeError) as e:
        orc.decode(bad)
    assert e.value.code == -0x45574150
    # 9/7 without the HTIRV capability bit (jpeg2000dec.c:1056-1059): clear bit 5 of Ccap15
    irv = bytearray(vecgen.encode(streams._img(64, 64, 1, 8, 3), nlevels=1, transform=0, qstep=1))
    i = irv.index(b"\xff\x50")
    irv[i + 9] &= ~0x20
    with pytest.raises(oracle.DecodeError) as e:
        orc.decode(bytes(irv))
    assert e.value.code == -0x41444E49


def test_corrupt_block_is_zeroed_not_fatal(orc):
    """per-block HT errors leave the block zero and the frame is still returned
    (jpeg2000dec.c:2275-2278, jpeg2000htdec.c:1305-1306)"""
    data = bytearray(vecgen.encode(streams._img(64, 64, 1, 8, 3), nlevels=0))
    # the single block's Dcup ends right before EOC: make Scup exceed Lcup
    data[-3] = 0xFF
    info, planes, _ = orc.decode(bytes(data))
    assert orc.block_errors() == 1
    assert np.all(planes[0] == 128)
