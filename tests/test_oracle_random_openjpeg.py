"""Random reversible codestreams from the test-vector factory, decoded by the oracle (+ shared host parser) and by
OpenJPEG 2.5.4 (via Pillow): every conforming decoder must return the same pixels, whatever the encoder did.
A third-party pin of the oracle across geometry (sizes, levels, block shapes, tiles, offsets, precincts, progression
orders), HT pass counts and every Part-1 mode switch -- CPU only."""
import io

import numpy as np
import pytest

import vecgen

try:
    from PIL import Image, features
    HAVE_OPJ = bool(features.check("jpg_2000"))
except Exception:  # pragma: no cover
    HAVE_OPJ = False


def _draw(rng, it):
    w, h = int(rng.integers(1, 300)), int(rng.integers(1, 220))
    nc = int(rng.choice([1, 1, 3, 4]))
    depth = int(rng.choice([8, 8, 12, 16])) if nc == 1 else 8             # Pillow keeps more than 8 bits for grey only
    nl = int(rng.integers(0, 6))
    cbw = int(rng.integers(2, 8))
    cbh = int(rng.integers(2, min(10, 12 - cbw) + 1))
    kw = dict(nlevels=nl, cb=(cbw, cbh), depth=depth)
    if rng.random() < 0.45:
        kw.update(part1=True, cblk_style=int(rng.choice([0, 0, 1, 2, 4, 8, 0x20, 5, 9, 0x2F])))
    elif rng.random() < 0.4:
        kw["passes"] = int(rng.choice([2, 3]))
        if rng.random() < 0.4:
            kw["vsc"] = True
    if nc >= 3 and rng.random() < 0.6:
        kw["mct"] = 1
    if rng.random() < 0.25:
        kw["tile"] = (int(rng.choice([32, 64, 96, 100])), int(rng.choice([32, 48, 64, 70])))
    if rng.random() < 0.2:
        kw["offset"] = (int(rng.integers(0, 9)), int(rng.integers(0, 9)))
    if rng.random() < 0.25:
        kw["prog"] = int(rng.integers(0, 5))
    if rng.random() < 0.15:
        kw["prec"] = [(int(rng.integers(5, 9)), int(rng.integers(5, 9))), (int(rng.integers(4, 8)), int(rng.integers(4, 8)))]
    if rng.random() < 0.1:
        kw.update(sop=True, eph=bool(rng.integers(0, 2)))
    img = vecgen.synth_image(w, h, nc, depth=depth, seed=it + 3, noise=int(rng.choice([0, 4, 20])))
    return (w, h, nc, depth), kw, img


@pytest.mark.skipif(not HAVE_OPJ, reason="Pillow/OpenJPEG not importable")
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_reversible_streams_agree_with_openjpeg(orc, seed):
    rng = np.random.default_rng(seed)
    compared = 0
    for it in range(80):
        key, kw, img = _draw(rng, 100 * seed + it)
        try:
            data = vecgen.encode(img, **kw)
        except RuntimeError:
            continue                                                       # e.g. not enough guard bits for this draw
        info, planes, _ = orc.decode(data)
        assert orc.block_errors() == 0, (key, kw)
        im = Image.open(io.BytesIO(data))
        im.load()
        a = np.array(im)
        got = planes[0].reshape(a.shape)
        assert np.array_equal(got.astype(np.int64), a.astype(np.int64)), (key, kw)
        compared += 1
    assert compared >= 60


def _draw_97(rng, it):
    """irreversible 9/7 draws: 8-bit (Pillow keeps more only for grey), optional ICT, HT or Part-1 blocks"""
    w, h = int(rng.integers(8, 300)), int(rng.integers(8, 220))
    nc = int(rng.choice([1, 3]))
    depth = int(rng.choice([8, 12])) if nc == 1 else 8
    cbw = int(rng.integers(2, 7))
    kw = dict(nlevels=int(rng.integers(0, 6)), cb=(cbw, int(rng.integers(2, min(10, 12 - cbw) + 1))), depth=depth,
              transform=0, qstep=float(rng.choice([1.0 / 64, 1.0 / 16, 0.25, 1.0])))
    if rng.random() < 0.4:
        kw.update(part1=True)
    if nc == 3 and rng.random() < 0.6:
        kw["mct"] = 1
    if rng.random() < 0.2:
        kw["tile"] = (int(rng.choice([64, 96, 100])), int(rng.choice([48, 64, 70])))
    if rng.random() < 0.2:
        kw["prog"] = int(rng.integers(0, 5))
    img = vecgen.synth_image(w, h, nc, depth=depth, seed=it + 11, noise=int(rng.choice([0, 4, 20])))
    return (w, h, nc, depth), kw, img


@pytest.mark.skipif(not HAVE_OPJ, reason="Pillow/OpenJPEG not importable")
@pytest.mark.parametrize("seed", [11, 12])
def test_random_irreversible_streams_agree_with_openjpeg_within_one_lsb(orc, seed):
    """9/7 + ICT: two float implementations of the same synthesis may round a sample differently, by one LSB at most
    (the tolerance north_star states for the irreversible path)"""
    rng = np.random.default_rng(seed)
    compared = worst = 0
    for it in range(60):
        key, kw, img = _draw_97(rng, 100 * seed + it)
        try:
            data = vecgen.encode(img, **kw)
        except RuntimeError:
            continue
        info, planes, _ = orc.decode(data)
        assert orc.block_errors() == 0, (key, kw)
        im = Image.open(io.BytesIO(data))
        im.load()
        a = np.array(im)
        got = planes[0].reshape(a.shape)
        d = int(np.abs(got.astype(np.int64) - a.astype(np.int64)).max())
        lsb = 1 << (16 - key[3]) if key[3] > 8 else 1                      # both write 12-bit samples << 4 into 16-bit words
        assert d <= lsb, (key, kw, d)
        worst = max(worst, d // lsb)
        compared += 1
    assert compared >= 45


@pytest.mark.skipif(not HAVE_OPJ, reason="Pillow/OpenJPEG not importable")
@pytest.mark.parametrize("seed", [21, 22])
def test_random_reversible_streams_at_reduced_resolution_agree_with_openjpeg(orc, seed):
    """`lowres` (jpeg2000dec.c:2913-2917, reduction_factor) against OpenJPEG's reduce: the lower resolution of a reversible
    stream is an exact integer reconstruction too"""
    rng = np.random.default_rng(seed)
    compared = 0
    for it in range(90):
        key, kw, img = _draw(rng, 100 * seed + it)
        kw.pop("offset", None)
        if kw["nlevels"] < 2:
            continue
        red = int(rng.integers(1, kw["nlevels"]))                          # (Pillow's OpenJPEG refuses reduce == levels)
        try:
            data = vecgen.encode(img, **kw)
        except RuntimeError:
            continue
        info, planes, _ = orc.decode(data, reduction_factor=red)
        assert orc.block_errors() == 0, (key, kw, red)
        im = Image.open(io.BytesIO(data))
        im.reduce = red
        try:
            im.load()
        except (OSError, ValueError):
            continue                                                       # Pillow's OpenJPEG gives up on some reduced sizes
        a = np.array(im)
        got = planes[0].reshape(a.shape)
        assert np.array_equal(got.astype(np.int64), a.astype(np.int64)), (key, kw, red)
        compared += 1
    assert compared >= 25, compared
