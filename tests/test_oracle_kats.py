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
