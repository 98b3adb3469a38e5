"""htj2k_splitter_*: the host-side restatement of the reference's jpeg2000 AVCodecParser
(libavcodec/jpeg2000_parser.c).  CPU only; the .so is loaded, no device call is made.
The reference's parser has no FATE reference of its own offline, so these are property tests: a concatenation
of frames comes apart into exactly those frames whatever the chunking, marker look-alikes in packet bodies do
not cut a frame, JP2 files end where the next file or codestream begins."""
import numpy as np
import pytest

import streams
import vecgen


@pytest.fixture(scope="module")
def m():
    import ffmpeg_ht_amd as mod
    return mod


def _frames():
    names = ["gray_l5_cb64", "rgb_tiles_offsets", "tiny_3x1_l2", "p1_gray_cb32", "gray_sop_eph", "p1_bypass_termall", "noise_max"]
    return [streams.get(n)[0] for n in names]


@pytest.mark.parametrize("chunk", [1, 7, 64, 1000, 4096, 1 << 20])
def test_codestreams_come_apart_at_eoc(m, chunk):
    frames = _frames()
    sp = m.Splitter()
    got = sp.split(b"".join(frames), chunk=chunk)
    assert [len(g) for g in got] == [len(f) for f in frames]
    assert got == frames
    sp.close()


def test_find_end_offsets(m):
    """find_frame_end(): offset of the first byte of the next frame, relative to the buffer of the call"""
    a, b = _frames()[0], _frames()[2]
    sp = m.Splitter()
    assert sp.find_end(a + b) == len(a)                 # EOC inside this buffer
    assert sp.find_end(b[:10]) == sp.END_NOT_FOUND      # state was reset at the boundary; new frame under way
    assert sp.find_end(b[10:]) == len(b) - 10
    sp.close()
    sp = m.Splitter()
    assert sp.find_end(a[:-1]) == sp.END_NOT_FOUND
    assert sp.find_end(a[-1:] + b) == 1
    sp.close()


def test_marker_lookalikes_in_marker_segments_do_not_cut(m):
    """packet bodies cannot hold FFD9 / FF4F (bit stuffing), marker segments can: a COM payload with both must be
    skipped by its length field"""
    frames = []
    for seed in range(6):
        f = vecgen.encode([np.random.default_rng(seed).integers(0, 256, (96, 96))], nlevels=2, part1=bool(seed & 1),
                          comment=b"end\xff\xd9 start\xff\x4f sot\xff\x90 tail")
        assert f.count(b"\xff\xd9") >= 2 and f.count(b"\xff\x4f") >= 2
        frames.append(f)
    sp = m.Splitter()
    for chunk in (13, 997, 1 << 16):
        assert sp.split(b"".join(frames), chunk=chunk) == frames
    sp.close()


def test_jp2_files_and_mixed_sequences(m):
    img = streams._img(96, 64, 3, 10, 21)
    cs = vecgen.encode(img, depth=10, nlevels=3)
    jp2 = vecgen.jp2_wrap(cs, 96, 64, 3, 10, colourspace=16)
    raw = streams.get("tiny_3x1_l2")[0]
    for seq in ([jp2, jp2, jp2], [jp2, raw], [raw, jp2, raw, jp2]):
        for chunk in (5, 333, 1 << 20):
            sp = m.Splitter()
            got = sp.split(b"".join(seq), chunk=chunk)
            sp.close()
            # a JP2 file ends where the next signature box / SOC begins; the raw codestream at its EOC
            assert b"".join(got) == b"".join(seq)
            assert len(got) == len(seq), (len(got), chunk)
            assert got == seq


def test_bad_arguments(m):
    sp = m.Splitter()
    L = m.load_library()
    assert L.htj2k_splitter_find_end(None, None, 0) == -22
    assert L.htj2k_splitter_find_end(sp.h, None, 5) == -22
    assert sp.find_end(b"") == 0
    used, fr = sp.parse(b"")
    assert used == 0 and not fr
    sp.close()
