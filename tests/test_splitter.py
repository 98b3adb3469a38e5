"""htj2k_splitter_*: the product's frame splitter (ffmpeg-ht_amd/csrc/j2k_split.c: walks markers and boxes by their
length fields) in the role of the reference's jpeg2000 AVCodecParser (libavcodec/jpeg2000_parser.c).
CPU only; the .so is loaded, no device call is made.  The reference's parser has no FATE reference of its own offline,
so these are property tests -- a concatenation of frames comes apart into exactly those frames whatever the chunking,
marker look-alikes do not cut a frame, JP2 files end where the next file or codestream begins -- plus a comparison
with the oracle's splitter (oracle/j2k_oracle_split.c, the statement-for-statement restatement of the reference's
byte scanner) on the same input in the same pieces."""
import ctypes

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


def _oracle_split(stream, pieces):
    """the same loop as Splitter.split, through the oracle's restatement of find_frame_end + ff_combine_frame"""
    import oracle
    L = oracle.lib()
    h = ctypes.c_void_p()
    assert L.orc_splitter_open(ctypes.byref(h)) == 0
    out, pos = [], 0
    sizes = list(pieces)
    while pos < len(stream):
        piece = stream[pos:pos + (sizes.pop(0) if sizes else 4096)]
        while True:
            buf = (ctypes.c_uint8 * (len(piece) + 64)).from_buffer_copy(bytes(piece) + bytes(64))
            fr, n = ctypes.POINTER(ctypes.c_uint8)(), ctypes.c_int()
            used = L.orc_splitter_parse(h, buf, len(piece), ctypes.byref(fr), ctypes.byref(n))
            assert used >= 0
            if fr:
                out.append(ctypes.string_at(fr, n.value))
            pos += used
            piece = piece[used:]
            if not piece or (used == 0 and not fr):
                break
    buf = (ctypes.c_uint8 * 64)()
    fr, n = ctypes.POINTER(ctypes.c_uint8)(), ctypes.c_int()
    L.orc_splitter_parse(h, buf, 0, ctypes.byref(fr), ctypes.byref(n))
    if fr and n.value:
        out.append(ctypes.string_at(fr, n.value))
    L.orc_splitter_close(h)
    return out


def test_same_cuts_as_the_reference_scanner(m):
    """random sequences of codestreams (one or several tile-parts, Psot = 0, COM look-alikes, SOP/EPH) and JP2 files in
    random pieces: the marker / box walker and the oracle's byte scanner cut at the same places"""
    import cs_rewrite
    rng = np.random.default_rng(11)
    img = streams._img(96, 64, 3, 10, 21)
    sopeph = vecgen.encode(streams._img(190, 131, 3, 8, 6), mct=1, tile=(100, 70), nlevels=3, sop=True, eph=True)
    pool = _frames() + [
        vecgen.jp2_wrap(vecgen.encode(img, depth=10, nlevels=3), 96, 64, 3, 10, colourspace=16),
        vecgen.jp2_wrap(streams.get("tiny_3x1_l2")[0], 3, 1, 1, 8, colourspace=17),
        vecgen.encode([np.random.default_rng(3).integers(0, 256, (96, 96))], nlevels=2, comment=b"end\xff\xd9 start\xff\x4f sot\xff\x90 t"),
    ] + [d for n, d in cs_rewrite.variants(sopeph, True) if n in ("tp3_tlm_plt", "tp_each_packet", "ppm", "ppt_tp3")]
    for trial in range(16):
        seq = [pool[int(k)] for k in rng.integers(0, len(pool), int(rng.integers(2, 7)))]
        stream = b"".join(seq)
        maxpiece = int(rng.choice([5, 40, 300, 5000, 1 << 20]))
        pieces = [int(v) for v in rng.integers(1, maxpiece + 1, 4 + 2 * len(stream) // max(1, maxpiece // 2))]
        sp = m.Splitter()
        got, pos, sizes = [], 0, list(pieces)
        while pos < len(stream):
            piece = stream[pos:pos + (sizes.pop(0) if sizes else 4096)]
            while True:
                used, fr = sp.parse(piece)
                if fr is not None:
                    got.append(fr)
                pos += used
                piece = piece[used:]
                if not piece or (used == 0 and fr is None):
                    break
        used, fr = sp.parse(b"")
        if fr:
            got.append(fr)
        sp.close()
        assert got == seq, (trial, [len(g) for g in got], [len(f) for f in seq])
        assert _oracle_split(stream, pieces) == seq, trial


def test_open_ended_tile_parts(m):
    """Psot = 0 ("this tile-part runs up to EOC", T.800 A.4.2): the walker scans such a tile-part for EOC.  The
    reference's scanner computes Psot - 9 in unsigned arithmetic there (jpeg2000_parser.c:123-124) and skips 4 GB --
    it swallows every frame that follows; that defect is not reproduced."""
    z = streams.get("psot_zero")[0]
    a = _frames()[0]
    seq = [z, a, z, z, a]
    for chunk in (7, 4096, 1 << 20):
        sp = m.Splitter()
        assert sp.split(b"".join(seq), chunk=chunk) == seq
        sp.close()
    assert len(_oracle_split(b"".join(seq), [4096] * 64)) == 1
