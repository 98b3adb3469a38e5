"""htj2k_mxf_next_essence: the KLV layer of the reference's MXF demuxer for JPEG 2000 picture essence
(libavformat/mxfdec.c:432-504, 4034-4160).  CPU only; the .so is loaded, no device call is made.

No MXF file ships with the reference offline (its FATE MXF samples are rsync'd) and no MXF muxer is in the image, so
the files here are assembled by hand below -- partition packs, primer, fill items, sound elements and picture
elements as KLV triplets with BER lengths of every form: parity with the reference's demuxer is UNPINNED, these are
property tests (the packets that went in come out, in order, whatever surrounds them)."""
import struct

import numpy as np
import pytest

import streams
import vecgen

UL = bytes.fromhex
HEADER_PACK = UL("060e2b34020501010d01020101020400")        # header partition, closed & complete (mxfdec.c:338)
FOOTER_PACK = UL("060e2b34020501010d01020101040400")
PRIMER_PACK = UL("060e2b34020501010d01020101050100")
FILL_ITEM = UL("060e2b34010101020301021001000000")          # KLV fill (mxfdec.c: mxf_klv_fill)
RANDOM_INDEX = UL("060e2b34020501010d01020101110100")
SOUND_ELEM = UL("060e2b34010201010d01030116010101")         # GC sound item, BWF frame-wrapped
PICT_J2K = UL("060e2b34010201010d01030115010801")           # SMPTE 422M frame-wrapped JPEG 2000 (mxfenc.c:217)
PICT_J2K_2 = UL("060e2b34010201010d01030115020802")         # second picture track (stereoscopic)
PICT_J2K_CLIP = UL("060e2b34010201010d01030115010901")      # clip-wrapped
PICT_MPEG = UL("060e2b34010201010d01030115010501")          # GC picture item, MPEG frame-wrapped: not ours
ENCRYPTED = UL("060e2b34020401070d010301027e0100")          # encrypted triplet (mxfdec.c:349)


def ber(n, form):
    if form == "short":
        assert n < 128
        return bytes([n])
    k = {"b1": 1, "b2": 2, "b3": 3, "b4": 4, "b8": 8}[form]
    return bytes([0x80 | k]) + n.to_bytes(k, "big")


def klv(key, value, form="b4"):
    return key + ber(len(value), form) + value


@pytest.fixture(scope="module")
def m():
    import ffmpeg_ht_amd as mod
    return mod


def _frames():
    names = ["gray_l5_cb64", "rgb_tiles_offsets", "tiny_3x1_l2", "p1_gray_cb32", "noise_max"]
    return [streams.get(n)[0] for n in names]


def _mxf(frames, run_in=b"", picture_key=PICT_J2K, forms=("b4", "b3", "b8", "b4", "b4")):
    rng = np.random.default_rng(7)
    out = [run_in, klv(HEADER_PACK, bytes(88 + 16)), klv(PRIMER_PACK, struct.pack(">II", 0, 18)), klv(FILL_ITEM, bytes(300), "b2")]
    for i, f in enumerate(frames):
        out.append(klv(picture_key, f, forms[i % len(forms)]))
        out.append(klv(SOUND_ELEM, rng.integers(0, 256, 1920 * 3, dtype=np.uint8).tobytes()))
        if i & 1:
            out.append(klv(FILL_ITEM, bytes(17), "short"))
    out += [klv(FOOTER_PACK, bytes(88)), klv(RANDOM_INDEX, bytes(28))]
    return b"".join(out)


def test_frame_wrapped_elements_come_out_in_order(m):
    frames = _frames()
    data = _mxf(frames)
    got = m.mxf_essence(data)
    assert [g[0] for g in got] == frames
    assert all(g[1] == 0x15010801 and g[2] == m.MXF_FRAME_WRAPPED for g in got)
    for payload, _, _, off in got:                       # klv_offset is AVPacket.pos: the element's key
        assert data[off:off + 16] == PICT_J2K
        assert data.find(payload) > off


def test_run_in_and_garbage_are_stepped_over(m):
    """klv_read_packet() resynchronises on 06 0E 2B 34: run-in bytes and a damaged stretch between triplets do not
    lose the elements behind them"""
    frames = _frames()[:3]
    clean = _mxf(frames, run_in=bytes(range(1, 200)) + b"\x06\x0e\x2b")
    assert [g[0] for g in m.mxf_essence(clean)] == frames
    two = _mxf(frames[:2]) + b"\xde\xad\xbe\xef" * 5 + klv(PICT_J2K, frames[2])
    assert [g[0] for g in m.mxf_essence(two)] == frames


def test_other_elements_are_not_picture_essence(m):
    frames = _frames()[:2]
    data = (klv(HEADER_PACK, bytes(104)) + klv(PICT_MPEG, b"\x00\x00\x01\xb3" + bytes(100)) + klv(PICT_J2K, frames[0]) +
            klv(ENCRYPTED, bytes(200)) + klv(PICT_J2K_2, frames[1]))
    got = m.mxf_essence(data)
    assert [g[0] for g in got] == frames
    assert [g[1] for g in got] == [0x15010801, 0x15020802]         # two picture tracks: the caller selects by number


def test_clip_wrapped_element_goes_through_the_splitter(m):
    frames = _frames()
    data = _mxf([b"".join(frames)], picture_key=PICT_J2K_CLIP, forms=("b8",))
    got = m.mxf_essence(data)
    assert len(got) == 1 and got[0][2] == m.MXF_CLIP_WRAPPED
    sp = m.Splitter()
    assert sp.split(got[0][0], chunk=5000) == frames
    sp.close()


def test_truncated_and_malformed_files(m):
    import ctypes
    frames = _frames()[:2]
    data = _mxf(frames)
    cut = data[:data.find(frames[1]) + len(frames[1]) // 2]          # file ends inside the second picture element
    got = m.mxf_essence(cut)
    assert got[0][0] == frames[0] and got[1][0] == frames[1][:len(frames[1]) // 2]
    for n in range(0, 40):                                           # ends inside a key or a length field
        m.mxf_essence(data[:n])
    assert m.mxf_essence(b"") == []
    with pytest.raises(m.Htj2kError):                                # nine length bytes: SMPTE 379M 5.3.4
        m.mxf_essence(PICT_J2K + bytes([0x89]) + bytes(9) + b"x")
    with pytest.raises(m.Htj2kError):                                # length beyond INT64_MAX
        m.mxf_essence(PICT_J2K + bytes([0x88]) + b"\xff" * 8)
    L = m.load_library()
    pos, e = ctypes.c_size_t(5), m.MxfEssence()
    L.htj2k_mxf_next_essence.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(m.MxfEssence)]
    assert L.htj2k_mxf_next_essence(b"abc", 3, ctypes.byref(pos), ctypes.byref(e)) == -22      # pos beyond the buffer
    assert L.htj2k_mxf_next_essence(None, 0, ctypes.byref(pos), ctypes.byref(e)) == -22


def test_essence_decodes_like_the_bare_codestream():
    """the packets that come out of the MXF file are what the decoder's oracle reads directly"""
    import ffmpeg_ht_amd as m
    import oracle
    frames = [vecgen.encode(vecgen.synth_image(80, 48, 3, seed=s), nlevels=3, mct=1, transform=1, part1=bool(s & 1)) for s in range(3)]
    orc = oracle.OracleDecoder()
    for payload, f in zip([g[0] for g in m.mxf_essence(_mxf(frames))], frames):
        a, b = orc.decode(payload), orc.decode(f)
        assert a[2] == b[2] and all(np.array_equal(x, y) for x, y in zip(a[1], b[1]))
    orc.close()
