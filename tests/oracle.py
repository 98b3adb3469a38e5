"""ctypes binding of the CPU oracle (oracle/j2k_oracle.c).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by
the product package."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = os.path.join(ROOT, "oracle", "libj2k_oracle.so")


class Opts(ctypes.Structure):
    _fields_ = [("bitexact", ctypes.c_int), ("reduction_factor", ctypes.c_int), ("max_pixels", ctypes.c_int64),
                ("strict", ctypes.c_int), ("device_id", ctypes.c_int), ("frames_in_flight", ctypes.c_int),
                ("req_pix_fmt", ctypes.c_int)]


class Info(ctypes.Structure):
    _fields_ = [("width", ctypes.c_int), ("height", ctypes.c_int), ("pix_fmt", ctypes.c_int),
                ("bits_per_raw_sample", ctypes.c_int), ("profile", ctypes.c_int), ("lossless", ctypes.c_int),
                ("sar_num", ctypes.c_int), ("sar_den", ctypes.c_int), ("ncomponents", ctypes.c_int),
                ("is_ht", ctypes.c_int), ("nplanes", ctypes.c_int), ("plane_width", ctypes.c_int * 4),
                ("plane_height", ctypes.c_int * 4), ("plane_bytes_per_sample", ctypes.c_int * 4),
                ("has_palette", ctypes.c_int)]


class Frame(ctypes.Structure):
    _fields_ = [("data", ctypes.c_void_p * 4), ("linesize", ctypes.c_int * 4), ("width", ctypes.c_int),
                ("height", ctypes.c_int), ("pix_fmt", ctypes.c_int)]


PIX_NAMES = ["pal8", "rgb24", "rgba", "rgb48le", "rgba64le", "gray", "ya8", "gray16le", "ya16le",
             "yuv410p", "yuv411p", "yuva420p", "yuv420p", "yuv422p", "yuva422p", "yuv440p", "yuv444p", "yuva444p",
             "yuv420p9le", "yuv422p9le", "yuv444p9le", "yuva420p9le", "yuva422p9le", "yuva444p9le",
             "yuv420p10le", "yuv422p10le", "yuv444p10le", "yuva420p10le", "yuva422p10le", "yuva444p10le",
             "yuv420p12le", "yuv422p12le", "yuv444p12le", "yuv420p14le", "yuv422p14le", "yuv444p14le",
             "yuv420p16le", "yuv422p16le", "yuv444p16le", "yuva420p16le", "yuva422p16le", "yuva444p16le", "xyz12le"]

ERR = {-0x41444E49: "INVALIDDATA", -0x45574150: "PATCHWELCOME", -0x21475542: "BUG", -12: "ENOMEM", -22: "EINVAL",
       -38: "ENOSYS", -0x20545845: "EXTERNAL"}


BLOCK_DTYPE = np.dtype([("data_off", "<u4"), ("plane_off", "<u4"), ("lcup", "<u2"), ("lref", "<u2"), ("w", "<u2"), ("h", "<u2"),
                        ("stride", "<u2"), ("npasses", "u1"), ("zbp", "u1"), ("M_b", "u1"), ("flags", "u1"), ("roi_shift", "u1"),
                        ("tcomp", "u1"), ("f_step", "<f4"), ("i_step", "<i4")])
assert BLOCK_DTYPE.itemsize == 32


class Plan(ctypes.Structure):
    """struct J2kPlan of oracle/j2k_oracle_plan.h (the oracle's frozen copy of the descriptor layout)"""
    _fields_ = [("info", Info), ("bytes_consumed", ctypes.c_int32), ("precision", ctypes.c_int32), ("out_bytes", ctypes.c_int32),
                ("out_shift_precision", ctypes.c_int32), ("ntiles", ctypes.c_int32), ("ntilecomps", ctypes.c_int32),
                ("tilecomps", ctypes.c_void_p), ("nblocks", ctypes.c_int32), ("blocks", ctypes.c_void_p), ("bytes", ctypes.c_void_p),
                ("nbytes", ctypes.c_size_t), ("nsamples", ctypes.c_size_t), ("max_lcup", ctypes.c_uint32), ("max_lref", ctypes.c_uint32),
                ("max_pcup", ctypes.c_uint32), ("max_scup", ctypes.c_uint32), ("max_qw", ctypes.c_uint32), ("max_bm_words", ctypes.c_uint32),
                ("have_part1", ctypes.c_int32), ("palette", ctypes.c_uint32 * 256)]


class DecodeError(RuntimeError):
    def __init__(self, code):
        super().__init__("decode failed: %s (%d)" % (ERR.get(code, "?"), code))
        self.code = code


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            subprocess.check_call(["make", "-C", ROOT, "oracle"])
        L = ctypes.CDLL(_LIB)
        L.orc_frame_new.restype = ctypes.c_void_p
        L.orc_frame_plane.restype = ctypes.c_void_p
        L.orc_vlc_table.restype = ctypes.POINTER(ctypes.c_uint16)
        for n in ("orc_frame_free", "orc_frame_decode_blocks", "orc_frame_parse", "orc_frame_decode_parsed", "orc_frame_underrun_blocks",
                  "orc_frame_block_note", "orc_frame_idwt", "orc_frame_write", "orc_frame_info",
                  "orc_frame_bytes_consumed", "orc_frame_block_errors", "orc_frame_num_blocks",
                  "orc_frame_num_tilecomps", "orc_frame_tilecomp_dims", "orc_frame_plane", "orc_decode", "orc_probe"):
            getattr(L, n).argtypes = None
        _lib = L
    return _lib


def make_opts(bitexact=0, reduction_factor=0, req_pix_fmt=-1, strict=0, max_pixels=0):
    o = Opts()
    o.bitexact = bitexact
    o.reduction_factor = reduction_factor
    o.req_pix_fmt = req_pix_fmt
    o.strict = strict
    o.max_pixels = max_pixels
    return o


def alloc_frame(info, align=1):
    """numpy planes + an htj2k_frame pointing at them"""
    planes, fr = [], Frame()
    for p in range(info.nplanes):
        rowbytes = info.plane_width[p] * info.plane_bytes_per_sample[p]
        ls = -(-rowbytes // align) * align
        a = np.zeros((info.plane_height[p], ls), dtype=np.uint8)
        planes.append(a)
        fr.data[p] = a.ctypes.data
        fr.linesize[p] = ls
    return planes, fr


def planes_to_arrays(info, planes):
    """strip padding and view as uint8/uint16 sample arrays (h, w*samples_per_pixel)"""
    out = []
    for p in range(info.nplanes):
        rowbytes = info.plane_width[p] * info.plane_bytes_per_sample[p]
        a = np.ascontiguousarray(planes[p][:, :rowbytes])
        spp = info.plane_bytes_per_sample[p]
        bps = 2 if info.bits_per_raw_sample > 8 else 1
        a = a.view(np.uint16) if bps == 2 else a
        out.append(a.reshape(info.plane_height[p], -1))
        del spp
    return out


class OracleDecoder:
    def __init__(self):
        self.L = lib()
        self.h = ctypes.c_void_p(self.L.orc_frame_new())
        self._buf = None

    def close(self):
        if self.h:
            self.L.orc_frame_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _pkt(self, data):
        self._buf = ctypes.create_string_buffer(bytes(data) + b"\0" * 64, len(data) + 64)
        return self._buf

    def probe(self, data, **kw):
        info = Info()
        o = make_opts(**kw)
        r = self.L.orc_probe(self.h, self._pkt(data), len(data), ctypes.byref(o), ctypes.byref(info))
        if r < 0:
            raise DecodeError(r)
        return info

    def decode(self, data, **kw):
        """-> (info, [plane arrays], bytes_consumed)"""
        info = self.probe(data, **kw)
        planes, fr = alloc_frame(info)
        o = make_opts(**kw)
        r = self.L.orc_decode(self.h, self._pkt(data), len(data), ctypes.byref(o), ctypes.byref(fr))
        if r < 0:
            raise DecodeError(r)
        return info, planes_to_arrays(info, planes), r

    def decode_blocks(self, data, **kw):
        o = make_opts(**kw)
        r = self.L.orc_frame_decode_blocks(self.h, self._pkt(data), len(data), ctypes.byref(o))
        if r < 0:
            raise DecodeError(r)

    def plan_blocks(self, data, **kw):
        """the oracle parser's block table (numpy structured array, fields of struct J2kBlock)"""
        L = self.L
        L.orc_parser_new.restype = ctypes.c_void_p
        p = ctypes.c_void_p(L.orc_parser_new())
        try:
            o = make_opts(**kw)
            plan = ctypes.POINTER(Plan)()
            buf = self._pkt(data)
            r = L.orc_parse(p, buf, len(data), ctypes.byref(o), 0, ctypes.byref(plan))
            if r < 0:
                raise DecodeError(r)
            n = plan.contents.nblocks
            raw = ctypes.string_at(plan.contents.blocks, n * BLOCK_DTYPE.itemsize)
            return np.frombuffer(raw, dtype=BLOCK_DTYPE).copy()
        finally:
            L.orc_parser_free(p)

    def parse(self, data, **kw):
        """host parsing alone; decode_parsed() / idwt() / write() carry on from it"""
        o = make_opts(**kw)
        r = self.L.orc_frame_parse(self.h, self._pkt(data), len(data), ctypes.byref(o))
        if r < 0:
            raise DecodeError(r)

    def decode_parsed(self):
        r = self.L.orc_frame_decode_parsed(self.h)
        if r < 0:
            raise DecodeError(r)

    def write(self):
        """mct + write_frame of the decoded, inverse-transformed frame -> (info, [plane arrays])"""
        info = Info()
        self.L.orc_frame_info(self.h, ctypes.byref(info))
        planes, fr = alloc_frame(info)
        r = self.L.orc_frame_write(self.h, ctypes.byref(fr))
        if r < 0:
            raise DecodeError(r)
        return info, planes_to_arrays(info, planes)

    def underrun_blocks(self):
        """blocks of the last decode whose VLC / MagRef reader consumed bits beyond its stream (corrupt blocks only)"""
        return self.L.orc_frame_underrun_blocks(self.h)

    def underrun_windows(self):
        """[(plane_off, w, h, stride)] of those blocks in the frame's coefficient buffer"""
        out = []
        po, w, h, st = ctypes.c_uint32(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        for i in range(self.num_blocks()):
            if self.L.orc_frame_block_note(self.h, i, ctypes.byref(po), ctypes.byref(w), ctypes.byref(h), ctypes.byref(st)) > 0:
                out.append((po.value, w.value, h.value, st.value))
        return out

    def plane_offset(self, tc):
        """sample offset of a tile-component's plane in the coefficient buffer (plane 0 starts at 0)"""
        base = self.L.orc_frame_plane(self.h, 0)
        return (self.L.orc_frame_plane(self.h, tc) - base) // 4

    def idwt(self):
        r = self.L.orc_frame_idwt(self.h)
        if r < 0:
            raise DecodeError(r)

    def num_tilecomps(self):
        return self.L.orc_frame_num_tilecomps(self.h)

    def num_blocks(self):
        return self.L.orc_frame_num_blocks(self.h)

    def block_errors(self):
        return self.L.orc_frame_block_errors(self.h)

    def plane(self, tc):
        w, h, f = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        self.L.orc_frame_tilecomp_dims(self.h, tc, ctypes.byref(w), ctypes.byref(h), ctypes.byref(f))
        ptr = self.L.orc_frame_plane(self.h, tc)
        dt = np.float32 if f.value else np.int32
        n = w.value * h.value
        buf = (ctypes.c_char * (n * 4)).from_address(ptr)
        return np.frombuffer(buf, dtype=dt).reshape(h.value, w.value).copy()


def ht_decode_block(data, lcup, lref, npasses, zbp, w, h, M_b, roi_shift=0, vsc=False):
    """ff_jpeg2000_decode_htj2k on raw segment bytes -> (ret, int32[h, w] sign-magnitude samples)"""
    buf = ctypes.create_string_buffer(bytes(data) + b"\0" * 16, len(data) + 16)
    out = np.zeros((h, w), dtype=np.int32)
    r = lib().orc_ht_decode_block(buf, lcup, lref, npasses, zbp, w, h, M_b, roi_shift, int(vsc),
                                  out.ctypes.data_as(ctypes.c_void_p), w)
    return r, out


def mq_needs_termination(style, passno):
    """needs_termination(), libavcodec/jpeg2000.h:302-317"""
    if style & 1:
        typ, k = passno % 3, passno // 3
        if typ == 0 and k > 2:
            return 2
        if typ == 2 and k > 2:
            return 1
        if style & 4:
            return 2 if k > 2 else 1
    return 1 if style & 4 else 0


def mq_block_layout(segbytes, seglens, segpasses, style):
    """Lay the codeword segments of one Part-1 block out as the reference's Tier-2 does (jpeg2000dec.c:1236-1253,
    1508-1516): 0xFF 0xFF and a data_start entry behind every segment whose last pass needs a termination.
    -> (data bytes, length, [data_start[1..]])"""
    data, starts, pos, passno = b"", [], 0, 0
    for ln, npz in zip(seglens, segpasses):
        data += segbytes[pos:pos + ln]
        pos += ln
        passno += npz
        if mq_needs_termination(style, passno - 1):
            data += b"\xff\xff"
            starts.append(len(data))
    return data, len(data), starts


def mq_decode_block(data, length, npasses, nonzerobits, w, h, M_b, style=0, bandpos=0, starts=(), roi_shift=0):
    """decode_cblk on the laid-out bytes -> (ret, int32[h, w] sign-magnitude samples)"""
    buf = ctypes.create_string_buffer(bytes(data) + b"\0" * 16, len(data) + 16)
    out = np.zeros((h, w), dtype=np.int32)
    st = (ctypes.c_uint16 * (len(starts) + 2))(0, *starts)          # data_start[0] is never read
    r = lib().orc_mq_decode_block(buf, length, npasses, nonzerobits, w, h, M_b, roi_shift, style, bandpos, len(starts), st,
                                  out.ctypes.data_as(ctypes.c_void_p), w)
    return r, out


def mq_block_region(data, length, style, bandpos, starts):
    """the block's region of the byte pool as ffmpeg-ht_amd/csrc/j2k_plan.h describes it: bytes, 0xFF 0xFF, trailer"""
    import struct
    body = bytes(data[:length]) + b"\xff\xff"
    body += b"\0" * ((-len(body)) % 4)
    assert len(body) == (length + 2 + 3) & ~3
    body += struct.pack("<BBH", style, bandpos, len(starts)) + b"".join(struct.pack("<H", v) for v in starts)
    return body + b"\0" * ((-len(body)) % 16)


def _border(b):
    arr = (ctypes.c_int * 4)(b[0][0], b[0][1], b[1][0], b[1][1])
    return arr


def idwt(plane, border, levels, type_):
    """ff_dwt_decode on a copy of `plane` (int32 or float32, shape = border spans)"""
    a = np.ascontiguousarray(plane).copy()
    r = lib().orc_idwt_border(a.ctypes.data_as(ctypes.c_void_p), _border(border), levels, type_)
    if r < 0:
        raise DecodeError(r)
    return a


def fdwt(plane, border, levels, type_):
    a = np.ascontiguousarray(plane).copy()
    r = lib().orc_fdwt_border(a.ctypes.data_as(ctypes.c_void_p), _border(border), levels, type_)
    if r < 0:
        raise DecodeError(r)
    return a


def mct(type_, p0, p1, p2):
    a, b, c = (np.ascontiguousarray(x).copy() for x in (p0, p1, p2))
    lib().orc_mct(type_, a.ctypes.data_as(ctypes.c_void_p), b.ctypes.data_as(ctypes.c_void_p),
                  c.ctypes.data_as(ctypes.c_void_p), a.size)
    return a, b, c


def framecrc(arrays):
    """FFmpeg framecrc of a raw frame: Adler-32 seeded with 0 over the packed planes"""
    import zlib
    v = 0
    for a in arrays:
        v = zlib.adler32(np.ascontiguousarray(a).tobytes(), v)
    return v & 0xFFFFFFFF
