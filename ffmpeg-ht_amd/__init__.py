"""ffmpeg-ht_amd -- thin Python (ctypes) binding of the MI355X-native HTJ2K decode library.

The product is the C-ABI shared library `libhtj2k_amd.so` (include/htj2k_amd.h): host C
parser + HIP kernels behind the plugin surface of FFmpeg's `ff_jpeg2000_decoder`
(libavcodec/jpeg2000dec.c:2926-2939).  This module only loads it for tests, bench.py and
scripting; it contains no decoding logic and NO fallback: if the library or a GPU is
missing, construction fails loudly.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HTJ2K_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libhtj2k_amd.so")   # HTJ2K_LIB: A/B builds side by side (tools/)

ERR_NAMES = {-0x41444E49: "INVALIDDATA", -0x45574150: "PATCHWELCOME", -0x21475542: "BUG", -0x20545845: "EXTERNAL",
             -12: "ENOMEM", -22: "EINVAL", -38: "ENOSYS"}
DWT97, DWT53, DWT97_INT = 0, 1, 2

PIX_NAMES = ["pal8", "rgb24", "rgba", "rgb48le", "rgba64le", "gray", "ya8", "gray16le", "ya16le",
             "yuv410p", "yuv411p", "yuva420p", "yuv420p", "yuv422p", "yuva422p", "yuv440p", "yuv444p", "yuva444p",
             "yuv420p9le", "yuv422p9le", "yuv444p9le", "yuva420p9le", "yuva422p9le", "yuva444p9le",
             "yuv420p10le", "yuv422p10le", "yuv444p10le", "yuva420p10le", "yuva422p10le", "yuva444p10le",
             "yuv420p12le", "yuv422p12le", "yuv444p12le", "yuv420p14le", "yuv422p14le", "yuv444p14le",
             "yuv420p16le", "yuv422p16le", "yuv444p16le", "yuva420p16le", "yuva422p16le", "yuva444p16le", "xyz12le"]


class Htj2kError(RuntimeError):
    def __init__(self, code, what=""):
        super().__init__("%s failed: %s (%d)" % (what or "htj2k", ERR_NAMES.get(code, "?"), code))
        self.code = code


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


class Stats(ctypes.Structure):
    _fields_ = [("n_codeblocks", ctypes.c_int), ("n_block_errors", ctypes.c_int), ("ms_parse", ctypes.c_float),
                ("ms_h2d", ctypes.c_float), ("ms_kernels", ctypes.c_float), ("ms_d2h", ctypes.c_float),
                ("ms_ht", ctypes.c_float), ("ms_idwt", ctypes.c_float), ("ms_pack", ctypes.c_float)]


class BlockDesc(ctypes.Structure):
    """struct J2kBlock (csrc/j2k_plan.h): one codeblock descriptor, 32 bytes"""
    _fields_ = [("data_off", ctypes.c_uint32), ("plane_off", ctypes.c_uint32), ("lcup", ctypes.c_uint16),
                ("lref", ctypes.c_uint16), ("w", ctypes.c_uint16), ("h", ctypes.c_uint16), ("stride", ctypes.c_uint16),
                ("npasses", ctypes.c_uint8), ("zbp", ctypes.c_uint8), ("M_b", ctypes.c_uint8), ("flags", ctypes.c_uint8),
                ("roi_shift", ctypes.c_uint8), ("tcomp", ctypes.c_uint8), ("f_step", ctypes.c_float),
                ("i_step", ctypes.c_int32)]


assert ctypes.sizeof(BlockDesc) == 32

EXPORTS = ["htj2k_open", "htj2k_close", "htj2k_set_log", "htj2k_probe", "htj2k_decode", "htj2k_job_parse",
           "htj2k_job_upload", "htj2k_job_run", "htj2k_job_download", "htj2k_job_wait", "htj2k_job_info",
           "htj2k_job_bytes_consumed", "htj2k_job_free", "htj2k_job_num_tilecomps", "htj2k_job_tilecomp_dims",
           "htj2k_job_read_plane", "htj2k_job_run_stages", "htj2k_job_stage_ms", "htj2k_idwt_plane",
           "htj2k_idwt_bench", "htj2k_copy_bench", "htj2k_mct_planes", "htj2k_ht_blocks", "htj2k_mq_blocks", "htj2k_job_block_errors",
           "htj2k_job_num_blocks", "htj2k_job_device_plane", "htj2k_set_int", "htj2k_version", "htj2k_device_name",
           "htj2k_job_parse_batch", "htj2k_job_parse_batch_ex", "htj2k_job_num_frames", "htj2k_job_host_ms", "htj2k_job_frame_info", "htj2k_job_download_frame",
           "htj2k_job_idwt_launches", "htj2k_job_idwt_hbm_bytes", "htj2k_job_coef16", "htj2k_job_ll16", "htj2k_job_idwt_packed", "htj2k_pk16_lift_bound", "htj2k_pk16_bounds", "htj2k_job_ht_blocks_per_wave",
           "htj2k_pipe_open", "htj2k_pipe_send", "htj2k_pipe_send_ref", "htj2k_pipe_flush", "htj2k_pipe_info", "htj2k_pipe_receive",
           "htj2k_pipe_skip", "htj2k_pipe_close", "htj2k_host_alloc", "htj2k_host_free",
           "htj2k_pipe_receive_device", "htj2k_pipe_receive_device_ref", "htj2k_pipe_release_device", "htj2k_job_device_frame", "htj2k_device_to_host",
           "htj2k_splitter_open", "htj2k_splitter_find_end", "htj2k_splitter_parse", "htj2k_splitter_close",
           "htj2k_mxf_next_essence"]

_lib = None


def load_library():
    """Load libhtj2k_amd.so.  Raises if it has not been built: there is no other implementation."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libhtj2k_amd.so is not built (run `make lib` / __graft_entry__.build()); "
                              "this package has no CPU fallback")
        L = ctypes.CDLL(LIB_PATH)
        L.htj2k_version.restype = ctypes.c_char_p
        L.htj2k_device_name.restype = ctypes.c_char_p
        L.htj2k_device_name.argtypes = [ctypes.c_void_p]
        L.htj2k_job_device_plane.restype = ctypes.c_void_p
        L.htj2k_host_alloc.restype = ctypes.c_void_p
        L.htj2k_host_alloc.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
        L.htj2k_host_free.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        _lib = L
    return _lib


def _check(r, what):
    if r < 0:
        raise Htj2kError(r, what)
    return r


def _pkt(data):
    # AVPacket data carries AV_INPUT_BUFFER_PADDING_SIZE (64) zero bytes of padding (libavcodec/defs.h:40)
    return ctypes.create_string_buffer(bytes(data) + b"\0" * 64, len(data) + 64)


def packet(data):
    """(padded ctypes buffer, size): a packet that can be sent many times without re-copying it in Python"""
    return _pkt(data), len(data)


class Job:
    """One frame in the staged pipeline (htj2k_job_*)."""

    def __init__(self, dec):
        self.dec = dec
        self.h = ctypes.c_void_p(None)
        self._buf = None

    def parse(self, data):
        self._buf = _pkt(data)
        _check(self.dec.L.htj2k_job_parse(self.dec.h, self._buf, len(data), ctypes.byref(self.h)), "htj2k_job_parse")
        return self

    def parse_batch(self, datas):
        """several independent frames -> one job whose stages are single launches over the whole batch"""
        n = len(datas)
        pk = [d if isinstance(d, tuple) else packet(d) for d in datas]       # packet(): padded ctypes buffer, reusable
        self._bufs = [b for b, _ in pk]
        ptrs = (ctypes.c_void_p * n)(*[ctypes.cast(b, ctypes.c_void_p) for b in self._bufs])
        sizes = (ctypes.c_int * n)(*[sz for _, sz in pk])
        _check(self.dec.L.htj2k_job_parse_batch(self.dec.h, ptrs, sizes, n, ctypes.byref(self.h)), "htj2k_job_parse_batch")
        return self

    def num_frames(self):
        return _check(self.dec.L.htj2k_job_num_frames(self.h), "htj2k_job_num_frames")

    def host_ms(self):
        """(parse ms, staging-copy ms) per frame of the last parse / parse_batch, summed over the threads that worked"""
        a, b = ctypes.c_float(), ctypes.c_float()
        _check(self.dec.L.htj2k_job_host_ms(self.h, ctypes.byref(a), ctypes.byref(b)), "htj2k_job_host_ms")
        return a.value, b.value

    def frame_info(self, f):
        info = Info()
        _check(self.dec.L.htj2k_job_frame_info(self.h, f, ctypes.byref(info)), "htj2k_job_frame_info")
        return info

    def download_frame(self, f):
        info = self.frame_info(f)
        planes, fr = alloc_frame(info)
        _check(self.dec.L.htj2k_job_download_frame(self.dec.h, self.h, f, ctypes.byref(fr)), "htj2k_job_download_frame")
        return info, planes_to_arrays(info, planes)

    def idwt_launches(self, cap=256):
        """[(ms, algorithmic_bytes)] of the IDWT launches of the last run"""
        ms = (ctypes.c_float * cap)()
        by = (ctypes.c_double * cap)()
        n = _check(self.dec.L.htj2k_job_idwt_launches(self.dec.h, self.h, ms, by, cap), "htj2k_job_idwt_launches")
        return [(ms[i], by[i]) for i in range(min(n, cap))]

    def idwt_hbm_bytes(self, cap=256):
        """least HBM bytes of the same launches (differs from the algorithmic figure for a fused final level)"""
        by = (ctypes.c_double * cap)()
        n = _check(self.dec.L.htj2k_job_idwt_hbm_bytes(self.dec.h, self.h, by, cap), "htj2k_job_idwt_hbm_bytes")
        return [by[i] for i in range(min(n, cap))]

    def upload(self):
        _check(self.dec.L.htj2k_job_upload(self.dec.h, self.h), "htj2k_job_upload")
        return self

    def run(self, stages=7):
        _check(self.dec.L.htj2k_job_run_stages(self.dec.h, self.h, stages), "htj2k_job_run_stages")
        return self

    def wait(self):
        _check(self.dec.L.htj2k_job_wait(self.dec.h, self.h), "htj2k_job_wait")
        return self

    def info(self):
        info = Info()
        _check(self.dec.L.htj2k_job_info(self.h, ctypes.byref(info)), "htj2k_job_info")
        return info

    def stage_ms(self):
        a, b, c = ctypes.c_float(), ctypes.c_float(), ctypes.c_float()
        _check(self.dec.L.htj2k_job_stage_ms(self.dec.h, self.h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)),
               "htj2k_job_stage_ms")
        return a.value, b.value, c.value

    def coef16(self):
        """did the last run keep the sub-bands as 16-bit samples between block decoder and IDWT?"""
        return bool(_check(self.dec.L.htj2k_job_coef16(self.h), "htj2k_job_coef16"))

    def ht_blocks_per_wave(self):
        """codeblocks per wavefront in the MagSgn kernel of the last run: 1, 2 or 4 (htj2k_job_ht_blocks_per_wave)"""
        return _check(self.dec.L.htj2k_job_ht_blocks_per_wave(self.h), "htj2k_job_ht_blocks_per_wave")

    def ll16(self):
        """0 / 1 / 2: LL bands between the IDWT levels 32-bit / 16-bit / 16-bit, overflowed and run again (htj2k_job_ll16)"""
        return _check(self.dec.L.htj2k_job_ll16(self.h), "htj2k_job_ll16")

    def idwt_packed(self):
        """0, or the bits the LL bands had to fit for the last run's packed 16-bit final level (htj2k_job_idwt_packed)"""
        return _check(self.dec.L.htj2k_job_idwt_packed(self.h), "htj2k_job_idwt_packed")

    def num_tilecomps(self):
        return _check(self.dec.L.htj2k_job_num_tilecomps(self.h), "htj2k_job_num_tilecomps")

    def num_blocks(self):
        return _check(self.dec.L.htj2k_job_num_blocks(self.h), "htj2k_job_num_blocks")

    def block_errors(self):
        return _check(self.dec.L.htj2k_job_block_errors(self.dec.h, self.h), "htj2k_job_block_errors")

    def plane(self, tc):
        w, h, f = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _check(self.dec.L.htj2k_job_tilecomp_dims(self.h, tc, ctypes.byref(w), ctypes.byref(h), ctypes.byref(f)),
               "htj2k_job_tilecomp_dims")
        a = np.empty((h.value, w.value), dtype=np.float32 if f.value else np.int32)
        _check(self.dec.L.htj2k_job_read_plane(self.dec.h, self.h, tc, a.ctypes.data_as(ctypes.c_void_p),
                                               ctypes.c_size_t(a.nbytes)), "htj2k_job_read_plane")
        return a

    def download(self):
        info = self.info()
        planes, fr = alloc_frame(info)
        _check(self.dec.L.htj2k_job_download(self.dec.h, self.h, ctypes.byref(fr)), "htj2k_job_download")
        return info, planes_to_arrays(info, planes)

    def free(self):
        if self.h:
            self.dec.L.htj2k_job_free(self.dec.h, self.h)
            self.h = ctypes.c_void_p(None)


def alloc_frame(info, align=1):
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
    out = []
    for p in range(info.nplanes):
        rowbytes = info.plane_width[p] * info.plane_bytes_per_sample[p]
        a = np.ascontiguousarray(planes[p][:, :rowbytes])
        if info.bits_per_raw_sample > 8:
            a = a.view(np.uint16)
        out.append(a.reshape(info.plane_height[p], -1))
    return out


EAGAIN = -11


class Splitter:
    """htj2k_splitter_*: cuts a byte stream of back-to-back codestreams / JP2 files into packets (the reference's
    jpeg2000 AVCodecParser, libavcodec/jpeg2000_parser.c).  Host only: works without a GPU."""
    END_NOT_FOUND = -100

    def __init__(self):
        self.L = load_library()
        self.h = ctypes.c_void_p()
        _check(self.L.htj2k_splitter_open(ctypes.byref(self.h)), "htj2k_splitter_open")

    def find_end(self, data):
        buf = (ctypes.c_uint8 * max(len(data), 1)).from_buffer_copy(bytes(data) or b"\0")
        return self.L.htj2k_splitter_find_end(self.h, buf, len(data))

    def parse(self, data):
        """-> (bytes consumed, frame bytes or None)"""
        buf = (ctypes.c_uint8 * (len(data) + 64)).from_buffer_copy(bytes(data) + bytes(64))
        fr, n = ctypes.POINTER(ctypes.c_uint8)(), ctypes.c_int()
        used = _check(self.L.htj2k_splitter_parse(self.h, buf, len(data), ctypes.byref(fr), ctypes.byref(n)), "htj2k_splitter_parse")
        return used, (ctypes.string_at(fr, n.value) if fr else None)

    def split(self, stream, chunk=4096):
        """all frames of `stream`, fed `chunk` bytes at a time (the av_parser_parse2 loop)"""
        out, pos = [], 0
        while pos < len(stream):
            piece = stream[pos:pos + chunk]
            while True:
                used, fr = self.parse(piece)
                if fr is not None:
                    out.append(fr)
                pos += used
                piece = piece[used:]
                if not piece or (used == 0 and fr is None):
                    break
        used, fr = self.parse(b"")
        if fr:
            out.append(fr)
        return out

    def close(self):
        if self.h:
            self.L.htj2k_splitter_close(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MxfEssence(ctypes.Structure):
    """struct htj2k_mxf_essence (include/htj2k_amd.h)"""
    _fields_ = [("data", ctypes.POINTER(ctypes.c_uint8)), ("size", ctypes.c_size_t), ("klv_offset", ctypes.c_size_t),
                ("track_number", ctypes.c_uint32), ("wrapping", ctypes.c_int)]


MXF_FRAME_WRAPPED, MXF_CLIP_WRAPPED = 1, 2


def mxf_essence(data):
    """htj2k_mxf_next_essence over a whole MXF file: [(bytes, track_number, wrapping, klv_offset)] of its JPEG 2000
    picture elements (the KLV layer of libavformat/mxfdec.c).  Host only: works without a GPU."""
    L = load_library()
    L.htj2k_mxf_next_essence.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(MxfEssence)]
    data = bytes(data)
    pos, e, out = ctypes.c_size_t(0), MxfEssence(), []
    while _check(L.htj2k_mxf_next_essence(data, len(data), ctypes.byref(pos), ctypes.byref(e)), "htj2k_mxf_next_essence") == 1:
        out.append((ctypes.string_at(e.data, e.size), e.track_number, e.wrapping, e.klv_offset))
    return out


class Pipe:
    """htj2k_pipe_*: packets in, frames out (in order), `depth` batches of `batch` frames in flight"""

    def __init__(self, dec, batch=8, depth=3):
        self.dec = dec
        self.h = ctypes.c_void_p(None)
        _check(dec.L.htj2k_pipe_open(dec.h, batch, depth, ctypes.byref(self.h)), "htj2k_pipe_open")

    def send(self, data):
        """False when `depth` batches are waiting to be received.  `data`: bytes, or a (ctypes buffer, size) pair
        from packet() -- building the padded buffer costs two copies of the packet in Python"""
        if isinstance(data, tuple):                        # caller keeps the buffer alive: no copy in the library either
            buf, size = data
            r = self.dec.L.htj2k_pipe_send_ref(self.h, buf, size, None, None)
        else:
            r = self.dec.L.htj2k_pipe_send(self.h, _pkt(data), len(data))
        if r == EAGAIN:
            return False
        _check(r, "htj2k_pipe_send")
        return True

    def flush(self):
        _check(self.dec.L.htj2k_pipe_flush(self.h), "htj2k_pipe_flush")

    def receive(self, into=None):
        """-> (info, [plane arrays]) of the next frame, None when nothing is in flight; raises for a
        packet that failed (Htj2kError; the frame is consumed).  `into` = (planes, Frame) from alloc_frame to reuse buffers."""
        info = Info()
        r = self.dec.L.htj2k_pipe_info(self.h, ctypes.byref(info))
        if r == EAGAIN:
            return None
        if r < 0:
            self.dec.L.htj2k_pipe_skip(self.h)
            _check(r, "htj2k_pipe_info")
        planes, fr = into if into is not None else alloc_frame(info)
        _check(self.dec.L.htj2k_pipe_receive(self.h, ctypes.byref(fr)), "htj2k_pipe_receive")
        return info, planes_to_arrays(info, planes)

    def receive_device(self):
        """-> Frame whose data[] are device pointers (no copy), None when nothing is in flight"""
        fr = Frame()
        r = self.dec.L.htj2k_pipe_receive_device(self.h, ctypes.byref(fr))
        if r == EAGAIN:
            return None
        _check(r, "htj2k_pipe_receive_device")
        return fr

    def receive_device_ref(self):
        """-> (Frame of device pointers, token) valid until release_device(token); None when nothing is in flight"""
        fr, tok = Frame(), ctypes.c_uint64()
        r = self.dec.L.htj2k_pipe_receive_device_ref(self.h, ctypes.byref(fr), ctypes.byref(tok))
        if r == EAGAIN:
            return None
        _check(r, "htj2k_pipe_receive_device_ref")
        return fr, tok.value

    def release_device(self, token):
        _check(self.dec.L.htj2k_pipe_release_device(self.h, ctypes.c_uint64(token)), "htj2k_pipe_release_device")

    def close(self):
        if self.h:
            self.dec.L.htj2k_pipe_close(self.h)
            self.h = ctypes.c_void_p(None)


class Decoder:
    """htj2k_open / htj2k_probe / htj2k_decode / htj2k_close: the FFCodec init/decode/close trio."""

    def __init__(self, device_id=0, bitexact=0, reduction_factor=0, req_pix_fmt=-1, strict=0, max_pixels=0, frames_in_flight=0):
        self.L = load_library()
        o = Opts()
        o.device_id = device_id
        o.frames_in_flight = frames_in_flight
        o.bitexact = bitexact
        o.reduction_factor = reduction_factor
        o.req_pix_fmt = req_pix_fmt
        o.strict = strict
        o.max_pixels = max_pixels
        self.h = ctypes.c_void_p(None)
        _check(self.L.htj2k_open(ctypes.byref(o), ctypes.byref(self.h)), "htj2k_open")
        self._buf = None

    def close(self):
        if self.h:
            self.L.htj2k_close(self.h)
            self.h = ctypes.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def device_name(self):
        return self.L.htj2k_device_name(self.h).decode()

    def set_int(self, name, value):
        _check(self.L.htj2k_set_int(self.h, name.encode(), int(value)), "htj2k_set_int")

    def pipe(self, batch=8, depth=3):
        return Pipe(self, batch, depth)

    def fetch_device_frame(self, info, fr):
        """[plane arrays] of a frame whose data[] are device pointers (Pipe.receive_device, Job.device_frame)"""
        planes = []
        for p in range(info.nplanes):
            ls = fr.linesize[p]
            a = np.zeros((info.plane_height[p], ls), dtype=np.uint8)
            _check(self.L.htj2k_device_to_host(self.h, a.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(fr.data[p]),
                                               ctypes.c_size_t(a.nbytes)), "htj2k_device_to_host")
            planes.append(a)
        return planes_to_arrays(info, planes)

    def alloc_frame_pinned(self, info):
        """frame planes in page-locked host memory (htj2k_host_alloc); free with free_frame_pinned"""
        planes, fr, ptrs = [], Frame(), []
        for p in range(info.nplanes):
            ls = info.plane_width[p] * info.plane_bytes_per_sample[p]
            n = ls * info.plane_height[p]
            ptr = self.L.htj2k_host_alloc(self.h, n)
            if not ptr:
                raise MemoryError("htj2k_host_alloc")
            ptrs.append(ptr)
            a = np.ctypeslib.as_array((ctypes.c_uint8 * n).from_address(ptr)).reshape(info.plane_height[p], ls)
            planes.append(a)
            fr.data[p] = ptr
            fr.linesize[p] = ls
        return (planes, fr), ptrs

    def free_frame_pinned(self, ptrs):
        for ptr in ptrs:
            self.L.htj2k_host_free(self.h, ptr)

    def probe(self, data):
        info = Info()
        self._buf = _pkt(data)
        _check(self.L.htj2k_probe(self.h, self._buf, len(data), ctypes.byref(info)), "htj2k_probe")
        return info

    def decode(self, data, align=1):
        """-> (info, [plane arrays], bytes_consumed, Stats)"""
        info = self.probe(data)
        planes, fr = alloc_frame(info, align)
        st = Stats()
        self._buf = _pkt(data)
        r = _check(self.L.htj2k_decode(self.h, self._buf, len(data), ctypes.byref(fr), ctypes.byref(st)), "htj2k_decode")
        return info, planes_to_arrays(info, planes), r, st

    def decode_into(self, pkt, buf):
        """htj2k_decode of a packet() into the planes of alloc_frame(): nothing is copied or allocated in Python"""
        st = Stats()
        return _check(self.L.htj2k_decode(self.h, pkt[0], pkt[1], ctypes.byref(buf[1]), ctypes.byref(st)), "htj2k_decode"), st

    def job(self):
        return Job(self)

    # ---- kernel-level entry points ----
    def idwt(self, plane, border, levels, type_):
        a = np.ascontiguousarray(plane).copy()
        b = (ctypes.c_int * 4)(border[0][0], border[0][1], border[1][0], border[1][1])
        _check(self.L.htj2k_idwt_plane(self.h, a.ctypes.data_as(ctypes.c_void_p), b, levels, type_), "htj2k_idwt_plane")
        return a

    def idwt_bench(self, w, h, levels, type_, nplanes=1, iters=10):
        ms = ctypes.c_float()
        _check(self.L.htj2k_idwt_bench(self.h, w, h, levels, type_, nplanes, iters, ctypes.byref(ms)), "htj2k_idwt_bench")
        return ms.value

    def copy_bench(self, mbytes=512, iters=10):
        """GB/s (read + written) of a plain device-to-device copy kernel on this box: the measured copy ceiling"""
        g = ctypes.c_float()
        _check(self.L.htj2k_copy_bench(self.h, int(mbytes), int(iters), ctypes.byref(g)), "htj2k_copy_bench")
        return g.value

    def mct(self, type_, p0, p1, p2):
        a, b, c = (np.ascontiguousarray(x).copy() for x in (p0, p1, p2))
        _check(self.L.htj2k_mct_planes(self.h, a.ctypes.data_as(ctypes.c_void_p), b.ctypes.data_as(ctypes.c_void_p),
                                       c.ctypes.data_as(ctypes.c_void_p), a.size, type_), "htj2k_mct_planes")
        return a, b, c

    def mq_blocks(self, descs, pool, nsamples, dtype=np.int32):
        """Part-1 blocks (BlockDesc.flags & 4, bytes + trailer as in j2k_plan.h) -> (samples[nsamples], status[n])"""
        n = len(descs)
        arr = (BlockDesc * n)(*descs)
        buf = ctypes.create_string_buffer(bytes(pool) + b"\0" * 64, len(pool) + 64)
        out = np.full(nsamples, 0x7FFFFFFF if dtype == np.int32 else np.nan, dtype=dtype)
        status = np.zeros(n, dtype=np.int32)
        _check(self.L.htj2k_mq_blocks(self.h, arr, n, buf, ctypes.c_size_t(len(pool) + 64),
                                      out.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(nsamples),
                                      status.ctypes.data_as(ctypes.c_void_p)), "htj2k_mq_blocks")
        return out, status

    def ht_blocks(self, descs, pool, nsamples, dtype=np.int32):
        """descs: list of BlockDesc; pool: bytes.  -> (samples[nsamples], status[n])"""
        n = len(descs)
        arr = (BlockDesc * n)(*descs)
        buf = ctypes.create_string_buffer(bytes(pool) + b"\0" * 64, len(pool) + 64)
        out = np.full(nsamples, 0x7FFFFFFF if dtype == np.int32 else np.nan, dtype=dtype)
        status = np.zeros(n, dtype=np.int32)
        _check(self.L.htj2k_ht_blocks(self.h, arr, n, buf, ctypes.c_size_t(len(pool) + 64),
                                      out.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(nsamples),
                                      status.ctypes.data_as(ctypes.c_void_p)), "htj2k_ht_blocks")
        return out, status
