"""ctypes binding of the HTJ2K test-vector factory (tools/vecgen/htj2k_enc.c) and the
seeded synthetic images of BASELINE.md section 3.  Test tooling only."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = os.path.join(ROOT, "tools", "vecgen", "libhtj2k_vecgen.so")


class EncParams(ctypes.Structure):
    _fields_ = (
        [(n, ctypes.c_int) for n in ("width", "height", "x_off", "y_off", "tile_w", "tile_h", "tx_off", "ty_off", "ncomp")]
        + [("depth", ctypes.c_int * 4), ("sgnd", ctypes.c_int * 4), ("dx", ctypes.c_int * 4), ("dy", ctypes.c_int * 4)]
        + [(n, ctypes.c_int) for n in ("nlevels", "cb_w_log2", "cb_h_log2", "transform", "mct", "guard_bits", "prog_order", "nprec")]
        + [("prec_w_log2", ctypes.c_int * 34), ("prec_h_log2", ctypes.c_int * 34), ("qstep", ctypes.c_double)]
        + [(n, ctypes.c_int) for n in ("expn_bias", "passes", "placeholder_sets", "cblk_style", "sop", "eph", "force_include",
                                       "never_empty_packets", "psot_zero", "rsiz", "cap_extra_bits")]
        + [("comment", ctypes.c_char_p), ("part1", ctypes.c_int), ("p1_drop_passes", ctypes.c_int),
           ("mixed", ctypes.c_int)]
    )


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            subprocess.check_call(["make", "-C", ROOT, "vecgen"])
        _lib = ctypes.CDLL(_LIB)
        _lib.htj2k_encode.restype = ctypes.c_int
        _lib.htj2k_encode_block.restype = ctypes.c_int
    return _lib


def encode(comps, depth=8, sgnd=False, dx=None, dy=None, nlevels=5, cb=(6, 6), transform=1, mct=0,
           guard_bits=0, prog=0, prec=None, qstep=1.0 / 32, expn_bias=0, passes=1, placeholder_sets=0,
           vsc=False, sop=False, eph=False, force_include=False, never_empty_packets=False, psot_zero=False,
           tile=(0, 0), offset=(0, 0), tile_offset=(0, 0), rsiz=0, cap_extra_bits=0, width=None, height=None,
           comment=None, part1=False, cblk_style=None, drop_passes=0, mixed=False):
    """comps: list of 2-D integer arrays (one per component, already subsampled).  Returns bytes."""
    if isinstance(comps, np.ndarray):
        comps = [comps] if comps.ndim == 2 else [comps[..., i] for i in range(comps.shape[-1])]
    p = EncParams()
    nc = len(comps)
    dx = dx or [1] * nc
    dy = dy or [1] * nc
    p.ncomp = nc
    p.x_off, p.y_off = offset
    p.tx_off, p.ty_off = tile_offset
    p.tile_w, p.tile_h = tile
    if width is None:
        # image area from component 0 (its subsampling must be 1 unless width/height are given)
        width, height = comps[0].shape[1] * dx[0], comps[0].shape[0] * dy[0]
    p.width, p.height = width, height
    depths = depth if isinstance(depth, (list, tuple)) else [depth] * nc
    for i in range(nc):
        p.depth[i] = depths[i]
        p.sgnd[i] = int(bool(sgnd))
        p.dx[i] = dx[i]
        p.dy[i] = dy[i]
    p.nlevels = nlevels
    p.cb_w_log2, p.cb_h_log2 = cb
    p.transform = transform
    p.mct = mct
    p.guard_bits = guard_bits
    p.prog_order = prog
    if prec:
        p.nprec = len(prec)
        for i, (pw, ph) in enumerate(prec):
            p.prec_w_log2[i] = pw
            p.prec_h_log2[i] = ph
    p.qstep = qstep
    p.expn_bias = expn_bias
    p.passes = passes
    p.placeholder_sets = placeholder_sets
    p.cblk_style = (0x08 if vsc else 0) if cblk_style is None else cblk_style
    p.part1 = int(part1)
    p.p1_drop_passes = drop_passes
    p.mixed = int(mixed)
    p.sop, p.eph = int(sop), int(eph)
    p.force_include = int(force_include)
    p.never_empty_packets = int(never_empty_packets)
    p.psot_zero = int(psot_zero)
    p.rsiz = rsiz
    p.cap_extra_bits = cap_extra_bits
    p.comment = comment
    arrs = [np.ascontiguousarray(c, dtype=np.int32) for c in comps]
    ptrs = (ctypes.POINTER(ctypes.c_int32) * 4)()
    for i, a in enumerate(arrs):
        ptrs[i] = a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
    out = ctypes.POINTER(ctypes.c_uint8)()
    n = ctypes.c_size_t()
    r = lib().htj2k_encode(ctypes.byref(p), ptrs, ctypes.byref(out), ctypes.byref(n))
    if r != 0:
        raise RuntimeError("htj2k_encode failed: %d" % r)
    data = ctypes.string_at(out, n.value)
    lib().htj2k_enc_free(out)
    return data


def encode_block(vals, passes=1, causal=False):
    """vals: 2-D int array of quantisation indices -> (bytes(Dcup||Dref, padded), lcup, lref, max_U)."""
    a = np.ascontiguousarray(vals, dtype=np.int32)
    h, w = a.shape
    out = ctypes.POINTER(ctypes.c_uint8)()
    lcup, lref, mu = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    r = lib().htj2k_encode_block(a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), w, h, passes, int(causal),
                                 ctypes.byref(out), ctypes.byref(lcup), ctypes.byref(lref), ctypes.byref(mu))
    if r != 0:
        raise RuntimeError("htj2k_encode_block failed: %d" % r)
    data = ctypes.string_at(out, lcup.value + lref.value + 8)
    lib().htj2k_enc_free(out)
    return data, lcup.value, lref.value, mu.value


def encode_block_p1(vals, band=0, style=0, drop_passes=0):
    """vals: 2-D int array of quantisation indices -> (segment bytes back to back, [segment lengths], [passes per
    segment], bit-planes K, coding passes)"""
    a = np.ascontiguousarray(vals, dtype=np.int32)
    h, w = a.shape
    out = ctypes.POINTER(ctypes.c_uint8)()
    kb, npz, ns = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    seglen, segpasses = (ctypes.c_int * 128)(), (ctypes.c_int * 128)()
    r = lib().htj2k_encode_block_p1(a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), w, h, band, style, drop_passes,
                                    ctypes.byref(out), ctypes.byref(kb), ctypes.byref(npz), ctypes.byref(ns), seglen, segpasses)
    if r != 0:
        raise RuntimeError("htj2k_encode_block_p1 failed: %d" % r)
    lens = [seglen[i] for i in range(ns.value)]
    data = ctypes.string_at(out, sum(lens)) if kb.value else b""
    if kb.value:
        lib().htj2k_enc_free(out)
    return data, lens, [segpasses[i] for i in range(ns.value)], kb.value, npz.value


def jp2_wrap(codestream, width, height, ncomp, depth, colourspace=None, cdef=None, res=None, palette=None):
    """Minimal JP2 file around a codestream (jP signature, ftyp, jp2h{ihdr,colr[,cdef][,res ]}, jp2c)."""
    import struct

    def box(t, payload):
        return struct.pack(">I4s", 8 + len(payload), t) + payload

    ihdr = struct.pack(">IIHBBBB", height, width, ncomp, depth - 1, 7, 0, 0)
    inner = box(b"ihdr", ihdr)
    if colourspace is not None:
        inner += box(b"colr", struct.pack(">BBBI", 1, 0, 0, colourspace))
    if cdef:
        payload = struct.pack(">H", len(cdef))
        for cn, typ, asoc in cdef:
            payload += struct.pack(">HHH", cn, typ, asoc)
        inner += box(b"cdef", payload)
    if palette is not None:
        # pclr: NE entries, 3 columns of 8 bits; cmap: component 0 through palette columns 0..2
        payload = struct.pack(">HBBBB", len(palette), 3, 7, 7, 7)
        for r, g, b in palette:
            payload += struct.pack(">BBB", r, g, b)
        inner += box(b"pclr", payload)
        inner += box(b"cmap", b"".join(struct.pack(">HBB", 0, 1, k) for k in range(3)))
    if res:
        vn, vd, hn, hd, ve, he = res
        inner += box(b"res ", box(b"resc", struct.pack(">HHHHBB", vn, vd, hn, hd, ve, he)))
    return (struct.pack(">I4sI", 12, b"jP  ", 0x0D0A870A) + box(b"ftyp", b"jp2 " + b"\0\0\0\0" + b"jp2 ")
            + box(b"jp2h", inner) + box(b"jp2c", codestream))


def synth_image(width, height, ncomp=1, depth=8, seed=1, noise=8, dx=None, dy=None, signed=False):
    """Seeded natural-looking test image: smooth field + LCG noise
    r = r*1664525 + 1013904223.  Returns a list of int32 component arrays."""
    dx = dx or [1] * ncomp
    dy = dy or [1] * ncomp
    out = []
    amp = (1 << depth) * 0.35
    mid = 0 if signed else (1 << (depth - 1))
    for c in range(ncomp):
        w, h = -(-width // dx[c]), -(-height // dy[c])
        y, x = np.mgrid[0:h, 0:w].astype(np.float64)
        x *= dx[c]
        y *= dy[c]
        f = (np.sin(x / (37.0 + 5 * c) + seed) + np.cos(y / (23.0 + 3 * c) - seed) + np.sin((x + y) / 91.0)) / 3.0
        n = w * h
        # vectorised LCG: r_k = a^k r_0 + c (a^k - 1)/(a - 1) mod 2^32, generated by doubling
        r = np.empty(n, dtype=np.uint32)
        r0 = np.uint32((seed * 2654435761 + c * 40503 + 12345) & 0xFFFFFFFF)
        a, cc = np.uint64(1664525), np.uint64(1013904223)
        r[0] = r0
        filled = 1
        A, C = a, cc
        m = np.uint64(0xFFFFFFFF)
        while filled < n:
            k = min(filled, n - filled)
            r[filled:filled + k] = ((r[:k].astype(np.uint64) * A + C) & m).astype(np.uint32)
            C = (C * A + C) & m
            A = (A * A) & m
            filled += k
        nz = ((r >> np.uint32(16)) % np.uint32(2 * noise + 1)).astype(np.int64) - noise if noise else 0
        v = np.rint(f * amp).astype(np.int64) + mid + (nz.reshape(h, w) if noise else 0)
        lo, hi = (-(1 << (depth - 1)), (1 << (depth - 1)) - 1) if signed else (0, (1 << depth) - 1)
        out.append(np.clip(v, lo, hi).astype(np.int32))
    return out
