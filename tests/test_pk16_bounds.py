"""The packed 16-bit IDWT kernels (knob "idwt_pk") are exact only if no intermediate sum leaves 16 bits; the host decides
that per launch by interval arithmetic (htj2k_pk16_lift_bound / htj2k_pk16_bounds, htj2k_device.hip).  This test holds the
decision against a numpy model of what the kernel computes -- one level of the inverse 5/3 transform
(jpeg2000dwt.c:309-385: horizontal then vertical lifting) and the inverse RCT (jpeg2000dsp.c:78-91) in wrapping int16
arithmetic, with the two saturating adds and the clip to 8 bits of the fused store -- and the same in int64:
  * wherever the bound function says "fits", the two agree on extreme and on random inputs within the bounds, and no
    output exceeds the bound it returned;
  * the function is not vacuous: just outside what it accepts there are inputs on which 16-bit arithmetic goes wrong.
Runs on the CPU: the functions are plain host code behind the C ABI."""
import ctypes
import itertools

import numpy as np
import pytest

import ffmpeg_ht_amd as m


@pytest.fixture(scope="module")
def L():
    lib = m.load_library()
    lib.htj2k_pk16_lift_bound.restype = ctypes.c_long
    lib.htj2k_pk16_lift_bound.argtypes = [ctypes.c_long] * 4
    lib.htj2k_pk16_bounds.restype = ctypes.c_int
    lib.htj2k_pk16_bounds.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    return lib


def lift53(ll, hl, lh, hh, dt):
    """one 2-D synthesis level on a patch: ll/hl/lh/hh are (rows, cols) arrays of the four bands; arithmetic in dtype dt
    (np.int16 wraps like the v_pk_* instructions, np.int64 is the reference).  Symmetric extension at the patch edges."""
    def s1(c, a, b):            # even sample: c - ((a + b + 2) >> 2)
        return (c - ((a + b + dt(2)) >> dt(2))).astype(dt)
    def s2(c, a, b):            # odd sample: c + ((a + b) >> 1)
        return (c + ((a + b) >> dt(1))).astype(dt)
    def line(lo, hi):           # 1-D along the last axis: lo = even-position band, hi = odd-position band
        left = np.concatenate([hi[..., :1], hi[..., :-1]], -1)          # the odd sample left of each even one (mirror at 0)
        e = s1(lo, left, hi)
        right = np.concatenate([e[..., 1:], e[..., -1:]], -1)           # the even sample right of each odd one (mirror at the end)
        o = s2(hi, e, right)
        return e, o
    with np.errstate(over="ignore"):
        ll, hl, lh, hh = (x.astype(dt) for x in (ll, hl, lh, hh))
        le, lo = line(ll, hl)                    # vertical-low rows, horizontally synthesised: even and odd columns
        he, ho = line(lh, hh)                    # vertical-high rows
        out = []
        for low, high in ((le, he), (lo, ho)):   # vertical, per column parity
            e, o = line(low.T, high.T)
            out.append((e.T, o.T))
    return out                                   # [(even rows, odd rows) of even columns, (...) of odd columns]


def rct_clip(y, cb, cr, dt):
    with np.errstate(over="ignore"):
        y, cb, cr = (x.astype(dt) for x in (y, cb, cr))
        g = (y - ((cr + cb) >> dt(2))).astype(dt)
        if dt == np.int16:                       # v_pk_add_i16 clamp
            r = np.clip(g.astype(np.int32) + cr, -32768, 32767).astype(dt)
            b = np.clip(g.astype(np.int32) + cb, -32768, 32767).astype(dt)
        else:
            r, b = g + cr, g + cb
    return [np.clip(x, -128, 127) for x in (r, g, b)]


def patches(bounds, rng, n_random=40):
    """4 x 4 patches of the four bands: every combination of +-bound per band (constant patches), checkerboards, and
    random values within the bounds"""
    shape = (4, 4)
    for signs in itertools.product((-1, 1), repeat=4):
        yield [np.full(shape, s * b) for s, b in zip(signs, bounds)]
        chk = (np.indices(shape).sum(0) & 1) * 2 - 1
        yield [chk * s * b for s, b in zip(signs, bounds)]
        yield [(chk if k & 1 else -chk) * s * b for k, (s, b) in enumerate(zip(signs, bounds))]
    for _ in range(n_random):
        yield [rng.integers(-b, b + 1, shape) if b else np.zeros(shape, np.int64) for b in bounds]
        yield [np.where(rng.random(shape) < 0.5, -b, b) for b in bounds]


def test_lift_bound_is_sound_and_tight(L):
    rng = np.random.default_rng(5)
    accepted = rejected = 0
    for mb_hl, mb_hh in ((7, 8), (10, 11), (11, 12), (12, 13), (13, 13), (13, 14), (14, 14), (14, 15), (15, 15)):
        for k in range(9, 17):
            bounds = (1 << (k - 1), (1 << mb_hl) - 1, (1 << mb_hl) - 1, (1 << mb_hh) - 1)
            top = L.htj2k_pk16_lift_bound(*bounds)
            if top >= 0:
                accepted += 1
                for p in patches(bounds, rng):
                    a = lift53(*p, np.int16)
                    b = lift53(*p, np.int64)
                    for (ae, ao), (be, bo) in zip(a, b):
                        assert np.array_equal(ae, be) and np.array_equal(ao, bo), (bounds, "16-bit arithmetic differs")
                        assert max(np.abs(be).max(), np.abs(bo).max()) <= top, (bounds, top)
            else:
                rejected += 1
                # not vacuous: some patch within these bounds does overflow an int16 intermediate or result
                wrong = False
                for p in patches(bounds, rng, n_random=10):
                    a, b = lift53(*p, np.int16), lift53(*p, np.int64)
                    big = max(max(np.abs(e).max(), np.abs(o).max()) for e, o in b)
                    if big > 32767 or any(not (np.array_equal(ae, be) and np.array_equal(ao, bo)) for (ae, ao), (be, bo) in zip(a, b)):
                        wrong = True
                        break
                # the bound is conservative by at most a factor of two in the inputs: with every bound halved it must accept
                half = [x // 2 for x in bounds]
                assert wrong or L.htj2k_pk16_lift_bound(*half) >= 0, bounds
    assert accepted >= 20 and rejected >= 10


def test_bounds_of_an_rgb_group_with_the_rct(L):
    rng = np.random.default_rng(6)
    # the bench's frames: 8-bit RGB, two guard bits -- M_b 10 / 10 / 11 for the luma's finest HL / LH / HH, 11 / 11 / 12 for chroma
    def group(k):
        return [(1 << (k - 1), 1023, 1023, 2047), (1 << (k - 1), 2047, 2047, 4095), (1 << (k - 1), 2047, 2047, 4095)]
    def ask(g, rct):
        arr = ((ctypes.c_long * 4) * 3)(*[(ctypes.c_long * 4)(*c) for c in g])
        return L.htj2k_pk16_bounds(arr, 3, rct)
    assert ask(group(11), 1) == 1 and ask(group(12), 1) == 0, "k = 11 is what the device layer finds for the bench's frames"
    assert ask(group(12), 0) == 1                                # without the RCT the lifting alone allows one bit more
    g = group(11)
    for _ in range(60):
        comps16, comps64 = [], []
        for c in g:
            p = [np.where(rng.random((4, 4)) < 0.5, -b, b) if rng.random() < 0.5 else rng.integers(-b, b + 1, (4, 4)) for b in c]
            comps16.append(lift53(*p, np.int16)); comps64.append(lift53(*p, np.int64))
        for col in range(2):
            for row in range(2):
                y16, cb16, cr16 = (comps16[c][col][row] for c in range(3))
                y64, cb64, cr64 = (comps64[c][col][row] for c in range(3))
                assert all(np.array_equal(a, b) for a, b in zip(rct_clip(y16, cb16, cr16, np.int16), rct_clip(y64, cb64, cr64, np.int64)))
    # extreme corner: every band at its bound with the signs that push one output highest
    for sy, sc in itertools.product((-1, 1), repeat=2):
        ps = [[np.full((4, 4), s * b) * np.array([[1, -1, 1, -1]] * 4) ** t for t, b in enumerate(c)] for s, c in zip((sy, sc, sc), g)]
        c16 = [lift53(*p, np.int16) for p in ps]; c64 = [lift53(*p, np.int64) for p in ps]
        for col in range(2):
            for row in range(2):
                a = rct_clip(*(c16[c][col][row] for c in range(3)), np.int16)
                b = rct_clip(*(c64[c][col][row] for c in range(3)), np.int64)
                assert all(np.array_equal(x, y) for x, y in zip(a, b))
