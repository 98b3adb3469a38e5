"""The oracle's DWT restatement against the reference's own golden file for its DWT unit
test (libavcodec/tests/jpeg2000dwt.c, golden tests/ref/fate/j2k-dwt, copied verbatim to
tests/golden/j2k-dwt.ref).  The unit test seeds an LFG (libavutil/lfg.c:30-45,
lfg.h:53-57) with 1, fills a 256x256 array with lfg % 2048, draws 100 random borders and
decomposition depths, runs forward+inverse of each transform and prints error sums."""
import hashlib
import os
import struct

import numpy as np

import oracle

HERE = os.path.dirname(os.path.abspath(__file__))


class LFG:
    """av_lfg_init / av_lfg_get"""

    def __init__(self, seed):
        self.state = [0] * 64
        tmp = bytes(16)
        for i in range(8, 64, 4):
            tmp = struct.pack("<I", seed) + bytes([i]) + tmp[5:]
            tmp = hashlib.md5(tmp).digest()
            self.state[i:i + 4] = struct.unpack("<4I", tmp)
        self.index = 0

    def get(self):
        a = (self.state[(self.index - 24) & 63] + self.state[(self.index - 55) & 63]) & 0xFFFFFFFF
        self.state[self.index & 63] = a
        self.index += 1
        return a


def run_reference_unit_test():
    prng = LFG(1)
    W = 256
    ref = np.array([prng.get() % 2048 for _ in range(W * W)], dtype=np.int32)
    lines = []
    for _ in range(100):
        b = [prng.get() % W for _ in range(4)]
        border = [[b[0], b[1]], [b[2], b[3]]]
        if border[0][0] >= border[0][1] or border[1][0] >= border[1][1]:
            continue
        levels = prng.get() % 32
        w, h = border[0][1] - border[0][0], border[1][1] - border[1][0]
        n = w * h
        # the reference transforms the first w*h elements of the flat array in place
        for typ, name in ((oracle_types["53"], "5/3i"), (oracle_types["97i"], "9/7i")):
            arr = ref.copy()
            sub = arr[:n].reshape(h, w)
            enc = oracle.fdwt(sub, border, levels, typ)
            if typ == oracle_types["97i"] and levels > 0:
                enc = (enc.astype(np.int64) << 8).astype(np.int32)
            dec = oracle.idwt(enc, border, levels, typ)
            arr[:n] = dec.reshape(-1)
            diff = arr.astype(np.int64) - ref
            maxdiff = 0 if typ == oracle_types["53"] else min(7 + 5 * levels, 15 + 3 * levels)
            assert np.abs(diff).max() <= maxdiff
            err2 = int((diff * diff).sum())
            lines.append("%s, decomp:%2d border %3d %3d %3d %3d milli-err2:%9d" %
                         (name, levels, b[0], b[1], b[2], b[3], 1000 * err2 // n))
        arrf = ref.astype(np.float32)
        sub = arrf[:n].reshape(h, w)
        dec = oracle.idwt(oracle.fdwt(sub, border, levels, oracle_types["97f"]), border, levels, oracle_types["97f"])
        arrf[:n] = dec.reshape(-1)
        d = arrf - ref.astype(np.float32)          # float subtraction, as the C test does
        assert np.abs(d).max() <= 0.05
        err2 = float((d.astype(np.float64) * d.astype(np.float64)).sum())
        lines.append("9/7f, decomp:%2d border %3d %3d %3d %3d err2:%20.3f" % (levels, b[0], b[1], b[2], b[3], err2 / n))
    return lines


oracle_types = {"97f": 0, "53": 1, "97i": 2}


def test_j2k_dwt_golden_file():
    want = open(os.path.join(HERE, "golden", "j2k-dwt.ref")).read().splitlines()
    got = run_reference_unit_test()
    assert got == want
