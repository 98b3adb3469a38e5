"""debug helper: where does the streaming IDWT (fused / unfused) differ from the oracle?"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ffmpeg_ht_amd as m
import oracle, streams

dec = m.Decoder()
orc = oracle.OracleDecoder()
names = sys.argv[1:] or sorted(streams.CASES)
bad = 0
for fuse in (0, 1):
    dec.set_int("idwt_mode", 3); dec.set_int("fuse_pack", fuse)
    for name in names:
        data, kw = streams.get(name)
        dec.set_int("bitexact", kw.get("bitexact", 0)); dec.set_int("reduction_factor", kw.get("reduction_factor", 0))
        io, po, _ = orc.decode(data, **kw)
        i, p, _, st = dec.decode(data)
        for k, (a, b) in enumerate(zip(p, po)):
            if not np.array_equal(a, b):
                d = np.argwhere(a != b)
                bad += 1
                print("fuse=%d %s plane %d shape %s: %d diffs, rows %d..%d cols %d..%d first %s got %s want %s" % (
                    fuse, name, k, a.shape, len(d), d[:, 0].min(), d[:, 0].max(), d[:, 1].min(), d[:, 1].max(), d[0],
                    a[tuple(d[0])], b[tuple(d[0])]))
                break
print("bad", bad)
