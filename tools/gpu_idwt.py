import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ffmpeg_ht_amd as m
dec = m.Decoder()
for md in (1, 2):
    dec.set_int("idwt_mode", md)
    for typ in (1, 0, 2):
        for (w, h, L, n) in ((3840, 2160, 5, 3), (3840, 2160, 5, 24), (7680, 4320, 6, 1)):
            ms = dec.idwt_bench(w, h, L, typ, n, 10)
            by = sum(8.0 * (-(-w // (1 << k))) * (-(-h // (1 << k))) for k in range(L)) * n
            print(f"mode {md} type {typ} {w}x{h} L{L} x{n}: {ms:.3f} ms  {by/ms/1e6:.0f} GB/s  frac {by/ms/1e6/8000:.3f}", flush=True)
