import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ffmpeg_ht_amd as m
dec = m.Decoder()
for md in (2, 3):
    dec.set_int("idwt_mode", md)
    for strip in (["0"] if md == 2 else ["16", "32", "64", "128", "256"]):
        if strip == "0": os.environ.pop("HTJ2K_STRIP", None)
        else: os.environ["HTJ2K_STRIP"] = strip
        for (w, h, L, n) in ((3840, 2160, 1, 24), (1920, 1080, 1, 24), (3840, 2160, 1, 6)):
            ms = dec.idwt_bench(w, h, L, 1, n, 10)
            by = 8.0 * w * h * n
            print(f"mode {md} strip {strip} {w}x{h} L{L} x{n}: {ms*1e3:.1f} us  {by/ms/1e6:.0f} GB/s", flush=True)
