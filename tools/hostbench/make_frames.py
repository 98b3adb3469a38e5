"""the bench's 4K RGB frame as a codestream with SOP / EPH markers, as it is and with a PLT marker segment, with the default
(maximal) precincts and with 256 x 256 ones: inputs for tools/hostbench/parse_time.c
usage: python tools/hostbench/make_frames.py OUTDIR"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import vecgen, cs_rewrite
out = sys.argv[1] if len(sys.argv) > 1 else "."
img = vecgen.synth_image(3840, 2160, 3, seed=2, noise=8)
for tag, kw in (("4k", {}), ("4k_prec256", dict(prec=[(8, 8)]))):
    cs = vecgen.encode(img, mct=1, nlevels=5, cb=(6, 6), transform=1, sop=True, eph=True, **kw)
    for vn, data in cs_rewrite.variants(cs, True):
        if vn in ("same", "plt"):
            path = os.path.join(out, "%s.%s.j2c" % (tag, vn))
            open(path, "wb").write(bytes(data))
            print(path, len(data))
