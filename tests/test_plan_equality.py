"""The product's host parser (ffmpeg-ht_amd/csrc/j2k_syntax.c, j2k_tier2.c, j2k_plan.c: table-driven marker dispatch,
closed-form geometry, 64-bit-window packet headers, flat tables) against the oracle's parser (oracle/j2k_oracle_parse.c:
a close restatement of the reference's jpeg2000dec.c / jpeg2000.c with its node tree).  The two share no code.  On every
packet they must return the same code, and for accepted packets the same plan: stream facts, tile-component table
(geometry, linelen / mod, placement), block table (offsets, sizes, M_b, step sizes, pass counts, Lcup / Lref), byte
pool, LDS sizing figures -- also when the product parser only emits the gather table and the pool is rebuilt from it.

Corpora: the stream catalogue, OpenJPEG-encoded fixtures, container variants made by tests/cs_rewrite.py (tile-parts,
TLM / PLT, PPM / PPT, COC / QCC / RGN / POC), >= 5000 random configurations, fresh OpenJPEG streams when Pillow can
encode (layers, precincts, all progressions, JP2), and seeded mutations of all of them (the error paths).
The harness (tests/native/plan_diff.c) runs under AddressSanitizer + UBSan where the runtime is installed."""
import io
import os
import shutil
import subprocess

import numpy as np
import pytest

import cs_rewrite
import streams
import vecgen

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(ROOT, "ffmpeg-ht_amd", "csrc")
ORACLE_SO = os.path.join(ROOT, "oracle", "libj2k_oracle.so")

pytestmark = pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")


@pytest.fixture(scope="module")
def plan_diff(tmp_path_factory):
    if not os.path.exists(ORACLE_SO):
        subprocess.check_call(["make", "-C", ROOT, "oracle"])
    d = tmp_path_factory.mktemp("plan_diff")
    exe = d / "plan_diff"
    srcs = [os.path.join(HERE, "native", "plan_diff.c")] + [os.path.join(CSRC, f) for f in ("j2k_syntax.c", "j2k_tier2.c", "j2k_plan.c")]
    base = ["gcc", "-g", "-std=gnu11", "-pthread", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-o", str(exe)] + srcs + ["-lm", "-ldl"]
    r = subprocess.run(base[:1] + ["-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"] + base[1:],
                       capture_output=True, text=True)
    if r.returncode != 0:
        r = subprocess.run(base[:1] + ["-O2"] + base[1:], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]

    def run(files, mutations, seed=1, workdir=d, threads=0):
        lst = workdir / ("corpus_%d.list" % abs(hash(tuple(files))))
        lst.write_text("\n".join(str(f) for f in files) + "\n")
        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
        r = subprocess.run([str(exe), "-m", str(mutations), "-s", str(seed), "-t", str(threads), ORACLE_SO, str(lst)], capture_output=True,
                           text=True, env=env, timeout=1200)
        tail = (r.stdout[-3000:], r.stderr[-3000:])
        assert r.returncode == 0, tail
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("plan_diff:")]
        line = lines[-2].split()
        parses, accepted, diffs = int(line[1]), int(line[3]), int(line[5])
        assert diffs == 0, tail
        if threads:
            par = lines[-1].replace(",", "").split()            # plan_diff: parallel tiles N sequential retries M
            return parses, accepted, int(par[3]), int(par[6])
        return parses, accepted
    return run


def _write(d, name, data):
    p = d / name
    p.write_bytes(bytes(data))
    return p


def test_catalogue_streams_and_their_mutations(plan_diff, tmp_path):
    files = [_write(tmp_path, n + ".j2c", streams.get(n)[0]) for n in streams.CASES]
    parses, accepted = plan_diff(files, 60)
    assert parses >= len(files) * 60 and accepted > parses // 3


def test_openjpeg_fixtures(plan_diff, tmp_path):
    z = np.load(os.path.join(HERE, "golden", "opj_part1.npz"))
    files = [_write(tmp_path, k, z[k].tobytes()) for k in z.files if k.endswith(".j2k")]
    assert len(files) >= 8
    parses, accepted = plan_diff(files, 300)
    assert accepted > parses // 4


REWRITE_BASES = {
    "ht_gray":          ((200, 150, 1, 8, 3), dict()),
    "ht_rgb_tiles":     ((190, 131, 3, 8, 6), dict(mct=1, tile=(100, 70), nlevels=3, offset=(7, 9), tile_offset=(2, 3))),
    "ht_rgb_rpcl":      ((190, 131, 3, 8, 5), dict(mct=1, prog=2, prec=[(7, 7), (6, 6)], nlevels=3)),
    "ht_rgb_cprl":      ((190, 131, 3, 8, 5), dict(prog=4, prec=[(7, 7), (6, 6)], nlevels=3)),
    "ht_rlcp_3p":       ((190, 131, 3, 8, 5), dict(prog=1, nlevels=4, passes=3)),
    "p1_rgb":           ((190, 131, 3, 8, 5), dict(mct=1, part1=True)),
    "p1_termall_tiles": ((190, 131, 3, 8, 6), dict(part1=True, cblk_style=0x05, tile=(64, 64), nlevels=3)),
    "mixed":            ((190, 131, 3, 8, 5), dict(mct=1, mixed=True, cb=(5, 5), nlevels=3)),
    "yuv420":           ((190, 130, 3, 8, 12, 8, (1, 2, 2), (1, 2, 2)), dict(dx=[1, 2, 2], dy=[1, 2, 2], width=190, height=130)),
}


def rewritten_streams():
    """(name, codestream) for every container variant of every base"""
    for bn, (args, kw) in REWRITE_BASES.items():
        ht = not kw.get("part1")
        cs = vecgen.encode(streams._img(*args), sop=True, eph=True, cap_extra_bits=0x1800 if ht else 0, **kw)
        for vn, data in cs_rewrite.variants(cs, ht):
            yield bn + "." + vn, data


def test_container_variants(plan_diff, tmp_path):
    """tile-parts (also interleaved across tiles, also one packet each: beyond 32 of them the reference gives up), TLM,
    PLT, PPM, PPT, COC / QCC in main and tile-part headers, RGN, POC"""
    files = [_write(tmp_path, n + ".j2c", d) for n, d in rewritten_streams()]
    assert len(files) > 100
    parses, accepted = plan_diff(files, 120)
    assert accepted > parses // 3


def test_plt_streams_read_by_several_threads(plan_diff, tmp_path):
    """with a PLT packet-length list (and one layer, no PPM / PPT) the packets of a tile are read by
    several threads (j2k_tier2.c: read_tile_parallel).  Same plans, byte pools and return codes as the oracle's parser on
    every container variant and on thousands of damaged copies (where the list and the packets disagree the frame is
    parsed again the sequential way); the variants with a usable list must really have gone the parallel way."""
    files = [_write(tmp_path, n + ".j2c", d) for n, d in rewritten_streams()]
    big = vecgen.encode(streams._img(1024, 640, 3, 8, 21), sop=True, eph=True, mct=1, nlevels=5, prec=[(7, 7), (6, 6)], cb=(5, 5))
    for vn, data in cs_rewrite.variants(big, True):
        if "plt" in vn:
            files.append(_write(tmp_path, "big." + vn + ".j2c", data))
    parses, accepted, ptiles, retries = plan_diff(files, 150, seed=5, threads=4)
    assert accepted > parses // 3
    assert ptiles > 1000 and retries > 100, (ptiles, retries)
    only_plt = [f for f in files if ".plt." in str(f) or "tlm_plt" in str(f)]
    parses, accepted, ptiles, retries = plan_diff(only_plt, 0, threads=3)
    assert ptiles >= 2 * len(only_plt) and retries == 0, (ptiles, retries, len(only_plt))


def random_stream(rng, it):
    """one small random configuration of the vector factory, or None when the encoder refuses it"""
    even = rng.random() < 0.5
    w = int(rng.integers(1, 6)) * 32 if even else int(rng.integers(1, 150))
    h = int(rng.integers(1, 6)) * 16 if even else int(rng.integers(1, 120))
    nc = int(rng.choice([1, 3, 3, 4]))
    depth = int(rng.choice([8, 8, 8, 10, 12, 16]))
    nl = int(rng.integers(0, 6))
    cbw = int(rng.integers(2, 8)); cbh = int(rng.integers(2, min(10, 12 - cbw) + 1))
    kw = dict(nlevels=nl, cb=(cbw, cbh), depth=depth)
    mode = int(rng.integers(0, 10))
    if 5 < mode <= 7: kw.update(part1=True, cblk_style=int(rng.choice([0, 0, 1, 4, 8, 0x20, 5, 0x2F])))
    elif mode > 7: kw.update(mixed=True)
    if rng.random() < 0.25 and not kw.get("part1"): kw["passes"] = int(rng.choice([2, 3]))
    if rng.random() < 0.3: kw.update(transform=0, qstep=float(rng.choice([0.25, 1.0, 4.0])))
    sub = nc == 3 and rng.random() < 0.25
    dx = [1, 2, 2] if sub else None
    dy = [1, int(rng.choice([1, 2])), 0] if sub else None
    if sub: dy[2] = dy[1]
    if nc >= 3 and not sub and rng.random() < 0.6: kw["mct"] = 1
    if rng.random() < 0.25: kw["tile"] = (int(rng.choice([32, 64, 96, 100])), int(rng.choice([32, 48, 64, 70])))
    if rng.random() < 0.2:
        kw["offset"] = (int(rng.integers(0, 9)), int(rng.integers(0, 9)))
        if "tile" in kw and rng.random() < 0.5:
            kw["tile_offset"] = (int(rng.integers(0, kw["offset"][0] + 1)), int(rng.integers(0, kw["offset"][1] + 1)))
    if rng.random() < 0.3: kw["prog"] = int(rng.integers(0, 5))
    if rng.random() < 0.25: kw["prec"] = [(int(rng.integers(5, 9)), int(rng.integers(5, 9))), (int(rng.integers(4, 8)), int(rng.integers(4, 8)))]
    if rng.random() < 0.15: kw.update(sop=True, eph=bool(rng.integers(0, 2)))
    if rng.random() < 0.1 and not kw.get("part1") and not kw.get("mixed"): kw["placeholder_sets"] = int(rng.integers(1, 3))
    if rng.random() < 0.1: kw["guard_bits"] = int(rng.integers(1, 5))
    if rng.random() < 0.1 and kw.get("part1"): kw["drop_passes"] = int(rng.integers(1, 6))
    if rng.random() < 0.05: kw["force_include"] = True
    if rng.random() < 0.05: kw["psot_zero"] = True
    try:
        img = vecgen.synth_image(w, h, nc, depth=depth, seed=it + 7, noise=int(rng.choice([0, 4, 20])), dx=dx, dy=dy)
        if sub: kw.update(dx=dx, dy=dy, width=w, height=h)
        data = vecgen.encode(img, **kw)
        if rng.random() < 0.15 and nc in (1, 3):
            cs = 17 if nc == 1 else int(rng.choice([16, 18]))
            cdef = None
            if nc == 3 and rng.random() < 0.5:
                perm = rng.permutation(3)
                cdef = [(c, 0, int(perm[c]) + 1) for c in range(3)]
            res = (300, 1, int(rng.choice([150, 300])), 1, 0, int(rng.integers(0, 3))) if rng.random() < 0.3 else None
            data = vecgen.jp2_wrap(data, w, h, nc, depth, colourspace=cs, cdef=cdef, res=res)
    except Exception:
        return None
    return data


def test_five_thousand_random_configurations(plan_diff, tmp_path):
    rng = np.random.default_rng(20261004)
    files = []
    for it in range(5400):
        data = random_stream(rng, it)
        if data is not None:
            files.append(_write(tmp_path, "r%04d.j2c" % it, data))
    assert len(files) >= 5000, len(files)
    parses, accepted = plan_diff(files, 6)           # 8 plain variants (lowres, bitexact, strict, headers only) + 6 mutations each
    assert parses >= 14 * 5000 and accepted > parses // 2


def test_fresh_openjpeg_streams(plan_diff, tmp_path):
    """Part-1 streams from OpenJPEG's encoder (through Pillow, when it is there): quality layers, precincts, the five
    progression orders, tiles with offsets, PLT, JP2 boxes -- what the vector factory cannot make"""
    PIL = pytest.importorskip("PIL")
    from PIL import Image, features
    if not features.check_codec("jpg_2000"):
        pytest.skip("Pillow without OpenJPEG")
    del PIL
    import random
    rnd = random.Random(5)
    files = []
    for i in range(120):
        w = rnd.choice([17, 33, 64, 97, 131, 200]); h = rnd.choice([9, 31, 64, 77, 128])
        mode = rnd.choice(["L", "L", "RGB", "RGB", "RGBA", "LA", "I;16"])
        c = {"L": 1, "RGB": 3, "RGBA": 4, "LA": 2, "I;16": 1}[mode]
        y, x = np.mgrid[0:h, 0:w]
        a = np.clip((128 + 60 * np.sin(x / 17.0 + i) + 50 * np.cos(y / 23.0))[..., None] +
                    np.random.default_rng(i).integers(-9, 10, (h, w, c)), 0, 255).astype(np.uint8)
        img = Image.fromarray(a[..., 0].astype(np.uint16) * 200) if mode == "I;16" else Image.fromarray(a[..., 0] if c == 1 else a, mode)
        kw = dict(irreversible=rnd.random() < 0.3)
        nl = rnd.choice([1, 1, 2, 3, 5])
        if nl > 1:
            kw["quality_mode"] = "rates"
            kw["quality_layers"] = sorted([rnd.choice([60, 40, 20, 10, 5, 2]) for _ in range(nl - 1)], reverse=True) + [1]
        cb = rnd.choice([(64, 64), (32, 32), (16, 64), (64, 16), (4, 4), (32, 64)]); kw["codeblock_size"] = cb
        if rnd.random() < 0.5:
            ps = rnd.choice([(64, 64), (128, 128), (32, 32), (256, 256), (128, 64)])
            if ps[0] >= cb[0] and ps[1] >= cb[1]: kw["precinct_size"] = ps
        kw["progression"] = rnd.choice(["LRCP", "RLCP", "RPCL", "PCRL", "CPRL"])
        mind = min(w, h)
        if rnd.random() < 0.4:
            kw["tile_size"] = rnd.choice([(64, 64), (32, 48), (100, 70)])
            mind = min(mind, 16)
            if rnd.random() < 0.3:
                kw["offset"] = (rnd.choice([0, 3, 8]), rnd.choice([0, 5]))
                kw["tile_offset"] = (min(kw["offset"][0], rnd.choice([0, 2])), min(kw["offset"][1], rnd.choice([0, 3])))
        kw["num_resolutions"] = min(rnd.choice([1, 2, 3, 4, 6]), max(1, int(np.log2(max(mind, 2))) - 1))   # OpenJPEG asserts on 1-sample bands
        if c >= 3: kw["mct"] = rnd.choice([0, 1])
        kw["plt"] = rnd.random() < 0.5
        kw["no_jp2"] = rnd.random() < 0.6
        b = io.BytesIO()
        try:
            img.save(b, "JPEG2000", **kw)
        except Exception:
            continue
        files.append(_write(tmp_path, "o%03d.%s" % (i, "j2k" if kw["no_jp2"] else "jp2"), b.getvalue()))
    assert len(files) > 60
    parses, accepted = plan_diff(files, 100)
    assert accepted > parses // 3
