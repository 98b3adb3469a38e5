"""Part-1 (MQ) block decoder on the GPU: parity vs the oracle on the p1_* streams and the OpenJPEG-encoded
fixtures, then timing on a 4K frame."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ffmpeg_ht_amd as m
import oracle, vecgen, streams

dec = m.Decoder()
orc = oracle.OracleDecoder()
bad = 0
for name in sorted(n for n in streams.CASES if n.startswith("p1_")):
    data, kw = streams.get(name)
    dec.set_int("bitexact", kw.get("bitexact", 0)); dec.set_int("reduction_factor", kw.get("reduction_factor", 0))
    info_o, planes_o, n_o = orc.decode(data, **kw)
    try:
        info, planes, n, st = dec.decode(data)
    except Exception as e:
        print("FAIL", name, repr(e), flush=True); bad += 1; continue
    ok = all(np.array_equal(a, b) for a, b in zip(planes, planes_o))
    if not ok:
        bad += 1
        d = [int(np.abs(a.astype(np.int64) - b.astype(np.int64)).max()) for a, b in zip(planes, planes_o)]
        nz = [int((a != b).sum()) for a, b in zip(planes, planes_o)]
        print("MISMATCH", name, "maxdiff", d, "count", nz, flush=True)
    else:
        print("ok", name, flush=True)
dec.set_int("bitexact", 0); dec.set_int("reduction_factor", 0)
fx = np.load(os.path.join(ROOT, "tests", "golden", "opj_part1.npz"))
for name in sorted(k[:-4] for k in fx.files if k.endswith(".j2k")):
    data = fx[name + ".j2k"].tobytes()
    pix = fx[name + ".pix"]
    info, planes, n, st = dec.decode(data)
    got = planes[0].reshape(pix.shape)
    diff = int(np.abs(got.astype(int) - pix.astype(int)).max())
    tol = 0 if fx[name + ".lossless"][0] else 1
    print("ok" if diff <= tol else "MISMATCH", name, "maxdiff", diff, flush=True)
    bad += diff > tol
print("parity mismatches:", bad, flush=True)

if os.environ.get("TIMING", "1") == "1":
    nb = int(os.environ.get("BATCH", "4"))
    img = vecgen.synth_image(3840, 2160, 3, seed=2)
    data = vecgen.encode(img, mct=1, part1=True)
    print("4K Part-1 stream:", len(data), "bytes", flush=True)
    job = dec.job()
    job.parse_batch([data] * nb); job.upload(); job.run(); job.wait()
    t0 = time.perf_counter()
    for _ in range(3):
        job.run()
    job.wait()
    dt = (time.perf_counter() - t0) / 3
    ms = job.stage_ms()
    print("batch %d: %.2f ms/step  %.1f Mpixel/s  stages %s" % (nb, dt * 1e3, nb * 3840 * 2160 / dt / 1e6, ms), flush=True)
    if os.environ.get("SWEEP", "0") == "1":
        for nb2 in [int(v) for v in os.environ.get("SWEEP_BATCHES", "16,48").split(",")]:
            job.parse_batch([data] * nb2); job.upload(); job.run(); job.wait()
            t0 = time.perf_counter()
            for _ in range(3):
                job.run()
            job.wait()
            dt = (time.perf_counter() - t0) / 3
            print("batch %d: %.2f ms/step  %.1f Mpixel/s  stages %s" % (nb2, dt * 1e3, nb2 * 3840 * 2160 / dt / 1e6, job.stage_ms()), flush=True)
        t0 = time.perf_counter()
        orc.decode(data)
        dt = time.perf_counter() - t0
        print("oracle (1 core): %.1f Mpixel/s" % (3840 * 2160 / dt / 1e6), flush=True)
