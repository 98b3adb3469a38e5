"""soak: many frames of mixed formats through the asynchronous pipeline with random flush points; every frame
must equal the one-shot decode of the same packet (detects races between the pipeline's worker threads, stale
descriptor tables when a job is reused for a different geometry, leaks)"""
import os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ffmpeg_ht_amd as m
import streams, bench

dec = m.Decoder()
names = [n for n in sorted(streams.CASES) if not streams.get(n)[1]]
import vecgen
# frames that qualify for 16-bit sub-bands (8-bit RGB, even geometry), several times over so that whole batches of them occur
c16 = [vecgen.encode(vecgen.synth_image(256, 192, 3, seed=40 + i), mct=1, nlevels=4) for i in range(3)]
pk = [streams.get(n)[0] for n in names] + bench.make_streams(2, 0) + c16 * 8
ref = []
dec.set_int("coef16", 0)                      # the references with int32 sub-bands, the pipeline with the default
for p in pk:
    info, planes, _, st = dec.decode(p)
    ref.append([zlib.crc32(a.tobytes()) for a in planes])
dec.set_int("coef16", 1)
rng = np.random.default_rng(7)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
order = rng.integers(0, len(pk), N)
t0 = time.time()
for batch, depth in ((3, 2), (8, 3), (5, 4)):
    pipe = dec.pipe(batch=batch, depth=depth)
    sent = got = 0
    while got < N:
        while sent < N and pipe.send(pk[order[sent]]):
            sent += 1
            if rng.random() < 0.05:
                pipe.flush()
        if sent == N:
            pipe.flush()
        r = pipe.receive()
        if r is None:
            pipe.flush()
            continue
        info, planes = r
        assert [zlib.crc32(a.tobytes()) for a in planes] == ref[order[got]], (batch, depth, got, names[order[got]] if order[got] < len(names) else "4K")
        got += 1
    pipe.close()
    print("batch %d depth %d: %d frames ok, %.1fs" % (batch, depth, N, time.time() - t0), flush=True)
import resource
print("max RSS %.0f MB" % (resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024))
