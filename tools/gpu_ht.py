"""A/B of the HT decode modes on the GPU: parity vs oracle + timing."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ffmpeg_ht_amd as m
import oracle, vecgen, streams

dec = m.Decoder()
orc = oracle.OracleDecoder()
bad = 0
for mode in (1, 0):
    dec.set_int("ht_mode", mode)
    for name in sorted(streams.CASES):
        data, kw = streams.get(name)
        dec.set_int("bitexact", kw.get("bitexact", 0)); dec.set_int("reduction_factor", kw.get("reduction_factor", 0))
        info_o, planes_o, n_o = orc.decode(data, **kw)
        info, planes, n, st = dec.decode(data)
        ok = all(np.array_equal(a, b) for a, b in zip(planes, planes_o))
        if not ok:
            bad += 1
            print("MISMATCH", mode, name, flush=True)
    dec.set_int("bitexact", 0); dec.set_int("reduction_factor", 0)
print("parity mismatches:", bad, flush=True)

nb = int(os.environ.get("BATCH", "8"))
imgs = [vecgen.synth_image(3840, 2160, 3, seed=2 + i) for i in range(2)]
datas = [vecgen.encode(im, mct=1) for im in imgs]
for mode in (0, 1):
    dec.set_int("ht_mode", mode)
    for batch in (1, nb):
        job = dec.job().parse_batch([datas[i % 2] for i in range(batch)]).upload()
        for _ in range(3):
            job.run(7)
        job.wait()
        t = [job.run(7).stage_ms() for _ in range(5)]
        ht = np.mean([x[0] for x in t]); idwt = np.mean([x[1] for x in t]); pk = np.mean([x[2] for x in t])
        info, planes = job.download_frame(batch - 1)
        ok = np.array_equal(planes[0].reshape(2160, 3840, 3), np.stack(imgs[(batch - 1) % 2], -1))
        print(f"ht_mode={mode} batch={batch}: ht={ht:.3f} idwt={idwt:.3f} pack={pk:.3f} ms  lossless={ok} errs={job.block_errors()}", flush=True)
        job.free()
