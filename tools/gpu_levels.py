import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ffmpeg_ht_amd as m
import vecgen
dec = m.Decoder()
img = vecgen.synth_image(3840, 2160, 3, seed=2)
data = vecgen.encode(img, mct=1)
for batch in (1, 8):
    job = dec.job().parse_batch([data] * batch).upload()
    for _ in range(3): job.run(7)
    job.wait()
    acc = None
    for _ in range(5):
        job.run(7).wait()
        l = job.idwt_launches()
        acc = [(a[0] + b[0], b[1]) for a, b in zip(acc, l)] if acc else l
    st = job.stage_ms()
    print(f"batch {batch}: stages ht={st[0]:.3f} idwt={st[1]:.3f} pack={st[2]:.3f}")
    tot_ms = sum(a[0] for a in acc) / 5; tot_b = sum(a[1] for a in acc)
    for i, (ms, by) in enumerate(acc):
        print(f"   level {i}: {ms/5*1e3:8.1f} us  {by/1e6:8.1f} MB  {by/(ms/5)/1e6:7.0f} GB/s")
    print(f"   sum of launches {tot_ms*1e3:.1f} us -> {tot_b/tot_ms/1e6:.0f} GB/s", flush=True)
    job.free()
