import os, sys, time
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
    for _ in range(3): job.run(1)
    job.wait()
    t = [job.run(1).stage_ms()[0] for _ in range(5)]
    print(f"DBG={os.environ.get('HTJ2K_DBG')} batch={batch} ht={np.mean(t):.3f} ms", flush=True)
    job.free()
