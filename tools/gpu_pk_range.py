import os, sys
ROOT = "/root/repo" if os.path.isdir("/root/repo/tests") else os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ffmpeg_ht_amd as m
import vecgen
dec = m.Decoder()
W, H = 1024, 512
rng = np.random.default_rng(1)
yy, xx = np.mgrid[0:H, 0:W]
pats = {
 "uniform noise": [rng.integers(0, 256, (H, W)).astype(np.int32) for _ in range(3)],
 "colour bars":   [(((xx // 128) >> k) & 1).astype(np.int32) * 255 for k in range(3)],
 "checker 1px":   [(((xx + yy) & 1) * 255).astype(np.int32), (((xx + yy + 1) & 1) * 255).astype(np.int32), (((xx) & 1) * 255).astype(np.int32)],
 "magenta/green stripes 2px": [(((xx >> 1) & 1) * 255).astype(np.int32), ((((xx >> 1) + 1) & 1) * 255).astype(np.int32), (((xx >> 1) & 1) * 255).astype(np.int32)],
 "blocks 8px random colours": [np.kron(rng.integers(0, 2, (H // 8, W // 8)), np.ones((8, 8), int)).astype(np.int32) * 255 for _ in range(3)],
 "blocks 32px random colours": [np.kron(rng.integers(0, 2, (H // 32, W // 32)), np.ones((32, 32), int)).astype(np.int32) * 255 for _ in range(3)],
}
for name, img in pats.items():
    data = vecgen.encode(img, mct=1, nlevels=5)
    job = dec.job().parse_batch([data]).upload().run().wait()
    got = job.download_frame(0)[1][0].reshape(H, W, 3)
    print("%-28s coef16 %d ll16 %d packed %d lossless %s" % (name, job.coef16(), job.ll16(), job.idwt_packed(), np.array_equal(got, np.stack(img, -1))), flush=True)
    job.free()
