"""corrupt codeblock bodies (headers intact) through the GPU path: the decoder must return (frames with zeroed or
garbage blocks, or an error code) and never fault; afterwards a clean stream must still decode bit-exactly"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ffmpeg_ht_amd as m
import streams

dec = m.Decoder()
rng = np.random.default_rng(11)
names = ["gray_l5_cb64", "rgb_mct", "gray_3passes", "rgb_3passes_cb32", "gray_97_q2", "placeholder_2_3p", "noise_max", "gray_l3_cb256x16", "gray_3passes_vsc"]
import vecgen
# frames that take the 16-bit sub-band path (k_ht_decode_pair, 16-bit IDWT loads, k_ht_vlc<true>)
extra = {"c16_256x192": vecgen.encode(vecgen.synth_image(256, 192, 3, seed=5, noise=10), mct=1, nlevels=4),
         "c16_512x256_cb32": vecgen.encode(vecgen.synth_image(512, 256, 3, seed=6, noise=30), mct=1, nlevels=5, cb=(5, 5))}
n_ok = n_err = 0
t0 = time.time()
for name in names + sorted(extra):
    data, kw = (extra[name], {}) if name in extra else streams.get(name)
    clean = dec.decode(data)[1]
    start = data.index(b"\xff\x93") + 2 if b"\xff\x93" in data else len(data) // 4      # after SOD
    for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 150):
        b = bytearray(data)
        mode = it % 4
        k = [1, 8, 64, 400][mode]
        for _ in range(k):
            pos = int(rng.integers(start, len(b) - 2))
            if mode == 0: b[pos] ^= 1 << int(rng.integers(0, 8))
            elif mode == 3: b[pos] = 0xFF
            else: b[pos] = int(rng.integers(0, 256))
        try:
            info, planes, _, st = dec.decode(bytes(b))
            n_ok += 1
        except m.Htj2kError:
            n_err += 1
    again = dec.decode(data)[1]
    assert all(np.array_equal(a, c) for a, c in zip(again, clean)), name
print("fuzzed bodies: %d decoded, %d rejected, %.1fs; clean streams still bit-exact" % (n_ok, n_err, time.time() - t0))
