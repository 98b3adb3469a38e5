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
n_ok = n_err = 0
t0 = time.time()
for name in names:
    data, kw = streams.get(name)
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
