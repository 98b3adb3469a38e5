"""Soak: damaged Part-1 / MIXED code bytes through the GPU path and the oracle; Part-1 results must be identical
(the MQ decoder is total), frame-level errors must agree.  usage: python tools/gpu_fuzz_part1.py [iterations]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ffmpeg_ht_amd as m
import oracle, streams

n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dec = m.Decoder()
orc = oracle.OracleDecoder()
names = sorted(n for n in streams.CASES if n.startswith("p1_") and not streams.get(n)[1])
rng = np.random.default_rng(20261004)
stat = {"compared": 0, "frame_errors": 0, "mismatch": 0, "block_errors": 0}
for it in range(n_iter):
    name = names[int(rng.integers(0, len(names)))]
    data = bytearray(streams.get(name)[0])
    kind = int(rng.integers(0, 4))
    lo = min(120, len(data) - 3)
    for pos in rng.integers(lo, len(data) - 2, int(rng.integers(1, 24))):
        if kind == 0: data[pos] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1: data[pos] = 0xFF
        elif kind == 2: data[pos] = int(rng.integers(0x90, 0x100))
        else: data[pos] = int(rng.integers(0, 256))
    if int(rng.integers(0, 8)) == 0:
        data = data[:int(rng.integers(lo, len(data)))]
    data = bytes(data)
    try:
        info_o, planes_o, _ = orc.decode(data)
        err_o = 0
    except oracle.DecodeError as e:
        err_o = e.code
    try:
        info, planes, _, st = dec.decode(data)
        err = 0
    except m.Htj2kError as e:
        err = e.code
    if err or err_o:
        stat["frame_errors"] += 1
        if err != err_o:
            stat["mismatch"] += 1
            print("ERROR CODE MISMATCH", name, it, err, err_o, flush=True)
        continue
    stat["compared"] += 1
    stat["block_errors"] += orc.block_errors()
    if st.n_block_errors != orc.block_errors() or not all(np.array_equal(a, b) for a, b in zip(planes, planes_o)):
        stat["mismatch"] += 1
        print("MISMATCH", name, it, kind, st.n_block_errors, orc.block_errors(), flush=True)
    if it % 50 == 49:
        print(it + 1, stat, flush=True)
print("done", stat, flush=True)
sys.exit(1 if stat["mismatch"] else 0)
