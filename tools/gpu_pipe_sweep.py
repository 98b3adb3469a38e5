"""packets in host memory -> device-resident frames through htj2k_pipe: rate against batch size, depth and parse threads
usage: python tools/gpu_pipe_sweep.py "batch,depth[,parse_threads]" ..."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ffmpeg_ht_amd as m
import bench
streams = bench.make_streams(4, 0)
dec = m.Decoder()
pk = [m.packet(x) for x in streams]
W, H = 3840, 2160
def run(batch, depth, nfr=256):
    pipe = dec.pipe(batch=batch, depth=depth)
    nwarm = batch * (2 * depth - 1) + batch
    sent = got = 0; t0 = None
    while got < nwarm + nfr:
        while sent < nwarm + nfr and pipe.send(pk[sent % len(pk)]): sent += 1
        if sent == nwarm + nfr: pipe.flush()
        if pipe.receive_device() is None: break
        got += 1
        if got == nwarm: t0 = time.perf_counter()
    r = (got - nwarm) * W * H / (time.perf_counter() - t0) / 1e9
    pipe.close()
    return r
for spec in sys.argv[1:] or ["8,3"]:
    v = [int(x) for x in spec.split(",")]
    dec.set_int("parse_threads", v[2] if len(v) > 2 else 0)
    print("batch %2d depth %d parse_threads %2d: %.2f Gpixel/s" % (v[0], v[1], v[2] if len(v) > 2 else 0, run(v[0], v[1])), flush=True)
