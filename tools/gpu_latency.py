"""one 4K frame through htj2k_decode(): where the 2 ms go (wall clock per call and the library's own stage times), into
pageable and into page-locked planes, from a pageable and from a page-locked packet"""
import os, sys, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ffmpeg_ht_amd as m
import bench
streams = bench.make_streams(4, 0)
dec = m.Decoder()
info = dec.probe(streams[0])
pk = [m.packet(x) for x in streams]
for name, mk in (("pageable planes", lambda: m.alloc_frame(info)), ("page-locked planes", lambda: dec.alloc_frame_pinned(info)[0])):
    buf = mk()
    dec.decode_into(pk[0], buf)
    N = 16
    acc = np.zeros(7)
    t0 = time.perf_counter()
    for i in range(N):
        r, st = dec.decode_into(pk[i % 4], buf)
        acc += [st.ms_parse, st.ms_h2d, st.ms_kernels, st.ms_d2h, st.ms_ht, st.ms_idwt, st.ms_pack]
    dt = (time.perf_counter() - t0) / N * 1e3
    a = acc / N
    print("%-20s %.3f ms per call = %.0f Mpixel/s | parse(+staging) %.3f  h2d %.3f  kernels %.3f (ht %.3f idwt %.3f)  download(wait + d2h) %.3f" % (
        name, dt, info.width * info.height / dt / 1e3, a[0], a[1], a[2], a[4], a[5], a[3]), flush=True)
