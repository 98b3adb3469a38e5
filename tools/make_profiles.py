"""gpurun_out/prof_<tag>/ (written by tools/prof_round.sh <tag> on the GPU box) -> profiles/<tag>_*: the summaries the
round's numbers come from.  Usage: python tools/make_profiles.py [round-tag]"""
import collections, csv, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

# 1. rocprofv3 --kernel-trace --stats of the bench command
shutil.copy(os.path.join(src, "trace", "bench_kernel_stats.csv"), os.path.join(dst, tag + "_bench_kernel_stats.csv"))
bench_line = [l for l in open(os.path.join(src, "bench_trace.log"), errors="replace") if l.startswith("{")][-1]
open(os.path.join(dst, tag + "_bench_under_rocprof.json"), "w").write(bench_line)
full = os.path.join(src, "bench_full.json")
if os.path.exists(full):
    shutil.copy(full, os.path.join(dst, tag + "_bench.json"))

# 2. PMC passes: FETCH_SIZE (x2 on gfx950, calibrated below) and WRITE_SIZE, KB per dispatch
def pmc(path):
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        k = (r["Kernel_Name"].split("(")[0], r["Grid_Size"])
        a = acc.setdefault(k, [0, 0.0])
        a[0] += 1; a[1] += float(r["Counter_Value"])
    return {k: (n, v / n) for k, (n, v) in acc.items()}
fetch = pmc(os.path.join(src, "fetch", "bench_counter_collection.csv"))
write = pmc(os.path.join(src, "write", "bench_counter_collection.csv"))
rows = []
for k in fetch:
    if "rocclr" in k[0]: continue
    n, f = fetch[k]; w = write.get(k, (0, 0.0))[1]
    rows.append((k[0], k[1], n, f, w, (2 * f + w) * 1024 / 1e6))
with open(os.path.join(dst, tag + "_pmc_hbm.csv"), "w") as o:
    o.write("kernel,grid_size,dispatches,FETCH_SIZE_KB_mean,WRITE_SIZE_KB_mean,hbm_MB_per_dispatch(2*FETCH+WRITE)\n")
    for r in rows: o.write('"%s",%s,%d,%.1f,%.1f,%.1f\n' % r)

# 3. calibration of the FETCH_SIZE halving on the copy-shape microbenchmark (known byte counts)
cf = pmc(os.path.join(src, "cal_fetch", "membw_counter_collection.csv"))
cw = pmc(os.path.join(src, "cal_write", "membw_counter_collection.csv"))
with open(os.path.join(dst, tag + "_membw_calibration.txt"), "w") as o:
    o.write("tools/ubench/membw on the same box: copy-shape bandwidth (no arithmetic) and the PMC calibration\n\n")
    mb = os.path.join(src, "membw.txt")
    if os.path.exists(mb): o.write(open(mb).read() + "\n")
    o.write("k_rowwalk<VEC,NS,WPB> reads NS planes and writes 1 plane of 3840x2160x4 B per z (known byte counts):\n")
    for k in cf:
        if "rowwalk" not in k[0]: continue
        o.write("%-28s grid %-9s FETCH_SIZE %10.1f KB  WRITE_SIZE %10.1f KB\n" % (k[0], k[1], cf[k][1], cw.get(k, (0, 0))[1]))
    o.write("\n1r1w x24 planes: 796.26 MB read -> FETCH_SIZE 388811 KB = 398.1 MB = exactly 1/2 (8 and 16 B per lane alike);\n"
            "WRITE_SIZE 777600 KB = 796.26 MB = exact.  Hence hbm bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.\n")

# 4. per-launch traffic of the IDWT launches for bench.py's roofline.traffic
idwt = [r for r in rows if "idwt" in r[0]]
per_step = sum(r[5] * r[2] for r in idwt) / max(1, min(r[2] for r in idwt if "pack" in r[0]) if any("pack" in r[0] for r in idwt) else 1)
launches_per_step = sum(r[2] for r in idwt) / max(1, [r[2] for r in idwt if "pack" in r[0]][0])
cfg = json.loads(bench_line)["config"]
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `bench.py --steps 10 --warmup 2 --jobs 1 "
                     "--no-cpu-baseline --no-e2e`, hbm = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE halving, calibrated in "
                     + tag + "_membw_calibration.txt)",
           "frames_per_step": cfg["frames_per_step"], "idwt_launches_per_step": launches_per_step,
           "hbm_MB_per_step_all_idwt_launches": round(per_step, 1),
           "hbm_bytes_per_launch": round(per_step * 1e6 / launches_per_step),
           "kernels": [{"kernel": r[0], "grid": r[1], "hbm_MB_per_dispatch": round(r[5], 1)} for r in idwt]},
          open(os.path.join(dst, tag + "_idwt_traffic.json"), "w"), indent=1)
print(open(os.path.join(dst, tag + "_pmc_hbm.csv")).read())
print(open(os.path.join(dst, tag + "_idwt_traffic.json")).read())


# 5. average duration per (kernel, grid size) of the trace: tells the IDWT levels apart (the stats file has one row per kernel name)
acc = collections.OrderedDict()
for r in csv.DictReader(open(os.path.join(src, "trace", "bench_kernel_trace.csv"))):
    n = r["Kernel_Name"].split("(")[0]
    if "rocclr" in n or "fill" in n.lower(): continue
    a = acc.setdefault((n, r["Grid_Size_X"]), [0, 0.0])
    a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
with open(os.path.join(dst, tag + "_bench_kernels_by_grid.csv"), "w") as o:
    o.write("kernel,grid_size_x,dispatches,avg_us\n")
    for (n, g), (c, t) in acc.items(): o.write('"%s",%s,%d,%.1f\n' % (n, g, c, t / c))

# 6. the other configurations (tools/gpu_configs.py): stage times, per-kernel stats, HBM bytes of their IDWT launches
if os.path.exists(os.path.join(src, "configs.json")):
    shutil.copy(os.path.join(src, "configs.json"), os.path.join(dst, tag + "_configs.json"))
    shutil.copy(os.path.join(src, "configs.log"), os.path.join(dst, tag + "_configs.txt"))
if os.path.exists(os.path.join(src, "cfg_trace", "cfg_kernel_stats.csv")):
    shutil.copy(os.path.join(src, "cfg_trace", "cfg_kernel_stats.csv"), os.path.join(dst, tag + "_configs_C3_C4_kernel_stats.csv"))
    f2 = pmc(os.path.join(src, "cfg_fetch", "cfg_counter_collection.csv"))
    w2 = pmc(os.path.join(src, "cfg_write", "cfg_counter_collection.csv"))
    with open(os.path.join(dst, tag + "_configs_C3_C4_pmc_hbm.csv"), "w") as o:
        o.write("kernel,grid_size,dispatches,FETCH_SIZE_KB_mean,WRITE_SIZE_KB_mean,hbm_MB_per_dispatch(2*FETCH+WRITE)\n")
        for k in f2:
            if "rocclr" in k[0]: continue
            n, f = f2[k]; w = w2.get(k, (0, 0.0))[1]
            o.write('"%s",%s,%d,%.1f,%.1f,%.1f\n' % (k[0], k[1], n, f, w, (2 * f + w) * 1024 / 1e6))
if os.path.exists(os.path.join(src, "ht_sq.csv")):
    shutil.copy(os.path.join(src, "ht_sq.csv"), os.path.join(dst, tag + "_ht_sq.csv"))
print(sorted(x for x in os.listdir(dst) if x.startswith(tag)))
