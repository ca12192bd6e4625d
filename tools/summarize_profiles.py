"""gpurun_out/<dir> (rocprofv3 --kernel-trace --stats run + two --pmc passes of bench.py) -> profiles/<round>/.
usage: tools/summarize_profiles.py gpurun_out/prof_r1e profiles/r01"""
import csv, collections, json, shutil, os, sys
src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
for f in os.listdir(dst):
    os.remove(os.path.join(dst, f))
rows = list(csv.reader(open(src + '/bench_kernel_stats.csv')))
shutil.copy(src + '/bench_kernel_stats.csv', dst + '/bench_kernel_stats_all.csv')
with open(dst + '/bench_kernel_stats_grb.csv', 'w', newline='') as f:
    w = csv.writer(f); w.writerow(rows[0])
    for r in rows[1:]:
        if 'grb::' in r[0]:
            w.writerow(r)
shutil.copy(src + '/bench_stdout.log', dst + '/bench_stdout_under_rocprof.log')
shutil.copy(src + '/bench_plain.log', dst + '/bench_stdout.log')
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rr = list(csv.DictReader(open(src + '/pmc_%s/p_counter_collection.csv' % c)))
    acc = collections.defaultdict(list)
    for r in rr:
        if r['Counter_Name'] != c or 'grb::' not in r['Kernel_Name']:
            continue
        acc[r['Kernel_Name'].split('(')[0].replace('void ', '')].append(float(r['Counter_Value']))
    for k, v in acc.items():
        out.setdefault(k, {})[c + "_KB_mean"] = sum(v) / len(v)
        out[k]["launches_" + c] = len(v)
    with open(dst + '/pmc_%s_per_kernel.csv' % c, 'w', newline='') as f:
        w = csv.writer(f); w.writerow(["kernel", "launches", "mean_KB", "min_KB", "max_KB"])
        for k, v in sorted(acc.items()):
            w.writerow([k, len(v), round(sum(v) / len(v), 1), min(v), max(v)])
for k, v in out.items():
    f_, w_ = v.get("FETCH_SIZE_KB_mean", 0), v.get("WRITE_SIZE_KB_mean", 0)
    v["hbm_bytes_per_launch_raw"] = int((f_ + w_) * 1024)
    v["hbm_bytes_per_launch"] = int((2 * f_ + w_) * 1024)
cal = {k: out[k] for k in out if k.startswith("grb::reduce_kernel") or "ewise_add_dd" in k or "assign_dense_mask_dense" in k or "fill_kernel" in k}
doc = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, each with --kernel-trace only) over "
                 "`python bench.py --steps 8 --warmup 2 --no-cpu-baseline` on MI355X",
       "note": "hbm_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 B.  The factor 2 on reads is MI355X_MICROARCH.md's gfx950 "
               "correction (128-B requests tallied at 64 B), calibrated in this same run on this library's own 4 B/lane kernels of "
               "known byte count (`calibration`: the 64 Mi-float reduce reads 262144 KB, eWiseAdd 524288 KB, assign 262144 KB; each is "
               "reported at exactly 1/2; WRITE_SIZE matches the written 262144 KB).  hbm_bytes_per_launch_raw is the uncorrected sum.",
       "calibration": cal, "kernels": out}
json.dump(doc, open(dst + '/pmc_traffic.json', 'w'), indent=1, sort_keys=True)
for k in ("grb::bfs_persistent_kernel", "grb::spmv_hub_kernel<1, float>", "grb::pack_vector_kernel<float>"):
    print(k, out.get(k, {}).get("hbm_bytes_per_launch"))
for k, v in cal.items():
    print("calibration", k, v.get("FETCH_SIZE_KB_mean"), v.get("WRITE_SIZE_KB_mean"))
for r in rows[1:]:
    if 'grb::' in r[0]:
        print(r[0][:70], r[1], r[3])
