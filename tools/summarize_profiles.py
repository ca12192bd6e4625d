"""gpurun_out/<dir> (tools/profile_round.sh: rocprofv3 --kernel-trace --stats run + two --pmc passes of bench.py)
-> profiles/<round>/.      usage: tools/summarize_profiles.py gpurun_out/prof_r2 profiles/r02"""
import csv, collections, json, shutil, os, sys
src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
ALL_W = ("lj_bfs", "road_sssp", "orkut_tc")
# a partial take (PROFILE_WORKLOADS="" or a subset): what belongs to a workload that was not run again stays as it is
kept_w = [W for W in ALL_W if not os.path.exists(src + '/%s.log' % W)]
prev_doc, prev_lines = {}, {}
if kept_w and os.path.exists(dst + '/pmc_traffic.json'):
    prev_doc = json.load(open(dst + '/pmc_traffic.json')).get("workloads", {})
if kept_w and os.path.exists(dst + '/other_workloads.jsonl'):
    for ln in open(dst + '/other_workloads.jsonl').read().strip().splitlines():
        for W, key in (("lj_bfs", "soc-L"), ("road_sssp", "road"), ("orkut_tc", "rkut")):
            if key in json.loads(ln).get("metric", "") + json.dumps(json.loads(ln).get("config", {})):
                prev_lines.setdefault(W, ln)
keep_prefixes = ("full_suite", "smoke", "batch", "bfs_", "tests_", "sssp_", "tc_", "spmv_", "bench_spread", "orkut_tc_kernel_stats_before", "core_", "co", "prep", "driver_", "README") + tuple(kept_w) + \
    tuple("pmc_%s_" % W for W in kept_w)
for f in os.listdir(dst):
    if not f.startswith(keep_prefixes):     # logs of test / A-B runs kept beside the profiles
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
# ---- bfs_persistent_kernel by phase of the default command.  Launch order in bench.py: W warm-up, K timed, K with HIP
# events (the roofline pass), then accounting / reference-rule / spot-check launches.  The whole-run mean in the stats
# file mixes them (the reference-rule launches run 2x longer by design); the timed region is launches [W, W + K).
trace = src + '/bench_kernel_trace_grb.csv'
if os.path.exists(trace):
    shutil.copy(trace, dst + '/bench_kernel_trace_grb.csv')
    line = json.loads(open(src + '/bench_stdout.log').read().strip().splitlines()[-1])
    W, K = line["warmup"], line["steps"]
    if "bfs_prep" in line:                   # since round 4 the bench times a matrix's FIRST traversal before the warm-up
        W += 1
    tr = [r for r in csv.DictReader(open(trace))]
    # (since round 6 the kernel is a template on the workgroup's width: <1024> one traversal per launch, <512> / <256> /
    # <128> the launches of several traversals)
    bfs = [int(r["duration_ns"]) for r in tr if r["kernel"].startswith("grb::bfs_persistent_kernel<1024>")]
    co = {}
    for r in tr:
        for wdt in ("512", "256", "128"):
            if r["kernel"].startswith("grb::bfs_persistent_kernel<%s>" % wdt):
                co.setdefault(wdt, []).append(int(r["duration_ns"]))
    spmv = {}
    for r in tr:
        for key in ("spmv_cband_kernel", "cband_pack_kernel", "spmv_cband_fold_kernel", "spmv_hub_kernel", "pack_vector_kernel",
                    "spmv_long_finalize_kernel"):
            if key in r["kernel"]:
                spmv.setdefault(key, []).append(int(r["duration_ns"]))
    # since round 5: the K timed (queued) steps are followed by min(warmup, 2) blocking warm-ups and the K steps of the
    # blocking loop; the HIP-event pass comes after those
    E = W + K + (min(line["warmup"], 2) + K if "blocking_loop" in line else 0)
    if "lanes" in line:                      # the two-lane sibling: max(warmup, 2) warm-ups + K steps, then 2 + K on one lane again
        E += max(line["warmup"], 2) + K + 2 + K
    if "coscheduled" in line:                # ... its launches are other kernels, but K one-at-a-time steps follow it
        E += K
    ph = {"source": "rocprofv3 --kernel-trace of the default `python bench.py` (same run as bench_kernel_stats_*.csv)",
          "warmup": W, "steps": K, "bfs_persistent_kernel_launches": len(bfs),
          "bfs_persistent_kernel_mean_us": {
              "timed_region_launches_W_to_W+K": round(sum(bfs[W:W + K]) / K / 1e3, 2),
              "hip_event_pass_launches": round(sum(bfs[E:E + K]) / K / 1e3, 2),
              "hip_event_pass_first_launch_index": E,
              "all_launches_of_the_run": round(sum(bfs) / len(bfs) / 1e3, 2)},
          "coscheduled_launches": {("bfs_persistent_kernel<%s>" % k_): {"launches": len(v_), "durations_us": [round(x / 1e3, 1) for x in v_]}
                                   for k_, v_ in co.items()},
          "coscheduled_hip_event_ms_reported_by_bench": {k_: line["coscheduled"][k_]["launch_ms_total_by_hip_events"]
                                                         for k_ in ("4", "8", "12") if k_ in line["coscheduled"]} if "coscheduled" in line else None,
          "hip_event_mean_us_reported_by_bench": round(line["roofline"]["avg_launch_ms"] * 1e3, 2),
          "spmv_kernels_mean_us": {k: round(sum(v) / len(v) / 1e3, 2) for k, v in spmv.items()},
          "spmv_hip_event_mean_us_reported_by_bench": round(line["spmv"]["avg_launch_ms"] * 1e3, 2)}
    json.dump(ph, open(dst + '/bench_kernel_phases.json', 'w'), indent=1)
    print(json.dumps(ph["bfs_persistent_kernel_mean_us"]), ph["hip_event_mean_us_reported_by_bench"], ph["spmv_kernels_mean_us"])
def pmc_tables(prefix, csv_prefix):
    """per-kernel means of the two passes under src/<prefix>FETCH_SIZE, src/<prefix>WRITE_SIZE"""
    out = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = src + '/%s%s/p_counter_collection.csv' % (prefix, c)
        if not os.path.exists(f):
            continue
        rr = list(csv.DictReader(open(f)))
        acc = collections.defaultdict(list)
        for r in rr:
            if r['Counter_Name'] != c or 'grb::' not in r['Kernel_Name']:
                continue
            acc[r['Kernel_Name'].split('(')[0].replace('void ', '')].append(float(r['Counter_Value']))
        for k, v in acc.items():
            out.setdefault(k, {})[c + "_KB_mean"] = sum(v) / len(v)
            out[k]["launches_" + c] = len(v)
        with open(dst + '/%s%s_per_kernel.csv' % (csv_prefix, c), 'w', newline='') as fo:
            w = csv.writer(fo); w.writerow(["kernel", "launches", "mean_KB", "min_KB", "max_KB"])
            for k, v in sorted(acc.items()):
                w.writerow([k, len(v), round(sum(v) / len(v), 1), min(v), max(v)])
    for k, v in out.items():
        f_, w_ = v.get("FETCH_SIZE_KB_mean", 0), v.get("WRITE_SIZE_KB_mean", 0)
        v["hbm_bytes_per_launch_raw"] = int((f_ + w_) * 1024)
        v["hbm_bytes_per_launch"] = int((2 * f_ + w_) * 1024)
    # the launches of several traversals carry different numbers of them (warm-up, timed, profiled passes): per TRAVERSAL,
    # from the command's own step / warm-up counts (bench.py: max(warmup, 2 k) warm-up + 3 x steps timed + steps profiled)
    plog = src + '/%sFETCH_SIZE.log' % prefix
    if os.path.exists(plog):
        try:
            pl = json.loads([x for x in open(plog).read().strip().splitlines() if x.startswith("{")][-1])
            for wdt, kks in (("256", (4,)), ("128", (8, 12))):
                name = "grb::bfs_persistent_kernel<%s>" % wdt
                if name in out and "coscheduled" in pl:
                    trav = sum(max(pl["warmup"], 2 * kk) + 4 * pl["steps"] for kk in kks if str(kk) in pl["coscheduled"])
                    v = out[name]
                    tot = 1024.0 * (2 * v.get("FETCH_SIZE_KB_mean", 0) * v.get("launches_FETCH_SIZE", 0) +
                                    v.get("WRITE_SIZE_KB_mean", 0) * v.get("launches_WRITE_SIZE", 0))
                    v["traversals_in_these_launches"] = trav
                    v["hbm_bytes_per_traversal"] = int(tot / trav)
        except Exception as exc:                # noqa: BLE001
            print("co-scheduled traffic not derived:", exc)
    return out


out = pmc_tables("pmc_", "pmc_")
cal = {k: out[k] for k in out if k.startswith("grb::reduce_kernel") or "ewise_add_dd" in k or "assign_dense_mask_dense" in k or "fill_kernel" in k}
# a unit made of several kernels: the 64-source sweep = every batch_* launch between two batch_seed_kernel launches
groups = {}
seeds = [v for k, v in out.items() if "batch_seed_kernel" in k]
if seeds:
    sweeps = min(seeds[0].get("launches_FETCH_SIZE", 0), seeds[0].get("launches_WRITE_SIZE", 0))
    if sweeps > 0:
        tot = 0.0
        for k, v in out.items():
            if "grb::batch_" in k:
                tot += 1024.0 * (2 * v.get("FETCH_SIZE_KB_mean", 0) * v.get("launches_FETCH_SIZE", 0)
                                 + v.get("WRITE_SIZE_KB_mean", 0) * v.get("launches_WRITE_SIZE", 0))
        groups["bfs_batch_sweep"] = {"hbm_bytes_per_unit": int(tot / sweeps), "units": sweeps,
                                     "what": "sum over every batch_* kernel of (2 x FETCH_SIZE + WRITE_SIZE) x launches, per batch_seed_kernel launch"}
workloads = {}
lines = []
for W in ALL_W:
    if W in kept_w:
        if W in prev_lines:
            lines.append(prev_lines[W])
        if W in prev_doc:
            workloads[W] = prev_doc[W]
        continue
    if os.path.exists(src + '/%s.log' % W):
        txt = open(src + '/%s.log' % W).read().strip().splitlines()
        if txt:
            lines.append(txt[-1])
    if os.path.exists(src + '/%s_kernel_stats.csv' % W):
        rr = list(csv.reader(open(src + '/%s_kernel_stats.csv' % W)))
        with open(dst + '/%s_kernel_stats_grb.csv' % W, 'w', newline='') as fo:
            w = csv.writer(fo); w.writerow(rr[0])
            for r in rr[1:]:
                if 'grb::' in r[0]:
                    w.writerow(r)
    ow = pmc_tables("pmc_%s_" % W, "pmc_%s_" % W)
    if ow:
        workloads[W] = {"command": "python bench.py --workload %s --no-cpu-baseline" % W, "kernels": ow}
        if W == "orkut_tc":
            # one masked SpGEMM = one fill_value_kernel launch + the pivot kernels of both passes
            calls = [v for k, v in ow.items() if "fill_value_kernel" in k]
            ncall = min(calls[0].get("launches_FETCH_SIZE", 0), calls[0].get("launches_WRITE_SIZE", 0)) if calls else 0
            if ncall > 0:
                tot = sum(1024.0 * (2 * v.get("FETCH_SIZE_KB_mean", 0) * v.get("launches_FETCH_SIZE", 0)
                                    + v.get("WRITE_SIZE_KB_mean", 0) * v.get("launches_WRITE_SIZE", 0))
                          for k, v in ow.items() if "spgemm_" in k or "fill_value_kernel" in k)
                workloads[W]["groups"] = {"masked_spgemm_call": {"hbm_bytes_per_unit": int(tot / ncall), "units": ncall,
                                                                 "what": "every spgemm_* kernel + fill_value_kernel, per mxm call"}}
            # one count without the product = one tc_total_kernel launch + the counting kernels (the preparation's
            # kernels run once per matrix and are not in the sum)
            calls = [v for k, v in ow.items() if "tc_total_kernel" in k]
            ncall = min(calls[0].get("launches_FETCH_SIZE", 0), calls[0].get("launches_WRITE_SIZE", 0)) if calls else 0
            if ncall > 0:
                tot = sum(1024.0 * (2 * v.get("FETCH_SIZE_KB_mean", 0) * v.get("launches_FETCH_SIZE", 0)
                                    + v.get("WRITE_SIZE_KB_mean", 0) * v.get("launches_WRITE_SIZE", 0))
                          for k, v in ow.items() if "tc_count_" in k or "tc_total_kernel" in k)
                workloads[W].setdefault("groups", {})["tc_count_call"] = {
                    "hbm_bytes_per_unit": int(tot / ncall), "units": ncall,
                    "what": "tc_count_bitmap_kernel + tc_count_small_kernel (+ tc_count_pivot_kernel) + tc_total_kernel, per grb_tc call on a prepared matrix"}
if lines:
    open(dst + '/other_workloads.jsonl', 'w').write("\n".join(lines) + "\n")
doc = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, each with --kernel-trace only) over "
                 "`python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-refrule` on MI355X",
       "note": "hbm_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 B.  The factor 2 on reads is MI355X_MICROARCH.md's gfx950 "
               "correction (128-B requests tallied at 64 B), calibrated in this same run on this library's own 4 B/lane kernels of "
               "known byte count (`calibration`: the 64 Mi-float reduce reads 262144 KB, eWiseAdd 524288 KB, assign 262144 KB; each is "
               "reported at exactly 1/2; WRITE_SIZE matches the written 262144 KB).  hbm_bytes_per_launch_raw is the uncorrected sum.",
       "calibration": cal, "kernels": out, "groups": groups, "workloads": workloads}
json.dump(doc, open(dst + '/pmc_traffic.json', 'w'), indent=1, sort_keys=True)
for k in ("grb::bfs_persistent_kernel<1024>", "grb::bfs_persistent_kernel<256>", "grb::bfs_persistent_kernel<128>", "grb::spmv_cband_kernel<1, float, false>", "grb::spmv_cband_kernel<1, float, true>", "grb::cband_pack_kernel<float>"):
    print(k, out.get(k, {}).get("hbm_bytes_per_launch"))
print("groups", groups)
for W, d in workloads.items():
    for k, v in d["kernels"].items():
        if v["hbm_bytes_per_launch"] > (1 << 26):
            print(W, k[:80], v["hbm_bytes_per_launch"])
for k, v in cal.items():
    print("calibration", k, v.get("FETCH_SIZE_KB_mean"), v.get("WRITE_SIZE_KB_mean"))
for r in rows[1:]:
    if 'grb::' in r[0]:
        print(r[0][:70], r[1], r[3])
