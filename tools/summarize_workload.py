"""One of BASELINE.json's other configurations taken alone
(PROFILE_SKIP_DEFAULT=1 PROFILE_WORKLOADS="orkut_tc" bash tools/profile_round.sh <dir>) -> its files under profiles/<round>/:
<W>_kernel_stats_grb.csv, pmc_<W>_{FETCH,WRITE}_SIZE_per_kernel.csv, its entry in pmc_traffic.json and its line in
other_workloads.jsonl.  Nothing else in the directory is touched.
usage: tools/summarize_workload.py gpurun_out/<dir> profiles/r04 orkut_tc"""
import collections, csv, json, os, sys
src, dst, W = sys.argv[1], sys.argv[2], sys.argv[3]
KEY = {"lj_bfs": "soc-L", "road_sssp": "road", "orkut_tc": "rkut"}[W]
rr = list(csv.reader(open(src + '/%s_kernel_stats.csv' % W)))
with open(dst + '/%s_kernel_stats_grb.csv' % W, 'w', newline='') as fo:
    w = csv.writer(fo); w.writerow(rr[0])
    for r in rr[1:]:
        if 'grb::' in r[0]:
            w.writerow(r)
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = src + '/pmc_%s_%s/p_counter_collection.csv' % (W, c)
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == c and 'grb::' in r['Kernel_Name']:
            acc[r['Kernel_Name'].split('(')[0].replace('void ', '')].append(float(r['Counter_Value']))
    for k, v in acc.items():
        out.setdefault(k, {})[c + "_KB_mean"] = sum(v) / len(v)
        out[k]["launches_" + c] = len(v)
    with open(dst + '/pmc_%s_%s_per_kernel.csv' % (W, c), 'w', newline='') as fo:
        w = csv.writer(fo); w.writerow(["kernel", "launches", "mean_KB", "min_KB", "max_KB"])
        for k, v in sorted(acc.items()):
            w.writerow([k, len(v), round(sum(v) / len(v), 1), min(v), max(v)])
for k, v in out.items():
    f_, w_ = v.get("FETCH_SIZE_KB_mean", 0), v.get("WRITE_SIZE_KB_mean", 0)
    v["hbm_bytes_per_launch_raw"] = int((f_ + w_) * 1024)
    v["hbm_bytes_per_launch"] = int((2 * f_ + w_) * 1024)
entry = {"command": "python bench.py --workload %s --no-cpu-baseline" % W, "kernels": out}
if W == "orkut_tc":
    calls = [v for k, v in out.items() if "fill_value_kernel" in k]
    ncall = min(calls[0].get("launches_FETCH_SIZE", 0), calls[0].get("launches_WRITE_SIZE", 0)) if calls else 0
    if ncall > 0:
        tot = sum(1024.0 * (2 * v.get("FETCH_SIZE_KB_mean", 0) * v.get("launches_FETCH_SIZE", 0)
                            + v.get("WRITE_SIZE_KB_mean", 0) * v.get("launches_WRITE_SIZE", 0))
                  for k, v in out.items() if "spgemm_" in k or "fill_value_kernel" in k)
        entry["groups"] = {"masked_spgemm_call": {"hbm_bytes_per_unit": int(tot / ncall), "units": ncall,
                                                  "what": "every spgemm_* kernel + fill_value_kernel, per mxm call"}}
doc = json.load(open(dst + '/pmc_traffic.json'))
doc.setdefault("workloads", {})[W] = entry
json.dump(doc, open(dst + '/pmc_traffic.json', 'w'), indent=1, sort_keys=True)
new_line = open(src + '/%s.log' % W).read().strip().splitlines()[-1]
lines = open(dst + '/other_workloads.jsonl').read().strip().splitlines()
lines = [new_line if KEY in json.loads(ln).get("metric", "") + json.dumps(json.loads(ln).get("config", {})) else ln for ln in lines]
open(dst + '/other_workloads.jsonl', 'w').write("\n".join(lines) + "\n")
print(W, json.loads(new_line)["value"], entry.get("groups"))
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:6]:
    print("  ", k[:90], v["hbm_bytes_per_launch"])
