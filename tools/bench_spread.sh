#!/bin/bash
# The driver's exact command in N fresh processes (default 10): the spread of `value`, of the blocking loop beside it and of
# the kernel's HIP-event time.  usage (on the GPU box): bash tools/bench_spread.sh [N] [outdir]
cd $GRAFT_REPO_ROOT 2>/dev/null || cd "$(dirname "$0")/.."
N=${1:-10}; O=${2:-gpurun_out/bench_spread}; mkdir -p $O
for i in $(seq 1 $N); do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-batch --no-refrule > $O/run_$i.log 2> $O/run_$i.err
done
python - $O $N <<'PY'
import json, sys, statistics as st
O, N = sys.argv[1], int(sys.argv[2])
rows = []
for i in range(1, N + 1):
    try:
        d = json.loads(open("%s/run_%d.log" % (O, i)).read().strip().splitlines()[-1])
    except Exception as e:
        print("run", i, "failed:", e); continue
    rows.append({"run": i, "value": d["value"], "ms_per_step": d["ms_per_step"], "blocking_ms_per_step": d["blocking_loop"]["ms_per_step"],
                 "kernel_ms_by_hip_events": d["roofline"]["avg_launch_ms"], "frac": d["roofline"]["frac"],
                 "per_step_wall_ms_blocking": d["blocking_loop"]["per_step_wall_ms"],
                 "coscheduled_4_ms_per_step": d.get("coscheduled", {}).get("4", {}).get("ms_per_step"),
                 "coscheduled_8_ms_per_step": d.get("coscheduled", {}).get("8", {}).get("ms_per_step"),
                 "coscheduled_12_ms_per_step": d.get("coscheduled", {}).get("12", {}).get("ms_per_step"),
                 "coscheduled_12_frac": d.get("coscheduled", {}).get("12", {}).get("roofline", {}).get("frac")})
with open(O + "/spread.jsonl", "w") as f:
    for r in rows: f.write(json.dumps(r) + "\n")
v = [r["ms_per_step"] for r in rows]; b = [r["blocking_ms_per_step"] for r in rows]; k = [r["kernel_ms_by_hip_events"] for r in rows]
print("command: python bench.py --gpus 1 --steps 20 --warmup 5 (fresh process each), %d runs" % len(rows))
print("queued   ms_per_step min %.4f median %.4f max %.4f   TEPS median %.4g" % (min(v), st.median(v), max(v), st.median([r["value"] for r in rows])))
print("blocking ms_per_step min %.4f median %.4f max %.4f" % (min(b), st.median(b), max(b)))
print("kernel by HIP events  min %.4f median %.4f max %.4f" % (min(k), st.median(k), max(k)))
for key in ("coscheduled_4_ms_per_step", "coscheduled_8_ms_per_step", "coscheduled_12_ms_per_step"):
    c = [r[key] for r in rows if r.get(key) is not None]
    if c: print("%s min %.4f median %.4f max %.4f" % (key, min(c), st.median(c), max(c)))
PY
