# per-kernel times of grb_tc's count path (preparation + counting kernels) under rocprofv3 --kernel-trace --stats
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r06/tc_trace
mkdir -p $out
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o t -- python tools/tc_count_bench.py 22 28 > $out/run.log 2>&1
grep -E "count:|product" $out/run.log
f=$(find $out/kt -name "t_kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    nm = r["Name"].split("(")[0].replace("void grb::", "").replace("grb::", "")
    if any(k in nm for k in ("tc_", "radix", "scan", "sort", "fillBuffer", "copyBuffer")):
        print("%-60s calls %4s total %10.1f us avg %10.1f us" % (nm[:60], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3))
PY
rm -rf $out/kt
