cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for k in 0 32768; do
out=gpurun_out/r06/tck$k
mkdir -p $out
GRB_TC_CORE_K=$k timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o b -- python bench.py --workload orkut_tc --steps 2 --no-cpu-baseline > $out/stdout.log 2>&1
f=$(find $out/kt -name "b_kernel_stats.csv" | head -1)
echo "== K=$k"
python - "$f" <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
for r in rows[1:]:
    if r and any(k in r[0] for k in ("spgemm", "core_", "fill_value", "sum_i32", "scan", "entry_rows")):
        name = r[0].replace("void grb::", "").split("(")[0]
        print("   %-70s calls %5s total %9.2f ms  avg %9.3f ms" % (name[:70], r[1], int(r[2]) / 1e6, int(r[2]) / 1e6 / int(r[1])))
PY
rm -rf $out/kt
done
