cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2k
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r2k/kt -o b -- python tools/batch_bench.py 22 0 > gpurun_out/r2k/out.log 2>&1
f=$(find gpurun_out/r2k/kt -name "b_kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]         # every kernel (memsets and copies show up as fill / copy kernels)
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last sweep: find the last batch_seed_kernel
idx = max(i for i, r in enumerate(rows) if "batch_seed" in r["Kernel_Name"])
t0 = int(rows[idx]["Start_Timestamp"])
for r in rows[idx:]:
    print("%-34s start %8.1f us  dur %8.1f us" % (r["Kernel_Name"].split("(")[0].replace("grb::", ""), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
rm -rf gpurun_out/r2k/kt
