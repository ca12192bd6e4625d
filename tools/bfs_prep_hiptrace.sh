cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r06/prep_hip
mkdir -p $out
bash tools/bfs_prep_trace.sh r06/prep_x > /dev/null 2>&1   # writes /tmp/prep_one.py
timeout 300 rocprofv3 --hip-runtime-trace --output-format csv -d $out/kt -o b -- python /tmp/prep_one.py > $out/stdout.log 2>&1
f=$(find $out/kt -name "b_hip_api_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
# find the window of the first traversal: between the last hipDeviceSynchronize-ish before first bfs kernel... use the first launch of a kernel named bfs/oc; simpler: print the API calls in the 8 ms before the first hipLaunchKernel following the last big gap
calls = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"]) for r in rows]
calls.sort()
# locate the second-to-last hipDeviceSynchronize / hipStreamSynchronize cluster: take the last 400 calls
tail = calls[-400:]
t_end = tail[-1][1]
agg = {}
for s, e, f in tail:
    if t_end - s < 9_000_000:    # last 9 ms of the process's HIP activity ~ first + second traversal
        a = agg.setdefault(f, [0, 0]); a[0] += 1; a[1] += e - s
for f, (c, t) in sorted(agg.items(), key=lambda x: -x[1][1])[:14]:
    print("   %-36s calls %4d total %8.3f ms" % (f, c, t / 1e6))
PY
grep "traversal ms" $out/stdout.log
rm -rf $out/kt
