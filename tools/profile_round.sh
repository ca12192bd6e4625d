# The rocprofv3 passes behind profiles/rNN (run on the GPU box through gpurun; outputs under gpurun_out/).
#   usage: bash tools/profile_round.sh prof_r3
# 1. plain `python bench.py`                       -> bench_plain.log (the JSON line)
# 2. the same command under --kernel-trace --stats -> per-kernel statistics + the kernel trace (kept: the
#    default command also times the reference's direction rule and the 64-source sweep AFTER the main
#    measurement; tools/summarize_profiles.py splits bfs_persistent_kernel's launches by order)
# 3. two separate --pmc passes (FETCH_SIZE / WRITE_SIZE need different counter slots), each with
#    --kernel-trace only, over `bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-refrule` (the 64-source
#    sweep and the sparse x dense product included, so their kernels get counters too)
# 4. BASELINE.json's other configurations: the plain JSON line, a --kernel-trace --stats run and the same two --pmc
#    passes each (`--no-cpu-baseline`); PROFILE_WORKLOADS="" skips them (their kernels unchanged since the last take),
#    PROFILE_SKIP_DEFAULT=1 PROFILE_WORKLOADS="orkut_tc" takes one of them alone
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P=gpurun_out/${1:-prof_r3}; mkdir -p $P
grb_trace() {   # $1: a rocprofv3 kernel trace csv -> $2: this library's kernels in launch order
  python - "$1" "$2" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "grb::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "start_ns", "duration_ns"])
    t0 = int(rows[0]["Start_Timestamp"]) if rows else 0
    for r in rows:
        w.writerow([r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Start_Timestamp"]) - t0,
                    int(r["End_Timestamp"]) - int(r["Start_Timestamp"])])
PY
}
if [ -z "$PROFILE_SKIP_DEFAULT" ]; then
timeout 300 python bench.py > $P/bench_plain.log 2> $P/bench_plain.err
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -o bench -- python bench.py > $P/bench_stdout.log 2> $P/stats.err
for f in $(find $P/stats -name "bench_kernel_stats.csv"); do cp $f $P/bench_kernel_stats.csv; done
for f in $(find $P/stats -name "bench_kernel_trace.csv"); do grb_trace "$f" "$P/bench_kernel_trace_grb.csv"; done
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 500 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $P/pmc_$C -o p -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-refrule > $P/pmc_$C.log 2>&1
  for f in $(find $P/pmc_$C -name "p_counter_collection.csv"); do cp $f $P/pmc_$C/p_counter_collection.csv 2>/dev/null; done
done
fi
for W in ${PROFILE_WORKLOADS-lj_bfs road_sssp orkut_tc}; do
  timeout 400 python bench.py --workload $W > $P/$W.log 2> $P/$W.err
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats_$W -o w -- python bench.py --workload $W --no-cpu-baseline > $P/${W}_under_rocprof.log 2> $P/stats_$W.err
  for f in $(find $P/stats_$W -name "w_kernel_stats.csv"); do cp $f $P/${W}_kernel_stats.csv; done
  rm -rf $P/stats_$W
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $P/pmc_${W}_$C -o p -- python bench.py --workload $W --no-cpu-baseline > $P/pmc_${W}_$C.log 2>&1
    for f in $(find $P/pmc_${W}_$C -name "p_counter_collection.csv"); do cp $f $P/pmc_${W}_$C/p_counter_collection.csv 2>/dev/null; done
  done
done
rm -rf $P/stats; find $P -name "*kernel_trace.csv" ! -name "bench_kernel_trace_grb.csv" -delete; find $P -name "*agent_info.csv" -delete
find $P -type d -empty -delete
du -sh $P; [ -f $P/bench_plain.log ] && cut -c1-300 $P/bench_plain.log; for W in ${PROFILE_WORKLOADS-lj_bfs road_sssp orkut_tc}; do cut -c1-200 $P/$W.log; done
