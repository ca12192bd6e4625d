# The rocprofv3 passes behind profiles/rNN (run on the GPU box through gpurun; outputs under gpurun_out/).
#   usage: bash tools/profile_round.sh prof_r2
# 1. plain `python bench.py`                      -> bench_plain.log (the JSON line)
# 2. the same command under --kernel-trace --stats -> per-kernel statistics + the kernel trace (kept: the
#    default command also times the reference's direction rule and the 64-source sweep AFTER the main
#    measurement; tools/summarize_profiles.py splits bfs_persistent_kernel's launches by order)
# 3. two separate --pmc passes (FETCH_SIZE / WRITE_SIZE need different counter slots), each with
#    --kernel-trace only, over the main measurement alone (--no-refrule --no-batch)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P=gpurun_out/${1:-prof_r2}; mkdir -p $P
timeout 300 python bench.py > $P/bench_plain.log 2> $P/bench_plain.err
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -o bench -- python bench.py > $P/bench_stdout.log 2> $P/stats.err
for f in $(find $P/stats -name "bench_kernel_stats.csv"); do cp $f $P/bench_kernel_stats.csv; done
for f in $(find $P/stats -name "bench_kernel_trace.csv"); do
  python - "$f" "$P/bench_kernel_trace_grb.csv" <<'PY'
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
done
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $P/pmc_$C -o p -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-refrule --no-batch > $P/pmc_$C.log 2>&1
  for f in $(find $P/pmc_$C -name "p_counter_collection.csv"); do cp $f $P/pmc_$C/p_counter_collection.csv 2>/dev/null; done
done
rm -rf $P/stats; find $P -name "*kernel_trace.csv" -delete; find $P -name "*agent_info.csv" -delete
du -sh $P; cut -c1-400 $P/bench_plain.log
