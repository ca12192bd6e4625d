cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P=gpurun_out/prof_r1f; mkdir -p $P
timeout 600 python -m pytest tests -q -m gpu --timeout 300 2>&1 | tail -15 > $P/pytest_gpu.log
timeout 300 python bench.py > $P/bench_plain.log 2> $P/bench_plain.err
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -o bench -- python bench.py > $P/bench_stdout.log 2> $P/stats.err
for f in $(find $P/stats -name "bench_kernel_stats.csv"); do cp $f $P/bench_kernel_stats.csv; done
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $P/pmc_$C -o p -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline > $P/pmc_$C.log 2>&1
  for f in $(find $P/pmc_$C -name "p_counter_collection.csv"); do cp $f $P/pmc_$C/p_counter_collection.csv 2>/dev/null; done
done
rm -rf $P/stats/*/*kernel_trace* ; find $P -name "*kernel_trace.csv" -delete; find $P -name "*agent_info.csv" -delete
du -sh $P; cat $P/pytest_gpu.log | tail -5; cat $P/bench_plain.log | cut -c1-600
