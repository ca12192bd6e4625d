cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_algorithms.py -x -q -k "bfs" > $O/t_algo.log 2>&1; echo "algo rc $?" >> $O/t_algo.log
for v in default nolw wt; do
  if [ $v = default ]; then unset GRB_HIP_LIB; else export GRB_HIP_LIB=build/libgrb_hip_$v.so; fi
  timeout 300 python tools/bfs_ab.py > $O/ab_$v.log 2>&1
done
unset GRB_HIP_LIB
timeout 300 python tools/bfs_trace.py 22 2000702,2887554 > $O/trace.log 2>&1
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-batch > $O/bench.log 2> $O/bench.err
tail -n 3 $O/t_algo.log; grep -h -A2 "^LIB" $O/ab_*.log; grep "^bfs trace" $O/trace.log | tail -n 2; cut -c1-300 $O/bench.log
