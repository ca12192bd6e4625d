cd $GRAFT_REPO_ROOT
for v in default coal0 wt default coal0; do
  if [ $v = default ]; then unset GRB_HIP_LIB; else export GRB_HIP_LIB=build/libgrb_hip_$v.so; fi
  timeout 300 python tools/bfs_ab.py 2>&1 | grep "^LIB" | cut -c1-140
done
unset GRB_HIP_LIB
timeout 600 python -m pytest tests/test_gpu_algorithms.py -m gpu -x -q -k bfs 2>&1 | tail -2
