cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5aa; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_spmv_cband.py tests/test_gpu_spmv_bands.py tests/test_gpu_ops.py tests/test_gpu_algorithms.py -x -q -k "not bfs" > $O/t.log 2>&1; echo "rc $?" >> $O/t.log
tail -n 3 $O/t.log
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o b -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-batch --no-lanes --no-refrule > $O/bench.log 2> $O/bench.err
f=$(find $O/prof -name "b_kernel_stats.csv" | head -1); grep -E "cband_pack|cband_fold|spmv_cband_kernel" $f | cut -c1-60,120-220
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5aa/bench.log").read().strip().splitlines()[-1])
for k in ("spmv","spmv_valued"): print(k, d[k]["avg_launch_ms"], d[k]["frac"])
PY
rm -rf $O/prof
