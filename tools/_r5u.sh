cd $GRAFT_REPO_ROOT
O=gpurun_out/r5u; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_spmm.py -x -q > $O/t.log 2>&1; echo "rc $?" >> $O/t.log
tail -n 2 $O/t.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.log 2> $O/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5u/bench.log").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["bfs_batch64"]["ms_per_sweep"], d["bfs_batch64"].get("frac"))
PY
