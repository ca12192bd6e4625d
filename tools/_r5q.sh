cd $GRAFT_REPO_ROOT
O=gpurun_out/r5q; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_algorithms.py tests/test_gpu_golden.py tests/test_gpu_part_run.py tests/test_gpu_fullsize.py -x -q > $O/t.log 2>&1; echo "rc $?" >> $O/t.log
tail -n 3 $O/t.log
timeout 400 python bench.py --workload road_sssp --no-cpu-baseline > $O/road.log 2> $O/road.err
timeout 300 python bench.py --partitioned --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/part.log 2> $O/part.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r5q/road.log").read().strip().splitlines()[-1])
print("road", d["value"], {k:v for k,v in d.items() if "round" in k.lower()})
d=json.loads(open("gpurun_out/r5q/part.log").read().strip().splitlines()[-1])
print("part", d["value"], d["ms_per_step"])
PY
