python -m pytest tests/test_gpu_batch.py -x -q 2>&1 | tail -2
for cfg in "4096 4096 0.25" "1024 1024 0.25" "512 512 0.25" "256 256 0.25" "512 4096 0.25" "4096 512 0.25" "512 512 0" "256 256 0" "512 512 0.05"; do
set -- $cfg
echo "== big_in $1 big_out $2 budget $3"
GRB_BATCH_BIG_IN=$1 GRB_BATCH_BIG_OUT=$2 GRB_BATCH_BUDGET=$3 GRB_BATCH_TRACE=1 python tools/batch_bench.py 22 0 2>&1 | tail -10 | grep -v "labels\|level [5678]"
done
