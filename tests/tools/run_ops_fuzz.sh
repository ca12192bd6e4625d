# usage (GPU box): [SEQS=500] [SEED=0] [FLAGS="--struconly --int"] bash tests/tools/run_ops_fuzz.sh
mkdir -p gpurun_out
timeout 500 python tests/tools/ops_fuzz.py --seqs ${SEQS:-500} --len 40 --seed ${SEED:-0} $FLAGS > gpurun_out/fuzz.log 2>&1
tail -1 gpurun_out/fuzz.log
for s in $(grep "^SEQ" gpurun_out/fuzz.log | awk '{print $2}' | head -4); do
  timeout 120 python tests/tools/ops_fuzz.py --only $s --len 40 --seed ${SEED:-0} $FLAGS > gpurun_out/fuzz_replay_$s.log 2>&1
done
