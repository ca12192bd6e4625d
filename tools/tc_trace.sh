#!/bin/bash
# per-kernel durations of `bench.py --workload orkut_tc` (masked SpGEMM kernels)  [GPU box]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/${1:-r3g}
mkdir -p $out
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o b -- python bench.py --workload orkut_tc --steps 2 --no-cpu-baseline > $out/tc_trace_stdout.log 2>&1
f=$(find $out/kt -name "b_kernel_stats.csv" | head -1)
grep -E "spgemm|entry_rows|sum_i32|fill_value|Name" $f | cut -c1-260 > $out/tc_kernel_stats.txt
cat $out/tc_kernel_stats.txt
grep -E "^\{" $out/tc_trace_stdout.log | cut -c1-600
rm -rf $out/kt
