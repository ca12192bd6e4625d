#!/bin/bash
# SQ counters of the masked-SpGEMM kernels of `bench.py --workload orkut_tc`  [GPU box]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/${1:-r3g}
mkdir -p $out
pass=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_FLAT" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  pass=$((pass+1))
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $out/pmc$pass -o p -- python bench.py --workload orkut_tc --steps 1 --warmup 0 --no-cpu-baseline > $out/tc_pmc$pass.log 2>&1
  f=$(find $out/pmc$pass -name "p_counter_collection.csv" | head -1)
  python - "$f" <<'PY' >> $out/tc_pmc.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "spgemm" in k:
        acc[k.split("(")[0][-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in d.items():
        print("   %-24s %.4g (mean of %d launches)" % (c, sum(v) / len(v), len(v)))
PY
  rm -rf $out/pmc$pass
done
cat $out/tc_pmc.txt
tail -3 $out/tc_pmc1.log | cut -c1-300
