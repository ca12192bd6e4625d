#!/bin/bash
# PMC counters of the band SpMV kernel (valued and pattern): waves waiting, instruction mix, L2 requests   [GPU box]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r06/pmc_spmv
mkdir -p $out
for iso in "" 1; do
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  ISO=$iso timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/$tag -o p -- python tools/spmv_cband_quick.py > $out/$tag.log 2>&1
  f=$(find $out/$tag -name "p_counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" "iso=$iso" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    if "spmv_cband_kernel" in k:
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(sys.argv[2], k.replace("void grb::", ""), {c: "%.4g (x%d)" % (sum(v) / len(v), len(v)) for c, v in d.items()})
PY
  rm -rf $out/$tag
done
done
