#!/bin/bash
# kernel-by-kernel durations of the SpMV launches of tools/spmv_format_ab.py (CSR and column-sorted format)  [GPU box]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/${2:-r3c}
mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out/kt -o b -- python tools/spmv_format_ab.py ${1:-22} 16 > $out/trace_stdout.log 2>&1
f=$(find $out/kt -name "b_kernel_trace.csv" | head -1)
python - "$f" <<'PY' > $out/spmv_kernel_trace.txt
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
agg = collections.OrderedDict()
for r in rows:
    k = r["Kernel_Name"].replace("grb::", "")
    if "spmv" in k or "pack_vector" in k:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        agg.setdefault(k[:110], []).append(d)
for k, v in agg.items():
    v2 = v[len(v) // 2:]
    print("%-112s n %3d  mean(last half) %8.1f us  min %8.1f" % (k, len(v), sum(v2) / len(v2), min(v)))
PY
cat $out/spmv_kernel_trace.txt
rm -rf $out/kt
