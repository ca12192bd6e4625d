#!/bin/bash
# kernel-by-kernel durations of the SpMV launches of tools/spmv_band_probe.py (first run only)  [GPU box]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/spk
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/spk/kt -o b -- python tools/spmv_band_probe.py ${1:-22} first > gpurun_out/spk/out.log 2>&1
f=$(find gpurun_out/spk/kt -name "b_kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
agg = collections.OrderedDict()
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("grb::", "")
    if "spmv" in k or "pack_vector" in k or "band_" in k:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        agg.setdefault(k, []).append(d)
for k, v in agg.items():
    v2 = v[len(v) // 2:]
    print("%-60s n %3d  mean(last half) %8.1f us  min %8.1f" % (k[:60], len(v), sum(v2) / len(v2), min(v)))
PY
rm -rf gpurun_out/spk/kt
