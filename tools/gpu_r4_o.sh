#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/slice_trace -- python tools/_slice_trace.py > /tmp/st.log 2>&1
f=$(find /tmp/slice_trace -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY' > gpurun_out/r04_slice_kernel_stats.txt
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "spmm_narrow" in r["Name"]:
        print("%-80s calls %4s avg %9.1f us" % (r["Name"].replace("(anonymous namespace)::","")[:80], r["Calls"], float(r["AverageNs"])/1e3))
PY
cat gpurun_out/r04_slice_kernel_stats.txt
