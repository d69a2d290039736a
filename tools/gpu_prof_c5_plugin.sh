#!/bin/bash
# kernel-trace summary of the config-5 plugin run (plain FREEDOM only: MMREC_C5_PLAIN_ONLY=1)
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp; rm -rf /tmp/prof_c5
MMREC_C5_PLAIN_ONLY=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c5 -o c -- python $R/tools/run_c5_plugin.py 40 > $R/gpurun_out/prof_c5_plugin.log 2>&1
cd $R
grep "ms/step" gpurun_out/prof_c5_plugin.log
python - <<PY
import csv
rows = list(csv.DictReader(open("$(find /tmp/prof_c5 -name '*kernel_stats.csv' | head -1)")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.1f ms" % (tot / 1e6))
for r in rows[:30]:
    print("%6.2f%% %6d calls %9.1f us avg  %s" % (float(r["Percentage"]), int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:110]))
PY
