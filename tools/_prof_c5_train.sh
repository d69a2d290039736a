cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_c5_train
MMREC_C5_PLAIN_ONLY=1 MMREC_C5_LATE_STEPS=${LATE:-600} timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c5_train -- python tools/run_c5_plugin.py 30 > gpurun_out/prof_c5_train/run.log 2>&1
f=$(find gpurun_out/prof_c5_train -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys,re
rows=list(csv.reader(open(sys.argv[1])))
for r in rows[1:]:
    n=re.sub(r"\(anonymous namespace\)::","",r[0]); n=re.sub(r"^void ","",n); n=n.split("(")[0][:70]
    if float(r[2])/1e6 < 1.0 or "gemm_nt" in n or "select_topk" in n: continue
    print("%-60s calls %6s total_ms %9.2f avg_us %9.1f max_us %9.1f" % (n, r[1], float(r[2])/1e6, float(r[3])/1e3, float(r[6])/1e3))
PY
grep "\[c5\]" gpurun_out/prof_c5_train/run.log | cut -c1-160
