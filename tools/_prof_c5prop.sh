cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "$@"; do
  mkdir -p gpurun_out/prof_$v
  TFW_ONLY=c5prop MMREC_HIP_LIB=$GRAFT_REPO_ROOT/tools/probe_libs/libmmrec_tfw_$v.so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$v -- python tools/prof_topk_variants.py one > gpurun_out/prof_$v/run.log 2>&1
  f=$(find gpurun_out/prof_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"; grep "filter_" $f | sed -e 's/(anonymous namespace):://g' -e 's/^"void //' -e 's/^"//' | python3 -c "
import sys,csv
for line in sys.stdin:
    name=line.split('(')[0][:48]; rest=line.rsplit('\",',1)[-1] if '\",' in line else ''
    parts=line.strip().split(',')
    print('%-50s calls %s avg_us %.1f' % (name, parts[-7], float(parts[-5])/1e3))
"
done
