(timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "topk" 2>&1 | tail -2)
for rnd in 1 2; do for v in current prepool; do
  echo -n "$v "; TFW_ONLY=c5prop MMREC_HIP_LIB=$GRAFT_REPO_ROOT/tools/probe_libs/libmmrec_tfw_$v.so timeout 200 python tools/prof_topk_variants.py one 2>&1 | tail -1 | cut -c1-120
done; done
