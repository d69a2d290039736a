#!/bin/bash
# word-list variant of the top-K filter: parity tests + the config-5 evaluation companion of bench.py under rocprofv3
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_config_shapes_gpu.py -m gpu -x -q -k "topk or eval or score" > gpurun_out/topk_sparse_tests.log 2>&1
grep -n "passed\|failed\|Error" gpurun_out/topk_sparse_tests.log | tail -5
cd /tmp; rm -rf /tmp/prof_tk
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tk -o c -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pmc > $R/gpurun_out/topk_sparse_bench.json 2> $R/gpurun_out/topk_sparse_bench.err
cd $R
python - <<PY
import csv, json
rows = list(csv.DictReader(open("$(find /tmp/prof_tk -name '*kernel_stats.csv' | head -1)")))
for r in rows:
    if "filter" in r["Name"] or "topk" in r["Name"]:
        print("%6d calls %9.1f us avg  %s" % (int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:100]))
d = json.loads(open("gpurun_out/topk_sparse_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "c5_full_eval", d["extra"]["c5_full_eval"], "fwd_bwd", d["extra"].get("c5_propagate_fwd_bwd"))
print({k: v for k, v in d["extra"].items() if k.startswith("baby_linear") or k.startswith("baby_score")})
PY
