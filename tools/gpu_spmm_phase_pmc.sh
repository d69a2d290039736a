#!/bin/bash
# L2 hit rate / fabric bytes of the c5 SpMM per phase (user rows, item rows) -- tools/spmm_lab.py mask 0 only, under rocprofv3
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp
for set in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $set | cut -c1-8 | tr ' ' '_'); rm -rf /tmp/pmc_$tag
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_$tag -o pm -- python $R/tools/spmm_lab.py run 2 > $R/gpurun_out/spmm_phase_pmc_$tag.log 2>&1
  f=$(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "spmm_rows_kernel" in k:
        agg[r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X", "?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    print("grid", k, {c: (sum(v) / len(v), len(v)) for c, v in d.items()})
PY
done
