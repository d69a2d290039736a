#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for set in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
  rm -rf /tmp/slice_pmc
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/slice_pmc -- python tools/_slice_trace.py > /tmp/sp.log 2>&1
  f=$(find /tmp/slice_pmc -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY' >> gpurun_out/r04_slice_pmc.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
per = collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    if "spmm_narrow" in r["Kernel_Name"]:
        per[(r["Kernel_Name"].replace("(anonymous namespace)::","")[:44], r["Grid_Size"], r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
for (k, g, d, c), v in per.items():
    acc[(k, g)][c].append(v)
for (k, g), cs in sorted(acc.items()):
    print("%-46s grid %9s  " % (k, g) + "  ".join("%s %.0f" % (c, sum(v) / len(v)) for c, v in cs.items()))
PY
done
cat gpurun_out/r04_slice_pmc.txt
