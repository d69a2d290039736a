#!/bin/bash
# ONE parametrised GPU-box script (replaces round 4's 26 one-off tools/gpu_r4_*.sh).  Every step writes
# gpurun_out/<tag>/<step>.log (+ its artefacts) and prints a two-line digest; copy what should be judged into profiles/.
#
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/gpu_session.sh r05a tests:linear bench lds_lab'
#
# steps
#   suite                  the whole `-m gpu` suite                                  (round-end evidence)
#   tests:<expr>           pytest tests -m gpu -k "<expr>"      (use + for spaces: tests:linear+and+guard)
#   file:<path>[:<expr>]   pytest <path> -m gpu [-k expr] -s
#   smoke                  __graft_entry__.smoke()
#   bench[:<flags>]        python bench.py <flags>  -> bench.json          (flags with + for spaces: bench:--no-cpu-baseline)
#   headline_trace         rocprofv3 --kernel-trace --stats -- python bench.py --headline-only -> headline_kernel_stats.csv
#   trace:<script+args>    rocprofv3 --kernel-trace --stats -- python <script args> -> <script>_kernel_stats.csv (top 30 printed)
#   pmc:<workload>         python tools/pmc_kernels.py <workload> (topk | linear | spmm): counters in their own passes
#   py:<script+args>       python <script args>
#   list_avail:<regex>     rocprofv3 --list-avail | grep -i <regex>
#   env:VAR=VALUE          export VAR=VALUE for the steps that follow
# env: STEP_TIMEOUT (seconds per step, default 900)
tag=${1:?tag}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/$tag; mkdir -p "$out"
export MMREC_TEST_OBSERVED=$PWD/$out/observed.tsv
T=${STEP_TIMEOUT:-900}
stats_digest() {   # kernel_stats.csv -> top kernels
  python3 - "$1" "${2:-30}" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.2f ms" % (tot / 1e6))
for r in rows[:int(sys.argv[2])]:
    n = re.sub(r"\(anonymous namespace\)::|^void ", "", r["Name"]).split("(")[0][:72]
    print("%6.2f%% %7d calls %10.1f us avg %10.1f us max  %s" % (float(r["Percentage"]), int(r["Calls"]), float(r["AverageNs"]) / 1e3,
                                                                float(r["MaxNs"]) / 1e3, n))
PY
}
for step in "$@"; do
  name=${step%%:*}; arg=""; [[ "$step" == *:* ]] && arg=${step#*:}; arg=${arg//+/ }
  safe=$(echo "$step" | tr -c 'A-Za-z0-9_.-' '_' | cut -c1-60); log=$out/$safe.log
  echo "=== [$tag] $step"; t0=$(date +%s)
  case $name in
    suite)   (timeout 1700 python -m pytest tests -m gpu -q -p no:cacheprovider > $log 2>&1; echo rc=$? >> $log); grep -E "passed|failed|error|rc=" $log | tail -3 ;;
    tests)   (timeout $T python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "$arg" > $log 2>&1; echo rc=$? >> $log); tail -4 $log ;;
    file)    f=${arg%%:*}; k=""; [[ "$arg" == *:* ]] && k=${arg#*:}
             (timeout $T python -m pytest $f -m gpu -q -s -p no:cacheprovider ${k:+-k "$k"} > $log 2>&1; echo rc=$? >> $log); tail -6 $log ;;
    smoke)   (timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $log 2>&1; echo rc=$? >> $log); tail -2 $log ;;
    bench)   (timeout $T python bench.py $arg > $out/bench${arg:+_$(echo $arg | tr -c 'A-Za-z0-9' '_' | cut -c1-30)}.json 2> $log; echo rc=$? >> $log); tail -2 $log | cut -c1-300
             head -c 600 $out/bench*.json | head -c 600; echo ;;
    headline_trace)
             rm -rf $out/_trace
             (timeout $T rocprofv3 --kernel-trace --stats --output-format csv -d $out/_trace -- python bench.py --headline-only > $out/bench_headline_only.json 2> $log; echo rc=$? >> $log)
             f=$(find $out/_trace -name "*kernel_stats.csv" | head -1); cp "$f" $out/headline_kernel_stats.csv; rm -rf $out/_trace
             stats_digest $out/headline_kernel_stats.csv 6; head -c 300 $out/bench_headline_only.json; echo ;;
    trace)   rm -rf $out/_trace; base=$(basename ${arg%% *} .py)
             (timeout $T rocprofv3 --kernel-trace --stats --output-format csv -d $out/_trace -- python $arg > $log 2>&1; echo rc=$? >> $log)
             f=$(find $out/_trace -name "*kernel_stats.csv" | head -1); cp "$f" $out/${base}_kernel_stats.csv
             python tools/summarize_trace.py $(find $out/_trace -name "*kernel_trace.csv" | head -1) $out/${base}_kernel_by_grid.csv
             rm -rf $out/_trace
             tail -5 $log; stats_digest $out/${base}_kernel_stats.csv 30 | tee $out/${base}_kernel_stats.txt ;;
    pmc)     (timeout $T python tools/pmc_kernels.py $arg $out/pmc_${arg%% *} > $log 2>&1; echo rc=$? >> $log); tail -5 $log ;;
    py)      (timeout $T python $arg > $log 2>&1; echo rc=$? >> $log); tail -25 $log ;;
    env)     export "$arg"; echo "exported $arg" ;;
    list_avail) (timeout 120 rocprofv3 --list-avail 2>&1 | grep -i -E "$arg" > $log; echo rc=$? >> $log); head -40 $log ;;
    *)       echo "unknown step $step" ;;
  esac
  echo "--- $step: $(( $(date +%s) - t0 )) s"
done
