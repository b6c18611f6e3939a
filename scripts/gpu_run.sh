#!/bin/bash
# The ONE script for work on the GPU box (replaces the per-run scripts of rounds 3 and 4; what each of those ran is recorded in
# profiles/README.md).  Usage, from the build container:
#
#   gpurun --timeout 900 -- 'bash scripts/gpu_run.sh <tag> <step> [<step> ...]'
#
# Every step writes under gpurun_out/<tag>/ (merged back by gpurun); a step is one of
#   smoke                      __graft_entry__.smoke()
#   tests[:<pytest args>]      python -m pytest -m gpu (default: the whole suite); e.g. 'tests:tests/test_gpu_sor.py -k strand'
#   bench[:<bench args>]       python bench.py <args> (default: the driver's own command, --gpus 1 --steps 20 --warmup 5); the line, the
#                              detail file and the per-kernel counter CSVs are kept
#   stats:<name>:<bench args>  rocprofv3 --kernel-trace --stats of `bench.py <args>` -> <name>_kernel_stats.csv
#   pmc:<name>:<counters>:<bench args>   one rocprofv3 --pmc pass (counters separated by commas) with --kernel-trace only
#   py:<script> [args]         python <script> [args] (stdout/stderr -> <script name>.txt)
#   sh:<command>               bash -c '<command>' (stdout/stderr -> sh_<n>.txt)
# A step that exceeds its own timeout is killed (no hung box); the wall time of every step is appended to gpurun_out/<tag>/steps.txt.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
T=${1:-run}
shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$T
mkdir -p "$O"
n=0
for step in "$@"; do
  n=$((n + 1))
  kind=${step%%:*}
  rest=""
  [ "$kind" != "$step" ] && rest=${step#*:}
  S0=$SECONDS
  case $kind in
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$O/smoke.txt" 2>&1
      rc=$?
      tail -1 "$O/smoke.txt"
      ;;
    tests)
      # shellcheck disable=SC2086
      timeout ${HIPX_TESTS_TIMEOUT:-1500} python -m pytest ${rest:-tests} -m gpu -q --timeout 900 -p no:cacheprovider -rf > "$O/pytest_$n.txt" 2>&1
      rc=$?
      grep -E "passed|failed|error" "$O/pytest_$n.txt" | tail -2
      grep -E "^FAILED|^ERROR" "$O/pytest_$n.txt" | head -20
      [ -f gpurun_out/parity_measured.json ] && cp gpurun_out/parity_measured.json "$O/parity_measured.json"
      ;;
    bench)
      # shellcheck disable=SC2086
      HIPX_BENCH_KEEP_PROFILES=$O/pmc timeout ${HIPX_BENCH_TIMEOUT:-900} python bench.py ${rest:---gpus 1 --steps 20 --warmup 5} > "$O/bench_$n.out" 2> "$O/bench_$n.err"
      rc=$?
      tail -1 "$O/bench_$n.out" > "$O/bench_$n.json"
      cp bench_detail.json "$O/bench_${n}_detail.json" 2>/dev/null
      echo "bench line: $(wc -c < "$O/bench_$n.json") bytes"
      cut -c1-700 "$O/bench_$n.json"
      ;;
    stats)
      name=${rest%%:*}
      args=${rest#*:}
      # shellcheck disable=SC2086
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_$name" -o s -- python "$R/bench.py" $args > "$O/stats_$name.out" 2>&1)
      rc=$?
      f=$(find "$O/prof_$name" -name "*kernel_stats.csv" | head -1)
      [ -n "$f" ] && cp "$f" "$O/${name}_kernel_stats.csv" && head -6 "$f" | cut -c1-160
      rm -rf "$O/prof_$name"
      ;;
    pmc)
      name=${rest%%:*}
      r2=${rest#*:}
      ctrs=${r2%%:*}
      args=${r2#*:}
      # shellcheck disable=SC2086
      (cd /tmp && timeout 900 rocprofv3 --pmc ${ctrs//,/ } --kernel-trace --output-format csv -d "$O/pmc_$name" -o pmc -- python "$R/bench.py" $args > "$O/pmc_$name.out" 2>&1)
      rc=$?
      find "$O/pmc_$name" -name "*kernel_trace.csv" -delete 2>/dev/null
      f=$(find "$O/pmc_$name" -name "*counter_collection.csv" | head -1)
      [ -n "$f" ] && python scripts/pmc_summary.py --table "$f" > "$O/pmc_$name.txt" 2>&1 && head -30 "$O/pmc_$name.txt" | cut -c1-200
      ;;
    py)
      # shellcheck disable=SC2086
      timeout ${HIPX_PY_TIMEOUT:-900} python $rest > "$O/$(basename "${rest%% *}" .py)_$n.txt" 2>&1
      rc=$?
      tail -${HIPX_PY_TAIL:-25} "$O/$(basename "${rest%% *}" .py)_$n.txt" | cut -c1-240
      ;;
    sh)
      timeout ${HIPX_SH_TIMEOUT:-900} bash -c "$rest" > "$O/sh_$n.txt" 2>&1
      rc=$?
      tail -25 "$O/sh_$n.txt" | cut -c1-240
      ;;
    *)
      echo "unknown step: $step"
      rc=64
      ;;
  esac
  echo "step $n [$step] rc $rc $((SECONDS - S0)) s" | tee -a "$O/steps.txt"
done
echo "total ${SECONDS}s"
