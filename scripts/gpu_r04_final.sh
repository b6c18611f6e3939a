#!/bin/bash
# round-4 final run: the whole GPU suite, smoke, the default bench line (counter passes kept), rocprofv3 kernel stats of the headline leg and of the
# 27-pt 512^3 / config-5 legs, stand-alone vector-kernel timings.  Usage: bash scripts/gpu_r04_final.sh [tag]
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
T=${1:-r04z}
O=$GRAFT_REPO_ROOT/gpurun_out/$T
mkdir -p $O
SECONDS=0
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -rf > $O/pytest.log 2>&1
echo "pytest exit $? after ${SECONDS}s" >> $O/pytest.log
grep -E "passed|failed" $O/pytest.log | tail -2
grep -E "^FAILED|^ERROR" $O/pytest.log | head -20
cp gpurun_out/parity_measured.json $O/parity_measured.json 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
echo "smoke exit $?" >> $O/smoke.log
tail -2 $O/smoke.log
S0=$SECONDS
HIPX_BENCH_KEEP_PROFILES=$O/pmc timeout 1500 python bench.py > $O/bench_default.json 2>$O/bench_default.err
echo "default bench: rc $? $((SECONDS - S0)) s" | tee -a $O/bench_default.err
tail -1 $O/bench_default.json | cut -c1-260
stats() { # name, bench args...
  local name=$1; shift
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o s -- python $GRAFT_REPO_ROOT/bench.py "$@" > $O/prof_$name.json 2>/dev/null)
  f=$(find $O/prof_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $O/${name}_kernel_stats.csv && rm -rf $O/prof_$name
}
stats headline --no-traffic --no-plugin --no-cpu-baseline --no-other --no-general
stats 27pt_512 --quick --stencil 27 --grid 512 --steps 30 --warmup 3
stats config5_share --quick --grid 1024 --scaling weak --pc none --steps 50 --warmup 5
stats gmres_sor_27pt_256 --quick --ksp gmres --pc sor --stencil 27 --grid 256 --steps 60 --warmup 5
python scripts/cg_kernels_timing.py > $O/vector_kernels_standalone.txt 2>/dev/null
head -5 $O/headline_kernel_stats.csv | cut -c1-150
echo "total ${SECONDS}s"
