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
if [ -n "$HIPX_FINAL_LITE" ]; then echo "lite run: suite + smoke + default bench only; total ${SECONDS}s"; exit 0; fi
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
stats config4_standin_cg_sor --matrix-file standin --pc sor --steps 30 --warmup 3 --no-cpu-baseline
python scripts/cg_kernels_timing.py > $O/vector_kernels_standalone.txt 2>/dev/null
# 2 and 4 ranks on this one GPU (IPC transport; RCCL refuses shared devices): a functional run whose per-rank halo / all-reduce section times are
# lower bounds for the real thing (no xGMI hop, but four processes time-slicing one device)
for np_ in 2 4; do for pl in 1 3; do
  HIPX_ALL_RANKS_DEVICE0=1 timeout 600 python bench.py --gpus $np_ --grid 128 --quick --transport ipc --pipeline $pl --steps 200 --warmup 20 2>/dev/null | tail -1 > $O/multirank_one_gpu_np${np_}_pipeline${pl}.json
done; done
python - <<PY
import json, glob, os
out = {}
for f in sorted(glob.glob("$O/multirank_one_gpu_np*_pipeline*.json")):
    try:
        d = json.loads(open(f).read())
    except Exception as e:
        out[os.path.basename(f)] = {"error": str(e)}
        continue
    pr = (d.get("multi_gpu") or {}).get("per_rank") or d.get("per_rank") or []
    out[os.path.basename(f)[:-5]] = {"iterations_per_s": d.get("value"), "ms_per_step": d.get("ms_per_step"), "parity": (d.get("parity_gate") or {}).get("max_rel_diff"),
                                     "per_rank": [{k: r.get(k) for k in ("rank", "rows", "ghosts", "spmv_ms", "halo_ms", "allreduce_ms", "offdiag_ms", "cg_update_ms")} for r in pr]}
json.dump(out, open("$O/multirank_one_gpu.json", "w"), indent=1)
print(json.dumps(out)[:600])
PY
head -5 $O/headline_kernel_stats.csv | cut -c1-150
echo "total ${SECONDS}s"
