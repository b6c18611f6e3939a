#!/bin/bash
# SQ counter passes over the SpMV / AXPY kernels of `bench.py --spmv-only` (the dominant kernel of the headline configuration):
# where the waves' cycles go (parked on s_waitcnt / barriers, issue stalls, active by instruction class), instruction counts.
# Usage: bash scripts/pmc_sq.sh <tag> [variant] ; results: gpurun_out/<tag>_sq_<pass>.csv (+ a summary table on stdout)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
T=${1:-sq}
V=${2:-0}
R=$(pwd)
rocprofv3 -L > gpurun_out/${T}_counters_list.txt 2>&1
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM"
P2="SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS"
P3="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_INSTS_SENDMSG"
P4="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"
P5="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS"  # (a pass of its own: if a name is unknown only this pass fails)
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  i=$((i + 1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d "$R/gpurun_out/${T}_sq$i" -o pmc -- python "$R/bench.py" --spmv-only 6 --variant $V > "$R/gpurun_out/${T}_sq$i.log" 2>&1)
  echo "pass $i exit $?"
done
python - "$T" <<'PY'
import csv, glob, sys, collections
T = sys.argv[1]
tab = collections.defaultdict(lambda: collections.defaultdict(list))
for i in (1, 2, 3, 4, 5):
    for f in glob.glob("gpurun_out/%s_sq%d/**/*counter_collection.csv" % (T, i), recursive=True):
        per = collections.defaultdict(float)
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            name = "spmv" if "spmv_" in k else ("axpy" if "ew2_kernel" in k else None)
            if name:
                per[(name, row["Counter_Name"], row["Dispatch_Id"])] += float(row["Counter_Value"])
        for (name, c, d), v in per.items():
            tab[name][c].append(v)
for name in tab:
    print("== %s (mean per launch over %d launches)" % (name, max(len(v) for v in tab[name].values())))
    for c in sorted(tab[name]):
        v = tab[name][c]
        print("  %-34s %16.1f" % (c, sum(v) / len(v)))
PY
find gpurun_out/${T}_sq* -name "*kernel_trace.csv" -delete 2>/dev/null
