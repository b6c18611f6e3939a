#!/bin/bash
# idle gaps (> 300 us between consecutive kernels) in a kernel trace of the drop-in's two identical solves (stock KSPCG, 400 iterations, -resolve): where the first solve
# loses the time the second does not.  -> gpurun_out/<tag>/first_solve_gaps.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-r06}
O=$R/gpurun_out/$T
mkdir -p $O
export HIPX_NO_TORCH=1 MKL_NUM_THREADS=1 OMP_NUM_THREADS=1 TMPDIR=/tmp
A="-stencil 7 -n 256 -pc_type jacobi -ksp_rtol 1e-50 -ksp_norm_type preconditioned -dll_prepend $R/petsc_amd/lib/libpetschipx.so -vec_type hipx -mat_type aijhipx -ksp_type cg -ksp_max_it 400 -resolve"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof_fg -o t -- $R/oracle/_ref/bin/ref_driver $A > $O/fg.out 2>&1)
python3 - "$O" > $O/first_solve_gaps.txt <<'PY'
import csv, glob, sys, re
O = sys.argv[1]
f = glob.glob(O + "/prof_fg/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
prev_end, prev_name = None, None
def short(n): return re.sub(r"\(anonymous namespace\)::|^void ", "", n)[:60]
print("kernels:", len(rows))
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if prev_end is not None and s - prev_end > 300000:
        print("gap %8.2f ms at %9.2f ms   after %-60s before %s" % ((s - prev_end) / 1e6, (prev_end - t0) / 1e6, short(prev_name), short(r["Kernel_Name"])))
    prev_end, prev_name = max(e, prev_end or 0), r["Kernel_Name"]
print("total span %.2f ms" % ((prev_end - t0) / 1e6))
# the iteration's kernels, first solve against second: launches split in halves by time order
from collections import defaultdict
by = defaultdict(list)
for r in rows:
    by[short(r["Kernel_Name"])].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
for k, v in sorted(by.items(), key=lambda kv: -sum(d for _, d in kv[1]))[:8]:
    if len(v) < 100:
        continue
    h = len(v) // 2
    a, b = v[:h], v[h:]
    gap = lambda L: sum(L[i + 1][0] - (L[i][0] + L[i][1]) for i in range(len(L) - 1)) / max(len(L) - 1, 1)
    print("%-62s n %4d  first half avg %8.2f us  second half avg %8.2f us   first 20: %8.2f us" % (k, len(v), sum(d for _, d in a) / len(a) / 1e3, sum(d for _, d in b) / len(b) / 1e3, sum(d for _, d in v[:20]) / 20e3))
# the busy fraction of each solve: from the first to the last launch of the dominant kernel in each half
dom = max(by.items(), key=lambda kv: sum(d for _, d in kv[1]))[1]
h = len(dom) // 2
for name, part in (("first solve", dom[:h]), ("second solve", dom[h:])):
    lo, hi = part[0][0], part[-1][0] + part[-1][1]
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if lo <= int(r["Start_Timestamp"]) <= hi)
    print("%s: span %.2f ms, kernels busy %.2f ms (%.1f %%)" % (name, (hi - lo) / 1e6, busy / 1e6, 100.0 * busy / (hi - lo)))
PY
rm -rf $O/prof_fg
cat $O/first_solve_gaps.txt; cat $O/fg.out | grep -E "iterations|second"
