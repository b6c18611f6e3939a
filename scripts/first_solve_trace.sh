#!/bin/bash
# what the drop-in's FIRST KSPSolve does that the second does not: kernel + memory-copy trace of `ref_driver ... -ksp_max_it 400` (one solve), summary of copies and of the
# kernels that are not part of the iteration.  -> gpurun_out/<tag>/first_solve_trace.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-r06}
O=$R/gpurun_out/$T
mkdir -p $O
export HIPX_NO_TORCH=1 MKL_NUM_THREADS=1 OMP_NUM_THREADS=1 TMPDIR=/tmp
A="-stencil 7 -n 256 -pc_type jacobi -ksp_rtol 1e-50 -ksp_norm_type preconditioned -dll_prepend $R/petsc_amd/lib/libpetschipx.so -vec_type hipx -mat_type aijhipx -ksp_type cg -ksp_max_it 400"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $O/prof_fs -o s -- $R/oracle/_ref/bin/ref_driver $A > $O/fs.out 2>&1)
for f in $(find $O/prof_fs -name "*stats.csv"); do echo "== $(basename $f)"; head -14 $f | cut -c1-170; done > $O/first_solve_trace.txt
python3 - "$O" >> $O/first_solve_trace.txt <<'PY'
import csv, glob, sys
O = sys.argv[1]
for f in glob.glob(O + "/prof_fs/**/*memory_copy_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    print("== memory copies:", len(rows))
    t0 = min(int(r["Start_Timestamp"]) for r in rows) if rows else 0
    for r in rows:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if d > 200:
            print("  %10.1f us  at %10.1f ms  %s" % (d, (int(r["Start_Timestamp"]) - t0) / 1e6, r.get("Direction") or r.get("Name")))
PY
rm -rf $O/prof_fs
cat $O/first_solve_trace.txt
