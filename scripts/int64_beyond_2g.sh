#!/bin/bash
# One operator beyond 2^31 nonzeros THROUGH THE DROP-IN on one GPU: 27-pt 432^3 (80.6 M rows, 2.17e9 nonzeros) assembled by the reference's
# MatSetValues in a 64-bit-PetscInt libpetsc (oracle/_ref/int64), MATSEQAIJHIPX / VECSEQHIPX, reference KSPSolve_CG + PCJACOBI, 30 iterations.
# The same run on the CPU types is the check (history to 1e-10).  Usage: bash scripts/int64_beyond_2g.sh [n=432]
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
N=${1:-432}
A="-stencil 27 -n $N -ksp_type cg -pc_type jacobi -ksp_rtol 1e-50 -ksp_max_it 30 -ksp_norm_type preconditioned -history"
export MKL_NUM_THREADS=1 HIPX_NO_TORCH=1
( time oracle/_ref/int64/bin/ref_driver $A -dll_prepend $PWD/petsc_amd/lib/libpetschipx_int64.so -vec_type hipx -mat_type aijhipx ) > gpurun_out/int64_gpu.log 2>&1 &
( time oracle/_ref/int64/bin/ref_driver $A -mat_type aij -vec_type standard ) > gpurun_out/int64_cpu.log 2>&1 &
wait
python - <<'PY'
import numpy as np
h = {}
for k in ("gpu", "cpu"):
    t = open("gpurun_out/int64_%s.log" % k).read()
    h[k] = np.array([float(l.split()[2]) for l in t.splitlines() if l.startswith("hist ")])
    print(k, [l for l in t.splitlines() if l.startswith("iterations") or l.startswith("real")])
if len(h["gpu"]) == len(h["cpu"]) > 0:
    print("entries %d  max relative difference GPU vs CPU (MKL reductions) %.3e" % (len(h["cpu"]), float((np.abs(h["gpu"] - h["cpu"]) / h["cpu"]).max())))
else:
    print("history lengths", len(h["gpu"]), len(h["cpu"]))
PY
