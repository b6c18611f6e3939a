#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp HIPX_NO_TORCH=1
B=oracle/_ref/mpich/bin; P=petsc_amd/lib/libpetschipx_mpich.so; M=/opt/conda/bin/mpiexec
A="-m 5 -n 5 -ksp_gmres_cgs_refinement_type refine_always"
{
echo "== np2 vec only gdb"; timeout 100 $M -n 2 /opt/rocm/bin/rocgdb -batch -ex run -ex bt --args $B/ex2 $A -dll_prepend $P -vec_type hipx 2>&1 | grep -v "New Thread\|Thread.*exited\|libthread_db" | head -150
echo "== np1 vec only"; timeout 60 $M -n 1 $B/ex2 $A -dll_prepend $P -vec_type hipx 2>&1 | tail -3
echo "== np2 ex31"; timeout 60 $M -n 2 $B/kat_vec_ex31 -dll_prepend $P -vec_type hipx 2>&1 | tail -3
for i in 1 2 3; do echo "== np2 full gdb $i"; timeout 100 $M -n 2 /opt/rocm/bin/rocgdb -batch -ex run -ex bt --args $B/ex2 $A -dll_prepend $P -vec_type hipx -mat_type aijhipx 2>&1 | grep -v "New Thread\|Thread.*exited\|libthread_db" | tail -40; done
} > gpurun_out/mpi_dbg.log 2>&1
tail -250 gpurun_out/mpi_dbg.log
