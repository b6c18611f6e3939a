#!/bin/bash
# A/B library for the plane-march PCSOR probes: libhipx.so with hipx_sorbox.hip of an earlier commit (default f42296b = the round-5 kernel as it stood when round 6's
# PCSOR work began), every other object as built now, into ab/old/ (HIPX_LIBDIR=$PWD/ab/old selects it: scripts/r06_sorbox_hop.sh, scripts/r06_sorbox_g.sh).
#   bash scripts/ab_build_old.sh [commit]
cd "$(dirname "$0")/.." || exit 1
REV=${1:-f42296b}
python -c "from petsc_amd import build; build.build_hipx(); build.build_host()" || exit 1
mkdir -p ab/old /tmp/ab_old
git show "$REV":petsc_amd/csrc/hipx_sorbox.hip > /tmp/ab_old/hipx_sorbox.hip || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -DHIPX_BUILD -Iinclude -Ipetsc_amd/csrc -c /tmp/ab_old/hipx_sorbox.hip -o /tmp/ab_old/hipx_sorbox.o || exit 1
O=petsc_amd/lib/obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/old/libhipx.so $O/hipx_runtime.o $O/hipx_vec.o $O/hipx_pipe.o $O/hipx_mat.o $O/hipx_sell.o $O/hipx_sor.o /tmp/ab_old/hipx_sorbox.o $O/hipx_comm.o \
  -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib -Wl,-z,nodelete || exit 1
cp petsc_amd/lib/libhipxksp.so ab/old/
echo "ab/old: hipx_sorbox.hip of $REV"
