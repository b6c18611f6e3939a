#!/bin/bash
# step time / chunk hop of the plane march (scripts/sor_box_hop.py) for a list of variants: new | old (scripts/ab_build_old.sh -> ab/old) | dN (HIPX_SORBOX_DEBUG=N)
cd "$(dirname "$0")/.." || exit 1
for V in "$@"; do
  unset HIPX_LIBDIR HIPX_SORBOX_DEBUG
  case $V in
    old) export HIPX_LIBDIR=$PWD/ab/old ;;
    d*) export HIPX_SORBOX_DEBUG=${V#d} ;;
  esac
  echo "== $V"
  timeout 120 python scripts/sor_box_hop.py ${NX:-2048} 2>&1 | grep -v amdgpu | tail -${TAILN:-1}
done
