#!/bin/bash
# Plane-march PCSOR A/B: bit parity on the small boxes, timings on the large ones, per-workgroup stamps of one application.
#   bash scripts/r06_sorbox_g.sh [variant ...]     variant = new | old (the library scripts/ab_build_old.sh puts into ab/old) | dN (HIPX_SORBOX_DEBUG=N: timing probes, wrong results)
#   QUICK=1: timings only; QUICK=2: timings and stamps
cd "$(dirname "$0")/.." || exit 1
for V in ${@:-new}; do
  unset HIPX_LIBDIR HIPX_SORBOX_DEBUG
  case $V in
    old) export HIPX_LIBDIR=$PWD/ab/old ;;
    d*) export HIPX_SORBOX_DEBUG=${V#d} ;;  # timing probes, wrong results: 4 = no south poller, 8 = top plane flushed in groups of eight (nothing stored), 12 = both
  esac
  echo "=== variant $V"
  if [ -z "$QUICK" ]; then
    timeout 600 python scripts/sor_box_check.py quick 2>&1 | grep -v amdgpu.ids > /tmp/sbq.txt
    echo "small boxes: $(grep -c 'bit-identical to the oracle' /tmp/sbq.txt) bit-identical, $(grep -c 'DIFFERENT\|ERROR' /tmp/sbq.txt) different"
    grep 'DIFFERENT\|ERROR' /tmp/sbq.txt | head -5
    grep ' rows ' /tmp/sbq.txt
  fi
  timeout 600 python scripts/sor_box_check.py timeonly 2>&1 | grep ' rows ' | grep "256^3\|slab"
  [ "$QUICK" = 1 ] || HIPX_SORBOX_STATS=2 timeout 600 python scripts/sor_box_check.py stats 2>&1 | grep -A4 "sorbox" | grep -v "J=[2-9]\|J=1[0-9]" | cut -c1-420 | head -12
done
