#!/bin/bash
# A/B of two engine builds on the SAME box, interleaved: tools/ab_curve.sh libA.so libB.so [rounds]
A=$1; B=$2; R=${3:-3}
for ((i = 0; i < R; i++)); do
  for L in "$A" "$B"; do
    echo "== $L"
    HWMC_LIB=$L python tools/block_sweep.py 2>/dev/null | grep -E "zcb +(256|512)|zbc +512"
  done
done
