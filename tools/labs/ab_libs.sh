#!/bin/bash
# A/B of several engine builds on the SAME box, interleaved: tools/labs/ab_libs.sh ROUNDS lib1.so lib2.so ...
R=$1; shift
for ((i = 0; i < R; i++)); do
  for L in "$@"; do
    echo "== $(basename $L)"
    HWMC_LIB=$L python tools/labs/block_sweep.py 2>/dev/null | grep -E "zcb +(256|512)"
  done
done
