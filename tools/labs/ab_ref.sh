#!/bin/bash
# A/B of the REFERENCE-stepper kernel for several engine builds on one box: tools/labs/ab_ref.sh ROUNDS lib1.so lib2.so ...
R=$1; shift
for ((i = 0; i < R; i++)); do
  for L in "$@"; do
    echo "== $(basename $L)"
    HWMC_LIB=$L python tools/time_kernels.py 2>/dev/null | grep -E "zbc reference|zbc superposed"
  done
done
