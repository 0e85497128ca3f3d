#!/bin/bash
# A/B timing of library variants on the same box: ITERS=30 tools/ab.sh "lib1.so lib2.so ..." "N P lag" ["N P lag" ...]
libs="$1"; shift
for cfg in "$@"; do
  for l in $libs; do
    echo "== $l :: $cfg"
    NGP_HIP_LIB=$PWD/$l timeout -k 10 200 python tools/shape_sweep.py $cfg ${ITERS:-30} | grep -v invariant || exit 1
  done
done
