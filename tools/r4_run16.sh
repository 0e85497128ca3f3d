#!/bin/bash
O=gpurun_out/r04q; mkdir -p $O
for rep in 1 2 3; do
  echo "== new :: C2"; timeout -k 10 200 python tools/shape_sweep.py 10000 100000 8 60 | grep -v invariant
  echo "== late :: C2"; NGP_HIP_LIB=$PWD/build_ab/late.so timeout -k 10 200 python tools/shape_sweep.py 10000 100000 8 60 | grep -v invariant
  echo "== new :: C4"; NGP_TOOL_KNOB=32768 timeout -k 10 200 python tools/shape_sweep.py 50000 600000 6 40 | grep -v invariant
  echo "== late :: C4"; NGP_HIP_LIB=$PWD/build_ab/late.so NGP_TOOL_KNOB=32768 timeout -k 10 200 python tools/shape_sweep.py 50000 600000 6 40 | grep -v invariant
  echo "== new u8 :: C4"; NGP_TOOL_STORAGE=u8 timeout -k 10 200 python tools/shape_sweep.py 50000 600000 8 40 | grep -v invariant
  echo "== late u8 :: C4"; NGP_HIP_LIB=$PWD/build_ab/late.so NGP_TOOL_STORAGE=u8 timeout -k 10 200 python tools/shape_sweep.py 50000 600000 8 40 | grep -v invariant
done 2>&1 | tee $O/ab.txt
timeout -k 10 200 python tools/stamps.py 8 10000 100000 > $O/stamps_c2.txt 2>&1
NGP_HIP_LIB=$PWD/build_ab/late.so timeout -k 10 200 python tools/stamps.py 8 10000 100000 > $O/stamps_c2_late.txt 2>&1
