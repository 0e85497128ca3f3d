#!/bin/bash
O=gpurun_out/r04k; mkdir -p $O
for rep in 1 2; do
  echo "== steps u8 :: C4"; NGP_TOOL_CHAIN_FORM=0 NGP_TOOL_STORAGE=u8 timeout -k 10 200 python tools/shape_sweep.py 50000 600000 8 40 | grep -v invariant
  echo "== steps u8 lag6 :: C4"; NGP_TOOL_CHAIN_FORM=0 NGP_TOOL_STORAGE=u8 timeout -k 10 200 python tools/shape_sweep.py 50000 600000 6 40 | grep -v invariant
  echo "== steps u8 lag4 :: C4"; NGP_TOOL_CHAIN_FORM=0 NGP_TOOL_STORAGE=u8 timeout -k 10 200 python tools/shape_sweep.py 50000 600000 4 40 | grep -v invariant
  echo "== tform u8 lag6 :: C4"; NGP_TOOL_STORAGE=u8 timeout -k 10 200 python tools/shape_sweep.py 50000 600000 6 40 | grep -v invariant
  echo "== r3 u8 :: C4"; NGP_TOOL_STORAGE=u8 NGP_HIP_LIB=$PWD/build_ab/r3.so timeout -k 10 200 python tools/shape_sweep.py 50000 600000 8 40 | grep -v invariant
  for lag in 4 6 8; do
    echo "== steps phase lag$lag :: C2"; NGP_TOOL_CHAIN_FORM=0 timeout -k 10 200 python tools/shape_sweep.py 10000 100000 $lag 60 1 1 | grep -v invariant
  done
  for lag in 3 4 5 6; do
    echo "== steps rows lag$lag :: C2"; NGP_TOOL_CHAIN_FORM=0 timeout -k 10 200 python tools/shape_sweep.py 10000 100000 $lag 60 1 2 | grep -v invariant
    echo "== tform rows lag$lag :: C2"; timeout -k 10 200 python tools/shape_sweep.py 10000 100000 $lag 60 1 2 | grep -v invariant
  done
  echo "== tform rows lag4 near1 :: C2"; NGP_TOOL_NEAR=1 timeout -k 10 200 python tools/shape_sweep.py 10000 100000 4 60 1 2 | grep -v invariant
  echo "== tform rows lag4 near3 :: C2"; NGP_TOOL_NEAR=3 timeout -k 10 200 python tools/shape_sweep.py 10000 100000 4 60 1 2 | grep -v invariant
done 2>&1 | tee $O/ab.txt
