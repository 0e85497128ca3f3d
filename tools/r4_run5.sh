#!/bin/bash
set -o pipefail
O=gpurun_out/r04e; mkdir -p $O
for rep in 1 2 3; do
  echo "== tform :: C2"; timeout -k 10 200 python tools/shape_sweep.py 10000 100000 8 60 | grep -v invariant
  echo "== tform near=2 :: C2"; NGP_TOOL_NEAR=2 timeout -k 10 200 python tools/shape_sweep.py 10000 100000 8 60 | grep -v invariant
  echo "== steps :: C2"; NGP_TOOL_CHAIN_FORM=0 timeout -k 10 200 python tools/shape_sweep.py 10000 100000 8 60 | grep -v invariant
  echo "== r3 :: C2"; NGP_HIP_LIB=$PWD/build_ab/r3.so timeout -k 10 200 python tools/shape_sweep.py 10000 100000 8 60 | grep -v invariant
  echo "== tform lean :: C4"; NGP_TOOL_KNOB=32768 timeout -k 10 200 python tools/shape_sweep.py 50000 600000 6 40 | grep -v invariant
  echo "== tform tup :: C4"; timeout -k 10 200 python tools/shape_sweep.py 50000 600000 6 40 | grep -v invariant
  echo "== steps lean :: C4"; NGP_TOOL_KNOB=32768 NGP_TOOL_CHAIN_FORM=0 timeout -k 10 200 python tools/shape_sweep.py 50000 600000 6 40 | grep -v invariant
  echo "== steps tup :: C4"; NGP_TOOL_CHAIN_FORM=0 timeout -k 10 200 python tools/shape_sweep.py 50000 600000 6 40 | grep -v invariant
  echo "== r3 :: C4"; NGP_HIP_LIB=$PWD/build_ab/r3.so timeout -k 10 200 python tools/shape_sweep.py 50000 600000 6 40 | grep -v invariant
done 2>&1 | tee $O/ab.txt
