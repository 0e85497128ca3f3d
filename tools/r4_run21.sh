#!/bin/bash
O=gpurun_out/r04t; mkdir -p $O
for rep in 1 2 3; do
  echo "== main4 :: C4"; env NGP_HIP_LIB=$PWD/build_ab/main4.so timeout -k 10 200 python tools/shape_sweep.py 50000 600000 6 40 | grep -v invariant
  echo "== new(3 far) :: C4"; timeout -k 10 200 python tools/shape_sweep.py 50000 600000 6 40 | grep -v invariant
  for sh in 250 252; do
    echo "== new(3 far) shards $sh :: C4"; NGP_TOOL_SHARDS=$sh timeout -k 10 200 python tools/shape_sweep.py 50000 600000 6 40 | grep -v invariant
  done
done 2>&1 | tee $O/ab2.txt
