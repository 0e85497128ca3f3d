#!/bin/bash
O=gpurun_out/r04s; mkdir -p $O
for rep in 1 2 3; do
  for v in main warm3 warm3nt; do
    L=""; if [ $v != main ]; then L="NGP_HIP_LIB=$PWD/build_ab/$v.so"; fi
    echo "== $v :: C4"; env $L timeout -k 10 200 python tools/shape_sweep.py 50000 600000 6 40 | grep -v invariant
    echo "== $v u8 :: C4"; env $L NGP_TOOL_STORAGE=u8 timeout -k 10 200 python tools/shape_sweep.py 50000 600000 8 40 | grep -v invariant
  done
done 2>&1 | tee $O/ab.txt
