#!/bin/bash
# the lag-2 / lag-3 waves of the sampler ask for their Gram rows a little later, so that the lag-1 wave's rows (the critical ones) are served first
O=gpurun_out/r04z; mkdir -p $O
for rep in 1 2; do
  for v in main6 fsl4 fsl12 fsl24; do
    L="NGP_HIP_LIB=$PWD/build_ab/$v.so"
    echo "== $v :: C2"; env $L timeout -k 10 200 python tools/shape_sweep.py 10000 100000 8 200 | grep -v invariant
    echo "== $v :: C4"; env $L timeout -k 10 200 python tools/shape_sweep.py 50000 600000 6 40 | grep -v invariant
    echo "== $v u8 :: C4"; env $L NGP_TOOL_STORAGE=u8 timeout -k 10 200 python tools/shape_sweep.py 50000 600000 8 40 | grep -v invariant
  done
done 2>&1 | tee $O/ab.txt
