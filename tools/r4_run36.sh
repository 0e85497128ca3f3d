#!/bin/bash
O=gpurun_out/r04ad; mkdir -p $O
for v in pre16 pre32 pre48; do NGP_HIP_LIB=$PWD/build_ab/$v.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pr_ or multi or b_single" 2>&1 | tail -1; done | tee $O/tests2.txt
for rep in 1 2; do
  for v in nopre pre16 pre32 pre48; do
    L="NGP_HIP_LIB=$PWD/build_ab/$v.so"
    echo "== $v :: C2"; env $L timeout -k 10 200 python tools/shape_sweep.py 10000 100000 8 200 | grep -v invariant
    echo "== $v :: C4"; env $L timeout -k 10 200 python tools/shape_sweep.py 50000 600000 6 40 | grep -v invariant
    echo "== $v u8 :: C4"; env $L NGP_TOOL_STORAGE=u8 timeout -k 10 200 python tools/shape_sweep.py 50000 600000 8 40 | grep -v invariant
  done
done 2>&1 | tee $O/ab2.txt
