#!/bin/bash
# sampler with two near lags: the lag-1 term of the next block formed while the chain runs (NGP_CHUNKED, on top of NGP_ALT_LAGS)
O=gpurun_out/r04af; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_compact.py tests/test_gpu_chains_per_pass.py -m gpu -x -q 2>&1 | tail -2 | tee $O/tests.txt
for rep in 1 2 3; do
  for v in noalt chunked; do
    L="NGP_HIP_LIB=$PWD/build_ab/$v.so"; if [ $v = chunked ]; then L=""; fi
    echo "== $v :: C4"; env $L timeout -k 10 200 python tools/shape_sweep.py 50000 600000 6 40 | grep -v invariant
    echo "== $v u8 :: C4"; env $L NGP_TOOL_STORAGE=u8 timeout -k 10 200 python tools/shape_sweep.py 50000 600000 8 40 | grep -v invariant
    echo "== $v :: 28k"; env $L timeout -k 10 200 python tools/shape_sweep.py 28000 100000 6 100 | grep -v invariant
    echo "== $v near2 :: C2"; env $L NGP_TOOL_NEAR=2 timeout -k 10 200 python tools/shape_sweep.py 10000 100000 8 200 | grep -v invariant
    echo "== $v u8 near2 :: C2"; env $L NGP_TOOL_NEAR=2 NGP_TOOL_STORAGE=u8 timeout -k 10 200 python tools/shape_sweep.py 10000 100000 8 200 | grep -v invariant
  done
done 2>&1 | tee $O/ab.txt
