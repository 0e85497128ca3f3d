#!/bin/bash
# accumulator copies per block: 8 (shards s mod 8 share a memory line's atomics) against 16 and 32
O=gpurun_out/r04ak; mkdir -p $O
for v in cp4 cp2; do NGP_HIP_LIB=$PWD/build_ab/$v.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pr_ or multi or r_" 2>&1 | tail -1; done | tee $O/tests.txt
for rep in 1 2 3; do
  for v in main cp4 cp2; do
    L="NGP_HIP_LIB=$PWD/build_ab/$v.so"; if [ $v = main ]; then L=""; fi
    echo "== $v :: C2"; env $L timeout -k 10 200 python tools/shape_sweep.py 10000 100000 6 300 | grep -v invariant
    echo "== $v :: C2 streamers alone"; env $L NGP_TOOL_DEBUG_MODE=3 timeout -k 10 200 python tools/shape_sweep.py 10000 100000 6 50 | grep -v invariant
    echo "== $v :: C4"; env $L timeout -k 10 200 python tools/shape_sweep.py 50000 600000 6 40 | grep -v invariant
    echo "== $v u8 :: C4"; env $L NGP_TOOL_STORAGE=u8 timeout -k 10 200 python tools/shape_sweep.py 50000 600000 8 40 | grep -v invariant
  done
done 2>&1 | tee $O/ab.txt
