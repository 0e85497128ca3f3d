#!/bin/bash
# chain wave: the diagonal block of a BayesPR block copied to the registers before the wave waits for its total (NGP_CHAIN_PRELOAD)
O=gpurun_out/r04ad; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2 | tee $O/tests.txt
for rep in 1 2; do
  for v in nopre pre; do
    L="NGP_HIP_LIB=$PWD/build_ab/$v.so"; if [ $v = pre ]; then L=""; fi
    echo "== $v :: C2"; env $L timeout -k 10 200 python tools/shape_sweep.py 10000 100000 8 200 | grep -v invariant
    echo "== $v :: C4"; env $L timeout -k 10 200 python tools/shape_sweep.py 50000 600000 6 40 | grep -v invariant
    echo "== $v u8 :: C4"; env $L NGP_TOOL_STORAGE=u8 timeout -k 10 200 python tools/shape_sweep.py 50000 600000 8 40 | grep -v invariant
  done
done 2>&1 | tee $O/ab.txt
