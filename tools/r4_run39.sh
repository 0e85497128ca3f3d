#!/bin/bash
# sampler's lag-1 product: dlt through four 8-byte LDS reads per lane + DPP broadcast (NGP_LAG1_DPP) against sixteen broadcast reads per chunk
O=gpurun_out/r04ag; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_compact.py -m gpu -x -q 2>&1 | tail -2 | tee $O/tests.txt
for rep in 1 2 3; do
  for v in main8 dpp; do
    L="NGP_HIP_LIB=$PWD/build_ab/$v.so"; if [ $v = dpp ]; then L=""; fi
    echo "== $v :: C2"; env $L timeout -k 10 200 python tools/shape_sweep.py 10000 100000 8 200 | grep -v invariant
    echo "== $v :: C4"; env $L timeout -k 10 200 python tools/shape_sweep.py 50000 600000 6 40 | grep -v invariant
    echo "== $v u8 :: C4"; env $L NGP_TOOL_STORAGE=u8 timeout -k 10 200 python tools/shape_sweep.py 50000 600000 8 40 | grep -v invariant
  done
done 2>&1 | tee $O/ab.txt
