#!/bin/bash
# deeper tile prefetch of the row-owning streamers (ring slots of tile u+2 one block ahead: as many as LDS holds) against the default
O=gpurun_out/r04x; mkdir -p $O
NGP_HIP_LIB=$PWD/build_ab/hdeep.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_compact.py -m gpu -x -q 2>&1 | tail -3 | tee $O/tests.txt
for rep in 1 2 3; do
  for v in main5 hdeep; do
    L="NGP_HIP_LIB=$PWD/build_ab/$v.so"
    echo "== $v :: C4"; env $L timeout -k 10 200 python tools/shape_sweep.py 50000 600000 6 40 | grep -v invariant
    echo "== $v :: C4 lag 4"; env $L timeout -k 10 200 python tools/shape_sweep.py 50000 600000 4 40 | grep -v invariant
    echo "== $v u8 :: C4"; env $L NGP_TOOL_STORAGE=u8 timeout -k 10 200 python tools/shape_sweep.py 50000 600000 8 40 | grep -v invariant
    echo "== $v :: 28k x 100k"; env $L timeout -k 10 200 python tools/shape_sweep.py 28000 100000 6 100 | grep -v invariant
  done
done 2>&1 | tee $O/ab.txt
