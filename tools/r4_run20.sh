#!/bin/bash
# far-lag workgroups 8 -> 1 (more CUs for streamers) A/B against the library before (build_ab/main4.so)
O=gpurun_out/r04t; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_compact.py tests/test_gpu_chains_per_pass.py -m gpu -x -q 2>&1 | tail -3 | tee $O/tests.txt || exit 1
for rep in 1 2; do
  for v in main4 new; do
    L=""; if [ $v != new ]; then L="NGP_HIP_LIB=$PWD/build_ab/$v.so"; fi
    echo "== $v :: C4"; env $L timeout -k 10 200 python tools/shape_sweep.py 50000 600000 6 40 | grep -v invariant
    echo "== $v u8 :: C4"; env $L NGP_TOOL_STORAGE=u8 timeout -k 10 200 python tools/shape_sweep.py 50000 600000 8 40 | grep -v invariant
    echo "== $v :: C2"; env $L timeout -k 10 200 python tools/shape_sweep.py 10000 100000 8 200 | grep -v invariant
    echo "== $v u8 :: C2"; env $L NGP_TOOL_STORAGE=u8 timeout -k 10 200 python tools/shape_sweep.py 10000 100000 8 200 | grep -v invariant
  done
  for sh in 250 252 254; do
    echo "== new shards $sh :: C4"; NGP_TOOL_SHARDS=$sh timeout -k 10 200 python tools/shape_sweep.py 50000 600000 6 40 | grep -v invariant
    echo "== new shards $sh :: C2"; NGP_TOOL_SHARDS=$sh timeout -k 10 200 python tools/shape_sweep.py 10000 100000 8 200 | grep -v invariant
  done
  echo "== new shards 254 u8 :: C4"; NGP_TOOL_SHARDS=254 NGP_TOOL_STORAGE=u8 timeout -k 10 200 python tools/shape_sweep.py 50000 600000 8 40 | grep -v invariant
done 2>&1 | tee $O/ab.txt
