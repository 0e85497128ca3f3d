#!/bin/bash
# y of a quad by one 8-byte LDS read per lane + DPP broadcast in (a) the byte streamer, (b) the two-chain row-owning streamers
O=gpurun_out/r04aa; mkdir -p $O
NGP_HIP_LIB=$PWD/build_ab/dppu8.so timeout -k 10 600 python -m pytest tests/test_gpu_compact.py -m gpu -x -q 2>&1 | tail -2 | tee $O/tests_u8.txt
NGP_HIP_LIB=$PWD/build_ab/dppm.so timeout -k 10 600 python -m pytest tests/test_gpu_chains_per_pass.py -m gpu -x -q 2>&1 | tail -2 | tee $O/tests_m.txt
for rep in 1 2; do
  for v in main7 dppu8; do
    L="NGP_HIP_LIB=$PWD/build_ab/$v.so"
    echo "== $v u8 :: C4"; env $L NGP_TOOL_STORAGE=u8 timeout -k 10 200 python tools/shape_sweep.py 50000 600000 8 40 | grep -v invariant
    echo "== $v u8 :: C2"; env $L NGP_TOOL_STORAGE=u8 timeout -k 10 200 python tools/shape_sweep.py 10000 100000 8 200 | grep -v invariant
  done
  for v in main7 dppm; do
    L="NGP_HIP_LIB=$PWD/build_ab/$v.so"
    echo "== $v fp32 2 chains :: C4"; env $L timeout -k 10 300 python tools/chains_per_pass.py 50000 600000 2 20 4
    echo "== $v u8 2 chains :: C4"; env $L NGP_TOOL_STORAGE=u8 timeout -k 10 300 python tools/chains_per_pass.py 50000 600000 2 20
    echo "== $v u8 2 chains :: C2"; env $L NGP_TOOL_STORAGE=u8 timeout -k 10 300 python tools/chains_per_pass.py 10000 100000 2 100
  done
done 2>&1 | tee $O/ab.txt
