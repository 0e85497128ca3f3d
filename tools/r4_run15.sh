#!/bin/bash
O=gpurun_out/r04p; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.txt; tail -3 $O/pytest.txt
for rep in 1 2 3; do
  echo "== new :: C2"; timeout -k 10 200 python tools/shape_sweep.py 10000 100000 8 60 | grep -v invariant
  echo "== new tform :: C2"; NGP_TOOL_CHAIN_FORM=1 timeout -k 10 200 python tools/shape_sweep.py 10000 100000 8 60 | grep -v invariant
  echo "== r3 :: C2"; NGP_HIP_LIB=$PWD/build_ab/r3.so timeout -k 10 200 python tools/shape_sweep.py 10000 100000 8 60 | grep -v invariant
  echo "== new lean :: C4"; NGP_TOOL_KNOB=32768 timeout -k 10 200 python tools/shape_sweep.py 50000 600000 6 40 | grep -v invariant
  echo "== new tup :: C4"; timeout -k 10 200 python tools/shape_sweep.py 50000 600000 6 40 | grep -v invariant
  echo "== new lean tform :: C4"; NGP_TOOL_CHAIN_FORM=1 NGP_TOOL_KNOB=32768 timeout -k 10 200 python tools/shape_sweep.py 50000 600000 6 40 | grep -v invariant
  echo "== r3 :: C4"; NGP_HIP_LIB=$PWD/build_ab/r3.so timeout -k 10 200 python tools/shape_sweep.py 50000 600000 6 40 | grep -v invariant
  echo "== new u8 :: C4"; NGP_TOOL_STORAGE=u8 timeout -k 10 200 python tools/shape_sweep.py 50000 600000 8 40 | grep -v invariant
  echo "== new u8 tform :: C4"; NGP_TOOL_CHAIN_FORM=1 NGP_TOOL_STORAGE=u8 timeout -k 10 200 python tools/shape_sweep.py 50000 600000 8 40 | grep -v invariant
  echo "== r3 u8 :: C4"; NGP_TOOL_STORAGE=u8 NGP_HIP_LIB=$PWD/build_ab/r3.so timeout -k 10 200 python tools/shape_sweep.py 50000 600000 8 40 | grep -v invariant
done 2>&1 | tee $O/ab.txt
timeout -k 10 200 python tools/stamps.py 6 50000 600000 > $O/stamps_c4.txt 2>&1
timeout -k 10 200 python tools/stamps.py 8 10000 100000 > $O/stamps_c2.txt 2>&1
