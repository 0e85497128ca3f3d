#!/bin/bash
O=gpurun_out/r04j; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.txt; tail -3 $O/pytest.txt
for rep in 1 2; do
  echo "== tform :: C2"; timeout -k 10 200 python tools/shape_sweep.py 10000 100000 8 60 | grep -v invariant
  echo "== steps :: C2"; NGP_TOOL_CHAIN_FORM=0 timeout -k 10 200 python tools/shape_sweep.py 10000 100000 8 60 | grep -v invariant
  echo "== r3 :: C2"; NGP_HIP_LIB=$PWD/build_ab/r3.so timeout -k 10 200 python tools/shape_sweep.py 10000 100000 8 60 | grep -v invariant
  echo "== tform lean :: C4"; NGP_TOOL_KNOB=32768 timeout -k 10 200 python tools/shape_sweep.py 50000 600000 6 40 | grep -v invariant
  echo "== tform tup :: C4"; timeout -k 10 200 python tools/shape_sweep.py 50000 600000 6 40 | grep -v invariant
  echo "== steps lean :: C4"; NGP_TOOL_KNOB=32768 NGP_TOOL_CHAIN_FORM=0 timeout -k 10 200 python tools/shape_sweep.py 50000 600000 6 40 | grep -v invariant
  echo "== steps tup :: C4"; NGP_TOOL_CHAIN_FORM=0 timeout -k 10 200 python tools/shape_sweep.py 50000 600000 6 40 | grep -v invariant
  echo "== r3 :: C4"; NGP_HIP_LIB=$PWD/build_ab/r3.so timeout -k 10 200 python tools/shape_sweep.py 50000 600000 6 40 | grep -v invariant
  echo "== steps lean lag5 :: C4"; NGP_TOOL_KNOB=32768 NGP_TOOL_CHAIN_FORM=0 timeout -k 10 200 python tools/shape_sweep.py 50000 600000 5 40 | grep -v invariant
  echo "== steps lean lag4 :: C4"; NGP_TOOL_KNOB=32768 NGP_TOOL_CHAIN_FORM=0 timeout -k 10 200 python tools/shape_sweep.py 50000 600000 4 40 | grep -v invariant
  echo "== tform u8 :: C4"; NGP_TOOL_STORAGE=u8 timeout -k 10 200 python tools/shape_sweep.py 50000 600000 8 40 | grep -v invariant
  echo "== steps u8 :: C4"; NGP_TOOL_CHAIN_FORM=0 NGP_TOOL_STORAGE=u8 timeout -k 10 200 python tools/shape_sweep.py 50000 600000 8 40 | grep -v invariant
  echo "== r3 u8 :: C4"; NGP_TOOL_STORAGE=u8 NGP_HIP_LIB=$PWD/build_ab/r3.so timeout -k 10 200 python tools/shape_sweep.py 50000 600000 8 40 | grep -v invariant
done 2>&1 | tee $O/ab.txt
NGP_TOOL_CHAIN_FORM=0 timeout -k 10 200 python tools/stamps.py 6 50000 600000 > $O/stamps_c4_new.txt 2>&1
