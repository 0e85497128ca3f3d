#!/bin/bash
set -o pipefail
O=gpurun_out/r04d; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.txt; tail -2 $O/pytest.txt
for rep in 1 2; do
  for near in 2 3; do
    echo "== tform near=$near :: C2"; NGP_TOOL_NEAR=$near timeout -k 10 200 python tools/shape_sweep.py 10000 100000 8 60 | grep -v invariant
  done
  echo "== steps near=3 :: C2"; NGP_TOOL_CHAIN_FORM=0 timeout -k 10 200 python tools/shape_sweep.py 10000 100000 8 60 | grep -v invariant
  echo "== r3 :: C2"; NGP_HIP_LIB=$PWD/build_ab/r3.so timeout -k 10 200 python tools/shape_sweep.py 10000 100000 8 60 | grep -v invariant
  echo "== tform lean :: C4"; NGP_TOOL_KNOB=32768 timeout -k 10 200 python tools/shape_sweep.py 50000 600000 6 40 | grep -v invariant
  echo "== tform tup :: C4"; timeout -k 10 200 python tools/shape_sweep.py 50000 600000 6 40 | grep -v invariant
  echo "== steps lean :: C4"; NGP_TOOL_KNOB=32768 NGP_TOOL_CHAIN_FORM=0 timeout -k 10 200 python tools/shape_sweep.py 50000 600000 6 40 | grep -v invariant
  echo "== steps tup :: C4"; NGP_TOOL_CHAIN_FORM=0 timeout -k 10 200 python tools/shape_sweep.py 50000 600000 6 40 | grep -v invariant
  echo "== r3 :: C4"; NGP_HIP_LIB=$PWD/build_ab/r3.so timeout -k 10 200 python tools/shape_sweep.py 50000 600000 6 40 | grep -v invariant
  echo "== tform u8 :: C4"; NGP_TOOL_STORAGE=u8 timeout -k 10 200 python tools/shape_sweep.py 50000 600000 8 40 | grep -v invariant
  echo "== r3 u8 :: C4"; NGP_TOOL_STORAGE=u8 NGP_HIP_LIB=$PWD/build_ab/r3.so timeout -k 10 200 python tools/shape_sweep.py 50000 600000 8 40 | grep -v invariant
done 2>&1 | tee $O/ab.txt
NGP_TOOL_NEAR=2 NGP_TOOL_CHAIN_FORM=1 timeout -k 10 200 python tools/stamps.py 8 10000 100000 > $O/stamps_c2_tform_near2.txt 2>&1
grep "wave\|period\|chain time" $O/stamps_c2_tform_near2.txt
