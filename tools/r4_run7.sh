#!/bin/bash
set -o pipefail
O=gpurun_out/r04h; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.txt; tail -15 $O/pytest.txt
NGP_TOOL_CHAIN_FORM=0 timeout -k 10 200 python tools/stamps.py 6 50000 600000 > $O/stamps_c4_new.txt 2>&1
NGP_HIP_LIB=$PWD/build_ab/r3.so timeout -k 10 200 python tools/stamps.py 6 50000 600000 > $O/stamps_c4_r3.txt 2>&1
NGP_TOOL_CHAIN_FORM=0 timeout -k 10 200 python tools/stamps.py 8 10000 100000 > $O/stamps_c2_new.txt 2>&1
NGP_HIP_LIB=$PWD/build_ab/r3.so timeout -k 10 200 python tools/stamps.py 8 10000 100000 > $O/stamps_c2_r3.txt 2>&1
for rep in 1 2; do
  echo "== tform u8 :: C4"; NGP_TOOL_STORAGE=u8 timeout -k 10 200 python tools/shape_sweep.py 50000 600000 8 40 | grep -v invariant
  echo "== steps u8 :: C4"; NGP_TOOL_CHAIN_FORM=0 NGP_TOOL_STORAGE=u8 timeout -k 10 200 python tools/shape_sweep.py 50000 600000 8 40 | grep -v invariant
  echo "== r3 u8 :: C4"; NGP_TOOL_STORAGE=u8 NGP_HIP_LIB=$PWD/build_ab/r3.so timeout -k 10 200 python tools/shape_sweep.py 50000 600000 8 40 | grep -v invariant
done 2>&1 | tee $O/ab.txt
