#!/bin/bash
O=gpurun_out/r04n; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.txt; tail -8 $O/pytest.txt
for k in 2 3 4; do
  for f in 0 1; do echo "== tuple k=$k form=$f"; NGP_TOOL_CHAIN_FORM=$f timeout -k 10 300 python tools/tuple_time.py 10000 100000 $k 30 | grep tuple; done
done 2>&1 | tee $O/tuple_time.txt
echo "== methods (BayesR with the division-free exponential)"; timeout -k 10 300 python tools/method_time.py 10000 100000 30 2>&1 | tee $O/method_time.txt
echo "== r3 methods"; NGP_HIP_LIB=$PWD/build_ab/r3.so timeout -k 10 300 python tools/method_time.py 10000 100000 30 2>&1 | tee $O/method_time_r3.txt
