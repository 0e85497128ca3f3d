#!/bin/bash
# BayesR chain: branch-free interleaved exponentials, class coefficients chosen by selects -- parity and time against main4
O=gpurun_out/r04u; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "bayesr or BayesR or class or rform or methods or det_exp" 2>&1 | tail -3 | tee $O/tests.txt
for rep in 1 2; do
  for v in main4 new; do
    L=""; if [ $v != new ]; then L="NGP_HIP_LIB=$PWD/build_ab/$v.so"; fi
    for M in R4 R8; do
      echo "== $v $M"; env $L NGP_TOOL_METHODS=$M timeout -k 10 200 python tools/method_time.py 10000 100000 10
    done
  done
done 2>&1 | tee $O/ab.txt
