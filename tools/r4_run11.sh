#!/bin/bash
O=gpurun_out/r04l; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.txt; tail -6 $O/pytest.txt
for rep in 1 2; do
  for near in 2 3; do for lag in 6 8; do
    echo "== near=$near lag=$lag"; NGP_TOOL_NEAR=$near NGP_TOOL_LAG=$lag timeout -k 10 300 python tools/method_time.py 10000 100000 40 | grep -v "^R"
  done; done
done 2>&1 | tee $O/methods_c2.txt
