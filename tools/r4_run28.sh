#!/bin/bash
O=gpurun_out/r04y; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/tests_full.txt
for rep in 1 2; do for M in C R4 R8; do NGP_TOOL_METHODS=$M timeout -k 10 200 python tools/method_time.py 10000 100000 10; done; done 2>&1 | tee $O/steps.txt
NGP_TOOL_METHODS=R4 timeout -k 10 300 python tools/method_time.py 50000 200000 10 2>&1 | tee -a $O/steps.txt
