#!/bin/bash
O=gpurun_out/r04y; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 | tee $O/tests_p.txt
for rep in 1 2; do for M in R4 R8; do NGP_TOOL_METHODS=$M timeout -k 10 200 python tools/method_time.py 10000 100000 10; done; done 2>&1 | tee $O/steps3.txt
