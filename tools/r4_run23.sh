#!/bin/bash
O=gpurun_out/r04u; mkdir -p $O
for M in B C R4 R8; do NGP_TOOL_METHODS=$M timeout -k 10 200 python tools/method_time.py 10000 100000 10; done 2>&1 | tee $O/steps.txt
NGP_TOOL_METHODS=R4 timeout -k 10 200 python tools/method_time.py 10000 100000 100 2>&1 | tee -a $O/steps.txt
