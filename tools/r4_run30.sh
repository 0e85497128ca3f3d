#!/bin/bash
O=gpurun_out/r04y; mkdir -p $O
for v in nolazy; do for M in R4 R8; do NGP_HIP_LIB=$PWD/build_ab/$v.so NGP_TOOL_METHODS=$M timeout -k 10 300 python tools/method_time.py 10000 100000 5; done; done 2>&1 | tee $O/nolazy.txt
