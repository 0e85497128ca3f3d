#!/bin/bash
O=gpurun_out/r04w; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_chains_per_pass.py tests/test_gpu_compact.py -m gpu -x -q 2>&1 | tail -6 | tee $O/tests.txt
for K in 1 2; do NGP_TOOL_STORAGE=u8 timeout -k 10 300 python tools/chains_per_pass.py 50000 600000 $K 20; done 2>&1 | tee $O/u8_chains.txt
for K in 1 2; do NGP_TOOL_STORAGE=u8 timeout -k 10 300 python tools/chains_per_pass.py 10000 100000 $K 100; done 2>&1 | tee -a $O/u8_chains.txt
