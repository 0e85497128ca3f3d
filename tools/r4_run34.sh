#!/bin/bash
O=gpurun_out/r04ab; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_chains_per_pass.py -m gpu -x -q 2>&1 | tail -3 | tee $O/tests.txt
for K in 1 2 3 4; do NGP_TOOL_STORAGE=u8 timeout -k 10 300 python tools/chains_per_pass.py 10000 100000 $K 100; done 2>&1 | tee $O/u8_chains.txt
for K in 2 3 4; do NGP_TOOL_STORAGE=u8 timeout -k 10 300 python tools/chains_per_pass.py 50000 600000 $K 20; done 2>&1 | tee -a $O/u8_chains.txt
timeout -k 10 300 python tools/chains_per_pass.py 50000 600000 2 20 4 2>&1 | tee -a $O/u8_chains.txt
timeout -k 10 300 python tools/chains_per_pass.py 10000 100000 8 100 2>&1 | tee -a $O/u8_chains.txt
