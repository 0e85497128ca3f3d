#!/bin/bash
O=gpurun_out/r04v; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_tuple.py tests/test_gpu_chains_per_pass.py -m gpu -x -q 2>&1 | tail -4 | tee $O/tests.txt
for K in 2 4 8; do timeout -k 10 300 python tools/tuple_chains_time.py 10000 100000 2 $K 20; done 2>&1 | tee $O/tuple_chains.txt
timeout -k 10 300 python tools/tuple_chains_time.py 50000 200000 2 2 10 2>&1 | tee -a $O/tuple_chains.txt
