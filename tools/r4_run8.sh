#!/bin/bash
O=gpurun_out/r04i; mkdir -p $O
NGP_TOOL_CHAIN_FORM=0 timeout -k 10 200 python tools/stamps.py 6 50000 600000 > $O/stamps_c4_new.txt 2>&1
NGP_HIP_LIB=$PWD/build_ab/r3.so timeout -k 10 200 python tools/stamps.py 6 50000 600000 > $O/stamps_c4_r3.txt 2>&1
tail -12 $O/stamps_c4_r3.txt; tail -12 $O/stamps_c4_new.txt
