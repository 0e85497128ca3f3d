#!/bin/bash
O=gpurun_out/r04o; mkdir -p $O
for m in 1 3 6 0; do echo "== mode $m C4 lag6"; NGP_TOOL_DEBUG_MODE=$m timeout -k 10 200 python tools/shape_sweep.py 50000 600000 6 30 | grep -v invariant; done 2>&1 | tee $O/modes_c4.txt
for m in 1 3; do echo "== mode $m C4 lag4"; NGP_TOOL_DEBUG_MODE=$m timeout -k 10 200 python tools/shape_sweep.py 50000 600000 4 30 | grep -v invariant; done 2>&1 | tee -a $O/modes_c4.txt
timeout -k 10 200 python tools/fine.py 6 50000 600000 > $O/fine_c4.txt 2>&1; cat $O/fine_c4.txt | tail -12
NGP_TOOL_DEBUG_MODE=3 timeout -k 10 200 python tools/fine.py 6 50000 600000 > $O/fine_c4_mode3.txt 2>&1; cat $O/fine_c4_mode3.txt | tail -12
NGP_TOOL_DEBUG_MODE=1 timeout -k 10 200 python tools/fine.py 6 50000 600000 > $O/fine_c4_mode1.txt 2>&1; cat $O/fine_c4_mode1.txt | tail -12
