#!/bin/bash
# 10k x 100k: the streamers alone (timing mode 3 of the diagnostic kernel) and the whole sweep, phase streamer against row-owning streamer
O=gpurun_out/r04ah; mkdir -p $O
for st in 1 2; do
  for lag in 4 6; do
    echo "== streamer $st lag $lag :: whole sweep"; timeout -k 10 200 python tools/shape_sweep.py 10000 100000 $lag 200 1 $st | grep -v invariant
    echo "== streamer $st lag $lag :: streamers alone (mode 3)"; NGP_TOOL_DEBUG_MODE=3 timeout -k 10 200 python tools/shape_sweep.py 10000 100000 $lag 50 1 $st | grep -v invariant
    echo "== streamer $st lag $lag :: sampler alone (mode 2)"; NGP_TOOL_DEBUG_MODE=2 timeout -k 10 200 python tools/shape_sweep.py 10000 100000 $lag 50 1 $st | grep -v invariant
  done
done 2>&1 | tee $O/modes.txt
echo "== phase lag 8"; timeout -k 10 200 python tools/shape_sweep.py 10000 100000 8 200 | grep -v invariant
NGP_TOOL_DEBUG_MODE=3 timeout -k 10 200 python tools/shape_sweep.py 10000 100000 8 50 | grep -v invariant
NGP_TOOL_DEBUG_MODE=2 timeout -k 10 200 python tools/shape_sweep.py 10000 100000 8 50 | grep -v invariant
