#!/bin/bash
# 10k x 100k (phase streamer): the lag again, now that the hand-off is one hop
O=gpurun_out/r04ai; mkdir -p $O
for rep in 1 2 3; do
  for lag in 4 5 6 7 8; do
    echo "== lag $lag"; timeout -k 10 200 python tools/shape_sweep.py 10000 100000 $lag 300 | grep -v invariant
  done
done 2>&1 | tee $O/lags.txt
for lag in 6 8; do echo "== 8k lag $lag"; timeout -k 10 200 python tools/shape_sweep.py 8000 100000 $lag 300 | grep -v invariant; echo "== 14k lag $lag"; timeout -k 10 200 python tools/shape_sweep.py 14000 100000 $lag 300 | grep -v invariant; echo "== 4k lag $lag"; timeout -k 10 200 python tools/shape_sweep.py 4000 50000 $lag 300 | grep -v invariant; done 2>&1 | tee -a $O/lags.txt
for lag in 6 8; do echo "== chains8 lag $lag"; timeout -k 10 200 python tools/chains_per_pass.py 10000 100000 8 100 $lag | cut -c1-150; done 2>&1 | tee -a $O/lags.txt
