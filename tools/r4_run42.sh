#!/bin/bash
# 10k x 100k, row-owning streamer: the whole tile u+2 requested two blocks ahead (build_ab/hdeep.so) -- the streamers alone and the whole sweep
O=gpurun_out/r04aj; mkdir -p $O
for v in main hdeep; do
  L="NGP_FORCE_STREAMER=1 NGP_HIP_LIB=$PWD/build_ab/$v.so"; if [ $v = main ]; then L="NGP_FORCE_STREAMER=1"; fi
  for lag in 4 6; do
    echo "== $v rows lag $lag :: whole"; env $L timeout -k 10 200 python tools/shape_sweep.py 10000 100000 $lag 200 1 2 | grep -v invariant
    echo "== $v rows lag $lag :: streamers alone"; env $L NGP_TOOL_DEBUG_MODE=3 timeout -k 10 200 python tools/shape_sweep.py 10000 100000 $lag 50 1 2 | grep -v invariant
  done
  echo "== $v u8 lag 8 :: whole"; env $L NGP_TOOL_STORAGE=u8 timeout -k 10 200 python tools/shape_sweep.py 10000 100000 8 200 | grep -v invariant
  echo "== $v u8 lag 8 :: streamers alone"; env $L NGP_TOOL_STORAGE=u8 NGP_TOOL_DEBUG_MODE=3 timeout -k 10 200 python tools/shape_sweep.py 10000 100000 8 50 | grep -v invariant
done 2>&1 | tee $O/hdeep_c2.txt
