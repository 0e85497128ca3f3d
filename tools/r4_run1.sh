#!/bin/bash
# round 4, first GPU call: parity of the inverse-form chain + A/B timings (new library, its step-chain form, round-3 library, nt tile DMA)
set -o pipefail
O=gpurun_out/r04a; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.txt; tail -3 $O/pytest.txt
NEW=nextgp.jl_amd/libnextgp_hip.so
for rep in 1 2 3; do
  for cfg in "10000 100000 8 60" "50000 600000 6 40"; do
    echo "== new tform :: $cfg"; timeout -k 10 200 python tools/shape_sweep.py $cfg | grep -v invariant
    echo "== new steps :: $cfg"; NGP_TOOL_CHAIN_FORM=0 timeout -k 10 200 python tools/shape_sweep.py $cfg | grep -v invariant
    echo "== r3 :: $cfg"; NGP_HIP_LIB=$PWD/build_ab/r3.so timeout -k 10 200 python tools/shape_sweep.py $cfg | grep -v invariant
  done
  echo "== nt :: 50000 600000 6 40"; NGP_HIP_LIB=$PWD/build_ab/nt.so timeout -k 10 200 python tools/shape_sweep.py 50000 600000 6 40 | grep -v invariant
  echo "== new tform u8 :: 50000 600000"; NGP_TOOL_STORAGE=u8 timeout -k 10 200 python tools/shape_sweep.py 50000 600000 8 40 | grep -v invariant
  echo "== r3 u8 :: 50000 600000"; NGP_TOOL_STORAGE=u8 NGP_HIP_LIB=$PWD/build_ab/r3.so timeout -k 10 200 python tools/shape_sweep.py 50000 600000 8 40 | grep -v invariant
done 2>&1 | tee $O/ab.txt
# sampler alone / streamers alone, 10k x 100k (diagnostic kernel, results invalid)
for m in 2 3 1; do echo "== mode $m new"; NGP_TOOL_DEBUG_MODE=$m timeout -k 10 200 python tools/shape_sweep.py 10000 100000 8 30 | grep -v invariant; done 2>&1 | tee $O/modes.txt
timeout -k 10 200 python tools/fine.py 8 10000 100000 > $O/fine_c2.txt 2>&1
timeout -k 10 200 python tools/stamps.py 8 10000 100000 > $O/stamps_c2.txt 2>&1
tail -5 $O/fine_c2.txt
