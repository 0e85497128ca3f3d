mkdir -p gpurun_out/r02ac
timeout -k 10 900 python -m pytest tests/test_gpu_compact.py -m gpu -x -q > gpurun_out/r02ac/t.txt 2>&1 || { tail -25 gpurun_out/r02ac/t.txt; exit 1; }
tail -2 gpurun_out/r02ac/t.txt
export NGP_TOOL_STORAGE=u8
for lag in 6 8 12; do
timeout -k 10 200 python tools/shape_sweep.py 50000 600000 $lag 30 1 2>&1 | grep -v invariant | tee -a gpurun_out/r02ac/c4b.txt
done
NGP_TOOL_DEBUG_MODE=3 timeout -k 10 200 python tools/shape_sweep.py 50000 600000 8 20 1 2>&1 | grep -v invariant | tee -a gpurun_out/r02ac/c4b.txt
unset NGP_TOOL_STORAGE
for m in 0 3 4; do
NGP_TOOL_DEBUG_MODE=$m timeout -k 10 200 python tools/shape_sweep.py 50000 600000 6 20 1 2>&1 | grep -v invariant | tee -a gpurun_out/r02ac/c4b.txt
done
