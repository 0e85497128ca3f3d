mkdir -p gpurun_out/r02ah
timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_compact.py -m gpu -x -q > gpurun_out/r02ah/t.txt 2>&1 || { tail -25 gpurun_out/r02ah/t.txt; exit 1; }
tail -2 gpurun_out/r02ah/t.txt
for rep in 1 2; do
timeout -k 10 120 python tools/shape_sweep.py 10000 100000 8 50 1 1 2>&1 | grep -v invariant | tee -a gpurun_out/r02ah/c.txt
NGP_TOOL_DEBUG_MODE=2 timeout -k 10 120 python tools/shape_sweep.py 10000 100000 8 50 1 1 2>&1 | grep -v invariant | tee -a gpurun_out/r02ah/c.txt
done
timeout -k 10 200 python tools/shape_sweep.py 50000 600000 6 30 1 2>&1 | grep -v invariant | tee -a gpurun_out/r02ah/c.txt
NGP_TOOL_STORAGE=u8 timeout -k 10 200 python tools/shape_sweep.py 50000 600000 8 30 1 2>&1 | grep -v invariant | tee -a gpurun_out/r02ah/c.txt
NGP_TOOL_STORAGE=u8 timeout -k 10 200 python tools/shape_sweep.py 10000 100000 8 50 1 2>&1 | grep -v invariant | tee -a gpurun_out/r02ah/c.txt
