mkdir -p gpurun_out/r02f
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "rows or shard_height" > gpurun_out/r02f/pytest_rows.log 2>&1 || { tail -30 gpurun_out/r02f/pytest_rows.log; exit 1; }
tail -2 gpurun_out/r02f/pytest_rows.log
timeout -k 10 120 python tools/shape_sweep.py 50000 600000 5 10 1 2 2>&1 | grep -v invariant | tee -a gpurun_out/r02f/c4.txt
timeout -k 10 120 python tools/shape_sweep.py 50000 600000 6 10 1 2 2>&1 | grep -v invariant | tee -a gpurun_out/r02f/c4.txt
for m in 1 3 4; do NGP_TOOL_DEBUG_MODE=$m timeout -k 10 120 python tools/shape_sweep.py 50000 600000 5 10 1 2 2>&1 | grep -v invariant | tee -a gpurun_out/r02f/c4.txt; done
timeout -k 10 200 python tools/fine.py 5 50000 600000 2 > gpurun_out/r02f/c4_fine.txt 2>&1
timeout -k 10 200 python tools/stamps.py 5 50000 600000 2 > gpurun_out/r02f/c4_stamps.txt 2>&1
head -9 gpurun_out/r02f/c4_fine.txt; head -30 gpurun_out/r02f/c4_stamps.txt
