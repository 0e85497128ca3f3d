mkdir -p gpurun_out/r02k
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r02k/pytest_parity.log 2>&1 || { tail -30 gpurun_out/r02k/pytest_parity.log; exit 1; }
tail -2 gpurun_out/r02k/pytest_parity.log
for lag in 5 6; do timeout -k 10 120 python tools/shape_sweep.py 50000 600000 $lag 10 1 2 2>&1 | grep -v invariant | tee -a gpurun_out/r02k/c4.txt; done
timeout -k 10 120 python tools/shape_sweep.py 50000 600000 5 10 1 1 2>&1 | grep -v invariant | tee -a gpurun_out/r02k/c4.txt
timeout -k 10 120 python tools/shape_sweep.py 10000 100000 8 20 1 1 2>&1 | grep -v invariant | tee -a gpurun_out/r02k/c4.txt
timeout -k 10 200 python tools/stamps.py 6 50000 600000 2 2>&1 | head -24 > gpurun_out/r02k/c4_stamps6.txt
cat gpurun_out/r02k/c4_stamps6.txt
