mkdir -p gpurun_out/r02ab
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rows_lag8 or rows_lag10 or rows_lag12 or rows_lag6_near2 or rows_lag3" > gpurun_out/r02ab/t.txt 2>&1 || { tail -30 gpurun_out/r02ab/t.txt; exit 1; }
tail -3 gpurun_out/r02ab/t.txt
for rep in 1 2; do
timeout -k 10 120 python tools/shape_sweep.py 10000 100000 8 50 1 1 2>&1 | grep -v invariant | tee -a gpurun_out/r02ab/c2.txt
for lag in 6 8 10 12; do
timeout -k 10 120 python tools/shape_sweep.py 10000 100000 $lag 50 1 2 2>&1 | grep -v invariant | tee -a gpurun_out/r02ab/c2.txt
done; done
for near in 2 4; do for lag in 8 12; do
NGP_TOOL_NEAR=$near timeout -k 10 120 python tools/shape_sweep.py 10000 100000 $lag 50 1 2 2>&1 | grep -v invariant | sed "s/^/near=$near /" | tee -a gpurun_out/r02ab/c2.txt
done; done
