mkdir -p gpurun_out/r02ac
export NGP_TOOL_STORAGE=u8
for lag in 6 8 12; do
timeout -k 10 200 python tools/shape_sweep.py 50000 600000 $lag 30 1 2>&1 | tee -a gpurun_out/r02ac/c4.txt
done
for lag in 8 12; do
timeout -k 10 120 python tools/shape_sweep.py 10000 100000 $lag 50 1 2>&1 | tee -a gpurun_out/r02ac/c2.txt
done
timeout -k 10 300 python tools/shape_sweep.py 190000 600000 4 10 1 2>&1 | tee -a gpurun_out/r02ac/big.txt
