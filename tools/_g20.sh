mkdir -p gpurun_out/r02ac
export NGP_TOOL_STORAGE=u8
for m in 1 2 3 4 6; do
NGP_TOOL_DEBUG_MODE=$m timeout -k 10 200 python tools/shape_sweep.py 50000 600000 8 20 1 2>&1 | grep -v invariant | tee -a gpurun_out/r02ac/c4m.txt
done
for near in 1 3 4; do
NGP_TOOL_NEAR=$near timeout -k 10 200 python tools/shape_sweep.py 50000 600000 8 20 1 2>&1 | grep -v invariant | tee -a gpurun_out/r02ac/c4m.txt
done
timeout -k 10 300 python tools/shape_sweep.py 190000 600000 4 10 1 2>&1 | tee -a gpurun_out/r02ac/big.txt
