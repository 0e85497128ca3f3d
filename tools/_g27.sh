mkdir -p gpurun_out/r02af
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "share_a_device" > gpurun_out/r02af/t.txt 2>&1 || { tail -15 gpurun_out/r02af/t.txt; exit 1; }
tail -2 gpurun_out/r02af/t.txt
for K in 2 3; do
timeout -k 10 200 python tools/chains_per_gpu.py 10000 100000 $K 100 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r02af/k2.txt
done
NGP_TOOL_LAG=8 timeout -k 10 200 python tools/chains_per_gpu.py 10000 100000 2 100 u8 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r02af/k2.txt
timeout -k 10 300 python tools/chains_per_gpu.py 50000 600000 3 30 u8 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r02af/k2.txt
