mkdir -p gpurun_out/r02af
for K in 1 2 4; do
timeout -k 10 200 python tools/chains_per_gpu.py 10000 100000 $K 100 2>&1 | tee -a gpurun_out/r02af/k.txt
done
for K in 1 2 3; do
timeout -k 10 300 python tools/chains_per_gpu.py 50000 600000 $K 30 u8 2>&1 | tee -a gpurun_out/r02af/k.txt
done
