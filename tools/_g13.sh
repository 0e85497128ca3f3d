mkdir -p gpurun_out/r02aa
for rep in 1 2; do for lag in 6 7; do
timeout -k 10 120 python tools/shape_sweep.py 50000 600000 $lag 30 1 2 2>&1 | grep -v invariant | sed "s/^/lag=$lag /" | tee -a gpurun_out/r02aa/c5.txt
done; done
timeout -k 10 120 python tools/shape_sweep.py 10000 100000 8 50 1 1 2>&1 | grep -v invariant | tee -a gpurun_out/r02aa/c5.txt
