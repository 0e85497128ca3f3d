mkdir -p gpurun_out/r03x
for rep in 1 2; do for lib in build_ab/head.so nextgp.jl_amd/libnextgp_hip.so; do for K in 3 8; do NGP_HIP_LIB=$PWD/$lib timeout -k 10 120 python tools/chains_per_pass.py 10000 100000 $K 100 8 | sed "s|^|$lib |"; done; done; done > gpurun_out/r03x/sdma_ab.txt 2>&1
grep -o "^[a-z_./A-Z]* \|chains per pass=[0-9]*\|[0-9.]* it/s" gpurun_out/r03x/sdma_ab.txt | paste - - -
