mkdir -p gpurun_out/r02e; cd tools/microbench
{
for ni in 1 2 3; do ./dma_bench 3 $ni 24 51 3000 1; done
./dma_bench 3 1 16 51 3000 1
./dma_bench 3 1 32 51 3000 1
./dma_bench 1 1 24 51 3000 1
./dma_bench 3 1 24 55 3000 1
./dma_bench 3 1 11 11 12000 1
./dma_bench 3 1 5 11 12000 1
./dma_bench 3 1 10 21 8000 1
} 2>&1 | tee ../../gpurun_out/r02e/dma_bench2.txt
