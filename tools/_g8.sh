mkdir -p gpurun_out/r02w
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r02w/bench_c4.json 2> gpurun_out/r02w/bench_c4.err; echo "rc=$?"; cat gpurun_out/r02w/bench_c4.json | head -c 6000; tail -5 gpurun_out/r02w/bench_c4.err
timeout -k 10 300 python -m pytest tests/test_bench_contract.py -x -q -m gpu 2>&1 | tail -5
