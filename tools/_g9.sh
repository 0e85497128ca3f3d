bash tools/profile_round.sh r02 C4 > gpurun_out/r02_profile_C4.log 2>&1; tail -12 gpurun_out/r02_profile_C4.log
bash tools/profile_round.sh r02 C2 > gpurun_out/r02_profile_C2.log 2>&1; tail -12 gpurun_out/r02_profile_C2.log
timeout -k 10 300 python bench.py --config C3 --steps 100 --warmup 10 > gpurun_out/r02_C3_bench.json 2> gpurun_out/r02_C3_bench.err; tail -c 1500 gpurun_out/r02_C3_bench.json
