mkdir -p gpurun_out/r02x
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r02x/pytest_gpu.log 2>&1; echo "rc=$?"; tail -25 gpurun_out/r02x/pytest_gpu.log
