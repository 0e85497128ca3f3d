mkdir -p gpurun_out/r02y
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r02y/pytest_gpu.log 2>&1; echo "rc=$?"; tail -8 gpurun_out/r02y/pytest_gpu.log
