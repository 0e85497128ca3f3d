mkdir -p gpurun_out/r02y
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r02y/pytest_gpu2.log 2>&1; echo "rc=$?"; tail -12 gpurun_out/r02y/pytest_gpu2.log
