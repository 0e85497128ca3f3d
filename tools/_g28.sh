mkdir -p gpurun_out/r02ag
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r02ag/all.txt 2>&1 || { tail -30 gpurun_out/r02ag/all.txt; exit 1; }
tail -3 gpurun_out/r02ag/all.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r02ag/smoke.txt 2>&1 || { tail -20 gpurun_out/r02ag/smoke.txt; exit 1; }
tail -1 gpurun_out/r02ag/smoke.txt
