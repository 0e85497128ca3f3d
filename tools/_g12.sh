mkdir -p gpurun_out/r02final
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r02final/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r02final/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/profile_round.sh r02 C4 > gpurun_out/r02_profile_C4.log 2>&1; tail -12 gpurun_out/r02_profile_C4.log
timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02final/bench_default.json 2> gpurun_out/r02final/bench_default.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02final/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["launch_avg_ms"], d["roofline"]["traffic"], d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["effective_samples"]["ess_min_per_sec"], d["config"]["layout"])
PY
