"""bench.py prints ONE JSON line with the fields the driver reads (small shape, seconds on the GPU)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_has_the_contract_fields():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "C4", "--N", "2000", "--P", "6400", "--steps", "8", "--warmup", "1",
                          "--cpu-cols", "640", "--cpu-seconds", "2", "--chains-per-gpu", "2", "--chains-per-pass", "3"], capture_output=True, text=True,
                         timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == "gibbs_iterations_per_sec" and d["unit"] == "it/s" and d["n_gpus"] == 1 and d["steps"] == 8 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert "traffic" in r
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c and c["cpu_model"] and c["one_thread_it_per_s"] > 0
    assert len(d["config"]["sets"]) == 3 and sum(x["ncol"] for x in d["config"]["sets"]) == 6400      # three BayesPR sets, like configs[3]
    e = d["effective_samples"]   # (an ESS figure needs at least 200 timed iterations behind it: below that the field says so)
    assert e["ess"] is None and e["ess_min"] is None and e["ess_min_per_sec"] is None and "fewer than 200" in e["note"]
    kp = d["chains_per_pass"]   # optional leg: three chains in ONE fused sweep launch per iteration, aggregate rate beside the headline
    assert kp["chains"] == 3 and kp["value"] > 0 and kp["sweep_launches"] == 8 and kp["algorithmic_bytes_per_pass"] == 4.0 * 2000 * 6400
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    kc = d["chains_per_gpu"]    # optional leg: two chains side by side on the GPU, aggregate rate
    assert kc["chains"] == 2 and kc["value"] > 0 and kc["shards_per_chain"] <= 123
    cs = d["compact_storage"]   # the extra leg, beside the fp32 headline
    assert cs["value"] > 0 and cs["panel_bytes"] == 2000 * 6400 and cs["layout"]["streamer"] == 3 and cs["roofline"]["peak"] == 8000.0
    # a launch that ended at its census and was run again alone would be an invisible slowdown: every leg reports it, and a
    # single-process run must not have one
    assert d["census_retries"] == 0 and d["exclusive"] is False
    assert kp["census_retries"] == 0 and kp["exclusive"] is False and kp["fused"] is True
    assert kc["census_retries"] == 0


def test_two_ranks_on_one_gpu_pool_their_posterior_sums(tmp_path):
    """The N > 1 leg of bench.py, rehearsed with two fresh child processes on ONE GPU (gloo instead of RCCL, --same-device): rank 0
    prints one JSON line with n_gpus == 2, the all-reduce timed, 2 x 8 kept samples pooled, and the pooled posterior mean of varE
    equal to the mean over the two chains run alone (what an 8-GPU node would do differently is the ncclAllReduce call itself)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "C4", "--N", "2000", "--P", "6400", "--steps", "8", "--warmup", "0",
            "--no-cpu-baseline", "--no-compact", "--chains-per-pass", "0"]
    procs = []
    for rank in range(2):
        env = dict(os.environ, WORLD_SIZE="2", RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen(base + ["--gpus", "2", "--backend", "gloo", "--same-device"], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True, cwd=ROOT))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
    lines = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(lines) == 1 and not [l for l in outs[1][0].splitlines() if l.startswith("{")]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["allreduce_ms"] is not None and d["pooled_kept_samples"] == 16 and d["config"]["chains"] == 2
    assert abs(d["value"] - 2 * 8 / (d["ms_per_step"] * 8e-3)) < 1e-6 * d["value"]           # whole-job rate over both ranks
    # the two chains alone (seeds 1001 and 1002, as ranks 0 and 1 use them)
    alone = []
    for rank in range(2):
        o = subprocess.run(base + ["--gpus", "1", "--seed-offset", str(rank)], capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert o.returncode == 0, o.stderr[-2000:]
        alone.append(json.loads([l for l in o.stdout.splitlines() if l.startswith("{")][0]))
    assert all(a["pooled_kept_samples"] == 8 for a in alone)
    want = 0.5 * (alone[0]["posterior_mean_varE"] + alone[1]["posterior_mean_varE"])
    assert abs(d["posterior_mean_varE"] - want) <= 1e-12 * abs(want)
