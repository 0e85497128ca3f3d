"""K independent chains in ONE fused sweep launch per iteration (ngp_share_panel + ngp_run_many): python tools/chains_per_pass.py N P K iters [lag]
Prints the aggregate Gibbs iterations/s and the time of a pass over the panel.  Profiling: rocprofv3 --kernel-trace --stats / --pmc FETCH_SIZE
on this script show one k_sweep_multi launch per iteration whose fetched bytes are ONE panel (4 N P), whatever K."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ngp_pkg import load_pkg
ngp = load_pkg()
N, P, K, iters = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
lag = int(sys.argv[5]) if len(sys.argv) > 5 else 0
method = os.environ.get("NGP_TOOL_METHOD", "PR")
chains = []
for k in range(K):
    s = ngp.Sampler(device=0, seed=1001 + k, chain=k, storage=os.environ.get("NGP_TOOL_STORAGE"), **({"mode": 1, "lag": lag} if lag else {}))
    if "NGP_TOOL_KNOB" in os.environ: s.debug_set_knob(int(os.environ["NGP_TOOL_KNOB"]))  # 2048 / 4096: reducers serve two chains each / one
    if k == 0:
        if K > 1:
            s.set_max_shards(int(os.environ.get("NGP_TOOL_SHARDS", "0")) or s.shards_for_pass(K))
        s.generate_panel(N, P)
        rng = np.random.default_rng(1); bt = np.zeros(P); idx = rng.choice(P, max(10, P // 100), replace=False); bt[idx] = rng.normal(size=len(idx))
        g = s.xbeta(bt); y = 10 + g + np.random.default_rng(2).normal(size=N) * np.sqrt(g.var())
        v = 0.5 * y.var() / (s.mpm().sum() / N)
    else:
        s.share_panel(chains[0])
    if method == "R":
        s.add_marker_set_r(0, P, 4.0, v * 0.5, v, [0.0, 0.01, 0.1, 1.0], [0.95, 0.03, 0.015, 0.005], estPi=True)
    elif method == "B":
        s.add_marker_set(0, P, 1, 4.0, v * 0.5, [(j, j + 1) for j in range(P)], np.full(P, v), pi0=0.01, estPi=True)
    else:
        s.add_marker_set(0, P, 0, 4.0, v * 0.5, [(0, P)], [v])
    s.set_y(y); s.set_residual_prior(4.0, 0.25 * y.var())
    chains.append(s)
run = (lambda n: ngp.Sampler.run_many(chains, n)) if K > 1 else (lambda n: chains[0].run(n))
run(3)
chains[0].get_timing()
t0 = time.perf_counter(); run(iters); dt = time.perf_counter() - t0
tm = chains[0].get_timing()
R, S, nblk = chains[0].layout()
bpe = 1.0 if os.environ.get("NGP_TOOL_STORAGE") == "u8" else 4.0
print(f"N={N} P={P} method={method} chains per pass={K} layout R={R} S={S} lag={chains[0].config()[1]} grid={chains[0].census()['grid']}: "
      f"{K * iters / dt:.1f} it/s aggregate, {dt / iters * 1e3:.3f} ms per pass ({dt / iters / nblk * 1e6:.3f} us per 64-SNP block), "
      f"device {tm['iter_ms'] / max(tm['iters'], 1):.3f} ms, sweep launches {tm['sweep_launches']}, panel stream {bpe * N * P * iters / dt / 1e12:.2f} TB/s", flush=True)
for s in chains:
    st = s.get_state()
    assert np.isfinite(st["beta"]).all() and st["varE"] > 0
