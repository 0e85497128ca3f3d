"""Times the sweep with a correlated (Tuple BayesPR) marker set beside the same panel as Symbol sets: python tools/tuple_time.py N P k [iters]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ngp_pkg import load_pkg
ngp = load_pkg()
N, P, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 30
for mode in ("symbol", "tuple"):
    s = ngp.Sampler(device=0, seed=1001, chain=0)
    if "NGP_TOOL_CHAIN_FORM" in os.environ: s.set_chain_form(int(os.environ["NGP_TOOL_CHAIN_FORM"]))
    s.generate_panel(N, P)
    rng = np.random.default_rng(1); bt = np.zeros(P); idx = rng.choice(P, max(10, P // 100), replace=False); bt[idx] = rng.normal(size=len(idx))
    g = s.xbeta(bt); y = 10 + g + np.random.default_rng(2).normal(size=N) * np.sqrt(g.var())
    v = 0.5 * y.var() / (s.mpm().sum() / N)
    if mode == "symbol":
        s.add_marker_set(0, P, 0, 4.0, v * 0.5, [(0, P)], [v])
    else:
        Lb = 64 // k
        nloc = (P // 64) * Lb                      # whole blocks: every block holds Lb loci of k columns
        V = v * (0.7 * np.eye(k) + 0.3)
        s.add_marker_set_tuple(0, nloc, k, 3.0 + k, V * 0.5, [(0, nloc)], V)
    s.set_y(y); s.set_residual_prior(4.0, 0.25 * y.var())
    s.run(3)
    t = time.perf_counter(); s.run(iters); dt = (time.perf_counter() - t) / iters
    R, S, nb = s.layout()
    st = s.get_state()
    inv = np.abs(st["ycorr"] - (y - st["b"] - s.xbeta(st["beta"]))).max()
    print(f"{mode:6s} N={N} P={P} k={k} layout R={R} S={S} lag={s.config()[1]}: {dt * 1e3:.3f} ms/iter, {dt / nb * 1e6:.2f} us/block, invariant {inv:.1e}, varE {st['varE']:.3f}", flush=True)
