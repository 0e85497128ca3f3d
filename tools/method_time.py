"""Sweep time per method on one generated panel: python tools/method_time.py N P [iters]  (PR, B, C, R with 4 / 8 classes)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ngp_pkg import load_pkg
ngp = load_pkg()
N, P = int(sys.argv[1]), int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 30
for kind in os.environ.get("NGP_TOOL_METHODS", "PR,B,C,R4,R8").split(","):
    s = ngp.Sampler(device=0, seed=1001, chain=0, lag=int(os.environ["NGP_TOOL_LAG"]) if "NGP_TOOL_LAG" in os.environ else None)
    if "NGP_TOOL_NEAR" in os.environ: s.set_near(int(os.environ["NGP_TOOL_NEAR"]))
    if "NGP_TOOL_CHAIN_FORM" in os.environ: s.set_chain_form(int(os.environ["NGP_TOOL_CHAIN_FORM"]))
    s.generate_panel(N, P)
    rng = np.random.default_rng(1); bt = np.zeros(P); idx = rng.choice(P, max(10, P // 100), replace=False); bt[idx] = rng.normal(size=len(idx))
    g = s.xbeta(bt); y = 10 + g + np.random.default_rng(2).normal(size=N) * np.sqrt(g.var())
    v = 0.5 * y.var() / (s.mpm().sum() / N)
    if kind == "PR": s.add_marker_set(0, P, 0, 4.0, v * 0.5, [(0, P)], [v])
    elif kind == "B": s.add_marker_set(0, P, 1, 4.0, v * 0.5, [(j, j + 1) for j in range(P)], np.full(P, v), pi0=0.01, estPi=True)
    elif kind == "C": s.add_marker_set(0, P, 2, 4.0, v * 0.5, [(0, P)], [v], pi0=0.01, estPi=True)
    elif kind == "R4": s.add_marker_set_r(0, P, 4.0, v * 0.5, v, [0.0, 0.01, 0.1, 1.0], [0.95, 0.03, 0.015, 0.005], estPi=True)
    else: s.add_marker_set_r(0, P, 4.0, v * 0.5, v, [0.0, 1e-4, 1e-3, 0.01, 0.05, 0.2, 0.5, 1.0], [0.9, 0.03, 0.02, 0.02, 0.01, 0.01, 0.005, 0.005], estPi=True)
    s.set_y(y); s.set_residual_prior(4.0, 0.25 * y.var())
    s.run(5)
    b0 = s.get_state()["beta"][:P] != 0
    t = time.perf_counter(); s.run(iters); dt = (time.perf_counter() - t) / iters
    R, S, nb = s.layout()
    b1 = s.get_state()["beta"][:P] != 0   # steps of a sparse block's chain: loci with an old or a new effect
    print(f"{kind:3s} N={N} P={P} layout R={R} S={S} lag={s.config()[1]} near={s.near()}: {dt * 1e3:.3f} ms/iter, {dt / nb * 1e6:.2f} us/block"
          f"   (non-zero effects {b0.mean() * 100:.2f} % before, {b1.mean() * 100:.2f} % after: {64 * b1.mean():.1f} per block)", flush=True)
