"""Soak run: long chains on several shapes / lags / methods; checks the residual invariant and that no hand-off was lost.
   python tools/soak.py [iters]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ngp_pkg import load_pkg
ngp = load_pkg()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
storage = os.environ.get("NGP_TOOL_STORAGE")   # "u8": the same soak in compact storage (lags are rounded to the instantiated ones)
cases = [(10000, 100000, 6, [("PR", 100000)]), (10000, 100000, 8, [("B", 100000)]), (3000, 40000, 3, [("C", 20000), ("PR", 20000)]),
         (20000, 60000, 6, [("PR", 30000), ("B", 30000)]), (777, 9999, 4, [("B", 5000), ("C", 4999)]), (50000, 40000, 5, [("PR", 40000)])]
for N, P, lag, sets in cases:
    s = ngp.Sampler(device=0, seed=77, chain=0, mode=1, lag=lag, storage=storage)
    s.generate_panel(N, P)
    rng = np.random.default_rng(1); bt = np.zeros(P); idx = rng.choice(P, max(10, P // 100), replace=False); bt[idx] = rng.normal(size=len(idx))
    g = s.xbeta(bt); y = 10 + g + np.random.default_rng(2).normal(size=N) * np.sqrt(g.var())
    v = 0.5 * y.var() / (s.mpm().sum() / N)
    c0 = 0
    for kind, n in sets:
        if kind == "PR": s.add_marker_set(c0, n, 0, 4.0, v * 0.5, [(0, n)], [v])
        elif kind == "C": s.add_marker_set(c0, n, 2, 4.0, v * 0.5, [(0, n)], [v], pi0=0.02, estPi=True)
        else: s.add_marker_set(c0, n, 1, 4.0, v * 0.5, [(j, j + 1) for j in range(n)], np.full(n, v), pi0=0.02, estPi=True)
        c0 += n
    s.set_y(y); s.set_residual_prior(4.0, 0.25 * y.var())
    t0 = time.perf_counter(); done = 0
    while done < iters:
        k = min(500, iters - done); s.run(k); done += k
        print(f"  N={N} P={P} lag={s.config()[1]} {[k for k, _ in sets]}: {done} iterations, {time.perf_counter() - t0:.1f} s", flush=True)
    st = s.get_state()
    inv = float(np.abs(st["ycorr"] - (y - st["b"] - s.xbeta(st["beta"]))).max())
    assert np.isfinite(st["varE"]) and np.all(np.isfinite(st["beta"])) and inv < 1e-8 * max(1.0, np.abs(y).max()), (inv, st["varE"])
    print(f"ok N={N} P={P}: invariant {inv:.2e}, varE {st['varE']:.4f}, included {int(st['delta'].sum())}", flush=True)
print("soak passed")
