"""Long run of the headline shape: python tools/soak_c4.py iters [storage]; checks the residual invariant every 250 iterations."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ngp_pkg import load_pkg
ngp = load_pkg()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
storage = sys.argv[2] if len(sys.argv) > 2 else None
N, P = 50000, 600000
s = ngp.Sampler(device=0, seed=1001, chain=0, storage=storage)
s.generate_panel(N, P)
rng = np.random.default_rng(1); bt = np.zeros(P); idx = rng.choice(P, P // 100, replace=False); bt[idx] = rng.normal(size=len(idx))
g = s.xbeta(bt); y = 10 + g + np.random.default_rng(2).normal(size=N) * np.sqrt(g.var())
v = 0.5 * y.var() / (s.mpm().sum() / N)
for c in range(3):
    s.add_marker_set(c * 200000, 200000, 0, 4.0, v * 0.5, [(0, 200000)], [v])
s.set_y(y); s.set_residual_prior(4.0, 0.25 * y.var())
t0 = time.perf_counter(); done = 0
while done < iters:
    k = min(250, iters - done); s.run(k); done += k
    st = s.get_state()
    inv = float(np.abs(st["ycorr"] - (y - st["b"] - s.xbeta(st["beta"]))).max())
    print(f"{storage or 'f32'}: {done} iterations, {time.perf_counter() - t0:.1f} s, invariant {inv:.2e}, varE {st['varE']:.3f}", flush=True)
    assert inv < 1e-8 * np.abs(y).max() and np.isfinite(st["varE"])
print("soak ok")
