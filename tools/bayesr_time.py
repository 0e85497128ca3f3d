import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from ngp_pkg import load_pkg
ngp = load_pkg()
N, P = 10000, 100000
s = ngp.Sampler(device=0, seed=1001, chain=0)
s.generate_panel(N, P)
rng = np.random.default_rng(1); bt = np.zeros(P); idx = rng.choice(P, P // 100, replace=False); bt[idx] = rng.normal(size=len(idx))
g = s.xbeta(bt); y = 10 + g + np.random.default_rng(2).normal(size=N) * np.sqrt(g.var())
v = 0.5 * y.var() / (s.mpm().sum() / N)
s.add_marker_set_r(0, P, 4.0, v * 0.5, v, [0.0, 0.01, 0.1, 1.0], [0.95, 0.03, 0.015, 0.005], estPi=True)
s.set_y(y); s.set_residual_prior(4.0, 0.25 * y.var())
for k in range(8):
    t = time.perf_counter(); s.run(50); dt = (time.perf_counter() - t) / 50
    st = s.get_state(); cs = s.get_class_state(0)
    print(f"iterations {50*(k+1)}: {dt*1e3:.2f} ms/iter, {dt/1563*1e6:.2f} us/block, non-zero loci {np.mean(st['delta'] > 1)*100:.1f} %, pi {np.round(cs['piHat'], 4)}", flush=True)
