import sys, os, numpy as np
sys.path.insert(0, "/root/repo")
from ngp_pkg import load_pkg
ngp = load_pkg()
for N, P in ((63000, 1300), (63232, 700), (63300, 400), (100000, 300)):
    s = ngp.Sampler(device=0, seed=5, chain=0)
    s.generate_panel(N, P)
    rng = np.random.default_rng(1); bt = np.zeros(P); bt[rng.choice(P, 10, replace=False)] = rng.normal(size=10)
    g = s.xbeta(bt); y = 3 + g + np.random.default_rng(2).normal(size=N) * np.sqrt(g.var())
    v = 0.5 * y.var() / (s.mpm().sum() / N)
    h = P // 2
    s.add_marker_set(0, h, 0, 4.0, v * 0.5, [(0, h)], [v]); s.add_marker_set(h, P - h, 1, 4.0, v * 0.5, [(j, j + 1) for j in range(P - h)], np.full(P - h, v), pi0=0.1, estPi=True)
    s.set_y(y); s.set_residual_prior(4.0, 0.25 * y.var()); s.run(8)
    st = s.get_state()
    inv = np.abs(st["ycorr"] - (y - st["b"] - s.xbeta(st["beta"]))).max()
    print(N, P, "layout", s.layout(), "config", s.config(), "near", s.near(), "invariant %.2e" % inv, "varE %.4f" % st["varE"], flush=True)
    assert inv < 1e-9
print("edge ok")
