import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ngp_pkg import load_pkg
ngp = load_pkg()
lag = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N, P = 10000, 100000
s = ngp.Sampler(device=0, seed=1001, chain=0, mode=1, lag=lag)
s.generate_panel(N, P)
rng = np.random.default_rng(1); bt = np.zeros(P); idx = rng.choice(P, P // 100, replace=False); bt[idx] = rng.normal(size=P // 100)
g = s.xbeta(bt); y = 10 + g + np.random.default_rng(2).normal(size=N) * np.sqrt(g.var())
v = 0.5 * y.var() / (s.mpm().sum() / N)
s.add_marker_set(0, P, 0, 4.0, v * 0.5, [(0, P)], [v]); s.set_y(y); s.set_residual_prior(4.0, 0.25 * y.var())
s.run(3); s.debug_stamps(True); s.run(1)
nb = s.layout()[2]
base = (7 << 17) + 16384
d = s.debug_stamps(True, n=base + 4 * nb).astype(np.int64)
W = d[base:base + 4 * nb].reshape(nb, 4)[200:1400]
S0 = d[:4 * nb].reshape(nb, 4)[200:1400, 0]
print("w3 block-start -> begin wait us:", np.median(W[:, 0] - S0) / 100)
print("w3 wait for rows (issued one block ago) us:", np.median(W[:, 1] - W[:, 0]) / 100, "p90", np.percentile(W[:, 1] - W[:, 0], 90) / 100)
print("w3 scale+LDS write us:", np.median(W[:, 2] - W[:, 1]) / 100)
print("w3 issue 65 loads us:", np.median(W[:, 3] - W[:, 2]) / 100)
print("rows: issue(u-1) -> arrival(u) us:", np.median(W[1:, 1] - W[:-1, 2]) / 100, "p10", np.percentile(W[1:, 1] - W[:-1, 2], 10) / 100)
print("period us:", np.median(np.diff(S0)) / 100)
