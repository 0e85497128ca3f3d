"""Runs one of the BASELINE.json configurations end to end and prints a JSON line (diagnostic / report tool).
   python tests/run_config.py C1|C2|C3|C3c|C4|C4s [iters] [warm]   (C3c = C3 with BayesC)"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ngp_pkg import load_pkg
ngp = load_pkg()
cfg = sys.argv[1]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 5
N, P, sets = {"C1": (500, 5000, [("PR", 5000)]), "C2": (10000, 100000, [("PR", 100000)]), "C3": (10000, 100000, [("B", 100000)]), "C3c": (10000, 100000, [("C", 100000)]),
              "C4": (50000, 600000, [("PR", 200000)] * 3), "C4s": (50000, 60000, [("PR", 20000)] * 3)}[cfg]
s = ngp.Sampler(device=0, seed=1001, chain=0)
t0 = time.perf_counter(); s.generate_panel(N, P); setup = time.perf_counter() - t0
rng = np.random.default_rng(1); bt = np.zeros(P); idx = rng.choice(P, max(10, P // 100), replace=False); bt[idx] = rng.normal(size=len(idx))
t0 = time.perf_counter(); g = s.xbeta(bt); txb = time.perf_counter() - t0
y = 10 + g + np.random.default_rng(2).normal(size=N) * np.sqrt(g.var())
v = 0.5 * y.var() / (s.mpm().sum() / N)
c0 = 0
for kind, n in sets:
    if kind == "PR": s.add_marker_set(c0, n, 0, 4.0, v * 0.5, [(0, n)], [v])
    elif kind == "C": s.add_marker_set(c0, n, 2, 4.0, v * 0.5, [(0, n)], [v], pi0=0.01, estPi=True)
    else: s.add_marker_set(c0, n, 1, 4.0, v * 0.5, [(j, j + 1) for j in range(n)], np.full(n, v), pi0=0.01, estPi=True)
    c0 += n
s.set_y(y); s.set_residual_prior(4.0, 0.25 * y.var()); s.set_schedule(warm + iters, warm, 1)
s.run(warm)
t0 = time.perf_counter(); s.run(iters); dt = (time.perf_counter() - t0) / iters
st = s.get_state(); ps = s.get_posterior_sums()
resid = y - st["b"] - s.xbeta(st["beta"])
R, S, nb = s.layout(); mode, lag = s.config()
corr = float(np.corrcoef(ps["sum_beta"], bt)[0, 1])


def ess_geyer(x):
    """Effective sample size by Geyer's initial positive sequence estimator (SURVEY.md section 8 d)."""
    x = np.asarray(x, float) - np.mean(x)
    n = len(x)
    if n < 8 or not np.any(x):
        return float(n)
    f = np.fft.rfft(x, 2 * n)
    acf = np.fft.irfft(f * np.conj(f))[:n] / (x @ x)
    tau, k = -1.0, 0
    while k + 1 < n:
        pair = acf[k] + acf[k + 1]
        if pair <= 0:
            break
        tau += 2.0 * pair
        k += 2
    return float(n / max(tau, 1e-12))


cpu = None
if cfg == "C1":  # the reference's own CPU-runnable case: the reference-order oracle timed beside it (same model, same seeds)
    from oracle import oracle as O
    X, mu = O.generate_panel(N, P)
    o = O.Oracle(0, seed=1001, chain=0); o.set_panel_f32(X)
    o.add_marker_set(0, P, 0, 4.0, v * 0.5, [(0, P)], [v]); o.set_y(y); o.set_residual_prior(4.0, 0.25 * y.var())
    o.run(warm)
    t0 = time.perf_counter(); o.run(iters); cdt = (time.perf_counter() - t0) / iters
    so = o.get_state()
    otr = o.get_trace(iters)   # (the last ora_run's trace: the timed iterations)
    oess = ess_geyer(otr["varE"])
    cpu = dict(ms_per_iter=cdt * 1e3, it_per_s=1 / cdt, cores=1, kind="reference-order C port",
               max_abs_dbeta_vs_gpu=float(np.abs(so["beta"] - st["beta"]).max()), varE=so["varE"],
               ess_varE=oess, ess_varE_per_iteration=oess / iters, ess_varE_per_s=oess / (cdt * iters))
tr = s.get_trace(iters)
ess_varE = ess_geyer(tr["varE"])
print(json.dumps(dict(config=cfg, N=N, P=P, sets=[k for k, _ in sets], mode=mode, lag=lag, R=R, S=S, nblk=nb, ms_per_iter=dt * 1e3, it_per_s=1 / dt,
                      GBs=4.0 * N * P / dt / 1e9, frac_of_8TBs=4.0 * N * P / dt / 8e12, setup_s=setup, xbeta_s=txb,
                      invariant_max=float(np.abs(st["ycorr"] - resid).max()), varE=st["varE"], piHat=list(map(float, st["piHat"])),
                      included=int(st["delta"].sum()), corr_postmean_true=corr,
                      ess_varE=ess_varE, ess_varE_per_iteration=ess_varE / iters, ess_varE_per_s=ess_varE / (dt * iters), cpu_oracle=cpu)))
