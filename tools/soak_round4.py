"""Soak run of the round-4 engines: long chains through the fused kernels added this round -- three chains over byte tiles, eight chains with
a correlated (Tuple) set, a BayesR chain with the lazy class search (four and twelve classes) -- checking every chain's residual invariant
ycorr = y - b - X beta, the census (no launch had to be run again) and that a fused chain equals the same chain run alone, bit for bit.
   python tools/soak_round4.py [iters]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ngp_pkg import load_pkg
ngp = load_pkg()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1000


def problem(s, N, P):
    rng = np.random.default_rng(1); bt = np.zeros(P); idx = rng.choice(P, max(10, P // 100), replace=False); bt[idx] = rng.normal(size=len(idx))
    g = s.xbeta(bt); y = 10 + g + np.random.default_rng(2).normal(size=N) * np.sqrt(g.var())
    return y, 0.5 * y.var() / (s.mpm().sum() / N)


def invariant(s, y):
    st = s.get_state()
    return np.abs(st["ycorr"] - (y - st["b"] - s.xbeta(st["beta"]))).max() / np.abs(y).max()


def fused(name, N, P, K, storage, model):
    chains = []
    for c in range(K):
        s = ngp.Sampler(device=0, seed=500 + c, chain=c, storage=storage)
        if c == 0:
            s.set_max_shards(s.shards_for_pass(K)); s.generate_panel(N, P); y, v = problem(s, N, P)
        else:
            s.share_panel(chains[0])
        model(s, P, v); s.set_y(y + 0.01 * c); s.set_residual_prior(4.0, 0.25 * y.var()); chains.append(s)
    t0 = time.perf_counter(); ngp.Sampler.run_many(chains, iters); dt = time.perf_counter() - t0
    cen = chains[0].census(); tm = chains[0].get_timing()
    inv = max(invariant(s, y + 0.01 * c) for c, s in enumerate(chains))
    last = K - 1
    alone = ngp.Sampler(device=0, seed=500 + last, chain=last, storage=storage)
    alone.set_max_shards(chains[0].layout()[1]); alone.generate_panel(N, P); model(alone, P, v); alone.set_y(y + 0.01 * last)
    alone.set_residual_prior(4.0, 0.25 * y.var()); alone.run(iters)
    same = all(np.array_equal(alone.get_state()[k], chains[last].get_state()[k]) for k in ("beta", "ycorr", "varBeta"))
    print(f"{name}: {K} chains x {iters} iterations in {dt:.1f} s ({K * iters / dt:.0f} it/s), fused launches {tm['sweep_launches']} of {iters}, "
          f"census retries {cen['retries']}, invariant {inv:.1e}, chain {last} == the chain alone: {same}", flush=True)
    assert tm["sweep_launches"] == iters and cen["retries"] == 0 and inv < 1e-9 and same
    for s in chains[::-1] + [alone]: s.close()


pr = lambda s, P, v: s.add_marker_set(0, P, 0, 4.0, v * 0.5, [(0, P)], [v])
def tup(s, P, v):
    s.set_chain_form(1); nloc = (P // 64) * 32
    V = v * (0.7 * np.eye(2) + 0.3); s.add_marker_set_tuple(0, nloc, 2, 5.0, V * 0.5, [(0, nloc)], V)
fused("three chains over byte tiles, 30k x 60k", 30000, 60000, 3, "u8", pr)
fused("eight chains with a Tuple set, 10k x 50k", 10000, 50016, 8, None, tup)
for K, vc, pi in ((4, [0.0, 0.01, 0.1, 1.0], [0.95, 0.03, 0.015, 0.005]),
                  (12, [0.0, 1e-5, 3e-5, 1e-4, 3e-4, 1e-3, 3e-3, 0.01, 0.03, 0.1, 0.3, 1.0], [0.9] + [0.1 / 11] * 11)):
    s = ngp.Sampler(device=0, seed=900, chain=0); s.generate_panel(10000, 50000); y, v = problem(s, 10000, 50000)
    s.add_marker_set_r(0, 50000, 4.0, v * 0.5, v, vc, pi, estPi=True); s.set_y(y); s.set_residual_prior(4.0, 0.25 * y.var())
    t0 = time.perf_counter(); s.run(iters); dt = time.perf_counter() - t0
    st = s.get_state(); inv = invariant(s, y)
    print(f"BayesR {K} classes, 10k x 50k: {iters} iterations in {dt:.1f} s, invariant {inv:.1e}, non-zero effects {(st['beta'] != 0).mean() * 100:.2f} %, "
          f"class probabilities {np.round(s.get_class_state(0)['piHat'][:4], 4)}, census retries {s.census()['retries']}", flush=True)
    assert inv < 1e-9 and np.isfinite(st["beta"]).all() and s.census()["retries"] == 0
    s.close()
print("soak ok")
