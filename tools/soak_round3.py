"""Soak run of the round-3 engines: tall fp32 panels (two / three shards per streamer workgroup) and eight chains per pass, long chains;
checks the residual invariant of every chain and that fused chain 0 and 7 equal the same chains run alone, bit for bit.
   python tools/soak_round3.py [iters]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ngp_pkg import load_pkg
ngp = load_pkg()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1000


def problem(s, N, P):
    rng = np.random.default_rng(1); bt = np.zeros(P); idx = rng.choice(P, max(10, P // 100), replace=False); bt[idx] = rng.normal(size=len(idx))
    g = s.xbeta(bt); y = 10 + g + np.random.default_rng(2).normal(size=N) * np.sqrt(g.var())
    return y, 0.5 * y.var() / (s.mpm().sum() / N)


def add(s, P, v, kind):
    h = P // 2
    s.add_marker_set(0, h, 0, 4.0, v * 0.5, [(0, h)], [v])
    if kind == "B": s.add_marker_set(h, P - h, 1, 4.0, v * 0.5, [(j, j + 1) for j in range(P - h)], np.full(P - h, v), pi0=0.02, estPi=True)
    else: s.add_marker_set(h, P - h, 3 if False else 2, 4.0, v * 0.5, [(0, P - h)], [v], pi0=0.02, estPi=True)


for N, P, kind in ((70000, 30000, "B"), (120000, 20000, "C")):
    s = ngp.Sampler(device=0, seed=77, chain=0)
    s.generate_panel(N, P)
    y, v = problem(s, N, P); add(s, P, v, kind); s.set_y(y); s.set_residual_prior(4.0, 0.25 * y.var())
    t0 = time.perf_counter(); done = 0
    while done < iters:
        k = min(250, iters - done); s.run(k); done += k
        print(f"  tall N={N} P={P} layout {s.layout()} lag={s.config()[1]}: {done} iterations, {time.perf_counter() - t0:.1f} s", flush=True)
    st = s.get_state()
    inv = float(np.abs(st["ycorr"] - (y - st["b"] - s.xbeta(st["beta"]))).max())
    assert inv < 1e-8 and np.isfinite(st["beta"]).all() and st["varE"] > 0, inv
    print(f"tall N={N}: invariant {inv:.2e}, varE {st['varE']:.4f}, census retries {s.census()['retries']}", flush=True)
    del s

N, P, K = 10000, 50000, 8
chains = []
for k in range(K):
    s = ngp.Sampler(device=0, seed=1001 + k, chain=k, mode=1, lag=8)
    if k == 0:
        s.set_max_shards(s.shards_for_pass(K)); s.generate_panel(N, P); y, v = problem(s, N, P)
    else:
        s.share_panel(chains[0])
    add(s, P, v, "B"); s.set_y(y + 0.01 * k); s.set_residual_prior(4.0, 0.25 * y.var()); chains.append(s)
t0 = time.perf_counter(); done = 0
while done < iters:
    k = min(250, iters - done); ngp.Sampler.run_many(chains, k); done += k
    print(f"  fused K={K} N={N} P={P} layout {chains[0].layout()} grid {chains[0].census()['grid']}: {done} iterations, {time.perf_counter() - t0:.1f} s", flush=True)
R, S, _ = chains[0].layout()
for c in (0, K - 1):
    st = chains[c].get_state()
    inv = float(np.abs(st["ycorr"] - (y + 0.01 * c - st["b"] - chains[c].xbeta(st["beta"]))).max())
    a = ngp.Sampler(device=0, seed=1001 + c, chain=c, mode=1, lag=8)
    a.set_max_shards(S); a.generate_panel(N, P); assert a.layout()[:2] == (R, S)
    add(a, P, v, "B"); a.set_y(y + 0.01 * c); a.set_residual_prior(4.0, 0.25 * y.var()); a.run(iters)
    sa = a.get_state()
    same = all(np.array_equal(st[k], sa[k]) for k in ("ycorr", "beta", "delta", "varBeta", "piHat")) and st["varE"] == sa["varE"]
    print(f"fused chain {c}: invariant {inv:.2e}, identical to the chain alone after {iters} iterations: {same}", flush=True)
    assert inv < 1e-8 and same
print("soak ok")
