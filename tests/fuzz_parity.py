"""Randomised parity: random shapes, method mixes, lags, near lags, streamers, storages and shard limits, the device against the
blocked oracle with the layout the library reports, bit for bit.  python tests/fuzz_parity.py [cases] [seed]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from ngp_pkg import load_pkg
from oracle import oracle as O
from conftest import add_sets
ngp = load_pkg()
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
KINDS = ["PR", "B", "Bfix", "C", "Cfix", "PR1", "R", "Rfix", "R2", "R6", "R8", "R12", "R16", ("PRw", 37)]
bad = 0
for case in range(ncases):
    N = int(rng.choice([7, 33, 100, 257, 500, 900, 1500, 2600, 4000]))
    P = int(rng.choice([3, 64, 65, 130, 200, 333, 640, 1000]))
    storage = rng.choice(["f32", "u8"])
    lag = int(rng.choice([1, 2, 3, 4, 5, 6, 8] if storage == "f32" else [3, 4, 6, 8, 12]))
    near = int(rng.choice([0, 1, 2, 3, 4]))
    streamer = int(rng.choice([0, 1, 2, 4, 6])) if storage == "f32" else 0   # 4 / 6: two / three shards per streamer workgroup (lag 3 / 2)
    shards = int(rng.choice([0, 0, 3, 17, 60]))
    X, mu = O.generate_panel(N, P, seed=int(rng.integers(1, 1 << 30)))
    G = np.rint(X.astype(np.float64) + mu[None, :]).astype(np.uint8)
    bt = np.zeros(P); idx = rng.choice(P, min(5, P), replace=False); bt[idx] = rng.normal(size=len(idx))
    y = 5.0 + X.astype(np.float64) @ bt + rng.normal(size=N)
    v = 0.02
    # random consecutive sets
    cuts = sorted(set([0, P] + [int(c) for c in rng.choice(np.arange(1, P), size=min(int(rng.integers(0, 3)), max(P - 1, 0)), replace=False)])) if P > 1 else [0, P]
    spec = [(a, b - a, KINDS[int(rng.integers(0, len(KINDS)))]) for a, b in zip(cuts[:-1], cuts[1:])]
    s = ngp.Sampler(device=0, seed=int(rng.integers(1, 1 << 30)), chain=int(rng.integers(0, 8)), mode=1, lag=lag, storage=storage)
    seed, chain = None, None
    try:
        if near: s.set_near(near)
        if streamer: s.set_streamer(streamer)
        if shards: s.set_max_shards(shards)
        s.set_panel(G if storage == "u8" else X, centre=(storage == "u8"))
    except ngp.NextGPHipError as e:
        print(f"case {case}: skipped ({str(e)[:80]})"); continue
    R, S, nblk = s.layout(); mode, D = s.config(); variant, nchain = s.streamer()
    # the oracle takes seed / chain from the sampler's construction arguments: rebuild them
    o = None
    desc = f"N={N} P={P} {storage} lag={D} near={s.near()} streamer={variant} R={R} S={S} sets={[k if isinstance(k, str) else k[0] for _, _, k in spec]}"
    try:
        # same seed/chain: re-create the sampler with known values
        s.close()
        sd, ch = int(rng.integers(1, 1 << 30)), int(rng.integers(0, 8))
        s = ngp.Sampler(device=0, seed=sd, chain=ch, mode=1, lag=lag, storage=storage)
        if near: s.set_near(near)
        if streamer: s.set_streamer(streamer)
        if shards: s.set_max_shards(shards)
        s.set_panel(G if storage == "u8" else X, centre=(storage == "u8"))
        o = O.Oracle(order=1, seed=sd, chain=ch)
        if storage == "u8": o.set_panel_u8(G, R=R, S=S, D=D, near=s.near(), tform=s.chain_form())
        else: o.set_panel_f32(X, R=R, S=S, D=D, near=s.near(), nchain=nchain, tform=s.chain_form())
        niter = int(rng.integers(3, 9))
        for m in (s, o):
            add_sets(m, spec, v); m.set_y(y); m.set_residual_prior(4.0, 0.5); m.set_schedule(niter, 1, 2); m.run(niter)
        a, b = s.get_state(), o.get_state()
        ok = np.array_equal(a["delta"], b["delta"]) and all(np.array_equal(a[k], b[k]) for k in ("ycorr", "beta", "varBeta", "piHat")) and a["varE"] == b["varE"] and a["b"] == b["b"]
        pa, pb = s.get_posterior_sums(), o.get_posterior_sums()
        ok = ok and all(np.array_equal(pa[k], pb[k]) for k in ("sum_beta", "sum_beta2", "sum_delta", "sum_varBeta", "sum_pi"))
    except Exception as e:  # noqa: BLE001
        ok = False; desc += f"  EXC {type(e).__name__}: {str(e)[:120]}"
    print(("ok   " if ok else "FAIL ") + desc, flush=True)
    bad += (not ok)
    s.close()
print(f"fuzz: {ncases} cases, {bad} failures")
sys.exit(1 if bad else 0)
