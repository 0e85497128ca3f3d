"""Full-size checks on the GPU (BASELINE.json configs[1]/[2] shapes): the CPU oracle cannot run these sizes in
seconds, so parity is carried by size-independent properties of the sampler."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N, P = 10000, 100000


def _chain(ngp, method, niter, engine=(1, 6), seed=1001, P_=P):
    s = ngp.Sampler(device=0, seed=seed, chain=0, mode=engine[0], lag=engine[1])
    s.generate_panel(N, P_)
    rng = np.random.default_rng(1)
    bt = np.zeros(P_); idx = rng.choice(P_, P_ // 100, replace=False); bt[idx] = rng.normal(size=len(idx))
    g = s.xbeta(bt)
    y = 10.0 + g + np.random.default_rng(2).normal(size=N) * np.sqrt(g.var())
    v = 0.5 * y.var() / (s.mpm().sum() / N)
    if method == "PR":
        s.add_marker_set(0, P_, 0, 4.0, v * 0.5, [(0, P_)], [v])
    elif method == "B":
        s.add_marker_set(0, P_, 1, 4.0, v * 0.5, [(j, j + 1) for j in range(P_)], np.full(P_, v), pi0=0.01, estPi=True)
    elif method == "C":
        s.add_marker_set(0, P_, 2, 4.0, v * 0.5, [(0, P_)], [v], pi0=0.01, estPi=True)
    else:  # three consecutive sets, like the multi-breed configuration
        third = P_ // 3
        s.add_marker_set(0, third, 0, 4.0, v * 0.5, [(0, third)], [v])
        s.add_marker_set(third, third, 1, 4.0, v * 0.5, [(j, j + 1) for j in range(third)], np.full(third, v), pi0=0.02, estPi=True)
        s.add_marker_set(2 * third, P_ - 2 * third, 0, 4.0, v * 0.5, [(0, P_ - 2 * third)], [v])
    s.set_y(y); s.set_residual_prior(4.0, 0.25 * y.var()); s.set_schedule(niter, 0, 1)
    s.run(niter)
    return s, y


@pytest.mark.parametrize("method", ["PR", "B", "C", "multi"])
def test_residual_invariant_and_indicator_consistency(ngp, O, method):
    s, y = _chain(ngp, method, 12)
    st = s.get_state()
    resid = y - st["b"] - s.xbeta(st["beta"])                       # ycorr recomputed from scratch
    assert np.abs(st["ycorr"] - resid).max() <= 1e-9 * np.abs(y).max()
    # and on the host, for 64 rows regenerated from the panel generator (no device arithmetic in the check)
    rr = np.random.default_rng(5)
    rows, cols = np.sort(rr.choice(N, 64, replace=False)), rr.choice(P, 64, replace=False)
    means = s.means()
    assert np.array_equal(means[cols], O.column_sums(cols, N) / float(N))
    Xr = (O.generate_rows(rows, P).astype(np.float64) - means[None, :]).astype(np.float32).astype(np.float64)
    assert np.abs(st["ycorr"][rows] - (y[rows] - st["b"] - Xr @ st["beta"])).max() <= 1e-9 * np.abs(y).max()
    assert np.isfinite(st["beta"]).all() and st["varE"] > 0 and st["iter"] == 12
    d = st["delta"]
    assert set(np.unique(d)) <= {0, 1}
    if method == "PR":
        assert d.min() == 1 and st["varBeta"][0] > 0
    if method == "B":                                               # functions.jl:184-186
        assert np.all(st["beta"][d == 0] == 0.0) and np.all(st["varBeta"][d == 0] == 0.0) and np.all(st["varBeta"][d == 1] > 0.0)
        assert 0.0 < st["piHat"][1] < 0.2 and abs(st["piHat"].sum() - 1.0) < 1e-15
    if method == "C":                                               # functions.jl:226-231: one variance, excluded effects are 0
        assert np.all(st["beta"][d == 0] == 0.0) and st["varBeta"].shape == (1,) and st["varBeta"][0] > 0.0
        assert 0.0 < st["piHat"][1] < 0.2 and abs(st["piHat"].sum() - 1.0) < 1e-15
    ps = s.get_posterior_sums()
    assert ps["nKept"] == 12 and np.all(ps["sum_delta"] <= 12) and np.all(ps["sum_beta2"] >= 0)


def test_bitwise_reproducible_and_seed_sensitive(ngp):
    a, _ = _chain(ngp, "B", 6)
    b, _ = _chain(ngp, "B", 6)
    c, _ = _chain(ngp, "B", 6, seed=1002)
    sa, sb, sc = a.get_state(), b.get_state(), c.get_state()
    for k in ("ycorr", "beta", "delta", "varBeta", "piHat"):
        assert np.array_equal(sa[k], sb[k]), k
    assert sa["varE"] == sb["varE"] and not np.array_equal(sa["beta"], sc["beta"])


def test_engines_draw_the_same_chain(ngp):
    """Per-block launches (lag 1) and the persistent sweep (lag 5): identical indicators, floats to 1e-9."""
    P_ = 20032
    a, _ = _chain(ngp, "multi", 8, engine=(0, 1), P_=P_)
    b, _ = _chain(ngp, "multi", 8, engine=(1, 6), P_=P_)
    sa, sb = a.get_state(), b.get_state()
    assert np.array_equal(sa["delta"], sb["delta"])
    assert np.abs(sa["beta"] - sb["beta"]).max() <= 1e-9 * max(1e-3, np.abs(sa["beta"]).max())
    assert np.abs(sa["ycorr"] - sb["ycorr"]).max() <= 1e-9 * np.abs(sa["ycorr"]).max()
    assert abs(sa["varE"] - sb["varE"]) <= 1e-9 * sa["varE"]


def test_xbeta_is_linear(ngp):
    s = ngp.Sampler(device=0, seed=1, chain=0)
    s.generate_panel(N, 4096)
    rng = np.random.default_rng(0)
    u, w = rng.normal(size=4096), rng.normal(size=4096)
    lhs = s.xbeta(2.0 * u - 3.0 * w)
    rhs = 2.0 * s.xbeta(u) - 3.0 * s.xbeta(w)
    assert np.abs(lhs - rhs).max() <= 1e-10 * np.abs(lhs).max()


@pytest.mark.parametrize("N_,P_,mode_,V_", [(63000, 700, 1, 1), (63232, 400, 1, 1), (63300, 300, 1, 2), (100000, 1000, 1, 2), (107520, 200, 1, 2),
                                            (107600, 200, 1, 3), (156576, 300, 1, 3), (156600, 200, 0, 1)],
                         ids=["R256", "R256_full", "two_shards", "100k", "two_shards_full", "three_shards", "three_shards_full", "fallback"])
def test_tallest_shards_and_fallback(ngp, N_, P_, mode_, V_):
    """The persistent sweep holds shards of at most 256 rows, one per streamer workgroup (247 streamers -> N <= 63232); above
    that a workgroup owns two shards of at most 224 rows (240 workgroups -> N <= 107520, lag 3), then three (233 workgroups ->
    N <= 156576, lag 2); beyond that the per-block engine takes over.  (A layout rule that preferred 4*odd rows once produced 260-row shards here and dropped update tasks.)"""
    def chain(**kw):
        s = ngp.Sampler(device=0, seed=5, chain=0, **kw)
        s.generate_panel(N_, P_)
        rng = np.random.default_rng(1)
        bt = np.zeros(P_); bt[rng.choice(P_, 10, replace=False)] = rng.normal(size=10)
        g = s.xbeta(bt)
        y = 3.0 + g + np.random.default_rng(2).normal(size=N_) * np.sqrt(g.var())
        v = 0.5 * y.var() / (s.mpm().sum() / N_)
        h = P_ // 2
        s.add_marker_set(0, h, 0, 4.0, v * 0.5, [(0, h)], [v])
        s.add_marker_set(h, P_ - h, 1, 4.0, v * 0.5, [(j, j + 1) for j in range(P_ - h)], np.full(P_ - h, v), pi0=0.1, estPi=True)
        s.set_y(y); s.set_residual_prior(4.0, 0.25 * y.var())
        return s, y
    s, y = chain()
    R, S, nblk = s.layout()
    assert s.config()[0] == mode_ and (R <= 256 if mode_ == 1 else True) and R * S >= N_
    if V_ > 1:
        assert R <= 224 and S % V_ == 0 and s.config() == (1, 5 - V_) and s.streamer() == (2, 7)
    s.run(6)
    st = s.get_state()
    assert np.abs(st["ycorr"] - (y - st["b"] - s.xbeta(st["beta"]))).max() < 1e-9
    if V_ > 1:
        assert s.census()["grid"] == 1 + (S + 31) // 32 + S // V_
        # the same chain in the per-block engine (another layout, another summation order): indicators identical, floats to 1e-9
        # (the residual invariant alone once passed with the sampler adding only 8 of these layouts' 15 group sums)
        r, _ = chain(mode=0, lag=1)
        assert r.config()[0] == 0
        r.run(6)
        sr = r.get_state()
        assert np.array_equal(st["delta"], sr["delta"])
        assert np.abs(st["beta"] - sr["beta"]).max() <= 1e-9 * max(1e-3, np.abs(sr["beta"]).max())
        assert abs(st["varE"] - sr["varE"]) <= 1e-9 * sr["varE"] and np.abs(st["varBeta"] - sr["varBeta"]).max() <= 1e-9 * np.abs(sr["varBeta"]).max()


def test_north_star_shape_50k_x_600k(ngp, O):
    """BASELINE.json configs[3] at full size (N = 50,000, P = 600,000 as three BayesPR sets; 120 GB of tiles, 64-bit tile
    offsets, 204-row shards, lag 6, the row-owning streamer): the oracle cannot run this, so the size-independent properties
    carry parity -- ycorr == y - 1 b - X beta recomputed from scratch, bitwise reproducibility of two chains with the same
    seed, sensitivity to the seed, and agreement (1e-9) of the row-owning streamer with the phase streamer, a different
    summation order of the same chain."""
    N_, P_ = 50000, 600000
    out = []
    for seed, streamer in ((1001, None), (1001, None), (1002, None), (1001, 1)):
        s = ngp.Sampler(device=0, seed=seed, chain=0, streamer=streamer)
        s.generate_panel(N_, P_)
        if not out:
            R, S, nblk = s.layout()
            assert (R, S, nblk) == (204, 246, 9375) and s.config() == (1, 6) and s.near() == 2 and s.streamer() == (2, 7)
            rng = np.random.default_rng(1)
            bt = np.zeros(P_); idx = rng.choice(P_, P_ // 100, replace=False); bt[idx] = rng.normal(size=len(idx))
            g = s.xbeta(bt)
            y = 10.0 + g + np.random.default_rng(2).normal(size=N_) * np.sqrt(g.var())
            v = 0.5 * y.var() / (s.mpm().sum() / N_)
        for c in range(3):
            s.add_marker_set(c * 200000, 200000, 0, 4.0, v * 0.5, [(0, 200000)], [v])
        s.set_y(y); s.set_residual_prior(4.0, 0.25 * y.var()); s.set_schedule(6, 0, 1)
        s.run(6)
        st = s.get_state()
        if len(out) == 0:
            resid = y - st["b"] - s.xbeta(st["beta"])
            assert np.abs(st["ycorr"] - resid).max() <= 1e-9 * np.abs(y).max()
            # the same invariant on the HOST for 64 random rows, without the device's own X beta: the rows' genotype codes are
            # regenerated from the counter-based panel generator (oracle), centred with the library's column means (64 of them
            # audited against the generator's integer column sums), rounded to fp32 as the tiles are, and y_i - b - x_i'beta formed
            # in numpy
            rr = np.random.default_rng(17)
            rows = np.sort(rr.choice(N_, 64, replace=False))
            means = s.means()
            cols = rr.choice(P_, 64, replace=False)
            assert np.array_equal(means[cols], O.column_sums(cols, N_) / float(N_))
            G = O.generate_rows(rows, P_)
            Xr = (G.astype(np.float64) - means[None, :]).astype(np.float32).astype(np.float64)
            host = y[rows] - st["b"] - Xr @ st["beta"]
            assert np.abs(st["ycorr"][rows] - host).max() <= 1e-9 * np.abs(y).max()
            assert np.isfinite(st["beta"]).all() and st["varE"] > 0 and st["iter"] == 6 and st["delta"].min() == 1 and np.all(st["varBeta"] > 0)
            assert s.get_posterior_sums()["nKept"] == 6
        out.append(st)
        s.close()
    a, b, c, d = out
    for k in ("ycorr", "beta", "varBeta"):
        assert np.array_equal(a[k], b[k]), k
        assert np.abs(a[k] - d[k]).max() <= 1e-9 * np.abs(a[k]).max(), k
    assert a["varE"] == b["varE"] and not np.array_equal(a["beta"], c["beta"]) and abs(a["varE"] - d["varE"]) <= 1e-9 * a["varE"]


@pytest.mark.parametrize("lag", [6, 8], ids=["lag6_default", "lag8"])
def test_default_engine_at_full_size(ngp, lag):
    """The production engine of short shards (lag 6 by default since round 4, lag 8 before) at 10k x 100k: residual invariant and
    equality with the per-block engine."""
    P_ = 20032
    a, y = _chain(ngp, "multi", 8, engine=(0, 1), P_=P_)
    b, _ = _chain(ngp, "multi", 8, engine=(1, lag), P_=P_)
    sa, sb = a.get_state(), b.get_state()
    assert b.config() == (1, lag)
    if lag == 6:   # what the library picks when the caller leaves the lag to it
        d = ngp.Sampler(device=0, seed=1, chain=0); d.generate_panel(N, 6400); assert d.config() == (1, 6) and d.streamer() == (1, 8)
    assert np.array_equal(sa["delta"], sb["delta"])
    assert np.abs(sa["beta"] - sb["beta"]).max() <= 1e-9 * max(1e-3, np.abs(sa["beta"]).max())
    c, y2 = _chain(ngp, "PR", 12, engine=(1, lag))
    st = c.get_state()
    assert np.abs(st["ycorr"] - (y2 - st["b"] - c.xbeta(st["beta"]))).max() <= 1e-9 * np.abs(y2).max()


def test_north_star_shape_in_compact_storage(ngp):
    """50k x 600k with one byte per genotype (30 GB instead of 120 GB): residual invariant, bitwise reproducibility, and the distance
    to the fp32-storage chain of the same seed on the same generated genotypes -- the fp32 rounding of the panel, nothing else."""
    N_, P_ = 50000, 600000
    out = []
    for storage in ("u8", "u8", None):
        s = ngp.Sampler(device=0, seed=1001, chain=0, storage=storage)
        s.generate_panel(N_, P_)
        if not out:
            R, S, nblk = s.layout()
            assert (R, S, nblk) == (208, 241, 9375) and s.config() == (1, 8) and s.streamer() == (3, 7) and s.storage() == 1
            rng = np.random.default_rng(1)
            bt = np.zeros(P_); idx = rng.choice(P_, P_ // 100, replace=False); bt[idx] = rng.normal(size=len(idx))
            g = s.xbeta(bt)
            y = 10.0 + g + np.random.default_rng(2).normal(size=N_) * np.sqrt(g.var())
            v = 0.5 * y.var() / (s.mpm().sum() / N_)
        for c in range(3):
            s.add_marker_set(c * 200000, 200000, 0, 4.0, v * 0.5, [(0, 200000)], [v])
        s.set_y(y); s.set_residual_prior(4.0, 0.25 * y.var()); s.set_schedule(6, 0, 1)
        s.run(6)
        st = s.get_state()
        if not out:
            resid = y - st["b"] - s.xbeta(st["beta"])
            assert np.abs(st["ycorr"] - resid).max() <= 1e-9 * np.abs(y).max()
            assert np.isfinite(st["beta"]).all() and st["varE"] > 0 and st["iter"] == 6
        out.append(st)
        s.close()
    a, b, c = out
    for k in ("ycorr", "beta", "varBeta"):
        assert np.array_equal(a[k], b[k]), k
    dev = np.abs(a["beta"] - c["beta"]).max() / np.abs(a["beta"]).max()
    assert 0 < dev < 1e-3, dev     # fp32 panel rounding (~6e-8 per element) through 6 iterations of 600k updates
    assert abs(a["varE"] - c["varE"]) < 1e-5 * a["varE"]


def test_compact_storage_takes_panels_the_fp32_sweep_cannot(ngp):
    """150,000 rows: above the 63k-row limit of the persistent sweep on fp32 tiles; 624-row shards, four update tasks per lane."""
    N_, P_ = 150000, 30016
    s = ngp.Sampler(device=0, seed=5, chain=0, storage="u8")
    s.generate_panel(N_, P_)
    R, S, nblk = s.layout()
    assert R % 16 == 0 and R > 448 and R * S >= N_ and s.config()[0] == 1
    rng = np.random.default_rng(1)
    bt = np.zeros(P_); idx = rng.choice(P_, 300, replace=False); bt[idx] = rng.normal(size=300)
    g = s.xbeta(bt)
    y = 10.0 + g + np.random.default_rng(2).normal(size=N_) * np.sqrt(g.var())
    v = 0.5 * y.var() / (s.mpm().sum() / N_)
    s.add_marker_set(0, P_, 0, 4.0, v * 0.5, [(0, P_)], [v])
    s.set_y(y); s.set_residual_prior(4.0, 0.25 * y.var())
    s.run(5)
    st = s.get_state()
    assert np.abs(st["ycorr"] - (y - st["b"] - s.xbeta(st["beta"]))).max() <= 1e-9 * np.abs(y).max()
    assert np.isfinite(st["beta"]).all() and st["varE"] > 0
    # the effects found are the simulated ones (a sampler that mixed up rows or columns would not correlate)
    assert np.corrcoef(st["beta"], bt)[0, 1] > 0.3
