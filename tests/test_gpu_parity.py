"""GPU parity tests proper: the HIP path, called through the C ABI, against the CPU oracle.

Bar (north_star): bit-exact for indices / indicators; floating point: bit-exact against the
blocked-order oracle (same summation trees), and |diff| <= 1e-9 * scale against the
reference-order oracle (different summation order of the same chain)."""
import numpy as np
import pytest

from conftest import add_sets, make_problem

pytestmark = pytest.mark.gpu

KINDS = dict(VARE_CHI2=1, FIXED_NORMAL=2, BETA_NORMAL=3, REGION_CHI2=4, B_UNIFORM=5, B_LOCUS_CHI2=6, PI_BETA=7)


@pytest.fixture(scope="module")
def dev(ngp):
    return ngp.Sampler(device=0, seed=1234, chain=3)


def test_math_bit_exact(dev, O):
    rng = np.random.default_rng(0)
    x = np.concatenate([10.0 ** rng.uniform(-300, 300, 20000), rng.uniform(0.5, 2.0, 20000), [1.0, 2.0, 0.5, 1e-310]])
    got = dev.eval_math(0, x)
    exp = np.array([O.det_log(v) for v in x])
    assert np.array_equal(got, exp)
    p = np.concatenate([rng.uniform(0, 1, 40000), [1e-300, 1e-20, 0.075, 0.925, 0.5, 1 - 2.0 ** -53]])
    assert np.array_equal(dev.eval_math(1, p), np.array([O.ppnd16(v) for v in p]))
    assert np.array_equal(dev.eval_math(2, x), np.sqrt(x))        # IEEE sqrt
    assert np.array_equal(dev.eval_math(3, x), 1.0 / x)           # IEEE divide


@pytest.mark.parametrize("what,p1,p2", [(0, 0, 0), (1, 0, 0), (2, 5.0, 0), (2, 10004.0, 0), (3, 3.0, 98.0), (4, 1.0, 0)])
def test_draws_bit_exact(dev, O, what, p1, p2):
    n = 5000
    got = dev.draws_indexed(17, KINDS["BETA_NORMAL"], (2 << 40) | 5, what, n, p1, p2)
    exp = O.draws(1234, 3, 17, KINDS["BETA_NORMAL"], (2 << 40) | 5, what, n, p1, p2, indexed=True)
    assert np.array_equal(got, exp)


# (mode, lag[, near lags]): per-block launches / persistent sweep; near = 4 is what tall shards (R > 128) run with
# (mode, lag[, near lags[, streamer variant]]): per-block launches / persistent sweep; near = 4 is what tall shards (R > 128) run
# with; streamer 2 = row-owning waves + loader wave (7 GEMV chains; the default for shards of 132..224 rows, forced here)
ENGINES = [(0, 1), (1, 1), (1, 2), (1, 3), (1, 4), (1, 6), (1, 8), (1, 5, 4), (1, 8, 4), (1, 3, 3, 2), (1, 4, 3, 2), (1, 5, 4, 2), (1, 6, 4, 2),
           (1, 8, 2), (1, 6, 2, 2), (1, 6, 1, 2), (1, 8, 1)]
ENGINE_IDS = ["blocklaunch", "persist_lag1", "persist_lag2", "persist_lag3", "persist_lag4", "persist_lag6", "persist_lag8",
              "persist_lag5_near4", "persist_lag8_near4", "rows_lag3", "rows_lag4", "rows_lag5_near4", "rows_lag6_near4",
              "persist_lag8_near2", "rows_lag6_near2", "rows_lag6_near1", "persist_lag8_near1"]


def _pair(ngp, O, X, seed=1001, chain=0, engine=(1, 6), chain_form=None):
    """engine = (mode, lag), (mode, lag, near lags) or (mode, lag, near lags, streamer variant)."""
    s = ngp.Sampler(device=0, seed=seed, chain=chain, mode=engine[0], lag=engine[1], streamer=engine[3] if len(engine) > 3 else 1)
    if chain_form is not None:
        s.set_chain_form(chain_form)
    if len(engine) > 2:
        s.set_near(engine[2])
    s.set_panel(X)
    R, S, nblk = s.layout()
    mode, D = s.config()
    assert mode == engine[0] and D == (engine[1] if mode == 1 else 1)
    assert s.near() == (engine[2] if len(engine) > 2 else 3)   # short shards: 3 unless asked otherwise
    variant, nchain = s.streamer()
    assert (variant, nchain) == ((2, 7) if len(engine) > 3 and engine[3] == 2 else ((1, 8) if mode == 1 else (0, 8)))
    o = O.Oracle(order=1, seed=seed, chain=chain)
    o.set_panel_f32(X, R=R, S=S, D=D, near=s.near(), nchain=nchain, tform=s.chain_form())
    return s, o


def test_gram_and_mpm(ngp, O):
    X, y, bt, v = make_problem(O, 333, 150)
    s, o = _pair(ngp, O, X)
    for t in range(3):
        assert np.array_equal(s.gram(t), o.get_gram(t))
    G = X.astype(np.float64).T @ X.astype(np.float64)
    assert np.allclose(s.mpm(), np.diag(G), rtol=1e-12)
    b = np.random.default_rng(1).normal(size=150)
    assert np.allclose(s.xbeta(b), X.astype(np.float64) @ b, rtol=1e-11, atol=1e-11)
    # The Gram window can be built on the matrix cores (v_mfma_f64_16x16x4_f64, knob bit 10): the matrix core adds the four products of
    # a quad in row order, i.e. forms the same sequential fma chains over the shard's rows as the fp64 VALU kernel -- every plane of the
    # window identical, and so the chains drawn with either
    v = ngp.Sampler(device=0, seed=3, chain=0, mode=1, lag=6)
    v.debug_set_knob(1024); v.set_panel(X)
    m = ngp.Sampler(device=0, seed=3, chain=0, mode=1, lag=6)
    m.set_panel(X)
    for t in range(3):
        assert np.array_equal(v.gram(t), m.gram(t)) and np.array_equal(m.gram(t), o.get_gram(t))
    assert np.array_equal(v.mpm(), m.mpm())
    for q in (v, m):
        add_sets(q, [(0, 150, "PR")], 0.01); q.set_y(y); q.run(8)     # lag 6: the cross planes d = 1..5 enter every block
    assert np.array_equal(v.get_state()["beta"], m.get_state()["beta"]) and np.array_equal(v.get_state()["ycorr"], m.get_state()["ycorr"])


def test_panel_in_column_ranges_equals_whole_panel(ngp, O):
    """ngp_begin_panel / ngp_panel_columns_* / ngp_end_panel (one marker set after another, as M[set].data of src/mme.jl:296-311 come,
    without a concatenated host copy) build bit for bit the panel of ngp_set_panel_f64: means (sequential sum / N), tiles, mpm, Gram."""
    rng = np.random.default_rng(3)
    N, P = 333, 300
    G = rng.integers(0, 3, size=(N, P)).astype(np.float64) + rng.normal(size=(N, P)) * 0.01
    whole = ngp.Sampler(device=0, seed=1, chain=0)
    whole.set_panel(G, centre=True)
    parts = ngp.Sampler(device=0, seed=1, chain=0)
    parts.begin_panel(N, P)
    with pytest.raises(ngp.NextGPHipError, match="still open"):
        parts.add_marker_set(0, P, 0, 4.0, 1.0, [(0, P)], [1.0]); parts.set_y(np.zeros(N)); parts.run(1)
    parts = ngp.Sampler(device=0, seed=1, chain=0)
    parts.begin_panel(N, P)
    for call in (parts.mpm, lambda: parts.xbeta(np.zeros(P)), lambda: parts.gram(0)):   # x'x and the Gram window do not exist yet
        with pytest.raises(ngp.NextGPHipError, match="still open"):
            call()
    for a, b in ((130, 300), (0, 57), (57, 130)):       # any order, boundaries inside 64-column blocks
        parts.panel_columns(a, G[:, a:b], centre=True)
    with pytest.raises(ngp.NextGPHipError, match="outside the panel"):
        parts.panel_columns(250, G[:, :100], centre=True)
    parts.end_panel()
    mw, mp = whole.means(), parts.means()
    assert np.array_equal(mw, mp) and np.array_equal(mw, np.cumsum(G, axis=0)[-1] / N)   # the sequential sum of the host loop it replaces
    assert np.array_equal(whole.mpm(), parts.mpm())
    for t in range((P + 63) // 64):
        assert np.array_equal(whole.gram(t), parts.gram(t))
    for j in (0, 56, 57, 129, 130, 299):                 # X e_j recovers column j of the tiles exactly
        e = np.zeros(P); e[j] = 1.0
        col = (G[:, j] - mw[j]).astype(np.float32).astype(np.float64)
        assert np.array_equal(whole.xbeta(e), col) and np.array_equal(parts.xbeta(e), col)
    # Float32 input, not centred; a column range never written stays a zero column
    F = G.astype(np.float32)
    f = ngp.Sampler(device=0, seed=1, chain=0)
    f.begin_panel(N, P); f.panel_columns(64, F[:, 64:200]); f.end_panel()
    e = np.zeros(P); e[70] = 1.0
    assert np.array_equal(f.xbeta(e), F[:, 70].astype(np.float64)) and not f.means().any()
    e = np.zeros(P); e[10] = 1.0
    assert not f.xbeta(e).any() and f.mpm()[10] == 0.0
    # genotype codes in column ranges, fp32 tiles and compact storage: the panel of ngp_set_panel_u8 bit for bit
    C8 = rng.integers(0, 3, size=(N, P)).astype(np.uint8)
    yy = rng.normal(size=N)
    for storage in (None, "u8"):
        w8 = ngp.Sampler(device=0, seed=1, chain=0, storage=storage); w8.set_panel(C8, centre=True)
        p8 = ngp.Sampler(device=0, seed=1, chain=0, storage=storage); p8.begin_panel(N, P)
        for a, b in ((200, 300), (0, 61), (61, 200)):
            p8.panel_columns(a, C8[:, a:b], centre=True)
        p8.end_panel()
        assert np.array_equal(w8.means(), p8.means()) and np.array_equal(w8.mpm(), p8.mpm()) and w8.layout() == p8.layout()
        for t in range((P + 63) // 64):
            assert np.array_equal(w8.gram(t), p8.gram(t))
        for q in (w8, p8):
            q.add_marker_set(0, P, 0, 4.0, 0.01, [(0, P)], [0.02]); q.set_y(yy); q.set_residual_prior(4.0, 0.5); q.run(6)
        sa, sb = w8.get_state(), p8.get_state()
        assert np.array_equal(sa["beta"], sb["beta"]) and np.array_equal(sa["ycorr"], sb["ycorr"]) and sa["varE"] == sb["varE"]
    cs = ngp.Sampler(device=0, seed=1, chain=0, storage="u8"); cs.begin_panel(N, P)
    with pytest.raises(ngp.NextGPHipError, match="genotype codes"):
        cs.panel_columns(0, G[:, :10])
    bad = G.copy(); bad[5, 7] = np.inf
    b = ngp.Sampler(device=0, seed=1, chain=0)
    with pytest.raises(ngp.NextGPHipError, match="non-finite"):
        b.set_panel(bad, centre=True)
    with pytest.raises(ngp.NextGPHipError, match="open panel"):
        ngp.Sampler(device=0, seed=1, chain=0).end_panel()


def test_generated_panel_matches_oracle(ngp, O):
    N, P = 517, 130
    X, mu = O.generate_panel(N, P, seed=99)
    s = ngp.Sampler(device=0, seed=1, chain=0)
    s.generate_panel(N, P, seed=99)
    # X*e_j recovers column j exactly (single non-zero term)
    for j in (0, 63, 64, 129):
        e = np.zeros(P); e[j] = 1.0
        assert np.array_equal(s.xbeta(e), X[:, j].astype(np.float64))


CASES = [
    ("pr_single", 500, 1000, [(0, 1000, "PR")]),
    ("pr_ragged", 257, 130, [(0, 130, "PR")]),                 # P not a multiple of 64, N not of R
    ("pr_regions", 300, 200, [(0, 200, ("PRw", 37))]),
    ("pr_r1", 200, 100, [(0, 100, "PR1")]),                    # BayesA-like, one region per SNP
    ("b_single", 500, 600, [(0, 600, "B")]),
    ("b_fixpi", 300, 128, [(0, 128, "Bfix")]),
    ("multi", 400, 450, [(0, 150, "PR"), (150, 170, "B"), (320, 130, "PR")]),   # sets straddle 64-blocks
    ("c_single", 400, 520, [(0, 520, "C")]),
    ("c_fixpi_multi", 300, 330, [(0, 100, "Cfix"), (100, 90, "B"), (190, 140, "C")]),
    ("tiny", 7, 3, [(0, 3, "PR")]),
    ("r_single", 400, 520, [(0, 520, "R")]),                                     # BayesR (src/functions.jl:238-289)
    ("r_multi", 300, 330, [(0, 100, "Rfix"), (100, 90, "B"), (190, 100, "R2"), (290, 40, "PR")]),   # BayesR lanes beside others in one block
    ("r_six_eight", 300, 400, [(0, 150, "R6"), (150, 100, "PR"), (250, 150, "R8")]),             # more classes than the chain keeps in registers
    ("r_twelve_sixteen", 300, 400, [(0, 150, "R12"), (150, 100, "B"), (250, 150, "R16")]),       # more classes than the sampler stages in LDS
]


def test_inverse_form_of_the_bayespr_chain(ngp, O):
    """ngp_set_chain_form(1): BayesPR blocks as dlt = T e0 (k_tinv) -- bit for bit the blocked oracle told the same (every engine
    family, models with and without non-linear blocks, the fine seam), and the chain of the 64-step form to rounding."""
    for name, N, P, spec in [c for c in CASES if c[0] in ("pr_single", "pr_ragged", "pr_regions", "multi", "tiny")]:
        X, y, bt, v = make_problem(O, N, P, seed=5)
        for engine in [(0, 1), (1, 1), (1, 4), (1, 8), (1, 6, 2, 2), (1, 3, 3, 2)]:
            s, o = _pair(ngp, O, X, engine=engine, chain_form=1)
            assert s.chain_form() == 1
            for m in (s, o):
                add_sets(m, spec, v); m.set_y(y); m.set_residual_prior(4.0, 0.25 * y.var()); m.set_schedule(10, 4, 2); m.run(10)
            a, b = s.get_state(), o.get_state()
            for k in ("ycorr", "beta", "delta", "varBeta", "piHat"):
                assert np.array_equal(a[k], b[k]), (name, engine, k)
            assert a["varE"] == b["varE"] and a["b"] == b["b"]
        s0, _ = _pair(ngp, O, X, engine=(1, 8), chain_form=0)
        add_sets(s0, spec, v); s0.set_y(y); s0.set_residual_prior(4.0, 0.25 * y.var()); s0.set_schedule(10, 4, 2); s0.run(10)
        c = s0.get_state()
        assert np.array_equal(c["delta"], a["delta"]) or name == "multi"   # (the last engine's chain, same draws)
        assert np.abs(c["beta"] - a["beta"]).max() <= 1e-8 * max(1e-300, np.abs(c["beta"]).max())


@pytest.mark.parametrize("engine", ENGINES, ids=ENGINE_IDS)
@pytest.mark.parametrize("name,N,P,spec", CASES, ids=[c[0] for c in CASES])
def test_chain_bit_exact_vs_blocked_oracle(ngp, O, name, N, P, spec, engine):
    X, y, bt, v = make_problem(O, N, P, seed=5)
    s, o = _pair(ngp, O, X, engine=engine)
    niter = 12
    for m in (s, o):
        add_sets(m, spec, v)
        m.set_y(y)
        m.set_residual_prior(4.0, 0.5 * y.var() * 0.5)
        m.set_schedule(niter, 4, 2)
        m.run(niter)
    a, b = s.get_state(), o.get_state()
    assert np.array_equal(a["delta"], b["delta"])
    for k in ("ycorr", "beta", "varBeta", "piHat"):
        assert np.array_equal(a[k], b[k]), k
    assert a["varE"] == b["varE"] and a["b"] == b["b"] and a["iter"] == b["iter"] == niter
    ta, tb = s.get_trace(niter), o.get_trace(niter)
    assert np.array_equal(ta["varE"], tb["varE"]) and np.array_equal(ta["b"], tb["b"])
    pa, pb = s.get_posterior_sums(), o.get_posterior_sums()
    assert pa["nKept"] == pb["nKept"] == 4
    for k in ("sum_beta", "sum_beta2", "sum_delta", "sum_varBeta", "sum_pi"):
        assert np.array_equal(pa[k], pb[k]), k
    assert pa["sum_varE"] == pb["sum_varE"] and pa["sum_b"] == pb["sum_b"]
    for si, (_, _, kind) in enumerate(spec):
        if isinstance(kind, str) and kind.startswith("R"):          # BayesR: class probabilities and their posterior sums
            ca, cb = s.get_class_state(si), o.get_class_state(si)
            assert np.array_equal(ca["piHat"], cb["piHat"]) and np.array_equal(ca["sum_pi"], cb["sum_pi"]) and abs(ca["piHat"].sum() - 1.0) < 1e-14
            cls = a["delta"][spec[si][0]:spec[si][0] + spec[si][1]]
            assert cls.min() >= 1 and cls.max() <= len(ca["piHat"])
    # self-consistency: ycorr == y - 1 b - X beta recomputed from scratch
    resid = y - a["b"] - s.xbeta(a["beta"])
    assert np.abs(a["ycorr"] - resid).max() <= 1e-10 * max(1.0, np.abs(y).max())


TALL = [  # (N, P, sets, max shards, V): streamer variants 4 / 6 = V = 2 / 3 shards per streamer workgroup (what fp32 panels above 63k rows run in)
    ("r4_default", 500, 1000, [(0, 1000, "PR")], None, 2),                                     # 4-row shards: one quad per tile
    ("r220_w4", 1700, 300, [(0, 300, "PR")], 8, 2),                                            # the tallest tiles the ring holds twice
    ("r188_multi", 1500, 450, [(0, 150, "PR"), (150, 170, "B"), (320, 130, "R")], 8, 2),
    ("odd_S_padded", 50, 200, [(0, 200, ("PRw", 37))], 6, 2),                                  # 5 shards -> a sixth, all padding
    ("r76_ragged", 601, 130, [(0, 130, "C")], 8, 2),
    ("ng15", 1900, 200, [(0, 120, "PR"), (120, 80, "B")], None, 2),                            # 476 shards: 15 reducer groups (> 8: further batches of group sums)
    ("v3_r220", 2600, 300, [(0, 300, "PR")], 12, 3),                                           # three shards per workgroup, lag 2
    ("v3_multi", 1000, 450, [(0, 150, "PR"), (150, 170, "B"), (320, 130, "R")], 9, 3),
    ("v3_padded", 50, 200, [(0, 200, ("PRw", 37))], 6, 3),                                     # 5 shards -> 6
    ("v3_ng22", 2790, 200, [(0, 120, "PR"), (120, 80, "B")], None, 3),                         # 699 shards: 22 reducer groups
]


@pytest.mark.parametrize("name,N,P,spec,shards,V", TALL, ids=[c[0] for c in TALL])
def test_several_shards_per_workgroup_bit_exact(ngp, O, name, N, P, spec, shards, V):
    """role_streamer_rows_tall: the layout (R, S), the partial of every shard and the reducers' groups are those of the row-owning
    streamer, a workgroup merely owns V shards -- so the chain must equal the blocked oracle's for that layout (7 GEMV chains,
    lag 3 with two shards, lag 2 with three), bit for bit."""
    X, y, bt, v = make_problem(O, N, P, seed=5)
    s = ngp.Sampler(device=0, seed=1001, chain=0, mode=1, lag=6, streamer=2 * V)
    if shards:
        s.set_max_shards(shards)
    s.set_panel(X)
    R, S, nblk = s.layout()
    D = 5 - V
    assert s.config() == (1, D) and s.streamer() == (2, 7) and S % V == 0 and R <= 224 and R * S >= N
    if name == "ng15":
        assert (R, S) == (4, 476)
    if name == "v3_ng22":
        assert (R, S) == (4, 699)
    if "padded" in name:
        assert R * (S - 1) >= N                 # the last shard holds no row of the panel
    o = O.Oracle(order=1, seed=1001, chain=0)
    o.set_panel_f32(X, R=R, S=S, D=D, near=s.near(), nchain=7, tform=s.chain_form())
    niter = 12
    for m in (s, o):
        add_sets(m, spec, v)
        m.set_y(y); m.set_residual_prior(4.0, 0.25 * y.var()); m.set_schedule(niter, 4, 2); m.run(niter)
    a, b = s.get_state(), o.get_state()
    assert np.array_equal(a["delta"], b["delta"])
    for k in ("ycorr", "beta", "varBeta", "piHat"):
        assert np.array_equal(a[k], b[k]), k
    assert a["varE"] == b["varE"] and a["b"] == b["b"] and a["iter"] == niter
    pa, pb = s.get_posterior_sums(), o.get_posterior_sums()
    for k in ("sum_beta", "sum_beta2", "sum_delta", "sum_varBeta", "sum_pi"):
        assert np.array_equal(pa[k], pb[k]), k
    assert s.census()["grid"] == 1 + (S + 31) // 32 + S // V    # S / V streamers
    resid = y - a["b"] - s.xbeta(a["beta"])
    assert np.abs(a["ycorr"] - resid).max() <= 1e-10 * max(1.0, np.abs(y).max())


@pytest.mark.parametrize("engine", [(0, 1), (1, 6)], ids=["blocklaunch", "persist_lag6"])
@pytest.mark.parametrize("kind", ["PR", "B", "C", "R"])
def test_chain_vs_reference_order_oracle(ngp, O, kind, engine):
    """Same Markov chain, reference summation order: indicators identical, floats within 1e-9 relative."""
    N, P = 500, 1000
    X, y, bt, v = make_problem(O, N, P, seed=9)
    s = ngp.Sampler(device=0, seed=77, chain=1, mode=engine[0], lag=engine[1])
    s.set_panel(X)
    o = O.Oracle(order=0, seed=77, chain=1)
    o.set_panel_f32(X)
    for m in (s, o):
        add_sets(m, [(0, P, kind)], v)
        m.set_y(y); m.set_residual_prior(4.0, 0.25 * y.var()); m.set_schedule(30, 10, 5); m.run(30)
    a, b = s.get_state(), o.get_state()
    assert np.array_equal(a["delta"], b["delta"])
    tol = 1e-9
    assert np.abs(a["beta"] - b["beta"]).max() <= tol * max(1e-3, np.abs(b["beta"]).max())
    assert np.abs(a["ycorr"] - b["ycorr"]).max() <= tol * np.abs(b["ycorr"]).max()
    assert abs(a["varE"] - b["varE"]) <= tol * b["varE"]
    assert np.abs(a["varBeta"] - b["varBeta"]).max() <= tol * max(1e-12, np.abs(b["varBeta"]).max())
    pa, pb = s.get_posterior_sums(), o.get_posterior_sums()
    assert pa["nKept"] == pb["nKept"]
    assert np.abs(pa["sum_beta"] - pb["sum_beta"]).max() <= tol * max(1e-3, np.abs(pb["sum_beta"]).max())


@pytest.mark.parametrize("engine", [(0, 1), (1, 4)], ids=["blocklaunch", "persist_lag4"])
def test_fine_seam_matches_coarse(ngp, O, engine):
    """ngp_sweep_set driven by host-side varE/intercept == the blocked oracle's set sweep on the same state."""
    N, P = 300, 192
    X, y, bt, v = make_problem(O, N, P, seed=2)
    s = ngp.Sampler(device=0, seed=5, chain=0, mode=engine[0], lag=engine[1])
    s.set_panel(X)
    add_sets(s, [(0, P, "PR")], v)
    ycorr = y - y.mean()
    beta = np.zeros(P); vb = np.array([v])
    for it in range(3):
        d = s.sweep_set(0, 1.3, ycorr, beta, vb)
        assert d.min() == 1 and np.isfinite(beta).all() and vb[0] > 0
        resid = (y - y.mean()) - s.xbeta(beta)
        assert np.abs(ycorr - resid).max() < 1e-10


@pytest.mark.parametrize("method", ["PR", "B"])
def test_fine_seam_over_device_arrays(ngp, O, method):
    """ngp_sweep_set_dev: the caller's state stays in device memory (torch tensors here, ROCArrays in a Julia host) -- the same
    keyed draws, so bit for bit what ngp_sweep_set leaves in host arrays."""
    import torch
    N, P = 300, 192
    X, y, bt, v = make_problem(O, N, P, seed=2)
    host, devs = (ngp.Sampler(device=0, seed=5, chain=0, mode=1, lag=4) for _ in range(2))
    nvb = P if method == "B" else 1
    for s in (host, devs):
        s.set_panel(X); add_sets(s, [(0, P, method)], v)
    ycorr = y - y.mean(); beta = np.zeros(P); vb = np.full(nvb, v); pi = np.array([0.7, 0.3]) if method == "B" else None
    t = {k: torch.tensor(a, dtype=torch.float64, device="cuda:0") for k, a in (("ycorr", ycorr), ("beta", beta), ("vb", vb))}
    t["delta"] = torch.zeros(P, dtype=torch.int64, device="cuda:0")
    t["pi"] = torch.tensor(pi if pi is not None else [0.0, 0.0], dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    for it in range(3):
        d = host.sweep_set(0, 1.3, ycorr, beta, vb, pi)
        devs.sweep_set_dev(0, 1.3, t["ycorr"].data_ptr(), t["beta"].data_ptr(), t["vb"].data_ptr(), t["delta"].data_ptr(),
                           t["pi"].data_ptr() if pi is not None else 0)
        assert np.array_equal(t["ycorr"].cpu().numpy(), ycorr) and np.array_equal(t["beta"].cpu().numpy(), beta)
        assert np.array_equal(t["vb"].cpu().numpy(), vb) and np.array_equal(t["delta"].cpu().numpy(), d)
        if pi is not None:
            assert np.array_equal(t["pi"].cpu().numpy(), pi)
    with pytest.raises(ngp.NextGPHipError, match="null state pointer"):
        devs.sweep_set_dev(0, 1.3, 0, t["beta"].data_ptr(), t["vb"].data_ptr())


def test_resume_state_roundtrip(ngp, O):
    N, P = 200, 128
    X, y, bt, v = make_problem(O, N, P, seed=4)
    runs = []
    for split in (None, 5):
        s = ngp.Sampler(device=0, seed=3, chain=0)
        s.set_panel(X); add_sets(s, [(0, P, "B")], v); s.set_y(y); s.set_residual_prior(4.0, 1.0); s.set_schedule(0, 0, 1)
        if split is None:
            s.run(10)
        else:
            s.run(split)
            st = s.get_state()
            s2 = ngp.Sampler(device=0, seed=3, chain=0)
            s2.set_panel(X); add_sets(s2, [(0, P, "B")], v); s2.set_y(y); s2.set_residual_prior(4.0, 1.0)
            s2.set_state(st)
            s2.run(10 - split)
            s = s2
        runs.append(s.get_state())
    for k in ("ycorr", "beta", "delta", "varBeta", "piHat"):
        assert np.array_equal(runs[0][k], runs[1][k]), k


def test_errors_are_reported_not_thrown(ngp):
    s = ngp.Sampler(device=0, seed=1, chain=0)
    with pytest.raises(ngp.NextGPHipError, match="panel not set"):
        s.set_y(np.zeros(3))
    s.set_panel(np.zeros((4, 2), dtype=np.float32))
    with pytest.raises(ngp.NextGPHipError, match="N entries"):
        s.set_y(np.zeros(3))
    with pytest.raises(ngp.NextGPHipError, match="non-finite"):
        s.set_y(np.array([0.0, np.nan, 0.0, 0.0]))
    with pytest.raises(ngp.NextGPHipError, match="outside the panel"):
        s.add_marker_set(0, 5, 0, 4.0, 0.1, [(0, 5)], [0.1])
    with pytest.raises(ngp.NextGPHipError, match="no marker set"):
        s.set_y(np.zeros(4)); s.run(1)


def test_u8_panel_equals_f64_panel(ngp, O):
    """ngp_set_panel_u8: one byte per genotype, centred and converted on the device -- bit for bit the chain of the same
    values handed over as Float64 (src/prepMatVec.jl:129 centring)."""
    N, P = 333, 210
    X1 = O.generate_panel(N, P, seed=8)[0]
    G = np.rint(X1.astype(np.float64) - X1.astype(np.float64).min(axis=0)).astype(np.uint8)
    assert set(np.unique(G)) <= {0, 1, 2}
    y = np.random.default_rng(4).normal(size=N) + G[:, 3] - G[:, 77]
    out = []
    for panel in (G, G.astype(np.float64)):
        s = ngp.Sampler(device=0, seed=3, chain=0)
        s.set_panel(np.asfortranarray(panel), centre=True)
        s.add_marker_set(0, P, 0, 4.0, 0.01, [(0, P)], [0.02])
        s.set_y(y); s.set_residual_prior(4.0, 0.5)
        s.run(5)
        out.append((s.mpm(), s.get_state()))
    assert np.array_equal(out[0][0], out[1][0])
    for k in ("beta", "ycorr", "varBeta"):
        assert np.array_equal(out[0][1][k], out[1][1][k]), k
    assert out[0][1]["varE"] == out[1][1]["varE"]
    # not centred: the values themselves
    s = ngp.Sampler(device=0, seed=3, chain=0)
    s.set_panel(np.asfortranarray(G), centre=False)
    assert np.allclose(s.mpm(), (G.astype(np.float64) ** 2).sum(axis=0))


@pytest.mark.parametrize("N", [14700, 16000, 30400, 31700, 54000, 55000, 62000, 63232],
                         ids=["R60", "R68", "R124", "R132", "R220", "R228", "R252", "R256"])
def test_shard_height_boundaries_bit_exact(ngp, O, N):
    """Rows per shard at the boundaries of the update-task mappings (1, 2, 4 rows per thread), of the lag / near-lag
    rules for tall shards and of the LDS budget: bit-exact against the blocked oracle with the layout the library reports."""
    P = 448 + 17
    X = O.generate_panel(N, P, seed=3)[0]
    rng = np.random.default_rng(2)
    y = 1.0 + X[:, :5].astype(np.float64) @ rng.normal(size=5) + rng.normal(size=N)
    s = ngp.Sampler(device=0, seed=21, chain=0)
    s.set_panel(X)
    R, S, nblk = s.layout()
    mode, D = s.config()
    assert mode == 1 and R == {14700: 60, 16000: 68, 30400: 124, 31700: 132, 54000: 220, 55000: 228, 62000: 252, 63232: 256}[N]
    variant, nchain = s.streamer()
    assert (variant, nchain) == ((2, 7) if 64 <= R <= 224 else (1, 8))   # row-owning waves from 64-row shards on (measured ahead there)
    assert (D, s.near()) == ((6, 2) if variant == 2 else ((6, 3) if R <= 128 else (5, 4)))   # (short shards: lag 6 since round 4, 8 before)
    o = O.Oracle(order=1, seed=21, chain=0)
    o.set_panel_f32(X, R=R, S=S, D=D, near=s.near(), nchain=nchain, tform=s.chain_form())
    v = 0.01
    for m in (s, o):
        add_sets(m, [(0, 300, "PR"), (300, P - 300, "B")], v)
        m.set_y(y); m.set_residual_prior(4.0, 0.5); m.run(4)
    a, b = s.get_state(), o.get_state()
    assert np.array_equal(a["delta"], b["delta"])
    for k in ("beta", "ycorr", "varBeta", "piHat"):
        assert np.array_equal(a[k], b[k]), k
    assert a["varE"] == b["varE"] and a["b"] == b["b"]


@pytest.mark.parametrize("engine", [(0, 1), (1, 6), (1, 8, 4)], ids=["blocklaunch", "persist_lag6", "persist_lag8_near4"])
def test_summary_stat_terms_no_intercept_long_regions(ngp, O, engine):
    """M.lhs / M.rhs of summary statistics (src/mme.jl:314-322; BayesC drops M.rhs, src/functions.jl:220), a model without
    intercept (src/functions.jl:41-47 skipped) and variance regions longer than one 256-locus segment."""
    N, P = 300, 1500
    X, y, bt, v = make_problem(O, N, P, seed=21)
    rng = np.random.default_rng(5)
    s, o = _pair(ngp, O, X, seed=31, chain=2, engine=engine)
    ro = O.Oracle(order=0, seed=31, chain=2); ro.set_panel_f32(X)
    df = 4.0
    sets = [(0, 700, 0, [(0, 300), (300, 650), (650, 700)], None), (700, 400, 1, [(j, j + 1) for j in range(400)], 0.1),
            (1100, 400, 2, [(0, 400)], 0.2)]
    for m in (s, o, ro):
        for col0, ncol, method, regs, pi0 in sets:
            lhs0 = np.abs(np.random.default_rng(col0).normal(size=ncol)) * 0.3
            rhs0 = np.random.default_rng(col0 + 1).normal(size=ncol) * 0.2
            m.add_marker_set(col0, ncol, method, df, v * (df - 2) / df, regs, [v] * len(regs), pi0=pi0 or 0.0, estPi=bool(pi0),
                             lhs0=lhs0, rhs0=rhs0)
        m.set_y(y); m.set_intercept(False); m.set_residual_prior(4.0, 0.25 * y.var()); m.run(10)
    a, b, c = s.get_state(), o.get_state(), ro.get_state()
    assert a["b"] == 0.0 and b["b"] == 0.0
    assert np.array_equal(a["delta"], b["delta"]) and np.array_equal(a["delta"], c["delta"])
    for k in ("beta", "ycorr", "varBeta", "piHat"):
        assert np.array_equal(a[k], b[k]), k
        assert np.abs(a[k] - c[k]).max() <= 1e-9 * max(1e-6, np.abs(c[k]).max()), k
    assert a["varE"] == b["varE"]


def test_fine_seam_bayesc_and_second_set(ngp, O):
    """Fine seam (src/samplers.jl:52) for a BayesC set that does not start at column 0: the sweep covers only the blocks of
    that set; ycorr, beta, delta, the single variance and piHat come back; the other set's columns are untouched."""
    N, P = 250, 400
    X, y, bt, v = make_problem(O, N, P, seed=14)
    s = ngp.Sampler(device=0, seed=8, chain=0)
    s.set_panel(X)
    s.add_marker_set(0, 150, 0, 4.0, v * 0.5, [(0, 150)], [v])
    s.add_marker_set(150, 250, 2, 4.0, v * 0.5, [(0, 250)], [v], pi0=0.3, estPi=True)
    ycorr = y - y.mean()
    beta = np.zeros(250); vb = np.array([v]); pi = np.array([0.7, 0.3])
    for it in range(4):
        d = s.sweep_set(1, 0.9, ycorr, beta, vb, pi)
        assert set(np.unique(d)) <= {0, 1} and np.all(beta[d == 0] == 0.0) and vb[0] > 0 and abs(pi.sum() - 1.0) < 1e-15
        full = np.zeros(P); full[150:] = beta
        assert np.abs(ycorr - ((y - y.mean()) - s.xbeta(full))).max() < 1e-10
    assert 0.0 < pi[1] < 1.0 and pi[1] != 0.3


def test_monomorphic_columns(ngp, O):
    """Columns without variation (x'x = 0 after centring): the reference's arithmetic gives lhs = 1/varBeta for BayesPR and an
    inclusion probability of NaN, i.e. never included, for BayesB / BayesC (log 0, 0/0 in src/functions.jl:169-173).  Oracle
    (both orders) and device agree."""
    N, P = 120, 192
    X, y, bt, v = make_problem(O, N, P, seed=15)
    X = np.asfortranarray(X)
    X[:, [3, 70, 100, 150, 191]] = 0.0
    s, o = _pair(ngp, O, X, seed=12, chain=0, engine=(1, 3))
    ro = O.Oracle(order=0, seed=12, chain=0); ro.set_panel_f32(X)
    for m in (s, o, ro):
        add_sets(m, [(0, 64, "PR"), (64, 64, "Bfix"), (128, 64, "Cfix")], v)
        m.set_y(y); m.set_residual_prior(4.0, 0.25 * y.var()); m.run(6)
    a, b, c = s.get_state(), o.get_state(), ro.get_state()
    assert np.array_equal(a["delta"], b["delta"]) and np.array_equal(a["delta"], c["delta"])
    assert a["delta"][70] == 0 and a["delta"][100] == 0 and a["delta"][150] == 0 and a["delta"][191] == 0 and a["delta"][3] == 1
    for k in ("beta", "ycorr", "varBeta"):
        assert np.array_equal(a[k], b[k]), k
        assert np.all(np.isfinite(a[k]))
        assert np.abs(a[k] - c[k]).max() <= 1e-9 * max(1e-6, np.abs(c[k]).max()), k


# ----------------------------------------------------------------------------------------------
# round 2: golden fixtures on the device, BASELINE configs[0] at full size, and the ABI v2 entry points
# ----------------------------------------------------------------------------------------------
GOLD = __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "golden")


@pytest.mark.parametrize("engine", [(0, 1), (1, 6), (1, 4, 3, 2)], ids=["blocklaunch", "persist_lag6", "rows_lag4"])
@pytest.mark.parametrize("name", ["pr_50x200", "b_50x200", "c_50x200"])
def test_golden_vectors_on_the_device(ngp, name, engine):
    """The committed fixtures (tests/golden/*.npz: inputs + state after iterations 1, 2, 10, written by the reference-order
    oracle) are compared with the HIP path directly -- so oracle and kernels cannot drift together unnoticed."""
    g = np.load(__import__("os").path.join(GOLD, name + ".npz"))
    X = np.asfortranarray(g["X"])
    s = ngp.Sampler(device=0, seed=int(g["seed"]), chain=int(g["chain"]), mode=engine[0], lag=engine[1], streamer=engine[3] if len(engine) > 3 else 1)
    s.set_panel(X)
    add_sets(s, [(0, X.shape[1], {"pr": "PR", "b_": "B", "c_": "C"}[name[:2]])], float(g["v"]))
    s.set_y(g["y"]); s.set_residual_prior(4.0, float(g["e_scale"]))
    done, tol = 0, 1e-9
    for it in (1, 2, 10):
        s.run(it - done); done = it
        st = s.get_state()
        assert np.array_equal(st["delta"], g[f"delta_{it}"])
        assert np.abs(st["beta"] - g[f"beta_{it}"]).max() <= tol * max(1e-3, np.abs(g[f"beta_{it}"]).max())
        assert abs(st["varE"] - float(g[f"varE_{it}"])) <= tol * float(g[f"varE_{it}"])
        assert np.abs(st["varBeta"] - g[f"varBeta_{it}"]).max() <= tol * max(1e-9, np.abs(g[f"varBeta_{it}"]).max())
        assert abs(st["b"] - float(g[f"b_{it}"])) <= tol * max(1.0, abs(float(g[f"b_{it}"])))
        assert np.abs(st["piHat"] - g[f"piHat_{it}"]).max() <= tol


def test_config0_full_size_vs_reference_order(ngp, O):
    """BASELINE.json configs[0] as stated: BayesPR, 500 individuals x 5,000 SNPs, 1,000 iterations -- the device chain against the
    reference-order CPU oracle over the whole run (indicators trivially 1; effects, residuals, variances and the posterior
    means of the kept half within 1e-9 relative)."""
    N, P, niter = 500, 5000, 1000
    X, mu = O.generate_panel(N, P)
    y = 10.0 + X.astype(np.float64)[:, ::97] @ np.random.default_rng(1).normal(size=len(range(0, P, 97))) + np.random.default_rng(2).normal(size=N) * 3.0
    v = 0.5 * y.var() / float((mu * (1 - mu / 2)).sum())
    s = ngp.Sampler(device=0, seed=1001, chain=0)
    s.set_panel(X)
    o = O.Oracle(0, seed=1001, chain=0); o.set_panel_f32(X)
    for m in (s, o):
        add_sets(m, [(0, P, "PR")], v); m.set_y(y); m.set_residual_prior(4.0, 0.25 * y.var()); m.set_schedule(niter, 500, 1); m.run(niter)
    a, b = s.get_state(), o.get_state()
    tol = 1e-9
    assert np.abs(a["beta"] - b["beta"]).max() <= tol * np.abs(b["beta"]).max()
    assert np.abs(a["ycorr"] - b["ycorr"]).max() <= tol * np.abs(b["ycorr"]).max()
    assert abs(a["varE"] - b["varE"]) <= tol * b["varE"] and abs(a["varBeta"][0] - b["varBeta"][0]) <= tol * b["varBeta"][0]
    pa, pb = s.get_posterior_sums(), o.get_posterior_sums()
    assert pa["nKept"] == pb["nKept"] == 500
    assert np.abs(pa["sum_beta"] - pb["sum_beta"]).max() <= tol * np.abs(pb["sum_beta"]).max()
    assert abs(pa["sum_varE"] - pb["sum_varE"]) <= tol * pb["sum_varE"]
    ta, tb = s.get_trace(niter), o.get_trace(niter)
    assert np.abs(ta["varE"] - tb["varE"]).max() <= tol * tb["varE"].max()


def _small_model(ngp, O, seed=3, kind=None):
    N, P = 200, 192
    X, y, bt, v = make_problem(O, N, P, seed=4)
    s = ngp.Sampler(device=0, seed=seed, chain=0)
    s.set_panel(X); add_sets(s, kind or [(0, 100, "PR"), (100, 92, "B")], v); s.set_y(y); s.set_residual_prior(4.0, 1.0)
    return s, X, y, v


def test_snapshot_resume_reproduces_the_uninterrupted_run(ngp, O, tmp_path):
    """ngp_save_snapshot / ngp_load_snapshot (chain state + posterior sums + stream identity in one file): a run interrupted
    after 7 of 20 iterations and resumed on a NEW handle ends with bit-identical state, traces and posterior sums -- what the
    reference gets from its append-only *Out files (src/outFiles.jl:17-21)."""
    full, *_ = _small_model(ngp, O)
    full.set_schedule(20, 4, 2); full.run(20)
    first, *_ = _small_model(ngp, O)
    first.set_schedule(20, 4, 2); first.run(7)
    path = str(tmp_path / "chain.ngpsnap")
    first.save_snapshot(path)
    second, *_ = _small_model(ngp, O, seed=999)           # the snapshot carries seed and chain id
    second.set_schedule(20, 4, 2)
    second.load_snapshot(path)
    second.run(13)
    a, b = full.get_state(), second.get_state()
    for k in ("ycorr", "beta", "delta", "varBeta", "piHat"):
        assert np.array_equal(a[k], b[k]), k
    assert a["varE"] == b["varE"] and a["b"] == b["b"] and a["iter"] == b["iter"] == 20
    pa, pb = full.get_posterior_sums(), second.get_posterior_sums()
    assert pa["nKept"] == pb["nKept"] == 8
    for k in ("sum_beta", "sum_beta2", "sum_delta", "sum_varBeta", "sum_pi"):
        assert np.array_equal(pa[k], pb[k]), k
    assert pa["sum_varE"] == pb["sum_varE"] and pa["sum_b"] == pb["sum_b"]
    # a snapshot of another model is refused, a truncated file too
    other = ngp.Sampler(device=0, seed=1, chain=0)
    other.set_panel(np.zeros((10, 5), dtype=np.float32)); other.add_marker_set(0, 5, 0, 4.0, 0.1, [(0, 5)], [0.1]); other.set_y(np.zeros(10))
    with pytest.raises(ngp.NextGPHipError, match="does not match"):
        other.load_snapshot(path)
    # the same counts are not enough: a model with BayesC where the snapshot's has BayesB (same N, P, sets; nvb made equal
    # by regions) is refused by the model signature in the header
    twin = ngp.Sampler(device=0, seed=3, chain=0)
    Xs, ys, _, vs = make_problem(O, 200, 192, seed=4)
    twin.set_panel(Xs); add_sets(twin, [(0, 100, "PR"), (100, 92, "PR1")], vs); twin.set_y(ys)   # BayesPR, one region per locus: 92 variances too
    assert twin.nvb == full.nvb
    with pytest.raises(ngp.NextGPHipError, match="methods, classes, regions"):
        twin.load_snapshot(path)
    open(path, "r+b").truncate(100)
    with pytest.raises(ngp.NextGPHipError, match="truncated|magic"):
        second.load_snapshot(path)


def test_set_y_starts_a_fresh_chain(ngp, O):
    """A second chain on the same handle (ngp_set_y again) must not inherit variances, pi or posterior sums of the first."""
    s, X, y, v = _small_model(ngp, O)
    s.set_schedule(10, 2, 1); s.run(10)
    first = (s.get_state(), s.get_posterior_sums())
    s.set_y(y); s.set_schedule(10, 2, 1)
    z = s.get_posterior_sums()
    assert z["nKept"] == 0 and not z["sum_varBeta"].any() and not z["sum_pi"].any() and not z["sum_beta"].any()
    s.run(10)
    second = (s.get_state(), s.get_posterior_sums())
    for k in ("beta", "ycorr", "varBeta", "piHat", "delta"):
        assert np.array_equal(first[0][k], second[0][k]), k
    for k in ("sum_beta", "sum_varBeta", "sum_pi", "sum_delta"):
        assert np.array_equal(first[1][k], second[1][k]), k
    assert first[1]["nKept"] == second[1]["nKept"] == 8


def test_export_posterior_device_equals_packed_host_sums(ngp, O):
    import torch
    s, *_ = _small_model(ngp, O)
    s.set_schedule(12, 2, 2); s.run(12)
    n = s.posterior_len()
    buf = torch.zeros(n, device="cuda", dtype=torch.float64)
    s.export_posterior_device(buf.data_ptr(), n)
    torch.cuda.synchronize()
    exp = ngp.multichain.pack_posterior(s.get_posterior_sums(), s.P, s.nvb, s.nsets)
    assert np.array_equal(buf.cpu().numpy(), exp)
    # with a BayesR set and a fixed-effect set: their sums travel in the packed vector too
    s, X, y, v = _model_with_fixed_and_classes(ngp, O, seed=3)
    s.set_schedule(12, 2, 2); s.run(12)
    n = s.posterior_len()
    buf = torch.zeros(n, device="cuda", dtype=torch.float64)
    s.export_posterior_device(buf.data_ptr(), n)
    torch.cuda.synchronize()
    exp = ngp.multichain.pack_posterior(s.get_posterior_sums(), s.P, s.nvb, s.nsets, class_sums=s.get_class_state(1)["sum_pi"],
                                        fixed_sums=s.get_fixed()["sum_b"])
    assert n == len(exp) == ngp.multichain.posterior_len(s.P, s.nvb, s.nsets, 4, 2)
    assert np.array_equal(buf.cpu().numpy(), exp)
    m = ngp.multichain.unpack_means(exp, s.P, s.nvb, s.nsets, 4, 2)
    assert np.array_equal(m["b_fixed"], s.get_fixed()["sum_b"] / 5) and m["nKept"] == 5


def _model_with_fixed_and_classes(ngp, O, seed):
    """BayesPR set + BayesR set + a two-column fixed-effect block (beyond the intercept)."""
    N, P = 200, 192
    X, y, bt, v = make_problem(O, N, P, seed=4)
    s = ngp.Sampler(device=0, seed=seed, chain=0)
    s.set_panel(X); add_sets(s, [(0, 100, "PR"), (100, 92, "R")], v)
    rng = np.random.default_rng(77)
    s.add_fixed_set(rng.normal(size=(N, 2)))
    s.set_y(y); s.set_residual_prior(4.0, 1.0)
    return s, X, y, v


def test_allreduce_posterior_pools_chains_inside_the_library(ngp, O):
    """ngp_allreduce_posterior with the handles of three chains that share this box's one GPU (added on the device; handles on
    different devices go through ONE RCCL all-reduce -- exercised at round end on the 8-GPU node): afterwards every handle
    holds the sums over all chains."""
    chains, sums = [], []
    for c in range(3):
        s, *_ = _small_model(ngp, O, seed=1001 + c)
        s.set_schedule(10, 2, 2); s.run(10)
        chains.append(s); sums.append(s.get_posterior_sums())
    ngp.Sampler.allreduce_posterior(chains)
    for s in chains:
        p = s.get_posterior_sums()
        assert p["nKept"] == sum(x["nKept"] for x in sums) == 12
        for k in ("sum_beta", "sum_beta2", "sum_delta", "sum_varBeta", "sum_pi"):
            assert np.array_equal(p[k], (sums[0][k] + sums[1][k]) + sums[2][k]), k
        assert p["sum_varE"] == (sums[0]["sum_varE"] + sums[1]["sum_varE"]) + sums[2]["sum_varE"]
    # fixed-effect sums and BayesR class sums are pooled with the rest (every handle ends with the pooled nKept, so a caller that
    # divides ngp_get_fixed's sum_b by nKept -- runLMEM does -- needs them pooled too)
    chains, fx, cl = [], [], []
    for c in range(3):
        s, *_ = _model_with_fixed_and_classes(ngp, O, seed=2001 + c)
        s.set_schedule(10, 2, 2); s.run(10)
        chains.append(s); fx.append(s.get_fixed()["sum_b"]); cl.append(s.get_class_state(1)["sum_pi"])
    ngp.Sampler.allreduce_posterior(chains)
    for s in chains:
        assert s.get_posterior_sums()["nKept"] == 12
        assert np.array_equal(s.get_fixed()["sum_b"], (fx[0] + fx[1]) + fx[2])
        assert np.array_equal(s.get_class_state(1)["sum_pi"], (cl[0] + cl[1]) + cl[2])
    other, *_ = _small_model(ngp, O, seed=9)                  # another model (no fixed set, no classes) is refused
    with pytest.raises(ngp.NextGPHipError, match="do not share one model"):
        ngp.Sampler.allreduce_posterior([chains[0], other])
    single, *_ = _small_model(ngp, O, seed=5)                 # n = 1 is the identity
    single.set_schedule(4, 0, 1); single.run(4)
    before = single.get_posterior_sums()
    ngp.Sampler.allreduce_posterior([single])
    assert np.array_equal(before["sum_beta"], single.get_posterior_sums()["sum_beta"])


def test_traces_of_selected_effects_and_variances(ngp, O):
    s, X, y, v = _small_model(ngp, O)
    loci = [0, 5, 99, 100, 191]
    s.set_trace_loci(loci, n_varBeta=3)
    rows = []
    for it in range(6):
        s.run(1)
        st = s.get_state(); rows.append((st["beta"][loci].copy(), st["varBeta"][:3].copy(), st["piHat"][1::2].copy()))
        t = s.get_trace_ext(1)
        assert np.array_equal(t["beta"][0], rows[-1][0]) and np.array_equal(t["varBeta"][0], rows[-1][1]) and np.array_equal(t["pi"][0], rows[-1][2])
    s.set_y(y); s.run(6)
    t = s.get_trace_ext(6)
    assert np.array_equal(t["beta"], np.array([r[0] for r in rows])) and np.array_equal(t["varBeta"], np.array([r[1] for r in rows]))


def test_diagnostic_mode_is_explicit_and_flagged(ngp, O):
    """Timing modes are a per-handle setting (never the environment); while one is active ngp_run says so."""
    import os
    os.environ["NGP_DEBUG_MODE"] = "3"                       # what round 1 read on every launch: now ignored
    try:
        a, *_ = _small_model(ngp, O); a.run(3)
        b, *_ = _small_model(ngp, O); b.run(3)
        assert np.array_equal(a.get_state()["beta"], b.get_state()["beta"])
    finally:
        del os.environ["NGP_DEBUG_MODE"]
    c, X, y, v = _small_model(ngp, O)
    c.debug_set_mode(5)
    with pytest.raises(ngp.NextGPHipError, match="diagnostic"):
        c.run(2)
    c.debug_set_mode(0)
    c.set_y(y)                                               # a fresh, valid chain again
    c.run(2)
    ref, *_ = _small_model(ngp, O)                           # ... bit for bit the chain a handle that never saw the mode draws
    ref.run(2)
    assert np.array_equal(c.get_state()["beta"], ref.get_state()["beta"]) and c.get_state()["varE"] == ref.get_state()["varE"]


@pytest.mark.parametrize("engine", [(0, 1), (1, 6), (1, 4, 3, 2)], ids=["blocklaunch", "persist_lag6", "rows_lag4"])
def test_fixed_effect_sets_beyond_the_intercept(ngp, O, engine, tmp_path):
    """Covariates and a blocked group of fixed effects on the device (sampleX! for one column, sampleb! with its ridge for a
    block; src/functions.jl:22-53, src/mme.jl:120-152): bit-exact against the blocked oracle, 1e-9 against the reference-order
    oracle, the true effects recovered, and a snapshot carries them."""
    N, P = 300, 200
    X, y, bt, v = make_problem(O, N, P, seed=5)
    rng = np.random.default_rng(3)
    F1 = rng.normal(size=(N, 1))
    F3 = np.column_stack([rng.integers(0, 2, N).astype(float), rng.normal(size=N), rng.normal(size=N) * 3])
    y = y + 2.0 * F1[:, 0] - 1.5 * F3[:, 1]
    s, o = _pair(ngp, O, X, seed=7, chain=0, engine=engine)
    ro = O.Oracle(order=0, seed=7, chain=0); ro.set_panel_f32(X)
    for m in (s, o, ro):
        m.add_fixed_set(F1, lhs0=[0.3], rhs0=[0.1]); m.add_fixed_set(F3)
        add_sets(m, [(0, 120, "PR"), (120, 80, "B")], v)
        m.set_y(y); m.set_residual_prior(4.0, 0.25 * y.var()); m.set_schedule(40, 10, 1); m.run(40)
    a, b, c = s.get_state(), o.get_state(), ro.get_state()
    fa, fb, fc = s.get_fixed(), o.get_fixed(), ro.get_fixed()
    assert np.array_equal(fa["b"], fb["b"]) and np.array_equal(fa["sum_b"], fb["sum_b"])
    assert np.abs(fa["b"] - fc["b"]).max() <= 1e-9 * np.abs(fc["b"]).max() and np.abs(fa["sum_b"] - fc["sum_b"]).max() <= 1e-9 * np.abs(fc["sum_b"]).max()
    assert np.array_equal(a["delta"], b["delta"]) and np.array_equal(a["delta"], c["delta"])
    for k in ("beta", "ycorr", "varBeta"):
        assert np.array_equal(a[k], b[k]), k
        assert np.abs(a[k] - c[k]).max() <= 1e-9 * max(1e-6, np.abs(c[k]).max()), k
    post = fa["sum_b"] / 30
    assert abs(post[0] - 2.0) < 0.4 and abs(post[2] + 1.5) < 0.4 and abs(post[1]) < 0.6 and abs(post[3]) < 0.2
    resid = y - a["b"] - np.column_stack([F1, F3]) @ fa["b"] - s.xbeta(a["beta"])
    assert np.abs(a["ycorr"] - resid).max() <= 1e-9 * np.abs(y).max()
    # resume through a snapshot
    path = str(tmp_path / "fx.ngpsnap")
    s.save_snapshot(path)
    s2 = ngp.Sampler(device=0, seed=1, chain=9, mode=engine[0], lag=engine[1], streamer=engine[3] if len(engine) > 3 else 1)
    if len(engine) > 2:
        s2.set_near(engine[2])
    s2.set_panel(X); s2.add_fixed_set(F1, lhs0=[0.3], rhs0=[0.1]); s2.add_fixed_set(F3)
    add_sets(s2, [(0, 120, "PR"), (120, 80, "B")], v); s2.set_y(y); s2.set_residual_prior(4.0, 0.25 * y.var()); s2.set_schedule(40, 10, 1)
    s2.load_snapshot(path)
    s.set_schedule(50, 10, 1); s2.set_schedule(50, 10, 1)
    s.run(10); s2.run(10)
    assert np.array_equal(s.get_fixed()["b"], s2.get_fixed()["b"]) and np.array_equal(s.get_fixed()["sum_b"], s2.get_fixed()["sum_b"])
    assert np.array_equal(s.get_state()["beta"], s2.get_state()["beta"])


@pytest.mark.parametrize("shards", [100, 0], ids=["disjoint_cus", "oversubscribed_take_turns"])
def test_chains_of_one_process_share_a_device(ngp, O, shards):
    """Several chains on ONE GPU, each in its own host thread (the reference's one-Julia-task-per-chain, here per handle): with
    ngp_set_max_shards their persistent sweeps run side by side on disjoint CUs; grids that do not fit together take turns (the
    library leases CUs per call) instead of waiting for each other's workgroups.  Either way every chain is bit for bit the
    chain it is when it has the device to itself."""
    import threading
    X, y, bt, v = make_problem(O, 3000, 3200, seed=5)

    def build(seed):
        s = ngp.Sampler(device=0, seed=seed, chain=seed - 1001, mode=1, lag=8)
        if shards:
            s.set_max_shards(shards)
        s.set_panel(X)
        add_sets(s, [(0, 2000, "PR"), (2000, 1200, "B")], v)
        s.set_y(y); s.set_residual_prior(4.0, 0.25 * y.var())
        return s

    alone = []
    for seed in (1001, 1002, 1003):
        s = build(seed); s.run(40); alone.append(s.get_state()); s.close()
    chains = [build(seed) for seed in (1001, 1002, 1003)]
    if shards:
        assert all(c.layout()[1] <= shards for c in chains)
        assert [chains[0].shards_for_chains(k) for k in (1, 2, 3, 4)] == [247, 123, 76, 61]   # 256 CUs
    errs = []

    def work(c):
        try:
            for _ in range(4):
                c.run(10)
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ths = [threading.Thread(target=work, args=(c,)) for c in chains[:2]]
    for t in ths: t.start()
    for t in ths: t.join()
    assert not errs, errs
    chains[2].run(10)
    ngp.Sampler.run_many(chains[2:] + [build(1004)], 30)      # the same through the library's own threads (ngp_run_many)
    for c, ref in zip(chains, alone):
        st = c.get_state()
        for k in ("ycorr", "beta", "delta", "varBeta", "piHat"):
            assert np.array_equal(st[k], ref[k]), k
        assert st["varE"] == ref["varE"] and st["iter"] == 40


def test_census_of_a_sweep_and_resume_after_a_failed_one(ngp, O):
    """Every launch of the persistent kernel opens with a census of its own grid.  (1) An undisturbed launch: every workgroup
    reported in, spread over the 8 XCDs.  (2) A launch whose census fails (forced here through ngp_debug_fail_census: the state a
    grid that is not co-resident reaches after 20 ms) ends before any role has run, the kernels queued behind it return at once,
    and the call resumes that iteration with the device to itself: the chain ends bit for bit where an undisturbed one ends."""
    ref, X, y, v = _small_model(ngp, O)
    ref.set_schedule(40, 4, 2); ref.run(40)
    c = ref.census()
    assert c["grid"] == len(c["xcc"]) and (c["xcc"] >= 0).all() and c["retries"] == 0 and not c["exclusive"]
    assert len(set(c["xcc"].tolist())) == min(8, c["grid"])
    for fail_at in (1, 19, 40):                       # first launch of the call, the middle of a queue of 16, the last one
        s, *_ = _small_model(ngp, O)
        s.set_schedule(40, 4, 2)
        s.debug_fail_census(fail_at)
        s.run(40)
        cs = s.census()
        assert cs["retries"] == 1 and cs["exclusive"]
        a, b = ref.get_state(), s.get_state()
        for k in ("ycorr", "beta", "delta", "varBeta", "piHat"):
            assert np.array_equal(a[k], b[k]), (fail_at, k)
        assert a["varE"] == b["varE"] and a["b"] == b["b"] and a["iter"] == b["iter"] == 40
        pa, pb = ref.get_posterior_sums(), s.get_posterior_sums()
        assert pa["nKept"] == pb["nKept"] and np.array_equal(pa["sum_beta"], pb["sum_beta"]) and pa["sum_varE"] == pb["sum_varE"]
        assert np.array_equal(ref.get_trace(40)["varE"], s.get_trace(40)["varE"])
    # the fine seam takes the same way out
    s, *_ = _small_model(ngp, O)
    t, *_ = _small_model(ngp, O)
    st = s.get_state()
    res = []
    for m, forced in ((s, True), (t, False)):
        yc, be, vb, pi = st["ycorr"].copy(), np.zeros(100), np.array([v]), np.array([0.5, 0.5])
        if forced:
            m.debug_fail_census(1)
        m.sweep_set(0, 1.0, yc, be, vb, pi)
        res.append((yc, be, vb))
    assert all(np.array_equal(p, q) for p, q in zip(res[0], res[1])) and s.census()["retries"] == 1


def test_allreduce_multi_device_branch_on_one_gpu(ngp, O):
    """ngp_allreduce_posterior's multi-device branch (one leader per device, the other handles of a device added into it, ONE
    collective over the leaders, every handle unpacking its leader's buffer) driven on this box's one GPU: five handles grouped
    under three virtual devices.  Everything but the ncclAllReduce call itself is the code an 8-GPU node runs."""
    chains, sums, fx = [], [], []
    vdevs = [0, 1, 0, 2, 1]
    for c, vd in enumerate(vdevs):
        s, *_ = _model_with_fixed_and_classes(ngp, O, seed=3001 + c)
        s.set_schedule(8, 2, 2); s.run(8)
        s.debug_set_virtual_device(vd)
        chains.append(s); sums.append(s.get_posterior_sums()); fx.append(s.get_fixed()["sum_b"])
    ngp.Sampler.allreduce_posterior(chains)
    # leaders 0 (+2), 1 (+4), 3; collective adds leader buffers in leader order
    tot = lambda f: ((f(0) + f(2)) + (f(1) + f(4))) + f(3)
    for s in chains:
        p = s.get_posterior_sums()
        assert p["nKept"] == 15
        for k in ("sum_beta", "sum_beta2", "sum_delta", "sum_varBeta", "sum_pi"):
            assert np.array_equal(p[k], tot(lambda i: sums[i][k])), k
        assert p["sum_varE"] == tot(lambda i: sums[i]["sum_varE"])
        assert np.array_equal(s.get_fixed()["sum_b"], tot(lambda i: fx[i]))


def test_sample_stream_does_not_stop_the_chain_and_loses_nothing(ngp, O, tmp_path):
    """ngp_set_sample_file: ONE ngp_run of the whole chain; every kept iteration leaves a binary record (ring of four slots on the device,
    second stream, writer thread) -- the values the stop-and-copy read-back sees at those iterations, bit for bit.  A launch that
    ends at its census in the middle (records of the skipped iterations are marked invalid and dropped) is run again: no record is
    lost or doubled."""
    ref, X, y, v = _small_model(ngp, O)
    ref.set_schedule(40, 4, 3)
    want = {}
    for it in range(4 + 3, 41, 3):
        ref.run(it - ref.get_state()["iter"])
        st = ref.get_state()
        want[it] = (st["beta"].copy(), st["delta"].copy(), st["varBeta"].copy(), st["piHat"].copy(), st["varE"], st["b"])
    for fail_at in (0, 17):
        s, *_ = _small_model(ngp, O)
        s.set_schedule(40, 4, 3)
        path = str(tmp_path / f"s{fail_at}.ngpsmp")
        s.set_sample_file(path)
        if fail_at:
            s.debug_fail_census(fail_at)
        s.run(40)
        s.set_sample_file(None)
        S = ngp.read_sample_file(path)
        assert S["iter"].tolist() == sorted(want), fail_at
        for i, it in enumerate(S["iter"].tolist()):
            b, d, vb, pi, vE, bb = want[it]
            assert np.array_equal(S["beta"][i], b) and np.array_equal(S["delta"][i], d.astype(np.uint8)) and np.array_equal(S["varBeta"][i], vb)
            assert np.array_equal(S["piHat"][i], pi) and S["varE"][i] == vE and S["b"][i] == bb
        assert [x["method"] for x in S["sets"]] == [0, 1] and S["sets"][1]["col0"] == 100
    # many more samples than ring slots, every iteration kept
    s, *_ = _small_model(ngp, O)
    s.set_schedule(300, 0, 1)
    s.set_sample_file(str(tmp_path / "all.ngpsmp"))
    s.run(300)
    s.set_sample_file(None)
    S = ngp.read_sample_file(str(tmp_path / "all.ngpsmp"))
    assert S["iter"].tolist() == list(range(1, 301)) and np.array_equal(S["beta"][-1], s.get_state()["beta"])
    assert np.allclose(S["beta"].sum(axis=0), s.get_posterior_sums()["sum_beta"], rtol=1e-12, atol=1e-12)
