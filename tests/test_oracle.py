"""CPU tests of the oracle itself (no GPU): draw layer against scipy, closed forms, the two
orderings of the chain against each other, committed golden vectors."""
import math
import os

import numpy as np
import pytest
from scipy import stats

from conftest import add_sets, make_problem

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_det_log_within_one_ulp(O):
    rng = np.random.default_rng(0)
    xs = np.concatenate([10.0 ** rng.uniform(-300, 300, 5000), rng.uniform(0.5, 2.0, 5000), [1.0, 2.0, 4.9e-324, 1e-310]])
    for x in xs:
        ref = math.log(x)
        assert abs(O.det_log(x) - ref) <= 1.01 * np.spacing(abs(ref)) + 1e-323
    assert O.det_log(1.0) == 0.0 and O.det_log(0.0) == -math.inf and math.isnan(O.det_log(-1.0))


def test_ppnd16_matches_scipy(O):
    ps = np.concatenate([np.random.default_rng(1).uniform(0, 1, 5000), [1e-300, 1e-20, 0.075, 0.0749, 0.925, 0.5, 1 - 1e-16]])
    for p in ps:
        ref = stats.norm.ppf(p)
        assert abs(O.ppnd16(p) - ref) <= 2e-15 * max(1.0, abs(ref))
    assert O.ppnd16(0.5) == 0.0


def test_streams_are_keyed_and_reproducible(O):
    a = O.draws(5, 1, 9, 3, 77, 1, 100)
    assert np.array_equal(a, O.draws(5, 1, 9, 3, 77, 1, 100))
    for other in [(6, 1, 9, 3, 77), (5, 2, 9, 3, 77), (5, 1, 10, 3, 77), (5, 1, 9, 4, 77), (5, 1, 9, 3, 78)]:
        assert not np.array_equal(a, O.draws(*other, 1, 100))
    # first draw of consecutive indices: no serial correlation
    z = O.draws(5, 1, 9, 3, 0, 1, 50000, indexed=True)
    assert abs(np.corrcoef(z[:-1], z[1:])[0, 1]) < 0.02 and abs(z.mean()) < 0.02 and abs(z.var() - 1) < 0.03
    u = O.draws(5, 1, 9, 5, 0, 0, 50000, indexed=True)
    assert 0.0 < u.min() and u.max() < 1.0


@pytest.mark.parametrize("what,p1,p2,dist,args", [
    (0, 0, 0, "uniform", ()), (1, 0, 0, "norm", ()), (2, 5.0, 0, "chi2", (5.0,)), (2, 10004.0, 0, "chi2", (10004.0,)),
    (3, 3.0, 98.0, "beta", (3.0, 98.0)), (4, 1.0, 0, "gamma", (1.0,)), (4, 2.5, 0, "gamma", (2.5,))])
def test_draw_distributions(O, what, p1, p2, dist, args):
    x = O.draws(11, 0, 1, 1, 0, what, 40000, p1, p2)
    assert stats.kstest(x, dist, args=args).pvalue > 1e-3


def test_panel_generator(O):
    X, mu = O.generate_panel(400, 50, seed=3)
    assert X.dtype == np.float32 and X.shape == (400, 50)
    raw = X.astype(np.float64) + mu
    assert np.abs(raw - np.rint(raw)).max() < 1e-6 and set(np.unique(np.rint(raw))) <= {0.0, 1.0, 2.0}
    assert np.abs(X.astype(np.float64).sum(axis=0)).max() < 1e-3      # centred
    assert 0.02 <= mu.min() and mu.max() <= 1.15                      # sample mean of 2p, p in [0.05, 0.5]
    X2, _ = O.generate_panel(400, 50, seed=3)
    assert np.array_equal(X, X2)


def test_single_snp_conjugate_posterior(O):
    """One SNP, variances effectively fixed: beta | rest ~ N(x'y/(x'x+varE/varB), varE/(x'x+varE/varB))."""
    N = 50
    rng = np.random.default_rng(0)
    x = rng.normal(size=(N, 1)).astype(np.float32)
    x -= x.mean()
    y = 0.7 * x[:, 0].astype(np.float64) + rng.normal(size=N)
    big = 1e12  # huge prior df pins varE and varBeta at their scales
    varE, varB = 1.3, 0.4
    means = []
    for chain in range(400):
        o = O.Oracle(0, seed=2, chain=chain)
        o.set_panel_f32(x)
        o.add_marker_set(0, 1, 0, big, varB, [(0, 1)], [varB])
        o.set_y(y); o.set_intercept(False); o.set_residual_prior(big, varE); o.run(1)
        means.append(o.get_state()["beta"][0])
    xx = float((x.astype(np.float64) ** 2).sum()); xy = float(x[:, 0].astype(np.float64) @ y)
    lhs = xx + varE / varB
    m, sd = xy / lhs, math.sqrt(varE / lhs)
    assert abs(np.mean(means) - m) < 4 * sd / math.sqrt(400)
    assert abs(np.std(means) - sd) < 0.15 * sd


def test_ridge_limit(O):
    """Variances pinned: the long-run mean of beta is the ridge / BLUP solution (X'X + lambda I)^-1 X'y."""
    N, P = 120, 30
    X, y, bt, v = make_problem(O, N, P, seed=3)
    big, varE, varB = 1e12, 1.0, 0.05
    o = O.Oracle(0, seed=8, chain=0)
    o.set_panel_f32(X)
    o.add_marker_set(0, P, 0, big, varB, [(0, P)], [varB])
    o.set_y(y); o.set_residual_prior(big, varE); o.set_schedule(4000, 500, 1); o.run(4000)
    ps = o.get_posterior_sums()
    Xd = X.astype(np.float64)
    yc = y - y.mean()
    ridge = np.linalg.solve(Xd.T @ Xd + (varE / varB) * np.eye(P), Xd.T @ yc)
    mean = ps["sum_beta"] / ps["nKept"]
    sd = np.sqrt(np.maximum(ps["sum_beta2"] / ps["nKept"] - mean ** 2, 1e-12))
    assert ps["nKept"] == 3500
    assert np.abs(mean - ridge).max() < 6 * sd.max() / math.sqrt(3500 / 10)   # autocorrelation allowance
    assert abs(ps["sum_b"] / ps["nKept"] - y.mean()) < 0.05


@pytest.mark.parametrize("spec", [[(0, 300, "PR")], [(0, 300, "B")], [(0, 300, "C")], [(0, 300, "Cfix")],
                                  [(0, 100, "PR"), (100, 120, "B"), (220, 80, ("PRw", 17))], [(0, 140, "C"), (140, 160, "B")]],
                         ids=["PR", "B", "C", "Cfix", "multi", "multiC"])
def test_blocked_order_equals_reference_order(O, spec):
    N, P = 257, 300
    X, y, bt, v = make_problem(O, N, P, seed=6)
    a = O.Oracle(0, seed=42, chain=1); a.set_panel_f32(X)
    b = O.Oracle(1, seed=42, chain=1); b.set_panel_f32(X, R=12, S=22)
    for m in (a, b):
        add_sets(m, spec, v); m.set_y(y); m.set_residual_prior(4.0, 0.25 * y.var()); m.set_schedule(40, 10, 3); m.run(40)
    sa, sb = a.get_state(), b.get_state()
    assert np.array_equal(sa["delta"], sb["delta"])                      # bit-exact indicators
    for k in ("beta", "ycorr", "varBeta", "piHat"):
        assert np.abs(sa[k] - sb[k]).max() <= 1e-10 * max(1e-3, np.abs(sa[k]).max()), k
    assert abs(sa["varE"] - sb["varE"]) <= 1e-10 * sa["varE"]
    # residual invariant of both
    for s in (sa, sb):
        assert np.abs(s["ycorr"] - (y - s["b"] - X.astype(np.float64) @ s["beta"])).max() < 1e-10
    assert a.get_posterior_sums()["nKept"] == b.get_posterior_sums()["nKept"] == 10


@pytest.mark.parametrize("D,near", [(2, 3), (4, 3), (6, 3), (8, 3), (5, 4), (8, 4)])
def test_look_ahead_is_the_same_chain(O, D, near):
    """Lag D with the corrections split between sampler (lags 1..near) and reducers (the rest): same chain as the reference
    order, whatever the split."""
    N, P = 130, 64 * 11 + 5
    X, y, bt, v = make_problem(O, N, P, seed=13)
    a = O.Oracle(0, seed=5, chain=2); a.set_panel_f32(X)
    b = O.Oracle(1, seed=5, chain=2); b.set_panel_f32(X, R=8, S=17, D=D, near=near)
    for m in (a, b):
        add_sets(m, [(0, 400, "PR"), (400, P - 400, "B")], v); m.set_y(y); m.set_residual_prior(4.0, 0.25 * y.var()); m.run(12)
    sa, sb = a.get_state(), b.get_state()
    assert np.array_equal(sa["delta"], sb["delta"])
    assert np.abs(sa["beta"] - sb["beta"]).max() <= 1e-10 * max(1e-3, np.abs(sa["beta"]).max())
    assert np.abs(sa["ycorr"] - sb["ycorr"]).max() <= 1e-10 * np.abs(sa["ycorr"]).max()


def test_layout_independence_of_blocked_order(O):
    """Different shard layouts change only the summation tree: results agree to rounding, indicators exactly."""
    N, P = 200, 128
    X, y, bt, v = make_problem(O, N, P, seed=7)
    out = []
    for R, S in ((4, 50), (20, 10), (200, 1)):
        o = O.Oracle(1, seed=3, chain=0); o.set_panel_f32(X, R=R, S=S)
        add_sets(o, [(0, P, "B")], v); o.set_y(y); o.set_residual_prior(4.0, 1.0); o.run(15)
        out.append(o.get_state())
    for s in out[1:]:
        assert np.array_equal(s["delta"], out[0]["delta"])
        assert np.abs(s["beta"] - out[0]["beta"]).max() < 1e-11


def test_bayesb_quirk_excluded_locus_has_zero_variance(O):
    """functions.jl:184-186: an excluded locus gets beta = 0, delta = 0, varBeta = 0."""
    N, P = 150, 64
    X, y, bt, v = make_problem(O, N, P, seed=8)
    for order, kw in ((0, {}), (1, dict(R=4, S=38))):
        o = O.Oracle(order, seed=5, chain=0); o.set_panel_f32(X, **kw)
        add_sets(o, [(0, P, "Bfix")], v); o.set_y(y); o.set_residual_prior(4.0, 1.0); o.run(8)
        s = o.get_state()
        out = s["delta"] == 0
        assert out.any() and (~out).any()
        assert np.all(s["beta"][out] == 0.0) and np.all(s["varBeta"][out] == 0.0) and np.all(s["varBeta"][~out] > 0.0)


def test_bayesc_one_variance_and_inclusion_count(O):
    """functions.jl:197-235: BayesC keeps ONE variance per set; excluded loci have beta = 0; with pi fixed and a long chain
    the mean of varBeta follows its conditional (scale df + sum beta^2) / chisq(df + nLoci) -- checked through the identity
    E[(scale df + ssq) / varBeta] = df + nLoci on the kept draws (both sides are known per iteration)."""
    N, P = 120, 48
    X, y, bt, v = make_problem(O, N, P, seed=9)
    for order, kw in ((0, {}), (1, dict(R=8, S=15))):
        o = O.Oracle(order, seed=11, chain=0); o.set_panel_f32(X, **kw)
        add_sets(o, [(0, P, "Cfix")], v); o.set_y(y); o.set_residual_prior(4.0, 1.0)
        ratios = []
        for _ in range(400):
            o.run(1)
            s = o.get_state()
            assert s["varBeta"].shape == (1,) and s["varBeta"][0] > 0
            out = s["delta"] == 0
            assert np.all(s["beta"][out] == 0.0)
            nl = int((~out).sum())
            df = 4.0
            ratios.append(((v * (df - 2) / df) * df + float(s["beta"] @ s["beta"])) / s["varBeta"][0] - (df + nl))
        ratios = np.array(ratios)
        # chi-square(nu) - nu has mean 0 and variance 2 nu (nu ~ 4 + nLoci <= 52)
        assert abs(ratios.mean()) < 5 * math.sqrt(2 * 52 / len(ratios))


def test_schedule_keeps_reference_iterations(O):
    """samplers.jl:26: kept = (burnIn+thin):thin:chainLength."""
    X, y, bt, v = make_problem(O, 40, 8, seed=1)
    o = O.Oracle(0, seed=1, chain=0); o.set_panel_f32(X); add_sets(o, [(0, 8, "PR")], v); o.set_y(y)
    o.set_schedule(23, 5, 4); o.run(30)
    assert o.get_posterior_sums()["nKept"] == len(range(5 + 4, 23 + 1, 4))


@pytest.mark.parametrize("name", ["pr_50x200", "b_50x200", "c_50x200"])
def test_golden_vectors(O, name):
    """Committed fixtures generated by tests/golden/make_golden.py from the reference-order oracle."""
    g = np.load(os.path.join(GOLD, name + ".npz"))
    X = np.asfortranarray(g["X"])
    for order, kw, tol in ((0, {}, 0.0), (1, dict(R=4, S=13), 1e-11)):
        o = O.Oracle(order, seed=int(g["seed"]), chain=int(g["chain"])); o.set_panel_f32(X, **kw)
        kind = {"pr": "PR", "b_": "B", "c_": "C"}[name[:2]]
        add_sets(o, [(0, X.shape[1], kind)], float(g["v"])); o.set_y(g["y"]); o.set_residual_prior(4.0, float(g["e_scale"]))
        done = 0
        for it in (1, 2, 10):
            o.run(it - done); done = it
            s = o.get_state()
            assert np.array_equal(s["delta"], g[f"delta_{it}"])
            assert np.abs(s["beta"] - g[f"beta_{it}"]).max() <= tol * max(1.0, np.abs(g[f"beta_{it}"]).max())
            assert abs(s["varE"] - float(g[f"varE_{it}"])) <= tol * float(g[f"varE_{it}"])
            assert np.abs(s["varBeta"] - g[f"varBeta_{it}"]).max() <= tol * max(1e-9, np.abs(g[f"varBeta_{it}"]).max())


# ----------------------------------------------------------------------------------------------
# independent numpy restatement of the reference (tests/ref_numpy.py) against the C oracle
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("spec,intercept", [
    ([(0, 45, "PR")], True), ([(0, 45, ("PRw", 7))], True), ([(0, 45, "PR1")], False), ([(0, 45, "B")], True), ([(0, 45, "Bfix")], True),
    ([(0, 45, "C")], True), ([(0, 15, "PR"), (15, 18, "B"), (33, 12, "Cfix")], True)],
    ids=["pr", "pr_regions", "pr_r1_no_intercept", "b", "b_fixpi", "c", "multi"])
def test_numpy_restatement_of_the_reference_agrees(O, spec, intercept):
    """The C oracle's reference-order path and a separate numpy transcription of functions.jl:118-235 + samplers.jl:29-53
    consume the same draws and must produce the same chain (indicators identical, floats to rounding: numpy's dots are
    summed pairwise, the oracle's in eight interleaved chains)."""
    from ref_numpy import RefChain
    N, P = 60, 45
    X, y, bt, v = make_problem(O, N, P, seed=3)
    rng = np.random.default_rng(8)
    o = O.Oracle(0, seed=21, chain=2); o.set_panel_f32(X)
    ref = RefChain(O, X.astype(np.float64), y, seed=21, chain=2, intercept=intercept)
    df = 4.0
    for col0, ncol, kind in spec:
        lhs0 = np.abs(rng.normal(size=ncol)) * 0.2
        rhs0 = rng.normal(size=ncol) * 0.1
        if kind == "PR": args = (0, [(0, ncol)], [v], 0.0, False)
        elif kind == "PR1": args = (0, [(j, j + 1) for j in range(ncol)], [v] * ncol, 0.0, False)
        elif isinstance(kind, tuple): args = (0, [(a, min(a + kind[1], ncol)) for a in range(0, ncol, kind[1])], None, 0.0, False)
        elif kind == "B": args = (1, [(j, j + 1) for j in range(ncol)], [v] * ncol, 0.05, True)
        elif kind == "Bfix": args = (1, [(j, j + 1) for j in range(ncol)], [v] * ncol, 0.3, False)
        elif kind == "C": args = (2, [(0, ncol)], [v], 0.1, True)
        else: args = (2, [(0, ncol)], [v], 0.4, False)
        method, regs, vb0, pi0, estPi = args
        vb0 = [v] * len(regs) if vb0 is None else vb0
        o.add_marker_set(col0, ncol, method, df, v * (df - 2) / df, regs, vb0, pi0=pi0, estPi=estPi, lhs0=lhs0, rhs0=rhs0)
        ref.add_set(col0, ncol, method, df, v * (df - 2) / df, regs, vb0, pi0=pi0, estPi=estPi, lhs=lhs0, rhs=rhs0)
    o.set_y(y); o.set_intercept(intercept); o.set_residual_prior(4.0, 0.3 * y.var())
    ref.E_df, ref.E_scale = 4.0, 0.3 * y.var()
    for it in range(8):
        o.run(1); ref.run(1)
        a, b = o.get_state(), ref.state()
        assert np.array_equal(a["delta"], b["delta"]), it
        for k in ("beta", "ycorr", "varBeta", "piHat"):
            assert np.abs(a[k] - b[k]).max() <= 1e-10 * max(1e-6, np.abs(b[k]).max()), (it, k)
        assert abs(a["varE"] - b["varE"]) <= 1e-12 * b["varE"] and abs(a["b"] - b["b"]) <= 1e-10 * max(1.0, abs(b["b"]))


def test_hand_derived_one_snp_chain(O):
    """A chain small enough to write out by hand from functions.jl:128-135 and :523-525 with the draws fixed: one SNP, no
    intercept.  Every number below is computed here with plain Python floats from the formulas, not by the oracle."""
    N = 6
    x = np.array([1.0, -1.0, 0.5, -0.5, 2.0, -2.0], dtype=np.float32)
    y = np.array([0.3, -0.2, 0.1, 0.4, 1.1, -0.9])
    o = O.Oracle(0, seed=4, chain=0); o.set_panel_f32(x.reshape(N, 1))
    df, scale, vb0, e_df, e_scale = 4.0, 0.02, 0.05, 4.0, 0.1
    o.add_marker_set(0, 1, 0, df, scale, [(0, 1)], [vb0]); o.set_y(y); o.set_intercept(False); o.set_residual_prior(e_df, e_scale)
    o.run(2)
    ycorr, beta, vb = [float(t) for t in y], 0.0, vb0
    xs = [float(t) for t in x]
    for it in (1, 2):
        chi_e = float(O.draws(4, 0, it, 1, 0, 2, 1, e_df + N, indexed=True)[0])
        z = float(O.draws(4, 0, it, 3, 0, 1, 1, indexed=True)[0])
        chi_b = float(O.draws(4, 0, it, 4, 0, 2, 1, df + 1, indexed=True)[0])
        varE = (e_df * e_scale + sum(t * t for t in ycorr)) / chi_e                  # :523-525
        ycorr = [t + beta * xi for t, xi in zip(ycorr, xs)]                           # :128
        rhs = sum(xi * t for xi, t in zip(xs, ycorr)) / varE                          # :129
        lhs = sum(xi * xi for xi in xs) / varE + 1.0 / vb                             # :130
        beta = rhs / lhs + math.sqrt(1.0 / lhs) * z                                   # :131-132, :493-495
        ycorr = [t - beta * xi for t, xi in zip(ycorr, xs)]                           # :133
        vb = (scale * df + beta * beta) / chi_b                                       # :135, :509-511
    st = o.get_state()
    assert abs(st["beta"][0] - beta) <= 1e-14 * abs(beta) and abs(st["varBeta"][0] - vb) <= 1e-14 * vb and abs(st["varE"] - varE) <= 1e-14 * varE
    assert np.abs(st["ycorr"] - np.array(ycorr)).max() <= 1e-14


@pytest.mark.parametrize("kind", ["PR", "B"])
def test_fp32_panel_deviation(O, kind):
    """How far does the product's fp32 panel (centred value rounded to fp32) move the chain away from the reference's Float64
    panel (prepMatVec.jl:129)?  BASELINE.json configs[0] size: 500 x 5,000, 1,000 iterations, same draws.  BayesPR is a smooth
    function of the data: the chains stay together to 1e-6 of the effect scale over the whole run.  BayesB compares a
    uniform draw with an inclusion probability: a locus whose draw falls within the rounding difference flips, and from
    there the two chains are different realisations of the same posterior -- reported as the first flip and the Monte-Carlo
    size difference of the posterior means ("fp32 panel")."""
    N, P, niter = 500, 5000, 1000
    X32, mu = O.generate_panel(N, P)
    G = np.rint(X32.astype(np.float64) + mu)                    # the 0/1/2 genotypes
    X64 = G - G.mean(axis=0)                                     # centred in Float64, as the reference does
    assert np.abs(X64 - X32).max() < 1e-6
    y = 10.0 + X64[:, ::97] @ np.random.default_rng(1).normal(size=len(range(0, P, 97))) + np.random.default_rng(2).normal(size=N) * 3.0
    v = 0.5 * y.var() / float((mu * (1 - mu / 2)).sum())
    chains = []
    for panel in ("f32", "f64"):
        o = O.Oracle(0, seed=1001, chain=0)
        if panel == "f32": o.set_panel_f32(X32)
        else: o.set_panel_f64(X64)
        add_sets(o, [(0, P, kind)], v); o.set_y(y); o.set_residual_prior(4.0, 0.25 * y.var()); o.set_schedule(niter, 200, 1)
        first_flip, deltas = None, []
        if kind == "B":
            for it in range(niter // 50):
                o.run(50); deltas.append(o.get_state()["delta"].copy())
        else:
            o.run(niter)
        chains.append((o.get_state(), o.get_posterior_sums(), deltas))
    (sa, pa, da), (sb, pb, db) = chains
    scale = np.abs(pb["sum_beta"] / pb["nKept"]).max()
    dmean = np.abs(pa["sum_beta"] / pa["nKept"] - pb["sum_beta"] / pb["nKept"]).max()
    if kind == "PR":
        print(f"fp32 vs Float64 panel, BayesPR 500 x 5000 x 1000 it: max |d posterior mean| = {dmean:.3e} (scale {scale:.3e}), "
              f"max |d beta| at the end = {np.abs(sa['beta'] - sb['beta']).max():.3e}, d varE = {abs(sa['varE'] - sb['varE']) / sb['varE']:.3e}")
        assert dmean <= 1e-6 * scale and np.abs(sa["beta"] - sb["beta"]).max() <= 1e-5 * scale
    else:
        same = [np.array_equal(x, z) for x, z in zip(da, db)]
        first = (same.index(False) + 1) * 50 if False in same else None
        flips = int((da[-1] != db[-1]).sum())
        sd = np.sqrt(np.maximum(pb["sum_beta2"] / pb["nKept"] - (pb["sum_beta"] / pb["nKept"]) ** 2, 0)).max()
        print(f"fp32 vs Float64 panel, BayesB 500 x 5000 x 1000 it: indicators identical through iteration "
              f"{'1000 (no flip)' if first is None else first - 50}, differing indicators at the end {flips} of {P}, "
              f"max |d posterior mean| = {dmean:.3e} (largest posterior sd {sd:.3e})")
        assert same[0]                                            # the first 50 iterations agree locus by locus
        assert dmean <= 1.0 * sd + 1e-12                          # afterwards: two realisations of one posterior


def test_numpy_restatement_bayesr(O):
    """BayesR (functions.jl:238-289) in the numpy restatement against both orders of the C oracle: classes identical, floats to
    rounding; the fresh-uniform-per-comparison class search (:261) is part of what is compared."""
    from ref_numpy import RefChain
    N, P = 80, 70
    X, y, bt, v = make_problem(O, N, P, seed=6)
    vcl, pi0 = [0.0, 0.01, 0.1, 1.0], [0.8, 0.12, 0.06, 0.02]
    o0 = O.Oracle(0, seed=9, chain=1); o0.set_panel_f32(X)
    o1 = O.Oracle(1, seed=9, chain=1); o1.set_panel_f32(X, R=4, S=20, D=3, near=3)
    ref = RefChain(O, X.astype(np.float64), y, seed=9, chain=1)
    for m in (o0, o1):
        m.add_marker_set(0, 20, 0, 4.0, v * 0.5, [(0, 20)], [v])
        m.add_marker_set_r(20, 50, 4.0, v * 0.5, v, vcl, pi0, estPi=True)
        m.set_y(y); m.set_residual_prior(4.0, 0.3 * y.var())
    ref.add_set(0, 20, 0, 4.0, v * 0.5, [(0, 20)], [v]); ref.add_set_r(20, 50, 4.0, v * 0.5, v, vcl, pi0, estPi=True)
    ref.E_df, ref.E_scale = 4.0, 0.3 * y.var()
    for it in range(10):
        o0.run(1); o1.run(1); ref.run(1)
        a, b, c = o0.get_state(), o1.get_state(), ref.state()
        assert np.array_equal(a["delta"], c["delta"]) and np.array_equal(a["delta"], b["delta"]), it
        for k in ("beta", "ycorr", "varBeta"):
            assert np.abs(a[k] - c[k]).max() <= 1e-10 * max(1e-6, np.abs(c[k]).max()), (it, k)
            assert np.abs(a[k] - b[k]).max() <= 1e-10 * max(1e-6, np.abs(a[k]).max()), (it, k)
        assert np.abs(o0.get_class_state(1)["piHat"] - c["class_pi"][1]).max() <= 1e-12
        assert np.abs(o1.get_class_state(1)["piHat"] - c["class_pi"][1]).max() <= 1e-12
    cls = a["delta"][20:]
    assert set(np.unique(cls)) <= {1, 2, 3, 4}
    assert np.all(a["beta"][20:][cls == 1] == 0.0)                                # :275


def test_bayesr_with_more_than_four_classes(O):
    """The reference sizes BayesR by length(vClass) (functions.jl:241-262); six and eight classes in both oracle orders and in the numpy
    restatement: identical classes, floats to rounding."""
    from ref_numpy import RefChain
    N, P = 90, 120
    X, y, bt, v = make_problem(O, N, P, seed=16)
    vc6, pi6 = [0.0, 0.0001, 0.001, 0.01, 0.1, 1.0], [0.5, 0.2, 0.12, 0.1, 0.05, 0.03]
    vc8, pi8 = [1e-5, 1e-4, 1e-3, 0.01, 0.05, 0.2, 0.5, 1.0], [0.3, 0.2, 0.15, 0.1, 0.1, 0.06, 0.05, 0.04]
    o0 = O.Oracle(0, seed=19, chain=0); o0.set_panel_f32(X)
    o1 = O.Oracle(1, seed=19, chain=0); o1.set_panel_f32(X, R=8, S=12, D=4, near=3)
    ref = RefChain(O, X.astype(np.float64), y, seed=19, chain=0)
    for m in (o0, o1):
        m.add_marker_set_r(0, 70, 4.0, v * 0.5, v, vc6, pi6, estPi=True)
        m.add_marker_set_r(70, 50, 4.0, v * 0.5, v, vc8, pi8, estPi=False)
        m.set_y(y); m.set_residual_prior(4.0, 0.3 * y.var())
    ref.add_set_r(0, 70, 4.0, v * 0.5, v, vc6, pi6, estPi=True); ref.add_set_r(70, 50, 4.0, v * 0.5, v, vc8, pi8, estPi=False)
    ref.E_df, ref.E_scale = 4.0, 0.3 * y.var()
    for it in range(8):
        o0.run(1); o1.run(1); ref.run(1)
        a, b, c = o0.get_state(), o1.get_state(), ref.state()
        assert np.array_equal(a["delta"], c["delta"]) and np.array_equal(a["delta"], b["delta"]), it
        for k in ("beta", "ycorr", "varBeta"):
            assert np.abs(a[k] - c[k]).max() <= 1e-10 * max(1e-6, np.abs(c[k]).max()), (it, k)
            assert np.abs(a[k] - b[k]).max() <= 1e-10 * max(1e-6, np.abs(a[k]).max()), (it, k)
    assert a["delta"][:70].max() <= 6 and a["delta"][70:].max() <= 8 and len(np.unique(a["delta"])) > 4
    assert len(o1.get_class_state(0)["piHat"]) == 6 and abs(o1.get_class_state(0)["piHat"].sum() - 1) < 1e-12

def test_bayesr_with_twelve_and_sixteen_classes(O):
    """... and twelve / sixteen classes (the device reads classes 9..16 from memory): both oracle orders and the numpy restatement."""
    from ref_numpy import RefChain
    N, P = 90, 120
    X, y, bt, v = make_problem(O, N, P, seed=17)
    vc12 = [0.0, 1e-5, 3e-5, 1e-4, 3e-4, 1e-3, 3e-3, 0.01, 0.03, 0.1, 0.3, 1.0]
    pi12 = [0.4, 0.1, 0.08, 0.08, 0.07, 0.06, 0.05, 0.05, 0.04, 0.03, 0.02, 0.02]
    vc16, pi16 = [2.0 ** (i - 15) for i in range(16)], [1.0 / 16] * 16
    o0 = O.Oracle(0, seed=23, chain=0); o0.set_panel_f32(X)
    o1 = O.Oracle(1, seed=23, chain=0); o1.set_panel_f32(X, R=8, S=12, D=4, near=3)
    ref = RefChain(O, X.astype(np.float64), y, seed=23, chain=0)
    for m in (o0, o1):
        m.add_marker_set_r(0, 70, 4.0, v * 0.5, v, vc12, pi12, estPi=True)
        m.add_marker_set_r(70, 50, 4.0, v * 0.5, v, vc16, pi16, estPi=False)
        m.set_y(y); m.set_residual_prior(4.0, 0.3 * y.var())
    ref.add_set_r(0, 70, 4.0, v * 0.5, v, vc12, pi12, estPi=True); ref.add_set_r(70, 50, 4.0, v * 0.5, v, vc16, pi16, estPi=False)
    ref.E_df, ref.E_scale = 4.0, 0.3 * y.var()
    for it in range(6):
        o0.run(1); o1.run(1); ref.run(1)
        a, b, c = o0.get_state(), o1.get_state(), ref.state()
        assert np.array_equal(a["delta"], c["delta"]) and np.array_equal(a["delta"], b["delta"]), it
        for k in ("beta", "ycorr", "varBeta"):
            assert np.abs(a[k] - c[k]).max() <= 1e-10 * max(1e-6, np.abs(c[k]).max()), (it, k)
            assert np.abs(a[k] - b[k]).max() <= 1e-10 * max(1e-6, np.abs(a[k]).max()), (it, k)
    assert a["delta"][:70].max() <= 12 and a["delta"][70:].max() <= 16 and a["delta"].max() > 8
    assert len(o1.get_class_state(1)["piHat"]) == 16
    with pytest.raises(Exception):
        o0.add_marker_set_r(0, 10, 4.0, v * 0.5, v, [0.1] * 17, [1.0 / 17] * 17)



def test_det_exp_within_one_ulp(O):
    xs = -np.concatenate([10.0 ** np.random.default_rng(0).uniform(-12, 2.8, 20000), [0.0, 1e-30, 0.3465, 0.3466, 0.35, 707.9]])
    for x in xs:
        ref = math.exp(x)
        assert abs(O.det_exp(x) - ref) <= 1.01 * np.spacing(ref)
    assert O.det_exp(0.0) == 1.0 and O.det_exp(-800.0) == 0.0 and math.isnan(O.det_exp(float("nan")))


def test_numpy_restatement_fixed_effect_sets(O):
    """sampleX! (one column) and sampleb! (a block, with the ridge of mme.jl:149-152) in the numpy restatement against both orders
    of the C oracle."""
    from ref_numpy import RefChain
    N, P = 70, 40
    X, y, bt, v = make_problem(O, N, P, seed=2)
    rng = np.random.default_rng(5)
    F1 = rng.normal(size=N); F3 = rng.normal(size=(N, 3)); F3[:, 0] = (rng.uniform(size=N) < 0.4)
    y = y + 1.5 * F1 - 0.7 * F3[:, 2]
    o0 = O.Oracle(0, seed=3, chain=0); o0.set_panel_f32(X)
    o1 = O.Oracle(1, seed=3, chain=0); o1.set_panel_f32(X, R=4, S=18, D=1)
    ref = RefChain(O, X.astype(np.float64), y, seed=3, chain=0)
    for m in (o0, o1):
        m.add_fixed_set(F1, lhs0=[0.2], rhs0=[0.05]); m.add_fixed_set(F3)
        m.add_marker_set(0, P, 0, 4.0, v * 0.5, [(0, P)], [v]); m.set_y(y); m.set_residual_prior(4.0, 0.3 * y.var())
    ref.add_fixed(F1, lhs=[0.2], rhs=[0.05]); ref.add_fixed(F3)
    ref.add_set(0, P, 0, 4.0, v * 0.5, [(0, P)], [v]); ref.E_df, ref.E_scale = 4.0, 0.3 * y.var()
    for it in range(8):
        o0.run(1); o1.run(1); ref.run(1)
        bref = np.concatenate([x["b"] for x in ref.Xfix])
        for o in (o0, o1):
            assert np.abs(o.get_fixed()["b"] - bref).max() <= 1e-10 * max(1.0, np.abs(bref).max()), it
            assert np.abs(o.get_state()["beta"] - ref.state()["beta"]).max() <= 1e-10, it


def test_compact_blocked_arithmetic_matches_the_reference_float64_panel(O):
    """oracle/ngp_oracle.c ora_set_panel_u8 (bytes + analytic centring, the arithmetic the product's compact storage follows)
    against the reference order on the Float64 panel the reference itself would hold for these genotypes (g - mean,
    src/prepMatVec.jl:129): agreement to rounding of the sums, where fp32 tiles are ~1e-7 away."""
    rng = np.random.default_rng(3)
    N, P = 210, 300
    maf = rng.uniform(0.05, 0.5, P)
    G = ((rng.random((N, P)) < maf).astype(np.uint8) + (rng.random((N, P)) < maf).astype(np.uint8))
    Xc = G - G.mean(0)
    bt = np.zeros(P); idx = rng.choice(P, 20, replace=False); bt[idx] = rng.normal(size=20)
    y = 10 + Xc @ bt + rng.normal(size=N)
    v = 0.5 * y.var() / (Xc ** 2).sum(0).mean()
    res = {}
    for name, order, kw in (("ref", 0, {}), ("blk", 1, dict(R=32, S=7, D=4, near=2)), ("blk12", 1, dict(R=16, S=14, D=12, near=3))):
        o = O.Oracle(order=order, seed=7, chain=0)
        o.set_panel_u8(G, **kw)
        o.add_marker_set(0, 150, 0, 4.0, v * 0.5, [(0, 150)], [v])
        o.add_marker_set(150, P - 150, 1, 4.0, v * 0.5, [(j, j + 1) for j in range(P - 150)], [v] * (P - 150), pi0=0.3, estPi=True)
        o.set_y(y); o.set_residual_prior(4.0, 0.25 * y.var()); o.set_schedule(15, 3, 2); o.run(15)
        res[name] = o.get_state()
    for name in ("blk", "blk12"):
        a, b = res["ref"], res[name]
        assert np.array_equal(a["delta"][:P], b["delta"][:P])
        assert np.abs(a["beta"][:P] - b["beta"][:P]).max() < 1e-12 * np.abs(a["beta"]).max()   # rounding of fp64 sums, 15 iterations
        assert abs(a["varE"] - b["varE"]) < 1e-12 * a["varE"]
        assert np.abs(a["ycorr"][:N] - b["ycorr"][:N]).max() < 1e-11
    # padding rows (224 > 210) never leave zero: the residual invariant holds on the real rows
    st = res["blk"]
    assert np.abs(st["ycorr"][:N] - (y - st["b"] - Xc @ st["beta"][:P])).max() < 1e-11
