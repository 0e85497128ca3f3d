"""The correlated (Tuple) BayesPR path inside the MAIN oracle (oracle/ngp_oracle.c: reference order = src/functions.jl:140-154 line by
line; blocked order = what the device computes), on the CPU: the two orders agree to rounding, k = 1 IS the Symbol path, and an
independent numpy restatement written from the Julia source draws the same chain.  (Closed forms of the conditional and of the
inverse-Wishart draw: tests/test_tuple_oracle.py, on the standalone specification oracle.)"""
import numpy as np
import pytest

from conftest import make_problem


def tuple_problem(O, ngp, N, nloc, k, seed=5, extra=0):
    """k correlated marker sets of nloc loci (+ `extra` plain columns behind them), interleaved as the device wants them."""
    P = nloc * k + extra
    X, y, bt, v = make_problem(O, N, P, seed=seed, ncausal=12)
    sets = [np.asfortranarray(X[:, m * nloc:(m + 1) * nloc]) for m in range(k)]
    Xt = ngp.tuple_panel(sets)
    span = Xt.shape[1]
    pad = (-span) % 64 if extra else 0
    Xp = np.asfortranarray(np.hstack([Xt, np.zeros((N, pad), dtype=X.dtype), X[:, nloc * k:]])) if extra else Xt
    vm = v * (0.6 * np.eye(k) + 0.4 * np.ones((k, k)))       # prior v: k x k, positive definite
    return Xp, y, vm, v, span, span + pad


def add_tuple(m, nloc, k, vm, regions):
    df = 3.0 + k                                              # mme.jl:493
    scale = vm * (df - k - 1.0) if k > 1 else vm * (df - 2.0) / df   # mme.jl:501
    m.add_marker_set_tuple(0, nloc, k, df, scale, regions, vm)


@pytest.mark.parametrize("tform", [0, 1], ids=["steps", "inverse_form"])
@pytest.mark.parametrize("k,nloc", [(2, 75), (3, 50), (4, 40)])
def test_blocked_order_equals_reference_order(O, ngp, k, nloc, tform):
    N = 150
    Xp, y, vm, v, span, off = tuple_problem(O, ngp, N, nloc, k, extra=40)
    regions = [(0, nloc // 3), (nloc // 3, nloc)]
    res = []
    for order in (0, 1):
        o = O.Oracle(order=order, seed=21, chain=0)
        o.set_panel_f32(Xp, R=40, S=4, D=3, near=2, nchain=8, tform=tform) if order else o.set_panel_f32(Xp)
        add_tuple(o, nloc, k, vm, regions)
        o.add_marker_set(off, 40, 1, 4.0, v * 0.5, [(j, j + 1) for j in range(40)], [v] * 40, pi0=0.2, estPi=True)
        o.set_y(y); o.set_residual_prior(4.0, 0.5); o.run(12)
        res.append(o.get_state())
    a, b = res
    scale = np.abs(a["beta"]).max()
    assert np.abs(a["beta"] - b["beta"]).max() < 1e-9 * scale and np.array_equal(a["delta"], b["delta"])
    assert np.allclose(a["varBeta"], b["varBeta"], rtol=1e-8, atol=0) and abs(a["varE"] / b["varE"] - 1) < 1e-10
    assert np.abs(a["ycorr"] - b["ycorr"][:N]).max() < 1e-8 * np.abs(y).max()
    vb = a["varBeta"][:2 * k * k].reshape(2, k, k)
    for r in range(2):                                        # variance matrices stay symmetric positive definite
        assert np.allclose(vb[r], vb[r].T, rtol=1e-9) and (np.linalg.eigvalsh((vb[r] + vb[r].T) / 2) > 0).all()


@pytest.mark.parametrize("order", [0, 1, 2], ids=["reference", "blocked_steps", "blocked_inverse_form"])
def test_one_set_tuple_is_the_symbol_path(O, ngp, order):
    """k = 1: sampleBayesPR!(::Tuple) with a 1 x 1 variance is sampleBayesPR!(::Symbol) (the inverse Wishart in one dimension is the
    scaled inverse chi-square) -- bit for bit in the blocked order, to rounding in the reference order (the Tuple method divides by
    varE where the Symbol method multiplies by 1 / varE)."""
    N, nloc = 120, 100
    X, y, bt, v = make_problem(O, N, 128 + 30, seed=8)          # a tuple set owns its blocks to the end: the next set starts at column 128
    regions = [(0, 37), (37, 100)]
    res = []
    for tup in (True, False):
        o = O.Oracle(order=min(order, 1), seed=9, chain=2)
        o.set_panel_f32(X, R=32, S=4, D=4, near=3, nchain=8, tform=order - 1) if order else o.set_panel_f32(X)
        s = v * 0.5                                           # the Symbol path's scale = v (df - 2) / df, df = 4
        if tup:
            o.add_marker_set_tuple(0, nloc, 1, 4.0, [[s * 4.0]], regions, [[v]])   # InverseWishart(df + n, scale + b'b): scale = s df
        else:
            o.add_marker_set(0, nloc, 0, 4.0, s, regions, [v, v])
        o.add_marker_set(128, 30, 0, 4.0, s, [(0, 30)], [v])
        o.set_y(y); o.set_residual_prior(4.0, 0.5); o.run(15)
        res.append(o.get_state())
    a, b = res
    if order >= 1:
        for key in ("ycorr", "beta", "delta", "varBeta"):
            assert np.array_equal(a[key], b[key]), key
        assert a["varE"] == b["varE"] and a["b"] == b["b"]
    else:
        assert np.abs(a["beta"] - b["beta"]).max() < 1e-11 * np.abs(a["beta"]).max() and np.allclose(a["varBeta"], b["varBeta"], rtol=1e-11)


@pytest.mark.parametrize("k", [2, 3])
def test_numpy_restatement_draws_the_same_chain(O, ngp, k):
    """tests/ref_numpy.py restates src/functions.jl:140-154 with the reference's own data shapes (X_l as N x k matrices, numpy's inv and
    cholesky); fed the same keyed draws it agrees with the C restatement to rounding."""
    from ref_numpy import RefChain
    N, nloc = 90, 30
    Xp, y, vm, v, span, off = tuple_problem(O, ngp, N, nloc, k, extra=20)
    regions = [(0, 12), (12, 30)]
    o = O.Oracle(order=0, seed=4, chain=1)
    o.set_panel_f32(Xp)
    add_tuple(o, nloc, k, vm, regions)
    o.add_marker_set(off, 20, 0, 4.0, v * 0.5, [(0, 20)], [v])
    o.set_y(y); o.set_residual_prior(4.0, 0.5); o.run(6)
    rc = RefChain(O, Xp, y, seed=4, chain=1)
    rc.E_df, rc.E_scale = 4.0, 0.5
    df = 3.0 + k
    rc.add_set_tuple(ngp.tuple_columns(0, nloc, k), df, vm * (df - k - 1.0), regions, vm)
    rc.M[0]["span"] = off                                     # the plain set starts on the next block boundary
    rc.delta[0] = np.ones(off, dtype=np.int64)
    rc.add_set(off, 20, 0, 4.0, v * 0.5, regions=[(0, 20)], varBeta0=[v])
    rc.run(6)
    a, b = o.get_state(), rc.state()
    assert np.abs(a["beta"] - b["beta"]).max() < 1e-9 * np.abs(a["beta"]).max()
    assert np.allclose(a["varBeta"], b["varBeta"], rtol=1e-8) and abs(a["varE"] / b["varE"] - 1) < 1e-10
