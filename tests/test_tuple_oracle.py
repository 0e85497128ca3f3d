"""Closed-form checks of the specification oracle for correlated (Tuple) BayesPR marker sets (oracle/ngp_tuple_oracle.c;
reference: src/functions.jl:140-154, 513-516, set-up src/mme.jl:448-489).  CPU only: the path has no device counterpart."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def T():
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    L = C.CDLL(os.path.join(ROOT, "oracle", "libngp_tuple_oracle.so"))
    return L


def _p(a, t=C.c_double):
    return a.ctypes.data_as(C.POINTER(t))


class Chain:
    def __init__(self, L, X, y, regions, v, seed=1, chain=0):
        self.L = L
        X = np.ascontiguousarray(X, dtype=np.float64)            # [set][locus][N]
        self.k, self.P, self.N = X.shape
        rs = np.ascontiguousarray([r[0] for r in regions], dtype=np.int64); re = np.ascontiguousarray([r[1] for r in regions], dtype=np.int64)
        self.nreg = len(rs)
        v = np.ascontiguousarray(np.atleast_2d(v), dtype=np.float64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        self.h = C.c_void_p()
        assert L.tup_create(C.c_int(self.k), C.c_int64(self.N), C.c_int64(self.P), _p(X), _p(y), _p(rs, C.c_int64), _p(re, C.c_int64),
                            C.c_int64(self.nreg), _p(v), C.c_uint64(seed), C.c_uint32(chain), C.byref(self.h)) == 0

    def run(self, n):
        assert self.L.tup_run(self.h, C.c_int64(n)) == 0

    def fix(self, on, varE):
        self.L.tup_fix_variances(self.h, C.c_int(int(on)), C.c_double(varE))

    def prior_e(self, df, scale):
        self.L.tup_set_residual_prior(self.h, C.c_double(df), C.c_double(scale))

    def state(self):
        b = np.empty((self.P, self.k)); yc = np.empty(self.N); vb = np.empty((self.nreg, self.k, self.k)); ve = C.c_double()
        self.L.tup_get_state(self.h, _p(b), _p(yc), _p(vb), C.byref(ve))
        return dict(beta=b, ycorr=yc, varBeta=vb, varE=ve.value)

    def __del__(self):
        self.L.tup_destroy(self.h)


def test_conditional_of_one_locus_is_the_closed_form(T):
    """One locus, two correlated sets, variances held fixed: beta ~ N(invLHS RHS, invLHS) with invLHS = inv(X'X / varE + inv(V))
    (src/functions.jl:146-149); mean and covariance of 20,000 draws against the formula."""
    rng = np.random.default_rng(1)
    N, k = 40, 2
    X = rng.normal(size=(k, 1, N))
    y = 0.8 * X[0, 0] - 0.5 * X[1, 0] + rng.normal(size=N) * 0.7
    V = np.array([[0.5, 0.2], [0.2, 0.4]])
    varE = 0.6
    c = Chain(T, X, y, [(0, 1)], V, seed=3)
    c.fix(True, varE)
    draws = []
    for _ in range(20000):
        c.run(1); draws.append(c.state()["beta"][0].copy())
    D = np.array(draws)
    Xj = X[:, 0, :].T
    invLHS = np.linalg.inv(Xj.T @ Xj / varE + np.linalg.inv(V))
    mean = invLHS @ (Xj.T @ y / varE)
    assert np.abs(D.mean(axis=0) - mean).max() < 4 * np.sqrt(np.diag(invLHS).max() / len(D))
    assert np.abs(np.cov(D.T) - invLHS).max() < 0.05 * np.abs(invLHS).max()
    st = c.state()
    assert np.abs(st["ycorr"] - (y - Xj @ st["beta"][0])).max() < 1e-12       # :145, :150 leave ycorr = y - X beta


def test_inverse_wishart_draw_has_the_right_mean(T):
    """E[InverseWishart(nu, P)] = P / (nu - k - 1) (the prior mean the reference's scale = v (df - k - 1) is built on, mme.jl:501)."""
    k, nu = 3, 12.0
    A = np.random.default_rng(2).normal(size=(k, k)); Psi = A @ A.T + np.eye(k)
    acc = np.zeros((k, k)); out = np.empty((k, k))
    n = 20000
    for i in range(n):
        assert T.tup_inverse_wishart(C.c_uint64(5), C.c_uint64(0), C.c_uint64(i + 1), C.c_uint64(0), C.c_double(nu), _p(np.ascontiguousarray(Psi)),
                                     C.c_int(k), _p(out)) == 0
        acc += out
        assert np.allclose(out, out.T) and np.all(np.linalg.eigvalsh(out) > 0)
    assert np.abs(acc / n - Psi / (nu - k - 1)).max() < 0.05 * np.abs(Psi / (nu - k - 1)).max()


def test_chain_recovers_correlated_effects(T):
    """Two breeds with correlated effects (r = 0.8): posterior means correlate with the truth and the sampled covariance keeps the sign."""
    rng = np.random.default_rng(4)
    N, P, k = 300, 30, 2
    X = rng.normal(size=(k, P, N))
    Sig = np.array([[1.0, 0.8], [0.8, 1.0]]) * 0.05
    B = rng.multivariate_normal(np.zeros(k), Sig, size=P)
    y = sum(X[s].T @ B[:, s] for s in range(k)) + rng.normal(size=N) * 0.5
    c = Chain(T, X, y, [(0, P)], Sig, seed=7)
    c.prior_e(4.0, 0.125)
    c.run(200)
    acc = np.zeros((P, k)); cov = np.zeros((k, k)); m = 400
    for _ in range(m):
        c.run(1); st = c.state(); acc += st["beta"]; cov += st["varBeta"][0]
    post = acc / m
    assert np.corrcoef(post.ravel(), B.ravel())[0, 1] > 0.8
    assert (cov / m)[0, 1] > 0 and abs(st["varE"] - 0.25) < 0.15


def test_k1_is_the_scalar_conditional(T):
    """k = 1 reduces to the scalar conditional of sampleBayesPR!(::Symbol) (src/functions.jl:128-133): same closed form."""
    rng = np.random.default_rng(6)
    N = 50
    X = rng.normal(size=(1, 1, N)); y = 1.1 * X[0, 0] + rng.normal(size=N)
    c = Chain(T, X, y, [(0, 1)], [[0.3]], seed=2)
    c.fix(True, 0.9)
    d = []
    for _ in range(20000):
        c.run(1); d.append(c.state()["beta"][0, 0])
    d = np.array(d)
    lhs = X[0, 0] @ X[0, 0] / 0.9 + 1 / 0.3
    assert abs(d.mean() - (X[0, 0] @ y / 0.9) / lhs) < 4 * np.sqrt(1 / lhs / len(d)) and abs(d.var() - 1 / lhs) < 0.05 / lhs
