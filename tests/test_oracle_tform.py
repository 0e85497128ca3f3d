"""CPU: the inverse form of the BayesPR block chain (oracle tform = 1, what ngp_set_chain_form(1) runs on the device) against the
64-step form and against the reference order -- the same chain to rounding, also on a panel with strong LD inside the blocks."""
import numpy as np
from conftest import add_sets, make_problem


def _ld_panel(N, P, rho, seed):
    """columns with AR(1)-like correlation rho between neighbours (what linkage disequilibrium does inside a 64-SNP block)"""
    rng = np.random.default_rng(seed)
    Z = rng.normal(size=(N, P))
    X = np.empty_like(Z)
    X[:, 0] = Z[:, 0]
    for j in range(1, P):
        X[:, j] = rho * X[:, j - 1] + np.sqrt(1 - rho * rho) * Z[:, j]
    X -= X.mean(axis=0)
    return np.asfortranarray(X.astype(np.float32))


def test_inverse_form_equals_the_step_chain_and_the_reference_order(O):
    for rho, tol in ((0.0, 1e-10), (0.9, 1e-9), (0.99, 1e-8)):
        N, P = 400, 200
        X = _ld_panel(N, P, rho, seed=3)
        rng = np.random.default_rng(4)
        bt = np.zeros(P); bt[rng.choice(P, 10, replace=False)] = rng.normal(size=10)
        y = 5.0 + X.astype(np.float64) @ bt + rng.normal(size=N)
        v = 0.5 * y.var() / (X.astype(np.float64) ** 2).sum(axis=0).mean() * N / N
        res = {}
        for key, order, tform in (("ref", 0, 0), ("steps", 1, 0), ("inv", 1, 1)):
            o = O.Oracle(order=order, seed=17, chain=1)
            if order:
                o.set_panel_f32(X, R=40, S=10, D=4, near=2, nchain=8, tform=tform)
            else:
                o.set_panel_f32(X)
            add_sets(o, [(0, 120, "PR"), (120, 80, "PR")], v)
            o.set_y(y); o.set_residual_prior(4.0, 0.25 * y.var()); o.run(30)
            res[key] = o.get_state()
        scale = np.abs(res["ref"]["beta"]).max()
        for key in ("steps", "inv"):
            assert np.abs(res[key]["beta"] - res["ref"]["beta"]).max() <= tol * scale, (rho, key)
            assert abs(res[key]["varE"] / res["ref"]["varE"] - 1) <= tol
        assert not np.array_equal(res["inv"]["beta"], res["steps"]["beta"])   # another summation order, not the same bits
