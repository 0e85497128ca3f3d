import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def O():
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def ngp():
    try:  # torch's HIP initialisation first (tests that hand torch device buffers to the library found it failing when it came second)
        import torch
        torch.cuda.is_available()
    except Exception:
        pass
    from ngp_pkg import load_pkg
    return load_pkg()


def make_problem(O, N, P, seed=1, ncausal=10, h2=0.5, panel_seed=20250509):
    """Synthetic panel + phenotype as BASELINE.md section 4 describes (small sizes)."""
    X, mu = O.generate_panel(N, P, seed=panel_seed)
    rng = np.random.default_rng(seed)
    bt = np.zeros(P)
    idx = rng.choice(P, min(ncausal, P), replace=False)
    bt[idx] = rng.normal(size=len(idx))
    g = X.astype(np.float64) @ bt
    vg = g.var() if g.var() > 0 else 1.0
    e = rng.normal(size=N) * np.sqrt(vg * (1 - h2) / h2)
    y = 10.0 + g + e
    twopq = float((2 * mu / 2 * (1 - mu / 2)).sum())
    v = 0.5 * y.var() / max(twopq, 1e-9)
    return X, y, bt, v


def add_sets(m, spec, v):
    """spec: list of (col0, ncol, 'PR'|'B'|'Bfix'|'C'|'Cfix'|'PR1'|('PRw', width)|'R'|'Rfix'|'R2'|'R6'|'R8'); same calls on oracle and product."""
    df = 4.0
    for col0, ncol, kind in spec:
        if kind == "PR":
            m.add_marker_set(col0, ncol, 0, df, v * (df - 2) / df, [(0, ncol)], [v])
        elif kind == "PR1":
            m.add_marker_set(col0, ncol, 0, df, v * (df - 2) / df, [(j, j + 1) for j in range(ncol)], [v] * ncol)
        elif isinstance(kind, tuple) and kind[0] == "PRw":
            w = kind[1]
            regs = [(a, min(a + w, ncol)) for a in range(0, ncol, w)]
            m.add_marker_set(col0, ncol, 0, df, v * (df - 2) / df, regs, [v] * len(regs))
        elif kind == "B":
            m.add_marker_set(col0, ncol, 1, df, v * (df - 2) / df, [(j, j + 1) for j in range(ncol)], [v] * ncol, pi0=0.05,
                             estPi=True)
        elif kind == "Bfix":
            m.add_marker_set(col0, ncol, 1, df, v * (df - 2) / df, [(j, j + 1) for j in range(ncol)], [v] * ncol, pi0=0.2,
                             estPi=False)
        elif kind == "C":
            m.add_marker_set(col0, ncol, 2, df, v * (df - 2) / df, [(0, ncol)], [v], pi0=0.05, estPi=True)
        elif kind == "Cfix":
            m.add_marker_set(col0, ncol, 2, df, v * (df - 2) / df, [(0, ncol)], [v], pi0=0.3, estPi=False)
        elif kind == "R":      # BayesR: zero class + three non-zero classes, pi estimated (src/mme.jl:374-383)
            m.add_marker_set_r(col0, ncol, df, v * (df - 2) / df, v, [0.0, 0.01, 0.1, 1.0], [0.85, 0.10, 0.04, 0.01], estPi=True)
        elif kind == "Rfix":   # three classes, none of them zero, pi fixed
            m.add_marker_set_r(col0, ncol, df, v * (df - 2) / df, v, [0.001, 0.1, 1.0], [0.6, 0.3, 0.1], estPi=False)
        elif kind == "R6":     # six variance classes (more than the four the block chain keeps in registers), pi estimated
            m.add_marker_set_r(col0, ncol, df, v * (df - 2) / df, v, [0.0, 0.0001, 0.001, 0.01, 0.1, 1.0], [0.5, 0.2, 0.12, 0.1, 0.05, 0.03], estPi=True)
        elif kind == "R8":     # eight classes, no zero class, pi fixed
            m.add_marker_set_r(col0, ncol, df, v * (df - 2) / df, v, [1e-5, 1e-4, 1e-3, 0.01, 0.05, 0.2, 0.5, 1.0], [0.3, 0.2, 0.15, 0.1, 0.1, 0.06, 0.05, 0.04], estPi=False)
        elif kind == "R12":    # twelve classes (a zero class first): classes 9..12 come from memory, not from the sampler's LDS copy
            m.add_marker_set_r(col0, ncol, df, v * (df - 2) / df, v, [0.0, 1e-5, 3e-5, 1e-4, 3e-4, 1e-3, 3e-3, 0.01, 0.03, 0.1, 0.3, 1.0],
                               [0.4, 0.1, 0.08, 0.08, 0.07, 0.06, 0.05, 0.05, 0.04, 0.03, 0.02, 0.02], estPi=True)
        elif kind == "R16":    # sixteen classes, no zero class, pi fixed
            m.add_marker_set_r(col0, ncol, df, v * (df - 2) / df, v, [2.0 ** (i - 15) for i in range(16)], [1.0 / 16] * 16, estPi=False)
        elif kind == "R2":     # two classes: the zero class first is not required by the reference
            m.add_marker_set_r(col0, ncol, df, v * (df - 2) / df, v, [1.0, 0.0], [0.3, 0.7], estPi=True)
        else:
            raise ValueError(kind)
