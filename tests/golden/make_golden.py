"""Generates the golden fixtures from the reference-order CPU oracle (the reference itself cannot run
here: no Julia toolchain, no vendored dependencies -- SURVEY.md section 8c).  Run from the repo root:
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import add_sets, make_problem  # noqa: E402
from oracle import oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

for name, kind in (("pr_50x200", "PR"), ("b_50x200", "B"), ("c_50x200", "C")):
    N, P = 50, 200
    X, y, bt, v = make_problem(O, N, P, seed=12)
    e_scale = 0.25 * y.var()
    o = O.Oracle(0, seed=2025, chain=0)
    o.set_panel_f32(X)
    add_sets(o, [(0, P, kind)], v)
    o.set_y(y)
    o.set_residual_prior(4.0, e_scale)
    out = dict(X=np.ascontiguousarray(X), y=y, v=v, e_scale=e_scale, seed=2025, chain=0)
    done = 0
    for it in (1, 2, 10):
        o.run(it - done)
        done = it
        s = o.get_state()
        out[f"beta_{it}"] = s["beta"]; out[f"delta_{it}"] = s["delta"]; out[f"varE_{it}"] = s["varE"]
        out[f"varBeta_{it}"] = s["varBeta"]; out[f"b_{it}"] = s["b"]; out[f"piHat_{it}"] = s["piHat"]
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name)
