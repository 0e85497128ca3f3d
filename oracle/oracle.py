"""ctypes wrapper for the CPU oracle (oracle/ngp_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libngp_oracle.so")

PR, B = 0, 1
KIND = dict(VARE_CHI2=1, FIXED_NORMAL=2, BETA_NORMAL=3, REGION_CHI2=4, B_UNIFORM=5, B_LOCUS_CHI2=6, PI_BETA=7)


def build(force=False):
    src = os.path.join(_HERE, "ngp_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.ora_det_log.restype = C.c_double
        L.ora_det_log.argtypes = [C.c_double]
        L.ora_ppnd16.restype = C.c_double
        L.ora_ppnd16.argtypes = [C.c_double]
        L.ora_det_exp.restype = C.c_double
        L.ora_det_exp.argtypes = [C.c_double]
        L.ora_last_error.restype = C.c_char_p
        L.ora_nvb.restype = C.c_int64
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def set_threads(t):
    """Threads of the reference-order BLAS-1 calls (bench.py's all-core baseline only; every test runs with 1)."""
    lib().ora_set_threads(int(t))


def det_log(x):
    return lib().ora_det_log(float(x))


def ppnd16(p):
    return lib().ora_ppnd16(float(p))


def det_exp(x):
    return lib().ora_det_exp(float(x))


def draws(seed, chain, it, kind, index, what, n, p1=0.0, p2=0.0, indexed=False):
    """what: 0 uniform, 1 normal, 2 chisq(p1), 3 beta(p1,p2), 4 gamma(p1)."""
    out = np.empty(n, dtype=np.float64)
    f = lib().ora_draws_indexed if indexed else lib().ora_draws
    f(C.c_uint64(seed), C.c_uint64(chain), C.c_uint64(it), C.c_uint64(kind), C.c_uint64(index), C.c_int(what),
      C.c_double(p1), C.c_double(p2), C.c_int64(n), _p(out, C.c_double))
    return out


def generate_panel(N, P, maf_lo=0.05, maf_hi=0.5, seed=20250509):
    """Synthetic centred fp32 panel, column-major (returned as an (N,P) Fortran-ordered array)."""
    X = np.empty((N, P), dtype=np.float32, order="F")
    mu = np.empty(P, dtype=np.float64)
    lib().ora_generate_panel(C.c_int64(N), C.c_int64(P), C.c_double(maf_lo), C.c_double(maf_hi), C.c_uint64(seed),
                             _p(X, C.c_float), _p(mu, C.c_double))
    return X, mu


def generate_rows(rows, P, maf_lo=0.05, maf_hi=0.5, seed=20250509):
    """Genotype codes (uint8, [len(rows), P]) of chosen rows of the synthetic panel -- without generating the panel."""
    rows = np.ascontiguousarray(rows, dtype=np.int64)
    G = np.empty((len(rows), P), dtype=np.uint8)
    lib().ora_generate_rows(_p(rows, C.c_int64), C.c_int64(len(rows)), C.c_int64(P), C.c_double(maf_lo), C.c_double(maf_hi), C.c_uint64(seed),
                            _p(G, C.c_uint8))
    return G


def column_sums(cols, N, maf_lo=0.05, maf_hi=0.5, seed=20250509):
    """Genotype sums over all N rows of chosen columns of the synthetic panel."""
    cols = np.ascontiguousarray(cols, dtype=np.int64)
    out = np.empty(len(cols), dtype=np.int64)
    lib().ora_column_sums(_p(cols, C.c_int64), C.c_int64(len(cols)), C.c_int64(N), C.c_double(maf_lo), C.c_double(maf_hi), C.c_uint64(seed),
                          _p(out, C.c_int64))
    return out


class Oracle:
    """Handle-style driver with the same call sequence as the product's C ABI."""

    def __init__(self, order=0, seed=1, chain=0):
        self.L = lib()
        self.h = C.c_void_p()
        self._chk(self.L.ora_create(C.c_int(order), C.c_uint64(seed), C.c_uint32(chain), C.byref(self.h)))
        self.order = order
        self.nsets = 0

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError("oracle: " + (self.L.ora_last_error(self.h) or b"?").decode())

    def close(self):
        if self.h:
            self.L.ora_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_panel_f32(self, X, R=0, S=0, D=1, near=3, nchain=8, tform=0):
        """Blocked order: the layout the library reports (ngp_get_layout, ngp_get_config, ngp_get_near_lags, ngp_get_streamer,
        ngp_get_chain_form)."""
        X = np.asfortranarray(X, dtype=np.float32)
        self.N, self.P = X.shape
        self._chk(self.L.ora_set_near(self.h, C.c_int64(near)))
        self._chk(self.L.ora_set_nchain(self.h, C.c_int64(nchain)))
        self.set_tform(tform)
        self._chk(self.L.ora_set_panel_f32(self.h, _p(X, C.c_float), C.c_int64(self.N), C.c_int64(self.P), C.c_int64(R),
                                           C.c_int64(S), C.c_int64(D)))

    def set_tform(self, on=True):
        """Blocked order: linear blocks as dlt = T e0 (the library's default, ngp_get_chain_form) or as the 64-step chain."""
        self._chk(self.L.ora_set_tform(self.h, C.c_int(1 if on else 0)))

    def set_panel_u8(self, G, R=0, S=0, D=1, near=3, centre=True, tform=0):
        """Compact storage (one byte per genotype, analytic centring).  Blocked order: the layout the library reports; reference
        order: the Float64 panel the reference would hold for these genotypes (g - mean, mean = integer column sum / N)."""
        G = np.asfortranarray(G, dtype=np.uint8)
        self.N, self.P = G.shape
        if self.order == 0:
            mu = G.sum(axis=0, dtype=np.int64).astype(np.float64) / float(self.N) if centre else np.zeros(self.P)
            return self.set_panel_f64(G.astype(np.float64) - mu[None, :])
        self._chk(self.L.ora_set_near(self.h, C.c_int64(near)))
        self.set_tform(tform)
        self._chk(self.L.ora_set_panel_u8(self.h, _p(G, C.c_uint8), C.c_int64(self.N), C.c_int64(self.P), C.c_int64(R), C.c_int64(S),
                                          C.c_int64(D), C.c_int(1 if centre else 0)))

    def means(self):
        out = np.zeros(self.P)
        self._chk(self.L.ora_get_means(self.h, _p(out, C.c_double), C.c_int64(self.P)))
        return out

    def set_panel_f64(self, X):
        """Reference order only: the Float64 panel of the reference (already centred in Float64)."""
        X = np.asfortranarray(X, dtype=np.float64)
        self.N, self.P = X.shape
        self._chk(self.L.ora_set_panel_f64(self.h, _p(X, C.c_double), C.c_int64(self.N), C.c_int64(self.P)))

    def add_marker_set(self, col0, ncol, method, df, scale, regions, varBeta0, pi0=0.0, estPi=False, lhs0=None, rhs0=None):
        rs = np.ascontiguousarray([r[0] for r in regions], dtype=np.int64)
        re = np.ascontiguousarray([r[1] for r in regions], dtype=np.int64)
        vb = np.ascontiguousarray(varBeta0, dtype=np.float64)
        assert len(vb) == len(rs)
        l0 = None if lhs0 is None else np.ascontiguousarray(lhs0, dtype=np.float64)
        r0 = None if rhs0 is None else np.ascontiguousarray(rhs0, dtype=np.float64)
        sid = C.c_int()
        self._chk(self.L.ora_add_marker_set(self.h, C.c_int64(col0), C.c_int64(ncol), C.c_int(method), C.c_double(df),
                                            C.c_double(scale), _p(rs, C.c_int64), _p(re, C.c_int64), C.c_int64(len(rs)),
                                            _p(vb, C.c_double), C.c_double(pi0), C.c_int(int(estPi)), _p(l0, C.c_double),
                                            _p(r0, C.c_double), C.byref(sid)))
        self.nsets += 1
        return sid.value

    def add_fixed_set(self, X, lhs0=None, rhs0=None):
        """Columns of one fixed-effect term / block (functions.jl:22-53), sampled after the intercept in the order added."""
        X = np.asfortranarray(X, dtype=np.float64)
        if X.ndim == 1:
            X = X[:, None]
        l0 = None if lhs0 is None else np.ascontiguousarray(lhs0, dtype=np.float64)
        r0 = None if rhs0 is None else np.ascontiguousarray(rhs0, dtype=np.float64)
        sid = C.c_int()
        self._chk(self.L.ora_add_fixed_set(self.h, _p(X, C.c_double), C.c_int64(X.shape[0]), C.c_int64(X.shape[1]), _p(l0, C.c_double),
                                           _p(r0, C.c_double), C.byref(sid)))
        return sid.value

    def get_fixed(self):
        b = np.empty(1024); sb = np.empty(1024); n = C.c_int64()
        self._chk(self.L.ora_get_fixed(self.h, _p(b, C.c_double), _p(sb, C.c_double), C.byref(n)))
        return dict(b=b[:n.value].copy(), sum_b=sb[:n.value].copy())

    def add_marker_set_r(self, col0, ncol, df, scale, varBeta0, vClass, pi, estPi=False, lhs0=None, rhs0=None):
        """BayesR set (mme.jl:374-383): class multipliers vClass and class probabilities pi."""
        vc = np.ascontiguousarray(vClass, dtype=np.float64); pp = np.ascontiguousarray(pi, dtype=np.float64)
        assert len(vc) == len(pp)
        l0 = None if lhs0 is None else np.ascontiguousarray(lhs0, dtype=np.float64)
        r0 = None if rhs0 is None else np.ascontiguousarray(rhs0, dtype=np.float64)
        sid = C.c_int()
        self._chk(self.L.ora_add_marker_set_r(self.h, C.c_int64(col0), C.c_int64(ncol), C.c_double(df), C.c_double(scale), C.c_double(varBeta0),
                                              _p(vc, C.c_double), _p(pp, C.c_double), C.c_int(len(vc)), C.c_int(int(estPi)), _p(l0, C.c_double),
                                              _p(r0, C.c_double), C.byref(sid)))
        self.nsets += 1
        return sid.value

    def add_marker_set_tuple(self, col0, nloc, k, df, scale, regions, varBeta0):
        """Correlated sets (Tuple BayesPR, mme.jl:448-489): k sets, nloc loci, locus-major columns from col0 (a block boundary) on;
        scale and varBeta0 are k x k, regions ranges of loci."""
        rs = np.ascontiguousarray([r[0] for r in regions], dtype=np.int64)
        re = np.ascontiguousarray([r[1] for r in regions], dtype=np.int64)
        sc = np.ascontiguousarray(np.asarray(scale, dtype=np.float64).reshape(k, k)); vb = np.ascontiguousarray(np.asarray(varBeta0, dtype=np.float64).reshape(k, k))
        sid = C.c_int()
        self._chk(self.L.ora_add_marker_set_tuple(self.h, C.c_int64(col0), C.c_int64(nloc), C.c_int(k), C.c_double(df), _p(sc, C.c_double),
                                                  _p(rs, C.c_int64), _p(re, C.c_int64), C.c_int64(len(rs)), _p(vb, C.c_double), C.byref(sid)))
        self.nsets += 1
        return sid.value

    def get_class_state(self, set_id):
        pi = np.empty(16); sp = np.empty(16); K = C.c_int64()
        self._chk(self.L.ora_get_class_state(self.h, C.c_int(set_id), _p(pi, C.c_double), _p(sp, C.c_double), C.byref(K)))
        return dict(piHat=pi[:K.value].copy(), sum_pi=sp[:K.value].copy())

    def set_y(self, y):
        y = np.ascontiguousarray(y, dtype=np.float64)
        self._chk(self.L.ora_set_y(self.h, _p(y, C.c_double), C.c_int64(len(y))))

    def set_residual_prior(self, df, scale):
        self._chk(self.L.ora_set_residual_prior(self.h, C.c_double(df), C.c_double(scale)))

    def set_intercept(self, on):
        self._chk(self.L.ora_set_intercept(self.h, C.c_int(int(on))))

    def set_schedule(self, chainLength, burnIn, thin):
        self._chk(self.L.ora_set_schedule(self.h, C.c_int64(chainLength), C.c_int64(burnIn), C.c_int64(thin)))

    def run(self, niter):
        self._chk(self.L.ora_run(self.h, C.c_int64(niter)))

    def run_pr_threaded(self, niter, threads):
        """bench.py's all-core baseline: one parallel region per sweep, BayesPR sets, reference order (timing only)."""
        self._chk(self.L.ora_run_pr_threaded(self.h, C.c_int64(niter), C.c_int(int(threads))))

    def get_state(self):
        nvb = self.L.ora_nvb(self.h)
        yc = np.empty(self.N); beta = np.empty(self.P); delta = np.empty(self.P, dtype=np.int64)
        vb = np.empty(nvb); pi = np.empty(2 * max(self.nsets, 1))
        varE = C.c_double(); b = C.c_double(); it = C.c_int64()
        self._chk(self.L.ora_get_state(self.h, _p(yc, C.c_double), _p(beta, C.c_double), _p(delta, C.c_int64),
                                       _p(vb, C.c_double), _p(pi, C.c_double), C.byref(varE), C.byref(b), C.byref(it)))
        return dict(ycorr=yc, beta=beta, delta=delta, varBeta=vb, piHat=pi[:2 * self.nsets], varE=varE.value, b=b.value,
                    iter=it.value)

    def get_trace(self, n):
        v = np.empty(n); b = np.empty(n)
        self._chk(self.L.ora_get_trace(self.h, _p(v, C.c_double), _p(b, C.c_double), C.c_int64(n)))
        return dict(varE=v, b=b)

    def get_posterior_sums(self):
        nvb = self.L.ora_nvb(self.h)
        sb = np.empty(self.P); sb2 = np.empty(self.P); sd = np.empty(self.P); sv = np.empty(nvb)
        sp = np.empty(2 * max(self.nsets, 1)); se = C.c_double(); sbb = C.c_double(); nk = C.c_int64()
        self._chk(self.L.ora_get_posterior_sums(self.h, _p(sb, C.c_double), _p(sb2, C.c_double), _p(sd, C.c_double),
                                                _p(sv, C.c_double), _p(sp, C.c_double), C.byref(se), C.byref(sbb),
                                                C.byref(nk)))
        return dict(sum_beta=sb, sum_beta2=sb2, sum_delta=sd, sum_varBeta=sv, sum_pi=sp[:2 * self.nsets], sum_varE=se.value,
                    sum_b=sbb.value, nKept=nk.value)

    def get_gram(self, t):
        out = np.empty((64, 64))
        self._chk(self.L.ora_get_gram(self.h, C.c_int64(t), _p(out, C.c_double)))
        return out
