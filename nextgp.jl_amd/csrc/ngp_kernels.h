// ngp_kernels.h -- gfx950 kernels of the blocked marker-effect Gibbs sweep ("stage A": one
// streaming kernel + one recursion kernel per 64-SNP block).  DESIGN.md "Blocked sweep
// arithmetic" is the normative description of every summation order used here; the CPU
// oracle's order-1 path reproduces it operation by operation.
//
// Reference arithmetic being replaced (under /root/reference):
//   src/functions.jl:118-137  sampleBayesPR!   (add-back daxpy, ddot, draw, update daxpy per SNP)
//   src/functions.jl:157-195  sampleBayesB!
//   src/functions.jl:41-47    intercept        src/functions.jl:523-525 sampleVarE
//   src/functions.jl:509-511  sampleVarBetaPR  src/functions.jl:531-533 samplePi
#pragma once
#include "ngp_common.h"

#pragma clang fp contract(off)

namespace ngp {


// ------------------------------------------------------------------------------------------
// iteration head: varE draw from ycorr'ycorr (functions.jl:523-525), intercept draw and
// ycorr -= db (functions.jl:41-47).  ONE workgroup of 1024 threads.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_head(double *__restrict__ ycorr, long long L, long long N, DScal *__restrict__ sc,
                                               double e_df, double e_scale, int intercept, int draw_varE, uint64_t seed,
                                               uint64_t chain, uint64_t it, double *__restrict__ tr_varE,
                                               double *__restrict__ tr_b, long long trace_idx, const unsigned *__restrict__ abort_w,
                                               double mpm_max) {
    if (abort_w && *abort_w != 0u) return;  // an earlier sweep of this call gave up (ngp_sweep_args.h, abort_w)
    __shared__ double wyy[16], wsy[16];
    __shared__ double s_db;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    double ayy = 0.0, asy = 0.0;
    for (long long i = tid; i < L; i += 1024) {
        double v = ycorr[i];
        ayy = __builtin_fma(v, v, ayy);
        asy = asy + v;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        ayy = ayy + __shfl_xor(ayy, off);
        asy = asy + __shfl_xor(asy, off);
    }
    if (lane == 0) {
        wyy[wv] = ayy;
        wsy[wv] = asy;
    }
    __syncthreads();
    if (tid == 0) {
        double yy = wyy[0], sy = wsy[0];
        for (int k = 1; k < 16; k++) {
            yy = yy + wyy[k];
            sy = sy + wsy[k];
        }
        {   // fixed-point scale of this iteration's X_t'ycorr accumulators (ngp_common.h): from ycorr'ycorr as it stands here
            const int fe = fx_exponent(mpm_max * yy);
            sc->fx_scale = fx_pow2(52 - fe);
            sc->fx_inv = fx_pow2(fe - 52);
        }
        double varE = sc->varE, iVarE = sc->iVarE;
        if (draw_varE) {
            Rng r = rng_seed(seed, chain, it, NGP_KIND_VARE_CHI2, 0);
            double chi = rng_chisq(r, e_df + (double)N);
            double t = e_df * e_scale;
            t = t + yy;
            varE = t / chi;
            iVarE = 1.0 / varE;
            sc->varE = varE;
            sc->iVarE = iVarE;
        }
        double db = 0.0;
        if (intercept) {
            double Nd = (double)N;
            double bo = sc->b;
            double tb = Nd * bo;
            double sb = sy + tb;
            double rhs = sb * iVarE;
            double lhs = Nd * iVarE;
            double mean = rhs / lhs;
            double sd = det_sqrt(1.0 / lhs);
            Rng r = rng_seed(seed, chain, it, NGP_KIND_FIXED_NORMAL, 0);
            double z = rng_normal(r);
            double tz = sd * z;
            double bn = mean + tz;
            db = bn - bo;
            sc->b = bn;
        }
        sc->db = db;
        s_db = db;
        if (tr_varE) {
            tr_varE[trace_idx] = varE;
            tr_b[trace_idx] = sc->b;
        }
    }
    __syncthreads();
    if (intercept) {
        double db = s_db;
        for (long long i = tid; i < N; i += 1024) ycorr[i] = ycorr[i] - db;
    }
}

// ------------------------------------------------------------------------------------------
// one fixed-effect set beyond the intercept (src/functions.jl:22-53: sampleX! for one column, sampleb! for a block; set-up
// src/mme.jl:120-152).  ONE workgroup of 1024 threads, one formulation for every width:
//   Yi_a = (x_a'ycorr + sum_c X'X[a][c] b_c) iVarE   -- X'(ycorr + X b) without touching ycorr (the 1024-lane dot of k_head)
//   Gauss-Seidel over the ridged X'X (thread 0; a single column adds M.rhs / M.lhs instead), normal draws keyed
//   (NGP_KIND_FIXED_NORMAL, (set + 1) << 20 | column)
//   ycorr_i -= sum_a x_ia (b_a' - b_a)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_fixed(double *__restrict__ ycorr, long long N, const double *__restrict__ X, int nc,
                                                const double *__restrict__ xpx0, const double *__restrict__ xpxR,
                                                const double *__restrict__ lhs0, const double *__restrict__ rhs0, double *__restrict__ b,
                                                const DScal *__restrict__ sc, int fset, uint64_t seed, uint64_t chain, uint64_t it, const unsigned *__restrict__ abort_w) {
    if (abort_w && *abort_w != 0u) return;  // an earlier sweep of this call gave up (ngp_sweep_args.h, abort_w)
    __shared__ double wsum[16];
    __shared__ double Yi[64], db[64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const double iVarE = sc->iVarE;
    for (int a = 0; a < nc; a++) {
        const double *xa = X + (size_t)a * N;
        double acc = 0.0;
        for (long long i = tid; i < N; i += 1024) acc = __builtin_fma(xa[i], ycorr[i], acc);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) acc = acc + __shfl_xor(acc, off);
        if (lane == 0) wsum[wv] = acc;
        __syncthreads();
        if (tid == 0) {
            double d = wsum[0];
            for (int k = 1; k < 16; k++) d = d + wsum[k];
            double xb = 0.0;
            for (int c = 0; c < nc; c++) xb = __builtin_fma(xpx0[a * nc + c], b[c], xb);
            const double tot = d + xb;
            Yi[a] = tot * iVarE;
        }
        __syncthreads();
    }
    if (tid == 0) {
        double bVec[64];
        for (int a = 0; a < nc; a++) bVec[a] = b[a];
        for (int a = 0; a < nc; a++) {
            bVec[a] = 0.0;
            double d = 0.0;
            for (int c = 0; c < nc; c++) d = __builtin_fma(xpxR[a * nc + c], bVec[c], d);
            const double t1 = d * iVarE;
            const double rhsb = (nc == 1) ? (Yi[a] + rhs0[0]) : (Yi[a] - t1);
            double lhsb = xpxR[a * nc + a] * iVarE;
            if (nc == 1) lhsb = lhsb + lhs0[0];
            const double inv = 1.0 / lhsb;
            const double meanb = inv * rhsb;
            Rng r = rng_seed(seed, chain, it, NGP_KIND_FIXED_NORMAL, ((uint64_t)(fset + 1) << 20) | (uint64_t)a);
            const double z = rng_normal(r);
            const double sd = det_sqrt(inv);
            const double tz = sd * z;
            bVec[a] = meanb + tz;
        }
        for (int a = 0; a < nc; a++) {
            db[a] = bVec[a] - b[a];
            b[a] = bVec[a];
        }
    }
    __syncthreads();
    for (long long i = tid; i < N; i += 1024) {
        double t = 0.0;
        for (int a = 0; a < nc; a++) t = __builtin_fma(X[(size_t)a * N + i], db[a], t);
        ycorr[i] = ycorr[i] - t;
    }
}
__global__ void k_accum_fixed(long long n, const double *__restrict__ b, double *__restrict__ sum_b, const unsigned *__restrict__ abort_w) {
    if (abort_w && *abort_w != 0u) return;  // an earlier sweep of this call gave up (ngp_sweep_args.h, abort_w)
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) sum_b[k] += b[k];
}

// ------------------------------------------------------------------------------------------
// per-locus coefficients of the block recursion (everything the serial chain does NOT need to
// compute): c, w, q, T and the pre-drawn chi-square of BayesB.  active_set < 0: all sets.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_prep(long long Ppad, const int8_t *__restrict__ setof, const int32_t *__restrict__ loc,
                                              const int32_t *__restrict__ vbidx, const DSet *__restrict__ sets,
                                              const DScal *__restrict__ sc, const double *__restrict__ varBeta,
                                              const double *__restrict__ mpm, const double *__restrict__ lhs0,
                                              const double *__restrict__ rhs0, const double *__restrict__ beta,
                                              double *__restrict__ c, double *__restrict__ w, double *__restrict__ q,
                                              double *__restrict__ T, double *__restrict__ chi, int active_set, uint64_t seed,
                                              uint64_t chain, uint64_t it, long long nreg, const DReg *__restrict__ regs,
                                              double *__restrict__ regchi, double *__restrict__ rcls, unsigned *__restrict__ ccnt,
                                              long long ccnt_words, const unsigned *__restrict__ abort_w, const DTup *__restrict__ tup,
                                              double *__restrict__ tupc, const double *__restrict__ tupg) {
    if (abort_w && *abort_w != 0u) return;  // an earlier sweep of this call gave up (ngp_sweep_args.h, abort_w)
    // the hand-off counters of the persistent sweep that follows in the stream start from zero (was a memset launch of its own)
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < ccnt_words; i += (long long)gridDim.x * 256) ccnt[i] = 0u;
    long long k = (long long)blockIdx.x * 256 + threadIdx.x;
    if (k < nreg) {  // data-independent draws of the region variances (functions.jl:509-511): off the post-sweep path
        const DReg Rg = regs[k];
        if (sets[Rg.set].method == 0) {  // BayesC's degrees of freedom depend on the sweep: drawn in k_regdraw
            Rng rr = rng_seed(seed, chain, it, NGP_KIND_REGION_CHI2, ((uint64_t)Rg.set << 40) | (uint64_t)Rg.rg);
            regchi[k] = rng_chisq(rr, sets[Rg.set].df + (double)Rg.n);
        }
    }
    if (k >= Ppad) return;
    int si = setof[k];
    if (si < 0 || (active_set >= 0 && si != active_set)) {
        c[k] = 0.0;
        w[k] = 0.0;
        q[k] = -1.0;
        T[k] = 1.0;
        chi[k] = 1.0;
        return;
    }
    const DSet S = sets[si];
    const double varE = sc->varE, iVarE = sc->iVarE;
    const uint64_t l = (uint64_t)loc[k];
    const uint64_t key = ((uint64_t)si << 40) | l;
    if (S.method == NGP_METHOD_TUPLE_DEV) {
        // Tuple set (src/functions.jl:143-149): the k x k conditional of this column's locus -- every column of the locus forms it
        // (k <= 4: a few dozen flops) and keeps its own row: C[m][.] = iVarE inv(LHS)[m][.], W_m = (L z)_m - beta_m, L = chol(inv(LHS)),
        // LHS = X_l'X_l iVarE + inv(varBeta_r).  loc = locus * k + component (also the key of the component's normal draw).
        const DTup Tp = tup[si];
        const int kk = Tp.k, m = (int)(l % (uint64_t)kk);
        const long long base = k - m;
        double invB[NGP_KMAX * NGP_KMAX], LHS[NGP_KMAX * NGP_KMAX], invLHS[NGP_KMAX * NGP_KMAX], Lc[NGP_KMAX * NGP_KMAX];
        double vbm[NGP_KMAX * NGP_KMAX];
        for (int a = 0; a < kk * kk; a++) vbm[a] = varBeta[vbidx[k] + a];
        int bad = t_spd_inv(vbm, kk, invB);
        for (int a = 0; a < kk; a++)
            for (int b = 0; b < kk; b++) {
                const double t1 = tupg[(size_t)b * Ppad + base + a] * iVarE;
                LHS[a * kk + b] = t1 + invB[a * kk + b];
            }
        bad |= t_spd_inv(LHS, kk, invLHS);
        bad |= t_chol1(invLHS, kk, Lc);
        for (int b = 0; b < kk; b++) tupc[(size_t)b * Ppad + k] = iVarE * invLHS[m * kk + b];
        double acc = 0.0;
        for (int b = 0; b <= m; b++) {
            Rng rz = rng_seed(seed, chain, it, NGP_KIND_BETA_NORMAL, ((uint64_t)si << 40) | (l - (uint64_t)m + (uint64_t)b));
            const double zb = rng_normal(rz);
            acc = (b == 0) ? Lc[m * kk] * zb : __builtin_fma(Lc[m * kk + b], zb, acc);
        }
        c[k] = 0.0;
        w[k] = bad ? __builtin_nan("") : acc - beta[k];  // a variance matrix that is not positive definite poisons the chain visibly
        q[k] = -1.0;
        T[k] = 1.0;
        chi[k] = 1.0;
        return;
    }
    if (S.method == 3) {  // BayesR: per-class coefficients (src/functions.jl:254-255); the class is chosen inside the block chain
        double *rq = rcls, *ra = rcls + (size_t)NGP_RMAX * Ppad, *rt = rcls + (size_t)2 * NGP_RMAX * Ppad, *ru = rcls + (size_t)3 * NGP_RMAX * Ppad;
        const double varB = varBeta[vbidx[k]];
        const double t1r = mpm[k] * iVarE;
        const double t2r = t1r + lhs0[k];
        Rng rz = rng_seed(seed, chain, it, NGP_KIND_BETA_NORMAL, key);
        const double zr = rng_normal(rz);
        double qv[NGP_RMAX], av[NGP_RMAX], u0 = 1.0;  // (kept for the lazy threshold below)
        for (int v = 0; v < S.K; v++) {
            const double varc = varB * S.vcls[v];
            Rng ruu = rng_seed(seed, chain, it, NGP_KIND_R_UNIFORM, ((uint64_t)si << 40) | (l << 3) | (uint64_t)v);
            const double uv = rng_uniform(ruu);
            ru[(size_t)v * Ppad + k] = uv;
            if (v == 0) u0 = uv;
            if (varc == 0.0) {
                rq[(size_t)v * Ppad + k] = 0.0;
                ra[(size_t)v * Ppad + k] = S.logpic[v];
                rt[(size_t)v * Ppad + k] = 0.0;
                qv[v] = 0.0; av[v] = S.logpic[v];
            } else {
                const double iv = 1.0 / varc;
                const double lhsv = t2r + iv;
                const double ilhs = 1.0 / lhsv;
                const double prod = varc * lhsv;
                const double lg = det_log(prod);
                const double hl = 0.5 * lg;
                const double sd = det_sqrt(ilhs);
                rq[(size_t)v * Ppad + k] = ilhs;
                ra[(size_t)v * Ppad + k] = S.logpic[v] - hl;
                rt[(size_t)v * Ppad + k] = sd * zr;
                qv[v] = ilhs; av[v] = S.logpic[v] - hl;
            }
        }
        // Lazy threshold of the class search (speed only): with a zero first class the search ends at class 0
        // iff e_0 >= u_0 S, i.e. F(hs) = sum_{v>=1} exp(q_v hs + a_v - a_0) <= 1 / u_0 - 1, and F grows with hs = rhs^2 / 2.  h* below
        // satisfies F(h*) <= (1 - 1e-6)(1 / u_0 - 1) AS EVALUATED HERE (bisection, its lower end): for hs <= h* the comparison of the
        // full evaluation holds with a relative margin of 1e-6 (1 - u_0) >= 1e-9, six orders above its rounding -- such a locus without an
        // old effect is passed over without being evaluated, with the result the evaluation would give (class 1, dlt = -beta).
        // -1: never lazy.  The value travels in the slot of class 0's sd * z, which a zero class does not use.
        if (S.K >= 2 && qv[0] == 0.0) {
            double hst = -1.0;
            if (u0 > 0.0 && u0 <= 0.999) {
                const double tau = (1.0 / u0 - 1.0) * (1.0 - 1e-6);
                const double lt = det_log(tau);
                double dv[NGP_RMAX];
                double hi = __builtin_inf();
                bool ok = true;
                for (int v = 1; v < S.K; v++) {
                    dv[v] = (av[v] - av[0]) - lt;
                    if (qv[v] > 0.0) { const double hb = -dv[v] / qv[v]; hi = (hb < hi) ? hb : hi; }
                    else if (!(dv[v] <= 0.0)) ok = false;
                }
                if (ok && hi >= 0.0) {  // every exponent below is <= 0 on [0, hi]
                    double g0 = 0.0;
                    for (int v = 1; v < S.K; v++) g0 += det_exp(dv[v]);
                    if (g0 <= 1.0) {
                        if (hi == __builtin_inf()) hst = 1.7976931348623157e308;
                        else {
                            double lo = 0.0;
                            for (int itb = 0; itb < 30; itb++) {
                                const double mid = 0.5 * (lo + hi);
                                double g = 0.0;
                                for (int v = 1; v < S.K; v++) g += det_exp(__builtin_fma(mid, qv[v], dv[v]));
                                if (g <= 1.0) lo = mid; else hi = mid;
                            }
                            hst = lo;
                        }
                    }
                }
            }
#ifdef NGP_NO_LAZY  /* timing experiment: every locus evaluated */
            hst = -1.0;
#endif
            rt[k] = hst;  // (class 0)
        }
        c[k] = 0.0;
        w[k] = 0.0;
        q[k] = -1.0;
        T[k] = 1.0;
        chi[k] = 1.0;
        return;
    }
    double vbk = varBeta[vbidx[k]];
    double m = mpm[k];
    double t1 = m * iVarE;
    double t2 = t1 + lhs0[k];
    double ivb = 1.0 / vbk;
    double lhs = t2 + ivb;
    double ilhs = 1.0 / lhs;
    double cc = iVarE * ilhs;
    double s = det_sqrt(ilhs);
    Rng r = rng_seed(seed, chain, it, NGP_KIND_BETA_NORMAL, key);
    double z = rng_normal(r);
    double sz = s * z;
    double tw = (S.method == 2) ? 0.0 : rhs0[k] * ilhs;  // BayesC drops M.rhs (src/functions.jl:220)
    tw = tw + sz;
    c[k] = cc;
    w[k] = tw - beta[k];
    if (S.method >= 1) {  // BayesB / BayesC inclusion step (src/functions.jl:169-174, 210-217)
        double v0 = m * varE;
        double m2 = m * m;
        m2 = m2 * vbk;
        double v1 = m2 + v0;
        double i1 = 1.0 / v1, i0 = 1.0 / v0;
        double dq = i1 - i0;
        double qq = 0.5 * dq;
        Rng ru = rng_seed(seed, chain, it, NGP_KIND_B_UNIFORM, key);
        double u = rng_uniform(ru);
        double om = 1.0 - u;
        double Lu = det_log(om) - det_log(u);
        double dl = det_log(v1) - det_log(v0);
        dl = 0.5 * dl;
        double TT = Lu - dl;
        double lp = S.logPi0 - S.logPi1;
        TT = TT - lp;
        // inclusion test r^2*q < T rearranged so that the serial chain only compares |r| with a threshold
        double st;
        if (qq < 0.0) {
            double thr2 = TT / qq;
            st = (thr2 < 0.0) ? -1.0 : det_sqrt(thr2);
        } else {
            st = (0.0 < TT) ? -1.0 : __builtin_huge_val();
        }
        // threshold on f = c r instead of r (c > 0 when finite): thr = st c; -1 always, inf never
        q[k] = (st < 0.0) ? -1.0 : ((st == __builtin_huge_val()) ? st : st * cc);
        T[k] = TT;
        if (S.method == 1) {
            Rng rc = rng_seed(seed, chain, it, NGP_KIND_B_LOCUS_CHI2, key);
            chi[k] = rng_chisq(rc, S.df + 1.0);
        } else {
            chi[k] = 1.0;
        }
    } else {
        q[k] = -1.0;  // BayesPR: always included
        T[k] = 1.0;
        chi[k] = 1.0;
    }
}

// ------------------------------------------------------------------------------------------
// Inverse form of the LINEAR blocks.  In a block whose lanes are all BayesPR (always included,
// src/functions.jl:124-136) or unowned -- or that belongs to a Tuple set (src/functions.jl:140-154) -- the serial steps
//   dlt_k = e_k,  e_j += H_jk dlt_k  (j in a later locus than k)
// are the forward substitution of  L dlt = e0,  L = I - H' : dlt = T e0 with T = inv(L).  L depends on this iteration's
// coefficients (k_prep) and on the Gram block only -- not on the residual -- so T is formed here, before the sweep, for every such
// block at once (one wave per block, the whole device), and the sampler's critical wave replaces its dependent cross-lane steps
// (v_readlane -> SGPR -> fma: 24.6 clocks each) by one 64 x 64 product from LDS.
// Lane i forms column i of T by forward substitution, row after row in the order of the steps it replaces:
//   x_m = 0 (m < i), 1 (m = i), else  -(c_m acc_m)                                   BayesPR / unowned block (K = 0)
//                                     -(C_m[0] acc_g (+) fma(C_m[b], acc_{g+b}))       block of a k-set Tuple, g = first column of m's locus
//   then acc_j = fma(G[m][j], x_m, acc_j) for the columns j of later loci (K = 0: j > m)
// (G[m][j] is wave-uniform: broadcast reads from the block's copy in LDS; nothing but the coefficients of row m crosses a lane).
// Output: tinv[t][i][j] = T[j][i], the layout of the one-sided diagonal Gram block it stands in for (lane j of the chain reads row j
// with stride 64).  blin[t]: 0 = not linear, 1 = BayesPR / unowned, 1 + k = block of a k-set Tuple.
// ------------------------------------------------------------------------------------------
template <int K>
__host__ __device__ constexpr int tinv_gbase(const int j) { return K <= 1 ? j : (K == 2 ? (j & ~1) : (K == 3 ? (j / 3) * 3 : (j & ~3))); }
// One wave forms T of one block: the Gram block goes through LDS once (coalesced), its rows come back as broadcast reads (G[m][j] is
// the same for every lane: the lanes differ in the column of T they carry), the substitution is unrolled completely -- 2016 fma with
// the accumulators in registers.  (A first version fed G through scalar loads into SGPRs, double-buffered by hand: the compiler
// spilled SGPR tuples whose loads were still in flight -- it cannot know that -- and T came out wrong in a few rows.)
template <int K>
__device__ __attribute__((always_inline)) inline void tinv_block(const double *__restrict__ G1, const double (&cv)[K > 0 ? K : 1], const int i,
                                                                 double *__restrict__ dst_col, double *__restrict__ Gs) {
#pragma unroll 8
    for (int idx = i; idx < NGP_BLK * NGP_BLK; idx += NGP_BLK) Gs[idx] = G1[idx];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (one wave: LDS serves it in order)
    double acc[NGP_BLK];
#pragma unroll
    for (int j = 0; j < NGP_BLK; j++) acc[j] = 0.0;
#pragma unroll
    for (int m = 0; m < NGP_BLK; m++) {
        // every acc of m's locus is complete: x of the whole locus at its first column (the sums acc_g .. acc_{g+K-1} feed all K of them,
        // so none is replaced by its x before the last one is formed)
        if constexpr (K <= 1) {
            const double t = readlane_d(cv[0], m) * acc[m];
            acc[m] = (i == m) ? 1.0 : ((i > m) ? 0.0 : -t);
        } else {
            const int g = tinv_gbase<K>(m);
            if (m == g) {
                double tx[K];
#pragma unroll
                for (int bp = 0; bp < K; bp++) {
                    if (g + K > NGP_BLK) {
                        tx[bp] = 0.0;  // (the unused last column of a 3-set block)
                    } else {
                        double t = readlane_d(cv[0], g + bp) * acc[g];
#pragma unroll
                        for (int b = 1; b < K; b++) t = __builtin_fma(readlane_d(cv[b], g + bp), acc[g + b], t);
                        tx[bp] = t;
                    }
                }
#pragma unroll
                for (int bp = 0; bp < K; bp++)
                    if (g + bp < NGP_BLK) acc[g + bp] = (i == g + bp) ? 1.0 : ((i > g + bp) ? 0.0 : -tx[bp]);
            }
        }
        const double xm = acc[m];
#pragma unroll
        for (int j = m + 1; j < NGP_BLK; j++) {
            if (tinv_gbase<K>(j) <= m) continue;  // (K = 0: never) same locus as m
            acc[j] = __builtin_fma(Gs[m * NGP_BLK + j], xm, acc[j]);
        }
    }
    double2 *dst = (double2 *)dst_col;
#pragma unroll
    for (int jj = 0; jj < NGP_BLK / 2; jj++) dst[jj] = make_double2(acc[2 * jj], acc[2 * jj + 1]);
}
__global__ __launch_bounds__(64) void k_tinv(const double *__restrict__ gramx, int D, const double *__restrict__ c,
                                             const unsigned *__restrict__ blin, double *__restrict__ tinv, const unsigned *__restrict__ abort_w,
                                             const double *__restrict__ tupc, long long Ppad) {
    if (abort_w && *abort_w != 0u) return;
    const int t = blockIdx.x, i = threadIdx.x;
    const unsigned code = blin[t];  // (which blocks are linear is static for a model: the host's table, ngp_api.hip sync_linear_blocks)
    if (code == 0u) return;
    __shared__ double Gs[NGP_BLK * NGP_BLK];
    const double *G1 = gramx + (size_t)t * D * (NGP_BLK * NGP_BLK);  // element (m, j) at m * 64 + j, zero for j <= m
    double *dst = tinv + (size_t)t * (NGP_BLK * NGP_BLK) + (size_t)i * NGP_BLK;
    const size_t col = (size_t)t * NGP_BLK + i;
    if (code == 1u) {
        const double cv[1] = {c[col]};  // lane m holds c_m
        tinv_block<0>(G1, cv, i, dst, Gs);
    } else if (code == 2u) {
        const double cv[1] = {tupc[col]};
        tinv_block<1>(G1, cv, i, dst, Gs);
    } else if (code == 3u) {
        const double cv[2] = {tupc[col], tupc[(size_t)Ppad + col]};
        tinv_block<2>(G1, cv, i, dst, Gs);
    } else if (code == 4u) {
        const double cv[3] = {tupc[col], tupc[(size_t)Ppad + col], tupc[(size_t)2 * Ppad + col]};
        tinv_block<3>(G1, cv, i, dst, Gs);
    } else {
        const double cv[4] = {tupc[col], tupc[(size_t)Ppad + col], tupc[(size_t)2 * Ppad + col], tupc[(size_t)3 * Ppad + col]};
        tinv_block<4>(G1, cv, i, dst, Gs);
    }
}

// ------------------------------------------------------------------------------------------
// streaming step of block t: (U) ycorr -= X_{t-1} dlt_{t-1}  then  (G) partial r = X_t' ycorr.
// grid = S shards, 256 threads; tile (t,s) = [64 columns][R rows] fp32, contiguous.
// Tile (t, s) = R rows x 64 columns of fp32, stored QUAD-MAJOR: element (row i, column j) sits at
// (i >> 2) * 256 + j * 4 + (i & 3).  One quad (rows 4p..4p+3 of all 64 columns) is 1 KiB contiguous -- the unit of the
// LDS-DMA, of the GEMV's 16-byte reads (lane = column: consecutive lanes, consecutive 16 bytes) and of the streamers'
// row-half pipelining.

// dynamic LDS: R*256 (tile) + R*8 (ycorr shard) + 4096 (chain partials)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_step(const float *__restrict__ tiles, double *__restrict__ ycorr,
                                              const double *__restrict__ dlt, double *__restrict__ part, int R, int S, int t,
                                              int do_upd, int do_gemv) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *tl = (float *)smem;
    double *ys = (double *)(smem + (size_t)R * 256);
    double *red = ys + R;
    const int s = blockIdx.x, tid = threadIdx.x;
    const size_t tile_elems = (size_t)R * NGP_BLK;
    double *yg = ycorr + (size_t)s * R;
    if (do_upd) {
        const float *tp = tiles + ((size_t)(t - 1) * S + s) * tile_elems;
        for (int i = tid; i < R; i += 256) {
            // y_i -= sum_j x_ij dlt_j: eight chains of eight columns, then a fixed pairwise tree
            double p8[8];
#pragma unroll
            for (int c = 0; c < 8; c++) {
                double p = 0.0;
#pragma unroll
                for (int jj = 0; jj < 8; jj++) p = __builtin_fma((double)tp[tile_off(i, 8 * c + jj)], dlt[8 * c + jj], p);
                p8[c] = p;
            }
            double T = ((p8[0] + p8[1]) + (p8[2] + p8[3])) + ((p8[4] + p8[5]) + (p8[6] + p8[7]));
            double yv = yg[i] - T;
            yg[i] = yv;
            ys[i] = yv;
        }
    } else {
        for (int i = tid; i < R; i += 256) ys[i] = yg[i];
    }
    if (!do_gemv) return;
    const float4 *src = (const float4 *)(tiles + ((size_t)t * S + s) * tile_elems);
    float4 *dst = (float4 *)tl;
    for (int idx = tid; idx < R * 16; idx += 256) dst[idx] = src[idx];
    __syncthreads();
    const int wv = tid >> 6, j = tid & 63;
    const float *col = tl + 4 * j;  // quad qd of column j: tl + 256 qd + 4 j
    // 8 chains over strided row quads (chain c: quads c, c+8, ...); this 4-wave kernel runs chains wv and wv+4
#pragma unroll
    for (int h = 0; h < 2; h++) {
        double acc = 0.0;
        for (int qd = wv + 4 * h; qd < (R >> 2); qd += 8) {
            float4 x = *(const float4 *)(col + 256 * qd);
            const double *yq = ys + 4 * qd;
            acc = __builtin_fma((double)x.x, yq[0], acc);
            acc = __builtin_fma((double)x.y, yq[1], acc);
            acc = __builtin_fma((double)x.z, yq[2], acc);
            acc = __builtin_fma((double)x.w, yq[3], acc);
        }
        red[(wv + 4 * h) * 64 + j] = acc;
    }
    __syncthreads();
    if (wv == 0) {
        double p = ((red[j] + red[64 + j]) + (red[128 + j] + red[192 + j])) + ((red[256 + j] + red[320 + j]) + (red[384 + j] + red[448 + j]));
        part[(size_t)s * NGP_BLK + j] = p;
    }
}

// ------------------------------------------------------------------------------------------
// block recursion of block t: reduce the S shard partials (groups of 32, sequential), then
// the 64-step serial chain on wave 0 with the Gram block in registers.  ONE workgroup.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_recur(const double *__restrict__ part, const double *__restrict__ gramx, int D, int S, int t,
                                               double *__restrict__ beta, uint8_t *__restrict__ delta,
                                               const double *__restrict__ c, const double *__restrict__ w,
                                               const double *__restrict__ q, const double *__restrict__ mpm,
                                               const double *__restrict__ chi, const int8_t *__restrict__ setof,
                                               const int32_t *__restrict__ vbidx, DSet *__restrict__ sets,
                                               double *__restrict__ varBeta, double *__restrict__ dlt, const double *__restrict__ rcls,
                                               long long Ppad, const double *__restrict__ rhs0, const DScal *__restrict__ sc,
                                               const DTup *__restrict__ tup, const double *__restrict__ tupc, const double *__restrict__ tupg,
                                               const double *__restrict__ tinv, const unsigned *__restrict__ blin) {
    // the S shard partials as the order-free fixed-point sum of the persistent sweep (ngp_common.h): four waves, integer addition
    __shared__ double gs[32 * NGP_BLK];
    __shared__ long long qs[4 * NGP_BLK];
    const int tid = threadIdx.x, j = tid & 63, g4 = tid >> 6;
    const double fxs = sc->fx_scale, fxi = sc->fx_inv;
    {
        long long q = 0;
        for (int s = g4; s < S; s += 4) q += fx_from_f64(part[(size_t)s * NGP_BLK + j] * fxs);
        qs[g4 * NGP_BLK + j] = q;
    }
    __syncthreads();
    if (g4 != 0) return;
    const double tot = fx_to_f64(((qs[j] + qs[NGP_BLK + j]) + (qs[2 * NGP_BLK + j] + qs[3 * NGP_BLK + j]))) * fxi;
    const long long k = (long long)t * NGP_BLK + j;
    const double *G = gramx + (size_t)t * D * NGP_BLK * NGP_BLK;
    const double gd = mpm[k];
    const double bo = beta[k], cc = c[k], ww = w[k], st = q[k];
    const double r = __builtin_fma(gd, bo, tot);
    const int si0 = setof[k];
    const int meth0 = (si0 >= 0) ? sets[si0].method : -1;
    if (tinv && blin[t] != 0u) {  // linear block: dlt = T e0 (k_tinv)
        const double *Tt = tinv + (size_t)t * (NGP_BLK * NGP_BLK) + j;  // element (i, j) = T[j][i]
        double e0 = __builtin_fma(r, cc, ww);
        if (blin[t] > 1u) {  // a Tuple block: e0 from the k x k conditional of the lane's locus
            const DTup Tp = tup[__builtin_amdgcn_readfirstlane(__shfl(si0, 0))];  // (column 0 of a Tuple block is always the set's)
            const TupLane TL = load_tuplane(tupc, tupg, w, Ppad, k);
            const long long first_locus = ((long long)t - Tp.col0 / NGP_BLK) * (NGP_BLK / Tp.k);
            e0 = tuple_e0(Tp.k, tuple_nvalid(Tp.k, Tp.nloc, first_locus), j, tot, bo, TL);
        }
        gs[j] = e0;  // (only wave 0 is left and it has read its group sums: LDS serves a wave in order, no barrier needed)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        double s4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 16
        for (int i = 0; i < NGP_BLK; i += 4) {
            s4[0] = __builtin_fma(Tt[(size_t)(i + 0) * NGP_BLK], gs[i + 0], s4[0]);
            s4[1] = __builtin_fma(Tt[(size_t)(i + 1) * NGP_BLK], gs[i + 1], s4[1]);
            s4[2] = __builtin_fma(Tt[(size_t)(i + 2) * NGP_BLK], gs[i + 2], s4[2]);
            s4[3] = __builtin_fma(Tt[(size_t)(i + 3) * NGP_BLK], gs[i + 3], s4[3]);
        }
        const double dfin = (s4[0] + s4[1]) + (s4[2] + s4[3]);
        beta[k] = bo + dfin;
        delta[k] = (uint8_t)1;
        dlt[j] = dfin;
        return;
    }
    {
        const unsigned long long tm = __ballot(meth0 == NGP_METHOD_TUPLE_DEV);
        if (tm != 0ull) {  // a block of a Tuple set: one step per locus (tuple_chain)
            const int sit = __builtin_amdgcn_readfirstlane(__shfl(si0, __builtin_ctzll(tm)));
            const DTup Tp = tup[sit];
            const TupLane TL = load_tuplane(tupc, tupg, w, Ppad, k);
            const long long first_locus = ((long long)t - Tp.col0 / NGP_BLK) * (NGP_BLK / Tp.k);
            const double dfin = tuple_chain(Tp.k, Tp.nloc, first_locus, j, tot, bo, TL, [&](int sl, int cc2) { return G[(size_t)sl * NGP_BLK + cc2]; });
            beta[k] = bo + dfin;
            delta[k] = (uint8_t)1;
            dlt[j] = dfin;
            return;
        }
    }
    if (__ballot(meth0 == 3) != 0ull) {  // a BayesR locus in the block: r-form chain (eval_rform)
        RLane RL = empty_rlane();
        if (meth0 == 3) RL = load_rlane(rcls, Ppad, k, sets[si0].K, rhs0);
        const double iVarE = sc->iVarE;
        double rcur = r, dfin = 0.0;
        int cfin = 1, kstart = 0;
        for (int guard = 0; guard < NGP_BLK + 1; ++guard) {
            double cand;
            int cls;
            eval_rform(meth0, rcur, bo, cc, ww, st, RL, iVarE, cand, cls);
            if (j >= kstart) { dfin = cand; cfin = cls; }
            const unsigned long long todo = __ballot(cand != 0.0) & (~0ull << kstart);
            if (!todo) break;
            const int kk = __builtin_ctzll(todo);
            const double dk = readlane_d(cand, kk);
            const double Hk = -(G[(size_t)kk * NGP_BLK + j]);
            rcur = __builtin_fma(Hk, dk, rcur);
            kstart = kk + 1;
            if (kstart >= NGP_BLK) break;
        }
        const double bn = bo + dfin;
        beta[k] = bn;
        delta[k] = (uint8_t)cfin;
        dlt[j] = dfin;
        if (meth0 == 1) {
            double vb = 0.0;
            if (cfin) {
                double tt = sets[si0].sdf;
                double b2 = bn * bn;
                tt = tt + b2;
                vb = tt / chi[k];
                atomicAdd(&sets[si0].nloci, 1);
            }
            varBeta[vbidx[k]] = vb;
        } else if (meth0 == 2) {
            if (cfin) atomicAdd(&sets[si0].nloci, 1);
        } else if (meth0 == 3) {
            atomicAdd(&sets[si0].ncls[cfin - 1], 1);
        }
        return;
    }
    double Gr[NGP_BLK];
#pragma unroll
    for (int kk = 0; kk < NGP_BLK; kk++) Gr[kk] = G[kk * NGP_BLK + j];
    // scaled recursion: e = c r + w (candidate draw), f = c r (inclusion test |f| > thr), H_k = -(c G[.][k])
    double e = __builtin_fma(r, cc, ww), f = r * cc;
#pragma unroll
    for (int kk = 0; kk < NGP_BLK; kk++) {
        int in = __builtin_fabs(f) > st;
        double dl = in ? e : -bo;
        double dk = readlane_d(dl, kk);
        double H = -(cc * Gr[kk]);  // -0 for lanes <= kk: their e, f stay frozen at their own step's value
        e = __builtin_fma(H, dk, e);
        f = __builtin_fma(H, dk, f);
    }
    const int isave = __builtin_fabs(f) > st;
    const double dsave = isave ? e : -bo;
    const double bn = bo + dsave;
    beta[k] = bn;
    delta[k] = (uint8_t)isave;
    dlt[j] = dsave;
    const int si = setof[k];
    if (si >= 0 && sets[si].method == 1) {
        double vb = 0.0;
        if (isave) {
            double tt = sets[si].sdf;
            double b2 = bn * bn;
            tt = tt + b2;
            vb = tt / chi[k];
            atomicAdd(&sets[si].nloci, 1);
        }
        varBeta[vbidx[k]] = vb;
    } else if (si >= 0 && sets[si].method == 2) {
        if (isave) atomicAdd(&sets[si].nloci, 1);  // BayesC: one variance per set, drawn after the sweep
    }
}

// ------------------------------------------------------------------------------------------
// region variances of BayesPR sets (functions.jl:135, :509-511): 256-locus segment partials,
// then one thread per region.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_regssq(long long nseg, const long long *__restrict__ seg_k0,
                                                const int32_t *__restrict__ seg_len, const double *__restrict__ beta,
                                                double *__restrict__ segpart, const unsigned *__restrict__ abort_w) {
    if (abort_w && *abort_w != 0u) return;  // an earlier sweep of this call gave up (ngp_sweep_args.h, abort_w)
    // one wave per 256-locus segment: lane l takes loci l, l+64, l+128, l+192 (coalesced), then the xor butterfly
    const long long sg = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (sg >= nseg) return;
    const int lane = threadIdx.x & 63;
    const double *b = beta + seg_k0[sg];
    const int n = seg_len[sg];
    double a = 0.0;
#pragma unroll
    for (int m = 0; m < 4; m++) {
        const int i = lane + 64 * m;
        const double v = b[min(i, n - 1)];
        if (i < n) a = __builtin_fma(v, v, a);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) a = a + __shfl_xor(a, off);
    if (lane == 0) segpart[sg] = a;
}


// BayesR: sum of beta^2 / vClass[class] over the loci of non-zero classes (sumS, src/functions.jl:272-273), same segment
// pattern; overwrites the plain sums of the segments that belong to a BayesR set
__global__ __launch_bounds__(256) void k_rssq(long long nseg, const long long *__restrict__ seg_k0, const int32_t *__restrict__ seg_len,
                                              const int32_t *__restrict__ seg_set, const DSet *__restrict__ sets,
                                              const double *__restrict__ beta, const uint8_t *__restrict__ delta,
                                              double *__restrict__ segpart, const unsigned *__restrict__ abort_w) {
    if (abort_w && *abort_w != 0u) return;  // an earlier sweep of this call gave up (ngp_sweep_args.h, abort_w)
    const long long sg = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (sg >= nseg) return;
    const int si = seg_set[sg];
    if (sets[si].method != 3) return;
    const int lane = threadIdx.x & 63;
    const long long k0 = seg_k0[sg];
    const int n = seg_len[sg];
    double a = 0.0;
#pragma unroll
    for (int m = 0; m < 4; m++) {
        const int i = lane + 64 * m;
        if (i < n) {
            const double bv = beta[k0 + i];
            const int cl = (int)delta[k0 + i] - 1;
            const double vc = sets[si].vcls[cl < 0 ? 0 : cl];
            if (vc != 0.0) {
                const double b2 = bv * bv;
                const double term = b2 / vc;
                a = a + term;
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) a = a + __shfl_xor(a, off);
    if (lane == 0) segpart[sg] = a;
}

__global__ __launch_bounds__(64) void k_regdraw(long long nreg, const DReg *__restrict__ regs, const double *__restrict__ segpart,
                                                const DSet *__restrict__ sets, double *__restrict__ varBeta, int active_set,
                                                const double *__restrict__ regchi, uint64_t seed, uint64_t chain, uint64_t it, const unsigned *__restrict__ abort_w) {
    if (abort_w && *abort_w != 0u) return;  // an earlier sweep of this call gave up (ngp_sweep_args.h, abort_w)
    long long rg = (long long)blockIdx.x * 64 + threadIdx.x;
    if (rg >= nreg) return;
    const DReg R = regs[rg];
    if (active_set >= 0 && R.set != active_set) return;
    // segments of the region added in order; the loads of 32 segments go out together (one dependent load per add made this
    // kernel 34 us for a single region of 100,000 loci: 391 segments)
    double tot = segpart[R.seg0];
    int sg = 1;
    for (; sg + 32 <= R.nseg; sg += 32) {
        double v[32];
#pragma unroll
        for (int i = 0; i < 32; i++) v[i] = segpart[R.seg0 + sg + i];
#pragma unroll
        for (int i = 0; i < 32; i++) tot = tot + v[i];
    }
    for (; sg < R.nseg; sg++) tot = tot + segpart[R.seg0 + sg];
    const DSet S = sets[R.set];
    double ch = regchi[rg];  // chi-square(df + n_r) of this iteration, drawn ahead of the sweep by k_prep
    if (S.method == 2) {     // BayesC (src/functions.jl:231): df + number of loci the sweep has just included
        Rng rr = rng_seed(seed, chain, it, NGP_KIND_REGION_CHI2, ((uint64_t)R.set << 40) | (uint64_t)R.rg);
        ch = rng_chisq(rr, S.df + (double)S.nloci);
    }
    if (S.method == 3) {     // BayesR (src/functions.jl:281, :518-520): df + loci in non-zero classes; tot is sumS (k_rssq)
        long long nnz = 0;
        for (int v = 0; v < S.K; v++)
            if (S.vcls[v] != 0.0) nnz += S.ncls[v];
        Rng rr = rng_seed(seed, chain, it, NGP_KIND_REGION_CHI2, ((uint64_t)R.set << 40));
        ch = rng_chisq(rr, S.df + (double)nnz);
    }
    double tt = S.scale * S.df;
    tt = tt + tot;
    varBeta[R.vb] = tt / ch;
}

// ------------------------------------------------------------------------------------------
// Tuple sets: region variance MATRICES (src/functions.jl:152, :513-516).  Sb = B_r'B_r entry by entry in the segment pattern of
// k_regssq (a wave per 256 loci of a region; lane l: loci l, l+64, l+128, l+192 by fma; xor butterfly), then one thread per region:
// segments in order, Psi = scale + Sb, varBeta_r ~ InverseWishart(df + n_r, Psi) by Bartlett's construction on keyed draws
// (oracle/ngp_oracle.c t_inverse_wishart, operation for operation; k = 1: Psi / chi2, the Symbol path's form and key).
// ------------------------------------------------------------------------------------------
struct DTReg {  // one variance region of a tuple set
    long long seg0, rg, n;
    int nseg, set;
};
#define NGP_TPAIRS 10  // entries a <= b of a symmetric 4 x 4 matrix
__global__ __launch_bounds__(256) void k_tuple_ssq(long long nseg, const long long *__restrict__ seg_l0, const int32_t *__restrict__ seg_len,
                                                   const int32_t *__restrict__ seg_set, const DTup *__restrict__ tup,
                                                   const double *__restrict__ beta, double *__restrict__ tsegpart,
                                                   const unsigned *__restrict__ abort_w) {
    if (abort_w && *abort_w != 0u) return;
    const long long sg = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (sg >= nseg) return;
    const int lane = threadIdx.x & 63;
    const DTup Tp = tup[seg_set[sg]];
    const long long l0 = seg_l0[sg];
    const int n = seg_len[sg], kk = Tp.k;
    int pr = 0;
    for (int a = 0; a < kk; a++)
        for (int b = a; b < kk; b++, pr++) {
            double acc = 0.0;
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const int i = lane + 64 * m;
                if (i < n) {
                    const double ba = beta[tuple_col(Tp.col0, kk, l0 + i, a)], bb = beta[tuple_col(Tp.col0, kk, l0 + i, b)];
                    acc = __builtin_fma(ba, bb, acc);
                }
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) acc = acc + __shfl_xor(acc, off);
            if (lane == 0) tsegpart[sg * NGP_TPAIRS + pr] = acc;
        }
}
__global__ __launch_bounds__(64) void k_tuple_draw(long long nreg, const DTReg *__restrict__ regs, const double *__restrict__ tsegpart,
                                                   const DTup *__restrict__ tup, double *__restrict__ varBeta, int active_set,
                                                   uint64_t seed, uint64_t chain, uint64_t it, const unsigned *__restrict__ abort_w) {
    if (abort_w && *abort_w != 0u) return;
    const long long rgi = (long long)blockIdx.x * 64 + threadIdx.x;
    if (rgi >= nreg) return;
    const DTReg Rg = regs[rgi];
    if (active_set >= 0 && Rg.set != active_set) return;
    const DTup Tp = tup[Rg.set];
    const int k = Tp.k;
    double Psi[NGP_KMAX * NGP_KMAX];
    int pr = 0;
    for (int a = 0; a < k; a++)
        for (int b = a; b < k; b++, pr++) {
            double tot = tsegpart[Rg.seg0 * NGP_TPAIRS + pr];
            for (int sg = 1; sg < Rg.nseg; sg++) tot = tot + tsegpart[(Rg.seg0 + sg) * NGP_TPAIRS + pr];
            const double pab = Tp.scale[a * k + b] + tot, pba = Tp.scale[b * k + a] + tot;
            Psi[a * k + b] = pab; Psi[b * k + a] = pba;
        }
    const double nu = Tp.df + (double)Rg.n;
    double *out = varBeta + Tp.vb_off + Rg.rg * k * k;
    if (k == 1) {
        Rng r = rng_seed(seed, chain, it, NGP_KIND_REGION_CHI2, ((uint64_t)Rg.set << 40) | (uint64_t)Rg.rg);
        const double ch = rng_chisq(r, nu);
        out[0] = Psi[0] / ch;
        return;
    }
    double Pi[NGP_KMAX * NGP_KMAX], L[NGP_KMAX * NGP_KMAX], A[NGP_KMAX * NGP_KMAX], LA[NGP_KMAX * NGP_KMAX], W[NGP_KMAX * NGP_KMAX], res[NGP_KMAX * NGP_KMAX];
    int bad = t_spd_inv(Psi, k, Pi);
    bad |= t_chol(Pi, k, L);
    for (int a = 0; a < k * k; a++) A[a] = 0.0;
    for (int i = 0; i < k; i++)
        for (int j = 0; j <= i; j++) {
            Rng r = (i == 0) ? rng_seed(seed, chain, it, NGP_KIND_REGION_CHI2, ((uint64_t)Rg.set << 40) | (uint64_t)Rg.rg)
                             : rng_seed(seed, chain, it, NGP_KIND_T_WISHART, ((uint64_t)Rg.set << 40) | ((uint64_t)Rg.rg << 8) | ((uint64_t)i << 4) | (uint64_t)j);
            if (i == j) { const double ch = rng_chisq(r, nu - (double)i); A[i * k + i] = det_sqrt(ch); }
            else A[i * k + j] = rng_normal(r);
        }
    for (int i = 0; i < k; i++)
        for (int j = 0; j < k; j++) {
            double s = 0.0;
            for (int m = 0; m < k; m++) s = __builtin_fma(L[i * k + m], A[m * k + j], s);
            LA[i * k + j] = s;
        }
    for (int i = 0; i < k; i++)
        for (int j = 0; j < k; j++) {
            double s = 0.0;
            for (int m = 0; m < k; m++) s = __builtin_fma(LA[i * k + m], LA[j * k + m], s);
            W[i * k + j] = s;
        }
    bad |= t_spd_inv(W, k, res);
    for (int a = 0; a < k * k; a++) out[a] = bad ? __builtin_nan("") : res[a];
}
// X_l'X_l of every locus of a tuple set, row m for column (l, m): from the one-sided diagonal Gram block (entry [a][b] kept for b > a)
// and x'x in mpm.  Once per set.
__global__ __launch_bounds__(256) void k_tuple_gkk(const double *__restrict__ gramx, int D, const double *__restrict__ mpm, DTup Tp,
                                                   double *__restrict__ tupg, long long Ppad) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;  // locus * k + component
    if (idx >= Tp.nloc * Tp.k) return;
    const int k = Tp.k, m = (int)(idx % k);
    const long long l = idx / k, c = tuple_col(Tp.col0, k, l, m), t = c / NGP_BLK;
    const int jm = (int)(c % NGP_BLK);
    const double *G0 = gramx + (size_t)t * D * NGP_BLK * NGP_BLK;  // plane d = 0
    for (int b = 0; b < k; b++) {
        const int jb = jm - m + b;
        tupg[(size_t)b * Ppad + c] = (b == m) ? mpm[c] : G0[(size_t)min(jm, jb) * NGP_BLK + max(jm, jb)];
    }
}

// pi draw of BayesB sets (functions.jl:190-194, :531-533); also clears the inclusion counters
__global__ void k_pidraw(int nsets, DSet *__restrict__ sets, int active_set, uint64_t seed, uint64_t chain, uint64_t it, const unsigned *__restrict__ abort_w) {
    if (abort_w && *abort_w != 0u) return;  // an earlier sweep of this call gave up (ngp_sweep_args.h, abort_w)
    int si = threadIdx.x;
    if (si >= nsets) return;
    if (active_set >= 0 && si != active_set) return;
    DSet *S = &sets[si];
    if (S->method == 3) {  // BayesR: Dirichlet(nLoci + 1) as normalised gammas (src/functions.jl:284-288, :536-538)
        if (S->estPi) {
            double g[NGP_RMAX], gsum = 0.0;
            for (int v = 0; v < S->K; v++) {
                Rng rg = rng_seed(seed, chain, it, NGP_KIND_R_DIRICHLET, ((uint64_t)si << 40) | (uint64_t)v);
                g[v] = rng_gamma(rg, (double)S->ncls[v] + 1.0);
                gsum = gsum + g[v];
            }
            for (int v = 0; v < S->K; v++) {
                S->pic[v] = g[v] / gsum;
                S->logpic[v] = det_log(S->pic[v]);
            }
        }
        for (int v = 0; v < NGP_RMAX; v++) S->ncls[v] = 0;
        S->nloci = 0;
        return;
    }
    if (S->method >= 1 && S->estPi) {
        int nLoci = S->nloci;
        Rng r = rng_seed(seed, chain, it, NGP_KIND_PI_BETA, (uint64_t)si);
        double piIn = rng_beta(r, (double)nLoci + 1.0, (double)(S->ncol - nLoci) + 1.0);
        S->piHat0 = 1.0 - piIn;
        S->piHat1 = piIn;
        S->logPi0 = det_log(S->piHat0);
        S->logPi1 = det_log(piIn);
    }
    S->nloci = 0;
}

__global__ void k_set_pi(DSet *__restrict__ sets, int si, double p0, double p1) {
    sets[si].piHat0 = p0;
    sets[si].piHat1 = p1;
    sets[si].logPi0 = det_log(p0);
    sets[si].logPi1 = det_log(p1);
    sets[si].nloci = 0;
}
__global__ void k_set_class_state(DSet *__restrict__ sets, int si, int K, const double *__restrict__ pi, const double *__restrict__ sum_pi) {
    for (int v = 0; v < K; v++) {
        if (pi) { sets[si].pic[v] = pi[v]; sets[si].logpic[v] = det_log(pi[v]); }
        if (sum_pi) sets[si].sum_pic[v] = sum_pi[v];
    }
    for (int v = 0; v < NGP_RMAX; v++) sets[si].ncls[v] = 0;
}
__global__ void k_set_sum_pi(DSet *__restrict__ sets, int si, double s0, double s1) {
    sets[si].sum_pi0 = s0;
    sets[si].sum_pi1 = s1;
}
// per-iteration traces of selected effects, the first ntvb variances and pi of every set (bench.py: effective sample sizes)
__global__ __launch_bounds__(256) void k_add_inplace(double *__restrict__ out, const double *__restrict__ in, long long n) {
    const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
    if (k < n) out[k] = out[k] + in[k];
}
// the fine seam with device-resident state (ngp_sweep_set_dev): pi in and out, delta as the caller's 64-bit integers
__global__ void k_set_pi_dev(DSet *__restrict__ sets, int si, const double *__restrict__ pi) {  // (k_set_pi, the values read here)
    const double p0 = pi[0], p1 = pi[1];
    sets[si].piHat0 = p0;
    sets[si].piHat1 = p1;
    sets[si].logPi0 = det_log(p0);
    sets[si].logPi1 = det_log(p1);
    sets[si].nloci = 0;
}
__global__ void k_get_pi_dev(const DSet *__restrict__ sets, int si, double *__restrict__ pi) {
    pi[0] = sets[si].piHat0;
    pi[1] = sets[si].piHat1;
}
__global__ void k_delta_widen(const uint8_t *__restrict__ d8, long long *__restrict__ d64, long long n) {
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) d64[k] = d8[k];
}
__global__ void k_set_varE(DScal *__restrict__ sc, double varE) {
    sc->varE = varE;
    sc->iVarE = 1.0 / varE;
}

// posterior sums of a kept iteration (samplers.jl:56-103 writes rows; misc.jl:241-244 averages)
// end of an iteration, one launch: the per-iteration traces of selected effects / variances / pi (every iteration, if asked for)
// and the posterior sums (kept iterations only; src/samplers.jl:56-104)
__global__ __launch_bounds__(256) void k_post(int do_accum, long long P, long long nvb, int nsets, const double *__restrict__ beta,
                                              const uint8_t *__restrict__ delta, const double *__restrict__ varBeta,
                                              double *__restrict__ sum_beta, double *__restrict__ sum_beta2,
                                              double *__restrict__ sum_delta, double *__restrict__ sum_varBeta,
                                              DSet *__restrict__ sets, DScal *__restrict__ sc, int do_trace, long long ntl,
                                              const long long *__restrict__ loci, long long ntvb, double *__restrict__ tr_beta,
                                              double *__restrict__ tr_vb, double *__restrict__ tr_pi, long long idx, const unsigned *__restrict__ abort_w) {
    if (abort_w && *abort_w != 0u) return;  // an earlier sweep of this call gave up (ngp_sweep_args.h, abort_w)
    const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
    if (do_trace) {
        if (k < ntl) tr_beta[idx * ntl + k] = beta[loci[k]];
        if (k < ntvb) tr_vb[idx * ntvb + k] = varBeta[k];
        if (k < nsets) tr_pi[idx * nsets + k] = sets[k].piHat1;
    }
    if (!do_accum) return;
    if (k < P) {
        double b = beta[k];
        sum_beta[k] += b;
        sum_beta2[k] += b * b;
        sum_delta[k] += (double)delta[k];
    }
    if (k < nvb) sum_varBeta[k] += varBeta[k];
    if (k < nsets) {
        sets[k].sum_pi0 += sets[k].piHat0;
        sets[k].sum_pi1 += sets[k].piHat1;
        for (int v = 0; v < sets[k].K; v++) sets[k].sum_pic[v] += sets[k].pic[v];
    }
    if (k == 0) {
        sc->sum_varE += sc->varE;
        sc->sum_b += sc->b;
        sc->nKept += 1;
    }
}

// ------------------------------------------------------------------------------------------
// One kept sample into a slot of the sample ring (ngp_set_sample_file: what the reference appends as text rows of b / varE / beta<set> /
// delta<set> / pi<set> / var<set>Out at src/samplers.jl:56-104).  Record: int64 iteration (-1: invalid -- an earlier sweep of the
// call gave up and this iteration will be run again) | varE | b | b_fixed[nfix] | beta[P] | varBeta[nvb] | piHat[2 nsets] |
// class probabilities [nclass] | delta[P] as bytes.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sample_pack(unsigned char *__restrict__ rec, long long P, long long nvb, int nsets, long long nfix,
                                                     long long nclass, long long iter, const double *__restrict__ beta,
                                                     const uint8_t *__restrict__ delta, const double *__restrict__ varBeta,
                                                     const DSet *__restrict__ sets, const DScal *__restrict__ sc, const double *__restrict__ bfix,
                                                     const unsigned *__restrict__ abort_w) {
    const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
    long long *hd = (long long *)rec;
    double *d = (double *)rec + 1;
    if (abort_w && *abort_w != 0u) { if (k == 0) hd[0] = -1; return; }
    double *o_fix = d + 2, *o_beta = o_fix + nfix, *o_vb = o_beta + P, *o_pi = o_vb + nvb, *o_cls = o_pi + 2 * nsets;
    uint8_t *o_delta = (uint8_t *)(o_cls + nclass);
    if (k == 0) {
        hd[0] = iter; d[0] = sc->varE; d[1] = sc->b;
        long long c = 0;
        for (int s = 0; s < nsets; s++) {
            o_pi[2 * s] = sets[s].piHat0; o_pi[2 * s + 1] = sets[s].piHat1;
            for (int v = 0; v < sets[s].K; v++) o_cls[c++] = sets[s].pic[v];
        }
    }
    if (k < nfix) o_fix[k] = bfix[k];
    if (k < P) { o_beta[k] = beta[k]; o_delta[k] = delta[k]; }
    if (k < nvb) o_vb[k] = varBeta[k];
}

// ------------------------------------------------------------------------------------------
// set-up kernels: Gram blocks, synthetic panel, X*beta
// ------------------------------------------------------------------------------------------
// shard partial of Gx[t][d][k][j] = x_{t-d,k}' x_{t,j} (d = 0: the symmetric diagonal block): 256 threads,
// thread (tk,tj) owns a 4x4 sub-block; rows ascending; both tiles staged through LDS in chunks of RC rows
#define NGP_GRAM_RC 112
#define NGP_GRAM_LD (NGP_GRAM_RC + 4)  // padded row stride of the staging tiles (bank spread of the transposing writes)
__global__ __launch_bounds__(256) void k_gram_part(const float *__restrict__ tiles, double *__restrict__ gpart, int R, int S, int t0,
                                                   int d) {
    __shared__ __attribute__((aligned(16))) float ta[NGP_BLK * NGP_GRAM_LD];
    __shared__ __attribute__((aligned(16))) float tt[NGP_BLK * NGP_GRAM_LD];
    const int s = blockIdx.x, tb = blockIdx.y, tid = threadIdx.x;
    const int t = t0 + tb;
    const size_t tile_elems = (size_t)R * NGP_BLK;
    const float *src_t = tiles + ((size_t)t * S + s) * tile_elems;
    const float *src_a = tiles + ((size_t)(t - d) * S + s) * tile_elems;
    const int tk = tid >> 4, tj = tid & 15;
    double acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = 0.0;
    if (t - d < 0) return;  // block-uniform
    for (int i0 = 0; i0 < R; i0 += NGP_GRAM_RC) {
        const int rc = min(NGP_GRAM_RC, R - i0);  // multiple of 4
        __syncthreads();
        for (int idx = tid; idx < NGP_BLK * (rc >> 2); idx += 256) {  // quad-major source: lanes run over the columns
            const int j = idx & (NGP_BLK - 1), q4 = idx >> 6;
            *(float4 *)(tt + j * NGP_GRAM_LD + 4 * q4) = *(const float4 *)(src_t + (size_t)((i0 >> 2) + q4) * 256 + 4 * j);
            *(float4 *)(ta + j * NGP_GRAM_LD + 4 * q4) = *(const float4 *)(src_a + (size_t)((i0 >> 2) + q4) * 256 + 4 * j);
        }
        __syncthreads();
        for (int i = 0; i < rc; i += 4) {
            float4 xk[4], xj[4];
#pragma unroll
            for (int a = 0; a < 4; a++) {
                xk[a] = *(const float4 *)(ta + (4 * tk + a) * NGP_GRAM_LD + i);
                xj[a] = *(const float4 *)(tt + (4 * tj + a) * NGP_GRAM_LD + i);
            }
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    acc[a][b] = __builtin_fma((double)xk[a].x, (double)xj[b].x, acc[a][b]);
                    acc[a][b] = __builtin_fma((double)xk[a].y, (double)xj[b].y, acc[a][b]);
                    acc[a][b] = __builtin_fma((double)xk[a].z, (double)xj[b].z, acc[a][b]);
                    acc[a][b] = __builtin_fma((double)xk[a].w, (double)xj[b].w, acc[a][b]);
                }
        }
    }
    double *out = gpart + ((size_t)tb * S + s) * (NGP_BLK * NGP_BLK);
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) out[(4 * tk + a) * NGP_BLK + 4 * tj + b] = acc[a][b];
}

// The same shard partials on the matrix cores: v_mfma_f64_16x16x4_f64 (the Gram window is the one dense contraction of the path,
// SURVEY.md section 7.1).  256 threads = 4 waves; wave w owns the 16 columns j = 16 w .. 16 w + 15 of tile t against all 64 columns
// k of tile t - d: four 16 x 16 accumulators (k = 16 m .. 16 m + 15).  One MFMA contracts FOUR rows (a quad of the quad-major
// tile): lane l supplies A[k = 16 m + (l & 15)][row l >> 4] and B[row l >> 4][j = 16 w + (l & 15)] -- the 64 lanes of one operand
// read one contiguous 256-byte piece of the tile, straight from global memory (no LDS staging), converted to f64 once per
// element.  The matrix core adds the four products to the accumulator one after the other, rows ascending (products of two
// floats are exact in f64), so every entry is the sequential fma chain over the shard's rows that k_gram_part forms -- bit for
// bit (tests/test_gpu_parity.py::test_gram_and_mpm compares both with the oracle's chain).  D layout: lane l, register v holds
// row (l >> 4) + 4 v, column l & 15 (cdna_hip_programming.md, f64 MFMA).
typedef double ngp_d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_gram_part_mfma(const float *__restrict__ tiles, double *__restrict__ gpart, int R, int S, int t0, int d) {
    const int s = blockIdx.x, tb = blockIdx.y, tid = threadIdx.x;
    const int t = t0 + tb;
    if (t - d < 0) return;  // block-uniform
    const int w = tid >> 6, l = tid & 63;
    const size_t tile_elems = (size_t)R * NGP_BLK;
    const float *src_t = tiles + ((size_t)t * S + s) * tile_elems;
    const float *src_a = tiles + ((size_t)(t - d) * S + s) * tile_elems;
    // element (row 4 q + (l >> 4), column c0 + (l & 15)) of a quad-major tile: q * 256 + (c0 + (l & 15)) * 4 + (l >> 4)
    const int lo = (l & 15) * 4 + (l >> 4);
    const float *pb = src_t + 64 * w + lo;
    const float *pa = src_a + lo;
    ngp_d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    const int nq = R >> 2;
#pragma unroll 4
    for (int q = 0; q < nq; q++) {
        const double b = (double)pb[(size_t)q * 256];
        const double a0 = (double)pa[(size_t)q * 256], a1 = (double)pa[(size_t)q * 256 + 64], a2 = (double)pa[(size_t)q * 256 + 128],
                     a3 = (double)pa[(size_t)q * 256 + 192];
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, b, acc3, 0, 0, 0);
    }
    double *out = gpart + ((size_t)tb * S + s) * (NGP_BLK * NGP_BLK);
    const int j = 16 * w + (l & 15), kr = l >> 4;
#pragma unroll
    for (int v = 0; v < 4; v++) {
        out[(size_t)(0 + kr + 4 * v) * NGP_BLK + j] = acc0[v];
        out[(size_t)(16 + kr + 4 * v) * NGP_BLK + j] = acc1[v];
        out[(size_t)(32 + kr + 4 * v) * NGP_BLK + j] = acc2[v];
        out[(size_t)(48 + kr + 4 * v) * NGP_BLK + j] = acc3[v];
    }
}

// group sums over shards -> gramx[t][d][k][j]; d == 0 also fills mpm
__global__ __launch_bounds__(256) void k_gram_reduce(const double *__restrict__ gpart, double *__restrict__ gramx,
                                                     double *__restrict__ mpm, int S, int t0, int nb, int d, int D) {
    long long e = (long long)blockIdx.x * 256 + threadIdx.x;  // (tb, k, j)
    if (e >= (long long)nb * NGP_BLK * NGP_BLK) return;
    int tb = (int)(e / (NGP_BLK * NGP_BLK)), kj = (int)(e % (NGP_BLK * NGP_BLK));
    if (t0 + tb - d < 0) return;
    const double *p = gpart + (size_t)tb * S * (NGP_BLK * NGP_BLK) + kj;
    const int ngroups = (S + NGP_GRP - 1) / NGP_GRP;
    double tot = 0.0;
    for (int g = 0; g < ngroups; g++) {
        int s0 = g * NGP_GRP, s1 = min(s0 + NGP_GRP, S);
        double v = p[(size_t)s0 * (NGP_BLK * NGP_BLK)];
        for (int s = s0 + 1; s < s1; s++) v = v + p[(size_t)s * (NGP_BLK * NGP_BLK)];
        tot = (g == 0) ? v : tot + v;
    }
    int k = kj / NGP_BLK, j = kj % NGP_BLK;
    if (d == 0 && k == j) mpm[(size_t)(t0 + tb) * NGP_BLK + k] = tot;
    // the diagonal block is stored strictly "one-sided": entry [k][j] is kept for j > k only (what step k of the
    // recursion applies to the later lanes), zero elsewhere, so the chain needs no per-step masking; x'x lives in mpm
    // cross blocks (d >= 1): row pairs interleaved, element (k, j) at ((k >> 1) * 64 + j) * 2 + (k & 1) (ngp_sweep.h)
    const size_t off = (d == 0) ? (size_t)kj : ((size_t)(k >> 1) * NGP_BLK + j) * 2 + (k & 1);
    gramx[((size_t)(t0 + tb) * D + d) * (NGP_BLK * NGP_BLK) + off] = (d == 0 && j <= k) ? 0.0 : tot;
}

// synthetic genotypes: per-column mean of g_ij (integer sum), then centred fp32 tiles
__global__ __launch_bounds__(256) void k_gen_colmean(long long N, long long P, double lo, double hi, uint64_t pseed,
                                                     double *__restrict__ mu, uint32_t *__restrict__ thr) {
    __shared__ int wsum[4];
    const long long j = blockIdx.x;
    const int tid = threadIdx.x;
    double pj = panel_pj(pseed, j, lo, hi);
    uint32_t th = (uint32_t)(pj * 4294967296.0);
    uint64_t ck = panel_colkey(pseed, j);
    int sum = 0;
    for (long long i = tid; i < N; i += 256) sum += panel_gij(ck, i, th);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
    if ((tid & 63) == 0) wsum[tid >> 6] = sum;
    __syncthreads();
    if (tid == 0) {
        long long tot = (long long)wsum[0] + wsum[1] + wsum[2] + wsum[3];
        mu[j] = (double)tot / (double)N;
        thr[j] = th;
    }
}
__global__ __launch_bounds__(256) void k_gen_fill(float *__restrict__ tiles, long long N, long long P, int R, int S, uint64_t pseed,
                                                  const double *__restrict__ mu, const uint32_t *__restrict__ thr) {
    const int s = blockIdx.x;
    const long long t = blockIdx.y;
    float *tp = tiles + ((size_t)t * S + s) * ((size_t)R * NGP_BLK);
    for (int idx = threadIdx.x; idx < R * NGP_BLK; idx += 256) {
        const int ii = ((idx >> 8) << 2) + (idx & 3), jj = (idx >> 2) & (NGP_BLK - 1);  // idx is the quad-major offset
        long long i = (long long)s * R + ii, j = t * NGP_BLK + jj;
        float v = 0.0f;
        if (i < N && j < P) {
            int g = panel_gij(panel_colkey(pseed, j), i, thr[j]);
            v = (float)((double)g - mu[j]);
        }
        tp[idx] = v;
    }
}

// Host panels in Float64 / Float32 (src/prepMatVec.jl:116-131 hands the sampler a centred Float64 matrix per marker set): a staged
// chunk of whole columns (column-major, leading dimension ld) -> column means and centred fp32 quad-major tiles, on the device.
// The mean is the sequential sum over the rows divided by N (one thread per column: the order of the host loop this replaces,
// so panels uploaded before and after that change are bit-identical); a non-finite sum flags the chunk.
template <typename TIn>
__global__ __launch_bounds__(64) void k_cols_mean(const TIn *__restrict__ g, long long N, long long ld, long long nc, int centre,
                                                  double *__restrict__ mu, unsigned *__restrict__ bad) {
    const long long c = (long long)blockIdx.x * 64 + threadIdx.x;
    if (c >= nc) return;
    const TIn *col = g + (size_t)c * ld;
    double sum = 0.0;
    for (long long i = 0; i < N; i++) sum += (double)col[i];
    if (!(sum - sum == 0.0)) atomicOr(bad, 1u);  // inf or nan somewhere in the column
    mu[c] = centre ? sum / (double)N : 0.0;
}
// thread = (column c of the chunk, quad Q of the padded panel): rows 4 Q .. 4 Q + 3 live in one shard (R is a multiple of 4)
template <typename TIn>
__global__ __launch_bounds__(256) void k_cols_fill(float *__restrict__ tiles, const TIn *__restrict__ g, long long N, long long ld,
                                                   long long col0, int R, int S, const double *__restrict__ mu) {
    const long long Q = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long i0 = 4 * Q;
    if (i0 >= (long long)R * S) return;
    const long long c = blockIdx.y, j = col0 + c;
    const int s = (int)(i0 / R), ii = (int)(i0 - (long long)s * R), jj = (int)(j & (NGP_BLK - 1));
    const TIn *col = g + (size_t)c * ld;
    const double m = mu[c];
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; r++) v[r] = (i0 + r < N) ? (float)((double)col[i0 + r] - m) : 0.0f;
    float *tp = tiles + ((size_t)(j >> 6) * S + s) * ((size_t)R * NGP_BLK) + tile_off(ii, jj);
    *(float4 *)tp = make_float4(v[0], v[1], v[2], v[3]);
}

// compact storage, a column range of genotype codes: thread = (column c of the chunk, unit U of 16 rows of the padded panel) -> the
// unit's 16 bytes of that column (tile8_off: units of 16 rows, a column's 16 bytes contiguous); rows beyond N are zero
__global__ __launch_bounds__(256) void k_cols_fill8(uint8_t *__restrict__ tiles8, const uint8_t *__restrict__ g, long long N, long long ld,
                                                    long long col0, int R, int S) {
    const long long U = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long i0 = 16 * U;
    if (i0 >= (long long)R * S) return;
    const long long c = blockIdx.y, j = col0 + c;
    const int s = (int)(i0 / R), ii = (int)(i0 - (long long)s * R), jj = (int)(j & (NGP_BLK - 1));
    const uint8_t *col = g + (size_t)c * ld;
    unsigned w[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        unsigned v = 0;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const long long i = i0 + 4 * q + r;
            v |= (unsigned)((i < N) ? col[i] : (uint8_t)0) << (8 * r);
        }
        w[q] = v;
    }
    uint8_t *tp = tiles8 + ((size_t)(j >> 6) * S + s) * ((size_t)R * NGP_BLK) + tile8_off(ii, jj);
    *(uint4 *)tp = make_uint4(w[0], w[1], w[2], w[3]);
}

// compact genotype input (one byte per genotype, column-major staging chunk of ncols columns, leading dimension ld):
// integer column sums -> mean = sum / N exactly as the host path computes it, then centred fp32 quad-major tiles
__global__ __launch_bounds__(256) void k_u8_colmean(const uint8_t *__restrict__ G, long long N, long long ld, int centre,
                                                    double *__restrict__ mu) {
    __shared__ unsigned long long wsum[4];
    const long long jc = blockIdx.x;  // column of the chunk
    const uint8_t *col = G + (size_t)jc * ld;
    unsigned long long sum = 0;
    for (long long i = threadIdx.x; i < N; i += 256) sum += col[i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) mu[jc] = centre ? (double)(wsum[0] + wsum[1] + wsum[2] + wsum[3]) / (double)N : 0.0;
}
__global__ __launch_bounds__(256) void k_u8_fill(float *__restrict__ tiles, const uint8_t *__restrict__ G, long long N, long long ld,
                                                 long long ncols, int R, int S, long long t0, const double *__restrict__ mu) {
    const int s = blockIdx.x;
    const long long tb = blockIdx.y;  // block of the chunk
    float *tp = tiles + ((size_t)(t0 + tb) * S + s) * ((size_t)R * NGP_BLK);
    for (int idx = threadIdx.x; idx < R * NGP_BLK; idx += 256) {
        const int ii = ((idx >> 8) << 2) + (idx & 3), jj = (idx >> 2) & (NGP_BLK - 1);  // idx is the quad-major offset
        const long long i = (long long)s * R + ii, jc = tb * NGP_BLK + jj;
        float v = 0.0f;
        if (i < N && jc < ncols) v = (float)((double)G[(size_t)jc * ld + i] - mu[jc]);
        tp[idx] = v;
    }
}

// ------------------------------------------------------------------------------------------
// compact storage: byte tiles (genotype codes) + Float64 column means, centred analytically (ngp_sweep.h, variant 3)
// ------------------------------------------------------------------------------------------
// byte offset of element (row i, column j) inside a tile: units of 16 rows, a column's 16 bytes contiguous (one lane, one 16-byte read)

__global__ __launch_bounds__(256) void k_u8_fill8(uint8_t *__restrict__ tiles8, const uint8_t *__restrict__ G, long long N, long long ld,
                                                  long long ncols, int R, int S, long long t0) {
    const int s = blockIdx.x;
    const long long tb = blockIdx.y;  // block of the chunk
    uint8_t *tp = tiles8 + ((size_t)(t0 + tb) * S + s) * ((size_t)R * NGP_BLK);
    for (int idx = threadIdx.x; idx < R * NGP_BLK; idx += 256) {
        const int ii = ((idx >> 10) << 4) + (idx & 15), jj = (idx >> 4) & (NGP_BLK - 1);  // idx is the unit-major offset
        const long long i = (long long)s * R + ii, jc = tb * NGP_BLK + jj;
        tp[idx] = (i < N && jc < ncols) ? G[(size_t)jc * ld + i] : (uint8_t)0;
    }
}
__global__ __launch_bounds__(256) void k_gen_fill8(uint8_t *__restrict__ tiles8, long long N, long long P, int R, int S, uint64_t pseed,
                                                   const uint32_t *__restrict__ thr) {
    const int s = blockIdx.x;
    const long long t = blockIdx.y;
    uint8_t *tp = tiles8 + ((size_t)t * S + s) * ((size_t)R * NGP_BLK);
    for (int idx = threadIdx.x; idx < R * NGP_BLK; idx += 256) {
        const int ii = ((idx >> 10) << 4) + (idx & 15), jj = (idx >> 4) & (NGP_BLK - 1);
        const long long i = (long long)s * R + ii, j = t * NGP_BLK + jj;
        tp[idx] = (i < N && j < P) ? (uint8_t)panel_gij(panel_colkey(pseed, j), i, thr[j]) : (uint8_t)0;
    }
}

// shard partial of the integer dot products g_{t-d,k}' g_{t,j} (exact): thread (tk, tj) owns a 4x4 sub-block, 4 rows per v_dot4
#define NGP_GRAM8_RC 208
#define NGP_GRAM8_LD (NGP_GRAM8_RC + 4)
__global__ __launch_bounds__(256) void k_gram8_part(const uint8_t *__restrict__ tiles8, uint32_t *__restrict__ gpart, int R, int S, int t0, int d) {
    __shared__ __attribute__((aligned(16))) uint8_t ta[NGP_BLK * NGP_GRAM8_LD];
    __shared__ __attribute__((aligned(16))) uint8_t tt[NGP_BLK * NGP_GRAM8_LD];
    const int s = blockIdx.x, tb = blockIdx.y, tid = threadIdx.x;
    const int t = t0 + tb;
    if (t - d < 0) return;  // block-uniform
    const size_t tile_bytes = (size_t)R * NGP_BLK;
    const uint8_t *src_t = tiles8 + ((size_t)t * S + s) * tile_bytes;
    const uint8_t *src_a = tiles8 + ((size_t)(t - d) * S + s) * tile_bytes;
    const int tk = tid >> 4, tj = tid & 15;
    uint32_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = 0u;
    for (int i0 = 0; i0 < R; i0 += NGP_GRAM8_RC) {
        const int rc = min(NGP_GRAM8_RC, R - i0);  // multiple of 16
        __syncthreads();
        for (int idx = tid; idx < NGP_BLK * (rc >> 2); idx += 256) {  // words of 4 rows: unit-major source, lanes run over the columns
            const int j = idx & (NGP_BLK - 1), q4 = idx >> 6;          // q4: group of 4 rows inside the chunk
            const int i = i0 + 4 * q4;
            *(uint32_t *)(tt + j * NGP_GRAM8_LD + 4 * q4) = *(const uint32_t *)(src_t + tile8_off(i, j));
            *(uint32_t *)(ta + j * NGP_GRAM8_LD + 4 * q4) = *(const uint32_t *)(src_a + tile8_off(i, j));
        }
        __syncthreads();
        for (int i = 0; i < rc; i += 4) {
            uint32_t xk[4], xj[4];
#pragma unroll
            for (int a = 0; a < 4; a++) {
                xk[a] = *(const uint32_t *)(ta + (4 * tk + a) * NGP_GRAM8_LD + i);
                xj[a] = *(const uint32_t *)(tt + (4 * tj + a) * NGP_GRAM8_LD + i);
            }
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) acc[a][b] = __builtin_amdgcn_udot4(xk[a], xj[b], acc[a][b], false);
        }
    }
    uint32_t *out = gpart + ((size_t)tb * S + s) * (NGP_BLK * NGP_BLK);
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) out[(4 * tk + a) * NGP_BLK + 4 * tj + b] = acc[a][b];
}
// sum over shards (exact, 64-bit) -> G = dot - N (m_k m_j) -> gramx[t][d][k][j] in the layout of k_gram_reduce; d == 0 also fills mpm
__global__ __launch_bounds__(256) void k_gram8_reduce(const uint32_t *__restrict__ gpart, double *__restrict__ gramx, double *__restrict__ mpm,
                                                      const double *__restrict__ mean, long long N, int S, int t0, int nb, int d, int D) {
    long long e = (long long)blockIdx.x * 256 + threadIdx.x;  // (tb, k, j)
    if (e >= (long long)nb * NGP_BLK * NGP_BLK) return;
    int tb = (int)(e / (NGP_BLK * NGP_BLK)), kj = (int)(e % (NGP_BLK * NGP_BLK));
    if (t0 + tb - d < 0) return;
    const uint32_t *p = gpart + (size_t)tb * S * (NGP_BLK * NGP_BLK) + kj;
    unsigned long long dot = 0;
    for (int s = 0; s < S; s++) dot += p[(size_t)s * (NGP_BLK * NGP_BLK)];
    const int k = kj / NGP_BLK, j = kj % NGP_BLK;
    const double mm = mean[(size_t)(t0 + tb - d) * NGP_BLK + k] * mean[(size_t)(t0 + tb) * NGP_BLK + j];
    const double nm = (double)N * mm;
    const double tot = (double)dot - nm;
    if (d == 0 && k == j) mpm[(size_t)(t0 + tb) * NGP_BLK + k] = tot;
    const size_t off = (d == 0) ? (size_t)kj : ((size_t)(k >> 1) * NGP_BLK + j) * 2 + (k & 1);
    gramx[((size_t)(t0 + tb) * D + d) * (NGP_BLK * NGP_BLK) + off] = (d == 0 && j <= k) ? 0.0 : tot;
}

// out_i = sum_k (g_ik - m_k) beta_k = sum_k g_ik beta_k - sum_k m_k beta_k, k ascending (utility, not on the hot path)
__global__ __launch_bounds__(256) void k_xbeta8(const uint8_t *__restrict__ tiles8, const double *__restrict__ mean,
                                                const double *__restrict__ beta, double *__restrict__ out, int R, int S, long long NBLK,
                                                long long N) {
    const int s = blockIdx.x;
    for (int i = threadIdx.x; i < R; i += 256) {
        double acc = 0.0, accm = 0.0;
        for (long long t = 0; t < NBLK; t++) {
            const uint8_t *tp = tiles8 + ((size_t)t * S + s) * ((size_t)R * NGP_BLK);
            for (int j = 0; j < NGP_BLK; j++) {
                acc = __builtin_fma((double)tp[tile8_off(i, j)], beta[t * NGP_BLK + j], acc);
                accm = __builtin_fma(mean[t * NGP_BLK + j], beta[t * NGP_BLK + j], accm);
            }
        }
        out[(size_t)s * R + i] = ((long long)s * R + i < N) ? acc - accm : 0.0;
    }
}

// out_i = sum_k x_ik beta_k, k ascending (utility, not on the hot path)
__global__ __launch_bounds__(256) void k_xbeta(const float *__restrict__ tiles, const double *__restrict__ beta,
                                               double *__restrict__ out, int R, int S, long long NBLK) {
    const int s = blockIdx.x;
    for (int i = threadIdx.x; i < R; i += 256) {
        double acc = 0.0;
        for (long long t = 0; t < NBLK; t++) {
            const float *tp = tiles + ((size_t)t * S + s) * ((size_t)R * NGP_BLK);
            for (int j = 0; j < NGP_BLK; j++) acc = __builtin_fma((double)tp[tile_off(i, j)], beta[t * NGP_BLK + j], acc);
        }
        out[(size_t)s * R + i] = acc;
    }
}

// ------------------------------------------------------------------------------------------
// probes
// ------------------------------------------------------------------------------------------
__global__ void k_draws_indexed(uint64_t seed, uint64_t chain, uint64_t it, uint64_t kind, uint64_t index0, int what, double p1,
                                double p2, long long n, double *__restrict__ out) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Rng r = rng_seed(seed, chain, it, kind, index0 + (uint64_t)i);
    double v;
    switch (what) {
        case 0: v = rng_uniform(r); break;
        case 1: v = rng_normal(r); break;
        case 2: v = rng_chisq(r, p1); break;
        case 3: v = rng_beta(r, p1, p2); break;
        default: v = rng_gamma(r, p1); break;
    }
    out[i] = v;
}
__global__ void k_eval_math(int which, const double *__restrict__ in, long long n, double *__restrict__ out) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double x = in[i];
    double v;
    switch (which) {
        case 0: v = det_log(x); break;
        case 1: v = ppnd16(x); break;
        case 2: v = det_sqrt(x); break;
        default: v = 1.0 / x; break;
    }
    out[i] = v;
}

}  // namespace ngp
