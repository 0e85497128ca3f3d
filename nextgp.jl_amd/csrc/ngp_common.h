// ngp_common.h -- structures and device helpers shared by the kernels of ngp_kernels.h and the persistent sweep of ngp_sweep.h
// (split off so that the sweep kernel can be compiled in translation units of its own).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ngp_rng.h"

#pragma clang fp contract(off)

#define NGP_BLK 64
#define NGP_SEG 256
#define NGP_GRP 32
#define NGP_RMAX 16  // variance classes of a BayesR set (src/functions.jl:241-262 sizes everything by length(vClass))
#define NGP_RLDS 8   // ... of which the sampler stages the coefficients of the first eight in LDS / register arrays; classes 9..16 come from memory each time
#define NGP_RREG 4  // ... of which the block chain keeps the coefficients of the first four in registers; further classes are read
                    // from the coefficient arrays of k_prep each time a candidate is formed
#define NGP_KMAX 4  // marker sets of one tuple (correlated BayesPR, src/functions.jl:140-154)
#define NGP_METHOD_TUPLE_DEV 4

namespace ngp {

struct DSet {  // one marker set (src/mme.jl:324-361, 492-520)
    int method, estPi;
    double df, scale, sdf;  // sdf = scale*df
    long long col0, ncol;
    double piHat0, piHat1, logPi0, logPi1;
    int nloci;  // included loci of the running BayesB sweep
    int pad_;
    double sum_pi0, sum_pi1;
    // BayesR (src/functions.jl:238-289, set-up src/mme.jl:374-383): K classes, multipliers of the set's single variance
    int K, pad2_;
    double vcls[NGP_RMAX], pic[NGP_RMAX], logpic[NGP_RMAX], sum_pic[NGP_RMAX];
    int ncls[NGP_RMAX];  // loci per class of the running sweep
};

// Correlated (Tuple) BayesPR set (src/functions.jl:140-154, 513-516; set-up src/mme.jl:448-489): k sets share nloc loci, the k
// columns of a locus adjacent in the panel; the set's marker-set entry (DSet) carries method 4, this the rest.
struct DTup {
    int k, pad_;
    long long col0, nloc, vb_off;  // first panel column (a block boundary), loci, offset of the nreg k x k variance matrices in varBeta
    double df, scale[NGP_KMAX * NGP_KMAX];
};
// panel column of component m of tuple locus l: floor(64 / k) whole loci per 64-column block, component-minor
__host__ __device__ inline long long tuple_col(long long col0, int k, long long l, int m) {
    const long long Lb = NGP_BLK / k;
    return col0 + (long long)NGP_BLK * (l / Lb) + (long long)k * (l % Lb) + m;
}
// ---- k x k helpers of the Tuple path (row-major, k <= NGP_KMAX), operation for operation the text of oracle/ngp_oracle.c
// (t_chol, t_spd_inv, t_chol1): fma where written, everything else separately rounded; k = 1 takes the scalar forms ----
__device__ inline int t_chol(const double *S, int k, double *L) {
    for (int a = 0; a < k * k; a++) L[a] = 0.0;
    for (int i = 0; i < k; i++)
        for (int j = 0; j <= i; j++) {
            double s = S[i * k + j];
            for (int m = 0; m < j; m++) s = __builtin_fma(-L[i * k + m], L[j * k + m], s);
            if (i == j) { if (!(s > 0.0)) return -1; L[i * k + i] = det_sqrt(s); }
            else L[i * k + j] = s / L[j * k + j];
        }
    return 0;
}
__device__ inline int t_spd_inv(const double *S, int k, double *out) {
    if (k == 1) { out[0] = 1.0 / S[0]; return (S[0] > 0.0) ? 0 : -1; }
    double L[NGP_KMAX * NGP_KMAX], Li[NGP_KMAX * NGP_KMAX];
    if (t_chol(S, k, L)) return -1;
    for (int a = 0; a < k * k; a++) Li[a] = 0.0;
    for (int c = 0; c < k; c++)
        for (int i = c; i < k; i++) {
            double s = (i == c) ? 1.0 : 0.0;
            for (int m = c; m < i; m++) s = __builtin_fma(-L[i * k + m], Li[m * k + c], s);
            Li[i * k + c] = s / L[i * k + i];
        }
    for (int i = 0; i < k; i++)
        for (int j = 0; j < k; j++) {
            double s = 0.0;
            for (int m = 0; m < k; m++) s = __builtin_fma(Li[m * k + i], Li[m * k + j], s);
            out[i * k + j] = s;
        }
    return 0;
}
__device__ inline int t_chol1(const double *S, int k, double *L) {
    if (k == 1) { L[0] = det_sqrt(S[0]); return (S[0] >= 0.0) ? 0 : -1; }
    return t_chol(S, k, L);
}

struct DReg {  // one BayesPR variance region
    long long seg0;
    int nseg, set, rg, vb;
    long long n;
};

struct DScal {  // chain scalars
    double varE, iVarE, b, db;
    double sum_varE, sum_b;
    long long nKept;
    double fx_scale, fx_inv;  // fixed-point scale of this iteration's dot-product accumulators and its reciprocal (powers of two, k_head)
};

// ------------------------------------------------------------------------------------------
// X_t'ycorr over the shards as an ORDER-FREE sum.  Every shard partial p_s (and every far look-ahead
// term -v_d) is scaled by a power of two, rounded to the nearest integer and added -- by a 64-bit integer atomic -- into the block's
// accumulator: integer addition is associative, so the total does not depend on the order the terms arrive in, and the hand-off
// needs no reducer stage (streamer -> reducer -> sampler was two cross-CU hops of the sweep's latency loop).  The low
// NGP_FX_CNT_BITS bits of every term carry a 1: the accumulator counts its own terms, the reader knows when the sum is complete.
//   term = (rn(p * scale) << CNT) + 1;   total = fx_to_f64(sum >> CNT) * (1 / scale);   complete: (sum & (2^CNT - 1)) == terms
// scale = 2^(52 - E), 2^E > 16 sqrt(max_j x_j'x_j * ycorr'ycorr) at the head of the iteration (Cauchy-Schwarz bounds sum_s |p_s|
// by the square root, so the sum stays below 2^48 of the 2^53 it may reach: 32 x head room for what ycorr does during the sweep);
// a term beyond 2^53 stops the sweep (abort 7).
// ------------------------------------------------------------------------------------------
#define NGP_FX_CNT_BITS 10  // up to 1023 terms per block (699 shards of the tall layouts + far lags)
#ifndef NGP_FX_COPIES
#define NGP_FX_COPIES 8     // accumulator copies per block (shard s adds to copy s mod 8: eight memory lines share the atomics)
#endif
// exact conversions (|x| <= 2^53: x is rounded to an integer first, then split into halves the 32-bit converts carry exactly)
__host__ __device__ inline long long fx_from_f64(const double x) {
    const double r = __builtin_rint(x);  // round to nearest, ties to even
    const double hi = __builtin_floor(r * 0x1p-32);
    const double lo = __builtin_fma(-hi, 0x1p32, r);
    return (long long)(((unsigned long long)(unsigned)(int)hi << 32) | (unsigned long long)(unsigned)lo);
}
__host__ __device__ inline double fx_to_f64(const long long q) {  // = (double)q, one rounding
    const double hi = (double)(int)(q >> 32), lo = (double)(unsigned)q;
    return __builtin_fma(hi, 0x1p32, lo);
}
__host__ __device__ inline unsigned long long fx_term(const double p, const double scale) {
    return ((unsigned long long)fx_from_f64(p * scale) << NGP_FX_CNT_BITS) + 1ull;
}
// scale exponent from m = max x'x * ycorr'ycorr: E = floor(ilogb(m) / 2) + 5 (2^E > 16 sqrt(m)), clamped; scale = 2^(52 - E)
__host__ __device__ inline int fx_exponent(const double m) {
    union { double d; unsigned long long u; } v;
    v.d = m;
    const int ef = (int)((v.u >> 52) & 0x7ffull);
    int e2 = 0;
    if (m > 0.0 && ef != 0x7ff) e2 = ef - 1023;
    int e = (e2 >> 1) + 5;
    if (e < -900) e = -900;
    if (e > 900) e = 900;
    return e;
}
__host__ __device__ inline double fx_pow2(const int e) {  // 2^e, -1022 <= e <= 1023
    union { double d; unsigned long long u; } v;
    v.u = (unsigned long long)(1023 + e) << 52;
    return v.d;
}

__device__ inline double readlane_d(double v, int lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}

// ------------------------------------------------------------------------------------------
// "r-form" of the block chain (blocks that hold a BayesR locus; DESIGN.md section 2, step 5'): every lane forms its candidate
// dlt from the current r = x'(ycorr + x beta); the first lane at or behind the cursor with a non-zero candidate takes its
// step, the candidates behind it are formed again.  One lane's rule:
//   BayesR   class search of src/functions.jl:250-261 in a stable form (L_v - max L through det_exp; the v-th comparison
//            cum_v >= u_v * sum with its own uniform u_v), then dlt = rhs / lhs_c + sd_c z - beta, or -beta in a zero class
//   others   in = |r c| > thr, dlt = in ? r c + w : -beta   (BayesPR: always in)
// ------------------------------------------------------------------------------------------
struct RLane {  // class coefficients of one BayesR locus (k_prep): 1/lhs_v, log-weight a_v, sd_v z, uniform u_v; M.rhs
    double q[NGP_RREG], a[NGP_RREG], t[NGP_RREG], u[NGP_RREG];
    double rhs0;
    int K;
    // K > NGP_RREG: the classes from the fifth on, array arr (q, a, t, u) of class v at ext[v * Ppad + arr * astride] -- in memory
    // rcls + locus with astride = NGP_RMAX * Ppad, or the sampler's LDS copy (Ppad = 64, astride = 4 x 64, base moved back by 4 classes)
    const double *ext;
    long long Ppad, astride;
};
__device__ inline RLane load_rlane(const double *__restrict__ rcls, long long Ppad, long long k, int K, const double *__restrict__ rhs0) {
    RLane L;
    L.K = K;
    L.rhs0 = rhs0[k];
    L.ext = rcls + k;
    L.Ppad = Ppad;
    L.astride = (long long)NGP_RMAX * Ppad;
#pragma unroll
    for (int v = 0; v < NGP_RREG; v++) {
        const bool on = v < K;
        const size_t o = (size_t)(on ? v : 0) * (size_t)Ppad + (size_t)k;
        // (a register class the set does not have: log-weight -inf, so its exponential is 0 and no maximum; comparison uniform +inf,
        // so the search never stops at it -- eval_rform then needs no "does this class exist" select anywhere)
        L.q[v] = on ? rcls[o] : 0.0;
        L.a[v] = on ? rcls[(size_t)NGP_RMAX * Ppad + o] : -__builtin_inf();
        L.t[v] = on ? rcls[(size_t)2 * NGP_RMAX * Ppad + o] : 0.0;
        L.u[v] = on ? rcls[(size_t)3 * NGP_RMAX * Ppad + o] : __builtin_inf();
    }
    return L;
}
__device__ inline RLane empty_rlane() {
    RLane L;
    L.K = 2; L.rhs0 = 0.0; L.ext = nullptr; L.Ppad = 0; L.astride = 0;
#pragma unroll
    for (int v = 0; v < NGP_RREG; v++) { L.q[v] = 0.0; L.a[v] = (v < 2) ? 0.0 : -__builtin_inf(); L.t[v] = 0.0; L.u[v] = (v < 2) ? 0.0 : __builtin_inf(); }
    return L;
}
// log-weight of class v given hs = rhs^2 / 2 (src/functions.jl:255 in the stable form): classes >= NGP_RREG come from memory
__device__ inline double rlane_L(const RLane &L, const int v, const double hs) {
    const double q = L.ext[(size_t)v * L.Ppad], a = L.ext[(size_t)v * L.Ppad + (size_t)L.astride];
    return (q == 0.0) ? a : __builtin_fma(hs, q, a);
}
// candidate of a lane of another method inside an r-form block (the rule of step 5 in r-form)
__device__ inline void eval_rform_other(const double r, const double bo, const double cc, const double ww, const double st, double &cand, int &cls) {
    const double f = r * cc;
    const int in = __builtin_fabs(f) > st;
    const double e1 = __builtin_fma(r, cc, ww);
    cand = in ? e1 : -bo;
    cls = in;
}
__device__ inline void eval_rform(const int meth, const double r, const double bo, const double cc, const double ww, const double st,
                                  const RLane &L, const double iVarE, double &cand, int &cls) {
    if (meth == 3) {
        const double t = r * iVarE;
        const double rhs = t + L.rhs0;
        const double s2 = rhs * rhs;
        const double hs = 0.5 * s2;
        double Lv[NGP_RREG], e[NGP_RREG];
#pragma unroll
        for (int v = 0; v < NGP_RREG; v++) Lv[v] = (L.q[v] == 0.0) ? L.a[v] : __builtin_fma(hs, L.q[v], L.a[v]);
        // The four register classes are handled without a branch (selects): their four det_exp -- forty dependent fp64 operations
        // each -- then run interleaved instead of one after the other behind "does this class exist" tests; the values are the same.
        const bool more = L.K > NGP_RREG;
        double m = Lv[0];
#pragma unroll
        for (int v = 1; v < NGP_RREG; v++) m = (Lv[v] > m) ? Lv[v] : m;  // (classes the set does not have: -inf, load_rlane)
        // (more than four classes: the same steps, class after class; their log-weights come from memory ONCE per evaluation)
        double Lx[NGP_RLDS - NGP_RREG], ex[NGP_RLDS - NGP_RREG];
#pragma unroll
        for (int v = NGP_RREG; v < NGP_RLDS; v++) { Lx[v - NGP_RREG] = 0.0; ex[v - NGP_RREG] = 0.0; }
        if (more) {
#pragma unroll
            for (int v = NGP_RREG; v < NGP_RLDS; v++) Lx[v - NGP_RREG] = (v < L.K) ? rlane_L(L, v, hs) : 0.0;
#pragma unroll
            for (int v = NGP_RREG; v < NGP_RLDS; v++)
                if (v < L.K && Lx[v - NGP_RREG] > m) m = Lx[v - NGP_RREG];
            // classes 9..16 (rare): nothing kept, their log-weights are formed again wherever they are needed -- the same operations, the same bits
            for (int v = NGP_RLDS; v < L.K; v++) {
                const double Lw = rlane_L(L, v, hs);
                if (Lw > m) m = Lw;
            }
        }
        double S = 0.0;
        {
            static_assert(NGP_RREG == 4, "det_exp4 serves the four register classes");
            double xm[4], ev[4];
#pragma unroll
            for (int v = 0; v < NGP_RREG; v++) xm[v] = Lv[v] - m;
            det_exp4(xm, ev);
#pragma unroll
            for (int v = 0; v < NGP_RREG; v++) e[v] = ev[v];  // (0 for a class the set does not have: S + 0 = S, cum + 0 = cum)
        }
#pragma unroll
        for (int v = 0; v < NGP_RREG; v++) S = S + e[v];
        if (more) {
            static_assert(NGP_RLDS - NGP_RREG == 4, "det_exp4 serves the four memory classes");
            double xm[4], ev[4];
#pragma unroll
            for (int v = 0; v < 4; v++) xm[v] = (NGP_RREG + v < L.K) ? Lx[v] - m : 0.0;
            det_exp4(xm, ev);
#pragma unroll
            for (int v = NGP_RREG; v < NGP_RLDS; v++)
                if (v < L.K) {
                    ex[v - NGP_RREG] = ev[v - NGP_RREG];
                    S = S + ex[v - NGP_RREG];
                }
            for (int v = NGP_RLDS; v < L.K; v++) S = S + det_exp(rlane_L(L, v, hs) - m);
        }
        int c = L.K - 1;
        double cum = 0.0;
        bool found = false;
#pragma unroll
        for (int v = 0; v < NGP_RREG; v++) {
            const bool take = !found;  // (a class the set does not have: thr = inf, never a hit)
            const double cn = cum + e[v];
            const double thr = L.u[v] * S;
            const bool hit = take && (cn >= thr);
            cum = take ? cn : cum;
            c = hit ? v : c;
            found = found || hit;
        }
        if (more) {
#pragma unroll
            for (int v = NGP_RREG; v < NGP_RLDS; v++)
                if (v < L.K && !found) {
                    cum = cum + ex[v - NGP_RREG];
                    const double thr = L.ext[(size_t)v * L.Ppad + (size_t)3 * L.astride] * S;
                    if (cum >= thr) { c = v; found = true; }
                }
            for (int v = NGP_RLDS; v < L.K && !found; v++) {
                cum = cum + det_exp(rlane_L(L, v, hs) - m);
                const double thr = L.ext[(size_t)v * L.Ppad + (size_t)3 * L.astride] * S;
                if (cum >= thr) { c = v; found = true; }
            }
        }
        double qc = L.q[0], tc = L.t[0];
#pragma unroll
        for (int v = 1; v < NGP_RREG; v++) {
            qc = (c == v) ? L.q[v] : qc;
            tc = (c == v) ? L.t[v] : tc;
        }
        // (the register classes are chosen by selects, and stay so: joined with the memory classes below the compiler made ONE load
        // through a chosen pointer of it, with L.t spilled to scratch to have an address -- a flat load per evaluation)
        asm volatile("" : "+v"(qc), "+v"(tc));
        if (more && c >= NGP_RREG) { qc = L.ext[(size_t)c * L.Ppad]; tc = L.ext[(size_t)c * L.Ppad + (size_t)2 * L.astride]; }
        if (qc != 0.0) {
            const double d = __builtin_fma(rhs, qc, tc);
            cand = d - bo;
        } else cand = -bo;
        cls = c + 1;
    } else {
        eval_rform_other(r, bo, cc, ww, st, cand, cls);
    }
}

// ------------------------------------------------------------------------------------------
// Block chain of a Tuple (correlated BayesPR) block (src/functions.jl:144-151 in 64-column space).
// Lane j = column j of the block; the k columns of a locus are adjacent, floor(64 / k) loci per block.  tot = x_j'ycorr as the
// look-ahead pipeline delivers it (group sums minus corrections).  One step per LOCUS: its k effects are drawn together from
// the r of its k columns, dlt_m = W_m + sum_b C[m][b] r_b (W = L z - beta, C = iVarE inv(LHS): k_prep), then applied component
// after component to the columns of the later loci, r_j -= G[m][j] dlt_m.  G(kk) returns G[kk][j] of the one-sided diagonal block
// (0 for j <= kk).  Returns dlt of this lane's column (0 for the unused lanes of the block).
// ------------------------------------------------------------------------------------------
struct TupLane {  // per-column coefficients of a tuple lane (prefetched): rows of C and of X_l'X_l, W
    double crow[NGP_KMAX], grow[NGP_KMAX], ww;
};
__device__ inline TupLane load_tuplane(const double *__restrict__ tupc, const double *__restrict__ tupg, const double *__restrict__ w,
                                       long long Ppad, long long kcol) {
    TupLane T;
#pragma unroll
    for (int b = 0; b < NGP_KMAX; b++) {
        T.crow[b] = tupc[(size_t)b * Ppad + kcol];
        T.grow[b] = tupg[(size_t)b * Ppad + kcol];
    }
    T.ww = w[kcol];
    return T;
}
// loci per 64-column block and first column of lane j's locus, without a run-time division (k = 1..NGP_KMAX)
__device__ inline int tuple_loci_per_block(const int k) { return k == 1 ? 64 : (k == 2 ? 32 : (k == 3 ? 21 : 16)); }
__device__ inline int tuple_gbase(const int k, const int j) { return k == 1 ? j : (k == 2 ? (j & ~1) : (k == 3 ? (j / 3) * 3 : (j & ~3))); }
// used lanes of the block that begins at locus first_locus (wave-uniform)
__device__ inline int tuple_nvalid(const int k, const long long nloc, const long long first_locus) {
    const int Lb = tuple_loci_per_block(k);
    const long long left = nloc - first_locus;
    return (int)(left < Lb ? left : Lb) * k;
}
// K = k as a compile-time constant: with a run-time k every "component b exists" test is a branch, every load sits in a basic block
// of its own behind a full wait (64 LDS round trips in a row per sixteen steps: 7 us per block); here the loads of sixteen steps
// leave together
// e0 of a Tuple block's lane: its candidate dlt before any column of the block has been drawn
template <int K>
__device__ __attribute__((always_inline)) inline double tuple_e0_k(const int nvalid, const int j, const double tot, const double bo, const TupLane &T) {
    const int gbase = tuple_gbase(K, j);
    const bool valid = j < nvalid;
    // x_m'(ycorr + X_l beta_l): the add-back of all k effects of the locus, components in order
    double rfull = tot;
    double bm[K];
#pragma unroll
    for (int b = 0; b < K; b++) bm[b] = __shfl(bo, gbase + b);
#pragma unroll
    for (int b = 0; b < K; b++)
        if (valid) rfull = __builtin_fma(T.grow[b], bm[b], rfull);
    // scaled form, as the Symbol path's chain: e = W + sum_b C[b] r_b is this lane's candidate dlt
    double e = T.ww;
    double rb[K];
#pragma unroll
    for (int b = 0; b < K; b++) rb[b] = __shfl(rfull, gbase + b);
#pragma unroll
    for (int b = 0; b < K; b++) e = __builtin_fma(rb[b], T.crow[b], e);
    if (!valid) e = 0.0;
    return e;
}
__device__ __attribute__((always_inline)) inline double tuple_e0(const int k, const int nvalid, const int j, const double tot, const double bo, const TupLane &T) {
    switch (k) {
        case 1: return tuple_e0_k<1>(nvalid, j, tot, bo, T);
        case 2: return tuple_e0_k<2>(nvalid, j, tot, bo, T);
        case 3: return tuple_e0_k<3>(nvalid, j, tot, bo, T);
        default: return tuple_e0_k<4>(nvalid, j, tot, bo, T);
    }
}
template <int K, typename GAt>
__device__ __attribute__((always_inline)) inline double tuple_chain_k(const int nvalid, const int j, const double tot, const double bo, const TupLane &T, GAt G) {
    const int gbase = tuple_gbase(K, j);
    const bool valid = j < nvalid;
    double e = tuple_e0_k<K>(nvalid, j, tot, bo, T);
    // a finished column s changes e by H(s) dlt_s, H(s) = -(sum_b C[b] G[s][column b of this lane's locus]), for the columns of
    // LATER loci only (the k effects of a locus are drawn together).  The H of sixteen steps are formed first -- independent loads
    // and fma, in flight together -- so that the serial part of a step is what it is on the Symbol path: one readlane, one fma.
    // Steps beyond the block's used lanes have H = 0 and dlt = 0.
    for (int s0 = 0; s0 < nvalid; s0 += 16) {
        double g[16][K];
#pragma unroll
        for (int i = 0; i < 16; i++)
#pragma unroll
            for (int b = 0; b < K; b++) g[i][b] = G(s0 + i, gbase + b);
        double H[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            double t = T.crow[0] * g[i][0];
#pragma unroll
            for (int b = 1; b < K; b++) t = __builtin_fma(T.crow[b], g[i][b], t);
            H[i] = (valid && s0 + i < gbase) ? -t : 0.0;
        }
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const double dk = readlane_d(e, s0 + i);
            e = __builtin_fma(H[i], dk, e);
        }
    }
    return e;
}
template <typename GAt>
__device__ __attribute__((always_inline)) inline double tuple_chain_nv(const int k, const int nvalid, const int j, const double tot, const double bo, const TupLane &T, GAt G) {
    switch (k) {
        case 1: return tuple_chain_k<1>(nvalid, j, tot, bo, T, G);
        case 2: return tuple_chain_k<2>(nvalid, j, tot, bo, T, G);
        case 3: return tuple_chain_k<3>(nvalid, j, tot, bo, T, G);
        default: return tuple_chain_k<4>(nvalid, j, tot, bo, T, G);
    }
}
template <typename GAt>
__device__ __attribute__((always_inline)) inline double tuple_chain(const int k, const long long nloc, const long long first_locus, const int j, const double tot,
                                     const double bo, const TupLane &T, GAt G) {
    return tuple_chain_nv(k, tuple_nvalid(k, nloc, first_locus), j, tot, bo, T, G);
}

// Tile (t, s) = R rows x 64 columns of fp32, stored QUAD-MAJOR: element (row i, column j) sits at (i >> 2) * 256 + j * 4 + (i & 3).
__host__ __device__ inline size_t tile_off(int i, int j) { return ((size_t)(i >> 2) << 8) + (size_t)(j << 2) + (size_t)(i & 3); }
// compact storage: byte offset of element (row i, column j) inside a tile: units of 16 rows, a column's 16 bytes contiguous
__host__ __device__ inline size_t tile8_off(int i, int j) { return ((size_t)(i >> 4) << 10) + (size_t)(j << 4) + (size_t)(i & 15); }

}  // namespace ngp
