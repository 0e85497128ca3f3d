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
#define NGP_RMAX 4  // variance classes of a BayesR set (the chain keeps their coefficients in registers)

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

struct DReg {  // one BayesPR variance region
    long long seg0;
    int nseg, set, rg, vb;
    long long n;
};

struct DScal {  // chain scalars
    double varE, iVarE, b, db;
    double sum_varE, sum_b;
    long long nKept;
};

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
    double q[NGP_RMAX], a[NGP_RMAX], t[NGP_RMAX], u[NGP_RMAX];
    double rhs0;
    int K;
};
__device__ inline RLane load_rlane(const double *__restrict__ rcls, long long Ppad, long long k, int K, const double *__restrict__ rhs0) {
    RLane L;
    L.K = K;
    L.rhs0 = rhs0[k];
#pragma unroll
    for (int v = 0; v < NGP_RMAX; v++) {
        const bool on = v < K;
        const size_t o = (size_t)(on ? v : 0) * (size_t)Ppad + (size_t)k;
        L.q[v] = on ? rcls[o] : 0.0;
        L.a[v] = on ? rcls[(size_t)NGP_RMAX * Ppad + o] : 0.0;
        L.t[v] = on ? rcls[(size_t)2 * NGP_RMAX * Ppad + o] : 0.0;
        L.u[v] = on ? rcls[(size_t)3 * NGP_RMAX * Ppad + o] : 0.0;
    }
    return L;
}
__device__ inline void eval_rform(const int meth, const double r, const double bo, const double cc, const double ww, const double st,
                                  const RLane &L, const double iVarE, double &cand, int &cls) {
    if (meth == 3) {
        const double t = r * iVarE;
        const double rhs = t + L.rhs0;
        const double s2 = rhs * rhs;
        const double hs = 0.5 * s2;
        double Lv[NGP_RMAX], e[NGP_RMAX];
#pragma unroll
        for (int v = 0; v < NGP_RMAX; v++) Lv[v] = (L.q[v] == 0.0) ? L.a[v] : __builtin_fma(hs, L.q[v], L.a[v]);
        double m = Lv[0];
#pragma unroll
        for (int v = 1; v < NGP_RMAX; v++)
            if (v < L.K && Lv[v] > m) m = Lv[v];
        double S = 0.0;
#pragma unroll
        for (int v = 0; v < NGP_RMAX; v++)
            if (v < L.K) {
                e[v] = det_exp(Lv[v] - m);
                S = S + e[v];
            } else e[v] = 0.0;
        int c = L.K - 1;
        double cum = 0.0;
        bool found = false;
#pragma unroll
        for (int v = 0; v < NGP_RMAX; v++)
            if (v < L.K && !found) {
                cum = cum + e[v];
                const double thr = L.u[v] * S;
                if (cum >= thr) { c = v; found = true; }
            }
        double qc = L.q[0], tc = L.t[0];
#pragma unroll
        for (int v = 1; v < NGP_RMAX; v++)
            if (c == v) { qc = L.q[v]; tc = L.t[v]; }
        if (qc != 0.0) {
            const double d = __builtin_fma(rhs, qc, tc);
            cand = d - bo;
        } else cand = -bo;
        cls = c + 1;
    } else {
        const double f = r * cc;
        const int in = __builtin_fabs(f) > st;
        const double e1 = __builtin_fma(r, cc, ww);
        cand = in ? e1 : -bo;
        cls = in;
    }
}

// Tile (t, s) = R rows x 64 columns of fp32, stored QUAD-MAJOR: element (row i, column j) sits at (i >> 2) * 256 + j * 4 + (i & 3).
__host__ __device__ inline size_t tile_off(int i, int j) { return ((size_t)(i >> 2) << 8) + (size_t)(j << 2) + (size_t)(i & 3); }
// compact storage: byte offset of element (row i, column j) inside a tile: units of 16 rows, a column's 16 bytes contiguous
__host__ __device__ inline size_t tile8_off(int i, int j) { return ((size_t)(i >> 4) << 10) + (size_t)(j << 4) + (size_t)(i & 15); }

}  // namespace ngp
