/*
 * ngp_oracle.c -- CPU oracle for the NextGP.jl marker-effect Gibbs hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this file's shared object.  The product
 * (nextgp.jl_amd/csrc -> libnextgp_hip.so) never links, loads or calls it.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures
 * for this path (/root/reference/test/runtests.jl:1-7 is an empty testset) and
 * no Julia toolchain exists in the build container, so the reference itself
 * cannot be executed.  This oracle is a plain-C restatement of the reference
 * algorithm, pinned only by closed-form / statistical checks (tests/) and by
 * its own committed golden vectors (tests/golden/).
 *
 * Third-party arithmetic the reference delegates to packages that are absent
 * from /root/reference (Distributions.jl compat 0.25.58, Julia stdlib Random,
 * OpenBLAS level-1; Project.toml:7-36) is replaced by an explicit, portable
 * specification (docs in DESIGN.md "RNG and draw spec"):
 *   - xoshiro256++ streams keyed by (seed, chain, iteration, kind, index)
 *   - Normal: inverse CDF, Wichura AS241 PPND16
 *   - Gamma/Chi-square/Beta: Marsaglia-Tsang (2000) on the same stream
 *   - log: a fixed sequence of IEEE-754 double operations (det_log) so that
 *     CPU and GPU produce identical bits.
 *
 * Two orderings of the same Markov chain are provided:
 *   order 0  "reference order": per SNP add-back / dot / draw / subtract with
 *            a second transposed copy of the panel, exactly the memory and
 *            arithmetic pattern of
 *              src/samplers.jl:23-106   (iteration driver)
 *              src/functions.jl:39-53   (intercept, sampleX!)
 *              src/functions.jl:118-137 (sampleBayesPR!, Symbol method)
 *              src/functions.jl:157-195 (sampleBayesB!)
 *              src/functions.jl:493-495, 509-511, 523-525, 531-533 (draws)
 *              src/mme.jl:57,87-94,294-361,443-444,493-516 (set-up)
 *            This is the spec oracle and the CPU timing baseline.
 *   order 1  "blocked order": the 64-SNP block recursion with Gram
 *            corrections and the exact reduction trees the HIP kernels use
 *            (DESIGN.md "Blocked sweep arithmetic"), for bit-parity tests.
 *
 * Build: see oracle/Makefile (gcc -O3 -mavx2 -mfma -ffp-contract=off).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ORA_OK 0
#define ORA_ERR (-1)

#define KIND_VARE_CHI2 1
#define KIND_FIXED_NORMAL 2
#define KIND_BETA_NORMAL 3
#define KIND_REGION_CHI2 4
#define KIND_B_UNIFORM 5
#define KIND_B_LOCUS_CHI2 6
#define KIND_PI_BETA 7
#define KIND_R_UNIFORM 8   /* BayesR: the fresh uniform of every comparison of the class search (functions.jl:261) */
#define KIND_R_DIRICHLET 9
/* fixed-effect columns beyond the intercept draw from KIND_FIXED_NORMAL with index ((set + 1) << 20) | column */ /* BayesR: gamma draws of the Dirichlet (functions.jl:536-538) */

#define METHOD_PR 0
#define METHOD_B 1
#define METHOD_C 2 /* BayesC: src/functions.jl:197-235 */
#define METHOD_R 3 /* BayesR: src/functions.jl:238-289 */
#define RMAX 16    /* variance classes of a BayesR set (functions.jl:241-262 sizes everything by length(vClass)) */
#define METHOD_T 4 /* correlated (Tuple) BayesPR: src/functions.jl:140-154, 513-516; set-up src/mme.jl:448-489 */
#define KMAX 4     /* marker sets of one tuple */
#define KIND_T_WISHART 11 /* Bartlett factor of a region's inverse-Wishart draw: index (set << 40) | (region << 8) | (i << 4) | j */

#define BLK 64
#define SEG 256
#define GRP 32

/* ------------------------------------------------------------------ */
/* RNG: xoshiro256++ keyed streams                                      */
/* ------------------------------------------------------------------ */
#define GOLD 0x9E3779B97F4A7C15ULL

static inline uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

typedef struct { uint64_t s[4]; } rng_t;

static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

static inline uint64_t rng_next(rng_t *r) {
    uint64_t *s = r->s;
    uint64_t res = rotl64(s[0] + s[3], 23) + s[0];
    uint64_t t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
    s[2] ^= t; s[3] = rotl64(s[3], 45);
    return res;
}

static inline uint64_t absorb(uint64_t h, uint64_t v) { return mix64(h ^ mix64(v + GOLD)); }

static void rng_seed(rng_t *r, uint64_t seed, uint64_t chain, uint64_t iter, uint64_t kind, uint64_t index) {
    uint64_t h = mix64(seed + GOLD);
    h = absorb(h, chain);
    h = absorb(h, iter);
    h = absorb(h, kind);
    h = absorb(h, index);
    for (int i = 0; i < 4; i++) r->s[i] = mix64(h + (uint64_t)(i + 1) * GOLD);
}

/* uniform on the open interval (0,1): (k + 0.5) * 2^-52, k = top 52 bits */
static inline double rng_uniform(rng_t *r) {
    uint64_t k = rng_next(r) >> 12;
    return ((double)k + 0.5) * 2.220446049250313080847263336181640625e-16;
}

/* ------------------------------------------------------------------ */
/* det_log: fixed IEEE sequence (argument reduction + degree-14 series) */
/* classic algorithm: x = 2^k (1+f), s = f/(2+f), log(1+f) = f - s(f-R) */
/* ------------------------------------------------------------------ */
static inline double det_log(double x) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                 Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                 Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    uint64_t bits; memcpy(&bits, &x, 8);
    if (x <= 0.0) return (x == 0.0) ? -INFINITY : NAN;
    if ((bits >> 52) == 0x7FF) return x; /* inf / nan */
    int k = 0;
    if ((bits >> 52) == 0) { x *= 18014398509481984.0; memcpy(&bits, &x, 8); k = -54; } /* subnormal */
    uint32_t hx = (uint32_t)(bits >> 32);
    k += (int)(hx >> 20) - 1023;
    hx &= 0x000fffffu;
    uint32_t i = (hx + 0x95f64u) & 0x100000u;
    uint64_t nb = ((uint64_t)(hx | (i ^ 0x3ff00000u)) << 32) | (bits & 0xffffffffULL);
    k += (int)(i >> 20);
    double xn; memcpy(&xn, &nb, 8);
    double f = xn - 1.0;
    double s = f / (2.0 + f);
    double dk = (double)k;
    double z = s * s;
    double w = z * z;
    double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    double R = t2 + t1;
    double hfsq = 0.5 * f * f;
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

/* det_exp for x <= 0: fixed IEEE sequence (fdlibm's argument reduction, then the Taylor series to degree 13 by Horner -- round 4:
   no division, whose v_div_scale / v_div_fmas pairs serialised the four exponentials of a BayesR class evaluation on the device),
   so that CPU and GPU agree bitwise.  Results below 2^-1021 are returned as 0; the blocked BayesR path only ever calls it on
   L - max(L) <= 0. */
static inline double det_exp(double x) {
    const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10, invln2 = 1.44269504088896338700e+00;
    if (x != x) return x;
    if (x < -708.0) return 0.0;
    if (x > 0.0) x = 0.0;
    const int k = (int)(invln2 * x - 0.5);
    const double t = (double)k;
    const double hi = x - t * ln2HI, lo = t * ln2LO;
    const double xr = hi - lo;   /* |xr| <= ln 2 / 2 */
    /* exp(xr) = 1 + xr + xr^2 q(xr), q = the Taylor series to degree 13 by Horner (fma): no division, no data-dependent branch */
    double q = 1.0 / 6227020800.0;
    q = __builtin_fma(q, xr, 1.0 / 479001600.0);
    q = __builtin_fma(q, xr, 1.0 / 39916800.0);
    q = __builtin_fma(q, xr, 1.0 / 3628800.0);
    q = __builtin_fma(q, xr, 1.0 / 362880.0);
    q = __builtin_fma(q, xr, 1.0 / 40320.0);
    q = __builtin_fma(q, xr, 1.0 / 5040.0);
    q = __builtin_fma(q, xr, 1.0 / 720.0);
    q = __builtin_fma(q, xr, 1.0 / 120.0);
    q = __builtin_fma(q, xr, 1.0 / 24.0);
    q = __builtin_fma(q, xr, 1.0 / 6.0);
    q = __builtin_fma(q, xr, 0.5);
    const double tt = xr * xr;
    const double y = 1.0 + __builtin_fma(tt, q, xr);
    uint64_t sb = (uint64_t)(k + 1023) << 52; /* 2^k, k >= -1021 */
    double sc; memcpy(&sc, &sb, 8);
    return y * sc;
}
double ora_det_exp(double x) { return det_exp(x); }

/* ------------------------------------------------------------------ */
/* Normal quantile, Wichura (1988) AS241 PPND16                         */
/* ------------------------------------------------------------------ */
static inline double ppnd16(double p) {
    double q = p - 0.5, r, val;
    if (fabs(q) <= 0.425) {
        r = 0.180625 - q * q;
        double num = (((((((2.5090809287301226727e3 * r + 3.3430575583588128105e4) * r + 6.7265770927008700853e4) * r
                          + 4.5921953931549871457e4) * r + 1.3731693765509461125e4) * r + 1.9715909503065514427e3) * r
                       + 1.3314166789178437745e2) * r + 3.3871328727963666080e0);
        double den = (((((((5.2264952788528545610e3 * r + 2.8729085735721942674e4) * r + 3.9307895800092710610e4) * r
                          + 2.1213794301586595867e4) * r + 5.3941960214247511077e3) * r + 6.8718700749205790830e2) * r
                       + 4.2313330701600911252e1) * r + 1.0);
        return q * num / den;
    }
    r = (q < 0.0) ? p : 1.0 - p;
    r = sqrt(-det_log(r));
    if (r <= 5.0) {
        r = r - 1.6;
        double num = (((((((7.74545014278341407640e-4 * r + 2.27238449892691845833e-2) * r + 2.41780725177450611770e-1) * r
                          + 1.27045825245236838258e0) * r + 3.64784832476320460504e0) * r + 5.76949722146069140550e0) * r
                       + 4.63033784615654529590e0) * r + 1.42343711074968357734e0);
        double den = (((((((1.05075007164441684324e-9 * r + 5.47593808499534494600e-4) * r + 1.51986665636164571966e-2) * r
                          + 1.48103976427480074590e-1) * r + 6.89767334985100004550e-1) * r + 1.67638483018380384940e0) * r
                       + 2.05319162663775882187e0) * r + 1.0);
        val = num / den;
    } else {
        r = r - 5.0;
        double num = (((((((2.01033439929228813265e-7 * r + 2.71155556874348757815e-5) * r + 1.24266094738807843860e-3) * r
                          + 2.65321895265761230930e-2) * r + 2.96560571828504891230e-1) * r + 1.78482653991729133580e0) * r
                       + 5.46378491116411436990e0) * r + 6.65790464350110377720e0);
        double den = (((((((2.04426310338993978564e-15 * r + 1.42151175831644588870e-7) * r + 1.84631831751005468180e-5) * r
                          + 7.86869131145613259100e-4) * r + 1.48753612908506148525e-2) * r + 1.36929880922735805310e-1) * r
                       + 5.99832206555887937690e-1) * r + 1.0);
        val = num / den;
    }
    return (q < 0.0) ? -val : val;
}

static inline double rng_normal(rng_t *r) { return ppnd16(rng_uniform(r)); }

/* Marsaglia & Tsang (2000), shape a >= 1, scale 1; no squeeze step */
static double rng_gamma(rng_t *r, double a) {
    double d = a - 1.0 / 3.0;
    double c = 1.0 / sqrt(9.0 * d);
    for (;;) {
        double x, v;
        do { x = rng_normal(r); v = 1.0 + c * x; } while (v <= 0.0);
        v = v * v * v;
        double u = rng_uniform(r);
        double x2 = x * x;
        double lv = det_log(v);
        double t = 1.0 - v;
        t = t + lv;
        t = d * t;
        double h = 0.5 * x2;
        double rhs = h + t;
        if (det_log(u) < rhs) return d * v;
    }
}
static inline double rng_chisq(rng_t *r, double nu) { return 2.0 * rng_gamma(r, 0.5 * nu); }
static inline double rng_beta(rng_t *r, double a, double b) {
    double ga = rng_gamma(r, a);
    double gb = rng_gamma(r, b);
    return ga / (ga + gb);
}

/* exported scalar probes so tests can pin the draw layer */
double ora_det_log(double x) { return det_log(x); }
double ora_ppnd16(double p) { return ppnd16(p); }
void ora_draws(uint64_t seed, uint64_t chain, uint64_t iter, uint64_t kind, uint64_t index, int what, double p1, double p2,
               int64_t n, double *out) {
    /* what: 0 uniform, 1 normal, 2 chisq(p1), 3 beta(p1,p2), 4 gamma(p1); n successive draws of ONE stream */
    rng_t r; rng_seed(&r, seed, chain, iter, kind, index);
    for (int64_t i = 0; i < n; i++) {
        switch (what) {
            case 0: out[i] = rng_uniform(&r); break;
            case 1: out[i] = rng_normal(&r); break;
            case 2: out[i] = rng_chisq(&r, p1); break;
            case 3: out[i] = rng_beta(&r, p1, p2); break;
            default: out[i] = rng_gamma(&r, p1); break;
        }
    }
}
/* first draw of n different streams index0..index0+n-1 (how the sampler uses them) */
void ora_draws_indexed(uint64_t seed, uint64_t chain, uint64_t iter, uint64_t kind, uint64_t index0, int what, double p1,
                       double p2, int64_t n, double *out) {
    for (int64_t i = 0; i < n; i++) ora_draws(seed, chain, iter, kind, index0 + (uint64_t)i, what, p1, p2, 1, out + i);
}

/* ------------------------------------------------------------------ */
/* Synthetic panel: g_ij ~ Binomial(2,p_j),      */
/* p_j ~ U(maf_lo, maf_hi), counter-based so CPU and GPU agree bitwise.  */
/* ------------------------------------------------------------------ */
static inline double panel_pj(uint64_t pseed, int64_t j, double lo, double hi) {
    uint64_t h = mix64(mix64(pseed ^ 0xA5A5A5A55A5A5A5AULL) + (uint64_t)j * GOLD);
    double u = ((double)(h >> 12) + 0.5) * 2.220446049250313080847263336181640625e-16;
    return lo + (hi - lo) * u;
}
static inline int panel_gij(uint64_t pseed, int64_t i, int64_t j, uint32_t thr) {
    uint64_t h = mix64(mix64(pseed + (uint64_t)j * 0xD1342543DE82EF95ULL) ^ ((uint64_t)i * GOLD + 0x632BE59BD9B4E019ULL));
    return (int)((uint32_t)h < thr) + (int)((uint32_t)(h >> 32) < thr);
}
/* genotype codes of chosen ROWS of the synthetic panel (all P columns), row-major [nrows][P], and the column sums over ALL N rows of
   chosen COLUMNS: a host-side check of a full-size panel regenerates a few rows and audits a few column means without the panel */
void ora_generate_rows(const int64_t *rows, int64_t nrows, int64_t P, double maf_lo, double maf_hi, uint64_t pseed, uint8_t *G) {
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < P; j++) {
        const uint32_t thr = (uint32_t)(panel_pj(pseed, j, maf_lo, maf_hi) * 4294967296.0);
        for (int64_t r = 0; r < nrows; r++) G[r * P + j] = (uint8_t)panel_gij(pseed, rows[r], j, thr);
    }
}
void ora_column_sums(const int64_t *cols, int64_t ncols, int64_t N, double maf_lo, double maf_hi, uint64_t pseed, int64_t *sums) {
#pragma omp parallel for schedule(static)
    for (int64_t c = 0; c < ncols; c++) {
        const uint32_t thr = (uint32_t)(panel_pj(pseed, cols[c], maf_lo, maf_hi) * 4294967296.0);
        int64_t sm = 0;
        for (int64_t i = 0; i < N; i++) sm += panel_gij(pseed, i, cols[c], thr);
        sums[c] = sm;
    }
}
/* column-major N x P, centred fp32; optionally returns raw genotype sums per column */
void ora_generate_panel(int64_t N, int64_t P, double maf_lo, double maf_hi, uint64_t pseed, float *X, double *col_mean) {
    for (int64_t j = 0; j < P; j++) {
        double pj = panel_pj(pseed, j, maf_lo, maf_hi);
        uint32_t thr = (uint32_t)(pj * 4294967296.0);
        float *col = X + j * N;
        int64_t sum = 0;
        for (int64_t i = 0; i < N; i++) { int g = panel_gij(pseed, i, j, thr); col[i] = (float)g; sum += g; }
        double mu = (double)sum / (double)N;
        for (int64_t i = 0; i < N; i++) col[i] = (float)((double)col[i] - mu);
        if (col_mean) col_mean[j] = mu;
    }
}

/* ------------------------------------------------------------------ */
/* handle                                                               */
/* ------------------------------------------------------------------ */
typedef struct {
    int64_t col0, ncol;
    int method;
    double df, scale;
    int64_t nreg;
    int64_t *reg_start, *reg_stop; /* 0-based [start,stop) relative to the set */
    int64_t vb_off;                /* offset into varBeta[] */
    double piHat[2], logPi[2];
    int estPi;
    double sum_pi[2];
    /* BayesR (functions.jl:238-289): K variance classes with multipliers vcls of the set's single variance */
    int K; double vcls[RMAX], pic[RMAX], logpic[RMAX], sum_pic[RMAX];
    /* Tuple BayesPR (functions.jl:140-154): tk correlated sets share nloc loci; the k columns of a locus are adjacent in the panel
       (tcol); scale tk x tk (mme.jl:501), nreg variance MATRICES (tk x tk each, row-major) in varBeta */
    int tk; int64_t nloc; double tscale[KMAX * KMAX];
    double *tmpm;  /* reference order: X_l'X_l of every locus, [nloc][tk][tk] (mme.jl:462) */
} oset_t;

/* Panel column of component m of tuple locus l: every 64-column block holds floor(64 / k) whole loci, component-minor (a locus
   never straddles two blocks: the block chain draws its k effects in one step); for k = 3 column 63 of each block stays unused. */
static inline int64_t tcol(const oset_t *S, int64_t l, int m) {
    const int64_t Lb = BLK / S->tk;
    return S->col0 + BLK * (l / Lb) + (int64_t)S->tk * (l % Lb) + m;
}

/* ---- k x k helpers of the Tuple path (row-major, k <= KMAX).  Every operation is written out (fma where stated, everything
   else separately rounded): the device repeats this text operation for operation (csrc/ngp_common.h), so both give the same bits.
   k = 1 takes the scalar forms of the Symbol path (1 / x, sqrt x), which makes a one-set tuple that path, bit for bit. ---- */
static int t_chol(const double *S, int k, double *L) { /* S = L L', L lower; returns -1 if S is not positive definite */
    for (int a = 0; a < k * k; a++) L[a] = 0.0;
    for (int i = 0; i < k; i++)
        for (int j = 0; j <= i; j++) {
            double s = S[i * k + j];
            for (int m = 0; m < j; m++) s = __builtin_fma(-L[i * k + m], L[j * k + m], s);
            if (i == j) { if (!(s > 0.0)) return -1; L[i * k + i] = sqrt(s); }
            else L[i * k + j] = s / L[j * k + j];
        }
    return 0;
}
static int t_spd_inv(const double *S, int k, double *out) { /* inv(S) through the Cholesky factor: inv(L)' inv(L) */
    if (k == 1) { out[0] = 1.0 / S[0]; return (S[0] > 0.0) ? 0 : -1; }
    double L[KMAX * KMAX], Li[KMAX * KMAX];
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
static int t_chol1(const double *S, int k, double *L) { /* Cholesky factor with the scalar form for k = 1 */
    if (k == 1) { L[0] = sqrt(S[0]); return (S[0] >= 0.0) ? 0 : -1; }
    return t_chol(S, k, L);
}
/* varBeta ~ InverseWishart(nu, Psi) (functions.jl:513-516; Distributions.jl is absent: Bartlett's construction on the keyed
   streams).  W = (L A)(L A)' with L = chol(inv(Psi)), A lower triangular, A_ii = sqrt(chi2(nu - i)), A_ij ~ N(0,1) (i > j);
   varBeta = inv(W).  k = 1: Psi / chi2(nu), the Symbol path's form.  The (0,0) chi-square is the region's chi-square of the
   Symbol path (KIND_REGION_CHI2, (set << 40) | region). */
static int t_inverse_wishart(uint64_t seed, uint64_t chain, uint64_t it, int si, int64_t rg, double nu, const double *Psi, int k, double *out) {
    rng_t r;
    if (k == 1) {
        rng_seed(&r, seed, chain, it, KIND_REGION_CHI2, ((uint64_t)si << 40) | (uint64_t)rg);
        const double ch = rng_chisq(&r, nu);
        out[0] = Psi[0] / ch;
        return 0;
    }
    double Pi[KMAX * KMAX], L[KMAX * KMAX], A[KMAX * KMAX], LA[KMAX * KMAX], W[KMAX * KMAX];
    if (t_spd_inv(Psi, k, Pi) || t_chol(Pi, k, L)) return -1;
    for (int a = 0; a < k * k; a++) A[a] = 0.0;
    for (int i = 0; i < k; i++)
        for (int j = 0; j <= i; j++) {
            if (i == 0) rng_seed(&r, seed, chain, it, KIND_REGION_CHI2, ((uint64_t)si << 40) | (uint64_t)rg);
            else rng_seed(&r, seed, chain, it, KIND_T_WISHART, ((uint64_t)si << 40) | ((uint64_t)rg << 8) | ((uint64_t)i << 4) | (uint64_t)j);
            if (i == j) { const double ch = rng_chisq(&r, nu - (double)i); A[i * k + i] = sqrt(ch); }
            else A[i * k + j] = rng_normal(&r);
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
    return t_spd_inv(W, k, out);
}

typedef struct {
    int order; /* 0 reference, 1 blocked */
    uint64_t seed; uint64_t chain;
    int64_t N, P;
    /* reference order storage (fp64 data + second transposed copy, mme.jl:308) */
    double *data, *Mp;
    /* blocked order storage */
    int64_t R, S, NBLK, Ppad; /* rows per shard, shards, blocks */
    int64_t D;                /* lag of the pipelined sweep (1 = no look-ahead) */
    int64_t near;             /* look-ahead lags 1..near corrected by the sampler, farther ones folded into the group sums */
    int64_t nchain;           /* GEMV chains of a shard partial: 8 (phase streamer, per-block engine) or 7 (row-owning waves) */
    int tform;                /* blocked order: linear blocks (every lane BayesPR or unowned) take the chain as dlt = T e0 with the
                                 explicit inverse T of the block's unit lower-triangular system */
    double *gramx;            /* [t][d][k][j], d = 1..D-1: x_{t-d,k}' x_{t,j} */
    float *tiles;             /* [s][t][j][i] */
    int storage;              /* 0: fp32 centred tiles; 1: compact -- one byte per genotype + fp64 column means, centred analytically */
    uint8_t *tiles8;          /* [s][t][j][i] genotype codes (storage 1; rows beyond N are 0) */
    double *mean;             /* Ppad column means (storage 1) */
    double *gram;             /* [t][64][64] */
    double *mpm;              /* Ppad (reference order: P) */
    double *lhs0, *rhs0;      /* Ppad */
    /* model */
    int nsets; oset_t sets[16];
    int64_t nvb; double *varBeta; double *sum_varBeta;
    double e_df, e_scale;
    int intercept;
    /* state */
    double *y;     /* N */
    double *ycorr; /* N (blocked: S*R padded) */
    double b;
    double *beta;  /* Ppad */
    int64_t *delta;
    double varE;
    int64_t iter;
    /* schedule + posterior sums */
    int64_t chainLength, burnIn, thin;
    int64_t nKept;
    double *sum_beta, *sum_beta2, *sum_delta;
    double sum_varE, sum_b;
    /* traces of the last ora_run */
    int64_t ntrace; double *tr_varE, *tr_b;
    /* scratch for blocked */
    double *c, *w, *q, *T, *chi;
    /* fixed-effect sets beyond the intercept (functions.jl:22-53; set-up mme.jl:120-152): columns, X'X, X'X + ridge */
    int nfix; struct { int64_t ncol, off; double *X, *xpx0, *xpxR, *lhs0, *rhs0; } fix[16];
    int64_t nfixcol; double *bfix, *sum_bfix;
    double *rcls; /* BayesR per-locus class coefficients of the blocked order: [4][RMAX][Ppad] = 1/lhs, a, sd z, u */
    double *tupc, *tupg; /* Tuple sets, blocked order: [KMAX][Ppad] row of C = iVarE inv(LHS) and of X_l'X_l of every column */
    char err[256];
} ora_t;

int ora_create(int order, uint64_t seed, uint32_t chain, ora_t **out) {
    ora_t *h = (ora_t *)calloc(1, sizeof(ora_t));
    if (!h) return ORA_ERR;
    h->order = order; h->seed = seed; h->chain = chain;
    h->e_df = 4.0; h->e_scale = 0.0005; h->intercept = 1;
    h->chainLength = 0; h->burnIn = 0; h->thin = 1;
    h->near = 3; h->nchain = 8; h->tform = 0;
    *out = h; return ORA_OK;
}
/* which look-ahead lags the sampler corrects itself (the library reports its choice: ngp_get_near_lags) */
int ora_set_near(ora_t *h, int64_t near) {
    if (near < 1 || near > 8) { snprintf(h->err, 256, "near lags out of range"); return ORA_ERR; }
    h->near = near; return ORA_OK;
}
/* GEMV chains per shard partial (the library reports its choice: ngp_get_streamer) */
int ora_set_nchain(ora_t *h, int64_t n) {
    if (n != 7 && n != 8) { snprintf(h->err, 256, "GEMV chains must be 7 or 8"); return ORA_ERR; }
    h->nchain = n; return ORA_OK;
}
/* chain form of the linear blocks (the library reports its choice: ngp_get_chain_form): 0 = 64 steps (default), 1 = inverse form */
int ora_set_tform(ora_t *h, int on) { h->tform = on ? 1 : 0; return ORA_OK; }
static void free_sets(ora_t *h) {
    for (int s = 0; s < h->nsets; s++) { free(h->sets[s].reg_start); free(h->sets[s].reg_stop); free(h->sets[s].tmpm); }
}
void ora_destroy(ora_t *h) {
    if (!h) return;
    free(h->data); free(h->Mp); free(h->tiles); free(h->tiles8); free(h->mean); free(h->gram); free(h->gramx); free(h->mpm); free(h->lhs0); free(h->rhs0);
    free_sets(h); free(h->varBeta); free(h->sum_varBeta); free(h->y); free(h->ycorr); free(h->beta); free(h->delta);
    free(h->sum_beta); free(h->sum_beta2); free(h->sum_delta); free(h->tr_varE); free(h->tr_b);
    free(h->c); free(h->w); free(h->q); free(h->T); free(h->chi); free(h->rcls); free(h->tupc); free(h->tupg);
    free(h);
}
const char *ora_last_error(ora_t *h) { return h->err; }

static inline double dot8_1(const double *a, const double *b, int64_t n);
static double wave_butterfly(double *v);
/* hierarchical sequential sum of n shard partials in groups of GRP */
static double group_sum(const double *p, int64_t n, int64_t stride) {
    double tot = 0.0;
    for (int64_t g = 0; g * GRP < n; g++) {
        int64_t s0 = g * GRP, s1 = s0 + GRP < n ? s0 + GRP : n;
        double gs = p[s0 * stride];
        for (int64_t s = s0 + 1; s < s1; s++) gs = gs + p[s * stride];
        tot = (g == 0) ? gs : tot + gs;
    }
    return tot;
}

/* the panel as the product stores it: fp32, column-major N x P, already centred.
   For order 1 the caller states the device layout (rows per shard R, shards S). */
int ora_set_panel_f32(ora_t *h, const float *X, int64_t N, int64_t P, int64_t R, int64_t S, int64_t D) {
    h->N = N; h->P = P;
    if (h->order == 0) {
        h->data = (double *)malloc(sizeof(double) * N * P);
        h->Mp = (double *)malloc(sizeof(double) * N * P);
        h->mpm = (double *)calloc(P, sizeof(double));
        if (!h->data || !h->Mp || !h->mpm) { snprintf(h->err, 256, "out of memory"); return ORA_ERR; }
        for (int64_t j = 0; j < P; j++) {
            const float *c = X + j * N; double *d = h->data + j * N, *m = h->Mp + j * N;
            double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            int64_t i = 0;
            for (; i + 8 <= N; i += 8)
                for (int l = 0; l < 8; l++) { double v = (double)c[i + l]; d[i + l] = v; m[i + l] = v; acc[l] = __builtin_fma(v, v, acc[l]); }
            double tail = 0;
            for (; i < N; i++) { double v = (double)c[i]; d[i] = v; m[i] = v; tail = __builtin_fma(v, v, tail); }
            h->mpm[j] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7])) + tail; /* mme.jl:305-307 */
        }
        h->Ppad = P;
    } else {
        if (R % 4 || R * S < N || R <= 0) { snprintf(h->err, 256, "bad layout R=%lld S=%lld", (long long)R, (long long)S); return ORA_ERR; }
        if (D < 1 || D > 16) { snprintf(h->err, 256, "bad lag D=%lld", (long long)D); return ORA_ERR; }
        h->R = R; h->S = S; h->D = D; h->NBLK = (P + BLK - 1) / BLK; h->Ppad = h->NBLK * BLK;
        size_t tile = (size_t)R * BLK;
        h->tiles = (float *)calloc((size_t)S * h->NBLK * tile, sizeof(float));
        h->gram = (double *)calloc((size_t)h->NBLK * BLK * BLK, sizeof(double));
        h->mpm = (double *)calloc(h->Ppad, sizeof(double));
        if (!h->tiles || !h->gram || !h->mpm) { snprintf(h->err, 256, "out of memory"); return ORA_ERR; }
        for (int64_t j = 0; j < P; j++) {
            int64_t t = j / BLK, jj = j % BLK;
            for (int64_t i = 0; i < N; i++) {
                int64_t s = i / R, ii = i % R;
                h->tiles[((size_t)s * h->NBLK + t) * tile + jj * R + ii] = X[j * N + i];
            }
        }
        /* Gram blocks: shard partial = sequential fma over the shard's rows, then group sums */
        double *part = (double *)malloc(sizeof(double) * S);
        for (int64_t t = 0; t < h->NBLK; t++)
            for (int k = 0; k < BLK; k++)
                for (int j = 0; j <= k; j++) {
                    for (int64_t s = 0; s < S; s++) {
                        const float *tl = h->tiles + ((size_t)s * h->NBLK + t) * tile;
                        double acc = 0.0;
                        for (int64_t i = 0; i < R; i++) acc = __builtin_fma((double)tl[k * R + i], (double)tl[j * R + i], acc);
                        part[s] = acc;
                    }
                    double g = group_sum(part, S, 1);
                    h->gram[((size_t)t * BLK + k) * BLK + j] = g;
                    h->gram[((size_t)t * BLK + j) * BLK + k] = g;
                }
        /* cross blocks of the look-ahead window: Gx[t][d][k][j] = x_{t-d,k}' x_{t,j} */
        if (D > 1) {
            h->gramx = (double *)calloc((size_t)h->NBLK * D * BLK * BLK, sizeof(double));
            if (!h->gramx) { snprintf(h->err, 256, "out of memory"); return ORA_ERR; }
            for (int64_t t = 0; t < h->NBLK; t++)
                for (int64_t d = 1; d < D && d <= t; d++)
                    for (int k = 0; k < BLK; k++)
                        for (int j = 0; j < BLK; j++) {
                            for (int64_t s = 0; s < S; s++) {
                                const float *ta = h->tiles + ((size_t)s * h->NBLK + (t - d)) * tile;
                                const float *tt = h->tiles + ((size_t)s * h->NBLK + t) * tile;
                                double acc = 0.0;
                                for (int64_t i = 0; i < R; i++) acc = __builtin_fma((double)ta[k * R + i], (double)tt[j * R + i], acc);
                                part[s] = acc;
                            }
                            h->gramx[(((size_t)t * D + d) * BLK + k) * BLK + j] = group_sum(part, S, 1);
                        }
        }
        free(part);
        for (int64_t k = 0; k < h->Ppad; k++) h->mpm[k] = h->gram[((size_t)(k / BLK) * BLK + k % BLK) * BLK + k % BLK];
    }
    h->lhs0 = (double *)calloc(h->Ppad, sizeof(double));
    h->rhs0 = (double *)calloc(h->Ppad, sizeof(double));
    h->beta = (double *)calloc(h->Ppad, sizeof(double));               /* mme.jl:443 */
    h->delta = (int64_t *)malloc(sizeof(int64_t) * h->Ppad);          /* mme.jl:444 */
    for (int64_t k = 0; k < h->Ppad; k++) h->delta[k] = 1;
    h->sum_beta = (double *)calloc(h->Ppad, sizeof(double));
    h->sum_beta2 = (double *)calloc(h->Ppad, sizeof(double));
    h->sum_delta = (double *)calloc(h->Ppad, sizeof(double));
    h->c = (double *)calloc(h->Ppad, sizeof(double)); h->w = (double *)calloc(h->Ppad, sizeof(double));
    h->q = (double *)calloc(h->Ppad, sizeof(double)); h->T = (double *)calloc(h->Ppad, sizeof(double));
    h->chi = (double *)calloc(h->Ppad, sizeof(double));
    return ORA_OK;
}

/* Compact storage (blocked order only): the panel stays one byte per genotype, the centred value
   x_ij = g_ij - m_j (prepMatVec.jl:129) is never formed -- m_j = (sum_i g_ij) / N in Float64 as the reference computes it, and
   every product with a centred column is taken analytically:
     x_j' y        = sum_i g_ij y_i - m_j sum_i y_i                     (per shard: chains over the rows + the shard's sum of y)
     y -= x_j dlt  = y_i - (sum_j g_ij dlt_j - sum_j m_j dlt_j)          (valid rows only; padding rows stay zero)
     x_k' x_j      = (exact integer dot product) - N (m_k m_j)
   R must be a multiple of 16 (a row-owning wave holds units of 16 rows); 7 GEMV chains as in the row-owning streamer. */
int ora_set_panel_u8(ora_t *h, const uint8_t *G, int64_t N, int64_t P, int64_t R, int64_t S, int64_t D, int centre) {
    if (h->order != 1) { snprintf(h->err, 256, "compact panel: blocked order only (reference order: pass g - mean as Float64)"); return ORA_ERR; }
    if (R % 16 || R * S < N || R <= 0) { snprintf(h->err, 256, "bad layout R=%lld S=%lld", (long long)R, (long long)S); return ORA_ERR; }
    if (D < 1 || D > 16) { snprintf(h->err, 256, "bad lag D=%lld", (long long)D); return ORA_ERR; }
    h->N = N; h->P = P; h->storage = 1; h->nchain = 7;
    h->R = R; h->S = S; h->D = D; h->NBLK = (P + BLK - 1) / BLK; h->Ppad = h->NBLK * BLK;
    const size_t tile = (size_t)R * BLK;
    h->tiles8 = (uint8_t *)calloc((size_t)S * h->NBLK * tile, 1);
    h->mean = (double *)calloc(h->Ppad, sizeof(double));
    h->gram = (double *)calloc((size_t)h->NBLK * BLK * BLK, sizeof(double));
    h->mpm = (double *)calloc(h->Ppad, sizeof(double));
    if (!h->tiles8 || !h->mean || !h->gram || !h->mpm) { snprintf(h->err, 256, "out of memory"); return ORA_ERR; }
    for (int64_t j = 0; j < P; j++) {
        const int64_t t = j / BLK, jj = j % BLK;
        uint64_t sum = 0;
        for (int64_t i = 0; i < N; i++) {
            const int64_t s = i / R, ii = i % R;
            h->tiles8[((size_t)s * h->NBLK + t) * tile + jj * R + ii] = G[j * N + i];
            sum += G[j * N + i];
        }
        h->mean[j] = centre ? (double)sum / (double)N : 0.0;
    }
    const double Nd = (double)N;
    if (D > 1) {
        h->gramx = (double *)calloc((size_t)h->NBLK * D * BLK * BLK, sizeof(double));
        if (!h->gramx) { snprintf(h->err, 256, "out of memory"); return ORA_ERR; }
    }
    for (int64_t t = 0; t < h->NBLK; t++)
        for (int64_t d = 0; d < D && d <= t; d++)
            for (int k = 0; k < BLK; k++)
                for (int j = 0; j < BLK; j++) {
                    uint64_t dot = 0;
                    for (int64_t s = 0; s < S; s++) {
                        const uint8_t *ta = h->tiles8 + ((size_t)s * h->NBLK + (t - d)) * tile + (size_t)k * R;
                        const uint8_t *tt = h->tiles8 + ((size_t)s * h->NBLK + t) * tile + (size_t)j * R;
                        uint32_t acc = 0;
                        for (int64_t i = 0; i < R; i++) acc += (uint32_t)ta[i] * (uint32_t)tt[i];
                        dot += acc;
                    }
                    const double mm = h->mean[(t - d) * BLK + k] * h->mean[t * BLK + j];
                    const double nm = Nd * mm;
                    const double g = (double)dot - nm;
                    if (d == 0) h->gram[((size_t)t * BLK + k) * BLK + j] = g;
                    else h->gramx[(((size_t)t * D + d) * BLK + k) * BLK + j] = g;
                }
    for (int64_t k = 0; k < h->Ppad; k++) h->mpm[k] = h->gram[((size_t)(k / BLK) * BLK + k % BLK) * BLK + k % BLK];
    h->lhs0 = (double *)calloc(h->Ppad, sizeof(double));
    h->rhs0 = (double *)calloc(h->Ppad, sizeof(double));
    h->beta = (double *)calloc(h->Ppad, sizeof(double));
    h->delta = (int64_t *)malloc(sizeof(int64_t) * h->Ppad);
    for (int64_t k = 0; k < h->Ppad; k++) h->delta[k] = 1;
    h->sum_beta = (double *)calloc(h->Ppad, sizeof(double));
    h->sum_beta2 = (double *)calloc(h->Ppad, sizeof(double));
    h->sum_delta = (double *)calloc(h->Ppad, sizeof(double));
    h->c = (double *)calloc(h->Ppad, sizeof(double)); h->w = (double *)calloc(h->Ppad, sizeof(double));
    h->q = (double *)calloc(h->Ppad, sizeof(double)); h->T = (double *)calloc(h->Ppad, sizeof(double));
    h->chi = (double *)calloc(h->Ppad, sizeof(double));
    return ORA_OK;
}
int ora_get_means(ora_t *h, double *out, int64_t P) {
    if (!h->mean || P != h->P) { snprintf(h->err, 256, "no compact panel / size mismatch"); return ORA_ERR; }
    memcpy(out, h->mean, sizeof(double) * P);
    return ORA_OK;
}

/* Reference order only: the panel as the REFERENCE holds it -- Float64, centred in Float64 (prepMatVec.jl:129) -- so that the
   effect of the product's fp32 tiles can be measured against it (tests/test_oracle.py::test_fp32_panel_deviation). */
int ora_set_panel_f64(ora_t *h, const double *X, int64_t N, int64_t P) {
    if (h->order != 0) { snprintf(h->err, 256, "Float64 panel: reference order only"); return ORA_ERR; }
    h->N = N; h->P = P;
    h->data = (double *)malloc(sizeof(double) * N * P);
    h->Mp = (double *)malloc(sizeof(double) * N * P);
    h->mpm = (double *)calloc(P, sizeof(double));
    if (!h->data || !h->Mp || !h->mpm) { snprintf(h->err, 256, "out of memory"); return ORA_ERR; }
    memcpy(h->data, X, sizeof(double) * N * P);
    memcpy(h->Mp, X, sizeof(double) * N * P);
    for (int64_t j = 0; j < P; j++) h->mpm[j] = dot8_1(h->data + j * N, h->data + j * N, N); /* mme.jl:305-307 */
    h->Ppad = P;
    h->lhs0 = (double *)calloc(h->Ppad, sizeof(double));
    h->rhs0 = (double *)calloc(h->Ppad, sizeof(double));
    h->beta = (double *)calloc(h->Ppad, sizeof(double));
    h->delta = (int64_t *)malloc(sizeof(int64_t) * h->Ppad);
    for (int64_t k = 0; k < h->Ppad; k++) h->delta[k] = 1;
    h->sum_beta = (double *)calloc(h->Ppad, sizeof(double));
    h->sum_beta2 = (double *)calloc(h->Ppad, sizeof(double));
    h->sum_delta = (double *)calloc(h->Ppad, sizeof(double));
    h->c = (double *)calloc(h->Ppad, sizeof(double)); h->w = (double *)calloc(h->Ppad, sizeof(double));
    h->q = (double *)calloc(h->Ppad, sizeof(double)); h->T = (double *)calloc(h->Ppad, sizeof(double));
    h->chi = (double *)calloc(h->Ppad, sizeof(double));
    return ORA_OK;
}

/* marker set = consecutive column range with its prior (mme.jl:324-361, 493-516).
   regions: 0-based [start,stop) relative to the set.  varBeta0: nreg initial values. */
int ora_add_marker_set(ora_t *h, int64_t col0, int64_t ncol, int method, double df, double scale, const int64_t *reg_start,
                       const int64_t *reg_stop, int64_t nreg, const double *varBeta0, double pi0, int estPi,
                       const double *lhs0, const double *rhs0, int *set_id) {
    if (h->nsets >= 16) { snprintf(h->err, 256, "too many sets"); return ORA_ERR; }
    if (col0 < 0 || col0 + ncol > h->P) { snprintf(h->err, 256, "set outside panel"); return ORA_ERR; }
    for (int q = 0; q < h->nsets; q++) {  /* a tuple set owns its 64-column blocks to the end of the last one */
        const oset_t *o = &h->sets[q];
        const int64_t oend = (o->method == METHOD_T) ? o->col0 + BLK * ((o->ncol + BLK - 1) / BLK) : o->col0 + o->ncol;
        if (col0 < oend && o->col0 < col0 + ncol) { snprintf(h->err, 256, "marker sets overlap"); return ORA_ERR; }
    }
    oset_t *s = &h->sets[h->nsets];
    memset(s, 0, sizeof(*s));
    s->col0 = col0; s->ncol = ncol; s->method = method; s->df = df; s->scale = scale; s->nreg = nreg; s->estPi = estPi;
    s->reg_start = (int64_t *)malloc(sizeof(int64_t) * nreg); s->reg_stop = (int64_t *)malloc(sizeof(int64_t) * nreg);
    memcpy(s->reg_start, reg_start, sizeof(int64_t) * nreg); memcpy(s->reg_stop, reg_stop, sizeof(int64_t) * nreg);
    s->vb_off = h->nvb;
    h->varBeta = (double *)realloc(h->varBeta, sizeof(double) * (h->nvb + nreg));
    h->sum_varBeta = (double *)realloc(h->sum_varBeta, sizeof(double) * (h->nvb + nreg));
    for (int64_t r = 0; r < nreg; r++) { h->varBeta[h->nvb + r] = varBeta0[r]; h->sum_varBeta[h->nvb + r] = 0.0; }
    h->nvb += nreg;
    s->piHat[0] = 1.0 - pi0; s->piHat[1] = pi0; /* mme.jl:351,360 */
    if (h->order == 0) { s->logPi[0] = log(1.0 - pi0); s->logPi[1] = log(pi0); }
    else { s->logPi[0] = det_log(1.0 - pi0); s->logPi[1] = det_log(pi0); }
    for (int64_t k = 0; k < ncol; k++) {
        h->lhs0[col0 + k] = lhs0 ? lhs0[k] : 0.0; /* mme.jl:314-322 */
        h->rhs0[col0 + k] = rhs0 ? rhs0[k] : 0.0;
    }
    if (set_id) *set_id = h->nsets;
    h->nsets++;
    return ORA_OK;
}

/* A fixed-effect set = the columns of one model term (or of one `blockThese` group): N x ncol, column-major.  X'X as the
   reference forms it (mme.jl:137), plus, for ncol > 1, the ridge min|diag| / 10000 on the diagonal (mme.jl:149-152). */
int ora_add_fixed_set(ora_t *h, const double *X, int64_t N, int64_t ncol, const double *lhs0, const double *rhs0, int *set_id) {
    if (h->nfix >= 16 || N != h->N || ncol < 1 || ncol > 64) { snprintf(h->err, 256, "bad fixed-effect set"); return ORA_ERR; }
    int f = h->nfix;
    h->fix[f].ncol = ncol; h->fix[f].off = h->nfixcol;
    h->fix[f].X = (double *)malloc(sizeof(double) * N * ncol); memcpy(h->fix[f].X, X, sizeof(double) * N * ncol);
    h->fix[f].xpx0 = (double *)calloc(ncol * ncol, sizeof(double)); h->fix[f].xpxR = (double *)calloc(ncol * ncol, sizeof(double));
    h->fix[f].lhs0 = (double *)calloc(ncol, sizeof(double)); h->fix[f].rhs0 = (double *)calloc(ncol, sizeof(double));
    for (int64_t a = 0; a < ncol; a++) {
        for (int64_t b = 0; b <= a; b++) {
            double acc = 0.0;
            for (int64_t i = 0; i < N; i++) acc = __builtin_fma(X[a * N + i], X[b * N + i], acc);
            h->fix[f].xpx0[a * ncol + b] = acc; h->fix[f].xpx0[b * ncol + a] = acc;
        }
        if (lhs0) h->fix[f].lhs0[a] = lhs0[a];
        if (rhs0) h->fix[f].rhs0[a] = rhs0[a];
    }
    memcpy(h->fix[f].xpxR, h->fix[f].xpx0, sizeof(double) * ncol * ncol);
    if (ncol > 1) {
        double mn = fabs(h->fix[f].xpx0[0]);
        for (int64_t a = 1; a < ncol; a++) { double d = fabs(h->fix[f].xpx0[a * ncol + a]); if (d < mn) mn = d; }
        for (int64_t a = 0; a < ncol; a++) h->fix[f].xpxR[a * ncol + a] += mn / 10000.0;
    }
    h->nfixcol += ncol;
    h->bfix = (double *)realloc(h->bfix, sizeof(double) * h->nfixcol); h->sum_bfix = (double *)realloc(h->sum_bfix, sizeof(double) * h->nfixcol);
    for (int64_t a = h->fix[f].off; a < h->nfixcol; a++) { h->bfix[a] = 0.0; h->sum_bfix[a] = 0.0; }
    if (set_id) *set_id = f;
    h->nfix++;
    return ORA_OK;
}
int ora_get_fixed(ora_t *h, double *b, double *sum_b, int64_t *n) {
    if (n) *n = h->nfixcol;
    for (int64_t a = 0; a < h->nfixcol; a++) { if (b) b[a] = h->bfix[a]; if (sum_b) sum_b[a] = h->sum_bfix[a]; }
    return ORA_OK;
}

/* reference order: sampleX! / sampleb! (functions.jl:22-53) for every fixed set, in the order the sets were added */
static void fixed_ref(ora_t *h, int64_t it, double iVarE) {
    const int64_t N = h->N;
    rng_t r;
    for (int f = 0; f < h->nfix; f++) {
        const int64_t nc = h->fix[f].ncol;
        const double *X = h->fix[f].X;
        double *b = h->bfix + h->fix[f].off;
        if (nc == 1) {                                                                     /* :41-47 */
            for (int64_t i = 0; i < N; i++) h->ycorr[i] += X[i] * b[0];
            double rhs = dot8_1(X, h->ycorr, N) * iVarE + h->fix[f].rhs0[0];
            double lhs = h->fix[f].xpx0[0] * iVarE + h->fix[f].lhs0[0];
            double mean = rhs / lhs;
            rng_seed(&r, h->seed, h->chain, it, KIND_FIXED_NORMAL, ((uint64_t)(f + 1) << 20));
            b[0] = mean + sqrt(1.0 / lhs) * rng_normal(&r);
            for (int64_t i = 0; i < N; i++) h->ycorr[i] -= X[i] * b[0];
        } else {                                                                           /* :48-52, :22-36 */
            double Yi[64], bVec[64];
            for (int64_t i = 0; i < N; i++) { double t = 0.0; for (int64_t a = 0; a < nc; a++) t += X[a * N + i] * b[a]; h->ycorr[i] += t; }   /* :49 */
            for (int64_t a = 0; a < nc; a++) { Yi[a] = dot8_1(X + a * N, h->ycorr, N) * iVarE; bVec[a] = b[a]; }                           /* :25 */
            for (int64_t a = 0; a < nc; a++) {
                bVec[a] = 0.0;                                                             /* :28 */
                double d = 0.0;
                for (int64_t c = 0; c < nc; c++) d += h->fix[f].xpxR[a * nc + c] * bVec[c];
                double rhsb = Yi[a] - d * iVarE;                                           /* :29 */
                double lhsb = h->fix[f].xpxR[a * nc + a] * iVarE;                          /* :30 */
                double inv = 1.0 / lhsb;
                double meanb = inv * rhsb;
                rng_seed(&r, h->seed, h->chain, it, KIND_FIXED_NORMAL, ((uint64_t)(f + 1) << 20) | (uint64_t)a);
                bVec[a] = meanb + sqrt(inv) * rng_normal(&r);                              /* :33 */
            }
            for (int64_t a = 0; a < nc; a++) b[a] = bVec[a];
            for (int64_t i = 0; i < N; i++) { double t = 0.0; for (int64_t a = 0; a < nc; a++) t += X[a * N + i] * b[a]; h->ycorr[i] -= t; }   /* :51 */
        }
    }
}

/* blocked order: one formulation for every width (mirrors k_fixed): Yi_a = (x_a'ycorr + sum_c X'X[a][c] b_c) iVarE with the
   1024-lane dot of the iteration head, Gauss-Seidel on the ridged matrix, ycorr -= X (b_new - b_old) */
static void fixed_blocked(ora_t *h, int64_t it, double iVarE) {
    const int64_t N = h->N;
    rng_t r;
    for (int f = 0; f < h->nfix; f++) {
        const int64_t nc = h->fix[f].ncol;
        const double *X = h->fix[f].X;
        double *b = h->bfix + h->fix[f].off;
        double Yi[64], bVec[64], db[64];
        for (int64_t a = 0; a < nc; a++) {
            double w16[16];
            for (int wv = 0; wv < 16; wv++) {
                double lane[64];
                for (int l = 0; l < 64; l++) {
                    double acc = 0.0;
                    for (int64_t i = wv * 64 + l; i < N; i += 1024) acc = __builtin_fma(X[a * N + i], h->ycorr[i], acc);
                    lane[l] = acc;
                }
                w16[wv] = wave_butterfly(lane);
            }
            double d = w16[0];
            for (int wv = 1; wv < 16; wv++) d = d + w16[wv];
            double xb = 0.0;
            for (int64_t c = 0; c < nc; c++) xb = __builtin_fma(h->fix[f].xpx0[a * nc + c], b[c], xb);
            double tot = d + xb;
            Yi[a] = tot * iVarE;
            bVec[a] = b[a];
        }
        for (int64_t a = 0; a < nc; a++) {
            bVec[a] = 0.0;
            double d = 0.0;
            for (int64_t c = 0; c < nc; c++) d = __builtin_fma(h->fix[f].xpxR[a * nc + c], bVec[c], d);
            double t1 = d * iVarE;
            double rhsb = (nc == 1) ? (Yi[a] + h->fix[f].rhs0[0]) : (Yi[a] - t1);
            double lhsb = h->fix[f].xpxR[a * nc + a] * iVarE;
            if (nc == 1) lhsb = lhsb + h->fix[f].lhs0[0];
            double inv = 1.0 / lhsb;
            double meanb = inv * rhsb;
            rng_seed(&r, h->seed, h->chain, it, KIND_FIXED_NORMAL, ((uint64_t)(f + 1) << 20) | (uint64_t)a);
            double z = rng_normal(&r);
            double sd = sqrt(inv); double tz = sd * z;
            bVec[a] = meanb + tz;
        }
        for (int64_t a = 0; a < nc; a++) { db[a] = bVec[a] - b[a]; b[a] = bVec[a]; }
        for (int64_t i = 0; i < N; i++) {
            double t = 0.0;
            for (int64_t a = 0; a < nc; a++) t = __builtin_fma(X[a * N + i], db[a], t);
            h->ycorr[i] = h->ycorr[i] - t;
        }
    }
}

/* BayesR set (mme.jl:374-383): one variance (nVarCov = 1), class multipliers vClass and class probabilities pi (K each) */
int ora_add_marker_set_r(ora_t *h, int64_t col0, int64_t ncol, double df, double scale, double varBeta0, const double *vClass,
                         const double *pi, int K, int estPi, const double *lhs0, const double *rhs0, int *set_id) {
    if (K < 2 || K > RMAX) { snprintf(h->err, 256, "BayesR: 2..%d variance classes", RMAX); return ORA_ERR; }
    const int64_t rs = 0, re = ncol;
    int rc = ora_add_marker_set(h, col0, ncol, METHOD_R, df, scale, &rs, &re, 1, &varBeta0, 0.5, estPi, lhs0, rhs0, set_id);
    if (rc) return rc;
    oset_t *s = &h->sets[h->nsets - 1];
    if (h->order == 1 && !h->rcls) h->rcls = (double *)calloc((size_t)4 * RMAX * h->Ppad, sizeof(double));
    s->K = K;
    for (int v = 0; v < K; v++) {
        s->vcls[v] = vClass[v]; s->pic[v] = pi[v]; s->sum_pic[v] = 0.0;
        s->logpic[v] = (h->order == 0) ? log(pi[v]) : det_log(pi[v]);   /* mme.jl:375 */
    }
    return ORA_OK;
}
/* Correlated marker sets (Tuple BayesPR, mme.jl:448-489): k sets share nloc loci; the panel holds them locus-major (tcol), from a
   block boundary on.  df = 3 + k, scale = v (df - k - 1) as k x k (mme.jl:493, 501: the caller computes them), regions are ranges
   of LOCI, varBeta0 one k x k matrix (every region starts from it, mme.jl:516). */
int ora_add_marker_set_tuple(ora_t *h, int64_t col0, int64_t nloc, int k, double df, const double *scale, const int64_t *reg_start,
                             const int64_t *reg_stop, int64_t nreg, const double *varBeta0, int *set_id) {
    if (h->nsets >= 16 || k < 1 || k > KMAX || nloc < 1 || col0 % BLK) { snprintf(h->err, 256, "bad tuple set (1..4 sets, first column on a block boundary)"); return ORA_ERR; }
    const int64_t Lb = BLK / k, nblk = (nloc + Lb - 1) / Lb, span = BLK * (nblk - 1) + (int64_t)k * (nloc - Lb * (nblk - 1));
    if (col0 < 0 || col0 + span > h->P) { snprintf(h->err, 256, "tuple set outside panel"); return ORA_ERR; }
    for (int q = 0; q < h->nsets; q++)
        if (col0 < h->sets[q].col0 + h->sets[q].ncol && h->sets[q].col0 < col0 + BLK * nblk) { snprintf(h->err, 256, "marker sets overlap"); return ORA_ERR; }
    oset_t *s = &h->sets[h->nsets];
    memset(s, 0, sizeof(*s));
    s->col0 = col0; s->ncol = span; s->method = METHOD_T; s->df = df; s->nreg = nreg; s->tk = k; s->nloc = nloc;
    for (int a = 0; a < k * k; a++) s->tscale[a] = scale[a];
    s->reg_start = (int64_t *)malloc(sizeof(int64_t) * nreg); s->reg_stop = (int64_t *)malloc(sizeof(int64_t) * nreg);
    memcpy(s->reg_start, reg_start, sizeof(int64_t) * nreg); memcpy(s->reg_stop, reg_stop, sizeof(int64_t) * nreg);
    s->vb_off = h->nvb;
    const int64_t nv = nreg * k * k;
    h->varBeta = (double *)realloc(h->varBeta, sizeof(double) * (h->nvb + nv));
    h->sum_varBeta = (double *)realloc(h->sum_varBeta, sizeof(double) * (h->nvb + nv));
    for (int64_t r = 0; r < nreg; r++)
        for (int a = 0; a < k * k; a++) { h->varBeta[h->nvb + r * k * k + a] = varBeta0[a]; h->sum_varBeta[h->nvb + r * k * k + a] = 0.0; }
    h->nvb += nv;
    s->piHat[0] = 0.5; s->piHat[1] = 0.5;
    if (h->order == 0) {  /* mme.jl:462: mpm[l] = X_l'X_l */
        s->tmpm = (double *)malloc(sizeof(double) * nloc * k * k);
        for (int64_t l = 0; l < nloc; l++)
            for (int a = 0; a < k; a++)
                for (int b = 0; b < k; b++)
                    s->tmpm[(l * k + a) * k + b] = dot8_1(h->data + tcol(s, l, a) * h->N, h->data + tcol(s, l, b) * h->N, h->N);
    } else {
        const size_t PP = (size_t)h->Ppad;
        if (!h->tupc) { h->tupc = (double *)calloc(KMAX * PP, sizeof(double)); h->tupg = (double *)calloc(KMAX * PP, sizeof(double)); }
        for (int64_t l = 0; l < nloc; l++)
            for (int m = 0; m < k; m++) {
                const int64_t c = tcol(s, l, m), t = c / BLK;
                for (int b = 0; b < k; b++) {
                    const int64_t cb = tcol(s, l, b);
                    h->tupg[(size_t)b * PP + c] = h->gram[((size_t)t * BLK + c % BLK) * BLK + cb % BLK];
                }
            }
    }
    if (set_id) *set_id = h->nsets;
    h->nsets++;
    return ORA_OK;
}
int ora_get_class_state(ora_t *h, int si, double *piHat, double *sum_pi, int64_t *K) {
    if (si < 0 || si >= h->nsets) return ORA_ERR;
    oset_t *s = &h->sets[si];
    if (K) *K = s->K;
    for (int v = 0; v < s->K; v++) { if (piHat) piHat[v] = s->pic[v]; if (sum_pi) sum_pi[v] = s->sum_pic[v]; }
    return ORA_OK;
}

int ora_set_y(ora_t *h, const double *y, int64_t N) {
    if (N != h->N) { snprintf(h->err, 256, "y length mismatch"); return ORA_ERR; }
    int64_t L = (h->order == 0) ? N : h->R * h->S;
    free(h->y); free(h->ycorr);
    h->y = (double *)malloc(sizeof(double) * N);
    h->ycorr = (double *)calloc(L, sizeof(double));
    memcpy(h->y, y, sizeof(double) * N);
    memcpy(h->ycorr, y, sizeof(double) * N); /* mme.jl:57 */
    h->b = 0.0; h->iter = 0;
    for (int64_t a = 0; a < h->nfixcol; a++) { h->bfix[a] = 0.0; h->sum_bfix[a] = 0.0; }
    return ORA_OK;
}
int ora_set_residual_prior(ora_t *h, double df, double scale) { h->e_df = df; h->e_scale = scale; return ORA_OK; }
int ora_set_intercept(ora_t *h, int on) { h->intercept = on; return ORA_OK; }
int ora_set_schedule(ora_t *h, int64_t chainLength, int64_t burnIn, int64_t thin) {
    h->chainLength = chainLength; h->burnIn = burnIn; h->thin = thin < 1 ? 1 : thin; return ORA_OK;
}

/* samplers.jl:26  these2Keep = (burnIn+thin):thin:chainLength */
static int is_kept(const ora_t *h, int64_t iter) {
    if (iter < h->burnIn + h->thin || iter > h->chainLength) return 0;
    return ((iter - h->burnIn) % h->thin) == 0;
}

static void accumulate(ora_t *h) {
    for (int64_t k = 0; k < h->P; k++) {
        double b = h->beta[k];
        h->sum_beta[k] += b; h->sum_beta2[k] += b * b; h->sum_delta[k] += (double)h->delta[k];
    }
    for (int64_t r = 0; r < h->nvb; r++) h->sum_varBeta[r] += h->varBeta[r];
    for (int s = 0; s < h->nsets; s++) { h->sets[s].sum_pi[0] += h->sets[s].piHat[0]; h->sets[s].sum_pi[1] += h->sets[s].piHat[1]; }
    for (int s = 0; s < h->nsets; s++) for (int v = 0; v < h->sets[s].K; v++) h->sets[s].sum_pic[v] += h->sets[s].pic[v];
    for (int64_t a = 0; a < h->nfixcol; a++) h->sum_bfix[a] += h->bfix[a];
    h->sum_varE += h->varE; h->sum_b += h->b; h->nKept++;
}

/* ------------------------------------------------------------------ */
/* order 0: reference order                                             */
/* ------------------------------------------------------------------ */
/* Threads of the reference-order BLAS-1 calls (the reference runs OpenBLAS, whose ddot / daxpy are multi-threaded for
   long vectors).  1 (default) = the deterministic sequences every test uses; T > 1 (bench.py's all-core baseline only):
   T contiguous chunks, partial dots added in chunk order. */
static int g_threads = 1;
void ora_set_threads(int t) { g_threads = t < 1 ? 1 : t; }
int ora_get_threads(void) { return g_threads; }

static inline double dot8_1(const double *a, const double *b, int64_t n) {
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int64_t i = 0;
    for (; i + 8 <= n; i += 8)
        for (int l = 0; l < 8; l++) acc[l] = __builtin_fma(a[i + l], b[i + l], acc[l]);
    double tail = 0;
    for (; i < n; i++) tail = __builtin_fma(a[i], b[i], tail);
    return ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7])) + tail;
}
static inline double dot8(const double *a, const double *b, int64_t n) {
    if (g_threads <= 1) return dot8_1(a, b, n);
    double part[64];
    const int T = g_threads > 64 ? 64 : g_threads;
    const int64_t chunk = ((n + T - 1) / T + 7) & ~(int64_t)7;
#pragma omp parallel for num_threads(T) schedule(static, 1)
    for (int t = 0; t < T; t++) {
        int64_t lo = t * chunk, hi = lo + chunk < n ? lo + chunk : n;
        part[t] = lo < hi ? dot8_1(a + lo, b + lo, hi - lo) : 0.0;
    }
    double tot = 0.0;
    for (int t = 0; t < T; t++) tot += part[t];
    return tot;
}
static inline void axpy(double a, const double *x, double *y, int64_t n) {
    if (g_threads <= 1) {
        for (int64_t i = 0; i < n; i++) y[i] = __builtin_fma(a, x[i], y[i]);
        return;
    }
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int64_t i = 0; i < n; i++) y[i] = __builtin_fma(a, x[i], y[i]);
}

static void iter_ref(ora_t *h) {
    const int64_t N = h->N;
    const int64_t it = h->iter + 1;
    rng_t r;
    /* samplers.jl:32-35 -> functions.jl:523-525 */
    rng_seed(&r, h->seed, h->chain, it, KIND_VARE_CHI2, 0);
    double varE = (h->e_df * h->e_scale + dot8(h->ycorr, h->ycorr, N)) / rng_chisq(&r, h->e_df + (double)N);
    h->varE = varE;
    double iVarE = 1.0 / varE;
    /* samplers.jl:39-41 -> functions.jl:41-47 (single column of ones) */
    if (h->intercept) {
        double s = 0.0;
        for (int64_t i = 0; i < N; i++) { h->ycorr[i] += h->b; }
        double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; int64_t i = 0;
        for (; i + 8 <= N; i += 8) for (int l = 0; l < 8; l++) acc[l] += h->ycorr[i + l];
        for (; i < N; i++) s += h->ycorr[i];
        s += ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
        double rhs = s * iVarE + 0.0;
        double lhs = (double)N * iVarE + 0.0;
        double mean = rhs / lhs;
        rng_seed(&r, h->seed, h->chain, it, KIND_FIXED_NORMAL, 0);
        h->b = mean + sqrt(1.0 / lhs) * rng_normal(&r);
        for (int64_t k = 0; k < N; k++) h->ycorr[k] -= h->b;
    }
    fixed_ref(h, it, iVarE);   /* the other fixed-effect sets (samplers.jl:39-41) */
    /* samplers.jl:50-53 */
    for (int si = 0; si < h->nsets; si++) {
        oset_t *S = &h->sets[si];
        double *vb = h->varBeta + S->vb_off;
        if (S->method == METHOD_PR) {
            /* functions.jl:118-137 */
            for (int64_t rg = 0; rg < S->nreg; rg++) {
                double iVarBeta = 1.0 / vb[rg];
                double ssq = 0.0;
                for (int64_t l = S->reg_start[rg]; l < S->reg_stop[rg]; l++) {
                    int64_t j = S->col0 + l;
                    const double *col = h->data + j * N, *mp = h->Mp + j * N;
                    axpy(h->beta[j], col, h->ycorr, N);                                   /* :128 */
                    double rhs = dot8(mp, h->ycorr, N) * iVarE + h->rhs0[j];              /* :129 */
                    double lhs = h->mpm[j] * iVarE + h->lhs0[j] + iVarBeta;               /* :130 */
                    double mean = rhs / lhs;                                              /* :131 */
                    rng_seed(&r, h->seed, h->chain, it, KIND_BETA_NORMAL, ((uint64_t)si << 40) | (uint64_t)l);
                    h->beta[j] = mean + sqrt(1.0 / lhs) * rng_normal(&r);                 /* :132, :493-495 */
                    axpy(-1.0 * h->beta[j], col, h->ycorr, N);                            /* :133 */
                    ssq = __builtin_fma(h->beta[j], h->beta[j], ssq);
                }
                rng_seed(&r, h->seed, h->chain, it, KIND_REGION_CHI2, ((uint64_t)si << 40) | (uint64_t)rg);
                double n_r = (double)(S->reg_stop[rg] - S->reg_start[rg]);
                vb[rg] = (S->scale * S->df + ssq) / rng_chisq(&r, S->df + n_r);         /* :135, :509-511 */
            }
        } else if (S->method == METHOD_T) {
            /* functions.jl:140-154, the Tuple method, line by line */
            const int k = S->tk;
            for (int64_t rg = 0; rg < S->nreg; rg++) {
                double *vbm = vb + rg * k * k;
                double invB[KMAX * KMAX], Sb[KMAX * KMAX];
                if (t_spd_inv(vbm, k, invB)) { snprintf(h->err, 256, "tuple: varBeta not positive definite"); return; }   /* :143 */
                for (int a = 0; a < k * k; a++) Sb[a] = 0.0;
                for (int64_t l = S->reg_start[rg]; l < S->reg_stop[rg]; l++) {
                    const double *col[KMAX]; double bj[KMAX];
                    for (int m = 0; m < k; m++) { col[m] = h->data + tcol(S, l, m) * N; bj[m] = h->beta[tcol(S, l, m)]; }
                    for (int64_t i = 0; i < N; i++) { double t = 0.0; for (int m = 0; m < k; m++) t = __builtin_fma(col[m][i], bj[m], t); h->ycorr[i] += t; }  /* :145 */
                    double RHS[KMAX], LHS[KMAX * KMAX], invLHS[KMAX * KMAX], Lc[KMAX * KMAX], mean[KMAX], z[KMAX];
                    for (int m = 0; m < k; m++) RHS[m] = dot8(h->Mp + tcol(S, l, m) * N, h->ycorr, N) / varE;                 /* :146 */
                    for (int a = 0; a < k * k; a++) LHS[a] = S->tmpm[l * k * k + a] / varE + invB[a];                          /* :147 */
                    if (t_spd_inv(LHS, k, invLHS) || t_chol1(invLHS, k, Lc)) { snprintf(h->err, 256, "tuple: LHS not positive definite"); return; }
                    for (int a = 0; a < k; a++) { double t = 0.0; for (int b = 0; b < k; b++) t += invLHS[a * k + b] * RHS[b]; mean[a] = t; }  /* :148 */
                    for (int m = 0; m < k; m++) {
                        rng_seed(&r, h->seed, h->chain, it, KIND_BETA_NORMAL, ((uint64_t)si << 40) | (uint64_t)(l * k + m));
                        z[m] = rng_normal(&r);
                    }
                    for (int a = 0; a < k; a++) {                                                                             /* :149 MvNormal(mean, invLHS) = mean + L z */
                        double t = mean[a];
                        for (int b = 0; b <= a; b++) t += Lc[a * k + b] * z[b];
                        bj[a] = t; h->beta[tcol(S, l, a)] = t;
                    }
                    for (int64_t i = 0; i < N; i++) { double t = 0.0; for (int m = 0; m < k; m++) t = __builtin_fma(col[m][i], bj[m], t); h->ycorr[i] -= t; }  /* :150 */
                    for (int a = 0; a < k; a++) for (int b = 0; b < k; b++) Sb[a * k + b] += bj[a] * bj[b];                   /* :514 */
                }
                double Psi[KMAX * KMAX];
                for (int a = 0; a < k * k; a++) Psi[a] = S->tscale[a] + Sb[a];
                const double nu = S->df + (double)(S->reg_stop[rg] - S->reg_start[rg]);
                if (t_inverse_wishart(h->seed, h->chain, it, si, rg, nu, Psi, k, vbm)) { snprintf(h->err, 256, "tuple: inverse Wishart failed"); return; }  /* :152, :513-516 */
            }
        } else if (S->method == METHOD_C) {
            /* functions.jl:197-235: BayesC = BayesB's inclusion step with ONE variance for the whole set, redrawn after the
               sweep from all effects (excluded ones are 0) with nLoci included loci (:231, :509-511).  Note the
               reference's rhs omits M.rhs here (:220, commented out) while lhs keeps M.lhs (:221). */
            int64_t nLoci = 0;
            const double iVarBeta = 1.0 / vb[0];                                          /* :205 */
            double ssq = 0.0;
            for (int64_t l = 0; l < S->ncol; l++) {
                int64_t j = S->col0 + l;
                const double *col = h->data + j * N, *mp = h->Mp + j * N;
                axpy(h->beta[j], col, h->ycorr, N);                                       /* :208 */
                double rrr = dot8(col, h->ycorr, N);                                      /* :209 */
                double v0 = h->mpm[j] * varE;                                             /* :210 */
                double v1 = (h->mpm[j] * h->mpm[j]) * vb[0] + v0;                         /* :211 */
                double logDelta0 = -0.5 * (log(v0) + (rrr * rrr) / v0) + S->logPi[0];    /* :213 */
                double logDelta1 = -0.5 * (log(v1) + (rrr * rrr) / v1) + S->logPi[1];    /* :214 */
                double probDelta1 = 1.0 / (1.0 + exp(logDelta0 - logDelta1));            /* :216 */
                rng_seed(&r, h->seed, h->chain, it, KIND_B_UNIFORM, ((uint64_t)si << 40) | (uint64_t)l);
                double u = rng_uniform(&r);
                if (u < probDelta1) {                                                     /* :217 */
                    h->delta[j] = 1; nLoci++;
                    double rhs = dot8(mp, h->ycorr, N) * iVarE;                           /* :220 */
                    double lhs = h->mpm[j] * iVarE + h->lhs0[j] + iVarBeta;               /* :221 */
                    double mean = rhs / lhs;
                    rng_seed(&r, h->seed, h->chain, it, KIND_BETA_NORMAL, ((uint64_t)si << 40) | (uint64_t)l);
                    h->beta[j] = mean + sqrt(1.0 / lhs) * rng_normal(&r);                 /* :223 */
                    axpy(-1.0 * h->beta[j], col, h->ycorr, N);                            /* :224 */
                } else {
                    h->beta[j] = 0.0; h->delta[j] = 0;                                    /* :226-227 */
                }
                ssq = __builtin_fma(h->beta[j], h->beta[j], ssq);
            }
            rng_seed(&r, h->seed, h->chain, it, KIND_REGION_CHI2, ((uint64_t)si << 40));
            vb[0] = (S->scale * S->df + ssq) / rng_chisq(&r, S->df + (double)nLoci);     /* :231 */
            if (S->estPi) {                                                               /* :232-236 */
                rng_seed(&r, h->seed, h->chain, it, KIND_PI_BETA, (uint64_t)si);
                double piIn = rng_beta(&r, (double)nLoci + 1.0, (double)(S->ncol - nLoci) + 1.0);
                S->piHat[0] = 1.0 - piIn; S->piHat[1] = piIn;
                S->logPi[0] = log(1.0 - piIn); S->logPi[1] = log(piIn);
            }
        } else if (S->method == METHOD_R) {
            /* functions.jl:238-289.  The class search `findfirst(x->x>=rand(), cumProbs)` (:261) draws a FRESH uniform for every
               comparison -- a biased categorical draw for more than two classes; reproduced, with one keyed uniform per
               (locus, comparison).  If no comparison succeeds (cumProbs[end] rounded below a uniform, or NaN probabilities
               after an overflow of exp) the reference fails with an indexing error; here the last class is taken. */
            const int K = S->K;
            int64_t nLoci[RMAX] = {0};
            int64_t nNonZero = 0;
            double varc[RMAX];
            for (int v = 0; v < K; v++) varc[v] = vb[0] * S->vcls[v];                             /* :244 */
            double sumS = 0.0;
            for (int64_t l = 0; l < S->ncol; l++) {
                int64_t j = S->col0 + l;
                const double *col = h->data + j * N, *mp = h->Mp + j * N;
                axpy(h->beta[j], col, h->ycorr, N);                                               /* :249 */
                double rhs = dot8(mp, h->ycorr, N) * iVarE + h->rhs0[j];                          /* :250 */
                double lhs[RMAX], ExpLogL[RMAX], sum = 0.0;
                for (int v = 0; v < K; v++) {
                    lhs[v] = (varc[v] == 0.0) ? 0.0 : h->mpm[j] * iVarE + h->lhs0[j] + 1.0 / varc[v];                        /* :254 */
                    double logLc = (varc[v] == 0.0) ? S->logpic[v] : -0.5 * (log(varc[v] * lhs[v]) - ((rhs * rhs) / lhs[v])) + S->logpic[v]; /* :255 */
                    ExpLogL[v] = exp(logLc);                                                      /* :256 */
                    sum += ExpLogL[v];
                }
                int cls = K - 1;
                double cum = 0.0;
                for (int v = 0; v < K; v++) {
                    cum += ExpLogL[v] / sum;                                                      /* :259-260 */
                    rng_seed(&r, h->seed, h->chain, it, KIND_R_UNIFORM, ((uint64_t)si << 40) | ((uint64_t)l << 3) | (uint64_t)v);
                    if (cum >= rng_uniform(&r)) { cls = v; break; }                               /* :261 */
                }
                h->delta[j] = cls + 1;                                                            /* :262, classes count from 1 */
                nLoci[cls]++;
                if (varc[cls] != 0.0) {                                                           /* :265 */
                    nNonZero++;
                    double mean = rhs / lhs[cls];
                    rng_seed(&r, h->seed, h->chain, it, KIND_BETA_NORMAL, ((uint64_t)si << 40) | (uint64_t)l);
                    h->beta[j] = mean + sqrt(1.0 / lhs[cls]) * rng_normal(&r);                    /* :268 */
                    axpy(-1.0 * h->beta[j], col, h->ycorr, N);                                    /* :270 */
                    sumS += (h->beta[j] * h->beta[j]) / S->vcls[cls];                             /* :272-273 */
                } else h->beta[j] = 0.0;                                                          /* :275 */
            }
            rng_seed(&r, h->seed, h->chain, it, KIND_REGION_CHI2, ((uint64_t)si << 40));
            vb[0] = (S->scale * S->df + sumS) / rng_chisq(&r, S->df + (double)nNonZero);          /* :281, :518-520 */
            if (S->estPi) {                                                                       /* :284-288, :536-538 */
                double g[RMAX], gs = 0.0;
                for (int v = 0; v < K; v++) {
                    rng_seed(&r, h->seed, h->chain, it, KIND_R_DIRICHLET, ((uint64_t)si << 40) | (uint64_t)v);
                    g[v] = rng_gamma(&r, (double)nLoci[v] + 1.0);
                    gs += g[v];
                }
                for (int v = 0; v < K; v++) { S->pic[v] = g[v] / gs; S->logpic[v] = log(S->pic[v]); }
            }
        } else {
            /* functions.jl:157-195; one region per locus (mme.jl:356-358) */
            int64_t nLoci = 0;
            for (int64_t l = 0; l < S->ncol; l++) {
                int64_t j = S->col0 + l;
                const double *col = h->data + j * N, *mp = h->Mp + j * N;
                double iVarBeta = 1.0 / vb[l];                                            /* :165 */
                axpy(h->beta[j], col, h->ycorr, N);                                       /* :167 */
                double rrr = dot8(col, h->ycorr, N);                                      /* :168 */
                double v0 = h->mpm[j] * varE;                                             /* :169 */
                double v1 = (h->mpm[j] * h->mpm[j]) * vb[l] + v0;                         /* :170 */
                double logDelta0 = -0.5 * (log(v0) + (rrr * rrr) / v0) + S->logPi[0];    /* :171 */
                double logDelta1 = -0.5 * (log(v1) + (rrr * rrr) / v1) + S->logPi[1];    /* :172 */
                double probDelta1 = 1.0 / (1.0 + exp(logDelta0 - logDelta1));            /* :173 */
                rng_seed(&r, h->seed, h->chain, it, KIND_B_UNIFORM, ((uint64_t)si << 40) | (uint64_t)l);
                double u = rng_uniform(&r);
                if (u < probDelta1) {                                                     /* :174 */
                    h->delta[j] = 1; nLoci++;
                    double rhs = dot8(mp, h->ycorr, N) * iVarE + h->rhs0[j];              /* :177 */
                    double lhs = h->mpm[j] * iVarE + h->lhs0[j] + iVarBeta;               /* :178 */
                    double mean = rhs / lhs;
                    rng_seed(&r, h->seed, h->chain, it, KIND_BETA_NORMAL, ((uint64_t)si << 40) | (uint64_t)l);
                    h->beta[j] = mean + sqrt(1.0 / lhs) * rng_normal(&r);                 /* :180 */
                    axpy(-1.0 * h->beta[j], col, h->ycorr, N);                            /* :181 */
                    rng_seed(&r, h->seed, h->chain, it, KIND_B_LOCUS_CHI2, ((uint64_t)si << 40) | (uint64_t)l);
                    vb[l] = (S->scale * S->df + h->beta[j] * h->beta[j]) / rng_chisq(&r, S->df + 1.0); /* :182 */
                } else {
                    h->beta[j] = 0.0; h->delta[j] = 0; vb[l] = 0.0;                       /* :184-186 */
                }
            }
            if (S->estPi) {                                                               /* :190-194, :531-533 */
                rng_seed(&r, h->seed, h->chain, it, KIND_PI_BETA, (uint64_t)si);
                double piIn = rng_beta(&r, (double)nLoci + 1.0, (double)(S->ncol - nLoci) + 1.0);
                S->piHat[0] = 1.0 - piIn; S->piHat[1] = piIn;
                S->logPi[0] = log(1.0 - piIn); S->logPi[1] = log(piIn);
            }
        }
    }
    h->iter = it;
}

/* ------------------------------------------------------------------ */
/* order 1: blocked order (mirrors the HIP kernels operation by op)     */
/* ------------------------------------------------------------------ */
static double wave_butterfly(double *v) { /* 64 lanes, xor 32..1 */
    double t[64];
    for (int off = 32; off >= 1; off >>= 1) {
        for (int l = 0; l < 64; l++) t[l] = v[l] + v[l ^ off];
        memcpy(v, t, sizeof(t));
    }
    return v[0];
}

static double wave_butterfly_up(double *v) { /* 64 lanes, xor 1..32 (compact storage: sum_j m_j dlt_j) */
    double t[64];
    for (int off = 1; off <= 32; off <<= 1) {
        for (int l = 0; l < 64; l++) t[l] = v[l] + v[l ^ off];
        memcpy(v, t, sizeof(t));
    }
    return v[0];
}

/* exponent of the fixed-point scale (csrc/ngp_common.h, fx_exponent): E = floor(ilogb(m) / 2) + 5 for finite m > 0, else 5; clamped */
static int fx_exponent(double m) {
    union { double d; uint64_t u; } v; v.d = m;
    const int ef = (int)((v.u >> 52) & 0x7ff);
    int e2 = 0;
    if (m > 0.0 && ef != 0x7ff) e2 = ef - 1023;
    int e = (e2 >> 1) + 5;
    if (e < -900) e = -900;
    if (e > 900) e = 900;
    return e;
}
static void iter_blocked(ora_t *h) {
    const int64_t N = h->N, R = h->R, S = h->S, L = R * S, NBLK = h->NBLK;
    const int64_t it = h->iter + 1;
    rng_t r;
    /* ---- iter_head: 1024-thread strided reduce, wave butterflies, 16 wave sums */
    double wyy[16], wsy[16];
    for (int wv = 0; wv < 16; wv++) {
        double lyy[64], lsy[64];
        for (int l = 0; l < 64; l++) {
            double ayy = 0.0, asy = 0.0;
            for (int64_t i = wv * 64 + l; i < L; i += 1024) { double v = h->ycorr[i]; ayy = __builtin_fma(v, v, ayy); asy = asy + v; }
            lyy[l] = ayy; lsy[l] = asy;
        }
        wyy[wv] = wave_butterfly(lyy); wsy[wv] = wave_butterfly(lsy);
    }
    double yy = wyy[0], sy = wsy[0];
    for (int wv = 1; wv < 16; wv++) { yy = yy + wyy[wv]; sy = sy + wsy[wv]; }
    /* fixed-point scale of this iteration's X_t'ycorr accumulators (csrc/ngp_common.h): every shard
       partial and every far look-ahead term is scaled by 2^(52 - E), rounded to the nearest integer and added as an integer -- an
       order-free sum.  2^E > 16 sqrt(max_j x_j'x_j * ycorr'ycorr), the norms as they stand at the head of the iteration. */
    double mpm_max = 0.0;
    for (int64_t k = 0; k < h->Ppad; k++) if (h->mpm[k] > mpm_max) mpm_max = h->mpm[k];
    const int fe = fx_exponent(mpm_max * yy);
    const double fxs = ldexp(1.0, 52 - fe), fxi = ldexp(1.0, fe - 52);
    rng_seed(&r, h->seed, h->chain, it, KIND_VARE_CHI2, 0);
    double chi = rng_chisq(&r, h->e_df + (double)N);
    double t = h->e_df * h->e_scale; t = t + yy;
    double varE = t / chi;
    double iVarE = 1.0 / varE;
    h->varE = varE;
    if (h->intercept) {
        double Nd = (double)N;
        double tb = Nd * h->b; double sb = sy + tb;
        double rhs = sb * iVarE; double lhs = Nd * iVarE;
        double mean = rhs / lhs; double sd = sqrt(1.0 / lhs);
        rng_seed(&r, h->seed, h->chain, it, KIND_FIXED_NORMAL, 0);
        double z = rng_normal(&r);
        double tz = sd * z; double bn = mean + tz;
        double db = bn - h->b;
        h->b = bn;
        for (int64_t i = 0; i < N; i++) h->ycorr[i] = h->ycorr[i] - db;
    }
    fixed_blocked(h, it, iVarE);
    /* ---- set_prep: per-locus coefficients */
    for (int64_t k = 0; k < h->Ppad; k++) { h->c[k] = 0.0; h->w[k] = 0.0; h->q[k] = -1.0; h->T[k] = 1.0; h->chi[k] = 1.0; }
    for (int si = 0; si < h->nsets; si++) {
        oset_t *Sx = &h->sets[si];
        double *vb = h->varBeta + Sx->vb_off;
        if (Sx->method == METHOD_T) {
            /* Tuple set: per locus the k x k conditional (functions.jl:143-149) with everything the block chain does not need to
               compute: C = iVarE inv(X_l'X_l iVarE + inv(varBeta_r)) (row m for column (l, m)) and W = L z - beta, L = chol(inv(LHS)) */
            const int kk = Sx->tk; const size_t PP = (size_t)h->Ppad;
            for (int64_t rg = 0; rg < Sx->nreg; rg++) {
                double invB[KMAX * KMAX];
                if (t_spd_inv(vb + rg * kk * kk, kk, invB)) { snprintf(h->err, 256, "tuple: varBeta not positive definite"); return; }
                for (int64_t l = Sx->reg_start[rg]; l < Sx->reg_stop[rg]; l++) {
                    double LHS[KMAX * KMAX], invLHS[KMAX * KMAX], Lc[KMAX * KMAX], z[KMAX];
                    for (int a = 0; a < kk; a++)
                        for (int b = 0; b < kk; b++) { double t1 = h->tupg[(size_t)b * PP + tcol(Sx, l, a)] * iVarE; LHS[a * kk + b] = t1 + invB[a * kk + b]; }
                    if (t_spd_inv(LHS, kk, invLHS) || t_chol1(invLHS, kk, Lc)) { snprintf(h->err, 256, "tuple: LHS not positive definite"); return; }
                    for (int m = 0; m < kk; m++) {
                        rng_seed(&r, h->seed, h->chain, it, KIND_BETA_NORMAL, ((uint64_t)si << 40) | (uint64_t)(l * kk + m));
                        z[m] = rng_normal(&r);
                    }
                    for (int m = 0; m < kk; m++) {
                        const int64_t c = tcol(Sx, l, m);
                        for (int b = 0; b < kk; b++) h->tupc[(size_t)b * PP + c] = iVarE * invLHS[m * kk + b];
                        double acc = Lc[m * kk + 0] * z[0];
                        for (int b = 1; b <= m; b++) acc = __builtin_fma(Lc[m * kk + b], z[b], acc);
                        h->w[c] = acc - h->beta[c];
                    }
                }
            }
            continue;
        }
        for (int64_t rg = 0; rg < Sx->nreg; rg++)
            for (int64_t l = Sx->reg_start[rg]; l < Sx->reg_stop[rg]; l++) {
                int64_t k = Sx->col0 + l;
                if (Sx->method == METHOD_R) { /* per-class coefficients; the class is chosen inside the block chain */
                    const size_t PP = (size_t)h->Ppad;
                    double *rq = h->rcls, *ra = h->rcls + RMAX * PP, *rt = h->rcls + 2 * RMAX * PP, *ru = h->rcls + 3 * RMAX * PP;
                    double t1r = h->mpm[k] * iVarE; double t2r = t1r + h->lhs0[k];
                    rng_seed(&r, h->seed, h->chain, it, KIND_BETA_NORMAL, ((uint64_t)si << 40) | (uint64_t)l);
                    double zr = rng_normal(&r);
                    for (int v = 0; v < Sx->K; v++) {
                        double varc = vb[0] * Sx->vcls[v];
                        rng_seed(&r, h->seed, h->chain, it, KIND_R_UNIFORM, ((uint64_t)si << 40) | ((uint64_t)l << 3) | (uint64_t)v);
                        ru[v * PP + k] = rng_uniform(&r);
                        if (varc == 0.0) { rq[v * PP + k] = 0.0; ra[v * PP + k] = Sx->logpic[v]; rt[v * PP + k] = 0.0; }
                        else {
                            double iv = 1.0 / varc; double lhsv = t2r + iv; double ilhs = 1.0 / lhsv;
                            double prod = varc * lhsv; double lg = det_log(prod); double hl = 0.5 * lg;
                            double sd = sqrt(ilhs);
                            rq[v * PP + k] = ilhs; ra[v * PP + k] = Sx->logpic[v] - hl; rt[v * PP + k] = sd * zr;
                        }
                    }
                    continue; /* c = w = 0, q = -1 stay (unused by the r-form chain for this lane) */
                }
                double vbk = (Sx->method == METHOD_B) ? vb[l] : vb[rg];
                double mpm = h->mpm[k];
                double t1 = mpm * iVarE; double t2 = t1 + h->lhs0[k];
                double ivb = 1.0 / vbk; double lhs = t2 + ivb;
                double ilhs = 1.0 / lhs;
                double c = iVarE * ilhs; double s = sqrt(ilhs);
                rng_seed(&r, h->seed, h->chain, it, KIND_BETA_NORMAL, ((uint64_t)si << 40) | (uint64_t)l);
                double z = rng_normal(&r);
                double sz = s * z;
                double tw = (Sx->method == METHOD_C) ? 0.0 : h->rhs0[k] * ilhs; /* BayesC drops M.rhs (functions.jl:220) */
                tw = tw + sz;
                h->c[k] = c; h->w[k] = tw - h->beta[k];
                if (Sx->method == METHOD_B || Sx->method == METHOD_C) {
                    double v0 = mpm * varE;
                    double m2 = mpm * mpm; m2 = m2 * vbk; double v1 = m2 + v0;
                    double i1 = 1.0 / v1, i0 = 1.0 / v0; double dq = i1 - i0;
                    double qq = 0.5 * dq;
                    rng_seed(&r, h->seed, h->chain, it, KIND_B_UNIFORM, ((uint64_t)si << 40) | (uint64_t)l);
                    double u = rng_uniform(&r);
                    double om = 1.0 - u;
                    double Lu = det_log(om) - det_log(u);
                    double dl = det_log(v1) - det_log(v0); dl = 0.5 * dl;
                    double T = Lu - dl; double lp = Sx->logPi[0] - Sx->logPi[1];
                    double TT = T - lp;
                    /* inclusion test rr*q < T rearranged so the serial chain only compares |r| with a threshold:
                       q < 0: r^2 > T/q  <=>  |r| > sqrt(T/q) (always true when T/q < 0);  q == 0: 0 < T */
                    double st;
                    if (qq < 0.0) { double thr2 = TT / qq; st = (thr2 < 0.0) ? -1.0 : sqrt(thr2); }
                    else st = (0.0 < TT) ? -1.0 : INFINITY;
                    /* threshold on f = c r instead of r (c > 0 when finite): thr = st c; -1 always, inf never */
                    h->q[k] = (st < 0.0) ? -1.0 : (isinf(st) ? INFINITY : st * c); h->T[k] = TT;
                    if (Sx->method == METHOD_B) {
                        rng_seed(&r, h->seed, h->chain, it, KIND_B_LOCUS_CHI2, ((uint64_t)si << 40) | (uint64_t)l);
                        h->chi[k] = rng_chisq(&r, Sx->df + 1.0);
                    }
                }
            }
    }
    /* ---- block sweep with lag D: the GEMV of block t sees ycorr with the updates of blocks <= t-D;
       the missing updates enter through the cross Gram blocks (DESIGN.md "Blocked sweep arithmetic") */
    const size_t tile = (size_t)R * BLK;
    const int64_t D = h->D;
    double *part = (double *)malloc(sizeof(double) * S * BLK);
    double *hist = (double *)calloc((size_t)NBLK * BLK, sizeof(double)); /* dlt of every block */
    for (int64_t tb = 0; tb < NBLK + D; tb++) {
        if (tb >= D && h->storage == 1) { /* compact storage: y_i -= (sum_j g_ij dlt_j - sum_j m_j dlt_j), valid rows only */
            const int64_t a = tb - D;
            double md[64];
            for (int j = 0; j < BLK; j++) md[j] = h->mean[a * BLK + j] * hist[a * BLK + j];
            const double cm = wave_butterfly_up(md);
            for (int64_t s = 0; s < S; s++) {
                const uint8_t *tl = h->tiles8 + ((size_t)s * NBLK + a) * tile;
                double *ys = h->ycorr + s * R;
                for (int64_t i = 0; i < R && s * R + i < N; i++) {
                    double p8[8];
                    for (int c = 0; c < 8; c++) {
                        double p = 0.0;
                        for (int jj = 0; jj < 8; jj++) p = __builtin_fma((double)tl[(8 * c + jj) * R + i], hist[a * BLK + 8 * c + jj], p);
                        p8[c] = p;
                    }
                    double T = ((p8[0] + p8[1]) + (p8[2] + p8[3])) + ((p8[4] + p8[5]) + (p8[6] + p8[7]));
                    double Tc = T - cm;
                    ys[i] = ys[i] - Tc;
                }
            }
        } else
        if (tb >= D) { /* y update with block a = tb - D: per row, columns ascending */
            const int64_t a = tb - D;
            for (int64_t s = 0; s < S; s++) {
                const float *tl = h->tiles + ((size_t)s * NBLK + a) * tile;
                double *ys = h->ycorr + s * R;
                for (int64_t i = 0; i < R; i++) {
                    /* y_i -= sum_j x_ij dlt_j: eight chains of eight columns, then a fixed pairwise tree */
                    double p8[8];
                    for (int c = 0; c < 8; c++) {
                        double p = 0.0;
                        for (int jj = 0; jj < 8; jj++) p = __builtin_fma((double)tl[(8 * c + jj) * R + i], hist[a * BLK + 8 * c + jj], p);
                        p8[c] = p;
                    }
                    double T = ((p8[0] + p8[1]) + (p8[2] + p8[3])) + ((p8[4] + p8[5]) + (p8[6] + p8[7]));
                    ys[i] = ys[i] - T;
                }
            }
        }
        if (tb >= NBLK) continue;
        const int64_t k0 = tb * BLK;
        /* compact storage: 7 chains over strided units of 16 rows (lane = column), the shard's sum of y in task order, then
           partial = A - m_j Sy */
        if (h->storage == 1) for (int64_t s = 0; s < S; s++) {
            const uint8_t *tl = h->tiles8 + ((size_t)s * NBLK + tb) * tile;
            const double *ys = h->ycorr + s * R;
            const int64_t NU = R / 16;
            double syw[7];
            for (int wv = 0; wv < 7; wv++) {
                const int64_t nu = (NU - wv + 6) / 7;   /* units of this wave */
                double v[8] = {0, 0, 0, 0, 0, 0, 0, 0}; /* lane slots: task tau = slot + 8 pass -> unit wv + 7 (tau >> 2), quad tau & 3 */
                for (int64_t tau = 0; tau < 4 * nu; tau++) {
                    const int64_t i0 = 16 * (wv + 7 * (tau >> 2)) + 4 * (tau & 3);
                    const double q4 = (ys[i0] + ys[i0 + 1]) + (ys[i0 + 2] + ys[i0 + 3]);
                    v[tau & 7] = (tau < 8) ? q4 : v[tau & 7] + q4;
                }
                syw[wv] = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
            }
            const double Sy = ((syw[0] + syw[1]) + (syw[2] + syw[3])) + ((syw[4] + syw[5]) + syw[6]);
            for (int j = 0; j < BLK; j++) {
                double a7[7];
                for (int wv = 0; wv < 7; wv++) {
                    double acc = 0.0;
                    for (int64_t u16 = wv; u16 < NU; u16 += 7)
                        for (int e = 0; e < 16; e++) { int64_t i = 16 * u16 + e; acc = __builtin_fma((double)tl[j * R + i], ys[i], acc); }
                    a7[wv] = acc;
                }
                const double A = ((a7[0] + a7[1]) + (a7[2] + a7[3])) + ((a7[4] + a7[5]) + a7[6]);
                const double ms = h->mean[k0 + j] * Sy;
                part[s * BLK + j] = A - ms;
            }
        }
        else
        /* GEMV partials: 8 chains over strided row quads, lane = column */
        for (int64_t s = 0; s < S; s++) {
            const float *tl = h->tiles + ((size_t)s * NBLK + tb) * tile;
            const double *ys = h->ycorr + s * R;
            for (int j = 0; j < BLK; j++) {
                double a8[8];
                const int64_t nch = h->nchain;
                for (int wv = 0; wv < nch; wv++) { /* chain wv: row quads wv, wv+nch, ... */
                    double acc = 0.0;
                    for (int64_t qd = wv; qd < R / 4; qd += nch)
                        for (int e = 0; e < 4; e++) { int64_t i = 4 * qd + e; acc = __builtin_fma((double)tl[j * R + i], ys[i], acc); }
                    a8[wv] = acc;
                }
                part[s * BLK + j] = (nch == 8) ? ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]))
                                               : ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + a8[6]);
            }
        }
        double rr[BLK], dlt[BLK]; int inc[BLK];
        const double *G = h->gram + (size_t)tb * BLK * BLK;
        int has_r = 0, has_t = -1;
        int lane_set[BLK];
        for (int j = 0; j < BLK; j++) {
            lane_set[j] = -1;
            for (int si = 0; si < h->nsets; si++)
                if (k0 + j >= h->sets[si].col0 && k0 + j < h->sets[si].col0 + h->sets[si].ncol) lane_set[j] = si;
            if (lane_set[j] >= 0 && h->sets[lane_set[j]].method == METHOD_R) has_r = 1;
            if (lane_set[j] >= 0 && h->sets[lane_set[j]].method == METHOD_T) has_t = lane_set[j];
        }
        /* linear block: every lane BayesPR or unowned */
        int raw_linear = 1;
        for (int j = 0; j < BLK; j++)
            if (lane_set[j] >= 0 && h->sets[lane_set[j]].method != METHOD_PR) raw_linear = 0;
        for (int j = 0; j < BLK; j++) {
            /* the shard partials as fixed-point terms: integer addition, no order */
            long long qsum = 0;
            for (int64_t s = 0; s < S; s++) qsum += llrint(part[s * BLK + j] * fxs);
            /* look-ahead corrections v_d = G[tb, tb-d] dlt_{tb-d} of the blocks whose update the GEMV has not seen.
               Far lags d = h->near+1 .. D-1 are folded into the group sums (the reducer workgroups compute them):
               lag d goes to group (d-h->near-1) mod NG, ascending d.  Lags h->near .. 1 stay with the sampler. */
            double vd[17]; int hv[17];
            for (int64_t d = 1; d < D; d++) {
                hv[d] = (tb - d >= 0);
                if (!hv[d]) continue;
                const double *Gx = h->gramx + (((size_t)tb * D + d) * BLK) * BLK;
                const double *da = hist + (tb - d) * BLK;
                double s4[4] = {0, 0, 0, 0};
                for (int k = 0; k < BLK; k++) s4[k & 3] = __builtin_fma(Gx[k * BLK + j], da[k], s4[k & 3]);
                vd[d] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
            }
            for (int64_t d = h->near + 1; d < D; d++)
                if (hv[d]) qsum += llrint((-vd[d]) * fxs);   /* far lags: one more term each (the "reducer" workgroups) */
            double tot = (double)qsum * fxi;
            {   /* near lags stay with the sampler: cor = (((v_near + ...) + v_2) + v_1) over the terms that exist */
                double c = 0.0; int have = 0;
                for (int64_t d = (D - 1 < h->near ? D - 1 : h->near); d >= 1; d--)
                    if (hv[d]) { c = have ? c + vd[d] : vd[d]; have = 1; }
                if (have) tot = tot - c;
            }
            rr[j] = __builtin_fma(G[j * BLK + j], h->beta[k0 + j], tot);
            if (has_t >= 0) {   /* tuple block: x_m'(ycorr + X_l beta_l) -- the add-back of ALL k effects of the locus, components in order */
                const oset_t *Tx = &h->sets[has_t]; const int kk = Tx->tk;
                rr[j] = tot;
                if (j < (BLK / kk) * kk)
                    for (int b = 0; b < kk; b++) rr[j] = __builtin_fma(h->tupg[(size_t)b * h->Ppad + k0 + j], h->beta[k0 + (j / kk) * kk + b], rr[j]);
            }
        }
        if (has_t >= 0) {
            /* Tuple block (functions.jl:144-151 in 64-column space): the k effects of a locus are drawn in ONE step from the r of its
               k columns -- dlt_m = W_m + sum_b C[m][b] r_b -- and then applied, component after component, to the columns of the
               later loci of the block. */
            const oset_t *Tx = &h->sets[has_t]; const int kk = Tx->tk; const size_t PP = (size_t)h->Ppad;
            const int64_t Lb = BLK / kk, lb0 = (tb - Tx->col0 / BLK) * Lb;
            /* scaled form, as the Symbol path's chain: lane j carries e_j = W_j + sum_b C_j[b] r_b, its candidate dlt; a finished
               column s changes it by H_j(s) dlt_s, H_j(s) = -(sum_b C_j[b] G[s][column b of j's locus]) -- for the columns of LATER
               loci only (the k effects of a locus are drawn together: its own columns do not see each other's dlt) */
            double cand[BLK];
            int64_t nvalid = (Tx->nloc - lb0 < Lb ? Tx->nloc - lb0 : Lb) * kk;   /* used lanes of this block */
            for (int j = 0; j < BLK; j++) {
                cand[j] = 0.0;
                if (j >= nvalid) continue;
                const int gb = (j / kk) * kk; const int64_t c = k0 + j;
                double e = h->w[c];
                for (int b = 0; b < kk; b++) e = __builtin_fma(rr[gb + b], h->tupc[(size_t)b * PP + c], e);
                cand[j] = e;
            }
            if (h->tform) {
                /* inverse form (k_tinv with K = kk): L dlt = e0, L = I - H'; column i of T = inv(L) by forward substitution, row after
                   row: x_m = -(C_m[0] acc_g (+) fma(C_m[b], acc_{g+b})), g = first column of m's locus (0 for the unused last column
                   of a 3-set block); acc_j += G[m][j] x_m for the columns j of later loci.  Then dlt = T e0, four accumulators. */
                double Tm[BLK][BLK];
                for (int i = 0; i < BLK; i++) {
                    double acc[BLK], x[BLK];
                    for (int j = 0; j < BLK; j++) acc[j] = 0.0;
                    for (int m = 0; m < BLK; m++) {
                        double xm;
                        if (m < i) xm = 0.0;
                        else if (m == i) xm = 1.0;
                        else {
                            const int g = (m / kk) * kk;
                            double t = 0.0;
                            if (g + kk <= BLK) {
                                t = h->tupc[k0 + m] * acc[g];
                                for (int b = 1; b < kk; b++) t = __builtin_fma(h->tupc[(size_t)b * PP + k0 + m], acc[g + b], t);
                            }
                            xm = -t;
                        }
                        x[m] = xm;
                        for (int j = m + 1; j < BLK; j++)
                            if ((j / kk) * kk > m) acc[j] = __builtin_fma(G[j * BLK + m], xm, acc[j]);
                    }
                    for (int m = 0; m < BLK; m++) Tm[m][i] = x[m];
                }
                double dl[BLK];
                for (int j = 0; j < BLK; j++) {
                    double s4[4] = {0, 0, 0, 0};
                    for (int i = 0; i < BLK; i++) s4[i & 3] = __builtin_fma(Tm[j][i], cand[i], s4[i & 3]);
                    dl[j] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
                }
                for (int j = 0; j < BLK; j++) cand[j] = dl[j];
            } else
            for (int sl = 0; sl < nvalid; sl++) {
                const double dk = cand[sl];
                for (int j = (sl / kk + 1) * kk; j < nvalid; j++) {
                    const int gb = (j / kk) * kk; const int64_t c = k0 + j;
                    double t = h->tupc[c] * G[(gb) * BLK + sl];
                    for (int b = 1; b < kk; b++) t = __builtin_fma(h->tupc[(size_t)b * PP + c], G[(gb + b) * BLK + sl], t);
                    const double Hjs = -t;
                    cand[j] = __builtin_fma(Hjs, dk, cand[j]);
                }
            }
            for (int k = 0; k < BLK; k++) {
                hist[tb * BLK + k] = cand[k];
                h->beta[k0 + k] = h->beta[k0 + k] + cand[k];
                h->delta[k0 + k] = 1;
            }
            continue;
        }
        /* Blocks that hold a BayesR locus: the chain runs on r itself ("r-form").  All lanes form their candidate dlt from the
           current r; the first lane at or behind the cursor whose candidate is not zero takes its step (r_j -= G_jk dlt_k for
           the later lanes) and the candidates behind it are formed again; lanes whose candidate is zero change nothing and
           are passed over.  Lanes of other methods in such a block follow their own rule, written in r. */
        if (has_r) {
            const size_t PP = (size_t)h->Ppad;
            const double *rq = h->rcls, *ra = h->rcls + RMAX * PP, *rt = h->rcls + 2 * RMAX * PP, *ru = h->rcls + 3 * RMAX * PP;
            double cand[BLK]; int cls[BLK];
            int kstart = 0;
            for (;;) {
                for (int j = kstart; j < BLK; j++) {
                    const int64_t k = k0 + j;
                    const double bo = h->beta[k];
                    const int si = lane_set[j];
                    if (si >= 0 && h->sets[si].method == METHOD_R) {
                        const oset_t *Sx = &h->sets[si];
                        double t = rr[j] * iVarE; double rhs = t + h->rhs0[k]; double s2 = rhs * rhs; double hs = 0.5 * s2;
                        double L[RMAX], e[RMAX];
                        for (int v = 0; v < Sx->K; v++) L[v] = (rq[v * PP + k] == 0.0) ? ra[v * PP + k] : __builtin_fma(hs, rq[v * PP + k], ra[v * PP + k]);
                        double m = L[0];
                        for (int v = 1; v < Sx->K; v++) if (L[v] > m) m = L[v];
                        double Ssum = 0.0;
                        for (int v = 0; v < Sx->K; v++) { e[v] = det_exp(L[v] - m); Ssum = Ssum + e[v]; }
                        int c = Sx->K - 1; double cum = 0.0;
                        for (int v = 0; v < Sx->K; v++) { cum = cum + e[v]; double thr = ru[v * PP + k] * Ssum; if (cum >= thr) { c = v; break; } }
                        if (rq[c * PP + k] != 0.0) { double d = __builtin_fma(rhs, rq[c * PP + k], rt[c * PP + k]); cand[j] = d - bo; }
                        else cand[j] = -bo;
                        cls[j] = c + 1;
                    } else {
                        double f = rr[j] * h->c[k];
                        int in = fabs(f) > h->q[k];
                        double e1 = __builtin_fma(rr[j], h->c[k], h->w[k]);
                        cand[j] = in ? e1 : -bo;
                        cls[j] = in;
                    }
                }
                int kk = -1;
                for (int j = kstart; j < BLK; j++) if (cand[j] != 0.0) { kk = j; break; }
                if (kk < 0) break;
                for (int j = kk + 1; j < BLK; j++) rr[j] = __builtin_fma(-G[j * BLK + kk], cand[kk], rr[j]);
                kstart = kk + 1;
                if (kstart >= BLK) break;
            }
            for (int k = 0; k < BLK; k++) {
                hist[tb * BLK + k] = cand[k];
                h->beta[k0 + k] = h->beta[k0 + k] + cand[k];
                h->delta[k0 + k] = cls[k];
            }
            continue;
        }
        /* recursion in the scaled variables e_j = c_j r_j + w_j (the candidate draw) and f_j = c_j r_j (for the
           inclusion test |f_j| > thr_j = st_j c_j): one fma per step on the serial path, H_jk = -(c_j G_jk) */
        double ee[BLK], ff[BLK];
        for (int j = 0; j < BLK; j++) {
            ee[j] = __builtin_fma(rr[j], h->c[k0 + j], h->w[k0 + j]);
            ff[j] = rr[j] * h->c[k0 + j];
        }
        /* Linear block -- every lane BayesPR (always included: src/functions.jl:124-136) or unowned: the 64 steps
             dlt_k = e_k,  e_j += H_jk dlt_k (j > k)
           are the forward substitution of L dlt = e0, L = I + diag(c) strictLower(G): dlt = T e0 with T = inv(L) formed explicitly
           (k_tinv) -- the same chain in real arithmetic, a 64 x 64 product instead of 64 dependent cross-lane steps.
           Formation, column i of T (lane i of k_tinv): x_m = 0 (m < i), 1 (m = i), -(c_m acc_m) (m > i); after every x_m:
           acc_j = fma(G[m][j], x_m, acc_j) for j > m.  Application: four accumulators over i mod 4, ((s0+s1)+(s2+s3)). */
        const int linear = h->tform && raw_linear;
        if (linear) {
            double Tm[BLK][BLK];
            for (int i = 0; i < BLK; i++) {
                double acc[BLK], x[BLK];
                for (int j = 0; j < BLK; j++) acc[j] = 0.0;
                for (int m = 0; m < BLK; m++) {
                    double xm;
                    if (m < i) xm = 0.0;
                    else if (m == i) xm = 1.0;
                    else { double tcm = h->c[k0 + m] * acc[m]; xm = -tcm; }
                    x[m] = xm;
                    for (int j = m + 1; j < BLK; j++) acc[j] = __builtin_fma(G[j * BLK + m], xm, acc[j]);
                }
                for (int m = 0; m < BLK; m++) Tm[m][i] = x[m];
            }
            for (int j = 0; j < BLK; j++) {
                double s4[4] = {0, 0, 0, 0};
                for (int i = 0; i < BLK; i++) s4[i & 3] = __builtin_fma(Tm[j][i], ee[i], s4[i & 3]);
                dlt[j] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
            }
            for (int k = 0; k < BLK; k++) {
                hist[tb * BLK + k] = dlt[k];
                h->beta[k0 + k] = h->beta[k0 + k] + dlt[k];
                h->delta[k0 + k] = 1;
            }
            continue;
        }
        for (int k = 0; k < BLK; k++) {
            int in = fabs(ff[k]) > h->q[k0 + k];
            double dk = in ? ee[k] : -h->beta[k0 + k];
            dlt[k] = dk; inc[k] = in;
            for (int j = k + 1; j < BLK; j++) {
                double Hjk = -(h->c[k0 + j] * G[j * BLK + k]);
                ee[j] = __builtin_fma(Hjk, dk, ee[j]);
                ff[j] = __builtin_fma(Hjk, dk, ff[j]);
            }
        }
        for (int k = 0; k < BLK; k++) {
            hist[tb * BLK + k] = dlt[k];
            h->beta[k0 + k] = h->beta[k0 + k] + dlt[k];
            h->delta[k0 + k] = inc[k];
        }
    }
    free(part); free(hist);
    /* ---- variance components / pi */
    for (int si = 0; si < h->nsets; si++) {
        oset_t *Sx = &h->sets[si];
        double *vb = h->varBeta + Sx->vb_off;
        if (Sx->method == METHOD_T) {
            /* functions.jl:152, :513-516: Sb = B_r'B_r entry by entry in the segment pattern of the Symbol path (a wave per 256 loci,
               lane l: loci l, l+64, l+128, l+192 by fma, xor butterfly, segments in order), then the inverse-Wishart draw */
            const int kk = Sx->tk;
            for (int64_t rg = 0; rg < Sx->nreg; rg++) {
                double Sb[KMAX * KMAX], Psi[KMAX * KMAX];
                for (int a = 0; a < kk; a++)
                    for (int b = a; b < kk; b++) {
                        double tot = 0.0; int first = 1;
                        for (int64_t l0 = Sx->reg_start[rg]; l0 < Sx->reg_stop[rg]; l0 += SEG) {
                            int64_t l1 = l0 + SEG < Sx->reg_stop[rg] ? l0 + SEG : Sx->reg_stop[rg];
                            double lane[64];
                            for (int l = 0; l < 64; l++) {
                                double acc = 0.0;
                                for (int m = 0; m < 4; m++) {
                                    int64_t ll = l0 + l + 64 * m;
                                    if (ll < l1) acc = __builtin_fma(h->beta[tcol(Sx, ll, a)], h->beta[tcol(Sx, ll, b)], acc);
                                }
                                lane[l] = acc;
                            }
                            double p = wave_butterfly(lane);
                            tot = first ? p : tot + p; first = 0;
                        }
                        Sb[a * kk + b] = tot; Sb[b * kk + a] = tot;
                    }
                for (int a = 0; a < kk * kk; a++) Psi[a] = Sx->tscale[a] + Sb[a];
                const double nu = Sx->df + (double)(Sx->reg_stop[rg] - Sx->reg_start[rg]);
                if (t_inverse_wishart(h->seed, h->chain, it, si, rg, nu, Psi, kk, vb + rg * kk * kk)) { snprintf(h->err, 256, "tuple: inverse Wishart failed"); return; }
            }
            continue;
        }
        if (Sx->method == METHOD_R) {
            int64_t nL[RMAX] = {0}, nNonZero = 0;
            for (int64_t l = 0; l < Sx->ncol; l++) nL[h->delta[Sx->col0 + l] - 1]++;
            for (int v = 0; v < Sx->K; v++) if (Sx->vcls[v] != 0.0) nNonZero += nL[v];
            double tot = 0.0; int first = 1;
            for (int64_t l0 = 0; l0 < Sx->ncol; l0 += SEG) {
                int64_t l1 = l0 + SEG < Sx->ncol ? l0 + SEG : Sx->ncol;
                double lane[64];
                for (int l = 0; l < 64; l++) {
                    double a = 0.0;
                    for (int m = 0; m < 4; m++) {
                        int64_t ll = l0 + l + 64 * m;
                        if (ll < l1) {
                            double bv = h->beta[Sx->col0 + ll]; double vc = Sx->vcls[h->delta[Sx->col0 + ll] - 1];
                            if (vc != 0.0) { double b2 = bv * bv; double term = b2 / vc; a = a + term; }
                        }
                    }
                    lane[l] = a;
                }
                double p = wave_butterfly(lane);
                tot = first ? p : tot + p; first = 0;
            }
            rng_seed(&r, h->seed, h->chain, it, KIND_REGION_CHI2, ((uint64_t)si << 40));
            double ch = rng_chisq(&r, Sx->df + (double)nNonZero);
            double tt = Sx->scale * Sx->df; tt = tt + tot;
            vb[0] = tt / ch;
            if (Sx->estPi) {
                double g[RMAX], gs = 0.0;
                for (int v = 0; v < Sx->K; v++) {
                    rng_seed(&r, h->seed, h->chain, it, KIND_R_DIRICHLET, ((uint64_t)si << 40) | (uint64_t)v);
                    g[v] = rng_gamma(&r, (double)nL[v] + 1.0);
                    gs = gs + g[v];
                }
                for (int v = 0; v < Sx->K; v++) { Sx->pic[v] = g[v] / gs; Sx->logpic[v] = det_log(Sx->pic[v]); }
            }
            continue;
        }
        if (Sx->method == METHOD_PR || Sx->method == METHOD_C) {
            int64_t nLoci = 0;
            if (Sx->method == METHOD_C)
                for (int64_t l = 0; l < Sx->ncol; l++) nLoci += h->delta[Sx->col0 + l] ? 1 : 0;
            for (int64_t rg = 0; rg < Sx->nreg; rg++) {
                double tot = 0.0; int first = 1;
                for (int64_t l0 = Sx->reg_start[rg]; l0 < Sx->reg_stop[rg]; l0 += SEG) {
                    int64_t l1 = l0 + SEG < Sx->reg_stop[rg] ? l0 + SEG : Sx->reg_stop[rg];
                    /* 256-locus segment on one wave: lane l takes loci l, l+64, l+128, l+192, then the xor butterfly */
                    double lane[64];
                    for (int l = 0; l < 64; l++) {
                        double a = 0.0;
                        for (int m = 0; m < 4; m++) {
                            int64_t ll = l0 + l + 64 * m;
                            if (ll < l1) { double bv = h->beta[Sx->col0 + ll]; a = __builtin_fma(bv, bv, a); }
                        }
                        lane[l] = a;
                    }
                    double p = wave_butterfly(lane);
                    tot = first ? p : tot + p; first = 0;
                }
                rng_seed(&r, h->seed, h->chain, it, KIND_REGION_CHI2, ((uint64_t)si << 40) | (uint64_t)rg);
                double n_r = (Sx->method == METHOD_C) ? (double)nLoci : (double)(Sx->reg_stop[rg] - Sx->reg_start[rg]);
                double ch = rng_chisq(&r, Sx->df + n_r);
                double tt = Sx->scale * Sx->df; tt = tt + tot;
                vb[rg] = tt / ch;
            }
            if (Sx->method == METHOD_C && Sx->estPi) {
                rng_seed(&r, h->seed, h->chain, it, KIND_PI_BETA, (uint64_t)si);
                double piIn = rng_beta(&r, (double)nLoci + 1.0, (double)(Sx->ncol - nLoci) + 1.0);
                Sx->piHat[0] = 1.0 - piIn; Sx->piHat[1] = piIn;
                Sx->logPi[0] = det_log(Sx->piHat[0]); Sx->logPi[1] = det_log(piIn);
            }
        } else {
            int64_t nLoci = 0;
            for (int64_t l = 0; l < Sx->ncol; l++) {
                int64_t k = Sx->col0 + l;
                if (h->delta[k]) {
                    nLoci++;
                    double bv = h->beta[k];
                    double tt = Sx->scale * Sx->df; double b2 = bv * bv; tt = tt + b2;
                    vb[l] = tt / h->chi[k];
                } else vb[l] = 0.0;
            }
            if (Sx->estPi) {
                rng_seed(&r, h->seed, h->chain, it, KIND_PI_BETA, (uint64_t)si);
                double piIn = rng_beta(&r, (double)nLoci + 1.0, (double)(Sx->ncol - nLoci) + 1.0);
                Sx->piHat[0] = 1.0 - piIn; Sx->piHat[1] = piIn;
                Sx->logPi[0] = det_log(Sx->piHat[0]); Sx->logPi[1] = det_log(piIn);
            }
        }
    }
    h->iter = it;
}

/* ------------------------------------------------------------------ */
/* order 0, all cores: ONE parallel region per sweep                    */
/* ------------------------------------------------------------------ */
/* bench.py's all-core CPU baseline (never used by a test).  The reference runs OpenBLAS, whose level-1 calls thread over
   the rows of a long vector; a fork/join per ddot / daxpy (what ora_set_threads gives) costs more than it saves on a busy
   box, so this path keeps the same arithmetic per SNP (add-back daxpy over M.data, ddot over the Mp copy, draw, update
   daxpy: 24 N bytes of DRAM traffic, src/functions.jl:128-133) but gives every thread a fixed chunk of rows for the whole
   sweep: per SNP each thread adds the old effect back into its rows and forms its part of the dot product, ONE barrier,
   then every thread adds the T parts in the same order, draws the same beta_j (the keyed draw layer makes that free of
   communication) and updates its rows.  Partial dot products are double-buffered by SNP parity, so one barrier per SNP
   suffices.  Summation order differs from the 1-thread path (chunked dot): a timing path, not a parity path.
   BayesPR sets only (the configuration the headline is quoted on); returns ORA_ERR otherwise. */
int ora_run_pr_threaded(ora_t *h, int64_t niter, int T) {
    if (h->order != 0 || !h->ycorr) { snprintf(h->err, 256, "threaded baseline: reference order only"); return ORA_ERR; }
    for (int si = 0; si < h->nsets; si++)
        if (h->sets[si].method != METHOD_PR) { snprintf(h->err, 256, "threaded baseline: BayesPR sets only"); return ORA_ERR; }
    if (T < 1) T = 1;
    if (T > 64) T = 64;
    const int64_t N = h->N;
    double (*part)[64][8] = (double (*)[64][8])calloc(2, sizeof(*part)); /* [parity][thread][pad to a cache line] */
    if (!part) return ORA_ERR;
    for (int64_t n = 0; n < niter; n++) {
        const int64_t it = h->iter + 1;
        rng_t r;
        rng_seed(&r, h->seed, h->chain, it, KIND_VARE_CHI2, 0);
        double varE = (h->e_df * h->e_scale + dot8_1(h->ycorr, h->ycorr, N)) / rng_chisq(&r, h->e_df + (double)N);
        h->varE = varE;
        const double iVarE = 1.0 / varE;
        if (h->intercept) {
            double s = 0.0;
            for (int64_t i = 0; i < N; i++) { h->ycorr[i] += h->b; s += h->ycorr[i]; }
            double lhs = (double)N * iVarE;
            rng_seed(&r, h->seed, h->chain, it, KIND_FIXED_NORMAL, 0);
            h->b = (s * iVarE) / lhs + sqrt(1.0 / lhs) * rng_normal(&r);
            for (int64_t k = 0; k < N; k++) h->ycorr[k] -= h->b;
        }
#pragma omp parallel num_threads(T)
        {
            int tid = 0, nt = 1;
#ifdef _OPENMP
            extern int omp_get_thread_num(void); extern int omp_get_num_threads(void);
            tid = omp_get_thread_num(); nt = omp_get_num_threads();
#endif
            const int64_t chunk = (((N + nt - 1) / nt) + 7) & ~(int64_t)7;
            const int64_t lo = tid * chunk < N ? tid * chunk : N, hi = lo + chunk < N ? lo + chunk : N;
            rng_t rr;
            int64_t snp = 0;
            for (int si = 0; si < h->nsets; si++) {
                oset_t *S = &h->sets[si];
                double *vb = h->varBeta + S->vb_off;
                for (int64_t rg = 0; rg < S->nreg; rg++) {
                    const double iVarBeta = 1.0 / vb[rg];
                    double ssq = 0.0;
                    for (int64_t l = S->reg_start[rg]; l < S->reg_stop[rg]; l++, snp++) {
                        const int64_t j = S->col0 + l;
                        const double *col = h->data + j * N, *mp = h->Mp + j * N;
                        const double bold = h->beta[j];
                        double *y = h->ycorr;
                        for (int64_t i = lo; i < hi; i++) y[i] = __builtin_fma(bold, col[i], y[i]);          /* :128 */
                        part[snp & 1][tid][0] = dot8_1(mp + lo, y + lo, hi - lo);                            /* :129, this chunk */
#pragma omp barrier
                        double d = 0.0;
                        for (int t = 0; t < nt; t++) d += part[snp & 1][t][0];
                        const double rhs = d * iVarE + h->rhs0[j];
                        const double lhs = h->mpm[j] * iVarE + h->lhs0[j] + iVarBeta;                         /* :130 */
                        rng_seed(&rr, h->seed, h->chain, it, KIND_BETA_NORMAL, ((uint64_t)si << 40) | (uint64_t)l);
                        const double bn = rhs / lhs + sqrt(1.0 / lhs) * rng_normal(&rr);                      /* :131-132 */
                        for (int64_t i = lo; i < hi; i++) y[i] = __builtin_fma(-bn, col[i], y[i]);           /* :133 */
                        ssq = __builtin_fma(bn, bn, ssq);
                        /* beta[j] is read again only in the next iteration; one writer keeps the store race-free, and the
                           barrier of the next SNP orders it before anybody's next read */
                        if (tid == 0) h->beta[j] = bn;
                    }
                    rng_seed(&rr, h->seed, h->chain, it, KIND_REGION_CHI2, ((uint64_t)si << 40) | (uint64_t)rg);
                    const double vnew = (S->scale * S->df + ssq) / rng_chisq(&rr, S->df + (double)(S->reg_stop[rg] - S->reg_start[rg]));
#pragma omp barrier
                    if (tid == 0) vb[rg] = vnew;                                                              /* :135 */
#pragma omp barrier
                }
            }
        }
        h->iter = it;
        if (is_kept(h, h->iter)) accumulate(h);
    }
    free(part);
    return ORA_OK;
}

/* advance the chain by niter iterations (samplers.jl:29-105) */
int ora_run(ora_t *h, int64_t niter) {
    if (!h->ycorr || !h->beta) { snprintf(h->err, 256, "panel / y not set"); return ORA_ERR; }
    free(h->tr_varE); free(h->tr_b);
    h->tr_varE = (double *)malloc(sizeof(double) * (niter > 0 ? niter : 1));
    h->tr_b = (double *)malloc(sizeof(double) * (niter > 0 ? niter : 1));
    h->ntrace = niter;
    for (int64_t n = 0; n < niter; n++) {
        const int64_t before = h->iter;
        if (h->order == 0) iter_ref(h); else iter_blocked(h);
        if (h->iter == before) return ORA_ERR;  /* the iteration gave up (message in h->err) */
        h->tr_varE[n] = h->varE; h->tr_b[n] = h->b;
        if (is_kept(h, h->iter)) accumulate(h);
    }
    return ORA_OK;
}

int ora_get_state(ora_t *h, double *ycorr, double *beta, int64_t *delta, double *varBeta, double *piHat, double *varE, double *b,
                  int64_t *iter) {
    if (ycorr) memcpy(ycorr, h->ycorr, sizeof(double) * h->N);
    if (beta) memcpy(beta, h->beta, sizeof(double) * h->P);
    if (delta) memcpy(delta, h->delta, sizeof(int64_t) * h->P);
    if (varBeta) memcpy(varBeta, h->varBeta, sizeof(double) * h->nvb);
    if (piHat) for (int s = 0; s < h->nsets; s++) { piHat[2 * s] = h->sets[s].piHat[0]; piHat[2 * s + 1] = h->sets[s].piHat[1]; }
    if (varE) *varE = h->varE;
    if (b) *b = h->b;
    if (iter) *iter = h->iter;
    return ORA_OK;
}
int ora_get_trace(ora_t *h, double *varE, double *b, int64_t n) {
    if (n > h->ntrace) n = h->ntrace;
    if (varE) memcpy(varE, h->tr_varE, sizeof(double) * n);
    if (b) memcpy(b, h->tr_b, sizeof(double) * n);
    return ORA_OK;
}
int ora_get_posterior_sums(ora_t *h, double *sum_beta, double *sum_beta2, double *sum_delta, double *sum_varBeta, double *sum_pi,
                           double *sum_varE, double *sum_b, int64_t *nKept) {
    if (sum_beta) memcpy(sum_beta, h->sum_beta, sizeof(double) * h->P);
    if (sum_beta2) memcpy(sum_beta2, h->sum_beta2, sizeof(double) * h->P);
    if (sum_delta) memcpy(sum_delta, h->sum_delta, sizeof(double) * h->P);
    if (sum_varBeta) memcpy(sum_varBeta, h->sum_varBeta, sizeof(double) * h->nvb);
    if (sum_pi) for (int s = 0; s < h->nsets; s++) { sum_pi[2 * s] = h->sets[s].sum_pi[0]; sum_pi[2 * s + 1] = h->sets[s].sum_pi[1]; }
    if (sum_varE) *sum_varE = h->sum_varE;
    if (sum_b) *sum_b = h->sum_b;
    if (nKept) *nKept = h->nKept;
    return ORA_OK;
}
int64_t ora_nvb(ora_t *h) { return h->nvb; }
/* Gram block access for parity tests of the device Gram kernel (order 1 only) */
int ora_get_gram(ora_t *h, int64_t t, double *out) {
    if (h->order != 1 || t < 0 || t >= h->NBLK) return ORA_ERR;
    memcpy(out, h->gram + (size_t)t * BLK * BLK, sizeof(double) * BLK * BLK);
    return ORA_OK;
}
