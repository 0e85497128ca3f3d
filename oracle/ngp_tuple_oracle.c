/*
 * ngp_tuple_oracle.c -- SPEC ORACLE for the correlated (Tuple) BayesPR marker sets of NextGP.jl.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT, and deliberately without a device counterpart yet: in the reference snapshot this path is
 * unreachable (:method / :funct / :nRegions are never set for tuple keys, /root/reference/src/mme.jl:530,590,
 * src/samplers.jl:52 would fail), so there is nothing to be a drop-in for.  This file pins down WHAT the path computes, so that
 * SURVEY.md section 8 rows a9 / f2 are a specification with tests instead of a blank:
 *
 *   src/functions.jl:140-154   sampleBayesPR!(::Tuple): per locus, k correlated sets (breeds):
 *                                ycorr += X_j beta_j ; RHS = X_j' ycorr / varE ; invLHS = inv(X_j'X_j / varE + inv(varBeta[r]))
 *                                beta_j ~ MvNormal(invLHS RHS, invLHS) ; ycorr -= X_j beta_j
 *   src/functions.jl:513-516   sampleVarCovBetaPR: varBeta[r] ~ InverseWishart(df + #r, scale + B_r' B_r)
 *   src/mme.jl:448-489         set-up: X_j = N x k (column j of each set), mpm[j] = X_j'X_j (k x k), regions
 *   src/mme.jl:493, 501, 516   df = 3 + k, scale = v (df - k - 1), varBeta[r] = v (k x k)
 *
 * PARITY UNPINNED (as the main oracle): no fixtures in the reference, no Julia here.  Draw layer: the MvNormal and
 * InverseWishart samplers live in Distributions.jl (absent); they are restated by their textbook constructions on the keyed
 * xoshiro streams of the main oracle:
 *   MvNormal(m, S)        = m + L z,   L = lower Cholesky factor of S, z_i iid N(0,1)
 *   InverseWishart(nu, P) = inv(W),    W = (L A)(L A)',  L = chol(inv(P)),  A lower triangular (Bartlett): A_ii = sqrt(chi2(nu - i)),
 *                           A_ij ~ N(0,1), i > j   (i = 0..k-1)
 * Pinned by closed forms (tests/test_tuple_oracle.py): conditional mean / covariance of one locus, E[IW] = P / (nu - k - 1),
 * and k = 1 reducing to the scalar sampleBayesPR!(::Symbol) path of ngp_oracle.c.
 *
 * Build: oracle/Makefile (libngp_tuple_oracle.so).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define KMAX 4
#define GOLD 0x9E3779B97F4A7C15ULL
#define KIND_VARE_CHI2 1
#define KIND_T_NORMAL 10  /* components of the MvNormal draw: index (locus << 3) | component */
#define KIND_T_WISHART 11 /* Bartlett factor of region r: index (r << 8) | (i << 4) | j */

static inline uint64_t mix64(uint64_t z) { z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31); }
typedef struct { uint64_t s[4]; } rng_t;
static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static inline uint64_t rng_next(rng_t *r) {
    uint64_t *s = r->s, res = rotl64(s[0] + s[3], 23) + s[0], t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl64(s[3], 45);
    return res;
}
static inline uint64_t absorb(uint64_t h, uint64_t v) { return mix64(h ^ mix64(v + GOLD)); }
static void rng_seed(rng_t *r, uint64_t seed, uint64_t chain, uint64_t iter, uint64_t kind, uint64_t index) {
    uint64_t h = mix64(seed + GOLD);
    h = absorb(h, chain); h = absorb(h, iter); h = absorb(h, kind); h = absorb(h, index);
    for (int i = 0; i < 4; i++) r->s[i] = mix64(h + (uint64_t)(i + 1) * GOLD);
}
static inline double rng_uniform(rng_t *r) { return ((double)(rng_next(r) >> 12) + 0.5) * 2.220446049250313080847263336181640625e-16; }
/* libm suffices here: this oracle has no bit-parity partner */
static double rng_normal(rng_t *r) { /* inverse CDF through erfinv-free Acklam-style refinement is overkill: Box-Muller on two uniforms */
    double u1 = rng_uniform(r), u2 = rng_uniform(r);
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925286766559 * u2);
}
static double rng_gamma(rng_t *r, double a) { /* Marsaglia-Tsang, a >= 1 */
    double d = a - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    for (;;) {
        double x, v;
        do { x = rng_normal(r); v = 1.0 + c * x; } while (v <= 0.0);
        v = v * v * v;
        double u = rng_uniform(r);
        if (log(u) < 0.5 * x * x + d * (1.0 - v + log(v))) return d * v;
    }
}
static double rng_chisq(rng_t *r, double nu) { return 2.0 * rng_gamma(r, 0.5 * nu); }

/* ---- k x k helpers (row-major, k <= KMAX) ---- */
static int chol_lower(const double *S, int k, double *L) { /* S = L L' */
    memset(L, 0, sizeof(double) * k * k);
    for (int i = 0; i < k; i++)
        for (int j = 0; j <= i; j++) {
            double s = S[i * k + j];
            for (int m = 0; m < j; m++) s -= L[i * k + m] * L[j * k + m];
            if (i == j) { if (s <= 0.0) return -1; L[i * k + i] = sqrt(s); }
            else L[i * k + j] = s / L[j * k + j];
        }
    return 0;
}
static int inv_spd(const double *S, int k, double *out) { /* through the Cholesky factor */
    double L[KMAX * KMAX], Li[KMAX * KMAX];
    if (chol_lower(S, k, L)) return -1;
    memset(Li, 0, sizeof(Li));
    for (int c = 0; c < k; c++)
        for (int i = c; i < k; i++) {
            double s = (i == c) ? 1.0 : 0.0;
            for (int m = c; m < i; m++) s -= L[i * k + m] * Li[m * k + c];
            Li[i * k + c] = s / L[i * k + i];
        }
    for (int i = 0; i < k; i++)
        for (int j = 0; j < k; j++) {
            double s = 0.0;
            for (int m = 0; m < k; m++) s += Li[m * k + i] * Li[m * k + j];
            out[i * k + j] = s;
        }
    return 0;
}

/* InverseWishart(nu, P): Bartlett on inv(P), then invert */
static int draw_inverse_wishart(uint64_t seed, uint64_t chain, uint64_t iter, uint64_t region, double nu, const double *P, int k, double *out) {
    double Pi[KMAX * KMAX], L[KMAX * KMAX], A[KMAX * KMAX], LA[KMAX * KMAX], W[KMAX * KMAX];
    if (inv_spd(P, k, Pi) || chol_lower(Pi, k, L)) return -1;
    memset(A, 0, sizeof(A));
    rng_t r;
    for (int i = 0; i < k; i++)
        for (int j = 0; j <= i; j++) {
            rng_seed(&r, seed, chain, iter, KIND_T_WISHART, (region << 8) | ((uint64_t)i << 4) | (uint64_t)j);
            A[i * k + j] = (i == j) ? sqrt(rng_chisq(&r, nu - (double)i)) : rng_normal(&r);
        }
    for (int i = 0; i < k; i++)
        for (int j = 0; j < k; j++) { double s = 0.0; for (int m = 0; m < k; m++) s += L[i * k + m] * A[m * k + j]; LA[i * k + j] = s; }
    for (int i = 0; i < k; i++)
        for (int j = 0; j < k; j++) { double s = 0.0; for (int m = 0; m < k; m++) s += LA[i * k + m] * LA[j * k + m]; W[i * k + j] = s; }
    return inv_spd(W, k, out);
}
/* test probe */
int tup_inverse_wishart(uint64_t seed, uint64_t chain, uint64_t iter, uint64_t region, double nu, const double *P, int k, double *out) {
    return draw_inverse_wishart(seed, chain, iter, region, nu, P, k, out);
}

typedef struct {
    int k; int64_t N, P, nreg;
    double *data;   /* [locus][N][k]: X_j, N x k row-major */
    double *mpm;    /* [locus][k][k] */
    int64_t *reg_start, *reg_stop;
    double *varBeta; /* [region][k][k] */
    double df, *scale; /* scale k x k */
    double *beta;   /* [locus][k] */
    double *ycorr, e_df, e_scale, varE;
    uint64_t seed, chain; int64_t iter;
    int fix_var;    /* tests: keep varBeta and varE at their current values */
} tup_t;

int tup_create(int k, int64_t N, int64_t P, const double *X /* [set][locus][N] */, const double *y, const int64_t *reg_start,
               const int64_t *reg_stop, int64_t nreg, const double *v /* k x k */, uint64_t seed, uint32_t chain, tup_t **out) {
    if (k < 1 || k > KMAX) return -1;
    tup_t *h = (tup_t *)calloc(1, sizeof(tup_t));
    h->k = k; h->N = N; h->P = P; h->nreg = nreg; h->seed = seed; h->chain = chain;
    h->data = (double *)malloc(sizeof(double) * P * N * k);
    h->mpm = (double *)calloc(P * k * k, sizeof(double));
    for (int64_t j = 0; j < P; j++) {                      /* mme.jl:456-457, :462 */
        for (int64_t i = 0; i < N; i++)
            for (int s = 0; s < k; s++) h->data[(j * N + i) * k + s] = X[((int64_t)s * P + j) * N + i];
        for (int a = 0; a < k; a++)
            for (int b = 0; b < k; b++) {
                double acc = 0.0;
                for (int64_t i = 0; i < N; i++) acc += h->data[(j * N + i) * k + a] * h->data[(j * N + i) * k + b];
                h->mpm[(j * k + a) * k + b] = acc;
            }
    }
    h->reg_start = (int64_t *)malloc(sizeof(int64_t) * nreg); h->reg_stop = (int64_t *)malloc(sizeof(int64_t) * nreg);
    memcpy(h->reg_start, reg_start, sizeof(int64_t) * nreg); memcpy(h->reg_stop, reg_stop, sizeof(int64_t) * nreg);
    h->df = 3.0 + (double)k;                                /* mme.jl:493 */
    h->scale = (double *)malloc(sizeof(double) * k * k);
    h->varBeta = (double *)malloc(sizeof(double) * nreg * k * k);
    for (int a = 0; a < k * k; a++) h->scale[a] = (k > 1) ? v[a] * (h->df - (double)k - 1.0) : v[a] * (h->df - 2.0) / h->df; /* mme.jl:501 */
    for (int64_t r = 0; r < nreg; r++) memcpy(h->varBeta + r * k * k, v, sizeof(double) * k * k);                          /* mme.jl:516 */
    h->beta = (double *)calloc(P * k, sizeof(double));
    h->ycorr = (double *)malloc(sizeof(double) * N); memcpy(h->ycorr, y, sizeof(double) * N);
    h->e_df = 4.0; h->e_scale = 0.0005; h->varE = 1.0;
    *out = h;
    return 0;
}
void tup_destroy(tup_t *h) {
    if (!h) return;
    free(h->data); free(h->mpm); free(h->reg_start); free(h->reg_stop); free(h->scale); free(h->varBeta); free(h->beta); free(h->ycorr); free(h);
}
void tup_set_residual_prior(tup_t *h, double df, double scale) { h->e_df = df; h->e_scale = scale; }
void tup_fix_variances(tup_t *h, int on, double varE) { h->fix_var = on; if (on) h->varE = varE; }

int tup_run(tup_t *h, int64_t niter) {
    const int k = h->k; const int64_t N = h->N;
    rng_t r;
    for (int64_t n = 0; n < niter; n++) {
        const uint64_t it = (uint64_t)(h->iter + 1);
        if (!h->fix_var) {                                                   /* samplers.jl:32-35, functions.jl:523-525 */
            double yy = 0.0; for (int64_t i = 0; i < N; i++) yy += h->ycorr[i] * h->ycorr[i];
            rng_seed(&r, h->seed, h->chain, it, KIND_VARE_CHI2, 0);
            h->varE = (h->e_df * h->e_scale + yy) / rng_chisq(&r, h->e_df + (double)N);
        }
        const double varE = h->varE;
        for (int64_t rg = 0; rg < h->nreg; rg++) {                           /* functions.jl:141 */
            double invB[KMAX * KMAX];
            if (inv_spd(h->varBeta + rg * k * k, k, invB)) return -2;       /* :143 */
            double Sb[KMAX * KMAX]; memset(Sb, 0, sizeof(Sb));
            for (int64_t j = h->reg_start[rg]; j < h->reg_stop[rg]; j++) {
                const double *Xj = h->data + j * N * k; double *bj = h->beta + j * k;
                for (int64_t i = 0; i < N; i++) { double t = 0.0; for (int s = 0; s < k; s++) t += Xj[i * k + s] * bj[s]; h->ycorr[i] += t; }   /* :145 */
                double RHS[KMAX], LHS[KMAX * KMAX], invLHS[KMAX * KMAX], L[KMAX * KMAX], mean[KMAX], z[KMAX];
                for (int s = 0; s < k; s++) { double t = 0.0; for (int64_t i = 0; i < N; i++) t += Xj[i * k + s] * h->ycorr[i]; RHS[s] = t / varE; }  /* :146 */
                for (int a = 0; a < k * k; a++) LHS[a] = h->mpm[j * k * k + a] / varE + invB[a];                                              /* :147 */
                if (inv_spd(LHS, k, invLHS) || chol_lower(invLHS, k, L)) return -3;
                for (int a = 0; a < k; a++) { double t = 0.0; for (int b = 0; b < k; b++) t += invLHS[a * k + b] * RHS[b]; mean[a] = t; }       /* :148 */
                for (int s = 0; s < k; s++) { rng_seed(&r, h->seed, h->chain, it, KIND_T_NORMAL, ((uint64_t)j << 3) | (uint64_t)s); z[s] = rng_normal(&r); }
                for (int a = 0; a < k; a++) { double t = mean[a]; for (int b = 0; b <= a; b++) t += L[a * k + b] * z[b]; bj[a] = t; }           /* :149 */
                for (int64_t i = 0; i < N; i++) { double t = 0.0; for (int s = 0; s < k; s++) t += Xj[i * k + s] * bj[s]; h->ycorr[i] -= t; }   /* :150 */
                for (int a = 0; a < k; a++) for (int b = 0; b < k; b++) Sb[a * k + b] += bj[a] * bj[b];                                         /* :514 */
            }
            if (!h->fix_var) {
                double Psi[KMAX * KMAX];
                for (int a = 0; a < k * k; a++) Psi[a] = h->scale[a] + Sb[a];
                const double nu = h->df + (double)(h->reg_stop[rg] - h->reg_start[rg]);
                if (k == 1) {  /* InverseWishart(nu, psi) in one dimension = psi / chi2(nu); the scalar path's (scale df + ssq) / chi2 differs
                                  only in how mme.jl:501 defines scale for one component */
                    rng_seed(&r, h->seed, h->chain, it, KIND_T_WISHART, ((uint64_t)rg << 8));
                    h->varBeta[rg] = Psi[0] / rng_chisq(&r, nu);
                } else if (draw_inverse_wishart(h->seed, h->chain, it, (uint64_t)rg, nu, Psi, k, h->varBeta + rg * k * k)) return -4;       /* :152, :513-516 */
            }
        }
        h->iter++;
    }
    return 0;
}
void tup_get_state(tup_t *h, double *beta /* [locus][k] */, double *ycorr, double *varBeta, double *varE) {
    if (beta) memcpy(beta, h->beta, sizeof(double) * h->P * h->k);
    if (ycorr) memcpy(ycorr, h->ycorr, sizeof(double) * h->N);
    if (varBeta) memcpy(varBeta, h->varBeta, sizeof(double) * h->nreg * h->k * h->k);
    if (varE) *varE = h->varE;
}
