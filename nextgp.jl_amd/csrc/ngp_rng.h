// ngp_rng.h -- device-resident draw layer (gfx950): keyed xoshiro256++ streams, inverse-CDF
// normal, Marsaglia-Tsang gamma / chi-square / beta, and a fixed-sequence double log.
//
// Replaces the draws the reference delegates to Distributions.jl / Random
// (/root/reference/src/functions.jl:493-495 Normal, :509-511 & :523-525 Chisq, :531-533 Beta,
//  :174 rand()).  Every function is a fixed sequence of IEEE-754 double operations (no FMA
//  contraction: this TU is compiled with -ffp-contract=off), so results are reproducible bit
//  for bit on any IEEE host; DESIGN.md "RNG and draw spec" is the normative text.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

#define NGP_KIND_VARE_CHI2 1
#define NGP_KIND_FIXED_NORMAL 2
#define NGP_KIND_BETA_NORMAL 3
#define NGP_KIND_REGION_CHI2 4
#define NGP_KIND_B_UNIFORM 5
#define NGP_KIND_B_LOCUS_CHI2 6
#define NGP_KIND_PI_BETA 7
#define NGP_KIND_R_UNIFORM 8    // BayesR: the fresh uniform of every comparison of the class search (src/functions.jl:261)
#define NGP_KIND_R_DIRICHLET 9  // BayesR: gamma draws of the Dirichlet (src/functions.jl:536-538)
#define NGP_KIND_T_WISHART 11   // Tuple sets: Bartlett factor of a region's inverse-Wishart draw, (set << 40) | (region << 8) | (i << 4) | j

#define NGP_GOLD 0x9E3779B97F4A7C15ULL

namespace ngp {

__host__ __device__ inline uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

struct Rng {
    uint64_t s0, s1, s2, s3;
};

__host__ __device__ inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

__host__ __device__ inline uint64_t rng_next(Rng &r) {
    uint64_t res = rotl64(r.s0 + r.s3, 23) + r.s0;
    uint64_t t = r.s1 << 17;
    r.s2 ^= r.s0;
    r.s3 ^= r.s1;
    r.s1 ^= r.s2;
    r.s0 ^= r.s3;
    r.s2 ^= t;
    r.s3 = rotl64(r.s3, 45);
    return res;
}

__host__ __device__ inline uint64_t absorb(uint64_t h, uint64_t v) { return mix64(h ^ mix64(v + NGP_GOLD)); }

__host__ __device__ inline Rng rng_seed(uint64_t seed, uint64_t chain, uint64_t iter, uint64_t kind, uint64_t index) {
    uint64_t h = mix64(seed + NGP_GOLD);
    h = absorb(h, chain);
    h = absorb(h, iter);
    h = absorb(h, kind);
    h = absorb(h, index);
    Rng r;
    r.s0 = mix64(h + 1 * NGP_GOLD);
    r.s1 = mix64(h + 2 * NGP_GOLD);
    r.s2 = mix64(h + 3 * NGP_GOLD);
    r.s3 = mix64(h + 4 * NGP_GOLD);
    return r;
}

// (k + 0.5) * 2^-52 with k the top 52 bits: strictly inside (0,1)
__host__ __device__ inline double rng_uniform(Rng &r) {
    uint64_t k = rng_next(r) >> 12;
    return ((double)k + 0.5) * 2.220446049250313080847263336181640625e-16;
}

__device__ inline double det_log(double x) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    uint64_t bits = (uint64_t)__double_as_longlong(x);
    if (x <= 0.0) return (x == 0.0) ? -__builtin_huge_val() : __builtin_nan("");
    if ((bits >> 52) == 0x7FF) return x;
    int k = 0;
    if ((bits >> 52) == 0) {
        x *= 18014398509481984.0;
        bits = (uint64_t)__double_as_longlong(x);
        k = -54;
    }
    uint32_t hx = (uint32_t)(bits >> 32);
    k += (int)(hx >> 20) - 1023;
    hx &= 0x000fffffu;
    uint32_t i = (hx + 0x95f64u) & 0x100000u;
    uint64_t nb = ((uint64_t)(hx | (i ^ 0x3ff00000u)) << 32) | (bits & 0xffffffffULL);
    k += (int)(i >> 20);
    double xn = __longlong_as_double((long long)nb);
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

// det_exp for x <= 0: fixed IEEE sequence (fdlibm's argument reduction, then the Taylor series to degree 13 by Horner), bit for bit the
// oracle's; results below 2^-1021 come back as 0.  Only the blocked BayesR class search uses it (on L - max L <= 0).  No division
// since round 4: the four exponentials of a class evaluation run interleaved, and the v_div_scale / v_div_fmas pairs of the
// rational form serialised them on VCC (2.2 us per evaluation of a block's lanes).  Within 1 ulp of libm (tests/test_oracle.py).
__device__ inline double det_exp(double x) {
    const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10, invln2 = 1.44269504088896338700e+00;
    // (no branch: NaN and underflow replace the result at the end, so that the exponentials of a class evaluation stay in one basic
    // block and their dependent chains interleave -- behind two early returns each ran alone, 4 x 40 dependent instructions)
    const bool isnan_x = x != x, under = x < -708.0;
    const double x_in = x;
    x = (x > 0.0) ? 0.0 : x;
    x = (isnan_x || under) ? 0.0 : x;
    const int k = (int)(invln2 * x - 0.5);
    const double t = (double)k;
    const double hi = x - t * ln2HI, lo = t * ln2LO;
    const double xr = hi - lo;  // |xr| <= ln 2 / 2
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
    const double sc = __longlong_as_double((long long)((uint64_t)(k + 1023) << 52));
    const double res = under ? 0.0 : y * sc;
    return isnan_x ? x_in : res;
}

// det_exp for an argument that is <= 0 or a NaN (the class search's L - max L): the same operations, so the same bits, without the
// guards such an argument does not need -- x < -708 (and -inf: a class that does not exist) gives 0, a NaN comes back a NaN.
__device__ inline double det_exp_le0(const double xin) {
    const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10, invln2 = 1.44269504088896338700e+00;
    const bool under = xin < -708.0;
    const double x = under ? 0.0 : xin;
    const int k = (int)(invln2 * x - 0.5);
    const double t = (double)k;
    const double hi = x - t * ln2HI, lo = t * ln2LO;
    const double xr = hi - lo;
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
    const double sc = __longlong_as_double((long long)((uint64_t)(k + 1023) << 52));
    return under ? 0.0 : y * sc;
}

// Four det_exp side by side for arguments x <= 0 (the class search's L - max L; the same operations per element, so the same bits):
// every step of the four dependent chains is issued before the next step of any -- the empty asm statements pin that order, which the
// scheduler does not choose by itself (it kept the four chains one behind the other, 4 x 40 dependent fp64 instructions per class
// evaluation of the BayesR chain).  x < -708 (and -inf: a class that does not exist) gives 0; a NaN goes through the arithmetic and
// comes back a NaN.
__device__ inline void det_exp4(const double (&xin)[4], double (&out)[4]) {
    const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10, invln2 = 1.44269504088896338700e+00;
    const double coef[11] = {1.0 / 479001600.0, 1.0 / 39916800.0, 1.0 / 3628800.0, 1.0 / 362880.0, 1.0 / 40320.0, 1.0 / 5040.0,
                             1.0 / 720.0, 1.0 / 120.0, 1.0 / 24.0, 1.0 / 6.0, 0.5};
    double xr[4], q[4], t[4];
    int k[4];
#pragma unroll
    for (int v = 0; v < 4; v++) xr[v] = (xin[v] < -708.0) ? 0.0 : xin[v];
#pragma unroll
    for (int v = 0; v < 4; v++) t[v] = invln2 * xr[v] - 0.5;
#pragma unroll
    for (int v = 0; v < 4; v++) { k[v] = (int)t[v]; t[v] = (double)k[v]; }
    asm volatile("" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]));
#pragma unroll
    for (int v = 0; v < 4; v++) {
        const double hi = xr[v] - t[v] * ln2HI, lo = t[v] * ln2LO;
        xr[v] = hi - lo;
    }
    asm volatile("" : "+v"(xr[0]), "+v"(xr[1]), "+v"(xr[2]), "+v"(xr[3]));
#pragma unroll
    for (int v = 0; v < 4; v++) q[v] = 1.0 / 6227020800.0;
#pragma unroll
    for (int i = 0; i < 11; i++) {
#pragma unroll
        for (int v = 0; v < 4; v++) q[v] = __builtin_fma(q[v], xr[v], coef[i]);
        asm volatile("" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]));
    }
#pragma unroll
    for (int v = 0; v < 4; v++) {
        const double tt = xr[v] * xr[v];
        const double y = 1.0 + __builtin_fma(tt, q[v], xr[v]);
        const double sc = __longlong_as_double((long long)((uint64_t)(k[v] + 1023) << 52));
        out[v] = (xin[v] < -708.0) ? 0.0 : y * sc;
    }
}

// IEEE correctly rounded square root (sqrt() lowers to the ocml routine, which is)
__device__ inline double det_sqrt(double x) { return __builtin_sqrt(x); }

// Wichura (1988) AS241 PPND16
__device__ inline double ppnd16(double p) {
    double q = p - 0.5, r, val;
    if (__builtin_fabs(q) <= 0.425) {
        r = 0.180625 - q * q;
        double num = (((((((2.5090809287301226727e3 * r + 3.3430575583588128105e4) * r + 6.7265770927008700853e4) * r +
                          4.5921953931549871457e4) * r + 1.3731693765509461125e4) * r + 1.9715909503065514427e3) * r +
                       1.3314166789178437745e2) * r + 3.3871328727963666080e0);
        double den = (((((((5.2264952788528545610e3 * r + 2.8729085735721942674e4) * r + 3.9307895800092710610e4) * r +
                          2.1213794301586595867e4) * r + 5.3941960214247511077e3) * r + 6.8718700749205790830e2) * r +
                       4.2313330701600911252e1) * r + 1.0);
        return q * num / den;
    }
    r = (q < 0.0) ? p : 1.0 - p;
    r = det_sqrt(-det_log(r));
    if (r <= 5.0) {
        r = r - 1.6;
        double num = (((((((7.74545014278341407640e-4 * r + 2.27238449892691845833e-2) * r + 2.41780725177450611770e-1) * r +
                          1.27045825245236838258e0) * r + 3.64784832476320460504e0) * r + 5.76949722146069140550e0) * r +
                       4.63033784615654529590e0) * r + 1.42343711074968357734e0);
        double den = (((((((1.05075007164441684324e-9 * r + 5.47593808499534494600e-4) * r + 1.51986665636164571966e-2) * r +
                          1.48103976427480074590e-1) * r + 6.89767334985100004550e-1) * r + 1.67638483018380384940e0) * r +
                       2.05319162663775882187e0) * r + 1.0);
        val = num / den;
    } else {
        r = r - 5.0;
        double num = (((((((2.01033439929228813265e-7 * r + 2.71155556874348757815e-5) * r + 1.24266094738807843860e-3) * r +
                          2.65321895265761230930e-2) * r + 2.96560571828504891230e-1) * r + 1.78482653991729133580e0) * r +
                       5.46378491116411436990e0) * r + 6.65790464350110377720e0);
        double den = (((((((2.04426310338993978564e-15 * r + 1.42151175831644588870e-7) * r + 1.84631831751005468180e-5) * r +
                          7.86869131145613259100e-4) * r + 1.48753612908506148525e-2) * r + 1.36929880922735805310e-1) * r +
                       5.99832206555887937690e-1) * r + 1.0);
        val = num / den;
    }
    return (q < 0.0) ? -val : val;
}

__device__ inline double rng_normal(Rng &r) { return ppnd16(rng_uniform(r)); }

// Marsaglia & Tsang (2000), shape a >= 1, no squeeze step
__device__ inline double rng_gamma(Rng &r, double a) {
    double d = a - 1.0 / 3.0;
    double c = 1.0 / det_sqrt(9.0 * d);
    for (;;) {
        double x, v;
        do {
            x = rng_normal(r);
            v = 1.0 + c * x;
        } while (v <= 0.0);
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
__device__ inline double rng_chisq(Rng &r, double nu) { return 2.0 * rng_gamma(r, 0.5 * nu); }
__device__ inline double rng_beta(Rng &r, double a, double b) {
    double ga = rng_gamma(r, a);
    double gb = rng_gamma(r, b);
    return ga / (ga + gb);
}

// synthetic panel hash
__host__ __device__ inline double panel_pj(uint64_t pseed, int64_t j, double lo, double hi) {
    uint64_t h = mix64(mix64(pseed ^ 0xA5A5A5A55A5A5A5AULL) + (uint64_t)j * NGP_GOLD);
    double u = ((double)(h >> 12) + 0.5) * 2.220446049250313080847263336181640625e-16;
    return lo + (hi - lo) * u;
}
__host__ __device__ inline uint64_t panel_colkey(uint64_t pseed, int64_t j) {
    return mix64(pseed + (uint64_t)j * 0xD1342543DE82EF95ULL);
}
__host__ __device__ inline int panel_gij(uint64_t colkey, int64_t i, uint32_t thr) {
    uint64_t h = mix64(colkey ^ ((uint64_t)i * NGP_GOLD + 0x632BE59BD9B4E019ULL));
    return (int)((uint32_t)h < thr) + (int)((uint32_t)(h >> 32) < thr);
}

}  // namespace ngp
