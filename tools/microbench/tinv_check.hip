// k_tinv against a plain host restatement of the same forward substitution (bit for bit), for K = 0 (BayesPR) and Tuple blocks k = 1..4:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I nextgp.jl_amd/csrc -o tinv_check tools/microbench/tinv_check.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
#include "ngp_kernels.h"
using namespace ngp;
int main() {
    const int NB = 5, D = 2;
    const long long Ppad = NB * 64;
    std::mt19937_64 rng(7);
    std::normal_distribution<double> nd(0.0, 1.0);
    std::vector<double> gram((size_t)NB * D * 4096, 0.0), c(Ppad), tupc(4 * Ppad), tinv((size_t)NB * 4096, -7.0);
    std::vector<unsigned> blin = {1u, 2u, 3u, 4u, 5u};
    for (int t = 0; t < NB; t++)
        for (int m = 0; m < 64; m++)
            for (int j = m + 1; j < 64; j++) gram[((size_t)t * D) * 4096 + m * 64 + j] = 0.05 * nd(rng);
    for (auto &v : c) v = 0.5 + 0.1 * nd(rng);
    for (auto &v : tupc) v = 0.3 * nd(rng);
    double *dg, *dc, *dt, *dtc; unsigned *db;
    (void)hipMalloc(&dg, gram.size() * 8); (void)hipMalloc(&dc, c.size() * 8); (void)hipMalloc(&dt, tinv.size() * 8); (void)hipMalloc(&dtc, tupc.size() * 8); (void)hipMalloc(&db, NB * 4);
    (void)hipMemcpy(dg, gram.data(), gram.size() * 8, hipMemcpyHostToDevice); (void)hipMemcpy(dc, c.data(), c.size() * 8, hipMemcpyHostToDevice);
    (void)hipMemcpy(dtc, tupc.data(), tupc.size() * 8, hipMemcpyHostToDevice); (void)hipMemcpy(db, blin.data(), NB * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dt, tinv.data(), tinv.size() * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_tinv, dim3(NB), dim3(64), 0, 0, (const double *)dg, D, (const double *)dc, (const unsigned *)db, dt, (const unsigned *)nullptr, (const double *)dtc, Ppad);
    hipError_t e = hipDeviceSynchronize();
    printf("launch: %s\n", hipGetErrorString(e));
    (void)hipMemcpy(tinv.data(), dt, tinv.size() * 8, hipMemcpyDeviceToHost);
    for (int t = 0; t < NB; t++) {
        const int K = (int)blin[t] - 1;  // 0 symbol, 1..4 tuple
        const double *G1 = &gram[((size_t)t * D) * 4096];
        long long bad = 0; double worst = 0; static char mism[64][64];
        for (int i = 0; i < 64; i++) {
            double acc[64], x[64];
            for (int j = 0; j < 64; j++) acc[j] = 0.0;
            for (int m = 0; m < 64; m++) {
                double xm;
                if (m < i) xm = 0.0; else if (m == i) xm = 1.0;
                else if (K == 0) { double tt = c[t * 64 + m] * acc[m]; xm = -tt; }
                else { const int g = (m / K) * K; double tt = 0.0;
                       if (g + K <= 64) { tt = tupc[t * 64 + m] * acc[g]; for (int b = 1; b < K; b++) tt = std::fma(tupc[b * Ppad + t * 64 + m], acc[g + b], tt); }
                       xm = -tt; }
                x[m] = xm;
                for (int j = m + 1; j < 64; j++) { const int gj = K <= 1 ? j : (j / K) * K; if (gj > m) acc[j] = std::fma(G1[m * 64 + j], xm, acc[j]); }
            }
            for (int m = 0; m < 64; m++) { const double dv = tinv[(size_t)t * 4096 + i * 64 + m]; mism[m][i] = !(dv == x[m]); if (!(dv == x[m])) { bad++; worst = std::fmax(worst, std::fabs(dv - x[m])); } }
        }
        printf("block %d (code %u): %lld of 4096 entries differ, worst %.3g\n", t, blin[t], bad, worst);
        if (t == 0) {
            // pattern: for each row m, the columns i whose entry differs
            for (int m = 0; m < 64; m++) {
                int cnt = 0, first = -1, last = -1;
                for (int i = 0; i < 64; i++) {
                    double acc[64]; (void)acc;
                }
                (void)cnt; (void)first; (void)last;
                printf("row m=%2d: ", m); for (int i = 0; i < 64; i++) putchar(mism[m][i] ? 'X' : '.'); putchar('\n');
            }
        }
    }
    {   // time: 9375 blocks (the 50k x 600k panel) of BayesPR blocks, the same Gram block for all
        const int NT = 9375;
        double *dg2, *dc2, *dt2; unsigned *db2;
        (void)hipMalloc(&dg2, (size_t)NT * 4096 * 8); (void)hipMalloc(&dc2, (size_t)NT * 64 * 8); (void)hipMalloc(&dt2, (size_t)NT * 4096 * 8); (void)hipMalloc(&db2, NT * 4);
        for (int t = 0; t < NT; t++) (void)hipMemcpy(dg2 + (size_t)t * 4096, gram.data(), 4096 * 8, hipMemcpyHostToDevice);
        (void)hipMemset(dc2, 0, (size_t)NT * 64 * 8);
        std::vector<unsigned> ones(NT, 1u);
        (void)hipMemcpy(db2, ones.data(), NT * 4, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int rep = 0; rep < 3; rep++) {
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k_tinv, dim3(NT), dim3(64), 0, 0, (const double *)dg2, 1, (const double *)dc2, (const unsigned *)db2, dt2, (const unsigned *)nullptr, (const double *)dtc, Ppad);
            (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            printf("k_tinv, %d blocks: %.3f ms\n", NT, ms);
        }
    }
    return 0;
}
