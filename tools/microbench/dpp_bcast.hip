// v_fmac_f64_dpp with row_newbcast:N -- src0 of every lane is lane N of its row of 16 (both dwords), fused like __builtin_fma.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/microbench/dpp_bcast.hip -o /tmp/dpp_bcast && /tmp/dpp_bcast
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(double *out, const double *d, const float *x) {
    const int lane = threadIdx.x & 63;
    const double dv = d[lane & 7];
    double p = 0.0;
#define STEP(N)                                                                                                     \
    {                                                                                                               \
        const double xv = (double)x[threadIdx.x * 8 + N];                                                           \
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #N " row_mask:0xf bank_mask:0xf" : "+v"(p) : "v"(dv), "v"(xv)); \
    }
    STEP(0) STEP(1) STEP(2) STEP(3) STEP(4) STEP(5) STEP(6) STEP(7)
    out[threadIdx.x] = p;
}
int main() {
    const int n = 256;
    std::vector<double> d(8), out(n), ref(n);
    std::vector<float> x(n * 8);
    for (int i = 0; i < 8; i++) d[i] = std::sin(1.0 + i) * 1e-3 + 1.0 / (3 + i);
    for (int i = 0; i < n * 8; i++) x[i] = (float)std::cos(0.37 * i) * 2.0f;
    for (int t = 0; t < n; t++) { double p = 0.0; for (int h = 0; h < 8; h++) p = std::fma((double)x[t * 8 + h], d[h], p); ref[t] = p; }
    double *dd, *dout; float *dx;
    hipMalloc(&dd, 64); hipMalloc(&dout, n * 8); hipMalloc(&dx, n * 32);
    hipMemcpy(dd, d.data(), 64, hipMemcpyHostToDevice); hipMemcpy(dx, x.data(), n * 32, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(n), 0, 0, dout, dd, dx);
    hipMemcpy(out.data(), dout, n * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < n; t++) if (out[t] != ref[t]) { if (bad < 5) printf("lane %d: %.17g vs %.17g\n", t, out[t], ref[t]); bad++; }
    printf("v_fmac_f64_dpp row_newbcast: %d of %d lanes differ from the sequential fma chain\n", bad, n);
    return bad != 0;
}
