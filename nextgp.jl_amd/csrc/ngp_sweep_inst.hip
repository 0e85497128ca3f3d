// ngp_sweep_inst.hip -- one instantiation of the persistent sweep kernel (ngp_sweep.h) and its host-side launch stubs.
// Compiled four times: -DNGP_INST_DBG=0 (lean production kernel, K chains per pass), =1 (diagnostic kernel: time stamps, timing modes;
// tall fp32 panels), =2 (full production kernel: Tuple sets, tall shards), =3 (models with a BayesR set).
#include <hip/hip_runtime.h>

#include "ngp_sweep.h"

#ifndef NGP_INST_DBG
#error "compile with -DNGP_INST_DBG=0, 1, 2 or 3"
#endif

namespace ngp {

#if NGP_INST_DBG == 3
// the production kernel of models with a BayesR set (its coefficients fetched one block ahead through LDS): a translation unit of its own
hipError_t sweep_r_set_max_lds(int bytes) {
    return hipFuncSetAttribute((const void *)k_sweep_r, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}
void sweep_r_launch(unsigned grid, size_t lds_bytes, hipStream_t stream, const SweepArgs &A) {
    hipLaunchKernelGGL(k_sweep_r, dim3(grid), dim3(NGP_WG), lds_bytes, stream, A);
}
// ... and K such chains per pass
hipError_t sweep_multi_r_set_max_lds(int bytes) {
    return hipFuncSetAttribute((const void *)k_sweep_multi_r, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}
void sweep_multi_r_launch(unsigned grid, size_t lds_bytes, hipStream_t stream, const MultiArgs &M) {
    hipLaunchKernelGGL(k_sweep_multi_r, dim3(grid), dim3(NGP_WG), lds_bytes, stream, M);
}
#elif NGP_INST_DBG == 2
// the production kernel of models with a Tuple (correlated BayesPR) set: a translation unit of its own (the units compile in parallel)
hipError_t sweep_tup_set_max_lds(int bytes) {
    return hipFuncSetAttribute((const void *)k_sweep_tup, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}
void sweep_tup_launch(unsigned grid, size_t lds_bytes, hipStream_t stream, const SweepArgs &A) {
    hipLaunchKernelGGL(k_sweep_tup, dim3(grid), dim3(NGP_WG), lds_bytes, stream, A);
}
// ... and K such chains per pass
hipError_t sweep_multi_tup_set_max_lds(int bytes) {
    return hipFuncSetAttribute((const void *)k_sweep_multi_tup, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}
void sweep_multi_tup_launch(unsigned grid, size_t lds_bytes, hipStream_t stream, const MultiArgs &M) {
    hipLaunchKernelGGL(k_sweep_multi_tup, dim3(grid), dim3(NGP_WG), lds_bytes, stream, M);
}
#else
#if NGP_INST_DBG
#define NGP_SFX(name) name##_1
#else
#define NGP_SFX(name) name##_0
#endif
static constexpr bool kDbg = (NGP_INST_DBG != 0);

hipError_t NGP_SFX(sweep_set_max_lds)(int bytes) {
    return hipFuncSetAttribute((const void *)k_sweep<kDbg>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}
#if !NGP_INST_DBG
// K chains per pass: the production instantiation only
hipError_t sweep_multi_set_max_lds(int bytes) {
    return hipFuncSetAttribute((const void *)k_sweep_multi, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}
void sweep_multi_launch(unsigned grid, size_t lds_bytes, hipStream_t stream, const MultiArgs &M) {
    hipLaunchKernelGGL(k_sweep_multi, dim3(grid), dim3(NGP_WG), lds_bytes, stream, M);
}
hipError_t sweep_occupancy_0(int *wg_per_cu, size_t lds_bytes) {
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(wg_per_cu, (const void *)k_sweep<false>, NGP_WG, lds_bytes);
}
#endif
#if NGP_INST_DBG
// several shards per streamer workgroup (tall fp32 panels): defined in this translation unit, the shorter one
hipError_t sweep_tall_set_max_lds(int bytes) {
    return hipFuncSetAttribute((const void *)k_sweep_tall, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}
hipError_t sweep_tall_occupancy(int *wg_per_cu, size_t lds_bytes) {
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(wg_per_cu, (const void *)k_sweep_tall, NGP_WG, lds_bytes);
}
void sweep_tall_launch(unsigned grid, size_t lds_bytes, hipStream_t stream, const SweepArgs &A) {
    hipLaunchKernelGGL(k_sweep_tall, dim3(grid), dim3(NGP_WG), lds_bytes, stream, A);
}
#endif
void NGP_SFX(sweep_launch)(unsigned grid, size_t lds_bytes, hipStream_t stream, const SweepArgs &A) {
    hipLaunchKernelGGL(k_sweep<kDbg>, dim3(grid), dim3(NGP_WG), lds_bytes, stream, A);
}
#endif

}  // namespace ngp
