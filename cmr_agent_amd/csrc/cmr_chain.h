// Building block of the layer-level kernels (la_fused.hip, vecattn_fused.hip): one GEMM of a chain computed TRANSPOSED
// (D'[channel][row], weights = MFMA A operand from LDS, rows = B operand from registers), so that the accumulators of
// one GEMM -- register 4q+e of tile t = channel 32t + 8q + 4h + e -- are the B fragments (k-group 4t+q) of the next.
#pragma once
#include "cmr_common.h"

// acc[t] = sum_kg sum_j W[32t + l31][8kg + 4h + j] * bfrag(kg, j)   for kg < KG, j < 4.  Ws: LDS, row stride LD floats
// (LD = K + 4: conflict-free ds_read_b128).  Weight fragments are software-pipelined one k-group ahead; the scheduling
// barrier keeps hipcc from hoisting every LDS read of the unrolled loop (hundreds of VGPRs).
template <int T, int KG, int LD, typename BF>
__device__ __forceinline__ void cmr_chain_gemm(const float* __restrict__ Ws, int l31, int h, f32x16 (&acc)[T], BF bfrag) {
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const float* wrow = Ws + l31 * LD + 4 * h;
  f32x4 wc[T], wn[T];
#pragma unroll
  for (int t = 0; t < T; ++t) wc[t] = *reinterpret_cast<const f32x4*>(wrow + t * 32 * LD);
#pragma unroll
  for (int kg = 0; kg < KG; ++kg) {
    if (kg + 1 < KG) {
#pragma unroll
      for (int t = 0; t < T; ++t) wn[t] = *reinterpret_cast<const f32x4*>(wrow + t * 32 * LD + (kg + 1) * 8);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float b = bfrag(kg, j);
#pragma unroll
      for (int t = 0; t < T; ++t) acc[t] = cmr_mfma32(wc[t][j], b, acc[t]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < T; ++t) wc[t] = wn[t];
  }
}

// acc[t][4q + e] += bias[32t + 8q + 4h + e] (bias in LDS), then optional ReLU
template <int T>
__device__ __forceinline__ void cmr_chain_bias(f32x16 (&acc)[T], const float* __restrict__ bs, int h, bool relu) {
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(bs + 32 * t + 8 * q + 4 * h);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = acc[t][4 * q + e] + b[e];
        acc[t][4 * q + e] = relu ? (v > 0.f ? v : 0.f) : v;
      }
    }
}
