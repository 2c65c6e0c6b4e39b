// Building blocks of the row-tile kernels on v_mfma_f32_16x16x4_f32 (train-mode transformer block, csrc/vit_train.hip).
//
// Operand layout (as in csrc/vit_fused.hip): weights are the A operand, lane = 16 g + m supplies W[m][4 ks + g]; activations are the B
// operand, lane = 16 g + n supplies x[row n][4 ks + g]; D: lane 16 g + n holds channels 4 g + r (r = 0..3) of row n.  A lane therefore owns
// ONE row and, of every 16-channel tile T, the channels 16 T + 4 g + r -- again the B operand of k-tile T of the next GEMM when its weights
// are stored in that order ([n_out/16][k/16][64 lanes][4], lane = 16 g + m holding W[16 To + m][16 T + 4 g + r]: frag16 layout,
// cmr_pack_frags_f32 kind 1 = cmr_agent_amd/models/_pack.py:frag_pack16).  Rows read from memory are read in the same order (float4 at
// channel 16 T + 4 g).
#pragma once
#include "cmr_common.h"

__device__ __forceinline__ f32x4 m16_mfma(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// acc[To] (+)= sum over k-tiles T (KT of them, from kt0) and r of Wf[tile0 + To][kt0 + T][lane][r] * bfrag(T, r); fragments requested D
// k-tiles ahead through a register ring (they come straight from L2).  ZERO: clear the accumulators first.
template <int TO, int KT, int D, bool ZERO = true, typename BF>
__device__ __forceinline__ void m16_gemm(const float* __restrict__ wf, int kt_total, int tile0, int kt0, int lane, f32x4 (&acc)[TO], BF bfrag) {
  static_assert(D >= 1 && D <= KT, "prefetch depth");
  if (ZERO) {
#pragma unroll
    for (int t = 0; t < TO; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float* wp = wf + ((int64_t)tile0 * kt_total + kt0) * 256 + lane * 4;
  const int64_t tstride = (int64_t)kt_total * 256;
  f32x4 ring[D][TO];
#pragma unroll
  for (int d = 0; d < D; ++d)
#pragma unroll
    for (int t = 0; t < TO; ++t) ring[d][t] = *reinterpret_cast<const f32x4*>(wp + t * tstride + d * 256);
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float b = bfrag(kt, r);
#pragma unroll
      for (int t = 0; t < TO; ++t) acc[t] = m16_mfma(ring[kt % D][t][r], b, acc[t]);
    }
    if (kt + D < KT) {
#pragma unroll
      for (int t = 0; t < TO; ++t) ring[kt % D][t] = *reinterpret_cast<const f32x4*>(wp + t * tstride + (kt + D) * 256);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// sum over the four lane groups g (lanes l, l ^ 16, l ^ 32, l ^ 48): the 64 channels of a row
__device__ __forceinline__ float m16_allg(float v) {
  v += cmr_xor16(v);
  return v + cmr_xhalf(v);
}

// sum over the 16 lanes of a DPP row (= the 16 rows n of a tile at fixed lane group g); every lane ends with the total.  Quad permutes,
// then the two mirrors: fixed order, nothing through LDS.
template <int CTRL>
__device__ __forceinline__ float m16_dppf(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float m16_sum16(float v) {
  v += m16_dppf<0xB1>(v);       // quad_perm:[1,0,3,2]
  v += m16_dppf<0x4E>(v);       // quad_perm:[2,3,0,1]
  v += m16_dppf<0x141>(v);      // row_half_mirror
  v += m16_dppf<0x140>(v);      // row_mirror
  // The result must exist BEFORE any divergent branch that follows (typically `if (lane % 16 == 0) store`): hipcc sinks a DPP chain whose
  // only use sits in such a branch into it, where the disabled lanes then read as 0 (bound_ctrl) -- seen as wrong sums of exactly the last
  // element reduced in front of the branch (bn_linear_fwd_kernel<2, *>).  The empty asm is a use in the dominating block.
  asm volatile("" : "+v"(v));
  return v;
}

__device__ __forceinline__ float m16_gelu(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f)); }
// gelu(v) and its derivative from ONE erf: Phi = 0.5 (1 + erf(v / sqrt 2)); gelu = v Phi (the same bits as m16_gelu: 0.5 v (1 + erf) is
// evaluated as written); gelu' = Phi + v phi(v)
__device__ __forceinline__ void m16_gelu_both(float v, float& g, float& dg) {
  const float e = 1.f + erff(v * 0.70710678118654752440f);
  g = 0.5f * v * e;
  dg = 0.5f * e + v * 0.39894228040143267794f * expf(-0.5f * v * v);
}
// d/dv gelu(v) = Phi(v) + v phi(v)
__device__ __forceinline__ float m16_gelu_grad(float v) {
  return 0.5f * (1.f + erff(v * 0.70710678118654752440f)) + v * 0.39894228040143267794f * expf(-0.5f * v * v);
}

// counter-based dropout of one site (csrc/cmr_common.h:cmr_keep with the wave-uniform half hoisted): keep(idx) and the 1 / (1 - p) scale
struct M16Drop {
  uint64_t key;
  uint32_t thr;
  float scale;
  __device__ __forceinline__ void init(const int64_t* seed, uint64_t site, uint32_t thr_, float scale_) {
    key = cmr_mix64((uint64_t)seed[0] + site * 0x9E3779B97F4A7C15ull);
    thr = thr_;
    scale = scale_;
  }
  __device__ __forceinline__ float mul(uint64_t idx) const { return (uint32_t)cmr_mix64(key ^ idx) >= thr ? scale : 0.f; }
};
