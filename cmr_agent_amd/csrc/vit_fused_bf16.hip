// bf16 matrix-core variants of the two fused transformer-block kernels (vit_fused.hip; SURVEY.md 7 step 9, BASELINE configs[2] / [3]):
//
//   cmr_ln64_linear_bf16_f32   LayerNorm(64) + projection(s) of up to two row sets
//   cmr_vit_out_ffn_bf16_f32   x1 = ctx Wo + bo + x ;  out = x1 + W2 gelu(W1 LN(x1) + b1) + b2
//
// Same contracts as the fp32 kernels (fp32 rows, biases, LayerNorm parameters); the weights arrive as bf16 MFMA A fragments
// [n_out / 32][k / 16][64 lanes][8] (cmr_agent_amd/models/_pack.py:frag_pack_bf16: one 1 KB coalesced wave load per matrix
// instruction, straight from L2) and the products run on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  LayerNorm, GELU, biases
// and residuals are fp32.  Fragment orders:
//   * rows read from memory as GEMM operands (x of the projections, ctx of the out-projection) are natural-order fragments: k step s =
//     row[16 s + 8 h .. + 7] (frag_pack_bf16(w));
//   * operands that live in the accumulator layout -- LN(x1), which is built from the out-projection's accumulators, and the hidden
//     activations -- are taken as they stand (registers 8 s'' .. 8 s'' + 7 of tile t = channels 32 t + 8 (2 s'' + (j >> 2)) + 4 h + (j & 3))
//     against weights whose k slots are packed in that order (frag_pack_bf16(w, acc_order=True)).
#include "cmr_common.h"

namespace {

typedef __bf16 vb_bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float vb_xhalf(float v) { return cmr_xhalf(v); }
__device__ __forceinline__ float vb_gelu(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f)); }

// LayerNorm over the 64 channels of a row held as 8 float4 pieces (4 in this lane, 4 in lane ^ 32); chan(i) = first channel of piece i
template <typename CH>
__device__ __forceinline__ void vb_layernorm(f32x4 (&v)[8], const float* __restrict__ g, const float* __restrict__ b, float eps, CH chan) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
  s += vb_xhalf(s);
  const float mean = s * (1.f / 64.f);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d = v[i][e] - mean;
      v[i][e] = d;
      q += d * d;
    }
  q += vb_xhalf(q);
  const float rstd = 1.f / sqrtf(q * (1.f / 64.f) + eps);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const f32x4 gv = *reinterpret_cast<const f32x4*>(g + chan(i));
    const f32x4 bv = *reinterpret_cast<const f32x4*>(b + chan(i));
#pragma unroll
    for (int e = 0; e < 4; ++e) v[i][e] = v[i][e] * rstd * gv[e] + bv[e];
  }
}

__device__ __forceinline__ vb_bf16x8 vb_pack(const f32x4& a, const f32x4& b) {
  vb_bf16x8 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    r[i] = (__bf16)a[i];
    r[4 + i] = (__bf16)b[i];
  }
  return r;
}

struct VbProblem {
  const float* x; int64_t ldx; uint32_t rows;
  const vb_bf16x8* wf; const float* bias; int npair;   // n_out = 64 * npair; wf [2 npair][4][64]
  float* y; int64_t ldy;
};
struct VbLnArgs {
  VbProblem p[2];
  uint32_t tiles0, tiles;
  const float* g; const float* b; float eps;
};

__global__ __launch_bounds__(256) void ln64_linear_bf16_kernel(const VbLnArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  const uint32_t gt = blockIdx.x * 4 + wave;
  if (gt >= a.tiles) return;
  const VbProblem& P = gt < a.tiles0 ? a.p[0] : a.p[1];
  const uint32_t tile = gt < a.tiles0 ? gt : gt - a.tiles0;
  const uint32_t row = tile * 32 + l31;
  const bool valid = row < P.rows;
  const float* xp = P.x + (int64_t)(valid ? row : 0) * P.ldx + 8 * h;
  f32x4 xf[8];                                           // piece 2 s + half = channels 16 s + 8 h + 4 half .. + 3
#pragma unroll
  for (int i = 0; i < 8; ++i) xf[i] = *reinterpret_cast<const f32x4*>(xp + 16 * (i >> 1) + 4 * (i & 1));
  vb_layernorm(xf, a.g, a.b, a.eps, [&](int i) { return 16 * (i >> 1) + 8 * h + 4 * (i & 1); });
  vb_bf16x8 xb[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) xb[s] = vb_pack(xf[2 * s], xf[2 * s + 1]);
  for (int pr = 0; pr < P.npair; ++pr) {
    vb_bf16x8 wv[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int s = 0; s < 4; ++s) wv[t][s] = P.wf[((2 * pr + t) * 4 + s) * 64 + lane];
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wv[t][s], xb[s], acc[t], 0, 0, 0);
    }
    f32x4 ov[8];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(P.bias + 64 * pr + 32 * t + 8 * qd + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) ov[4 * t + qd][e] = acc[t][4 * qd + e] + bv[e];
      }
#pragma unroll
    for (int i = 0; i < 8; ++i) cmr_pin(ov[i]);
    if (valid) {
      float* yp = P.y + (int64_t)row * P.ldy + 64 * pr + 4 * h;
#pragma unroll
      for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(yp + 8 * i) = ov[i];
    }
  }
}

struct VbFfnArgs {
  const float* ctx; int64_t ldc;
  const float* x; int64_t ldx;
  const vb_bf16x8* wo_f; const float* bo;              // [2][4][64] natural order
  const float* g2; const float* b2n; float eps;
  const vb_bf16x8* w1_f; const float* b1;              // [32][4][64] accumulator order
  const vb_bf16x8* w2_f; const float* b2;              // [2][64][64] accumulator order
  float* out; int64_t ldo; uint32_t rows;
};

// One 32-row tile per workgroup of 8 waves: every wave redoes the small out-projection and LayerNorm, takes 128 of the 1024
// hidden units through fc1 / GELU / its K slice of fc2, and the 8 partial outputs are summed through LDS in a fixed order.
__global__ __launch_bounds__(512) void vit_out_ffn_bf16_kernel(const VbFfnArgs a) {
  __shared__ __attribute__((aligned(16))) float red[7 * 8 * 64 * 4];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  const uint32_t row = blockIdx.x * 32 + l31;
  const bool valid = row < a.rows;
  const uint32_t rowc = valid ? row : 0;
  const float* cp = a.ctx + (int64_t)rowc * a.ldc + 8 * h;
  const float* xp = a.x + (int64_t)rowc * a.ldx + 4 * h;
  // this wave's weight fragments of the out-projection and of fc1 are requested up front (8 + 16 KB-sized wave loads in flight while
  // the rows arrive); those of fc2 follow once fc1 has released its registers (256 registers per wave at 2 waves / SIMD)
  vb_bf16x8 wo[2][4], w1[4][4], w2[2][8];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int s = 0; s < 4; ++s) wo[t][s] = a.wo_f[(t * 4 + s) * 64 + lane];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int s = 0; s < 4; ++s) w1[t][s] = a.w1_f[((4 * wave + t) * 4 + s) * 64 + lane];
  f32x4 cl[4], ch[4], x1[8];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    cl[s] = *reinterpret_cast<const f32x4*>(cp + 16 * s);
    ch[s] = *reinterpret_cast<const f32x4*>(cp + 16 * s + 4);
  }
#pragma unroll
  for (int kg = 0; kg < 8; ++kg) x1[kg] = *reinterpret_cast<const f32x4*>(xp + kg * 8);      // residual stream, output layout
  // ---- x1 = ctx Wo + bo + x
  {
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wo[t][s], vb_pack(cl[s], ch[s]), acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int kg = 0; kg < 8; ++kg) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bo + 8 * kg + 4 * h);
#pragma unroll
      for (int e = 0; e < 4; ++e) x1[kg][e] = (acc[kg / 4][4 * (kg % 4) + e] + bv[e]) + x1[kg][e];
    }
  }
  f32x4 xn[8];
#pragma unroll
  for (int kg = 0; kg < 8; ++kg) xn[kg] = x1[kg];
  vb_layernorm(xn, a.g2, a.b2n, a.eps, [&](int i) { return 8 * i + 4 * h; });
  // LN(x1) sits in the accumulator layout: k step (t, s'') = pieces 4 t + 2 s'', 4 t + 2 s'' + 1
  vb_bf16x8 nb[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) nb[s] = vb_pack(xn[2 * s], xn[2 * s + 1]);
  // ---- this wave's 128 hidden units: fc1 + GELU, then its K slice of fc2
  f32x16 hid[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) hid[t][r] = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) hid[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1[t][s], nb[s], hid[t], 0, 0, 0);
  }
  // fc2's fragments take over the registers fc1's have released; the GELU below covers their latency
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int s = 0; s < 8; ++s) w2[n][s] = a.w2_f[(n * 64 + 8 * wave + s) * 64 + lane];
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(a.b1 + 128 * wave + 32 * t + 8 * qd + 4 * h);
#pragma unroll
      for (int e = 0; e < 4; ++e) hid[t][4 * qd + e] = vb_gelu(hid[t][4 * qd + e] + bv[e]);
    }
    __builtin_amdgcn_sched_barrier(0);        // one tile's biases / erf temporaries at a time (register budget)
  }
  f32x16 part[2];
#pragma unroll
  for (int n = 0; n < 2; ++n) {
#pragma unroll
    for (int r = 0; r < 16; ++r) part[n][r] = 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      vb_bf16x8 hb;
#pragma unroll
      for (int j = 0; j < 8; ++j) hb[j] = (__bf16)hid[s >> 1][8 * (s & 1) + j];
      part[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2[n][s], hb, part[n], 0, 0, 0);
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const f32x4 v = {part[i / 4][4 * (i % 4)], part[i / 4][4 * (i % 4) + 1], part[i / 4][4 * (i % 4) + 2], part[i / 4][4 * (i % 4) + 3]};
      *reinterpret_cast<f32x4*>(&red[(((wave - 1) * 8 + i) * 64 + lane) * 4]) = v;
    }
  }
  __syncthreads();
  if (wave != 0) return;
  f32x4 ov[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    f32x4 s = {part[i / 4][4 * (i % 4)], part[i / 4][4 * (i % 4) + 1], part[i / 4][4 * (i % 4) + 2], part[i / 4][4 * (i % 4) + 3]};
#pragma unroll
    for (int w = 0; w < 7; ++w) s += *reinterpret_cast<const f32x4*>(&red[((w * 8 + i) * 64 + lane) * 4]);
    const f32x4 bv = *reinterpret_cast<const f32x4*>(a.b2 + 8 * i + 4 * h);
#pragma unroll
    for (int e = 0; e < 4; ++e) ov[i][e] = (s[e] + bv[e]) + x1[i][e];
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) cmr_pin(ov[i]);
  if (valid) {
    float* yp = a.out + (int64_t)row * a.ldo + 4 * h;
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(yp + 8 * i) = ov[i];
  }
}

}  // namespace

extern "C" int cmr_ln64_linear_bf16_f32(const float* x, int64_t ldx, int64_t rows_x, const void* wf_x, const float* bias_x, int n_out_x,
                                        float* out_x, int64_t ldo_x, const float* y, int64_t ldy, int64_t rows_y, const void* wf_y,
                                        const float* bias_y, int n_out_y, float* out_y, int64_t ldo_y, const float* gamma, const float* beta,
                                        float eps, hipStream_t stream) {
  CMR_REQUIRE(x && wf_x && bias_x && out_x && gamma && beta && rows_x > 0 && rows_x < (int64_t)0x7fffffc0);
  CMR_REQUIRE(n_out_x > 0 && n_out_x % 64 == 0 && ldx % 4 == 0 && ldo_x % 4 == 0 && cmr_aligned16(x) && cmr_aligned16(wf_x) &&
              cmr_aligned16(bias_x) && cmr_aligned16(out_x) && cmr_aligned16(gamma) && cmr_aligned16(beta));
  VbLnArgs a{};
  a.p[0] = VbProblem{x, ldx, (uint32_t)rows_x, (const vb_bf16x8*)wf_x, bias_x, n_out_x / 64, out_x, ldo_x};
  a.tiles0 = (uint32_t)((rows_x + 31) / 32);
  a.tiles = a.tiles0;
  if (y) {
    CMR_REQUIRE(wf_y && bias_y && out_y && rows_y > 0 && rows_y < (int64_t)0x7fffffc0 && n_out_y > 0 && n_out_y % 64 == 0);
    CMR_REQUIRE(ldy % 4 == 0 && ldo_y % 4 == 0 && cmr_aligned16(y) && cmr_aligned16(wf_y) && cmr_aligned16(bias_y) && cmr_aligned16(out_y));
    a.p[1] = VbProblem{y, ldy, (uint32_t)rows_y, (const vb_bf16x8*)wf_y, bias_y, n_out_y / 64, out_y, ldo_y};
    a.tiles += (uint32_t)((rows_y + 31) / 32);
  } else {
    a.p[1] = a.p[0];
  }
  a.g = gamma; a.b = beta; a.eps = eps;
  hipLaunchKernelGGL(ln64_linear_bf16_kernel, dim3((a.tiles + 3) / 4), dim3(256), 0, stream, a);
  return cmr_launch_status();
}

extern "C" int cmr_vit_out_ffn_bf16_f32(const float* ctx, int64_t ldc, const float* x, int64_t ldx, const void* wo_f, const float* bo,
                                        const float* ln_g, const float* ln_b, float eps, const void* w1_f, const float* b1, const void* w2_f,
                                        const float* b2, float* out, int64_t ldo, int64_t rows, hipStream_t stream) {
  CMR_REQUIRE(ctx && x && wo_f && bo && ln_g && ln_b && w1_f && b1 && w2_f && b2 && out && rows > 0 && rows < (int64_t)0x7fffffc0);
  CMR_REQUIRE(ldc % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && cmr_aligned16(ctx) && cmr_aligned16(x) && cmr_aligned16(out) &&
              cmr_aligned16(wo_f) && cmr_aligned16(w1_f) && cmr_aligned16(w2_f) && cmr_aligned16(bo) && cmr_aligned16(b1) &&
              cmr_aligned16(b2) && cmr_aligned16(ln_g) && cmr_aligned16(ln_b));
  const VbFfnArgs a{ctx, ldc, x, ldx, (const vb_bf16x8*)wo_f, bo, ln_g, ln_b, eps, (const vb_bf16x8*)w1_f, b1, (const vb_bf16x8*)w2_f, b2,
                    out, ldo, (uint32_t)rows};
  hipLaunchKernelGGL(vit_out_ffn_bf16_kernel, dim3((unsigned)((rows + 31) / 32)), dim3(512), 0, stream, a);
  return cmr_launch_status();
}
