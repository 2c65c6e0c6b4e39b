// bf16 matrix-core variant of the query side of a linear-attention layer (la_fused.hip:la_query_layer_kernel; reference
// models/LinearAttention.py:38-73; SURVEY.md 7 step 9, BASELINE configs[2] / [3]):
//   Q = elu(Wq x) + 1 -> message = Q KV / (Q . Ksum + eps) * S -> merge -> LayerNorm -> mlp(cat[x, message]) (128 -> 128 ReLU -> 64)
//   -> LayerNorm -> x + .
// Same contract as cmr_la_query_layer_f32 (fp32 rows, weights, state).  The four GEMMs run on v_mfma_f32_32x32x16_bf16 with fp32
// accumulation: 32 matrix instructions per 32 rows instead of 512, which leaves the per-head state product, the two LayerNorms
// (all fp32, as before) and the row traffic as the cost of the layer.  The chain still never leaves the registers:
//   * a GEMM whose input comes from global memory (x) takes natural-order fragments: k step s = x[row][16 s + 8 h .. + 7];
//   * a GEMM whose input is the previous GEMM's accumulator tile takes registers 8 s'' .. 8 s'' + 7 of tile t as they stand -- the
//     channels 32 t + 8 (2 s'' + (j >> 2)) + 4 h + (j & 3) -- and its weight fragments are written to LDS with their k slots
//     permuted to that order (la_stage_frags, ORDER_ACC);
//   * the residual needs x in the OUTPUT layout (channels 8 kg + 4 h .. + 3): half of those pieces are this lane's own loads,
//     the other half the partner lane's (lane ^ 32), one exchange per k step.
#include "cmr_common.h"

namespace {

typedef __bf16 lb_bf16x8 __attribute__((ext_vector_type(8)));

constexpr int LB_D = 64, LB_HID = 128, LB_STATE = 576;

__device__ __forceinline__ float lb_elu1(float v) { return v > 0.f ? v + 1.f : __builtin_amdgcn_exp2f(v * 1.4426950408889634f); }   // see la_fused.hip
__device__ __forceinline__ float lb_xhalf(float v) { return cmr_xhalf(v); }

__device__ __forceinline__ void lb_layernorm(f32x16 (&v)[2], const float* __restrict__ gs, const float* __restrict__ bs, int h, float eps) {
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += v[t][r];
  s += lb_xhalf(s);
  const float mean = s * (1.f / 64.f);
  float q = 0.f;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float d = v[t][r] - mean;
      v[t][r] = d;
      q += d * d;
    }
  q += lb_xhalf(q);
  const float rstd = 1.f / sqrtf(q * (1.f / 64.f) + eps);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(gs + 32 * t + 8 * qd + 4 * h);
      const f32x4 b = *reinterpret_cast<const f32x4*>(bs + 32 * t + 8 * qd + 4 * h);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[t][4 * qd + e] = v[t][4 * qd + e] * rstd * g[e] + b[e];
    }
}

// input channel of k slot (step s, lane half hh, element j): natural order, or the order in which an accumulator tile holds
// its channels (step s = tile s / 2, register half s % 2)
__device__ __forceinline__ int lb_kchan(bool acc_order, int s, int hh, int j) {
  return acc_order ? 32 * (s >> 1) + (j & 3) + 8 * (2 * (s & 1) + (j >> 2)) + 4 * hh : 16 * s + 8 * hh + j;
}

// W [N][K] fp32 (PyTorch [out][in]) -> A fragments [N / 32][K / 16][64 lanes][8 bf16]; steps >= acc_from use the accumulator order
// (relative to the first such step)
__device__ __forceinline__ void lb_stage_frags(lb_bf16x8* dst, const float* __restrict__ w, int N, int K, int acc_from, int tid) {
  const int S = K / 16;
  for (int e = tid; e < (N / 32) * S * 64; e += 512) {
    const int ln = e & 63, s = (e >> 6) % S, t = (e >> 6) / S;
    const int n = 32 * t + (ln & 31), hh = ln >> 5;
    lb_bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = s < acc_from ? lb_kchan(false, s, hh, j) : 16 * acc_from + lb_kchan(true, s - acc_from, hh, j);
      v[j] = (__bf16)w[(int64_t)n * K + c];
    }
    dst[e] = v;
  }
}

__device__ __forceinline__ lb_bf16x8 lb_pack_acc(const f32x16& a, int s2) {
  lb_bf16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (__bf16)a[8 * s2 + j];
  return r;
}

struct LbArgs {
  const float* x; int64_t ldx;
  const float* kv;
  const float *wq, *wm, *w0, *w3;
  const float *g1, *b1, *g2, *b2;
  float* out; int64_t ldo;
  uint32_t rows, L; int B;
  float s, eps, ln_eps;
};

__global__ __launch_bounds__(512) void la_query_layer_bf16_kernel(const LbArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  lb_bf16x8* Wq = reinterpret_cast<lb_bf16x8*>(smem_raw);      // [2][4][64]   x (natural order)
  lb_bf16x8* Wm = Wq + 2 * 4 * 64;                              // [2][4][64]   message (accumulator order)
  lb_bf16x8* W0 = Wm + 2 * 4 * 64;                              // [4][8][64]   steps 0..3 x, 4..7 merged message (accumulator order)
  lb_bf16x8* W3 = W0 + 4 * 8 * 64;                              // [2][8][64]   hidden (accumulator order)
  float* Ln = reinterpret_cast<float*>(W3 + 2 * 8 * 64);        // g1 | b1 | g2 | b2
  float* Kv = Ln + 4 * LB_D;                                    // [B][576]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  lb_stage_frags(Wq, a.wq, LB_D, LB_D, 4, tid);
  lb_stage_frags(Wm, a.wm, LB_D, LB_D, 0, tid);
  lb_stage_frags(W0, a.w0, LB_HID, LB_HID, 4, tid);
  lb_stage_frags(W3, a.w3, LB_D, LB_HID, 0, tid);
  if (tid < LB_D) {
    Ln[tid] = a.g1[tid]; Ln[LB_D + tid] = a.b1[tid]; Ln[2 * LB_D + tid] = a.g2[tid]; Ln[3 * LB_D + tid] = a.b2[tid];
  }
  for (int e = tid; e < a.B * (LB_STATE / 4); e += 512)
    *reinterpret_cast<f32x4*>(&Kv[e * 4]) = *reinterpret_cast<const f32x4*>(a.kv + e * 4);
  __syncthreads();

  const uint32_t ntiles = (a.rows + 31) / 32, tstride = gridDim.x * 8;
  auto load_x = [&](uint32_t tile, f32x4 (&lo)[4], f32x4 (&hi)[4]) {
    uint32_t row = tile * 32 + l31;
    row = (tile < ntiles && row < a.rows) ? row : 0;
    const float* xp = a.x + (int64_t)row * a.ldx + 8 * h;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      lo[s] = *reinterpret_cast<const f32x4*>(xp + 16 * s);
      hi[s] = *reinterpret_cast<const f32x4*>(xp + 16 * s + 4);
    }
  };
  uint32_t tile = blockIdx.x * 8 + wave;
  f32x4 xlo[4], xhi[4], nlo[4], nhi[4];
  load_x(tile, xlo, xhi);
  for (; tile < ntiles; tile += tstride) {
    const uint32_t row = tile * 32 + l31;
    const bool valid = row < a.rows;
    const uint32_t rowc = valid ? row : 0;
    load_x(tile + tstride, nlo, nhi);                    // next tile's rows fly under this tile's work
    const float* kvb = Kv + (rowc / a.L) * LB_STATE + 4 * h;
    lb_bf16x8 xb[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        xb[s][j] = (__bf16)xlo[s][j];
        xb[s][4 + j] = (__bf16)xhi[s][j];
      }
    }
    // ---- Q = elu(Wq x) + 1 ; message, head by head (head = tile t, quad qd) -- fp32, as in the fp32 kernel
    f32x16 msg[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) msg[t][r] = 0.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) msg[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wq[(t * 4 + s) * 64 + lane], xb[s], msg[t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int hd = 4 * t + qd;
        float qo[4], qp[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) qo[e] = lb_elu1(msg[t][4 * qd + e]);
#pragma unroll
        for (int e = 0; e < 4; ++e) qp[e] = lb_xhalf(qo[e]);
        float qd8[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          qd8[e] = h == 0 ? qo[e] : qp[e];
          qd8[4 + e] = h == 0 ? qp[e] : qo[e];
        }
        const f32x4 ks0 = *reinterpret_cast<const f32x4*>(kvb - 4 * h + 512 + hd * 8);
        const f32x4 ks1 = *reinterpret_cast<const f32x4*>(kvb - 4 * h + 512 + hd * 8 + 4);
        float den = 0.f;
        f32x4 num = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int d = 0; d < 8; ++d) {
          const f32x4 kvr = *reinterpret_cast<const f32x4*>(kvb + hd * 64 + d * 8);
          den += qd8[d] * (d < 4 ? ks0[d] : ks1[d - 4]);
#pragma unroll
          for (int e = 0; e < 4; ++e) num[e] += qd8[d] * kvr[e];
        }
        const float z = 1.f / (den + a.eps);
#pragma unroll
        for (int e = 0; e < 4; ++e) msg[t][4 * qd + e] = num[e] * z * a.s;
      }
    // ---- merge + LayerNorm 1
    f32x16 mrg[2];
    {
      lb_bf16x8 mb[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) mb[s] = lb_pack_acc(msg[s >> 1], s & 1);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) mrg[t][r] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) mrg[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wm[(t * 4 + s) * 64 + lane], mb[s], mrg[t], 0, 0, 0);
      }
    }
    lb_layernorm(mrg, Ln, Ln + LB_D, h, a.ln_eps);
    // ---- mlp: 128 -> 128 (ReLU) -> 64 on cat[x, message], LayerNorm 2
    f32x16 hid[4];
    {
      lb_bf16x8 mb[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) mb[s] = lb_pack_acc(mrg[s >> 1], s & 1);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) hid[t][r] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) hid[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W0[(t * 8 + s) * 64 + lane], xb[s], hid[t], 0, 0, 0);
#pragma unroll
        for (int s = 0; s < 4; ++s) hid[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W0[(t * 8 + 4 + s) * 64 + lane], mb[s], hid[t], 0, 0, 0);
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) hid[t][r] = hid[t][r] > 0.f ? hid[t][r] : 0.f;
    f32x16 o[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
#pragma unroll
      for (int s = 0; s < 8; ++s)
        o[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W3[(t * 8 + s) * 64 + lane], lb_pack_acc(hid[s >> 1], s & 1), o[t], 0, 0, 0);
    }
    lb_layernorm(o, Ln + 2 * LB_D, Ln + 3 * LB_D, h, a.ln_eps);
    // ---- residual: x in output layout.  Piece kg = 2 s (channels 16 s + 4 h ..): h = 0 own lo, h = 1 the partner's hi;
    //      piece kg = 2 s + 1 (channels 16 s + 8 + 4 h ..): h = 0 the partner's lo, h = 1 own hi.
    f32x4 ov[8];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      f32x4 send, recv;
#pragma unroll
      for (int e = 0; e < 4; ++e) send[e] = h == 0 ? xhi[s][e] : xlo[s][e];
#pragma unroll
      for (int e = 0; e < 4; ++e) recv[e] = lb_xhalf(send[e]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xe = h == 0 ? xlo[s][e] : recv[e];
        const float xo = h == 0 ? recv[e] : xhi[s][e];
        ov[2 * s][e] = xe + o[(2 * s) / 4][4 * ((2 * s) % 4) + e];
        ov[2 * s + 1][e] = xo + o[(2 * s + 1) / 4][4 * ((2 * s + 1) % 4) + e];
      }
    }
#pragma unroll
    for (int kg = 0; kg < 8; ++kg) cmr_pin(ov[kg]);
    if (valid) {
      float* yp = a.out + (int64_t)row * a.ldo + 4 * h;
#pragma unroll
      for (int kg = 0; kg < 8; ++kg) *reinterpret_cast<f32x4*>(yp + kg * 8) = ov[kg];
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      xlo[s] = nlo[s];
      xhi[s] = nhi[s];
    }
  }
}

}  // namespace

extern "C" int cmr_la_query_layer_bf16_f32(const float* x, int64_t ldx, const float* kvsum, const float* wq, const float* wmerge,
                                           const float* ln1_g, const float* ln1_b, const float* w_mlp0, const float* w_mlp3,
                                           const float* ln2_g, const float* ln2_b, float* out, int64_t ldo, int B, int L, int S,
                                           float eps, float ln_eps, hipStream_t stream) {
  CMR_REQUIRE(x && kvsum && wq && wmerge && ln1_g && ln1_b && w_mlp0 && w_mlp3 && ln2_g && ln2_b && out);
  CMR_REQUIRE(B > 0 && L > 0 && S > 0 && (int64_t)B * L < (int64_t)0x7fffffc0);
  CMR_REQUIRE(ldx % 4 == 0 && ldo % 4 == 0 && cmr_aligned16(x) && cmr_aligned16(out) && cmr_aligned16(kvsum));
  const size_t smem = (size_t)(2 * 4 + 2 * 4 + 4 * 8 + 2 * 8) * 1024 + (size_t)(4 * LB_D + (size_t)B * LB_STATE) * sizeof(float);
  if (smem > 160 * 1024) return CMR_EUNSUPPORTED;
  static CmrSmemCache granted{};
  if (cmr_grant_smem(reinterpret_cast<const void*>(la_query_layer_bf16_kernel), smem, granted) != CMR_OK) return CMR_ELAUNCH;
  const uint32_t rows = (uint32_t)((int64_t)B * L);
  const uint32_t ntiles = (rows + 31) / 32;
  uint32_t grid = (ntiles + 7) / 8;
  if (grid > 512) grid = 512;
  const LbArgs a{x, ldx, kvsum, wq, wmerge, w_mlp0, w_mlp3, ln1_g, ln1_b, ln2_g, ln2_b, out, ldo, rows, (uint32_t)L, B, (float)S, eps, ln_eps};
  hipLaunchKernelGGL(la_query_layer_bf16_kernel, dim3(grid), dim3(512), smem, stream, a);
  return cmr_launch_status();
}
