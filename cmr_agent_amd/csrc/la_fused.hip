// Fused linear-attention layer of the fine matcher (reference models/LinearAttention.py:38-73), two kernels:
//
//   cmr_la_kv_state_f32    y rows -> K = elu(Wk y)+1, V = (Wv y)/S -> per (batch, head) state  KV = K^T V, Ksum
//   cmr_la_query_layer_f32 x rows -> Q = elu(Wq x)+1 -> message = Q KV / (Q.Ksum + eps) * S -> merge -> LayerNorm
//                          -> mlp(cat[x, message]) (128 -> 128 ReLU -> 64) -> LayerNorm -> x + .
//
// The unfused path makes 11 passes over the token rows per layer (each projection, the state reduction, the
// application, both LayerNorms and the MLP read and write [rows][64..128] arrays); here the source rows are
// read once and the query rows are read once and written once.  Every GEMM runs on v_mfma_f32_32x32x2_f32 with
// the weights resident in LDS; the query kernel computes TRANSPOSED (D'[channel][row], weights = A operand), so
// a lane owns one row, accumulator register 4q+e of tile t is channel 32t + 8q + 4h + e, and the accumulators of
// one GEMM are the B fragments (k-group 4t+q) of the next: the whole chain stays in registers.  A head is one
// k-group (8 channels, 4 in each lane half), so the 8x8 state product and the LayerNorm statistics need a single
// exchange with the partner lane (lane ^ 32).
#include "cmr_chain.h"

namespace {

constexpr int LA_D = 64, LA_HID = 128, LA_STATE = 576;    // 8 heads x (8x8 KV + 8 Ksum)
constexpr int LA_LD64 = LA_D + 4, LA_LD128 = LA_HID + 4;  // padded LDS weight rows (conflict-free b128 reads)

// F.elu(v) + 1; the negative branch as v_exp_f32(v log2 e) -- libm's expf costs five more instructions per value for a range reduction
// that buys nothing at |v| of a projected feature (state kernel 67 -> 62 us fp32, 39 -> 34 us bf16 at 8 x 26 752 rows)
__device__ __forceinline__ float la_elu1(float v) { return v > 0.f ? v + 1.f : __builtin_amdgcn_exp2f(v * 1.4426950408889634f); }
__device__ __forceinline__ float la_xhalf(float v) { return cmr_xhalf(v); }            // partner lane (other 4 dims of the head)

// LayerNorm over the 64 channels of this lane's row (32 here, 32 in the partner lane), in place
__device__ __forceinline__ void la_layernorm(f32x16 (&v)[2], const float* __restrict__ gs, const float* __restrict__ bs,
                                             int h, float eps) {
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += v[t][r];
  s += la_xhalf(s);
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
  q += la_xhalf(q);
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

struct LaQueryArgs {
  const float* x; int64_t ldx;
  const float* kv;                     // [B][576]
  const float *wq, *wm, *w0, *w3;      // [64][64], [64][64], [128][128], [64][128]  (PyTorch [out][in])
  const float *g1, *b1, *g2, *b2;      // LayerNorm 1 / 2
  float* out; int64_t ldo;
  uint32_t rows, L; int B;
  float s, eps, ln_eps;
  // train mode (TRAIN instance only): what the backward needs, written once per row -- qf = elu(Wq x) + 1, msg (the attention message),
  // mm = Wm msg (input of LayerNorm 1), d1 = drop(LN1(mm)), hid = drop(relu(W0 [x | d1])) [rows][128], o = drop(W3 hid) (input of
  // LayerNorm 2) -- and the three nn.Dropout sites of LinearAttention.py:28-34 / :61-70 (counter-based masks of cmr_dropout_f32)
  float *sv_qf, *sv_msg, *sv_mm, *sv_d1, *sv_hid, *sv_o;
  const int64_t* seed; uint64_t site_att, site_hid, site_out;
  uint32_t thr; float ks;
};

__device__ __forceinline__ float la_drop_mul(uint64_t key, uint64_t idx, uint32_t thr, float ks) {
  return (uint32_t)cmr_mix64(key ^ idx) >= thr ? ks : 0.f;
}

template <bool TRAIN>
__global__ __launch_bounds__(512) void la_query_layer_kernel(const LaQueryArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Wq = smem;                              // [64][68]
  float* Wm = Wq + LA_D * LA_LD64;               // [64][68]
  float* W0 = Wm + LA_D * LA_LD64;               // [128][132]
  float* W3 = W0 + LA_HID * LA_LD128;            // [64][132]
  float* Ln = W3 + LA_D * LA_LD128;              // g1 | b1 | g2 | b2
  float* Kv = Ln + 4 * LA_D;                     // [B][576]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  {
    // all 16 weight loads of a thread are issued before the first LDS store (one memory round trip, not 16)
    f32x4 wq4[2], wm4[2], w04[8], w34[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      wq4[i] = *reinterpret_cast<const f32x4*>(a.wq + (tid + 512 * i) * 4);
      wm4[i] = *reinterpret_cast<const f32x4*>(a.wm + (tid + 512 * i) * 4);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) w04[i] = *reinterpret_cast<const f32x4*>(a.w0 + (tid + 512 * i) * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) w34[i] = *reinterpret_cast<const f32x4*>(a.w3 + (tid + 512 * i) * 4);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = tid + 512 * i, n = e / (LA_D / 4), c = (e % (LA_D / 4)) * 4;
      *reinterpret_cast<f32x4*>(&Wq[n * LA_LD64 + c]) = wq4[i];
      *reinterpret_cast<f32x4*>(&Wm[n * LA_LD64 + c]) = wm4[i];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = tid + 512 * i, n = e / (LA_HID / 4), c = (e % (LA_HID / 4)) * 4;
      *reinterpret_cast<f32x4*>(&W0[n * LA_LD128 + c]) = w04[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + 512 * i, n = e / (LA_HID / 4), c = (e % (LA_HID / 4)) * 4;
      *reinterpret_cast<f32x4*>(&W3[n * LA_LD128 + c]) = w34[i];
    }
  }
  if (tid < LA_D) {
    Ln[tid] = a.g1[tid]; Ln[LA_D + tid] = a.b1[tid]; Ln[2 * LA_D + tid] = a.g2[tid]; Ln[3 * LA_D + tid] = a.b2[tid];
  }
  for (int e = tid; e < a.B * (LA_STATE / 4); e += 512)
    *reinterpret_cast<f32x4*>(&Kv[e * 4]) = *reinterpret_cast<const f32x4*>(a.kv + e * 4);
  __syncthreads();

  uint64_t key_att = 0, key_hid = 0, key_out = 0;
  const bool drop = TRAIN && a.seed != nullptr;
  if (drop) {
    const uint64_t sd = (uint64_t)a.seed[0];
    key_att = cmr_mix64(sd + a.site_att * 0x9E3779B97F4A7C15ull);
    key_hid = cmr_mix64(sd + a.site_hid * 0x9E3779B97F4A7C15ull);
    key_out = cmr_mix64(sd + a.site_out * 0x9E3779B97F4A7C15ull);
  }
  const uint32_t ntiles = (a.rows + 31) / 32;
  for (uint32_t tile = blockIdx.x * 8 + wave; tile < ntiles; tile += gridDim.x * 8) {
    const uint32_t row = tile * 32 + l31;
    const bool valid = row < a.rows;
    const uint32_t rowc = valid ? row : 0;       // rows past the end recompute row 0 and are not stored
    // ---- input fragments: B operand of the q projection and of the MLP's first half, and the residual
    const float* xp = a.x + (int64_t)rowc * a.ldx + 4 * h;
    f32x4 xf[8];
#pragma unroll
    for (int kg = 0; kg < 8; ++kg) xf[kg] = *reinterpret_cast<const f32x4*>(xp + kg * 8);
    const float* kvb = Kv + (rowc / a.L) * LA_STATE + 4 * h;

    // ---- Q = elu(Wq x) + 1 ; message = Q KV / (Q . Ksum + eps) * S, head by head (head = tile t, quad qd)
    f32x16 msg[2];
    cmr_chain_gemm<2, 8, LA_LD64>(Wq, l31, h, msg, [&](int kg, int j) { return xf[kg][j]; });
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const int hd = 4 * t + qd;
        float qo[4], qp[4];                      // this lane's 4 dims of the head (4h..4h+3), the partner's 4
#pragma unroll
        for (int e = 0; e < 4; ++e) qo[e] = la_elu1(msg[t][4 * qd + e]);
        if (TRAIN) *reinterpret_cast<f32x4*>(a.sv_qf + (int64_t)row * LA_D + 8 * hd + 4 * h) = f32x4{qo[0], qo[1], qo[2], qo[3]};
#pragma unroll
        for (int e = 0; e < 4; ++e) qp[e] = la_xhalf(qo[e]);
        // dims in natural order d = 0..7: lane half 0 owns 0..3, half 1 owns 4..7
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
          const f32x4 kvr = *reinterpret_cast<const f32x4*>(kvb + hd * 64 + d * 8);     // KV[hd][d][4h .. 4h+3]
          den += qd8[d] * (d < 4 ? ks0[d] : ks1[d - 4]);
#pragma unroll
          for (int e = 0; e < 4; ++e) num[e] += qd8[d] * kvr[e];
        }
        const float z = 1.f / (den + a.eps);
#pragma unroll
        for (int e = 0; e < 4; ++e) msg[t][4 * qd + e] = num[e] * z * a.s;
      }

    // ---- merge + LayerNorm 1 (+ dropout)
    f32x16 mrg[2];
    cmr_chain_gemm<2, 8, LA_LD64>(Wm, l31, h, mrg, [&](int kg, int j) { return msg[kg / 4][4 * (kg % 4) + j]; });
    if (TRAIN) {                                 // (the save buffers hold whole tiles: rows past the end are written, never read)
#pragma unroll
      for (int kg = 0; kg < 8; ++kg) {
        const int t = kg / 4, qd = kg % 4;
        *reinterpret_cast<f32x4*>(a.sv_msg + (int64_t)row * LA_D + 8 * kg + 4 * h) = f32x4{msg[t][4 * qd], msg[t][4 * qd + 1], msg[t][4 * qd + 2], msg[t][4 * qd + 3]};
        *reinterpret_cast<f32x4*>(a.sv_mm + (int64_t)row * LA_D + 8 * kg + 4 * h) = f32x4{mrg[t][4 * qd], mrg[t][4 * qd + 1], mrg[t][4 * qd + 2], mrg[t][4 * qd + 3]};
      }
    }
    la_layernorm(mrg, Ln, Ln + LA_D, h, a.ln_eps);
    if (TRAIN) {
#pragma unroll
      for (int kg = 0; kg < 8; ++kg) {
        const int t = kg / 4, qd = kg % 4;
        if (drop) {
#pragma unroll
          for (int e = 0; e < 4; ++e) mrg[t][4 * qd + e] *= la_drop_mul(key_att, (uint64_t)row * LA_D + 8 * kg + 4 * h + e, a.thr, a.ks);
        }
        *reinterpret_cast<f32x4*>(a.sv_d1 + (int64_t)row * LA_D + 8 * kg + 4 * h) = f32x4{mrg[t][4 * qd], mrg[t][4 * qd + 1], mrg[t][4 * qd + 2], mrg[t][4 * qd + 3]};
        __builtin_amdgcn_sched_barrier(0);     // one k-group at a time: interleaved, the 64-bit mask arithmetic of all of them spills
      }
    }

    // ---- mlp: 128 -> 128 (ReLU) -> 64 on cat[x, message], LayerNorm 2, residual
    f32x16 hid[4];
    cmr_chain_gemm<4, 16, LA_LD128>(W0, l31, h, hid, [&](int kg, int j) {
      return kg < 8 ? xf[kg & 7][j] : mrg[(kg - 8) / 4 & 1][4 * ((kg - 8) % 4 & 3) + j];
    });
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) hid[t][r] = hid[t][r] > 0.f ? hid[t][r] : 0.f;
    if (TRAIN) {
#pragma unroll
      for (int kg = 0; kg < 16; ++kg) {
        const int t = kg / 4, qd = kg % 4;
        if (drop) {
#pragma unroll
          for (int e = 0; e < 4; ++e) hid[t][4 * qd + e] *= la_drop_mul(key_hid, (uint64_t)row * LA_HID + 8 * kg + 4 * h + e, a.thr, a.ks);
        }
        *reinterpret_cast<f32x4*>(a.sv_hid + (int64_t)row * LA_HID + 8 * kg + 4 * h) = f32x4{hid[t][4 * qd], hid[t][4 * qd + 1], hid[t][4 * qd + 2], hid[t][4 * qd + 3]};
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    f32x16 o[2];
    cmr_chain_gemm<2, 16, LA_LD128>(W3, l31, h, o, [&](int kg, int j) { return hid[kg / 4][4 * (kg % 4) + j]; });
    if (TRAIN) {
#pragma unroll
      for (int kg = 0; kg < 8; ++kg) {
        const int t = kg / 4, qd = kg % 4;
        if (drop) {
#pragma unroll
          for (int e = 0; e < 4; ++e) o[t][4 * qd + e] *= la_drop_mul(key_out, (uint64_t)row * LA_D + 8 * kg + 4 * h + e, a.thr, a.ks);
        }
        *reinterpret_cast<f32x4*>(a.sv_o + (int64_t)row * LA_D + 8 * kg + 4 * h) = f32x4{o[t][4 * qd], o[t][4 * qd + 1], o[t][4 * qd + 2], o[t][4 * qd + 3]};
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    la_layernorm(o, Ln + 2 * LA_D, Ln + 3 * LA_D, h, a.ln_eps);

    f32x4 ov[8];
#pragma unroll
    for (int kg = 0; kg < 8; ++kg)
#pragma unroll
      for (int e = 0; e < 4; ++e) ov[kg][e] = xf[kg][e] + o[kg / 4][4 * (kg % 4) + e];
#pragma unroll
    for (int kg = 0; kg < 8; ++kg) cmr_pin(ov[kg]);
    if (valid) {
      float* yp = a.out + (int64_t)row * a.ldo + 4 * h;
#pragma unroll
      for (int kg = 0; kg < 8; ++kg) *reinterpret_cast<f32x4*>(yp + kg * 8) = ov[kg];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// State kernel.  Standard orientation (D[row][channel]: rows = A operand, weights = B operand): a lane owns one
// output CHANNEL and accumulator register r is row (r&3) + 8(r>>2) + 4h of the tile -- which is the operand layout of
// a second MFMA that contracts over the ROWS:  KV_tile[i][j] += sum_rows K[row][i] V[row][j]  (A = K registers,
// B = V registers, 16 MFMAs per 32-channel tile).  Only the 8x8 diagonal blocks (one per head) of that 32x32
// product are kept.  Workgroup (slab, b): wave w reduces tiles [(8 slab + w) TPW, + TPW) of batch b into registers
// and writes ONE partial state; la_state_final sums the partials in a fixed order (deterministic, no atomics).
struct LaStateArgs {
  const float* y; int64_t ldy;
  const float *wk, *wv;                // [64][64]
  float* part;                         // [B][nslab * 8][576]
  uint32_t S; int tiles_per_wave; float s;
  float *sv_kf, *sv_v;                 // SAVE instance: kf = elu(Wk y) + 1 and v = Wv y [B * S][64] for the backward (cmr_la_bwd_f32)
};

// BF16 (the bf16 mode of BASELINE configs[2] / [3]): the two projections run on v_mfma_f32_32x32x16_bf16 (16 instructions per 32-row tile
// instead of 128 fp32 ones; rows and weights rounded to bf16 on the way, fp32 accumulate), whose 32x32 result has the SAME register
// layout -- so elu+1, 1/S and the row contraction K^T V (fp32 MFMA, the state is a sum over up to 10^5 rows) are unchanged.
typedef __bf16 la_bf16x8 __attribute__((ext_vector_type(8)));
constexpr int LA_WPS = LA_D * 2 + 16;            // bytes per weight row in LDS (bf16): conflict-free ds_read_b128 over 32 consecutive rows

__device__ __forceinline__ la_bf16x8 la_pack8(const f32x4& lo, const f32x4& hi) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  typedef __bf16 b2 __attribute__((ext_vector_type(2)));
  uint4 w;
  w.x = __builtin_bit_cast(unsigned, __builtin_convertvector((f2){lo[0], lo[1]}, b2));
  w.y = __builtin_bit_cast(unsigned, __builtin_convertvector((f2){lo[2], lo[3]}, b2));
  w.z = __builtin_bit_cast(unsigned, __builtin_convertvector((f2){hi[0], hi[1]}, b2));
  w.w = __builtin_bit_cast(unsigned, __builtin_convertvector((f2){hi[2], hi[3]}, b2));
  return __builtin_bit_cast(la_bf16x8, w);
}

template <bool BF16, bool SAVE = false>
__global__ __launch_bounds__(512) void la_state_partial_kernel(const LaStateArgs a) {
  // fp32: [64][LA_LD64] floats per matrix; bf16: [64] rows of LA_WPS bytes
  __shared__ __attribute__((aligned(16))) float Wk[BF16 ? LA_D * LA_WPS / 4 : LA_D * LA_LD64];
  __shared__ __attribute__((aligned(16))) float Wv[BF16 ? LA_D * LA_WPS / 4 : LA_D * LA_LD64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  if constexpr (BF16) {
    for (int e = tid; e < LA_D * (LA_D / 8); e += 512) {
      const int n = e / (LA_D / 8), c = (e % (LA_D / 8)) * 8;
      *reinterpret_cast<la_bf16x8*>(reinterpret_cast<unsigned char*>(Wk) + n * LA_WPS + c * 2) =
          la_pack8(*reinterpret_cast<const f32x4*>(a.wk + n * LA_D + c), *reinterpret_cast<const f32x4*>(a.wk + n * LA_D + c + 4));
      *reinterpret_cast<la_bf16x8*>(reinterpret_cast<unsigned char*>(Wv) + n * LA_WPS + c * 2) =
          la_pack8(*reinterpret_cast<const f32x4*>(a.wv + n * LA_D + c), *reinterpret_cast<const f32x4*>(a.wv + n * LA_D + c + 4));
    }
  } else {
    for (int e = tid; e < LA_D * (LA_D / 4); e += 512) {
      const int n = e / (LA_D / 4), c = (e % (LA_D / 4)) * 4;
      *reinterpret_cast<f32x4*>(&Wk[n * LA_LD64 + c]) = *reinterpret_cast<const f32x4*>(a.wk + n * LA_D + c);
      *reinterpret_cast<f32x4*>(&Wv[n * LA_LD64 + c]) = *reinterpret_cast<const f32x4*>(a.wv + n * LA_D + c);
    }
  }
  __syncthreads();
  const int b = blockIdx.y;
  const uint32_t tiles_b = (a.S + 31) / 32;
  const uint32_t t0 = (blockIdx.x * 8 + wave) * a.tiles_per_wave;
  const float* yb = a.y + (int64_t)b * a.S * a.ldy;

  f32x16 kv[2];                                  // kv[t]: 32x32 product of channel tile t with itself
  float ksum[2] = {0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) kv[t][r] = 0.f;

  for (int it = 0; it < a.tiles_per_wave; ++it) {
    const uint32_t tile = t0 + it;
    if (tile >= tiles_b) break;
    const uint32_t row = tile * 32 + l31;
    const uint32_t rowc = row < a.S ? row : 0;
    f32x16 kk[2], vv[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) { kk[t][r] = 0.f; vv[t][r] = 0.f; }
    if constexpr (BF16) {
      // A = rows: lane (row l31, half h) holds y[row][16 ks + 8 h .. + 7]; B = weights: lane (channel l31, half h) holds W[32 t + l31][same k]
      const float* yp = yb + (int64_t)rowc * a.ldy + 8 * h;
      f32x4 yf[8];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        yf[2 * ks] = *reinterpret_cast<const f32x4*>(yp + ks * 16);
        yf[2 * ks + 1] = *reinterpret_cast<const f32x4*>(yp + ks * 16 + 4);
      }
      const unsigned char* wkr = reinterpret_cast<const unsigned char*>(Wk) + l31 * LA_WPS + h * 16;
      const unsigned char* wvr = reinterpret_cast<const unsigned char*>(Wv) + l31 * LA_WPS + h * 16;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const la_bf16x8 ya = la_pack8(yf[2 * ks], yf[2 * ks + 1]);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const la_bf16x8 wkb = *reinterpret_cast<const la_bf16x8*>(wkr + t * 32 * LA_WPS + ks * 32);
          const la_bf16x8 wvb = *reinterpret_cast<const la_bf16x8*>(wvr + t * 32 * LA_WPS + ks * 32);
          kk[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ya, wkb, kk[t], 0, 0, 0);      // D[row][channel 32t + l31]
          vv[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ya, wvb, vv[t], 0, 0, 0);
        }
      }
    } else {
      const float* yp = yb + (int64_t)rowc * a.ldy + 4 * h;
      f32x4 yf[8];
#pragma unroll
      for (int kg = 0; kg < 8; ++kg) yf[kg] = *reinterpret_cast<const f32x4*>(yp + kg * 8);
      const float* wkr = Wk + l31 * LA_LD64 + 4 * h;
      const float* wvr = Wv + l31 * LA_LD64 + 4 * h;
#pragma unroll
      for (int kg = 0; kg < 8; ++kg) {
        f32x4 wk4[2], wv4[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          wk4[t] = *reinterpret_cast<const f32x4*>(wkr + t * 32 * LA_LD64 + kg * 8);
          wv4[t] = *reinterpret_cast<const f32x4*>(wvr + t * 32 * LA_LD64 + kg * 8);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            kk[t] = cmr_mfma32(yf[kg][j], wk4[t][j], kk[t]);      // D[row][channel 32t + l31]
            vv[t] = cmr_mfma32(yf[kg][j], wv4[t][j], vv[t]);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // K = elu + 1 (rows past the end contribute nothing); V's 1 / S is applied to the summed state in la_state_final_kernel (one division
    // per state entry instead of 32 ten-instruction fp32 divisions per lane and tile)
    const bool partial = tile * 32 + 32 > a.S;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float k = la_elu1(kk[t][r]);
        const uint32_t srow = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (partial && srow >= a.S) k = 0.f;
        if (SAVE && srow < a.S) {                  // lane = channel 32 t + l31: 32 consecutive floats per row and lane half
          const int64_t o = ((int64_t)b * a.S + srow) * LA_D + 32 * t + l31;
          a.sv_kf[o] = k;
          a.sv_v[o] = vv[t][r];
        }
        kk[t][r] = k;
        ksum[t] += k;
      }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) kv[t] = cmr_mfma32(kk[t][r], vv[t][r], kv[t]);   // contracts rows {r-th of h=0, r-th of h=1}
  }
  // partial state of this wave: KV[hd][d][v] from the diagonal 8x8 blocks, Ksum[hd][d]
  float* p = a.part + ((int64_t)b * gridDim.x * 8 + blockIdx.x * 8 + wave) * LA_STATE;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const float ks = ksum[t] + la_xhalf(ksum[t]);
    if (h == 0) p[512 + 32 * t + l31] = ks;                      // channel 32t + l31 = hd*8 + d
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * h;              // K channel (within tile t) of this register
      if ((i >> 3) == (l31 >> 3)) p[(4 * t + (i >> 3)) * 64 + (i & 7) * 8 + (l31 & 7)] = kv[t][r];
    }
  }
}

// kvsum[b][c] = sum over the partial states, fixed order: 16 interleaved groups of partials per column, then
// the 16 group sums in sequence.  grid (9 column blocks of 64, B), 1024 threads = 16 groups x 64 columns.
__global__ __launch_bounds__(1024) void la_state_final_kernel(const float* __restrict__ part, float* __restrict__ kvsum,
                                                              int npart, float s_len) {
  __shared__ float red[16][64];
  const int b = blockIdx.y, col = blockIdx.x * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;
  const float* p = part + (int64_t)b * npart * LA_STATE + col;
  // npart <= 256 (32 slabs x 8 waves): the <= 16 partials of this thread are requested together and summed in index order -- the same
  // sum as the one-load-per-iteration loop, without 16 memory round trips behind each other
  float v[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int i = grp + 16 * j;
    v[j] = p[(int64_t)(i < npart ? i : 0) * LA_STATE];
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) s += grp + 16 * j < npart ? v[j] : 0.f;
  for (int i = grp + 256; i < npart; i += 16) s += p[(int64_t)i * LA_STATE];       // (not reached with the launch geometry above)
  red[grp][threadIdx.x & 63] = s;
  __syncthreads();
  if (grp == 0) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) t += red[g][threadIdx.x];
    kvsum[(int64_t)b * LA_STATE + col] = col < 512 ? t / s_len : t;        // KV = K^T (V / S); Ksum is not scaled
  }
}

// The partition of a batch element's tiles into partial sums depends on S ONLY: a sample's result must not change
// with the batch it is in (ranks shard the batch, tests/test_fullsize_gpu.py checks independence).
int la_state_tiles_per_wave(int S) {
  const int tiles_b = (S + 31) / 32;
  int tpw = 1;
  while (tpw < 64 && (tiles_b + 8 * tpw - 1) / (8 * tpw) > 32) tpw *= 2;     // at most 32 workgroups per batch element (26 752 pixels: 27 x 8 = 216 workgroups at B = 8, 4 tiles per wave; 64 measured 71 us against 66)
  return tpw;
}
int la_state_nslab(int S) {
  const int tiles_b = (S + 31) / 32, tpw = la_state_tiles_per_wave(S);
  return (tiles_b + 8 * tpw - 1) / (8 * tpw);
}

}  // namespace

extern "C" int64_t cmr_la_kv_state_workspace_bytes(int B, int S) {
  if (B <= 0 || S <= 0) return 0;
  return (int64_t)B * la_state_nslab(S) * 8 * LA_STATE * sizeof(float);
}

static int la_kv_state_launch(bool bf16, const float* y, int64_t ldy, const float* wk, const float* wv, float* kvsum, void* workspace,
                              int64_t workspace_bytes, int B, int S, hipStream_t stream, float* sv_kf = nullptr, float* sv_v = nullptr) {
  CMR_REQUIRE(y && wk && wv && kvsum && workspace && B > 0 && B <= 65535 && S > 0 && (sv_kf == nullptr) == (sv_v == nullptr));
  CMR_REQUIRE(ldy % 4 == 0 && cmr_aligned16(y) && cmr_aligned16(wk) && cmr_aligned16(wv));
  CMR_REQUIRE(workspace_bytes >= cmr_la_kv_state_workspace_bytes(B, S));
  const int nslab = la_state_nslab(S);
  const LaStateArgs a{y, ldy, wk, wv, (float*)workspace, (uint32_t)S, la_state_tiles_per_wave(S), (float)S, sv_kf, sv_v};
  if (sv_kf) {
    if (bf16) return CMR_EUNSUPPORTED;
    hipLaunchKernelGGL((la_state_partial_kernel<false, true>), dim3(nslab, B), dim3(512), 0, stream, a);
  } else if (bf16) hipLaunchKernelGGL((la_state_partial_kernel<true>), dim3(nslab, B), dim3(512), 0, stream, a);
  else hipLaunchKernelGGL((la_state_partial_kernel<false>), dim3(nslab, B), dim3(512), 0, stream, a);
  hipLaunchKernelGGL(la_state_final_kernel, dim3(LA_STATE / 64, B), dim3(1024), 0, stream, (const float*)workspace, kvsum,
                     nslab * 8, (float)S);
  return cmr_launch_status();
}

extern "C" int cmr_la_kv_state_f32(const float* y, int64_t ldy, const float* wk, const float* wv, float* kvsum,
                                   void* workspace, int64_t workspace_bytes, int B, int S, hipStream_t stream) {
  return la_kv_state_launch(false, y, ldy, wk, wv, kvsum, workspace, workspace_bytes, B, S, stream);
}

/* train mode: the same state + the projected source rows the backward needs */
extern "C" int cmr_la_kv_state_train_f32(const float* y, int64_t ldy, const float* wk, const float* wv, float* kvsum, float* kf, float* v,
                                         void* workspace, int64_t workspace_bytes, int B, int S, hipStream_t stream) {
  CMR_REQUIRE(kf && v);
  return la_kv_state_launch(false, y, ldy, wk, wv, kvsum, workspace, workspace_bytes, B, S, stream, kf, v);
}

extern "C" int cmr_la_kv_state_bf16_f32(const float* y, int64_t ldy, const float* wk, const float* wv, float* kvsum,
                                        void* workspace, int64_t workspace_bytes, int B, int S, hipStream_t stream) {
  return la_kv_state_launch(true, y, ldy, wk, wv, kvsum, workspace, workspace_bytes, B, S, stream);
}

extern "C" int cmr_la_query_layer_f32(const float* x, int64_t ldx, const float* kvsum, const float* wq, const float* wmerge,
                                      const float* ln1_g, const float* ln1_b, const float* w_mlp0, const float* w_mlp3,
                                      const float* ln2_g, const float* ln2_b, float* out, int64_t ldo, int B, int L, int S,
                                      float eps, float ln_eps, hipStream_t stream) {
  CMR_REQUIRE(x && kvsum && wq && wmerge && ln1_g && ln1_b && w_mlp0 && w_mlp3 && ln2_g && ln2_b && out);
  CMR_REQUIRE(B > 0 && L > 0 && S > 0 && (int64_t)B * L < (int64_t)0x7fffffc0);
  CMR_REQUIRE(ldx % 4 == 0 && ldo % 4 == 0 && cmr_aligned16(x) && cmr_aligned16(out) && cmr_aligned16(kvsum) &&
              cmr_aligned16(wq) && cmr_aligned16(wmerge) && cmr_aligned16(w_mlp0) && cmr_aligned16(w_mlp3));
  const size_t smem = (size_t)(2 * LA_D * LA_LD64 + LA_HID * LA_LD128 + LA_D * LA_LD128 + 4 * LA_D + (size_t)B * LA_STATE) *
                      sizeof(float);
  if (smem > 160 * 1024) return CMR_EUNSUPPORTED;          // the per-batch states no longer fit beside the weights
  static CmrSmemCache granted{};
  if (cmr_grant_smem(reinterpret_cast<const void*>(la_query_layer_kernel<false>), smem, granted) != CMR_OK) return CMR_ELAUNCH;
  const uint32_t rows = (uint32_t)((int64_t)B * L);
  const uint32_t ntiles = (rows + 31) / 32;
  uint32_t grid = (ntiles + 7) / 8;
  if (grid > 256) grid = 256;                              // one persistent workgroup per CU (LDS bound)
  LaQueryArgs a{};
  a.x = x; a.ldx = ldx; a.kv = kvsum; a.wq = wq; a.wm = wmerge; a.w0 = w_mlp0; a.w3 = w_mlp3; a.g1 = ln1_g; a.b1 = ln1_b; a.g2 = ln2_g; a.b2 = ln2_b;
  a.out = out; a.ldo = ldo; a.rows = rows; a.L = (uint32_t)L; a.B = B; a.s = (float)S; a.eps = eps; a.ln_eps = ln_eps;
  hipLaunchKernelGGL(la_query_layer_kernel<false>, dim3(grid), dim3(512), smem, stream, a);
  return cmr_launch_status();
}

/* train mode: LinearAttention.forward under model.train() (LinearAttention.py:38-73 with its three nn.Dropout(0.1) active), query side;
 * saves [B L][64] qf, msg, mm, d1, o and [B L][128] hid for cmr_la_mlp_bwd_f32 / cmr_la_bwd_f32 / cmr_wgrad_group_f32.  The six save
 * buffers must hold WHOLE 32-row tiles (ceil(B L / 32) * 32 rows): rows past the end are written (garbage) so that no store is predicated */
extern "C" int cmr_la_query_layer_train_f32(const float* x, int64_t ldx, const float* kvsum, const float* wq, const float* wmerge,
                                            const float* ln1_g, const float* ln1_b, const float* w_mlp0, const float* w_mlp3,
                                            const float* ln2_g, const float* ln2_b, float* out, int64_t ldo, float* qf, float* msg, float* mm,
                                            float* d1, float* hid, float* o, int B, int L, int S, float eps, float ln_eps, float p,
                                            const int64_t* seed, int64_t site_att, int64_t site_hid, int64_t site_out, hipStream_t stream) {
  CMR_REQUIRE(x && kvsum && wq && wmerge && ln1_g && ln1_b && w_mlp0 && w_mlp3 && ln2_g && ln2_b && out && qf && msg && mm && d1 && hid && o);
  CMR_REQUIRE(B > 0 && L > 0 && S > 0 && (int64_t)B * L < (int64_t)0x7fffffc0 && p >= 0.f && p < 1.f);
  CMR_REQUIRE(ldx % 4 == 0 && ldo % 4 == 0 && cmr_aligned16(x) && cmr_aligned16(out) && cmr_aligned16(kvsum) && cmr_aligned16(wq) &&
              cmr_aligned16(wmerge) && cmr_aligned16(w_mlp0) && cmr_aligned16(w_mlp3) && cmr_aligned16(qf) && cmr_aligned16(msg) &&
              cmr_aligned16(mm) && cmr_aligned16(d1) && cmr_aligned16(hid) && cmr_aligned16(o));
  const size_t smem = (size_t)(2 * LA_D * LA_LD64 + LA_HID * LA_LD128 + LA_D * LA_LD128 + 4 * LA_D + (size_t)B * LA_STATE) * sizeof(float);
  if (smem > 160 * 1024) return CMR_EUNSUPPORTED;
  static CmrSmemCache granted{};
  if (cmr_grant_smem(reinterpret_cast<const void*>(la_query_layer_kernel<true>), smem, granted) != CMR_OK) return CMR_ELAUNCH;
  const uint32_t rows = (uint32_t)((int64_t)B * L);
  const uint32_t ntiles = (rows + 31) / 32;
  uint32_t grid = (ntiles + 7) / 8;
  if (grid > 256) grid = 256;
  LaQueryArgs a{};
  a.x = x; a.ldx = ldx; a.kv = kvsum; a.wq = wq; a.wm = wmerge; a.w0 = w_mlp0; a.w3 = w_mlp3; a.g1 = ln1_g; a.b1 = ln1_b; a.g2 = ln2_g; a.b2 = ln2_b;
  a.out = out; a.ldo = ldo; a.rows = rows; a.L = (uint32_t)L; a.B = B; a.s = (float)S; a.eps = eps; a.ln_eps = ln_eps;
  a.sv_qf = qf; a.sv_msg = msg; a.sv_mm = mm; a.sv_d1 = d1; a.sv_hid = hid; a.sv_o = o;
  a.seed = (seed && p > 0.f) ? seed : nullptr;
  a.site_att = (uint64_t)site_att; a.site_hid = (uint64_t)site_hid; a.site_out = (uint64_t)site_out;
  a.thr = cmr_drop_threshold(p); a.ks = 1.f / (1.f - p);
  hipLaunchKernelGGL(la_query_layer_kernel<true>, dim3(grid), dim3(512), smem, stream, a);
  return cmr_launch_status();
}
