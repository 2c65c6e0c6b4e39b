// Train-mode linear-attention layer, backward (reference models/LinearAttention.py:38-73 under model.train(); Train_Geo.py:166-174).
// Forward = cmr_la_kv_state_train_f32 + cmr_la_query_layer_train_f32 (la_fused.hip), which save qf, msg, mm, d1, hid, o and kf, v.
//
//   cmr_la_mlp_bwd_f32     query rows, from d out:  LayerNorm-2 backward -> dropout -> W3^T -> ReLU / dropout mask -> W0^T -> split into
//                          the gradient of x (first 64 inputs of the MLP) and of d1 -> dropout -> LayerNorm-1 backward -> Wm^T = gradient of
//                          the attention message.  One pass over the rows, GEMMs chained through the accumulators (transposed orientation,
//                          lane = row), weights resident in LDS and read COLUMN-wise for the transposed products.  Leaves the operands of the
//                          weight gradients (do, dhid, dmm) and per-tile sums of both LayerNorms' parameter gradients.
//   cmr_la_bwd_f32         (train_geo.hip) the attention core: d msg -> d qf, d kf, d v
//   cmr_la_proj_bwd_f32    projections: e = d * elu1'(f) per (gradient, saved activation) pair, dx = sum_i e_i W_i (+ residual gradients), for
//                          the query rows (Wq) and the source rows (Wk, Wv) -- or all three on one row set for self-attention -- in one launch
//   cmr_wgrad_group_f32    (wgrad_group.hip) every weight and LayerNorm-parameter gradient of the layer
#include "cmr_chain.h"
#include "cmr_mfma16.h"

namespace {

constexpr int LT_D = 64, LT_HID = 128;
constexpr int LT_LD64 = LT_D + 4, LT_LD128 = LT_HID + 4;

// Transposed product through a weight matrix that sits in LDS as [n_out][LD] (PyTorch [out][in] rows, as the forward reads them):
// acc[t][..] = d in[32 t + l31] = sum_o W[o][32 t + l31] * g[o],  o = 8 kg + 4 h + j  (KG k-groups of the OUTPUT dim, T tiles of the INPUT dim).
// A operand = one ds_read_b32 per (k-group, j, tile): lanes l31 read consecutive floats of row o (conflict free).
template <int T, int KG, int LD, typename BF>
__device__ __forceinline__ void lt_gemm_t(const float* __restrict__ Ws, int l31, int h, f32x16 (&acc)[T], BF bfrag) {
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const float* wcol = Ws + 4 * h * LD + l31;
#pragma unroll
  for (int kg = 0; kg < KG; ++kg) {
    float wv[4][T];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int t = 0; t < T; ++t) wv[j][t] = wcol[(8 * kg + j) * LD + 32 * t];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float b = bfrag(kg, j);
#pragma unroll
      for (int t = 0; t < T; ++t) acc[t] = cmr_mfma32(wv[j][t], b, acc[t]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

__device__ __forceinline__ float lt_sum32(float v) {   // over the 32 rows of a tile (lanes of one half)
  v = m16_sum16(v);
  return v + cmr_xor16(v);
}

// LayerNorm(64) backward on a row held as 8 float4 (channels 8 kg + 4 h + e; the other 32 channels in lane ^ 32).
// in: dy (gradient at the LayerNorm output), x (its input); out: dx; pg / pb: this row's contribution to d gamma / d beta.
__device__ __forceinline__ void lt_ln_bwd(const f32x4 (&dy)[8], const f32x4 (&x)[8], const float* __restrict__ gam, int h, float eps,
                                          f32x4 (&dx)[8], f32x4 (&pg)[8], f32x4 (&pb)[8]) {
  float s = 0.f;
#pragma unroll
  for (int kg = 0; kg < 8; ++kg) s += (x[kg][0] + x[kg][1]) + (x[kg][2] + x[kg][3]);
  s += cmr_xhalf(s);
  const float mean = s * (1.f / 64.f);
  float q = 0.f;
  f32x4 xh[8];
#pragma unroll
  for (int kg = 0; kg < 8; ++kg)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      xh[kg][e] = x[kg][e] - mean;
      q += xh[kg][e] * xh[kg][e];
    }
  q += cmr_xhalf(q);
  const float rstd = 1.f / sqrtf(q * (1.f / 64.f) + eps);
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int kg = 0; kg < 8; ++kg) {
    const f32x4 gv = *reinterpret_cast<const f32x4*>(gam + 8 * kg + 4 * h);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      xh[kg][e] *= rstd;
      dx[kg][e] = dy[kg][e] * gv[e];
      s1 += dx[kg][e];
      s2 += dx[kg][e] * xh[kg][e];
      pg[kg][e] = dy[kg][e] * xh[kg][e];
      pb[kg][e] = dy[kg][e];
    }
  }
  s1 += cmr_xhalf(s1);
  s2 += cmr_xhalf(s2);
  s1 *= (1.f / 64.f);
  s2 *= (1.f / 64.f);
#pragma unroll
  for (int kg = 0; kg < 8; ++kg)
#pragma unroll
    for (int e = 0; e < 4; ++e) dx[kg][e] = rstd * (dx[kg][e] - s1 - xh[kg][e] * s2);
}

struct LaMlpBwdArgs {
  const float* dout; int64_t lddo;
  const float *o, *hid, *mm;               // saved by the forward: [rows][64], [rows][128], [rows][64]
  const float *wm, *w0, *w3;               // [64][64], [128][128], [64][128]
  const float *g1, *g2;                    // LayerNorm gammas
  float *d_o, *d_hid, *d_mm, *d_msg, *d_xa;   // [rows][64], [rows][128], [rows][64], [rows][64], [rows][64]; each with room for WHOLE 32-row tiles
  float* lnpart1; float* lnpart2;          // [workgroups][128] each: sums of d gamma | d beta over the workgroup's tiles
  uint32_t rows;
  float ln_eps;
  const int64_t* seed; uint64_t site_att, site_hid, site_out;
  uint32_t thr; float ks;
};

template <bool DROP>
__global__ __launch_bounds__(512) void la_mlp_bwd_kernel(const LaMlpBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Wm = smem;                              // [64][68]
  float* W0 = Wm + LT_D * LT_LD64;               // [128][132]
  float* W3 = W0 + LT_HID * LT_LD128;            // [64][132]
  float* Ln = W3 + LT_D * LT_LD128;              // g1 | g2
  float* Acc = Ln + 2 * LT_D;                    // [8 waves][2 LayerNorms][128]: this wave's running sums of d gamma | d beta over its tiles
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  {
    f32x4 wm4[2], w04[8], w34[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) wm4[i] = *reinterpret_cast<const f32x4*>(a.wm + (tid + 512 * i) * 4);
#pragma unroll
    for (int i = 0; i < 8; ++i) w04[i] = *reinterpret_cast<const f32x4*>(a.w0 + (tid + 512 * i) * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) w34[i] = *reinterpret_cast<const f32x4*>(a.w3 + (tid + 512 * i) * 4);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = tid + 512 * i, n = e / (LT_D / 4), c = (e % (LT_D / 4)) * 4;
      *reinterpret_cast<f32x4*>(&Wm[n * LT_LD64 + c]) = wm4[i];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = tid + 512 * i, n = e / (LT_HID / 4), c = (e % (LT_HID / 4)) * 4;
      *reinterpret_cast<f32x4*>(&W0[n * LT_LD128 + c]) = w04[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + 512 * i, n = e / (LT_HID / 4), c = (e % (LT_HID / 4)) * 4;
      *reinterpret_cast<f32x4*>(&W3[n * LT_LD128 + c]) = w34[i];
    }
  }
  if (tid < LT_D) {
    Ln[tid] = a.g1[tid];
    Ln[LT_D + tid] = a.g2[tid];
  }
  for (int e = tid; e < 8 * 2 * 128; e += 512) Acc[e] = 0.f;
  __syncthreads();
  uint64_t key_att = 0, key_hid = 0, key_out = 0;
  if (DROP) {
    const uint64_t sd = (uint64_t)a.seed[0];
    key_att = cmr_mix64(sd + a.site_att * 0x9E3779B97F4A7C15ull);
    key_hid = cmr_mix64(sd + a.site_hid * 0x9E3779B97F4A7C15ull);
    key_out = cmr_mix64(sd + a.site_out * 0x9E3779B97F4A7C15ull);
  }
  auto keep = [&](uint64_t key, uint64_t idx) { return (uint32_t)cmr_mix64(key ^ idx) >= a.thr ? a.ks : 0.f; };

  const uint32_t ntiles = (a.rows + 31) / 32;
  for (uint32_t tile = blockIdx.x * 8 + wave; tile < ntiles; tile += gridDim.x * 8) {
    const uint32_t row = tile * 32 + l31;
    const bool valid = row < a.rows;
    const uint32_t rowc = valid ? row : 0;
    const float vmul = valid ? 1.f : 0.f;
    // ---- LayerNorm 2 backward: out = x + LN2(o)  =>  d LN2 = d out
    f32x4 dy[8], xo[8], dO[8], pg[8], pb[8];
    {
      const float* dp = a.dout + (int64_t)rowc * a.lddo + 4 * h;
      const float* op = a.o + (int64_t)rowc * LT_D + 4 * h;
#pragma unroll
      for (int kg = 0; kg < 8; ++kg) {
        dy[kg] = *reinterpret_cast<const f32x4*>(dp + kg * 8);
        xo[kg] = *reinterpret_cast<const f32x4*>(op + kg * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) dy[kg][e] *= vmul;              // rows past the end contribute nothing
      }
    }
    lt_ln_bwd(dy, xo, Ln + LT_D, h, a.ln_eps, dO, pg, pb);
#pragma unroll
    for (int kg = 0; kg < 8; ++kg)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        pg[kg][e] = lt_sum32(pg[kg][e]);
        pb[kg][e] = lt_sum32(pb[kg][e]);
      }
    if (l31 == 0) {                              // lanes 0 and 32: own slot, own channels -- plain read-modify-write, fixed order
      float* lp = Acc + (wave * 2 + 1) * 128 + 4 * h;
#pragma unroll
      for (int kg = 0; kg < 8; ++kg) {
        *reinterpret_cast<f32x4*>(lp + 8 * kg) += pg[kg];
        *reinterpret_cast<f32x4*>(lp + 64 + 8 * kg) += pb[kg];
      }
    }
    // ---- through the output dropout: gradient at W3's output (operand of dW3)
    if (DROP) {
#pragma unroll
      for (int kg = 0; kg < 8; ++kg)
#pragma unroll
        for (int e = 0; e < 4; ++e) dO[kg][e] *= keep(key_out, (uint64_t)row * LT_D + 8 * kg + 4 * h + e);
    }
    {                                            // (output buffers hold whole tiles: no predicated stores)
      float* p = a.d_o + (int64_t)row * LT_D + 4 * h;
#pragma unroll
      for (int kg = 0; kg < 8; ++kg) *reinterpret_cast<f32x4*>(p + 8 * kg) = dO[kg];
    }
    // ---- d hidden = dO W3, through ReLU and the hidden dropout (hid > 0 iff kept and active)
    f32x16 dh[4];
    lt_gemm_t<4, 8, LT_LD128>(W3, l31, h, dh, [&](int kg, int j) { return dO[kg][j]; });
    {
      const float* hp = a.hid + (int64_t)rowc * LT_HID + 4 * h;
#pragma unroll
      for (int kg = 0; kg < 16; ++kg) {
        const f32x4 hv = *reinterpret_cast<const f32x4*>(hp + kg * 8);
        const int t = kg / 4, qd = kg % 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) dh[t][4 * qd + e] = hv[e] > 0.f ? dh[t][4 * qd + e] * (DROP ? a.ks : 1.f) : 0.f;
      }
    }
    {                                            // (output buffers hold whole tiles: no predicated stores)
      float* p = a.d_hid + (int64_t)row * LT_HID + 4 * h;
#pragma unroll
      for (int kg = 0; kg < 16; ++kg) {
        const int t = kg / 4, qd = kg % 4;
        *reinterpret_cast<f32x4*>(p + 8 * kg) = f32x4{dh[t][4 * qd], dh[t][4 * qd + 1], dh[t][4 * qd + 2], dh[t][4 * qd + 3]};
      }
    }
    // ---- d [x | d1] = dhid W0: tiles 0, 1 = the gradient of x through the MLP, tiles 2, 3 = the gradient of d1
    f32x16 dc[4];
    lt_gemm_t<4, 16, LT_LD128>(W0, l31, h, dc, [&](int kg, int j) { return dh[kg / 4][4 * (kg % 4) + j]; });
    {                                            // (output buffers hold whole tiles: no predicated stores)
      float* p = a.d_xa + (int64_t)row * LT_D + 4 * h;
#pragma unroll
      for (int kg = 0; kg < 8; ++kg) {
        const int t = kg / 4, qd = kg % 4;
        *reinterpret_cast<f32x4*>(p + 8 * kg) = f32x4{dc[t][4 * qd], dc[t][4 * qd + 1], dc[t][4 * qd + 2], dc[t][4 * qd + 3]};
      }
    }
    // ---- through the attention dropout and LayerNorm 1 (input mm)
    f32x4 dn[8], xm[8], dM[8];
    {
      const float* mp = a.mm + (int64_t)rowc * LT_D + 4 * h;
#pragma unroll
      for (int kg = 0; kg < 8; ++kg) {
        const int t = 2 + kg / 4, qd = kg % 4;
        xm[kg] = *reinterpret_cast<const f32x4*>(mp + kg * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = dc[t][4 * qd + e];
          if (DROP) v *= keep(key_att, (uint64_t)row * LT_D + 8 * kg + 4 * h + e);
          dn[kg][e] = v;
        }
      }
    }
    lt_ln_bwd(dn, xm, Ln, h, a.ln_eps, dM, pg, pb);
#pragma unroll
    for (int kg = 0; kg < 8; ++kg)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        pg[kg][e] = lt_sum32(pg[kg][e]);
        pb[kg][e] = lt_sum32(pb[kg][e]);
      }
    if (l31 == 0) {
      float* lp = Acc + (wave * 2 + 0) * 128 + 4 * h;
#pragma unroll
      for (int kg = 0; kg < 8; ++kg) {
        *reinterpret_cast<f32x4*>(lp + 8 * kg) += pg[kg];
        *reinterpret_cast<f32x4*>(lp + 64 + 8 * kg) += pb[kg];
      }
    }
    {                                            // (output buffers hold whole tiles: no predicated stores)
      float* p = a.d_mm + (int64_t)row * LT_D + 4 * h;
#pragma unroll
      for (int kg = 0; kg < 8; ++kg) *reinterpret_cast<f32x4*>(p + 8 * kg) = dM[kg];
    }
    // ---- d message = dmm Wm
    f32x16 dg[2];
    lt_gemm_t<2, 8, LT_LD64>(Wm, l31, h, dg, [&](int kg, int j) { return dM[kg][j]; });
    {                                            // (output buffers hold whole tiles: no predicated stores)
      float* p = a.d_msg + (int64_t)row * LT_D + 4 * h;
#pragma unroll
      for (int kg = 0; kg < 8; ++kg) {
        const int t = kg / 4, qd = kg % 4;
        *reinterpret_cast<f32x4*>(p + 8 * kg) = f32x4{dg[t][4 * qd], dg[t][4 * qd + 1], dg[t][4 * qd + 2], dg[t][4 * qd + 3]};
      }
    }
  }
  // one partial row per WORKGROUP and LayerNorm (the per-tile rows of a 214 016-row map were 6 688 partials per output for the reduction)
  __syncthreads();
  if (tid < 256) {
    const int ln = tid >> 7, c = tid & 127;
    float sacc = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) sacc += Acc[(w * 2 + ln) * 128 + c];
    (ln == 0 ? a.lnpart1 : a.lnpart2)[(int64_t)blockIdx.x * 128 + c] = sacc;
  }
}

// ---- projections backward: per row set, dx = sum_i (d_i * elu1'(f_i)) W_i + residual gradients ---------------------------------------
// 32-row tiles, lane = 32 h + l owns row l; W_i^T as frag32 fragments [2][8][64][4] from L2 (cmr_pack_frags_f32 kind 0, transposed).
// f_i = the SAVED activation elu(z) + 1 (derivative: 1 where f > 1, else f); null = no activation (the value projection).
struct LaProjTerm { const float* d; int64_t ldd; const float* f; const float* wt_f; float* e_out; };
struct LaProjProblem {
  LaProjTerm t[3]; int nterm;
  const float* res0; int64_t ldr0; const float* res1; int64_t ldr1;
  float* dx; int64_t lddx; uint32_t rows;
};
struct LaProjArgs { LaProjProblem p[2]; uint32_t tiles0, tiles; };

__global__ __launch_bounds__(256) void la_proj_bwd_kernel(const LaProjArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  const uint32_t gt = blockIdx.x * (blockDim.x >> 6) + wave;
  if (gt >= a.tiles) return;
  const LaProjProblem& P = gt < a.tiles0 ? a.p[0] : a.p[1];
  const uint32_t tile = gt < a.tiles0 ? gt : gt - a.tiles0;
  const uint32_t row = tile * 32 + l31;
  const bool valid = row < P.rows;
  const uint32_t rowc = valid ? row : 0;
  f32x16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  for (int i = 0; i < P.nterm; ++i) {
    const LaProjTerm& T = P.t[i];
    const float* dp = T.d + (int64_t)rowc * T.ldd + 4 * h;
    f32x4 df[8];
#pragma unroll
    for (int kg = 0; kg < 8; ++kg) df[kg] = *reinterpret_cast<const f32x4*>(dp + kg * 8);
    if (T.f) {
      const float* fp = T.f + (int64_t)rowc * LT_D + 4 * h;
#pragma unroll
      for (int kg = 0; kg < 8; ++kg) {
        const f32x4 fv = *reinterpret_cast<const f32x4*>(fp + kg * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) df[kg][e] *= fv[e] > 1.f ? 1.f : fv[e];
      }
      if (valid && T.e_out) {
        float* ep = T.e_out + (int64_t)row * LT_D + 4 * h;
#pragma unroll
        for (int kg = 0; kg < 8; ++kg) *reinterpret_cast<f32x4*>(ep + kg * 8) = df[kg];
      }
    }
    const float* wp = T.wt_f + lane * 4;
    f32x4 wr[8][2];
#pragma unroll
    for (int kg = 0; kg < 8; ++kg)
#pragma unroll
      for (int t = 0; t < 2; ++t) wr[kg][t] = *reinterpret_cast<const f32x4*>(wp + ((int64_t)t * 8 + kg) * 256);
#pragma unroll
    for (int kg = 0; kg < 8; ++kg)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t] = cmr_mfma32(wr[kg][t][j], df[kg][j], acc[t]);
  }
  f32x4 ov[8];
#pragma unroll
  for (int kg = 0; kg < 8; ++kg) {
    const int t = kg / 4, qd = kg % 4;
    ov[kg] = f32x4{acc[t][4 * qd], acc[t][4 * qd + 1], acc[t][4 * qd + 2], acc[t][4 * qd + 3]};
  }
  if (P.res0) {
    const float* rp = P.res0 + (int64_t)rowc * P.ldr0 + 4 * h;
#pragma unroll
    for (int kg = 0; kg < 8; ++kg) ov[kg] += *reinterpret_cast<const f32x4*>(rp + kg * 8);
  }
  if (P.res1) {
    const float* rp = P.res1 + (int64_t)rowc * P.ldr1 + 4 * h;
#pragma unroll
    for (int kg = 0; kg < 8; ++kg) ov[kg] += *reinterpret_cast<const f32x4*>(rp + kg * 8);
  }
#pragma unroll
  for (int kg = 0; kg < 8; ++kg) cmr_pin(ov[kg]);
  if (valid) {
    float* op = P.dx + (int64_t)row * P.lddx + 4 * h;
#pragma unroll
    for (int kg = 0; kg < 8; ++kg) *reinterpret_cast<f32x4*>(op + kg * 8) = ov[kg];
  }
}

}  // namespace

extern "C" int cmr_la_mlp_bwd_f32(const float* dout, int64_t lddo, const float* o, const float* hid, const float* mm, const float* wmerge,
                                  const float* w_mlp0, const float* w_mlp3, const float* ln1_g, const float* ln2_g, float* d_o, float* d_hid,
                                  float* d_mm, float* d_msg, float* d_xa, float* lnpart1, float* lnpart2, int64_t rows, float ln_eps, float p,
                                  const int64_t* seed, int64_t site_att, int64_t site_hid, int64_t site_out, hipStream_t stream) {
  CMR_REQUIRE(dout && o && hid && mm && wmerge && w_mlp0 && w_mlp3 && ln1_g && ln2_g && d_o && d_hid && d_mm && d_msg && d_xa && lnpart1 && lnpart2);
  CMR_REQUIRE(rows > 0 && rows < (int64_t)0x7fffffc0 && lddo % 4 == 0 && p >= 0.f && p < 1.f);
  CMR_REQUIRE(cmr_aligned16(dout) && cmr_aligned16(o) && cmr_aligned16(hid) && cmr_aligned16(mm) && cmr_aligned16(wmerge) && cmr_aligned16(w_mlp0) &&
              cmr_aligned16(w_mlp3) && cmr_aligned16(d_o) && cmr_aligned16(d_hid) && cmr_aligned16(d_mm) && cmr_aligned16(d_msg) && cmr_aligned16(d_xa) &&
              cmr_aligned16(lnpart1) && cmr_aligned16(lnpart2));
  const size_t smem = (size_t)(LT_D * LT_LD64 + LT_HID * LT_LD128 + LT_D * LT_LD128 + 2 * LT_D + 8 * 2 * 128) * sizeof(float);
  const bool drop = seed != nullptr && p > 0.f;
  static CmrSmemCache g0{}, g1{};
  if (cmr_grant_smem(reinterpret_cast<const void*>(la_mlp_bwd_kernel<false>), smem, g0) != CMR_OK) return CMR_ELAUNCH;
  if (cmr_grant_smem(reinterpret_cast<const void*>(la_mlp_bwd_kernel<true>), smem, g1) != CMR_OK) return CMR_ELAUNCH;
  const uint32_t ntiles = (uint32_t)((rows + 31) / 32);
  uint32_t grid = (ntiles + 7) / 8;
  if (grid > 256) grid = 256;
  const LaMlpBwdArgs a{dout, lddo, o, hid, mm, wmerge, w_mlp0, w_mlp3, ln1_g, ln2_g, d_o, d_hid, d_mm, d_msg, d_xa, lnpart1, lnpart2, (uint32_t)rows,
                       ln_eps, drop ? seed : nullptr, (uint64_t)site_att, (uint64_t)site_hid, (uint64_t)site_out, cmr_drop_threshold(p), 1.f / (1.f - p)};
  if (drop) hipLaunchKernelGGL(la_mlp_bwd_kernel<true>, dim3(grid), dim3(512), smem, stream, a);
  else hipLaunchKernelGGL(la_mlp_bwd_kernel<false>, dim3(grid), dim3(512), smem, stream, a);
  return cmr_launch_status();
}

// desc: HOST array, 23 int64 per row set: rows, nterm, dx, lddx, res0, ldr0, res1, ldr1, then 3 x {d, ldd, f (0: none), wt_f, e_out (0: none)}
extern "C" int cmr_la_proj_bwd_f32(const int64_t* desc, int nprob, hipStream_t stream) {
  CMR_REQUIRE(desc && (nprob == 1 || nprob == 2));
  LaProjArgs a{};
  for (int i = 0; i < nprob; ++i) {
    const int64_t* d = desc + 23 * i;
    LaProjProblem& P = a.p[i];
    P.rows = (uint32_t)d[0];
    P.nterm = (int)d[1];
    P.dx = reinterpret_cast<float*>(d[2]); P.lddx = d[3];
    P.res0 = reinterpret_cast<const float*>(d[4]); P.ldr0 = d[5];
    P.res1 = reinterpret_cast<const float*>(d[6]); P.ldr1 = d[7];
    CMR_REQUIRE(d[0] > 0 && d[0] < (int64_t)0x7fffffc0 && P.nterm >= 1 && P.nterm <= 3 && P.dx && P.lddx % 4 == 0 && cmr_aligned16(P.dx));
    CMR_REQUIRE((!P.res0 || (cmr_aligned16(P.res0) && P.ldr0 % 4 == 0)) && (!P.res1 || (cmr_aligned16(P.res1) && P.ldr1 % 4 == 0)));
    for (int j = 0; j < P.nterm; ++j) {
      const int64_t* t = d + 8 + 5 * j;
      P.t[j] = LaProjTerm{reinterpret_cast<const float*>(t[0]), t[1], reinterpret_cast<const float*>(t[2]), reinterpret_cast<const float*>(t[3]),
                          reinterpret_cast<float*>(t[4])};
      CMR_REQUIRE(P.t[j].d && P.t[j].wt_f && t[1] % 4 == 0 && cmr_aligned16(P.t[j].d) && cmr_aligned16(P.t[j].wt_f) &&
                  (!P.t[j].f || cmr_aligned16(P.t[j].f)) && (!P.t[j].e_out || cmr_aligned16(P.t[j].e_out)));
    }
  }
  a.tiles0 = (a.p[0].rows + 31) / 32;
  a.tiles = a.tiles0 + (nprob == 2 ? (a.p[1].rows + 31) / 32 : 0);
  if (nprob == 1) a.p[1] = a.p[0];
  if (a.tiles <= 2048) hipLaunchKernelGGL(la_proj_bwd_kernel, dim3(a.tiles), dim3(64), 0, stream, a);
  else hipLaunchKernelGGL(la_proj_bwd_kernel, dim3((a.tiles + 3) / 4), dim3(256), 0, stream, a);
  return cmr_launch_status();
}
