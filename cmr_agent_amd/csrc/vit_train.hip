// Train-mode pre-LN transformer block (reference models/ImageViT.py:61-158, PointViT.py:96-183, IMGPCEncoder.py:14-102 under
// model.train(), Train_Geo.py:166-174): forward in three launches, backward in four (+ the attention backward), instead of one launch per
// reference op in each direction (16 + ~20 C-ABI calls per block, 30 blocks per step).
//
//   forward    cmr_ln64_linear_f32 (vit_fused.hip)   LayerNorm + q / k / v projections                         -> q, k, v
//              cmr_mha_dropout_f32 (attention.hip)   softmax attention, dropout on the probabilities           -> ctx
//              cmr_vit_out_ffn16_train_f32           x1 = x + drop(ctx Wo^T + bo);  out = x1 + drop(W2 drop(gelu(W1 LN(x1) + b1)) + b2)
//                                                    -> out, x1 (the only activation the backward needs besides q, k, v, ctx)
//   backward   cmr_vit_ffn_bwd16_f32                 from d out: recomputes LN(x1), the fc1 pre-activations and the masks; MLP data
//                                                    gradients chained through the accumulators; LayerNorm backward; d x1; then the
//                                                    out-projection's data gradient d ctx = drop(d x1) Wo.  Leaves the operands of the
//                                                    weight gradients (masked d out, hidden activations, d fc1, LN(x1), drop(d x1)) and the
//                                                    per-tile sums of the LayerNorm parameter gradients
//              cmr_mha_dropout_bwd_f32 (train_geo.hip)
//              cmr_vit_lnqkv_bwd_f32                 d LN(x) = [dq | dk | dv] [Wq; Wk; Wv], LayerNorm backward (+ the residual stream's
//                                                    gradient), for the query rows and (cross block) the source rows in one launch
//              cmr_wgrad_group_f32 (wgrad_group.hip) every weight / bias / LayerNorm gradient of the block
//   per step   cmr_pack_frags_f32                    the blocks' weights from the flat parameter bucket into MFMA fragment order
//
// Dropout masks are the counter-based ones of cmr_dropout_f32 (element index = row * width + column of the tensor the reference's
// nn.Dropout sees), so this path and the op-by-op path draw the SAME masks for the same (seed, site).
#include "cmr_mfma16.h"

namespace {

struct OutFfnTrainArgs {
  const float* ctx; int64_t ldc;
  const float* x; int64_t ldx;
  const float* wo_f; const float* bo;                  // frag16 [4][4][64][4], [64]
  const float* g2; const float* b2n; float eps;        // ffn_norm
  const float* w1_f; const float* b1;                  // frag16 [64][4][64][4], [1024]
  const float* w2_f; const float* b2;                  // frag16 [4][64][64][4], [64]
  float* out; int64_t ldo;
  float* x1; int64_t ldx1;
  uint32_t rows;
  const int64_t* seed; uint64_t site_proj, site_act, site_fc2;
  uint32_t thr_proj, thr_mlp; float ks_proj, ks_mlp;
};

template <bool DROP>
__global__ __launch_bounds__(512) void vit_out_ffn16_train_kernel(const OutFfnTrainArgs a) {
  __shared__ __attribute__((aligned(16))) float red[7 * 4 * 64 * 4];     // partial outputs of waves 1..7: [w][tile][lane][4]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, n = lane & 15;
  const uint32_t row = blockIdx.x * 16 + n;
  const bool valid = row < a.rows;
  const uint32_t rowc = valid ? row : 0;
  const float* cp = a.ctx + (int64_t)rowc * a.ldc + 4 * g;
  const float* xp = a.x + (int64_t)rowc * a.ldx + 4 * g;
  f32x4 cf[4], x1[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    cf[t] = *reinterpret_cast<const f32x4*>(cp + 16 * t);
    x1[t] = *reinterpret_cast<const f32x4*>(xp + 16 * t);
  }
  M16Drop dp, da, df;
  if (DROP) {
    dp.init(a.seed, a.site_proj, a.thr_proj, a.ks_proj);
    da.init(a.seed, a.site_act, a.thr_mlp, a.ks_mlp);
    df.init(a.seed, a.site_fc2, a.thr_mlp, a.ks_mlp);
  }
  // ---- x1 = x + drop(ctx Wo^T + bo)     (every wave: 64 MFMAs, cheaper than a broadcast through LDS)
  {
    f32x4 acc[4];
    m16_gemm<4, 4, 4>(a.wo_f, 4, 0, 0, lane, acc, [&](int t, int r) { return cf[t][r]; });
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bo + 16 * t + 4 * g);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = acc[t][e] + bv[e];
        if (DROP) v *= dp.mul((uint64_t)row * 64 + 16 * t + 4 * g + e);
        x1[t][e] = v + x1[t][e];
      }
    }
  }
  // ---- LayerNorm(64) of the row: 16 channels here, the others in the three partner lanes
  f32x4 xn[4];
  {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) s += (x1[t][0] + x1[t][1]) + (x1[t][2] + x1[t][3]);
    const float mean = m16_allg(s) * (1.f / 64.f);
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = x1[t][e] - mean;
        xn[t][e] = d;
        q += d * d;
      }
    const float rstd = 1.f / sqrtf(m16_allg(q) * (1.f / 64.f) + a.eps);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const f32x4 gv = *reinterpret_cast<const f32x4*>(a.g2 + 16 * t + 4 * g);
      const f32x4 bv = *reinterpret_cast<const f32x4*>(a.b2n + 16 * t + 4 * g);
#pragma unroll
      for (int e = 0; e < 4; ++e) xn[t][e] = xn[t][e] * rstd * gv[e] + bv[e];
    }
  }
  // ---- this wave's 128 hidden units: fc1 + GELU + dropout (8 tiles of 16), then its K-slice of fc2
  f32x4 hid[8];
  m16_gemm<8, 4, 4>(a.w1_f, 4, 8 * wave, 0, lane, hid, [&](int t, int r) { return xn[t][r]; });
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const f32x4 bv = *reinterpret_cast<const f32x4*>(a.b1 + 128 * wave + 16 * t + 4 * g);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = m16_gelu(hid[t][e] + bv[e]);
      if (DROP) v *= da.mul((uint64_t)row * 1024 + 128 * wave + 16 * t + 4 * g + e);
      hid[t][e] = v;
    }
  }
  f32x4 part[4];
  m16_gemm<4, 8, 8>(a.w2_f, 64, 0, 8 * wave, lane, part, [&](int t, int r) { return hid[t][r]; });
  if (wave > 0) {
#pragma unroll
    for (int t = 0; t < 4; ++t) *reinterpret_cast<f32x4*>(&red[(((wave - 1) * 4 + t) * 64 + lane) * 4]) = part[t];
  }
  __syncthreads();
  if (wave != 0) return;
  f32x4 ov[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    f32x4 s = part[t];
#pragma unroll
    for (int w = 0; w < 7; ++w) s += *reinterpret_cast<const f32x4*>(&red[((w * 4 + t) * 64 + lane) * 4]);
    const f32x4 bv = *reinterpret_cast<const f32x4*>(a.b2 + 16 * t + 4 * g);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = s[e] + bv[e];
      if (DROP) v *= df.mul((uint64_t)row * 64 + 16 * t + 4 * g + e);
      ov[t][e] = v + x1[t][e];
    }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) cmr_pin(ov[t]);
  if (valid) {
    float* yp = a.out + (int64_t)row * a.ldo + 4 * g;
    float* x1p = a.x1 + (int64_t)row * a.ldx1 + 4 * g;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      *reinterpret_cast<f32x4*>(yp + 16 * t) = ov[t];
      *reinterpret_cast<f32x4*>(x1p + 16 * t) = x1[t];
    }
  }
}

// ---- backward of the MLP half + LayerNorm + out-projection data gradient ------------------------------------------------------------
struct FfnBwdArgs {
  const float* dout; int64_t lddo;                     // gradient of the block output
  const float* x1; int64_t ldx1;                       // saved by the forward
  const float* g2; const float* b2n; float eps;
  const float* w1_f; const float* b1;                  // frag16 of W1 [1024, 64]          ([64][4] tiles)
  const float* w2t_f;                                  // frag16 of W2^T [1024, 64]        ([64][4] tiles): d hidden = dm W2
  const float* w1t_f;                                  // frag16 of W1^T [64, 1024]        ([4][64] tiles): d LN(x1) = du W1
  const float* wot_f;                                  // frag16 of Wo^T [64, 64]          ([4][4] tiles):  d ctx = da Wo
  float* dx1; int64_t lddx1;                           // gradient w.r.t. x1 (both uses: residual + LayerNorm)
  float* dctx; int64_t lddc;
  float* gs;                                           // [rows][1024] hidden activations after dropout (operand of dW2)
  float* du;                                           // [rows][1024] gradient at the fc1 pre-activations (operand of dW1, db1)
  float* h;                                            // [rows][64]   LN(x1)                              (operand of dW1)
  float* dm;                                           // [rows][64]   drop(d out)                         (operand of dW2, db2)
  float* da;                                           // [rows][64]   drop(d x1)                          (operand of dWo, dbo)
  float* lnpart;                                       // [tiles][128] per-tile sums of d gamma | d beta of ffn_norm
  uint32_t rows;
  const int64_t* seed; uint64_t site_proj, site_act, site_fc2;
  uint32_t thr_proj, thr_mlp; float ks_proj, ks_mlp;
};

template <bool DROP>
__global__ __launch_bounds__(512) void vit_ffn_bwd16_kernel(const FfnBwdArgs a) {
  __shared__ __attribute__((aligned(16))) float red[7 * 4 * 64 * 4];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, n = lane & 15;
  const uint32_t row = blockIdx.x * 16 + n;
  const bool valid = row < a.rows;
  const uint32_t rowc = valid ? row : 0;
  const float vmul = valid ? 1.f : 0.f;
  const float* dp_ = a.dout + (int64_t)rowc * a.lddo + 4 * g;
  const float* xp = a.x1 + (int64_t)rowc * a.ldx1 + 4 * g;
  f32x4 dO[4], x1[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    dO[t] = *reinterpret_cast<const f32x4*>(dp_ + 16 * t);
    x1[t] = *reinterpret_cast<const f32x4*>(xp + 16 * t);
  }
  M16Drop dp, da, df;
  if (DROP) {
    dp.init(a.seed, a.site_proj, a.thr_proj, a.ks_proj);
    da.init(a.seed, a.site_act, a.thr_mlp, a.ks_mlp);
    df.init(a.seed, a.site_fc2, a.thr_mlp, a.ks_mlp);
  }
  // rows past the end contribute nothing: their upstream gradient is zero
  f32x4 dm[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      dO[t][e] *= vmul;
      dm[t][e] = DROP ? dO[t][e] * df.mul((uint64_t)row * 64 + 16 * t + 4 * g + e) : dO[t][e];
    }
  // ---- recompute LayerNorm(x1): normalised values xh, rstd, and h = xh gamma + beta
  f32x4 xh[4], hn[4];
  float rstd;
  {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) s += (x1[t][0] + x1[t][1]) + (x1[t][2] + x1[t][3]);
    const float mean = m16_allg(s) * (1.f / 64.f);
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = x1[t][e] - mean;
        xh[t][e] = d;
        q += d * d;
      }
    rstd = 1.f / sqrtf(m16_allg(q) * (1.f / 64.f) + a.eps);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const f32x4 gv = *reinterpret_cast<const f32x4*>(a.g2 + 16 * t + 4 * g);
      const f32x4 bv = *reinterpret_cast<const f32x4*>(a.b2n + 16 * t + 4 * g);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        xh[t][e] *= rstd;
        hn[t][e] = xh[t][e] * gv[e] + bv[e];
      }
    }
  }
  // ---- this wave's 128 hidden units: pre-activations u (recomputed), d hidden = dm W2 (its slice), through dropout and GELU'
  f32x4 u[8], dg[8];
  m16_gemm<8, 4, 4>(a.w1_f, 4, 8 * wave, 0, lane, u, [&](int t, int r) { return hn[t][r]; });
  m16_gemm<8, 4, 4>(a.w2t_f, 4, 8 * wave, 0, lane, dg, [&](int t, int r) { return dm[t][r]; });
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const f32x4 bv = *reinterpret_cast<const f32x4*>(a.b1 + 128 * wave + 16 * t + 4 * g);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float uu = u[t][e] + bv[e];
      const float sc = DROP ? da.mul((uint64_t)row * 1024 + 128 * wave + 16 * t + 4 * g + e) : 1.f;
      float ge, gd;
      m16_gelu_both(uu, ge, gd);
      u[t][e] = ge * sc;                               // hidden activation as the forward's fc2 saw it
      dg[t][e] = dg[t][e] * sc * gd;                   // gradient at the pre-activation
    }
  }
#pragma unroll
  for (int t = 0; t < 8; ++t) { cmr_pin(u[t]); cmr_pin(dg[t]); }
  {                                                    // (every row output holds whole 16-row tiles: no predicated stores)
    float* gp = a.gs + (int64_t)row * 1024 + 128 * wave + 4 * g;
    float* up = a.du + (int64_t)row * 1024 + 128 * wave + 4 * g;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      *reinterpret_cast<f32x4*>(gp + 16 * t) = u[t];
      *reinterpret_cast<f32x4*>(up + 16 * t) = dg[t];
    }
  }
  // ---- d LN(x1): this wave's K-slice of du W1, summed over the waves through LDS in a fixed order
  f32x4 part[4];
  m16_gemm<4, 8, 8>(a.w1t_f, 64, 0, 8 * wave, lane, part, [&](int t, int r) { return dg[t][r]; });
  if (wave > 0) {
#pragma unroll
    for (int t = 0; t < 4; ++t) *reinterpret_cast<f32x4*>(&red[(((wave - 1) * 4 + t) * 64 + lane) * 4]) = part[t];
  }
  __syncthreads();
  if (wave != 0) return;
  f32x4 dh[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    f32x4 s = part[t];
#pragma unroll
    for (int w = 0; w < 7; ++w) s += *reinterpret_cast<const f32x4*>(&red[((w * 4 + t) * 64 + lane) * 4]);
    dh[t] = s;
  }
  // ---- LayerNorm backward: dx = rstd (dh gamma - mean(dh gamma) - xh mean(dh gamma xh));  d x1 = d out + dx
  f32x4 dx1[4], pg[4], pb[4];
  {
    float s1 = 0.f, s2 = 0.f;
    f32x4 dxh[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const f32x4 gv = *reinterpret_cast<const f32x4*>(a.g2 + 16 * t + 4 * g);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        dxh[t][e] = dh[t][e] * gv[e];
        s1 += dxh[t][e];
        s2 += dxh[t][e] * xh[t][e];
        pg[t][e] = dh[t][e] * xh[t][e];
        pb[t][e] = dh[t][e];
      }
    }
    s1 = m16_allg(s1) * (1.f / 64.f);
    s2 = m16_allg(s2) * (1.f / 64.f);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) dx1[t][e] = dO[t][e] + rstd * (dxh[t][e] - s1 - xh[t][e] * s2);
  }
  // per-tile sums of the LayerNorm parameter gradients over the tile's 16 rows (a DPP row = the 16 rows at fixed lane group)
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      pg[t][e] = m16_sum16(pg[t][e]);
      pb[t][e] = m16_sum16(pb[t][e]);
    }
  if (n == 0) {
    float* lp = a.lnpart + (int64_t)blockIdx.x * 128 + 4 * g;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      *reinterpret_cast<f32x4*>(lp + 16 * t) = pg[t];
      *reinterpret_cast<f32x4*>(lp + 64 + 16 * t) = pb[t];
    }
  }
  // ---- out-projection: da = drop(d x1), d ctx = da Wo
  f32x4 dav[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) dav[t][e] = DROP ? dx1[t][e] * dp.mul((uint64_t)row * 64 + 16 * t + 4 * g + e) : dx1[t][e];
  f32x4 dc[4];
  m16_gemm<4, 4, 4>(a.wot_f, 4, 0, 0, lane, dc, [&](int t, int r) { return dav[t][r]; });
#pragma unroll
  for (int t = 0; t < 4; ++t) { cmr_pin(dx1[t]); cmr_pin(dc[t]); }
  {
    const int64_t o = (int64_t)row * 64 + 4 * g;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      *reinterpret_cast<f32x4*>(a.dx1 + (int64_t)row * a.lddx1 + 4 * g + 16 * t) = dx1[t];
      *reinterpret_cast<f32x4*>(a.dctx + (int64_t)row * a.lddc + 4 * g + 16 * t) = dc[t];
      *reinterpret_cast<f32x4*>(a.h + o + 16 * t) = hn[t];
      *reinterpret_cast<f32x4*>(a.dm + o + 16 * t) = dm[t];
      *reinterpret_cast<f32x4*>(a.da + o + 16 * t) = dav[t];
    }
  }
}

// ---- backward of LayerNorm + q / k / v projections: 32-row tiles on v_mfma_f32_32x32x2_f32 (the layout of ln64_linear_kernel) ---------
// lane = 32 h + l owns row l of the tile; of k-group kg (8 channels) it holds channels 8 kg + 4 h + e.  A = W_cat^T fragments
// [2 out tiles][K / 8 k-groups][64 lanes][4] (cmr_pack_frags_f32 kind 0, transposed): d LN(x)[c] = sum_j d[j] W_cat[j][c].
struct LnQkvBwdProblem {
  const float* d; int64_t ldd; int kchunks;            // [rows][64 kchunks]: dq | (dk | dv) | (dq | dk | dv)
  const float* wt_f;                                   // frag32 of W_cat^T [64][64 kchunks]
  const float* x; int64_t ldx;                         // rows the LayerNorm saw
  const float* res; int64_t ldres;                     // optional: gradient arriving on the residual stream, added to dx
  float* dx; int64_t lddx;
  float* xn; int64_t ldxn;                             // LN(x) (operand of the projections' weight gradients)
  uint32_t rows;
};
struct LnQkvBwdArgs {
  LnQkvBwdProblem p[2];
  uint32_t tiles0, tiles;
  const float* g; const float* b; float eps;
  float* lnpart;                                       // [tiles][128] per-tile sums of d gamma | d beta
};

__device__ __forceinline__ float vt_sum32(float v) {   // over the 32 rows of a tile (lanes of one half)
  v = m16_sum16(v);
  return v + cmr_xor16(v);
}

__global__ __launch_bounds__(256) void vit_lnqkv_bwd_kernel(const LnQkvBwdArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  const uint32_t gt = blockIdx.x * (blockDim.x >> 6) + wave;
  if (gt >= a.tiles) return;
  const LnQkvBwdProblem& P = gt < a.tiles0 ? a.p[0] : a.p[1];
  const uint32_t tile = gt < a.tiles0 ? gt : gt - a.tiles0;
  const uint32_t row = tile * 32 + l31;
  const bool valid = row < P.rows;
  const uint32_t rowc = valid ? row : 0;
  const float vmul = valid ? 1.f : 0.f;
  const int kg_total = 8 * P.kchunks;
  f32x16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  for (int ch = 0; ch < P.kchunks; ++ch) {
    const float* dp = P.d + (int64_t)rowc * P.ldd + 64 * ch + 4 * h;
    f32x4 df[8];
#pragma unroll
    for (int kg = 0; kg < 8; ++kg) {
      df[kg] = *reinterpret_cast<const f32x4*>(dp + kg * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) df[kg][e] *= vmul;
    }
    const float* wp = P.wt_f + ((int64_t)8 * ch) * 256 + lane * 4;
    f32x4 wr[8][2];
#pragma unroll
    for (int kg = 0; kg < 8; ++kg)
#pragma unroll
      for (int t = 0; t < 2; ++t) wr[kg][t] = *reinterpret_cast<const f32x4*>(wp + ((int64_t)t * kg_total + kg) * 256);
#pragma unroll
    for (int kg = 0; kg < 8; ++kg)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t] = cmr_mfma32(wr[kg][t][j], df[kg][j], acc[t]);
  }
  // LayerNorm statistics of the row
  const float* xp = P.x + (int64_t)rowc * P.ldx + 4 * h;
  f32x4 xh[8];
#pragma unroll
  for (int kg = 0; kg < 8; ++kg) xh[kg] = *reinterpret_cast<const f32x4*>(xp + kg * 8);
  float s = 0.f;
#pragma unroll
  for (int kg = 0; kg < 8; ++kg) s += (xh[kg][0] + xh[kg][1]) + (xh[kg][2] + xh[kg][3]);
  s += cmr_xhalf(s);
  const float mean = s * (1.f / 64.f);
  float q = 0.f;
#pragma unroll
  for (int kg = 0; kg < 8; ++kg)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d = xh[kg][e] - mean;
      xh[kg][e] = d;
      q += d * d;
    }
  q += cmr_xhalf(q);
  const float rstd = 1.f / sqrtf(q * (1.f / 64.f) + a.eps);
  f32x4 dxh[8], xnv[8], pg[8], pb[8];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int kg = 0; kg < 8; ++kg) {
    const f32x4 gv = *reinterpret_cast<const f32x4*>(a.g + 8 * kg + 4 * h);
    const f32x4 bv = *reinterpret_cast<const f32x4*>(a.b + 8 * kg + 4 * h);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float dn = acc[kg / 4][4 * (kg % 4) + e];          // gradient at LN(x), channel 8 kg + 4 h + e
      xh[kg][e] *= rstd;
      xnv[kg][e] = xh[kg][e] * gv[e] + bv[e];
      dxh[kg][e] = dn * gv[e];
      s1 += dxh[kg][e];
      s2 += dxh[kg][e] * xh[kg][e];
      pg[kg][e] = dn * xh[kg][e];
      pb[kg][e] = dn;
    }
  }
  s1 += cmr_xhalf(s1);
  s2 += cmr_xhalf(s2);
  s1 *= (1.f / 64.f);
  s2 *= (1.f / 64.f);
  const float* rp = P.res ? P.res + (int64_t)rowc * P.ldres + 4 * h : nullptr;
  f32x4 ov[8];
#pragma unroll
  for (int kg = 0; kg < 8; ++kg) {
    f32x4 rv = {0.f, 0.f, 0.f, 0.f};
    if (rp) rv = *reinterpret_cast<const f32x4*>(rp + kg * 8);
#pragma unroll
    for (int e = 0; e < 4; ++e) ov[kg][e] = rstd * (dxh[kg][e] - s1 - xh[kg][e] * s2) + rv[e];
  }
#pragma unroll
  for (int kg = 0; kg < 8; ++kg)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      pg[kg][e] = vt_sum32(pg[kg][e]);
      pb[kg][e] = vt_sum32(pb[kg][e]);
    }
  if (l31 == 0) {
    float* lp = a.lnpart + (int64_t)gt * 128 + 4 * h;
#pragma unroll
    for (int kg = 0; kg < 8; ++kg) {
      *reinterpret_cast<f32x4*>(lp + 8 * kg) = pg[kg];
      *reinterpret_cast<f32x4*>(lp + 64 + 8 * kg) = pb[kg];
    }
  }
#pragma unroll
  for (int kg = 0; kg < 8; ++kg) { cmr_pin(ov[kg]); cmr_pin(xnv[kg]); }
  if (valid) {
    float* op = P.dx + (int64_t)row * P.lddx + 4 * h;
    float* np = P.xn + (int64_t)row * P.ldxn + 4 * h;
#pragma unroll
    for (int kg = 0; kg < 8; ++kg) {
      *reinterpret_cast<f32x4*>(op + 8 * kg) = ov[kg];
      *reinterpret_cast<f32x4*>(np + 8 * kg) = xnv[kg];
    }
  }
}

// ---- weights from the flat parameter bucket into MFMA fragment order, all slots of a step in one launch -------------------------------
// table [nslots][10] int64: src offset, n, k (shape of this slot's block of the packed matrix W'), row stride of the STORED matrix, dst
// offset of W', kind, transpose (the block = the transpose of the stored [k][n] matrix), elements, ktot (columns of the whole W': the
// fragment layout's tile stride), koff (first column of this block inside W'; row blocks are expressed through the dst offset).
// kind 0: 32x32x2 fragments [n/32][ktot/8][64][4], lane = 32 h + l holding W'[32 T + l][8 kg + 4 h + e] (_pack.frag_pack);
// kind 1: 16x16x4 fragments [n/16][ktot/16][64][4], lane = 16 g + m holding W'[16 To + m][16 T + 4 g + r] (_pack.frag_pack16);
// kind 2: plain copy of a vector of n floats.
__global__ __launch_bounds__(256) void pack_frags_kernel(const float* __restrict__ src, float* __restrict__ dst, const int64_t* __restrict__ table) {
  const int64_t* e = table + (int64_t)blockIdx.y * 10;
  const int64_t soff = e[0], n = e[1], k = e[2], ld = e[3], doff = e[4], kind = e[5], tr = e[6], total = e[7], ktot = e[8], koff = e[9];
  const int64_t q = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;        // first of this thread's four floats of the block
  if (q >= total) return;
  const float* s = src + soff;
  if (kind == 2) {
    float* o = dst + doff + q;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (q + i < n) o[i] = s[q + i];
    return;
  }
  const int lane = (int)((q >> 2) & 63);
  const int64_t blk = q >> 8;                                             // (tile, local k-tile) index inside the block
  int64_t row, col0, o;
  if (kind == 0) {
    const int64_t kgs = k / 8, T = blk / kgs, kg = blk % kgs;
    row = 32 * T + (lane & 31);
    col0 = 8 * kg + 4 * (lane >> 5);
    o = ((T * (ktot / 8) + koff / 8 + kg) * 64 + lane) * 4;
  } else {
    const int64_t kts = k / 16, T = blk / kts, kt = blk % kts;
    row = 16 * T + (lane & 15);
    col0 = 16 * kt + 4 * (lane >> 4);
    o = ((T * (ktot / 16) + koff / 16 + kt) * 64 + lane) * 4;
  }
  f32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = tr ? s[(col0 + i) * ld + row] : s[row * ld + col0 + i];
  *reinterpret_cast<f32x4*>(dst + doff + o) = v;
}

}  // namespace

extern "C" int cmr_vit_out_ffn16_train_f32(const float* ctx, int64_t ldc, const float* x, int64_t ldx, const float* wo_f16, const float* bo,
                                           const float* ln_g, const float* ln_b, float eps, const float* w1_f16, const float* b1,
                                           const float* w2_f16, const float* b2, float* out, int64_t ldo, float* x1, int64_t ldx1, int64_t rows,
                                           float p_proj, float p_mlp, const int64_t* seed, int64_t site_proj, int64_t site_act, int64_t site_fc2,
                                           hipStream_t stream) {
  CMR_REQUIRE(ctx && x && wo_f16 && bo && ln_g && ln_b && w1_f16 && b1 && w2_f16 && b2 && out && x1 && rows > 0 && rows < (int64_t)0x7fffffe0);
  CMR_REQUIRE(ldc % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && ldx1 % 4 == 0 && cmr_aligned16(ctx) && cmr_aligned16(x) && cmr_aligned16(out) &&
              cmr_aligned16(x1) && cmr_aligned16(wo_f16) && cmr_aligned16(w1_f16) && cmr_aligned16(w2_f16) && cmr_aligned16(bo) && cmr_aligned16(b1) &&
              cmr_aligned16(b2) && cmr_aligned16(ln_g) && cmr_aligned16(ln_b));
  const bool drop = seed != nullptr && (p_proj > 0.f || p_mlp > 0.f);
  CMR_REQUIRE(p_proj >= 0.f && p_proj < 1.f && p_mlp >= 0.f && p_mlp < 1.f);
  const OutFfnTrainArgs a{ctx, ldc, x, ldx, wo_f16, bo, ln_g, ln_b, eps, w1_f16, b1, w2_f16, b2, out, ldo, x1, ldx1, (uint32_t)rows, seed,
                          (uint64_t)site_proj, (uint64_t)site_act, (uint64_t)site_fc2, cmr_drop_threshold(p_proj), cmr_drop_threshold(p_mlp),
                          1.f / (1.f - p_proj), 1.f / (1.f - p_mlp)};
  const dim3 grid((unsigned)((rows + 15) / 16));
  if (drop) hipLaunchKernelGGL(vit_out_ffn16_train_kernel<true>, grid, dim3(512), 0, stream, a);
  else hipLaunchKernelGGL(vit_out_ffn16_train_kernel<false>, grid, dim3(512), 0, stream, a);
  return cmr_launch_status();
}

extern "C" int cmr_vit_ffn_bwd16_f32(const float* dout, int64_t lddo, const float* x1, int64_t ldx1, const float* ln_g, const float* ln_b, float eps,
                                     const float* w1_f16, const float* b1, const float* w2t_f16, const float* w1t_f16, const float* wot_f16,
                                     float* dx1, int64_t lddx1, float* dctx, int64_t lddc, float* gs, float* du, float* h, float* dm, float* da,
                                     float* lnpart, int64_t rows, float p_proj, float p_mlp, const int64_t* seed, int64_t site_proj,
                                     int64_t site_act, int64_t site_fc2, hipStream_t stream) {
  CMR_REQUIRE(dout && x1 && ln_g && ln_b && w1_f16 && b1 && w2t_f16 && w1t_f16 && wot_f16 && dx1 && dctx && gs && du && h && dm && da && lnpart);
  CMR_REQUIRE(rows > 0 && rows < (int64_t)0x7fffffe0 && lddo % 4 == 0 && ldx1 % 4 == 0 && lddx1 % 4 == 0 && lddc % 4 == 0);
  CMR_REQUIRE(cmr_aligned16(dout) && cmr_aligned16(x1) && cmr_aligned16(dx1) && cmr_aligned16(dctx) && cmr_aligned16(gs) && cmr_aligned16(du) &&
              cmr_aligned16(h) && cmr_aligned16(dm) && cmr_aligned16(da) && cmr_aligned16(lnpart) && cmr_aligned16(w1_f16) && cmr_aligned16(w2t_f16) &&
              cmr_aligned16(w1t_f16) && cmr_aligned16(wot_f16) && cmr_aligned16(b1) && cmr_aligned16(ln_g) && cmr_aligned16(ln_b));
  CMR_REQUIRE(p_proj >= 0.f && p_proj < 1.f && p_mlp >= 0.f && p_mlp < 1.f);
  const bool drop = seed != nullptr && (p_proj > 0.f || p_mlp > 0.f);
  const FfnBwdArgs a{dout, lddo, x1, ldx1, ln_g, ln_b, eps, w1_f16, b1, w2t_f16, w1t_f16, wot_f16, dx1, lddx1, dctx, lddc, gs, du, h, dm, da, lnpart,
                     (uint32_t)rows, seed, (uint64_t)site_proj, (uint64_t)site_act, (uint64_t)site_fc2, cmr_drop_threshold(p_proj),
                     cmr_drop_threshold(p_mlp), 1.f / (1.f - p_proj), 1.f / (1.f - p_mlp)};
  const dim3 grid((unsigned)((rows + 15) / 16));
  if (drop) hipLaunchKernelGGL(vit_ffn_bwd16_kernel<true>, grid, dim3(512), 0, stream, a);
  else hipLaunchKernelGGL(vit_ffn_bwd16_kernel<false>, grid, dim3(512), 0, stream, a);
  return cmr_launch_status();
}

extern "C" int cmr_vit_lnqkv_bwd_f32(const float* d_x, int64_t ldd_x, int k_x, const float* wt_f_x, const float* x, int64_t ldx, const float* res,
                                     int64_t ldres, float* dx, int64_t lddx, float* xn, int64_t ldxn, int64_t rows_x, const float* d_y,
                                     int64_t ldd_y, int k_y, const float* wt_f_y, const float* y, int64_t ldy, float* dy, int64_t lddy, float* yn,
                                     int64_t ldyn, int64_t rows_y, const float* gamma, const float* beta, float eps, float* lnpart,
                                     hipStream_t stream) {
  CMR_REQUIRE(d_x && wt_f_x && x && dx && xn && gamma && beta && lnpart && rows_x > 0 && rows_x < (int64_t)0x7fffffc0);
  CMR_REQUIRE((k_x == 64 || k_x == 128 || k_x == 192) && ldd_x % 4 == 0 && ldx % 4 == 0 && lddx % 4 == 0 && ldxn % 4 == 0);
  CMR_REQUIRE(cmr_aligned16(d_x) && cmr_aligned16(wt_f_x) && cmr_aligned16(x) && cmr_aligned16(dx) && cmr_aligned16(xn) && cmr_aligned16(gamma) &&
              cmr_aligned16(beta) && cmr_aligned16(lnpart) && (!res || (cmr_aligned16(res) && ldres % 4 == 0)));
  LnQkvBwdArgs a{};
  a.p[0] = LnQkvBwdProblem{d_x, ldd_x, k_x / 64, wt_f_x, x, ldx, res, ldres, dx, lddx, xn, ldxn, (uint32_t)rows_x};
  a.tiles0 = (uint32_t)((rows_x + 31) / 32);
  a.tiles = a.tiles0;
  if (d_y) {
    CMR_REQUIRE(wt_f_y && y && dy && yn && rows_y > 0 && rows_y < (int64_t)0x7fffffc0 && (k_y == 64 || k_y == 128 || k_y == 192));
    CMR_REQUIRE(ldd_y % 4 == 0 && ldy % 4 == 0 && lddy % 4 == 0 && ldyn % 4 == 0 && cmr_aligned16(d_y) && cmr_aligned16(wt_f_y) && cmr_aligned16(y) &&
                cmr_aligned16(dy) && cmr_aligned16(yn));
    a.p[1] = LnQkvBwdProblem{d_y, ldd_y, k_y / 64, wt_f_y, y, ldy, nullptr, 0, dy, lddy, yn, ldyn, (uint32_t)rows_y};
    a.tiles += (uint32_t)((rows_y + 31) / 32);
  } else {
    a.p[1] = a.p[0];
  }
  a.g = gamma; a.b = beta; a.eps = eps; a.lnpart = lnpart;
  if (a.tiles <= 2048) hipLaunchKernelGGL(vit_lnqkv_bwd_kernel, dim3(a.tiles), dim3(64), 0, stream, a);
  else hipLaunchKernelGGL(vit_lnqkv_bwd_kernel, dim3((a.tiles + 3) / 4), dim3(256), 0, stream, a);
  return cmr_launch_status();
}

extern "C" int cmr_pack_frags_f32(const float* src, float* dst, const int64_t* table, int nslots, int64_t max_elements, hipStream_t stream) {
  CMR_REQUIRE(src && dst && table && nslots > 0 && nslots <= 65535 && max_elements > 0 && cmr_aligned16(dst));
  hipLaunchKernelGGL(pack_frags_kernel, dim3((unsigned)((max_elements / 4 + 255) / 256), (unsigned)nslots), dim3(256), 0, stream, src, dst, table);
  return cmr_launch_status();
}
