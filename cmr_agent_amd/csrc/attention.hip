// Attention kernels of the coarse (softmax) and fine (linear) matchers.
//
// cmr_mha_f32        softmax(Q K^T / sqrt(dh)) V for 8 heads x 8 dims on <= ~2k tokens
//                    (ImageViT.py:81-108, PointViT.py:117-140, IMGPCEncoder.py:36-58).  K/V of one head live in
//                    LDS.  Default: mha_mfma_kernel -- both contractions on v_mfma_f32_16x16x4_f32, the score tile
//                    stays in the accumulator registers and IS the operand of the second contraction, online softmax
//                    (13 us at 8 x 418 x 256 tokens).  mha_kernel (one query per 4 lanes on the vector ALUs, two passes:
//                    max, then exp / sum; 19 us) stays selectable through cmr_set_mha_variant for A/B runs.
// cmr_la_reduce_f32  KV[h,d,v] = sum_s K~[s,h,d] * V[s,h,v] / S  and  Ksum[h,d] = sum_s K~[s,h,d]
// cmr_la_apply_f32   msg[l,h,v] = (Q~[l,h,:] . KV[h,:,v]) * S / (Q~[l,h,:] . Ksum[h,:] + eps)
//                    (LinearAttention.py:53-60; K~,Q~ = elu+1 are produced by the projection epilogue).
#include "cmr_common.h"
#include <cstdlib>

namespace {

constexpr int DH = 8, NH = 8, CH = 64;

// 256 threads = 64 queries of one (batch, head) x 4 lanes; lane `sub` of a query takes keys sub, sub+4, ...
// and the four partial (max, sum, weighted value) states are merged with two xor-shuffles each.
// DROP: dropout on the attention probabilities (ImageViT.py:100, PointViT.py:129, IMGPCEncoder.py:47, train mode): the softmax
// normaliser runs over all keys, the value sum over the kept ones scaled by 1 / (1 - p); mask element = ((b 8 + head) Tq + query) Tk + key.
constexpr int MHA_Q = 64, MHA_SPLIT = 4;
template <bool DROP>
__global__ __launch_bounds__(256) void mha_kernel(const float* __restrict__ q, int64_t ldq, const float* __restrict__ k,
                                                  int64_t ldk, const float* __restrict__ v, int64_t ldv,
                                                  float* __restrict__ o, int64_t ldo, int Tq, int Tk, float scale, uint32_t thr = 0,
                                                  float keep_scale = 1.f, const int64_t* __restrict__ seed_ptr = nullptr, uint64_t site = 0) {
  extern __shared__ __attribute__((aligned(16))) float kv[];  // K_h [Tk][8] then V_h [Tk][8]
  float* ks = kv;
  float* vs = kv + (size_t)Tk * DH;
  const int head = blockIdx.y, b = blockIdx.z;
  const float* kb = k + (int64_t)b * Tk * ldk + head * DH;
  const float* vb = v + (int64_t)b * Tk * ldv + head * DH;
  for (int e = threadIdx.x; e < Tk * 2; e += 256) {
    const int t = e >> 1, half = (e & 1) * 4;
    *reinterpret_cast<f32x4*>(&ks[t * DH + half]) = *reinterpret_cast<const f32x4*>(kb + (int64_t)t * ldk + half);
    *reinterpret_cast<f32x4*>(&vs[t * DH + half]) = *reinterpret_cast<const f32x4*>(vb + (int64_t)t * ldv + half);
  }
  __syncthreads();
  const int sub = threadIdx.x & (MHA_SPLIT - 1);
  int tq = blockIdx.x * MHA_Q + (threadIdx.x >> 2);
  const bool qvalid = tq < Tq;
  if (!qvalid) tq = Tq - 1;                       // keep the lane alive for the shuffles
  const float* qp = q + ((int64_t)b * Tq + tq) * ldq + head * DH;
  const f32x4 q0 = *reinterpret_cast<const f32x4*>(qp), q1 = *reinterpret_cast<const f32x4*>(qp + 4);
  auto score = [&](int t) {
    const f32x4 k0 = *reinterpret_cast<const f32x4*>(&ks[t * DH]);
    const f32x4 k1 = *reinterpret_cast<const f32x4*>(&ks[t * DH + 4]);
    float s = q0[0] * k0[0];
    s += q0[1] * k0[1]; s += q0[2] * k0[2]; s += q0[3] * k0[3];
    s += q1[0] * k1[0]; s += q1[1] * k1[1]; s += q1[2] * k1[2]; s += q1[3] * k1[3];
    return s * scale;
  };
  float m = -INFINITY;
  for (int t = sub; t < Tk; t += MHA_SPLIT) m = fmaxf(m, score(t));
  m = fmaxf(m, __shfl_xor(m, 1));
  m = fmaxf(m, __shfl_xor(m, 2));
  float l = 0.f;
  float acc[DH] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const uint64_t seed = DROP ? (uint64_t)seed_ptr[0] : 0;
  const uint64_t mrow = (((uint64_t)b * NH + head) * Tq + tq) * Tk;
  for (int t = sub; t < Tk; t += MHA_SPLIT) {
    const float p = expf(score(t) - m);
    l += p;
    const float pd = DROP ? (cmr_keep(seed, site, mrow + t, thr) ? p * keep_scale : 0.f) : p;
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(&vs[t * DH]);
    const f32x4 v1 = *reinterpret_cast<const f32x4*>(&vs[t * DH + 4]);
#pragma unroll
    for (int i = 0; i < 4; ++i) { acc[i] += pd * v0[i]; acc[4 + i] += pd * v1[i]; }
  }
  l += __shfl_xor(l, 1);
  l += __shfl_xor(l, 2);
#pragma unroll
  for (int i = 0; i < DH; ++i) {
    acc[i] += __shfl_xor(acc[i], 1);
    acc[i] += __shfl_xor(acc[i], 2);
  }
  if (!qvalid || sub != 0) return;
  const float inv = 1.f / l;
  float* op = o + ((int64_t)b * Tq + tq) * ldo + head * DH;
  f32x4 o0 = {acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv};
  f32x4 o1 = {acc[4] * inv, acc[5] * inv, acc[6] * inv, acc[7] * inv};
  *reinterpret_cast<f32x4*>(op) = o0;
  *reinterpret_cast<f32x4*>(op + 4) = o1;
}


// ---- the same attention on the matrix cores (v_mfma_f32_16x16x4_f32) ---------------------------------------------------
// One wave = 16 queries of one (batch, head); a step = 16 keys.  S^T = K Q^T (keys on the rows, queries on the columns) is
// two MFMAs (8 head dims = 2 x 4): lane (i, g) = (l & 15, l >> 4) supplies K[key i][2g + s] and Q[query i][2g + s] in step s.  In
// the 16x16 accumulator layout lane (q, g) then holds the scores of query q for the keys 4g .. 4g + 3 -- which is exactly the B
// operand of the next contraction, O^T += V^T P^T (4 MFMAs, one per register: contraction index = lane group g <-> key 4g + r),
// with A = V^T[d][key] read as ONE b128 per lane from a transposed V image in LDS (rows 8.. are a zero row: 8 of the 16 output
// rows are head dims).  Online softmax: running max / sum per query; the max is combined over the four lane groups with two
// xor-shuffles per step, the sums once at the end.  The output tile has the head dims on the rows: lane (q, g < 2) ends up with
// O[q][4g .. 4g + 3] -- one b128 store.  No P tile ever leaves the registers.
typedef float mha_f32x4 __attribute__((ext_vector_type(4)));
// Keys are staged in chunks of at most MHA_KCH (the online softmax simply runs on across chunks): the K / V image of a head with 1 400
// keys (nuScenes: 28 x 50 proxies) is 95 KB -- ONE workgroup per CU, one wave per SIMD, and nothing to cover the dependent
// LDS -> MFMA -> exp -> MFMA chain of a step; 512-key chunks are 35 KB, four workgroups share a CU.  Up to 512 keys (KITTI: 418 / 256
// proxies) there is one chunk and the kernel is the round-2 one.
constexpr int MHA_KCH = 512;
// LIBM_EXP: exp through libm's expf instead of v_exp_f32(x log2 e) (cmr_mha_expf_f32: the training tape's no-dropout forward, whose
// backward kernels recompute the probabilities with expf)
template <bool LIBM_EXP>
__device__ __forceinline__ float mha_exp(float x) {
  if constexpr (LIBM_EXP) return expf(x);
  else return __builtin_amdgcn_exp2f(x * 1.4426950408889634f);
}
template <bool LIBM_EXP>
__global__ __launch_bounds__(256) void mha_mfma_kernel(const float* __restrict__ q, int64_t ldq, const float* __restrict__ k, int64_t ldk,
                                                       const float* __restrict__ v, int64_t ldv, float* __restrict__ o, int64_t ldo, int Tq,
                                                       int Tk, int Tcp, float scale) {
  extern __shared__ __attribute__((aligned(16))) float kv[];  // K_h [Tcp][8], then V_h^T [9][Tcp] (row 8 = zeros); Tcp = padded chunk length
  float* ks = kv;
  float* vt = kv + (size_t)Tcp * DH;
  const int head = blockIdx.y, b = blockIdx.z;
  const float* kb = k + (int64_t)b * Tk * ldk + head * DH;
  const float* vb = v + (int64_t)b * Tk * ldv + head * DH;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int q0 = (blockIdx.x * 4 + wave) * 16;
  const bool active = q0 < Tq;                                   // (wave-uniform) idle waves still stage and meet the barriers
  const int tq = q0 + i < Tq ? q0 + i : Tq - 1;
  const float* qp = q + ((int64_t)b * Tq + tq) * ldq + head * DH + 2 * g;
  const float qa = qp[0], qb = qp[1];
  const float* krow = ks + i * DH + 2 * g;                       // + k0 * DH
  const float* vrow = vt + (i < 8 ? i : 8) * Tcp + 4 * g;        // + k0
  float m = -INFINITY, l = 0.f;
  mha_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int t = threadIdx.x; t < Tcp; t += 256) vt[8 * Tcp + t] = 0.f;
  for (int c0 = 0; c0 < Tk; c0 += Tcp) {
    if (c0 > 0) __syncthreads();                                 // everybody done with the previous chunk
    for (int e = threadIdx.x; e < Tcp * 2; e += 256) {
      const int t = e >> 1, half = (e & 1) * 4;
      f32x4 kk = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
      if (c0 + t < Tk) {
        kk = *reinterpret_cast<const f32x4*>(kb + (int64_t)(c0 + t) * ldk + half);
        vv = *reinterpret_cast<const f32x4*>(vb + (int64_t)(c0 + t) * ldv + half);
      }
      *reinterpret_cast<f32x4*>(&ks[t * DH + half]) = kk;
#pragma unroll
      for (int j = 0; j < 4; ++j) vt[(half + j) * Tcp + t] = vv[j];
    }
    __syncthreads();
    if (!active) continue;
    const int nk = Tk - c0 < Tcp ? Tk - c0 : Tcp;                // keys of this chunk
    for (int k0 = 0; k0 < nk; k0 += 32) {                        // two 16-key steps per running-max update
      const float2 ka0 = *reinterpret_cast<const float2*>(krow + k0 * DH);
      const float2 ka1 = *reinterpret_cast<const float2*>(krow + (k0 + 16) * DH);
      const mha_f32x4 vf0 = *reinterpret_cast<const mha_f32x4*>(vrow + k0);
      const mha_f32x4 vf1 = *reinterpret_cast<const mha_f32x4*>(vrow + k0 + 16);
      mha_f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
      s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ka0.x, qa, s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ka1.x, qa, s1, 0, 0, 0);
      s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ka0.y, qb, s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ka1.y, qb, s1, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s0[r] = k0 + 4 * g + r < nk ? s0[r] * scale : -INFINITY;
        s1[r] = k0 + 16 + 4 * g + r < nk ? s1[r] * scale : -INFINITY;
      }
      float bm = fmaxf(fmaxf(fmaxf(s0[0], s0[1]), fmaxf(s0[2], s0[3])), fmaxf(fmaxf(s1[0], s1[1]), fmaxf(s1[2], s1[3])));
      bm = fmaxf(bm, cmr_xor16(bm));                              // (VALU lane exchanges: two LDS round trips sat on the
      bm = fmaxf(bm, cmr_xhalf(bm));                              //  serial max -> exp -> multiply chain of every key block)
      const float mn = fmaxf(m, bm);                             // finite from the first step on (key 0 of a chunk is always valid)
      // exp(x) as v_exp_f32(x log2 e): libm's expf spends five more instructions per value on a range reduction that a softmax does not
      // need (x <= 0; terms below 2^-126 vanish either way) -- 20 -> 16 us at 418 keys, 80 -> 61 us at 1 400
      const float alpha = mha_exp<LIBM_EXP>(m - mn);
      m = mn;
      mha_f32x4 p0, p1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        p0[r] = mha_exp<LIBM_EXP>(s0[r] - mn);
        p1[r] = mha_exp<LIBM_EXP>(s1[r] - mn);
      }
      l = l * alpha + (((p0[0] + p0[1]) + (p0[2] + p0[3])) + ((p1[0] + p1[1]) + (p1[2] + p1[3])));
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] *= alpha;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(vf0[r], p0[r], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(vf1[r], p1[r], acc, 0, 0, 0);
    }
  }
  if (!active) return;
  l += cmr_xor16(l);
  l += cmr_xhalf(l);
  if (g < 2 && q0 + i < Tq) {
    const float inv = 1.f / l;
    f32x4 out = {acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv};
    *reinterpret_cast<f32x4*>(o + ((int64_t)b * Tq + q0 + i) * ldo + head * DH + 4 * g) = out;
  }
}

// ---- LayerNorm + Q / K / V projections + attention in one launch (inference: ops.mha_ln, one launch less per transformer block) ---------
// The projections of a block are 64 -> 192 on a few thousand rows: as their own launch (ln64_linear) they are a 12 us round trip through
// HBM in front of a 13 us attention launch.  Here every workgroup (64 queries of one (batch, head)) normalises the source rows itself
// and projects them onto ITS head's 8 + 8 key / value dims while it stages them -- 16 MFMAs per 16 source rows, recomputed by the
// ~7 workgroups that share a (batch, head): a few microseconds, no launch, no q / k / v arrays.  Layouts (v_mfma_f32_16x16x4_f32):
//   source tile of 16 rows: lane (i, g) holds the normalised channels 16g .. 16g+15 of row i; MFMA step s contracts channel 16g + s of
//     lane group g (A = rows); B = wkv[head][s][lane (n, g)] = W_k|v[8 head + n][16g + s] (n < 8: key dim n, else value dim n - 8).
//     The accumulator of lane (n, g) then holds rows 4g .. 4g+3 of output column n -> K image / transposed V image in LDS.
//   queries: Q^T = W_q x^T with the rows of W_q arranged so that lane (query, g) ends up with dims 2g, 2g+1 in its first two
//     accumulator registers -- exactly the (qa, qb) operands of the score MFMAs above.
struct MhaLnArgs {
  const float* x; int64_t ldx; const float* y; int64_t ldy;      // query rows [B*Tq], source rows [B*Tk] (y == x: self-attention)
  const float *gamma, *beta; float eps;
  const float *wq, *wkv;                                          // [8 heads][16 steps][64 lanes]
  const float *bq, *bk, *bv;                                      // [64] each
  float* o; int64_t ldo; int Tq, Tk, Tcp; float scale;
};

// normalised channels 16g .. 16g+15 of one row (LayerNorm over all 64: the other 48 sit in the three partner lanes); the row's 16 raw
// channels are loaded by the caller (mha_ln_load), a tile ahead of their use
__device__ __forceinline__ void mha_ln_load(const float* __restrict__ rp, int g, f32x4 (&v)[4]) {
#pragma unroll
  for (int t = 0; t < 4; ++t) v[t] = *reinterpret_cast<const f32x4*>(rp + 16 * g + 4 * t);
}
__device__ __forceinline__ void mha_ln_norm(const f32x4 (&v)[4], const float* __restrict__ gamma, const float* __restrict__ beta, float eps, int g,
                                            float (&xn)[16]) {
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 4; ++t) s += (v[t][0] + v[t][1]) + (v[t][2] + v[t][3]);
  s += cmr_xor16(s);
  s += cmr_xhalf(s);
  const float mean = s * (1.f / 64.f);
  float q = 0.f;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d = v[t][e] - mean;
      xn[4 * t + e] = d;
      q += d * d;
    }
  q += cmr_xor16(q);
  q += cmr_xhalf(q);
  const float rstd = 1.f / sqrtf(q * (1.f / 64.f) + eps);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const f32x4 gv = *reinterpret_cast<const f32x4*>(gamma + 16 * g + 4 * t);
    const f32x4 bv = *reinterpret_cast<const f32x4*>(beta + 16 * g + 4 * t);
#pragma unroll
    for (int e = 0; e < 4; ++e) xn[4 * t + e] = xn[4 * t + e] * rstd * gv[e] + bv[e];
  }
}

__global__ __launch_bounds__(256) void mha_ln_kernel(const MhaLnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float kv[];  // K_h [Tcp][8], then V_h^T [9][Tcp] (row 8 = zeros)
  const int Tcp = a.Tcp, Tq = a.Tq, Tk = a.Tk;
  float* ks = kv;
  float* vt = kv + (size_t)Tcp * DH;
  const int head = blockIdx.y, b = blockIdx.z;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int q0 = (blockIdx.x * 4 + wave) * 16;
  const bool active = q0 < Tq;
  // this head's projection fragments: one float per lane and step
  float wkv[16], wq[16];
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    wkv[s] = a.wkv[(head * 16 + s) * 64 + lane];
    wq[s] = a.wq[(head * 16 + s) * 64 + lane];
  }
  const float bkv = i < 8 ? a.bk[head * DH + i] : a.bv[head * DH + i - 8];
  // ---- queries of this wave: qa, qb = Q[query i][2g], Q[query i][2g + 1]
  float qa, qb;
  {
    const int tq = q0 + i < Tq ? q0 + i : Tq - 1;
    float xn[16];
    f32x4 raw[4];
    mha_ln_load(a.x + ((int64_t)b * Tq + tq) * a.ldx, g, raw);
    mha_ln_norm(raw, a.gamma, a.beta, a.eps, g, xn);
    mha_f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 16; s += 2) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[s], xn[s], c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[s + 1], xn[s + 1], c1, 0, 0, 0);
    }
    qa = (c0[0] + c1[0]) + a.bq[head * DH + 2 * g];
    qb = (c0[1] + c1[1]) + a.bq[head * DH + 2 * g + 1];
  }
  const float* krow = ks + i * DH + 2 * g;                       // + k0 * DH
  const float* vrow = vt + (i < 8 ? i : 8) * Tcp + 4 * g;        // + k0
  float m = -INFINITY, l = 0.f;
  mha_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int t = threadIdx.x; t < Tcp; t += 256) vt[8 * Tcp + t] = 0.f;
  const float* yb = a.y + (int64_t)b * Tk * a.ldy;
  for (int c0k = 0; c0k < Tk; c0k += Tcp) {
    if (c0k > 0) __syncthreads();                                // everybody done with the previous chunk
    // ---- stage the chunk: wave w projects the 16-row source tiles w, w + 4, ... (rows past the end: zeros)
    // (the next tile's rows are requested before this tile is normalised and multiplied: one memory round trip per chunk, not per tile)
    auto src_row = [&](int t0) __attribute__((always_inline)) {
      const int src = c0k + t0 + i;
      return yb + (int64_t)(src < Tk ? src : Tk - 1) * a.ldy;
    };
    f32x4 nxt[4];
    mha_ln_load(src_row(wave * 16), g, nxt);
    for (int t0 = wave * 16; t0 < Tcp; t0 += 64) {
      f32x4 cur[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) cur[t] = nxt[t];
      mha_ln_load(src_row(t0 + 64 < Tcp ? t0 + 64 : t0), g, nxt);
      float xn[16];
      mha_ln_norm(cur, a.gamma, a.beta, a.eps, g, xn);
      mha_f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 16; s += 2) {
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xn[s], wkv[s], d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xn[s + 1], wkv[s + 1], d1, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int t = t0 + 4 * g + r;                            // row of the chunk; column i of the projection
        const float val = c0k + t < Tk ? (d0[r] + d1[r]) + bkv : 0.f;
        if (i < 8) ks[t * DH + i] = val;
        else vt[(i - 8) * Tcp + t] = val;
      }
    }
    __syncthreads();
    if (!active) continue;
    const int nk = Tk - c0k < Tcp ? Tk - c0k : Tcp;
    for (int k0 = 0; k0 < nk; k0 += 32) {
      const float2 ka0 = *reinterpret_cast<const float2*>(krow + k0 * DH);
      const float2 ka1 = *reinterpret_cast<const float2*>(krow + (k0 + 16) * DH);
      const mha_f32x4 vf0 = *reinterpret_cast<const mha_f32x4*>(vrow + k0);
      const mha_f32x4 vf1 = *reinterpret_cast<const mha_f32x4*>(vrow + k0 + 16);
      mha_f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
      s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ka0.x, qa, s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ka1.x, qa, s1, 0, 0, 0);
      s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ka0.y, qb, s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ka1.y, qb, s1, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s0[r] = k0 + 4 * g + r < nk ? s0[r] * a.scale : -INFINITY;
        s1[r] = k0 + 16 + 4 * g + r < nk ? s1[r] * a.scale : -INFINITY;
      }
      float bm = fmaxf(fmaxf(fmaxf(s0[0], s0[1]), fmaxf(s0[2], s0[3])), fmaxf(fmaxf(s1[0], s1[1]), fmaxf(s1[2], s1[3])));
      bm = fmaxf(bm, cmr_xor16(bm));
      bm = fmaxf(bm, cmr_xhalf(bm));
      const float mn = fmaxf(m, bm);
      const float alpha = mha_exp<false>(m - mn);
      m = mn;
      mha_f32x4 p0, p1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        p0[r] = mha_exp<false>(s0[r] - mn);
        p1[r] = mha_exp<false>(s1[r] - mn);
      }
      l = l * alpha + (((p0[0] + p0[1]) + (p0[2] + p0[3])) + ((p1[0] + p1[1]) + (p1[2] + p1[3])));
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] *= alpha;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(vf0[r], p0[r], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(vf1[r], p1[r], acc, 0, 0, 0);
    }
  }
  if (!active) return;
  l += cmr_xor16(l);
  l += cmr_xhalf(l);
  if (g < 2 && q0 + i < Tq) {
    const float inv = 1.f / l;
    f32x4 out = {acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv};
    *reinterpret_cast<f32x4*>(a.o + ((int64_t)b * Tq + q0 + i) * a.ldo + head * DH + 4 * g) = out;
  }
}

// partial sums over a slab of TS tokens: thread (h,d,v) accumulates K[s,h,d]*V[s,h,v]/S; threads with
// v == 0 also accumulate Ksum.  part layout: [B][nslab][576]  (512 KV entries then 64 Ksum entries)
constexpr int LA_TS = 32;
__global__ __launch_bounds__(512) void la_reduce_partial_kernel(const float* __restrict__ kf, int64_t ldk,
                                                                const float* __restrict__ v, int64_t ldv,
                                                                float* __restrict__ part, int S, int slab_tokens) {
  __shared__ __attribute__((aligned(16))) float ks[LA_TS * CH];
  __shared__ __attribute__((aligned(16))) float vs[LA_TS * CH];
  const int b = blockIdx.y, slab = blockIdx.x, tid = threadIdx.x;
  const int hh = tid >> 6, d = (tid >> 3) & 7, vv = tid & 7;
  const int s_begin = slab * slab_tokens;
  const int s_end = min(S, s_begin + slab_tokens);
  const float invS = (float)S;
  float acc = 0.f, ksum = 0.f;
  for (int s0 = s_begin; s0 < s_end; s0 += LA_TS) {
    __syncthreads();
    {  // 32 tokens x 16 float4 = 512 float4 per tensor: one per thread
      const int t = tid >> 4, c = (tid & 15) * 4;
      f32x4 a = {0.f, 0.f, 0.f, 0.f}, bq = {0.f, 0.f, 0.f, 0.f};
      if (s0 + t < s_end) {
        a = *reinterpret_cast<const f32x4*>(kf + ((int64_t)b * S + s0 + t) * ldk + c);
        bq = *reinterpret_cast<const f32x4*>(v + ((int64_t)b * S + s0 + t) * ldv + c);
        bq[0] = bq[0] / invS; bq[1] = bq[1] / invS; bq[2] = bq[2] / invS; bq[3] = bq[3] / invS;  // value / v_length
      }
      *reinterpret_cast<f32x4*>(&ks[t * CH + c]) = a;
      *reinterpret_cast<f32x4*>(&vs[t * CH + c]) = bq;
    }
    __syncthreads();
#pragma unroll 8
    for (int t = 0; t < LA_TS; ++t) {
      const float kk = ks[t * CH + hh * 8 + d];
      acc += kk * vs[t * CH + hh * 8 + vv];
      ksum += kk;
    }
  }
  float* p = part + ((int64_t)b * gridDim.x + slab) * 576;
  p[tid] = acc;
  if (vv == 0) p[512 + hh * 8 + d] = ksum;
}

__global__ __launch_bounds__(576) void la_reduce_final_kernel(const float* __restrict__ part, float* __restrict__ kvsum,
                                                              int nslab) {
  const int b = blockIdx.x, tid = threadIdx.x;
  float s = 0.f;
  for (int i = 0; i < nslab; ++i) s += part[((int64_t)b * nslab + i) * 576 + tid];
  kvsum[(int64_t)b * 576 + tid] = s;
}

// one wave per token group; lane = output channel c = h*8+v
__global__ __launch_bounds__(256) void la_apply_kernel(const float* __restrict__ qf, int64_t ldq,
                                                       const float* __restrict__ kvsum, float* __restrict__ msg,
                                                       int64_t ldm, int L, int S, float eps, int tokens_per_wave) {
  const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hh = lane >> 3, vv = lane & 7;
  const float* kvb = kvsum + (int64_t)b * 576;
  float kvr[8], ksr[8];
#pragma unroll
  for (int d = 0; d < 8; ++d) {
    kvr[d] = kvb[hh * 64 + d * 8 + vv];
    ksr[d] = kvb[512 + hh * 8 + d];
  }
  const float fs = (float)S;
  const int l0 = (blockIdx.x * 4 + wave) * tokens_per_wave;
  const int l1 = min(L, l0 + tokens_per_wave);
  for (int l = l0; l < l1; ++l) {
    const float qv = qf[((int64_t)b * L + l) * ldq + lane];
    float num = 0.f, den = 0.f;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      const float qd = __shfl(qv, hh * 8 + d);
      num += qd * kvr[d];
      den += qd * ksr[d];
    }
    const float z = 1.f / (den + eps);
    msg[((int64_t)b * L + l) * ldm + lane] = num * z * fs;
  }
}

}  // namespace

#ifdef CMR_AB_SWITCHES          // A/B build only (libcmr_hip_ab.so): the product library dispatches by constants
static int CMR_MHA_KEY_CHUNK = MHA_KCH;   // keys staged per chunk (env CMR_MHA_KEY_CHUNK at first use)
static int CMR_MHA_MFMA = 1;     // 1 = matrix-core kernel (default), 0 = the one-query-per-lane-group VALU kernel
extern "C" int cmr_set_mha_variant(int mfma) {
  const int old = CMR_MHA_MFMA;
  CMR_MHA_MFMA = mfma;
  return old;
}
#else
static constexpr int CMR_MHA_KEY_CHUNK = MHA_KCH, CMR_MHA_MFMA = 1;
#endif

static int mha_launch(bool libm_exp, const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                           float* o, int64_t ldo, int B, int Tq, int Tk, hipStream_t stream) {
  CMR_REQUIRE(q && k && v && o && B > 0 && B <= 65535 && Tq > 0 && Tk > 0);
  CMR_REQUIRE(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && ldo % 4 == 0);
  CMR_REQUIRE(cmr_aligned16(q) && cmr_aligned16(k) && cmr_aligned16(v) && cmr_aligned16(o));
  if (CMR_MHA_MFMA) {
#ifdef CMR_AB_SWITCHES
    static const bool kch_env = [] { const char* e = getenv("CMR_MHA_KEY_CHUNK"); if (e && atoi(e) >= 32) CMR_MHA_KEY_CHUNK = atoi(e) / 32 * 32; return true; }();
    (void)kch_env;
#endif
    const int Tkp = (Tk + 31) / 32 * 32;
    const int Tcp = Tkp < CMR_MHA_KEY_CHUNK ? Tkp : CMR_MHA_KEY_CHUNK;      // chunk length (the last chunk may be shorter)
    const size_t smem = (size_t)Tcp * (DH + 9) * sizeof(float);
    CMR_REQUIRE(smem <= 160 * 1024);
    dim3 grid((Tq + 63) / 64, NH, B);
    if (libm_exp) {
      static CmrSmemCache granted_l{};
      if (cmr_grant_smem(reinterpret_cast<const void*>(mha_mfma_kernel<true>), smem, granted_l) != CMR_OK) return CMR_ELAUNCH;
      hipLaunchKernelGGL(mha_mfma_kernel<true>, grid, dim3(256), smem, stream, q, ldq, k, ldk, v, ldv, o, ldo, Tq, Tk, Tcp, 0.35355339059327373f);
    } else {
      static CmrSmemCache granted{};
      if (cmr_grant_smem(reinterpret_cast<const void*>(mha_mfma_kernel<false>), smem, granted) != CMR_OK) return CMR_ELAUNCH;
      hipLaunchKernelGGL(mha_mfma_kernel<false>, grid, dim3(256), smem, stream, q, ldq, k, ldk, v, ldv, o, ldo, Tq, Tk, Tcp, 0.35355339059327373f);
    }
    return cmr_launch_status();
  }
  const size_t smem = (size_t)Tk * DH * 2 * sizeof(float);
  CMR_REQUIRE(smem <= 160 * 1024);
  static CmrSmemCache granted{};
  if (cmr_grant_smem(reinterpret_cast<const void*>(mha_kernel<false>), smem, granted) != CMR_OK) return CMR_ELAUNCH;
  dim3 grid((Tq + 63) / 64, NH, B);
  hipLaunchKernelGGL(mha_kernel<false>, grid, dim3(256), smem, stream, q, ldq, k, ldk, v, ldv, o, ldo, Tq, Tk,
                     0.35355339059327373f, 0u, 1.f, (const int64_t*)nullptr, (uint64_t)0);
  return cmr_launch_status();
}

extern "C" int cmr_mha_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                           float* o, int64_t ldo, int B, int Tq, int Tk, hipStream_t stream) {
  return mha_launch(false, q, ldq, k, ldk, v, ldv, o, ldo, B, Tq, Tk, stream);
}

extern "C" int cmr_mha_expf_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                                float* o, int64_t ldo, int B, int Tq, int Tk, hipStream_t stream) {
  return mha_launch(true, q, ldq, k, ldk, v, ldv, o, ldo, B, Tq, Tk, stream);
}

extern "C" int cmr_mha_ln_f32(const float* x, int64_t ldx, const float* y, int64_t ldy, const float* gamma, const float* beta, float eps,
                              const float* wq_frag, const float* wkv_frag, const float* bq, const float* bk, const float* bv, float* o,
                              int64_t ldo, int B, int Tq, int Tk, hipStream_t stream) {
  CMR_REQUIRE(x && y && gamma && beta && wq_frag && wkv_frag && bq && bk && bv && o && B > 0 && B <= 65535 && Tq > 0 && Tk > 0);
  CMR_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0 && ldo % 4 == 0 && cmr_aligned16(x) && cmr_aligned16(y) && cmr_aligned16(o) &&
              cmr_aligned16(gamma) && cmr_aligned16(beta));
  const int Tkp = (Tk + 31) / 32 * 32;
  int Tcp = Tkp < CMR_MHA_KEY_CHUNK ? Tkp : CMR_MHA_KEY_CHUNK;
  Tcp = (Tcp + 63) / 64 * 64;                          // whole 16-row tiles for each of the four staging waves
  const size_t smem = (size_t)Tcp * (DH + 9) * sizeof(float);
  CMR_REQUIRE(smem <= 160 * 1024);
  static CmrSmemCache granted{};
  if (cmr_grant_smem(reinterpret_cast<const void*>(mha_ln_kernel), smem, granted) != CMR_OK) return CMR_ELAUNCH;
  const MhaLnArgs a{x, ldx, y, ldy, gamma, beta, eps, wq_frag, wkv_frag, bq, bk, bv, o, ldo, Tq, Tk, Tcp, 0.35355339059327373f};
  hipLaunchKernelGGL(mha_ln_kernel, dim3((Tq + 63) / 64, NH, B), dim3(256), smem, stream, a);
  return cmr_launch_status();
}

extern "C" int cmr_mha_dropout_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, float* o,
                                   int64_t ldo, int B, int Tq, int Tk, float p, const int64_t* seed, int64_t site, hipStream_t stream) {
  CMR_REQUIRE(q && k && v && o && seed && B > 0 && B <= 65535 && Tq > 0 && Tk > 0 && p >= 0.f && p < 1.f);
  CMR_REQUIRE(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && ldo % 4 == 0);
  CMR_REQUIRE(cmr_aligned16(q) && cmr_aligned16(k) && cmr_aligned16(v) && cmr_aligned16(o));
  const size_t smem = (size_t)Tk * DH * 2 * sizeof(float);
  CMR_REQUIRE(smem <= 160 * 1024);
  static CmrSmemCache granted{};
  if (cmr_grant_smem(reinterpret_cast<const void*>(mha_kernel<true>), smem, granted) != CMR_OK) return CMR_ELAUNCH;
  dim3 grid((Tq + 63) / 64, NH, B);
  hipLaunchKernelGGL(mha_kernel<true>, grid, dim3(256), smem, stream, q, ldq, k, ldk, v, ldv, o, ldo, Tq, Tk, 0.35355339059327373f,
                     cmr_drop_threshold(p), 1.f / (1.f - p), seed, (uint64_t)site);
  return cmr_launch_status();
}

extern "C" int64_t cmr_la_reduce_workspace_bytes(int B, int S) {
  const int slab_tokens = 512;
  const int nslab = (S + slab_tokens - 1) / slab_tokens;
  return (int64_t)B * nslab * 576 * sizeof(float);
}

extern "C" int cmr_la_reduce_f32(const float* kf, int64_t ldk, const float* v, int64_t ldv, float* kvsum,
                                 void* workspace, int64_t workspace_bytes, int B, int S, hipStream_t stream) {
  CMR_REQUIRE(kf && v && kvsum && workspace && B > 0 && B <= 65535 && S > 0);
  CMR_REQUIRE(ldk % 4 == 0 && ldv % 4 == 0 && cmr_aligned16(kf) && cmr_aligned16(v));
  CMR_REQUIRE(workspace_bytes >= cmr_la_reduce_workspace_bytes(B, S));
  const int slab_tokens = 512;
  const int nslab = (S + slab_tokens - 1) / slab_tokens;
  hipLaunchKernelGGL(la_reduce_partial_kernel, dim3(nslab, B), dim3(512), 0, stream, kf, ldk, v, ldv,
                     (float*)workspace, S, slab_tokens);
  hipLaunchKernelGGL(la_reduce_final_kernel, dim3(B), dim3(576), 0, stream, (const float*)workspace, kvsum, nslab);
  return cmr_launch_status();
}

extern "C" int cmr_la_apply_f32(const float* qf, int64_t ldq, const float* kvsum, float* msg, int64_t ldm, int B, int L,
                                int S, float eps, hipStream_t stream) {
  CMR_REQUIRE(qf && kvsum && msg && B > 0 && B <= 65535 && L > 0 && S > 0);
  const int tokens_per_wave = 16;
  dim3 grid((L + 4 * tokens_per_wave - 1) / (4 * tokens_per_wave), B);
  hipLaunchKernelGGL(la_apply_kernel, grid, dim3(256), 0, stream, qf, ldq, kvsum, msg, ldm, L, S, eps, tokens_per_wave);
  return cmr_launch_status();
}
