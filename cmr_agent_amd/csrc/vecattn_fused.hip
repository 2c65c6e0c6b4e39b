// Front half of the vector attention of both point transformers (reference models/PointNN.py:151-170, 219-226) in ONE
// kernel per layer: everything that is computed per (point, owning node) / (node, neighbour) row --
//     x = fc1_0(feat) ; k = Wk x ; v = Wv x                       (group transformer: from the point's features)
//     | k, v = gathered rows of a precomputed [k|v] table          (kNN transformer: the neighbour's)
//     pos = fc_delta(p_a - p_b)                                    (3 -> 64 ReLU -> 64)
//     a   = fc_gamma(q[owner] - k + pos)                           (64 -> 64 ReLU -> 64)
//     vp  = v + pos
// -- leaving only the per-segment softmax / weighted sum (cmr_segment_softmax_f32) after it.  The unfused path is seven
// streaming launches that each write and re-read a [rows, 64] array (1.5 K floats of HBM traffic per row against 192
// here) and are launch-latency bound at the 131 072-row size of this workload.  GEMMs are chained through accumulator
// registers (cmr_chain.h); all weights sit in LDS; waves stream 32-row tiles independently.
#include "cmr_chain.h"

namespace {

constexpr int VA_LD = 68, VA_LD8 = 12;      // padded LDS rows for K = 64 and K = 8

struct VaArgs {
  const float* feat; int64_t ldf; const float *w10, *b10, *wkv;          // computed k/v (feat != null)
  const float* kv; int64_t ldkv; const int32_t* ik;                      // gathered k/v (feat == null): k at column 0, v at 64
  const float* q; int64_t ldq; const int32_t* iq; uint32_t divq;
  const float* pa4; const int32_t* ia; uint32_t diva; const float* pb4; const int32_t* ib;
  const float *wd0, *bd0, *wd2, *bd2, *wg0, *bg0, *wg2, *bg2;
  float* a_out; float* vp_out; uint32_t rows;
  // training (TRAIN; rows a multiple of 32): v from its own map instead of kv + 64, and the three activations the layer-by-layer
  // backward reads -- relu(fc_delta[0]), q - k + pos, relu(fc_gamma[0]) -- stored beside the outputs
  const float* v2; int64_t ldv2;
  float *hd_out, *t_out, *g1_out;
  float* x_out;                                  // TRAIN + computed k/v: x = fc1_0(feat) [rows][64], what the k / v projections' backward reads
  const float* wv2;                              // Wv [64][64] when it does not follow Wk in memory (training: two parameters), else null
};

__device__ __attribute__((aligned(16))) int32_t va_izero[4] = {0, 0, 0, 0};   // NOT const (see cmr_common.h: cmr_pin)

template <bool COMPUTE_KV, bool TRAIN = false>
__global__ __launch_bounds__(512) void vecattn_front_kernel(const VaArgs a) {

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Wd0 = smem;                          // [64][12]  (K = 3 padded to 8)
  float* Wd2 = Wd0 + 64 * VA_LD8;             // [64][68]
  float* Wg0 = Wd2 + 64 * VA_LD;
  float* Wg2 = Wg0 + 64 * VA_LD;
  float* Bs = Wg2 + 64 * VA_LD;               // b10 | bd0 | bd2 | bg0 | bg2
  float* W10 = Bs + 5 * 64;                   // [64][68]   (COMPUTE_KV only)
  float* Wkv = W10 + 64 * VA_LD;              // [128][68]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  for (int e = tid; e < 64 * 2; e += 512) {   // Wd0: [64][4] (3 real columns, _pack.lin pads K to 4) -> [64][8], upper half zero
    const int n = e >> 1, half = e & 1;
    *reinterpret_cast<f32x4*>(&Wd0[n * VA_LD8 + 4 * half]) = half ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(a.wd0 + n * 4);
  }
  for (int e = tid; e < 64 * 16; e += 512) {
    const int n = e >> 4, c = (e & 15) * 4;
    *reinterpret_cast<f32x4*>(&Wd2[n * VA_LD + c]) = *reinterpret_cast<const f32x4*>(a.wd2 + n * 64 + c);
    *reinterpret_cast<f32x4*>(&Wg0[n * VA_LD + c]) = *reinterpret_cast<const f32x4*>(a.wg0 + n * 64 + c);
    *reinterpret_cast<f32x4*>(&Wg2[n * VA_LD + c]) = *reinterpret_cast<const f32x4*>(a.wg2 + n * 64 + c);
    if (COMPUTE_KV) *reinterpret_cast<f32x4*>(&W10[n * VA_LD + c]) = *reinterpret_cast<const f32x4*>(a.w10 + n * 64 + c);
  }
  if (COMPUTE_KV)
    for (int e = tid; e < 128 * 16; e += 512) {
      const int n = e >> 4, c = (e & 15) * 4;
      const float* src = (n >= 64 && a.wv2) ? a.wv2 + (n - 64) * 64 + c : a.wkv + n * 64 + c;
      *reinterpret_cast<f32x4*>(&Wkv[n * VA_LD + c]) = *reinterpret_cast<const f32x4*>(src);
    }
  if (tid < 64) {
    Bs[tid] = COMPUTE_KV ? a.b10[tid] : 0.f;
    Bs[64 + tid] = a.bd0[tid]; Bs[128 + tid] = a.bd2[tid]; Bs[192 + tid] = a.bg0[tid]; Bs[256 + tid] = a.bg2[tid];
  }
  __syncthreads();

  const uint32_t ntiles = (a.rows + 31) / 32;
  for (uint32_t tile = blockIdx.x * 8 + wave; tile < ntiles; tile += gridDim.x * 8) {
    const uint32_t row = tile * 32 + l31;
    const bool valid = row < a.rows;
    const uint32_t r = valid ? row : 0;          // rows past the end recompute row 0 and are not stored
    // ---- row maps (one batch of index loads; absent maps read a dummy word)
    const int32_t vq = (a.iq ? a.iq : va_izero)[a.iq ? r : 0];
    const int32_t va = (a.ia ? a.ia : va_izero)[a.ia ? r : 0];
    const int32_t vb = a.ib[r];
    const int32_t vk = (a.ik ? a.ik : va_izero)[a.ik ? r : 0];
    const uint32_t rq = a.iq ? (uint32_t)vq : r / a.divq;
    const uint32_t ra = a.ia ? (uint32_t)va : r / a.diva;
    // ---- gathers: q fragments, the two positions, k/v source rows
    const float* qp = a.q + (int64_t)rq * a.ldq + 4 * h;
    f32x4 qf[8];
#pragma unroll
    for (int kg = 0; kg < 8; ++kg) qf[kg] = *reinterpret_cast<const f32x4*>(qp + kg * 8);
    const f32x4 pa = *reinterpret_cast<const f32x4*>(a.pa4 + (int64_t)ra * 4);
    const f32x4 pb = *reinterpret_cast<const f32x4*>(a.pb4 + (int64_t)(uint32_t)vb * 4);
    f32x16 kk[2], vv[2];
    if (COMPUTE_KV) {
      const float* fp = a.feat + (int64_t)r * a.ldf + 4 * h;
      f32x4 ff[8];
#pragma unroll
      for (int kg = 0; kg < 8; ++kg) ff[kg] = *reinterpret_cast<const f32x4*>(fp + kg * 8);
      f32x16 x[2];
      cmr_chain_gemm<2, 8, VA_LD>(W10, l31, h, x, [&](int kg, int j) { return ff[kg][j]; });
      cmr_chain_bias<2>(x, Bs, h, false);
      if (TRAIN) {
        float* xp = a.x_out + (int64_t)row * 64 + 4 * h;
#pragma unroll
        for (int kg = 0; kg < 8; ++kg) {
          f32x4 sx;
#pragma unroll
          for (int e = 0; e < 4; ++e) sx[e] = x[kg / 4][4 * (kg % 4) + e];
          cmr_pin(sx);
          *reinterpret_cast<f32x4*>(xp + kg * 8) = sx;
        }
      }
      cmr_chain_gemm<2, 8, VA_LD>(Wkv, l31, h, kk, [&](int kg, int j) { return x[kg / 4][4 * (kg % 4) + j]; });
      cmr_chain_gemm<2, 8, VA_LD>(Wkv + 64 * VA_LD, l31, h, vv, [&](int kg, int j) { return x[kg / 4][4 * (kg % 4) + j]; });
    } else {
      const float* kp = a.kv + (int64_t)(a.ik ? (uint32_t)vk : r) * a.ldkv + 4 * h;
#pragma unroll
      for (int kg = 0; kg < 8; ++kg) {
        const f32x4 k4 = *reinterpret_cast<const f32x4*>(kp + kg * 8);
        const f32x4 v4 = (TRAIN && !COMPUTE_KV) ? *reinterpret_cast<const f32x4*>(a.v2 + (int64_t)(a.ik ? (uint32_t)vk : r) * a.ldv2 + 4 * h + kg * 8)
                                                : *reinterpret_cast<const f32x4*>(kp + 64 + kg * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) { kk[kg / 4][4 * (kg % 4) + e] = k4[e]; vv[kg / 4][4 * (kg % 4) + e] = v4[e]; }
      }
    }
    // ---- pos = fc_delta(pa - pb): the 3 coordinates are the k = 0..2 entries of one k-group (lane half 0)
    const f32x4 rel = {pa[0] - pb[0], pa[1] - pb[1], pa[2] - pb[2], 0.f};
    f32x16 hd[2], pos[2];
    cmr_chain_gemm<2, 1, VA_LD8>(Wd0, l31, h, hd, [&](int, int j) { return h == 0 ? rel[j] : 0.f; });
    cmr_chain_bias<2>(hd, Bs + 64, h, true);
    cmr_chain_gemm<2, 8, VA_LD>(Wd2, l31, h, pos, [&](int kg, int j) { return hd[kg / 4][4 * (kg % 4) + j]; });
    cmr_chain_bias<2>(pos, Bs + 128, h, false);
    // ---- a = fc_gamma(q - k + pos) ; vp = v + pos
    f32x16 t[2];
#pragma unroll
    for (int kg = 0; kg < 8; ++kg)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int ti = kg / 4, ri = 4 * (kg % 4) + e;
        t[ti][ri] = qf[kg][e] - kk[ti][ri] + pos[ti][ri];
        vv[ti][ri] = vv[ti][ri] + pos[ti][ri];
      }
    f32x16 g1[2], ao[2];
    cmr_chain_gemm<2, 8, VA_LD>(Wg0, l31, h, g1, [&](int kg, int j) { return t[kg / 4][4 * (kg % 4) + j]; });
    cmr_chain_bias<2>(g1, Bs + 192, h, true);
    cmr_chain_gemm<2, 8, VA_LD>(Wg2, l31, h, ao, [&](int kg, int j) { return g1[kg / 4][4 * (kg % 4) + j]; });
    cmr_chain_bias<2>(ao, Bs + 256, h, false);
    if (TRAIN) {                                 // (rows % 32 == 0: unconditional stores, the three saved activations first)
      float* hp = a.hd_out + (int64_t)row * 64 + 4 * h;
      float* tp = a.t_out + (int64_t)row * 64 + 4 * h;
      float* gp = a.g1_out + (int64_t)row * 64 + 4 * h;
#pragma unroll
      for (int kg = 0; kg < 8; ++kg) {
        f32x4 sh, st, sg;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          sh[e] = hd[kg / 4][4 * (kg % 4) + e];
          st[e] = t[kg / 4][4 * (kg % 4) + e];
          sg[e] = g1[kg / 4][4 * (kg % 4) + e];
        }
        cmr_pin(sh); cmr_pin(st); cmr_pin(sg);
        *reinterpret_cast<f32x4*>(hp + kg * 8) = sh;
        *reinterpret_cast<f32x4*>(tp + kg * 8) = st;
        *reinterpret_cast<f32x4*>(gp + kg * 8) = sg;
      }
    }
    f32x4 oa[8], ov[8];
#pragma unroll
    for (int kg = 0; kg < 8; ++kg)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        oa[kg][e] = ao[kg / 4][4 * (kg % 4) + e];
        ov[kg][e] = vv[kg / 4][4 * (kg % 4) + e];
      }
#pragma unroll
    for (int kg = 0; kg < 8; ++kg) { cmr_pin(oa[kg]); cmr_pin(ov[kg]); }
    if (valid) {
      float* ap = a.a_out + (int64_t)row * 64 + 4 * h;
      float* vp = a.vp_out + (int64_t)row * 64 + 4 * h;
#pragma unroll
      for (int kg = 0; kg < 8; ++kg) {
        *reinterpret_cast<f32x4*>(ap + kg * 8) = oa[kg];
        *reinterpret_cast<f32x4*>(vp + kg * 8) = ov[kg];
      }
    }
  }
}

template <bool COMPUTE_KV, bool TRAIN = false>
int launch_va(const VaArgs& a, hipStream_t stream) {
  const size_t smem = (size_t)(64 * VA_LD8 + 3 * 64 * VA_LD + 5 * 64 + (COMPUTE_KV ? 192 * VA_LD : 0)) * sizeof(float);
  static CmrSmemCache granted{};
  if (cmr_grant_smem(reinterpret_cast<const void*>(vecattn_front_kernel<COMPUTE_KV, TRAIN>), smem, granted) != CMR_OK) return CMR_ELAUNCH;
  const uint32_t ntiles = (a.rows + 31) / 32;
  uint32_t grid = (ntiles + 7) / 8;
  const uint32_t cap = COMPUTE_KV ? 256 : 512;          // persistent: 1 (109 KB) or 2 (57 KB) workgroups per CU
  if (grid > cap) grid = cap;
  hipLaunchKernelGGL((vecattn_front_kernel<COMPUTE_KV, TRAIN>), dim3(grid), dim3(512), smem, stream, a);
  return cmr_launch_status();
}

}  // namespace

extern "C" int cmr_vecattn_front_f32(const float* feat, int64_t ldf, const float* w10, const float* b10, const float* wkv,
                                     const float* kv, int64_t ldkv, const int32_t* ik, const float* q, int64_t ldq,
                                     const int32_t* iq, int64_t divq, const float* pa4, const int32_t* ia, int64_t diva,
                                     const float* pb4, const int32_t* ib, const float* wd0, const float* bd0, const float* wd2,
                                     const float* bd2, const float* wg0, const float* bg0, const float* wg2, const float* bg2,
                                     float* a_out, float* vp_out, int64_t rows, hipStream_t stream) {
  CMR_REQUIRE(q && pa4 && pb4 && ib && wd0 && bd0 && wd2 && bd2 && wg0 && bg0 && wg2 && bg2 && a_out && vp_out);
  CMR_REQUIRE(rows > 0 && rows < (int64_t)0x7fffffc0 && ldq % 4 == 0 && cmr_aligned16(q) && cmr_aligned16(pa4) && cmr_aligned16(pb4));
  CMR_REQUIRE((iq || divq >= 1) && (ia || diva >= 1) && cmr_aligned16(a_out) && cmr_aligned16(vp_out));
  CMR_REQUIRE(cmr_aligned16(wd0) && cmr_aligned16(wd2) && cmr_aligned16(wg0) && cmr_aligned16(wg2));
  VaArgs a{feat, ldf, w10, b10, wkv, kv, ldkv, ik, q, ldq, iq, (uint32_t)(divq < 1 ? 1 : divq), pa4, ia,
           (uint32_t)(diva < 1 ? 1 : diva), pb4, ib, wd0, bd0, wd2, bd2, wg0, bg0, wg2, bg2, a_out, vp_out, (uint32_t)rows,
           nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr};
  if (feat) {
    CMR_REQUIRE(w10 && b10 && wkv && ldf % 4 == 0 && cmr_aligned16(feat) && cmr_aligned16(w10) && cmr_aligned16(wkv));
    return launch_va<true>(a, stream);
  }
  CMR_REQUIRE(kv && ldkv % 4 == 0 && cmr_aligned16(kv));
  return launch_va<false>(a, stream);
}

// Training forward of the same front (reference under model.train(); Train_Geo.py:166-174): k and v [*][64] from their own maps, row r of
// both = ikv ? ikv[r] : r (kNN transformer: the neighbour's row of the per-node tables), and beside a_out / vp_out the three activations the layer-by-layer backward reads: hd = relu(fc_delta[0]),
// t = q - k + pos, g1 = relu(fc_gamma[0]), all [rows][64].  rows must be a multiple of 32.
extern "C" int cmr_vecattn_front_train_f32(const float* k, int64_t ldk, const float* v, int64_t ldv, const int32_t* ikv, const float* q, int64_t ldq,
                                           const int32_t* iq, int64_t divq, const float* pa4, const int32_t* ia, int64_t diva, const float* pb4,
                                           const int32_t* ib, const float* wd0, const float* bd0, const float* wd2, const float* bd2,
                                           const float* wg0, const float* bg0, const float* wg2, const float* bg2, float* a_out,
                                           float* vp_out, float* hd_out, float* t_out, float* g1_out, int64_t rows, hipStream_t stream) {
  CMR_REQUIRE(k && v && q && pa4 && pb4 && ib && wd0 && bd0 && wd2 && bd2 && wg0 && bg0 && wg2 && bg2 && a_out && vp_out && hd_out && t_out && g1_out);
  if (rows <= 0 || rows % 32 || rows >= (int64_t)0x7fffffc0) return CMR_EUNSUPPORTED;
  CMR_REQUIRE(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && cmr_aligned16(q) && cmr_aligned16(k) && cmr_aligned16(v) && cmr_aligned16(pa4) &&
              cmr_aligned16(pb4));
  CMR_REQUIRE((iq || divq >= 1) && (ia || diva >= 1) && cmr_aligned16(a_out) && cmr_aligned16(vp_out) && cmr_aligned16(hd_out) &&
              cmr_aligned16(t_out) && cmr_aligned16(g1_out));
  CMR_REQUIRE(cmr_aligned16(wd0) && cmr_aligned16(wd2) && cmr_aligned16(wg0) && cmr_aligned16(wg2));
  VaArgs a{nullptr, 0, nullptr, nullptr, nullptr, k, ldk, ikv, q, ldq, iq, (uint32_t)(divq < 1 ? 1 : divq), pa4, ia,
           (uint32_t)(diva < 1 ? 1 : diva), pb4, ib, wd0, bd0, wd2, bd2, wg0, bg0, wg2, bg2, a_out, vp_out, (uint32_t)rows,
           v, ldv, hd_out, t_out, g1_out, nullptr, nullptr};
  return launch_va<false, true>(a, stream);
}

// ... with k and v computed inside as well (group transformer: x = W10 feat + b10, k = Wk x, v = Wv x; Wk, Wv [64][64]): x is
// stored for the projections' backward, k and v never leave the registers.
extern "C" int cmr_vecattn_front_kv_train_f32(const float* feat, int64_t ldf, const float* w10, const float* b10, const float* wk,
                                              const float* wv, const float* q, int64_t ldq, const int32_t* iq, int64_t divq, const float* pa4,
                                              const int32_t* ia, int64_t diva, const float* pb4, const int32_t* ib, const float* wd0,
                                              const float* bd0, const float* wd2, const float* bd2, const float* wg0, const float* bg0,
                                              const float* wg2, const float* bg2, float* a_out, float* vp_out, float* hd_out, float* t_out,
                                              float* g1_out, float* x_out, int64_t rows, hipStream_t stream) {
  CMR_REQUIRE(feat && w10 && b10 && wk && wv && q && pa4 && pb4 && ib && wd0 && bd0 && wd2 && bd2 && wg0 && bg0 && wg2 && bg2);
  CMR_REQUIRE(a_out && vp_out && hd_out && t_out && g1_out && x_out);
  if (rows <= 0 || rows % 32 || rows >= (int64_t)0x7fffffc0) return CMR_EUNSUPPORTED;
  CMR_REQUIRE(ldq % 4 == 0 && ldf % 4 == 0 && cmr_aligned16(q) && cmr_aligned16(feat) && cmr_aligned16(pa4) && cmr_aligned16(pb4));
  CMR_REQUIRE((iq || divq >= 1) && (ia || diva >= 1) && cmr_aligned16(a_out) && cmr_aligned16(vp_out) && cmr_aligned16(hd_out) &&
              cmr_aligned16(t_out) && cmr_aligned16(g1_out) && cmr_aligned16(x_out));
  CMR_REQUIRE(cmr_aligned16(wd0) && cmr_aligned16(wd2) && cmr_aligned16(wg0) && cmr_aligned16(wg2) && cmr_aligned16(w10) && cmr_aligned16(wk) && cmr_aligned16(wv));
  VaArgs a{feat, ldf, w10, b10, wk, nullptr, 0, nullptr, q, ldq, iq, (uint32_t)(divq < 1 ? 1 : divq), pa4, ia,
           (uint32_t)(diva < 1 ? 1 : diva), pb4, ib, wd0, bd0, wd2, bd2, wg0, bg0, wg2, bg2, a_out, vp_out, (uint32_t)rows,
           nullptr, 0, hd_out, t_out, g1_out, x_out, wv};
  return launch_va<true, true>(a, stream);
}
