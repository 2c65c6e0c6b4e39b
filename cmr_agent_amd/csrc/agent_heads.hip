// Tail of the CMRAgent forward (reference models/CMRAgent.py:52-56,59-86,101-116) in ONE launch:
//   global average pool of the last 2-D feature map -> conv1x1 + LeakyReLU -> conv1x1        (embed_2d, 128)
//   cat([embed_2d, embed_3d])                                                                (state, 256)
//   policy_r / policy_t / value: three 3-layer MLPs with LeakyReLU                           (logits, value)
// All of it is 8-row (one row per sample) matrix-vector work: as thirteen separate launches it is pure launch
// latency (~90 us per agent step).  One workgroup per (sample, head); a wave produces one output channel at a time
// (lanes split the input vector in float4 pieces, one coalesced row read, xor-shuffle reduction).
#include "cmr_common.h"

namespace {

constexpr int AH_THREADS = 1024, AH_WAVES = AH_THREADS / 64;
constexpr int AH_C = 128, AH_STATE = 256, AH_MAXW = 256;      // 2-D channels, state width, widest hidden layer

struct AhHead { const float *w0, *b0, *w1, *b1, *w2, *b2; int n0, n1, n2; float* out; int ldo; int64_t* act; int degree; };
struct AhArgs {
  const float* x; int npix;              // [B][npix][128] last feature map (already activated)
  const float *w24, *b24, *w26, *b26;    // conv1x1 128 -> 128 (LeakyReLU), conv1x1 128 -> 128
  const float* e3d;                      // [B][128]
  AhHead h[3];
  float slope;
  int num_steps;                         // > 0: heads 0 / 1 also emit argmax over each group of num_steps logits (CMRAgent.py:118-123)
  // training forward (cmr_agent_heads_train_f32): every intermediate the backward needs, in one buffer --
  // pooled [B][128] | t1 = lrelu(conv 24) [B][128] | e2d = conv 26 [B][128] | per head: h0 [B][n0], h1 [B][n1] (post-activation)
  float* saves;
};

// y[n] = act(b[n] + sum_k W[n][k] x[k]) for the outputs n = wave, wave + 16, ...; x in LDS, k % 4 == 0, k <= 256.
// Eight weight rows per wave are in flight at once (branch-free: out-of-range rows / columns re-read a valid
// element, lanes past k multiply by the zero x they hold), so a layer costs about one memory round trip.
__device__ __forceinline__ void ah_gemv(const float* __restrict__ W, const float* __restrict__ bias, const float* xin, int k, int n_out,
                                        float* yout, bool lrelu, float slope, int wave, int lane) {
  constexpr int R = 8;
  const int c = lane * 4;
  const int cc = c < k ? c : 0;
  f32x4 xv = *reinterpret_cast<const f32x4*>(xin + cc);
  if (c >= k) xv = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int n0 = wave; n0 < n_out; n0 += R * AH_WAVES) {
    f32x4 wv[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const int n = n0 + i * AH_WAVES;
      wv[i] = *reinterpret_cast<const f32x4*>(W + (int64_t)(n < n_out ? n : n_out - 1) * k + cc);
    }
    float s[R];
#pragma unroll
    for (int i = 0; i < R; ++i) s[i] = (wv[i][0] * xv[0] + wv[i][1] * xv[1]) + (wv[i][2] * xv[2] + wv[i][3] * xv[3]);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
#pragma unroll
      for (int i = 0; i < R; ++i) s[i] += __shfl_xor(s[i], m);
#pragma unroll
    for (int i = 0; i < R; ++i) {
      const int n = n0 + i * AH_WAVES;
      if (lane == 0 && n < n_out) {
        const float v = s[i] + bias[n];
        yout[n] = lrelu ? (v > 0.f ? v : v * slope) : v;
      }
    }
  }
}

__global__ __launch_bounds__(AH_THREADS) void agent_heads_kernel(const AhArgs a) {
  __shared__ __attribute__((aligned(16))) float red[8][AH_C];
  __shared__ __attribute__((aligned(16))) float va[AH_C], vb[AH_C], state[AH_STATE];
  __shared__ __attribute__((aligned(16))) float h0[1][AH_MAXW], h1[1][AH_MAXW];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // ---- AvgPool2d((H, W)): per-sample channel mean, 8 pixel groups x 128 channels, fixed summation order
  {
    const int c = tid & (AH_C - 1), g = tid >> 7;
    const float* xp = a.x + (int64_t)b * a.npix * AH_C + c;
    float s = 0.f;
    int p = g;
    for (; p + 120 < a.npix; p += 128) {                      // 16 loads in flight per thread
      float v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = xp[(int64_t)(p + 8 * i) * AH_C];
#pragma unroll
      for (int i = 0; i < 16; ++i) s += v[i];
    }
    for (; p + 56 < a.npix; p += 64) {                        // 8 loads in flight per thread
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = xp[(int64_t)(p + 8 * i) * AH_C];
#pragma unroll
      for (int i = 0; i < 8; ++i) s += v[i];
    }
    for (; p < a.npix; p += 8) s += xp[(int64_t)p * AH_C];
    red[g][c] = s;
    __syncthreads();
    if (tid < AH_C) {
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) t += red[i][tid];
      va[tid] = t / (float)a.npix;
    }
    if (tid >= AH_C && tid < 2 * AH_C) state[tid] = a.e3d[(int64_t)b * AH_C + tid - AH_C];
    __syncthreads();
  }
  const int B = gridDim.x;
  const bool save = a.saves != nullptr;                         // (uniform)
  const bool save0 = save && blockIdx.y == 0 && tid < AH_C;     // the shared part is written by the workgroup of head 0
  if (save0) a.saves[(int64_t)b * AH_C + tid] = va[tid];
  ah_gemv(a.w24, a.b24, va, AH_C, AH_C, vb, true, a.slope, wave, lane);
  __syncthreads();
  if (save0) a.saves[((int64_t)B + b) * AH_C + tid] = vb[tid];
  ah_gemv(a.w26, a.b26, vb, AH_C, AH_C, state, false, a.slope, wave, lane);
  __syncthreads();
  if (save0) a.saves[((int64_t)2 * B + b) * AH_C + tid] = state[tid];
  // ---- ONE head per workgroup (blockIdx.y): the pool and the two 1x1 convs above are recomputed by the three workgroups
  // of a sample (345 KB of reads), which is cheaper than streaming the three heads' 1.2 MB through one CU layer by layer
  const AhHead& hd = a.h[blockIdx.y];
  int64_t soff = (int64_t)3 * B * AH_C;                        // this head's h0 / h1 block in the saves
  for (int i = 0; i < (int)blockIdx.y; ++i) soff += (int64_t)B * (a.h[i].n0 + a.h[i].n1);
  ah_gemv(hd.w0, hd.b0, state, AH_STATE, hd.n0, h0[0], true, a.slope, wave, lane);
  __syncthreads();
  if (save && tid < hd.n0) a.saves[soff + (int64_t)b * hd.n0 + tid] = h0[0][tid];
  ah_gemv(hd.w1, hd.b1, h0[0], hd.n0, hd.n1, h1[0], true, a.slope, wave, lane);
  __syncthreads();
  if (save && tid < hd.n1) a.saves[soff + (int64_t)B * hd.n0 + (int64_t)b * hd.n1 + tid] = h1[0][tid];
  ah_gemv(hd.w2, hd.b2, h1[0], hd.n1, hd.n2, hd.out + (int64_t)b * hd.ldo, false, a.slope, wave, lane);
  // deterministic action of this head: argmax of Categorical(logits).probs = argmax of the logits, first maximum on ties (what
  // cmr_argmax_rows_f32 returns for the same rows) -- saves the two argmax launches of every agent step
  if (hd.act != nullptr && a.num_steps > 0) {
    __syncthreads();                                          // the logits above were written by lane 0 of several waves
    const int d = hd.degree;
    if (tid < d) {
      const float* p = hd.out + (int64_t)b * hd.ldo + tid * a.num_steps;
      float best = p[0];
      int bi = 0;
      for (int i = 1; i < a.num_steps; ++i) {
        const float v = p[i];
        if (v > best) { best = v; bi = i; }
      }
      hd.act[(int64_t)b * d + tid] = bi;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// The same tail with the weights stored TRANSPOSED, [in][out4] (round 5).  In the kernel above a wave produces output channels one
// weight row at a time and reduces each dot product over its 64 lanes (six xor-shuffle steps per row: 96 cross-lane operations per batch
// of 16 rows, two batches per 256-wide layer) -- 35 us per agent step on the serial chain of a registration for 1.3 MFLOP.  Here lane l of
// every wave owns the outputs 4 l .. 4 l + 3 and wave w the input slice [w k / 16, (w + 1) k / 16): each lane issues ALL its weight loads
// (k / 16 <= 16 float4, coalesced over the outputs: a wave reads whole 1 KB rows of W^T) before the first multiply -- one memory round trip
// per layer, no cross-lane operation at all -- and the 16 slices meet in LDS (16 x 256 partial sums, one add chain per output in a fixed
// order).  The global average pool reads its sample as float4 channel quads, 13 - 14 pixels per thread, all in flight.
// ------------------------------------------------------------------------------------------------------------------
struct AtHead { const float *w0t, *b0, *w1t, *b1, *w2t, *b2; int n0, n1, n2, ld2; float* out; int ldo; int64_t* act; int degree; };
struct AtArgs {
  const float* x; int npix;
  const float *w24t, *b24, *w26t, *b26;
  const float* e3d;
  AtHead h[3];
  float slope;
  int num_steps;
};

// One layer y[0 .. n_out) = act(b + x W^T), wt = W^T stored [k][ldw] (ldw % 4 == 0, zero padded columns), x in LDS, k % 16 == 0, k <= 256,
// n_out <= 256, in two halves so that the NEXT layer's weight loads fly while this one's partial sums meet in LDS:
//   at_issue: this lane's <= 16 float4 of its wave's input slice -> registers (one memory round trip per layer, nothing waits for it here)
//   at_apply: multiply with the loaded slice, partial sums -> LDS scratch [16][256], THEN `next()` (the following layer's at_issue: the
//             registers are free again), barrier, one add chain per output in wave order, activation, barrier.
__device__ __forceinline__ void at_issue(const float* __restrict__ wt, int ldw, int k, f32x4 (&wv)[16], int wave, int lane) {
  const int ks = k >> 4;                               // input slice of this wave: ks <= 16 rows of W^T
  const int c = 4 * lane;
  const bool on = c < ldw;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int kk = wave * ks + (i < ks ? i : ks - 1);
    wv[i] = *reinterpret_cast<const f32x4*>(wt + (int64_t)kk * ldw + (on ? c : 0));
  }
}

template <typename Next>
__device__ __forceinline__ void at_apply(f32x4 (&wv)[16], const float* __restrict__ bias, const float* xin, int k, int n_out, float* yout, bool lrelu,
                                         float slope, float (*part)[256], int wave, int lane, Next next) {
  const int ks = k >> 4;
  const int c = 4 * lane;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const float xv = i < ks ? xin[wave * ks + i] : 0.f;     // LDS broadcast
    s += wv[i] * xv;
  }
  *reinterpret_cast<f32x4*>(&part[wave][c]) = s;
  next();
  __syncthreads();
  const int tid = wave * 64 + lane;
  if (tid < n_out) {
    float t = part[0][tid];
#pragma unroll
    for (int w = 1; w < 16; ++w) t += part[w][tid];
    t += bias[tid];
    yout[tid] = lrelu ? (t > 0.f ? t : t * slope) : t;
  }
  __syncthreads();
}

__global__ __launch_bounds__(1024) void agent_heads_t_kernel(const AtArgs a) {
  __shared__ __attribute__((aligned(16))) float part[16][256];
  __shared__ __attribute__((aligned(16))) float va[AH_C], vb[AH_C], state[AH_STATE], h0[AH_MAXW], h1[AH_MAXW];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const AtHead& hd = a.h[blockIdx.y];
  f32x4 wv[16];
  at_issue(a.w24t, AH_C, AH_C, wv, wave, lane);          // the first layer's weights are on their way while the map is pooled
  // ---- AvgPool2d((H, W)): thread = (channel quad q of 32, pixel group g of 32); 8 loads of the thread in flight at once
  {
    const int q = tid & 31, g = tid >> 5;
    const float* xp = a.x + (int64_t)b * a.npix * AH_C + 4 * q;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int p0 = g; p0 < a.npix; p0 += 32 * 8) {
      f32x4 v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int p = p0 + 32 * i;
        v[i] = *reinterpret_cast<const f32x4*>(xp + (int64_t)(p < a.npix ? p : p0) * AH_C);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) s += p0 + 32 * i < a.npix ? v[i] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // 32 pixel groups x 128 channels of partial sums through the [16][256] scratch viewed as [32][128]
    float* red = &part[0][0];
    *reinterpret_cast<f32x4*>(red + g * AH_C + 4 * q) = s;
    __syncthreads();
    if (tid < AH_C) {
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < 32; ++i) t += red[i * AH_C + tid];
      va[tid] = t / (float)a.npix;
    }
    if (tid >= AH_C && tid < 2 * AH_C) state[tid] = a.e3d[(int64_t)b * AH_C + tid - AH_C];
    __syncthreads();
  }
  at_apply(wv, a.b24, va, AH_C, AH_C, vb, true, a.slope, part, wave, lane, [&] { at_issue(a.w26t, AH_C, AH_C, wv, wave, lane); });
  at_apply(wv, a.b26, vb, AH_C, AH_C, state, false, a.slope, part, wave, lane, [&] { at_issue(hd.w0t, hd.n0, AH_STATE, wv, wave, lane); });
  at_apply(wv, hd.b0, state, AH_STATE, hd.n0, h0, true, a.slope, part, wave, lane, [&] { at_issue(hd.w1t, hd.n1, hd.n0, wv, wave, lane); });
  at_apply(wv, hd.b1, h0, hd.n0, hd.n1, h1, true, a.slope, part, wave, lane, [&] { at_issue(hd.w2t, hd.ld2, hd.n1, wv, wave, lane); });
  at_apply(wv, hd.b2, h1, hd.n1, hd.n2, hd.out + (int64_t)b * hd.ldo, false, a.slope, part, wave, lane, [] {});
  if (hd.act != nullptr && a.num_steps > 0) {                 // (the logits were written before at_apply's closing barrier)
    const int d = hd.degree;
    if (tid < d) {
      const float* p = hd.out + (int64_t)b * hd.ldo + tid * a.num_steps;
      float best = p[0];
      int bi = 0;
      for (int i = 1; i < a.num_steps; ++i) {
        const float v = p[i];
        if (v > best) { best = v; bi = i; }
      }
      hd.act[(int64_t)b * d + tid] = bi;
    }
  }
}

}  // namespace

extern "C" int cmr_agent_heads_t_f32(const float* x, int B, int npix, const float* w24t, const float* b24, const float* w26t, const float* b26,
                                     const float* e3d,
                                     const float* r_w0t, const float* r_b0, const float* r_w1t, const float* r_b1, const float* r_w2t, const float* r_b2,
                                     int r_n0, int r_n1, int r_n2, float* r_out, int r_ldo,
                                     const float* t_w0t, const float* t_b0, const float* t_w1t, const float* t_b1, const float* t_w2t, const float* t_b2,
                                     int t_n0, int t_n1, int t_n2, float* t_out, int t_ldo,
                                     const float* v_w0t, const float* v_b0, const float* v_w1t, const float* v_b1, const float* v_w2t, const float* v_b2,
                                     int v_n0, int v_n1, int v_n2, float* v_out, int v_ldo,
                                     int num_steps, int degree_r, int degree_t, int64_t* r_act, int64_t* t_act, float slope, hipStream_t stream) {
  CMR_REQUIRE(x && w24t && b24 && w26t && b26 && e3d && B > 0 && npix > 0);
  CMR_REQUIRE(cmr_aligned16(x) && cmr_aligned16(w24t) && cmr_aligned16(w26t));
  AtArgs a{};
  a.x = x; a.npix = npix; a.w24t = w24t; a.b24 = b24; a.w26t = w26t; a.b26 = b26; a.e3d = e3d; a.slope = slope;
  CMR_REQUIRE((!r_act && !t_act) || (num_steps > 0 && degree_r > 0 && degree_t > 0 && degree_r * num_steps <= r_n2 && degree_t * num_steps <= t_n2));
  a.num_steps = (r_act || t_act) ? num_steps : 0;
  a.h[0] = AtHead{r_w0t, r_b0, r_w1t, r_b1, r_w2t, r_b2, r_n0, r_n1, r_n2, (r_n2 + 3) / 4 * 4, r_out, r_ldo, r_act, degree_r};
  a.h[1] = AtHead{t_w0t, t_b0, t_w1t, t_b1, t_w2t, t_b2, t_n0, t_n1, t_n2, (t_n2 + 3) / 4 * 4, t_out, t_ldo, t_act, degree_t};
  a.h[2] = AtHead{v_w0t, v_b0, v_w1t, v_b1, v_w2t, v_b2, v_n0, v_n1, v_n2, (v_n2 + 3) / 4 * 4, v_out, v_ldo, nullptr, 0};
  for (int i = 0; i < 3; ++i) {
    const AtHead& h = a.h[i];
    CMR_REQUIRE(h.w0t && h.w1t && h.w2t && h.b0 && h.b1 && h.b2 && h.out && h.n2 > 0 && h.n2 <= 256 && h.ldo >= h.n2);
    CMR_REQUIRE(h.n0 > 0 && h.n0 <= AH_MAXW && h.n0 % 16 == 0 && h.n1 > 0 && h.n1 <= AH_MAXW && h.n1 % 16 == 0);
    CMR_REQUIRE(cmr_aligned16(h.w0t) && cmr_aligned16(h.w1t) && cmr_aligned16(h.w2t));
  }
  hipLaunchKernelGGL(agent_heads_t_kernel, dim3(B, 3), dim3(AH_THREADS), 0, stream, a);
  return cmr_launch_status();
}

extern "C" int cmr_agent_heads_f32(const float* x, int B, int npix, const float* w24, const float* b24, const float* w26,
                                   const float* b26, const float* e3d,
                                   const float* r_w0, const float* r_b0, const float* r_w1, const float* r_b1, const float* r_w2, const float* r_b2,
                                   int r_n0, int r_n1, int r_n2, float* r_out, int r_ldo,
                                   const float* t_w0, const float* t_b0, const float* t_w1, const float* t_b1, const float* t_w2, const float* t_b2,
                                   int t_n0, int t_n1, int t_n2, float* t_out, int t_ldo,
                                   const float* v_w0, const float* v_b0, const float* v_w1, const float* v_b1, const float* v_w2, const float* v_b2,
                                   int v_n0, int v_n1, int v_n2, float* v_out, int v_ldo,
                                   int num_steps, int degree_r, int degree_t, int64_t* r_act, int64_t* t_act,
                                   float slope, hipStream_t stream) {
  CMR_REQUIRE(x && w24 && b24 && w26 && b26 && e3d && B > 0 && npix > 0);
  CMR_REQUIRE(cmr_aligned16(x) && cmr_aligned16(w24) && cmr_aligned16(w26));
  AhArgs a{};
  a.x = x; a.npix = npix; a.w24 = w24; a.b24 = b24; a.w26 = w26; a.b26 = b26; a.e3d = e3d; a.slope = slope;
  // actions (optional): the logical widths degree * num_steps must fit the (padded) head widths
  CMR_REQUIRE((!r_act && !t_act) || (num_steps > 0 && degree_r > 0 && degree_t > 0 && degree_r * num_steps <= r_n2 && degree_t * num_steps <= t_n2));
  a.num_steps = (r_act || t_act) ? num_steps : 0;
  a.h[0] = AhHead{r_w0, r_b0, r_w1, r_b1, r_w2, r_b2, r_n0, r_n1, r_n2, r_out, r_ldo, r_act, degree_r};
  a.h[1] = AhHead{t_w0, t_b0, t_w1, t_b1, t_w2, t_b2, t_n0, t_n1, t_n2, t_out, t_ldo, t_act, degree_t};
  a.h[2] = AhHead{v_w0, v_b0, v_w1, v_b1, v_w2, v_b2, v_n0, v_n1, v_n2, v_out, v_ldo, nullptr, 0};
  for (int i = 0; i < 3; ++i) {
    const AhHead& h = a.h[i];
    CMR_REQUIRE(h.w0 && h.w1 && h.w2 && h.b0 && h.b1 && h.b2 && h.out && h.n2 > 0 && h.ldo >= h.n2);
    CMR_REQUIRE(h.n0 > 0 && h.n0 <= AH_MAXW && h.n0 % 4 == 0 && h.n1 > 0 && h.n1 <= AH_MAXW && h.n1 % 4 == 0);
    CMR_REQUIRE(cmr_aligned16(h.w0) && cmr_aligned16(h.w1) && cmr_aligned16(h.w2));
  }
  hipLaunchKernelGGL(agent_heads_kernel, dim3(B, 3), dim3(AH_THREADS), 0, stream, a);
  return cmr_launch_status();
}

extern "C" int cmr_agent_heads_train_f32(const float* x, int B, int npix, const float* w24, const float* b24, const float* w26,
                                   const float* b26, const float* e3d,
                                   const float* r_w0, const float* r_b0, const float* r_w1, const float* r_b1, const float* r_w2, const float* r_b2,
                                   int r_n0, int r_n1, int r_n2, float* r_out, int r_ldo,
                                   const float* t_w0, const float* t_b0, const float* t_w1, const float* t_b1, const float* t_w2, const float* t_b2,
                                   int t_n0, int t_n1, int t_n2, float* t_out, int t_ldo,
                                   const float* v_w0, const float* v_b0, const float* v_w1, const float* v_b1, const float* v_w2, const float* v_b2,
                                   int v_n0, int v_n1, int v_n2, float* v_out, int v_ldo,
                                   float* saves, int64_t saves_floats, float slope, hipStream_t stream) {
  // the training forward of the same tail (Train_Agent.py:263-305 through CMRAgent.py:52-56, 101-116): the logits AND every intermediate
  // the backward needs (AhArgs::saves) from one launch -- 13 launches (column mean, two 1x1 convs, nine head layers) on the serial
  // stretch between the towers' join and the loss of every update before round 6
  int64_t* r_act = nullptr; int64_t* t_act = nullptr;
  const int num_steps = 0, degree_r = 0, degree_t = 0;
  CMR_REQUIRE(saves && saves_floats >= (int64_t)B * (3 * AH_C + r_n0 + r_n1 + t_n0 + t_n1 + v_n0 + v_n1));
  CMR_REQUIRE(x && w24 && b24 && w26 && b26 && e3d && B > 0 && npix > 0);
  CMR_REQUIRE(cmr_aligned16(x) && cmr_aligned16(w24) && cmr_aligned16(w26));
  AhArgs a{};
  a.x = x; a.npix = npix; a.w24 = w24; a.b24 = b24; a.w26 = w26; a.b26 = b26; a.e3d = e3d; a.slope = slope;
  // actions (optional): the logical widths degree * num_steps must fit the (padded) head widths
  CMR_REQUIRE((!r_act && !t_act) || (num_steps > 0 && degree_r > 0 && degree_t > 0 && degree_r * num_steps <= r_n2 && degree_t * num_steps <= t_n2));
  a.num_steps = (r_act || t_act) ? num_steps : 0;
  a.h[0] = AhHead{r_w0, r_b0, r_w1, r_b1, r_w2, r_b2, r_n0, r_n1, r_n2, r_out, r_ldo, r_act, degree_r};
  a.h[1] = AhHead{t_w0, t_b0, t_w1, t_b1, t_w2, t_b2, t_n0, t_n1, t_n2, t_out, t_ldo, t_act, degree_t};
  a.h[2] = AhHead{v_w0, v_b0, v_w1, v_b1, v_w2, v_b2, v_n0, v_n1, v_n2, v_out, v_ldo, nullptr, 0};
  for (int i = 0; i < 3; ++i) {
    const AhHead& h = a.h[i];
    CMR_REQUIRE(h.w0 && h.w1 && h.w2 && h.b0 && h.b1 && h.b2 && h.out && h.n2 > 0 && h.ldo >= h.n2);
    CMR_REQUIRE(h.n0 > 0 && h.n0 <= AH_MAXW && h.n0 % 4 == 0 && h.n1 > 0 && h.n1 <= AH_MAXW && h.n1 % 4 == 0);
    CMR_REQUIRE(cmr_aligned16(h.w0) && cmr_aligned16(h.w1) && cmr_aligned16(h.w2));
  }
  a.saves = saves;
  hipLaunchKernelGGL(agent_heads_kernel, dim3(B, 3), dim3(AH_THREADS), 0, stream, a);
  return cmr_launch_status();
}
