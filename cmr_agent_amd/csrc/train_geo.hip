// Backward kernels of the geometric model's training step (second half: reference Train_Geo.py:166-174 =
// `loss.backward()` through MultiHeadModel).  The training forward of cmr_agent_amd/train/geo_update.py is the op-level
// composition of the inference entry points (linear, conv, layer norm, attention cores, gathers, segment softmax ...) with
// BatchNorm in batch-statistics mode; every op records its backward on a tape, and the backward of each op is one of the
// streaming kernels below (or a contraction of csrc/wgrad.hip / a forward kernel on transposed weights).  All of it is
// HBM- or latency-class row work over [rows, 64]-shaped maps; reductions are two-stage with a fixed combination order.
#include "cmr_common.h"

namespace {

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
inline unsigned ew_grid(int64_t items) {
  int64_t g = (items + 255) / 256;
  return (unsigned)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

// y += alpha * x   (gradient accumulation when a value feeds several consumers)
__global__ __launch_bounds__(256) void axpy_kernel(float* __restrict__ y, int64_t ldy, const float* __restrict__ x, int64_t ldx, float alpha,
                                                   int64_t rows, int C) {
  const int q = C >> 2;
  const int64_t total = rows * q;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / q;
    const int c = (int)(i - r * q) * 4;
    st4(y + r * ldy + c, ld4(y + r * ldy + c) + alpha * ld4(x + r * ldx + c));
  }
}

__device__ __forceinline__ float act_grad(float x, int act, float p) {
  switch (act) {
    case CMR_ACT_RELU: return x > 0.f ? 1.f : 0.f;
    case CMR_ACT_LRELU: return x > 0.f ? 1.f : p;
    case CMR_ACT_GELU: return 0.5f * (1.f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * expf(-0.5f * x * x);
    case CMR_ACT_ELU1: return x > 0.f ? 1.f : expf(x);
    default: return 1.f;
  }
}

// y = act(x)  /  dx (+)= dy * act'(x)   (x = the pre-activation)
__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ x, int64_t ldx, float* __restrict__ y, int64_t ldy, int64_t rows,
                                                      int C, int act, float p) {
  const int q = C >> 2;
  const int64_t total = rows * q;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / q;
    const int c = (int)(i - r * q) * 4;
    f32x4 v = ld4(x + r * ldx + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = cmr_act(v[e], act, p);
    st4(y + r * ldy + c, v);
  }
}

__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dy, int64_t lddy, const float* __restrict__ x, int64_t ldx,
                                                      float* __restrict__ dx, int64_t lddx, int64_t rows, int C, int act, float p, int accumulate) {
  const int q = C >> 2;
  const int64_t total = rows * q;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / q;
    const int c = (int)(i - r * q) * 4;
    const f32x4 g = ld4(dy + r * lddy + c), v = ld4(x + r * ldx + c);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = g[e] * act_grad(v[e], act, p);
    if (accumulate) o += ld4(dx + r * lddx + c);
    st4(dx + r * lddx + c, o);
  }
}

// ---- LayerNorm over 64 channels: dx = rstd (g - mean(g) - xhat mean(g xhat)), g = dy gamma; dgamma / dbeta partials -------------
// 16 lanes per row (float4 each); a workgroup's 16 row slots walk the rows with stride; per-lane partial sums of
// dy * xhat and dy are combined over the row slots through LDS -> part[blk][2][64]
__global__ __launch_bounds__(256) void ln64_bwd_kernel(const float* __restrict__ dy, int64_t lddy, const float* __restrict__ x, int64_t ldx,
                                                       const float* __restrict__ gamma, float eps, float* __restrict__ dx, int64_t lddx,
                                                       int accumulate, int64_t rows, float* __restrict__ part) {
  __shared__ float sm[16][2][64];
  const int slot = threadIdx.x >> 4, c = (threadIdx.x & 15) * 4;
  const f32x4 gm = ld4(gamma + c);
  f32x4 sg = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};
  for (int64_t r = (int64_t)blockIdx.x * 16 + slot; r < rows; r += (int64_t)gridDim.x * 16) {
    const f32x4 v = ld4(x + r * ldx + c), d = ld4(dy + r * lddy + c);
    float s = (v[0] + v[1]) + (v[2] + v[3]);
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) s += __shfl_xor(s, m);
    const float mean = s * (1.f / 64.f);
    const f32x4 xc = v - mean;
    float q = (xc[0] * xc[0] + xc[1] * xc[1]) + (xc[2] * xc[2] + xc[3] * xc[3]);
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) q += __shfl_xor(q, m);
    const float rstd = rsqrtf(q * (1.f / 64.f) + eps);
    const f32x4 xh = xc * rstd;
    const f32x4 g = d * gm;
    float a = (g[0] + g[1]) + (g[2] + g[3]);
    float b = (g[0] * xh[0] + g[1] * xh[1]) + (g[2] * xh[2] + g[3] * xh[3]);
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) {
      a += __shfl_xor(a, m);
      b += __shfl_xor(b, m);
    }
    f32x4 o = rstd * (g - a * (1.f / 64.f) - xh * (b * (1.f / 64.f)));
    if (accumulate) o += ld4(dx + r * lddx + c);
    st4(dx + r * lddx + c, o);
    sg += d * xh;
    sb += d;
  }
  st4(&sm[slot][0][c], sg);
  st4(&sm[slot][1][c], sb);
  __syncthreads();
  if (threadIdx.x < 128) {
    const int k = threadIdx.x >> 6, ch = threadIdx.x & 63;
    float t = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) t += sm[s2][k][ch];
    part[(int64_t)blockIdx.x * 128 + threadIdx.x] = t;
  }
}

// out[i] (+)= sum_b part[b][i]  for i < n: one wave per output, fixed order (shared by every two-stage reduction here)
__global__ __launch_bounds__(64) void reduce_partials_kernel(const float* __restrict__ part, int nblk, int n, float* __restrict__ out,
                                                             const int32_t* __restrict__ out_map, int accumulate) {
  const int i = blockIdx.x, lane = threadIdx.x;
  double s = 0.0;
  for (int b = lane; b < nblk; b += 64) s += (double)part[(int64_t)b * n + i];
  s = wave_sum_d(s);
  if (lane == 0) {
    float* d = out + (out_map ? out_map[i] : i);
    *d = accumulate ? *d + (float)s : (float)s;
  }
}

// the same for two interleaved outputs: part[b][0..n) -> out0, part[b][n..2n) -> out1 (LayerNorm's dgamma / dbeta in one launch)
__global__ __launch_bounds__(64) void reduce_partials2_kernel(const float* __restrict__ part, int nblk, int n, float* __restrict__ out0,
                                                              float* __restrict__ out1, int accumulate) {
  const int i = blockIdx.x, lane = threadIdx.x;
  double s = 0.0;
  for (int b = lane; b < nblk; b += 64) s += (double)part[(int64_t)b * 2 * n + i];
  s = wave_sum_d(s);
  if (lane == 0) {
    float* d = i < n ? out0 + i : out1 + (i - n);
    *d = accumulate ? *d + (float)s : (float)s;
  }
}

// ---- F.normalize over 64 channels: y = x / max(|x|, 1e-12);  dx = (dy - y (y . dy)) / max(|x|, 1e-12) -------------------------
__global__ __launch_bounds__(256) void l2norm64_bwd_kernel(const float* __restrict__ dy, int64_t lddy, const float* __restrict__ x, int64_t ldx,
                                                           float* __restrict__ dx, int64_t lddx, int accumulate, int64_t rows) {
  const int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int c = (threadIdx.x & 15) * 4;
  const bool ok = row < rows;
  f32x4 v = {0.f, 0.f, 0.f, 0.f}, g = {0.f, 0.f, 0.f, 0.f};
  if (ok) {
    v = ld4(x + row * ldx + c);
    g = ld4(dy + row * lddy + c);
  }
  float q = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
  float t = (v[0] * g[0] + v[1] * g[1]) + (v[2] * g[2] + v[3] * g[3]);
#pragma unroll
  for (int m = 1; m < 16; m <<= 1) {
    q += __shfl_xor(q, m);
    t += __shfl_xor(t, m);
  }
  const float n = sqrtf(q);
  const float d = fmaxf(n, 1e-12f);
  if (ok) {
    // n >= eps: y = x / n, dx = dy / n - x (x . dy) / n^3;  n < eps (clamped): y = x / eps, dx = dy / eps
    f32x4 o = g / d;
    if (n >= 1e-12f) o -= v * (t / (d * d * d));
    if (accumulate) o += ld4(dx + row * lddx + c);
    st4(dx + row * lddx + c, o);
  }
}

// ---- layout ops -------------------------------------------------------------------------------------------------------------
// out[b, 2y, 2x, :] = g[b, y, x, :], zero elsewhere: the data / weight gradient of a stride-2 convolution are the stride-1 ones
// of its zero-inserted output gradient
__global__ __launch_bounds__(256) void zero_insert2_kernel(const float* __restrict__ g, float* __restrict__ out, int B, int Ho, int Wo, int H,
                                                           int W, int C) {
  const int q = C >> 2;
  const int64_t total = (int64_t)B * H * W * q;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % q) * 4;
    int64_t p = i / q;
    const int xx = (int)(p % W);
    p /= W;
    const int yy = (int)(p % H), b = (int)(p / H);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (!(xx & 1) && !(yy & 1) && (yy >> 1) < Ho && (xx >> 1) < Wo) v = ld4(g + (((int64_t)b * Ho + (yy >> 1)) * Wo + (xx >> 1)) * C + c);
    st4(out + (i / q) * C + c, v);
  }
}

// inverse of patchify: dx[b, ty P + ky, tx P + kx, c] = dpatches[b T + ty Wp + tx, (ky P + kx) C + c]
__global__ __launch_bounds__(256) void patchify_bwd_kernel(const float* __restrict__ dp, float* __restrict__ dx, int B, int H, int W, int C, int P,
                                                           int accumulate) {
  const int c4n = C / 4, Hp = H / P, Wp = W / P;
  const int64_t total = (int64_t)B * Hp * Wp * P * P * c4n;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % c4n) * 4;
    int64_t q = e / c4n;
    const int kx = (int)(q % P); q /= P;
    const int ky = (int)(q % P); q /= P;
    const int tx = (int)(q % Wp); q /= Wp;
    const int ty = (int)(q % Hp);
    const int b = (int)(q / Hp);
    float* d = dx + (((int64_t)b * H + ty * P + ky) * W + tx * P + kx) * C + c;
    f32x4 v = ld4(dp + e * 4);
    if (accumulate) v += ld4(d);
    st4(d, v);
  }
}

// backward of the nearest x s up-sampling of the proxies: dproxy[b, t, :] = sum over the s x s pixels of token t of dcat[.., C1:]
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const float* __restrict__ dcat, int64_t ldc, int coff, float* __restrict__ dproxy,
                                                           int B, int H, int W, int C2, int s, int accumulate) {
  const int c4n = C2 / 4, Hp = H / s, Wp = W / s;
  const int64_t total = (int64_t)B * Hp * Wp * c4n;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % c4n) * 4;
    int64_t t = e / c4n;
    const int tx = (int)(t % Wp), ty = (int)((t / Wp) % Hp), b = (int)(t / ((int64_t)Wp * Hp));
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    for (int dy = 0; dy < s; ++dy)
      for (int dxx = 0; dxx < s; ++dxx) a += ld4(dcat + (((int64_t)b * H + ty * s + dy) * W + tx * s + dxx) * ldc + coff + c);
    float* d = dproxy + t * C2 + c;
    if (accumulate) a += ld4(d);
    st4(d, a);
  }
}

// 3-channel 3x3 patches of an NHWC image with C = 4 (xyz0-style padding): cols[p, (ky 3 + kx) 4 + c] (36 used of 36, zero
// outside the image) -- the stem's convolutions as row GEMMs (K = 36) in the training path; and its adjoint
__global__ __launch_bounds__(256) void im2col3_kernel(const float* __restrict__ x, float* __restrict__ cols, int B, int H, int W) {
  const int64_t total = (int64_t)B * H * W * 9;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int t = (int)(e % 9);
    const int64_t p = e / 9;
    const int xx = (int)(p % W), yy = (int)((p / W) % H);
    const int64_t b = p / ((int64_t)W * H);
    const int iy = yy + t / 3 - 1, ix = xx + t % 3 - 1;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = ld4(x + ((b * H + iy) * W + ix) * 4);
    st4(cols + p * 36 + t * 4, v);
  }
}

__global__ __launch_bounds__(256) void col2im3_kernel(const float* __restrict__ dcols, float* __restrict__ dx, int B, int H, int W, int accumulate) {
  const int64_t total = (int64_t)B * H * W;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (int64_t)gridDim.x * blockDim.x) {
    const int xx = (int)(p % W), yy = (int)((p / W) % H);
    const int64_t b = p / ((int64_t)W * H);
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 9; ++t) {                       // pixel p is tap t of the output pixel at (yy - (t/3 - 1), xx - (t%3 - 1))
      const int oy = yy - (t / 3 - 1), ox = xx - (t % 3 - 1);
      if (oy >= 0 && oy < H && ox >= 0 && ox < W) a += ld4(dcols + ((b * H + oy) * W + ox) * 36 + t * 4);
    }
    if (accumulate) a += ld4(dx + p * 4);
    st4(dx + p * 4, a);
  }
}


// ---- dropout (nn.Dropout in train mode): y = x * keep / (1 - p), element index = row * C + column; in place allowed.  The backward
// pass is the same launch on the gradient (same seed and site -> same mask).
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, int64_t ldx, float* __restrict__ y, int64_t ldy, int64_t rows,
                                                      int C, uint32_t thr, float keep_scale, const int64_t* __restrict__ seed_ptr,
                                                      uint64_t site) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int c4 = C / 4;
  const int64_t r = e / c4;
  if (r >= rows) return;
  const int c = (int)(e - r * c4) * 4;
  const uint64_t seed = (uint64_t)seed_ptr[0];
  f32x4 v = ld4(x + r * ldx + c);
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = cmr_keep(seed, site, (uint64_t)r * C + c + i, thr) ? v[i] * keep_scale : 0.f;
  st4(y + r * ldy + c, v);
}

// ---- softmax attention backward (8 heads x 8 dims; forward = cmr_mha_f32) ------------------------------------------------------
// P = softmax(Q K^T / sqrt 8); O = P V.  dV = P^T dO; dS = P o (dO V^T - D), D_i = dO_i . O_i; dQ = dS K / sqrt 8; dK = dS^T Q / sqrt 8.
// Kernel 1: four lanes per (query, head), K / V of the head in LDS: log-sum-exp of the row (online, one pass), D, dQ (written) -- lse / D
// are kept for kernel 2: four lanes per (key, head), Q / dO / lse / D of the head in LDS: dK, dV.  No atomics.
constexpr int AH_DH = 8, AH_NH = 8;

// With dropout on the probabilities (forward cmr_mha_dropout_f32): O = (P o M) V with M = mask / (1 - p), so dV = (P o M)^T dO and
// dP = M o (dO V^T); D_i = dO_i . O_i still equals sum_k P_ik dP_ik, and dS = P o (dP - D) as before.  The mask is regenerated.
template <bool DROP>
__global__ __launch_bounds__(256) void mha_bwd_dq_kernel(const float* __restrict__ q, int64_t ldq, const float* __restrict__ k, int64_t ldk,
                                                         const float* __restrict__ v, int64_t ldv, const float* __restrict__ o, int64_t ldo,
                                                         const float* __restrict__ dout, int64_t lddo, float* __restrict__ dq, int64_t lddq,
                                                         int acc_dq, float* __restrict__ lse, float* __restrict__ dsum, int Tq, int Tk,
                                                         float scale, uint32_t thr, float keep_scale, const int64_t* __restrict__ seed_ptr,
                                                         uint64_t site) {
  extern __shared__ __attribute__((aligned(16))) float kv[];
  float* ks = kv;
  float* vs = kv + (size_t)Tk * AH_DH;
  const int head = blockIdx.y, b = blockIdx.z;
  for (int e = threadIdx.x; e < Tk * 2; e += 256) {
    const int t = e >> 1, half = (e & 1) * 4;
    st4(&ks[t * AH_DH + half], ld4(k + ((int64_t)b * Tk + t) * ldk + head * AH_DH + half));
    st4(&vs[t * AH_DH + half], ld4(v + ((int64_t)b * Tk + t) * ldv + head * AH_DH + half));
  }
  __syncthreads();
  // four lanes per query (keys t = sub, sub + 4, ...): the one-thread-per-query form walked the keys three times in a serial loop of
  // up to 418 iterations, 64 workgroups on the chip -- 96 us for an 80 x 256-token problem
  const int sub = threadIdx.x & 3;
  const int tqr = blockIdx.x * 64 + (threadIdx.x >> 2);
  const bool live = tqr < Tq;
  const int tq = live ? tqr : Tq - 1;                   // clamped: every lane takes part in the shuffles
  const int64_t row = (int64_t)b * Tq + tq;
  float qv[8], gv[8], ov[8];
#pragma unroll
  for (int d = 0; d < 8; ++d) {
    qv[d] = q[row * ldq + head * 8 + d];
    gv[d] = dout[row * lddo + head * 8 + d];
    ov[d] = o[row * ldo + head * 8 + d];
  }
  float D = 0.f;
#pragma unroll
  for (int d = 0; d < 8; ++d) D += gv[d] * ov[d];
  // log-sum-exp of the row: online (max, sum) over this lane's keys, combined over the four lanes
  float m = -INFINITY, l = 0.f;
  for (int t = sub; t < Tk; t += 4) {
    float sc = 0.f;
#pragma unroll
    for (int d = 0; d < 8; ++d) sc += qv[d] * ks[t * 8 + d];
    sc *= scale;
    const float mn = fmaxf(m, sc);
    l = l * expf(m - mn) + expf(sc - mn);
    m = mn;
  }
#pragma unroll
  for (int x = 1; x <= 2; x <<= 1) {
    const float mo = __shfl_xor(m, x), lo = __shfl_xor(l, x);
    const float mn = fmaxf(m, mo);
    l = (m == -INFINITY ? 0.f : l * expf(m - mn)) + (mo == -INFINITY ? 0.f : lo * expf(mo - mn));
    m = mn;
  }
  const float L = m + logf(l);
  const uint64_t seed = DROP ? (uint64_t)seed_ptr[0] : 0;
  const uint64_t mrow = (((uint64_t)b * AH_NH + head) * Tq + tq) * Tk;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int t = sub; t < Tk; t += 4) {
    float sc = 0.f, dp = 0.f;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      sc += qv[d] * ks[t * 8 + d];
      dp += gv[d] * vs[t * 8 + d];
    }
    if (DROP) dp = cmr_keep(seed, site, mrow + t, thr) ? dp * keep_scale : 0.f;
    const float ds = expf(sc * scale - L) * (dp - D) * scale;
#pragma unroll
    for (int d = 0; d < 8; ++d) acc[d] += ds * ks[t * 8 + d];
  }
#pragma unroll
  for (int d = 0; d < 8; ++d) {
    acc[d] += __shfl_xor(acc[d], 1);
    acc[d] += __shfl_xor(acc[d], 2);
  }
  if (!live || sub != 0) return;
#pragma unroll
  for (int d = 0; d < 8; ++d) {
    float* dst = dq + row * lddq + head * 8 + d;
    *dst = acc_dq ? *dst + acc[d] : acc[d];
  }
  lse[row * 8 + head] = L;
  dsum[row * 8 + head] = D;
}

template <bool DROP>
__global__ __launch_bounds__(256) void mha_bwd_dkv_kernel(const float* __restrict__ q, int64_t ldq, const float* __restrict__ k, int64_t ldk,
                                                          const float* __restrict__ v, int64_t ldv, const float* __restrict__ dout, int64_t lddo,
                                                          const float* __restrict__ lse, const float* __restrict__ dsum, float* __restrict__ dk,
                                                          int64_t lddk, int acc_dk, float* __restrict__ dv, int64_t lddv, int acc_dv, int Tq,
                                                          int Tk, float scale, uint32_t thr, float keep_scale,
                                                          const int64_t* __restrict__ seed_ptr, uint64_t site) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* qs = sm;                              // [Tq][8]
  float* gs = qs + (size_t)Tq * 8;             // [Tq][8]
  float* ls = gs + (size_t)Tq * 8;             // [Tq]
  float* ds_ = ls + Tq;                        // [Tq]
  const int head = blockIdx.y, b = blockIdx.z;
  for (int e = threadIdx.x; e < Tq * 2; e += 256) {
    const int t = e >> 1, half = (e & 1) * 4;
    st4(&qs[t * 8 + half], ld4(q + ((int64_t)b * Tq + t) * ldq + head * 8 + half));
    st4(&gs[t * 8 + half], ld4(dout + ((int64_t)b * Tq + t) * lddo + head * 8 + half));
  }
  for (int t = threadIdx.x; t < Tq; t += 256) {
    ls[t] = lse[((int64_t)b * Tq + t) * 8 + head];
    ds_[t] = dsum[((int64_t)b * Tq + t) * 8 + head];
  }
  __syncthreads();
  const int sub = threadIdx.x & 3;                      // four lanes per key (queries t = sub, sub + 4, ...)
  const int tkr = blockIdx.x * 64 + (threadIdx.x >> 2);
  const bool live = tkr < Tk;
  const int tk = live ? tkr : Tk - 1;
  const int64_t row = (int64_t)b * Tk + tk;
  float kvv[8], vv[8];
#pragma unroll
  for (int d = 0; d < 8; ++d) {
    kvv[d] = k[row * ldk + head * 8 + d];
    vv[d] = v[row * ldv + head * 8 + d];
  }
  float ak[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, av[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const uint64_t seed = DROP ? (uint64_t)seed_ptr[0] : 0;
  const uint64_t mhead = ((uint64_t)b * AH_NH + head) * Tq;
  for (int t = sub; t < Tq; t += 4) {
    float sc = 0.f, dp = 0.f;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      sc += qs[t * 8 + d] * kvv[d];
      dp += gs[t * 8 + d] * vv[d];
    }
    const float p = expf(sc * scale - ls[t]);
    float pm = p;
    if (DROP) {
      const float mf = cmr_keep(seed, site, (mhead + t) * Tk + tk, thr) ? keep_scale : 0.f;
      pm = p * mf;
      dp *= mf;
    }
    const float dsv = p * (dp - ds_[t]) * scale;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      av[d] += pm * gs[t * 8 + d];
      ak[d] += dsv * qs[t * 8 + d];
    }
  }
#pragma unroll
  for (int d = 0; d < 8; ++d) {
    ak[d] += __shfl_xor(ak[d], 1); ak[d] += __shfl_xor(ak[d], 2);
    av[d] += __shfl_xor(av[d], 1); av[d] += __shfl_xor(av[d], 2);
  }
  if (!live || sub != 0) return;
#pragma unroll
  for (int d = 0; d < 8; ++d) {
    float* a = dk + row * lddk + head * 8 + d;
    float* c = dv + row * lddv + head * 8 + d;
    *a = acc_dk ? *a + ak[d] : ak[d];
    *c = acc_dv ? *c + av[d] : av[d];
  }
}

// ---- linear attention core backward (forward = cmr_la_reduce_f32 + cmr_la_apply_f32; state layout [h][d][v] x 512 | [h][d] x 64) ----
// msg = S num / den, num[l,h,v] = sum_d Qf[l,h,d] KV[h,d,v], den[l,h] = sum_d Qf[l,h,d] Ksum[h,d] + eps, KV = sum_s Kf V / S.
// query side: dQf, and per-wave partials of dKV / dKsum;  source side: dKf, dV from the reduced dstate.
__global__ __launch_bounds__(256) void la_bwd_query_kernel(const float* __restrict__ qf, int64_t ldq, const float* __restrict__ kvsum,
                                                           const float* __restrict__ dmsg, int64_t lddm, float* __restrict__ dqf, int64_t lddq,
                                                           int acc_dq, float* __restrict__ part, int L, int S, float eps, int tokens_per_wave) {
  const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hh = lane >> 3, vv = lane & 7;
  const float* kvb = kvsum + (int64_t)b * 576;
  float kvr[8], ksr[8], akv[8], aks = 0.f;
#pragma unroll
  for (int d = 0; d < 8; ++d) {
    kvr[d] = kvb[hh * 64 + d * 8 + vv];
    ksr[d] = kvb[512 + hh * 8 + d];
    akv[d] = 0.f;
  }
  const float fs = (float)S;
  const int l0 = (blockIdx.x * 4 + wave) * tokens_per_wave;
  const int l1 = min(L, l0 + tokens_per_wave);
  for (int l = l0; l < l1; ++l) {
    const int64_t row = (int64_t)b * L + l;
    const float qv = qf[row * ldq + lane];
    const float g = dmsg[row * lddm + lane];
    float qd[8], num = 0.f, den = 0.f;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      qd[d] = __shfl(qv, hh * 8 + d);
      num += qd[d] * kvr[d];
      den += qd[d] * ksr[d];
    }
    den += eps;
    const float dnum = fs * g / den;                 // d loss / d num[l,h,v]
    float t = g * (fs * num / den);                  // dmsg * msg, summed over v below
    t += __shfl_xor(t, 1); t += __shfl_xor(t, 2); t += __shfl_xor(t, 4);
    const float dden = -t / den;
    float mine = 0.f;                                // dQf[l, h, d = vv]
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      float c = dnum * kvr[d];
      c += __shfl_xor(c, 1); c += __shfl_xor(c, 2); c += __shfl_xor(c, 4);
      if (d == vv) mine = c + dden * ksr[d];
      akv[d] += qd[d] * dnum;
    }
    aks += dden * qv;                                 // Qf[l, h, d = vv] is this lane's own element
    float* dst = dqf + row * lddq + lane;
    *dst = acc_dq ? *dst + mine : mine;
  }
  float* p = part + ((int64_t)b * gridDim.x * 4 + blockIdx.x * 4 + wave) * 576;
#pragma unroll
  for (int d = 0; d < 8; ++d) p[hh * 64 + d * 8 + vv] = akv[d];
  p[512 + hh * 8 + vv] = aks;
}

__global__ __launch_bounds__(64) void la_bwd_state_kernel(const float* __restrict__ part, int nper, float* __restrict__ dstate) {
  const int i = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
  double s = 0.0;
  for (int k = lane; k < nper; k += 64) s += (double)part[((int64_t)b * nper + k) * 576 + i];
  s = wave_sum_d(s);
  if (lane == 0) dstate[(int64_t)b * 576 + i] = (float)s;
}

__global__ __launch_bounds__(256) void la_bwd_source_kernel(const float* __restrict__ kf, int64_t ldk, const float* __restrict__ v, int64_t ldv,
                                                            const float* __restrict__ dstate, float* __restrict__ dkf, int64_t lddk, int acc_dk,
                                                            float* __restrict__ dv, int64_t lddv, int acc_dv, int S, int tokens_per_wave) {
  const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hh = lane >> 3, vv = lane & 7;
  const float* db = dstate + (int64_t)b * 576;
  float dkv[8];
#pragma unroll
  for (int d = 0; d < 8; ++d) dkv[d] = db[hh * 64 + d * 8 + vv];
  const float dks = db[512 + hh * 8 + vv];           // dKsum[h, d = vv]
  const float inv = 1.f / (float)S;
  const int s0 = (blockIdx.x * 4 + wave) * tokens_per_wave;
  const int s1 = min(S, s0 + tokens_per_wave);
  for (int s = s0; s < s1; ++s) {
    const int64_t row = (int64_t)b * S + s;
    const float kk = kf[row * ldk + lane], val = v[row * ldv + lane];
    float dvv = 0.f, mine = 0.f;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      dvv += __shfl(kk, hh * 8 + d) * dkv[d];
      float c = dkv[d] * val;
      c += __shfl_xor(c, 1); c += __shfl_xor(c, 2); c += __shfl_xor(c, 4);
      if (d == vv) mine = c * inv + dks;
    }
    float* a = dkf + row * lddk + lane;
    float* c2 = dv + row * lddv + lane;
    *a = acc_dk ? *a + mine : mine;
    *c2 = acc_dv ? *c2 + dvv * inv : dvv * inv;
  }
}

// ---- segment softmax backward (forward = cmr_segment_softmax_f32): out[s,c] = sum_i p_i[c] vp_i[c], p = softmax_i(attn_i[c] scale) ----
// d vp_i = p_i dout;  d attn_i = scale p_i dout (vp_i - out).  One wave per segment, lane = channel; rows belong to one segment.
__global__ __launch_bounds__(256) void segment_softmax_bwd_kernel(const float* __restrict__ attn, const float* __restrict__ vp,
                                                                  const int32_t* __restrict__ order, const int32_t* __restrict__ offsets,
                                                                  int fixed_len, float scale, const float* __restrict__ dout,
                                                                  float* __restrict__ dattn, float* __restrict__ dvp, int64_t nseg) {
  const int64_t seg = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (seg >= nseg) return;
  const int64_t lo = offsets ? offsets[seg] : seg * fixed_len;
  const int64_t hi = offsets ? offsets[seg + 1] : lo + fixed_len;
  // eight members in flight per pass, as in segment_softmax_kernel (same order of operations: bit-identical)
  constexpr int U = 8;
  float m = -INFINITY;
  int64_t i = lo;
  for (; i + U <= hi; i += U) {
    int64_t r[U];
    float a[U];
#pragma unroll
    for (int u = 0; u < U; ++u) r[u] = order ? (int64_t)order[i + u] : i + u;
#pragma unroll
    for (int u = 0; u < U; ++u) a[u] = attn[r[u] * 64 + lane];
#pragma unroll
    for (int u = 0; u < U; ++u) m = fmaxf(m, a[u] * scale);
  }
  for (; i < hi; ++i) {
    const int64_t r = order ? order[i] : i;
    m = fmaxf(m, attn[r * 64 + lane] * scale);
  }
  float l = 0.f, acc = 0.f;
  for (i = lo; i + U <= hi; i += U) {
    int64_t r[U];
    float a[U], w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) r[u] = order ? (int64_t)order[i + u] : i + u;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      a[u] = attn[r[u] * 64 + lane];
      w[u] = vp[r[u] * 64 + lane];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float p = expf(a[u] * scale - m);
      l += p;
      acc += p * w[u];
    }
  }
  for (; i < hi; ++i) {
    const int64_t r = order ? order[i] : i;
    const float p = expf(attn[r * 64 + lane] * scale - m);
    l += p;
    acc += p * vp[r * 64 + lane];
  }
  const float o = hi > lo ? acc / l : 0.f;
  const float g = dout[seg * 64 + lane];
  for (i = lo; i + U <= hi; i += U) {
    int64_t r[U];
    float a[U], w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) r[u] = order ? (int64_t)order[i + u] : i + u;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      a[u] = attn[r[u] * 64 + lane];
      w[u] = vp[r[u] * 64 + lane];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float p = expf(a[u] * scale - m) / l;
      dvp[r[u] * 64 + lane] = p * g;
      dattn[r[u] * 64 + lane] = scale * p * g * (w[u] - o);
    }
  }
  for (; i < hi; ++i) {
    const int64_t r = order ? order[i] : i;
    const float p = expf(attn[r * 64 + lane] * scale - m) / l;
    const float w = vp[r * 64 + lane];
    dvp[r * 64 + lane] = p * g;
    dattn[r * 64 + lane] = scale * p * g * (w - o);
  }
}

// ---- focal loss backward (forward = cmr_focal_metrics_f32; MultiHeadModel.py:49-50, focal_loss.py:55-110) -----------------------
// L = mean_r sum_c (onehot_c + 1e-6) f(s_c), f(s) = -alpha (1 - s)^2 log s, s = softmax + 1e-6
__global__ __launch_bounds__(256) void focal_bwd_kernel(const float* __restrict__ logits, int64_t ld, const int64_t* __restrict__ label,
                                                        float alpha, int64_t rows, float gscale, float* __restrict__ dl, int64_t ldd) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  const float z0 = logits[r * ld], z1 = logits[r * ld + 1];
  const float m = fmaxf(z0, z1);
  const float e0 = expf(z0 - m), e1 = expf(z1 - m);
  const float p[2] = {e0 / (e0 + e1), e1 / (e0 + e1)};
  const int lab = (int)label[r];
  float c[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const float s = p[k] + 1e-6f;
    const float fp = -alpha * (-2.f * (1.f - s) * logf(s) + (1.f - s) * (1.f - s) / s);
    c[k] = ((lab == k ? 1.f : 0.f) + 1e-6f) * fp * p[k];          // w_c f'(s_c) p_c
  }
  const float tot = c[0] + c[1];
  const float sc = gscale / (float)rows;
  dl[r * ldd] = sc * (c[0] - p[0] * tot);
  dl[r * ldd + 1] = sc * (c[1] - p[1] * tot);
}

// ---- circle loss backward (forward = cmr_circle_loss_f32; MultiHeadModel.py:141-178, :240-262) ----------------------------------
// pass 1: d_ij, masks, the four exponent tables' row / column log-sum-exps; pass 2: G_ij = dL/dd_ij / d_ij; pass 3: gradients of the
// sampled feature vectors; pass 4: sequential scatter into the dense gradient maps (samples may repeat an index)
struct CircleArgs {
  const float* pc_feat; const float* img_feat; const int64_t* pc_idx; const int64_t* xy_int; const float* xy_float;
  int B, N, h, w, n;
  float dist_thres, pos_margin, neg_margin, log_scale, gscale;
  float* dmat;   // [B][n][n] distances, later G
  float* stats;  // [B][8][n]: row (max, sum) of tp, tn; column (max, sum) of tp, tn   -> stored as lse: [B][4][n] used
  float* dpts;   // [B][n][64]
  float* dpix;   // [B][n][64]
};

__device__ __forceinline__ void circle_terms(const CircleArgs& a, int b, int i, int j, float d, float& tp, float& tn, float& pw, float& nw) {
  const float fx = a.xy_float[((int64_t)b * 2 + 0) * a.n + i] - (float)a.xy_int[((int64_t)b * 2 + 0) * a.n + j];
  const float fy = a.xy_float[((int64_t)b * 2 + 1) * a.n + i] - (float)a.xy_int[((int64_t)b * 2 + 1) * a.n + j];
  const bool posm = sqrtf(fx * fx + fy * fy) <= a.dist_thres;
  const float pos = d - (posm ? 0.f : 1e5f);
  pw = fmaxf(pos - a.pos_margin, 0.f);
  tp = a.log_scale * (pos - a.pos_margin) * pw;
  const float neg = d + (posm ? 1e5f : 0.f);
  nw = fmaxf(a.neg_margin - neg, 0.f);
  tn = a.log_scale * (a.neg_margin - neg) * nw;
}

__global__ __launch_bounds__(64) void circle_dist_kernel(const CircleArgs a) {
  // one wave per (i, b): lane = channel; d_ij for all j.  Eight gathered pixel rows in flight per trip (the one-row-per-iteration loop was a
  // chain of 512 dependent memory round trips per wave: ~300 us of the 800 us this loss's backward took); same sums in the same order.
  const int i = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
  const float pv = a.pc_feat[((int64_t)b * a.N + a.pc_idx[(int64_t)b * a.n + i]) * 64 + lane];
  const int64_t* xi = a.xy_int + (int64_t)b * 2 * a.n;
  constexpr int U = 8;
  int j = 0;
  for (; j + U <= a.n; j += U) {
    float iv[U], sq[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t px = xi[j + u], py = xi[a.n + j + u];
      iv[u] = a.img_feat[(((int64_t)b * a.h + py) * a.w + px) * 64 + lane];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) sq[u] = (pv - iv[u]) * (pv - iv[u]);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
#pragma unroll
      for (int u = 0; u < U; ++u) sq[u] += __shfl_xor(sq[u], m);
    if (lane < U) {
      float v = sq[0];
#pragma unroll
      for (int u = 1; u < U; ++u) v = lane == u ? sq[u] : v;      // (every lane holds every total after the butterfly)
      a.dmat[((int64_t)b * a.n + i) * a.n + j + lane] = sqrtf(v);
    }
  }
  for (; j < a.n; ++j) {
    const int64_t px = xi[j], py = xi[a.n + j];
    const float iv = a.img_feat[(((int64_t)b * a.h + py) * a.w + px) * 64 + lane];
    float s = (pv - iv) * (pv - iv);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
    if (lane == 0) a.dmat[((int64_t)b * a.n + i) * a.n + j] = sqrtf(s);
  }
}

// lse over rows (dir 0: fixed i, over j) or columns (dir 1): stats[b][dir*2 + which][index]
__global__ __launch_bounds__(64) void circle_lse_kernel(const CircleArgs a) {
  const int idx = blockIdx.x, b = blockIdx.y, dir = blockIdx.z, lane = threadIdx.x;
  float mp = -INFINITY, mn = -INFINITY;
  for (int k = lane; k < a.n; k += 64) {
    const int i = dir ? k : idx, j = dir ? idx : k;
    float tp, tn, pw, nw;
    circle_terms(a, b, i, j, a.dmat[((int64_t)b * a.n + i) * a.n + j], tp, tn, pw, nw);
    mp = fmaxf(mp, tp);
    mn = fmaxf(mn, tn);
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    mp = fmaxf(mp, __shfl_xor(mp, m));
    mn = fmaxf(mn, __shfl_xor(mn, m));
  }
  float sp = 0.f, sn = 0.f;
  for (int k = lane; k < a.n; k += 64) {
    const int i = dir ? k : idx, j = dir ? idx : k;
    float tp, tn, pw, nw;
    circle_terms(a, b, i, j, a.dmat[((int64_t)b * a.n + i) * a.n + j], tp, tn, pw, nw);
    sp += expf(tp - mp);
    sn += expf(tn - mn);
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    sp += __shfl_xor(sp, m);
    sn += __shfl_xor(sn, m);
  }
  if (lane == 0) {
    a.stats[(((int64_t)b * 4 + dir * 2 + 0) * a.n) + idx] = mp + logf(sp);
    a.stats[(((int64_t)b * 4 + dir * 2 + 1) * a.n) + idx] = mn + logf(sn);
  }
}

__global__ __launch_bounds__(256) void circle_weight_kernel(const CircleArgs a) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)a.B * a.n * a.n) return;
  const int j = (int)(e % a.n), i = (int)((e / a.n) % a.n), b = (int)(e / ((int64_t)a.n * a.n));
  const float d = a.dmat[e];
  float tp, tn, pw, nw;
  circle_terms(a, b, i, j, d, tp, tn, pw, nw);
  const float* st = a.stats + (int64_t)b * 4 * a.n;
  const float lpr = st[0 * a.n + i], lnr = st[1 * a.n + i], lpc = st[2 * a.n + j], lnc = st[3 * a.n + j];
  const float sr = 1.f / (1.f + expf(-(lpr + lnr))), scl = 1.f / (1.f + expf(-(lpc + lnc)));
  const float gp = sr * expf(tp - lpr) + scl * expf(tp - lpc);
  const float gn = sr * expf(tn - lnr) + scl * expf(tn - lnc);
  // d loss / d d_ij = (log_scale pw gp - log_scale nw gn) / log_scale / (B n)
  const float g = a.gscale * (pw * gp - nw * gn) / (float)((int64_t)a.B * a.n);
  a.dmat[e] = g / fmaxf(d, 1e-12f);
}

__global__ __launch_bounds__(64) void circle_feat_kernel(const CircleArgs a) {
  // blockIdx.z = 0: d pts_i = sum_j G_ij (pts_i - pix_j);  1: d pix_j = - sum_i G_ij (pts_i - pix_j).  Eight gathered rows (and their
  // G entries) in flight per trip, accumulated in index order as before.
  const int idx = blockIdx.x, b = blockIdx.y, side = blockIdx.z, lane = threadIdx.x;
  const int64_t* xi = a.xy_int + (int64_t)b * 2 * a.n;
  const int64_t* pi = a.pc_idx + (int64_t)b * a.n;
  auto pts = [&](int i) { return a.pc_feat[((int64_t)b * a.N + pi[i]) * 64 + lane]; };
  auto pix = [&](int j) { return a.img_feat[(((int64_t)b * a.h + xi[a.n + j]) * a.w + xi[j]) * 64 + lane]; };
  constexpr int U = 8;
  float acc = 0.f;
  if (side == 0) {
    const float me = pts(idx);
    const float* grow = a.dmat + ((int64_t)b * a.n + idx) * a.n;
    int j = 0;
    for (; j + U <= a.n; j += U) {
      float g[U], v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) { g[u] = grow[j + u]; v[u] = pix(j + u); }
#pragma unroll
      for (int u = 0; u < U; ++u) acc += g[u] * (me - v[u]);
    }
    for (; j < a.n; ++j) acc += grow[j] * (me - pix(j));
    a.dpts[((int64_t)b * a.n + idx) * 64 + lane] = acc;
  } else {
    const float me = pix(idx);
    const float* gcol = a.dmat + (int64_t)b * a.n * a.n + idx;
    int i = 0;
    for (; i + U <= a.n; i += U) {
      float g[U], v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) { g[u] = gcol[(int64_t)(i + u) * a.n]; v[u] = pts(i + u); }
#pragma unroll
      for (int u = 0; u < U; ++u) acc -= g[u] * (v[u] - me);
    }
    for (; i < a.n; ++i) acc -= gcol[(int64_t)i * a.n] * (pts(i) - me);
    a.dpix[((int64_t)b * a.n + idx) * 64 + lane] = acc;
  }
}

__global__ __launch_bounds__(64) void circle_scatter_kernel(const CircleArgs a, float* __restrict__ d_pc, float* __restrict__ d_img) {
  // Samples may repeat an index, and the additions into one row have to happen in sample order (deterministic, = the sequential loop this
  // replaces: one wave per (batch, side) walking 512 read-modify-writes).  One wave per (sample k, batch, side): the wave of the FIRST
  // occurrence of a row adds every sample of that row, in sample order; the others have nothing to do.
  const int k = blockIdx.x, b = blockIdx.y, side = blockIdx.z, lane = threadIdx.x;
  const int64_t* xi = a.xy_int + (int64_t)b * 2 * a.n;
  const int64_t* pi = a.pc_idx + (int64_t)b * a.n;
  auto key = [&](int q) -> int64_t { return side == 0 ? pi[q] : xi[a.n + q] * a.w + xi[q]; };
  const int64_t mine = key(k);
  // any earlier sample with the same row?  (lanes split the earlier samples)
  int earlier = 0;
  for (int q = lane; q < k; q += 64) earlier |= key(q) == mine ? 1 : 0;
  if (__any(earlier)) return;
  float* dst = side == 0 ? d_pc + ((int64_t)b * a.N + mine) * 64 + lane : d_img + ((int64_t)b * a.h * a.w + mine) * 64 + lane;
  const float* src = (side == 0 ? a.dpts : a.dpix) + (int64_t)b * a.n * 64 + lane;
  float v = *dst + src[(int64_t)k * 64];
  for (int q0 = k + 1; q0 < a.n; q0 += 64) {            // later samples of the same row, in order
    const int q = q0 + lane;
    const bool same = q < a.n && key(q) == mine;
    uint64_t mask = __ballot(same);
    while (mask) {
      const int l = __builtin_ctzll(mask);
      mask &= mask - 1;
      v += src[(int64_t)(q0 + l) * 64];
    }
  }
  *dst = v;
}

}  // namespace

extern "C" int cmr_axpy_f32(float* y, int64_t ldy, const float* x, int64_t ldx, float alpha, int64_t rows, int C, hipStream_t stream) {
  CMR_REQUIRE(y && x && rows >= 0 && C > 0 && C % 4 == 0 && ldy % 4 == 0 && ldx % 4 == 0 && cmr_aligned16(y) && cmr_aligned16(x));
  if (rows == 0) return CMR_OK;
  hipLaunchKernelGGL(axpy_kernel, dim3(ew_grid(rows * (C / 4))), dim3(256), 0, stream, y, ldy, x, ldx, alpha, rows, C);
  return cmr_launch_status();
}

extern "C" int cmr_act_f32(const float* x, int64_t ldx, float* y, int64_t ldy, int64_t rows, int C, int act, float act_param,
                           hipStream_t stream) {
  CMR_REQUIRE(x && y && rows >= 0 && C > 0 && C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && cmr_aligned16(x) && cmr_aligned16(y));
  CMR_REQUIRE(act >= CMR_ACT_NONE && act <= CMR_ACT_ELU1);
  if (rows == 0) return CMR_OK;
  hipLaunchKernelGGL(act_fwd_kernel, dim3(ew_grid(rows * (C / 4))), dim3(256), 0, stream, x, ldx, y, ldy, rows, C, act, act_param);
  return cmr_launch_status();
}

extern "C" int cmr_act_bwd_x_f32(const float* dy, int64_t lddy, const float* x, int64_t ldx, float* dx, int64_t lddx, int64_t rows, int C,
                                 int act, float act_param, int accumulate, hipStream_t stream) {
  CMR_REQUIRE(dy && x && dx && rows >= 0 && C > 0 && C % 4 == 0 && lddy % 4 == 0 && ldx % 4 == 0 && lddx % 4 == 0);
  CMR_REQUIRE(cmr_aligned16(dy) && cmr_aligned16(x) && cmr_aligned16(dx) && act >= CMR_ACT_NONE && act <= CMR_ACT_ELU1);
  if (rows == 0) return CMR_OK;
  hipLaunchKernelGGL(act_bwd_kernel, dim3(ew_grid(rows * (C / 4))), dim3(256), 0, stream, dy, lddy, x, ldx, dx, lddx, rows, C, act, act_param,
                     accumulate);
  return cmr_launch_status();
}

static inline int ln_blocks(int64_t rows) {
  int64_t nb = (rows + 31) / 32;                 // 16 rows per pass of a workgroup: two passes, so that 640 token rows still fill 20 CUs
  return (int)(nb > 1024 ? 1024 : (nb < 1 ? 1 : nb));
}

extern "C" int64_t cmr_layernorm64_bwd_workspace_bytes(int64_t rows) { return (int64_t)ln_blocks(rows) * 128 * sizeof(float); }

extern "C" int cmr_layernorm64_bwd_f32(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma, float eps, float* dx,
                                       int64_t lddx, int accumulate_dx, float* dgamma, float* dbeta, int accumulate_params, int64_t rows,
                                       void* ws, int64_t ws_bytes, hipStream_t stream) {
  CMR_REQUIRE(dy && x && gamma && dx && dgamma && dbeta && ws && rows > 0 && lddy % 4 == 0 && ldx % 4 == 0 && lddx % 4 == 0);
  CMR_REQUIRE(cmr_aligned16(dy) && cmr_aligned16(x) && cmr_aligned16(dx) && cmr_aligned16(gamma));
  const int nb = ln_blocks(rows);
  CMR_REQUIRE(ws_bytes >= (int64_t)nb * 128 * (int64_t)sizeof(float));
  hipLaunchKernelGGL(ln64_bwd_kernel, dim3(nb), dim3(256), 0, stream, dy, lddy, x, ldx, gamma, eps, dx, lddx, accumulate_dx, rows, (float*)ws);
  // part[blk][0][64] -> dgamma, part[blk][1][64] -> dbeta
  hipLaunchKernelGGL(reduce_partials2_kernel, dim3(128), dim3(64), 0, stream, (const float*)ws, nb, 64, dgamma, dbeta, accumulate_params);
  return cmr_launch_status();
}

extern "C" int cmr_l2norm64_bwd_f32(const float* dy, int64_t lddy, const float* x, int64_t ldx, float* dx, int64_t lddx, int accumulate,
                                    int64_t rows, hipStream_t stream) {
  CMR_REQUIRE(dy && x && dx && rows > 0 && lddy % 4 == 0 && ldx % 4 == 0 && lddx % 4 == 0 && cmr_aligned16(dy) && cmr_aligned16(x) &&
              cmr_aligned16(dx));
  hipLaunchKernelGGL(l2norm64_bwd_kernel, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, stream, dy, lddy, x, ldx, dx, lddx, accumulate, rows);
  return cmr_launch_status();
}

extern "C" int cmr_zero_insert2_f32(const float* g, float* out, int B, int Ho, int Wo, int H, int W, int C, hipStream_t stream) {
  CMR_REQUIRE(g && out && B > 0 && Ho > 0 && Wo > 0 && H >= 2 * Ho - 1 && W >= 2 * Wo - 1 && C % 4 == 0 && cmr_aligned16(g) && cmr_aligned16(out));
  hipLaunchKernelGGL(zero_insert2_kernel, dim3(ew_grid((int64_t)B * H * W * (C / 4))), dim3(256), 0, stream, g, out, B, Ho, Wo, H, W, C);
  return cmr_launch_status();
}

extern "C" int cmr_patchify_bwd_f32(const float* dpatches, float* dx, int B, int H, int W, int C, int P, int accumulate, hipStream_t stream) {
  CMR_REQUIRE(dpatches && dx && B > 0 && C % 4 == 0 && P > 0 && H % P == 0 && W % P == 0);
  hipLaunchKernelGGL(patchify_bwd_kernel, dim3(ew_grid((int64_t)B * H * W * (C / 4))), dim3(256), 0, stream, dpatches, dx, B, H, W, C, P, accumulate);
  return cmr_launch_status();
}

extern "C" int cmr_upsample_bwd_f32(const float* dcat, int64_t ldc, int coff, float* dproxy, int B, int H, int W, int C2, int scale,
                                    int accumulate, hipStream_t stream) {
  CMR_REQUIRE(dcat && dproxy && B > 0 && C2 % 4 == 0 && coff % 4 == 0 && ldc % 4 == 0 && scale > 0 && H % scale == 0 && W % scale == 0);
  hipLaunchKernelGGL(upsample_bwd_kernel, dim3(ew_grid((int64_t)B * (H / scale) * (W / scale) * (C2 / 4))), dim3(256), 0, stream, dcat, ldc,
                     coff, dproxy, B, H, W, C2, scale, accumulate);
  return cmr_launch_status();
}

extern "C" int cmr_im2col3_f32(const float* x4, float* cols, int B, int H, int W, hipStream_t stream) {
  CMR_REQUIRE(x4 && cols && B > 0 && H > 0 && W > 0 && cmr_aligned16(x4) && cmr_aligned16(cols));
  hipLaunchKernelGGL(im2col3_kernel, dim3(ew_grid((int64_t)B * H * W * 9)), dim3(256), 0, stream, x4, cols, B, H, W);
  return cmr_launch_status();
}

extern "C" int cmr_col2im3_f32(const float* dcols, float* dx4, int B, int H, int W, int accumulate, hipStream_t stream) {
  CMR_REQUIRE(dcols && dx4 && B > 0 && H > 0 && W > 0 && cmr_aligned16(dcols) && cmr_aligned16(dx4));
  hipLaunchKernelGGL(col2im3_kernel, dim3(ew_grid((int64_t)B * H * W)), dim3(256), 0, stream, dcols, dx4, B, H, W, accumulate);
  return cmr_launch_status();
}

template <bool DROP>
static int mha_bwd_launch(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, const float* o, int64_t ldo,
                          const float* dout, int64_t lddo, float* dq, int64_t lddq, int acc_dq, float* dk, int64_t lddk, int acc_dk, float* dv,
                          int64_t lddv, int acc_dv, float* ws, int64_t ws_bytes, int B, int Tq, int Tk, float p, const int64_t* seed,
                          int64_t site, hipStream_t stream) {
  CMR_REQUIRE(q && k && v && o && dout && dq && dk && dv && ws && B > 0 && B <= 65535 && Tq > 0 && Tk > 0);
  CMR_REQUIRE(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && lddo % 4 == 0 && cmr_aligned16(q) && cmr_aligned16(k) && cmr_aligned16(v) &&
              cmr_aligned16(dout));
  CMR_REQUIRE(ws_bytes >= (int64_t)B * Tq * 16 * (int64_t)sizeof(float));
  float* lse = ws;
  float* dsum = ws + (int64_t)B * Tq * 8;
  const size_t sm1 = (size_t)Tk * 16 * sizeof(float), sm2 = (size_t)Tq * 18 * sizeof(float);
  CMR_REQUIRE(sm1 <= 160 * 1024 && sm2 <= 160 * 1024);
  static CmrSmemCache g1{}, g2{};
  if (cmr_grant_smem(reinterpret_cast<const void*>(mha_bwd_dq_kernel<DROP>), sm1, g1) != CMR_OK) return CMR_ELAUNCH;
  if (cmr_grant_smem(reinterpret_cast<const void*>(mha_bwd_dkv_kernel<DROP>), sm2, g2) != CMR_OK) return CMR_ELAUNCH;
  const float scale = 0.35355339059327373f;
  const uint32_t thr = cmr_drop_threshold(p);
  const float ks = 1.f / (1.f - p);
  hipLaunchKernelGGL(mha_bwd_dq_kernel<DROP>, dim3((Tq + 63) / 64, AH_NH, B), dim3(256), sm1, stream, q, ldq, k, ldk, v, ldv, o, ldo, dout,
                     lddo, dq, lddq, acc_dq, lse, dsum, Tq, Tk, scale, thr, ks, seed, (uint64_t)site);
  hipLaunchKernelGGL(mha_bwd_dkv_kernel<DROP>, dim3((Tk + 63) / 64, AH_NH, B), dim3(256), sm2, stream, q, ldq, k, ldk, v, ldv, dout, lddo,
                     (const float*)lse, (const float*)dsum, dk, lddk, acc_dk, dv, lddv, acc_dv, Tq, Tk, scale, thr, ks, seed, (uint64_t)site);
  return cmr_launch_status();
}

extern "C" int cmr_mha_bwd_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, const float* o, int64_t ldo,
                               const float* dout, int64_t lddo, float* dq, int64_t lddq, int acc_dq, float* dk, int64_t lddk, int acc_dk,
                               float* dv, int64_t lddv, int acc_dv, float* ws, int64_t ws_bytes, int B, int Tq, int Tk, hipStream_t stream) {
  return mha_bwd_launch<false>(q, ldq, k, ldk, v, ldv, o, ldo, dout, lddo, dq, lddq, acc_dq, dk, lddk, acc_dk, dv, lddv, acc_dv, ws, ws_bytes, B,
                               Tq, Tk, 0.f, nullptr, 0, stream);
}

extern "C" int cmr_mha_dropout_bwd_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, const float* o,
                                       int64_t ldo, const float* dout, int64_t lddo, float* dq, int64_t lddq, int acc_dq, float* dk,
                                       int64_t lddk, int acc_dk, float* dv, int64_t lddv, int acc_dv, float* ws, int64_t ws_bytes, int B, int Tq,
                                       int Tk, float p, const int64_t* seed, int64_t site, hipStream_t stream) {
  CMR_REQUIRE(seed && p >= 0.f && p < 1.f);
  return mha_bwd_launch<true>(q, ldq, k, ldk, v, ldv, o, ldo, dout, lddo, dq, lddq, acc_dq, dk, lddk, acc_dk, dv, lddv, acc_dv, ws, ws_bytes, B,
                              Tq, Tk, p, seed, site, stream);
}

extern "C" int cmr_dropout_f32(const float* x, int64_t ldx, float* y, int64_t ldy, int64_t rows, int C, float p, const int64_t* seed,
                               int64_t site, hipStream_t stream) {
  CMR_REQUIRE(x && y && seed && rows > 0 && C > 0 && C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && p >= 0.f && p < 1.f);
  CMR_REQUIRE(cmr_aligned16(x) && cmr_aligned16(y));
  const int64_t pieces = rows * (C / 4);
  CMR_REQUIRE((pieces + 255) / 256 < 0x7fffffff);
  hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, stream, x, ldx, y, ldy, rows, C, cmr_drop_threshold(p),
                     1.f / (1.f - p), seed, (uint64_t)site);
  return cmr_launch_status();
}

static inline int la_tokens_per_wave(int L) { return L >= 65536 ? 64 : (L >= 8192 ? 16 : 4); }

extern "C" int64_t cmr_la_bwd_workspace_bytes(int B, int L) {
  const int tpw = la_tokens_per_wave(L);
  const int nblk = (L + 4 * tpw - 1) / (4 * tpw);
  return ((int64_t)B * nblk * 4 + B) * 576 * sizeof(float);
}

extern "C" int cmr_la_bwd_f32(const float* qf, int64_t ldq, const float* kf, int64_t ldk, const float* v, int64_t ldv, const float* kvsum,
                              const float* dmsg, int64_t lddm, float* dqf, int64_t lddq, int acc_dq, float* dkf, int64_t lddk, int acc_dk,
                              float* dv, int64_t lddv, int acc_dv, void* ws, int64_t ws_bytes, int B, int L, int S, float eps,
                              hipStream_t stream) {
  CMR_REQUIRE(qf && kf && v && kvsum && dmsg && dqf && dkf && dv && ws && B > 0 && B <= 65535 && L > 0 && S > 0);
  CMR_REQUIRE(ws_bytes >= cmr_la_bwd_workspace_bytes(B, L));
  const int tpw = la_tokens_per_wave(L);
  const int nblk = (L + 4 * tpw - 1) / (4 * tpw);
  float* part = (float*)ws;
  float* dstate = part + (int64_t)B * nblk * 4 * 576;
  hipLaunchKernelGGL(la_bwd_query_kernel, dim3(nblk, B), dim3(256), 0, stream, qf, ldq, kvsum, dmsg, lddm, dqf, lddq, acc_dq, part, L, S, eps, tpw);
  hipLaunchKernelGGL(la_bwd_state_kernel, dim3(576, B), dim3(64), 0, stream, (const float*)part, nblk * 4, dstate);
  const int tps = la_tokens_per_wave(S);
  hipLaunchKernelGGL(la_bwd_source_kernel, dim3((S + 4 * tps - 1) / (4 * tps), B), dim3(256), 0, stream, kf, ldk, v, ldv, (const float*)dstate,
                     dkf, lddk, acc_dk, dv, lddv, acc_dv, S, tps);
  return cmr_launch_status();
}

extern "C" int cmr_segment_softmax_bwd_f32(const float* attn, const float* vp, const int32_t* order, const int32_t* offsets, int fixed_len,
                                           float scale, const float* dout, float* dattn, float* dvp, int64_t nseg, hipStream_t stream) {
  CMR_REQUIRE(attn && vp && dout && dattn && dvp && nseg > 0 && (offsets || fixed_len > 0));
  hipLaunchKernelGGL(segment_softmax_bwd_kernel, dim3((unsigned)((nseg + 3) / 4)), dim3(256), 0, stream, attn, vp, order, offsets, fixed_len,
                     scale, dout, dattn, dvp, nseg);
  return cmr_launch_status();
}

extern "C" int cmr_focal_bwd_f32(const float* logits, int64_t ld, const int64_t* label, float alpha, int64_t rows, float grad_scale, float* dlogits,
                                 int64_t ldd, hipStream_t stream) {
  CMR_REQUIRE(logits && label && dlogits && rows > 0 && ld >= 2 && ldd >= 2);
  hipLaunchKernelGGL(focal_bwd_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, stream, logits, ld, label, alpha, rows, grad_scale,
                     dlogits, ldd);
  return cmr_launch_status();
}

extern "C" int64_t cmr_circle_bwd_workspace_bytes(int B, int n) { return ((int64_t)B * n * n + (int64_t)B * 4 * n + 2 * (int64_t)B * n * 64) * sizeof(float); }

extern "C" int cmr_circle_loss_bwd_f32(const float* pc_feat, const float* img_feat, const int64_t* pc_idx, const int64_t* xy_int,
                                       const float* xy_float, int B, int N, int h, int w, int n, float dist_thres, float pos_margin,
                                       float neg_margin, float log_scale, float grad_scale, float* d_pc_feat, float* d_img_feat, void* ws,
                                       int64_t ws_bytes, hipStream_t stream) {
  CMR_REQUIRE(pc_feat && img_feat && pc_idx && xy_int && xy_float && d_pc_feat && d_img_feat && ws && B > 0 && B <= 65535 && n > 0 && n <= 65535);
  CMR_REQUIRE(ws_bytes >= cmr_circle_bwd_workspace_bytes(B, n));
  float* f = (float*)ws;
  CircleArgs a{pc_feat, img_feat, pc_idx, xy_int, xy_float, B, N, h, w, n, dist_thres, pos_margin, neg_margin, log_scale, grad_scale,
               f, f + (int64_t)B * n * n, f + (int64_t)B * n * n + (int64_t)B * 4 * n, f + (int64_t)B * n * n + (int64_t)B * 4 * n + (int64_t)B * n * 64};
  hipLaunchKernelGGL(circle_dist_kernel, dim3(n, B), dim3(64), 0, stream, a);
  hipLaunchKernelGGL(circle_lse_kernel, dim3(n, B, 2), dim3(64), 0, stream, a);
  hipLaunchKernelGGL(circle_weight_kernel, dim3((unsigned)(((int64_t)B * n * n + 255) / 256)), dim3(256), 0, stream, a);
  hipLaunchKernelGGL(circle_feat_kernel, dim3(n, B, 2), dim3(64), 0, stream, a);
  hipLaunchKernelGGL(circle_scatter_kernel, dim3(n, B, 2), dim3(64), 0, stream, a, d_pc_feat, d_img_feat);
  return cmr_launch_status();
}
