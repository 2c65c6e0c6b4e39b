// Training-side kernels of the agent update (reference Train_Agent.py:263-305 = forward of CMRAgent in
// train() mode, loss.backward(), Adam.step()).  Everything here is HBM- or latency-bound streaming work over row maps
// [rows, C] (channels-last): batch-statistics BatchNorm forward / backward, activation and pooling backward, per-batch
// column reductions with arg-max, the small-batch linear backward of the heads, the BC + PPO loss with its gradient
// w.r.t. the logits, and the fused Adam step over the flat parameter bucket.  The MFMA-class backward contractions
// (conv3x3 wgrad, linear wgrad) live in wgrad.hip; data-gradient contractions reuse the forward kernels with
// transposed weights (cmr_pack_conv3x3_f32).
//
// Reductions are two-stage and deterministic: fixed partition of the rows over workgroups, fp32 partials, partials
// combined in double in a fixed order (so a result never depends on scheduling, and the data-parallel ranks of
// utils/flatbucket.py stay bit-identical after the all-reduce).
#include "cmr_common.h"

namespace {

constexpr int RED_THREADS = 256;
constexpr int MAX_C = 256;

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

inline int red_blocks(int64_t rows, int C) {
  // enough workgroups to fill 256 CUs twice, at least 64 rows per row-group pass
  const int rg = RED_THREADS / (C / 4);
  int64_t nb = (rows + (int64_t)rg * 16 - 1) / ((int64_t)rg * 16);
  if (nb > 512) nb = 512;
  if (nb < 1) nb = 1;
  return (int)nb;
}

// ------------------------------------------------------------------------------------------------------------------
// BatchNorm, training mode: statistics
// ------------------------------------------------------------------------------------------------------------------
// part[blk][0][c] = sum (x - pivot_c), part[blk][1][c] = sum (x - pivot_c)^2 over the block's rows; pivot = row 0 (keeps the
// E[x^2] - E[x]^2 cancellation mild whatever the channel mean is).
__global__ __launch_bounds__(RED_THREADS) void bn_stats_partial_kernel(const float* __restrict__ x, int64_t ldx, int64_t rows,
                                                                       int C, float* __restrict__ part) {
  __shared__ float sm[2 * 4 * RED_THREADS];   // RG * 2 * C = (256 / (C/4)) * 2 * C = 2048 floats whatever C is
  const int q = C >> 2, tid = threadIdx.x;
  const int cq = tid % q, rg = tid / q, RG = RED_THREADS / q;
  const int64_t per = (rows + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * per, r1 = r0 + per < rows ? r0 + per : rows;
  const f32x4 pivot = ld4(x + 4 * cq);
  f32x4 s = {0.f, 0.f, 0.f, 0.f}, ss = {0.f, 0.f, 0.f, 0.f};
  const bool on = rg < RG;                                 // widths whose quad count does not divide 256 (24, 48, ...): the tail threads idle
  int64_t r = on ? r0 + rg : r1;
  for (; r + 7 * RG < r1; r += 8 * RG) {                 // eight independent row loads in flight per thread, summed in row order
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = ld4(x + (r + (int64_t)u * RG) * ldx + 4 * cq);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const f32x4 d = v[u] - pivot;
      s += d;
      ss += d * d;
    }
  }
  for (; r < r1; r += RG) {
    const f32x4 v = ld4(x + r * ldx + 4 * cq) - pivot;
    s += v;
    ss += v * v;
  }
  if (on) {
    st4(&sm[(rg * 2 + 0) * C + 4 * cq], s);
    st4(&sm[(rg * 2 + 1) * C + 4 * cq], ss);
  }
  __syncthreads();
  for (int i = tid; i < 2 * C; i += RED_THREADS) {
    float a = 0.f;
    for (int g = 0; g < RG; ++g) a += sm[g * 2 * C + i];
    part[(int64_t)blockIdx.x * 2 * C + i] = a;
  }
}

// sum over a wave of doubles, same value in every lane afterwards (fixed butterfly order: deterministic)
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// one wave per channel: lane l adds partials l, l + 64, ... in double, then a butterfly over the wave
__global__ __launch_bounds__(64) void bn_stats_final_kernel(const float* __restrict__ x, const float* __restrict__ part, int nblk, int64_t rows,
                                                            int C, float eps, float momentum, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ running_mean,
                                                            float* __restrict__ running_var, float* __restrict__ stat) {
  const int c = blockIdx.x, lane = threadIdx.x;
  double s = 0.0, ss = 0.0;
  for (int b = lane; b < nblk; b += 64) {
    s += (double)part[(int64_t)b * 2 * C + c];
    ss += (double)part[(int64_t)b * 2 * C + C + c];
  }
  s = wave_sum(s);
  ss = wave_sum(ss);
  if (lane != 0) return;
  const double n = (double)rows;
  const double pm = s / n;
  double var = ss / n - pm * pm;
  var = var > 0.0 ? var : 0.0;
  const double mean = (double)x[c] + pm;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
  const float scale = g * rstd;
  stat[c] = (float)mean;
  stat[C + c] = rstd;
  stat[2 * C + c] = scale;
  stat[3 * C + c] = bt - (float)mean * scale;
  if (running_mean) {
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    const double unbiased = rows > 1 ? var * n / (n - 1.0) : var;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

// y = lrelu(x * scale + shift + [res * rscale + rshift | res]).  scale / shift null -> identity on x.
__global__ __launch_bounds__(256) void affine_act_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, const float* __restrict__ res, int64_t ldres,
                                                         const float* __restrict__ rscale, const float* __restrict__ rshift,
                                                         float* __restrict__ y, int64_t ldy, int64_t rows, int C, float slope) {
  const int q = C >> 2;
  const int64_t total = rows * q;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / q;
    const int c = (int)(i - r * q) * 4;
    f32x4 v = ld4(x + r * ldx + c);
    if (scale) {
      // explicitly fused: the lazily evaluated layers (csrc/bn_linear.hip) recompute this pre-activation and must land on the same side of zero
      const f32x4 sc = ld4(scale + c), sh = ld4(shift + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(v[e], sc[e], sh[e]);
    }
    if (res) {
      f32x4 t = ld4(res + r * ldres + c);
      if (rscale) t = t * ld4(rscale + c) + ld4(rshift + c);
      v += t;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * slope;
    st4(y + r * ldy + c, v);
  }
}

// The elementwise glue of a vector-attention layer (models/PointNN.py:151-170, 219-226) in one pass each way:
//   forward   a_in = q - k + pos,  vp = v + pos            (three torch additions: 9 map passes -> 6)
//   backward  dk = -da_in,  dpos = da_in + dvp             (dq = da_in and dv = dvp are the incoming buffers themselves)
__global__ __launch_bounds__(256) void vecattn_mix_kernel(const float* __restrict__ q, int64_t ldq, const float* __restrict__ k, int64_t ldk,
                                                          const float* __restrict__ v, int64_t ldv, const float* __restrict__ pos, int64_t ldp,
                                                          float* __restrict__ a_in, float* __restrict__ vp, int64_t rows, int C) {
  const int qd = C >> 2;
  const int64_t total = rows * qd;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / qd;
    const int c = (int)(i - r * qd) * 4;
    const f32x4 p = ld4(pos + r * ldp + c);
    st4(a_in + r * C + c, ld4(q + r * ldq + c) - ld4(k + r * ldk + c) + p);
    st4(vp + r * C + c, ld4(v + r * ldv + c) + p);
  }
}

__global__ __launch_bounds__(256) void vecattn_mix_bwd_kernel(const float* __restrict__ da, int64_t ldda, const float* __restrict__ dvp, int64_t lddv,
                                                              float* __restrict__ dk, float* __restrict__ dpos, int64_t rows, int C) {
  const int qd = C >> 2;
  const int64_t total = rows * qd;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / qd;
    const int c = (int)(i - r * qd) * 4;
    const f32x4 a = ld4(da + r * ldda + c);
    st4(dk + r * C + c, f32x4{0.f, 0.f, 0.f, 0.f} - a);
    st4(dpos + r * C + c, a + ld4(dvp + r * lddv + c));
  }
}

// ------------------------------------------------------------------------------------------------------------------
// BatchNorm, training mode: backward
// ------------------------------------------------------------------------------------------------------------------
// dy = dz * act'(z) (z = activation output, null -> dy = dz);  part[blk][0][c] = sum dy,  [1][c] = sum dy * xhat
__global__ __launch_bounds__(RED_THREADS) void bn_bwd_partial_kernel(const float* __restrict__ dz, int64_t lddz, const float* __restrict__ z,
                                                                     int64_t ldz, float slope, const float* __restrict__ x, int64_t ldx,
                                                                     const float* __restrict__ stat, int64_t rows, int C,
                                                                     float* __restrict__ part) {
  __shared__ float sm[2 * 4 * RED_THREADS];
  const int q = C >> 2, tid = threadIdx.x;
  const int cq = tid % q, rg = tid / q, RG = RED_THREADS / q;
  const int64_t per = (rows + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * per, r1 = r0 + per < rows ? r0 + per : rows;
  const f32x4 mean = ld4(stat + 4 * cq), rstd = ld4(stat + C + 4 * cq);
  // z null with an activation (slope != 1): the mask is the sign of the BatchNorm output itself, x * scale + shift with the SAME fused
  // multiply-add as affine_act_kernel (no residual in front of the activation) -- one map read less than reading the stored output
  const bool mx = !z && slope != 1.f;
  const f32x4 msc = ld4(stat + 2 * C + 4 * cq), msh = ld4(stat + 3 * C + 4 * cq);
  f32x4 s = {0.f, 0.f, 0.f, 0.f}, sx = {0.f, 0.f, 0.f, 0.f};
  const bool on = rg < RG;
  int64_t r = on ? r0 + rg : r1;
  for (; r + 3 * RG < r1; r += 4 * RG) {                 // four rows (12 loads) in flight per thread, summed in row order
    f32x4 dv[4], av[4], xv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t ru = r + (int64_t)u * RG;
      dv[u] = ld4(dz + ru * lddz + 4 * cq);
      av[u] = z ? ld4(z + ru * ldz + 4 * cq) : f32x4{1.f, 1.f, 1.f, 1.f};
      xv[u] = ld4(x + ru * ldx + 4 * cq);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      f32x4 d = dv[u];
      if (mx) {
#pragma unroll
        for (int e = 0; e < 4; ++e) av[u][e] = __builtin_fmaf(xv[u][e], msc[e], msh[e]);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) d[e] = av[u][e] > 0.f ? d[e] : d[e] * slope;
      const f32x4 xh = (xv[u] - mean) * rstd;
      s += d;
      sx += d * xh;
    }
  }
  for (; r < r1; r += RG) {
    f32x4 d = ld4(dz + r * lddz + 4 * cq);
    const f32x4 xr = ld4(x + r * ldx + 4 * cq);
    if (z || mx) {
      f32x4 a;
      if (z) a = ld4(z + r * ldz + 4 * cq);
      else {
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] = __builtin_fmaf(xr[e], msc[e], msh[e]);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) d[e] = a[e] > 0.f ? d[e] : d[e] * slope;
    }
    const f32x4 xh = (xr - mean) * rstd;
    s += d;
    sx += d * xh;
  }
  if (on) {
    st4(&sm[(rg * 2 + 0) * C + 4 * cq], s);
    st4(&sm[(rg * 2 + 1) * C + 4 * cq], sx);
  }
  __syncthreads();
  for (int i = tid; i < 2 * C; i += RED_THREADS) {
    float a = 0.f;
    for (int g = 0; g < RG; ++g) a += sm[g * 2 * C + i];
    part[(int64_t)blockIdx.x * 2 * C + i] = a;
  }
}

// coef[0][c] = sum dy / n, coef[1][c] = sum dy xhat / n;  dgamma / dbeta written (or accumulated) into the gradient bucket
__global__ __launch_bounds__(64) void bn_bwd_final_kernel(const float* __restrict__ part, int nblk, int64_t rows, int C, float* __restrict__ coef,
                                                          float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.x, lane = threadIdx.x;
  double s = 0.0, sx = 0.0;
  for (int b = lane; b < nblk; b += 64) {
    s += (double)part[(int64_t)b * 2 * C + c];
    sx += (double)part[(int64_t)b * 2 * C + C + c];
  }
  s = wave_sum(s);
  sx = wave_sum(sx);
  if (lane != 0) return;
  coef[c] = (float)(s / (double)rows);
  coef[C + c] = (float)(sx / (double)rows);
  if (dbeta) dbeta[c] = (float)s;
  if (dgamma) dgamma[c] = (float)sx;
}

// dx = scale * (dy - c1 - xhat * c2) (+ add),  scale = gamma * rstd = stat[2]
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dz, int64_t lddz, const float* __restrict__ z, int64_t ldz,
                                                           float slope, const float* __restrict__ x, int64_t ldx, const float* __restrict__ stat,
                                                           const float* __restrict__ coef, const float* __restrict__ add, int64_t ldadd,
                                                           float* __restrict__ dx, int64_t lddx, float* __restrict__ dzm, int64_t lddzm,
                                                           int64_t rows, int C) {
  const int q = C >> 2;
  const int64_t total = rows * q;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / q;
    const int c = (int)(i - r * q) * 4;
    f32x4 d = ld4(dz + r * lddz + c);
    const f32x4 xr = ld4(x + r * ldx + c);
    if (z) {
      const f32x4 a = ld4(z + r * ldz + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) d[e] = a[e] > 0.f ? d[e] : d[e] * slope;
    } else if (slope != 1.f) {                     // mask from the BatchNorm output's sign (see bn_bwd_partial_kernel)
      const f32x4 sc = ld4(stat + 2 * C + c), sh = ld4(stat + 3 * C + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) d[e] = __builtin_fmaf(xr[e], sc[e], sh[e]) > 0.f ? d[e] : d[e] * slope;
    }
    if (dzm) st4(dzm + r * lddzm + c, d);          // the gradient at the activation's input: what a residual branch added there receives
    const f32x4 xh = (xr - ld4(stat + c)) * ld4(stat + C + c);
    f32x4 g = ld4(stat + 2 * C + c) * (d - ld4(coef + c) - xh * ld4(coef + C + c));
    if (add) g += ld4(add + r * ldadd + c);
    st4(dx + r * lddx + c, g);
  }
}

// dy = dz * act'(z) (+ add): activation backward without a BatchNorm (identity shortcut of ConvBNReLURes1D)
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dz, int64_t lddz, const float* __restrict__ z, int64_t ldz,
                                                      float slope, const float* __restrict__ add, int64_t ldadd, float* __restrict__ dy,
                                                      int64_t lddy, int64_t rows, int C) {
  const int q = C >> 2;
  const int64_t total = rows * q;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / q;
    const int c = (int)(i - r * q) * 4;
    f32x4 d = ld4(dz + r * lddz + c);
    const f32x4 a = ld4(z + r * ldz + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) d[e] = a[e] > 0.f ? d[e] : d[e] * slope;
    if (add) d += ld4(add + r * ldadd + c);
    st4(dy + r * lddy + c, d);
  }
}

// AvgPool2d(ph, pw) backward fused with the backward of the activation that precedes the pool:
// dc[b,y,x,:] = g[b, y / ph, x / pw, :] / (ph pw) * act'(d[b,y,x,:])
__global__ __launch_bounds__(256) void pool_act_bwd_kernel(const float* __restrict__ g, const float* __restrict__ d, float* __restrict__ dc,
                                                           int B, int H, int W, int C, int ph, int pw, float slope) {
  const int q = C >> 2;
  const int64_t total = (int64_t)B * H * W * q;
  const int Hp = H / ph, Wp = W / pw;
  const float inv = 1.f / (float)(ph * pw);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % q) * 4;
    int64_t p = i / q;
    const int xx = (int)(p % W);
    p /= W;
    const int yy = (int)(p % H);
    const int b = (int)(p / H);
    const f32x4 gv = ld4(g + (((int64_t)b * Hp + yy / ph) * Wp + xx / pw) * C + c);
    const f32x4 a = ld4(d + (i / q) * C + c);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = gv[e] * inv * (a[e] > 0.f ? 1.f : slope);
    st4(dc + (i / q) * C + c, o);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// per-batch column reductions: sum, and max with arg-max (first index on ties, like torch.max(dim))
// ------------------------------------------------------------------------------------------------------------------
template <bool MAXARG>
__global__ __launch_bounds__(RED_THREADS) void col_partial_kernel(const float* __restrict__ x, int64_t ldx, int N, int C, float* __restrict__ pv,
                                                                  int32_t* __restrict__ pi) {
  __shared__ float sv[4 * RED_THREADS];             // RG * C = 1024 floats
  __shared__ int32_t si[MAXARG ? 4 * RED_THREADS : 1];
  const int q = C >> 2, tid = threadIdx.x;
  const int cq = tid % q, rg = tid / q, RG = RED_THREADS / q;
  const int b = blockIdx.y;
  const int per = (N + gridDim.x - 1) / gridDim.x;
  const int n0 = blockIdx.x * per, n1 = n0 + per < N ? n0 + per : N;
  f32x4 acc;
  int idx[4] = {0, 0, 0, 0};
#pragma unroll
  for (int e = 0; e < 4; ++e) acc[e] = MAXARG ? -INFINITY : 0.f;
  const bool on = rg < RG;
  for (int n = on ? n0 + rg : n1; n < n1; n += RG) {
    const f32x4 v = ld4(x + ((int64_t)b * N + n) * ldx + 4 * cq);
    if (MAXARG) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (v[e] > acc[e]) { acc[e] = v[e]; idx[e] = n; }
    } else {
      acc += v;
    }
  }
  if (on) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      sv[rg * C + 4 * cq + e] = acc[e];
      if (MAXARG) si[rg * C + 4 * cq + e] = idx[e];
    }
  }
  __syncthreads();
  for (int c = tid; c < C; c += RED_THREADS) {
    float a = sv[c];
    int ia = MAXARG ? si[c] : 0;
    for (int g = 1; g < RG; ++g) {
      const float v = sv[g * C + c];
      if (MAXARG) {
        const int iv = si[g * C + c];
        if (v > a || (v == a && iv < ia)) { a = v; ia = iv; }
      } else {
        a += v;
      }
    }
    const int64_t o = ((int64_t)b * gridDim.x + blockIdx.x) * C + c;
    pv[o] = a;
    if (MAXARG) pi[o] = ia;
  }
}

// one wave per (batch, channel): lane l folds partials l, l + 64, ..., then a butterfly over the wave
template <bool MAXARG>
__global__ __launch_bounds__(64) void col_final_kernel(const float* __restrict__ pv, const int32_t* __restrict__ pi, int nblk, int C,
                                                       float* __restrict__ out, int32_t* __restrict__ arg) {
  const int c = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
  if (MAXARG) {
    float a = -INFINITY;
    int ia = 0x7fffffff;
    for (int k = lane; k < nblk; k += 64) {
      const float v = pv[((int64_t)b * nblk + k) * C + c];
      const int iv = pi[((int64_t)b * nblk + k) * C + c];
      if (v > a || (v == a && iv < ia)) { a = v; ia = iv; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float v = __shfl_xor(a, o, 64);
      const int iv = __shfl_xor(ia, o, 64);
      if (v > a || (v == a && iv < ia)) { a = v; ia = iv; }
    }
    if (lane == 0) {
      out[(int64_t)b * C + c] = a;
      arg[(int64_t)b * C + c] = ia;
    }
  } else {
    double a = 0.0;
    for (int k = lane; k < nblk; k += 64) a += (double)pv[((int64_t)b * nblk + k) * C + c];
    a = wave_sum(a);
    if (lane == 0) out[(int64_t)b * C + c] = (float)a;
  }
}

// backward of the per-batch max: dx[b, arg[b,c], c] += g[b,c]
__global__ void add_at_arg_kernel(float* __restrict__ dx, int64_t lddx, const int32_t* __restrict__ arg, const float* __restrict__ g, int64_t ldg,
                                  int B, int N, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C, c = i - b * C;
  dx[((int64_t)b * N + arg[i]) * lddx + c] += g[(int64_t)b * ldg + c];
}

// ------------------------------------------------------------------------------------------------------------------
// linear layer backward on a handful of rows (the 1x1 convs after the global pool and the three heads: rows = minibatch)
//   dYe = dY * act'(Y);  dW[n][k] = sum_r dYe[r][n] X[r][k];  db[n] = sum_r dYe[r][n];  dX[r][k] (+)= sum_n dYe[r][n] W[n][k]
// X = [X1 | X2] (the heads read cat([embed_2d, embed_3d]))
// ------------------------------------------------------------------------------------------------------------------
struct SmallBwd {
  const float *x1, *x2, *y, *dy, *w;
  float *dw, *db, *dx1, *dx2;
  int64_t ldx1, ldx2, ldy, lddy, ldw, lddw, lddx1, lddx2;
  int rows, n, k1, k2, acc_dx;
  float slope;
};

__global__ __launch_bounds__(256) void linear_bwd_small_kernel(const SmallBwd a) {
  const int K = a.k1 + a.k2;
  const int64_t nw = (int64_t)a.n * K;
  const int64_t nx = (a.dx1 || a.dx2) ? (int64_t)a.rows * K : 0;
  const int64_t total = nw + a.n + nx;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    if (i < nw + a.n) {
      const bool isb = i >= nw;
      const int n = isb ? (int)(i - nw) : (int)(i / K);
      const int k = isb ? 0 : (int)(i - (int64_t)n * K);
      float s = 0.f;
      for (int r = 0; r < a.rows; ++r) {
        float d = a.dy[(int64_t)r * a.lddy + n];
        if (a.y) d = a.y[(int64_t)r * a.ldy + n] > 0.f ? d : d * a.slope;
        const float xv = isb ? 1.f : (k < a.k1 ? a.x1[(int64_t)r * a.ldx1 + k] : a.x2[(int64_t)r * a.ldx2 + k - a.k1]);
        s += d * xv;
      }
      if (isb) {
        if (a.db) a.db[n] = s;
      } else if (a.dw) {
        a.dw[(int64_t)n * a.lddw + k] = s;
      }
    } else {
      const int64_t j = i - nw - a.n;
      const int r = (int)(j / K), k = (int)(j - (int64_t)r * K);
      // 8 weight rows in flight per thread (independent loads, then the multiplies in the SAME order as a one-by-one loop): the
      // one-load-per-iteration loop was a chain of up to 256 memory round trips per thread -- 44 us per call for a 10-row head layer,
      // 16 calls per update
      float s = 0.f;
      int n = 0;
      for (; n + 8 <= a.n; n += 8) {
        float d[8], wv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          d[u] = a.dy[(int64_t)r * a.lddy + n + u];
          wv[u] = a.w[(int64_t)(n + u) * a.ldw + k];
        }
        if (a.y) {                                        // (uniform)
          float yv[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) yv[u] = a.y[(int64_t)r * a.ldy + n + u];
#pragma unroll
          for (int u = 0; u < 8; ++u) d[u] = yv[u] > 0.f ? d[u] : d[u] * a.slope;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) s = __builtin_fmaf(d[u], wv[u], s);       // fused, like the tail loop's contraction: the unrolled form
      }                                                                        // otherwise becomes packed multiplies + adds (other rounding)
      for (; n < a.n; ++n) {
        float d = a.dy[(int64_t)r * a.lddy + n];
        if (a.y) d = a.y[(int64_t)r * a.ldy + n] > 0.f ? d : d * a.slope;
        s += d * a.w[(int64_t)n * a.ldw + k];
      }
      float* dst = k < a.k1 ? (a.dx1 ? a.dx1 + (int64_t)r * a.lddx1 + k : nullptr) : (a.dx2 ? a.dx2 + (int64_t)r * a.lddx2 + k - a.k1 : nullptr);
      if (dst) *dst = a.acc_dx ? *dst + s : s;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// BC + PPO loss of one minibatch and its gradient w.r.t. the logits / value (Train_Agent.py:268-302, CMRAgent.py:129-144)
// one thread per (sample, degree of freedom) row of S logits; single workgroup (minibatch = 10 samples in the reference)
// ------------------------------------------------------------------------------------------------------------------
struct LossArgs {
  const float *r_logits, *t_logits, *value;
  int64_t ldr, ldt, ldv;
  const int64_t *expert_r, *expert_t, *act_r, *act_t;
  const float *old_logprob, *returns, *adv;
  float *d_r, *d_t, *d_v;
  int64_t lddr, lddt, lddv;
  float* out;     // [8]: loss, clone, policy, value, entropy, ppo, 0, 0
  int B, dr, dt, S;
  float alpha, clip_eps, w_value, w_entropy, grad_scale;
};

__global__ __launch_bounds__(256) void agent_loss_kernel(const LossArgs a) {
  __shared__ float red[5][256];
  const int tid = threadIdx.x;
  const int D = a.dr + a.dt;
  float ce_r = 0.f, ce_t = 0.f, pol = 0.f, ent = 0.f, val = 0.f;
  for (int i = tid; i < a.B * D; i += blockDim.x) {
    const int b = i / D, j = i - b * D;
    const bool isr = j < a.dr;
    const int jj = isr ? j : j - a.dr;
    const float* lg = isr ? a.r_logits + (int64_t)b * a.ldr + jj * a.S : a.t_logits + (int64_t)b * a.ldt + jj * a.S;
    float* dl = isr ? a.d_r + (int64_t)b * a.lddr + jj * a.S : a.d_t + (int64_t)b * a.lddt + jj * a.S;
    const int64_t ex = isr ? a.expert_r[(int64_t)b * a.dr + jj] : a.expert_t[(int64_t)b * a.dt + jj];
    const int64_t ac = isr ? a.act_r[(int64_t)b * a.dr + jj] : a.act_t[(int64_t)b * a.dt + jj];
    float m = -INFINITY;
    for (int s = 0; s < a.S; ++s) m = fmaxf(m, lg[s]);
    float sum = 0.f;
    for (int s = 0; s < a.S; ++s) sum += expf(lg[s] - m);
    const float lse = m + logf(sum);
    float H = 0.f;
    for (int s = 0; s < a.S; ++s) {
      const float lp = lg[s] - lse;
      H -= expf(lp) * lp;
    }
    const float lp_ex = lg[ex] - lse, lp_ac = lg[ac] - lse;
    const float n_ce = 1.f / (float)(a.B * (isr ? a.dr : a.dt));     // CrossEntropyLoss mean over the rows of its own call
    if (isr) ce_r += -lp_ex * n_ce; else ce_t += -lp_ex * n_ce;
    float w_lp = 0.f;                                                   // d loss / d logprob(action)
    const float n_pd = 1.f / (float)(a.B * D);
    if (a.alpha > 0.f) {
      const float A = a.adv[b];
      const float ratio = expf(lp_ac - a.old_logprob[(int64_t)b * D + j]);
      const float clipped = fminf(fmaxf(ratio, 1.f - a.clip_eps), 1.f + a.clip_eps);
      const float s1 = ratio * A, s2 = clipped * A;
      pol += -fminf(s1, s2) * n_pd;
      ent += H * n_pd;
      // torch.min(a, b) backward: gradient to the smaller one, split in halves on ties; clamp passes the gradient inside
      // [lo, hi] (bounds included)
      const bool inside = ratio >= 1.f - a.clip_eps && ratio <= 1.f + a.clip_eps;
      float ga = s1 < s2 ? 1.f : (s1 == s2 ? 0.5f : 0.f);
      float gb = s2 < s1 ? 1.f : (s1 == s2 ? 0.5f : 0.f);
      const float dratio = ga * A + (inside ? gb * A : 0.f);
      w_lp = -a.alpha * n_pd * dratio * ratio;
    }
    for (int s = 0; s < a.S; ++s) {
      const float lp = lg[s] - lse, p = expf(lp);
      float g = n_ce * (p - (s == ex ? 1.f : 0.f));                     // behaviour cloning
      g += w_lp * ((s == ac ? 1.f : 0.f) - p);                          // PPO policy term through log pi(action)
      if (a.alpha > 0.f) g += a.alpha * a.w_entropy * n_pd * p * (lp + H);   // - w_entropy * mean(H): dH/dz_s = -p (lp + H)
      dl[s] = g * a.grad_scale;
    }
  }
  for (int b = tid; b < a.B; b += blockDim.x) {
    const float v = a.value[(int64_t)b * a.ldv];
    if (a.alpha > 0.f) {
      const float d = v - a.returns[b];
      val += d * d / (float)a.B;
      a.d_v[(int64_t)b * a.lddv] = a.alpha * a.w_value * 2.f * d / (float)a.B * a.grad_scale;
    } else {
      a.d_v[(int64_t)b * a.lddv] = 0.f;
    }
  }
  red[0][tid] = ce_r; red[1][tid] = ce_t; red[2][tid] = pol; red[3][tid] = ent; red[4][tid] = val;
  __syncthreads();
  if (tid == 0) {
    float t[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < 5; ++k)
      for (int i = 0; i < (int)blockDim.x; ++i) t[k] += red[k][i];
    const float clone = t[0] + t[1];
    const float ppo = t[2] + t[4] * a.w_value - t[3] * a.w_entropy;
    a.out[0] = a.alpha > 0.f ? clone + ppo * a.alpha : clone;
    a.out[1] = clone; a.out[2] = t[2]; a.out[3] = t[4]; a.out[4] = t[3]; a.out[5] = ppo; a.out[6] = 0.f; a.out[7] = 0.f;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// fused Adam over the flat bucket (torch.optim.Adam semantics, Train_Agent.py:121-127: L2 weight decay added to the
// gradient, bias-corrected moments).  g is scaled by gscale first (1 / world size after the sum all-reduce).
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                   int64_t n4, float lr, float b1, float b2, float eps, float wd, float bc1, float sqrt_bc2,
                                                   float gscale, float clip) {
  const float step = lr / bc1;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    f32x4 pv = ld4(p + 4 * i), gv = ld4(g + 4 * i) * gscale, mv = ld4(m + 4 * i), vv = ld4(v + 4 * i);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float gr = gv[e];
      if (clip > 0.f) gr = fminf(fmaxf(gr, -clip), clip);      // nn.utils.clip_grad_value_ (Train_Geo.py:172), before weight decay
      const float gg = gr + wd * pv[e];
      mv[e] = mv[e] + (gg - mv[e]) * (1.f - b1);               // exp_avg.lerp_(grad, 1 - beta1)
      vv[e] = vv[e] * b2 + (1.f - b2) * gg * gg;               // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
      const float denom = sqrtf(vv[e]) / sqrt_bc2 + eps;        // (exp_avg_sq.sqrt() / sqrt(bias_correction2)).add_(eps)
      pv[e] = pv[e] - step * (mv[e] / denom);
    }
    st4(p + 4 * i, pv);
    st4(m + 4 * i, mv);
    st4(v + 4 * i, vv);
  }
}

// fused SGD with momentum over the flat bucket (torch.optim.SGD semantics, the 'SGD' branch of Train_Agent.py:111-117 /
// Train_Geo.py:65-71: dampening 0, no Nesterov): g += wd * p; buf = g on the first step, else momentum * buf + g; p -= lr * buf.
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, int64_t n4,
                                                  float lr, float momentum, float wd, float gscale, float clip, int first) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    f32x4 pv = ld4(p + 4 * i), gv = ld4(g + 4 * i) * gscale, bv = ld4(buf + 4 * i);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float gr = gv[e];
      if (clip > 0.f) gr = fminf(fmaxf(gr, -clip), clip);
      const float gg = gr + wd * pv[e];
      bv[e] = first ? gg : bv[e] * momentum + gg;               // buf.mul_(momentum).add_(grad)
      pv[e] = pv[e] - lr * bv[e];
    }
    st4(p + 4 * i, pv);
    st4(buf + 4 * i, bv);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Transposed shadow of the matrix parameters of a flat bucket: table row = (src offset, n, k, dst offset, first tile); one
// workgroup per 32 x 32 tile of one matrix: dst[k][n] = src[n][k].  One launch per optimizer step instead of one per layer.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_slots_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                              const int64_t* __restrict__ table, int nslots) {
  __shared__ float tile[32][33];
  const int64_t t = blockIdx.x;
  int lo = 0, hi = nslots - 1;                               // last slot whose first tile <= t
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (table[mid * 5 + 4] <= t) lo = mid; else hi = mid - 1;
  }
  const int64_t* e = table + lo * 5;
  const int n = (int)e[1], k = (int)e[2];
  const int tk = (k + 31) / 32;
  const int local = (int)(t - e[4]);
  const int r0 = (local / tk) * 32, c0 = (local % tk) * 32;
  const float* a = src + e[0];
  float* b = dst + e[3];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    tile[ty + 8 * i][tx] = (r < n && c < k) ? a[(int64_t)r * k + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, r = r0 + tx;
    if (c < k && r < n) b[(int64_t)c * n + r] = tile[tx][ty + 8 * i];
  }
}

inline unsigned ew_grid(int64_t items) {
  int64_t g = (items + 255) / 256;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (unsigned)g;
}

inline bool chan_ok(int C) { return C >= 4 && C <= 1024 && C % 4 == 0; }

}  // namespace

extern "C" int64_t cmr_bn_workspace_bytes(int64_t rows, int C) { return (int64_t)red_blocks(rows, C) * 2 * C * sizeof(float); }

extern "C" int cmr_bn_stats_f32(const float* x, int64_t ldx, int64_t rows, int C, float eps, float momentum, const float* gamma,
                                const float* beta, float* running_mean, float* running_var, float* stat, void* ws, int64_t ws_bytes,
                                hipStream_t stream) {
  CMR_REQUIRE(x && stat && ws && rows > 0 && chan_ok(C) && ldx % 4 == 0 && cmr_aligned16(x));
  CMR_REQUIRE((running_mean == nullptr) == (running_var == nullptr));
  const int nb = red_blocks(rows, C);
  CMR_REQUIRE(ws_bytes >= (int64_t)nb * 2 * C * (int64_t)sizeof(float));
  hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(nb), dim3(RED_THREADS), 0, stream, x, ldx, rows, C, (float*)ws);
  hipLaunchKernelGGL(bn_stats_final_kernel, dim3(C), dim3(64), 0, stream, x, (const float*)ws, nb, rows, C, eps, momentum,
                     gamma, beta, running_mean, running_var, stat);
  return cmr_launch_status();
}

// BatchNorm statistics from partial sums a producer left: part [parts][2][C] = sums of (x - pivot) and (x - pivot)^2 over disjoint row sets
// that cover all `rows` rows (cmr_conv3x3_wino_stats_nhwc_f32: pivot = the convolution's bias).  One wave per channel, double, fixed order;
// stat and the running statistics exactly as cmr_bn_stats_f32 leaves them.
static __global__ __launch_bounds__(64) void bn_stats_from_sums_kernel(const float* __restrict__ part, int64_t parts, int64_t rows, int C,
                                                                const float* __restrict__ pivot, float eps, float momentum,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                float* __restrict__ running_mean, float* __restrict__ running_var,
                                                                float* __restrict__ stat) {
  const int c = blockIdx.x, lane = threadIdx.x;
  double s = 0.0, ss = 0.0;
  for (int64_t b0 = lane; b0 < parts; b0 += 64 * 8) {
    float v0[8], v1[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t b = b0 + 64 * u < parts ? b0 + 64 * u : b0;
      v0[u] = part[b * 2 * C + c];
      v1[u] = part[b * 2 * C + C + c];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (b0 + 64 * u < parts) {
        s += (double)v0[u];
        ss += (double)v1[u];
      }
  }
  s = wave_sum(s);
  ss = wave_sum(ss);
  if (lane != 0) return;
  const double n = (double)rows;
  const double pm = s / n;
  double var = ss / n - pm * pm;
  var = var > 0.0 ? var : 0.0;
  const double mean = (pivot ? (double)pivot[c] : 0.0) + pm;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
  const float scale = g * rstd;
  stat[c] = (float)mean;
  stat[C + c] = rstd;
  stat[2 * C + c] = scale;
  stat[3 * C + c] = bt - (float)mean * scale;
  if (running_mean) {
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    const double unbiased = rows > 1 ? var * n / (n - 1.0) : var;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

extern "C" int cmr_bn_stats_from_sums_f32(const float* part, int64_t parts, int64_t rows, int C, const float* pivot, float eps, float momentum,
                                          const float* gamma, const float* beta, float* running_mean, float* running_var, float* stat,
                                          hipStream_t stream) {
  CMR_REQUIRE(part && stat && parts > 0 && rows > 0 && C > 0 && C <= 1024);
  CMR_REQUIRE((running_mean == nullptr) == (running_var == nullptr));
  hipLaunchKernelGGL(bn_stats_from_sums_kernel, dim3(C), dim3(64), 0, stream, part, parts, rows, C, pivot, eps, momentum, gamma, beta,
                     running_mean, running_var, stat);
  return cmr_launch_status();
}

extern "C" int cmr_affine_act_f32(const float* x, int64_t ldx, const float* scale, const float* shift, const float* res, int64_t ldres,
                                  const float* rscale, const float* rshift, float* y, int64_t ldy, int64_t rows, int C, float slope,
                                  hipStream_t stream) {
  CMR_REQUIRE(x && y && rows >= 0 && C > 0 && C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && cmr_aligned16(x) && cmr_aligned16(y));
  CMR_REQUIRE((scale == nullptr) == (shift == nullptr) && (rscale == nullptr) == (rshift == nullptr));
  if (res) CMR_REQUIRE(ldres % 4 == 0 && cmr_aligned16(res));
  if (rows == 0) return CMR_OK;
  hipLaunchKernelGGL(affine_act_kernel, dim3(ew_grid(rows * (C / 4))), dim3(256), 0, stream, x, ldx, scale, shift, res, ldres, rscale,
                     rshift, y, ldy, rows, C, slope);
  return cmr_launch_status();
}

extern "C" int cmr_bn_bwd_f32(const float* dz, int64_t lddz, const float* z, int64_t ldz, float slope, const float* x, int64_t ldx,
                              const float* stat, const float* add, int64_t ldadd, float* dx, int64_t lddx, float* dzm, int64_t lddzm,
                              float* dgamma, float* dbeta, int64_t rows, int C, void* ws, int64_t ws_bytes, hipStream_t stream) {
  if (dzm) CMR_REQUIRE(lddzm % 4 == 0 && cmr_aligned16(dzm));
  CMR_REQUIRE(dz && x && stat && dx && ws && rows > 0 && chan_ok(C));
  CMR_REQUIRE(lddz % 4 == 0 && ldx % 4 == 0 && lddx % 4 == 0 && cmr_aligned16(dz) && cmr_aligned16(x) && cmr_aligned16(dx));
  if (z) CMR_REQUIRE(ldz % 4 == 0 && cmr_aligned16(z));
  else CMR_REQUIRE(slope >= 0.f && slope <= 1.f);      // mask recomputed from x: equals the forward's only for a LeakyReLU slope in [0, 1]
  if (add) CMR_REQUIRE(ldadd % 4 == 0 && cmr_aligned16(add));
  const int nb = red_blocks(rows, C);
  CMR_REQUIRE(ws_bytes >= (int64_t)(nb + 1) * 2 * C * (int64_t)sizeof(float));
  float* part = (float*)ws;
  float* coef = part + (int64_t)nb * 2 * C;
  hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(nb), dim3(RED_THREADS), 0, stream, dz, lddz, z, ldz, slope, x, ldx, stat, rows, C, part);
  hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(C), dim3(64), 0, stream, (const float*)part, nb, rows, C, coef, dgamma, dbeta);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_grid(rows * (C / 4))), dim3(256), 0, stream, dz, lddz, z, ldz, slope, x, ldx, stat,
                     (const float*)coef, add, ldadd, dx, lddx, dzm, lddzm, rows, C);
  return cmr_launch_status();
}

// cmr_bn_bwd_f32 with the reduction's partial sums ALREADY formed by the producer of dz (cmr_conv3x3_wino_bnbwd_nhwc_f32: part [parts][2][C]
// = sums of d and d xhat over disjoint pixel sets): the final reduction + the apply pass; the activation mask from the sign of
// x * stat[2] + stat[3] (no residual in front of the activation).  ws: 2 C floats.
extern "C" int cmr_bn_bwd_from_sums_f32(const float* dz, int64_t lddz, float slope, const float* x, int64_t ldx, const float* stat, const float* part,
                                        int64_t parts, float* dx, int64_t lddx, float* dgamma, float* dbeta, int64_t rows, int C, void* ws,
                                        int64_t ws_bytes, hipStream_t stream) {
  CMR_REQUIRE(dz && x && stat && part && dx && ws && rows > 0 && parts > 0 && parts < 0x7fffffff && chan_ok(C));
  CMR_REQUIRE(lddz % 4 == 0 && ldx % 4 == 0 && lddx % 4 == 0 && cmr_aligned16(dz) && cmr_aligned16(x) && cmr_aligned16(dx));
  CMR_REQUIRE(ws_bytes >= (int64_t)2 * C * (int64_t)sizeof(float));
  float* coef = (float*)ws;
  hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(C), dim3(64), 0, stream, part, (int)parts, rows, C, coef, dgamma, dbeta);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_grid(rows * (C / 4))), dim3(256), 0, stream, dz, lddz, (const float*)nullptr, (int64_t)0, slope, x, ldx,
                     stat, (const float*)coef, (const float*)nullptr, (int64_t)0, dx, lddx, (float*)nullptr, (int64_t)0, rows, C);
  return cmr_launch_status();
}

// The reduction half of cmr_bn_bwd_f32 alone: coef [2][C] = (mean(dy), mean(dy xhat)), dgamma / dbeta; the apply half then rides in the
// prologue of the layer's fused weight / data gradient (cmr_bn_linear_bwd_f32).
extern "C" int cmr_bn_bwd_coef_f32(const float* dz, int64_t lddz, const float* z, int64_t ldz, float slope, const float* x, int64_t ldx,
                                   const float* stat, float* coef, float* dgamma, float* dbeta, int64_t rows, int C, void* ws,
                                   int64_t ws_bytes, hipStream_t stream) {
  CMR_REQUIRE(dz && x && stat && coef && ws && rows > 0 && chan_ok(C));
  CMR_REQUIRE(lddz % 4 == 0 && ldx % 4 == 0 && cmr_aligned16(dz) && cmr_aligned16(x));
  if (z) CMR_REQUIRE(ldz % 4 == 0 && cmr_aligned16(z));
  else CMR_REQUIRE(slope >= 0.f && slope <= 1.f);
  const int nb = red_blocks(rows, C);
  CMR_REQUIRE(ws_bytes >= (int64_t)nb * 2 * C * (int64_t)sizeof(float));
  float* part = (float*)ws;
  hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(nb), dim3(RED_THREADS), 0, stream, dz, lddz, z, ldz, slope, x, ldx, stat, rows, C, part);
  hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(C), dim3(64), 0, stream, (const float*)part, nb, rows, C, coef, dgamma, dbeta);
  return cmr_launch_status();
}

extern "C" int64_t cmr_bn_bwd_workspace_bytes(int64_t rows, int C) {
  return (int64_t)(red_blocks(rows, C) + 1) * 2 * C * sizeof(float);
}

extern "C" int cmr_vecattn_mix_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv, const float* pos,
                                   int64_t ldp, float* a_in, float* vp, int64_t rows, int C, hipStream_t stream) {
  CMR_REQUIRE(q && k && v && pos && a_in && vp && rows >= 0 && C > 0 && C % 4 == 0 && ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && ldp % 4 == 0);
  CMR_REQUIRE(cmr_aligned16(q) && cmr_aligned16(k) && cmr_aligned16(v) && cmr_aligned16(pos) && cmr_aligned16(a_in) && cmr_aligned16(vp));
  if (rows == 0) return CMR_OK;
  hipLaunchKernelGGL(vecattn_mix_kernel, dim3(ew_grid(rows * (C / 4))), dim3(256), 0, stream, q, ldq, k, ldk, v, ldv, pos, ldp, a_in, vp, rows, C);
  return cmr_launch_status();
}

extern "C" int cmr_vecattn_mix_bwd_f32(const float* da, int64_t ldda, const float* dvp, int64_t lddv, float* dk, float* dpos, int64_t rows,
                                       int C, hipStream_t stream) {
  CMR_REQUIRE(da && dvp && dk && dpos && rows >= 0 && C > 0 && C % 4 == 0 && ldda % 4 == 0 && lddv % 4 == 0);
  CMR_REQUIRE(cmr_aligned16(da) && cmr_aligned16(dvp) && cmr_aligned16(dk) && cmr_aligned16(dpos));
  if (rows == 0) return CMR_OK;
  hipLaunchKernelGGL(vecattn_mix_bwd_kernel, dim3(ew_grid(rows * (C / 4))), dim3(256), 0, stream, da, ldda, dvp, lddv, dk, dpos, rows, C);
  return cmr_launch_status();
}

extern "C" int cmr_act_bwd_f32(const float* dz, int64_t lddz, const float* z, int64_t ldz, float slope, const float* add, int64_t ldadd,
                               float* dy, int64_t lddy, int64_t rows, int C, hipStream_t stream) {
  CMR_REQUIRE(dz && z && dy && rows >= 0 && C > 0 && C % 4 == 0 && lddz % 4 == 0 && ldz % 4 == 0 && lddy % 4 == 0);
  CMR_REQUIRE(cmr_aligned16(dz) && cmr_aligned16(z) && cmr_aligned16(dy));
  if (add) CMR_REQUIRE(ldadd % 4 == 0 && cmr_aligned16(add));
  if (rows == 0) return CMR_OK;
  hipLaunchKernelGGL(act_bwd_kernel, dim3(ew_grid(rows * (C / 4))), dim3(256), 0, stream, dz, lddz, z, ldz, slope, add, ldadd, dy, lddy,
                     rows, C);
  return cmr_launch_status();
}

extern "C" int cmr_pool_act_bwd_f32(const float* g, const float* d, float* dc, int B, int H, int W, int C, int ph, int pw, float slope,
                                    hipStream_t stream) {
  CMR_REQUIRE(g && d && dc && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && ph > 0 && pw > 0 && H % ph == 0 && W % pw == 0);
  CMR_REQUIRE(cmr_aligned16(g) && cmr_aligned16(d) && cmr_aligned16(dc));
  hipLaunchKernelGGL(pool_act_bwd_kernel, dim3(ew_grid((int64_t)B * H * W * (C / 4))), dim3(256), 0, stream, g, d, dc, B, H, W, C, ph, pw,
                     slope);
  return cmr_launch_status();
}

static inline int col_blocks(int N, int C) {
  const int rg = RED_THREADS / (C / 4);
  int nb = (N + rg * 16 - 1) / (rg * 16);
  return nb > 128 ? 128 : (nb < 1 ? 1 : nb);
}

extern "C" int64_t cmr_colarg_workspace_bytes(int B, int N, int C) { return (int64_t)B * col_blocks(N, C) * C * 8; }

extern "C" int cmr_colsum_f32(const float* x, int64_t ldx, float* out, void* ws, int64_t ws_bytes, int B, int N, int C, hipStream_t stream) {
  CMR_REQUIRE(x && out && ws && B > 0 && B <= 65535 && N > 0 && chan_ok(C) && ldx % 4 == 0 && cmr_aligned16(x));   // B = gridDim.y
  const int nb = col_blocks(N, C);
  CMR_REQUIRE(ws_bytes >= (int64_t)B * nb * C * 4);
  hipLaunchKernelGGL(col_partial_kernel<false>, dim3(nb, B), dim3(RED_THREADS), 0, stream, x, ldx, N, C, (float*)ws, (int32_t*)nullptr);
  hipLaunchKernelGGL(col_final_kernel<false>, dim3(C, B), dim3(64), 0, stream, (const float*)ws, (const int32_t*)nullptr, nb, C,
                     out, (int32_t*)nullptr);
  return cmr_launch_status();
}

extern "C" int cmr_colmax_arg_f32(const float* x, int64_t ldx, float* out, int32_t* arg, void* ws, int64_t ws_bytes, int B, int N, int C,
                                  hipStream_t stream) {
  CMR_REQUIRE(x && out && arg && ws && B > 0 && B <= 65535 && N > 0 && chan_ok(C) && ldx % 4 == 0 && cmr_aligned16(x));   // B = gridDim.y
  const int nb = col_blocks(N, C);
  CMR_REQUIRE(ws_bytes >= (int64_t)B * nb * C * 8);
  float* pv = (float*)ws;
  int32_t* pi = (int32_t*)(pv + (int64_t)B * nb * C);
  hipLaunchKernelGGL(col_partial_kernel<true>, dim3(nb, B), dim3(RED_THREADS), 0, stream, x, ldx, N, C, pv, pi);
  hipLaunchKernelGGL(col_final_kernel<true>, dim3(C, B), dim3(64), 0, stream, (const float*)pv, (const int32_t*)pi, nb, C, out,
                     arg);
  return cmr_launch_status();
}

extern "C" int cmr_add_at_arg_f32(float* dx, int64_t lddx, const int32_t* arg, const float* g, int64_t ldg, int B, int N, int C,
                                  hipStream_t stream) {
  CMR_REQUIRE(dx && arg && g && B > 0 && N > 0 && C > 0);
  hipLaunchKernelGGL(add_at_arg_kernel, dim3((B * C + 255) / 256), dim3(256), 0, stream, dx, lddx, arg, g, ldg, B, N, C);
  return cmr_launch_status();
}

extern "C" int cmr_linear_bwd_small_f32(const float* x1, int64_t ldx1, int k1, const float* x2, int64_t ldx2, int k2, const float* y,
                                        int64_t ldy, float slope, const float* dy, int64_t lddy, const float* w, int64_t ldw, float* dw,
                                        int64_t lddw, float* db, float* dx1, int64_t lddx1, float* dx2, int64_t lddx2, int acc_dx, int rows,
                                        int n, hipStream_t stream) {
  CMR_REQUIRE(x1 && dy && w && rows > 0 && rows <= 1024 && n > 0 && k1 > 0 && k2 >= 0 && (k2 == 0 || x2));
  SmallBwd a{x1, x2, y, dy, w, dw, db, dx1, dx2, ldx1, ldx2, ldy, lddy, ldw, lddw, lddx1, lddx2, rows, n, k1, k2, acc_dx, slope};
  const int64_t total = (int64_t)n * (k1 + k2) + n + (int64_t)rows * (k1 + k2);
  hipLaunchKernelGGL(linear_bwd_small_kernel, dim3(ew_grid(total)), dim3(256), 0, stream, a);
  return cmr_launch_status();
}

extern "C" int cmr_agent_loss_f32(const float* r_logits, int64_t ldr, const float* t_logits, int64_t ldt, const float* value, int64_t ldv,
                                  const int64_t* expert_r, const int64_t* expert_t, const int64_t* act_r, const int64_t* act_t,
                                  const float* old_logprob, const float* returns, const float* adv, float* d_r, int64_t lddr, float* d_t,
                                  int64_t lddt, float* d_v, int64_t lddv, float* out, int B, int dr, int dt, int S, float alpha,
                                  float clip_eps, float w_value, float w_entropy, float grad_scale, hipStream_t stream) {
  CMR_REQUIRE(r_logits && t_logits && value && expert_r && expert_t && act_r && act_t && d_r && d_t && d_v && out);
  CMR_REQUIRE(B > 0 && dr > 0 && dt > 0 && S > 0 && S <= 64);
  if (alpha > 0.f) CMR_REQUIRE(old_logprob && returns && adv);
  LossArgs a{r_logits, t_logits, value, ldr, ldt, ldv, expert_r, expert_t, act_r, act_t, old_logprob, returns, adv, d_r, d_t, d_v,
             lddr, lddt, lddv, out, B, dr, dt, S, alpha, clip_eps, w_value, w_entropy, grad_scale};
  hipLaunchKernelGGL(agent_loss_kernel, dim3(1), dim3(256), 0, stream, a);
  return cmr_launch_status();
}

extern "C" int cmr_adam_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                            float weight_decay, float bias_correction1, float bias_correction2, float grad_scale, float grad_clip,
                            hipStream_t stream) {
  CMR_REQUIRE(p && g && m && v && n >= 0 && n % 4 == 0 && cmr_aligned16(p) && cmr_aligned16(g) && cmr_aligned16(m) && cmr_aligned16(v));
  CMR_REQUIRE(bias_correction1 > 0.f && bias_correction2 > 0.f);
  if (n == 0) return CMR_OK;
  hipLaunchKernelGGL(adam_kernel, dim3(ew_grid(n / 4)), dim3(256), 0, stream, p, g, m, v, n / 4, lr, beta1, beta2, eps, weight_decay,
                     bias_correction1, sqrtf(bias_correction2), grad_scale, grad_clip);
  return cmr_launch_status();
}

extern "C" int cmr_sgd_f32(float* p, const float* g, float* buf, int64_t n, float lr, float momentum, float weight_decay,
                           float grad_scale, float grad_clip, int first_step, hipStream_t stream) {
  CMR_REQUIRE(p && g && buf && n >= 0 && n % 4 == 0 && cmr_aligned16(p) && cmr_aligned16(g) && cmr_aligned16(buf));
  if (n == 0) return CMR_OK;
  hipLaunchKernelGGL(sgd_kernel, dim3(ew_grid(n / 4)), dim3(256), 0, stream, p, g, buf, n / 4, lr, momentum, weight_decay, grad_scale,
                     grad_clip, first_step);
  return cmr_launch_status();
}

extern "C" int cmr_transpose_slots_f32(const float* src, float* dst, const int64_t* table, int nslots, int64_t total_tiles,
                                       hipStream_t stream) {
  CMR_REQUIRE(src && dst && table && nslots > 0 && total_tiles > 0 && total_tiles < 0x7fffffff);
  hipLaunchKernelGGL(transpose_slots_kernel, dim3((unsigned)total_tiles), dim3(256), 0, stream, src, dst, table, nslots);
  return cmr_launch_status();
}
