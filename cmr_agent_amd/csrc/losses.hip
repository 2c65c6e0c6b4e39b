// Loss / metric values that the reference's heads add to the batch dict in EVERY forward, training or not
// (MultiHeadModel.py:52-109: focal overlap losses + precision / recall / accuracy; :141-178, 218-272: circle loss on
// the sampled (point, pixel) pairs).  Forward values only -- they complete the data-dict contract of geo_model(data);
// the backward pass belongs to the training path.  All reductions are two-stage with a fixed order.
#include "cmr_common.h"

namespace {

// ---- focal loss (models/focal_loss.py:55-110, 153-166: gamma = 2, reduction = mean, eps = 1e-6) + overlap metrics over 2-class logits --------
// per row: soft = softmax(l) + 1e-6 ; focal_c = -alpha (1 - soft_c)^2 log(soft_c) ; loss = sum_c (onehot_c + 1e-6) focal_c
// partial[wg] = {loss sum, #(pred = 1 and label = 1), #(pred = 1), #(label = 1), #(pred = label)}
__global__ __launch_bounds__(256) void focal_partial_kernel(const float* __restrict__ logits, int64_t ld,
                                                            const int64_t* __restrict__ label, float alpha, int64_t rows,
                                                            float* __restrict__ part) {
  __shared__ float red[5][256];
  float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < rows; r += (int64_t)gridDim.x * 256) {
    const float l0 = logits[r * ld], l1 = logits[r * ld + 1];
    const int lab = label[r] != 0 ? 1 : 0;
    const float m = fmaxf(l0, l1);
    const float e0 = expf(l0 - m), e1 = expf(l1 - m);
    const float inv = 1.f / (e0 + e1);
    const float s0 = e0 * inv + 1e-6f, s1 = e1 * inv + 1e-6f;    // FocalLoss.eps (focal_loss.py:160)
    const float f0 = -alpha * ((1.f - s0) * (1.f - s0)) * logf(s0);
    const float f1 = -alpha * ((1.f - s1) * (1.f - s1)) * logf(s1);
    acc[0] += ((lab == 0 ? 1.f : 0.f) + 1e-6f) * f0 + ((lab == 1 ? 1.f : 0.f) + 1e-6f) * f1;
    const int pred = l1 > l0 ? 1 : 0;                 // torch.argmax: first maximum on ties
    acc[1] += (pred == 1 && lab == 1) ? 1.f : 0.f;
    acc[2] += pred == 1 ? 1.f : 0.f;
    acc[3] += lab == 1 ? 1.f : 0.f;
    acc[4] += pred == lab ? 1.f : 0.f;
  }
#pragma unroll
  for (int k = 0; k < 5; ++k) red[k][threadIdx.x] = acc[k];
  __syncthreads();
  for (int st = 128; st >= 1; st >>= 1) {
    if ((int)threadIdx.x < st)
#pragma unroll
      for (int k = 0; k < 5; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x < 5) part[blockIdx.x * 5 + threadIdx.x] = red[threadIdx.x][0];
}

// out = {loss, precision, recall, accuracy}.  One wave: lane l sums partials l, l + 64, ... in ascending order, then a fixed xor-shuffle
// tree over the 64 lanes (deterministic; a single thread walking all partials took 50 us at 512 workgroups).
__global__ __launch_bounds__(64) void focal_final_kernel(const float* __restrict__ part, int nwg, int64_t rows, int B, float* __restrict__ out) {
  float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  for (int i = threadIdx.x; i < nwg; i += 64)
#pragma unroll
    for (int k = 0; k < 5; ++k) s[k] += part[i * 5 + k];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1)
#pragma unroll
    for (int k = 0; k < 5; ++k) s[k] += __shfl_xor(s[k], m);
  if (threadIdx.x != 0) return;
  out[0] = s[0] / (float)rows;
  out[1] = s[1] / s[2];                                // (label[pred == 1]).sum() / pred.sum()   (NaN when nothing is predicted)
  out[2] = s[1] / s[3];                                // (pred[label == 1]).sum() / label.sum()
  out[3] = (s[4] / (float)B) / (float)(rows / B);      // (pred == label).sum() / b / n
}

// ---- circle loss on n x n (point, pixel) pairs per sample (MultiHeadModel.py:141-178) ------------------------------
// point features pts[b][i] = pc_feat[b*N + pc_idx[b][i]], pixel features pix[b][j] = img_feat[b][y_j][x_j] (NHWC rows).
// EP[b][i][j] = s (pos - pm) max(pos - pm, 0), pos = d - 1e5 (1 - mask) ;  EN = s (nm - neg) max(nm - neg, 0), neg = d + 1e5 mask
// with d = |pts_i - pix_j|, mask = |xy_float_i - xy_int_j| <= dist_thres.
__global__ __launch_bounds__(256) void circle_pairs_kernel(const float* __restrict__ pc_feat, const float* __restrict__ img_feat,
                                                           const int64_t* __restrict__ pc_idx, const int64_t* __restrict__ xy_int,
                                                           const float* __restrict__ xy_float, int N, int h, int w, int n,
                                                           float dist_thres, float pos_margin, float neg_margin, float log_scale,
                                                           float* __restrict__ EP, float* __restrict__ EN) {
  __shared__ float pts[16][65], pix[16][65];
  const int b = blockIdx.z, i0 = blockIdx.y * 16, j0 = blockIdx.x * 16;
  const int tid = threadIdx.x;
  for (int e = tid; e < 16 * 64; e += 256) {
    const int r = e >> 6, c = e & 63;
    const int i = i0 + r < n ? i0 + r : n - 1, j = j0 + r < n ? j0 + r : n - 1;
    pts[r][c] = pc_feat[((int64_t)b * N + pc_idx[(int64_t)b * n + i]) * 64 + c];
    const int64_t x = xy_int[((int64_t)b * 2 + 0) * n + j], y = xy_int[((int64_t)b * 2 + 1) * n + j];
    pix[r][c] = img_feat[(((int64_t)b * h + y) * w + x) * 64 + c];
  }
  __syncthreads();
  const int li = tid >> 4, lj = tid & 15;
  const int i = i0 + li, j = j0 + lj;
  if (i >= n || j >= n) return;
  float s = 0.f;
#pragma unroll 8
  for (int c = 0; c < 64; ++c) {
    const float df = pts[li][c] - pix[lj][c];
    s += df * df;
  }
  const float d = sqrtf(s);
  const float dx = xy_float[((int64_t)b * 2 + 0) * n + i] - (float)xy_int[((int64_t)b * 2 + 0) * n + j];
  const float dy = xy_float[((int64_t)b * 2 + 1) * n + i] - (float)xy_int[((int64_t)b * 2 + 1) * n + j];
  const float mask = sqrtf(dx * dx + dy * dy) <= dist_thres ? 1.f : 0.f;
  const float pos = d - 1e5f * (1.f - mask), neg = d + 1e5f * mask;
  const int64_t o = ((int64_t)b * n + i) * n + j;
  EP[o] = log_scale * (pos - pos_margin) * fmaxf(pos - pos_margin, 0.f);
  EN[o] = log_scale * (neg_margin - neg) * fmaxf(neg_margin - neg, 0.f);
}

__device__ __forceinline__ float softplus20(float x) { return x > 20.f ? x : log1pf(expf(x)); }   // F.softplus defaults

// Row and column log-sum-exps of EP / EN and their softplus terms: terms[b][0 .. n) = rows, [n .. 2 n) = columns.
// Blocks [0, ceil(n / 4)): four waves = four rows, lanes along the row (coalesced), online (max, sum) per lane then a wave reduction.
// Blocks beyond: 256 threads = 256 columns, each walks the rows of its column (consecutive threads read consecutive addresses).
// (The first version gave one workgroup a whole sample and one THREAD a row and a column: 2 x 512 strided serial steps, 8 workgroups on
// the chip -- 630 us at n = 512 against ~10 us of memory time.)
__device__ __forceinline__ void lse_push(float& m, float& s, float v) {
  const float mn = fmaxf(m, v);
  s = s * expf(m - mn) + expf(v - mn);
  m = mn;
}
__device__ __forceinline__ void lse_merge(float& m, float& s, float mo, float so) {
  const float mn = fmaxf(m, mo);
  s = (m == -INFINITY ? 0.f : s * expf(m - mn)) + (mo == -INFINITY ? 0.f : so * expf(mo - mn));
  m = mn;
}

__global__ __launch_bounds__(256) void circle_terms_kernel(const float* __restrict__ EP, const float* __restrict__ EN, int n, float log_scale,
                                                           float* __restrict__ terms) {
  const int b = blockIdx.y;
  const float* ep = EP + (int64_t)b * n * n;
  const float* en = EN + (int64_t)b * n * n;
  const int row_blocks = (n + 3) / 4;
  if ((int)blockIdx.x < row_blocks) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= n) return;
    float mp = -INFINITY, sp = 0.f, mn = -INFINITY, sn = 0.f;
    for (int k = lane; k < n; k += 64) {
      lse_push(mp, sp, ep[(int64_t)r * n + k]);
      lse_push(mn, sn, en[(int64_t)r * n + k]);
    }
#pragma unroll
    for (int x = 32; x >= 1; x >>= 1) {
      lse_merge(mp, sp, __shfl_xor(mp, x), __shfl_xor(sp, x));
      lse_merge(mn, sn, __shfl_xor(mn, x), __shfl_xor(sn, x));
    }
    if (lane == 0) terms[(int64_t)b * 2 * n + r] = softplus20((mp + logf(sp)) + (mn + logf(sn))) / log_scale;
  } else {
    const int c = (blockIdx.x - row_blocks) * 256 + threadIdx.x;
    if (c >= n) return;
    float mp = -INFINITY, sp = 0.f, mn = -INFINITY, sn = 0.f;
    int k = 0;
    for (; k + 16 <= n; k += 16) {                       // 32 loads in flight per thread (a walk of n dependent round trips otherwise); same order
      float vp[16], vn[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        vp[u] = ep[(int64_t)(k + u) * n + c];
        vn[u] = en[(int64_t)(k + u) * n + c];
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        lse_push(mp, sp, vp[u]);
        lse_push(mn, sn, vn[u]);
      }
    }
    for (; k < n; ++k) {
      lse_push(mp, sp, ep[(int64_t)k * n + c]);
      lse_push(mn, sn, en[(int64_t)k * n + c]);
    }
    terms[(int64_t)b * 2 * n + n + c] = softplus20((mp + logf(sp)) + (mn + logf(sn))) / log_scale;
  }
}

// partial[b] = sum of the 2 n terms of sample b: thread t adds terms t, t + 512, ... in order, then a fixed tree
__global__ __launch_bounds__(512) void circle_reduce_kernel(const float* __restrict__ terms, int n, float* __restrict__ part) {
  __shared__ float red[512];
  const int b = blockIdx.x, t = threadIdx.x;
  float term = 0.f;
  for (int r = t; r < 2 * n; r += 512) term += terms[(int64_t)b * 2 * n + r];
  red[t] = term;
  __syncthreads();
  for (int st = 256; st >= 1; st >>= 1) {
    if (t < st) red[t] += red[t + st];
    __syncthreads();
  }
  if (t == 0) part[b] = red[0];
}

__global__ void circle_final_kernel(const float* __restrict__ part, int B, int n, float lambda, float* __restrict__ out) {
  if (threadIdx.x != 0) return;
  float s = 0.f;
  for (int b = 0; b < B; ++b) s += part[b];
  out[0] = lambda * (s / (float)((int64_t)B * n));
}

}  // namespace

extern "C" int64_t cmr_focal_metrics_workspace_bytes(int64_t rows) {
  const int64_t nwg = rows <= 0 ? 0 : ((rows + 255) / 256 < 512 ? (rows + 255) / 256 : 512);
  return nwg * 5 * (int64_t)sizeof(float);
}

extern "C" int cmr_focal_metrics_f32(const float* logits, int64_t ld, const int64_t* label, float alpha, int64_t rows, int B,
                                     float* out4, void* workspace, int64_t workspace_bytes, hipStream_t stream) {
  CMR_REQUIRE(logits && label && out4 && workspace && rows > 0 && B > 0 && rows % B == 0 && ld >= 2);
  CMR_REQUIRE(workspace_bytes >= cmr_focal_metrics_workspace_bytes(rows));
  const int nwg = (int)(cmr_focal_metrics_workspace_bytes(rows) / (5 * sizeof(float)));
  hipLaunchKernelGGL(focal_partial_kernel, dim3(nwg), dim3(256), 0, stream, logits, ld, label, alpha, rows, (float*)workspace);
  hipLaunchKernelGGL(focal_final_kernel, dim3(1), dim3(64), 0, stream, (const float*)workspace, nwg, rows, B, out4);
  return cmr_launch_status();
}

extern "C" int64_t cmr_circle_loss_workspace_bytes(int B, int n) {
  return B <= 0 || n <= 0 ? 0 : ((int64_t)2 * B * n * n + B + (int64_t)2 * B * n) * (int64_t)sizeof(float);
}

extern "C" int cmr_circle_loss_f32(const float* pc_feat, const float* img_feat, const int64_t* pc_idx, const int64_t* xy_int,
                                   const float* xy_float, int B, int N, int h, int w, int n, float dist_thres, float pos_margin,
                                   float neg_margin, float log_scale, float lambda, float* out, void* workspace,
                                   int64_t workspace_bytes, hipStream_t stream) {
  CMR_REQUIRE(pc_feat && img_feat && pc_idx && xy_int && xy_float && out && workspace && B > 0 && B <= 65535 && N > 0 && h > 0 &&
              w > 0 && n > 0);
  CMR_REQUIRE(workspace_bytes >= cmr_circle_loss_workspace_bytes(B, n));
  float* EP = (float*)workspace;
  float* EN = EP + (int64_t)B * n * n;
  float* part = EN + (int64_t)B * n * n;
  float* terms = part + B;
  hipLaunchKernelGGL(circle_pairs_kernel, dim3((n + 15) / 16, (n + 15) / 16, B), dim3(256), 0, stream, pc_feat, img_feat, pc_idx,
                     xy_int, xy_float, N, h, w, n, dist_thres, pos_margin, neg_margin, log_scale, EP, EN);
  hipLaunchKernelGGL(circle_terms_kernel, dim3((n + 3) / 4 + (n + 255) / 256, B), dim3(256), 0, stream, (const float*)EP, (const float*)EN, n,
                     log_scale, terms);
  hipLaunchKernelGGL(circle_reduce_kernel, dim3(B), dim3(512), 0, stream, (const float*)terms, n, part);
  hipLaunchKernelGGL(circle_final_kernel, dim3(1), dim3(64), 0, stream, (const float*)part, B, n, lambda, out);
  return cmr_launch_status();
}
