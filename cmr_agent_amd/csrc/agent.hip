// Agent-iteration kernels: the pose-conditioned observation (environment/environment.py:24-126),
// the pose update (environment.py:179-260), action selection (CMRAgent.py:118-127) and the small
// head post-processing ops of MultiHeadModel.forward (MultiHeadModel.py:327-348, F.normalize :233,:241).
#include "cmr_common.h"

namespace {

// state_3d[b*N+n, 0:8] = (x, y, z, overlap_pred, in_cam, 0, 0, 0)        (environment.py:88-124)
// and, for predicted-overlap points that land in view, acc[b, y*w+x, :] += feat, cnt[b, y*w+x] += 1
// (environment.py:39-80: scatter_mean numerator / denominator).  One wave per 4 points, lane = channel.
__global__ __launch_bounds__(256) void project_scatter_kernel(const float* __restrict__ pc4, const float* __restrict__ feat,
                                                              const uint8_t* __restrict__ overlap,
                                                              const float* __restrict__ pose /*[B,4,4]*/,
                                                              const float* __restrict__ Kmat /*[B,3,3]*/,
                                                              const float* __restrict__ mean4 /*[B,4]*/,
                                                              float* __restrict__ acc, float* __restrict__ cnt,
                                                              float* __restrict__ state3d, int B, int N, int h, int w) {
  const int lane = threadIdx.x & 63;
  const int64_t p0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
  const int64_t total = (int64_t)B * N;
  for (int q = 0; q < 4; ++q) {
    const int64_t p = p0 + q;
    if (p >= total) return;
    const int b = (int)(p / N);
    const f32x4 pt = *reinterpret_cast<const f32x4*>(pc4 + p * 4);
    const float* R = pose + b * 16;
    const float* Kb = Kmat + b * 9;
    const float mx = mean4[b * 4 + 0], my = mean4[b * 4 + 1], mz = mean4[b * 4 + 2];
    const float cx = pt[0] - mx, cy = pt[1] - my, cz = pt[2] - mz;
    // R @ (p - mu) + mu + t
    const float tx = (R[0] * cx + R[1] * cy + R[2] * cz) + mx + R[3];
    const float ty = (R[4] * cx + R[5] * cy + R[6] * cz) + my + R[7];
    const float tz = (R[8] * cx + R[9] * cy + R[10] * cz) + mz + R[11];
    float u = Kb[0] * tx + Kb[1] * ty + Kb[2] * tz;
    float v = Kb[3] * tx + Kb[4] * ty + Kb[5] * tz;
    const float zc = Kb[6] * tx + Kb[7] * ty + Kb[8] * tz;
    u = u / zc;
    v = v / zc;
    const bool inside = (u >= 0.f) && (u <= (float)(w - 1)) && (v >= 0.f) && (v <= (float)(h - 1)) && (zc > 0.f);
    const bool ov = overlap[p] != 0;
    if (lane < 8) {
      float s = 0.f;
      if (lane < 3) s = pt[lane];
      else if (lane == 3) s = ov ? 1.f : 0.f;
      else if (lane == 4) s = inside ? 1.f : 0.f;
      state3d[p * 8 + lane] = s;
    }
    if (ov && inside) {
      const int xi = (int)rintf(u), yi = (int)rintf(v);        // torch.round = half to even
      const int64_t cell = (int64_t)b * h * w + (int64_t)yi * w + xi;
      atomicAdd(acc + cell * 64 + lane, feat[p * 64 + lane]);
      if (lane == 0) atomicAdd(cnt + cell, 1.f);
    }
  }
}

// state2d[b,y,x,0:64] = img_feat ; state2d[b,y,x,64:128] = acc / max(cnt,1)
// clear: the cells that received points are zeroed again after they have been read (16 lanes share a cell; lane 0 of the
// 16 resets the counter after all of them have read it), so the next scatter needs no 55 MB memset.
__global__ __launch_bounds__(256) void observation_finalize_kernel(const float* __restrict__ img_feat,
                                                                   float* __restrict__ acc,
                                                                   float* __restrict__ cnt, float* __restrict__ state2d,
                                                                   float* __restrict__ proj, int64_t cells, int write_img, int clear) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t cell = e >> 4;
  if (cell >= cells) return;
  const int c = (int)(e & 15) * 4;
  const float nraw = cnt[cell];
  const float n = fmaxf(nraw, 1.f);
  f32x4 a = *reinterpret_cast<const f32x4*>(acc + cell * 64 + c);
  if (clear && nraw > 0.f) {
    *reinterpret_cast<f32x4*>(acc + cell * 64 + c) = f32x4{0.f, 0.f, 0.f, 0.f};
    if ((threadIdx.x & 15) == 0) cnt[cell] = 0.f;      // the cell's 16 lanes sit in one wave and have all loaded the count above
  }
  a[0] /= n; a[1] /= n; a[2] /= n; a[3] /= n;
  if (proj) *reinterpret_cast<f32x4*>(proj + cell * 64 + c) = a;
  if (state2d == nullptr) return;                       // (uniform) only the projected half is wanted: 110 of 275 MB per step at 8 x 88 x 304
  *reinterpret_cast<f32x4*>(state2d + cell * 128 + 64 + c) = a;
  if (write_img) *reinterpret_cast<f32x4*>(state2d + cell * 128 + c) = *reinterpret_cast<const f32x4*>(img_feat + cell * 64 + c);
}

// ---- the projected half of the observation written directly (inference loops that consume the two halves separately) -----------
// project_scatter + observation_finalize make two passes over the whole [B,h,w,64] map per agent step (accumulate, then normalise /
// copy / re-zero: 165 MB at 8 x 88 x 304) although only the cells that received points change.  Here the map `proj` is kept across
// steps and touched per POINT only:
//   obs_project_kernel  (thread per point)  zeroes the cell its point occupied in the previous step (proj row + count), projects the
//                                           point with the new pose, writes the state_3d row and the new cell index (-1: not in view)
//   obs_count_kernel    (thread per point)  cnt[cell] += 1
//   obs_scatter_kernel  (wave per 4 points) proj[cell][c] += feat[p][c] / cnt[cell]      (scatter_mean as a sum of pre-divided terms)
// The clears of a step never meet the increments of the same step (different launches), so no buffer parity has to survive between
// hipGraph replays; cells hit by nobody stay zero from the initial fill.
__global__ __launch_bounds__(256) void obs_project_kernel(const float* __restrict__ pc4, const uint8_t* __restrict__ overlap,
                                                          const float* __restrict__ pose, const float* __restrict__ Kmat,
                                                          const float* __restrict__ mean4, float* __restrict__ proj, float* __restrict__ cnt,
                                                          int32_t* __restrict__ cell, float* __restrict__ state3d, int B, int N, int h, int w) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= (int64_t)B * N) return;
  const int32_t prev = cell[p];
  if (prev >= 0) {
    f32x4* row = reinterpret_cast<f32x4*>(proj + (int64_t)prev * 64);
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 16; ++i) row[i] = z;
    cnt[prev] = 0.f;
  }
  const int b = (int)(p / N);
  const f32x4 pt = *reinterpret_cast<const f32x4*>(pc4 + p * 4);
  const float* R = pose + b * 16;
  const float* Kb = Kmat + b * 9;
  const float mx = mean4[b * 4 + 0], my = mean4[b * 4 + 1], mz = mean4[b * 4 + 2];
  const float cx = pt[0] - mx, cy = pt[1] - my, cz = pt[2] - mz;
  const float tx = (R[0] * cx + R[1] * cy + R[2] * cz) + mx + R[3];
  const float ty = (R[4] * cx + R[5] * cy + R[6] * cz) + my + R[7];
  const float tz = (R[8] * cx + R[9] * cy + R[10] * cz) + mz + R[11];
  float u = Kb[0] * tx + Kb[1] * ty + Kb[2] * tz;
  float v = Kb[3] * tx + Kb[4] * ty + Kb[5] * tz;
  const float zc = Kb[6] * tx + Kb[7] * ty + Kb[8] * tz;
  u = u / zc;
  v = v / zc;
  const bool inside = (u >= 0.f) && (u <= (float)(w - 1)) && (v >= 0.f) && (v <= (float)(h - 1)) && (zc > 0.f);
  const bool ov = overlap[p] != 0;
  *reinterpret_cast<f32x4*>(state3d + p * 8) = f32x4{pt[0], pt[1], pt[2], ov ? 1.f : 0.f};
  *reinterpret_cast<f32x4*>(state3d + p * 8 + 4) = f32x4{inside ? 1.f : 0.f, 0.f, 0.f, 0.f};
  int32_t c = -1;
  if (ov && inside) {
    const int xi = (int)rintf(u), yi = (int)rintf(v);            // torch.round = half to even
    c = (int32_t)((int64_t)b * h * w + (int64_t)yi * w + xi);
  }
  cell[p] = c;
}

__global__ __launch_bounds__(256) void obs_count_kernel(const int32_t* __restrict__ cell, float* __restrict__ cnt, int64_t total) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= total) return;
  const int32_t c = cell[p];
  if (c >= 0) atomicAdd(cnt + c, 1.f);
}

__global__ __launch_bounds__(256) void obs_scatter_kernel(const float* __restrict__ feat, const int32_t* __restrict__ cell,
                                                          const float* __restrict__ cnt, float* __restrict__ proj, int64_t total) {
  const int lane = threadIdx.x & 63;
  const int64_t p0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int64_t p = p0 + q;
    if (p >= total) return;
    const int32_t c = cell[p];
    if (c >= 0) atomicAdd(proj + (int64_t)c * 64 + lane, feat[p * 64 + lane] / cnt[c]);
  }
}

// pose[b] <- [E_xyz(move_r) @ R | t + move_t]   with move_* looked up in the float64 step tables and
// rounded to float32 on assignment (environment.py:186-205).  One thread per sample.
__global__ void pose_step_kernel(float* __restrict__ pose, const int64_t* __restrict__ act_r, const int64_t* __restrict__ act_t,
                                 const double* __restrict__ r_steps, const double* __restrict__ t_steps, int B, int six_dof) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float mr[3] = {0.f, 0.f, 0.f}, mt[3] = {0.f, 0.f, 0.f};
  if (six_dof) {
    for (int i = 0; i < 3; ++i) { mr[i] = (float)r_steps[act_r[b * 3 + i]]; mt[i] = (float)t_steps[act_t[b * 3 + i]]; }
  } else {
    mr[1] = (float)r_steps[act_r[b]];
    mt[0] = (float)t_steps[act_t[b * 2 + 0]];
    mt[2] = (float)t_steps[act_t[b * 2 + 1]];
  }
  const float cxv = cosf(mr[0]), sxv = sinf(mr[0]), cyv = cosf(mr[1]), syv = sinf(mr[1]), czv = cosf(mr[2]),
              szv = sinf(mr[2]);
  const float Rx[9] = {1, 0, 0, 0, cxv, -sxv, 0, sxv, cxv};
  const float Ry[9] = {cyv, 0, syv, 0, 1, 0, -syv, 0, cyv};
  const float Rz[9] = {czv, -szv, 0, szv, czv, 0, 0, 0, 1};
  float T[9], E[9], Rn[9];
  auto mm = [](const float* A, const float* Bm, float* C) {
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        float s = 0.f;
        for (int k = 0; k < 3; ++k) s += A[i * 3 + k] * Bm[k * 3 + j];
        C[i * 3 + j] = s;
      }
  };
  mm(Rx, Ry, T);
  mm(T, Rz, E);
  float* P = pose + b * 16;
  float Rold[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Rold[i * 3 + j] = P[i * 4 + j];
  mm(E, Rold, Rn);
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) P[i * 4 + j] = Rn[i * 3 + j];
    P[i * 4 + 3] += mt[i];
  }
}

// to_disentangled (environment.py:14-21): t <- t - mu + R mu
__global__ void to_disentangled_kernel(float* __restrict__ pose, const float* __restrict__ mean4, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float* P = pose + b * 16;
  const float mx = mean4[b * 4], my = mean4[b * 4 + 1], mz = mean4[b * 4 + 2];
  const float mu[3] = {mx, my, mz};
  for (int i = 0; i < 3; ++i) {
    const float rm = P[i * 4 + 0] * mx + P[i * 4 + 1] * my + P[i * 4 + 2] * mz;
    P[i * 4 + 3] = P[i * 4 + 3] - mu[i] + rm;
  }
}

// argmax over the last dim (first index on ties): logits x[o*so + i*si + 0..n) -> int64 out[o*inner + i]
__global__ void argmax_rows_kernel(const float* __restrict__ x, int64_t* __restrict__ out, int outer, int inner, int n,
                                   int64_t so, int64_t si) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= outer * inner) return;
  const float* p = x + (r / inner) * so + (r % inner) * si;
  float best = p[0];
  int bi = 0;
  for (int i = 1; i < n; ++i) {
    const float v = p[i];
    if (v > best) { best = v; bi = i; }
  }
  out[r] = bi;
}

// two-class softmax of row-major logits [R,2]: prob of class 1, and the two thresholds
__global__ __launch_bounds__(256) void softmax2_kernel(const float* __restrict__ logits, int64_t ld, float* __restrict__ prob,
                                                       uint8_t* __restrict__ pred_lo, uint8_t* __restrict__ pred_hi,
                                                       float thr_lo, float thr_hi, int64_t rows) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  const float a = logits[r * ld], b = logits[r * ld + 1];
  const float m = fmaxf(a, b);
  const float ea = expf(a - m), eb = expf(b - m);
  const float p = eb / (ea + eb);
  prob[r] = p;
  if (pred_lo) pred_lo[r] = p > thr_lo;
  if (pred_hi) pred_hi[r] = p > thr_hi;
}

// y = x / max(||x||_2, 1e-12) over 64 channels (F.normalize(dim=1)); 16 lanes per row
__global__ __launch_bounds__(256) void l2norm64_kernel(const float* __restrict__ x, int64_t ldx, float* __restrict__ y,
                                                       int64_t ldy, int64_t rows) {
  const int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int c = (threadIdx.x & 15) * 4;
  const bool ok = row < rows;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (ok) v = *reinterpret_cast<const f32x4*>(x + row * ldx + c);
  float q = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
#pragma unroll
  for (int m = 1; m < 16; m <<= 1) q += __shfl_xor(q, m);
  const float d = fmaxf(sqrtf(q), 1e-12f);
  if (ok) {
    f32x4 o = {v[0] / d, v[1] / d, v[2] / d, v[3] / d};
    *reinterpret_cast<f32x4*>(y + row * ldy + c) = o;
  }
}

}  // namespace

extern "C" int cmr_project_scatter_f32(const float* pc4, const float* feat, const uint8_t* overlap, const float* pose,
                                       const float* Kmat, const float* mean4, float* acc, float* cnt, float* state3d,
                                       int B, int N, int h, int w, int zero_first, hipStream_t stream) {
  CMR_REQUIRE(pc4 && feat && overlap && pose && Kmat && mean4 && acc && cnt && state3d && B > 0 && N > 0 && h > 0 && w > 0);
  const int64_t cells = (int64_t)B * h * w;
  if (zero_first) {      // otherwise the caller guarantees zeroed accumulators (cmr_observation_finalize_f32 with clear = 1 leaves them so)
    if (hipMemsetAsync(acc, 0, cells * 64 * sizeof(float), stream) != hipSuccess) return CMR_ELAUNCH;
    if (hipMemsetAsync(cnt, 0, cells * sizeof(float), stream) != hipSuccess) return CMR_ELAUNCH;
  }
  const int64_t total = (int64_t)B * N;
  hipLaunchKernelGGL(project_scatter_kernel, dim3((unsigned)((total + 15) / 16)), dim3(256), 0, stream, pc4, feat, overlap,
                     pose, Kmat, mean4, acc, cnt, state3d, B, N, h, w);
  return cmr_launch_status();
}

extern "C" int cmr_observation_proj_f32(const float* pc4, const float* feat, const uint8_t* overlap, const float* pose, const float* Kmat,
                                        const float* mean4, float* proj, float* cnt, int32_t* cell, float* state3d, int B, int N, int h,
                                        int w, hipStream_t stream) {
  CMR_REQUIRE(pc4 && feat && overlap && pose && Kmat && mean4 && proj && cnt && cell && state3d && B > 0 && N > 0 && h > 0 && w > 0);
  CMR_REQUIRE((int64_t)B * h * w < 0x7fffffff && cmr_aligned16(proj) && cmr_aligned16(state3d) && cmr_aligned16(pc4));
  const int64_t total = (int64_t)B * N;
  const unsigned per_point = (unsigned)((total + 255) / 256);
  hipLaunchKernelGGL(obs_project_kernel, dim3(per_point), dim3(256), 0, stream, pc4, overlap, pose, Kmat, mean4, proj, cnt, cell, state3d, B, N,
                     h, w);
  hipLaunchKernelGGL(obs_count_kernel, dim3(per_point), dim3(256), 0, stream, (const int32_t*)cell, cnt, total);
  hipLaunchKernelGGL(obs_scatter_kernel, dim3((unsigned)((total + 15) / 16)), dim3(256), 0, stream, feat, (const int32_t*)cell,
                     (const float*)cnt, proj, total);
  return cmr_launch_status();
}

extern "C" int cmr_observation_finalize_f32(const float* img_feat, float* acc, float* cnt, float* state2d,
                                            float* proj, int B, int h, int w, int write_img, int clear, hipStream_t stream) {
  CMR_REQUIRE(img_feat && acc && cnt && (state2d || proj) && B > 0 && h > 0 && w > 0);
  const int64_t cells = (int64_t)B * h * w;
  hipLaunchKernelGGL(observation_finalize_kernel, dim3((unsigned)((cells * 16 + 255) / 256)), dim3(256), 0, stream,
                     img_feat, acc, cnt, state2d, proj, cells, write_img, clear);
  return cmr_launch_status();
}

extern "C" int cmr_pose_step_f32(float* pose, const int64_t* act_r, const int64_t* act_t, const double* r_steps,
                                 const double* t_steps, int B, int six_dof, hipStream_t stream) {
  CMR_REQUIRE(pose && act_r && act_t && r_steps && t_steps && B > 0);
  hipLaunchKernelGGL(pose_step_kernel, dim3((B + 63) / 64), dim3(64), 0, stream, pose, act_r, act_t, r_steps, t_steps, B,
                     six_dof);
  return cmr_launch_status();
}

extern "C" int cmr_to_disentangled_f32(float* pose, const float* mean4, int B, hipStream_t stream) {
  CMR_REQUIRE(pose && mean4 && B > 0);
  hipLaunchKernelGGL(to_disentangled_kernel, dim3((B + 63) / 64), dim3(64), 0, stream, pose, mean4, B);
  return cmr_launch_status();
}

extern "C" int cmr_argmax_rows_f32(const float* x, int64_t* out, int outer, int inner, int n, int64_t stride_outer,
                                   int64_t stride_inner, hipStream_t stream) {
  CMR_REQUIRE(x && out && outer > 0 && inner > 0 && n > 0);
  hipLaunchKernelGGL(argmax_rows_kernel, dim3((outer * inner + 63) / 64), dim3(64), 0, stream, x, out, outer, inner, n,
                     stride_outer, stride_inner);
  return cmr_launch_status();
}

extern "C" int cmr_softmax2_f32(const float* logits, int64_t ld, float* prob, uint8_t* pred_lo, uint8_t* pred_hi,
                                float thr_lo, float thr_hi, int64_t rows, hipStream_t stream) {
  CMR_REQUIRE(logits && prob && rows > 0 && ld >= 2);
  hipLaunchKernelGGL(softmax2_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, stream, logits, ld, prob,
                     pred_lo, pred_hi, thr_lo, thr_hi, rows);
  return cmr_launch_status();
}

extern "C" int cmr_l2norm64_f32(const float* x, int64_t ldx, float* y, int64_t ldy, int64_t rows, hipStream_t stream) {
  CMR_REQUIRE(x && y && rows > 0 && ldx % 4 == 0 && ldy % 4 == 0 && cmr_aligned16(x) && cmr_aligned16(y));
  hipLaunchKernelGGL(l2norm64_kernel, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, stream, x, ldx, y, ldy, rows);
  return cmr_launch_status();
}
