// Rollout ops of the training loop, so that the per-step bookkeeping of Train_Agent.py:218-250 stays on
// the device: the expert action (reference environment/environment.py:143-176 -- there a host round trip through
// scipy.spatial.transform.Rotation for every step), the dense step reward (:263-302) and the discounted return scan of
// the replay buffer (environment/buffer.py:24-33).  Tiny, latency-bound kernels.
#include "cmr_common.h"

namespace {

// nearest entry of a float64 step table (first index on ties, torch.argmin)
__device__ __forceinline__ int64_t nearest_step(double v, const double* __restrict__ steps, int S) {
  double best = fabs(v - steps[0]);
  int64_t bi = 0;
  for (int i = 1; i < S; ++i) {
    const double d = fabs(v - steps[i]);
    if (d < best) { best = d; bi = i; }
  }
  return bi;
}

// one thread per sample.  delta_R = R_t R_s^T and delta_t in float32 (the reference's tensors); Euler angles in float64
// (scipy): extrinsic xyz, R = Rz(c) Ry(b) Rx(a):  a = atan2(R21, R22), b = atan2(-R20, |(R00, R10)|), c = atan2(R10, R00).
__global__ void expert_action_kernel(const float* __restrict__ src, const float* __restrict__ tgt, const double* __restrict__ r_steps,
                                     const double* __restrict__ t_steps, int S, int six_dof, int64_t* __restrict__ act_r,
                                     int64_t* __restrict__ act_t, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float* ps = src + b * 16;
  const float* pt = tgt + b * 16;
  float dR[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) dR[i][j] = (pt[i * 4 + 0] * ps[j * 4 + 0] + pt[i * 4 + 1] * ps[j * 4 + 1]) + pt[i * 4 + 2] * ps[j * 4 + 2];
  double ang[3];
  ang[0] = atan2((double)dR[2][1], (double)dR[2][2]);
  ang[1] = atan2(-(double)dR[2][0], sqrt((double)dR[0][0] * dR[0][0] + (double)dR[1][0] * dR[1][0]));
  ang[2] = atan2((double)dR[1][0], (double)dR[0][0]);
  const double pi = 3.14159265358979323846;
  if (ang[0] > 3.0) {                 // environment.py:154-161: the decomposition flipped over (|yaw| > 90 deg)
    ang[0] = 0.0;
    ang[2] = 0.0;
    if (ang[1] > 0.0) ang[1] = pi - ang[1];
    else if (ang[1] < 0.0) ang[1] = -pi - ang[1];
  }
  int64_t ar[3], at[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    ar[i] = nearest_step(ang[i], r_steps, S);
    at[i] = nearest_step((double)(pt[i * 4 + 3] - ps[i * 4 + 3]), t_steps, S);
  }
  if (six_dof) {
#pragma unroll
    for (int i = 0; i < 3; ++i) { act_r[b * 3 + i] = ar[i]; act_t[b * 3 + i] = at[i]; }
  } else {
    act_r[b] = ar[1];
    act_t[b * 2 + 0] = at[0];
    act_t[b * 2 + 1] = at[2];
  }
}

// one workgroup per sample: centroid of the cloud, then the mean over the masked points of |cam - (p - centroid)|^2
__global__ __launch_bounds__(256) void reward_kernel(const float* __restrict__ pc /*[B,3,N]*/, const float* __restrict__ cam,
                                                     const int64_t* __restrict__ mask /*[B,N]*/, const float* __restrict__ prev,
                                                     float* __restrict__ dist, float* __restrict__ reward, int N) {
  __shared__ float red[4][256];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* p = pc + (int64_t)b * 3 * N;
  const float* c = cam + (int64_t)b * 3 * N;
  const int64_t* m = mask + (int64_t)b * N;
  float s[3] = {0.f, 0.f, 0.f};
  for (int i = tid; i < N; i += 256) { s[0] += p[i]; s[1] += p[N + i]; s[2] += p[2 * N + i]; }
#pragma unroll
  for (int k = 0; k < 3; ++k) red[k][tid] = s[k];
  __syncthreads();
  for (int st = 128; st >= 1; st >>= 1) {
    if (tid < st)
#pragma unroll
      for (int k = 0; k < 3; ++k) red[k][tid] += red[k][tid + st];
    __syncthreads();
  }
  const float mx = red[0][0] / (float)N, my = red[1][0] / (float)N, mz = red[2][0] / (float)N;
  __syncthreads();
  float acc = 0.f, cnt = 0.f;
  for (int i = tid; i < N; i += 256) {
    if (m[i] != 0) {
      const float dx = c[i] - (p[i] - mx), dy = c[N + i] - (p[N + i] - my), dz = c[2 * N + i] - (p[2 * N + i] - mz);
      acc += (dx * dx + dy * dy) + dz * dz;
      cnt += 1.f;
    }
  }
  red[0][tid] = acc; red[1][tid] = cnt;
  __syncthreads();
  for (int st = 128; st >= 1; st >>= 1) {
    if (tid < st) { red[0][tid] += red[0][tid + st]; red[1][tid] += red[1][tid + st]; }
    __syncthreads();
  }
  if (tid == 0) {
    const float d = red[0][0] / red[1][0];                 // NaN for an empty mask, like the reference's mean of nothing
    dist[b] = d;
    reward[b] = prev ? (d < prev[b] ? 0.5f : 0.f) - (d > prev[b] ? 0.5f : 0.f) : 0.f;
  }
}

// out[r][i] = vals[r][i] + gamma * out[r][i + 1]   (one thread per row, T is a handful of steps)
__global__ void discounted_kernel(const float* __restrict__ vals, float* __restrict__ out, float gamma, int64_t rows, int T) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  float g = 0.f;
  for (int i = T - 1; i >= 0; --i) {
    g = vals[r * T + i] + gamma * g;
    out[r * T + i] = g;
  }
}

}  // namespace

extern "C" int cmr_expert_action_f32(const float* pose_source, const float* pose_target, const double* r_steps,
                                     const double* t_steps, int num_steps, int six_dof, int64_t* act_r, int64_t* act_t, int B,
                                     hipStream_t stream) {
  CMR_REQUIRE(pose_source && pose_target && r_steps && t_steps && act_r && act_t && B > 0 && num_steps > 0);
  hipLaunchKernelGGL(expert_action_kernel, dim3((B + 63) / 64), dim3(64), 0, stream, pose_source, pose_target, r_steps, t_steps,
                     num_steps, six_dof, act_r, act_t, B);
  return cmr_launch_status();
}

extern "C" int cmr_reward_f32(const float* pc, const float* pc_in_cam, const int64_t* mask, const float* prev_distance,
                              float* distance, float* reward, int B, int N, hipStream_t stream) {
  CMR_REQUIRE(pc && pc_in_cam && mask && distance && reward && B > 0 && B <= 65535 && N > 0);
  hipLaunchKernelGGL(reward_kernel, dim3(B), dim3(256), 0, stream, pc, pc_in_cam, mask, prev_distance, distance, reward, N);
  return cmr_launch_status();
}

extern "C" int cmr_discounted_f32(const float* vals, float* out, float gamma, int64_t rows, int T, hipStream_t stream) {
  CMR_REQUIRE(vals && out && rows > 0 && T > 0);
  hipLaunchKernelGGL(discounted_kernel, dim3((unsigned)((rows + 63) / 64)), dim3(64), 0, stream, vals, out, gamma, rows, T);
  return cmr_launch_status();
}
