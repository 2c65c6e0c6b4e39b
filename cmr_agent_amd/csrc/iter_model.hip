// IterModel (models/IterModel.py:24-475): the pose cost volume around the 3x3 convolution chain -- pose sampling, the warp of the
// predicted-overlap points under all n^3 sampled poses with scatter-mean of their features onto the 1/4-scale image grid, the
// one-channel halves of the first convolution, the global pool + 1x1 head, and the decision (softmax marginals, arg-maxes, cross
// entropy, the chosen step's inverse transform, the update of the cloud and of the accumulated pose).  The 3x3 convolutions
// themselves run on cmr_conv3x3_*_nhwc_f32 with the volume as a batch of n^3 maps (Conv3d kernels are (1, 3, 3)).
//
// HBM layout: acc / warped [P][h*w][64] (NHWC, the convolution's input), cnt / occ [P][h*w], rt [P][12], logits [P].
#include "cmr_common.h"

namespace {

// pose p = (i, j, k): rotation delta_R[i] about y, translation (delta_T[j], 0, delta_T[k]); rt[p] = rows 0..2 of its inverse
// (IterModel.py:132-173; the reference inverts the 4x4 numerically, here [R | t]^-1 = [R^T | -R^T t])
__global__ void iter_poses_kernel(const float* __restrict__ r_amp, const float* __restrict__ t_amp, int n, float* __restrict__ delta_r,
                                  float* __restrict__ delta_t, float* __restrict__ rt) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int P = n * n * n;
  const int lo = -((n - 1) / 2);
  const float dr = 2.f * r_amp[0] / (float)(n - 1), dt = 2.f * t_amp[0] / (float)(n - 1);
  if (p < n) {
    delta_r[p] = dr * (float)(lo + p);
    delta_t[p] = dt * (float)(lo + p);
  }
  if (p >= P) return;
  const int i = p / (n * n), j = (p / n) % n, k = p % n;
  const float a = dr * (float)(lo + i), tx = dt * (float)(lo + j), tz = dt * (float)(lo + k);
  const float s = sinf(a), c = cosf(a);
  float* o = rt + (int64_t)p * 12;
  // R = [[c, 0, s], [0, 1, 0], [-s, 0, c]]
  o[0] = c;   o[1] = 0.f; o[2] = -s;  o[3] = -(c * tx - s * tz);
  o[4] = 0.f; o[5] = 1.f; o[6] = 0.f; o[7] = 0.f;
  o[8] = s;   o[9] = 0.f; o[10] = c;  o[11] = -(s * tx + c * tz);
}

// sel = mask if it has any point, else standby (IterModel.py:273-275); one workgroup
__global__ __launch_bounds__(1024) void iter_mask_select_kernel(const uint8_t* __restrict__ mask, const uint8_t* __restrict__ standby,
                                                                 uint8_t* __restrict__ sel, int N) {
  __shared__ int any;
  if (threadIdx.x == 0) any = 0;
  __syncthreads();
  int mine = 0;
  for (int i = threadIdx.x; i < N; i += 1024) mine |= mask[i] != 0;
  if (mine) any = 1;
  __syncthreads();
  const uint8_t* src = any ? mask : standby;
  for (int i = threadIdx.x; i < N; i += 1024) sel[i] = src[i] != 0;
}

// One wave per 16 points of one pose, lane = channel: acc[p, y w + x, :] += feat[pt, :], cnt += 1, occ += score for the selected points
// that land in view (IterModel.py:277-345: warp, K, perspective divide, in-view test on the float coordinates, round half to even)
__global__ __launch_bounds__(256) void iter_warp_scatter_kernel(const float* __restrict__ pc /*[3][N]*/, const float* __restrict__ feat /*[N][64]*/,
                                                                const float* __restrict__ score, const uint8_t* __restrict__ sel,
                                                                const float* __restrict__ rt, const float* __restrict__ Kmat,
                                                                float* __restrict__ acc, float* __restrict__ cnt, float* __restrict__ occ,
                                                                int N, int h, int w) {
  const int lane = threadIdx.x & 63;
  const int p = blockIdx.y;
  const int n0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
  const float* R = rt + (int64_t)p * 12;
  const float r0 = R[0], r1 = R[1], r2 = R[2], r3 = R[3], r4 = R[4], r5 = R[5], r6 = R[6], r7 = R[7], r8 = R[8], r9 = R[9], r10 = R[10],
              r11 = R[11];
  const float k0 = Kmat[0], k1 = Kmat[1], k2 = Kmat[2], k3 = Kmat[3], k4 = Kmat[4], k5 = Kmat[5], k6 = Kmat[6], k7 = Kmat[7], k8 = Kmat[8];
  const int64_t cell0 = (int64_t)p * h * w;
  for (int q = 0; q < 16; ++q) {
    const int n = n0 + q;
    if (n >= N) return;
    if (!sel[n]) continue;
    const float x = pc[n], y = pc[N + n], z = pc[2 * (int64_t)N + n];
    const float tx = (r0 * x + r1 * y + r2 * z) + r3;
    const float ty = (r4 * x + r5 * y + r6 * z) + r7;
    const float tz = (r8 * x + r9 * y + r10 * z) + r11;
    float u = k0 * tx + k1 * ty + k2 * tz;
    float v = k3 * tx + k4 * ty + k5 * tz;
    const float zc = k6 * tx + k7 * ty + k8 * tz;
    u = u / zc;
    v = v / zc;
    if (!((u >= 0.f) && (u <= (float)(w - 1)) && (v >= 0.f) && (v <= (float)(h - 1)) && (zc > 0.f))) continue;
    const int64_t cell = cell0 + (int64_t)((int)rintf(v)) * w + (int)rintf(u);
    atomicAdd(acc + cell * 64 + lane, feat[(int64_t)n * 64 + lane]);
    if (lane == 0) atomicAdd(cnt + cell, 1.f);
    if (lane == 1) atomicAdd(occ + cell, score[n]);
  }
}

// The same warp + scatter-mean WITHOUT global atomics (they execute at the memory side at ~1.3 TB/s of added bytes, MI355X_MICROARCH.md:
// 459 M of them were 2.6 of the model's 7.4 ms) and without 64-lane LDS float atomics either (measured: ~240 cycles per ds_add_f32
// wave-instruction, 2.1 ms of a first version of this kernel).  A workgroup owns one pose and one BAND of whole map rows (384 cells x 64
// channels = 96 KB of LDS).  Per chunk of 4 096 points: (1) every thread projects 4 points (all workgroups of a pose project all points:
// 14 x redundant and still cheap), bins the scores of band + halo rows and the per-cell counts with single-lane LDS atomics and lists the
// points of the band; (2) a counting sort of the list by cell (prefix sum over the 384 counts, one integer LDS atomic per listed point);
// (3) wave k sums the feature rows of cells k, k + 16, ... in registers (lane = channel, four rows in flight) and adds the sum to its
// slab row with a plain read-modify-write -- the wave owns the cell.  The band is written ONCE, coalesced: the scatter mean (the
// convolution's NHWC input), the occupancy plane and -- since the band also holds the scores of its two halo rows -- the residual
// operand of the first convolution (image-half convolution + 3x3 stencil of the occupancy plane).  No memset, no accumulators in HBM, no
// separate finalisation pass.
constexpr int IB_THREADS = 1024, IB_WAVES = IB_THREADS / 64, IB_CELLS = 384, IB_CHUNK = 4096;
__global__ __launch_bounds__(IB_THREADS) void iter_warp_band_kernel(const float* __restrict__ pc /*[3][N]*/, const float* __restrict__ feat /*[N][64]*/,
                                                                    const float* __restrict__ score, const uint8_t* __restrict__ sel,
                                                                    const float* __restrict__ rt, const float* __restrict__ Kmat,
                                                                    const float* __restrict__ w1 /*[9][64]*/, const float* __restrict__ base /*[h*w][64]*/,
                                                                    float* __restrict__ warped, float* __restrict__ res, float* __restrict__ occ,
                                                                    int N, int h, int w, int band_rows) {
  extern __shared__ __attribute__((aligned(16))) float ib_smem[];
  float* slab = ib_smem;                               // [band_rows * w][64]
  float* cnt_s = slab + IB_CELLS * 64;                 // [band_rows * w] points per cell (all chunks)
  float* occ_s = cnt_s + IB_CELLS;                     // [(band_rows + 2) * w]: halo row above, the band, halo row below
  uint32_t* list = reinterpret_cast<uint32_t*>(occ_s + IB_CELLS + 2 * 384);   // [IB_CHUNK] (cell << 16) | point offset in the chunk
  uint16_t* sorted = reinterpret_cast<uint16_t*>(list + IB_CHUNK);            // [IB_CHUNK] point offsets grouped by cell
  int* ccnt = reinterpret_cast<int*>(sorted + IB_CHUNK);                      // [IB_CELLS] points per cell in this chunk
  int* coff = ccnt + IB_CELLS;                                                // [IB_CELLS + 1] exclusive prefix sum
  int* cfill = coff + IB_CELLS + 1;                                           // [IB_CELLS]
  __shared__ int nlist;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = blockIdx.y, r0 = blockIdx.x * band_rows;
  const int rows = r0 + band_rows <= h ? band_rows : h - r0;
  const int ncell = rows * w;
  for (int e = tid; e < IB_CELLS * 16; e += IB_THREADS) reinterpret_cast<f32x4*>(slab)[e] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int e = tid; e < IB_CELLS + IB_CELLS + 2 * 384; e += IB_THREADS) cnt_s[e] = 0.f;       // cnt_s and occ_s are adjacent
  const float* R = rt + (int64_t)p * 12;
  const float r0_ = R[0], r1 = R[1], r2 = R[2], r3 = R[3], r4 = R[4], r5 = R[5], r6 = R[6], r7 = R[7], r8 = R[8], r9 = R[9], r10 = R[10],
              r11 = R[11];
  const float k0 = Kmat[0], k1 = Kmat[1], k2 = Kmat[2], k3 = Kmat[3], k4 = Kmat[4], k5 = Kmat[5], k6 = Kmat[6], k7 = Kmat[7], k8 = Kmat[8];
  for (int c0 = 0; c0 < N; c0 += IB_CHUNK) {
    if (tid == 0) nlist = 0;
    for (int e = tid; e < IB_CELLS; e += IB_THREADS) { ccnt[e] = 0; cfill[e] = 0; }
    __syncthreads();
    // ---- (1) project this chunk's points (4 per thread, all their loads issued first), bin scores and counts, list the band's points
    {
      constexpr int PT = IB_CHUNK / IB_THREADS;
      float px[PT], py[PT], pz[PT], ps[PT];
      bool on[PT];
#pragma unroll
      for (int i = 0; i < PT; ++i) {
        const int n = c0 + tid + i * IB_THREADS;
        const int nc = n < N ? n : N - 1;
        on[i] = n < N && sel[nc] != 0;
        px[i] = pc[nc]; py[i] = pc[N + nc]; pz[i] = pc[2 * (int64_t)N + nc]; ps[i] = score[nc];
      }
#pragma unroll
      for (int i = 0; i < PT; ++i) {
        const float x = px[i], y = py[i], z = pz[i];
        const float tx = (r0_ * x + r1 * y + r2 * z) + r3;
        const float ty = (r4 * x + r5 * y + r6 * z) + r7;
        const float tz = (r8 * x + r9 * y + r10 * z) + r11;
        float u = k0 * tx + k1 * ty + k2 * tz;
        float v = k3 * tx + k4 * ty + k5 * tz;
        const float zc = k6 * tx + k7 * ty + k8 * tz;
        u = u / zc;
        v = v / zc;
        const bool in_view = on[i] && (u >= 0.f) && (u <= (float)(w - 1)) && (v >= 0.f) && (v <= (float)(h - 1)) && (zc > 0.f);
        const int yi = in_view ? (int)rintf(v) : -4, xi = in_view ? (int)rintf(u) : 0;
        const int ry = yi - (r0 - 1);                  // row inside band + halo
        if (in_view && ry >= 0 && ry <= rows + 1) {
          atomicAdd(&occ_s[ry * w + xi], ps[i]);
          if (ry >= 1 && ry <= rows) {
            const int cell = (ry - 1) * w + xi;
            atomicAdd(&cnt_s[cell], 1.f);
            atomicAdd(&ccnt[cell], 1);
            const int slot = atomicAdd(&nlist, 1);
            list[slot] = ((uint32_t)cell << 16) | (uint32_t)(tid + i * IB_THREADS);
          }
        }
      }
    }
    __syncthreads();
    // ---- (2) counting sort by cell: prefix sum of the 384 counts (one wave), then one integer atomic per listed point
    if (wave == 0) {
      int run = 0;
      for (int b0 = 0; b0 < IB_CELLS; b0 += 64) {
        const int c = ccnt[b0 + lane];
        int incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const int t = __shfl_up(incl, o);
          if (lane >= o) incl += t;
        }
        coff[b0 + lane] = run + incl - c;
        run += __shfl(incl, 63);
      }
      if (lane == 0) coff[IB_CELLS] = run;
    }
    __syncthreads();
    const int nl = nlist;
    for (int e = tid; e < nl; e += IB_THREADS) {
      const uint32_t it = list[e];
      const int cell = (int)(it >> 16);
      sorted[coff[cell] + atomicAdd(&cfill[cell], 1)] = (uint16_t)(it & 0xffffu);
    }
    __syncthreads();
    // ---- (3) wave k owns cells k, k + 16, ...: sum of their feature rows in registers, plain read-modify-write of the slab row
    for (int cell = wave; cell < ncell; cell += IB_WAVES) {
      const int b0 = coff[cell], b1 = coff[cell + 1];
      if (b0 == b1) continue;
      float acc = 0.f;
      for (int j = b0; j < b1; j += 4) {
        float fv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int jj = j + q < b1 ? j + q : b0;
          fv[q] = feat[(int64_t)(c0 + sorted[jj]) * 64 + lane];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (j + q < b1) acc += fv[q];
      }
      slab[cell * 64 + lane] += acc;
    }
    __syncthreads();
  }
  // ---- write the band: mean features, occupancy, residual operand (16 threads per cell, float4 each)
  const int64_t cell0 = (int64_t)p * h * w + (int64_t)r0 * w;
  for (int e = tid; e < ncell * 16; e += IB_THREADS) {
    const int cell = e >> 4, c = (e & 15) * 4;
    const float n = fmaxf(cnt_s[cell], 1.f);
    f32x4 a = *reinterpret_cast<const f32x4*>(&slab[cell * 64 + c]);
    a[0] /= n; a[1] /= n; a[2] /= n; a[3] /= n;
    *reinterpret_cast<f32x4*>(warped + (cell0 + cell) * 64 + c) = a;
    const int yy = cell / w, xx = cell - yy * w;       // row inside the band; occ_s row index = yy + 1
    f32x4 r = *reinterpret_cast<const f32x4*>(base + ((int64_t)(r0 * w) + cell) * 64 + c);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int xs = xx + t % 3 - 1;
      if (xs < 0 || xs >= w) continue;
      const float sv = occ_s[(yy + t / 3) * w + xs];   // rows outside the image never received a point: zero padding
      const f32x4 wv = *reinterpret_cast<const f32x4*>(w1 + t * 64 + c);
      r[0] += wv[0] * sv; r[1] += wv[1] * sv; r[2] += wv[2] * sv; r[3] += wv[3] * sv;
    }
    *reinterpret_cast<f32x4*>(res + (cell0 + cell) * 64 + c) = r;
    if ((e & 15) == 0) occ[cell0 + cell] = occ_s[(yy + 1) * w + xx];
  }
}

// 16 threads per cell: acc <- acc / max(cnt, 1) (scatter_mean) and res = base[cell] + sum_taps w1[tap][c] plane[p, cell + tap]: the
// one-channel input halves of the first convolution (occupancy per pose, the image overlap prediction once) added to its
// pose-independent image half, so that the matrix cores only see the 64 warped channels (IterModel.py:373-377 cat order)
__global__ __launch_bounds__(256) void iter_finalize_kernel(float* __restrict__ acc, const float* __restrict__ cnt, const float* __restrict__ plane,
                                                            const float* __restrict__ w1 /*[9][64]*/, const float* __restrict__ base /*[h*w][64]*/,
                                                            float* __restrict__ res, int P, int h, int w) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t cell = e >> 4;
  const int hw = h * w;
  if (cell >= (int64_t)P * hw) return;
  const int c = (int)(e & 15) * 4;
  const int pix = (int)(cell % hw);
  const int y = pix / w, x = pix % w;
  if (acc) {
    const float n = fmaxf(cnt[cell], 1.f);
    f32x4 a = *reinterpret_cast<const f32x4*>(acc + cell * 64 + c);
    a[0] /= n; a[1] /= n; a[2] /= n; a[3] /= n;
    *reinterpret_cast<f32x4*>(acc + cell * 64 + c) = a;
  }
  f32x4 r = *reinterpret_cast<const f32x4*>(base + (int64_t)pix * 64 + c);
  const float* pl = plane + (cell - pix);
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
    if (yy < 0 || yy >= h || xx < 0 || xx >= w) continue;
    const float s = pl[yy * w + xx];
    const f32x4 wv = *reinterpret_cast<const f32x4*>(w1 + t * 64 + c);
    r[0] += wv[0] * s; r[1] += wv[1] * s; r[2] += wv[2] * s; r[3] += wv[3] * s;
  }
  *reinterpret_cast<f32x4*>(res + cell * 64 + c) = r;
}

// one wave per pose: global average of the first 8 channels of x [P][cells][ldc], 1x1 conv 8 -> 4, LeakyReLU, 1x1 conv 4 -> 1
// (IterModel.py:62-66)
__global__ __launch_bounds__(256) void iter_head_kernel(const float* __restrict__ x, int ldc, int cells, const float* __restrict__ w24 /*[4][8]*/,
                                                        const float* __restrict__ b24, const float* __restrict__ w26 /*[4]*/,
                                                        const float* __restrict__ b26, float slope, float* __restrict__ logits, int P) {
  const int lane = threadIdx.x & 63;
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= P) return;
  float m[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int cell = lane; cell < cells; cell += 64) {
    const float* xp = x + ((int64_t)p * cells + cell) * ldc;
    const f32x4 a = *reinterpret_cast<const f32x4*>(xp), b = *reinterpret_cast<const f32x4*>(xp + 4);
    m[0] += a[0]; m[1] += a[1]; m[2] += a[2]; m[3] += a[3];
    m[4] += b[0]; m[5] += b[1]; m[6] += b[2]; m[7] += b[3];
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m[i] += __shfl_xor(m[i], o);
    m[i] *= 1.f / (float)cells;
  }
  if (lane != 0) return;
  float out = b26[0];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float hsum = b24[j];
#pragma unroll
    for (int i = 0; i < 8; ++i) hsum += w24[j * 8 + i] * m[i];
    out += w26[j] * (hsum > 0.f ? hsum : hsum * slope);
  }
  logits[p] = out;
}

// One workgroup: label = outer product of the three label rows and its arg-max, cross entropy of the logits against it, softmax,
// marginal sums over (ry), (tx), (tz) and their arg-maxes (lowest index on ties), arg-max of the joint, and the inverse transform
// of the chosen step (IterModel.py:175-193, 391-473).  out_f = [loss, ry, tx, tz]; out_i = [label, i_ry, i_tx, i_tz, i_joint].
constexpr int ITER_MAX_N = 16;
__global__ __launch_bounds__(256) void iter_decide_kernel(const float* __restrict__ logits, int n, const float* __restrict__ label_r,
                                                          const float* __restrict__ label_tx, const float* __restrict__ label_tz,
                                                          const float* __restrict__ delta_r, const float* __restrict__ delta_t,
                                                          float* __restrict__ label_out, float* __restrict__ out_f, int64_t* __restrict__ out_i,
                                                          float* __restrict__ matrix_i) {
  __shared__ float red[256];
  __shared__ int redi[256];
  __shared__ float marg[3][ITER_MAX_N];
  const int P = n * n * n, tid = threadIdx.x;
  auto argmax_block = [&](auto value) -> int {        // arg-max over 0..P-1 of value(p), lowest index on ties
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int p = tid; p < P; p += 256) {
      const float v = value(p);
      if (v > best) { best = v; bi = p; }
    }
    red[tid] = best; redi[tid] = bi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (tid < s && (red[tid + s] > red[tid] || (red[tid + s] == red[tid] && redi[tid + s] < redi[tid]))) {
        red[tid] = red[tid + s]; redi[tid] = redi[tid + s];
      }
      __syncthreads();
    }
    const int r = redi[0];
    __syncthreads();
    return r;
  };
  auto lab = [&](int p) { return label_r[p / (n * n)] * (label_tx[(p / n) % n] * label_tz[p % n]); };
  for (int p = tid; p < P; p += 256) label_out[p] = lab(p);
  const int label = argmax_block(lab);
  const int joint = argmax_block([&](int p) { return logits[p]; });
  const float mx = logits[joint];
  float se = 0.f;
  for (int p = tid; p < P; p += 256) se += expf(logits[p] - mx);
  red[tid] = se;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  const float denom = red[0];
  __syncthreads();
  // marginals of the softmax: thread (m, a) sums the n^2 entries with index a on axis m
  if (tid < 3 * n) {
    const int m = tid / n, a = tid % n;
    float s = 0.f;
    for (int b = 0; b < n; ++b)
      for (int c = 0; c < n; ++c) {
        const int p = m == 0 ? (a * n + b) * n + c : m == 1 ? (b * n + a) * n + c : (b * n + c) * n + a;
        s += expf(logits[p] - mx) / denom;
      }
    marg[m][a] = s;
  }
  __syncthreads();
  if (tid != 0) return;
  int im[3];
  for (int m = 0; m < 3; ++m) {
    int bi = 0;
    for (int a = 1; a < n; ++a)
      if (marg[m][a] > marg[m][bi]) bi = a;
    im[m] = bi;
  }
  const float ry = delta_r[im[0]], tx = delta_t[im[1]], tz = delta_t[im[2]];
  out_f[0] = (logf(denom) + mx) - logits[label];
  out_f[1] = ry; out_f[2] = tx; out_f[3] = tz;
  out_i[0] = label; out_i[1] = im[0]; out_i[2] = im[1]; out_i[3] = im[2]; out_i[4] = joint;
  const float s = sinf(ry), c = cosf(ry);
  const float mi[16] = {c, 0.f, -s, -(c * tx - s * tz), 0.f, 1.f, 0.f, 0.f, s, 0.f, c, -(s * tx + c * tz), 0.f, 0.f, 0.f, 1.f};
  for (int i = 0; i < 16; ++i) matrix_i[i] = mi[i];
}

// pc_out = M[0:3, 0:3] pc + M[0:3, 3] (planar [3][N]); thread 0 also writes acc_out = M acc_in (IterModel.py:466-472)
__global__ __launch_bounds__(256) void iter_apply_kernel(const float* __restrict__ M, const float* __restrict__ pc, float* __restrict__ pc_out, int N,
                                                         const float* __restrict__ acc_in, float* __restrict__ acc_out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i == 0) {
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c) {
        float s = 0.f;
        for (int k = 0; k < 4; ++k) s += M[r * 4 + k] * acc_in[k * 4 + c];
        acc_out[r * 4 + c] = s;
      }
  }
  if (i >= N) return;
  const float x = pc[i], y = pc[N + i], z = pc[2 * (int64_t)N + i];
  pc_out[i] = (M[0] * x + M[1] * y + M[2] * z) + M[3];
  pc_out[N + i] = (M[4] * x + M[5] * y + M[6] * z) + M[7];
  pc_out[2 * (int64_t)N + i] = (M[8] * x + M[9] * y + M[10] * z) + M[11];
}

}  // namespace

extern "C" int cmr_iter_sample_poses_f32(const float* r_amp, const float* t_amp, int nlabel, float* delta_r, float* delta_t, float* rt,
                                         hipStream_t stream) {
  CMR_REQUIRE(r_amp && t_amp && delta_r && delta_t && rt && nlabel >= 3 && nlabel <= ITER_MAX_N && nlabel % 2 == 1);
  const int P = nlabel * nlabel * nlabel;
  hipLaunchKernelGGL(iter_poses_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, r_amp, t_amp, nlabel, delta_r, delta_t, rt);
  return cmr_launch_status();
}

extern "C" int cmr_iter_warp_scatter_f32(const float* pc, const float* feat, const float* score, const uint8_t* mask, const uint8_t* standby,
                                         uint8_t* sel, const float* rt, const float* Kmat, float* acc, float* cnt, float* occ, int N, int P,
                                         int h, int w, hipStream_t stream) {
  CMR_REQUIRE(pc && feat && score && mask && standby && sel && rt && Kmat && acc && cnt && occ);
  CMR_REQUIRE(N > 0 && N < (1 << 24) && P > 0 && P <= 65535 && h > 0 && w > 0 && (int64_t)P * h * w < ((int64_t)1 << 31));
  const int64_t cells = (int64_t)P * h * w;
  if (hipMemsetAsync(acc, 0, cells * 64 * sizeof(float), stream) != hipSuccess) return CMR_ELAUNCH;
  if (hipMemsetAsync(cnt, 0, cells * sizeof(float), stream) != hipSuccess) return CMR_ELAUNCH;
  if (hipMemsetAsync(occ, 0, cells * sizeof(float), stream) != hipSuccess) return CMR_ELAUNCH;
  hipLaunchKernelGGL(iter_mask_select_kernel, dim3(1), dim3(1024), 0, stream, mask, standby, sel, N);
  hipLaunchKernelGGL(iter_warp_scatter_kernel, dim3((N + 63) / 64, P), dim3(256), 0, stream, pc, feat, score, sel, rt, Kmat, acc, cnt, occ,
                     N, h, w);
  return cmr_launch_status();
}

extern "C" int cmr_iter_warp_bin_f32(const float* pc, const float* feat, const float* score, const uint8_t* mask, const uint8_t* standby,
                                     uint8_t* sel, const float* rt, const float* Kmat, const float* w1, const float* base, float* warped,
                                     float* res, float* occ, int N, int P, int h, int w, hipStream_t stream) {
  CMR_REQUIRE(pc && feat && score && mask && standby && sel && rt && Kmat && w1 && base && warped && res && occ);
  CMR_REQUIRE(N > 0 && N < (1 << 24) && P > 0 && P <= 65535 && h > 0 && w > 0 && w <= IB_CELLS && (int64_t)P * h * w < ((int64_t)1 << 31));
  CMR_REQUIRE(cmr_aligned16(w1) && cmr_aligned16(base) && cmr_aligned16(warped) && cmr_aligned16(res) && cmr_aligned16(feat));
  const int band_rows = IB_CELLS / w;
  CMR_REQUIRE(band_rows * w < 65536);
  const size_t smem = (size_t)(IB_CELLS * 64 + IB_CELLS + IB_CELLS + 2 * 384) * sizeof(float) + IB_CHUNK * (sizeof(uint32_t) + sizeof(uint16_t)) +
                      (3 * IB_CELLS + 1) * sizeof(int);
  static CmrSmemCache granted{};
  if (cmr_grant_smem(reinterpret_cast<const void*>(iter_warp_band_kernel), smem, granted) != CMR_OK) return CMR_ELAUNCH;
  hipLaunchKernelGGL(iter_mask_select_kernel, dim3(1), dim3(1024), 0, stream, mask, standby, sel, N);
  hipLaunchKernelGGL(iter_warp_band_kernel, dim3((h + band_rows - 1) / band_rows, P), dim3(IB_THREADS), smem, stream, pc, feat, score, sel, rt,
                     Kmat, w1, base, warped, res, occ, N, h, w, band_rows);
  return cmr_launch_status();
}

extern "C" int cmr_iter_finalize_f32(float* acc, const float* cnt, const float* plane, const float* w1, const float* base, float* res, int P,
                                     int h, int w, hipStream_t stream) {
  CMR_REQUIRE(plane && w1 && base && res && P > 0 && h > 0 && w > 0 && (acc == nullptr) == (cnt == nullptr));
  CMR_REQUIRE(cmr_aligned16(w1) && cmr_aligned16(base) && cmr_aligned16(res) && (!acc || cmr_aligned16(acc)));
  const int64_t cells = (int64_t)P * h * w;
  CMR_REQUIRE(cells * 16 / 256 + 1 < 0x7fffffff);
  hipLaunchKernelGGL(iter_finalize_kernel, dim3((unsigned)((cells * 16 + 255) / 256)), dim3(256), 0, stream, acc, cnt, plane, w1, base, res,
                     P, h, w);
  return cmr_launch_status();
}

extern "C" int cmr_iter_head_f32(const float* x, int ldc, int cells, const float* w24, const float* b24, const float* w26, const float* b26,
                                 float slope, float* logits, int P, hipStream_t stream) {
  CMR_REQUIRE(x && w24 && b24 && w26 && b26 && logits && P > 0 && cells > 0 && ldc >= 8 && ldc % 4 == 0 && cmr_aligned16(x));
  hipLaunchKernelGGL(iter_head_kernel, dim3((P + 3) / 4), dim3(256), 0, stream, x, ldc, cells, w24, b24, w26, b26, slope, logits, P);
  return cmr_launch_status();
}

extern "C" int cmr_iter_decide_f32(const float* logits, int nlabel, const float* label_r, const float* label_tx, const float* label_tz,
                                   const float* delta_r, const float* delta_t, float* label_out, float* out_f, int64_t* out_i,
                                   float* matrix_i, hipStream_t stream) {
  CMR_REQUIRE(logits && label_r && label_tx && label_tz && delta_r && delta_t && label_out && out_f && out_i && matrix_i);
  CMR_REQUIRE(nlabel >= 3 && nlabel <= ITER_MAX_N);
  hipLaunchKernelGGL(iter_decide_kernel, dim3(1), dim3(256), 0, stream, logits, nlabel, label_r, label_tx, label_tz, delta_r, delta_t,
                     label_out, out_f, out_i, matrix_i);
  return cmr_launch_status();
}

extern "C" int cmr_iter_apply_f32(const float* matrix_i, const float* pc, float* pc_out, int N, const float* acc_in, float* acc_out,
                                  hipStream_t stream) {
  CMR_REQUIRE(matrix_i && pc && pc_out && acc_in && acc_out && N > 0);
  hipLaunchKernelGGL(iter_apply_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, matrix_i, pc, pc_out, N, acc_in, acc_out);
  return cmr_launch_status();
}
