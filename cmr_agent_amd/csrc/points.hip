// Point-cloud side kernels (all HBM / latency bound -- no MFMA here on purpose):
//   layout      planar [B,3,N] xyz -> row-major [B*N,4]; int64 per-batch indices -> int32 global rows
//   grouping    CSR of "points of each node" (stable, deterministic), kNN-16 on the nodes,
//               nearest row (node->proxy, point->node)
//   attention   vector-attention glue of GroupPointTransformer / KnnPointTransformer
//               (PointNN.py:149-185, 209-232): t = q[.] - k[.] + pos ; vp = v[.] + pos ; then the
//               per-segment, per-channel softmax + weighted sum (replaces 3 torch_scatter calls +
//               4 gathers per layer)
//   pointnet_util  farthest point sampling, ball query, square distance, index_points
//               (models/pointnet_util.py:19-93) with the module's exact arithmetic order
//   reductions  per-batch channel max / mean
// Distances use explicit __fmul_rn/__fadd_rn so that no FMA contraction changes the rounding
// relative to torch's sum((a-b)**2, -1): index results are then bit-identical except on exact ties.
#include "cmr_common.h"

// hipcc contracts a*b+c into fma by default (also through __fmul_rn/__fadd_rn, which are plain
// operators in the HIP headers); distances must round like torch's separate mul / add.
#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ float sqdist3(float ax, float ay, float az, float bx, float by, float bz) {
  // plain operators on purpose: the __f*_rn helpers are header functions compiled with the default
  // contract(fast) and get fused after inlining; these expressions are under this file's contract(off)
  const float dx = ax - bx, dy = ay - by, dz = az - bz;
  const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
  return (xx + yy) + zz;
}

// ---- layout ---------------------------------------------------------------------------------
// planar [B,C,N] -> rows [B*N, CP] (CP = 4 or 8, zero padded)
__global__ __launch_bounds__(256) void planar_to_rows_kernel(const float* __restrict__ x, float* __restrict__ y, int B,
                                                             int C, int N, int CP) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= (int64_t)B * N) return;
  const int b = (int)(r / N), n = (int)(r % N);
  for (int c0 = 0; c0 < CP; c0 += 4) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < 4; ++c)
      if (c0 + c < C) v[c] = x[((int64_t)b * C + c0 + c) * N + n];
    *reinterpret_cast<f32x4*>(y + r * CP + c0) = v;
  }
}

// out[r] = [ x1[r, :C1] | x2[map(r), :C2] ]   (materialised torch.cat, only where a consumer needs it)
__global__ __launch_bounds__(256) void concat_rows_kernel(const float* __restrict__ x1, int64_t ld1, int C1,
                                                          const float* __restrict__ x2, int64_t ld2, int C2,
                                                          const int32_t* __restrict__ idx2, int64_t div2,
                                                          float* __restrict__ out, int64_t rows) {
  const int c4n = (C1 + C2) / 4;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t r = e / c4n;
  if (r >= rows) return;
  const int c = (int)(e % c4n) * 4;
  f32x4 v;
  if (c < C1) v = *reinterpret_cast<const f32x4*>(x1 + r * ld1 + c);
  else {
    const int64_t s = idx2 ? (int64_t)idx2[r] : (div2 > 1 ? r / div2 : r);
    v = *reinterpret_cast<const f32x4*>(x2 + s * ld2 + (c - C1));
  }
  *reinterpret_cast<f32x4*>(out + r * (C1 + C2) + c) = v;
}

__global__ __launch_bounds__(256) void index_to_global_kernel(const int64_t* __restrict__ idx, int32_t* __restrict__ out,
                                                              int B, int N, int M) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= (int64_t)B * N) return;
  out[r] = (int32_t)(idx[r] + (r / N) * M);
}

// ---- CSR (segments = destination rows, members listed in ascending source order) ------------
// Counting sort: count the keys with atomics, scan, drop every source row into its segment through a per-segment cursor (any order),
// then ONE wave per segment puts its members into ascending order (rank by counting in LDS; segments are a handful to a few hundred
// rows).  Deterministic and stable like the round-1 build (one wave per segment scanning ALL keys of its batch, twice: 10 240 segments x
// 16 384 keys = 153 us at BASELINE configs[1]; 212 us per call at 65 536 points), at the cost of the keys themselves.  Keys outside
// their batch's segment range are ignored, as before.
__device__ __forceinline__ bool csr_key_ok(int32_t k, int b, int seg_per_batch) {
  return k >= b * seg_per_batch && k < (b + 1) * seg_per_batch;
}

__global__ __launch_bounds__(256) void csr_zero_kernel(int32_t* __restrict__ count, int total_seg) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < total_seg) count[i] = 0;
}

__global__ __launch_bounds__(256) void csr_count_kernel(const int32_t* __restrict__ key, int32_t* __restrict__ count,
                                                        int n_per_batch, int seg_per_batch, int64_t total_keys) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total_keys) return;
  const int32_t k = key[i];
  if (csr_key_ok(k, (int)(i / n_per_batch), seg_per_batch)) atomicAdd(count + k, 1);
}

__global__ __launch_bounds__(1024) void exclusive_scan_kernel(const int32_t* __restrict__ count,
                                                              int32_t* __restrict__ offsets, int n) {
  // single block; offsets has n+1 entries
  __shared__ int32_t sums[1024];
  const int tid = threadIdx.x;
  const int per = (n + 1023) / 1024;
  const int lo = tid * per, hi = min(n, lo + per);
  int s = 0;
  for (int i = lo; i < hi; ++i) s += count[i];
  sums[tid] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    int v = tid >= off ? sums[tid - off] : 0;
    __syncthreads();
    sums[tid] += v;
    __syncthreads();
  }
  int run = sums[tid] - s;
  for (int i = lo; i < hi; ++i) { offsets[i] = run; run += count[i]; }
  if (tid == 1023) offsets[n] = sums[1023];
}

// every source row takes the next free slot of its segment (count[] runs down to zero: it is the cursor)
__global__ __launch_bounds__(256) void csr_scatter_kernel(const int32_t* __restrict__ key, const int32_t* __restrict__ offsets,
                                                          int32_t* __restrict__ count, int32_t* __restrict__ order, int n_per_batch,
                                                          int seg_per_batch, int64_t total_keys) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total_keys) return;
  const int32_t k = key[i];
  if (!csr_key_ok(k, (int)(i / n_per_batch), seg_per_batch)) return;
  const int slot = atomicSub(count + k, 1) - 1;
  order[offsets[k] + slot] = (int32_t)i;
}

// one wave per segment: members into ascending order (they are distinct row numbers: rank = how many are smaller), count[] restored
constexpr int CSR_SORT_CAP = 1024;
__global__ __launch_bounds__(256) void csr_sort_kernel(const int32_t* __restrict__ key, const int32_t* __restrict__ offsets,
                                                       int32_t* __restrict__ count, int32_t* __restrict__ order, int n_per_batch,
                                                       int seg_per_batch, int total_seg) {
  __shared__ int32_t buf[4][CSR_SORT_CAP];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int seg = blockIdx.x * 4 + w;
  if (seg >= total_seg) return;
  const int lo = offsets[seg], n = offsets[seg + 1] - lo;
  if (lane == 0) count[seg] = n;
  if (n <= 1) return;
  if (n <= CSR_SORT_CAP) {
    volatile int32_t* mine = buf[w];
    for (int j = lane; j < n; j += 64) mine[j] = order[lo + j];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int j = lane; j < n; j += 64) {
      const int32_t x = mine[j];
      int rank = 0;
      for (int q = 0; q < n; ++q) rank += mine[q] < x;
      order[lo + rank] = x;
    }
    return;
  }
  // a segment longer than the LDS buffer: the round-1 way for this segment (scan the batch's keys in order)
  const int b = seg / seg_per_batch;
  const int32_t* kb = key + (int64_t)b * n_per_batch;
  int pos = lo;
  for (int i0 = 0; i0 < n_per_batch; i0 += 64) {
    const int i = i0 + lane;
    const bool hit = i < n_per_batch && kb[i] == seg;
    const unsigned long long mask = __ballot(hit);
    if (hit) order[pos + __popcll(mask & ((1ull << lane) - 1ull))] = (int32_t)((int64_t)b * n_per_batch + i);
    pos += __popcll(mask);
  }
}

// ---- kNN (k = 16) among the rows of each batch ------------------------------------------------------------------
// Candidates of the batch element in LDS; a query is handled by KNN_SPLIT = 8 adjacent lanes: lane `sub` scans the
// candidates sub, sub + 8, ... (ascending index) into its own sorted top-K, then the 8 lists are merged by K rounds of
// "take the lexicographic (distance, index) minimum of the list heads" -- the same result as one serial scan with
// "on ties the smaller index stays ahead" (torch.argsort of the reference on the fixtures), at 8 x the parallelism
// (one thread per query left 216 of the 256 CUs idle for 0.75 ms).
constexpr int KNN_SPLIT = 8;
template <int K>
__global__ __launch_bounds__(256) void knn_kernel(const float* __restrict__ xyz4, int32_t* __restrict__ out, int M) {
  extern __shared__ __attribute__((aligned(16))) float cand[];  // [M][4]
  const int b = blockIdx.y;
  const float* xb = xyz4 + (int64_t)b * M * 4;
  for (int i = threadIdx.x; i < M; i += 256)
    *reinterpret_cast<f32x4*>(&cand[i * 4]) = *reinterpret_cast<const f32x4*>(xb + (int64_t)i * 4);
  __syncthreads();
  const int sub = threadIdx.x & (KNN_SPLIT - 1);
  const int q0 = blockIdx.x * (256 / KNN_SPLIT) + threadIdx.x / KNN_SPLIT;
  const int qi = q0 < M ? q0 : M - 1;               // keep the lanes of a partial group alive for the shuffles
  const f32x4 qp = *reinterpret_cast<const f32x4*>(&cand[qi * 4]);
  float bd[K];
  int bi[K];
#pragma unroll
  for (int j = 0; j < K; ++j) { bd[j] = INFINITY; bi[j] = 0x7fffffff; }
  for (int c = sub; c < M; c += KNN_SPLIT) {
    const f32x4 cp = *reinterpret_cast<const f32x4*>(&cand[c * 4]);
    float d = sqdist3(qp[0], qp[1], qp[2], cp[0], cp[1], cp[2]);
    if (d < bd[K - 1]) {     // strict: on ties the smaller index (seen first) stays ahead
      int id = c;
      bool ins = false;      // once the new element has gone in, everything behind it SHIFTS (equal neighbours keep their order)
#pragma unroll
      for (int j = 0; j < K; ++j) {
        const bool sw = ins || d < bd[j];
        ins = sw;
        const float td = sw ? bd[j] : d; const int ti = sw ? bi[j] : id;
        bd[j] = sw ? d : bd[j]; bi[j] = sw ? id : bi[j];
        d = td; id = ti;
      }
    }
  }
  int res[K / KNN_SPLIT];
#pragma unroll
  for (int r = 0; r < K; ++r) {
    float dm = bd[0];
    int im = bi[0];
#pragma unroll
    for (int x = 1; x < KNN_SPLIT; x <<= 1) {
      const float od = __shfl_xor(dm, x);
      const int oi = __shfl_xor(im, x);
      const bool take = od < dm || (od == dm && oi < im);
      dm = take ? od : dm;
      im = take ? oi : im;
    }
    if (bi[0] == im) {                              // this lane's head won: pop it
#pragma unroll
      for (int j = 0; j + 1 < K; ++j) { bd[j] = bd[j + 1]; bi[j] = bi[j + 1]; }
      bd[K - 1] = INFINITY; bi[K - 1] = 0x7fffffff;
    }
    if ((r & (KNN_SPLIT - 1)) == sub) res[r / KNN_SPLIT] = im;
  }
  if (q0 < M) {
#pragma unroll
    for (int t = 0; t < K / KNN_SPLIT; ++t)
      out[((int64_t)b * M + q0) * K + t * KNN_SPLIT + sub] = (int32_t)((int64_t)b * M + res[t]);
  }
}

// ---- general kNN: queries != candidates, K <= 64 (pointnet_util.py:114-116, 232-234: square_distance + argsort()[:, :, :K]) ------------
// The same scheme as knn_kernel with the candidates streamed through LDS in tiles (clouds of 65 536 points do not fit): a query is handled
// by KNN_SPLIT adjacent lanes, lane `sub` scans the candidates sub, sub + 8, ... of every tile (ascending index across tiles) into its own
// sorted top-KT, then the 8 lists are merged by K rounds of "lexicographic (distance, index) minimum of the heads" = a stable argsort's
// first K.  A candidate count below K pads with the index 0x7fffffff -> reported as -1 (the reference would raise on such a slice).
constexpr int KNN_TILE = 2048;
template <int KT>
__global__ __launch_bounds__(256) void knn_general_kernel(const float* __restrict__ q4, const float* __restrict__ c4, int64_t* __restrict__ out,
                                                           int S, int N, int K) {
  __shared__ __attribute__((aligned(16))) float tile[KNN_TILE * 4];
  const int b = blockIdx.y;
  const int sub = threadIdx.x & (KNN_SPLIT - 1);
  const int q0 = blockIdx.x * (256 / KNN_SPLIT) + threadIdx.x / KNN_SPLIT;
  const int qi = q0 < S ? q0 : S - 1;
  const f32x4 qp = *reinterpret_cast<const f32x4*>(q4 + ((int64_t)b * S + qi) * 4);
  float bd[KT];
  int bi[KT];
#pragma unroll
  for (int j = 0; j < KT; ++j) { bd[j] = INFINITY; bi[j] = 0x7fffffff; }
  for (int c0 = 0; c0 < N; c0 += KNN_TILE) {
    const int n = min(KNN_TILE, N - c0);
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256)
      *reinterpret_cast<f32x4*>(&tile[i * 4]) = *reinterpret_cast<const f32x4*>(c4 + ((int64_t)b * N + c0 + i) * 4);
    __syncthreads();
    for (int c = sub; c < n; c += KNN_SPLIT) {
      const f32x4 cp = *reinterpret_cast<const f32x4*>(&tile[c * 4]);
      float d = sqdist3(qp[0], qp[1], qp[2], cp[0], cp[1], cp[2]);
      if (d < bd[KT - 1]) {     // strict: on ties the smaller index (seen first) stays ahead
        int id = c0 + c;
        bool ins = false;       // once the new element has gone in, everything behind it SHIFTS (a carried element that equals its successor must not hop over it)
#pragma unroll
        for (int j = 0; j < KT; ++j) {
          const bool sw = ins || d < bd[j];
          ins = sw;
          const float td = sw ? bd[j] : d; const int ti = sw ? bi[j] : id;
          bd[j] = sw ? d : bd[j]; bi[j] = sw ? id : bi[j];
          d = td; id = ti;
        }
      }
    }
  }
  int64_t* o = out + ((int64_t)b * S + qi) * K;
  for (int r = 0; r < K; ++r) {
    float dm = bd[0];
    int im = bi[0];
#pragma unroll
    for (int x = 1; x < KNN_SPLIT; x <<= 1) {
      const float od = __shfl_xor(dm, x);
      const int oi = __shfl_xor(im, x);
      const bool take = od < dm || (od == dm && oi < im);
      dm = take ? od : dm;
      im = take ? oi : im;
    }
    if (bi[0] == im && im != 0x7fffffff) {          // this lane's head won: pop it
#pragma unroll
      for (int j = 0; j + 1 < KT; ++j) { bd[j] = bd[j + 1]; bi[j] = bi[j + 1]; }
      bd[KT - 1] = INFINITY; bi[KT - 1] = 0x7fffffff;
    }
    if (sub == (r & (KNN_SPLIT - 1)) && q0 < S) o[r] = im == 0x7fffffff ? (int64_t)-1 : (int64_t)im;
  }
}

// nearest candidate row for every query row (first index on ties); candidates tiled through LDS
__global__ __launch_bounds__(256) void nearest_kernel(const float* __restrict__ q4, const float* __restrict__ c4,
                                                      int32_t* __restrict__ out_global, int64_t* __restrict__ out_local,
                                                      int Nq, int Nc) {
  __shared__ __attribute__((aligned(16))) float tile[1024 * 4];
  const int b = blockIdx.y;
  const int qi = blockIdx.x * 256 + threadIdx.x;
  f32x4 qp = {0.f, 0.f, 0.f, 0.f};
  if (qi < Nq) qp = *reinterpret_cast<const f32x4*>(q4 + ((int64_t)b * Nq + qi) * 4);
  float best = INFINITY;
  int besti = 0;
  for (int c0 = 0; c0 < Nc; c0 += 1024) {
    const int n = min(1024, Nc - c0);
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256)
      *reinterpret_cast<f32x4*>(&tile[i * 4]) = *reinterpret_cast<const f32x4*>(c4 + ((int64_t)b * Nc + c0 + i) * 4);
    __syncthreads();
    for (int i = 0; i < n; ++i) {
      const f32x4 cp = *reinterpret_cast<const f32x4*>(&tile[i * 4]);
      const float d = sqdist3(qp[0], qp[1], qp[2], cp[0], cp[1], cp[2]);
      if (d < best) { best = d; besti = c0 + i; }
    }
  }
  if (qi < Nq) {
    if (out_global) out_global[(int64_t)b * Nq + qi] = (int32_t)((int64_t)b * Nc + besti);
    if (out_local) out_local[(int64_t)b * Nq + qi] = besti;
  }
}

// three nearest candidates per query (ascending distance) + normalised inverse-distance weights
// (PointNetFeaturePropagation, pointnet_util.py:287-295): idx int32 GLOBAL rows [B*Nq,3], w [B*Nq,3]
__global__ __launch_bounds__(256) void three_nn_kernel(const float* __restrict__ q4, const float* __restrict__ c4,
                                                       int32_t* __restrict__ idx, float* __restrict__ wgt, int Nq, int Nc) {
  __shared__ __attribute__((aligned(16))) float tile[1024 * 4];
  const int b = blockIdx.y;
  const int qi = blockIdx.x * 256 + threadIdx.x;
  f32x4 qp = {0.f, 0.f, 0.f, 0.f};
  if (qi < Nq) qp = *reinterpret_cast<const f32x4*>(q4 + ((int64_t)b * Nq + qi) * 4);
  float d0 = INFINITY, d1 = INFINITY, d2 = INFINITY;
  int i0 = 0, i1 = 0, i2 = 0;
  for (int c0 = 0; c0 < Nc; c0 += 1024) {
    const int n = min(1024, Nc - c0);
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256)
      *reinterpret_cast<f32x4*>(&tile[i * 4]) = *reinterpret_cast<const f32x4*>(c4 + ((int64_t)b * Nc + c0 + i) * 4);
    __syncthreads();
    for (int i = 0; i < n; ++i) {
      const f32x4 cp = *reinterpret_cast<const f32x4*>(&tile[i * 4]);
      const float d = sqdist3(qp[0], qp[1], qp[2], cp[0], cp[1], cp[2]);
      const int id = c0 + i;
      if (d < d2) {
        if (d < d1) {
          d2 = d1; i2 = i1;
          if (d < d0) { d1 = d0; i1 = i0; d0 = d; i0 = id; }
          else { d1 = d; i1 = id; }
        } else { d2 = d; i2 = id; }
      }
    }
  }
  if (qi < Nq) {
    const float r0 = 1.0f / (d0 + 1e-8f), r1 = 1.0f / (d1 + 1e-8f), r2 = 1.0f / (d2 + 1e-8f);
    const float norm = (r0 + r1) + r2;
    const int64_t o = ((int64_t)b * Nq + qi) * 3;
    idx[o] = (int32_t)((int64_t)b * Nc + i0); idx[o + 1] = (int32_t)((int64_t)b * Nc + i1); idx[o + 2] = (int32_t)((int64_t)b * Nc + i2);
    wgt[o] = r0 / norm; wgt[o + 1] = r1 / norm; wgt[o + 2] = r2 / norm;
  }
}

// out[r, c] = sum_j w[r, j] * src[idx[r, j], c], j = 0..2
__global__ __launch_bounds__(256) void weighted_gather3_kernel(const float* __restrict__ src, int64_t lds,
                                                               const int32_t* __restrict__ idx, const float* __restrict__ wgt,
                                                               float* __restrict__ out, int64_t ldo, int64_t rows, int C) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t r = e / C;
  if (r >= rows) return;
  const int c = (int)(e % C);
  const float a0 = src[(int64_t)idx[r * 3] * lds + c] * wgt[r * 3];
  const float a1 = src[(int64_t)idx[r * 3 + 1] * lds + c] * wgt[r * 3 + 1];
  const float a2 = src[(int64_t)idx[r * 3 + 2] * lds + c] * wgt[r * 3 + 2];
  out[r * ldo + c] = (a0 + a1) + a2;
}

// backward of weighted_gather3 w.r.t. src: dsrc[t, c] = sum over the entries e = 3 r + j with idx[e] == t of w[e] * dy[r, c], entries taken
// in ascending e through the CSR (offsets, order) of idx over the target rows: no atomics, the same sum on every run
__global__ __launch_bounds__(256) void weighted_scatter3_kernel(const float* __restrict__ dy, int64_t ldd, const float* __restrict__ wgt,
                                                                const int32_t* __restrict__ order, const int32_t* __restrict__ offsets,
                                                                float* __restrict__ out, int64_t ldo, int64_t nseg, int C) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t t = e / C;
  if (t >= nseg) return;
  const int c = (int)(e % C);
  float acc = 0.f;
  for (int j = offsets[t]; j < offsets[t + 1]; ++j) {
    const int en = order[j];
    acc += wgt[en] * dy[(int64_t)(en / 3) * ldd + c];
  }
  out[t * ldo + c] = acc;
}

// ---- row gathers ------------------------------------------------------------------------------
__device__ __forceinline__ int64_t map_row(const int32_t* idx, int64_t div, int64_t r) {
  return idx ? (int64_t)idx[r] : (div > 1 ? r / div : r);
}

// out[r, :4] = a[map_a(r)] - b[map_b(r)]         (relative positions; 4th lane stays 0)
__global__ __launch_bounds__(256) void rel_pos_kernel(const float* __restrict__ a, const int32_t* __restrict__ ia,
                                                      int64_t diva, const float* __restrict__ b,
                                                      const int32_t* __restrict__ ib, int64_t divb, float* __restrict__ out,
                                                      int64_t rows) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  const f32x4 av = *reinterpret_cast<const f32x4*>(a + map_row(ia, diva, r) * 4);
  const f32x4 bv = *reinterpret_cast<const f32x4*>(b + map_row(ib, divb, r) * 4);
  f32x4 o = {av[0] - bv[0], av[1] - bv[1], av[2] - bv[2], 0.f};
  *reinterpret_cast<f32x4*>(out + r * 4) = o;
}

// t[r] = q[map_q(r)] - k[map_k(r)] + pos[r];  vp[r] = v[map_k(r)] + pos[r]      (64 channels)
__global__ __launch_bounds__(256) void vecattn_prep_kernel(const float* __restrict__ q, int64_t ldq,
                                                           const int32_t* __restrict__ iq, int64_t divq,
                                                           const float* __restrict__ k, int64_t ldk,
                                                           const float* __restrict__ v, int64_t ldv,
                                                           const int32_t* __restrict__ ik, const float* __restrict__ pos,
                                                           float* __restrict__ t, float* __restrict__ vp, int64_t rows) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t r = e >> 4;
  if (r >= rows) return;
  const int c = (int)(e & 15) * 4;
  const int64_t rq = map_row(iq, divq, r), rk = ik ? (int64_t)ik[r] : r;
  const f32x4 qv = *reinterpret_cast<const f32x4*>(q + rq * ldq + c);
  const f32x4 kv = *reinterpret_cast<const f32x4*>(k + rk * ldk + c);
  const f32x4 vv = *reinterpret_cast<const f32x4*>(v + rk * ldv + c);
  const f32x4 pv = *reinterpret_cast<const f32x4*>(pos + r * 64 + c);
  f32x4 to, vo;
#pragma unroll
  for (int i = 0; i < 4; ++i) { to[i] = qv[i] - kv[i] + pv[i]; vo[i] = vv[i] + pv[i]; }
  *reinterpret_cast<f32x4*>(t + r * 64 + c) = to;
  *reinterpret_cast<f32x4*>(vp + r * 64 + c) = vo;
}

// out[s, c] = sum_i softmax_i(attn[row_i, c] * scale) * vp[row_i, c] over the rows of segment s
// one wave per segment, lane = channel (64)
__global__ __launch_bounds__(256) void segment_softmax_kernel(const float* __restrict__ attn, const float* __restrict__ vp,
                                                              const int32_t* __restrict__ order,
                                                              const int32_t* __restrict__ offsets, int fixed_len,
                                                              float scale, float* __restrict__ out, int64_t nseg) {
  const int64_t seg = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (seg >= nseg) return;
  const int64_t lo = offsets ? offsets[seg] : seg * fixed_len;
  const int64_t hi = offsets ? offsets[seg + 1] : lo + fixed_len;
  // Eight members of the segment in flight per pass (their row indices first, then the rows): the one-member-per-iteration loops left
  // every wave with a single 256-byte load outstanding (2-3 TB/s over the 524 288-row maps of the training step); same order of
  // operations, so the results are bit-identical.
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
  out[seg * 64 + lane] = hi > lo ? acc / l : 0.f;
}

// torch_scatter semantics over CSR segments: out[s, c] = reduce_i src[row_i, c] over the rows of segment s
// (mode 0 sum, 1 max, 2 mean = sum / max(count, 1); empty segments give 0, like torch_scatter's zero-filled
// output).  One wave per segment, lanes stride over channels.
__global__ __launch_bounds__(256) void segment_reduce_kernel(const float* __restrict__ src, int64_t lds,
                                                             const int32_t* __restrict__ order,
                                                             const int32_t* __restrict__ offsets, float* __restrict__ out,
                                                             int64_t ldo, int64_t nseg, int C, int mode) {
  const int64_t seg = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (seg >= nseg) return;
  const int64_t lo = offsets[seg], hi = offsets[seg + 1];
  for (int c = lane; c < C; c += 64) {
    float acc = mode == 1 ? -INFINITY : 0.f;
    int64_t i = lo;
    for (; i + 8 <= hi; i += 8) {                        // eight member rows in flight (same order of the sums)
      int64_t r[8];
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) r[u] = order[i + u];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[r[u] * lds + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = mode == 1 ? fmaxf(acc, v[u]) : acc + v[u];
    }
    for (; i < hi; ++i) {
      const float v = src[(int64_t)order[i] * lds + c];
      acc = mode == 1 ? fmaxf(acc, v) : acc + v;
    }
    if (hi == lo) acc = 0.f;
    else if (mode == 2) acc = acc / (float)(hi - lo);
    out[seg * ldo + c] = acc;
  }
}

// out[r, :C] = src[idx[r], :C]      (pointnet_util.index_points / torch.gather of rows)
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, int64_t lds,
                                                          const int32_t* __restrict__ idx, float* __restrict__ out,
                                                          int64_t ldo, int64_t rows, int C) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t r = e / C;
  if (r >= rows) return;
  const int c = (int)(e % C);
  out[r * ldo + c] = src[(int64_t)idx[r] * lds + c];
}

__global__ __launch_bounds__(256) void gather_rows4_kernel(const float* __restrict__ src, int64_t lds, const int32_t* __restrict__ idx,
                                                           float* __restrict__ out, int64_t ldo, int64_t rows, int Q) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t r = e / Q;
  if (r >= rows) return;
  const int c = (int)(e % Q) * 4;
  *reinterpret_cast<f32x4*>(out + r * ldo + c) = *reinterpret_cast<const f32x4*>(src + (int64_t)idx[r] * lds + c);
}

// ---- pointnet_util --------------------------------------------------------------------------
// Farthest point sampling: one workgroup (1024 threads) per cloud, running min-distance in
// registers, argmax by wave shuffle + LDS across the 16 waves (lowest index wins ties, like
// torch.max).  xyz4 [B,N,4]; start [B]; out [B,npoint] int64 (local indices).
constexpr int FPS_T = 1024;
// start index of a cloud, forced into [0, N): the reference indexes xyz[start] and raises for an index outside the cloud
// (pointnet_util.py:62-66); a kernel cannot raise, and must not read outside the cloud either.
__device__ __forceinline__ int fps_start(int64_t s, int N) { return s < 0 ? 0 : (s >= N ? N - 1 : (int)s); }
template <int PER, bool CACHE>
__global__ __launch_bounds__(FPS_T) void fps_kernel(const float* __restrict__ xyz4, const int64_t* __restrict__ start,
                                                    int64_t* __restrict__ out, int N, int npoint) {
  __shared__ float red_d[16];
  __shared__ int red_i[16];
  __shared__ int far_s;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* xb = xyz4 + (int64_t)b * N * 4;
  // coordinates stay in registers for clouds up to 16K points; larger clouds re-read them (L2)
  constexpr int PC = CACHE ? PER : 1;
  float px[PC], py[PC], pz[PC], dist[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int i = tid + j * FPS_T;
    dist[j] = i < N ? 1e10f : -1.f;
    if (CACHE) {
      f32x4 p = {0.f, 0.f, 0.f, 0.f};
      if (i < N) p = *reinterpret_cast<const f32x4*>(xb + (int64_t)i * 4);
      px[j] = p[0]; py[j] = p[1]; pz[j] = p[2];
    }
  }
  int far = fps_start(start[b], N);
  for (int it = 0; it < npoint; ++it) {
    if (tid == 0) out[(int64_t)b * npoint + it] = far;
    const f32x4 c = *reinterpret_cast<const f32x4*>(xb + (int64_t)far * 4);
    float bd = -1.f;
    int bi = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int i = tid + j * FPS_T;
      if (i < N) {
        float qx, qy, qz;
        if (CACHE) { qx = px[j]; qy = py[j]; qz = pz[j]; }
        else { const f32x4 p = *reinterpret_cast<const f32x4*>(xb + (int64_t)i * 4); qx = p[0]; qy = p[1]; qz = p[2]; }
        const float d = sqdist3(qx, qy, qz, c[0], c[1], c[2]);
        dist[j] = fminf(dist[j], d);
        if (dist[j] > bd) { bd = dist[j]; bi = i; }   // ascending i within a thread: first max kept
      }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      const float od = __shfl_xor(bd, m);
      const int oi = __shfl_xor(bi, m);
      if (od > bd || (od == bd && oi < bi)) { bd = od; bi = oi; }
    }
    if (lane == 0) { red_d[wave] = bd; red_i[wave] = bi; }
    __syncthreads();
    if (wave == 0) {
      float d2 = lane < 16 ? red_d[lane] : -2.f;
      int i2 = lane < 16 ? red_i[lane] : 0x7fffffff;
#pragma unroll
      for (int m = 8; m >= 1; m >>= 1) {
        const float od = __shfl_xor(d2, m);
        const int oi = __shfl_xor(i2, m);
        if (od > d2 || (od == d2 && oi < i2)) { d2 = od; i2 = oi; }
      }
      if (lane == 0) far_s = i2;
    }
    __syncthreads();
    far = far_s;
  }
}


// Farthest point sampling for clouds that do not fit one workgroup's registers (> 16 384 points; BASELINE configs[4] = 65 536):
// G workgroups of 256 threads per cloud, each keeping its slice (coordinates + running min-distance, PER points per thread)
// in registers.  A round = local arg-max (wave shuffles + LDS), then every workgroup PUBLISHES its candidate as one 64-bit word in
// its own slot of the round (key = distance bits << 32 | ~index: distances are >= 0, so the unsigned order of the bits is the float
// order, and among equal distances the lowest index wins, like torch.max; a workgroup without a candidate publishes 1, which loses
// against every key) and the first wave of every workgroup polls the round's G slots -- one 8 G-byte load -- until none is zero, then
// takes the maximum itself.  Two dependent memory round trips per round (the poll that sees the last slot, the winner's coordinates)
// where the first version had five (atomic max on a shared word, release increment of an arrival counter, poll of the counter, load
// of the word, coordinates): 3.8 -> 2.x us per round at 8 x 65 536 points.  Every round has its own slots (zeroed by the entry point),
// so nothing is ever reset; fps_unpack_kernel turns them into plain indices afterwards.  The G workgroups of a cloud must be
// co-resident (the entry point sizes G so that B * G <= 256); a workgroup that polls longer than FPS_SPIN times raises the error
// word and leaves, and everybody who sees the error word leaves too: the grid always drains.
constexpr int FPS_CT = 256;
constexpr unsigned FPS_SPIN = 1u << 22;
template <int PER>
__global__ __launch_bounds__(FPS_CT) void fps_coop_kernel(const float* __restrict__ xyz4, const int64_t* __restrict__ start,
                                                          unsigned long long* __restrict__ win /*[B][npoint][G]*/,
                                                          unsigned* __restrict__ cnt /*[B] unused, then [B] error flags*/, int B, int G,
                                                          int N, int npoint, unsigned spin_limit) {
  __shared__ float red_d[4];
  __shared__ int red_i[4];
  __shared__ int far_s;
  const int b = blockIdx.x % B, w = blockIdx.x / B;             // the workgroups of cloud b are blockIdx = b (mod B): one XCD when B = 8
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* xb = xyz4 + (int64_t)b * N * 4;
  const int slice = (N + G - 1) / G;
  const int lo = w * slice, hi = min(N, lo + slice);
  float px[PER], py[PER], pz[PER], dist[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int i = lo + tid + j * FPS_CT;
    f32x4 p = {0.f, 0.f, 0.f, 0.f};
    if (i < hi) p = *reinterpret_cast<const f32x4*>(xb + (int64_t)i * 4);
    px[j] = p[0]; py[j] = p[1]; pz[j] = p[2];
    dist[j] = i < hi ? 1e10f : -1.f;
  }
  unsigned long long* wb = win + (int64_t)b * npoint * G;
  unsigned* err = cnt + B + b;
  int far = fps_start(start[b], N);
  if (w == 0 && tid == 0) wb[0] = (unsigned long long)far;        // round 0's first slot holds the start index as it is
  for (int it = 0; it + 1 < npoint; ++it) {
    const f32x4 c = *reinterpret_cast<const f32x4*>(xb + (int64_t)far * 4);
    float bd = -1.f;
    int bi = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int i = lo + tid + j * FPS_CT;
      const float d = sqdist3(px[j], py[j], pz[j], c[0], c[1], c[2]);
      dist[j] = i < hi ? fminf(dist[j], d) : -1.f;
      if (dist[j] > bd) { bd = dist[j]; bi = i; }             // ascending i within a thread: first max kept
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      const float od = __shfl_xor(bd, m);
      const int oi = __shfl_xor(bi, m);
      if (od > bd || (od == bd && oi < bi)) { bd = od; bi = oi; }
    }
    if (lane == 0) { red_d[wave] = bd; red_i[wave] = bi; }
    __syncthreads();
    if (wave == 0) {                                            // (uniform per wave) the first wave publishes and polls
      unsigned long long* slots = wb + (int64_t)(it + 1) * G;
      if (lane == 0) {
#pragma unroll
        for (int k = 1; k < 4; ++k)
          if (red_d[k] > bd || (red_d[k] == bd && red_i[k] < bi)) { bd = red_d[k]; bi = red_i[k]; }
        const unsigned long long key =
            bd >= 0.f ? ((unsigned long long)__float_as_uint(bd) << 32) | (unsigned long long)(0xffffffffu - (unsigned)bi) : 1ull;
        __hip_atomic_store(&slots[w], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      unsigned spins = 0;
      bool bad = spin_limit == 0u;                              // 0 = "fail at once" (tests of the repair path)
      unsigned long long k = 1ull;
      while (!bad) {
        k = lane < G ? __hip_atomic_load(&slots[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 1ull;
        if (__builtin_amdgcn_ballot_w64(k == 0ull) == 0ull) break;                     // every slot of the round is filled
        if (++spins > spin_limit || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { bad = true; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      if (lane >= G) k = 0ull;
#pragma unroll
      for (int m = 1; m < 16; m <<= 1) {                        // G <= 16 slots: maximum over lanes 0 .. 15
        const unsigned long long o = __shfl_xor(k, m);
        k = o > k ? o : k;
      }
      if (lane == 0) {
        if (bad || k <= 1ull) {                                 // (k <= 1: no workgroup had a candidate this round: not a valid index either)
          __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          far_s = -1;
        } else {
          far_s = (int)(0xffffffffu - (unsigned)(k & 0xffffffffull));
        }
      }
    }
    __syncthreads();
    far = far_s;
    if (far < 0) return;                                        // uniform: the whole workgroup leaves
  }
}

__global__ __launch_bounds__(256) void fps_unpack_kernel(const unsigned long long* __restrict__ win, const unsigned* __restrict__ cnt,
                                                         int64_t* __restrict__ out, int B, int npoint, int G) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)B * npoint) return;
  const int b = (int)(i / npoint), it = (int)(i % npoint);
  unsigned long long k = win[i * G];                            // round 0: the start index in the first slot; else the maximum of the round's slots
  if (it > 0)
    for (int w = 1; w < G; ++w) k = win[i * G + w] > k ? win[i * G + w] : k;
  int64_t v = it == 0 ? (int64_t)k : (int64_t)(0xffffffffu - (unsigned)(k & 0xffffffffull));
  if (cnt[B + b] != 0u) v = -1;                                 // a cloud whose workgroups could not meet: fps_repair_kernel rewrites it
  out[i] = v;
}

// Repair pass behind fps_coop_kernel: one workgroup per cloud, which leaves at once unless the cloud's error word is set (its
// workgroups did not all become resident within the spin bound -- e.g. persistent kernels of another stream held the CUs).  Then
// it redoes the whole cloud alone: same distances, same tie rule, running min-distances in the workspace (any N).  The caller
// therefore never sees the -1 of a failed cloud, with no host round trip and nothing a hipGraph capture could not record.
__global__ __launch_bounds__(FPS_T) void fps_repair_kernel(const float* __restrict__ xyz4, const int64_t* __restrict__ start,
                                                           unsigned* __restrict__ cnt, float* __restrict__ distws,
                                                           int64_t* __restrict__ out, int B, int N, int npoint) {
  __shared__ float red_d[16];
  __shared__ int red_i[16];
  __shared__ int far_s;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (cnt[B + b] == 0u) return;                                 // uniform per workgroup
  const float* xb = xyz4 + (int64_t)b * N * 4;
  float* db = distws + (int64_t)b * N;
  for (int i = tid; i < N; i += FPS_T) db[i] = 1e10f;           // each thread only ever touches its own entries
  int far = fps_start(start[b], N);
  for (int it = 0; it < npoint; ++it) {
    if (tid == 0) out[(int64_t)b * npoint + it] = far;
    const f32x4 c = *reinterpret_cast<const f32x4*>(xb + (int64_t)far * 4);
    float bd = -1.f;
    int bi = 0x7fffffff;
    for (int i = tid; i < N; i += FPS_T) {
      const f32x4 p = *reinterpret_cast<const f32x4*>(xb + (int64_t)i * 4);
      const float d = fminf(db[i], sqdist3(p[0], p[1], p[2], c[0], c[1], c[2]));
      db[i] = d;
      if (d > bd) { bd = d; bi = i; }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      const float od = __shfl_xor(bd, m);
      const int oi = __shfl_xor(bi, m);
      if (od > bd || (od == bd && oi < bi)) { bd = od; bi = oi; }
    }
    if (lane == 0) { red_d[wave] = bd; red_i[wave] = bi; }
    __syncthreads();
    if (wave == 0) {
      float d2 = lane < 16 ? red_d[lane] : -2.f;
      int i2 = lane < 16 ? red_i[lane] : 0x7fffffff;
#pragma unroll
      for (int m = 8; m >= 1; m >>= 1) {
        const float od = __shfl_xor(d2, m);
        const int oi = __shfl_xor(i2, m);
        if (od > d2 || (od == d2 && oi < i2)) { d2 = od; i2 = oi; }
      }
      if (lane == 0) far_s = i2 < N ? i2 : 0;
    }
    __syncthreads();
    far = far_s;
  }
  if (tid == 0) cnt[B + b] = 2u;                                // 2 = "was repaired" (visible to the caller in the workspace)
}

// ball query: one wave per query; ascending scan, first `nsample` hits with d2 <= r2, padded
// with the first hit (pointnet_util.py:86-92).  A query with no hit yields N (as the reference).
__global__ __launch_bounds__(256) void ball_query_kernel(const float* __restrict__ xyz4, const float* __restrict__ new4,
                                                         int64_t* __restrict__ out, int N, int S, int nsample, float r2) {
  const int b = blockIdx.y, lane = threadIdx.x & 63;
  const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= S) return;
  const f32x4 qp = *reinterpret_cast<const f32x4*>(new4 + ((int64_t)b * S + s) * 4);
  const float* xb = xyz4 + (int64_t)b * N * 4;
  int64_t* ob = out + ((int64_t)b * S + s) * nsample;
  int cnt = 0, first = N;
  for (int i0 = 0; i0 < N && cnt < nsample; i0 += 64) {
    const int i = i0 + lane;
    bool hit = false;
    if (i < N) {
      const f32x4 p = *reinterpret_cast<const f32x4*>(xb + (int64_t)i * 4);
      hit = !(sqdist3(qp[0], qp[1], qp[2], p[0], p[1], p[2]) > r2);
    }
    const unsigned long long mask = __ballot(hit);
    if (mask) {
      if (cnt == 0) first = i0 + __ffsll((long long)mask) - 1;
      const int slot = cnt + __popcll(mask & ((1ull << lane) - 1ull));
      if (hit && slot < nsample) ob[slot] = i;
      cnt += __popcll(mask);
    }
  }
  if (cnt > nsample) cnt = nsample;
  for (int j = cnt + lane; j < nsample; j += 64) ob[j] = first;
}

// square_distance: [B,N,4] x [B,M,4] -> [B,N,M]
__global__ __launch_bounds__(256) void square_distance_kernel(const float* __restrict__ a4, const float* __restrict__ b4,
                                                              float* __restrict__ out, int N, int M) {
  const int b = blockIdx.z;
  const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
  if (j >= M) return;
  const f32x4 p = *reinterpret_cast<const f32x4*>(a4 + ((int64_t)b * N + i) * 4);
  const f32x4 q = *reinterpret_cast<const f32x4*>(b4 + ((int64_t)b * M + j) * 4);
  out[((int64_t)b * N + i) * M + j] = sqdist3(p[0], p[1], p[2], q[0], q[1], q[2]);
}

// ---- per-batch channel reductions over rows: partial [B][nslab][C] then final [B][C] --------
template <bool IS_MAX>
__global__ __launch_bounds__(256) void colreduce_partial_kernel(const float* __restrict__ x, int64_t ldx,
                                                                float* __restrict__ part, int N, int C, int slab_rows) {
  // block (C/4 float4 columns x rows-in-flight): thread -> column group c4, row phase rp
  const int b = blockIdx.y, slab = blockIdx.x;
  const int c4n = C / 4, rows_par = 256 / c4n;
  const int c = (threadIdx.x % c4n) * 4, rp = threadIdx.x / c4n;
  const int lo = slab * slab_rows, hi = min(N, lo + slab_rows);
  f32x4 acc;
  const float init = IS_MAX ? -INFINITY : 0.f;
  acc[0] = acc[1] = acc[2] = acc[3] = init;
  if (rp < rows_par)
    for (int r = lo + rp; r < hi; r += rows_par) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(x + ((int64_t)b * N + r) * ldx + c);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = IS_MAX ? fmaxf(acc[i], v[i]) : acc[i] + v[i];
    }
  __shared__ __attribute__((aligned(16))) float sm[256 * 4];
  *reinterpret_cast<f32x4*>(&sm[threadIdx.x * 4]) = acc;
  __syncthreads();
  if (rp == 0) {
    for (int p = 1; p < rows_par; ++p) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(&sm[(p * c4n + threadIdx.x) * 4]);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = IS_MAX ? fmaxf(acc[i], v[i]) : acc[i] + v[i];
    }
    *reinterpret_cast<f32x4*>(part + ((int64_t)b * gridDim.x + slab) * C + c) = acc;
  }
}

template <bool IS_MAX>
__global__ __launch_bounds__(256) void colreduce_final_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                              int nslab, int C, float mul) {
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += 256) {
    float acc = IS_MAX ? -INFINITY : 0.f;
    for (int s = 0; s < nslab; ++s) {
      const float v = part[((int64_t)b * nslab + s) * C + c];
      acc = IS_MAX ? fmaxf(acc, v) : acc + v;
    }
    out[(int64_t)b * C + c] = IS_MAX ? acc : acc * mul;
  }
}

}  // namespace

#define GRID1D(n) dim3((unsigned)(((n) + 255) / 256))

extern "C" int cmr_planar_to_rows_f32(const float* x, float* y, int B, int C, int N, int Cpad, hipStream_t stream) {
  CMR_REQUIRE(x && y && B > 0 && C >= 1 && (Cpad == 4 || Cpad == 8) && C <= Cpad && N > 0 && cmr_aligned16(y));
  hipLaunchKernelGGL(planar_to_rows_kernel, GRID1D((int64_t)B * N), dim3(256), 0, stream, x, y, B, C, N, Cpad);
  return cmr_launch_status();
}

extern "C" int cmr_concat_rows_f32(const float* x1, int64_t ld1, int C1, const float* x2, int64_t ld2, int C2,
                                   const int32_t* idx2, int64_t div2, float* out, int64_t rows, hipStream_t stream) {
  CMR_REQUIRE(x1 && x2 && out && rows > 0 && C1 > 0 && C2 > 0 && C1 % 4 == 0 && C2 % 4 == 0 && ld1 % 4 == 0 &&
              ld2 % 4 == 0 && cmr_aligned16(x1) && cmr_aligned16(x2) && cmr_aligned16(out));
  hipLaunchKernelGGL(concat_rows_kernel, GRID1D(rows * ((C1 + C2) / 4)), dim3(256), 0, stream, x1, ld1, C1, x2, ld2, C2,
                     idx2, div2, out, rows);
  return cmr_launch_status();
}

extern "C" int cmr_index_to_global_i32(const int64_t* idx, int32_t* out, int B, int N, int M, hipStream_t stream) {
  CMR_REQUIRE(idx && out && B > 0 && N > 0 && M > 0 && (int64_t)B * M < 2147483647LL && (int64_t)B * N < 2147483647LL);
  hipLaunchKernelGGL(index_to_global_kernel, GRID1D((int64_t)B * N), dim3(256), 0, stream, idx, out, B, N, M);
  return cmr_launch_status();
}

// key [B*n_per_batch] global segment ids; count/offsets: [B*seg_per_batch (+1)]; order [B*n_per_batch]
extern "C" int cmr_csr_build_i32(const int32_t* key, int32_t* count, int32_t* offsets, int32_t* order, int B,
                                 int n_per_batch, int seg_per_batch, hipStream_t stream) {
  CMR_REQUIRE(key && count && offsets && order && B > 0 && n_per_batch > 0 && seg_per_batch > 0);
  CMR_REQUIRE((int64_t)B * seg_per_batch < 2147483647LL && (int64_t)B * n_per_batch < 2147483647LL);
  const int total = B * seg_per_batch;
  const int64_t nkeys = (int64_t)B * n_per_batch;
  hipLaunchKernelGGL(csr_zero_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, count, total);
  hipLaunchKernelGGL(csr_count_kernel, GRID1D(nkeys), dim3(256), 0, stream, key, count, n_per_batch, seg_per_batch, nkeys);
  hipLaunchKernelGGL(exclusive_scan_kernel, dim3(1), dim3(1024), 0, stream, (const int32_t*)count, offsets, total);
  hipLaunchKernelGGL(csr_scatter_kernel, GRID1D(nkeys), dim3(256), 0, stream, key, (const int32_t*)offsets, count, order, n_per_batch,
                     seg_per_batch, nkeys);
  hipLaunchKernelGGL(csr_sort_kernel, dim3((total + 3) / 4), dim3(256), 0, stream, key, (const int32_t*)offsets, count, order, n_per_batch,
                     seg_per_batch, total);
  return cmr_launch_status();
}

extern "C" int cmr_knn16_f32(const float* xyz4, int32_t* out, int B, int M, hipStream_t stream) {
  CMR_REQUIRE(xyz4 && out && B > 0 && B <= 65535 && M >= 16 && cmr_aligned16(xyz4));
  const size_t smem = (size_t)M * 4 * sizeof(float);
  CMR_REQUIRE(smem <= 160 * 1024);
  static CmrSmemCache granted{};
  if (cmr_grant_smem(reinterpret_cast<const void*>(knn_kernel<16>), smem, granted) != CMR_OK) return CMR_ELAUNCH;
  constexpr int QPW = 256 / KNN_SPLIT;          // queries per workgroup
  hipLaunchKernelGGL(knn_kernel<16>, dim3((M + QPW - 1) / QPW, B), dim3(256), smem, stream, xyz4, out, M);
  return cmr_launch_status();
}

extern "C" int cmr_knn_f32(const float* q4, const float* c4, int64_t* out, int B, int S, int N, int K, hipStream_t stream) {
  CMR_REQUIRE(q4 && c4 && out && B > 0 && B <= 65535 && S > 0 && N > 0 && K >= 1 && K <= 64 && cmr_aligned16(q4) && cmr_aligned16(c4));
  constexpr int QPW = 256 / KNN_SPLIT;
  const dim3 grid((S + QPW - 1) / QPW, B), block(256);
  if (K <= 8) hipLaunchKernelGGL(knn_general_kernel<8>, grid, block, 0, stream, q4, c4, out, S, N, K);
  else if (K <= 16) hipLaunchKernelGGL(knn_general_kernel<16>, grid, block, 0, stream, q4, c4, out, S, N, K);
  else if (K <= 32) hipLaunchKernelGGL(knn_general_kernel<32>, grid, block, 0, stream, q4, c4, out, S, N, K);
  else hipLaunchKernelGGL(knn_general_kernel<64>, grid, block, 0, stream, q4, c4, out, S, N, K);
  return cmr_launch_status();
}

extern "C" int cmr_nearest_f32(const float* q4, const float* c4, int32_t* out_global, int64_t* out_local, int B, int Nq,
                               int Nc, hipStream_t stream) {
  CMR_REQUIRE(q4 && c4 && (out_global || out_local) && B > 0 && B <= 65535 && Nq > 0 && Nc > 0);
  hipLaunchKernelGGL(nearest_kernel, dim3((Nq + 255) / 256, B), dim3(256), 0, stream, q4, c4, out_global, out_local, Nq,
                     Nc);
  return cmr_launch_status();
}

extern "C" int cmr_three_nn_f32(const float* q4, const float* c4, int32_t* idx, float* wgt, int B, int Nq, int Nc,
                                hipStream_t stream) {
  CMR_REQUIRE(q4 && c4 && idx && wgt && B > 0 && B <= 65535 && Nq > 0 && Nc >= 3);
  hipLaunchKernelGGL(three_nn_kernel, dim3((Nq + 255) / 256, B), dim3(256), 0, stream, q4, c4, idx, wgt, Nq, Nc);
  return cmr_launch_status();
}

extern "C" int cmr_weighted_gather3_f32(const float* src, int64_t lds, const int32_t* idx, const float* wgt, float* out,
                                        int64_t ldo, int64_t rows, int C, hipStream_t stream) {
  CMR_REQUIRE(src && idx && wgt && out && rows > 0 && C > 0);
  hipLaunchKernelGGL(weighted_gather3_kernel, GRID1D(rows * C), dim3(256), 0, stream, src, lds, idx, wgt, out, ldo, rows, C);
  return cmr_launch_status();
}

extern "C" int cmr_weighted_scatter3_f32(const float* dy, int64_t ldd, const float* wgt, const int32_t* order, const int32_t* offsets, float* out,
                                         int64_t ldo, int64_t nseg, int C, hipStream_t stream) {
  CMR_REQUIRE(dy && wgt && order && offsets && out && nseg > 0 && C > 0);
  hipLaunchKernelGGL(weighted_scatter3_kernel, GRID1D(nseg * C), dim3(256), 0, stream, dy, ldd, wgt, order, offsets, out, ldo, nseg, C);
  return cmr_launch_status();
}

extern "C" int cmr_rel_pos_f32(const float* a, const int32_t* ia, int64_t diva, const float* b, const int32_t* ib,
                               int64_t divb, float* out, int64_t rows, hipStream_t stream) {
  CMR_REQUIRE(a && b && out && rows > 0 && cmr_aligned16(a) && cmr_aligned16(b) && cmr_aligned16(out));
  hipLaunchKernelGGL(rel_pos_kernel, GRID1D(rows), dim3(256), 0, stream, a, ia, diva, b, ib, divb, out, rows);
  return cmr_launch_status();
}

extern "C" int cmr_vecattn_prep_f32(const float* q, int64_t ldq, const int32_t* iq, int64_t divq, const float* k,
                                    int64_t ldk, const float* v, int64_t ldv, const int32_t* ik, const float* pos,
                                    float* t, float* vp, int64_t rows, hipStream_t stream) {
  CMR_REQUIRE(q && k && v && pos && t && vp && rows > 0 && ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0);
  hipLaunchKernelGGL(vecattn_prep_kernel, GRID1D(rows * 16), dim3(256), 0, stream, q, ldq, iq, divq, k, ldk, v, ldv, ik,
                     pos, t, vp, rows);
  return cmr_launch_status();
}

extern "C" int cmr_segment_softmax_f32(const float* attn, const float* vp, const int32_t* order, const int32_t* offsets,
                                       int fixed_len, float scale, float* out, int64_t nseg, hipStream_t stream) {
  CMR_REQUIRE(attn && vp && out && nseg > 0 && (offsets || fixed_len > 0));
  hipLaunchKernelGGL(segment_softmax_kernel, dim3((unsigned)((nseg + 3) / 4)), dim3(256), 0, stream, attn, vp, order,
                     offsets, fixed_len, scale, out, nseg);
  return cmr_launch_status();
}

extern "C" int cmr_segment_reduce_f32(const float* src, int64_t lds, const int32_t* order, const int32_t* offsets,
                                      float* out, int64_t ldo, int64_t nseg, int C, int mode, hipStream_t stream) {
  CMR_REQUIRE(src && order && offsets && out && nseg > 0 && C > 0 && mode >= 0 && mode <= 2);
  hipLaunchKernelGGL(segment_reduce_kernel, dim3((unsigned)((nseg + 3) / 4)), dim3(256), 0, stream, src, lds, order, offsets,
                     out, ldo, nseg, C, mode);
  return cmr_launch_status();
}

extern "C" int cmr_gather_rows_f32(const float* src, int64_t lds, const int32_t* idx, float* out, int64_t ldo,
                                   int64_t rows, int C, hipStream_t stream) {
  CMR_REQUIRE(src && idx && out && rows > 0 && C > 0);
  if (C % 4 == 0 && lds % 4 == 0 && ldo % 4 == 0 && cmr_aligned16(src) && cmr_aligned16(out)) {      // 16 bytes per thread
    hipLaunchKernelGGL(gather_rows4_kernel, GRID1D(rows * (C / 4)), dim3(256), 0, stream, src, lds, idx, out, ldo, rows, C / 4);
    return cmr_launch_status();
  }
  hipLaunchKernelGGL(gather_rows_kernel, GRID1D(rows * C), dim3(256), 0, stream, src, lds, idx, out, ldo, rows, C);
  return cmr_launch_status();
}

extern "C" int cmr_fps_f32(const float* xyz4, const int64_t* start, int64_t* out, int B, int N, int npoint,
                           hipStream_t stream) {
  CMR_REQUIRE(xyz4 && start && out && B > 0 && N > 0 && npoint > 0 && cmr_aligned16(xyz4));
  const int per = (N + FPS_T - 1) / FPS_T;
  CMR_REQUIRE(per <= 64);                                       // N <= 65536
#define FPS_CASE(P, C) \
  hipLaunchKernelGGL((fps_kernel<P, C>), dim3(B), dim3(FPS_T), 0, stream, xyz4, start, out, N, npoint)
  if (per <= 1) FPS_CASE(1, true);
  else if (per <= 4) FPS_CASE(4, true);
  else if (per <= 16) FPS_CASE(16, true);
  else if (per <= 40) FPS_CASE(40, false);
  else FPS_CASE(64, false);
#undef FPS_CASE
  return cmr_launch_status();
}


// Cooperative variant for N > 16 384 (and any N the caller prefers): G workgroups per cloud.  ws: [B][npoint][G] 64-bit round slots
// (room for 16 per round), then [2 B] 32-bit status words, then [B][N] floats for the repair pass.  A cloud whose workgroups did not all become
// resident within the spin bound is recomputed by fps_repair_kernel in the same call: out never holds the -1 of a failed cloud.
static int fps_groups(int B, int N) {
  int g = 256 / B;                                              // all B * G workgroups resident at once, one per CU
  const int cap = 16;                                           // measured at 65 536 points: 16 groups 2.5 us / round at B = 1, 2.3 at B = 8 (8 groups: 2.7)
  if (g > cap) g = cap;
  const int need = (N + 64 * FPS_CT - 1) / (64 * FPS_CT);       // <= 64 points per thread
  if (g < need) g = need;
  return g < 1 ? 1 : g;
}

// workspace: [B][npoint][16] 64-bit round slots | [B] (unused) | [B] status words (0 ok, 2 repaired) | pad | [B][N] floats (repair)
constexpr int FPS_GMAX = 16;                                   // slots per round in the workspace (G <= 16)
static int64_t fps_words_bytes(int B, int npoint) { return ((int64_t)B * npoint * FPS_GMAX * 8 + (int64_t)2 * B * 4 + 15) / 16 * 16; }

extern "C" int64_t cmr_fps_workspace_bytes(int B, int N, int npoint) {
  return fps_words_bytes(B, npoint) + (int64_t)B * N * 4;
}

static int fps_ws_launch(const float* xyz4, const int64_t* start, int64_t* out, int B, int N, int npoint, void* ws, int64_t ws_bytes,
                         unsigned spin_limit, hipStream_t stream) {
  CMR_REQUIRE(xyz4 && start && out && ws && B > 0 && N > 0 && npoint > 0 && cmr_aligned16(xyz4) && (reinterpret_cast<uintptr_t>(ws) & 7u) == 0);
  CMR_REQUIRE(ws_bytes >= cmr_fps_workspace_bytes(B, N, npoint));
  const int G = fps_groups(B, N);
  CMR_REQUIRE((int64_t)B * G <= 256 && G <= FPS_GMAX);           // co-residency of every cloud's workgroups; slots per round
  const int per = ((N + G - 1) / G + FPS_CT - 1) / FPS_CT;
  CMR_REQUIRE(per <= 64);
  if (hipMemsetAsync(ws, 0, (size_t)fps_words_bytes(B, npoint), stream) != hipSuccess) return CMR_ELAUNCH;
  unsigned long long* win = (unsigned long long*)ws;             // [B][npoint][G] (the workspace holds FPS_GMAX slots per round)
  unsigned* cnt = (unsigned*)(win + (int64_t)B * npoint * FPS_GMAX);
  float* distws = (float*)((char*)ws + fps_words_bytes(B, npoint));
#define FPS_COOP(P) \
  hipLaunchKernelGGL((fps_coop_kernel<P>), dim3(B * G), dim3(FPS_CT), 0, stream, xyz4, start, win, cnt, B, G, N, npoint, spin_limit)
  if (per <= 4) FPS_COOP(4);
  else if (per <= 8) FPS_COOP(8);
  else if (per <= 16) FPS_COOP(16);
  else if (per <= 32) FPS_COOP(32);
  else FPS_COOP(64);
#undef FPS_COOP
  hipLaunchKernelGGL(fps_unpack_kernel, dim3((unsigned)(((int64_t)B * npoint + 255) / 256)), dim3(256), 0, stream, (const unsigned long long*)win,
                     (const unsigned*)cnt, out, B, npoint, G);
  // clouds whose workgroups did not meet are redone by ONE workgroup each (a no-op launch of B workgroups otherwise)
  hipLaunchKernelGGL(fps_repair_kernel, dim3(B), dim3(FPS_T), 0, stream, xyz4, start, cnt, distws, out, B, N, npoint);
  return cmr_launch_status();
}

extern "C" int cmr_fps_ws_f32(const float* xyz4, const int64_t* start, int64_t* out, int B, int N, int npoint, void* ws, int64_t ws_bytes,
                              hipStream_t stream) {
  return fps_ws_launch(xyz4, start, out, B, N, npoint, ws, ws_bytes, FPS_SPIN, stream);
}

extern "C" int cmr_fps_ws_spin_f32(const float* xyz4, const int64_t* start, int64_t* out, int B, int N, int npoint, void* ws,
                                   int64_t ws_bytes, int spin_limit, hipStream_t stream) {
  CMR_REQUIRE(spin_limit >= 0);
  return fps_ws_launch(xyz4, start, out, B, N, npoint, ws, ws_bytes, (unsigned)spin_limit, stream);
}

extern "C" int cmr_ball_query_f32(const float* xyz4, const float* new4, int64_t* out, int B, int N, int S, int nsample,
                                  float radius2, hipStream_t stream) {
  CMR_REQUIRE(xyz4 && new4 && out && B > 0 && B <= 65535 && N > 0 && S > 0 && nsample > 0);
  hipLaunchKernelGGL(ball_query_kernel, dim3((S + 3) / 4, B), dim3(256), 0, stream, xyz4, new4, out, N, S, nsample,
                     radius2);
  return cmr_launch_status();
}

extern "C" int cmr_square_distance_f32(const float* a4, const float* b4, float* out, int B, int N, int M,
                                       hipStream_t stream) {
  CMR_REQUIRE(a4 && b4 && out && B > 0 && B <= 65535 && N > 0 && N <= 65535 && M > 0);
  hipLaunchKernelGGL(square_distance_kernel, dim3((M + 255) / 256, N, B), dim3(256), 0, stream, a4, b4, out, N, M);
  return cmr_launch_status();
}

static int colreduce(bool is_max, const float* x, int64_t ldx, float* out, float* ws, int64_t ws_bytes, int B, int N,
                     int C, float mul, hipStream_t stream) {
  CMR_REQUIRE(x && out && ws && B > 0 && B <= 65535 && N > 0 && C % 4 == 0 && C >= 4 && C <= 1024 && ldx % 4 == 0);
  const int slab_rows = 256;
  const int nslab = (N + slab_rows - 1) / slab_rows;
  CMR_REQUIRE(ws_bytes >= (int64_t)B * nslab * C * (int64_t)sizeof(float));
  if (is_max) {
    hipLaunchKernelGGL(colreduce_partial_kernel<true>, dim3(nslab, B), dim3(256), 0, stream, x, ldx, ws, N, C, slab_rows);
    hipLaunchKernelGGL(colreduce_final_kernel<true>, dim3(B), dim3(256), 0, stream, ws, out, nslab, C, mul);
  } else {
    hipLaunchKernelGGL(colreduce_partial_kernel<false>, dim3(nslab, B), dim3(256), 0, stream, x, ldx, ws, N, C, slab_rows);
    hipLaunchKernelGGL(colreduce_final_kernel<false>, dim3(B), dim3(256), 0, stream, ws, out, nslab, C, mul);
  }
  return cmr_launch_status();
}

extern "C" int64_t cmr_colreduce_workspace_bytes(int B, int N, int C) {
  return (int64_t)B * ((N + 255) / 256) * C * (int64_t)sizeof(float);
}

extern "C" int cmr_colmax_f32(const float* x, int64_t ldx, float* out, void* ws, int64_t ws_bytes, int B, int N, int C,
                              hipStream_t stream) {
  return colreduce(true, x, ldx, out, (float*)ws, ws_bytes, B, N, C, 1.f, stream);
}

extern "C" int cmr_colmean_f32(const float* x, int64_t ldx, float* out, void* ws, int64_t ws_bytes, int B, int N, int C,
                               hipStream_t stream) {
  return colreduce(false, x, ldx, out, (float*)ws, ws_bytes, B, N, C, 1.f / (float)N, stream);
}
