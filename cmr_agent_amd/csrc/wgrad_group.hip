// Weight / bias gradients of SEVERAL small row-map linears and the parameter gradients of their LayerNorms in ONE call of two kernels
// (reference: loss.backward() through the nn.Linear / nn.LayerNorm modules of a transformer block or a linear-attention layer,
// Train_Geo.py:166-174).  cmr_linear_wgrad_f32 costs two launches per Linear; a transformer block has six Linears and two LayerNorms.
//
//   kernel 1  one workgroup per (problem, row slice, 64-wide n block, 64-wide k block): the loop of linear_wgrad_kernel<2, 2>
//             (wgrad.hip: dW[n][k] = sum_r dY[r][n] X[r][k] on v_mfma_f32_32x32x2_f32, operands by dword loads with lane = channel, two
//             register sets), partial tiles + partial column sums of dY into the workspace;
//   kernel 2  every output element of every problem = the sum of its slices (double, fixed order), written or accumulated in place in the
//             gradient bucket; and the "vector jobs": sums over the row tiles of the per-tile LayerNorm parameter gradients the row
//             kernels (vit_train.hip) left behind.
// Deterministic: no atomics.  Problems / jobs are passed BY VALUE (host descriptor array -> kernel arguments): nothing to upload.
#include "cmr_common.h"

namespace {

constexpr int WG_MAXP = 8;       // linear problems per call
constexpr int WG_MAXV = 4;       // vector jobs per call
constexpr int WG_T = 64;         // tile edge (NT = KT = 2 tiles of 32)

struct WgProblem {
  const float* dy; const float* x; float* dw; float* db;
  int64_t lddy, ldx, lddw, part_off, partb_off;      // offsets (floats) of this problem's partials in the workspace
  int rows, n, k, slices, nblk, kblk, acc, acc_db;
  int item0;                                         // first workgroup of kernel 1 / first output block of kernel 2
  int red0;
};
struct WgVector {
  const float* part; float* out_a; float* out_b;     // part [nparts][2 len]: sums of (a | b) per tile
  int nparts, len, acc, red0;
};
struct WgArgs {
  WgProblem p[WG_MAXP];
  WgVector v[WG_MAXV];
  int nprob, nvec, items, red_blocks;
  float* ws;
};

__global__ __launch_bounds__(256) void wgrad_group_kernel(const WgArgs a) {
  constexpr int KT = 2, NT = 2, NSPLIT = 2, UNR = 4;
  int pi = 0;
#pragma unroll
  for (int i = 1; i < WG_MAXP; ++i)
    if (i < a.nprob && (int)blockIdx.x >= a.p[i].item0) pi = i;
  const WgProblem& P = a.p[pi];
  int it = blockIdx.x - P.item0;
  const int sl_blk = it % P.slices;
  it /= P.slices;
  const int nb = it % P.nblk, kb = it / P.nblk;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int n_t = wave % NT, split = wave / NT;
  const int n0 = nb * WG_T, k0 = kb * WG_T;
  const int64_t rows = P.rows;
  const int64_t nsteps = (rows + 2 * UNR - 1) / (2 * UNR);
  const int64_t per_blk = (nsteps + P.slices - 1) / P.slices;
  const int64_t s0 = min((int64_t)sl_blk * per_blk, nsteps), s1 = min(s0 + per_blk, nsteps);
  const int64_t per_w = (s1 - s0 + NSPLIT - 1) / NSPLIT;
  const int64_t w0 = min(s0 + split * per_w, s1), w1 = min(w0 + per_w, s1);

  f32x16 acc[KT];
#pragma unroll
  for (int t = 0; t < KT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;
  const int ncol = n0 + n_t * 32 + l31;
  const bool n_ok = ncol < P.n;
  const float* dcol = P.dy + (n_ok ? ncol : 0);
  bool k_ok[KT];
  const float* xcol[KT];
#pragma unroll
  for (int t = 0; t < KT; ++t) {
    k_ok[t] = k0 + t * 32 + l31 < P.k;
    xcol[t] = P.x + (k_ok[t] ? k0 + t * 32 + l31 : 0);
  }
  const int64_t lddy = P.lddy, ldx = P.ldx;
  float sa[2][UNR], sv[2][UNR][KT];
  auto load = [&](float (&av)[UNR], float (&v)[UNR][KT], int64_t step) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int64_t r = (step * UNR + u) * 2 + h;
      const int64_t rc = r < rows ? r : rows - 1;
      av[u] = dcol[rc * lddy];
#pragma unroll
      for (int t = 0; t < KT; ++t) v[u][t] = xcol[t][rc * ldx];
    }
  };
  auto mask = [&](float (&av)[UNR], float (&v)[UNR][KT], int64_t step, bool live) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const bool ok = live && (step * UNR + u) * 2 + h < rows;
      av[u] = ok && n_ok ? av[u] : 0.f;
#pragma unroll
      for (int t = 0; t < KT; ++t) v[u][t] = k_ok[t] ? v[u][t] : 0.f;
    }
  };
  const int64_t last = nsteps - 1;
  if (w0 < w1) load(sa[0], sv[0], w0);
#define CMR_WG_STEP(C, L, S)                                                                 \
  {                                                                                          \
    load(sa[L], sv[L], (S) + 1 < nsteps ? (S) + 1 : last);                                   \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    mask(sa[C], sv[C], (S), (S) < w1);                                                       \
    _Pragma("unroll") for (int u = 0; u < UNR; ++u) {                                        \
      bsum += sa[C][u];                                                                      \
      _Pragma("unroll") for (int t = 0; t < KT; ++t) acc[t] = cmr_mfma32(sa[C][u], sv[C][u][t], acc[t]); \
    }                                                                                        \
  }
  for (int64_t s = w0; s < w1; s += 2) {
    CMR_WG_STEP(0, 1, s)
    CMR_WG_STEP(1, 0, s + 1)
  }
#undef CMR_WG_STEP
  const int64_t nsl = (int64_t)P.slices * NSPLIT, sl = (int64_t)sl_blk * NSPLIT + split;
  float* out = a.ws + P.part_off + (((int64_t)kb * P.nblk + nb) * nsl + sl) * WG_T * WG_T;
#pragma unroll
  for (int t = 0; t < KT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = cmr_mfma_row(r, lane);
      out[(int64_t)(n_t * 32 + row) * WG_T + t * 32 + l31] = acc[t][r];
    }
  if (P.db && kb == 0) {
    bsum += cmr_xhalf(bsum);                                            // the two row parities
    if (h == 0) a.ws[P.partb_off + ((int64_t)nb * nsl + sl) * WG_T + n_t * 32 + l31] = bsum;
  }
}

constexpr int WR_OUT = 64, WR_GRP = 4, WR_U = 8;     // outputs per block, slice groups per output, loads in flight per thread

__global__ __launch_bounds__(WR_OUT * WR_GRP) void wgrad_group_reduce_kernel(const WgArgs a) {
  __shared__ double sm[WR_GRP][WR_OUT];
  const int o = threadIdx.x % WR_OUT, grp = threadIdx.x / WR_OUT;
  const int b = blockIdx.x;
  // ---- which job does this block belong to?
  int vi = -1, pi = 0;
#pragma unroll
  for (int i = 0; i < WG_MAXV; ++i)
    if (i < a.nvec && b >= a.v[i].red0) vi = i;
  if (vi < 0) {
#pragma unroll
    for (int i = 1; i < WG_MAXP; ++i)
      if (i < a.nprob && b >= a.p[i].red0) pi = i;
  }
  double s = 0.0;
  const float* p = a.ws;
  int64_t stride = 0, total = 0, i = 0;
  int nsl = 0;
  if (vi >= 0) {
    const WgVector& V = a.v[vi];
    i = (int64_t)(b - V.red0) * WR_OUT + o;
    total = 2 * V.len;
    p = V.part + i;
    stride = 2 * V.len;
    nsl = V.nparts;
  } else {
    const WgProblem& P = a.p[pi];
    i = (int64_t)(b - P.red0) * WR_OUT + o;
    const int64_t nk = (int64_t)P.n * P.k;
    total = nk + (P.db ? P.n : 0);
    nsl = P.slices * 2;
    if (i < nk) {
      const int row = (int)(i / P.k), col = (int)(i - (int64_t)row * P.k);
      const int nb = row / WG_T, kb = col / WG_T;
      p = a.ws + P.part_off + ((int64_t)kb * P.nblk + nb) * nsl * WG_T * WG_T + (int64_t)(row - nb * WG_T) * WG_T + (col - kb * WG_T);
      stride = WG_T * WG_T;
    } else if (i < total) {
      const int row = (int)(i - nk);
      const int nb = row / WG_T;
      p = a.ws + P.partb_off + (int64_t)nb * nsl * WG_T + (row - nb * WG_T);
      stride = WG_T;
    }
  }
  if (i < total)
    for (int j0 = grp; j0 < nsl; j0 += WR_GRP * WR_U) {       // WR_U loads in flight per thread (one dependent load per iteration left the
      float v[WR_U];                                             // reduction a chain of memory round trips: 28 us for a transformer block)
#pragma unroll
      for (int u = 0; u < WR_U; ++u) {
        const int j = j0 + u * WR_GRP;
        v[u] = p[(int64_t)(j < nsl ? j : j0) * stride];          // branch-free: a slice past the end re-reads j0 and is dropped below
      }
#pragma unroll
      for (int u = 0; u < WR_U; ++u) s += j0 + u * WR_GRP < nsl ? (double)v[u] : 0.0;
    }
  sm[grp][o] = s;
  __syncthreads();
  if (grp != 0 || i >= total) return;
#pragma unroll
  for (int j = 1; j < WR_GRP; ++j) s += sm[j][o];
  if (vi >= 0) {
    const WgVector& V = a.v[vi];
    float* d = i < V.len ? V.out_a + i : V.out_b + (i - V.len);
    *d = V.acc ? *d + (float)s : (float)s;
  } else {
    const WgProblem& P = a.p[pi];
    const int64_t nk = (int64_t)P.n * P.k;
    if (i < nk) {
      const int row = (int)(i / P.k), col = (int)(i - (int64_t)row * P.k);
      float* d = P.dw + (int64_t)row * P.lddw + col;
      *d = P.acc ? *d + (float)s : (float)s;
    } else {
      float* d = P.db + (i - nk);
      *d = P.acc_db ? *d + (float)s : (float)s;
    }
  }
}

inline int wg_slices(int64_t rows) {
  // ~256 rows per workgroup (two sub-slices of 128) on the token maps (<= 4 096 rows: 16 workgroup slices), growing to ~1 280 rows on the
  // 40 960-row pixel maps: every slice is a 64 x 64 partial tile per (n, k) block that the reduction reads back
  int64_t s = (rows + 255) / 256;
  if (s > 16) s = 16 + (s - 16) / 8;
  return (int)(s < 1 ? 1 : (s > 64 ? 64 : s));
}

// desc: nprob x 12 int64 {dy, lddy, n, x, ldx, k, rows, dw, lddw, accumulate, db, accumulate_db}, then nvec x 6 int64
// {part, nparts, len, out_a, out_b, accumulate}
int wg_plan(const int64_t* desc, int nprob, int nvec, WgArgs& a, int64_t& ws_floats) {
  if (nprob < 0 || nprob > WG_MAXP || nvec < 0 || nvec > WG_MAXV || nprob + nvec == 0 || !desc) return CMR_EINVAL;
  a.nprob = nprob;
  a.nvec = nvec;
  int items = 0, red = 0;
  int64_t off = 0;
  for (int i = 0; i < nprob; ++i) {
    const int64_t* d = desc + 12 * i;
    WgProblem& P = a.p[i];
    P.dy = reinterpret_cast<const float*>(d[0]); P.lddy = d[1]; P.n = (int)d[2];
    P.x = reinterpret_cast<const float*>(d[3]); P.ldx = d[4]; P.k = (int)d[5]; P.rows = (int)d[6];
    P.dw = reinterpret_cast<float*>(d[7]); P.lddw = d[8]; P.acc = (int)d[9];
    P.db = reinterpret_cast<float*>(d[10]); P.acc_db = (int)d[11];
    if (!P.dy || !P.x || !P.dw || P.n <= 0 || P.k <= 0 || P.rows <= 0 || d[6] > 0x7fffffff) return CMR_EINVAL;
    P.slices = wg_slices(P.rows);
    P.nblk = (P.n + WG_T - 1) / WG_T;
    P.kblk = (P.k + WG_T - 1) / WG_T;
    P.item0 = items;
    items += P.slices * P.nblk * P.kblk;
    P.part_off = off;
    off += (int64_t)P.nblk * P.kblk * P.slices * 2 * WG_T * WG_T;
    P.partb_off = off;
    off += (int64_t)P.nblk * P.slices * 2 * WG_T;
    P.red0 = red;
    red += (int)(((int64_t)P.n * P.k + (P.db ? P.n : 0) + WR_OUT - 1) / WR_OUT);
  }
  for (int i = 0; i < nvec; ++i) {
    const int64_t* d = desc + 12 * nprob + 6 * i;
    WgVector& V = a.v[i];
    V.part = reinterpret_cast<const float*>(d[0]); V.nparts = (int)d[1]; V.len = (int)d[2];
    V.out_a = reinterpret_cast<float*>(d[3]); V.out_b = reinterpret_cast<float*>(d[4]); V.acc = (int)d[5];
    if (!V.part || !V.out_a || !V.out_b || V.nparts <= 0 || V.len <= 0) return CMR_EINVAL;
    V.red0 = red;
    red += (2 * V.len + WR_OUT - 1) / WR_OUT;
  }
  // vector jobs come last in the block numbering: a block b belongs to vector job i iff b >= v[i].red0
  a.items = items;
  a.red_blocks = red;
  ws_floats = off;
  return CMR_OK;
}

}  // namespace

extern "C" int64_t cmr_wgrad_group_workspace_bytes(const int64_t* desc, int nprob, int nvec) {
  WgArgs a{};
  int64_t f = 0;
  if (wg_plan(desc, nprob, nvec, a, f) != CMR_OK) return -1;
  return (f + 4) * (int64_t)sizeof(float);
}

extern "C" int cmr_wgrad_group_f32(const int64_t* desc, int nprob, int nvec, void* ws, int64_t ws_bytes, hipStream_t stream) {
  WgArgs a{};
  int64_t f = 0;
  CMR_REQUIRE(wg_plan(desc, nprob, nvec, a, f) == CMR_OK);
  CMR_REQUIRE((ws || f == 0) && ws_bytes >= f * (int64_t)sizeof(float));
  a.ws = static_cast<float*>(ws);
  if (a.items > 0) hipLaunchKernelGGL(wgrad_group_kernel, dim3((unsigned)a.items), dim3(256), 0, stream, a);
  hipLaunchKernelGGL(wgrad_group_reduce_kernel, dim3((unsigned)a.red_blocks), dim3(WR_OUT * WR_GRP), 0, stream, a);
  return cmr_launch_status();
}
