// Row-streaming fp32 GEMM with fused epilogue, and LayerNorm over 64 channels.
//
// cmr_linear_f32:  Y[r, :] = act( [X1[r, :k1] | X2[map(r), :k2]] . W^T + bias + RES[r, :] )
//   * replaces every nn.Linear / Conv1d(k=1) / Conv2d(k=1) of the path (PointNN.py:96-282,
//     ImageViT.py:61-133, LinearAttention.py:17-31, MultiHeadModel.py:34-47,126-139, CMRAgent.py:57-86);
//     the optional second source is the torch.cat([...], dim=1) + torch.gather idiom of
//     PointViT.py:66-67, IMGPCEnDecoder.py:77-78, MultiHeadModel.py:61-62 done without materialising it.
//   * W is the PyTorch layout [n_out][k1+k2] (row stride ldw): both MFMA operands are "k-contiguous
//     rows", staged in LDS with a 4-float pad so ds_read_b128 is bank-conflict free.
//   * math: v_mfma_f32_32x32x2_f32 (exact fp32).  A k-group of 8 is covered by one b128 read per
//     operand: lane half h holds k = 4h..4h+3, MFMA j consumes k = {j, 4+j}.
#include "cmr_common.h"

namespace {

constexpr int KC = 32;          // K chunk held in LDS
constexpr int LDS_LD = KC + 4;  // padded LDS row, floats

struct LinearArgs {
  const float* x1; int64_t ld1; int k1;
  const float* x2; int64_t ld2; int k2; const int32_t* idx2; int64_t div2;
  const float* w; int64_t ldw;
  const float* bias;
  const float* res; int64_t ldres; int64_t res_mod;
  float* y; int64_t ldy;
  int64_t rows; int n_out; int act; float act_param;
};

template <int MT>
__global__ __launch_bounds__(256) void linear_kernel(const LinearArgs a) {
  constexpr int BM = 128 * MT;       // rows per workgroup (4 waves x MT x 32)
  constexpr int AL = BM / 32;        // float4 A loads per thread per chunk
  __shared__ __attribute__((aligned(16))) float As[BM * LDS_LD];
  __shared__ __attribute__((aligned(16))) float Bs[64 * LDS_LD];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int c4 = tid & 7, r0 = tid >> 3;
  const int64_t row_base = (int64_t)blockIdx.x * BM;
  const int col_base = blockIdx.y * 64;
  const int nch1 = (a.k1 + KC - 1) / KC;
  const int nch2 = a.x2 ? (a.k2 + KC - 1) / KC : 0;
  const int nch = nch1 + nch2;
  const int ntiles = (a.n_out - col_base > 32) ? 2 : 1;

  // source-2 row of each A row this thread stages (fixed over the K loop)
  int64_t srow2[AL];
  if (a.x2) {
#pragma unroll
    for (int i = 0; i < AL; ++i) {
      int64_t r = row_base + r0 + 32 * i;
      srow2[i] = (r < a.rows) ? (a.idx2 ? (int64_t)a.idx2[r] : r / a.div2) : 0;
    }
  }

  f32x4 ra[AL], rb[2];
  auto load_chunk = [&](int c) {   // branch-free (clamped addresses, masked afterwards): all loads in flight together
    const bool second = c >= nch1;
    const int kofs = (second ? c - nch1 : c) * KC;
    const int kvalid = (second ? a.k2 : a.k1) - kofs;
    const bool kin = c4 * 4 < kvalid;
    const float* xs = second ? a.x2 : a.x1;
    const int64_t ld = second ? a.ld2 : a.ld1;
    const int kcol = kin ? kofs + c4 * 4 : 0;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < AL; ++i) {
      const int64_t r = row_base + r0 + 32 * i;
      const bool ok = kin && r < a.rows;
      const int64_t s = ok ? (second ? srow2[i] : r) : 0;
      const f32x4 v = *reinterpret_cast<const f32x4*>(xs + s * ld + kcol);
      ra[i] = ok ? v : zero;
    }
    const int kw = kin ? (second ? a.k1 : 0) + kofs + c4 * 4 : 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int n = col_base + r0 + 32 * i;
      const bool ok = kin && n < a.n_out;
      const f32x4 v = *reinterpret_cast<const f32x4*>(a.w + (int64_t)(ok ? n : 0) * a.ldw + kw);
      rb[i] = ok ? v : zero;
    }
  };

  f32x16 acc[MT][2];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

  load_chunk(0);
  for (int c = 0; c < nch; ++c) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < AL; ++i) *reinterpret_cast<f32x4*>(&As[(r0 + 32 * i) * LDS_LD + c4 * 4]) = ra[i];
#pragma unroll
    for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(&Bs[(r0 + 32 * i) * LDS_LD + c4 * 4]) = rb[i];
    __syncthreads();
    if (c + 1 < nch) load_chunk(c + 1);
#pragma unroll
    for (int kg = 0; kg < KC / 8; ++kg) {
      f32x4 av[MT], bv[2];
#pragma unroll
      for (int m = 0; m < MT; ++m)
        av[m] = *reinterpret_cast<const f32x4*>(&As[(wave * 32 * MT + m * 32 + l31) * LDS_LD + kg * 8 + 4 * h]);
      bv[0] = *reinterpret_cast<const f32x4*>(&Bs[l31 * LDS_LD + kg * 8 + 4 * h]);
      bv[1] = *reinterpret_cast<const f32x4*>(&Bs[(32 + l31) * LDS_LD + kg * 8 + 4 * h]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          acc[m][0] = cmr_mfma32(av[m][j], bv[0][j], acc[m][0]);
          if (ntiles == 2) acc[m][1] = cmr_mfma32(av[m][j], bv[1][j], acc[m][1]);
        }
      }
    }
  }

  // epilogue: lane holds column col for 16 rows of each 32x32 tile
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int col = col_base + n * 32 + l31;
    if (n >= ntiles || col >= a.n_out) continue;
    const float bsv = a.bias ? a.bias[col] : 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = row_base + wave * 32 * MT + m * 32 + cmr_mfma_row(r, lane);
        if (row < a.rows) {
          float v = acc[m][n][r] + bsv;
          if (a.res) {
            const int64_t rr = a.res_mod > 0 ? row % a.res_mod : row;
            v += a.res[rr * a.ldres + col];
          }
          a.y[row * a.ldy + col] = cmr_act(v, a.act, a.act_param);
        }
      }
    }
  }
}

// Skinny GEMM for <= 16 rows (per-sample vectors: 8 rows x K = 128..256).  A wave owns 4
// output channels: its 64 lanes read a weight row as one coalesced 1-KiB float4 load per 256 k,
// multiply with the input rows held in LDS and finish with a wave reduction.  Replaces 128-row
// MFMA tiles that would be > 90 % padding; latency-bound by construction (a few microseconds).
constexpr int SK_ROWS = 16;
template <int R>
__global__ __launch_bounds__(256) void linear_skinny_kernel(const LinearArgs a) {
  extern __shared__ __attribute__((aligned(16))) float xs[];   // [R][ktot], rows past a.rows are zero
  const int ktot = a.k1 + a.k2;
  const int rows = (int)a.rows;
  for (int e = threadIdx.x; e < R * (ktot / 4); e += 256) {
    const int r = e / (ktot / 4), c = (e % (ktot / 4)) * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (r < rows) {
      const float* p = c < a.k1 ? a.x1 + (int64_t)r * a.ld1 + c
                                : a.x2 + (a.idx2 ? (int64_t)a.idx2[r] : (int64_t)r / a.div2) * a.ld2 + (c - a.k1);
      v = *reinterpret_cast<const f32x4*>(p);
    }
    *reinterpret_cast<f32x4*>(&xs[r * ktot + c]) = v;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = (blockIdx.x * 4 + wave) * 4;
  float acc[4][R];
#pragma unroll
  for (int o = 0; o < 4; ++o)
#pragma unroll
    for (int r = 0; r < R; ++r) acc[o][r] = 0.f;
  for (int kc = 0; kc < ktot; kc += 256) {
    const int k = kc + lane * 4;
    const bool kin = k < ktot;
    f32x4 wv[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      const int n = n0 + o < a.n_out ? n0 + o : 0;
      wv[o] = *reinterpret_cast<const f32x4*>(a.w + (int64_t)n * a.ldw + (kin ? k : 0));
    }
    if (kin) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(&xs[r * ktot + k]);
#pragma unroll
        for (int o = 0; o < 4; ++o) acc[o][r] += wv[o][0] * xv[0] + wv[o][1] * xv[1] + wv[o][2] * xv[2] + wv[o][3] * xv[3];
      }
    }
  }
#pragma unroll
  for (int o = 0; o < 4; ++o)
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float v = acc[o][r];
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
      acc[o][r] = v;
    }
  if (lane < 4 * R) {                         // lane -> (o, r); every lane holds all sums after the xor reduction
    const int o = lane / R, r = lane % R;
    const int n = n0 + o;
    if (n < a.n_out && r < rows) {
      float v = 0.f;
#pragma unroll
      for (int oo = 0; oo < 4; ++oo)
#pragma unroll
        for (int rr = 0; rr < R; ++rr) v = (oo == o && rr == r) ? acc[oo][rr] : v;
      v += a.bias ? a.bias[n] : 0.f;
      if (a.res) v += a.res[(a.res_mod > 0 ? r % a.res_mod : r) * a.ldres + n];
      a.y[(int64_t)r * a.ldy + n] = cmr_act(v, a.act, a.act_param);
    }
  }
}

__device__ __attribute__((aligned(16))) float cmr_zero16[4] = {0.f, 0.f, 0.f, 0.f};   // NOT const: hipcc folds loads of a const zero page into predicated loads + 0

// Weight-stationary, barrier-free variant for K = k1 + k2 <= 128 (every hot-path GEMM except the two
// K = 1024 / 4096 ones), computed TRANSPOSED: D'[cout][row] = W . X^T.
//   * the [32*NT couts][K] weight block is loaded into LDS once per workgroup and is the MFMA A operand;
//   * X rows are the B operand and come straight from global memory in operand layout (lane (row, h)
//     reads the 16 B at k = 8g + 4h of ITS row; every row is read exactly once, nothing to share), the
//     next 64-deep K segment / next tile is prefetched into registers while the current one is multiplied;
//   * there is no __syncthreads in the loop: waves stream 32-row tiles independently;
//   * in the transposed product a lane owns one output row and its 16 accumulator registers are 4 x 4
//     consecutive output channels, so bias / residual / store are float4 accesses (and the registers
//     of one layer are directly the B operand of the next -- the hook for chained layers).
// Rows are HBM-bound at these shapes ((K + Nout) * 4 bytes vs 2*K*Nout flops per row).
// NT = 32-wide cout tiles per workgroup (1, 2, 4); G = k-groups (of 8) per register segment (1, 2, 4, 8); NSEG
// segments cover K: KPAD = NSEG * 8 * G >= k1 + k2 (weights beyond k1 + k2 are zero in LDS, X loads beyond it read
// zeros).  AC = activation class: 0 -> v > 0 ? v : v * slope (none / ReLU / LeakyReLU with slope 1 / 0 / p),
// 1 -> erf-GELU, 2 -> elu+1.
//
// The tile loop is written so that hipcc sees ONE straight-line body (segments unrolled, no conditional load, no
// 64-bit division, gather indices fetched a tile ahead): its s_waitcnt bookkeeping is exact only then, and with any
// branch around a load it falls back to vmcnt(0), which serialises the X prefetch of the next tile, the residual
// loads and every store (stores count in vmcnt on gfx9) behind each other.
template <int AC>
__device__ __forceinline__ float ws_act(float v, float slope) {
  if (AC == 1) return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
  if (AC == 2) return v > 0.f ? v + 1.f : expf(v);
  return v > 0.f ? v : v * slope;
}

#ifndef CMR_WS_MINB
#define CMR_WS_MINB 1
#endif
template <int NT, int G, int NSEG, int AC>
__global__ __launch_bounds__(256, CMR_WS_MINB) void linear_ws_kernel(const LinearArgs a) {
  constexpr int KPAD = NSEG * 8 * G, LDWS = KPAD + 4;
  constexpr bool RES_EARLY = NT < 4;       // residual rows requested before the tile's MFMAs (registers permitting)
  extern __shared__ __attribute__((aligned(16))) float Ws[];   // [32*NT][KPAD + 4] weights, then [32*NT] bias
  float* Bs = Ws + 32 * NT * LDWS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  const int col_base = blockIdx.y * 32 * NT;
  const int ktot = a.k1 + a.k2;
  const uint32_t rows = (uint32_t)a.rows;  // < 2^31 (dispatch)
  const uint32_t ntiles = (rows + 31) / 32;
  const uint32_t tstride = gridDim.x * 4;
  const float* x2 = a.x2 ? a.x2 : a.x1;

  struct TileRows { const float* p1; const float* p2; bool valid; };
  auto load_idx = [&](uint32_t tile) -> int32_t {            // gather index of this lane's row (zero page when unused)
    const uint32_t row = tile * 32 + l31;
    const int32_t* p = (a.idx2 && tile < ntiles && row < rows) ? a.idx2 + row : reinterpret_cast<const int32_t*>(cmr_zero16);
    return *p;
  };
  auto tile_rows = [&](uint32_t tile, int32_t idxv) -> TileRows {
    uint32_t row = tile * 32 + l31;
    const bool valid = tile < ntiles && row < rows;
    row = valid ? row : 0;
    const int64_t r2 = a.idx2 ? (int64_t)idxv : (int64_t)(row / (uint32_t)a.div2);
    return {a.x1 + (int64_t)row * a.ld1, x2 + r2 * a.ld2, valid};
  };
  auto load_seg = [&](const TileRows& r, int seg, f32x4 (&dst)[G]) {
#pragma unroll
    for (int i = 0; i < G; ++i) {
      const int kk = (seg * G + i) * 8 + 4 * h;
      const float* p = kk < a.k1 ? r.p1 + kk : r.p2 + (kk - a.k1);
      p = (r.valid && kk < ktot) ? p : cmr_zero16;
      dst[i] = *reinterpret_cast<const f32x4*>(p);
    }
  };
  // residual rows: unconditional loads at clamped indices (rows / couts past the end re-read a valid element and
  // are never stored); without a residual the base is the zero page with stride 0.  Index clamps stay selects,
  // pointer selects on a per-lane condition are turned back into branches by hipcc.
  const float* resb = a.res ? a.res : cmr_zero16;
  const int64_t res_ld = a.res ? a.ldres : 0;
  const int res_cmax = a.res ? a.n_out - 4 : 0;
  auto load_res = [&](uint32_t tile, f32x4 (&r4)[NT][4]) {
    uint32_t row = tile * 32 + l31;
    row = row < rows ? row : 0;
    const uint32_t rr = a.res_mod > 0 ? row % (uint32_t)a.res_mod : row;
    const float* rp = resb + (int64_t)rr * res_ld;
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = col_base + n * 32 + 8 * q + 4 * h;
        r4[n][q] = *reinterpret_cast<const f32x4*>(rp + (c < res_cmax ? c : res_cmax));
      }
  };

  f32x4 xcur[G], xnxt[G];
  uint32_t tile = blockIdx.x * 4 + wave;
  TileRows rc = tile_rows(tile, load_idx(tile));
  load_seg(rc, 0, xcur);                   // in flight while the weights are staged
  int32_t idn = load_idx(tile + tstride);
  {
    constexpr int C4N = KPAD / 4;
    for (int e = tid; e < 32 * NT * C4N; e += 256) {
      const int n = e / C4N, c = (e % C4N) * 4;
      const bool ok = col_base + n < a.n_out && c < ktot;
      const float* p = ok ? a.w + (int64_t)(col_base + n) * a.ldw + c : cmr_zero16;
      *reinterpret_cast<f32x4*>(&Ws[n * LDWS + c]) = *reinterpret_cast<const f32x4*>(p);
    }
    if (tid < 32 * NT) Bs[tid] = (a.bias && col_base + tid < a.n_out) ? a.bias[col_base + tid] : 0.f;
  }
  __syncthreads();

  for (; tile < ntiles; tile += tstride) {
    f32x4 r4[NT][4];
    if (RES_EARLY) load_res(tile, r4);
    const TileRows rn = tile_rows(tile + tstride, idn);      // idn was requested one tile ago
    idn = load_idx(tile + 2 * tstride);
    f32x16 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
#pragma unroll
    for (int seg = 0; seg < NSEG; ++seg) {
      if (seg + 1 < NSEG) load_seg(rc, seg + 1, xnxt);
      else load_seg(rn, 0, xnxt);                            // next tile of this wave (zero page past the end)
      const float* wrow = Ws + l31 * LDWS + seg * G * 8 + 4 * h;
#pragma unroll
      for (int i = 0; i < G; ++i) {
        f32x4 wv[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) wv[n] = *reinterpret_cast<const f32x4*>(wrow + n * 32 * LDWS + i * 8);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int n = 0; n < NT; ++n) acc[n] = cmr_mfma32(wv[n][j], xcur[i][j], acc[n]);
      }
#pragma unroll
      for (int i = 0; i < G; ++i) xcur[i] = xnxt[i];
    }
    rc = rn;
    if (!RES_EARLY) load_res(tile, r4);
    // epilogue: this lane's row, 4 consecutive couts per register quad.  All values first, then the stores.
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int cl = n * 32 + 8 * q + 4 * h;
        f32x4 v = {acc[n][4 * q], acc[n][4 * q + 1], acc[n][4 * q + 2], acc[n][4 * q + 3]};
        v += *reinterpret_cast<const f32x4*>(&Bs[cl]);
        v += r4[n][q];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = ws_act<AC>(v[e], a.act_param);
        r4[n][q] = v;
      }
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int q = 0; q < 4; ++q) cmr_pin(r4[n][q]);
    const uint32_t row = tile * 32 + l31;
    if (row < rows) {
      float* yp = a.y + (int64_t)row * a.ldy + col_base + 4 * h;
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (col_base + n * 32 + 8 * q + 4 * h < a.n_out) *reinterpret_cast<f32x4*>(yp + n * 32 + 8 * q) = r4[n][q];
    }
  }
}

// Row-streaming fast path of the weight-stationary kernel for the shapes that carry most of this entry point's bytes:
// X [rows][64] contiguous (K = 64, no second source), Y [rows][32 NT] contiguous, residual (optional) laid out like Y.
// linear_ws_kernel reads X in MFMA operand layout -- lane (row, h) takes 16 bytes of ITS row, so one wave instruction touches 32 cache
// lines for 32 useful bytes each, and the stores do the same: the texture path, not HBM, sets its rate (2.2-3.4 TB/s measured).  Here
// a tile of 32 rows is what it is in memory, 8 KB of contiguous bytes: lane (r4, c) = (lane / 16, lane % 16) reads chunk c of row
// 4 i + r4 with instruction i (1 KB contiguous per wave instruction, whole 256-B rows), TWO tiles ahead; the tile is turned into operand
// layout through a wave-private 8 KB LDS buffer (chunk c of row r stored at position c ^ (r & 15): conflict-free for the row-wise writes
// and for the operand reads, where 16 lanes read 16 different rows at one chunk), the accumulators go back through the same buffer and
// bias / residual / activation / store happen in the coalesced layout (float4 of 4 consecutive couts per lane, whole rows per
// instruction).  No workgroup barrier after the weights are staged; every wave walks its own tiles.
template <int NT, int AC>
__global__ __launch_bounds__(256) void linear_row64_kernel(const LinearArgs a) {
  constexpr int LDWS = 64 + 4;
  constexpr int OC = 8 * NT;                 // 16-byte chunks per output row
  constexpr int RPI = 64 / OC;               // output rows per store instruction
  constexpr int NI = 32 / RPI;               // store instructions per tile
  extern __shared__ __attribute__((aligned(16))) float Ws[];    // [32 NT][68] weights | [32 NT] bias | [4 waves][32][64] tile buffers
  float* Bs = Ws + 32 * NT * LDWS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* T = Bs + 32 * NT + wave * 2048;
  const int h = lane >> 5, l31 = lane & 31;
  const int c = lane & 15, r4 = lane >> 4;
  const uint32_t rows = (uint32_t)a.rows;
  const uint32_t ntiles = (rows + 31) / 32;
  const uint32_t tstride = gridDim.x * 4;
  const uint32_t last16 = rows * 16 - 1;     // index of the last 16-byte chunk of X (clamp: rows past the end re-read it, never stored)

  auto load_tile = [&](uint32_t tile, f32x4 (&dst)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      uint32_t q = (tile * 32 + 4 * i + r4) * 16 + c;            // chunk index in X
      q = q < last16 ? q : last16;
      dst[i] = *reinterpret_cast<const f32x4*>(a.x1 + (size_t)q * 4);
    }
  };
  f32x4 xa[8], xb[8];
  uint32_t tile = blockIdx.x * 4 + wave;
  load_tile(tile, xa);
  load_tile(tile + tstride, xb);
  for (int e = tid; e < 32 * NT * 16; e += 256) {
    const int n = e >> 4, cc = (e & 15) * 4;
    *reinterpret_cast<f32x4*>(&Ws[n * LDWS + cc]) = *reinterpret_cast<const f32x4*>(a.w + (int64_t)n * a.ldw + cc);
  }
  if (tid < 32 * NT) Bs[tid] = a.bias ? a.bias[tid] : 0.f;
  __syncthreads();
  const float* resb = a.res ? a.res : cmr_zero16;
  const uint32_t res_on = a.res ? 1u : 0u;
  const uint32_t lasto = rows * OC - 1;

  for (; tile < ntiles; tile += tstride) {
    // ---- rows -> operand layout through the wave's buffer
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = 4 * i + r4;
      *reinterpret_cast<f32x4*>(&T[r * 64 + ((c ^ (r & 15)) << 2)]) = xa[i];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) xa[i] = xb[i];
    load_tile(tile + 2 * tstride, xb);                          // two tiles ahead
    f32x16 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    const float* wrow = Ws + l31 * LDWS + 4 * h;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      const f32x4 xv = *reinterpret_cast<const f32x4*>(&T[l31 * 64 + (((2 * g + h) ^ (l31 & 15)) << 2)]);
      f32x4 wv[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) wv[n] = *reinterpret_cast<const f32x4*>(wrow + n * 32 * LDWS + g * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[n] = cmr_mfma32(wv[n][j], xv[j], acc[n]);
    }
    // ---- accumulators (lane = row l31, couts 32 n + 8 q + 4 h ..) -> coalesced layout through the same buffer (row stride 4 OC floats)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v = {acc[n][4 * q], acc[n][4 * q + 1], acc[n][4 * q + 2], acc[n][4 * q + 3]};
        const int ch = 8 * n + 2 * q + h;
        *reinterpret_cast<f32x4*>(&T[l31 * (4 * OC) + ((ch ^ (l31 & (OC - 1))) << 2)]) = v;
      }
    const int oc = lane & (OC - 1), orr = lane / OC;            // this lane's output chunk and row within a store instruction
    const f32x4 bv = *reinterpret_cast<const f32x4*>(&Bs[4 * oc]);
    f32x4 ov[NI], rv[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      uint32_t q = (tile * 32 + RPI * i + orr) * OC + oc;
      q = q < lasto ? q : lasto;
      rv[i] = *reinterpret_cast<const f32x4*>(resb + (size_t)(q * res_on) * 4);
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int r = RPI * i + orr;
      ov[i] = *reinterpret_cast<const f32x4*>(&T[r * (4 * OC) + ((oc ^ (r & (OC - 1))) << 2)]);
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      f32x4 v = ov[i] + bv;
      if (a.res) v += rv[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = ws_act<AC>(v[e], a.act_param);
      ov[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) cmr_pin(ov[i]);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const uint32_t row = tile * 32 + RPI * i + orr;
      if (row < rows) *reinterpret_cast<f32x4*>(a.y + ((size_t)row * OC + oc) * 4) = ov[i];
    }
  }
}

template <int NT, int AC>
int launch_linear_row64_a(LinearArgs a, hipStream_t stream) {
  const size_t smem = ((size_t)32 * NT * (64 + 4 + 1) + 4 * 2048) * sizeof(float);
  static CmrSmemCache granted{};
  if (cmr_grant_smem(reinterpret_cast<const void*>(linear_row64_kernel<NT, AC>), smem, granted) != CMR_OK) return CMR_ELAUNCH;
  if (AC == 0) a.act_param = a.act == CMR_ACT_NONE ? 1.f : (a.act == CMR_ACT_RELU ? 0.f : a.act_param);
  const int64_t ntiles = (a.rows + 31) / 32;
  int64_t gx = (ntiles + 3) / 4;
  if (gx > 768) gx = 768;                   // up to 3 resident workgroups per CU
  hipLaunchKernelGGL((linear_row64_kernel<NT, AC>), dim3((unsigned)gx), dim3(256), smem, stream, a);
  return cmr_launch_status();
}

template <int NT>
int launch_linear_row64(const LinearArgs& a, hipStream_t stream) {
  if (a.act == CMR_ACT_GELU) return launch_linear_row64_a<NT, 1>(a, stream);
  if (a.act == CMR_ACT_ELU1) return launch_linear_row64_a<NT, 2>(a, stream);
  return launch_linear_row64_a<NT, 0>(a, stream);
}

// Split-K variant: one workgroup = one 32-row tile x 64 output channels, its 8 waves each take 1/8 of K
// (operands straight from global / L2 in fragment layout, 4 k-groups prefetched ahead), partial
// accumulators are summed through LDS and wave 0 runs the float4 epilogue.
// PATCH: the rows are the P x P patches of an NHWC map [B][H][W][C] (a stride-P convolution as a GEMM: ImageViT.py:37-56), read in place:
// row (b, Y, X) = the P row segments of P C contiguous floats at ((b H + P Y + py) W + P X) C -- no patchified copy of the map
// (cmr_patchify_nhwc_f32 wrote and this kernel re-read 55 MB at BASELINE configs[1]: 22 + 71 us).
struct PatchGeom { int H, W, C, P; };
template <bool PATCH>
__global__ __launch_bounds__(512) void linear_splitk_kernel(const LinearArgs a, const PatchGeom pg) {
  __shared__ __attribute__((aligned(16))) float red[7 * 32 * 64];      // partial tiles of waves 1..7
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int col_base = blockIdx.y * 64;
  const int64_t row0 = (int64_t)blockIdx.x * 32;
  int64_t row = row0 + l31;
  const bool valid = row < a.rows;
  if (!valid) row = 0;
  const int kper = a.k1 / 8;                          // multiple of 8 (k1 % 64 == 0)
  const float* xp;
  int pc = 0, seg_skip = 0;                           // PATCH: floats per patch row segment, distance from a segment's end to the next one's start
  if constexpr (PATCH) {
    const int tw = pg.W / pg.P, th = pg.H / pg.P;
    const int X = (int)(row % tw), Y = (int)((row / tw) % th), b = (int)(row / ((int64_t)tw * th));
    pc = pg.P * pg.C;
    seg_skip = pg.W * pg.C - pc;
    const int k0 = wave * kper + 4 * h, py0 = k0 / pc;
    xp = a.x1 + (((int64_t)b * pg.H + pg.P * Y + py0) * pg.W + pg.P * X) * pg.C + (k0 - py0 * pc);
  } else {
    xp = a.x1 + row * a.ld1 + wave * kper + 4 * h;
  }
  // offset of k-group g of this lane from xp: g * 8 floats, plus the gaps between the patch row segments crossed on the way
  const int koff0 = PATCH ? (wave * kper + 4 * h) % (pc > 0 ? pc : 1) : 0;
  auto xoff = [&](int g) __attribute__((always_inline)) {
    if constexpr (PATCH) return g * 8 + ((koff0 + g * 8) / pc) * seg_skip;
    else return g * 8;
  };
  const int c0n = col_base + l31, c1n = col_base + 32 + l31;
  const float* w0 = a.w + (int64_t)(c0n < a.n_out ? c0n : 0) * a.ldw + wave * kper + 4 * h;
  const float* w1 = a.w + (int64_t)(c1n < a.n_out ? c1n : 0) * a.ldw + wave * kper + 4 * h;
  f32x16 acc[2];
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  const int ng = kper / 8;
  f32x4 xc[4], wc0[4], wc1[4], xn[4], wn0[4], wn1[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int g = i < ng ? i : 0;
    xc[i] = *reinterpret_cast<const f32x4*>(xp + xoff(g));
    wc0[i] = *reinterpret_cast<const f32x4*>(w0 + g * 8);
    wc1[i] = *reinterpret_cast<const f32x4*>(w1 + g * 8);
  }
  for (int g0 = 0; g0 < ng; g0 += 4) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int g = g0 + 4 + i < ng ? g0 + 4 + i : 0;
      xn[i] = *reinterpret_cast<const f32x4*>(xp + xoff(g));
      wn0[i] = *reinterpret_cast<const f32x4*>(w0 + g * 8);
      wn1[i] = *reinterpret_cast<const f32x4*>(w1 + g * 8);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (g0 + i < ng) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[0] = cmr_mfma32(wc0[i][j], xc[i][j], acc[0]);
          acc[1] = cmr_mfma32(wc1[i][j], xc[i][j], acc[1]);
        }
      }
#pragma unroll
    for (int i = 0; i < 4; ++i) { xc[i] = xn[i]; wc0[i] = wn0[i]; wc1[i] = wn1[i]; }
  }
  if (wave > 0) {
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[((wave - 1) * 32 + n * 16 + r) * 64 + lane] = acc[n][r];
  }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int w = 0; w < 7; ++w) acc[n][r] += red[(w * 32 + n * 16 + r) * 64 + lane];
  if (!valid) return;
  row = row0 + l31;
  const int64_t rr = a.res ? (a.res_mod > 0 ? row % a.res_mod : row) : 0;
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c0 = col_base + n * 32 + 8 * q + 4 * h;
      if (c0 >= a.n_out) continue;
      f32x4 v = {acc[n][4 * q], acc[n][4 * q + 1], acc[n][4 * q + 2], acc[n][4 * q + 3]};
      if (a.bias) v += *reinterpret_cast<const f32x4*>(a.bias + c0);
      if (a.res) v += *reinterpret_cast<const f32x4*>(a.res + rr * a.ldres + c0);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = cmr_act(v[e], a.act, a.act_param);
      *reinterpret_cast<f32x4*>(a.y + row * a.ldy + c0) = v;
    }
}

template <int NT, int G, int NSEG, int AC>
int launch_linear_ws_ga(LinearArgs a, hipStream_t stream) {
  const size_t smem = (size_t)32 * NT * (NSEG * 8 * G + 4 + 1) * sizeof(float);      // weights + bias
  static CmrSmemCache granted{};
  if (cmr_grant_smem(reinterpret_cast<const void*>(linear_ws_kernel<NT, G, NSEG, AC>), smem, granted) != CMR_OK) return CMR_ELAUNCH;
  if (AC == 0) a.act_param = a.act == CMR_ACT_NONE ? 1.f : (a.act == CMR_ACT_RELU ? 0.f : a.act_param);
  const int64_t ntiles = (a.rows + 31) / 32;
  int64_t gx = (ntiles + 3) / 4;
  if (gx > 256 * (CMR_WS_MINB > 2 ? CMR_WS_MINB : 2)) gx = 256 * (CMR_WS_MINB > 2 ? CMR_WS_MINB : 2);   // resident workgroups per CU x 256; each wave walks its tiles with stride
  dim3 grid((unsigned)gx, (unsigned)((a.n_out + 32 * NT - 1) / (32 * NT)));
  hipLaunchKernelGGL((linear_ws_kernel<NT, G, NSEG, AC>), grid, dim3(256), smem, stream, a);
  return cmr_launch_status();
}

template <int NT, int G, int NSEG>
int launch_linear_ws_g(const LinearArgs& a, hipStream_t stream) {
  if (a.act == CMR_ACT_GELU) return launch_linear_ws_ga<NT, G, NSEG, 1>(a, stream);
  if (a.act == CMR_ACT_ELU1) return launch_linear_ws_ga<NT, G, NSEG, 2>(a, stream);
  return launch_linear_ws_ga<NT, G, NSEG, 0>(a, stream);
}

template <int NT>
int launch_linear_ws(const LinearArgs& a, hipStream_t stream) {
  const int ng = (a.k1 + a.k2 + 7) / 8;     // k-groups of 8 (<= 16)
  if (NT < 4) {
    if (ng > 8) return launch_linear_ws_g<NT, 8, 2>(a, stream);
    if (ng > 4) return launch_linear_ws_g<NT, 8, 1>(a, stream);
  } else {                                  // NT = 4 keeps 4-group segments (registers)
    if (ng > 12) return launch_linear_ws_g<NT, 4, 4>(a, stream);
    if (ng > 8) return launch_linear_ws_g<NT, 4, 3>(a, stream);
    if (ng > 4) return launch_linear_ws_g<NT, 4, 2>(a, stream);
  }
  if (ng > 2) return launch_linear_ws_g<NT, 4, 1>(a, stream);
  if (ng == 2) return launch_linear_ws_g<NT, 2, 1>(a, stream);
  return launch_linear_ws_g<NT, 1, 1>(a, stream);
}

// LayerNorm over exactly 64 channels; 16 lanes per row (float4 each).  y = LN(x)*g + b (+ res)
__global__ __launch_bounds__(256) void layernorm64_kernel(const float* __restrict__ x, int64_t ldx,
                                                          const float* __restrict__ g, const float* __restrict__ b,
                                                          float eps, const float* __restrict__ res, int64_t ldres,
                                                          float* __restrict__ y, int64_t ldy, int64_t rows) {
  const int tid = threadIdx.x;
  const int64_t row = (int64_t)blockIdx.x * 16 + (tid >> 4);
  const int c = (tid & 15) * 4;
  const bool ok = row < rows;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (ok) v = *reinterpret_cast<const f32x4*>(x + row * ldx + c);
  float s = (v[0] + v[1]) + (v[2] + v[3]);
#pragma unroll
  for (int m = 1; m < 16; m <<= 1) s += __shfl_xor(s, m);
  const float mean = s * (1.f / 64.f);
  f32x4 d = {v[0] - mean, v[1] - mean, v[2] - mean, v[3] - mean};
  float q = (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
#pragma unroll
  for (int m = 1; m < 16; m <<= 1) q += __shfl_xor(q, m);
  const float rstd = 1.f / sqrtf(q * (1.f / 64.f) + eps);
  if (ok) {
    const f32x4 gv = *reinterpret_cast<const f32x4*>(g + c);
    const f32x4 bv = *reinterpret_cast<const f32x4*>(b + c);
    f32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = d[i] * rstd * gv[i] + bv[i];
    if (res) {
      const f32x4 rv = *reinterpret_cast<const f32x4*>(res + row * ldres + c);
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] += rv[i];
    }
    *reinterpret_cast<f32x4*>(y + row * ldy + c) = o;
  }
}

}  // namespace

// ---- fp32 row maps with the weights in REGISTERS (K = 64, n_out <= 64, no second source) --------------------------------------------------
// linear_ws_kernel reads its weight fragments from LDS in front of every k-group (32 ds_read_b128 per tile) and keeps the generality of two
// sources / gathers in its loop; for the shape that carries most of this entry point's bytes the whole matrix is 64 VGPRs per lane (NT = 2:
// 16 KB over 64 lanes), so the tile loop here is: rows of the next tile requested, 64 MFMAs fed from registers only, float4 epilogue.
// SAME arithmetic in the SAME order as linear_ws_kernel (k-groups ascending, bias / residual / activation after the products): bit-identical.
template <int NT, int AC, bool RES>
__global__ __launch_bounds__(256) void linear_wreg_kernel(const LinearArgs a) {
  __shared__ __attribute__((aligned(16))) float Bs[32 * NT];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  if (tid < 32 * NT) Bs[tid] = (a.bias && tid < a.n_out) ? a.bias[tid] : 0.f;
  f32x4 wf[NT][8];                                  // lane (cout l31, half h): W[32 n + l31][8 i + 4 h .. + 3]
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int c = 32 * n + l31;
    const float* wr = a.w + (int64_t)(c < a.n_out ? c : 0) * a.ldw + 4 * h;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      wf[n][i] = *reinterpret_cast<const f32x4*>(wr + 8 * i);
      if (c >= a.n_out) wf[n][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  __syncthreads();
  const uint32_t rows = (uint32_t)a.rows;
  const uint32_t ntiles = (rows + 31) / 32, tstride = gridDim.x * 4;
  auto load_tile = [&](uint32_t tile, f32x4 (&dst)[8]) __attribute__((always_inline)) {
    uint32_t row = tile * 32 + l31;
    const bool valid = tile < ntiles && row < rows;
    const float* xp = valid ? a.x1 + (int64_t)row * a.ld1 + 4 * h : cmr_zero16;
    const int st = valid ? 8 : 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) dst[i] = *reinterpret_cast<const f32x4*>(xp + st * i);
  };
  // The launch used to cost its vector-memory issue time PLUS its matrix time (profiles/r03_pmc_linear.txt: waves issue-stalled 74 % of
  // their life at 42 % MFMA busy): a wave issued its 8 row loads and 8 stores in bursts, in order, in front of and behind its 64 MFMAs.
  // Here ONE load of the next tile and ONE store of the PREVIOUS tile's result follow each k-group's 8 MFMAs, so a memory instruction
  // waits for the address path while matrix instructions of the same wave are executing.  (Two tiles of rows in flight, or more resident
  // waves, were slower: every load instruction re-touches the lines of its 32 rows in a 32 KB L1.)
  f32x4 xc[8], xn[8];
  f32x4 prev[NT][4];                                // outputs of the previous tile of this wave, stored under this tile's products
  float* pyp = a.y;                                 // where they go (valid only when phave)
  bool phave = false;
  uint32_t tile = blockIdx.x * 4 + wave;
  load_tile(tile, xc);
  for (; tile < ntiles; tile += tstride) {
    const uint32_t row = tile * 32 + l31;
    f32x4 rv[RES ? NT : 1][4];
    if constexpr (RES) {
      const uint32_t rowc = row < rows ? row : 0;
      const uint32_t rr = a.res_mod > 0 ? rowc % (uint32_t)a.res_mod : rowc;
      const float* rp = a.res + (int64_t)rr * a.ldres + 4 * h;
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = 32 * n + 8 * q;
          rv[n][q] = *reinterpret_cast<const f32x4*>(rp + (c + 4 * h < a.n_out ? c : 0));
        }
    }
    const uint32_t nrow = (tile + tstride) * 32 + l31;
    const bool nvalid = tile + tstride < ntiles && nrow < rows;
    const float* nxp = nvalid ? a.x1 + (int64_t)nrow * a.ld1 + 4 * h : cmr_zero16;
    const int nst = nvalid ? 8 : 0;
    f32x16 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[n] = cmr_mfma32(wf[n][i][j], xc[i][j], acc[n]);
      xn[i] = *reinterpret_cast<const f32x4*>(nxp + nst * i);
      if (i < 4 * NT) {                                          // store i of the previous tile: (n, q) = (i / 4, i % 4)
        const int n = i / 4, q = i % 4;
        if (phave && 32 * n + 8 * q + 4 * h < a.n_out) *reinterpret_cast<f32x4*>(pyp + 32 * n + 8 * q) = prev[n][q];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v = {acc[n][4 * q], acc[n][4 * q + 1], acc[n][4 * q + 2], acc[n][4 * q + 3]};
        v += *reinterpret_cast<const f32x4*>(&Bs[32 * n + 8 * q + 4 * h]);
        if constexpr (RES) v += rv[n][q];
        else v += f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = ws_act<AC>(v[e], a.act_param);
        prev[n][q] = v;
      }
    phave = row < rows;
    pyp = a.y + (int64_t)(phave ? row : 0) * a.ldy + 4 * h;
#pragma unroll
    for (int i = 0; i < 8; ++i) xc[i] = xn[i];
  }
  if (phave) {                                                   // the last tile's result
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (32 * n + 8 * q + 4 * h < a.n_out) *reinterpret_cast<f32x4*>(pyp + 32 * n + 8 * q) = prev[n][q];
  }
}

template <int NT, int AC>
int launch_linear_wreg_a(LinearArgs a, hipStream_t stream) {
  if (AC == 0) a.act_param = a.act == CMR_ACT_NONE ? 1.f : (a.act == CMR_ACT_RELU ? 0.f : a.act_param);
  const int64_t ntiles = (a.rows + 31) / 32;
  int64_t gx = (ntiles + 3) / 4;
  if (gx > 512) gx = 512;
  if (a.res) hipLaunchKernelGGL((linear_wreg_kernel<NT, AC, true>), dim3((unsigned)gx), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((linear_wreg_kernel<NT, AC, false>), dim3((unsigned)gx), dim3(256), 0, stream, a);
  return cmr_launch_status();
}

template <int NT>
int launch_linear_wreg(const LinearArgs& a, hipStream_t stream) {
  if (a.act == CMR_ACT_GELU) return launch_linear_wreg_a<NT, 1>(a, stream);
  if (a.act == CMR_ACT_ELU1) return launch_linear_wreg_a<NT, 2>(a, stream);
  return launch_linear_wreg_a<NT, 0>(a, stream);
}

// Kernel-variant switches exist only in the A/B build (-DCMR_AB_SWITCHES -> cmr_agent_amd/lib/libcmr_hip_ab.so, include/cmr_hip_ab.h: tests
// that compare variants bit for bit, tools/*_bench.py); the product library has no mutable state: the dispatch below is a constant.
#ifdef CMR_AB_SWITCHES
static int g_linear_wreg = 1;            // register-weights kernel for K = 64 row maps
static int64_t g_linear_wreg_min_rows = 65537;
extern "C" int cmr_set_linear_wreg(int on, int64_t min_rows) {
  const int old = g_linear_wreg;
  g_linear_wreg = on ? 1 : 0;
  if (min_rows > 0) g_linear_wreg_min_rows = min_rows;
  return old;
}

static int g_linear_row64 = 1;
// 1 = use the row-streaming fast path where it applies (default): returns the previous value.
extern "C" int cmr_set_linear_row64(int on) {
  const int old = g_linear_row64;
  g_linear_row64 = on ? 1 : 0;
  return old;
}
#else
static constexpr int g_linear_wreg = 1, g_linear_row64 = 1;
static constexpr int64_t g_linear_wreg_min_rows = 65537;
#endif

extern "C" int cmr_linear_f32(const float* x1, int64_t ld1, int k1, const float* x2, int64_t ld2, int k2,
                              const int32_t* idx2, int64_t div2, const float* w, int64_t ldw, const float* bias,
                              const float* res, int64_t ldres, int64_t res_mod, float* y, int64_t ldy, int64_t rows,
                              int n_out, int act, float act_param, hipStream_t stream) {
  CMR_REQUIRE(x1 && w && y && rows >= 0 && n_out > 0 && k1 > 0);
  CMR_REQUIRE(k1 % 4 == 0 && ld1 % 4 == 0 && ldw % 4 == 0 && cmr_aligned16(x1) && cmr_aligned16(w));
  if (x2) CMR_REQUIRE(k2 > 0 && k2 % 4 == 0 && ld2 % 4 == 0 && cmr_aligned16(x2) && (idx2 || div2 >= 1));
  CMR_REQUIRE(act >= CMR_ACT_NONE && act <= CMR_ACT_ELU1);
  if (res) CMR_REQUIRE(ldres >= n_out || res_mod > 0);
  if (rows == 0) return CMR_OK;
  LinearArgs a{x1, ld1, k1, x2, ld2, x2 ? k2 : 0, idx2, div2 < 1 ? 1 : div2, w, ldw, bias, res, ldres, res_mod,
               y, ldy, rows, n_out, act, act_param};
  if (rows <= SK_ROWS && (size_t)SK_ROWS * (a.k1 + a.k2) * sizeof(float) <= 64 * 1024) {
    const dim3 grid((unsigned)((n_out + 15) / 16));
    if (rows <= 8)
      hipLaunchKernelGGL(linear_skinny_kernel<8>, grid, dim3(256), (size_t)8 * (a.k1 + a.k2) * sizeof(float), stream, a);
    else
      hipLaunchKernelGGL(linear_skinny_kernel<16>, grid, dim3(256), (size_t)16 * (a.k1 + a.k2) * sizeof(float), stream, a);
    return cmr_launch_status();
  }
  // float4 epilogue needs 4-aligned output channels (the host layer pads the few odd heads)
  const bool vec_ok = (n_out % 4 == 0) && (ldy % 4 == 0) && cmr_aligned16(y) && (!bias || cmr_aligned16(bias)) &&
                      (!res || (ldres % 4 == 0 && cmr_aligned16(res)));
  // contiguous [rows][64] -> [rows][64 | 32] (+ residual laid out like the output): the row-streaming fast path.  Measured
  // (tools/linear_bench.py, profiles/r03_linear_bench.txt): 10 240 rows 7.7 -> 6.2 us, 53 504 rows 12.6 -> 10.4 us, but 131 072 /
  // 214 016 rows 24.8 -> 28.8 / 33.8 -> 38.5 us: at 2 waves per SIMD the longer per-tile chain (two LDS round trips) costs more than
  // the whole-row accesses save, and the generic kernel takes the same 34 us with and without a residual (4.8 vs 3.2 TB/s), i.e. it
  // is not the texture path that bounds it at those sizes.  So: node / proxy-sized row sets only.
  if (vec_ok && !a.x2 && a.k1 == 64 && a.ld1 == 64 && (n_out == 64 || n_out == 32) && ldy == n_out && a.ldw % 4 == 0 &&
      (!res || (ldres == n_out && res_mod <= 0)) && rows >= 2048 && rows <= 65536 && g_linear_row64) {
    return n_out == 64 ? launch_linear_row64<2>(a, stream) : launch_linear_row64<1>(a, stream);
  }
  if (g_linear_wreg && vec_ok && !a.x2 && a.k1 == 64 && n_out <= 64 && rows >= g_linear_wreg_min_rows && rows < (int64_t)0x7fffffc0) {
    return n_out <= 32 ? launch_linear_wreg<1>(a, stream) : launch_linear_wreg<2>(a, stream);
  }
  if (a.k1 + a.k2 <= 128 && vec_ok && rows < (int64_t)0x7fffffc0) {       // weights fit in LDS: weight-stationary streaming kernel
    if (n_out <= 32 || (n_out > 64 && n_out <= 96)) return launch_linear_ws<1>(a, stream);
    if (n_out % 128 == 0) return launch_linear_ws<4>(a, stream);
    return launch_linear_ws<2>(a, stream);
  }
  // big K on few rows (ViT MLP fc2: K = 1024, patch embedding: K = 4096, a few thousand token rows): split K over
  // the 8 waves of a workgroup so that a 32-row tile is not one wave's serial chain of K/2 MFMAs
  if (vec_ok && !a.x2 && a.k1 % 64 == 0 && a.k1 >= 512 && rows <= 65536) {
    dim3 grid((unsigned)((rows + 31) / 32), (unsigned)((n_out + 63) / 64));
    hipLaunchKernelGGL(linear_splitk_kernel<false>, grid, dim3(512), 0, stream, a, PatchGeom{0, 0, 0, 0});
    return cmr_launch_status();
  }
  const unsigned gy = (unsigned)((n_out + 63) / 64);
  if (rows >= 16384) {
    dim3 grid((unsigned)((rows + 255) / 256), gy);
    hipLaunchKernelGGL(linear_kernel<2>, grid, dim3(256), 0, stream, a);
  } else {
    dim3 grid((unsigned)((rows + 127) / 128), gy);
    hipLaunchKernelGGL(linear_kernel<1>, grid, dim3(256), 0, stream, a);
  }
  return cmr_launch_status();
}

// ---- bf16 mode: the big contiguous row maps (K <= 128, n_out <= 128) on v_mfma_f32_32x32x16_bf16 -----------------------------------------
// In fp32 a 64 -> 64 layer sits at the fp32 ridge (16 FLOP/B) and linear_ws_kernel is bound by its MFMA chain (34 us for 110 MB at 214 016
// rows); on the bf16 cores the products take 1/16 of the time and the layer is a pure stream.  Transposed product (D[channel][row]: weights
// = A operand, rows = B operand): a lane owns one row, its accumulator registers are 4 x 4 consecutive output channels -> float4 epilogue.
// The whole weight matrix lives in registers as bf16 fragments (NT x KS x 4 VGPRs, converted once per wave); rows are converted on the fly.
typedef __bf16 lin_bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ lin_bf16x8 lin_pack8(const f32x4& lo, const f32x4& hi) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  typedef __bf16 b2 __attribute__((ext_vector_type(2)));
  uint4 w;
  w.x = __builtin_bit_cast(unsigned, __builtin_convertvector((f2){lo[0], lo[1]}, b2));
  w.y = __builtin_bit_cast(unsigned, __builtin_convertvector((f2){lo[2], lo[3]}, b2));
  w.z = __builtin_bit_cast(unsigned, __builtin_convertvector((f2){hi[0], hi[1]}, b2));
  w.w = __builtin_bit_cast(unsigned, __builtin_convertvector((f2){hi[2], hi[3]}, b2));
  return __builtin_bit_cast(lin_bf16x8, w);
}

// AC as in linear_ws_kernel (0: none / ReLU / LeakyReLU through the slope, 1: GELU, 2: elu + 1); RES: a residual operand is given.
// Register budget <= 128 (four waves per SIMD: a stream, not a GEMM): weight fragments 8 NT KS / 4 ... 32, the raw rows of the NEXT tile 8 KS,
// this tile's rows as bf16 4 KS, accumulators 16 NT (initialised with the bias from LDS).
template <int NT, int KS, int AC, bool RES>      // n_out <= 32 NT, K = 16 KS
#ifndef CMR_LRB_MINB
#define CMR_LRB_MINB 2
#endif
__global__ __launch_bounds__(256, CMR_LRB_MINB) void linear_rows_bf16_kernel(const LinearArgs a) {
  __shared__ __attribute__((aligned(16))) float Bs[32 * NT];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  if (tid < 32 * NT) Bs[tid] = (a.bias && tid < a.n_out) ? a.bias[tid] : 0.f;
  // weight fragments: lane (cout l31, half h) of tile t, step ks holds W[32 t + l31][16 ks + 8 h .. + 7] (rows past n_out: zeros)
  lin_bf16x8 wf[NT][KS];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int n = 32 * t + l31;
    const float* wr = a.w + (int64_t)(n < a.n_out ? n : 0) * a.ldw + 8 * h;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      f32x4 lo = *reinterpret_cast<const f32x4*>(wr + 16 * ks), hi = *reinterpret_cast<const f32x4*>(wr + 16 * ks + 4);
      if (n >= a.n_out) { lo = f32x4{0.f, 0.f, 0.f, 0.f}; hi = lo; }
      wf[t][ks] = lin_pack8(lo, hi);
    }
  }
  __syncthreads();
  const uint32_t rows = (uint32_t)a.rows;
  const uint32_t ntiles = (rows + 31) / 32, tstride = gridDim.x * 4;
  auto load_tile = [&](uint32_t tile, f32x4 (&dst)[2 * KS]) __attribute__((always_inline)) {
    uint32_t row = tile * 32 + l31;
    row = (tile < ntiles && row < rows) ? row : 0;
    const float* xp = a.x1 + (int64_t)row * a.ld1 + 8 * h;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      dst[2 * ks] = *reinterpret_cast<const f32x4*>(xp + 16 * ks);
      dst[2 * ks + 1] = *reinterpret_cast<const f32x4*>(xp + 16 * ks + 4);
    }
  };
  f32x4 raw[2 * KS];
  uint32_t tile = blockIdx.x * 4 + wave;
  load_tile(tile, raw);
  for (; tile < ntiles; tile += tstride) {
    lin_bf16x8 xb[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) xb[ks] = lin_pack8(raw[2 * ks], raw[2 * ks + 1]);
    load_tile(tile + tstride, raw);                            // next tile of this wave in flight under this one's products / stores
    const uint32_t row = tile * 32 + l31;
    f32x4 rv[RES ? NT : 1][4];
    if constexpr (RES) {
      const uint32_t rowc = row < rows ? row : 0;
      const uint32_t rr = a.res_mod > 0 ? rowc % (uint32_t)a.res_mod : rowc;
      const float* rp = a.res + (int64_t)rr * a.ldres + 4 * h;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = 32 * t + 8 * q;
          rv[t][q] = *reinterpret_cast<const f32x4*>(rp + (c + 4 * h < a.n_out ? c : 0));
        }
    }
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 bq = *reinterpret_cast<const f32x4*>(&Bs[32 * t + 8 * q + 4 * h]);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[t][4 * q + e] = bq[e];
      }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[t][ks], xb[ks], acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = acc[t][4 * q + e];
          if constexpr (RES) v += rv[t][q][e];
          acc[t][4 * q + e] = ws_act<AC>(v, a.act_param);
        }
    if (row < rows) {
      float* yp = a.y + (int64_t)row * a.ldy + 4 * h;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (32 * t + 8 * q + 4 * h < a.n_out)
            *reinterpret_cast<f32x4*>(yp + 32 * t + 8 * q) = f32x4{acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]};
    }
  }
}

template <int NT, int KS, int AC>
int launch_linear_rows_bf16_a(LinearArgs a, hipStream_t stream) {
  if (AC == 0) a.act_param = a.act == CMR_ACT_NONE ? 1.f : (a.act == CMR_ACT_RELU ? 0.f : a.act_param);
  const int64_t ntiles = (a.rows + 31) / 32;
  int64_t gx = (ntiles + 3) / 4;
  if (gx > 256 * CMR_LRB_MINB) gx = 256 * CMR_LRB_MINB;     // resident workgroups per CU x 256
  if (a.res) hipLaunchKernelGGL((linear_rows_bf16_kernel<NT, KS, AC, true>), dim3((unsigned)gx), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((linear_rows_bf16_kernel<NT, KS, AC, false>), dim3((unsigned)gx), dim3(256), 0, stream, a);
  return cmr_launch_status();
}

template <int NT, int KS>
int launch_linear_rows_bf16(const LinearArgs& a, hipStream_t stream) {
  if (a.act == CMR_ACT_GELU) return launch_linear_rows_bf16_a<NT, KS, 1>(a, stream);
  if (a.act == CMR_ACT_ELU1) return launch_linear_rows_bf16_a<NT, KS, 2>(a, stream);
  return launch_linear_rows_bf16_a<NT, KS, 0>(a, stream);
}

extern "C" int cmr_linear_rows_bf16_f32(const float* x, int64_t ldx, int k, const float* w, int64_t ldw, const float* bias, const float* res,
                                        int64_t ldres, int64_t res_mod, float* y, int64_t ldy, int64_t rows, int n_out, int act,
                                        float act_param, hipStream_t stream) {
  CMR_REQUIRE(x && w && y && rows > 0 && rows < (int64_t)0x7fffffc0 && n_out > 0 && act >= CMR_ACT_NONE && act <= CMR_ACT_ELU1);
  CMR_REQUIRE(ldx % 4 == 0 && ldw % 4 == 0 && ldy % 4 == 0 && n_out % 4 == 0 && cmr_aligned16(x) && cmr_aligned16(w) && cmr_aligned16(y) &&
              (!bias || cmr_aligned16(bias)) && (!res || (ldres % 4 == 0 && cmr_aligned16(res) && (ldres >= n_out || res_mod > 0))));
  const LinearArgs a{x, ldx, k, nullptr, 0, 0, nullptr, 1, w, ldw, bias, res, ldres, res_mod, y, ldy, rows, n_out, act, act_param};
  if (k == 64 && n_out <= 32) return launch_linear_rows_bf16<1, 4>(a, stream);
  if (k == 64 && n_out <= 64) return launch_linear_rows_bf16<2, 4>(a, stream);
  if (k == 64 && n_out <= 128) return launch_linear_rows_bf16<4, 4>(a, stream);
  if (k == 128 && n_out <= 32) return launch_linear_rows_bf16<1, 8>(a, stream);
  if (k == 128 && n_out <= 64) return launch_linear_rows_bf16<2, 8>(a, stream);
  if (k == 32 && n_out <= 32) return launch_linear_rows_bf16<1, 2>(a, stream);
  if (k == 32 && n_out <= 64) return launch_linear_rows_bf16<2, 2>(a, stream);
  return CMR_EUNSUPPORTED;
}

extern "C" int cmr_patch_embed_f32(const float* x_nhwc, int B, int H, int W, int C, int P, const float* w, int64_t ldw, const float* bias,
                                   const float* res, int64_t ldres, int64_t res_mod, float* y, int64_t ldy, int n_out, hipStream_t stream) {
  CMR_REQUIRE(x_nhwc && w && y && B > 0 && H > 0 && W > 0 && C > 0 && P > 0 && n_out > 0 && H % P == 0 && W % P == 0);
  const int K = P * P * C;
  const int64_t rows = (int64_t)B * (H / P) * (W / P);
  // each of the 8 waves takes K / 8 of the contraction in float4 pieces that must not straddle a patch row segment
  CMR_REQUIRE(K % 64 == 0 && K >= 512 && (P * C) % 8 == 0 && rows <= 65536 && (int64_t)B * H * W * C < 0x7fffffff);
  CMR_REQUIRE(n_out % 4 == 0 && ldy % 4 == 0 && ldw % 4 == 0 && cmr_aligned16(x_nhwc) && cmr_aligned16(w) && cmr_aligned16(y) &&
              (!bias || cmr_aligned16(bias)) && (!res || (ldres % 4 == 0 && cmr_aligned16(res))));
  const LinearArgs a{x_nhwc, (int64_t)K, K, nullptr, 0, 0, nullptr, 1, w, ldw, bias, res, ldres, res_mod, y, ldy, rows, n_out, CMR_ACT_NONE, 0.f};
  dim3 grid((unsigned)((rows + 31) / 32), (unsigned)((n_out + 63) / 64));
  hipLaunchKernelGGL(linear_splitk_kernel<true>, grid, dim3(512), 0, stream, a, PatchGeom{H, W, C, P});
  return cmr_launch_status();
}

extern "C" int cmr_layernorm64_f32(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps,
                                   const float* res, int64_t ldres, float* y, int64_t ldy, int64_t rows,
                                   hipStream_t stream) {
  CMR_REQUIRE(x && gamma && beta && y && rows >= 0);
  CMR_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0 && cmr_aligned16(x) && cmr_aligned16(y) && cmr_aligned16(gamma) &&
              cmr_aligned16(beta));
  if (res) CMR_REQUIRE(ldres % 4 == 0 && cmr_aligned16(res));
  if (rows == 0) return CMR_OK;
  hipLaunchKernelGGL(layernorm64_kernel, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, stream, x, ldx, gamma,
                     beta, eps, res, ldres, y, ldy, rows);
  return cmr_launch_status();
}
