// Fused ConvBNReLURes1D block (PointNN.py:260-282) in ONE kernel:
//     hid = LReLU(W1 x + b1) ;  y = LReLU(W2 hid + b2 + shortcut(x)) ,  shortcut = Wsc x + bsc | x
// with x = [x1 | x2[map]] (the un-materialised torch.cat / gather of its callers), BatchNorm folded.
//
// Everything is computed TRANSPOSED (D'[cout][row], weights = MFMA A operand, rows = B operand), so a
// lane owns ONE row and accumulator register 4q+e of cout-tile t holds channel 32t + 8q + 4h + e.  That
// is exactly the B-operand fragment layout of k-group 4t+q of the NEXT layer: the hidden activations go
// from accumulator registers straight into the second GEMM -- no LDS / HBM round trip -- and the
// identity shortcut is the input fragment itself.  All weights sit in LDS for the workgroup's lifetime
// (8 waves share one copy); waves stream 32-row tiles independently (no barrier in the loop).
//
// Per-batch bias: the agent's 3-D branch concatenates the broadcast global max-pool of the previous
// layer (CMRAgent.py:95-99); W [f | g] = Wa f + (Wb g) and the second term is a per-sample constant,
// so it enters as bias rows [B][C] (bias stride > 0) and the streamed K halves.
// Column max: the same branch max-pools the block's output over the points; the per-tile maxima are
// written as partials [tiles][CO] (reduced by cmr_colmax_partials_f32), saving a pass over y.
#include "cmr_common.h"

namespace {

__device__ __attribute__((aligned(16))) float cbr_zero[256] = {0.f};   // NOT const: hipcc folds loads of a const zero page into branches

struct CbrArgs {
  const float* x1; int64_t ld1;
  const float* x2; int64_t ld2; const int32_t* idx2; int64_t div2;   // second source (k2 = KX - k1) or null
  int k1;
  const float* w1; const float* b1; int64_t b1_stride;   // [CH][KX], bias [CH] or per batch [B][CH]
  const float* w2; const float* b2; int64_t b2_stride;   // [CO][CH], bias [CO] (b2 + bsc folded) or per batch
  const float* wsc;                                      // [CO][KX] or null (identity on the first KX couts)
  float* y; int64_t ldy;                                 // [rows][CO] or null
  float* colmax_part;                                    // [ceil(rows/32)][CO] or null
  int64_t rows; int64_t rows_per_batch; float slope;
};

constexpr int CBR_MAXB = 16;     // distinct per-batch bias rows staged in LDS

// The tile loop follows the rules of linear_ws_kernel (csrc/linear.hip): one straight-line body, no conditional load,
// 32-bit index math, gather indices and input fragments fetched a tile ahead, biases read from LDS, outputs pinned
// before the predicated stores.  (The first version loaded every bias float4 under a branch: 24 serialised global
// round trips per 32-row tile, 3 x the MFMA time of the tile.)
template <int KX, int CH, int CO, bool CONV_SC>
__global__ __launch_bounds__(512) void cbr_block_kernel(const CbrArgs a) {
  constexpr int LDX = KX + 4, LDH = CH + 4;
  constexpr int GX = KX / 8, GH = CH / 8;      // k-groups of the two GEMMs
  constexpr int T1 = (CH + 31) / 32, T2 = CO / 32;
  constexpr int CHP = 32 * T1;                 // hidden width padded to whole tiles
  constexpr bool PREFETCH = KX <= 64;          // next tile's input fragments in registers during this tile (KX = 128: no room)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* W1s = smem;                            // [32*T1][LDX]   (rows >= CH are zero)
  float* W2s = W1s + 32 * T1 * LDX;             // [CO][LDH]
  float* Wss = W2s + CO * LDH;                  // [CO][LDX]      (only if CONV_SC)
  float* B1s = Wss + (CONV_SC ? CO * LDX : 0);  // [nb][CHP]
  float* B2s = B1s + CBR_MAXB * CHP;            // [nb][CO]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  const uint32_t rows = (uint32_t)a.rows, rpb = (uint32_t)a.rows_per_batch;
  const int nb1 = a.b1_stride > 0 ? (int)((rows + rpb - 1) / rpb) : 1, nb2 = a.b2_stride > 0 ? (int)((rows + rpb - 1) / rpb) : 1;
  for (int e = tid; e < 32 * T1 * (KX / 4); e += 512) {
    const int n = e / (KX / 4), c = (e % (KX / 4)) * 4;
    *reinterpret_cast<f32x4*>(&W1s[n * LDX + c]) = *reinterpret_cast<const f32x4*>(n < CH ? a.w1 + (int64_t)n * KX + c : cbr_zero);
  }
  for (int e = tid; e < CO * (CH / 4); e += 512) {
    const int n = e / (CH / 4), c = (e % (CH / 4)) * 4;
    *reinterpret_cast<f32x4*>(&W2s[n * LDH + c]) = *reinterpret_cast<const f32x4*>(a.w2 + (int64_t)n * CH + c);
  }
  if (CONV_SC)
    for (int e = tid; e < CO * (KX / 4); e += 512) {
      const int n = e / (KX / 4), c = (e % (KX / 4)) * 4;
      *reinterpret_cast<f32x4*>(&Wss[n * LDX + c]) = *reinterpret_cast<const f32x4*>(a.wsc + (int64_t)n * KX + c);
    }
  for (int e = tid; e < nb1 * CHP; e += 512) B1s[e] = (e % CHP) < CH ? a.b1[(int64_t)(e / CHP) * a.b1_stride + e % CHP] : 0.f;
  for (int e = tid; e < nb2 * CO; e += 512) B2s[e] = a.b2[(int64_t)(e / CO) * a.b2_stride + e % CO];
  __syncthreads();

  const uint32_t ntiles = (rows + 31) / 32, tstride = gridDim.x * 8;
  const float* x2b = a.x2 ? a.x2 : a.x1;
  const int32_t* izero = reinterpret_cast<const int32_t*>(cbr_zero);
  auto load_idx = [&](uint32_t tile) -> int32_t {      // gather index of this lane's row one tile ahead (zero page when unused)
    const uint32_t row = tile * 32 + l31;
    const uint32_t ok = (a.idx2 != nullptr && tile < ntiles && row < rows) ? 1u : 0u;
    return (a.idx2 ? a.idx2 : izero)[ok * row];
  };
  auto load_x = [&](uint32_t tile, int32_t idxv, f32x4 (&xf)[GX]) {
    uint32_t row = tile * 32 + l31;
    row = (tile < ntiles && row < rows) ? row : 0;       // rows past the end recompute row 0 and are not stored
    const int64_t r2 = a.idx2 ? (int64_t)idxv : (int64_t)(row / (uint32_t)a.div2);
    const float* p1 = a.x1 + (int64_t)row * a.ld1 + 4 * h;
    const float* p2 = x2b + r2 * a.ld2 + 4 * h - a.k1;
#pragma unroll
    for (int g = 0; g < GX; ++g) xf[g] = *reinterpret_cast<const f32x4*>((g * 8 + 4 * h < a.k1 ? p1 : p2) + g * 8);
  };

  uint32_t tile = blockIdx.x * 8 + wave;
  f32x4 xf[GX], xn[PREFETCH ? GX : 1];
  int32_t idn = load_idx(tile);
  if (PREFETCH) {
    load_x(tile, idn, xf);
    idn = load_idx(tile + tstride);
  }
  for (; tile < ntiles; tile += tstride) {
    const uint32_t row = tile * 32 + l31;
    const bool valid = row < rows;
    const uint32_t batch = (valid ? row : 0) / rpb;
    if constexpr (PREFETCH) {
      load_x(tile + tstride, idn, xn);                   // next tile's fragments fly under this tile's GEMMs
      idn = load_idx(tile + 2 * tstride);
    } else {
      load_x(tile, idn, xf);
      idn = load_idx(tile + tstride);
    }
    // ---- GEMM 1: hid'[c1][row]
    f32x16 hid[T1];
#pragma unroll
    for (int t = 0; t < T1; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) hid[t][r] = 0.f;
    // weight fragments are software-pipelined one k-group ahead; the scheduling barriers keep the compiler
    // from hoisting ALL LDS reads of the fully unrolled loop (hundreds of VGPRs, spills)
    {
      f32x4 wc[T1], wn[T1];
#pragma unroll
      for (int t = 0; t < T1; ++t) wc[t] = *reinterpret_cast<const f32x4*>(&W1s[(t * 32 + l31) * LDX + 4 * h]);
#pragma unroll
      for (int g = 0; g < GX; ++g) {
        if (g + 1 < GX) {
#pragma unroll
          for (int t = 0; t < T1; ++t) wn[t] = *reinterpret_cast<const f32x4*>(&W1s[(t * 32 + l31) * LDX + (g + 1) * 8 + 4 * h]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int t = 0; t < T1; ++t) hid[t] = cmr_mfma32(wc[t][j], xf[g][j], hid[t]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < T1; ++t) wc[t] = wn[t];
      }
    }
    // bias + LeakyReLU on the hidden activations, in place (register 4q+e <-> channel 32t + 8q + 4h + e)
    {
      const float* b1 = B1s + (a.b1_stride > 0 ? batch : 0) * CHP + 4 * h;
#pragma unroll
      for (int g = 0; g < GH; ++g) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(b1 + g * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = hid[g / 4][(g % 4) * 4 + e] + bv[e];
          hid[g / 4][(g % 4) * 4 + e] = v > 0.f ? v : v * a.slope;
        }
      }
    }
    // ---- GEMM 2 (+ shortcut GEMM) : y'[c2][row]
    f32x16 acc[T2];
#pragma unroll
    for (int n = 0; n < T2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    {
      f32x4 wc[T2], wn[T2];
#pragma unroll
      for (int n = 0; n < T2; ++n) wc[n] = *reinterpret_cast<const f32x4*>(&W2s[(n * 32 + l31) * LDH + 4 * h]);
#pragma unroll
      for (int g = 0; g < GH; ++g) {
        if (g + 1 < GH) {
#pragma unroll
          for (int n = 0; n < T2; ++n) wn[n] = *reinterpret_cast<const f32x4*>(&W2s[(n * 32 + l31) * LDH + (g + 1) * 8 + 4 * h]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int n = 0; n < T2; ++n) acc[n] = cmr_mfma32(wc[n][j], hid[g / 4][(g % 4) * 4 + j], acc[n]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < T2; ++n) wc[n] = wn[n];
      }
    }
    if (CONV_SC) {
      f32x4 wc[T2], wn[T2];
#pragma unroll
      for (int n = 0; n < T2; ++n) wc[n] = *reinterpret_cast<const f32x4*>(&Wss[(n * 32 + l31) * LDX + 4 * h]);
#pragma unroll
      for (int g = 0; g < GX; ++g) {
        if (g + 1 < GX) {
#pragma unroll
          for (int n = 0; n < T2; ++n) wn[n] = *reinterpret_cast<const f32x4*>(&Wss[(n * 32 + l31) * LDX + (g + 1) * 8 + 4 * h]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int n = 0; n < T2; ++n) acc[n] = cmr_mfma32(wc[n][j], xf[g][j], acc[n]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < T2; ++n) wc[n] = wn[n];
      }
    }
    // ---- epilogue: all values first (bias from LDS), pinned, then the predicated stores / the column maxima
    const float* b2 = B2s + (a.b2_stride > 0 ? batch : 0) * CO + 4 * h;
    f32x4 ov[T2][4];
#pragma unroll
    for (int n = 0; n < T2; ++n)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = n * 32 + q * 8;
        const f32x4 bv = *reinterpret_cast<const f32x4*>(b2 + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float s = acc[n][4 * q + e] + bv[e];
          if (!CONV_SC && c < KX) s += xf[(c / 8)][e];          // identity shortcut: channel c+4h+e of x
          ov[n][q][e] = s > 0.f ? s : s * a.slope;
        }
      }
#pragma unroll
    for (int n = 0; n < T2; ++n)
#pragma unroll
      for (int q = 0; q < 4; ++q) cmr_pin(ov[n][q]);
    if (a.y && valid) {
      float* yrow = a.y + (int64_t)row * a.ldy + 4 * h;
#pragma unroll
      for (int n = 0; n < T2; ++n)
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(yrow + n * 32 + q * 8) = ov[n][q];
    }
    if (a.colmax_part) {
      const bool whole = tile * 32 + 32 <= rows;          // (wave-uniform) only the last tile has rows to mask
#pragma unroll
      for (int n = 0; n < T2; ++n)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 v = ov[n][q];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = cmr_rowmax32(whole || valid ? v[e] : -INFINITY);
          if (l31 == 16) *reinterpret_cast<f32x4*>(a.colmax_part + (int64_t)tile * CO + n * 32 + q * 8 + 4 * h) = v;
        }
    }
    if constexpr (PREFETCH) {
#pragma unroll
      for (int g = 0; g < GX; ++g) xf[g] = xn[g];
    }
  }
}

// out[b][c] = max over the tiles of batch b of part[tile][c]  (tiles_per_batch consecutive tiles per batch).
// Two launches: out <- -inf, then gridDim.z tile chunks per (batch, 64-channel block) fold their maxima into out with
// an atomic max on the ordered-integer image of the float (order independent, hence deterministic).  Round 1 ran one
// workgroup per (batch, channel block): 8-16 workgroups on the whole chip, 31 us for <= 1 MB.
__global__ void colmax_init_kernel(float* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = -INFINITY;
}

__device__ __forceinline__ void cbr_atomic_max(float* addr, float v) {
  if (v >= 0.f) atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
  else atomicMin(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

__global__ __launch_bounds__(256) void colmax_partials_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                              int tiles_per_batch, int C) {
  __shared__ float sm[256];
  const int b = blockIdx.x, cblk = blockIdx.y * 64;
  const int c = cblk + (threadIdx.x & 63), grp = threadIdx.x >> 6;        // 4 tile groups x 64 channels
  const int per = (tiles_per_batch + gridDim.z - 1) / gridDim.z;
  const int t0 = blockIdx.z * per, t1 = min(t0 + per, tiles_per_batch);
  float m = -INFINITY;
  if (c < C)
    for (int t = t0 + grp; t < t1; t += 4) m = fmaxf(m, part[((int64_t)b * tiles_per_batch + t) * C + c]);
  sm[threadIdx.x] = m;
  __syncthreads();
  if (grp == 0 && c < C) {
#pragma unroll
    for (int g = 1; g < 4; ++g) m = fmaxf(m, sm[threadIdx.x + 64 * g]);
    if (m > -INFINITY) cbr_atomic_max(out + (int64_t)b * C + c, m);
  }
}

// The same reduction in ONE launch, without the -inf fill and without atomics, when C is a multiple of 64: a 1024-thread
// workgroup per (batch, 64-channel block) = 64 tile groups x 16 channel quads, 8 x 16-byte loads in flight per thread
// (512 tiles x 64 channels = 128 KB per workgroup, L2 resident: the block kernel has just written it), then the 64 groups
// fold through LDS in two steps.  max is order independent: the same bits as the atomic form.
__global__ __launch_bounds__(1024) void colmax_direct_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                             int tiles_per_batch, int C) {
  __shared__ f32x4 red[64][16];
  const int b = blockIdx.x, c = blockIdx.y * 64 + 4 * (threadIdx.x & 15);
  const int q = threadIdx.x & 15, g = threadIdx.x >> 4;
  const float* p = part + (int64_t)b * tiles_per_batch * C + c;
  const f32x4 ninf = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  f32x4 m = ninf;
  for (int t0 = g; t0 < tiles_per_batch; t0 += 64 * 8) {
    f32x4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int t = t0 + 64 * i;
      v[i] = *reinterpret_cast<const f32x4*>(p + (int64_t)(t < tiles_per_batch ? t : t0) * C);      // (t0 again: harmless under max)
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[i][e]);
  }
  red[g][q] = m;
  __syncthreads();
  if (g < 8) {
#pragma unroll
    for (int i = 1; i < 8; ++i) {
      const f32x4 o = red[g + 8 * i][q];
#pragma unroll
      for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], o[e]);
    }
    red[g][q] = m;
  }
  __syncthreads();
  if (g == 0) {
#pragma unroll
    for (int i = 1; i < 8; ++i) {
      const f32x4 o = red[i][q];
#pragma unroll
      for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], o[e]);
    }
    *reinterpret_cast<f32x4*>(out + (int64_t)b * C + c) = m;
  }
}

// Round 6: the glue between two blocks of the agent's 3-D branch in ONE launch (CMRAgent.py:92-101: x = cat([feat, max over the points
// broadcast back]) in front of every ConvBNReLURes1D after the first).  The block kernel leaves per-tile maxima; the next block wants the
// broadcast half folded into per-sample biases  b1b = g W1[:, f:]^T + b1  and  b2b = g Wsc[:, f:]^T + b2.  That was three launches on the
// serial chain of every agent step (colmax_direct_kernel + 2 x linear_skinny_kernel: 30 - 45 us of a 280 - 370 us chain); here a
// 1024-thread workgroup per sample reduces its 64-channel maxima exactly as colmax_direct_kernel does (max is order independent) and then
// forms both bias rows the way linear_skinny_kernel does -- lane l < 16 multiplies the 4 consecutive k = 4 l .. 4 l + 3, a wave owns 4
// outputs, xor-shuffle reduction over the 64 lanes, bias added last.  The maxima are the same bits; the products agree with the three
// launches to the last place or two (the compiler pairs the four products of a lane differently in the 8-row kernel), 64-term fp32 sums.
__global__ __launch_bounds__(1024) void colmax_bias2_kernel(const float* __restrict__ part, int tiles_per_batch, const float* __restrict__ w1,
                                                            const float* __restrict__ b1, int n1, const float* __restrict__ w2,
                                                            const float* __restrict__ b2, int n2, float* __restrict__ gout,
                                                            float* __restrict__ y1, float* __restrict__ y2) {
  constexpr int C = 64;
  __shared__ f32x4 red[64][16];
  __shared__ __attribute__((aligned(16))) float gs[C];
  const int b = blockIdx.x, q = threadIdx.x & 15, g = threadIdx.x >> 4;
  const float* p = part + (int64_t)b * tiles_per_batch * C + 4 * q;
  f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  for (int t0 = g; t0 < tiles_per_batch; t0 += 64 * 8) {
    f32x4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int t = t0 + 64 * i;
      v[i] = *reinterpret_cast<const f32x4*>(p + (int64_t)(t < tiles_per_batch ? t : t0) * C);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[i][e]);
  }
  red[g][q] = m;
  __syncthreads();
  if (g < 8) {
#pragma unroll
    for (int i = 1; i < 8; ++i) {
      const f32x4 o = red[g + 8 * i][q];
#pragma unroll
      for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], o[e]);
    }
    red[g][q] = m;
  }
  __syncthreads();
  if (g == 0) {
#pragma unroll
    for (int i = 1; i < 8; ++i) {
      const f32x4 o = red[i][q];
#pragma unroll
      for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], o[e]);
    }
    *reinterpret_cast<f32x4*>(gs + 4 * q) = m;
    if (gout) *reinterpret_cast<f32x4*>(gout + (int64_t)b * C + 4 * q) = m;
  }
  __syncthreads();
  // the two skinny products: outputs [0, n1) of (w1, b1) then [0, n2) of (w2, b2); wave = 4 consecutive outputs per pass
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k = lane * 4;
  const bool kin = k < C;
  const f32x4 xv = *reinterpret_cast<const f32x4*>(gs + (kin ? k : 0));
  const int ntot = n1 + n2;
  for (int n0 = wave * 4; n0 < ntot; n0 += 64) {
    // (n1 % 4 == 0: the four outputs of a wave belong to one of the two products)
    const bool first = n0 < n1;
    const float* w = first ? w1 : w2;
    const float* bias = first ? b1 : b2;
    float* y = first ? y1 + (int64_t)b * n1 : y2 + (int64_t)b * n2;
    const int nn = first ? n0 : n0 - n1, nlim = first ? n1 : n2;
    float acc[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      const int n = nn + o < nlim ? nn + o : 0;
      const f32x4 wv = *reinterpret_cast<const f32x4*>(w + (int64_t)n * C + (kin ? k : 0));
      // one multiply + three fused multiply-adds, spelled out so that the result does not depend on the compiler's contraction choice
      float a = 0.f;
      if (kin) a += fmaf(wv[3], xv[3], fmaf(wv[2], xv[2], fmaf(wv[1], xv[1], wv[0] * xv[0])));
      acc[o] = a;
    }
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      float v = acc[o];
#pragma unroll
      for (int mm = 32; mm >= 1; mm >>= 1) v += __shfl_xor(v, mm);
      acc[o] = v;
    }
    if (lane < 4 && nn + lane < nlim) {
      float v = lane == 0 ? acc[0] : (lane == 1 ? acc[1] : (lane == 2 ? acc[2] : acc[3]));
      v += bias[nn + lane];
      y[nn + lane] = v;
    }
  }
}

template <int KX, int CH, int CO, bool CONV_SC>
int launch_cbr(const CbrArgs& a, hipStream_t stream) {
  constexpr int T1 = (CH + 31) / 32;
  constexpr size_t smem = (size_t)(32 * T1 * (KX + 4) + CO * (CH + 4) + (CONV_SC ? CO * (KX + 4) : 0) + CBR_MAXB * (32 * T1 + CO)) *
                          sizeof(float);      // weights + staged bias rows
  static_assert(smem <= 160 * 1024, "weights must fit in LDS");
  static CmrSmemCache granted{};
  if (cmr_grant_smem(reinterpret_cast<const void*>(cbr_block_kernel<KX, CH, CO, CONV_SC>), smem, granted) != CMR_OK) return CMR_ELAUNCH;
  const int64_t ntiles = (a.rows + 31) / 32;
  const int per_cu = smem > 80 * 1024 ? 1 : 2;
  int64_t grid = (ntiles + 7) / 8;
  if (grid > 256 * per_cu) grid = 256 * per_cu;
  hipLaunchKernelGGL((cbr_block_kernel<KX, CH, CO, CONV_SC>), dim3((unsigned)grid), dim3(512), smem, stream, a);
  return cmr_launch_status();
}

}  // namespace

// x = [x1[:, :k1] | x2[map][:, :kx-k1]]; hidden width ch = kx (ConvBNReLURes1D keeps the width in its first conv) except
// for the agent's per-batch-bias form, where the broadcast half of the input has been folded into b1 (ch > kx).
extern "C" int cmr_cbr_block_f32(const float* x1, int64_t ld1, int k1, const float* x2, int64_t ld2, const int32_t* idx2,
                                 int64_t div2, int kx, int ch, int co, const float* w1, const float* b1,
                                 int64_t b1_stride, const float* w2, const float* b2, int64_t b2_stride, const float* wsc,
                                 float* y, int64_t ldy, float* colmax_part, int64_t rows, int64_t rows_per_batch,
                                 float slope, hipStream_t stream) {
  CMR_REQUIRE(x1 && w1 && b1 && w2 && b2 && (y || colmax_part) && rows > 0 && rows_per_batch > 0);
  CMR_REQUIRE(k1 > 0 && k1 % 4 == 0 && k1 <= kx && (k1 == kx || x2) && ld1 % 4 == 0 && cmr_aligned16(x1));
  if (x2) CMR_REQUIRE(ld2 % 4 == 0 && cmr_aligned16(x2) && (idx2 || div2 >= 1));
  if (y) CMR_REQUIRE(ldy % 4 == 0 && cmr_aligned16(y));
  CMR_REQUIRE(cmr_aligned16(w1) && cmr_aligned16(w2) && cmr_aligned16(b1) && cmr_aligned16(b2) && b1_stride % 4 == 0 &&
              b2_stride % 4 == 0 && (!wsc || cmr_aligned16(wsc)) && (!colmax_part || cmr_aligned16(colmax_part)));
  const CbrArgs a{x1, ld1, x2, ld2, idx2, div2 < 1 ? 1 : div2, k1, w1, b1, b1_stride, w2, b2, b2_stride, wsc, y, ldy,
                  colmax_part, rows, rows_per_batch, slope};
  const bool conv = wsc != nullptr;
  if (rows >= (int64_t)0x7fffffc0 || ((b1_stride > 0 || b2_stride > 0) && (rows + rows_per_batch - 1) / rows_per_batch > CBR_MAXB))
    return CMR_EUNSUPPORTED;                 // the unfused path (cmr_linear_f32 per layer) serves these
  if (kx == 64 && ch == 64 && co == 64 && !conv) return launch_cbr<64, 64, 64, false>(a, stream);
  if (kx == 128 && ch == 128 && co == 64 && conv) return launch_cbr<128, 128, 64, true>(a, stream);
  if (kx == 64 && ch == 128 && co == 64 && conv) return launch_cbr<64, 128, 64, true>(a, stream);
  if (kx == 64 && ch == 128 && co == 128 && !conv) return launch_cbr<64, 128, 128, false>(a, stream);
  if (kx == 8 && ch == 8 && co == 64 && conv) return launch_cbr<8, 8, 64, true>(a, stream);
  return CMR_EUNSUPPORTED;
}

extern "C" int cmr_colmax_partials_f32(const float* part, float* out, int B, int tiles_per_batch, int C,
                                       hipStream_t stream) {
  CMR_REQUIRE(part && out && B > 0 && B <= 65535 && tiles_per_batch > 0 && C > 0);
  if (C % 64 == 0 && cmr_aligned16(part) && cmr_aligned16(out)) {
    hipLaunchKernelGGL(colmax_direct_kernel, dim3(B, C / 64), dim3(1024), 0, stream, part, out, tiles_per_batch, C);
    return cmr_launch_status();
  }
  hipLaunchKernelGGL(colmax_init_kernel, dim3((B * C + 255) / 256), dim3(256), 0, stream, out, B * C);
  int z = (tiles_per_batch + 31) / 32;             // >= 32 tiles per chunk: 8 loads per thread
  z = z < 1 ? 1 : (z > 64 ? 64 : z);
  hipLaunchKernelGGL(colmax_partials_kernel, dim3(B, (C + 63) / 64, z), dim3(256), 0, stream, part, out, tiles_per_batch, C);
  return cmr_launch_status();
}

// colmax of the block kernel's per-tile maxima (C = 64) + the two per-sample bias rows of the NEXT block in one launch (see
// colmax_bias2_kernel): g [B][64] (optional output), y1 [B][n1] = g w1^T + b1, y2 [B][n2] = g w2^T + b2; w1 [n1][64], w2 [n2][64] contiguous,
// n1 % 4 == 0, n2 % 4 == 0.  Other widths: CMR_EUNSUPPORTED (cmr_colmax_partials_f32 + cmr_linear_f32).
extern "C" int cmr_colmax_bias2_f32(const float* part, int B, int tiles_per_batch, int C, const float* w1, const float* b1, int n1,
                                    const float* w2, const float* b2, int n2, float* g, float* y1, float* y2, hipStream_t stream) {
  CMR_REQUIRE(part && w1 && b1 && w2 && b2 && y1 && y2 && B > 0 && B <= 65535 && tiles_per_batch > 0 && n1 > 0 && n2 > 0);
  if (C != 64 || n1 % 4 || n2 % 4) return CMR_EUNSUPPORTED;
  CMR_REQUIRE(cmr_aligned16(part) && cmr_aligned16(w1) && cmr_aligned16(w2) && (!g || cmr_aligned16(g)));
  hipLaunchKernelGGL(colmax_bias2_kernel, dim3(B), dim3(1024), 0, stream, part, tiles_per_batch, w1, b1, n1, w2, b2, n2, g, y1, y2);
  return cmr_launch_status();
}
