// Backward of a train-mode [1x1 conv / Linear -> BatchNorm -> LeakyReLU (+ residual)] layer on a big row map in ONE pass over the maps
// (round 4; reference Train_Geo.py:166-174 `loss.backward()` through models/PointNN.py:96-123 MiniPointNet and :260-282 ConvBNReLURes1D,
// Train_Agent.py:296-305 through CMRAgent.py:25-33).
//
// Layer:  h = x W^T + b,  z = lrelu_s(scale (h - mean) rstd ... ) = lrelu_s(BN(h) (+ res)).  Given dz:
//   d  = dz * act'(z)                                   (the gradient a residual branch added in front of the activation receives: dzm)
//   dh = scale * (d - c1 - xhat * c2),  xhat = (h - mean) rstd,  c1 = mean_rows(d),  c2 = mean_rows(d xhat)      (cmr_bn_bwd_coef_f32)
//   dW += dh^T x,   dx = dh W (+ what x already received)
// Op by op that is cmr_bn_bwd_f32's apply pass (reads dz, z, h, writes dh), cmr_linear_wgrad_f32 (reads dh, x) and cmr_linear_f32 on W^T
// (reads dh, writes dx): 6 reads + 2 writes of a [rows, 64] map.  Here: 4 reads + 1 write.  A workgroup (4 waves) walks blocks of 32 rows
// with a static stride; per block the 256 threads fetch whole rows of dz, z, h, x with float4 loads (one block ahead, in registers), turn
// (dz, z, h) into dh on the way into a double-buffered LDS tile, and then
//   weight gradient: the whole [n x k] gradient lives in the four waves' accumulators (v_mfma_f32_32x32x2_f32, operands ds_read_b32 with
//     lane = channel; one step contracts the row pair (r, r + 8): row stride = width + 36 floats puts the pair 32 banks apart);
//   data gradient:   wave w owns the input channels [w k/4, (w+1) k/4): v_mfma_f32_16x16x4_f32 with W in registers as the A operand
//     (lane 16 g + i: W[4 s + g][channel i]) and dh rows as the B operand (ds_read_b32, lane 16 g + j: dh[row j][4 s + g]; 36 j mod 64
//     are 16 distinct multiples of 4: conflict free), one float4 store per lane and 16 rows x 16 channels.
// Per block and workgroup: 64 + 64 MFMA32-equivalents = 2 048 matrix cycles per wave for 40 KB of HBM traffic (64 -> 64): balanced against
// 8 TB/s at ~75 % matrix occupancy; three workgroups per CU (51 KB of LDS each).  rows must be a multiple of 32 (every map this serves is
// B x a power of two); no predicated access anywhere.  Partials [workgroups][n][k] are summed in double in a fixed order.
#include "cmr_common.h"
#include "cmr_mfma16.h"

namespace {

struct BlbArgs {
  const float* dz; int64_t lddz;
  const float* z; int64_t ldz;          // activation output (mask)
  float slope;
  const float* h; int64_t ldh;          // BatchNorm input (BN only)
  const float* stat;                    // [4][n]: mean, rstd, scale, shift (cmr_bn_stats_f32)
  const float* coef;                    // [2][n]: c1, c2 (cmr_bn_bwd_coef_f32)
  float* dzm; int64_t lddzm;            // optional output: dz * act'(z)
  const float* x; int64_t ldx;
  const float* w; int64_t ldw;          // [n][k]
  const float* res; int64_t ldres;      // optional: added to dx (dx may alias it)
  float* dx; int64_t lddx;              // optional
  float* part;                          // [gridDim.x][n][k]
  float* part_b;                        // optional [gridDim.x][n]: column sums of dh (the bias gradient of a layer without BatchNorm)
  int64_t rows;
  // XL: x is the PREVIOUS layer's BatchNorm input; this layer's operand is lrelu_{xslope}(x * xstat[2] + xstat[3]) (never stored), and the
  // BatchNorm-backward sums of the previous layer (sum d', sum d' xhat' with d' = dx * act', over this workgroup's rows) go to xpart
  const float* xstat;                   // [4][k]
  float xslope;
  float* xpart;                         // [gridDim.x][2][k]
  int64_t seg_blocks;                   // row blocks per segment (rows / 32 when the map is one segment)
  int seg_groups;                       // workgroups per segment (gridDim.x = segments * seg_groups)
};

__device__ __forceinline__ f32x4 blb_fma4(f32x4 a, f32x4 b, f32x4 c) {
  return f32x4{__builtin_fmaf(a[0], b[0], c[0]), __builtin_fmaf(a[1], b[1], c[1]), __builtin_fmaf(a[2], b[2], c[2]), __builtin_fmaf(a[3], b[3], c[3])};
}

// ZH: the activation mask is the sign of this layer's own pre-activation h * scale + shift (the layer's output was never stored: it was
// consumed through the next layer's prologue); XL: see BlbArgs.  Pre-activations are formed with the SAME fused multiply-add in the forward
// prologue (bn_linear_fwd_kernel), here and in the mask, so that a value within rounding of zero takes the same branch everywhere.
// rows per block: 32, or 16 (RB) for 128 outputs and for small maps -- at 10 240 rows 32-row blocks leave 40 workgroups on 256 CUs
inline int blb_rows_per_block(int64_t rows, int n) { return (n == 128 || rows <= 65536) ? 16 : 32; }

template <int NT, int KT, bool BN, bool DEEP, bool ZH, bool XL, int RB>
__global__ __launch_bounds__(256) void bn_linear_bwd_kernel(const BlbArgs a) {
  static_assert(BN || !(ZH || XL), "lazy operands belong to BatchNorm layers");
  // 128 outputs: 16-row blocks -- the LDS tiles and the staging registers halve, so that three workgroups share a CU (with 32 rows: 84 KB
  // of LDS and 220 registers = ONE workgroup per CU, nothing to hide its load latency behind)
  constexpr int N = 32 * NT, K = 32 * KT, DS = N + 36, XS = K + 36, R = RB, RH = R / 16;
  constexpr int TPW = NT * KT / 4;               // weight-gradient tiles per wave
  constexpr int KQ = K / 64;                     // data gradient: 16-channel tiles per wave
  constexpr int NS = N / 4;                      // data gradient: contraction steps
  constexpr int NLD = NT * R / 32, NLX = KT * R / 32;      // float4 per thread and block: R rows x N / 4 = 256 NLD
  extern __shared__ __attribute__((aligned(16))) float blb_smem[];
  float* Dl = blb_smem;                          // [2][R][DS]  dh
  float* Xl = blb_smem + 2 * R * DS;             // [2][R][XS]  x
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31, g16 = lane >> 4, j16 = lane & 15;
  const bool want_dx = a.dx != nullptr;

  // staging coordinates: element e = tid + 256 i of a [32][W/4] block of float4 -> row e / (W/4), float4 column e % (W/4) (constant per thread)
  const int dc = tid % (N / 4), dr0 = tid / (N / 4);          // rows dr0 + (1024 / N) i
  const int xc = tid % (K / 4), xr0 = tid / (K / 4);
  constexpr int DRS = 1024 / N, XRS = 1024 / K;

  f32x4 mean, rstd, scale, shift, c1, c2;
  if (BN) {
    mean = *reinterpret_cast<const f32x4*>(a.stat + 4 * dc);
    rstd = *reinterpret_cast<const f32x4*>(a.stat + N + 4 * dc);
    scale = *reinterpret_cast<const f32x4*>(a.stat + 2 * N + 4 * dc);
    if (ZH) shift = *reinterpret_cast<const f32x4*>(a.stat + 3 * N + 4 * dc);
    c1 = *reinterpret_cast<const f32x4*>(a.coef + 4 * dc);
    c2 = *reinterpret_cast<const f32x4*>(a.coef + N + 4 * dc);
  }
  // data gradient: W as the A operand, wa[q][s] = W[4 s + g16][kbase + 16 q + j16]
  const int kbase = wave * (K / 4);
  float wa[KQ][NS];
  if (want_dx) {
#pragma unroll
    for (int q = 0; q < KQ; ++q)
#pragma unroll
      for (int s = 0; s < NS; ++s) wa[q][s] = a.w[(int64_t)(4 * s + g16) * a.ldw + kbase + 16 * q + j16];
  }

  f32x16 acc[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum[TPW];                               // column sums of dh over this lane's rows of channel 32 nt + l31 (kept for the kt == 0 tiles)
#pragma unroll
  for (int t = 0; t < TPW; ++t) bsum[t] = 0.f;
  // XL: the previous layer's affine for the weight-gradient operand (lane = channel 32 kt + l31 of tile t) and, in the data gradient's
  // layout (channels kbase + 16 q + 4 g16 .. + 3), its affine and its normalisation; running sums of its BatchNorm backward
  float xs_w[TPW], xb_w[TPW];
  f32x4 xs_d[KQ], xb_d[KQ], xm_d[KQ], xr_d[KQ], s1[KQ], s2[KQ];
  if (XL) {
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int kt = (wave * TPW + t) % KT;
      xs_w[t] = a.xstat[2 * K + 32 * kt + l31];
      xb_w[t] = a.xstat[3 * K + 32 * kt + l31];
    }
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      const int c = kbase + 16 * q + 4 * g16;
      xm_d[q] = *reinterpret_cast<const f32x4*>(a.xstat + c);
      xr_d[q] = *reinterpret_cast<const f32x4*>(a.xstat + K + c);
      xs_d[q] = *reinterpret_cast<const f32x4*>(a.xstat + 2 * K + c);
      xb_d[q] = *reinterpret_cast<const f32x4*>(a.xstat + 3 * K + c);
      s1[q] = f32x4{0.f, 0.f, 0.f, 0.f};
      s2[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }

  struct Stage {
    f32x4 d[NLD], m[NLD], hv[NLD], x[NLX], r[RH * KQ];
  };
  const bool has_res = a.res != nullptr;

  auto load_block = [&](int64_t blk, Stage& s) {
    const int64_t r0 = blk * R;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int64_t row = r0 + dr0 + DRS * i;
      s.d[i] = *reinterpret_cast<const f32x4*>(a.dz + row * a.lddz + 4 * dc);
      if (!ZH) s.m[i] = *reinterpret_cast<const f32x4*>(a.z + row * a.ldz + 4 * dc);
      if (BN) s.hv[i] = *reinterpret_cast<const f32x4*>(a.h + row * a.ldh + 4 * dc);
    }
#pragma unroll
    for (int i = 0; i < NLX; ++i) {
      const int64_t row = r0 + xr0 + XRS * i;
      s.x[i] = *reinterpret_cast<const f32x4*>(a.x + row * a.ldx + 4 * xc);
    }
    if (want_dx && has_res) {
#pragma unroll
      for (int rh = 0; rh < RH; ++rh)
#pragma unroll
        for (int q = 0; q < KQ; ++q)
          s.r[rh * KQ + q] = *reinterpret_cast<const f32x4*>(a.res + (r0 + 16 * rh + j16) * a.ldres + kbase + 16 * q + 4 * g16);
    }
  };
  // (dz, z, h) -> dh into the LDS tile (and dzm to memory), x into its tile
  auto store_block = [&](int64_t blk, int buf, const Stage& s) {
    const int64_t r0 = blk * R;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int r = dr0 + DRS * i;
      f32x4 d = s.d[i];
      const f32x4 m = ZH ? blb_fma4(s.hv[i], scale, shift) : s.m[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) d[e] = m[e] > 0.f ? d[e] : d[e] * a.slope;
      if (a.dzm) *reinterpret_cast<f32x4*>(a.dzm + (r0 + r) * a.lddzm + 4 * dc) = d;
      if (BN) {
        const f32x4 xh = (s.hv[i] - mean) * rstd;
        d = scale * (d - c1 - xh * c2);
      }
      *reinterpret_cast<f32x4*>(Dl + (buf * R + r) * DS + 4 * dc) = d;
    }
#pragma unroll
    for (int i = 0; i < NLX; ++i) {
      const int r = xr0 + XRS * i;
      *reinterpret_cast<f32x4*>(Xl + (buf * R + r) * XS + 4 * xc) = s.x[i];
    }
  };

  auto multiply = [&](int64_t blk, int buf, const f32x4 (&rv)[RH * KQ]) {
    const float* dl = Dl + buf * R * DS;
    const float* xl = Xl + buf * R * XS;
    // weight gradient: step j contracts the rows rr = (j & 7) + 16 (j >> 3) and rr + 8 (lane halves)
#pragma unroll
    for (int j = 0; j < R / 2; ++j) {
      const int rr = (j & 7) + 16 * (j >> 3) + 8 * h;
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        const int tile = wave * TPW + t, nt = tile / KT, kt = tile % KT;
        float b = xl[rr * XS + 32 * kt + l31];
        if (XL) {
          b = __builtin_fmaf(b, xs_w[t], xb_w[t]);
          b = b > 0.f ? b : b * a.xslope;
        }
        const float av = dl[rr * DS + 32 * nt + l31];
        bsum[t] += av;
        acc[t] = cmr_mfma32(av, b, acc[t]);
      }
    }
    if (!want_dx) return;
    // data gradient of rows 16 rh + j16, input channels kbase + 16 q + 4 g16 .. + 3
    // (the row halves side by side: RH KQ independent accumulator chains instead of KQ)
    f32x4 dacc[RH][KQ];
#pragma unroll
    for (int rh = 0; rh < RH; ++rh)
#pragma unroll
      for (int q = 0; q < KQ; ++q) dacc[rh][q] = has_res ? rv[rh * KQ + q] : f32x4{0.f, 0.f, 0.f, 0.f};
    const float* drow = dl + j16 * DS + g16;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
      for (int rh = 0; rh < RH; ++rh) {
        const float b = drow[16 * rh * DS + 4 * s];
#pragma unroll
        for (int q = 0; q < KQ; ++q) dacc[rh][q] = m16_mfma(wa[q][s], b, dacc[rh][q]);
      }
    }
#pragma unroll
    for (int rh = 0; rh < RH; ++rh)
#pragma unroll
      for (int q = 0; q < KQ; ++q) {
        *reinterpret_cast<f32x4*>(a.dx + (blk * R + 16 * rh + j16) * a.lddx + kbase + 16 * q + 4 * g16) = dacc[rh][q];
        if (XL) {
          // dx is the gradient at the previous layer's (never stored) output: its BatchNorm-backward sums from the raw BatchNorm input in LDS
          const f32x4 raw = *reinterpret_cast<const f32x4*>(xl + (16 * rh + j16) * XS + kbase + 16 * q + 4 * g16);
          const f32x4 pre = blb_fma4(raw, xs_d[q], xb_d[q]);
          f32x4 dp = dacc[rh][q];
#pragma unroll
          for (int e = 0; e < 4; ++e) dp[e] = pre[e] > 0.f ? dp[e] : dp[e] * a.xslope;
          s1[q] += dp;
          s2[q] += dp * ((raw - xm_d[q]) * xr_d[q]);
        }
      }
  };

  // Workgroup -> row blocks.  The map is cut into segments of seg_blocks blocks (one segment = the whole map unless the caller wants
  // per-segment column sums of dh, e.g. per SAMPLE: CMRAgent.py:95-99 broadcasts a per-sample vector to every point, whose gradient is
  // that sum); seg_groups workgroups stride through each segment.
  const int64_t g = a.seg_groups;
  const int64_t seg = blockIdx.x / a.seg_groups;
  const int64_t nblocks = (seg + 1) * a.seg_blocks;            // (end of this workgroup's segment)
  auto clampb = [&](int64_t b) { return b < nblocks ? b : nblocks - 1; };
  int64_t blk = seg * a.seg_blocks + blockIdx.x % a.seg_groups;
  f32x4 rcur[RH * KQ];
  if (DEEP) {
    // two blocks ahead in registers (blk + g in sa, blk + 2 g in sb; blk itself in LDS), the body unrolled twice so that the two register
    // sets swap roles without moves
    Stage sa, sb;
    load_block(clampb(blk), sa);
    store_block(blk, 0, sa);
#pragma unroll
    for (int i = 0; i < RH * KQ; ++i) rcur[i] = sa.r[i];
    load_block(clampb(blk + g), sa);
    __syncthreads();
    for (; blk < nblocks; blk += 2 * g) {
      load_block(clampb(blk + 2 * g), sb);
      multiply(blk, 0, rcur);
      if (blk + g < nblocks) store_block(blk + g, 1, sa);        // (uniform)
#pragma unroll
      for (int i = 0; i < RH * KQ; ++i) rcur[i] = sa.r[i];
      __syncthreads();
      if (blk + g >= nblocks) break;
      load_block(clampb(blk + 3 * g), sa);
      multiply(blk + g, 1, rcur);
      if (blk + 2 * g < nblocks) store_block(blk + 2 * g, 0, sb);
#pragma unroll
      for (int i = 0; i < RH * KQ; ++i) rcur[i] = sb.r[i];
      __syncthreads();
    }
  } else {
    Stage st;
    load_block(clampb(blk), st);
    store_block(blk, 0, st);
#pragma unroll
    for (int i = 0; i < RH * KQ; ++i) rcur[i] = st.r[i];
    __syncthreads();
    int buf = 0;
    for (; blk < nblocks; blk += g) {
      const int64_t nb = blk + g;
      load_block(clampb(nb), st);                    // next block of this workgroup: in flight under the MFMAs
      multiply(blk, buf, rcur);
      if (nb < nblocks) store_block(nb, buf ^ 1, st);      // (uniform) the other buffer: last read one iteration ago, behind a barrier
#pragma unroll
      for (int i = 0; i < RH * KQ; ++i) rcur[i] = st.r[i];
      __syncthreads();
      buf ^= 1;
    }
  }
  float* out = a.part + (int64_t)blockIdx.x * N * K;
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int tile = wave * TPW + t, nt = tile / KT, kt = tile % KT;
#pragma unroll
    for (int r = 0; r < 16; ++r) out[(int64_t)(nt * 32 + cmr_mfma_row(r, lane)) * K + kt * 32 + l31] = acc[t][r];
    if (a.part_b) {
      const float bs = bsum[t] + cmr_xhalf(bsum[t]);          // the two lane halves held the rows r and r + 8 of every step
      if (kt == 0 && h == 0) a.part_b[(int64_t)blockIdx.x * N + nt * 32 + l31] = bs;
    }
  }
  if (XL) {
    // the 16 row lanes of a lane group hold the same channels: fixed-order DPP sum, lane 16 g writes (bn_bwd_partial_kernel's layout)
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s1[q][e] = m16_sum16(s1[q][e]);
        s2[q][e] = m16_sum16(s2[q][e]);
      }
      if (j16 == 0) {
        float* p = a.xpart + (int64_t)blockIdx.x * 2 * K + kbase + 16 * q + 4 * g16;
        *reinterpret_cast<f32x4*>(p) = s1[q];
        *reinterpret_cast<f32x4*>(p + K) = s2[q];
      }
    }
  }
}

// coef[0][c] = sum d / rows, coef[1][c] = sum d xhat / rows, dgamma = sum d xhat, dbeta = sum d: one wave per channel over the workgroups'
// partials, in double (train.hip: bn_bwd_final_kernel's arithmetic)
__global__ __launch_bounds__(64) void blb_coef_final_kernel(const float* __restrict__ part, int nblk, int64_t rows, int C, float* __restrict__ coef,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.x, lane = threadIdx.x;
  double s = 0.0, sx = 0.0;
  for (int b = lane; b < nblk; b += 64) {
    s += (double)part[(int64_t)b * 2 * C + c];
    sx += (double)part[(int64_t)b * 2 * C + C + c];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o, 64);
    sx += __shfl_xor(sx, o, 64);
  }
  if (lane != 0) return;
  coef[c] = (float)(s / (double)rows);
  coef[C + c] = (float)(sx / (double)rows);
  if (dbeta) dbeta[c] = (float)s;
  if (dgamma) dgamma[c] = (float)sx;
}

// dw[i] (+)= sum over the workgroups' partials, in double, fixed order (32 slice groups x 8 loads in flight)
constexpr int BR_OUT = 32, BR_GRP = 32, BR_U = 8;
// outputs [0, n k): dw entries; [n k, n k + n): db entries (when part_b)
__global__ __launch_bounds__(BR_OUT * BR_GRP) void blb_reduce_kernel(const float* __restrict__ part, const float* __restrict__ part_b, int nslices,
                                                                     int n, int k, float* __restrict__ dw, int64_t lddw, int accumulate,
                                                                     float* __restrict__ db, int accumulate_db) {
  __shared__ double sm[BR_GRP][BR_OUT];
  const int o = threadIdx.x % BR_OUT, gq = threadIdx.x / BR_OUT;
  const int64_t nk = (int64_t)n * k, total = nk + (part_b ? n : 0);
  const int64_t i = (int64_t)blockIdx.x * BR_OUT + o;
  const float* p = i < nk ? part + i : (i < total ? part_b + (i - nk) : part);
  const int64_t stride = i < nk ? nk : n;
  double s = 0.0;
  for (int j0 = gq; j0 < nslices; j0 += BR_GRP * BR_U) {
    float v[BR_U];
#pragma unroll
    for (int u = 0; u < BR_U; ++u) {
      const int j = j0 + u * BR_GRP;
      v[u] = p[(int64_t)(j < nslices ? j : j0) * stride];
    }
#pragma unroll
    for (int u = 0; u < BR_U; ++u) s += j0 + u * BR_GRP < nslices ? (double)v[u] : 0.0;
  }
  sm[gq][o] = s;
  __syncthreads();
  if (gq == 0 && i < total) {
#pragma unroll
    for (int j = 1; j < BR_GRP; ++j) s += sm[j][o];
    if (i < nk) {
      float* d = dw + (i / k) * lddw + (i % k);
      *d = accumulate ? *d + (float)s : (float)s;
    } else {
      db[i - nk] = accumulate_db ? db[i - nk] + (float)s : (float)s;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Forward of the same layer: h = x W^T + b for 64 output channels WITH the BatchNorm statistics of h from the same pass (the separate
// cmr_bn_stats_f32 sweep read h again), and optionally with the PREVIOUS layer's BatchNorm + LeakyReLU applied to x on the way in (PRO: x
// is then the previous layer's BatchNorm input, its activated output is never written: PointNN.py:96-123 layer_1 -> layer_2 -> layer_3,
// :260-282 net[0] -> net[3]).  Same skeleton as the backward: 32-row blocks staged through LDS one block ahead; wave w owns the output
// channels [16 w, 16 w + 16): W in registers as the A operand of v_mfma_f32_16x16x4_f32, x rows as the B operand; a lane owns 4 channels
// of one row -> the per-channel sums are per-lane running sums, reduced over the 16 row lanes once at the end.  Sums are taken relative to
// a per-workgroup pivot (the workgroup's first row) and merged in double with the parallel-variance formula (bn_stats_merge_kernel).
// ------------------------------------------------------------------------------------------------------------------
struct BlfArgs {
  const float* x; int64_t ldx;        // [rows][k]
  const float* pro;                   // PRO: the previous layer's stat [4][k] (scale at 2 k, shift at 3 k)
  float pro_slope;
  const float* w; int64_t ldw;        // [n][k]
  const float* bias; int64_t bias_stride;   // [n] or null; bias_stride != 0: one bias row PER SEGMENT (bias + segment * bias_stride)
  float* h; int64_t ldh;              // [rows][n]
  float* part;                        // [gridDim.x][3][n]: pivot, sum (h - pivot), sum (h - pivot)^2
  int64_t rows;
  int64_t seg_blocks;                 // row blocks per segment (rows / 32 when the map is one segment)
  int seg_groups;                     // workgroups per segment
};

// NQ = n / 64: 16-channel tiles per wave (wave w owns the output channels [16 NQ w, 16 NQ (w + 1)))
template <int NQ, int KT, bool PRO>
__global__ __launch_bounds__(256) void bn_linear_fwd_kernel(const BlfArgs a) {
  constexpr int N = 64 * NQ, K = 32 * KT, XS = K + 36, R = 32, NS = K / 4, NLX = KT;
  extern __shared__ __attribute__((aligned(16))) float blf_smem[];
  float* Xl = blf_smem;                          // [2][R][XS]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g16 = lane >> 4, j16 = lane & 15;
  const int xc = tid % (K / 4), xr0 = tid / (K / 4);
  constexpr int XRS = 1024 / K;
  f32x4 pscale, pshift;
  if (PRO) {
    pscale = *reinterpret_cast<const f32x4*>(a.pro + 2 * K + 4 * xc);
    pshift = *reinterpret_cast<const f32x4*>(a.pro + 3 * K + 4 * xc);
  }
  // workgroup -> row blocks: seg_groups workgroups stride through each segment of seg_blocks blocks (one segment = the whole map unless the
  // bias is per segment, e.g. per SAMPLE: the broadcast half of cat([feat, max]) folded into the bias, CMRAgent.py:95-99)
  const int64_t g = a.seg_groups;
  const int64_t seg = blockIdx.x / a.seg_groups;
  const int64_t nblocks = (seg + 1) * a.seg_blocks;
  const int cbase = 16 * NQ * wave;
  float wa[NQ][NS];                              // A operand: lane 16 g + i holds W[cbase + 16 q + i][4 s + g]
  f32x4 bias4[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
#pragma unroll
    for (int s = 0; s < NS; ++s) wa[q][s] = a.w[(int64_t)(cbase + 16 * q + j16) * a.ldw + 4 * s + g16];
    bias4[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (a.bias) bias4[q] = *reinterpret_cast<const f32x4*>(a.bias + seg * a.bias_stride + cbase + 16 * q + 4 * g16);
  }

  auto load_block = [&](int64_t blk, f32x4 (&xv)[NLX]) {
#pragma unroll
    for (int i = 0; i < NLX; ++i) xv[i] = *reinterpret_cast<const f32x4*>(a.x + (blk * R + xr0 + XRS * i) * a.ldx + 4 * xc);
  };
  auto store_block = [&](int buf, const f32x4 (&xv)[NLX]) {
#pragma unroll
    for (int i = 0; i < NLX; ++i) {
      f32x4 v = xv[i];
      if (PRO) {
        v = f32x4{__builtin_fmaf(v[0], pscale[0], pshift[0]), __builtin_fmaf(v[1], pscale[1], pshift[1]), __builtin_fmaf(v[2], pscale[2], pshift[2]),
                  __builtin_fmaf(v[3], pscale[3], pshift[3])};          // (the same fused multiply-add as the backward's recomputation)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * a.pro_slope;
      }
      *reinterpret_cast<f32x4*>(Xl + (buf * R + xr0 + XRS * i) * XS + 4 * xc) = v;
    }
  };
  f32x4 pivot[NQ], sum[NQ], sq[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) pivot[q] = sum[q] = sq[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  bool first = true;
  auto multiply = [&](int64_t blk, int buf) {
    const float* xl = Xl + buf * R * XS;
    f32x4 acc[2][NQ];                              // the two row halves side by side: 2 NQ independent accumulator chains
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[0][q] = acc[1][q] = bias4[q];
    const float* xrow = xl + j16 * XS + g16;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const float b0 = xrow[4 * s], b1 = xrow[16 * XS + 4 * s];
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        acc[0][q] = m16_mfma(wa[q][s], b0, acc[0][q]);
        acc[1][q] = m16_mfma(wa[q][s], b1, acc[1][q]);
      }
    }
    // pivot = this workgroup's first row (lane 16 g of each lane group holds it).  Branch-free on purpose: with `if (first)` here hipcc put
    // the branch between the last MFMA and the v_accvgpr_read of its result, and on the taken path (every block but the first) only an
    // s_nop 0 separated the two -- acc[1][NQ-1][3] was read before the matrix core had written it (seen as wrong statistics of exactly the
    // channels 16 (2 w + 1) + 4 g + 3 at n = 128, with every h correct)
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float p0 = __shfl(acc[0][q][e], lane & 48, 64);
        pivot[q][e] = first ? p0 : pivot[q][e];
      }
    first = false;
#pragma unroll
    for (int rh = 0; rh < 2; ++rh)
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        *reinterpret_cast<f32x4*>(a.h + (blk * R + 16 * rh + j16) * a.ldh + cbase + 16 * q + 4 * g16) = acc[rh][q];
        const f32x4 d = acc[rh][q] - pivot[q];
        sum[q] += d;
        sq[q] += d * d;
      }
  };
  auto clampb = [&](int64_t b) { return b < nblocks ? b : nblocks - 1; };
  int64_t blk = seg * a.seg_blocks + blockIdx.x % a.seg_groups;
  // Two blocks ahead in registers (blk + g in xa, blk + 2 g in xb; blk itself in LDS): 8 - 16 KB per block and workgroup, so with ONE block
  // in flight a CU has 32 KB outstanding -- 8 MB on the chip against the ~16 MB that 8 TB/s x 2 us of loaded latency asks for.  (Measured
  // the same 71 - 73 us either way in back-to-back launches; 57 us = 4.7 TB/s alone.)  The body is unrolled twice so that the two register
  // sets swap roles without moves.
  f32x4 xa[NLX], xb[NLX];
  load_block(clampb(blk), xa);
  store_block(0, xa);
  load_block(clampb(blk + g), xa);
  __syncthreads();
  for (; blk < nblocks; blk += 2 * g) {
    load_block(clampb(blk + 2 * g), xb);
    multiply(blk, 0);
    store_block(1, xa);
    __syncthreads();
    if (blk + g >= nblocks) break;                 // (uniform)
    load_block(clampb(blk + 3 * g), xa);
    multiply(blk + g, 1);
    store_block(0, xb);
    __syncthreads();
  }
  // reduce over the 16 row lanes (DPP, fixed order); lane 16 g writes channels cbase + 16 q + 4 g .. + 3
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      sum[q][e] = m16_sum16(sum[q][e]);
      sq[q][e] = m16_sum16(sq[q][e]);
    }
    if (j16 == 0) {
      float* p = a.part + (int64_t)blockIdx.x * 3 * N + cbase + 16 * q + 4 * g16;
      *reinterpret_cast<f32x4*>(p) = pivot[q];
      *reinterpret_cast<f32x4*>(p + N) = sum[q];
      *reinterpret_cast<f32x4*>(p + 2 * N) = sq[q];
    }
  }
}

__device__ __forceinline__ double blf_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// one wave per channel: merge the workgroups' (pivot, sum, sum of squares) over their row counts in double (parallel variance), then what
// cmr_bn_stats_f32's final step does: stat = (mean, rstd, scale, shift), running statistics.  Workgroup w = (segment, i) of seg_groups per
// segment held the blocks i, i + seg_groups, ... of its segment's seg_blocks.
__global__ __launch_bounds__(64) void bn_stats_merge_kernel(const float* __restrict__ part, int G, int seg_groups, int64_t seg_blocks, int64_t rows,
                                                            int C, float eps, float momentum, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ running_mean,
                                                            float* __restrict__ running_var, float* __restrict__ stat) {
  const int c = blockIdx.x, lane = threadIdx.x;
  // ONE pass over the partials: with d_w = mean - p_w,  M2 = sum_w [ss_w - 2 d_w s_w + n_w d_w^2]
  //   = sum ss - 2 mean sum s + 2 sum p s + mean^2 sum n - 2 mean sum n p + sum n p^2   (double: the terms are <= rows * mean^2 ~ 1e8, M2 ~ 1e6)
  double s_s = 0.0, s_ss = 0.0, s_ps = 0.0, s_np = 0.0, s_npp = 0.0;
  for (int w = lane; w < G; w += 64) {
    const double n = 32.0 * (double)((seg_blocks - w % seg_groups + seg_groups - 1) / seg_groups);
    const double p = (double)part[(int64_t)w * 3 * C + c], s = (double)part[(int64_t)w * 3 * C + C + c];
    s_s += s;
    s_ss += (double)part[(int64_t)w * 3 * C + 2 * C + c];
    s_ps += p * s;
    s_np += n * p;
    s_npp += n * p * p;
  }
  s_s = blf_wave_sum(s_s);
  s_ss = blf_wave_sum(s_ss);
  s_ps = blf_wave_sum(s_ps);
  s_np = blf_wave_sum(s_np);
  s_npp = blf_wave_sum(s_npp);
  if (lane != 0) return;
  const double n = (double)rows;
  const double mean = (s_np + s_s) / n;
  const double m2 = s_ss - 2.0 * mean * s_s + 2.0 * s_ps + mean * mean * n - 2.0 * mean * s_np + s_npp;
  double var = m2 / n;
  var = var > 0.0 ? var : 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float gm = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
  const float scale = gm * rstd;
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

inline int blf_groups(int64_t rows, int n, int k) {
  const int64_t nblocks = rows / 32;
  int64_t groups = 256 * ((k == 64 && n == 64) ? 4 : (n == 64 || k == 64 ? 3 : 2));
  if (groups > nblocks / 8) groups = nblocks / 8 > 0 ? nblocks / 8 : 1;      // >= 8 row blocks per workgroup: the W fragments and the pipeline fill are paid once
  return (int)groups;
}

template <int NQ, int KT, bool PRO>
int blf_launch(const BlfArgs& a, int groups, hipStream_t stream) {
  const size_t smem = (size_t)2 * 32 * (32 * KT + 36) * sizeof(float);
  hipLaunchKernelGGL((bn_linear_fwd_kernel<NQ, KT, PRO>), dim3(groups), dim3(256), smem, stream, a);
  return CMR_OK;
}

// cs[seg][c] = sum over the segment's workgroups of part_b[.][c] (double, fixed order): per-segment column sums of dh
__global__ __launch_bounds__(64) void blb_seg_reduce_kernel(const float* __restrict__ part_b, int seg_groups, int n, float* __restrict__ cs) {
  const int seg = blockIdx.x, c = blockIdx.y * 64 + threadIdx.x;
  if (c >= n) return;
  const float* p = part_b + (int64_t)seg * seg_groups * n + c;
  double s = 0.0;
  int w = 0;
  for (; w + 8 <= seg_groups; w += 8) {          // eight independent loads in flight (one dependent load per workgroup took 18 us for 76)
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(w + u) * n];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += (double)v[u];
  }
  for (; w < seg_groups; ++w) s += (double)p[(int64_t)w * n];
  cs[(int64_t)seg * n + c] = (float)s;
}

inline int blb_groups(int64_t rows, int n, int k) {
  const int R = blb_rows_per_block(rows, n);
  const int64_t nblocks = rows / R;
  const size_t smem = (size_t)2 * R * (n + k + 72) * sizeof(float);
  int per_cu = (int)((size_t)160 * 1024 / smem);
  if (per_cu > 3) per_cu = 3;
  if (n == 128 && per_cu > 2) per_cu = 2;        // (registers allow two; every workgroup also costs a 4 n k byte partial that the reduction reads back)
  if (per_cu < 1) per_cu = 1;
  int64_t groups = 256 * per_cu;
  if (groups > nblocks / 8) groups = nblocks / 8 > 0 ? nblocks / 8 : 1;      // >= 8 row blocks per workgroup
  return (int)groups;
}

#ifndef CMR_BLB_DEEP
#define CMR_BLB_DEEP 0        // two blocks ahead in registers: measured slower at 64 x 64 (174 VGPRs = 2 workgroups per CU: 147 us against 135)
#endif
template <int NT, int KT, bool BN, bool ZH, bool XL, int RB>
int blb_launch_r(const BlbArgs& a, int groups, hipStream_t stream) {
  constexpr bool DEEP = CMR_BLB_DEEP != 0 && NT == 2 && KT == 2 && !XL && RB == 32;
  const size_t smem = (size_t)2 * RB * (32 * NT + 32 * KT + 72) * sizeof(float);
  static CmrSmemCache granted{};
  if (cmr_grant_smem(reinterpret_cast<const void*>(bn_linear_bwd_kernel<NT, KT, BN, DEEP, ZH, XL, RB>), smem, granted) != CMR_OK) return CMR_ELAUNCH;
  hipLaunchKernelGGL((bn_linear_bwd_kernel<NT, KT, BN, DEEP, ZH, XL, RB>), dim3(groups), dim3(256), smem, stream, a);
  return CMR_OK;
}

template <int NT, int KT, bool BN, bool ZH, bool XL>
int blb_launch(const BlbArgs& a, int groups, hipStream_t stream) {
  if constexpr (NT == 4) {
    return blb_launch_r<NT, KT, BN, ZH, XL, 16>(a, groups, stream);
  } else {
    return blb_rows_per_block(a.rows, 32 * NT) == 16 ? blb_launch_r<NT, KT, BN, ZH, XL, 16>(a, groups, stream)
                                                     : blb_launch_r<NT, KT, BN, ZH, XL, 32>(a, groups, stream);
  }
}

template <int NT, int KT>
int blb_dispatch(const BlbArgs& a, bool bn, bool zh, bool xl, int groups, hipStream_t stream) {
  if (!bn) return blb_launch<NT, KT, false, false, false>(a, groups, stream);
  if (xl) {
    if constexpr (NT == 2) {                       // the lazily fed layers of the reference all have 64 outputs
      return zh ? blb_launch<NT, KT, true, true, true>(a, groups, stream) : blb_launch<NT, KT, true, false, true>(a, groups, stream);
    } else {
      return CMR_EUNSUPPORTED;
    }
  }
  return zh ? blb_launch<NT, KT, true, true, false>(a, groups, stream) : blb_launch<NT, KT, true, false, false>(a, groups, stream);
}

inline bool blb_shape_ok(int64_t rows, int n, int k) { return (n == 64 || n == 128) && (k == 64 || k == 128) && rows >= 32 && rows % 32 == 0; }

}  // namespace

// dW partials [groups][n][k] + (lazy x operand) the previous layer's BatchNorm-backward partials [groups][2][k]
extern "C" int64_t cmr_bn_linear_bwd_workspace_bytes(int64_t rows, int n, int k) {
  if (!blb_shape_ok(rows, n, k)) return 0;
  return (int64_t)(blb_groups(rows, n, k) + 256) * ((int64_t)n * k + 2 * k + n) * (int64_t)sizeof(float);       // (+ 256: a segmented launch rounds the grid)
}

// stat / coef null: no BatchNorm (dh = dz * act'(z)).  dx null: weight gradient only.  db non-null: (+)= the column sums of dh, the bias
// gradient of a layer without BatchNorm (in front of a BatchNorm it is identically zero).
// seg_db non-null: the map is seg_rows-row segments (samples) and seg_db [rows / seg_rows][n] receives the column sums of dh per segment (the
//   gradient of a per-sample vector broadcast to the segment's rows, CMRAgent.py:95-99).
// mask_from_h: the layer's output z was never stored (it was consumed through the next layer's prologue, cmr_linear_bn_fwd_f32): the
//   activation mask is the sign of h * stat[2] + stat[3]; z is ignored.
// xstat non-null: x is the PREVIOUS layer's BatchNorm input and this layer's operand is lrelu_{xslope}(x * xstat[2] + xstat[3]); dx is
//   then the gradient at that (never stored) activation, and the previous layer's BatchNorm-backward reduction comes out of the same
//   pass: xcoef [2][k] (what cmr_bn_bwd_coef_f32 would return for it), xdgamma, xdbeta [k] (written when non-null).
// Returns CMR_EUNSUPPORTED for shapes it does not serve (n, k in {64, 128}, rows a multiple of 32; xstat: n = 64): the caller composes
// cmr_bn_bwd_f32 / cmr_linear_wgrad_f32 / cmr_linear_f32.
extern "C" int cmr_bn_linear_bwd_f32(const float* dz, int64_t lddz, const float* z, int64_t ldz, float slope, const float* h, int64_t ldh,
                                     const float* stat, const float* coef, int mask_from_h, float* dzm, int64_t lddzm, const float* x,
                                     int64_t ldx, const float* xstat, float xslope, float* xcoef, float* xdgamma, float* xdbeta,
                                     const float* w, int64_t ldw, const float* res, int64_t ldres, float* dx, int64_t lddx, int64_t rows,
                                     int n, int k, float* dw, int64_t lddw, int accumulate, float* db, int accumulate_db,
                                     int64_t seg_rows, float* seg_db, void* ws, int64_t ws_bytes, hipStream_t stream) {
  CMR_REQUIRE(dz && x && w && dw && ws && rows > 0 && n > 0 && k > 0);
  if (!blb_shape_ok(rows, n, k) || (xstat && n != 64)) return CMR_EUNSUPPORTED;
  if (seg_db && (seg_rows < 128 || seg_rows % 32 || rows % seg_rows)) return CMR_EUNSUPPORTED;
  const bool bn = stat != nullptr, zh = mask_from_h != 0, xl = xstat != nullptr;
  CMR_REQUIRE((stat == nullptr) == (coef == nullptr) && (!bn || h) && (bn || !(zh || xl)));
  CMR_REQUIRE(lddz % 4 == 0 && ldx % 4 == 0 && cmr_aligned16(dz) && cmr_aligned16(x) && lddz >= n && ldx >= k && ldw >= k && lddw >= k);
  if (z && !zh) CMR_REQUIRE(ldz % 4 == 0 && cmr_aligned16(z) && ldz >= n);
  if (bn) CMR_REQUIRE(ldh % 4 == 0 && cmr_aligned16(h) && ldh >= n && cmr_aligned16(stat) && cmr_aligned16(coef));
  if (dzm) CMR_REQUIRE(lddzm % 4 == 0 && cmr_aligned16(dzm) && lddzm >= n);
  if (dx) CMR_REQUIRE(lddx % 4 == 0 && cmr_aligned16(dx) && lddx >= k);
  if (res) CMR_REQUIRE(dx && ldres % 4 == 0 && cmr_aligned16(res) && ldres >= k);
  if (xl) CMR_REQUIRE(dx && xcoef && cmr_aligned16(xstat));
  int groups = blb_groups(rows, n, k);
  const int R = blb_rows_per_block(rows, n);
  int64_t seg_blocks = rows / R;
  int seg_groups = groups;
  if (seg_db) {                                  // per-segment column sums: a whole number of workgroups per segment, >= 4 blocks each
    const int64_t nseg = rows / seg_rows;
    seg_blocks = seg_rows / R;
    int64_t gps = (groups + nseg - 1) / nseg;
    if (gps > seg_blocks / 4) gps = seg_blocks / 4;
    if (gps < 1) gps = 1;
    CMR_REQUIRE(nseg * gps <= groups + 256);
    seg_groups = (int)gps;
    groups = (int)(nseg * gps);
  }
  CMR_REQUIRE(ws_bytes >= (int64_t)groups * ((int64_t)n * k + 2 * k + n) * (int64_t)sizeof(float));
  float* xpart = (float*)ws + (int64_t)groups * n * k;
  float* part_b = (db || seg_db) ? xpart + (int64_t)groups * 2 * k : nullptr;
  // no activation: the mask operand is dz itself with slope 1 (d = dz either way; the second read of the line hits the cache)
  const bool no_act = !zh && !z;
  const BlbArgs a{dz, lddz, no_act ? dz : z, no_act ? lddz : ldz, no_act ? 1.f : slope, h, ldh, stat, coef, dzm, lddzm, x, ldx, w, ldw, res, ldres,
                  dx, lddx, (float*)ws, part_b, rows, xstat, xslope, xpart, seg_blocks, seg_groups};
  int rc;
  if (n == 64 && k == 64) rc = blb_dispatch<2, 2>(a, bn, zh, xl, groups, stream);
  else if (n == 64 && k == 128) rc = blb_dispatch<2, 4>(a, bn, zh, xl, groups, stream);
  else if (n == 128 && k == 64) rc = blb_dispatch<4, 2>(a, bn, zh, xl, groups, stream);
  else rc = blb_dispatch<4, 4>(a, bn, zh, xl, groups, stream);
  if (rc != CMR_OK) return rc;
  const int64_t outs = (int64_t)n * k + (db ? n : 0);
  hipLaunchKernelGGL(blb_reduce_kernel, dim3((unsigned)((outs + BR_OUT - 1) / BR_OUT)), dim3(BR_OUT * BR_GRP), 0, stream, (const float*)ws,
                     (const float*)(db ? part_b : nullptr), groups, n, k, dw, lddw, accumulate, db, accumulate_db);
  if (seg_db)
    hipLaunchKernelGGL(blb_seg_reduce_kernel, dim3((unsigned)(rows / seg_rows), (unsigned)((n + 63) / 64)), dim3(64), 0, stream, (const float*)part_b,
                       seg_groups, n, seg_db);
  if (xl) hipLaunchKernelGGL(blb_coef_final_kernel, dim3(k), dim3(64), 0, stream, (const float*)xpart, groups, rows, k, xcoef, xdgamma, xdbeta);
  return cmr_launch_status();
}

extern "C" int64_t cmr_linear_bn_fwd_workspace_bytes(int64_t rows, int n, int k) {
  if (!(k == 64 || k == 128) || !(n == 64 || n == 128) || rows < 32 || rows % 32) return 0;
  return (int64_t)(blf_groups(rows, n, k) + 256) * 3 * n * (int64_t)sizeof(float);
}

// h [rows][n] = x' W^T + bias with x' = x, or (pro_stat non-null) x' = lrelu_{pro_slope}(x * pro_stat[2] + pro_stat[3]): the previous
// layer's BatchNorm + LeakyReLU applied on the way in; and stat [4][n] = the batch statistics of h as cmr_bn_stats_f32 returns them
// (running statistics updated when given).  bias_seg_rows > 0: bias is [rows / bias_seg_rows][n] (row stride bias_stride), one row per
// segment of bias_seg_rows rows -- a per-SAMPLE vector, e.g. the broadcast half of cat([feat, max]) times its weights (CMRAgent.py:95-99).
// Serves n, k in {64, 128}, rows a multiple of 32; else CMR_EUNSUPPORTED.
extern "C" int cmr_linear_bn_fwd_f32(const float* x, int64_t ldx, int k, const float* pro_stat, float pro_slope, const float* w, int64_t ldw,
                                     const float* bias, int64_t bias_seg_rows, int64_t bias_stride, float* h, int64_t ldh, int64_t rows, int n,
                                     float eps, float momentum, const float* gamma, const float* beta, float* running_mean,
                                     float* running_var, float* stat, void* ws, int64_t ws_bytes, hipStream_t stream) {
  CMR_REQUIRE(x && w && h && stat && ws && rows > 0 && k > 0 && n > 0);
  if (!(k == 64 || k == 128) || !(n == 64 || n == 128) || rows < 32 || rows % 32) return CMR_EUNSUPPORTED;
  if (bias_seg_rows > 0 && (bias_seg_rows < 128 || bias_seg_rows % 32 || rows % bias_seg_rows)) return CMR_EUNSUPPORTED;
  CMR_REQUIRE(ldx % 4 == 0 && ldh % 4 == 0 && ldx >= k && ldh >= n && ldw >= k && cmr_aligned16(x) && cmr_aligned16(h));
  CMR_REQUIRE((!bias || cmr_aligned16(bias)) && (!pro_stat || cmr_aligned16(pro_stat)));
  CMR_REQUIRE(bias_seg_rows <= 0 || (bias && bias_stride >= n && bias_stride % 4 == 0));
  CMR_REQUIRE((running_mean == nullptr) == (running_var == nullptr));
  int groups = blf_groups(rows, n, k);
  int64_t seg_blocks = rows / 32;
  int seg_groups = groups;
  if (bias_seg_rows > 0) {
    const int64_t nseg = rows / bias_seg_rows;
    seg_blocks = bias_seg_rows / 32;
    int64_t gps = (groups + nseg - 1) / nseg;
    if (gps > seg_blocks / 4) gps = seg_blocks / 4;
    if (gps < 1) gps = 1;
    CMR_REQUIRE(nseg * gps <= groups + 256);
    seg_groups = (int)gps;
    groups = (int)(nseg * gps);
  }
  CMR_REQUIRE(ws_bytes >= (int64_t)groups * 3 * n * (int64_t)sizeof(float));
  const BlfArgs a{x, ldx, pro_stat, pro_slope, w, ldw, bias, bias_seg_rows > 0 ? bias_stride : 0, h, ldh, (float*)ws, rows, seg_blocks, seg_groups};
  if (n == 64 && k == 64) {
    if (pro_stat) blf_launch<1, 2, true>(a, groups, stream); else blf_launch<1, 2, false>(a, groups, stream);
  } else if (n == 64) {
    if (pro_stat) blf_launch<1, 4, true>(a, groups, stream); else blf_launch<1, 4, false>(a, groups, stream);
  } else if (k == 64) {
    if (pro_stat) blf_launch<2, 2, true>(a, groups, stream); else blf_launch<2, 2, false>(a, groups, stream);
  } else {
    if (pro_stat) blf_launch<2, 4, true>(a, groups, stream); else blf_launch<2, 4, false>(a, groups, stream);
  }
  hipLaunchKernelGGL(bn_stats_merge_kernel, dim3(n), dim3(64), 0, stream, (const float*)ws, groups, seg_groups, seg_blocks, rows, n, eps, momentum,
                     gamma, beta, running_mean, running_var, stat);
  return cmr_launch_status();
}
