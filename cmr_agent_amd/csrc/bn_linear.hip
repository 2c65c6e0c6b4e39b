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

// ---- helpers of the bf16-product variants further down ----
typedef __bf16 bl_bf16x8 __attribute__((ext_vector_type(8)));
typedef short bl_s4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned bl_pack2(float lo, float hi) {            // (bf16(lo) | bf16(hi) << 16), round to nearest even
  typedef float f2 __attribute__((ext_vector_type(2)));
  typedef __bf16 b2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f2){lo, hi}, b2));
}
__device__ __forceinline__ int bl_off(int row, int chunk) { return 256 * row + 16 * (chunk ^ (((row & 3) << 2) | ((row >> 2) & 3))); }
__device__ __forceinline__ uint2 bl_tr(const unsigned char* base, int byte_off) {
  const bl_s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bl_s4*)(base + byte_off));
  return __builtin_bit_cast(uint2, v);
}
__device__ __forceinline__ bl_bf16x8 bl_tr8(const unsigned char* base, int off0, int off1) {
  const uint2 a0 = bl_tr(base, off0), a1 = bl_tr(base, off1);
  return __builtin_bit_cast(bl_bf16x8, uint4{a0.x, a0.y, a1.x, a1.y});
}
// sum over the 32 lanes that share lane >> 5 (the 32 rows of a block in the transposed product's layout): DPP row sum, then the partner row
__device__ __forceinline__ float bl_sum32(float v) {
  v = m16_sum16(v);
  v += cmr_xor16(v);
  asm volatile("" : "+v"(v));
  return v;
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

// ------------------------------------------------------------------------------------------------------------------
// bf16 products, forward (round 6): h = x' W^T + b with the batch statistics of h from the same pass, products on
// v_mfma_f32_32x32x16_bf16.  With 1/16 of the matrix time the layer is a stream, and the stream needs no workgroup: a WAVE owns 32-row
// tiles (lane = row, the pattern of linear_rows_bf16_kernel: 4.9 - 5.6 TB/s, profiles/r03_stream_patterns.txt), no barrier in the loop.
// Transposed product D[channel][row]: weights = A operand (bf16 fragments in LDS, written once per workgroup, fragment order: conflict-free
// ds_read_b128), rows = B operand (converted on the fly, PRO: the previous layer's BatchNorm + LeakyReLU applied first, its affine in
// LDS); a lane ends with 2 x 16 channels of ONE row -> float4 stores, per-lane running sums of (h - pivot) and (h - pivot)^2 (pivot = the
// wave's first row), reduced over the 32 row lanes once at the end.  A workgroup serves 64 output channels (blockIdx.y: the other 64 of a
// 128-wide layer; its x reads are L2 hits); partials [waves][4][n] = (pivot, sum, sum of squares, row count) merged in double.
// The bias (per segment when bias_stride != 0) sits in a per-wave LDS slot and is the accumulators' initial value.
// ------------------------------------------------------------------------------------------------------------------
template <int KS, bool PRO>
__global__ __launch_bounds__(256, 2) void bn_linear_fwd_bf16_kernel(const BlfArgs a, int n_total) {
  constexpr int K = 16 * KS;
  extern __shared__ __attribute__((aligned(16))) unsigned char blg_smem[];
  bl_bf16x8* Wf = reinterpret_cast<bl_bf16x8*>(blg_smem);                  // [2][KS][64 lanes]
  float* paff = reinterpret_cast<float*>(blg_smem + 2 * KS * 1024);        // PRO: [2][K] scale | shift
  float* bslot = paff + 2 * K;                                             // [4 waves][64]  bias
  float* pslot = bslot + 256;                                              // [4 waves][64]  pivot
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  const int cb = 64 * blockIdx.y;
  // weight fragments: lane (cout cb + 32 t + l31, half h), step ks holds W[cout][16 ks + 8 h .. + 7]
  for (int f = wave; f < 2 * KS; f += 4) {
    const int t = f / KS, ks = f % KS;
    const float* wr = a.w + (int64_t)(cb + 32 * t + l31) * a.ldw + 16 * ks + 8 * h;
    const f32x4 lo = *reinterpret_cast<const f32x4*>(wr), hi = *reinterpret_cast<const f32x4*>(wr + 4);
    Wf[f * 64 + lane] = __builtin_bit_cast(bl_bf16x8, uint4{bl_pack2(lo[0], lo[1]), bl_pack2(lo[2], lo[3]), bl_pack2(hi[0], hi[1]), bl_pack2(hi[2], hi[3])});
  }
  if (PRO) {
    for (int i = tid; i < 2 * K; i += 256) paff[i] = a.pro[2 * K + i];
  }
  float* myb = bslot + 64 * wave;
  float* myp = pslot + 64 * wave;
  myb[lane] = 0.f;
  myp[lane] = 0.f;
  __syncthreads();
  const int64_t ntiles = a.rows / 32, tstride = (int64_t)gridDim.x * 4;
  auto load_tile = [&](int64_t tile, f32x4 (&dst)[2 * KS]) __attribute__((always_inline)) {
    const int64_t row = (tile < ntiles ? tile : ntiles - 1) * 32 + l31;
    const float* xp = a.x + row * a.ldx + 8 * h;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      dst[2 * ks] = *reinterpret_cast<const f32x4*>(xp + 16 * ks);
      dst[2 * ks + 1] = *reinterpret_cast<const f32x4*>(xp + 16 * ks + 4);
    }
  };
  f32x4 raw[2 * KS];
  float sum[2][16], sq[2][16];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) sum[t][r] = sq[t][r] = 0.f;
  bool first = true;
  int64_t cur_seg = -1, count = 0;
  int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  load_tile(tile, raw);
#pragma clang loop unroll(disable)
  for (; tile < ntiles; tile += tstride) {
    // the LDS operands (affine, weight fragments) are loop invariant: without an opaque offset hipcc hoists all of them out of the tile
    // loop -- 128 + 64 registers of invariants, i.e. spills
    int opq = 0;
    asm volatile("" : "+v"(opq));
    const float* paff_ = paff + opq;
    const bl_bf16x8* Wf_ = Wf + opq;
    if (a.bias) {
      const int64_t seg = a.bias_stride ? tile / a.seg_blocks : 0;          // (wave-uniform)
      if (seg != cur_seg) {
        cur_seg = seg;
        myb[lane] = a.bias[seg * a.bias_stride + cb + lane];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 bq = *reinterpret_cast<const f32x4*>(myb + 32 * t + 8 * q + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[t][4 * q + e] = bq[e];
      }
    // one contraction step at a time: convert this step's 8 channels of the row (PRO: the previous layer's BatchNorm + LeakyReLU first), two
    // matrix instructions; the raw registers are dead behind the last step and take the NEXT tile's rows, which travel under the epilogue
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      f32x4 lo = raw[2 * ks], hi = raw[2 * ks + 1];
      if (PRO) {
        const int c = 16 * ks + 8 * h;
        lo = blb_fma4(lo, *reinterpret_cast<const f32x4*>(paff_ + c), *reinterpret_cast<const f32x4*>(paff_ + K + c));
        hi = blb_fma4(hi, *reinterpret_cast<const f32x4*>(paff_ + c + 4), *reinterpret_cast<const f32x4*>(paff_ + K + c + 4));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          lo[e] = lo[e] > 0.f ? lo[e] : lo[e] * a.pro_slope;
          hi[e] = hi[e] > 0.f ? hi[e] : hi[e] * a.pro_slope;
        }
      }
      const bl_bf16x8 xb = __builtin_bit_cast(bl_bf16x8, uint4{bl_pack2(lo[0], lo[1]), bl_pack2(lo[2], lo[3]), bl_pack2(hi[0], hi[1]), bl_pack2(hi[2], hi[3])});
#pragma unroll
      for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wf_[(t * KS + ks) * 64 + lane], xb, acc[t], 0, 0, 0);
      if (ks & 1) __builtin_amdgcn_sched_barrier(0);              // (keeps the steps in order: hoisting every operand read costs ~100 registers)
    }
    load_tile(tile + tstride, raw);
    __builtin_amdgcn_sched_barrier(0);
    float* hp = a.h + (tile * 32 + l31) * a.ldh + cb + 4 * h;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<f32x4*>(hp + 32 * t + 8 * q) = f32x4{acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]};
    // pivot = row 0 of the wave's first tile (lanes 0 / 32 hold it for their channels): kept in the wave's LDS slot, read back as float4
    if (first) {                                                 // (wave-uniform)
      if (l31 == 0) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *reinterpret_cast<f32x4*>(myp + 32 * t + 8 * q + 4 * h) = f32x4{acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]};
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      first = false;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 pv = *reinterpret_cast<const f32x4*>(myp + 32 * t + 8 * q + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = acc[t][4 * q + e] - pv[e];
          sum[t][4 * q + e] += d;
          sq[t][4 * q + e] = __builtin_fmaf(d, d, sq[t][4 * q + e]);
        }
      }
    count += 32;
  }
  // reduce over the 32 row lanes of each half (fixed order); lanes 0 and 32 write their channels cb + 32 t + 8 q + 4 h + e.  Partials are
  // stored [4][n][G] (G = waves of the launch): the merge reads a channel's G values of one kind from consecutive addresses
  const int64_t G = (int64_t)gridDim.x * 4, gw = (int64_t)blockIdx.x * 4 + wave;
  float* pp = a.part + (int64_t)(cb + 4 * h) * G + gw;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float s_ = bl_sum32(sum[t][r]), q_ = bl_sum32(sq[t][r]);
      if (l31 == 0) {
        const int c = 32 * t + 8 * (r >> 2) + (r & 3);
        pp[c * G] = myp[c + 4 * h];
        pp[((int64_t)n_total + c) * G] = s_;
        pp[((int64_t)2 * n_total + c) * G] = q_;
        pp[((int64_t)3 * n_total + c) * G] = (float)count;
      }
    }
}

// one 256-thread workgroup per channel: merge the waves' (pivot, sum, sum of squares, count) in double (parallel variance; every load of a
// thread independent of the others, consecutive threads on consecutive addresses), then what cmr_bn_stats_f32's final step does: stat =
// (mean, rstd, scale, shift), running statistics.  part [4][C][G].
__global__ __launch_bounds__(256) void bn_stats_merge_cnt_kernel(const float* __restrict__ part, int G, int64_t rows, int C, float eps, float momentum,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 float* __restrict__ running_mean, float* __restrict__ running_var,
                                                                 float* __restrict__ stat) {
  __shared__ double sm[4][5];
  const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* pv = part + (int64_t)c * G;
  const int64_t kind = (int64_t)C * G;
  double s_s = 0.0, s_ss = 0.0, s_ps = 0.0, s_np = 0.0, s_npp = 0.0;
  for (int w = tid; w < G; w += 256) {
    const double p = (double)pv[w], s = (double)pv[kind + w], n = (double)pv[3 * kind + w];
    s_s += s;
    s_ss += (double)pv[2 * kind + w];
    s_ps += p * s;
    s_np += n * p;
    s_npp += n * p * p;
  }
  s_s = blf_wave_sum(s_s);
  s_ss = blf_wave_sum(s_ss);
  s_ps = blf_wave_sum(s_ps);
  s_np = blf_wave_sum(s_np);
  s_npp = blf_wave_sum(s_npp);
  if (lane == 0) {
    sm[wave][0] = s_s; sm[wave][1] = s_ss; sm[wave][2] = s_ps; sm[wave][3] = s_np; sm[wave][4] = s_npp;
  }
  __syncthreads();
  if (tid != 0) return;
  s_s = sm[0][0] + sm[1][0] + sm[2][0] + sm[3][0];
  s_ss = sm[0][1] + sm[1][1] + sm[2][1] + sm[3][1];
  s_ps = sm[0][2] + sm[1][2] + sm[2][2] + sm[3][2];
  s_np = sm[0][3] + sm[1][3] + sm[2][3] + sm[3][3];
  s_npp = sm[0][4] + sm[1][4] + sm[2][4] + sm[3][4];
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

#ifndef CMR_BLG_WG_PER_CU
#define CMR_BLG_WG_PER_CU 2
#endif
#ifndef CMR_BLG_MIN_TILES
#define CMR_BLG_MIN_TILES 4
#endif
inline int blg_groups(int64_t rows) {
  const int64_t ntiles = rows / 32;
  int64_t groups = 256 * CMR_BLG_WG_PER_CU;
  if (groups * 4 * CMR_BLG_MIN_TILES > ntiles) groups = (ntiles + 4 * CMR_BLG_MIN_TILES - 1) / (4 * CMR_BLG_MIN_TILES);     // tiles per wave: the weight fragments and the pipeline fill are paid once
  return (int)(groups < 1 ? 1 : groups);
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

// ------------------------------------------------------------------------------------------------------------------
// bf16 products (round 6; VERDICT r05 #1a): the same layer backward for the bf16 mode of the agent update (BASELINE configs[2]:
// Train_Agent.py:296-305 through CMRAgent.py:25-33, 92-101).  At 18 FLOP/B the fp32 kernel above sits ON the fp32 matrix ridge (matrix
// pipe busy 0.47 of the launch, profiles/r04_pmc_bn_linear.txt); with the products on v_mfma_f32_32x32x16_bf16 (fp32 accumulate) the
// matrix work of a 32-row block is 4 - 12 instructions per wave and the layer is a stream.  Same skeleton (4 waves walk 32-row blocks with a
// static stride, whole rows fetched with float4 loads one block ahead, (dz, z, h) -> dh in fp32 on the way into LDS), but the LDS tiles
// are bf16 (RNE) in ONE image that serves both products: 256-byte rows, 16-byte chunk c of row r at chunk c ^ (((r & 3) << 2) | ((r >> 2) & 3))
// (cdna_hip_programming.md T10, image (b): conflict-free for the 8-byte staging writes, the ds_read_b128 row reads and the transposed reads)
//   weight gradient  dW[n][k] += sum_r dh[r][n] x'[r][k]: contraction over ROWS -- both operands are 8 consecutive rows of one channel per
//     lane = two ds_read_b64_tr_b16 each (the LDS hardware transposes on the way out); wave w owns the tiles w TPW .. + TPW - 1 (one nt:
//     the dh operand is shared);
//   data gradient    dx^T[k][r] = sum_n W^T[k][n] dh[r][n]: wave w < K / 32 owns input channels [32 w, 32 w + 32); W^T as bf16 fragments in
//     registers (A operand), dh rows by ds_read_b128 (B operand); a lane ends with 4 x 4 consecutive channels of ONE row: float4 stores,
//     the residual and (XL) the previous layer's raw BatchNorm input prefetched in the same pattern.
// Column sums of dh (bias gradient / per-segment sums) are taken by the staging threads from the UNROUNDED fp32 dh; everything that is not
// a product (masks, BatchNorm backward arithmetic, the XL sums, partial sums, reductions) stays fp32 / double as above.
// ------------------------------------------------------------------------------------------------------------------
constexpr int BLH_R = 32, BLH_IMG = BLH_R * 256;          // rows per block; bytes per image buffer
// dh enters both products as TWO bf16 terms, hi = bf16(dh) and lo = bf16(dh - hi) (16 mantissa bits instead of 8: two matrix instructions per
// product step instead of one, still 1/8 of the fp32 matrix time).  With single-term dh the 40-update trajectory of
// tests/test_train_gpu.py ended 10 % off the fp32 run (bar 5 %); x' and W stay single bf16 terms.
#ifndef CMR_BLH_HILO
#define CMR_BLH_HILO 1
#endif

template <int NT, int KT, bool BN, bool ZH, bool XL>
__global__ __launch_bounds__(256, (XL || NT == 4 || KT == 4) ? 2 : 3) void bn_linear_bwd_bf16_kernel(const BlbArgs a) {
  static_assert(BN || !(ZH || XL), "lazy operands belong to BatchNorm layers");
  constexpr int N = 32 * NT, K = 32 * KT, R = BLH_R, TPW = NT * KT / 4, NS = N / 16;
  constexpr int NLD = NT, NLX = KT, DRS = 1024 / N, XRS = 1024 / K;
  constexpr bool HL = CMR_BLH_HILO != 0;
  // the lo terms of dh: N = 64 -- in the unused half of dh's own 256-byte rows (chunk + 8, i.e. tile + 2 / step + 4); N = 128 -- an image of their own
  constexpr bool LOIMG = HL && N == 128;
  constexpr int NIMG = LOIMG ? 6 : 4, LO_T = (HL && !LOIMG) ? 2 : 0, LO_KS = (HL && !LOIMG) ? 4 : 0;
  extern __shared__ __attribute__((aligned(16))) unsigned char blh_smem[];
  unsigned char* DI = blh_smem;                                   // [2][32 rows][256 B]  dh  (bf16)
  unsigned char* XI = blh_smem + 2 * BLH_IMG;                     // [2][32 rows][256 B]  x'  (bf16)
  unsigned char* DLI = LOIMG ? blh_smem + 4 * BLH_IMG : DI;       // [2][32 rows][256 B]  lo terms of dh
  float* caff = reinterpret_cast<float*>(blh_smem + NIMG * BLH_IMG); // BN: [6][N] mean, rstd, scale, shift, c1, c2 (read per block: registers are scarcer)
  float* xaff = caff + 6 * N;                                     // XL: the previous layer's stat [4][K]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31, g16 = lane >> 4, q4 = (lane >> 2) & 3, p4 = lane & 3;
  const bool want_dx = a.dx != nullptr;
  const bool dg = want_dx && wave < KT;                           // (wave-uniform) this wave owns input channels [32 wave, 32 wave + 32)
  const bool has_res = a.res != nullptr;
  const bool want_b = a.part_b != nullptr;

  const int dc = tid % (N / 4), dr0 = tid / (N / 4);
  const int xc = tid % (K / 4), xr0 = tid / (K / 4);
  if (BN) {
    for (int i = tid; i < 4 * N; i += 256) caff[i] = a.stat[i];
    for (int i = tid; i < 2 * N; i += 256) caff[4 * N + i] = a.coef[i];
  }
  if (XL) {
    for (int i = tid; i < 4 * K; i += 256) xaff[i] = a.xstat[i];
  }
  if (BN || XL) __syncthreads();
  // data gradient: W^T as the A operand -- lane (channel 32 wave + l31, half h), step ks: W[16 ks + 8 h .. + 7][channel]
  bl_bf16x8 wa[NS];
  if (dg) {
    const float* wp = a.w + 32 * wave + l31;
#pragma unroll
    for (int ks = 0; ks < NS; ++ks) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = wp[(int64_t)(16 * ks + 8 * h + e) * a.ldw];
      wa[ks] = __builtin_bit_cast(bl_bf16x8, uint4{bl_pack2(v[0], v[1]), bl_pack2(v[2], v[3]), bl_pack2(v[4], v[5]), bl_pack2(v[6], v[7])});
    }
  }
  // transposed-read offsets of this lane (they do not change from block to block): 16-lane group g16 covers channels 16 (g16 & 1) .. + 15 of
  // a 32-channel tile and the row half h = g16 >> 1; lane 4 q + p of the group supplies the address of block row q, columns 4 p .. 4 p + 3.
  // With row = 16 s + 8 h + 4 j + q4 the swizzle of bl_off splits: chunk (4 T + cq) ^ (((row & 3) << 2) | ((row >> 2) & 3)) =
  // 4 (T ^ q4) + (cq ^ ((2 h + j) & 3)) -- four row bases per lane and one 64-byte tile term per tile instead of a register per (tile, s, j)
  const int nt_w = (wave * TPW) / KT;
  int trb[2][2], tqx[TPW];
  {
    const int sub8 = 8 * (p4 & 1), cq = 2 * (g16 & 1) + (p4 >> 1);
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
      for (int j = 0; j < 2; ++j) trb[s_][j] = 256 * (16 * s_ + 8 * h + 4 * j + q4) + 16 * (cq ^ ((2 * h + j) & 3)) + sub8;
#pragma unroll
    for (int t = 0; t < TPW; ++t) tqx[t] = 64 * (((wave * TPW + t) % KT) ^ q4);
  }
  const int tqd = 64 * (nt_w ^ q4), tql = 64 * ((nt_w + LO_T) ^ q4);
  // row reads of the data gradient's B operand: chunk (2 ks + h) of row l31 -> 256 l31 + 32 (ks ^ (sw >> 1)) + 16 (h ^ (sw & 1))
  const int swr = ((l31 & 3) << 2) | ((l31 >> 2) & 3);
  const int rdb = 256 * l31 + 16 * (h ^ (swr & 1)), rdx = swr >> 1;
  // N = 128: the W^T fragments live in LDS (fragment order, one ds_read_b128 per matrix instruction) -- 32 registers the kernel does not have
  constexpr bool WLDS = NS == 8;
  bl_bf16x8* wfr = reinterpret_cast<bl_bf16x8*>(blh_smem + NIMG * BLH_IMG + (6 * N + (XL ? 4 * K : 0)) * sizeof(float)) + wave * NS * 64;
  if (WLDS && dg) {
#pragma unroll
    for (int ks = 0; ks < NS; ++ks) wfr[ks * 64 + lane] = wa[ks];
  }

  f32x16 acc[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  f32x4 bs = {0.f, 0.f, 0.f, 0.f};                               // column sums of (fp32) dh over this thread's rows, channels 4 dc .. + 3
  f32x4 s1[4], s2[4];                                            // XL: the previous layer's BatchNorm-backward sums (channels 32 wave + 8 q + 4 h .. + 3)
#pragma unroll
  for (int q = 0; q < 4; ++q) s1[q] = s2[q] = f32x4{0.f, 0.f, 0.f, 0.f};

  struct Stage {
    f32x4 d[NLD], m[NLD], hv[NLD], x[NLX];
  };
  // the residual and (XL) the raw BatchNorm input of the previous layer in the data gradient's OUTPUT layout (row l31, channels 32 wave + 8 q
  // + 4 h .. + 3): requested at the top of a block's iteration, used behind its matrix instructions (the raw input was fetched by this
  // workgroup one iteration earlier for the image: an L2 hit)
  auto load_out_layout = [&](int64_t blk, f32x4 (&rv)[4], f32x4 (&xv)[4]) {
    if (!dg) return;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (has_res) rv[q] = *reinterpret_cast<const f32x4*>(a.res + (blk * R + l31) * a.ldres + 32 * wave + 8 * q + 4 * h);
      if (XL) xv[q] = *reinterpret_cast<const f32x4*>(a.x + (blk * R + l31) * a.ldx + 32 * wave + 8 * q + 4 * h);
    }
  };
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
    for (int i = 0; i < NLX; ++i) s.x[i] = *reinterpret_cast<const f32x4*>(a.x + (r0 + xr0 + XRS * i) * a.ldx + 4 * xc);
  };
  auto store_block = [&](int64_t blk, int buf, const Stage& s) {
    const int64_t r0 = blk * R;
    unsigned char* di = DI + buf * BLH_IMG;
    unsigned char* xi = XI + buf * BLH_IMG;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int r = dr0 + DRS * i;
      f32x4 d = s.d[i];
      const f32x4 scale = BN ? *reinterpret_cast<const f32x4*>(caff + 2 * N + 4 * dc) : f32x4{1.f, 1.f, 1.f, 1.f};
      const f32x4 m = ZH ? blb_fma4(s.hv[i], scale, *reinterpret_cast<const f32x4*>(caff + 3 * N + 4 * dc)) : s.m[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) d[e] = m[e] > 0.f ? d[e] : d[e] * a.slope;
      if (a.dzm) *reinterpret_cast<f32x4*>(a.dzm + (r0 + r) * a.lddzm + 4 * dc) = d;
      if (BN) {
        const f32x4 xh = (s.hv[i] - *reinterpret_cast<const f32x4*>(caff + 4 * dc)) * *reinterpret_cast<const f32x4*>(caff + N + 4 * dc);
        d = scale * (d - *reinterpret_cast<const f32x4*>(caff + 4 * N + 4 * dc) - xh * *reinterpret_cast<const f32x4*>(caff + 5 * N + 4 * dc));
      }
      if (want_b) bs += d;
      const unsigned w0 = bl_pack2(d[0], d[1]), w1 = bl_pack2(d[2], d[3]);
      *reinterpret_cast<uint2*>(di + bl_off(r, dc >> 1) + 8 * (dc & 1)) = uint2{w0, w1};
      if (HL) {                                              // residues dh - bf16(dh), rounded to bf16 again
        const float l0 = d[0] - __builtin_bit_cast(float, w0 << 16), l1 = d[1] - __builtin_bit_cast(float, w0 & 0xffff0000u);
        const float l2 = d[2] - __builtin_bit_cast(float, w1 << 16), l3 = d[3] - __builtin_bit_cast(float, w1 & 0xffff0000u);
        *reinterpret_cast<uint2*>(DLI + buf * BLH_IMG + bl_off(r, (dc >> 1) + 2 * LO_KS) + 8 * (dc & 1)) = uint2{bl_pack2(l0, l1), bl_pack2(l2, l3)};
      }
    }
#pragma unroll
    for (int i = 0; i < NLX; ++i) {
      const int r = xr0 + XRS * i;
      f32x4 v = s.x[i];
      if (XL) {
        v = blb_fma4(v, *reinterpret_cast<const f32x4*>(xaff + 2 * K + 4 * xc), *reinterpret_cast<const f32x4*>(xaff + 3 * K + 4 * xc));
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * a.xslope;
      }
      *reinterpret_cast<uint2*>(xi + bl_off(r, xc >> 1) + 8 * (xc & 1)) = uint2{bl_pack2(v[0], v[1]), bl_pack2(v[2], v[3])};
    }
  };
  auto multiply = [&](int64_t blk, int buf, const f32x4 (&rv)[4], const f32x4 (&xv)[4]) {
    const unsigned char* di = DI + buf * BLH_IMG;
    const unsigned char* xi = XI + buf * BLH_IMG;
    const unsigned char* dli = DLI + buf * BLH_IMG;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const bl_bf16x8 av = bl_tr8(di, trb[s][0] + tqd, trb[s][1] + tqd);
      bl_bf16x8 al;
      if (HL) al = bl_tr8(dli, trb[s][0] + tql, trb[s][1] + tql);
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        const bl_bf16x8 bv = bl_tr8(xi, trb[s][0] + tqx[t], trb[s][1] + tqx[t]);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[t], 0, 0, 0);
        if (HL) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bv, acc[t], 0, 0, 0);
      }
    }
    if (!dg) return;
    f32x16 dacc;
#pragma unroll
    for (int e = 0; e < 16; ++e) dacc[e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < NS; ++ks) {
      const bl_bf16x8 wv = WLDS ? wfr[ks * 64 + lane] : wa[ks];
      dacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wv, *reinterpret_cast<const bl_bf16x8*>(di + rdb + 32 * (ks ^ rdx)), dacc, 0, 0, 0);
      if (HL) dacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wv, *reinterpret_cast<const bl_bf16x8*>(dli + rdb + 32 * ((ks + LO_KS) ^ rdx)), dacc, 0, 0, 0);
    }
    float* dxp = a.dx + (blk * R + l31) * a.lddx + 32 * wave + 4 * h;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 o = {dacc[4 * q], dacc[4 * q + 1], dacc[4 * q + 2], dacc[4 * q + 3]};
      if (has_res) o += rv[q];
      *reinterpret_cast<f32x4*>(dxp + 8 * q) = o;
      if (XL) {
        // dx is the gradient at the previous layer's (never stored) output: its BatchNorm-backward sums from the raw BatchNorm input
        const int c = 32 * wave + 8 * q + 4 * h;
        const f32x4 raw = xv[q];
        const f32x4 pre = blb_fma4(raw, *reinterpret_cast<const f32x4*>(xaff + 2 * K + c), *reinterpret_cast<const f32x4*>(xaff + 3 * K + c));
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pre[e] > 0.f ? o[e] : o[e] * a.xslope;
        s1[q] += o;
        s2[q] += o * ((raw - *reinterpret_cast<const f32x4*>(xaff + c)) * *reinterpret_cast<const f32x4*>(xaff + K + c));
      }
    }
  };

  const int64_t g = a.seg_groups;
  const int64_t seg = blockIdx.x / a.seg_groups;
  const int64_t nblocks = (seg + 1) * a.seg_blocks;
  auto clampb = [&](int64_t b) { return b < nblocks ? b : nblocks - 1; };
  int64_t blk = seg * a.seg_blocks + blockIdx.x % a.seg_groups;
  {
    Stage st;
    load_block(clampb(blk), st);
    store_block(blk, 0, st);
    __syncthreads();
    int buf = 0;
    for (; blk < nblocks; blk += g) {
      const int64_t nb = blk + g;
      f32x4 rcur[4], xcur[4];
      load_out_layout(blk, rcur, xcur);
      load_block(clampb(nb), st);                    // next block of this workgroup: in flight under this block's products and stores
      multiply(blk, buf, rcur, xcur);
      if (nb < nblocks) store_block(nb, buf ^ 1, st);      // (uniform) the other buffer: last read one iteration ago, behind a barrier
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
  }
  if (XL && dg) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s1[q][e] = bl_sum32(s1[q][e]);
        s2[q][e] = bl_sum32(s2[q][e]);
      }
      if (l31 == 0) {
        float* p = a.xpart + (int64_t)blockIdx.x * 2 * K + 32 * wave + 8 * q + 4 * h;
        *reinterpret_cast<f32x4*>(p) = s1[q];
        *reinterpret_cast<f32x4*>(p + K) = s2[q];
      }
    }
  }
  if (want_b) {
    // the row groups of the staging layout meet in LDS (over the images: the loop's last barrier is behind every read of them), fixed order
    float* scr = reinterpret_cast<float*>(blh_smem);
    *reinterpret_cast<f32x4*>(scr + dr0 * N + 4 * dc) = bs;
    __syncthreads();
    if (tid < N) {
      float t = 0.f;
#pragma unroll
      for (int rg = 0; rg < DRS; ++rg) t += scr[rg * N + tid];
      a.part_b[(int64_t)blockIdx.x * N + tid] = t;
    }
  }
}

inline int blh_groups(int64_t rows) {
  const int64_t nblocks = rows / BLH_R;
  int64_t groups = 256 * 4;                                     // 34 KB of LDS and <= 128 registers: four workgroups per CU
  if (groups > nblocks / 8) groups = nblocks / 8 > 0 ? nblocks / 8 : 1;      // >= 8 row blocks per workgroup
  return (int)groups;
}

template <int NT, int KT, bool BN, bool ZH, bool XL>
int blh_launch(const BlbArgs& a, int groups, hipStream_t stream) {
  const size_t smem = (size_t)((CMR_BLH_HILO != 0 && NT == 4) ? 6 : 4) * BLH_IMG + (size_t)6 * 32 * NT * sizeof(float) + (XL ? (size_t)4 * 32 * KT * sizeof(float) : 0) +
                      (NT == 4 ? (size_t)4 * 8 * 1024 : 0);                  // (N = 128: the four waves' W^T fragments)
  static CmrSmemCache granted{};
  if (cmr_grant_smem(reinterpret_cast<const void*>(bn_linear_bwd_bf16_kernel<NT, KT, BN, ZH, XL>), smem, granted) != CMR_OK) return CMR_ELAUNCH;
  hipLaunchKernelGGL((bn_linear_bwd_bf16_kernel<NT, KT, BN, ZH, XL>), dim3(groups), dim3(256), smem, stream, a);
  return CMR_OK;
}

template <int NT, int KT>
int blh_dispatch(const BlbArgs& a, bool bn, bool zh, bool xl, int groups, hipStream_t stream) {
  if (!bn) return blh_launch<NT, KT, false, false, false>(a, groups, stream);
  if (xl) {
    if constexpr (NT == 2) {
      return zh ? blh_launch<NT, KT, true, true, true>(a, groups, stream) : blh_launch<NT, KT, true, false, true>(a, groups, stream);
    } else {
      return CMR_EUNSUPPORTED;
    }
  }
  return zh ? blh_launch<NT, KT, true, true, false>(a, groups, stream) : blh_launch<NT, KT, true, false, false>(a, groups, stream);
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
static int blb_entry(bool bf16, const float* dz, int64_t lddz, const float* z, int64_t ldz, float slope, const float* h, int64_t ldh,
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
  if (xl && bf16) CMR_REQUIRE(xslope >= 0.f && ldx % 4 == 0);
  int groups = bf16 ? blh_groups(rows) : blb_groups(rows, n, k);
  const int R = bf16 ? BLH_R : blb_rows_per_block(rows, n);
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
  if (bf16) {
    if (n == 64 && k == 64) rc = blh_dispatch<2, 2>(a, bn, zh, xl, groups, stream);
    else if (n == 64 && k == 128) rc = blh_dispatch<2, 4>(a, bn, zh, xl, groups, stream);
    else if (n == 128 && k == 64) rc = blh_dispatch<4, 2>(a, bn, zh, xl, groups, stream);
    else rc = blh_dispatch<4, 4>(a, bn, zh, xl, groups, stream);
  } else if (n == 64 && k == 64) rc = blb_dispatch<2, 2>(a, bn, zh, xl, groups, stream);
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

extern "C" int cmr_bn_linear_bwd_f32(const float* dz, int64_t lddz, const float* z, int64_t ldz, float slope, const float* h, int64_t ldh,
                                     const float* stat, const float* coef, int mask_from_h, float* dzm, int64_t lddzm, const float* x,
                                     int64_t ldx, const float* xstat, float xslope, float* xcoef, float* xdgamma, float* xdbeta,
                                     const float* w, int64_t ldw, const float* res, int64_t ldres, float* dx, int64_t lddx, int64_t rows,
                                     int n, int k, float* dw, int64_t lddw, int accumulate, float* db, int accumulate_db,
                                     int64_t seg_rows, float* seg_db, void* ws, int64_t ws_bytes, hipStream_t stream) {
  return blb_entry(false, dz, lddz, z, ldz, slope, h, ldh, stat, coef, mask_from_h, dzm, lddzm, x, ldx, xstat, xslope, xcoef, xdgamma, xdbeta, w, ldw,
                   res, ldres, dx, lddx, rows, n, k, dw, lddw, accumulate, db, accumulate_db, seg_rows, seg_db, ws, ws_bytes, stream);
}

// The same layer backward with the two products (dW = dh^T x', dx = dh W) on the bf16 matrix cores, fp32 accumulate: operands rounded to
// bf16 (RNE) on the way into LDS, everything else (masks, BatchNorm-backward arithmetic, column / segment sums, the lazy operand's sums,
// partial sums and their reduction) as cmr_bn_linear_bwd_f32.  Same arguments, shapes and return codes.
extern "C" int64_t cmr_bn_linear_bwd_bf16_workspace_bytes(int64_t rows, int n, int k) {
  if (!blb_shape_ok(rows, n, k)) return 0;
  return (int64_t)(blh_groups(rows) + 256) * ((int64_t)n * k + 2 * k + n) * (int64_t)sizeof(float);
}

extern "C" int cmr_bn_linear_bwd_bf16_f32(const float* dz, int64_t lddz, const float* z, int64_t ldz, float slope, const float* h, int64_t ldh,
                                          const float* stat, const float* coef, int mask_from_h, float* dzm, int64_t lddzm, const float* x,
                                          int64_t ldx, const float* xstat, float xslope, float* xcoef, float* xdgamma, float* xdbeta,
                                          const float* w, int64_t ldw, const float* res, int64_t ldres, float* dx, int64_t lddx, int64_t rows,
                                          int n, int k, float* dw, int64_t lddw, int accumulate, float* db, int accumulate_db,
                                          int64_t seg_rows, float* seg_db, void* ws, int64_t ws_bytes, hipStream_t stream) {
  return blb_entry(true, dz, lddz, z, ldz, slope, h, ldh, stat, coef, mask_from_h, dzm, lddzm, x, ldx, xstat, xslope, xcoef, xdgamma, xdbeta, w, ldw,
                   res, ldres, dx, lddx, rows, n, k, dw, lddw, accumulate, db, accumulate_db, seg_rows, seg_db, ws, ws_bytes, stream);
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
static int blf_entry(bool bf16, const float* x, int64_t ldx, int k, const float* pro_stat, float pro_slope, const float* w, int64_t ldw,
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
  if (bf16) {
    CMR_REQUIRE(ldw % 4 == 0 && cmr_aligned16(w));
    const int groups = blg_groups(rows);
    CMR_REQUIRE(ws_bytes >= (int64_t)groups * 4 * 4 * n * (int64_t)sizeof(float));
    const BlfArgs a{x, ldx, pro_stat, pro_slope, w, ldw, bias, bias_seg_rows > 0 ? bias_stride : 0, h, ldh, (float*)ws, rows,
                    bias_seg_rows > 0 ? bias_seg_rows / 32 : rows / 32, groups};
    const dim3 grid((unsigned)groups, (unsigned)(n / 64));
    const size_t smem = (size_t)2 * (k / 16) * 1024 + (size_t)(2 * k + 512) * sizeof(float);
    if (k == 64) {
      if (pro_stat) hipLaunchKernelGGL((bn_linear_fwd_bf16_kernel<4, true>), grid, dim3(256), smem, stream, a, n);
      else hipLaunchKernelGGL((bn_linear_fwd_bf16_kernel<4, false>), grid, dim3(256), smem, stream, a, n);
    } else {
      if (pro_stat) hipLaunchKernelGGL((bn_linear_fwd_bf16_kernel<8, true>), grid, dim3(256), smem, stream, a, n);
      else hipLaunchKernelGGL((bn_linear_fwd_bf16_kernel<8, false>), grid, dim3(256), smem, stream, a, n);
    }
    hipLaunchKernelGGL(bn_stats_merge_cnt_kernel, dim3(n), dim3(256), 0, stream, (const float*)ws, groups * 4, rows, n, eps, momentum, gamma, beta,
                       running_mean, running_var, stat);
    return cmr_launch_status();
  }
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

extern "C" int cmr_linear_bn_fwd_f32(const float* x, int64_t ldx, int k, const float* pro_stat, float pro_slope, const float* w, int64_t ldw,
                                     const float* bias, int64_t bias_seg_rows, int64_t bias_stride, float* h, int64_t ldh, int64_t rows, int n,
                                     float eps, float momentum, const float* gamma, const float* beta, float* running_mean,
                                     float* running_var, float* stat, void* ws, int64_t ws_bytes, hipStream_t stream) {
  return blf_entry(false, x, ldx, k, pro_stat, pro_slope, w, ldw, bias, bias_seg_rows, bias_stride, h, ldh, rows, n, eps, momentum, gamma, beta,
                   running_mean, running_var, stat, ws, ws_bytes, stream);
}

// The same layer forward with its product on the bf16 matrix cores (fp32 accumulate; x' and W rounded to bf16, round-to-nearest-even): h,
// the statistics of that h and the running statistics as cmr_linear_bn_fwd_f32 leaves them.  Same arguments and shapes; own workspace size.
extern "C" int64_t cmr_linear_bn_fwd_bf16_workspace_bytes(int64_t rows, int n, int k) {
  if (!(k == 64 || k == 128) || !(n == 64 || n == 128) || rows < 32 || rows % 32) return 0;
  return (int64_t)blg_groups(rows) * 4 * 4 * n * (int64_t)sizeof(float);
}

extern "C" int cmr_linear_bn_fwd_bf16_f32(const float* x, int64_t ldx, int k, const float* pro_stat, float pro_slope, const float* w, int64_t ldw,
                                          const float* bias, int64_t bias_seg_rows, int64_t bias_stride, float* h, int64_t ldh, int64_t rows,
                                          int n, float eps, float momentum, const float* gamma, const float* beta, float* running_mean,
                                          float* running_var, float* stat, void* ws, int64_t ws_bytes, hipStream_t stream) {
  return blf_entry(true, x, ldx, k, pro_stat, pro_slope, w, ldw, bias, bias_seg_rows, bias_stride, h, ldh, rows, n, eps, momentum, gamma, beta,
                   running_mean, running_var, stat, ws, ws_bytes, stream);
}
