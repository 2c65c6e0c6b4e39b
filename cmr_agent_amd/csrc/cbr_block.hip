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

__device__ __attribute__((aligned(16))) const float cbr_zero[256] = {0.f};

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

template <int KX, int CH, int CO, bool CONV_SC>
__global__ __launch_bounds__(512) void cbr_block_kernel(const CbrArgs a) {
  constexpr int LDX = KX + 4, LDH = CH + 4;
  constexpr int GX = KX / 8, GH = CH / 8;      // k-groups of the two GEMMs
  constexpr int T1 = (CH + 31) / 32, T2 = CO / 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* W1s = smem;                            // [32*T1][LDX]   (rows >= CH are zero)
  float* W2s = W1s + 32 * T1 * LDX;             // [CO][LDH]
  float* Wss = W2s + CO * LDH;                  // [CO][LDX]      (only if CONV_SC)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
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
  __syncthreads();

  const int64_t ntiles = (a.rows + 31) / 32;
  for (int64_t tile = (int64_t)blockIdx.x * 8 + wave; tile < ntiles; tile += (int64_t)gridDim.x * 8) {
    int64_t row = tile * 32 + l31;
    const bool valid = row < a.rows;
    if (!valid) row = 0;
    const int64_t batch = row / a.rows_per_batch;
    // ---- input fragments (B operand of GEMM 1 and of the shortcut; also the identity residual)
    const float* p1 = a.x1 + row * a.ld1 + 4 * h;
    const float* p2 = p1;
    if (a.x2) p2 = a.x2 + (a.idx2 ? (int64_t)a.idx2[row] : row / a.div2) * a.ld2 + 4 * h;
    f32x4 xf[GX];
#pragma unroll
    for (int g = 0; g < GX; ++g) {
      const int kk = g * 8;                    // + 4h is already in the pointers
      const float* p = kk + 4 * h < a.k1 ? p1 + kk : p2 + (kk - a.k1);
      xf[g] = *reinterpret_cast<const f32x4*>(p);
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
      const float* b1 = a.b1 + batch * a.b1_stride + 4 * h;
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
    // ---- epilogue
    const float* b2 = a.b2 + batch * a.b2_stride + 4 * h;
    float* yrow = a.y ? a.y + (tile * 32 + l31) * a.ldy + 4 * h : nullptr;
#pragma unroll
    for (int n = 0; n < T2; ++n)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = n * 32 + q * 8;
        const f32x4 bv = *reinterpret_cast<const f32x4*>(b2 + c);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float s = acc[n][4 * q + e] + bv[e];
          if (!CONV_SC && c < KX) s += xf[(c / 8)][e];          // identity shortcut: channel c+4h+e of x
          v[e] = s > 0.f ? s : s * a.slope;
        }
        if (yrow && valid) *reinterpret_cast<f32x4*>(yrow + c) = v;
        if (a.colmax_part) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float m = valid ? v[e] : -INFINITY;
#pragma unroll
            for (int s = 1; s < 32; s <<= 1) m = fmaxf(m, __shfl_xor(m, s));
            v[e] = m;
          }
          if (l31 == 0) *reinterpret_cast<f32x4*>(a.colmax_part + tile * CO + c + 4 * h) = v;
        }
      }
  }
}

// out[b][c] = max over the tiles of batch b of part[tile][c]  (tiles_per_batch consecutive tiles per batch)
__global__ __launch_bounds__(1024) void colmax_partials_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                               int tiles_per_batch, int C) {
  __shared__ float sm[1024];
  const int b = blockIdx.x, cblk = blockIdx.y * 64;
  const int c = cblk + (threadIdx.x & 63), grp = threadIdx.x >> 6;        // 16 tile groups x 64 channels
  float m = -INFINITY;
  if (c < C)
    for (int t = grp; t < tiles_per_batch; t += 16) m = fmaxf(m, part[((int64_t)b * tiles_per_batch + t) * C + c]);
  sm[threadIdx.x] = m;
  __syncthreads();
  if (grp == 0 && c < C) {
#pragma unroll
    for (int g = 1; g < 16; ++g) m = fmaxf(m, sm[threadIdx.x + 64 * g]);
    out[(int64_t)b * C + c] = m;
  }
}

template <int KX, int CH, int CO, bool CONV_SC>
int launch_cbr(const CbrArgs& a, hipStream_t stream) {
  constexpr int T1 = (CH + 31) / 32;
  constexpr size_t smem = (size_t)(32 * T1 * (KX + 4) + CO * (CH + 4) + (CONV_SC ? CO * (KX + 4) : 0)) * sizeof(float);
  static_assert(smem <= 160 * 1024, "weights must fit in LDS");
  static bool attr_set = false;
  if (!attr_set && smem > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(cbr_block_kernel<KX, CH, CO, CONV_SC>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return CMR_ELAUNCH;
    attr_set = true;
  }
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
  hipLaunchKernelGGL(colmax_partials_kernel, dim3(B, (C + 63) / 64), dim3(1024), 0, stream, part, out, tiles_per_batch, C);
  return cmr_launch_status();
}
