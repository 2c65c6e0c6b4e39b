// bf16 matrix-core variant of the fused ConvBNReLURes1D block (cbr_block.hip; SURVEY.md 7 step 9, BASELINE configs[2] / [3]):
//     hid = LReLU(W1 x + b1) ;  y = LReLU(W2 hid + b2 + shortcut(x)) ,  shortcut = Wsc x + bsc | x
// Same contract as cmr_cbr_block_f32 (fp32 rows in / out, fp32 weights and biases as the module's plan holds them, per-batch
// bias rows, per-tile column maxima); the products run on v_mfma_f32_32x32x16_bf16 with fp32 accumulation, operands rounded to
// bf16 (round to nearest even) on their way into the matrix core.  At 1/16 of the fp32 matrix time the block is HBM-class
// (256 B in + 256 B out per row against 41 kFLOP), so what matters is that nothing but x and y touches memory:
//   * weights are converted once per workgroup into LDS as ready-made A fragments ([tile][16-deep k step][64 lanes][8 bf16]:
//     one conflict-free ds_read_b128 per MFMA);
//   * everything is computed transposed (D'[channel][row]): a lane owns ONE row, its input fragment of k step s is the 32
//     contiguous bytes x[row][16 s + 8 h .. + 7] (two float4 loads, converted in registers);
//   * the hidden activations never leave the registers: accumulator registers 8 s'' .. 8 s'' + 7 of hidden tile t ARE the B
//     operand of k step (t, s'') of the second GEMM -- they hold the channels 32 t + 8 (2 s'' + (j >> 2)) + 4 h + (j & 3), and the
//     W2 fragments are laid out in LDS with their k slots permuted to exactly that order;
//   * the identity shortcut and the epilogue are fp32.
#include "cmr_common.h"

namespace {

typedef __bf16 cb_bf16x8 __attribute__((ext_vector_type(8)));

__device__ __attribute__((aligned(16))) float cbb_zero[256] = {0.f};   // NOT const (see cbr_block.hip)

struct CbbArgs {
  const float* x1; int64_t ld1;
  const float* x2; int64_t ld2; const int32_t* idx2; int64_t div2;
  int k1;
  const float* w1; const float* b1; int64_t b1_stride;
  const float* w2; const float* b2; int64_t b2_stride;
  const float* wsc;
  float* y; int64_t ldy;
  float* colmax_part;
  int64_t rows; int64_t rows_per_batch; float slope;
};

constexpr int CBB_MAXB = 16;

__device__ __forceinline__ cb_bf16x8 cbb_pack(const f32x4& a, const f32x4& b) {
  cb_bf16x8 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    r[i] = (__bf16)a[i];
    r[4 + i] = (__bf16)b[i];
  }
  return r;
}

// KX, CH may be 8 (the agent's first block: 5 + 3 padding channels): one k step whose upper half is zero.
template <int KX, int CH, int CO, bool CONV_SC>
__global__ __launch_bounds__(512) void cbr_block_bf16_kernel(const CbbArgs a) {
  constexpr int S1 = (KX + 15) / 16;            // k steps over the input width
  constexpr int T1 = (CH + 31) / 32, T2 = CO / 32;
  constexpr int CHP = 32 * T1;
  constexpr int S2 = 2 * T1;                    // k steps over the (padded) hidden width
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  cb_bf16x8* W1f = reinterpret_cast<cb_bf16x8*>(smem_raw);                  // [T1][S1][64]
  cb_bf16x8* W2f = W1f + T1 * S1 * 64;                                       // [T2][S2][64]
  cb_bf16x8* Wsf = W2f + T2 * S2 * 64;                                       // [T2][S1][64]   (only if CONV_SC)
  float* B1s = reinterpret_cast<float*>(Wsf + (CONV_SC ? T2 * S1 * 64 : 0)); // [nb][CHP]
  float* B2s = B1s + CBB_MAXB * CHP;                                          // [nb][CO]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  const uint32_t rows = (uint32_t)a.rows, rpb = (uint32_t)a.rows_per_batch;
  const int nb1 = a.b1_stride > 0 ? (int)((rows + rpb - 1) / rpb) : 1, nb2 = a.b2_stride > 0 ? (int)((rows + rpb - 1) / rpb) : 1;
  // ---- weights -> bf16 A fragments.  Fragment (t, s), lane (i, hh), element j = W[32 t + i][k(s, hh, j)]
  for (int e = tid; e < T1 * S1 * 64; e += 512) {
    const int ln = e & 63, s = (e >> 6) % S1, t = (e >> 6) / S1;
    const int n = 32 * t + (ln & 31), k0 = 16 * s + 8 * (ln >> 5);
    cb_bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (__bf16)((n < CH && k0 + j < KX) ? a.w1[(int64_t)n * KX + k0 + j] : 0.f);
    W1f[e] = v;
  }
  for (int e = tid; e < T2 * S2 * 64; e += 512) {
    const int ln = e & 63, s = (e >> 6) % S2, n = (e >> 6) / S2;
    const int t = s >> 1, s2 = s & 1, hh = ln >> 5;
    const int co = 32 * n + (ln & 31);
    cb_bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = 32 * t + (j & 3) + 8 * (2 * s2 + (j >> 2)) + 4 * hh;     // hidden channel held by accumulator register 8 s2 + j
      v[j] = (__bf16)(c < CH ? a.w2[(int64_t)co * CH + c] : 0.f);
    }
    W2f[e] = v;
  }
  if (CONV_SC)
    for (int e = tid; e < T2 * S1 * 64; e += 512) {
      const int ln = e & 63, s = (e >> 6) % S1, n = (e >> 6) / S1;
      const int co = 32 * n + (ln & 31), k0 = 16 * s + 8 * (ln >> 5);
      cb_bf16x8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (__bf16)(k0 + j < KX ? a.wsc[(int64_t)co * KX + k0 + j] : 0.f);
      Wsf[e] = v;
    }
  for (int e = tid; e < nb1 * CHP; e += 512) B1s[e] = (e % CHP) < CH ? a.b1[(int64_t)(e / CHP) * a.b1_stride + e % CHP] : 0.f;
  for (int e = tid; e < nb2 * CO; e += 512) B2s[e] = a.b2[(int64_t)(e / CO) * a.b2_stride + e % CO];
  __syncthreads();

  const uint32_t ntiles = (rows + 31) / 32, tstride = gridDim.x * 8;
  const float* x2b = a.x2 ? a.x2 : a.x1;
  const int32_t* izero = reinterpret_cast<const int32_t*>(cbb_zero);
  auto load_idx = [&](uint32_t tile) -> int32_t {
    const uint32_t row = tile * 32 + l31;
    const uint32_t ok = (a.idx2 != nullptr && tile < ntiles && row < rows) ? 1u : 0u;
    return (a.idx2 ? a.idx2 : izero)[ok * row];
  };
  struct RowPtr { const float* p1; const float* p2; };
  auto row_ptr = [&](uint32_t tile, int32_t idxv) -> RowPtr {
    uint32_t row = tile * 32 + l31;
    row = (tile < ntiles && row < rows) ? row : 0;
    const int64_t r2 = a.idx2 ? (int64_t)idxv : (int64_t)(row / (uint32_t)a.div2);
    return {a.x1 + (int64_t)row * a.ld1, x2b + r2 * a.ld2 - a.k1};
  };
  // input fragments of one tile: k step s = channels 16 s + 8 h .. + 7 of this lane's row (k1 is a multiple of 4, so a float4
  // never straddles the two sources); channels past KX read the zero page
  auto load_x = [&](const RowPtr& r, f32x4 (&lo)[S1], f32x4 (&hi)[S1]) {
#pragma unroll
    for (int s = 0; s < S1; ++s) {
      constexpr bool FULL = KX % 16 == 0;                       // compile-time for every width but the 8-channel first block
      const int c0 = 16 * s + 8 * h, c1 = c0 + 4;
      lo[s] = *reinterpret_cast<const f32x4*>((FULL || c0 < KX) ? (c0 < a.k1 ? r.p1 : r.p2) + c0 : cbb_zero);
      hi[s] = *reinterpret_cast<const f32x4*>((FULL || c1 < KX) ? (c1 < a.k1 ? r.p1 : r.p2) + c1 : cbb_zero);
    }
  };

  uint32_t tile = blockIdx.x * 8 + wave;
  f32x4 xlo[S1], xhi[S1], nlo[S1], nhi[S1];
  int32_t idn = load_idx(tile);
  RowPtr rc = row_ptr(tile, idn);
  load_x(rc, xlo, xhi);
  idn = load_idx(tile + tstride);
  for (; tile < ntiles; tile += tstride) {
    const uint32_t row = tile * 32 + l31;
    const bool valid = row < rows;
    const uint32_t batch = (valid ? row : 0) / rpb;
    const RowPtr rn = row_ptr(tile + tstride, idn);       // next tile's fragments fly under this tile's work
    load_x(rn, nlo, nhi);
    idn = load_idx(tile + 2 * tstride);
    // identity shortcut: this lane's row in OUTPUT layout (channels 32 n + 8 q + 4 h .. + 3), fp32
    f32x4 sc[T2][4];
    if (!CONV_SC) {
#pragma unroll
      for (int n = 0; n < T2; ++n)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = 32 * n + 8 * q + 4 * h;
          sc[n][q] = *reinterpret_cast<const f32x4*>(c < KX ? (c < a.k1 ? rc.p1 : rc.p2) + c : cbb_zero);
        }
    }
    cb_bf16x8 xb[S1];
#pragma unroll
    for (int s = 0; s < S1; ++s) xb[s] = cbb_pack(xlo[s], xhi[s]);
    // ---- GEMM 1
    f32x16 hid[T1];
#pragma unroll
    for (int t = 0; t < T1; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) hid[t][r] = 0.f;
#pragma unroll
    for (int s = 0; s < S1; ++s)
#pragma unroll
      for (int t = 0; t < T1; ++t) hid[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W1f[(t * S1 + s) * 64 + lane], xb[s], hid[t], 0, 0, 0);
    {
      const float* b1 = B1s + (a.b1_stride > 0 ? batch : 0) * CHP + 4 * h;
#pragma unroll
      for (int t = 0; t < T1; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 bv = *reinterpret_cast<const f32x4*>(b1 + 32 * t + 8 * q);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v = hid[t][4 * q + e] + bv[e];
            hid[t][4 * q + e] = v > 0.f ? v : v * a.slope;
          }
        }
    }
    // ---- GEMM 2 (+ shortcut GEMM)
    f32x16 acc[T2];
#pragma unroll
    for (int n = 0; n < T2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
#pragma unroll
    for (int t = 0; t < T1; ++t)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        cb_bf16x8 hb;
#pragma unroll
        for (int j = 0; j < 8; ++j) hb[j] = (__bf16)hid[t][8 * s2 + j];
#pragma unroll
        for (int n = 0; n < T2; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W2f[(n * S2 + 2 * t + s2) * 64 + lane], hb, acc[n], 0, 0, 0);
      }
    if (CONV_SC) {
#pragma unroll
      for (int s = 0; s < S1; ++s)
#pragma unroll
        for (int n = 0; n < T2; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wsf[(n * S1 + s) * 64 + lane], xb[s], acc[n], 0, 0, 0);
    }
    // ---- epilogue
    const float* b2 = B2s + (a.b2_stride > 0 ? batch : 0) * CO + 4 * h;
    f32x4 ov[T2][4];
#pragma unroll
    for (int n = 0; n < T2; ++n)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(b2 + n * 32 + q * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float s = acc[n][4 * q + e] + bv[e];
          if (!CONV_SC) s += sc[n][q][e];
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
    rc = rn;
#pragma unroll
    for (int s = 0; s < S1; ++s) {
      xlo[s] = nlo[s];
      xhi[s] = nhi[s];
    }
  }
}

template <int KX, int CH, int CO, bool CONV_SC>
int launch_cbb(const CbbArgs& a, hipStream_t stream) {
  constexpr int S1 = (KX + 15) / 16, T1 = (CH + 31) / 32, T2 = CO / 32, S2 = 2 * T1;
  constexpr size_t smem = (size_t)(T1 * S1 + T2 * S2 + (CONV_SC ? T2 * S1 : 0)) * 1024 + (size_t)CBB_MAXB * (32 * T1 + CO) * sizeof(float);
  static_assert(smem <= 160 * 1024, "weights must fit in LDS");
  static CmrSmemCache granted{};
  if (cmr_grant_smem(reinterpret_cast<const void*>(cbr_block_bf16_kernel<KX, CH, CO, CONV_SC>), smem, granted) != CMR_OK) return CMR_ELAUNCH;
  const int64_t ntiles = (a.rows + 31) / 32;
  int64_t grid = (ntiles + 7) / 8;
  if (grid > 512) grid = 512;                  // 2 resident workgroups per CU, tiles walked with stride
  hipLaunchKernelGGL((cbr_block_bf16_kernel<KX, CH, CO, CONV_SC>), dim3((unsigned)grid), dim3(512), smem, stream, a);
  return cmr_launch_status();
}

}  // namespace

extern "C" int cmr_cbr_block_bf16_f32(const float* x1, int64_t ld1, int k1, const float* x2, int64_t ld2, const int32_t* idx2,
                                      int64_t div2, int kx, int ch, int co, const float* w1, const float* b1, int64_t b1_stride,
                                      const float* w2, const float* b2, int64_t b2_stride, const float* wsc, float* y, int64_t ldy,
                                      float* colmax_part, int64_t rows, int64_t rows_per_batch, float slope, hipStream_t stream) {
  CMR_REQUIRE(x1 && w1 && b1 && w2 && b2 && (y || colmax_part) && rows > 0 && rows_per_batch > 0);
  CMR_REQUIRE(k1 > 0 && k1 % 4 == 0 && k1 <= kx && (k1 == kx || x2) && ld1 % 4 == 0 && cmr_aligned16(x1));
  if (x2) CMR_REQUIRE(ld2 % 4 == 0 && cmr_aligned16(x2) && (idx2 || div2 >= 1));
  if (y) CMR_REQUIRE(ldy % 4 == 0 && cmr_aligned16(y));
  CMR_REQUIRE(cmr_aligned16(b1) && cmr_aligned16(b2) && b1_stride % 4 == 0 && b2_stride % 4 == 0 && (!colmax_part || cmr_aligned16(colmax_part)));
  const CbbArgs a{x1, ld1, x2, ld2, idx2, div2 < 1 ? 1 : div2, k1, w1, b1, b1_stride, w2, b2, b2_stride, wsc, y, ldy,
                  colmax_part, rows, rows_per_batch, slope};
  const bool conv = wsc != nullptr;
  if (rows >= (int64_t)0x7fffffc0 || ((b1_stride > 0 || b2_stride > 0) && (rows + rows_per_batch - 1) / rows_per_batch > CBB_MAXB))
    return CMR_EUNSUPPORTED;
  if (kx == 64 && ch == 64 && co == 64 && !conv) return launch_cbb<64, 64, 64, false>(a, stream);
  if (kx == 128 && ch == 128 && co == 64 && conv) return launch_cbb<128, 128, 64, true>(a, stream);
  if (kx == 64 && ch == 128 && co == 64 && conv) return launch_cbb<64, 128, 64, true>(a, stream);
  if (kx == 64 && ch == 128 && co == 128 && !conv) return launch_cbb<64, 128, 128, false>(a, stream);
  if (kx == 8 && ch == 8 && co == 64 && conv) return launch_cbb<8, 8, 64, true>(a, stream);
  return CMR_EUNSUPPORTED;
}
