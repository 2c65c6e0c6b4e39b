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
  auto load_chunk = [&](int c) {
    const bool second = c >= nch1;
    const int kofs = (second ? c - nch1 : c) * KC;
    const int kvalid = (second ? a.k2 : a.k1) - kofs;
    const bool kin = c4 * 4 < kvalid;
    const float* xs = second ? a.x2 : a.x1;
    const int64_t ld = second ? a.ld2 : a.ld1;
#pragma unroll
    for (int i = 0; i < AL; ++i) {
      int64_t r = row_base + r0 + 32 * i;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (kin && r < a.rows) {
        int64_t s = second ? srow2[i] : r;
        v = *reinterpret_cast<const f32x4*>(xs + s * ld + kofs + c4 * 4);
      }
      ra[i] = v;
    }
    const int kw = (second ? a.k1 : 0) + kofs;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int n = col_base + r0 + 32 * i;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (kin && n < a.n_out) v = *reinterpret_cast<const f32x4*>(a.w + (int64_t)n * a.ldw + kw + c4 * 4);
      rb[i] = v;
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
