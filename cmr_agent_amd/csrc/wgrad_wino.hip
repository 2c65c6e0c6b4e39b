// Winograd-domain fp32 weight gradient of the stride-1 3x3 convolutions, 64 -> 64 channels (round 6; the geometric model's update,
// reference Train_Geo.py:166-174 `loss.backward()` through models/ImageResNet.py:5-40 and the 64-channel decoder convolutions).
//
// The forward and data-gradient convolutions run on F(2x2,3x3) -- Y = A^T[(G g G^T) (.) (B^T d B)]A, 16 of the 36 multiplies per 2x2 output
// tile -- while cmr_conv3x3_wgrad_f32 contracts all 9 taps directly (36/36) and is the largest entry point of the C5 step.  The transpose of
// the same identity gives the weight gradient with the same saving:
//
//     dg = sum over the 2x2 output tiles of   G^T [ (A dY A^T) (.) (B^T d B) ] G
//
// i.e. 16 position GEMMs  M[xi][co][ci] = sum_tiles P[xi][tile][co] V[xi][tile][ci]  (K = tiles; P = A dY A^T from the 2x2 block of output
// gradients, V = B^T d B from the 4x4 input window: additions only) and a 4x4 -> 3x3 fold with G once per launch.  Accuracy: float64 against
// fp32, direct against this form at 352 x 1216 x 8 (K = 856 064 tiles): tools/wino_wgrad_check.py (max error 1.5 x the direct fp32 sum's).
//
// Kernel: persistent workgroups of 8 waves, one per CU.  A STAGE is 8 horizontally adjacent tiles (2 x 16 output pixels):
//   * transform: wave t owns tile t, lane = channel -- the 4x4 input window and the 2x2 gradient block arrive as 16 + 4 coalesced 256-byte
//     row loads per wave (requested one stage ahead, into registers), are transformed in registers (32 + 12 additions per channel) and go
//     to LDS as P[xi][tile][64], V[xi][tile][64] (odd tiles rotated by 32 channels: the two lane halves of an operand read -- tiles 2k and
//     2k + 1 -- fall on opposite halves of the 64 banks);
//   * multiply: wave w owns positions 2w, 2w + 1 with the whole 64 x 64 output of each in 8 accumulator tiles (128 registers) for the
//     launch; per tile pair and position 4 ds_read_b32 feed 4 v_mfma_f32_32x32x2_f32 (A = P: lane l = (tile l>>5, cout l&31), B = V).
//   Two LDS buffers (2 x 64 KB), ONE barrier per stage: a stage's transform writes the other buffer while its matrix instructions read this one.
// Partials [workgroup][16][64][64] -> wgrad_wino_reduce_kernel: sums over workgroups in double in a fixed order (deterministic: data-parallel
// ranks must produce identical buckets), folds G^T M G and writes dw [Cout][Cin][3][3].
#include "cmr_common.h"

namespace {

constexpr int WW_T = 8;                        // tiles per stage
constexpr int WW_OP = 16 * WW_T * 64;          // floats of one operand image (P or V) of a stage
constexpr int WW_BUF = 2 * WW_OP;              // floats per buffer
constexpr int WW_SMEM = 2 * WW_BUF * 4;        // bytes: 131 072

__global__ __launch_bounds__(512) void conv3x3_wgrad_wino_kernel(const float* __restrict__ x, const float* __restrict__ dy, int B, int H, int W,
                                                                 int nsx, int nstages, float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float ww_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  const int TH = H >> 1, TWn = W >> 1;
  // stages of this workgroup: a contiguous range (consecutive stages walk along a tile row: the rows a stage shares with the one below
  // come back from L2)
  const int s_begin = (int)((int64_t)nstages * blockIdx.x / gridDim.x), s_end = (int)((int64_t)nstages * (blockIdx.x + 1) / gridDim.x);

  f32x16 acc[2][2][2];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][a][b][r] = 0.f;

  struct Where { int ty, tx; };                 // tile of THIS wave in a stage (wave uniform)
  auto where = [&](int s) __attribute__((always_inline)) {
    Where q;
    const int txs = s % nsx, r = s / nsx;
    q.ty = r % TH; q.tx = txs * WW_T + wave;
    return q;
  };
  // raw window / gradient block of this wave's tile of stage s: clamped addresses, branch free; nothing touches a loaded value before the
  // transform a stage later (a select here would put s_waitcnt vmcnt(0) in front of the stage's matrix instructions)
  auto request = [&](int s, float (&xw)[16], float (&yw)[4]) __attribute__((always_inline)) {
    const int txs = s % nsx, r = s / nsx;
    const int ty = r % TH, b = r / TH;
    const int tx = min(txs * WW_T + wave, TWn - 1);
    const float* xb = x + (int64_t)b * H * W * 64 + lane;
    const float* yb = dy + (int64_t)b * H * W * 64 + lane;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int iy = min(max(2 * ty - 1 + i, 0), H - 1);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ix = min(max(2 * tx - 1 + j, 0), W - 1);
        xw[4 * i + j] = xb[(unsigned)((iy * W + ix) * 64)];
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) yw[2 * i + j] = yb[(unsigned)(((2 * ty + i) * W + 2 * tx + j) * 64)];
  };
  // P = A dY A^T, V = B^T d B of this wave's tile -> buffer `buf`
  auto transform = [&](const Where q, const float (&xw)[16], const float (&yw)[4], int buf) __attribute__((always_inline)) {
    const bool live = q.tx < TWn;
    const bool r0 = q.ty > 0, r3 = 2 * q.ty + 2 < H, c0 = q.tx > 0, c3 = 2 * q.tx + 2 < W;
    float d[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool ok = (i == 0 ? r0 : (i == 3 ? r3 : true)) && (j == 0 ? c0 : (j == 3 ? c3 : true));
        d[i][j] = ok ? xw[4 * i + j] : 0.f;          // zero padding (the clamped load fetched a neighbour)
      }
    float t[4][4], V[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      t[0][j] = d[0][j] - d[2][j];
      t[1][j] = d[1][j] + d[2][j];
      t[2][j] = d[2][j] - d[1][j];
      t[3][j] = d[1][j] - d[3][j];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      V[i][0] = t[i][0] - t[i][2];
      V[i][1] = t[i][1] + t[i][2];
      V[i][2] = t[i][2] - t[i][1];
      V[i][3] = t[i][1] - t[i][3];
    }
    float y00 = live ? yw[0] : 0.f, y01 = live ? yw[1] : 0.f, y10 = live ? yw[2] : 0.f, y11 = live ? yw[3] : 0.f;
    float s[4][2], P[4][4];
    s[0][0] = y00;        s[0][1] = y01;
    s[1][0] = y00 + y10;  s[1][1] = y01 + y11;
    s[2][0] = y00 - y10;  s[2][1] = y01 - y11;
    s[3][0] = -y10;       s[3][1] = -y11;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      P[i][0] = s[i][0];
      P[i][1] = s[i][0] + s[i][1];
      P[i][2] = s[i][0] - s[i][1];
      P[i][3] = -s[i][1];
    }
    float* Pb = ww_smem + buf * WW_BUF + wave * 64 + ((lane + 32 * (wave & 1)) & 63);
    float* Vb = Pb + WW_OP;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        Pb[(4 * i + j) * (WW_T * 64)] = P[i][j];
        Vb[(4 * i + j) * (WW_T * 64)] = V[i][j];
      }
  };
  // the 4 tile pairs of buffer `buf` into this wave's two positions
  auto multiply = [&](int buf) __attribute__((always_inline)) {
    // lane address of tile 2k + h, channel tile ct: row (2k + h) 64 + ((ct ^ h) 32) + l31   (the rotation of the odd tiles)
    const float* Pb = ww_smem + buf * WW_BUF + (2 * wave) * (WW_T * 64) + h * 64 + l31;
    const float* Vb = Pb + WW_OP;
    const int o0 = h * 32, o1 = (1 - h) * 32;
#pragma unroll
    for (int k = 0; k < WW_T / 2; ++k)
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int base = p * (WW_T * 64) + k * 128;
        const float a0 = Pb[base + o0], a1 = Pb[base + o1];
        const float b0 = Vb[base + o0], b1 = Vb[base + o1];
        acc[p][0][0] = cmr_mfma32(a0, b0, acc[p][0][0]);
        acc[p][0][1] = cmr_mfma32(a0, b1, acc[p][0][1]);
        acc[p][1][0] = cmr_mfma32(a1, b0, acc[p][1][0]);
        acc[p][1][1] = cmr_mfma32(a1, b1, acc[p][1][1]);
      }
  };

  if (s_begin < s_end) {
    float xw[16], yw[4];
    request(s_begin, xw, yw);
    transform(where(s_begin), xw, yw, 0);
    Where qn = where(s_begin + 1 < s_end ? s_begin + 1 : s_begin);
    request(s_begin + 1 < s_end ? s_begin + 1 : s_begin, xw, yw);
    __syncthreads();
    for (int s = s_begin; s < s_end; ++s) {
      const int buf = (s - s_begin) & 1;
      multiply(buf);
      if (s + 1 < s_end) {                      // (uniform over the workgroup)
        transform(qn, xw, yw, buf ^ 1);
        const int s2 = s + 2 < s_end ? s + 2 : s + 1;     // past the end: a harmless re-read
        qn = where(s2);
        request(s2, xw, yw);
      }
      __syncthreads();
    }
  }
  float* out = part + (int64_t)blockIdx.x * 16 * 4096;
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          out[((2 * wave + p) * 64 + ct * 32 + cmr_mfma_row(r, lane)) * 64 + it * 32 + l31] = acc[p][ct][it][r];
}

// dw[co][ci][ky][kx] = (G^T M G)[ky][kx],  M[xi] = sum over the workgroups' partials (double, fixed order).
// Workgroup = 32 consecutive (co, ci) outputs x 16 slice groups.
constexpr int WWR_OUT = 32, WWR_GRP = 16;
__global__ __launch_bounds__(WWR_OUT * WWR_GRP) void wgrad_wino_reduce_kernel(const float* __restrict__ part, int groups, float* __restrict__ dw) {
  __shared__ double sm[WWR_GRP][16][WWR_OUT];
  const int o = threadIdx.x % WWR_OUT, g = threadIdx.x / WWR_OUT;
  const int e = blockIdx.x * WWR_OUT + o;          // co * 64 + ci
  double m[16];
#pragma unroll
  for (int xi = 0; xi < 16; ++xi) m[xi] = 0.0;
  for (int k = g; k < groups; k += WWR_GRP) {
    float v[16];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) v[xi] = part[((int64_t)k * 16 + xi) * 4096 + e];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) m[xi] += (double)v[xi];
  }
#pragma unroll
  for (int xi = 0; xi < 16; ++xi) sm[g][xi][o] = m[xi];
  __syncthreads();
  if (g == 0) {
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
      for (int k = 1; k < WWR_GRP; ++k) m[xi] += sm[k][xi][o];
    // G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]:  rows first (G^T M), then columns
    double q[3][4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const double m0 = m[b], m1 = m[4 + b], m2 = m[8 + b], m3 = m[12 + b];
      q[0][b] = m0 + 0.5 * (m1 + m2);
      q[1][b] = 0.5 * (m1 - m2);
      q[2][b] = 0.5 * (m1 + m2) + m3;
    }
    float* d = dw + (int64_t)e * 9;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      d[3 * a + 0] = (float)(q[a][0] + 0.5 * (q[a][1] + q[a][2]));
      d[3 * a + 1] = (float)(0.5 * (q[a][1] - q[a][2]));
      d[3 * a + 2] = (float)(0.5 * (q[a][1] + q[a][2]) + q[a][3]);
    }
  }
}

inline int ww_groups(int nstages) {
  int g = nstages / 4;                             // at least ~4 stages per workgroup (the pipeline's fill), at most one workgroup per CU
  if (g > 256) g = 256;
  if (g < 1) g = 1;
  return g;
}

}  // namespace

extern "C" int64_t cmr_conv3x3_wgrad_wino_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
  (void)B; (void)H; (void)W; (void)Cin; (void)Cout;
  return (int64_t)256 * 16 * 4096 * (int64_t)sizeof(float);
}

extern "C" int cmr_conv3x3_wgrad_wino_f32(const float* x, const float* dy, int B, int H, int W, int Cin, int Cout, float* dw, void* ws,
                                          int64_t ws_bytes, hipStream_t stream) {
  CMR_REQUIRE(x && dy && dw && ws && B > 0 && H > 0 && W > 0);
  if (Cin != 64 || Cout != 64 || (H & 1) || (W & 1) || H < 2 || W < 2) return CMR_EUNSUPPORTED;
  CMR_REQUIRE((int64_t)H * W * 64 < 0x7fffffff && (int64_t)B * (H / 2) * ((W / 2 + WW_T - 1) / WW_T) < 0x7fffffff);
  const int nsx = (W / 2 + WW_T - 1) / WW_T;
  const int nstages = B * (H / 2) * nsx;
  const int groups = ww_groups(nstages);
  CMR_REQUIRE(ws_bytes >= (int64_t)groups * 16 * 4096 * (int64_t)sizeof(float));
  static CmrSmemCache granted{};
  if (cmr_grant_smem(reinterpret_cast<const void*>(conv3x3_wgrad_wino_kernel), WW_SMEM, granted) != CMR_OK) return CMR_ELAUNCH;
  float* part = (float*)ws;
  hipLaunchKernelGGL(conv3x3_wgrad_wino_kernel, dim3(groups), dim3(512), WW_SMEM, stream, x, dy, B, H, W, nsx, nstages, part);
  hipLaunchKernelGGL(wgrad_wino_reduce_kernel, dim3(4096 / WWR_OUT), dim3(WWR_OUT * WWR_GRP), 0, stream, (const float*)part, groups, dw);
  return cmr_launch_status();
}
