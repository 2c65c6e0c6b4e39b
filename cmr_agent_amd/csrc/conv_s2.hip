// Stride-2 3x3 convolution, fp32 matrix cores (the two strided convolutions of every down-sampling ResidualBlock: ImageResNet.py:9-14,
// :24-27).  Winograd does not apply to a strided convolution, so this is the direct implicit GEMM -- restructured against what the tiled
// kernel of conv.hip (cmr_conv3x3_nhwc_f32) spends outside its matrix instructions at stride 2 (63 % matrix-pipe duty, 82 TFLOP/s):
//   * that kernel stages a [64 cout][16 cin] weight slab per TAP through registers into LDS: 36 barriers per tile with 16 matrix
//     instructions per wave between them.  Here the weights are ready-made MFMA A fragments [9 taps][Cout/32][Cin/8][64 lanes][4]
//     (cmr_agent_amd/models/_pack.py:conv_s2_frags) read straight from L2 as 1 KB coalesced wave loads, two fragments ahead of their use:
//     no weight traffic through LDS, and the only barriers left are the two per 16-channel halo chunk (8 per tile, 144 matrix
//     instructions per wave between them);
//   * a workgroup is 2 row pairs x 2 cout tiles: a wave multiplies TWO output rows by ONE 32-cout tile, so each weight fragment feeds 8
//     matrix instructions and is fetched by two waves instead of four;
//   * the tile loop is straight-line: optional operands through host-made 0 / 1 offset multipliers and a zero page, bias once per
//     workgroup, residual rows requested with the last chunk, every output pinned before the first predicated store (the
//     lessons of conv_bf16.hip).
// Tile = 4 output rows x 32 output columns x 64 couts; halo chunk = 9 x 65 input pixels x 16 channels (46.8 KB, 3 workgroups per CU).
#include "cmr_common.h"

namespace {

struct S2Args {
  const float* x; int B, H, W, Cin;
  const float* wfrag; const float* bias; const float* res; const float* post;
  float* y; int Ho, Wo, Cout; float slope;
  int tiles_x, tiles_y, ntiles;
  int res_mul, bias_mul;
};

__device__ __attribute__((aligned(16))) float s2_zero16[4] = {0.f, 0.f, 0.f, 0.f};

constexpr int S2_KC = 16, S2_LDP = S2_KC + 4, S2_HR = 9, S2_HC = 65;
constexpr int S2_C4 = S2_KC / 4;                       // float4 pieces per halo pixel
constexpr int S2_NPIECE = S2_HR * S2_HC * S2_C4;       // 2340
constexpr int S2_NLOAD = (S2_NPIECE + 255) / 256;      // 10

template <int NCHUNK>
__global__ __launch_bounds__(256, 2) void conv3x3_s2_kernel(const S2Args a) {
  __shared__ __attribute__((aligned(16))) float halo[S2_HR * S2_HC * S2_LDP];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rp = wave >> 1, ct = wave & 1;             // row pair (output rows 2 rp, 2 rp + 1), cout tile inside the 64-cout group
  const int h = lane >> 5, l31 = lane & 31;
  const int nco = a.Cout / 64;
  constexpr int kgroups = NCHUNK * 2;                  // Cin = 16 NCHUNK: the chunk loop is unrolled, the whole tile pass is straight-line

  struct Tile { int b, co0, oy0, ox0; };
  auto decode = [&](int t) __attribute__((always_inline)) {
    Tile r;
    r.ox0 = (t % a.tiles_x) * 32; t /= a.tiles_x;
    r.oy0 = (t % a.tiles_y) * 4; t /= a.tiles_y;
    r.b = t / nco; r.co0 = (t % nco) * 64;
    return r;
  };
  f32x4 hv[S2_NLOAD];
  auto load_halo = [&](const Tile& t, int chunk) __attribute__((always_inline)) {      // branch-free, clamped; padding applied at the LDS store
    const float* xb = a.x + (int64_t)t.b * a.H * a.W * a.Cin + chunk * S2_KC;
#pragma unroll
    for (int i = 0; i < S2_NLOAD; ++i) {
      int e = tid + 256 * i;
      e = e < S2_NPIECE ? e : S2_NPIECE - 1;
      const int p = e >> 2, c = e & 3;
      int iy = t.oy0 * 2 - 1 + p / S2_HC, ix = t.ox0 * 2 - 1 + p % S2_HC;
      iy = iy < 0 ? 0 : (iy >= a.H ? a.H - 1 : iy);
      ix = ix < 0 ? 0 : (ix >= a.W ? a.W - 1 : ix);
      hv[i] = *reinterpret_cast<const f32x4*>(xb + (unsigned)((iy * a.W + ix) * a.Cin + 4 * c));
    }
  };
  auto store_halo = [&](const Tile& t) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < S2_NLOAD; ++i) {
      const int e = tid + 256 * i;
      if (e < S2_NPIECE) {
        const int p = e >> 2, c = e & 3;
        const int iy = t.oy0 * 2 - 1 + p / S2_HC, ix = t.ox0 * 2 - 1 + p % S2_HC;
        const bool inb = ((unsigned)iy < (unsigned)a.H) & ((unsigned)ix < (unsigned)a.W);
        f32x4 v = hv[i];
        if (!inb) v = f32x4{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(&halo[p * S2_LDP + c * 4]) = v;
      }
    }
  };
  const float* rbase = a.res ? a.res : s2_zero16;
  f32x4 bsr[4];
  {
    const float* bbase = a.bias ? a.bias : s2_zero16;
    // the cout group is fixed per tile, not per workgroup (nco > 1): bias is re-read per tile below when nco > 1; preloaded for nco == 1
#pragma unroll
    for (int q = 0; q < 4; ++q) bsr[q] = *reinterpret_cast<const f32x4*>(bbase + (ct * 32 + q * 8 + 4 * h) * a.bias_mul);
  }
  // LDS float offset of this lane's two pixels (tap (0,0), k-group 0)
  int pbase_l[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) pbase_l[m] = (((rp * 2 + m) * 2) * S2_HC + l31 * 2) * S2_LDP + 4 * h;

  int t_cur = blockIdx.x;
  if (t_cur >= a.ntiles) return;
  Tile cur = decode(t_cur);
  const int64_t tap_stride = (int64_t)(a.Cout / 32) * kgroups * 64;     // in float4 units
  auto frag_base = [&](const Tile& t) __attribute__((always_inline)) {  // weight fragments of this wave's cout tile: [tap][cout tile][k-group][lane]
    return reinterpret_cast<const f32x4*>(a.wfrag) + ((int64_t)(t.co0 / 32 + ct) * kgroups) * 64 + lane;
  };
  // rotating fragment slots: the fragment of step n of a chunk lives in slot n % 3
  f32x4 wv[3];
  {
    const f32x4* wf0 = frag_base(cur);
    wv[0] = wf0[0];
    wv[1] = wf0[64];
  }
  load_halo(cur, 0);
  for (;;) {
    const int t_nxt = t_cur + gridDim.x;
    const bool has_next = t_nxt < a.ntiles;
    const Tile nxt = decode(has_next ? t_nxt : t_cur);
    const f32x4* wf = frag_base(cur);
    const f32x4* wf_nxt = frag_base(nxt);
    if (nco > 1) {
      const float* bbase = a.bias ? a.bias : s2_zero16;
#pragma unroll
      for (int q = 0; q < 4; ++q) bsr[q] = *reinterpret_cast<const f32x4*>(bbase + (cur.co0 + ct * 32 + q * 8 + 4 * h) * a.bias_mul);
    }
    f32x16 acc[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    f32x4 rv[2][4];
#pragma unroll
    for (int chunk = 0; chunk < NCHUNK; ++chunk) {
      __syncthreads();                                 // every wave is done reading the previous halo chunk
      store_halo(cur);
      __syncthreads();
      // 18 steps (9 taps x 2 k-groups of the chunk).  Loads retire in order, so a long-latency request (halo, residual) must not sit in
      // front of a short-latency weight fragment that is needed soon: fragments run two steps ahead (slots 0 / 1 are refilled with the
      // NEXT chunk's first two fragments at steps 16 / 17), and the next chunk's halo -- on the last chunk the next TILE's, behind the
      // tile's residual rows -- is requested after all of them.  Scheduling barriers keep that order (hipcc otherwise collapses the
      // rotation into load / s_waitcnt vmcnt(0) / multiply).
      const bool last = chunk + 1 == NCHUNK;             // compile time once the chunk loop is unrolled
      const int nc = last ? 0 : chunk + 1;
#pragma unroll
      for (int n = 0; n < 18; ++n) {
        if (n + 2 < 18) {
          const int tap = (n + 2) >> 1, kg = (n + 2) & 1;
          wv[(n + 2) % 3] = wf[tap * tap_stride + (int64_t)(chunk * 2 + kg) * 64];
        } else if (n == 16) {
          wv[0] = (last ? wf_nxt : wf)[(int64_t)(nc * 2 + 0) * 64];
        } else {
          wv[1] = (last ? wf_nxt : wf)[(int64_t)(nc * 2 + 1) * 64];
          if (last) {                                  // (compile time) residual rows of this tile
#pragma unroll
            for (int m = 0; m < 2; ++m) {
              const int oy = cur.oy0 + rp * 2 + m, ox = cur.ox0 + l31;
              const bool ok = oy < a.Ho && ox < a.Wo;
              const int64_t pix = (int64_t)(ok ? oy : 0) * a.Wo + (ok ? ox : 0);
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const int cq = cur.co0 + ct * 32 + q * 8 + 4 * h;
                rv[m][q] = *reinterpret_cast<const f32x4*>(rbase + (((int64_t)cur.b * a.Ho * a.Wo + pix) * a.Cout + cq) * a.res_mul);
              }
            }
          }
          load_halo(last ? nxt : cur, nc);
        }
        __builtin_amdgcn_sched_barrier(0);
        const int tap = n >> 1, kg = n & 1;
        const int toff = ((tap / 3) * S2_HC + (tap % 3)) * S2_LDP + kg * 8;
        const f32x4 p0 = *reinterpret_cast<const f32x4*>(&halo[pbase_l[0] + toff]);
        const f32x4 p1 = *reinterpret_cast<const f32x4*>(&halo[pbase_l[1] + toff]);
        const f32x4 w = wv[n % 3];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[0] = cmr_mfma32(w[j], p0[j], acc[0]);
          acc[1] = cmr_mfma32(w[j], p1[j], acc[1]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // ---- epilogue: lane = pixel (row rp*2+m, column l31); register quad q = couts ct*32 + 8q + 4h .. +3
    f32x4 ov[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float u = acc[m][4 * q + e] + bsr[q][e] + rv[m][q][e];
          v[e] = u > 0.f ? u : u * a.slope;
        }
        ov[m][q] = v;
      }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int q = 0; q < 4; ++q) cmr_pin(ov[m][q]);
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int oy = cur.oy0 + rp * 2 + m, ox = cur.ox0 + l31;
      if (oy < a.Ho && ox < a.Wo) {
        float* yp = a.y + (((int64_t)cur.b * a.Ho + oy) * a.Wo + ox) * a.Cout + cur.co0 + ct * 32 + 4 * h;
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(yp + q * 8) = ov[m][q];
      }
    }
    if (!has_next) break;
    t_cur = t_nxt;
    cur = nxt;
  }
}

}  // namespace

extern "C" int cmr_conv3x3_s2_nhwc_f32(const float* x, int B, int H, int W, int Cin, const float* wfrag, const float* bias, const float* res,
                                       const float* post, float* y, int Cout, float slope, hipStream_t stream) {
  CMR_REQUIRE(x && wfrag && y && B > 0 && H > 0 && W > 0 && Cout >= 64 && Cout % 64 == 0);
  if (Cin != 64 || post) return CMR_EUNSUPPORTED;              // instantiated for the model's 64-channel down-sampling blocks (no table operand); others: cmr_conv3x3_nhwc_f32
  CMR_REQUIRE(cmr_aligned16(x) && cmr_aligned16(wfrag) && cmr_aligned16(y) && (!bias || cmr_aligned16(bias)) && (!res || cmr_aligned16(res)) &&
              (!post || cmr_aligned16(post)));
  CMR_REQUIRE((int64_t)B * H * W * Cin < 0x7fffffff);
  S2Args a{x, B, H, W, Cin, wfrag, bias, res, post, y, (H - 1) / 2 + 1, (W - 1) / 2 + 1, Cout, slope, 0, 0, 0, res ? 1 : 0, bias ? 1 : 0};
  a.tiles_x = (a.Wo + 31) / 32;
  a.tiles_y = (a.Ho + 3) / 4;
  const int64_t ntiles = (int64_t)B * (Cout / 64) * a.tiles_x * a.tiles_y;
  CMR_REQUIRE(ntiles < 0x7fffffff);
  a.ntiles = (int)ntiles;
  const int grid = ntiles < 512 ? (int)ntiles : 512;   // two resident workgroups per CU (198 VGPRs)
  hipLaunchKernelGGL(conv3x3_s2_kernel<4>, dim3(grid), dim3(256), 0, stream, a);
  return cmr_launch_status();
}
