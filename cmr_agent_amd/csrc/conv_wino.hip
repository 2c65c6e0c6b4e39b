// Stride-1 3x3 convolution by Winograd F(2x2, 3x3) on fp32 MFMA, fully fused (input transform, the 16
// position GEMMs, output transform, bias / residual / LeakyReLU / table / optional 2x2 average pool).
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A        16 multiplies per 2x2 outputs instead of 36
//
// The direct implicit-GEMM kernel (conv.hip) runs at ~75 % of the fp32 MFMA peak, so the remaining lever
// is the multiply count itself.  U = G g G^T is precomputed per layer ([16][Cout][Cin], host side).
//
// Workgroup = 256 threads = 4 waves; output tile 8 rows x 16 cols = 4 x 8 Winograd tiles = ONE 32-wide
// MFMA dimension (lane = Winograd tile); 64 output channels per workgroup (2 x 32).
// Per 32-channel chunk of the input:
//   1. the 10 x 18 pixel halo chunk goes global -> registers -> LDS (double buffered: the next chunk's
//      loads are in flight during this chunk's GEMMs; ONE barrier per chunk);
//   2. wave w multiplies the 4 positions (w, 0..3).  It never materialises V = B^T d B: per k-group a lane
//      reads the two input rows its position row needs (8 x ds_read_b128 of ITS tile's patch), forms the 4
//      transformed fragments with 8 float4 adds, and feeds 32 MFMAs.  A operand = U fragments straight
//      from global / L2, prefetched one k-group ahead.
// Epilogue: T[w][b] = sum_j M[w][j] A[j][b] in registers, exchanged through LDS, then wave q finishes
// Y[a][b] = sum_i A^T[a][i] T[i][b] for register quad q (4 consecutive channels -> float4 stores).
#include "cmr_common.h"

namespace {

struct WinoArgs {
  const float* x; int B, H, W, Cin;
  const float* u;      // G g G^T (BN folded) as A fragments [16][Cout/32][Cin/8][64][4]
  const float* bias; const float* res; const float* post;
  float* y; int Cout; float slope; int pool;
};

constexpr int WT_TH = 8, WT_TW = 16;          // output tile
constexpr int WT_HR = 10, WT_HC = 18;         // halo
constexpr int WT_KC = 32, WT_LDP = 36;        // channels per chunk, padded LDS row (floats)
constexpr int WT_C4 = WT_KC / 4;
constexpr int WT_HALO_F4 = WT_HR * WT_HC * WT_C4;                 // 1440
constexpr int WT_HL = (WT_HALO_F4 + 255) / 256;                   // 6
// LDS image of the halo: row pitch 20 pixels, columns de-interleaved by parity (pixel x sits at (x&1)*10 + x/2).
// The Winograd tiles of a wave start at even columns (stride 2); with the plain [row][col] image every
// ds_read_b128 of the patch was a 4-way bank conflict (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.56).  In
// this image lane tx reads pixel slot tx + const, i.e. a stride of one 36-float slot: conflict free.
constexpr int WT_NUM_CU = 256;
constexpr int WT_PITCH = 20;
constexpr int WT_HALO_FLOATS = WT_HR * WT_PITCH * WT_LDP;         // 7200
constexpr int WT_SMEM_FLOATS = 16384;         // 64 KB: two halo buffers (14400) in the K loop, T (16384) after it
__device__ __forceinline__ int wt_slot(int py, int px) { return py * WT_PITCH + (px & 1) * (WT_PITCH / 2) + (px >> 1); }

__global__ __launch_bounds__(256, 2) void conv3x3_wino_kernel(const WinoArgs a, const int tiles_x, const int tiles_y) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // uniform: lets the U bases live in SGPRs
  const int h = lane >> 5, l31 = lane & 31;
  const int nco = a.Cout / 64;
  int t = blockIdx.x;
  const int ox0 = (t % tiles_x) * WT_TW; t /= tiles_x;
  const int oy0 = (t % tiles_y) * WT_TH; t /= tiles_y;
  const int b = t / nco, co0 = (t % nco) * 64;
  const int iy0 = oy0 - 1, ix0 = ox0 - 1;
  const int nchunk = a.Cin / WT_KC;
  const float* xb = a.x + (int64_t)b * a.H * a.W * a.Cin;

  f32x16 acc[4][2];                                 // [position column j][cout tile]
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][n][r] = 0.f;

  // halo chunk: global -> registers (batched, clamped addresses) -> LDS (zero padding applied there)
  // (per-lane 32-bit offsets from uniform bases: the loads take the saddr + voffset form, one VGPR per address)
  f32x4 hv[WT_HL];
  unsigned hoff[WT_HL];
#pragma unroll
  for (int i = 0; i < WT_HL; ++i) {
    int e = tid + 256 * i;
    if (e >= WT_HALO_F4) e = WT_HALO_F4 - 1;
    const int p = e / WT_C4, c = e % WT_C4;
    int iy = iy0 + p / WT_HC, ix = ix0 + p % WT_HC;
    iy = iy < 0 ? 0 : (iy >= a.H ? a.H - 1 : iy);
    ix = ix < 0 ? 0 : (ix >= a.W ? a.W - 1 : ix);
    hoff[i] = (unsigned)((iy * a.W + ix) * a.Cin + c * 4);
  }
  auto load_halo = [&](int chunk) {
    const float* xc = xb + chunk * WT_KC;
#pragma unroll
    for (int i = 0; i < WT_HL; ++i) hv[i] = *reinterpret_cast<const f32x4*>(xc + hoff[i]);
  };
  auto store_halo = [&](float* halo) {
#pragma unroll
    for (int i = 0; i < WT_HL; ++i) {
      const int e = tid + 256 * i;
      if (e < WT_HALO_F4) {
        const int p = e / WT_C4, c = e % WT_C4;
        const int iy = iy0 + p / WT_HC, ix = ix0 + p % WT_HC;
        const bool inb = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        *reinterpret_cast<f32x4*>(&halo[wt_slot(p / WT_HC, p % WT_HC) * WT_LDP + c * 4]) = inb ? hv[i] : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };

  // This wave multiplies the 4 positions (i = wave, j = 0..3).  Row i of B^T d needs two input rows:
  //   i=0: d0 - d2   i=1: d1 + d2   i=2: d2 - d1   i=3: d1 - d3       (ra, rb, sign below)
  const int ra = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
  const int rb = wave == 3 ? 3 : (wave == 2 ? 1 : 2);
  const float sgn = wave == 1 ? 1.f : -1.f;
  const int ty = l31 >> 3, tx = l31 & 7;
  const int pa = ((2 * ty + ra) * WT_PITCH + tx) * WT_LDP + 4 * h;      // LDS float offsets of the two patch rows (column 2*tx)
  const int pb = ((2 * ty + rb) * WT_PITCH + tx) * WT_LDP + 4 * h;
  // U is stored as ready-made A fragments [pos][Cout/32][Cin/8][64 lanes][4]: a wave load is 1 KB contiguous
  const int64_t utile = (int64_t)(a.Cin / 8) * 256; // stride between cout tiles
  const int64_t upos = (int64_t)a.Cout * a.Cin;     // stride between positions
  const float* ub = a.u + (int64_t)(4 * wave) * upos + (co0 / 32) * utile;   // position (wave, 0), first cout tile (uniform)
  const unsigned uoff = (unsigned)(lane * 4);

  // Loads retire in order (vmcnt), so a U fragment queued behind a halo load waits for HBM.  Order of issue per
  // chunk: U(kg1) | U(kg2) | U(kg3), halo(next chunk) | U(next chunk, kg0): every wait on U only has older U
  // loads ahead of it, and the halo has two k-groups of MFMAs (plus the barrier) to arrive.
  // Two workgroups share a CU (one wave of each per SIMD).  Launched together they stay in lock step -- both in
  // the load prologue, both in the MFMA loop, both in the epilogue -- and the MFMA pipe idles for every non-MFMA
  // phase.  The first generation of workgroups on odd wave slots starts half a period late; equal-length
  // workgroups preserve that offset, so one wave's loads / epilogue run under the other's MFMAs from then on.
  if (blockIdx.x < 2 * WT_NUM_CU && (__builtin_amdgcn_s_getreg((3 << 11) | 4) & 1)) {     // HW_ID.WAVE_ID bit 0
    for (int i = 0; i < 1 + nchunk; ++i) __builtin_amdgcn_s_sleep(127);
  }
  load_halo(0);
  f32x4 wc[4][2], wn[4][2];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int n = 0; n < 2; ++n) wc[j][n] = *reinterpret_cast<const f32x4*>(ub + j * upos + n * utile + uoff);
  int buf = 0;
  for (int chunk = 0; chunk < nchunk; ++chunk) {
    float* halo = smem + buf * WT_HALO_FLOATS;
    store_halo(halo);
    __syncthreads();                                // halo[buf] visible; everybody is past the GEMMs that read halo[buf] two chunks ago
    const float* uc = ub + chunk * (WT_KC / 8) * 256;
    const float* un = ub + (chunk + 1 < nchunk ? chunk + 1 : 0) * (WT_KC / 8) * 256;   // last chunk: a harmless in-bounds re-read
#pragma unroll
    for (int kg = 0; kg < WT_KC / 8; ++kg) {
      const float* up = kg + 1 < WT_KC / 8 ? uc + (kg + 1) * 256 : un;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int n = 0; n < 2; ++n) wn[j][n] = *reinterpret_cast<const f32x4*>(up + j * upos + n * utile + uoff);
      if (kg == WT_KC / 8 - 2 && chunk + 1 < nchunk) load_halo(chunk + 1);
      // on-the-fly input transform of this lane's tile: t[c] = d[ra][c] +- d[rb][c], then the 4 columns j
      f32x4 tc[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        constexpr int WT_COFS[4] = {0, (WT_PITCH / 2) * WT_LDP, WT_LDP, (WT_PITCH / 2 + 1) * WT_LDP};   // column 2*tx + c
        const f32x4 da = *reinterpret_cast<const f32x4*>(&halo[pa + WT_COFS[c] + kg * 8]);
        const f32x4 db = *reinterpret_cast<const f32x4*>(&halo[pb + WT_COFS[c] + kg * 8]);
        tc[c] = da + sgn * db;
      }
      f32x4 vf[4];
      vf[0] = tc[0] - tc[2];
      vf[1] = tc[1] + tc[2];
      vf[2] = tc[2] - tc[1];
      vf[3] = tc[1] - tc[3];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc[j][0] = cmr_mfma32(wc[j][0][e], vf[j][e], acc[j][0]);
          acc[j][1] = cmr_mfma32(wc[j][1][e], vf[j][e], acc[j][1]);
        }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int n = 0; n < 2; ++n) wc[j][n] = wn[j][n];
    }
    buf ^= 1;
  }
  __syncthreads();                                  // all waves done with the halo buffers before they are reused for T

  // ---- output transform, stage 1 (registers): T[w][b] = sum_j M[w][j] A[j][b]
  float* Ts = smem;                                 // [(w*2 + b)*2 + n][16][64]
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float t0 = (acc[0][n][r] + acc[1][n][r]) + acc[2][n][r];
      const float t1 = (acc[1][n][r] - acc[2][n][r]) - acc[3][n][r];
      Ts[(((wave * 2 + 0) * 2 + n) * 16 + r) * 64 + lane] = t0;
      Ts[(((wave * 2 + 1) * 2 + n) * 16 + r) * 64 + lane] = t1;
    }
  // ---- stage 2: wave q finishes the 4 outputs (a, b) of register quad q of both cout tiles.
  // Every epilogue operand (bias / residual / table) is requested HERE, before the barrier, in one batch of
  // branch-free loads (absent operands read a valid dummy address and are masked after the wait): a
  // conditional load per use costs one exposed L2/HBM round trip each, ~20 of them per workgroup.
  const int q = wave;
  const int Ho = a.H, Wo = a.W;
  const int cq = co0 + 8 * q + 4 * h;               // + 32 n
  f32x4 bs[2], rs[2][2][2], ps[2][2][2];
#pragma unroll
  for (int n = 0; n < 2; ++n) bs[n] = *reinterpret_cast<const f32x4*>(a.bias ? a.bias + cq + 32 * n : a.u);
  if (a.pool == 1) {
#pragma unroll
    for (int aa = 0; aa < 2; ++aa)
#pragma unroll
      for (int bb = 0; bb < 2; ++bb) {
        int oy = oy0 + 2 * ty + aa, ox = ox0 + 2 * tx + bb;
        oy = oy < Ho ? oy : Ho - 1;
        ox = ox < Wo ? ox : Wo - 1;
        const int64_t pix = (int64_t)oy * Wo + ox;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          rs[n][aa][bb] = *reinterpret_cast<const f32x4*>(a.res ? a.res + ((int64_t)b * Ho * Wo + pix) * a.Cout + cq + 32 * n : a.u);
          ps[n][aa][bb] = *reinterpret_cast<const f32x4*>(a.post ? a.post + pix * a.Cout + cq + 32 * n : a.u);
        }
      }
  }
  __syncthreads();
  f32x4 yv[2][2][2];                                // [n][a][b]
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int bb = 0; bb < 2; ++bb) {
      f32x4 tw[4];
#pragma unroll
      for (int w = 0; w < 4; ++w)
#pragma unroll
        for (int e = 0; e < 4; ++e) tw[w][e] = Ts[(((w * 2 + bb) * 2 + n) * 16 + 4 * q + e) * 64 + lane];
      yv[n][0][bb] = (tw[0] + tw[1]) + tw[2];
      yv[n][1][bb] = (tw[1] - tw[2]) - tw[3];
    }
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int n = 0; n < 2; ++n) bs[n] = a.bias ? bs[n] : zero4;
  if (a.pool == 2) {
    const int py = (oy0 >> 1) + ty, px = (ox0 >> 1) + tx;
    const int hp2 = Ho >> 1, wp2 = Wo >> 1;
    if (py < hp2 && px < wp2) {
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        f32x4 s = zero4;
#pragma unroll
        for (int aa = 0; aa < 2; ++aa)
#pragma unroll
          for (int bb = 0; bb < 2; ++bb)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float v = yv[n][aa][bb][e] + bs[n][e];
              s[e] += v > 0.f ? v : v * a.slope;
            }
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] *= 0.25f;
        *reinterpret_cast<f32x4*>(a.y + (((int64_t)b * hp2 + py) * wp2 + px) * a.Cout + cq + 32 * n) = s;
      }
    }
    return;
  }
#pragma unroll
  for (int aa = 0; aa < 2; ++aa)
#pragma unroll
    for (int bb = 0; bb < 2; ++bb) {
      const int oy = oy0 + 2 * ty + aa, ox = ox0 + 2 * tx + bb;
      const bool inb = oy < Ho && ox < Wo;
      const int64_t pix = (int64_t)oy * Wo + ox;
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        f32x4 v = yv[n][aa][bb] + bs[n];
        if (a.res) v += rs[n][aa][bb];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * a.slope;
        if (a.post) v += ps[n][aa][bb];
        if (inb) *reinterpret_cast<f32x4*>(a.y + ((int64_t)b * Ho * Wo + pix) * a.Cout + cq + 32 * n) = v;
      }
    }
}

}  // namespace

extern "C" int cmr_conv3x3_wino_nhwc_f32(const float* x, int B, int H, int W, int Cin, const float* u, const float* bias,
                                         const float* res, const float* post, float* y, int Cout, float slope, int pool,
                                         hipStream_t stream) {
  CMR_REQUIRE(x && u && y && B > 0 && H > 0 && W > 0 && Cin % 32 == 0 && Cin >= 32 && Cout % 64 == 0 && Cout >= 64);
  CMR_REQUIRE(cmr_aligned16(x) && cmr_aligned16(u) && cmr_aligned16(y) && (!bias || cmr_aligned16(bias)) &&
              (!res || cmr_aligned16(res)) && (!post || cmr_aligned16(post)));
  CMR_REQUIRE(pool == 1 || (pool == 2 && !res && !post));
  const int tiles_x = (W + WT_TW - 1) / WT_TW, tiles_y = (H + WT_TH - 1) / WT_TH;
  const int64_t ntiles = (int64_t)tiles_x * tiles_y * B * (Cout / 64);
  CMR_REQUIRE(ntiles < 0x7fffffff);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wino_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            WT_SMEM_FLOATS * (int)sizeof(float)) != hipSuccess)
      return CMR_ELAUNCH;
    attr_set = true;
  }
  const WinoArgs a{x, B, H, W, Cin, u, bias, res, post, y, Cout, slope, pool};
  hipLaunchKernelGGL(conv3x3_wino_kernel, dim3((unsigned)ntiles), dim3(256), WT_SMEM_FLOATS * sizeof(float), stream, a,
                     tiles_x, tiles_y);
  return cmr_launch_status();
}
