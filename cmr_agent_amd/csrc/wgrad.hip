// Weight-gradient contractions of the agent update (reference Train_Agent.py:304 `loss.backward()` through
// the nn.Conv2d / nn.Conv1d / nn.Linear layers of CMRAgent.py:25-86) on the fp32 matrix cores, and the per-step weight
// repacking that lets the data-gradient convolutions reuse the forward kernels.
//
// A weight gradient contracts over the PIXELS / ROWS of the minibatch (K = 10^5 .. 10^6) into a tiny output
// (Cout x Cin x 9 or n x k): every wave keeps its output tiles in MFMA accumulators for the whole launch and streams its
// slice of the rows through v_mfma_f32_32x32x2_f32 (A = dY^T: lane l supplies dY[row k = l>>5][channel i = l&31],
// B = X: X[row k][channel j = l&31]; both are 128-byte coalesced dword loads per half wave, no LDS, no barrier);
// per-wave partial outputs go to a workspace and a second kernel sums them in a fixed order (deterministic: the
// data-parallel ranks must produce bit-identical buckets from identical inputs).
#include "cmr_common.h"

namespace {

// ------------------------------------------------------------------------------------------------------------------
// conv3x3 (stride 1, pad 1, NHWC) weight gradient:  dW[co][ci][ky][kx] = sum_{b,y,x} dY[b,y,x,co] X[b,y+ky-1,x+kx-1,ci]
// grid (slices, Cout/32); workgroup = 4 waves; wave = (ci tile, pixel sub-slice); 9 accumulator tiles (one per tap).
// One MFMA step consumes the pixels (2q, 2q+1) of the flattened [B*H*W] map (each lane tracks the coordinates of its own
// pixel, so a pair may straddle a row or an image).  Loads run four steps ahead of the MFMAs (straight-line, clamped addresses,
// masks applied at consumption: see DESIGN.md "hipcc rules").
// ------------------------------------------------------------------------------------------------------------------
template <int NCI>
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, int B, int H, int W,
                                                                int Cin, int Cout, float* __restrict__ part) {
  constexpr int NSPLIT = 4 / NCI;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int ci_t = wave % NCI, split = wave / NCI;
  const int co_t = blockIdx.y;
  const int npix = B * H * W;
  const int npairs = (npix + 1) / 2;
  const int per_blk = (npairs + gridDim.x - 1) / gridDim.x;
  const int q0 = min((int)blockIdx.x * per_blk, npairs), q1 = min(q0 + per_blk, npairs);
  const int per_w = (q1 - q0 + NSPLIT - 1) / NSPLIT;
  const int w0 = min(q0 + split * per_w, q1), w1 = min(w0 + per_w, q1);

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const float* xb = x + ci_t * 32 + l31;
  const float* db = dy + co_t * 32 + l31;
  const int last_pix = B * H * W - 1;
  const int cin_sh = 31 - __builtin_clz(Cin), cout_sh = 31 - __builtin_clz(Cout);   // channel counts are powers of two (host check)

  // coordinates of this lane's pixel of pair q: linear index 2q + h, (x, y) only for the padding masks
  int p = 2 * w0 + h;
  int xx = p % W, yy = (p / W) % H;

  // Integer work per step is kept to adds, clamps and shifts: the first version recomputed (b H + y) W + x and multiplied by
  // the channel count for every tap -- 20 v_mul_lo_u32 + 12 v_mad_u64_u32 (quarter rate) per step, ~900 issue cycles
  // against 576 cycles of MFMA, i.e. the kernel was VALU-bound at 48 % of its matrix bound whatever the prefetch depth.
  // Addresses: any linear pixel index clamped to [0, last_pix] is readable; whether a tap lies inside the image is decided
  // by the masks (3 row tests x 3 column tests).
  // Column reuse: tap (ky, kx = -1) of the pixel p is the linear pixel p + (ky - 1) W - 1 = (p - 2) + (ky - 1) W + 1, i.e. tap
  // (ky, kx = +1) of the SAME lane one step earlier (a step advances every lane by two pixels).  So a step loads only the columns
  // kx = 0 and kx = +1 of the three rows (6 loads + dY instead of 9 + 1) and multiplies the previous step's kx = +1 registers
  // again, under THIS step's padding mask.  With ten dword loads per nine MFMAs the texture path was the ceiling: a wave load of
  // 64 dwords costs ~16 address cycles whatever it carries, four steps per 576 MFMA cycles and CU = 640 cycles of it.
  int toff[6];
#pragma unroll
  for (int t = 0; t < 6; ++t) toff[t] = (t / 2 - 1) * W + (t % 2);        // (ky, kx = 0 | +1)
  auto load = [&](float& a, float (&v)[6], int base) {
    a = db[(int64_t)(min(max(base, 0), last_pix) << cout_sh)];
#pragma unroll
    for (int t = 0; t < 6; ++t) v[t] = xb[(int64_t)(min(max(base + toff[t], 0), last_pix) << cin_sh)];
  };

  constexpr int NSET = 6;                         // consumed now | the previous step's (its kx = +1 columns are this step's kx = -1) |
                                                  // three loaded earlier | being loaded: four steps (28 loads, 2 300 MFMA cycles) in flight
  float sa[NSET], sv[NSET][6];
  int sp[NSET], sx[NSET], sy[NSET];
  auto advance = [&](int& xq, int& yq) {
    xq += 2;
    if (xq >= W) { xq -= W; ++yq; }              // W >= 2: one wrap per step
    if (yq >= H) yq -= H;
  };
  // set NSET - 1 plays "the step before the first": pixel p - 2 (clamped address, only its kx = +1 columns are read)
  sp[NSET - 1] = p - 2; sx[NSET - 1] = 0; sy[NSET - 1] = 0;
  sp[0] = p; sx[0] = xx; sy[0] = yy;
#pragma unroll
  for (int i = 1; i < NSET - 2; ++i) {
    sp[i] = sp[i - 1] + 2; sx[i] = sx[i - 1]; sy[i] = sy[i - 1];
    advance(sx[i], sy[i]);
  }
  if (w0 < w1) {
    load(sa[NSET - 1], sv[NSET - 1], sp[NSET - 1]);
#pragma unroll
    for (int i = 0; i < NSET - 2; ++i) load(sa[i], sv[i], sp[i]);   // past the end of the slice: valid, unused addresses
  }
  // Register sets rotate through the roles and nothing ever COPIES a loaded value: a rotation by assignment (cur = nxt) reads the
  // load's destination and drags an s_waitcnt vmcnt(0) to the top of every iteration.  The padding mask is applied when a set is
  // consumed (into fresh registers), for the same reason.
#define CMR_WG_STEP(C, VALID)                                                              \
  {                                                                                        \
    constexpr int L = (C + NSET - 2) % NSET, P = (C + NSET - 3) % NSET, M1 = (C + NSET - 1) % NSET; \
    sp[L] = sp[P] + 2; sx[L] = sx[P]; sy[L] = sy[P];                                       \
    advance(sx[L], sy[L]);                                                                 \
    load(sa[L], sv[L], sp[L]);                                                             \
    __builtin_amdgcn_sched_barrier(0); /* the loads are issued BEFORE this step's MFMAs */ \
    const bool live = (VALID) && sp[C] <= last_pix; /* odd pixel count: second lane half of the last pair */ \
    const float av = live ? sa[C] : 0.f;                                                   \
    const bool oy[3] = {sy[C] > 0, true, sy[C] < H - 1};                                   \
    const bool oxl = sx[C] > 0, oxr = sx[C] < W - 1;                                       \
    _Pragma("unroll") for (int ky = 0; ky < 3; ++ky) {                                     \
      const float vl = (oy[ky] && oxl) ? sv[M1][2 * ky + 1] : 0.f;                         \
      const float vc = oy[ky] ? sv[C][2 * ky] : 0.f;                                       \
      const float vr = (oy[ky] && oxr) ? sv[C][2 * ky + 1] : 0.f;                          \
      acc[3 * ky] = cmr_mfma32(av, vl, acc[3 * ky]);                                       \
      acc[3 * ky + 1] = cmr_mfma32(av, vc, acc[3 * ky + 1]);                               \
      acc[3 * ky + 2] = cmr_mfma32(av, vr, acc[3 * ky + 2]);                               \
    }                                                                                      \
  }
  // straight-line groups of NSET steps (one exit: extra exits made hipcc spill the accumulators); the last group's steps
  // past the end of the slice multiply by a zero dY
  for (int q = w0; q < w1; q += NSET) {
    CMR_WG_STEP(0, true)
    CMR_WG_STEP(1, q + 1 < w1)
    CMR_WG_STEP(2, q + 2 < w1)
    CMR_WG_STEP(3, q + 3 < w1)
    CMR_WG_STEP(4, q + 4 < w1)
    CMR_WG_STEP(5, q + 5 < w1)
  }
#undef CMR_WG_STEP

  float* out = part + ((int64_t)(blockIdx.x * NSPLIT + split) * 9) * Cout * Cin;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = cmr_mfma_row(r, lane);
      out[((int64_t)t * Cout + co_t * 32 + row) * Cin + ci_t * 32 + l31] = acc[t][r];
    }
}

// ------------------------------------------------------------------------------------------------------------------
// conv3x3 with STRIDE 2 (pad 1, NHWC; the down-sampling convolutions of a ResidualBlock, ImageResNet.py:9-14, :24-27):
//   dW[co][ci][ky][kx] = sum_{b,oy,ox} dY[b,oy,ox,co] X[b, 2 oy + ky - 1, 2 ox + kx - 1, ci]
// Round 2 obtained it from the stride-1 kernel on the zero-inserted gradient (cmr_zero_insert2_f32): a contraction over ALL input pixels of
// which three quarters multiply zeros -- 252 GFLOP instead of 63 for the 352x1216 64 -> 64 layer.  Here the contraction runs over the OUTPUT
// pixels: same wave / accumulator layout as conv3x3_wgrad_kernel (A = dY^T, B = X at the tap's input pixel, lane = channel), one dY load and
// nine X loads per step (consecutive output pixels of a lane are four input columns apart: no column reuse), loads three steps ahead.
// Padding: only the top row / left column can fall outside (H, W even: 2 oy + 1 <= H - 1).
// ------------------------------------------------------------------------------------------------------------------
template <int NCI>
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_s2_kernel(const float* __restrict__ x, const float* __restrict__ dy, int B, int H, int W,
                                                                   int Cin, int Cout, float* __restrict__ part) {
  constexpr int NSPLIT = 4 / NCI;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int ci_t = wave % NCI, split = wave / NCI;
  const int co_t = blockIdx.y;
  const int Ho = H / 2, Wo = W / 2;
  const int npix = B * Ho * Wo;                    // output pixels
  const int npairs = (npix + 1) / 2;
  const int per_blk = (npairs + gridDim.x - 1) / gridDim.x;
  const int q0 = min((int)blockIdx.x * per_blk, npairs), q1 = min(q0 + per_blk, npairs);
  const int per_w = (q1 - q0 + NSPLIT - 1) / NSPLIT;
  const int w0 = min(q0 + split * per_w, q1), w1 = min(w0 + per_w, q1);

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const float* xb = x + ci_t * 32 + l31;
  const float* db = dy + co_t * 32 + l31;
  const int last_out = npix - 1, last_in = B * H * W - 1;
  const int cin_sh = 31 - __builtin_clz(Cin), cout_sh = 31 - __builtin_clz(Cout);

  // this lane's output pixel of pair q: linear index 2 q + h -> (b, oy, ox); its window's centre in the input: (b H + 2 oy) W + 2 ox
  int p = 2 * w0 + h;
  int ox = p % Wo, oyb = p / Wo;                   // oyb = b Ho + oy: the input row of the centre is 2 oyb (H = 2 Ho)
  int toff[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) toff[t] = (t / 3 - 1) * W + (t % 3 - 1);
  auto load = [&](float& a, float (&v)[9], int pout, int pin) {
    a = db[(int64_t)(min(pout, last_out) << cout_sh)];
#pragma unroll
    for (int t = 0; t < 9; ++t) v[t] = xb[(int64_t)(min(max(pin + toff[t], 0), last_in) << cin_sh)];
  };
  constexpr int NSET = 4;                          // consumed now | two loaded earlier | being loaded
  float sa[NSET], sv[NSET][9];
  int sp[NSET], sx[NSET], sr[NSET];               // output pixel, ox, b Ho + oy
  auto advance = [&](int& xq, int& rq) {
    xq += 2;
    if (xq >= Wo) { xq -= Wo; ++rq; }             // Wo >= 2: one wrap per step
  };
  sp[0] = p; sx[0] = ox; sr[0] = oyb;
#pragma unroll
  for (int i = 1; i < NSET - 1; ++i) {
    sp[i] = sp[i - 1] + 2; sx[i] = sx[i - 1]; sr[i] = sr[i - 1];
    advance(sx[i], sr[i]);
  }
  if (w0 < w1) {
#pragma unroll
    for (int i = 0; i < NSET - 1; ++i) load(sa[i], sv[i], sp[i], (2 * sr[i]) * W + 2 * sx[i]);
  }
#define CMR_WG2_STEP(C, VALID)                                                             \
  {                                                                                        \
    constexpr int L = (C + NSET - 1) % NSET, P = (C + NSET - 2) % NSET;                    \
    sp[L] = sp[P] + 2; sx[L] = sx[P]; sr[L] = sr[P];                                       \
    advance(sx[L], sr[L]);                                                                 \
    load(sa[L], sv[L], sp[L], (2 * sr[L]) * W + 2 * sx[L]);                                \
    __builtin_amdgcn_sched_barrier(0); /* the loads are issued BEFORE this step's MFMAs */ \
    const bool live = (VALID) && sp[C] <= last_out;                                        \
    const float av = live ? sa[C] : 0.f;                                                   \
    const bool top = (sr[C] % Ho) > 0, left = sx[C] > 0;                                   \
    _Pragma("unroll") for (int ky = 0; ky < 3; ++ky) {                                     \
      const bool oky = ky > 0 || top;                                                      \
      const float vl = (oky && left) ? sv[C][3 * ky] : 0.f;                                \
      const float vc = oky ? sv[C][3 * ky + 1] : 0.f;                                      \
      const float vr = oky ? sv[C][3 * ky + 2] : 0.f;                                      \
      acc[3 * ky] = cmr_mfma32(av, vl, acc[3 * ky]);                                       \
      acc[3 * ky + 1] = cmr_mfma32(av, vc, acc[3 * ky + 1]);                               \
      acc[3 * ky + 2] = cmr_mfma32(av, vr, acc[3 * ky + 2]);                               \
    }                                                                                      \
  }
  for (int q = w0; q < w1; q += NSET) {
    CMR_WG2_STEP(0, true)
    CMR_WG2_STEP(1, q + 1 < w1)
    CMR_WG2_STEP(2, q + 2 < w1)
    CMR_WG2_STEP(3, q + 3 < w1)
  }
#undef CMR_WG2_STEP
  float* out = part + ((int64_t)(blockIdx.x * NSPLIT + split) * 9) * Cout * Cin;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = cmr_mfma_row(r, lane);
      out[((int64_t)t * Cout + co_t * 32 + row) * Cin + ci_t * 32 + l31] = acc[t][r];
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The same contraction with the operands staged in LDS (round 3).  conv3x3_wgrad_kernel fetches every operand of every MFMA with a
// dword load (7 per 9 MFMAs after the column reuse) and every tap of a pixel comes back from L2 once per tap row: 0.44-0.46 of the
// fp32 matrix peak whatever the prefetch depth.  Here a workgroup walks DOWN a 32-pixel-wide column strip of one image with a ring of
// four input rows in LDS (rows y-1, y, y+1 in use, row y+2 arriving): every input pixel is read from memory once per strip (plus the
// two halo columns), as float4, and the nine taps are nine LDS addresses.  wave = ci tile (x pair split when Cin = 64), nine
// accumulator tiles per wave for the workgroup's cout tile, as before; workgroups are persistent over strips (static stride) and
// write ONE partial per wave at the end, summed by conv3x3_wgrad_reduce_kernel in the same fixed order.
//   LDS: X ring [4][34][Cin + 32] floats (pixel stride Cin + 32: the two lane halves of an operand read -- pixels 2q and 2q + 1 --
//   fall 32 banks apart), dY [2][32][32] floats.  One __syncthreads per image row.
// ------------------------------------------------------------------------------------------------------------------
template <int NCI>
__global__ __launch_bounds__(256) void conv3x3_wgrad_lds_kernel(const float* __restrict__ x, const float* __restrict__ dy, int B, int H, int W,
                                                                int Cout, int rps, float* __restrict__ part) {
  constexpr int CIN = 32 * NCI, XS = CIN + 32, TW = 32, HC = TW + 2, NSPLIT = 4 / NCI;
  constexpr int C4 = CIN / 4, NPIECE = HC * C4, NL = (NPIECE + 255) / 256;
  constexpr int NPAIR = 16 / NSPLIT;                   // pixel pairs of a row tile per wave
  extern __shared__ __attribute__((aligned(16))) float wg_smem[];
  float* Xr = wg_smem;                                  // [4][HC][XS]
  float* Dr = wg_smem + 4 * HC * XS;                    // [2][TW][32]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  const int ci_t = wave % NCI, split = wave / NCI;
  const int co_t = blockIdx.y;
  const int ntx = (W + TW - 1) / TW, nys = (H + rps - 1) / rps;
  const int nstrips = B * ntx * nys;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // staging roles of this thread (fixed): NL float4 pieces of an input row (halo pixel px, channels 4 c4 ..), one float4 of a dY row
  int spx[NL], sc4[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    int e = tid + 256 * i;
    e = e < NPIECE ? e : NPIECE - 1;                    // duplicates of the last piece: identical value to the identical address
    spx[i] = e / C4; sc4[i] = e - spx[i] * C4;
  }
  const int dpx = tid >> 3, dc4 = tid & 7;
  const int64_t img_px = (int64_t)H * W;

  for (int strip = blockIdx.x; strip < nstrips; strip += gridDim.x) {
    const int ys = strip % nys, xt = (strip / nys) % ntx, b = strip / (nys * ntx);
    const int y0 = ys * rps, y1 = min(H, y0 + rps), x0 = xt * TW;
    const float* xb = x + (int64_t)b * img_px * CIN;
    const float* db = dy + (int64_t)b * img_px * Cout + co_t * 32;
    // loads are branch-free (clamped addresses) and NOTHING touches a loaded value before it is stored to LDS, where the padding mask
    // is applied: a select right after the load would put an s_waitcnt vmcnt(0) in front of the row's MFMAs (hipcc rules)
    auto load_xrow = [&](int r, f32x4 (&dst)[NL]) {
      const int rc = min(max(r, 0), H - 1);
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        const int xc = min(max(x0 - 1 + spx[i], 0), W - 1);
        dst[i] = *reinterpret_cast<const f32x4*>(xb + (unsigned)((rc * W + xc) * CIN + 4 * sc4[i]));
      }
    };
    auto store_xrow = [&](int r, const f32x4 (&src)[NL]) {
      float* slot = Xr + ((r + 1) & 3) * (HC * XS);
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        const int xx = x0 - 1 + spx[i];
        const bool ok = r >= 0 && r < H && xx >= 0 && xx < W;
        *reinterpret_cast<f32x4*>(slot + spx[i] * XS + 4 * sc4[i]) = ok ? src[i] : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    };
    auto load_drow = [&](int r) -> f32x4 {
      const int rc = min(r, H - 1), xc = min(x0 + dpx, W - 1);
      return *reinterpret_cast<const f32x4*>(db + (unsigned)((rc * W + xc) * Cout + 4 * dc4));
    };
    auto store_drow = [&](int r, bool live, const f32x4& v) {
      const bool ok = live && r < H && x0 + dpx < W;
      *reinterpret_cast<f32x4*>(Dr + (r & 1) * (TW * 32) + dpx * 32 + 4 * dc4) = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    };

    // ---- prime the ring: rows y0 - 1, y0, y0 + 1 and dY row y0 (the previous strip's last row ended with a barrier)
    {
      f32x4 r0[NL], r1[NL], r2[NL];
      load_xrow(y0 - 1, r0); load_xrow(y0, r1); load_xrow(y0 + 1, r2);
      const f32x4 d0 = load_drow(y0);
      store_xrow(y0 - 1, r0); store_xrow(y0, r1); store_xrow(y0 + 1, r2);
      store_drow(y0, true, d0);
    }
    __syncthreads();
    for (int y = y0; y < y1; ++y) {
      f32x4 nx[NL];
      load_xrow(y + 2, nx);                             // in flight under this row's MFMAs
      const f32x4 nd = load_drow(y + 1);
      const float* dr = Dr + (y & 1) * (TW * 32) + l31;
      const float* xr0 = Xr + ((y) & 3) * (HC * XS) + ci_t * 32 + l31;        // row y - 1
      const float* xr1 = Xr + ((y + 1) & 3) * (HC * XS) + ci_t * 32 + l31;    // row y
      const float* xr2 = Xr + ((y + 2) & 3) * (HC * XS) + ci_t * 32 + l31;    // row y + 1
      // operands of pair q: pixel px = 2 q + h of the tile; tap column kx = halo pixel px + kx.  Pair q + 1 shares its kx = 0 column
      // with pair q's kx = 2: per pair 2 new X values per row + 1 dY value are read from LDS, ONE PAIR AHEAD of the MFMAs that use
      // them (scheduling barrier: hipcc otherwise sinks every read to just before its use and waits lgkmcnt(0) twice per pair)
      const int q0 = split * NPAIR;
      float av, bv[9], an, n1[3], n2[3];
      {
        const int px = 2 * q0 + h;
        av = dr[px * 32];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          bv[kx] = xr0[(px + kx) * XS];
          bv[3 + kx] = xr1[(px + kx) * XS];
          bv[6 + kx] = xr2[(px + kx) * XS];
        }
      }
#pragma unroll
      for (int i = 0; i < NPAIR; ++i) {
        if (i + 1 < NPAIR) {
          const int px = 2 * (q0 + i + 1) + h;
          an = dr[px * 32];
          n1[0] = xr0[(px + 1) * XS]; n2[0] = xr0[(px + 2) * XS];
          n1[1] = xr1[(px + 1) * XS]; n2[1] = xr1[(px + 2) * XS];
          n1[2] = xr2[(px + 1) * XS]; n2[2] = xr2[(px + 2) * XS];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[t] = cmr_mfma32(av, bv[t], acc[t]);
        __builtin_amdgcn_sched_barrier(0);
        if (i + 1 < NPAIR) {
          av = an;
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) {
            bv[3 * ky] = bv[3 * ky + 2];
            bv[3 * ky + 1] = n1[ky];
            bv[3 * ky + 2] = n2[ky];
          }
        }
      }
      store_xrow(y + 2, nx);                            // slot of row y - 2: nobody reads it any more
      store_drow(y + 1, y + 1 < y1, nd);                // past the strip: zeros (never read)
      __syncthreads();
    }
  }
  float* out = part + ((int64_t)(blockIdx.x * NSPLIT + split) * 9) * Cout * CIN;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = cmr_mfma_row(r, lane);
      out[((int64_t)t * Cout + co_t * 32 + row) * CIN + ci_t * 32 + l31] = acc[t][r];
    }
}

// ------------------------------------------------------------------------------------------------------------------
// bf16 variant of the weight gradient (round 3; bf16 training mode, BASELINE configs[2]): the same contraction on
// v_mfma_f32_32x32x16_bf16 -- 16 pixels per instruction instead of 2, fp32 accumulate, operands rounded to bf16 (RNE).  At 16 x the
// fp32 matrix rate the multiplies are free (18 instructions of 16 cycles per 32-pixel row tile and wave); what has to be organised is
// the operand layout: the contraction index (pixels) must run INSIDE a lane's 16-byte operand, while NHWC memory runs channels.  So
// the rows are transposed on their way into LDS: a lane loads 4 channels of ONE pixel (lanes 0..31 an even pixel, 32..63 its odd
// neighbour), exchanges two of them with the partner lane (v_permlane32_swap), packs (even, odd) pixel pairs per channel and writes
// two dwords into X^T[channel][pixel pair].  Operand of tap column kx for the 16-pixel block pb: the 8 pixels e0 + kx .. + 7 with
// e0 = 16 pb + 8 h of channel row ci: five dwords from LDS, kx = 1 through four v_alignbit.  Column strips, ring of four input rows,
// persistent workgroups and partial layout exactly as conv3x3_wgrad_lds_kernel (same reduction kernel).
//   LDS: X^T ring [4][Cin][21 dwords] (21: odd row stride -> the 32 channel rows of an operand read fall on 32 banks; 42 >= 34 halo
//   pixels), dY^T [2][32][21 dwords].  45 KB at Cin = 128: three workgroups per CU.
// ------------------------------------------------------------------------------------------------------------------
typedef __bf16 wg_bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned wg_pack2(float lo, float hi) {            // (bf16(lo) | bf16(hi) << 16), round to nearest even
  typedef float f2 __attribute__((ext_vector_type(2)));
  typedef __bf16 b2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f2){lo, hi}, b2));
}

// OCC = workgroups per CU the register budget is set for.  2 (the Cin = 128 instances, round 5): 144 accumulator registers + < 112 others.
// One workgroup per CU kept ~80 KB of loads in flight per CU and waited for them 70 % of the time (an image row tile took 4.6 us for 1 150
// matrix-pipe cycles); two co-resident workgroups double the bytes in flight and multiply while the other waits.  The bias gradient (BIAS)
// is one more matrix instruction per pixel block against a vector of ones (wave ci tile 0 only: 16 accumulator registers) instead of
// per-lane fp32 sums over the staged dY rows -- the lane sums alone cost the instance 100 registers of live ranges.
template <int NCI, int NCO, int TW, bool BIAS, int OCC = 1>
__global__ __launch_bounds__(256, OCC) void conv3x3_wgrad_bf16_kernel(const float* __restrict__ x, const float* __restrict__ dy, int B, int H, int W,
                                                                 int Cout, int rps, float* __restrict__ part, float* __restrict__ part_b) {
  // TW = 32 | 64 pixels per row tile (64: twice the bytes in flight per barrier interval -- the kernel waits on memory, not on its 18 / 36
  // matrix instructions per row)
  constexpr int CIN = 32 * NCI, RS = TW / 2 + 5, NSPLIT = 4 / NCI;   // RS odd: 21 | 37 dwords >= TW / 2 + 1 pairs (+ the dword past an operand)
  constexpr int NPAIRX = TW / 2 + 1, NPAIRD = TW / 2, NBLK = TW / 16;
  constexpr int C4 = CIN / 4;                          // channel quads per pixel: 32 (Cin 128) | 16 (Cin 64)
  constexpr int PPI = 32 / C4;                         // pixel pairs per wave instruction of the X staging: 1 | 2
  constexpr int NXI = (NPAIRX + 4 * PPI - 1) / (4 * PPI);   // staging iterations per wave for the pixel pairs of a halo row
  extern __shared__ __attribute__((aligned(16))) unsigned wgb_smem[];
  unsigned* XT = wgb_smem;                              // [4][CIN][RS]
  unsigned* DT = wgb_smem + 4 * CIN * RS;               // [2][32 NCO][RS]
  unsigned* DL = DT + 2 * 32 * NCO * RS;                // [2][32 NCO][RS] (BIAS): the bf16 rounding residues of dY, for the fp32-exact column sums
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  const int ci_t = wave % NCI, split = wave / NCI;
  const int co_t = blockIdx.y;
  const int ntx = (W + TW - 1) / TW, nys = (H + rps - 1) / rps;
  const int nstrips = B * ntx * nys;

  // NCO cout tiles per workgroup share one staging of the input rows (the input is then read Cout / (32 NCO) times per launch)
  f32x16 acc[NCO][9];
#pragma unroll
  for (int c = 0; c < NCO; ++c)
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][t][r] = 0.f;

  // bias gradient (BIAS): column sums of dY = dY^T times a vector of ones on the matrix cores (every column of the accumulator tile is the
  // sum; cmr_colsum_f32 used to re-read the whole map for it: 25 us per 88x304 layer of the agent update).  The bias is the sum of the
  // UNROUNDED gradients: dY = hi + lo with hi = bf16(dY) (the A operand the wave holds anyway) and lo = bf16(dY - hi) staged next to it --
  // two instructions per pixel block, error 2^-17 per element.  The waves of ci tile 0 do it (Cin = 64: each for its own pixel blocks).
  f32x16 accb[NCO];
#pragma unroll
  for (int c = 0; c < NCO; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) accb[c][r] = 0.f;
  // staging roles: X: lane -> (channel quad xc4, pair slot within the instruction xps, parity h); dY: (quad dc4 of 8 NCO, pair slot dps)
  constexpr int DQ = 8 * NCO, DPI = 32 / DQ;            // dY channel quads; pixel pairs per wave instruction: 4 | 2
  constexpr int NDI = NPAIRD / (4 * DPI);               // dY staging iterations per wave
  const int xc4 = l31 % C4, xps = l31 / C4;
  const int dc4 = l31 % DQ, dps = l31 / DQ;
  const int64_t img_px = (int64_t)H * W;

  for (int strip = blockIdx.x; strip < nstrips; strip += gridDim.x) {
    const int ys = strip % nys, xt = (strip / nys) % ntx, b = strip / (nys * ntx);
    const int y0 = ys * rps, y1 = min(H, y0 + rps), x0 = xt * TW;
    const float* xb = x + (int64_t)b * img_px * CIN;
    const float* db = dy + (int64_t)b * img_px * Cout + co_t * (32 * NCO);
    // pixel pair pp of a halo row = halo elements 2 pp, 2 pp + 1 = image columns x0 - 1 + 2 pp (+ 1); this lane's element: 2 pp + h
    auto xpair = [&](int i) { return min((wave + 4 * i) * PPI + xps, NPAIRX - 1); };   // duplicates of the last pair: identical writes
    auto load_xrow = [&](int r, f32x4 (&dst)[NXI]) {
      const int rc = min(max(r, 0), H - 1);
#pragma unroll
      for (int i = 0; i < NXI; ++i) {
        const int xc = min(max(x0 - 1 + 2 * xpair(i) + h, 0), W - 1);
        dst[i] = *reinterpret_cast<const f32x4*>(xb + (unsigned)((rc * W + xc) * CIN + 4 * xc4));
      }
    };
    // (even, odd) pixel pairs per channel: lanes 0..31 hold the even pixel, 32..63 the odd one, of the same 4 channels.  The low half
    // finishes channels 0, 1 of the quad, the high half channels 2, 3: each sends the partner the two values it needs.
    auto store_pairs = [&](unsigned* row0, int quad, int pp, f32x4 v, bool ok) {
      v = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
      const float s0 = h ? v[0] : v[2], s1 = h ? v[1] : v[3];
      const float r0 = cmr_xhalf(s0), r1 = cmr_xhalf(s1);
      const unsigned w0 = h ? wg_pack2(r0, v[2]) : wg_pack2(v[0], r0);
      const unsigned w1 = h ? wg_pack2(r1, v[3]) : wg_pack2(v[1], r1);
      unsigned* p = row0 + (4 * quad + 2 * h) * RS + pp;
      p[0] = w0;
      p[RS] = w1;
    };
    auto store_xrow = [&](int r, const f32x4 (&src)[NXI]) {
      unsigned* slot = XT + ((r + 1) & 3) * (CIN * RS);
#pragma unroll
      for (int i = 0; i < NXI; ++i) {
        const int xx = x0 - 1 + 2 * xpair(i) + h;
        store_pairs(slot, xc4, xpair(i), src[i], r >= 0 && r < H && xx >= 0 && xx < W);
      }
    };
    // dY row: 16 pixel pairs x 8 NCO channel quads; element 2 pp + h = image column x0 + 2 pp + h
    auto dpair = [&](int i) { return (wave * NDI + i) * DPI + dps; };
    auto load_drow = [&](int r, f32x4 (&dst)[NDI]) {
      const int rc = min(r, H - 1);
#pragma unroll
      for (int i = 0; i < NDI; ++i) {
        const int xc = min(x0 + 2 * dpair(i) + h, W - 1);
        dst[i] = *reinterpret_cast<const f32x4*>(db + (unsigned)((rc * W + xc) * Cout + 4 * dc4));
      }
    };
    auto store_drow = [&](int r, bool live, const f32x4 (&src)[NDI]) {
#pragma unroll
      for (int i = 0; i < NDI; ++i) {
        const bool ok = live && r < H && x0 + 2 * dpair(i) + h < W;
        if (BIAS) {
          f32x4 lo;
#pragma unroll
          for (int e = 0; e < 4; ++e) lo[e] = src[i][e] - __builtin_bit_cast(float, wg_pack2(0.f, src[i][e]) & 0xffff0000u);
          store_pairs(DL + (r & 1) * (32 * NCO * RS), dc4, dpair(i), lo, ok);
        }
        store_pairs(DT + (r & 1) * (32 * NCO * RS), dc4, dpair(i), src[i], ok);
      }
    };

    // Rows are requested TWO iterations before they are staged: a row tile is 18 matrix instructions (~0.15 us), so with the loads
    // of row y + 2 issued at the top of iteration y and stored at its bottom every iteration waited a full memory round trip (2 us
    // per row: 254 us at 10 x 88 x 304).  Two register sets alternate (rows y + 2 / y + 3) under static names, two rows per trip.
    f32x4 xa[NXI], xb2[NXI], da[NDI], dbv[NDI];
    if (OCC == 1) {
      f32x4 r0[NXI], r1[NXI], r2[NXI], d0[NDI];
      load_xrow(y0 - 1, r0); load_xrow(y0, r1); load_xrow(y0 + 1, r2);
      load_drow(y0, d0);
      load_xrow(y0 + 2, xa);                             // stays in registers until the bottom of iteration y0
      load_drow(y0 + 1, da);
      store_xrow(y0 - 1, r0); store_xrow(y0, r1); store_xrow(y0 + 1, r2);
      store_drow(y0, true, d0);
    } else {                                             // half the register budget: the strip's first rows in two batches
      {
        f32x4 r0[NXI], r1[NXI];
        load_xrow(y0 - 1, r0); load_xrow(y0, r1);
        store_xrow(y0 - 1, r0); store_xrow(y0, r1);
      }
      f32x4 r2[NXI], d0[NDI];
      load_xrow(y0 + 1, r2);
      load_drow(y0, d0);
      load_xrow(y0 + 2, xa);
      load_drow(y0 + 1, da);
      store_xrow(y0 + 1, r2);
      store_drow(y0, true, d0);
    }
    __syncthreads();
    auto multiply_row = [&](int y) __attribute__((always_inline)) {
      const unsigned* dr = DT + (y & 1) * (32 * NCO * RS) + l31 * RS;
      const unsigned* xr[3] = {XT + ((y) & 3) * (CIN * RS) + (ci_t * 32 + l31) * RS, XT + ((y + 1) & 3) * (CIN * RS) + (ci_t * 32 + l31) * RS,
                               XT + ((y + 2) & 3) * (CIN * RS) + (ci_t * 32 + l31) * RS};
#pragma unroll
      for (int pbi = 0; pbi < NBLK / NSPLIT; ++pbi) {
        const int pb = split * (NBLK / NSPLIT) + pbi;   // Cin = 64: the pixel blocks of a row tile are split over two waves
        const int d0 = 8 * pb + 4 * h;                  // first dword of this lane's 8 pixels (dY: element = pixel; X: element = pixel + kx)
        wg_bf16x8 av[NCO];
#pragma unroll
        for (int c = 0; c < NCO; ++c) {
          uint4 aw;
          aw.x = dr[c * 32 * RS + d0]; aw.y = dr[c * 32 * RS + d0 + 1]; aw.z = dr[c * 32 * RS + d0 + 2]; aw.w = dr[c * 32 * RS + d0 + 3];
          av[c] = __builtin_bit_cast(wg_bf16x8, aw);
        }
        if (BIAS && ci_t == 0) {                        // (wave-uniform)
          const uint4 one8 = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
          const unsigned* lr = DL + (y & 1) * (32 * NCO * RS) + l31 * RS;
#pragma unroll
          for (int c = 0; c < NCO; ++c) {
            uint4 lw;
            lw.x = lr[c * 32 * RS + d0]; lw.y = lr[c * 32 * RS + d0 + 1]; lw.z = lr[c * 32 * RS + d0 + 2]; lw.w = lr[c * 32 * RS + d0 + 3];
            accb[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[c], __builtin_bit_cast(wg_bf16x8, one8), accb[c], 0, 0, 0);
            accb[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wg_bf16x8, lw), __builtin_bit_cast(wg_bf16x8, one8), accb[c], 0, 0, 0);
          }
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          unsigned q[5];
#pragma unroll
          for (int i = 0; i < 5; ++i) q[i] = xr[ky][d0 + i];
          const uint4 b0 = {q[0], q[1], q[2], q[3]};
          const uint4 b1 = {__builtin_amdgcn_alignbit(q[1], q[0], 16), __builtin_amdgcn_alignbit(q[2], q[1], 16),
                            __builtin_amdgcn_alignbit(q[3], q[2], 16), __builtin_amdgcn_alignbit(q[4], q[3], 16)};
          const uint4 b2 = {q[1], q[2], q[3], q[4]};
#pragma unroll
          for (int c = 0; c < NCO; ++c) {
            acc[c][3 * ky] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[c], __builtin_bit_cast(wg_bf16x8, b0), acc[c][3 * ky], 0, 0, 0);
            acc[c][3 * ky + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[c], __builtin_bit_cast(wg_bf16x8, b1), acc[c][3 * ky + 1], 0, 0, 0);
            acc[c][3 * ky + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[c], __builtin_bit_cast(wg_bf16x8, b2), acc[c][3 * ky + 2], 0, 0, 0);
          }
        }
      }
    };
    for (int y = y0; y < y1; y += 2) {
      load_xrow(y + 3, xb2);                            // set B: row y + 3, staged at the bottom of iteration y + 1
      load_drow(y + 2, dbv);
      multiply_row(y);
      store_xrow(y + 2, xa);                            // set A was requested one iteration ago; slot of row y - 2
      store_drow(y + 1, y + 1 < y1, da);
      __syncthreads();
      if (y + 1 >= y1) break;                           // (uniform)
      load_xrow(y + 4, xa);
      load_drow(y + 3, da);
      multiply_row(y + 1);
      store_xrow(y + 3, xb2);
      store_drow(y + 2, y + 2 < y1, dbv);
      __syncthreads();
    }
  }
  float* out = part + ((int64_t)(blockIdx.x * NSPLIT + split) * 9) * Cout * CIN;
#pragma unroll
  for (int c = 0; c < NCO; ++c)
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = cmr_mfma_row(r, lane);
        out[((int64_t)t * Cout + (co_t * NCO + c) * 32 + row) * CIN + ci_t * 32 + l31] = acc[c][t][r];
      }
  if (BIAS && ci_t == 0 && l31 == 0) {                  // column 0 of the ones product: lanes 0 and 32 hold its 32 rows
#pragma unroll
    for (int c = 0; c < NCO; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        part_b[(int64_t)(blockIdx.x * NSPLIT + split) * Cout + (co_t * NCO + c) * 32 + cmr_mfma_row(r, lane)] = accb[c][r];
  }
}

// ------------------------------------------------------------------------------------------------------------------
// bf16 weight gradient, second generation (round 5; Cin = 128, Cout % 64 == 0, maps of >= 32 768 pixels): the operands are transposed
// by the LDS hardware on the way OUT (ds_read_b64_tr_b16) instead of by the lanes on the way in.
//
// conv3x3_wgrad_bf16_kernel above spends its time in the staging pass: per 4-channel load two cross-lane swaps, two packs and two scattered
// ds_write_b32 (4- to 8-way bank conflicts at the odd row stride the transposed image needs) -- ~2 500 issue cycles per row tile and wave
// against 576 cycles of matrix instructions (0.19 of the bf16 matrix rate standing alone, profiles/r05_wgrad_bench.txt).  Here the rows go
// to LDS as they lie in memory, [pixel][channel] in bf16 (one v_cvt_pk pair and ONE conflict-free ds_write_b64 per load), and an operand
// -- 8 consecutive pixels of one channel per lane -- is two ds_read_b64_tr_b16: the instruction hands lane i of a 16-lane group column i
// of a 4-row x 16-column block.  Image = 256-byte pixel rows, 16-byte chunk c of pixel p at chunk c ^ (((p & 3) << 2) | ((p >> 2) & 3)):
// conflict-free for the writes and the transposed reads (cdna_hip_programming.md T10, image (b)).
//
// Workgroup = 8 waves = 4 ci tiles x 2 cout tiles (64 couts x 128 cins x 9 taps in 144 accumulator registers per wave, two waves per
// SIMD): the input rows are staged once per 64 couts (twice per launch at Cout = 128; the two workgroups of a strip sit on the same XCD
// and walk in step, so the second read is an L2 hit) and all 512 lanes share the staging.  dY rows carry their bf16 rounding residues in
// the upper 64 channels of the same 256-byte pixel row: the bias gradient is two more matrix instructions against a vector of ones
// (hi + lo: the sum of the unrounded gradients to 2^-17).  Column strips, ring of four input rows, persistent workgroups, rows requested
// two iterations ahead, partial layout and reduction exactly as above.
// ------------------------------------------------------------------------------------------------------------------
typedef short wg_s4 __attribute__((ext_vector_type(4)));
constexpr int WT2_TW = 32, WT2_NPX = WT2_TW + 4;       // 34 halo pixels + 2 of slack (the last transposed read of a row runs two pixels past it)
constexpr int WT2_XROW = WT2_NPX * 256;                // bytes per ring slot
constexpr int WT2_DROW = WT2_TW * 256;                 // bytes per dY row image (64 channels hi | 64 channels lo)
constexpr int WT2_SMEM = 4 * WT2_XROW + 2 * WT2_DROW;  // 53 248 B

__device__ __forceinline__ int wt2_off(int pixel, int chunk) { return 256 * pixel + 16 * (chunk ^ (((pixel & 3) << 2) | ((pixel >> 2) & 3))); }
__device__ __forceinline__ uint2 wt2_tr(const unsigned char* base, int byte_off) {
  const wg_s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wg_s4*)(base + byte_off));
  return __builtin_bit_cast(uint2, v);
}

// XPRO: x is a BatchNorm INPUT and the operand is lrelu_{xslope}(x * xscale + xshift) per channel (the activation was never stored:
// cmr_conv3x3_bf16_pro_nhwc_f32 formed it the same way in the forward); the affine sits in LDS behind the images, the staging pass applies
// it with the fused multiply-add and the select of cmr_affine_act_f32 -- bit for bit the operand read from the stored map.
template <bool BIAS, bool XPRO = false>
__global__ __launch_bounds__(512, 1) void conv3x3_wgrad_bf16_tr_kernel(const float* __restrict__ x, const float* __restrict__ dy, int B, int H, int W,
                                                                       int Cout, int rps, int groups, float* __restrict__ part, float* __restrict__ part_b,
                                                                       const float* __restrict__ xscale = nullptr, const float* __restrict__ xshift = nullptr,
                                                                       float xslope = 1.f) {
  constexpr int CIN = 128, TW = WT2_TW;
  extern __shared__ __attribute__((aligned(16))) unsigned char wt2_smem[];
  float* xaff = reinterpret_cast<float*>(wt2_smem + WT2_SMEM);            // XPRO: [2][128] scale | shift
  if (XPRO) {
    if (threadIdx.x < 128) xaff[threadIdx.x] = xscale[threadIdx.x];
    else if (threadIdx.x < 256) xaff[threadIdx.x] = xshift[threadIdx.x - 128];
    __syncthreads();
  }
  unsigned char* XI = wt2_smem;                         // [4][36 px][256 B]
  unsigned char* DI = wt2_smem + 4 * WT2_XROW;          // [2][32 px][256 B]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ci_t = wave & 3, co_h = wave >> 2;
  // workgroup -> (strip group, cout pair): the cout pairs of one group get consecutive slots of ONE XCD (ids are dealt to the XCDs round robin)
  const int ncp = Cout / 64;
  int group, co_p;
  if ((groups & 7) == 0) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    co_p = j % ncp;
    group = (j / ncp) * 8 + xcd;
  } else {
    co_p = blockIdx.x % ncp;
    group = blockIdx.x / ncp;
  }
  const int ntx = (W + TW - 1) / TW, nys = (H + rps - 1) / rps;
  const int nstrips = B * ntx * nys;
  const int64_t img_px = (int64_t)H * W;

  f32x16 acc[9], accb;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) accb[r] = 0.f;

  // transposed-read addresses of this lane: 16-lane group g = lane >> 4 covers channels 16 (g & 1) .. + 15 of the wave's 32-channel tile and
  // the pixel half h = g >> 1; lane 4 q + p of the group supplies the address of block row q, columns 4 p .. 4 p + 3
  const int g16 = lane >> 4, q4 = (lane >> 2) & 3, p4 = lane & 3, h = g16 >> 1;
  const int xchunk = 4 * ci_t + 2 * (g16 & 1) + (p4 >> 1);            // 16-byte chunk of the input row image
  const int dchunk = 4 * co_h + 2 * (g16 & 1) + (p4 >> 1);            // ... of the dY row image (hi half; lo = + 8)
  const int sub8 = 8 * (p4 & 1);
  // the swizzled byte offsets of this lane's transposed reads do not change from row to row: [pixel block][4-pixel group]
  int ox[2][3], od[2][2], ol[2][2];
#pragma unroll
  for (int pb = 0; pb < 2; ++pb) {
    const int pd = 16 * pb + 8 * h + q4;
#pragma unroll
    for (int j = 0; j < 3; ++j) { ox[pb][j] = wt2_off(pd + 4 * j, xchunk) + sub8; asm volatile("" : "+v"(ox[pb][j])); }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      od[pb][j] = wt2_off(pd + 4 * j, dchunk) + sub8; asm volatile("" : "+v"(od[pb][j]));
      ol[pb][j] = wt2_off(pd + 4 * j, dchunk + 8) + sub8; asm volatile("" : "+v"(ol[pb][j]));
    }
  }
  // staging roles: X: thread -> (halo pixel, channel quad); 34 x 32 = 1 088 quads over 512 threads: 3 passes (the tail re-writes its last quad)
  const int xq = tid & 31;                                           // channel quad of the X loads
  const int xp0 = tid >> 5;                                          // halo pixel of pass 0 (pass i: + 16 i)
  const int dq = tid & 15, dp = tid >> 4;                            // dY: 32 pixels x 16 channel quads (this pair's 64 couts) = 512 loads

  for (int strip = group; strip < nstrips; strip += groups) {
    const int ys = strip % nys, xt = (strip / nys) % ntx, b = strip / (nys * ntx);
    const int y0 = ys * rps, y1 = min(H, y0 + rps), x0 = xt * TW;
    const float* xb = x + (int64_t)b * img_px * CIN;
    const float* db = dy + (int64_t)b * img_px * Cout + co_p * 64;
    auto xpix = [&](int i) { return min(xp0 + 16 * i, TW + 1); };
    auto load_xrow = [&](int r, f32x4 (&dst)[3]) {
      const int rc = min(max(r, 0), H - 1);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int xc = min(max(x0 - 1 + xpix(i), 0), W - 1);
        dst[i] = *reinterpret_cast<const f32x4*>(xb + (unsigned)((rc * W + xc) * CIN + 4 * xq));
      }
    };
    auto store_xrow = [&](int r, const f32x4 (&src)[3]) {
      unsigned char* slot = XI + ((r + 1) & 3) * WT2_XROW;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int px = xpix(i), xx = x0 - 1 + px;
        const bool ok = r >= 0 && r < H && xx >= 0 && xx < W;
        f32x4 t = src[i];
        if (XPRO) {
          const f32x4 sc = *reinterpret_cast<const f32x4*>(xaff + 4 * xq), sh = *reinterpret_cast<const f32x4*>(xaff + 128 + 4 * xq);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            t[e] = __builtin_fmaf(t[e], sc[e], sh[e]);
            t[e] = fmaxf(t[e], t[e] * xslope);                  // = the select for 0 <= slope <= 1 (checked by the entry point)
          }
        }
        const f32x4 v = ok ? t : f32x4{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<uint2*>(slot + wt2_off(px, xq >> 1) + 8 * (xq & 1)) = uint2{wg_pack2(v[0], v[1]), wg_pack2(v[2], v[3])};
      }
    };
    auto load_drow = [&](int r, f32x4& dst) {
      const int rc = min(r, H - 1), xc = min(x0 + dp, W - 1);
      dst = *reinterpret_cast<const f32x4*>(db + (unsigned)((rc * W + xc) * Cout + 4 * dq));
    };
    auto store_drow = [&](int r, bool live, const f32x4& src) {
      unsigned char* img = DI + (r & 1) * WT2_DROW;
      const bool ok = live && r < H && x0 + dp < W;
      const f32x4 v = ok ? src : f32x4{0.f, 0.f, 0.f, 0.f};
      const unsigned w0 = wg_pack2(v[0], v[1]), w1 = wg_pack2(v[2], v[3]);
      *reinterpret_cast<uint2*>(img + wt2_off(dp, dq >> 1) + 8 * (dq & 1)) = uint2{w0, w1};
      if (BIAS) {                                         // residues dY - bf16(dY), in the upper 64 channels of the same pixel row
        const float l0 = v[0] - __builtin_bit_cast(float, w0 << 16), l1 = v[1] - __builtin_bit_cast(float, w0 & 0xffff0000u);
        const float l2 = v[2] - __builtin_bit_cast(float, w1 << 16), l3 = v[3] - __builtin_bit_cast(float, w1 & 0xffff0000u);
        *reinterpret_cast<uint2*>(img + wt2_off(dp, 8 + (dq >> 1)) + 8 * (dq & 1)) = uint2{wg_pack2(l0, l1), wg_pack2(l2, l3)};
      }
    };

    f32x4 xa[3], xb2[3], da, dbv;
    {
      {
        f32x4 r0[3], r1[3];
        load_xrow(y0 - 1, r0); load_xrow(y0, r1);
        store_xrow(y0 - 1, r0); store_xrow(y0, r1);
      }
      f32x4 r2[3], d0;
      load_xrow(y0 + 1, r2);
      load_drow(y0, d0);
      load_xrow(y0 + 2, xa);
      load_drow(y0 + 1, da);
      store_xrow(y0 + 1, r2);
      store_drow(y0, true, d0);
    }
    __syncthreads();
    auto multiply_row = [&](int y) __attribute__((always_inline)) {
      const unsigned char* dimg = DI + (y & 1) * WT2_DROW;
#pragma unroll
      for (int pb = 0; pb < TW / 16; ++pb) {
        const uint2 a0 = wt2_tr(dimg, od[pb][0]), a1 = wt2_tr(dimg, od[pb][1]);
        const wg_bf16x8 av = __builtin_bit_cast(wg_bf16x8, uint4{a0.x, a0.y, a1.x, a1.y});
        if (BIAS && ci_t == 0) {                          // (wave-uniform)
          const uint4 one8 = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
          const uint2 l0 = wt2_tr(dimg, ol[pb][0]), l1 = wt2_tr(dimg, ol[pb][1]);
          accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(wg_bf16x8, one8), accb, 0, 0, 0);
          accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wg_bf16x8, uint4{l0.x, l0.y, l1.x, l1.y}), __builtin_bit_cast(wg_bf16x8, one8), accb, 0, 0, 0);
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const unsigned char* ximg = XI + ((y + ky) & 3) * WT2_XROW;
          // halo pixels 16 pb + 8 h .. + 11 of this lane's channel: three 4-pixel reads; tap kx takes pixels + kx .. + kx + 7
          const uint2 r0 = wt2_tr(ximg, ox[pb][0]), r1 = wt2_tr(ximg, ox[pb][1]), r2 = wt2_tr(ximg, ox[pb][2]);
          const uint4 b0 = {r0.x, r0.y, r1.x, r1.y};
          const uint4 b1 = {__builtin_amdgcn_alignbit(r0.y, r0.x, 16), __builtin_amdgcn_alignbit(r1.x, r0.y, 16),
                            __builtin_amdgcn_alignbit(r1.y, r1.x, 16), __builtin_amdgcn_alignbit(r2.x, r1.y, 16)};
          const uint4 b2 = {r0.y, r1.x, r1.y, r2.x};
          acc[3 * ky] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(wg_bf16x8, b0), acc[3 * ky], 0, 0, 0);
          acc[3 * ky + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(wg_bf16x8, b1), acc[3 * ky + 1], 0, 0, 0);
          acc[3 * ky + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(wg_bf16x8, b2), acc[3 * ky + 2], 0, 0, 0);
        }
      }
    };
    for (int y = y0; y < y1; y += 2) {
      load_xrow(y + 3, xb2);
      load_drow(y + 2, dbv);
      multiply_row(y);
      store_xrow(y + 2, xa);
      store_drow(y + 1, y + 1 < y1, da);
      __syncthreads();
      if (y + 1 >= y1) break;                           // (uniform)
      load_xrow(y + 4, xa);
      load_drow(y + 3, da);
      multiply_row(y + 1);
      store_xrow(y + 3, xb2);
      store_drow(y + 2, y + 2 < y1, dbv);
      __syncthreads();
    }
  }
  const int l31 = lane & 31;
  float* out = part + (int64_t)group * 9 * Cout * CIN;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      out[((int64_t)t * Cout + (co_p * 2 + co_h) * 32 + cmr_mfma_row(r, lane)) * CIN + ci_t * 32 + l31] = acc[t][r];
  if (BIAS && ci_t == 0 && l31 == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) part_b[(int64_t)group * Cout + (co_p * 2 + co_h) * 32 + cmr_mfma_row(r, lane)] = accb[r];
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Third generation (round 5): the same operands, images and matrix instructions as the kernel above -- and the same sums in the same order,
// bit for bit -- with the two things its timeline showed (8.6 strips x [2 dependent HBM round trips to fill the pipeline + 8 rows at one
// row per HALF an HBM latency, the register prefetch being two rows deep]: 138 us for 0.48 us of matrix work per row) taken out:
//   * rows arrive by LDS-DMA (global_load_lds_dwordx4, no registers): a ring of WT3_D raw fp32 rows (34 pixels x 128 channels of x and 32
//     pixels x 64 channels of dY, 25 KB) filled WT3_D - 1 iterations ahead of their use, across strip boundaries -- the stream of rows of
//     a workgroup's strips is ONE pipeline that is filled once per launch.  Zero padding comes from the DMA's per-lane source address (a
//     zero page), so the conversion pass (raw fp32 -> the swizzled bf16 images, LDS to LDS through the VALU) has no bounds logic at all;
//   * one INPUT row per iteration: the staged x row r meets the dY rows r + 1, r, r - 1 (kernel rows 0, 1, 2) whose operand fragments stay
//     in registers for three iterations (24 registers -- the register prefetch they replace was 32), so an iteration reads 5 fragments per
//     16-pixel block from LDS instead of 11 and needs ONE x row and ONE dY row image (two slots each) instead of a ring of four.
// The DMA is issued through inline asm (M0 written in the same statement): hipcc drains the builtin's DMA with vmcnt(0) in front of every
// LDS read of the same wave, which would serialise the ring; the waits here are counted (every wave issues exactly WT3_NDMA per row).
// One raw s_barrier per row: [wait: row i + 1 landed] barrier [issue row i + D] [convert row i + 1 between the matrix instructions of row i].
// MEASURED (round 5, tools/wgrad_dma_ablate.py -> profiles/r05_wgrad_dma_ablate.txt, PMC profiles/r05_pmc_wgrad_bf16_gen3.txt): bit-identical
// on the same strips and 15 - 25 % SLOWER than the second generation (10 x 88 x 304 x 128 -> 128 with the bias gradient: 160 - 185 us against
// 144 - 161 us on the same boxes), although it fetches less from HBM (299 MB against 379 MB: longer strips).  The ablations: without the DMA
// the same launch takes 80 us, the DMA and its conversion alone 101 us, the conversion is hidden (- 5 us without it), the matrix instructions
// nearly (- 14 us), the transposed reads are not (- 37 us); the launch moves 563 MB from L2 into LDS (both workgroups of a cout pair stage
// the whole x row: 9 x 128 x 64 accumulators are what a CU's registers hold) at 2.2 TB/s of HBM traffic with the waves waiting 0.43 of
// their cycles -- neither HBM nor the matrix pipe (busy 0.18) is the limit, the one-barrier-per-row lockstep of DMA wait, conversion and
// multiplication is.  The library keeps the second generation (CMR_WGRAD_BF16_GEN 1); this kernel is built into the A/B library only.
// ------------------------------------------------------------------------------------------------------------------
constexpr int WT3_D = 5;                                  // raw rows in the ring
constexpr int WT3_NDMA = 4;                               // DMA instructions per wave and row (3 of x, 1 of dY)
constexpr int WT3_XRAW = (WT2_TW + 2) * 512;              // 17 408 B: 34 pixels x 128 channels fp32
constexpr int WT3_DRAW = WT2_TW * 256;                    //  8 192 B: 32 pixels x 64 channels fp32
constexpr int WT3_RAW = WT3_XRAW + WT3_DRAW;
constexpr int WT3_XI = WT3_D * WT3_RAW;                   // bf16 x row images [2][WT2_XROW]
constexpr int WT3_DI = WT3_XI + 2 * WT2_XROW;             // bf16 dY row images [2][WT2_DROW] (hi | lo)
constexpr int WT3_SMEM = WT3_DI + 2 * WT2_DROW;           // 162 816 B of the 163 840
__device__ __attribute__((aligned(16))) float wt3_zero[4] = {0.f, 0.f, 0.f, 0.f};     // NOT const (loads of a const zero page fold into branches)

__device__ __forceinline__ void wt3_glds16(const float* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// ABL: compile-time ablation mask (timing only, results meaningless; instantiated in the A/B library alone): 1 no DMA, 2 no conversion
// pass, 4 no transposed reads (operands left as they are), 8 no matrix instructions
template <bool BIAS, int ABL = 0>
__global__ __launch_bounds__(512, 1) void conv3x3_wgrad_bf16_dma_kernel(const float* __restrict__ x, const float* __restrict__ dy, int B, int H, int W,
                                                                        int Cout, int rps, int groups, float* __restrict__ part, float* __restrict__ part_b) {
  constexpr int CIN = 128, TW = WT2_TW, D = WT3_D;
  extern __shared__ __attribute__((aligned(16))) unsigned char wt3_smem[];
  if (ABL == 32) return;                                  // (ablation: the launch alone)
  const unsigned lds0 = (unsigned)(size_t)wt3_smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ci_t = wave & 3, co_h = wave >> 2;
  const int ncp = Cout / 64;
  int group, co_p;
  if ((groups & 7) == 0) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    co_p = j % ncp;
    group = (j / ncp) * 8 + xcd;
  } else {
    co_p = blockIdx.x % ncp;
    group = blockIdx.x / ncp;
  }
  const int ntx = (W + TW - 1) / TW, nys = (H + rps - 1) / rps;
  const int nstrips = B * ntx * nys;
  const int64_t img_px = (int64_t)H * W;

  f32x16 acc[9], accb;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) accb[r] = 0.f;

  // transposed-read addresses (as in the kernel above)
  const int g16 = lane >> 4, q4 = (lane >> 2) & 3, p4 = lane & 3, h = g16 >> 1;
  const int xchunk = 4 * ci_t + 2 * (g16 & 1) + (p4 >> 1);
  const int dchunk = 4 * co_h + 2 * (g16 & 1) + (p4 >> 1);
  const int sub8 = 8 * (p4 & 1);
  int ox[2][3], od[2][2], ol[2][2];
#pragma unroll
  for (int pb = 0; pb < 2; ++pb) {
    const int pd = 16 * pb + 8 * h + q4;
#pragma unroll
    for (int j = 0; j < 3; ++j) { ox[pb][j] = wt2_off(pd + 4 * j, xchunk) + sub8; asm volatile("" : "+v"(ox[pb][j])); }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      od[pb][j] = wt2_off(pd + 4 * j, dchunk) + sub8; asm volatile("" : "+v"(od[pb][j]));
      ol[pb][j] = wt2_off(pd + 4 * j, dchunk + 8) + sub8; asm volatile("" : "+v"(ol[pb][j]));
    }
  }
  // conversion roles: x: thread -> (halo pixel xp0 + 16 i, channel quad xq); dY: thread -> (pixel dp, channel quad dq)
  const int xq = tid & 31, xp0 = tid >> 5;
  const int dq = tid & 15, dp = tid >> 4;
  // DMA roles: x block k = raw bytes [1024 k, + 1024) = halo pixels 2 k, 2 k + 1 (lane: pixel 2 k + (lane >> 5), 16-byte chunk lane & 31);
  // wave w issues blocks w, w + 8 and min(w + 16, 16) (17 blocks: the waves 1..7 repeat block 16 -- the same bytes to the same place -- so
  // that every wave has the same number of DMA instructions in flight); dY block w = pixels 4 w .. 4 w + 3 (lane: pixel 4 w + (lane >> 4),
  // chunk lane & 15)
  int xk[3];
  xk[0] = wave; xk[1] = wave + 8; xk[2] = wave + 16 < 17 ? wave + 16 : 16;
  const int xlp = lane >> 5, xlc = lane & 31;
  const int dlp = 4 * wave + (lane >> 4), dlc = lane & 15;

  // The stream of row jobs of this workgroup: strip after strip, x rows y0 - 1 .. y1 of each (job = x row r + dY row r + 1).  ONE cursor
  // (the DMA's); the per-lane column parts of the source addresses change with the strip only.  The conversion and the multiplication need
  // no position at all: padding and rows outside a strip's range arrive as ZEROS (the zero page), so every job runs the same straight-line
  // body -- a strip's first and last job multiply two zero fragments each (exact zeros onto the accumulators: + 25 % matrix instructions
  // against eight strip-dependent branches per row; an empty loop of the branchy form took 0.55 us per row, profiles/r05_wgrad_dma_ablate.txt).
  int cs = group, cr = 0, cy0 = 0, cy1 = 0, cb = 0;      // cursor: strip, x row, output rows [cy0, cy1), sample
  int xoff[3], doff = 0;                                  // per lane: byte offset of the lane's 16 bytes within an x row / dY row, < 0: outside
  auto open_strip = [&](int strip) {
    cs = strip;
    if (strip < nstrips) {
      const int ys = strip % nys, xt = (strip / nys) % ntx;
      cb = strip / (nys * ntx);
      cy0 = ys * rps;
      cy1 = min(H, cy0 + rps);
      cr = cy0 - 1;
      const int x0 = xt * TW;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int xx = x0 - 1 + 2 * xk[i] + xlp;
        xoff[i] = xx >= 0 && xx < W ? (xx * CIN + 4 * xlc) * 4 : -1;
      }
      const int xx = x0 + dlp;
      doff = xx < W ? (xx * Cout + co_p * 64 + 4 * dlc) * 4 : -1;
    }
  };
  int njobs = 0;
  for (int st = group; st < nstrips; st += groups) {
    const int y0 = (st % nys) * rps;
    njobs += min(H, y0 + rps) - y0 + 2;
  }
  auto issue = [&](int slot) {
    if (ABL & 1) return;
    const bool live = cs < nstrips;
    const unsigned base = lds0 + (unsigned)(slot * WT3_RAW);
    const bool rowx = live && cr >= 0 && cr < H;
    const char* xb = reinterpret_cast<const char*>(x + ((int64_t)cb * img_px + (int64_t)(rowx ? cr : 0) * W) * CIN);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float* src = !(ABL & 256) && rowx && xoff[i] >= 0 ? reinterpret_cast<const float*>(xb + (unsigned)xoff[i]) : wt3_zero;      // (256: every DMA from the zero page)
      wt3_glds16(src, __builtin_amdgcn_readfirstlane(base + 1024u * (unsigned)xk[i]));
    }
    const int rd = cr + 1;
    const bool rowd = live && rd >= cy0 && rd < cy1;
    const char* db = reinterpret_cast<const char*>(dy + ((int64_t)cb * img_px + (int64_t)(rowd ? rd : 0) * W) * Cout);
    const float* src = !(ABL & 256) && rowd && doff >= 0 ? reinterpret_cast<const float*>(db + (unsigned)doff) : wt3_zero;
    wt3_glds16(src, __builtin_amdgcn_readfirstlane(base + (unsigned)WT3_XRAW + 1024u * (unsigned)wave));
    // next job
    if (live) {
      if (cr < cy1) ++cr;
      else open_strip(cs + groups);
    }
  };
  // the bias gradient's matrix instructions (dY fragment x ones: hi and lo image of both 16-pixel blocks = 4 per row and cout tile) are
  // dealt to the four ci waves of the cout tile, one each and row -- all of them in the waves ci_t = 0 they made SIMD 0 the slowest of the
  // four by 8 of 44 instructions per row.  Wave ci_t: block ci_t >> 1, image ci_t & 1; its own partial sum (slice 4 group + ci_t).
  int obias[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) obias[j] = (ci_t & 1) ? ol[ci_t >> 1][j] : od[ci_t >> 1][j];
  uint4 ap1[2], a0[2], am1[2];                           // dY fragments of the rows r + 1, r, r - 1 (per 16-pixel block)
#pragma unroll
  for (int pb = 0; pb < 2; ++pb) ap1[pb] = a0[pb] = am1[pb] = uint4{0u, 0u, 0u, 0u};
  uint2 abl_keep = {0x3f803f80u, 0x3f803f80u};
  auto trr = [&](const unsigned char* base, int off) __attribute__((always_inline)) -> uint2 {
    if (ABL & 4) { asm volatile("" : "+v"(abl_keep.x), "+v"(abl_keep.y)); return abl_keep; }
    return wt2_tr(base, off);
  };
  auto mm = [&](wg_bf16x8 a, wg_bf16x8 b, f32x16 c) __attribute__((always_inline)) -> f32x16 {
    if (ABL & 8) { asm volatile("" : "+v"(a), "+v"(b)); return c; }
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  };
  // One row job: multiply row j from the images `img` while row j + 1 is converted from raw slot `slot` into the other images.  ONE
  // straight-line block (the conversion's LDS reads and writes and its VALU work sit between the matrix instructions in program order, so
  // that they issue under them; the two are independent: other images, other slot).
  auto row = [&](int slot, int img) __attribute__((always_inline)) {
    const unsigned char* ximg = wt3_smem + WT3_XI + img * WT2_XROW;
    const unsigned char* dimg = wt3_smem + WT3_DI + img * WT2_DROW;
    const unsigned char* raw = wt3_smem + slot * WT3_RAW;
    unsigned char* xi = wt3_smem + WT3_XI + (img ^ 1) * WT2_XROW;
    unsigned char* di = wt3_smem + WT3_DI + (img ^ 1) * WT2_DROW;
    // ---- block 0 operands, the bias fragment, the raw row
    uint2 f0 = trr(dimg, od[0][0]), f1 = trr(dimg, od[0][1]);
    uint2 r0 = trr(ximg, ox[0][0]), r1 = trr(ximg, ox[0][1]), r2 = trr(ximg, ox[0][2]);
    uint2 g0, g1;
    if (BIAS) { g0 = trr(dimg, obias[0]); g1 = trr(dimg, obias[1]); }
    f32x4 v[2], d;
    if (!(ABL & 2)) {
#pragma unroll
      for (int i = 0; i < 2; ++i) v[i] = *reinterpret_cast<const f32x4*>(raw + (xp0 + 16 * i) * 512 + xq * 16);
      d = *reinterpret_cast<const f32x4*>(raw + WT3_XRAW + dp * 256 + dq * 16);
    }
    ap1[0] = uint4{f0.x, f0.y, f1.x, f1.y};
    {
      const uint4 b0 = {r0.x, r0.y, r1.x, r1.y};
      const uint4 b1 = {__builtin_amdgcn_alignbit(r0.y, r0.x, 16), __builtin_amdgcn_alignbit(r1.x, r0.y, 16),
                        __builtin_amdgcn_alignbit(r1.y, r1.x, 16), __builtin_amdgcn_alignbit(r2.x, r1.y, 16)};
      const uint4 b2 = {r0.y, r1.x, r1.y, r2.x};
      const wg_bf16x8 B0 = __builtin_bit_cast(wg_bf16x8, b0), B1 = __builtin_bit_cast(wg_bf16x8, b1), B2 = __builtin_bit_cast(wg_bf16x8, b2);
      const wg_bf16x8 A0 = __builtin_bit_cast(wg_bf16x8, ap1[0]), A1 = __builtin_bit_cast(wg_bf16x8, a0[0]), A2 = __builtin_bit_cast(wg_bf16x8, am1[0]);
      acc[0] = mm(A0, B0, acc[0]);                              // kernel row 0: output row r + 1
      acc[1] = mm(A0, B1, acc[1]);
      acc[2] = mm(A0, B2, acc[2]);
      acc[3] = mm(A1, B0, acc[3]);                              // kernel row 1: output row r
      acc[4] = mm(A1, B1, acc[4]);
      acc[5] = mm(A1, B2, acc[5]);
      acc[6] = mm(A2, B0, acc[6]);                              // kernel row 2: output row r - 1
      acc[7] = mm(A2, B1, acc[7]);
      acc[8] = mm(A2, B2, acc[8]);
    }
    // ---- block 1 operands; the converted row goes out
    f0 = trr(dimg, od[1][0]); f1 = trr(dimg, od[1][1]);
    r0 = trr(ximg, ox[1][0]); r1 = trr(ximg, ox[1][1]); r2 = trr(ximg, ox[1][2]);
    if (!(ABL & 2)) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
        *reinterpret_cast<uint2*>(xi + wt2_off(xp0 + 16 * i, xq >> 1) + 8 * (xq & 1)) = uint2{wg_pack2(v[i][0], v[i][1]), wg_pack2(v[i][2], v[i][3])};
      const unsigned w0 = wg_pack2(d[0], d[1]), w1 = wg_pack2(d[2], d[3]);
      *reinterpret_cast<uint2*>(di + wt2_off(dp, dq >> 1) + 8 * (dq & 1)) = uint2{w0, w1};
      if (BIAS) {
        const float l0 = d[0] - __builtin_bit_cast(float, w0 << 16), l1 = d[1] - __builtin_bit_cast(float, w0 & 0xffff0000u);
        const float l2 = d[2] - __builtin_bit_cast(float, w1 << 16), l3 = d[3] - __builtin_bit_cast(float, w1 & 0xffff0000u);
        *reinterpret_cast<uint2*>(di + wt2_off(dp, 8 + (dq >> 1)) + 8 * (dq & 1)) = uint2{wg_pack2(l0, l1), wg_pack2(l2, l3)};
      }
    }
    ap1[1] = uint4{f0.x, f0.y, f1.x, f1.y};
    if (BIAS) {
      const uint4 one8 = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
      accb = mm(__builtin_bit_cast(wg_bf16x8, uint4{g0.x, g0.y, g1.x, g1.y}), __builtin_bit_cast(wg_bf16x8, one8), accb);
    }
    {
      const uint4 b0 = {r0.x, r0.y, r1.x, r1.y};
      const uint4 b1 = {__builtin_amdgcn_alignbit(r0.y, r0.x, 16), __builtin_amdgcn_alignbit(r1.x, r0.y, 16),
                        __builtin_amdgcn_alignbit(r1.y, r1.x, 16), __builtin_amdgcn_alignbit(r2.x, r1.y, 16)};
      const uint4 b2 = {r0.y, r1.x, r1.y, r2.x};
      const wg_bf16x8 B0 = __builtin_bit_cast(wg_bf16x8, b0), B1 = __builtin_bit_cast(wg_bf16x8, b1), B2 = __builtin_bit_cast(wg_bf16x8, b2);
      const wg_bf16x8 A0 = __builtin_bit_cast(wg_bf16x8, ap1[1]), A1 = __builtin_bit_cast(wg_bf16x8, a0[1]), A2 = __builtin_bit_cast(wg_bf16x8, am1[1]);
      acc[0] = mm(A0, B0, acc[0]);
      acc[1] = mm(A0, B1, acc[1]);
      acc[2] = mm(A0, B2, acc[2]);
      acc[3] = mm(A1, B0, acc[3]);
      acc[4] = mm(A1, B1, acc[4]);
      acc[5] = mm(A1, B2, acc[5]);
      acc[6] = mm(A2, B0, acc[6]);
      acc[7] = mm(A2, B1, acc[7]);
      acc[8] = mm(A2, B2, acc[8]);
    }
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) { am1[pb] = a0[pb]; a0[pb] = ap1[pb]; }
    // halo pixels 32, 33 of the converted row: the 64 quads left over after two passes of 512 threads (wave 0; uniform branch, kept out of
    // the block above)
    if (wave == 0 && !(ABL & 2)) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(raw + (xp0 + 32) * 512 + xq * 16);
      *reinterpret_cast<uint2*>(xi + wt2_off(xp0 + 32, xq >> 1) + 8 * (xq & 1)) = uint2{wg_pack2(t[0], t[1]), wg_pack2(t[2], t[3])};
    }
  };
  // row 0 of the stream: conversion alone (into images 0)
  auto convert0 = [&]() {
    const unsigned char* raw = wt3_smem;
    unsigned char* xi = wt3_smem + WT3_XI;
    unsigned char* di = wt3_smem + WT3_DI;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if (i == 2 && wave != 0) break;
      const f32x4 t = *reinterpret_cast<const f32x4*>(raw + (xp0 + 16 * i) * 512 + xq * 16);
      *reinterpret_cast<uint2*>(xi + wt2_off(xp0 + 16 * i, xq >> 1) + 8 * (xq & 1)) = uint2{wg_pack2(t[0], t[1]), wg_pack2(t[2], t[3])};
    }
    const f32x4 d = *reinterpret_cast<const f32x4*>(raw + WT3_XRAW + dp * 256 + dq * 16);
    const unsigned w0 = wg_pack2(d[0], d[1]), w1 = wg_pack2(d[2], d[3]);
    *reinterpret_cast<uint2*>(di + wt2_off(dp, dq >> 1) + 8 * (dq & 1)) = uint2{w0, w1};
    if (BIAS) {
      const float l0 = d[0] - __builtin_bit_cast(float, w0 << 16), l1 = d[1] - __builtin_bit_cast(float, w0 & 0xffff0000u);
      const float l2 = d[2] - __builtin_bit_cast(float, w1 << 16), l3 = d[3] - __builtin_bit_cast(float, w1 & 0xffff0000u);
      *reinterpret_cast<uint2*>(di + wt2_off(dp, 8 + (dq >> 1)) + 8 * (dq & 1)) = uint2{wg_pack2(l0, l1), wg_pack2(l2, l3)};
    }
  };

  open_strip(group);
#pragma unroll
  for (int d = 0; d < D; ++d) issue(d);                 // rows 0 .. D - 1 of the stream
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(WT3_NDMA * (D - 1)) : "memory");            // row 0 landed, in every wave
  if (!(ABL & 2)) convert0();
  int slot = 0;                                         // raw slot of row j
  for (int j = 0; j < njobs; ++j) {
    // row j + 1 landed (rows j + 2 .. j + D - 1 stay in flight); this wave's image writes of the last iteration are complete
    if (ABL & 128) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // (ablation: the barrier without the wait for the DMA)
    else if (!(ABL & 64)) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(WT3_NDMA * (D - 2)) : "memory");
    issue(slot);                                        // row j + D into the slot row j left (converted in the last iteration)
    slot = slot + 1 == D ? 0 : slot + 1;
    row(slot, j & 1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the ring's last (dummy) rows: nothing of this workgroup's LDS may still be written
  const int l31 = lane & 31;
  float* out = part + (int64_t)group * 9 * Cout * CIN;
  if (ABL & 16) {                                       // (ablation: one store per accumulator tile instead of 16)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      float v = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) v += acc[t][r];
      out[((int64_t)t * Cout + (co_p * 2 + co_h) * 32 + cmr_mfma_row(0, lane)) * CIN + ci_t * 32 + l31] = v;
    }
    return;
  }
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      out[((int64_t)t * Cout + (co_p * 2 + co_h) * 32 + cmr_mfma_row(r, lane)) * CIN + ci_t * 32 + l31] = acc[t][r];
  if (BIAS && l31 == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) part_b[(int64_t)(4 * group + ci_t) * Cout + (co_p * 2 + co_h) * 32 + cmr_mfma_row(r, lane)] = accb[r];
  }
}

// db[c] = sum over the workgroup partials of the kernel above: 32 channels x 32 slice groups per workgroup, double accumulation, fixed order
__global__ __launch_bounds__(1024) void conv_bias_reduce_kernel(const float* __restrict__ part_b, int nslices, int Cout, float* __restrict__ db) {
  __shared__ double sm[32][32];
  const int o = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + o;
  double s = 0.0;
  if (c < Cout)
    for (int j = g; j < nslices; j += 32) s += (double)part_b[(int64_t)j * Cout + c];
  sm[g][o] = s;
  __syncthreads();
  if (g == 0 && c < Cout) {
#pragma unroll
    for (int j = 1; j < 32; ++j) s += sm[j][o];
    db[c] = (float)s;
  }
}

// Sum of the per-wave partial outputs.  Workgroup = 32 consecutive outputs x 8 slice groups: thread (o, g) adds slices
// g, g + 8, ... in double (coalesced over o), the 8 group sums are combined through LDS in a fixed order.
constexpr int RED_OUT = 32, RED_GRP = 8;

__global__ __launch_bounds__(256) void conv3x3_wgrad_reduce_kernel(const float* __restrict__ part, int nslices, int Cout, int Cin,
                                                                   float* __restrict__ dw) {
  __shared__ double sm[RED_GRP][RED_OUT];
  const int total = 9 * Cout * Cin;
  const int o = threadIdx.x % RED_OUT, g = threadIdx.x / RED_OUT;
  const int i = blockIdx.x * RED_OUT + o;
  double s = 0.0;
  if (i < total) {
    int k = g;
    for (; k + 7 * RED_GRP < nslices; k += 8 * RED_GRP) {         // eight slices in flight per thread, added in slice order
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(int64_t)(k + u * RED_GRP) * total + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += (double)v[u];
    }
    for (; k < nslices; k += RED_GRP) s += (double)part[(int64_t)k * total + i];
  }
  sm[g][o] = s;
  __syncthreads();
  if (g == 0 && i < total) {
#pragma unroll
    for (int k = 1; k < RED_GRP; ++k) s += sm[k][o];
    const int ci = i % Cin, co = (i / Cin) % Cout, tap = i / (Cin * Cout);
    dw[((int64_t)co * Cin + ci) * 9 + tap] = (float)s;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// linear / conv1d(k=1) weight gradient over a row map:  dW[n][k] = sum_r dY[r][n] X[r][k],  db[n] = sum_r dY[r][n]
// grid (row slices, n blocks of NT*32, k blocks of KT*32); workgroup = 4 waves; wave = (n tile, row sub-slice) holding KT
// accumulator tiles.  The slices of one (n block, k block) are summed by linear_wgrad_reduce_kernel (double accumulation,
// fixed order: deterministic).  Small row counts (tokens: 640 .. 2048 rows) still get one slice per 128 rows, so that a
// 1024 x 64 MLP gradient runs on ~100 CUs instead of one.
// ------------------------------------------------------------------------------------------------------------------
template <int NT, int KT>
__global__ __launch_bounds__(256) void linear_wgrad_kernel(const float* __restrict__ dy, int64_t lddy, int n, const float* __restrict__ x,
                                                           int64_t ldx, int k, int64_t rows, float* __restrict__ part,
                                                           float* __restrict__ part_b) {
  constexpr int NSPLIT = 4 / NT;
  constexpr int UNR = 4;                       // row pairs per step: (1 + KT) * UNR loads in flight
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int n_t = wave % NT, split = wave / NT;
  const int n0 = blockIdx.y * (NT * 32), k0 = blockIdx.z * (KT * 32);
  const int64_t nsteps = (rows + 2 * UNR - 1) / (2 * UNR);
  const int64_t per_blk = (nsteps + gridDim.x - 1) / gridDim.x;
  const int64_t s0 = min((int64_t)blockIdx.x * per_blk, nsteps), s1 = min(s0 + per_blk, nsteps);
  const int64_t per_w = (s1 - s0 + NSPLIT - 1) / NSPLIT;
  const int64_t w0 = min(s0 + split * per_w, s1), w1 = min(w0 + per_w, s1);

  f32x16 acc[KT];
#pragma unroll
  for (int t = 0; t < KT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;

  const int ncol = n0 + n_t * 32 + l31;
  const bool n_ok = ncol < n;
  const float* dcol = dy + (n_ok ? ncol : 0);
  bool k_ok[KT];
  const float* xcol[KT];
#pragma unroll
  for (int t = 0; t < KT; ++t) {
    k_ok[t] = k0 + t * 32 + l31 < k;
    xcol[t] = x + (k_ok[t] ? k0 + t * 32 + l31 : 0);
  }

  // two register sets alternate between "consumed now" and "being loaded" (no copies of loaded values, masks applied at
  // consumption: see conv3x3_wgrad_kernel); steps past the end of the slice are masked to zero by their row test
  float sa[2][UNR], sv[2][UNR][KT];
  auto load = [&](float (&a)[UNR], float (&v)[UNR][KT], int64_t step) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int64_t r = (step * UNR + u) * 2 + h;
      const int64_t rc = r < rows ? r : rows - 1;
      a[u] = dcol[rc * lddy];
#pragma unroll
      for (int t = 0; t < KT; ++t) v[u][t] = xcol[t][rc * ldx];
    }
  };
  auto mask = [&](float (&a)[UNR], float (&v)[UNR][KT], int64_t step, bool live) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const bool ok = live && (step * UNR + u) * 2 + h < rows;
      a[u] = ok && n_ok ? a[u] : 0.f;
#pragma unroll
      for (int t = 0; t < KT; ++t) v[u][t] = k_ok[t] ? v[u][t] : 0.f;
    }
  };
  const int64_t last = nsteps - 1;
  if (w0 < w1) load(sa[0], sv[0], w0);
#define CMR_LW_STEP(C, L, S)                                                                 \
  {                                                                                          \
    load(sa[L], sv[L], (S) + 1 < nsteps ? (S) + 1 : last);                                   \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    mask(sa[C], sv[C], (S), (S) < w1);                                                       \
    _Pragma("unroll") for (int u = 0; u < UNR; ++u) {                                        \
      bsum += sa[C][u];                                                                      \
      _Pragma("unroll") for (int t = 0; t < KT; ++t) acc[t] = cmr_mfma32(sa[C][u], sv[C][u][t], acc[t]); \
    }                                                                                        \
  }
  for (int64_t s = w0; s < w1; s += 2) {
    CMR_LW_STEP(0, 1, s)
    CMR_LW_STEP(1, 0, s + 1)
  }
#undef CMR_LW_STEP
  const int64_t nsl = (int64_t)gridDim.x * NSPLIT, sl = (int64_t)blockIdx.x * NSPLIT + split;
  float* out = part + (((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * nsl + sl) * (NT * 32) * (KT * 32);
#pragma unroll
  for (int t = 0; t < KT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = cmr_mfma_row(r, lane);
      out[(int64_t)(n_t * 32 + row) * (KT * 32) + t * 32 + l31] = acc[t][r];
    }
  if (part_b && blockIdx.z == 0) {
    bsum += __shfl_xor(bsum, 32);                                       // the two row parities
    if (h == 0) part_b[((int64_t)blockIdx.y * nsl + sl) * (NT * 32) + n_t * 32 + l31] = bsum;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// The same row-map weight gradient with the operands staged in LDS (round 3), for n, k in {32, 64, 96, 128}: linear_wgrad_kernel
// fetches every MFMA operand with a dword load (lane = channel), reads dY once per k block and X once per n block of the workgroup,
// and sits at ~2 TB/s algorithmic on the 524 288-row point maps (130 us for 268 MB).  Here a workgroup stages blocks of 32 rows of
// dY and X with float4 loads (whole rows, each byte read ONCE per workgroup) into a double-buffered LDS tile, holds the WHOLE
// [n x k] gradient in its four waves' accumulators ((n / 32) (k / 32) / 4 tiles per wave) and walks the row blocks with a static
// stride; operands are ds_read_b32 (lane = channel, the two lane halves = the two rows of a 32x32x2 step; row stride = width + 32
// floats, so the halves fall 32 banks apart).  One __syncthreads per block; 3 workgroups per CU at 64 x 64.  Partials go to the
// workspace in linear_wgrad_reduce_kernel's layout (one n block, one k block, gridDim.x slices).
// ------------------------------------------------------------------------------------------------------------------
#ifndef CMR_LW_DEEP
#define CMR_LW_DEEP 1
#endif
constexpr bool LW_DEEP = CMR_LW_DEEP != 0;
template <int NT, int KT>
__global__ __launch_bounds__(256) void linear_wgrad_lds_kernel(const float* __restrict__ dy, int64_t lddy, const float* __restrict__ x, int64_t ldx,
                                                               int64_t rows, float* __restrict__ part, float* __restrict__ part_b) {
  constexpr int N = 32 * NT, K = 32 * KT, DS = N + 32, XS = K + 32, R = 32;
  constexpr int TPW = NT * KT / 4 > 0 ? NT * KT / 4 : 1;          // tiles per wave (NT KT in {1, 2, 4, ...}: waves beyond the tile count idle)
  constexpr int NLD = (R * N / 4 + 255) / 256, NLX = (R * K / 4 + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) float lw_smem[];
  float* Dl = lw_smem;                                   // [2][R][DS]
  float* Xl = lw_smem + 2 * R * DS;                      // [2][R][XS]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  const int64_t nblocks = (rows + R - 1) / R;

  f32x16 acc[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;
  const bool active = wave * TPW < NT * KT;

  auto load_block = [&](int64_t blk, f32x4 (&dv)[NLD], f32x4 (&xv)[NLX]) {     // branch-free: clamped row, masked at the LDS store
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int e = tid + 256 * i, r = e / (N / 4), c = e % (N / 4);
      int64_t row = blk * R + (r < R ? r : R - 1);
      row = row < rows ? row : rows - 1;
      dv[i] = *reinterpret_cast<const f32x4*>(dy + row * lddy + 4 * c);
    }
#pragma unroll
    for (int i = 0; i < NLX; ++i) {
      const int e = tid + 256 * i, r = e / (K / 4), c = e % (K / 4);
      int64_t row = blk * R + (r < R ? r : R - 1);
      row = row < rows ? row : rows - 1;
      xv[i] = *reinterpret_cast<const f32x4*>(x + row * ldx + 4 * c);
    }
  };
  auto store_block = [&](int64_t blk, int buf, const f32x4 (&dv)[NLD], const f32x4 (&xv)[NLX]) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int e = tid + 256 * i, r = e / (N / 4), c = e % (N / 4);
      if (r < R) *reinterpret_cast<f32x4*>(Dl + (buf * R + r) * DS + 4 * c) = (blk < nblocks && blk * R + r < rows) ? dv[i] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < NLX; ++i) {
      const int e = tid + 256 * i, r = e / (K / 4), c = e % (K / 4);
      if (r < R) *reinterpret_cast<f32x4*>(Xl + (buf * R + r) * XS + 4 * c) = (blk < nblocks && blk * R + r < rows) ? xv[i] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };

  auto multiply = [&](int buf) {
    if (!active) return;
    const float* dl = Dl + buf * R * DS + l31;
    const float* xl = Xl + buf * R * XS + l31;
#pragma unroll
    for (int j = 0; j < R / 2; ++j) {
      const int rr = 2 * j + h;
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        const int tile = wave * TPW + t, nt = tile / KT, kt = tile % KT;
        const float a = dl[rr * DS + 32 * nt];
        const float b = xl[rr * XS + 32 * kt];
        if (kt == 0) bsum += a;
        acc[t] = cmr_mfma32(a, b, acc[t]);
      }
    }
  };
  const int64_t g = gridDim.x;
  auto clampb = [&](int64_t b) { return b < nblocks ? b : nblocks - 1; };
  int64_t blk = blockIdx.x;
  if (LW_DEEP && NLD + NLX <= 8) {
    // Two blocks ahead in registers (blocks blk + g in d1 / x1, blk + 2 g in d2 / x2; blk itself in LDS): one block of MFMAs per wave
    // (1 024 cycles at 64 x 64) is shorter than a loaded HBM round trip, so with ONE block in flight every wave met its own loads again at
    // the LDS store.  The body is unrolled twice so that the two register sets swap roles without moves; blocks are still multiplied
    // in the same order: bit-identical partials.
    f32x4 d1[NLD], x1[NLX], d2[NLD], x2[NLX];
    load_block(clampb(blk), d1, x1);
    store_block(blk, 0, d1, x1);
    load_block(clampb(blk + g), d1, x1);
    __syncthreads();
    for (; blk < nblocks; blk += 2 * g) {
      load_block(clampb(blk + 2 * g), d2, x2);
      multiply(0);
      store_block(blk + g, 1, d1, x1);
      __syncthreads();
      if (blk + g >= nblocks) break;                                // (uniform over the workgroup)
      load_block(clampb(blk + 3 * g), d1, x1);
      multiply(1);
      store_block(blk + 2 * g, 0, d2, x2);
      __syncthreads();
    }
  } else {
    f32x4 dv[NLD], xv[NLX];
    load_block(clampb(blk), dv, xv);
    store_block(blk, 0, dv, xv);
    __syncthreads();
    int buf = 0;
    for (; blk < nblocks; blk += g) {
      const int64_t nb = blk + g;
      load_block(clampb(nb), dv, xv);                               // next block of this workgroup: in flight under the MFMAs
      multiply(buf);
      store_block(nb, buf ^ 1, dv, xv);                             // the other buffer: last read one iteration ago, behind a barrier
      __syncthreads();
      buf ^= 1;
    }
  }
  if (!active) return;
  float* out = part + (int64_t)blockIdx.x * N * K;
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int tile = wave * TPW + t, nt = tile / KT, kt = tile % KT;
#pragma unroll
    for (int r = 0; r < 16; ++r) out[(int64_t)(nt * 32 + cmr_mfma_row(r, lane)) * K + kt * 32 + l31] = acc[t][r];
  }
  if (part_b) {
    // bsum holds this lane's rows of channel(s) 32 nt + l31 for the tiles with kt == 0: with TPW > 1 and KT < TPW several nt share the
    // register -- handled by the host (part_b only with KT >= TPW, i.e. one nt per wave among its kt == 0 tiles)
    bsum += __shfl_xor(bsum, 32);
    const int tile0 = wave * TPW, nt0 = tile0 / KT;
    if (h == 0 && tile0 % KT == 0) part_b[(int64_t)blockIdx.x * N + nt0 * 32 + l31] = bsum;
  }
}

// outputs [0, n k): dW entries; [n k, n k + n): db entries (when part_b)
// (round 3) A thread used to walk its 96 slices of a 768-slice sum one dependent load at a time: 130 workgroups of 256 threads took 30-35 us
// for 12.8 MB (profiles/r03_pmc_lwgrad.txt: 92 % of the wave cycles waiting on memory) -- more than a third of the weight gradient of a
// 524 288-row map.  Now 32 slice groups (1 024 threads) and LRED_U independent loads in flight per thread; sums still in double, fixed order.
constexpr int LRED_GRP = 32, LRED_U = 8;
__global__ __launch_bounds__(RED_OUT * LRED_GRP) void linear_wgrad_reduce_kernel(const float* __restrict__ part, const float* __restrict__ part_b, int nslices,
                                                                  int npad, int kpad, int nblk, int n, int k, float* __restrict__ dw,
                                                                  int64_t lddw, int accumulate, float* __restrict__ db, int acc_db) {
  __shared__ double sm[LRED_GRP][RED_OUT];
  const int o = threadIdx.x % RED_OUT, g = threadIdx.x / RED_OUT;
  const int64_t i = (int64_t)blockIdx.x * RED_OUT + o;
  const int64_t nk = (int64_t)n * k, total = nk + (part_b ? n : 0);
  double s = 0.0;
  int row = 0, col = 0;
  const float* p = part;
  int64_t stride = 0;
  if (i < nk) {
    row = (int)(i / k), col = (int)(i - (int64_t)row * k);
    const int nb = row / npad, kb = col / kpad;
    p = part + ((int64_t)kb * nblk + nb) * nslices * npad * kpad + (int64_t)(row - nb * npad) * kpad + (col - kb * kpad);
    stride = (int64_t)npad * kpad;
  } else if (i < total) {
    row = (int)(i - nk);
    const int nb = row / npad;
    p = part_b + (int64_t)nb * nslices * npad + (row - nb * npad);
    stride = npad;
  }
  if (i < total)
    for (int j0 = g; j0 < nslices; j0 += LRED_GRP * LRED_U) {
      float v[LRED_U];
#pragma unroll
      for (int u = 0; u < LRED_U; ++u) {
        const int j = j0 + u * LRED_GRP;
        v[u] = p[(int64_t)(j < nslices ? j : j0) * stride];          // (branch-free: a slice past the end re-reads j0 and is dropped below)
      }
#pragma unroll
      for (int u = 0; u < LRED_U; ++u) s += j0 + u * LRED_GRP < nslices ? (double)v[u] : 0.0;
    }
  sm[g][o] = s;
  __syncthreads();
  if (g == 0 && i < total) {
#pragma unroll
    for (int j = 1; j < LRED_GRP; ++j) s += sm[j][o];
    if (i < nk) {
      float* d = dw + (int64_t)row * lddw + col;
      *d = accumulate ? *d + (float)s : (float)s;
    } else {
      db[row] = acc_db ? db[row] + (float)s : (float)s;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// conv3x3 weights [Co][Ci][3][3] (the nn.Conv2d parameter, read straight from the flat bucket) -> the operand layouts
// of the forward kernels: w9 [9][Co'][Ci'] and Winograd U = G g G^T as MFMA A fragments [16][Co'/32][Ci'/8][64][4].
// transpose != 0 gives the data-gradient convolution: W'[ci][co][ky][kx] = W[co][ci][2-ky][2-kx].
// ------------------------------------------------------------------------------------------------------------------
typedef __bf16 wg_bf16;

__device__ __forceinline__ void pack_conv3x3_one(const float* __restrict__ w, int Co, int Ci, int transpose, float* __restrict__ w9,
                                                 float* __restrict__ ufrag, wg_bf16* __restrict__ bfrag, int nt, int i) {
  const int Cop = transpose ? Ci : Co, Cip = transpose ? Co : Ci;
  if (i >= Cop * Cip) return;
  const int o = i / Cip, c = i - o * Cip;
  float g[3][3];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
      g[ky][kx] = transpose ? w[((int64_t)c * Ci + o) * 9 + (2 - ky) * 3 + (2 - kx)] : w[((int64_t)o * Ci + c) * 9 + ky * 3 + kx];
#pragma unroll
  for (int t = 0; t < 9; ++t) w9[((int64_t)t * Cop + o) * Cip + c] = g[t / 3][t % 3];
  if (bfrag) {
    // bf16 A fragments of cmr_conv3x3_bf16_nhwc_f32: [Co'/(32 nt)][9][Ci'/16][nt][64 lanes][8], lane = 32 h + (o & 31) holding
    // W'[o][c = 16 ks + 8 h + j]
    const int grp = o / (32 * nt), tt = (o / 32) % nt, ks = c >> 4, hh = (c >> 3) & 1, j = c & 7;
#pragma unroll
    for (int t = 0; t < 9; ++t)
      bfrag[(((((int64_t)grp * 9 + t) * (Cip / 16) + ks) * nt + tt) * 64 + hh * 32 + (o & 31)) * 8 + j] = (wg_bf16)g[t / 3][t % 3];
  }
  if (!ufrag) return;
  // G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
  float gg[4][3];
#pragma unroll
  for (int l = 0; l < 3; ++l) {
    gg[0][l] = g[0][l];
    gg[1][l] = 0.5f * g[0][l] + 0.5f * g[1][l] + 0.5f * g[2][l];
    gg[2][l] = 0.5f * g[0][l] - 0.5f * g[1][l] + 0.5f * g[2][l];
    gg[3][l] = g[2][l];
  }
  const int t32 = o >> 5, l = o & 31, kg = c >> 3, hh = (c >> 2) & 1, e = c & 3;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const float u0 = gg[a][0];
    const float u1 = 0.5f * gg[a][0] + 0.5f * gg[a][1] + 0.5f * gg[a][2];
    const float u2 = 0.5f * gg[a][0] - 0.5f * gg[a][1] + 0.5f * gg[a][2];
    const float u3 = gg[a][2];
    const float uu[4] = {u0, u1, u2, u3};
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int pos = a * 4 + b;
      ufrag[((((int64_t)(pos * (Cop / 32) + t32) * (Cip / 8) + kg) * 2 + hh) * 32 + l) * 4 + e] = uu[b];
    }
  }
}

__global__ __launch_bounds__(256) void pack_conv3x3_kernel(const float* __restrict__ w, int Co, int Ci, int transpose, float* __restrict__ w9,
                                                           float* __restrict__ ufrag, wg_bf16* __restrict__ bfrag, int nt) {
  pack_conv3x3_one(w, Co, Ci, transpose, w9, ufrag, bfrag, nt, blockIdx.x * blockDim.x + threadIdx.x);
}

// every convolution of a training step (both orientations) in ONE launch: table [nslots][8] int64 {src offset, Co, Ci, transpose, w9
// offset, U offset (-1: none), bf16 offset in bf16 elements (-1: none), nt}; blockIdx.y = slot
__global__ __launch_bounds__(256) void pack_conv3x3_slots_kernel(const float* __restrict__ src, float* __restrict__ dst, wg_bf16* __restrict__ dst_bf,
                                                                 const int64_t* __restrict__ table) {
  const int64_t* e = table + (int64_t)blockIdx.y * 8;
  pack_conv3x3_one(src + e[0], (int)e[1], (int)e[2], (int)e[3], dst + e[4], e[5] >= 0 ? dst + e[5] : nullptr,
                   (e[6] >= 0 && dst_bf) ? dst_bf + e[6] : nullptr, (int)e[7], blockIdx.x * blockDim.x + threadIdx.x);
}

inline int wgrad_slices(int64_t work_items) {
  int64_t s = work_items / 64;
  if (s > 256) s = 256;
  if (s < 1) s = 1;
  return (int)s;
}

}  // namespace

// persistent workgroups of the LDS-staged kernel per cout tile, and rows per column strip
struct WgLdsPlan { int use, groups, rps; };
#ifdef CMR_AB_SWITCHES          // variant switches: A/B build only (libcmr_hip_ab.so)
static int g_wgrad_lds = 1;
#else
static constexpr int g_wgrad_lds = 1;
#endif
inline WgLdsPlan wgrad_lds_plan(int B, int H, int W, int Cin, int Cout) {
  WgLdsPlan p{0, 0, 0};
  // Measured (tools/wgrad_bench.py, profiles/r03_wgrad_bench.txt; minibatch 10, 128 -> 128): 44x152 263 -> 227 us, 22x76 111 -> 87 us,
  // 11x38 48 -> 45 us, but 88x304 902 -> 979 us and 8 x 160x512 64 -> 64 558 -> 599 us: both kernels sit at 0.51-0.56 of the nominal
  // fp32 matrix peak on large maps (one wave per SIMD and a barrier per image row here, two waves per SIMD and L2 operand traffic
  // there), so the staged kernel takes the maps whose row tiles the direct kernel's long slices balance badly, and nothing else
  const int64_t npx = (int64_t)B * H * W;
  if (!g_wgrad_lds || (Cin != 128 && Cin != 64) || npx < 4096 || npx > 150000) return p;
  const int per_cu = Cin == 128 ? 1 : 2;                // LDS: 95 KB / 60 KB per workgroup
  int groups = 256 * per_cu / (Cout / 32);
  const int ntx = (W + 31) / 32;
  int rps = (int)(((int64_t)B * ntx * H) / ((int64_t)8 * groups));       // ~8 strips per workgroup (balance); >= 4 rows (halo rows amortised)
  if (rps < 4) rps = 4;
  if (rps > H) rps = H;
  const int64_t nstrips = (int64_t)B * ntx * ((H + rps - 1) / rps);
  if (groups > nstrips) groups = (int)nstrips;
  p.use = 1; p.groups = groups; p.rps = rps;
  return p;
}

// bf16 weight gradient: maps of at least this many pixels take the transposed-read kernel (below, the fixed costs of either kernel -- partial
// write-back, reduction -- dominate and the first-generation kernel's finer slices win)
constexpr int64_t WGB_TR_MIN_PX = 32768;
#ifndef CMR_WGRAD_BF16_GEN
#define CMR_WGRAD_BF16_GEN 1            // 0: first generation everywhere, 1: transposed reads (rows through registers: the library's choice),
                                        // 2: + rows by LDS-DMA (A/B library only: measured 15 - 25 % slower, see the kernel's header)
#endif
#ifndef CMR_WGRAD_BF16_SPW
#define CMR_WGRAD_BF16_SPW 4            // generation 3: strips per workgroup the row count of a strip is sized for
#endif
#ifndef CMR_WGRAD_BF16_SPW2
#define CMR_WGRAD_BF16_SPW2 8           // generation 2: the same (every strip refills its two-row prefetch: fewer, longer strips against balance)
#endif
#ifdef CMR_AB_SWITCHES
static int g_wgrad_tr = CMR_WGRAD_BF16_GEN;
static int g_wgrad_spw = CMR_WGRAD_BF16_SPW;
static int g_wgrad_spw2 = CMR_WGRAD_BF16_SPW2;
extern "C" int cmr_set_wgrad_bf16_strips(int per_workgroup) {      // > 0: third generation; < 0: second generation (- per_workgroup)
  const int old = g_wgrad_spw;
  if (per_workgroup > 0) g_wgrad_spw = per_workgroup;
  if (per_workgroup < 0) g_wgrad_spw2 = -per_workgroup;
  return old;
}
#else
static constexpr int g_wgrad_tr = CMR_WGRAD_BF16_GEN;
static constexpr int g_wgrad_spw = CMR_WGRAD_BF16_SPW;
static constexpr int g_wgrad_spw2 = CMR_WGRAD_BF16_SPW2;
#endif
#ifdef CMR_AB_SWITCHES
static int g_lwgrad_lds = 1;
#else
static constexpr int g_lwgrad_lds = 1;
#endif
// Measured (tools/lwgrad_bench.py, profiles/r03_lwgrad_bench.txt): 524 288 rows 64x64 122 -> 84 us, 128x64 218 -> 129 us; 163 840 rows 5-20 %
// faster; at 40 960 rows and below the direct kernel's finer slices win (31 vs 38 us): the staged kernel takes maps of >= 65 536 rows.
#ifdef CMR_AB_SWITCHES
static int64_t g_lwgrad_min_rows = 65536;
extern "C" int cmr_set_linear_wgrad_variant(int lds_staged) {
  const int old = g_lwgrad_lds;
  g_lwgrad_lds = lds_staged ? 1 : 0;
  g_lwgrad_min_rows = lds_staged == 2 ? 8192 : 65536;      // 2: also mid-size maps (tests of the ragged cases)
  return old;
}

extern "C" int cmr_set_wgrad_bf16_variant(int transposed_reads) {
  const int old = g_wgrad_tr;
  g_wgrad_tr = transposed_reads < 0 ? 0 : (transposed_reads > 2 && transposed_reads < 16 ? 2 : transposed_reads);      // 16 + mask: ablations
  return old;
}
extern "C" int cmr_set_wgrad_variant(int lds_staged) {
  const int old = g_wgrad_lds;
  g_wgrad_lds = lds_staged ? 1 : 0;
  return old;
}
#else
static constexpr int64_t g_lwgrad_min_rows = 65536;
#endif

extern "C" int64_t cmr_conv3x3_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
  const int nci = Cin / 32;
  const int nsplit = nci > 0 && nci <= 4 ? 4 / nci : 1;
  int slices = wgrad_slices(((int64_t)B * H * W + 1) / 2);
  const int groups = 512;                               // upper bound of wgrad_lds_plan().groups, whichever variant is switched on
  if (slices < groups) slices = groups;
  return (int64_t)slices * nsplit * 9 * Cout * Cin * (int64_t)sizeof(float);
}

extern "C" int cmr_conv3x3_wgrad_f32(const float* x, const float* dy, int B, int H, int W, int Cin, int Cout, float* dw, void* ws,
                                     int64_t ws_bytes, hipStream_t stream) {
  CMR_REQUIRE(x && dy && dw && ws && B > 0 && H > 0 && W >= 2);
  CMR_REQUIRE((Cout == 32 || Cout == 64 || Cout == 128 || Cout == 256) && (Cin == 32 || Cin == 64 || Cin == 128));
  CMR_REQUIRE((int64_t)B * H * W * (Cin > Cout ? Cin : Cout) < (int64_t)0x7fffffff * 16);
  CMR_REQUIRE((int64_t)B * H * W < 0x7fffffff);
  const int nci = Cin / 32, nsplit = 4 / nci;
  float* part = (float*)ws;
  const WgLdsPlan lp = wgrad_lds_plan(B, H, W, Cin, Cout);
  if (lp.use) {
    CMR_REQUIRE(ws_bytes >= (int64_t)lp.groups * nsplit * 9 * Cout * Cin * (int64_t)sizeof(float));
    const size_t smem = ((size_t)4 * 34 * (Cin + 32) + 2 * 32 * 32) * sizeof(float);
    dim3 grid(lp.groups, Cout / 32);
    if (nci == 4) {
      static CmrSmemCache granted{};
      if (cmr_grant_smem(reinterpret_cast<const void*>(conv3x3_wgrad_lds_kernel<4>), smem, granted) != CMR_OK) return CMR_ELAUNCH;
      hipLaunchKernelGGL(conv3x3_wgrad_lds_kernel<4>, grid, dim3(256), smem, stream, x, dy, B, H, W, Cout, lp.rps, part);
    } else {
      static CmrSmemCache granted{};
      if (cmr_grant_smem(reinterpret_cast<const void*>(conv3x3_wgrad_lds_kernel<2>), smem, granted) != CMR_OK) return CMR_ELAUNCH;
      hipLaunchKernelGGL(conv3x3_wgrad_lds_kernel<2>, grid, dim3(256), smem, stream, x, dy, B, H, W, Cout, lp.rps, part);
    }
    hipLaunchKernelGGL(conv3x3_wgrad_reduce_kernel, dim3((9 * Cout * Cin + RED_OUT - 1) / RED_OUT), dim3(256), 0, stream, (const float*)part,
                       lp.groups * nsplit, Cout, Cin, dw);
    return cmr_launch_status();
  }
  const int slices = wgrad_slices(((int64_t)B * H * W + 1) / 2);
  CMR_REQUIRE(ws_bytes >= (int64_t)slices * nsplit * 9 * Cout * Cin * (int64_t)sizeof(float));
  dim3 grid(slices, Cout / 32);
  if (nci == 4)
    hipLaunchKernelGGL(conv3x3_wgrad_kernel<4>, grid, dim3(256), 0, stream, x, dy, B, H, W, Cin, Cout, part);
  else if (nci == 2)
    hipLaunchKernelGGL(conv3x3_wgrad_kernel<2>, grid, dim3(256), 0, stream, x, dy, B, H, W, Cin, Cout, part);
  else
    hipLaunchKernelGGL(conv3x3_wgrad_kernel<1>, grid, dim3(256), 0, stream, x, dy, B, H, W, Cin, Cout, part);
  hipLaunchKernelGGL(conv3x3_wgrad_reduce_kernel, dim3((9 * Cout * Cin + RED_OUT - 1) / RED_OUT), dim3(256), 0, stream, (const float*)part,
                     slices * nsplit, Cout, Cin, dw);
  return cmr_launch_status();
}

// stride-2 convolution: x [B][H][W][Cin] (H, W even), dy [B][H/2][W/2][Cout]; workspace = cmr_conv3x3_wgrad_workspace_bytes(B, H/2, W/2, ...)
extern "C" int cmr_conv3x3_wgrad_s2_f32(const float* x, const float* dy, int B, int H, int W, int Cin, int Cout, float* dw, void* ws,
                                        int64_t ws_bytes, hipStream_t stream) {
  CMR_REQUIRE(x && dy && dw && ws && B > 0 && H >= 2 && W >= 4 && H % 2 == 0 && W % 2 == 0);
  CMR_REQUIRE((Cout == 32 || Cout == 64 || Cout == 128 || Cout == 256) && (Cin == 32 || Cin == 64 || Cin == 128));
  CMR_REQUIRE((int64_t)B * H * W * (Cin > Cout ? Cin : Cout) < (int64_t)0x7fffffff * 16 && (int64_t)B * H * W < 0x7fffffff);
  const int nci = Cin / 32, nsplit = 4 / nci;
  const int slices = wgrad_slices(((int64_t)B * (H / 2) * (W / 2) + 1) / 2);
  CMR_REQUIRE(ws_bytes >= (int64_t)slices * nsplit * 9 * Cout * Cin * (int64_t)sizeof(float));
  float* part = (float*)ws;
  dim3 grid(slices, Cout / 32);
  if (nci == 4) hipLaunchKernelGGL(conv3x3_wgrad_s2_kernel<4>, grid, dim3(256), 0, stream, x, dy, B, H, W, Cin, Cout, part);
  else if (nci == 2) hipLaunchKernelGGL(conv3x3_wgrad_s2_kernel<2>, grid, dim3(256), 0, stream, x, dy, B, H, W, Cin, Cout, part);
  else hipLaunchKernelGGL(conv3x3_wgrad_s2_kernel<1>, grid, dim3(256), 0, stream, x, dy, B, H, W, Cin, Cout, part);
  hipLaunchKernelGGL(conv3x3_wgrad_reduce_kernel, dim3((9 * Cout * Cin + RED_OUT - 1) / RED_OUT), dim3(256), 0, stream, (const float*)part,
                     slices * nsplit, Cout, Cin, dw);
  return cmr_launch_status();
}

extern "C" int cmr_conv3x3_wgrad_bias_bf16_f32(const float* x, const float* dy, int B, int H, int W, int Cin, int Cout, float* dw, float* db,
                                               void* ws, int64_t ws_bytes, hipStream_t stream);
extern "C" int cmr_conv3x3_wgrad_bf16_f32(const float* x, const float* dy, int B, int H, int W, int Cin, int Cout, float* dw, void* ws,
                                          int64_t ws_bytes, hipStream_t stream) {
  return cmr_conv3x3_wgrad_bias_bf16_f32(x, dy, B, H, W, Cin, Cout, dw, nullptr, ws, ws_bytes, stream);
}

static int wgrad_bias_bf16(const float* x, const float* xscale, const float* xshift, float xslope, const float* dy, int B, int H, int W, int Cin,
                           int Cout, float* dw, float* db, void* ws, int64_t ws_bytes, hipStream_t stream);
extern "C" int cmr_conv3x3_wgrad_bias_bf16_f32(const float* x, const float* dy, int B, int H, int W, int Cin, int Cout, float* dw, float* db,
                                               void* ws, int64_t ws_bytes, hipStream_t stream) {
  return wgrad_bias_bf16(x, nullptr, nullptr, 1.f, dy, B, H, W, Cin, Cout, dw, db, ws, ws_bytes, stream);
}
// The same with x given as a BatchNorm INPUT: the operand is lrelu_{xslope}(x * xscale + xshift) per input channel, formed in the staging
// pass (the layer's forward ran cmr_conv3x3_bf16_pro_nhwc_f32 and never stored the activation).  Served where the second-generation
// kernel runs (Cin = 128, Cout % 64 == 0, >= 32 768 pixels); CMR_EUNSUPPORTED otherwise.
extern "C" int cmr_conv3x3_wgrad_bias_bf16_pro_f32(const float* x, const float* xscale, const float* xshift, float xslope, const float* dy, int B,
                                                   int H, int W, int Cin, int Cout, float* dw, float* db, void* ws, int64_t ws_bytes,
                                                   hipStream_t stream) {
  CMR_REQUIRE(xscale && xshift && cmr_aligned16(xscale) && cmr_aligned16(xshift));
  if (!(xslope >= 0.f && xslope <= 1.f)) return CMR_EUNSUPPORTED;
  return wgrad_bias_bf16(x, xscale, xshift, xslope, dy, B, H, W, Cin, Cout, dw, db, ws, ws_bytes, stream);
}

static int wgrad_bias_bf16(const float* x, const float* xscale, const float* xshift, float xslope, const float* dy, int B, int H, int W, int Cin,
                           int Cout, float* dw, float* db, void* ws, int64_t ws_bytes, hipStream_t stream) {
  CMR_REQUIRE(x && dy && dw && ws && B > 0 && H > 0 && W >= 2);
  CMR_REQUIRE((Cout == 32 || Cout == 64 || Cout == 128 || Cout == 256) && (Cin == 64 || Cin == 128));
  CMR_REQUIRE((int64_t)B * H * W * (Cin > Cout ? Cin : Cout) < (int64_t)0x7fffffff);       // 32-bit element offsets within an image batch
  const int nci = Cin / 32, nsplit = 4 / nci;
  if (xscale && !(Cin == 128 && Cout % 64 == 0 && (int64_t)B * H * W >= WGB_TR_MIN_PX && g_wgrad_tr == 1)) return CMR_EUNSUPPORTED;
  if (Cin == 128 && Cout % 64 == 0 && (int64_t)B * H * W >= WGB_TR_MIN_PX && g_wgrad_tr) {
    // second-generation kernel (hardware-transposed operand reads, 64 couts per 8-wave workgroup): one persistent workgroup per CU
    const int ncp = Cout / 64;
    int groups = 256 / ncp;
    const int ntx = (W + WT2_TW - 1) / WT2_TW;
    const bool gen3 = g_wgrad_tr >= 2;
    // strips per workgroup: ~8 (generation 2), ~4 of twice the rows (generation 3: every strip costs it two rows of zero fragments)
    int rps = (int)(((int64_t)B * ntx * H) / ((int64_t)(gen3 ? g_wgrad_spw : g_wgrad_spw2) * groups));
    if (rps < 4) rps = 4;
    if (rps > H) rps = H;
    const int64_t nstrips = (int64_t)B * ntx * ((H + rps - 1) / rps);
    if (groups > nstrips) groups = (int)nstrips;
    const int64_t part_floats = (int64_t)groups * 9 * Cout * Cin;
    const int bslices = gen3 ? 4 * groups : groups;          // bias partials: generation 3 keeps one per ci wave
    CMR_REQUIRE(ws_bytes >= (part_floats + (db ? (int64_t)bslices * Cout : 0)) * (int64_t)sizeof(float));
    float* part = (float*)ws;
    float* part_b = db ? part + part_floats : nullptr;
    static CmrSmemCache granted_b{}, granted_n{}, granted_b3{}, granted_n3{};
#ifdef CMR_AB_SWITCHES
    if (g_wgrad_tr >= 16) {                               // ablations of the third-generation kernel (tools/wgrad_dma_ablate.py)
      static CmrSmemCache g_abl[32]{};
      const void* f[32] = {
          (const void*)conv3x3_wgrad_bf16_dma_kernel<true, 0>, (const void*)conv3x3_wgrad_bf16_dma_kernel<true, 1>, (const void*)conv3x3_wgrad_bf16_dma_kernel<true, 2>,
          (const void*)conv3x3_wgrad_bf16_dma_kernel<true, 3>, (const void*)conv3x3_wgrad_bf16_dma_kernel<true, 4>, nullptr, nullptr, nullptr,
          (const void*)conv3x3_wgrad_bf16_dma_kernel<true, 8>, nullptr, nullptr, nullptr, (const void*)conv3x3_wgrad_bf16_dma_kernel<true, 12>, nullptr, nullptr,
          (const void*)conv3x3_wgrad_bf16_dma_kernel<true, 15>, (const void*)conv3x3_wgrad_bf16_dma_kernel<true, 16>,
          nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
          (const void*)conv3x3_wgrad_bf16_dma_kernel<true, 31>};
      int m = g_wgrad_tr - 16;
      const void* fx[4] = {(const void*)conv3x3_wgrad_bf16_dma_kernel<true, 32>, (const void*)conv3x3_wgrad_bf16_dma_kernel<true, 95>,
                           (const void*)conv3x3_wgrad_bf16_dma_kernel<true, 128>, (const void*)conv3x3_wgrad_bf16_dma_kernel<true, 256>};
      static CmrSmemCache gx[4]{};
      const int xi = m == 32 ? 0 : (m == 95 ? 1 : (m == 128 ? 2 : (m == 256 ? 3 : -1)));
      if (xi >= 0) {
        if (!db || cmr_grant_smem(fx[xi], WT3_SMEM, gx[xi]) != CMR_OK) return CMR_ELAUNCH;
        void* args2[] = {(void*)&x, (void*)&dy, (void*)&B, (void*)&H, (void*)&W, (void*)&Cout, (void*)&rps, (void*)&groups, (void*)&part, (void*)&part_b};
        if (hipLaunchKernel(fx[xi], dim3(groups * ncp), dim3(512), args2, WT3_SMEM, stream) != hipSuccess) return CMR_ELAUNCH;
        return cmr_launch_status();
      }
      if (m < 0 || m > 31 || !f[m] || !db) return CMR_EINVAL;
      if (cmr_grant_smem(f[m], WT3_SMEM, g_abl[m]) != CMR_OK) return CMR_ELAUNCH;
      void* args[] = {(void*)&x, (void*)&dy, (void*)&B, (void*)&H, (void*)&W, (void*)&Cout, (void*)&rps, (void*)&groups, (void*)&part, (void*)&part_b};
      if (hipLaunchKernel(f[m], dim3(groups * ncp), dim3(512), args, WT3_SMEM, stream) != hipSuccess) return CMR_ELAUNCH;
      return cmr_launch_status();
    }
#endif
#ifdef CMR_AB_SWITCHES
    if (g_wgrad_tr == 2) {
      if (db) {
        if (cmr_grant_smem(reinterpret_cast<const void*>(conv3x3_wgrad_bf16_dma_kernel<true>), WT3_SMEM, granted_b3) != CMR_OK) return CMR_ELAUNCH;
        hipLaunchKernelGGL(conv3x3_wgrad_bf16_dma_kernel<true>, dim3(groups * ncp), dim3(512), WT3_SMEM, stream, x, dy, B, H, W, Cout, rps, groups, part, part_b);
      } else {
        if (cmr_grant_smem(reinterpret_cast<const void*>(conv3x3_wgrad_bf16_dma_kernel<false>), WT3_SMEM, granted_n3) != CMR_OK) return CMR_ELAUNCH;
        hipLaunchKernelGGL(conv3x3_wgrad_bf16_dma_kernel<false>, dim3(groups * ncp), dim3(512), WT3_SMEM, stream, x, dy, B, H, W, Cout, rps, groups, part, part_b);
      }
    } else
#endif
    if (xscale) {
      static CmrSmemCache granted_bp{}, granted_np{};
      constexpr int smem = WT2_SMEM + 1024;
      if (db) {
        if (cmr_grant_smem(reinterpret_cast<const void*>(conv3x3_wgrad_bf16_tr_kernel<true, true>), smem, granted_bp) != CMR_OK) return CMR_ELAUNCH;
        hipLaunchKernelGGL((conv3x3_wgrad_bf16_tr_kernel<true, true>), dim3(groups * ncp), dim3(512), smem, stream, x, dy, B, H, W, Cout, rps, groups, part,
                           part_b, xscale, xshift, xslope);
      } else {
        if (cmr_grant_smem(reinterpret_cast<const void*>(conv3x3_wgrad_bf16_tr_kernel<false, true>), smem, granted_np) != CMR_OK) return CMR_ELAUNCH;
        hipLaunchKernelGGL((conv3x3_wgrad_bf16_tr_kernel<false, true>), dim3(groups * ncp), dim3(512), smem, stream, x, dy, B, H, W, Cout, rps, groups, part,
                           part_b, xscale, xshift, xslope);
      }
    } else if (db) {
      if (cmr_grant_smem(reinterpret_cast<const void*>(conv3x3_wgrad_bf16_tr_kernel<true>), WT2_SMEM, granted_b) != CMR_OK) return CMR_ELAUNCH;
      hipLaunchKernelGGL(conv3x3_wgrad_bf16_tr_kernel<true>, dim3(groups * ncp), dim3(512), WT2_SMEM, stream, x, dy, B, H, W, Cout, rps, groups, part, part_b);
    } else {
      if (cmr_grant_smem(reinterpret_cast<const void*>(conv3x3_wgrad_bf16_tr_kernel<false>), WT2_SMEM, granted_n) != CMR_OK) return CMR_ELAUNCH;
      hipLaunchKernelGGL(conv3x3_wgrad_bf16_tr_kernel<false>, dim3(groups * ncp), dim3(512), WT2_SMEM, stream, x, dy, B, H, W, Cout, rps, groups, part, part_b);
    }
    hipLaunchKernelGGL(conv3x3_wgrad_reduce_kernel, dim3((9 * Cout * Cin + RED_OUT - 1) / RED_OUT), dim3(256), 0, stream, (const float*)part, groups, Cout, Cin, dw);
    if (db) hipLaunchKernelGGL(conv_bias_reduce_kernel, dim3((Cout + 31) / 32), dim3(1024), 0, stream, (const float*)part_b, bslices, Cout, db);
    return cmr_launch_status();
  }
  // persistent workgroups, one per CU (288 accumulator registers per wave at two cout tiles), over all cout groups; ~8 column strips
  // of >= 4 rows per workgroup
  // Cin = 128: 32-pixel row tiles and two workgroups per CU (see the kernel's OCC note); Cin = 64: 64-pixel row tiles unless they pad the
  // width more than 32-pixel ones (152 -> 192 vs 160)
  const int occ = nci == 4 ? 2 : 1;
  const int tw = occ == 2 ? 32 : (((W + 63) / 64 * 64 <= (W + 31) / 32 * 32) ? 64 : 32);
  // two cout tiles per workgroup (288 accumulator registers) spill at Cin = 128 and with 64-pixel tiles (measured slower): narrow Cin = 64 maps only
  // (with the bias sums the two-tile instance spills 176 registers: one tile then)
  const int nco = (nci == 2 && tw == 32 && Cout % 64 == 0 && !db) ? 2 : 1;
  int groups = 256 * occ / (Cout / (32 * nco));
  const int ntx = (W + tw - 1) / tw;
  int rps = (int)(((int64_t)B * ntx * H) / ((int64_t)8 * groups));
  if (rps < 4) rps = 4;
  if (rps > H) rps = H;
  const int64_t nstrips = (int64_t)B * ntx * ((H + rps - 1) / rps);
  if (groups > nstrips) groups = (int)nstrips;
  if (groups > 512) groups = 512;                        // workspace bound of cmr_conv3x3_wgrad_workspace_bytes
  const int64_t part_floats = (int64_t)groups * nsplit * 9 * Cout * Cin;
  CMR_REQUIRE(ws_bytes >= (part_floats + (db ? (int64_t)groups * nsplit * Cout : 0)) * (int64_t)sizeof(float));
  const int rs = tw / 2 + 5;
  const size_t smem = ((size_t)4 * Cin * rs + (db ? 4 : 2) * 32 * nco * rs) * sizeof(unsigned);
  float* part = (float*)ws;
  float* part_b = db ? part + part_floats : nullptr;
  dim3 grid(groups, Cout / (32 * nco));
#define CMR_WGB_LAUNCH(NCI_, NCO_, TW_, BIAS_, OCC_)                                                                                  \
  {                                                                                                                                   \
    static CmrSmemCache granted{};                                                                                                    \
    if (cmr_grant_smem(reinterpret_cast<const void*>(conv3x3_wgrad_bf16_kernel<NCI_, NCO_, TW_, BIAS_, OCC_>), smem, granted) != CMR_OK) return CMR_ELAUNCH; \
    hipLaunchKernelGGL((conv3x3_wgrad_bf16_kernel<NCI_, NCO_, TW_, BIAS_, OCC_>), grid, dim3(256), smem, stream, x, dy, B, H, W, Cout, rps, part, part_b);   \
  }
  if (db) {
    if (nci == 4) CMR_WGB_LAUNCH(4, 1, 32, true, 2)
    else if (tw == 64) CMR_WGB_LAUNCH(2, 1, 64, true, 1)
    else CMR_WGB_LAUNCH(2, 1, 32, true, 1)
  } else if (nci == 4) CMR_WGB_LAUNCH(4, 1, 32, false, 2)
  else if (nco == 2) CMR_WGB_LAUNCH(2, 2, 32, false, 1)
  else if (tw == 64) CMR_WGB_LAUNCH(2, 1, 64, false, 1)
  else CMR_WGB_LAUNCH(2, 1, 32, false, 1)
#undef CMR_WGB_LAUNCH
  hipLaunchKernelGGL(conv3x3_wgrad_reduce_kernel, dim3((9 * Cout * Cin + RED_OUT - 1) / RED_OUT), dim3(256), 0, stream, (const float*)part,
                     groups * nsplit, Cout, Cin, dw);
  if (db) hipLaunchKernelGGL(conv_bias_reduce_kernel, dim3((Cout + 31) / 32), dim3(1024), 0, stream, (const float*)part_b, groups * nsplit, Cout, db);
  return cmr_launch_status();
}

namespace {
struct LwPlan { int ntp, ktp, nblk, kblk, slices; int64_t part_floats, partb_floats; };
inline LwPlan lw_plan(int64_t rows, int n, int k) {
  LwPlan p;
  const int nt = (n + 31) / 32, kt = (k + 31) / 32;
  p.ntp = nt <= 1 ? 1 : (nt == 2 ? 2 : 4);
  p.ktp = kt <= 1 ? 1 : (kt == 2 ? 2 : 4);
  p.nblk = (n + p.ntp * 32 - 1) / (p.ntp * 32);
  p.kblk = (k + p.ktp * 32 - 1) / (p.ktp * 32);
  int64_t s = rows / 128;                       // >= 128 rows (16 steps) per slice
  const int64_t cap = 1024 / ((int64_t)p.nblk * p.kblk) > 8 ? 1024 / ((int64_t)p.nblk * p.kblk) : 8;   // ~4 workgroups per CU in total
  if (s > cap) s = cap;
  if (s > 256) s = 256;
  if (s < 1) s = 1;
  p.slices = (int)s;
  const int64_t nsl = (int64_t)p.slices * (4 / p.ntp);
  p.part_floats = (int64_t)p.nblk * p.kblk * nsl * (p.ntp * 32) * (p.ktp * 32);
  p.partb_floats = (int64_t)p.nblk * nsl * (p.ntp * 32);
  return p;
}
}  // namespace

extern "C" int64_t cmr_linear_wgrad_workspace_bytes(int64_t rows, int n, int k) {
  const LwPlan p = lw_plan(rows, n, k);
  int64_t bytes = (p.part_floats + p.partb_floats) * (int64_t)sizeof(float);
  if (n % 32 == 0 && k % 32 == 0 && n <= 128 && k <= 128) {          // the LDS-staged variant: up to 768 whole-gradient partials
    const int64_t lds = (int64_t)768 * ((int64_t)n * k + n) * (int64_t)sizeof(float);
    if (lds > bytes) bytes = lds;
  }
  return bytes;
}

extern "C" int cmr_linear_wgrad_f32(const float* dy, int64_t lddy, int n, const float* x, int64_t ldx, int k, int64_t rows, float* dw,
                                    int64_t lddw, int accumulate, float* db, int accumulate_db, void* ws, int64_t ws_bytes,
                                    hipStream_t stream) {
  CMR_REQUIRE(dy && x && dw && ws && rows > 0 && n > 0 && k > 0 && (int64_t)n * k < 0x7fffffff);
  const LwPlan p = lw_plan(rows, n, k);
  CMR_REQUIRE(p.nblk <= 65535 && p.kblk <= 65535);
  CMR_REQUIRE(ws_bytes >= (p.part_floats + p.partb_floats) * (int64_t)sizeof(float));
  float* part = (float*)ws;
  {
    // LDS-staged variant: whole-row float4 staging, the full [n x k] gradient per workgroup (see linear_wgrad_lds_kernel)
    const int nt = n / 32, kt = k / 32;
    const bool shape_ok = n % 32 == 0 && k % 32 == 0 && ((nt == 2 && kt == 2) || (nt == 4 && kt == 2) || (nt == 2 && kt == 4) || (nt == 4 && kt == 4) ||
                                                         (nt == 1 && kt == 2) || (nt == 2 && kt == 1) || (nt == 1 && kt == 4) || (nt == 4 && kt == 1));
    if (g_lwgrad_lds && shape_ok && rows >= g_lwgrad_min_rows && lddy % 4 == 0 && ldx % 4 == 0 && cmr_aligned16(dy) && cmr_aligned16(x)) {
      const int64_t nblocks = (rows + 31) / 32;
      const size_t smem_wg = (size_t)2 * 32 * (n + k + 64) * sizeof(float);
      int per_cu = (int)((size_t)160 * 1024 / smem_wg);  // 3 workgroups per CU at 64 x 64, 2 at 128 x 64, 1 at 128 x 128
      if (per_cu > 3) per_cu = 3;
      int groups = 256 * per_cu;
      if (groups > nblocks / 8) groups = (int)(nblocks / 8 > 0 ? nblocks / 8 : 1);    // >= 8 row blocks per workgroup: every partial is 4 n k bytes the reduction reads back
      const int64_t need = ((int64_t)groups * n * k + (int64_t)groups * n) * (int64_t)sizeof(float);
      if (ws_bytes >= need) {
        float* pb = db ? part + (int64_t)groups * n * k : nullptr;
        const size_t smem = (size_t)2 * 32 * (n + k + 64) * sizeof(float);
#define CMR_LWL(NT_, KT_)                                                                                                             \
  {                                                                                                                                   \
    static CmrSmemCache granted{};                                                                                                    \
    if (cmr_grant_smem(reinterpret_cast<const void*>(linear_wgrad_lds_kernel<NT_, KT_>), smem, granted) != CMR_OK) return CMR_ELAUNCH;   \
    hipLaunchKernelGGL((linear_wgrad_lds_kernel<NT_, KT_>), dim3(groups), dim3(256), smem, stream, dy, lddy, x, ldx, rows, part, pb);   \
  }
        if (nt == 2 && kt == 2) CMR_LWL(2, 2)
        else if (nt == 4 && kt == 2) CMR_LWL(4, 2)
        else if (nt == 2 && kt == 4) CMR_LWL(2, 4)
        else if (nt == 4 && kt == 4) CMR_LWL(4, 4)
        else if (nt == 1 && kt == 2) CMR_LWL(1, 2)
        else if (nt == 2 && kt == 1) CMR_LWL(2, 1)
        else if (nt == 1 && kt == 4) CMR_LWL(1, 4)
        else CMR_LWL(4, 1)
#undef CMR_LWL
        const int64_t outs = (int64_t)n * k + (db ? n : 0);
        hipLaunchKernelGGL(linear_wgrad_reduce_kernel, dim3((unsigned)((outs + RED_OUT - 1) / RED_OUT)), dim3(RED_OUT * LRED_GRP), 0, stream, (const float*)part,
                           (const float*)pb, groups, n, k, 1, n, k, dw, lddw, accumulate, db, accumulate_db);
        return cmr_launch_status();
      }
    }
  }
  float* part_b = db ? part + p.part_floats : nullptr;
  const dim3 grid(p.slices, p.nblk, p.kblk);
#define CMR_LW(NT, KT) hipLaunchKernelGGL((linear_wgrad_kernel<NT, KT>), grid, dim3(256), 0, stream, dy, lddy, n, x, ldx, k, rows, part, part_b)
  if (p.ntp == 1 && p.ktp == 1) CMR_LW(1, 1);
  else if (p.ntp == 1 && p.ktp == 2) CMR_LW(1, 2);
  else if (p.ntp == 1 && p.ktp == 4) CMR_LW(1, 4);
  else if (p.ntp == 2 && p.ktp == 1) CMR_LW(2, 1);
  else if (p.ntp == 2 && p.ktp == 2) CMR_LW(2, 2);
  else if (p.ntp == 2 && p.ktp == 4) CMR_LW(2, 4);
  else if (p.ntp == 4 && p.ktp == 1) CMR_LW(4, 1);
  else if (p.ntp == 4 && p.ktp == 2) CMR_LW(4, 2);
  else CMR_LW(4, 4);
#undef CMR_LW
  const int64_t outs = (int64_t)n * k + (db ? n : 0);
  hipLaunchKernelGGL(linear_wgrad_reduce_kernel, dim3((unsigned)((outs + RED_OUT - 1) / RED_OUT)), dim3(RED_OUT * LRED_GRP), 0, stream, (const float*)part,
                     (const float*)part_b, p.slices * (4 / p.ntp), p.ntp * 32, p.ktp * 32, p.nblk, n, k, dw, lddw, accumulate, db, accumulate_db);
  return cmr_launch_status();
}

extern "C" int cmr_pack_conv3x3_f32(const float* w, int Cout, int Cin, int transpose, float* w9, float* ufrag, void* bf16_frag, int bf16_nt,
                                    hipStream_t stream) {
  CMR_REQUIRE(w && w9 && Cout > 0 && Cin > 0);
  if (ufrag) CMR_REQUIRE(Cout % 32 == 0 && Cin % 32 == 0);
  if (bf16_frag) CMR_REQUIRE(Cout % 32 == 0 && Cin % 32 == 0 && (bf16_nt == 1 || bf16_nt == 2) && (transpose ? Cin : Cout) % (32 * bf16_nt) == 0);
  hipLaunchKernelGGL(pack_conv3x3_kernel, dim3((Cout * Cin + 255) / 256), dim3(256), 0, stream, w, Cout, Cin, transpose, w9, ufrag,
                     (wg_bf16*)bf16_frag, bf16_nt);
  return cmr_launch_status();
}

extern "C" int cmr_pack_conv3x3_slots_f32(const float* src, float* dst, void* dst_bf16, const int64_t* table, int nslots, int64_t max_pairs,
                                          hipStream_t stream) {
  CMR_REQUIRE(src && dst && table && nslots > 0 && nslots <= 65535 && max_pairs > 0 && cmr_aligned16(dst));
  hipLaunchKernelGGL(pack_conv3x3_slots_kernel, dim3((unsigned)((max_pairs + 255) / 256), (unsigned)nslots), dim3(256), 0, stream, src, dst,
                     (wg_bf16*)dst_bf16, table);
  return cmr_launch_status();
}

// ------------------------------------------------------------------------------------------------------------------
// Backward of y = act(x W^T + b) on a SMALL row map (the transformer / proxy layers of Train_Geo.py: 640 - 4 096 rows, widths <= 128) in
// ONE launch:   dYe = dY * act'(Y);   dW (+)= dYe^T X;   db (+)= column sums of dYe;   dX = dYe W (+ res)
// These layers used to cost four launch-sized kernels each (activation backward, weight-gradient partials, their reduction, the
// data-gradient GEMM: ~30 us for ~10 MFLOP); there are ~150 of them per geometric update.  Workgroups of 16 waves take one of two roles:
//   blocks [0, n k / 256): one 16 x 16 tile of dW (v_mfma_f32_16x16x4_f32: 16 - 64 workgroups instead of the 4 - 16 of 32 x 32 tiles -- the
//                        role is a latency chain, not arithmetic).  Wave w multiplies the row quads w, w + 16, ... (48 operand loads in
//                        flight per lane), the 16 partial tiles are summed through LDS in a fixed order (double), db rides with k block 0.
//   the other blocks:    4 row tiles of dX each (one per SIMD); W (<= 64 KB) is staged in LDS by all 16 waves, a lane owns one row (the transposed
//                        form of linear_ws_kernel: operands W[n][.] from LDS, dYe[row][n .. n + 3] from registers).
// Deterministic: no atomics, fixed summation orders.
// ------------------------------------------------------------------------------------------------------------------
namespace {
struct LbrArgs {
  const float *dy, *y, *x, *w, *res;
  float *dw, *db, *dx;
  int64_t lddy, ldy, ldx, ldw, lddw, lddx, ldres;
  int rows, n, k, acc_dw, acc_db;
  float slope;
};
constexpr int LBR_U = 16, LBR_MAX_ROWS = 4096;

template <int KT>
__global__ __launch_bounds__(1024) void linear_bwd_rows_kernel(const LbrArgs a) {
  __shared__ __attribute__((aligned(16))) float lbr_smem[16 * 32 * 33 + 16 * 64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  constexpr int K16 = 2 * KT;                            // 16-column blocks of k
  const int ntw = (a.n / 16) * (K16);                    // 16 x 16 tiles of dW
  if ((int)blockIdx.x < ntw) {
    // ---------------- one 16 x 16 tile of dW (+ db): v_mfma_f32_16x16x4_f32, four rows per instruction (lane >> 4 = row of the quad)
    float* red = lbr_smem;                               // [16 waves][16][17]
    float* redb = lbr_smem + 16 * 16 * 17;               // [16][64]
    const int tn = blockIdx.x / K16, tk = blockIdx.x % K16;
    const int l15 = lane & 15, rq = lane >> 4;
    const int nquads = (a.rows + 3) / 4;
    const float* dyp = a.dy + tn * 16 + l15;
    const float* yp = a.y ? a.y + tn * 16 + l15 : nullptr;
    const float* xp = a.x + tk * 16 + l15;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    for (int q0 = wave; q0 < nquads; q0 += 16 * LBR_U) {
      float av[LBR_U], bv[LBR_U], yv[LBR_U];
#pragma unroll
      for (int u = 0; u < LBR_U; ++u) {
        const int r = 4 * (q0 + 16 * u) + rq;
        const int64_t rc = r < a.rows ? r : 0;           // (branch-free: rows past the end re-read row 0 and are zeroed below)
        av[u] = dyp[rc * a.lddy];
        bv[u] = xp[rc * a.ldx];
        yv[u] = yp ? yp[rc * a.ldy] : 1.f;
      }
#pragma unroll
      for (int u = 0; u < LBR_U; ++u) {
        const int r = 4 * (q0 + 16 * u) + rq;
        float d = yv[u] > 0.f ? av[u] : av[u] * a.slope;
        d = r < a.rows ? d : 0.f;
        bsum += d;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(d, bv[u], acc, 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[(wave * 16 + 4 * rq + r) * 17 + l15] = acc[r];      // D[4 (lane >> 4) + r][lane & 15]
    redb[wave * 64 + lane] = bsum;
    __syncthreads();
    if (tid < 256) {
      const int row = tid >> 4, col = tid & 15;
      double s = 0.0;
#pragma unroll
      for (int w = 0; w < 16; ++w) s += (double)red[(w * 16 + row) * 17 + col];
      float* d = a.dw + (int64_t)(tn * 16 + row) * a.lddw + tk * 16 + col;
      *d = a.acc_dw ? *d + (float)s : (float)s;
    }
    if (a.db && tk == 0 && tid < 16) {
      double s = 0.0;
#pragma unroll
      for (int w = 0; w < 16; ++w)
#pragma unroll
        for (int g = 0; g < 4; ++g) s += (double)redb[w * 64 + 16 * g + tid];
      float* d = a.db + tn * 16 + tid;
      *d = a.acc_db ? *d + (float)s : (float)s;
    }
    return;
  }
  // ---------------- 4 row tiles of dX
  if (a.dx == nullptr) return;
  constexpr int K = 32 * KT;
  // FOUR row tiles per workgroup (waves 0-3, one per SIMD; waves 4-15 only help staging W): with 16 tiles the workgroup's CU multiplied all of
  // them through its four matrix pipes -- 7.5 us for 640 rows on 2 of 256 CUs
  const int tile = ((int)blockIdx.x - ntw) * 4 + (wave < 4 ? wave : 0);
  const int row = tile * 32 + l31;
  const bool valid = row < a.rows;
  const int64_t rc = valid ? row : 0;
  const float* dyr = a.dy + rc * a.lddy + 4 * h;
  const float* yr = a.y ? a.y + rc * a.ldy + 4 * h : nullptr;
  const int ng = a.n / 8;                                // k-groups of the contraction over n (a multiple of 4)
  // the first k-groups of this lane's row are requested BEFORE W is staged: one memory round trip for both
  constexpr int TRIP = KT == 4 ? 4 : 8;                  // (64 accumulator registers at k = 128: four groups per trip)
  f32x4 dv[TRIP], yv[TRIP];
  auto load_trip = [&](int g0) {
#pragma unroll
    for (int i = 0; i < TRIP; ++i) {
      const int g = g0 + i < ng ? g0 + i : 0;            // (uniform) groups past the end re-read group 0 and are skipped below
      dv[i] = *reinterpret_cast<const f32x4*>(dyr + g * 8);
      yv[i] = yr ? *reinterpret_cast<const f32x4*>(yr + g * 8) : f32x4{1.f, 1.f, 1.f, 1.f};
    }
  };
  load_trip(0);
  float* Ws = lbr_smem;                                  // [n][K]  (n <= 128: <= 64 KB)
  for (int e = tid; e < a.n * (K / 4); e += 1024) {
    const int nn = e / (K / 4), c = (e % (K / 4)) * 4;
    *reinterpret_cast<f32x4*>(&Ws[nn * K + c]) = *reinterpret_cast<const f32x4*>(a.w + (int64_t)nn * a.ldw + c);
  }
  __syncthreads();
  if (wave >= 4 || tile * 32 >= a.rows) return;          // (wave-uniform; no barrier below)
  f32x16 acc[KT];
#pragma unroll
  for (int t = 0; t < KT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  for (int g0 = 0; g0 < ng; g0 += TRIP) {
    if (g0 > 0) load_trip(g0);
#pragma unroll
    for (int i = 0; i < TRIP; ++i) {
      if (g0 + i >= ng) break;                           // (uniform)
#pragma unroll
      for (int e = 0; e < 4; ++e) dv[i][e] = yv[i][e] > 0.f ? dv[i][e] : dv[i][e] * a.slope;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float* wr = Ws + ((g0 + i) * 8 + 4 * h + j) * K + l31;
#pragma unroll
        for (int t = 0; t < KT; ++t) acc[t] = cmr_mfma32(wr[32 * t], dv[i][j], acc[t]);
      }
    }
  }
  if (valid) {
    float* dxr = a.dx + (int64_t)row * a.lddx + 4 * h;
    const float* rr = a.res ? a.res + (int64_t)row * a.ldres + 4 * h : nullptr;
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v = {acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]};
        if (rr) v += *reinterpret_cast<const f32x4*>(rr + 32 * t + 8 * q);
        *reinterpret_cast<f32x4*>(dxr + 32 * t + 8 * q) = v;
      }
  }
}
}  // namespace

extern "C" int cmr_linear_bwd_rows_f32(const float* dy, int64_t lddy, const float* y, int64_t ldy, float slope, const float* x, int64_t ldx,
                                       const float* w, int64_t ldw, int64_t rows, int n, int k, float* dw, int64_t lddw, int accumulate_dw,
                                       float* db, int accumulate_db, const float* res, int64_t ldres, float* dx, int64_t lddx,
                                       hipStream_t stream) {
  CMR_REQUIRE(dy && x && w && dw && rows > 0 && n > 0 && k > 0);
  if (rows > LBR_MAX_ROWS || n % 32 != 0 || n > 128 || !(k == 32 || k == 64 || k == 128)) return CMR_EUNSUPPORTED;
  CMR_REQUIRE(lddy % 4 == 0 && cmr_aligned16(dy) && (!y || (ldy % 4 == 0 && cmr_aligned16(y))) && ldw % 4 == 0 && cmr_aligned16(w));
  CMR_REQUIRE(!dx || (lddx % 4 == 0 && cmr_aligned16(dx) && (!res || (ldres % 4 == 0 && cmr_aligned16(res)))));
  const LbrArgs a{dy, y, x, w, res, dw, db, dx, lddy, ldy, ldx, ldw, lddw, lddx, ldres, (int)rows, n, k, accumulate_dw, accumulate_db, slope};
  const int kt = k / 32, tiles = (int)((rows + 31) / 32);
  const dim3 grid((unsigned)((n / 16) * (k / 16) + (dx ? (tiles + 3) / 4 : 0)));
  if (kt == 1) hipLaunchKernelGGL(linear_bwd_rows_kernel<1>, grid, dim3(1024), 0, stream, a);
  else if (kt == 2) hipLaunchKernelGGL(linear_bwd_rows_kernel<2>, grid, dim3(1024), 0, stream, a);
  else hipLaunchKernelGGL(linear_bwd_rows_kernel<4>, grid, dim3(1024), 0, stream, a);
  return cmr_launch_status();
}
