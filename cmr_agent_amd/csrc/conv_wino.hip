// Stride-1 3x3 convolution by Winograd F(2x2, 3x3) on fp32 MFMA, fully fused (input transform, the 16
// position GEMMs, output transform, bias / residual / LeakyReLU / table / optional 2x2 average pool).
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A        16 multiplies per 2x2 outputs instead of 36
//
// The direct implicit-GEMM kernel (conv.hip) runs at ~75 % of the fp32 MFMA peak, so the remaining lever
// is the multiply count itself.  U = G g G^T is precomputed per layer ([16][Cout][Cin], host side).
//
// Workgroup = 256 threads = 4 waves; output tile 8 rows x 16 cols = 4 x 8 Winograd tiles = ONE 32-wide
// MFMA dimension (lane = Winograd tile); 64 output channels per workgroup (2 x 32).  Per 32-channel chunk of the input:
//   1. the 10 x 18 pixel halo chunk sits in LDS, double buffered, as 128-byte pixel slots whose eight 16-byte
//      pieces are XOR-swizzled with the slot number (no padding).  Chunk 0 goes through registers (zero padding applied on the
//      way); chunks 1.. are written by global_load_lds (LDS-DMA, no registers)
//      while the previous chunk is being multiplied; ONE barrier per chunk;
//   2. wave w multiplies the 4 positions (w, 0..3).  It never materialises V = B^T d B: per k-group a lane
//      reads the two input rows its position row needs (8 x ds_read_b128 of ITS tile's patch), forms the 4
//      transformed fragments with 8 float4 adds, and feeds 32 MFMAs.  A operand = U fragments straight
//      from global / L2 (stored host-side in fragment order), prefetched one k-group ahead.
// Epilogue: T[w][b] = sum_j M[w][j] A[j][b] in registers, exchanged through LDS, then wave q finishes
// Y[a][b] = sum_i A^T[a][i] T[i][b] for register quad q (4 consecutive channels -> float4 stores).
#include "cmr_common.h"


namespace {

struct WinoArgs {
  const float* x; int B, H, W, Cin;
  const float* u;      // G g G^T (BN folded) as A fragments [16][Cout/32][Cin/8][64][4]
  const float* bias; const float* res; const float* post;
  float* y; int Cout; float slope; int pool;
  // training (cmr_conv3x3_wino_stats_nhwc_f32; wave-specialised kernel, Cout = 64): per-helper-wave sums of the RAW convolution output
  // (before the bias) and of its square over the wave's pixels, [workgroup][4 helper waves][2][64] -- the BatchNorm statistics of the
  // layer without a pass over its output (pivot = the bias)
  float* stats;
  // ... or (bn_x non-null: this launch is a DATA GRADIENT whose output dz is the gradient at lrelu_{bn_slope}(BatchNorm(bn_x))) the two sums
  // of that BatchNorm's backward reduction, sum d and sum d xhat with d = dz * act'(.), xhat = (bn_x - mean) rstd (cmr_bn_bwd_f32's first
  // pass): bn_x [B][H][W][64] is read like a residual (and NOT added), bn_stat = the layer's stat [4][64]
  const float* bn_x; const float* bn_stat; float bn_slope;
};

constexpr int WT_TH = 8, WT_TW = 16;          // output tile
constexpr int WT_HR = 10, WT_HC = 18;         // halo
constexpr int WT_KC = 32;                     // channels per chunk = one 128-byte pixel slot
constexpr int WT_C4 = WT_KC / 4;
constexpr int WT_HALO_F4 = WT_HR * WT_HC * WT_C4;                 // 1440
constexpr int WT_HL = (WT_HALO_F4 + 255) / 256;                   // 6
// LDS image of a halo chunk: 10 rows x 20 slots (row pitch 20, columns de-interleaved by parity: pixel x sits in
// slot (x&1)*10 + x/2, slots 9 and 19 of a row are unused), 32 floats per slot, piece c of slot s stored at piece
// c ^ (s & 7).  The Winograd tiles of a wave start at even columns, so lane tx reads slot tx + const: with the
// swizzle the 8 lanes of a row hit 8 different 16-byte pieces and the ds_read_b128 are conflict free without
// padding -- which is what lets global_load_lds (64 lanes x 16 contiguous bytes) write the image directly.
constexpr int WT_PITCH = 20;
constexpr int WT_SLOTS = WT_HR * WT_PITCH;                        // 200
constexpr int WT_HALO_FLOATS = WT_SLOTS * WT_KC;                  // 6400
constexpr int WT_DMA = WT_SLOTS / 8;                              // 25 wave-wide LDS-DMA blocks of 8 slots (1 KB)
constexpr int WT_DMA_PER_WAVE = (WT_DMA + 3) / 4;                 // 7
// LDS: two halo buffers (12800 floats) in the K loop, T (8192 floats per 32-cout tile) after it
template <int NT> constexpr int wt_smem_floats() { return 8192 * NT > 2 * WT_HALO_FLOATS ? 8192 * NT : 2 * WT_HALO_FLOATS; }
__device__ __attribute__((aligned(16))) float wt_zero16[4] = {0.f, 0.f, 0.f, 0.f};   // NOT const (hipcc would fold the loads into branches)
// Re-materialises a thread-index-derived value at its point of use: the persistent tile loop would otherwise
// hoist every per-lane address / predicate of the prologue and epilogue out of the loop and keep ~100 of them
// live (spilled) across the MFMA loop.
__device__ __forceinline__ int wt_fresh(int v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ int wt_slot(int py, int px) { return py * WT_PITCH + (px & 1) * (WT_PITCH / 2) + (px >> 1); }
// piece c of slot s lives at piece c ^ key(s); bit 0 of the key also flips with the row pair, because the two tile rows a
// 16-lane ds_read_b128 pass covers are 2 rows = 40 slots = 0 (mod 8) apart and would otherwise share banks
__device__ __forceinline__ int wt_key(int slot) { return (slot & 7) ^ (((slot / WT_PITCH) >> 1) & 1); }
__device__ __forceinline__ int wt_lds(int slot, int c) { return slot * WT_KC + 4 * (c ^ wt_key(slot)); }   // float offset of piece c

// NT = 32-cout tiles per workgroup.  NT = 2: 248 VGPRs, 64 KB LDS, 2 workgroups / CU.  NT = 1: half the accumulators
// (<= 168 VGPRs, 51 KB), 3 workgroups / CU and twice the workgroups -- for maps too small to fill the chip.
template <int NT>
__global__ __launch_bounds__(256, NT == 1 ? 3 : 2) void conv3x3_wino_kernel(const WinoArgs a, const int tiles_x, const int tiles_y) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // uniform: lets the U bases live in SGPRs
  const int h = lane >> 5, l31 = lane & 31;
  const int nco = a.Cout / (32 * NT);
  const int nchunk = a.Cin / WT_KC;

  struct Tile { int b, co0, oy0, ox0; };
  // Workgroup -> tile.  Workgroups are dealt to the 8 XCDs round-robin (id % 8) and every XCD has its own L2, so XCD x
  // takes a CONTIGUOUS band of the tile order (ids x, x + 8, x + 16, ... -> consecutive tiles): horizontally adjacent
  // tiles (shared halo columns), the next tile row (shared halo rows) and the cout groups of one input tile (fastest
  // index: they read the same halo) are then resident in the same L2 at about the same time.
  auto decode = [&](int id) {
    const int nblk = gridDim.x, per = nblk >> 3, rem = nblk & 7, xcd = id & 7;
    int t = xcd * per + (xcd < rem ? xcd : rem) + (id >> 3);
    Tile r;
    r.co0 = (t % nco) * 32 * NT; t /= nco;
    r.ox0 = (t % tiles_x) * WT_TW; t /= tiles_x;
    r.oy0 = (t % tiles_y) * WT_TH;
    r.b = t / tiles_y;
    return r;
  };

  // chunk 0 of a tile: global -> registers (batched, clamped addresses) -> LDS (zero padding applied there)
  f32x4 hv[WT_HL];
  auto load_halo = [&](const Tile& t) {
    const float* xc = a.x + (int64_t)t.b * a.H * a.W * a.Cin;
    const int tf = wt_fresh(tid);
#pragma unroll
    for (int i = 0; i < WT_HL; ++i) {
      int e = tf + 256 * i;
      if (e >= WT_HALO_F4) e = WT_HALO_F4 - 1;
      const int p = e / WT_C4, c = e % WT_C4;
      int iy = t.oy0 - 1 + p / WT_HC, ix = t.ox0 - 1 + p % WT_HC;
      iy = iy < 0 ? 0 : (iy >= a.H ? a.H - 1 : iy);
      ix = ix < 0 ? 0 : (ix >= a.W ? a.W - 1 : ix);
      hv[i] = *reinterpret_cast<const f32x4*>(xc + (unsigned)((iy * a.W + ix) * a.Cin + c * 4));
    }
  };
  auto store_halo = [&](const Tile& t, float* halo) {
    const int tf = wt_fresh(tid);
#pragma unroll
    for (int i = 0; i < WT_HL; ++i) {
      const int e = tf + 256 * i;
      if (e < WT_HALO_F4) {
        const int p = e / WT_C4, c = e % WT_C4;
        const int iy = t.oy0 - 1 + p / WT_HC, ix = t.ox0 - 1 + p % WT_HC;
        const bool inb = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        *reinterpret_cast<f32x4*>(&halo[wt_lds(wt_slot(p / WT_HC, p % WT_HC), c)]) = inb ? hv[i] : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };
  // chunks 1..: LDS-DMA.  Block k = slots 8k .. 8k+7; lane = (slot 8k + lane/8, stored piece lane%8) fetches the
  // channel piece (lane%8) ^ (lane/8) of its pixel (zero page for padding / unused slots).  Wave w issues blocks
  // w, w+4, ... (clamped to the last block: a duplicate, identical write instead of a branch).
  auto dma_halo = [&](const Tile& t, int chunk, float* halo) {
    const int lf = wt_fresh(lane);
    const float* xc = a.x + (int64_t)t.b * a.H * a.W * a.Cin + chunk * WT_KC;
#pragma unroll
    for (int i = 0; i < WT_DMA_PER_WAVE; ++i) {
      int k = wave + 4 * i;
      k = k < WT_DMA ? k : WT_DMA - 1;
      const int slot = 8 * k + (lf >> 3);
      const int py = slot / WT_PITCH, r = slot % WT_PITCH;
      const int px = r < WT_PITCH / 2 ? 2 * r : 2 * r - (WT_PITCH - 1);
      const int iy = t.oy0 - 1 + py, ix = t.ox0 - 1 + px;
      const bool inb = px < WT_HC && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      const float* src = inb ? xc + (unsigned)((iy * a.W + ix) * a.Cin + ((lf & 7) ^ wt_key(slot)) * 4) : wt_zero16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(halo + k * 256), 16, 0, 0);
    }
  };

  // This wave multiplies the 4 positions (i = wave, j = 0..3).  Row i of B^T d needs two input rows:
  //   i=0: d0 - d2   i=1: d1 + d2   i=2: d2 - d1   i=3: d1 - d3       (ra, rb, sign below)
  const int ra = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
  const int rb = wave == 3 ? 3 : (wave == 2 ? 1 : 2);
  const float sgn = wave == 1 ? 1.f : -1.f;
  const int ty = l31 >> 3, tx = l31 & 7;
  // LDS float offsets of this lane's patch pieces (rows ra / rb, columns 2*tx + c) for k-group 0; k-group kg is
  // the same offset ^ 8*kg (pieces 2kg + h, swizzled)
  int pa[4], pb[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    pa[c] = wt_lds((2 * ty + ra) * WT_PITCH + (c & 1) * (WT_PITCH / 2) + tx + (c >> 1), h);
    pb[c] = wt_lds((2 * ty + rb) * WT_PITCH + (c & 1) * (WT_PITCH / 2) + tx + (c >> 1), h);
  }
  // U is stored as ready-made A fragments [pos][Cout/32][Cin/8][64 lanes][4]: a wave load is 1 KB contiguous
  const int64_t utile = (int64_t)(a.Cin / 8) * 256; // stride between cout tiles
  const int64_t upos = (int64_t)a.Cout * a.Cin;     // stride between positions
  const float* uw = a.u + (int64_t)(4 * wave) * upos + (unsigned)(lane * 4);   // position (wave, 0), this lane's slot
  const int q = wave;                               // epilogue: this wave finishes register quad q
  const int Ho = a.H, Wo = a.W;

  const Tile cur = decode(blockIdx.x);
  f32x4 wc[4][NT], wn[4][NT];
  load_halo(cur);
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int n = 0; n < NT; ++n) wc[j][n] = *reinterpret_cast<const f32x4*>(uw + (cur.co0 / 32 + n) * utile + j * upos);

  {
    f32x16 acc[4][NT];                               // [position column j][cout tile]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][n][r] = 0.f;
    const float* ub = uw + (cur.co0 / 32) * utile;

    // Loads retire in order (vmcnt), so a U fragment queued behind a halo load waits for HBM.  Order of issue per
    // chunk: U(kg1) | U(kg2) | U(kg3), halo(next chunk) | U(next chunk, kg0): every wait on U only has older U
    // loads ahead of it, and the halo has two k-groups of MFMAs (plus the barrier) to arrive.
    store_halo(cur, smem);
    int buf = 0;
    for (int chunk = 0; chunk < nchunk; ++chunk) {
      float* halo = smem + buf * WT_HALO_FLOATS;
      __syncthreads();                              // halo[buf] complete and visible; everybody is past the GEMMs that read halo[buf ^ 1]
      const float* uc = ub + chunk * (WT_KC / 8) * 256;
      const float* un = ub + (chunk + 1 < nchunk ? chunk + 1 : 0) * (WT_KC / 8) * 256;   // last chunk: a harmless in-bounds re-read
#pragma unroll
      for (int kg = 0; kg < WT_KC / 8; ++kg) {
        const float* up = kg + 1 < WT_KC / 8 ? uc + (kg + 1) * 256 : un;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int n = 0; n < NT; ++n) wn[j][n] = *reinterpret_cast<const f32x4*>(up + j * upos + n * utile);
        if (kg == WT_KC / 8 - 2 && chunk + 1 < nchunk) dma_halo(cur, chunk + 1, smem + (buf ^ 1) * WT_HALO_FLOATS);
        // on-the-fly input transform of this lane's tile: t[c] = d[ra][c] +- d[rb][c], then the 4 columns j
        f32x4 tc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const f32x4 da = *reinterpret_cast<const f32x4*>(&halo[wt_fresh(pa[c]) ^ (kg * 8)]);
          const f32x4 db = *reinterpret_cast<const f32x4*>(&halo[wt_fresh(pb[c]) ^ (kg * 8)]);
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
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[j][n] = cmr_mfma32(wc[j][n][e], vf[j][e], acc[j][n]);
          }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int n = 0; n < NT; ++n) wc[j][n] = wn[j][n];
      }
      buf ^= 1;
    }
    __syncthreads();                                // all waves done with the halo buffers before they are reused for T

    // ---- output transform, stage 1 (registers): T[w][b] = sum_j M[w][j] A[j][b]
    float* Ts = smem;                               // [(w*2 + b)*2 + n][register quad][lane][4]: b128 both ways
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        f32x4 t0, t1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * rq + e;
          t0[e] = (acc[0][n][r] + acc[1][n][r]) + acc[2][n][r];
          t1[e] = (acc[1][n][r] - acc[2][n][r]) - acc[3][n][r];
        }
        *reinterpret_cast<f32x4*>(&Ts[((((wave * 2 + 0) * NT + n) * 4 + rq) * 64 + lane) * 4]) = t0;
        *reinterpret_cast<f32x4*>(&Ts[((((wave * 2 + 1) * NT + n) * 4 + rq) * 64 + lane) * 4]) = t1;
      }

    __syncthreads();

    // ---- stage 2: wave q finishes the 4 outputs (a, b) of register quad q of both cout tiles.  Bias / residual:
    // one branch-free batch (absent operands read the zero page), in flight during the T reads.
    const int lf = wt_fresh(lane);
    const int tyf = (lf & 31) >> 3, txf = lf & 7;
    const int cq = cur.co0 + 8 * q + 4 * (lf >> 5); // + 32 n
    f32x4 bs[NT], rs[NT][2][2];
    {
      const float* bp = a.bias ? a.bias + cq : wt_zero16;
      const int bst = a.bias ? 32 : 0;
#pragma unroll
      for (int n = 0; n < NT; ++n) bs[n] = *reinterpret_cast<const f32x4*>(bp + bst * n);
      const float* rp = a.res ? a.res + (int64_t)cur.b * Ho * Wo * a.Cout + cq : wt_zero16;
      const int rst = a.res ? a.Cout : 0, rn = a.res ? 32 : 0;
      if (a.pool == 1) {
#pragma unroll
        for (int aa = 0; aa < 2; ++aa)
#pragma unroll
          for (int bb = 0; bb < 2; ++bb) {
            int oy = cur.oy0 + 2 * tyf + aa, ox = cur.ox0 + 2 * txf + bb;
            oy = oy < Ho ? oy : Ho - 1;
            ox = ox < Wo ? ox : Wo - 1;
            const int pix = oy * Wo + ox;
#pragma unroll
            for (int n = 0; n < NT; ++n) rs[n][aa][bb] = *reinterpret_cast<const f32x4*>(rp + (int64_t)pix * rst + rn * n);
          }
      }
    }
    f32x4 yv[NT][2][2];                              // [n][a][b]
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int bb = 0; bb < 2; ++bb) {
        f32x4 tw[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) tw[w] = *reinterpret_cast<const f32x4*>(&Ts[((((w * 2 + bb) * NT + n) * 4 + q) * 64 + lane) * 4]);
        yv[n][0][bb] = (tw[0] + tw[1]) + tw[2];
        yv[n][1][bb] = (tw[1] - tw[2]) - tw[3];
        __builtin_amdgcn_sched_barrier(0);          // keeps the 64 T reads from being hoisted into one 64-register burst
      }
    if (a.pool == 2) {
      const int py = (cur.oy0 >> 1) + tyf, px = (cur.ox0 >> 1) + txf;
      const int hp2 = Ho >> 1, wp2 = Wo >> 1;
      f32x4 sv[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
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
        sv[n] = s;
      }
#pragma unroll
      for (int n = 0; n < NT; ++n) cmr_pin(sv[n]);
      if (py < hp2 && px < wp2) {
        float* yp = a.y + (((int64_t)cur.b * hp2 + py) * wp2 + px) * a.Cout + cq;
#pragma unroll
        for (int n = 0; n < NT; ++n) *reinterpret_cast<f32x4*>(yp + 32 * n) = sv[n];
      }
    } else {
#pragma unroll
      for (int aa = 0; aa < 2; ++aa)
#pragma unroll
        for (int bb = 0; bb < 2; ++bb)
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            f32x4 v = yv[n][aa][bb] + bs[n];
            v += rs[n][aa][bb];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * a.slope;
            yv[n][aa][bb] = v;
          }
      if (a.post) {                                 // positional table (one conv per forward): fetched late, registers are short
#pragma unroll
        for (int aa = 0; aa < 2; ++aa)
#pragma unroll
          for (int bb = 0; bb < 2; ++bb) {
            int oy = cur.oy0 + 2 * tyf + aa, ox = cur.ox0 + 2 * txf + bb;
            oy = oy < Ho ? oy : Ho - 1;
            ox = ox < Wo ? ox : Wo - 1;
            const float* pp = a.post + (int64_t)(oy * Wo + ox) * a.Cout + cq;
#pragma unroll
            for (int n = 0; n < NT; ++n) yv[n][aa][bb] += *reinterpret_cast<const f32x4*>(pp + 32 * n);
          }
      }
#pragma unroll
      for (int aa = 0; aa < 2; ++aa)
#pragma unroll
        for (int bb = 0; bb < 2; ++bb)
#pragma unroll
          for (int n = 0; n < NT; ++n) cmr_pin(yv[n][aa][bb]);
#pragma unroll
      for (int aa = 0; aa < 2; ++aa)
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
          const int oy = cur.oy0 + 2 * tyf + aa, ox = cur.ox0 + 2 * txf + bb;
          if (oy < Ho && ox < Wo) {
            float* yp = a.y + (((int64_t)cur.b * Ho + oy) * Wo + ox) * a.Cout + cq;
#pragma unroll
            for (int n = 0; n < NT; ++n) *reinterpret_cast<f32x4*>(yp + 32 * n) = yv[n][aa][bb];
          }
        }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Wave-specialised persistent variant (64 couts per tile, Cin >= 64, maps with many tiles per CU).
//
// In the kernel above a workgroup is serial -- halo + U prologue, K loop, output transform + stores -- and the matrix pipe
// is only busy in the middle part; two co-resident workgroups overlap by chance (measured duty 52 %).  Here ONE workgroup
// of 8 waves owns a CU and walks over its tiles: waves 0-3 do nothing but [ds_read + input transform + MFMA] and hand their
// accumulators over through LDS (stage 1 of the output transform); waves 4-7 feed them (LDS-DMA of the next halo chunk,
// issued one chunk ahead) and finish the PREVIOUS tile (stage 2, bias / residual / activation / table, global stores)
// while the MFMA waves are already multiplying the next one.  One s_barrier per 32-channel chunk, executed by all 8 waves,
// is the only synchronisation:
//   interval (tile k, chunk c):  MFMA waves: GEMM(k, c) from halo[g & 1]; after the last chunk: T(k) -> LDS
//                                helpers:    DMA of the next chunk in sequence -> halo[(g + 1) & 1];
//                                            c == 0: T(k - 1) -> registers (stage 2);  c == 1: epilogue math + stores of tile k - 1
// T(k) is written after the LAST chunk's GEMM and T(k - 1) is read in the FIRST chunk's interval, so with >= 2 chunks the
// two never meet.  Inside a chunk the MFMA waves prefetch the next k-group's patch rows and U fragments under the current
// k-group's MFMAs (U straight into the registers the just-issued MFMAs have consumed).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int WS_T_OFF = 2 * WT_HALO_FLOATS;
constexpr int WS_T_ROW = 68;                          // floats per (position pair, Winograd tile) row of T: 64 couts + 4 of padding (bank spread)
constexpr int WS_SMEM_FLOATS = WS_T_OFF + 8 * 32 * WS_T_ROW;

__device__ __forceinline__ void ws_barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// dbg: compile-time ablation mask used while tuning (see launch_wino_ws); the library instantiates 0 only, where every
// `dbg & ...` test folds away.
// MW = MFMA waves per SIMD.  1: waves 0-3 multiply (4 positions x 2 cout tiles = 128 accumulator registers each), waves 4-7 help:
// 8 waves x 256 registers.  2: waves 0-7 multiply -- wave m owns position row m & 3 and cout tile m >> 2 (64 accumulator registers,
// the same products in the same order: bit-identical results) -- and waves 8-11 help: 12 waves x 168 registers.  Two MFMA waves share
// a SIMD's matrix pipe, so one wave's wait for its U fragments is the other's issue slot; the price is that both waves of a SIMD read
// and transform the same patch rows (LDS reads and VALU x 2) and that 168 registers make the helpers spill.  MEASURED (round 5,
// tools/wino_mw_ab.py, profiles/r05_wino_mw_ab.txt): bit-identical and 10 - 25 % slower at every shape of the step (352x1216 64->64:
// 1 112 -> 1 432 us; 88x304 128->128: 257 -> 293 us) -- the single MFMA wave was not starved (it issues 80 % of the launch's cycles at
// the ~1.93 GHz the chip sustains under this load), so MW = 1 stays the library's choice; MW = 2 is kept for the A/B.
#ifndef CMR_WS_STATS
#define CMR_WS_STATS 1        // BatchNorm sums in the helpers' epilogue (0 / CMR_WS_BNBWD 0: compiled out, for same-box A/B builds)
#endif
#ifndef CMR_WS_BNBWD
#define CMR_WS_BNBWD 0        // the BatchNorm-BACKWARD sums in a data gradient's epilogue: built, parity-tested and measured (round 5, same box,
                              // tools/wino_ws_bench.py on -D builds): 16 more live registers in the helpers make EVERY launch of the kernel
                              // 0.5 - 1 % slower on the 64 -> 64 full-resolution maps (1 114 -> 1 123 us) while the update gains 0.06 ms of
                              // 76; compiled out (cmr_conv3x3_wino_bnbwd_nhwc_f32 answers -3), -DCMR_WS_BNBWD=1 brings it back
#endif
constexpr bool WS_STATS = CMR_WS_STATS != 0, WS_BNBWD = CMR_WS_STATS != 0 && CMR_WS_BNBWD != 0;
template <int dbg, int MW>
__global__ __launch_bounds__(256 * (MW + 1), 1) void conv3x3_wino_ws_kernel(const WinoArgs a, const int tiles_x, const int tiles_y, const int ntiles) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NT = 2 / MW;                         // cout tiles per MFMA wave
  constexpr int NMW = 4 * MW;                        // MFMA waves
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nco = a.Cout / 64;
  const int nchunk = a.Cin / WT_KC;
  const int nk = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;     // tiles of this workgroup (>= 1)

  struct Tile { int b, co0, oy0, ox0; };
  auto decode = [&](int id) {                                      // same XCD-banded order as above, over ntiles virtual workgroups
    id = id < ntiles ? id : ntiles - 1;
    const int per = ntiles >> 3, rem = ntiles & 7, xcd = id & 7;
    int t = xcd * per + (xcd < rem ? xcd : rem) + (id >> 3);
    Tile r;
    r.co0 = (t % nco) * 64; t /= nco;
    r.ox0 = (t % tiles_x) * WT_TW; t /= tiles_x;
    r.oy0 = (t % tiles_y) * WT_TH;
    r.b = t / tiles_y;
    return r;
  };

  if (wave < NMW) {
    // =========================================== MFMA waves ===========================================
    const int h = lane >> 5, l31 = lane & 31;
    const int wr = wave & 3, nsel = (wave >> 2) * NT;  // position row, first cout tile of this wave
    const int ra = wr == 0 ? 0 : (wr == 2 ? 2 : 1);
    const int rb = wr == 3 ? 3 : (wr == 2 ? 1 : 2);
    const float sgn = wr == 1 ? 1.f : -1.f;
    const int ty = l31 >> 3, tx = l31 & 7;
    int pa[4], pb[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      pa[c] = wt_lds((2 * ty + ra) * WT_PITCH + (c & 1) * (WT_PITCH / 2) + tx + (c >> 1), h);
      pb[c] = wt_lds((2 * ty + rb) * WT_PITCH + (c & 1) * (WT_PITCH / 2) + tx + (c >> 1), h);
    }
    // U fragments through a buffer resource: ONE per-lane VGPR offset (lane * 16 bytes), everything else is a scalar offset --
    // eight 64-bit per-lane pointers (16 VGPRs) would not fit next to 128 accumulators + the software pipeline
    const __amdgpu_buffer_rsrc_t ursrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.u, 0, 16 * a.Cout * a.Cin * 4, 0x00020000);
    const int lane_off = lane * 16;
    const int utile = (a.Cin / 8) * 1024;             // bytes between cout tiles
    const int upos = a.Cout * a.Cin * 4;              // bytes between positions
    const int uwave = 4 * wr * upos + nsel * utile;   // position (wr, 0), this wave's first cout tile
    auto load_u = [&](int soff) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ursrc, lane_off, soff, 0)); };
    int ub = uwave + (decode(blockIdx.x).co0 / 32) * utile;
    f32x4 wc[4][NT];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int n = 0; n < NT; ++n) wc[j][n] = load_u(ub + j * upos + n * utile);
    int g = 0;
    [[maybe_unused]] uint64_t tm_busy = 0, tm_wait = 0, tm_mark = (dbg & 64) ? __builtin_amdgcn_s_memtime() : 0;
    for (int k = 0; k <= nk; ++k) {
      const bool live = k < nk;
      const int ubn = uwave + (decode((int)blockIdx.x + (k + 1 < nk ? k + 1 : k) * (int)gridDim.x).co0 / 32) * utile;   // next tile's U (or this one's again)
      f32x16 acc[4][NT];                                // not zeroed: the first k-step of a tile multiplies onto the constant 0
      for (int chunk = 0; chunk < nchunk; ++chunk, ++g) {
        if (dbg & 64) { const uint64_t t = __builtin_amdgcn_s_memtime(); tm_busy += t - tm_mark; tm_mark = t; }
        ws_barrier_lds();                               // halo[g & 1] landed (helpers waited for their DMA before arriving)
        if (dbg & 64) { const uint64_t t = __builtin_amdgcn_s_memtime(); tm_wait += t - tm_mark; tm_mark = t; }
        if (!live) continue;
        const float* halo = smem + (g & 1) * WT_HALO_FLOATS;
        const int uc = ub + chunk * (WT_KC / 8) * 1024;
        const int un = chunk + 1 < nchunk ? ub + (chunk + 1) * (WT_KC / 8) * 1024 : ubn;
        f32x4 vf[4];
        {                                               // k-group 0 of the chunk: nothing to hide behind
          f32x4 tc[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const f32x4 da = *reinterpret_cast<const f32x4*>(&halo[pa[c]]);
            const f32x4 db = *reinterpret_cast<const f32x4*>(&halo[pb[c]]);
            tc[c] = da + sgn * db;
          }
          vf[0] = tc[0] - tc[2];
          vf[1] = tc[1] + tc[2];
          vf[2] = tc[2] - tc[1];
          vf[3] = tc[1] - tc[3];
        }
#pragma unroll
        for (int kg = 0; kg < WT_KC / 8; ++kg) {
          const int up = kg + 1 < WT_KC / 8 ? uc + (kg + 1) * 1024 : un;
          const bool more = kg + 1 < WT_KC / 8;         // compile-time after unrolling
          const int kx = (kg + 1) * 8;
          // next k-group's fragments, built up in place: vn[0] = t0 - t2, vn[1] = t1 + t2, vn[2] = t2 - t1, vn[3] = t1 - t3
          f32x4 da[2], db[2], vn[4], t1;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
              for (int n = 0; n < NT; ++n) {
                if (dbg & 32) continue;
                if (kg == 0 && e == 0 && chunk == 0) {      // (wave-uniform branch around one MFMA; 128 v_mov per tile otherwise)
                  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                  acc[j][n] = cmr_mfma32(wc[j][n][e], vf[j][e], zero);
                } else {
                  acc[j][n] = cmr_mfma32(wc[j][n][e], vf[j][e], acc[j][n]);
                }
              }
            // the 8 MFMAs above have read wc[j][*]: refill them with the next k-group's fragments (24 MFMAs = 1 536 cycles of lead)
            if (!(dbg & 8)) {
#pragma unroll
              for (int n = 0; n < NT; ++n) wc[j][n] = load_u(up + j * upos + n * utile);
            }
            if (more && !(dbg & 16)) {
              if (j == 0) {
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                  da[c] = *reinterpret_cast<const f32x4*>(&halo[pa[c] ^ kx]);
                  db[c] = *reinterpret_cast<const f32x4*>(&halo[pb[c] ^ kx]);
                }
              } else if (j == 1) {
                vn[0] = da[0] + sgn * db[0];
                t1 = da[1] + sgn * db[1];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                  da[c] = *reinterpret_cast<const f32x4*>(&halo[pa[c + 2] ^ kx]);
                  db[c] = *reinterpret_cast<const f32x4*>(&halo[pb[c + 2] ^ kx]);
                }
              } else if (j == 2) {
                const f32x4 t2 = da[0] + sgn * db[0];
                vn[0] -= t2;
                vn[1] = t1 + t2;
                vn[2] = t2 - t1;
                vn[3] = t1 - (da[1] + sgn * db[1]);
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          if (more) {
#pragma unroll
            for (int j = 0; j < 4; ++j) vf[j] = vn[j];
          }
        }
      }
      if (live && !(dbg & 4)) {
        // stage 1 of the output transform: T[w][b] = sum_j M[w][j] A[j][b] -> LDS, pixel-major: row (w, b, tile) holds the 64 couts, so
        // that the helpers can finish WHOLE pixels (16 lanes x 16 bytes = the 256 contiguous bytes of a pixel's cout group)
        float* Ts = smem + WS_T_OFF;
        const int h = lane >> 5, l31 = lane & 31;
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
          for (int rq = 0; rq < 4; ++rq) {
            f32x4 t0, t1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int r = 4 * rq + e;
              t0[e] = (acc[0][n][r] + acc[1][n][r]) + acc[2][n][r];
              t1[e] = (acc[1][n][r] - acc[2][n][r]) - acc[3][n][r];
            }
            *reinterpret_cast<f32x4*>(&Ts[((wr * 2 + 0) * 32 + l31) * WS_T_ROW + (nsel + n) * 32 + 8 * rq + 4 * h]) = t0;
            *reinterpret_cast<f32x4*>(&Ts[((wr * 2 + 1) * 32 + l31) * WS_T_ROW + (nsel + n) * 32 + 8 * rq + 4 * h]) = t1;
          }
      }
      ub = ubn;
    }
    if ((dbg & 64) && lane == 0) {                       // timing build only: a.post is the stamp buffer [workgroup][wave][2]
      float* o = const_cast<float*>(a.post) + ((int)blockIdx.x * (NMW + 4) + wave) * 2;
      o[0] = (float)tm_busy; o[1] = (float)tm_wait;
    }
  } else {
    // ============================================ helpers ============================================
    const int hw = wave - NMW;
    const int Ho = a.H, Wo = a.W;
    // The helpers are the younger wave of every SIMD's pair and would get the VALU issue slots the MFMA wave leaves over (their
    // epilogue then takes longer than a chunk of MFMAs and the MFMA waves wait at the barrier: 12-20 % of the launch by s_memtime
    // stamps, tools/wino_timing.py); the MFMA wave needs one issue slot per 64 cycles and does not notice.
    __builtin_amdgcn_s_setprio(1);
    // Everything per-lane is tile-invariant and computed once: byte offsets relative to a per-tile SCALAR base (the address
    // arithmetic of a tile is a handful of SALU instructions instead of ~600 VALU / readlane instructions per helper wave).
    // Halo DMA: block kb = hw + 4 i covers LDS slots 8 kb .. 8 kb + 7; lane >> 3 = slot in the block, lane & 7 = 16-byte piece.
    unsigned doff[WT_DMA_PER_WAVE];
    int dpy[WT_DMA_PER_WAVE], dpx[WT_DMA_PER_WAVE];
#pragma unroll
    for (int i = 0; i < WT_DMA_PER_WAVE; ++i) {
      int kb = hw + 4 * i;
      kb = kb < WT_DMA ? kb : WT_DMA - 1;
      const int slot = 8 * kb + (lane >> 3);
      const int py = slot / WT_PITCH, r = slot % WT_PITCH;
      const int px = r < WT_PITCH / 2 ? 2 * r : 2 * r - (WT_PITCH - 1);
      doff[i] = (unsigned)(((py * a.W + px) * a.Cin + ((lane & 7) ^ wt_key(slot)) * 4) * 4);
      dpy[i] = px < WT_HC ? py : 0x40000000;              // the unused slots of a row never pass the bounds test: zero page
      dpx[i] = px;
    }
    auto dma_halo = [&](const Tile& t, int chunk, float* halo) __attribute__((always_inline)) {
      const int y0 = t.oy0 - 1, x0 = t.ox0 - 1;
      const char* base = reinterpret_cast<const char*>(a.x + ((int64_t)t.b * a.H * a.W + (int64_t)y0 * a.W + x0) * a.Cin + chunk * WT_KC);
#pragma unroll
      for (int i = 0; i < WT_DMA_PER_WAVE; ++i) {
        if (hw + 4 * i >= WT_DMA) continue;                 // (wave-uniform) blocks 25..27 do not exist
        const bool inb = (unsigned)(dpy[i] + y0) < (unsigned)a.H && (unsigned)(dpx[i] + x0) < (unsigned)a.W;
        const char* src = inb ? base + doff[i] : reinterpret_cast<const char*>(wt_zero16);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(halo + (hw + 4 * i) * 256), 16, 0, 0);
      }
    };
    // epilogue items of this lane: channel quad cq4 (couts 4 cq4 .. 4 cq4 + 3 of the tile's 64) of Winograd tiles t0 and t0 + 16;
    // 16 consecutive lanes = one whole pixel's cout group (256 contiguous bytes in y / res / post)
    const int hidx = hw * 64 + lane;
    const int cq4 = hidx & 15, t0 = hidx >> 4;
    unsigned poff[2][2][2], qoff[2];                      // byte offsets of the lane's 8 pixels (pooled: 2) from the tile's first pixel
    int pdy[2], pdx[2];
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
      const int t = t0 + 16 * ii;
      pdy[ii] = 2 * (t >> 3);
      pdx[ii] = 2 * (t & 7);
      qoff[ii] = (unsigned)((((t >> 3) * (Wo >> 1) + (t & 7)) * a.Cout + 4 * cq4) * 4);
#pragma unroll
      for (int aa = 0; aa < 2; ++aa)
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) poff[ii][aa][bb] = (unsigned)((((pdy[ii] + aa) * Wo + pdx[ii] + bb) * a.Cout + 4 * cq4) * 4);
    }
    const unsigned coff = (unsigned)(16 * cq4);            // the tile's first pixel: where loads of pixels outside the map go
    Tile prev = decode(blockIdx.x), cur = prev;
    dma_halo(cur, 0, smem);                              // chunk (0, 0); completed by the wait in front of the first barrier
    f32x4 yv[2][2][2];                                   // [item][a][b]
    f32x4 st_s = {0.f, 0.f, 0.f, 0.f}, st_q = {0.f, 0.f, 0.f, 0.f};   // a.stats: sums of this lane's channel quad over its pixels
    f32x4 bmean, brstd, bmsc, bmsh;                      // a.bn_x: the BatchNorm of this lane's channel quad (Cout = 64: one cout group)
    if (WS_BNBWD && a.bn_x) {
      bmean = *reinterpret_cast<const f32x4*>(a.bn_stat + 4 * cq4);
      brstd = *reinterpret_cast<const f32x4*>(a.bn_stat + 64 + 4 * cq4);
      bmsc = *reinterpret_cast<const f32x4*>(a.bn_stat + 128 + 4 * cq4);
      bmsh = *reinterpret_cast<const f32x4*>(a.bn_stat + 192 + 4 * cq4);
    }
    int g = 0, pend = 0;
    [[maybe_unused]] uint64_t tm_busy = 0, tm_wait = 0, tm_vm = 0, tm_dma = 0, tm_epi = 0, tm_st = 0, tm_mark = (dbg & 64) ? __builtin_amdgcn_s_memtime() : 0;
    for (int k = 0; k <= nk; ++k) {
      const Tile nxt = decode((int)blockIdx.x + (k + 1) * (int)gridDim.x);
      for (int chunk = 0; chunk < nchunk; ++chunk, ++g) {
        // This wave's DMA of the previous interval must have landed before the barrier.  vmcnt counts loads, LDS-DMA and stores
        // together in issue order, and the stores of interval 1 are issued BEHIND that interval's DMA: on a tile that lies fully
        // inside the map their number is known (8, or 2 pooled), so the wait leaves exactly them in flight instead of tying
        // the barrier to the completion of 32 KB of stores; any other interval drains everything.
        if (dbg & 64) { const uint64_t t = __builtin_amdgcn_s_memtime(); tm_busy += t - tm_mark; tm_mark = t; }
        if (pend == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (pend == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        pend = 0;
        if (dbg & 64) { const uint64_t t = __builtin_amdgcn_s_memtime(); tm_vm += t - tm_mark; tm_mark = t; }
        ws_barrier_lds();
        if (dbg & 64) { const uint64_t t = __builtin_amdgcn_s_memtime(); tm_wait += t - tm_mark; tm_mark = t; }
        // feed: the next chunk in sequence goes into the buffer the MFMA waves have just left (first thing in the interval:
        // it has to land before the next barrier)
        if (chunk + 1 < nchunk) {
          if (k < nk) dma_halo(cur, chunk + 1, smem + ((g + 1) & 1) * WT_HALO_FLOATS);
        } else if (k + 1 < nk) {
          dma_halo(nxt, 0, smem + ((g + 1) & 1) * WT_HALO_FLOATS);
        }
        [[maybe_unused]] uint64_t tm_p = 0;
        if (dbg & 64) { __builtin_amdgcn_sched_barrier(0); tm_p = __builtin_amdgcn_s_memtime(); tm_dma += tm_p - tm_mark; }
        if (k == 0 || (dbg & 2)) continue;
        if (chunk == 0) {
          // interval 0 of the next tile: everything that computes.  Residual / bias loads first (in flight during the LDS
          // work), stage 2 from T, then bias + residual + LeakyReLU (+ table); the finished pixels stay in registers.
          const float* Ts = smem + WS_T_OFF;
          const f32x4 bsv = *reinterpret_cast<const f32x4*>(a.bias ? a.bias + prev.co0 + 4 * cq4 : wt_zero16);
          f32x4 rs[2][2][2];
          const int64_t pix0 = ((int64_t)prev.b * Ho + prev.oy0) * Wo + prev.ox0;          // the tile's first pixel (scalar)
          const int ly = Ho - prev.oy0, lx = Wo - prev.ox0;
          const bool full = ly >= WT_TH && lx >= WT_TW;
          if (a.pool == 1) {
            if (a.res || (WS_BNBWD && a.bn_x)) {                         // (bn_x: the BatchNorm input takes the residual's registers; it is not added below)
              const char* rb = reinterpret_cast<const char*>(((WS_BNBWD && a.bn_x) ? a.bn_x : a.res) + pix0 * a.Cout + prev.co0);
#pragma unroll
              for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int aa = 0; aa < 2; ++aa)
#pragma unroll
                  for (int bb = 0; bb < 2; ++bb) {
                    const unsigned o = full || (pdy[ii] + aa < ly && pdx[ii] + bb < lx) ? poff[ii][aa][bb] : coff;
                    rs[ii][aa][bb] = *reinterpret_cast<const f32x4*>(rb + o);
                  }
            } else {
#pragma unroll
              for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int aa = 0; aa < 2; ++aa)
#pragma unroll
                  for (int bb = 0; bb < 2; ++bb) rs[ii][aa][bb] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
          }
#pragma unroll
          for (int ii = 0; ii < 2; ++ii)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
              f32x4 tw[4];
#pragma unroll
              for (int w = 0; w < 4; ++w)
                tw[w] = *reinterpret_cast<const f32x4*>(&Ts[((w * 2 + bb) * 32 + t0 + 16 * ii) * WS_T_ROW + 4 * cq4]);
              yv[ii][0][bb] = (tw[0] + tw[1]) + tw[2];
              yv[ii][1][bb] = (tw[1] - tw[2]) - tw[3];
            }
          if (WS_STATS && a.stats) {                       // (uniform) the helpers wait 25 - 50 % of a launch at the barrier: these adds are free
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
              for (int aa = 0; aa < 2; ++aa)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) {
                  const bool ok = full || (pdy[ii] + aa < ly && pdx[ii] + bb < lx);
                  const f32x4 r = yv[ii][aa][bb];
                  if (WS_BNBWD && a.bn_x) {                // (uniform) BatchNorm-backward sums: the arithmetic of bn_bwd_partial_kernel
                    const f32x4 xv = rs[ii][aa][bb];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                      const float pre = __builtin_fmaf(xv[e], bmsc[e], bmsh[e]);
                      const float d = pre > 0.f ? r[e] : r[e] * a.bn_slope;
                      const float xh = (xv[e] - bmean[e]) * brstd[e];
                      st_s[e] += ok ? d : 0.f;
                      st_q[e] += ok ? d * xh : 0.f;
                    }
                  } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                      st_s[e] += ok ? r[e] : 0.f;
                      st_q[e] += ok ? r[e] * r[e] : 0.f;
                    }
                  }
                }
          }
          if (a.pool == 2) {
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
              f32x4 sacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
              for (int aa = 0; aa < 2; ++aa)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                  for (int e = 0; e < 4; ++e) {
                    const float v = yv[ii][aa][bb][e] + bsv[e];
                    sacc[e] += fmaxf(v, v * a.slope);                 // LeakyReLU for 0 <= slope <= 1 (the entry point routes other slopes elsewhere)
                  }
#pragma unroll
              for (int e = 0; e < 4; ++e) sacc[e] *= 0.25f;
              yv[ii][0][0] = sacc;
            }
          } else {
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
              for (int aa = 0; aa < 2; ++aa)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) {
                  f32x4 v = yv[ii][aa][bb] + bsv;
                  if (!(WS_BNBWD && a.bn_x)) v += rs[ii][aa][bb];
#pragma unroll
                  for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], v[e] * a.slope);
                  yv[ii][aa][bb] = v;
                }
            if (a.post && !(dbg & 64)) {
              const char* pb = reinterpret_cast<const char*>(a.post + ((int64_t)prev.oy0 * Wo + prev.ox0) * a.Cout + prev.co0);
#pragma unroll
              for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int aa = 0; aa < 2; ++aa)
#pragma unroll
                  for (int bb = 0; bb < 2; ++bb) {
                    const unsigned o = full || (pdy[ii] + aa < ly && pdx[ii] + bb < lx) ? poff[ii][aa][bb] : coff;
                    yv[ii][aa][bb] += *reinterpret_cast<const f32x4*>(pb + o);
                  }
            }
          }
          if (dbg & 64) {
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
              for (int aa = 0; aa < 2; ++aa)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) cmr_pin(yv[ii][aa][bb]);
            __builtin_amdgcn_sched_barrier(0);
            tm_epi += __builtin_amdgcn_s_memtime() - tm_p;
          }
        } else if (chunk == 1) {
          // interval 1: nothing but the stores, issued right behind the DMA
          if (dbg & 1) continue;
          __builtin_amdgcn_sched_barrier(0);
          const int ly = Ho - prev.oy0, lx = Wo - prev.ox0;
          const bool full = ly >= WT_TH && lx >= WT_TW;
          pend = full ? (a.pool == 2 ? 2 : 8) : 0;
          if (a.pool == 2) {
            const int hp2 = Ho >> 1, wp2 = Wo >> 1;
            char* yb = reinterpret_cast<char*>(a.y + (((int64_t)prev.b * hp2 + (prev.oy0 >> 1)) * wp2 + (prev.ox0 >> 1)) * a.Cout + prev.co0);
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
              if (full || ((prev.oy0 + pdy[ii]) >> 1 < hp2 && (prev.ox0 + pdx[ii]) >> 1 < wp2))
                __builtin_nontemporal_store(yv[ii][0][0], reinterpret_cast<f32x4*>(yb + qoff[ii]));
          } else {
            char* yb = reinterpret_cast<char*>(a.y + ((dbg & 128) ? (int64_t)(blockIdx.x & 7) * 16 * Wo : (((int64_t)prev.b * Ho + prev.oy0) * Wo + prev.ox0)) * a.Cout + prev.co0);
            if (full) {
#pragma unroll
              for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int aa = 0; aa < 2; ++aa)
#pragma unroll
                  for (int bb = 0; bb < 2; ++bb) __builtin_nontemporal_store(yv[ii][aa][bb], reinterpret_cast<f32x4*>(yb + poff[ii][aa][bb]));
            } else {
#pragma unroll
              for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int aa = 0; aa < 2; ++aa)
#pragma unroll
                  for (int bb = 0; bb < 2; ++bb)
                    if (pdy[ii] + aa < ly && pdx[ii] + bb < lx)
                      __builtin_nontemporal_store(yv[ii][aa][bb], reinterpret_cast<f32x4*>(yb + poff[ii][aa][bb]));
            }
          }
          if (dbg & 64) { __builtin_amdgcn_sched_barrier(0); tm_st += __builtin_amdgcn_s_memtime() - tm_p; }
        }
      }
      prev = cur;
      cur = nxt;
    }
    if ((dbg & 64) && lane == 0) {
      float* o = const_cast<float*>(a.post) + ((int)blockIdx.x * (NMW + 4) + wave) * 2;
      o[0] = (float)tm_busy; o[1] = (float)tm_wait;
      float* q = const_cast<float*>(a.post) + 2 * (NMW + 4) * gridDim.x + ((int)blockIdx.x * 4 + hw) * 4;
      q[0] = (float)tm_vm; q[1] = (float)tm_dma; q[2] = (float)tm_epi; q[3] = (float)tm_st;
    }
    if (WS_STATS && a.stats) {
      // lanes l, l + 16, l + 32, l + 48 of a helper wave hold the same channel quad (cq4 = lane & 15): fixed-order sum, lane < 16 writes
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        st_s[e] += cmr_xor16(st_s[e]);
        st_s[e] += cmr_xhalf(st_s[e]);
        st_q[e] += cmr_xor16(st_q[e]);
        st_q[e] += cmr_xhalf(st_q[e]);
      }
      if (lane < 16) {
        float* o = a.stats + ((int64_t)blockIdx.x * 4 + hw) * 128 + 4 * lane;
        *reinterpret_cast<f32x4*>(o) = st_s;
        *reinterpret_cast<f32x4*>(o + 64) = st_q;
      }
    }
  }
}

#ifndef CMR_WS_DBG
#define CMR_WS_DBG 0
#endif
#ifndef CMR_WINO_MW
#define CMR_WINO_MW 1
#endif
#ifdef CMR_AB_SWITCHES
static int g_wino_mw = CMR_WINO_MW;                  // cmr_set_wino_mfma_waves: A/B measurements, libcmr_hip_ab.so only
#else
static constexpr int g_wino_mw = CMR_WINO_MW;
#endif
// workgroups of a wave-specialised launch (= partial-sum slots of a statistics launch / 4)
inline unsigned wino_ws_grid(int64_t ntiles, int cu_budget, int slices);
template <int DBG, int MW>
int launch_wino_ws_t(const WinoArgs& a, int tiles_x, int tiles_y, int64_t ntiles, int cu_budget, int slices, hipStream_t stream) {
  constexpr int smem = WS_SMEM_FLOATS * (int)sizeof(float);
  static CmrSmemCache granted{};
  if (cmr_grant_smem(reinterpret_cast<const void*>(conv3x3_wino_ws_kernel<DBG, MW>), smem, granted) != CMR_OK) return CMR_ELAUNCH;
  const unsigned grid = wino_ws_grid(ntiles, cu_budget, slices);
  hipLaunchKernelGGL((conv3x3_wino_ws_kernel<DBG, MW>), dim3(grid), dim3(256 * (MW + 1)), smem, stream, a, tiles_x, tiles_y, (int)ntiles);
  return cmr_launch_status();
}
inline unsigned wino_ws_grid(int64_t ntiles, int cu_budget, int slices) {
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
  }
  if (cu_budget > 0 && cu_budget < cus) cus = cu_budget;
  if (cus < 8) cus = 8;
  cus -= cus % 8;                                          // a multiple of the XCD count keeps a workgroup's tiles on one XCD's band
  // time slicing (`slices` argument): n x as many workgroups, each walking 1 / n of the tiles -- a CU is handed back to the dispatcher
  // n times per launch, so a branch on another stream is served in between instead of after the launch; never below 8 tiles per workgroup
  int64_t want = cus;
  if (slices > 1) {
    int sl = slices > 64 ? 64 : slices;
    while (sl > 1 && ntiles / ((int64_t)cus * sl) < 8) --sl;
    want = (int64_t)cus * sl;
  }
  return (unsigned)(ntiles < want ? ntiles : want);
}
int launch_wino_ws(const WinoArgs& a, int tiles_x, int tiles_y, int64_t ntiles, int cu_budget, int slices, hipStream_t stream) {
  // DBG bits (compile-time ablations used while tuning: 1 no stores, 2 no epilogue, 4 no T hand-over, 8 no U loads, 16 no LDS
  // prefetch, 32 no MFMAs, 128 every tile stored over the same few tiles (stores without HBM write traffic), 64 s_memtime stamps of busy / barrier-wait cycles per wave into the buffer passed as `post`) are not
  // instantiated in the shipped library: tools/ab_build.sh cmr_agent_amd/csrc/conv_wino.hip <tag> -DCMR_WS_DBG=<mask>
  if (g_wino_mw == 2) return launch_wino_ws_t<CMR_WS_DBG, 2>(a, tiles_x, tiles_y, ntiles, cu_budget, slices, stream);
  return launch_wino_ws_t<CMR_WS_DBG, 1>(a, tiles_x, tiles_y, ntiles, cu_budget, slices, stream);
}

template <int NT>
int launch_wino(const WinoArgs& a, int tiles_x, int tiles_y, int64_t ntiles, hipStream_t stream) {
  constexpr int smem = wt_smem_floats<NT>() * (int)sizeof(float);
  static CmrSmemCache granted{};
  if (cmr_grant_smem(reinterpret_cast<const void*>(conv3x3_wino_kernel<NT>), smem, granted) != CMR_OK) return CMR_ELAUNCH;
  hipLaunchKernelGGL(conv3x3_wino_kernel<NT>, dim3((unsigned)ntiles), dim3(256), smem, stream, a, tiles_x, tiles_y);
  return cmr_launch_status();
}

}  // namespace

// cu_budget: CUs the PERSISTENT kernel (wave-specialised Winograd here, the two-team / matrix-class bf16 kernels in conv_bf16.hip) may
// occupy; 0 = all.  Those kernels fill a CU completely (registers), so a concurrent branch on another stream only runs between their
// launches; a caller that forks such a branch (the point tower beside the image tower) leaves it a few CUs for the duration.
// slices: workgroups per CU the persistent Winograd kernel is split into (<= 1: one workgroup per CU walks all of the CU's tiles).
// Both are ARGUMENTS of the call: the library keeps no launch policy of its own (round 3 had process-global setters here).
#ifdef CMR_AB_SWITCHES
static int CMR_WINO_WS = 1;      // wave-specialised persistent kernel for large maps (cmr_set_wino_variant: A/B measurements, libcmr_hip_ab.so only)
extern "C" int cmr_set_wino_mfma_waves(int per_simd) {
  const int old = g_wino_mw;
  g_wino_mw = per_simd == 2 ? 2 : 1;
  return old;
}
extern "C" int cmr_set_wino_variant(int wave_specialised) {
  const int old = CMR_WINO_WS;
  CMR_WINO_WS = wave_specialised & 1;
  return old;
}
#else
static constexpr int CMR_WINO_WS = 1;
#endif

// Training forward of a convolution that feeds a batch-statistics BatchNorm: y = conv(x) + bias (no residual / table / activation / pool) AND
// the sums the BatchNorm needs, from the helpers' epilogue of the wave-specialised kernel: part [parts][2][64] = per helper wave the sums of
// (y - bias) and (y - bias)^2 over its pixels.  parts = cmr_conv3x3_wino_stats_parts(...) (0: not served -- Cout = 64, Cin >= 64 and maps
// of >= 200 8x16 tiles are; the caller then runs the plain entry point and cmr_bn_stats_f32).  cmr_bn_stats_from_sums_f32 finishes.
extern "C" int64_t cmr_conv3x3_wino_stats_parts(int B, int H, int W, int Cin, int Cout, int cu_budget, int slices) {
  if (B <= 0 || H <= 0 || W <= 0 || Cout != 64 || Cin % 32 != 0 || Cin < 64 || !CMR_WINO_WS) return 0;
  const int tiles_x = (W + WT_TW - 1) / WT_TW, tiles_y = (H + WT_TH - 1) / WT_TH;
  const int64_t ntiles64 = (int64_t)tiles_x * tiles_y * B;
  if (ntiles64 < 200 || 2 * ntiles64 >= 0x7fffffff) return 0;
  return (int64_t)wino_ws_grid(ntiles64, cu_budget, slices) * 4;
}

extern "C" int cmr_conv3x3_wino_stats_nhwc_f32(const float* x, int B, int H, int W, int Cin, const float* u, const float* bias, float* y, int Cout,
                                               int cu_budget, int slices, float* part, int64_t parts, hipStream_t stream) {
  CMR_REQUIRE(x && u && y && part && cmr_aligned16(x) && cmr_aligned16(u) && cmr_aligned16(y) && cmr_aligned16(part) && (!bias || cmr_aligned16(bias)));
  const int64_t want = cmr_conv3x3_wino_stats_parts(B, H, W, Cin, Cout, cu_budget, slices);
  if (want == 0) return CMR_EUNSUPPORTED;
  CMR_REQUIRE(parts == want);
  const int tiles_x = (W + WT_TW - 1) / WT_TW, tiles_y = (H + WT_TH - 1) / WT_TH;
  const WinoArgs a{x, B, H, W, Cin, u, bias, nullptr, nullptr, y, Cout, 1.f, 1, part};
  return launch_wino_ws(a, tiles_x, tiles_y, (int64_t)tiles_x * tiles_y * B, cu_budget, slices, stream);
}

// Data gradient of a stride-1 3x3 convolution whose INPUT was lrelu_{bn_slope}(BatchNorm(bn_x)) and had no other consumer
// (ImageResNet.py:9-14 conv -> BatchNorm -> LeakyReLU -> conv): dx = conv(dy, u) (u = the transposed / flipped weights' G g G^T, no bias, no
// activation) AND the two sums of that BatchNorm's backward reduction over the pixels each helper wave finishes -- part [parts][2][64] as
// cmr_conv3x3_wino_stats_nhwc_f32 lays it out, the arithmetic of cmr_bn_bwd_f32's first pass (mask from the sign of bn_x * stat[2] + stat[3]).
// cmr_bn_bwd_from_sums_f32 finishes the BatchNorm backward without that pass over (dx, bn_x).  Same shapes as the statistics launch.
extern "C" int cmr_conv3x3_wino_bnbwd_nhwc_f32(const float* dy, int B, int H, int W, int Cin, const float* u, float* dx, int Cout, const float* bn_x,
                                               const float* bn_stat, float bn_slope, int cu_budget, int slices, float* part, int64_t parts,
                                               hipStream_t stream) {
  CMR_REQUIRE(dy && u && dx && bn_x && bn_stat && part && cmr_aligned16(dy) && cmr_aligned16(u) && cmr_aligned16(dx) && cmr_aligned16(bn_x) &&
              cmr_aligned16(bn_stat) && cmr_aligned16(part));
  if (!WS_BNBWD) return CMR_EUNSUPPORTED;              // (compiled out by default: see CMR_WS_BNBWD)
  const int64_t want = cmr_conv3x3_wino_stats_parts(B, H, W, Cin, Cout, cu_budget, slices);
  if (want == 0) return CMR_EUNSUPPORTED;
  CMR_REQUIRE(parts == want);
  const int tiles_x = (W + WT_TW - 1) / WT_TW, tiles_y = (H + WT_TH - 1) / WT_TH;
  const WinoArgs a{dy, B, H, W, Cin, u, nullptr, nullptr, nullptr, dx, Cout, 1.f, 1, part, bn_x, bn_stat, bn_slope};
  return launch_wino_ws(a, tiles_x, tiles_y, (int64_t)tiles_x * tiles_y * B, cu_budget, slices, stream);
}

extern "C" int cmr_conv3x3_wino_nhwc_f32(const float* x, int B, int H, int W, int Cin, const float* u, const float* bias,
                                         const float* res, const float* post, float* y, int Cout, float slope, int pool,
                                         int cu_budget, int slices, hipStream_t stream) {
  CMR_REQUIRE(x && u && y && B > 0 && H > 0 && W > 0 && Cin % 32 == 0 && Cin >= 32 && Cout % 64 == 0 && Cout >= 64);
  CMR_REQUIRE(cmr_aligned16(x) && cmr_aligned16(u) && cmr_aligned16(y) && (!bias || cmr_aligned16(bias)) &&
              (!res || cmr_aligned16(res)) && (!post || cmr_aligned16(post)));
  CMR_REQUIRE(pool == 1 || (pool == 2 && !res && !post));
  const int tiles_x = (W + WT_TW - 1) / WT_TW, tiles_y = (H + WT_TH - 1) / WT_TH;
  const WinoArgs a{x, B, H, W, Cin, u, bias, res, post, y, Cout, slope, pool};
  // 64 couts per workgroup unless that leaves the chip under-filled (2 x 256 resident workgroups): small maps
  // take 32-cout workgroups, twice as many and three per CU
  const int64_t ntiles64 = (int64_t)tiles_x * tiles_y * B * (Cout / 64);
  CMR_REQUIRE(2 * ntiles64 < 0x7fffffff);
  if (CMR_WINO_WS && Cin >= 64 && ntiles64 >= 200 && slope >= 0.f && slope <= 1.f) return launch_wino_ws(a, tiles_x, tiles_y, ntiles64, cu_budget, slices, stream);
  if (ntiles64 >= 512) return launch_wino<2>(a, tiles_x, tiles_y, ntiles64, stream);
  return launch_wino<1>(a, tiles_x, tiles_y, 2 * ntiles64, stream);
}
