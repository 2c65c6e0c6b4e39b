// Stride-1 / stride-2 3x3 convolution on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, fp32 accumulate) -- the bf16 variant of
// the convolution kernels (SURVEY.md 7 step 9 / BASELINE configs[2], configs[3]).  Activations stay fp32 NHWC in HBM (same
// buffers, same epilogue as the fp32 kernels: folded-BN bias, residual, LeakyReLU, positional table, fused 2x2 average
// pool); operands are rounded to bf16 (round-to-nearest-even, v_cvt_pk_bf16_f32) when the halo tile is staged in LDS,
// weights are packed to bf16 MFMA fragments when the module's plan is built.
//
// The bf16 MFMA rate is 16 x the fp32 one, so the direct form is already HBM-bound (a 64 -> 64 layer moves 512 B per pixel
// for 73.7 kFLOP: 3.5 GB / 8 TB/s = 0.44 ms at 352x1216x8 against 0.10 ms of matrix time) and Winograd's transforms
// would only cost accuracy.  What has to be organised is operand reuse, not multiplies:
//   * persistent workgroups (one per CU): the 9 x Cin x (32 NT) weight slice of the workgroup's cout group is loaded
//     into LDS ONCE (72 KB as ready-made A fragments, 1 KB per wave read) and reused for every spatial tile it processes;
//   * per tile the (8+2) x (TW+2) pixel halo is converted to bf16 into LDS (pixel stride Cin*2 + 16 bytes: conflict-free
//     ds_read_b128 of 8 channels for 16 consecutive pixels), the next tile's global loads are already in flight in
//     registers while the current tile is multiplied (one wave per SIMD: up to 512 VGPRs, the prefetch costs nothing);
//   * B operand = 32 pixels x 16 channels of one tap straight from the halo image (lane = pixel), A = weight fragment;
//     every MFMA is fed by at most two ds_read_b128, which the LDS array sustains (MI355X_MICROARCH.md, LDS issue rates).
#include "cmr_common.h"
#include <type_traits>


namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

struct B16Args {
  const float* x; int B, H, W;
  const void* wfrag;     // [Cout/(32 NT)][9 taps][Cin/16][NT][64 lanes][8] bf16
  const float* bias; const float* res; const float* post;
  float* y; int Cout; float slope; int pool;
  int tiles_x, tiles_y;
  // 0 / 1 multipliers of the optional operands' offsets, set by the HOST: a null operand is read at offset 0 of a zero page.  Written
  // as `ptr ? offset : 0` in the kernel, hipcc turns every such select into a branch around the address arithmetic, and one branch
  // in the tile loop degrades each later s_waitcnt to vmcnt(0) (81 of them in the first version of this kernel)
  int res_mul, bias_mul;
  int x_bf16, y_bf16;    // activations stored as bf16 NHWC (input / output): a convolution whose output only feeds another bf16 convolution
                         // writes the bf16 values that one would round its fp32 input to anyway -- same results, half the bytes
  int res_bf16;          // the residual operand is a bf16 NHWC map (bf16-STORED towers: the block input of a ResidualBlock, round 3)
  int wnt;               // cout tiles per group in the fragment layout of wfrag (the matrix-class kernel walks it by 32-cout tile)
  int cu_budget;         // host only: CUs the persistent kernels may occupy (0 = all); an argument of the entry points
  // input prologue of the matrix-class kernel (training, cmr_conv3x3_bf16_pro_nhwc_f32): x is the PREVIOUS layer's BatchNorm input and the
  // operand is lrelu_{in_slope}(x * in_scale + in_shift) per input channel, formed in the fp32 -> bf16 staging pass (zero padding after it)
  const float* in_scale; const float* in_shift; float in_slope;
};

__device__ __attribute__((aligned(16))) float b16_zero16[4] = {0.f, 0.f, 0.f, 0.f};

// four floats -> four bf16 (round to nearest even, v_cvt_pk_bf16_f32) as 8 bytes
__device__ __forceinline__ uint2 b16_pack4(const f32x4& v) {
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  uint2 w;
  w.x = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){v[0], v[1]}, bf16x2_t));
  w.y = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){v[2], v[3]}, bf16x2_t));
  return w;
}

// TW = 32: wave w owns rows 2w, 2w+1 as two 32-pixel blocks (lane = column).
// TW = 16: wave w owns rows 2w, 2w+1 as ONE block (lane = 16 (row & 1) + column).
// S = stride (1 | 2): the output tile stays TH x TW, the halo image grows to ((TH - 1) S + 3) x ((TW - 1) S + 3) input pixels and a
// lane's pixel sits at (S row + ky, S column + kx) in it (S = 2: 288 bytes between neighbouring lanes = 8 banks apart, still
// conflict free).  a.H / a.W are the INPUT sizes; the output is ((H - 1) / S + 1) x ((W - 1) / S + 1).
template <int CIN, int NT, int TW, int S = 1, bool POOL = false, bool POST = false, bool OUT16 = false>
__global__ __launch_bounds__(256) void conv3x3_bf16_kernel(const B16Args a) {
  constexpr int KS = CIN / 16, NB = TW / 16;
  constexpr int TH = 8, HR = (TH - 1) * S + 3, HC = (TW - 1) * S + 3;
  constexpr int PS = CIN * 2 + 16;                      // bytes per halo pixel
  constexpr int WBYTES = 9 * KS * NT * 1024;
  constexpr int C4 = CIN / 4;
  constexpr int NPIECE = HR * HC * C4;
  constexpr int NLOAD = (NPIECE + 255) / 256;
  // halo images kept in flight in registers: two where the register budget (512 per lane at one wave per SIMD) allows it
  constexpr int NBUF = (NLOAD * 8 + NT * NB * (POST ? 48 : 32)) <= 280 ? 2 : 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* Ws = smem;
  unsigned char* Xs = smem + WBYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  const int ngroups = a.Cout / (32 * NT);
  // Workgroup -> (cout group, first tile, tile stride).  Workgroups land on the 8 XCDs round-robin by blockIdx and every XCD has
  // its own L2: when the grid divides evenly and there are several cout groups, the `ngroups` workgroups that walk over the SAME
  // tiles (one per cout group) and the families that take NEIGHBOURING tiles are put on one XCD, and every XCD gets one contiguous
  // band of tiles -- the input is then fetched from HBM once per XCD instead of once per cout group (88x304 128->128: 188 -> 177 us).
  // With ONE cout group the banded order measured 2-7 % slower at 352x1216, so it keeps the plain round-robin order.
  int group, s, step, s_end;
  {
    const int nsp_all = a.B * a.tiles_y * a.tiles_x;
    const int nfam = gridDim.x / ngroups;                     // families = workgroups per cout group
    if (nfam % 8 == 0 && ngroups > 1) {
      const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
      const int fam_x = nfam >> 3;                            // families per XCD
      group = slot % ngroups;
      const int fam = slot / ngroups;                         // 0 .. fam_x - 1
      const int band0 = (int)((int64_t)nsp_all * xcd / 8), band1 = (int)((int64_t)nsp_all * (xcd + 1) / 8);
      s = band0 + fam; step = fam_x; s_end = band1;
    } else {
      group = blockIdx.x % ngroups;
      s = blockIdx.x / ngroups; step = nfam; s_end = nsp_all;
    }
  }
  const int co0 = group * 32 * NT;
  {
    const uint4* src = reinterpret_cast<const uint4*>(static_cast<const unsigned char*>(a.wfrag) + (size_t)group * WBYTES);
    for (int i = tid; i < WBYTES / 16; i += 256) reinterpret_cast<uint4*>(Ws)[i] = src[i];
  }
  const int nsp = s_end;
  if (s >= nsp) return;                                  // (uniform) nothing to do for this workgroup

  struct Tile { int b, oy0, ox0; };
  auto decode = [&](int st) __attribute__((always_inline)) {
    Tile t;
    t.ox0 = (st % a.tiles_x) * TW; st /= a.tiles_x;
    t.oy0 = (st % a.tiles_y) * TH;
    t.b = st / a.tiles_y;
    return t;
  };
  // The tile loop is straight-line code with no load in its epilogue.  What that took (each item was a measured stall):
  //  * optional operands are read through host-made 0 / 1 offset multipliers (B16Args) from a zero page instead of `ptr ? .. : ..`
  //    (hipcc turns the selects into branches, and a branch degrades every later s_waitcnt to vmcnt(0));
  //  * the folded-BN bias is loaded ONCE per workgroup; residual (and table) rows of the tile are requested before its multiplies,
  //    AHEAD of the next halo image in issue order (loads retire in order: the epilogue then waits with a counted vmcnt that leaves
  //    the halo prefetch in flight).  The first version loaded bias / residual / table inside the epilogue, where the second pixel
  //    block's loads were sunk into the first block's predicated store region: 16 dependent loads, each behind s_waitcnt vmcnt(0),
  //    ~10 of the 12.7 us a tile took;
  //  * every output of the tile is computed in place in the accumulators and pinned before the first predicated store.
  f32x4 pv[NBUF][NLOAD];
  f32x4 rv[NB][NT][4];
  f32x4 tv[POST ? NB : 1][POST ? NT : 1][4];
  f32x4 bsr[NT][4];
  const int Ho = (a.H - 1) / S + 1, Wo = (a.W - 1) / S + 1;
  const float* rbase = a.res ? a.res : b16_zero16;       // pointer selects once, outside the tile loop
  const float* bbase = a.bias ? a.bias : b16_zero16;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int q = 0; q < 4; ++q) bsr[nt][q] = *reinterpret_cast<const f32x4*>(bbase + (co0 + nt * 32 + q * 8 + 4 * h) * a.bias_mul);
  // this lane's pixel inside the tile, per block
  int prow[NB], pcol[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    prow[nb] = TW == 32 ? 2 * wave + nb : 2 * wave + (l31 >> 4);
    pcol[nb] = TW == 32 ? l31 : (l31 & 15);
  }
  auto issue_loads = [&](int st, f32x4 (&dst)[NLOAD]) __attribute__((always_inline)) {   // branch-free: clamped addresses, padding applied at the LDS store
    const Tile t = decode(st);
    const float* xb = a.x + (int64_t)t.b * a.H * a.W * CIN;
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
      int e = tid + 256 * i;
      e = e < NPIECE ? e : NPIECE - 1;
      const int p = e / C4, c = e - p * C4;
      int iy = t.oy0 * S - 1 + p / HC, ix = t.ox0 * S - 1 + p % HC;
      iy = iy < 0 ? 0 : (iy >= a.H ? a.H - 1 : iy);
      ix = ix < 0 ? 0 : (ix >= a.W ? a.W - 1 : ix);
      dst[i] = *reinterpret_cast<const f32x4*>(xb + (unsigned)((iy * a.W + ix) * CIN + 4 * c));
    }
  };
  auto store_lds = [&](int st, const f32x4 (&src)[NLOAD]) __attribute__((always_inline)) {
    const Tile t = decode(st);
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
      const int e = tid + 256 * i;
      if (e < NPIECE) {
        const int p = e / C4, c = e - p * C4;
        const int iy = t.oy0 * S - 1 + p / HC, ix = t.ox0 * S - 1 + p % HC;
        const bool inb = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        bf16x4 v;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (__bf16)(inb ? src[i][k] : 0.f);
        *reinterpret_cast<bf16x4*>(Xs + p * PS + c * 8) = v;
      }
    }
  };
  auto issue_rows = [&](const Tile& t) __attribute__((always_inline)) {   // residual (+ table) rows of this tile; null operands: the zero page
    if constexpr (!POOL) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int oy = t.oy0 + prow[nb], ox = t.ox0 + pcol[nb];
        const bool ok = oy < Ho && ox < Wo;
        const int oyc = ok ? oy : 0, oxc = ok ? ox : 0;
        const int64_t pix = ((int64_t)t.b * Ho + oyc) * Wo + oxc;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int cq = co0 + nt * 32 + q * 8 + 4 * h;
            rv[nb][nt][q] = *reinterpret_cast<const f32x4*>(rbase + (pix * a.Cout + cq) * a.res_mul);
            if constexpr (POST) tv[nb][nt][q] = *reinterpret_cast<const f32x4*>(a.post + ((int64_t)oyc * Wo + oxc) * a.Cout + cq);
          }
      }
    }
  };
  issue_loads(s, pv[0]);
  if (NBUF == 2) issue_loads(s + step < nsp ? s + step : s, pv[NBUF - 1]);
  auto tile_pass = [&](auto PTAG) __attribute__((always_inline)) {
    constexpr int P = decltype(PTAG)::value % NBUF;
    __syncthreads();                                    // weights resident (first pass) / everybody done with the previous halo image
    store_lds(s, pv[P]);
    __syncthreads();
    const Tile t = decode(s);
    issue_rows(t);
    issue_loads(s + NBUF * step < nsp ? s + NBUF * step : s, pv[P]);   // past the end: a harmless re-read instead of a branch around the loads
    f32x16 acc[NT][NB];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][nb][r] = 0.f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap % 3;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        bf16x8 av[NT], bv[NB];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          av[nt] = *reinterpret_cast<const bf16x8*>(Ws + ((tap * KS + ks) * NT + nt) * 1024 + lane * 16);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          bv[nb] = *reinterpret_cast<const bf16x8*>(Xs + ((prow[nb] * S + ky) * HC + pcol[nb] * S + kx) * PS + ks * 32 + h * 16);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) acc[nt][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[nt], bv[nb], acc[nt][nb], 0, 0, 0);
      }
    }
    // ---- epilogue: register 4q+e of tile nt = channel co0 + 32 nt + 8q + 4h + e of this lane's pixel
    if constexpr (POOL) {
      // LeakyReLU(conv + bias), then the 2x2 mean: rows 2w / 2w+1 are the two blocks (TW = 32) or lane halves 16 apart
      // (TW = 16); the column partner is lane ^ 1
      const int py = (t.oy0 >> 1) + wave, px = (t.ox0 >> 1) + (pcol[0] >> 1);
      const bool writer = (l31 & 1) == 0 && (TW == 32 || (l31 & 16) == 0) && py < (Ho >> 1) && px < (Wo >> 1);
      f32x4 pooled[NT][4];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float v = acc[nt][nb][4 * q + e] + bsr[nt][q][e];
              sum[e] += v > 0.f ? v : v * a.slope;
            }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = sum[e];
            v += cmr_xor1(v);
            if (TW == 16) v += cmr_xor16(v);
            sum[e] = 0.25f * v;
          }
          pooled[nt][q] = sum;
        }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) cmr_pin(pooled[nt][q]);
      if (writer) {
        float* yp = a.y + (((int64_t)t.b * (Ho >> 1) + py) * (Wo >> 1) + px) * a.Cout + co0 + 4 * h;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(yp + nt * 32 + q * 8) = pooled[nt][q];
      }
    } else {
      f32x4 ov[NB][NT][4];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float u = acc[nt][nb][4 * q + e] + bsr[nt][q][e] + rv[nb][nt][q][e];
              v[e] = u > 0.f ? u : u * a.slope;
            }
            if constexpr (POST) v += tv[nb][nt][q];
            ov[nb][nt][q] = v;
          }
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int q = 0; q < 4; ++q) cmr_pin(ov[nb][nt][q]);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int oy = t.oy0 + prow[nb], ox = t.ox0 + pcol[nb];
        if (oy < Ho && ox < Wo) {
          const int64_t o = (((int64_t)t.b * Ho + oy) * Wo + ox) * a.Cout + co0 + 4 * h;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              if constexpr (OUT16) *reinterpret_cast<uint2*>(reinterpret_cast<__bf16*>(a.y) + o + nt * 32 + q * 8) = b16_pack4(ov[nb][nt][q]);
              else *reinterpret_cast<f32x4*>(a.y + o + nt * 32 + q * 8) = ov[nb][nt][q];
            }
        }
      }
    }
  };
  for (;;) {                                            // two passes per trip: the register sets alternate under static names
    tile_pass(std::integral_constant<int, 0>{});
    s += step;
    if (s >= nsp) break;
    tile_pass(std::integral_constant<int, 1>{});
    s += step;
    if (s >= nsp) break;
  }
}

template <int CIN, int NT, int TW, int S, bool POOL, bool POST, bool OUT16 = false>
int launch_b16p(B16Args a, hipStream_t stream) {
  constexpr int smem = 9 * (CIN / 16) * NT * 1024 + (7 * S + 3) * ((TW - 1) * S + 3) * (CIN * 2 + 16);
  static_assert(smem <= 160 * 1024, "weight slice + halo image must fit in LDS");
  static CmrSmemCache granted{};
  if (cmr_grant_smem(reinterpret_cast<const void*>(conv3x3_bf16_kernel<CIN, NT, TW, S, POOL, POST, OUT16>), smem, granted) != CMR_OK) return CMR_ELAUNCH;
  a.tiles_x = ((a.W - 1) / S + 1 + TW - 1) / TW;
  a.tiles_y = ((a.H - 1) / S + 1 + 7) / 8;
  const int ngroups = a.Cout / (32 * NT);
  const int64_t nsp = (int64_t)a.B * a.tiles_x * a.tiles_y;
  int per_group = 256 / ngroups;                       // one persistent workgroup per CU
  if (per_group < 1) per_group = 1;
  if (per_group > nsp) per_group = (int)nsp;
  hipLaunchKernelGGL((conv3x3_bf16_kernel<CIN, NT, TW, S, POOL, POST, OUT16>), dim3(ngroups * per_group), dim3(256), smem, stream, a);
  return cmr_launch_status();
}

template <int CIN, int NT, int TW, int S = 1>
int launch_b16(B16Args a, hipStream_t stream) {
  a.res_mul = a.res ? 1 : 0; a.bias_mul = a.bias ? 1 : 0;
  if constexpr (S == 1) {
    if (a.pool == 2) return launch_b16p<CIN, NT, TW, S, true, false>(a, stream);
    if (a.post) return launch_b16p<CIN, NT, TW, S, false, true>(a, stream);
  } else {
    if (a.post) return a.y_bf16 ? CMR_EUNSUPPORTED : launch_b16p<CIN, NT, TW, S, false, true>(a, stream);
    if (a.y_bf16) return launch_b16p<CIN, NT, TW, S, false, false, true>(a, stream);
  }
  if (a.y_bf16) return CMR_EUNSUPPORTED;               // bf16 output of the one-team kernel: the strided instance only
  return launch_b16p<CIN, NT, TW, S, false, false>(a, stream);
}

// ---- two-team kernel (stride 1) ------------------------------------------------------------------------------------------------
// The kernel above runs ONE wave per SIMD, so nothing overlaps: a tile is [convert + stage the halo] [issue the next loads] [multiply]
// [epilogue + stores], back to back (7.5 us per 256 pixels at 64 -> 64, of which 2.2 us multiply).  Here a workgroup is TWO teams of
// four waves (two waves per SIMD) that share the weight slice in LDS and own one 8x16-pixel halo buffer each; while one team
// multiplies, the other finishes its previous tile, stages its next one and requests the one after -- the hardware interleaves the two
// waves of a SIMD, no instruction scheduling is asked of the compiler.  Both teams run the SAME straight-line loop
//     M(u) ; barrier ; X(u) ; barrier          M = multiply unit u from the team's buffer, X = [epilogue] + stage u+1 + request u+2
// with team 1 delayed by one barrier, so M of one team always faces X of the other.  A unit is a (tile, 64-channel K chunk): Cin = 128
// takes two units per tile through the same 26 KB buffer, which is what lets two buffers sit next to the 72 KB of weights.
// S = 2 (the strided convolutions of the down-sampling blocks, Cin = 64, NT = 2): a team's tile is 4 x 16 OUTPUT pixels (9 x 33 input
// pixels: two 42 KB buffers still fit next to the weights), and its four waves are 2 pixel blocks x 2 cout tiles -- one accumulator each.
template <int CIN, int NT, bool POOL, bool POST, int IO = 0, int S = 1>      // IO: bit 0 = bf16 input, bit 1 = bf16 output, bit 2 = bf16 residual
__global__ __launch_bounds__(512) void conv3x3_bf16_tt_kernel(const B16Args a) {
  constexpr bool IN16 = (IO & 1) != 0, OUT16 = (IO & 2) != 0, RES16 = (IO & 4) != 0;
  static_assert(S == 1 || (S == 2 && NT == 2 && CIN == 64 && !POOL), "strided instance: 64 -> 64 k");
  static_assert(!RES16 || (!POOL && !POST), "a bf16 residual goes with plain epilogues");
  constexpr int NTW = S == 1 ? NT : 1;                  // cout tiles per wave
  constexpr int KC = CIN / 64, KS = CIN / 16;
  constexpr int TH = S == 1 ? 8 : 4, TW = 16, HR = (TH - 1) * S + 3, HC = (TW - 1) * S + 3;
  constexpr int PS = 64 * 2 + 16;                       // bytes per halo pixel of one K chunk
  constexpr int WBYTES = 9 * KS * NT * 1024;
  constexpr int XBYTES = HR * HC * PS;
  constexpr int PPP = IN16 ? 8 : 16;                    // 16-byte pieces per halo pixel of one unit (8 bf16 | 4 floats each)
  constexpr int NPIECE = HR * HC * PPP;
  constexpr int NLOAD = (NPIECE + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int team = wave >> 2, tw = wave & 3, ttid = tid & 255;
  unsigned char* Ws = smem;
  unsigned char* Xs = smem + WBYTES + team * XBYTES;
  const int h = lane >> 5, l31 = lane & 31;
  const int ngroups = a.Cout / (32 * NT);
  int group, s0, step, s_end;
  {
    const int nsp_all = a.B * a.tiles_y * a.tiles_x;
    const int nfam = gridDim.x / ngroups;
    if (nfam % 8 == 0 && ngroups > 1) {                 // cout groups of one tile and neighbouring tiles on one XCD (see above)
      const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
      const int fam_x = nfam >> 3;
      group = slot % ngroups;
      const int band0 = (int)((int64_t)nsp_all * xcd / 8), band1 = (int)((int64_t)nsp_all * (xcd + 1) / 8);
      s0 = band0 + slot / ngroups; step = fam_x; s_end = band1;
    } else {
      group = blockIdx.x % ngroups;
      s0 = blockIdx.x / ngroups; step = nfam; s_end = nsp_all;
    }
  }
  const int co0 = group * 32 * NT;
  {
    const uint4* src = reinterpret_cast<const uint4*>(static_cast<const unsigned char*>(a.wfrag) + (size_t)group * WBYTES);
    for (int i = tid; i < WBYTES / 16; i += 512) reinterpret_cast<uint4*>(Ws)[i] = src[i];
  }
  if (s0 >= s_end) return;                              // (uniform over the workgroup)
  // tiles of this workgroup: s0 + j step; team 0 takes the even j, team 1 the odd ones; both make the same number of passes (a team
  // that runs out re-reads its last tile and keeps its stores to itself), because every wave has to meet every barrier
  const int ntile_wg = (s_end - s0 + step - 1) / step;
  const int npass = (ntile_wg + 1) / 2;
  const int s_last = s0 + (ntile_wg - 1) * step;
  auto tile_of = [&](int i) __attribute__((always_inline)) { const int sj = s0 + (2 * i + team) * step; return sj < s_end ? sj : s_last; };

  struct Tile { int b, oy0, ox0; };
  auto decode = [&](int st) __attribute__((always_inline)) {
    Tile t;
    t.ox0 = (st % a.tiles_x) * TW; st /= a.tiles_x;
    t.oy0 = (st % a.tiles_y) * TH;
    t.b = st / a.tiles_y;
    return t;
  };
  // ONE halo image in flight per team, requested a whole period (two phases) before it is staged.  Two register sets for Cin = 128
  // (each chunk requested a tile ahead) measured 5 % slower.
  f32x4 pv[1][NLOAD];
  f32x4 rv[RES16 ? 1 : NTW][4];
  uint2 rw[RES16 ? NTW : 1][4];                         // bf16 residual: the raw words, widened in the epilogue (nothing touches a loaded
                                                        // value before its use: a conversion here would drag s_waitcnt vmcnt(0) in front of the multiplies)
  f32x4 tv[POST ? NTW : 1][4];
  const int Ho = (a.H - 1) / S + 1, Wo = (a.W - 1) / S + 1;
  const int nt0 = S == 1 ? 0 : (tw >> 1);               // first cout tile of this wave
  const float* rbase = a.res ? a.res : b16_zero16;
  // the folded-BN bias of this cout group sits in LDS (the epilogue runs in the X phase, where an LDS read costs nothing and a
  // register array would cost 8 NT registers of a 256-register budget)
  float* Bs = reinterpret_cast<float*>(smem + WBYTES + 2 * XBYTES);
  if (tid < 32 * NT) Bs[tid] = (a.bias ? a.bias : b16_zero16)[(co0 + tid) * a.bias_mul];
  const int prow = 2 * (S == 1 ? tw : (tw & 1)) + (l31 >> 4), pcol = l31 & 15;       // this lane's pixel inside the tile
  auto issue_loads = [&](const Tile& t, int kc, f32x4 (&dst)[NLOAD]) __attribute__((always_inline)) {
    constexpr int ES = IN16 ? 2 : 4;                    // bytes per stored activation
    const unsigned char* xb = reinterpret_cast<const unsigned char*>(a.x) + ((int64_t)t.b * a.H * a.W * CIN + kc * 64) * ES;
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
      int e = ttid + 256 * i;
      e = e < NPIECE ? e : NPIECE - 1;
      const int p = e / PPP, c = e % PPP;
      int iy = t.oy0 * S - 1 + p / HC, ix = t.ox0 * S - 1 + p % HC;
      iy = iy < 0 ? 0 : (iy >= a.H ? a.H - 1 : iy);
      ix = ix < 0 ? 0 : (ix >= a.W ? a.W - 1 : ix);
      dst[i] = *reinterpret_cast<const f32x4*>(xb + (unsigned)((iy * a.W + ix) * CIN * ES + 16 * c));
    }
  };
  auto store_lds = [&](const Tile& t, const f32x4 (&src)[NLOAD]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
      const int e = ttid + 256 * i;
      if (e < NPIECE) {
        const int p = e / PPP, c = e % PPP;
        const int iy = t.oy0 * S - 1 + p / HC, ix = t.ox0 * S - 1 + p % HC;
        const bool inb = ((unsigned)iy < (unsigned)a.H) & ((unsigned)ix < (unsigned)a.W);
        const unsigned keep = inb ? 0xffffffffu : 0u;
        if constexpr (IN16) {                           // already bf16: 16 bytes = 8 channels, the padding as a mask
          uint4 w = __builtin_bit_cast(uint4, src[i]);
          w.x &= keep; w.y &= keep; w.z &= keep; w.w &= keep;
          *reinterpret_cast<uint4*>(Xs + p * PS + c * 16) = w;
        } else {
          // two packed conversions (v_cvt_pk_bf16_f32) and the padding as a mask on the packed words: 5 instead of 10 instructions per piece
          uint2 w = b16_pack4(src[i]);
          w.x &= keep; w.y &= keep;
          *reinterpret_cast<uint2*>(Xs + p * PS + c * 8) = w;
        }
      }
    }
  };
  auto issue_rows = [&](const Tile& t) __attribute__((always_inline)) {
    if constexpr (!POOL) {
      const int oy = t.oy0 + prow, ox = t.ox0 + pcol;
      const bool ok = oy < Ho && ox < Wo;
      const int oyc = ok ? oy : 0, oxc = ok ? ox : 0;
      const int64_t pix = ((int64_t)t.b * Ho + oyc) * Wo + oxc;
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int cq = co0 + (nt0 + nt) * 32 + q * 8 + 4 * h;
          if constexpr (RES16) rw[nt][q] = *reinterpret_cast<const uint2*>(reinterpret_cast<const __bf16*>(rbase) + (pix * a.Cout + cq) * a.res_mul);
          else rv[nt][q] = *reinterpret_cast<const f32x4*>(rbase + (pix * a.Cout + cq) * a.res_mul);
          if constexpr (POST) tv[nt][q] = *reinterpret_cast<const f32x4*>(a.post + ((int64_t)oyc * Wo + oxc) * a.Cout + cq);
        }
    }
  };
  f32x16 acc[NTW];
  // 36 steps (9 taps x 4 k-steps of the chunk), each NT weight fragments + 1 pixel fragment from LDS feeding NT matrix instructions.
  // hipcc emits read / s_waitcnt lgkmcnt(0) / multiply per step, i.e. the LDS latency in front of every instruction -- and that is
  // the faster form HERE: issuing the reads one or two steps ahead into rotating register sets (measured, 2 and 3 sets, with scheduling
  // barriers) made every shape 3-6 % slower, because the other team's X phase lives in exactly those gaps.
  auto multiply = [&](int kc) __attribute__((always_inline)) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap % 3;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bf16x8 av[NTW];
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
          av[nt] = *reinterpret_cast<const bf16x8*>(Ws + ((tap * KS + kc * 4 + ks) * NT + nt0 + nt) * 1024 + lane * 16);
        const bf16x8 bv = *reinterpret_cast<const bf16x8*>(Xs + ((prow * S + ky) * HC + pcol * S + kx) * PS + ks * 32 + h * 16);
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[nt], bv, acc[nt], 0, 0, 0);
      }
    }
  };
  auto epilogue = [&](const Tile& t, bool live) __attribute__((always_inline)) {
    // register 4q+e of tile nt = channel co0 + 32 nt + 8q + 4h + e of this lane's pixel
    f32x4 ov[NTW][4];
    if constexpr (POOL) {
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = acc[nt][4 * q + e] + Bs[(nt0 + nt) * 32 + q * 8 + 4 * h + e];
            v = v > 0.f ? v : v * a.slope;
            v += cmr_xor1(v);                  // column partner
            v += cmr_xor16(v);                 // row partner (rows 2 tw / 2 tw + 1 sit 16 lanes apart)
            ov[nt][q][e] = 0.25f * v;
          }
    } else {
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 rr;
          if constexpr (RES16) {
            const uint2 w = rw[nt][q];                  // bf16 -> fp32: the 16 bits are the top half of the float
            rr = f32x4{__builtin_bit_cast(float, w.x << 16), __builtin_bit_cast(float, w.x & 0xffff0000u),
                       __builtin_bit_cast(float, w.y << 16), __builtin_bit_cast(float, w.y & 0xffff0000u)};
          } else {
            rr = rv[nt][q];
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float u = acc[nt][4 * q + e] + Bs[(nt0 + nt) * 32 + q * 8 + 4 * h + e] + rr[e];
            ov[nt][q][e] = u > 0.f ? u : u * a.slope;
          }
          if constexpr (POST) ov[nt][q] += tv[nt][q];
        }
    }
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
      for (int q = 0; q < 4; ++q) cmr_pin(ov[nt][q]);
    auto put = [&](int64_t o) __attribute__((always_inline)) {   // the lane's 4 x NTW quads at element offset o (+ 32 nt + 8 q), fp32 or bf16
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if constexpr (OUT16) *reinterpret_cast<uint2*>(reinterpret_cast<__bf16*>(a.y) + o + (nt0 + nt) * 32 + q * 8) = b16_pack4(ov[nt][q]);
          else *reinterpret_cast<f32x4*>(a.y + o + (nt0 + nt) * 32 + q * 8) = ov[nt][q];
        }
    };
    if constexpr (POOL) {
      const int py = (t.oy0 >> 1) + tw, px = (t.ox0 >> 1) + (pcol >> 1);
      if (live && (l31 & 17) == 0 && py < (Ho >> 1) && px < (Wo >> 1)) {
        put((((int64_t)t.b * (Ho >> 1) + py) * (Wo >> 1) + px) * a.Cout + co0 + 4 * h);
      }
    } else {
      const int oy = t.oy0 + prow, ox = t.ox0 + pcol;
      if (live && oy < Ho && ox < Wo) put((((int64_t)t.b * Ho + oy) * Wo + ox) * a.Cout + co0 + 4 * h);
    }
  };

  // prologue: unit 0 staged, the next ones (and the first tile's rows, if its last chunk comes next) requested
  Tile tcur = decode(tile_of(0));
  Tile tnext = decode(tile_of(npass > 1 ? 1 : 0));
  issue_loads(tcur, 0, pv[0]);
  store_lds(tcur, pv[0]);
  if constexpr (KC == 2) {
    issue_loads(tcur, 1, pv[0]);
  } else {
    issue_rows(tcur);
    issue_loads(tnext, 0, pv[0]);
  }
  __syncthreads();                                      // weights and both teams' first units are in LDS
  // Team 1's waves are the younger of every SIMD's pair and lose the issue arbitration in every phase (by age); one static priority
  // for that half, no per-phase flips: mid-size maps -8...10 % (176x608 64->64 159 -> 144 us, 88x304 128->128 125 -> 115 us), the
  // HBM-bound full-resolution maps unchanged
  if (team == 1) __builtin_amdgcn_s_setprio(1);
  if (team == 1) __syncthreads();                       // team 1 runs one phase behind
  for (int i = 0; i < npass; ++i) {
    // past the end: a harmless re-read instead of a branch around the loads
    const Tile tnn = decode(tile_of(i + 2 < npass ? i + 2 : (i + 1 < npass ? i + 1 : i)));
    const bool live = s0 + (2 * i + team) * step < s_end;
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
    if constexpr (KC == 2) {
      multiply(0);
      __syncthreads();
      store_lds(tcur, pv[0]);                           // (tile, chunk 1)
      issue_rows(tcur);                                 // ahead of the halo request in issue order: the epilogue's wait leaves that in flight
      issue_loads(tnext, 0, pv[0]);
      __syncthreads();
      multiply(1);
      __syncthreads();
      epilogue(tcur, live);
      store_lds(tnext, pv[0]);                          // (next tile, chunk 0)
      issue_loads(tnext, 1, pv[0]);
      __syncthreads();
    } else {
      multiply(0);
      __syncthreads();
      epilogue(tcur, live);
      store_lds(tnext, pv[0]);
      issue_rows(tnext);
      issue_loads(tnn, 0, pv[0]);
      __syncthreads();
    }
    tcur = tnext;
    tnext = tnn;
  }
  if (team == 0) __syncthreads();                       // team 0 waits out team 1's last phase
}

template <int CIN, int NT, bool POOL, bool POST, int IO, int S = 1>
int launch_tt_p(B16Args a, hipStream_t stream) {
  constexpr int TH = S == 1 ? 8 : 4;
  constexpr int smem = 9 * (CIN / 16) * NT * 1024 + 2 * ((TH - 1) * S + 3) * (15 * S + 3) * 144 + 256;
  static_assert(smem <= 160 * 1024, "weight slice + two halo buffers must fit in LDS");
  static CmrSmemCache granted{};
  if (cmr_grant_smem(reinterpret_cast<const void*>(conv3x3_bf16_tt_kernel<CIN, NT, POOL, POST, IO, S>), smem, granted) != CMR_OK) return CMR_ELAUNCH;
  a.tiles_x = ((a.W - 1) / S + 1 + 15) / 16;
  a.tiles_y = ((a.H - 1) / S + 1 + TH - 1) / TH;
  const int ngroups = a.Cout / (32 * NT);
  const int64_t nsp = (int64_t)a.B * a.tiles_x * a.tiles_y;
  const int cus = a.cu_budget >= 8 && a.cu_budget < 256 ? a.cu_budget - a.cu_budget % 8 : 256;
  int per_group = cus / ngroups;                       // one persistent workgroup per CU
  if (per_group < 1) per_group = 1;
  if (per_group > (nsp + 1) / 2) per_group = (int)((nsp + 1) / 2);      // a workgroup has two teams
  hipLaunchKernelGGL((conv3x3_bf16_tt_kernel<CIN, NT, POOL, POST, IO, S>), dim3(ngroups * per_group), dim3(512), smem, stream, a);
  return cmr_launch_status();
}

template <int CIN, int NT, int IO>
int launch_tt_io(B16Args a, hipStream_t stream) {
  if (a.pool == 2) return launch_tt_p<CIN, NT, true, false, IO>(a, stream);
  if (a.post) {
    if constexpr (IO == 0) return launch_tt_p<CIN, NT, false, true, 0>(a, stream);
    else return CMR_EUNSUPPORTED;                      // the table operand only occurs with fp32 activations
  }
  return launch_tt_p<CIN, NT, false, false, IO>(a, stream);
}

template <int CIN, int NT>
int launch_tt(B16Args a, hipStream_t stream) {
  a.res_mul = a.res ? 1 : 0; a.bias_mul = a.bias ? 1 : 0;
  if (a.res && a.res_bf16) {
    // bf16 residual (the input of a ResidualBlock in a bf16-stored tower): 64 -> 64 layers with a bf16 input, plain epilogue
    if constexpr (CIN == 64 && NT == 2) {
      if (a.pool == 2 || a.post || !a.x_bf16) return CMR_EUNSUPPORTED;
      return a.y_bf16 ? launch_tt_p<CIN, NT, false, false, 7>(a, stream) : launch_tt_p<CIN, NT, false, false, 5>(a, stream);
    } else {
      return CMR_EUNSUPPORTED;
    }
  }
  switch ((a.x_bf16 ? 1 : 0) | (a.y_bf16 ? 2 : 0)) {
    case 1: return launch_tt_io<CIN, NT, 1>(a, stream);
    case 2: return launch_tt_io<CIN, NT, 2>(a, stream);
    case 3: return launch_tt_io<CIN, NT, 3>(a, stream);
    default: return launch_tt_io<CIN, NT, 0>(a, stream);
  }
}

// ---- matrix-class kernel (stride 1, 128-cout groups, no residual / table) -----------------------------------------------------------
// The kernels above are built for the HBM-class layers (64 -> 64: 74 kFLOP per 512 B).  A 128 -> 128 layer does 295 kFLOP per pixel,
// sits AT the bf16 ridge, and there the two-team kernel is bound by LDS, not HBM: with one accumulator per wave every MFMA is fed by
// two ds_read_b128 (1 KB each; four waves x 2 KB per 32-cycle instruction = twice what the LDS array delivers), and the input is read
// once per 32-cout group.  Here the REGISTER tile is what is organised:
//   * workgroup = 8 x 32 pixels x 128 couts, 8 waves.  Waves 0-3 multiply: wave (c, p) owns cout tiles 2c, 2c+1 and pixel rows
//     4p..4p+3 -> 2 x 4 accumulators; per k-step 4 pixel fragments from LDS + 2 weight fragments feed 8 MFMAs (0.5 LDS reads per
//     instruction instead of 2);
//   * the weight fragments never touch LDS (295 KB per layer would not fit): a wave streams ITS two cout tiles from L2 into a ring
//     of D k-steps of registers, refilled into the slot the just-issued MFMAs consumed -- the stream is cyclic over the tile's
//     9 x Cin/16 k-steps and tile independent, so it never drains between tiles;
//   * waves 4-7 only stage: halo of the next (tile, 64-channel chunk) from HBM/L2 through registers (conversion + zero padding) into
//     the other LDS buffer.  Their long-latency loads live in their own vmcnt queues -- in ONE wave they would sit between the ring
//     loads and every ring wait would also wait for HBM.
// One barrier per (tile, chunk) unit, met by all eight waves.  Results: same products as the kernels above, accumulated chunk-major.
template <int CIN, bool POOL, int IO, bool PRO = false>      // IO: bit 0 = bf16 input, bit 1 = bf16 output; PRO: BatchNorm + LeakyReLU of the input in the staging pass
__global__ __launch_bounds__(512) void conv3x3_bf16_mm_kernel(const B16Args a) {
  constexpr bool IN16 = (IO & 1) != 0, OUT16 = (IO & 2) != 0;
  static_assert(!PRO || !IN16, "the input prologue works on fp32 activations");
  constexpr int KC = CIN / 64, KSG = CIN / 16, KT = 36 * KC;
  constexpr int TH = 8, TW = 32, HR = TH + 2, HC = TW + 2, NPIX = HR * HC;
  constexpr int PS = 64 * 2 + 16;
  constexpr int XBYTES = NPIX * PS;
  constexpr int D = 6;                                  // ring depth in k-steps (36 % D == 0: static slots under the unrolled loop)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* Bs = reinterpret_cast<float*>(smem + 2 * XBYTES);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ngroups = a.Cout / 128;
  const int group = blockIdx.x % ngroups;
  const int s0 = blockIdx.x / ngroups, step = gridDim.x / ngroups;
  const int nsp = a.B * a.tiles_y * a.tiles_x;
  if (s0 >= nsp) return;                                // (uniform over the workgroup)
  const int ntile = (nsp - s0 + step - 1) / step;
  const int nunits = ntile * KC;
  struct Tile { int b, oy0, ox0; };
  auto decode = [&](int st) __attribute__((always_inline)) {
    Tile t;
    t.ox0 = (st % a.tiles_x) * TW; st /= a.tiles_x;
    t.oy0 = (st % a.tiles_y) * TH;
    t.b = st / a.tiles_y;
    return t;
  };
  if (tid < 128) Bs[tid] = (a.bias ? a.bias : b16_zero16)[(group * 128 + tid) * a.bias_mul];

  if (wave >= 4) {
    // ------------------------------------------------------------------------------------------------ staging waves
    const int ht = tid - 256;
    constexpr int PPP = IN16 ? 8 : 16;                  // 16-byte pieces per halo pixel of one unit
    constexpr int NPIECE = NPIX * PPP;
    constexpr int NLOAD = (NPIECE + 255) / 256;
    constexpr int ES = IN16 ? 2 : 4;
    f32x4 pv[NLOAD];
    // PRO: a staging thread's 16-byte piece is the same channel quad of every pixel it handles (256 % PPP == 0): the affine of the unit's
    // 64-channel chunk sits in registers.  The same fused multiply-add and the same select as cmr_affine_act_f32: the operand is bit for
    // bit what that pass would have stored
    f32x4 psc0, psh0, psc1, psh1;                       // (named registers: an indexed pair became a stack object)
    if constexpr (PRO) {
      psc0 = *reinterpret_cast<const f32x4*>(a.in_scale + 4 * (ht % PPP));
      psh0 = *reinterpret_cast<const f32x4*>(a.in_shift + 4 * (ht % PPP));
      psc1 = *reinterpret_cast<const f32x4*>(a.in_scale + (KC - 1) * 64 + 4 * (ht % PPP));
      psh1 = *reinterpret_cast<const f32x4*>(a.in_shift + (KC - 1) * 64 + 4 * (ht % PPP));
    }
    auto issue_loads = [&](int u) __attribute__((always_inline)) {
      const Tile t = decode(s0 + (u / KC) * step);
      const int kc = u % KC;
      const unsigned char* xb = reinterpret_cast<const unsigned char*>(a.x) + ((int64_t)t.b * a.H * a.W * CIN + kc * 64) * ES;
#pragma unroll
      for (int i = 0; i < NLOAD; ++i) {
        int e = ht + 256 * i;
        e = e < NPIECE ? e : NPIECE - 1;
        const int p = e / PPP, c = e % PPP;
        int iy = t.oy0 - 1 + p / HC, ix = t.ox0 - 1 + p % HC;
        iy = iy < 0 ? 0 : (iy >= a.H ? a.H - 1 : iy);
        ix = ix < 0 ? 0 : (ix >= a.W ? a.W - 1 : ix);
        pv[i] = *reinterpret_cast<const f32x4*>(xb + (unsigned)((iy * a.W + ix) * CIN * ES + 16 * c));
      }
    };
    auto store_lds = [&](int u) __attribute__((always_inline)) {
      const Tile t = decode(s0 + (u / KC) * step);
      unsigned char* Xs = smem + (u & 1) * XBYTES;
#pragma unroll
      for (int i = 0; i < NLOAD; ++i) {
        const int e = ht + 256 * i;
        if (e < NPIECE) {
          const int p = e / PPP, c = e % PPP;
          const int iy = t.oy0 - 1 + p / HC, ix = t.ox0 - 1 + p % HC;
          const bool inb = ((unsigned)iy < (unsigned)a.H) & ((unsigned)ix < (unsigned)a.W);
          const unsigned keep = inb ? 0xffffffffu : 0u;
          if constexpr (IN16) {
            uint4 w = __builtin_bit_cast(uint4, pv[i]);
            w.x &= keep; w.y &= keep; w.z &= keep; w.w &= keep;
            *reinterpret_cast<uint4*>(Xs + p * PS + c * 16) = w;
          } else {
            f32x4 v = pv[i];
            if constexpr (PRO) {
              const bool hi = KC == 2 && (u % KC) != 0;                       // (uniform) the unit's 64-channel chunk
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                v[q] = __builtin_fmaf(v[q], hi ? psc1[q] : psc0[q], hi ? psh1[q] : psh0[q]);
                v[q] = fmaxf(v[q], v[q] * a.in_slope);              // = the select for 0 <= slope <= 1 (the entry point checks), one instruction less
              }
            }
            uint2 w = b16_pack4(v);
            w.x &= keep; w.y &= keep;
            *reinterpret_cast<uint2*>(Xs + p * PS + c * 8) = w;
          }
        }
      }
    };
    issue_loads(0);
    store_lds(0);
    issue_loads(nunits > 1 ? 1 : 0);
    __syncthreads();                                    // unit 0 staged
    for (int u = 0; u < nunits; ++u) {
      // the multiplying waves read buffer u & 1 now; the other one was read during unit u - 1 and is free.  Past the end: a harmless
      // re-stage / re-read of the last unit instead of branches around the loads
      const int un = u + 1 < nunits ? u + 1 : u;
      if (u + 1 < nunits) store_lds(un);
      issue_loads(u + 2 < nunits ? u + 2 : un);
      __syncthreads();
    }
    return;
  }

  // -------------------------------------------------------------------------------------------------- multiplying waves
  const int h = lane >> 5, l31 = lane & 31;
  const int cw = wave & 1, pw = wave >> 1;
  // wave-uniform fragment bases (scalar registers) + ONE per-lane byte offset that is re-materialised per unit (b16_fresh): every ring
  // load is `v_add_u32 voff, lane16, literal ; global_load_dwordx4 v, voff, s[base]`.  Written as per-lane 64-bit addresses the 144
  // loop-invariant address pairs are hoisted out of the tile loop and spilled (250 spilled registers in the first version).
  constexpr int WNT = CIN == 64 ? 2 : 1;                // cout tiles per group in the fragment layout (_pack.conv_bf16_frags)
  const unsigned char* wp[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int T = group * 4 + 2 * cw + nt;              // 32-cout tile of the layer; fragments are stored [T / WNT][tap][ks][T % WNT][lane][8]
    wp[nt] = static_cast<const unsigned char*>(a.wfrag) + ((size_t)(T / WNT) * 9 * KSG * WNT + (T % WNT)) * 1024;
  }
  unsigned lane16 = lane * 16;
  bf16x8 ring[D][2];
  auto load_a = [&](int j, bf16x8 (&dst)[2]) __attribute__((always_inline)) {     // stream position j of the tile: (chunk, tap, ks)
    const int r = j % 36;
    const unsigned off = (unsigned)(((r / 4) * KSG + (j / 36) * 4 + (r % 4)) * WNT * 1024);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) dst[nt] = *reinterpret_cast<const bf16x8*>(wp[nt] + (size_t)(lane16 + off));
  };
#pragma unroll
  for (int d = 0; d < D; ++d) load_a(d, ring[d]);
  const int lanebase = ((4 * pw) * HC + l31) * PS + h * 16;
  const int Ho = a.H, Wo = a.W;
  f32x16 acc[2][4];
  auto multiply = [&](auto CH, int buf) __attribute__((always_inline)) {
    constexpr int ch = decltype(CH)::value;
    asm volatile("" : "+v"(lane16));                    // see load_a
    const unsigned char* xs = smem + buf * XBYTES + lanebase;
    bf16x8 bv[2][4];
    auto read_b = [&](int jj, bf16x8 (&dst)[4]) __attribute__((always_inline)) {
      const int tap = jj / 4, ks = jj % 4, ky = tap / 3, kx = tap % 3;
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) dst[nb] = *reinterpret_cast<const bf16x8*>(xs + ((nb + ky) * HC + kx) * PS + ks * 32);
    };
    read_b(0, bv[0]);
#pragma unroll
    for (int jj = 0; jj < 36; ++jj) {
      if (jj + 1 < 36) read_b(jj + 1, bv[(jj + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);                // the next k-step's pixel fragments are requested BEFORE this k-step's 8 MFMAs, not under the last one
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
          acc[nt][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[jj % D][nt], bv[jj & 1][nb], acc[nt][nb], 0, 0, 0);
      load_a((ch * 36 + jj + D) % KT, ring[jj % D]);
      __builtin_amdgcn_sched_barrier(0);                // keeps the 36 k-steps in program order (the scheduler would pull the ring loads up and spill)
    }
  };
  auto epilogue = [&](const Tile& t) __attribute__((always_inline)) {
    // register 4q+e of accumulator (nt, nb) = channel 128 group + 32 (2 cw + nt) + 8q + 4h + e of pixel (row 4 pw + nb, column l31)
    const int cbase = (2 * cw) * 32 + 4 * h;
    if constexpr (POOL) {
      const int px = (t.ox0 >> 1) + (l31 >> 1);
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        const int py = (t.oy0 >> 1) + 2 * pw + pr;
        f32x4 ov[2][4];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 bq = *reinterpret_cast<const f32x4*>(&Bs[cbase + nt * 32 + q * 8]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float v0 = acc[nt][2 * pr][4 * q + e] + bq[e], v1 = acc[nt][2 * pr + 1][4 * q + e] + bq[e];
              v0 = v0 > 0.f ? v0 : v0 * a.slope;
              v1 = v1 > 0.f ? v1 : v1 * a.slope;
              float v = v0 + v1;
              v += cmr_xor1(v);
              ov[nt][q][e] = 0.25f * v;
            }
          }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int q = 0; q < 4; ++q) cmr_pin(ov[nt][q]);
        if ((l31 & 1) == 0 && py < (Ho >> 1) && px < (Wo >> 1)) {
          const int64_t o = (((int64_t)t.b * (Ho >> 1) + py) * (Wo >> 1) + px) * a.Cout + group * 128 + cbase;
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              if constexpr (OUT16) *reinterpret_cast<uint2*>(reinterpret_cast<__bf16*>(a.y) + o + nt * 32 + q * 8) = b16_pack4(ov[nt][q]);
              else *reinterpret_cast<f32x4*>(a.y + o + nt * 32 + q * 8) = ov[nt][q];
            }
        }
      }
    } else {
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        const int oy = t.oy0 + 4 * pw + nb, ox = t.ox0 + l31;
        f32x4 ov[2][4];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 bq = *reinterpret_cast<const f32x4*>(&Bs[cbase + nt * 32 + q * 8]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float u = acc[nt][nb][4 * q + e] + bq[e];
              ov[nt][q][e] = u > 0.f ? u : u * a.slope;
            }
          }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int q = 0; q < 4; ++q) cmr_pin(ov[nt][q]);
        if (oy < Ho && ox < Wo) {
          const int64_t o = (((int64_t)t.b * Ho + oy) * Wo + ox) * a.Cout + group * 128 + cbase;
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              if constexpr (OUT16) *reinterpret_cast<uint2*>(reinterpret_cast<__bf16*>(a.y) + o + nt * 32 + q * 8) = b16_pack4(ov[nt][q]);
              else *reinterpret_cast<f32x4*>(a.y + o + nt * 32 + q * 8) = ov[nt][q];
            }
        }
      }
    }
  };
  __syncthreads();                                      // unit 0 staged (and the bias)
  for (int i = 0; i < ntile; ++i) {
    const Tile t = decode(s0 + i * step);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][nb][r] = 0.f;
    if constexpr (KC == 2) {
      multiply(std::integral_constant<int, 0>{}, 0);
      __syncthreads();
      multiply(std::integral_constant<int, 1>{}, 1);
    } else {
      multiply(std::integral_constant<int, 0>{}, i & 1);
    }
    epilogue(t);
    __syncthreads();
  }
}

template <int CIN, bool POOL, int IO, bool PRO = false>
int launch_mm_p(B16Args a, hipStream_t stream) {
  constexpr int smem = 2 * 10 * 34 * 144 + 512;
  static CmrSmemCache granted{};
  if (cmr_grant_smem(reinterpret_cast<const void*>(conv3x3_bf16_mm_kernel<CIN, POOL, IO, PRO>), smem, granted) != CMR_OK) return CMR_ELAUNCH;
  a.tiles_x = (a.W + 31) / 32;
  a.tiles_y = (a.H + 7) / 8;
  a.bias_mul = a.bias ? 1 : 0;
  const int ngroups = a.Cout / 128;
  const int64_t nsp = (int64_t)a.B * a.tiles_x * a.tiles_y;
  const int cus = a.cu_budget >= 8 && a.cu_budget < 256 ? a.cu_budget - a.cu_budget % 8 : 256;
  int per_group = cus / ngroups;                        // one persistent workgroup per CU
  if (per_group < 1) per_group = 1;
  if (per_group > nsp) per_group = (int)nsp;
  hipLaunchKernelGGL((conv3x3_bf16_mm_kernel<CIN, POOL, IO, PRO>), dim3(ngroups * per_group), dim3(512), smem, stream, a);
  return cmr_launch_status();
}

template <int CIN>
int launch_mm(const B16Args& a, hipStream_t stream) {
  const int io = (a.x_bf16 ? 1 : 0) | (a.y_bf16 ? 2 : 0);
  if (a.pool == 2) {
    switch (io) {
      case 0: return launch_mm_p<CIN, true, 0>(a, stream);
      case 1: return launch_mm_p<CIN, true, 1>(a, stream);
      case 2: return launch_mm_p<CIN, true, 2>(a, stream);
      default: return launch_mm_p<CIN, true, 3>(a, stream);
    }
  }
  switch (io) {
    case 0: return launch_mm_p<CIN, false, 0>(a, stream);
    case 1: return launch_mm_p<CIN, false, 1>(a, stream);
    case 2: return launch_mm_p<CIN, false, 2>(a, stream);
    default: return launch_mm_p<CIN, false, 3>(a, stream);
  }
}

}  // namespace

#ifdef CMR_AB_SWITCHES
static int CMR_B16_MM = 1;             // matrix-class kernel for the 128-cout layers (cmr_set_conv_bf16_variant: A/B measurements, libcmr_hip_ab.so only)
static int CMR_B16_MM_MIN_TILES = 128;
extern "C" int cmr_set_conv_bf16_variant(int matrix_class, int min_tiles) {
  CMR_REQUIRE(matrix_class == 0 || matrix_class == 1);
  CMR_B16_MM = matrix_class;
  if (min_tiles > 0) CMR_B16_MM_MIN_TILES = min_tiles;
  return CMR_OK;
}
#else
static constexpr int CMR_B16_MM = 1, CMR_B16_MM_MIN_TILES = 128;
#endif

static int conv3x3_bf16_dispatch(const void* x, int x_bf16, int B, int H, int W, int Cin, const void* wfrag, int nt, const float* bias,
                                 const void* res, int res_bf16, const float* post, void* y, int y_bf16, int Cout, int stride, float slope, int pool,
                                 int cu_budget, hipStream_t stream) {
  CMR_REQUIRE(x && wfrag && y && B > 0 && H > 0 && W > 0 && Cout > 0 && (stride == 1 || stride == 2));
  CMR_REQUIRE(cmr_aligned16(x) && cmr_aligned16(wfrag) && cmr_aligned16(y) && (!bias || cmr_aligned16(bias)) && (!res || cmr_aligned16(res)) &&
              (!post || cmr_aligned16(post)));
  CMR_REQUIRE(pool == 1 || (pool == 2 && !res && !post && H % 2 == 0 && W % 2 == 0));
  CMR_REQUIRE((int64_t)B * H * W * (Cin > Cout ? Cin : Cout) < 0x7fffffff);
  B16Args a{static_cast<const float*>(x), B, H, W, wfrag, bias, static_cast<const float*>(res), post, static_cast<float*>(y), Cout, slope, pool,
            0, 0, 0, 0, x_bf16, y_bf16, res && res_bf16 ? 1 : 0, nt, cu_budget};
  if (stride == 2) {
    if (pool != 1) return CMR_EINVAL;
    if (a.res_bf16) return CMR_EUNSUPPORTED;
#ifndef B16_S2_ONE_TEAM
    if (Cin == 64 && nt == 2 && Cout % 64 == 0 && !post) {
      a.res_mul = a.res ? 1 : 0; a.bias_mul = a.bias ? 1 : 0;
      if (x_bf16) return y_bf16 ? launch_tt_p<64, 2, false, false, 3, 2>(a, stream) : launch_tt_p<64, 2, false, false, 1, 2>(a, stream);
      return y_bf16 ? launch_tt_p<64, 2, false, false, 2, 2>(a, stream) : launch_tt_p<64, 2, false, false, 0, 2>(a, stream);
    }
#endif
    if (x_bf16) return CMR_EUNSUPPORTED;               // the one-team strided instances read fp32 activations
    if (Cin == 64 && nt == 2 && Cout % 64 == 0) return launch_b16<64, 2, 16, 2>(a, stream);
    if (Cin == 64 && nt == 1 && Cout % 32 == 0) return launch_b16<64, 1, 16, 2>(a, stream);
    return CMR_EUNSUPPORTED;
  }
  // 128-cout layers without residual / table on maps with enough 8x32 tiles: the matrix-class kernel (register-tiled, weights streamed)
  if (CMR_B16_MM && Cout % 128 == 0 && !res && !post && (Cin == 64 || Cin == 128) && nt == (Cin == 64 ? 2 : 1) &&
      (int64_t)B * ((H + 7) / 8) * ((W + 31) / 32) * (Cout / 128) >= CMR_B16_MM_MIN_TILES) {
    return Cin == 64 ? launch_mm<64>(a, stream) : launch_mm<128>(a, stream);
  }
  if (Cin == 64 && nt == 2 && Cout % 64 == 0) return launch_tt<64, 2>(a, stream);
  if (Cin == 64 && nt == 1 && Cout % 32 == 0) return launch_tt<64, 1>(a, stream);
  if (Cin == 128 && nt == 1 && Cout % 32 == 0) return launch_tt<128, 1>(a, stream);
  return CMR_EUNSUPPORTED;
}

extern "C" int cmr_conv3x3_bf16_nhwc_f32(const float* x, int B, int H, int W, int Cin, const void* wfrag, int nt, const float* bias,
                                         const float* res, const float* post, float* y, int Cout, int stride, float slope, int pool,
                                         int cu_budget, hipStream_t stream) {
  return conv3x3_bf16_dispatch(x, 0, B, H, W, Cin, wfrag, nt, bias, res, 0, post, y, 0, Cout, stride, slope, pool, cu_budget, stream);
}

// Training forward of [BatchNorm -> LeakyReLU(in_slope) -> 3x3 conv (+ bias, LeakyReLU(slope))] without the activated map in memory: x is the
// BatchNorm INPUT (fp32 NHWC), in_scale / in_shift [Cin] its folded affine (stat[2], stat[3] of cmr_bn_stats_f32); the operand
// lrelu(x * in_scale + in_shift) is formed while the halo is converted to bf16 (padding stays zero) -- bit for bit the operand the
// convolution reads from the map cmr_affine_act_f32 would have written.  Served by the matrix-class kernel only (Cin = 128, Cout % 128 == 0,
// maps of >= 128 8x32-pixel tiles x Cout / 128); CMR_EUNSUPPORTED otherwise: the caller materialises the activation.
extern "C" int cmr_conv3x3_bf16_pro_nhwc_f32(const float* x, const float* in_scale, const float* in_shift, float in_slope, int B, int H, int W,
                                             int Cin, const void* wfrag, int nt, const float* bias, float* y, int Cout, float slope, int cu_budget,
                                             hipStream_t stream) {
  CMR_REQUIRE(x && in_scale && in_shift && wfrag && y && B > 0 && H > 0 && W > 0 && Cout > 0);
  CMR_REQUIRE(cmr_aligned16(x) && cmr_aligned16(in_scale) && cmr_aligned16(in_shift) && cmr_aligned16(wfrag) && cmr_aligned16(y) &&
              (!bias || cmr_aligned16(bias)));
  CMR_REQUIRE((int64_t)B * H * W * (Cin > Cout ? Cin : Cout) < 0x7fffffff);
  if (!(in_slope >= 0.f && in_slope <= 1.f)) return CMR_EUNSUPPORTED;
  if (!(CMR_B16_MM && Cin == 128 && nt == 1 && Cout % 128 == 0 &&
        (int64_t)B * ((H + 7) / 8) * ((W + 31) / 32) * (Cout / 128) >= CMR_B16_MM_MIN_TILES))
    return CMR_EUNSUPPORTED;
  B16Args a{x, B, H, W, wfrag, bias, nullptr, nullptr, y, Cout, slope, 1, 0, 0, 0, 0, 0, 0, 0, nt, cu_budget, in_scale, in_shift, in_slope};
  return launch_mm_p<128, false, 0, true>(a, stream);
}

extern "C" int cmr_conv3x3_bf16io_nhwc(const void* x, int x_bf16, int B, int H, int W, int Cin, const void* wfrag, int nt, const float* bias,
                                       const void* res, int res_bf16, const float* post, void* y, int y_bf16, int Cout, int stride, float slope,
                                       int pool, int cu_budget, hipStream_t stream) {
  return conv3x3_bf16_dispatch(x, x_bf16, B, H, W, Cin, wfrag, nt, bias, res, res_bf16, post, y, y_bf16, Cout, stride, slope, pool, cu_budget, stream);
}
