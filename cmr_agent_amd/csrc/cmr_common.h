// Shared device/host helpers for libcmr_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CMR_OK 0
#define CMR_EINVAL -1   // bad argument (shape / alignment / null pointer)
#define CMR_ELAUNCH -2  // hipGetLastError() after the launch was not hipSuccess
#define CMR_EUNSUPPORTED -3  // valid request that this build serves through another entry point (caller falls back)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CMR_REQUIRE(cond) \
  do {                    \
    if (!(cond)) return CMR_EINVAL; \
  } while (0)

static inline int cmr_launch_status() { return hipGetLastError() == hipSuccess ? CMR_OK : CMR_ELAUNCH; }

// Raises a kernel's dynamic-LDS limit (needed above 64 KB), once per (kernel, device): `cache` is a per-kernel array of
// the sizes already granted, indexed by device, so a process that drives several GPUs sets it on each of them.
constexpr int CMR_MAX_DEVICES = 16;
struct CmrSmemCache { size_t granted[CMR_MAX_DEVICES]; };
static inline int cmr_grant_smem(const void* kernel, size_t bytes, CmrSmemCache& cache) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return CMR_ELAUNCH;
  dev = dev < 0 || dev >= CMR_MAX_DEVICES ? 0 : dev;
  if (bytes <= 64 * 1024 || bytes <= cache.granted[dev]) return CMR_OK;
  if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return CMR_ELAUNCH;
  cache.granted[dev] = bytes;
  return CMR_OK;
}

static inline bool cmr_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// activation codes shared by the GEMM / conv epilogues
enum { CMR_ACT_NONE = 0, CMR_ACT_RELU = 1, CMR_ACT_LRELU = 2, CMR_ACT_GELU = 3, CMR_ACT_ELU1 = 4 };

__device__ __forceinline__ float cmr_act(float v, int act, float p) {
  switch (act) {
    case CMR_ACT_RELU: return v > 0.f ? v : 0.f;
    case CMR_ACT_LRELU: return v > 0.f ? v : v * p;
    case CMR_ACT_GELU: return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
    case CMR_ACT_ELU1: return v > 0.f ? v + 1.f : expf(v);  // elu(v) + 1
    default: return v;
  }
}

// 32x32x2 fp32 MFMA: D[i][j] += sum_k A[i][k] B[k][j];  lane l supplies A[i=l&31][k=l>>5] and
// B[k=l>>5][j=l&31]; D register r of lane l is D[row=(r&3)+8*(r>>2)+4*(l>>5)][col=l&31].
// Pins a value in registers at this point of the program.  Epilogues compute every output first, pin it, and
// only then enter the predicated stores: hipcc otherwise sinks "acc + loaded operand" into each predicated
// block, where its waitcnt bookkeeping degrades to s_waitcnt vmcnt(0) -- every store then waits for the
// previous store (stores count in vmcnt on gfx9) and for the next tile's prefetch.
__device__ __forceinline__ void cmr_pin(f32x4& v) { asm volatile("" : "+v"(v)); }

__device__ __forceinline__ f32x16 cmr_mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// The value the partner lane (lane ^ 32) holds: v_permlane32_swap (gfx950) + a select, on the VALU -- __shfl_xor(v, 32) is a
// ds_bpermute_b32 round trip through LDS with an address and a wait (36 of them per tile in the linear-attention query layer).
// For 1-D workgroups whose size is a multiple of 64.
__device__ __forceinline__ float cmr_xhalf(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);      // r[0] = [lo | lo], r[1] = [hi | hi]
  return __builtin_bit_cast(float, (threadIdx.x & 32) ? r[0] : r[1]);
}

// Maximum over the 32 lanes that share lane >> 5 (the 32 rows of a tile in the MFMA result layout), on the DPP data path: quad
// permutes, the two row mirrors and row_bcast:15 into rows 1 / 3 -- five VALU instructions per value and nothing through LDS
// (__shfl_xor compiles to ds_bpermute_b32: 160 LDS round trips + their waits per tile).  The result is valid in lanes 16..31 of
// each half.  max is exact and order independent: the same bits as before.
// (Done on the order-preserving integer image of the float -- k = bits ^ ((bits >> 31) & 0x7fffffff), an involution -- because a
// float max behind a DPP move is canonicalised first (two extra v_max per step) and is not folded into v_max_f32_dpp.)
template <int CTRL>
__device__ __forceinline__ int cmr_dpp(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true); }      // every lane has a source
template <int CTRL, int ROWS>
__device__ __forceinline__ int cmr_dpp_rows(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, ROWS, 0xf, false); }
__device__ __forceinline__ int cmr_imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ float cmr_rowmax32(float x) {
  int m = __builtin_bit_cast(int, x);
  m ^= (m >> 31) & 0x7fffffff;
  m = cmr_imax(m, cmr_dpp<0xB1>(m));          // quad_perm:[1,0,3,2]
  m = cmr_imax(m, cmr_dpp<0x4E>(m));          // quad_perm:[2,3,0,1]
  m = cmr_imax(m, cmr_dpp<0x141>(m));         // row_half_mirror
  m = cmr_imax(m, cmr_dpp<0x140>(m));         // row_mirror
  m = cmr_imax(m, cmr_dpp_rows<0x142, 0xa>(m));    // row_bcast:15 -> rows 1 and 3 (rows 0 and 2 keep their own value)
  m ^= (m >> 31) & 0x7fffffff;
  return __builtin_bit_cast(float, m);
}

// Partner values for lane ^ 1 (DPP quad permute: folds into the consuming VALU instruction) and lane ^ 16 (v_permlane16_swap + select),
// instead of ds_bpermute_b32 round trips.  1-D workgroups, size a multiple of 64.
__device__ __forceinline__ float cmr_xor1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
}
__device__ __forceinline__ float cmr_xor16(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);      // r[0] = rows [0 0 2 2], r[1] = rows [1 1 3 3]
  return __builtin_bit_cast(float, (threadIdx.x & 16) ? r[0] : r[1]);
}

__device__ __forceinline__ int cmr_mfma_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// Dropout mask, counter based (no state, no stored mask): element `idx` of dropout site `site` in the step whose seed is `seed` is KEPT
// iff the low 32 bits of a 64-bit mix (murmur3 finaliser) of the three reach thr = p * 2^32.  Forward and backward call it with the same
// arguments, so the backward pass regenerates the forward mask.  (seed, site) go through the finaliser FIRST and the element index is
// mixed into that key: with `(idx + site * G) ^ seed` in one mix, the mask of seed s + 1 was the mask of seed s with neighbouring
// elements swapped ((A ^ (s + 1)) == ((A ^ 1) ^ s) for even s), i.e. consecutive steps -- and ranks, whose seeds differ by a constant --
// dropped almost the same pattern.  The key is wave-uniform, so the second mix is the only per-element cost.
__device__ __forceinline__ uint64_t cmr_mix64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull;
  x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull;
  x ^= x >> 33;
  return x;
}
__device__ __forceinline__ bool cmr_keep(uint64_t seed, uint64_t site, uint64_t idx, uint32_t thr) {
  const uint64_t key = cmr_mix64(seed + site * 0x9E3779B97F4A7C15ull);
  return (uint32_t)cmr_mix64(key ^ idx) >= thr;
}
static inline uint32_t cmr_drop_threshold(float p) {
  const double t = (double)p * 4294967296.0;
  return t <= 0.0 ? 0u : (t >= 4294967295.0 ? 4294967295u : (uint32_t)t);
}

