// Sustained fp32 / bf16 matrix rate of the device this runs on: a grid of waves that do nothing but dependent-free chains of
// v_mfma_f32_32x32x2_f32 (and v_mfma_f32_32x32x16_bf16) for a given number of iterations; HIP events; several durations so that clock
// throttling under sustained load shows.  Development tool (docs/MEASUREMENT_ROUNDS_1_5.md 5d quotes it next to the nominal peaks of MI355X_MICROARCH.md):
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o gpurun_out/mfma_peak && gpurun_out/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void mfma_f32_kernel(float* out, int iters) {
  f32x16 acc[8];
  for (int t = 0; t < 8; ++t)
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  // eight different operand pairs with pseudo-random mantissas per lane (constant operands toggle few bits and draw less power)
  float a[8], b[8];
  unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
  for (int t = 0; t < 8; ++t) {
    h = h * 1664525u + 1013904223u; a[t] = __uint_as_float(0x3f000000u | (h >> 9)) - 0.75f;
    h = h * 1664525u + 1013904223u; b[t] = __uint_as_float(0x3f000000u | (h >> 9)) - 0.75f;
  }
  for (int i = 0; i < iters; i += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u)                    // (static register indices: the pairing rotates without moves)
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[(t + u) & 7], acc[t], 0, 0, 0);
  }
  float s = 0.f;
  for (int t = 0; t < 8; ++t) s += acc[t][0];
  if (s == 12345.678f) out[0] = s;          // never true: keeps the chain alive
}

__global__ __launch_bounds__(256) void mfma_bf16_kernel(float* out, int iters) {
  f32x16 acc[8];
  for (int t = 0; t < 8; ++t)
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(threadIdx.x * 1e-3f + e); b[e] = (__bf16)(1.0f + e); }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
  }
  float s = 0.f;
  for (int t = 0; t < 8; ++t) s += acc[t][0];
  if (s == 12345.678f) out[0] = s;
}

int main() {
  float* out;
  hipMalloc(&out, 4);
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount, wgs = cus * 2;              // 2 workgroups x 4 waves per CU = 2 waves per SIMD
  printf("%s: %d CUs, clock %d MHz (reported)\n", p.name, cus, p.clockRate / 1000);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int kind = 0; kind < 2; ++kind) {
    for (int iters : {2000, 20000, 200000}) {
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (kind == 0) hipLaunchKernelGGL(mfma_f32_kernel, dim3(wgs), dim3(256), 0, 0, out, iters);
        else hipLaunchKernelGGL(mfma_bf16_kernel, dim3(wgs), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop_per_mfma = kind == 0 ? 2.0 * 32 * 32 * 2 : 2.0 * 32 * 32 * 16;
        const double flops = (double)wgs * 4 * iters * 8 * flop_per_mfma;
        if (rep == 2)
          printf("%s  %7d iterations x 8 chains, %4d workgroups: %8.3f ms  %8.1f TFLOP/s\n", kind == 0 ? "v_mfma_f32_32x32x2_f32  " : "v_mfma_f32_32x32x16_bf16",
                 iters, wgs, ms, flops / (ms * 1e-3) / 1e12);
      }
    }
  }
  return 0;
}
