// Probe: does an LDS-DMA load (global_load_lds_dwordx4) reach LDS offsets above 64 KB on gfx950 (160 KB of LDS per workgroup)?
// One workgroup; wave 0 copies 1 KB from global memory to LDS byte offset `off` (builtin form and the inline-asm form with M0 written in the
// same statement), every thread then reads the block back with ds_read and compares.   hipcc --offload-arch=gfx950 tools/glds_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void probe(const float* __restrict__ src, int off_bytes, int use_asm, int* __restrict__ bad) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 256) reinterpret_cast<float*>(smem)[i] = -1.f;
  __syncthreads();
  if (threadIdx.x < 64) {
    const float* g = src + threadIdx.x * 4;
    if (use_asm) {
      unsigned keep;
      const unsigned lds_dst = (unsigned)(size_t)(smem + off_bytes);
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(lds_dst) : "memory");
    } else {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)(smem + off_bytes), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  int nbad = 0;
  for (int i = threadIdx.x; i < 256; i += 256) nbad += reinterpret_cast<float*>(smem + off_bytes)[i] != src[i];
  // anything else disturbed?
  for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 256) {
    const int b = i * 4;
    if (b >= off_bytes && b < off_bytes + 1024) continue;
    nbad += reinterpret_cast<float*>(smem)[i] != -1.f ? 1000 : 0;
  }
  if (nbad) atomicAdd(bad, nbad);
}

int main() {
  float* src; int* bad;
  hipMalloc(&src, 1024); hipMalloc(&bad, 4);
  std::vector<float> h(256);
  for (int i = 0; i < 256; ++i) h[i] = 1.f + i;
  hipMemcpy(src, h.data(), 1024, hipMemcpyHostToDevice);
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int use_asm = 0; use_asm < 2; ++use_asm)
    for (int off : {0, 32768, 65536 - 1024, 65536, 98304, 131072, 160 * 1024 - 1024}) {
      int z = 0; hipMemcpy(bad, &z, 4, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(probe, dim3(1), dim3(256), 160 * 1024, 0, src, off, use_asm, bad);
      hipError_t e = hipDeviceSynchronize();
      hipMemcpy(&z, bad, 4, hipMemcpyDeviceToHost);
      printf("%s  LDS offset %6d: %s (mismatch count %d, %s)\n", use_asm ? "asm    " : "builtin", off, z == 0 ? "ok" : "WRONG", z, hipGetErrorString(e));
    }
  return 0;
}
