"""Streaming limits of the row-map access patterns on MI355X: copy of a [rows][64] fp32 map (a) lane = row, 16-byte pieces (the MFMA operand
pattern of linear_ws_kernel / its float4 epilogue), (b) fully coalesced float4 (lane i of a wave reads bytes 16 i .. of a 1 KB run),
(c) like (a) for the loads, coalesced stores through an LDS transposition.  Built with torch's cpp_extension-free route: hipcc -> .so -> ctypes."""
import ctypes, os, subprocess, sys, tempfile
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kbench import timeit

SRC = r'''
#include <hip/hip_runtime.h>
typedef float f4 __attribute__((ext_vector_type(4)));
extern "C" __global__ __launch_bounds__(256) void copy_rowlane(const float* __restrict__ x, float* __restrict__ y, unsigned rows) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, l31 = lane & 31;
  const unsigned ntiles = (rows + 31) / 32, stride = gridDim.x * 4;
  for (unsigned tile = blockIdx.x * 4 + wave; tile < ntiles; tile += stride) {
    unsigned row = tile * 32 + l31; if (row >= rows) row = rows - 1;
    const float* xp = x + (size_t)row * 64 + 8 * h;
    f4 v[8];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { v[2*ks] = *(const f4*)(xp + 16*ks); v[2*ks+1] = *(const f4*)(xp + 16*ks + 4); }
    float* yp = y + (size_t)row * 64 + 4 * h;
#pragma unroll
    for (int q = 0; q < 8; ++q) *(f4*)(yp + 8 * q) = v[q];
  }
}
extern "C" __global__ __launch_bounds__(256) void copy_coalesced(const float* __restrict__ x, float* __restrict__ y, unsigned rows) {
  const size_t n4 = (size_t)rows * 16;
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride * 4) {
    f4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const size_t j = i + u * stride; v[u] = j < n4 ? ((const f4*)x)[j] : f4{0,0,0,0}; }
#pragma unroll
    for (int u = 0; u < 4; ++u) { const size_t j = i + u * stride; if (j < n4) ((f4*)y)[j] = v[u]; }
  }
}
extern "C" void launch(int which, const float* x, float* y, unsigned rows, int grid, hipStream_t s) {
  if (which == 0) hipLaunchKernelGGL(copy_rowlane, dim3(grid), dim3(256), 0, s, x, y, rows);
  else hipLaunchKernelGGL(copy_coalesced, dim3(grid), dim3(256), 0, s, x, y, rows);
}
'''

def main():
    d = tempfile.mkdtemp()
    open(os.path.join(d, "s.hip"), "w").write(SRC)
    so = os.path.join(d, "s.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(d, "s.hip"), "-o", so])
    lib = ctypes.CDLL(so)
    lib.launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint, ctypes.c_int, ctypes.c_void_p]
    for rows in (214016, 524288):
        x = torch.randn(rows, 64, device="cuda"); y = torch.empty_like(x)
        st = lambda: torch.cuda.current_stream().cuda_stream
        for which, name in ((0, "lane = row, 16-byte pieces"), (1, "coalesced float4")):
            for grid in (512, 1024, 2048):
                t = timeit(lambda: lib.launch(which, x.data_ptr(), y.data_ptr(), rows, grid, st()), 20)
                print("rows %6d %-28s grid %4d : %6.1f us = %5.2f TB/s" % (rows, name, grid, t, 2 * rows * 256 / t / 1e6))
        t = timeit(lambda: y.copy_(x), 20)
        print("rows %6d torch copy_                             : %6.1f us = %5.2f TB/s" % (rows, t, 2 * rows * 256 / t / 1e6))

main()
