// Access-pattern microbenchmark: how fast can [C][T] row-tiled copies go on MI355X with the conv kernel's store pattern?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// pattern A: the conv epilogue: block tile = 32 rows x 512 cols, wave w owns cols [w*128, +128); per lane 16 regs x 4 an, dword each
__global__ __launch_bounds__(256) void copy_tile_dword(const float* __restrict__ x, float* __restrict__ y, long long ld, int T, int do_load) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
  const long long n0 = (long long)blockIdx.x * 512; const int r0 = blockIdx.y * 32;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = r0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
#pragma unroll
    for (int an = 0; an < 4; ++an) {
      const long long n = n0 + (wave * 4 + an) * 32 + li;
      if (n < T) { float v = do_load ? x[m * ld + n] : (float)lane; y[m * ld + n] = v + 1.f; }
    }
  }
}
// pattern B: same tile, but each wave instruction covers 1 row x 64 lanes x float4 (1 KB contiguous)
__global__ __launch_bounds__(256) void copy_tile_vec4(const float* __restrict__ x, float* __restrict__ y, long long ld, int T, int do_load) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long n0 = (long long)blockIdx.x * 512; const int r0 = blockIdx.y * 32;
#pragma unroll
  for (int rr = 0; rr < 8; ++rr) {
    const int m = r0 + wave * 8 + rr;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const long long n = n0 + h * 256 + lane * 4;
      if (n + 3 < T) {
        float4 v = do_load ? *reinterpret_cast<const float4*>(x + m * ld + n) : make_float4(1, 2, 3, 4);
        v.x += 1.f; *reinterpret_cast<float4*>(y + m * ld + n) = v;
      }
    }
  }
}
__global__ void copy_linear(const float4* __restrict__ x, float4* __restrict__ y, long long n4, int do_load) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; const long long st = (long long)gridDim.x * blockDim.x;
  for (; i < n4; i += st) { float4 v = do_load ? x[i] : make_float4(1, 2, 3, 4); v.x += 1.f; y[i] = v; }
}
int main() {
  const int C = 32, T = 1279200; const long long ld = T; const size_t n = (size_t)C * ld;
  float *x, *y; CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&y, n * 4)); CK(hipMemset(x, 0, n * 4));
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int mode = 0; mode < 6; ++mode) {
    const int do_load = mode < 3;
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(a);
      dim3 g((T + 511) / 512, C / 32);
      if (mode % 3 == 0) hipLaunchKernelGGL(copy_tile_dword, g, dim3(256), 0, 0, x, y, ld, T, do_load);
      else if (mode % 3 == 1) hipLaunchKernelGGL(copy_tile_vec4, g, dim3(256), 0, 0, x, y, ld, T, do_load);
      else hipLaunchKernelGGL(copy_linear, dim3(2048), dim3(256), 0, 0, (const float4*)x, (float4*)y, (long long)(n / 4), do_load);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    const double bytes = (do_load ? 2.0 : 1.0) * n * 4;
    printf("%-16s %-10s %8.1f us  %7.2f TB/s\n", mode % 3 == 0 ? "tile dword" : mode % 3 == 1 ? "tile float4" : "linear float4", do_load ? "copy" : "store-only", best * 1e3, bytes / best / 1e9);
  }
  return 0;
}
