// LDS-DMA throughput microbenchmark: how many bytes per cycle and CU does global_load_lds (16 B per lane, 1 KiB per wave-instruction) deliver
// from L2-resident data, by pieces in flight per wave and workgroups per CU - against plain global_load_dwordx4 into registers.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/dmabench.hip -o /tmp/dmabench && /tmp/dmabench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int D> __device__ __forceinline__ void wait_le() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D) : "memory"); }

// every wave: ITER pieces of 1 KiB, D in flight; source = region of `span` bytes per workgroup (walked cyclically), LDS ring of D + 1 KiB per wave
template <int D, bool DMA>
__global__ __launch_bounds__(256) void dma_kernel(const unsigned char* __restrict__ src, long long wg_stride, int span, int iters, unsigned* sink) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned char* base = src + (long long)blockIdx.x * wg_stride + lane * 16;
  unsigned char* ring = lds + wave * (D + 1) * 1024;
  int off = wave * 1024, slot = 0;
  u32x4 acc = {0, 0, 0, 0};
  for (int i = 0; i < iters; ++i) {
    if constexpr (DMA) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off), (__attribute__((address_space(3))) void*)(ring + slot * 1024), 16, 0, 0);
      wait_le<D>();
    } else {
      const u32x4 v = *reinterpret_cast<const u32x4*>(base + off);
      acc ^= v;      // (dependent use: the compiler counts its own waits; D is the unroll below)
    }
    off += 4096; if (off >= span) off -= span;
    slot = slot == D ? 0 : slot + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc[0] == 0x12345u) sink[0] = acc[1] + acc[2] + acc[3];
}

int main() {
  const size_t bytes = (size_t)1536 << 20;
  unsigned char* buf; unsigned* sink; CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 64)); CK(hipMemset(buf, 1, bytes));
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  const int cus = pr.multiProcessorCount;
  printf("CUs %d, clock %d MHz\n", cus, pr.clockRate / 1000);
  const int iters = 4096;
  auto run = [&](auto kern, int D, int wgs_per_cu, int span, long long stride, const char* name) {
    const int blocks = cus * wgs_per_cu;
    const size_t ldsb = (size_t)4 * (D + 1) * 1024;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(a);
      hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), ldsb, 0, buf, stride, span, iters, sink);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    const double tot = (double)blocks * 4 * iters * 1024;
    printf("%-10s D %2d  WG/CU %d  span %4d KB %s: %8.1f us  %6.2f TB/s  %6.1f GB/s/CU  %5.1f B/clk/CU @2.1GHz\n", name, D, wgs_per_cu, span >> 10,
           stride ? "private" : "shared ", best * 1e3, tot / best / 1e9, tot / best / 1e6 / cus, tot / (best * 1e-3) / cus / 2.1e9);
  };
  // L1-resident (16 KB per workgroup), L2-resident (128 KB / 512 KB per workgroup: 3 per CU x 32 CUs per XCD = 12 / 48 MB per XCD > 4 MB L2 for the
  // larger one), shared stream (every workgroup walks the SAME 4 MB: one L2 copy per XCD, the GEMM's weight operand)
  for (int wg : {1, 2, 3}) {
    for (int span_kb : {16, 128, 512}) {
      const long long stride = (long long)span_kb << 10;
      run(dma_kernel<4, true>, 4, wg, span_kb << 10, stride, "lds-dma");
      run(dma_kernel<16, true>, 16, wg, span_kb << 10, stride, "lds-dma");
      run(dma_kernel<8, false>, 8, wg, span_kb << 10, stride, "vgpr-load");
    }
    run(dma_kernel<4, true>, 4, wg, 4 << 20, 0, "lds-dma");
    run(dma_kernel<16, true>, 16, wg, 4 << 20, 0, "lds-dma");
  }
  return 0;
}
