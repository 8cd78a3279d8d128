// What does the matrix pipe deliver under sustained load?  Bare v_mfma_f32_32x32x16_bf16 chains (no memory), W waves per SIMD, for a short (~0.3 ms) and a long (~20 ms)
// launch; then the same with the operand traffic of the generator's kernel (12 ds_read_b128 per 24 MFMAs).  Prints dense bf16 TFLOP/s and the bf16x3-algorithmic equivalent (/ 3).
//   hipcc --offload-arch=gfx950 -O3 -o mfmabench tools/micro/mfmabench.hip && ./mfmabench
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int NACC, int MODE>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, unsigned long long* clk, const unsigned char* src) {
  __shared__ __attribute__((aligned(1024))) unsigned char sm[MODE ? (MODE >= 5 ? 65536 : 32768) : 16];
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  u32x4 a = {0x3f803f80u + threadIdx.x, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  constexpr bool LDS = MODE != 0;
  if (LDS) { for (int i = threadIdx.x; i < 32768 / 4; i += 256) reinterpret_cast<unsigned*>(sm)[i] = 0x3f803f80u; __syncthreads(); }
  const unsigned char* base = sm + (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 4096;
  u32x4 cur[12];
  for (int q = 0; q < 12; ++q) cur[q] = a;
  unsigned sx = 0;
  float vx[8], vy = 0.999f + 1e-9f * threadIdx.x;
  for (int q = 0; q < 8; ++q) vx[q] = (float)q;
  const long long t0 = (long long)__builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 1) {
      // 12 operand reads for 24 MFMAs (conv_x3q_kernel's 2 x 4 wave tile: ah, al x 2 rows, bh, bl x 4 columns), all reads first
      u32x4 o[12];
#pragma unroll
      for (int q = 0; q < 12; ++q) o[q] = *reinterpret_cast<const u32x4*>(base + ((q * 1024 + it * 16) & 16383));
#pragma unroll
      for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int i = 0; i < 8; ++i)
          acc[i % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, o[(g + i) % 4]), __builtin_bit_cast(bf16x8, o[4 + (g * 3 + i) % 8]), acc[i % NACC], 0, 0, 0);
    } else if (MODE == 2 || MODE == 3) {
      // software-pipelined: the reads of iteration it + 1 are issued between the MFMAs of iteration it (one read every 2 (MODE 2) / 3 (MODE 3) MFMAs), waited for at the end
      constexpr int NR = MODE == 2 ? 12 : 8, EVERY = 24 / NR;
      static u32x4 dummy;
      u32x4 nx[12];
      if (it == 0) for (int q = 0; q < 12; ++q) cur[q] = *reinterpret_cast<const u32x4*>(base + q * 1024);
#pragma unroll
      for (int i = 0; i < 24; ++i) {
        if (i % EVERY == 0) { const int q = i / EVERY; asm volatile("ds_read_b128 %0, %1" : "=v"(nx[q]) : "v"((unsigned)(size_t)(base - sm) + ((q * 1024 + (it + 1) * 16) & 16383))); }
        acc[i % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, cur[i % 4]), __builtin_bit_cast(bf16x8, cur[4 + (i % (NR - 4))]), acc[i % NACC], 0, 0, 0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int q = 0; q < NR; ++q) { asm volatile("" : "+v"(nx[q])); cur[q] = nx[q]; }
    } else if (MODE >= 5) {
      constexpr int NP = MODE == 5 ? 2 : (MODE == 7 ? 8 : 4), NR = 12, EVERY = 2;      // MODE 8 / 9: 4 pieces and one s_barrier per 24 / 12 MFMAs
      u32x4 nx[12];
      if (it == 0) for (int q = 0; q < 12; ++q) cur[q] = *reinterpret_cast<const u32x4*>(base + q * 1024);
      __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(src), 0, 262144, 0x00020000);
      unsigned char* dst = sm + 32768 + (threadIdx.x >> 6) * 8192;
#pragma unroll
      for (int i = 0; i < 24; ++i) {
        if (i % EVERY == 0) { const int q = i / EVERY; asm volatile("ds_read_b128 %0, %1" : "=v"(nx[q]) : "v"((unsigned)(size_t)(base - sm) + ((q * 1024 + (it + 1) * 16) & 16383))); }
        if (i % (24 / NP) == 1) {
          const int pc = i / (24 / NP);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + pc * 1024), 16, (int)((threadIdx.x & 63) * 16), (int)(((it * NP + pc) * 1024 + blockIdx.x * 4096) & 262143), 0, 0);
        }
        acc[i % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, cur[i % 4]), __builtin_bit_cast(bf16x8, cur[4 + (i % (NR - 4))]), acc[i % NACC], 0, 0, 0);
        if ((MODE == 9 && i == 11) || (MODE >= 8 && i == 23)) asm volatile("s_barrier" ::: "memory");
        // MODE 10 / 11: 1 / 2 independent VALU instructions per MFMA; 12 / 13: one per 2 / per 4 MFMAs (the generator kernel's loops: 0.7 per MFMA where it converts its
        // fp32 input, 0.2 where it reads an image); 14: one SALU instruction per MFMA and no VALU (the kernels run ~1 per MFMA: offsets of the DMA pieces, counters)
        if (MODE == 10 || MODE == 11 || (MODE == 12 && (i & 1) == 0) || (MODE == 13 && (i & 3) == 0)) {
          asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(vx[i % 8]) : "v"(vy));
          if (MODE == 11) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(vx[(i + 4) % 8]) : "v"(vy));
        }
        if (MODE == 14) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sx) : : "scc");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (NP == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else if (NP == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // one iteration's pieces stay in flight
#pragma unroll
      for (int q = 0; q < NR; ++q) { asm volatile("" : "+v"(nx[q])); cur[q] = nx[q]; }
    } else if (MODE == 4) {
      typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
      u32x4 o[12];
#pragma unroll
      for (int q = 0; q < 12; ++q) {
        const u32x2 lo2 = *reinterpret_cast<const u32x2*>(base + ((q * 1024 + it * 16) & 16383)), hi2 = *reinterpret_cast<const u32x2*>(base + ((q * 1024 + it * 16) & 16383) + 8);
        o[q] = u32x4{lo2[0], lo2[1], hi2[0], hi2[1]};
      }
#pragma unroll
      for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int i = 0; i < 8; ++i)
          acc[i % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, o[(g + i) % 4]), __builtin_bit_cast(bf16x8, o[4 + (g * 3 + i) % 8]), acc[i % NACC], 0, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < 24; ++i) acc[i % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[i % NACC], 0, 0, 0);
    }
  }
  const long long t1 = (long long)__builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int q = 0; q < 8; ++q) s += vx[q];
  s += (float)sx;
  if (s == 12345.678f) out[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = (unsigned long long)(t1 - t0);
}

static double g_sustain_s = 0.0;      // > 0: repeat the launch for this many seconds (clock / power of the mix: poll rocm-smi beside it)
template <int NACC, int MODE>
static int run(const char* name, int wgs_per_cu, int iters) {
  float* out; unsigned long long* clk; CK(hipMalloc(&out, 64)); CK(hipMalloc(&clk, 64));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * wgs_per_cu;
  unsigned char* src; CK(hipMalloc(&src, 262144 + 4096)); CK(hipMemset(src, 0x3f, 262144 + 4096));
  hipLaunchKernelGGL((mfma_loop<NACC, MODE>), dim3(grid), dim3(256), 0, 0, out, 64, clk, src);
  float best = 1e9f;
  const int nrep = g_sustain_s > 0 ? (int)(g_sustain_s / 0.009) : 3;
  for (int rep = 0; rep < nrep; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((mfma_loop<NACC, MODE>), dim3(grid), dim3(256), 0, 0, out, iters, clk, src);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  unsigned long long c = 0; CK(hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost));
  const double flops = (double)grid * 4 * (double)iters * 24 * 2.0 * 32 * 32 * 16;
  // s_memtime counts at a constant 100 MHz: cycles of the shader clock per MFMA of one wave = (time / MFMAs per wave per SIMD)
  printf("%-44s %d WG/CU  %9.1f us  %7.0f TFLOP/s dense bf16 = %5.0f as bf16x3 (%.3f of 2500)   refclk ticks of workgroup 0: %llu\n", name, wgs_per_cu, best * 1e3, flops / best / 1e9,
         flops / best / 1e9 / 3, flops / best / 1e9 / 2500.0, c);
  hipFree(out); hipFree(clk); hipFree(src);
  return 0;
}
#include <cstdlib>
#include <cstring>
int main(int argc, char** argv) {
  if (argc >= 4 && !strcmp(argv[1], "sustain")) {
    g_sustain_s = atof(argv[3]);
    switch (atoi(argv[2])) {
      case 0: return run<8, 0>("bare MFMA (sustained)", 2, 30000);
      case 2: return run<8, 2>("12 reads between 24 MFMA (sustained)", 2, 10000);
      case 6: return run<8, 6>("12 reads + 4 LDS-DMA pieces (sustained)", 2, 10000);
      case 8: return run<8, 8>("12 reads + 4 pieces + barrier (sustained)", 2, 10000);
      case 10: return run<8, 10>("... + 1 VALU per MFMA (sustained)", 2, 10000);
      case 12: return run<8, 12>("... + 1 VALU per 2 MFMA (sustained)", 2, 10000);
      default: return 1;
    }
  }
  // iters: 24 MFMAs x 32 cycles = 768 cycles per iteration per wave at one wave per SIMD
  for (int w = 1; w <= 3; ++w) {
    run<4, 0>("bare MFMA, 4 accumulators, short", w, 1000 / w);
    run<4, 0>("bare MFMA, 4 accumulators, long (20 ms)", w, 60000 / w);
  }
  run<8, 0>("bare MFMA, 8 accumulators, long", 2, 30000);
  for (int w = 1; w <= 3; ++w) {
    run<8, 1>("12 ds_read_b128 then 24 MFMA, short", w, 1000 / w);
    run<8, 1>("12 ds_read_b128 then 24 MFMA, long", w, 60000 / w);
  }
  for (int w = 1; w <= 2; ++w) {
    run<8, 2>("12 ds_read_b128 between 24 MFMA (pipelined)", w, 20000 / w);
    run<8, 3>("8 ds_read_b128 between 24 MFMA (pipelined)", w, 20000 / w);
    run<8, 4>("24 ds_read_b64 then 24 MFMA", w, 20000 / w);
  }
  // the same pipelined loop plus LDS-DMA pieces (1 KiB per wave-instruction, L2-resident source): the generator kernel issues ~2.5 per wave and 24 MFMAs,
  // the split-resident GEMM kernel 8 (128 x 128 tile: 4 per 12)
  for (int w = 1; w <= 2; ++w) {
    run<8, 5>("12 reads + 2 LDS-DMA pieces between 24 MFMA", w, 20000 / w);
    run<8, 6>("12 reads + 4 LDS-DMA pieces between 24 MFMA", w, 20000 / w);
    run<8, 7>("12 reads + 8 LDS-DMA pieces between 24 MFMA", w, 20000 / w);
    run<8, 8>("12 reads + 4 pieces + 1 s_barrier per 24 MFMA", w, 20000 / w);
    run<8, 9>("12 reads + 4 pieces + 2 s_barrier per 24 MFMA", w, 20000 / w);
    run<8, 10>("... + 1 barrier + 1 VALU per MFMA", w, 20000 / w);
    run<8, 11>("... + 1 barrier + 2 VALU per MFMA", w, 20000 / w);
    run<8, 12>("... + 1 barrier + 1 VALU per 2 MFMA", w, 20000 / w);
    run<8, 13>("... + 1 barrier + 1 VALU per 4 MFMA", w, 20000 / w);
    run<8, 14>("... + 1 barrier + 1 SALU per MFMA", w, 20000 / w);
  }
  return 0;
}
