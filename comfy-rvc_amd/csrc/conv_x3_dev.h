// Device helpers shared by the bf16x3 kernels (conv_x3.hip: staged kernel for every geometry; conv_x3p.hip: software-pipelined kernel
// for the stride-1 1-D convolutions of the generator).  Private to csrc/.
#pragma once
#include "conv_kernels.h"

namespace rvc {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// 8-float register slots for the prefetched input tile per tile width (checked against the launch geometry on the host)
__host__ __device__ constexpr int x3_slots(int BN, int NW = 4) { return NW == 8 ? 3 : (BN >= 512 ? 7 : (BN >= 128 ? 5 : 2)); }

__device__ __forceinline__ unsigned bf16_bits(__bf16 h) { return (unsigned)__builtin_bit_cast(unsigned short, h); }
// (a, b) -> packed bf16 pairs hi = {bf16(a), bf16(b)} and lo = {bf16(a - hi_a), bf16(b - hi_b)}, round-to-nearest-even: 2 + 2 + 2 VALU
__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& lo) {
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t v = {a, b};
  hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
  const f32x2_t r = {a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u)};
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2_t));
}

// fp16x2 arithmetic of the ResBlock pairs (conv_x3q.hip, H2): x = hi + lo with hi = fp16(x) rounded toward zero (never past the largest finite
// value: no infinities from finite inputs below 2 x 65504) and lo = fp16(x - hi), round-to-nearest - 22 bits of x; the weight is ONE fp16 term.
// fp32(hi) is NOT obtained by converting hi back: rounding toward zero to fp16 keeps the top 10 mantissa bits, so fp32(hi) = x & 0xffffe000 over fp16's
// normal range (below 2^-14 the mask keeps bits fp16 cannot hold: an error under 2^-24 absolute, the resolution of fp16 subnormals anyway).
// Besides being one VALU cheaper per element, this avoids a sequence that CORRUPTS DATA on gfx950 / ROCm 7.2 (round 6, profiles/r6_sdwa_pk_hazard.txt): the
// back-conversion compiles to v_cvt_f32_f16_e32 vN + v_cvt_f32_f16_sdwa vN+1 (src0_sel:WORD_1) feeding v_pk_add_f32 ..., v[N:N+1] neg_lo neg_hi, and beside
// in-flight MFMAs under memory load that pair read stale registers in lanes 12-15 / 28-31 of the lower half-wave (first failing tile ~32 of 1250, different
// positions every run).  Keeping the two subtractions scalar (no v_pk_add_f32) or removing the SDWA conversion each makes it disappear.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split2h(float a, float b, unsigned& hi, unsigned& lo) {
  typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  hi = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a, b));
  const f32x2_t r = {a - __uint_as_float(__float_as_uint(a) & 0xffffe000u), b - __uint_as_float(__float_as_uint(b) & 0xffffe000u)};
  // round-to-nearest-even (v_cvt_pk_f16_f32): below |x| = 0.125 lo is an fp16 subnormal and a truncation there would be a bias that does not average out
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2_t));
}

// s_waitcnt vmcnt(n) for a wave-uniform run-time n: waits until at most n of this wave's VMEM operations (LDS-DMA pieces and register
// loads alike, retired in issue order) are outstanding.  A smaller count than necessary only waits longer, so n is clamped to the table.
__device__ __forceinline__ void wait_vmcnt_le(int n) {
#define RVC_VMC(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
  switch (n < 0 ? 0 : (n > 56 ? 56 : (n > 24 ? (n & ~7) : n))) {
    RVC_VMC(0) RVC_VMC(1) RVC_VMC(2) RVC_VMC(3) RVC_VMC(4) RVC_VMC(5) RVC_VMC(6) RVC_VMC(7) RVC_VMC(8) RVC_VMC(9) RVC_VMC(10) RVC_VMC(11) RVC_VMC(12)
    RVC_VMC(13) RVC_VMC(14) RVC_VMC(15) RVC_VMC(16) RVC_VMC(17) RVC_VMC(18) RVC_VMC(19) RVC_VMC(20) RVC_VMC(21) RVC_VMC(22) RVC_VMC(23) RVC_VMC(24)
    RVC_VMC(32) RVC_VMC(40) RVC_VMC(48) RVC_VMC(56)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
#undef RVC_VMC
}
// workgroup barrier that does NOT drain the VMEM queue (a __syncthreads() waits vmcnt(0) while LDS-DMA is in flight): LDS traffic of this
// wave is complete (lgkmcnt(0)), DMA pieces of later stages stay in flight across it
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
// Consecutive hardware workgroup ids go round-robin to the 8 XCDs (each with its own L2).  Tiles are renumbered so that every XCD works
// on one contiguous run of tiles (n fastest, same weight rows): neighbouring tiles share their halo columns and the weight image in
// one L2 instead of fetching them eight times.  Bijective for any tile count; a wrong guess about the placement is only slower.
__device__ __forceinline__ unsigned xcd_tile(unsigned b, unsigned total) {
  const unsigned q = total >> 3, r = total & 7u, x = b & 7u, i = b >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}


// Split-resident output (ConvEpilogue::ys_out): v = act(acc + bias) -> bf16 hi / lo rows [chunk][hi | lo][half][margin + n][8 ch].  A lane
// holds 4 + 4 channels of each 16-channel chunk of its column; v_permlane32_swap trades quads with the lane 32 away so that every lane
// owns one 16-B row of a half-plane: 4 b128 stores per accumulator instead of 16 dword stores, each half-wave a contiguous 512 B.
template <int WM, int WN, int AM, int AN>
__device__ __forceinline__ void ysplit_epilogue(const ConvArgsX& p, f32x16 (&acc)[AM][AN], int co0, int n0, int wm, int wn, int li, int lh) {
    const float sl = p.ys_slope;
      const float* __restrict__ bias = p.bias;
  #pragma unroll
      for (int am = 0; am < AM; ++am)
  #pragma unroll
        for (int an = 0; an < AN; ++an) {
          const int n = n0 + (wn * AN + an) * 32 + li;
          const int mb = co0 + (wm * AM + am) * 32;
          const long long pos = (long long)n + kSplitMargin;
  #pragma unroll
          for (int g2 = 0; g2 < 2; ++g2) {
            unsigned hA[2], lA[2], hB[2], lB[2];
  #pragma unroll
            for (int e2 = 0; e2 < 2; ++e2) {
              const int ma = mb + 16 * g2 + 4 * lh + 2 * e2;
              float a0 = acc[am][an][8 * g2 + 2 * e2] + (bias ? bias[ma] : 0.f), a1 = acc[am][an][8 * g2 + 2 * e2 + 1] + (bias ? bias[ma + 1] : 0.f);
              float b0 = acc[am][an][8 * g2 + 4 + 2 * e2] + (bias ? bias[ma + 8] : 0.f), b1 = acc[am][an][8 * g2 + 5 + 2 * e2] + (bias ? bias[ma + 9] : 0.f);
              split2(fmaxf(a0, a0 * sl), fmaxf(a1, a1 * sl), hA[e2], lA[e2]);
              split2(fmaxf(b0, b0 * sl), fmaxf(b1, b1 * sl), hB[e2], lB[e2]);
            }
            u32x4 hi, lo;
  #pragma unroll
            for (int e2 = 0; e2 < 2; ++e2) {
              typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
              const u32x2_t sh = __builtin_amdgcn_permlane32_swap(hA[e2], hB[e2], false, false);
              const u32x2_t sl2 = __builtin_amdgcn_permlane32_swap(lA[e2], lB[e2], false, false);
              hi[e2] = sh.x; hi[2 + e2] = sh.y; lo[e2] = sl2.x; lo[2 + e2] = sl2.y;
            }
            if (n < p.Tout && mb + 16 * g2 < p.Co) {
              const long long chunk = (mb >> 4) + g2;
              unsigned char* row = p.Ys + ((chunk * 4 + lh) * p.ysTp + pos) * 16;     // [chunk][hi | lo][half][position][8 ch]
              *reinterpret_cast<u32x4*>(row) = hi;
              *reinterpret_cast<u32x4*>(row + p.ysTp * 32) = lo;
            }
          }
        }
}

}  // namespace rvc
