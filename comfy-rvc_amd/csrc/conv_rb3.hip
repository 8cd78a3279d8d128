// A whole ResBlock1 of the generator's narrow stages in ONE launch, fp16x2 arithmetic (gfx950 only): all three ResBlocks (3 / 7 / 11 taps) of the 32-channel
// stage, the 3- and 7-tap ResBlocks of the 64-channel stage.
//
//   x1 = x  + c2_0(lrelu(c1_0(lrelu(x )) + b))      (dilation 1)
//   x2 = x1 + c2_1(lrelu(c1_1(lrelu(x1)) + b))      (dilation 3)
//   x3 = x2 + c2_2(lrelu(c1_2(lrelu(x2)) + b))      (dilation 5)
//   y  = x3 * scale [+ y]                                          (reference lib/infer_pack/modules.py:295-308, models.py:555-560: xs += resblock(x); x = xs / 3)
//
// conv_rbh_kernel (conv_rbh.hip) runs ONE (c1, c2) pair per launch: every pair reads the 164 MB stage tensor twice (input tile + residual) and writes it
// once, the third pair reads the running sum as well, and its two convolutions are 2 - 7 k cycles of a 17 - 24 k-cycle tile - the rest is the memory
// path (profiles/r6c_rbh_phase_cycles.txt: "the next step is bytes, not scheduling").  Here the three pairs of a ResBlock stay in the workgroup:
//   * a tile is 32 channels x 512 columns (a wave owns 32 rows x 64 columns: one row block, two column blocks) or 64 channels x 256 columns (64 rows x 32
//     columns: two row blocks, one column block); a wave owns its columns in EVERY convolution and in the fp32 residual stream, which never leaves its
//     registers between pairs (the second convolution of a pair accumulates INTO x_i + b2: x_{i+1} with no copy);
//   * convolutions are centred (column c reads c + (tap - P2) dil), so coordinates never shift and the residual needs no realignment; garbage creeps in
//     from the tile edges by the halo of each convolution - HALO = P2 (1 + 3 + 5 + 3) columns per side in total - and only the inner
//     NO = tile - 2 HALO columns are stored (32 channels: 488 / 440 / 392 at 3 / 7 / 11 taps; 64 channels: 232 / 184 at 3 / 7): 5 - 39 % more matrix work for a
//     third of the bytes (64 channels x 7 taps: a sixth - those pairs ran as two launches each with the intermediate image through HBM);
//   * x is read ONCE, in the accumulator layout (it is the residual), and its leaky-ReLU'd fp16 hi / lo image is written to LDS from those registers -
//     the pair kernel's second read of the tile is gone; the next tile's x is requested at the top of the tile and consumed a tile later, the previous
//     output (ACC) under or behind the last convolution: the barriers are LDS-only (lds_barrier), global requests stay in flight across the convolutions;
//   * weights: one-plane fp16 images of the six layers, [unit][half][C rows][16 B].  WM = 0: all six resident in LDS (32 channels, 3 / 7 taps: 36 / 84 KiB).
//     WM = 1: the NEXT convolution's rows are fetched from L2 into registers under the current one and stored into the other of two LDS buffers before the
//     barrier that precedes their use (32 channels x 11 taps: 2 x 22 KiB; 64 x 3: 2 x 24 KiB).  WM = 2: ONE buffer (64 channels x 7 taps: 56 KiB beside the 82 KiB
//     image) - the rows travel the same way and are stored in the image interval behind the convolution, between the barrier that retires it and the one that
//     opens the next;
//   * the last stage's noise branch (Conv1d(1, 32, 1) of the harmonic source) can be added where x is read (Rb3Args::nsrc): noise_add_kernel's pass disappears.
// Numerics: at 32 channels operation for operation the chain of three conv_rbh_kernel launches (same unit order, lo term before hi term, fp32 accumulation, bias and
// residual as the accumulator's initial value), so the result is BIT-IDENTICAL to it (tests/test_hip_ops.py::test_fused_resblock_matches_pair_chain); at 64 channels
// the same arithmetic against fp64 (the pairs it replaces ran in bf16x3 / as split pairs).
// Where the time goes (profiles/r6f_rb3_resblock.txt): the fp16 hi / lo conversion of every image (6.5 VALU instructions per element) costs as much issue time as
// the MFMAs at 3 taps, and the two waves of a SIMD run their phases one after the other - a variant with two phase-shifted groups of waves did not overlap them.
// LDS: weights + 2 KiB biases / noise + (C / 16) chunks x (hi | lo) x 2 halves x P rows x 16 B, P = tile + 2 x 5 P2: 105 / 157 / 118 KiB (32 ch), 117 / 142 KiB (64 ch).
#include "conv_x3_dev.h"

namespace rvc {

struct Rb3Args {
  const float* X; long long ldX; float* Y; long long ldY;
  const unsigned char* W[6];      // c1_0, c2_0, c1_1, c2_1, c1_2, c2_2: one-plane fp16 images [chunk][tap][half][CoPx rows][8 ch]
  const float* B[6];              // their biases (or null)
  int CoPx;
  int T, halo, NO;                // sequence length; columns lost per side of a tile; columns stored per tile (dilations 1 / 3 / 5, the margins and the rows of the image are compiled in)
  float pre_slope, mid_slope, out_scale;
  // NSF noise branch of the last generator stage folded into the read of x (reference models.py GeneratorNSF.forward: x = ups(x) + noise_convs[i](har), the
  // last stage's Conv1d(1, C, 1)): x[c][t] + fmaf(nw[c], nsrc[t], nb[c]) - the same operations as noise_add_kernel<1> (ops.hip), whose pass over the
  // tensor (one read, one write of 164 MB) disappears; null: x as it is
  const float* nsrc; const float* nw; const float* nb;
};

template <int T, int N, class F> __device__ __forceinline__ void rb3_for(F& f) {
  if constexpr (T < N) { f(std::integral_constant<int, T>{}); rb3_for<T + 1, N>(f); }
}
typedef float f32x4q __attribute__((ext_vector_type(4)));

#ifdef RVC_CONV_TIMING
__device__ unsigned long long g_rb3_timing[8];   // [0] tiles, [1] x image + requests, [2] first convolutions, [3] intermediate images, [4] second convolutions, [5] epilogue, [6] total, [7] barrier waits
void conv_rb3_timing_read(unsigned long long* out8, bool reset) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_rb3_timing), sizeof(unsigned long long) * 8);
  if (reset) { unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_rb3_timing), z, sizeof(z)); }
}
#define R3TICK() ((long long)__builtin_readcyclecounter())
#define R3ACC(i, v) do { r3t[i] += (v); } while (0)
#else
void conv_rb3_timing_read(unsigned long long* out8, bool) { for (int i = 0; i < 8; ++i) out8[i] = 0; }
#define R3TICK() 0ll
#define R3ACC(i, v) do {} while (0)
#endif

template <int CH, int KT, bool ACC, int WM>
__global__ __launch_bounds__(512, 2) void conv_rb3_kernel(const Rb3Args p) {
  // CH = 32: a wave owns 32 rows x 64 columns (one row block, two column blocks), the tile is 512 columns; CH = 64: 64 rows x 32 columns, 256 columns
  constexpr int C = CH, NCK = C / 16, NW = 8, AM = C / 32, AN = 64 / C, TILE = NW * AN * 32, P2 = (KT - 1) / 2;
  constexpr int DMAX = 5, M = P2 * DMAX, P = TILE + 2 * M;   // dilations 1, 3, 5 (the host declines anything else): every LDS offset is an immediate
  constexpr int NU = NCK * KT;                               // (chunk, tap) units of one convolution
  constexpr int WB = NU * 2 * C * 16;                        // bytes of one convolution's weights: [unit][half][C rows][16 B]
  constexpr int WROWS = NU * 2 * C;                          // 16-byte rows of one convolution's weights
  // WM: where the six convolutions' weights live.  0: all resident in LDS.  1: two LDS buffers - the next convolution's rows travel through registers under the
  // current convolution and are stored into the other buffer behind it.  2: ONE LDS buffer (64 channels x 7 taps: 56 KiB per convolution) - the rows
  // travel the same way but are stored in the image interval that follows (after the barrier that retires the convolution, before the one that opens the next)
  constexpr int NWB = WM == 0 ? 6 : (WM == 1 ? 2 : 1);
  constexpr int WQ = (WROWS + NW * 64 - 1) / (NW * 64);      // rows per thread when a convolution's weights travel through registers
  constexpr int xplane = P * 32, xhalf = P * 16, xbuf = 2 * xplane;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem3[];
  unsigned char* Ws = smem3;
  float* Bs = reinterpret_cast<float*>(smem3 + NWB * WB);    // 6 x C biases [, noise weights | biases]
  unsigned char* Xs = smem3 + NWB * WB + 2048;

  const int tid0 = threadIdx.x;
  const int lane = tid0 & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const float pre_slope = p.pre_slope, hs = p.mid_slope, oscale = p.out_scale;
  const int T = p.T, NO = p.NO, HALO = p.halo;
  const int ntiles = (T + NO - 1) / NO;
  const __amdgpu_buffer_rsrc_t xrs = make_rsrc(p.X, (unsigned)C * (unsigned)p.ldX * 4u);
  const __amdgpu_buffer_rsrc_t yrs = make_rsrc(p.Y, (unsigned)C * (unsigned)p.ldY * 4u);

  // ---- weights: source row r of a convolution = (unit, half) r / C, channel row r % C
  u32x4 wreg[WQ];
  auto wload = [&](int c) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t wrs = make_rsrc(p.W[c], (unsigned)(NU * 2) * (unsigned)p.CoPx * 16u);
#pragma unroll
    for (int q = 0; q < WQ; ++q) {
      const int r = tid0 + NW * 64 * q;
      wreg[q] = __builtin_amdgcn_raw_buffer_load_b128(wrs, (int)(r < WROWS ? (unsigned)((r / C) * p.CoPx + (r % C)) * 16u : kOOB), 0, 0);
    }
  };
  auto wstore = [&](unsigned char* dst) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < WQ; ++q) {
      const int r = tid0 + NW * 64 * q;
      if (r < WROWS) *reinterpret_cast<u32x4*>(dst + r * 16) = wreg[q];
    }
  };
  if constexpr (WM == 0) {
    for (int c = 0; c < 6; ++c) { wload(c); wstore(Ws + c * WB); }
  } else if constexpr (WM == 1) {
    wload(0); wstore(Ws);
  } else {
    wload(0);                                                  // (stored at the top of the first tile)
  }
  if (tid0 < 6 * C) Bs[tid0] = p.B[tid0 / C] ? p.B[tid0 / C][tid0 % C] : 0.f;
  const bool noise = C == 32 && p.nsrc != nullptr;
  if (noise && tid0 < 2 * C) Bs[6 * C + tid0] = tid0 < C ? p.nw[tid0] : p.nb[tid0 - C];      // noise weights | biases
  // the margins (M rows in front of column 0 and behind the last column of every half-plane) are read by the edge columns' taps and never written again
  for (int r = tid0; r < NCK * 4 * 2 * M; r += NW * 64) {
    const int pl = r / (2 * M), q = r - pl * 2 * M;           // half-plane (chunk, hi | lo, half), margin row
    const int row = q < M ? q : TILE + q;
    *reinterpret_cast<u32x4*>(Xs + (pl >> 2) * xbuf + ((pl >> 1) & 1) * xplane + (pl & 1) * xhalf + row * 16) = u32x4{0u, 0u, 0u, 0u};
  }

  // ---- global <-> accumulator layout: register r of block (am, j) = channel 32 am + (r & 3) + 8 (r >> 2) + 4 lh, tile column (AN wave + j) 32 + li.  The row
  // part of an address that does not depend on the lane is the instruction's scalar offset: one VGPR per column block instead of sixteen
  auto tile_voff = [&](long long ld, int tile, int j, int c_lo, int c_hi) __attribute__((always_inline)) -> unsigned {
    const int c = (wave * AN + j) * 32 + li;
    const int n = tile * NO - HALO + c;
    const bool ok = tile < ntiles && c >= c_lo && c < c_hi && n >= 0 && n < T;
    return ok ? ((unsigned)(4 * lh) * (unsigned)ld + (unsigned)n) * 4u : kOOB;
  };
  auto load_tile = [&](const __amdgpu_buffer_rsrc_t& rs, long long ld, int tile, f32x16 (&v)[AM][AN], int c_lo, int c_hi) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < AN; ++j) {
      const unsigned voff = tile_voff(ld, tile, j, c_lo, c_hi);
#pragma unroll
      for (int am = 0; am < AM; ++am)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          v[am][j][r] = buf_load(rs, voff, (unsigned)(32 * am + (r & 3) + 8 * (r >> 2)) * (unsigned)ld * 4u);
    }
  };
  // ---- accumulator layout -> fp16 hi / lo rows of this wave's columns: v = lrelu(a + bias) inside the sequence, 0 outside (every convolution pads with zeros)
  // (EDGE = false: the whole tile lies inside the sequence - every tile but the first and the last one or two - and the selects are not compiled)
  auto put_image_t = [&](auto edge_c, const f32x16 (&a)[AM][AN], const float* bias, float slope, int n0) __attribute__((always_inline)) {
    constexpr bool EDGE = decltype(edge_c)::value;
#pragma unroll
    for (int j = 0; j < AN; ++j) {
      const int c = (wave * AN + j) * 32 + li;
      const int pos = n0 + c;
      const bool inside = !EDGE || (pos >= 0 && pos < T);
#pragma unroll
      for (int am = 0; am < AM; ++am)
#pragma unroll
      for (int g2 = 0; g2 < 2; ++g2) {
        f32x4q ba = {0.f, 0.f, 0.f, 0.f}, bb = {0.f, 0.f, 0.f, 0.f};
        if (bias) { ba = *reinterpret_cast<const f32x4q*>(bias + 32 * am + 16 * g2 + 4 * lh); bb = *reinterpret_cast<const f32x4q*>(bias + 32 * am + 16 * g2 + 4 * lh + 8); }
        unsigned hA[2], lA[2], hB[2], lB[2];
#pragma unroll
        for (int e2 = 0; e2 < 2; ++e2) {
          float a0 = a[am][j][8 * g2 + 2 * e2] + ba[2 * e2], a1 = a[am][j][8 * g2 + 2 * e2 + 1] + ba[2 * e2 + 1];
          float b0 = a[am][j][8 * g2 + 4 + 2 * e2] + bb[2 * e2], b1 = a[am][j][8 * g2 + 5 + 2 * e2] + bb[2 * e2 + 1];
          a0 = inside ? fmaxf(a0, a0 * slope) : 0.f; a1 = inside ? fmaxf(a1, a1 * slope) : 0.f;
          b0 = inside ? fmaxf(b0, b0 * slope) : 0.f; b1 = inside ? fmaxf(b1, b1 * slope) : 0.f;
          split2h(a0, a1, hA[e2], lA[e2]);
          split2h(b0, b1, hB[e2], lB[e2]);
        }
        u32x4 hi, lo;
#pragma unroll
        for (int e2 = 0; e2 < 2; ++e2) {
          typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
          const u32x2_t sh = __builtin_amdgcn_permlane32_swap(hA[e2], hB[e2], false, false);
          const u32x2_t sl = __builtin_amdgcn_permlane32_swap(lA[e2], lB[e2], false, false);
          hi[e2] = sh.x; hi[2 + e2] = sh.y; lo[e2] = sl.x; lo[2 + e2] = sl.y;
        }
        unsigned char* d = Xs + (2 * am + g2) * xbuf + lh * xhalf + (M + c) * 16;
        *reinterpret_cast<u32x4*>(d) = hi;
        *reinterpret_cast<u32x4*>(d + xplane) = lo;
      }
    }
  };
  auto put_image = [&](const f32x16 (&a)[AM][AN], const float* bias, float slope, int n0) __attribute__((always_inline)) {
    if (n0 >= 0 && n0 + TILE <= T) put_image_t(std::false_type{}, a, bias, slope, n0); else put_image_t(std::true_type{}, a, bias, slope, n0);
  };

  // ---- one centred convolution over the wave's rows x columns: NU units, the operands of unit u + 1 requested before the MFMAs of unit u
  const int aoff = (lh * C + li) * 16;
  const int boff = lh * xhalf + (wave * AN * 32 + li) * 16;
  // ZERO: the accumulators start from zero - the first matrix instruction takes the constant as its addend, nothing is cleared beforehand
  auto conv = [&](auto zero_c, auto dil_c, f32x16 (&acc)[AM][AN], const unsigned char* W) __attribute__((always_inline)) {
    constexpr bool ZERO = decltype(zero_c)::value;
    constexpr int dil = decltype(dil_c)::value;
    constexpr int d16 = dil * 16;
    const int base = (M - P2 * dil) * 16 + boff;
    u32x4 a[AM], bh[AN], bl[AN], an_[AM], bhn[AN], bln[AN];
    auto read_ops = [&](u32x4 (&a_)[AM], u32x4 (&b_h)[AN], u32x4 (&b_l)[AN], int u, int xoff) __attribute__((always_inline)) {
#pragma unroll
      for (int am = 0; am < AM; ++am) a_[am] = *reinterpret_cast<const u32x4*>(W + u * (2 * C * 16) + am * 512 + aoff);
      const unsigned char* xa = Xs + xoff + base;
#pragma unroll
      for (int j = 0; j < AN; ++j) { b_h[j] = *reinterpret_cast<const u32x4*>(xa + j * 512); b_l[j] = *reinterpret_cast<const u32x4*>(xa + xplane + j * 512); }
    };
    read_ops(a, bh, bl, 0, 0);
    auto unit = [&](auto uc) __attribute__((always_inline)) {
      constexpr int U = decltype(uc)::value;
      if constexpr (U + 1 < NU) {
        constexpr int Tn = (U + 1) % KT, Cn = (U + 1) / KT;
        read_ops(an_, bhn, bln, U + 1, Cn * xbuf + Tn * d16);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int am = 0; am < AM; ++am)
#pragma unroll
        for (int j = 0; j < AN; ++j) {
          if constexpr (ZERO && U == 0) {
            const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[am][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[am]), __builtin_bit_cast(f16x8, bl[j]), z, 0, 0, 0);
          } else {
            acc[am][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[am]), __builtin_bit_cast(f16x8, bl[j]), acc[am][j], 0, 0, 0);
          }
        }
#pragma unroll
      for (int am = 0; am < AM; ++am)
#pragma unroll
        for (int j = 0; j < AN; ++j)
          acc[am][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[am]), __builtin_bit_cast(f16x8, bh[j]), acc[am][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (U + 1 < NU) {
#pragma unroll
        for (int am = 0; am < AM; ++am) a[am] = an_[am];
#pragma unroll
        for (int j = 0; j < AN; ++j) { bh[j] = bhn[j]; bl[j] = bln[j]; }
      }
    };
    rb3_for<0, NU>(unit);
  };
  // convolution c of the tile (0 .. 5): resident weights are where they are; streamed ones alternate between the two buffers and the next
  // convolution's travel through registers under this one
  auto run_conv = [&](auto zero_c, auto dil_c, f32x16 (&acc)[AM][AN], int c) __attribute__((always_inline)) {
    if constexpr (WM == 0) {
      conv(zero_c, dil_c, acc, Ws + c * WB);
    } else if constexpr (WM == 1) {
      wload(c == 5 ? 0 : c + 1);
      conv(zero_c, dil_c, acc, Ws + (c & 1) * WB);
      wstore(Ws + ((c + 1) & 1) * WB);                        // (last read by convolution c - 1: a barrier ago)
    } else {
      wload(c == 5 ? 0 : c + 1);                               // (stored by the image interval behind this convolution)
      conv(zero_c, dil_c, acc, Ws);
    }
  };

#ifdef RVC_CONV_TIMING
  long long r3t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  // xn: the next tile's x, in flight across the whole tile.  rs: the fp32 residual stream x_i - the second convolution of every pair accumulates INTO it
  // (x_{i+1} = x_i + b2 + c2(h): no copies).  ac: the first convolution's accumulators; dead once h is written, so the previous output (ACC) is loaded into them
  f32x16 xn[AM][AN], rs[AM][AN], ac[AM][AN];
  float sn[AN];                                               // the noise source at this lane's columns of the next tile
  const __amdgpu_buffer_rsrc_t srs = make_rsrc(noise ? p.nsrc : p.X, noise ? (unsigned)T * 4u : 0u);
  auto load_src = [&](int tile) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < AN; ++j) {
      const int n = tile * NO - HALO + (wave * AN + j) * 32 + li;
      sn[j] = buf_load(srs, (noise && tile < ntiles && n >= 0 && n < T) ? (unsigned)n * 4u : kOOB);
    }
  };
  load_tile(xrs, p.ldX, (int)blockIdx.x, xn, 0, TILE);
  load_src((int)blockIdx.x);
  __syncthreads();                                            // weights, biases and the zero margins are in LDS
  [[maybe_unused]] long long tq = R3TICK();
  [[maybe_unused]] const long long tq0 = tq;
#define R3PHASE(i) do { [[maybe_unused]] const long long t_ = R3TICK(); R3ACC(i, t_ - tq); tq = t_; } while (0)
#define R3BARRIER() do { [[maybe_unused]] const long long t0_ = R3TICK(); lds_barrier(); [[maybe_unused]] const long long t1_ = R3TICK(); R3ACC(7, t1_ - t0_); } while (0)
  for (int tile = (int)blockIdx.x; tile < ntiles; tile += (int)gridDim.x) {
    const int n0 = tile * NO - HALO;                          // position of tile column 0
    // ---- x (requested a tile ago) becomes the residual stream and, leaky-ReLU'd, the first image; the next tile's x is requested
#pragma unroll
    for (int am = 0; am < AM; ++am)
#pragma unroll
      for (int j = 0; j < AN; ++j) rs[am][j] = xn[am][j];
    if constexpr (C == 32) {
      if (noise) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4q w4 = *reinterpret_cast<const f32x4q*>(Bs + 6 * C + 8 * g + 4 * lh), b4 = *reinterpret_cast<const f32x4q*>(Bs + 7 * C + 8 * g + 4 * lh);
#pragma unroll
          for (int j = 0; j < AN; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) rs[0][j][4 * g + e] += fmaf(w4[e], sn[j], b4[e]);
        }
      }
    }
    if constexpr (WM == 2) wstore(Ws);                         // convolution 0's weights (the previous tile's last convolution was retired by its closing barrier)
    put_image(rs, nullptr, pre_slope, n0);
    load_tile(xrs, p.ldX, tile + (int)gridDim.x, xn, 0, TILE);
    load_src(tile + (int)gridDim.x);
    R3BARRIER();
    R3PHASE(1);
    auto pair = [&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      run_conv(std::true_type{}, std::integral_constant<int, 2 * i + 1>{}, ac, 2 * i);      // dilation 1, 3, 5
      R3PHASE(2);
      R3BARRIER();                                            // every wave is done with the pair's input image
      if constexpr (WM == 2) wstore(Ws);
      put_image(ac, Bs + (2 * i) * C, hs, n0);                 // h = lrelu(c1 + b1) over it
#pragma unroll
      for (int am = 0; am < AM; ++am)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4q b2 = *reinterpret_cast<const f32x4q*>(Bs + (2 * i + 1) * C + 32 * am + 8 * g + 4 * lh);
#pragma unroll
          for (int j = 0; j < AN; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) rs[am][j][4 * g + e] += b2[e];
        }
      // the previous output, under the last convolution (at 7 and 11 taps and at 64 channels there are no registers for it beside the convolution's operands: requested behind it)
      if constexpr (ACC && KT < 7 && C == 32 && i == 2) load_tile(yrs, p.ldY, tile, ac, HALO, HALO + NO);
      R3BARRIER();                                            // the intermediate is complete
      R3PHASE(3);
      run_conv(std::false_type{}, std::integral_constant<int, 1>{}, rs, 2 * i + 1);          // rs = x_{i+1}
      if constexpr (ACC && !(KT < 7 && C == 32) && i == 2) load_tile(yrs, p.ldY, tile, ac, HALO, HALO + NO);
      R3PHASE(4);
      if constexpr (i < 2) {
        R3BARRIER();                                          // every wave is done with the intermediate
        if constexpr (WM == 2) wstore(Ws);
        put_image(rs, nullptr, pre_slope, n0);
        R3BARRIER();
        R3PHASE(3);
      }
    };
    rb3_for<0, 3>(pair);
    // ---- epilogue: the inner NO columns
#pragma unroll
    for (int j = 0; j < AN; ++j) {
      const unsigned voff = tile_voff(p.ldY, tile, j, HALO, HALO + NO);
#pragma unroll
      for (int am = 0; am < AM; ++am)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = rs[am][j][r] * oscale;
          if constexpr (ACC) v += ac[am][j][r];
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), yrs, (int)voff, (int)((unsigned)(32 * am + (r & 3) + 8 * (r >> 2)) * (unsigned)p.ldY * 4u), 0);
        }
    }
    R3BARRIER();                                              // every wave is done with the last intermediate: the next tile's image may be written over it
    R3PHASE(5);
    R3ACC(0, 1);
  }
#undef R3PHASE
#undef R3BARRIER
#ifdef RVC_CONV_TIMING
  r3t[6] = R3TICK() - tq0;
  if (threadIdx.x == 0) for (int i = 0; i < 8; ++i) atomicAdd(&g_rb3_timing[i], (unsigned long long)r3t[i]);
#endif
}

template <int CH, int KT, bool ACC, int WM>
static void launch_rb3c(const Rb3Args& a, dim3 grid, size_t lds, hipStream_t s) {
  auto kern = conv_rb3_kernel<CH, KT, ACC, WM>;
  RVC_ALLOW_BIG_LDS(kern);
  conv_launch(kern, grid, dim3(512), lds, s, a);
}
template <int CH, int KT, int WM>
static void launch_rb3(const Rb3Args& a, bool acc, dim3 grid, size_t lds, hipStream_t s) {
  if (acc) launch_rb3c<CH, KT, true, WM>(a, grid, lds, s); else launch_rb3c<CH, KT, false, WM>(a, grid, lds, s);
}

// c1[i] / c2[i]: the three (dilated, plain) pairs of one ResBlock1.  32 channels, equal odd kernel size 3 / 7 / 11, "same" padding, every layer with its
// one-plane fp16 image, the fp16x2 pair arithmetic switched on, at least two rounds of tiles; false: not this kernel's (the caller runs the pairs one by one).
bool conv_rb3_try(const ConvLayer* const* c1, const ConvLayer* const* c2, hipStream_t s, const float* X, long long ldX, int T, float* Y, long long ldY,
                  float pre_slope, float out_scale, int accumulate, bool dry, const float* nsrc, const float* nw, const float* nb) {
  static const int on = exp_int("RVC_RB3", 1);
  static const int on64 = exp_int("RVC_RB3_64", 1);             // the 64-channel stage's 3-tap ResBlock (otherwise three conv_x3pf_kernel launches in bf16x3)
  if (!on || !conv_x3_enabled() || !conv_set_pair_arithmetic(-1)) return false;
  const int k = c1[0]->k, C = c1[0]->Co;
  static const int on64k7 = exp_int("RVC_RB3_64K7", 1);         // the 64-channel stage's 7-tap ResBlock (otherwise six conv_x3q_kernel launches with the intermediate images through HBM)
  if (!((C == 32 && (k == 3 || k == 7 || k == 11)) || (C == 64 && k == 3 && on64) || (C == 64 && k == 7 && on64 && on64k7))) return false;
  int dsum = 0;
  for (int i = 0; i < 3; ++i) {
    const ConvLayer& a = *c1[i]; const ConvLayer& b = *c2[i];
    if (!a.Wh_ || !b.Wh_ || a.mode != 1 || b.mode != 1 || a.groups != 1 || b.groups != 1 || a.stride != 1 || b.stride != 1 || a.tconv_u || b.tconv_u) return false;
    if (a.Ci != C || a.Co != C || b.Ci != C || b.Co != C || a.k != k || b.k != k || b.dil != 1 || a.dil != 2 * i + 1) return false;      // dilations 1, 3, 5: the kernel's LDS offsets are compiled for these
    if (a.pad != (k - 1) / 2 * a.dil || b.pad != (k - 1) / 2 || a.CoPx != c1[0]->CoPx || b.CoPx != c1[0]->CoPx || a.CoPx < C) return false;
    dsum += a.dil;
  }
  const int TILE = C == 32 ? 512 : 256;
  const int P2 = (k - 1) / 2, M = P2 * 5, P = TILE + 2 * M, halo = P2 * (dsum + 3), NO = TILE - 2 * halo;
  if ((double)C * (double)ldX * 4.0 >= 2147483648.0 || (double)C * (double)ldY * 4.0 >= 2147483648.0) return false;
  if (nsrc && C != 32) return false;
  int dev = 0, ncu = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
  if (ncu <= 0) ncu = 256;
  const long long ntiles = ((long long)T + NO - 1) / NO;
  if (ntiles < 2LL * ncu) return false;                       // short sequences: the pair kernels' smaller tiles fill the chip better
  if (dry) return true;
  Rb3Args a{};
  a.X = X; a.ldX = ldX; a.Y = Y; a.ldY = ldY; a.CoPx = c1[0]->CoPx;
  for (int i = 0; i < 3; ++i) {
    a.W[2 * i] = reinterpret_cast<const unsigned char*>(c1[i]->Wh_); a.W[2 * i + 1] = reinterpret_cast<const unsigned char*>(c2[i]->Wh_);
    a.B[2 * i] = c1[i]->bd_; a.B[2 * i + 1] = c2[i]->bd_;
  }
  a.T = T; a.halo = halo; a.NO = NO;
  a.pre_slope = pre_slope; a.mid_slope = pre_slope; a.out_scale = out_scale;
  RVC_REQUIRE((nsrc == nullptr) == (nw == nullptr) && (nsrc == nullptr) == (nb == nullptr), "conv_rb3_try: the noise branch is source, weights and biases together");
  a.nsrc = nsrc; a.nw = nw; a.nb = nb;
  const size_t wb = (size_t)(C / 16) * k * 2 * C * 16;
  const size_t tile_bytes = (size_t)P * (C / 16) * 64;
  // six weight sets beside the image: 36 / 84 KiB fit (32 channels, 3 / 7 taps); two buffers at 11 taps (2 x 22 KiB) and at 64 channels x 3 taps (2 x 24 KiB); one at
  // 64 channels x 7 taps (56 KiB beside the 82 KiB image)
  const int wm = (C == 32 && k <= 7) ? 0 : ((C == 64 && k == 7) ? 2 : 1);
  const size_t lds = (wm == 0 ? 6 : (wm == 1 ? 2 : 1)) * wb + 2048 + tile_bytes;
  RVC_REQUIRE(lds <= 160 * 1024, "conv_rb3_try: LDS budget");
  dim3 grid((unsigned)(ntiles < ncu ? ntiles : ncu), 1, 1);
  ProfTicket tk = conv_prof_begin(s);
  if (C == 64 && k == 3) launch_rb3<64, 3, 1>(a, accumulate != 0, grid, lds, s);
  else if (C == 64) launch_rb3<64, 7, 2>(a, accumulate != 0, grid, lds, s);
  else if (k == 3) launch_rb3<32, 3, 0>(a, accumulate != 0, grid, lds, s);
  else if (k == 7) launch_rb3<32, 7, 0>(a, accumulate != 0, grid, lds, s);
  else launch_rb3<32, 11, 1>(a, accumulate != 0, grid, lds, s);
  if (tk.on) {
    ConvArgsX pa{};
    pa.Ci = C; pa.Co = C; pa.ktaps = k; pa.kreal = k; pa.dil = 5; pa.stride = 1; pa.Tin = T; pa.Tout = T; pa.ksplit = 1; pa.h2 = 1;
    pa.R = X; pa.X = X; pa.accumulate = accumulate;
    // algorithmic traffic of the ResBlock: x read once, y written once (+ the previous y when accumulating), six weight sets
    const double bytes = 4.0 * ((double)C * T * (2.0 + (accumulate ? 1.0 : 0.0)) + 6.0 * C * C * k);
    conv_prof_end(tk, s, 3.0 * 2.0 * 2.0 * (double)C * C * k * (double)T, 14 + (C == 32 ? 1 : 5), bytes, &pa, (long long)grid.x, 3 | (7 << 4));
  }
  return true;
}

}  // namespace rvc
