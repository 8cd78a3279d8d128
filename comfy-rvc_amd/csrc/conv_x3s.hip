// bf16x3 GEMM on SPLIT-RESIDENT operands (gfx950 only): Y[m][n] = epilogue( sum_k W[m][k] X[k][n] ) for the k = 1 projections of the
// transformer stacks (HuBERT q/k/v, out, FFN; reference modeling_hubert.py:291-477 through lib/infer_pack/loaders.py:55-61).
//
// conv_x3g_kernel (conv_x3p.hip) reads the activation as fp32 and converts it to bf16 hi / lo in EVERY workgroup that needs it - a 768 -> 3072
// projection converts the same 768 x 128 input tile 24 (64-row tiles: 48) times, through registers, 45 VALU instructions per 12 MFMAs - and
// under-filled grids were cut along K with a second launch summing the partials.  Measured (profiles/r2o_launch_classes.md): 125 - 139
// TFLOP/s of 833, 4 - 5.7x the algorithmic bytes, 131 reduction launches per clip.  Here the activation LIVES as the image the kernel
// stages ([16-channel chunk][hi | lo][8-channel half][kSplitMargin + t][8 ch], the same format as the ResBlock intermediates of the
// generator), written once by its producer (LayerNorm, the attention's epilogue, this kernel's own epilogue), so that
//   * both operand tiles of a unit (one 16-channel chunk: BM weight rows + BN positions, 64 B each) reach LDS by DMA
//     (global_load_lds, 1 KiB per wave-instruction), no registers, no VALU; a ring of RS unit slots, a unit requested RS - 1 units ahead,
//     every wait an immediate (VMEM operations of a wave retire in order and the issue sequence is static);
//   * the wave runs the three MFMA groups of a unit as in conv_x3p_kernel: (hi_w lo_x), (lo_w hi_x), (hi_w hi_x), the second group's operands
//     requested before the first is issued, the next unit's before the third; one barrier per unit publishes the next slot;
//   * tiles are small (128 x 64 default: N = 1599 is 25 x 64 exactly, 600 workgroups for the 768 -> 3072 layer, three per CU) and the K split
//     of the deep reductions (3072 -> 768) is reduced INSIDE the launch: every slice writes its accumulators as a slab in register order
//     with write-through (sc1) 16-byte stores, drains, takes a ticket; the slice that draws the last ticket sums the slabs in slice order
//     (bit-identical whoever is last) with sc1 loads and runs the epilogue.  No fences (they flush the XCD's L2: round 2 measured 61 -> 126 us),
//     no second launch;
//   * bias and residual are loaded into slice 0's accumulators at tile start (the oldest VMEM operations of the wave), so the epilogue is
//     activation + store: fp32 rows, and / or the bf16 hi / lo image for the next GEMM (exact-erf GELU in fp32 before the split).
#include "conv_x3_dev.h"

namespace rvc {

struct GemmSArgs {
  const unsigned char* Wx; int CoPx;        // weight image [chunk][tap][hi | lo][half][CoPx rows][16 B]
  const unsigned char* Xs; long long xsTp;  // activation image [chunk][hi | lo][half][xsTp rows][16 B]; position t lives at row margin + t
  unsigned wx_bytes, xs_bytes;              // extents of the two images (buffer descriptors)
  int Co, T, nunits;                        // rows stored, columns, units of the reduction: (16-channel chunk, tap), tap fastest
  int ktaps, margin;                        // taps per chunk; image rows in front of position 0 (>= the largest |tap offset|, kept zero by the producers)
  int toff[16];                             // tap -> position offset (1-D: tap * dil - pad; padded 2-D: dh * padw + dw)
  int groups, co_g, rows_pg, cig_chunks;    // grouped convolution: groups, output rows per group, row tiles per group, input chunks per group
  unsigned wg_bytes;                        // weight image bytes per group (groups > 1)
  int tdil, tpad;                           // 1-D taps beyond the table (ktaps > 16): offset = tap * tdil - tpad
  int padw; unsigned padmagic;              // padded 2-D images: row pitch (W + 2) - columns 0 and padw - 1 of every row are written as zeros - and ceil(2^32 / padw)
  const float* bias; const float* R; long long ldR;
  float* Y; long long ldY;                  // fp32 output [Co][ldY] or null
  unsigned char* Ys; long long ysTp;        // split output image or null
  int act; float act_slope; int act_before_res; float out_scale;
  int ksplit; float* slabs; unsigned* tickets;
  int gx, gy;                               // column / row tiles
  int xcd_remap;
  int ymargin;                              // output image: position n lives at row ymargin + n (= margin except for the swapped product)
  int zero_tail;                            // rows >= Co of the last stored 16-row chunk are written as zeros (swapped product: keys past the end)
  int row_fast;                             // tile numbering inside an XCD's run: 1 = row tiles fastest (the XCD owns a column range), 0 = column tiles fastest
  unsigned char* Vt; long long vtTp;        // row tiles from vt_row0 on (the v rows of a fused q | k | v projection) are written TRANSPOSED, as the V^T image of the attention
  int vt_row0, vt_rows;                     // ([16-position chunk][hi | lo][8-position half][kSplitMargin + row - vt_row0][8 positions], like conv_x3s_run_swapped); INT_MAX: none
  const float* gate_g; int gate_h;          // WaveNet gate in the epilogue (gate_h = H > 0): the 2 H rows are packed so that every 32-row block holds 16 tanh rows and the 16 sigmoid rows of the
                                            // same channels; the image gets tanh(a + g[c]) sigmoid(a' + g[H + c]) of H channels; gate_g: the conditioning vector [2 H] (or null)
  int ydeint;                               // > 0: output image de-interleaved for a stride-2 consumer - position n at row ymargin + (n >> 1) + (n & 1) ydeint
  int seg2_u, seg2_soff;                    // units >= seg2_u read a SECOND image (same rows per plane, same margin) at byte offset seg2_soff of Xs, one tap of offset 0 per
                                            // chunk: two products over one accumulator; INT_MAX: none
};

// exact-erf GELU (torch F.gelu default), branch-free: erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7 absolute, i.e. ~1e-7 of the
// activation - far inside the fp32 noise of the 768-term sums that feed it; ocml's erff takes two divergent paths per element)
__device__ __forceinline__ float x3s_gelu(float v) {
  const float x = v * 0.70710678118654752440f, ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.f));
  const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * x * x);
  const float erf_abs = fmaf(-poly, e, 1.f);
  return 0.5f * v * (1.f + copysignf(erf_abs, x));
}
// 1 KiB (16 B per lane) from a buffer straight into LDS: address = descriptor base + per-lane voffset + wave-uniform soffset
__device__ __forceinline__ void x3s_dma(__amdgpu_buffer_rsrc_t rs, unsigned char* lds_dst, int voffset, int soffset) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_dst, 16, voffset, soffset, 0, 0);
}
template <int N> __device__ __forceinline__ void x3s_wait_vmcnt() {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

#ifdef RVC_CONV_TIMING
__device__ unsigned long long g_x3s_timing[8];   // [0] workgroups, [1] prologue (to the first barrier), [2] between barriers (operand reads + MFMA issue), [3] DMA wait, [4] barrier, [5] split-K + epilogue, [6] total
void conv_x3s_timing_read(unsigned long long* out8, bool reset) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_x3s_timing), sizeof(unsigned long long) * 8);
  if (reset) { unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_x3s_timing), z, sizeof(z)); }
}
#define XSTICK() ((long long)__builtin_readcyclecounter())
#define XSACC(i, v) do { xst[i] += (v); } while (0)
#else
void conv_x3s_timing_read(unsigned long long* out8, bool) { for (int i = 0; i < 8; ++i) out8[i] = 0; }
#define XSTICK() 0ll
#define XSACC(i, v) do {} while (0)
#endif

// UC = (chunk, tap) units per ring slot = per barrier.  With one unit per barrier a 64 x 64 tile issues 3 MFMAs per wave between two barriers:
// measured (round 4, 768 -> 768 at T = 1599) a unit costs ~600 cycles for 96 cycles of MFMA issue - the wait, the barrier, the DMA issue and the
// operand reads ARE the kernel.  Two units per barrier halve that fixed cost per product; the slot is two units wide, everything else is unchanged.
//
// DIRECT (round 5): the same tiles, accumulators, unit order and epilogue with NO LDS in the reduction at all.  The image formats are the MFMA operand
// layout row by row - lane (li, lh) of a 32 x 32 x 16 block needs the 16 bytes [chunk][hi | lo][half lh][row li] - so a wave can load its operands
// straight from L2 into the registers the MFMA reads: one buffer_load_dwordx4 per operand block (two contiguous 512-byte runs per instruction).
// No DMA pieces (60 - 185 cycles of issue each: MI355X_MICROARCH.md, per-instruction constants), no barrier, no ring slot to publish: the four
// waves of a workgroup are four independent streams, each RS units deep in registers (the compiler counts the plain loads itself).  What is
// given up is the sharing of operand tiles between the waves of a workgroup (a 2 x 2 wave grid loads every byte twice from L2 / L1 instead of
// once into LDS): at N = 1599 - 3198 columns the chip's L2 -> CU bandwidth (~64 B / clk / CU x 256) is abundant, the ring's latency chain is not.
// workgroups per CU the register file admits: ring kernel as measured; register pipeline: accumulators + RS operand stages + ~28 others
constexpr int x3s_wgs(int AM, int AN, int RS, bool DIRECT) {
  if (!DIRECT) return (AM * AN >= 4 && RS != 3) ? 2 : 3;
  const int regs = 16 * AM * AN + 8 * (AM + AN) * RS + 28;
  return regs <= 128 ? 4 : (regs <= 168 ? 3 : (regs <= 256 ? 2 : 1));
}
template <int AM, int AN, int RS, int UC = 1, bool DIRECT = false>
__global__ __launch_bounds__(256, x3s_wgs(AM, AN, RS, DIRECT)) void conv_x3s_kernel(const GemmSArgs p) {
  constexpr int WN = 2, NW = 4;
  constexpr int BM = 64 * AM, BN = 64 * AN;
  constexpr int NPA = BM / 16, NPB = BN / 16, NPW = (NPA + NPB) / NW;      // 1-KiB pieces of a unit: weights, input, per wave
  static_assert((NPA % NW) == 0 && (NPB % NW) == 0, "every wave's i-th piece is of one kind");
  constexpr int aslot = BM * 64, bslot = BN * 64, uslot = aslot + bslot;   // one unit: [hi | lo][half][rows][16 B]
  constexpr int slot = UC * uslot;                                        // one ring slot: UC units
  static_assert(DIRECT || (RS >= 3 && (RS - 2) * NPW * UC <= 63), "ring");
  static_assert(!DIRECT || (UC == 1 && RS >= 2 && RS * 2 * (AM + AN) <= 60), "register pipeline: at most 60 loads in flight");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem3s[];

  const int tid0 = threadIdx.x;
  int lane = tid0 & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;
  // 1-D grid: the S slices of a tile are neighbours in an XCD's run of blocks (the reducer reads same-XCD slabs), tiles column-fastest
  // (an XCD works on a run of row tiles: its weight rows stay in its L2, the activation streams)
  const unsigned total = gridDim.x;
  const unsigned bid = p.xcd_remap ? xcd_tile(blockIdx.x, total) : blockIdx.x;
  const int S = p.ksplit;
  const int tile = (int)(bid / (unsigned)S), ks = (int)(bid - (unsigned)tile * (unsigned)S);
  // An XCD works on one contiguous run of tiles.  Column tiles fastest: the run is a few rows of tiles - the XCD's L2 keeps ITS weight rows and streams the whole
  // activation image (fabric traffic ~ W + 8 X over the eight L2s).  Row tiles fastest: the run is a column range - it keeps its slice of the activations and streams
  // the whole weight image (8 W + X).  The host picks the smaller (row_fast = rows < columns: the 768 -> 768 and 3072 -> 768 layers of HuBERT at T = 1599).
  const int tile_y = p.row_fast ? tile % p.gy : tile / p.gx, tile_x = p.row_fast ? tile / p.gy : tile - tile_y * p.gx;
  const int grp = tile_y / p.rows_pg;                       // (1 group: rows_pg = gy, grp = 0)
  const int co0 = (tile_y - grp * p.rows_pg) * BM, n0 = tile_x * BN;      // first row INSIDE the group
  const int row0 = grp * p.co_g, CoG = p.co_g;              // global row = row0 + m for m < CoG
  const int U1 = p.nunits / S, u0 = ks * U1, U = U1 / UC;            // single units of this slice (a multiple of UC: host), ring steps
#ifdef RVC_CONV_TIMING
  long long xst[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  [[maybe_unused]] const long long xs_begin = XSTICK();
  [[maybe_unused]] long long xs_last = xs_begin;

  // ---- accumulators: slice 0 starts from bias (+ residual unless an activation sits between the sum and the residual)
  const bool r_pre = p.R != nullptr && !(p.act_before_res && p.act != ACT_NONE);
  f32x16 acc[AM][AN];
  if (ks == 0 && (p.bias != nullptr || r_pre)) {
    const __amdgpu_buffer_rsrc_t rrs = make_rsrc(r_pre ? p.R : (const float*)p.Wx, r_pre ? (unsigned)(p.groups * p.co_g) * (unsigned)p.ldR * 4u : 0u);
#pragma unroll
    for (int am = 0; am < AM; ++am)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = co0 + (wm * AM + am) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float bv = (p.bias && m < CoG) ? p.bias[row0 + m] : 0.f;
#pragma unroll
        for (int an = 0; an < AN; ++an) {
          const int n = n0 + (wn * AN + an) * 32 + li;
          acc[am][an][r] = buf_load(rrs, (r_pre && m < CoG && n < p.T) ? ((unsigned)(row0 + m) * (unsigned)p.ldR + (unsigned)n) * 4u : kOOB) + bv;
        }
      }
  } else {
#pragma unroll
    for (int am = 0; am < AM; ++am)
#pragma unroll
      for (int an = 0; an < AN; ++an)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[am][an][r] = 0.f;
  }

  // ---- DMA: piece i of this wave is piece wave + NW i of the unit (pieces 0 .. NPA - 1: weights, then the input).  Buffer addressing: the
  // per-lane offset of a piece never changes, what moves from unit to unit is ONE scalar per operand - weights: the next unit's planes;
  // input: the chunk's planes plus the tap's row offset (im2col by address: a tap is a shifted view of the same image).
  const __amdgpu_buffer_rsrc_t ars = make_rsrc(p.Wx, p.wx_bytes), brs = make_rsrc(p.Xs, p.xs_bytes);
  int voff[NPW], dsto[NPW];
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int pi = wave + NW * i;
    if (i * NW < NPA) {
      constexpr int RBK = BM / 64;
      const int plane = pi / RBK, rb = pi - plane * RBK;
      voff[i] = (plane * p.CoPx + co0 + rb * 64 + lane) * 16;
      dsto[i] = plane * (BM * 16) + rb * 1024;
    } else {
      constexpr int CBK = BN / 64;
      const int pj = pi - NPA, plane = pj / CBK, cb = pj - plane * CBK;
      voff[i] = (int)((plane * p.xsTp + n0 + cb * 64 + lane) * 16);
      dsto[i] = aslot + plane * (BN * 16) + cb * 1024;
    }
  }
  const int KT = p.ktaps;
  const int wstep = p.CoPx * 64, cstep = (int)(p.xsTp * 64);
  int tap = u0 % KT;
  bool seg2 = false;
  const bool arith = KT > 16;                                // long 1-D kernels: offsets by formula instead of the table
  auto tap_off = [&](int t) { return arith ? t * p.tdil - p.tpad : p.toff[t & 15]; };
  int soff_a = (int)((unsigned)grp * p.wg_bytes) + u0 * wstep, soff_c = (grp * p.cig_chunks + u0 / KT) * cstep, soff_b = soff_c + (p.margin + tap_off(tap)) * 16;
  if (u0 >= p.seg2_u) { seg2 = true; soff_b = p.seg2_soff + (u0 - p.seg2_u) * cstep + p.margin * 16; }     // (a K slice that starts inside the second image)
  const int seg2_at = p.seg2_u - u0;                         // unit of this slice at which the second image starts
  if constexpr (DIRECT) {
    // ---- register pipeline: stage i of RS holds the operands of unit g RS + i; the loads of unit u + RS are issued right after unit u's MFMAs
    // (same registers).  Units past the end read out of range (offset bit 31: the hardware range check returns zeros, no memory traffic) so that
    // every group issues the same instructions; their MFMAs are skipped in the last group.
    constexpr int D = RS;
    int va[AM], vb[AN];
#pragma unroll
    for (int am = 0; am < AM; ++am) va[am] = (lh * p.CoPx + co0 + (wm * AM + am) * 32 + li) * 16;
#pragma unroll
    for (int an = 0; an < AN; ++an) vb[an] = (int)((lh * p.xsTp + n0 + (wn * AN + an) * 32 + li) * 16);
    const int lo_a = p.CoPx * 32, lo_b = (int)(p.xsTp * 32);      // hi -> lo plane pair of the same unit
    struct OpsD { u32x4 ah[AM], al[AM], bh[AN], bl[AN]; };
    OpsD st[D];
    int uw = 0;
    // scalar offsets of the next unit.  k = 1 products without a second image (the transformer projections): two adds.  Otherwise the tap offset
    // comes out of a per-lane copy of the table by v_readlane (a dynamically indexed kernel-argument load would put an s_load + lgkmcnt(0) in
    // front of every unit), and tap wrap / second image are selects, not branches.
    const bool simple = KT == 1 && seg2_at > U1;
    const int toff_lane = arith ? 0 : p.toff[lane & 15];
    auto pipeline = [&](auto simple_c) {
    constexpr bool SIMPLE = decltype(simple_c)::value;
    auto tap_off_v = [&](int t) { return arith ? t * p.tdil - p.tpad : __builtin_amdgcn_readlane(toff_lane, t); };
    auto advance = [&]() {
      ++uw;
      soff_a += wstep;
      if constexpr (SIMPLE) { soff_b += cstep; return; }
      const bool in2 = uw > seg2_at, at2 = uw == seg2_at;
      const bool wrap = tap + 1 == KT;
      tap = (in2 || at2) ? tap : (wrap ? 0 : tap + 1);
      soff_c += (!in2 && !at2 && wrap) ? cstep : 0;
      const int b1 = soff_c + (p.margin + tap_off_v(tap)) * 16;
      soff_b = at2 ? p.seg2_soff + p.margin * 16 : (in2 ? soff_b + cstep : b1);
    };
    auto load_unit = [&](OpsD& o, bool guard) {
      const int poison = (guard && uw >= U1) ? (int)0x80000000u : 0;
#pragma unroll
      for (int am = 0; am < AM; ++am) {
        o.ah[am] = __builtin_amdgcn_raw_buffer_load_b128(ars, va[am] | poison, soff_a, 0);
        o.al[am] = __builtin_amdgcn_raw_buffer_load_b128(ars, va[am] | poison, soff_a + lo_a, 0);
      }
#pragma unroll
      for (int an = 0; an < AN; ++an) {
        o.bh[an] = __builtin_amdgcn_raw_buffer_load_b128(brs, vb[an] | poison, soff_b, 0);
        o.bl[an] = __builtin_amdgcn_raw_buffer_load_b128(brs, vb[an] | poison, soff_b + lo_b, 0);
      }
      advance();
    };
    auto mfmas_d = [&](const OpsD& o) {                          // group order of the ring kernel: (hi_w lo_x), (lo_w hi_x), (hi_w hi_x)
#pragma unroll
      for (int am = 0; am < AM; ++am)
#pragma unroll
        for (int an = 0; an < AN; ++an)
          acc[am][an] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, o.ah[am]), __builtin_bit_cast(bf16x8, o.bl[an]), acc[am][an], 0, 0, 0);
#pragma unroll
      for (int am = 0; am < AM; ++am)
#pragma unroll
        for (int an = 0; an < AN; ++an)
          acc[am][an] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, o.al[am]), __builtin_bit_cast(bf16x8, o.bh[an]), acc[am][an], 0, 0, 0);
#pragma unroll
      for (int am = 0; am < AM; ++am)
#pragma unroll
        for (int an = 0; an < AN; ++an)
          acc[am][an] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, o.ah[am]), __builtin_bit_cast(bf16x8, o.bh[an]), acc[am][an], 0, 0, 0);
    };
#pragma unroll
    for (int i = 0; i < D; ++i) load_unit(st[i], true);
    xs_last = XSTICK(); XSACC(1, xs_last - xs_begin);
    const int G = (U1 + D - 1) / D;                              // groups of D units; the last one may be partial
    int g = 0;
    for (; g + 2 < G; ++g) {                                     // the next group is full: no guard in the steady state
#pragma unroll
      for (int i = 0; i < D; ++i) { mfmas_d(st[i]); load_unit(st[i], false); }
    }
    if (g + 1 < G) {
#pragma unroll
      for (int i = 0; i < D; ++i) { mfmas_d(st[i]); load_unit(st[i], true); }
      ++g;
    }
#pragma unroll
    for (int i = 0; i < D; ++i)
      if (g * D + i < U1) mfmas_d(st[i]);
    };
    if (simple) pipeline(std::true_type{}); else pipeline(std::false_type{});
  } else {
  int slw = 0, uw = 0;
  auto issue = [&]() {                                       // next UC units into slot slw; past the end the last unit is requested again
#pragma unroll
    for (int q = 0; q < UC; ++q) {
      unsigned char* base = smem3s + slw * slot + q * uslot;
#pragma unroll
      for (int i = 0; i < NPW; ++i) {
        if (i * NW < NPA) x3s_dma(ars, base + dsto[i], voff[i], soff_a);
        else x3s_dma(brs, base + dsto[i], voff[i], soff_b);
      }
      ++uw;
      if (uw < U1) {
        soff_a += wstep;
        if (uw == seg2_at) { seg2 = true; soff_b = p.seg2_soff + p.margin * 16; }
        else if (KT > 1 && !seg2) {
          ++tap;
          if (tap == KT) { tap = 0; soff_c += cstep; }
          soff_b = soff_c + (p.margin + tap_off(tap)) * 16;
        } else {
          soff_b += cstep;
        }
      }
    }
    slw = slw + 1 == RS ? 0 : slw + 1;
  };

  const int aoff = lh * (BM * 16) + ((wm * AM) * 32 + li) * 16;
  const int boff = aslot + lh * (BN * 16) + ((wn * AN) * 32 + li) * 16;

  // ---- prologue: [bias / residual] units 0 .. RS - 2
#pragma unroll
  for (int i = 0; i < RS - 1; ++i) issue();
  x3s_wait_vmcnt<(RS - 2) * NPW * UC>();                     // slot 0 (younger: slots 1 .. RS - 2)
#pragma unroll
  for (int am = 0; am < AM; ++am)
#pragma unroll
    for (int an = 0; an < AN; ++an) asm volatile("" : "+v"(acc[am][an]));     // (the residual loads are complete from here on)
  lds_barrier();

  // Two operand register sets: the reads of unit u + 1 are issued right after the barrier that publishes its slot and land under the
  // twelve (AM AN 3) MFMAs of unit u - the only stall of a unit is that one wait + barrier.
  struct Ops { u32x4 ah[UC][AM], al[UC][AM], bh[UC][AN], bl[UC][AN]; };
  Ops o0, o1;
  auto read_ops = [&](Ops& o, int slot_i) {
#pragma unroll
    for (int q = 0; q < UC; ++q) {
      const unsigned char* wa = smem3s + slot_i * slot + q * uslot + aoff;
      const unsigned char* xa = smem3s + slot_i * slot + q * uslot + boff;
#pragma unroll
      for (int am = 0; am < AM; ++am) { o.ah[q][am] = *reinterpret_cast<const u32x4*>(wa + am * 512); o.al[q][am] = *reinterpret_cast<const u32x4*>(wa + BM * 32 + am * 512); }
#pragma unroll
      for (int an = 0; an < AN; ++an) { o.bh[q][an] = *reinterpret_cast<const u32x4*>(xa + an * 512); o.bl[q][an] = *reinterpret_cast<const u32x4*>(xa + BN * 32 + an * 512); }
    }
  };
  auto mfmas = [&](const Ops& o) {
#pragma unroll
    for (int q = 0; q < UC; ++q) {
#pragma unroll
      for (int am = 0; am < AM; ++am)
#pragma unroll
        for (int an = 0; an < AN; ++an)
          acc[am][an] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, o.ah[q][am]), __builtin_bit_cast(bf16x8, o.bl[q][an]), acc[am][an], 0, 0, 0);
#pragma unroll
      for (int am = 0; am < AM; ++am)
#pragma unroll
        for (int an = 0; an < AN; ++an)
          acc[am][an] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, o.al[q][am]), __builtin_bit_cast(bf16x8, o.bh[q][an]), acc[am][an], 0, 0, 0);
#pragma unroll
      for (int am = 0; am < AM; ++am)
#pragma unroll
        for (int an = 0; an < AN; ++an)
          acc[am][an] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, o.ah[q][am]), __builtin_bit_cast(bf16x8, o.bh[q][an]), acc[am][an], 0, 0, 0);
    }
  };
  int sl = 0;
  auto body = [&](const Ops& cur, Ops& nxt) {
    // slot u + 1 was requested RS - 2 steps ago; younger: slots u + 2 .. u + RS - 2
    [[maybe_unused]] const long long ta = XSTICK();
    x3s_wait_vmcnt<(RS - 3) * NPW * UC>();
    [[maybe_unused]] const long long tb = XSTICK();
    lds_barrier();
    [[maybe_unused]] const long long tc = XSTICK();
    XSACC(2, ta - xs_last); XSACC(3, tb - ta); XSACC(4, tc - tb); xs_last = tc;
    issue();                                                 // slot u + RS - 1 over the one step u - 1 was read from (those reads fed step u - 1's MFMAs)
    sl = sl + 1 == RS ? 0 : sl + 1;
    read_ops(nxt, sl);
    __builtin_amdgcn_sched_barrier(0);
    mfmas(cur);
    __builtin_amdgcn_sched_barrier(0);
  };
  read_ops(o0, 0);
  xs_last = XSTICK(); XSACC(1, xs_last - xs_begin);
  int u = 0;
  for (; u + 1 < U; u += 2) { body(o0, o1); body(o1, o0); }
  if (u < U) body(o0, o1);
  XSACC(2, XSTICK() - xs_last);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // (the requests past the end: nothing may land in LDS after the workgroup has gone)
  }
  [[maybe_unused]] const long long xs_epi = XSTICK();

  // ---- K split: slabs in register order, write-through; the last slice to arrive sums them in slice order
  if (S > 1) {
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    constexpr int QN = AM * AN * 4;                          // 16-byte granules per lane
    const unsigned slab_bytes = 256u * QN * 16u;
    const __amdgpu_buffer_rsrc_t srs = make_rsrc(p.slabs, (unsigned)(total) * slab_bytes);
    const unsigned lane_off = ((unsigned)wave * QN * 64u + (unsigned)lane) * 16u;
    {
      const unsigned mine = ((unsigned)tile * (unsigned)S + (unsigned)ks) * slab_bytes + lane_off;
#pragma unroll
      for (int am = 0; am < AM; ++am)
#pragma unroll
        for (int an = 0; an < AN; ++an)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const u32x4 v = {__float_as_uint(acc[am][an][4 * q]), __float_as_uint(acc[am][an][4 * q + 1]), __float_as_uint(acc[am][an][4 * q + 2]),
                             __float_as_uint(acc[am][an][4 * q + 3])};
            __builtin_amdgcn_raw_buffer_store_b128(v, srs, (int)(mine + (unsigned)(((am * AN + an) * 4 + q) * 64) * 16u), 0, 16);      // aux 16 = sc1
          }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // EVERY storing wave drains before the ticket
    __syncthreads();
    unsigned* flag = reinterpret_cast<unsigned*>(smem3s);    // (the ring is dead: one LDS object throughout)
    if (tid0 == 0) {
      const unsigned t = __hip_atomic_fetch_add(p.tickets + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (t == (unsigned)S - 1u) __hip_atomic_store(p.tickets + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch on this stream
      *flag = t;
    }
    __syncthreads();
    if (*flag != (unsigned)S - 1u) {
#ifdef RVC_CONV_TIMING
    { const long long te = XSTICK(); XSACC(5, te - xs_epi); XSACC(6, te - xs_begin); XSACC(0, 1);
      if (threadIdx.x == 0) for (int i = 0; i < 8; ++i) atomicAdd(&g_x3s_timing[i], (unsigned long long)xst[i]); }
#endif
      return;
    }
#pragma unroll
    for (int am = 0; am < AM; ++am)
#pragma unroll
      for (int an = 0; an < AN; ++an)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4_t own = {acc[am][an][4 * q], acc[am][an][4 * q + 1], acc[am][an][4 * q + 2], acc[am][an][4 * q + 3]};
          f32x4_t sum = {0.f, 0.f, 0.f, 0.f};
          for (int z = 0; z < S; ++z) {
            f32x4_t v = own;
            if (z != ks) {
              const unsigned off = ((unsigned)tile * (unsigned)S + (unsigned)z) * slab_bytes + lane_off + (unsigned)(((am * AN + an) * 4 + q) * 64) * 16u;
              const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(srs, (int)off, 0, 16);                                   // sc1: past L1, every XCD's view
              v = f32x4_t{__uint_as_float(w[0]), __uint_as_float(w[1]), __uint_as_float(w[2]), __uint_as_float(w[3])};
            }
            sum = z == 0 ? v : sum + v;
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[am][an][4 * q + j] = sum[j];
        }
  }

  // ---- epilogue: activation, late residual, scale; fp32 rows and / or the split image.  The activation is chosen once per launch
  // (wave-uniform branches around straight-line loops); the exact-erf GELU of the reference (F.gelu) is evaluated branch-free.
  const float oscale = p.out_scale;
  if (p.act == ACT_GELU) {
#pragma unroll
    for (int am = 0; am < AM; ++am)
#pragma unroll
      for (int an = 0; an < AN; ++an)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[am][an][r] = x3s_gelu(acc[am][an][r]);
  } else if (p.act != ACT_NONE) {
    const float slope = p.act == ACT_RELU ? 0.f : p.act_slope;
#pragma unroll
    for (int am = 0; am < AM; ++am)
#pragma unroll
      for (int an = 0; an < AN; ++an)
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float v = acc[am][an][r]; acc[am][an][r] = fmaxf(v, v * slope); }
  }
  if (p.R != nullptr && !r_pre) {                              // act(W x + b) + R: the residual after the activation (one batch of loads per accumulator)
    const __amdgpu_buffer_rsrc_t rrs2 = make_rsrc(p.R, (unsigned)(p.groups * p.co_g) * (unsigned)p.ldR * 4u);
#pragma unroll
    for (int am = 0; am < AM; ++am)
#pragma unroll
      for (int an = 0; an < AN; ++an) {
        const int n = n0 + (wn * AN + an) * 32 + li;
        float rv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = co0 + (wm * AM + am) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          rv[r] = buf_load(rrs2, (m < CoG && n < p.T) ? ((unsigned)(row0 + m) * (unsigned)p.ldR + (unsigned)n) * 4u : kOOB);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[am][an][r] += rv[r];
      }
  }
  if (oscale != 1.f) {
#pragma unroll
    for (int am = 0; am < AM; ++am)
#pragma unroll
      for (int an = 0; an < AN; ++an)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[am][an][r] *= oscale;
  }
  if (p.padw > 0) {                                            // padded 2-D rows: the two pad columns stay zero (they are the next layer's zero padding)
#pragma unroll
    for (int an = 0; an < AN; ++an) {
      const unsigned n = (unsigned)(n0 + (wn * AN + an) * 32 + li);
      const unsigned q = __umulhi(n, p.padmagic), w = n - q * (unsigned)p.padw;      // exact for n < 2^31 / padw (host)
      if (w == 0u || w == (unsigned)p.padw - 1u) {
#pragma unroll
        for (int am = 0; am < AM; ++am)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[am][an][r] = 0.f;
      }
    }
  }
  const bool vt_tile = co0 >= p.vt_row0;                       // (tile-uniform: vt_row0 is a multiple of the tile height)
  if constexpr (!DIRECT) {
  if (vt_tile) {
    // The v rows of a fused q | k | v projection: the attention's P V product reduces over keys, so it wants V with 8 consecutive POSITIONS per 16-byte row.  A lane
    // holds one position (column) of 16 channels; the 32 x 32 block goes through a wave-private 4 KiB of the (now dead) ring as fp32 [channel][position] and comes
    // back as (channel, 8 positions) items, two per lane.  Positions past T are written as zeros (keys past the end), like conv_x3s_run_swapped's zero_tail.
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    unsigned char* tl = smem3s + 1024 + wave * 4096;
    const int Tc16 = (p.T + 15) & ~15;
#pragma unroll
    for (int am = 0; am < AM; ++am)
#pragma unroll
      for (int an = 0; an < AN; ++an) {
        const int nb = n0 + (wn * AN + an) * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int jj = (r & 3) + 8 * (r >> 2) + 4 * lh;
          *reinterpret_cast<float*>(tl + (jj * 32 + li) * 4) = (nb + li < p.T) ? acc[am][an][r] : 0.f;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int item = lane + 64 * it, jj = item >> 2, tg = item & 3;
          const f32x4v va = *reinterpret_cast<const f32x4v*>(tl + (jj * 32 + tg * 8) * 4), vb = *reinterpret_cast<const f32x4v*>(tl + (jj * 32 + tg * 8 + 4) * 4);
          u32x4 hi, lo;
          unsigned h_, l_;
          split2(va[0], va[1], h_, l_); hi[0] = h_; lo[0] = l_;
          split2(va[2], va[3], h_, l_); hi[1] = h_; lo[1] = l_;
          split2(vb[0], vb[1], h_, l_); hi[2] = h_; lo[2] = l_;
          split2(vb[2], vb[3], h_, l_); hi[3] = h_; lo[3] = l_;
          const int t0 = nb + tg * 8, j = co0 + (wm * AM + am) * 32 + jj - p.vt_row0;
          if (t0 < Tc16 && j < p.vt_rows) {
            unsigned char* row = p.Vt + ((((long long)(t0 >> 4)) * 4 + ((t0 >> 3) & 1)) * p.vtTp + kSplitMargin + j) * 16;
            *reinterpret_cast<u32x4*>(row) = hi;
            *reinterpret_cast<u32x4*>(row + p.vtTp * 32) = lo;
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
  }
  }
  if (p.Y && !vt_tile) {
    const __amdgpu_buffer_rsrc_t yrs = make_rsrc(p.Y, (unsigned)(p.groups * p.co_g) * (unsigned)p.ldY * 4u);
#pragma unroll
    for (int am = 0; am < AM; ++am)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = co0 + (wm * AM + am) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
#pragma unroll
        for (int an = 0; an < AN; ++an) {
          const int n = n0 + (wn * AN + an) * 32 + li;
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[am][an][r]), yrs, (int)((m < CoG && n < p.T) ? ((unsigned)(row0 + m) * (unsigned)p.ldY + (unsigned)n) * 4u : kOOB), 0, 0);
        }
      }
  }
  if (p.Ys && p.gate_h > 0) {
    // in_layer -> fused_add_tanh_sigmoid_multiply (reference lib/infer_pack/modules.py WN.forward, commons.py) without the 2 H-row tensor ever reaching memory: registers
    // r and r + 8 of a lane are the tanh and the sigmoid row of one channel (the host packed the rows that way); the product leaves as ONE 16-channel chunk of the image per
    // 32-row block.
    const int H = p.gate_h;
#pragma unroll
    for (int am = 0; am < AM; ++am) {
      const int blk = (co0 + (wm * AM + am) * 32) >> 5;          // 16-channel chunk of the gate's output
      float gt[8], gsg[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int c = 16 * blk + (r & 3) + 8 * (r >> 2) + 4 * lh;
        gt[r] = (p.gate_g && c < H) ? p.gate_g[c] : 0.f; gsg[r] = (p.gate_g && c < H) ? p.gate_g[H + c] : 0.f;
      }
#pragma unroll
      for (int an = 0; an < AN; ++an) {
        const int n = n0 + (wn * AN + an) * 32 + li;
        float v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          // hardware exp2 / reciprocal (1 ulp each), as the GRU scan's gates: tanh(x) = 1 - 2 / (2^(2 x log2 e) + 1), sigmoid(x) = 1 / (1 + 2^(-x log2 e))
          constexpr float kL2E = 1.44269504088896340736f;
          const float ta = acc[am][an][r] + gt[r], sa = acc[am][an][r + 8] + gsg[r];
          const float th = 1.f - 2.f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(2.f * kL2E * ta) + 1.f);
          v[r] = th * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-kL2E * sa));
        }
        unsigned hA[2], lA[2], hB[2], lB[2];
#pragma unroll
        for (int e2 = 0; e2 < 2; ++e2) {
          split2(v[2 * e2], v[2 * e2 + 1], hA[e2], lA[e2]);
          split2(v[4 + 2 * e2], v[5 + 2 * e2], hB[e2], lB[e2]);
        }
        u32x4 hi, lo;
#pragma unroll
        for (int e2 = 0; e2 < 2; ++e2) {
          typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
          const u32x2_t sh = __builtin_amdgcn_permlane32_swap(hA[e2], hB[e2], false, false);
          const u32x2_t sl2 = __builtin_amdgcn_permlane32_swap(lA[e2], lB[e2], false, false);
          hi[e2] = sh.x; hi[2 + e2] = sh.y; lo[e2] = sl2.x; lo[2 + e2] = sl2.y;
        }
        if (n < p.T && 16 * blk < H) {
          unsigned char* row = p.Ys + (((long long)blk * 4 + lh) * p.ysTp + (long long)n + p.ymargin) * 16;
          *reinterpret_cast<u32x4*>(row) = hi;
          *reinterpret_cast<u32x4*>(row + p.ysTp * 32) = lo;
        }
      }
    }
  } else if (p.Ys && !vt_tile) {
    if (p.zero_tail) {
#pragma unroll
      for (int am = 0; am < AM; ++am)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = co0 + (wm * AM + am) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (m >= CoG) {
#pragma unroll
            for (int an = 0; an < AN; ++an) acc[am][an][r] = 0.f;
          }
        }
    }
    // a lane holds 4 + 4 channels of each 16-channel chunk of its column; v_permlane32_swap trades quads with the lane 32 away so that
    // every lane owns one 16-B row of a half-plane (see ysplit_epilogue, conv_x3_dev.h)
#pragma unroll
    for (int am = 0; am < AM; ++am)
#pragma unroll
      for (int an = 0; an < AN; ++an) {
        const int n = n0 + (wn * AN + an) * 32 + li;
        const int mb = co0 + (wm * AM + am) * 32;
        const long long pos = p.ydeint > 0 ? (long long)p.ymargin + (n >> 1) + (long long)(n & 1) * p.ydeint : (long long)n + p.ymargin;
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {
          unsigned hA[2], lA[2], hB[2], lB[2];
#pragma unroll
          for (int e2 = 0; e2 < 2; ++e2) {
            split2(acc[am][an][8 * g2 + 2 * e2], acc[am][an][8 * g2 + 2 * e2 + 1], hA[e2], lA[e2]);
            split2(acc[am][an][8 * g2 + 4 + 2 * e2], acc[am][an][8 * g2 + 5 + 2 * e2], hB[e2], lB[e2]);
          }
          u32x4 hi, lo;
#pragma unroll
          for (int e2 = 0; e2 < 2; ++e2) {
            typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
            const u32x2_t sh = __builtin_amdgcn_permlane32_swap(hA[e2], hB[e2], false, false);
            const u32x2_t sl2 = __builtin_amdgcn_permlane32_swap(lA[e2], lB[e2], false, false);
            hi[e2] = sh.x; hi[2 + e2] = sh.y; lo[e2] = sl2.x; lo[2 + e2] = sl2.y;
          }
          if (n < p.T && mb + 16 * g2 < CoG) {
            const long long chunk = ((row0 + mb) >> 4) + g2;
            unsigned char* row = p.Ys + ((chunk * 4 + lh) * p.ysTp + pos) * 16;
            *reinterpret_cast<u32x4*>(row) = hi;
            *reinterpret_cast<u32x4*>(row + p.ysTp * 32) = lo;
          }
        }
      }
  }
#ifdef RVC_CONV_TIMING
  { const long long te = XSTICK(); XSACC(5, te - xs_epi); XSACC(6, te - xs_begin); XSACC(0, 1);
    if (threadIdx.x == 0) for (int i = 0; i < 8; ++i) atomicAdd(&g_x3s_timing[i], (unsigned long long)xst[i]); }
#endif
}

// ---------------------------------------------------------------------------- fp32 [C][T] <-> split image (producers without an image epilogue, tests)
// One thread = one 16-byte row (8 channels of one position) of the hi and of the lo plane.
__global__ __launch_bounds__(256) void split_image_kernel(const float* __restrict__ X, long long ldX, int C, int T, unsigned char* __restrict__ img, long long tp) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int hp = blockIdx.y;                                  // chunk * 2 + half
  if (t >= T) return;
  const float* x = X + (long long)(hp * 8) * ldX + t;
  u32x4 hi, lo;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float a = (hp * 8 + 2 * j) < C ? x[(long long)(2 * j) * ldX] : 0.f, b = (hp * 8 + 2 * j + 1) < C ? x[(long long)(2 * j + 1) * ldX] : 0.f;
    unsigned h_, l_;
    split2(a, b, h_, l_);
    hi[j] = h_; lo[j] = l_;
  }
  const int chunk = hp >> 1, half = hp & 1;
  unsigned char* row = img + (((long long)chunk * 4 + half) * tp + kSplitMargin + t) * 16;
  *reinterpret_cast<u32x4*>(row) = hi;
  *reinterpret_cast<u32x4*>(row + tp * 32) = lo;
}
__global__ __launch_bounds__(256) void unsplit_image_kernel(const unsigned char* __restrict__ img, long long tp, int C, int T, float* __restrict__ Y, long long ldY) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int hp = blockIdx.y;
  if (t >= T) return;
  const int chunk = hp >> 1, half = hp & 1;
  const unsigned char* row = img + (((long long)chunk * 4 + half) * tp + kSplitMargin + t) * 16;
  const u32x4 hi = *reinterpret_cast<const u32x4*>(row), lo = *reinterpret_cast<const u32x4*>(row + tp * 32);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = hp * 8 + 2 * j;
    if (c < C) Y[(long long)c * ldY + t] = __uint_as_float(hi[j] << 16) + __uint_as_float(lo[j] << 16);
    if (c + 1 < C) Y[(long long)(c + 1) * ldY + t] = __uint_as_float(hi[j] & 0xffff0000u) + __uint_as_float(lo[j] & 0xffff0000u);
  }
}
void split_image_from_f32(hipStream_t s, const float* X, long long ldX, int C, int T, unsigned char* img, long long tp) {
  // (channels past C inside the last 16-channel chunk are written as zeros)
  hipLaunchKernelGGL(split_image_kernel, dim3((T + 255) / 256, (C + 15) / 16 * 2), dim3(256), 0, s, X, ldX, C, T, img, tp);
}
// x [C][T][M] fp32 (per channel: T rows of M contiguous values, M a multiple of 8) -> the image of the (C M) x T tensor whose channel index is
// c M + m (RMVPE: the 3 x 128 output of its last convolution flattened per frame, lib/rmvpe.py:356): 8 consecutive channels are 32 contiguous bytes
__global__ __launch_bounds__(256) void split_image_tm_kernel(const float* __restrict__ x, int C, int T, int M, unsigned char* __restrict__ img, long long tp) {
  typedef float f32x4_t __attribute__((ext_vector_type(4)));
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int hp = blockIdx.y;                                  // group of 8 channels = chunk * 2 + half
  if (t >= T) return;
  const int f0 = hp * 8, c = f0 / M, m0 = f0 - c * M;
  u32x4 hi, lo;
  if (c < C) {
    const float* src = x + ((long long)c * T + t) * M + m0;
    const f32x4_t a = *reinterpret_cast<const f32x4_t*>(src), b = *reinterpret_cast<const f32x4_t*>(src + 4);
    unsigned h_, l_;
    split2(a[0], a[1], h_, l_); hi[0] = h_; lo[0] = l_;
    split2(a[2], a[3], h_, l_); hi[1] = h_; lo[1] = l_;
    split2(b[0], b[1], h_, l_); hi[2] = h_; lo[2] = l_;
    split2(b[2], b[3], h_, l_); hi[3] = h_; lo[3] = l_;
  } else {
    hi = u32x4{0u, 0u, 0u, 0u}; lo = hi;
  }
  unsigned char* row = img + (((long long)(hp >> 1) * 4 + (hp & 1)) * tp + kSplitMargin + t) * 16;
  *reinterpret_cast<u32x4*>(row) = hi;
  *reinterpret_cast<u32x4*>(row + tp * 32) = lo;
}
void split_image_from_tm(hipStream_t s, const float* x, int C, int T, int M, unsigned char* img, long long tp) {
  RVC_REQUIRE((M & 7) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0, "split_image_from_tm: rows of a multiple of 8 values, 16-byte aligned");
  hipLaunchKernelGGL(split_image_tm_kernel, dim3((T + 255) / 256, (C * M + 15) / 16 * 2), dim3(256), 0, s, x, C, T, M, img, tp);
}
// fp32 [C][T] <-> DE-INTERLEAVED image (split_geom_s2: position t at row margin + (t >> 1) + (t & 1) H): tests of the stride-2 path; in the models the producers' epilogues write it
__global__ __launch_bounds__(256) void split_image_deint_kernel(const float* __restrict__ X, long long ldX, int C, int T, unsigned char* __restrict__ img, long long tp, int H) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int hp = blockIdx.y;
  if (t >= T) return;
  const float* x = X + (long long)(hp * 8) * ldX + t;
  u32x4 hi, lo;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float a = (hp * 8 + 2 * j) < C ? x[(long long)(2 * j) * ldX] : 0.f, b = (hp * 8 + 2 * j + 1) < C ? x[(long long)(2 * j + 1) * ldX] : 0.f;
    unsigned h_, l_;
    split2(a, b, h_, l_);
    hi[j] = h_; lo[j] = l_;
  }
  unsigned char* row = img + (((long long)(hp >> 1) * 4 + (hp & 1)) * tp + kSplitMargin + (t >> 1) + (long long)(t & 1) * H) * 16;
  *reinterpret_cast<u32x4*>(row) = hi;
  *reinterpret_cast<u32x4*>(row + tp * 32) = lo;
}
__global__ __launch_bounds__(256) void unsplit_image_deint_kernel(const unsigned char* __restrict__ img, long long tp, int H, int C, int T, float* __restrict__ Y, long long ldY) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int hp = blockIdx.y;
  if (t >= T) return;
  const unsigned char* row = img + (((long long)(hp >> 1) * 4 + (hp & 1)) * tp + kSplitMargin + (t >> 1) + (long long)(t & 1) * H) * 16;
  const u32x4 hi = *reinterpret_cast<const u32x4*>(row), lo = *reinterpret_cast<const u32x4*>(row + tp * 32);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = hp * 8 + 2 * j;
    if (c < C) Y[(long long)c * ldY + t] = __uint_as_float(hi[j] << 16) + __uint_as_float(lo[j] << 16);
    if (c + 1 < C) Y[(long long)(c + 1) * ldY + t] = __uint_as_float(hi[j] & 0xffff0000u) + __uint_as_float(lo[j] & 0xffff0000u);
  }
}
void split_image_deint_from_f32(hipStream_t s, const float* X, long long ldX, int C, int T, unsigned char* img, long long tp, int H) {
  hipLaunchKernelGGL(split_image_deint_kernel, dim3((T + 255) / 256, (C + 15) / 16 * 2), dim3(256), 0, s, X, ldX, C, T, img, tp, H);
}
void split_image_deint_to_f32(hipStream_t s, const unsigned char* img, long long tp, int H, int C, int T, float* Y, long long ldY) {
  hipLaunchKernelGGL(unsplit_image_deint_kernel, dim3((T + 255) / 256, (C + 15) / 16 * 2), dim3(256), 0, s, img, tp, H, C, T, Y, ldY);
}
void split_image_to_f32(hipStream_t s, const unsigned char* img, long long tp, int C, int T, float* Y, long long ldY) {
  hipLaunchKernelGGL(unsplit_image_kernel, dim3((T + 255) / 256, (C + 15) / 16 * 2), dim3(256), 0, s, img, tp, C, T, Y, ldY);
}

// ---------------------------------------------------------------------------- host side
template <int AM, int AN, int RS, int UC = 1>
static void launch_x3s(const GemmSArgs& a, unsigned blocks, hipStream_t s) {
  auto kern = conv_x3s_kernel<AM, AN, RS, UC>;
  constexpr size_t lds = (size_t)RS * UC * (64 * AM + 64 * AN) * 64;
  RVC_ALLOW_BIG_LDS(kern);
  conv_launch(kern, dim3(blocks), dim3(256), lds, s, a);
}
// The register-direct reduction loop (round 5: bit-identical to the ring, neutral in the pipeline, profiles/r5_exp_x3s_direct.txt) exists in
// -DRVC_EXPERIMENTS builds only: the product library neither instantiates nor reaches it.
// decided by x3s_plan for the launch that follows on this thread - ONE launch is planned and dispatched under one mode
static thread_local bool t_plan_direct = false;
#ifdef RVC_EXPERIMENTS
// register-direct variant: D units deep in registers, 1 KiB of LDS (the split-K ticket word)
template <int AM, int AN, int D>
static void launch_x3s_direct(const GemmSArgs& a, unsigned blocks, hipStream_t s) {
  hipLaunchKernelGGL((conv_x3s_kernel<AM, AN, D, 1, true>), dim3(blocks), dim3(256), 1024, s, a);
}
// 0: by shape (x3s_plan), 1: the LDS-ring kernel always, 2: the register-direct kernel always (rvc_debug_set_x3s_mode / RVC_X3S_MODE)
static std::atomic<int> g_x3s_mode{-1};
void conv_x3s_set_mode(int m) { g_x3s_mode.store(m, std::memory_order_relaxed); }
static int x3s_mode() {
  int m = g_x3s_mode.load(std::memory_order_relaxed);
  if (m < 0) { m = exp_int("RVC_X3S_MODE", 0); g_x3s_mode.store(m, std::memory_order_relaxed); }
  return m;
}
#else
void conv_x3s_set_mode(int) {}
static int x3s_mode() { return 1; }
#endif
static void x3s_dispatch(const GemmSArgs& a, int AM, int AN, unsigned blocks, hipStream_t s) {
#ifdef RVC_EXPERIMENTS
  if (t_plan_direct) {
    static const int d_env = exp_int("RVC_X3S_D", 0);      // pipeline depth override (units in flight per wave)
    if (AM == 2 && AN == 2) { if (d_env == 3) launch_x3s_direct<2, 2, 3>(a, blocks, s); else if (d_env == 5) launch_x3s_direct<2, 2, 5>(a, blocks, s); else launch_x3s_direct<2, 2, 4>(a, blocks, s); }
    else if (AM == 2 && AN == 1) { if (d_env == 3) launch_x3s_direct<2, 1, 3>(a, blocks, s); else if (d_env == 8) launch_x3s_direct<2, 1, 8>(a, blocks, s); else launch_x3s_direct<2, 1, 4>(a, blocks, s); }
    else if (AM == 1 && AN == 2) { if (d_env == 3) launch_x3s_direct<1, 2, 3>(a, blocks, s); else if (d_env == 8) launch_x3s_direct<1, 2, 8>(a, blocks, s); else launch_x3s_direct<1, 2, 4>(a, blocks, s); }
    else { if (d_env == 4) launch_x3s_direct<1, 1, 4>(a, blocks, s); else if (d_env == 12) launch_x3s_direct<1, 1, 12>(a, blocks, s); else launch_x3s_direct<1, 1, 7>(a, blocks, s); }
    return;
  }
#endif
  // ring depth: RVC_X3S_RS = 3 / 4 / 6 for every launch; default 4, and 3 for the 128 x 128 tile on grids of three workgroups per CU and deep reductions - a ring of
  // three slots is 48 KiB and the kernel then fits 168 VGPRs, so THREE workgroups share a CU (MDX23C's 3x3 layers at 128 / 256 channels: 263 -> 235, 250 -> 217,
  // 526 -> 451 us in one call; smaller grids and the transformer projections lose 5 - 10 % with three slots and keep four)
  static const int rs_force = exp_int("RVC_X3S_RS", 0);
  const int rs_env = rs_force ? rs_force : ((AM == 2 && AN == 2 && blocks >= 768u && a.nunits / a.ksplit >= 64) ? 3 : 4);
  // units per barrier (RVC_X3S_UC=2: two, where the slice's unit count is even and the tile is small).  Measured in round 4 and NOT the default: the
  // average launch stays at 20.0 us (20.3 with one unit per barrier) - the K loop of these 300 - 600-workgroup grids is bound by the L2 -> LDS
  // latency per ring step, not by the barrier - and the doubled LDS footprint costs the three-lane bench 2.5 % (2095 -> 2040 xRT, same box).
#ifdef RVC_EXPERIMENTS
  static const int uc_env = exp_int("RVC_X3S_UC", 1);
  const bool two = uc_env >= 2 && rs_env != 3 && rs_env < 6 && ((a.nunits / a.ksplit) & 1) == 0 && (a.nunits / a.ksplit) >= 8 && AM * AN <= 2;
  if (two) {
    if (AM == 2 && AN == 1) launch_x3s<2, 1, 4, 2>(a, blocks, s);
    else if (AM == 1 && AN == 2) launch_x3s<1, 2, 4, 2>(a, blocks, s);
    else launch_x3s<1, 1, 4, 2>(a, blocks, s);
    return;
  }
  if (rs_env >= 6) {
    if (AM == 2 && AN == 2) launch_x3s<2, 2, 6>(a, blocks, s); else if (AM == 2 && AN == 1) launch_x3s<2, 1, 6>(a, blocks, s);
    else if (AM == 1 && AN == 2) launch_x3s<1, 2, 6>(a, blocks, s); else launch_x3s<1, 1, 6>(a, blocks, s);
    return;
  }
#endif
  if (AM == 2 && AN == 2) { if (rs_env == 3) launch_x3s<2, 2, 3>(a, blocks, s); else launch_x3s<2, 2, 4>(a, blocks, s); }
  else if (AM == 2 && AN == 1) { if (rs_env == 3) launch_x3s<2, 1, 3>(a, blocks, s); else launch_x3s<2, 1, 4>(a, blocks, s); }
  else if (AM == 1 && AN == 2) { if (rs_env == 3) launch_x3s<1, 2, 3>(a, blocks, s); else launch_x3s<1, 2, 4>(a, blocks, s); }
  else { if (rs_env == 3) launch_x3s<1, 1, 3>(a, blocks, s); else launch_x3s<1, 1, 4>(a, blocks, s); }
}

// Tile and K split for an M x N x K problem: enough workgroups for two per CU (two waves per SIMD from different tiles cover each other's
// barriers and operand reads), tiles as large as that allows (L2 -> LDS bytes per MFMA fall with the tile), the deep reductions cut along K.
static thread_local int t_force_s = 0, t_force_am = 0, t_force_an = 0;
void conv_x3s_force(int ksplit, int am, int an) { t_force_s = ksplit; t_force_am = am; t_force_an = an; }
static void x3s_plan(int M, int N, int units, int& AM, int& AN, int& S, int groups = 1, int ktaps = 1, bool swapped = false) {
  // Which reduction loop (measured on MI355X, tools/bench_gemm.py, profiles/r5_bench_gemm_direct.txt; cold weights, both loops at their best tile / split):
  //   3 x 3 over RMVPE's deep levels, 512 ch @ 606 positions: ring 27.4 us (split 6) vs direct 23.0 (split 3); 256 ch @ 2020: 21.7 (split 4) vs 18.9 (split 2)
  //   k = 1, <= 768 rows: 768 -> 768 17.0 vs 16.5 (64 x 128), 512 -> 768 13.3 vs 12.6, 192 -> 576 @ 3198 9.2 vs 8.8, 192 -> 192 7.3 vs 6.6
  //   k = 1, wide: 768 -> 3072 31.8 vs 38.3, 3072 -> 768 37.5 vs 40.4, 768 -> 2304 25.6 vs 24.3 (within the spread) -> ring
  // i.e. the register pipeline wins where a launch is a few hundred small tiles of a short or tap-rich reduction (nothing to share through LDS that
  // L2 does not deliver as fast, and no DMA / barrier chain per unit); the ring wins where operand sharing halves the L2 bytes of a wide product.
  // MDX23C's large planes (N >= 64 K positions) keep the ring and its tuned tiles.
  static const int d_auto = exp_int("RVC_X3S_DIRECT_AUTO", 0);      // (default OFF: in the pipeline the two loops measure the same - profiles/r5_exp_x3s_direct.txt)
  const bool conv_small = ktaps >= 3 && ktaps <= 16 && groups == 1 && N <= 4096 && units >= 36;
  const bool gemm_small = ktaps == 1 && groups == 1 && !swapped && M <= 768 && N <= 8192;
  const int mode_now = x3s_mode();                     // latched: the launch is planned and dispatched under this one value
  t_plan_direct = d_auto && mode_now != 1 && (conv_small || gemm_small);
  if (mode_now == 2) t_plan_direct = true;
  static const int f_am = exp_int("RVC_X3S_AM", 0), f_an = exp_int("RVC_X3S_AN", 0);
  static const int f_s = exp_int("RVC_X3S_SPLIT", 0);
  static const int target = exp_int("RVC_X3S_BLK", 440);
  auto tiles = [&](int am, int an) { return (long long)groups * ((M + 64 * am - 1) / (64 * am)) * ((N + 64 * an - 1) / (64 * an)); };
  // measured on MI355X at N = 1599 (tools/bench_gemm.py, profiles/r3b_bench_gemm.txt): 768 -> 3072 128 x 64 35 us (64 x 128 the same, 128 x 128 38),
  // 768 -> 2304 128 x 64 28 us, 768 -> 768 64 x 64 16.3 us un-split (17.6 split in two), 3072 -> 768 64 x 64 split in two 37.8 us (128 x 64 in three 39.0)
  AM = 2; AN = 2;
  if (t_plan_direct && mode_now != 2) {
    // register pipeline: 64 x 128 tiles for the k = 1 products while ~140 of them remain (the column operand is the one re-read per row tile), 64 x 64
    // otherwise; K split to ~220 workgroups (half the ring's target: a wave streams deeper on its own)
    auto tl = [&](int am, int an) { return (long long)((M + 64 * am - 1) / (64 * am)) * ((N + 64 * an - 1) / (64 * an)); };
    AM = 1; AN = (ktaps == 1 && tl(1, 2) >= 140) ? 2 : 1;
    if (f_am && groups == 1) AM = f_am;
    if (f_an) AN = f_an;
    if ((t_force_am == 1 || t_force_am == 2)) AM = t_force_am;
    if (t_force_an == 1 || t_force_an == 2) AN = t_force_an;
    S = 1;
    const long long nt = tl(AM, AN);
    for (int c : {2, 3, 4, 6, 8}) {
      if (nt * S >= 220) break;
      if (units % c == 0 && units / c >= 32) S = c;
    }
    if (f_s) S = (units % f_s == 0) ? f_s : 1;
    if (t_force_s > 0) S = (units % t_force_s == 0 && units / t_force_s >= 4) ? t_force_s : 1;
    return;
  }
  // (deep reductions - MDX23C's 3 x 3 layers at 384+ channels, 216+ units - keep the 128 x 128 tile from one tile per CU on: 384 -> 384 over 64 x 258 positions
  //  143 us on 387 tiles against 168 on 774 of 128 x 64; the transformer projections above have fewer tiles or shorter reductions and are not touched)
  const bool deep = units >= 128 && groups == 1 && tiles(2, 2) >= 256;
  if (!deep && tiles(AM, AN) < target) AN = 1;
  if (!deep && tiles(AM, AN) < target) AM = 1;
  if (groups > 1) AM = 1;                                    // (a group's rows fit one 64-row tile: pack_x3_grouped)
  if (f_am && groups == 1) AM = f_am;
  if (f_an) AN = f_an;
  if ((t_force_am == 1 || t_force_am == 2) && groups == 1) AM = t_force_am;
  if (t_force_an == 1 || t_force_an == 2) AN = t_force_an;
  S = 1;
  const long long nt = tiles(AM, AN);
  static const int min_units = exp_int("RVC_X3S_MINUNITS", 32);
  for (int c : {2, 3, 4, 6, 8, 12, 16}) {
    if (nt * S >= target) break;
    if (units % c == 0 && units / c >= min_units) S = c;      // (a slice shorter than K = 512 does not pay for its slab round trip)
  }
  if (f_s) S = (units % f_s == 0) ? f_s : 1;
  if (t_force_s > 0) S = (units % t_force_s == 0 && units / t_force_s >= 4) ? t_force_s : 1;
}

bool conv_x3s_eligible(const ConvLayer& L) {
  if (!conv_x3_enabled() || L.Wx_ == nullptr || (L.Ci & 15) != 0 || L.Ci * L.ktaps < 32 || L.tconv_u != 0 || L.up2 != 0) return false;
  if (L.mode == 2) return L.ktaps <= 16;                                      // 3 x 3 (or KH x KW) over a padded 2-D image: the caller supplies the geometry
  if (L.mode != 1 || L.stride != 1 || L.pad > kSplitMargin || (L.k - 1) * L.dil - L.pad > kSplitMargin) return false;
  if (L.groups > 1) return (L.Co & 15) == 0 && (L.CoPx & 63) == 0 && L.CoPx <= 64;        // grouped: one 64-row tile per group (pack_x3_grouped)
  return L.k <= 16 && L.pad == (L.k - 1) / 2 * L.dil && (L.k & 1) == 1;        // 1-D "same" convolution, taps as row offsets of the image
}

SplitGeom split_geom_s2(int k, long long Tin) {
  SplitGeom g; g.ktaps = k; g.s2_h = split_s2_h(Tin); g.margin = kSplitMargin; g.padw = 0;
  RVC_REQUIRE(k >= 1 && k <= 16, "split_geom_s2: at most 16 taps");
  for (int t = 0; t < k; ++t) g.toff[t] = (t & 1) * g.s2_h + (t >> 1);
  return g;
}
bool conv_x3s_s2_eligible(const ConvLayer& L) {
  return conv_x3_enabled() && L.Wx_ != nullptr && L.mode == 1 && L.stride == 2 && L.pad == 0 && L.dil == 1 && L.groups == 1 && L.tconv_u == 0 && L.up2 == 0 &&
         (L.Ci & 15) == 0 && (L.Co & 15) == 0 && L.k >= 2 && L.k <= 16 && L.seg2_chunks == 0;
}

SplitGeom split_geom_2d(int Wd, int KH, int KW, int PH, int PWL) {
  SplitGeom g; g.padw = Wd + 2; g.ktaps = KH * KW;
  RVC_REQUIRE(g.ktaps <= 16 && PWL <= 1 && KW - 1 - PWL <= 1, "split_geom_2d: at most one pad column on each side");
  for (int kh = 0; kh < KH; ++kh) for (int kw = 0; kw < KW; ++kw) g.toff[kh * KW + kw] = (kh - PH) * g.padw + (kw - PWL);
  int maxoff = 0;
  for (int t = 0; t < g.ktaps; ++t) maxoff = std::max(maxoff, std::abs(g.toff[t]));
  g.margin = kSplitMargin;
  while (g.margin < maxoff) g.margin += 64;
  return g;
}

void conv_x3s_run(const ConvLayer& L, hipStream_t s, const unsigned char* Xs, long long xsTp, int T, float* Y, long long ldY, const ConvEpilogue& e,
                  const SplitGeom* geom) {
  const bool s2 = geom != nullptr && geom->s2_h > 0;
  RVC_REQUIRE(s2 ? conv_x3s_s2_eligible(L) : conv_x3s_eligible(L), "conv_x3s_run: layer without a bf16x3 weight image or not a stride-1 'same' (stride-2 'valid' with a de-interleaved image) geometry");
  SplitGeom g1;
  if (!geom) {
    RVC_REQUIRE(L.mode == 1, "conv_x3s_run: a 2-D layer needs its padded-image geometry");
    g1.ktaps = L.k; g1.margin = kSplitMargin; g1.padw = 0;
    for (int t = 0; t < L.k && t < 16; ++t) g1.toff[t] = t * L.dil - L.pad;      // (longer kernels: the offsets follow tap * dil - pad in the kernel)
    geom = &g1;
  }
  RVC_REQUIRE(geom->ktaps == L.ktaps || (L.mode == 1 && geom->ktaps == L.k), "conv_x3s_run: geometry and layer disagree about the taps");
  if (s2) {
    // (T = output positions; the taps reach rows margin .. margin + s2_h + T of a plane - all inside the producer's rows or the untouched gap between the planes, whose columns are not stored)
    RVC_REQUIRE(Xs != nullptr && xsTp >= geom->margin + 2LL * geom->s2_h + 704 && (long long)T + 1 <= geom->s2_h, "conv_x3s_run: de-interleaved input image missing or too short");
  } else {
  int maxoff = L.mode == 1 ? std::max(L.pad, (L.k - 1) * L.dil - L.pad) : 0;
  for (int t = 0; t < geom->ktaps && t < 16; ++t) maxoff = std::max(maxoff, std::abs(geom->toff[t]));
  RVC_REQUIRE(geom->margin >= maxoff && geom->margin >= kSplitMargin, "conv_x3s_run: image margin smaller than the largest tap offset");
  RVC_REQUIRE(Xs != nullptr && xsTp >= geom->margin + T + 704, "conv_x3s_run: split-resident input image missing or too short (margin + T + 704 rows per plane)");
  }
  RVC_REQUIRE(Y != nullptr || e.ys_out != nullptr, "conv_x3s_run: no output");
  RVC_REQUIRE(e.pre_act == ACT_NONE && !e.accumulate && !e.tout_limit && !e.xs_in, "conv_x3s_run: unsupported epilogue option");
  RVC_REQUIRE(e.act == ACT_NONE || e.act == ACT_LRELU || e.act == ACT_RELU || e.act == ACT_GELU, "conv_x3s_run: activation must be identity / (leaky) ReLU / GELU");
  RVC_REQUIRE(!e.ys_out || ((e.ys_deint_h > 0 ? (e.ys_tp >= geom->margin + 2LL * e.ys_deint_h + 704 && (T + 1) / 2 + 1 <= e.ys_deint_h) : e.ys_tp >= geom->margin + T + 704) && (L.Co & 15) == 0),
              "conv_x3s_run: split output image too short or Co not a multiple of 16");
  RVC_REQUIRE((double)L.groups * L.Co * (double)(Y ? ldY : 1) * 4.0 < 2147483648.0 && (double)L.groups * L.Co * (double)e.ldR * 4.0 < 2147483648.0, "tensor extent exceeds 32-bit buffer addressing");
  const int G = L.mode == 1 ? L.groups : 1;
  RVC_REQUIRE(L.seg2_chunks == 0 || (G == 1 && geom->seg2_off > 0), "conv_x3s_run: a layer with an appended product needs its second image");
  const double xs_bytes = std::max((double)G * (L.Ci / 16) * 4.0 * (double)xsTp * 16.0, L.seg2_chunks ? (double)geom->seg2_off + (double)L.seg2_chunks * 4.0 * (double)xsTp * 16.0 : 0.0);
  const double wx_bytes = (double)G * (L.Ci / 16) * geom->ktaps * 4.0 * (double)L.CoPx * 16.0 + (double)L.seg2_chunks * 4.0 * (double)L.CoPx * 16.0;
  RVC_REQUIRE(G == 1 || (!e.ys_out || (L.Co & 15) == 0), "grouped layer: rows per group must be a multiple of 16 for the image output");
  RVC_REQUIRE(xs_bytes < 2147483648.0 && wx_bytes < 2147483648.0, "operand image exceeds 32-bit buffer addressing");
  GemmSArgs a{};
  a.Wx = reinterpret_cast<const unsigned char*>(L.Wx_); a.CoPx = L.CoPx; a.Xs = Xs; a.xsTp = xsTp;
  a.wx_bytes = (unsigned)wx_bytes; a.xs_bytes = (unsigned)xs_bytes;
  a.Co = L.Co; a.T = T; a.nunits = L.Ci / 16 * geom->ktaps + L.seg2_chunks;
  a.seg2_u = L.seg2_chunks ? L.Ci / 16 * geom->ktaps : 0x7fffffff; a.seg2_soff = (int)geom->seg2_off;
  a.ktaps = geom->ktaps; a.margin = geom->margin; a.ymargin = geom->margin; a.padw = geom->padw; a.ydeint = e.ys_out ? e.ys_deint_h : 0;
  if (e.gate_h > 0) {
    RVC_REQUIRE(G == 1 && L.mode == 1 && L.Co == 2 * e.gate_h && (e.gate_h & 15) == 0 && e.ys_out && Y == nullptr && !e.R && e.act == ACT_NONE && e.out_scale == 1.f && !e.vt_out && !e.ys_deint_h,
                "conv_x3s_run: the gate epilogue wants a 2 H-row layer packed by wn_gate_row_order, an image output and nothing else");
    a.gate_h = e.gate_h; a.gate_g = e.gate_g;
  }
  a.vt_row0 = 0x7fffffff;
  if (e.vt_out) {
    RVC_REQUIRE(G == 1 && L.mode == 1 && e.vt_row0 > 0 && (e.vt_row0 & 127) == 0 && e.vt_row0 < L.Co && e.vt_tp >= kSplitMargin + (L.Co - e.vt_row0) && Y == nullptr && !e.R && !s2,
                "conv_x3s_run: the transposed rows are the tail of an image-only k = 1 projection, from a multiple of 128 rows on");
    a.Vt = e.vt_out; a.vtTp = e.vt_tp; a.vt_row0 = e.vt_row0; a.vt_rows = L.Co - e.vt_row0;
  }
  a.padmagic = geom->padw > 0 ? (unsigned)((0x100000000ULL + (unsigned)geom->padw - 1) / (unsigned)geom->padw) : 0u;
  for (int t = 0; t < 16; ++t) a.toff[t] = t < geom->ktaps ? geom->toff[t] : 0;
  a.tdil = L.mode == 1 ? L.dil : 1; a.tpad = L.mode == 1 ? L.pad : 0;
  if (s2) { a.tdil = 1; a.tpad = 0; }
  a.groups = G; a.co_g = L.Co; a.cig_chunks = L.Ci / 16; a.wg_bytes = (unsigned)((double)(L.Ci / 16) * geom->ktaps * 4.0 * (double)L.CoPx * 16.0);
  a.bias = e.bias_override ? e.bias_override : L.bd_; a.R = e.R; a.ldR = e.ldR; a.Y = Y; a.ldY = ldY; a.Ys = e.ys_out; a.ysTp = e.ys_tp;
  a.act = e.act; a.act_slope = e.act_slope; a.act_before_res = e.act_before_res; a.out_scale = e.out_scale;
  int AM, AN, S;
  x3s_plan(L.Co, T, a.nunits, AM, AN, S, G, geom->ktaps);
  const int BM = 64 * AM, BN = 64 * AN;
  RVC_REQUIRE(L.CoPx % BM == 0, "weight image rows are padded to the tile");
  a.rows_pg = (L.Co + BM - 1) / BM;
  a.gx = (T + BN - 1) / BN; a.gy = a.rows_pg * G; a.ksplit = S;
  static const int xcd_env = exp_int("RVC_X3_XCD", 1);
  a.xcd_remap = xcd_env;
  static const int rf_env = exp_int("RVC_X3S_ROWFAST", 0);      // 0: column tiles fastest always, 1: by operand size, 2: row tiles fastest always
  a.row_fast = (G == 1 && xcd_env && (rf_env == 2 || (rf_env == 1 && (long long)L.Co < (long long)T))) ? 1 : 0;
  const unsigned blocks = (unsigned)((long long)a.gx * a.gy * S);
  if (S > 1) {
    // slabs + one ticket word per tile (zeroed when the scratch is first handed out and by every launch's last arriver)
    const size_t slab_bytes = (size_t)blocks * BM * BN * 4;
    const size_t tick_bytes = ((size_t)a.gx * a.gy * 4 + 255) & ~(size_t)255;
    static const size_t kTickCap = 64 * 1024;                  // ticket words live in their own scratch slot, sized once: never re-allocated dirty
    RVC_REQUIRE(tick_bytes <= kTickCap, "too many tiles for the ticket array");
    bool fresh = false;
    a.tickets = (unsigned*)stream_scratch_zeroed(s, 7, kTickCap, &fresh);
    a.slabs = (float*)stream_scratch(s, 6, slab_bytes);
  }
  ProfTicket tk = conv_prof_begin(s);
  x3s_dispatch(a, AM, AN, blocks, s);
  if (tk.on) {
    ConvArgsX pa{};
    pa.Ci = L.Ci; pa.Co = L.Co; pa.ktaps = geom->ktaps; pa.kreal = geom->ktaps; pa.dil = L.mode == 1 ? L.dil : 1; pa.stride = s2 ? 2 : 1; pa.Tin = T; pa.Tout = T;
    pa.Wd = geom->padw > 0 ? geom->padw - 2 : 0; pa.ksplit = S;
    pa.R = e.R; pa.X = nullptr;
    // algorithmic bytes: the input image (4 B per element, like fp32), the outputs that are written, the residual, the weights
    const double bytes = 4.0 * ((double)L.Ci * T + (double)L.Co * T * ((Y ? 1.0 : 0.0) + (e.ys_out ? 1.0 : 0.0) + (e.R ? 1.0 : 0.0)) + (double)L.Co * L.Ci * geom->ktaps);
    const int id = AM == 2 ? (AN == 2 ? 3 : 5) : (AN == 2 ? 5 : 6);
    conv_prof_end(tk, s, 2.0 * (double)G * L.Co * T * (L.Ci * geom->ktaps + 16 * L.seg2_chunks), 14 + id, bytes * (G > 1 ? (double)G : 1.0), &pa, (long long)blocks, 4 << 4);
  }
}

// The SWAPPED product on the same kernel: out[t][j] = sum_c X[c][t] W[row0 + j][c] (j < rows) - the activation image is the row operand
// (its rows are positions), the layer's weight image the column operand - written as the image of the TRANSPOSED tensor:
// [16-position chunk][hi | lo][8-position half][kSplitMargin + j][8 positions].  That is the V^T operand of the attention's P V product
// (attention_dma.hip): the reduction of P V runs over keys, so the keys must be the 8-element rows.  No bias (the caller adds V's bias
// after the attention: softmax rows sum to 1); rows >= T of the last chunk are written as zeros.  Yrm (optional): the same product as fp32 rows
// out[t][j] with pitch ldYrm - a time-major result without a transposition pass (RMVPE's GRU input projection).
void conv_x3s_run_swapped(const ConvLayer& L, int row0, int rows, hipStream_t s, const unsigned char* Xs, long long xsTp, int T, unsigned char* Ys, long long ysTp,
                          float* Yrm, long long ldYrm, const float* Rrm, long long ldRrm) {
  RVC_REQUIRE(L.Wx_ != nullptr && L.mode == 1 && L.k == 1 && L.groups == 1 && (L.Ci & 15) == 0, "conv_x3s_run_swapped: a k = 1 projection with a bf16x3 weight image");
  RVC_REQUIRE(row0 >= 0 && rows > 0 && row0 + rows <= L.Co && (row0 & 15) == 0, "conv_x3s_run_swapped: row range");
  RVC_REQUIRE(Xs != nullptr && xsTp >= kSplitMargin + T + 704 && (Ys != nullptr || Yrm != nullptr) && (!Ys || ysTp >= kSplitMargin + rows), "conv_x3s_run_swapped: images missing or too short");
  RVC_REQUIRE(!Yrm || (ldYrm >= rows && (double)T * (double)ldYrm * 4.0 < 2147483648.0), "conv_x3s_run_swapped: row-major output pitch");
  RVC_REQUIRE(!Rrm || (Yrm && ldRrm >= rows && (double)T * (double)ldRrm * 4.0 < 2147483648.0), "conv_x3s_run_swapped: the residual goes with the row-major output");
  const double xs_bytes = (double)(L.Ci / 16) * 4.0 * (double)xsTp * 16.0, wx_bytes = (double)(L.Ci / 16) * 4.0 * (double)L.CoPx * 16.0;
  RVC_REQUIRE(xs_bytes < 2147483648.0 && wx_bytes < 2147483648.0, "operand image exceeds 32-bit buffer addressing");
  GemmSArgs a{};
  a.Wx = Xs + (size_t)kSplitMargin * 16; a.CoPx = (int)xsTp; a.wx_bytes = (unsigned)xs_bytes - (unsigned)kSplitMargin * 16u;
  a.Xs = reinterpret_cast<const unsigned char*>(L.Wx_) + (size_t)row0 * 16; a.xsTp = L.CoPx; a.xs_bytes = (unsigned)wx_bytes - (unsigned)row0 * 16u;
  a.Co = T; a.T = rows; a.nunits = L.Ci / 16; a.ktaps = 1; a.margin = 0; a.ymargin = kSplitMargin; a.zero_tail = 1; a.seg2_u = 0x7fffffff; a.vt_row0 = 0x7fffffff;
  a.groups = 1; a.co_g = T; a.cig_chunks = L.Ci / 16; a.tdil = 1;
  a.Ys = Ys; a.ysTp = ysTp; a.act = ACT_NONE; a.out_scale = 1.f;
  a.Y = Yrm; a.ldY = ldYrm;                                    // fp32 out[t][j], row-major (the GRU's input projection)
  a.R = Rrm; a.ldR = ldRrm;
  int AM, AN, S;
  x3s_plan(T, rows, a.nunits, AM, AN, S, 1, 1, true);
  S = 1;                                                       // (48-unit reductions: never split)
  const int BM = 64 * AM, BN = 64 * AN;
  RVC_REQUIRE((long long)((T + BM - 1) / BM) * BM <= xsTp - kSplitMargin && (long long)row0 + (long long)((rows + BN - 1) / BN) * BN <= L.CoPx, "conv_x3s_run_swapped: a tile would read past an operand image");
  a.rows_pg = (T + BM - 1) / BM; a.gx = (rows + BN - 1) / BN; a.gy = a.rows_pg; a.ksplit = 1;
  static const int xcd_env = exp_int("RVC_X3_XCD", 1);
  a.xcd_remap = xcd_env;
  const unsigned blocks = (unsigned)((long long)a.gx * a.gy);
  ProfTicket tk = conv_prof_begin(s);
  x3s_dispatch(a, AM, AN, blocks, s);
  if (tk.on) {
    ConvArgsX pa{};
    pa.Ci = L.Ci; pa.Co = rows; pa.ktaps = 1; pa.kreal = 1; pa.dil = 1; pa.stride = 1; pa.Tin = T; pa.Tout = T; pa.ksplit = 1;
    const double bytes = 4.0 * ((double)L.Ci * T + (double)rows * T * (Rrm ? 2.0 : 1.0) + (double)rows * L.Ci);
    const int id = AM == 2 ? (AN == 2 ? 3 : 5) : (AN == 2 ? 5 : 6);
    conv_prof_end(tk, s, 2.0 * (double)rows * T * L.Ci, 14 + id, bytes, &pa, (long long)blocks, 4 << 4);
  }
}

}  // namespace rvc
