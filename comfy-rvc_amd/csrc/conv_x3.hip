// bf16x3 split-precision implicit-GEMM Conv1d on the CDNA4 matrix cores (v_mfma_f32_32x32x16_bf16), gfx950 only.
//
// Every fp32 operand is split x = hi + lo with hi = bf16(x), lo = bf16(x - hi); the product is evaluated as
// hi*hi + hi*lo + lo*hi with fp32 accumulation (the dropped lo*lo term is 2^-18 relative).  Three bf16 MFMAs do the work of
// eight fp32 ones (32x32x16 vs 32x32x2 per instruction at twice the issue cost), so the generator's ResBlock convolutions run
// at ~5x the fp32 matrix rate while the end-to-end synthesizer output stays within 3e-5 of the fp32 path (tolerance 1e-3).
//
// Same GEMM view, tiles and epilogue as conv_mfma.hip (stride-1 1-D convolutions only):
//   Y[m][n] = sum_{chunk, tap, c16} W[m][chunk*16 + c16][tap] * X[chunk*16 + c16][n - pad + tap*dil]
// One MFMA consumes 16 channels of one tap: lane (i, half) holds channels half*8 .. half*8+7 of row / position i.
// LDS images: the 16 channels of a row are kept as two half-planes of 8 channels = 16 B per row ([half][row][8 ch]).  Lane (i, half)
// reads one 16-B row of its half-plane: the 16 lanes a ds_read_b128 services per LDS cycle ({0-3,12-15,20-27}, ...) touch 16 rows that
// are distinct mod 16 = all 64 banks once, for ANY tap offset, and an operand address is base + row * 16 (one add per tap; round 1-2
// kept 32-B rows with the halves swapped by bit 3 of the row index, which cost ~7 VALU per operand read to undo).
//   X: [chunk][hi|lo][half][position q][8 ch] bf16, converted from the fp32 [C][T] activation while staging (registers -> ds_write_b128)
//   W: [chunk][tap][hi|lo][half][row m][8 ch] bf16, copied from the host-packed image by global_load_lds_dwordx4 (no registers),
//      double-buffered so that the DMA of stage s+1 runs under the MFMAs of stage s; one barrier per stage.
// A stage is NC 16-channel chunks x KT taps: NC = 1 for the wide dilated kernels of the generator, up to 4 for k = 1 (GEMM).
#include "conv_x3_dev.h"

#ifndef RVC_X3_SCHED
#define RVC_X3_SCHED 0
#endif

namespace rvc {

#ifdef RVC_CONV_TIMING
__device__ unsigned long long g_x3_timing[8];   // [0] blocks, [1] prologue, [2] X store + DMA wait + barrier, [3] prefetch issue, [4] MFMA loops, [5] epilogue, [6] total
// (accumulated in registers and published once per workgroup: an atomic per stage is a VMEM operation on the path the counted vmcnt waits watch)
#define X3TICK() ((long long)__builtin_readcyclecounter())
#define X3TACC(i, v) do { x3t[i] += (v); } while (0)
#define X3TDECL() long long x3t[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define X3TFLUSH() do { if (threadIdx.x == 0) { for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&g_x3_timing[i_], (unsigned long long)x3t[i_]); } } while (0)
void conv_x3_timing_read(unsigned long long* out8, bool reset) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_x3_timing), sizeof(unsigned long long) * 8);
  if (reset) { unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_x3_timing), z, sizeof(z)); }
}
#else
#define X3TICK() 0ll
#define X3TACC(i, v) do {} while (0)
#define X3TDECL() do {} while (0)
#define X3TFLUSH() do {} while (0)
#endif

// FUSE: ResBlock pair  y = (x + W2 * lrelu(W1 *_d lrelu(x) + b1) + b2) * scale [+ y]  in one launch (Ci = Co, all channels of the
// tile resident in LDS): pass 1 is the ordinary stage loop of the dilated conv over BN columns; its accumulators (+ b1, leaky
// ReLU, zero outside the sequence = the second conv's zero padding) are split and written over the input tile in LDS in the
// same row format; pass 2 runs the stage loop of the second conv (dilation 1) on those rows and the ordinary epilogue stores the
// BN - (k - 1) columns that have their full halo.  The intermediate tensor never goes to HBM: x (tile + residual) and y instead of
// x, t, t, x, y.  Used for the 32-channel generator stage (conv_x3_pair_try): k3 261 -> 196 us, k7 304 -> 233, k11 348 -> 308.
// XSPLIT: the input arrives as a split-resident image (ConvEpilogue::xs_in) and is copied into LDS by DMA; no staging registers.
// YSPLIT: the output is written as a split-resident image (ConvEpilogue::ys_out) - separate instantiations, so that the plain kernels keep
// their register budgets (the 128 x 128 tile its three workgroups per CU).
// WM x WN = 4 waves (256 threads, two or three workgroups per CU) or 8 waves (512 threads, ONE workgroup per CU, two waves per SIMD: the
// eight waves share one weight stream, so the L2 -> LDS traffic and the DMA issue per output halve against two 4-wave workgroups, and the
// whole LDS of the CU is one workgroup's: 4-5 taps per stage instead of 1)
template <int WM, int WN, int AM, int AN, bool FUSE = false, bool XSPLIT = false, bool YSPLIT = false>
__global__ __launch_bounds__(WM * WN * 64, WM * WN == 8 ? 1 : ((WM == 2 && WN == 2 && AM == 2 && AN == 2 && !FUSE) ? 3 : 2)) void conv_x3_kernel(const ConvArgsX p) {   // (128 x 128 tile: three workgroups per CU = 168 VGPRs)
  constexpr int NW = WM * WN, NT = NW * 64;
  constexpr int BM = WM * AM * 32, BN = WN * AN * 32, XS = XSPLIT ? 1 : ((FUSE && BM == 64) ? 7 : (FUSE ? 5 : x3_slots(BN, NW))), RB = BM / 32;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem3[];
  const int P = p.WROW;                     // staged input positions: (BN - 1) * stride + (ktaps - 1) * dil + 1
  const int NC = p.NC;                      // 16-channel chunks per stage
  const int st = p.stride;                  // strided convs keep one sub-plane per input phase (position mod stride): unit-stride reads
  const int Pm = XSPLIT ? ((P + 63) & ~63) : (P + st - 1) / st;   // rows per phase sub-plane (split-resident input: whole 1-KiB pieces of a half-plane)
  const int xplane = st * Pm * 32;          // bytes of one hi / lo plane
  const int xhalf = xplane >> 1;            // ... of one of its two 8-channel half-planes
  const int xbuf = NC * 2 * xplane;         // bytes of one X buffer
  const int wbuf = NC * p.KT * 2 * BM * 32; // bytes of one weight buffer
  unsigned char* Xs = smem3;
  unsigned char* Ws = smem3 + ((p.xbufs * xbuf + 1023) & ~1023);
  const int NS = p.wbufs;                   // weight slabs in the ring (2 .. 4): the DMA of stage s + NS - 1 is issued during stage s

  const int tid0 = threadIdx.x, lane0 = tid0 & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane0 & 31, lh = lane0 >> 5;
  const int z = blockIdx.z / p.ksplit, ks = blockIdx.z - z * p.ksplit;   // batch index, K-split index
  const unsigned tile = p.xcd_remap ? xcd_tile(blockIdx.x + blockIdx.y * gridDim.x, gridDim.x * gridDim.y) : blockIdx.x + blockIdx.y * gridDim.x;
  const int tile_y = (int)(tile / gridDim.x), tile_x = (int)(tile - (unsigned)tile_y * gridDim.x);
  const int co0 = tile_y * BM;
  const int P2 = FUSE ? p.fuse_p2 : 0;                                    // halo of the fused second conv (each side)
  const int n0 = tile_x * (BN - 2 * P2);
  const float* __restrict__ X = p.X + (long long)z * p.xBatch;
  const unsigned char* __restrict__ Wg = p.Wx + (long long)z * p.wxBatch * 2;

  X3TDECL();
  f32x16 acc[AM][AN];
#pragma unroll
  for (int am = 0; am < AM; ++am)
#pragma unroll
    for (int an = 0; an < AN; ++an)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[am][an][r] = 0.f;

  const int ntb = (p.ktaps + p.KT - 1) / p.KT;
  const int ngroups = p.nchunk / NC;                  // p.nchunk is a multiple of NC (host)
  const int gps = (ngroups + p.ksplit - 1) / p.ksplit;        // chunk groups per K split
  const int g0 = ks * gps, g1 = min(ngroups, g0 + gps);
  const int nstages = max(g1 - g0, 0) * ntb;
  const int bx = n0 * st - p.pad - P2;                 // (fused: column 0 of the tile is the first column of the intermediate)
  // 2-D 3x3 (p.Wd > 0): the tile is BH image rows x BWd columns; the staged "positions" are the (BH + 2) x PW halo patch in
  // row-major order, tap (dh, dw) is the position offset dh * PW + dw
  const bool two_d = p.Wd > 0;
  const int kKW = p.KW > 0 ? p.KW : 3, kPH = p.KW > 0 ? p.PH : 1, kPWL = p.KW > 0 ? p.PWL : 1;   // 2-D window (default 3 x 3, pad 1)
  int h0 = 0, w0 = 0;
  if (two_d) { h0 = n0 / p.Wd; w0 = n0 - h0 * p.Wd; }
  const int ni = p.ni;                                                      // 64-position groups per plane row set
  const __amdgpu_buffer_rsrc_t xrs = make_rsrc(X, (unsigned)p.Ci * (unsigned)p.ldX * 4u);
  const float pre_slope = p.pre_act == ACT_LRELU ? p.pre_slope : 1.f;      // input activation: leaky ReLU (slope 1 = identity)

  float xr[XS][8];

  // ---- input tile: global -> registers.  Slot s of wave w covers 64 positions x 8 channels (one 16-B half of the LDS rows).
  auto load_x = [&](int grp) {
    if constexpr (XSPLIT) return;
#ifdef RVC_X3_NOX
    return;
#endif
    int lane = lane0;
    asm volatile("" : "+v"(lane));
#pragma unroll
    for (int s = 0; s < XS; ++s) {
      const int t = wave + NW * s;
      const int cc = (t >= 2 * ni) + (t >= 4 * ni) + (t >= 6 * ni);       // NC <= 4
      const int g = t - cc * 2 * ni;
      const int hb = g >= ni ? 1 : 0;
      const int q = (g - hb * ni) * 64 + lane;
      int x = bx + q;
      bool ok = cc < NC && q < P && x >= 0 && x < p.Tin;
      if (two_d) {
        const int rr = (int)__umulhi((unsigned)q, p.magPW), cw = q - rr * p.PW;
        const int hh = h0 - kPH + rr, ww = w0 - kPWL + cw;
        ok = cc < NC && q < P && hh >= 0 && hh < p.Tin && ww >= 0 && ww < p.Wd;
        x = hh * p.Wd + ww;
      }
      const unsigned voff = ok ? (unsigned)x * 4u : kOOB;
      const unsigned c0 = (unsigned)((grp * NC + cc) * 16 + hb * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        xr[s][j] = buf_load(xrs, voff, (c0 + j) * (unsigned)p.ldX * 4u);   // raw: the activation is applied in store_x, so nothing here waits for the data
      }
    }
  };
  // ---- input tile: registers -> hi/lo bf16 -> LDS.  Two floats per v_cvt_pk_bf16_f32; the stride cases are separate straight-line
  // bodies (a nested select over the stride compiled into ~100 instructions of divisions and branches per slot: 46 k of a 128 x 256
  // tile's 340 k cycles went into this function)
  auto store_x = [&](int xb) {
    if constexpr (XSPLIT) return;
#ifdef RVC_X3_NOX
    return;
#endif
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    unsigned char* xbase0 = Xs + xb * xbuf;
#pragma unroll
    for (int s = 0; s < XS; ++s) {
      const int t = wave + NW * s;
      const int cc = (t >= 2 * ni) + (t >= 4 * ni) + (t >= 6 * ni);
      const int g = t - cc * 2 * ni;
      const int hb = g >= ni ? 1 : 0;
      const int q = (g - hb * ni) * 64 + lane;
      unsigned char* xbase = xbase0 + cc * 2 * xplane;
      if (cc < NC && q < P) {
        u32x4 hi, lo;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float a = xr[s][2 * j], b = xr[s][2 * j + 1];
          unsigned h_, l_;
          split2(fmaxf(a, a * pre_slope), fmaxf(b, b * pre_slope), h_, l_);     // input activation: leaky ReLU (slope 1 = identity)
          hi[j] = h_; lo[j] = l_;
        }
        int off;
        if (st == 1) off = hb * xhalf + q * 16;
        else {
          const int m = st == 2 ? (q >> 1) : q / st, ph = q - m * st;
          off = hb * xhalf + (ph * Pm + m) * 16;
        }
        *reinterpret_cast<u32x4*>(xbase + off) = hi;
        *reinterpret_cast<u32x4*>(xbase + xplane + off) = lo;
      }
    }
  };
  // ---- split-resident input: every (chunk, hi | lo, half) half-plane of the tile is a contiguous run of the image: 1 KiB (64 positions)
  // per wave-instruction straight into LDS.  Returns the number of pieces this wave issued.
  auto issue_x = [&](int grp, int xb) -> int {
    if constexpr (!XSPLIT) return 0;
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int pph = xhalf >> 10;                                    // pieces per half-plane (xhalf is a multiple of 1 KiB)
    const int npieces = NC * 4 * pph;
    for (int pi = wave; pi < npieces; pi += NW) {
      const int hp = pi / pph, j = pi - hp * pph;                   // half-plane = ((chunk in group) * 2 + (hi | lo)) * 2 + half
      const long long row = (long long)((grp * NC) * 4 + hp) * p.xsTp + (bx + kSplitMargin) + (long long)j * 64;
      const unsigned char* src = p.Xs + row * 16 + lane * 16;
      unsigned char* dst = Xs + xb * xbuf + hp * xhalf + j * 1024;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
    return wave < npieces ? (npieces - wave + NW - 1) / NW : 0;
  };
  // rows of the staged tile that lie outside the sequence are the convolution's zero padding: the image holds no defined data there
  const bool x_edge = XSPLIT && (bx < 0 || bx + P > p.Tin);
  auto zero_edges = [&](int xb) {
    for (int q = tid0; q < P; q += NT) {
      const int t = bx + q;
      if (t >= 0 && t < p.Tin) continue;
      for (int pl = 0; pl < NC * 2; ++pl) {
        unsigned char* r = Xs + xb * xbuf + pl * xplane + q * 16;
        *reinterpret_cast<u32x4*>(r) = u32x4{0u, 0u, 0u, 0u}; *reinterpret_cast<u32x4*>(r + xhalf) = u32x4{0u, 0u, 0u, 0u};
      }
    }
  };

  // ---- weight slab of (chunk, tap block): global -> LDS by DMA, 1 KiB (64 rows of one half-plane; BM = 32: both halves) per
  // wave-instruction.  LDS 16-B row r of a (chunk, tap, hi | lo) plane is (half, m) = (r / BM, r % BM); the image keeps [half][CoPx rows].
  auto issue_w = [&](const unsigned char* __restrict__ Wimg, int grp, int tb, int buf) -> int {
#ifdef RVC_X3_NOW
    return 0;
#endif
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int ut = min(p.KT, p.ktaps - tb * p.KT);
    const int npieces = NC * ut * 2 * RB;
    auto lane_src = [&](int rblk) -> long long {                   // byte offset of this lane's row inside a plane of the image
      const int r = rblk * 64 + lane;
      return ((long long)(r / BM) * p.CoPx + co0 + (r % BM)) * 16;
    };
    if (NC == 1 || ut == p.ktaps) {
      // the (chunk, tap, hi | lo) rows of this stage are consecutive in the image: piece pi = wave + 4 i is row j0 + i * (4 / RB), 32-row
      // block rblk - one multiply-add per piece instead of three integer divisions (the generic path below cost ~400 cycles per piece,
      // a fifth of a stage of the 128 x 256 tile)
      static_assert(NW % RB == 0, "row blocks per wave round");
      constexpr int JS = NW / RB;
      const int j0 = wave / RB, rblk = wave % RB;
      const long long row0 = (long long)((grp * NC) * p.ktaps + tb * p.KT) * 2 + j0;
      const unsigned char* src = Wimg + row0 * p.CoPx * 32 + lane_src(rblk);
      const long long sstep = (long long)JS * p.CoPx * 32;
      unsigned char* dst = Ws + buf * wbuf + wave * 1024;
      for (int pi = wave; pi < npieces; pi += NW, src += sstep, dst += NW * 1024)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    } else
    for (int pi = wave; pi < npieces; pi += NW) {
      const int j = pi / RB, rblk = pi - j * RB;            // j = ((chunk in group) * ut + tap in block) * 2 + (hi | lo)
      const int cc = j / (2 * ut), jr = j - cc * 2 * ut;
      const long long row = (long long)((grp * NC + cc) * p.ktaps + tb * p.KT + (jr >> 1)) * 2 + (jr & 1);
      const unsigned char* src = Wimg + row * p.CoPx * 32 + lane_src(rblk);
      unsigned char* dst = Ws + buf * wbuf + pi * 1024;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
    return wave < npieces ? (npieces - wave + NW - 1) / NW : 0;
  };

  int aoff[AM], bq[AN];
#pragma unroll
  for (int am = 0; am < AM; ++am) { const int m = (wm * AM + am) * 32 + li; aoff[am] = lh * (BM * 16) + m * 16; }
#pragma unroll
  for (int an = 0; an < AN; ++an) {
    const int nl = (wn * AN + an) * 32 + li;
    bq[an] = (two_d ? (nl / p.BWd) * p.PW + (nl % p.BWd) : nl) * 16 + lh * xhalf;      // byte offset of this lane's row in its half-plane
  }

  // ---- MFMAs of one stage: NC chunks x ut taps of weight buffer `buf` against input buffer `xb`
  auto mfma_stage = [&](int buf, int xb, int tb, int dil_eff) {
    const int ut = min(p.KT, p.ktaps - tb * p.KT);
    const unsigned char* wb = Ws + buf * wbuf;
#ifdef RVC_X3_NOMFMA
    for (int cu = 0; cu < 0; ++cu) {
#else
    for (int cu = 0, cc = 0, uu = 0; cu < NC * ut; ++cu, cc += (uu + 1 == ut), uu = (uu + 1 == ut) ? 0 : uu + 1) {
#endif
      const unsigned char* xp = Xs + xb * xbuf + cc * 2 * xplane;
      const int u = tb * p.KT + uu;
      const int toff = two_d ? (u / kKW) * p.PW + (u % kKW) : (st == 1 ? u * dil_eff : u / st);   // row offset inside the (phase) plane
      if (st > 1) xp += (u - toff * st) * Pm * 16;
      xp += toff * 16;
      const unsigned char* wt = wb + cu * 2 * BM * 32;
      u32x4 ah[AM], al[AM];
#pragma unroll
      for (int am = 0; am < AM; ++am) {
        ah[am] = *reinterpret_cast<const u32x4*>(wt + aoff[am]);
        al[am] = *reinterpret_cast<const u32x4*>(wt + BM * 32 + aoff[am]);
      }
      // B operands in groups of at most four column blocks (eight-block tiles: 32 instead of 64 operand registers live at once)
      constexpr int ANG = AN > 4 ? 4 : AN;
#pragma unroll
      for (int a0 = 0; a0 < AN; a0 += ANG) {
        u32x4 bh[ANG], bl[ANG];
#pragma unroll
        for (int an = 0; an < ANG; ++an) {
          const int off = bq[a0 + an];
          bh[an] = *reinterpret_cast<const u32x4*>(xp + off);
          bl[an] = *reinterpret_cast<const u32x4*>(xp + xplane + off);
        }
#if RVC_X3_SCHED == 1
        __builtin_amdgcn_sched_barrier(0);     // experiment: every operand read of the unit issued before its first MFMA (one LDS wait per unit)
#endif
#pragma unroll
        for (int am = 0; am < AM; ++am)
#pragma unroll
          for (int an = 0; an < ANG; ++an)
            acc[am][a0 + an] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al[am]), __builtin_bit_cast(bf16x8, bh[an]), acc[am][a0 + an], 0, 0, 0);
#pragma unroll
        for (int am = 0; am < AM; ++am)
#pragma unroll
          for (int an = 0; an < ANG; ++an)
            acc[am][a0 + an] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[am]), __builtin_bit_cast(bf16x8, bl[an]), acc[am][a0 + an], 0, 0, 0);
#pragma unroll
        for (int am = 0; am < AM; ++am)
#pragma unroll
          for (int an = 0; an < ANG; ++an)
            acc[am][a0 + an] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[am]), __builtin_bit_cast(bf16x8, bh[an]), acc[am][a0 + an], 0, 0, 0);
      }
    }
  };

  // ---- stage loop.  VMEM operations of a wave retire in issue order, so "stage s's weights have landed" = at most (operations this
  // wave issued after the last piece of stage s) outstanding: `issued` counts them, we0 .. we3 remember the count right after the
  // weights of stages it .. it + 3.  The input rows of chunk c + 1 are requested as soon as chunk c's registers are free (first stage
  // of chunk c), a whole chunk ahead of their use; nothing in the loop waits for vmcnt(0).
  int chunk = g0, tb = 0;
  const long long t_begin = X3TICK();
  int issued = 0, we0 = 0, we1 = 0, we2 = 0, we3 = 0, xe = 0;   // xe: `issued` right after the input pieces of the current chunk
  int nchunk_i = g0, ntb_i = 0, issued_stages = 0;         // next (chunk, tap block) whose weights are to be requested
  auto issue_next = [&]() {
    const int pieces = issue_w(Wg, nchunk_i, ntb_i, issued_stages % NS);
    issued += pieces;
    ++issued_stages;
    if (++ntb_i == ntb) { ntb_i = 0; ++nchunk_i; }
    return issued;
  };
  if (nstages > 0) {
    we0 = issue_next();
    if (NS > 2 && nstages > 1) we1 = issue_next();
    if (NS > 3 && nstages > 2) we2 = issue_next();
    if constexpr (XSPLIT) { issued += issue_x(g0, p.xbufs == 2 ? (g0 & 1) : 0); xe = issued; }
    else { load_x(g0); issued += XS * 8; }
  }
  X3TACC(1, X3TICK() - t_begin);
  for (int it = 0; it < nstages; ++it) {
    const long long ta = X3TICK();
    const int buf = it % NS;
    const int xb = p.xbufs == 2 ? (chunk & 1) : 0;
    if (!XSPLIT && tb == 0) {
      if (p.xbufs == 1 && it > 0) lds_barrier();           // single X buffer: every wave is done with the previous chunk
      store_x(xb);
      if (chunk + 1 < g1) { load_x(chunk + 1); issued += XS * 8; }
    }
    const long long tw0 = X3TICK();
    // this stage's weight pieces (of this wave) have landed - and, at the start of a chunk, its split-resident input pieces
    wait_vmcnt_le((XSPLIT && tb == 0) ? min(issued - we0, issued - xe) : issued - we0);
    const long long tw1 = X3TICK();
    lds_barrier();                                          // ... and everybody else's; the slab of stage it - 1 is free again
    const long long tb_ = X3TICK();
    X3TACC(2, tw0 - ta); X3TACC(7, tw1 - tw0); X3TACC(1, tb_ - tw1);      // [2] input store, [7] DMA wait, [1] += barrier
    if (XSPLIT && tb == 0) {
      if (x_edge) { zero_edges(xb); lds_barrier(); }        // (first / last tiles only) zero padding rows, published before the MFMAs
      // the other input buffer was read during the previous chunk; every wave is past it now (barrier above): request the next chunk
      if (chunk + 1 < g1) { issued += issue_x(chunk + 1, xb ^ 1); xe = issued; }
    }
    we0 = we1; we1 = we2; we2 = we3;
    if (issued_stages < nstages) {
      const int e = issue_next();
      // (bit selects: an if-chain here becomes an indexed store to scratch, and the scratch load that follows waits vmcnt(0))
      const int m2 = -(int)(NS == 2), m3 = -(int)(NS == 3), m4 = -(int)(NS >= 4);
      we0 = (e & m2) | (we0 & ~m2); we1 = (e & m3) | (we1 & ~m3); we2 = (e & m4) | (we2 & ~m4);
    }
    const long long tc = X3TICK();
    X3TACC(3, tc - tb_);
    mfma_stage(buf, xb, tb, p.dil);
    if (++tb == ntb) { tb = 0; ++chunk; }
    X3TACC(4, X3TICK() - tc);
  }
  if constexpr (FUSE) {
    // ---- pass 1 -> LDS: h = lrelu(acc + b1) (0 outside the sequence), split, written over the input tile (row = tile column)
    __syncthreads();                                        // every wave is done with the input tile and both weight buffers
    const unsigned char* __restrict__ Wg2 = p.Wx2;
    issue_w(Wg2, 0, 0, 0);                                  // first slab of the second conv lands while h is written
    {
      const float hs = p.fuse_slope;
#pragma unroll
      for (int am = 0; am < AM; ++am)
#pragma unroll
        for (int an = 0; an < AN; ++an) {
          const int nl = (wn * AN + an) * 32 + li;
          const int gh = n0 - P2 + nl;
          const bool inside = gh >= 0 && gh < p.Tin;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int mb = (wm * AM + am) * 32 + 8 * g;     // rows mb + 4 lh + {0..3}: one 8-byte quarter of an LDS row
            u32x4 hl;                                       // {hi01, hi23, lo01, lo23}
#pragma unroll
            for (int e2 = 0; e2 < 2; ++e2) {
              float a = acc[am][an][4 * g + 2 * e2] + p.bias1[mb + 4 * lh + 2 * e2];
              float b = acc[am][an][4 * g + 2 * e2 + 1] + p.bias1[mb + 4 * lh + 2 * e2 + 1];
              a = inside ? fmaxf(a, a * hs) : 0.f;
              b = inside ? fmaxf(b, b * hs) : 0.f;
              const __bf16 ah = (__bf16)a, bh = (__bf16)b;
              const __bf16 al = (__bf16)(a - (float)ah), bl = (__bf16)(b - (float)bh);
              hl[e2] = bf16_bits(ah) | (bf16_bits(bh) << 16);
              hl[2 + e2] = bf16_bits(al) | (bf16_bits(bl) << 16);
            }
            const int cc = mb >> 4, hb = (mb >> 3) & 1;
            unsigned char* row = Xs + cc * 2 * xplane + hb * xhalf + nl * 16 + lh * 8;
            *reinterpret_cast<unsigned long long*>(row) = (unsigned long long)hl[0] | ((unsigned long long)hl[1] << 32);
            *reinterpret_cast<unsigned long long*>(row + xplane) = (unsigned long long)hl[2] | ((unsigned long long)hl[3] << 32);
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[am][an][r] = 0.f;
        }
    }
    // ---- pass 2: the second conv (dilation 1) over the rows just written
    for (int it = 0; it < ntb; ++it) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                                      // (it = 0: also publishes h)
      if (it + 1 < ntb) issue_w(Wg2, 0, it + 1, (it + 1) & 1);
      mfma_stage(it & 1, 0, it, 1);
    }
  }
  const long long t_epi = X3TICK();
#ifdef RVC_X3_NOEPI
  if (acc[0][0][0] == 12345.678f)
#endif
  if (p.ksplit > 1) {
    // split-K: raw partial sums; bias / activation / residual are applied by splitk_reduce_kernel in a fixed order
    float* Pp = p.partial + ((long long)blockIdx.z * p.Co) * p.ldP;
    const __amdgpu_buffer_rsrc_t prs = make_rsrc(Pp, (unsigned)p.Co * (unsigned)p.ldP * 4u);
#pragma unroll
    for (int am = 0; am < AM; ++am)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = co0 + (wm * AM + am) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
#pragma unroll
        for (int an = 0; an < AN; ++an) {
          const int n = n0 + (wn * AN + an) * 32 + li;
          const bool ok = m < p.Co && n < p.Tout;
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[am][an][r]), prs, (int)(ok ? ((unsigned)m * (unsigned)p.ldP + (unsigned)n) * 4u : kOOB), 0, 0);
        }
      }
  } else if (YSPLIT) {
    ysplit_epilogue<WM, WN, AM, AN>(p, acc, co0, n0, wm, wn, li, lh);
  } else if (p.ostride == 1) {
    if constexpr (FUSE) {
      ConvArgsX pe = p;
      pe.Tout = min(p.Tout, n0 + BN - 2 * P2);               // columns without their full halo belong to the neighbouring tiles
      dense_epilogue<WM, WN, AM, AN, (AM * AN >= 8 ? 4 : 8)>(pe, acc, z, co0, n0, wm, wn, li, lh);
    } else {
      dense_epilogue<WM, WN, AM, AN, (AM * AN >= 8 ? (AN >= 8 ? 2 : 4) : (XSPLIT ? 4 : 8))>(p, acc, z, co0, n0, wm, wn, li, lh);
    }
  } else {
    // interleaved store of the ConvTranspose1d phases: row m = co * ostride + phase goes to Y[co][n * ostride + phase]
    const float* __restrict__ bias = p.bias;
    float* Y = p.Y;
    const float lslope = p.act == ACT_NONE ? 1.f : (p.act == ACT_RELU ? 0.f : p.act_slope);
#pragma unroll
    for (int am = 0; am < AM; ++am) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = co0 + (wm * AM + am) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m >= p.Co) continue;
        const int co = m / p.ostride, ph = m - co * p.ostride;      // phase-fastest rows (tconv1d_layer_init)
        const float bv = bias ? bias[co] : 0.f;
#pragma unroll
        for (int an = 0; an < AN; ++an) {
          const int n = n0 + (wn * AN + an) * 32 + li;
          const long long to = (long long)n * p.ostride + ph;
          if (n >= p.Tout || to >= p.ldY) continue;   // ldY doubles as the true output length for interleaved stores
          const long long oidx = (long long)co * p.ldY + to;
          float v = acc[am][an][r] + bv;
          v = fmaxf(v, v * lslope) * p.out_scale;
          if (p.accumulate) v += Y[oidx];
          Y[oidx] = v;
        }
      }
    }
  }
  X3TACC(5, X3TICK() - t_epi); X3TACC(6, X3TICK() - t_begin); X3TACC(0, 1);
  X3TFLUSH();
}

// ============================================================================ host side
bool conv_x3_enabled() {
  static const bool on = (exp_int("RVC_X3", 1) != 0);
  return on;
}

template <int WM, int WN, int AM, int AN, bool FUSE = false, bool XSPLIT = false, bool YSPLIT = false>
static void launch_x3(const ConvArgsX& a, dim3 grid, size_t lds, hipStream_t s) {
  auto kern = conv_x3_kernel<WM, WN, AM, AN, FUSE, XSPLIT, YSPLIT>;
  RVC_ALLOW_BIG_LDS(kern);
  conv_launch(kern, grid, dim3(WM * WN * 64), lds, s, a);
}

bool conv_x3_try(ConvArgsX& a0, int batch, hipStream_t s, double flops, bool dry) {
  if (!conv_x3_enabled() || !a0.Wx) return false;
  const bool xs = a0.Xs != nullptr;
  if ((xs || a0.Ys) && (a0.Wd > 0 || a0.stride != 1 || a0.ostride != 1 || (a0.Co & 31) || batch != 1)) return false;
  if (xs && a0.pre_act != ACT_NONE) return false;                     // the producer applied the activation
  // the image's rows are addressed as margin + t (global_load_lds is not bounds-checked): "same" convolutions whose left halo fits the
  // margin only, on an image with the documented number of rows per plane
  if ((xs || a0.Ys) && (a0.pad > kSplitMargin || a0.Tout != a0.Tin)) return false;
  if (!dry && xs) RVC_REQUIRE(a0.xsTp >= split_image_tp(a0.Tin), "split-resident input image is shorter than split_image_tp(T)");
  if (!dry && a0.Ys) RVC_REQUIRE(a0.ysTp >= split_image_tp(a0.Tout), "split-resident output image is shorter than split_image_tp(T)");
  const int KH2 = a0.KW > 0 ? a0.KH : 3, KW2 = a0.KW > 0 ? a0.KW : 3;
  if (a0.Wd > 0 && (a0.ktaps != KH2 * KW2 || a0.stride != 1)) return false;
  if ((a0.stride != 1 && a0.dil != 1) || a0.up2 || (a0.ostride != 1 && a0.R) || (a0.Ci & 15) || batch != 1) return false;
  if (!(a0.act == ACT_NONE || a0.act == ACT_LRELU || a0.act == ACT_RELU) || !(a0.pre_act == ACT_NONE || a0.pre_act == ACT_LRELU)) return false;
  if ((double)a0.orows * (double)a0.ldY * 4.0 >= 2147483648.0 || (double)a0.orows * (double)a0.ldR * 4.0 >= 2147483648.0 ||
      (double)a0.Ci * (double)a0.ldX * 4.0 >= 2147483648.0) return false;
  ConvArgsX a = a0;
  if (a.Wd == 0) a.ktaps = a0.kreal;                      // true taps (the fp32 kernel folds the stride phases into virtual channels)
  if (batch == 1 && a.up2 == 0 && !a.h2) {
    // k = 1, the 3 x 3 convolutions of small images and short 1-D sequences: the pipelined GEMM kernel (conv_x3p.hip)
    dim3 g; int S = 1;
    if (conv_x3g_try(a, s, g, S, true)) {
      if (dry) return true;
      ProfTicket tk = conv_prof_begin(s);
      conv_x3g_try(a, s, g, S, false);
      conv_prof_end(tk, s, flops, 14 + ((a.Co > 64 && a.Ci * a.ktaps > 1024) ? 3 : 5), conv_alg_bytes(a, batch), &a, (long long)g.x * g.y * g.z, 2 << 4);
      return true;
    }
  }
  TileCfg t = choose_tile(a.Co, a.Tout, batch);
  {
    // wide tiles (8 accumulators per wave): every workgroup re-fetches the whole weight image from L2, so the L2 -> LDS stream
    // per output halves with twice the positions per workgroup; taken when the grid still fills the chip several times over
    // (measured: C128 k11 610 -> 470 us; deeper weight buffering instead of wider tiles was slower)
    static const int wide_blk = exp_int("RVC_X3_WIDE", 600);
    auto blocks = [&](int bm, int bn) { return (long long)((a.Co + bm - 1) / bm) * ((a.Tout + bn - 1) / bn); };
    if (wide_blk > 0 && a.stride == 1 && a.Wd == 0) {
      // (k <= 3 at 128+ channels is HBM-bound: three 128 x 128 workgroups per CU beat two wide ones, C128 k3 238 -> 219 us)
      // (round 2, after the staging / epilogue changes and with split-resident inputs: the 64 x 256 tile at three workgroups per CU now
      // beats 64 x 512 at two - C64 k7 205 -> 187 us - and a split-input consumer with k <= 7 prefers 128 x 128: 333 -> 317 us)
      static const int wide64 = exp_int("RVC_X3_WIDE64", 0);
      // 8-wave workgroups (128 x 512 tile, one per CU): RVC_X3_W8 = minimum tap count that takes them (0 = never)
      static const int w8_taps = exp_int("RVC_X3_W8", 0);
      static const int w8_blk = exp_int("RVC_X3_W8_BLK", 400);
      static const int wide_xs7 = exp_int("RVC_X3_WIDE_XS7", 1);   // the wide tile also for a split-input consumer with k = 7 (persistent kernel: C128 k7 pair 399 -> 377 us; the per-tile kernel preferred 128 x 128 there)
      static const int wide_k3 = exp_int("RVC_X3_WIDE_K3", 2);      // the wide tile also for k = 3 (1: fp32 inputs, 2: split inputs too): on the pipelined kernel C128 k3 135 -> 129 us; not for the up-samplers (240 -> 250)
      if (w8_taps > 0 && a.Co > 64 && a.ktaps >= w8_taps && blocks(128, 512) >= w8_blk) t = TileCfg{2, 4, 2, 4};
      else
      if (a.Co > 64 && blocks(128, 256) >= wide_blk && (a.ktaps > 3 || (wide_k3 >= 1 && a.ostride == 1)) &&
          !(xs && a.ktaps <= 7 && !(a.ktaps <= 3 && wide_k3 >= 2) && !(a.ktaps == 7 && wide_xs7))) t = TileCfg{2, 2, 2, 4};
      else if (wide64 && a.Co > 32 && a.Co <= 64 && blocks(64, 512) >= wide_blk) t = TileCfg{1, 4, 2, 4};
    }
  }
  if (const char* f = RVC_EXP_STR("RVC_FORCE_TILE")) {
    int w[4]; if (sscanf(f, "%d,%d,%d,%d", &w[0], &w[1], &w[2], &w[3]) == 4 && (a.Co > 32 || w[0] == 1)) t = TileCfg{w[0], w[1], w[2], w[3]};
  }
  int id = tile_cfg_id(t);
  if (t.WM == 2 && t.WN == 2 && t.AM == 2 && t.AN == 4) id = 7;
  if (t.WM == 1 && t.WN == 4 && t.AM == 2 && t.AN == 4) id = 8;
  if (t.WM == 2 && t.WN == 4 && t.AM == 2 && t.AN == 4) id = 9;
  const int NW = t.WM * t.WN;
  if (id < 0) return false;
  if ((xs || a0.Ys) && !(id == 3 || id == 4 || id == 7 || id == 8 || id == 9)) return false;   // tiles instantiated with the split-resident paths
  if (xs && a0.Ys) return false;                                                    // (one side at a time)
  const int BM = t.WM * t.AM * 32, BN = t.WN * t.AN * 32;
  const long long nblk = (long long)((a.Tout + BN - 1) / BN) * ((a.Co + BM - 1) / BM);
  static const int min_blk = exp_int("RVC_X3_MINBLK", 250);
  static const int min_blk2d = exp_int("RVC_X3_MINBLK2D", 20);   // deep U-Net levels: bf16x3 + split-K beats fp32 + split-K
  if (nblk < (a.Wd > 0 ? min_blk2d : min_blk)) return false;   // under-filled grids go to the fp32 kernel's split-K path
  if (t.WM == 2 && t.WN == 2 && batch == 1) {
    // the generator's stride-1 convolutions: software-pipelined kernel (conv_x3p.hip)
    dim3 g;
    // (the stride-2 mode exists for 128 x 128 tiles only: a shorter layer that would take 64-row tiles uses it as long as >= 150 tiles remain)
    const bool s2_up = a.stride == 2 && a.ktaps == 3 && a.Co >= 128 && !(t.AM == 2 && t.AN == 2) &&
                       (long long)((a.Co + 127) / 128) * ((a.Tout + 127) / 128) >= 150;
    const int pam = s2_up ? 2 : t.AM, pan = s2_up ? 2 : t.AN;
    if (!s2_up && conv_x3q_try(a, t.AM, t.AN, s, g, true)) {
      // the ResBlock convolutions: persistent workgroups (conv_x3q.hip)
      if (dry) return true;
      ProfTicket tk = conv_prof_begin(s);
      RVC_REQUIRE(conv_x3q_try(a, t.AM, t.AN, s, g, false), "conv_x3q_try accepted the layer in its dry run and declined the launch");
      conv_prof_end(tk, s, flops, 14 + id, conv_alg_bytes(a, batch), &a, (long long)g.x * g.y, 6 << 4);
      return true;
    }
    if (a.h2) return false;                                  // fp16x2 images are the persistent kernel's alone
    if (conv_x3p_try(a, pam, pan, s, g, true)) {
      if (dry) return true;
      ProfTicket tk = conv_prof_begin(s);
      RVC_REQUIRE(conv_x3p_try(a, pam, pan, s, g, false), "conv_x3p_try accepted the layer in its dry run and declined the launch");
      conv_prof_end(tk, s, flops, 14 + (s2_up ? 3 : id), conv_alg_bytes(a, batch), &a, (long long)g.x * g.y, 1 << 4);
      return true;
    }
  }
  if (a.h2) return false;
  static const int x3_split_blk = exp_int("RVC_X3_SPLITK_BLK", 600);
  int P = (BN - 1) * a.stride + (a.ktaps - 1) * a.dil + 1;
  if (a.Wd > 0) {
    // tile = whole image rows or a power-of-two fraction of one row (Wd is a power of two)
    a.BWd = BN < a.Wd ? BN : a.Wd; a.BH = BN < a.Wd ? 1 : BN / a.Wd; a.PW = a.BWd + KW2 - 1;
    P = (a.BH + KH2 - 1) * a.PW;
    a.magPW = (unsigned)((0x100000000ULL + a.PW - 1) / a.PW);
  }
  const int Pm = xs ? ((P + 63) & ~63) : (P + a.stride - 1) / a.stride;
  a.ni = (P + 63) / 64;
  const int nchunk = a.Ci / 16;
  // chunks per stage: short reductions per chunk (k <= 3) take several chunks per stage so that a stage outlasts its DMA
  int NC = a.ktaps == 1 ? 4 : (a.ktaps <= 3 ? 2 : 1);
  while (NC > 1 && (nchunk % NC != 0 || (!xs && (NC * 2 * a.ni + NW - 1) / NW > x3_slots(BN, NW)))) NC >>= 1;
  if (!xs && (NC * 2 * a.ni + NW - 1) / NW > x3_slots(BN, NW)) return false;
  // LDS budget per workgroup: 53 KiB = three workgroups per CU for the tiles whose registers allow it (<= 170 VGPRs), two for the
  // 8-accumulator tiles.  Measured: occupancy matters more than stage length (one 156 KiB workgroup per CU with 3x longer stages:
  // +32 % time; three 128x128 workgroups instead of two: -12 %).  X double-buffered when that still leaves >= 2 taps per stage.
  static const int budget_kb = exp_int("RVC_X3_LDS_KB", 53);
  static const int budget8_kb = exp_int("RVC_X3_LDS8_KB", 156);   // 8-wave workgroups own the CU's LDS
  const int budget = (NW == 8 ? budget8_kb : budget_kb) * 1024;
  // weight slabs: a ring of NS: the DMA of a slab is issued NS - 1 stages before its MFMAs, waited for with a counted vmcnt and published
  // with a barrier that does not drain the queue.  Measured: NS = 3 / 4 lose to NS = 2 (C128 k11 490 vs 433 us): the LDS they take
  // halves the taps per stage, and the per-stage costs (DMA issue, barrier) outweigh the ~400 cycles of DMA wait they would hide.
  static const int wbufs_env = exp_int("RVC_X3_WBUFS", 2);
  int NS = wbufs_env < 2 ? 2 : (wbufs_env > 4 ? 4 : wbufs_env);
  int xbufs = 2, xbytes = 0, ktmax = 0;
  for (;;) {
    const int per_tap = NS * NC * 2 * BM * 32;              // NS slabs x NC chunks x {hi, lo} x BM rows x 32 B
    xbufs = 2; xbytes = (xbufs * NC * 2 * a.stride * Pm * 32 + 1023) & ~1023;
    ktmax = (budget - xbytes) / per_tap;
    if (!xs && ktmax < 2 && a.ktaps > ktmax && a.ktaps > 1) { xbufs = 1; xbytes = (NC * 2 * a.stride * Pm * 32 + 1023) & ~1023; ktmax = (budget - xbytes) / per_tap; }
    if (ktmax >= 1) break;
    if (NS > 2) { --NS; continue; }
    if (NC == 1) break;
    NC >>= 1;
  }
  if (ktmax < 1) return false;
  if (ktmax > a.ktaps) ktmax = a.ktaps;
  const int ntb = (a.ktaps + ktmax - 1) / ktmax;
  a.KT = (a.ktaps + ntb - 1) / ntb;                        // balanced tap blocks
  if (dry) return true;
  a.CK = 16; a.nchunk = nchunk; a.NC = NC; a.WROW = P; a.xbufs = xbufs; a.ksplit = 1; a.partial = nullptr;
  static const int xcd_env = exp_int("RVC_X3_XCD", 1);
  a.wbufs = NS; a.xcd_remap = xcd_env;
  // split-K (k = 1 GEMMs on small grids): every stage of such a workgroup is a dependent global -> LDS round trip, so slicing the
  // reduction over S workgroups shortens the chain and puts more of them on a CU; partials are reduced in a fixed order
  int S = 1;
  // (measured on the HuBERT projections: K = 3072 203 -> 120 us, K = 768 unchanged or worse: only deep reductions are split)
  static const int split_blk2d = exp_int("RVC_X3_SPLITK_BLK2D", 400);
  const int split_blk = a.Wd > 0 ? split_blk2d : x3_split_blk;
  if (split_blk > 0 && a.ostride == 1 && ((a.Wd == 0 && a.ktaps == 1 && a.Ci >= 2048) || (a.Wd > 0 && a.Ci >= 64)) && nblk < split_blk) {
    const int ngroups = nchunk / NC;
    S = (int)((2LL * split_blk + nblk - 1) / nblk);
    if (S > 8) S = 8;
    if (S > ngroups / (a.Wd > 0 ? 1 : 3)) S = ngroups / (a.Wd > 0 ? 1 : 3);
    if (S < 1) S = 1;
    while (S > 1 && ((ngroups + S - 1) / S) * (S - 1) >= ngroups) --S;
  }
  a.ksplit = S; a.ldP = (a.Tout + 31) & ~31;
  if (S > 1) a.partial = (float*)stream_scratch(s, 0, (size_t)S * a.Co * a.ldP * sizeof(float));
  const size_t lds = (size_t)xbytes + (size_t)NS * NC * a.KT * 2 * BM * 32;
  dim3 grid((unsigned)((a.Tout + BN - 1) / BN), (unsigned)((a.Co + BM - 1) / BM), (unsigned)S);
  ProfTicket tk = conv_prof_begin(s);
  if (xs) {
    switch (id) {
      case 3: launch_x3<2, 2, 2, 2, false, true>(a, grid, lds, s); break;
      case 4: launch_x3<2, 2, 1, 4, false, true>(a, grid, lds, s); break;
      case 7: launch_x3<2, 2, 2, 4, false, true>(a, grid, lds, s); break;
      case 9: launch_x3<2, 4, 2, 4, false, true>(a, grid, lds, s); break;
      default: launch_x3<1, 4, 2, 4, false, true>(a, grid, lds, s); break;
    }
  } else if (a.Ys) {
    switch (id) {
      case 3: launch_x3<2, 2, 2, 2, false, false, true>(a, grid, lds, s); break;
      case 4: launch_x3<2, 2, 1, 4, false, false, true>(a, grid, lds, s); break;
      case 7: launch_x3<2, 2, 2, 4, false, false, true>(a, grid, lds, s); break;
      case 9: launch_x3<2, 4, 2, 4, false, false, true>(a, grid, lds, s); break;
      default: launch_x3<1, 4, 2, 4, false, false, true>(a, grid, lds, s); break;
    }
  } else
  switch (id) {
    case 0: launch_x3<1, 4, 1, 4>(a, grid, lds, s); break;
    case 1: launch_x3<1, 4, 1, 2>(a, grid, lds, s); break;
    case 2: launch_x3<1, 4, 1, 1>(a, grid, lds, s); break;
    case 3: launch_x3<2, 2, 2, 2>(a, grid, lds, s); break;
    case 4: launch_x3<2, 2, 1, 4>(a, grid, lds, s); break;
    case 5: launch_x3<2, 2, 1, 2>(a, grid, lds, s); break;
    case 7: launch_x3<2, 2, 2, 4>(a, grid, lds, s); break;
    case 8: launch_x3<1, 4, 2, 4>(a, grid, lds, s); break;
    case 9: launch_x3<2, 4, 2, 4>(a, grid, lds, s); break;
    default: launch_x3<2, 2, 1, 1>(a, grid, lds, s); break;
  }
  if (S > 1) splitk_reduce_launch(a, S, 1, s);
  conv_prof_end(tk, s, flops, 14 + id, conv_alg_bytes(a, batch), &a, (long long)grid.x * grid.y * grid.z);
  return true;
}

// ---------------------------------------------------------------------------- fused ResBlock pair (narrow generator stages)
// The C = 32 stage of the generator is HBM-bound on the unfused kernels (each conv reads + writes [C][T] and the second one reads the
// residual too: 5 tensor passes per pair); fused, a pair reads x (+ halo) twice (tile + residual, the second from L2) and writes y.
// dry_only: answers whether the pair would run in the fp16x2 arithmetic on the LDS-resident-weights kernel (nothing is launched)
bool conv_x3_pair_try(const ConvLayer& c1, const ConvLayer& c2, hipStream_t s, const float* X, long long ldX, int T, float* Y, long long ldY,
                      const ConvEpilogue& e2, bool dry_only) {
  static const bool on = (exp_int("RVC_PAIR", 1) != 0);
  static const bool pair64 = (exp_int("RVC_PAIR64", 0) != 0);      // experiment: 64-channel stage (2 WGs per CU)
  if (!on || !conv_x3_enabled() || !c1.Wx_ || !c2.Wx_) return false;
  const int C = c1.Co, k = c1.k;
  if (c1.mode != 1 || c2.mode != 1 || c1.groups != 1 || c2.groups != 1 || c1.stride != 1 || c2.stride != 1 || c1.tconv_u || c2.tconv_u) return false;
  if (c1.Ci != C || c2.Ci != C || c2.Co != C || c2.k != k || c2.dil != 1 || (k & 1) == 0 || !(C == 32 || C == 64)) return false;
  if (c1.pad != (k - 1) / 2 * c1.dil || c2.pad != (k - 1) / 2) return false;                      // "same" convolutions
  if (e2.pre_act != ACT_LRELU || e2.act != ACT_NONE || e2.bias_override || e2.tout_limit || e2.R != X) return false;
  if ((double)C * (double)ldX * 4.0 >= 2147483648.0 || (double)C * (double)ldY * 4.0 >= 2147483648.0) return false;
  static const int bn_env = exp_int("RVC_PAIR_BN", 256);
  const int BM = C;
  const int BN = (C == 64 || bn_env == 128) ? 128 : 256;
  const int P2 = (k - 1) / 2, P1 = c1.pad;
  const int NO = BN - 2 * P2;
  if ((long long)(T + NO - 1) / NO < 512) return false;          // short sequences: the unfused path fills the chip better
  ConvArgsX a{};
  a.X = X; a.ldX = ldX; a.Y = Y; a.ldY = ldY; a.W = nullptr; a.bias = c2.bd_; a.bias1 = c1.bd_;
  a.R = e2.R; a.ldR = e2.ldR; a.pre_act = ACT_LRELU; a.pre_slope = 0.1f; a.fuse_slope = e2.pre_slope;
  a.act = ACT_NONE; a.act_slope = 0.f; a.act_before_res = 0; a.out_scale = e2.out_scale; a.accumulate = e2.accumulate;
  a.Ci = C; a.Co = C; a.CoP = c1.CoP; a.Tin = T; a.Tout = T; a.Wd = 0; a.ktaps = k; a.kreal = k; a.dil = c1.dil; a.stride = 1; a.pad = P1;
  a.ostride = 1; a.orows = C; a.up2 = 0;
  a.xBatch = a.wBatch = a.yBatch = a.rBatch = 0; a.bBatch = 0; a.wxBatch = 0;
  a.Wx = reinterpret_cast<const unsigned char*>(c1.Wx_); a.Wx2 = reinterpret_cast<const unsigned char*>(c2.Wx_); a.CoPx = c1.CoPx;
  a.fuse_p2 = P2;
  if (c1.CoPx != c2.CoPx) return false;
  if (C == 32 && c1.Wh_ && c2.Wh_ && conv_set_pair_arithmetic(-1)) {
    // fp16x2 pair arithmetic: persistent workgroups with both weight sets resident in LDS (conv_rbh.hip)
    ConvArgsX h = a;
    h.Wx = reinterpret_cast<const unsigned char*>(c1.Wh_); h.Wx2 = reinterpret_cast<const unsigned char*>(c2.Wh_); h.h2 = 1;
    dim3 gh;
    if (conv_rbh_try(h, T, s, gh, true)) {
      if (dry_only) return true;
      ProfTicket tk = conv_prof_begin(s);
      RVC_REQUIRE(conv_rbh_try(h, T, s, gh, false), "conv_rbh_try accepted the pair in its dry run and declined the launch");
      const double bytes = 4.0 * ((double)C * T * (2.0 + (e2.accumulate ? 1.0 : 0.0)) + 2.0 * C * C * k);
      conv_prof_end(tk, s, 2.0 * 2.0 * (double)C * C * k * T, 14 + 1, bytes, &h, (long long)gh.x, 1 | (5 << 4));
      return true;
    }
  }
  if (dry_only) return false;
  {
    // the software-pipelined fused pair (conv_x3p.hip): 32 and 64 channels
    static const int xcd_env = exp_int("RVC_X3_XCD", 1);
    a.xcd_remap = xcd_env;
    dim3 gpf;
    if (conv_x3pf_try(a, T, s, gpf, true)) {
      ProfTicket tk = conv_prof_begin(s);
      conv_x3pf_try(a, T, s, gpf, false);
      const double bytes = 4.0 * ((double)C * T * (2.0 + (e2.accumulate ? 1.0 : 0.0)) + 2.0 * C * C * k);   // x ONCE (the residual is the same tensor), y [, previous y]
      conv_prof_end(tk, s, 2.0 * 2.0 * (double)C * C * k * T, 14 + (C == 32 ? 1 : 5), bytes, &a, (long long)gpf.x, 1 | (3 << 4));
      return true;
    }
    if (C == 64 && !pair64) return false;
  }
  const int P = BN + 2 * P1;                                      // staged input columns
  a.ni = (P + 63) / 64;
  const int nchunk = C / 16, NC = nchunk;                         // every channel of the tile resident: one chunk group
  if ((NC * 2 * a.ni + 3) / 4 > (C == 64 ? 7 : 5)) return false;
  const int xbytes = (NC * 2 * P * 32 + 1023) & ~1023;
  // three workgroups per CU with single-tap stages beat two with 4-tap stages (k3 233 -> 196 us, k7 279 -> 233, k11 352 -> 308):
  // occupancy is what hides the per-stage latencies of this narrow tile
  static const int budget_kb_env = exp_int("RVC_PAIR_LDS_KB", 53);
  const int budget_kb = C == 64 ? 80 : budget_kb_env;
  const int per_tap = 2 * NC * 2 * BM * 32;
  int ktmax = (budget_kb * 1024 - xbytes) / per_tap;
  if (ktmax < 1) return false;
  if (ktmax > k) ktmax = k;
  const int ntb = (k + ktmax - 1) / ktmax;
  a.KT = (k + ntb - 1) / ntb;
  a.CK = 16; a.nchunk = nchunk; a.NC = NC; a.WROW = P; a.xbufs = 1; a.ksplit = 1; a.partial = nullptr; a.ldP = 0;
  static const int xcd_env = exp_int("RVC_X3_XCD", 1);
  a.wbufs = 2; a.xcd_remap = xcd_env;
  const size_t lds = (size_t)xbytes + (size_t)2 * NC * a.KT * 2 * BM * 32;
  dim3 grid((unsigned)((T + NO - 1) / NO), 1, 1);
  ProfTicket tk = conv_prof_begin(s);
  if (C == 64) launch_x3<2, 2, 1, 2, true>(a, grid, lds, s);
  else if (BN == 128) launch_x3<1, 4, 1, 1, true>(a, grid, lds, s);
  else launch_x3<1, 4, 1, 2, true>(a, grid, lds, s);
  // algorithmic traffic of the pair: x read (the residual is x itself: one tensor, counted once), y write (+ previous y when accumulating) + both weight sets
  const double bytes = 4.0 * ((double)C * T * (2.0 + (e2.accumulate ? 1.0 : 0.0)) + 2.0 * C * C * k);   // x ONCE (the residual is the same tensor), y [, previous y]
  conv_prof_end(tk, s, 2.0 * 2.0 * (double)C * C * k * T, 14 + 1, bytes, &a, (long long)grid.x, 1);
  return true;
}

}  // namespace rvc
