// Software-pipelined bf16x3 split-MFMA Conv1d for the stride-1 convolutions of the generator (gfx950 only).
//
// Same arithmetic, LDS / image layouts and epilogues as conv_x3.hip (hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16, half-plane
// rows); what differs is the order in which ONE WAVE does its work.  Measured on the staged kernel (C128 k11, 128 x 256 tiles): a
// workgroup alone on a CU needs 241 k cycles for 67.6 k cycles of MFMA issue - a wave issues in order, and every (chunk, tap) unit was
// DMA issue -> wait -> barrier -> address arithmetic -> operand reads -> wait -> 24 MFMAs, with the tile's input conversion, its first
// loads and its residual reads in bursts in between; a second workgroup on the CU hides part of that (338 k for two tiles), a third
// does not fit the registers.  Here the unit is the pipeline stage and nothing waits for what it has just asked for:
//   * the three MFMA groups of a unit run in the order (hi_w * lo_x), (lo_w * hi_x), (hi_w * hi_x); the operands of the second group
//     are requested before the first is issued, those of the NEXT unit's first group before the third (one spare register set for the
//     hi weights): every ds_read_b128 has a group of 8 MFMAs (256 cycles) to land;
//   * weights go through a ring of 3 - 4 single-unit slots, requested 2 - 3 units ahead by LDS-DMA; one barrier per unit, between the
//     second and the third group, publishes the next unit's slot while MFMAs are still queued;
//   * the fp32 input of chunk c + 1 is converted and stored one 8-channel x 64-position slot at a time, spread over the units of chunk
//     c, and the slot's registers are refilled at once with chunk c + 2 (a whole chunk of latency); split-resident inputs are DMA'd
//     a chunk ahead;
//   * residual + bias are loaded straight into the accumulators at the start of the tile (when no activation sits between them and
//     the sum), so the epilogue is a burst of stores with no load in front of it.
#include "conv_x3_dev.h"

#ifndef RVC_X3P_R24
#define RVC_X3P_R24 4          // weight ring slots of the 128 x 256 tile (two workgroups per CU: 80 KB each)
#endif

namespace rvc {

// conversion schedule: slot s (of XS) of the next chunk is converted during tap (s * KT) / XS
constexpr int x3p_cv(int t, int KT, int XS) {
  t = ((t % KT) + KT) % KT;
  for (int s = 0; s < XS; ++s) if ((s * KT) / XS == t) return 1;
  return 0;
}
constexpr int x3p_cvsum(int t0, int t1, int KT, int XS) { int n = 0; for (int t = t0; t <= t1; ++t) n += x3p_cv(t, KT, XS); return n; }

#ifdef RVC_X3P_SETPRIO
#define X3P_PRIO(n) __builtin_amdgcn_s_setprio(n)
#else
#define X3P_PRIO(n) do {} while (0)
#endif
template <int N> __device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int T, int N, class F> __device__ __forceinline__ void static_for(F& f) {
  if constexpr (T < N) { f(std::integral_constant<int, T>{}); static_for<T + 1, N>(f); }
}
#ifdef RVC_X3P_CHECK
__device__ int g_x3p_bad;       // waits whose compile-time count exceeded the exact run-time one (must stay 0)
int conv_x3p_check_read() { int v = 0, z = 0; (void)hipDeviceSynchronize(); (void)hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_x3p_bad), sizeof(int)); (void)hipMemcpyToSymbol(HIP_SYMBOL(g_x3p_bad), &z, sizeof(int)); return v; }
#define X3P_CHECK(N, exact) do { if ((N) > (exact) && (threadIdx.x & 63) == 0) atomicAdd(&g_x3p_bad, 1); } while (0)
#else
int conv_x3p_check_read() { return -1; }
#define X3P_CHECK(N, exact) do {} while (0)
#endif

#ifdef RVC_CONV_TIMING
__device__ unsigned long long g_x3p_timing[8];   // [0] tiles, [1] prologue, [2] compute between barriers, [3] weight wait, [4] barrier, [5] epilogue, [6] total
void conv_x3p_timing_read(unsigned long long* out8, bool reset) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_x3p_timing), sizeof(unsigned long long) * 8);
  if (reset) { unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_x3p_timing), z, sizeof(z)); }
}
#define XPTICK() ((long long)__builtin_readcyclecounter())
#define XPACC(i, v) do { xpt[i] += (v); } while (0)
#else
void conv_x3p_timing_read(unsigned long long* out8, bool) { for (int i = 0; i < 8; ++i) out8[i] = 0; }
#define XPTICK() 0ll
#define XPACC(i, v) do {} while (0)
#endif

// KT = taps (compile-time: the units of a chunk are unrolled, so every vmcnt wait is an immediate - see the counting rules at the waits).
// S2: stride 2 (HuBERT's feature encoder, k = 3): the staged input keeps one sub-plane per input phase (even / odd positions), tap t reads
// phase t & 1 at row offset t >> 1 - unit-stride operand reads, no wasted MFMAs.
template <int AM, int AN, int KT, bool XSPLIT, bool YSPLIT, bool S2 = false>
__global__ __launch_bounds__(256, (AM * AN >= 8) ? 2 : 3) void conv_x3p_kernel(const ConvArgsX p) {
  static_assert(!S2 || (!XSPLIT && !YSPLIT), "strided: fp32 in, fp32 out");
  constexpr int WM = 2, WN = 2, NW = 4;
  constexpr int BM = WM * AM * 32, BN = WN * AN * 32, RB = BM / 32;
  constexpr int R = (AM == 2 && AN == 4) ? RVC_X3P_R24 : 3;  // weight slots in the ring (LDS of two / three workgroups per CU)
  constexpr int XS = 3;                                     // fp32 staging slots per wave (8 channels x 64 positions each): P <= 384
  constexpr int NPW = 2 * RB / NW;                          // weight pieces per unit and wave (BM = 64: 1, BM = 128: 2)
  constexpr int NPX = (BN + 64) / 64;                       // split-resident input: pieces per chunk and wave (Pm = BN + 64)
  constexpr int wslot = 2 * BM * 32;                        // bytes of one weight slot: [hi | lo][half][BM rows][16 B]
  static_assert(RB == 2 || RB == 4, "64- or 128-row tiles");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem3p[];
  const int P = p.WROW;                                     // staged input positions: BN + (KT - 1) * dil
  const int Pm2 = (P + 1) >> 1;                              // (stride 2) rows of one phase sub-plane
  const int Pm = XSPLIT ? BN + 64 : (S2 ? 2 * Pm2 : P);
  const int xplane = Pm * 32, xhalf = xplane >> 1, xbuf = 2 * xplane;
  unsigned char* Xs = smem3p;
  unsigned char* Ws = smem3p + ((2 * xbuf + 1023) & ~1023);

  const int tid0 = threadIdx.x;
  int lane = tid0 & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;
  const unsigned tile = p.xcd_remap ? xcd_tile(blockIdx.x + blockIdx.y * gridDim.x, gridDim.x * gridDim.y) : blockIdx.x + blockIdx.y * gridDim.x;
  // Row tiles fastest: the gridDim.y workgroups that share one input column range (and differ in their weight rows) are neighbours in the
  // XCD's run of tiles, so the input tile is read from HBM once and from L2 gridDim.y - 1 times.  (Column-fastest, as the staged kernel
  // numbers them, re-read the input once per row tile: the stride-2 layers of HuBERT fetched 972 MB for 318 MB algorithmic, the 10x
  // up-sampler 535 for 200.)
  const int tile_x = (int)(tile / gridDim.y), tile_y = (int)(tile - (unsigned)tile_x * gridDim.y);
  const int co0 = tile_y * BM, n0 = tile_x * BN;
  const int nck = p.nchunk;                                  // >= 3 (host)
  const int bx = (S2 ? 2 * n0 : n0) - p.pad;
  const int ni = p.ni;                                       // 64-position groups of the staged row
  auto tap_off = [&](int t) -> int {                         // byte offset of tap t inside a half-plane
    if constexpr (S2) return ((t & 1) * Pm2 + (t >> 1)) * 16;
    else return t * (p.dil * 16);
  };
  const float pre_slope = p.pre_act == ACT_LRELU ? p.pre_slope : 1.f;
  const __amdgpu_buffer_rsrc_t xrs = make_rsrc(p.X, XSPLIT ? 0u : (unsigned)p.Ci * (unsigned)p.ldX * 4u);

#ifdef RVC_CONV_TIMING
  long long xpt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  const long long t_begin = XPTICK();
  long long t_last = t_begin;
  // ---- accumulators: zero, or residual + bias when nothing but the scale follows the sum (the loads are the oldest VMEM operations of
  // the wave; their latency lies under the prologue's input loads and first weight slots)
  const bool r_init = !YSPLIT && p.R != nullptr && p.act == ACT_NONE;
  f32x16 acc[AM][AN];
  if (r_init) {
    const __amdgpu_buffer_rsrc_t rrs = make_rsrc(p.R, (unsigned)p.orows * (unsigned)p.ldR * 4u);
#pragma unroll
    for (int am = 0; am < AM; ++am)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = co0 + (wm * AM + am) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float bv = (p.bias && m < p.Co) ? p.bias[m] : 0.f;
#pragma unroll
        for (int an = 0; an < AN; ++an) {
          const int n = n0 + (wn * AN + an) * 32 + li;
          acc[am][an][r] = buf_load(rrs, (m < p.Co && n < p.Tout) ? ((unsigned)m * (unsigned)p.ldR + (unsigned)n) * 4u : kOOB) + bv;
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

#ifdef RVC_X3P_CHECK
  int issued = 0, mk_w[8] = {0, 0, 0, 0, 0, 0, 0, 0}, mk_x[XS] = {0, 0, 0}, mk_xs = 0;   // exact bookkeeping (debug build only)
#define X3P_ISSUED(n) (issued += (n))
#else
#define X3P_ISSUED(n) do {} while (0)
#endif

  // ---- weights: unit u = (chunk, tap) is 2 * RB pieces of 1 KiB; consecutive units are consecutive planes of the image
  const unsigned char* wsrc;
  {
    const int hl0 = wave / RB, r0 = (wave % RB) * 64 + lane;
    wsrc = p.Wx + (long long)hl0 * p.CoPx * 32 + ((long long)(r0 / BM) * p.CoPx + co0 + (r0 % BM)) * 16;
  }
  const long long wstep = (long long)p.CoPx * 64;            // one unit = hi + lo planes
  int slw = 0;                                               // slot of the next unit to request
#ifdef RVC_X3P_CHECK
  int uw = 0;
#endif
  auto issue_w = [&]() {
    unsigned char* dst = Ws + slw * wslot + wave * 1024;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)wsrc, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    if constexpr (RB == 4)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + (long long)p.CoPx * 32),
                                       (__attribute__((address_space(3))) void*)(dst + NW * 1024), 16, 0, 0);
    wsrc += wstep; slw = slw + 1 == R ? 0 : slw + 1;
    X3P_ISSUED(NPW);
#ifdef RVC_X3P_CHECK
    mk_w[uw & 7] = issued; ++uw;
#endif
  };

  // ---- fp32 input: slot s of this wave = 8 channels (one half-plane) x 64 positions of the staged tile.  Slots beyond the tile are
  // loaded through out-of-range offsets (no memory traffic) so that every wave issues the same number of operations.
  float xr[XS][8];
  auto slot_geom = [&](int s, int& hb, int& q) -> bool {
    const int t = wave + NW * s;
    hb = t >= ni ? 1 : 0;
    q = (t - hb * ni) * 64 + lane;
    return t < 2 * ni && q < P;
  };
  auto load_slot = [&](int s, int chunk) {
    int hb, q;
    const bool ok = slot_geom(s, hb, q);
    const int x = bx + q;
    const unsigned voff = (ok && x >= 0 && x < p.Tin) ? (unsigned)x * 4u : kOOB;
    const unsigned c0 = (unsigned)(chunk * 16 + hb * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) xr[s][j] = buf_load(xrs, voff, (c0 + j) * (unsigned)p.ldX * 4u);
    X3P_ISSUED(8);
#ifdef RVC_X3P_CHECK
    mk_x[s] = issued;
#endif
  };
  auto store_slot = [&](int s, int xb) {
    int hb, q;
    if (slot_geom(s, hb, q)) {
      u32x4 hi, lo;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float a = xr[s][2 * j], b = xr[s][2 * j + 1];
        unsigned h_, l_;
        split2(fmaxf(a, a * pre_slope), fmaxf(b, b * pre_slope), h_, l_);       // input activation: leaky ReLU (slope 1 = identity)
        hi[j] = h_; lo[j] = l_;
      }
      const int row = S2 ? (q & 1) * Pm2 + (q >> 1) : q;
      unsigned char* d = Xs + xb * xbuf + hb * xhalf + row * 16;
      *reinterpret_cast<u32x4*>(d) = hi;
      *reinterpret_cast<u32x4*>(d + xplane) = lo;
    }
  };
  // ---- split-resident input: chunk -> X buffer by DMA, 1 KiB (64 positions of one half-plane) per wave-instruction, NPX per wave
  auto issue_x = [&](int chunk, int xb) {
    constexpr int pph = (BN + 64) / 64;                          // pieces per half-plane
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
      const int pi = wave + NW * i;
      const int hp = pi / pph, j = pi - hp * pph;               // half-plane = (hi | lo) * 2 + half
      const long long row = (long long)(chunk * 4 + hp) * p.xsTp + (bx + kSplitMargin) + (long long)j * 64;
      const unsigned char* src = p.Xs + row * 16 + lane * 16;
      unsigned char* dst = Xs + xb * xbuf + hp * xhalf + j * 1024;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
    X3P_ISSUED(NPX);
#ifdef RVC_X3P_CHECK
    mk_xs = issued;
#endif
  };
  const bool x_edge = XSPLIT && (bx < 0 || bx + P > p.Tin);
  auto zero_edges = [&](int xb) {                                // rows outside the sequence are the convolution's zero padding
    for (int q = tid0; q < P; q += NW * 64) {
      const int t = bx + q;
      if (t >= 0 && t < p.Tin) continue;
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
        unsigned char* r = Xs + xb * xbuf + pl * xplane + q * 16;
        *reinterpret_cast<u32x4*>(r) = u32x4{0u, 0u, 0u, 0u}; *reinterpret_cast<u32x4*>(r + xhalf) = u32x4{0u, 0u, 0u, 0u};
      }
    }
  };

  // ---- operand addresses: lane (i, half) reads the 16-B row i of its half-plane; column / row blocks are 512 B apart
  const int aoff = lh * (BM * 16) + ((wm * AM) * 32 + li) * 16;
  const int boff = lh * xhalf + ((wn * AN) * 32 + li) * 16;

  // ---- prologue.  Issue order (the waits below count on it): [residual] input of chunk 0 | weight units 0 .. R - 2 | input of chunk 1
  if constexpr (XSPLIT) {
    issue_x(0, 0);
  } else {
#pragma unroll
    for (int s = 0; s < XS; ++s) load_slot(s, 0);
  }
#pragma unroll
  for (int i = 0; i < R - 1; ++i) issue_w();
  if constexpr (XSPLIT) {
    issue_x(1, 1);
    wait_vmcnt<(R - 2) * NPW + NPX>();                           // chunk 0 and unit 0 have landed (younger: units 1 .. R - 2, chunk 1)
  } else {
    wait_vmcnt<(R - 1) * NPW>();                                 // chunk 0's loads (younger: the weight units)
#pragma unroll
    for (int s = 0; s < XS; ++s) store_slot(s, 0);
#pragma unroll
    for (int s = 0; s < XS; ++s) load_slot(s, 1);
    wait_vmcnt<(R - 2) * NPW + 8 * XS>();                        // unit 0 (younger: units 1 .. R - 2, chunk 1's loads)
  }
#pragma unroll
  for (int am = 0; am < AM; ++am)
#pragma unroll
    for (int an = 0; an < AN; ++an) asm volatile("" : "+v"(acc[am][an]));     // (the residual loads are complete from here on: no wait for them inside the loop)
  lds_barrier();
  if (x_edge) { zero_edges(0); lds_barrier(); }

  u32x4 ah[AM], ahn[AM], al[AM], bh[AN], bl[AN];
  {
    const unsigned char* wa = Ws + aoff;
    const unsigned char* xa = Xs + boff;
#pragma unroll
    for (int am = 0; am < AM; ++am) ah[am] = *reinterpret_cast<const u32x4*>(wa + am * 512);
#pragma unroll
    for (int an = 0; an < AN; ++an) bl[an] = *reinterpret_cast<const u32x4*>(xa + xplane + an * 512);
  }

  int sl = 0;                                                    // weight slot of the current unit
  t_last = XPTICK(); XPACC(1, t_last - t_begin);
  for (int c = 0; c < nck; ++c) {
    const int xb = c & 1;
    const bool tail2 = c + 2 >= nck, tail1 = c + 1 >= nck;       // no chunk c + 2 / c + 1
    auto unit = [&](auto tc) {
      constexpr int T = decltype(tc)::value;
      // ---- operands of the second group
      {
        const unsigned char* wa = Ws + sl * wslot + BM * 32 + aoff;
        const unsigned char* xa = Xs + xb * xbuf + tap_off(T) + boff;
#pragma unroll
        for (int am = 0; am < AM; ++am) al[am] = *reinterpret_cast<const u32x4*>(wa + am * 512);
#pragma unroll
        for (int an = 0; an < AN; ++an) bh[an] = *reinterpret_cast<const u32x4*>(xa + an * 512);
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- group 1: hi_w * lo_x
      X3P_PRIO(1);
#pragma unroll
      for (int am = 0; am < AM; ++am)
#pragma unroll
        for (int an = 0; an < AN; ++an)
          acc[am][an] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[am]), __builtin_bit_cast(bf16x8, bl[an]), acc[am][an], 0, 0, 0);
      X3P_PRIO(0);
      // ---- input of chunk c + 1: slot s is converted and stored during tap (s * KT) / XS, its registers refilled with chunk c + 2
      if constexpr (!XSPLIT) {
        auto stage = [&](auto sc) {
          constexpr int s = decltype(sc)::value;
          if constexpr ((s * KT) / XS == T) {
            if (!tail1) {
              // operations issued after this slot's loads: chunk 0 - the later slots of the prologue's batch, one unit of weights per tap so
              // far, the refills of the earlier slots; otherwise KT units of weights and the refills of the other slots (of the later
              // ones only when chunk c + 1 is the last)
              constexpr int T0 = 8 * (XS - 1) + NPW * T, TS = NPW * KT + 8 * (XS - 1), TL = NPW * KT + 8 * (XS - 1 - s);
              if (c == 0) { X3P_CHECK(T0, issued - mk_x[s]); wait_vmcnt<T0>(); }
              else if (tail2) { X3P_CHECK(TL, issued - mk_x[s]); wait_vmcnt<TL>(); }
              else { X3P_CHECK(TS, issued - mk_x[s]); wait_vmcnt<TS>(); }
              store_slot(s, xb ^ 1);
              if (!tail2) load_slot(s, c + 2);
            }
          }
        };
        static_for<0, XS>(stage);
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- group 2: lo_w * hi_x
      X3P_PRIO(1);
#pragma unroll
      for (int am = 0; am < AM; ++am)
#pragma unroll
        for (int an = 0; an < AN; ++an)
          acc[am][an] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al[am]), __builtin_bit_cast(bf16x8, bh[an]), acc[am][an], 0, 0, 0);
      X3P_PRIO(0);
      __builtin_amdgcn_sched_barrier(0);
      // ---- next unit: its weight slot (and, at a chunk boundary, its input buffer) published; the slot of unit u - 1 refilled
      constexpr bool last_tap = T + 1 == KT;
      if (!(last_tap && tail1)) {
        // operations issued after the pieces of unit u + 1 (requested R - 2 units ago): unit u + 2 (R = 4) and the input traffic of the
        // window - fp32: a slot refill (8 loads) in every unit that converts one; split: the chunk pieces issued after a last tap
        // General R: the pieces of unit u + 1 were requested R - 2 units ago; younger are units u + 2 .. u + R - 2 and the input traffic
        // issued since: fp32 - the refill (8 loads) of every converting unit u - (R - 3) .. u; split - the chunk pieces that follow the
        // weight request of a last tap, i.e. when this tap is one of 0 .. R - 3.
        constexpr int SX = (R - 3) * NPW + (XSPLIT ? (T <= R - 3 ? NPX : 0) : 8 * x3p_cvsum(T - (R - 3), T, KT, XS));
        // chunk 0, units 0 .. R - 3: their weights were requested in the prologue, before chunk 1's input batch
        constexpr int S0 = (XSPLIT || T > R - 3) ? SX : (R - 3) * NPW + 8 * XS + 8 * x3p_cvsum(0, T, KT, XS);
        // last two chunks (fp32) / last chunk (split): no input traffic any more; the last units request no weights either
        constexpr int nyw = KT - 2 - T < 0 ? 0 : (KT - 2 - T > R - 3 ? R - 3 : KT - 2 - T);
        constexpr int SL = nyw * NPW, SL2 = (R - 3) * NPW;
#ifdef RVC_X3P_CHECK
        const int exact = issued - mk_w[(c * KT + T + 1) & 7];
#endif
        const long long ta = XPTICK();
        if (tail1) { X3P_CHECK(SL, exact); wait_vmcnt<SL>(); }
        else if (XSPLIT ? false : tail2) { X3P_CHECK(SL2, exact); wait_vmcnt<SL2>(); }
        else if (c == 0) { X3P_CHECK(S0, exact); wait_vmcnt<S0>(); }
        else { X3P_CHECK(SX, exact); wait_vmcnt<SX>(); }
        const long long tb = XPTICK();
        lds_barrier();
        const long long tcc = XPTICK();
        XPACC(2, ta - t_last); XPACC(3, tb - ta); XPACC(4, tcc - tb); t_last = tcc;
        const int sn = sl + 1 == R ? 0 : sl + 1;
        if (XSPLIT && last_tap && x_edge) { zero_edges(xb ^ 1); lds_barrier(); }
        if (!(tail1 && T + R - 1 >= KT)) issue_w();               // unit u + R - 1 into the slot unit u - 1 was read from
        if constexpr (XSPLIT && last_tap) { if (!tail2) issue_x(c + 2, xb); }   // this chunk's buffer is free: every wave is past its last read
        const unsigned char* wa = Ws + sn * wslot + aoff;
        const unsigned char* xa = Xs + (last_tap ? xb ^ 1 : xb) * xbuf + xplane + (last_tap ? 0 : tap_off(T + 1)) + boff;
#pragma unroll
        for (int am = 0; am < AM; ++am) ahn[am] = *reinterpret_cast<const u32x4*>(wa + am * 512);
#pragma unroll
        for (int an = 0; an < AN; ++an) bl[an] = *reinterpret_cast<const u32x4*>(xa + an * 512);
        sl = sn;
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- group 3: hi_w * hi_x
      X3P_PRIO(1);
#pragma unroll
      for (int am = 0; am < AM; ++am)
#pragma unroll
        for (int an = 0; an < AN; ++an)
          acc[am][an] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[am]), __builtin_bit_cast(bf16x8, bh[an]), acc[am][an], 0, 0, 0);
      X3P_PRIO(0);
#pragma unroll
      for (int am = 0; am < AM; ++am) ah[am] = ahn[am];
    };
    static_for<0, KT>(unit);
  }

  // ---- epilogue
  const long long t_epi = XPTICK();
  XPACC(2, t_epi - t_last);
  if constexpr (YSPLIT) {
    ysplit_epilogue<WM, WN, AM, AN>(p, acc, co0, n0, wm, wn, li, lh);
  } else if (p.ostride != 1) {
    // ConvTranspose1d: row m = co * u + phase goes to Y[co][n * u + phase] (phase-fastest rows: a tile holds every phase of its
    // channels, so its stores complete whole cache lines in L2).  No residual / accumulate on this path (host).  u = 2: a lane's rows
    // come in (phase 0, phase 1) pairs of one channel: one 8-byte store per pair and column.
    const int u = p.ostride;
    const float lslope = p.act == ACT_NONE ? 1.f : (p.act == ACT_RELU ? 0.f : p.act_slope);
    const __amdgpu_buffer_rsrc_t yrs = make_rsrc(p.Y, (unsigned)p.orows * (unsigned)p.ldY * 4u);
#pragma unroll
    for (int am = 0; am < AM; ++am) {
      if (u == 2) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const int m = co0 + (wm * AM + am) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;     // even: phase 0 of channel m / 2
          const int co = m >> 1;
          const float bv = (p.bias && m < p.Co) ? p.bias[co] : 0.f;
#pragma unroll
          for (int an = 0; an < AN; ++an) {
            const int n = n0 + (wn * AN + an) * 32 + li;
            float v0 = acc[am][an][r] + bv, v1 = acc[am][an][r + 1] + bv;
            v0 = fmaxf(v0, v0 * lslope) * p.out_scale; v1 = fmaxf(v1, v1 * lslope) * p.out_scale;
            const long long to = 2LL * n;
            const bool in = m < p.Co && n < p.Tout;
            typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
            const unsigned off = ((unsigned)co * (unsigned)p.ldY + (unsigned)to) * 4u;
            if (in && to + 1 < p.ldY) __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{__float_as_uint(v0), __float_as_uint(v1)}, yrs, (int)off, 0, 0);
            else if (in && to < p.ldY) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v0), yrs, (int)off, 0, 0);
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = co0 + (wm * AM + am) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          const int co = m / u, ph = m - co * u;
          const float bv = (p.bias && m < p.Co) ? p.bias[co] : 0.f;
#pragma unroll
          for (int an = 0; an < AN; ++an) {
            const int n = n0 + (wn * AN + an) * 32 + li;
            const long long to = (long long)n * u + ph;
            float v = acc[am][an][r] + bv;
            v = fmaxf(v, v * lslope) * p.out_scale;
            const bool ok = m < p.Co && n < p.Tout && to < p.ldY;      // ldY doubles as the true output length for interleaved stores
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), yrs, (int)(ok ? ((unsigned)co * (unsigned)p.ldY + (unsigned)to) * 4u : kOOB), 0, 0);
          }
        }
      }
    }
  } else if (r_init) {

    ConvArgsX pe = p;
    pe.R = nullptr; pe.bias = nullptr;                            // already inside the accumulators
    dense_epilogue<WM, WN, AM, AN, 4>(pe, acc, 0, co0, n0, wm, wn, li, lh);
  } else {
    dense_epilogue<WM, WN, AM, AN, 4>(p, acc, 0, co0, n0, wm, wn, li, lh);
  }
#ifdef RVC_CONV_TIMING
  { const long long te = XPTICK(); XPACC(5, te - t_epi); XPACC(6, te - t_begin); XPACC(0, 1);
    if (threadIdx.x == 0) for (int i = 0; i < 8; ++i) atomicAdd(&g_x3p_timing[i], (unsigned long long)xpt[i]); }
#endif
}

// ============================================================================ k = 1 (GEMM): the projections of HuBERT / the text encoder / the flow
// Y[m][n] = sum_k W[m][k] X[k][n] with the fp32 activation X = [K][N] (time contiguous).  The same pipeline with one 16-channel chunk
// as the unit: every unit brings its own input slot (BN = 128 columns x 16 channels = one 8-channel x 64-column piece per wave) and its
// own weight slot; both go through rings of three.  A wave keeps four input slots in flight in registers (loaded four units ahead,
// converted two units ahead of their use), the unit loop is unrolled by four so that the register slot is a compile-time index and every
// wait an immediate.  Loads and weight requests run past the end of the reduction with out-of-range / repeated addresses instead of
// stopping (nobody reads what they deliver), so the counts of the steady state hold to the last unit.  blockIdx.z = K split.
// C2D: 3 x 3 convolution over an H x W image (W a power of two, zero padding 1) as the same GEMM with K = 9 Ci: unit = (chunk, tap), the
// unit's input piece is loaded from the tap-shifted positions (im2col on the fly; the deep U-Net levels of RMVPE, whose activations live
// in L2: 512 channels x 404 positions, split over K because 16 tiles do not fill anything).
// C2D = 2: 1-D convolution with taps (stride 1, any dilation) the same way, K = ktaps * Ci - the short sequences (100 frames per second) of
// the text encoder's FFN and the flow's WaveNet, whose grids are too small for the tiled kernels.
template <int AM, int AN, int C2D = 0>
__global__ __launch_bounds__(256, 3) void conv_x3g_kernel(const ConvArgsX p) {
  constexpr int WM = 2, WN = 2, NW = 4;
  constexpr int BM = WM * AM * 32, BN = WN * AN * 32, RB = BM / 32;
  static_assert(BN == 128 && (RB == 2 || RB == 4), "128 columns: one staging piece per wave and unit");
  constexpr int NPW = 2 * RB / NW;                          // weight pieces per unit and wave
  constexpr int RW = 3, RX = 3, LX = 4;                     // weight / input slots in LDS, input slots in flight in registers
  constexpr int wslot = 2 * BM * 32, xslot = 2 * BN * 32;   // [hi | lo][half][rows][16 B]
  constexpr int xplane = BN * 32, xhalf = xplane / 2;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem3g[];
  unsigned char* Xs = smem3g;
  unsigned char* Ws = smem3g + RX * xslot;

  const int tid0 = threadIdx.x;
  int lane = tid0 & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;
  const unsigned tile = p.xcd_remap ? xcd_tile(blockIdx.x + blockIdx.y * gridDim.x, gridDim.x * gridDim.y) : blockIdx.x + blockIdx.y * gridDim.x;
  const int tile_y = (int)(tile / gridDim.x), tile_x = (int)(tile - (unsigned)tile_y * gridDim.x);
  const int co0 = tile_y * BM, n0 = tile_x * BN;
  const int ks = blockIdx.z;
  const int upk = p.nchunk / p.ksplit;                       // units per K split (a multiple of 4, >= 8: host)
  const int u0 = ks * upk, U = upk;
  const float pre_slope = p.pre_act == ACT_LRELU ? p.pre_slope : 1.f;
  const __amdgpu_buffer_rsrc_t xrs = make_rsrc(p.X, (unsigned)p.Ci * (unsigned)p.ldX * 4u);

  const bool r_init = p.ksplit == 1 && p.R != nullptr && p.act == ACT_NONE;
  f32x16 acc[AM][AN];
  if (r_init) {
    const __amdgpu_buffer_rsrc_t rrs = make_rsrc(p.R, (unsigned)p.orows * (unsigned)p.ldR * 4u);
#pragma unroll
    for (int am = 0; am < AM; ++am)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = co0 + (wm * AM + am) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float bv = (p.bias && m < p.Co) ? p.bias[m] : 0.f;
#pragma unroll
        for (int an = 0; an < AN; ++an) {
          const int n = n0 + (wn * AN + an) * 32 + li;
          acc[am][an][r] = buf_load(rrs, (m < p.Co && n < p.Tout) ? ((unsigned)m * (unsigned)p.ldR + (unsigned)n) * 4u : kOOB) + bv;
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

  // ---- weights: unit u = chunk u0 + u, 2 * RB pieces of 1 KiB; the request for a unit past the end repeats the last one
  const unsigned char* wsrc;
  {
    const int hl0 = wave / RB, r0 = (wave % RB) * 64 + lane;
    wsrc = p.Wx + ((long long)u0 * 2 + hl0) * p.CoPx * 32 + ((long long)(r0 / BM) * p.CoPx + co0 + (r0 % BM)) * 16;
  }
  const long long wstep = (long long)p.CoPx * 64;
  int slw = 0, uw = 0;
  auto issue_w = [&]() {
    unsigned char* dst = Ws + slw * wslot + wave * 1024;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)wsrc, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    if constexpr (RB == 4)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + (long long)p.CoPx * 32),
                                       (__attribute__((address_space(3))) void*)(dst + NW * 1024), 16, 0, 0);
    ++uw;
    wsrc += uw < U ? wstep : 0;
    slw = slw + 1 == RW ? 0 : slw + 1;
  };
  // ---- input: this wave's piece of a unit = 8 channels (half hb) x 64 columns
  const int hb = wave >> 1, xq = (wave & 1) * 64 + lane;
  const int xn = n0 + xq;                                      // this lane's column (1-D) / linear image position (2-D)
  const unsigned xvoff = (xn < (C2D ? p.Tout : p.Tin)) ? (unsigned)xn * 4u : kOOB;
  const int wlog = C2D == 1 ? 31 - __builtin_clz((unsigned)p.Wd) : 0;
  const int xh = C2D == 1 ? xn >> wlog : 0, xw = C2D == 1 ? xn & (p.Wd - 1) : 0;
  float xr[LX][8];
  auto load_unit = [&](int j, int u) {                        // register slot j <- unit u (out of range past the end: zeros, no traffic)
    unsigned c0, vo;
    if constexpr (C2D == 2) {
      const int gu = u0 + u, ch = gu / p.ktaps, tap = gu - p.ktaps * ch;
      const int x = xn - p.pad + tap * p.dil;
      const bool ok = u < U && xvoff != kOOB && x >= 0 && x < p.Tin;
      c0 = (unsigned)(ch * 16 + hb * 8);
      vo = ok ? (unsigned)x * 4u : kOOB;
    } else if constexpr (C2D == 1) {
      const int gu = u0 + u, ch = gu / 9, tap = gu - 9 * ch, dh = tap / 3 - 1, dw = tap - 3 * (tap / 3) - 1;
      const int hh = xh + dh, ww = xw + dw;
      const bool ok = u < U && xvoff != kOOB && hh >= 0 && hh < p.Tin && ww >= 0 && ww < p.Wd;
      c0 = (unsigned)(ch * 16 + hb * 8);
      vo = ok ? (unsigned)((hh << wlog) + ww) * 4u : kOOB;
    } else {
      c0 = (unsigned)((u0 + u) * 16 + hb * 8);
      vo = u < U ? xvoff : kOOB;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) xr[j][i] = buf_load(xrs, vo, (c0 + i) * (unsigned)p.ldX * 4u);
  };
  auto store_unit = [&](int j, int xl) {                      // register slot j -> input slot xl of the ring
    u32x4 hi, lo;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float a = xr[j][2 * i], b = xr[j][2 * i + 1];
      unsigned h_, l_;
      split2(fmaxf(a, a * pre_slope), fmaxf(b, b * pre_slope), h_, l_);
      hi[i] = h_; lo[i] = l_;
    }
    unsigned char* d = Xs + xl * xslot + hb * xhalf + xq * 16;
    *reinterpret_cast<u32x4*>(d) = hi;
    *reinterpret_cast<u32x4*>(d + xplane) = lo;
  };

  const int aoff = lh * (BM * 16) + ((wm * AM) * 32 + li) * 16;
  const int boff = lh * xhalf + ((wn * AN) * 32 + li) * 16;

  // ---- prologue.  Issue order: [residual] input units 0 .. 3 | weight units 0, 1 | (units 0, 1 converted) input units 4, 5
#pragma unroll
  for (int j = 0; j < LX; ++j) load_unit(j, j);
  issue_w(); issue_w();
  wait_vmcnt<16 + 2 * NPW>();                                  // units 0, 1 (younger: units 2, 3, the weights)
  store_unit(0, 0); store_unit(1, 1);
  load_unit(0, 4); load_unit(1, 5);
  wait_vmcnt<NPW + 16>();                                      // weight unit 0 (younger: weight unit 1, input units 4, 5)
#pragma unroll
  for (int am = 0; am < AM; ++am)
#pragma unroll
    for (int an = 0; an < AN; ++an) asm volatile("" : "+v"(acc[am][an]));
  lds_barrier();

  u32x4 ah[AM], ahn[AM], al[AM], bh[AN], bl[AN];
  {
    const unsigned char* wa = Ws + aoff;
    const unsigned char* xa = Xs + boff;
#pragma unroll
    for (int am = 0; am < AM; ++am) ah[am] = *reinterpret_cast<const u32x4*>(wa + am * 512);
#pragma unroll
    for (int an = 0; an < AN; ++an) bl[an] = *reinterpret_cast<const u32x4*>(xa + xplane + an * 512);
  }

  int sl = 0, xl = 0;                                          // weight / input slot of the current unit
  for (int ug = 0; ug < U; ug += 4) {
    const bool first = ug == 0;
    auto unit = [&](auto jc) {
      constexpr int J = decltype(jc)::value;
      const int u = ug + J;
      {
        const unsigned char* wa = Ws + sl * wslot + BM * 32 + aoff;
        const unsigned char* xa = Xs + xl * xslot + boff;
#pragma unroll
        for (int am = 0; am < AM; ++am) al[am] = *reinterpret_cast<const u32x4*>(wa + am * 512);
#pragma unroll
        for (int an = 0; an < AN; ++an) bh[an] = *reinterpret_cast<const u32x4*>(xa + an * 512);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int am = 0; am < AM; ++am)
#pragma unroll
        for (int an = 0; an < AN; ++an)
          acc[am][an] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[am]), __builtin_bit_cast(bf16x8, bl[an]), acc[am][an], 0, 0, 0);
      {
        // unit u + 2's input: loaded four units ago into register slot (J + 2) % 4; operations issued since - steady state: that unit's
        // weight request and three units of (8 loads + weights); first group: see the prologue's issue order
        constexpr int JS = (J + 2) % LX;
        constexpr int NS = 24 + 4 * NPW;
        constexpr int NF = J == 0 ? 24 + 2 * NPW : (J == 1 ? 24 + 3 * NPW : (J == 2 ? 24 + 2 * NPW : 24 + 3 * NPW));
        if (first) wait_vmcnt<NF>(); else wait_vmcnt<NS>();
        const int xs2 = xl + 2 >= RX ? xl + 2 - RX : xl + 2;
        store_unit(JS, xs2);
        load_unit(JS, u + 6);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int am = 0; am < AM; ++am)
#pragma unroll
        for (int an = 0; an < AN; ++an)
          acc[am][an] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al[am]), __builtin_bit_cast(bf16x8, bh[an]), acc[am][an], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      {
        // weight unit u + 1 (requested one unit ago; younger: this unit's 8 loads - first unit: the prologue's units 4, 5 as well)
        if (first && J == 0) wait_vmcnt<24>(); else wait_vmcnt<8>();
        lds_barrier();
        issue_w();                                             // unit u + 2 into the slot unit u - 1 was read from
        const int sn = sl + 1 == RW ? 0 : sl + 1, xn = xl + 1 == RX ? 0 : xl + 1;
        const unsigned char* wa = Ws + sn * wslot + aoff;
        const unsigned char* xa = Xs + xn * xslot + xplane + boff;
#pragma unroll
        for (int am = 0; am < AM; ++am) ahn[am] = *reinterpret_cast<const u32x4*>(wa + am * 512);
#pragma unroll
        for (int an = 0; an < AN; ++an) bl[an] = *reinterpret_cast<const u32x4*>(xa + an * 512);
        sl = sn; xl = xn;
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int am = 0; am < AM; ++am)
#pragma unroll
        for (int an = 0; an < AN; ++an)
          acc[am][an] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[am]), __builtin_bit_cast(bf16x8, bh[an]), acc[am][an], 0, 0, 0);
#pragma unroll
      for (int am = 0; am < AM; ++am) ah[am] = ahn[am];
    };
    static_for<0, 4>(unit);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // (the requests past the end: nothing may land in LDS after the workgroup has gone)

  if (p.ksplit > 1) {
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
  } else if (r_init) {
    ConvArgsX pe = p;
    pe.R = nullptr; pe.bias = nullptr;
    dense_epilogue<WM, WN, AM, AN, 4>(pe, acc, 0, co0, n0, wm, wn, li, lh);
  } else {
    dense_epilogue<WM, WN, AM, AN, 4>(p, acc, 0, co0, n0, wm, wn, li, lh);
  }
}

// ============================================================================ fused ResBlock pair of the 32-channel generator stage
// y = (x + c2(lrelu(c1_d(lrelu(x)) + b1)) + b2) * scale [+ y] in one launch, the intermediate in LDS (what conv_x3_kernel<FUSE> did with
// one barrier, one weight wait and one round of operand reads per 6 MFMAs: 86 k cycles per tile for 8.4 k cycles of MFMA issue).  Here:
// both 16-channel chunks of the input tile are converted once (they are the two input buffers of the pipeline), the weights of BOTH
// convolutions are one stream of 4 KT single-tap units through a ring of four 2-KiB slots (a unit requested three units ahead; requests
// past the end repeat the last unit so that one immediate wait count holds throughout), every unit's six operand reads are issued a whole
// unit ahead of its MFMAs, and the residual tile is read at the very start into registers of its own (32: the tile is narrow) and becomes
// the initial value of the second pass's accumulators.
// Tile: 32 rows x 256 intermediate columns = 256 - (KT - 1) output columns; 4 waves side by side (2 column blocks each).
// WM = 1: 32 channels, 4 waves side by side, 256 intermediate columns (three workgroups per CU); WM = 2: 64 channels, 2 x 2 waves, 128
// intermediate columns (four chunk buffers of 128 + halo columns: two workgroups per CU).
template <int KT, int WM>
__global__ __launch_bounds__(256, WM == 1 ? 3 : 2) void conv_x3pf_kernel(const ConvArgsX p) {
  constexpr int NW = 4, WN = NW / WM, AN = 2, BM = 32 * WM, BN = WN * AN * 32, R = 4, P2 = (KT - 1) / 2;
  constexpr int NCK = BM / 16;                               // 16-channel chunks = input buffers
  constexpr int wslot = 2 * BM * 32;                         // [hi | lo][half][BM rows][16 B]: 2 / 4 KiB
  constexpr int NIMAX = WM == 1 ? 5 : 3;                     // 64-column groups of the staged tile (256 / 128 + halo <= 64 columns)
  constexpr int XS = NCK * 2 * NIMAX / NW;                   // staging slots per wave (5 / 6)
  constexpr int NU = NCK * KT;                               // units per pass
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem3f[];
  const int P = p.WROW;                                      // staged input columns: 256 + (KT - 1) * dil  (<= 320)
  const int xplane = P * 32, xhalf = xplane >> 1, xbuf = 2 * xplane;
  unsigned char* Xs = smem3f;                                // one buffer per chunk: x, later the intermediate
  unsigned char* Ws = smem3f + ((NCK * xbuf + 1023) & ~1023);

  const int tid0 = threadIdx.x;
  int lane = tid0 & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  const int tile_x = (int)(p.xcd_remap ? xcd_tile(blockIdx.x, gridDim.x) : blockIdx.x);
  const int n0 = tile_x * (BN - 2 * P2);                     // first output column; intermediate column nl <-> position n0 - P2 + nl
  const int bx = n0 - P2 - p.pad;                            // first staged input column
  const int ni = p.ni;                                       // 64-column groups of the staged tile (<= 5)
  const int dil16 = p.dil * 16;
  const float pre_slope = p.pre_slope, hs = p.fuse_slope;
  const __amdgpu_buffer_rsrc_t xrs = make_rsrc(p.X, (unsigned)p.Ci * (unsigned)p.ldX * 4u);

  // ---- weights: units 0 .. NCK KT - 1 of conv1, then of conv2; every wave requests one 1-KiB piece per unit (32 channels: a unit is two
  // pieces, waves 2, 3 repeat those of waves 0, 1 - one count for all; 64 channels: four pieces, (hi | lo, half) = wave)
  const long long lane_w = WM == 1 ? (long long)(wave & 1) * p.CoPx * 32 + ((long long)lh * p.CoPx + li) * 16
                                   : (long long)(wave >> 1) * p.CoPx * 32 + ((long long)(wave & 1) * p.CoPx + lane) * 16;
  const long long wstep = (long long)p.CoPx * 64;
  const unsigned char* wsrc = p.Wx + lane_w;
  int slw = 0, uw = 0;
  auto issue_w = [&]() {
    unsigned char* dst = Ws + slw * wslot + (WM == 1 ? (wave & 1) : wave) * 1024;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)wsrc, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    ++uw;
    if (uw == NU) wsrc = p.Wx2 + lane_w;                     // conv2's image
    else if (uw < 2 * NU) wsrc += wstep;                     // (past the end: the last unit again)
    slw = slw + 1 == R ? 0 : slw + 1;
  };

  // ---- prologue: the residual tile (oldest loads; needed only at the second pass), the whole input tile (both chunks), three weight units
  float rr[AN][16];
  {
    const __amdgpu_buffer_rsrc_t rrs = make_rsrc(p.R, (unsigned)p.Co * (unsigned)p.ldR * 4u);
    const int nend = min(p.Tout, n0 + BN - 2 * P2);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
#pragma unroll
      for (int an = 0; an < AN; ++an) {
        const int n = n0 + (wn * AN + an) * 32 + li;
        rr[an][r] = buf_load(rrs, n < nend ? ((unsigned)m * (unsigned)p.ldR + (unsigned)n) * 4u : kOOB);
      }
    }
  }
  {
    float xr[XS][8];
#pragma unroll
    for (int s = 0; s < XS; ++s) {
      const int t = wave + NW * s;                            // chunk, half, column group
      const int cc = t / (2 * ni), g = t - cc * 2 * ni, hb = g >= ni ? 1 : 0, q = (g - hb * ni) * 64 + lane;
      const int x = bx + q;
      const unsigned voff = (t < NCK * 2 * ni && q < P && x >= 0 && x < p.Tin) ? (unsigned)x * 4u : kOOB;
      const unsigned c0 = (unsigned)(cc * 16 + hb * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) xr[s][j] = buf_load(xrs, voff, (c0 + j) * (unsigned)p.ldX * 4u);
    }
    issue_w(); issue_w(); issue_w();
    wait_vmcnt<3>();                                          // the input (younger: three weight pieces)
#pragma unroll
    for (int s = 0; s < XS; ++s) {
      const int t = wave + NW * s;
      const int cc = t / (2 * ni), g = t - cc * 2 * ni, hb = g >= ni ? 1 : 0, q = (g - hb * ni) * 64 + lane;
      if (t < NCK * 2 * ni && q < P) {
        u32x4 hi, lo;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float a = xr[s][2 * j], b = xr[s][2 * j + 1];
          unsigned h_, l_;
          split2(fmaxf(a, a * pre_slope), fmaxf(b, b * pre_slope), h_, l_);
          hi[j] = h_; lo[j] = l_;
        }
        unsigned char* d = Xs + cc * xbuf + hb * xhalf + q * 16;
        *reinterpret_cast<u32x4*>(d) = hi;
        *reinterpret_cast<u32x4*>(d + xplane) = lo;
      }
    }
  }
  wait_vmcnt<2>();                                            // weight unit 0
  lds_barrier();

  f32x16 acc[AN];
#pragma unroll
  for (int an = 0; an < AN; ++an)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[an][r] = 0.f;

  const int aoff = lh * (BM * 16) + (wm * 32 + li) * 16;
  const int boff = lh * xhalf + (wn * AN * 32 + li) * 16;
  u32x4 ah, al, bh[AN], bl[AN], ahn, aln, bhn[AN], bln[AN];
  auto read_ops = [&](u32x4& a_h, u32x4& a_l, u32x4 (&b_h)[AN], u32x4 (&b_l)[AN], int slot, int xoff) {
    const unsigned char* wa = Ws + slot * wslot + aoff;
    const unsigned char* xa = Xs + xoff + boff;
    a_h = *reinterpret_cast<const u32x4*>(wa); a_l = *reinterpret_cast<const u32x4*>(wa + BM * 32);
#pragma unroll
    for (int an = 0; an < AN; ++an) { b_h[an] = *reinterpret_cast<const u32x4*>(xa + an * 512); b_l[an] = *reinterpret_cast<const u32x4*>(xa + xplane + an * 512); }
  };
  read_ops(ah, al, bh, bl, 0, 0);

  int sl = 0;
  // one pass = 2 chunks x KT taps over the two chunk buffers; D16 = tap distance in bytes
  auto pass = [&](int D16, bool last_pass) {
    auto unit = [&](auto uc) {
      constexpr int Uu = decltype(uc)::value;                 // unit inside the pass
      constexpr int T = Uu % KT, Cc = Uu / KT;
      constexpr bool last = Uu + 1 == NU;
      // next unit's weights are published; its operands are requested now, a whole unit ahead (at the end of the first pass only the
      // weights: the intermediate is not written yet)
      wait_vmcnt<1>();                                        // unit u + 1 (younger: unit u + 2)
      lds_barrier();
      issue_w();                                              // unit u + 3 into the slot unit u - 1 was read from
      const int sn = sl + 1 == R ? 0 : sl + 1;
      if constexpr (!last) {
        constexpr int Tn = (Uu + 1) % KT, Cn = (Uu + 1) / KT;
        read_ops(ahn, aln, bhn, bln, sn, Cn * xbuf + Tn * D16);
      } else {
        const unsigned char* wa = Ws + sn * wslot + aoff;
        ahn = *reinterpret_cast<const u32x4*>(wa); aln = *reinterpret_cast<const u32x4*>(wa + BM * 32);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int an = 0; an < AN; ++an)
        acc[an] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bl[an]), acc[an], 0, 0, 0);
#pragma unroll
      for (int an = 0; an < AN; ++an)
        acc[an] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al), __builtin_bit_cast(bf16x8, bh[an]), acc[an], 0, 0, 0);
#pragma unroll
      for (int an = 0; an < AN; ++an)
        acc[an] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bh[an]), acc[an], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      ah = ahn; al = aln;
      if constexpr (!last) {
#pragma unroll
        for (int an = 0; an < AN; ++an) { bh[an] = bhn[an]; bl[an] = bln[an]; }
      }
      sl = sn;
      (void)T; (void)Cc;
    };
    static_for<0, NU>(unit);
    (void)last_pass;
  };

  // ---- pass 1: the dilated conv over the 256 columns
  pass(dil16, false);
  // ---- the intermediate h = lrelu(acc + b1) (0 outside the sequence: the second conv's zero padding), split, over the input tile
  lds_barrier();                                              // every wave is done with the input tile
  {
#pragma unroll
    for (int an = 0; an < AN; ++an) {
      const int nl = (wn * AN + an) * 32 + li;
      const int gh = n0 - P2 + nl;
      const bool inside = gh >= 0 && gh < p.Tin;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int mb = 8 * g;                                 // rows mb + 4 lh + {0..3}: one 8-byte quarter of a 16-B row
        u32x4 hl;
#pragma unroll
        for (int e2 = 0; e2 < 2; ++e2) {
          float a = acc[an][4 * g + 2 * e2] + p.bias1[wm * 32 + mb + 4 * lh + 2 * e2];
          float b = acc[an][4 * g + 2 * e2 + 1] + p.bias1[wm * 32 + mb + 4 * lh + 2 * e2 + 1];
          a = inside ? fmaxf(a, a * hs) : 0.f;
          b = inside ? fmaxf(b, b * hs) : 0.f;
          unsigned h_, l_;
          split2(a, b, h_, l_);
          hl[e2] = h_; hl[2 + e2] = l_;
        }
        const int cc = wm * 2 + (mb >> 4), hb = (mb >> 3) & 1;
        unsigned char* row = Xs + cc * xbuf + hb * xhalf + nl * 16 + lh * 8;
        *reinterpret_cast<unsigned long long*>(row) = (unsigned long long)hl[0] | ((unsigned long long)hl[1] << 32);
        *reinterpret_cast<unsigned long long*>(row + xplane) = (unsigned long long)hl[2] | ((unsigned long long)hl[3] << 32);
      }
    }
  }
  lds_barrier();                                              // h published
  // ---- pass 2 accumulators: residual + bias2 (nothing but the scale follows the sum)
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float bv = p.bias ? p.bias[wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh] : 0.f;
#pragma unroll
    for (int an = 0; an < AN; ++an) acc[an][r] = rr[an][r] + bv;
  }
  {
    const unsigned char* xa = Xs + boff;
#pragma unroll
    for (int an = 0; an < AN; ++an) { bh[an] = *reinterpret_cast<const u32x4*>(xa + an * 512); bl[an] = *reinterpret_cast<const u32x4*>(xa + xplane + an * 512); }
  }
  pass(16, true);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the requests past the end
  {
    ConvArgsX pe = p;
    pe.R = nullptr; pe.bias = nullptr;                         // inside the accumulators
    pe.Tout = min(p.Tout, n0 + BN - 2 * P2);                   // columns without their full halo belong to the neighbouring tiles
    f32x16 a2[1][AN];
#pragma unroll
    for (int an = 0; an < AN; ++an) a2[0][an] = acc[an];
    dense_epilogue<WM, WN, 1, AN, 4>(pe, a2, 0, 0, n0, wm, wn, li, lh);
  }
}

// ============================================================================ host side
template <int AM, int AN, int KT, bool XSPLIT, bool YSPLIT, bool S2 = false>
static void launch_x3p(const ConvArgsX& a, dim3 grid, size_t lds, hipStream_t s) {
  auto kern = conv_x3p_kernel<AM, AN, KT, XSPLIT, YSPLIT, S2>;
  RVC_ALLOW_BIG_LDS(kern);
  conv_launch(kern, grid, dim3(256), lds, s, a);
}
template <int AM, int AN, int KT>
static void launch_x3p_io(const ConvArgsX& a, dim3 grid, size_t lds, hipStream_t s) {
  if (a.Xs) launch_x3p<AM, AN, KT, true, false>(a, grid, lds, s);
  else if (a.Ys) launch_x3p<AM, AN, KT, false, true>(a, grid, lds, s);
  else launch_x3p<AM, AN, KT, false, false>(a, grid, lds, s);
}
template <int AM, int AN>
static void launch_x3p_k(const ConvArgsX& a, dim3 grid, size_t lds, hipStream_t s) {
  if (a.ktaps == 3) launch_x3p_io<AM, AN, 3>(a, grid, lds, s);
  else if (a.ktaps == 7) launch_x3p_io<AM, AN, 7>(a, grid, lds, s);
  else launch_x3p_io<AM, AN, 11>(a, grid, lds, s);
}

// a: arguments as conv_x3_try prepared them (true taps, tile chosen: WM = WN = 2).  Returns false when the geometry is not the
// pipelined kernel's (the staged kernel takes it): kernel sizes 3 / 7 / 11 (the generator's), at least three 16-channel chunks.
bool conv_x3p_try(ConvArgsX& a, int AM, int AN, hipStream_t s, dim3& grid_out, bool dry) {
  static const int on = exp_int("RVC_X3P", 1);
  if (!on) return false;
  const bool xs = a.Xs != nullptr, ys = a.Ys != nullptr;
  if (a.Wd > 0 || (a.Ci & 15) || a.Ci < 48 || (xs && ys)) return false;
  // stride 2: the k = 3 layers of HuBERT's feature encoder on 128 x 128 tiles
  const bool s2 = a.stride == 2;
  if (a.stride != 1 && !(s2 && a.ktaps == 3 && a.dil == 1 && AM == 2 && AN == 2 && !xs && !ys && a.ostride == 1)) return false;
  // transposed conv: plain interleaved store
  if (a.ostride != 1 && (xs || ys || a.R || a.accumulate || (double)a.orows * (double)a.ldY * 4.0 >= 2147483648.0)) return false;
  if (!(a.ktaps == 3 || a.ktaps == 7 || a.ktaps == 11)) return false;
  if (!((AM == 2 && AN == 4) || (AM == 1 && AN == 4) || (AM == 2 && AN == 2))) return false;
  const int BM = 64 * AM, BN = 64 * AN;
  const int P = s2 ? 2 * (BN - 1) + a.ktaps : BN + (a.ktaps - 1) * a.dil;
  if (P > 384 || (!s2 && P > BN + 64)) return false;              // three staging slots per wave; split input: BN + 64 rows per half-plane
  const int Pm = xs ? BN + 64 : (s2 ? 2 * ((P + 1) >> 1) : P);
  const int xbytes = (2 * 2 * Pm * 32 + 1023) & ~1023;
  const int wslot = 2 * BM * 32;
  const int R = (AM == 2 && AN == 4) ? RVC_X3P_R24 : 3;
  const size_t lds = (size_t)xbytes + (size_t)R * wslot;
  if (lds > (size_t)(AM * AN >= 8 || s2 ? 80 : 53) * 1024) return false;   // two / three workgroups per CU
  if (dry) return true;
  a.WROW = P; a.ni = (P + 63) / 64; a.nchunk = a.Ci / 16; a.NC = 1; a.KT = 1; a.xbufs = 2; a.ksplit = 1; a.partial = nullptr; a.wbufs = R;
  static const int xcd_env = exp_int("RVC_X3_XCD", 1);
  a.xcd_remap = xcd_env;
  dim3 grid((unsigned)((a.Tout + BN - 1) / BN), (unsigned)((a.Co + BM - 1) / BM), 1);
  grid_out = grid;
  if (s2) launch_x3p<2, 2, 3, false, false, true>(a, grid, lds, s);
  else if (AM == 2 && AN == 4) launch_x3p_k<2, 4>(a, grid, lds, s);
  else if (AM == 1 && AN == 4) launch_x3p_k<1, 4>(a, grid, lds, s);
  else launch_x3p_k<2, 2>(a, grid, lds, s);
  return true;
}

template <int AM, int AN, int C2D = 0>
static void launch_x3g(const ConvArgsX& a, dim3 grid, size_t lds, hipStream_t s) {
  auto kern = conv_x3g_kernel<AM, AN, C2D>;
  RVC_ALLOW_BIG_LDS(kern);
  conv_launch(kern, grid, dim3(256), lds, s, a);
}

// k = 1 convolutions (GEMMs) on the pipelined kernel: fp32 [K][N] input, K a multiple of 64 and >= 128.  Small grids are split over K
// (deterministic second pass: splitk_reduce_launch).  Returns false when the geometry is not this kernel's.
bool conv_x3g_try(ConvArgsX& a, hipStream_t s, dim3& grid_out, int& ksplit_out, bool dry) {
  static const int on = exp_int("RVC_X3G", 1);
  if (!on) return false;
  // 2-D: 3 x 3, pad 1, plain (no 2 x 2 up-sampling interleave), on images small enough that the nine shifted reads come from L2
  static const int on2d = exp_int("RVC_X3G_2D", 1);
  static const int max2d = exp_int("RVC_X3G_2D_MAXPOS", 30000);
  const bool two_d = a.Wd > 0;
  if (two_d && !(on2d && a.ktaps == 9 && a.KW == 0 && !a.up2 && (a.Wd & (a.Wd - 1)) == 0 && a.Tout <= max2d)) return false;
  // (measured against the staged kernel + split-K, launch incl. the reduction: 128 / 256 channels 36 -> 31 us, the level changes 50 -> 38;
  // 512 x 512 on 404 positions 34.5 -> 37: that one stays)
  if (two_d && a.Ci >= 512 && a.Co <= 512 && a.Tout < 1000 && on2d < 2) return false;
  // 1-D with taps: only where the tiled kernels would not run (grids below their minimum: the 100-frames-per-second layers)
  static const int on1d = exp_int("RVC_X3G_TAPS", 1);
  const bool taps1d = !two_d && a.ktaps > 1;
  if (taps1d && !(on1d && a.ktaps <= 16 && (long long)((a.Co + 127) / 128) * ((a.Tout + 127) / 128) < 250)) return false;
  if (a.stride != 1 || a.ostride != 1 || a.Xs || a.Ys || (a.Ci & 15)) return false;
  if (!two_d && !taps1d && a.Tin != a.Tout) return false;
  static const int am_env = exp_int("RVC_X3G_AM", 0);
  // 64-row tiles for short reductions (K <= 1024: q/k/v 49 -> 40 us, flow 192 -> 192 16 -> 11), 128-row tiles for long ones (FFN2, K = 3072: 61 vs 75 us)
  const int AM = am_env ? am_env : ((a.Co > 64 && a.Ci * a.ktaps > 1024) ? 2 : 1), BM = 64 * AM, BN = 128;
  const int U = a.Ci / 16 * a.ktaps;
  if ((U & 3) || U < 8) return false;                              // unit loop unrolled by four
  const long long nblk = (long long)((a.Co + BM - 1) / BM) * ((a.Tout + BN - 1) / BN);
  static const int min_blk = exp_int("RVC_X3G_MINBLK", 24);
  if (nblk < (two_d ? 8 : min_blk)) return false;
  // K split: enough workgroups for the chip (a 128 x 128 tile of a K = 768 GEMM is 9 us of MFMAs), at least 8 units per split, groups of 4
  static const int target = exp_int("RVC_X3G_BLK", 256);
  int S = 1;
  for (int c : {2, 3, 4, 6, 8}) {
    if (nblk * S >= target) break;
    if (U % (4 * c) == 0 && U / c >= 8) S = c;
  }
  if (dry) return true;
  a.nchunk = U; a.NC = 1; a.KT = 1; a.ksplit = S; a.xcd_remap = 0; a.wbufs = 3; a.xbufs = 3;
  a.ldP = (a.Tout + 31) & ~31; a.partial = nullptr;
  if (S > 1) a.partial = (float*)stream_scratch(s, 0, (size_t)S * a.Co * a.ldP * sizeof(float));
  const size_t lds = (size_t)3 * (2 * BN * 32) + (size_t)3 * (2 * BM * 32);
  dim3 grid((unsigned)((a.Tout + BN - 1) / BN), (unsigned)((a.Co + BM - 1) / BM), (unsigned)S);
  grid_out = grid; ksplit_out = S;
  if (two_d) { if (AM == 2) launch_x3g<2, 2, 1>(a, grid, lds, s); else launch_x3g<1, 2, 1>(a, grid, lds, s); }
  else if (taps1d) { if (AM == 2) launch_x3g<2, 2, 2>(a, grid, lds, s); else launch_x3g<1, 2, 2>(a, grid, lds, s); }
  else if (AM == 2) launch_x3g<2, 2>(a, grid, lds, s); else launch_x3g<1, 2>(a, grid, lds, s);
  if (S > 1) splitk_reduce_launch(a, S, 1, s);
  return true;
}

template <int KT, int WM>
static void launch_x3pf(const ConvArgsX& a, dim3 grid, size_t lds, hipStream_t s) {
  auto kern = conv_x3pf_kernel<KT, WM>;
  RVC_ALLOW_BIG_LDS(kern);
  conv_launch(kern, grid, dim3(256), lds, s, a);
}
// a: the fused pair's arguments as conv_x3_pair_try prepared them (C = 32: 256 intermediate columns per tile; C = 64: 128).  false: not
// this kernel's geometry.
bool conv_x3pf_try(ConvArgsX& a, int T, hipStream_t s, dim3& grid_out, bool dry) {
  static const int on = exp_int("RVC_X3PF", 1);
  static const int on64 = exp_int("RVC_X3PF64", 1);
  if (!on || !(a.Ci == 32 || (a.Ci == 64 && on64)) || a.Co != a.Ci || !(a.ktaps == 3 || a.ktaps == 7 || a.ktaps == 11)) return false;
  const int C = a.Ci, BN = C == 32 ? 256 : 128;
  // 64 channels: the narrow wave tile (32 rows: one operand read per MFMA) only wins where the pair is HBM-bound - k = 3: 220 -> 151 us;
  // k = 7: 284 -> 296, k = 11: 393 -> 507 against the two split-resident launches (RVC_X3PF64=2 forces it)
  // (128 channels, 4 waves on top of each other on 64-column tiles, was tried for k = 3: 307 vs 294 us - no gain, not kept)
  // (RVC_X3PF64=3: k = 3 and k = 7 fused, k = 11 split - round 4: with clips in flight the bytes a fused pair keeps off the HBM count)
  if (C == 64 && a.ktaps != 3 && !(on64 == 2 || (on64 == 3 && a.ktaps == 7))) return false;
  const int P = BN + (a.ktaps - 1) * a.dil;
  if (P > BN + 64) return false;
  const int NO = BN - (a.ktaps - 1);
  if ((long long)(T + NO - 1) / NO < 512) return false;          // short sequences: the unfused path fills the chip better
  if (dry) return true;
  a.WROW = P; a.ni = (P + 63) / 64;
  const size_t lds = (size_t)(((C / 16) * 2 * P * 32 + 1023) & ~1023) + 4 * (size_t)(2 * C * 32);
  dim3 grid((unsigned)((T + NO - 1) / NO), 1, 1);
  grid_out = grid;
  if (C == 32) { if (a.ktaps == 3) launch_x3pf<3, 1>(a, grid, lds, s); else if (a.ktaps == 7) launch_x3pf<7, 1>(a, grid, lds, s); else launch_x3pf<11, 1>(a, grid, lds, s); }
  else { if (a.ktaps == 3) launch_x3pf<3, 2>(a, grid, lds, s); else if (a.ktaps == 7) launch_x3pf<7, 2>(a, grid, lds, s); else launch_x3pf<11, 2>(a, grid, lds, s); }
  return true;
}

}  // namespace rvc
