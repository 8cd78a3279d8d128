// Persistent split-MFMA Conv1d for the generator's stride-1 ResBlock convolutions (gfx950 only), in two arithmetics: bf16x3 (three bf16 MFMAs
// per product: hi_w lo_x + hi_w hi_x + lo_w hi_x) and fp16x2 (H2: two fp16 MFMAs, w lo_x + w hi_x with the weight as ONE fp16 term and the
// activation as fp16 hi + lo - see the note above conv_x3q_try for what that costs in accuracy and why it is only offered to ResBlock pairs).
//
// conv_x3p_kernel (conv_x3p.hip) made the WAVE a pipeline; its tiles still were separate workgroups.  Per-phase cycle counters of that kernel
// (profiles/r4a_x3p_phase_cycles.txt: C128 k11, 128 x 256 tiles) show what that costs: of 227 k cycles per tile 33 k are the prologue (all 512
// resident workgroups load their first input chunk and the whole 128-KiB residual tile at once - an HBM burst while every matrix pipe idles),
// 9 - 22 k the epilogue; for k = 3 the two are 48 % of the tile.  Here a workgroup is resident for the whole launch and walks over its tiles:
//   * ONE stream of (tile, chunk, tap) units: the weight ring keeps cycling through the layer's image, the input chunks c + 2 of the stream
//     are requested while chunk c is multiplied - at a tile's last two chunks those are the NEXT tile's first two, so a tile boundary has no
//     prologue at all: epilogue stores, and the next unit's MFMAs follow;
//   * the residual (c2 of a ResBlock pair: y = c2(t) + x) no longer initialises the accumulators in one 128-load burst: block b (one 32 x 32
//     accumulator = 16 rows per lane) is loaded during chunk b - 1 and added to the accumulators in the middle of chunk b by VALU adds
//     (together with the bias, which lives in an LDS table); the loads of a tile's block 0 are issued in the previous tile's last chunk;
//   * every vmcnt wait is still an immediate: the issue sequence of a unit depends on its tap index only, so the number of operations
//     younger than the one waited for is a compile-time function of the tap (q_*() below); the only irregularity, the epilogue's stores
//     between two tiles, is added to the windows that span it (first chunk of a tile; clamped to the counter's 6 bits, which only waits longer).
//     A -DRVC_X3P_CHECK build keeps exact run-time bookkeeping beside the immediates (rvc_debug_x3p_check counts violations: must be 0).
// Arithmetic, LDS / image layouts and the split-resident in / out formats are those of conv_x3p.hip; the order of the fp32 additions differs
// (residual + bias join the sum after chunk b instead of before chunk 0), results agree to rounding.
#include "conv_x3_dev.h"

namespace rvc {

template <int N> __device__ __forceinline__ void q_wait() {
  static_assert(N >= 0, "vmcnt");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N > 63 ? 63 : N) : "memory");      // 6-bit counter: a smaller count than necessary only waits longer
}
template <int T, int N, class F> __device__ __forceinline__ void q_for(F& f) {
  if constexpr (T < N) { f(std::integral_constant<int, T>{}); q_for<T + 1, N>(f); }
}
constexpr int q_mod(int t, int KT) { return ((t % KT) + KT) % KT; }
// fp32 input: staging slot s (of XS) of the next chunk is converted during tap (s * KT) / XS
constexpr int q_cv(int t, int KT, int XS) {
  t = q_mod(t, KT);
  for (int s = 0; s < XS; ++s) if ((s * KT) / XS == t) return 1;
  return 0;
}

#ifdef RVC_X3P_CHECK
__device__ int g_x3q_bad;
int conv_x3q_check_read() { int v = 0, z = 0; (void)hipDeviceSynchronize(); (void)hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_x3q_bad), sizeof(int)); (void)hipMemcpyToSymbol(HIP_SYMBOL(g_x3q_bad), &z, sizeof(int)); return v; }
#define X3Q_CHECK(N, exact) do { if (((N) > 63 ? 63 : (N)) > (exact) && (threadIdx.x & 63) == 0) atomicAdd(&g_x3q_bad, 1); } while (0)
#define X3Q_ISSUED(n) (issued += (n))
#else
int conv_x3q_check_read() { return -1; }
#define X3Q_CHECK(N, exact) do {} while (0)
#define X3Q_ISSUED(n) do {} while (0)
#endif

#ifdef RVC_CONV_TIMING
__device__ unsigned long long g_x3q_timing[8];   // [0] tiles, [1] prologue (once per workgroup), [2] compute between barriers, [3] weight wait, [4] barrier, [5] epilogue, [6] total
void conv_x3q_timing_read(unsigned long long* out8, bool reset) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_x3q_timing), sizeof(unsigned long long) * 8);
  if (reset) { unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_x3q_timing), z, sizeof(z)); }
}
#define XQTICK() ((long long)__builtin_readcyclecounter())
#define XQACC(i, v) do { xqt[i] += (v); } while (0)
#else
void conv_x3q_timing_read(unsigned long long* out8, bool) { for (int i = 0; i < 8; ++i) out8[i] = 0; }
#define XQTICK() 0ll
#define XQACC(i, v) do {} while (0)
#endif

struct QTile { int co0, n0, bx; bool edge, valid; };
// One 32 x 32 x 16 product.  SWAP = false: D[channel][position], a lane holds ONE position (its column) and 16 channels - what the split-image
// epilogue needs (8 channels of a position are one 16-byte row).  SWAP = true: the operands trade places, D[position][channel]: a lane holds ONE
// channel and 4 x 4 consecutive positions, so residual loads and fp32 stores become 16-byte accesses without any transposition.  Measured
// (round 4, profiles/r4b_x3q_steps.txt) and NOT used: a store instruction then writes 32 bytes per channel row instead of whole 128-byte lines,
// the epilogue of a C128 k11 tile went from 10.4 k to 15.5 k cycles and every class got slower; the residual variant stages through LDS instead.
template <bool SWAP, bool H2 = false>
__device__ __forceinline__ f32x16 q_mfma(const u32x4& w, const u32x4& x, const f32x16& c) {
  if constexpr (H2) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, x), c, 0, 0, 0);
  else if constexpr (SWAP) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, w), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), c, 0, 0, 0);
}
typedef float f32x4q __attribute__((ext_vector_type(4)));

// AM x AN accumulators per wave (2 x 2 waves), KT taps.  XSPLIT: split-resident input image (DMA) instead of fp32 rows (registers, converted);
// YSPLIT: split-resident output image; RADD: residual (+ bias) added block by block during the tile (fp32 output, no activation after the sum).
// H2: fp16x2 arithmetic - the weight image holds ONE fp16 plane per unit ([chunk][tap][half][CoPx rows][8 ch]: half the L2 -> LDS stream, one DMA
// instruction per unit and wave), the activation image (staged from fp32 or split-resident) fp16 hi / lo planes, two MFMA groups per unit.
template <int AM, int AN, int KT, int R, bool XSPLIT, bool YSPLIT, bool RADD, bool H2>
__global__ __launch_bounds__(256, (AM * AN >= 8) ? 2 : 3) void conv_x3q_kernel(const ConvArgsX p) {
  static_assert(!(YSPLIT && RADD), "an image output has no residual");
  static_assert(XSPLIT != YSPLIT, "c1 of a pair (fp32 in, image out) or c2 (image in, fp32 out)");
  constexpr int WM = 2, WN = 2, NW = 4;
  constexpr int BM = WM * AM * 32, BN = WN * AN * 32, RB = BM / 32;
  // R = weight slots in the ring (a unit is requested R - 2 units before the barrier that publishes it; as many as the LDS budget admits)
  constexpr int XS = 3;                                     // fp32 staging slots per wave (8 channels x 64 positions each): P <= 384
  constexpr int NPW = H2 ? 1 : 2 * RB / NW;                 // weight DMA instructions per unit and wave (H2, 64-row tiles: half-wave pieces)
  constexpr int NPX = (BN + 64) / 64;                       // split-resident input: pieces per chunk and wave
  constexpr int wslot = (H2 ? 1 : 2) * BM * 32;
  constexpr int NBLK = AM * AN;                             // accumulator blocks = residual batches of 16 loads
  constexpr int TR = KT / 2;                                // tap at which the residual step runs
  static_assert(KT >= XS && (RB == 2 || RB == 4), "geometry");
  // ---- operations a unit issues, in program order: A (fp32 refill of a converted slot) | C (weight unit) | D (split input chunk, last tap) |
  // E (residual block)
  struct Q {
    static constexpr int nA(int t) { return XSPLIT ? 0 : 8 * q_cv(t, KT, XS); }
    static constexpr int nD(int t) { return (XSPLIT && q_mod(t, KT) == KT - 1) ? NPX : 0; }
    static constexpr int nE(int t) { return (RADD && q_mod(t, KT) == TR) ? 16 : 0; }
    static constexpr int tot(int t) { return nA(t) + NPW + nD(t) + nE(t); }
    static constexpr int sum() { int n = 0; for (int t = 0; t < KT; ++t) n += tot(t); return n; }
    // younger than the weight pieces of unit u + 1 (requested in step C of unit u - (R - 2)) at the wait of unit u (tap t), which follows step A
    static constexpr int w(int t) { int n = nD(t - (R - 2)) + nE(t - (R - 2)); for (int j = 1; j <= R - 3; ++j) n += tot(t - (R - 2) + j); return n + nA(t); }
  };
  // the epilogue's memory operations (a lower bound is what the waits need): stores only
  constexpr int EP = (YSPLIT || RADD) ? 4 * AM * AN : 16 * AM * AN;

  extern __shared__ __attribute__((aligned(1024))) unsigned char smemq[];
  const int P = p.WROW;                                     // staged input positions: BN + (KT - 1) * dil
  const int Pm = (P + 7) & ~7;                              // rows of a half-plane (split input: the last 64-row DMA piece is cut off there)
  const int xplane = Pm * 32, xhalf = xplane >> 1, xbuf = 2 * xplane;
  unsigned char* Xs = smemq;
  unsigned char* Ws = smemq + ((2 * xbuf + 1023) & ~1023);
  float* Bs = reinterpret_cast<float*>(Ws + R * wslot);      // bias of every output row
  unsigned char* Ss = reinterpret_cast<unsigned char*>(Bs) + ((p.Co * 4 + 255) & ~255);   // RADD: epilogue staging, 1 KiB per wave

  const int tid0 = threadIdx.x;
  int lane = tid0 & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;
  const int nck = p.nchunk;                                  // >= 3 (host)
  const int ni = p.ni;
  const int dil16 = p.dil * 16;
  const float pre_slope = p.pre_act == ACT_LRELU ? p.pre_slope : 1.f;
  const __amdgpu_buffer_rsrc_t xrs = make_rsrc(p.X, XSPLIT ? 0u : (unsigned)p.Ci * (unsigned)p.ldX * 4u);
  const __amdgpu_buffer_rsrc_t rrs = make_rsrc(RADD ? (const void*)p.R : (const void*)p.Y, RADD ? (unsigned)p.Co * (unsigned)p.ldR * 4u : 0u);

  // ---- tiles of this workgroup: linear ids blockIdx.x + it * gridDim.x (gridDim.x a multiple of 8: the id keeps its XCD), renumbered so that an
  // XCD works on one contiguous run of tiles, row tiles fastest inside the run (conv_x3p.hip)
  const unsigned nty = (unsigned)(p.Co / BM), ntiles = (unsigned)((p.Tout + BN - 1) / BN) * nty;
  auto geom = [&](int it) -> QTile {
    const unsigned L = blockIdx.x + (unsigned)it * gridDim.x;
    QTile t;
    t.valid = L < ntiles;
    const unsigned tile = !t.valid ? 0u : (p.xcd_remap ? xcd_tile(L, ntiles) : L);
    const int tx = (int)(tile / nty), ty = (int)(tile - (unsigned)tx * nty);
    t.co0 = ty * BM; t.n0 = tx * BN; t.bx = t.n0 - p.pad;
    t.edge = XSPLIT && (t.bx < 0 || t.bx + P > p.Tin);
    return t;
  };
  const int nit = (int)((ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x);

#ifdef RVC_CONV_TIMING
  long long xqt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  const long long t_begin = XQTICK();
  [[maybe_unused]] long long t_last = t_begin;
#ifdef RVC_X3P_CHECK
  int issued = 0, mk_w[8] = {0, 0, 0, 0, 0, 0, 0, 0}, mk_x[XS] = {0, 0, 0}, mk_r = 0, uwc = 0, mk_xs[2] = {0, 0};
#endif

  // ---- bias table (LDS): row m of the layer
  for (int m = tid0; m < p.Co; m += NW * 64) Bs[m] = p.bias ? p.bias[m] : 0.f;

  // ---- weights: unit (chunk, tap) of a tile with rows co0 is 2 * RB pieces of 1 KiB; consecutive units are consecutive planes of the image.
  // Buffer DMA: one constant per-lane offset, the unit / row-tile position is a scalar offset (no 64-bit vector address arithmetic)
  const int NU = nck * KT;
  const unsigned wstep = (unsigned)p.CoPx * (H2 ? 32u : 64u);
  const __amdgpu_buffer_rsrc_t wrs = make_rsrc(p.Wx, (unsigned)NU * wstep);
  // H2: a unit is [half][BM rows] of 16 B = RB pieces of 1 KiB: one per wave at BM = 128, half a piece (lanes 0 - 31: 32 rows) per wave at BM = 64
  const int wrow_h2 = RB == 4 ? wave * 64 + lane : wave * 32 + (lane & 31);
  const int wvoff = H2 ? ((wrow_h2 / BM) * p.CoPx + (wrow_h2 % BM)) * 16
                       : ((wave / RB) * p.CoPx * 32) + (((((wave % RB) * 64 + lane) / BM) * p.CoPx + (((wave % RB) * 64 + lane) % BM)) * 16);
  QTile cur = geom(0), nxt = geom(1);
  unsigned wsoff = (unsigned)cur.co0 * 16u;                    // scalar offset of the next request
  int slw = 0, uw = 0;                                       // ring slot / unit-in-tile of the next request
  int wtile_co0_next = nxt.valid ? nxt.co0 : cur.co0;        // rows of the tile the request stream enters at its next wrap
  auto issue_w = [&]() {
    if constexpr (H2) {
      // (64-row tiles: the upper half-wave is switched off - the instruction still issues, so the counts below hold)
      unsigned char* dst = Ws + slw * wslot + wave * (RB == 4 ? 1024 : 512);
      if (RB == 4 || lane < 32)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)dst, 16, wvoff, (int)wsoff, 0, 0);
    } else {
      unsigned char* dst = Ws + slw * wslot + wave * 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)dst, 16, wvoff, (int)wsoff, 0, 0);
      if constexpr (RB == 4)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)(dst + NW * 1024), 16, wvoff, (int)(wsoff + (unsigned)p.CoPx * 32u), 0, 0);
    }
    ++uw;
    if (uw == NU) { uw = 0; wsoff = (unsigned)wtile_co0_next * 16u; } else wsoff += wstep;
    slw = slw + 1 == R ? 0 : slw + 1;
    X3Q_ISSUED(NPW);
#ifdef RVC_X3P_CHECK
    mk_w[uwc & 7] = issued; ++uwc;
#endif
  };

  // ---- fp32 input: slot s of this wave = 8 channels (one half-plane) x 64 positions of the staged tile (conv_x3p.hip)
  float xr[XS][8];
  auto slot_geom = [&](int s, int& hb, int& q) -> bool {
    const int t = wave + NW * s;
    hb = t >= ni ? 1 : 0;
    q = (t - hb * ni) * 64 + lane;
    return t < 2 * ni && q < P;
  };
  auto load_slot = [&](int s, const QTile& tl, int chunk) {
    int hb, q;
    const bool ok = slot_geom(s, hb, q) && tl.valid;
    const int x = tl.bx + q;
    const unsigned voff = (ok && x >= 0 && x < p.Tin) ? (unsigned)x * 4u : kOOB;
    const unsigned c0 = (unsigned)(chunk * 16 + hb * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) xr[s][j] = buf_load(xrs, voff, (c0 + j) * (unsigned)p.ldX * 4u);
    X3Q_ISSUED(8);
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
        if constexpr (H2) split2h(fmaxf(a, a * pre_slope), fmaxf(b, b * pre_slope), h_, l_);
        else split2(fmaxf(a, a * pre_slope), fmaxf(b, b * pre_slope), h_, l_);
        hi[j] = h_; lo[j] = l_;
      }
      unsigned char* d = Xs + xb * xbuf + hb * xhalf + q * 16;
      *reinterpret_cast<u32x4*>(d) = hi;
      *reinterpret_cast<u32x4*>(d + xplane) = lo;
    }
  };
  // ---- split-resident input: chunk -> X buffer by DMA (conv_x3p.hip); a tile that does not exist repeats the current one's addresses
  const __amdgpu_buffer_rsrc_t xsr = make_rsrc(XSPLIT ? (const void*)p.Xs : (const void*)p.Wx, XSPLIT ? (unsigned)((long long)(p.Ci / 16) * 4 * p.xsTp * 16) : 0u);
  auto issue_x = [&](const QTile& tl, int chunk, int xb) {
    constexpr int pph = (BN + 64) / 64;
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
      const int pi = wave + NW * i;
      const int hp = pi / pph, j = pi - hp * pph;
      const unsigned row = (unsigned)((chunk * 4 + hp) * (int)p.xsTp + (tl.bx + kSplitMargin) + j * 64);
      unsigned char* dst = Xs + xb * xbuf + hp * xhalf + j * 1024;
      // (lanes past the half-plane's last row are switched off: the instruction still issues, so the counts below hold)
      if (j * 64 + lane < Pm)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xsr, (__attribute__((address_space(3))) void*)dst, 16, lane * 16, (int)(row * 16u), 0, 0);
    }
    X3Q_ISSUED(NPX);
#ifdef RVC_X3P_CHECK
    mk_xs[xb] = issued;
#endif
  };
  auto zero_edges = [&](const QTile& tl, int xb) {
    for (int q = tid0; q < P; q += NW * 64) {
      const int t = tl.bx + q;
      if (t >= 0 && t < p.Tin) continue;
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
        unsigned char* r = Xs + xb * xbuf + pl * xplane + q * 16;
        *reinterpret_cast<u32x4*>(r) = u32x4{0u, 0u, 0u, 0u}; *reinterpret_cast<u32x4*>(r + xhalf) = u32x4{0u, 0u, 0u, 0u};
      }
    }
  };
  // ---- residual: block b of tile tl = 16 rows (per lane) x 32 columns; out-of-range (no such block / tile / column): zeros, no traffic
  float rst[16];
  auto load_res = [&](const QTile& tl, int b) {
    if constexpr (RADD) {
      const int am = b / AN, an = b - am * AN;
      const int n = tl.n0 + (wn * AN + an) * 32 + li;
      const int mb = tl.co0 + (wm * AM + am) * 32 + 4 * lh;
      const unsigned voff = (tl.valid && b < NBLK && n < p.Tout) ? ((unsigned)mb * (unsigned)p.ldR + (unsigned)n) * 4u : kOOB;
#pragma unroll
      for (int r = 0; r < 16; ++r) rst[r] = buf_load(rrs, voff, (unsigned)((r & 3) + 8 * (r >> 2)) * (unsigned)p.ldR * 4u);
      X3Q_ISSUED(16);
#ifdef RVC_X3P_CHECK
      mk_r = issued;
#endif
    }
  };

  // ---- operand addresses
  const int aoff = lh * (BM * 16) + ((wm * AM) * 32 + li) * 16;
  const int boff = lh * xhalf + ((wn * AN) * 32 + li) * 16;

  // ---- prologue (once per workgroup): chunk 0 staged, chunk 1 on its way, weight units 0 .. R - 2, residual block 0; everything has landed
  // when the loop starts, so the steady-state counts hold from the first unit (fewer operations outstanding than they allow)
  if constexpr (XSPLIT) {
    issue_x(cur, 0, 0);
    issue_x(cur, 1, 1);
  } else {
#pragma unroll
    for (int s = 0; s < XS; ++s) load_slot(s, cur, 0);
    q_wait<0>();
#pragma unroll
    for (int s = 0; s < XS; ++s) store_slot(s, 0);
#pragma unroll
    for (int s = 0; s < XS; ++s) load_slot(s, cur, 1);
  }
#pragma unroll
  for (int i = 0; i < R - 1; ++i) issue_w();
  load_res(cur, 0);
  q_wait<0>();
#pragma unroll
  for (int s = 0; s < XS; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(xr[s][j]));
  lds_barrier();
  if (XSPLIT && cur.edge) { zero_edges(cur, 0); lds_barrier(); }

  f32x16 acc[AM][AN];
  u32x4 ah[AM], al[AM], bh[AN], bl[AN];
  {
    const unsigned char* wa = Ws + aoff;
    const unsigned char* xa = Xs + boff;
#pragma unroll
    for (int am = 0; am < AM; ++am) ah[am] = *reinterpret_cast<const u32x4*>(wa + am * 512);
#pragma unroll
    for (int an = 0; an < AN; ++an) bl[an] = *reinterpret_cast<const u32x4*>(xa + xplane + an * 512);
  }

  int sl = 0, xb = 0;                                        // weight slot of the current unit, input buffer of the current chunk
  t_last = XQTICK(); XQACC(1, t_last - t_begin);
  for (int it = 0; it < nit; ++it) {
    // accumulators start at the bias where residual and bias join the sum inside the tile (RADD), at zero otherwise (bias in the epilogue)
#pragma unroll
    for (int am = 0; am < AM; ++am) {
      f32x4q bv[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) bv[g] = RADD ? *reinterpret_cast<const f32x4q*>(Bs + cur.co0 + (wm * AM + am) * 32 + 4 * lh + 8 * g) : f32x4q{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int an = 0; an < AN; ++an)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[am][an][r] = bv[r >> 2][r & 3];
    }
    for (int c = 0; c < nck; ++c) {
      const bool c0 = c == 0;                                  // windows that reach back into the previous tile contain its epilogue
      // the stream's chunks c + 1 and c + 2
      const bool in1 = c + 1 < nck, in2 = c + 2 < nck;
      const QTile& t1 = in1 ? cur : nxt;
      const QTile& t2 = in2 ? cur : nxt;
      const int ch2 = in2 ? c + 2 : c + 2 - nck;
      auto unit = [&](auto tc) {
        constexpr int T = decltype(tc)::value;
        constexpr bool last_tap = T + 1 == KT;
        // registers hold hi_w (ah) and lo_x (bl) of this unit.  Groups: hi_w lo_x | hi_w hi_x | lo_w hi_x; the operand a group needs next is
        // requested before the group in front of it is issued, and nothing is copied: hi_w / lo_x of the NEXT unit land in ah / bl while the
        // third group (which reads neither) runs.  H2: w lo_x | w hi_x, no third group.
        {
          const unsigned char* xa = Xs + xb * xbuf + T * dil16 + boff;
#pragma unroll
          for (int an = 0; an < AN; ++an) bh[an] = *reinterpret_cast<const u32x4*>(xa + an * 512);
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- group 1: hi_w * lo_x
#pragma unroll
        for (int am = 0; am < AM; ++am)
#pragma unroll
          for (int an = 0; an < AN; ++an)
            acc[am][an] = q_mfma<false, H2>(ah[am], bl[an], acc[am][an]);
        if constexpr (!H2) {
          const unsigned char* wa = Ws + sl * wslot + BM * 32 + aoff;
#pragma unroll
          for (int am = 0; am < AM; ++am) al[am] = *reinterpret_cast<const u32x4*>(wa + am * 512);
        }
        // ---- step A: slot s of the stream's next chunk converted and stored, its registers refilled with the chunk after that
        if constexpr (!XSPLIT) {
          auto stage = [&](auto sc) {
            constexpr int s = decltype(sc)::value;
            if constexpr ((s * KT) / XS == T) {
              constexpr int NS = Q::sum() - 8;                 // younger than this slot's loads (issued in step A one chunk ago)
              // (check build: waits whose target was issued in the prologue - first chunk of the first tile - are exempt: everything had landed there)
              if (c0) { if (it > 0) X3Q_CHECK(NS + EP, issued - mk_x[s]); q_wait<NS + EP>(); }
              else { X3Q_CHECK(NS, issued - mk_x[s]); q_wait<NS>(); }
              store_slot(s, xb ^ 1);
              load_slot(s, t2, ch2);
            }
          };
          q_for<0, XS>(stage);
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- group 2: hi_w * hi_x
#pragma unroll
        for (int am = 0; am < AM; ++am)
#pragma unroll
          for (int an = 0; an < AN; ++an)
            acc[am][an] = q_mfma<false, H2>(ah[am], bh[an], acc[am][an]);
        __builtin_amdgcn_sched_barrier(0);
        // ---- next unit: its weight slot (and, at a chunk boundary, its input buffer) published; the slot of unit u - 1 refilled
        {
          constexpr int NWT = Q::w(T);
#ifdef RVC_X3P_CHECK
          const int exact = issued - mk_w[(uwc - (R - 2)) & 7];
#endif
          [[maybe_unused]] const long long ta = XQTICK();
          if (c0 && T < R - 2) { if (it > 0) X3Q_CHECK(NWT + EP, exact); q_wait<NWT + EP>(); }
          else { X3Q_CHECK(NWT, exact); q_wait<NWT>(); }
          // the split-resident chunk this unit's tail starts to read (requested one chunk ago, right behind that unit's weight request) is covered by the
          // same wait as long as the weight request waited for is not older than it: R - 2 <= KT - 1 (conv_x3q_try clamps the ring for 3-tap layers)
          static_assert(!XSPLIT || R - 2 <= KT - 1, "weight ring deeper than a chunk: the input chunk's DMA would not be covered by the weight wait");
          if constexpr (XSPLIT && last_tap) { if (it > 0 || c > 0) X3Q_CHECK(NWT, issued - mk_xs[xb ^ 1]); }
          [[maybe_unused]] const long long tb = XQTICK();
          lds_barrier();
          [[maybe_unused]] const long long tcc = XQTICK();
          XQACC(2, ta - t_last); XQACC(3, tb - ta); XQACC(4, tcc - tb); t_last = tcc;
          const int sn = sl + 1 == R ? 0 : sl + 1;
          if (XSPLIT && last_tap && t1.edge) { zero_edges(t1, xb ^ 1); lds_barrier(); }
          issue_w();                                             // unit u + R - 1 into the slot unit u - 1 was read from
          if constexpr (XSPLIT && last_tap) issue_x(t2, ch2, xb);     // this chunk's buffer is free: every wave is past its last read
          const unsigned char* wa = Ws + sn * wslot + aoff;
          const unsigned char* xa = Xs + (last_tap ? xb ^ 1 : xb) * xbuf + xplane + (last_tap ? 0 : (T + 1) * dil16) + boff;
#pragma unroll
          for (int am = 0; am < AM; ++am) ah[am] = *reinterpret_cast<const u32x4*>(wa + am * 512);
#pragma unroll
          for (int an = 0; an < AN; ++an) bl[an] = *reinterpret_cast<const u32x4*>(xa + an * 512);
          sl = sn;
        }
        // ---- step E: residual + bias of block c join the sum; block c + 1 (the next tile's block 0 in a tile's last chunk) requested
        if constexpr (RADD && T == TR) {
          constexpr int NR = Q::sum() - 16;
          if (c0) { if (it > 0) X3Q_CHECK(NR + EP, issued - mk_r); q_wait<NR + EP>(); }
          else { X3Q_CHECK(NR, issued - mk_r); q_wait<NR>(); }
          // The block that receives the residual is chosen at run time (block = chunk), its registers are not.  A branch the compiler can see
          // turns into copy-in / copy-out of the whole 16-register accumulator around every test (a phi of two 512-bit tuples: 256 v_mov per step
          // and 50 spilled registers), so the wave-uniform test lives INSIDE the asm statement: to the compiler every block is updated in place,
          // unconditionally.  The asm is invisible to the hazard recogniser: the leading s_nops cover the group-2 MFMA that last wrote the block
          // (8 passes: 11 wait states before a VALU may read its result), the trailing one the VALU write -> MFMA source-C read.
          auto radd = [&](auto bc) {
            constexpr int b = decltype(bc)::value, am = b / AN, an = b % AN;
            f32x16& A = acc[am][an];
            float (&V)[16] = rst;
            const int cc = c;
#define X3Q_ADD8(o)                                                                                                                                   \
            asm volatile("s_cmp_lg_u32 %16, %17\n\ts_cbranch_scc1 .Lx3q_skip_%=\n\ts_nop 15\n\ts_nop 7\n\t"                                                 \
                         "v_add_f32 %0, %0, %8\n\tv_add_f32 %1, %1, %9\n\tv_add_f32 %2, %2, %10\n\tv_add_f32 %3, %3, %11\n\t"                              \
                         "v_add_f32 %4, %4, %12\n\tv_add_f32 %5, %5, %13\n\tv_add_f32 %6, %6, %14\n\tv_add_f32 %7, %7, %15\n\ts_nop 3\n.Lx3q_skip_%=:"      \
                         : "+v"(A[o + 0]), "+v"(A[o + 1]), "+v"(A[o + 2]), "+v"(A[o + 3]), "+v"(A[o + 4]), "+v"(A[o + 5]), "+v"(A[o + 6]), "+v"(A[o + 7])    \
                         : "v"(V[o + 0]), "v"(V[o + 1]), "v"(V[o + 2]), "v"(V[o + 3]), "v"(V[o + 4]), "v"(V[o + 5]), "v"(V[o + 6]), "v"(V[o + 7]),           \
                           "s"(cc), "n"(b) : "scc")
            X3Q_ADD8(0);
            X3Q_ADD8(8);
#undef X3Q_ADD8
          };
          q_for<0, NBLK>(radd);
          load_res(t1, in1 ? c + 1 : 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- group 3: lo_w * hi_x (bf16x3 only.  Leaving it out with bf16 operands - the weights rounded to bf16 - was measured in round 5:
        // 130 LSB on the full-size goldens, profiles/r5_two_mfma.txt; the fp16 weight term of H2 is eight times finer.)
        if constexpr (!H2) {
#pragma unroll
          for (int am = 0; am < AM; ++am)
#pragma unroll
            for (int an = 0; an < AN; ++an)
              acc[am][an] = q_mfma<false>(al[am], bh[an], acc[am][an]);
        }
      };
      q_for<0, KT>(unit);
      xb ^= 1;
    }

    // ---- epilogue of tile `cur` (the next tile's first chunks and weight units are already on their way)
    [[maybe_unused]] const long long t_epi = XQTICK();
    XQACC(2, t_epi - t_last);
    if constexpr (YSPLIT) {
      // v = lrelu(acc + bias) as the bf16 (H2: fp16) hi / lo image the next layer stages by DMA (conv_x3_dev.h::ysplit_epilogue with the bias from LDS)
      const float sl2 = p.ys_slope;
      const __amdgpu_buffer_rsrc_t ysr = make_rsrc(p.Ys, (unsigned)((long long)(p.Co / 16) * 4 * p.ysTp * 16));
#pragma unroll
      for (int am = 0; am < AM; ++am)
#pragma unroll
        for (int an = 0; an < AN; ++an) {
          const int n = cur.n0 + (wn * AN + an) * 32 + li;
          const int mb = cur.co0 + (wm * AM + am) * 32;
          const long long pos = (long long)n + kSplitMargin;
#pragma unroll
          for (int g2 = 0; g2 < 2; ++g2) {
            const f32x4q ba = *reinterpret_cast<const f32x4q*>(Bs + mb + 16 * g2 + 4 * lh);
            const f32x4q bb = *reinterpret_cast<const f32x4q*>(Bs + mb + 16 * g2 + 4 * lh + 8);
            unsigned hA[2], lA[2], hB[2], lB[2];
#pragma unroll
            for (int e2 = 0; e2 < 2; ++e2) {
              float a0 = acc[am][an][8 * g2 + 2 * e2] + ba[2 * e2], a1 = acc[am][an][8 * g2 + 2 * e2 + 1] + ba[2 * e2 + 1];
              float b0 = acc[am][an][8 * g2 + 4 + 2 * e2] + bb[2 * e2], b1 = acc[am][an][8 * g2 + 5 + 2 * e2] + bb[2 * e2 + 1];
              if constexpr (H2) {
                split2h(fmaxf(a0, a0 * sl2), fmaxf(a1, a1 * sl2), hA[e2], lA[e2]);
                split2h(fmaxf(b0, b0 * sl2), fmaxf(b1, b1 * sl2), hB[e2], lB[e2]);
              } else {
                split2(fmaxf(a0, a0 * sl2), fmaxf(a1, a1 * sl2), hA[e2], lA[e2]);
                split2(fmaxf(b0, b0 * sl2), fmaxf(b1, b1 * sl2), hB[e2], lB[e2]);
              }
            }
            u32x4 hi, lo;
#pragma unroll
            for (int e2 = 0; e2 < 2; ++e2) {
              typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
              const u32x2_t sh = __builtin_amdgcn_permlane32_swap(hA[e2], hB[e2], false, false);
              const u32x2_t sl3 = __builtin_amdgcn_permlane32_swap(lA[e2], lB[e2], false, false);
              hi[e2] = sh.x; hi[2 + e2] = sh.y; lo[e2] = sl3.x; lo[2 + e2] = sl3.y;
            }
            // buffer stores, masked by an out-of-range offset instead of a branch: every tile issues the same number of operations
            const long long chunk = (mb >> 4) + g2;
            const unsigned off = n < p.Tout ? (unsigned)(((chunk * 4 + lh) * p.ysTp + pos) * 16) : kOOB;
            __builtin_amdgcn_raw_buffer_store_b128(hi, ysr, (int)off, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(lo, ysr, (int)off, (int)(unsigned)(p.ysTp * 32), 0);
          }
        }
      X3Q_ISSUED(4 * AM * AN);
    } else if constexpr (RADD) {
      // residual and bias are inside the sum: v = acc * scale [+ previous output].  The MFMA layout gives a lane ONE column: stored directly
      // that is 16 dword stores per accumulator (128 per tile) - and every store stands in the in-order vmcnt queue in front of the next tile's
      // first waits, whose window then exceeds the counter's 6 bits (measured: 11 - 19 % of a tile spent waiting for stores to drain).  Through
      // a 1-KiB staging area per wave a quarter block (8 channels x 32 positions) is re-read row-wise: a lane gets 4 consecutive positions,
      // 8 lanes a whole 128-byte line, 4 stores per accumulator.  One wave's LDS operations execute in order: no barrier.
      const __amdgpu_buffer_rsrc_t yrs = make_rsrc(p.Y, (unsigned)p.Co * (unsigned)p.ldY * 4u);
      const float oscale = p.out_scale;
      const bool has_acc = p.accumulate != 0;
      float* stg = reinterpret_cast<float*>(Ss + wave * 1024);
      const int rl = lane >> 3, cq = lane & 7;                  // read side: row rl of the quarter block, positions 4 cq .. 4 cq + 3
#pragma unroll
      for (int am = 0; am < AM; ++am)
#pragma unroll
        for (int an = 0; an < AN; ++an) {
          const int n = cur.n0 + (wn * AN + an) * 32 + 4 * cq;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            // quarter block = the 8 channels 8 g + 4 lh + e of the accumulator's 32: staged row 4 lh + e
#pragma unroll
            for (int e = 0; e < 4; ++e) stg[(4 * lh + e) * 32 + li] = acc[am][an][4 * g + e] * oscale;
            const int m = cur.co0 + (wm * AM + am) * 32 + 8 * g + rl;
            const unsigned row = (unsigned)m * (unsigned)p.ldY;
            f32x4q v = *reinterpret_cast<const f32x4q*>(stg + rl * 32 + 4 * cq);
            if (has_acc) {
              const u32x4 y = __builtin_amdgcn_raw_buffer_load_b128(yrs, (int)(n < p.Tout ? (row + (unsigned)n) * 4u : kOOB), 0, 0);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] += __uint_as_float(y[e]);
            }
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = __float_as_uint(v[e]);
            // masked by an out-of-range offset (every tile issues the same operations); the group that straddles the end goes element-wise
            __builtin_amdgcn_raw_buffer_store_b128(o, yrs, (int)(n + 3 < p.Tout ? (row + (unsigned)n) * 4u : kOOB), 0, 0);
            if (n < p.Tout && n + 3 >= p.Tout) {
#pragma unroll
              for (int e = 0; e < 3; ++e)
                if (n + e < p.Tout) __builtin_amdgcn_raw_buffer_store_b32(o[e], yrs, (int)((row + (unsigned)(n + e)) * 4u), 0, 0);
            }
#ifdef RVC_X3P_CHECK
#pragma unroll
            for (int e = 0; e < 3; ++e) if (__ballot(n < p.Tout && n + 3 >= p.Tout && n + e < p.Tout)) X3Q_ISSUED(1);      // (an instruction issues when any lane is active)
#endif
          }
        }
      X3Q_ISSUED(4 * AM * AN + (p.accumulate ? 4 * AM * AN : 0));
    } else {
      dense_epilogue<WM, WN, AM, AN, 4>(p, acc, 0, cur.co0, cur.n0, wm, wn, li, lh);
      X3Q_ISSUED(EP);                                          // (a lower bound here: the check build only knows the stores)
    }
    XQACC(5, XQTICK() - t_epi); XQACC(0, 1);
    t_last = XQTICK();
    cur = nxt; nxt = geom(it + 2);
    wtile_co0_next = nxt.valid ? nxt.co0 : cur.co0;
  }
  q_wait<0>();                                                  // (requests past the end: nothing may land in LDS after the workgroup has gone)
#ifdef RVC_CONV_TIMING
  { XQACC(6, XQTICK() - t_begin);
    if (threadIdx.x == 0) for (int i = 0; i < 8; ++i) atomicAdd(&g_x3q_timing[i], (unsigned long long)xqt[i]); }
#endif
}


// ============================================================================ host side
template <int AM, int AN, int KT, int R, bool XSPLIT, bool YSPLIT, bool RADD, bool H2>
static void launch_x3q(const ConvArgsX& a, dim3 grid, size_t lds, hipStream_t s) {
  auto kern = conv_x3q_kernel<AM, AN, KT, R, XSPLIT, YSPLIT, RADD, H2>;
  RVC_ALLOW_BIG_LDS(kern);
  conv_launch(kern, grid, dim3(256), lds, s, a);
}
template <int AM, int AN, int KT, int R>
static void launch_x3q_io(const ConvArgsX& a, int mode, dim3 grid, size_t lds, hipStream_t s) {
  constexpr bool c2_ok = R - 2 <= KT - 1;      // image in: the ring may not be deeper than a chunk (see the static_assert in the kernel; conv_x3q_try clamps R)
  if (a.h2) {
    if (mode == 0) launch_x3q<AM, AN, KT, R, false, true, false, true>(a, grid, lds, s);
    else if constexpr (c2_ok) launch_x3q<AM, AN, KT, R, true, false, true, true>(a, grid, lds, s);
  } else {
    if (mode == 0) launch_x3q<AM, AN, KT, R, false, true, false, false>(a, grid, lds, s);      // fp32 in, image out (c1 of a split pair)
    else if constexpr (c2_ok) launch_x3q<AM, AN, KT, R, true, false, true, false>(a, grid, lds, s);   // image in, fp32 out + residual (c2 of a split pair)
  }
  if (mode != 0 && !c2_ok) throw Error("conv_x3q: weight ring deeper than a chunk on an image-in launch");
}
template <int AM, int AN, int R>
static void launch_x3q_k(const ConvArgsX& a, int mode, dim3 grid, size_t lds, hipStream_t s) {
  if (a.ktaps == 3) launch_x3q_io<AM, AN, 3, R>(a, mode, grid, lds, s);
  else if (a.ktaps == 7) launch_x3q_io<AM, AN, 7, R>(a, mode, grid, lds, s);
  else launch_x3q_io<AM, AN, 11, R>(a, mode, grid, lds, s);
}
template <int AM, int AN>
static void launch_x3q_r(const ConvArgsX& a, int R, int mode, dim3 grid, size_t lds, hipStream_t s) {
  if constexpr (AM == 2 && AN == 4) {
    if (R >= 5) launch_x3q_k<AM, AN, 5>(a, mode, grid, lds, s); else launch_x3q_k<AM, AN, 4>(a, mode, grid, lds, s);
  } else {
    if (R >= 5) launch_x3q_k<AM, AN, 5>(a, mode, grid, lds, s); else if (R == 4) launch_x3q_k<AM, AN, 4>(a, mode, grid, lds, s); else launch_x3q_k<AM, AN, 3>(a, mode, grid, lds, s);
  }
}

// a: arguments as conv_x3_try prepared them (true taps, tile chosen: WM = WN = 2).  Returns false when the layer is not this kernel's (the
// per-tile pipelined kernel of conv_x3p.hip takes it): the two halves of a split-resident ResBlock pair - stride-1 Conv1d with 3 / 7 / 11 taps,
// whole row tiles, at least three 16-channel chunks, at least 8 tiles.
//
// a.h2 (fp16x2): a.Wx is the layer's ONE-plane fp16 image (ConvLayer::Wh_) and the pair's intermediate image is fp16 hi / lo.  Accuracy: the
// activation keeps 22 bits (bf16x3: 24 with the 2^-18 lo lo term dropped), the weight 11 instead of 16 - every product carries a relative weight
// rounding of <= 2^-12.  Measured on the CPU oracle with the ResBlock weights of the three wide stages rounded to fp16 and everything else exact
// (tools/exp/fp16_weight_rounding.py): max 16 LSB / mean 2.2 on the 30 s golden (gate 33; bf16 weights: 142 / 16.9, which is what round 5
// measured on the GPU for the bf16 2-term variant).  Only the pair convolutions of a residual branch are offered this arithmetic: their errors
// enter the stage tensor additively beside the exact skip path.  fp16 range: hi saturates at 65504 (round toward zero), |x| < 131008 stays finite.
bool conv_x3q_try(ConvArgsX& a, int AM, int AN, hipStream_t s, dim3& grid_out, bool dry) {
  static const int on = exp_int("RVC_X3Q", 1);
  if (!on) return false;
  const bool xs = a.Xs != nullptr, ys = a.Ys != nullptr;
  if (a.Wd > 0 || (a.Ci & 15) || a.Ci < 48 || xs == ys || a.stride != 1 || a.ostride != 1 || a.orows != a.Co) return false;
  if (!(a.ktaps == 3 || a.ktaps == 7 || a.ktaps == 11)) return false;
  if (!((AM == 2 && AN == 4) || (AM == 1 && AN == 4) || (AM == 2 && AN == 2))) return false;
  const int BM = 64 * AM, BN = 64 * AN;
  if (a.Co % BM || a.Co > 1024) return false;
  if (xs && !(a.R != nullptr && a.act == ACT_NONE)) return false;            // image in: c2 of a pair (residual, nothing after the sum)
  if (ys && (a.R || a.accumulate)) return false;
  if ((double)(a.Co / 16) * 4.0 * (double)a.ysTp * 16.0 >= 2147483648.0 || (double)(a.Ci / 16) * 4.0 * (double)a.xsTp * 16.0 >= 2147483648.0) return false;
  const int mode = ys ? 0 : 1;
  const int P = BN + (a.ktaps - 1) * a.dil;
  if (P > 384 || P > BN + 64) return false;
  const int Pm = (P + 7) & ~7;
  const int xbytes = (2 * 2 * Pm * 32 + 1023) & ~1023;
  const int wslot = (a.h2 ? 1 : 2) * BM * 32;
  const int per_cu = AM * AN >= 8 ? 2 : 3;
  // (LDS is allocated in granules: a footprint a few hundred bytes under a third of 160 KiB still admits only two workgroups per CU - measured,
  // C64 k7: ring of 4 at 54 528 B 279 us, ring of 3 251 us - so the budget is cut to whole 2-KiB granules)
  const size_t budget = (size_t)(160 * 1024 / per_cu) & ~(size_t)2047;
  const size_t fixed = (size_t)xbytes + (size_t)((a.Co * 4 + 255) & ~255) + (mode == 1 ? 4096 : 0);   // (mode 1: the epilogue's staging area)
  static const int r_env = exp_int("RVC_X3Q_R", 0);
  const int rmin = (AM == 2 && AN == 4) ? 4 : 3;
  // image in, three taps: at most four slots - the input chunk's DMA (requested KT units before its first read) is covered by the wait for the
  // weight unit requested R - 2 units before, which must not be the older of the two (latent in rounds 4 - 5 with R = 5: the window was a whole chunk
  // of MFMAs wide and never observed; the shorter fp16x2 units of round 6 made the full-size repeat test differ)
  int R = (mode == 1 && a.ktaps == 3) ? 4 : 5;
  while (R > rmin && fixed + (size_t)R * wslot > budget) --R;
  if (r_env >= rmin && r_env < R) R = r_env;
  const size_t lds = fixed + (size_t)R * wslot;
  if (lds > budget) return false;
  const long long ntiles = (long long)((a.Tout + BN - 1) / BN) * (a.Co / BM);
  // (per device: one process may drive several GPUs, and the persistent grid is sized by the CURRENT device's CU count)
  static std::atomic<int> ncu_of[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64) dev = 0;
  int ncu = ncu_of[dev].load(std::memory_order_relaxed);
  if (ncu == 0) { int n = 256; (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); ncu = n > 0 ? n : 256; ncu_of[dev].store(ncu, std::memory_order_relaxed); }
  static const int wg_env = exp_int("RVC_X3Q_WGS", 0);     // workgroups per CU (0: what the tile's LDS / registers admit)
  // (RVC_X3Q_MINROUNDS: grids of fewer rounds of resident workgroups stay on the per-tile kernel.  Default 0: every eligible pair, also where a
  // workgroup owns a single tile - the 256-channel stage, short clips: measured neutral there against the per-tile kernel (C256 pairs 1018 -> 1008 us),
  // one kernel for every ResBlock pair of the three wide stages)
  static const int min_rounds = exp_int("RVC_X3Q_MINROUNDS", 0);
  const long long slots = (long long)(wg_env > 0 ? wg_env : per_cu) * ncu;
  if (ntiles < min_rounds * slots || ntiles < 8) return false;
  // a multiple of 8 (a workgroup's later tiles stay on its XCD) unless every workgroup owns exactly one tile; decided before the dry-run answer and
  // before `a` is touched, so that "yes" in the dry run is "launched" in the real call
  const long long G = ntiles <= slots ? ntiles : (slots & ~7LL);
  if (G < 8) return false;
  if (dry) return true;
  a.WROW = P; a.ni = (P + 63) / 64; a.nchunk = a.Ci / 16; a.NC = 1; a.KT = 1; a.xbufs = 2; a.ksplit = 1; a.partial = nullptr; a.wbufs = R;
  static const int xcd_env = exp_int("RVC_X3_XCD", 1);
  a.xcd_remap = xcd_env;
  dim3 grid((unsigned)G, 1, 1);
  grid_out = grid;
  if (AM == 2 && AN == 4) launch_x3q_r<2, 4>(a, R, mode, grid, lds, s);
  else if (AM == 1 && AN == 4) launch_x3q_r<1, 4>(a, R, mode, grid, lds, s);
  else launch_x3q_r<2, 2>(a, R, mode, grid, lds, s);
  return true;
}

}  // namespace rvc
