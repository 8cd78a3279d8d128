// Persistent fused ResBlock pair of the generator's 32-channel stage in the fp16x2 arithmetic, weights RESIDENT in LDS (gfx950 only).
//
//   y = (x + c2(lrelu(c1_d(lrelu(x)) + b1)) + b2) * scale [+ y]        (reference lib/infer_pack/modules.py:295-308, one (c1, c2) pair of ResBlock1)
//
// What conv_x3pf_kernel (conv_x3p.hip) does for this pair costs 118 - 238 us per launch at T = 1 279 200 (profiles/r5y_launch_classes.md) with the matrix pipe
// 19 % busy and HBM at 2.1 - 2.7 TB/s: its tile is 32 rows x 256 columns, so a (chunk, tap) unit is six MFMAs per wave between two barriers and a weight-ring
// wait - 44 barriers per tile at 11 taps - and three 4-wave workgroups per CU have nothing else to hide them with.  With ONE fp16 term per weight (the pair
// arithmetic of round 6, conv_x3q.hip) both convolutions' weights are 2 x KT x 2 KiB <= 44 KiB: they fit LDS for the whole launch.  So here
//   * a workgroup (8 waves, one per CU) loads both weight sets ONCE and walks over its tiles (persistent: grid = number of CUs);
//   * a tile is 32 rows x 512 intermediate columns (wave w owns columns 64 w .. 64 w + 63 of both convolutions: nothing but the input tile and the
//     intermediate is shared), 4 barriers per tile instead of 4 KT + 2: input staged | conv1 | intermediate written over the input | conv2 + epilogue;
//   * the residual (x again: L2) is requested before the first convolution and becomes - with the bias - the initial value of the second one's
//     accumulators; the next tile's input rows and, when accumulating (ACC), the previous output are requested into registers before the second
//     convolution and consumed after it: HBM latency sits behind 88 (KT = 11) MFMAs per wave;
//   * per (chunk, tap): one weight operand read serves 2 column blocks x 2 terms; activations as fp16 hi / lo rows (split2h), two MFMAs per product.
// Where its time goes (per-phase cycle counters, -DRVC_CONV_TIMING, profiles/r6c_rbh_phase_cycles.txt): the two convolutions take 3.6 k cycles each of a
// 24 k-cycle tile at 11 taps; staging, the intermediate and the epilogue stretch as the convolutions get shorter - a CU moves ~100 KB per 250 outputs
// (input tile with halo, the residual again, the output) at 8 - 9.4 B / cycle, the vector-memory path's limit: a 3-tap pair takes ~100 us whichever kernel
// runs it.  Three restructurings were built and measured in round 6 and NOT kept (same file's history, same evidence file): 16-byte global accesses with
// a per-wave LDS transposition (26 instead of 104 VMEM instructions per wave and tile: slower, 1269 us for the nine pairs against 1089); two halves of four
// waves one phase apart, so that a matrix phase always runs beside a memory phase (1117 - 1143 us); the residual recovered from the staged fp16 tile instead
// of read again (1083 us).  What would move it is fewer bytes per output: a whole ResBlock (three pairs) per launch.
// LDS: 2 KT x 2 KiB weights + 256 B biases + 2 chunks x (hi | lo) x 2 halves x P rows x 16 B, P = 512 + (KT - 1) dil <= 562: 118 KiB at KT = 11.
// Numerics: the formula, the fp16 weight rounding and the hi / lo split are those of the persistent pair kernel's H2 mode; fp32 accumulation, bias and
// residual in fp32.  The bf16x3 arithmetic (rvc_set_pair_arithmetic(0), or a layer without an fp16 image) keeps conv_x3pf_kernel.
#include "conv_x3_dev.h"

namespace rvc {

template <int T, int N, class F> __device__ __forceinline__ void rbh_for(F& f) {
  if constexpr (T < N) { f(std::integral_constant<int, T>{}); rbh_for<T + 1, N>(f); }
}
typedef float f32x4r __attribute__((ext_vector_type(4)));

#ifdef RVC_CONV_TIMING
__device__ unsigned long long g_rbh_timing[8];   // [0] tiles, [1] stage + barrier, [2] conv1, [3] barrier + h + requests + barrier, [4] conv2, [5] epilogue + barrier, [6] total
void conv_rbh_timing_read(unsigned long long* out8, bool reset) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_rbh_timing), sizeof(unsigned long long) * 8);
  if (reset) { unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_rbh_timing), z, sizeof(z)); }
}
#define RBTICK() ((long long)__builtin_readcyclecounter())
#define RBACC(i, v) do { rbt[i] += (v); } while (0)
#else
void conv_rbh_timing_read(unsigned long long* out8, bool) { for (int i = 0; i < 8; ++i) out8[i] = 0; }
#define RBTICK() 0ll
#define RBACC(i, v) do {} while (0)
#endif

template <int KT, bool ACC>
__global__ __launch_bounds__(512, 2) void conv_rbh_kernel(const ConvArgsX p) {
  constexpr int C = 32, NCK = 2, NW = 8, AN = 2, BN = NW * AN * 32, P2 = (KT - 1) / 2, NO = BN - 2 * P2;
  constexpr int NU = NCK * KT;                               // (chunk, tap) units of one convolution
  constexpr int WB = NU * 2 * C * 16;                        // bytes of one convolution's weights: [unit][half][32 rows][16 B]
  constexpr int NIMAX = 9;                                   // 64-column groups of the staged tile: ceil(562 / 64)
  constexpr int XS = (NCK * 2 * NIMAX + NW - 1) / NW;        // staging slots per wave (8 channels x 64 positions each): 5
  extern __shared__ __attribute__((aligned(1024))) unsigned char smemr[];
  const int P = p.WROW;                                      // staged input columns: 512 + (KT - 1) dil
  const int xplane = P * 32, xhalf = P * 16, xbuf = 2 * xplane;
  unsigned char* W1s = smemr;
  unsigned char* W2s = smemr + WB;
  float* Bs = reinterpret_cast<float*>(smemr + 2 * WB);      // b1[32] | b2[32]
  unsigned char* Xs = smemr + 2 * WB + 256;

  const int tid0 = threadIdx.x;
  const int lane = tid0 & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int ni = p.ni;                                       // ceil(P / 64)
  const int dil16 = p.dil * 16;
  const float pre_slope = p.pre_slope, hs = p.fuse_slope, oscale = p.out_scale;
  const int T = p.Tout;
  const int ntiles = (T + NO - 1) / NO;
  const __amdgpu_buffer_rsrc_t xrs = make_rsrc(p.X, (unsigned)C * (unsigned)p.ldX * 4u);
  const __amdgpu_buffer_rsrc_t yrs = make_rsrc(p.Y, (unsigned)C * (unsigned)p.ldY * 4u);

  // ---- both weight sets and the biases, once per workgroup (one-plane fp16 images: [chunk][tap][half][CoPx rows][8 ch], rows 0 .. 31 used)
  for (int r = tid0; r < 2 * NU * 2 * C; r += NW * 64) {
    const int conv = r / (NU * 2 * C), q = r - conv * (NU * 2 * C);
    const int uh = q / C, m = q - uh * C;                    // (unit, half), row
    const unsigned char* src = (conv ? p.Wx2 : p.Wx) + ((long long)uh * p.CoPx + m) * 16;
    *reinterpret_cast<u32x4*>((conv ? W2s : W1s) + (uh * C + m) * 16) = *reinterpret_cast<const u32x4*>(src);
  }
  if (tid0 < 64) Bs[tid0] = tid0 < 32 ? (p.bias1 ? p.bias1[tid0] : 0.f) : (p.bias ? p.bias[tid0 - 32] : 0.f);

  // ---- input rows of a tile -> registers (slot s of this wave = 8 channels of one half-chunk x 64 positions), later -> fp16 hi / lo rows in LDS
  float xr[XS][8];
  auto slot_geom = [&](int s, int& cc, int& hb, int& q) -> bool {
    const int t = wave + NW * s;
    cc = t / (2 * ni);
    const int g = t - cc * 2 * ni;
    hb = g >= ni ? 1 : 0;
    q = (g - hb * ni) * 64 + lane;
    return t < NCK * 2 * ni && q < P;
  };
  auto load_x = [&](int tile) {
    const int bx = tile * NO - P2 - p.pad;                   // first staged input position
#pragma unroll
    for (int s = 0; s < XS; ++s) {
      int cc, hb, q;
      const bool ok = slot_geom(s, cc, hb, q) && tile < ntiles;
      const int x = bx + q;
      const unsigned voff = (ok && x >= 0 && x < p.Tin) ? (unsigned)x * 4u : kOOB;
      const unsigned c0 = (unsigned)(cc * 16 + hb * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) xr[s][j] = buf_load(xrs, voff, (c0 + j) * (unsigned)p.ldX * 4u);
    }
  };
  auto stage_x = [&]() {
#pragma unroll
    for (int s = 0; s < XS; ++s) {
      int cc, hb, q;
      if (slot_geom(s, cc, hb, q)) {
        u32x4 hi, lo;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float a = xr[s][2 * j], b = xr[s][2 * j + 1];
          unsigned h_, l_;
          split2h(fmaxf(a, a * pre_slope), fmaxf(b, b * pre_slope), h_, l_);
          hi[j] = h_; lo[j] = l_;
        }
        unsigned char* d = Xs + cc * xbuf + hb * xhalf + q * 16;
        *reinterpret_cast<u32x4*>(d) = hi;
        *reinterpret_cast<u32x4*>(d + xplane) = lo;
      }
    }
  };

  // ---- one convolution over the wave's 64 columns: NU units, the operands of unit u + 1 requested before the MFMAs of unit u
  const int aoff = (lh * C + li) * 16;
  const int boff = lh * xhalf + (wave * AN * 32 + li) * 16;
  f32x16 acc[AN];
  auto conv = [&](const unsigned char* W, int d16) {
    u32x4 a, bh[AN], bl[AN], an_, bhn[AN], bln[AN];
    auto read_ops = [&](u32x4& a_, u32x4 (&b_h)[AN], u32x4 (&b_l)[AN], int u, int xoff) {
      a_ = *reinterpret_cast<const u32x4*>(W + u * (2 * C * 16) + aoff);
      const unsigned char* xa = Xs + xoff + boff;
#pragma unroll
      for (int j = 0; j < AN; ++j) { b_h[j] = *reinterpret_cast<const u32x4*>(xa + j * 512); b_l[j] = *reinterpret_cast<const u32x4*>(xa + xplane + j * 512); }
    };
    read_ops(a, bh, bl, 0, 0);
    auto unit = [&](auto uc) {
      constexpr int U = decltype(uc)::value;
      if constexpr (U + 1 < NU) {
        constexpr int Tn = (U + 1) % KT, Cn = (U + 1) / KT;
        read_ops(an_, bhn, bln, U + 1, Cn * xbuf + Tn * d16);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < AN; ++j)
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, bl[j]), acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < AN; ++j)
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, bh[j]), acc[j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (U + 1 < NU) {
        a = an_;
#pragma unroll
        for (int j = 0; j < AN; ++j) { bh[j] = bhn[j]; bl[j] = bln[j]; }
      }
    };
    rbh_for<0, NU>(unit);
  };

#ifdef RVC_CONV_TIMING
  long long rbt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  load_x((int)blockIdx.x);
  __syncthreads();                                            // weights and biases are in LDS
  [[maybe_unused]] long long tq = RBTICK();
  [[maybe_unused]] const long long tq0 = tq;
  float rr[AN][16];
  [[maybe_unused]] float yo[AN][16];
  for (int tile = (int)blockIdx.x; tile < ntiles; tile += (int)gridDim.x) {
    const int n0 = tile * NO;
    // ---- phase 1: this tile's input rows (requested during the previous tile's second convolution) -> LDS; the residual rows are requested
    stage_x();
#pragma unroll
    for (int j = 0; j < AN; ++j) {
      const int no = (wave * AN + j) * 32 + li;
      const int n = n0 + no;
      const bool ok = no < NO && n < T;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        rr[j][r] = buf_load(xrs, ok ? ((unsigned)((r & 3) + 8 * (r >> 2) + 4 * lh) * (unsigned)p.ldX + (unsigned)n) * 4u : kOOB);
    }
    __syncthreads();
    { [[maybe_unused]] const long long t = RBTICK(); RBACC(1, t - tq); tq = t; }
    // ---- phase 2: the dilated convolution over the 512 intermediate columns
#pragma unroll
    for (int j = 0; j < AN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    conv(W1s, dil16);
    { [[maybe_unused]] const long long t = RBTICK(); RBACC(2, t - tq); tq = t; }
    __syncthreads();                                          // every wave is done with the input tile
    // ---- phase 3: h = lrelu(acc + b1) (0 outside the sequence: the second convolution's zero padding) as fp16 hi / lo rows over the input tile
#pragma unroll
    for (int j = 0; j < AN; ++j) {
      const int nl = (wave * AN + j) * 32 + li;
      const int pos = n0 - P2 + nl;
      const bool inside = pos >= 0 && pos < T;
#pragma unroll
      for (int g2 = 0; g2 < 2; ++g2) {
        const f32x4r ba = *reinterpret_cast<const f32x4r*>(Bs + 16 * g2 + 4 * lh);
        const f32x4r bb = *reinterpret_cast<const f32x4r*>(Bs + 16 * g2 + 4 * lh + 8);
        unsigned hA[2], lA[2], hB[2], lB[2];
#pragma unroll
        for (int e2 = 0; e2 < 2; ++e2) {
          float a0 = acc[j][8 * g2 + 2 * e2] + ba[2 * e2], a1 = acc[j][8 * g2 + 2 * e2 + 1] + ba[2 * e2 + 1];
          float b0 = acc[j][8 * g2 + 4 + 2 * e2] + bb[2 * e2], b1 = acc[j][8 * g2 + 5 + 2 * e2] + bb[2 * e2 + 1];
          a0 = inside ? fmaxf(a0, a0 * hs) : 0.f; a1 = inside ? fmaxf(a1, a1 * hs) : 0.f;
          b0 = inside ? fmaxf(b0, b0 * hs) : 0.f; b1 = inside ? fmaxf(b1, b1 * hs) : 0.f;
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
        unsigned char* d = Xs + g2 * xbuf + lh * xhalf + nl * 16;
        *reinterpret_cast<u32x4*>(d) = hi;
        *reinterpret_cast<u32x4*>(d + xplane) = lo;
      }
    }
    // ---- requests that the second convolution hides: the next tile's input rows and the previous output
    load_x(tile + (int)gridDim.x);
    if constexpr (ACC) {
#pragma unroll
      for (int j = 0; j < AN; ++j) {
        const int no = (wave * AN + j) * 32 + li;
        const int n = n0 + no;
        const bool ok = no < NO && n < T;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          yo[j][r] = buf_load(yrs, ok ? ((unsigned)((r & 3) + 8 * (r >> 2) + 4 * lh) * (unsigned)p.ldY + (unsigned)n) * 4u : kOOB);
      }
    }
    // ---- phase 4: the second convolution (dilation 1), its accumulators starting at residual + bias, and the epilogue
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4r b2 = *reinterpret_cast<const f32x4r*>(Bs + 32 + 8 * g + 4 * lh);
#pragma unroll
      for (int j = 0; j < AN; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[j][4 * g + e] = rr[j][4 * g + e] + b2[e];
    }
    __syncthreads();                                          // the intermediate is complete
    { [[maybe_unused]] const long long t = RBTICK(); RBACC(3, t - tq); tq = t; }
    conv(W2s, 16);
    { [[maybe_unused]] const long long t = RBTICK(); RBACC(4, t - tq); tq = t; }
#pragma unroll
    for (int j = 0; j < AN; ++j) {
      const int no = (wave * AN + j) * 32 + li;
      const int n = n0 + no;
      const bool ok = no < NO && n < T;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
        float v = acc[j][r] * oscale;
        if constexpr (ACC) v += yo[j][r];
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), yrs, (int)(ok ? ((unsigned)m * (unsigned)p.ldY + (unsigned)n) * 4u : kOOB), 0, 0);
      }
    }
    __syncthreads();                                          // every wave is done with the intermediate: the next tile may be staged over it
    { [[maybe_unused]] const long long t = RBTICK(); RBACC(5, t - tq); tq = t; RBACC(0, 1); }
  }
#ifdef RVC_CONV_TIMING
  rbt[6] = RBTICK() - tq0;
  if (threadIdx.x == 0) for (int i = 0; i < 8; ++i) atomicAdd(&g_rbh_timing[i], (unsigned long long)rbt[i]);
#endif
}

template <int KT, bool ACC>
static void launch_rbh2(const ConvArgsX& a, dim3 grid, size_t lds, hipStream_t s) {
  auto kern = conv_rbh_kernel<KT, ACC>;
  RVC_ALLOW_BIG_LDS(kern);
  conv_launch(kern, grid, dim3(512), lds, s, a);
}
template <int KT>
static void launch_rbh(const ConvArgsX& a, dim3 grid, size_t lds, hipStream_t s) {
  if (a.accumulate) launch_rbh2<KT, true>(a, grid, lds, s); else launch_rbh2<KT, false>(a, grid, lds, s);
}

// a: the fused pair's arguments as conv_x3_pair_try prepared them, with Wx / Wx2 = the two layers' ONE-plane fp16 images (ConvLayer::Wh_).  32 channels,
// 3 / 7 / 11 taps, a sequence of at least two rounds of tiles; false: not this kernel's (conv_x3pf_kernel takes the pair in bf16x3).
bool conv_rbh_try(ConvArgsX& a, int T, hipStream_t s, dim3& grid_out, bool dry) {
  static const int on = exp_int("RVC_RBH", 1);
  if (!on || a.Ci != 32 || a.Co != 32 || !(a.ktaps == 3 || a.ktaps == 7 || a.ktaps == 11) || a.CoPx < 32) return false;
  const int BN = 512, NO = BN - (a.ktaps - 1);
  const int P = BN + (a.ktaps - 1) * a.dil;
  if (P > 576 || a.dil < 1) return false;
  int dev = 0, ncu = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
  if (ncu <= 0) ncu = 256;
  const long long ntiles = ((long long)T + NO - 1) / NO;
  if (ntiles < 2LL * ncu) return false;                       // short sequences: conv_x3pf_kernel's smaller tiles fill the chip better
  if (dry) return true;
  a.WROW = P; a.ni = (P + 63) / 64;
  const size_t lds = (size_t)2 * (2 * a.ktaps) * 2 * 32 * 16 + 256 + (size_t)2 * 2 * P * 32;
  dim3 grid((unsigned)(ntiles < ncu ? ntiles : ncu), 1, 1);
  grid_out = grid;
  if (a.ktaps == 3) launch_rbh<3>(a, grid, lds, s); else if (a.ktaps == 7) launch_rbh<7>(a, grid, lds, s); else launch_rbh<11>(a, grid, lds, s);
  return true;
}

}  // namespace rvc
