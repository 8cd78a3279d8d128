// One ConvBlockRes of RMVPE's U-Net at the two shallow levels (16 and 32 channels; reference lib/rmvpe.py:233-268) in ONE launch (gfx950 only):
//   out = relu(conv3x3_2(relu(conv3x3_1(x) + b1)) + b2) + x        (BatchNorm folded into both convolutions, in == out channels)
//
// At these levels the tensors are large (16 x 3232 x 128 fp32 = 26 MB) and the arithmetic small (0.15 GFLOP per MB): the staged kernel
// ran each convolution at 0.17 of the HBM rate (profiles/r3m_launch_classes.md: 50 us / 26 us per launch, 28 launches per clip).  Here a
// workgroup owns a TH x 64 tile of positions:
//   * the input tile with a 2-position halo is converted ONCE to the bf16 hi / lo image the MFMAs read ([chunk][hi | lo][half][position][8 ch],
//     position = (row + 2) P + (col + 2), P = 68: a 3 x 3 tap is the constant offset dh P + dw) - positions outside the image are zeros, the
//     convolution's padding;
//   * the weights of a convolution (9 or 18 (chunk, tap) units x hi / lo) live in REGISTERS: the only LDS reads are the position operands,
//     two 16-byte reads per three MFMAs;
//   * conv1 runs over the tile + 1 halo; its result (bias, ReLU, zero outside the image) stays in accumulators until every wave has finished
//     reading x, then overwrites x in LDS as the image conv2 reads - the intermediate never leaves the CU;
//   * conv2's epilogue adds bias, ReLU and the fp32 residual (read again from global: L2) and stores fp32 rows.
#include "conv_x3_dev.h"
#include "ops.h"
#include <type_traits>

namespace rvc {

#ifdef RVC_CONV_TIMING
__device__ unsigned long long g_cbr_timing[8];   // [0] workgroups, [1] staging (loads + convert), [2] conv1, [3] y1 -> LDS, [4] conv2, [5] epilogue, [6] total
#define CTICK() wall_clock64()
#define CTACC(i, v) do { if (threadIdx.x == 0) atomicAdd(&g_cbr_timing[i], (unsigned long long)(v)); } while (0)
void cbr2_timing_read(unsigned long long* out8, bool reset) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_cbr_timing), sizeof(unsigned long long) * 8);
  if (reset) { unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_cbr_timing), z, sizeof(z)); }
}
#else
#define CTICK() 0ull
#define CTACC(i, v) do {} while (0)
#endif

// LDS bytes of cbr2_small_kernel<C, TH> (the kernel's NPOS, restated for the launch)
constexpr size_t cbr2_lds_bytes(int C, int TH) {
  const int P = 68, NP1 = (TH + 4) * P, NB1 = ((TH + 2) * P + 31) / 32, NB2 = (TH * P + 31) / 32;
  const int END1 = P + 32 * NB1 + P + 2, END2 = 2 * P + 32 * NB2 + P + 2, ENDM = END1 > END2 ? END1 : END2;
  const int SLACK = ENDM > NP1 ? ((ENDM - NP1 + 15) & ~15) : 16;
  return (size_t)(C / 16) * 4 * (16 + NP1 + SLACK) * 16;
}

struct Cbr2Args {
  const float* __restrict__ X; float* __restrict__ Y; long long plane;      // [C][H W] fp32, channel pitch `plane`
  int H, W;
  const unsigned char* W1; const unsigned char* W2; int CoPx1, CoPx2;      // bf16x3 weight images (pack_x3, 9 taps)
  const float* b1; const float* b2;
};

template <int C, int TH>
__global__ __launch_bounds__(256, 2) void cbr2_small_kernel(const Cbr2Args p) {
  constexpr int TW = 64, P = TW + 4, NCH = C / 16, NU = NCH * 9;
  constexpr int FRONT = 16, NP1 = (TH + 4) * P;
  constexpr int N1 = (TH + 2) * P, NB1 = (N1 + 31) / 32, NBW1 = (NB1 + 3) / 4;  // conv1: rows -1 .. TH, every column of the pitch
  constexpr int N2 = TH * P, NB2 = (N2 + 31) / 32, NBW2 = (NB2 + 3) / 4;        // conv2: rows 0 .. TH - 1
  // slack behind the tile: the last block of 32 may end past the region, + the largest tap offset
  constexpr int END1 = P + 32 * NB1 + P + 2, END2 = 2 * P + 32 * NB2 + P + 2, ENDM = END1 > END2 ? END1 : END2;
  constexpr int SLACK = ENDM > NP1 ? ((ENDM - NP1 + 15) & ~15) : 16, NPOS = FRONT + NP1 + SLACK;
  constexpr int RB = C == 16 ? 1 : 2;                       // 16-row chunks of output channels that exist (rows 16 .. 31 of a 16-channel layer are zero weights)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_cbr[];      // [chunk][plane 4][NPOS][16 B]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int r0 = blockIdx.y * TH, c0 = blockIdx.x * TW;
  const int H = p.H, W = p.W;

  // ---- weights of a convolution -> registers: unit u = chunk * 9 + tap, lane (row li, half lh).  Requested first: their latency hides
  // behind the staging of x.
  const unsigned long long t_begin = CTICK();
  u32x4 wh[NU], wl[NU];
  auto load_w = [&](const unsigned char* Wimg, int CoPx) {
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const unsigned char* a = Wimg + ((long long)(u * 4 + lh) * CoPx + li) * 16;
      wh[u] = *reinterpret_cast<const u32x4*>(a);
      wl[u] = *reinterpret_cast<const u32x4*>(a + (long long)2 * CoPx * 16);
    }
  };
  if constexpr (C == 16) load_w(p.W1, p.CoPx1);              // (32 channels: 144 weight registers + the staging registers would spill - loaded after the staging)

  // ---- stage x: one task = 8 channels x 4 consecutive columns (16-byte loads over the 16-byte-aligned superset of the tile's columns,
  // image columns c0 - 4 .. c0 + TW + 3) -> up to four 16-byte rows of the hi and of the lo plane.  Two passes over a compile-time number of
  // tasks per thread: every global load of the tile is in flight before the first conversion waits for one
  {
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    constexpr int NQ = (TW + 8) / 4, NTASK = (C / 8) * (TH + 4) * NQ, NIT = (NTASK + 255) / 256;
    f32x4_t v[NIT][8];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int task = tid + 256 * it;
      const int gr = task / NQ, k = task - gr * NQ, g = gr / (TH + 4), row = gr - g * (TH + 4);
      const int ir = r0 - 2 + row, icq = c0 - 4 + 4 * k;
      const bool ok = task < NTASK && ir >= 0 && ir < H && icq >= 0 && icq < W;
      const float* src = p.X + (long long)(g * 8) * p.plane + (long long)ir * W + icq;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[it][j] = ok ? *reinterpret_cast<const f32x4_t*>(src + (long long)j * p.plane) : f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int task = tid + 256 * it;
      const int gr = task / NQ, k = task - gr * NQ, g = gr / (TH + 4), row = gr - g * (TH + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int col = 4 * k + e - 2;                          // column of the pitch
        u32x4 hi, lo;
#pragma unroll
        for (int j = 0; j < 4; ++j) { unsigned h_, l_; split2(v[it][2 * j][e], v[it][2 * j + 1][e], h_, l_); hi[j] = h_; lo[j] = l_; }
        if (task < NTASK && col >= 0 && col < P) {
          unsigned char* dst = smem_cbr + ((((g >> 1) * 4 + (g & 1)) * NPOS) + FRONT + row * P + col) * 16;
          *reinterpret_cast<u32x4*>(dst) = hi;
          *reinterpret_cast<u32x4*>(dst + 2 * NPOS * 16) = lo;
        }
      }
    }
  }
  if constexpr (C != 16) load_w(p.W1, p.CoPx1);
  // the slack on either side of the tile: finite values (they only ever reach outputs that are discarded)
  for (int task = tid; task < (FRONT + SLACK) * NCH * 4; task += 256) {
    const int pl = task / (FRONT + SLACK), q = task - pl * (FRONT + SLACK);
    const int pos = q < FRONT ? q : NP1 + q;
    *reinterpret_cast<u32x4*>(smem_cbr + (pl * NPOS + pos) * 16) = u32x4{0u, 0u, 0u, 0u};
  }

  __syncthreads();
  const unsigned long long t_staged = CTICK();
  CTACC(1, t_staged - t_begin);

  // this wave's blocks of 32 positions (block wave + 4 i starts at LDS position qbase + 32 (wave + 4 i)): acc[i][co][position] over the 9 taps x
  // NCH chunks.  The (chunk, tap) unit is the OUTER loop: the blocks are independent accumulation chains (a lone chain would wait out the
  // latency of every MFMA: 27 or 54 dependent ones per block), and a unit's weights are read from registers once for all of them.
  auto conv_blocks = [&](auto& acc, auto nbw_c, int nb, int qbase) {
    constexpr int NBW = decltype(nbw_c)::value;
#pragma unroll
    for (int i = 0; i < NBW; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    // straight-line: every wave runs NBW blocks (a wave with fewer repeats the last block of the region; the copy is discarded), so the
    // operand reads of the next unit can be scheduled under the MFMAs of this one
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int off = (t / 3 - 1) * P + (t % 3 - 1);
        const int u = ch * 9 + t;
        u32x4 bh[NBW], bl[NBW];
#pragma unroll
        for (int i = 0; i < NBW; ++i) {
          const unsigned char* b = smem_cbr + ((ch * 4 + lh) * NPOS + FRONT + qbase + 32 * min(wave + 4 * i, nb - 1) + li + off) * 16;
          bh[i] = *reinterpret_cast<const u32x4*>(b); bl[i] = *reinterpret_cast<const u32x4*>(b + 2 * NPOS * 16);
        }
#pragma unroll
        for (int i = 0; i < NBW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wh[u]), __builtin_bit_cast(bf16x8, bl[i]), acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NBW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wl[u]), __builtin_bit_cast(bf16x8, bh[i]), acc[i], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NBW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wh[u]), __builtin_bit_cast(bf16x8, bh[i]), acc[i], 0, 0, 0);
      }
  };
  // this lane's bias values (rows (r & 3) + 8 (r >> 2) + 4 lh), requested before the first convolution
  float bias1[8 * RB], bias2[8 * RB];
#pragma unroll
  for (int r = 0; r < 8 * RB; ++r) { const int co = (r & 3) + 8 * (r >> 2) + 4 * lh; bias1[r] = p.b1[co]; bias2[r] = p.b2[co]; }

  // ---- conv1 over rows -1 .. TH (LDS rows 1 .. TH + 2): bias, ReLU, zero outside the image; kept in registers
  f32x16 y1[NBW1];
  conv_blocks(y1, std::integral_constant<int, NBW1>{}, NB1, P);
#pragma unroll
  for (int i = 0; i < NBW1; ++i) {
    const int blk = wave + 4 * i;
    if (blk < NB1) {
      const int q = P + 32 * blk + li, row = q / P, col = q - row * P;
      const int ir = r0 - 2 + row, ic = c0 - 2 + col;
      const bool ok = ir >= 0 && ir < H && ic >= 0 && ic < W && q < P + N1;
#pragma unroll
      for (int r = 0; r < 8 * RB; ++r) y1[i][r] = ok ? fmaxf(y1[i][r] + bias1[r], 0.f) : 0.f;
    }
  }
  const unsigned long long t_c1 = CTICK();
  CTACC(2, t_c1 - t_staged);
  if constexpr (C == 16) load_w(p.W2, p.CoPx2);              // (conv1's MFMAs have read their weight registers: in order)
  __syncthreads();                                             // every wave has finished reading x
#pragma unroll
  for (int i = 0; i < NBW1; ++i) {
    const int blk = wave + 4 * i;
    if (blk < NB1) {
      const int q = P + 32 * blk + li;
#pragma unroll
      for (int g2 = 0; g2 < RB; ++g2) {
        unsigned hA[2], lA[2], hB[2], lB[2];
#pragma unroll
        for (int e2 = 0; e2 < 2; ++e2) {
          split2(y1[i][8 * g2 + 2 * e2], y1[i][8 * g2 + 2 * e2 + 1], hA[e2], lA[e2]);
          split2(y1[i][8 * g2 + 4 + 2 * e2], y1[i][8 * g2 + 5 + 2 * e2], hB[e2], lB[e2]);
        }
        u32x4 hi, lo;
#pragma unroll
        for (int e2 = 0; e2 < 2; ++e2) {
          typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
          const u32x2_t sh = __builtin_amdgcn_permlane32_swap(hA[e2], hB[e2], false, false);
          const u32x2_t sl2 = __builtin_amdgcn_permlane32_swap(lA[e2], lB[e2], false, false);
          hi[e2] = sh.x; hi[2 + e2] = sh.y; lo[e2] = sl2.x; lo[2 + e2] = sl2.y;
        }
        unsigned char* dst = smem_cbr + ((g2 * 4 + lh) * NPOS + FRONT + q) * 16;      // chunk g2, half lh, position q
        *reinterpret_cast<u32x4*>(dst) = hi;
        *reinterpret_cast<u32x4*>(dst + 2 * NPOS * 16) = lo;
      }
    }
  }
  if constexpr (C != 16) load_w(p.W2, p.CoPx2);              // (32 channels: after the intermediate has left the registers)
  __syncthreads();

  const unsigned long long t_y1 = CTICK();
  CTACC(3, t_y1 - t_c1);
  // ---- conv2 over rows 0 .. TH - 1: bias, ReLU, + x, fp32 rows
  f32x16 y2[NBW2];
  conv_blocks(y2, std::integral_constant<int, NBW2>{}, NB2, 2 * P);
  asm volatile("s_nop 0" ::: "memory");
  const unsigned long long t_c2 = CTICK();
  CTACC(4, t_c2 - t_y1);
  // epilogue through LDS: relu(acc + b2) as an fp32 tile [co][row][col] over the dead image, then 16-byte residual loads / stores
  // (the MFMA layout holds one position per lane: 4-byte accesses, 5x the instructions)
  typedef float f32x4_t __attribute__((ext_vector_type(4)));
  constexpr int NTO = C * TH * (TW / 4), NITO = NTO / 256;
  static_assert(NTO % 256 == 0, "output tasks per thread");
  f32x4_t res[NITO];
#pragma unroll
  for (int it = 0; it < NITO; ++it) {                          // residual quads requested first: they fly during the transpose
    const int task = tid + 256 * it, kq = task % (TW / 4), cr = task / (TW / 4), row = cr % TH, co = cr / TH;
    const int ir = r0 + row, ic = c0 + 4 * kq;
    res[it] = (ir < H && ic < W) ? *reinterpret_cast<const f32x4_t*>(p.X + (long long)co * p.plane + (long long)ir * W + ic) : f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
  __syncthreads();                                             // every wave's conv2 operand reads are done
  float* ot = reinterpret_cast<float*>(smem_cbr);
#pragma unroll
  for (int i = 0; i < NBW2; ++i) {
    const int blk = wave + 4 * i;
    const int q = 2 * P + 32 * blk + li, row = q / P - 2, col = q - (row + 2) * P - 2;
    if (blk < NB2 && row < TH && col >= 0 && col < TW) {
#pragma unroll
      for (int r = 0; r < 8 * RB; ++r) {
        const int co = (r & 3) + 8 * (r >> 2) + 4 * lh;
        ot[(co * TH + row) * TW + col] = fmaxf(y2[i][r] + bias2[r], 0.f);
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < NITO; ++it) {
    const int task = tid + 256 * it, kq = task % (TW / 4), cr = task / (TW / 4), row = cr % TH, co = cr / TH;
    const int ir = r0 + row, ic = c0 + 4 * kq;
    if (ir < H && ic < W) {
      const f32x4_t o = *reinterpret_cast<const f32x4_t*>(ot + (co * TH + row) * TW + 4 * kq) + res[it];
      *reinterpret_cast<f32x4_t*>(p.Y + (long long)co * p.plane + (long long)ir * W + ic) = o;
    }
  }
  const unsigned long long t_end = CTICK();
  CTACC(5, t_end - t_c2); CTACC(6, t_end - t_begin); CTACC(0, 1);
}

template <int C, int TH>
static void launch_cbr2(const Cbr2Args& a, hipStream_t s) {
  auto kern = cbr2_small_kernel<C, TH>;
  constexpr size_t lds = cbr2_lds_bytes(C, TH);
  static_assert(lds <= 160 * 1024, "LDS");
  RVC_ALLOW_BIG_LDS(kern);
  hipLaunchKernelGGL(kern, dim3((a.W + 63) / 64, (a.H + TH - 1) / TH), dim3(256), lds, s, a);
}

bool cbr2_small_eligible(const ConvLayer& c1, const ConvLayer& c2) {
  auto ok = [](const ConvLayer& L) { return L.mode == 2 && L.Wx_ != nullptr && L.ktaps == 9 && L.kh == 3 && L.kw == 3 && L.up2 == 0 && L.tconv_u == 0 && L.bd_ != nullptr; };
  return conv_x3_enabled() && ok(c1) && ok(c2) && c1.Ci == c1.Co && c2.Ci == c2.Co && c1.Co == c2.Co && (c1.Co == 16 || c1.Co == 32);
}

// out = relu(c2(relu(c1(x)))) + x for a 16- or 32-channel ConvBlockRes; x, out fp32 [C][H W] (distinct buffers)
void cbr2_small_run(const ConvLayer& c1, const ConvLayer& c2, hipStream_t s, const float* x, int H, int W, float* out) {
  RVC_REQUIRE(cbr2_small_eligible(c1, c2), "cbr2_small_run: two 3 x 3 convolutions of 16 or 32 channels with bf16x3 weight images");
  RVC_REQUIRE(x != out, "cbr2_small_run: in place is not supported");
  RVC_REQUIRE((W & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0, "cbr2_small_run: 16-byte rows (W a multiple of 4, aligned tensors)");
  Cbr2Args a{};
  a.X = x; a.Y = out; a.plane = (long long)H * W; a.H = H; a.W = W;
  a.W1 = reinterpret_cast<const unsigned char*>(c1.Wx_); a.W2 = reinterpret_cast<const unsigned char*>(c2.Wx_); a.CoPx1 = c1.CoPx; a.CoPx2 = c2.CoPx;
  a.b1 = c1.bd_; a.b2 = c2.bd_;
  if (c1.Co == 16) launch_cbr2<16, 8>(a, s); else launch_cbr2<32, 4>(a, s);
}

// ---------------------------------------------------------------------------------------------- one 3 x 3 convolution of few channels
// The blocks of those levels that change the channel count (first block of a level: 1 x 1 shortcut beside the first convolution) and RMVPE's
// 16 -> 3 output convolution, on the same structure: Y[co] = act(conv3x3(x)[co] + b[co]) [+ R[co]], CI = 16 / 32 input channels, up to 32 RB
// output rows, weights in registers, input tile + halo 1 as the image in LDS, output through an fp32 LDS tile (16-byte accesses).  Rows below
// `relu_rows` get the ReLU; rows >= `split_row` go to a second tensor Y2 (row - split_row) - so the first convolution and the 1 x 1 shortcut of a
// block (its weights at the centre tap of a second group of rows) are ONE launch reading x once.
struct Conv3SmallArgs {
  const float* __restrict__ X; long long plane; int H, W;      // [CI][H W]
  const unsigned char* Wimg; int CoPx; const float* bias;      // bf16x3 weight image (pack_x3, 9 taps), bias [Co]
  float* __restrict__ Y; float* __restrict__ Y2; int Co, split_row, relu_rows;
  const float* __restrict__ R;                                  // residual for the rows of Y (pitch plane) or null
};

template <int CI, int RB, int TH>
__global__ __launch_bounds__(256, 2) void conv3_small_kernel(const Conv3SmallArgs p) {
  constexpr int TW = 64, P = TW + 2, NCH = CI / 16, NU = NCH * 9;
  constexpr int FRONT = 16, NP1 = (TH + 2) * P;
  constexpr int N2 = TH * P, NB = (N2 + 31) / 32, NBW = (NB + 3) / 4;
  constexpr int ENDM = P + 32 * NB + P + 2;
  constexpr int SLACK = ENDM > NP1 ? ((ENDM - NP1 + 15) & ~15) : 16, NPOS = FRONT + NP1 + SLACK;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_c3[];      // [chunk][plane 4][NPOS][16 B]; later the fp32 output tile
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int r0 = blockIdx.y * TH, c0 = blockIdx.x * TW;
  const int H = p.H, W = p.W;
  typedef float f32x4_t __attribute__((ext_vector_type(4)));

  // ---- stage x: 8 channels x 4 consecutive columns per task, 16-byte loads over the aligned superset c0 - 4 .. c0 + TW + 3
  {
    constexpr int NQ = (TW + 8) / 4, NTASK = (CI / 8) * (TH + 2) * NQ, NIT = (NTASK + 255) / 256;
    f32x4_t v[NIT][8];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int task = tid + 256 * it;
      const int gr = task / NQ, k = task - gr * NQ, g = gr / (TH + 2), row = gr - g * (TH + 2);
      const int ir = r0 - 1 + row, icq = c0 - 4 + 4 * k;
      const bool ok = task < NTASK && ir >= 0 && ir < H && icq >= 0 && icq < W;
      const float* src = p.X + (long long)(g * 8) * p.plane + (long long)ir * W + icq;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[it][j] = ok ? *reinterpret_cast<const f32x4_t*>(src + (long long)j * p.plane) : f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int task = tid + 256 * it;
      const int gr = task / NQ, k = task - gr * NQ, g = gr / (TH + 2), row = gr - g * (TH + 2);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int col = 4 * k + e - 3;                          // column of the pitch: image column c0 - 1 + col
        u32x4 hi, lo;
#pragma unroll
        for (int j = 0; j < 4; ++j) { unsigned h_, l_; split2(v[it][2 * j][e], v[it][2 * j + 1][e], h_, l_); hi[j] = h_; lo[j] = l_; }
        if (task < NTASK && col >= 0 && col < P) {
          unsigned char* dst = smem_c3 + ((((g >> 1) * 4 + (g & 1)) * NPOS) + FRONT + row * P + col) * 16;
          *reinterpret_cast<u32x4*>(dst) = hi;
          *reinterpret_cast<u32x4*>(dst + 2 * NPOS * 16) = lo;
        }
      }
    }
  }
  for (int task = tid; task < (FRONT + SLACK) * NCH * 4; task += 256) {
    const int pl = task / (FRONT + SLACK), q = task - pl * (FRONT + SLACK);
    const int pos = q < FRONT ? q : NP1 + q;
    *reinterpret_cast<u32x4*>(smem_c3 + (pl * NPOS + pos) * 16) = u32x4{0u, 0u, 0u, 0u};
  }
  // ---- weights -> registers: unit u = chunk * 9 + tap, row block rb, lane (row li, half lh)
  u32x4 wh[RB][NU], wl[RB][NU];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const unsigned char* a = p.Wimg + ((long long)(u * 4 + lh) * p.CoPx + 32 * rb + li) * 16;
      wh[rb][u] = *reinterpret_cast<const u32x4*>(a);
      wl[rb][u] = *reinterpret_cast<const u32x4*>(a + (long long)2 * p.CoPx * 16);
    }
  __syncthreads();

  // ---- the convolution over rows 0 .. TH - 1 (LDS rows 1 .. TH), every column of the pitch; unit-outer over NBW x RB independent chains
  f32x16 acc[NBW][RB];
#pragma unroll
  for (int i = 0; i < NBW; ++i)
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][rb][r] = 0.f;
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int off = (t / 3 - 1) * P + (t % 3 - 1);
      const int u = ch * 9 + t;
      u32x4 bh[NBW], bl[NBW];
#pragma unroll
      for (int i = 0; i < NBW; ++i) {
        const unsigned char* b = smem_c3 + ((ch * 4 + lh) * NPOS + FRONT + P + 32 * min(wave + 4 * i, NB - 1) + li + off) * 16;
        bh[i] = *reinterpret_cast<const u32x4*>(b); bl[i] = *reinterpret_cast<const u32x4*>(b + 2 * NPOS * 16);
      }
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
        for (int i = 0; i < NBW; ++i) acc[i][rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wh[rb][u]), __builtin_bit_cast(bf16x8, bl[i]), acc[i][rb], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NBW; ++i) acc[i][rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wl[rb][u]), __builtin_bit_cast(bf16x8, bh[i]), acc[i][rb], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NBW; ++i) acc[i][rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wh[rb][u]), __builtin_bit_cast(bf16x8, bh[i]), acc[i][rb], 0, 0, 0);
      }
    }

  // ---- epilogue through an fp32 LDS tile [row][TH][TW]: bias, ReLU for the rows below relu_rows, residual, 16-byte stores
  constexpr int ROWS = 32 * RB;
  __syncthreads();                                             // every wave's operand reads are done
  float* ot = reinterpret_cast<float*>(smem_c3);
#pragma unroll
  for (int i = 0; i < NBW; ++i) {
    const int blk = wave + 4 * i;
    const int q = P + 32 * blk + li, row = q / P - 1, col = q - (row + 1) * P - 1;
    if (blk < NB && row < TH && col >= 0 && col < TW) {
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = 32 * rb + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (co < p.Co) {
            const float v = acc[i][rb][r] + p.bias[co];
            ot[(co * TH + row) * TW + col] = co < p.relu_rows ? fmaxf(v, 0.f) : v;
          }
        }
    }
  }
  __syncthreads();
  for (int task = tid; task < p.Co * TH * (TW / 4); task += 256) {
    const int kq = task % (TW / 4), cr = task / (TW / 4), row = cr % TH, co = cr / TH;
    const int ir = r0 + row, ic = c0 + 4 * kq;
    if (ir < H && ic < W) {
      f32x4_t o = *reinterpret_cast<const f32x4_t*>(ot + (co * TH + row) * TW + 4 * kq);
      const long long off = (long long)ir * W + ic;
      if (co < p.split_row) {
        if (p.R) o += *reinterpret_cast<const f32x4_t*>(p.R + (long long)co * p.plane + off);
        *reinterpret_cast<f32x4_t*>(p.Y + (long long)co * p.plane + off) = o;
      } else {
        *reinterpret_cast<f32x4_t*>(p.Y2 + (long long)(co - p.split_row) * p.plane + off) = o;
      }
    }
  }
}

constexpr size_t conv3_small_lds_bytes(int CI, int RB, int TH) {
  const int P = 66, NP1 = (TH + 2) * P, NB = (TH * P + 31) / 32, ENDM = P + 32 * NB + P + 2;
  const int SLACK = ENDM > NP1 ? ((ENDM - NP1 + 15) & ~15) : 16;
  const size_t img = (size_t)(CI / 16) * 4 * (16 + NP1 + SLACK) * 16, tile = (size_t)32 * RB * TH * 64 * 4;
  return img > tile ? img : tile;
}
template <int CI, int RB, int TH>
static void launch_conv3_small(const Conv3SmallArgs& a, hipStream_t s) {
  auto kern = conv3_small_kernel<CI, RB, TH>;
  constexpr size_t lds = conv3_small_lds_bytes(CI, RB, TH);
  static_assert(lds <= 80 * 1024, "two workgroups per CU");
  RVC_ALLOW_BIG_LDS(kern);
  hipLaunchKernelGGL(kern, dim3((a.W + 63) / 64, (a.H + TH - 1) / TH), dim3(256), lds, s, a);
}

bool conv3_small_eligible(const ConvLayer& L) {
  return conv_x3_enabled() && L.mode == 2 && L.Wx_ != nullptr && L.ktaps == 9 && L.kh == 3 && L.kw == 3 && L.up2 == 0 && L.tconv_u == 0 && L.bd_ != nullptr &&
         ((L.Ci == 16 && L.Co <= 64) || (L.Ci == 32 && L.Co <= 32)) && L.CoPx >= 64;      // (32 inputs x 64 rows of weights do not fit the registers)
}
// Y[co] = act(conv3x3(x)[co] + b[co]) [+ R[co]] for co < split_row, Y2[co - split_row] = the same without residual for the rest; ReLU on the rows below relu_rows
void conv3_small_run(const ConvLayer& L, hipStream_t s, const float* x, int H, int W, float* Y, float* Y2, int split_row, int relu_rows, const float* R) {
  RVC_REQUIRE(conv3_small_eligible(L), "conv3_small_run: a 3 x 3 convolution of 16 / 32 input and <= 64 output channels with a bf16x3 weight image");
  RVC_REQUIRE(x != Y && x != Y2 && (W & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(Y) & 15) == 0 &&
              (!Y2 || (reinterpret_cast<uintptr_t>(Y2) & 15) == 0) && (!R || (reinterpret_cast<uintptr_t>(R) & 15) == 0), "conv3_small_run: distinct, 16-byte aligned tensors, W a multiple of 4");
  RVC_REQUIRE(split_row >= 0 && split_row <= L.Co && (split_row == L.Co || Y2 != nullptr), "conv3_small_run: row split");
  Conv3SmallArgs a{};
  a.X = x; a.plane = (long long)H * W; a.H = H; a.W = W;
  a.Wimg = reinterpret_cast<const unsigned char*>(L.Wx_); a.CoPx = L.CoPx; a.bias = L.bd_;
  a.Y = Y; a.Y2 = Y2; a.Co = L.Co; a.split_row = split_row; a.relu_rows = relu_rows; a.R = R;
  const int RB = L.Co > 32 ? 2 : 1;
  if (L.Ci == 16 && RB == 1) launch_conv3_small<16, 1, 8>(a, s);
  else if (L.Ci == 16) launch_conv3_small<16, 2, 4>(a, s);
  else launch_conv3_small<32, 1, 4>(a, s);
}

}  // namespace rvc
