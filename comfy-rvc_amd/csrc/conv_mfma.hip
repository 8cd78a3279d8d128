// fp32 implicit-GEMM convolution on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32), gfx950 only.
//
// One kernel family serves every dense contraction on the RVC inference path:
//   * Conv1d (dilated / strided / grouped), ConvTranspose1d (polyphase rows), Linear (k = 1)
//   * batched A^T.B products for attention (activation tensor as the "weight" operand)
//   * Conv2d 3x3 pad 1 and ConvTranspose2d k3 s2 (2x2 phases) for the RMVPE U-Net
// Layout: activations are channel-major [C][T] (time contiguous) == the reference's NCL / NCHW.
//
// GEMM view:  Y[m][n] = sum_kk Wp[kk][m] * Xtile[kk][n],  m = output row, n = output position,
//             kk = (chunk, tap, virtual channel).  Virtual channel = (ci, phase) where phase is the
//             input position modulo the stride, so that the MFMA loop only ever sees unit-stride
//             rows in LDS (im2col-free: the input tile with its halo is staged once per chunk).
// MFMA 32x32x2 f32: A = Wp[kk][m] (lane i = m, lane half = kk parity), B = Xs[kk][n] (lane i = n);
// both are single conflict-free ds_read_b32 per operand; fp32 accumulate, bitwise an fmaf chain.
// Workgroup = 4 waves (WM x WN), each wave owns AM x AN accumulators of 32x32.
#include "conv_kernels.h"
#include <atomic>
#include <tuple>

#ifndef RVC_EPI_TWOPHASE
#define RVC_EPI_TWOPHASE 0
#endif

namespace rvc {

#ifdef RVC_CONV_TIMING
__device__ unsigned long long g_conv_timing[8];   // [0] blocks, [1] prologue, [2] stage sync+LDS fill, [3] prefetch issue, [4] MFMA loops, [5] epilogue, [6] total
#define TICK() clock64()
#define TACC(i, v) do { if (threadIdx.x == 0) atomicAdd(&g_conv_timing[i], (unsigned long long)(v)); } while (0)
#else
#define TICK() 0ll
#define TACC(i, v) do {} while (0)
#endif

constexpr int kWSlots = 32;   // register slots (floats per thread) for the prefetched weight slab (<= 32 KB / 256 threads)
// register slots for the prefetched input tile, per tile width and mode (checked against the launch geometry on the host)
__host__ __device__ constexpr int x_slots(int BN, int MODE) {
  return MODE == 2 ? (BN >= 512 ? 26 : (BN >= 256 ? 18 : 14)) : (BN >= 512 ? 36 : (BN >= 256 ? 20 : (BN >= 128 ? 32 : 16)));
}

// Software pipeline: the global loads of stage s+1 (input tile with halo + weight slab) are issued into registers before the
// MFMA loop of stage s and written to LDS after it, so HBM/L2 latency is covered by matrix work instead of a barrier wait.
// MODE 1: 1-D stride 1, MODE 2: 2-D 3x3, MODE 3: 1-D strided (phase-decomposed rows).
// In-kernel activations are the max(v, slope*v) family (identity / ReLU / leaky ReLU); GELU, tanh, sigmoid and log-clamp run as
// a separate elementwise pass (act_res_inplace) so that this kernel stays small enough for the instruction cache.
template <int WM, int WN, int AM, int AN, int MODE>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const ConvArgsX p) {
  constexpr int BM = WM * AM * 32, BN = WN * AN * 32;
  constexpr int XS = x_slots(BN, MODE);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int CK = p.CK, WROW = p.WROW;
  float* Xs = smem;
  float* Ws = smem + ((CK * WROW + 3) & ~3);

  const int tid0 = threadIdx.x, lane0 = tid0 & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane0 & 31, lh = lane0 >> 5;
  const int z = blockIdx.z / p.ksplit, ks = blockIdx.z - z * p.ksplit;   // batch index, K-split index
  const int co0 = blockIdx.y * BM;
  const int n0 = blockIdx.x * BN;

  const float* __restrict__ X = p.X + (long long)z * p.xBatch;
  const float* __restrict__ W = p.W + (long long)z * p.wBatch;

  int h0 = 0, w0 = 0;
  if (MODE == 2) { h0 = n0 / p.Wd; w0 = n0 % p.Wd; }

  int bb[AN];
#pragma unroll
  for (int an = 0; an < AN; ++an) {
    const int nl = (wn * AN + an) * 32 + li;
    bb[an] = (MODE == 2) ? (nl / p.BWd) * p.PW + (nl % p.BWd) : nl;
  }

  f32x16 acc[AM][AN];
#pragma unroll
  for (int am = 0; am < AM; ++am)
#pragma unroll
    for (int an = 0; an < AN; ++an)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[am][an][r] = 0.f;

  const int ntb = (p.ktaps + p.KT - 1) / p.KT;
  const int cps = (p.nchunk + p.ksplit - 1) / p.ksplit;                  // chunks per K split
  const int chunk0 = ks * cps, chunk1 = min(p.nchunk, chunk0 + cps);
  const int nstages = max(chunk1 - chunk0, 0) * ntb;
  const bool wvec = (p.ldW & 3) == 0 && (p.Wcols & 3) == 0 && ((((uintptr_t)W) & 15) == 0);
  const int used1 = BN + (p.ktaps - 1) * (MODE == 1 ? p.dil : 1);   // 1-D: LDS columns in use per row
  const int span = MODE == 3 ? used1 * p.stride : used1;             // 1-D: input samples staged per (channel) row
  const int nrows = MODE == 3 ? CK / p.stride : CK;                  // 1-D: rows fetched per chunk (channels)
  const int bx = MODE == 3 ? n0 * p.stride - p.pad : n0 - p.pad;
  const __amdgpu_buffer_rsrc_t wrs = make_rsrc(W, (unsigned)p.Wrows * (unsigned)p.ldW * 4u);
  const __amdgpu_buffer_rsrc_t xrs = make_rsrc(X, (unsigned)p.Ci * (unsigned)p.ldX * 4u);   // channel tail falls out of range
  const float pre_slope = p.pre_act == ACT_LRELU ? p.pre_slope : 1.f;   // input activation: leaky ReLU (slope 1 = identity)

  float xr[XS];
  float wr[kWSlots];

  // ---- input tile: global -> registers.  Slot s of wave w covers 64 columns of row (w + 4s) / ni (uniform scalar math).
  auto load_x = [&](int chunk) {
    int lane = lane0, tid = tid0;
    asm volatile("" : "+v"(lane), "+v"(tid));      // keeps the address math out of loop-invariant registers
    if (MODE != 2) {
#pragma unroll
      for (int s = 0; s < XS; ++s) {
        const unsigned g = (unsigned)(wave + 4 * s);
        const int row = p.ni == 1 ? (int)g : (int)__umulhi(g, p.magNI), qb = ((int)g - row * p.ni) * 64;   // (2^32 / 1 does not fit the magic)
        const int ci = chunk * nrows + row;
        const int e = qb + lane, x = bx + e;
        const bool ok = row < nrows && ci < p.Ci && e < span && x >= 0 && x < p.Tin;
        float v = buf_load(xrs, ok ? (unsigned)x * 4u : kOOB, (unsigned)ci * (unsigned)p.ldX * 4u);
        xr[s] = fmaxf(v, v * pre_slope);
      }
    } else {
      const int RP = (p.BH + 2) * p.PW, total = CK * RP;
#pragma unroll
      for (int s = 0; s < XS; ++s) {
        const int e = tid + 256 * s;
        const int vcc = __umulhi((unsigned)e, p.magRP), rem = e - vcc * RP;
        const int rr = __umulhi((unsigned)rem, p.magPW), cc = rem - rr * p.PW;
        const int ci = chunk * CK + vcc, hh = h0 - 1 + rr, ww = w0 - 1 + cc;
        const bool ok = e < total && hh >= 0 && hh < p.Tin && ww >= 0 && ww < p.Wd;
        xr[s] = buf_load(xrs, ok ? ((unsigned)ci * (unsigned)p.ldX + (unsigned)hh * (unsigned)p.Wd + (unsigned)ww) * 4u : kOOB);
      }
    }
  };
  // ---- input tile: registers -> LDS
  auto store_x = [&]() {
    int lane = lane0, tid = tid0;
    asm volatile("" : "+v"(lane), "+v"(tid));
    if (MODE != 2) {
#pragma unroll
      for (int s = 0; s < XS; ++s) {
        const unsigned g = (unsigned)(wave + 4 * s);
        const int row = p.ni == 1 ? (int)g : (int)__umulhi(g, p.magNI), qb = ((int)g - row * p.ni) * 64;   // (2^32 / 1 does not fit the magic)
        const int e = qb + lane;
        if (row < nrows && e < span) {
          if (MODE == 1) Xs[row * WROW + e] = xr[s];
          else { const int st = p.stride; const int q = (st == 2) ? (e >> 1) : e / st; Xs[(row * st + (e - q * st)) * WROW + q] = xr[s]; }
        }
      }
    } else {
      const int RP = (p.BH + 2) * p.PW, total = CK * RP;
#pragma unroll
      for (int s = 0; s < XS; ++s) {
        const int e = tid + 256 * s;
        if (e < total) { const int vcc = __umulhi((unsigned)e, p.magRP); Xs[vcc * WROW + (e - vcc * RP)] = xr[s]; }
      }
    }
  };
  // ---- weight slab of (chunk, tap block): global -> registers -> LDS.  Rows past the slab / matrix read as zero; columns
  //      past Wcols only feed output rows that are never stored.
  auto load_w = [&](int chunk, int tb) {
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    const int ut = min(p.KT, p.ktaps - tb * p.KT);
    const int rows = ut * CK;
    const unsigned row0 = (unsigned)((chunk * p.ktaps + tb * p.KT) * CK);
    if (wvec) {
      constexpr int V4 = BM / 4;
#pragma unroll
      for (int s = 0; s < kWSlots / 4; ++s) {
        const int e = tid + 256 * s;
        const int rr = e / V4, c4 = (e - rr * V4) * 4;
        const unsigned off = rr < rows ? ((row0 + rr) * (unsigned)p.ldW + (unsigned)(co0 + c4)) * 4u : kOOB;
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrs, (int)off, 0, 0);
        wr[4 * s] = __uint_as_float(v.x); wr[4 * s + 1] = __uint_as_float(v.y); wr[4 * s + 2] = __uint_as_float(v.z); wr[4 * s + 3] = __uint_as_float(v.w);
      }
    } else {
#pragma unroll
      for (int s = 0; s < kWSlots; ++s) {
        const int e = tid + 256 * s;
        const int rr = e / BM, c = e - rr * BM;
        wr[s] = buf_load(wrs, (rr < rows && co0 + c < p.Wcols) ? ((row0 + rr) * (unsigned)p.ldW + (unsigned)(co0 + c)) * 4u : kOOB);
      }
    }
  };
  auto store_w = [&](int tb) {
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    const int ut = min(p.KT, p.ktaps - tb * p.KT);
    const int rows = ut * CK;
    if (wvec) {
      constexpr int V4 = BM / 4;
#pragma unroll
      for (int s = 0; s < kWSlots / 4; ++s) {
        const int e = tid + 256 * s;
        const int rr = e / V4, c4 = (e - rr * V4) * 4;
        if (rr < rows) *reinterpret_cast<float4*>(Ws + rr * BM + c4) = make_float4(wr[4 * s], wr[4 * s + 1], wr[4 * s + 2], wr[4 * s + 3]);
      }
    } else {
#pragma unroll
      for (int s = 0; s < kWSlots; ++s) {
        const int e = tid + 256 * s;
        const int rr = e / BM, c = e - rr * BM;
        if (rr < rows) Ws[rr * BM + c] = wr[s];
      }
    }
  };

  // Iteration `it` moves stage it-1 from registers to LDS, prefetches stage `it` into registers and runs the MFMAs of stage
  // it-1 (one copy of every code section; the prefetch latency hides under the matrix work).
  const long long t_begin = TICK();
  int chunk = chunk0, tb = 0, pchunk = chunk0, ptb = 0;
  for (int it = 0; it <= nstages; ++it) {
    const long long ta = TICK();
    if (it > 0) {
      __syncthreads();                       // every wave is done reading the previous stage from LDS
      if (ptb == 0) store_x();
      store_w(ptb);
      __syncthreads();
    }
    const long long tb_ = TICK();
    TACC(it <= 1 ? 1 : 2, tb_ - ta);
    if (it < nstages) {
      if (tb == 0) load_x(chunk);
      load_w(chunk, tb);
    }
    const long long tc = TICK();
    TACC(3, tc - tb_);
    if (it > 0) {
      const int ut = min(p.KT, p.ktaps - ptb * p.KT);
      for (int uu = 0; uu < ut; ++uu) {
        const int u = ptb * p.KT + uu;
        int toff;
        if (MODE == 2) toff = (u / 3) * p.PW + (u % 3);
        else toff = u * (MODE == 1 ? p.dil : 1);
        const float* wrow = Ws + (uu * CK + lh) * BM + wm * AM * 32 + li;
        const float* xrow = Xs + lh * WROW + toff;
        for (int m = 0; m < CK / 2; ++m) {
          float a[AM], b[AN];
#pragma unroll
          for (int am = 0; am < AM; ++am) a[am] = wrow[2 * m * BM + am * 32];
#pragma unroll
          for (int an = 0; an < AN; ++an) b[an] = xrow[2 * m * WROW + bb[an]];
#pragma unroll
          for (int am = 0; am < AM; ++am)
#pragma unroll
            for (int an = 0; an < AN; ++an)
              acc[am][an] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[am], b[an], acc[am][an], 0, 0, 0);
        }
      }
    }
    TACC(4, TICK() - tc);
    pchunk = chunk; ptb = tb;
    if (++tb == ntb) { tb = 0; ++chunk; }
  }
  (void)pchunk;
  const long long t_epi = TICK();

  // -------------------------------------------------------------------------- epilogue
  // v = acc + bias; act in {identity, ReLU, leaky ReLU} as max(v, slope * v), before or after the residual; * out_scale;
  // (+ previous output).  Row r+1's residual / accumulate operands are requested before row r is stored.
  const float* __restrict__ bias = p.bias ? p.bias + (long long)z * p.bBatch : nullptr;
  float* Y = p.Y + (long long)z * p.yBatch;
  const float lslope = p.act == ACT_NONE ? 1.f : (p.act == ACT_RELU ? 0.f : p.act_slope);
  const float oscale = p.out_scale;
  if (p.ksplit > 1) {
    // split-K: raw partial sums; bias / activation / residual are applied by splitk_reduce_kernel in a fixed order
    float* P = p.partial + ((long long)blockIdx.z * p.Co) * p.ldP;
    const __amdgpu_buffer_rsrc_t prs = make_rsrc(P, (unsigned)p.Co * (unsigned)p.ldP * 4u);
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
  } else if (p.ostride == 1 && !p.up2) {
    dense_epilogue<WM, WN, AM, AN>(p, acc, z, co0, n0, wm, wn, li, lh);
  } else {
    // interleaved stores (ConvTranspose1d phases, ConvTranspose2d 2x2 phases): generic 64-bit indexing, accumulate optional
#pragma unroll
    for (int am = 0; am < AM; ++am) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = co0 + (wm * AM + am) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m >= p.Co) continue;
        const int co = MODE == 2 ? m % p.orows : m / p.ostride, ph = MODE == 2 ? m / p.orows : m - co * p.ostride;
        const float bv = bias ? bias[co] : 0.f;
#pragma unroll
        for (int an = 0; an < AN; ++an) {
          const int n = n0 + (wn * AN + an) * 32 + li;
          if (n >= p.Tout) continue;
          long long oidx;
          if (MODE == 2) {
            const int hh = n / p.Wd, ww = n - hh * p.Wd;
            oidx = (long long)co * p.ldY + (long long)(2 * hh + (ph >> 1)) * (2 * p.Wd) + 2 * ww + (ph & 1);
          } else {
            const long long to = (long long)n * p.ostride + ph;
            if (to >= p.ldY) continue;   // ldY doubles as the true output length for interleaved stores
            oidx = (long long)co * p.ldY + to;
          }
          float v = acc[am][an][r] + bv;
          v = fmaxf(v, v * lslope) * oscale;
          if (p.accumulate) v += Y[oidx];
          Y[oidx] = v;
        }
      }
    }
  }
  TACC(5, TICK() - t_epi); TACC(6, TICK() - t_begin); TACC(0, 1);
}

// Elementwise pass for the activations kept out of the MFMA kernel: y = act(y) [+ r]  or  y = act(y + r)
__device__ __forceinline__ float apply_act(float v, int act, float slope) {
  switch (act) {
    case ACT_LRELU: return v > 0.f ? v : v * slope;
    case ACT_RELU: return v > 0.f ? v : 0.f;
    case ACT_GELU: return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
    case ACT_TANH: return tanhf(v);
    case ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    case ACT_LOGCLAMP: return logf(fmaxf(v, slope));
    default: return v;
  }
}
__global__ void act_res_kernel(float* __restrict__ y, const float* __restrict__ r, int rows, int T, long long ldY, long long ldR, int act,
                               float slope, int act_before_res) {
  const long long n = (long long)rows * T;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long st = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += st) {
    const int c = (int)(i / T); const int t = (int)(i - (long long)c * T);
    float v = y[(long long)c * ldY + t];
    const float rv = r ? r[(long long)c * ldR + t] : 0.f;
    v = act_before_res ? apply_act(v, act, slope) + rv : apply_act(v + rv, act, slope);
    y[(long long)c * ldY + t] = v;
  }
}
static void act_res_inplace(hipStream_t s, float* y, const float* r, int rows, int T, long long ldY, long long ldR, int act, float slope,
                            int act_before_res) {
  const long long n = (long long)rows * T;
  int blocks = (int)((n + 255) / 256); if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(act_res_kernel, dim3(blocks), dim3(256), 0, s, y, r, rows, T, ldY, ldR, act, slope, act_before_res);
}

// Second pass of a split-K launch: sums the partials in a fixed order and applies the dense epilogue.
__global__ void splitk_reduce_kernel(const float* __restrict__ P, int S, int batch, int M, int N, long long ldP, const float* __restrict__ bias, int bBatch,
                                     const float* __restrict__ R, long long ldR, long long rBatch, float* __restrict__ Y, long long ldY, long long yBatch,
                                     int orows, float lslope, int abr, float oscale, int accumulate) {
  const long long total = (long long)batch * M * N;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long st = (long long)gridDim.x * blockDim.x;
  for (; i < total; i += st) {
    const int n = (int)(i % N); const long long t = i / N; const int m = (int)(t % M); const int z = (int)(t / M);
    float v = 0.f;
    for (int k = 0; k < S; ++k) v += P[(((long long)z * S + k) * M + m) * ldP + n];
    const int co = m % orows;
    if (bias) v += bias[(long long)z * bBatch + co];
    const float rv = R ? R[(long long)z * rBatch + (long long)co * ldR + n] : 0.f;
    if (abr) v = fmaxf(v, v * lslope) + rv; else { v += rv; v = fmaxf(v, v * lslope); }
    const long long o = (long long)z * yBatch + (long long)co * ldY + n;
    v *= oscale;
    if (accumulate) v += Y[o];
    Y[o] = v;
  }
}

void splitk_reduce_launch(const ConvArgsX& a, int S, int batch, hipStream_t s) {
  const long long total = (long long)batch * a.Co * a.Tout;
  int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096;
  const float lslope = a.act == ACT_NONE ? 1.f : (a.act == ACT_RELU ? 0.f : a.act_slope);
  { ProfKernelEvents& pe = prof_kernel_events(); if (pe.armed) ++pe.launches; }      // (a second launch of an open profiling bracket: the bracket is timed as a whole)
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, a.partial, S, batch, a.Co, a.Tout, a.ldP, a.bias, a.bBatch, a.R, a.ldR,
                     a.rBatch, a.Y, a.ldY, a.yBatch, a.orows, lslope, a.act_before_res, a.out_scale, a.accumulate);
}

// ============================================================================ host side
#ifdef RVC_CONV_TIMING
void conv_x3_timing_read(unsigned long long* out8, bool reset);
void attention_timing_read(unsigned long long* out8, bool reset);
void conv_x3p_timing_read(unsigned long long* out8, bool reset);
void conv_x3q_timing_read(unsigned long long* out8, bool reset);
void conv_rbh_timing_read(unsigned long long* out8, bool reset);
void conv_rb3_timing_read(unsigned long long* out8, bool reset);
void conv_x3s_timing_read(unsigned long long* out8, bool reset);
void cbr2_timing_read(unsigned long long* out8, bool reset);
void attention_dma_timing_read(unsigned long long* out8, bool reset);
void conv_timing_read(unsigned long long* out8, bool reset) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_conv_timing), sizeof(unsigned long long) * 8);
  if (reset) { unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_conv_timing), z, sizeof(z)); }
  unsigned long long x3[8]; conv_x3_timing_read(x3, reset);
  for (int i = 0; i < 8; ++i) out8[i] += x3[i];
  attention_timing_read(x3, reset);
  for (int i = 0; i < 8; ++i) out8[i] += x3[i];
  attention_dma_timing_read(x3, reset); // (attention on images: [0] workgroups, [1] prologue, [2] tile loop, [3] slab store + ticket, [4] merge, [5] epilogue, [6] total)
  for (int i = 0; i < 8; ++i) out8[i] += x3[i];
  cbr2_timing_read(x3, reset);          // (fused ConvBlockRes: [0] workgroups, [1] staging, [2] conv1, [3] y1 -> LDS, [4] conv2, [5] epilogue, [6] total)
  for (int i = 0; i < 8; ++i) out8[i] += x3[i];
  conv_x3p_timing_read(x3, reset);      // (pipelined kernel: [0] tiles, [1] prologue, [2] compute, [3] weight wait, [4] barrier, [5] epilogue, [6] total)
  for (int i = 0; i < 8; ++i) out8[i] += x3[i];
  conv_x3q_timing_read(x3, reset);      // (persistent kernel: the same slots; [1] once per workgroup, [6] per workgroup)
  for (int i = 0; i < 8; ++i) out8[i] += x3[i];
  conv_rbh_timing_read(x3, reset);      // (persistent fused pair, LDS-resident weights: [0] tiles, [1] stage, [2] conv1, [3] h + requests, [4] conv2, [5] epilogue, [6] total per workgroup)
  for (int i = 0; i < 8; ++i) out8[i] += x3[i];
  conv_rb3_timing_read(x3, reset);      // (whole ResBlock per launch: [0] tiles, [1] x image + requests, [2] first convolutions, [3] images, [4] second convolutions, [5] epilogue, [6] total per workgroup, [7] barrier waits)
  for (int i = 0; i < 8; ++i) out8[i] += x3[i];
  conv_x3s_timing_read(x3, reset);      // (split-resident GEMM: [0] workgroups, [1] prologue, [2] reads + MFMA issue, [3] DMA wait, [4] barrier, [5] split-K + epilogue, [6] total)
  for (int i = 0; i < 8; ++i) out8[i] += x3[i];
}
#else
void conv_timing_read(unsigned long long* out8, bool) { for (int i = 0; i < 8; ++i) out8[i] = 0; }
#endif
float* dev_upload(const float* host, size_t n) {
  float* d = nullptr;
  RVC_HIP_CHECK(hipMalloc(&d, (n ? n : 1) * sizeof(float)));
  if (n) RVC_HIP_CHECK(hipMemcpy(d, host, n * sizeof(float), hipMemcpyHostToDevice));
  return d;
}
void dev_free(void* p) { if (p) (void)hipFree(p); }

// Per-stream scratch buffers (slot 0: split-K partials, slot 1/2: pipeline temporaries).  Work on one stream is ordered, so a
// buffer can be reused by consecutive launches; growing (warm-up only) synchronises, steady state never allocates.
// The key holds the device: the null / legacy stream handle is the same value on every device.  Only the registry is locked; a
// buffer belongs to one stream, whose work is enqueued by one host thread at a time, so growing it synchronises that stream only and
// does not stall the other lanes.
namespace {
struct ScratchBuf { void* p = nullptr; size_t cap = 0; };
using ScratchKey = std::tuple<int, hipStream_t, int>;
std::map<ScratchKey, ScratchBuf>& scratch_pool() { static std::map<ScratchKey, ScratchBuf> pool; return pool; }
std::mutex& scratch_mu() { static std::mutex mu; return mu; }
}  // namespace
void* stream_scratch(hipStream_t s, int slot, size_t bytes) {
  int dev = 0;
  RVC_HIP_CHECK(hipGetDevice(&dev));
  ScratchBuf* b;
  { std::lock_guard<std::mutex> lk(scratch_mu()); b = &scratch_pool()[ScratchKey{dev, s, slot}]; }   // (std::map nodes are address-stable)
  if (bytes > b->cap) {
    RVC_HIP_CHECK(hipStreamSynchronize(s));
    if (b->p) (void)hipFree(b->p);
    b->p = nullptr; b->cap = 0;
    const size_t want = bytes + bytes / 4 + (1 << 20);
    RVC_HIP_CHECK(hipMalloc(&b->p, want));
    b->cap = want;
  }
  return b->p;
}
// Same pool; the buffer is zero-filled (on `s`, ahead of whatever the caller enqueues next) whenever it is (re)allocated, never otherwise:
// for words whose users leave them zero (the K-split tickets of conv_x3s_kernel).
void* stream_scratch_zeroed(hipStream_t s, int slot, size_t bytes, bool* fresh) {
  int dev = 0;
  RVC_HIP_CHECK(hipGetDevice(&dev));
  ScratchBuf* b;
  { std::lock_guard<std::mutex> lk(scratch_mu()); b = &scratch_pool()[ScratchKey{dev, s, slot}]; }
  if (fresh) *fresh = false;
  if (bytes > b->cap) {
    RVC_HIP_CHECK(hipStreamSynchronize(s));
    if (b->p) (void)hipFree(b->p);
    b->p = nullptr; b->cap = 0;
    RVC_HIP_CHECK(hipMalloc(&b->p, bytes));
    b->cap = bytes;
    RVC_HIP_CHECK(hipMemsetAsync(b->p, 0, bytes, s));
    if (fresh) *fresh = true;
  }
  return b->p;
}
void stream_scratch_release(int device) {
  std::lock_guard<std::mutex> lk(scratch_mu());
  auto& pool = scratch_pool();
  for (auto it = pool.begin(); it != pool.end();) {
    if (std::get<0>(it->first) == device) { if (it->second.p) (void)hipFree(it->second.p); it = pool.erase(it); } else ++it;
  }
}

void Arena::ensure(size_t bytes) {
  if (bytes <= cap) return;
  if (base) { RVC_HIP_CHECK(hipDeviceSynchronize()); (void)hipFree(base); base = nullptr; cap = 0; }
  size_t want = bytes + bytes / 8 + (1 << 20);
  RVC_HIP_CHECK(hipMalloc(&base, want));
  cap = want; ++gen;
}
void Arena::release() { if (base) (void)hipFree(base); base = nullptr; cap = 0; ++gen; }

static int pick_ck(int V, int ktaps, int stride) {
  // k = 1 (GEMM): 64 channels per stage so that one stage of MFMA work outlasts the latency of the next stage's prefetch
  int ck = ktaps >= 4 ? 8 : (ktaps >= 2 ? 16 : 64);
  if (stride > 1) { ck = ((ck + stride - 1) / stride) * stride; if (ck & 1) ck *= 2; }
  if (V < ck) { ck = stride > 1 ? ((V + stride - 1) / stride) * stride : V; if (ck & 1) ck += (stride > 1 ? stride : 1); if (ck < 2) ck = 2; }
  return ck;
}

static void upload_layer(ConvLayer& L, const std::vector<float>& packed, const float* bias, int nbias) {
  L.Wd_ = dev_upload(packed.data(), packed.size());
  L.bd_ = bias ? dev_upload(bias, nbias) : nullptr;
}

void conv_layer_free(ConvLayer& L) { dev_free(L.Wd_); dev_free(L.bd_); dev_free(L.Wx_); dev_free(L.Wh_); dev_free(L.bd4_); L.Wd_ = L.bd_ = L.bd4_ = nullptr; L.Wx_ = L.Wh_ = nullptr; }

// bf16x3 weight image (conv_x3.hip): every fp32 weight is split w = hi + lo (both bf16, round-to-nearest-even) and stored
// [16-channel chunk][tap][hi|lo][8-channel half][CoPx rows][8 channels]: 16-B rows per half-plane, which is the layout the kernel
// keeps in LDS (conflict-free ds_read_b128 at any row offset without a swizzle).
static uint16_t bf16_rne(float f) {
  uint32_t u; memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static float bf16_to_f32(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static thread_local bool g_x3_default = false;
static thread_local int g_precision = 1;      // 0: fp32 kernel only, 1: layers initialised under conv_x3_set_default(true), 2: every eligible layer
bool conv_x3_set_default(bool on) { const bool prev = g_x3_default; g_x3_default = on; return prev; }
int conv_set_precision(int mode) { const int prev = g_precision; if (mode >= 0) g_precision = mode; return prev; }
// grouped layers (HuBERT's positional convolution: 16 groups of 48 -> 48 channels, k = 128): one image per group, rows padded to 64 (the
// split-resident kernel's 64-row tile, conv_x3s.hip) - [group][chunk][tap][hi | lo][half][CoPx rows][8 ch]
static void pack_x3_grouped(ConvLayer& L, const float* w /*[groups * Cog][Cig][k]*/, int groups, int Cog, int Cig, int k) {
  L.CoPx = (Cog + 63) & ~63;
  const int nch = Cig / 16;
  L.wxBatch = (long long)nch * k * 2 * L.CoPx * 16;        // uint16 elements per group
  std::vector<uint16_t> P((size_t)groups * L.wxBatch, 0);
  for (int g = 0; g < groups; ++g)
    for (int co = 0; co < Cog; ++co)
      for (int ci = 0; ci < Cig; ++ci)
        for (int u = 0; u < k; ++u) {
          const float v = w[((size_t)(g * Cog + co) * Cig + ci) * k + u];
          const uint16_t hi = bf16_rne(v), lo = bf16_rne(v - bf16_to_f32(hi));
          const int chunk = ci >> 4, c16 = ci & 15;
          const size_t base = (((size_t)chunk * k + u) * 2 * 2 + (size_t)(c16 >> 3)) * L.CoPx + co;
          uint16_t* Q = P.data() + (size_t)g * L.wxBatch;
          Q[base * 8 + (c16 & 7)] = hi;
          Q[(base + 2 * (size_t)L.CoPx) * 8 + (c16 & 7)] = lo;
        }
  RVC_HIP_CHECK(hipMalloc(&L.Wx_, P.size() * sizeof(uint16_t)));
  RVC_HIP_CHECK(hipMemcpy(L.Wx_, P.data(), P.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
}
static void pack_x3(ConvLayer& L, const float* w, int Co, int Ci, int k) {
  L.CoPx = (Co + 127) & ~127;
  const int nch = Ci / 16;
  L.wxBatch = (long long)nch * k * 2 * L.CoPx * 16;
  std::vector<uint16_t> P((size_t)L.wxBatch, 0);
  for (int co = 0; co < Co; ++co)
    for (int ci = 0; ci < Ci; ++ci)
      for (int u = 0; u < k; ++u) {
        const float v = w[((size_t)co * Ci + ci) * k + u];
        const uint16_t hi = bf16_rne(v), lo = bf16_rne(v - bf16_to_f32(hi));
        const int chunk = ci >> 4, c16 = ci & 15;
        const size_t base = (((size_t)chunk * k + u) * 2 * 2 + (size_t)(c16 >> 3)) * L.CoPx + co;   // (chunk, tap, hi, half) plane, row co
        P[base * 8 + (c16 & 7)] = hi;
        P[(base + 2 * (size_t)L.CoPx) * 8 + (c16 & 7)] = lo;
      }
  RVC_HIP_CHECK(hipMalloc(&L.Wx_, P.size() * sizeof(uint16_t)));
  RVC_HIP_CHECK(hipMemcpy(L.Wx_, P.data(), P.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
}

// fp16x2 (H2) image of a ResBlock-pair layer (conv_x3q.hip): ONE fp16 term per weight, round-to-nearest-even, [chunk][tap][half][CoPx rows][8 ch].
// fp16 keeps 11 significant bits down to 2^-14 and loses them gradually below (subnormals down to 2^-24): a layer whose weights are all tiny, or
// that holds a weight beyond fp16's range, keeps the bf16x3 arithmetic (no image, conv1d_pair_h2_eligible says no).
static uint16_t f16_rne(float f) {
  const _Float16 h = (_Float16)f;                       // host conversion: round-to-nearest-even, subnormals kept
  uint16_t u; memcpy(&u, &h, 2); return u;
}
static void pack_h2(ConvLayer& L, const float* w, int Co, int Ci, int k) {
  float amax = 0.f;
  for (size_t i = 0; i < (size_t)Co * Ci * k; ++i) amax = std::fmax(amax, std::fabs(w[i]));
  if (!(amax < 60000.f) || amax < 0x1p-10f) return;
  RVC_REQUIRE(L.CoPx == ((Co + 127) & ~127), "pack_h2 after pack_x3");
  const int nch = Ci / 16;
  std::vector<uint16_t> P((size_t)nch * k * 2 * L.CoPx * 8, 0);
  for (int co = 0; co < Co; ++co)
    for (int ci = 0; ci < Ci; ++ci)
      for (int u = 0; u < k; ++u) {
        const int chunk = ci >> 4, c16 = ci & 15;
        const size_t base = (((size_t)chunk * k + u) * 2 + (size_t)(c16 >> 3)) * L.CoPx + co;   // (chunk, tap, half) plane, row co
        P[base * 8 + (c16 & 7)] = f16_rne(w[((size_t)co * Ci + ci) * k + u]);
      }
  RVC_HIP_CHECK(hipMalloc(&L.Wh_, P.size() * sizeof(uint16_t)));
  RVC_HIP_CHECK(hipMemcpy(L.Wh_, P.data(), P.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
}
static std::atomic<int> g_pair_h2{-1};
int conv_set_pair_arithmetic(int mode) {
  int cur = g_pair_h2.load(std::memory_order_relaxed);
  if (cur < 0) { cur = knob_int("RVC_H2", 1) ? 1 : 0; g_pair_h2.store(cur, std::memory_order_relaxed); }
  if (mode >= 0) g_pair_h2.store(mode ? 1 : 0, std::memory_order_relaxed);
  return cur;
}

void conv_layer_append_x3(ConvLayer& dst, const ConvLayer& extra) {
  RVC_REQUIRE(dst.Wx_ && extra.Wx_ && dst.CoPx == extra.CoPx && dst.Co == extra.Co && dst.groups == 1 && extra.groups == 1 && extra.mode == 1 && extra.k == 1 &&
              (extra.Ci & 15) == 0 && dst.seg2_chunks == 0 && !dst.bd_ == !extra.bd_, "conv_layer_append_x3: layers do not combine");
  RVC_REQUIRE(!dst.bd_, "conv_layer_append_x3: bias-free layers only");
  uint16_t* W2 = nullptr;
  RVC_HIP_CHECK(hipMalloc(&W2, (size_t)(dst.wxBatch + extra.wxBatch) * sizeof(uint16_t)));
  RVC_HIP_CHECK(hipMemcpy(W2, dst.Wx_, (size_t)dst.wxBatch * sizeof(uint16_t), hipMemcpyDeviceToDevice));
  RVC_HIP_CHECK(hipMemcpy(W2 + dst.wxBatch, extra.Wx_, (size_t)extra.wxBatch * sizeof(uint16_t), hipMemcpyDeviceToDevice));
  dev_free(dst.Wx_);
  dst.Wx_ = W2; dst.seg2_chunks = extra.Ci / 16;
}

void conv1d_layer_init(ConvLayer& L, const float* w, const float* bias, int Co, int Ci, int k, int stride, int pad,
                       int dil, int groups) {
  RVC_REQUIRE(Co % groups == 0 && Ci % groups == 0, "groups must divide channels");
  RVC_REQUIRE(stride == 1 || dil == 1, "strided convs must have dilation 1");
  const int Cog = Co / groups, Cig = Ci / groups;
  L.mode = 1; L.groups = groups; L.Ci = Cig; L.Co = Cog; L.co_real = Cog; L.CoP = (Cog + 31) & ~31;
  L.k = k; L.stride = stride; L.dil = dil; L.pad = pad; L.tconv_u = 0; L.up2 = 0;
  L.ktaps = (k + stride - 1) / stride;
  const int V = Cig * stride;
  L.CK = pick_ck(V, L.ktaps, stride);
  L.nchunk = (V + L.CK - 1) / L.CK;
  L.wBatch = (long long)L.nchunk * L.ktaps * L.CK * L.CoP;
  std::vector<float> P((size_t)groups * L.wBatch, 0.f);
  for (int g = 0; g < groups; ++g)
    for (int co = 0; co < Cog; ++co)
      for (int ci = 0; ci < Cig; ++ci)
        for (int tap = 0; tap < k; ++tap) {
          const int u = tap / stride, r = tap % stride;
          const int vc = ci * stride + r;
          const int chunk = vc / L.CK, vcc = vc % L.CK;
          P[(size_t)g * L.wBatch + (((size_t)chunk * L.ktaps + u) * L.CK + vcc) * L.CoP + co] =
              w[((size_t)(g * Cog + co) * Cig + ci) * k + tap];
        }
  upload_layer(L, P, bias, Co);
  if ((g_precision == 2 || (g_precision == 1 && g_x3_default)) && groups == 1 && Ci % 16 == 0 && Co >= 32) {
    pack_x3(L, w, Co, Ci, k);
    // a candidate for the persistent ResBlock kernel: square, stride 1, 3 / 7 / 11 taps, "same" padding, whole 64-row tiles, an even number of chunks
    // ... or the 32-channel stage's fused pair with LDS-resident weights (conv_rbh.hip)
    if (stride == 1 && Ci == Co && (k == 3 || k == 7 || k == 11) && ((Co % 64 == 0 && Ci >= 64) || Co == 32) && pad == dil * (k - 1) / 2) pack_h2(L, w, Co, Ci, k);
  }
  if ((g_precision == 2 || (g_precision == 1 && g_x3_default)) && groups > 1 && stride == 1 && Cig % 16 == 0 && Cog % 16 == 0 && dil >= 1)
    pack_x3_grouped(L, w, groups, Cog, Cig, k);            // only conv_x3s_run reads it (the tiled bf16x3 kernels refuse groups > 1)
}

// ConvTranspose1d as u polyphase stride-1 convolutions: GEMM row co * u + r = output channel co, phase r, i.e. the PHASE is the fastest
// row index.  One tile then holds every phase of its channels, so the interleaved stores of a workgroup fill whole cache lines between
// them (with phase-major rows the u partial writes of a line came from u different workgroups / L2s: 9x the algorithmic HBM traffic
// on the 10x up-sampler, rocprofv3 FETCH / WRITE_SIZE).
void tconv1d_layer_init(ConvLayer& L, const float* w, const float* bias, int Ci, int Co, int k, int u, int pad) {
  const int E2 = (k - 1 - pad) / u, E1 = (u - 1 + pad) / u;
  L.mode = 1; L.groups = 1; L.Ci = Ci; L.co_real = Co; L.Co = u * Co; L.CoP = (L.Co + 31) & ~31;
  L.k = E1 + E2 + 1; L.stride = 1; L.dil = 1; L.pad = E2; L.tconv_u = u; L.up2 = 0;
  L.ktaps = L.k;
  L.CK = pick_ck(Ci, L.ktaps, 1);
  L.nchunk = (Ci + L.CK - 1) / L.CK;
  L.wBatch = (long long)L.nchunk * L.ktaps * L.CK * L.CoP;
  std::vector<float> P((size_t)L.wBatch, 0.f);
  for (int r = 0; r < u; ++r)
    for (int co = 0; co < Co; ++co)
      for (int ci = 0; ci < Ci; ++ci)
        for (int j = 0; j < L.ktaps; ++j) {
          const int e = E2 - j;
          const int kk = e * u + r + pad;
          if (kk < 0 || kk >= k) continue;
          const int chunk = ci / L.CK, vcc = ci % L.CK;
          P[(((size_t)chunk * L.ktaps + j) * L.CK + vcc) * L.CoP + (size_t)co * u + r] = w[((size_t)ci * Co + co) * k + kk];
        }
  if ((g_precision == 2 || (g_precision == 1 && g_x3_default)) && Ci % 16 == 0 && L.Co >= 32) {
    // bf16x3 image of the equivalent stride-1 convolution: Weq[co * u + r][ci][j]
    std::vector<float> Weq((size_t)L.Co * Ci * L.ktaps, 0.f);
    for (int r = 0; r < u; ++r)
      for (int co = 0; co < Co; ++co)
        for (int ci = 0; ci < Ci; ++ci)
          for (int j = 0; j < L.ktaps; ++j) {
            const int kk = (E2 - j) * u + r + pad;
            if (kk >= 0 && kk < k) Weq[(((size_t)co * u + r) * Ci + ci) * L.ktaps + j] = w[((size_t)ci * Co + co) * k + kk];
          }
    pack_x3(L, Weq.data(), L.Co, Ci, L.ktaps);
  }
  // remember the true transposed-conv geometry for the output length
  L.k = k; L.pad = pad;
  L.dil = 1;
  L.conv_pad = E2;
  upload_layer(L, P, bias, Co);
}

void conv2d_kx_layer_init(ConvLayer& L, const float* w, const float* bias, int Co, int Ci, int KH, int KW, int PH, int PWL) {
  RVC_REQUIRE(Ci % 16 == 0 && KH >= 1 && KW >= 1 && PH >= 0 && PH < KH && PWL >= 0 && PWL < KW, "conv2d_kx: Ci must be a multiple of 16, padding inside the window");
  L.mode = 2; L.groups = 1; L.Ci = Ci; L.Co = Co; L.co_real = Co; L.CoP = (Co + 31) & ~31;
  L.k = KH * KW; L.stride = 1; L.dil = 1; L.pad = 0; L.tconv_u = 0; L.up2 = 0; L.ktaps = KH * KW;
  L.kh = KH; L.kw = KW; L.ph = PH; L.pwl = PWL;
  L.CK = 16; L.nchunk = Ci / 16; L.wBatch = 0;
  pack_x3(L, w, Co, Ci, KH * KW);                          // [Co][Ci][KH * KW] is the tap order dh * KW + dw the kernel walks
  L.bd_ = bias ? dev_upload(bias, Co) : nullptr;
}
void conv2d3x3_layer_init(ConvLayer& L, const float* w, const float* bias, int Co, int Ci) {
  L.mode = 2; L.groups = 1; L.Ci = Ci; L.Co = Co; L.co_real = Co; L.CoP = (Co + 31) & ~31;
  L.k = 3; L.stride = 1; L.dil = 1; L.pad = 1; L.tconv_u = 0; L.up2 = 0; L.ktaps = 9;
  L.CK = pick_ck(Ci, 9, 1);
  L.nchunk = (Ci + L.CK - 1) / L.CK;
  L.wBatch = (long long)L.nchunk * 9 * L.CK * L.CoP;
  std::vector<float> P((size_t)L.wBatch, 0.f);
  for (int co = 0; co < Co; ++co)
    for (int ci = 0; ci < Ci; ++ci)
      for (int u = 0; u < 9; ++u) {
        const int chunk = ci / L.CK, vcc = ci % L.CK;
        P[(((size_t)chunk * 9 + u) * L.CK + vcc) * L.CoP + co] = w[((size_t)co * Ci + ci) * 9 + u];
      }
  upload_layer(L, P, bias, Co);
  if ((g_precision == 2 || (g_precision == 1 && g_x3_default)) && Ci % 16 == 0) pack_x3(L, w, Co, Ci, 9);
}

void conv2d1x1_layer_init(ConvLayer& L, const float* w, const float* bias, int Co, int Ci) {
  conv1d_layer_init(L, w, bias, Co, Ci, 1, 1, 0, 1, 1);
}

void tconv2d_layer_init(ConvLayer& L, const float* w, const float* bias, int Ci, int Co) {
  // ConvTranspose2d(k=3, stride=2, padding=1, output_padding=1): out[2h+a][2w+b] uses x[h+dh][w+dw], dh,dw in {0,1},
  // with kernel index kh = a + 1 - 2*dh (valid iff 0 <= kh <= 2).  Stated as a 3x3 conv whose taps (dh+1, dw+1) carry
  // the weights and all other taps are zero, with 4*Co phase-major output rows.
  L.mode = 2; L.groups = 1; L.Ci = Ci; L.co_real = Co; L.Co = 4 * Co; L.CoP = (L.Co + 31) & ~31;
  L.k = 3; L.stride = 1; L.dil = 1; L.pad = 1; L.tconv_u = 0; L.up2 = 1; L.ktaps = 9;
  L.CK = pick_ck(Ci, 9, 1);
  L.nchunk = (Ci + L.CK - 1) / L.CK;
  L.wBatch = (long long)L.nchunk * 9 * L.CK * L.CoP;
  std::vector<float> P((size_t)L.wBatch, 0.f);
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b)
      for (int co = 0; co < Co; ++co)
        for (int ci = 0; ci < Ci; ++ci)
          for (int dh = 0; dh < 2; ++dh)
            for (int dw = 0; dw < 2; ++dw) {
              const int kh = a + 1 - 2 * dh, kw = b + 1 - 2 * dw;
              if (kh < 0 || kh > 2 || kw < 0 || kw > 2) continue;
              const int u = (dh + 1) * 3 + (dw + 1);
              const int chunk = ci / L.CK, vcc = ci % L.CK;
              P[(((size_t)chunk * 9 + u) * L.CK + vcc) * L.CoP + (size_t)(a * 2 + b) * Co + co] =
                  w[(((size_t)ci * Co + co) * 3 + kh) * 3 + kw];
            }
  upload_layer(L, P, bias, Co);
  if ((g_precision == 2 || (g_precision == 1 && g_x3_default)) && Ci % 16 == 0) {
    // bf16x3 path: the same phase-major 3x3 convolution as a dense launch (split-K capable) into a [4 Co][H W] scratch, followed
    // by a phase interleave; needs the weights as [4 Co][Ci][9] and the bias once per phase row
    std::vector<float> Weq((size_t)4 * Co * Ci * 9, 0.f);
    for (int a = 0; a < 2; ++a)
      for (int b = 0; b < 2; ++b)
        for (int co = 0; co < Co; ++co)
          for (int ci = 0; ci < Ci; ++ci)
            for (int dh = 0; dh < 2; ++dh)
              for (int dw = 0; dw < 2; ++dw) {
                const int kh = a + 1 - 2 * dh, kw = b + 1 - 2 * dw;
                if (kh < 0 || kh > 2 || kw < 0 || kw > 2) continue;
                Weq[(((size_t)(a * 2 + b) * Co + co) * Ci + ci) * 9 + (dh + 1) * 3 + (dw + 1)] = w[(((size_t)ci * Co + co) * 3 + kh) * 3 + kw];
              }
    pack_x3(L, Weq.data(), 4 * Co, Ci, 9);
    if (bias) {
      std::vector<float> b4((size_t)4 * Co);
      for (int ph = 0; ph < 4; ++ph) for (int co = 0; co < Co; ++co) b4[(size_t)ph * Co + co] = bias[co];
      L.bd4_ = dev_upload(b4.data(), b4.size());
    }
  }
}

// out[co][2h + a][2w + b] = ph[(a * 2 + b) * Co + co][h][w]
__global__ void interleave2x2_kernel(const float* __restrict__ ph, float* __restrict__ out, int Co, int H, int Wd) {
  const long long n = (long long)Co * H * Wd * 4;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long st = (long long)gridDim.x * blockDim.x;
  const int W2 = 2 * Wd; const long long plane = (long long)4 * H * Wd;
  for (; i < n; i += st) {
    const int co = (int)(i / plane); const long long rem = i - (long long)co * plane;
    const int y = (int)(rem / W2), x = (int)(rem - (long long)y * W2);
    out[i] = ph[((long long)((y & 1) * 2 + (x & 1)) * Co + co) * ((long long)H * Wd) + (long long)(y >> 1) * Wd + (x >> 1)];
  }
}

// ---------------------------------------------------------------------------- launch
template <int WM, int WN, int AM, int AN, int MODE>
static void launch_cfg(const ConvArgsX& a, dim3 grid, size_t lds, hipStream_t s) {
  auto kern = conv_mfma_kernel<WM, WN, AM, AN, MODE>;
  RVC_ALLOW_BIG_LDS(kern);
  conv_launch(kern, grid, dim3(256), lds, s, a);
}

TileCfg choose_tile(int M, long long N, int batch) {
  // candidates ordered by preference for large problems; pick the first that yields enough workgroups
  const int Mp = (M + 31) / 32 * 32;
  TileCfg best{2, 2, 1, 1};
  if (Mp <= 32) {
    if (N * batch >= 512LL * 512) return TileCfg{1, 4, 1, 4};
    if (N * batch >= 256LL * 384) return TileCfg{1, 4, 1, 2};
    return TileCfg{1, 4, 1, 1};
  }
  auto blocks = [&](int bm, int bn) { return (long long)((M + bm - 1) / bm) * ((N + bn - 1) / bn) * batch; };
  static const int fill = exp_int("RVC_TILE_MINBLK", 384);    // workgroups a launch must have to take the larger tile
  if (Mp >= 128 && blocks(128, 128) >= fill) return TileCfg{2, 2, 2, 2};
  if (Mp <= 64 && blocks(64, 256) >= fill) return TileCfg{2, 2, 1, 4};
  if (blocks(64, 128) >= fill) return TileCfg{2, 2, 1, 2};
  return best;   // 64 x 64
}

// ---------------------------------------------------------------------------- optional per-launch profiling (HIP events)
struct ProfRec { hipEvent_t a, b; double flops; int cfg; double bytes; int Ci, Co, k, dil, stride, Tout, Wd, ksplit, fused; long long blocks; int h2; };
static std::atomic<bool> g_prof_on{false};
static std::vector<ProfRec> g_prof;
static std::mutex g_prof_mu;
static const char* kCfgNames[kProfCfgs] = {"1x4x1x4/1d", "1x4x1x2/1d", "1x4x1x1/1d", "2x2x2x2/1d", "2x2x1x4/1d", "2x2x1x2/1d", "2x2x1x1/1d",
                                           "1x4x1x4/2d", "1x4x1x2/2d", "1x4x1x1/2d", "2x2x2x2/2d", "2x2x1x4/2d", "2x2x1x2/2d", "2x2x1x1/2d",
                                           "1x4x1x4/x3", "1x4x1x2/x3", "1x4x1x1/x3", "2x2x2x2/x3", "2x2x1x4/x3", "2x2x1x2/x3", "2x2x1x1/x3",
                                           "2x2x2x4/x3", "1x4x2x4/x3", "2x4x2x4/x3"};
void conv_prof_enable(bool on) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& r : g_prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  g_prof.clear();
  g_prof_on = on;
}
// Sums the event-timed conv launches recorded since conv_prof_enable(true).  Per tile configuration: ms, flops, launches.
int conv_prof_collect(double* ms, double* flops, long long* launches) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (int i = 0; i < kProfCfgs; ++i) { ms[i] = 0; flops[i] = 0; launches[i] = 0; }
  for (auto& r : g_prof) {
    (void)hipEventSynchronize(r.b);
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) continue;
    ms[r.cfg] += t; flops[r.cfg] += r.flops; launches[r.cfg] += 1;
    if (exp_int("RVC_PROF_DUMP", 0)) fprintf(stderr, "conv launch %-12s %9.1f us %8.2f GFLOP %7.1f TFLOP/s\n", kCfgNames[r.cfg], t * 1e3, r.flops / 1e9, r.flops / t / 1e9);
  }
  return (int)g_prof.size();
}
const char* conv_prof_cfg_name(int i) { return (i >= 0 && i < kProfCfgs) ? kCfgNames[i] : ""; }
ProfKernelEvents& prof_kernel_events() { static thread_local ProfKernelEvents pe; return pe; }
ProfTicket conv_prof_begin(hipStream_t s) {
  ProfTicket t; t.on = g_prof_on.load(std::memory_order_relaxed);
  if (t.on) {
    (void)hipEventCreate(&t.a); (void)hipEventCreate(&t.b); (void)hipEventRecord(t.a, s);
    ProfKernelEvents& pe = prof_kernel_events();
    (void)hipEventCreate(&pe.ka); (void)hipEventCreate(&pe.kb); pe.armed = true; pe.launches = 0;
  }
  return t;
}
void conv_prof_end(ProfTicket& t, hipStream_t s, double flops, int cfg, double bytes, const ConvArgsX* a, long long blocks, int fused) {
  if (!t.on) return;
  (void)hipEventRecord(t.b, s);
  ProfKernelEvents& pe = prof_kernel_events();
  pe.armed = false;
  if (pe.launches == 1) {                                     // one kernel in the bracket: its own begin / end timestamps
    (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b);
    t.a = pe.ka; t.b = pe.kb;
  } else {
    (void)hipEventDestroy(pe.ka); (void)hipEventDestroy(pe.kb);
  }
  pe.ka = pe.kb = nullptr;
  ProfRec r{t.a, t.b, flops, cfg, bytes, 0, 0, 0, 0, 0, 0, 0, 1, fused, blocks, 0};
  if (a) { r.h2 = a->h2; r.Ci = a->Ci; r.Co = a->Co; r.k = a->kreal > 0 ? a->kreal : a->ktaps; r.dil = a->dil; r.stride = a->stride; r.Tout = a->Tout; r.Wd = a->Wd; r.ksplit = a->ksplit > 0 ? a->ksplit : 1; }
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof.push_back(r);
}
int conv_prof_dump_csv(const char* path) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  FILE* f = fopen(path, "w");
  if (!f) return -1;
  fprintf(f, "launch,kernel,tile,Ci,Co,k,dil,stride,Tout,Wd,fused_pair,ksplit,workgroups,us,alg_gflop,alg_mbytes,tflops,alg_gbps,mfma_per_product\n");
  int i = 0;
  for (auto& r : g_prof) {
    (void)hipEventSynchronize(r.b);
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) continue;
    // (fused >> 4 names the kernel of the split-MFMA family: 0 staged, 1 pipelined conv, 2 pipelined GEMM, 3 pipelined fused pair, 4 split-resident GEMM, 5 persistent fused pair with LDS-resident fp16 weights, 6 persistent pipelined conv, 7 whole ResBlock (three pairs) per launch)
    static const char* kX3Fam[8] = {"conv_x3_kernel", "conv_x3p_kernel", "conv_x3g_kernel", "conv_x3pf_kernel", "conv_x3s_kernel", "conv_rbh_kernel", "conv_x3q_kernel", "conv_rb3_kernel"};
    // (last column: matrix instructions per algorithmic product - 3 bf16x3, 2 fp16x2 (conv_x3q_kernel, H2), 16 fp32 MFMA at the bf16 rate's scale: 1 fp32 MFMA)
    fprintf(f, "%d,%s,%s,%d,%d,%d,%d,%d,%d,%d,%d,%d,%lld,%.2f,%.4f,%.3f,%.2f,%.1f,%d\n", i++, r.cfg >= 14 ? kX3Fam[(r.fused >> 4) & 7] : "conv_mfma_kernel", kCfgNames[r.cfg],
            r.Ci, r.Co, r.k, r.dil, r.stride, r.Tout, r.Wd, r.fused & 15, r.ksplit, r.blocks, t * 1e3, r.flops / 1e9, r.bytes / 1e6,
            t > 0 ? r.flops / t / 1e9 : 0.0, t > 0 ? r.bytes / t / 1e6 : 0.0, r.cfg >= 14 ? (r.h2 ? 2 : 3) : 1);
  }
  fclose(f);
  return i;
}
// algorithmic HBM bytes of one launch: input + output (+ residual unless it is the input itself, + previous output when accumulating) + weights, fp32
double conv_alg_bytes(const ConvArgsX& a, int batch) {
  const double in = (double)a.Ci * (a.Wd > 0 ? (double)a.Tin * a.Wd : (double)a.Tin);
  const double out = (double)a.Co * a.Tout;
  const double w = (double)a.Co * a.Ci * (a.kreal > 0 ? a.kreal : a.ktaps);
  const bool r_is_x = a.R != nullptr && a.R == a.X;            // a residual that IS the input tensor crosses HBM once
  return 4.0 * (batch * (in + out * (1.0 + ((a.R && !r_is_x) ? 1.0 : 0.0) + (a.accumulate ? 1.0 : 0.0))) + w);
}
// Per kernel configuration and roofline regime.  A launch counts as HBM-bound when its arithmetic intensity is below the ridge of
// its kernel (peak FLOP/s / 8 TB/s): out[cfg][0..3] = {ms, flops, bytes, launches} of the MFMA-bound launches, [4..7] of the
// HBM-bound ones.
int conv_prof_collect_ex(double* out, double ridge_fp32, double ridge_x3) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (int i = 0; i < kProfCfgs * 8; ++i) out[i] = 0;
  for (auto& r : g_prof) {
    (void)hipEventSynchronize(r.b);
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) continue;
    const double ridge = r.cfg >= 14 ? ridge_x3 : ridge_fp32;
    const int o = (r.bytes > 0 && r.flops / r.bytes < ridge) ? 4 : 0;
    double* q = out + r.cfg * 8 + o;
    q[0] += t; q[1] += r.flops; q[2] += r.bytes; q[3] += 1;
  }
  return (int)g_prof.size();
}
int tile_cfg_id(const TileCfg& t) {
  static const TileCfg all[7] = {{1, 4, 1, 4}, {1, 4, 1, 2}, {1, 4, 1, 1}, {2, 2, 2, 2}, {2, 2, 1, 4}, {2, 2, 1, 2}, {2, 2, 1, 1}};
  for (int i = 0; i < 7; ++i) if (all[i].WM == t.WM && all[i].WN == t.WN && all[i].AM == t.AM && all[i].AN == t.AN) return i;
  return -1;
}

// Fills the tile-dependent launch geometry; returns false when the tile does not fit the register prefetch slots / LDS.
static bool setup_tile(ConvArgsX& a, int mode, const TileCfg& t, size_t& lds) {
  const int BM = t.WM * t.AM * 32, BN = t.WN * t.AN * 32;
  const int kmode = mode == 2 ? 2 : (a.stride > 1 ? 3 : 1);
  int need;
  if (mode == 2) {
    // tile must be whole rows or a power-of-two fraction of a row
    a.BWd = BN < a.Wd ? BN : a.Wd;
    a.BH = BN < a.Wd ? 1 : BN / a.Wd;
    a.PW = a.BWd + 2;
    a.WROW = (a.BH + 2) * a.PW;
    if ((a.WROW & 1) == 0) a.WROW += 1;
    const unsigned RP = (unsigned)((a.BH + 2) * a.PW);
    a.magRP = (unsigned)((0x100000000ULL + RP - 1) / RP); a.magPW = (unsigned)((0x100000000ULL + a.PW - 1) / a.PW);
    a.ni = 1; a.magNI = 0;
    need = (a.CK * (int)RP + 255) / 256;
  } else {
    const int used = BN + (a.ktaps - 1) * (a.stride == 1 ? a.dil : 1);
    a.WROW = used | 1;
    const int span = a.stride == 1 ? used : used * a.stride;
    const int nrows = a.stride == 1 ? a.CK : a.CK / a.stride;
    a.ni = (span + 63) / 64;
    a.magNI = (unsigned)((0x100000000ULL + a.ni - 1) / a.ni);
    need = (nrows * a.ni + 3) / 4;
  }
  if (need > x_slots(BN, kmode)) return false;
  // taps per weight stage: the slab must fit the register prefetch slots (32 KB)
  int kt = a.ktaps;
  const int maxrows = (kWSlots * 256) / BM;
  if (a.CK > maxrows) return false;
  if (kt * a.CK > maxrows) kt = maxrows / a.CK;
  a.KT = kt;
  lds = ((size_t)((a.CK * a.WROW + 3) & ~3) + (size_t)a.KT * a.CK * BM) * sizeof(float);
  return lds <= 160 * 1024;
}

static void run_conv(ConvArgsX a, int mode, int batch, hipStream_t s, double flops) {
  RVC_REQUIRE(a.act == ACT_NONE || a.act == ACT_LRELU || a.act == ACT_RELU, "in-kernel activations are identity / ReLU / leaky ReLU");
  RVC_REQUIRE(a.pre_act == ACT_NONE || a.pre_act == ACT_LRELU, "input activation must be identity or leaky ReLU");
  TileCfg t = choose_tile(a.Co, a.Tout, batch);
  if (const char* f = RVC_EXP_STR("RVC_FORCE_TILE")) {   // experiments: "WM,WN,AM,AN"
    int w[4]; if (sscanf(f, "%d,%d,%d,%d", &w[0], &w[1], &w[2], &w[3]) == 4 && (a.Co > 32 || w[0] == 1)) t = TileCfg{w[0], w[1], w[2], w[3]};
  }
  size_t lds = 0;
  if (!setup_tile(a, mode, t, lds)) {
    // fall back to narrower tiles (fewer staged columns per row)
    const TileCfg alts[] = {{2, 2, 1, 2}, {2, 2, 1, 1}, {1, 4, 1, 1}};
    bool ok = false;
    for (const TileCfg& c : alts) { if (a.Co <= 32 && c.WM == 2) continue; t = c; if (setup_tile(a, mode, t, lds)) { ok = true; break; } }
    if (!ok) { t = TileCfg{2, 2, 1, 1}; ok = setup_tile(a, mode, t, lds); }
    RVC_REQUIRE(ok, "no tile configuration fits this convolution");
  }
  const int BM = t.WM * t.AM * 32, BN = t.WN * t.AN * 32;
  RVC_REQUIRE((double)a.orows * (double)a.ldY * 4.0 < 2147483648.0 && (double)a.orows * (double)a.ldR * 4.0 < 2147483648.0 &&
              (double)a.Ci * (double)a.ldX * 4.0 < 2147483648.0, "tensor extent exceeds the 32-bit buffer addressing of the conv kernel");
  // split-K: small grids (deep U-Net levels, 1599-frame GEMMs) leave most CUs idle and expose every stage's load latency;
  // slicing the reduction over S workgroups restores occupancy.  Partials are reduced in a fixed order (deterministic).
  const long long nblk = (long long)((a.Tout + BN - 1) / BN) * ((a.Co + BM - 1) / BM) * batch;
  int S = 1;
  static const int max_split = exp_int("RVC_SPLITK", 8);
  static const int split_blk = exp_int("RVC_SPLITK_BLK", 400);
  if (a.ostride == 1 && !a.up2 && nblk < split_blk) {
    S = (int)((2 * split_blk + nblk - 1) / nblk);
    if (S > max_split) S = max_split;
    if (S > a.nchunk / 2) S = a.nchunk / 2;
    if (S < 1) S = 1;
    while (S > 1 && ((a.nchunk + S - 1) / S) * (S - 1) >= a.nchunk) --S;     // every split must own at least one chunk
  }
  a.ksplit = S; a.partial = nullptr; a.ldP = (a.Tout + 31) & ~31;
  if (S > 1) a.partial = (float*)stream_scratch(s, 0, (size_t)S * batch * a.Co * a.ldP * sizeof(float));
  dim3 grid((unsigned)((a.Tout + BN - 1) / BN), (unsigned)((a.Co + BM - 1) / BM), (unsigned)(batch * S));
  ProfTicket rec; int cfg_id = 0;
  auto finish = [&]() {
    if (S > 1) {
      const long long total = (long long)batch * a.Co * a.Tout;
      (void)total;
      splitk_reduce_launch(a, S, batch, s);
    }
    conv_prof_end(rec, s, flops, cfg_id, conv_alg_bytes(a, batch), &a, (long long)grid.x * grid.y * grid.z);
  };
#define RVC_LAUNCH(ID_, WM_, WN_, AM_, AN_)                                                     \
  if (t.WM == WM_ && t.WN == WN_ && t.AM == AM_ && t.AN == AN_) {                               \
    cfg_id = ID_ + (mode == 2 ? 7 : 0);                                                         \
    rec = conv_prof_begin(s);                                                                   \
    if (mode == 2) launch_cfg<WM_, WN_, AM_, AN_, 2>(a, grid, lds, s);                          \
    else if (a.stride > 1) launch_cfg<WM_, WN_, AM_, AN_, 3>(a, grid, lds, s);                  \
    else launch_cfg<WM_, WN_, AM_, AN_, 1>(a, grid, lds, s);                                    \
    finish();                                                                                   \
    return;                                                                                     \
  }
  RVC_LAUNCH(0, 1, 4, 1, 4)
  RVC_LAUNCH(1, 1, 4, 1, 2)
  RVC_LAUNCH(2, 1, 4, 1, 1)
  RVC_LAUNCH(3, 2, 2, 2, 2)
  RVC_LAUNCH(4, 2, 2, 1, 4)
  RVC_LAUNCH(5, 2, 2, 1, 2)
  RVC_LAUNCH(6, 2, 2, 1, 1)
#undef RVC_LAUNCH
  throw Error("no tile configuration matched");
}

static bool simple_act(int act) { return act == ACT_NONE || act == ACT_LRELU || act == ACT_RELU; }
// Activations outside the max(v, slope v) family are applied by a second elementwise pass over the conv output.
static ConvEpilogue split_epilogue(const ConvEpilogue& e, bool& post) {
  post = !simple_act(e.act);
  if (!post) return e;
  RVC_REQUIRE(e.out_scale == 1.f && !e.accumulate, "GELU/tanh/sigmoid/log epilogues do not combine with scale/accumulate");
  ConvEpilogue c = e;
  c.act = ACT_NONE; c.act_slope = 0.f; c.R = nullptr; c.ldR = 0;
  return c;
}
static void fill_epilogue(ConvArgsX& a, const ConvEpilogue& e) {
  a.R = e.R; a.ldR = e.ldR; a.pre_act = e.pre_act; a.pre_slope = e.pre_slope; a.act = e.act; a.act_slope = e.act_slope;
  a.act_before_res = e.act_before_res; a.out_scale = e.out_scale; a.accumulate = e.accumulate;
  a.Xs = e.xs_in; a.xsTp = e.xs_tp; a.Ys = e.ys_out; a.ysTp = e.ys_tp; a.ys_slope = e.ys_slope;
}

int conv1d_out_len(const ConvLayer& L, int Tin) {
  if (L.tconv_u > 0) return (Tin - 1) * L.tconv_u - 2 * L.pad + L.k;
  return (Tin + 2 * L.pad - L.dil * (L.k - 1) - 1) / L.stride + 1;
}

void conv1d_run(const ConvLayer& L, hipStream_t s, const float* X, long long ldX, int Tin, float* Y, long long ldY,
                const ConvEpilogue& e0) {
  RVC_REQUIRE(L.mode == 1 && L.Wd_, "conv1d_run on an uninitialised / non-1D layer");
  bool post; const ConvEpilogue e = split_epilogue(e0, post);
  ConvArgsX a{};
  a.X = X; a.W = L.Wd_; a.bias = L.bd_; a.Y = Y;
  fill_epilogue(a, e);
  a.Ci = L.Ci; a.Co = L.Co; a.CoP = L.CoP; a.Tin = Tin; a.Wd = 0; a.ktaps = L.ktaps; a.dil = L.dil; a.stride = L.stride;
  a.CK = L.CK; a.nchunk = L.nchunk;
  a.ldX = ldX; a.ldY = ldY; a.up2 = 0;
  a.ldW = L.CoP; a.Wcols = L.CoP; a.Wrows = L.nchunk * L.ktaps * L.CK;
  if (e.bias_override) a.bias = e.bias_override;
  int Tout = conv1d_out_len(L, Tin);
  if (e.tout_limit > 0 && e.tout_limit < Tout) { RVC_REQUIRE(L.tconv_u == 0, "tout_limit on a transposed conv"); Tout = e.tout_limit; }
  if (L.tconv_u > 0) {
    a.pad = L.conv_pad;                             // left pad of the equivalent stride-1 conv
    a.Tout = (Tout + L.tconv_u - 1) / L.tconv_u;    // GEMM positions q
    a.ostride = L.tconv_u; a.orows = L.co_real;
    RVC_REQUIRE(ldY == Tout, "interleaved ConvTranspose1d store needs a dense output (ldY == Tout)");
    RVC_REQUIRE(e.R == nullptr, "residual not supported on the interleaved store");
  } else {
    a.pad = L.pad; a.Tout = Tout; a.ostride = 1; a.orows = L.Co;
  }
  a.xBatch = (long long)L.Ci * ldX; a.wBatch = L.wBatch; a.yBatch = (long long)L.Co * ldY; a.rBatch = (long long)L.Co * e.ldR;
  a.bBatch = L.Co;
  const double flops = L.tconv_u > 0 ? 2.0 * Tin * L.Ci * L.co_real * L.k
                                     : 2.0 * L.groups * (double)L.Co * a.Tout * L.Ci * L.k;
  a.Wx = reinterpret_cast<const unsigned char*>(L.Wx_); a.CoPx = L.CoPx; a.wxBatch = L.wxBatch; a.kreal = L.tconv_u > 0 ? L.ktaps : L.k;
  if (e.h2) {
    RVC_REQUIRE(L.Wh_ && (e.xs_in || e.ys_out), "fp16x2 arithmetic is for the two halves of a split-resident ResBlock pair (conv1d_pair_h2_eligible)");
    a.Wx = reinterpret_cast<const unsigned char*>(L.Wh_); a.h2 = 1;
  }
  if (!(L.Wx_ && conv_x3_try(a, L.groups, s, flops))) {
    RVC_REQUIRE(!e.xs_in && !e.ys_out, "split-resident tensors need the bf16x3 kernel (check conv1d_split_eligible first)");
    run_conv(a, 1, L.groups, s, flops);
  }
  if (post) {
    RVC_REQUIRE(L.tconv_u == 0, "post-activation on a transposed conv");
    act_res_inplace(s, Y, e0.R, L.groups * L.Co, Tout, ldY, e0.ldR, e0.act, e0.act_slope, e0.act_before_res);
  }
}

bool conv1d_pair_h2_eligible(const ConvLayer& c1, const ConvLayer& c2, int Tin) {
  if (!conv_set_pair_arithmetic(-1) || !c1.Wh_ || !c2.Wh_) return false;
  return conv1d_split_eligible(c1, Tin, SPLIT_PRODUCER, 1) && conv1d_split_eligible(c2, Tin, SPLIT_CONSUMER, 1);
}

bool conv1d_split_eligible(const ConvLayer& L, int Tin, SplitRole role, int h2) {
  if (L.mode != 1 || !L.Wx_ || (h2 && !L.Wh_) || L.tconv_u || L.stride != 1 || L.groups != 1 || (L.Co & 31) || (L.Ci & 15)) return false;
  if (conv1d_out_len(L, Tin) != Tin || L.pad > kSplitMargin) return false;      // the image is addressed as a "same" convolution's: row = margin + t
  ConvArgsX a{};
  a.Ci = L.Ci; a.Co = L.Co; a.CoP = L.CoP; a.Tin = Tin; a.Wd = 0; a.ktaps = L.ktaps; a.dil = L.dil; a.stride = 1; a.pad = L.pad;
  a.Tout = conv1d_out_len(L, Tin); a.ostride = 1; a.orows = L.Co; a.ldX = Tin; a.ldY = a.Tout; a.ldR = a.Tout;
  a.Wx = reinterpret_cast<const unsigned char*>(L.Wx_); a.CoPx = L.CoPx; a.kreal = L.k; a.h2 = h2;
  if (role == SPLIT_CONSUMER) a.R = reinterpret_cast<const float*>(L.Wx_);      // (c2 of a pair has a residual; the persistent kernel asks for it)
  // any non-null value asks for the role's geometry (dry run: never dereferenced)
  if (role == SPLIT_PRODUCER) { a.Ys = reinterpret_cast<unsigned char*>(L.Wx_); a.ysTp = split_image_tp(Tin); a.pre_act = ACT_LRELU; a.pre_slope = 0.1f; }
  else { a.Xs = reinterpret_cast<const unsigned char*>(L.Wx_); a.xsTp = split_image_tp(Tin); }
  return conv_x3_try(a, 1, nullptr, 0.0, true);
}

void gemm_tn_run(hipStream_t s, const float* A, long long ldA, long long aBatch, const float* B, long long ldB, long long bBatch,
                 float* Y, long long ldY, long long yBatch, int M, int N, int K, int batch, const float* bias, int biasBatch,
                 const ConvEpilogue& e) {
  RVC_REQUIRE(simple_act(e.act), "gemm_tn_run supports identity / ReLU / leaky ReLU epilogues");
  ConvArgsX a{};
  a.X = B; a.W = A; a.bias = bias; a.Y = Y;
  fill_epilogue(a, e);
  a.Ci = K; a.Co = M; a.CoP = M; a.Tin = N; a.Tout = N; a.Wd = 0; a.ktaps = 1; a.dil = 1; a.stride = 1; a.pad = 0;
  a.CK = K >= 64 ? 64 : ((K + 1) & ~1); a.nchunk = (K + a.CK - 1) / a.CK;
  a.ldX = ldB; a.ldY = ldY; a.up2 = 0; a.ostride = 1; a.orows = M;
  a.ldW = ldA; a.Wcols = M; a.Wrows = K;
  a.xBatch = bBatch; a.wBatch = aBatch; a.yBatch = yBatch; a.rBatch = 0; a.bBatch = biasBatch;
  run_conv(a, 1, batch, s, 2.0 * M * (double)N * K * batch);
}

void conv2d_run(const ConvLayer& L, hipStream_t s, const float* X, long long ldX, int H, int Wd, float* Y, long long ldY,
                const ConvEpilogue& e) {
  RVC_REQUIRE(simple_act(e.act), "conv2d_run supports identity / ReLU / leaky ReLU epilogues");
  RVC_REQUIRE(L.mode == 2 && L.Wd_, "conv2d_run on an uninitialised / non-2D layer");
  RVC_REQUIRE((Wd & (Wd - 1)) == 0 && Wd >= 2, "width must be a power of two");
  ConvArgsX a{};
  a.X = X; a.W = L.Wd_; a.bias = L.bd_; a.Y = Y;
  fill_epilogue(a, e);
  a.Ci = L.Ci; a.Co = L.Co; a.CoP = L.CoP; a.Tin = H; a.Tout = H * Wd; a.Wd = Wd; a.ktaps = 9; a.dil = 1; a.stride = 1; a.pad = 1;
  a.CK = L.CK; a.nchunk = L.nchunk;
  a.ldX = ldX; a.ldY = ldY; a.up2 = L.up2; a.ostride = 1; a.orows = L.up2 ? L.co_real : L.Co;
  a.ldW = L.CoP; a.Wcols = L.CoP; a.Wrows = L.nchunk * 9 * L.CK;
  a.xBatch = 0; a.wBatch = 0; a.yBatch = 0; a.rBatch = 0; a.bBatch = 0;
  const double flops = 2.0 * H * (double)Wd * 9 * L.Ci * (L.up2 ? L.co_real : L.Co);
  a.Wx = reinterpret_cast<const unsigned char*>(L.Wx_); a.CoPx = L.CoPx; a.wxBatch = L.wxBatch; a.kreal = 9;
  if (L.up2 && L.Wx_ && !e.R && !e.accumulate) {
    // ConvTranspose2d on the bf16x3 kernel: dense phase-major rows, then the 2x2 interleave
    ConvArgsX d = a;
    float* tmp = (float*)stream_scratch(s, 4, (size_t)L.Co * H * Wd * sizeof(float));
    d.Y = tmp; d.ldY = (long long)H * Wd; d.up2 = 0; d.orows = L.Co; d.bias = L.bd4_;
    if (conv_x3_try(d, 1, s, flops)) {
      const long long n = (long long)L.Co * H * Wd;
      int blocks = (int)((n + 255) / 256); if (blocks > 8192) blocks = 8192;
      hipLaunchKernelGGL(interleave2x2_kernel, dim3(blocks), dim3(256), 0, s, tmp, Y, L.co_real, H, Wd);
      return;
    }
  }
  if (!(L.Wx_ && !L.up2 && conv_x3_try(a, 1, s, flops))) run_conv(a, 2, 1, s, flops);
}

bool conv2d_kx_try(const ConvLayer& L, hipStream_t s, const float* X, long long ldX, int H, int Wd, float* Y, long long ldY, const ConvEpilogue& e, bool dry) {
  RVC_REQUIRE(L.mode == 2 && L.Wx_, "conv2d_kx_try on a layer without a bf16x3 image");
  if ((Wd & (Wd - 1)) != 0 || Wd < 2 || !simple_act(e.act)) return false;
  ConvArgsX a{};
  a.X = X; a.W = nullptr; a.bias = L.bd_; a.Y = Y;
  fill_epilogue(a, e);
  a.Ci = L.Ci; a.Co = L.Co; a.CoP = L.CoP; a.Tin = H; a.Tout = H * Wd; a.Wd = Wd; a.ktaps = L.ktaps; a.dil = 1; a.stride = 1; a.pad = 0;
  a.KH = L.kh; a.KW = L.kw; a.PH = L.ph; a.PWL = L.pwl;
  a.CK = 16; a.nchunk = L.nchunk;
  a.ldX = ldX; a.ldY = ldY; a.up2 = 0; a.ostride = 1; a.orows = L.Co;
  a.xBatch = a.wBatch = a.yBatch = a.rBatch = 0; a.bBatch = L.Co;
  a.Wx = reinterpret_cast<const unsigned char*>(L.Wx_); a.CoPx = L.CoPx; a.wxBatch = L.wxBatch; a.kreal = L.ktaps;
  return conv_x3_try(a, 1, s, 2.0 * H * (double)Wd * L.ktaps * L.Ci * L.Co, dry);
}

}  // namespace rvc
