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
#include "rvc_internal.h"

#ifndef RVC_EPI_TWOPHASE
#define RVC_EPI_TWOPHASE 0
#endif

namespace rvc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvArgsX : ConvArgs {
  long long ldW;   // pitch (floats) of one packed-weight row
  int Wcols;       // valid columns in a weight row
  int Wrows;       // valid rows of the weight matrix
  unsigned magRP, magPW;   // 2-D: ceil(2^32 / d) for d = (BH+2)*PW and d = PW (exact division of small tile indices)
};

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

constexpr int kXSlots = 36;   // register slots (floats per thread) for the prefetched input tile
constexpr int kWSlots = 32;   // register slots (floats per thread) for the prefetched weight slab (<= 32 KB / 256 threads)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float buf_load(__amdgpu_buffer_rsrc_t r, unsigned voff) {   // out-of-range -> 0 (hardware bounds check)
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, 0, 0));
}

// Software pipeline: the global loads of stage s+1 (input tile with halo + weight slab) are issued into registers before the
// MFMA loop of stage s and written to LDS after it, so HBM/L2 latency is covered by matrix work instead of a barrier wait.
// Loads go through buffer descriptors: zero padding, channel tails and ragged edges come from the hardware range check.
// MODE 1: 1-D stride 1, MODE 2: 2-D 3x3, MODE 3: 1-D strided (phase-decomposed rows).
template <int WM, int WN, int AM, int AN, int MODE>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const ConvArgsX p) {
  constexpr int BM = WM * AM * 32, BN = WN * AN * 32;
  constexpr unsigned OOB = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int CK = p.CK, WROW = p.WROW;
  float* Xs = smem;
  float* Ws = smem + ((CK * WROW + 3) & ~3);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;
  const int z = blockIdx.z;
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
  const int nstages = p.nchunk * ntb;
  const bool wvec = (p.ldW & 3) == 0 && (p.Wcols & 3) == 0 && ((((uintptr_t)W) & 15) == 0);
  const int used1 = BN + (p.ktaps - 1) * (MODE == 1 ? p.dil : 1);   // 1-D: LDS columns in use per row
  const __amdgpu_buffer_rsrc_t wrs = make_rsrc(W, (unsigned)p.Wrows * (unsigned)p.ldW * 4u);

  const float pre_slope = p.pre_act == ACT_LRELU ? p.pre_slope : 1.f;   // input activation: leaky ReLU (slope 1 = identity)
  float xr[kXSlots];
  float wr[kWSlots];
  const int lane0 = lane, tid0 = tid;   // re-materialised inside the staging lambdas (keeps their address math out of loop-invariant registers)

  // ---- input tile: global -> registers
  auto load_x = [&](int chunk) {
    int lane = lane0, tid = tid0;
    asm volatile("" : "+v"(lane), "+v"(tid));
    if (MODE == 1) {
      int row = wave, qb = 0;
      const int ci0 = chunk * CK;
      __amdgpu_buffer_rsrc_t rs = make_rsrc(X + (long long)(ci0 + row) * p.ldX, (ci0 + row < p.Ci && row < CK) ? (unsigned)p.Tin * 4u : 0u);
#pragma unroll
      for (int s = 0; s < kXSlots; ++s) {
        const int x = n0 - p.pad + qb + lane;
        float v = buf_load(rs, x >= 0 ? (unsigned)x * 4u : OOB);
        v = fmaxf(v, v * pre_slope);
        xr[s] = v;
        qb += 64;
        if (qb >= used1) {
          qb = 0; row += 4;
          rs = make_rsrc(X + (long long)(ci0 + row) * p.ldX, (ci0 + row < p.Ci && row < CK) ? (unsigned)p.Tin * 4u : 0u);
        }
      }
    } else if (MODE == 3) {
      const int st = p.stride, span = used1 * st, cpc = CK / st, c0 = chunk * cpc;
      int cl = wave, eb = 0;
      __amdgpu_buffer_rsrc_t rs = make_rsrc(X + (long long)(c0 + cl) * p.ldX, (c0 + cl < p.Ci && cl < cpc) ? (unsigned)p.Tin * 4u : 0u);
#pragma unroll
      for (int s = 0; s < kXSlots; ++s) {
        const int x = n0 * st - p.pad + eb + lane;
        float v = buf_load(rs, x >= 0 ? (unsigned)x * 4u : OOB);
        v = fmaxf(v, v * pre_slope);
        xr[s] = v;
        eb += 64;
        if (eb >= span) {
          eb = 0; cl += 4;
          rs = make_rsrc(X + (long long)(c0 + cl) * p.ldX, (c0 + cl < p.Ci && cl < cpc) ? (unsigned)p.Tin * 4u : 0u);
        }
      }
    } else {
      const int RP = (p.BH + 2) * p.PW, total = CK * RP;
      const __amdgpu_buffer_rsrc_t rs = make_rsrc(X, (unsigned)p.Ci * (unsigned)p.ldX * 4u);   // whole tensor: the channel tail falls out of range
#pragma unroll
      for (int s = 0; s < kXSlots; ++s) {
        const int e = tid + 256 * s;
        const int vcc = __umulhi((unsigned)e, p.magRP), rem = e - vcc * RP;
        const int rr = __umulhi((unsigned)rem, p.magPW), cc = rem - rr * p.PW;
        const int ci = chunk * CK + vcc, hh = h0 - 1 + rr, ww = w0 - 1 + cc;
        const bool ok = e < total && hh >= 0 && hh < p.Tin && ww >= 0 && ww < p.Wd;
        xr[s] = buf_load(rs, ok ? ((unsigned)ci * (unsigned)p.ldX + (unsigned)hh * (unsigned)p.Wd + (unsigned)ww) * 4u : OOB);
      }
    }
  };
  // ---- input tile: registers -> LDS
  auto store_x = [&]() {
    int lane = lane0, tid = tid0;
    asm volatile("" : "+v"(lane), "+v"(tid));
    if (MODE == 1) {
      int row = wave, qb = 0;
#pragma unroll
      for (int s = 0; s < kXSlots; ++s) {
        const int q = qb + lane;
        if (row < CK && q < used1) Xs[row * WROW + q] = xr[s];
        qb += 64;
        if (qb >= used1) { qb = 0; row += 4; }
      }
    } else if (MODE == 3) {
      const int st = p.stride, span = used1 * st, cpc = CK / st;
      int cl = wave, eb = 0;
#pragma unroll
      for (int s = 0; s < kXSlots; ++s) {
        const int e = eb + lane;
        if (cl < cpc && e < span) { const int q = (st == 2) ? (e >> 1) : e / st; const int r = e - q * st; Xs[(cl * st + r) * WROW + q] = xr[s]; }
        eb += 64;
        if (eb >= span) { eb = 0; cl += 4; }
      }
    } else {
      const int RP = (p.BH + 2) * p.PW, total = CK * RP;
#pragma unroll
      for (int s = 0; s < kXSlots; ++s) {
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
        const unsigned off = rr < rows ? ((row0 + rr) * (unsigned)p.ldW + (unsigned)(co0 + c4)) * 4u : OOB;
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wrs, (int)off, 0, 0);
        wr[4 * s] = __uint_as_float(v.x); wr[4 * s + 1] = __uint_as_float(v.y); wr[4 * s + 2] = __uint_as_float(v.z); wr[4 * s + 3] = __uint_as_float(v.w);
      }
    } else {
#pragma unroll
      for (int s = 0; s < kWSlots; ++s) {
        const int e = tid + 256 * s;
        const int rr = e / BM, c = e - rr * BM;
        wr[s] = buf_load(wrs, (rr < rows && co0 + c < p.Wcols) ? ((row0 + rr) * (unsigned)p.ldW + (unsigned)(co0 + c)) * 4u : OOB);
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

  load_x(0);
  load_w(0, 0);
  int chunk = 0, tb = 0;
  for (int stg = 0; stg < nstages; ++stg) {
    __syncthreads();                       // every wave is done reading the previous stage from LDS
    if (tb == 0) store_x();
    store_w(tb);
    __syncthreads();
    int nchunk2 = chunk, ntb2 = tb + 1;
    if (ntb2 == ntb) { ntb2 = 0; nchunk2 = chunk + 1; }
    if (stg + 1 < nstages) {               // prefetch the next stage; its latency hides under the MFMA loop below
      if (ntb2 == 0) load_x(nchunk2);
      load_w(nchunk2, ntb2);
    }
    const int ut = min(p.KT, p.ktaps - tb * p.KT);
    for (int uu = 0; uu < ut; ++uu) {
      const int u = tb * p.KT + uu;
      int toff;
      if (MODE == 2) toff = (u / 3) * p.PW + (u % 3);
      else toff = u * p.dil;
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
    chunk = nchunk2; tb = ntb2;
  }

  // -------------------------------------------------------------------------- epilogue
  // Two phases per accumulator row block: first every residual / accumulate operand is loaded (all loads in flight at once),
  // then the results are computed and stored.  Interleaving loads with stores would serialise on vmcnt, which counts both.
  const float* __restrict__ bias = p.bias ? p.bias + (long long)z * p.bBatch : nullptr;
  const float* R = p.R ? p.R + (long long)z * p.rBatch : nullptr;
  float* Y = p.Y + (long long)z * p.yBatch;
  auto out_index = [&](int co, int ph, int n, long long& oidx) -> bool {
    if (MODE == 2) {
      if (p.up2) {
        const int hh = n / p.Wd, ww = n - hh * p.Wd;
        oidx = (long long)co * p.ldY + (long long)(2 * hh + (ph >> 1)) * (2 * p.Wd) + 2 * ww + (ph & 1);
      } else {
        oidx = (long long)co * p.ldY + n;
      }
      return true;
    }
    if (p.ostride == 1) { oidx = (long long)co * p.ldY + n; return true; }
    const long long to = (long long)n * p.ostride + ph;
    oidx = (long long)co * p.ldY + to;
    return to < p.ldY;          // ldY doubles as the true output length for interleaved stores
  };
#if RVC_EPI_TWOPHASE
#pragma unroll
  for (int am = 0; am < AM; ++am) {
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {          // 4 accumulator registers (= 4 consecutive output rows) at a time
      float rv[AN][4], yv[AN][4], bvv[4];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int m = co0 + (wm * AM + am) * 32 + rr + 8 * rg + 4 * lh;
        const bool mok = m < p.Co;
        const int co = mok ? m % p.orows : 0, ph = mok ? m / p.orows : 0;
        bvv[rr] = (bias && mok) ? bias[co] : 0.f;
#pragma unroll
        for (int an = 0; an < AN; ++an) {
          const int n = n0 + (wn * AN + an) * 32 + li;
          long long oidx;
          const bool ok = mok && n < p.Tout && out_index(co, ph, n, oidx);
          rv[an][rr] = (ok && R) ? R[(long long)co * p.ldR + n] : 0.f;
          yv[an][rr] = (ok && p.accumulate) ? Y[oidx] : 0.f;
        }
      }
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int m = co0 + (wm * AM + am) * 32 + rr + 8 * rg + 4 * lh;
        if (m >= p.Co) continue;
        const int co = m % p.orows, ph = m / p.orows;
#pragma unroll
        for (int an = 0; an < AN; ++an) {
          const int n = n0 + (wn * AN + an) * 32 + li;
          long long oidx;
          if (n >= p.Tout || !out_index(co, ph, n, oidx)) continue;
          float v = acc[am][an][rg * 4 + rr] + bvv[rr];
          if (p.act_before_res) v = apply_act(v, p.act, p.act_slope) + rv[an][rr];
          else v = apply_act(v + rv[an][rr], p.act, p.act_slope);
          Y[oidx] = v * p.out_scale + yv[an][rr];
        }
      }
    }
  }
}
#else
#pragma unroll
  for (int am = 0; am < AM; ++am) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = co0 + (wm * AM + am) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (m >= p.Co) continue;
      const int co = m % p.orows, ph = m / p.orows;
      const float bv = bias ? bias[co] : 0.f;
#pragma unroll
      for (int an = 0; an < AN; ++an) {
        const int n = n0 + (wn * AN + an) * 32 + li;
        long long oidx;
        if (n >= p.Tout || !out_index(co, ph, n, oidx)) continue;
        float v = acc[am][an][r] + bv;
        if (p.act_before_res) { v = apply_act(v, p.act, p.act_slope); if (R) v += R[(long long)co * p.ldR + n]; }
        else { if (R) v += R[(long long)co * p.ldR + n]; v = apply_act(v, p.act, p.act_slope); }
        v *= p.out_scale;
        if (p.accumulate) v += Y[oidx];
        Y[oidx] = v;
      }
    }
  }
}
#endif

// ============================================================================ host side
float* dev_upload(const float* host, size_t n) {
  float* d = nullptr;
  RVC_HIP_CHECK(hipMalloc(&d, (n ? n : 1) * sizeof(float)));
  if (n) RVC_HIP_CHECK(hipMemcpy(d, host, n * sizeof(float), hipMemcpyHostToDevice));
  return d;
}
void dev_free(void* p) { if (p) (void)hipFree(p); }

void Arena::ensure(size_t bytes) {
  if (bytes <= cap) return;
  if (base) { RVC_HIP_CHECK(hipDeviceSynchronize()); (void)hipFree(base); base = nullptr; cap = 0; }
  size_t want = bytes + bytes / 8 + (1 << 20);
  RVC_HIP_CHECK(hipMalloc(&base, want));
  cap = want;
}
void Arena::release() { if (base) (void)hipFree(base); base = nullptr; cap = 0; }

static int pick_ck(int V, int ktaps, int stride) {
  int ck = ktaps >= 4 ? 8 : (ktaps >= 2 ? 16 : 32);
  if (stride > 1) { ck = ((ck + stride - 1) / stride) * stride; if (ck & 1) ck *= 2; }
  if (V < ck) { ck = stride > 1 ? ((V + stride - 1) / stride) * stride : V; if (ck & 1) ck += (stride > 1 ? stride : 1); if (ck < 2) ck = 2; }
  return ck;
}

static void upload_layer(ConvLayer& L, const std::vector<float>& packed, const float* bias, int nbias) {
  L.Wd_ = dev_upload(packed.data(), packed.size());
  L.bd_ = bias ? dev_upload(bias, nbias) : nullptr;
}

void conv_layer_free(ConvLayer& L) { dev_free(L.Wd_); dev_free(L.bd_); L.Wd_ = L.bd_ = nullptr; }

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
}

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
          P[(((size_t)chunk * L.ktaps + j) * L.CK + vcc) * L.CoP + (size_t)r * Co + co] = w[((size_t)ci * Co + co) * k + kk];
        }
  // remember the true transposed-conv geometry for the output length
  L.k = k; L.pad = pad;
  L.dil = 1;
  L.conv_pad = E2;
  upload_layer(L, P, bias, Co);
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
}

// ---------------------------------------------------------------------------- launch
struct TileCfg { int WM, WN, AM, AN; };

template <int WM, int WN, int AM, int AN, int MODE>
static void launch_cfg(const ConvArgsX& a, dim3 grid, size_t lds, hipStream_t s) {
  auto kern = conv_mfma_kernel<WM, WN, AM, AN, MODE>;
  static bool attr_set = false;
  if (!attr_set) {
    RVC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
}

static TileCfg choose_tile(int M, long long N, int batch, int mode, int Wd) {
  // candidates ordered by preference for large problems; pick the first that yields enough workgroups
  const int Mp = (M + 31) / 32 * 32;
  TileCfg best{2, 2, 1, 1};
  if (Mp <= 32) {
    if (N * batch >= 512LL * 512) return TileCfg{1, 4, 1, 4};
    if (N * batch >= 256LL * 384) return TileCfg{1, 4, 1, 2};
    return TileCfg{1, 4, 1, 1};
  }
  auto blocks = [&](int bm, int bn) { return (long long)((M + bm - 1) / bm) * ((N + bn - 1) / bn) * batch; };
  if (Mp >= 128 && blocks(128, 128) >= 384) return TileCfg{2, 2, 2, 2};
  if (Mp <= 64 && blocks(64, 256) >= 384) return TileCfg{2, 2, 1, 4};
  if (blocks(64, 128) >= 384) return TileCfg{2, 2, 1, 2};
  (void)mode; (void)Wd;
  return best;   // 64 x 64
}

// ---------------------------------------------------------------------------- optional per-launch profiling (HIP events)
struct ProfRec { hipEvent_t a, b; double flops; int cfg; };
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;
static const char* kCfgNames[14] = {"1x4x1x4/1d", "1x4x1x2/1d", "1x4x1x1/1d", "2x2x2x2/1d", "2x2x1x4/1d", "2x2x1x2/1d", "2x2x1x1/1d",
                                    "1x4x1x4/2d", "1x4x1x2/2d", "1x4x1x1/2d", "2x2x2x2/2d", "2x2x1x4/2d", "2x2x1x2/2d", "2x2x1x1/2d"};
void conv_prof_enable(bool on) {
  for (auto& r : g_prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  g_prof.clear();
  g_prof_on = on;
}
// Sums the event-timed conv launches recorded since conv_prof_enable(true).  Per tile configuration: ms, flops, launches.
int conv_prof_collect(double* ms, double* flops, long long* launches) {
  for (int i = 0; i < 14; ++i) { ms[i] = 0; flops[i] = 0; launches[i] = 0; }
  for (auto& r : g_prof) {
    (void)hipEventSynchronize(r.b);
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) continue;
    ms[r.cfg] += t; flops[r.cfg] += r.flops; launches[r.cfg] += 1;
  }
  return (int)g_prof.size();
}
const char* conv_prof_cfg_name(int i) { return (i >= 0 && i < 14) ? kCfgNames[i] : ""; }

static void run_conv(ConvArgsX a, int mode, int batch, hipStream_t s, double flops) {
  TileCfg t = choose_tile(a.Co, a.Tout, batch, mode, a.Wd);
  int BM = t.WM * t.AM * 32, BN = t.WN * t.AN * 32;
  if (mode == 2) {
    // tile must be whole rows or a power-of-two fraction of a row
    a.BWd = BN < a.Wd ? BN : a.Wd;
    a.BH = BN < a.Wd ? 1 : BN / a.Wd;
    a.PW = a.BWd + 2;
    a.WROW = (a.BH + 2) * a.PW;
    if ((a.WROW & 1) == 0) a.WROW += 1;
  } else {
    const int used = BN + (a.ktaps - 1) * (a.stride == 1 ? a.dil : 1);
    a.WROW = used | 1;
  }
  // taps per weight stage: the slab must fit the register prefetch slots (32 KB)
  int kt = a.ktaps;
  const int maxrows = (kWSlots * 256) / BM;
  if (kt * a.CK > maxrows) kt = maxrows / a.CK > 0 ? maxrows / a.CK : 1;
  a.KT = kt;
  if (mode == 2) {
    const unsigned RP = (unsigned)((a.BH + 2) * a.PW);
    a.magRP = (unsigned)((0x100000000ULL + RP - 1) / RP); a.magPW = (unsigned)((0x100000000ULL + a.PW - 1) / a.PW);
    RVC_REQUIRE((a.CK * (int)RP + 255) / 256 <= kXSlots, "2-D input tile exceeds the register prefetch slots");
  } else if (a.stride == 1) {
    const int used = BN + (a.ktaps - 1) * a.dil;
    RVC_REQUIRE(((a.CK + 3) / 4) * ((used + 63) / 64) <= kXSlots, "1-D input tile exceeds the register prefetch slots");
  } else {
    const int span = (BN + a.ktaps - 1) * a.stride;
    RVC_REQUIRE(((a.CK / a.stride + 3) / 4) * ((span + 63) / 64) <= kXSlots, "strided input tile exceeds the register prefetch slots");
  }
  RVC_REQUIRE(a.KT * a.CK * BM <= kWSlots * 256, "weight slab exceeds the register prefetch slots");
  const size_t lds = ((size_t)((a.CK * a.WROW + 3) & ~3) + (size_t)a.KT * a.CK * BM) * sizeof(float);
  RVC_REQUIRE(lds <= 160 * 1024, "conv tile does not fit LDS");
  dim3 grid((unsigned)((a.Tout + BN - 1) / BN), (unsigned)((a.Co + BM - 1) / BM), (unsigned)batch);
  ProfRec rec{}; int cfg_id = 0;
#define RVC_LAUNCH(ID_, WM_, WN_, AM_, AN_)                                                     \
  if (t.WM == WM_ && t.WN == WN_ && t.AM == AM_ && t.AN == AN_) {                               \
    cfg_id = ID_ + (mode == 2 ? 7 : 0);                                                         \
    if (g_prof_on) { (void)hipEventCreate(&rec.a); (void)hipEventCreate(&rec.b); (void)hipEventRecord(rec.a, s); } \
    if (mode == 2) launch_cfg<WM_, WN_, AM_, AN_, 2>(a, grid, lds, s);                          \
    else if (a.stride > 1) launch_cfg<WM_, WN_, AM_, AN_, 3>(a, grid, lds, s);                  \
    else launch_cfg<WM_, WN_, AM_, AN_, 1>(a, grid, lds, s);                                    \
    if (g_prof_on) { (void)hipEventRecord(rec.b, s); rec.flops = flops; rec.cfg = cfg_id; g_prof.push_back(rec); } \
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

static void fill_epilogue(ConvArgsX& a, const ConvEpilogue& e) {
  a.R = e.R; a.ldR = e.ldR; a.pre_act = e.pre_act; a.pre_slope = e.pre_slope; a.act = e.act; a.act_slope = e.act_slope;
  a.act_before_res = e.act_before_res; a.out_scale = e.out_scale; a.accumulate = e.accumulate;
}

int conv1d_out_len(const ConvLayer& L, int Tin) {
  if (L.tconv_u > 0) return (Tin - 1) * L.tconv_u - 2 * L.pad + L.k;
  return (Tin + 2 * L.pad - L.dil * (L.k - 1) - 1) / L.stride + 1;
}

void conv1d_run(const ConvLayer& L, hipStream_t s, const float* X, long long ldX, int Tin, float* Y, long long ldY,
                const ConvEpilogue& e) {
  RVC_REQUIRE(L.mode == 1 && L.Wd_, "conv1d_run on an uninitialised / non-1D layer");
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
  run_conv(a, 1, L.groups, s, flops);
}

void gemm_tn_run(hipStream_t s, const float* A, long long ldA, long long aBatch, const float* B, long long ldB, long long bBatch,
                 float* Y, long long ldY, long long yBatch, int M, int N, int K, int batch, const float* bias, int biasBatch,
                 const ConvEpilogue& e) {
  ConvArgsX a{};
  a.X = B; a.W = A; a.bias = bias; a.Y = Y;
  fill_epilogue(a, e);
  a.Ci = K; a.Co = M; a.CoP = M; a.Tin = N; a.Tout = N; a.Wd = 0; a.ktaps = 1; a.dil = 1; a.stride = 1; a.pad = 0;
  a.CK = K >= 32 ? 32 : ((K + 1) & ~1); a.nchunk = (K + a.CK - 1) / a.CK;
  a.ldX = ldB; a.ldY = ldY; a.up2 = 0; a.ostride = 1; a.orows = M;
  a.ldW = ldA; a.Wcols = M; a.Wrows = K;
  a.xBatch = bBatch; a.wBatch = aBatch; a.yBatch = yBatch; a.rBatch = 0; a.bBatch = biasBatch;
  run_conv(a, 1, batch, s, 2.0 * M * (double)N * K * batch);
}

void conv2d_run(const ConvLayer& L, hipStream_t s, const float* X, long long ldX, int H, int Wd, float* Y, long long ldY,
                const ConvEpilogue& e) {
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
  run_conv(a, 2, 1, s, 2.0 * H * (double)Wd * 9 * L.Ci * (L.up2 ? L.co_real : L.Co));
}

}  // namespace rvc
