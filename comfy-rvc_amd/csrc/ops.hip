// Non-GEMM device ops of the RVC inference path (gfx950): normalisations, softmax, gates, scans, layout
// changes, SineGen source, GRU recurrence, RMVPE decode.  All tensors are channel-major [C][T] unless noted.
#include "rvc_internal.h"
#include "ops.h"

namespace rvc {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float gelu_f(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f)); }

// ---------------------------------------------------------------------------------------------- LayerNorm over channels
// y[c][t] = (x[c][t] (+ r[c][t]) - mean_t) * rstd_t * gamma[c] + beta[c];  block = 16 columns x 16 channel slices
// (64-byte row segments keep the loads coalesced while a 1599-frame sequence still yields 100 workgroups).
__global__ __launch_bounds__(256) void layernorm_c_kernel(const float* __restrict__ x, const float* __restrict__ r,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ y, int C, int T, long long ld, float eps) {
  __shared__ float s_a[16][17], s_b[16][17];
  const int col = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int t = blockIdx.x * 16 + col;
  const bool ok = t < T;
  // single sweep: shifted sums (shift = first channel's value) keep the variance free of cancellation
  const float shift = ok ? (x[t] + (r ? r[t] : 0.f)) : 0.f;
  float sum = 0.f, sq = 0.f;
  // four channels per iteration: their (independent) loads are in flight together - the kernel is latency-bound at ~100 workgroups
  int c = sl;
  if (ok) {
    for (; c + 48 < C; c += 64) {
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = x[(long long)(c + 16 * j) * ld + t];
      if (r) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += r[(long long)(c + 16 * j) * ld + t];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float d = v[j] - shift; sum += d; sq += d * d; }
    }
    for (; c < C; c += 16) { float v = x[(long long)c * ld + t]; if (r) v += r[(long long)c * ld + t]; v -= shift; sum += v; sq += v * v; }
  }
  s_a[sl][col] = sum; s_b[sl][col] = sq;
  __syncthreads();
  float ts = 0.f, tq = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) { ts += s_a[i][col]; tq += s_b[i][col]; }
  const float md = ts / (float)C;                       // mean - shift
  const float var = fmaxf(tq / (float)C - md * md, 0.f);
  const float mean = md + shift, rstd = rsqrtf(var + eps);
  if (ok) {
    c = sl;
    for (; c + 48 < C; c += 64) {
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = x[(long long)(c + 16 * j) * ld + t];
      if (r) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += r[(long long)(c + 16 * j) * ld + t];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) y[(long long)(c + 16 * j) * ld + t] = (v[j] - mean) * rstd * gamma[c + 16 * j] + beta[c + 16 * j];
    }
    for (; c < C; c += 16) {
      float v = x[(long long)c * ld + t]; if (r) v += r[(long long)c * ld + t];
      y[(long long)c * ld + t] = (v - mean) * rstd * gamma[c] + beta[c];
    }
  }
}
void layernorm_c(hipStream_t s, const float* x, const float* r, const float* gamma, const float* beta, float* y, int C, int T,
                 long long ld, float eps) {
  hipLaunchKernelGGL(layernorm_c_kernel, dim3((T + 15) / 16), dim3(256), 0, s, x, r, gamma, beta, y, C, T, ld, eps);
}

// The same LayerNorm with a second output: the bf16 hi / lo image of y that the split-resident GEMM stages (conv_x3s.hip;
// [16-channel chunk][hi | lo][8-channel half][margin + t][8 ch]).  The statistics sweep is the kernel above's; in the second sweep a thread
// owns GROUPS of 8 consecutive channels of its column (group = slice + 16 j), so that it holds one whole 16-byte image row: per channel
// the 16 columns of a block are still one 64-byte segment, and the 16 image rows of a (group, block) are 256 contiguous bytes.
// NG = groups of 8 channels per thread (C <= 128 NG): the column's values stay in registers between the statistics and the normalisation - the
// tensor is read once (NG = 0: any C, second sweep from memory)
template <int NG>
__global__ __launch_bounds__(256) void layernorm_c_split_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                float* __restrict__ y, unsigned char* __restrict__ img, long long tp, int margin,
                                                                int C, int T, long long ld, float eps) {
  __shared__ float s_a[16][17], s_b[16][17];
  const int col = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int t = blockIdx.x * 16 + col;
  const bool ok = t < T;
  const float shift = ok ? x[t] : 0.f;
  float sum = 0.f, sq = 0.f;
  const int G = C >> 3;
  constexpr int NGC = NG > 0 ? NG : 1;
  float vc[NGC][8];
  if (NG > 0) {
#pragma unroll
    for (int i = 0; i < NGC; ++i) {
      const int g = sl + 16 * i;
#pragma unroll
      for (int j = 0; j < 8; ++j) vc[i][j] = (ok && g < G) ? x[(long long)(8 * g + j) * ld + t] : shift;
    }
#pragma unroll
    for (int i = 0; i < NGC; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = vc[i][j] - shift; sum += d; sq += d * d; }      // (absent groups hold `shift`: d = 0)
  } else if (ok) {
    for (int g = sl; g < G; g += 16) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = x[(long long)(8 * g + j) * ld + t];
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = v[j] - shift; sum += d; sq += d * d; }
    }
  }
  s_a[sl][col] = sum; s_b[sl][col] = sq;
  __syncthreads();
  float ts = 0.f, tq = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) { ts += s_a[i][col]; tq += s_b[i][col]; }
  const float md = ts / (float)C;
  const float var = fmaxf(tq / (float)C - md * md, 0.f);
  const float mean = md + shift, rstd = rsqrtf(var + eps);
  if (!ok) return;
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
  auto emit = [&](int g, float (&v)[8]) {
    u32x4_t hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (v[j] - mean) * rstd * gamma[8 * g + j] + beta[8 * g + j];
    if (y) {
#pragma unroll
      for (int j = 0; j < 8; ++j) y[(long long)(8 * g + j) * ld + t] = v[j];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const __bf16 ah = (__bf16)v[2 * j], bh = (__bf16)v[2 * j + 1];
      const __bf16 al = (__bf16)(v[2 * j] - (float)ah), bl = (__bf16)(v[2 * j + 1] - (float)bh);
      hi[j] = (unsigned)__builtin_bit_cast(unsigned short, ah) | ((unsigned)__builtin_bit_cast(unsigned short, bh) << 16);
      lo[j] = (unsigned)__builtin_bit_cast(unsigned short, al) | ((unsigned)__builtin_bit_cast(unsigned short, bl) << 16);
    }
    const int chunk = g >> 1, half = g & 1;
    unsigned char* row = img + (((long long)chunk * 4 + half) * tp + margin + t) * 16;
    *reinterpret_cast<u32x4_t*>(row) = hi;
    *reinterpret_cast<u32x4_t*>(row + tp * 32) = lo;
  };
  if (NG > 0) {
#pragma unroll
    for (int i = 0; i < NGC; ++i) { const int g = sl + 16 * i; if (g < G) emit(g, vc[i]); }
  } else {
    for (int g = sl; g < G; g += 16) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = x[(long long)(8 * g + j) * ld + t];
      emit(g, v);
    }
  }
}
void layernorm_c_split(hipStream_t s, const float* x, const float* gamma, const float* beta, float* y, unsigned char* img, long long tp, int margin,
                       int C, int T, long long ld, float eps) {
  RVC_REQUIRE((C & 15) == 0 && img != nullptr, "layernorm_c_split: channels must be a multiple of 16");
  const dim3 grid((T + 15) / 16);
  if (C <= 256) hipLaunchKernelGGL(layernorm_c_split_kernel<2>, grid, dim3(256), 0, s, x, gamma, beta, y, img, tp, margin, C, T, ld, eps);
  else if (C <= 768) hipLaunchKernelGGL(layernorm_c_split_kernel<6>, grid, dim3(256), 0, s, x, gamma, beta, y, img, tp, margin, C, T, ld, eps);
  else hipLaunchKernelGGL(layernorm_c_split_kernel<0>, grid, dim3(256), 0, s, x, gamma, beta, y, img, tp, margin, C, T, ld, eps);
}

// ---------------------------------------------------------------------------------------------- GroupNorm(C, C) over time + GELU
__global__ __launch_bounds__(256) void groupnorm_t_gelu_kernel(float* __restrict__ x, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, int T, long long ld, float eps) {
  __shared__ double red[4];
  float* row = x + (long long)blockIdx.x * ld;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double s = 0.0;
  for (int t = tid; t < T; t += 256) s += (double)row[t];
  s = wave_sum_d(s);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  const double mean = (red[0] + red[1] + red[2] + red[3]) / (double)T;
  __syncthreads();
  double q = 0.0;
  for (int t = tid; t < T; t += 256) { const double d = (double)row[t] - mean; q += d * d; }
  q = wave_sum_d(q);
  if (lane == 0) red[wave] = q;
  __syncthreads();
  const double var = (red[0] + red[1] + red[2] + red[3]) / (double)T;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float g = gamma[blockIdx.x], b = beta[blockIdx.x], mf = (float)mean;
  for (int t = tid; t < T; t += 256) row[t] = gelu_f((row[t] - mf) * rstd * g + b);
}
void groupnorm_t_gelu(hipStream_t s, float* x, const float* gamma, const float* beta, int C, int T, long long ld, float eps) {
  hipLaunchKernelGGL(groupnorm_t_gelu_kernel, dim3(C), dim3(256), 0, s, x, gamma, beta, T, ld, eps);
}

// ---------------------------------------------------------------------------------------------- HuBERT conv0 + GroupNorm + GELU, fused
// feature_extractor.conv_layers[0] (transformers modeling_hubert.py:108-133 with feat_extract_norm = "group"): Conv1d(1, 512, 10, stride 5,
// no bias) -> GroupNorm(512, 512) (= per-channel statistics over time) -> GELU.  An output element costs 10 FMAs on the raw audio, the
// tensor is 512 x 102399 (210 MB at 30 s): instead of writing the convolution, re-reading it twice for the statistics and once more to
// normalise (frames + GEMM + groupnorm_t_gelu_kernel: 15 + 90 + 387 us), the convolution is evaluated TWICE from the 2 MB of audio - once
// for per-tile partial sums (fixed-order reduction: repeats are bit-identical), once fused with the normalisation and GELU - and the
// tensor crosses HBM exactly once, as the write of the result.
// Tile: 1024 positions x 64 channels per 256-thread block; a thread owns 4 positions (t, t + 256, ..) and walks the tile's channels with
// wave-uniform weights (scalar loads).
constexpr int kC0T = 1024, kC0C = 64, kC0K = 10, kC0S = 5;
constexpr int kC0ST = 512;                                  // positions per statistics tile
__device__ __forceinline__ float gelu_bf(float v) {        // exact-erf GELU, branch-free (Abramowitz & Stegun 7.1.26: |erf error| <= 1.5e-7)
  const float x = v * 0.70710678118654752440f, ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.f));
  const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * x * x);
  return 0.5f * v * (1.f + copysignf(fmaf(-poly, e, 1.f), x));
}
// Statistics pass: a thread IS a channel (its 10 weights in registers) and walks the tile's positions - every thread reads the same audio
// samples (LDS broadcast), so there is no cross-thread reduction at all: partial[tile][c] = {sum, sum of squares} over 512 positions,
// accumulated in float64 in groups of 8 positions.
__global__ __launch_bounds__(256) void hubert_conv0_stats_kernel(const float* __restrict__ audio, long long L, const float* __restrict__ w, int C, int T1,
                                                                 double* __restrict__ partial) {
  __shared__ __attribute__((aligned(16))) float xs[kC0ST * kC0S + kC0K + 6];
  const int tid = threadIdx.x;
  const int t0 = blockIdx.x * kC0ST, c = blockIdx.y * 256 + tid;
  const long long a0 = (long long)t0 * kC0S;
  for (int i = tid; i < kC0ST * kC0S + kC0K; i += 256) xs[i] = (a0 + i < L) ? audio[a0 + i] : 0.f;
  __syncthreads();
  float wc[kC0K];
#pragma unroll
  for (int k = 0; k < kC0K; ++k) wc[k] = c < C ? w[(long long)c * kC0K + k] : 0.f;
  const int np = min(kC0ST, T1 - t0);
  double s1 = 0.0, s2 = 0.0;
  for (int p0 = 0; p0 < np; p0 += 8) {
    float f1 = 0.f, f2 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float* xp = xs + (p0 + j) * kC0S;
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < kC0K; ++k) a = fmaf(wc[k], xp[k], a);
      if (p0 + j < np) { f1 += a; f2 = fmaf(a, a, f2); }
    }
    s1 += (double)f1; s2 += (double)f2;
  }
  if (c < C) { double* q = partial + ((long long)blockIdx.x * C + c) * 2; q[0] = s1; q[1] = s2; }
}
// per channel: fixed-order sum of the tile partials (16 lanes per channel, then a fixed shuffle tree) -> mean and rstd * gamma
__global__ __launch_bounds__(256) void hubert_conv0_reduce_kernel(const double* __restrict__ partial, int stiles, int C, int T1, const float* __restrict__ gamma,
                                                                  float eps, float* __restrict__ stat) {
  const int c = blockIdx.x * 16 + (threadIdx.x >> 4), part = threadIdx.x & 15;
  double s1 = 0.0, s2 = 0.0;
  if (c < C)
    for (int i = part; i < stiles; i += 16) { s1 += partial[((long long)i * C + c) * 2]; s2 += partial[((long long)i * C + c) * 2 + 1]; }
#pragma unroll
  for (int m = 1; m < 16; m <<= 1) { s1 += __shfl_xor(s1, m); s2 += __shfl_xor(s2, m); }
  if (part == 0 && c < C) {
    const double mean = s1 / (double)T1;
    double var = s2 / (double)T1 - mean * mean;
    if (var < 0.0) var = 0.0;
    stat[2 * c] = (float)mean; stat[2 * c + 1] = (float)(1.0 / sqrt(var + (double)eps)) * gamma[c];
  }
}
// Apply pass: tile of 1024 positions x 64 channels; a thread owns 4 positions (t, t + 256, ..) and walks the channels with wave-uniform
// weights and statistics.
__global__ __launch_bounds__(256) void hubert_conv0_apply_kernel(const float* __restrict__ audio, long long L, const float* __restrict__ w, int C, int T1,
                                                                 const float* __restrict__ stat, const float* __restrict__ beta, float* __restrict__ out, long long ld) {
  __shared__ float xs[kC0T * kC0S + kC0K];
  const int tid = threadIdx.x;
  const int t0 = blockIdx.x * kC0T, c0 = blockIdx.y * kC0C;
  const long long a0 = (long long)t0 * kC0S;
  for (int i = tid; i < kC0T * kC0S + kC0K; i += 256) xs[i] = (a0 + i < L) ? audio[a0 + i] : 0.f;
  __syncthreads();
  float x[4][kC0K];
  bool ok[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int tl = tid + 256 * j;
    ok[j] = t0 + tl < T1;
#pragma unroll
    for (int k = 0; k < kC0K; ++k) x[j][k] = xs[tl * kC0S + k];
  }
  for (int cc = 0; cc < kC0C; ++cc) {
    const int c = c0 + cc;
    if (c >= C) break;
    const float* wc = w + (long long)c * kC0K;
    const float mean = stat[2 * c], sc = stat[2 * c + 1], b = beta[c];          // sc = rstd * gamma
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < kC0K; ++k) a = fmaf(wc[k], x[j][k], a);
      if (ok[j]) out[(long long)c * ld + t0 + tid + 256 * j] = gelu_bf(fmaf(a - mean, sc, b));
    }
  }
}
// The same apply pass writing the bf16 hi / lo image of the output, DE-INTERLEAVED for the stride-2 layer that reads it (rvc_internal.h split_geom_s2: position t at
// row margin + (t >> 1) + (t & 1) H): a thread owns 4 positions and walks the channels eight at a time - one 16-byte row of the hi and of the lo plane per step.
__global__ __launch_bounds__(256) void hubert_conv0_apply_img_kernel(const float* __restrict__ audio, long long L, const float* __restrict__ w, int C, int T1,
                                                                     const float* __restrict__ stat, const float* __restrict__ beta, unsigned char* __restrict__ img,
                                                                     long long tp, int margin, int H) {
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  __shared__ float xs[kC0T * kC0S + kC0K];
  const int tid = threadIdx.x;
  const int t0 = blockIdx.x * kC0T, c0 = blockIdx.y * kC0C;
  const long long a0 = (long long)t0 * kC0S;
  for (int i = tid; i < kC0T * kC0S + kC0K; i += 256) xs[i] = (a0 + i < L) ? audio[a0 + i] : 0.f;
  __syncthreads();
  float x[4][kC0K];
  bool ok[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int tl = tid + 256 * j;
    ok[j] = t0 + tl < T1;
#pragma unroll
    for (int k = 0; k < kC0K; ++k) x[j][k] = xs[tl * kC0S + k];
  }
  for (int cg = 0; cg < kC0C; cg += 8) {
    const int cb = c0 + cg;                                   // eight channels = one half of a 16-channel chunk (C is a multiple of 16: host)
    if (cb >= C) break;
    float v[4][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = cb + e;
      const float* wc = w + (long long)c * kC0K;
      const float mean = stat[2 * c], sc = stat[2 * c + 1], b = beta[c];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < kC0K; ++k) a = fmaf(wc[k], x[j][k], a);
        v[j][e] = gelu_bf(fmaf(a - mean, sc, b));
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (!ok[j]) continue;
      u32x4_t hi, lo;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const __bf16 ah = (__bf16)v[j][2 * q], bh = (__bf16)v[j][2 * q + 1];
        const __bf16 al = (__bf16)(v[j][2 * q] - (float)ah), bl = (__bf16)(v[j][2 * q + 1] - (float)bh);
        hi[q] = (unsigned)__builtin_bit_cast(unsigned short, ah) | ((unsigned)__builtin_bit_cast(unsigned short, bh) << 16);
        lo[q] = (unsigned)__builtin_bit_cast(unsigned short, al) | ((unsigned)__builtin_bit_cast(unsigned short, bl) << 16);
      }
      const int t = t0 + tid + 256 * j;
      const int chunk = cb >> 4, half = (cb >> 3) & 1;
      unsigned char* row = img + (((long long)chunk * 4 + half) * tp + margin + (t >> 1) + (long long)(t & 1) * H) * 16;
      *reinterpret_cast<u32x4_t*>(row) = hi;
      *reinterpret_cast<u32x4_t*>(row + tp * 32) = lo;
    }
  }
}
void hubert_conv0_gn_gelu_img(hipStream_t s, const float* audio, long long L, const float* w, const float* gamma, const float* beta, int C, int T1, float eps,
                              unsigned char* img, long long tp, int margin, int H, double* partial, float* stat) {
  RVC_REQUIRE((C & 15) == 0 && (T1 + 1) / 2 + 1 <= H && tp >= margin + 2LL * H, "hubert_conv0_gn_gelu_img: image geometry");
  const int stiles = (T1 + kC0ST - 1) / kC0ST, tiles = (T1 + kC0T - 1) / kC0T;
  hipLaunchKernelGGL(hubert_conv0_stats_kernel, dim3((unsigned)stiles, (unsigned)((C + 255) / 256)), dim3(256), 0, s, audio, L, w, C, T1, partial);
  hipLaunchKernelGGL(hubert_conv0_reduce_kernel, dim3((unsigned)((C + 15) / 16)), dim3(256), 0, s, partial, stiles, C, T1, gamma, eps, stat);
  hipLaunchKernelGGL(hubert_conv0_apply_img_kernel, dim3((unsigned)tiles, (unsigned)((C + kC0C - 1) / kC0C)), dim3(256), 0, s, audio, L, w, C, T1, stat, beta, img, tp, margin, H);
}
void hubert_conv0_gn_gelu(hipStream_t s, const float* audio, long long L, const float* w, const float* gamma, const float* beta, int C, int T1, float eps,
                          float* out, long long ld, double* partial, float* stat) {
  const int stiles = (T1 + kC0ST - 1) / kC0ST, tiles = (T1 + kC0T - 1) / kC0T;
  hipLaunchKernelGGL(hubert_conv0_stats_kernel, dim3((unsigned)stiles, (unsigned)((C + 255) / 256)), dim3(256), 0, s, audio, L, w, C, T1, partial);
  hipLaunchKernelGGL(hubert_conv0_reduce_kernel, dim3((unsigned)((C + 15) / 16)), dim3(256), 0, s, partial, stiles, C, T1, gamma, eps, stat);
  hipLaunchKernelGGL(hubert_conv0_apply_kernel, dim3((unsigned)tiles, (unsigned)((C + kC0C - 1) / kC0C)), dim3(256), 0, s, audio, L, w, C, T1, stat, beta, out, ld);
}
size_t hubert_conv0_scratch_doubles(int C, int T1) { return (size_t)((T1 + kC0ST - 1) / kC0ST) * C * 2; }

// ---------------------------------------------------------------------------------------------- column softmax of S^T [Tk][Tq]
// Softmax over the key axis (rows) for every query column; optional relative-position bias
// rel[(k - q + win)][q] for |k - q| <= win (enc_p, reference attentions.py:230-239) and optional gather of the banded
// probabilities pb[r][q] = P[q][q + r - win] (reference attentions.py:260-267).  Block = 64 columns x 16 row slices: every
// wave touches 256 contiguous bytes of a row.  Two sweeps: online (max, sum) then normalise-and-store: 2 reads + 1 write of the
// score matrix (the second read comes from the 256 MB memory-side cache).
__global__ __launch_bounds__(1024) void softmax_cols_kernel(float* __restrict__ S, int Tk, int Tq, long long ld, long long batchS,
                                                            const float* __restrict__ rel, long long batchRel, int win,
                                                            float* __restrict__ pb, long long batchPb) {
  __shared__ float s_m[16][64], s_s[16][64];
  const int col = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int q = blockIdx.x * 64 + col;
  const bool ok = q < Tq;
  float* Sb = S + (long long)blockIdx.y * batchS;
  const float* relb = rel ? rel + (long long)blockIdx.y * batchRel : nullptr;
  float mx = -3.0e38f, sum = 0.f;
  if (ok) {
    int k = sl;
    for (; k + 48 < Tk; k += 64) {           // four rows in flight per thread
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = Sb[(long long)(k + 16 * j) * ld + q];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (relb) { const int d = k + 16 * j - q + win; if (d >= 0 && d <= 2 * win) v[j] += relb[(long long)d * Tq + q]; }
        const float nm = fmaxf(mx, v[j]);
        sum = sum * expf(mx - nm) + expf(v[j] - nm);
        mx = nm;
      }
    }
    for (; k < Tk; k += 16) {
      float v = Sb[(long long)k * ld + q];
      if (relb) { const int d = k - q + win; if (d >= 0 && d <= 2 * win) v += relb[(long long)d * Tq + q]; }
      const float nm = fmaxf(mx, v);
      sum = sum * expf(mx - nm) + expf(v - nm);
      mx = nm;
    }
  }
  s_m[sl][col] = mx; s_s[sl][col] = sum;
  __syncthreads();
  float gm = -3.0e38f;
#pragma unroll
  for (int i = 0; i < 16; ++i) gm = fmaxf(gm, s_m[i][col]);
  float gs = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) gs += s_s[i][col] * expf(s_m[i][col] - gm);
  const float inv = 1.f / gs;
  if (!ok) return;
  auto emit = [&](int k, float v) {
    const int d = k - q + win;
    const bool band = d >= 0 && d <= 2 * win;
    if (relb && band) v += relb[(long long)d * Tq + q];
    const float pv = expf(v - gm) * inv;
    Sb[(long long)k * ld + q] = pv;
    if (pb && band) pb[(long long)blockIdx.y * batchPb + (long long)d * Tq + q] = pv;
  };
  int k = sl;
  for (; k + 48 < Tk; k += 64) {
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = Sb[(long long)(k + 16 * j) * ld + q];
#pragma unroll
    for (int j = 0; j < 4; ++j) emit(k + 16 * j, v[j]);
  }
  for (; k < Tk; k += 16) emit(k, Sb[(long long)k * ld + q]);
}
void softmax_cols(hipStream_t s, float* S, int Tk, int Tq, long long ld, long long batchS, int batch, const float* rel,
                  long long batchRel, int win, float* pb, long long batchPb) {
  hipLaunchKernelGGL(softmax_cols_kernel, dim3((Tq + 63) / 64, batch), dim3(1024), 0, s, S, Tk, Tq, ld, batchS, rel, batchRel,
                     win, pb, batchPb);
}

// ---------------------------------------------------------------------------------------------- small elementwise ops
__global__ void fill_kernel(float* p, float v, long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long st = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += st) p[i] = v;
}
void fill(hipStream_t s, float* p, float v, long long n) {
  if (n <= 0) return;
  int blocks = (int)((n + 255) / 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(fill_kernel, dim3(blocks), dim3(256), 0, s, p, v, n);
}

// WN gate: out[c][t] = tanh(a[c][t] + g[c]) * sigmoid(a[c+H][t] + g[c+H])     (reference commons.py:211-218)
__global__ void wn_gate_kernel(const float* __restrict__ a, const float* __restrict__ g, float* __restrict__ out, int H, int T) {
  const long long n = (long long)H * T;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long st = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += st) {
    const int c = (int)(i / T);
    const float ta = a[i] + g[c], sa = a[i + n] + g[c + H];
    out[i] = tanhf(ta) * (1.f / (1.f + expf(-sa)));
  }
}
void wn_gate(hipStream_t s, const float* a, const float* g, float* out, int H, int T) {
  long long n = (long long)H * T; int blocks = (int)((n + 255) / 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(wn_gate_kernel, dim3(blocks), dim3(256), 0, s, a, g, out, H, T);
}

// The same gate with the result written ONLY as the split-resident image the res / skip GEMM stages (conv_x3s.hip): one thread = 8 channels of
// one frame = one 16-byte row of the hi and of the lo plane.
__global__ __launch_bounds__(256) void wn_gate_split_kernel(const float* __restrict__ a, const float* __restrict__ g, unsigned char* __restrict__ img, long long tp,
                                                            int margin, int H, int T) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int grp = blockIdx.y;
  if (t >= T) return;
  const long long n = (long long)H * T;
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = grp * 8 + j;
    const long long i = (long long)c * T + t;
    const float ta = a[i] + g[c], sa = a[i + n] + g[c + H];
    v[j] = tanhf(ta) * (1.f / (1.f + expf(-sa)));
  }
  u32x4_t hi, lo;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const __bf16 ah = (__bf16)v[2 * j], bh = (__bf16)v[2 * j + 1];
    const __bf16 al = (__bf16)(v[2 * j] - (float)ah), bl = (__bf16)(v[2 * j + 1] - (float)bh);
    hi[j] = (unsigned)__builtin_bit_cast(unsigned short, ah) | ((unsigned)__builtin_bit_cast(unsigned short, bh) << 16);
    lo[j] = (unsigned)__builtin_bit_cast(unsigned short, al) | ((unsigned)__builtin_bit_cast(unsigned short, bl) << 16);
  }
  unsigned char* row = img + (((long long)(grp >> 1) * 4 + (grp & 1)) * tp + margin + t) * 16;
  *reinterpret_cast<u32x4_t*>(row) = hi;
  *reinterpret_cast<u32x4_t*>(row + tp * 32) = lo;
}
void wn_gate_split(hipStream_t s, const float* a, const float* g, unsigned char* img, long long tp, int margin, int H, int T) {
  RVC_REQUIRE((H & 15) == 0, "wn_gate_split: hidden channels must be a multiple of 16");
  hipLaunchKernelGGL(wn_gate_split_kernel, dim3((T + 255) / 256, H / 8), dim3(256), 0, s, a, g, img, tp, margin, H, T);
}

// y = W x + b for a single vector (speaker conditioning: cond layers applied to g = emb_g[sid]).
__global__ __launch_bounds__(64) void gemv_kernel(const float* __restrict__ W, const float* __restrict__ x, const float* __restrict__ b,
                                                   float* __restrict__ y, int K, const float* __restrict__ add) {
  const int row = blockIdx.x, lane = threadIdx.x;
  float s = 0.f;
  for (int k = lane; k < K; k += 64) s += W[(long long)row * K + k] * x[k];
  s = wave_sum(s);
  if (lane == 0) y[row] = s + (b ? b[row] : 0.f) + (add ? add[row] : 0.f);
}
void gemv(hipStream_t s, const float* W, const float* x, const float* b, float* y, int N, int K, const float* add) {
  hipLaunchKernelGGL(gemv_kernel, dim3(N), dim3(64), 0, s, W, x, b, y, K, add);
}

// z_p = m + exp(logs) * noise * 0.66666      (reference models.py:801)
__global__ void zp_kernel(const float* __restrict__ stats, const float* __restrict__ noise, float* __restrict__ zp, int C, int T) {
  const long long n = (long long)C * T;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long st = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += st) zp[i] = stats[i] + expf(stats[i + n]) * noise[i] * 0.66666f;
}
void zp_sample(hipStream_t s, const float* stats, const float* noise, float* zp, int C, int T) {
  long long n = (long long)C * T; int blocks = (int)((n + 255) / 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(zp_kernel, dim3(blocks), dim3(256), 0, s, stats, noise, zp, C, T);
}

// channel flip (reference modules.py:373-380): y[c] = x[C-1-c]
__global__ void flip_c_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int T) {
  const long long n = (long long)C * T;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long st = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += st) { const int c = (int)(i / T); const int t = (int)(i - (long long)c * T); y[i] = x[(long long)(C - 1 - c) * T + t]; }
}
void flip_c(hipStream_t s, const float* x, float* y, int C, int T) {
  long long n = (long long)C * T; int blocks = (int)((n + 255) / 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(flip_c_kernel, dim3(blocks), dim3(256), 0, s, x, y, C, T);
}

// enc_p embedding: x[c][t] = lrelu((lin[c][t] + emb_pitch[pitch[t]][c]) * sqrt(C), 0.1)   (reference models.py:90-97)
__global__ void encp_embed_kernel(float* __restrict__ x, const float* __restrict__ emb, const long long* __restrict__ pitch, int C,
                                  int T, float scale) {
  const long long n = (long long)C * T;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long st = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += st) {
    const int c = (int)(i / T); const int t = (int)(i - (long long)c * T);
    float v = (x[i] + (emb ? emb[pitch[t] * C + c] : 0.f)) * scale;        // emb null: text encoder without pitch embedding (no-f0 models)
    x[i] = v > 0.f ? v : v * 0.1f;
  }
}
void encp_embed(hipStream_t s, float* x, const float* emb, const long long* pitch, int C, int T) {
  long long n = (long long)C * T; int blocks = (int)((n + 255) / 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(encp_embed_kernel, dim3(blocks), dim3(256), 0, s, x, emb, pitch, C, T, sqrtf((float)C));
}

// ---------------------------------------------------------------------------------------------- Conv1d to ONE output channel
// y[t] = act(sum_c sum_j w[c][j] * pre(x[c][t + j - pad])): the generator's conv_post (Ci = 32 or 16, k = 7, no bias, leaky-ReLU 0.01
// before, tanh after; reference models.py:561-563).  On the MFMA kernels a 1-row output wastes 31 of 32 tile rows and ran at
// 0.65 TB/s; this is a plain streaming kernel: one thread per output sample, weights through scalar loads, the 7-fold reuse of x
// served by L1 / L2.  HBM-bound: Ci * T * 4 bytes in, T * 4 out.
__global__ __launch_bounds__(256) void conv_to1_kernel(const float* __restrict__ x, long long ldx, const float* __restrict__ w, int Ci, int K,
                                                       int pad, int T, float pre_slope, int act_tanh, float* __restrict__ y) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  float acc = 0.f;
  for (int c = 0; c < Ci; ++c) {
    const float* xr = x + (long long)c * ldx;
    for (int j = 0; j < K; ++j) {
      const int q = t + j - pad;
      float v = (q >= 0 && q < T) ? xr[q] : 0.f;
      v = fmaxf(v, v * pre_slope);
      acc = fmaf(w[c * K + j], v, acc);
    }
  }
  y[t] = act_tanh ? tanhf(acc) : acc;
}
// k <= 9 with rows that start 16-byte aligned: a thread owns FOUR consecutive outputs and reads its 12-sample window as three float4 per channel
// (the one-output kernel issued 7 dword loads + 7 max + 7 FMAs per output and channel: instruction-bound at 0.78 TB/s on the 164 MB input)
__global__ __launch_bounds__(256) void conv_to1x4_kernel(const float* __restrict__ x, long long ldx, const float* __restrict__ w, int Ci, int K,
                                                         int pad, int T, float pre_slope, int act_tanh, float* __restrict__ y) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int t0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (t0 >= T) return;
  const bool inner = t0 >= 4 && t0 + 8 <= T;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  for (int c = 0; c < Ci; ++c) {
    const float* xr = x + (long long)c * ldx;
    float v[12];
    if (inner) {
      const f4 l = *reinterpret_cast<const f4*>(xr + t0 - 4), m = *reinterpret_cast<const f4*>(xr + t0), r = *reinterpret_cast<const f4*>(xr + t0 + 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) { v[i] = l[i]; v[4 + i] = m[i]; v[8 + i] = r[i]; }
    } else {
#pragma unroll
      for (int i = 0; i < 12; ++i) { const int q = t0 - 4 + i; v[i] = (q >= 0 && q < T) ? xr[q] : 0.f; }
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) v[i] = fmaxf(v[i], v[i] * pre_slope);
    const float* wc = w + c * 7;                                // (k = 7, pad = 3: compile-time indices keep v[] in registers)
#pragma unroll
    for (int j = 0; j < 7; ++j) {                               // output t0 + o reads v[4 + o + j - 3]
      const float wj = wc[j];
      a0 = fmaf(wj, v[1 + j], a0); a1 = fmaf(wj, v[2 + j], a1); a2 = fmaf(wj, v[3 + j], a2); a3 = fmaf(wj, v[4 + j], a3);
    }
  }
  const float o[4] = {a0, a1, a2, a3};
#pragma unroll
  for (int i = 0; i < 4; ++i) if (t0 + i < T) y[t0 + i] = act_tanh ? tanhf(o[i]) : o[i];
}
void conv_to1(hipStream_t s, const float* x, long long ldx, const float* w, int Ci, int K, int pad, int T, float pre_slope, int act_tanh,
              float* y) {
  const bool x4 = K == 7 && pad == 3 && (ldx & 3) == 0 && ((uintptr_t)x & 15) == 0;
  if (x4) hipLaunchKernelGGL(conv_to1x4_kernel, dim3((T + 1023) / 1024), dim3(256), 0, s, x, ldx, w, Ci, K, pad, T, pre_slope, act_tanh, y);
  else hipLaunchKernelGGL(conv_to1_kernel, dim3((T + 255) / 256), dim3(256), 0, s, x, ldx, w, Ci, K, pad, T, pre_slope, act_tanh, y);
}

// ---------------------------------------------------------------------------------------------- rational resampling
// y[n] = sum_m x[m] h[m U - n D + half]: polyphase evaluation of a linear-phase low-pass designed on the U-times up-sampled grid
// (host: lib/audio.py::design_resample_filter, float64).  One thread per output sample, float64 accumulation; the input is taken
// as zero outside [0, n_in).  Stands in for librosa.resample(res_type="soxr_hq") at the two places the reference calls it
// (lib/audio.py:150 input -> 16 kHz, vc_infer_pipeline.py:186 output -> resample_sr); 480 k outputs x 550 taps: well under 1 ms.
__global__ __launch_bounds__(256) void resample_kernel(const float* __restrict__ x, long long n_in, const double* __restrict__ h, int half,
                                                       int U, int D, float* __restrict__ y, long long n_out) {
  const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= n_out) return;
  const long long c = n * D;                       // position of output n on the up-sampled grid
  long long lo = c - half, hi = c + half;
  long long m0 = lo <= 0 ? 0 : (lo + U - 1) / U;   // ceil(lo / U), clamped to the signal
  long long m1 = hi / U;                           // floor (hi >= 0)
  if (m1 > n_in - 1) m1 = n_in - 1;
  double acc = 0.0;
  for (long long m = m0; m <= m1; ++m) acc += (double)x[m] * h[m * U - c + half];
  y[n] = (float)acc;
}
void resample(hipStream_t s, const float* x, long long n_in, const double* h, int half, int U, int D, float* y, long long n_out) {
  if (n_out <= 0) return;
  hipLaunchKernelGGL(resample_kernel, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, s, x, n_in, h, half, U, D, y, n_out);
}

// ---------------------------------------------------------------------------------------------- transpose [R][C] -> [C][R]
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int C,
                                                        long long ldin, long long ldout, long long bin, long long bout) {
  __shared__ float tile[32][33];
  in += (long long)blockIdx.z * bin; out += (long long)blockIdx.z * bout;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8) { const int r = r0 + j, c = c0 + tx; tile[j][tx] = (r < R && c < C) ? in[(long long)r * ldin + c] : 0.f; }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) { const int c = c0 + j, r = r0 + tx; if (c < C && r < R) out[(long long)c * ldout + r] = tile[tx][j]; }
}
void transpose(hipStream_t s, const float* in, float* out, int R, int C, long long ldin, long long ldout, int batch, long long bin,
               long long bout) {
  hipLaunchKernelGGL(transpose_kernel, dim3((C + 31) / 32, (R + 31) / 32, batch), dim3(256), 0, s, in, out, R, C, ldin, ldout, bin, bout);
}

// features for the synthesizer: nearest x2 upsample of [D][Th] and the protect blend (reference vc_infer_pipeline.py:77-95)
// f0 = features before index retrieval (feats0 of the reference); null when no index is used (then feats0 == feats)
__global__ void feats_prepare_kernel(const float* __restrict__ f, const float* __restrict__ f0, const float* __restrict__ pitchf,
                                     float* __restrict__ out, int D, int Th, int T, float protect, int do_protect) {
  const long long n = (long long)D * T;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long st = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += st) {
    const int c = (int)(i / T); const int t = (int)(i - (long long)c * T);
    const float v = f[(long long)c * Th + (t >> 1)];
    float o = v;
    if (do_protect) {
      const float pf = pitchf[t];
      float w = pf;
      if (pf > 0.f) w = 1.f;
      if (pf < 1.f) w = protect;
      const float v0 = f0 ? f0[(long long)c * Th + (t >> 1)] : v;
      o = v * w + v0 * (1.f - w);
    }
    out[i] = o;
  }
}
void feats_prepare(hipStream_t s, const float* f, const float* f0, const float* pitchf, float* out, int D, int Th, int T, float protect,
                   int do_protect) {
  long long n = (long long)D * T; int blocks = (int)((n + 255) / 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(feats_prepare_kernel, dim3(blocks), dim3(256), 0, s, f, f0, pitchf, out, D, Th, T, protect, do_protect);
}

// ---------------------------------------------------------------------------------------------- im2col for single-channel convs
// out[j][t] = src[t*stride + j - pad]  (pad_mode 0: zeros, 1: reflect);  j < k, t < Tout
__global__ void frames_kernel(const float* __restrict__ src, float* __restrict__ out, int L, int k, int stride, int pad, int Tout,
                              int reflect) {
  const long long n = (long long)k * Tout;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long st = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += st) {
    const int j = (int)(i / Tout); const int t = (int)(i - (long long)j * Tout);
    long long x = (long long)t * stride + j - pad;
    float v = 0.f;
    if (reflect) { if (x < 0) x = -x; if (x >= L) x = 2LL * (L - 1) - x; v = src[x]; }
    else if (x >= 0 && x < L) v = src[x];
    out[i] = v;
  }
}
void frames(hipStream_t s, const float* src, float* out, int L, int k, int stride, int pad, int Tout, int reflect) {
  long long n = (long long)k * Tout; int blocks = (int)((n + 255) / 256); if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(frames_kernel, dim3(blocks), dim3(256), 0, s, src, out, L, k, stride, pad, Tout, reflect);
}

// ---------------------------------------------------------------------------------------------- NSF noise branch of the narrow generator stages
// x[c][t] += b[c] + sum_j w[c][j] src[t stride + j - pad]  (reference models.py GeneratorNSF.forward: x = ups(x) + noise_convs(har), Conv1d(1, C, 2 s, stride s,
// padding s / 2) resp. Conv1d(1, C, 1) in the last stage).  K <= 8 taps on ONE source channel: 2 K FLOP per 8 bytes moved - a streaming kernel (16-byte
// accesses of x, the source window of a quad from L1), where the im2col + fp32-MFMA GEMM ran at 0.30 - 0.41 of the HBM rate.
template <int K>
__global__ __launch_bounds__(256) void noise_add_kernel(float* __restrict__ x, long long ld, int C, int T, const float* __restrict__ src, long long L, int stride, int pad,
                                                        const float* __restrict__ w, const float* __restrict__ b) {
  // a thread owns 4 consecutive positions and walks 16 channels: the source window (4 K samples) is fetched once per 64 outputs, the
  // weights of a channel are wave-uniform (scalar loads)
  typedef float f32x4_t __attribute__((ext_vector_type(4)));
  const int c0 = blockIdx.y * 16;
  const int t4 = 4 * (blockIdx.x * 256 + threadIdx.x);
  if (t4 >= T) return;
  float sv[4][K];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const long long p0 = (long long)(t4 + e) * stride - pad;
#pragma unroll
    for (int j = 0; j < K; ++j) { const long long q = p0 + j; sv[e][j] = (q >= 0 && q < L) ? src[q] : 0.f; }
  }
  f32x4_t v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = (c0 + i < C) ? *reinterpret_cast<const f32x4_t*>(x + (long long)(c0 + i) * ld + t4) : f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = c0 + i;
    if (c < C) {
      const float bc = b[c];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a = bc;
#pragma unroll
        for (int j = 0; j < K; ++j) a = fmaf(w[c * K + j], sv[e][j], a);
        v[i][e] += a;
      }
      *reinterpret_cast<f32x4_t*>(x + (long long)c * ld + t4) = v[i];
    }
  }
}
bool noise_add(hipStream_t s, float* x, long long ld, int C, int T, const float* src, long long L, int k, int stride, int pad, const float* w, const float* b) {
  if ((T & 3) || (ld & 3) || (reinterpret_cast<uintptr_t>(x) & 15) || !(k == 1 || k == 4 || k == 8)) return false;
  const dim3 grid((unsigned)((T / 4 + 255) / 256), (unsigned)((C + 15) / 16));
  if (k == 1) hipLaunchKernelGGL(noise_add_kernel<1>, grid, dim3(256), 0, s, x, ld, C, T, src, L, stride, pad, w, b);
  else if (k == 4) hipLaunchKernelGGL(noise_add_kernel<4>, grid, dim3(256), 0, s, x, ld, C, T, src, L, stride, pad, w, b);
  else hipLaunchKernelGGL(noise_add_kernel<8>, grid, dim3(256), 0, s, x, ld, C, T, src, L, stride, pad, w, b);
  return true;
}

// ---------------------------------------------------------------------------------------------- pitch post-processing at 100 frames per second
// get_f0's tail (reference pitch_extraction.py / vc_infer_pipeline.py get_f0: f0 *= 2^(key / 12); mel = 2595 log10(1 + f0 / 700) scaled to 1 .. bins - 1,
// rint) in float64 like the numpy original: pitchf (Hz, float32) and the coarse pitch (int64) the synthesizer takes, without a host round trip
__global__ void f0_post_kernel(const double* __restrict__ f0, long long n, double factor, double mel_min, double mel_max, int bins,
                               long long* __restrict__ pitch, float* __restrict__ pitchf) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double f = f0[i] * factor;
  double m = (2595.0 * log10(1.0 + f / 700.0) - mel_min) * (double)(bins - 2) / (mel_max - mel_min) + 1.0;
  m = fmin(fmax(m, 1.0), (double)(bins - 1));
  pitch[i] = (long long)rint(m);                            // (round half to even, as np.rint)
  pitchf[i] = (float)f;
}
void f0_post(hipStream_t s, const double* f0, long long n, double factor, double mel_min, double mel_max, int bins, long long* pitch, float* pitchf) {
  hipLaunchKernelGGL(f0_post_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, f0, n, factor, mel_min, mel_max, bins, pitch, pitchf);
}

// |STFT|: mag[f][t] = sqrt(re^2 + im^2) with re = ft[f][t], im = ft[f + F][t]   (reference lib/rmvpe.py:143-147)
__global__ void magnitude_kernel(const float* __restrict__ ft, float* __restrict__ mag, int F, int T) {
  const long long n = (long long)F * T;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long st = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += st) { const float re = ft[i], im = ft[i + n]; mag[i] = sqrtf(re * re + im * im); }
}
void magnitude(hipStream_t s, const float* ft, float* mag, int F, int T) {
  long long n = (long long)F * T; int blocks = (int)((n + 255) / 256); if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(magnitude_kernel, dim3(blocks), dim3(256), 0, s, ft, mag, F, T);
}

// RMVPE U-Net input: x[t][m] = bn(logmel[m][reflect_right(t)])  for t < Tr  (reference lib/rmvpe.py:590-594,:302,:466)
__global__ __launch_bounds__(256) void mel_to_unet_kernel(const float* __restrict__ mel, float* __restrict__ x, int n, int Tr, float a, float b) {
  __shared__ float tile[32][33];
  const int t0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8) {
    int t = t0 + tx; float v = 0.f;
    if (t < Tr) { if (t >= n) t = 2 * (n - 1) - t; v = mel[(long long)(m0 + j) * n + t]; }
    tile[j][tx] = v;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) { const int t = t0 + j; if (t < Tr) x[(long long)t * 128 + m0 + tx] = tile[tx][j] * a + b; }
}
void mel_to_unet(hipStream_t s, const float* mel, float* x, int n, int Tr, float a, float b) {
  hipLaunchKernelGGL(mel_to_unet_kernel, dim3((Tr + 31) / 32, 4), dim3(256), 0, s, mel, x, n, Tr, a, b);
}

// AvgPool2d(2,2) on [C][H][W] -> [C][H/2][W/2]
__global__ void avgpool2_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int H, int W, long long ldx) {
  const int Ho = H / 2, Wo = W / 2;
  const long long n = (long long)C * Ho * Wo;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long st = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += st) {
    const int w = (int)(i % Wo); const long long r = i / Wo; const int h = (int)(r % Ho); const int c = (int)(r / Ho);
    const float* p = x + (long long)c * ldx + (long long)(2 * h) * W + 2 * w;
    y[i] = (p[0] + p[1] + p[W] + p[W + 1]) * 0.25f;
  }
}
void avgpool2(hipStream_t s, const float* x, float* y, int C, int H, int W, long long ldx) {
  long long n = (long long)C * (H / 2) * (W / 2); int blocks = (int)((n + 255) / 256); if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(avgpool2_kernel, dim3(blocks), dim3(256), 0, s, x, y, C, H, W, ldx);
}

// ---------------------------------------------------------------------------------------------- GRU recurrence (bidirectional)
// 16 workgroups: (direction, slice of 32 hidden units).  Each keeps its 96 x 256 block of W_hh in registers
// (8 threads per row, 32 weights each, 768 threads).  h_t is exchanged through 8-byte {step tag, value} granules written with
// agent-scope relaxed atomics (write-through) into a 2-deep ring and gathered by polling (bounded spin).
// Placement: workgroups are dealt round-robin to the 8 XCDs, so the launch is 64 workgroups of which those with
// blockIdx % 8 == 0 (direction 0) and == 4 (direction 1) work and the rest exit: the 8 slices of one direction then share one
// XCD (measured: 6.3 -> 5.0 ms for 3200 steps).  Agent-scope loads are correct for any placement; workgroup-scope (sc0) polling
// was tried for an L2-local hand-off and never observes the remote store (it is served from the CU's L1).
// gi: [T][1536] = W_ih x + (bias added here); out: channel-major [512][T].   nn.GRU gate order r, z, n.
#ifndef RVC_GRU_SLEEP
#define RVC_GRU_SLEEP 1      // back-off of the polling loop, in units of 64 cycles (measured: see DESIGN.md)
#endif
#ifdef RVC_EXPERIMENTS      // the round-3 scan kernel (three barriers per step): kept for A/B in variant builds (RVC_GRU_V=1)
__global__ __launch_bounds__(768) void gru_scan_kernel_v1(const float* __restrict__ gi, const float* __restrict__ b_ih,
                                                          const float* __restrict__ w_hh, const float* __restrict__ b_hh,
                                                          float* __restrict__ out, unsigned long long* xbuf, int* err, int T,
                                                          unsigned spin_limit, int fault) {
  constexpr int H = 256, HS = 32;
  constexpr int HP = 36;                                   // padded pitch of one 32-value segment of h in LDS (float4 reads of the
                                                           // eight segments then fall on disjoint banks)
  // h and the gate pre-activations are double-buffered by step parity: the gate threads of step s read buffer s & 1 while the other threads
  // already gather / multiply step s + 1 in the other one - no barrier at the end of a step; the data dependency orders the rest (nobody passes
  // the gather of step s + 1 before every slice, this one included, has published h_s).  Same-box A/B, 6201 steps: 6.33 -> 6.11 ms.
  __shared__ __attribute__((aligned(16))) float hs[2][8 * HP];
  __shared__ float ghs[2][96];
  const int xcd = blockIdx.x & 7;
  if (xcd != 0 && xcd != 4) return;
  const int dir = xcd >> 2, sl = blockIdx.x >> 3;
  if (fault && dir == 1 && sl == 5) return;                 // fault injection (tests): one slice never publishes, its peers must time out
  const int tid = threadIdx.x;
  bool failed = false;                                       // sticky per thread: after one timeout the stale value is used without polling
  const float* W = w_hh + (long long)dir * 3 * H * H;
  const float* BH = b_hh + dir * 3 * H;
  const float* BI = b_ih + dir * 3 * H;
  unsigned long long* xb = xbuf + dir * 2 * H;

  // weights of this thread: local row lr = tid >> 3 (gate g = lr / 32, unit jj = lr % 32), column segment seg = tid & 7 (32 columns).
  // Eight lanes per row keep the per-step dot product at 8 LDS reads + 32 FMAs per thread (with 2 threads per row the serial
  // read -> FMA chain was 1.0 us of a 1.5 us step).  Measured per step now (cycles): hand-off 1000, dot product 475, gates 200.
  float w[32];
  const int lr = tid >> 3, seg = tid & 7;
  {
    const int grow = (lr / HS) * H + sl * HS + (lr % HS);
#pragma unroll
    for (int c = 0; c < 32; ++c) w[c] = W[(long long)grow * H + seg * 32 + c];
  }
  float c_r = 0.f, c_z = 0.f, bi_n = 0.f, bh_n = 0.f;
  const int unit = sl * HS + tid;   // valid for tid < 32
  if (tid < HS) {
    c_r = BI[unit] + BH[unit]; c_z = BI[H + unit] + BH[H + unit];          // (b_ih + b_hh) of the r and z gates
    bi_n = BI[2 * H + unit]; bh_n = BH[2 * H + unit];
  }
  if (tid < H) hs[0][(tid >> 5) * HP + (tid & 31)] = 0.f;
  __syncthreads();
  for (int step = 0; step < T; ++step) {
    float* hsb = hs[step & 1]; float* gb = ghs[step & 1];
    const int t = dir ? (T - 1 - step) : step;
    float gr = 0.f, gz = 0.f, gn = 0.f;
    if (tid < HS) {
      const float* g = gi + (long long)t * (6 * H) + dir * 3 * H + unit;
      gr = g[0] + c_r; gz = g[H] + c_z; gn = g[2 * H] + bi_n;
    }
    if (step > 0) {
      if (tid < H) {
        // gather h_{step-1}: one granule per thread
        const unsigned long long* gp = xb + ((step - 1) & 1) * H + tid;
        unsigned long long v;
        unsigned spins = 0;
        for (;;) {
          v = __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((unsigned)(v >> 32) == (unsigned)step) break;
          if (failed || ++spins > spin_limit) { if (!failed && err) atomicExch(err, 1); failed = true; break; }
          __builtin_amdgcn_s_sleep(RVC_GRU_SLEEP);
        }
        hsb[(tid >> 5) * HP + (tid & 31)] = __uint_as_float((unsigned)v);
      }
      __syncthreads();
    }
    {
      const float4* hv = reinterpret_cast<const float4*>(hsb + seg * HP);
      float4 h4[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) h4[c] = hv[c];
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        a0 = fmaf(w[4 * c + 0], h4[c].x, a0); a1 = fmaf(w[4 * c + 1], h4[c].y, a1);
        a2 = fmaf(w[4 * c + 2], h4[c].z, a2); a3 = fmaf(w[4 * c + 3], h4[c].w, a3);
      }
      float a = (a0 + a1) + (a2 + a3);
      // sum over the 8 lanes of a row with DPP moves (VALU, no LDS crossbar round trips)
      a += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
      a += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
      a += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x141, 0xF, 0xF, true));   // row_half_mirror
      if (seg == 0) gb[lr] = a;
    }
    __syncthreads();
    if (tid < HS) {
      // gates on the hardware exp2 / reciprocal (1 ulp each): sigmoid(x) = 1 / (1 + 2^(-x log2 e)), tanh(x) = 1 - 2 / (2^(2 x log2 e) + 1).
      // The library expf / tanhf were 770 of the 2550 cycles of a step, on the critical path of all 16 workgroups.
      constexpr float kL2E = 1.44269504088896340736f;
      const float r = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-kL2E * (gr + gb[tid])));
      const float zg = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-kL2E * (gz + gb[HS + tid])));
      const float xn = gn + r * (gb[2 * HS + tid] + bh_n);
      const float nn = 1.f - 2.f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(2.f * kL2E * xn) + 1.f);
      const float hprev = hsb[(unit >> 5) * HP + (unit & 31)];
      const float hnew = (1.f - zg) * nn + zg * hprev;
      const unsigned long long gran = ((unsigned long long)(unsigned)(step + 1) << 32) | (unsigned long long)__float_as_uint(hnew);
      __hip_atomic_store(xb + (step & 1) * H + unit, gran, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      out[(long long)(dir * H + unit) * T + t] = hnew;
    }
  }
}
#endif
// Round 4: the same scan with ONE barrier per step.  The three gate rows (r, z, n) of a hidden unit used to live in three different waves:
// their dot products met in LDS (write, barrier, read) before 32 threads computed the gates.  Here a unit's eight lanes hold all three rows
// (96 weights per thread, 256 threads = one wave per SIMD: the same 384 FMA issue cycles per SIMD as twelve waves of 32), the DPP tree leaves
// the three sums in the unit's lane 0, which computes the gates at once - no second barrier, no LDS round trip.  Segments, FMA chains and the
// reduction tree are those of gru_scan_kernel_v1: identical numerics.  (RVC_GRU_V=1 selects the old kernel.)
// NP polls in flight per thread (round 5).  With one, a granule that lands just after a poll was issued is seen a whole L2 round trip + back-off
// later, and a step waits for the LATEST of seven peers: close to a full poll period on top of the hand-off.  With NP staggered loads outstanding
// (each re-issued the moment it returns un-tagged) the detection delay falls to a period / NP; the loads retire in order, so checking the oldest
// needs vmcnt(NP - 1) only - the compiler's own count.
template <int NP>
__device__ __forceinline__ unsigned long long gru_poll(const unsigned long long* gp, unsigned step, unsigned spin_limit, bool& failed, int* err) {
  unsigned long long q[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    q[i] = __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (i + 1 < NP) __builtin_amdgcn_s_sleep(1);
  }
  unsigned spins = 0;
  for (;;) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      if ((unsigned)(q[i] >> 32) == step) return q[i];
      if (failed || ++spins > spin_limit * (unsigned)NP) { if (!failed && err) atomicExch(err, 1); failed = true; return q[i]; }      // (spins counts per outstanding load: NP of them per poll period)
      q[i] = __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
template <int NP, int SL = RVC_GRU_SLEEP>
__global__ __launch_bounds__(256) void gru_scan_kernel(const float* __restrict__ gi, const float* __restrict__ b_ih,
                                                       const float* __restrict__ w_hh, const float* __restrict__ b_hh,
                                                       float* __restrict__ out, unsigned long long* xbuf, int* err, int T,
                                                       unsigned spin_limit, int fault) {
  constexpr int H = 256, HS = 32, HP = 36;
  __shared__ __attribute__((aligned(16))) float hs[2][8 * HP];
  const int xcd = blockIdx.x & 7;
  if (xcd != 0 && xcd != 4) return;
  const int dir = xcd >> 2, sl = blockIdx.x >> 3;
  if (fault && dir == 1 && sl == 5) return;                 // fault injection (tests): one slice never publishes, its peers must time out
  const int tid = threadIdx.x;
  bool failed = false;
  const float* W = w_hh + (long long)dir * 3 * H * H;
  const float* BH = b_hh + dir * 3 * H;
  const float* BI = b_ih + dir * 3 * H;
  unsigned long long* xb = xbuf + dir * 2 * H;
  const int jj = tid >> 3, seg = tid & 7, unit = sl * HS + jj;
  float w[3][32];
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int c = 0; c < 32; ++c) w[g][c] = W[(long long)(g * H + unit) * H + seg * 32 + c];
  float c_r = 0.f, c_z = 0.f, bi_n = 0.f, bh_n = 0.f;
  if (seg == 0) { c_r = BI[unit] + BH[unit]; c_z = BI[H + unit] + BH[H + unit]; bi_n = BI[2 * H + unit]; bh_n = BH[2 * H + unit]; }
  hs[0][(tid >> 5) * HP + (tid & 31)] = 0.f;
  __syncthreads();
  for (int step = 0; step < T; ++step) {
    float* hsb = hs[step & 1];
    const int t = dir ? (T - 1 - step) : step;
    float gr = 0.f, gz = 0.f, gn = 0.f;
    if (seg == 0) {
      const float* g = gi + (long long)t * (6 * H) + dir * 3 * H + unit;
      gr = g[0] + c_r; gz = g[H] + c_z; gn = g[2 * H] + bi_n;
    }
    if (step > 0) {
      // gather h_{step-1}: one granule per thread
      const unsigned long long* gp = xb + ((step - 1) & 1) * H + tid;
      unsigned long long v;
      if constexpr (NP > 1) {
        v = gru_poll<NP>(gp, (unsigned)step, spin_limit, failed, err);
      } else {
        unsigned spins = 0;
        for (;;) {
          v = __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((unsigned)(v >> 32) == (unsigned)step) break;
          if (failed || ++spins > spin_limit) { if (!failed && err) atomicExch(err, 1); failed = true; break; }
          __builtin_amdgcn_s_sleep(SL);
        }
      }
      hsb[(tid >> 5) * HP + (tid & 31)] = __uint_as_float((unsigned)v);
      __syncthreads();
    }
    float a[3];
    {
      const float4* hv = reinterpret_cast<const float4*>(hsb + seg * HP);
      float4 h4[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) h4[c] = hv[c];
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          a0 = fmaf(w[g][4 * c + 0], h4[c].x, a0); a1 = fmaf(w[g][4 * c + 1], h4[c].y, a1);
          a2 = fmaf(w[g][4 * c + 2], h4[c].z, a2); a3 = fmaf(w[g][4 * c + 3], h4[c].w, a3);
        }
        float x = (a0 + a1) + (a2 + a3);
        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));   // row_half_mirror
        a[g] = x;
      }
    }
    if (seg == 0) {
      constexpr float kL2E = 1.44269504088896340736f;
      const float r = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-kL2E * (gr + a[0])));
      const float zg = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-kL2E * (gz + a[1])));
      const float xn = gn + r * (a[2] + bh_n);
      const float nn = 1.f - 2.f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(2.f * kL2E * xn) + 1.f);
      const float hprev = hsb[(unit >> 5) * HP + (unit & 31)];
      const float hnew = (1.f - zg) * nn + zg * hprev;
      const unsigned long long gran = ((unsigned long long)(unsigned)(step + 1) << 32) | (unsigned long long)__float_as_uint(hnew);
      __hip_atomic_store(xb + (step & 1) * H + unit, gran, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      out[(long long)(dir * H + unit) * T + t] = hnew;
    }
  }
}

// Guaranteed outcome (round 4).  The scan above needs its 16 working workgroups resident at the same time; the bounded spin detects the case that
// they are not, it cannot repair it.  This kernel can: ONE workgroup per direction computes the same recurrence with no inter-workgroup traffic at
// all (768 threads = 768 rows of W_hh, read k-major from L2 - 768 KiB per step and direction, ~6 us per step instead of ~1), so it terminates
// whatever else shares the GPU.  It is enqueued behind every scan and returns at once unless the scan raised its time-out flag (err[0]); then it
// rewrites `out` completely and counts itself in err[1] (2 = both directions repaired: rmvpe_decode / rvc_rmvpe_status treat the forward as
// good).  The products are summed in exactly the order of the fast kernel (eight 32-column segments: four FMA chains, (a0 + a1) + (a2 + a3), then
// the pairwise tree of the DPP reduction), the gates use the same hardware exp2 / rcp forms: measured, the repaired hidden states agree with
// a healthy run of the fast kernel to 2.4e-7 (rounding of the compiler's instruction selection; the f0 to 1e-6 relative).  No host round trip, no
// re-exec: a fresh launch on the same stream.
__global__ __launch_bounds__(768) void gru_serial_kernel(const float* __restrict__ gi, const float* __restrict__ b_ih, const float* __restrict__ w_hh_t,
                                                         const float* __restrict__ b_hh, float* __restrict__ out, int* err, int T) {
  constexpr int H = 256;
  if (*reinterpret_cast<volatile int*>(err) == 0) return;
  __shared__ float hs[H];
  __shared__ float ghs[3 * H];
  const int dir = blockIdx.x, tid = threadIdx.x;
  const float* WT = w_hh_t + (long long)dir * 3 * H * H;     // [column][row]: thread `tid` (= row) reads consecutive addresses across the wave
  const float* BH = b_hh + dir * 3 * H;
  const float* BI = b_ih + dir * 3 * H;
  float c_r = 0.f, c_z = 0.f, bi_n = 0.f, bh_n = 0.f;
  if (tid < H) { c_r = BI[tid] + BH[tid]; c_z = BI[H + tid] + BH[H + tid]; bi_n = BI[2 * H + tid]; bh_n = BH[2 * H + tid]; hs[tid] = 0.f; }
  __syncthreads();
  for (int step = 0; step < T; ++step) {
    const int t = dir ? (T - 1 - step) : step;
    float sg[8];
#pragma unroll
    for (int seg = 0; seg < 8; ++seg) {
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int k = seg * 32 + 4 * c;
        a0 = fmaf(WT[(long long)(k + 0) * (3 * H) + tid], hs[k + 0], a0); a1 = fmaf(WT[(long long)(k + 1) * (3 * H) + tid], hs[k + 1], a1);
        a2 = fmaf(WT[(long long)(k + 2) * (3 * H) + tid], hs[k + 2], a2); a3 = fmaf(WT[(long long)(k + 3) * (3 * H) + tid], hs[k + 3], a3);
      }
      sg[seg] = (a0 + a1) + (a2 + a3);
    }
    ghs[tid] = ((sg[0] + sg[1]) + (sg[2] + sg[3])) + ((sg[7] + sg[6]) + (sg[5] + sg[4]));      // the DPP tree of gru_scan_kernel, lane 0's view
    __syncthreads();
    if (tid < H) {
      constexpr float kL2E = 1.44269504088896340736f;
      const float* g = gi + (long long)t * (6 * H) + dir * 3 * H + tid;
      const float gr = g[0] + c_r, gz = g[H] + c_z, gn = g[2 * H] + bi_n;
      const float r = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-kL2E * (gr + ghs[tid])));
      const float zg = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-kL2E * (gz + ghs[H + tid])));
      const float xn = gn + r * (ghs[2 * H + tid] + bh_n);
      const float nn = 1.f - 2.f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(2.f * kL2E * xn) + 1.f);
      const float hnew = (1.f - zg) * nn + zg * hs[tid];
      out[(long long)(dir * H + tid) * T + t] = hnew;
      hs[tid] = hnew;                                           // (every dot product of this step is done: nobody else reads hs before the barrier)
    }
    __syncthreads();
  }
  if (tid == 0) { __threadfence(); atomicAdd(err + 1, 1); }
}

// *err is a sticky device flag: set when a workgroup gave up waiting for its peers (they need co-residency: 16 workgroups of 768
// threads; a spin limit of 2^24 polls is ~10 s).  It is cleared here, consumed by rmvpe_decode (f0 becomes NaN) and reported by
// rvc_rmvpe_status.
void gru_scan(hipStream_t s, const float* gi, const float* b_ih, const float* w_hh, const float* w_hh_t, const float* b_hh, float* out,
              unsigned long long* xbuf, int* err, int T, unsigned spin_limit, int fault) {
  (void)hipMemsetAsync(xbuf, 0, sizeof(unsigned long long) * 2 * 2 * 256, s);
  (void)hipMemsetAsync(err, 0, 2 * sizeof(int), s);
  unsigned sl = spin_limit ? spin_limit : (1u << 24);
#ifdef RVC_EXPERIMENTS
  // variant builds only (tools/build_variant.sh): cooperative launch (RVC_GRU_COOP=1: 1793 -> 1621 xRT with three clips in flight, round 3 - a cooperative dispatch
  // waits until the grid is launchable as a whole, which idles the chip under the other lanes' kernels), the round-3 scan kernel (RVC_GRU_V=1), several polls in
  // flight per thread (RVC_GRU_POLL) and longer poll back-off (RVC_GRU_SLEEPN) - all measured neutral or worse (profiles/r5_gru_poll.txt, r5_exp_gru_sleep.txt)
  static const bool coop = (exp_int("RVC_GRU_COOP", 0) != 0);
  static const int ver = exp_int("RVC_GRU_V", 2);
  static const int np = exp_int("RVC_GRU_POLL", 1);
  static const int slp = exp_int("RVC_GRU_SLEEPN", RVC_GRU_SLEEP);
  if (coop) {
    void* args[] = {(void*)&gi, (void*)&b_ih, (void*)&w_hh, (void*)&b_hh, (void*)&out, (void*)&xbuf, (void*)&err, (void*)&T, (void*)&sl, (void*)&fault};
    if (ver == 1) RVC_HIP_CHECK(hipLaunchCooperativeKernel(reinterpret_cast<const void*>(gru_scan_kernel_v1), dim3(64), dim3(768), args, 0, s));
    else RVC_HIP_CHECK(hipLaunchCooperativeKernel(reinterpret_cast<const void*>(gru_scan_kernel<1>), dim3(64), dim3(256), args, 0, s));
  } else if (ver == 1) hipLaunchKernelGGL(gru_scan_kernel_v1, dim3(64), dim3(768), 0, s, gi, b_ih, w_hh, b_hh, out, xbuf, err, T, sl, fault);
  else if (np == 2) hipLaunchKernelGGL(gru_scan_kernel<2>, dim3(64), dim3(256), 0, s, gi, b_ih, w_hh, b_hh, out, xbuf, err, T, sl, fault);
  else if (np == 3) hipLaunchKernelGGL(gru_scan_kernel<3>, dim3(64), dim3(256), 0, s, gi, b_ih, w_hh, b_hh, out, xbuf, err, T, sl, fault);
  else if (np >= 4) hipLaunchKernelGGL(gru_scan_kernel<4>, dim3(64), dim3(256), 0, s, gi, b_ih, w_hh, b_hh, out, xbuf, err, T, sl, fault);
  else if (slp >= 16) hipLaunchKernelGGL((gru_scan_kernel<1, 16>), dim3(64), dim3(256), 0, s, gi, b_ih, w_hh, b_hh, out, xbuf, err, T, sl, fault);
  else if (slp >= 8) hipLaunchKernelGGL((gru_scan_kernel<1, 8>), dim3(64), dim3(256), 0, s, gi, b_ih, w_hh, b_hh, out, xbuf, err, T, sl, fault);
  else if (slp >= 4) hipLaunchKernelGGL((gru_scan_kernel<1, 4>), dim3(64), dim3(256), 0, s, gi, b_ih, w_hh, b_hh, out, xbuf, err, T, sl, fault);
  else
#endif
  // plain launch: the 16 working slices become resident as soon as ANY 16 CUs have free waves (every other kernel of the path terminates without waiting for
  // anything); the bounded spin detects the case that they do not, and the serial kernel behind it repairs it
  hipLaunchKernelGGL((gru_scan_kernel<1>), dim3(64), dim3(256), 0, s, gi, b_ih, w_hh, b_hh, out, xbuf, err, T, sl, fault);
  static const bool repair = (exp_int("RVC_GRU_REPAIR", 1) != 0);
  if (repair && w_hh_t) hipLaunchKernelGGL(gru_serial_kernel, dim3(2), dim3(768), 0, s, gi, b_ih, w_hh_t, b_hh, out, err, T);
}

// ---------------------------------------------------------------------------------------------- RMVPE decode
// salience channel-major [360][ld]; f0[t] = 10 * 2^(cents/1200) with the 9-bin local weighted average around the argmax
// (reference lib/rmvpe.py:607-612,:661-685); 0 where max <= thred.  Output float64 like the numpy reference.
__global__ void rmvpe_decode_kernel(const float* __restrict__ sal, double* __restrict__ f0, int n, long long ld, float thred, const int* __restrict__ err) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  if (err && err[0] && err[1] < 2) { f0[t] = __longlong_as_double(0x7ff8000000000000LL); return; }     // the recurrence failed upstream and was not repaired: never hand out a plausible-looking pitch
  float mx = -1.f; int am = 0;
  for (int c = 0; c < 360; ++c) { const float v = sal[(long long)c * ld + t]; if (v > mx) { mx = v; am = c; } }
  double ps = 0.0, ws = 0.0;
  for (int d = -4; d <= 4; ++d) {
    const int c = am + d;
    if (c < 0 || c >= 360) continue;   // zero padding contributes nothing
    const float v = sal[(long long)c * ld + t];
    ps += (double)v * (20.0 * (double)c + 1997.3794084376191);
    ws += (double)v;
  }
  double cents = ps / ws;
  if (mx <= thred) cents = 0.0;
  double f = 10.0 * exp2(cents / 1200.0);
  if (f == 10.0) f = 0.0;
  f0[t] = f;
}
void rmvpe_decode(hipStream_t s, const float* sal, double* f0, int n, long long ld, float thred, const int* err) {
  hipLaunchKernelGGL(rmvpe_decode_kernel, dim3((n + 127) / 128), dim3(128), 0, s, sal, f0, n, ld, thred, err);
}

// ---------------------------------------------------------------------------------------------- SineGen / SourceModuleHnNSF
// (reference lib/infer_pack/models.py:361-411,:455-467, harmonic_num = 0).  fp64 running sums restate torch.cumsum on CPU.
// Stage 1 (one block): rad[t] = (f0/sr) % 1; tmp[t] = float(cumsum_fp64(rad)) * upp
__global__ __launch_bounds__(1024) void sine_frame_kernel(const float* __restrict__ f0, float* __restrict__ rad, float* __restrict__ tmp,
                                                          int T, float sr, float upp) {
  __shared__ double part[1024];
  const int tid = threadIdx.x;
  const int per = (T + 1023) / 1024;
  const int b = tid * per, e = min(T, b + per);
  double s = 0.0;
  for (int t = b; t < e; ++t) { const float r = fmodf(f0[t] / sr, 1.f); rad[t] = r; s += (double)r; }
  part[tid] = s;
  __syncthreads();
  if (tid == 0) { double run = 0.0; for (int i = 0; i < 1024; ++i) { const double v = part[i]; part[i] = run; run += v; } }
  __syncthreads();
  double run = part[tid];
  for (int t = b; t < e; ++t) { run += (double)rad[t]; tmp[t] = (float)run * upp; }
}

__device__ __forceinline__ float sine_interp(const float* __restrict__ tmp, int T, float scale, long long i) {
  // torch's CPU upsample evaluates, each step rounded to float32 on its own:
  //   src = scale * i;  l1 = src - floor(src);  l0 = 1 - l1;  v = fma(l0, in0, fl(l1 * in1))
  // On flat (unvoiced) stretches the last-bit jitter of this expression is what trips the reference's wrap detector, and on a
  // 40 s segment thousands of its decisions change if `src - i0` is taken from the unrounded product, so the exact rounding
  // sequence is part of the algorithm.  hipcc contracts a * b - c into an fma even through __fmul_rn / __fsub_rn (they are plain
  // operators in the HIP headers): contraction is switched off for this function and the product is pinned in a register.
#pragma clang fp contract(off)
  float src = scale * (float)i;
  asm volatile("" : "+v"(src));
  int i0 = (int)src;
  const int i1 = i0 + (i0 < T - 1 ? 1 : 0);
  float l1 = src - (float)i0; l1 = fminf(fmaxf(l1, 0.f), 1.f);
  const float l0 = 1.f - l1;
  float p1 = l1 * tmp[i1];
  asm volatile("" : "+v"(p1));
  const float v = __builtin_fmaf(l0, tmp[i0], p1);
  return fmodf(v, 1.f);
}
__device__ __forceinline__ float sine_incr(const float* __restrict__ rad, const float* __restrict__ tmp, int T, int upp, float scale, long long i) {
  float v = rad[i / upp];
  if (i > 0) { if (sine_interp(tmp, T, scale, i) - sine_interp(tmp, T, scale, i - 1) < 0.f) v += -1.f; }
  return v;
}
// Stage 2: per-block (1024 samples) sums of the phase increments
__global__ __launch_bounds__(256) void sine_blocksum_kernel(const float* __restrict__ rad, const float* __restrict__ tmp, double* __restrict__ bsum,
                                                            int T, int upp, float scale, long long N) {
  __shared__ double red[4];
  const long long base = (long long)blockIdx.x * 1024;
  double s = 0.0;
  for (int j = 0; j < 4; ++j) { const long long i = base + threadIdx.x * 4 + j; if (i < N) s += (double)sine_incr(rad, tmp, T, upp, scale, i); }
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) bsum[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// Stage 3 (one block): exclusive scan of the block sums
__global__ __launch_bounds__(1024) void sine_scan_kernel(double* __restrict__ bsum, int nb) {
  __shared__ double part[1024];
  const int tid = threadIdx.x;
  const int per = (nb + 1023) / 1024;
  const int b = tid * per, e = min(nb, b + per);
  double s = 0.0;
  for (int i = b; i < e; ++i) s += bsum[i];
  part[tid] = s;
  __syncthreads();
  if (tid == 0) { double run = 0.0; for (int i = 0; i < 1024; ++i) { const double v = part[i]; part[i] = run; run += v; } }
  __syncthreads();
  double run = part[tid];
  for (int i = b; i < e; ++i) { const double v = bsum[i]; bsum[i] = run; run += v; }
}
// Stage 4: in-block scan + sin + uv gating + noise + Linear(1,1) + tanh -> har[i]
__global__ __launch_bounds__(256) void sine_final_kernel(const float* __restrict__ f0, const float* __restrict__ rad, const float* __restrict__ tmp,
                                                         const double* __restrict__ bsum, const float* __restrict__ noise, float* __restrict__ har,
                                                         float* __restrict__ sine_out, float* __restrict__ phase_out, int T, int upp, float scale, long long N, float lw, float lb) {
  __shared__ double wsum[4];
  const long long base = (long long)blockIdx.x * 1024;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double v[4]; double loc = 0.0;
  for (int j = 0; j < 4; ++j) { const long long i = base + tid * 4 + j; v[j] = (i < N) ? (double)sine_incr(rad, tmp, T, upp, scale, i) : 0.0; loc += v[j]; }
  // inclusive scan of `loc` across the wave
  double inc = loc;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const double nbr = __shfl_up(inc, o); if (lane >= o) inc += nbr; }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  double off = bsum[blockIdx.x];
  for (int w = 0; w < wave; ++w) off += wsum[w];
  double run = off + (inc - loc);
  for (int j = 0; j < 4; ++j) {
    const long long i = base + tid * 4 + j;
    run += v[j];
    if (i < N) {
      // The reference evaluates sin(fl32(fl32(c) * 2 * pi)) with |c| reaching several hundred cycles, so the fp32 rounding
      // of the argument is part of its result; reproduce that rounding exactly and take the sine itself in fp64.
      const float ph = (float)run;
      if (phase_out) phase_out[i] = ph;
      const float arg = ph * 2.f * 3.14159265358979323846f;
      float sw = (float)sin((double)arg) * 0.1f;
      const float uv = f0[i / upp] > 0.f ? 1.f : 0.f;
      const float namp = uv * 0.003f + (1.f - uv) * 0.1f / 3.f;
      sw = sw * uv + namp * noise[i];
      if (sine_out) sine_out[i] = sw;
      har[i] = tanhf(sw * lw + lb);
    }
  }
}
void sine_source(hipStream_t s, const float* f0, const float* noise, float* har, float* sine_out, float* rad, float* tmp, double* bsum,
                 int T, int upp, float sr, float lw, float lb, float* phase_out) {
  const long long N = (long long)T * upp;
  const int nb = (int)((N + 1023) / 1024);
  const float scale = N > 1 ? (float)(T - 1) / (float)(N - 1) : 0.f;
  hipLaunchKernelGGL(sine_frame_kernel, dim3(1), dim3(1024), 0, s, f0, rad, tmp, T, sr, (float)upp);
  hipLaunchKernelGGL(sine_blocksum_kernel, dim3(nb), dim3(256), 0, s, rad, tmp, bsum, T, upp, scale, N);
  hipLaunchKernelGGL(sine_scan_kernel, dim3(1), dim3(1024), 0, s, bsum, nb);
  hipLaunchKernelGGL(sine_final_kernel, dim3(nb), dim3(256), 0, s, f0, rad, tmp, bsum, noise, har, sine_out, phase_out, T, upp, scale, N, lw, lb);
}

}  // namespace rvc

namespace rvc {
// ---------------------------------------------------------------------------------------------- output post-processing on device
// change_rms (reference lib/model_utils.py:39-57) + peak normalisation to int16 (reference vc_infer_pipeline.py:188-189).
// RMS frames of the output: librosa.feature.rms semantics (zero centre-padding, frame = sr, hop = sr / 2), float32 result.
__global__ __launch_bounds__(256) void rms_frames_kernel(const float* __restrict__ x, float* __restrict__ rms, long long N, int frame, int hop) {
  __shared__ double red[4];
  const long long start = (long long)blockIdx.x * hop - frame / 2;
  double s = 0.0;
  for (int i = threadIdx.x; i < frame; i += 256) {
    const long long j = start + i;
    if (j >= 0 && j < N) { const float v = x[j]; s += (double)(v * v); }
  }
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) rms[blockIdx.x] = sqrtf((float)((red[0] + red[1] + red[2] + red[3]) / (double)frame));
}
// F.interpolate(mode="linear", align_corners=False) source coordinate and weights
template <typename T> __device__ __forceinline__ T interp_linear(const T* __restrict__ a, int n, long long i, T scale) {
  T src = scale * ((T)i + (T)0.5) - (T)0.5;
  if (src < (T)0) src = (T)0;
  int i0 = (int)src; if (i0 > n - 1) i0 = n - 1;
  const int i1 = i0 + (i0 < n - 1 ? 1 : 0);
  const T l1 = src - (T)i0, l0 = (T)1 - l1;
  return l0 * a[i0] + l1 * a[i1];
}
__global__ void rms_mix_absmax_kernel(float* __restrict__ x, long long N, const double* __restrict__ rms1, int n1, const float* __restrict__ rms2, int n2,
                                      double p1, double p2, int do_mix, unsigned* __restrict__ maxbits) {
  float mx = 0.f;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long st = (long long)gridDim.x * blockDim.x;
  const double s1 = (double)n1 / (double)N; const float s2 = (float)n2 / (float)N;
  for (; i < N; i += st) {
    float v = x[i];
    if (do_mix) {
      const double r1 = interp_linear<double>(rms1, n1, i, s1);
      const float r2 = fmaxf(interp_linear<float>(rms2, n2, i, s2), 1e-6f);
      // r1^p1 as 2^(p1 log2 r1) in float64 (1e-15 relative against pow(), far below the float32 product it feeds; pow() itself was most of this kernel's time)
      v = (float)((double)v * (exp2(p1 * log2(r1)) * (double)powf(r2, (float)p2)));
      x[i] = v;
    }
    mx = fmaxf(mx, fabsf(v));
  }
  // one atomic per workgroup (a wave-level atomicMax on the one word serialised ~19 k atomics per clip: 2/3 of this kernel's time)
  __shared__ float wmax[16];
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    float m = wmax[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) m = fmaxf(m, wmax[w]);
    atomicMax(maxbits, __float_as_uint(m));
  }
}
__global__ void to_int16_kernel(const float* __restrict__ x, short* __restrict__ y, long long N, const unsigned* __restrict__ maxbits) {
  const float amax = __uint_as_float(*maxbits) / 0.99f;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long st = (long long)gridDim.x * blockDim.x;
  for (; i < N; i += st) y[i] = (short)(x[i] * 32768.f / amax);     // C conversion truncates toward zero like ndarray.astype(int16)
}
void postprocess(hipStream_t s, float* x, long long N, const double* rms1, int n1, int sr2, float rate, short* out, float* rms2, unsigned* maxbits) {
  const int frame = sr2 / 2 * 2, hop = sr2 / 2;
  const int n2 = (int)(N / hop) + 1;
  const int do_mix = rate < 1.f && rms1 != nullptr;
  (void)hipMemsetAsync(maxbits, 0, sizeof(unsigned), s);
  if (do_mix) hipLaunchKernelGGL(rms_frames_kernel, dim3(n2), dim3(256), 0, s, x, rms2, N, frame, hop);
  int blocks = (int)((N + 255) / 256); if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(rms_mix_absmax_kernel, dim3(blocks), dim3(256), 0, s, x, N, rms1, n1, rms2, n2, (double)(1.f - rate), (double)(rate - 1.f), do_mix, maxbits);
  hipLaunchKernelGGL(to_int16_kernel, dim3(blocks), dim3(256), 0, s, x, out, N, maxbits);
}
}  // namespace rvc

// ================================================================================================ input pre-processing
// Zero-phase IIR high-pass of VC.pipeline (vc_infer_pipeline.py:121, scipy.signal.filtfilt(bh, ah, audio): odd extension by
// padlen = 3 * max(len(a), len(b)) samples, lfilter_zi initial conditions, direct form II transposed in float64, forward then
// backward), the reflect padding of :141 with the float32 cast the networks see, and the 0.5 s-hop RMS frames of the filtered
// signal that change_rms needs (lib/model_utils.py:45).
// The recurrence is sequential, but the filter forgets: a run that starts W samples early from a zero state differs from the true
// response by at most sum_{k >= W} |h_k| * max|x| (h = impulse response) = 2.3e-13 for W = 5120, so every thread filters W warm-up
// samples + its own L-sample chunk (overlap-discard); the first chunks start at sample 0 with filtfilt's lfilter_zi initial
// conditions.  float64, same direct-form-II-transposed update as scipy.  This 5th-order high-pass at 0.006 Nyquist amplifies
// rounding noise ~3e8 x: two float64 evaluations that differ only in rounding order (scipy's C loop and a literal Python
// transcription of it) already differ by 4e-8 of full scale, and so does this one - about one float32 ulp of the signal the
// networks consume.  Each lane streams its own window with 64-byte loads issued eight steps ahead of their use.
namespace rvc {

constexpr int kIirOrder = 5;
struct IirArgs {
  double b[kIirOrder + 1], a[kIirOrder + 1], zi[kIirOrder];
  const void* x; int is64;        // input samples (float32 or float64)
  long long n, N, Np; int padlen; // input length, extended length n + 2 padlen, Np = N rounded up to 8
  double* ext;                    // odd-extended input [Np]
  double* yr;                     // forward output, time-reversed [Np]: yr[N - 1 - j] = y_fwd[j]
  double* filt;                   // final output [n]
  int L, W, nchunks;
  // block-propagated evaluation (iir_block_kernel) of the SAME filter as a cascade of second-order sections: sos[k] = {b0, b1, b2, 1, a1, a2},
  // state 2 per section; M = state transition over one block of L samples; zero-state final states; true initial states; zis = sosfilt_zi
  double sos[3][6], zis[6];
  double M[6 * 6], Mg[6 * 6];       // state transition over one block / over the g blocks a lane of the scan owns
  double* Z0; double* Zin; int nb;
};

// odd extension, evaluated in the input's own precision like scipy's odd_ext; zero fill up to Np
__global__ void iir_extend_kernel(const IirArgs p) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.Np) return;
  const long long n = p.n; const int pl = p.padlen;
  double v = 0.0;
  if (i < p.N) {
    if (p.is64) {
      const double* x = (const double*)p.x;
      v = i < pl ? 2.0 * x[0] - x[pl - i] : (i >= n + pl ? 2.0 * x[n - 1] - x[n - 2 - (i - (n + pl))] : x[i - pl]);
    } else {
      const float* x = (const float*)p.x;
      v = i < pl ? (double)__fsub_rn(2.f * x[0], x[pl - i]) : (i >= n + pl ? (double)__fsub_rn(2.f * x[n - 1], x[n - 2 - (i - (n + pl))]) : (double)x[i - pl]);
    }
  }
  p.ext[i] = v;
  if (i >= p.N) p.yr[i] = 0.0;
}

__device__ __forceinline__ double iir_step(const IirArgs& p, double (&z)[kIirOrder], double x) {
  const double y = fma(p.b[0], x, z[0]);
#pragma unroll
  for (int i = 0; i < kIirOrder - 1; ++i) z[i] = fma(-p.a[i + 1], y, fma(p.b[i + 1], x, z[i + 1]));
  z[kIirOrder - 1] = fma(-p.a[kIirOrder], y, p.b[kIirOrder] * x);
  return y;
}

// DIR 0: forward over ext -> yr (reversed).  DIR 1: forward over yr (= backward in time) -> filt (un-reversed, extension dropped).
template <int DIR>
__global__ void iir_chunk_kernel(const IirArgs p) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= p.nchunks) return;
  const double* __restrict__ src = DIR == 0 ? p.ext : p.yr;
  const long long j0 = (long long)c * p.L, j1 = min(j0 + p.L, p.Np);
  long long js = j0 - p.W;
  double z[kIirOrder];
  if (js <= 0) {
    js = 0;
    const double x0 = src[0];
#pragma unroll
    for (int i = 0; i < kIirOrder; ++i) z[i] = p.zi[i] * x0;
  } else {
#pragma unroll
    for (int i = 0; i < kIirOrder; ++i) z[i] = 0.0;
  }
  // js, j0, j1 are multiples of 8: groups of eight samples, the next group in flight while the current one is filtered
  double2 cur[4], nxt[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) cur[q] = *reinterpret_cast<const double2*>(src + js + 2 * q);
  for (long long j = js; j < j1; j += 8) {
    const long long jn = j + 8 < j1 ? j + 8 : j;
#pragma unroll
    for (int q = 0; q < 4; ++q) nxt[q] = *reinterpret_cast<const double2*>(src + jn + 2 * q);
    double y[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) { y[2 * q] = iir_step(p, z, cur[q].x); y[2 * q + 1] = iir_step(p, z, cur[q].y); }
    if (j >= j0) {
      if (DIR == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {      // yr[N - 1 - (j + e)] = y[e]; out-of-range slots (j + e >= N) fall below index 0: skip
          const long long e0 = p.N - 1 - (j + 2 * q);
          if (e0 >= 0) p.yr[e0] = y[2 * q];
          if (e0 - 1 >= 0) p.yr[e0 - 1] = y[2 * q + 1];
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) { const long long i = p.N - 1 - (j + e) - p.padlen; if (i >= 0 && i < p.n) p.filt[i] = y[e]; }
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) cur[q] = nxt[q];
  }
}

// ---- the same filter, block-propagated (round 3).  The overlap-discard kernel above runs W + L = 5632 recurrence steps per 512 outputs and
// walks its window with per-lane strided loads (one 64-byte group in flight): 330 us per direction, latency-bound, at the very start of
// every conversion.  The filter is linear, so the state at a block boundary is  z_{c+1} = Z0_c + M z_c  with Z0_c the final state of block
// c run from a ZERO state and M the state transition over L samples:
//   pass 1 (parallel over blocks of L samples): Z0_c;   pass 2 (one thread, nb steps of a 6 x 6 product): the true z_c;
//   pass 3 (parallel): the block again, from z_c, writing the outputs.
// That needs a well-conditioned state: in the transfer-function form scipy.signal.filtfilt(b, a) runs (direct form II transposed, 5 states)
// M has entries of 1e7 that cancel to O(1) - the same ill-conditioning that makes that form amplify float64 rounding 3e8 x - and the
// propagation loses everything.  So the filter is evaluated as the cascade of its second-order sections (scipy.signal.butter(output="sos"),
// initial state sosfilt_zi * x[0]: what scipy.signal.sosfiltfilt does; identical to filtfilt(b, a) in exact arithmetic): states and M are
// O(1), the result is accurate to ~1e-13 and therefore differs from the reference's transfer-function evaluation by THAT evaluation's own
// rounding noise (~4e-8 of full scale, one float32 ulp of what the networks consume; tests/test_hip_ops.py gates 3e-7).
// 2 L steps per L outputs, and the samples travel through LDS tiles: 64 blocks x 32 samples are loaded as 256-byte rows (coalesced) and read
// back one row per lane (row pitch 33 doubles: conflict-free).
constexpr int kIirTile = 32, kSosN = 6;
__device__ __forceinline__ double sos_step(const IirArgs& p, double (&z)[kSosN], double x) {
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double y = fma(p.sos[k][0], x, z[2 * k]);
    z[2 * k] = fma(-p.sos[k][4], y, fma(p.sos[k][1], x, z[2 * k + 1]));
    z[2 * k + 1] = fma(-p.sos[k][5], y, p.sos[k][2] * x);
    x = y;
  }
  return x;
}
template <int DIR, bool OUT>
__global__ __launch_bounds__(64) void iir_block_kernel(const IirArgs p) {
  __shared__ double tile[2][64][kIirTile + 1];
  __shared__ double ytile[64][kIirTile + 1];
  const int lane = threadIdx.x;
  const int c0 = blockIdx.x * 64, c = c0 + lane;
  const double* __restrict__ src = DIR == 0 ? p.ext : p.yr;
  const int L = p.L, ntile = L / kIirTile;
  double z[kSosN];
#pragma unroll
  for (int i = 0; i < kSosN; ++i) z[i] = (OUT && c < p.nb) ? p.Zin[(long long)c * kSosN + i] : 0.0;
  const int lr = lane >> 5, lc = lane & 31;              // loader: two rows of 32 samples per pass
  auto load_tile = [&](int t, double (&v)[32]) {
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      const int row = 2 * k + lr;
      v[k] = (c0 + row < p.nb) ? src[(long long)(c0 + row) * L + t * kIirTile + lc] : 0.0;
    }
  };
  double v[32];
  load_tile(0, v);
  for (int t = 0; t < ntile; ++t) {
    double (*tl)[kIirTile + 1] = tile[t & 1];
#pragma unroll
    for (int k = 0; k < 32; ++k) tl[2 * k + lr][lc] = v[k];
    if (t + 1 < ntile) load_tile(t + 1, v);                // in flight under this tile's recurrence
    __syncthreads();
#pragma unroll 8
    for (int sidx = 0; sidx < kIirTile; ++sidx) {
      const double y = sos_step(p, z, tl[lane][sidx]);
      if (OUT) ytile[lane][sidx] = y;
    }
    if (OUT) {
      __syncthreads();
#pragma unroll 4
      for (int k = 0; k < 32; ++k) {
        const int row = 2 * k + lr;
        const long long j = (long long)(c0 + row) * L + t * kIirTile + lc;          // position in the filtered sequence
        const double y = ytile[row][lc];
        if (c0 + row < p.nb) {
          if (DIR == 0) { const long long e = p.N - 1 - j; if (e >= 0) p.yr[e] = y; }             // reversed for the backward pass
          else { const long long i = p.N - 1 - j - p.padlen; if (i >= 0 && i < p.n) p.filt[i] = y; }
        }
      }
      __syncthreads();
    }
  }
  if (!OUT && c < p.nb) {
#pragma unroll
    for (int i = 0; i < kSosN; ++i) p.Z0[(long long)c * kSosN + i] = z[i];
  }
}
// pass 2: z_0 = sosfilt_zi * x[0], z_{c+1} = Z0_c + M z_c - as a two-level scan in LDS by ONE wave (a thread walking the nb states through
// global memory took 300 us: one dependent load + store per block): lane l owns the g = ceil(nb / 64) consecutive blocks [l g, l g + g),
//   (i) runs them from a zero state -> V_l;  (ii) lane 0 chains the 64 groups: z_{l+1} = V_l + Mg z_l with Mg = M^g (host, by the recurrence
//   itself over g L zero samples);  (iii) every lane runs its blocks again from its true initial state, leaving z_c where Z0_c was.
// The nb x 6 states live in LDS (lane pitch g 6 + 1 doubles: conflict-free), read and written back with coalesced sweeps.
template <int DIR>
__global__ __launch_bounds__(256) void iir_scan_kernel(const IirArgs p) {
  extern __shared__ double zs[];                           // [64][g * 6 + 1] | V [64][7]
  const int lane = threadIdx.x;                            // 256 threads move the states, the first 64 scan
  const int nb = p.nb, g = (nb + 63) / 64, pitch = g * kSosN + 1;
  double* V = zs + 64 * pitch;
  for (int c = lane; c < nb; c += 256) {
    const double2* q = reinterpret_cast<const double2*>(p.Z0 + (long long)c * kSosN);
    const double2 a = q[0], b = q[1], d = q[2];
    double* dst = zs + (c / g) * pitch + (c % g) * kSosN;
    dst[0] = a.x; dst[1] = a.y; dst[2] = b.x; dst[3] = b.y; dst[4] = d.x; dst[5] = d.y;
  }
  __syncthreads();
  auto advance = [&](double (&z)[kSosN], const double* z0, const double* Mx) {
    double zn[kSosN];
#pragma unroll
    for (int i = 0; i < kSosN; ++i) {
      double a = z0[i];
#pragma unroll
      for (int jj = 0; jj < kSosN; ++jj) a = fma(Mx[i * kSosN + jj], z[jj], a);
      zn[i] = a;
    }
#pragma unroll
    for (int i = 0; i < kSosN; ++i) z[i] = zn[i];
  };
  double* mine = zs + (lane & 63) * pitch;
  double Mr[kSosN * kSosN];                                // (register copy: indexing the kernel argument re-loads it through the scalar cache every step)
#pragma unroll
  for (int i = 0; i < kSosN * kSosN; ++i) Mr[i] = p.M[i];
  if (lane < 64) {
    double z[kSosN] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < g; ++i) if (lane * g + i < nb) advance(z, mine + i * kSosN, Mr);
#pragma unroll
    for (int i = 0; i < kSosN; ++i) V[lane * 7 + i] = z[i];
  }
  __syncthreads();
  if (lane == 0) {
    const double x0 = (DIR == 0 ? p.ext : p.yr)[0];
    double z[kSosN], Mgr[kSosN * kSosN];
#pragma unroll
    for (int i = 0; i < kSosN * kSosN; ++i) Mgr[i] = p.Mg[i];
#pragma unroll
    for (int i = 0; i < kSosN; ++i) z[i] = p.zis[i] * x0;
    for (int l = 0; l < 64; ++l) {
      double v[kSosN];
#pragma unroll
      for (int i = 0; i < kSosN; ++i) { v[i] = V[l * 7 + i]; V[l * 7 + i] = z[i]; }      // V_l is replaced by the group's initial state
      advance(z, v, Mgr);
    }
  }
  __syncthreads();
  if (lane < 64) {
    double z[kSosN];
#pragma unroll
    for (int i = 0; i < kSosN; ++i) z[i] = V[lane * 7 + i];
    for (int i = 0; i < g; ++i) {
      if (lane * g + i >= nb) break;
      double z0[kSosN];
#pragma unroll
      for (int k = 0; k < kSosN; ++k) { z0[k] = mine[i * kSosN + k]; mine[i * kSosN + k] = z[k]; }
      advance(z, z0, Mr);
    }
  }
  __syncthreads();
  for (int c = lane; c < nb; c += 256) {
    const double* src = zs + (c / g) * pitch + (c % g) * kSosN;
    double2* q = reinterpret_cast<double2*>(p.Zin + (long long)c * kSosN);
    q[0] = double2{src[0], src[1]}; q[1] = double2{src[2], src[3]}; q[2] = double2{src[4], src[5]};
  }
}

// np.pad(x, t_pad, mode="reflect") + float32 cast; pads wider than the signal reflect repeatedly (period 2 (n - 1)), as numpy does
// for clips shorter than the 1 s pad
__global__ void pad_reflect_f32_kernel(const double* __restrict__ x, long long n, int t_pad, float* __restrict__ out) {
  const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n + 2LL * t_pad) return;
  const long long period = 2 * (n - 1);
  long long i = (j - t_pad) % period;
  if (i < 0) i += period;
  if (i >= n) i = period - i;
  out[j] = (float)x[i];
}

// librosa.feature.rms(y, frame_length, hop_length) on a float64 signal: centred frames, zero padding
__global__ __launch_bounds__(256) void rms_frames_f64_kernel(const double* __restrict__ x, long long n, int frame, int hop, double* __restrict__ rms) {
  __shared__ double red[256];
  const long long start = (long long)blockIdx.x * hop - frame / 2;
  double s = 0.0;
  for (int i = threadIdx.x; i < frame; i += 256) {
    const long long j = start + i;
    if (j >= 0 && j < n) { const double v = x[j]; s = fma(v, v, s); }
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) rms[blockIdx.x] = sqrt(red[0] / frame);
}

// Block length L and padded length Np of the filter pass for an input of n samples: ONE definition for the caller's scratch (2 Np doubles,
// preprocess_scratch_doubles) and for the kernels (advisor, round 3: the scratch used to assume L <= 65536, which clips beyond ~2e8 samples
// exceed, and an RVC_IIR_L that is not a power of two >= 32 silently dropped samples).
static void iir_geometry(long long n, bool blocked, int& L, long long& Np) {
  const long long N = n + 2 * 3 * (kIirOrder + 1);
  if (!blocked) { L = 512; Np = (N + 7) & ~7LL; return; }
  static const int blk_env = [] {
    const int v = exp_int("RVC_IIR_L", 256);      // block length of the propagated evaluation
    RVC_REQUIRE(v >= 32 && v <= (1 << 20) && (v & (v - 1)) == 0, "RVC_IIR_L must be a power of two >= 32");
    return v;
  }();
  long long l = blk_env;
  while ((N + l - 1) / l > 3072) l *= 2;                     // the scan keeps all block states in LDS (6 doubles each)
  RVC_REQUIRE(l <= (1LL << 30), "clip too long for the blocked filter");
  L = (int)l; Np = ((N + l - 1) / l) * l;                    // whole blocks (zero-filled behind N)
}
static bool iir_blocked(bool have_sos) {
  static const bool blocked_env = (exp_int("RVC_IIR_BLOCKED", 1) != 0);
  return blocked_env && have_sos;
}
size_t preprocess_scratch_doubles(long long n, bool have_sos) {
  int L; long long Np;
  iir_geometry(n, iir_blocked(have_sos), L, Np);
  return (size_t)(2 * Np);
}

void preprocess(hipStream_t s, const void* x, int is64, long long n, const double* b, const double* a, const double* zi, int t_pad,
                double* filt, float* padded, double* rms1, int n1, int frame, int hop, double* scratch /* preprocess_scratch_doubles(n, sos) */,
                const double* sos, const double* sos_zi) {
  IirArgs p{};
  for (int i = 0; i <= kIirOrder; ++i) { p.b[i] = b[i] / a[0]; p.a[i] = a[i] / a[0]; }
  for (int i = 0; i < kIirOrder; ++i) p.zi[i] = zi[i];
  const bool blocked = iir_blocked(sos != nullptr && sos_zi != nullptr);
  p.x = x; p.is64 = is64; p.n = n; p.padlen = 3 * (kIirOrder + 1); p.N = n + 2 * p.padlen;
  p.W = 5120;
  iir_geometry(n, blocked, p.L, p.Np);
  p.nchunks = (int)((p.Np + p.L - 1) / p.L); p.nb = p.nchunks;
  p.ext = scratch; p.yr = scratch + p.Np; p.filt = filt;
  hipLaunchKernelGGL(iir_extend_kernel, dim3((unsigned)((p.Np + 255) / 256)), dim3(256), 0, s, p);
  if (blocked) {
    for (int k = 0; k < 3; ++k) {
      RVC_REQUIRE(sos[k * 6 + 3] == 1.0, "sos sections must be normalised (a0 = 1)");
      for (int i = 0; i < 6; ++i) p.sos[k][i] = sos[k * 6 + i];
    }
    for (int i = 0; i < 6; ++i) p.zis[i] = sos_zi[i];
    // M / Mg: column j = the cascade's state after L (g L) zero-input samples from the unit state e_j (the recurrence itself, float64: O(1) entries)
    const int gsz = (p.nb + 63) / 64;
    auto transition = [&](long long steps, double* out) {
      for (int j = 0; j < 6; ++j) {
        double z[6] = {0, 0, 0, 0, 0, 0}; z[j] = 1.0;
        for (long long t = 0; t < steps; ++t) {
          double xx = 0.0;
          for (int k = 0; k < 3; ++k) {
            const double y = std::fma(p.sos[k][0], xx, z[2 * k]);
            z[2 * k] = std::fma(-p.sos[k][4], y, std::fma(p.sos[k][1], xx, z[2 * k + 1]));
            z[2 * k + 1] = std::fma(-p.sos[k][5], y, p.sos[k][2] * xx);
            xx = y;
          }
        }
        for (int i = 0; i < 6; ++i) out[i * 6 + j] = z[i];
      }
    };
    transition(p.L, p.M);
    transition((long long)p.L * gsz, p.Mg);
    double* st = (double*)stream_scratch(s, 8, (size_t)2 * p.nb * 6 * sizeof(double));
    p.Z0 = st; p.Zin = st + (size_t)p.nb * 6;
    const dim3 g((unsigned)((p.nb + 63) / 64));
    hipLaunchKernelGGL((iir_block_kernel<0, false>), g, dim3(64), 0, s, p);
    const size_t scan_lds = ((size_t)64 * (gsz * 6 + 1) + 64 * 7) * sizeof(double);
    RVC_ALLOW_BIG_LDS(iir_scan_kernel<0>);
    RVC_ALLOW_BIG_LDS(iir_scan_kernel<1>);
    hipLaunchKernelGGL((iir_scan_kernel<0>), dim3(1), dim3(256), scan_lds, s, p);
    hipLaunchKernelGGL((iir_block_kernel<0, true>), g, dim3(64), 0, s, p);
    hipLaunchKernelGGL((iir_block_kernel<1, false>), g, dim3(64), 0, s, p);
    hipLaunchKernelGGL((iir_scan_kernel<1>), dim3(1), dim3(256), scan_lds, s, p);
    hipLaunchKernelGGL((iir_block_kernel<1, true>), g, dim3(64), 0, s, p);
  } else {
    const dim3 grid((p.nchunks + 63) / 64), blk(64);
    hipLaunchKernelGGL((iir_chunk_kernel<0>), grid, blk, 0, s, p);
    hipLaunchKernelGGL((iir_chunk_kernel<1>), grid, blk, 0, s, p);
  }
  if (padded) {
    const long long np = n + 2LL * t_pad;
    hipLaunchKernelGGL(pad_reflect_f32_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, s, filt, n, t_pad, padded);
  }
  if (rms1 && n1 > 0) hipLaunchKernelGGL(rms_frames_f64_kernel, dim3(n1), dim3(256), 0, s, filt, n, frame, hop, rms1);
}

}  // namespace rvc
