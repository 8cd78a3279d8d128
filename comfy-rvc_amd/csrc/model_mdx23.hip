// MDX23C source-separation network (karafan TFC_TDF_net) as a HIP kernel graph: reference lib/karafan/tfc_tdf.py:47-235 - STFT
// (n_fft, hop, hann, centre / reflect, first dim_f bins), cac2cws sub-band split, first 1x1 conv, an encoder / bottleneck / decoder of
// TFC_TDF blocks (InstanceNorm2d + GELU in front of every 3x3 convolution, the two "TDF" linears over the frequency axis, the 2x2
// stride-2 down / up convolutions, skip concatenation), the mask head and the inverse STFT.  One call = one chunk [2, hop * (dim_t - 1)]
// -> [S, 2, chunk] (demix_mdxv3, lib/karafan/inference.py:32-74, drives it chunk by chunk from Python).
// Layout: planes [C][H = time frames][W = frequency bins of one sub-band], W contiguous (= the reference after its transpose(-1, -2)).
//   3x3 convolutions            conv2d_run (bf16x3 / fp32 MFMA implicit GEMM, conv_x3.hip / conv_mfma.hip)
//   1x1 convolutions, linears   conv1d_run with k = 1; the TDF linears contract over the contiguous axis, so the activation is
//                               transposed to [f][C * T] in front of them and back (with the residual add) behind them
//   2x2 stride-2 conv / deconv  space-to-depth / depth-to-space re-arrangement around a k = 1 GEMM
//   STFT / inverse STFT         framing + GEMM against windowed DFT matrices (built on the host in float64), overlap-add kernel
#include "model_common.h"
#include "models.h"

namespace rvc {

struct TfcBlock {
  int in_c = 0, c = 0, f = 0;
  DevVec n1g, n1b, t0g, t0b, t3g, t3b, n2g, n2b;
  ConvLayer tfc1, tfc2, shortcut, lin1, lin2;
};
struct MdxScale {
  std::vector<TfcBlock> blocks;
  DevVec ng, nb;            // norm in front of the down / up convolution
  ConvLayer rs;             // the 2x2 stride-2 (transposed) convolution as a k = 1 GEMM over re-arranged data
};
struct Mdx23 {
  Ctx* ctx = nullptr;
  Arena arena;
  TensorStore ts;
  bool ready = false;
  rvc_mdx23_config cfg{};
  ConvLayer stft, istft, first, fin0, fin2;
  DevVec window;
  std::vector<MdxScale> enc, dec;
  MdxScale bott;
};

Mdx23* mdx23_create(Ctx* ctx, const rvc_mdx23_config& c) {
  RVC_REQUIRE(c.n_fft > 0 && c.hop > 0 && c.dim_f > 0 && c.dim_f <= c.n_fft / 2 && c.dim_t > 1 && c.num_subbands > 0 && c.dim_f % c.num_subbands == 0, "bad STFT geometry");
  RVC_REQUIRE(c.num_scales >= 0 && c.num_scales < 8 && c.blocks_per_scale > 0 && c.bottleneck > 0 && c.num_targets > 0 && c.audio_channels == 2, "bad network geometry");
  const int f = c.dim_f / c.num_subbands;
  RVC_REQUIRE((f & (f - 1)) == 0 && (f >> c.num_scales) >= 4 && ((c.dim_t >> c.num_scales) << c.num_scales) == c.dim_t,
              "sub-band width must be a power of two and dim_t divisible by 2^num_scales");
  Mdx23* M = new Mdx23(); M->ctx = ctx; M->cfg = c; return M;
}
void mdx23_set_tensor(Mdx23* M, const char* name, const float* d, const long long* shape, int ndim) { M->ts.set(name, d, shape, ndim); }

static void tfc_free(TfcBlock& b) {
  b.n1g.free_(); b.n1b.free_(); b.t0g.free_(); b.t0b.free_(); b.t3g.free_(); b.t3b.free_(); b.n2g.free_(); b.n2b.free_();
  conv_layer_free(b.tfc1); conv_layer_free(b.tfc2); conv_layer_free(b.shortcut); conv_layer_free(b.lin1); conv_layer_free(b.lin2);
}
static void scale_free(MdxScale& s) { for (auto& b : s.blocks) tfc_free(b); s.blocks.clear(); s.ng.free_(); s.nb.free_(); conv_layer_free(s.rs); }
static void mdx23_free(Mdx23& M) {
  conv_layer_free(M.stft); conv_layer_free(M.istft); conv_layer_free(M.first); conv_layer_free(M.fin0); conv_layer_free(M.fin2);
  M.window.free_();
  for (auto& s : M.enc) scale_free(s);
  for (auto& s : M.dec) scale_free(s);
  scale_free(M.bott);
  M.enc.clear(); M.dec.clear();
}
void mdx23_destroy(Mdx23* M) { if (M) { mdx23_free(*M); M->arena.release(); delete M; } }

static void make_tfc(std::vector<TfcBlock>& out, const TensorStore& ts, const std::string& prefix, int in_c, int c, int f, int l, int bn) {
  out.resize((size_t)l);
  for (int i = 0; i < l; ++i) {
    TfcBlock& B = out[(size_t)i];
    const std::string p = prefix + ".blocks." + std::to_string(i) + ".";
    B.in_c = in_c; B.c = c; B.f = f;
    B.n1g.upload(ts.get(p + "tfc1.0.weight", {in_c}).data); B.n1b.upload(ts.get(p + "tfc1.0.bias", {in_c}).data);
    conv2d3x3_layer_init(B.tfc1, ts.get(p + "tfc1.2.weight", {c, in_c, 3, 3}).data.data(), nullptr, c, in_c);
    B.t0g.upload(ts.get(p + "tdf.0.weight", {c}).data); B.t0b.upload(ts.get(p + "tdf.0.bias", {c}).data);
    conv1d_layer_init(B.lin1, ts.get(p + "tdf.2.weight", {f / bn, f}).data.data(), nullptr, f / bn, f, 1, 1, 0, 1, 1);
    B.t3g.upload(ts.get(p + "tdf.3.weight", {c}).data); B.t3b.upload(ts.get(p + "tdf.3.bias", {c}).data);
    conv1d_layer_init(B.lin2, ts.get(p + "tdf.5.weight", {f, f / bn}).data.data(), nullptr, f, f / bn, 1, 1, 0, 1, 1);
    B.n2g.upload(ts.get(p + "tfc2.0.weight", {c}).data); B.n2b.upload(ts.get(p + "tfc2.0.bias", {c}).data);
    conv2d3x3_layer_init(B.tfc2, ts.get(p + "tfc2.2.weight", {c, c, 3, 3}).data.data(), nullptr, c, c);
    conv1d_layer_init(B.shortcut, ts.get(p + "shortcut.weight", {c, in_c, 1, 1}).data.data(), nullptr, c, in_c, 1, 1, 0, 1, 1);
    in_c = c;
  }
}

void mdx23_finalize(Mdx23* M) {
  const TensorStore& ts = M->ts;
  const rvc_mdx23_config& c = M->cfg;
  mdx23_free(*M);
  ConvBuildScope x3scope(M->ctx->precision);
  const int k = c.num_subbands, dim_c = k * c.audio_channels * 2, n = c.num_scales, l = c.blocks_per_scale, g = c.growth, bn = c.bottleneck;
  conv1d_layer_init(M->stft, ts.get("stft.basis", {2 * c.dim_f, c.n_fft}).data.data(), nullptr, 2 * c.dim_f, c.n_fft, 1, 1, 0, 1, 1);
  conv1d_layer_init(M->istft, ts.get("istft.basis", {c.n_fft, 2 * c.dim_f}).data.data(), nullptr, c.n_fft, 2 * c.dim_f, 1, 1, 0, 1, 1);
  M->window.upload(ts.get("window", {c.n_fft}).data);
  int ch = c.num_channels, f = c.dim_f / k;
  conv1d_layer_init(M->first, ts.get("first_conv.weight", {ch, dim_c, 1, 1}).data.data(), nullptr, ch, dim_c, 1, 1, 0, 1, 1);
  M->enc.resize((size_t)n); M->dec.resize((size_t)n);
  for (int i = 0; i < n; ++i) {
    MdxScale& S = M->enc[(size_t)i];
    const std::string p = "encoder_blocks." + std::to_string(i);
    make_tfc(S.blocks, ts, p + ".tfc_tdf", ch, ch, f, l, bn);
    S.ng.upload(ts.get(p + ".downscale.conv.0.weight", {ch}).data); S.nb.upload(ts.get(p + ".downscale.conv.0.bias", {ch}).data);
    // Conv2d(ch -> ch + g, 2x2, stride 2): GEMM over space-to-depth rows (ci * 4 + dy * 2 + dx) = the weight's own memory order
    conv1d_layer_init(S.rs, ts.get(p + ".downscale.conv.2.weight", {ch + g, ch, 2, 2}).data.data(), nullptr, ch + g, 4 * ch, 1, 1, 0, 1, 1);
    f /= 2; ch += g;
  }
  make_tfc(M->bott.blocks, ts, "bottleneck_block", ch, ch, f, l, bn);
  for (int i = 0; i < n; ++i) {
    MdxScale& S = M->dec[(size_t)i];
    const std::string p = "decoder_blocks." + std::to_string(i);
    S.ng.upload(ts.get(p + ".upscale.conv.0.weight", {ch}).data); S.nb.upload(ts.get(p + ".upscale.conv.0.bias", {ch}).data);
    // ConvTranspose2d(ch -> ch - g, 2x2, stride 2), weight [ci][co][dy][dx]: GEMM rows (co * 4 + dy * 2 + dx), then depth-to-space
    const HostTensor& w = ts.get(p + ".upscale.conv.2.weight", {ch, ch - g, 2, 2});
    std::vector<float> wt((size_t)4 * (ch - g) * ch);
    for (int ci = 0; ci < ch; ++ci) for (int r = 0; r < 4 * (ch - g); ++r) wt[(size_t)r * ch + ci] = w.data[(size_t)ci * 4 * (ch - g) + r];
    conv1d_layer_init(S.rs, wt.data(), nullptr, 4 * (ch - g), ch, 1, 1, 0, 1, 1);
    f *= 2; ch -= g;
    make_tfc(S.blocks, ts, p + ".tfc_tdf", 2 * ch, ch, f, l, bn);
  }
  conv1d_layer_init(M->fin0, ts.get("final_conv.0.weight", {ch, ch + dim_c, 1, 1}).data.data(), nullptr, ch, ch + dim_c, 1, 1, 0, 1, 1);
  conv1d_layer_init(M->fin2, ts.get("final_conv.2.weight", {c.num_targets * dim_c, ch, 1, 1}).data.data(), nullptr, c.num_targets * dim_c, ch, 1, 1, 0, 1, 1);
  M->ts.clear();
  M->ready = true;
}

// ---------------------------------------------------------------------------------------------- kernels
// InstanceNorm2d statistics of a tensor laid out [A][C][B] (channel c owns A runs of B contiguous values): per channel the folded
// scale a = gamma / sqrt(var + eps) and shift b = beta - mean * a (biased variance, float64 accumulation).
// Two deterministic stages (round 4): one workgroup per channel left half of the chip idle at 128 channels and streamed 1 MiB per workgroup
// (29 us average, 13 % of the separation's kernel time); now P workgroups per channel write partial (sum, sum of squares) pairs in float64 and a
// second, tiny launch adds them in part order - the same result on every run - and folds gamma / beta.
__global__ __launch_bounds__(256) void inorm_part_kernel(const float* __restrict__ x, int A, int C, long long B, int P, double* __restrict__ part) {
  const int c = blockIdx.x, q = blockIdx.y;
  const long long n = (long long)A * B;
  const long long per = ((n + P - 1) / P + 3) & ~3LL;          // elements of a part (a multiple of 4: the float4 path below stays aligned when B % 4 == 0)
  const long long i0 = (long long)q * per, i1 = i0 + per < n ? i0 + per : n;
  double s = 0.0, qq = 0.0;
  if (A == 1 && (B & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) & 15) == 0)) {
    const float4* xp = reinterpret_cast<const float4*>(x + (long long)c * B);
    for (long long i = i0 / 4 + threadIdx.x; i < i1 / 4; i += blockDim.x) {
      const float4 v = xp[i];
      s += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
      qq += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
    for (long long i = (i1 / 4) * 4 + threadIdx.x; i < i1; i += blockDim.x) { const float v = x[(long long)c * B + i]; s += v; qq += (double)v * v; }
  } else {
    for (long long i = i0 + threadIdx.x; i < i1; i += blockDim.x) {
      const long long a = i / B, b = i - a * B;
      const float v = x[(a * C + c) * B + b];
      s += v; qq += (double)v * v;
    }
  }
  __shared__ double ss[256], sq[256];
  ss[threadIdx.x] = s; sq[threadIdx.x] = qq;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { ss[threadIdx.x] += ss[threadIdx.x + o]; sq[threadIdx.x] += sq[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { part[((long long)c * P + q) * 2] = ss[0]; part[((long long)c * P + q) * 2 + 1] = sq[0]; }
}
__global__ void inorm_fold_kernel(const double* __restrict__ part, int C, int P, double n, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                  float* __restrict__ sc, float* __restrict__ sh) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s = 0.0, q = 0.0;
  for (int p = 0; p < P; ++p) { s += part[((long long)c * P + p) * 2]; q += part[((long long)c * P + p) * 2 + 1]; }
  const double m = s / n, var = q / n - m * m;
  const float a = gamma[c] * (float)(1.0 / sqrt((var > 0.0 ? var : 0.0) + (double)eps));
  sc[c] = a; sh[c] = beta[c] - (float)m * a;
}
__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f)); }
// y = gelu(x * sc[c] + sh[c]) over [A][C][B]
__global__ void inorm_apply_gelu_kernel(const float* __restrict__ x, float* __restrict__ y, int C, long long B, long long n, const float* __restrict__ sc,
                                        const float* __restrict__ sh) {
  const long long st = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += st) {
    const int c = (int)((i / B) % C);
    y[i] = gelu_erf(fmaf(x[i], sc[c], sh[c]));
  }
}
// out[c][r] = in[r][c] (+ res[c][r]) for an R x C matrix, 32 x 32 tiles through LDS
__global__ void tr2d_kernel(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ res, int R, int Cn) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int j = threadIdx.y; j < 32; j += blockDim.y) {
    const int r = r0 + j, c = c0 + threadIdx.x;
    tile[j][threadIdx.x] = (r < R && c < Cn) ? in[(long long)r * Cn + c] : 0.f;
  }
  __syncthreads();
  for (int j = threadIdx.y; j < 32; j += blockDim.y) {
    const int c = c0 + j, r = r0 + threadIdx.x;
    if (c < Cn && r < R) {
      const long long o = (long long)c * R + r;
      out[o] = tile[threadIdx.x][j] + (res ? res[o] : 0.f);
    }
  }
}
// space-to-depth: out[(ci * 4 + dy * 2 + dx)][y][x] = in[ci][2 y + dy][2 x + dx]
__global__ void s2d_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int H, int W) {
  const int Ho = H / 2, Wo = W / 2;
  const long long n = (long long)C * H * W, st = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += st) {
    const int x = (int)(i % Wo); long long r = i / Wo;
    const int y = (int)(r % Ho); r /= Ho;
    const int q = (int)(r & 3), ci = (int)(r >> 2);
    out[i] = in[((long long)ci * H + 2 * y + (q >> 1)) * W + 2 * x + (q & 1)];
  }
}
// depth-to-space: out[co][2 y + dy][2 x + dx] = in[(co * 4 + dy * 2 + dx)][y][x]     (H, W: input plane)
__global__ void d2s_kernel(const float* __restrict__ in, float* __restrict__ out, int Co, int H, int W) {
  const long long n = (long long)Co * 4 * H * W, st = (long long)gridDim.x * blockDim.x;
  const int Wo = 2 * W, Ho = 2 * H;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += st) {
    const int xo = (int)(i % Wo); long long r = i / Wo;
    const int yo = (int)(r % Ho); const int co = (int)(r / Ho);
    out[i] = in[(((long long)co * 4 + (yo & 1) * 2 + (xo & 1)) * H + (yo >> 1)) * W + (xo >> 1)];
  }
}
__global__ void mul_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, long long n) {
  const long long st = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += st) y[i] = a[i] * b[i];
}
// overlap-add of windowed inverse-FFT frames fr [n_fft][T] over the squared-window envelope, centre-trimmed (torch.istft, center=True):
// out[i] = sum_m fr[i + n_fft / 2 - m * hop][m] / sum_m w^2[i + n_fft / 2 - m * hop]
__global__ void ola_kernel(const float* __restrict__ fr, const float* __restrict__ w, float* __restrict__ out, int n_fft, int hop, int T, long long L) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= L) return;
  const long long p = i + n_fft / 2;
  long long m1 = p / hop; if (m1 > T - 1) m1 = T - 1;
  long long m0 = (p - n_fft + hop) / hop; if (p - n_fft + 1 < 0) m0 = 0; if (m0 < 0) m0 = 0;
  float s = 0.f, e = 0.f;
  for (long long m = m0; m <= m1; ++m) {
    const long long j = p - m * hop;
    if (j < 0 || j >= n_fft) continue;
    s += fr[j * T + m]; const float ww = w[j]; e = fmaf(ww, ww, e);
  }
  out[i] = s / e;
}

static int gridn(long long n) { long long g = (n + 255) / 256; return (int)(g > 32768 ? 32768 : (g < 1 ? 1 : g)); }
static void norm_gelu(hipStream_t s, const float* x, float* y, int A, int C, long long B, const float* g, const float* b, float* sc, float* sh) {
  const long long n = (long long)A * C * B, per_c = (long long)A * B;
  // parts per channel: ~1024 workgroups in all, at least 8192 elements each
  int P = (int)((1024 + C - 1) / C);
  while (P > 1 && per_c / P < 8192) --P;
  double* part = (double*)stream_scratch(s, 12, (size_t)C * P * 2 * sizeof(double));
  hipLaunchKernelGGL(inorm_part_kernel, dim3((unsigned)C, (unsigned)P), dim3(256), 0, s, x, A, C, B, P, part);
  hipLaunchKernelGGL(inorm_fold_kernel, dim3((unsigned)((C + 127) / 128)), dim3(128), 0, s, part, C, P, (double)per_c, g, b, 1e-5f, sc, sh);
  hipLaunchKernelGGL(inorm_apply_gelu_kernel, dim3(gridn(n)), dim3(256), 0, s, x, y, C, B, n, sc, sh);
}
static void tr2d(hipStream_t s, const float* in, float* out, const float* res, int R, int C) {
  hipLaunchKernelGGL(tr2d_kernel, dim3((unsigned)((C + 31) / 32), (unsigned)((R + 31) / 32)), dim3(32, 8), 0, s, in, out, res, R, C);
}

// scratch planes shared by every block (sized for the largest scale)
struct MdxScratch { float *t1, *t2, *t3, *sbuf, *sc, *sh; };

// TFC_TDF.forward (tfc_tdf.py:137-144) over planes [c][H][W]; the last block writes to `out`
static void run_tfc(const std::vector<TfcBlock>& blocks, hipStream_t s, const MdxScratch& K, const float* x, int H, int W, float* out, float* mid) {
  const long long plane = (long long)H * W;
  ConvEpilogue E0;
  const float* cur = x;
  for (size_t i = 0; i < blocks.size(); ++i) {
    const TfcBlock& B = blocks[i];
    float* dst = (i + 1 == blocks.size()) ? out : mid;
    conv1d_run(B.shortcut, s, cur, plane, (int)plane, K.sbuf, plane, E0);                       // s = shortcut(x)
    norm_gelu(s, cur, K.t1, 1, B.in_c, plane, B.n1g.p, B.n1b.p, K.sc, K.sh);
    conv2d_run(B.tfc1, s, K.t1, plane, H, W, K.t2, plane, E0);                                 // x1 = tfc1(x)            [c][H][W]
    // x2 = x1 + tdf(x1): the linears contract over W (contiguous), so they run on the transposed matrix [W][c * H]
    const int R = B.c * H, fb = B.lin1.Co;
    norm_gelu(s, K.t2, K.t1, 1, B.c, plane, B.t0g.p, B.t0b.p, K.sc, K.sh);
    tr2d(s, K.t1, K.t3, nullptr, R, W);                                                        // [W][R]
    conv1d_run(B.lin1, s, K.t3, R, R, K.t1, R, E0);                                            // [W / bn][R]
    norm_gelu(s, K.t1, K.t1, fb, B.c, H, B.t3g.p, B.t3b.p, K.sc, K.sh);                        // channel c owns fb runs of H values
    conv1d_run(B.lin2, s, K.t1, R, R, K.t3, R, E0);                                            // [W][R]
    tr2d(s, K.t3, K.t1, K.t2, W, R);                                                           // x2 = transpose back + x1   [R][W]
    norm_gelu(s, K.t1, K.t3, 1, B.c, plane, B.n2g.p, B.n2b.p, K.sc, K.sh);
    ConvEpilogue Er; Er.R = K.sbuf; Er.ldR = plane;
    conv2d_run(B.tfc2, s, K.t3, plane, H, W, dst, plane, Er);                                  // tfc2(x2) + s
    cur = dst;
  }
}

static void mdx23_graph(Mdx23* M, hipStream_t s, Arena& A, const float* audio, long long L, float* out) {
  const rvc_mdx23_config& c = M->cfg;
  const bool dry = A.dry;
  const int T = c.dim_t, k = c.num_subbands, f0 = c.dim_f / k, n = c.num_scales, g = c.growth, S = c.num_targets;
  const int dim_c = k * 4, c0 = c.num_channels;
  ConvEpilogue E0;
  const long long FT = (long long)f0 * T;
  float* fr = A.alloc<float>((size_t)c.n_fft * T);
  float* spec = A.alloc<float>((size_t)4 * c.dim_f * T);          // [ch][re | im][dim_f][T] = cac2cws view [16][f0][T]
  float* first = A.alloc<float>((size_t)c0 * FT);
  // per-scale geometry
  std::vector<int> ch((size_t)n + 1), Hs((size_t)n + 1), Ws((size_t)n + 1);
  for (int i = 0; i <= n; ++i) { ch[(size_t)i] = c0 + i * g; Hs[(size_t)i] = T >> i; Ws[(size_t)i] = f0 >> i; }
  size_t big = 0;
  for (int i = 0; i <= n; ++i) big = std::max(big, (size_t)2 * ch[(size_t)i] * Hs[(size_t)i] * Ws[(size_t)i]);
  big = std::max(big, (size_t)((long long)(c0 + dim_c) * FT));
  big = std::max(big, (size_t)((long long)S * dim_c * FT));        // the mask head's output [S dim_c][f0][T] lands in t1 as well
  MdxScratch K;
  K.t1 = A.alloc<float>(big); K.t2 = A.alloc<float>(big); K.t3 = A.alloc<float>(big); K.sbuf = A.alloc<float>(big);
  K.sc = A.alloc<float>((size_t)2 * ch[(size_t)n] + 64); K.sh = A.alloc<float>((size_t)2 * ch[(size_t)n] + 64);
  float* mid = A.alloc<float>(big);
  std::vector<float*> cat((size_t)n);                             // decoder inputs [2 c_i][H_i][W_i]: [up-sampled | encoder skip]
  for (int i = 0; i < n; ++i) cat[(size_t)i] = A.alloc<float>((size_t)2 * ch[(size_t)i] * Hs[(size_t)i] * Ws[(size_t)i]);
  float* x0 = A.alloc<float>((size_t)c0 * FT);
  float* cur = A.alloc<float>(big / 2 + 64);
  float* cur2 = A.alloc<float>(big / 2 + 64);
  if (dry) return;
  // ---- STFT of both channels: frames (reflect-padded by n_fft / 2) x windowed DFT matrix
  for (int a = 0; a < 2; ++a) {
    frames(s, audio + (long long)a * L, fr, (int)L, c.n_fft, c.hop, c.n_fft / 2, T, 1);
    conv1d_run(M->stft, s, fr, T, T, spec + (size_t)a * 2 * c.dim_f * T, T, E0);
  }
  conv1d_run(M->first, s, spec, FT, (int)FT, first, FT, E0);                                   // [c0][f0][T]
  transpose(s, first, x0, f0, T, T, f0, c0, FT, FT);                                           // [c0][T][f0]
  // ---- encoder
  const float* h = x0;
  for (int i = 0; i < n; ++i) {
    const int C = ch[(size_t)i], H = Hs[(size_t)i], W = Ws[(size_t)i];
    const long long plane = (long long)H * W;
    float* skip = cat[(size_t)i] + (size_t)C * plane;
    run_tfc(M->enc[(size_t)i].blocks, s, K, h, H, W, skip, mid);
    norm_gelu(s, skip, K.t1, 1, C, plane, M->enc[(size_t)i].ng.p, M->enc[(size_t)i].nb.p, K.sc, K.sh);
    hipLaunchKernelGGL(s2d_kernel, dim3(gridn((long long)C * plane)), dim3(256), 0, s, K.t1, K.t2, C, H, W);
    conv1d_run(M->enc[(size_t)i].rs, s, K.t2, plane / 4, (int)(plane / 4), cur, plane / 4, E0);  // [C + g][H / 2][W / 2]
    h = cur; std::swap(cur, cur2);
  }
  {
    float* o = cur;
    run_tfc(M->bott.blocks, s, K, h, Hs[(size_t)n], Ws[(size_t)n], o, mid);
    h = o; std::swap(cur, cur2);
  }
  // ---- decoder
  for (int i = 0; i < n; ++i) {
    const int lv = n - 1 - i;                                       // output level
    const int Ci = ch[(size_t)lv + 1], Co = ch[(size_t)lv], H = Hs[(size_t)lv + 1], W = Ws[(size_t)lv + 1];
    const long long pin = (long long)H * W;
    norm_gelu(s, h, K.t1, 1, Ci, pin, M->dec[(size_t)i].ng.p, M->dec[(size_t)i].nb.p, K.sc, K.sh);
    conv1d_run(M->dec[(size_t)i].rs, s, K.t1, pin, (int)pin, K.t2, pin, E0);                   // [(co, dy, dx)][H][W]
    hipLaunchKernelGGL(d2s_kernel, dim3(gridn((long long)Co * 4 * pin)), dim3(256), 0, s, K.t2, cat[(size_t)lv], Co, H, W);
    float* o = cur;
    run_tfc(M->dec[(size_t)i].blocks, s, K, cat[(size_t)lv], 2 * H, 2 * W, o, mid);
    h = o; std::swap(cur, cur2);
  }
  // ---- mask head: x^T * first_conv_out, cat(mix, x), 1x1 -> GELU -> 1x1
  float* hc = K.t3;                                                 // [dim_c + c0][f0][T]
  RVC_HIP_CHECK(hipMemcpyAsync(hc, spec, (size_t)dim_c * FT * sizeof(float), hipMemcpyDeviceToDevice, s));
  transpose(s, h, K.t1, T, f0, f0, T, c0, FT, FT);                                             // [c0][f0][T]
  hipLaunchKernelGGL(mul_kernel, dim3(gridn((long long)c0 * FT)), dim3(256), 0, s, K.t1, first, hc + (size_t)dim_c * FT, (long long)c0 * FT);
  ConvEpilogue Eg; Eg.act = ACT_GELU;
  conv1d_run(M->fin0, s, hc, FT, (int)FT, K.t2, FT, Eg);
  conv1d_run(M->fin2, s, K.t2, FT, (int)FT, K.t1, FT, E0);                                     // [S * 16][f0][T] = [S][ch][re | im][dim_f][T]
  // ---- inverse STFT per source and channel
  for (int q = 0; q < S * 2; ++q) {
    conv1d_run(M->istft, s, K.t1 + (size_t)q * 2 * c.dim_f * T, T, T, fr, T, E0);              // [n_fft][T] windowed inverse FFT frames
    hipLaunchKernelGGL(ola_kernel, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, s, fr, M->window.p, out + (long long)q * L, c.n_fft, c.hop, T, L);
  }
}

void mdx23_forward(Mdx23* M, hipStream_t s, const float* audio, long long L, float* out) {
  RVC_REQUIRE(M->ready, "mdx23_finalize has not been called");
  RVC_REQUIRE(L == (long long)M->cfg.hop * (M->cfg.dim_t - 1), "a chunk is hop * (dim_t - 1) samples per channel");
  Arena& A = M->arena;
  for (int pass = 0; pass < 2; ++pass) {
    A.dry = (pass == 0); A.reset(); if (pass == 0) A.peak = 0;
    mdx23_graph(M, s, A, audio, L, out);
    if (pass == 0) A.ensure(A.peak);
  }
  A.dry = false;
}
size_t mdx23_workspace(const Mdx23* M) { return M->arena.cap; }

}  // namespace rvc
