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
// Round 4: when every 3x3 / 2x2 layer has its bf16x3 weight image the U-Net runs on PADDED planes [C][H][W + 2] (one zero column on either side
// of a row, as RMVPE's deep levels: split2d.hip) and the split-resident GEMM kernel (conv_x3s.hip): InstanceNorm + GELU writes the bf16 hi / lo
// image the next product stages by DMA, the 3x3 convolutions are 9 row offsets into that image, the TDF linears take the image of the
// TRANSPOSED plane ([W / 16 chunks][c H positions]: 16 neighbouring bins of a row are one image row) and the second linear runs as the swapped
// product straight into the plane layout with the residual - no transposition passes - and the 2x2 stride-2 (de)convolutions read / write
// the padded layout.  `mdx23_graph_plain` (fp32 planes, staged kernels) remains for networks without the images (fp32 precision, odd channel counts).
#include "conv_x3_dev.h"
#include "model_common.h"
#include "models.h"

namespace rvc {

struct TfcBlock {
  int in_c = 0, c = 0, f = 0;
  DevVec n1g, n1b, t0g, t0b, t3g, t3b, n2g, n2b;
  ConvLayer tfc1, tfc2, shortcut, lin1, lin2;
  bool fused_sc = false;    // padded graph: the shortcut's weight image is appended to tfc2's - tfc2(x2) + shortcut(x) is ONE product over two images
};
struct MdxScale {
  std::vector<TfcBlock> blocks;
  DevVec ng, nb;            // norm in front of the down / up convolution
  ConvLayer rs;             // the 2x2 stride-2 (transposed) convolution as a k = 1 GEMM over re-arranged data
};
struct Mdx23 {
  Ctx* ctx = nullptr;
  // activations: one arena (+ the "images are zeroed" record) per chunk stream.  Lane 0 runs on the caller's stream; demix may alternate a clip's chunks over further
  // lanes with streams of their own (they are independent until the overlap-add): the small launches of the deep levels of one chunk run beside the chip-filling ones of another
  static constexpr int kMaxLanes = 4;
  struct Lane { Arena arena; const char* img_base = nullptr; size_t img_bytes = 0; unsigned img_gen = 0; hipStream_t st = nullptr; hipEvent_t done = nullptr; float* acc = nullptr; size_t acc_n = 0; };
  Lane lane[kMaxLanes];
  hipEvent_t ev_start = nullptr;
  int streams = 0;                           // chunk streams of demix (0: RVC_MDX_STREAMS or 1), mdx23_set_streams
  TensorStore ts;
  bool ready = false;
  rvc_mdx23_config cfg{};
  ConvLayer stft, istft, first, fin0, fin2;
  DevVec window;
  std::vector<MdxScale> enc, dec;
  MdxScale bott;
  bool pad_ok = false;                       // every 3x3 / 2x2 layer has its bf16x3 image: the padded split-resident graph (mdx23_graph_padded)
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
void mdx23_destroy(Mdx23* M) {
  if (!M) return;
  mdx23_free(*M);
  for (auto& Ln : M->lane) {
    if (Ln.st) { (void)hipStreamSynchronize(Ln.st); (void)hipStreamDestroy(Ln.st); }
    if (Ln.done) (void)hipEventDestroy(Ln.done);
    if (Ln.acc) (void)hipFree(Ln.acc);
    Ln.arena.release();
  }
  if (M->ev_start) (void)hipEventDestroy(M->ev_start);
  delete M;
}

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
  {
    static const bool off = (knob_int("RVC_MDX_X3S", 1) == 0);
    bool ok = !off && conv_x3_enabled();
    auto blocks_ok = [&](const std::vector<TfcBlock>& bs) { for (const TfcBlock& B : bs) ok = ok && conv_x3s_eligible(B.tfc1) && conv_x3s_eligible(B.tfc2); };
    for (auto& S : M->enc) { blocks_ok(S.blocks); ok = ok && conv_x3s_eligible(S.rs); }
    for (auto& S : M->dec) { blocks_ok(S.blocks); ok = ok && conv_x3s_eligible(S.rs); }
    blocks_ok(M->bott.blocks);
    M->pad_ok = ok; for (auto& Ln : M->lane) Ln.img_base = nullptr;
    static const bool fuse = (knob_int("RVC_MDX_FUSE_SC", 1) != 0);
    auto fuse_blocks = [&](std::vector<TfcBlock>& bs) {
      for (TfcBlock& B : bs)
        if (conv_x3s_eligible(B.shortcut) && B.shortcut.CoPx == B.tfc2.CoPx) { conv_layer_append_x3(B.tfc2, B.shortcut); B.fused_sc = true; }
    };
    if (ok && fuse) {
      for (auto& S : M->enc) fuse_blocks(S.blocks);
      for (auto& S : M->dec) fuse_blocks(S.blocks);
      fuse_blocks(M->bott.blocks);
    }
  }
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
__global__ void ola_kernel(const float* __restrict__ fr, const float* __restrict__ w, float* __restrict__ out, int n_fft, int hop, int T, long long L, long long ldf,
                           int accumulate) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= L) return;
  const long long p = i + n_fft / 2;
  long long m1 = p / hop; if (m1 > T - 1) m1 = T - 1;
  long long m0 = (p - n_fft + hop) / hop; if (p - n_fft + 1 < 0) m0 = 0; if (m0 < 0) m0 = 0;
  float s = 0.f, e = 0.f;
  for (long long m = m0; m <= m1; ++m) {
    const long long j = p - m * hop;
    if (j < 0 || j >= n_fft) continue;
    s += fr[j * ldf + m]; const float ww = w[j]; e = fmaf(ww, ww, e);
  }
  float v = s / e;
  if (accumulate) { if (v != v) v = 0.f; v += out[i]; }      // demix_mdxv3: X[..., window] += nan_to_num(chunk), chunks in stream order
  out[i] = v;
}


// ---------------------------------------------------------------------------------------------- padded split-resident graph (round 4)
// branch-free exact-erf GELU, the form conv_x3s.hip's epilogue evaluates (Abramowitz & Stegun 7.1.26, |error| <= 1.5e-7 absolute): the image producers
// below run at HBM rate only if the activation stays under ~20 VALU instructions per value (ocml's erff: two divergent paths)
__device__ __forceinline__ float gelu_as(float v) {
  const float x = v * 0.70710678118654752440f, ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.f));
  const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * x * x);
  return 0.5f * v * (1.f + copysignf(fmaf(-poly, e, 1.f), x));
}
// 8 values of one position -> one 16-byte row of the hi and of the lo plane; g = group of 8 rows = chunk * 2 + half, row = margin + position
__device__ __forceinline__ void split8_store(const float (&v)[8], unsigned char* __restrict__ img, long long tp, long long row, int g) {
  u32x4 hi, lo;
#pragma unroll
  for (int j = 0; j < 4; ++j) { unsigned h_, l_; split2(v[2 * j], v[2 * j + 1], h_, l_); hi[j] = h_; lo[j] = l_; }
  unsigned char* r = img + (((long long)(g >> 1) * 4 + (g & 1)) * tp + row) * 16;
  *reinterpret_cast<u32x4*>(r) = hi;
  *reinterpret_cast<u32x4*>(r + tp * 32) = lo;
}
// InstanceNorm statistics, stage 1, over ROWS: channel c = blockIdx.x owns `rows` rows of Wv contiguous values, value (j, w) at x[c cs + j rs + w] -
// padded planes [C][H][W + 2] (x + 1, cs = H (W + 2), rs = W + 2: the pad columns are never read, whatever they hold) and the first TDF linear's
// output [f / bn][C][H] (cs = H, rs = C H).  Part q = blockIdx.y takes a run of whole rows; four independent loads per thread and trip.
__global__ __launch_bounds__(256) void inorm_part_rows_kernel(const float* __restrict__ x, long long cs, long long rs, int rows, int Wv, int wshift, int P,
                                                              double* __restrict__ part) {
  const int c = blockIdx.x, q = blockIdx.y;
  const int rpp = (rows + P - 1) / P, j0 = q * rpp, j1 = j0 + rpp < rows ? j0 + rpp : rows;
  const long long n = j1 > j0 ? (long long)(j1 - j0) * Wv : 0;
  const float* xc = x + (long long)c * cs + (long long)j0 * rs;
  auto at = [&](long long i) -> float { const long long j = wshift >= 0 ? (i >> wshift) : (i / Wv); return xc[j * rs + (i - j * Wv)]; };
  double s = 0.0, qq = 0.0;
  long long i = threadIdx.x;
  for (; i + 768 < n; i += 1024) {
    const float v0 = at(i), v1 = at(i + 256), v2 = at(i + 512), v3 = at(i + 768);
    s += ((double)v0 + (double)v1) + ((double)v2 + (double)v3);
    qq += ((double)v0 * v0 + (double)v1 * v1) + ((double)v2 * v2 + (double)v3 * v3);
  }
  for (; i < n; i += 256) { const float v = at(i); s += v; qq += (double)v * v; }
  __shared__ double ss[256], sq[256];
  ss[threadIdx.x] = s; sq[threadIdx.x] = qq;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { ss[threadIdx.x] += ss[threadIdx.x + o]; sq[threadIdx.x] += sq[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { part[((long long)c * P + q) * 2] = ss[0]; part[((long long)c * P + q) * 2 + 1] = sq[0]; }
}
// The fold of inorm_fold_kernel inside the consumer (same order, same arithmetic: bit-identical): scale / shift of the channels [c0, c0 + nch) from the partial sums
// into LDS - 4700 launches of a 5-us kernel per clip were 3 % of the separation.  Ends with a barrier.
struct NormStat { const double* part; int P; double n; const float* gamma; const float* beta; int C; };
constexpr int kNormParts = 16;                                // most parts per channel (stats_part)
__device__ __forceinline__ void fold_to_lds(const NormStat& st, int c0, int nch, float* __restrict__ s_a, float* __restrict__ s_b) {
  __shared__ double s_part[8 * kNormParts * 2];
  const bool par = nch <= 8 && blockDim.x >= 8 * kNormParts;   // the loads of 8 channels' parts in one go; the sums below keep the part order
  if (par) {
    const int i = (int)threadIdx.x / st.P, q = (int)threadIdx.x - i * st.P;
    if (i < nch && c0 + i < st.C) {
      const double2 v = *reinterpret_cast<const double2*>(st.part + ((long long)(c0 + i) * st.P + q) * 2);
      s_part[(i * kNormParts + q) * 2] = v.x; s_part[(i * kNormParts + q) * 2 + 1] = v.y;
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < nch; i += blockDim.x) {
    const int c = c0 + i;
    float a = 0.f, b = 0.f;
    if (c < st.C) {
      double s = 0.0, q = 0.0;
      if (par) for (int p = 0; p < st.P; ++p) { s += s_part[(i * kNormParts + p) * 2]; q += s_part[(i * kNormParts + p) * 2 + 1]; }
      else for (int p = 0; p < st.P; ++p) { s += st.part[((long long)c * st.P + p) * 2]; q += st.part[((long long)c * st.P + p) * 2 + 1]; }
      const double m = s / st.n, var = q / st.n - m * m;
      a = st.gamma[c] * (float)(1.0 / sqrt((var > 0.0 ? var : 0.0) + (double)1e-5f));
      b = st.beta[c] - (float)m * a;
    }
    s_a[i] = a; s_b[i] = b;
  }
  __syncthreads();
}
// gelu(x a + b) written as the split image (one thread = one position x 8 rows, consecutive threads consecutive positions).
// COLCH false: the norm channel is the ROW - planes [rows][T], padded rows of padw positions whose two pad columns are written as zeros (the 3x3
// convolution's horizontal zero padding).  COLCH true: the norm channel is the column group t / chdiv (the first TDF linear's output [f / bn][c H]).
// RAW: the plane as it is (no norm, no activation: the shortcut's input).
template <bool COLCH, bool RAW = false>
__global__ __launch_bounds__(256) void inorm_apply_split_kernel(const float* __restrict__ x, long long ld, int rows, long long T, const NormStat st, int chdiv, int padw,
                                                                unsigned char* __restrict__ img, long long tp, int margin) {
  __shared__ float s_a[COLCH ? 72 : 8], s_b[COLCH ? 72 : 8];
  const long long t0 = (long long)blockIdx.x * 256, t = t0 + threadIdx.x;
  const int g = blockIdx.y;
  const int cfirst = COLCH ? (int)((unsigned)t0 / (unsigned)chdiv) : g * 8;
  if (!RAW) fold_to_lds(st, cfirst, COLCH ? (int)((unsigned)(t0 + 255) / (unsigned)chdiv) - cfirst + 1 : 8, s_a, s_b);      // (host: 256 / chdiv + 2 <= 72)
  if (t >= T) return;
  bool pad = false;
  if (padw > 0) { const unsigned w = (unsigned)t % (unsigned)padw; pad = (w == 0u || w == (unsigned)padw - 1u); }
  float a = 0.f, b = 0.f;
  if (COLCH) { const int c = (int)((unsigned)t / (unsigned)chdiv) - cfirst; a = s_a[c]; b = s_b[c]; }
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int row = g * 8 + j;
    v[j] = 0.f;
    if (!pad && row < rows) {
      if (RAW) { v[j] = x[(long long)row * ld + t]; continue; }
      if (!COLCH) { a = s_a[j]; b = s_b[j]; }
      v[j] = gelu_as(fmaf(x[(long long)row * ld + t], a, b));
    }
  }
  split8_store(v, img, tp, margin + t, g);
}
// The first TDF linear contracts over the bins of a row: it takes the image of the TRANSPOSED plane - chunk = 16 neighbouring bins, position = row
// r = c H + h.  32 rows x 64 bins per workgroup: coalesced reads along the rows, turned through LDS, 16-byte image rows written along r.
__global__ __launch_bounds__(256) void inorm_apply_tm_kernel(const float* __restrict__ x, int Wp, int W, int R, int H, int hshift, const NormStat st,
                                                             unsigned char* __restrict__ img, long long tp) {
  __shared__ float tile[32][65];
  __shared__ float s_a[32], s_b[32];
  const int w0 = blockIdx.x * 64, r0 = blockIdx.y * 32;
  const int cfirst = r0 / H;
  fold_to_lds(st, cfirst, (r0 + 31) / H - cfirst + 1, s_a, s_b);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int ri = i * 4 + (threadIdx.x >> 6), wi = threadIdx.x & 63, r = r0 + ri;
    float v = 0.f;
    if (r < R && w0 + wi < W) { const int c = (hshift >= 0 ? (r >> hshift) : (r / H)) - cfirst; v = gelu_as(fmaf(x[(long long)r * Wp + 1 + w0 + wi], s_a[c], s_b[c])); }
    tile[ri][wi] = v;
  }
  __syncthreads();
  const int ri = threadIdx.x & 31, g8 = threadIdx.x >> 5, r = r0 + ri;
  if (r < R && w0 + g8 * 8 < W) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = tile[ri][g8 * 8 + j];
    split8_store(v, img, tp, (long long)kSplitMargin + r, (w0 >> 3) + g8);
  }
}
// InstanceNorm + GELU + space-to-depth in front of the 2x2 stride-2 convolution: the image of the [4 C][H / 2][W / 2 + 2] tensor whose row
// ci * 4 + dy * 2 + dx is the (dy, dx) phase of channel ci (the weight's own memory order); one thread = one half-resolution position x 2 channels
__global__ __launch_bounds__(256) void inorm_apply_s2d_split_kernel(const float* __restrict__ x, long long TPin, int Wp, int C, int Ho, int Wo, const NormStat st,
                                                                    unsigned char* __restrict__ img, long long tp, int margin) {
  __shared__ float s_a[2], s_b[2];
  const int Wpo = Wo + 2;
  const long long P = (long long)Ho * Wpo, p = (long long)blockIdx.x * 256 + threadIdx.x;
  const int g = blockIdx.y;
  fold_to_lds(st, 2 * g, 2, s_a, s_b);
  if (p >= P) return;
  const int y = (int)(p / Wpo), xq = (int)(p - (long long)y * Wpo);
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (xq != 0 && xq != Wpo - 1) {
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      const int ch = 2 * g + cc;
      if (ch < C) {
        const float a = s_a[cc], b = s_b[cc];
        const float* src = x + (long long)ch * TPin + (long long)(2 * y) * Wp + 1 + 2 * (xq - 1);
        v[cc * 4 + 0] = gelu_as(fmaf(src[0], a, b)); v[cc * 4 + 1] = gelu_as(fmaf(src[1], a, b));
        v[cc * 4 + 2] = gelu_as(fmaf(src[Wp], a, b)); v[cc * 4 + 3] = gelu_as(fmaf(src[Wp + 1], a, b));
      }
    }
  }
  split8_store(v, img, tp, margin + p, g);
}
// depth-to-space behind the 2x2 stride-2 transposed convolution, padded in and out: out[co][2 y + dy][1 + 2 x + dx] = in[co * 4 + dy * 2 + dx][y][1 + x],
// as the fp32 plane and (img != null) as its raw image - the first half of the decoder block's concatenated input.  One thread = one position x 8 channels.
__global__ __launch_bounds__(256) void d2s_pad_split_kernel(const float* __restrict__ in, float* __restrict__ out, int Co, int H, int W, unsigned char* __restrict__ img,
                                                            long long tp, int margin) {
  const int Wpi = W + 2, Wpo = 2 * W + 2, Ho = 2 * H;
  const long long TPo = (long long)Ho * Wpo, p = (long long)blockIdx.x * 256 + threadIdx.x;
  const int g = blockIdx.y;
  if (p >= TPo) return;
  const int yo = (int)(p / Wpo), xq = (int)(p - (long long)yo * Wpo);
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (xq != 0 && xq != Wpo - 1) {
    const int xo = xq - 1;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int co = g * 8 + j;
      if (co < Co) v[j] = in[(((long long)co * 4 + (yo & 1) * 2 + (xo & 1)) * H + (yo >> 1)) * Wpi + 1 + (xo >> 1)];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) if (g * 8 + j < Co) out[(long long)(g * 8 + j) * TPo + p] = v[j];
  if (img) split8_store(v, img, tp, margin + p, g);
}
// small levels (linears without a weight image): padded plane -> [W][R] with InstanceNorm + GELU, and [W][R] -> padded plane with the residual
__global__ void tr2d_in_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int W, int Wp, int H, const float* __restrict__ sc, const float* __restrict__ sh) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int j = threadIdx.y; j < 32; j += blockDim.y) {
    const int r = r0 + j, c = c0 + threadIdx.x;
    float v = 0.f;
    if (r < R && c < W) { const int ch = r / H; v = gelu_as(fmaf(in[(long long)r * Wp + 1 + c], sc[ch], sh[ch])); }
    tile[j][threadIdx.x] = v;
  }
  __syncthreads();
  for (int j = threadIdx.y; j < 32; j += blockDim.y) {
    const int c = c0 + j, r = r0 + threadIdx.x;
    if (c < W && r < R) out[(long long)c * R + r] = tile[threadIdx.x][j];
  }
}
__global__ void tr2d_out_kernel(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ res, int W, int R, int Wp) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;      // in [W rows][R columns]: c = column of in = plane row
  for (int j = threadIdx.y; j < 32; j += blockDim.y) {
    const int r = r0 + j, c = c0 + threadIdx.x;
    tile[j][threadIdx.x] = (r < W && c < R) ? in[(long long)r * R + c] : 0.f;
  }
  __syncthreads();
  for (int j = threadIdx.y; j < 32; j += blockDim.y) {
    const int c = c0 + j, r = r0 + threadIdx.x;
    if (c < R && r < W) { const long long o = (long long)c * Wp + 1 + r; out[o] = tile[threadIdx.x][j] + res[o]; }
  }
}

// The DFT products read an 8192 x 8192 basis (268 MB as the bf16 hi / lo image) for T = 256 columns: one product over BOTH channels' frames (and one over the
// four separated signals' spectra) reads it once instead of 2 (4) times.  frames2: out[j][a T + t] = reflect-padded audio[a][t hop + j - n_fft / 2];
// cols_unbatch: out[b][r][t] = in[r][b T + t]; cols_batch: out[r][b T + t] = in[b][r][t]
__global__ void frames2_kernel(const float* __restrict__ audio, long long ld, float* __restrict__ out, long long L, int n_fft, int hop, int T) {
  const long long n = (long long)n_fft * 2 * T, st = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += st) {
    const int j = (int)(i / (2 * T)), at = (int)(i - (long long)j * 2 * T), a = at / T, t = at - a * T;
    long long x = (long long)t * hop + j - n_fft / 2;
    if (x < 0) x = -x;
    if (x >= L) x = 2 * (L - 1) - x;
    out[i] = audio[(long long)a * ld + x];
  }
}
__global__ void cols_unbatch_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int B, int T) {
  const long long n = (long long)rows * B * T, st = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += st) {
    const int t = (int)(i % T); long long q = i / T;
    const int r = (int)(q % rows), b = (int)(q / rows);
    out[i] = in[((long long)r * B + b) * T + t];
  }
}
__global__ void cols_batch_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int B, int T) {
  const long long n = (long long)rows * B * T, st = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += st) {
    const int t = (int)(i % T); long long q = i / T;
    const int b = (int)(q % B), r = (int)(q / B);
    out[i] = in[((long long)b * rows + r) * T + t];
  }
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

// where a chunk comes from and goes to: rows of pitch ld_in / ld_out (a window of the whole padded mix / of the accumulator of demix_mdxv3's overlap-add,
// or a free-standing chunk); accumulate: out += result with NaN as zero (upstream's nan_to_num in front of the accumulation)
struct MdxIO { long long ld_in, ld_out; int accumulate; };
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

static void mdx23_graph_plain(Mdx23* M, hipStream_t s, Arena& A, const float* audio, long long L, float* out, const MdxIO& io) {
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
    frames(s, audio + (long long)a * io.ld_in, fr, (int)L, c.n_fft, c.hop, c.n_fft / 2, T, 1);
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
    hipLaunchKernelGGL(ola_kernel, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, s, fr, M->window.p, out + (long long)q * io.ld_out, c.n_fft, c.hop, T, L, (long long)T, io.accumulate);
  }
}


// ---------------------------------------------------------------------------------------------- the padded graph
static int ilog2_exact(int v) { if (v <= 0 || (v & (v - 1))) return -1; int k = 0; while ((1 << k) < v) ++k; return k; }
// statistics, stage 1: value (j, w) of channel c at x[c cs + j rs + w], j < rows, w < Wv.  The consumer folds the parts itself (fold_to_lds); the parts live
// in the stream's scratch until the next call.
static NormStat stats_part(hipStream_t s, const float* x, long long cs, long long rs, int C, int rows, int Wv, const float* g, const float* b) {
  const long long per_c = (long long)rows * Wv;
  int P = std::min(kNormParts, std::max(1, 2048 / C));
  if (P > rows) P = rows;
  while (P > 1 && per_c / P < 4096) --P;
  double* part = (double*)stream_scratch(s, 12, (size_t)C * P * 2 * sizeof(double));
  hipLaunchKernelGGL(inorm_part_rows_kernel, dim3((unsigned)C, (unsigned)P), dim3(256), 0, s, x, cs, rs, rows, Wv, ilog2_exact(Wv), P, part);
  return NormStat{part, P, (double)per_c, g, b, C};
}
static void stats_fold(hipStream_t s, const NormStat& st, float* sc, float* sh) {      // (the fp32 consumers of the small levels take folded vectors)
  hipLaunchKernelGGL(inorm_fold_kernel, dim3((unsigned)((st.C + 127) / 128)), dim3(128), 0, s, st.part, st.C, st.P, st.n, st.gamma, st.beta, 1e-5f, sc, sh);
}
// ia: the conv-input image (norm + GELU applied); rin / rmid: raw images of a block's input / output (the fused shortcut's operand), ir: raw image of the
// decoder's concatenated input [up-sampled | skip].  All four are neighbours in one allocation: the fused product addresses the raw image as an offset of ia.
struct PadLv { int C = 0, H = 0, W = 0, Wp = 0; long long TP = 0, tp = 0; SplitGeom g, g1; unsigned char *ia = nullptr, *rin = nullptr, *rmid = nullptr, *ir = nullptr; };
static size_t pad_img_bytes(int chans, const PadLv& L) { return (size_t)((chans + 15) / 16) * 4 * (size_t)L.tp * 16; }
// plane [rows][H (W + 2)] -> the conv-input image of the level (pad columns zero)
static void apply_plane(hipStream_t s, const float* x, int rows, const PadLv& L, const NormStat& st) {
  hipLaunchKernelGGL(inorm_apply_split_kernel<false>, dim3((unsigned)((L.TP + 255) / 256), (unsigned)((rows + 7) / 8)), dim3(256), 0, s, x, L.TP, rows, L.TP, st, 1, L.Wp,
                     L.ia, L.tp, L.g.margin);
}

// TFC_TDF.forward (tfc_tdf.py:137-144) over padded planes [c][H][W + 2]; the last block writes to `out`.  il / im: images of the two linears' inputs.
// raw_in: raw image of x (null: none - the shortcuts run as launches of their own); raw_out: where the last block leaves the raw image of its output (or null).
static void run_tfc_padded(const std::vector<TfcBlock>& blocks, hipStream_t s, const MdxScratch& K, const PadLv& L, unsigned char* il, unsigned char* im, const float* x,
                           const unsigned char* raw_in, float* out, unsigned char* raw_out, float* mid) {
  const int H = L.H, W = L.W, Wp = L.Wp;
  const long long TP = L.TP;
  ConvEpilogue E0;
  const float* cur = x;
  const unsigned char* raw_cur = raw_in;
  for (size_t i = 0; i < blocks.size(); ++i) {
    const TfcBlock& B = blocks[i];
    const bool last = i + 1 == blocks.size();
    float* dst = last ? out : mid;
    const bool fused = B.fused_sc && raw_cur != nullptr;
    RVC_REQUIRE(fused || !B.fused_sc, "run_tfc_padded: a block with a fused shortcut needs the raw image of its input");
    if (!fused) conv1d_run(B.shortcut, s, cur, TP, (int)TP, K.sbuf, TP, E0);                    // s = shortcut(x) (pad columns: whatever, zeroed with tfc2's)
    apply_plane(s, cur, B.in_c, L, stats_part(s, cur + 1, TP, Wp, B.in_c, H, W, B.n1g.p, B.n1b.p));
    conv_x3s_run(B.tfc1, s, L.ia, L.tp, (int)TP, K.t2, TP, E0, &L.g);                           // x1 = tfc1(x)
    // x2 = x1 + tdf(x1)
    const int R = B.c * H, fb = B.lin1.Co;
    const NormStat st0 = stats_part(s, K.t2 + 1, TP, Wp, B.c, H, W, B.t0g.p, B.t0b.p);
    float* x2;
    if (il && conv_x3s_eligible(B.lin1) && conv_x3s_eligible(B.lin2) && (W & 15) == 0 && (fb & 15) == 0 && H >= 4) {
      const long long tpl = split_image_tp(R);
      hipLaunchKernelGGL(inorm_apply_tm_kernel, dim3((unsigned)((W + 63) / 64), (unsigned)((R + 31) / 32)), dim3(256), 0, s, K.t2, Wp, W, R, H, ilog2_exact(H), st0, il, tpl);
      conv_x3s_run(B.lin1, s, il, tpl, R, K.t1, R, E0);                                         // [W / bn][R]
      const NormStat st3 = stats_part(s, K.t1, H, R, B.c, fb, H, B.t3g.p, B.t3b.p);             // channel c owns the columns [c H, (c + 1) H) of every row
      hipLaunchKernelGGL(inorm_apply_split_kernel<true>, dim3((unsigned)((R + 255) / 256), (unsigned)((fb + 7) / 8)), dim3(256), 0, s, K.t1, (long long)R, fb, (long long)R,
                         st3, H, 0, im, tpl, kSplitMargin);
      conv_x3s_run_swapped(B.lin2, 0, W, s, im, tpl, R, nullptr, 0, K.t3 + 1, Wp, K.t2 + 1, Wp);  // the swapped product lands in the plane layout, + x1
      x2 = K.t3;
    } else {
      stats_fold(s, st0, K.sc, K.sh);
      hipLaunchKernelGGL(tr2d_in_kernel, dim3((unsigned)((W + 31) / 32), (unsigned)((R + 31) / 32)), dim3(32, 8), 0, s, K.t2, K.t3, R, W, Wp, H, K.sc, K.sh);   // [W][R]
      conv1d_run(B.lin1, s, K.t3, R, R, K.t1, R, E0);
      norm_gelu(s, K.t1, K.t1, fb, B.c, H, B.t3g.p, B.t3b.p, K.sc, K.sh);
      conv1d_run(B.lin2, s, K.t1, R, R, K.t3, R, E0);                                           // [W][R]
      hipLaunchKernelGGL(tr2d_out_kernel, dim3((unsigned)((R + 31) / 32), (unsigned)((W + 31) / 32)), dim3(32, 8), 0, s, K.t3, K.t1, K.t2, W, R, Wp);
      x2 = K.t1;
    }
    apply_plane(s, x2, B.c, L, stats_part(s, x2 + 1, TP, Wp, B.c, H, W, B.n2g.p, B.n2b.p));
    ConvEpilogue Er;
    SplitGeom g2 = L.g;
    if (fused) g2.seg2_off = (long long)(raw_cur - L.ia); else { Er.R = K.sbuf; Er.ldR = TP; }
    unsigned char* raw_dst = last ? raw_out : (raw_in ? ((i & 1) ? L.rin : L.rmid) : nullptr);  // (rin is free once block 0 has read it; the decoder's raw_in is ir)
    if (raw_dst) { Er.ys_out = raw_dst; Er.ys_tp = L.tp; }
    conv_x3s_run(B.tfc2, s, L.ia, L.tp, (int)TP, dst, TP, Er, &g2);                             // tfc2(x2) + shortcut(x), pad columns zero
    cur = dst; raw_cur = raw_dst;
  }
}

static void mdx23_graph_padded(Mdx23* M, Mdx23::Lane& Ln, hipStream_t s, Arena& A, const float* audio, long long L, float* out, const MdxIO& io) {
  const rvc_mdx23_config& c = M->cfg;
  const bool dry = A.dry;
  const int T = c.dim_t, k = c.num_subbands, f0 = c.dim_f / k, n = c.num_scales, g = c.growth, S = c.num_targets, bn = c.bottleneck;
  const int dim_c = k * 4, c0 = c.num_channels;
  ConvEpilogue E0;
  const long long FT = (long long)f0 * T;
  std::vector<PadLv> lv((size_t)n + 1);
  for (int i = 0; i <= n; ++i) {
    PadLv& P = lv[(size_t)i];
    P.C = c0 + i * g; P.H = T >> i; P.W = f0 >> i; P.Wp = P.W + 2; P.TP = (long long)P.H * P.Wp;
    P.g = split_geom_2d(P.W); P.g1 = P.g; P.g1.ktaps = 1; P.g1.toff[0] = 0;
    P.tp = ((long long)P.g.margin + P.TP + std::max(704, P.g.margin) + 63) & ~63LL;            // rows per plane: both vertical paddings inside the plane
  }
  // ---- the conv-input images: a block of their own, margins (= the vertical zero padding) zeroed once per layout - nothing else ever writes there
  const size_t img0 = A.off;
  for (int i = 0; i <= n; ++i) {
    int chans = i < n ? 2 * lv[(size_t)i].C : lv[(size_t)i].C;                                  // decoder input [up-sampled | skip]
    if (i > 0) chans = std::max(chans, 4 * lv[(size_t)i - 1].C);                                // space-to-depth rows of the level above
    PadLv& P = lv[(size_t)i];
    P.ia = A.alloc<unsigned char>(pad_img_bytes(chans, P));
    P.rin = A.alloc<unsigned char>(pad_img_bytes(P.C, P));
    P.rmid = A.alloc<unsigned char>(pad_img_bytes(P.C, P));
    P.ir = i < n ? A.alloc<unsigned char>(pad_img_bytes(2 * P.C, P)) : nullptr;
    RVC_REQUIRE((double)pad_img_bytes(chans, P) + 2.0 * (double)pad_img_bytes(P.C, P) + (double)pad_img_bytes(2 * P.C, P) < 2147483648.0, "a level's images exceed 32-bit buffer addressing");
  }
  const size_t img_bytes = A.off - img0;
  if (!dry && (Ln.img_base != A.base + img0 || Ln.img_gen != A.gen || Ln.img_bytes != img_bytes)) {
    RVC_HIP_CHECK(hipMemsetAsync(A.base + img0, 0, img_bytes, s));
    Ln.img_base = A.base + img0; Ln.img_gen = A.gen; Ln.img_bytes = img_bytes;
  }
  // ---- images of the TDF linears' inputs (k = 1 products: no taps, margins never multiplied into a kept column)
  size_t il_bytes = 0, im_bytes = 0;
  for (int i = 0; i <= n; ++i) {
    const PadLv& P = lv[(size_t)i];
    if ((P.W & 15) || ((P.W / bn) & 15)) continue;
    const long long tpl = split_image_tp((long long)P.C * P.H);
    il_bytes = std::max(il_bytes, (size_t)(P.W / 16) * 4 * (size_t)tpl * 16);
    im_bytes = std::max(im_bytes, (size_t)(P.W / bn / 16) * 4 * (size_t)tpl * 16);
  }
  unsigned char* il = il_bytes ? A.alloc<unsigned char>(il_bytes) : nullptr;
  unsigned char* im = im_bytes ? A.alloc<unsigned char>(im_bytes) : nullptr;
  float* fr = A.alloc<float>((size_t)c.n_fft * T * (size_t)std::max(2, 2 * S));      // frames of both channels / of every separated signal
  float* spec = A.alloc<float>((size_t)4 * c.dim_f * T);          // [ch][re | im][dim_f][T] = cac2cws view [16][f0][T]
  float* first = A.alloc<float>((size_t)c0 * FT);
  size_t big = 0, curmax = 0;
  for (int i = 0; i <= n; ++i) { big = std::max(big, (size_t)2 * lv[(size_t)i].C * (size_t)lv[(size_t)i].TP); curmax = std::max(curmax, (size_t)lv[(size_t)i].C * (size_t)lv[(size_t)i].TP); }
  for (int i = 0; i < n; ++i) big = std::max(big, (size_t)4 * lv[(size_t)i].C * (size_t)lv[(size_t)i + 1].TP);     // phase rows of the transposed convolution
  big = std::max(big, (size_t)((long long)(c0 + dim_c) * FT));
  big = std::max(big, (size_t)((long long)S * dim_c * FT));        // the mask head's output [S dim_c][f0][T] lands in t1 as well
  MdxScratch K;
  K.t1 = A.alloc<float>(big + 64); K.t2 = A.alloc<float>(big + 64); K.t3 = A.alloc<float>(big + 64); K.sbuf = A.alloc<float>(big + 64);
  K.sc = A.alloc<float>((size_t)2 * lv[(size_t)n].C + 64); K.sh = A.alloc<float>((size_t)2 * lv[(size_t)n].C + 64);
  float* mid = A.alloc<float>(big + 64);
  std::vector<float*> cat((size_t)n);                             // decoder inputs [2 c_i][H_i][W_i + 2]: [up-sampled | encoder skip]
  for (int i = 0; i < n; ++i) cat[(size_t)i] = A.alloc<float>((size_t)2 * lv[(size_t)i].C * (size_t)lv[(size_t)i].TP + 64);
  float* x0 = A.alloc<float>((size_t)c0 * (size_t)lv[0].TP + 64);
  float* cur = A.alloc<float>(curmax + 64);
  float* cur2 = A.alloc<float>(curmax + 64);
  if (dry) return;
  // ---- STFT of both channels in one product: frames (reflect-padded by n_fft / 2) x windowed DFT matrix
  hipLaunchKernelGGL(frames2_kernel, dim3(gridn((long long)c.n_fft * 2 * T)), dim3(256), 0, s, audio, io.ld_in, fr, L, c.n_fft, c.hop, T);
  conv1d_run(M->stft, s, fr, 2 * T, 2 * T, K.t1, 2 * T, E0);                                    // [2 dim_f][(a, t)]
  hipLaunchKernelGGL(cols_unbatch_kernel, dim3(gridn((long long)4 * c.dim_f * T)), dim3(256), 0, s, K.t1, spec, 2 * c.dim_f, 2, T);
  conv1d_run(M->first, s, spec, FT, (int)FT, first, FT, E0);                                   // [c0][f0][T]
  transpose(s, first, x0 + 1, f0, T, T, lv[0].Wp, c0, FT, lv[0].TP);                            // [c0][T][f0 + 2]
  hipLaunchKernelGGL((inorm_apply_split_kernel<false, true>), dim3((unsigned)((lv[0].TP + 255) / 256), (unsigned)((c0 + 7) / 8)), dim3(256), 0, s, x0, lv[0].TP, c0, lv[0].TP,
                     NormStat{}, 1, lv[0].Wp, lv[0].rin, lv[0].tp, lv[0].g.margin);
  // ---- encoder
  const float* h = x0;
  for (int i = 0; i < n; ++i) {
    const PadLv& P = lv[(size_t)i]; const PadLv& Q = lv[(size_t)i + 1];
    float* skip = cat[(size_t)i] + (size_t)P.C * (size_t)P.TP;
    run_tfc_padded(M->enc[(size_t)i].blocks, s, K, P, il, im, h, P.rin, skip, P.ir + pad_img_bytes(P.C, P), mid);
    const NormStat std_ = stats_part(s, skip + 1, P.TP, P.Wp, P.C, P.H, P.W, M->enc[(size_t)i].ng.p, M->enc[(size_t)i].nb.p);
    hipLaunchKernelGGL(inorm_apply_s2d_split_kernel, dim3((unsigned)((Q.TP + 255) / 256), (unsigned)((4 * P.C + 7) / 8)), dim3(256), 0, s, skip, P.TP, P.Wp, P.C, Q.H, Q.W, std_,
                       Q.ia, Q.tp, Q.g.margin);
    ConvEpilogue Ed; Ed.ys_out = Q.rin; Ed.ys_tp = Q.tp;
    conv_x3s_run(M->enc[(size_t)i].rs, s, Q.ia, Q.tp, (int)Q.TP, cur, Q.TP, Ed, &Q.g1);          // [C + g][H / 2][W / 2 + 2], plane + raw image
    h = cur; std::swap(cur, cur2);
  }
  {
    float* o = cur;
    run_tfc_padded(M->bott.blocks, s, K, lv[(size_t)n], il, im, h, lv[(size_t)n].rin, o, nullptr, mid);
    h = o; std::swap(cur, cur2);
  }
  // ---- decoder
  for (int i = 0; i < n; ++i) {
    const int lo = n - 1 - i;                                       // output level
    const PadLv& Pi = lv[(size_t)lo + 1]; const PadLv& Po = lv[(size_t)lo];
    apply_plane(s, h, Pi.C, Pi, stats_part(s, h + 1, Pi.TP, Pi.Wp, Pi.C, Pi.H, Pi.W, M->dec[(size_t)i].ng.p, M->dec[(size_t)i].nb.p));
    conv_x3s_run(M->dec[(size_t)i].rs, s, Pi.ia, Pi.tp, (int)Pi.TP, K.t2, Pi.TP, E0, &Pi.g1);   // [(co, dy, dx)][H][W + 2]
    hipLaunchKernelGGL(d2s_pad_split_kernel, dim3((unsigned)((Po.TP + 255) / 256), (unsigned)((Po.C + 7) / 8)), dim3(256), 0, s, K.t2, cat[(size_t)lo], Po.C, Pi.H, Pi.W, Po.ir,
                       Po.tp, Po.g.margin);
    float* o = cur;
    run_tfc_padded(M->dec[(size_t)i].blocks, s, K, Po, il, im, cat[(size_t)lo], Po.ir, o, nullptr, mid);
    h = o; std::swap(cur, cur2);
  }
  // ---- mask head: x^T * first_conv_out, cat(mix, x), 1x1 -> GELU -> 1x1
  float* hc = K.t3;                                                 // [dim_c + c0][f0][T]
  RVC_HIP_CHECK(hipMemcpyAsync(hc, spec, (size_t)dim_c * FT * sizeof(float), hipMemcpyDeviceToDevice, s));
  transpose(s, h + 1, K.t1, T, f0, lv[0].Wp, T, c0, lv[0].TP, FT);                             // [c0][f0][T]
  hipLaunchKernelGGL(mul_kernel, dim3(gridn((long long)c0 * FT)), dim3(256), 0, s, K.t1, first, hc + (size_t)dim_c * FT, (long long)c0 * FT);
  ConvEpilogue Eg; Eg.act = ACT_GELU;
  conv1d_run(M->fin0, s, hc, FT, (int)FT, K.t2, FT, Eg);
  conv1d_run(M->fin2, s, K.t2, FT, (int)FT, K.t1, FT, E0);                                     // [S * 16][f0][T] = [S][ch][re | im][dim_f][T]
  // ---- inverse STFT of every source and channel in one product
  const int Q = S * 2;
  hipLaunchKernelGGL(cols_batch_kernel, dim3(gridn((long long)Q * 2 * c.dim_f * T)), dim3(256), 0, s, K.t1, K.t2, 2 * c.dim_f, Q, T);      // [2 dim_f][(q, t)]
  conv1d_run(M->istft, s, K.t2, Q * T, Q * T, fr, Q * T, E0);                                    // [n_fft][(q, t)] windowed inverse FFT frames
  for (int q = 0; q < Q; ++q)
    hipLaunchKernelGGL(ola_kernel, dim3((unsigned)((L + 255) / 256)), dim3(256), 0, s, fr + (size_t)q * T, M->window.p, out + (long long)q * io.ld_out, c.n_fft, c.hop, T, L, (long long)Q * T, io.accumulate);
}

static void mdx23_chunk(Mdx23* M, int lane, hipStream_t s, const float* audio, long long L, float* out, const MdxIO& io) {
  Mdx23::Lane& Ln = M->lane[lane];
  Arena& A = Ln.arena;
  for (int pass = 0; pass < 2; ++pass) {
    A.dry = (pass == 0); A.reset(); if (pass == 0) A.peak = 0;
    if (M->pad_ok) mdx23_graph_padded(M, Ln, s, A, audio, L, out, io); else mdx23_graph_plain(M, s, A, audio, L, out, io);
    if (pass == 0) A.ensure(A.peak);
  }
  A.dry = false;
}
void mdx23_forward(Mdx23* M, hipStream_t s, const float* audio, long long L, float* out) {
  RVC_REQUIRE(M->ready, "mdx23_finalize has not been called");
  RVC_REQUIRE(L == (long long)M->cfg.hop * (M->cfg.dim_t - 1), "a chunk is hop * (dim_t - 1) samples per channel");
  mdx23_chunk(M, 0, s, audio, L, out, MdxIO{L, L, 0});
}
__global__ void div_kernel(float* __restrict__ x, long long n, float d) {
  const long long st = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += st) x[i] = x[i] / d;
}
__global__ void add_kernel(float* __restrict__ x, const float* __restrict__ y, long long n) {
  const long long st = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += st) x[i] += y[i];
}
// demix_mdxv3's chunk loop on the device (reference lib/karafan/inference.py:52-66): chunks of C = hop (dim_t - 1) samples every `step` samples of the zero-padded mix
// [2][Lp]; every chunk's separated signals are added into acc [S][2][Lp] at the chunk's offset (NaN as zero) and the sum is divided by `overlap` at the end.  The
// network reads its window of the mix in place and its inverse STFT adds into the accumulator: no chunk copies.
// One chunk stream (the default; what a server with several clips in flight wants - bench.py's lanes): chunks in order = the reference's own order of fp32 additions.
// mdx23_set_streams(K) / RVC_MDX_STREAMS=K (a single conversion alone on the GPU - the ComfyUI node sets 3: one clip 860 -> 700 ms; with three clips in flight it costs 4 - 6 %):
// K chunk streams: chunk c runs on stream c mod K with that stream's arena and accumulator, the K accumulators are added in stream order at the end - the same
// result on every run, an ulp-level re-association of the reference's sum.
void mdx23_demix(Mdx23* M, hipStream_t s, const float* mix, long long Lp, long long step, long long n_chunks, float overlap, float* acc) {
  RVC_REQUIRE(M->ready, "mdx23_finalize has not been called");
  const long long C = (long long)M->cfg.hop * (M->cfg.dim_t - 1);
  RVC_REQUIRE(step > 0 && n_chunks > 0 && (n_chunks - 1) * step + C <= Lp && overlap > 0.f, "demix: the last chunk must end inside the padded mix");
  static const int k_env = knob_int("RVC_MDX_STREAMS", 0);
  const int k_want = k_env > 0 ? k_env : (M->streams > 0 ? M->streams : 1);
  const int K = (int)std::min<long long>(std::max(1, std::min(k_want, (int)Mdx23::kMaxLanes)), n_chunks);
  const size_t n = (size_t)M->cfg.num_targets * 2 * (size_t)Lp;
  RVC_HIP_CHECK(hipMemsetAsync(acc, 0, n * sizeof(float), s));
  if (K > 1) {
    if (!M->ev_start) RVC_HIP_CHECK(hipEventCreateWithFlags(&M->ev_start, hipEventDisableTiming));
    RVC_HIP_CHECK(hipEventRecord(M->ev_start, s));
    for (int k = 1; k < K; ++k) {
      Mdx23::Lane& Ln = M->lane[k];
      if (!Ln.st) { RVC_HIP_CHECK(hipStreamCreateWithFlags(&Ln.st, hipStreamNonBlocking)); RVC_HIP_CHECK(hipEventCreateWithFlags(&Ln.done, hipEventDisableTiming)); }
      if (Ln.acc_n < n) { if (Ln.acc) (void)hipFree(Ln.acc); Ln.acc = nullptr; Ln.acc_n = 0; RVC_HIP_CHECK(hipMalloc(&Ln.acc, n * sizeof(float))); Ln.acc_n = n; }
      RVC_HIP_CHECK(hipStreamWaitEvent(Ln.st, M->ev_start, 0));          // (the mix is ready, the previous call's reads of this lane's accumulator are behind us)
      RVC_HIP_CHECK(hipMemsetAsync(Ln.acc, 0, n * sizeof(float), Ln.st));
    }
  }
  try {
    for (long long c = 0; c < n_chunks; ++c) {
      const int k = (int)(c % K);
      mdx23_chunk(M, k, k == 0 ? s : M->lane[k].st, mix + c * step, C, (k == 0 ? acc : M->lane[k].acc) + c * step, MdxIO{Lp, Lp, 1});
    }
  } catch (...) {
    // a chunk failed to enqueue: the lane streams may still be reading `mix` / writing their accumulators - the caller's stream waits for them before the
    // error reaches Python (which then hands both buffers back to torch's caching allocator)
    for (int k = 1; k < K; ++k)
      if (M->lane[k].st && M->lane[k].done && hipEventRecord(M->lane[k].done, M->lane[k].st) == hipSuccess) (void)hipStreamWaitEvent(s, M->lane[k].done, 0);
    throw;
  }
  for (int k = 1; k < K; ++k) {
    RVC_HIP_CHECK(hipEventRecord(M->lane[k].done, M->lane[k].st)); RVC_HIP_CHECK(hipStreamWaitEvent(s, M->lane[k].done, 0));
    hipLaunchKernelGGL(add_kernel, dim3(gridn((long long)n)), dim3(256), 0, s, acc, M->lane[k].acc, (long long)n);
  }
  hipLaunchKernelGGL(div_kernel, dim3(gridn((long long)n)), dim3(256), 0, s, acc, (long long)n, overlap);
}
void mdx23_set_streams(Mdx23* M, int k) { M->streams = k < 0 ? 0 : (k > Mdx23::kMaxLanes ? Mdx23::kMaxLanes : k); }
size_t mdx23_workspace(const Mdx23* M) { size_t b = 0; for (const auto& Ln : M->lane) b += Ln.arena.cap; return b; }

}  // namespace rvc
