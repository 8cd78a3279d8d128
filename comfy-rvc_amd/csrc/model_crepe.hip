// CREPE pitch network (torchcrepe.Crepe "full" / "tiny") as a HIP kernel graph: the f0 front-ends "crepe" / "mangio-crepe" of the
// reference (pitch_extraction.py:76-150) call torchcrepe.predict, a third-party package that is not vendored there and not available
// offline; what is restated is its published architecture and pre-processing (torchcrepe 0.0.23: core.py preprocess / infer,
// model.py Crepe):
//   frames of 1024 samples every `hop` (audio zero-padded by 512 on both sides), each frame minus its mean over its unbiased std
//   (floored at 1e-10); six layers  zero-pad -> Conv2d(k x 1) -> ReLU -> BatchNorm2d(eps 0.0010000000474974513) -> MaxPool(2 x 1)
//   with k = 512 / stride 4 / pad (254, 254) for the first and k = 64 / stride 1 / pad (31, 32) for the others; channels
//   1024,128,128,128,256,512 (full) or 128,16,16,16,32,64 (tiny); the [C6][4] map flattened position-major into Linear(4 C6 -> 360) and a sigmoid.
// Layout: channel-major [C][frames x positions].  Layer 1 is an im2col GEMM (K = 512 taps).  Layers 2-6 run on the 1-D convolution
// kernels over a "slot" layout: every frame's L positions sit in a slot of L + 63 columns behind 31 and in front of 32 zero columns, so
// that a plain k = 64 convolution over the concatenation sees exactly the per-frame zero padding; the 63 outputs per slot whose windows
// straddle two frames are computed and dropped (extra MFMA work (L + 63) / L, the price of reusing the dense 1-D kernels).
#include "model_common.h"
#include "models.h"

namespace rvc {

struct Crepe {
  Ctx* ctx = nullptr;
  Arena arena;
  TensorStore ts;
  bool ready = false, tiny = false;
  int ch[6] = {1024, 128, 128, 128, 256, 512};
  ConvLayer conv[6], fc;
  ConvLayer gemm[6];             // layers 5 and 6 (L = 16, 8 positions per frame) as GEMMs over an explicit im2col (K = 64 C_in)
  DevVec bn_a[6], bn_b[6];       // BatchNorm folded to y = a * x + b (applied after the ReLU, before the max-pool)
};

static const int kCrepeFull[6] = {1024, 128, 128, 128, 256, 512};
static const int kCrepeTiny[6] = {128, 16, 16, 16, 32, 64};

Crepe* crepe_create(Ctx* ctx, int tiny) {
  Crepe* M = new Crepe(); M->ctx = ctx; M->tiny = tiny != 0;
  for (int i = 0; i < 6; ++i) M->ch[i] = tiny ? kCrepeTiny[i] : kCrepeFull[i];
  return M;
}
void crepe_set_tensor(Crepe* M, const char* name, const float* d, const long long* shape, int ndim) { M->ts.set(name, d, shape, ndim); }
static void crepe_free(Crepe& M) {
  for (int i = 0; i < 6; ++i) { conv_layer_free(M.conv[i]); conv_layer_free(M.gemm[i]); M.bn_a[i].free_(); M.bn_b[i].free_(); }
  conv_layer_free(M.fc);
}
void crepe_destroy(Crepe* M) { if (M) { crepe_free(*M); M->arena.release(); delete M; } }

void crepe_finalize(Crepe* M) {
  const TensorStore& ts = M->ts;
  crepe_free(*M);
  ConvBuildScope x3scope(M->ctx->precision);
  int cin = 1;
  for (int i = 0; i < 6; ++i) {
    const int co = M->ch[i], k = i == 0 ? 512 : 64;
    const std::string n = "conv" + std::to_string(i + 1);
    const HostTensor& w = ts.get(n + ".weight", {co, cin, k, 1});
    const HostTensor& b = ts.get(n + ".bias", {co});
    if (i == 0) conv1d_layer_init(M->conv[0], w.data.data(), b.data.data(), co, 512, 1, 1, 0, 1, 1);   // Linear(512 taps -> C1) over im2col columns
    else if (i >= 4) conv1d_layer_init(M->gemm[i], w.data.data(), b.data.data(), co, cin * 64, 1, 1, 0, 1, 1);   // [Co][Ci][64] is already [Co][K]
    else conv1d_layer_init(M->conv[i], w.data.data(), b.data.data(), co, cin, 64, 1, 0, 1, 1);
    const HostTensor& g = ts.get(n + "_BN.weight", {co}); const HostTensor& be = ts.get(n + "_BN.bias", {co});
    const HostTensor& mu = ts.get(n + "_BN.running_mean", {co}); const HostTensor& var = ts.get(n + "_BN.running_var", {co});
    std::vector<float> a(co), sh(co);
    for (int c = 0; c < co; ++c) {
      const float inv = 1.f / std::sqrt(var.data[c] + 0.0010000000474974513f);
      a[c] = g.data[c] * inv; sh[c] = be.data[c] - mu.data[c] * g.data[c] * inv;
    }
    M->bn_a[i].upload(a); M->bn_b[i].upload(sh);
    cin = co;
  }
  const int F = 4 * M->ch[5];
  conv1d_layer_init(M->fc, ts.get("classifier.weight", {360, F}).data.data(), ts.get("classifier.bias", {360}).data.data(), 360, F, 1, 1, 0, 1, 1);
  M->ts.clear();
  M->ready = true;
}

long long crepe_num_frames(long long L, int hop, int pad) {
  if (hop <= 0) return 0;
  if (pad) return 1 + L / hop;
  return L >= 1024 ? 1 + (L - 1024) / hop : 0;
}

// ---------------------------------------------------------------------------------------------- kernels
// per-frame mean and 1 / max(1e-10, unbiased std) over the 1024 samples of the (zero-padded) frame; one wave per frame
__global__ __launch_bounds__(64) void crepe_stats_kernel(const float* __restrict__ audio, long long L, int hop, int off, int n, float* __restrict__ mean,
                                                        float* __restrict__ rstd) {
  const int f = blockIdx.x;
  if (f >= n) return;
  const long long g0 = (long long)f * hop - off;
  float v[16]; float s = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) { const long long g = g0 + threadIdx.x + 64 * j; v[j] = (g >= 0 && g < L) ? audio[g] : 0.f; s += v[j]; }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float m = s * (1.f / 1024.f);
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) { const float d = v[j] - m; q = fmaf(d, d, q); }
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
  if (threadIdx.x == 0) { mean[f] = m; rstd[f] = 1.f / fmaxf(1e-10f, sqrtf(q * (1.f / 1023.f))); }
}
// im2col of the first layer over normalised frames f0 .. f0 + B - 1: out[t][b * 256 + pos] = xn[b][4 pos + t - 254] (0 outside the frame)
__global__ void crepe_im2col_kernel(const float* __restrict__ audio, long long L, int hop, int off, const float* __restrict__ mean, const float* __restrict__ rstd,
                                    int f0, int B, float* __restrict__ out) {
  const long long N = (long long)B * 256;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long tot = 512 * N, st = (long long)gridDim.x * blockDim.x;
  for (; i < tot; i += st) {
    const int t = (int)(i / N); const long long c = i - (long long)t * N;
    const int b = (int)(c >> 8), pos = (int)(c & 255);
    const int j = 4 * pos + t - 254;
    float v = 0.f;
    if (j >= 0 && j < 1024) {
      const int f = f0 + b;
      const long long g = (long long)f * hop - off + j;
      const float x = (g >= 0 && g < L) ? audio[g] : 0.f;
      v = (x - mean[f]) * rstd[f];
    }
    out[i] = v;
  }
}
// BatchNorm (folded) + MaxPool(2) of a ReLU'd layer output, re-laid into the next layer's slot layout:
//   in  [C][ldin], frame b's position i at b * Sin + i            (i < Lin; Sin = Lin for the dense first layer, Lin + 63 afterwards)
//   out [C][B * Sout], Sout = Lin / 2 + 63, value at b * Sout + 31 + i', zeros in the 63 gap columns;
//   last layer (transposed != 0): out[(i' * C + c)][b] with row pitch ldout = B (the position-major flatten in front of the classifier).
__global__ void crepe_pool_kernel(const float* __restrict__ in, long long ldin, int Sin, int Lin, const float* __restrict__ a, const float* __restrict__ sh,
                                  float* __restrict__ out, int C, int B, int transposed) {
  const int Lo = Lin >> 1, Sout = transposed ? Lo : Lo + 63;
  const long long per = (long long)B * Sout, tot = per * C, st = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += st) {
    const int c = (int)(i / per); const long long r = i - (long long)c * per;
    const int b = (int)(r / Sout), u = (int)(r - (long long)b * Sout);
    const int io = transposed ? u : u - 31;
    float v = 0.f;
    if (io >= 0 && io < Lo) {
      const float* p = in + (long long)c * ldin + (long long)b * Sin + 2 * io;
      const float ac = a[c], bc = sh[c];
      v = fmaxf(fmaf(p[0], ac, bc), fmaf(p[1], ac, bc));
    }
    if (transposed) out[((long long)io * C + c) * B + b] = v;
    else out[i] = v;
  }
}

// ---------------------------------------------------------------------------------------------- Viterbi decoding (torchcrepe.decode.viterbi)
// Masked softmax over the bins [lo, hi) of every frame (float32, like the torch original), log with the float32 "tiny" floor
// (librosa.sequence.viterbi) - all frames in parallel - then one workgroup walks the frames: value[j] = logp[t][j] + max_k (value[k] + log A[k][j]) in float64 with the
// triangular transition matrix A[i][j] = max(12 - |i - j|, 0) / row sum, whose band |i - j| <= 11 is all that can win (out-of-band
// entries are log(2.2e-308) = -708 in librosa's dense form).  First maximum wins in ascending k, as np.argmax.  Back-pointers (int8
// offsets) go to global memory; the backtrack re-reads them 32 frames at a time through LDS.
// pass 1 (one wave per frame, all frames in parallel): logp[t][j] = log(softmax_j(masked probs)[j] + tiny32)
__global__ __launch_bounds__(64) void crepe_logsoftmax_kernel(const float* __restrict__ probs, int n, int lo, int hi, float* __restrict__ logp) {
  const int t = blockIdx.x, l = threadIdx.x;
  float p[6]; float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < 6; ++i) { const int j = l + 64 * i; p[i] = (j < 360 && j >= lo && j < hi) ? probs[(long long)j * n + t] : -INFINITY; mx = fmaxf(mx, p[i]); }
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float e[6], sm = 0.f;
#pragma unroll
  for (int i = 0; i < 6; ++i) { e[i] = p[i] == -INFINITY ? 0.f : expf(p[i] - mx); sm += e[i]; }
  for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o);
#pragma unroll
  for (int i = 0; i < 6; ++i) { const int j = l + 64 * i; if (j < 360) logp[(long long)t * 360 + j] = logf(e[i] / sm + 1.17549435e-38f); }
}
// pass 2 (one workgroup): the recurrence, one barrier per frame
__global__ __launch_bounds__(384) void crepe_viterbi_kernel(const float* __restrict__ probs, const float* __restrict__ logp, int n, signed char* __restrict__ ptr,
                                                           int* __restrict__ bins, float* __restrict__ per) {
  constexpr int NB = 360;
  __shared__ double val[2][NB + 22];
  __shared__ signed char pblk[32][NB];
  __shared__ int cur;
  const int j = threadIdx.x;
  double lt[23];
  if (j < NB) {
    for (int d = -11; d <= 11; ++d) {
      const int i = j + d;                      // source state
      if (i < 0 || i >= NB) { lt[d + 11] = -1e300; continue; }
      int rs = 0;
      for (int k = max(0, i - 11); k <= min(NB - 1, i + 11); ++k) rs += 12 - abs(i - k);
      lt[d + 11] = log((double)(12 - abs(d)) / (double)rs + 2.2250738585072014e-308);
    }
  }
  for (int q = j; q < 2 * (NB + 22); q += blockDim.x) (&val[0][0])[q] = -1e300;
  __syncthreads();
  float lp_next = j < NB ? logp[j] : 0.f;
  for (int t = 0; t < n; ++t) {
    const double lp = (double)lp_next;
    if (j < NB && t + 1 < n) lp_next = logp[(long long)(t + 1) * NB + j];          // next frame's row is in flight during this step
    if (j < NB) {
      double* vo = val[t & 1] + 11;
      if (t == 0) vo[j] = lp + log(1.0 / NB + 2.2250738585072014e-308);
      else {
        const double* vi = val[(t - 1) & 1] + 11;
        double best = -1e308; int bd = 0;
#pragma unroll
        for (int d = -11; d <= 11; ++d) { const double c = vi[j + d] + lt[d + 11]; if (c > best) { best = c; bd = d; } }
        vo[j] = lp + best;
        ptr[(long long)t * NB + j] = (signed char)bd;
      }
    }
    __syncthreads();
  }
  // arg-max of the last frame (first maximum), then the backtrack
  if (j == 0) {
    const double* v = val[(n - 1) & 1] + 11;
    int b = 0;
    for (int k = 1; k < NB; ++k) if (v[k] > v[b]) b = k;
    cur = b; bins[n - 1] = b; per[n - 1] = probs[(long long)b * n + (n - 1)];
  }
  __syncthreads();
  for (int t1 = n - 1; t1 >= 1; t1 -= 32) {
    const int cnt = min(32, t1);                               // frames t1, t1 - 1, ..., t1 - cnt + 1 hold the pointers into t - 1
    for (int q = j; q < cnt * NB; q += blockDim.x) { const int r = q / NB, c = q - r * NB; pblk[r][c] = ptr[(long long)(t1 - r) * NB + c]; }
    __syncthreads();
    if (j == 0) {
      int b = cur;
      for (int r = 0; r < cnt; ++r) { b += pblk[r][b]; const int t = t1 - r - 1; bins[t] = b; per[t] = probs[(long long)b * n + t]; }
      cur = b;
    }
    __syncthreads();
  }
}
void crepe_viterbi(hipStream_t s, const float* probs, int n, int lo, int hi, int* bins, float* per) {
  char* scr = (char*)stream_scratch(s, 3, (size_t)n * 360 * 5 + 256);
  float* logp = (float*)scr;
  signed char* ptr = (signed char*)(scr + (size_t)n * 360 * 4);
  hipLaunchKernelGGL(crepe_logsoftmax_kernel, dim3((unsigned)n), dim3(64), 0, s, probs, n, lo, hi, logp);
  hipLaunchKernelGGL(crepe_viterbi_kernel, dim3(1), dim3(384), 0, s, probs, logp, n, ptr, bins, per);
}

// im2col of a slot-layout tensor for the short late layers: col[(ci * 64 + t)][b * L + p] = x[ci][b * S + p + t]  (S = L + 63: the slots
// already hold the zero padding).  With L = 16 / 8 positions per frame the dense 1-D convolution over the slots would spend (L + 63) / L =
// 4.9x / 8.9x the MFMA work on columns that straddle two frames; the GEMM over K = 64 C_in does exactly the algorithmic work.
__global__ void crepe_slot_im2col_kernel(const float* __restrict__ x, long long ldx, int S, int L, int C, int B, float* __restrict__ col) {
  const long long N = (long long)B * L, tot = (long long)C * 64 * N, st = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += st) {
    const long long r = i / N, c = i - r * N;
    const int ci = (int)(r >> 6), t = (int)(r & 63), b = (int)(c / L), pp = (int)(c - (long long)b * L);
    col[i] = x[(long long)ci * ldx + (long long)b * S + pp + t];
  }
}

static int grid_for(long long n) { long long g = (n + 255) / 256; return (int)(g > 16384 ? 16384 : (g < 1 ? 1 : g)); }

// probabilities [360][n] (channel-major), frames in batches of at most `FB`
static void crepe_graph(Crepe* M, hipStream_t s, Arena& A, const float* audio, long long L, int hop, int pad, float* probs, long long n, const CrepeTaps* taps) {
  const bool dry = A.dry;
  const int off = pad ? 512 : 0;
  const int FB = (int)(n < 512 ? n : 512);
  float* mean = A.alloc<float>((size_t)n);
  float* rstd = A.alloc<float>((size_t)n);
  if (!dry) hipLaunchKernelGGL(crepe_stats_kernel, dim3((unsigned)n), dim3(64), 0, s, audio, L, hop, off, (int)n, mean, rstd);
  const size_t colsz = std::max((size_t)512 * FB * 256, std::max((size_t)M->ch[3] * 64 * FB * 16, (size_t)M->ch[4] * 64 * FB * 8));
  float* col = A.alloc<float>(colsz);
  size_t bufsz = (size_t)M->ch[0] * FB * 256;                      // largest activation: the dense first-layer output
  for (int i = 1; i < 6; ++i) bufsz = std::max(bufsz, (size_t)std::max(M->ch[i - 1], M->ch[i]) * FB * ((256 >> i) + 63));
  float* bufA = A.alloc<float>(bufsz + 64);
  float* bufB = A.alloc<float>(bufsz + 64);
  float* flat = A.alloc<float>((size_t)4 * M->ch[5] * FB);
  float* pb = A.alloc<float>((size_t)360 * FB);
  if (dry) return;
  for (long long f0 = 0; f0 < n; f0 += FB) {
    const int B = (int)std::min<long long>(FB, n - f0);
    hipLaunchKernelGGL(crepe_im2col_kernel, dim3(grid_for(512LL * B * 256)), dim3(256), 0, s, audio, L, hop, off, mean, rstd, (int)f0, B, col);
    ConvEpilogue Er; Er.act = ACT_RELU;
    const long long N1 = (long long)B * 256;
    conv1d_run(M->conv[0], s, col, N1, (int)N1, bufA, N1, Er);                       // [C1][B * 256], ReLU'd
    if (taps && taps->conv1 && f0 == 0)
      RVC_HIP_CHECK(hipMemcpy2DAsync(taps->conv1, 256 * sizeof(float), bufA, (size_t)N1 * sizeof(float), 256 * sizeof(float), (size_t)M->ch[0], hipMemcpyDeviceToDevice, s));
    float* cur = bufA; float* nxt = bufB;
    long long ldin = N1; int Sin = 256, Lin = 256;
    for (int i = 1; i < 6; ++i) {
      const int Lo = Lin >> 1, S = Lo + 63;
      const long long Tin = (long long)B * S, Tout = Tin - 63;
      hipLaunchKernelGGL(crepe_pool_kernel, dim3(grid_for((long long)M->ch[i - 1] * Tin)), dim3(256), 0, s, cur, ldin, Sin, Lin, M->bn_a[i - 1].p, M->bn_b[i - 1].p,
                         nxt, M->ch[i - 1], B, 0);                                    // [C_{i-1}][B * S] slots
      if (i >= 4) {
        // L = 16 / 8: GEMM over the explicit im2col; output dense [C_i][B * L] (frame b position p at b * L + p)
        const long long N = (long long)B * Lo;
        hipLaunchKernelGGL(crepe_slot_im2col_kernel, dim3(grid_for((long long)M->ch[i - 1] * 64 * N)), dim3(256), 0, s, nxt, Tin, S, Lo, M->ch[i - 1], B, col);
        conv1d_run(M->gemm[i], s, col, N, (int)N, cur, N, Er);
        ldin = N; Sin = Lo; Lin = Lo;
      } else {
        conv1d_run(M->conv[i], s, nxt, Tin, (int)Tin, cur, Tin, Er);                 // [C_i][B * S - 63], frame b position p at b * S + p
        ldin = Tin; Sin = S; Lin = Lo;
      }
      (void)Tout;
    }
    hipLaunchKernelGGL(crepe_pool_kernel, dim3(grid_for((long long)M->ch[5] * B * 4)), dim3(256), 0, s, cur, ldin, Sin, Lin, M->bn_a[5].p, M->bn_b[5].p, flat,
                       M->ch[5], B, 1);                                               // [4 * C6][B], row = pos * C6 + c
    if (taps && taps->embed && f0 == 0) RVC_HIP_CHECK(hipMemcpyAsync(taps->embed, flat, (size_t)4 * M->ch[5] * B * sizeof(float), hipMemcpyDeviceToDevice, s));
    ConvEpilogue Es; Es.act = ACT_SIGMOID;
    conv1d_run(M->fc, s, flat, B, B, pb, B, Es);                                      // [360][B]
    RVC_HIP_CHECK(hipMemcpy2DAsync(probs + f0, (size_t)n * sizeof(float), pb, (size_t)B * sizeof(float), (size_t)B * sizeof(float), 360, hipMemcpyDeviceToDevice, s));
  }
}

void crepe_forward(Crepe* M, hipStream_t s, const float* audio, long long L, int hop, int pad, float* probs, const CrepeTaps* taps) {
  RVC_REQUIRE(M->ready, "crepe_finalize has not been called");
  const long long n = crepe_num_frames(L, hop, pad);
  RVC_REQUIRE(n > 0 && n < (1LL << 24), "no frames (audio shorter than one 1024-sample window without padding?)");
  Arena& A = M->arena;
  for (int pass = 0; pass < 2; ++pass) {
    A.dry = (pass == 0); A.reset(); if (pass == 0) A.peak = 0;
    crepe_graph(M, s, A, audio, L, hop, pad, probs, n, taps);
    if (pass == 0) A.ensure(A.peak);
  }
  A.dry = false;
}

size_t crepe_workspace(const Crepe* M) { return M->arena.cap; }

}  // namespace rvc
