// Feature retrieval of VC.vc (reference vc_infer_pipeline.py:60-75): for every HuBERT frame the nearest training feature
// (index.search(npy, k=1) on a faiss index over big_npy [N][D]) is blended in with weight index_rate.  The reference's index is
// IVF-Flat with nprobe 1, i.e. an approximation of the exact nearest neighbour; this is the exact search, device resident:
//   argmin_j |f_i - b_j|^2 = argmax_j (b_j . f_i - |b_j|^2 / 2)
// as one fp32 MFMA GEMM per chunk of rows (conv_mfma.hip, activation-as-weight form, the -|b|^2/2 term is the GEMM bias) followed
// by a column arg-max that carries the running best across chunks.  Ties go to the smallest index (numpy argmin order).
#include "model_common.h"
#include "models.h"

namespace rvc {

struct FeatIndex {
  Ctx* ctx = nullptr;
  long long N = 0; int D = 0;
  float* rows = nullptr;      // big_npy [N][D] (gather source for the blend)
  float* cols = nullptr;      // big_npy^T [D][N] (k-major GEMM operand)
  float* nhalf = nullptr;     // -|b_j|^2 / 2
  // per 32768-row chunk: the rows as a k = 1 convolution layer (score = W f + bias) - on the bf16x3 kernel for large indices
  // (3-term split, fp32 accumulation: scores to ~1e-5 relative, i.e. ties closer than that may resolve to the other neighbour;
  // the reference's own IVF search with nprobe = 1 is far coarser), on the fp32 kernel for small ones
  std::vector<ConvLayer> chunks;
};

__global__ void index_prep_kernel(const float* __restrict__ rows, float* __restrict__ cols, float* __restrict__ nhalf, long long N, int D) {
  const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= N) return;
  float s = 0.f;
  for (int c = 0; c < D; ++c) { const float v = rows[j * D + c]; cols[(long long)c * N + j] = v; s = fmaf(v, v, s); }
  nhalf[j] = -0.5f * s;
}

// Y [M][T] (row pitch ldY): per column the arg-max over rows, merged into (best, bidx); rows are global indices m0 + m.
// Block = 64 columns x 16 row slices.
__global__ __launch_bounds__(1024) void index_argmax_kernel(const float* __restrict__ Y, int M, int T, long long ldY, long long m0,
                                                            float* __restrict__ best, long long* __restrict__ bidx, int first) {
  __shared__ float s_v[16][64];
  __shared__ long long s_i[16][64];
  const int col = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int t = blockIdx.x * 64 + col;
  float bv = -3.0e38f; long long bi = m0;          // (a column of NaN scores compares false everywhere: it keeps a valid row)
  if (t < T)
    for (int m = sl; m < M; m += 16) {
      const float v = Y[(long long)m * ldY + t];
      if (v > bv) { bv = v; bi = m0 + m; }          // strictly greater: the first (smallest) row wins inside a slice
    }
  s_v[sl][col] = bv; s_i[sl][col] = bi;
  __syncthreads();
  if (sl == 0 && t < T) {
    for (int i = 1; i < 16; ++i) {
      const float v = s_v[i][col]; const long long ix = s_i[i][col];
      if (v > bv || (v == bv && ix < bi)) { bv = v; bi = ix; }
    }
    if (!first) { const float pv = best[t]; const long long pi = bidx[t]; if (pv > bv || (pv == bv && pi < bi)) { bv = pv; bi = pi; } }
    best[t] = bv; bidx[t] = bi;
  }
}

// score[t] = |f_t|^2 - 2 best[t]  (squared L2 distance, what faiss returns for an L2 index)
__global__ void index_score_kernel(const float* __restrict__ f, const float* __restrict__ best, float* __restrict__ score, int D, int T) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  float s = 0.f;
  for (int c = 0; c < D; ++c) { const float v = f[(long long)c * T + t]; s = fmaf(v, v, s); }
  // cancellation can leave an exact member of the index at 0 or slightly below; the generic VC.vc path weighs by 1 / score^2
  // (reference vc_infer_pipeline.py:66-68), so the distance is kept strictly positive
  score[t] = fmaxf(s - 2.f * best[t], 1e-10f);
}

// out[c][t] = rate * rows[idx[t]][c] + (1 - rate) * f[c][t]      (reference :71-74; k = 1 makes the 1/score^2 weight exactly 1)
__global__ void index_blend_kernel(const float* __restrict__ f, const float* __restrict__ rows, const long long* __restrict__ idx, float rate,
                                   float* __restrict__ out, int D, int T, long long N) {
  const long long n = (long long)D * T;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long st = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += st) {
    const int c = (int)(i / T); const int t = (int)(i - (long long)c * T);
    long long r = idx[t]; r = r < 0 ? 0 : (r >= N ? N - 1 : r);        // caller-supplied indices are clamped into the table
    out[i] = rows[r * D + c] * rate + (1.f - rate) * f[i];
  }
}

FeatIndex* index_create(Ctx* ctx, const float* big_npy, long long N, int D) {
  RVC_REQUIRE(big_npy && N > 0 && D > 0 && D <= 4096, "bad index shape");
  FeatIndex* I = new FeatIndex(); I->ctx = ctx; I->N = N; I->D = D;
  try {
    RVC_HIP_CHECK(hipMalloc(&I->rows, (size_t)N * D * sizeof(float)));
    RVC_HIP_CHECK(hipMalloc(&I->cols, (size_t)N * D * sizeof(float)));
    RVC_HIP_CHECK(hipMalloc(&I->nhalf, (size_t)N * sizeof(float)));
    RVC_HIP_CHECK(hipMemcpy(I->rows, big_npy, (size_t)N * D * sizeof(float), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(index_prep_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, 0, I->rows, I->cols, I->nhalf, N, D);
    RVC_HIP_CHECK(hipDeviceSynchronize());
    static const bool as_conv = !(getenv("RVC_INDEX_X3") && atoi(getenv("RVC_INDEX_X3")) == 0);
    if (as_conv && D % 16 == 0 && N >= 4096) {
      std::vector<float> nh((size_t)N);
      RVC_HIP_CHECK(hipMemcpy(nh.data(), I->nhalf, (size_t)N * sizeof(float), hipMemcpyDeviceToHost));   // the same bias values as the fp32 path
      ConvBuildScope x3scope(ctx->precision);
      const long long chunk = 32768;
      I->chunks.resize((size_t)((N + chunk - 1) / chunk));
      for (size_t c = 0; c < I->chunks.size(); ++c) {
        const long long m0 = (long long)c * chunk;
        const int M = (int)((N - m0) < chunk ? (N - m0) : chunk);
        conv1d_layer_init(I->chunks[c], big_npy + m0 * D, nh.data() + m0, M, D, 1, 1, 0, 1, 1);
      }
      dev_free(I->cols); I->cols = nullptr;             // the k-major copy is only needed by the GEMM path
    }
  } catch (...) { for (auto& L : I->chunks) conv_layer_free(L); dev_free(I->rows); dev_free(I->cols); dev_free(I->nhalf); delete I; throw; }
  return I;
}
void index_destroy(FeatIndex* I) {
  if (I) { for (auto& L : I->chunks) conv_layer_free(L); dev_free(I->rows); dev_free(I->cols); dev_free(I->nhalf); delete I; }
}
long long index_size(const FeatIndex* I) { return I->N; }
int index_dim(const FeatIndex* I) { return I->D; }

void index_search(FeatIndex* I, hipStream_t s, const float* feats_cm, int T, long long* idx, float* score) {
  RVC_REQUIRE(T > 0, "empty query");
  const long long chunk = 32768;
  const long long mc = I->N < chunk ? I->N : chunk;
  // scratch: scores of one chunk [mc][T] + running best [T]
  float* Y = (float*)stream_scratch(s, 3, ((size_t)mc * T + T) * sizeof(float));
  float* best = Y + (size_t)mc * T;
  ConvEpilogue E0;
  for (long long m0 = 0; m0 < I->N; m0 += chunk) {
    const int M = (int)((I->N - m0) < chunk ? (I->N - m0) : chunk);
    if (!I->chunks.empty()) conv1d_run(I->chunks[(size_t)(m0 / chunk)], s, feats_cm, T, T, Y, T, E0);
    else gemm_tn_run(s, I->cols + m0, I->N, 0, feats_cm, T, 0, Y, T, 0, M, T, I->D, 1, I->nhalf + m0, 0, E0);
    hipLaunchKernelGGL(index_argmax_kernel, dim3((T + 63) / 64), dim3(1024), 0, s, Y, M, T, (long long)T, m0, best, idx, m0 == 0 ? 1 : 0);
  }
  if (score) hipLaunchKernelGGL(index_score_kernel, dim3((T + 255) / 256), dim3(256), 0, s, feats_cm, best, score, I->D, T);
}

void index_blend(FeatIndex* I, hipStream_t s, const float* feats_cm, const long long* idx, int T, float rate, float* out_cm) {
  const long long n = (long long)I->D * T; int blocks = (int)((n + 255) / 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(index_blend_kernel, dim3(blocks), dim3(256), 0, s, feats_cm, I->rows, idx, rate, out_cm, I->D, T, I->N);
}

}  // namespace rvc
