// Feature retrieval of VC.vc (reference vc_infer_pipeline.py:60-75): for every HuBERT frame the nearest training feature
// (index.search(npy, k=1) on a faiss index over big_npy [N][D]) is blended in with weight index_rate.  Device resident:
//   argmin_j |f_i - b_j|^2 = argmax_j (b_j . f_i - |b_j|^2 / 2)
// as one MFMA GEMM per chunk of rows (the -|b|^2/2 term is the GEMM bias) followed by a column arg-max that carries the running best
// across chunks.  Ties go to the smallest index (numpy argmin order).
// Two searches share that machinery.  EXACT (index_create: .npy / tuple inputs, which carry no cell structure): every row competes.
// IVF (index_create_ivf: the reference's own index type - faiss IndexIVFFlat built by custom_nodes/rvc_nodes.py:500-554 as "IVF{n},Flat",
// nprobe 1): faiss's published algorithm is (1) the coarse quantiser, an IndexFlatL2 over the nlist centroids, returns the nprobe nearest
// centroids of the query, (2) only the inverted lists of those cells are scanned and the smallest squared L2 distance among THEIR vectors
// wins; probed lists that are empty give label -1 / distance FLT_MAX.  Here: the same GEMM + arg-max over the centroids (top-nprobe by
// repeated selection), then the arg-max over the rows with a mask "row's list is one of the query's cells" - the scores of all rows are
// computed as for the exact search (retrieval parity, not the IVF's saving, is the point; 100 k rows x 1599 frames: 4.8 ms).
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
  // IVF probe (nlist > 0): centroids as the k-major GEMM operand [D][nlist], -|c|^2/2, the list every row was added to
  int nlist = 0, nprobe = 0;
  float* cen_cols = nullptr; float* cen_nhalf = nullptr; int* list_of = nullptr;
};

__global__ void index_prep_kernel(const float* __restrict__ rows, float* __restrict__ cols, float* __restrict__ nhalf, long long N, int D) {
  const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= N) return;
  float s = 0.f;
  for (int c = 0; c < D; ++c) { const float v = rows[j * D + c]; cols[(long long)c * N + j] = v; s = fmaf(v, v, s); }
  nhalf[j] = -0.5f * s;
}

// Y [M][T] (row pitch ldY): per column the arg-max over rows, merged into (best, bidx); rows are global indices m0 + m.
// Block = 64 columns x 16 row slices.  list_of != nullptr: only rows whose list is one of the column's P cells (cells[p][t]) compete; a
// column without any eligible row keeps index -1 (faiss's "no result" label).
constexpr int kMaxProbe = 16;
__global__ __launch_bounds__(1024) void index_argmax_kernel(const float* __restrict__ Y, int M, int T, long long ldY, long long m0,
                                                            float* __restrict__ best, long long* __restrict__ bidx, int first,
                                                            const int* __restrict__ list_of, const int* __restrict__ cells, int P) {
  __shared__ float s_v[16][64];
  __shared__ long long s_i[16][64];
  const int col = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int t = blockIdx.x * 64 + col;
  float bv = -3.0e38f; long long bi = -1;          // (a column of NaN scores compares false everywhere: it keeps "no row")
  int cell[kMaxProbe];
#pragma unroll
  for (int p = 0; p < kMaxProbe; ++p) cell[p] = (list_of && p < P && t < T) ? cells[(long long)p * T + t] : -1;
  if (t < T)
    for (int m = sl; m < M; m += 16) {
      if (list_of) {
        const int li = list_of[m0 + m];
        bool ok = false;
#pragma unroll
        for (int p = 0; p < kMaxProbe; ++p) ok = ok || li == cell[p];
        if (!ok) continue;
      }
      const float v = Y[(long long)m * ldY + t];
      if (v > bv || bi < 0) { bv = v; bi = m0 + m; }   // strictly greater: the first (smallest) row wins inside a slice
    }
  s_v[sl][col] = bv; s_i[sl][col] = bi;
  __syncthreads();
  if (sl == 0 && t < T) {
    for (int i = 1; i < 16; ++i) {
      const float v = s_v[i][col]; const long long ix = s_i[i][col];
      if (ix >= 0 && (bi < 0 || v > bv || (v == bv && ix < bi))) { bv = v; bi = ix; }
    }
    if (!first) { const float pv = best[t]; const long long pi = bidx[t]; if (pi >= 0 && (bi < 0 || pv > bv || (pv == bv && pi < bi))) { bv = pv; bi = pi; } }
    best[t] = bv; bidx[t] = bi;
  }
}

// coarse quantiser: the P best rows of Yc [nlist][T] per column in decreasing score (ties: smaller row first), cells[p][t].  One thread per
// column; pass p picks the best row that comes after pass p - 1's pick in that order.
__global__ void index_topk_kernel(const float* __restrict__ Yc, int nlist, int T, int P, int* __restrict__ cells) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  float pv = 3.0e38f; int pi = -1;
  for (int p = 0; p < P; ++p) {
    float bv = -3.0e38f; int bi = -1;
    for (int m = 0; m < nlist; ++m) {
      const float v = Yc[(long long)m * T + t];
      const bool after = v < pv || (v == pv && m > pi);        // not picked yet
      if (after && (bi < 0 || v > bv)) { bv = v; bi = m; }
    }
    cells[(long long)p * T + t] = bi;
    pv = bv; pi = bi;
    if (bi < 0) { for (int q = p + 1; q < P; ++q) cells[(long long)q * T + t] = -1; break; }
  }
}

// score[t] = |f_t|^2 - 2 best[t]  (squared L2 distance, what faiss returns for an L2 index)
__global__ void index_score_kernel(const float* __restrict__ f, const float* __restrict__ best, const long long* __restrict__ bidx,
                                   float* __restrict__ score, int D, int T) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  if (bidx[t] < 0) { score[t] = 3.4028234663852886e38f; return; }      // no vector in the probed lists: faiss reports FLT_MAX with label -1
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
    long long r = idx[t];
    // label -1 (IVF probe of empty lists): the reference computes weight = (1 / FLT_MAX)^2 = 0, weight / sum(weight) = 0 / 0 - the frame
    // becomes NaN (vc_infer_pipeline.py:66-74); the same here rather than a silent substitute.  Other indices are clamped into the table.
    if (r < 0) { out[i] = __int_as_float(0x7fc00000); continue; }
    r = r >= N ? N - 1 : r;
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
    static const bool as_conv = (exp_int("RVC_INDEX_X3", 1) != 0);
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
FeatIndex* index_create_ivf(Ctx* ctx, const float* big_npy, long long N, int D, const float* centroids, int nlist, const int* list_of, int nprobe) {
  RVC_REQUIRE(centroids && list_of && nlist > 0 && nprobe >= 1, "IVF index: centroids, list assignment, nlist >= 1 and nprobe >= 1 expected");
  RVC_REQUIRE(nprobe <= kMaxProbe || nprobe >= nlist, "IVF index: nprobe up to 16 (or >= nlist, which is the exact search)");
  for (long long j = 0; j < N; ++j) RVC_REQUIRE(list_of[j] >= 0 && list_of[j] < nlist, "IVF index: list assignment out of range");
  FeatIndex* I = index_create(ctx, big_npy, N, D);
  if (nprobe >= nlist) return I;                                   // every cell probed: the exact search
  try {
    std::vector<float> cc((size_t)D * nlist), nh((size_t)nlist);
    for (int m = 0; m < nlist; ++m) {
      float s = 0.f;
      for (int c = 0; c < D; ++c) { const float v = centroids[(size_t)m * D + c]; cc[(size_t)c * nlist + m] = v; s = fmaf(v, v, s); }
      nh[m] = -0.5f * s;
    }
    RVC_HIP_CHECK(hipMalloc(&I->cen_cols, cc.size() * sizeof(float)));
    RVC_HIP_CHECK(hipMalloc(&I->cen_nhalf, nh.size() * sizeof(float)));
    RVC_HIP_CHECK(hipMalloc(&I->list_of, (size_t)N * sizeof(int)));
    RVC_HIP_CHECK(hipMemcpy(I->cen_cols, cc.data(), cc.size() * sizeof(float), hipMemcpyHostToDevice));
    RVC_HIP_CHECK(hipMemcpy(I->cen_nhalf, nh.data(), nh.size() * sizeof(float), hipMemcpyHostToDevice));
    RVC_HIP_CHECK(hipMemcpy(I->list_of, list_of, (size_t)N * sizeof(int), hipMemcpyHostToDevice));
    I->nlist = nlist; I->nprobe = nprobe;
  } catch (...) { index_destroy(I); throw; }
  return I;
}
void index_destroy(FeatIndex* I) {
  if (I) {
    for (auto& L : I->chunks) conv_layer_free(L);
    dev_free(I->rows); dev_free(I->cols); dev_free(I->nhalf); dev_free(I->cen_cols); dev_free(I->cen_nhalf);
    if (I->list_of) (void)hipFree(I->list_of);
    delete I;
  }
}
int index_nprobe(const FeatIndex* I) { return I->nlist > 0 ? I->nprobe : 0; }
long long index_size(const FeatIndex* I) { return I->N; }
int index_dim(const FeatIndex* I) { return I->D; }

void index_search(FeatIndex* I, hipStream_t s, const float* feats_cm, int T, long long* idx, float* score) {
  RVC_REQUIRE(T > 0, "empty query");
  const long long chunk = 32768;
  const long long mc = I->N < chunk ? I->N : chunk;
  const bool ivf = I->nlist > 0;
  // scratch: scores of one chunk [mc][T] (the centroid scores [nlist][T] first) + running best [T] + the probed cells [nprobe][T]
  const size_t ymax = (size_t)(mc > I->nlist ? mc : I->nlist) * T;
  float* Y = (float*)stream_scratch(s, 3, (ymax + T + (size_t)(ivf ? I->nprobe : 0) * T) * sizeof(float));
  float* best = Y + ymax;
  int* cells = reinterpret_cast<int*>(best + T);
  ConvEpilogue E0;
  if (ivf) {
    // coarse quantiser (IndexFlatL2 over the centroids), fp32
    gemm_tn_run(s, I->cen_cols, I->nlist, 0, feats_cm, T, 0, Y, T, 0, I->nlist, T, I->D, 1, I->cen_nhalf, 0, E0);
    hipLaunchKernelGGL(index_topk_kernel, dim3((T + 127) / 128), dim3(128), 0, s, Y, I->nlist, T, I->nprobe, cells);
  }
  for (long long m0 = 0; m0 < I->N; m0 += chunk) {
    const int M = (int)((I->N - m0) < chunk ? (I->N - m0) : chunk);
    if (!I->chunks.empty()) conv1d_run(I->chunks[(size_t)(m0 / chunk)], s, feats_cm, T, T, Y, T, E0);
    else gemm_tn_run(s, I->cols + m0, I->N, 0, feats_cm, T, 0, Y, T, 0, M, T, I->D, 1, I->nhalf + m0, 0, E0);
    hipLaunchKernelGGL(index_argmax_kernel, dim3((T + 63) / 64), dim3(1024), 0, s, Y, M, T, (long long)T, m0, best, idx, m0 == 0 ? 1 : 0,
                       ivf ? I->list_of : nullptr, cells, ivf ? I->nprobe : 0);
  }
  if (score) hipLaunchKernelGGL(index_score_kernel, dim3((T + 255) / 256), dim3(256), 0, s, feats_cm, best, idx, score, I->D, T);
}

void index_blend(FeatIndex* I, hipStream_t s, const float* feats_cm, const long long* idx, int T, float rate, float* out_cm) {
  const long long n = (long long)I->D * T; int blocks = (int)((n + 255) / 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(index_blend_kernel, dim3(blocks), dim3(256), 0, s, feats_cm, I->rows, idx, rate, out_cm, I->D, T, I->N);
}

}  // namespace rvc
