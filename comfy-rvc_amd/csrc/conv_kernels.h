// Device-side helpers and launch plumbing shared by the convolution kernels (conv_mfma.hip: fp32 MFMA,
// conv_x3.hip: bf16x3 split MFMA).  Private to csrc/.
#pragma once
#include "rvc_internal.h"
#include <hip/hip_ext.h>

namespace rvc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct ConvArgsX : ConvArgs {
  long long ldW;   // pitch (floats) of one packed-weight row
  int Wcols;       // valid columns in a weight row
  int Wrows;       // valid rows of the weight matrix
  int ksplit; float* partial; long long ldP;   // split-K: chunk ranges over blockIdx.z, raw accumulators to partial[ks][m][ldP]
  unsigned magRP, magPW;   // 2-D: ceil(2^32 / d) for d = (BH+2)*PW and d = PW (exact division of small tile indices)
  int ni; unsigned magNI;  // 1-D: 64-wide column groups per staged row (ceil(span / 64)) and its division magic
  // bf16x3 kernel (conv_x3.hip): split weight image, its padded row count, X buffers in LDS (1 or 2)
  const unsigned char* Wx; int CoPx; int xbufs; int NC; int kreal; long long wxBatch;
  // fused ResBlock pair (conv_x3_kernel<..., FUSE>): second conv's weight image, first conv's bias, halo (k - 1) / 2 of the
  // second conv, slope of the leaky ReLU between the two
  const unsigned char* Wx2; const float* bias1; int fuse_p2; float fuse_slope;
  const unsigned char* Xs; long long xsTp;      // split-resident input image (replaces X; see ConvEpilogue) and its rows per plane
  unsigned char* Ys; long long ysTp; float ys_slope;   // split-resident output image (replaces Y), activation slope applied before the split
  int wbufs;       // weight slabs in the LDS ring (2 .. 4)
  int xcd_remap;   // 1: tiles renumbered so that each XCD (L2) works on a contiguous run of them
  int h2;          // conv_x3q_kernel only: fp16x2 arithmetic (Wx = the layer's one-plane fp16 image, activation images fp16 hi / lo)
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
// out-of-range offsets (>= the descriptor's extent; kOOB always is) read as 0 and drop stores: the hardware range check
// supplies zero padding, channel tails and ragged edges without branches
constexpr unsigned kOOB = 0x80000000u;
__device__ __forceinline__ float buf_load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff = 0) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}

// Dense-row epilogue shared by the MFMA kernels (32x32 accumulator layout: col = lane & 31,
// row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)):  v = acc + bias; act in {identity, ReLU, leaky ReLU} as max(v, slope v),
// before or after the residual; * out_scale; (+ previous output).  Row r+1's residual / accumulate operands are requested
// before row r is stored (vmcnt counts loads and stores alike).  32-bit element offsets through buffer descriptors.
// HAS_R / HAS_ACC: whether the residual / the previous output are read at all.  (Reading them through a zero-extent descriptor costs
// nothing in bandwidth but every dummy load is a VMEM instruction: half of the epilogue's memory instructions on a residual-only launch.)
template <int WM, int WN, int AM, int AN, int G, bool HAS_R, bool HAS_ACC>
__device__ __forceinline__ void dense_epilogue_t(const ConvArgsX& p, f32x16 (&acc)[AM][AN], int z, int co0, int n0, int wm, int wn, int li, int lh) {
  const float* __restrict__ bias = p.bias ? p.bias + (long long)z * p.bBatch : nullptr;
  const float* R = HAS_R ? p.R + (long long)z * p.rBatch : nullptr;
  float* Y = p.Y + (long long)z * p.yBatch;
  const float lslope = p.act == ACT_NONE ? 1.f : (p.act == ACT_RELU ? 0.f : p.act_slope);
  const float oscale = p.out_scale;
  const bool plain = p.orows == p.Co;
  const __amdgpu_buffer_rsrc_t yrs = make_rsrc(Y, (unsigned)p.orows * (unsigned)p.ldY * 4u);
  const __amdgpu_buffer_rsrc_t rrs = make_rsrc(HAS_R ? R : Y, HAS_R ? (unsigned)p.orows * (unsigned)p.ldR * 4u : 0u);
  const bool abr = p.act_before_res != 0;
  constexpr bool need_loads = HAS_R || HAS_ACC;
  auto row_info = [&](int am, int r, bool& mok, float& bv, unsigned& yrow, unsigned& rrow) {
    const int m = co0 + (wm * AM + am) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    mok = m < p.Co;
    const int co = plain ? m : m % p.orows;
    bv = (bias && mok) ? bias[co] : 0.f;
    yrow = (unsigned)co * (unsigned)p.ldY; rrow = (unsigned)co * (unsigned)p.ldR;
  };
  // rows are processed in groups of G: the residual / accumulate operands of group g + 1 are requested before group g is
  // stored, so G * AN loads per operand and wave are in flight (the epilogue of the HBM-bound launches is latency-bound)
  static_assert(16 % G == 0, "group size");
#pragma unroll
  for (int am = 0; am < AM; ++am) {
    float rv[G][AN], yv[G][AN], rvn[G][AN], yvn[G][AN];
#pragma unroll
    for (int j = 0; j < G; ++j)
#pragma unroll
      for (int an = 0; an < AN; ++an) { rv[j][an] = 0.f; yv[j][an] = 0.f; rvn[j][an] = 0.f; yvn[j][an] = 0.f; }
    auto load_group = [&](int r0, float (&rr)[G][AN], float (&yy)[G][AN]) {
#pragma unroll
      for (int j = 0; j < G; ++j) {
        bool mok; float bv; unsigned yrow, rrow;
        row_info(am, r0 + j, mok, bv, yrow, rrow);
#pragma unroll
        for (int an = 0; an < AN; ++an) {
          const int n = n0 + (wn * AN + an) * 32 + li;
          const bool ok = mok && n < p.Tout;
          if constexpr (HAS_R) rr[j][an] = buf_load(rrs, ok ? (rrow + (unsigned)n) * 4u : kOOB);
          if constexpr (HAS_ACC) yy[j][an] = buf_load(yrs, ok ? (yrow + (unsigned)n) * 4u : kOOB);
        }
      }
    };
    if constexpr (need_loads) load_group(0, rv, yv);
#pragma unroll
    for (int r0 = 0; r0 < 16; r0 += G) {
      if constexpr (need_loads) { if (r0 + G < 16) load_group(r0 + G, rvn, yvn); }
#pragma unroll
      for (int j = 0; j < G; ++j) {
        bool mok; float bv; unsigned yrow, rrow;
        row_info(am, r0 + j, mok, bv, yrow, rrow);
#pragma unroll
        for (int an = 0; an < AN; ++an) {
          const int n = n0 + (wn * AN + an) * 32 + li;
          const bool ok = mok && n < p.Tout;
          float v = acc[am][an][r0 + j] + bv;
          if (abr) v = fmaxf(v, v * lslope) + rv[j][an];
          else { v += rv[j][an]; v = fmaxf(v, v * lslope); }
          v = v * oscale + yv[j][an];
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), yrs, (int)(ok ? (yrow + (unsigned)n) * 4u : kOOB), 0, 0);
        }
      }
#pragma unroll
      for (int j = 0; j < G; ++j)
#pragma unroll
        for (int an = 0; an < AN; ++an) { rv[j][an] = rvn[j][an]; yv[j][an] = yvn[j][an]; }
    }
  }
}
// G = rows per load group when operands are read (bounded by registers: 2 x G x AN values per operand), GP for store-only launches
template <int WM, int WN, int AM, int AN, int G = 1>
__device__ __forceinline__ void dense_epilogue(const ConvArgsX& p, f32x16 (&acc)[AM][AN], int z, int co0, int n0, int wm, int wn, int li, int lh) {
  const bool has_r = p.R != nullptr, has_acc = p.accumulate != 0;
  if (has_r && has_acc) dense_epilogue_t<WM, WN, AM, AN, G, true, true>(p, acc, z, co0, n0, wm, wn, li, lh);
  else if (has_r) dense_epilogue_t<WM, WN, AM, AN, (AM * AN >= 8 ? 2 * G : G) <= 16 ? (AM * AN >= 8 ? 2 * G : G) : 16, true, false>(p, acc, z, co0, n0, wm, wn, li, lh);
  else if (has_acc) dense_epilogue_t<WM, WN, AM, AN, G, false, true>(p, acc, z, co0, n0, wm, wn, li, lh);
  else dense_epilogue_t<WM, WN, AM, AN, G, false, false>(p, acc, z, co0, n0, wm, wn, li, lh);
}

// ---- host side
struct TileCfg { int WM, WN, AM, AN; };
TileCfg choose_tile(int M, long long N, int batch);
int tile_cfg_id(const TileCfg& t);            // 0..6, -1 if not an instantiated tiling

// per-launch HIP-event profiling (cfg: 0..6 fp32 1-D, 7..13 fp32 2-D, 14..20 bf16x3 1-D)
constexpr int kProfCfgs = 24;
struct ProfTicket { hipEvent_t a = nullptr, b = nullptr; bool on = false; };
ProfTicket conv_prof_begin(hipStream_t s);
// While a profiling bracket is open on this thread, the FIRST kernel launched through conv_launch carries its own start / stop events (hipExtLaunchKernelGGL:
// the dispatch packet's begin / end timestamps - what rocprofv3's kernel trace reports).  A bracket that held exactly one such launch is timed by them; a
// bracket with several launches (split-K + reduction) or with launches that do not go through conv_launch keeps the event pair recorded around it, which
// also counts the command processor's ~2.4 us between the markers and the dispatch.
struct ProfKernelEvents { hipEvent_t ka = nullptr, kb = nullptr; bool armed = false; int launches = 0; };
ProfKernelEvents& prof_kernel_events();
template <class K, class A>
inline void conv_launch(K kern, dim3 grid, dim3 block, size_t lds, hipStream_t s, const A& a) {
  ProfKernelEvents& pe = prof_kernel_events();
  if (pe.armed && pe.launches++ == 0) hipExtLaunchKernelGGL(kern, grid, block, lds, s, pe.ka, pe.kb, 0, a);
  else hipLaunchKernelGGL(kern, grid, block, lds, s, a);
}
void conv_prof_end(ProfTicket& t, hipStream_t s, double flops, int cfg, double bytes = 0.0, const ConvArgsX* shape = nullptr, long long blocks = 0,
                   int fused = 0);
int conv_prof_dump_csv(const char* path);   // one row per recorded launch: shape, tile, grid, us, algorithmic FLOPs / bytes
double conv_alg_bytes(const ConvArgsX& a, int batch);
int conv_prof_collect_ex(double* out /* [kProfCfgs][8] */, double ridge_fp32, double ridge_x3);

// second pass of a split-K launch (conv_mfma.hip): fixed-order sum of the partials + dense epilogue
void splitk_reduce_launch(const ConvArgsX& a, int S, int batch, hipStream_t s);

// bf16x3 path: returns false when the layer / geometry is not eligible (caller falls back to the fp32 kernel)
bool conv_x3_try(ConvArgsX& a, int batch, hipStream_t s, double flops, bool dry = false);
bool conv_x3_enabled();
// software-pipelined kernel for stride-1 1-D convolutions on 2 x 2-wave tiles (conv_x3p.hip); `a` as conv_x3_try prepared it
bool conv_x3p_try(ConvArgsX& a, int AM, int AN, hipStream_t s, dim3& grid_out, bool dry);
int conv_x3p_check_read();
// persistent version of the above for the ResBlock convolutions: a workgroup per CU slot walks over its tiles, one continuous stream of
// weight units / input chunks, residual added block by block inside the tile (conv_x3q.hip)
bool conv_x3q_try(ConvArgsX& a, int AM, int AN, hipStream_t s, dim3& grid_out, bool dry);
int conv_x3q_check_read();
// k = 1 (GEMM) on the pipelined kernel, fp32 [K][N] input (conv_x3p.hip)
bool conv_x3g_try(ConvArgsX& a, hipStream_t s, dim3& grid_out, int& ksplit_out, bool dry);
// fused ResBlock pair of the 32-channel stage on the pipelined kernel (conv_x3p.hip)
bool conv_x3pf_try(ConvArgsX& a, int T, hipStream_t s, dim3& grid_out, bool dry);
// the same pair in the fp16x2 arithmetic with both weight sets resident in LDS, persistent workgroups (conv_rbh.hip); Wx / Wx2 = the one-plane fp16 images
bool conv_rbh_try(ConvArgsX& a, int T, hipStream_t s, dim3& grid_out, bool dry);
// a whole ResBlock1 (three (dilated, plain) pairs) of the 32-channel stage in one launch, fp16x2 arithmetic, bit-identical to the chain of three conv_rbh
// launches (conv_rb3.hip); y = x3 * out_scale [+ y]; false: not eligible (the caller runs the pairs one by one)
bool conv_rb3_try(const ConvLayer* const* c1, const ConvLayer* const* c2, hipStream_t s, const float* X, long long ldX, int T, float* Y, long long ldY,
                  float pre_slope, float out_scale, int accumulate, bool dry = false, const float* nsrc = nullptr, const float* nw = nullptr,
                  const float* nb = nullptr);      // nsrc [T], nw [32], nb [32]: x[c][t] + fmaf(nw[c], nsrc[t], nb[c]) is what the ResBlock reads (the last stage's noise branch)
// y = (x + c2(lrelu(c1(lrelu(x))))) * scale [+ y] for a ResBlock1 pair of narrow layers in ONE launch; false when not eligible
bool conv_x3_pair_try(const ConvLayer& c1, const ConvLayer& c2, hipStream_t s, const float* X, long long ldX, int T, float* Y, long long ldY,
                      const ConvEpilogue& e2, bool dry_only = false);

}  // namespace rvc
