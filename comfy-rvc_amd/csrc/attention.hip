// Fused multi-head self-attention for HuBERT (transformers modeling_hubert.py:245-330, eager path; the reference runs
// `sdpa`): out[h*64+d][q] = sum_k softmax_k(K_h[:,k] . Q_h[:,q]) * V[k][h*64+d] + bv, with Q already scaled.
// One workgroup (4 waves) owns 64 queries of one head and walks the keys in tiles of 64 with an online softmax, so the
// [heads][T][T] score matrix never exists in HBM (it was 123 MB written + 3 x 123 MB re-read per layer at 30 s).
//   S tile  (64 keys x 64 queries) = K^T Q   : v_mfma_f32_32x32x2_f32, A = Ks[d][key], B = Qs[d][query]
//   P tile  = exp(S - m_new) to LDS           : m, l are per query column = per lane of the 32x32 accumulator layout
//   O tile  (64 d x 64 queries) += V^T P      : A = Vs[key][d], B = Ps[key][query]; O is rescaled by exp(m_old - m_new)
// fp32 throughout (bitwise fp32 FMA chains in the MFMA units); keys beyond T are masked, query tails are not stored.
// Measured at 12 heads, T = 1599: 195 us (the three-kernel path it replaces took 312 us); ablations: MFMAs 96 us, K/V tile loads
// 25 us, exp 17 us, the rest scalar ds_read_b32 operand fetches at one wave per SIMD.  Next step: row-major Q / K and
// channel-major V so that every operand fragment is one ds_read_b128 for four MFMA steps.
#include "rvc_internal.h"

namespace rvc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifdef RVC_CONV_TIMING
__device__ unsigned long long g_att_timing[8];   // [0] blocks, [1] K/V store + barriers, [2] S MFMAs, [3] softmax + P, [4] PV MFMAs, [6] total
#define ATICK() wall_clock64()
#define ATACC(i, v) do { if (threadIdx.x == 0) atomicAdd(&g_att_timing[i], (unsigned long long)(v)); } while (0)
void attention_timing_read(unsigned long long* out8, bool reset) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_att_timing), sizeof(unsigned long long) * 8);
  if (reset) { unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_att_timing), z, sizeof(z)); }
}
#else
#define ATICK() 0ull
#define ATACC(i, v) do {} while (0)
#endif

// LDS-only workgroup barrier: __syncthreads() also fences global memory (s_waitcnt vmcnt(0)), which would drain the K / V prefetch
// of the next tile at every one of the four barriers of an iteration
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int kAD = 64;        // head dimension
constexpr int kAP = 65;        // LDS row pitch (floats): odd -> the 32-lane row / column reads are conflict-free

__global__ __launch_bounds__(256) void attention_kernel(const float* __restrict__ Q, const float* __restrict__ K, long long ldqk,
                                                        const float* __restrict__ V, long long ldv, const float* __restrict__ bv,
                                                        float* __restrict__ out, long long ldo, int T) {
  __shared__ float Qs[kAD * kAP], Ks[kAD * kAP], Vs[64 * kAP], Ps[64 * kAP];
  __shared__ float red[2][2][64];          // [stat: max | sum][wm][query]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int h = blockIdx.y, q0 = blockIdx.x * 64;
  const float* Qh = Q + (long long)h * kAD * ldqk;
  const float* Kh = K + (long long)h * kAD * ldqk;
  const float* Vh = V + h * kAD;

  // Q tile: rows d, columns queries (coalesced along the queries)
  for (int e = tid; e < kAD * 64; e += 256) {
    const int d = e >> 6, j = e & 63;
    Qs[d * kAP + j] = (q0 + j < T) ? Qh[(long long)d * ldqk + q0 + j] : 0.f;
  }
  float kr[16], vr[16];
  auto load_kv = [&](int k0) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int e = tid + 256 * s;
      const int r = e >> 6, j = e & 63;                    // K: r = d, j = key;  V: r = key, j = d
      kr[s] = (k0 + j < T) ? Kh[(long long)r * ldqk + k0 + j] : 0.f;
      vr[s] = (k0 + r < T) ? Vh[(long long)(k0 + r) * ldv + j] : 0.f;
    }
  };
  auto store_kv = [&]() {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int e = tid + 256 * s;
      const int r = e >> 6, j = e & 63;
      Ks[r * kAP + j] = kr[s];
      Vs[r * kAP + j] = vr[s];
    }
  };

  f32x16 o, o1;                             // two independent accumulation chains (even / odd key pairs), summed at the end
#pragma unroll
  for (int r = 0; r < 16; ++r) { o[r] = 0.f; o1[r] = 0.f; }
  float m_run = -1.0e30f, l_run = 0.f;       // per query column wn*32 + li (identical in the lanes / waves that share a column)
  const int ntiles = (T + 63) / 64;
  const unsigned long long t_begin = ATICK();
  load_kv(0);
  for (int it = 0; it < ntiles; ++it) {
    const int k0 = it * 64;
    const unsigned long long t0 = ATICK();
    lds_barrier();                          // previous tile's Ks / Vs / Ps reads are done
    store_kv();
    lds_barrier();
    if (it + 1 < ntiles) load_kv(k0 + 64);
    const unsigned long long t1 = ATICK();
    ATACC(1, t1 - t0);
    // ---- S = K^T Q for this wave's 32 keys x 32 queries
    f32x16 sacc, sacc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sacc[r] = 0.f; sacc1[r] = 0.f; }
#pragma unroll 4
    for (int kk = 0; kk < kAD; kk += 4) {
      const float a0 = Ks[(kk + lh) * kAP + wm * 32 + li], a1 = Ks[(kk + 2 + lh) * kAP + wm * 32 + li];
      const float b0 = Qs[(kk + lh) * kAP + wn * 32 + li], b1 = Qs[(kk + 2 + lh) * kAP + wn * 32 + li];
      sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, sacc, 0, 0, 0);
      sacc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, sacc1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[r] += sacc1[r];
    const unsigned long long t2 = ATICK();
    ATACC(2, t2 - t1);
    // rows of sacc: key = k0 + wm*32 + (r&3) + 8(r>>2) + 4 lh; column: query wn*32 + li
    if (k0 + 64 > T) {                        // only the last key tile has a masked tail
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = k0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (key >= T) sacc[r] = -1.0e30f;
      }
    }
    float mx = -1.0e30f;
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    if (lh == 0) red[0][wm][wn * 32 + li] = mx;
    lds_barrier();
    const float m_tile = fmaxf(red[0][0][wn * 32 + li], red[0][1][wn * 32 + li]);
    const float m_new = fmaxf(m_run, m_tile);
    const float alpha = expf(m_run - m_new);
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pv = expf(sacc[r] - m_new);             // masked keys: exp(-1e30 - m) = 0
      ps += pv;
      Ps[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * kAP + wn * 32 + li] = pv;
    }
    ps += __shfl_xor(ps, 32);
    if (lh == 0) red[1][wm][wn * 32 + li] = ps;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[r] *= alpha; o1[r] *= alpha; }
    lds_barrier();
    l_run = l_run * alpha + red[1][0][wn * 32 + li] + red[1][1][wn * 32 + li];
    m_run = m_new;
    const unsigned long long t3 = ATICK();
    ATACC(3, t3 - t2);
    // ---- O += V^T P : rows d = wm*32 + .., columns queries wn*32 + li
#pragma unroll 4
    for (int kk = 0; kk < 64; kk += 4) {
      const float a0 = Vs[(kk + lh) * kAP + wm * 32 + li], a1 = Vs[(kk + 2 + lh) * kAP + wm * 32 + li];
      const float b0 = Ps[(kk + lh) * kAP + wn * 32 + li], b1 = Ps[(kk + 2 + lh) * kAP + wn * 32 + li];
      o = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, o, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, o1, 0, 0, 0);
    }
    ATACC(4, ATICK() - t3);
  }
  ATACC(6, ATICK() - t_begin); ATACC(0, 1);
  const int q = q0 + wn * 32 + li;
  if (q < T) {
    const float inv = 1.f / l_run;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int d = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      out[(long long)(h * kAD + d) * ldo + q] = (o[r] + o1[r]) * inv + (bv ? bv[h * kAD + d] : 0.f);
    }
  }
}

// Q, K: channel-major [heads*64][T] (row pitch ldqk); V: row-major [T][heads*64] (row pitch ldv); out channel-major [heads*64][T].
void attention_fused(hipStream_t s, const float* Q, const float* K, long long ldqk, const float* V, long long ldv, const float* bv,
                     float* out, long long ldo, int heads, int dhead, int T) {
  RVC_REQUIRE(dhead == kAD, "fused attention is built for head dimension 64");
  hipLaunchKernelGGL(attention_kernel, dim3((T + 63) / 64, heads), dim3(256), 0, s, Q, K, ldqk, V, ldv, bv, out, ldo, T);
}

}  // namespace rvc
