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
constexpr int kAP = 68;        // LDS row pitch (floats): 16-byte aligned rows, 4-bank shift per row -> conflict-free ds_read_b128

typedef float f32x4 __attribute__((ext_vector_type(4)));

// LDS tiles are stored with the MFMA reduction index contiguous:  Qs[query][d], Ks[key][d], Vs[d][key], Ps[query][key].
// An MFMA step consumes the index pair (i, 32 + i) - lane half 0 takes i, half 1 takes 32 + i; any pairing is valid as long as A and
// B agree - so each lane reads its 32 values of a row as eight ds_read_b128, one per four MFMA steps.
__global__ __launch_bounds__(256) void attention_kernel(const float* __restrict__ Q, const float* __restrict__ K, long long ldqk,
                                                        const float* __restrict__ V, long long ldv, const float* __restrict__ bv,
                                                        float* __restrict__ out, long long ldo, int T) {
  __shared__ __attribute__((aligned(16))) float Qs[64 * kAP], Ks[64 * kAP], Vs[kAD * kAP], Ps[64 * kAP];
  __shared__ float red[2][2][64];          // [stat: max | sum][wm][query]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int h = blockIdx.y, q0 = blockIdx.x * 64;
  const float* Qh = Q + (long long)h * kAD * ldqk;
  const float* Kh = K + (long long)h * kAD * ldqk;
  const float* Vh = V + h * kAD;

  // Q tile -> Qs[query][d]: a thread takes 4 consecutive d of one query (4 coalesced loads) and writes them as one 16-byte store
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int e = tid + 256 * s, j = e & 63, d0 = (e >> 6) * 4;
    f32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (q0 + j < T) ? Qh[(long long)(d0 + i) * ldqk + q0 + j] : 0.f;
    *reinterpret_cast<f32x4*>(Qs + j * kAP + d0) = v;
  }
  f32x4 kr[4], vr[4];
  auto load_kv = [&](int k0) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int e = tid + 256 * s, j = e & 63, g4 = (e >> 6) * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        kr[s][i] = (k0 + j < T) ? Kh[(long long)(g4 + i) * ldqk + k0 + j] : 0.f;            // K[d = g4 + i][key = j]
        vr[s][i] = (k0 + g4 + i < T) ? Vh[(long long)(k0 + g4 + i) * ldv + j] : 0.f;        // V[key = g4 + i][d = j]
      }
    }
  };
  auto store_kv = [&]() {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int e = tid + 256 * s, j = e & 63, g4 = (e >> 6) * 4;
      *reinterpret_cast<f32x4*>(Ks + j * kAP + g4) = kr[s];       // Ks[key j][d g4..]
      *reinterpret_cast<f32x4*>(Vs + j * kAP + g4) = vr[s];       // Vs[d j][key g4..]
    }
  };

  f32x16 o, o1;                             // two independent accumulation chains, summed at the end
#pragma unroll
  for (int r = 0; r < 16; ++r) { o[r] = 0.f; o1[r] = 0.f; }
  float m_run = -1.0e30f, l_run = 0.f;       // per query column wn*32 + li (identical in the lanes / waves that share a column)
  const int ntiles = (T + 63) / 64;
  const float* arow_s = Ks + (wm * 32 + li) * kAP + lh * 32;     // A of S: key row
  const float* brow_s = Qs + (wn * 32 + li) * kAP + lh * 32;     // B of S: query row
  const float* arow_o = Vs + (wm * 32 + li) * kAP + lh * 32;     // A of O: d row
  const float* brow_o = Ps + (wn * 32 + li) * kAP + lh * 32;     // B of O: query row
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
    {
      f32x4 a[8], b[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) { a[c] = *reinterpret_cast<const f32x4*>(arow_s + 4 * c); b[c] = *reinterpret_cast<const f32x4*>(brow_s + 4 * c); }
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][0], b[c][0], sacc, 0, 0, 0);
        sacc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][1], b[c][1], sacc1, 0, 0, 0);
        sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][2], b[c][2], sacc, 0, 0, 0);
        sacc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][3], b[c][3], sacc1, 0, 0, 0);
      }
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
    for (int g = 0; g < 4; ++g) {
      f32x4 pv;
#pragma unroll
      for (int i = 0; i < 4; ++i) { pv[i] = expf(sacc[4 * g + i] - m_new); ps += pv[i]; }     // masked keys: exp(-1e30 - m) = 0
      *reinterpret_cast<f32x4*>(Ps + (wn * 32 + li) * kAP + wm * 32 + 8 * g + 4 * lh) = pv;   // Ps[query][key .. key + 3]
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
    {
      f32x4 a[8], b[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) { a[c] = *reinterpret_cast<const f32x4*>(arow_o + 4 * c); b[c] = *reinterpret_cast<const f32x4*>(brow_o + 4 * c); }
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        o = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][0], b[c][0], o, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][1], b[c][1], o1, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][2], b[c][2], o, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][3], b[c][3], o1, 0, 0, 0);
      }
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

// ---- bf16x3 variant: Q, K, V and P are split into bf16 hi / lo while they are staged in LDS and both products run as
// hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation (the arithmetic of conv_x3.hip): a 32 x 32 x 64 block
// costs 12 MFMAs of 8 passes instead of 32 fp32 MFMAs of 16.
constexpr int kXP = 72;        // LDS row pitch in bf16 elements (144 B)
typedef __bf16 bf16x8a __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4a __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned bf16_bits_a(__bf16 h) { return (unsigned)__builtin_bit_cast(unsigned short, h); }
__device__ __forceinline__ void split_store(const f32x4& v, unsigned short* hi, unsigned short* lo) {
  unsigned h[2], l[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const float a = v[2 * j], b = v[2 * j + 1];
    const __bf16 ah = (__bf16)a, bh = (__bf16)b;
    const __bf16 al = (__bf16)(a - (float)ah), bl = (__bf16)(b - (float)bh);
    h[j] = bf16_bits_a(ah) | (bf16_bits_a(bh) << 16);
    l[j] = bf16_bits_a(al) | (bf16_bits_a(bl) << 16);
  }
  *reinterpret_cast<unsigned long long*>(hi) = (unsigned long long)h[0] | ((unsigned long long)h[1] << 32);
  *reinterpret_cast<unsigned long long*>(lo) = (unsigned long long)l[0] | ((unsigned long long)l[1] << 32);
}
// acc0 / acc1 += A B^T over 64 reduction indices (4 steps of 16); two accumulation chains
__device__ __forceinline__ void x3_block(const unsigned short* ah_, const unsigned short* al_, const unsigned short* bh_, const unsigned short* bl_,
                                         f32x16& acc0, f32x16& acc1) {
  u32x4a ah[4], al[4], bh[4], bl[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    ah[c] = *reinterpret_cast<const u32x4a*>(ah_ + 16 * c); al[c] = *reinterpret_cast<const u32x4a*>(al_ + 16 * c);
    bh[c] = *reinterpret_cast<const u32x4a*>(bh_ + 16 * c); bl[c] = *reinterpret_cast<const u32x4a*>(bl_ + 16 * c);
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    f32x16& acc = (c & 1) ? acc1 : acc0;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8a, al[c]), __builtin_bit_cast(bf16x8a, bh[c]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8a, ah[c]), __builtin_bit_cast(bf16x8a, bl[c]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8a, ah[c]), __builtin_bit_cast(bf16x8a, bh[c]), acc, 0, 0, 0);
  }
}

__global__ __launch_bounds__(256) void attention_x3_kernel(const float* __restrict__ Q, const float* __restrict__ K, long long ldqk,
                                                        const float* __restrict__ V, long long ldv, const float* __restrict__ bv,
                                                        float* __restrict__ out, long long ldo, int T,
                                                        unsigned char* __restrict__ img, long long img_tp, int img_margin) {
  // bf16 hi / lo planes, rows of 64 reduction indices (128 B) + 16 B pad: ds_read_b128 of 32 consecutive rows is conflict-free
  __shared__ __attribute__((aligned(16))) unsigned short Qs[2][64 * kXP], Ks[2][64 * kXP], Vs[2][kAD * kXP], Ps[2][64 * kXP];
  __shared__ float red[2][2][64];          // [stat: max | sum][wm][query]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int h = blockIdx.y, q0 = blockIdx.x * 64;
  const float* Qh = Q + (long long)h * kAD * ldqk;
  const float* Kh = K + (long long)h * kAD * ldqk;
  const float* Vh = V + h * kAD;

  // Q tile -> Qs[query][d]: a thread takes 4 consecutive d of one query (4 coalesced loads) and writes them as one 16-byte store
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int e = tid + 256 * s, j = e & 63, d0 = (e >> 6) * 4;
    f32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (q0 + j < T) ? Qh[(long long)(d0 + i) * ldqk + q0 + j] : 0.f;
    split_store(v, &Qs[0][j * kXP + d0], &Qs[1][j * kXP + d0]);
  }
  f32x4 kr[4], vr[4];
  auto load_kv = [&](int k0) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int e = tid + 256 * s, j = e & 63, g4 = (e >> 6) * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        kr[s][i] = (k0 + j < T) ? Kh[(long long)(g4 + i) * ldqk + k0 + j] : 0.f;            // K[d = g4 + i][key = j]
        vr[s][i] = (k0 + g4 + i < T) ? Vh[(long long)(k0 + g4 + i) * ldv + j] : 0.f;        // V[key = g4 + i][d = j]
      }
    }
  };
  auto store_kv = [&]() {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int e = tid + 256 * s, j = e & 63, g4 = (e >> 6) * 4;
      split_store(kr[s], &Ks[0][j * kXP + g4], &Ks[1][j * kXP + g4]);       // Ks[key j][d g4..]
      split_store(vr[s], &Vs[0][j * kXP + g4], &Vs[1][j * kXP + g4]);       // Vs[d j][key g4..]
    }
  };

  f32x16 o, o1;                             // two independent accumulation chains, summed at the end
#pragma unroll
  for (int r = 0; r < 16; ++r) { o[r] = 0.f; o1[r] = 0.f; }
  float m_run = -1.0e30f, l_run = 0.f;       // per query column wn*32 + li (identical in the lanes / waves that share a column)
  const int ntiles = (T + 63) / 64;
  // a v_mfma_f32_32x32x16_bf16 lane (i, half) holds reduction indices 16 c + 8 half .. + 7 of row i: one ds_read_b128 per plane
  const int aoff_s = (wm * 32 + li) * kXP + lh * 8;              // A of S: key row
  const int boff_s = (wn * 32 + li) * kXP + lh * 8;              // B of S: query row
  const int aoff_o = (wm * 32 + li) * kXP + lh * 8;              // A of O: d row
  const int boff_o = (wn * 32 + li) * kXP + lh * 8;              // B of O: query row
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
    x3_block(&Ks[0][aoff_s], &Ks[1][aoff_s], &Qs[0][boff_s], &Qs[1][boff_s], sacc, sacc1);
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
    for (int g = 0; g < 4; ++g) {
      f32x4 pv;
#pragma unroll
      for (int i = 0; i < 4; ++i) { pv[i] = expf(sacc[4 * g + i] - m_new); ps += pv[i]; }     // masked keys: exp(-1e30 - m) = 0
      split_store(pv, &Ps[0][(wn * 32 + li) * kXP + wm * 32 + 8 * g + 4 * lh], &Ps[1][(wn * 32 + li) * kXP + wm * 32 + 8 * g + 4 * lh]);   // Ps[query][key .. key + 3]
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
    x3_block(&Vs[0][aoff_o], &Vs[1][aoff_o], &Ps[0][boff_o], &Ps[1][boff_o], o, o1);
    ATACC(4, ATICK() - t3);
  }
  ATACC(6, ATICK() - t_begin); ATACC(0, 1);
  const int q = q0 + wn * 32 + li;
  const float inv = 1.f / l_run;
  float val[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int d = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    val[r] = (o[r] + o1[r]) * inv + (bv ? bv[h * kAD + d] : 0.f);
  }
  if (out != nullptr && q < T) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int d = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      out[(long long)(h * kAD + d) * ldo + q] = val[r];
    }
  }
  if (img != nullptr) {
    // the split-resident image the out-projection GEMM stages (conv_x3s.hip): [16-channel chunk][hi | lo][8-channel half][margin + q][8 ch].
    // A lane holds rows {0..3, 8..11} + 4 lh of each 16-row chunk; v_permlane32_swap trades quads with the lane 32 away so that every
    // lane owns one 16-byte row of a half-plane (same exchange as ysplit_epilogue, conv_x3_dev.h).
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2) {
      unsigned hA[2], lA[2], hB[2], lB[2];
#pragma unroll
      for (int e2 = 0; e2 < 2; ++e2) {
        auto split = [](float a, float b, unsigned& hi, unsigned& lo) {
          const __bf16 ah = (__bf16)a, bh = (__bf16)b;
          const __bf16 al = (__bf16)(a - (float)ah), bl = (__bf16)(b - (float)bh);
          hi = bf16_bits_a(ah) | (bf16_bits_a(bh) << 16); lo = bf16_bits_a(al) | (bf16_bits_a(bl) << 16);
        };
        split(val[8 * g2 + 2 * e2], val[8 * g2 + 2 * e2 + 1], hA[e2], lA[e2]);
        split(val[8 * g2 + 4 + 2 * e2], val[8 * g2 + 5 + 2 * e2], hB[e2], lB[e2]);
      }
      u32x4a hi, lo;
#pragma unroll
      for (int e2 = 0; e2 < 2; ++e2) {
        const u32x2_t sh = __builtin_amdgcn_permlane32_swap(hA[e2], hB[e2], false, false);
        const u32x2_t sl2 = __builtin_amdgcn_permlane32_swap(lA[e2], lB[e2], false, false);
        hi[e2] = sh.x; hi[2 + e2] = sh.y; lo[e2] = sl2.x; lo[2 + e2] = sl2.y;
      }
      if (q < T) {
        const long long chunk = (h * kAD + wm * 32) / 16 + g2;
        unsigned char* row = img + ((chunk * 4 + lh) * img_tp + img_margin + q) * 16;
        *reinterpret_cast<u32x4a*>(row) = hi;
        *reinterpret_cast<u32x4a*>(row + img_tp * 32) = lo;
      }
    }
  }
}

// Q, K: channel-major [heads*64][T] (row pitch ldqk); V: row-major [T][heads*64] (row pitch ldv); out channel-major [heads*64][T].
void attention_fused(hipStream_t s, const float* Q, const float* K, long long ldqk, const float* V, long long ldv, const float* bv,
                     float* out, long long ldo, int heads, int dhead, int T, unsigned char* out_img, long long img_tp) {
  RVC_REQUIRE(dhead == kAD, "fused attention is built for head dimension 64");
  static const bool x3 = (exp_int("RVC_ATT_X3", 1) != 0);
  RVC_REQUIRE(out != nullptr || out_img != nullptr, "fused attention: no output");
  RVC_REQUIRE(out_img == nullptr || x3, "the split-image output needs the bf16x3 attention kernel");
  if (x3) { hipLaunchKernelGGL(attention_x3_kernel, dim3((T + 63) / 64, heads), dim3(256), 0, s, Q, K, ldqk, V, ldv, bv, out, ldo, T, out_img, img_tp, (int)kSplitMargin); return; }
  hipLaunchKernelGGL(attention_kernel, dim3((T + 63) / 64, heads), dim3(256), 0, s, Q, K, ldqk, V, ldv, bv, out, ldo, T);
}

// ================================================================================================ key-split variant
// attention_ks_kernel<D, WK, REL>: one workgroup = 32 queries of one head, WK waves; wave w owns keys [32 w, 32 w + 32) of every
// 32 WK-key tile and keeps its OWN online-softmax state (m, l, O) over that key subset, so an iteration needs no cross-wave
// exchange at all - only the two barriers around the shared K / V tile.  The WK partial results are merged once at the end:
//   m = max_w m_w,  l = sum_w l_w e^(m_w - m),  O = sum_w O_w e^(m_w - m).
// REL adds the windowed relative-position terms of the synthesizer's text encoder (reference attentions.py:230-267): scores of keys
// within +-win of the query get rel[k - q + win][q] added (rel = Q . E_k, computed by a 21-row projection beforehand), and the
// normalised probabilities of that band are returned as pb[r][q] = P[q][q + r - win] for the value-side projection.  Band scores are
// kept raw in LDS and normalised with the final (m, l), so the online rescaling never touches them.
template <int D, int WK, bool REL>
__global__ __launch_bounds__(64 * WK) void attention_ks_kernel(const float* __restrict__ Q, const float* __restrict__ K, long long ldqk,
                                                               const float* __restrict__ V, long long ldv, const float* __restrict__ bv,
                                                               const float* __restrict__ rel, float* __restrict__ pb, int win,
                                                               float* __restrict__ out, long long ldo, int T,
                                                               const float* __restrict__ ek, const float* __restrict__ ev) {
  // ek / ev (REL only): emb_rel_k / emb_rel_v [2 win + 1][D] - the two relative-position projections computed HERE instead of by four small
  // convolution launches per layer: rq[r][query] = E_k[r] . Q[query] right after the Q tile is staged (`rel` is then not read), and
  // out[d][query] += sum_r P_band[r][query] E_v[r][d] in the merge (`pb` may then be null).
  constexpr int NT = 64 * WK, KT = 32 * WK, DP = D + 4, VP = KT + 4, PP = 36, DT = D / 32, OP = 33;
  constexpr int kQs = 32 * DP, kKs = KT * DP, kVs = D * VP, kPs = WK * 32 * PP, kSb = REL ? 3 * 21 * 32 : 0;      // raw band scores | rq | band probabilities
  static_assert(WK * (D * OP + 64) <= kKs + kVs, "merge buffers must fit the K / V tiles");
  extern __shared__ __attribute__((aligned(16))) float smem_att[];
  float* Qs = smem_att; float* Ks = Qs + kQs; float* Vs = Ks + kKs; float* Ps = Vs + kVs; float* sb = Ps + kPs;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int h = blockIdx.y, q0 = blockIdx.x * 32;
  const float* Qh = Q + (long long)h * D * ldqk;
  const float* Kh = K + (long long)h * D * ldqk;
  const float* Vh = V + h * D;
  const float* relh = (REL && rel) ? rel + (long long)h * (2 * win + 1) * T : nullptr;
  float* rq = sb + 21 * 32; float* pbs = sb + 2 * 21 * 32;

  // Q tile -> Qs[query][d]
  for (int e = tid; e < 32 * D / 4; e += NT) {
    const int j = e & 31, d0 = (e >> 5) * 4;
    f32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (q0 + j < T) ? Qh[(long long)(d0 + i) * ldqk + q0 + j] : 0.f;
    *reinterpret_cast<f32x4*>(Qs + j * DP + d0) = v;
  }
  if (REL && ek) {
    lds_barrier();                          // the Q tile is staged
    for (int e = tid; e < (2 * win + 1) * 32; e += NT) {
      const int rr = e >> 5, j = e & 31;
      const float* er = ek + rr * D; const float* qr = Qs + j * DP;
      float a0 = 0.f, a1 = 0.f;
#pragma unroll 8
      for (int d = 0; d < D; d += 2) { a0 = fmaf(er[d], qr[d], a0); a1 = fmaf(er[d + 1], qr[d + 1], a1); }
      rq[e] = a0 + a1;
    }
  }
  f32x16 o[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
  float m_run = -1.0e30f, l_run = 0.f;
  const float* arow_s = Ks + (w * 32 + li) * DP + lh * (D / 2);
  const float* brow_s = Qs + li * DP + lh * (D / 2);
  float* prow = Ps + (w * 32 + li) * PP;
  const int q = q0 + li;
  const int ntiles = (T + KT - 1) / KT;
  // K / V tiles travel global -> registers -> LDS; the registers of tile it + 1 are requested before tile it is consumed, so the
  // HBM / L2 latency hides behind the two MFMA chains (one workgroup per CU: nothing else would cover it)
  constexpr int NL = KT * D / 4 / NT;
  static_assert(KT * D / 4 % NT == 0, "tile tasks must divide evenly");
  f32x4 kr[NL], vr[NL];
  auto load_tile = [&](int k0) {
#pragma unroll
    for (int x = 0; x < NL; ++x) {
      const int e = tid + NT * x;
      const int j = e % KT, d0 = (e / KT) * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) kr[x][i] = (k0 + j < T) ? Kh[(long long)(d0 + i) * ldqk + k0 + j] : 0.f;
    }
#pragma unroll
    for (int x = 0; x < NL; ++x) {
      const int e = tid + NT * x;
      const int j = e % D, g4 = (e / D) * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) vr[x][i] = (k0 + g4 + i < T) ? Vh[(long long)(k0 + g4 + i) * ldv + j] : 0.f;
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int x = 0; x < NL; ++x) {
      const int e = tid + NT * x;
      *reinterpret_cast<f32x4*>(Ks + (e % KT) * DP + (e / KT) * 4) = kr[x];      // Ks[key][d .. d + 3]
    }
#pragma unroll
    for (int x = 0; x < NL; ++x) {
      const int e = tid + NT * x;
      *reinterpret_cast<f32x4*>(Vs + (e % D) * VP + (e / D) * 4) = vr[x];        // Vs[d][key .. key + 3]
    }
  };
  load_tile(0);
  for (int it = 0; it < ntiles; ++it) {
    const int k0 = it * KT;
    lds_barrier();                          // previous tile's Ks / Vs reads are done
    store_tile();
    lds_barrier();
    if (it + 1 < ntiles) load_tile(k0 + KT);
    const int kw = k0 + w * 32;             // first key of this wave's slice
    if (kw < T) {                           // (wave-uniform) slices past the end contribute nothing
      // ---- S = K^T Q : 32 keys x 32 queries, reduction over d with the pairing (i, D/2 + i)
      f32x16 s0, s1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
#pragma unroll
      for (int c = 0; c < D / 8; ++c) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(arow_s + 4 * c), b = *reinterpret_cast<const f32x4*>(brow_s + 4 * c);
        s0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], s1, 0, 0, 0);
        s0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[2], s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[3], s1, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) s0[r] += s1[r];
      // rows: key = kw + (r&3) + 8(r>>2) + 4 lh; column: query q
      if (REL && kw + 32 + win > q0 && kw < q0 + 32 + win + 1) {          // slice intersects the band of this query group
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kw + (r & 3) + 8 * (r >> 2) + 4 * lh;
          const int rr = key - q + win;
          if (rr >= 0 && rr <= 2 * win && key < T && q < T) {
            s0[r] += relh ? relh[(long long)rr * T + q] : rq[rr * 32 + li];
            sb[rr * 32 + li] = s0[r];
          }
        }
      }
      if (kw + 32 > T) {
#pragma unroll
        for (int r = 0; r < 16; ++r) if (kw + (r & 3) + 8 * (r >> 2) + 4 * lh >= T) s0[r] = -1.0e30f;
      }
      float mx = -1.0e30f;
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s0[r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float m_new = fmaxf(m_run, mx);
      const float alpha = expf(m_run - m_new);
      float ps = 0.f;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 pv;
#pragma unroll
        for (int i = 0; i < 4; ++i) { pv[i] = expf(s0[4 * g + i] - m_new); ps += pv[i]; }
        *reinterpret_cast<f32x4*>(prow + 8 * g + 4 * lh) = pv;           // Ps[w][query][key_local .. + 3]
      }
      ps += __shfl_xor(ps, 32);
      l_run = l_run * alpha + ps;
      m_run = m_new;
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // this wave's P is in LDS (wave-private region)
      // ---- O[d][q] += V[d][keys of this slice] . P : reduction over the slice's 32 keys with the pairing (i, 16 + i)
      f32x4 pb4[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) pb4[c] = *reinterpret_cast<const f32x4*>(prow + lh * 16 + 4 * c);
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        const float* vrow = Vs + (t * 32 + li) * VP + w * 32 + lh * 16;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(vrow + 4 * c);
#pragma unroll
          for (int i = 0; i < 4; ++i) o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], pb4[c][i], o[t], 0, 0, 0);
        }
      }
    }
  }
  // ---- merge the WK partial states: O_w, m_w, l_w through LDS (re-using the K / V tiles)
  lds_barrier();
  float* Oc = Ks;                                   // [WK][D][OP]
  float* ml = Ks + WK * D * OP;                     // [WK][2][32]
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) Oc[(w * D + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * OP + li] = o[t][r];
  if (lh == 0) { ml[(w * 2 + 0) * 32 + li] = m_run; ml[(w * 2 + 1) * 32 + li] = l_run; }
  lds_barrier();
  if (REL) {                                    // band probabilities first: the value-side projection below needs them
    for (int e = tid; e < (2 * win + 1) * 32; e += NT) {
      const int rr = e >> 5, j = e & 31;
      const int qq = q0 + j, key = qq + rr - win;
      float m = -1.0e30f, l = 0.f;
#pragma unroll
      for (int x = 0; x < WK; ++x) m = fmaxf(m, ml[(x * 2) * 32 + j]);
#pragma unroll
      for (int x = 0; x < WK; ++x) l += ml[(x * 2 + 1) * 32 + j] * expf(ml[(x * 2) * 32 + j] - m);
      const float pv = (qq < T && key >= 0 && key < T) ? expf(sb[rr * 32 + j] - m) / l : 0.f;
      pbs[e] = pv;
      if (pb && qq < T) pb[((long long)h * (2 * win + 1) + rr) * T + qq] = pv;
    }
    lds_barrier();
  }
  for (int e = tid; e < D * 32; e += NT) {
    const int d = e >> 5, j = e & 31;
    float m = -1.0e30f;
#pragma unroll
    for (int x = 0; x < WK; ++x) m = fmaxf(m, ml[(x * 2) * 32 + j]);
    float l = 0.f, acc = 0.f;
#pragma unroll
    for (int x = 0; x < WK; ++x) {
      const float sc = expf(ml[(x * 2) * 32 + j] - m);
      l += ml[(x * 2 + 1) * 32 + j] * sc;
      acc += Oc[(x * D + d) * OP + j] * sc;
    }
    float v = acc / l + (bv ? bv[h * D + d] : 0.f);
    if (REL && ev) {
      float r0 = 0.f;
      for (int rr = 0; rr < 2 * win + 1; ++rr) r0 = fmaf(pbs[rr * 32 + j], ev[rr * D + d], r0);
      v += r0;
    }
    if (q0 + j < T) out[(long long)(h * D + d) * ldo + q0 + j] = v;
  }
}

template <int D, int WK, bool REL>
static void launch_att_ks(hipStream_t s, const float* Q, const float* K, long long ldqk, const float* V, long long ldv, const float* bv,
                          const float* rel, float* pb, int win, float* out, long long ldo, int heads, int T, const float* ek, const float* ev) {
  constexpr int KT = 32 * WK;
  const size_t lds = sizeof(float) * (32 * (D + 4) + KT * (D + 4) + D * (KT + 4) + WK * 32 * 36 + (REL ? 3 * 21 * 32 : 0));
  auto kern = attention_ks_kernel<D, WK, REL>;
  RVC_ALLOW_BIG_LDS(kern);
  hipLaunchKernelGGL(kern, dim3((T + 31) / 32, heads), dim3(64 * WK), lds, s, Q, K, ldqk, V, ldv, bv, rel, pb, win, out, ldo, T, ek, ev);
}

// Text-encoder attention of the synthesizer (2 heads x 96): relative-position bias rel [heads][2 win + 1][T] in, banded probabilities
// pb [heads][2 win + 1][T] out.  Q, K channel-major (Q pre-scaled), V row-major, out channel-major.
void attention_rel_fused(hipStream_t s, const float* Q, const float* K, long long ldqk, const float* V, long long ldv, const float* bv,
                         const float* rel, float* pb, int win, float* out, long long ldo, int heads, int dhead, int T, const float* ek, const float* ev) {
  RVC_REQUIRE(dhead == 96 && win == 10 && (rel || ek) && (pb || ev), "fused relative-position attention is built for head dimension 96, window 10");
  launch_att_ks<96, 4, true>(s, Q, K, ldqk, V, ldv, bv, rel, pb, win, out, ldo, heads, T, ek, ev);
}

}  // namespace rvc
