// attention_dma_kernel<D, NWQ, KS, REL> and its launcher: shared by attention_dma.hip (head dimension 64, built with MFMA accumulators in
// VGPRs) and attention_dma_rel.hip (96 + relative positions: that compiler option crashes on it).  Private to csrc/.
#pragma once
#include "conv_x3_dev.h"
#include "ops.h"

namespace rvc {

struct AttnDmaArgs {
  const unsigned char* QK; long long qkTp; unsigned qk_bytes;   // image holding the q and k channels
  int q_chunk0, k_chunk0;                                       // first 16-channel chunk of head 0's q / k
  const unsigned char* Vt; long long vtTp; unsigned vt_bytes;   // V^T image: [key chunk][plane][vtTp rows][16 B]; channel c of the model at row margin + c
  int margin, T;
  float scale;                                                  // scores are multiplied by this (1: q is pre-scaled)
  const float* bv; float* out; long long ldo;                   // + bv[c] after the normalisation; fp32 output [heads D][ldo] or null
  unsigned char* img; long long img_tp;                         // split output image or null
  // key split across workgroups (blockIdx.z = slice of the key tiles): partial (m, l, O) states as write-through slabs, a ticket per
  // (query tile, head); the workgroup that draws the last ticket merges the slices in slice order and runs the epilogue
  int kz; float* part; unsigned part_bytes; unsigned* tickets;
  int nqt, heads, xcd_remap;                                    // query tiles, heads (grid decoding; set by the launcher)
  // relative-position terms of the synthesizer's text encoder (REL): E_k as an image [D / 16 chunks][plane][32 rows r][8 d] (rows past
  // 2 win zero), E_v^T as [2 chunks of r][plane][D rows][8 r]; the raw band scores sband [heads][2 win + 1][T] (write-through scratch)
  int win; const unsigned char* ek_img; const unsigned char* evt_img; float* sband; unsigned sband_bytes;
};

#ifdef RVC_CONV_TIMING
static __device__ unsigned long long g_attd_timing[8];   // [0] workgroups, [1] prologue, [2] tile loop, [3] slab store + ticket, [4] merge, [5] epilogue, [6] total (all: first wave of each workgroup)
static inline void attd_timing_read_tu(unsigned long long* out8, bool reset) {      // this translation unit's copy
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_attd_timing), sizeof(unsigned long long) * 8);
  if (reset) { unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_attd_timing), z, sizeof(z)); }
}
#define DTICK() wall_clock64()
#define DTACC(i, v) do { if (threadIdx.x == 0) atomicAdd(&g_attd_timing[i], (unsigned long long)(v)); } while (0)
#else
#define DTICK() 0ull
#define DTACC(i, v) do {} while (0)
#endif

__device__ __forceinline__ void att_dma(__amdgpu_buffer_rsrc_t rs, unsigned char* lds_dst, int voffset, int soffset) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_dst, 16, voffset, soffset, 0, 0);
}
template <int D, int NWQ, int KS, bool REL>
__global__ __launch_bounds__(64 * NWQ * KS) void attention_dma_kernel(const AttnDmaArgs p) {
  constexpr int NC = D / 16, DB = D / 32;
  constexpr int KT_BYTES = NC * 4 * 1024, VT_BYTES = 16 * D * 16, BUF = KT_BYTES + VT_BYTES;
  constexpr int NPK = NC * 4, NPV = VT_BYTES / 1024, NPW = (NPK + NPV) / NWQ;
  static_assert(NPK % NWQ == 0 && NPV % NWQ == 0, "every wave's i-th piece is of one kind");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem_att[];

  const unsigned long long dt0 = DTICK();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_wg = __builtin_amdgcn_readfirstlane(tid >> 6);
  // KS wave groups share the workgroup's queries and take the key tiles round-robin, each with its own (m, l, O) and its own pair of
  // tile buffers: two waves per SIMD, one's softmax (VALU) under the other's MFMAs; the groups' states are merged once at the end
  const int grp = wave_wg / NWQ, wave = wave_wg - grp * NWQ;
  const int li = lane & 31, lh = lane >> 5;
  // 1-D grid, renumbered so that an XCD (consecutive hardware ids go round-robin to the 8 XCDs, each with its own 4 MB L2) works on ONE
  // contiguous run of (key slice, head, query tile) with the query tile fastest: the workgroups that stream the same K / V^T tiles share an L2
  // (with (x, y, z) = (query tile, head, slice) every XCD touched every head's K / V: 4.9 - 9.8 MB through each 4 MB L2)
  const unsigned bid = p.xcd_remap ? xcd_tile(blockIdx.x, gridDim.x) : blockIdx.x;
  const int qtx = (int)(bid % (unsigned)p.nqt), hz = (int)(bid / (unsigned)p.nqt);
  const int h = hz % p.heads;
  const int q0 = (qtx * NWQ + wave) * 32;
  const int T = p.T;
  unsigned char* const smem_g = smem_att + grp * (2 * BUF);

  // ---- DMA pieces of a tile: K (chunk c, plane) = 64 keys x 16 B; V^T flat rows ((key chunk, plane), channel)
  const __amdgpu_buffer_rsrc_t krs = make_rsrc(p.QK, p.qk_bytes), vrs = make_rsrc(p.Vt, p.vt_bytes);
  int voff[NPW];
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int pi = wave + NWQ * i;
    if (i * NWQ < NPK) {
      voff[i] = (int)(((long long)(p.k_chunk0 + h * NC) * 4 + pi) * p.qkTp + p.margin + lane) * 16;
    } else {
      const int fr = (pi - NPK) * 64 + lane, kp = fr / D, d = fr - kp * D;      // kp = key chunk * 4 + plane
      voff[i] = (int)((long long)kp * p.vtTp + p.margin + h * D + d) * 16;
    }
  }
  const int vstep = (int)(p.vtTp * 16 * 16);                   // 4 key chunks x 4 planes of V^T per tile
  auto issue = [&](int tile, int buf) {
    unsigned char* base = smem_g + buf * BUF;
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int pi = wave + NWQ * i;
      if (i * NWQ < NPK) att_dma(krs, base + pi * 1024, voff[i], tile * 1024);
      else att_dma(vrs, base + KT_BYTES + (pi - NPK) * 1024, voff[i], tile * vstep);
    }
  };
  // this workgroup's slice of the key tiles
  const int ntiles_all = (T + 63) / 64, kz = p.kz, z = hz / p.heads;
  const int tile0 = (int)((long long)z * ntiles_all / kz), ntiles = (int)((long long)(z + 1) * ntiles_all / kz) - tile0, nsteps = (ntiles + KS - 1) / KS;
  if (grp < ntiles) issue(tile0 + grp, 0);

  // ---- this wave's queries: B operand of S, resident in registers
  u32x4 qh[NC], ql[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const unsigned char* row = p.QK + ((((long long)(p.q_chunk0 + h * NC + c)) * 4 + lh) * p.qkTp + p.margin + q0 + li) * 16;
    qh[c] = *reinterpret_cast<const u32x4*>(row);
    ql[c] = *reinterpret_cast<const u32x4*>(row + p.qkTp * 32);
  }

  // ---- REL: rq[r][query] = E_k[r] . Q[query] (one 32 x 32 MFMA block) -> this wave's LDS table, read by band position below
  float* rqs = reinterpret_cast<float*>(smem_att + KS * 2 * BUF) + wave_wg * 1024;
  const __amdgpu_buffer_rsrc_t brs = make_rsrc(p.sband, REL ? p.sband_bytes : 0u);
  if (REL) {
    f32x16 rq;
#pragma unroll
    for (int r = 0; r < 16; ++r) rq[r] = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const unsigned char* a = p.ek_img + ((c * 4 + lh) * 32 + li) * 16;
      const u32x4 eh = *reinterpret_cast<const u32x4*>(a), el = *reinterpret_cast<const u32x4*>(a + 2 * 32 * 16);
      rq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, eh), __builtin_bit_cast(bf16x8, ql[c]), rq, 0, 0, 0);
      rq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, el), __builtin_bit_cast(bf16x8, qh[c]), rq, 0, 0, 0);
      rq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, eh), __builtin_bit_cast(bf16x8, qh[c]), rq, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) rqs[((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + li] = rq[r];
  }
  f32x16 o[DB];
#pragma unroll
  for (int db = 0; db < DB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
  const float c2 = p.scale * 1.4426950408889634f;              // exp(s scale - m scale) = exp2((s - m) c2)
  float m_run = -3.0e38f, l_run = 0.f;

  const unsigned long long dt1 = DTICK();
  DTACC(1, dt1 - dt0);
  for (int st = 0; st < nsteps; ++st) {
    const int itl = st * KS + grp, it = tile0 + itl;           // tile inside the slice, absolute tile
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's pieces of the tile (and, the first time, its queries)
    lds_barrier();                                             // everyone's pieces; everyone is done reading the other buffer
    if (itl + KS < ntiles) issue(it + KS, (st + 1) & 1);
    if (itl >= ntiles) continue;                               // (the last step of a group that has run out of tiles: barriers only)
    const unsigned char* kb_ = smem_g + (st & 1) * BUF;
    const unsigned char* vb_ = kb_ + KT_BYTES;
    // ---- S[key][query] = K^T Q: two 32-key blocks
    f32x16 s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      u32x4 ah[2], al[2];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const unsigned char* a = kb_ + ((c * 4 + lh) * 64 + 32 * kb + li) * 16;
        ah[kb] = *reinterpret_cast<const u32x4*>(a);
        al[kb] = *reinterpret_cast<const u32x4*>(a + 2048);
      }
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[kb]), __builtin_bit_cast(bf16x8, ql[c]), s[kb], 0, 0, 0);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al[kb]), __builtin_bit_cast(bf16x8, qh[c]), s[kb], 0, 0, 0);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[kb]), __builtin_bit_cast(bf16x8, qh[c]), s[kb], 0, 0, 0);
    }
    // rows of s[kb]: key = 64 it + 32 kb + (r & 3) + 8 (r >> 2) + 4 lh; column: query q0 + li
    if (REL && it * 64 + 64 + p.win > q0 && it * 64 < q0 + 32 + p.win + 1) {      // (wave-uniform) the tile meets the band of this wave's queries
      const int qq = q0 + li;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {                         // straight-line: table read at a clamped index, store dropped out of range
          const int key = it * 64 + 32 * kb + (r & 3) + 8 * (r >> 2) + 4 * lh;
          const int rr = key - qq + p.win;
          const bool ok = rr >= 0 && rr <= 2 * p.win && key < T && qq < T;
          const float bias = rqs[(ok ? rr : 0) * 32 + li];
          s[kb][r] += ok ? bias : 0.f;
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(s[kb][r]), brs, ok ? (int)(((unsigned)(h * (2 * p.win + 1) + rr) * (unsigned)T + (unsigned)qq) * 4u) : (int)kOOB, 0, 16);   // sc1
        }
    }
    if (it * 64 + 64 > T) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = it * 64 + 32 * kb + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (key >= T) s[kb][r] = -3.0e38f;
        }
    }
    float mx = s[0][0];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    // the running maximum follows the true one only when that has grown by more than 2^8 (in the exponent's units): probabilities stay
    // below 2^8, exact in fp32 and in the hi / lo split alike, and after the first tiles O is hardly ever rescaled (wave-uniform skip)
    const float m_new = ((mx - m_run) * c2 > 8.f) ? mx : m_run;
    const bool grow = __builtin_amdgcn_ballot_w64(m_new != m_run) != 0ull;
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c2);
    const float mc = m_new * c2;
    float ps = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[kb][r] = __builtin_amdgcn_exp2f(fmaf(s[kb][r], c2, -mc)); ps += s[kb][r]; }      // masked keys: exp2(-huge) = 0
    ps += __shfl_xor(ps, 32);
    l_run = l_run * alpha + ps;
    m_run = m_new;
    if (grow) {
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
    }
    // ---- P as the B operand of P V: key chunk kc = 2 kb + g2, lane (query li, half lh) <- 8 keys x {hi, lo}
    u32x4 ph[4], pl[4];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int g2 = 0; g2 < 2; ++g2) {
        unsigned hA[2], lA[2], hB[2], lB[2];
#pragma unroll
        for (int e2 = 0; e2 < 2; ++e2) {
          split2(s[kb][8 * g2 + 2 * e2], s[kb][8 * g2 + 2 * e2 + 1], hA[e2], lA[e2]);
          split2(s[kb][8 * g2 + 4 + 2 * e2], s[kb][8 * g2 + 5 + 2 * e2], hB[e2], lB[e2]);
        }
#pragma unroll
        for (int e2 = 0; e2 < 2; ++e2) {
          typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
          const u32x2_t sh = __builtin_amdgcn_permlane32_swap(hA[e2], hB[e2], false, false);
          const u32x2_t sl2 = __builtin_amdgcn_permlane32_swap(lA[e2], lB[e2], false, false);
          ph[2 * kb + g2][e2] = sh.x; ph[2 * kb + g2][2 + e2] = sh.y; pl[2 * kb + g2][e2] = sl2.x; pl[2 * kb + g2][2 + e2] = sl2.y;
        }
      }
    // ---- O[d][query] += V^T P
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      u32x4 ah[DB], al[DB];
#pragma unroll
      for (int db = 0; db < DB; ++db) {
        const unsigned char* a = vb_ + ((kc * 4 + lh) * D + 32 * db + li) * 16;
        ah[db] = *reinterpret_cast<const u32x4*>(a);
        al[db] = *reinterpret_cast<const u32x4*>(a + 2 * D * 16);
      }
#pragma unroll
      for (int db = 0; db < DB; ++db) o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[db]), __builtin_bit_cast(bf16x8, pl[kc]), o[db], 0, 0, 0);
#pragma unroll
      for (int db = 0; db < DB; ++db) o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al[db]), __builtin_bit_cast(bf16x8, ph[kc]), o[db], 0, 0, 0);
#pragma unroll
      for (int db = 0; db < DB; ++db) o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[db]), __builtin_bit_cast(bf16x8, ph[kc]), o[db], 0, 0, 0);
    }
  }

  const unsigned long long dt2 = DTICK();
  DTACC(2, dt2 - dt1);
  if (KS > 1) {
    // ---- merge the groups' states into group 0 (through the dead tile buffers): m = max m_g, l = sum l_g 2^((m_g - m) c2), O likewise
    constexpr int NV = 2 + 16 * DB;
    float* xch = reinterpret_cast<float*>(smem_att);
    __syncthreads();
    if (grp > 0) {
      float* dst = xch + (((grp - 1) * NWQ + wave) * NV) * 64 + lane;
      dst[0] = m_run; dst[64] = l_run;
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[(2 + db * 16 + r) * 64] = o[db][r];
    }
    __syncthreads();
    if (grp > 0) return;
#pragma unroll
    for (int g = 1; g < KS; ++g) {
      const float* src = xch + (((g - 1) * NWQ + wave) * NV) * 64 + lane;
      const float m1 = src[0], l1 = src[64];
      const float m = fmaxf(m_run, m1);
      const float a0 = __builtin_amdgcn_exp2f((m_run - m) * c2), a1 = __builtin_amdgcn_exp2f((m1 - m) * c2);
      l_run = l_run * a0 + l1 * a1; m_run = m;
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = o[db][r] * a0 + src[(2 + db * 16 + r) * 64] * a1;
    }
  }
  if (kz > 1) {
    // ---- merge the key slices: slabs [slice][wave][quad][lane] of 16-byte granules (O in register order, then (m, l)), write-through;
    // the last arriver reads ALL slices back (its own included: one code path, the sum order is the slice order whoever is last),
    // a slice's 4 DB + 1 loads in flight together
    constexpr int NQ = 4 * DB + 1;
    const unsigned qt = (unsigned)(h * p.nqt + qtx);
    const __amdgpu_buffer_rsrc_t prs = make_rsrc(p.part, p.part_bytes);
    const unsigned slab = (unsigned)(NWQ * NQ * 64) * 16u;
    const unsigned lane_off = ((unsigned)(wave * NQ) * 64u + (unsigned)lane) * 16u;
    {
      const unsigned mine = (qt * (unsigned)kz + (unsigned)z) * slab + lane_off;
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const u32x4 v = {__float_as_uint(o[db][4 * qd]), __float_as_uint(o[db][4 * qd + 1]), __float_as_uint(o[db][4 * qd + 2]), __float_as_uint(o[db][4 * qd + 3])};
          __builtin_amdgcn_raw_buffer_store_b128(v, prs, (int)(mine + (unsigned)(db * 4 + qd) * 1024u), 0, 16);      // aux 16 = sc1
        }
      const u32x4 ml = {__float_as_uint(m_run), __float_as_uint(l_run), 0u, 0u};
      __builtin_amdgcn_raw_buffer_store_b128(ml, prs, (int)(mine + (unsigned)(4 * DB) * 1024u), 0, 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // every storing wave drains (slabs and band scores) before the ticket
    __syncthreads();
    unsigned* flag = reinterpret_cast<unsigned*>(smem_att);
    if (tid == 0) {
      const unsigned t = __hip_atomic_fetch_add(p.tickets + qt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (t == (unsigned)kz - 1u) __hip_atomic_store(p.tickets + qt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *flag = t;
    }
    __syncthreads();
    const unsigned long long dt3 = DTICK();
    DTACC(3, dt3 - dt2);
    if (*flag != (unsigned)kz - 1u) { DTACC(0, 1); DTACC(6, dt3 - dt0); return; }
    // up to KZB slices per batch, EVERY load of a batch in flight before the first is used (a slab lives in another XCD's write-through
    // path: one round trip of ~2 us per dependent batch); slices past kz read through a zero-extent offset and weigh 0
    constexpr int KZB = 5;
    float m = -3.0e38f, l = 0.f;
    f32x16 om[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) om[db][r] = 0.f;
    for (int z0 = 0; z0 < kz; z0 += KZB) {
      u32x4 v[KZB][NQ];
#pragma unroll
      for (int j = 0; j < KZB; ++j) {
        const bool in = z0 + j < kz;
        const unsigned base = (qt * (unsigned)kz + (unsigned)(z0 + j)) * slab + lane_off;
#pragma unroll
        for (int i = 0; i < NQ; ++i) v[j][i] = __builtin_amdgcn_raw_buffer_load_b128(prs, in ? (int)(base + (unsigned)i * 1024u) : (int)kOOB, 0, 16);
      }
      float mb = m;
#pragma unroll
      for (int j = 0; j < KZB; ++j) if (z0 + j < kz) mb = fmaxf(mb, __uint_as_float(v[j][4 * DB][0]));
      const float a0 = __builtin_amdgcn_exp2f((m - mb) * c2);       // what the earlier batches have summed, re-based (1 for the first batch: nothing summed)
      l *= a0;
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) om[db][r] *= a0;
      m = mb;
#pragma unroll
      for (int j = 0; j < KZB; ++j) {
        const float az = (z0 + j < kz) ? __builtin_amdgcn_exp2f((__uint_as_float(v[j][4 * DB][0]) - m) * c2) : 0.f;
        l = fmaf(__uint_as_float(v[j][4 * DB][1]), az, l);
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
          for (int r = 0; r < 16; ++r) om[db][r] = fmaf(__uint_as_float(v[j][db * 4 + (r >> 2)][r & 3]), az, om[db][r]);
      }
    }
#pragma unroll
    for (int db = 0; db < DB; ++db) o[db] = om[db];
    m_run = m; l_run = l;
  }
  const unsigned long long dt4 = DTICK();
  DTACC(4, dt4 - dt2);
  // ---- normalise, + bv; fp32 rows and / or the image the out-projection stages
  const int q = q0 + li;
  const float inv = 1.f / l_run;
#pragma unroll
  for (int db = 0; db < DB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int d = 32 * db + (r & 3) + 8 * (r >> 2) + 4 * lh;
      o[db][r] = o[db][r] * inv + (p.bv ? p.bv[h * D + d] : 0.f);
    }
  if (REL) {
    // ---- value side of the relative positions: out[d][q] += sum_r P[q][q + r - win] E_v[r][d].  Lane (query, half) computes the 8 + 8
    // band probabilities of its own B-operand rows from the raw band scores and the final (m, l): no exchange
    if (kz == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (this wave's own band stores)
    u32x4 pbh[2], pbl[2];
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      float pv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int rr = 16 * cc + 8 * lh + j, key = q + rr - p.win;
        const bool ok = rr <= 2 * p.win && key >= 0 && key < T && q < T;
        const float sv = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(brs, ok ? (int)(((unsigned)(h * (2 * p.win + 1) + rr) * (unsigned)T + (unsigned)q) * 4u) : (int)kOOB, 0, 16));
        pv[j] = ok ? __builtin_amdgcn_exp2f((sv - m_run) * c2) * inv : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) { unsigned hh, ll; split2(pv[2 * j], pv[2 * j + 1], hh, ll); pbh[cc][j] = hh; pbl[cc][j] = ll; }
    }
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
      for (int db = 0; db < DB; ++db) {
        const unsigned char* a = p.evt_img + ((cc * 4 + lh) * D + 32 * db + li) * 16;
        const u32x4 eh = *reinterpret_cast<const u32x4*>(a), el = *reinterpret_cast<const u32x4*>(a + 2 * D * 16);
        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, eh), __builtin_bit_cast(bf16x8, pbl[cc]), o[db], 0, 0, 0);
        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, el), __builtin_bit_cast(bf16x8, pbh[cc]), o[db], 0, 0, 0);
        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, eh), __builtin_bit_cast(bf16x8, pbh[cc]), o[db], 0, 0, 0);
      }
  }
#pragma unroll
  for (int db = 0; db < DB; ++db) {
    if (p.out != nullptr && q < T) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int d = 32 * db + (r & 3) + 8 * (r >> 2) + 4 * lh;
        p.out[(long long)(h * D + d) * p.ldo + q] = o[db][r];
      }
    }
    if (p.img != nullptr) {
#pragma unroll
      for (int g2 = 0; g2 < 2; ++g2) {
        unsigned hA[2], lA[2], hB[2], lB[2];
#pragma unroll
        for (int e2 = 0; e2 < 2; ++e2) {
          split2(o[db][8 * g2 + 2 * e2], o[db][8 * g2 + 2 * e2 + 1], hA[e2], lA[e2]);
          split2(o[db][8 * g2 + 4 + 2 * e2], o[db][8 * g2 + 5 + 2 * e2], hB[e2], lB[e2]);
        }
        u32x4 hi, lo;
#pragma unroll
        for (int e2 = 0; e2 < 2; ++e2) {
          typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
          const u32x2_t sh = __builtin_amdgcn_permlane32_swap(hA[e2], hB[e2], false, false);
          const u32x2_t sl2 = __builtin_amdgcn_permlane32_swap(lA[e2], lB[e2], false, false);
          hi[e2] = sh.x; hi[2 + e2] = sh.y; lo[e2] = sl2.x; lo[2 + e2] = sl2.y;
        }
        if (q < T) {
          const long long chunk = (h * D + 32 * db) / 16 + g2;
          unsigned char* row = p.img + ((chunk * 4 + lh) * p.img_tp + p.margin + q) * 16;
          *reinterpret_cast<u32x4*>(row) = hi;
          *reinterpret_cast<u32x4*>(row + p.img_tp * 32) = lo;
        }
      }
    }
  }
  const unsigned long long dt5 = DTICK();
  DTACC(5, dt5 - dt4); DTACC(6, dt5 - dt0); DTACC(0, 1);
}

template <int D, int NWQ, int KS, bool REL>
static void launch_att_dma(const AttnDmaArgs& a, int heads, hipStream_t s) {
  auto kern = attention_dma_kernel<D, NWQ, KS, REL>;
  constexpr size_t lds = (size_t)KS * 2 * (size_t)((D / 16) * 4 * 1024 + 16 * D * 16) + (REL ? (size_t)NWQ * KS * 4096 : 0);
  static_assert(lds <= 160 * 1024, "LDS");
  RVC_ALLOW_BIG_LDS(kern);
  AttnDmaArgs b = a;
  b.nqt = (a.T + 32 * NWQ - 1) / (32 * NWQ); b.heads = heads;
  static const int xcd_env = exp_int("RVC_X3_XCD", 1);
  b.xcd_remap = xcd_env;
  hipLaunchKernelGGL(kern, dim3((unsigned)(b.nqt * heads * a.kz)), dim3(64 * NWQ * KS), lds, s, b);
}

}  // namespace rvc
