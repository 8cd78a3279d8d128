// Self-attention on SPLIT-RESIDENT operands (gfx950 only): softmax(K^T Q) V per head for HuBERT's encoder layers (reference
// modeling_hubert.py:291-477 through lib/infer_pack/loaders.py:55-61), bf16x3 products, online softmax, nothing materialised.
//
// attention_x3_kernel (attention.hip) reads Q / K / V as fp32, converts every 64-key tile to bf16 hi / lo through registers in every
// workgroup that needs it, exchanges the row maxima and sums of a tile between waves through LDS and stages the probabilities in LDS:
// four barriers per tile, 9.5 k cycles per tile, the matrix pipe 8 % busy (profiles/r3m_kernel_stats: 113 us per layer).  Here the
// three operands already ARE the images the MFMAs read:
//   * Q and K: the [16-channel chunk][hi | lo][8-channel half][margin + t][8 ch] image the q / k projection's epilogue writes
//     (conv_x3s.hip) - a head's d = 64 channels are 4 chunks, a key tile is 16 contiguous 1-KiB pieces, copied to LDS by DMA;
//   * V^T: the image of the transposed tensor, [16-key chunk][hi | lo][8-key half][margin + channel][8 keys], written by the swapped
//     product of the same GEMM kernel (conv_x3s_run_swapped) - the reduction of P V runs over keys, so keys are the 16-byte rows;
//   * a wave owns 32 queries and ALL 64 keys of a tile: the row maximum and sum of the online softmax are in-lane reductions plus one
//     exchange with the lane 32 away - no LDS, no barrier; the probabilities never leave registers: the v_permlane32_swap that turns
//     accumulator quads into 16-byte image rows (the GEMM's image epilogue) leaves lane (query i, half h) holding exactly the
//     8 keys x {hi, lo} the B operand of the P V MFMA wants from lane (i, h);
//   * K / V^T tiles are double-buffered: tile t + 1 is requested right after the barrier that publishes tile t - one barrier per tile.
#include "attention_dma_kernel.h"

namespace rvc {

#ifdef RVC_CONV_TIMING
void attention_dma_rel_timing_read(unsigned long long* out8, bool reset);      // attention_dma_rel.hip
void attention_dma_timing_read(unsigned long long* out8, bool reset) {
  unsigned long long r[8];
  attd_timing_read_tu(out8, reset); attention_dma_rel_timing_read(r, reset);
  for (int i = 0; i < 8; ++i) out8[i] += r[i];
}
#endif

void attention_dma_rel_launch(const AttnDmaArgs& a, int heads, hipStream_t s);      // attention_dma_rel.hip

// rows of the V^T image per plane and its size for T keys (4 key chunks per 64-key tile, whole tiles)
long long attention_vt_tp(int channels) { return ((long long)kSplitMargin + channels + 63) & ~63LL; }
size_t attention_vt_bytes(int channels, int T) { return (size_t)((T + 63) / 64) * 4 * 4 * (size_t)attention_vt_tp(channels) * 16; }

// key chunks that lie wholly past T inside the last 64-key tile are never written by the producer: zero them (P = 0 there, 0 x garbage must be 0)
void attention_vt_clear_tail(hipStream_t s, unsigned char* vt_img, int channels, int T) {
  const size_t chunk = (size_t)4 * attention_vt_tp(channels) * 16;
  const int first = (T + 15) / 16, last = (T + 63) / 64 * 4;
  if (last > first) RVC_HIP_CHECK(hipMemsetAsync(vt_img + (size_t)first * chunk, 0, (size_t)(last - first) * chunk, s));
}

// E_k / E_v [2 win + 1][D] (host fp32) -> the two operand images of the REL terms (see AttnDmaArgs)
void attention_rel_images(const float* ek, const float* ev, int D, int win, std::vector<uint16_t>& ek_img, std::vector<uint16_t>& evt_img) {
  const int R = 2 * win + 1;
  RVC_REQUIRE(R <= 32 && (D & 31) == 0, "attention_rel_images: window must fit 32 rows");
  auto bf = [](float v) { uint32_t u; memcpy(&u, &v, 4); const uint32_t r = u + 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(r >> 16); };
  auto fl = [](uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; };
  ek_img.assign((size_t)(D / 16) * 4 * 32 * 8, 0); evt_img.assign((size_t)2 * 4 * D * 8, 0);
  for (int r = 0; r < R; ++r)
    for (int d = 0; d < D; ++d) {
      { const float v = ek[r * D + d]; const uint16_t hi = bf(v), lo = bf(v - fl(hi));
        const int c = d / 16, half = (d % 16) / 8, j = d % 8;
        ek_img[(((size_t)c * 4 + half) * 32 + r) * 8 + j] = hi; ek_img[(((size_t)c * 4 + 2 + half) * 32 + r) * 8 + j] = lo; }
      { const float v = ev[r * D + d]; const uint16_t hi = bf(v), lo = bf(v - fl(hi));
        const int c = r / 16, half = (r % 16) / 8, j = r % 8;
        evt_img[(((size_t)c * 4 + half) * D + d) * 8 + j] = hi; evt_img[(((size_t)c * 4 + 2 + half) * D + d) * 8 + j] = lo; }
    }
}

static thread_local int t_force_kz = 0;
void attention_split_force_kz(int kz) { t_force_kz = kz; }      // tests: key slices of the calling thread's next launches (0 = automatic)

void attention_split(hipStream_t s, const unsigned char* qk_img, long long qk_tp, int qk_channels, int q_chunk0, int k_chunk0, const unsigned char* vt_img,
                     int heads, int dhead, int T, float scale, const float* bv, float* out, long long ldo, unsigned char* out_img, long long img_tp,
                     int win, const unsigned char* ek_img, const unsigned char* evt_img) {
  const bool rel = ek_img != nullptr;
  RVC_REQUIRE((dhead == 64 && !rel) || (dhead == 96 && rel && win == 10 && evt_img && scale == 1.f), "attention_split: head dimension 64, or 96 with the window-10 relative-position tables");
  RVC_REQUIRE(out != nullptr || out_img != nullptr, "attention_split: no output");
  RVC_REQUIRE(qk_tp >= kSplitMargin + T + 704, "attention_split: q / k image too short");
  AttnDmaArgs a{};
  a.QK = qk_img; a.qkTp = qk_tp; a.q_chunk0 = q_chunk0; a.k_chunk0 = k_chunk0;
  const double qb = (double)(qk_channels / 16) * 4.0 * (double)qk_tp * 16.0;
  a.Vt = vt_img; a.vtTp = attention_vt_tp(heads * dhead);
  const double vb = (double)attention_vt_bytes(heads * dhead, T);
  RVC_REQUIRE(qb < 2147483648.0 && vb < 2147483648.0, "attention_split: image exceeds 32-bit buffer addressing");
  a.qk_bytes = (unsigned)qb; a.vt_bytes = (unsigned)vb;
  a.margin = kSplitMargin; a.T = T; a.scale = scale; a.bv = bv; a.out = out; a.ldo = ldo; a.img = out_img; a.img_tp = img_tp;
  a.win = win; a.ek_img = ek_img; a.evt_img = evt_img;
  static const int nwq = exp_int("RVC_ATT_NWQ", 4);
  static const int ks = exp_int("RVC_ATT_KS", 2);
  static const int kz_env = exp_int("RVC_ATT_KZ", 0);
  a.kz = 1;
  if (rel) {
    // two heads: the key tiles are also cut across workgroups until the grid covers the chip (each slice at least two tiles)
    const int qt = (T + 127) / 128, ntiles = (T + 63) / 64;
    int kz = t_force_kz > 0 ? t_force_kz : kz_env > 0 ? kz_env : 256 / (qt * heads);      // (measured at T = 3198: 5 slices 49 us, 4: 53, 6: 67 - a second round of workgroups)
    kz = std::max(1, std::min(kz, ntiles / 2));
    a.kz = kz;
    const size_t sb = (size_t)heads * (2 * win + 1) * T * 4;
    RVC_REQUIRE(sb < 2147483648.0, "attention_split: band scratch exceeds 32-bit buffer addressing");
    a.sband = (float*)stream_scratch(s, 11, sb); a.sband_bytes = (unsigned)sb;
    if (kz > 1) {
      const size_t pbytes = (size_t)qt * heads * kz * 4 * (4 * 3 + 1) * 64 * 16;
      RVC_REQUIRE((size_t)qt * heads * 4 <= 16 * 1024 && pbytes < 2147483648.0, "attention_split: too many query tiles for the ticket array");
      a.tickets = (unsigned*)stream_scratch_zeroed(s, 9, 16 * 1024);
      a.part = (float*)stream_scratch(s, 10, pbytes); a.part_bytes = (unsigned)pbytes;
    }
    attention_dma_rel_launch(a, heads, s);
    return;
  }
  if (nwq == 2) { if (ks == 1) launch_att_dma<64, 2, 1, false>(a, heads, s); else launch_att_dma<64, 2, 2, false>(a, heads, s); }
  else { if (ks == 1) launch_att_dma<64, 4, 1, false>(a, heads, s); else launch_att_dma<64, 4, 2, false>(a, heads, s); }
}

}  // namespace rvc
