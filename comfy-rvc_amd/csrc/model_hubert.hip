// HubertModelWithFinalProj.extract_features as a HIP kernel graph (reference lib/infer_pack/loaders.py:55-61 over
// transformers/models/hubert/modeling_hubert.py:45-477,878-955): 7-layer conv feature encoder (GroupNorm on layer 0),
// LayerNorm + projection, grouped k=128 positional conv, post-LN transformer layers.  Channel-major [C][T] throughout.
#include "model_common.h"
#include "models.h"

namespace rvc {

struct HubLayer {
  ConvLayer qk;        // fused q (pre-scaled by head_dim^-0.5) and k projections: 768 -> 1536
  DevVec bv;           // v projection bias, added after P.V (softmax rows sum to 1); the v rows are part of the qk layer (768 -> 2304)
  ConvLayer o, ff1, ff2;
  DevVec g1, b1, g2, b2;
};

struct Hubert {
  Ctx* ctx = nullptr;
  Arena arena;
  TensorStore ts;
  bool ready = false;
  const void* img_base = nullptr; unsigned img_gen = 0; size_t img_bytes = 0; int img_T = -1;   // image of the positional convolution's input: margins known to be zero
  ConvLayer conv[7];
  DevVec gn_g, gn_b, fp_g, fp_b, enc_g, enc_b;
  DevVec w0;            // conv_layers.0 raw [512][10]: the fused conv0 + GroupNorm + GELU kernels evaluate it from the audio
  ConvLayer proj, pos, final_proj;
  std::vector<HubLayer> layers;
};

static const int kKern[7] = {10, 3, 3, 3, 3, 2, 2};
static const int kStride[7] = {5, 2, 2, 2, 2, 2, 2};

long long hubert_num_frames(long long L) {
  long long t = L;
  for (int i = 0; i < 7; ++i) { if (t < kKern[i]) return 0; t = (t - kKern[i]) / kStride[i] + 1; }
  return t;
}

Hubert* hubert_create(Ctx* ctx) { Hubert* H = new Hubert(); H->ctx = ctx; return H; }
void hubert_set_tensor(Hubert* H, const char* name, const float* d, const long long* shape, int ndim) { H->ts.set(name, d, shape, ndim); }

static void hubert_free(Hubert& H) {
  for (auto& c : H.conv) conv_layer_free(c);
  H.w0.free_(); H.gn_g.free_(); H.gn_b.free_(); H.fp_g.free_(); H.fp_b.free_(); H.enc_g.free_(); H.enc_b.free_();
  conv_layer_free(H.proj); conv_layer_free(H.pos); conv_layer_free(H.final_proj);
  for (auto& l : H.layers) { conv_layer_free(l.qk); l.bv.free_(); conv_layer_free(l.o); conv_layer_free(l.ff1); conv_layer_free(l.ff2); l.g1.free_(); l.b1.free_(); l.g2.free_(); l.b2.free_(); }
  H.layers.clear();
  H.img_base = nullptr; H.img_gen = 0; H.img_bytes = 0; H.img_T = -1;
}
void hubert_destroy(Hubert* H) { if (H) { hubert_free(*H); H->arena.release(); delete H; } }

static void linear_layer(ConvLayer& L, const TensorStore& ts, const std::string& p, int out, int in) {
  conv1d_layer_init(L, ts.get(p + ".weight", {out, in}).data.data(), ts.get(p + ".bias", {out}).data.data(), out, in, 1, 1, 0, 1, 1);
}

void hubert_finalize(Hubert* H) {
  const TensorStore& ts = H->ts;
  hubert_free(*H);
  // the k = 1 projections (feature projection, q/k, out, feed-forward, final_proj) also get a bf16x3 split weight image
  // (conv_x3.hip); strided / grouped convolutions and the activation x activation products of attention stay on the fp32 kernel
  ConvBuildScope x3scope(H->ctx->precision);
  // conv0: Conv1d(1, 512, 10, stride 5) == Linear(10 -> 512) over im2col frames
  conv1d_layer_init(H->conv[0], ts.get("feature_extractor.conv_layers.0.conv.weight", {512, 1, 10}).data.data(), nullptr, 512, 10, 1, 1, 0, 1, 1);
  H->w0.upload(ts.get("feature_extractor.conv_layers.0.conv.weight", {512, 1, 10}).data);
  for (int i = 1; i < 7; ++i)
    conv1d_layer_init(H->conv[i], ts.get("feature_extractor.conv_layers." + std::to_string(i) + ".conv.weight", {512, 512, kKern[i]}).data.data(),
                      nullptr, 512, 512, kKern[i], kStride[i], 0, 1, 1);
  H->gn_g.upload(ts.get("feature_extractor.conv_layers.0.layer_norm.weight", {512}).data);
  H->gn_b.upload(ts.get("feature_extractor.conv_layers.0.layer_norm.bias", {512}).data);
  H->fp_g.upload(ts.get("feature_projection.layer_norm.weight", {512}).data);
  H->fp_b.upload(ts.get("feature_projection.layer_norm.bias", {512}).data);
  linear_layer(H->proj, ts, "feature_projection.projection", 768, 512);
  {
    // weight_norm(dim=2): w[:, :, k] = v[:, :, k] * g[k] / ||v[:, :, k]||   (modeling_hubert.py:61-77); both key spellings accepted
    const bool par = ts.has("encoder.pos_conv_embed.conv.parametrizations.weight.original1");
    const HostTensor& v = ts.get(par ? "encoder.pos_conv_embed.conv.parametrizations.weight.original1" : "encoder.pos_conv_embed.conv.weight_v", {768, 48, 128});
    const HostTensor& g = ts.get(par ? "encoder.pos_conv_embed.conv.parametrizations.weight.original0" : "encoder.pos_conv_embed.conv.weight_g", {1, 1, 128});
    std::vector<float> w(v.numel());
    for (int k = 0; k < 128; ++k) {
      double s = 0.0;
      for (size_t r = 0; r < (size_t)768 * 48; ++r) { const double x = v.data[r * 128 + k]; s += x * x; }
      const float sc = (float)((double)g.data[k] / std::sqrt(s));
      for (size_t r = 0; r < (size_t)768 * 48; ++r) w[r * 128 + k] = v.data[r * 128 + k] * sc;
    }
    conv1d_layer_init(H->pos, w.data(), ts.get("encoder.pos_conv_embed.conv.bias", {768}).data.data(), 768, 768, 128, 1, 64, 1, 16);
  }
  H->enc_g.upload(ts.get("encoder.layer_norm.weight", {768}).data);
  H->enc_b.upload(ts.get("encoder.layer_norm.bias", {768}).data);
  int nl = 0;
  while (ts.has("encoder.layers." + std::to_string(nl) + ".attention.q_proj.weight")) ++nl;
  RVC_REQUIRE(nl >= 11, "expected at least 11 encoder layers");
  H->layers.resize(nl);
  const float qs = 0.125f;   // head_dim^-0.5 = 64^-0.5, exact in fp32
  for (int l = 0; l < nl; ++l) {
    HubLayer& Y = H->layers[l];
    const std::string p = "encoder.layers." + std::to_string(l) + ".";
    const HostTensor& wq = ts.get(p + "attention.q_proj.weight", {768, 768});
    const HostTensor& wk = ts.get(p + "attention.k_proj.weight", {768, 768});
    const HostTensor& bq = ts.get(p + "attention.q_proj.bias", {768});
    const HostTensor& bk = ts.get(p + "attention.k_proj.bias", {768});
    // one 768 -> 2304 projection: q (scaled), k, v rows; v's bias is added after the attention (softmax rows sum to 1)
    const HostTensor& wv = ts.get(p + "attention.v_proj.weight", {768, 768});
    std::vector<float> w(3 * (size_t)768 * 768), b(3 * 768, 0.f);
    for (size_t i = 0; i < (size_t)768 * 768; ++i) { w[i] = wq.data[i] * qs; w[(size_t)768 * 768 + i] = wk.data[i]; w[2 * (size_t)768 * 768 + i] = wv.data[i]; }
    for (int i = 0; i < 768; ++i) { b[i] = bq.data[i] * qs; b[768 + i] = bk.data[i]; }
    conv1d_layer_init(Y.qk, w.data(), b.data(), 2304, 768, 1, 1, 0, 1, 1);
    Y.bv.upload(ts.get(p + "attention.v_proj.bias", {768}).data);
    linear_layer(Y.o, ts, p + "attention.out_proj", 768, 768);
    linear_layer(Y.ff1, ts, p + "feed_forward.intermediate_dense", 3072, 768);
    linear_layer(Y.ff2, ts, p + "feed_forward.output_dense", 768, 3072);
    Y.g1.upload(ts.get(p + "layer_norm.weight", {768}).data); Y.b1.upload(ts.get(p + "layer_norm.bias", {768}).data);
    Y.g2.upload(ts.get(p + "final_layer_norm.weight", {768}).data); Y.b2.upload(ts.get(p + "final_layer_norm.bias", {768}).data);
  }
  linear_layer(H->final_proj, ts, "final_proj", 256, 768);
  H->ts.clear();
  H->ready = true;
}

static void hubert_graph(Hubert* H, hipStream_t s, Arena& A, const float* audio, long long L, int version, int n_layers, float* out_rm,
                         float* out_cm, const HubertTaps* taps) {
  const bool dry = A.dry;
  auto tap = [&](float* dst, const float* src, size_t n) {
    if (!dry && dst) RVC_HIP_CHECK(hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, s));
  };
  ConvEpilogue E0;
  int Tc[8]; Tc[0] = (int)L;
  for (int i = 0; i < 7; ++i) Tc[i + 1] = (Tc[i] - kKern[i]) / kStride[i] + 1;
  const int T = Tc[7];
  // the positional convolution (k = 128, 16 groups) reads its input as a split-resident image through row offsets -64 .. +63: the margins
  // of that image are its zero padding, so it lives in a block of its own (first allocation, never aliased) that is zeroed per (layout, T)
  static const bool x3s_on0 = (exp_int("RVC_X3S", 1) != 0);
  static const bool pos_on = (exp_int("RVC_X3S_POS", 1) != 0);
  const bool pos_s = x3s_on0 && pos_on && conv_x3_enabled() && conv_x3s_eligible(H->pos) && conv_x3s_eligible(H->proj);
  unsigned char* hpos_s = nullptr;
  if (pos_s) {
    const size_t img0 = A.off;
    hpos_s = A.alloc<unsigned char>(split_image_bytes(768, T));
    const size_t ib = A.off - img0;
    if (!dry && (H->img_base != A.base + img0 || H->img_gen != A.gen || H->img_bytes != ib || H->img_T != T)) {
      RVC_HIP_CHECK(hipMemsetAsync(A.base + img0, 0, ib, s));
      H->img_base = A.base + img0; H->img_gen = A.gen; H->img_bytes = ib; H->img_T = T;
    }
  }
  // ---- feature encoder
  static const bool fuse0 = (exp_int("RVC_HUBERT_FUSE0", 1) != 0);
  // Layers 1 .. 6 (k = 3 / 2, stride 2, no padding) on the split-resident GEMM: every layer's output is written by its producer's epilogue as the bf16 hi / lo
  // image of the next one, DE-INTERLEAVED (even | odd positions), so that a stride-2 tap is a row offset and no layer converts its input per tile; the exact-erf
  // GELU sits in the epilogue (the pipelined kernel ran it as a second pass over the tensor).  RVC_HUBERT_S2=0: the fp32 path on conv_x3p_kernel / conv_x3_kernel.
  static const bool s2_on = (exp_int("RVC_HUBERT_S2", 1) != 0);
  bool s2 = s2_on && fuse0 && conv_x3_enabled();
  for (int i = 1; i < 7 && s2; ++i) s2 = conv_x3s_s2_eligible(H->conv[i]) && Tc[i + 1] >= 1;
  float* feat = nullptr;
  if (s2) {
    unsigned char* imgA = A.alloc<unsigned char>(split_s2_bytes(512, Tc[1]));
    unsigned char* imgB = A.alloc<unsigned char>(split_s2_bytes(512, Tc[2]));
    double* c0part = A.alloc<double>(hubert_conv0_scratch_doubles(512, Tc[1]));
    float* c0stat = A.alloc<float>(1024);
    feat = A.alloc<float>((size_t)512 * Tc[7]);
    if (!dry) {
      hubert_conv0_gn_gelu_img(s, audio, L, H->w0.p, H->gn_g.p, H->gn_b.p, 512, Tc[1], 1e-5f, imgA, split_s2_tp(Tc[1]), kSplitMargin, split_s2_h(Tc[1]), c0part, c0stat);
      unsigned char* in_s = imgA; unsigned char* out_s = imgB;
      for (int i = 1; i < 7; ++i) {
        const SplitGeom g = split_geom_s2(kKern[i], Tc[i]);
        ConvEpilogue Eg; Eg.act = ACT_GELU;
        if (i < 6) { Eg.ys_out = out_s; Eg.ys_tp = split_s2_tp(Tc[i + 1]); Eg.ys_deint_h = split_s2_h(Tc[i + 1]); }
        conv_x3s_run(H->conv[i], s, in_s, split_s2_tp(Tc[i]), Tc[i + 1], i < 6 ? nullptr : feat, Tc[i + 1], Eg, &g);
        std::swap(in_s, out_s);      // ping-pong inside the two largest images (each layer's output is half its input)
      }
    }
  } else {
  float* fr = fuse0 ? nullptr : A.alloc<float>((size_t)10 * Tc[1]);
  float* c0 = A.alloc<float>((size_t)512 * Tc[1]);
  float* c1 = A.alloc<float>((size_t)512 * Tc[2]);
  double* c0part = fuse0 ? A.alloc<double>(hubert_conv0_scratch_doubles(512, Tc[1])) : nullptr;
  float* c0stat = fuse0 ? A.alloc<float>(1024) : nullptr;
  if (!dry) {
    if (fuse0) {
      hubert_conv0_gn_gelu(s, audio, L, H->w0.p, H->gn_g.p, H->gn_b.p, 512, Tc[1], 1e-5f, c0, Tc[1], c0part, c0stat);
    } else {
      frames(s, audio, fr, (int)L, 10, 5, 0, Tc[1], 0);
      conv1d_run(H->conv[0], s, fr, Tc[1], Tc[1], c0, Tc[1], E0);
      groupnorm_t_gelu(s, c0, H->gn_g.p, H->gn_b.p, 512, Tc[1], Tc[1], 1e-5f);
    }
  }
  float* in = c0; float* outb = c1;
  for (int i = 1; i < 7; ++i) {
    if (!dry) { ConvEpilogue Eg; Eg.act = ACT_GELU; conv1d_run(H->conv[i], s, in, Tc[i], Tc[i], outb, Tc[i + 1], Eg); }
    std::swap(in, outb);      // ping-pong inside the two largest buffers (each layer's output is smaller than its input)
  }
  feat = in;
  }
  // feat: [512][T]
  if (taps) tap(taps->conv_stack, feat, (size_t)512 * T);
  // Split-resident GEMM path (conv_x3s.hip): the activations that feed a k = 1 projection live as the bf16 hi / lo image the kernel stages,
  // written by their producers (LayerNorm, the attention's epilogue, FFN1's GELU epilogue); the fp32 copy is kept only where a residual or
  // the attention reads it.  Needs the bf16x3 weight images (context precision 1 / 2); RVC_X3S=0 selects the fp32-input kernels.
  static const bool x3s_on = (exp_int("RVC_X3S", 1) != 0);
  int need = version == 1 ? 8 : 11;
  if (n_layers > 0) need = n_layers;
  RVC_REQUIRE(need <= (int)H->layers.size(), "not enough encoder layers loaded");
  bool gs = x3s_on && conv_x3_enabled() && conv_x3s_eligible(H->proj) && (version != 1 || conv_x3s_eligible(H->final_proj));
  for (int l = 0; l < need && gs; ++l) {
    const HubLayer& Y = H->layers[l];
    gs = conv_x3s_eligible(Y.qk) && conv_x3s_eligible(Y.o) && conv_x3s_eligible(Y.ff1) && conv_x3s_eligible(Y.ff2);
  }
  const long long tp = split_image_tp(T);
  float* ln = gs ? nullptr : A.alloc<float>((size_t)512 * T);
  unsigned char* ln_s = gs ? A.alloc<unsigned char>(split_image_bytes(512, T)) : nullptr;
  unsigned char* hs = gs ? A.alloc<unsigned char>(split_image_bytes(768, T)) : nullptr;
  float* h = A.alloc<float>((size_t)768 * T);
  float* hb = A.alloc<float>((size_t)768 * T);
  if (!dry) {
    if (gs) {
      layernorm_c_split(s, feat, H->fp_g.p, H->fp_b.p, nullptr, ln_s, tp, kSplitMargin, 512, T, T, 1e-5f);
      ConvEpilogue Epj; if (pos_s) { Epj.ys_out = hpos_s; Epj.ys_tp = tp; }
      conv_x3s_run(H->proj, s, ln_s, tp, T, h, T, Epj);
    } else {
      layernorm_c(s, feat, nullptr, H->fp_g.p, H->fp_b.p, ln, 512, T, T, 1e-5f);
      conv1d_run(H->proj, s, ln, T, T, h, T, E0);
    }
    if (taps && taps->pos_conv) { ConvEpilogue Ep; Ep.act = ACT_GELU; Ep.tout_limit = T; conv1d_run(H->pos, s, h, T, T, taps->pos_conv, T, Ep); }
    ConvEpilogue Ep; Ep.act = ACT_GELU; Ep.act_before_res = 1; Ep.R = h; Ep.ldR = T; Ep.tout_limit = T;
    if (gs && pos_s) { Ep.tout_limit = 0; conv_x3s_run(H->pos, s, hpos_s, tp, T, hb, T, Ep); }      // 16 groups x one 64-row tile, 384 (chunk, tap) units
    else conv1d_run(H->pos, s, h, T, T, hb, T, Ep);
    if (gs) layernorm_c_split(s, hb, H->enc_g.p, H->enc_b.p, h, hs, tp, kSplitMargin, 768, T, T, 1e-5f);
    else layernorm_c(s, hb, nullptr, H->enc_g.p, H->enc_b.p, h, 768, T, T, 1e-5f);
  }
  {
    const size_t mark = A.off;
    // attention on split-resident operands (attention_dma.hip): q / k as one image, V^T by the swapped product; RVC_ATT_DMA=0: fp32 q / k / v
    static const bool att_dma = (exp_int("RVC_ATT_DMA", 1) != 0);
    const bool ad = gs && att_dma;
    float* qk = ad ? nullptr : A.alloc<float>((size_t)2304 * T);
    float* vr = ad ? nullptr : A.alloc<float>((size_t)T * 768);
    unsigned char* qk_s = ad ? A.alloc<unsigned char>(split_image_bytes(1536, T)) : nullptr;
    unsigned char* vt_s = ad ? A.alloc<unsigned char>(attention_vt_bytes(768, T)) : nullptr;
    float* attn = gs ? nullptr : A.alloc<float>((size_t)768 * T);
    float* ff = gs ? nullptr : A.alloc<float>((size_t)3072 * T);
    unsigned char* attn_s = gs ? A.alloc<unsigned char>(split_image_bytes(768, T)) : nullptr;
    unsigned char* ff_s = gs ? A.alloc<unsigned char>(split_image_bytes(3072, T)) : nullptr;
    if (!dry) {
      if (ad) attention_vt_clear_tail(s, vt_s, 768, T);
      for (int l = 0; l < need; ++l) {
        HubLayer& Y = H->layers[l];
        if (taps && l == 0) tap(taps->hidden_0, h, (size_t)768 * T);
        if (taps && l == 8) tap(taps->hidden_8, h, (size_t)768 * T);
        ConvEpilogue Er; Er.R = h; Er.ldR = T;
        if (gs) {
          if (ad) {
            static const bool qkv1 = (exp_int("RVC_QKV_FUSED", 1) != 0);
            if (qkv1) {
              // q | k | v in ONE launch: the q and k rows go to their image, the v rows through the transposing epilogue into the V^T image (v's bias after the attention)
              ConvEpilogue Eqk; Eqk.ys_out = qk_s; Eqk.ys_tp = tp; Eqk.vt_out = vt_s; Eqk.vt_tp = attention_vt_tp(768); Eqk.vt_row0 = 1536;
              conv_x3s_run(Y.qk, s, hs, tp, T, nullptr, T, Eqk);
            } else {
              ConvLayer qkL = Y.qk; qkL.Co = 1536;                               // the q and k rows of the 2304-row projection -> image only
              ConvEpilogue Eqk; Eqk.ys_out = qk_s; Eqk.ys_tp = tp;
              conv_x3s_run(qkL, s, hs, tp, T, nullptr, T, Eqk);
              conv_x3s_run_swapped(Y.qk, 1536, 768, s, hs, tp, T, vt_s, attention_vt_tp(768));      // V^T image by the swapped product
            }
            attention_split(s, qk_s, tp, 1536, 0, 48, vt_s, 12, 64, T, 1.f, Y.bv.p, nullptr, T, attn_s, tp);
          } else {
            conv_x3s_run(Y.qk, s, hs, tp, T, qk, T, E0);
            transpose(s, qk + (size_t)1536 * T, vr, 768, T, T, 768, 1, 0, 0);      // V row-major [T][768] for the fused attention
            attention_fused(s, qk, qk + (size_t)768 * T, T, vr, 768, Y.bv.p, nullptr, T, 12, 64, T, attn_s, tp);
          }
          conv_x3s_run(Y.o, s, attn_s, tp, T, hb, T, Er);
          layernorm_c_split(s, hb, Y.g1.p, Y.b1.p, h, hs, tp, kSplitMargin, 768, T, T, 1e-5f);
          ConvEpilogue Eg; Eg.act = ACT_GELU; Eg.ys_out = ff_s; Eg.ys_tp = tp;
          conv_x3s_run(Y.ff1, s, hs, tp, T, nullptr, T, Eg);                     // GELU in the epilogue, the 3072-channel tensor exists only as the image
          conv_x3s_run(Y.ff2, s, ff_s, tp, T, hb, T, Er);
          layernorm_c_split(s, hb, Y.g2.p, Y.b2.p, h, hs, tp, kSplitMargin, 768, T, T, 1e-5f);
          continue;
        }
        conv1d_run(Y.qk, s, h, T, T, qk, T, E0);
        transpose(s, qk + (size_t)1536 * T, vr, 768, T, T, 768, 1, 0, 0);      // V row-major [T][768] for the fused attention
        // softmax(K^T Q) V + bv without materialising the [12][T][T] scores (attention.hip)
        attention_fused(s, qk, qk + (size_t)768 * T, T, vr, 768, Y.bv.p, attn, T, 12, 64, T);
        conv1d_run(Y.o, s, attn, T, T, hb, T, Er);
        layernorm_c(s, hb, nullptr, Y.g1.p, Y.b1.p, h, 768, T, T, 1e-5f);
        ConvEpilogue Eg; Eg.act = ACT_GELU;
        conv1d_run(Y.ff1, s, h, T, T, ff, T, Eg);
        conv1d_run(Y.ff2, s, ff, T, T, hb, T, Er);
        layernorm_c(s, hb, nullptr, Y.g2.p, Y.b2.p, h, 768, T, T, 1e-5f);
      }
      if (taps && need == 8) tap(taps->hidden_8, h, (size_t)768 * T);
    }
    A.off = mark;
  }
  const float* res = h; int D = 768;
  if (version == 1) {
    float* fp = A.alloc<float>((size_t)256 * T);
    if (!dry) { if (gs) conv_x3s_run(H->final_proj, s, hs, tp, T, fp, T, E0); else conv1d_run(H->final_proj, s, h, T, T, fp, T, E0); }
    res = fp; D = 256;
  }
  if (!dry) {
    if (out_cm) RVC_HIP_CHECK(hipMemcpyAsync(out_cm, res, (size_t)D * T * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (out_rm) transpose(s, res, out_rm, D, T, T, D, 1, 0, 0);
  }
}

void hubert_forward(Hubert* H, hipStream_t s, const float* audio, long long L, int version, int n_layers, float* out_rm, float* out_cm,
                    const HubertTaps* taps) {
  RVC_REQUIRE(H->ready, "hubert_finalize has not been called");
  RVC_REQUIRE(version == 1 || version == 2, "version must be 1 or 2");
  RVC_REQUIRE(hubert_num_frames(L) >= 2, "audio too short");
  Arena& A = H->arena;
  for (int pass = 0; pass < 2; ++pass) {
    A.dry = (pass == 0); A.reset(); if (pass == 0) A.peak = 0;
    hubert_graph(H, s, A, audio, L, version, n_layers, out_rm, out_cm, taps);
    if (pass == 0) A.ensure(A.peak);
  }
  A.dry = false;
}

size_t hubert_workspace(const Hubert* M) { return M->arena.cap; }

}  // namespace rvc
