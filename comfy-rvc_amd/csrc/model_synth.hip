// SynthesizerTrnMs{256,768}NSFsid.infer as a HIP kernel graph (reference lib/infer_pack/models.py:682-693,:798-809):
// enc_p (TextEncoder + 6 relative-position attention layers) -> z_p sampling -> 4 reversed coupling layers (WN)
// -> GeneratorNSF (SineGen source, ConvTranspose1d stages, 3x ResBlock1 per stage, conv_post, tanh).
// Activations are channel-major [C][T]; weights arrive with the reference's state-dict names and are folded here.
#include "model_common.h"
#include "models.h"
#include "conv_kernels.h"

namespace rvc {

static ConvLayer make_conv1d(const TensorStore& ts, const std::string& p, int stride, int pad, int dil, bool wn, bool bias = true,
                             float wscale = 1.f, int row0 = 0, int rows = -1) {
  std::vector<float> w; std::vector<long long> shape;
  if (wn) { const HostTensor& v = ts.get(p + ".weight_v"); w = weight_norm0(v, ts.get(p + ".weight_g")); shape = v.shape; }
  else { const HostTensor& v = ts.get(p + ".weight"); w = v.data; shape = v.shape; }
  RVC_REQUIRE(shape.size() == 3, p + ": expected a [Co][Ci][k] weight");
  int Co = (int)shape[0], Ci = (int)shape[1], k = (int)shape[2];
  std::vector<float> b;
  if (bias) b = ts.get(p + ".bias").data;
  if (rows < 0) rows = Co;
  const size_t per = (size_t)Ci * k;
  std::vector<float> ws(w.begin() + row0 * per, w.begin() + (size_t)(row0 + rows) * per);
  if (wscale != 1.f) for (auto& x : ws) x *= wscale;
  std::vector<float> bs;
  if (bias) { bs.assign(b.begin() + row0, b.begin() + row0 + rows); if (wscale != 1.f) for (auto& x : bs) x *= wscale; }
  ConvLayer L;
  conv1d_layer_init(L, ws.data(), bias ? bs.data() : nullptr, rows, Ci, k, stride, pad, dil, 1);
  return L;
}

struct EncLayer {
  ConvLayer qk;            // fused conv_q (pre-scaled by 1/sqrt(kc)) and conv_k: C -> 2C
  DevVec bv;               // conv_v's bias, added after P.V; its rows are part of the qk layer (C -> 3 C)
  ConvLayer relk, relv;    // emb_rel_k as a 21-row projection of Q; emb_rel_v as a 21 -> kc projection of banded P (unfused path)
  DevVec ek, ev;           // the raw [21][kc] tables: the fused attention kernel does both projections itself
  DevVec rel_img;          // the same tables as the MFMA operand images of attention_split (kc = 96): E_k image, then E_v^T image (bf16 hi / lo)
  size_t evt_off = 0;      // byte offset of the E_v^T image
  ConvLayer o, ffn1, ffn2;
  DevVec g1, b1, g2, b2;
};
struct FlowLayer {
  ConvLayer pre, post, in[3], res[2], skip[3];
  ConvLayer in_gate[3];    // in_layers with their 2 H rows in wn_gate_row_order: the gate runs in the split-resident GEMM's epilogue (conv_x3s.hip), the 2 H-row tensor is never stored
  ConvLayer rs[2];         // res_skip_layers 0 / 1 whole (2 H rows: residual | skip) and
  ConvLayer post_neg;      // post with negated weights (x1 - m as a plain residual add): the split-resident path (conv_x3s.hip)
  DevVec cond_w, cond_b;   // weight-normed cond_layer [2*H*3][gin]
};
struct ResBlock { ConvLayer c1[3], c2[3]; };
struct GenStage { ConvLayer up, noise; DevVec noise_w, noise_b; int u = 1, k = 1, noise_k = 1, noise_s = 1; ResBlock rb[3]; };   // noise_w / _b: raw [C][k] / [C] for the streaming kernel (k <= 8)

struct Synth {
  Ctx* ctx = nullptr;
  Arena arena;           // activation workspace (grown on demand between launches)
  TensorStore ts;
  bool ready = false;
  // config
  int inter = 192, hidden = 192, filt = 768, n_heads = 2, n_layers = 6, ksz = 3, gin = 256, n_spk = 1, sr = 40000, feat_dim = 768;
  int up_init = 512; std::vector<int> rb_k, up_rates, up_k; std::vector<std::vector<int>> rb_d;
  int upp = 1;
  // weights
  DevVec emb_phone_wT, emb_phone_b, emb_pitch, emb_g;
  std::vector<EncLayer> enc;
  ConvLayer proj;
  FlowLayer flow[4];
  ConvLayer conv_pre, conv_post;
  DevVec dec_cond_w, dec_cond_b, conv_post_w;   // conv_post_w: raw [Ci][7] weights of the 1-channel output conv (ops.hip::conv_to1)
  std::vector<GenStage> stages;
  float lin_w = 1.f, lin_b = 0.f;
  const void* img_base = nullptr; unsigned img_gen = 0; size_t img_bytes = 0; int img_T = -1;   // split-resident image block whose margins are known to be zero (synth_graph)
  bool f0 = true;        // false: the *_nono family (no pitch embedding, plain Generator: reference models.py:244-311,:812-1022)
};

static void synth_free(Synth& S) {
  auto fl = [](ConvLayer& L) { conv_layer_free(L); };
  S.emb_phone_wT.free_(); S.emb_phone_b.free_(); S.emb_pitch.free_(); S.emb_g.free_();
  for (auto& e : S.enc) { fl(e.qk); e.bv.free_(); fl(e.relk); fl(e.relv); e.ek.free_(); e.ev.free_(); e.rel_img.free_(); fl(e.o); fl(e.ffn1); fl(e.ffn2); e.g1.free_(); e.b1.free_(); e.g2.free_(); e.b2.free_(); }
  S.enc.clear();
  fl(S.proj);
  for (auto& f : S.flow) { fl(f.pre); fl(f.post); for (auto& c : f.in) fl(c); for (auto& c : f.in_gate) fl(c); for (auto& c : f.res) fl(c); for (auto& c : f.skip) fl(c); for (auto& c : f.rs) fl(c); fl(f.post_neg); f.cond_w.free_(); f.cond_b.free_(); }
  fl(S.conv_pre); fl(S.conv_post); S.dec_cond_w.free_(); S.dec_cond_b.free_(); S.conv_post_w.free_();
  for (auto& st : S.stages) { fl(st.up); fl(st.noise); st.noise_w.free_(); st.noise_b.free_(); for (auto& rb : st.rb) for (int m = 0; m < 3; ++m) { fl(rb.c1[m]); fl(rb.c2[m]); } }
  S.stages.clear();
  S.img_base = nullptr; S.img_gen = 0; S.img_bytes = 0; S.img_T = -1;
}

Synth* synth_create(Ctx* ctx, const SynthConfig& c) {
  std::unique_ptr<Synth> S(new Synth());
  S->ctx = ctx;
  S->inter = c.inter_channels; S->hidden = c.hidden_channels; S->filt = c.filter_channels; S->n_heads = c.n_heads;
  S->n_layers = c.n_layers; S->ksz = c.kernel_size; S->gin = c.gin_channels; S->n_spk = c.spk_embed_dim; S->sr = c.sr;
  S->feat_dim = c.feat_dim; S->up_init = c.upsample_initial_channel;
  RVC_REQUIRE(c.n_resblock_kernels == 3, "ResBlock1 x3 expected");
  for (int i = 0; i < 3; ++i) { S->rb_k.push_back(c.resblock_kernel_sizes[i]); S->rb_d.push_back({c.resblock_dilations[i][0], c.resblock_dilations[i][1], c.resblock_dilations[i][2]}); }
  S->upp = 1;
  for (int i = 0; i < c.n_upsamples; ++i) { S->up_rates.push_back(c.upsample_rates[i]); S->up_k.push_back(c.upsample_kernel_sizes[i]); S->upp *= c.upsample_rates[i]; }
  RVC_REQUIRE(S->hidden % S->n_heads == 0, "heads must divide hidden");
  return S.release();
}
void synth_destroy(Synth* S) { if (S) { synth_free(*S); S->arena.release(); delete S; } }
void synth_set_tensor(Synth* S, const char* name, const float* d, const long long* shape, int ndim) { S->ts.set(name, d, shape, ndim); }
int synth_upp(const Synth* S) { return S->upp; }
int synth_feat_dim(const Synth* S) { return S->feat_dim; }
bool synth_has_f0(const Synth* S) { return S->f0; }

void synth_finalize(Synth* S) {
  const TensorStore& ts = S->ts;
  synth_free(*S);
  // every eligible Conv1d (stride 1, groups 1, Ci % 16 == 0) also gets a bf16x3 split weight image: the generator ResBlocks
  // (70 % of a clip's FLOPs), flow WaveNet, enc_p projections; conv_x3.hip, ~1e-5 relative error per layer
  ConvBuildScope x3scope(S->ctx->precision);
  const int C = S->hidden, kc = C / S->n_heads;
  {
    const HostTensor& w = ts.get("enc_p.emb_phone.weight", {C, S->feat_dim});
    S->emb_phone_wT.upload(transpose2d(w.data.data(), C, S->feat_dim));
    S->emb_phone_b.upload(ts.get("enc_p.emb_phone.bias", {C}).data);
    // the checkpoint decides the family, like `cpt["f0"]` does in the reference (vc_infer_pipeline.py:202-218)
    S->f0 = ts.has("enc_p.emb_pitch.weight");
    if (S->f0) S->emb_pitch.upload(ts.get("enc_p.emb_pitch.weight", {256, C}).data);
    S->emb_g.upload(ts.get("emb_g.weight").data);
    S->n_spk = (int)ts.get("emb_g.weight").shape[0];
  }
  S->enc.resize(S->n_layers);
  const float qscale = 1.f / std::sqrt((float)kc);
  for (int l = 0; l < S->n_layers; ++l) {
    EncLayer& e = S->enc[l];
    const std::string p = "enc_p.encoder.attn_layers." + std::to_string(l) + ".";
    const HostTensor& wq = ts.get(p + "conv_q.weight", {C, C, 1});
    const HostTensor& wk = ts.get(p + "conv_k.weight", {C, C, 1});
    // one C -> 3 C projection: q (scaled), k, v rows; v's bias is added after the attention
    const HostTensor& wv = ts.get(p + "conv_v.weight", {C, C, 1});
    std::vector<float> w(3 * (size_t)C * C), b(3 * (size_t)C, 0.f);
    for (size_t i = 0; i < (size_t)C * C; ++i) { w[i] = wq.data[i] * qscale; w[(size_t)C * C + i] = wk.data[i]; w[2 * (size_t)C * C + i] = wv.data[i]; }
    const HostTensor& bq = ts.get(p + "conv_q.bias", {C}); const HostTensor& bk = ts.get(p + "conv_k.bias", {C});
    for (int i = 0; i < C; ++i) { b[i] = bq.data[i] * qscale; b[C + i] = bk.data[i]; }
    conv1d_layer_init(e.qk, w.data(), b.data(), 3 * C, C, 1, 1, 0, 1, 1);
    e.bv.upload(ts.get(p + "conv_v.bias", {C}).data);
    const HostTensor& rk = ts.get(p + "emb_rel_k", {1, 21, kc});
    conv1d_layer_init(e.relk, rk.data.data(), nullptr, 21, kc, 1, 1, 0, 1, 1);            // [r][d]: out[r][q] = sum_d E_k[r][d] Q[d][q]
    const HostTensor& rv = ts.get(p + "emb_rel_v", {1, 21, kc});
    std::vector<float> rvT = transpose2d(rv.data.data(), 21, kc);                           // [d][r]: out[d][q] = sum_r E_v[r][d] Pb[r][q]
    conv1d_layer_init(e.relv, rvT.data(), nullptr, kc, 21, 1, 1, 0, 1, 1);
    e.ek.upload(rk.data); e.ev.upload(rv.data);
    if (kc == 96) {
      std::vector<uint16_t> eki, evi;
      attention_rel_images(rk.data.data(), rv.data.data(), kc, 10, eki, evi);
      e.evt_off = eki.size() * 2;
      eki.insert(eki.end(), evi.begin(), evi.end());
      e.rel_img.upload(reinterpret_cast<const float*>(eki.data()), eki.size() / 2);
    }
    e.o = make_conv1d(ts, p + "conv_o", 1, 0, 1, false);
    const std::string f = "enc_p.encoder.ffn_layers." + std::to_string(l) + ".";
    e.ffn1 = make_conv1d(ts, f + "conv_1", 1, (S->ksz - 1) / 2, 1, false);
    e.ffn2 = make_conv1d(ts, f + "conv_2", 1, (S->ksz - 1) / 2, 1, false);
    RVC_REQUIRE(S->ksz % 2 == 1, "enc_p FFN kernel must be odd (symmetric same-padding)");
    e.g1.upload(ts.get("enc_p.encoder.norm_layers_1." + std::to_string(l) + ".gamma", {C}).data);
    e.b1.upload(ts.get("enc_p.encoder.norm_layers_1." + std::to_string(l) + ".beta", {C}).data);
    e.g2.upload(ts.get("enc_p.encoder.norm_layers_2." + std::to_string(l) + ".gamma", {C}).data);
    e.b2.upload(ts.get("enc_p.encoder.norm_layers_2." + std::to_string(l) + ".beta", {C}).data);
  }
  S->proj = make_conv1d(ts, "enc_p.proj", 1, 0, 1, false);
  for (int f = 0; f < 4; ++f) {
    FlowLayer& F = S->flow[f];
    const std::string p = "flow.flows." + std::to_string(2 * f) + ".";
    F.pre = make_conv1d(ts, p + "pre", 1, 0, 1, false);
    F.post = make_conv1d(ts, p + "post", 1, 0, 1, false);
    F.post_neg = make_conv1d(ts, p + "post", 1, 0, 1, false, true, -1.f);
    for (int i = 0; i < 3; ++i) {
      F.in[i] = make_conv1d(ts, p + "enc.in_layers." + std::to_string(i), 1, 2, 1, true);
      if ((C & 15) == 0) {
        const std::string q = p + "enc.in_layers." + std::to_string(i);
        const HostTensor& v = ts.get(q + ".weight_v");
        const std::vector<float> w = weight_norm0(v, ts.get(q + ".weight_g"));
        const std::vector<float>& b = ts.get(q + ".bias").data;
        const int Ci = (int)v.shape[1], k = (int)v.shape[2];
        RVC_REQUIRE((int)v.shape[0] == 2 * C, q + ": expected 2 H rows");
        std::vector<float> wp(w.size()), bp(b.size());
        for (int r = 0; r < 2 * C; ++r) {
          const int src = wn_gate_row_order(r, C);
          std::copy(w.begin() + (size_t)src * Ci * k, w.begin() + (size_t)(src + 1) * Ci * k, wp.begin() + (size_t)r * Ci * k);
          bp[r] = b[src];
        }
        conv1d_layer_init(F.in_gate[i], wp.data(), bp.data(), 2 * C, Ci, k, 1, 2, 1, 1);
      }
      const std::string rs = p + "enc.res_skip_layers." + std::to_string(i);
      if (i < 2) {
        F.res[i] = make_conv1d(ts, rs, 1, 0, 1, true, true, 1.f, 0, C);
        F.skip[i] = make_conv1d(ts, rs, 1, 0, 1, true, true, 1.f, C, C);
        F.rs[i] = make_conv1d(ts, rs, 1, 0, 1, true);
      } else {
        F.skip[i] = make_conv1d(ts, rs, 1, 0, 1, true);
      }
    }
    F.cond_w.upload(weight_norm0(ts.get(p + "enc.cond_layer.weight_v"), ts.get(p + "enc.cond_layer.weight_g")));
    F.cond_b.upload(ts.get(p + "enc.cond_layer.bias").data);
  }
  S->conv_pre = make_conv1d(ts, "dec.conv_pre", 1, 3, 1, false);
  S->conv_post = make_conv1d(ts, "dec.conv_post", 1, 3, 1, false, false);
  S->conv_post_w.upload(ts.get("dec.conv_post.weight").data);
  S->dec_cond_w.upload(ts.get("dec.cond.weight").data);
  S->dec_cond_b.upload(ts.get("dec.cond.bias").data);
  if (S->f0) {
    S->lin_w = ts.get("dec.m_source.l_linear.weight").data[0];
    S->lin_b = ts.get("dec.m_source.l_linear.bias").data[0];
  }
  const int nu = (int)S->up_rates.size();
  S->stages.resize(nu);
  for (int i = 0; i < nu; ++i) {
    GenStage& st = S->stages[i];
    const int cin = S->up_init >> i, cout = S->up_init >> (i + 1);
    st.u = S->up_rates[i]; st.k = S->up_k[i];
    const std::string up = "dec.ups." + std::to_string(i);
    const HostTensor& v = ts.get(up + ".weight_v", {cin, cout, st.k});
    std::vector<float> w = weight_norm0(v, ts.get(up + ".weight_g"));
    tconv1d_layer_init(st.up, w.data(), ts.get(up + ".bias", {cout}).data.data(), cin, cout, st.k, st.u, (st.k - st.u) / 2);
    int sf0 = 1;
    for (int j = i + 1; j < nu; ++j) sf0 *= S->up_rates[j];
    const std::string nc = "dec.noise_convs." + std::to_string(i);
    if (i + 1 < nu) { st.noise_k = 2 * sf0; st.noise_s = sf0; } else { st.noise_k = 1; st.noise_s = 1; }
    if (S->f0) {
      const HostTensor& nw = ts.get(nc + ".weight", {cout, 1, st.noise_k});
      // Conv1d(1, C, k, stride) == Linear(k -> C) on the im2col frames of the source
      conv1d_layer_init(st.noise, nw.data.data(), ts.get(nc + ".bias", {cout}).data.data(), cout, st.noise_k, 1, 1, 0, 1, 1);
      if (st.noise_k <= 8) { st.noise_w.upload(nw.data); st.noise_b.upload(ts.get(nc + ".bias", {cout}).data); }
    }
    for (int j = 0; j < 3; ++j) {
      const std::string rb = "dec.resblocks." + std::to_string(i * 3 + j) + ".";
      const int k = S->rb_k[j];
      for (int m = 0; m < 3; ++m) {
        const int d = S->rb_d[j][m];
        st.rb[j].c1[m] = make_conv1d(ts, rb + "convs1." + std::to_string(m), 1, (k * d - d) / 2, d, true);
        st.rb[j].c2[m] = make_conv1d(ts, rb + "convs2." + std::to_string(m), 1, (k - 1) / 2, 1, true);
      }
    }
  }
  S->ts.clear();
  S->ready = true;
}

// ------------------------------------------------------------------------------------------------ forward
static void synth_graph(Synth* S, hipStream_t s, Arena& A, const float* feat_cm, const long long* pitch, const float* pitchf, int sid,
                        const float* noise_z, const float* noise_src, int T, float* out, const SynthTaps* taps) {
  const int C = S->hidden, H = S->n_heads, kc = C / H, IC = S->inter;
  const bool dry = A.dry;
  auto tap = [&](float* dst, const float* src, size_t n) {
    if (!dry && dst) RVC_HIP_CHECK(hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, s));
  };
  ConvEpilogue E0;
  // ---- split-resident front (conv_x3s.hip): the activations that feed enc_p's / the flow's / conv_pre's projections live as bf16 hi / lo
  // images written by their producers; k = 3 / 5 / 7 layers read them with taps as row offsets, so the images' margins (the zero padding)
  // must stay zero: the block is the graph's first allocation (nothing else ever occupies it) and is zeroed once per layout.
  static const bool x3s_on = (exp_int("RVC_X3S", 1) != 0);
  bool gs = x3s_on && conv_x3_enabled() && (C & 15) == 0 && (IC & 31) == 0 && conv_x3s_eligible(S->proj) && conv_x3s_eligible(S->conv_pre);
  for (int l = 0; l < S->n_layers && gs; ++l) {
    const EncLayer& e = S->enc[l];
    gs = conv_x3s_eligible(e.qk) && conv_x3s_eligible(e.o) && conv_x3s_eligible(e.ffn1) && conv_x3s_eligible(e.ffn2);
  }
  for (int f = 0; f < 4 && gs; ++f) {
    const FlowLayer& F = S->flow[f];
    gs = conv_x3s_eligible(F.pre) && conv_x3s_eligible(F.post_neg) && conv_x3s_eligible(F.skip[2]);
    for (int i = 0; i < 3 && gs; ++i) gs = conv_x3s_eligible(F.in[i]) && (i == 2 || conv_x3s_eligible(F.rs[i]));
  }
  const long long tp = split_image_tp(T);
  unsigned char *x_s = nullptr, *attn_s = nullptr, *ff_s = nullptr, *x0_s = nullptr, *hw_s = nullptr, *acts_s = nullptr, *z_s = nullptr;
  if (gs) {
    const size_t img0 = A.off;
    x_s = A.alloc<unsigned char>(split_image_bytes(C, T)); attn_s = A.alloc<unsigned char>(split_image_bytes(C, T));
    ff_s = A.alloc<unsigned char>(split_image_bytes(S->filt, T));
    x0_s = A.alloc<unsigned char>(split_image_bytes(IC / 2, T)); hw_s = A.alloc<unsigned char>(split_image_bytes(2 * C, T));
    acts_s = A.alloc<unsigned char>(split_image_bytes(C, T)); z_s = A.alloc<unsigned char>(split_image_bytes(IC, T));
    const size_t img_bytes = A.off - img0;
    // (a shorter sequence in the same allocation leaves the longer one's rows behind its end: the length is part of the layout)
    if (!dry && (S->img_base != A.base + img0 || S->img_gen != A.gen || S->img_bytes != img_bytes || S->img_T != T)) {
      RVC_HIP_CHECK(hipMemsetAsync(A.base + img0, 0, img_bytes, s));
      S->img_base = A.base + img0; S->img_gen = A.gen; S->img_bytes = img_bytes; S->img_T = T;
    }
  }
  // ---- speaker conditioning vectors
  const float* g = S->emb_g.p + (size_t)sid * S->gin;
  float* pre_bias = A.alloc<float>(S->up_init);
  float* gcond[4];
  for (int f = 0; f < 4; ++f) gcond[f] = A.alloc<float>(6 * C);
  if (!dry) {
    gemv(s, S->dec_cond_w.p, g, S->dec_cond_b.p, pre_bias, S->up_init, S->gin, S->conv_pre.bd_);
    for (int f = 0; f < 4; ++f) gemv(s, S->flow[f].cond_w.p, g, S->flow[f].cond_b.p, gcond[f], 6 * C, S->gin, nullptr);
  }
  // ---- enc_p
  float* x = A.alloc<float>((size_t)C * T);
  float* xb = A.alloc<float>((size_t)C * T);
  if (!dry) {
    gemm_tn_run(s, S->emb_phone_wT.p, C, 0, feat_cm, T, 0, x, T, 0, C, T, S->feat_dim, 1, S->emb_phone_b.p, 0, E0);
    encp_embed(s, x, S->f0 ? S->emb_pitch.p : nullptr, pitch, C, T);
    if (gs) split_image_from_f32(s, x, T, C, T, x_s, tp);
  }
  {
    const size_t mark = A.off;
    // attention on split-resident operands (attention_dma_kernel.h): q / k as one image, V^T by the swapped product; RVC_ATT_DMA=0: fp32 q / k / v
    static const bool att_dma = (exp_int("RVC_ATT_DMA", 1) != 0);
    static const bool rel_in = (exp_int("RVC_ENCP_REL_FUSED", 1) != 0);
    const bool ad = gs && att_dma && rel_in && kc == 96 && (exp_int("RVC_ENCP_FUSED", 1) != 0);
    unsigned char* qk_s = ad ? A.alloc<unsigned char>(split_image_bytes(2 * C, T)) : nullptr;
    unsigned char* vt_s = ad ? A.alloc<unsigned char>(attention_vt_bytes(C, T)) : nullptr;
    float* qk = ad ? nullptr : A.alloc<float>((size_t)3 * C * T);
    float* vr = ad ? nullptr : A.alloc<float>((size_t)T * C);
    static const bool fused_env = (exp_int("RVC_ENCP_FUSED", 1) != 0);
    const bool fused_att = fused_env && kc == 96;
    float* Sc = fused_att ? nullptr : A.alloc<float>((size_t)H * T * T);
    float* relk = ad ? nullptr : A.alloc<float>((size_t)H * 21 * T);
    float* pb = ad ? nullptr : A.alloc<float>((size_t)H * 21 * T);
    float* attn = ad ? nullptr : A.alloc<float>((size_t)C * T);
    float* ff = gs ? nullptr : A.alloc<float>((size_t)S->filt * T);
    if (!dry) {
      if (ad) attention_vt_clear_tail(s, vt_s, C, T);
      for (int l = 0; l < S->n_layers; ++l) {
        EncLayer& e = S->enc[l];
        if (ad) {
          static const bool qkv1 = (exp_int("RVC_QKV_FUSED", 1) != 0);
          if (qkv1 && ((2 * C) & 127) == 0) {
            // q | k | v in ONE launch: q and k rows to their image, the v rows through the transposing epilogue into the V^T image
            ConvEpilogue Eqk; Eqk.ys_out = qk_s; Eqk.ys_tp = tp; Eqk.vt_out = vt_s; Eqk.vt_tp = attention_vt_tp(C); Eqk.vt_row0 = 2 * C;
            conv_x3s_run(e.qk, s, x_s, tp, T, nullptr, T, Eqk);
          } else {
            ConvLayer qkL = e.qk; qkL.Co = 2 * C;                                 // the q and k rows of the 3 C-row projection -> image only
            ConvEpilogue Eqk; Eqk.ys_out = qk_s; Eqk.ys_tp = tp;
            conv_x3s_run(qkL, s, x_s, tp, T, nullptr, T, Eqk);
            conv_x3s_run_swapped(e.qk, 2 * C, C, s, x_s, tp, T, vt_s, attention_vt_tp(C));
          }
          const unsigned char* ri = reinterpret_cast<const unsigned char*>(e.rel_img.p);
          // softmax(K^T Q + banded rel-k bias) V + bv + banded P . E_v, written as the image the out-projection stages
          attention_split(s, qk_s, tp, 2 * C, 0, C / 16, vt_s, H, kc, T, 1.f, e.bv.p, nullptr, T, attn_s, tp, 10, ri, ri + e.evt_off);
          ConvEpilogue Er; Er.R = x; Er.ldR = T;
          conv_x3s_run(e.o, s, attn_s, tp, T, xb, T, Er);
          layernorm_c_split(s, xb, e.g1.p, e.b1.p, x, x_s, tp, kSplitMargin, C, T, T, 1e-5f);
          ConvEpilogue Ef; Ef.act = ACT_RELU; Ef.ys_out = ff_s; Ef.ys_tp = tp;
          conv_x3s_run(e.ffn1, s, x_s, tp, T, nullptr, T, Ef);
          conv_x3s_run(e.ffn2, s, ff_s, tp, T, xb, T, Er);
          layernorm_c_split(s, xb, e.g2.p, e.b2.p, x, x_s, tp, kSplitMargin, C, T, T, 1e-5f);
          if (l == 0 && taps) tap(taps->enc_p_layer0, x, (size_t)C * T);
          continue;
        }
        if (gs) conv_x3s_run(e.qk, s, x_s, tp, T, qk, T, E0); else
        conv1d_run(e.qk, s, x, T, T, qk, T, E0);
        transpose(s, qk + (size_t)2 * C * T, vr, C, T, T, C, 1, 0, 0);                                       // V row-major [T][C] (bias later)
        if (fused_att && rel_in) {
          // softmax(K^T Q + banded rel-k bias) V + bv + banded P . E_v in ONE kernel: both relative-position projections included
          attention_rel_fused(s, qk, qk + (size_t)C * T, T, vr, C, e.bv.p, nullptr, nullptr, 10, attn, T, H, kc, T, e.ek.p, e.ev.p);
        } else {
        for (int h = 0; h < H; ++h) conv1d_run(e.relk, s, qk + (size_t)h * kc * T, T, T, relk + (size_t)h * 21 * T, T, E0);
        if (fused_att) {
          // softmax(K^T Q + banded rel-k bias) V + bv in one kernel; the band of probabilities comes back in pb for the rel-v projection
          attention_rel_fused(s, qk, qk + (size_t)C * T, T, vr, C, e.bv.p, relk, pb, 10, attn, T, H, kc, T);
        } else {
          gemm_tn_run(s, qk + (size_t)C * T, T, (long long)kc * T, qk, T, (long long)kc * T, Sc, T, (long long)T * T, T, T, kc, H, nullptr, 0, E0);
          fill(s, pb, 0.f, (long long)H * 21 * T);
          softmax_cols(s, Sc, T, T, T, (long long)T * T, H, relk, 21LL * T, 10, pb, 21LL * T);
          gemm_tn_run(s, vr, C, kc, Sc, T, (long long)T * T, attn, T, (long long)kc * T, kc, T, T, H, e.bv.p, kc, E0);
        }
        ConvEpilogue Ea; Ea.accumulate = 1;
        for (int h = 0; h < H; ++h) conv1d_run(e.relv, s, pb + (size_t)h * 21 * T, T, T, attn + (size_t)h * kc * T, T, Ea);
        }
        ConvEpilogue Er; Er.R = x; Er.ldR = T;
        if (gs) {
          split_image_from_f32(s, attn, T, C, T, attn_s, tp);
          conv_x3s_run(e.o, s, attn_s, tp, T, xb, T, Er);
          layernorm_c_split(s, xb, e.g1.p, e.b1.p, x, x_s, tp, kSplitMargin, C, T, T, 1e-5f);
          ConvEpilogue Ef; Ef.act = ACT_RELU; Ef.ys_out = ff_s; Ef.ys_tp = tp;
          conv_x3s_run(e.ffn1, s, x_s, tp, T, nullptr, T, Ef);                  // k = 3: taps are row offsets into the image
          conv_x3s_run(e.ffn2, s, ff_s, tp, T, xb, T, Er);
          layernorm_c_split(s, xb, e.g2.p, e.b2.p, x, x_s, tp, kSplitMargin, C, T, T, 1e-5f);
          if (l == 0 && taps) tap(taps->enc_p_layer0, x, (size_t)C * T);
          continue;
        }
        conv1d_run(e.o, s, attn, T, T, xb, T, Er);
        layernorm_c(s, xb, nullptr, e.g1.p, e.b1.p, x, C, T, T, 1e-5f);
        ConvEpilogue Ef; Ef.act = ACT_RELU;
        conv1d_run(e.ffn1, s, x, T, T, ff, T, Ef);
        conv1d_run(e.ffn2, s, ff, T, T, xb, T, Er);
        layernorm_c(s, xb, nullptr, e.g2.p, e.b2.p, x, C, T, T, 1e-5f);
        if (l == 0 && taps) tap(taps->enc_p_layer0, x, (size_t)C * T);
      }
    }
    A.off = mark;
  }
  float* stats = A.alloc<float>((size_t)2 * IC * T);
  float* z = A.alloc<float>((size_t)IC * T);
  float* zf = A.alloc<float>((size_t)IC * T);
  if (!dry) {
    if (gs) conv_x3s_run(S->proj, s, x_s, tp, T, stats, T, E0); else
    conv1d_run(S->proj, s, x, T, T, stats, T, E0);
    if (taps) { tap(taps->m_p, stats, (size_t)IC * T); tap(taps->logs_p, stats + (size_t)IC * T, (size_t)IC * T); }
    zp_sample(s, stats, noise_z, z, IC, T);
    if (taps) tap(taps->z_p, z, (size_t)IC * T);
  }
  // ---- flow (reverse)
  {
    const size_t mark = A.off;
    float* hw = A.alloc<float>((size_t)2 * C * T);               // [h | wo]: the WaveNet's residual stream and its skip sum, adjacent rows
    float* h = hw; float* wo = hw + (size_t)C * T;
    float* xin = A.alloc<float>((size_t)2 * C * T);
    float* acts = gs ? nullptr : A.alloc<float>((size_t)C * T);
    const int half = IC / 2;
    if (!dry) {
      float* cur = z; float* oth = zf;
      for (int f = 3; f >= 0; --f) {
        FlowLayer& F = S->flow[f];
        flip_c(s, cur, oth, IC, T);
        std::swap(cur, oth);
        if (gs) {
          // every layer one launch of the split-resident GEMM: the res / skip pair of a WaveNet layer is ONE 2 H-row layer accumulating in place
          // onto [h | wo] (its image output is the next in-layer's input), post is packed negated (x1 - m = x1 + (-W) wo + (-b))
          split_image_from_f32(s, cur, T, half, T, x0_s, tp);
          unsigned char* wo_s = hw_s + split_image_bytes(C, T);
          ConvEpilogue Eh; Eh.ys_out = hw_s; Eh.ys_tp = tp;
          conv_x3s_run(F.pre, s, x0_s, tp, T, h, T, Eh);
          fill(s, wo, 0.f, (long long)C * T);
          for (int i = 0; i < 3; ++i) {
            static const bool gate1 = (exp_int("RVC_WN_GATE_FUSED", 1) != 0);
            if (gate1 && F.in_gate[i].Wx_ && conv_x3s_eligible(F.in_gate[i])) {
              // k = 5 over the first H channels of the [h | wo] image, the gate in the epilogue: acts leaves as its image, the 2 H-row tensor is never stored
              ConvEpilogue Eg; Eg.ys_out = acts_s; Eg.ys_tp = tp; Eg.gate_h = C; Eg.gate_g = gcond[f] + (size_t)i * 2 * C;
              conv_x3s_run(F.in_gate[i], s, hw_s, tp, T, nullptr, T, Eg);
            } else {
              conv_x3s_run(F.in[i], s, hw_s, tp, T, xin, T, E0);                 // k = 5 over the first H channels of the [h | wo] image
              wn_gate_split(s, xin, gcond[f] + (size_t)i * 2 * C, acts_s, tp, kSplitMargin, C, T);
            }
            if (i < 2) { ConvEpilogue Ea; Ea.R = hw; Ea.ldR = T; Ea.ys_out = hw_s; Ea.ys_tp = tp; conv_x3s_run(F.rs[i], s, acts_s, tp, T, hw, T, Ea); }
            else { ConvEpilogue Ea; Ea.R = wo; Ea.ldR = T; Ea.ys_out = wo_s; Ea.ys_tp = tp; conv_x3s_run(F.skip[2], s, acts_s, tp, T, wo, T, Ea); }
          }
          ConvEpilogue Ep; Ep.R = cur + (size_t)half * T; Ep.ldR = T;
          conv_x3s_run(F.post_neg, s, wo_s, tp, T, cur + (size_t)half * T, T, Ep);    // x1 = x1 - m
          continue;
        }
        conv1d_run(F.pre, s, cur, T, T, h, T, E0);
        for (int i = 0; i < 3; ++i) {
          conv1d_run(F.in[i], s, h, T, T, xin, T, E0);
          wn_gate(s, xin, gcond[f] + (size_t)i * 2 * C, acts, C, T);
          ConvEpilogue Es; Es.accumulate = (i > 0);
          conv1d_run(F.skip[i], s, acts, T, T, wo, T, Es);
          if (i < 2) { ConvEpilogue Er; Er.R = h; Er.ldR = T; conv1d_run(F.res[i], s, acts, T, T, h, T, Er); }
        }
        ConvEpilogue Ep; Ep.out_scale = -1.f; Ep.accumulate = 1;
        conv1d_run(F.post, s, wo, T, T, cur + (size_t)half * T, T, Ep);     // x1 = x1 - m
      }
      if (cur != z) RVC_HIP_CHECK(hipMemcpyAsync(z, cur, (size_t)IC * T * sizeof(float), hipMemcpyDeviceToDevice, s));
      if (taps) tap(taps->z, z, (size_t)IC * T);
    }
    A.off = mark;
  }
  // ---- generator
  const long long N = (long long)T * S->upp;
  float* har = S->f0 ? A.alloc<float>((size_t)N) : nullptr;
  if (S->f0) {
    float* rad = A.alloc<float>((size_t)T);
    float* tmp = A.alloc<float>((size_t)T);
    double* bsum = A.alloc<double>((size_t)((N + 1023) / 1024));
    if (!dry) {
      sine_source(s, pitchf, noise_src, har, taps ? taps->sine_waves : nullptr, rad, tmp, bsum, T, S->upp, (float)S->sr, S->lin_w, S->lin_b);
      if (taps) tap(taps->har_source, har, (size_t)N);
    }
  }
  float* cur = A.alloc<float>((size_t)S->up_init * T);
  if (!dry) {
    ConvEpilogue Eb; Eb.bias_override = pre_bias;
    if (gs) { split_image_from_f32(s, z, T, IC, T, z_s, tp); conv_x3s_run(S->conv_pre, s, z_s, tp, T, cur, T, Eb); }     // k = 7
    else conv1d_run(S->conv_pre, s, z, T, T, cur, T, Eb);
  }
  int Tc = T;
  const int nu = (int)S->stages.size();
  for (int i = 0; i < nu; ++i) {
    GenStage& st = S->stages[i];
    const int Cc = S->up_init >> (i + 1);
    const int Tn = Tc * st.u;
    float* up = A.alloc<float>((size_t)Cc * Tn);
    float* t1 = A.alloc<float>((size_t)Cc * Tn);
    float* ya = A.alloc<float>((size_t)Cc * Tn);
    float* yb = A.alloc<float>((size_t)Cc * Tn);
    float* xs = A.alloc<float>((size_t)Cc * Tn);
    float* fr = (S->f0 && st.noise_k > 1) ? A.alloc<float>((size_t)st.noise_k * Tn) : nullptr;
    // split-resident intermediate of a ResBlock pair: c1's epilogue writes t = lrelu(c1(..) + b1) as the bf16 hi / lo image c2 stages in
    // LDS (DMA, no conversion, no staging registers; 4 b128 stores per accumulator instead of 16 dword stores on c1's side)
    static const bool split_on = (exp_int("RVC_SPLIT", 1) != 0);
    // h2_pair: both halves on the persistent kernel in its fp16x2 arithmetic (two MFMAs per product; conv_x3q.hip) - the pair's image is then fp16 hi / lo
    bool split_pair[3][3], h2_pair[3][3];
    bool any_split = false;
    for (int j = 0; j < 3; ++j)
      for (int m = 0; m < 3; ++m) {
        split_pair[j][m] = split_on && conv1d_split_eligible(st.rb[j].c1[m], Tn, SPLIT_PRODUCER) && conv1d_split_eligible(st.rb[j].c2[m], Tn, SPLIT_CONSUMER);
        h2_pair[j][m] = split_pair[j][m] && conv1d_pair_h2_eligible(st.rb[j].c1[m], st.rb[j].c2[m], Tn);
        any_split = any_split || split_pair[j][m];
      }
    unsigned char* t1s = any_split ? A.alloc<unsigned char>(split_image_bytes(Cc, Tn)) : nullptr;
    if (!dry) {
      RVC_REQUIRE(conv1d_out_len(st.up, Tc) == Tn, "ConvTranspose1d geometry must give T_out = u * T_in");
      // up-sampled signal first (interleaved store of the transposed conv's phases, no read-modify-write), then the noise branch is
      // added by its own convolution's dense epilogue: the same two-operand fp32 sum as noise first / up-conv accumulating
      ConvEpilogue Eu; Eu.pre_act = ACT_LRELU; Eu.pre_slope = 0.1f;
      conv1d_run(st.up, s, cur, Tc, Tc, up, Tn, Eu);
      ConvEpilogue En; En.accumulate = 1;
      static const bool noise_stream = (exp_int("RVC_NOISE_STREAM", 1) != 0);
      // last stage (one tap of the source per position) with all three ResBlocks on conv_rb3_kernel: the noise term is added where x is read
      bool noise_in_rb3 = S->f0 && noise_stream && st.noise_k == 1 && st.noise_w.p != nullptr && N == (long long)Tn;
      for (int j = 0; j < 3 && noise_in_rb3; ++j) {
        const ConvLayer* r1[3] = {&st.rb[j].c1[0], &st.rb[j].c1[1], &st.rb[j].c1[2]};
        const ConvLayer* r2[3] = {&st.rb[j].c2[0], &st.rb[j].c2[1], &st.rb[j].c2[2]};
        noise_in_rb3 = conv_rb3_try(r1, r2, s, up, Tn, Tn, xs, Tn, 0.1f, 1.f / 3.f, j > 0, true);
      }
      if (taps && i == 0) noise_in_rb3 = false;                  // (the gen_ups0 tap wants the summed tensor)
      if (noise_in_rb3) {
        // nothing here
      } else if (!S->f0) {
        // plain Generator: nothing is added to the up-sampled signal
      } else if (noise_stream && st.noise_w.p && noise_add(s, up, Tn, Cc, Tn, har, N, st.noise_k, st.noise_s, st.noise_k > 1 ? st.noise_s / 2 : 0, st.noise_w.p, st.noise_b.p)) {
        // narrow stages (k <= 8 taps of the one source channel): a streaming add instead of im2col + GEMM
      } else if (st.noise_k > 1) {
        frames(s, har, fr, (int)N, st.noise_k, st.noise_s, st.noise_s / 2, Tn, 0);
        conv1d_run(st.noise, s, fr, Tn, Tn, up, Tn, En);
      } else {
        conv1d_run(st.noise, s, har, Tn, Tn, up, Tn, En);
      }
      if (taps && i == 0) tap(taps->gen_ups0, up, (size_t)Cc * Tn);
      for (int j = 0; j < 3; ++j) {
        const float* in = up;
        {
          // 32-channel stage in the fp16x2 arithmetic: the whole ResBlock (three pairs) in one launch, x read once, the sum written once (conv_rb3.hip)
          const ConvLayer* r1[3] = {&st.rb[j].c1[0], &st.rb[j].c1[1], &st.rb[j].c1[2]};
          const ConvLayer* r2[3] = {&st.rb[j].c2[0], &st.rb[j].c2[1], &st.rb[j].c2[2]};
          if (conv_rb3_try(r1, r2, s, up, Tn, Tn, xs, Tn, 0.1f, 1.f / 3.f, j > 0, false, noise_in_rb3 ? har : nullptr, noise_in_rb3 ? st.noise_w.p : nullptr,
                           noise_in_rb3 ? st.noise_b.p : nullptr)) continue;
          RVC_REQUIRE(!noise_in_rb3, "conv_rb3_try accepted the ResBlock in its dry run and declined the launch");
        }
        for (int m = 0; m < 3; ++m) {
          ConvEpilogue E2; E2.pre_act = ACT_LRELU; E2.pre_slope = 0.1f; E2.R = in; E2.ldR = Tn;
          float* dst = (m == 0) ? ya : (m == 1 ? yb : xs);
          if (m == 2) { E2.out_scale = 1.f / 3.f; E2.accumulate = (j > 0); }
          // narrow stages: both convs of the pair in one launch, the intermediate stays in LDS (conv_x3.hip, FUSE)
          if (!conv_x3_pair_try(st.rb[j].c1[m], st.rb[j].c2[m], s, in, Tn, Tn, dst, Tn, E2)) {
            ConvEpilogue E1; E1.pre_act = ACT_LRELU; E1.pre_slope = 0.1f;
            if (split_pair[j][m]) {
              E1.ys_out = t1s; E1.ys_tp = split_image_tp(Tn); E1.ys_slope = E2.pre_slope;       // c2's input activation, applied once by the producer
              E1.h2 = h2_pair[j][m] ? 1 : 0;
              conv1d_run(st.rb[j].c1[m], s, in, Tn, Tn, nullptr, Tn, E1);
              ConvEpilogue E2s = E2; E2s.pre_act = ACT_NONE; E2s.xs_in = t1s; E2s.xs_tp = E1.ys_tp; E2s.h2 = E1.h2;
              conv1d_run(st.rb[j].c2[m], s, nullptr, Tn, Tn, dst, Tn, E2s);
            } else {
              conv1d_run(st.rb[j].c1[m], s, in, Tn, Tn, t1, Tn, E1);
              conv1d_run(st.rb[j].c2[m], s, t1, Tn, Tn, dst, Tn, E2);
            }
          }
          in = dst;
        }
      }
      if (taps && i == nu - 1) tap(taps->gen_last, xs, (size_t)Cc * Tn);
    }
    cur = xs; Tc = Tn;
  }
  if (!dry) {
    static const bool stream_post = (exp_int("RVC_CONVPOST_STREAM", 1) != 0);
    if (stream_post) {
      conv_to1(s, cur, Tc, S->conv_post_w.p, S->up_init >> nu, 7, 3, Tc, 0.01f, 1, out);
    } else {
      ConvEpilogue Ep; Ep.pre_act = ACT_LRELU; Ep.pre_slope = 0.01f; Ep.act = ACT_TANH;
      conv1d_run(S->conv_post, s, cur, Tc, Tc, out, Tc, Ep);
    }
  }
}

void synth_infer(Synth* S, hipStream_t s, const float* feat, int feat_channel_major, const long long* pitch, const float* pitchf, int sid,
                 const float* noise_z, const float* noise_src, int T, float* out, const SynthTaps* taps) {
  RVC_REQUIRE(S->ready, "synth_finalize has not been called");
  RVC_REQUIRE(T >= 11, "need at least 11 frames (relative-position window)");
  RVC_REQUIRE(sid >= 0 && sid < S->n_spk, "speaker id out of range");
  Arena& A = S->arena;
  for (int pass = 0; pass < 2; ++pass) {
    A.dry = (pass == 0); A.reset(); if (pass == 0) A.peak = 0;
    const float* fcm = feat;
    if (!feat_channel_major) {
      float* t = A.alloc<float>((size_t)S->feat_dim * T);
      if (!A.dry) transpose(s, feat, t, T, S->feat_dim, S->feat_dim, T, 1, 0, 0);
      fcm = t;
    }
    synth_graph(S, s, A, fcm, pitch, pitchf, sid, noise_z, noise_src, T, out, taps);
    if (pass == 0) A.ensure(A.peak);
  }
  A.dry = false;
}

size_t synth_workspace(const Synth* M) { return M->arena.cap; }

}  // namespace rvc
