// RMVPE.infer_from_audio as a HIP kernel graph (reference lib/rmvpe.py:614-659): conv-STFT log-mel (:114-150,:510-556),
// Deep U-Net of 3x3 Conv2d blocks with folded BatchNorm (:233-428), Conv2d 16->3, BiGRU(384 -> 2x256), Linear(512 -> 360),
// sigmoid (:464-470) and the local-average-cents decoder (:607-612,:661-685).  NCHW with H = time, W = 128 mel bins.
#include "model_common.h"
#include "models.h"

namespace rvc {

struct CBR {           // ConvBlockRes with BatchNorm folded into both convs
  ConvLayer c1, c2, sc; bool has_sc = false; int cin = 0, cout = 0;
  ConvLayer c1sc;        // shallow levels: first convolution and 1 x 1 shortcut as ONE layer of 2 cout rows (the shortcut's weights at the centre tap) for conv3_small_run
};
struct Rmvpe {
  Ctx* ctx = nullptr;
  Arena arena;
  TensorStore ts;
  bool ready = false;
  ConvLayer stft, melproj;
  float bn_a = 1.f, bn_b = 0.f;
  CBR enc[5][4], inter[4][4], dec[5][4];
  ConvLayer dect[5];
  ConvLayer dect_t[4];     // the same transposed convolutions as 2 x 2-tap phase convolutions for the split-resident kernel (input levels 5 .. 2)
  bool pad_ok = false;     // every layer of levels >= 2 has its bf16x3 image: those levels run on padded split-resident images (conv_x3s.hip)
  const void* img_base = nullptr; unsigned img_gen = 0; size_t img_bytes = 0; int img_H1 = -1;   // image block whose margins are known to be zero (for this length)
  ConvLayer cnn;
  DevVec wihT, b_ih, w_hh, w_hh_t, b_hh;
  ConvLayer fc;
  ConvLayer wih;           // GRU input projection [1536][384] (both directions) as a k = 1 layer: its weight image feeds the swapped split-resident product
  unsigned long long* xbuf = nullptr; int* gru_err = nullptr;
  int gru_fault = 0; unsigned gru_spin_limit = 0; bool gru_no_repair = false;   // test hooks (rmvpe_debug_fault)
};

Rmvpe* rmvpe_create(Ctx* ctx) { Rmvpe* R = new Rmvpe(); R->ctx = ctx; return R; }
void rmvpe_set_tensor(Rmvpe* R, const char* name, const float* d, const long long* shape, int ndim) { R->ts.set(name, d, shape, ndim); }

static void cbr_free(CBR& b) { conv_layer_free(b.c1); conv_layer_free(b.c2); conv_layer_free(b.sc); conv_layer_free(b.c1sc); }
static void rmvpe_free(Rmvpe& R) {
  conv_layer_free(R.stft); conv_layer_free(R.melproj);
  for (auto& l : R.enc) for (auto& b : l) cbr_free(b);
  for (auto& l : R.inter) for (auto& b : l) cbr_free(b);
  for (auto& l : R.dec) for (auto& b : l) cbr_free(b);
  for (auto& c : R.dect) conv_layer_free(c);
  for (auto& c : R.dect_t) conv_layer_free(c);
  R.pad_ok = false; R.img_base = nullptr; R.img_gen = 0; R.img_bytes = 0; R.img_H1 = -1;
  conv_layer_free(R.cnn); conv_layer_free(R.fc); conv_layer_free(R.wih);
  R.wihT.free_(); R.b_ih.free_(); R.w_hh.free_(); R.w_hh_t.free_(); R.b_hh.free_();
  dev_free(R.xbuf); dev_free(R.gru_err); R.xbuf = nullptr; R.gru_err = nullptr;
}
void rmvpe_destroy(Rmvpe* R) { if (R) { rmvpe_free(*R); R->arena.release(); delete R; } }

// BatchNorm2d (eval) folded into the preceding bias-free conv: w' = w * g / sqrt(var + eps), b' = beta - mean * g / sqrt(var + eps)
static void bn_fold(const TensorStore& ts, const std::string& bn, int C, std::vector<float>& scale, std::vector<float>& shift) {
  const HostTensor& g = ts.get(bn + ".weight", {C}); const HostTensor& b = ts.get(bn + ".bias", {C});
  const HostTensor& m = ts.get(bn + ".running_mean", {C}); const HostTensor& v = ts.get(bn + ".running_var", {C});
  scale.resize(C); shift.resize(C);
  for (int c = 0; c < C; ++c) {
    const float inv = 1.f / std::sqrt(v.data[c] + 1e-5f);
    scale[c] = g.data[c] * inv; shift[c] = b.data[c] - m.data[c] * g.data[c] * inv;
  }
}

static void make_cbr(CBR& B, const TensorStore& ts, const std::string& p, int cin, int cout) {
  B.cin = cin; B.cout = cout;
  std::vector<float> sc, sh, w1f, b1f;
  {
    std::vector<float> w = ts.get(p + "conv.0.weight", {cout, cin, 3, 3}).data;
    bn_fold(ts, p + "conv.1", cout, sc, sh);
    for (int co = 0; co < cout; ++co) for (size_t i = 0; i < (size_t)cin * 9; ++i) w[(size_t)co * cin * 9 + i] *= sc[co];
    conv2d3x3_layer_init(B.c1, w.data(), sh.data(), cout, cin);
    w1f = w; b1f = sh;
  }
  {
    std::vector<float> w = ts.get(p + "conv.3.weight", {cout, cout, 3, 3}).data;
    bn_fold(ts, p + "conv.4", cout, sc, sh);
    for (int co = 0; co < cout; ++co) for (size_t i = 0; i < (size_t)cout * 9; ++i) w[(size_t)co * cout * 9 + i] *= sc[co];
    conv2d3x3_layer_init(B.c2, w.data(), sh.data(), cout, cout);
  }
  B.has_sc = cin != cout;
  if (B.has_sc) {
    const HostTensor& sw = ts.get(p + "shortcut.weight", {cout, cin, 1, 1}); const HostTensor& sb = ts.get(p + "shortcut.bias", {cout});
    conv2d1x1_layer_init(B.sc, sw.data.data(), sb.data.data(), cout, cin);
    if ((cin == 16 || cin == 32) && 2 * cout <= (cin == 16 ? 64 : 32)) {
      // rows [0, cout): the first convolution; rows [cout, 2 cout): the shortcut, a 3 x 3 kernel that is zero except for its centre tap
      std::vector<float> w((size_t)2 * cout * cin * 9, 0.f), b((size_t)2 * cout);
      std::copy(w1f.begin(), w1f.end(), w.begin());
      for (int co = 0; co < cout; ++co) for (int ci = 0; ci < cin; ++ci) w[((size_t)(cout + co) * cin + ci) * 9 + 4] = sw.data[(size_t)co * cin + ci];
      std::copy(b1f.begin(), b1f.end(), b.begin()); std::copy(sb.data.begin(), sb.data.end(), b.begin() + cout);
      conv2d3x3_layer_init(B.c1sc, w.data(), b.data(), 2 * cout, cin);
    }
  }
}

void rmvpe_finalize(Rmvpe* R) {
  // 3x3 convolutions with Ci % 16 == 0 also get a bf16x3 split weight image (conv_x3.hip); Ci = 1 / transposed convs stay fp32
  ConvBuildScope x3scope(R->ctx->precision);
  const TensorStore& ts = R->ts;
  rmvpe_free(*R);
  conv1d_layer_init(R->stft, ts.get("stft.forward_basis", {1026, 1024}).data.data(), nullptr, 1026, 1024, 1, 1, 0, 1, 1);
  conv1d_layer_init(R->melproj, ts.get("mel_basis", {128, 513}).data.data(), nullptr, 128, 513, 1, 1, 0, 1, 1);
  {
    const float g = ts.get("unet.encoder.bn.weight", {1}).data[0], b = ts.get("unet.encoder.bn.bias", {1}).data[0];
    const float m = ts.get("unet.encoder.bn.running_mean", {1}).data[0], v = ts.get("unet.encoder.bn.running_var", {1}).data[0];
    const float inv = 1.f / std::sqrt(v + 1e-5f);
    R->bn_a = g * inv; R->bn_b = b - m * g * inv;
  }
  int cin = 1, cout = 16;
  for (int i = 0; i < 5; ++i) {
    for (int b = 0; b < 4; ++b) make_cbr(R->enc[i][b], ts, "unet.encoder.layers." + std::to_string(i) + ".conv." + std::to_string(b) + ".", b == 0 ? cin : cout, cout);
    cin = cout; cout *= 2;
  }
  cin = 256; cout = 512;
  for (int i = 0; i < 4; ++i) {
    for (int b = 0; b < 4; ++b) make_cbr(R->inter[i][b], ts, "unet.intermediate.layers." + std::to_string(i) + ".conv." + std::to_string(b) + ".", b == 0 ? cin : cout, cout);
    cin = cout;
  }
  cin = 512;
  for (int i = 0; i < 5; ++i) {
    cout = cin / 2;
    const std::string p = "unet.decoder.layers." + std::to_string(i) + ".";
    std::vector<float> w = ts.get(p + "conv1.0.weight", {cin, cout, 3, 3}).data;
    std::vector<float> sc, sh;
    bn_fold(ts, p + "conv1.1", cout, sc, sh);
    for (int ci = 0; ci < cin; ++ci) for (int co = 0; co < cout; ++co) for (int k = 0; k < 9; ++k) w[((size_t)ci * cout + co) * 9 + k] *= sc[co];
    tconv2d_layer_init(R->dect[i], w.data(), sh.data(), cin, cout);
    if (i < 4 && R->dect[i].Wx_) {
      // out[2h + a][2w + b] = sum over (dh, dw) in {0, 1}^2 of W[a + 1 - 2 dh][b + 1 - 2 dw] x[h + dh][w + dw]: four phase convolutions with 2 x 2 taps
      std::vector<float> w4((size_t)4 * cout * cin * 4, 0.f), b4((size_t)4 * cout);
      for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int co = 0; co < cout; ++co) {
        b4[(size_t)(a * 2 + b) * cout + co] = sh[co];
        for (int ci = 0; ci < cin; ++ci) for (int dh = 0; dh < 2; ++dh) for (int dw = 0; dw < 2; ++dw) {
          const int kh = a + 1 - 2 * dh, kw = b + 1 - 2 * dw;
          if (kh < 0 || kh > 2 || kw < 0 || kw > 2) continue;
          w4[(((size_t)(a * 2 + b) * cout + co) * cin + ci) * 4 + dh * 2 + dw] = w[(((size_t)ci * cout + co) * 3 + kh) * 3 + kw];
        }
      }
      conv2d_kx_layer_init(R->dect_t[i], w4.data(), b4.data(), 4 * cout, cin, 2, 2, 0, 0);
    }
    for (int b = 0; b < 4; ++b) make_cbr(R->dec[i][b], ts, p + "conv2." + std::to_string(b) + ".", b == 0 ? 2 * cout : cout, cout);
    cin = cout;
  }
  conv2d3x3_layer_init(R->cnn, ts.get("cnn.weight", {3, 16, 3, 3}).data.data(), ts.get("cnn.bias", {3}).data.data(), 3, 16);
  {
    std::vector<float> wT((size_t)384 * 1536), bi(1536), wh((size_t)2 * 768 * 256), bh(1536);
    for (int d = 0; d < 2; ++d) {
      const std::string sfx = d ? "_reverse" : "";
      const HostTensor& wi = ts.get("fc.0.gru.weight_ih_l0" + sfx, {768, 384});
      for (int r = 0; r < 768; ++r) for (int k = 0; k < 384; ++k) wT[(size_t)k * 1536 + d * 768 + r] = wi.data[(size_t)r * 384 + k];
      const HostTensor& whh = ts.get("fc.0.gru.weight_hh_l0" + sfx, {768, 256});
      std::copy(whh.data.begin(), whh.data.end(), wh.begin() + (size_t)d * 768 * 256);
      const HostTensor& b1 = ts.get("fc.0.gru.bias_ih_l0" + sfx, {768}); const HostTensor& b2 = ts.get("fc.0.gru.bias_hh_l0" + sfx, {768});
      std::copy(b1.data.begin(), b1.data.end(), bi.begin() + d * 768);
      std::copy(b2.data.begin(), b2.data.end(), bh.begin() + d * 768);
    }
    R->wihT.upload(wT); R->b_ih.upload(bi); R->w_hh.upload(wh); R->b_hh.upload(bh);
    {
      std::vector<float> wt((size_t)2 * 768 * 256);          // [dir][column][row]: the serial repair kernel reads a column across its 768 threads
      for (int d = 0; d < 2; ++d) for (int r = 0; r < 768; ++r) for (int k = 0; k < 256; ++k) wt[((size_t)d * 256 + k) * 768 + r] = wh[((size_t)d * 768 + r) * 256 + k];
      R->w_hh_t.upload(wt);
    }
    std::vector<float> wrow((size_t)1536 * 384);            // [g][f] = wT[f][g]
    for (int g = 0; g < 1536; ++g) for (int k = 0; k < 384; ++k) wrow[(size_t)g * 384 + k] = wT[(size_t)k * 1536 + g];
    conv1d_layer_init(R->wih, wrow.data(), nullptr, 1536, 384, 1, 1, 0, 1, 1);
  }
  conv1d_layer_init(R->fc, ts.get("fc.1.weight", {360, 512}).data.data(), ts.get("fc.1.bias", {360}).data.data(), 360, 512, 1, 1, 0, 1, 1);
  {
    bool ok = true;
    auto cbr_ok = [&](const CBR& b) { return conv_x3s_eligible(b.c1) && conv_x3s_eligible(b.c2) && (!b.has_sc || conv_x3s_eligible(b.sc)); };
    for (int i = 2; i < 5; ++i) for (int b = 0; b < 4; ++b) ok = ok && cbr_ok(R->enc[i][b]);
    for (int i = 0; i < 4; ++i) for (int b = 0; b < 4; ++b) ok = ok && cbr_ok(R->inter[i][b]);
    for (int i = 0; i < 3; ++i) for (int b = 0; b < 4; ++b) ok = ok && cbr_ok(R->dec[i][b]);
    for (int i = 0; i < 4; ++i) ok = ok && R->dect_t[i].Wx_ != nullptr && conv_x3s_eligible(R->dect_t[i]);
    R->pad_ok = ok;
  }
  RVC_HIP_CHECK(hipMalloc(&R->xbuf, sizeof(unsigned long long) * 2 * 2 * 256));
  RVC_HIP_CHECK(hipMalloc(&R->gru_err, 2 * sizeof(int)));      // [0] hand-off timed out, [1] directions repaired by the serial kernel
  RVC_HIP_CHECK(hipMemset(R->gru_err, 0, 2 * sizeof(int)));
  R->ts.clear();
  R->ready = true;
}

// one ConvBlockRes: out = relu(bn(conv(relu(bn(conv(x)))))) + (shortcut(x) | x)       (reference lib/rmvpe.py:264-268)
static void run_cbr(const CBR& B, hipStream_t s, Arena& A, const float* x, int H, int W, float* out) {
  const long long plane = (long long)H * W;
  // the 16- and 32-channel blocks without a shortcut: both convolutions and the residual in one launch, the intermediate in LDS (conv_cbr2.hip)
  static const bool fuse_small = (exp_int("RVC_RMVPE_CBR2", 1) != 0);
  if (fuse_small && !B.has_sc && x != out && cbr2_small_eligible(B.c1, B.c2)) {
    if (!A.dry) cbr2_small_run(B.c1, B.c2, s, x, H, W, out);
    return;
  }
  const size_t mark = A.off;
  float* y1 = A.alloc<float>((size_t)B.cout * plane);
  float* scb = B.has_sc ? A.alloc<float>((size_t)B.cout * plane) : nullptr;
  if (!A.dry) {
    const bool small = fuse_small && (W & 3) == 0 && W >= 64;      // the register-weight kernels of conv_cbr2.hip (levels of 16 / 32 channels)
    const float* res = x;
    if (small && B.has_sc && B.c1sc.Wx_ && conv3_small_eligible(B.c1sc)) {
      conv3_small_run(B.c1sc, s, x, H, W, y1, scb, B.cout, B.cout, nullptr);      // relu(c1(x)) and sc(x) in one launch
      res = scb;
    } else {
      ConvEpilogue E1; E1.act = ACT_RELU;
      conv2d_run(B.c1, s, x, plane, H, W, y1, plane, E1);
      if (B.has_sc) { ConvEpilogue E0; conv1d_run(B.sc, s, x, plane, (int)plane, scb, plane, E0); res = scb; }
    }
    if (small && conv3_small_eligible(B.c2) && out != y1 && out != res) {
      conv3_small_run(B.c2, s, y1, H, W, out, nullptr, B.cout, B.cout, res);      // relu(c2(y1)) + res
    } else {
      ConvEpilogue E2; E2.act = ACT_RELU; E2.act_before_res = 1; E2.R = res; E2.ldR = plane;
      conv2d_run(B.c2, s, y1, plane, H, W, out, plane, E2);
    }
  }
  A.off = mark;
}


// Levels 2 .. 5 of the U-Net (64 .. 512 channels, 32 .. 4 mel bins) on padded split-resident images (conv_x3s.hip / split2d.hip): every 3 x 3
// convolution, shortcut and transposed convolution is one launch of the split-resident GEMM kernel - taps are row offsets into the image,
// the K split of the small levels is reduced inside the launch - and a ConvBlockRes is 2 (3 with a shortcut) launches:
//   y1 = relu(c1(x))                      image only
//   out = relu(c2(y1)) + (sc(x) | x)      fp32 (the next block's residual) + image (the next block's input)
// in: the encoder's level-1 skip tensor, plain [32][H1][64]; out: the decoder's level-1 up-sampled tensor, plain [32][H1][64] (first half of cat[1]).
struct PadLevel { int C, H, W, T; long long tp; SplitGeom g; };
static size_t pad_img_bytes(int C, const PadLevel& L) { return (size_t)(C / 16) * 4 * (size_t)L.tp * 16; }

struct PadPlan { PadLevel lv[6]; unsigned char *ipool[6], *ia[6], *ib[6], *iy[6], *icat[6]; };
// The images live in a block of their own that NOTHING else of the graph ever occupies (it is allocated before the first temporary of
// rmvpe_graph): their margins - the vertical zero padding - are zeroed once per layout and stay zero because no kernel writes there.
static void rmvpe_pad_plan(Rmvpe* R, hipStream_t s, Arena& A, int H1, PadPlan& P) {
  for (int l = 2; l <= 5; ++l) {
    PadLevel& L = P.lv[l];
    L.C = 16 << l; L.H = H1 >> (l - 1); L.W = 128 >> l; L.T = L.H * (L.W + 2); L.g = split_geom_2d(L.W);
    L.tp = ((long long)L.g.margin + L.T + 704 + 63) & ~63LL;
  }
  const size_t img0 = A.off;
  for (int l = 2; l <= 5; ++l) {
    P.ipool[l] = A.alloc<unsigned char>(pad_img_bytes(P.lv[l].C / 2 < 16 ? 16 : P.lv[l].C / 2, P.lv[l]));     // the pooled input of the level (C / 2 channels)
    P.ia[l] = A.alloc<unsigned char>(pad_img_bytes(P.lv[l].C, P.lv[l]));
    P.ib[l] = A.alloc<unsigned char>(pad_img_bytes(P.lv[l].C, P.lv[l]));
    P.iy[l] = A.alloc<unsigned char>(pad_img_bytes(P.lv[l].C, P.lv[l]));
    P.icat[l] = l <= 4 ? A.alloc<unsigned char>(pad_img_bytes(2 * P.lv[l].C, P.lv[l])) : nullptr;              // [deconv out | encoder skip]
  }
  const size_t img_bytes = A.off - img0;
  static const bool rezero = (exp_int("RVC_RMVPE_REZERO", 0) != 0);      // debugging: zero the image block on every forward
  // (a shorter clip in the same allocation leaves the longer one's rows behind its end: the length is part of the layout)
  if (!A.dry && (rezero || R->img_base != A.base + img0 || R->img_gen != A.gen || R->img_bytes != img_bytes || R->img_H1 != H1)) {
    RVC_HIP_CHECK(hipMemsetAsync(A.base + img0, 0, img_bytes, s));
    R->img_base = A.base + img0; R->img_gen = A.gen; R->img_bytes = img_bytes; R->img_H1 = H1;
  }
}

static void rmvpe_unet_padded(Rmvpe* R, hipStream_t s, Arena& A, const PadPlan& P, const float* skip1, int H1, float* up1) {
  const bool dry = A.dry;
  const PadLevel* lv = P.lv;
  unsigned char* const* ipool = P.ipool; unsigned char* const* ia = P.ia; unsigned char* const* ib = P.ib; unsigned char* const* iy = P.iy;
  unsigned char* const* icat = P.icat;
  // ---- fp32 twins (padded rows, no margins)
  float *fpool[6], *fa[6], *fb[6], *fsc[6], *fcat[6];
  for (int l = 2; l <= 5; ++l) {
    fpool[l] = A.alloc<float>((size_t)(lv[l].C / 2) * lv[l].T);
    fa[l] = A.alloc<float>((size_t)lv[l].C * lv[l].T); fb[l] = A.alloc<float>((size_t)lv[l].C * lv[l].T); fsc[l] = A.alloc<float>((size_t)lv[l].C * lv[l].T);
    fcat[l] = l <= 4 ? A.alloc<float>((size_t)2 * lv[l].C * lv[l].T) : nullptr;
  }
  size_t ph_need = 0;
  for (int l = 2; l <= 5; ++l) ph_need = std::max(ph_need, (size_t)4 * (lv[l].C / 2) * lv[l].T);
  float* ph = A.alloc<float>(ph_need);                            // phase rows [4 Co][T_in] of a transposed conv (largest: level 2 -> 1)
  if (dry) return;
  // one ConvBlockRes at level L: (xf, xs) -> (of, os); the image pointers may point into a cat image (channel offset = chunk offset)
  auto cbr = [&](const CBR& B, const PadLevel& L, const float* xf, const unsigned char* xs, float* of, unsigned char* os, unsigned char* y1s, float* scf) {
    ConvEpilogue E1; E1.act = ACT_RELU; E1.ys_out = y1s; E1.ys_tp = L.tp;
    conv_x3s_run(B.c1, s, xs, L.tp, L.T, nullptr, L.T, E1, &L.g);
    const float* res = xf;
    if (B.has_sc) {
      SplitGeom g1 = L.g; g1.ktaps = 1; g1.toff[0] = 0;                          // 1 x 1 shortcut: the same padded positions, no taps
      ConvEpilogue E0; conv_x3s_run(B.sc, s, xs, L.tp, L.T, scf, L.T, E0, &g1);
      res = scf;
    }
    ConvEpilogue E2; E2.act = ACT_RELU; E2.act_before_res = 1; E2.R = res; E2.ldR = L.T; E2.ys_out = os; E2.ys_tp = L.tp;
    conv_x3s_run(B.c2, s, y1s, L.tp, L.T, of, L.T, E2, &L.g);
  };
  // ---- encoder levels 2 .. 4
  const float* prev_f = skip1; bool prev_padded = false; int prevC = 32, prevH = H1, prevW = 64; long long prev_ld = (long long)H1 * 64;
  for (int l = 2; l <= 4; ++l) {
    const PadLevel& L = lv[l];
    pool2_pad_split(s, prev_f, prev_ld, prev_padded, prevC, prevH, prevW, fpool[l], L.T, ipool[l], L.tp, L.g.margin);
    const float* xf = fpool[l]; const unsigned char* xs = ipool[l];
    float* skip_f = fcat[l] + (size_t)L.C * L.T; unsigned char* skip_s = icat[l] + pad_img_bytes(L.C, L);
    for (int k = 0; k < 4; ++k) {
      float* of = k == 3 ? skip_f : ((k & 1) ? fb[l] : fa[l]);
      unsigned char* os = k == 3 ? skip_s : ((k & 1) ? ib[l] : ia[l]);
      cbr(R->enc[l][k], L, xf, xs, of, os, iy[l], fsc[l]);
      xf = of; xs = os;
    }
    prev_f = skip_f; prev_padded = true; prevC = L.C; prevH = L.H; prevW = L.W; prev_ld = L.T;
  }
  // ---- intermediate: level 5
  const float* cur_f; const unsigned char* cur_s;
  {
    const PadLevel& L = lv[5];
    pool2_pad_split(s, prev_f, prev_ld, true, prevC, prevH, prevW, fpool[5], L.T, ipool[5], L.tp, L.g.margin);
    cur_f = fpool[5]; cur_s = ipool[5];
    for (int i = 0; i < 4; ++i)
      for (int k = 0; k < 4; ++k) {
        const bool to_a = cur_f != fa[5];
        cbr(R->inter[i][k], L, cur_f, cur_s, to_a ? fa[5] : fb[5], to_a ? ia[5] : ib[5], iy[5], fsc[5]);
        cur_f = to_a ? fa[5] : fb[5]; cur_s = to_a ? ia[5] : ib[5];
      }
  }
  // ---- decoder: input levels 5 .. 2 (output levels 4 .. 1)
  for (int i = 0; i < 4; ++i) {
    const int lin = 5 - i, lout = lin - 1;
    const PadLevel& Li = lv[lin];
    const int Co = Li.C / 2;                                                    // channels of the output level
    const SplitGeom g2 = [&] { SplitGeom g = split_geom_2d(Li.W, 2, 2, 0, 0); g.margin = Li.g.margin; return g; }();
    ConvEpilogue Er; Er.act = ACT_RELU;
    conv_x3s_run(R->dect_t[i], s, cur_s, Li.tp, Li.T, ph, Li.T, Er, &g2);       // [4 Co][T_in]: phase-major rows, BatchNorm folded, ReLU
    if (lout >= 2) {
      const PadLevel& Lo = lv[lout];
      interleave2_pad_split(s, ph, Li.T, Co, Li.H, Li.W, fcat[lout], Lo.T, true, icat[lout], Lo.tp, Lo.g.margin);
      const float* xf = fcat[lout]; const unsigned char* xs = icat[lout];
      for (int k = 0; k < 4; ++k) {
        float* of = (k & 1) ? fb[lout] : fa[lout]; unsigned char* os = (k & 1) ? ib[lout] : ia[lout];
        cbr(R->dec[i][k], Lo, xf, xs, of, os, iy[lout], fsc[lout]);
        xf = of; xs = os;
      }
      cur_f = xf; cur_s = xs;
    } else {
      interleave2_pad_split(s, ph, Li.T, Co, Li.H, Li.W, up1, (long long)(2 * Li.H) * (2 * Li.W), false, nullptr, 0, 0);     // plain [32][H1][64]
    }
  }
}

static void rmvpe_graph(Rmvpe* R, hipStream_t s, Arena& A, const float* audio, long long L, float thred, float* mel_out, float* sal_out,
                        double* f0_out, const RmvpeTaps* taps) {
  const bool dry = A.dry;
  ConvEpilogue E0;
  const int n = (int)(L / 160) + 1;
  const int Tr = 32 * ((n - 1) / 32 + 1);
  static const bool x3s_on = (exp_int("RVC_X3S", 1) != 0);
  const bool padded = x3s_on && R->pad_ok && conv_x3_enabled();       // levels >= 2 on padded split-resident images (rmvpe_unet_padded)
  PadPlan pplan;
  if (padded) rmvpe_pad_plan(R, s, A, Tr >> 1, pplan);                // first allocation of the graph: the image block is never aliased
  // ---- log-mel
  float* mel = A.alloc<float>((size_t)128 * n);
  {
    const size_t mark = A.off;
    float* fr = A.alloc<float>((size_t)1024 * n);
    float* ft = A.alloc<float>((size_t)1026 * n);
    float* mag = A.alloc<float>((size_t)513 * n);
    if (!dry) {
      frames(s, audio, fr, (int)L, 1024, 160, 512, n, 1);
      conv1d_run(R->stft, s, fr, n, n, ft, n, E0);
      magnitude(s, ft, mag, 513, n);
      ConvEpilogue El; El.act = ACT_LOGCLAMP; El.act_slope = 1e-5f;
      conv1d_run(R->melproj, s, mag, n, n, mel, n, El);
      if (mel_out) RVC_HIP_CHECK(hipMemcpyAsync(mel_out, mel, (size_t)128 * n * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    A.off = mark;
  }
  // ---- U-Net
  int Hs[6], Ws[6];
  for (int i = 0; i < 6; ++i) { Hs[i] = Tr >> i; Ws[i] = 128 >> i; }
  float* x0 = A.alloc<float>((size_t)Tr * 128);
  if (!dry) mel_to_unet(s, mel, x0, n, Tr, R->bn_a, R->bn_b);
  float* cat[5];
  for (int i = 0; i < 5; ++i) cat[i] = A.alloc<float>((size_t)2 * (16 << i) * Hs[i] * Ws[i]);   // [deconv out | encoder skip]
  const float* cur = x0;
  for (int i = 0; i < (padded ? 2 : 5); ++i) {
    const int C = 16 << i; const long long plane = (long long)Hs[i] * Ws[i];
    float* a = A.alloc<float>((size_t)C * plane); float* b = A.alloc<float>((size_t)C * plane);
    float* skip = cat[i] + (size_t)C * plane;
    const float* in = cur;
    for (int k = 0; k < 4; ++k) { float* o = (k == 3) ? skip : ((k & 1) ? b : a); run_cbr(R->enc[i][k], s, A, in, Hs[i], Ws[i], o); in = o; }
    if (padded && i == 1) break;                                      // the level change 1 -> 2 pools straight into the padded layout
    float* pooled = A.alloc<float>((size_t)C * Hs[i + 1] * Ws[i + 1]);
    if (!dry) avgpool2(s, skip, pooled, C, Hs[i], Ws[i], plane);
    cur = pooled;
  }
  if (padded) {
    rmvpe_unet_padded(R, s, A, pplan, cat[1] + (size_t)32 * Hs[1] * Ws[1], Hs[1], cat[1]);
  } else {
    const long long plane = (long long)Hs[5] * Ws[5];
    float* a = A.alloc<float>((size_t)512 * plane); float* b = A.alloc<float>((size_t)512 * plane);
    for (int i = 0; i < 4; ++i)
      for (int k = 0; k < 4; ++k) { float* o = (cur == a) ? b : a; run_cbr(R->inter[i][k], s, A, cur, Hs[5], Ws[5], o); cur = o; }
  }
  for (int i = (padded ? 3 : 0); i < 5; ++i) {
    const int lvl = 4 - i; const int C = 16 << lvl; const long long plane = (long long)Hs[lvl] * Ws[lvl];
    if (!dry && !(padded && i == 3)) { ConvEpilogue Er; Er.act = ACT_RELU; conv2d_run(R->dect[i], s, cur, (long long)Hs[lvl + 1] * Ws[lvl + 1], Hs[lvl + 1], Ws[lvl + 1], cat[lvl], plane, Er); }
    float* a = A.alloc<float>((size_t)C * plane); float* b = A.alloc<float>((size_t)C * plane);
    const float* in = cat[lvl];
    for (int k = 0; k < 4; ++k) { float* o = (k & 1) ? b : a; run_cbr(R->dec[i][k], s, A, in, Hs[lvl], Ws[lvl], o); in = o; }
    cur = in;
  }
  if (!dry && taps && taps->unet_out) RVC_HIP_CHECK(hipMemcpyAsync(taps->unet_out, cur, (size_t)16 * Tr * 128 * sizeof(float), hipMemcpyDeviceToDevice, s));
  // ---- cnn -> BiGRU -> Linear -> sigmoid
  float* c3 = A.alloc<float>((size_t)3 * Tr * 128);
  float* feat = A.alloc<float>((size_t)384 * Tr);
  static const bool gi_env = (exp_int("RVC_RMVPE_GI_X3S", 1) != 0);
  const bool gi_x3s = gi_env && x3s_on && conv_x3_enabled() && R->wih.Wx_ != nullptr;
  unsigned char* feat_s = gi_x3s ? A.alloc<unsigned char>(split_image_bytes(384, Tr)) : nullptr;
  float* gi = A.alloc<float>((size_t)Tr * 1536);
  float* hid = A.alloc<float>((size_t)512 * Tr);
  float* sal = A.alloc<float>((size_t)360 * Tr);
  if (!dry) {
    static const bool cnn_small = (exp_int("RVC_RMVPE_CBR2", 1) != 0);
    if (cnn_small && conv3_small_eligible(R->cnn)) conv3_small_run(R->cnn, s, cur, Tr, 128, c3, nullptr, 3, 0, nullptr);      // 16 -> 3, no activation
    else conv2d_run(R->cnn, s, cur, (long long)Tr * 128, Tr, 128, c3, (long long)Tr * 128, E0);
    if (gi_x3s) {
      // gi[t][g] = sum_f W_ih[g][f] feat[f][t], f = c 128 + m: the frame features go straight from [c][t][m] into the image, the product is the
      // swapped split-resident GEMM writing fp32 rows [t][1536] (no transposition pass, no fp32-MFMA GEMM: 97 + 10 us -> ~25)
      split_image_from_tm(s, c3, 3, Tr, 128, feat_s, split_image_tp(Tr));
      conv_x3s_run_swapped(R->wih, 0, 1536, s, feat_s, split_image_tp(Tr), Tr, nullptr, 0, gi, 1536);
    } else {
      transpose(s, c3, feat, Tr, 128, 128, Tr, 3, (long long)Tr * 128, 128LL * Tr);           // [c][t][m] -> [c*128+m][t]
      gemm_tn_run(s, feat, Tr, 0, R->wihT.p, 1536, 0, gi, 1536, 0, Tr, 1536, 384, 1, nullptr, 0, E0);
    }
    gru_scan(s, gi, R->b_ih.p, R->w_hh.p, R->gru_no_repair ? nullptr : R->w_hh_t.p, R->b_hh.p, hid, R->xbuf, R->gru_err, Tr, R->gru_spin_limit, R->gru_fault);
    if (taps && taps->gru) RVC_HIP_CHECK(hipMemcpyAsync(taps->gru, hid, (size_t)512 * Tr * sizeof(float), hipMemcpyDeviceToDevice, s));
    ConvEpilogue Es; Es.act = ACT_SIGMOID;
    conv1d_run(R->fc, s, hid, Tr, Tr, sal, Tr, Es);
    if (sal_out) transpose(s, sal, sal_out, 360, n, Tr, 360, 1, 0, 0);
    if (f0_out) rmvpe_decode(s, sal, f0_out, n, Tr, thred, R->gru_err);
  }
}

void rmvpe_forward(Rmvpe* R, hipStream_t s, const float* audio, long long L, float thred, float* mel_out, float* salience_out, double* f0_out,
                   const RmvpeTaps* taps) {
  RVC_REQUIRE(R->ready, "rmvpe_finalize has not been called");
  RVC_REQUIRE(L >= 1024 && L / 160 + 1 >= 32, "audio too short for RMVPE (need >= 0.32 s)");
  Arena& A = R->arena;
  for (int pass = 0; pass < 2; ++pass) {
    A.dry = (pass == 0); A.reset(); if (pass == 0) A.peak = 0;
    rmvpe_graph(R, s, A, audio, L, thred, mel_out, salience_out, f0_out, taps);
    if (pass == 0) A.ensure(A.peak);
  }
  A.dry = false;
}

void rmvpe_decode_rm(Rmvpe* R, hipStream_t s, const float* sal_rm, long long n, float thred, double* f0) {
  Arena& A = R->arena;
  A.dry = false; A.reset(); A.ensure((size_t)360 * n * sizeof(float) + 4096);
  float* cm = A.alloc<float>((size_t)360 * n);
  transpose(s, sal_rm, cm, (int)n, 360, 360, n, 1, 0, 0);
  rmvpe_decode(s, cm, f0, (int)n, n, thred);
}

size_t rmvpe_workspace(const Rmvpe* M) { return M->arena.cap; }

// 0: the last forward's scan ran clean; 2: its hand-off timed out and the serial kernel repaired both directions (results valid);
// 1: timed out and not repaired (f0 is NaN)
int rmvpe_status(Rmvpe* R, hipStream_t s) {
  int flag[2] = {0, 0};
  if (!R->gru_err) return 0;
  RVC_HIP_CHECK(hipMemcpyAsync(flag, R->gru_err, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
  RVC_HIP_CHECK(hipStreamSynchronize(s));
  return flag[0] ? (flag[1] >= 2 ? 2 : 1) : 0;
}
// fault: 1 = one slice of the scan never publishes (its peers time out after spin_limit polls); | 2 = the repair kernel is not enqueued
void rmvpe_debug_fault(Rmvpe* R, int fault, unsigned spin_limit) { R->gru_fault = fault & 1; R->gru_no_repair = (fault & 2) != 0; R->gru_spin_limit = spin_limit; }

}  // namespace rvc
