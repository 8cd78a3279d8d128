// Internal declarations shared by the HIP translation units of librvc_hip.so (gfx950 only).
// Public C ABI: include/rvc_hip.h.
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include <stdexcept>

namespace rvc {

// ----------------------------------------------------------------------------- errors
void set_error(const std::string& msg);
struct Error : std::runtime_error { using std::runtime_error::runtime_error; };

#define RVC_HIP_CHECK(expr)                                                                       \
  do {                                                                                            \
    hipError_t _e = (expr);                                                                       \
    if (_e != hipSuccess)                                                                         \
      throw rvc::Error(std::string(#expr) + " failed: " + hipGetErrorString(_e) + " (" + __FILE__ + \
                       ":" + std::to_string(__LINE__) + ")");                                     \
  } while (0)

#define RVC_REQUIRE(cond, msg)                                                                    \
  do {                                                                                            \
    if (!(cond)) throw rvc::Error(std::string("requirement failed: ") + #cond + " : " + (msg));   \
  } while (0)

// ----------------------------------------------------------------------------- run-time knobs
// knob_int: a DOCUMENTED tuning / diagnosis knob of the product library (INTEGRATION.md lists every one with its test).
// exp_int: an experiment knob (tile choices, kernel selection, thresholds that a measurement once swept): compiled to its default in the product
// build - the name does not even reach the binary - and read from the environment only in -DRVC_EXPERIMENTS builds (tools/build_variant.sh).
inline int knob_int(const char* name, int def) { const char* e = getenv(name); return e ? atoi(e) : def; }
#ifdef RVC_EXPERIMENTS
inline int exp_int(const char* name, int def) { return knob_int(name, def); }
#define RVC_EXP_STR(name) getenv(name)
#else
#define exp_int(name, def) (def)
#define RVC_EXP_STR(name) ((const char*)nullptr)
#endif

// ----------------------------------------------------------------------------- activations
enum Act : int { ACT_NONE = 0, ACT_LRELU = 1, ACT_RELU = 2, ACT_GELU = 3, ACT_TANH = 4, ACT_SIGMOID = 5, ACT_LOGCLAMP = 6 };

// ----------------------------------------------------------------------------- conv (conv_mfma.hip)
// One implicit-GEMM convolution: Y[m, n] = epilogue( sum_k Wp[k, m] * Xtile[k, n] ).
// 1-D: X [Ci][Tin] (time contiguous), 2-D: X [Ci][H][Wd].  Weights are pre-packed on the host.
struct ConvArgs {
  const float* X; const float* W; const float* bias; const float* R; float* Y;
  int Ci;        // input channels (per group)
  int Co;        // GEMM rows that are stored (per group)
  int CoP;       // padded rows in the packed weights (multiple of 32)
  int Tin;       // 1-D: input length.  2-D: H
  int Tout;      // 1-D: output length. 2-D: H*Wd (output positions of the GEMM)
  int Wd;        // 2-D: width (power of two); 0 for 1-D
  int ktaps;     // taps per virtual channel (1-D: ceil(k/stride); 2-D: 9)
  int dil, stride, pad;
  int CK;        // virtual channels per chunk (even)
  int KT;        // taps per weight stage
  int nchunk;    // number of chunks
  int WROW;      // LDS row pitch in floats
  int PW, BWd, BH;  // 2-D: LDS row pitch of one image row, tile width, tile rows
  long long xBatch, wBatch, yBatch, rBatch; int bBatch;   // blockIdx.z strides (elements)
  long long ldX, ldY, ldR;                               // channel strides (elements)
  int pre_act; float pre_slope;                          // activation applied to X while staging
  int act; float act_slope;                              // epilogue activation
  int act_before_res;                                    // 1: act(v)+R   0: act(v+R)
  float out_scale; int accumulate;                       // y = [y +] out_scale * v
  int ostride, orows;                                    // 1-D interleaved store: m -> (co = m % orows, r = m / orows); addr = co*ldY + n*ostride + r
  int up2;                                               // 2-D: ConvTranspose 2x2 phase interleave
  int KH, KW, PH, PWL;                                   // 2-D window (0 = 3 x 3, pad 1): taps KH x KW, zero padding PH rows above / PWL columns left
};

// A convolution layer with its weights packed for the kernel and resident on the device.
struct ConvLayer {
  float* Wd_ = nullptr;    // packed weights (device)
  float* bd_ = nullptr;    // bias (device) or null
  float* bd4_ = nullptr;   // ConvTranspose2d on the bf16x3 path: bias repeated per output phase
  int mode = 1;            // 1: 1-D, 2: 2-D 3x3
  int Ci = 0, Co = 0, CoP = 0, groups = 1;
  int k = 1, stride = 1, dil = 1, pad = 0;
  int ktaps = 1, CK = 8, KT = 1, nchunk = 1;
  int tconv_u = 0;         // >0: ConvTranspose1d with stride u (polyphase rows), `Co` = u * co_real
  int co_real = 0;
  int up2 = 0;             // 2-D ConvTranspose (k3 s2 p1 op1) as 4 phase convs
  int conv_pad = 0;        // ConvTranspose1d: left pad of the equivalent stride-1 conv
  long long wBatch = 0;    // packed elements per group
  int kh = 3, kw = 3, ph = 1, pwl = 1;   // 2-D window (conv2d_kx_layer_init; the 3 x 3 layers keep the defaults)
  uint16_t* Wx_ = nullptr; // bf16x3 split image (conv_x3.hip), null when the layer only runs on the fp32 kernel
  int CoPx = 0; long long wxBatch = 0;
  uint16_t* Wh_ = nullptr; // fp16x2 (H2) image of a ResBlock-pair layer: ONE fp16 plane per (chunk, tap) unit, [chunk][tap][half][CoPx rows][8 ch] (conv_x3q.hip)
  int seg2_chunks = 0;     // conv_x3s only: 16-channel chunks of a SECOND input image appended to the reduction with tap offset 0 (conv_layer_append_x3:
                           // MDX23C's tfc2(x2) + shortcut(x) as one product); the caller names that image in SplitGeom::seg2_off
};

struct ConvEpilogue {
  const float* R = nullptr; long long ldR = 0;
  int pre_act = ACT_NONE; float pre_slope = 0.f;
  int act = ACT_NONE; float act_slope = 0.f;
  int act_before_res = 0;
  float out_scale = 1.f; int accumulate = 0;
  const float* bias_override = nullptr;   // per-call bias vector replacing the layer's own (speaker conditioning)
  int tout_limit = 0;                     // >0: compute only the first tout_limit output positions
  // split-resident activations (bf16x3 kernel only, conv_x3.hip): the tensor lives as the bf16 hi / lo image the kernel stages in LDS,
  // [16-channel chunk][hi | lo][8-channel half][kSplitMargin + t][8 ch], written by the producer's epilogue (activation `ys_slope` already applied)
  // and copied straight into LDS by the consumer (no conversion, no registers).  xs_in replaces X, ys_out replaces Y.
  const unsigned char* xs_in = nullptr; long long xs_tp = 0;
  unsigned char* ys_out = nullptr; long long ys_tp = 0; float ys_slope = 1.f;
  // conv_x3s_run only: rows [vt_row0, Co) of an image-only k = 1 projection are written TRANSPOSED as the V^T image of the attention (attention_vt_tp rows per plane) instead of
  // into ys_out - q | k | v in ONE launch (the rows below vt_row0 go to ys_out as usual); vt_row0 a multiple of 128
  unsigned char* vt_out = nullptr; long long vt_tp = 0; int vt_row0 = 0;
  // conv_x3s_run only: the WaveNet gate in the epilogue - the layer's 2 H rows packed in wn_gate_row_order (physical row p of 32-row block b = p / 32: tanh row of channel
  // 16 b + p % 32 for p % 32 < 16, sigmoid row of channel 16 b + p % 32 - 16 otherwise); ys_out receives tanh(a + g[c]) sigmoid(a' + g[H + c]) of the H channels
  int gate_h = 0; const float* gate_g = nullptr;
  int ys_deint_h = 0;                     // conv_x3s_run only: the output image is written de-interleaved for a stride-2 consumer (split_s2_h of the output length)
  // fp16x2 arithmetic (conv_x3q.hip, H2) for BOTH halves of a ResBlock pair: the intermediate image is fp16 hi / lo and each layer multiplies with its
  // one-plane fp16 weight image; only valid when conv1d_pair_h2_eligible(c1, c2, T) said so (no other kernel reads that image format)
  int h2 = 0;
};
constexpr int kSplitMargin = 64;                                                   // positions in front of t = 0 (covers every left halo)
inline long long split_image_tp(long long T) { return (T + kSplitMargin + 704 + 63) & ~63LL; }   // rows per plane: margin + T + the last tile's overhang
inline size_t split_image_bytes(int C, long long T) { return (size_t)(C / 16) * 2 * (size_t)split_image_tp(T) * 32; }
// true when this layer at this length runs on the bf16x3 kernel with a tile that has the split-resident path of its ROLE: a producer
// (writes the image: ConvEpilogue::ys_out) and a consumer (stages it: xs_in) take different branches of the tile choice, so each role
// is dry-run with its own geometry.  pre_lrelu: the producer's input activation (part of the real launch's arguments).
enum SplitRole : int { SPLIT_CONSUMER = 0, SPLIT_PRODUCER = 1 };
bool conv1d_split_eligible(const ConvLayer& L, int Tin, SplitRole role = SPLIT_CONSUMER, int h2 = 0);
// both layers of a ResBlock pair carry an fp16 image and both launches land on the persistent kernel at this length; false when the pair arithmetic is
// switched to bf16x3 (rvc_set_pair_arithmetic(0) / RVC_H2=0)
bool conv1d_pair_h2_eligible(const ConvLayer& c1, const ConvLayer& c2, int Tin);
int conv_set_pair_arithmetic(int mode);   // process-wide: 1 = fp16x2 on eligible ResBlock pairs (default), 0 = bf16x3 everywhere; < 0 queries; returns the previous mode

// host-side packing + upload (weights in PyTorch layouts)
void conv1d_layer_init(ConvLayer& L, const float* w /*[Co][Ci/groups][k]*/, const float* bias, int Co, int Ci, int k,
                       int stride, int pad, int dil, int groups);
void tconv1d_layer_init(ConvLayer& L, const float* w /*[Ci][Co][k]*/, const float* bias, int Ci, int Co, int k, int u, int pad);
void conv2d3x3_layer_init(ConvLayer& L, const float* w /*[Co][Ci][3][3]*/, const float* bias, int Co, int Ci);
void conv2d1x1_layer_init(ConvLayer& L, const float* w /*[Co][Ci]*/, const float* bias, int Co, int Ci);
void tconv2d_layer_init(ConvLayer& L, const float* w /*[Ci][Co][3][3]*/, const float* bias, int Ci, int Co);
// general KH x KW window with asymmetric zero padding, bf16x3 kernel only (no fp32 twin): conv2d_kx_try returns false when the
// geometry does not fit the kernel's tile / LDS limits (callers keep another formulation for that case); dry = eligibility only
void conv2d_kx_layer_init(ConvLayer& L, const float* w /*[Co][Ci][KH][KW]*/, const float* bias, int Co, int Ci, int KH, int KW, int PH, int PWL);
void conv_layer_free(ConvLayer& L);
// Layers initialised while this is on also get a bf16x3 split weight image and run on conv_x3_kernel when eligible
// (stride 1, groups 1, Ci % 16 == 0): 3 bf16 MFMAs per fp32 product, fp32 accumulate, ~1e-5 relative error.
// Both switches are THREAD-LOCAL (a model is built by the thread that calls *_finalize; two threads building models at the same time
// cannot see each other's scopes).  Models take their mode from their context (Ctx::precision) through ConvBuildScope.
bool conv_x3_set_default(bool on);   // returns the previous value
int conv_set_precision(int mode);    // 0: fp32 only, 1: layers initialised under conv_x3_set_default(true), 2: every eligible layer; returns the previous mode
struct ConvBuildScope {              // RAII: layers initialised inside get bf16x3 images according to `precision`
  int prev_mode; bool prev_default;
  explicit ConvBuildScope(int precision) : prev_mode(conv_set_precision(-1)), prev_default(conv_x3_set_default(true)) {   // < 0: keep the thread's mode
    if (precision >= 0) conv_set_precision(precision);
  }
  ~ConvBuildScope() { conv_x3_set_default(prev_default); conv_set_precision(prev_mode); }
  ConvBuildScope(const ConvBuildScope&) = delete; ConvBuildScope& operator=(const ConvBuildScope&) = delete;
};

// launches.  1-D: X [Ci][Tin] with channel stride ldX, Y [Co][Tout] with channel stride ldY.
int conv1d_out_len(const ConvLayer& L, int Tin);
void conv1d_run(const ConvLayer& L, hipStream_t s, const float* X, long long ldX, int Tin, float* Y, long long ldY,
                const ConvEpilogue& e);
// GEMM with an activation tensor as the "weight" operand: Y[z][m][n] = sum_k A[z][k][m] * B[z][k][n]  (k = channel)
void gemm_tn_run(hipStream_t s, const float* A, long long ldA, long long aBatch, const float* B, long long ldB, long long bBatch,
                 float* Y, long long ldY, long long yBatch, int M, int N, int K, int batch, const float* bias, int biasBatch,
                 const ConvEpilogue& e);
// 2-D: X [Ci][H][Wd], Y [Co][H][Wd] (or [Co][2H][2Wd] for up2).
void conv2d_run(const ConvLayer& L, hipStream_t s, const float* X, long long ldX, int H, int Wd, float* Y, long long ldY,
                const ConvEpilogue& e);

// bf16x3 GEMM of a k = 1 layer on a SPLIT-RESIDENT activation (conv_x3s.hip): both operand tiles by LDS-DMA, K split reduced inside the
// launch.  Xs: image of the [Ci][T] input (split_image_tp(T) rows per plane); Y (fp32 [Co][ldY]) and / or e.ys_out (image of the output, the
// activation e.act - identity, (leaky) ReLU, exact GELU - applied before the split).  e.R: residual, e.bias_override as in conv1d_run.
bool conv_x3_enabled();                       // bf16x3 kernels not switched off at run time (RVC_X3=0)
// Layers with taps run on the same kernel: a unit of the reduction is (16-channel chunk, tap) and a tap is a row offset into the image
// (im2col by address).  1-D "same" convolutions need nothing else (rows in front of position 0 / behind position T - 1 are the zero padding:
// producers keep the margins zero).  2-D convolutions run over PADDED images: row pitch W + 2 with a zero column on either side, position
// p = h (W + 2) + w + 1, T = H (W + 2); the kernel writes zeros into the pad columns of its outputs (SplitGeom from split_geom_2d).
struct SplitGeom { int ktaps = 1; int toff[16] = {0}; int padw = 0; int margin = kSplitMargin;
                   long long seg2_off = 0;      // byte offset (from the first image, < 2 GiB, same rows per plane and margin) of the second image of a layer with seg2_chunks
                   int s2_h = 0; };             // > 0: stride-2 "valid" convolution over a DE-INTERLEAVED image (split_geom_s2): even positions at rows margin + p, odd ones at margin + s2_h + p   // byte offset (from the first image, < 2 GiB, same rows per plane and margin) of the second image of a layer with seg2_chunks
// dst (a conv_x3s-eligible layer) += the k = 1 layer `extra` over a second input: the weight image of `extra` is appended as further units of dst's reduction
void conv_layer_append_x3(ConvLayer& dst, const ConvLayer& extra);
SplitGeom split_geom_2d(int Wd, int KH = 3, int KW = 3, int PH = 1, int PWL = 1);
// Stride-2 convolutions without padding (HuBERT's feature encoder, modeling_hubert.py conv_layers 1 .. 6: k = 3 / 2) on the split-resident GEMM: the producer writes the
// input image de-interleaved (ConvEpilogue::ys_deint_h / hubert_conv0_gn_gelu), so tap t of output p - input position 2 p + t - is row p + (t >> 1) of plane t & 1: a row
// offset like every other tap.  H = rows between the two planes (split_s2_h(T_in)); the image has split_s2_tp(T_in) rows per plane.
inline int split_s2_h(long long Tin) { return (int)(((Tin + 1) / 2 + 64 + 63) & ~63LL); }
inline long long split_s2_tp(long long Tin) { return (kSplitMargin + 2LL * split_s2_h(Tin) + 704 + 63) & ~63LL; }
inline size_t split_s2_bytes(int C, long long Tin) { return (size_t)(C / 16) * 2 * (size_t)split_s2_tp(Tin) * 32; }
SplitGeom split_geom_s2(int k, long long Tin);
bool conv_x3s_s2_eligible(const ConvLayer& L);
inline int wn_gate_row_order(int p, int H) { const int b = p >> 5, q = p & 31; return q < 16 ? 16 * b + q : H + 16 * b + (q - 16); }      // physical row -> row of the reference's [tanh H | sigmoid H] layer
void split_image_deint_from_f32(hipStream_t s, const float* X, long long ldX, int C, int T, unsigned char* img, long long tp, int H);      // (tests: in the models the producers' epilogues write the image)
void split_image_deint_to_f32(hipStream_t s, const unsigned char* img, long long tp, int H, int C, int T, float* Y, long long ldY);
bool conv_x3s_eligible(const ConvLayer& L);
void conv_x3s_run(const ConvLayer& L, hipStream_t s, const unsigned char* Xs, long long xsTp, int T, float* Y, long long ldY, const ConvEpilogue& e,
                  const SplitGeom* geom = nullptr);
// the swapped product: out[t][j] = sum_c X[c][t] W[row0 + j][c], written as the image of the transposed tensor (V^T for attention_split)
void conv_x3s_run_swapped(const ConvLayer& L, int row0, int rows, hipStream_t s, const unsigned char* Xs, long long xsTp, int T, unsigned char* Ys, long long ysTp,
                          float* Yrm = nullptr, long long ldYrm = 0,      // Yrm: also / instead fp32 out[t][j] with pitch ldYrm
                          const float* Rrm = nullptr, long long ldRrm = 0);  // Rrm: residual added to the product, laid out like Yrm (MDX23C's x + tdf(x))
void split_image_from_tm(hipStream_t s, const float* x, int C, int T, int M, unsigned char* img, long long tp);      // x [C][T][M] -> image of the (C M) x T tensor
void conv_x3s_force(int ksplit, int am, int an);
void conv_x3s_set_mode(int mode);      // 0: by shape, 1: LDS-ring kernel, 2: register-direct kernel
// one ConvBlockRes of 16 or 32 channels (3 x 3, 3 x 3, + x) in one launch (conv_cbr2.hip): x, out fp32 [C][H W], distinct
bool cbr2_small_eligible(const ConvLayer& c1, const ConvLayer& c2);
void cbr2_small_run(const ConvLayer& c1, const ConvLayer& c2, hipStream_t s, const float* x, int H, int W, float* out);
// one 3 x 3 convolution of 16 / 32 input and <= 64 output channels on the same structure: rows below relu_rows get the ReLU, rows >= split_row go to Y2, R added to Y's rows
bool conv3_small_eligible(const ConvLayer& L);
void conv3_small_run(const ConvLayer& L, hipStream_t s, const float* x, int H, int W, float* Y, float* Y2, int split_row, int relu_rows, const float* R);      // tests / benchmarks: K split and tile of the calling thread's next launches (0 = automatic)
void split_image_from_f32(hipStream_t s, const float* X, long long ldX, int C, int T, unsigned char* img, long long tp);
void split_image_to_f32(hipStream_t s, const unsigned char* img, long long tp, int C, int T, float* Y, long long ldY);

bool conv2d_kx_try(const ConvLayer& L, hipStream_t s, const float* X, long long ldX, int H, int Wd, float* Y, long long ldY, const ConvEpilogue& e,
                   bool dry = false);

// fused multi-head attention (attention.hip): Q, K channel-major [heads*64][T], V row-major [T][heads*64], out channel-major
// out (fp32, channel-major) and / or out_img (the split-resident image of conv_x3s.hip, img_tp rows per plane) receive the result
void attention_fused(hipStream_t s, const float* Q, const float* K, long long ldqk, const float* V, long long ldv, const float* bv,
                     float* out, long long ldo, int heads, int dhead, int T, unsigned char* out_img = nullptr, long long img_tp = 0);

// ek / ev: emb_rel_k / emb_rel_v [2 win + 1][dhead] (shared by the heads): the relative-position projections are then computed inside the
// kernel (rel / pb may be null); without them rel holds Q . E_k and pb returns the band for a separate value-side projection
// attention on split-resident operands (attention_dma.hip): q / k channels in one image (head h's chunks at q_chunk0 / k_chunk0 + 4 h), V^T image
// [16-key chunk][plane][attention_vt_tp rows][16 B] (conv_x3s_run_swapped; key chunks past T zeroed by attention_vt_clear_tail)
long long attention_vt_tp(int channels);
size_t attention_vt_bytes(int channels, int T);
void attention_vt_clear_tail(hipStream_t s, unsigned char* vt_img, int channels, int T);
void attention_split(hipStream_t s, const unsigned char* qk_img, long long qk_tp, int qk_channels, int q_chunk0, int k_chunk0, const unsigned char* vt_img,
                     int heads, int dhead, int T, float scale, const float* bv, float* out, long long ldo, unsigned char* out_img, long long img_tp,
                     int win = 0, const unsigned char* ek_img = nullptr, const unsigned char* evt_img = nullptr);
void attention_split_force_kz(int kz);
// the synthesizer's text encoder (head dimension 96, window 10): emb_rel_k / emb_rel_v [2 win + 1][D] (host) -> the operand images attention_split takes
void attention_rel_images(const float* ek, const float* ev, int D, int win, std::vector<uint16_t>& ek_img, std::vector<uint16_t>& evt_img);
void attention_rel_fused(hipStream_t s, const float* Q, const float* K, long long ldqk, const float* V, long long ldv, const float* bv,
                         const float* rel, float* pb, int win, float* out, long long ldo, int heads, int dhead, int T, const float* ek = nullptr,
                         const float* ev = nullptr);

// per-launch HIP-event profiling of the conv kernels (bench.py's roofline leg)
void conv_prof_enable(bool on);
int conv_prof_collect(double* ms, double* flops, long long* launches);   // arrays of RVC_PROF_CFGS (tile configuration x {fp32 1-D, fp32 2-D, bf16x3})
const char* conv_prof_cfg_name(int i);
void conv_timing_read(unsigned long long* out8, bool reset);   // debug builds (-DRVC_CONV_TIMING): per-phase cycle sums

// ----------------------------------------------------------------------------- device memory
float* dev_upload(const float* host, size_t n);
void* stream_scratch(hipStream_t s, int slot, size_t bytes);   // persistent per-(device, stream) scratch (grows on demand)
void* stream_scratch_zeroed(hipStream_t s, int slot, size_t bytes, bool* fresh = nullptr);   // zero-filled when (re)allocated; users leave it zero
void stream_scratch_release(int device);                        // frees the scratch of one device (last context of the device destroyed)
void dev_free(void* p);

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to a device's copy of the kernel: set it once per (call site = kernel instantiation, device),
// not once per process - rvc_ctx_create takes a device id, and a second GPU driven from the same process would otherwise launch with the default
// 64 KiB limit (advisor, round 3).  Two threads racing on the first launch both set it: harmless.
#define RVC_ALLOW_BIG_LDS(kern)                                                                                                              \
  do {                                                                                                                                       \
    static std::atomic<unsigned long long> rvc_lds_mask_[4];                                                                                 \
    int rvc_dev_ = 0; (void)hipGetDevice(&rvc_dev_);                                                                                         \
    const unsigned long long rvc_bit_ = 1ull << (rvc_dev_ & 63);                                                                             \
    std::atomic<unsigned long long>& rvc_m_ = rvc_lds_mask_[(rvc_dev_ >> 6) & 3];                                                            \
    if (!(rvc_m_.load(std::memory_order_acquire) & rvc_bit_)) {                                                                              \
      RVC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));      \
      rvc_m_.fetch_or(rvc_bit_, std::memory_order_release);                                                                                  \
    }                                                                                                                                        \
  } while (0)

struct Arena {
  char* base = nullptr; size_t cap = 0, off = 0, peak = 0; bool dry = false;
  unsigned gen = 0;   // bumped whenever the block is (re)allocated or released: a cached "these bytes are known to be zero" must not survive that
  void reset() { off = 0; }
  template <typename T> T* alloc(size_t n) {
    size_t bytes = (n * sizeof(T) + 255) & ~size_t(255);
    size_t o = off; off += bytes; if (off > peak) peak = off;
    if (dry) return reinterpret_cast<T*>(uintptr_t(0x1000) + o);   // never dereferenced
    RVC_REQUIRE(off <= cap, "arena overflow");
    return reinterpret_cast<T*>(base + o);
  }
  void ensure(size_t bytes);
  void release();
};

}  // namespace rvc
