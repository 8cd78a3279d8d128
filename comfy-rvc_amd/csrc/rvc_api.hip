// extern "C" layer of librvc_hip.so (see include/rvc_hip.h).  Every entry point converts C++ exceptions into a status code
// plus a thread-local message; nothing here has a CPU fallback.
#include "models.h"
#include "conv_kernels.h"

namespace rvc {
static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
}  // namespace rvc

using namespace rvc;

struct rvc_ctx { Ctx c; };
static std::mutex g_ctx_mu;
static std::map<int, int> g_ctx_count;       // live contexts per device (the per-stream scratch is released with the last one)
struct rvc_hubert { Hubert* m; rvc_ctx* ctx; };
struct rvc_rmvpe { Rmvpe* m; rvc_ctx* ctx; };
struct rvc_synth { Synth* m; rvc_ctx* ctx; };
struct rvc_conv1d_plan { ConvLayer L; };

#define RVC_TRY try {
#define RVC_CATCH                                                            \
  return 0;                                                                  \
  }                                                                          \
  catch (const std::exception& e) { rvc::set_error(e.what()); return 1; }    \
  catch (...) { rvc::set_error("unknown error"); return 2; }

static void check_launch() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) throw Error(std::string("kernel launch failed: ") + hipGetErrorString(e));
}

extern "C" {

const char* rvc_last_error(void) { return g_err.c_str(); }
const char* rvc_version(void) { return "rvc_hip 0.1.0 (gfx950, fp32 MFMA)"; }

int rvc_ctx_create(int device_id, rvc_ctx** out) {
  RVC_TRY
  int n = 0;
  RVC_HIP_CHECK(hipGetDeviceCount(&n));
  RVC_REQUIRE(n > 0 && device_id >= 0 && device_id < n, "no such HIP device (this library has no CPU path)");
  RVC_HIP_CHECK(hipSetDevice(device_id));
  hipDeviceProp_t prop;
  RVC_HIP_CHECK(hipGetDeviceProperties(&prop, device_id));
  RVC_REQUIRE(std::string(prop.gcnArchName).find("gfx950") != std::string::npos,
              std::string("kernels are built for gfx950 only, found ") + prop.gcnArchName);
  rvc_ctx* c = new rvc_ctx();
  c->c.device = device_id;
  { std::lock_guard<std::mutex> lk(g_ctx_mu); ++g_ctx_count[device_id]; }
  *out = c;
  RVC_CATCH
}
// Destroying the LAST context of a device frees that device's per-stream scratch: the caller must not have calls in flight on that device
// from other threads at that moment (include/rvc_hip.h says so at rvc_ctx_destroy; handles of a destroyed context are dead anyway).
int rvc_ctx_destroy(rvc_ctx* ctx) {
  if (!ctx) return 0;
  bool last;
  { std::lock_guard<std::mutex> lk(g_ctx_mu); last = --g_ctx_count[ctx->c.device] <= 0; }
  if (last) { (void)hipSetDevice(ctx->c.device); stream_scratch_release(ctx->c.device); }   // per-stream scratch goes with the last context of a device
  delete ctx;
  return 0;
}
int rvc_ctx_set_conv_precision(rvc_ctx* ctx, int mode) {
  RVC_TRY
  RVC_REQUIRE(ctx && mode >= -1 && mode <= 2, "precision mode must be -1 (thread default), 0, 1 or 2");
  ctx->c.precision = mode;
  RVC_CATCH
}
int64_t rvc_ctx_workspace_bytes(rvc_ctx* ctx) { return ctx ? (int64_t)ctx->c.workspace_bytes : 0; }

// ------------------------------------------------------------------------------------------------ hubert
int rvc_hubert_create(rvc_ctx* ctx, rvc_hubert** out) {
  RVC_TRY
  RVC_REQUIRE(ctx && out, "null argument");
  rvc_hubert* h = new rvc_hubert(); h->ctx = ctx; h->m = hubert_create(&ctx->c); *out = h;
  RVC_CATCH
}
int rvc_hubert_set_tensor(rvc_hubert* h, const char* name, const float* d, const int64_t* shape, int ndim) {
  RVC_TRY
  RVC_REQUIRE(h && name && d && ndim <= 8, "bad argument");
  long long sh[8]; for (int i = 0; i < ndim; ++i) sh[i] = shape[i];
  hubert_set_tensor(h->m, name, d, sh, ndim);
  RVC_CATCH
}
int rvc_hubert_finalize(rvc_hubert* h) { RVC_TRY RVC_HIP_CHECK(hipSetDevice(h->ctx->c.device)); hubert_finalize(h->m); RVC_CATCH }
int rvc_hubert_destroy(rvc_hubert* h) { if (h) { hubert_destroy(h->m); delete h; } return 0; }
int64_t rvc_hubert_num_frames(int64_t L) { return hubert_num_frames(L); }
int rvc_hubert_forward(rvc_hubert* h, void* stream, const float* audio, int64_t L, int version, int n_layers, float* out_rm, float* out_cm,
                       const rvc_hubert_taps* taps) {
  RVC_TRY
  RVC_REQUIRE(h && audio, "null argument");
  hubert_forward(h->m, (hipStream_t)stream, audio, L, version, n_layers, out_rm, out_cm, taps);
  check_launch();
  h->ctx->c.workspace_bytes = hubert_workspace(h->m);
  RVC_CATCH
}

// ------------------------------------------------------------------------------------------------ rmvpe
int rvc_rmvpe_create(rvc_ctx* ctx, rvc_rmvpe** out) {
  RVC_TRY
  RVC_REQUIRE(ctx && out, "null argument");
  rvc_rmvpe* r = new rvc_rmvpe(); r->ctx = ctx; r->m = rmvpe_create(&ctx->c); *out = r;
  RVC_CATCH
}
int rvc_rmvpe_set_tensor(rvc_rmvpe* r, const char* name, const float* d, const int64_t* shape, int ndim) {
  RVC_TRY
  RVC_REQUIRE(r && name && d && ndim <= 8, "bad argument");
  long long sh[8]; for (int i = 0; i < ndim; ++i) sh[i] = shape[i];
  rmvpe_set_tensor(r->m, name, d, sh, ndim);
  RVC_CATCH
}
int rvc_rmvpe_finalize(rvc_rmvpe* r) { RVC_TRY RVC_HIP_CHECK(hipSetDevice(r->ctx->c.device)); rmvpe_finalize(r->m); RVC_CATCH }
int rvc_rmvpe_destroy(rvc_rmvpe* r) { if (r) { rmvpe_destroy(r->m); delete r; } return 0; }
int rvc_rmvpe_forward(rvc_rmvpe* r, void* stream, const float* audio, int64_t L, float thred, float* mel, float* sal, double* f0,
                      const rvc_rmvpe_taps* taps) {
  RVC_TRY
  RVC_REQUIRE(r && audio, "null argument");
  rmvpe_forward(r->m, (hipStream_t)stream, audio, L, thred, mel, sal, f0, taps);
  check_launch();
  RVC_CATCH
}
int rvc_rmvpe_status(rvc_rmvpe* r, void* stream) {
  RVC_TRY
  RVC_REQUIRE(r, "null argument");
  // (2 = the hand-off timed out and the serial kernel behind the scan repaired it: the forward is valid; rvc_rmvpe_repaired tells)
  RVC_REQUIRE(rmvpe_status(r->m, (hipStream_t)stream) != 1,
              "RMVPE: the GRU scan's workgroups timed out waiting for each other (they need to be co-resident) and the repair pass did not run; the f0 of the last forward is invalid (NaN)");
  RVC_CATCH
}
int rvc_rmvpe_repaired(rvc_rmvpe* r, void* stream) { return (r && r->m && rmvpe_status(r->m, (hipStream_t)stream) == 2) ? 1 : 0; }
int rvc_rmvpe_debug_fault(rvc_rmvpe* r, int fault, unsigned spin_limit) { RVC_TRY RVC_REQUIRE(r, "null argument"); rmvpe_debug_fault(r->m, fault, spin_limit); RVC_CATCH }
int rvc_rmvpe_decode(rvc_rmvpe* r, void* stream, const float* sal, int64_t n, float thred, double* f0) {
  RVC_TRY
  RVC_REQUIRE(r && sal && f0 && n > 0, "bad argument");
  rmvpe_decode_rm(r->m, (hipStream_t)stream, sal, n, thred, f0);
  check_launch();
  RVC_CATCH
}
int rvc_f0_post(void* stream, const double* f0, int64_t n, double factor, double mel_min, double mel_max, int bins, int64_t* pitch, float* pitchf) {
  RVC_TRY
  RVC_REQUIRE(f0 && pitch && pitchf && n > 0 && bins > 2 && mel_max > mel_min, "bad argument");
  f0_post((hipStream_t)stream, f0, n, factor, mel_min, mel_max, bins, reinterpret_cast<long long*>(pitch), pitchf);
  check_launch();
  RVC_CATCH
}

// ------------------------------------------------------------------------------------------------ crepe
struct rvc_crepe { Crepe* m; rvc_ctx* ctx; };
int rvc_crepe_create(rvc_ctx* ctx, int tiny, rvc_crepe** out) {
  RVC_TRY
  RVC_REQUIRE(ctx && out, "null argument");
  rvc_crepe* c = new rvc_crepe(); c->ctx = ctx; c->m = nullptr;
  try { c->m = crepe_create(&ctx->c, tiny); } catch (...) { delete c; throw; }
  *out = c;
  RVC_CATCH
}
int rvc_crepe_set_tensor(rvc_crepe* c, const char* name, const float* d, const int64_t* shape, int ndim) {
  RVC_TRY
  RVC_REQUIRE(c && name && d && ndim <= 8, "bad argument");
  long long sh[8]; for (int i = 0; i < ndim; ++i) sh[i] = shape[i];
  crepe_set_tensor(c->m, name, d, sh, ndim);
  RVC_CATCH
}
int rvc_crepe_finalize(rvc_crepe* c) { RVC_TRY RVC_REQUIRE(c, "null argument"); RVC_HIP_CHECK(hipSetDevice(c->ctx->c.device)); crepe_finalize(c->m); RVC_CATCH }
int rvc_crepe_destroy(rvc_crepe* c) { if (c) { crepe_destroy(c->m); delete c; } return 0; }
int64_t rvc_crepe_num_frames(int64_t L, int hop, int pad) { return crepe_num_frames(L, hop, pad); }
int rvc_crepe_forward(rvc_crepe* c, void* stream, const float* audio, int64_t L, int hop, int pad, float* probs, const rvc_crepe_taps* taps) {
  RVC_TRY
  RVC_REQUIRE(c && audio && probs && hop > 0, "bad argument");
  crepe_forward(c->m, (hipStream_t)stream, audio, L, hop, pad, probs, taps);
  check_launch();
  RVC_CATCH
}

int rvc_crepe_viterbi(void* stream, const float* probs, int64_t n, int min_bin, int max_bin, int32_t* bins, float* periodicity) {
  RVC_TRY
  RVC_REQUIRE(probs && bins && periodicity && n > 0 && n < (1 << 24) && min_bin >= 0 && max_bin <= 360 && min_bin < max_bin, "bad argument");
  crepe_viterbi((hipStream_t)stream, probs, (int)n, min_bin, max_bin, bins, periodicity);
  check_launch();
  RVC_CATCH
}

// ------------------------------------------------------------------------------------------------ mdx23c
struct rvc_mdx23 { Mdx23* m; rvc_ctx* ctx; };
int rvc_mdx23_create(rvc_ctx* ctx, const rvc_mdx23_config* cfg, rvc_mdx23** out) {
  RVC_TRY
  RVC_REQUIRE(ctx && cfg && out, "null argument");
  rvc_mdx23* h = new rvc_mdx23(); h->ctx = ctx; h->m = nullptr;
  try { h->m = mdx23_create(&ctx->c, *cfg); } catch (...) { delete h; throw; }
  *out = h;
  RVC_CATCH
}
int rvc_mdx23_set_tensor(rvc_mdx23* m, const char* name, const float* d, const int64_t* shape, int ndim) {
  RVC_TRY
  RVC_REQUIRE(m && name && d && ndim <= 8, "bad argument");
  long long sh[8]; for (int i = 0; i < ndim; ++i) sh[i] = shape[i];
  mdx23_set_tensor(m->m, name, d, sh, ndim);
  RVC_CATCH
}
int rvc_mdx23_finalize(rvc_mdx23* m) { RVC_TRY RVC_REQUIRE(m, "null argument"); RVC_HIP_CHECK(hipSetDevice(m->ctx->c.device)); mdx23_finalize(m->m); RVC_CATCH }
int rvc_mdx23_destroy(rvc_mdx23* m) { if (m) { mdx23_destroy(m->m); delete m; } return 0; }
int rvc_mdx23_forward(rvc_mdx23* m, void* stream, const float* chunk, int64_t L, float* out) {
  RVC_TRY
  RVC_REQUIRE(m && chunk && out, "null argument");
  mdx23_forward(m->m, (hipStream_t)stream, chunk, L, out);
  check_launch();
  RVC_CATCH
}
int rvc_mdx23_set_streams(rvc_mdx23* m, int k) { RVC_TRY RVC_REQUIRE(m, "null argument"); mdx23_set_streams(m->m, k); RVC_CATCH }
int rvc_mdx23_demix(rvc_mdx23* m, void* stream, const float* mix, int64_t Lp, int64_t step, int64_t n_chunks, float overlap, float* acc) {
  RVC_TRY
  RVC_REQUIRE(m && mix && acc, "null argument");
  mdx23_demix(m->m, (hipStream_t)stream, mix, Lp, step, n_chunks, overlap, acc);
  check_launch();
  RVC_CATCH
}

// ------------------------------------------------------------------------------------------------ synth
int rvc_synth_create(rvc_ctx* ctx, const rvc_synth_config* cfg, rvc_synth** out) {
  RVC_TRY
  RVC_REQUIRE(ctx && cfg && out, "null argument");
  rvc_synth* s = new rvc_synth(); s->ctx = ctx; s->m = nullptr;
  try { s->m = synth_create(&ctx->c, *cfg); } catch (...) { delete s; throw; }
  *out = s;
  RVC_CATCH
}
int rvc_synth_set_tensor(rvc_synth* s, const char* name, const float* d, const int64_t* shape, int ndim) {
  RVC_TRY
  RVC_REQUIRE(s && name && d && ndim <= 8, "bad argument");
  long long sh[8]; for (int i = 0; i < ndim; ++i) sh[i] = shape[i];
  synth_set_tensor(s->m, name, d, sh, ndim);
  RVC_CATCH
}
int rvc_synth_finalize(rvc_synth* s) { RVC_TRY RVC_HIP_CHECK(hipSetDevice(s->ctx->c.device)); synth_finalize(s->m); RVC_CATCH }
int rvc_synth_destroy(rvc_synth* s) { if (s) { synth_destroy(s->m); delete s; } return 0; }
int rvc_synth_upp(rvc_synth* s) { return s ? synth_upp(s->m) : 0; }
int rvc_synth_has_f0(rvc_synth* s) { return (s && synth_has_f0(s->m)) ? 1 : 0; }
// pitch / pitchf / noise_src belong to the f0 family only: a no-f0 model takes (and must be given) NULL for all three
static void check_pitch_args(rvc_synth* s, const void* pitch, const void* pitchf, const void* noise_src, int do_protect) {
  if (synth_has_f0(s->m)) {
    RVC_REQUIRE(pitch && pitchf && noise_src, "this model was trained with f0: pitch, pitchf and the source noise are required");
  } else {
    RVC_REQUIRE(!pitch && !pitchf && !noise_src && !do_protect, "no-f0 model: pitch, pitchf, source noise must be NULL and protect off");
  }
}
int rvc_synth_infer(rvc_synth* s, void* stream, const float* phone, int phone_cm, const int64_t* pitch, const float* pitchf, int sid,
                    const float* noise_z, const float* noise_src, int64_t T, float* out, const rvc_synth_taps* taps) {
  RVC_TRY
  RVC_REQUIRE(s && phone && noise_z && out, "null argument");
  check_pitch_args(s, pitch, pitchf, noise_src, 0);
  synth_infer(s->m, (hipStream_t)stream, phone, phone_cm, (const long long*)pitch, pitchf, sid, noise_z, noise_src, (int)T, out, taps);
  check_launch();
  RVC_CATCH
}

int rvc_vc_segment(rvc_hubert* h, rvc_synth* s, void* stream, const float* audio, int64_t L, int version, const int64_t* pitch,
                   const float* pitchf, int sid, float protect, int do_protect, const float* noise_z, const float* noise_src, float* out) {
  RVC_TRY
  RVC_REQUIRE(h && s && audio && noise_z && out, "null argument");
  check_pitch_args(s, pitch, pitchf, noise_src, do_protect);
  hipStream_t st = (hipStream_t)stream;
  const long long Th = hubert_num_frames(L);
  const int D = version == 1 ? 256 : 768;
  RVC_REQUIRE(version == 1 || version == 2, "version must be 1 (256-d) or 2 (768-d)");
  RVC_REQUIRE(D == synth_feat_dim(s->m), "feature width of `version` does not match the synthesizer (v1 models take 256-d, v2 models 768-d features)");
  RVC_REQUIRE(Th > 0 && 2 * Th < (1LL << 30), "segment length out of range");
  const int T = (int)(2 * Th);
  // scratch for the channel-major features lives in two small allocations owned by this call's stream order
  float* fcm = (float*)stream_scratch(st, 1, (size_t)D * Th * sizeof(float));
  float* fup = (float*)stream_scratch(st, 2, (size_t)D * T * sizeof(float));
  hubert_forward(h->m, st, audio, L, version, 0, nullptr, fcm, nullptr);
  feats_prepare(st, fcm, nullptr, pitchf, fup, D, (int)Th, T, protect, do_protect);
  synth_infer(s->m, st, fup, 1, (const long long*)pitch, pitchf, sid, noise_z, noise_src, T, out, nullptr);
  check_launch();
  RVC_CATCH
}

int rvc_vc_segment_feats(rvc_synth* s, void* stream, const float* feats_cm, const float* feats0_cm, int64_t Th, int feat_dim, const int64_t* pitch,
                         const float* pitchf, int sid, float protect, int do_protect, const float* noise_z, const float* noise_src, float* out) {
  RVC_TRY
  RVC_REQUIRE(s && feats_cm && noise_z && out, "null argument");
  check_pitch_args(s, pitch, pitchf, noise_src, do_protect);
  hipStream_t st = (hipStream_t)stream;
  RVC_REQUIRE(feat_dim == synth_feat_dim(s->m), "feat_dim does not match the synthesizer (v1 models take 256-d, v2 models 768-d features)");
  RVC_REQUIRE(Th > 0 && 2 * Th < (1LL << 30), "segment length out of range");
  const int T = (int)(2 * Th);
  float* fup = (float*)stream_scratch(st, 2, (size_t)feat_dim * T * sizeof(float));
  feats_prepare(st, feats_cm, feats0_cm, pitchf, fup, feat_dim, (int)Th, T, protect, do_protect);
  synth_infer(s->m, st, fup, 1, (const long long*)pitch, pitchf, sid, noise_z, noise_src, T, out, nullptr);
  check_launch();
  RVC_CATCH
}

int rvc_resample(void* stream, const float* x, int64_t n_in, const double* taps, int half, int up, int down, float* y, int64_t n_out) {
  RVC_TRY
  RVC_REQUIRE(x && taps && y && n_in > 0 && half >= 0 && up > 0 && down > 0, "bad argument");
  resample((hipStream_t)stream, x, n_in, taps, half, up, down, y, n_out);
  check_launch();
  RVC_CATCH
}

// ------------------------------------------------------------------------------------------------ feature retrieval
struct rvc_index { FeatIndex* m; };
int rvc_index_create(rvc_ctx* ctx, const float* big_npy, int64_t N, int D, rvc_index** out) {
  RVC_TRY
  RVC_REQUIRE(ctx && out, "null argument");
  rvc_index* h = new rvc_index();
  try { h->m = index_create(&ctx->c, big_npy, N, D); } catch (...) { delete h; throw; }
  *out = h;
  RVC_CATCH
}
int rvc_index_destroy(rvc_index* h) { if (h) { index_destroy(h->m); delete h; } return 0; }
int64_t rvc_index_ntotal(const rvc_index* h) { return h ? index_size(h->m) : 0; }
int rvc_index_create_ivf(rvc_ctx* ctx, const float* big_npy, int64_t N, int D, const float* centroids, int nlist, const int32_t* list_of, int nprobe,
                         rvc_index** out) {
  RVC_TRY
  RVC_REQUIRE(ctx && out, "null argument");
  rvc_index* h = new rvc_index();
  try { h->m = index_create_ivf(&ctx->c, big_npy, N, D, centroids, nlist, (const int*)list_of, nprobe); } catch (...) { delete h; throw; }
  *out = h;
  RVC_CATCH
}
int rvc_index_nprobe(const rvc_index* h) { return h && h->m ? index_nprobe(h->m) : 0; }
int rvc_index_search(rvc_index* h, void* stream, const float* feats_cm, int64_t T, int64_t* idx, float* score) {
  RVC_TRY
  RVC_REQUIRE(h && feats_cm && idx && T > 0, "bad argument");
  index_search(h->m, (hipStream_t)stream, feats_cm, (int)T, (long long*)idx, score);
  check_launch();
  RVC_CATCH
}
int rvc_index_blend(rvc_index* h, void* stream, const float* feats_cm, const int64_t* idx, int64_t T, float index_rate, float* out_cm) {
  RVC_TRY
  RVC_REQUIRE(h && feats_cm && idx && out_cm && T > 0, "bad argument");
  index_blend(h->m, (hipStream_t)stream, feats_cm, (const long long*)idx, (int)T, index_rate, out_cm);
  check_launch();
  RVC_CATCH
}

int rvc_preprocess(void* stream, const void* audio, int is_f64, int64_t n, const double* b6, const double* a6, const double* zi5, int t_pad,
                   double* filt, float* padded, double* rms1, int n1, const double* sos18, const double* sos_zi6) {
  RVC_TRY
  RVC_REQUIRE(audio && b6 && a6 && zi5 && filt && n > 3 * 6 + 1 && t_pad >= 0, "bad argument");
  RVC_REQUIRE(a6[0] != 0.0, "a[0] must be non-zero");
  RVC_REQUIRE(rms1 == nullptr || n1 == (int)(n / 8000) + 1, "rms1 must hold n / 8000 + 1 frames");
  hipStream_t st = (hipStream_t)stream;
  RVC_REQUIRE((sos18 == nullptr) == (sos_zi6 == nullptr), "sos and its initial state come together");
  double* scratch = (double*)stream_scratch(st, 2, preprocess_scratch_doubles(n, sos18 != nullptr) * sizeof(double));      // ext | yr, each rounded up to whole blocks
  preprocess(st, audio, is_f64, n, b6, a6, zi5, t_pad, filt, padded, rms1, n1, 16000, 8000, scratch, sos18, sos_zi6);
  check_launch();
  RVC_CATCH
}
int rvc_postprocess(void* stream, float* wav, int64_t N, const double* rms1, int n1, int sr2, float rms_mix_rate, int16_t* out_i16) {
  RVC_TRY
  RVC_REQUIRE(wav && out_i16 && N > 0 && sr2 > 0, "bad argument");
  hipStream_t st = (hipStream_t)stream;
  const int n2 = (int)(N / (sr2 / 2)) + 1;
  float* scratch = (float*)stream_scratch(st, 1, (size_t)(n2 + 4) * sizeof(float));
  postprocess(st, wav, N, rms1, n1, sr2, rms_mix_rate, (short*)out_i16, scratch + 4, (unsigned*)scratch);
  check_launch();
  RVC_CATCH
}

// ------------------------------------------------------------------------------------------------ single ops
int rvc_op_conv1d(void* stream, const float* x, const float* w, const float* bias, const float* res, float* y, int Ci, int Co, int Tin, int k,
                  int stride, int pad, int dil, int groups, int pre_act, float pre_slope, int act, float act_slope, int act_before_res,
                  float out_scale, int accumulate) {
  RVC_TRY
  ConvLayer L;
  conv1d_layer_init(L, w, bias, Co, Ci, k, stride, pad, dil, groups);
  ConvEpilogue e; e.pre_act = pre_act; e.pre_slope = pre_slope; e.act = act; e.act_slope = act_slope; e.act_before_res = act_before_res;
  e.out_scale = out_scale; e.accumulate = accumulate;
  const int Tout = conv1d_out_len(L, Tin);
  e.R = res; e.ldR = Tout;
  try { conv1d_run(L, (hipStream_t)stream, x, Tin, Tin, y, Tout, e); check_launch(); RVC_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream)); }
  catch (...) { conv_layer_free(L); throw; }
  conv_layer_free(L);
  RVC_CATCH
}
int rvc_op_gemm_split(void* stream, const float* x, const float* w, const float* bias, const float* res, float* y, float* ysplit_f32, int Ci, int Co,
                      int T, int act, float act_slope, int act_before_res, float out_scale, int ksplit, int am, int an, int k, int dil) {
  RVC_TRY
  RVC_REQUIRE(x && w && (y || ysplit_f32) && Ci > 0 && Co > 0 && T > 0 && k >= 1 && (k & 1) == 1 && dil >= 1, "bad argument");
  hipStream_t s = (hipStream_t)stream;
  ConvLayer L;
  { ConvBuildScope scope(2); conv1d_layer_init(L, w, bias, Co, Ci, k, 1, (k - 1) / 2 * dil, dil, 1); }
  unsigned char* xs = nullptr; unsigned char* ys = nullptr;
  try {
    RVC_REQUIRE(conv_x3s_eligible(L), "layer not eligible for the split-resident GEMM (Ci % 16 == 0, Ci k >= 64, Co >= 32, pad <= 64)");
    const long long tp = split_image_tp(T);
    RVC_HIP_CHECK(hipMalloc(&xs, split_image_bytes(Ci, T)));
    RVC_HIP_CHECK(hipMemsetAsync(xs, 0, split_image_bytes(Ci, T), s));          // zero margins: the taps' zero padding
    split_image_from_f32(s, x, T, Ci, T, xs, tp);
    ConvEpilogue e; e.act = act; e.act_slope = act_slope; e.act_before_res = act_before_res; e.out_scale = out_scale; e.R = res; e.ldR = T;
    if (ysplit_f32) { RVC_HIP_CHECK(hipMalloc(&ys, split_image_bytes(Co, T))); e.ys_out = ys; e.ys_tp = tp; }
    conv_x3s_force(ksplit, am, an);
    try { conv_x3s_run(L, s, xs, tp, T, y, T, e); } catch (...) { conv_x3s_force(0, 0, 0); throw; }
    conv_x3s_force(0, 0, 0);
    if (ysplit_f32) split_image_to_f32(s, ys, tp, Co, T, ysplit_f32, T);
    check_launch();
    RVC_HIP_CHECK(hipStreamSynchronize(s));
  } catch (...) { if (xs) (void)hipFree(xs); if (ys) (void)hipFree(ys); conv_layer_free(L); throw; }
  (void)hipFree(xs); if (ys) (void)hipFree(ys);
  conv_layer_free(L);
  RVC_CATCH
}
int rvc_op_gemm_split_swapped(void* stream, const float* x, const float* w, float* yt, int Ci, int Co, int T, int row0, int rows) {
  RVC_TRY
  RVC_REQUIRE(x && w && yt && Ci > 0 && Co > 0 && T > 0 && rows > 0, "bad argument");
  hipStream_t s = (hipStream_t)stream;
  ConvLayer L;
  { ConvBuildScope scope(2); conv1d_layer_init(L, w, nullptr, Co, Ci, 1, 1, 0, 1, 1); }
  unsigned char* xs = nullptr; unsigned char* ys = nullptr;
  try {
    const long long tp = split_image_tp(T), vtp = attention_vt_tp(rows);
    const size_t vbytes = attention_vt_bytes(rows, T);
    RVC_HIP_CHECK(hipMalloc(&xs, split_image_bytes(Ci, T)));
    RVC_HIP_CHECK(hipMemsetAsync(xs, 0xff, split_image_bytes(Ci, T), s));       // NaN patterns past T: the tail rows must come out as zeros regardless
    split_image_from_f32(s, x, T, Ci, T, xs, tp);
    RVC_HIP_CHECK(hipMalloc(&ys, vbytes));
    RVC_HIP_CHECK(hipMemsetAsync(ys, 0xff, vbytes, s));
    conv_x3s_run_swapped(L, row0, rows, s, xs, tp, T, ys, vtp);
    attention_vt_clear_tail(s, ys, rows, T);
    split_image_to_f32(s, ys, vtp, (T + 63) / 64 * 64, rows, yt, rows);          // yt [ceil64(T)][rows]
    check_launch();
    RVC_HIP_CHECK(hipStreamSynchronize(s));
  } catch (...) { if (xs) (void)hipFree(xs); if (ys) (void)hipFree(ys); conv_layer_free(L); throw; }
  (void)hipFree(xs); (void)hipFree(ys);
  conv_layer_free(L);
  RVC_CATCH
}
int rvc_op_wn_in_gate_split(void* stream, const float* x, const float* w, const float* bias, const float* g_dev, float* y, int Ci, int H, int T, int k) {
  RVC_TRY
  RVC_REQUIRE(x && w && y && Ci > 0 && H > 0 && (H & 15) == 0 && T > 0 && k >= 1 && (k & 1) == 1, "bad argument");
  hipStream_t s = (hipStream_t)stream;
  // the 2 H rows in the order the gate epilogue wants (wn_gate_row_order)
  std::vector<float> wp((size_t)2 * H * Ci * k), bp((size_t)2 * H, 0.f);
  for (int r = 0; r < 2 * H; ++r) {
    const int src = wn_gate_row_order(r, H);
    std::copy(w + (size_t)src * Ci * k, w + (size_t)(src + 1) * Ci * k, wp.begin() + (size_t)r * Ci * k);
    if (bias) bp[r] = bias[src];
  }
  ConvLayer L;
  { ConvBuildScope scope(2); conv1d_layer_init(L, wp.data(), bp.data(), 2 * H, Ci, k, 1, (k - 1) / 2, 1, 1); }
  unsigned char* xs = nullptr; unsigned char* ys = nullptr;
  try {
    RVC_REQUIRE(conv_x3s_eligible(L), "layer not eligible for the split-resident kernel");
    const long long tp = split_image_tp(T);
    RVC_HIP_CHECK(hipMalloc(&xs, split_image_bytes(Ci, T)));
    RVC_HIP_CHECK(hipMemsetAsync(xs, 0, split_image_bytes(Ci, T), s));
    split_image_from_f32(s, x, T, Ci, T, xs, tp);
    RVC_HIP_CHECK(hipMalloc(&ys, split_image_bytes(H, T)));
    ConvEpilogue e; e.ys_out = ys; e.ys_tp = tp; e.gate_h = H; e.gate_g = g_dev;
    conv_x3s_run(L, s, xs, tp, T, nullptr, T, e);
    split_image_to_f32(s, ys, tp, H, T, y, T);
    check_launch();
    RVC_HIP_CHECK(hipStreamSynchronize(s));
  } catch (...) { if (xs) (void)hipFree(xs); if (ys) (void)hipFree(ys); conv_layer_free(L); throw; }
  (void)hipFree(xs); (void)hipFree(ys);
  conv_layer_free(L);
  RVC_CATCH
}
int rvc_op_gemm_split_qkv(void* stream, const float* x, const float* w, const float* bias, float* y_img_f32, float* yt, int Ci, int Co, int T, int vt_row0) {
  RVC_TRY
  RVC_REQUIRE(x && w && y_img_f32 && yt && Ci > 0 && Co > 0 && T > 0 && vt_row0 > 0 && vt_row0 < Co, "bad argument");
  hipStream_t s = (hipStream_t)stream;
  ConvLayer L;
  { ConvBuildScope scope(2); conv1d_layer_init(L, w, bias, Co, Ci, 1, 1, 0, 1, 1); }
  unsigned char* xs = nullptr; unsigned char* ys = nullptr; unsigned char* vt = nullptr;
  const int rows = Co - vt_row0;
  try {
    RVC_REQUIRE(conv_x3s_eligible(L), "layer not eligible for the split-resident kernel");
    const long long tp = split_image_tp(T), vtp = attention_vt_tp(rows);
    const size_t vbytes = attention_vt_bytes(rows, T);
    RVC_HIP_CHECK(hipMalloc(&xs, split_image_bytes(Ci, T)));
    RVC_HIP_CHECK(hipMemsetAsync(xs, 0xff, split_image_bytes(Ci, T), s));       // NaN patterns past T: the transposed rows past T must come out as zeros regardless
    split_image_from_f32(s, x, T, Ci, T, xs, tp);
    RVC_HIP_CHECK(hipMalloc(&ys, split_image_bytes(vt_row0, T)));
    RVC_HIP_CHECK(hipMalloc(&vt, vbytes));
    RVC_HIP_CHECK(hipMemsetAsync(vt, 0xff, vbytes, s));
    ConvEpilogue e; e.ys_out = ys; e.ys_tp = tp; e.vt_out = vt; e.vt_tp = vtp; e.vt_row0 = vt_row0;
    conv_x3s_run(L, s, xs, tp, T, nullptr, T, e);
    attention_vt_clear_tail(s, vt, rows, T);
    split_image_to_f32(s, ys, tp, vt_row0, T, y_img_f32, T);                      // y_img_f32 [vt_row0][T]: the rows below vt_row0, from their image
    split_image_to_f32(s, vt, vtp, (T + 63) / 64 * 64, rows, yt, rows);            // yt [ceil64(T)][rows]: the rows from vt_row0 on, transposed
    check_launch();
    RVC_HIP_CHECK(hipStreamSynchronize(s));
  } catch (...) { if (xs) (void)hipFree(xs); if (ys) (void)hipFree(ys); if (vt) (void)hipFree(vt); conv_layer_free(L); throw; }
  (void)hipFree(xs); (void)hipFree(ys); (void)hipFree(vt);
  conv_layer_free(L);
  RVC_CATCH
}
int rvc_op_conv2d3x3_plus_1x1(void* stream, const float* x1, const float* w1, const float* x2, const float* w2, float* y, float* y_img_f32, int Ci1, int Ci2, int Co, int H, int W,
                              int ksplit) {
  RVC_TRY
  RVC_REQUIRE(x1 && w1 && x2 && w2 && y && Ci1 > 0 && Ci2 > 0 && Co > 0 && H > 0 && W > 0, "bad argument");
  hipStream_t s = (hipStream_t)stream;
  ConvLayer L1, L2;
  { ConvBuildScope scope(2); conv2d3x3_layer_init(L1, w1, nullptr, Co, Ci1); conv1d_layer_init(L2, w2, nullptr, Co, Ci2, 1, 1, 0, 1, 1); }
  unsigned char* img = nullptr; unsigned char* yimg = nullptr; float* yp = nullptr;
  try {
    RVC_REQUIRE(conv_x3s_eligible(L1) && conv_x3s_eligible(L2), "layers not eligible for the split-resident GEMM (channels % 16, Co >= 32)");
    conv_layer_append_x3(L1, L2);
    SplitGeom g = split_geom_2d(W);
    const long long TP = (long long)H * (W + 2), tp = ((long long)g.margin + TP + std::max(704, g.margin) + 63) & ~63LL;
    const size_t b1 = (size_t)(Ci1 / 16) * 4 * (size_t)tp * 16, b2 = (size_t)(Ci2 / 16) * 4 * (size_t)tp * 16, bo = (size_t)((Co + 15) / 16) * 4 * (size_t)tp * 16;
    RVC_HIP_CHECK(hipMalloc(&img, b1 + b2)); RVC_HIP_CHECK(hipMemsetAsync(img, 0, b1 + b2, s));      // zero margins: the vertical zero padding
    RVC_HIP_CHECK(hipMalloc(&yp, (size_t)Co * TP * sizeof(float)));
    if (y_img_f32) { RVC_HIP_CHECK(hipMalloc(&yimg, bo)); RVC_HIP_CHECK(hipMemsetAsync(yimg, 0xff, bo, s)); }
    pad2d_split(s, x1, (long long)H * W, Ci1, H, W, nullptr, 0, img, tp, g.margin);
    pad2d_split(s, x2, (long long)H * W, Ci2, H, W, nullptr, 0, img + b1, tp, g.margin);
    g.seg2_off = (long long)b1;
    ConvEpilogue e;
    if (yimg) { e.ys_out = yimg; e.ys_tp = tp; }
    conv_x3s_force(ksplit, 0, 0);
    try { conv_x3s_run(L1, s, img, tp, (int)TP, yp, TP, e, &g); } catch (...) { conv_x3s_force(0, 0, 0); throw; }
    conv_x3s_force(0, 0, 0);
    unpad2d(s, yp, TP, Co, H, W, y, (long long)H * W);
    if (yimg) {                                                 // the raw image of the output, read back through the padded layout
      split_image_to_f32(s, yimg + (size_t)(g.margin - kSplitMargin) * 16, tp, Co, (int)TP, yp, TP);
      unpad2d(s, yp, TP, Co, H, W, y_img_f32, (long long)H * W);
    }
    check_launch();
    RVC_HIP_CHECK(hipStreamSynchronize(s));
  } catch (...) { if (img) (void)hipFree(img); if (yimg) (void)hipFree(yimg); if (yp) (void)hipFree(yp); conv_layer_free(L1); conv_layer_free(L2); throw; }
  (void)hipFree(img); if (yimg) (void)hipFree(yimg); (void)hipFree(yp);
  conv_layer_free(L1); conv_layer_free(L2);
  RVC_CATCH
}
int rvc_op_gemm_split_swapped_res(void* stream, const float* x, const float* w, const float* res, float* y, int Ci, int Co, int T, int ld, int off) {
  RVC_TRY
  RVC_REQUIRE(x && w && res && y && Ci > 0 && Co > 0 && T > 0 && ld >= Co + off && off >= 0, "bad argument");
  hipStream_t s = (hipStream_t)stream;
  ConvLayer L;
  { ConvBuildScope scope(2); conv1d_layer_init(L, w, nullptr, Co, Ci, 1, 1, 0, 1, 1); }
  unsigned char* xs = nullptr;
  try {
    const long long tp = split_image_tp(T);
    RVC_HIP_CHECK(hipMalloc(&xs, split_image_bytes(Ci, T)));
    RVC_HIP_CHECK(hipMemsetAsync(xs, 0, split_image_bytes(Ci, T), s));
    split_image_from_f32(s, x, T, Ci, T, xs, tp);
    conv_x3s_run_swapped(L, 0, Co, s, xs, tp, T, nullptr, 0, y + off, ld, res + off, ld);      // y[t][off + j] = sum_c x[c][t] w[j][c] + res[t][off + j]
    check_launch();
    RVC_HIP_CHECK(hipStreamSynchronize(s));
  } catch (...) { if (xs) (void)hipFree(xs); conv_layer_free(L); throw; }
  (void)hipFree(xs);
  conv_layer_free(L);
  RVC_CATCH
}
int rvc_op_attention_split(void* stream, const float* q, const float* k, const float* v, const float* bv, float* out, float* out_img_f32, int heads, int T) {
  RVC_TRY
  RVC_REQUIRE(q && k && v && (out || out_img_f32) && heads > 0 && T > 0, "bad argument");
  hipStream_t s = (hipStream_t)stream;
  const int C = heads * 64;
  unsigned char* qk = nullptr; unsigned char* vt = nullptr; unsigned char* oi = nullptr; float* vtf = nullptr;
  try {
    const long long tp = split_image_tp(T), vtp = attention_vt_tp(C);
    RVC_HIP_CHECK(hipMalloc(&qk, split_image_bytes(2 * C, T)));
    RVC_HIP_CHECK(hipMemsetAsync(qk, 0xff, split_image_bytes(2 * C, T), s));     // rows past T hold NaN patterns: masked keys must not leak
    split_image_from_f32(s, q, T, C, T, qk, tp);
    split_image_from_f32(s, k, T, C, T, qk + split_image_bytes(C, T), tp);
    RVC_HIP_CHECK(hipMalloc(&vtf, (size_t)T * C * 4));
    transpose((hipStream_t)s, v, vtf, C, T, T, C, 1, 0, 0);
    RVC_HIP_CHECK(hipMalloc(&vt, attention_vt_bytes(C, T)));
    RVC_HIP_CHECK(hipMemsetAsync(vt, 0xff, attention_vt_bytes(C, T), s));
    split_image_from_f32(s, vtf, C, T, C, vt, vtp);                              // "channels" = keys, "positions" = model channels
    attention_vt_clear_tail(s, vt, C, T);
    if (out_img_f32) { RVC_HIP_CHECK(hipMalloc(&oi, split_image_bytes(C, T))); }
    attention_split(s, qk, tp, 2 * C, 0, C / 16, vt, heads, 64, T, 1.f, bv, out, T, oi, tp);
    if (out_img_f32) split_image_to_f32(s, oi, tp, C, T, out_img_f32, T);
    check_launch();
    RVC_HIP_CHECK(hipStreamSynchronize(s));
  } catch (...) { (void)hipFree(qk); (void)hipFree(vt); (void)hipFree(oi); (void)hipFree(vtf); throw; }
  (void)hipFree(qk); (void)hipFree(vt); (void)hipFree(oi); (void)hipFree(vtf);
  RVC_CATCH
}
int rvc_op_attention_split_rel(void* stream, const float* q, const float* k, const float* v, const float* bv, const float* ek_host, const float* ev_host,
                               float* out, float* out_img_f32, int heads, int T, int kz) {
  RVC_TRY
  RVC_REQUIRE(q && k && v && ek_host && ev_host && (out || out_img_f32) && heads > 0 && T > 0, "bad argument");
  hipStream_t s = (hipStream_t)stream;
  const int C = heads * 96;
  unsigned char* qk = nullptr; unsigned char* vt = nullptr; unsigned char* oi = nullptr; float* vtf = nullptr; unsigned char* tab = nullptr;
  try {
    std::vector<uint16_t> eki, evi;
    attention_rel_images(ek_host, ev_host, 96, 10, eki, evi);
    RVC_HIP_CHECK(hipMalloc(&tab, (eki.size() + evi.size()) * 2));
    RVC_HIP_CHECK(hipMemcpy(tab, eki.data(), eki.size() * 2, hipMemcpyHostToDevice));
    RVC_HIP_CHECK(hipMemcpy(tab + eki.size() * 2, evi.data(), evi.size() * 2, hipMemcpyHostToDevice));
    const long long tp = split_image_tp(T), vtp = attention_vt_tp(C);
    RVC_HIP_CHECK(hipMalloc(&qk, split_image_bytes(2 * C, T)));
    RVC_HIP_CHECK(hipMemsetAsync(qk, 0xff, split_image_bytes(2 * C, T), s));
    split_image_from_f32(s, q, T, C, T, qk, tp);
    split_image_from_f32(s, k, T, C, T, qk + split_image_bytes(C, T), tp);
    RVC_HIP_CHECK(hipMalloc(&vtf, (size_t)T * C * 4));
    transpose((hipStream_t)s, v, vtf, C, T, T, C, 1, 0, 0);
    RVC_HIP_CHECK(hipMalloc(&vt, attention_vt_bytes(C, T)));
    RVC_HIP_CHECK(hipMemsetAsync(vt, 0xff, attention_vt_bytes(C, T), s));
    split_image_from_f32(s, vtf, C, T, C, vt, vtp);
    attention_vt_clear_tail(s, vt, C, T);
    if (out_img_f32) { RVC_HIP_CHECK(hipMalloc(&oi, split_image_bytes(C, T))); }
    attention_split_force_kz(kz);
    try { attention_split(s, qk, tp, 2 * C, 0, C / 16, vt, heads, 96, T, 1.f, bv, out, T, oi, tp, 10, tab, tab + eki.size() * 2); } catch (...) { attention_split_force_kz(0); throw; }
    attention_split_force_kz(0);
    if (out_img_f32) split_image_to_f32(s, oi, tp, C, T, out_img_f32, T);
    check_launch();
    RVC_HIP_CHECK(hipStreamSynchronize(s));
  } catch (...) { (void)hipFree(qk); (void)hipFree(vt); (void)hipFree(oi); (void)hipFree(vtf); (void)hipFree(tab); throw; }
  (void)hipFree(qk); (void)hipFree(vt); (void)hipFree(oi); (void)hipFree(vtf); (void)hipFree(tab);
  RVC_CATCH
}
int rvc_op_cbr2_small(void* stream, const float* x, const float* w1, const float* b1, const float* w2, const float* b2, float* y, int C, int H, int W) {
  RVC_TRY
  RVC_REQUIRE(x && w1 && b1 && w2 && b2 && y && (C == 16 || C == 32) && H > 0 && W > 0, "bad argument");
  hipStream_t s = (hipStream_t)stream;
  ConvLayer c1, c2;
  { ConvBuildScope scope(2); conv2d3x3_layer_init(c1, w1, b1, C, C); conv2d3x3_layer_init(c2, w2, b2, C, C); }
  try {
    cbr2_small_run(c1, c2, s, x, H, W, y);
    check_launch();
    RVC_HIP_CHECK(hipStreamSynchronize(s));
  } catch (...) { conv_layer_free(c1); conv_layer_free(c2); throw; }
  conv_layer_free(c1); conv_layer_free(c2);
  RVC_CATCH
}
int rvc_op_conv3_small(void* stream, const float* x, const float* w, const float* b, const float* res, float* y, float* y2, int Ci, int Co, int H, int W, int split_row,
                       int relu_rows) {
  RVC_TRY
  RVC_REQUIRE(x && w && b && y && Ci > 0 && Co > 0 && H > 0 && W > 0, "bad argument");
  hipStream_t s = (hipStream_t)stream;
  ConvLayer L;
  { ConvBuildScope scope(2); conv2d3x3_layer_init(L, w, b, Co, Ci); }
  try {
    conv3_small_run(L, s, x, H, W, y, y2, split_row, relu_rows, res);
    check_launch();
    RVC_HIP_CHECK(hipStreamSynchronize(s));
  } catch (...) { conv_layer_free(L); throw; }
  conv_layer_free(L);
  RVC_CATCH
}
int rvc_op_conv1d_split(void* stream, const float* x, const float* w, const float* bias, const float* res, float* y, int Ci, int Co, int T, int k, int pad,
                        int dil, int groups, int act, int act_before_res) {
  RVC_TRY
  RVC_REQUIRE(x && w && y && Ci > 0 && Co > 0 && T > 0 && k >= 1 && groups >= 1, "bad argument");
  hipStream_t s = (hipStream_t)stream;
  ConvLayer L;
  { ConvBuildScope scope(2); conv1d_layer_init(L, w, bias, Co, Ci, k, 1, pad, dil, groups); }
  unsigned char* xs = nullptr;
  try {
    RVC_REQUIRE(conv_x3s_eligible(L), "layer not eligible for the split-resident kernel");
    const long long tp = split_image_tp(T);
    RVC_HIP_CHECK(hipMalloc(&xs, split_image_bytes(Ci, T)));
    RVC_HIP_CHECK(hipMemsetAsync(xs, 0, split_image_bytes(Ci, T), s));
    split_image_from_f32(s, x, T, Ci, T, xs, tp);
    ConvEpilogue e; e.act = act; e.act_slope = 0.1f; e.act_before_res = act_before_res; e.R = res; e.ldR = T;
    conv_x3s_run(L, s, xs, tp, T, y, T, e);
    check_launch();
    RVC_HIP_CHECK(hipStreamSynchronize(s));
  } catch (...) { if (xs) (void)hipFree(xs); conv_layer_free(L); throw; }
  (void)hipFree(xs);
  conv_layer_free(L);
  RVC_CATCH
}
int rvc_op_conv1d_s2_split(void* stream, const float* x, const float* w, const float* bias, float* y, float* y_img_f32, int Ci, int Co, int T, int k, int act) {
  RVC_TRY
  RVC_REQUIRE(x && w && y && Ci > 0 && Co > 0 && k >= 2 && T >= k, "bad argument");
  hipStream_t s = (hipStream_t)stream;
  ConvLayer L;
  { ConvBuildScope scope(2); conv1d_layer_init(L, w, bias, Co, Ci, k, 2, 0, 1, 1); }
  unsigned char* xs = nullptr; unsigned char* ys = nullptr;
  try {
    RVC_REQUIRE(conv_x3s_s2_eligible(L), "layer not eligible for the stride-2 path of the split-resident kernel (Ci, Co multiples of 16)");
    const int Tout = (T - k) / 2 + 1;
    const SplitGeom g = split_geom_s2(k, T);
    RVC_HIP_CHECK(hipMalloc(&xs, split_s2_bytes(Ci, T)));
    RVC_HIP_CHECK(hipMemsetAsync(xs, 0xff, split_s2_bytes(Ci, T), s));      // (NaN patterns wherever the producer does not write: what a consumer reads there must not reach a stored column)
    split_image_deint_from_f32(s, x, T, Ci, T, xs, split_s2_tp(T), g.s2_h);
    ConvEpilogue e; e.act = act; e.act_slope = 0.1f;
    if (y_img_f32) {
      RVC_HIP_CHECK(hipMalloc(&ys, split_s2_bytes(Co, Tout)));
      e.ys_out = ys; e.ys_tp = split_s2_tp(Tout); e.ys_deint_h = split_s2_h(Tout);
    }
    conv_x3s_run(L, s, xs, split_s2_tp(T), Tout, y, Tout, e, &g);
    if (y_img_f32) split_image_deint_to_f32(s, ys, split_s2_tp(Tout), split_s2_h(Tout), Co, Tout, y_img_f32, Tout);
    check_launch();
    RVC_HIP_CHECK(hipStreamSynchronize(s));
  } catch (...) { if (xs) (void)hipFree(xs); if (ys) (void)hipFree(ys); conv_layer_free(L); throw; }
  (void)hipFree(xs); if (ys) (void)hipFree(ys);
  conv_layer_free(L);
  RVC_CATCH
}
int rvc_op_conv2d_split(void* stream, const float* x, const float* w, const float* bias, const float* res, float* y, float* ysplit_f32, int Ci, int Co,
                        int H, int W, int act, int act_before_res, int ksplit, int am, int an) {
  RVC_TRY
  RVC_REQUIRE(x && w && y && Ci > 0 && Co > 0 && H > 0 && W >= 2 && (W & 1) == 0, "bad argument");
  hipStream_t s = (hipStream_t)stream;
  ConvLayer L;
  { ConvBuildScope scope(2); conv2d3x3_layer_init(L, w, bias, Co, Ci); }
  unsigned char* xs = nullptr; unsigned char* ys = nullptr; float* rp = nullptr; float* yp = nullptr; float* t = nullptr;
  auto cleanup = [&]() { if (xs) (void)hipFree(xs); if (ys) (void)hipFree(ys); if (rp) (void)hipFree(rp); if (yp) (void)hipFree(yp); if (t) (void)hipFree(t); conv_layer_free(L); };
  try {
    RVC_REQUIRE(conv_x3s_eligible(L), "layer not eligible for the split-resident kernel (Ci % 16 == 0)");
    const SplitGeom g = split_geom_2d(W);
    const int T = H * (W + 2);
    const long long tp = g.margin + T + 704 + 64;
    const size_t ib = (size_t)(Ci / 16) * 4 * tp * 16, ob = (size_t)((Co + 15) / 16) * 4 * tp * 16;
    RVC_HIP_CHECK(hipMalloc(&xs, ib)); RVC_HIP_CHECK(hipMemsetAsync(xs, 0, ib, s));
    RVC_HIP_CHECK(hipMalloc(&yp, (size_t)Co * T * 4));
    pad2d_split(s, x, (long long)H * W, Ci, H, W, nullptr, 0, xs, tp, g.margin);
    ConvEpilogue e; e.act = act; e.act_slope = 0.1f; e.act_before_res = act_before_res;
    if (res) { RVC_HIP_CHECK(hipMalloc(&rp, (size_t)Co * T * 4)); pad2d_split(s, res, (long long)H * W, Co, H, W, rp, T, nullptr, 0, 0); e.R = rp; e.ldR = T; }
    if (ysplit_f32) { RVC_HIP_CHECK(hipMalloc(&ys, ob)); RVC_HIP_CHECK(hipMemsetAsync(ys, 0, ob, s)); e.ys_out = ys; e.ys_tp = tp; }
    conv_x3s_force(ksplit, am, an);
    try { conv_x3s_run(L, s, xs, tp, T, yp, T, e, &g); } catch (...) { conv_x3s_force(0, 0, 0); throw; }
    conv_x3s_force(0, 0, 0);
    unpad2d(s, yp, T, Co, H, W, y, (long long)H * W);
    if (ysplit_f32) {
      RVC_REQUIRE((Co & 15) == 0, "split output needs Co % 16 == 0");
      RVC_HIP_CHECK(hipMalloc(&t, (size_t)Co * T * 4));
      // (the image's rows start at its own margin: hand the reader the plane origin shifted so that its fixed 64-row margin lands on position 0)
      split_image_to_f32(s, ys + (size_t)(g.margin - kSplitMargin) * 16, tp, Co, T, t, T);
      unpad2d(s, t, T, Co, H, W, ysplit_f32, (long long)H * W);
    }
    check_launch();
    RVC_HIP_CHECK(hipStreamSynchronize(s));
  } catch (...) { cleanup(); throw; }
  cleanup();
  RVC_CATCH
}
int rvc_op_conv_transpose1d(void* stream, const float* x, const float* w, const float* bias, float* y, int Ci, int Co, int Tin, int k, int u,
                            int pad, int pre_act, float pre_slope, int accumulate) {
  RVC_TRY
  ConvLayer L;
  tconv1d_layer_init(L, w, bias, Ci, Co, k, u, pad);
  ConvEpilogue e; e.pre_act = pre_act; e.pre_slope = pre_slope; e.accumulate = accumulate;
  const int Tout = conv1d_out_len(L, Tin);
  try { conv1d_run(L, (hipStream_t)stream, x, Tin, Tin, y, Tout, e); check_launch(); RVC_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream)); }
  catch (...) { conv_layer_free(L); throw; }
  conv_layer_free(L);
  RVC_CATCH
}
int rvc_op_conv2d3x3(void* stream, const float* x, const float* w, const float* bias, const float* res, float* y, int Ci, int Co, int H, int W,
                     int relu) {
  RVC_TRY
  ConvLayer L;
  conv2d3x3_layer_init(L, w, bias, Co, Ci);
  ConvEpilogue e; e.act = relu ? ACT_RELU : ACT_NONE; e.act_before_res = 1; e.R = res; e.ldR = (long long)H * W;
  try { conv2d_run(L, (hipStream_t)stream, x, (long long)H * W, H, W, y, (long long)H * W, e); check_launch(); RVC_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream)); }
  catch (...) { conv_layer_free(L); throw; }
  conv_layer_free(L);
  RVC_CATCH
}
int rvc_op_conv_transpose2d(void* stream, const float* x, const float* w, const float* bias, float* y, int Ci, int Co, int H, int W, int relu) {
  RVC_TRY
  ConvLayer L;
  tconv2d_layer_init(L, w, bias, Ci, Co);
  ConvEpilogue e; e.act = relu ? ACT_RELU : ACT_NONE;
  try { conv2d_run(L, (hipStream_t)stream, x, (long long)H * W, H, W, y, 4LL * H * W, e); check_launch(); RVC_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream)); }
  catch (...) { conv_layer_free(L); throw; }
  conv_layer_free(L);
  RVC_CATCH
}
int rvc_op_gemm_tn(void* stream, const float* a, const float* b, float* y, int M, int N, int K, int batch) {
  RVC_TRY
  ConvEpilogue e;
  gemm_tn_run((hipStream_t)stream, a, M, (long long)K * M, b, N, (long long)K * N, y, N, (long long)M * N, M, N, K, batch, nullptr, 0, e);
  check_launch();
  RVC_CATCH
}
int rvc_conv1d_plan_create(const float* w, const float* bias, int Ci, int Co, int k, int stride, int pad, int dil, int groups,
                           rvc_conv1d_plan** out) {
  RVC_TRY
  rvc_conv1d_plan* p = new rvc_conv1d_plan();
  try { conv1d_layer_init(p->L, w, bias, Co, Ci, k, stride, pad, dil, groups); } catch (...) { delete p; throw; }
  *out = p;
  RVC_CATCH
}
int rvc_conv1d_plan_run(rvc_conv1d_plan* p, void* stream, const float* x, int Tin, const float* res, float* y, int pre_act, float pre_slope,
                        int act, float act_slope) {
  RVC_TRY
  ConvEpilogue e; e.pre_act = pre_act; e.pre_slope = pre_slope; e.act = act; e.act_slope = act_slope;
  const int Tout = conv1d_out_len(p->L, Tin);
  e.R = res; e.ldR = Tout;
  conv1d_run(p->L, (hipStream_t)stream, x, Tin, Tin, y, Tout, e);
  check_launch();
  RVC_CATCH
}
int rvc_conv1d_plan_pair_run(rvc_conv1d_plan* c1, rvc_conv1d_plan* c2, void* stream, const float* x, int T, float* y, float out_scale,
                             int accumulate) {
  RVC_TRY
  RVC_REQUIRE(c1 && c2 && x && y, "null argument");
  ConvEpilogue e; e.pre_act = ACT_LRELU; e.pre_slope = 0.1f; e.R = x; e.ldR = T; e.out_scale = out_scale; e.accumulate = accumulate;
  RVC_REQUIRE(conv_x3_pair_try(c1->L, c2->L, (hipStream_t)stream, x, T, T, y, T, e),
              "this pair of layers is not eligible for the fused ResBlock kernel (needs bf16x3 images, C = 32, equal odd k, long T)");
  check_launch();
  RVC_CATCH
}
int rvc_conv1d_plan_resblock_run(rvc_conv1d_plan* const* plans6, void* stream, const float* x, int T, float* y, float out_scale, int accumulate,
                                 int* ran_out, const float* noise_src, const float* noise_w, const float* noise_b) {
  RVC_TRY
  RVC_REQUIRE(plans6 && x && y && ran_out, "null argument");
  const ConvLayer* c1[3]; const ConvLayer* c2[3];
  for (int i = 0; i < 3; ++i) {
    RVC_REQUIRE(plans6[2 * i] && plans6[2 * i + 1], "null plan");
    c1[i] = &plans6[2 * i]->L; c2[i] = &plans6[2 * i + 1]->L;
  }
  *ran_out = conv_rb3_try(c1, c2, (hipStream_t)stream, x, T, T, y, T, 0.1f, out_scale, accumulate, false, noise_src, noise_w, noise_b) ? 1 : 0;
  check_launch();
  RVC_CATCH
}
int rvc_conv1d_plan_pair_split_run(rvc_conv1d_plan* c1, rvc_conv1d_plan* c2, void* stream, const float* x, int T, float* y, float out_scale,
                                   int accumulate) {
  RVC_TRY
  RVC_REQUIRE(c1 && c2 && x && y, "null argument");
  RVC_REQUIRE(conv1d_split_eligible(c1->L, T, SPLIT_PRODUCER) && conv1d_split_eligible(c2->L, T, SPLIT_CONSUMER), "layers not eligible for split-resident tensors at this length");
  hipStream_t s = (hipStream_t)stream;
  unsigned char* img = (unsigned char*)stream_scratch(s, 5, split_image_bytes(c1->L.Co, T));
  ConvEpilogue E1; E1.pre_act = ACT_LRELU; E1.pre_slope = 0.1f; E1.ys_out = img; E1.ys_tp = split_image_tp(T); E1.ys_slope = 0.1f;
  // the arithmetic the generator would use for this pair at this length: fp16x2 on the persistent kernel where eligible (rvc_set_pair_arithmetic), else bf16x3
  E1.h2 = conv1d_pair_h2_eligible(c1->L, c2->L, T) ? 1 : 0;
#ifdef RVC_EXPERIMENTS
  if (exp_int("RVC_EXP_PAIR_ZERO", 0)) RVC_HIP_CHECK(hipMemsetAsync(img, 0, split_image_bytes(c1->L.Co, T), s));
#endif
  conv1d_run(c1->L, s, x, T, T, nullptr, T, E1);
#ifdef RVC_EXPERIMENTS
  if (exp_int("RVC_EXP_PAIR_SYNC", 0)) RVC_HIP_CHECK(hipDeviceSynchronize());
  if (exp_int("RVC_EXP_PAIR_C1ONLY", 0)) { check_launch(); return 0; }
#endif
  ConvEpilogue E2; E2.R = x; E2.ldR = T; E2.out_scale = out_scale; E2.accumulate = accumulate; E2.xs_in = img; E2.xs_tp = E1.ys_tp; E2.h2 = E1.h2;
  conv1d_run(c2->L, s, nullptr, T, T, y, T, E2);
  check_launch();
  RVC_CATCH
}
int rvc_conv1d_plan_pair_arithmetic(rvc_conv1d_plan* c1, rvc_conv1d_plan* c2, int T) {
  if (!c1 || !c2) return -1;
  try {
    if (conv1d_pair_h2_eligible(c1->L, c2->L, T)) return 1;
    ConvEpilogue e; e.pre_act = ACT_LRELU; e.pre_slope = 0.1f; const float* dummy = reinterpret_cast<const float*>(c1->L.Wd_); e.R = dummy; e.ldR = T;
    return conv_x3_pair_try(c1->L, c2->L, nullptr, dummy, T, T, nullptr, T, e, true) ? 1 : 0;      // (the fused pair of the 32-channel stage: dry run)
  } catch (...) { return -1; }
}
int rvc_conv1d_plan_destroy(rvc_conv1d_plan* p) { if (p) { conv_layer_free(p->L); delete p; } return 0; }
int rvc_op_attention(void* stream, const float* q, const float* k, const float* v_rm, const float* bv, float* out, int heads, int T) {
  RVC_TRY
  attention_fused((hipStream_t)stream, q, k, T, v_rm, (long long)heads * 64, bv, out, T, heads, 64, T);
  check_launch();
  RVC_CATCH
}
int rvc_op_attention_rel(void* stream, const float* q, const float* k, const float* v_rm, const float* bv, const float* rel, float* pb,
                         float* out, int heads, int T, const float* ek, const float* ev) {
  RVC_TRY
  attention_rel_fused((hipStream_t)stream, q, k, T, v_rm, (long long)heads * 96, bv, rel, pb, 10, out, T, heads, 96, T, ek, ev);
  check_launch();
  RVC_CATCH
}
int rvc_op_layernorm_c(void* stream, const float* x, const float* res, const float* gamma, const float* beta, float* y, int C, int T) {
  RVC_TRY
  layernorm_c((hipStream_t)stream, x, res, gamma, beta, y, C, T, T, 1e-5f);
  check_launch();
  RVC_CATCH
}
int rvc_op_layernorm_c_split(void* stream, const float* x, const float* gamma, const float* beta, float* y, float* y_img_f32, int C, int T) {
  RVC_TRY
  RVC_REQUIRE(x && gamma && beta && y_img_f32 && C > 0 && (C & 15) == 0 && T > 0, "bad argument");
  hipStream_t s = (hipStream_t)stream;
  unsigned char* img = nullptr;
  try {
    const long long tp = split_image_tp(T);
    RVC_HIP_CHECK(hipMalloc(&img, split_image_bytes(C, T)));
    layernorm_c_split(s, x, gamma, beta, y, img, tp, kSplitMargin, C, T, T, 1e-5f);
    split_image_to_f32(s, img, tp, C, T, y_img_f32, T);
    check_launch();
    RVC_HIP_CHECK(hipStreamSynchronize(s));
  } catch (...) { (void)hipFree(img); throw; }
  (void)hipFree(img);
  RVC_CATCH
}
int rvc_op_sine_source(void* stream, const float* f0, const float* noise, float* har, float* sine, int T, int upp, float sr, float lw, float lb,
                       float* rad_out, float* tmp_out, float* phase_out) {
  RVC_TRY
  hipStream_t st = (hipStream_t)stream;
  const long long N = (long long)T * upp;
  RVC_REQUIRE(T > 0 && upp > 0, "bad argument");
  const size_t nb = (size_t)((N + 1023) / 1024);
  const size_t tb = ((size_t)T * sizeof(float) + 255) & ~size_t(255);
  char* scr = (char*)stream_scratch(st, 2, 2 * tb + nb * sizeof(double));        // temporaries from the stream's scratch: nothing to leak on failure
  float* rad = (float*)scr; float* tmp = (float*)(scr + tb); double* bsum = (double*)(scr + 2 * tb);
  sine_source(st, f0, noise, har, sine, rad, tmp, bsum, T, upp, sr, lw, lb, phase_out);
  if (rad_out) RVC_HIP_CHECK(hipMemcpyAsync(rad_out, rad, T * sizeof(float), hipMemcpyDeviceToDevice, st));
  if (tmp_out) RVC_HIP_CHECK(hipMemcpyAsync(tmp_out, tmp, T * sizeof(float), hipMemcpyDeviceToDevice, st));
  RVC_HIP_CHECK(hipStreamSynchronize(st));
  check_launch();
  RVC_CATCH
}

int rvc_prof_enable(int on) { RVC_TRY conv_prof_enable(on != 0); RVC_CATCH }
int rvc_set_conv_precision(int mode) {
  RVC_TRY
  RVC_REQUIRE(mode >= 0 && mode <= 2, "precision mode must be 0, 1 or 2");
  conv_set_precision(mode);
  RVC_CATCH
}
int rvc_set_pair_arithmetic(int mode) {
  RVC_TRY
  RVC_REQUIRE(mode == 0 || mode == 1, "pair arithmetic must be 0 (bf16x3) or 1 (fp16x2)");
  conv_set_pair_arithmetic(mode);
  RVC_CATCH
}
int rvc_get_pair_arithmetic(void) { return conv_set_pair_arithmetic(-1); }
int rvc_prof_collect(double* ms, double* flops, int64_t* launches) {
  RVC_TRY
  static_assert(RVC_PROF_CFGS == kProfCfgs, "profiling table size");
  long long l[RVC_PROF_CFGS];
  conv_prof_collect(ms, flops, l);
  for (int i = 0; i < RVC_PROF_CFGS; ++i) launches[i] = l[i];
  RVC_CATCH
}
int rvc_prof_collect_ex(double* out, double ridge_fp32, double ridge_x3) { RVC_TRY conv_prof_collect_ex(out, ridge_fp32, ridge_x3); RVC_CATCH }
const char* rvc_prof_cfg_name(int i) { return conv_prof_cfg_name(i); }
int rvc_prof_dump_csv(const char* path) { RVC_TRY RVC_REQUIRE(path && conv_prof_dump_csv(path) >= 0, "cannot write the launch table"); RVC_CATCH }
#ifdef RVC_EXPERIMENTS
int rvc_debug_read_scratch(void* stream, int slot, void* host_dst, size_t bytes) {
  RVC_TRY
  void* p = stream_scratch((hipStream_t)stream, slot, bytes);
  RVC_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
  RVC_HIP_CHECK(hipMemcpy(host_dst, p, bytes, hipMemcpyDeviceToHost));
  RVC_CATCH
}
int rvc_debug_conv_timing(uint64_t* out8, int reset) { RVC_TRY conv_timing_read((unsigned long long*)out8, reset != 0); RVC_CATCH }
int rvc_debug_set_x3s_mode(int mode) { conv_x3s_set_mode(mode < 0 || mode > 2 ? 0 : mode); return 0; }
int rvc_debug_x3p_check(void) { const int a = conv_x3p_check_read(), b = conv_x3q_check_read(); return a < 0 ? a : a + (b > 0 ? b : 0); }
int rvc_debug_gemm_split_bench(void* stream, int Ci, int Co, int T, int ksplit, int am, int an, int split_out, int reps, float* us_out, int w2d, int nlayers) {
  RVC_TRY
  RVC_REQUIRE(us_out && reps > 0 && Ci > 0 && Co > 0 && T > 0 && nlayers >= 1 && nlayers <= 64, "bad argument");
  hipStream_t s = (hipStream_t)stream;
  const int kt = w2d > 0 ? 9 : 1;
  std::vector<float> w((size_t)Co * Ci * kt), b((size_t)Co, 0.01f);
  uint32_t st = 12345u;
  std::vector<ConvLayer> Ls((size_t)nlayers);
  float* x = nullptr; float* r = nullptr; float* y = nullptr; unsigned char* xs = nullptr; unsigned char* ys = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  auto cleanup = [&]() {
    conv_x3s_force(0, 0, 0);
    if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1);
    dev_free(x); if (r) (void)hipFree(r); if (y) (void)hipFree(y); if (xs) (void)hipFree(xs); if (ys) (void)hipFree(ys);
    for (auto& L : Ls) conv_layer_free(L);
  };
  try {
    for (auto& L : Ls) {      // distinct weights per layer: cycled through, they come from HBM like a model's layers do
      for (auto& v : w) { st = st * 1664525u + 1013904223u; v = ((float)(st >> 8) / 8388608.f - 1.f) * 0.05f; }
      ConvBuildScope scope(2);
      if (w2d > 0) conv2d3x3_layer_init(L, w.data(), b.data(), Co, Ci); else conv1d_layer_init(L, w.data(), b.data(), Co, Ci, 1, 1, 0, 1, 1);
    }
    SplitGeom g; if (w2d > 0) g = split_geom_2d(w2d);
    const long long tp = ((long long)g.margin + T + 704 + 63) & ~63LL;
    std::vector<float> hx((size_t)Ci * T);
    for (auto& v : hx) { st = st * 1664525u + 1013904223u; v = (float)(st >> 8) / 8388608.f - 1.f; }
    x = dev_upload(hx.data(), hx.size());
    RVC_HIP_CHECK(hipMalloc(&r, (size_t)Co * T * 4)); RVC_HIP_CHECK(hipMemset(r, 0, (size_t)Co * T * 4));
    RVC_HIP_CHECK(hipMalloc(&y, (size_t)Co * T * 4));
    const size_t ib = (size_t)(Ci / 16) * 4 * tp * 16, ob = (size_t)((Co + 15) / 16) * 4 * tp * 16;
    RVC_HIP_CHECK(hipMalloc(&xs, ib)); RVC_HIP_CHECK(hipMemset(xs, 0, ib)); RVC_HIP_CHECK(hipMalloc(&ys, ob)); RVC_HIP_CHECK(hipMemset(ys, 0, ob));
    split_image_from_f32(s, x, T, Ci, T, xs + (size_t)(g.margin - kSplitMargin) * 16, tp);
    ConvEpilogue e;
    if (split_out) { e.act = ACT_GELU; e.ys_out = ys; e.ys_tp = tp; } else { e.R = r; e.ldR = T; }
    conv_x3s_force(ksplit, am, an);
    for (auto& L : Ls) conv_x3s_run(L, s, xs, tp, T, split_out ? nullptr : y, T, e, w2d > 0 ? &g : nullptr);
    RVC_HIP_CHECK(hipEventCreate(&e0)); RVC_HIP_CHECK(hipEventCreate(&e1));
    RVC_HIP_CHECK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) conv_x3s_run(Ls[(size_t)(i % nlayers)], s, xs, tp, T, split_out ? nullptr : y, T, e, w2d > 0 ? &g : nullptr);
    RVC_HIP_CHECK(hipEventRecord(e1, s));
    RVC_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.f; RVC_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    *us_out = ms * 1e3f / (float)reps;
  } catch (...) { cleanup(); throw; }
  cleanup();
  RVC_CATCH
}

#endif  // RVC_EXPERIMENTS

}  // extern "C"
