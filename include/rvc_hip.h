/* librvc_hip.so - C ABI of the MI355X-native RVC voice-conversion inference path.
 *
 * The reference (SayanoAI/Comfy-RVC) is pure Python with no FFI; its "operator API" for this path is a set of
 * duck-typed Python callables.  Each entry point below is what a ctypes binding of that callable binds
 * (comfy-rvc_amd/_lib.py is that binding; INTEGRATION.md shows the reference-side stubs):
 *
 *   rvc_hubert_forward   <- HubertModelWithFinalProj.extract_features   lib/infer_pack/loaders.py:55-61
 *   rvc_rmvpe_forward    <- RMVPE.infer_from_audio / _with_pitch         lib/rmvpe.py:614-659 (mel :510-556, E2E :464-470,
 *                                                                        decode :607-612,:661-685)
 *   rvc_synth_infer      <- SynthesizerTrnMs{256,768}NSFsid.infer        lib/infer_pack/models.py:682-693,:798-809
 *   rvc_vc_segment       <- VC.vc (features -> x2 upsample -> protect -> infer)   vc_infer_pipeline.py:25-114
 *   rvc_*_set_tensor     <- load_state_dict of the checkpoint tensors    vc_infer_pipeline.py:199-221,
 *                                                                        lib/infer_pack/loaders.py:19-31, lib/rmvpe.py:579-586
 *
 * Conventions: every function returns 0 on success and a non-zero status otherwise; rvc_last_error() gives the
 * thread-local message.  `stream` is a hipStream_t (NULL = default stream); all `*_dev` pointers are device pointers
 * owned by the caller (e.g. torch allocations); kernels are enqueued on `stream` and NOT synchronised.
 * Host pointers are only used for weights at load time.  Handles are not thread-safe; use one context per GPU/stream.
 * All tensors are float32 unless stated.  No CPU fallback exists: without a gfx950 device every call fails.
 */
#ifndef RVC_HIP_H
#define RVC_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rvc_ctx rvc_ctx;
typedef struct rvc_hubert rvc_hubert;
typedef struct rvc_rmvpe rvc_rmvpe;
typedef struct rvc_synth rvc_synth;

const char* rvc_last_error(void);
const char* rvc_version(void);

int rvc_ctx_create(int device_id, rvc_ctx** out);
/* Destroying the LAST context of a device also frees that device's per-stream scratch: no call may be in flight on that device from
 * another thread at that moment (destroy the handles first, then the context, from one thread). */
int rvc_ctx_destroy(rvc_ctx* ctx);
/* bytes of activation workspace currently held by the context */
int64_t rvc_ctx_workspace_bytes(rvc_ctx* ctx);
/* Matrix-core arithmetic of the models FINALIZED on this context afterwards (state of the handle, not of the process):
 *   0 fp32 MFMA everywhere, 1 / 2 bf16x3 split where a layer is eligible (see rvc_set_conv_precision), -1 (default) the mode of the
 *   thread that calls *_finalize. */
int rvc_ctx_set_conv_precision(rvc_ctx* ctx, int mode);

/* ------------------------------------------------------------------ HuBERT / ContentVec (HF HubertModel + final_proj) */
typedef struct rvc_hubert_taps {   /* optional device buffers for intermediate tensors (NULL = skip) */
  float* conv_stack;   /* [512][T_h]   output of the 7-layer feature encoder (channel-major == HF [1,512,T_h]) */
  float* pos_conv;     /* [768][T_h]   gelu(pos_conv(x)) before the residual add                                */
  float* hidden_0;     /* [768][T_h]   hidden_states[0]                                                         */
  float* hidden_8;     /* [768][T_h]   hidden_states[8]                                                         */
} rvc_hubert_taps;

int rvc_hubert_create(rvc_ctx* ctx, rvc_hubert** out);
int rvc_hubert_set_tensor(rvc_hubert* h, const char* name, const float* host_data, const int64_t* shape, int ndim);
int rvc_hubert_finalize(rvc_hubert* h);
int rvc_hubert_destroy(rvc_hubert* h);
/* T_h = floor((L - 400) / 320) + 1 */
int64_t rvc_hubert_num_frames(int64_t L);
/* version 1: final_proj(hidden_states[8]) -> D = 256; version 2: hidden_states[11] -> D = 768.
 * n_layers = 0 runs only the layers the output needs (8 / 11); 12 reproduces the reference's full stack (timing only).
 * out_rm_dev: [T_h][D] row-major (reference layout [1,T_h,D]) or NULL; out_cm_dev: [D][T_h] channel-major or NULL. */
int rvc_hubert_forward(rvc_hubert* h, void* stream, const float* audio_dev, int64_t L, int version, int n_layers,
                       float* out_rm_dev, float* out_cm_dev, const rvc_hubert_taps* taps);

/* ------------------------------------------------------------------ RMVPE */
typedef struct rvc_rmvpe_taps {
  float* unet_out;     /* [16][T_r][128] */
  float* gru;          /* [512][T_r] channel-major (fwd | bwd) */
} rvc_rmvpe_taps;

int rvc_rmvpe_create(rvc_ctx* ctx, rvc_rmvpe** out);
int rvc_rmvpe_set_tensor(rvc_rmvpe* r, const char* name, const float* host_data, const int64_t* shape, int ndim);
/* mel_basis [128][513] and the STFT forward basis [1026][1024] are tensors "mel_basis" / "stft.forward_basis" */
int rvc_rmvpe_finalize(rvc_rmvpe* r);
int rvc_rmvpe_destroy(rvc_rmvpe* r);
/* n = L / 160 + 1 frames.  mel_dev [128][n], salience_dev [n][360], f0_dev float64 [n]; any may be NULL. */
int rvc_rmvpe_forward(rvc_rmvpe* r, void* stream, const float* audio_dev, int64_t L, float thred, float* mel_dev,
                      float* salience_dev, double* f0_dev, const rvc_rmvpe_taps* taps);
/* Waits for `stream` and reports the last forward of this handle: 0 = ok.  The BiGRU scan's 16 workgroups exchange h_t through polled
 * device memory and need co-residency; when a hand-off times out (bounded spin) a serial, communication-free pass enqueued behind every scan
 * recomputes the recurrence (same summation order: hidden states within 2.4e-7 of a healthy run, ~6 us per step instead of ~1), so the forward still ends with valid f0 and
 * rvc_rmvpe_status stays 0 - rvc_rmvpe_repaired tells that it happened.  Non-zero (message via rvc_last_error) only when the time-out
 * was not repaired (the repair pass switched off by RVC_GRU_REPAIR=0 or by the test hook): f0_dev of that forward holds NaN then.
 * The reference has no counterpart (torch.nn.GRU, lib/rmvpe.py:420-428). */
int rvc_rmvpe_status(rvc_rmvpe* r, void* stream);
int rvc_rmvpe_repaired(rvc_rmvpe* r, void* stream);   /* 1: the last forward's scan timed out and was repaired by the serial pass */
/* test hook: fault & 1 makes one workgroup of the following scans exit without publishing; fault & 2 keeps the repair pass from being
 * enqueued; spin_limit (0 = default 2^24 polls) bounds how long the peers wait before they raise the flag */
int rvc_rmvpe_debug_fault(rvc_rmvpe* r, int fault, unsigned spin_limit);
/* decode alone: salience_dev [n][360] row-major -> f0 float64 [n] */
int rvc_rmvpe_decode(rvc_rmvpe* r, void* stream, const float* salience_dev, int64_t n, float thred, double* f0_dev);
/* Tail of get_f0 (reference pitch_extraction.py get_f0: transpose by `factor` = 2^(key / 12), mel-scale quantisation to 1 .. bins - 1 with np.rint), on the
 * device in float64: pitch_dev int64 [n] (coarse), pitchf_dev float32 [n] (Hz).  mel_min / mel_max = hz_to_mel(f0_min / f0_max) = 2595 log10(1 + f / 700). */
int rvc_f0_post(void* stream, const double* f0_dev, int64_t n, double factor, double mel_min, double mel_max, int bins, int64_t* pitch_dev, float* pitchf_dev);

/* ------------------------------------------------------------------ CREPE (torchcrepe.Crepe "full" / "tiny") */
/* Replaces the network inside torchcrepe.predict, which the reference calls for f0_method "crepe" / "mangio-crepe"
 * (pitch_extraction.py:76-150; torchcrepe is a third-party package, not vendored upstream: its published architecture is restated).
 * Tensors by their torchcrepe state-dict names: conv{1..6}.weight / .bias, conv{1..6}_BN.{weight,bias,running_mean,running_var},
 * classifier.weight / .bias. */
typedef struct rvc_crepe rvc_crepe;
typedef struct rvc_crepe_taps {
  float* conv1;        /* [C1][256]  ReLU(conv1) of frame 0 (before BatchNorm / pooling) */
  float* embed;        /* [4*C6][B]  flattened last feature map of the first batch of frames (B = min(n, 512)), row = position * C6 + channel */
} rvc_crepe_taps;
int rvc_crepe_create(rvc_ctx* ctx, int tiny, rvc_crepe** out);
int rvc_crepe_set_tensor(rvc_crepe* c, const char* name, const float* host_data, const int64_t* shape, int ndim);
int rvc_crepe_finalize(rvc_crepe* c);
int rvc_crepe_destroy(rvc_crepe* c);
/* frames of 1024 samples every `hop`: pad != 0 (torchcrepe's default) zero-pads the audio by 512 on both sides, n = 1 + L / hop;
 * pad == 0: n = 1 + (L - 1024) / hop */
int64_t rvc_crepe_num_frames(int64_t L, int hop, int pad);
/* audio_dev float32 [L] at 16 kHz -> probs_dev [360][n] channel-major: per-frame sigmoid outputs over the 360 pitch bins
 * (the mean / std normalisation of every frame, the six conv blocks and the classifier); decoding stays with the caller */
int rvc_crepe_forward(rvc_crepe* c, void* stream, const float* audio_dev, int64_t L, int hop, int pad, float* probs_dev, const rvc_crepe_taps* taps);

/* torchcrepe.decode.viterbi (its librosa.sequence.viterbi call included) and the periodicity gather of core.postprocess on the device:
 * bins outside [min_bin, max_bin) are masked, softmax over bins, most likely path under the triangular (width 12) transition matrix with a
 * uniform prior; bins_dev int32 [n], periodicity_dev [n] = probs[bin][frame].  Cents / dither / Hz stay on the host (numpy's RNG). */
int rvc_crepe_viterbi(void* stream, const float* probs_dev, int64_t n, int min_bin, int max_bin, int32_t* bins_dev, float* periodicity_dev);

/* ------------------------------------------------------------------ MDX23C vocal / instrumental separation (UVR chain) */
/* Replaces TFC_TDF_net.forward of reference lib/karafan/tfc_tdf.py:147-235 (with its STFT / inverse, :47-77); demix_mdxv3's chunking
 * (lib/karafan/inference.py:32-74) calls it once per chunk.  Fields = the entries of the model's yaml that the network reads
 * (lib/karafan/Data/model_2_stem_full_band_8k.yaml).  Tensors by their state-dict names plus three host-built constants:
 * "stft.basis" [2 dim_f][n_fft], "istft.basis" [n_fft][2 dim_f] (windowed DFT matrices) and "window" [n_fft] (periodic hann). */
typedef struct rvc_mdx23 rvc_mdx23;
typedef struct rvc_mdx23_config {
  int n_fft, hop, dim_f, dim_t;
  int num_channels, growth, num_scales, num_subbands, blocks_per_scale, bottleneck;
  int num_targets;       /* instruments the mask head separates (2: vocals, instrumental) */
  int audio_channels;    /* 2 */
} rvc_mdx23_config;
int rvc_mdx23_create(rvc_ctx* ctx, const rvc_mdx23_config* cfg, rvc_mdx23** out);
int rvc_mdx23_set_tensor(rvc_mdx23* m, const char* name, const float* host_data, const int64_t* shape, int ndim);
int rvc_mdx23_finalize(rvc_mdx23* m);
int rvc_mdx23_destroy(rvc_mdx23* m);
/* chunk_dev float32 [2][L] with L = hop * (dim_t - 1) -> out_dev [num_targets][2][L] */
int rvc_mdx23_forward(rvc_mdx23* m, void* stream, const float* chunk_dev, int64_t L, float* out_dev);
/* The chunk loop of demix_mdxv3 (reference lib/karafan/inference.py:52-66) in one call: n_chunks chunks of C = hop (dim_t - 1) samples every `step` samples of the
 * zero-padded stereo mix mix_dev [2][Lp] ((n_chunks - 1) step + C <= Lp); each chunk's separated signals are added (NaN as zero) into acc_dev [S][2][Lp] at the
 * chunk's offset, in chunk order, and the sum is divided by `overlap`.  acc_dev is overwritten.  The caller pads the mix and trims the result as the reference does. */
int rvc_mdx23_demix(rvc_mdx23* m, void* stream, const float* mix_dev, int64_t Lp, int64_t step, int64_t n_chunks, float overlap, float* acc_dev);
/* Chunk streams of rvc_mdx23_demix: 1 (default) = chunks in order on the caller's stream, the reference's order of additions; k = 2 .. 4: chunk c runs on stream
 * c mod k (streams, arenas and accumulators of the model's own, ordered against the caller's stream by events), the k accumulators are added in stream order -
 * reproducible, an ulp-level re-association.  For a single conversion alone on the GPU (the node); costs throughput when several clips are in flight. */
int rvc_mdx23_set_streams(rvc_mdx23* m, int k);

/* ------------------------------------------------------------------ synthesizer */
typedef struct rvc_synth_config {   /* the fields of cpt["config"] that the inference graph needs */
  int inter_channels, hidden_channels, filter_channels, n_heads, n_layers, kernel_size;
  int n_resblock_kernels; int resblock_kernel_sizes[3]; int resblock_dilations[3][3];
  int n_upsamples; int upsample_rates[8]; int upsample_kernel_sizes[8];
  int upsample_initial_channel, spk_embed_dim, gin_channels, sr;
  int feat_dim;                    /* 768 (v2) or 256 (v1) */
} rvc_synth_config;

typedef struct rvc_synth_taps {
  float* enc_p_layer0;  /* [192][T] */
  float* m_p;           /* [192][T] */
  float* logs_p;        /* [192][T] */
  float* z_p;           /* [192][T] */
  float* z;             /* [192][T] */
  float* sine_waves;    /* [T*upp]  SineGen output before Linear+tanh */
  float* har_source;    /* [T*upp] */
  float* gen_ups0;      /* [C0][T*u0] first upsample + noise conv */
  float* gen_last;      /* [C_last][T*upp] output of the last ResBlock stage */
} rvc_synth_taps;

int rvc_synth_create(rvc_ctx* ctx, const rvc_synth_config* cfg, rvc_synth** out);
int rvc_synth_set_tensor(rvc_synth* s, const char* name, const float* host_data, const int64_t* shape, int ndim);
int rvc_synth_finalize(rvc_synth* s);
int rvc_synth_destroy(rvc_synth* s);
int rvc_synth_upp(rvc_synth* s);
/* 1: SynthesizerTrnMs{256,768}NSFsid; 0: the no-f0 family SynthesizerTrnMs{256,768}NSFsid_nono (lib/infer_pack/models.py:812-1022),
 * decided at finalize by the checkpoint (no enc_p.emb_pitch.*, dec.m_source.*, dec.noise_convs.*).  A no-f0 model takes NULL for
 * pitch_dev, pitchf_dev and noise_src_dev in the calls below (infer(phone, phone_lengths, sid), one noise draw) and no protect blend
 * (vc_infer_pipeline.py:84,:105-108). */
int rvc_synth_has_f0(rvc_synth* s);
/* phone_dev: [T][D] row-major (reference [1,T,D]) if phone_channel_major == 0, else [D][T].
 * pitch_dev int64 [T] (coarse 1..255), pitchf_dev [T] (Hz), noise_z_dev [inter][T], noise_src_dev [T*upp]
 * (the two torch.randn_like draws of the reference, models.py:801 and :409).  out_dev [T*upp]. */
int rvc_synth_infer(rvc_synth* s, void* stream, const float* phone_dev, int phone_channel_major, const int64_t* pitch_dev,
                    const float* pitchf_dev, int sid, const float* noise_z_dev, const float* noise_src_dev, int64_t T,
                    float* out_dev, const rvc_synth_taps* taps);

/* ------------------------------------------------------------------ fused segment: VC.vc without index retrieval */
/* audio_dev [L] 16 kHz segment; pitch/pitchf as above with at least p_len = 2*T_h entries; out_dev [2*T_h*upp].
 * do_protect != 0 applies the protect blend (protect < 0.5 in the reference). */
int rvc_vc_segment(rvc_hubert* h, rvc_synth* s, void* stream, const float* audio_dev, int64_t L, int version,
                   const int64_t* pitch_dev, const float* pitchf_dev, int sid, float protect, int do_protect,
                   const float* noise_z_dev, const float* noise_src_dev, float* out_dev);

/* Same, starting from HuBERT features that already live on the device (channel-major [feat_dim][T_h]); lets the caller run
 * HuBERT on a second stream while RMVPE produces the pitch. */
int rvc_vc_segment_feats(rvc_synth* s, void* stream, const float* feats_cm_dev, const float* feats0_cm_dev, int64_t T_h, int feat_dim,
                         const int64_t* pitch_dev, const float* pitchf_dev, int sid, float protect, int do_protect,
                         const float* noise_z_dev, const float* noise_src_dev, float* out_dev);
/* feats0_cm_dev: the features before index retrieval (the reference's feats0 of the protect blend, :58-59,:89-95) or NULL when
 * no index is used. */

/* ------------------------------------------------------------------ feature retrieval (vc_infer_pipeline.py:60-75) */
/* The reference looks every HuBERT frame up in a faiss IVF-Flat index over the training features big_npy [N][D]
 * (index.search(npy, k=1); pitch_extraction.py:52-73 loads it, custom_nodes/rvc_nodes.py:500-554 builds it) and blends the
 * neighbour in with weight index_rate.  rvc_index_* is that search, device resident, over the same big_npy: search returns idx [T]
 * (and optionally the squared distances faiss would report), blend writes out = index_rate * big_npy[idx] + (1 - index_rate) * feats.
 * Features are channel-major [D][T].
 * rvc_index_create: exact L2 nearest neighbour (for inputs without a cell structure: a big_npy array / the reference's preloaded tuple).
 * rvc_index_create_ivf: the semantics of the reference's own index, faiss IndexIVFFlat as RVCTrainModelNode.train_index builds it
 * ("IVF{n},Flat", nprobe 1): the nprobe centroids nearest to the query (coarse IndexFlatL2 quantiser) select the cells, only vectors of
 * those cells compete (list_of[j] = the list row j was added to); probed cells without vectors give idx -1 / score FLT_MAX and - like the
 * reference's weight arithmetic, :66-74 - a NaN frame in the blend.  nprobe <= 16, or >= nlist (= exact). */
typedef struct rvc_index rvc_index;
int rvc_index_create(rvc_ctx* ctx, const float* big_npy_host, int64_t N, int D, rvc_index** out);
int rvc_index_create_ivf(rvc_ctx* ctx, const float* big_npy_host, int64_t N, int D, const float* centroids_host /* [nlist][D] */, int nlist,
                         const int32_t* list_of_host /* [N] */, int nprobe, rvc_index** out);
int rvc_index_nprobe(const rvc_index* h);   /* 0: exact search */
int rvc_index_destroy(rvc_index* h);
int64_t rvc_index_ntotal(const rvc_index* h);
int rvc_index_search(rvc_index* h, void* stream, const float* feats_cm_dev, int64_t T, int64_t* idx_dev, float* score_dev /* may be NULL */);
int rvc_index_blend(rvc_index* h, void* stream, const float* feats_cm_dev, const int64_t* idx_dev, int64_t T, float index_rate, float* out_cm_dev);
/* Input pre-processing of VC.pipeline on the device: the zero-phase 5th-order high-pass (vc_infer_pipeline.py:19,121:
 * scipy.signal.filtfilt(bh, ah, audio) - odd extension by 18 samples, lfilter_zi initial conditions, float64), the reflect
 * padding by t_pad with the float32 cast the networks consume (:141) and the RMS frames of the filtered input that change_rms
 * uses (lib/model_utils.py:45; frame 1 s, hop 0.5 s at 16 kHz, n1 = n / 8000 + 1; may be NULL).
 * audio_dev [n] float32 (is_f64 = 0) or float64 (1); b6 / a6 / zi5 are HOST arrays (butter coefficients, lfilter_zi);
 * filt_dev [n] float64 = filtered signal, padded_dev [n + 2 t_pad] float32 (may be NULL).
 * sos18_host / sos_zi6_host (both or neither, HOST arrays): the SAME filter as three second-order sections [3][6] = {b0, b1, b2, 1, a1, a2}
 * (scipy.signal.butter(..., output="sos"); a first-order section has b2 = a2 = 0) and scipy.signal.sosfilt_zi [3][2].  With them the filter
 * runs block-propagated in the cascade form (3 launches per direction of 2 x 512 steps per thread instead of 5632: ~20 x less work at the
 * start of every conversion); equal to filtfilt(b, a) in exact arithmetic, different from scipy's transfer-function evaluation by that
 * evaluation's float64 rounding noise (~4e-8 of full scale).  Without them: the overlap-discard transfer-function kernels. */
int rvc_preprocess(void* stream, const void* audio_dev, int is_f64, int64_t n, const double* b6_host, const double* a6_host,
                   const double* zi5_host, int t_pad, double* filt_dev, float* padded_dev, double* rms1_dev, int n1,
                   const double* sos18_host, const double* sos_zi6_host);
/* Output post-processing of VC.pipeline on the device: change_rms (lib/model_utils.py:39-57; skipped when rms_mix_rate >= 1 or
 * rms1_dev == NULL) followed by peak normalisation to int16 (vc_infer_pipeline.py:188-189).  wav_dev [N] float32 is modified in
 * place; rms1_dev = RMS frames of the 16 kHz input (float64 [n1], hop 0.5 s); sr2 = output rate. */
int rvc_postprocess(void* stream, float* wav_dev, int64_t N, const double* rms1_dev, int n1, int sr2, float rms_mix_rate, int16_t* out_i16_dev);

/* ------------------------------------------------------------------ single ops (parity tests / kernel benchmarks) */
/* Conv1d: x_dev [Ci][Tin], w_host [Co][Ci/groups][k], y_dev [Co][Tout]; act codes: 0 none 1 lrelu 2 relu 3 gelu 4 tanh 5 sigmoid */
int rvc_op_conv1d(void* stream, const float* x_dev, const float* w_host, const float* bias_host, const float* res_dev, float* y_dev,
                  int Ci, int Co, int Tin, int k, int stride, int pad, int dil, int groups, int pre_act, float pre_slope, int act,
                  float act_slope, int act_before_res, float out_scale, int accumulate);
/* k = 1 projection (torch.nn.Linear over [T, Ci] rows, reference transformers/models/hubert/modeling_hubert.py:291-477 via
 * lib/infer_pack/loaders.py:55-61) on the split-resident GEMM kernel (csrc/conv_x3s.hip): x_dev [Ci][T] fp32 is first written as the bf16
 * hi / lo image the kernel stages, y = act(W x + b [+ res]) (act_before_res: act(W x + b) + res).  y_dev fp32 [Co][T] or null;
 * ysplit_f32_dev [Co][T] or null receives the kernel's SPLIT output image read back as hi + lo.  ksplit > 0 forces the K split (reduced
 * inside the launch), am / an > 0 force the tile (64 am rows x 64 an columns).  k > 1: w_host [Co][Ci][k], a 1-D "same" convolution with
 * dilation dil whose taps are row offsets into the image (odd k, pad (k - 1) / 2 * dil <= 64). */
int rvc_op_gemm_split(void* stream, const float* x_dev, const float* w_host, const float* bias_host, const float* res_dev, float* y_dev,
                      float* ysplit_f32_dev, int Ci, int Co, int T, int act, float act_slope, int act_before_res, float out_scale, int ksplit,
                      int am, int an, int k, int dil);
/* General stride-1 Conv1d on the same kernel, grouped and with long kernels (HuBERT's positional convolution, transformers modeling_hubert.py:
 * 61-106: Conv1d(768, 768, 128, padding 64, groups 16), output cut to T, GELU, + residual): w_host [Co][Ci / groups][k], output length T
 * (the first T positions), taps are row offsets tap * dil - pad into the zero-margined image. */
int rvc_op_conv1d_split(void* stream, const float* x_dev, const float* w_host, const float* bias_host, const float* res_dev, float* y_dev,
                        int Ci, int Co, int T, int k, int pad, int dil, int groups, int act, int act_before_res);
/* Stride-2 Conv1d without padding on the split-resident kernel, the way HuBERT's feature-encoder layers 1 .. 6 run (transformers modeling_hubert.py HubertNoLayerNormConvLayer,
 * k = 3 / 2, stride 2, through lib/infer_pack/loaders.py:55-61): the input is turned into the DE-INTERLEAVED bf16 hi / lo image (even | odd positions) a producer's epilogue
 * would write, a tap is then a row offset; y_dev [Co][Tout] fp32, Tout = (T - k) / 2 + 1; y_img_f32_dev (or NULL): the same result read back from the de-interleaved OUTPUT
 * image (what the next stride-2 layer stages).  Test / benchmark op. */
int rvc_op_conv1d_s2_split(void* stream, const float* x_dev, const float* w_host, const float* bias_host, float* y_dev, float* y_img_f32_dev, int Ci, int Co, int T, int k,
                           int act);
/* Conv2d 3 x 3, pad 1 (reference lib/rmvpe.py:233-268 ConvBlockRes convolutions) on the same kernel over PADDED split-resident images (row pitch
 * W + 2, taps as row offsets): x_dev [Ci][H][W] plain fp32 is padded and split on the device, y = act(conv(x) + b) with the residual before
 * or after the activation, returned plain [Co][H][W]; ysplit_f32_dev as above (plain layout). */
int rvc_op_conv2d_split(void* stream, const float* x_dev, const float* w_host, const float* bias_host, const float* res_dev, float* y_dev,
                        float* ysplit_f32_dev, int Ci, int Co, int H, int W, int act, int act_before_res, int ksplit, int am, int an);
/* ConvTranspose1d: w_host [Ci][Co][k]; y_dev [Co][(Tin-1)*u - 2*pad + k] */
int rvc_op_conv_transpose1d(void* stream, const float* x_dev, const float* w_host, const float* bias_host, float* y_dev, int Ci, int Co,
                            int Tin, int k, int u, int pad, int pre_act, float pre_slope, int accumulate);
/* Conv2d 3x3 pad 1: x_dev [Ci][H][W], w_host [Co][Ci][3][3]; ReLU-then-residual epilogue if relu != 0 */
int rvc_op_conv2d3x3(void* stream, const float* x_dev, const float* w_host, const float* bias_host, const float* res_dev, float* y_dev,
                     int Ci, int Co, int H, int W, int relu);
/* ConvTranspose2d k3 s2 p1 op1: w_host [Ci][Co][3][3]; y_dev [Co][2H][2W] */
int rvc_op_conv_transpose2d(void* stream, const float* x_dev, const float* w_host, const float* bias_host, float* y_dev, int Ci, int Co,
                            int H, int W, int relu);
/* y[z][m][n] = sum_k a[z][k][m] * b[z][k][n] */
int rvc_op_gemm_tn(void* stream, const float* a_dev, const float* b_dev, float* y_dev, int M, int N, int K, int batch);
/* plan API for benchmarking the dominant kernel without host-side packing in the loop */
typedef struct rvc_conv1d_plan rvc_conv1d_plan;
int rvc_conv1d_plan_create(const float* w_host, const float* bias_host, int Ci, int Co, int k, int stride, int pad, int dil, int groups,
                           rvc_conv1d_plan** out);
int rvc_conv1d_plan_run(rvc_conv1d_plan* p, void* stream, const float* x_dev, int Tin, const float* res_dev, float* y_dev, int pre_act,
                        float pre_slope, int act, float act_slope);
/* One ResBlock1 pair of the generator in a single launch (reference lib/infer_pack/modules.py:295-308):
 * y = (x + conv2(lrelu(conv1(lrelu(x), dilated)))) * out_scale [+ y]; x_dev, y_dev [C][T].  Fails if the pair is not eligible. */
/* The same pair as two launches with a split-resident intermediate, the way the wide generator stages run it (reference
 * lib/infer_pack/modules.py:295-308): conv1 writes lrelu(conv1(lrelu(x)) + b1) as the bf16 hi / lo image conv2 stages by DMA.
 * Fails if the layers are not eligible for the split-resident paths at this length. */
int rvc_conv1d_plan_pair_split_run(rvc_conv1d_plan* c1, rvc_conv1d_plan* c2, void* stream, const float* x_dev, int T, float* y_dev,
                                   float out_scale, int accumulate);
int rvc_conv1d_plan_pair_run(rvc_conv1d_plan* c1, rvc_conv1d_plan* c2, void* stream, const float* x_dev, int T, float* y_dev, float out_scale,
                             int accumulate);
/* A whole ResBlock1 of the generator's 32-channel stage in ONE launch (reference lib/infer_pack/modules.py:295-308, the loop over convs1 / convs2; the
 * generator's xs += resblock(x), x = xs / 3 of models.py:555-560 is out_scale / accumulate): plans = {c1_0, c2_0, c1_1, c2_1, c1_2, c2_2};
 * y = x3 * out_scale [+ y] with x_{i+1} = x_i + c2_i(lrelu(c1_i(lrelu(x_i)))); x_dev, y_dev [32][T].  fp16x2 pair arithmetic only; bit-identical to three
 * rvc_conv1d_plan_pair_run calls.  *ran_out = 1 when the fused kernel ran, 0 when the layers / length are not eligible (nothing is written then: the caller
 * runs the pairs one by one).  noise_*_dev (all three or none): the last generator stage's noise branch (models.py GeneratorNSF.forward, x = ups(x) +
 * noise_convs[-1](har), a Conv1d(1, 32, 1)) folded into the read of x: the ResBlock sees x[c][t] + fmaf(noise_w[c], noise_src[t], noise_b[c]). */
int rvc_conv1d_plan_resblock_run(rvc_conv1d_plan* const* plans6, void* stream, const float* x_dev, int T, float* y_dev, float out_scale, int accumulate,
                                 int* ran_out, const float* noise_src_dev, const float* noise_w_dev, const float* noise_b_dev);
/* The arithmetic rvc_conv1d_plan_pair_split_run (and the generator) uses for this pair at length T under the current rvc_set_pair_arithmetic mode:
 * 1 fp16x2, 0 bf16x3 (mode 0, a layer without an fp16 image, or a length / shape the persistent kernel declines); -1 on a null argument. */
int rvc_conv1d_plan_pair_arithmetic(rvc_conv1d_plan* c1, rvc_conv1d_plan* c2, int T);
int rvc_conv1d_plan_destroy(rvc_conv1d_plan* p);
/* fused softmax(K^T Q) V + bias for head dimension 64: q_dev, k_dev channel-major [heads*64][T] (q pre-scaled), v_rm_dev row-major
 * [T][heads*64], bv_dev [heads*64] or NULL, out_dev channel-major [heads*64][T] */
int rvc_op_attention(void* stream, const float* q_dev, const float* k_dev, const float* v_rm_dev, const float* bv_dev, float* out_dev,
                     int heads, int T);
/* the synthesizer text encoder's attention (reference lib/infer_pack/attentions.py:230-267), head dimension 96, window 10:
 * softmax over keys of K^T Q + rel[k - q + 10][q] (|k - q| <= 10), times V, + bv.  rel_dev [heads][21][T] in (the Q . emb_rel_k
 * projection), pb_dev [heads][21][T] out: pb[r][q] = P[q][q + r - 10] (0 outside the sequence), the input of the rel-v projection.
 * ek_dev / ev_dev (both or neither): emb_rel_k / emb_rel_v [21][96], shared by the heads - the kernel then computes both projections itself
 * (rel = Q . E_k on the staged Q tile, out += P_band . E_v in the merge); rel_dev / pb_dev may be null in that form. */
/* attention on split-resident operands (attention_dma.hip; HuBERT's encoder layers, modeling_hubert.py:291-477): q, k, v fp32 channel-major
 * [heads*64][T]; the op builds the q / k image and the V^T image itself.  out fp32 [heads*64][T] and / or the output image read back as fp32. */
int rvc_op_attention_split(void* stream, const float* q_dev, const float* k_dev, const float* v_dev, const float* bv_dev, float* out_dev,
                           float* out_img_f32_dev, int heads, int T);
/* the text encoder's variant (head dimension 96, relative positions within +-10: reference attentions.py:230-267): ek / ev host [21][96];
 * kz > 0 forces the number of key slices merged inside the launch (0: automatic). */
int rvc_op_attention_split_rel(void* stream, const float* q_dev, const float* k_dev, const float* v_dev, const float* bv_dev, const float* ek_host,
                               const float* ev_host, float* out_dev, float* out_img_f32_dev, int heads, int T, int kz);
/* one ConvBlockRes of RMVPE's shallow U-Net levels in one launch (conv_cbr2.hip; reference lib/rmvpe.py:233-268 with BatchNorm folded):
 * y = relu(conv3x3(relu(conv3x3(x, w1) + b1), w2) + b2) + x; x, y device fp32 [C][H][W] (distinct), w host [C][C][3][3], C = 16 or 32. */
int rvc_op_cbr2_small(void* stream, const float* x_dev, const float* w1_host, const float* b1_host, const float* w2_host, const float* b2_host,
                      float* y_dev, int C, int H, int W);
/* LayerNorm over channels (column-wise on [C][T]) with the split-resident image as output (layernorm_c_split_kernel: HuBERT / text-encoder layers,
 * modeling_hubert.py:291-477): y fp32 (optional) and the image read back as fp32; C a multiple of 16. */
int rvc_op_layernorm_c_split(void* stream, const float* x_dev, const float* gamma_dev, const float* beta_dev, float* y_dev, float* y_img_f32_dev, int C, int T);
/* one 3 x 3 convolution of RMVPE's shallow levels (conv_cbr2.hip conv3_small_kernel; Ci = 16 with Co <= 64, or Ci = 32 with Co <= 32): rows below relu_rows get
 * the ReLU; rows < split_row go to y (+ res when given), the others to y2 (row - split_row): first convolution + 1 x 1 shortcut of a block in one launch. */
int rvc_op_conv3_small(void* stream, const float* x_dev, const float* w_host, const float* b_host, const float* res_dev, float* y_dev, float* y2_dev, int Ci, int Co,
                       int H, int W, int split_row, int relu_rows);
/* A WaveNet layer's in_layer with its gate in the GEMM's epilogue (reference lib/infer_pack/modules.py WN.forward: x_in = in_layers[i](x); acts =
 * fused_add_tanh_sigmoid_multiply(x_in, g_l), commons.py): y[c][t] = tanh(a[c][t] + g[c]) * sigmoid(a[H + c][t] + g[H + c]) with a = conv1d(x, w [2 H][Ci][k]) + b,
 * "same" padding; the 2 H-row tensor is never stored - the rows are packed so that a lane holds both halves of a channel - and y comes back from the image the
 * res / skip layer would stage.  g_dev [2 H] on the device (or NULL).  Test / benchmark op. */
int rvc_op_wn_in_gate_split(void* stream, const float* x_dev, const float* w_host, const float* bias_host, const float* g_dev, float* y_dev, int Ci, int H, int T, int k);
/* A fused q | k | v projection on the split-resident GEMM (the attentions of HuBERT and of the synthesizer's text encoder: transformers modeling_hubert.py HubertAttention,
 * lib/infer_pack/attentions.py:57-69): y = W x + b in ONE launch, rows [0, vt_row0) written as their bf16 hi / lo image (read back into y_img_f32_dev [vt_row0][T]), rows
 * [vt_row0, Co) written TRANSPOSED as the attention's V^T image (read back into yt_dev [ceil64(T)][Co - vt_row0]; rows T .. ceil64(T) exact zeros).  vt_row0 a multiple of
 * 128.  Test / benchmark op. */
int rvc_op_gemm_split_qkv(void* stream, const float* x_dev, const float* w_host, const float* bias_host, float* y_img_f32_dev, float* yt_dev, int Ci, int Co, int T,
                          int vt_row0);
/* the swapped product of the split-resident GEMM: yt[t][j] = sum_c x[c][t] w[row0 + j][c], j < rows (the V^T image of the attention, read back as
 * fp32 [ceil64(T)][rows]; rows t >= T are zeros).  w host [Co][Ci]. */
int rvc_op_gemm_split_swapped(void* stream, const float* x_dev, const float* w_host, float* yt_dev, int Ci, int Co, int T, int row0, int rows);
/* Test ops of the two-image split-resident product (MDX23C, csrc/model_mdx23.hip: reference lib/karafan/tfc_tdf.py:137-144, `s = shortcut(x) ... x = tfc2(x) + s`):
 * y = conv3x3(x1, w1 [Co][Ci1][3][3], zero padding) + conv1x1(x2, w2 [Co][Ci2]) as ONE launch over the padded images of x1 and x2 (plain [C][H][W] tensors in
 * and out; y_img_f32: the raw output image written beside it, read back as fp32, or null); and the swapped product with a residual in the row-major layout,
 * y[t][off + j] = sum_c x[c][t] w[j][c] + res[t][off + j] with row pitch ld (tfc_tdf.py:142, `x = x + self.tdf(x)`). */
int rvc_op_conv2d3x3_plus_1x1(void* stream, const float* x1_dev, const float* w1_host, const float* x2_dev, const float* w2_host, float* y_dev, float* y_img_f32_dev,
                              int Ci1, int Ci2, int Co, int H, int W, int ksplit);
int rvc_op_gemm_split_swapped_res(void* stream, const float* x_dev, const float* w_host, const float* res_dev, float* y_dev, int Ci, int Co, int T, int ld, int off);
int rvc_op_attention_rel(void* stream, const float* q_dev, const float* k_dev, const float* v_rm_dev, const float* bv_dev, const float* rel_dev,
                         float* pb_dev, float* out_dev, int heads, int T, const float* ek_dev, const float* ev_dev);
int rvc_op_layernorm_c(void* stream, const float* x_dev, const float* res_dev, const float* gamma_dev, const float* beta_dev, float* y_dev,
                       int C, int T);
/* optional debug outputs: rad_dev [T] per-frame phase increment, tmp_dev [T] scaled frame cumsum, phase_dev [T*upp] running phase (cycles) */
int rvc_op_sine_source(void* stream, const float* f0_dev, const float* noise_dev, float* har_dev, float* sine_dev, int T, int upp, float sr,
                       float lin_w, float lin_b, float* rad_dev, float* tmp_dev, float* phase_dev);

/* Rational resampling by up/down (lowest terms): y_dev[n] = sum_m x_dev[m] * taps_dev[m*up - n*down + half], n < n_out, float64
 * accumulation, zero extension outside the input.  taps_dev: 2*half+1 float64 coefficients of a linear-phase low-pass on the
 * up-times up-sampled grid, DC gain `up`.  Replaces librosa.resample (soxr_hq) at lib/audio.py:150 (input -> 16 kHz) and
 * vc_infer_pipeline.py:186 (output -> resample_sr); the coefficients are designed on the host (comfy-rvc_amd/lib/audio.py). */
int rvc_resample(void* stream, const float* x_dev, int64_t n_in, const double* taps_dev, int half, int up, int down, float* y_dev, int64_t n_out);

/* Matrix-core arithmetic of the Conv1d layers created AFTER the call BY THE CALLING THREAD (thread-local: free-standing ops / plans,
 * and models whose context is left at mode -1; rvc_ctx_set_conv_precision is the per-handle form):
 *   0  fp32 MFMA everywhere (v_mfma_f32_32x32x2_f32; bitwise an fp32 FMA chain)
 *   1  default: every eligible layer of the models (HuBERT incl. its projections, RMVPE, the synthesizer, the feature index, MDX23C,
 *      CREPE) uses the bf16x3 split (x = hi + lo in bf16, hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation;
 *      ~1e-5 relative error per layer); grouped / Ci % 16 != 0 / under-filled launches and free-standing ops / plans stay fp32
 *   2  bf16x3 for every eligible layer (stride 1, groups 1, Ci % 16 == 0), including rvc_op_conv1d / plans (parity tests) */
int rvc_set_conv_precision(int mode);
/* Arithmetic of the generator's ResBlock pairs (reference lib/infer_pack/modules.py:295-308: x + c2(lrelu(c1(lrelu(x))))) on the persistent kernel,
 * PROCESS-WIDE, read at every launch (switching needs no reload: layers carry both weight images):
 *   1  default: fp16x2 - the weight is ONE fp16 term (2^-12 relative rounding), the activation fp16 hi + lo (22 bits), two
 *      v_mfma_f32_32x32x16_f16 per product, fp32 accumulation.  Full-size goldens stay within the 33-LSB (1e-3) gate; see DESIGN.md.
 *      Activations beyond +-131008 would saturate (fp16 range): not reachable by a tanh-terminated vocoder, and a layer whose weights
 *      exceed 60000 or are all below 2^-10 keeps mode 0 by itself.
 *   0  bf16x3 as everywhere else (three MFMAs per product, ~1e-5 relative error per layer).
 * Environment RVC_H2=0 sets the initial mode to 0.  rvc_get_pair_arithmetic returns the current mode. */
int rvc_set_pair_arithmetic(int mode);
int rvc_get_pair_arithmetic(void);

/* ------------------------------------------------------------------ kernel profiling (bench.py roofline leg) */
/* While enabled, every launch of the MFMA convolution kernels is bracketed by HIP events on its own stream and tagged with its
 * algorithmic FLOPs.  rvc_prof_collect sums them per kernel configuration into arrays of RVC_PROF_CFGS entries
 * (7 tilings x {fp32 1-D, fp32 2-D, bf16x3 1-D}, rest unused); names via rvc_prof_cfg_name. */
#define RVC_PROF_CFGS 24
int rvc_prof_enable(int on);
int rvc_prof_collect(double* ms, double* flops, int64_t* launches);
/* Same records split by roofline regime: out[cfg][0..3] = {ms, FLOPs, algorithmic HBM bytes, launches} of the launches whose
 * arithmetic intensity (FLOP per algorithmic byte: input + output + residual / accumulate operands + weights) is at or above the
 * given ridge (peak FLOP/s over peak HBM B/s of the kernel family), out[cfg][4..7] of those below it.  RVC_PROF_CFGS * 8 doubles. */
int rvc_prof_collect_ex(double* out, double ridge_fp32, double ridge_bf16x3);
const char* rvc_prof_cfg_name(int i);
/* writes one CSV row per launch recorded since rvc_prof_enable(1): kernel, tile, Ci, Co, k, dilation, stride, Tout, workgroups, us,
 * algorithmic GFLOP / MB, TFLOP/s, GB/s (profiles/ per-launch-class tables) */
int rvc_prof_dump_csv(const char* path);
/* ------------------------------------------------------------------ experiment / instrumentation hooks
 * NOT part of the product ABI: librvc_hip.so exports them only when it is built with -DRVC_EXPERIMENTS (tools/build_variant.sh: variant builds for
 * A/B timing, wait-count checks, per-phase cycle counters).  The product build also compiles every experiment knob (tile choices, kernel
 * selection, thresholds) to its default; the run-time knobs it does read are listed in INTEGRATION.md. */
#ifdef RVC_EXPERIMENTS
/* debug builds only (-DRVC_CONV_TIMING): cycle sums {blocks, prologue, stage fill, prefetch issue, MFMA, epilogue, total, -}; zeros otherwise */
int rvc_debug_conv_timing(uint64_t* out8, int reset);
/* debug builds only (-DRVC_X3P_CHECK): number of waits of the pipelined bf16x3 kernel whose compile-time vmcnt exceeded the exact
 * run-time count since the last call (must be 0); -1 in ordinary builds */
int rvc_debug_x3p_check(void);
/* which reduction loop the split-resident GEMM / convolution (csrc/conv_x3s.hip) runs, process-wide: 0 = chosen by shape (default; RVC_X3S_MODE),
 * 1 = operands through the LDS ring (LDS-DMA, one barrier per unit), 2 = operands loaded straight into the MFMA registers (no LDS, no barrier).
 * Results are bit-identical between the two (same units, same MFMA order per accumulator); tools / tests A/B them in one process. */
int rvc_debug_set_x3s_mode(int mode);
/* kernel benchmark (tools/bench_gemm.py): `reps` back-to-back launches of the split-resident GEMM (csrc/conv_x3s.hip) for an [Co x Ci] layer
 * on T columns of device-resident random data (input image, fp32 output with a residual); ksplit / am / an as in rvc_op_gemm_split,
 * split_out != 0 writes the output as the split image (GELU epilogue) instead.  w2d > 0: a 3 x 3 convolution over a padded image of width w2d
 * (T = H (w2d + 2) positions).  nlayers distinct weight sets are cycled through (a model's layers are read once per clip: cold weights).
 * *us_out = mean microseconds per launch (HIP events). */
int rvc_debug_gemm_split_bench(void* stream, int Ci, int Co, int T, int ksplit, int am, int an, int split_out, int reps, float* us_out, int w2d,
                               int nlayers);

#endif /* RVC_EXPERIMENTS */

#ifdef __cplusplus
}
#endif
#endif /* RVC_HIP_H */
