// Launch wrappers of the non-GEMM device ops (ops.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace rvc {

// LayerNorm over channels that ALSO writes y as the split-resident image of conv_x3s.hip (y may be null: image only); margin = kSplitMargin
void wn_gate_split(hipStream_t s, const float* a, const float* g, unsigned char* img, long long tp, int margin, int H, int T);   // WN gate -> split image only
// HuBERT feature-encoder layer 0 fused: Conv1d(1, C, 10, stride 5) -> GroupNorm(C, C) -> GELU from the raw audio (the convolution is
// evaluated twice instead of being stored: ops.hip).  partial: hubert_conv0_scratch_doubles(C, T1) doubles, stat: 2 C floats.
void hubert_conv0_gn_gelu(hipStream_t s, const float* audio, long long L, const float* w, const float* gamma, const float* beta, int C, int T1, float eps,
                          float* out, long long ld, double* partial, float* stat);
// the same, the result written only as the DE-INTERLEAVED bf16 hi / lo image a stride-2 layer on the split-resident GEMM reads (split_geom_s2; H = split_s2_h(T1), tp = split_s2_tp(T1))
void hubert_conv0_gn_gelu_img(hipStream_t s, const float* audio, long long L, const float* w, const float* gamma, const float* beta, int C, int T1, float eps,
                              unsigned char* img, long long tp, int margin, int H, double* partial, float* stat);
size_t hubert_conv0_scratch_doubles(int C, int T1);
void layernorm_c_split(hipStream_t s, const float* x, const float* gamma, const float* beta, float* y, unsigned char* img, long long tp, int margin,
                       int C, int T, long long ld, float eps);
void layernorm_c(hipStream_t s, const float* x, const float* r, const float* gamma, const float* beta, float* y, int C, int T,
                 long long ld, float eps);
void groupnorm_t_gelu(hipStream_t s, float* x, const float* gamma, const float* beta, int C, int T, long long ld, float eps);
void softmax_cols(hipStream_t s, float* S, int Tk, int Tq, long long ld, long long batchS, int batch, const float* rel,
                  long long batchRel, int win, float* pb, long long batchPb);
void fill(hipStream_t s, float* p, float v, long long n);
void wn_gate(hipStream_t s, const float* a, const float* g, float* out, int H, int T);
void gemv(hipStream_t s, const float* W, const float* x, const float* b, float* y, int N, int K, const float* add);
void zp_sample(hipStream_t s, const float* stats, const float* noise, float* zp, int C, int T);
void flip_c(hipStream_t s, const float* x, float* y, int C, int T);
void encp_embed(hipStream_t s, float* x, const float* emb, const long long* pitch, int C, int T);
void conv_to1(hipStream_t s, const float* x, long long ldx, const float* w /*[Ci][K]*/, int Ci, int K, int pad, int T, float pre_slope,
              int act_tanh, float* y);
void resample(hipStream_t s, const float* x, long long n_in, const double* h, int half, int U, int D, float* y, long long n_out);
void transpose(hipStream_t s, const float* in, float* out, int R, int C, long long ldin, long long ldout, int batch, long long bin,
               long long bout);
void feats_prepare(hipStream_t s, const float* f, const float* f0, const float* pitchf, float* out, int D, int Th, int T, float protect,
                   int do_protect);
// x[c][t] += b[c] + sum_j w[c][j] src[t stride + j - pad] for k = 1 / 4 / 8 taps of one source channel (16-byte rows); false: not its shape
bool noise_add(hipStream_t s, float* x, long long ld, int C, int T, const float* src, long long L, int k, int stride, int pad, const float* w, const float* b);
void f0_post(hipStream_t s, const double* f0, long long n, double factor, double mel_min, double mel_max, int bins, long long* pitch, float* pitchf);
void frames(hipStream_t s, const float* src, float* out, int L, int k, int stride, int pad, int Tout, int reflect);
void magnitude(hipStream_t s, const float* ft, float* mag, int F, int T);
void mel_to_unet(hipStream_t s, const float* mel, float* x, int n, int Tr, float a, float b);
void avgpool2(hipStream_t s, const float* x, float* y, int C, int H, int W, long long ldx);
void gru_scan(hipStream_t s, const float* gi, const float* b_ih, const float* w_hh, const float* w_hh_t /* k-major copy for the repair kernel */, const float* b_hh, float* out,
              unsigned long long* xbuf, int* err, int T, unsigned spin_limit = 0, int fault = 0);
void rmvpe_decode(hipStream_t s, const float* sal, double* f0, int n, long long ld, float thred, const int* err = nullptr);
void sine_source(hipStream_t s, const float* f0, const float* noise, float* har, float* sine_out, float* rad, float* tmp, double* bsum,
                 int T, int upp, float sr, float lw, float lb, float* phase_out = nullptr);

size_t preprocess_scratch_doubles(long long n, bool have_sos);   // doubles of scratch preprocess() needs for an input of n samples
void preprocess(hipStream_t s, const void* x, int is64, long long n, const double* b, const double* a, const double* zi, int t_pad,
                double* filt, float* padded, double* rms1, int n1, int frame, int hop, double* scratch, const double* sos = nullptr,
                const double* sos_zi = nullptr);      // sos [3][6] + sosfilt_zi [3][2]: block-propagated cascade evaluation (ops.hip)
void postprocess(hipStream_t s, float* x, long long N, const double* rms1, int n1, int sr2, float rate, short* out, float* rms2, unsigned* maxbits);

// padded 2-D split-resident images (split2d.hip): level changes of RMVPE's U-Net in the layout conv_x3s.hip convolves
void pool2_pad_split(hipStream_t s, const float* x, long long ldx, bool x_padded, int C, int H, int W, float* y, long long ldy, unsigned char* img,
                     long long tp, int margin);
void interleave2_pad_split(hipStream_t s, const float* ph, long long ldp, int Co, int H, int W, float* y, long long ldy, bool y_padded, unsigned char* img,
                           long long tp, int margin);
void pad2d_split(hipStream_t s, const float* x, long long ldx, int C, int H, int W, float* y, long long ldy, unsigned char* img, long long tp, int margin);
void unpad2d(hipStream_t s, const float* x, long long ldx, int C, int H, int W, float* y, long long ldy);

}  // namespace rvc
