// Internal C++ interface of the three network graphs; the extern "C" layer (rvc_api.hip) wraps these.
#pragma once
#include "model_common.h"
#include "../../include/rvc_hip.h"

namespace rvc {

typedef rvc_synth_config SynthConfig;
typedef rvc_synth_taps SynthTaps;
typedef rvc_hubert_taps HubertTaps;
typedef rvc_rmvpe_taps RmvpeTaps;
typedef rvc_crepe_taps CrepeTaps;

struct Synth;
Synth* synth_create(Ctx* ctx, const SynthConfig& c);
void synth_destroy(Synth* S);
void synth_set_tensor(Synth* S, const char* name, const float* d, const long long* shape, int ndim);
void synth_finalize(Synth* S);
int synth_upp(const Synth* S);
int synth_feat_dim(const Synth* S);
bool synth_has_f0(const Synth* S);      // false: *_nono family (decided by the checkpoint: no enc_p.emb_pitch)
void synth_infer(Synth* S, hipStream_t s, const float* feat, int feat_channel_major, const long long* pitch, const float* pitchf, int sid,
                 const float* noise_z, const float* noise_src, int T, float* out, const SynthTaps* taps);

struct Hubert;
Hubert* hubert_create(Ctx* ctx);
void hubert_destroy(Hubert* H);
void hubert_set_tensor(Hubert* H, const char* name, const float* d, const long long* shape, int ndim);
void hubert_finalize(Hubert* H);
long long hubert_num_frames(long long L);
// out_rm: [T_h][D] row-major (the reference's [1,T_h,D]) or null; out_cm: channel-major [D][T_h] or null
void hubert_forward(Hubert* H, hipStream_t s, const float* audio, long long L, int version, int n_layers, float* out_rm, float* out_cm,
                    const HubertTaps* taps);

struct Rmvpe;
Rmvpe* rmvpe_create(Ctx* ctx);
void rmvpe_destroy(Rmvpe* R);
void rmvpe_set_tensor(Rmvpe* R, const char* name, const float* d, const long long* shape, int ndim);
void rmvpe_finalize(Rmvpe* R);
// mel_out [128][n], salience_out [n][360], f0_out [n] (float64); any may be null.  A failed GRU hand-off (the scan's workgroups poll each
// other) sets a sticky device flag: f0_out is then NaN and rmvpe_status reports it.
void rmvpe_forward(Rmvpe* R, hipStream_t s, const float* audio, long long L, float thred, float* mel_out, float* salience_out, double* f0_out,
                   const RmvpeTaps* taps);

// feature retrieval (index.hip): exact nearest neighbour over big_npy [N][D]
struct FeatIndex;
FeatIndex* index_create(Ctx* ctx, const float* big_npy, long long N, int D);
// the reference's own index type (faiss IndexIVFFlat): nprobe nearest centroids, then the nearest vector inside those cells only
FeatIndex* index_create_ivf(Ctx* ctx, const float* big_npy, long long N, int D, const float* centroids, int nlist, const int* list_of, int nprobe);
int index_nprobe(const FeatIndex* I);   // 0: exact search
void index_destroy(FeatIndex* I);
long long index_size(const FeatIndex* I);
int index_dim(const FeatIndex* I);
void index_search(FeatIndex* I, hipStream_t s, const float* feats_cm, int T, long long* idx, float* score);
void index_blend(FeatIndex* I, hipStream_t s, const float* feats_cm, const long long* idx, int T, float rate, float* out_cm);

void rmvpe_decode_rm(Rmvpe* R, hipStream_t s, const float* sal_rm, long long n, float thred, double* f0);
int rmvpe_status(Rmvpe* R, hipStream_t s);                 // waits for the stream; bit 0: the last forward's GRU scan timed out
void rmvpe_debug_fault(Rmvpe* R, int fault, unsigned spin_limit);   // tests: make the next scans fail / shorten their spin limit
size_t synth_workspace(const Synth* S);
size_t hubert_workspace(const Hubert* H);
size_t rmvpe_workspace(const Rmvpe* R);

// MDX23C separation network (model_mdx23.hip): one chunk [2][hop * (dim_t - 1)] -> [S][2][chunk]
struct Mdx23;
Mdx23* mdx23_create(Ctx* ctx, const rvc_mdx23_config& c);
void mdx23_destroy(Mdx23* M);
void mdx23_set_tensor(Mdx23* M, const char* name, const float* d, const long long* shape, int ndim);
void mdx23_finalize(Mdx23* M);
void mdx23_forward(Mdx23* M, hipStream_t s, const float* audio, long long L, float* out);
void mdx23_demix(Mdx23* M, hipStream_t s, const float* mix, long long Lp, long long step, long long n_chunks, float overlap, float* acc);
void mdx23_set_streams(Mdx23* M, int k);
size_t mdx23_workspace(const Mdx23* M);

// CREPE pitch network (model_crepe.hip): probabilities [360][n] for n = crepe_num_frames(L, hop, pad) frames
struct Crepe;
Crepe* crepe_create(Ctx* ctx, int tiny);
void crepe_destroy(Crepe* M);
void crepe_set_tensor(Crepe* M, const char* name, const float* d, const long long* shape, int ndim);
void crepe_finalize(Crepe* M);
long long crepe_num_frames(long long L, int hop, int pad);
void crepe_forward(Crepe* M, hipStream_t s, const float* audio, long long L, int hop, int pad, float* probs, const CrepeTaps* taps);
size_t crepe_workspace(const Crepe* M);
void crepe_viterbi(hipStream_t s, const float* probs, int n, int lo, int hi, int* bins, float* per);   // torchcrepe.decode.viterbi + periodicity

}  // namespace rvc
