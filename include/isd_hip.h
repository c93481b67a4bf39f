/*
 * isd_hip.h -- C ABI of libisd_hip.so: the MI355X (gfx950) hot path of the
 * EEG imagined-speech decoder.
 *
 * The reference (kidusabe1/Imagined-Speech-Decoding) is pure Python and has no
 * FFI; its boundaries for this path are duck-typed Python call sites
 * (SURVEY.md 8b).  Each entry point below names the reference call it
 * replaces.  Conventions (all functions):
 *   - plain pointers + sizes only, no torch types; every data pointer is a
 *     DEVICE pointer owned by the caller unless the name says "host";
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - nothing synchronises the host, allocates device memory or launches
 *     threads inside a *_forward / *_backward call (graph-capturable);
 *   - return 0 on success, a negative ISD_ERR_* otherwise; the message of the
 *     last failure on the calling thread is isd_last_error();
 *   - plans are opaque, immutable after create, shareable across streams.
 */
#ifndef ISD_HIP_H
#define ISD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ISD_ABI_VERSION 1

enum {
  ISD_OK = 0,
  ISD_ERR_INVALID = -1,     /* bad argument / shape */
  ISD_ERR_UNSUPPORTED = -2, /* valid request this build cannot run */
  ISD_ERR_HIP = -3,         /* HIP runtime error (message has hipGetErrorString) */
  ISD_ERR_NO_DEVICE = -4    /* no gfx950 device visible */
};

int isd_abi_version(void);
const char* isd_last_error(void);
/* Measurement aid (bench.py): one wave reads the shader-clock counter and the constant-rate counter around a spin of
 * about `spin_us` microseconds and stores {shader ticks, constant-rate ticks} as two uint64 at `out` (device memory);
 * shader MHz = ticks[0] / ticks[1] * isd_wall_clock_khz() / 1000.  Launch it on a second stream beside the kernels
 * being timed. */
int isd_shader_clock_probe(uint64_t* out, int spin_us, void* stream);
int isd_wall_clock_khz(void);
/* The exact, order-independent sum of n fp32 values rounded once to fp64: the accumulator the BatchNorm heads keep
 * their batch sums in (64-bit integer atomics on a fixed-point image of the partials; the same bits on every run --
 * src/fast/utils.py:104-114 asks for deterministic training).  `workspace`: isd_exact_sum_workspace_bytes() bytes. */
int64_t isd_exact_sum_workspace_bytes(void);
int isd_exact_sum(const float* x, int64_t n, double* out, void* workspace, void* stream);
/* number of visible HIP devices (0 on a CPU-only host; never fails) */
int isd_device_count(void);

/* ------------------------------------------------------------------------
 * Filterbank: IIR band-pass cascade, one Butterworth SOS cascade per band.
 * Replaces (spec S step 1-2) the scipy path
 *   butter(order, band, 'bandpass', fs, output='sos') ; sosfilt(sos, x, axis=-1)
 * anchored on notebooks/svm_baseline.ipynb:238 (the repo's only band-pass).
 * Sections are given in "resonator form": every section is
 *   H(z) = (1 - z^-2) / (1 + a1 z^-1 + a2 z^-2)   and one gain per band,
 * which is what a Butterworth band-pass factors into (zeros at z = +1, -1).
 * ---------------------------------------------------------------------- */
typedef struct isd_fb_plan isd_fb_plan;

enum { ISD_FB_F32 = 0, ISD_FB_F64 = 1, ISD_FB_AUTO = 2, ISD_FB_MIXED = 3 /* reported only */ };

/* a12  : host, [n_bands][n_sections][2] doubles (a1, a2)
 * gain : host, [n_bands] doubles
 * precision: arithmetic of the in-chunk recursion (the cross-chunk state scan
 *            is always fp64); AUTO decides PER BAND: a band whose poles are too
 *            close to z=1 for fp32 to keep 1e-4 runs in fp64, the others in fp32
 *            (see DESIGN.md).  n_sections <= 8. */
int isd_fb_plan_create(isd_fb_plan** out, int n_bands, int n_sections,
                       const double* a12, const double* gain, int precision);
int isd_fb_plan_destroy(isd_fb_plan* plan);
int isd_fb_plan_precision(const isd_fb_plan* plan); /* resolved ISD_FB_F32 / F64 / MIXED */

/* x [B][C][T] f32  ->  y [B][n_bands][C][T] f32   (zero initial state, causal) */
int isd_fb_forward(const isd_fb_plan* plan, const float* x, float* y,
                   int64_t B, int64_t C, int64_t T, void* stream);

/* ------------------------------------------------------------------------
 * STFT with the scipy-legacy defaults used at
 *   scripts/global_shap_analysis.py:132  scipy.signal.stft(sig, fs, nperseg=64, noverlap=32)
 * (periodic Hann, zero extension by nperseg/2, zero padding to whole frames,
 * one-sided rfft, scaling='spectrum'), and the inclusive band aggregation of
 *   scripts/global_shap_analysis.py:151-156.
 * nperseg must be a power of two in [8, 4096]; 0 <= noverlap < nperseg.
 * ---------------------------------------------------------------------- */
typedef struct isd_stft_plan isd_stft_plan;

int isd_stft_plan_create(isd_stft_plan** out, int T, int nperseg, int noverlap);
int isd_stft_plan_destroy(isd_stft_plan* plan);
int isd_stft_plan_frames(const isd_stft_plan* plan); /* J */
int isd_stft_plan_bins(const isd_stft_plan* plan);   /* nperseg/2 + 1 */

/* x [R][T] f32 -> Z [R][bins][J] complex64 (interleaved re,im), scipy layout */
int isd_stft_forward(const isd_stft_plan* plan, const float* x, float* Z,
                     int64_t R, void* stream);

enum {
  ISD_BP_MAGNITUDE = 0, /* mean |Z|        (global_shap_analysis.py:135,156) */
  ISD_BP_POWER = 1,     /* mean |Z|^2                                        */
  ISD_BP_LOGPOWER = 2   /* log(mean |Z|^2 + eps)   (spec S steps 4-5)        */
};

/* Band aggregation of per-band filtered signals.
 * y [B][n_bands][C][T] f32 -> feat [B][n_bands][C][J] f32; band b of the
 * input is reduced over its own bins klo[b]..khi[b] (inclusive, host int
 * arrays; khi < klo = empty band -> 0 before the log).
 * With n_bands_in == 1 the same signal y [B][1][C][T] is reduced over every
 * band (the reference's use: one trace, five bands) -> feat [B][n_bands][C][J]. */
int isd_stft_bandpower(const isd_stft_plan* plan, const float* y, float* feat,
                       int64_t B, int64_t C, int n_bands_in, int n_bands,
                       const int* klo, const int* khi, int mode, float eps, void* stream);

/* Fused spec-S feature extractor: filterbank -> STFT -> log band power without
 * materialising the filtered signals.  x [B][C][T] -> feat [B][n_bands][C][J].
 * Requires nperseg == 64, noverlap == 32 (the reference's STFT parameters). */
int isd_features_fused(const isd_fb_plan* fb, const isd_stft_plan* st, const float* x,
                       float* feat, int64_t B, int64_t C, const int* klo, const int* khi,
                       int mode, float eps, void* stream);
/* The same with the feature map written as bf16 (round to nearest even) -- BASELINE config 3: it is the rounding the
 * bf16 classifier (isd_featcnn_step_bf16) and the reference's autocast apply to the convolution's input anyway.
 * Rows of at most 1024 samples (the nperseg 64 / noverlap 32 extractor); ISD_ERR_UNSUPPORTED otherwise. */
int isd_features_fused_bf16(const isd_fb_plan* fb, const isd_stft_plan* st, const float* x, uint16_t* feat,
                            int64_t B, int64_t C, const int* klo, const int* khi, int mode, float eps, void* stream);

/* Which kernel family the calling thread's last isd_features_fused / _bf16 call launched for its fp32 bands
 * (diagnostic; the tests that hold the two extractors against each other read it):
 *   0 none yet, 1 sixteen lanes per row (fused_kernel), 2 one row per lane (fused_serial_kernel: rows of whole
 *   32-sample chunks, bands of one or two bins; ISD_FUSED_SERIAL=0 in the environment selects 1), 3 the long-row
 *   block-sum kernels. */
int isd_features_fused_last_path(void);

/* ------------------------------------------------------------------------
 * Zero-phase FIR filter.  Replaces the band-pass of the SVM baseline,
 *   mne.filter.filter_data(X, 250, l_freq=4, h_freq=40)   notebooks/svm_baseline.ipynb:238-239, :968-969
 * (MNE is a third-party dependency that is not vendored in the reference; its documented defaults are restated:
 * method 'fir', fir_design 'firwin', hamming window, phase 'zero', pad 'reflect_limited').
 * taps: host, [n_taps] doubles, n_taps odd and symmetric (linear-phase type I); the host-side design lives in
 * isd_amd.filter_design.fir_design.  y[r][n] = sum_k taps[k] * xe[r][n - (n_taps-1)/2 + k] where xe is the row
 * extended by odd reflection about its end points over min(n_taps, T) - 1 samples and zeros beyond.
 * x, y [rows][T], distinct buffers.  _f32 computes in fp32, _f64 in fp64 (the notebook filters float64).
 * ---------------------------------------------------------------------- */
typedef struct isd_fir_plan isd_fir_plan;
int isd_fir_plan_create(isd_fir_plan** out, int n_taps, const double* taps);
int isd_fir_plan_destroy(isd_fir_plan* plan);
int isd_fir_plan_taps(const isd_fir_plan* plan);
int isd_fir_zero_phase_f32(const isd_fir_plan* plan, const float* x, float* y, int64_t rows, int T, void* stream);
int isd_fir_zero_phase_f64(const isd_fir_plan* plan, const double* x, double* y, int64_t rows, int T, void* stream);

/* ------------------------------------------------------------------------
 * Zone-wise Conv4Layers stack over sliding windows: the reference's
 *   FAST.forward_head   src/fast/models/fast.py:242-252  (unfold window_len / slide_step)
 *   Head.forward        src/fast/models/fast.py:209-210  (zone gather, one encoder per zone, stack)
 *   Conv4Layers.forward src/fast/models/fast.py:111-119  (cnn1..cnn4, exact GELU, mean over time)
 * x [B][c_total][T] f32 -> feat [B*N][n_zones][F] f32 (== the reference's [B, N, Z, F]),
 * N = (T - window_len) / slide_step + 1.  The spec-S feature classifier uses the same
 * entry points with one zone of nb*C "channels" and window_len == T == J.
 *
 * Parameters live in ONE flat f32 block (so that data-parallel training
 * all-reduces a single bucket): per zone, in the reference's state_dict order,
 *   cnn1.weight [F,1,1,5] | cnn1.bias [F] | cnn2.weight [F,F,Cz,1] | cnn3.weight [F,F,1,5] | cnn4.weight [F,F,1,5]
 * (n_layers == 2 drops cnn3/cnn4: the build-defined "2-layer CNN" of BASELINE config 1).
 * ---------------------------------------------------------------------- */
typedef struct isd_conv4_plan isd_conv4_plan;

/* zone_sizes: host [n_zones]; zone_channels: host, concatenated channel indices (Head.index_dict, fast.py:206) */
int isd_conv4_plan_create(isd_conv4_plan** out, int c_total, int n_zones, const int* zone_sizes,
                          const int* zone_channels, int feature_dim, int n_layers, int window_len,
                          int slide_step);
int isd_conv4_plan_destroy(isd_conv4_plan* plan);
/* BASELINE config 3: ISD_ACT_BF16 stores activations and activation gradients as bf16 and rounds the staged
 * operands (inputs, weights) to bf16, accumulating in fp32 -- the arithmetic of bf16-mixed autocast
 * (scripts/train_fast.py:277).  Parameters, parameter gradients, features and the FC head stay fp32. */
enum { ISD_ACT_F32 = 0, ISD_ACT_BF16 = 1 };
int isd_conv4_plan_set_activation_dtype(isd_conv4_plan* plan, int dtype);
int64_t isd_conv4_param_count(const isd_conv4_plan* plan);
/* which: 0 cnn1.weight, 1 cnn1.bias, 2 cnn2.weight, 3 cnn3.weight, 4 cnn4.weight -> offset in floats */
int64_t isd_conv4_param_offset(const isd_conv4_plan* plan, int zone, int which);
int isd_conv4_windows(const isd_conv4_plan* plan, int64_t T);
int64_t isd_conv4_workspace_bytes(const isd_conv4_plan* plan, int64_t B, int64_t T);

/* forward keeps the activations the backward needs inside `workspace` (caller-owned,
 * isd_conv4_workspace_bytes bytes); backward must see the same x, params and workspace.
 * dparams: flat gradient block, same layout as params, overwritten. */
int isd_conv4_forward(const isd_conv4_plan* plan, const float* x, const float* params, float* feat,
                      void* workspace, int64_t B, int64_t T, void* stream);
int isd_conv4_backward(const isd_conv4_plan* plan, const float* x, const float* params,
                       const float* dfeat, float* dparams, void* workspace, int64_t B, int64_t T,
                       void* stream);
/* As isd_conv4_backward, and additionally dx [B][c_total][T] = dL/dx (zeroed here, then accumulated over zones and
 * overlapping windows).  Takes the layer-wise backward (the fused one keeps the activation gradients in LDS); fp32
 * activations only.  Off the hot path: it serves input attributions (scripts/explain_fast.py,
 * scripts/global_shap_analysis.py differentiate the model w.r.t. its input). */
int isd_conv4_backward_x(const isd_conv4_plan* plan, const float* x, const float* params, const float* dfeat,
                         float* dparams, float* dx, void* workspace, int64_t B, int64_t T, void* stream);

/* ------------------------------------------------------------------------
 * Dense head on the matrix cores (fp32-in MFMA).  nn.Linear semantics:
 *   y[M][N] = act(x[M][K] . w[N][K]^T + bias[N]),  act: 0 none, 1 exact-erf GELU.
 * Replaces FAST.input_layer (Linear 256->32 + GELU, fast.py:235) and
 * FAST.last_layer (Linear 32->5, fast.py:239) as used at fast.py:276-277.
 * `pre` (nullable) receives the pre-activation, which backward needs when act == 1.
 * ---------------------------------------------------------------------- */
int isd_linear_forward(const float* x, const float* w, const float* bias, float* y, float* pre,
                       int64_t M, int K, int N, int act, void* stream);
/* y = x . w^T + bias + res   (residual connection fused in the epilogue; fast.py:24-25) */
int isd_linear_residual_forward(const float* x, const float* w, const float* bias, const float* res, float* y,
                                int64_t M, int K, int N, void* stream);
int64_t isd_linear_workspace_bytes(int64_t M, int K, int N);
/* dx nullable; db nullable */
int isd_linear_backward(const float* x, const float* w, const float* dy, const float* pre, float* dx,
                        float* dw, float* db, void* workspace, int64_t M, int K, int N, int act,
                        void* stream);

/* ------------------------------------------------------------------------
 * Token mean + softmax cross-entropy + argmax:
 *   logits = logits_tok.mean(dim=1)                         fast.py:277
 *   loss   = nn.CrossEntropyLoss()(logits, y)               trainer.py:37,59  (mean; uint8 or int64 labels)
 *   pred   = argmax(logits, dim=1), ties -> lowest index    trainer.py:89
 * logits_tok [B][n_tok][n_cls]; grad_scale = 1 / global batch size (1/B on one GPU);
 * loss receives grad_scale * sum_b CE_b; dlogits_tok gets d loss / d logits_tok.
 * labels == NULL: inference (no loss / gradient).  Any output pointer may be NULL.
 * workspace: isd_softmax_ce_workspace_bytes(B) bytes of scratch (per-block partial sums + an
 * arrival ticket; contents need not be initialised); only needed when the loss is requested.
 * ---------------------------------------------------------------------- */
int64_t isd_softmax_ce_workspace_bytes(int64_t B);
int isd_softmax_ce(const float* logits_tok, const void* labels, int label_bytes, float* logits_mean,
                   float* loss, float* dlogits_tok, int64_t* pred, int64_t B, int n_tok, int n_cls,
                   float grad_scale, void* workspace, void* stream);

/* ------------------------------------------------------------------------
 * EEGNet_Encoder (src/fast/models/fast.py:122-167): the "EEGNet-style depthwise CNN" head.
 *   x [B][C][T] f32 -> out [B][feature_dim];  kernel_length even, <= 64 (reference default 64).
 * Flat parameter block (reference state_dict order, trainable tensors only):
 *   temporal_conv.0.weight [8,1,1,K] | temporal_conv.1.weight [8] | .bias [8] | spatial_conv.0.weight [16,1,C,1] |
 *   spatial_conv.1.weight [16] | .bias [16] | separable_conv.0.weight [16,1,1,16] | separable_conv.1.weight [16,16,1,1] |
 *   separable_conv.2.weight [16] | .bias [16] | projector.2.weight [F,16] | projector.2.bias [F]
 * Buffer block (80 floats): running_mean/var of BN1 (8,8), BN2 (16,16), BN3 (16,16); updated in place when training.
 * training != 0: batch statistics (biased variance for the normalisation, unbiased for the running update);
 * dropout_p (train only) uses a counter-based mask keyed by `seed` (pass the same p and seed to backward): it is
 * statistically, not bitwise, nn.Dropout.  The temporal-conv activation [B,8,C,T+1] is never formed.
 * backward: parameter gradients only (x is data); one backward per forward, same workspace.
 * ---------------------------------------------------------------------- */
typedef struct isd_eegnet_plan isd_eegnet_plan;
int isd_eegnet_plan_create(isd_eegnet_plan** out, int in_channels, int feature_dim, int kernel_length, int T);
/* CVBlock (src/fast/models/fast.py:32-100): the same plan type and the same isd_eegnet_* entry points below, with
 * stage 1 pooled by 8, a full 16->16 (1,16) convolution in stage 2 pooled by 2, and Linear(16*T3 -> dim_token) over
 * the flattened map (T3 = 16 for the reference's 250-sample window; isd_cvblock_flat_dim returns 16*T3).
 * Flat parameter block: conv1.weight [8,1,1,64] | bn1.weight [8] | bn1.bias [8] | conv2.weight [16,1,C,1] |
 *   bn2.weight [16] | bn2.bias [16] | conv3.weight [16,16,1,16] | bn3.weight [16] | bn3.bias [16] |
 *   projector.weight [F, 16*T3] | projector.bias [F];  buffer block as above (bn1, bn2, bn3). */
int isd_cvblock_plan_create(isd_eegnet_plan** out, int in_channels, int dim_token, int T);
int64_t isd_cvblock_flat_dim(const isd_eegnet_plan* plan);
/* Optional device-resident dropout step counter (uint64): when set, every pass mixes its current value into the dropout
 * seed, so a captured HIP graph -- which replays the same seed argument -- draws new masks by incrementing the counter
 * inside the graph.  Null (the default) = the seed argument alone. */
int isd_eegnet_plan_set_seed_counter(isd_eegnet_plan* plan, const uint64_t* seed_dev);
int isd_eegnet_plan_destroy(isd_eegnet_plan* plan);
int64_t isd_eegnet_param_count(const isd_eegnet_plan* plan);
int64_t isd_eegnet_buffer_count(const isd_eegnet_plan* plan);
int64_t isd_eegnet_workspace_bytes(const isd_eegnet_plan* plan, int64_t B);
int isd_eegnet_forward(const isd_eegnet_plan* plan, const float* x, const float* params, float* buffers,
                       float* out, void* workspace, int64_t B, int training, float momentum, float eps,
                       float dropout_p, uint64_t seed, void* stream);
int isd_eegnet_backward(const isd_eegnet_plan* plan, const float* x, const float* params, const float* dout,
                        float* dparams, void* workspace, int64_t B, float dropout_p, uint64_t seed,
                        void* stream);
/* The same plus the gradient w.r.t. the input trials, dx [B][C][T] (fast.py:203-210 asks for an autograd-differentiable
 * encoder; the attribution scripts differentiate the network w.r.t. x).  `training` = the value the forward was called
 * with: 1 (batch statistics: BatchNorm's mean / variance paths reach dx) or 2 (isd_eegnet_forward(..., training = 2):
 * eval-mode statistics with the activations kept for this call -- what attribution methods use).  Single device. */
int isd_eegnet_backward_x(const isd_eegnet_plan* plan, const float* x, const float* params, const float* dout,
                          float* dparams, float* dx, void* workspace, int64_t B, int training, float dropout_p,
                          uint64_t seed, void* stream);
/* Synchronised BatchNorm under data parallelism (the reference's heads use plain nn.BatchNorm2d on one
 * device, fast.py:46-63,133-159, so the single-device batch statistics are what a sharded batch must reproduce).
 * The forward and backward passes above are four stages each (0..3); after stages 0, 1 and 2 a block of fp64 batch
 * sums is complete in the workspace (isd_eegnet_sync_block gives its byte offset and length).  The caller runs the
 * stages one by one with the same arguments on every rank, all-reduces (SUM) that block between two stages, and
 * passes the number of ranks (equal shards): the statistics are then those of the global batch, and the parameter
 * gradients sum to the single-device gradient under the usual gradient all-reduce.  world = 1 without all-reduces is
 * isd_eegnet_forward / isd_eegnet_backward.  Evaluation (training == 0) needs no exchange. */
int isd_eegnet_forward_stage(const isd_eegnet_plan* plan, int stage, const float* x, const float* params,
                             float* buffers, float* out, void* workspace, int64_t B, int training, float momentum,
                             float eps, float dropout_p, uint64_t seed, int world, void* stream);
int isd_eegnet_backward_stage(const isd_eegnet_plan* plan, int stage, const float* x, const float* params,
                              const float* dout, float* dparams, void* workspace, int64_t B, float dropout_p,
                              uint64_t seed, int world, void* stream);
int isd_eegnet_sync_block(const isd_eegnet_plan* plan, int64_t B, int backward, int stage, int64_t* byte_offset,
                          int64_t* n_doubles);
/* What the 8-byte words of that block are: 0 = fp64 sums, 1 = the int64 words of exact accumulators (the partial sums of
 * the workgroups are added with integer atomics so that a pass gives the same bits on every run -- the reference's
 * cudnn.deterministic, src/fast/utils.py:104-114); all-reduce the block with that element type (SUM either way). */
int isd_eegnet_sync_block_kind(int backward, int stage);

/* ------------------------------------------------------------------------
 * Transformer tail of FAST (src/fast/models/fast.py:10-29 AttentionBlock, :260-268 forward_transformer).
 * The dense projections are isd_linear_* launches over the [B*S, D] token matrix; these are the pieces around them.
 *   embed:      tok[b,0] = cls + pos[0];  tok[b,1+n] = x[b,n] + pos[1+n]                 (fast.py:263-265)
 *   layernorm:  nn.LayerNorm(D), D <= 64; stats [M][2] = (mean, rstd) saved for backward (fast.py:11,13)
 *   attention:  nn.MultiheadAttention core, batch_first: qkv [B][S][3D] (q|k|v) -> ctx [B][S][D], probs [B][H][S][S];
 *               S <= 8 tokens, head_dim <= 8; attention dropout through a counter-based mask (seed)   (fast.py:12,23)
 * ---------------------------------------------------------------------- */
int isd_embed_forward(const float* x, const float* cls, const float* pos, float* tok, int64_t B, int N, int D,
                      void* stream);
int isd_embed_backward(const float* dtok, float* dx, float* dcls, float* dpos, int64_t B, int N, int D,
                       void* stream);
int isd_layernorm_forward(const float* x, const float* w, const float* b, float* y, float* stats, int64_t M, int D,
                          float eps, void* stream);
int isd_layernorm_backward(const float* x, const float* w, const float* dy, const float* stats, float* dx,
                           float* dw, float* db, int64_t M, int D, void* stream);
int isd_attention_forward(const float* qkv, float* ctx, float* probs, int64_t B, int S, int H, int head_dim,
                          float dropout_p, uint64_t seed, void* stream);
int isd_attention_backward(const float* qkv, const float* probs, const float* dctx, float* dqkv, int64_t B, int S,
                           int H, int head_dim, float dropout_p, uint64_t seed, void* stream);

/* ----------------------------------------------------------------------
 * The whole transformer tail in ONE launch per direction (replaces the per-operator sequence above for
 * FAST.forward_transformer, src/fast/models/fast.py:260-268 with AttentionBlock :10-29 -- the mode the reference
 * trains, src/fast/train/trainer.py:58):
 *   tokens = cat(cls, tokin) + pos[:N+1] -> L x [x += MHA(LN1 x); x += drop(W2 drop(gelu(W1 LN2 x)))] ->
 *   logits = last_layer(drop(x[:, 0])).
 * params: ONE flat f32 block,  pos_embedding [n_pos][D] | cls_token [D] | per block: layer_norm_1.{weight,bias} |
 *   attn.in_proj_{weight [3D][D], bias} | attn.out_proj.{weight,bias} | layer_norm_2.{weight,bias} |
 *   linear.0.{weight [2D][D], bias} | linear.3.{weight [D][2D], bias} | last_layer.{weight [n_cls][D], bias}
 *   (isd_tail_fused_param_count floats).  tokin [B][N][D] = input_layer's output; logits [B][n_cls].
 * Training: `save` (isd_tail_fused_save_floats(B, N+1, D, L) floats) and xfinal [B][D] receive what the backward
 *   needs; both null for inference (then every dropout probability must be 0).  Dropout masks are counter-based:
 *   pass the same seed to the backward.  seed_dev (may be null) points at a device-resident step counter that is
 *   mixed into the seed at launch: a captured HIP graph replays the same arguments, and advances its masks by
 *   incrementing that counter inside the graph.
 * Backward: dlogits [B][n_cls] -> dtokin [B][N][D] and dparams (the block's layout, overwritten; positional rows past
 *   N+1 get zeros); workspace = isd_tail_fused_workspace_floats floats (one partial block per wave, summed in a
 *   fixed order: deterministic).
 * Needs dim_token 16 or 32, hidden = 2 dim_token, N + 1 <= 8 tokens, head width 4 or 8, L <= 8, n_cls <= 16
 * (isd_tail_fused_supported; the per-operator entry points cover everything else).
 * ---------------------------------------------------------------------- */
int isd_tail_fused_supported(int N, int D, int H, int L, int hidden, int n_cls);
int64_t isd_tail_fused_param_count(int n_pos, int D, int L, int n_cls);
int64_t isd_tail_fused_save_floats(int64_t B, int S, int D, int L);
int64_t isd_tail_fused_workspace_floats(int64_t B, int N, int n_pos, int D, int L, int n_cls);
int isd_tail_fused_forward(const float* params, const float* tokin, float* logits, float* save, float* xfinal,
                           int64_t B, int N, int n_pos, int D, int H, int L, int hidden, int n_cls, float p_attn,
                           float p_mlp, float p_cls, uint64_t seed, const uint64_t* seed_dev, void* stream);
int isd_tail_fused_backward(const float* params, const float* save, const float* xfinal, const float* dlogits,
                            float* dtokin, float* dparams, float* workspace, int64_t B, int N, int n_pos, int D, int H,
                            int L, int hidden, int n_cls, float p_attn, float p_mlp, float p_cls, uint64_t seed,
                            const uint64_t* seed_dev, void* stream);

/* ----------------------------------------------------------------------
 * Whole classifier step on spec-S features in one call (the build-defined classifier of SURVEY 8d:
 * Conv4Layers(nb*C, 32) -> Linear(32, n_cls) -> softmax cross-entropy; replaces the call sequence
 * isd_conv4_forward / isd_linear_forward / isd_softmax_ce / isd_linear_backward / isd_conv4_backward).
 *   x [B][c_total][T] f32 features, params = the conv4 flat block, fc_w [n_cls][32], fc_b [n_cls].
 *   labels (uint8 / int64, may be null: inference): loss = sum_b nll_b * grad_scale, gradients scaled alike.
 *   dparams (conv4 layout) and dfc ([n_cls][32] then [n_cls]) may be null: forward + loss only.
 * Needs one zone, 4 layers, 32 filters, fp32 activations, T - 4 <= 16 output steps, n_cls <= 16
 * (isd_featcnn_supported); workspace = isd_conv4_workspace_bytes(plan, B, T).
 * ---------------------------------------------------------------------- */
int isd_featcnn_supported(const isd_conv4_plan* plan, int64_t B, int64_t T, int n_cls);
int isd_featcnn_step(const isd_conv4_plan* plan, const float* x, const float* params, const float* fc_w,
                     const float* fc_b, const void* labels, int label_bytes, float* dparams, float* dfc,
                     float* logits, int64_t* pred, float* loss, void* workspace, int64_t B, int64_t T, int n_cls,
                     float grad_scale, void* stream);
/* BASELINE config 3 with the feature map itself in bf16: x [B][c_total][T] bf16 as written by
 * isd_features_fused_bf16 (the first layer rounds an fp32 map to bf16 as it packs its operands: the results are bit
 * for bit those of isd_featcnn_step on the same values).  The plan must have bf16 activations and a multiple of 8
 * input channels; ISD_ERR_UNSUPPORTED otherwise. */
int isd_featcnn_step_bf16(const isd_conv4_plan* plan, const uint16_t* x, const float* params, const float* fc_w,
                          const float* fc_b, const void* labels, int label_bytes, float* dparams, float* dfc,
                          float* logits, int64_t* pred, float* loss, void* workspace, int64_t B, int64_t T,
                          int n_cls, float grad_scale, void* stream);

/* One AdamW step on a flat fp32 block (replaces the optimizer of the reference's training loop,
 * src/fast/train/trainer.py:49 `optim.AdamW(self.parameters(), lr=0.0005)`: decoupled weight decay, torch's
 * operation order -- p *= 1 - lr wd; m += (g - m)(1 - b1); v = b2 v + (1 - b2) g g;
 * p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)).  params / grads / exp_avg / exp_avg_sq: n floats each,
 * 16-byte aligned, device memory; exp_avg and exp_avg_sq start at zero.  `step` = t >= 1 of THIS step.  lr_dev /
 * step_dev (optional, device memory): learning rate and step count read by the kernel instead of the by-value
 * arguments, so that a captured HIP graph can replay the step (the caller advances *step_dev before each replay). */
int isd_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, double lr,
                   double beta1, double beta2, double eps, double weight_decay, int64_t step, const float* lr_dev,
                   const int64_t* step_dev, void* stream);
/* The same update over a list of tensors in one launch (per 96 tensors): host arrays of n_tensors DEVICE pointers and
 * element counts; tensors need not be aligned or adjacent.  step_dev (optional): FOUR int64 in device memory, zero
 * before the first step -- [0] = steps taken so far: the kernel uses t = [0] + 1 and stores it back itself; [1] is its
 * scratch; [2], [3] hold beta1^t, beta2^t as doubles (kept by recurrence; to resume at step t > 0 store the bit
 * patterns of beta^t there) -- so a captured graph replays the step with nothing to advance on the host. */
int isd_adamw_multi_step(int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                         float* const* exp_avg_sq, const int64_t* numel, double lr, double beta1, double beta2,
                         double eps, double weight_decay, int64_t step, const float* lr_dev, int64_t* step_dev,
                         void* stream);

/* ----------------------------------------------------------------------
 * HeadConv_Paper_Version  (replaces src/fast/models/fast.py:170-196 behind the head contract :203-210)
 *   x [B][C][T] f32 -> out [B][feature_dim];  feature_dim in [3,64] (F1 = F/2, F2 = F3 = F/3, F4 = F), T >= 46.
 * Flat parameter block (reference state_dict order, trainable tensors only):
 *   cnn1_t.weight [F1,1,1,3] | cnn1_t.bias [F1] | cnn1_s.weight [F1,F1,C,1] | norm1.weight [F1] | norm1.bias [F1] |
 *   cnn2.weight [F2,F1,1,3] | norm2.weight | norm2.bias | cnn3.weight [F3,F2,1,3] | norm3.* | cnn4.weight [F4,F3,1,3] |
 *   norm4.weight | norm4.bias
 * Buffer block: running_mean / running_var of norm1..norm4 (F1,F1,F2,F2,F3,F3,F4,F4 floats), updated when training.
 * training != 0: batch statistics (biased variance to normalise, unbiased for the running update).  MaxPool ties
 * route the gradient to the first element, as torch does.  backward: parameter gradients only (x is data); one
 * backward per train-mode forward, same workspace.
 * ---------------------------------------------------------------------- */
typedef struct isd_paperhead_plan isd_paperhead_plan;
int isd_paperhead_plan_create(isd_paperhead_plan** out, int in_channels, int feature_dim, int T);
int isd_paperhead_plan_destroy(isd_paperhead_plan* plan);
int64_t isd_paperhead_param_count(const isd_paperhead_plan* plan);
int64_t isd_paperhead_buffer_count(const isd_paperhead_plan* plan);
int64_t isd_paperhead_workspace_bytes(const isd_paperhead_plan* plan, int64_t B);
int isd_paperhead_forward(const isd_paperhead_plan* plan, const float* x, const float* params, float* buffers,
                          float* out, void* workspace, int64_t B, int training, float momentum, float eps, void* stream);
int isd_paperhead_backward(const isd_paperhead_plan* plan, const float* x, const float* params, const float* dout,
                           float* dparams, void* workspace, int64_t B, void* stream);
/* ... plus the gradient w.r.t. the input trials dx [B][C][T]; training = 1 after a batch-statistics forward, 0 after an
 * eval-mode forward (running statistics). */
int isd_paperhead_backward_x(const isd_paperhead_plan* plan, const float* x, const float* params, const float* dout,
                             float* dparams, float* dx, void* workspace, int64_t B, int training, void* stream);
/* Synchronised BatchNorm under data parallelism, as for the EEGNet head: forward stages 0..4 and backward stages 0..4;
 * after stage s < 4 one layer's block of fp64 batch sums (isd_paperhead_sync_block) is complete in the workspace and
 * the caller all-reduces it (SUM) before the next stage; `world` scales the counts and pre-divides the gamma / beta
 * gradients, which every rank assembles from global sums. */
int isd_paperhead_forward_stage(const isd_paperhead_plan* plan, int stage, const float* x, const float* params,
                                float* buffers, float* out, void* workspace, int64_t B, int training, float momentum,
                                float eps, int world, void* stream);
int isd_paperhead_backward_stage(const isd_paperhead_plan* plan, int stage, const float* x, const float* params,
                                 const float* dout, float* dparams, void* workspace, int64_t B, int world, void* stream);
int isd_paperhead_sync_block(const isd_paperhead_plan* plan, int64_t B, int backward, int stage, int64_t* byte_offset,
                             int64_t* n_doubles);
int isd_paperhead_sync_block_kind(int backward, int stage);   /* as isd_eegnet_sync_block_kind */

/* ------------------------------------------------------------------------
 * Zone-batched launches for the one-encoder-per-zone heads (Head.encoders, fast.py:203-210: eight EEGNet_Encoder /
 * CVBlock / HeadConv_Paper_Version instances on eight channel subsets).  Between isd_zone_batch_begin() and
 * isd_zone_batch_launch() the calls
 *   isd_eegnet_forward / isd_eegnet_backward / isd_paperhead_forward / isd_paperhead_backward
 * made by THIS thread record their kernel launches instead of issuing them; isd_zone_batch_next() separates the
 * zones (at most 8).  isd_zone_batch_launch() then issues launch i of all zones as ONE kernel (blockIdx.z = zone).
 * The zones must make the same calls on plans of the same kind, batch and length (channel counts may differ); a
 * call that cannot be recorded fails and poisons the batch (isd_zone_batch_launch then returns ISD_ERR_INVALID).
 * Nothing is read on the host in between: the recorded calls only have to use the stream given to the launch. */
int isd_zone_batch_begin(void);
int isd_zone_batch_next(void);
int isd_zone_batch_launch(void* stream);
int isd_zone_batch_abort(void);

#ifdef __cplusplus
}
#endif
#endif /* ISD_HIP_H */
