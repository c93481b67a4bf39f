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

enum { ISD_FB_F32 = 0, ISD_FB_F64 = 1, ISD_FB_AUTO = 2 };

/* a12  : host, [n_bands][n_sections][2] doubles (a1, a2)
 * gain : host, [n_bands] doubles
 * precision: arithmetic of the in-chunk recursion (the cross-chunk state scan
 *            is always fp64); AUTO picks F64 when a pole is too close to z=1
 *            for fp32 to keep 1e-4.  n_sections <= 8. */
int isd_fb_plan_create(isd_fb_plan** out, int n_bands, int n_sections,
                       const double* a12, const double* gain, int precision);
int isd_fb_plan_destroy(isd_fb_plan* plan);
int isd_fb_plan_precision(const isd_fb_plan* plan); /* resolved ISD_FB_F32 / F64 */

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

#ifdef __cplusplus
}
#endif
#endif /* ISD_HIP_H */
