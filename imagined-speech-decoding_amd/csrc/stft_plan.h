// STFT plan shared by stft.hip (generic FFT path) and fb.hip (fused 64/32 direct-DFT path).
#pragma once
#include "common.h"

struct isd_stft_plan {
  int T, n, hop, log2n, J;
  float scale;          // 1 / sum(window)  (scipy scaling='spectrum')
  float* d_win;         // [n] periodic Hann
  float2* d_tw;         // [n/2] exp(-2 pi i k / n)
  float2* d_dft;        // n == 64 only: [33][64] w[n] * exp(-2 pi i k n / 64)
  float2* d_blk;        // heavily overlapped frames (n = 2^a * hop, hop 32 or 64, T <= 64 * hop): [n/2+1][hop]
                        // exp(-2 pi i k i / n), the per-block DFT table of the block-sum band-power kernel
};

namespace isd {
constexpr int kMaxBands = 64;
// direct-DFT band aggregation (fb.hip) for nperseg 64 / hop 32 / T <= 512, per-band input
int bandpower_direct(const isd_stft_plan* st, const float* y, float* feat, int64_t R, int C, int nb, const int* klo,
                     const int* khi, int mode, float eps, hipStream_t stream);
int fill_band_args(const isd_stft_plan* p, int n_bands, const int* klo, const int* khi, int* oklo, int* okhi,
                   const char* who);
}  // namespace isd
