// STFT plan shared by stft.hip (generic FFT path) and fb.hip (fused 64/32 direct-DFT path).
#pragma once
#include "common.h"

struct isd_stft_plan {
  int T, n, hop, log2n, J;
  float scale;          // 1 / sum(window)  (scipy scaling='spectrum')
  float* d_win;         // [n] periodic Hann
  float2* d_tw;         // [n/2] exp(-2 pi i k / n)
  float2* d_dft;        // n == 64 only: [33][4][16] w[n] * exp(-2 pi i k n / 64) as sample pairs (stft.hip plan create)
  float2* d_sym;        // n == 64, hop == 32: [33][16] {cos(m th_k), -sin(m th_k)}, th_k = 2 pi k / 64, m = 1..16 -- the
                        // UNWINDOWED half-frame DFT over sample pairs symmetric about the middle of a 32-sample chunk
                        // (fb.hip fused_serial_kernel; the Hann window is applied in the frequency domain)
  float2* d_blk;        // heavily overlapped frames (n = 2^a * hop, hop 32 or 64, T <= 64 * hop): [n/2+1][hop]
                        // exp(-2 pi i k i / n), the per-block DFT table of the block-sum band-power kernel
};

namespace isd {
constexpr int kMaxBands = 64;

#if defined(__HIPCC__)
// Last stage of the block-sum band power (stft.hip bandpower_blocksum_kernel, fb.hip fused_long_kernel).
// In: lane m holds S[kk] = sum_{t in block m} y[t] e^{-2 pi i (k0+kk) (t - 64 m... block-local phase)} -- the per-block DFT
// sums with BLOCK-LOCAL phase for bins k0 .. k0+KB-1 (k0 = klo - 1); tw[u] = e^{-2 pi i u / nblk} in LDS.
// Applies the absolute block phase, forms the two sliding-window sums over lanes (zero fill outside [0, 64)),
// combines the Hann neighbours per frame and writes out_row[j], j < J (frame j = lane; frame 64 in a second,
// wave-uniform pass so that every lane takes part in the shuffles).
template <int KB>
__device__ __forceinline__ void blocksum_finish(const float2 (&S)[KB], const float2* tw, int lane, int k0, int nbin,
                                                int nblk, int J, float scale2, int mode, float eps,
                                                float* __restrict__ out_row) {
  const int half = nblk >> 1;
  float2 Cw[KB], Bw[KB];
#pragma unroll
  for (int kk = 0; kk < KB; ++kk) {
    const float2 w = tw[((k0 + kk) * lane) & (nblk - 1)];
    const float2 s = make_float2(S[kk].x * w.x - S[kk].y * w.y, S[kk].x * w.y + S[kk].y * w.x);
    Cw[kk] = s;
    Bw[kk] = s;
  }
  for (int d = 1; d < nblk; d <<= 1) {
#pragma unroll
    for (int kk = 0; kk < KB; ++kk) {
      const float cx = __shfl_down(Cw[kk].x, d, 64), cy = __shfl_down(Cw[kk].y, d, 64);
      const float bx = __shfl_up(Bw[kk].x, d, 64), by = __shfl_up(Bw[kk].y, d, 64);
      if (lane + d < 64) { Cw[kk].x += cx; Cw[kk].y += cy; }
      if (lane >= d) { Bw[kk].x += bx; Bw[kk].y += by; }
    }
  }
  const int n_pass = J > 64 ? 2 : 1;
  for (int f = 0; f < n_pass; ++f) {
    const int j = lane + 64 * f;
    const int from_c = j - half;                                  // leading window starting at block j - half
    float2 Rk[KB];
#pragma unroll
    for (int kk = 0; kk < KB; ++kk) {
      const float cx = __shfl(Cw[kk].x, from_c & 63, 64), cy = __shfl(Cw[kk].y, from_c & 63, 64);
      const float bx = __shfl(Bw[kk].x, (j + half - 1) & 63, 64), by = __shfl(Bw[kk].y, (j + half - 1) & 63, 64);
      Rk[kk] = j >= half ? make_float2(cx, cy) : make_float2(bx, by);
    }
    const float2 wn = tw[j & (nblk - 1)];                         // e^{-2 pi i j/nblk};  w^j = conj(wn)
    float acc = 0.f;
#pragma unroll
    for (int bq = 0; bq < KB - 2; ++bq) {
      if (bq < nbin) {
        const float2 lo = Rk[bq], mid = Rk[bq + 1], hi = Rk[bq + 2];
        // w^j * hi + w^{-j} * lo
        const float sx = hi.x * wn.x + hi.y * wn.y + lo.x * wn.x - lo.y * wn.y;
        const float sy = hi.y * wn.x - hi.x * wn.y + lo.y * wn.x + lo.x * wn.y;
        const float vx = 0.5f * mid.x + 0.25f * sx, vy = 0.5f * mid.y + 0.25f * sy;
        const float pw = (vx * vx + vy * vy) * scale2;
        acc += mode == ISD_BP_MAGNITUDE ? sqrtf(pw) : pw;
      }
    }
    float r = nbin > 0 ? acc / (float)nbin : 0.f;
    if (mode == ISD_BP_LOGPOWER) r = logf(r + eps);
    if (j < J) out_row[j] = r;
  }
}
#endif
// direct-DFT band aggregation (fb.hip) for nperseg 64 / hop 32 / T <= 512, per-band input
int bandpower_direct(const isd_stft_plan* st, const float* y, float* feat, int64_t R, int C, int nb, const int* klo,
                     const int* khi, int mode, float eps, hipStream_t stream);
int fill_band_args(const isd_stft_plan* p, int n_bands, const int* klo, const int* khi, int* oklo, int* okhi,
                   const char* who);
}  // namespace isd
