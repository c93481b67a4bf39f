// STFT (scipy-legacy defaults) and band aggregation.
//
// Replaces scipy.signal.stft(sig, fs, nperseg=64, noverlap=32) at
// scripts/global_shap_analysis.py:132 and the band means of :151-156.
//
// Generic path: radix-2 FFT in LDS, twiddles and window staged in LDS, several frames
// per 256-thread workgroup.  The epilogue either writes the one-sided complex spectrum
// in scipy's [row][bin][frame] layout or reduces it to band magnitude / power / log-power.
// (The headline 64/32 configuration is normally served by the fused kernel in fb.hip,
// which never materialises the filtered signals.)
#include "common.h"
#include "stft_plan.h"
#include <math.h>
#include <vector>

namespace isd {

struct BandArgs {
  int klo[kMaxBands];
  int khi[kMaxBands];
};

__device__ __forceinline__ unsigned bitrev(unsigned v, int bits) { return __brev(v) >> (32 - bits); }

// mode_out: 0 complex spectrum, 1 band reduction
template <int MODE_OUT>
__global__ __launch_bounds__(256) void stft_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                   const float* __restrict__ win, const float2* __restrict__ tw,
                                                   int64_t n_frames_total, int T, int n, int log2n, int hop, int J,
                                                   int tpf, float scale, int C, int nbi, int nb, BandArgs ba,
                                                   int bp_mode, float eps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int fpb = 256 / tpf;                        // frames per block
  float2* buf = reinterpret_cast<float2*>(smem_raw);                 // [fpb][n]
  float2* stw = buf + (size_t)fpb * n;                                // [n/2]
  float* swin = reinterpret_cast<float*>(stw + n / 2);                // [n]
  const int tid = threadIdx.x;
  for (int i = tid; i < n / 2; i += 256) stw[i] = tw[i];
  for (int i = tid; i < n; i += 256) swin[i] = win[i];
  __syncthreads();

  const int fl = tid / tpf;                         // local frame
  const int tl = tid - fl * tpf;                    // thread inside the frame
  const int64_t fid = (int64_t)blockIdx.x * fpb + fl;
  const bool live = fid < n_frames_total;
  const int64_t row = live ? fid / J : 0;
  const int j = live ? (int)(fid - row * J) : 0;
  float2* fb = buf + (size_t)fl * n;
  const int half = n / 2;

  // windowed load in bit-reversed order (zero extension by n/2 on both sides + zero padding)
  for (int idx = tl; idx < n; idx += tpf) {
    const int pos = j * hop + idx - half;
    float v = 0.f;
    if (live && pos >= 0 && pos < T) v = x[row * (int64_t)T + pos] * swin[idx];
    fb[bitrev((unsigned)idx, log2n)] = make_float2(v, 0.f);
  }
  __syncthreads();
  for (int s = 0; s < log2n; ++s) {
    const int hs = 1 << s;
    const int tstride = half >> s;                  // twiddle stride n / (2*hs)
    for (int t = tl; t < half; t += tpf) {
      const int k = t & (hs - 1);
      const int i0 = ((t >> s) << (s + 1)) + k;
      const int i1 = i0 + hs;
      const float2 w = stw[k * tstride];
      const float2 a = fb[i0], b = fb[i1];
      const float2 bw = make_float2(b.x * w.x - b.y * w.y, b.x * w.y + b.y * w.x);
      fb[i0] = make_float2(a.x + bw.x, a.y + bw.y);
      fb[i1] = make_float2(a.x - bw.x, a.y - bw.y);
    }
    __syncthreads();
  }
  if (!live) return;
  if (MODE_OUT == 0) {
    const int nfreq = half + 1;
    float2* Z = reinterpret_cast<float2*>(out);
    for (int k = tl; k < nfreq; k += tpf) {
      const float2 v = fb[k];
      Z[(row * nfreq + k) * (int64_t)J + j] = make_float2(v.x * scale, v.y * scale);
    }
  } else {
    // row = (b * nbi + bi) * C + c
    const int64_t bc = row / C;
    const int c = (int)(row - bc * C);
    const int bi = (int)(bc % nbi);
    const int64_t bt = bc / nbi;
    const int b_first = (nbi == 1) ? 0 : bi;
    const int b_count = (nbi == 1) ? nb : 1;
    for (int q = tl; q < b_count; q += tpf) {
      const int b = b_first + q;
      const int klo = ba.klo[b], khi = ba.khi[b];
      float acc = 0.f;
      for (int k = klo; k <= khi; ++k) {
        const float2 v = fb[k];
        const float p = (v.x * v.x + v.y * v.y) * (scale * scale);
        acc += (bp_mode == ISD_BP_MAGNITUDE) ? sqrtf(p) : p;
      }
      float r = khi >= klo ? acc / (float)(khi - klo + 1) : 0.f;
      if (bp_mode == ISD_BP_LOGPOWER) r = logf(r + eps);
      out[((bt * nb + b) * C + c) * (int64_t)J + j] = r;
    }
  }
}

// ---------------------------------------------------------------------------------------
// Band power for heavily overlapped frames (stress configuration: nperseg 1024, hop 64, 4096 samples): one wave
// per (trial, band, channel) row, lane m owns block m of `hop` samples.  With a Hann window
//   Z_k[j] = (-1)^k w^{kj} [ R_k[j]/2 + (w^j R_{k+1}[j] + w^{-j} R_{k-1}[j])/4 ] / sum(window),   w = e^{2 pi i hop/n},
//   R_k[j] = sum over the n/hop blocks m of frame j of S_k[m],   S_k[m] = sum_{t in block m} y[t] e^{-2 pi i k t/n},
// so a row costs one 64-sample DFT sum per lane for the band's bins and their two neighbours, two sliding-window
// sums over lanes (log2(n/hop) shuffles each) and a few complex multiplies per frame -- not a 1024-point FFT per
// frame (513 bins computed, 3 used).  Reads each row once, coalesced; the block layout is transposed through LDS,
// half a row at a time.
// ---------------------------------------------------------------------------------------
template <int H, int KB>
__global__ __launch_bounds__(64) void bandpower_blocksum_kernel(const float2* __restrict__ tab,
                                                                const float* __restrict__ y, float* __restrict__ feat,
                                                                int64_t R, int C, int T, int nb, int J, int n,
                                                                int log2_nblk, float scale2, BandArgs ba, int mode,
                                                                float eps) {
  // Two passes of 32 blocks: lane l owns HALF of block p*32 + (l >> 1) (H/2 samples), so the transposing LDS tile
  // holds half a row (9 KiB instead of 18 KiB at H = 64: 12 instead of 8 waves per CU on a pure streaming kernel).
  // Both halves use the first H/2 table columns; the odd lane rotates its sum by e^{-2 pi i k (H/2) / n}
  // (column H/2) and a DPP shift joins the pair.
  constexpr int HH = H / 2, RS = HH + 4;                      // padded half-block stride: conflict-free ds_read_b128
  __shared__ __attribute__((aligned(16))) float tile[64 * RS];
  __shared__ float2 Sblk[KB][64];
  __shared__ float2 tw[64];                                   // e^{-2 pi i u / nblk}
  const int lane = threadIdx.x;
  const int64_t row = blockIdx.x;
  const int band = (int)((row / C) % nb);
  const int klo = ba.klo[band], khi = ba.khi[band];
  const int nbin = khi - klo + 1, k0 = klo - 1;
  const int nblk = 1 << log2_nblk;
  const float* src = y + row * (int64_t)T;
  const bool vec = (T & 3) == 0 && ((reinterpret_cast<uintptr_t>(y) & 15) == 0);
  if (lane < nblk) {
    float sn, cs;
    sincospif(2.f * (float)lane / (float)nblk, &sn, &cs);
    tw[lane] = make_float2(cs, -sn);
  }
  const int kmax = n / 2;
  for (int p = 0; p < 2; ++p) {
    wave_lds_sync();                                            // the previous pass has read its tile
#pragma unroll
    for (int it = 0; it < HH / 4; ++it) {
      const int el = (it * 64 + lane) * 4;                      // position inside the half row
      const int e = p * 32 * H + el;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (vec && e + 3 < T) {
        v = *reinterpret_cast<const float4*>(src + e);
      } else {
        if (e + 0 < T) v.x = src[e];
        if (e + 1 < T) v.y = src[e + 1];
        if (e + 2 < T) v.z = src[e + 2];
        if (e + 3 < T) v.w = src[e + 3];
      }
      *reinterpret_cast<float4*>(tile + (el / HH) * RS + (el % HH)) = v;
    }
    wave_lds_sync();
    float2 P[KB];
#pragma unroll
    for (int kk = 0; kk < KB; ++kk) P[kk] = make_float2(0.f, 0.f);
    const float* sp = tile + lane * RS;
    for (int i0 = 0; i0 < HH; i0 += 4) {
      const float4 v = *reinterpret_cast<const float4*>(sp + i0);
#pragma unroll
      for (int kk = 0; kk < KB; ++kk) {
        const int k = k0 + kk < kmax ? k0 + kk : kmax;            // unused slots repeat a valid table row
        const float2* tb = tab + (int64_t)k * H + i0;             // wave-uniform -> scalar loads
        P[kk].x = fmaf(v.x, tb[0].x, fmaf(v.y, tb[1].x, fmaf(v.z, tb[2].x, fmaf(v.w, tb[3].x, P[kk].x))));
        P[kk].y = fmaf(v.x, tb[0].y, fmaf(v.y, tb[1].y, fmaf(v.z, tb[2].y, fmaf(v.w, tb[3].y, P[kk].y))));
      }
    }
#pragma unroll
    for (int kk = 0; kk < KB; ++kk) {
      const int k = k0 + kk < kmax ? k0 + kk : kmax;
      const float2 ph = tab[(int64_t)k * H + HH];               // the odd lane's offset inside the block
      const float2 q = (lane & 1) ? make_float2(P[kk].x * ph.x - P[kk].y * ph.y, P[kk].x * ph.y + P[kk].y * ph.x) : P[kk];
      const float sx = q.x + row_shl<1>(q.x), sy = q.y + row_shl<1>(q.y);
      if (!(lane & 1)) Sblk[kk][p * 32 + (lane >> 1)] = make_float2(sx, sy);
    }
  }
  wave_lds_sync();
  float2 S[KB];
#pragma unroll
  for (int kk = 0; kk < KB; ++kk) S[kk] = Sblk[kk][lane];
  blocksum_finish<KB>(S, tw, lane, k0, nbin, nblk, J, scale2, mode, eps, feat + row * (int64_t)J);
}

}  // namespace isd

using namespace isd;

extern "C" int isd_stft_plan_create(isd_stft_plan** out, int T, int nperseg, int noverlap) {
  ISD_CHECK_ARG(out, "isd_stft_plan_create: null argument");
  ISD_CHECK_ARG(T >= 1 && T <= (1 << 24), "isd_stft_plan_create: T=%d out of range", T);
  ISD_CHECK_ARG(nperseg >= 8 && nperseg <= 4096 && (nperseg & (nperseg - 1)) == 0,
                "isd_stft_plan_create: nperseg=%d must be a power of two in [8,4096]", nperseg);
  ISD_CHECK_ARG(noverlap >= 0 && noverlap < nperseg, "isd_stft_plan_create: noverlap=%d not in [0,%d)", noverlap,
                nperseg);
  isd_stft_plan* p = new isd_stft_plan();
  p->T = T; p->n = nperseg; p->hop = nperseg - noverlap;
  p->log2n = 0;
  while ((1 << p->log2n) < nperseg) ++p->log2n;
  int L = T + 2 * (nperseg / 2);
  const int rem = (L - nperseg) % p->hop;
  const int pad = ((p->hop - rem) % p->hop) % nperseg;   // scipy: (-(L-nperseg) % nstep) % nperseg
  L += pad;
  p->J = (L - nperseg) / p->hop + 1;
  std::vector<float> win(nperseg);
  std::vector<float2> tw(nperseg / 2);
  double wsum = 0.0;
  for (int i = 0; i < nperseg; ++i) {
    win[i] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * i / nperseg));   // periodic Hann, rounded to f32 like scipy
    wsum += (double)win[i];
  }
  p->scale = (float)(1.0 / wsum);
  for (int i = 0; i < nperseg / 2; ++i)
    tw[i] = make_float2((float)cos(2.0 * M_PI * i / nperseg), (float)(-sin(2.0 * M_PI * i / nperseg)));
  // direct-DFT table for the fused 64/32 kernel: w[n] * exp(-2 pi i k n / 64), laid out for a lane that holds its 32
  // samples as register pairs {j, j + 16}: tab[k][c][j] = (t_c[j], t_c[j + 16]), c = 0/1: real/imaginary part over
  // the first window half (n = 0..31), c = 2/3: over the second half (n = 32..63).  A packed FMA of a sample pair
  // with one table entry then needs no broadcast of either operand; the two lanes of the sum are added at the end.
  std::vector<float2> dft;
  if (nperseg == 64) {
    dft.resize(33 * 64);
    for (int k = 0; k <= 32; ++k)
      for (int c = 0; c < 4; ++c)
        for (int j = 0; j < 16; ++j) {
          float t[2];
          for (int h = 0; h < 2; ++h) {
            const int i = (c >> 1) * 32 + j + 16 * h;
            const double ph = 2.0 * M_PI * ((k * i) % 64) / 64.0;
            t[h] = (c & 1) ? (float)(-(double)win[i] * sin(ph)) : (float)((double)win[i] * cos(ph));
          }
          dft[k * 64 + c * 16 + j] = make_float2(t[0], t[1]);
        }
  }
  // symmetric-pair table of the serial-lane fused extractor (fb.hip): entry m - 1 of bin k = e^{-i m th_k}, m = 1..16
  // (128 bytes = two cache lines per bin)
  std::vector<float2> sym;
  if (nperseg == 64 && p->hop == 32) {
    sym.assign(33 * 16, make_float2(0.f, 0.f));
    for (int k = 0; k <= 32; ++k)
      for (int m = 1; m <= 16; ++m) {
        const double ph = 2.0 * M_PI * ((k * m) % 64) / 64.0;
        sym[k * 16 + m - 1] = make_float2((float)cos(ph), (float)(-sin(ph)));
      }
  }
  // block-sum table for heavily overlapped frames
  std::vector<float2> blk;
  {
    const int hop = p->hop, nblk = nperseg / hop;
    if ((hop == 32 || hop == 64) && nperseg % hop == 0 && nblk >= 4 && nblk <= 64 && (nblk & (nblk - 1)) == 0 &&
        T <= 64 * hop) {
      blk.resize((size_t)(nperseg / 2 + 1) * hop);
      for (int k = 0; k <= nperseg / 2; ++k)
        for (int i = 0; i < hop; ++i) {
          const double ph = 2.0 * M_PI * (double)(((int64_t)k * i) % nperseg) / nperseg;
          blk[(size_t)k * hop + i] = make_float2((float)cos(ph), (float)(-sin(ph)));
        }
    }
  }
  p->d_win = nullptr; p->d_tw = nullptr; p->d_dft = nullptr; p->d_blk = nullptr; p->d_sym = nullptr;
  hipError_t e = hipMalloc(&p->d_win, sizeof(float) * nperseg);
  if (e == hipSuccess) e = hipMalloc(&p->d_tw, sizeof(float2) * (nperseg / 2));
  if (e == hipSuccess) e = hipMemcpy(p->d_win, win.data(), sizeof(float) * nperseg, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(p->d_tw, tw.data(), sizeof(float2) * (nperseg / 2), hipMemcpyHostToDevice);
  if (e == hipSuccess && !dft.empty()) {
    e = hipMalloc(&p->d_dft, sizeof(float2) * dft.size());
    if (e == hipSuccess) e = hipMemcpy(p->d_dft, dft.data(), sizeof(float2) * dft.size(), hipMemcpyHostToDevice);
  }
  if (e == hipSuccess && !sym.empty()) {
    e = hipMalloc(&p->d_sym, sizeof(float2) * sym.size());
    if (e == hipSuccess) e = hipMemcpy(p->d_sym, sym.data(), sizeof(float2) * sym.size(), hipMemcpyHostToDevice);
  }
  if (e == hipSuccess && !blk.empty()) {
    e = hipMalloc(&p->d_blk, sizeof(float2) * blk.size());
    if (e == hipSuccess) e = hipMemcpy(p->d_blk, blk.data(), sizeof(float2) * blk.size(), hipMemcpyHostToDevice);
  }
  if (e != hipSuccess) {
    set_error("isd_stft_plan_create: %s", hipGetErrorString(e));
    isd_stft_plan_destroy(p);
    return e == hipErrorNoDevice ? ISD_ERR_NO_DEVICE : ISD_ERR_HIP;
  }
  *out = p;
  return ISD_OK;
}

extern "C" int isd_stft_plan_destroy(isd_stft_plan* p) {
  if (!p) return ISD_OK;
  if (p->d_win) (void)hipFree(p->d_win);
  if (p->d_tw) (void)hipFree(p->d_tw);
  if (p->d_dft) (void)hipFree(p->d_dft);
  if (p->d_blk) (void)hipFree(p->d_blk);
  if (p->d_sym) (void)hipFree(p->d_sym);
  delete p;
  return ISD_OK;
}

extern "C" int isd_stft_plan_frames(const isd_stft_plan* p) { return p ? p->J : ISD_ERR_INVALID; }
extern "C" int isd_stft_plan_bins(const isd_stft_plan* p) { return p ? p->n / 2 + 1 : ISD_ERR_INVALID; }

static int stft_launch(const isd_stft_plan* p, int mode_out, const float* x, float* out, int64_t R, int C, int nbi,
                       int nb, const BandArgs& ba, int bp_mode, float eps, hipStream_t st) {
  const int tpf = p->n / 2 < 256 ? p->n / 2 : 256;
  const int fpb = 256 / tpf;
  const int64_t frames = R * p->J;
  const int64_t blocks = cdiv(frames, fpb);
  ISD_CHECK_ARG(blocks <= 0x7fffffffLL, "stft: too many frames (%lld)", (long long)frames);
  const size_t lds = sizeof(float2) * ((size_t)fpb * p->n + p->n / 2) + sizeof(float) * p->n;
  if (mode_out == 0)
    hipLaunchKernelGGL((stft_kernel<0>), dim3((unsigned)blocks), dim3(256), lds, st, x, out, p->d_win, p->d_tw, frames,
                       p->T, p->n, p->log2n, p->hop, p->J, tpf, p->scale, C, nbi, nb, ba, bp_mode, eps);
  else
    hipLaunchKernelGGL((stft_kernel<1>), dim3((unsigned)blocks), dim3(256), lds, st, x, out, p->d_win, p->d_tw, frames,
                       p->T, p->n, p->log2n, p->hop, p->J, tpf, p->scale, C, nbi, nb, ba, bp_mode, eps);
  ISD_LAUNCH_CHECK();
  return ISD_OK;
}

extern "C" int isd_stft_forward(const isd_stft_plan* p, const float* x, float* Z, int64_t R, void* stream) {
  ISD_CHECK_ARG(p && (R == 0 || (x && Z)), "isd_stft_forward: null argument");
  ISD_CHECK_ARG(R >= 0, "isd_stft_forward: R=%lld", (long long)R);
  if (R == 0) return ISD_OK;
  BandArgs ba = {};
  return stft_launch(p, 0, x, Z, R, 1, 1, 1, ba, 0, 0.f, (hipStream_t)stream);
}

int isd::fill_band_args(const isd_stft_plan* p, int n_bands, const int* klo, const int* khi, int* oklo, int* okhi,
                        const char* who) {
  ISD_CHECK_ARG(n_bands >= 1 && n_bands <= kMaxBands, "%s: n_bands=%d not in [1,%d]", who, n_bands, kMaxBands);
  ISD_CHECK_ARG(klo && khi, "%s: null band table", who);
  for (int b = 0; b < n_bands; ++b) {
    ISD_CHECK_ARG(khi[b] < klo[b] || (klo[b] >= 0 && khi[b] <= p->n / 2), "%s: band %d bins [%d,%d] outside [0,%d]",
                  who, b, klo[b], khi[b], p->n / 2);
    oklo[b] = klo[b];
    okhi[b] = khi[b];
  }
  return ISD_OK;
}

extern "C" int isd_stft_bandpower(const isd_stft_plan* p, const float* y, float* feat, int64_t B, int64_t C,
                                  int n_bands_in, int n_bands, const int* klo, const int* khi, int mode, float eps,
                                  void* stream) {
  ISD_CHECK_ARG(p && (B == 0 || (y && feat)), "isd_stft_bandpower: null argument");
  ISD_CHECK_ARG(B >= 0 && C >= 1 && C <= (1 << 20), "isd_stft_bandpower: bad shape B=%lld C=%lld", (long long)B,
                (long long)C);
  ISD_CHECK_ARG(n_bands_in == 1 || n_bands_in == n_bands, "isd_stft_bandpower: n_bands_in must be 1 or n_bands");
  ISD_CHECK_ARG(mode >= ISD_BP_MAGNITUDE && mode <= ISD_BP_LOGPOWER, "isd_stft_bandpower: bad mode %d", mode);
  BandArgs ba = {};
  int rc = fill_band_args(p, n_bands, klo, khi, ba.klo, ba.khi, "isd_stft_bandpower");
  if (rc) return rc;
  if (B == 0) return ISD_OK;
  if (p->n == 64 && p->hop == 32 && p->T <= 512 && p->d_dft && n_bands_in == n_bands)
    return bandpower_direct(p, y, feat, B * n_bands * C, (int)C, n_bands, ba.klo, ba.khi, mode, eps,
                            (hipStream_t)stream);
  if (p->d_blk && n_bands_in == n_bands) {
    // per-band rows, every band 1..6 interior bins: block sums instead of one FFT per frame
    int nbmax = 0;
    bool ok = true;
    for (int b = 0; b < n_bands; ++b) {
      const int nbin = ba.khi[b] - ba.klo[b] + 1;
      if (nbin < 1 || nbin > 6 || ba.klo[b] < 1 || ba.khi[b] > p->n / 2 - 1) ok = false;
      if (nbin > nbmax) nbmax = nbin;
    }
    const int64_t rows = B * n_bands * C;
    if (ok && rows <= 0x7fffffffLL) {
      int log2_nblk = 0;
      while ((p->hop << log2_nblk) < p->n) ++log2_nblk;
      const int KB = nbmax + 2 <= 4 ? 4 : nbmax + 2 <= 5 ? 5 : nbmax + 2 <= 6 ? 6 : 8;
      hipStream_t st = (hipStream_t)stream;
#define ISD_BS_LAUNCH(H, K)                                                                                        \
  hipLaunchKernelGGL((bandpower_blocksum_kernel<H, K>), dim3((unsigned)rows), dim3(64), 0, st, p->d_blk, y, feat, rows, \
                     (int)C, p->T, n_bands, p->J, p->n, log2_nblk, p->scale * p->scale, ba, mode, eps)
      if (p->hop == 64) {
        if (KB == 4) ISD_BS_LAUNCH(64, 4); else if (KB == 5) ISD_BS_LAUNCH(64, 5);
        else if (KB == 6) ISD_BS_LAUNCH(64, 6); else ISD_BS_LAUNCH(64, 8);
      } else {
        if (KB == 4) ISD_BS_LAUNCH(32, 4); else if (KB == 5) ISD_BS_LAUNCH(32, 5);
        else if (KB == 6) ISD_BS_LAUNCH(32, 6); else ISD_BS_LAUNCH(32, 8);
      }
#undef ISD_BS_LAUNCH
      ISD_LAUNCH_CHECK();
      return ISD_OK;
    }
  }
  return stft_launch(p, 1, y, feat, B * n_bands_in * C, (int)C, n_bands_in, n_bands, ba, mode, eps,
                     (hipStream_t)stream);
}
