// EEGNet_Encoder (reference: src/fast/models/fast.py:122-167) forward + backward on gfx950.
//
//   temporal Conv2d(1->8,(1,K),pad K/2) -> BN1 -> depthwise spatial Conv2d(8->16,(C,1),groups 8) -> BN2 -> ELU
//   -> AvgPool(1,4) -> depthwise Conv2d(16,(1,16),pad 8) -> pointwise 1x1 -> BN3 -> ELU -> AvgPool(1,8)
//   -> AdaptiveAvgPool -> Linear(16 -> F).
//
// The reference materialises the temporal-conv output a1[B,8,C,T+1] (8x the input; 34 GB at the stress
// configuration).  Here it is never formed.  The temporal and spatial convolutions commute and BN1 is an
// affine map per filter, so
//     z[b,g,t]  = sum_c Ws[g,c] x[b,c,t]                       (16 rows instead of 8*C)
//     u[b,g,t'] = sum_k Wt[f(g),k] zpad[b,g,t'+k]
//     a2        = s1[f] u + o1[f] sum_c Ws[g,c]                (the BN1 + spatial-conv output)
// and the BN1 batch statistics follow from the input alone:
//     mean(a1_f)   = sum_k Wt[f,k] m[k],      m[k]    = mean xpad[.+k]
//     mean(a1_f^2) = Wt[f]^T G Wt[f],         G[k,k'] = mean xpad[.+k] xpad[.+k']  (autocorrelation of x
//                                                       with exact edge corrections for the zero padding).
// The parameter gradients of stage 1 are assembled from the same quantities (dWt needs G and the
// cross-correlation of da2 with z; dWs is one GEMM of x with the back-filtered da2).  Statistics and
// cross-batch reductions accumulate in fp64.  Dropout: inference / p = 0 only (train-mode dropout would
// break parity with any reference RNG stream).
#include "common.h"
#include "zonebatch.h"
#include "exact.h"
#include <math.h>
#include <stddef.h>
#include <string.h>
#include <vector>

namespace isd {

// every launch of this file goes through zone_launch: issued at once, or recorded for a zone-batched launch (zonebatch.h)
#define ISD_ZLAUNCH(...)                                    \
  do {                                                      \
    if (!zone_launch(__VA_ARGS__)) return ISD_ERR_INVALID;  \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kF1 = 8, kF2 = 16, kK2 = 16, kP2 = 8, kMaxK = 64;

// offsets (floats) inside the flat parameter block, reference state_dict order
struct EegOff {
  int Wt, g1, b1, Ws, g2, b2, Wd, Wp, g3, b3, Wl, bl, total;
};
// offsets inside the buffer block: rm1 rv1 rm2 rv2 rm3 rv3
constexpr int kRm1 = 0, kRv1 = 8, kRm2 = 16, kRv2 = 32, kRm3 = 48, kRv3 = 64, kBufTotal = 80;

// Statistics block in the workspace.  Sums that many workgroups add to are ExactAcc (exact.h: integer atomics, the same
// bits whatever the arrival order); what single workgroups derive from them is fp64.
struct EegStats {
  double A[kMaxK];              // sum_rows sum_s x[s] x[s+d]
  double H[32][kMaxK];          // head prefix:  H[a][d] = sum_rows sum_{s<a} x[s] x[s+d]
  double Tl[33][kMaxK];         // tail suffix:  Tl[e][d] = sum_rows sum_{s>T-e} x[s] x[s+d]
  double S;                     // sum of all samples
  double Hs[32];                // Hs[a] = sum_rows sum_{s<a} x[s]
  double Ts[33];                // Ts[e] = sum_rows sum_{s>T-e} x[s]
  // raw accumulators of eeg_stats_kernel (eeg_stats_derive_kernel turns them into A, H, Tl, Hs, Ts above)
  ExactAcc Sx;                  // sum of all samples (eeg_stats_kernel; S above is its fp64 reading)
  ExactAcc D[5][256];           // D[q][i][j] = sum_rows sum_k x[16k+i] x[16k+j+16q]   (matrix cores)
  ExactAcc Gs[80][80];          // short rows (T <= 79): upper triangle of the full Gram matrix sum_rows y[a] y[j], with
                                //   y = (x[0..T), 1, 0, ...) (eeg_stats_gram_kernel); everything below is derived from it
  ExactAcc Pe[2][32][96];       // Gram blocks of the head / tail windows (eeg_stats_edge_kernel):
                                //   sum_rows x[s] x[s+d] = Pe[0][s][s+d],  sum_rows x[T-i] x[T-i+d] = Pe[1][31-i][31-i+d],
                                //   sum_rows x[s] = Pe[0][31][s],          sum_rows x[T-i]        = Pe[1][31][31-i]
  ExactAcc u1[kF2], u2[kF2];      // sum u, sum u^2 per g
  ExactAcc a1[kF2], a2[kF2];      // sum a4, sum a4^2 per h
  ExactAcc dy3s[kF2], dy3x[kF2];  // BN3 backward sums
  ExactAcc dy2s[kF2], dy2x[kF2];  // BN2 backward sums
  ExactAcc Sd[kF2], Su[kF2];      // sum da2, sum da2*u
  ExactAcc T1[kF2][kMaxK];        // sum da2[t'] zpad[t'+k]
  double dWp[kF2][kF2], dWd[kF2][kK2];
};

// every block isd_eegnet_sync_block hands out holds ONE kind of word (isd_eegnet_sync_block_kind): fp64 sums in front of
// Sx, exact-accumulator words from u1 up to dWp
static_assert(offsetof(EegStats, Ts) + sizeof(double) * 33 == offsetof(EegStats, Sx), "fp64 sums end where Sx begins");
static_assert(offsetof(EegStats, Sx) < offsetof(EegStats, D) && offsetof(EegStats, D) < offsetof(EegStats, u1) &&
              offsetof(EegStats, T1) < offsetof(EegStats, dWp), "exact accumulators sit between Sx and dWp");

// derived per-step coefficients (floats/doubles) in the workspace
struct EegCoef {
  double m[kMaxK];              // mean xpad[.+k]
  double G[kMaxK][kMaxK];       // mean xpad[.+k] xpad[.+k']
  double mu1[kF1], sig1[kF1];
  float s1[kF1], o1[kF1];
  float wsum[kF2];
  double muu[kF2], varu[kF2];   // stats of u
  float mu2[kF2], sig2[kF2], A2[kF2], B2[kF2];   // y2 = A2 u + B2
  float mu3[kF2], sig3[kF2], A3[kF2], B3[kF2];   // y3 = A3 a4 + B3
  float cA3[kF2], cB3[kF2], cC3[kF2];            // da4 = cA3 (dy3 - cB3 - xhat3 cC3)
  float cA2[kF2], cB2[kF2], cC2[kF2];
  // added to the dropout seed of every kernel of this pass (forward and backward): 0, or a multiple of the plan's
  // device-resident step counter, so that a captured graph -- which replays the same seed argument -- draws new masks
  unsigned long long seed_add;
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// counter-based dropout mask: keep-scale 1/(1-p) or 0 for element `idx` of stream `seed` (own RNG stream;
// statistically equivalent to nn.Dropout, not bit-identical to any torch generator)
__device__ __forceinline__ float drop_scale(uint64_t seed, uint64_t idx, float p) {
  if (p <= 0.f) return 1.f;
  uint64_t v = (idx + 0x9E3779B97F4A7C15ull) ^ seed;
  v ^= v >> 30; v *= 0xBF58476D1CE4E5B9ull;
  v ^= v >> 27; v *= 0x94D049BB133111EBull;
  v ^= v >> 31;
  const float uu = (float)(v >> 40) * (1.f / 16777216.f);
  return uu >= p ? 1.f / (1.f - p) : 0.f;
}
__device__ __forceinline__ float elu_f(float y) { return y > 0.f ? y : expm1f(y); }
__device__ __forceinline__ float elu_grad_f(float y) { return y > 0.f ? 1.f : expf(y); }

// block-wide sum of one float per thread, result valid on thread 0
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float s = 0.f;
  if (threadIdx.x == 0)
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];
  return s;
}

// ------------------------------------------------------------------------------------------------
// x statistics: autocorrelation over 64 lags + the head/tail pieces the zero padding cuts off.
// One wave per workgroup, persistent over rows.
//  * bulk on the matrix cores: with A[i][k] = x[16k+i] and B_q[k][j] = x[16k+j+16q] one MFMA chain per q = 0..4 gives
//    D_q[i][j] = sum_k x[16k+i] x[16k+j+16q], and the lag-d autocorrelation is sum_i D_{(i+d)>>4}[i][(i+d)&15].
//    Both operands are plain contiguous reads of the row (lane l <-> sample 64 step + l (+16q)): 5 MFMAs per 64
//    samples instead of 64 FMA + 64 LDS reads.
//  * edges: per row only the products x[s] x[s+d] for the first / last 31 samples are accumulated (lane = lag d);
//    their prefix sums over s -- what the zero padding of the temporal convolution removes -- are formed once, in
//    eeg_stats_derive_kernel, not per row.
// fp32 accumulation stays within one workgroup's share of the rows (eight waves, a few hundred rows each), then
// fp64 atomics.
// ------------------------------------------------------------------------------------------------
constexpr int kStatWaves = 4;                          // waves per workgroup
// Two kernels, so that neither carries the other's accumulators: the bulk one (20 MFMA accumulators + 17 operands)
// runs at 8 waves per SIMD, the edge one (62 lag products per lane) at 4 -- every row costs one memory latency, and
// the short rows of the feature classifier ([B, 5120, 65]) are nothing but latency.
//  bulk: D_q and the sum of all samples;  edge: Qh, Qt, Sh, St.
// The waves of a workgroup meet in LDS (fp32) before ONE set of fp64 global atomics per workgroup: 2048 waves
// sending 85 atomics per lane to the same ~5 k addresses took 0.15 ms on their own.
__device__ __forceinline__ void eeg_stats_kernel_body(const float* __restrict__ x,
                                                                    EegStats* __restrict__ st, int64_t rows, int T,
    unsigned zgx, unsigned zgy, unsigned zbz, unsigned zgz) {   // this zone's own gridDim.x/.y, blockIdx.z, gridDim.z
  __shared__ float tot[21][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int e = threadIdx.x; e < 21 * 64; e += 64 * kStatWaves) (&tot[0][0])[e] = 0.f;
  f32x4 Dq[5];
#pragma unroll
  for (int q = 0; q < 5; ++q) Dq[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float rs = 0.f;
  // 256 samples per step: the 17 distinct operands x[s0 + lane + 16 j] are requested together (one load per MFMA
  // operand, issued right before its use, left the loop waiting a full memory latency per 64 samples), then the
  // 20 MFMAs run.  Two rows are in flight at once: short rows (the feature classifier's T = 65) are one step each.
  const int64_t stride = (int64_t)zgx * kStatWaves;
  for (int64_t r = (int64_t)blockIdx.x * kStatWaves + wave; r < rows; r += 2 * stride) {
    const float* srcA = x + r * (int64_t)T;
    const bool two = r + stride < rows;                           // wave-uniform
    const float* srcB = x + (two ? r + stride : r) * (int64_t)T;
    for (int s0 = 0; s0 < T; s0 += 256) {
      float vA[17], vB[17];
#pragma unroll
      for (int j = 0; j < 17; ++j) {                                // clamped addresses: no test around a load ...
        const int idx = s0 + lane + 16 * j;
        const int ic = idx < T ? idx : T - 1;
        vA[j] = srcA[ic];
        vB[j] = srcB[ic];
      }
#pragma unroll
      for (int j = 0; j < 17; ++j) {                                // ... the 34 loads are in flight together, then selects
        const bool in = s0 + lane + 16 * j < T;
        vA[j] = in ? vA[j] : 0.f;
        vB[j] = (two && in) ? vB[j] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (s0 + 64 * u >= T) break;                              // wave-uniform
        const float a = vA[4 * u], b = vB[4 * u];
        rs += a + b;
        Dq[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, a, Dq[0], 0, 0, 0);
#pragma unroll
        for (int q = 1; q < 5; ++q) Dq[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, vA[4 * u + q], Dq[q], 0, 0, 0);
        Dq[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, b, Dq[0], 0, 0, 0);
#pragma unroll
        for (int q = 1; q < 5; ++q) Dq[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, vB[4 * u + q], Dq[q], 0, 0, 0);
      }
    }
  }
  const float rtot = wave_sum(rs);
  for (int w = 0; w < kStatWaves; ++w) {                        // the waves add in wave order: the same fp32 sum every run
    __syncthreads();                                            // (first pass: tot[] is zero)
    if (wave == w) {
#pragma unroll
      for (int q = 0; q < 5; ++q)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) tot[q * 4 + rr][lane] += Dq[q][rr];
      if (lane == 0) tot[20][0] += rtot;
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 20 * 64; e += 64 * kStatWaves) {
    const int slot = e >> 6, l = e & 63;
    exact_add(&st->D[slot >> 2][(4 * (l >> 4) + (slot & 3)) * 16 + (l & 15)], tot[slot][l]);
  }
  if (threadIdx.x == 0) exact_add(&st->Sx, tot[20][0]);
}
ISD_ZONE_FN(eeg_stats_kernel, 64 * kStatWaves)
__global__ __launch_bounds__(64 * kStatWaves) void eeg_stats_kernel(const float* __restrict__ x,
                                                                    EegStats* __restrict__ st, int64_t rows, int T) {
  eeg_stats_kernel_body(x, st, rows, T,
      gridDim.x, gridDim.y, blockIdx.z, gridDim.z);
}
ISD_ZONE_REGISTER(eeg_stats_kernel)

// Edge sums on the matrix cores.  Per row let y[0..95) be its head window x[0..95) (blockIdx.y = 0) or its tail
// window x[T-31 .. T+64) (blockIdx.y = 1), zero outside the row.  Everything the zero padding needs is inside the
// Gram block  P[a][j] = sum_rows y[a] y[j],  a < 31, j < 96:   x[s] x[s+d] = P_head[s][s+d],
// x[T-i] x[T-i+d] = P_tail[31-i][31-i+d];  a row of ones in slot a = 31 adds the column sums sum_rows y[j].
// The contraction runs over ROWS: one MFMA step takes 4 rows (k = lane >> 4), A = y[16 mt + i] (2 tiles),
// B = y[16 nt + i] (6 tiles), 6 loads and 12 MFMAs per 4 rows, 48 accumulator registers.  (The per-lane form --
// 62 lag products per lane and row behind an LDS window -- held 200+ VGPRs and paid a memory latency per row.)
__device__ __forceinline__ void eeg_stats_edge_kernel_body(const float* __restrict__ x,
                                                                         EegStats* __restrict__ st, int64_t rows,
                                                                         int T,
    unsigned zgx, unsigned zgy, unsigned zbz, unsigned zgz) {   // this zone's own gridDim.x/.y, blockIdx.z, gridDim.z
  __shared__ float tot[32][96];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, q = lane >> 4;
  const int which = blockIdx.y;
  const int base = which ? T - 31 : 0;
  for (int e = threadIdx.x; e < 32 * 96; e += 64 * kStatWaves) (&tot[0][0])[e] = 0.f;
  f32x4 acc[2][6];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 6; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int64_t n_grp = (rows + 3) >> 2;
  const int64_t stride = (int64_t)zgx * kStatWaves;
  for (int64_t g = (int64_t)blockIdx.x * kStatWaves + wave; g < n_grp; g += 2 * stride) {   // two groups in flight
    float b[2][6];
    bool live[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int64_t r = (g + h * stride) * 4 + q;
      live[h] = g + h * stride < n_grp && r < rows;
      const float* src = x + (live[h] ? r : 0) * (int64_t)T;
#pragma unroll
      for (int nt = 0; nt < 6; ++nt) {                              // clamped addresses, selects below: no test around a load
        const int t = base + 16 * nt + i;
        b[h][nt] = src[t < 0 ? 0 : (t < T ? t : T - 1)];
      }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int nt = 0; nt < 6; ++nt) {
        const int t = base + 16 * nt + i;
        b[h][nt] = (live[h] && t >= 0 && t < T) ? b[h][nt] : 0.f;
      }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float a0 = b[h][0];
      const float a1 = i == 15 ? (live[h] ? 1.f : 0.f) : b[h][1];   // slot a = 31: the row of ones
#pragma unroll
      for (int nt = 0; nt < 6; ++nt) {
        acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b[h][nt], acc[0][nt], 0, 0, 0);
        acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b[h][nt], acc[1][nt], 0, 0, 0);
      }
    }
  }
  for (int w = 0; w < kStatWaves; ++w) {                        // wave order (see eeg_stats_kernel)
    __syncthreads();
    if (wave == w) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 6; ++nt)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) tot[16 * mt + 4 * q + rr][16 * nt + i] += acc[mt][nt][rr];
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 32 * 96; e += 64 * kStatWaves)
    exact_add(&st->Pe[which][e / 96][e % 96], tot[e / 96][e % 96]);
}
ISD_ZONE_FN(eeg_stats_edge_kernel, 64 * kStatWaves)
__global__ __launch_bounds__(64 * kStatWaves) void eeg_stats_edge_kernel(const float* __restrict__ x,
                                                                         EegStats* __restrict__ st, int64_t rows,
                                                                         int T) {
  eeg_stats_edge_kernel_body(x, st, rows, T,
      gridDim.x, gridDim.y, blockIdx.z, gridDim.z);
}
ISD_ZONE_REGISTER(eeg_stats_edge_kernel)

// Short rows (T <= 79, e.g. the 65 frames of the stress features): the whole row fits one Gram matrix.  With
// y = (x[0..T), 1, 0, ...) padded to 16 MT entries, G[a][j] = sum_rows y[a] y[j] holds every lag product
// (x[s] x[s+d] = G[s][s+d]) and, in column T, the per-sample sums.  Contracted over rows like the edge blocks: MT
// loads and MT (MT + 1) / 2 MFMAs (upper-triangular tiles) per 4 rows -- 15 at T = 65 against 64 for the bulk + edge
// pair, one pass over x instead of two.
template <int MT>
__device__ __forceinline__ void eeg_stats_gram_kernel_body(const float* __restrict__ x,
                                                                         EegStats* __restrict__ st, int64_t rows,
                                                                         int T,
    unsigned zgx, unsigned zgy, unsigned zbz, unsigned zgz) {   // this zone's own gridDim.x/.y, blockIdx.z, gridDim.z
  __shared__ float tot[16 * MT][16 * MT];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, q = lane >> 4;
  for (int e = threadIdx.x; e < 256 * MT * MT; e += 64 * kStatWaves) (&tot[0][0])[e] = 0.f;
  f32x4 acc[MT][MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < MT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int64_t n_grp = (rows + 3) >> 2;
  const int64_t stride = (int64_t)zgx * kStatWaves;
  for (int64_t g = (int64_t)blockIdx.x * kStatWaves + wave; g < n_grp; g += 2 * stride) {   // two groups in flight
    float y[2][MT];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int64_t r = (g + h * stride) * 4 + q;
      const bool live = g + h * stride < n_grp && r < rows;
      const float* src = x + (live ? r : 0) * (int64_t)T;
#pragma unroll
      for (int t = 0; t < MT; ++t) {                            // clamped address, then selects: no test around a load
        const int idx = 16 * t + i;
        y[h][t] = src[idx < T ? idx : T - 1];
      }
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        const int idx = 16 * t + i;
        y[h][t] = !live ? 0.f : (idx < T ? y[h][t] : (idx == T ? 1.f : 0.f));
      }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = mt; nt < MT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(y[h][mt], y[h][nt], acc[mt][nt], 0, 0, 0);
  }
  for (int w = 0; w < kStatWaves; ++w) {                        // wave order (see eeg_stats_kernel)
    __syncthreads();
    if (wave == w) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = mt; nt < MT; ++nt)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) tot[16 * mt + 4 * q + rr][16 * nt + i] += acc[mt][nt][rr];
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 256 * MT * MT; e += 64 * kStatWaves) {
    const int a = e / (16 * MT), j = e - a * (16 * MT);
    if (j >= a && a <= T && j <= T) exact_add(&st->Gs[a][j], tot[a][j]);
  }
}
ISD_ZONE_FN_T(eeg_stats_gram_kernel, 64 * kStatWaves, int)
template <int MT>
__global__ __launch_bounds__(64 * kStatWaves) void eeg_stats_gram_kernel(const float* __restrict__ x,
                                                                         EegStats* __restrict__ st, int64_t rows,
                                                                         int T) {
  eeg_stats_gram_kernel_body<MT>(x, st, rows, T,
      gridDim.x, gridDim.y, blockIdx.z, gridDim.z);
}
ISD_ZONE_REGISTER_T(eeg_stats_gram_kernel, 1)
ISD_ZONE_REGISTER_T(eeg_stats_gram_kernel, 2)
ISD_ZONE_REGISTER_T(eeg_stats_gram_kernel, 3)
ISD_ZONE_REGISTER_T(eeg_stats_gram_kernel, 4)
ISD_ZONE_REGISTER_T(eeg_stats_gram_kernel, 5)

// Gs -> the quantities eeg_finalize1_kernel consumes (same definitions as eeg_stats_derive_kernel: samples outside
// the row are zero).  One block of 64 threads (thread = lag d).
__device__ __forceinline__ void eeg_stats_gram_derive_kernel_body(EegStats* __restrict__ st, int T,
    unsigned zgx, unsigned zgy, unsigned zbz, unsigned zgz) {   // this zone's own gridDim.x/.y, blockIdx.z, gridDim.z
  const int d = threadIdx.x;
  double a = 0.0;
  for (int s = 0; s + d < T; ++s) a += exact_get(&st->Gs[s][s + d]);
  st->A[d] = a;
  double h = 0.0, t = 0.0;
  for (int k = 1; k < 32; ++k) {
    const int s = k - 1;
    if (s + d < T) h += exact_get(&st->Gs[s][s + d]);                // H[k][d] = sum_{s<k} x[s] x[s+d]
    st->H[k][d] = h;
  }
  for (int e = 2; e < 33; ++e) {
    const int s = T - (e - 1);
    if (s >= 0 && s + d < T) t += exact_get(&st->Gs[s][s + d]);      // Tl[e][d] = sum_{i<e} x[T-i] x[T-i+d]
    st->Tl[e][d] = t;
  }
  if (d == 0) {
    double tot = 0.0, hs = 0.0, ts = 0.0;
    for (int j = 0; j < T; ++j) tot += exact_get(&st->Gs[j][T]);
    st->S = tot;
    st->Hs[0] = 0.0;
    for (int k = 1; k < 32; ++k) { if (k - 1 < T) hs += exact_get(&st->Gs[k - 1][T]); st->Hs[k] = hs; }
    st->Ts[0] = st->Ts[1] = 0.0;
    for (int e = 2; e < 33; ++e) { if (T - (e - 1) >= 0) ts += exact_get(&st->Gs[T - (e - 1)][T]); st->Ts[e] = ts; }
  }
}
ISD_ZONE_FN(eeg_stats_gram_derive_kernel, 64)
__global__ __launch_bounds__(64) void eeg_stats_gram_derive_kernel(EegStats* __restrict__ st, int T) {
  eeg_stats_gram_derive_kernel_body(st, T,
      gridDim.x, gridDim.y, blockIdx.z, gridDim.z);
}
ISD_ZONE_REGISTER(eeg_stats_gram_derive_kernel)

// raw accumulators -> the quantities eeg_finalize1_kernel consumes.  One block of 64 threads (thread = lag d).
__device__ __forceinline__ void eeg_stats_derive_kernel_body(EegStats* __restrict__ st,
    unsigned zgx, unsigned zgy, unsigned zbz, unsigned zgz) {   // this zone's own gridDim.x/.y, blockIdx.z, gridDim.z
  const int d = threadIdx.x;
  double a = 0.0;
  for (int i = 0; i < 16; ++i) a += exact_get(&st->D[(i + d) >> 4][i * 16 + ((i + d) & 15)]);
  st->A[d] = a;
  double h = 0.0, t = 0.0;
  for (int k = 1; k < 32; ++k) {
    h += exact_get(&st->Pe[0][k - 1][k - 1 + d]);                    // H[k][d] = sum_{s<k} x[s] x[s+d]
    st->H[k][d] = h;
  }
  for (int e = 2; e < 33; ++e) {
    t += exact_get(&st->Pe[1][32 - e][32 - e + d]);                  // Tl[e][d] = sum_{i<e} x[T-i] x[T-i+d],  i = e - 1
    st->Tl[e][d] = t;
  }
  if (d == 0) {
    double hs = 0.0, ts = 0.0;
    st->S = exact_get(&st->Sx);
    st->Hs[0] = 0.0;
    for (int k = 1; k < 32; ++k) { hs += exact_get(&st->Pe[0][31][k - 1]); st->Hs[k] = hs; }
    st->Ts[0] = st->Ts[1] = 0.0;
    for (int e = 2; e < 33; ++e) { ts += exact_get(&st->Pe[1][31][32 - e]); st->Ts[e] = ts; }
  }
}
ISD_ZONE_FN(eeg_stats_derive_kernel, 64)
__global__ __launch_bounds__(64) void eeg_stats_derive_kernel(EegStats* __restrict__ st) {
  eeg_stats_derive_kernel_body(st,
      gridDim.x, gridDim.y, blockIdx.z, gridDim.z);
}
ISD_ZONE_REGISTER(eeg_stats_derive_kernel)

// BN1 coefficients.  training: from the x statistics; eval: from the running buffers.  One block.
__device__ __forceinline__ void eeg_finalize1_kernel_body(const float* __restrict__ params, float* __restrict__ bufs,
                                                            const EegStats* __restrict__ st, EegCoef* __restrict__ co,
                                                            EegOff off, int C, int K, int T, int64_t rows,
                                                            int training, float momentum, float eps,
                                                            const unsigned long long* __restrict__ seed_dev,
    unsigned zgx, unsigned zgy, unsigned zbz, unsigned zgz) {   // this zone's own gridDim.x/.y, blockIdx.z, gridDim.z
  const int P = K / 2;
  if (threadIdx.x == 0) co->seed_add = seed_dev ? 0xD1342543DE82EF95ull * *seed_dev : 0ull;
  const double N1 = (double)rows * (double)(T + 2 * P - K + 1);
  const float* Wt = params + off.Wt;
  const float* Ws = params + off.Ws;
  __shared__ float wpart[kF2][16];
  __shared__ double rowdot[kF1][kMaxK];
  {
    // wsum[g] = sum_c Ws[g,c]: 16 threads per row, then a 16-term finish
    const int g = threadIdx.x >> 4, j = threadIdx.x & 15;
    // eight independent partial sums: a single chain of C/16 dependent loads was 0.09 ms at C = 5120
    const float* wr = Ws + g * C;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int c = j;
    for (; c + 112 < C; c += 128) {
#pragma unroll
      for (int i = 0; i < 8; ++i) s[i] += wr[c + 16 * i];
    }
    for (; c < C; c += 16) s[0] += wr[c];
    wpart[g][j] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  }
  __syncthreads();
  if (threadIdx.x < kF2) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += wpart[threadIdx.x][j];
    co->wsum[threadIdx.x] = s;
  }
  if (training) {
    // m[k], G[k][k'] with a = k - P:  range of s = t'+a is [max(a,0), min(T+a, T-1-d)]
    for (int e = threadIdx.x; e < K * K; e += 256) {
      const int k = e / K, k2 = e - k * K;
      const int ka = k < k2 ? k : k2, kb = k < k2 ? k2 : k;
      const int a = ka - P, a2 = kb - P, d = kb - ka;
      double v = st->A[d];
      if (a > 0) v -= st->H[a][d];
      if (a2 <= -2) v -= st->Tl[-a][d];                   // s from T+a+1 .. T-1 (terms past T-1-d vanish)
      co->G[k][k2] = v / N1;
    }
    for (int k = threadIdx.x; k < K; k += 256) {
      const int a = k - P;
      double v = st->S;
      if (a > 0) v -= st->Hs[a];
      if (a < 0) v -= st->Ts[-a];                          // s from T+a+1 .. T-1
      co->m[k] = v / N1;
    }
  }
  __syncthreads();
  if (training) {
    // rowdot[f][k] = sum_k2 G[k][k2] Wt[f][k2]: the 8 x K row products spread over the block
    for (int e = threadIdx.x; e < kF1 * K; e += 256) {
      const int f = e / K, k = e - f * K;
      double row = 0.0;
      for (int k2 = 0; k2 < K; ++k2) row += co->G[k][k2] * (double)Wt[f * K + k2];
      rowdot[f][k] = row;
    }
  }
  __syncthreads();
  if (threadIdx.x < kF1) {
    const int f = threadIdx.x;
    double mu, var;
    if (training) {
      mu = 0.0;
      double e2 = 0.0;
      for (int k = 0; k < K; ++k) {
        mu += (double)Wt[f * K + k] * co->m[k];
        e2 += (double)Wt[f * K + k] * rowdot[f][k];
      }
      var = e2 - mu * mu;
      if (var < 0.0) var = 0.0;
      bufs[kRm1 + f] = (1.f - momentum) * bufs[kRm1 + f] + momentum * (float)mu;
      bufs[kRv1 + f] = (1.f - momentum) * bufs[kRv1 + f] + momentum * (float)(var * N1 / (N1 > 1.0 ? N1 - 1.0 : 1.0));
    } else {
      mu = bufs[kRm1 + f];
      var = bufs[kRv1 + f];
    }
    const double sig = sqrt(var + (double)eps);
    co->mu1[f] = mu;
    co->sig1[f] = sig;
    const float s1 = (float)((double)params[off.g1 + f] / sig);
    co->s1[f] = s1;
    co->o1[f] = params[off.b1 + f] - (float)mu * s1;
  }
}
ISD_ZONE_FN(eeg_finalize1_kernel, 256)
__global__ __launch_bounds__(256) void eeg_finalize1_kernel(const float* __restrict__ params, float* __restrict__ bufs,
                                                            const EegStats* __restrict__ st, EegCoef* __restrict__ co,
                                                            EegOff off, int C, int K, int T, int64_t rows,
                                                            int training, float momentum, float eps,
                                                            const unsigned long long* __restrict__ seed_dev) {
  eeg_finalize1_kernel_body(params, bufs, st, co, off, C, K, T, rows, training, momentum, eps, seed_dev,
      gridDim.x, gridDim.y, blockIdx.z, gridDim.z);
}
ISD_ZONE_REGISTER(eeg_finalize1_kernel)

// z[b,g,t] = sum_c Ws[g,c] x[b,c,t] on the matrix cores: M = the 16 rows g, N = 16 time steps, K = 4 channels per
// MFMA.  One workgroup per (trial, time tile); its four waves take a quarter of the channels each (16 channels =
// 8 loads and 4 independent MFMA chains per step) and meet in LDS in a fixed order.  (One wave per tile walking
// all channels left 640 waves on the chip at C = 5120, T = 65: 0.29 ms; the scalar version before it 2.3 ms.)
__device__ __forceinline__ void eeg_spatial_kernel_body(const float* __restrict__ x, const float* __restrict__ Ws,
                                                          float* __restrict__ z, int C, int T, int n_tiles,
    unsigned zgx, unsigned zgy, unsigned zbz, unsigned zgz) {   // this zone's own gridDim.x/.y, blockIdx.z, gridDim.z
  __shared__ float red[3][4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, jl = lane & 15;
  const int tile = blockIdx.x;
  const int b = blockIdx.y;
  const int t = tile * 16 + jl;
  const bool tv = t < T;
  const float* xb = x + (int64_t)b * C * T + (tv ? t : 0);
  const float* wr = Ws + jl * C;
  const int cq = ((C + 15) / 16) * 4;                           // channels per wave, a multiple of 4
  const int c_lo = wave * cq, c_hi = c_lo + cq < C ? c_lo + cq : C;
  f32x4 acc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int c0 = c_lo;
  // 64 channels per round: 32 loads in flight, none behind a test (xb is clamped to a valid column; columns past T
  // are zeroed by a select).  With a branch around each x load and 16 channels per round every round waited out
  // its own memory latency: 1.8 ms for the 2.7 GB of the stress configuration's features.
  for (; c0 + 64 <= c_hi; c0 += 64) {
    float av[16], bv[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int c = c0 + 4 * k + q;
      av[k] = wr[c];
      bv[k] = xb[(int64_t)c * T];
    }
#pragma unroll
    for (int k = 0; k < 16; ++k)
      acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[k], tv ? bv[k] : 0.f, acc[k & 3], 0, 0, 0);
  }
  for (; c0 + 16 <= c_hi; c0 += 16) {
    float av[4], bv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = c0 + 4 * k + q;
      av[k] = wr[c];
      bv[k] = xb[(int64_t)c * T];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[k], tv ? bv[k] : 0.f, acc[k], 0, 0, 0);
  }
  for (; c0 < c_hi; c0 += 4) {
    const int c = c0 + q;
    const bool cv = c < c_hi;
    const float a0 = cv ? wr[c] : 0.f;
    const float b0 = (cv && tv) ? xb[(int64_t)c * T] : 0.f;
    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[0], 0, 0, 0);
  }
  f32x4 sum;
#pragma unroll
  for (int r = 0; r < 4; ++r) sum[r] = (acc[0][r] + acc[1][r]) + (acc[2][r] + acc[3][r]);
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave - 1][r][lane] = sum[r];
  }
  __syncthreads();
  if (wave == 0 && tv) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      z[((int64_t)b * kF2 + 4 * q + r) * T + t] = ((sum[r] + red[0][r][lane]) + red[1][r][lane]) + red[2][r][lane];
  }
}
ISD_ZONE_FN(eeg_spatial_kernel, 256)
__global__ __launch_bounds__(256) void eeg_spatial_kernel(const float* __restrict__ x, const float* __restrict__ Ws,
                                                          float* __restrict__ z, int C, int T, int n_tiles) {
  eeg_spatial_kernel_body(x, Ws, z, C, T, n_tiles,
      gridDim.x, gridDim.y, blockIdx.z, gridDim.z);
}
ISD_ZONE_REGISTER(eeg_spatial_kernel)

// The same product for wide inputs (hundreds of channels: the stress configuration's 5120 x 65 feature maps): one
// workgroup takes NTG consecutive time tiles of a trial, so every x row is read whole by ONE workgroup (with a
// workgroup per tile the five 64-byte pieces of a 260-byte row went to five workgroups on five XCDs, each pulling the
// row's cache lines into its own L2) and a Ws fragment serves NTG MFMAs instead of one.
template <int NTG>
__device__ __forceinline__ void eeg_spatial_rows_kernel_body(const float* __restrict__ x, const float* __restrict__ Ws,
                                                               float* __restrict__ z, int C, int T,
    unsigned zgx, unsigned zgy, unsigned zbz, unsigned zgz) {   // this zone's own gridDim.x/.y, blockIdx.z, gridDim.z
  __shared__ float red[3][NTG][4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, jl = lane & 15;
  const int b = blockIdx.y;
  bool tv[NTG];
  const float* xb[NTG];
#pragma unroll
  for (int j = 0; j < NTG; ++j) {
    const int t = (blockIdx.x * NTG + j) * 16 + jl;
    tv[j] = t < T;
    xb[j] = x + (int64_t)b * C * T + (tv[j] ? t : 0);            // clamped: no test around a load
  }
  const float* wr = Ws + jl * C;
  const int cq = ((C + 15) / 16) * 4;                           // channels per wave, a multiple of 4
  const int c_lo = wave * cq, c_hi = c_lo + cq < C ? c_lo + cq : C;
  f32x4 acc[NTG];
#pragma unroll
  for (int j = 0; j < NTG; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int c0 = c_lo;
  for (; c0 + 16 <= c_hi; c0 += 16) {
    float av[4], bv[4][NTG];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = c0 + 4 * k + q;
      av[k] = wr[c];
#pragma unroll
      for (int j = 0; j < NTG; ++j) bv[k][j] = xb[j][(int64_t)c * T];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int j = 0; j < NTG; ++j)
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[k], tv[j] ? bv[k][j] : 0.f, acc[j], 0, 0, 0);
  }
  for (; c0 < c_hi; c0 += 4) {
    const int c = c0 + q;
    const bool cv = c < c_hi;
    const int cc = cv ? c : c_hi - 1;
    const float a0 = cv ? wr[cc] : 0.f;
#pragma unroll
    for (int j = 0; j < NTG; ++j) {
      const float b0 = xb[j][(int64_t)cc * T];
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, (cv && tv[j]) ? b0 : 0.f, acc[j], 0, 0, 0);
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int j = 0; j < NTG; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave - 1][j][r][lane] = acc[j][r];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int j = 0; j < NTG; ++j) {
      const int t = (blockIdx.x * NTG + j) * 16 + jl;
      if (tv[j]) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          z[((int64_t)b * kF2 + 4 * q + r) * T + t] =
              ((acc[j][r] + red[0][j][r][lane]) + red[1][j][r][lane]) + red[2][j][r][lane];
      }
    }
  }
}
ISD_ZONE_FN_T(eeg_spatial_rows_kernel, 256, int)
template <int NTG>
__global__ __launch_bounds__(256) void eeg_spatial_rows_kernel(const float* __restrict__ x, const float* __restrict__ Ws,
                                                               float* __restrict__ z, int C, int T) {
  eeg_spatial_rows_kernel_body<NTG>(x, Ws, z, C, T,
      gridDim.x, gridDim.y, blockIdx.z, gridDim.z);
}
ISD_ZONE_REGISTER_T(eeg_spatial_rows_kernel, 5)

// u[b,g,t'] = sum_k Wt[f,k] zpad[b,g,t'+k]; per-g sums of u and u^2 (fp64 atomics).  grid (ceil(Tp/256), B*16)
// KT: the filter length as a compile-time constant (16 / 32 / 64: fully unrolled tap loops without a test per tap -- a
// run-time bound left every LDS read in its own basic block behind its own wait), 0 = any length
template <int KT>
__device__ __forceinline__ void eeg_tconv_kernel_body(const float* __restrict__ z, const float* __restrict__ Wt,
                                                        float* __restrict__ u, EegStats* __restrict__ st, int Krt, int T,
                                                        int Tp, int want_stats, int n_rows,
    unsigned zgx, unsigned zgy, unsigned zbz, unsigned zgz) {   // this zone's own gridDim.x/.y, blockIdx.z, gridDim.z
  const int K = KT ? KT : Krt;
  __shared__ float red[4];
  __shared__ float zs[256 + kMaxK];                    // zpad[tp0 .. tp0 + 256 + K): the tile's inputs, staged once
  // A workgroup strides over its row AND over every zgy-th row (zgy is a multiple of 16, so they share the
  // filter): the sums reach the 16 fp64 accumulators through one atomic pair per workgroup.  35 k workgroups queueing
  // on 16 addresses were what the kernel spent its time on at the stress shape (0.37 ms), and 5 k short rows per zone
  // what the zone heads spent theirs on (57 us per launch for 1.3 M outputs).
  // Each tile of 256 outputs reads its 256 + K inputs through LDS (one global read per input instead of K
  // bounds-checked ones per output: the raw stress head went 1.96 -> 1.79 ms).
  const int g = blockIdx.y & (kF2 - 1), f = g >> 1, P = K / 2;
  const float* w = Wt + f * K;
  float t1 = 0.f, t2 = 0.f;
  if constexpr (KT != 0) {
    // Compile-time filter length: a WAVE takes a tile of 256 outputs of its own row (the workgroup's four waves walk
    // four rows that share the filter) and a lane FOUR consecutive outputs, with the K + 4 inputs they touch in
    // registers (aligned 16-byte LDS reads): 4 K FMAs on (K + 4) / 4 LDS reads, where one output per lane paid an LDS
    // read per FMA.
    __shared__ __attribute__((aligned(16))) float zw4[4][256 + KT + 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* zt = zw4[wave];
    float wk[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) wk[k] = w[k];           // wave-uniform: scalar registers
    for (int bg0 = blockIdx.y; bg0 < n_rows; bg0 += 4 * (int)zgy) {
      const int bg = bg0 + wave * (int)zgy;
      if (bg >= n_rows) continue;                         // (no workgroup barrier below)
      const float* zr = z + (int64_t)bg * T;
      for (int tp0 = blockIdx.x * 256; tp0 < Tp; tp0 += zgx * 256) {
        wave_lds_sync();                                  // the previous tile's readers are done
#pragma unroll
        for (int j = lane; j < 256 + KT + 4; j += 64) {
          const int t = tp0 + j - P;
          const float zv = zr[t < 0 ? 0 : (t < T ? t : T - 1)];
          zt[j] = (t >= 0 && t < T) ? zv : 0.f;
        }
        wave_lds_sync();
        float win[KT + 4];
#pragma unroll
        for (int m = 0; m < (KT + 4) / 4; ++m) {
          const float4 qv = *reinterpret_cast<const float4*>(zt + 4 * lane + 4 * m);
          win[4 * m] = qv.x; win[4 * m + 1] = qv.y; win[4 * m + 2] = qv.z; win[4 * m + 3] = qv.w;
        }
        float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < KT; ++k)
#pragma unroll
          for (int j = 0; j < 4; ++j) a[j] = fmaf(wk[k], win[j + k], a[j]);
        const int tp = tp0 + 4 * lane;
        float* uo = u + (int64_t)bg * Tp + tp;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (tp + j < Tp) {
            uo[j] = a[j];
            t1 += a[j];
            t2 = fmaf(a[j], a[j], t2);
          }
      }
    }
  } else
  for (int bg = blockIdx.y; bg < n_rows; bg += zgy) {
    const float* zr = z + (int64_t)bg * T;
    for (int tp0 = blockIdx.x * 256; tp0 < Tp; tp0 += zgx * 256) {
      __syncthreads();                                  // the previous tile's readers are done
      for (int j = threadIdx.x; j < 256 + K; j += 256) {
        const int t = tp0 + j - P;
        const float zv = zr[t < 0 ? 0 : (t < T ? t : T - 1)];
        zs[j] = (t >= 0 && t < T) ? zv : 0.f;
      }
      __syncthreads();
      const int tp = tp0 + threadIdx.x;
      if (tp < Tp) {
        float acc = 0.f;
        if (KT) {
#pragma unroll
          for (int k = 0; k < (KT ? KT : 1); ++k) acc = fmaf(w[k], zs[threadIdx.x + k], acc);
        } else {
          for (int k = 0; k < K; ++k) acc = fmaf(w[k], zs[threadIdx.x + k], acc);
        }
        u[(int64_t)bg * Tp + tp] = acc;
        t1 += acc;
        t2 = fmaf(acc, acc, t2);
      }
    }
  }
  if (want_stats) {
    const float s1 = block_sum(t1, red);
    const float s2 = block_sum(t2, red);
    if (threadIdx.x == 0) {
      exact_add(&st->u1[g], s1);
      exact_add(&st->u2[g], s2);
    }
  }
}
ISD_ZONE_FN_T(eeg_tconv_kernel, 256, int)
template <int KT>
__global__ __launch_bounds__(256) void eeg_tconv_kernel(const float* __restrict__ z, const float* __restrict__ Wt,
                                                        float* __restrict__ u, EegStats* __restrict__ st, int Krt, int T,
                                                        int Tp, int want_stats, int n_rows) {
  eeg_tconv_kernel_body<KT>(z, Wt, u, st, Krt, T, Tp, want_stats, n_rows,
      gridDim.x, gridDim.y, blockIdx.z, gridDim.z);
}
ISD_ZONE_REGISTER_T(eeg_tconv_kernel, 64)
ISD_ZONE_REGISTER_T(eeg_tconv_kernel, 32)
ISD_ZONE_REGISTER_T(eeg_tconv_kernel, 16)
ISD_ZONE_REGISTER_T(eeg_tconv_kernel, 0)

__device__ __forceinline__ void eeg_finalize2_kernel_body(const float* __restrict__ params, float* __restrict__ bufs,
                                     const EegStats* __restrict__ st, EegCoef* __restrict__ co, EegOff off, double N2,
                                     int training, float momentum, float eps,
    unsigned zgx, unsigned zgy, unsigned zbz, unsigned zgz) {   // this zone's own gridDim.x/.y, blockIdx.z, gridDim.z
  const int g = threadIdx.x;
  if (g >= kF2) return;
  const int f = g >> 1;
  const float s1 = co->s1[f], c1 = co->o1[f] * co->wsum[g];
  double mu2, var2;
  if (training) {
    const double muu = exact_get(&st->u1[g]) / N2;
    double varu = exact_get(&st->u2[g]) / N2 - muu * muu;
    if (varu < 0.0) varu = 0.0;
    co->muu[g] = muu;
    co->varu[g] = varu;
    mu2 = (double)s1 * muu + (double)c1;
    var2 = (double)s1 * (double)s1 * varu;
    bufs[kRm2 + g] = (1.f - momentum) * bufs[kRm2 + g] + momentum * (float)mu2;
    bufs[kRv2 + g] = (1.f - momentum) * bufs[kRv2 + g] + momentum * (float)(var2 * N2 / (N2 > 1.0 ? N2 - 1.0 : 1.0));
  } else {
    mu2 = bufs[kRm2 + g];
    var2 = bufs[kRv2 + g];
  }
  const double sig2 = sqrt(var2 + (double)eps);
  const double g2 = params[off.g2 + g], b2 = params[off.b2 + g];
  co->mu2[g] = (float)mu2;
  co->sig2[g] = (float)sig2;
  co->A2[g] = (float)(g2 * (double)s1 / sig2);
  co->B2[g] = (float)(b2 + g2 * ((double)c1 - mu2) / sig2);
}
ISD_ZONE_FN(eeg_finalize2_kernel, 1024)
__global__ void eeg_finalize2_kernel(const float* __restrict__ params, float* __restrict__ bufs,
                                     const EegStats* __restrict__ st, EegCoef* __restrict__ co, EegOff off, double N2,
                                     int training, float momentum, float eps) {
  eeg_finalize2_kernel_body(params, bufs, st, co, off, N2, training, momentum, eps,
      gridDim.x, gridDim.y, blockIdx.z, gridDim.z);
}
ISD_ZONE_REGISTER(eeg_finalize2_kernel)

// p2[b,g,v] = mean_{r<P1} ELU(A2 u[b,g,P1 v+r] + B2)   (P1 = 4 EEGNet, 8 CVBlock)
__device__ __forceinline__ void eeg_pool2_kernel_body(const float* __restrict__ u, const EegCoef* __restrict__ co,
                                                        float* __restrict__ p2, int Tp, int T2, int P1, float dp,
                                                        uint64_t seed,
    unsigned zgx, unsigned zgy, unsigned zbz, unsigned zgz) {   // this zone's own gridDim.x/.y, blockIdx.z, gridDim.z
  const int bg = blockIdx.y, g = bg & (kF2 - 1);
  const int v = blockIdx.x * 256 + threadIdx.x;
  if (v >= T2) return;
  const float A = co->A2[g], Bc = co->B2[g];
  const float* ur = u + (int64_t)bg * Tp + P1 * v;
  float s = 0.f;
  for (int r = 0; r < P1; ++r) s += elu_f(fmaf(A, ur[r], Bc));
  p2[(int64_t)bg * T2 + v] = s / (float)P1 * drop_scale(seed + co->seed_add, (uint64_t)bg * T2 + v, dp);
}
ISD_ZONE_FN(eeg_pool2_kernel, 256)
__global__ __launch_bounds__(256) void eeg_pool2_kernel(const float* __restrict__ u, const EegCoef* __restrict__ co,
                                                        float* __restrict__ p2, int Tp, int T2, int P1, float dp,
                                                        uint64_t seed) {
  eeg_pool2_kernel_body(u, co, p2, Tp, T2, P1, dp, seed,
      gridDim.x, gridDim.y, blockIdx.z, gridDim.z);
}
ISD_ZONE_REGISTER(eeg_pool2_kernel)

// a3[b,g,w] = sum_k Wd[g,k] p2pad[b,g,w+k];  a4[b,h,w] = sum_g Wp[h,g] a3[b,g,w];  BN3 sums.
// One thread per (b, w); optionally stores a3.  grid (ceil(T2p/256), B)
__device__ __forceinline__ void eeg_sep_kernel_body(const float* __restrict__ p2, const float* __restrict__ Wd,
                                                      const float* __restrict__ Wp, float* __restrict__ a3out,
                                                      float* __restrict__ a4, EegStats* __restrict__ st, int T2,
                                                      int T2p, int want_stats, int B,
    unsigned zgx, unsigned zgy, unsigned zbz, unsigned zgz) {   // this zone's own gridDim.x/.y, blockIdx.z, gridDim.z
  __shared__ float red[4];
  // a workgroup walks every zgy-th trial and keeps its 32 per-channel sums in registers: one round of block
  // sums and atomics per workgroup instead of one per trial (short windows: 63 live threads, 32 block sums each)
  const int w = blockIdx.x * 256 + threadIdx.x;
  const bool live = w < T2p;
  float s1[kF2], s2[kF2];
#pragma unroll
  for (int h = 0; h < kF2; ++h) s1[h] = s2[h] = 0.f;
  for (int b = blockIdx.y; b < B; b += zgy) {
    float a3[kF2];
#pragma unroll
    for (int g = 0; g < kF2; ++g) {
      float acc = 0.f;
      if (live) {
        const float* pr = p2 + ((int64_t)b * kF2 + g) * T2;
#pragma unroll
        for (int k = 0; k < kK2; ++k) {                 // clamped address + select: no branch around the load, so the
          const int v = w + k - kP2;                    // 16 loads of a channel go out together
          const float pv = pr[v < 0 ? 0 : (v < T2 ? v : T2 - 1)];
          acc = fmaf(Wd[g * kK2 + k], (v >= 0 && v < T2) ? pv : 0.f, acc);
        }
        if (a3out) a3out[((int64_t)b * kF2 + g) * T2p + w] = acc;
      }
      a3[g] = acc;
    }
#pragma unroll
    for (int h = 0; h < kF2; ++h) {
      float acc = 0.f;
#pragma unroll
      for (int g = 0; g < kF2; ++g) acc = fmaf(Wp[h * kF2 + g], a3[g], acc);
      if (live) {
        a4[((int64_t)b * kF2 + h) * T2p + w] = acc;
        s1[h] += acc;
        s2[h] = fmaf(acc, acc, s2[h]);
      }
    }
  }
  if (want_stats) {
    // the 32 sums in one round: wave sums, one barrier, 32 threads finish (32 block_sum calls were 64 barriers)
    __shared__ float part[4][2 * kF2];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int h = 0; h < kF2; ++h) {
      const float t1 = wave_sum(s1[h]), t2 = wave_sum(s2[h]);
      if (lane == 0) {
        part[wv][h] = t1;
        part[wv][kF2 + h] = t2;
      }
    }
    __syncthreads();
    if (threadIdx.x < 2 * kF2) {
      const float tot = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
      if (threadIdx.x < kF2) exact_add(&st->a1[threadIdx.x], tot);
      else exact_add(&st->a2[threadIdx.x - kF2], tot);
    }
  }
}
ISD_ZONE_FN(eeg_sep_kernel, 256)
__global__ __launch_bounds__(256) void eeg_sep_kernel(const float* __restrict__ p2, const float* __restrict__ Wd,
                                                      const float* __restrict__ Wp, float* __restrict__ a3out,
                                                      float* __restrict__ a4, EegStats* __restrict__ st, int T2,
                                                      int T2p, int want_stats, int B) {
  eeg_sep_kernel_body(p2, Wd, Wp, a3out, a4, st, T2, T2p, want_stats, B,
      gridDim.x, gridDim.y, blockIdx.z, gridDim.z);
}
ISD_ZONE_REGISTER(eeg_sep_kernel)

__device__ __forceinline__ void eeg_finalize3_kernel_body(const float* __restrict__ params, float* __restrict__ bufs,
                                     const EegStats* __restrict__ st, EegCoef* __restrict__ co, EegOff off, double N3,
                                     int training, float momentum, float eps,
    unsigned zgx, unsigned zgy, unsigned zbz, unsigned zgz) {   // this zone's own gridDim.x/.y, blockIdx.z, gridDim.z
  const int h = threadIdx.x;
  if (h >= kF2) return;
  double mu, var;
  if (training) {
    mu = exact_get(&st->a1[h]) / N3;
    var = exact_get(&st->a2[h]) / N3 - mu * mu;
    if (var < 0.0) var = 0.0;
    bufs[kRm3 + h] = (1.f - momentum) * bufs[kRm3 + h] + momentum * (float)mu;
    bufs[kRv3 + h] = (1.f - momentum) * bufs[kRv3 + h] + momentum * (float)(var * N3 / (N3 > 1.0 ? N3 - 1.0 : 1.0));
  } else {
    mu = bufs[kRm3 + h];
    var = bufs[kRv3 + h];
  }
  const double sig = sqrt(var + (double)eps);
  const double g3 = params[off.g3 + h], b3 = params[off.b3 + h];
  co->mu3[h] = (float)mu;
  co->sig3[h] = (float)sig;
  co->A3[h] = (float)(g3 / sig);
  co->B3[h] = (float)(b3 - g3 * mu / sig);
}
ISD_ZONE_FN(eeg_finalize3_kernel, 1024)
__global__ void eeg_finalize3_kernel(const float* __restrict__ params, float* __restrict__ bufs,
                                     const EegStats* __restrict__ st, EegCoef* __restrict__ co, EegOff off, double N3,
                                     int training, float momentum, float eps) {
  eeg_finalize3_kernel_body(params, bufs, st, co, off, N3, training, momentum, eps,
      gridDim.x, gridDim.y, blockIdx.z, gridDim.z);
}
ISD_ZONE_REGISTER(eeg_finalize3_kernel)

// pooled[b,h] = mean_{w < 8*T3} ELU(A3 a4 + B3)   (AvgPool(1,8) floor + AdaptiveAvgPool); one wave per row
__device__ __forceinline__ void eeg_pool3_kernel_body(const float* __restrict__ a4, const EegCoef* __restrict__ co,
                                                        float* __restrict__ pooled, int64_t rows, int T2p, int T3,
                                                        float dp, uint64_t seed,
    unsigned zgx, unsigned zgy, unsigned zbz, unsigned zgz) {   // this zone's own gridDim.x/.y, blockIdx.z, gridDim.z
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const int h = (int)(row & (kF2 - 1));
  const float A = co->A3[h], Bc = co->B3[h];
  const float* ar = a4 + row * T2p;
  float s = 0.f;
  for (int w = lane; w < 8 * T3; w += 64)
    s += elu_f(fmaf(A, ar[w], Bc)) * drop_scale((seed + co->seed_add) ^ 0x5bd1e995u, (uint64_t)row * T3 + (w >> 3), dp);
  s = wave_sum(s);
  if (lane == 0) pooled[row] = T3 > 0 ? s / (float)(8 * T3) : 0.f;
}
ISD_ZONE_FN(eeg_pool3_kernel, 256)
__global__ __launch_bounds__(256) void eeg_pool3_kernel(const float* __restrict__ a4, const EegCoef* __restrict__ co,
                                                        float* __restrict__ pooled, int64_t rows, int T2p, int T3,
                                                        float dp, uint64_t seed) {
  eeg_pool3_kernel_body(a4, co, pooled, rows, T2p, T3, dp, seed,
      gridDim.x, gridDim.y, blockIdx.z, gridDim.z);
}
ISD_ZONE_REGISTER(eeg_pool3_kernel)

// ---------------------------------------------------------------- backward
// dy3 = de3 * ELU'(y3); sums of dy3 and dy3*xhat3 per h.  One wave per (b,h) row.
__device__ __forceinline__ void eeg_bwd3_sums_kernel_body(const float* __restrict__ a4,
                                                            const float* __restrict__ dpooled,
                                                            const EegCoef* __restrict__ co, EegStats* __restrict__ st,
                                                            int64_t rows, int T2p, int T3, float dp, uint64_t seed,
    unsigned zgx, unsigned zgy, unsigned zbz, unsigned zgz) {   // this zone's own gridDim.x/.y, blockIdx.z, gridDim.z
  // zgx is a multiple of 4: a wave's rows (every 4 zgx-th) share h, and it sends ONE atomic pair
  const int64_t row0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row0 >= rows) return;
  const int h = (int)(row0 & (kF2 - 1));
  const float A = co->A3[h], Bc = co->B3[h], mu = co->mu3[h], isg = 1.f / co->sig3[h];
  float s1 = 0.f, s2 = 0.f;
  for (int64_t row = row0; row < rows; row += (int64_t)zgx * 4) {
    const float de = T3 > 0 ? dpooled[row] / (float)(8 * T3) : 0.f;
    const float* ar = a4 + row * T2p;
    for (int w = lane; w < 8 * T3; w += 64) {
      const float av = ar[w];
      const float dy = de * drop_scale((seed + co->seed_add) ^ 0x5bd1e995u, (uint64_t)row * T3 + (w >> 3), dp) *
                       elu_grad_f(fmaf(A, av, Bc));
      s1 += dy;
      s2 += dy * (av - mu) * isg;
    }
  }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  if (lane == 0) {
    exact_add(&st->dy3s[h], s1);
    exact_add(&st->dy3x[h], s2);
  }
}
ISD_ZONE_FN(eeg_bwd3_sums_kernel, 256)
__global__ __launch_bounds__(256) void eeg_bwd3_sums_kernel(const float* __restrict__ a4,
                                                            const float* __restrict__ dpooled,
                                                            const EegCoef* __restrict__ co, EegStats* __restrict__ st,
                                                            int64_t rows, int T2p, int T3, float dp, uint64_t seed) {
  eeg_bwd3_sums_kernel_body(a4, dpooled, co, st, rows, T2p, T3, dp, seed,
      gridDim.x, gridDim.y, blockIdx.z, gridDim.z);
}
ISD_ZONE_REGISTER(eeg_bwd3_sums_kernel)

// BN backward coefficients for stage `which` (3 or 2) and the gamma/beta gradients
// gs: 1 / world size under synchronised BatchNorm -- the sums are then global on every rank and the gradient
// all-reduce that follows adds the ranks' copies
__device__ __forceinline__ void eeg_bwd_bn_coef_kernel_body(const float* __restrict__ params, float* __restrict__ dparams,
                                       const EegStats* __restrict__ st, EegCoef* __restrict__ co, EegOff off, double N,
                                       int which, double gs, int bn_train,
    unsigned zgx, unsigned zgy, unsigned zbz, unsigned zgz) {   // this zone's own gridDim.x/.y, blockIdx.z, gridDim.z
  const int h = threadIdx.x;
  if (h >= kF2) return;
  const double inv = bn_train ? 1.0 / N : 0.0;           // running statistics do not depend on the batch: no mean terms
  if (which == 3) {
    dparams[off.g3 + h] = (float)(exact_get(&st->dy3x[h]) * gs);
    dparams[off.b3 + h] = (float)(exact_get(&st->dy3s[h]) * gs);
    co->cA3[h] = params[off.g3 + h] / co->sig3[h];
    co->cB3[h] = (float)(exact_get(&st->dy3s[h]) * inv);
    co->cC3[h] = (float)(exact_get(&st->dy3x[h]) * inv);
  } else {
    dparams[off.g2 + h] = (float)(exact_get(&st->dy2x[h]) * gs);
    dparams[off.b2 + h] = (float)(exact_get(&st->dy2s[h]) * gs);
    co->cA2[h] = params[off.g2 + h] / co->sig2[h];
    co->cB2[h] = (float)(exact_get(&st->dy2s[h]) * inv);
    co->cC2[h] = (float)(exact_get(&st->dy2x[h]) * inv);
  }
}
ISD_ZONE_FN(eeg_bwd_bn_coef_kernel, 1024)
__global__ void eeg_bwd_bn_coef_kernel(const float* __restrict__ params, float* __restrict__ dparams,
                                       const EegStats* __restrict__ st, EegCoef* __restrict__ co, EegOff off, double N,
                                       int which, double gs, int bn_train) {
  eeg_bwd_bn_coef_kernel_body(params, dparams, st, co, off, N, which, gs, bn_train,
      gridDim.x, gridDim.y, blockIdx.z, gridDim.z);
}
ISD_ZONE_REGISTER(eeg_bwd_bn_coef_kernel)

// da4 = cA3 (dy3 - cB3 - xhat3 cC3);  da3[b,g,w] = sum_h Wp[h,g] da4[b,h,w].  One thread per (b,w).
__device__ __forceinline__ void eeg_bwd_sep_kernel_body(const float* __restrict__ a4,
                                                          const float* __restrict__ dpooled,
                                                          const float* __restrict__ Wp, const EegCoef* __restrict__ co,
                                                          float* __restrict__ da4, float* __restrict__ da3, int T2p,
                                                          int T3, float dp, uint64_t seed,
    unsigned zgx, unsigned zgy, unsigned zbz, unsigned zgz) {   // this zone's own gridDim.x/.y, blockIdx.z, gridDim.z
  const int b = blockIdx.y;
  const int w = blockIdx.x * 256 + threadIdx.x;
  if (w >= T2p) return;
  float d4[kF2];
#pragma unroll
  for (int h = 0; h < kF2; ++h) {
    const int64_t row = (int64_t)b * kF2 + h;
    const float av = a4[row * T2p + w];
    const float de = (w < 8 * T3 && T3 > 0)
                         ? dpooled[row] / (float)(8 * T3) * drop_scale((seed + co->seed_add) ^ 0x5bd1e995u, (uint64_t)row * T3 + (w >> 3), dp)
                         : 0.f;
    const float dy = de * elu_grad_f(fmaf(co->A3[h], av, co->B3[h]));
    const float xh = (av - co->mu3[h]) / co->sig3[h];
    d4[h] = co->cA3[h] * (dy - co->cB3[h] - xh * co->cC3[h]);
    da4[row * T2p + w] = d4[h];
  }
#pragma unroll
  for (int g = 0; g < kF2; ++g) {
    float acc = 0.f;
#pragma unroll
    for (int h = 0; h < kF2; ++h) acc = fmaf(Wp[h * kF2 + g], d4[h], acc);
    da3[((int64_t)b * kF2 + g) * T2p + w] = acc;
  }
}
ISD_ZONE_FN(eeg_bwd_sep_kernel, 256)
__global__ __launch_bounds__(256) void eeg_bwd_sep_kernel(const float* __restrict__ a4,
                                                          const float* __restrict__ dpooled,
                                                          const float* __restrict__ Wp, const EegCoef* __restrict__ co,
                                                          float* __restrict__ da4, float* __restrict__ da3, int T2p,
                                                          int T3, float dp, uint64_t seed) {
  eeg_bwd_sep_kernel_body(a4, dpooled, Wp, co, da4, da3, T2p, T3, dp, seed,
      gridDim.x, gridDim.y, blockIdx.z, gridDim.z);
}
ISD_ZONE_REGISTER(eeg_bwd_sep_kernel)

// dWp[h,g] = sum_{b,w} da4[b,h,w] a3[b,g,w];  dWd[g,k] = sum_{b,w} da3[b,g,w] p2pad[b,g,w+k].
// One block per output element (512 blocks), fp64 block result.
__device__ __forceinline__ void eeg_bwd_sepw_kernel_body(const float* __restrict__ da4, const float* __restrict__ a3,
                                                           const float* __restrict__ da3, const float* __restrict__ p2,
                                                           EegStats* __restrict__ st, int B, int T2, int T2p,
    unsigned zgx, unsigned zgy, unsigned zbz, unsigned zgz) {   // this zone's own gridDim.x/.y, blockIdx.z, gridDim.z
  __shared__ float red[4];
  const int o = blockIdx.x;
  // thread = (trial b0 + 4 j, step w0 + 64 i): no 64-bit division per element, two independent accumulators
  const int b0 = threadIdx.x >> 6, w0 = threadIdx.x & 63;
  float sa = 0.f, sb = 0.f;
  if (o < kF2 * kF2) {
    const int h = o / kF2, g = o - h * kF2;
    for (int b = b0; b < B; b += 8) {
      const bool two = b + 4 < B;
      const float* d0 = da4 + ((int64_t)b * kF2 + h) * T2p;
      const float* a0 = a3 + ((int64_t)b * kF2 + g) * T2p;
      const float* d1 = da4 + ((int64_t)(two ? b + 4 : b) * kF2 + h) * T2p;
      const float* a1 = a3 + ((int64_t)(two ? b + 4 : b) * kF2 + g) * T2p;
      for (int w = w0; w < T2p; w += 64) {
        sa = fmaf(d0[w], a0[w], sa);
        sb = fmaf(two ? d1[w] : 0.f, a1[w], sb);
      }
    }
  } else {
    const int o2 = o - kF2 * kF2;
    const int g = o2 / kK2, k = o2 - g * kK2;
    for (int b = b0; b < B; b += 4) {
      const float* d0 = da3 + ((int64_t)b * kF2 + g) * T2p;
      const float* p0 = p2 + ((int64_t)b * kF2 + g) * T2;
      for (int w = w0; w < T2p; w += 64) {
        const int v = w + k - kP2;
        const float pv = p0[v < 0 ? 0 : (v < T2 ? v : T2 - 1)];
        sa = fmaf(d0[w], (v >= 0 && v < T2) ? pv : 0.f, sa);
      }
    }
  }
  const float s = sa + sb;
  const float tot = block_sum(s, red);
  if (threadIdx.x == 0) {
    if (o < kF2 * kF2) st->dWp[o / kF2][o % kF2] = (double)tot;
    else st->dWd[(o - kF2 * kF2) / kK2][(o - kF2 * kF2) % kK2] = (double)tot;
  }
}
ISD_ZONE_FN(eeg_bwd_sepw_kernel, 256)
__global__ __launch_bounds__(256) void eeg_bwd_sepw_kernel(const float* __restrict__ da4, const float* __restrict__ a3,
                                                           const float* __restrict__ da3, const float* __restrict__ p2,
                                                           EegStats* __restrict__ st, int B, int T2, int T2p) {
  eeg_bwd_sepw_kernel_body(da4, a3, da3, p2, st, B, T2, T2p,
      gridDim.x, gridDim.y, blockIdx.z, gridDim.z);
}
ISD_ZONE_REGISTER(eeg_bwd_sepw_kernel)

// dp2 -> de2 -> dy2 = de2 ELU'(y2), written to dy2[b,g,t'] (zero past 4*T2); BN2 backward sums.
__device__ __forceinline__ void eeg_bwd_pool2_kernel_body(const float* __restrict__ da3, const float* __restrict__ Wd,
                                                            const float* __restrict__ u, const EegCoef* __restrict__ co,
                                                            float* __restrict__ dy2, EegStats* __restrict__ st, int Tp,
                                                            int T2, int T2p, int P1, float dpr, uint64_t seed,
                                                            int n_rows,
    unsigned zgx, unsigned zgy, unsigned zbz, unsigned zgz) {   // this zone's own gridDim.x/.y, blockIdx.z, gridDim.z
  __shared__ float red[4];
  const int g = blockIdx.y & (kF2 - 1);                 // zgy is a multiple of 16: the rows of a workgroup share g
  float t1 = 0.f, t2 = 0.f;
  for (int bg = blockIdx.y; bg < n_rows; bg += zgy)
  for (int tp = blockIdx.x * 256 + threadIdx.x; tp < Tp; tp += zgx * 256) {   // one atomic pair per workgroup
    float dy = 0.f, xh = 0.f;
    const int v = tp / P1;
    if (v < T2) {
      float dp = 0.f;                                   // dp2[v] = sum_k Wd[g,k] da3[v - k + 8]
      if (Wd) {
#pragma unroll
        for (int k = 0; k < kK2; ++k) {
          const int w = v - k + kP2;
          const float dv = da3[(int64_t)bg * T2p + (w < 0 ? 0 : (w < T2p ? w : T2p - 1))];
          dp = fmaf(Wd[g * kK2 + k], (w >= 0 && w < T2p) ? dv : 0.f, dp);
        }
      } else {
        dp = da3[(int64_t)bg * T2 + v];                 // CVBlock: dp2 already formed by cv_bwd_dp2_kernel
      }
      dp *= drop_scale(seed + co->seed_add, (uint64_t)bg * T2 + v, dpr);
      const float uv = u[(int64_t)bg * Tp + tp];
      const float y2 = fmaf(co->A2[g], uv, co->B2[g]);
      dy = dp / (float)P1 * elu_grad_f(y2);
      // xhat2 = (a2 - mu2)/sig2 with a2 = s1 u + c1  ==  (y2 - beta2)/gamma2 ; use the direct form
      xh = (co->s1[g >> 1] * uv + co->o1[g >> 1] * co->wsum[g] - co->mu2[g]) / co->sig2[g];
    }
    dy2[(int64_t)bg * Tp + tp] = dy;
    t1 += dy;
    t2 = fmaf(dy, xh, t2);
  }
  const float s1 = block_sum(t1, red);
  const float s2 = block_sum(t2, red);
  if (threadIdx.x == 0) {
    exact_add(&st->dy2s[g], s1);
    exact_add(&st->dy2x[g], s2);
  }
}
ISD_ZONE_FN(eeg_bwd_pool2_kernel, 256)
__global__ __launch_bounds__(256) void eeg_bwd_pool2_kernel(const float* __restrict__ da3, const float* __restrict__ Wd,
                                                            const float* __restrict__ u, const EegCoef* __restrict__ co,
                                                            float* __restrict__ dy2, EegStats* __restrict__ st, int Tp,
                                                            int T2, int T2p, int P1, float dpr, uint64_t seed,
                                                            int n_rows) {
  eeg_bwd_pool2_kernel_body(da3, Wd, u, co, dy2, st, Tp, T2, T2p, P1, dpr, seed, n_rows,
      gridDim.x, gridDim.y, blockIdx.z, gridDim.z);
}
ISD_ZONE_REGISTER(eeg_bwd_pool2_kernel)

// da2 = cA2 (dy2 - cB2 - xhat2 cC2) in place; Sd = sum da2, Su = sum da2*u
__device__ __forceinline__ void eeg_bwd_bn2_kernel_body(float* __restrict__ dy2, const float* __restrict__ u,
                                                          const EegCoef* __restrict__ co, EegStats* __restrict__ st,
                                                          int Tp, int n_rows,
    unsigned zgx, unsigned zgy, unsigned zbz, unsigned zgz) {   // this zone's own gridDim.x/.y, blockIdx.z, gridDim.z
  __shared__ float red[4];
  const int g = blockIdx.y & (kF2 - 1);                 // zgy is a multiple of 16: the rows of a workgroup share g
  float t1 = 0.f, t2 = 0.f;
  for (int bg = blockIdx.y; bg < n_rows; bg += zgy)
  for (int tp = blockIdx.x * 256 + threadIdx.x; tp < Tp; tp += zgx * 256) {   // one atomic pair per workgroup
    const float uv = u[(int64_t)bg * Tp + tp];
    const float xh = (co->s1[g >> 1] * uv + co->o1[g >> 1] * co->wsum[g] - co->mu2[g]) / co->sig2[g];
    const float d = co->cA2[g] * (dy2[(int64_t)bg * Tp + tp] - co->cB2[g] - xh * co->cC2[g]);
    dy2[(int64_t)bg * Tp + tp] = d;
    t1 += d;
    t2 = fmaf(d, uv, t2);
  }
  const float s1 = block_sum(t1, red);
  const float s2 = block_sum(t2, red);
  if (threadIdx.x == 0) {
    exact_add(&st->Sd[g], s1);
    exact_add(&st->Su[g], s2);
  }
}
ISD_ZONE_FN(eeg_bwd_bn2_kernel, 256)
__global__ __launch_bounds__(256) void eeg_bwd_bn2_kernel(float* __restrict__ dy2, const float* __restrict__ u,
                                                          const EegCoef* __restrict__ co, EegStats* __restrict__ st,
                                                          int Tp, int n_rows) {
  eeg_bwd_bn2_kernel_body(dy2, u, co, st, Tp, n_rows,
      gridDim.x, gridDim.y, blockIdx.z, gridDim.z);
}
ISD_ZONE_REGISTER(eeg_bwd_bn2_kernel)

// T1[g][k] = sum_{b,t'} da2[b,g,t'] zpad[b,g,t'+k]  (lane = t' inside a 64 block, K accumulators in registers)
// and v[b,g,t] = sum_k Wt[f,k] da2[b,g,t-k+P].   grid (B*16), one wave per (b,g) row.
// grid (B*16, segments): a workgroup takes kCorrSeg time steps of its row (with the whole row in LDS -- 33 KiB at
// T = 4096 -- a CU held 4 of these one-wave workgroups and the kernel was the slowest of the raw stress head).
constexpr int kCorrSeg = 1024;
template <int KT>
__device__ __forceinline__ void eeg_bwd_corr_kernel_body(const float* __restrict__ da2, const float* __restrict__ z,
                                                          const float* __restrict__ Wt, float* __restrict__ v,
                                                          EegStats* __restrict__ st, int Krt, int T, int Tp, int n_rows,
    unsigned zgx, unsigned zgy, unsigned zbz, unsigned zgz) {   // this zone's own gridDim.x/.y, blockIdx.z, gridDim.z
  const int K = KT ? KT : Krt;
  constexpr int KA = KT ? KT : kMaxK;                   // accumulators kept
  // z[s0 - P + j], j < L + K + 64;  da2[s0 - K + j], j < L + 2K + 64
  extern __shared__ __attribute__((aligned(16))) float sm[];
  // zgx is a multiple of 16: the rows of a workgroup (every zgx-th) share g, their K lag sums stay in
  // registers and reach the fp64 accumulators once per workgroup (5 k rows x 64 lags on 1 k addresses were 0.15 ms)
  const int g = blockIdx.x & (kF2 - 1), f = g >> 1, P = K / 2, lane = threadIdx.x;
  const int s0 = blockIdx.y * kCorrSeg, L = kCorrSeg;
  float* zp = sm;
  float* dp = sm + L + K + 64;
  float* w = dp + L + 2 * K + 64;                       // the filter's taps (64 scalars in the SGPR file spilled)
  for (int k = lane; k < K; k += 64) w[k] = Wt[f * K + k];
  float acc[KA];
#pragma unroll
  for (int k = 0; k < KA; ++k) acc[k] = 0.f;
  for (int bg = blockIdx.x; bg < n_rows; bg += zgx) {
    const float* zr = z + (int64_t)bg * T;
    const float* dr = da2 + (int64_t)bg * Tp;
    wave_lds_sync();                                    // the previous row's readers are done
    const int n1 = Tp - s0 < L ? Tp - s0 : L;           // t' = s0 + u, u < n1
    // what this segment touches (short rows: a fraction of L); the register-window loops below walk 256 steps a round
    const int Lr = KT ? (n1 + 255) & ~255 : (n1 + 63) & ~63;
    // clamped addresses + selects: a branch around each load would serialise them (one L2 round trip per iteration)
#pragma unroll 4
    for (int j = lane; j < Lr + K + 64; j += 64) {
      const int t = s0 - P + j;
      const float zv = zr[t < 0 ? 0 : (t < T ? t : T - 1)];
      zp[j] = (t >= 0 && t < T) ? zv : 0.f;
    }
#pragma unroll 4
    for (int j = lane; j < Lr + 2 * K + 64; j += 64) {
      const int t = s0 - K + j;
      const float dv = dr[t < 0 ? 0 : (t < Tp ? t : Tp - 1)];
      dp[j] = (t >= 0 && t < Tp) ? dv : 0.f;
    }
    wave_lds_sync();
    const int n2 = T - s0 < L ? T - s0 : L;             // t = s0 + u, u < n2
    if constexpr (KT != 0) {
      // A lane owns FOUR consecutive steps and holds the K + 4 inputs they touch in registers (aligned 16-byte LDS
      // reads): 2 x 4 K FMAs on (K + 4) / 4 + 1 LDS reads each.  With one step per lane every FMA had its own LDS
      // read and the LDS pipe, not the vector ALU, set the pace (0.52 ms at the zone shape of the FAST heads).
      constexpr int NW4 = (KT + 4) / 4;
      for (int u0 = 0; u0 < n1; u0 += 256) {
        const int ub = u0 + 4 * lane;
        float zw[KT + 4];
#pragma unroll
        for (int m = 0; m < NW4; ++m) {
          const float4 qv = *reinterpret_cast<const float4*>(zp + ub + 4 * m);
          zw[4 * m] = qv.x; zw[4 * m + 1] = qv.y; zw[4 * m + 2] = qv.z; zw[4 * m + 3] = qv.w;
        }
        const float4 dq = *reinterpret_cast<const float4*>(dp + K + ub);      // da2[s0 + ub ..], 0 beyond Tp
        const float dv[4] = {dq.x, dq.y, dq.z, dq.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int k = 0; k < KT; ++k) acc[k] = fmaf(dv[j], zw[j + k], acc[k]);   // z[t' + k - P]
      }
      for (int u0 = 0; u0 < n2; u0 += 256) {
        const int ub = u0 + 4 * lane;
        float dw[KT + 4];                                // dw[m] = da2[s0 + ub - K + P + m]
#pragma unroll
        for (int m = 0; m < NW4; ++m) {
          const float4 qv = *reinterpret_cast<const float4*>(dp + ub + P + 4 * m);
          dw[4 * m] = qv.x; dw[4 * m + 1] = qv.y; dw[4 * m + 2] = qv.z; dw[4 * m + 3] = qv.w;
        }
        float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k4 = 0; k4 < KT / 4; ++k4) {
          const float4 wq = *reinterpret_cast<const float4*>(w + 4 * k4);      // same address in every lane
          const float wv[4] = {wq.x, wq.y, wq.z, wq.w};
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] = fmaf(wv[e], dw[KT + j - (4 * k4 + e)], a[j]);   // da2[t - k + P]
        }
        float* vo = v + (int64_t)bg * T + s0 + ub;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (ub + j < n2) vo[j] = a[j];
      }
    } else {
      for (int u0 = 0; u0 < n1; u0 += 64) {
        const float dv = dp[K + u0 + lane];               // da2[s0 + u0 + lane], 0 beyond Tp
#pragma unroll
        for (int k = 0; k < KA; ++k)
          if (k < K) acc[k] = fmaf(dv, zp[u0 + lane + k], acc[k]);          // z[t' + k - P]
      }
      for (int u = lane; u < n2; u += 64) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < KA; ++k)
          if (k < K) a = fmaf(w[k], dp[K + u - k + P], a);                  // da2[t - k + P]
        v[(int64_t)bg * T + s0 + u] = a;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < KA; ++k) {
    if (KT || k < K) {
      const float tot = wave_sum(acc[k]);
      if (lane == 0) exact_add(&st->T1[g][k], tot);
    }
  }
}
ISD_ZONE_FN_T(eeg_bwd_corr_kernel, 64, int)
template <int KT>
__global__ __launch_bounds__(64) void eeg_bwd_corr_kernel(const float* __restrict__ da2, const float* __restrict__ z,
                                                          const float* __restrict__ Wt, float* __restrict__ v,
                                                          EegStats* __restrict__ st, int Krt, int T, int Tp, int n_rows) {
  eeg_bwd_corr_kernel_body<KT>(da2, z, Wt, v, st, Krt, T, Tp, n_rows,
      gridDim.x, gridDim.y, blockIdx.z, gridDim.z);
}
ISD_ZONE_REGISTER_T(eeg_bwd_corr_kernel, 64)
ISD_ZONE_REGISTER_T(eeg_bwd_corr_kernel, 32)
ISD_ZONE_REGISTER_T(eeg_bwd_corr_kernel, 16)
ISD_ZONE_REGISTER_T(eeg_bwd_corr_kernel, 0)

// dWs_raw[g,c] = sum_{b,t} v[b,g,t] x[b,c,t] on the matrix cores; persistent waves, partial slabs.
// A[g][t] = v, B[t][c] = x; K = 4 time steps per MFMA; a wave owns all C/16 channel tiles (<= 16) of its 256 channels.
// A chunk is kDwsPasses x 16 time steps of one trial.  Lane (q, jl) owns the 4 consecutive steps t0 + 4q .. + 3 of its
// row (one 16-byte load -- dword alignment is enough, tools/ubench/unaligned_x4.hip -- instead of four scalar ones);
// MFMA step e contracts element e of every lane, i.e. the K index q stands for time t0 + 4q + e in both operands.
// The channel tile is the OUTER loop inside a chunk: the five 64-byte pieces of a 260-byte row (the stress
// configuration's 65 frames) are loaded back to back, so each cache line is used while it is still in the L2.  With
// the pass outermost a row's lines were revisited ~20 us later, after 25 MB of other waves' rows per XCD had gone
// through its 4 MB L2: the kernel ran at 2 TB/s whatever the load width.
typedef float F4U __attribute__((ext_vector_type(4), aligned(4)));   // four floats at any dword address
constexpr int kDwsPasses = 5;
__device__ __forceinline__ void dws_load_row(const float* __restrict__ row, int t_lo, int t_hi, int q, bool zero_tail,
                                             float (&f)[kDwsPasses][4]) {
#pragma unroll
  for (int p = 0; p < kDwsPasses; ++p) {
    const int t0 = t_lo + 16 * p, t = t0 + 4 * q;
    if (t0 + 16 <= t_hi) {                                  // wave-uniform: every lane's four steps lie inside the row
      const F4U w = *reinterpret_cast<const F4U*>(row + t);
      f[p][0] = w.x; f[p][1] = w.y; f[p][2] = w.z; f[p][3] = w.w;
    } else {                                                // clamped addresses (and selects): no test around a load
#pragma unroll
      for (int e = 0; e < 4; ++e) f[p][e] = row[t + e < t_hi ? t + e : t_hi - 1];
      if (zero_tail) {
#pragma unroll
        for (int e = 0; e < 4; ++e) f[p][e] = t + e < t_hi ? f[p][e] : 0.f;
      }
    }
  }
}
__device__ __forceinline__ void eeg_bwd_dws_kernel_body(const float* __restrict__ v, const float* __restrict__ x,
                                                         float* __restrict__ part, int B, int C, int T,
    unsigned zgx, unsigned zgy, unsigned zbz, unsigned zgz) {   // this zone's own gridDim.x/.y, blockIdx.z, gridDim.z
  const int lane = threadIdx.x, q = lane >> 4, jl = lane & 15;
  const int c_base = blockIdx.y * 256;                    // 16 channel tiles per wave
  const int n_ctile = (C - c_base + 15) / 16 < 16 ? (C - c_base + 15) / 16 : 16;
  f32x4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr int CH = kDwsPasses * 16;
  const int chunks_per_b = (T + CH - 1) / CH;
  const int64_t n_chunks = (int64_t)B * chunks_per_b;
  for (int64_t ch = blockIdx.x; ch < n_chunks; ch += zgx) {
    const int64_t b = ch / chunks_per_b;
    const int t_lo = (int)(ch - b * chunks_per_b) * CH;
    const int t_hi = t_lo + CH < T ? t_lo + CH : T;
    float af[kDwsPasses][4];                                // steps past the row's end: zero (x there is finite)
    dws_load_row(v + (b * kF2 + jl) * (int64_t)T, t_lo, t_hi, q, true, af);
    int ne[kDwsPasses];                                     // MFMA steps of a pass that see any valid time step
#pragma unroll
    for (int p = 0; p < kDwsPasses; ++p) {
      const int left = t_hi - (t_lo + 16 * p);
      ne[p] = left >= 4 ? 4 : (left > 0 ? left : 0);
    }
#pragma unroll
    for (int ct = 0; ct < 16; ++ct) {
      if (ct < n_ctile) {                                   // wave-uniform
        const int c = c_base + ct * 16 + jl;
        float bf[kDwsPasses][4];
        dws_load_row(x + (b * C + (c < C ? c : C - 1)) * (int64_t)T, t_lo, t_hi, q, false, bf);   // channels past C: never stored
#pragma unroll
        for (int p = 0; p < kDwsPasses; ++p)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (e < ne[p]) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[p][e], bf[p][e], acc[ct], 0, 0, 0);
      }
    }
  }
  float* slab = part + (int64_t)blockIdx.x * kF2 * C;
#pragma unroll
  for (int ct = 0; ct < 16; ++ct) {
    if (ct < n_ctile) {
      const int c = c_base + ct * 16 + jl;
      if (c < C) {
#pragma unroll
        for (int r = 0; r < 4; ++r) slab[(4 * q + r) * C + c] = acc[ct][r];
      }
    }
  }
}
ISD_ZONE_FN(eeg_bwd_dws_kernel, 64)
__global__ __launch_bounds__(64) void eeg_bwd_dws_kernel(const float* __restrict__ v, const float* __restrict__ x,
                                                         float* __restrict__ part, int B, int C, int T) {
  eeg_bwd_dws_kernel_body(v, x, part, B, C, T,
      gridDim.x, gridDim.y, blockIdx.z, gridDim.z);
}
ISD_ZONE_REGISTER(eeg_bwd_dws_kernel)

// dWs[g,c] = s1[f] * sum_slabs raw + o1[f] * Sd[g].  Block = 64 elements x 16 slab groups (coalesced 256-B rows, two
// independent chains per group, LDS combine in a fixed order): one thread per element walking up to 1024 slabs on
// its own was a 0.15 ms chain of dependent loads.
__device__ __forceinline__ void eeg_bwd_dws_reduce_kernel_body(const float* __restrict__ part, int n_slabs,
                                                                  const EegStats* __restrict__ st,
                                                                  const EegCoef* __restrict__ co,
                                                                  float* __restrict__ dWs, int C,
    unsigned zgx, unsigned zgy, unsigned zbz, unsigned zgz) {   // this zone's own gridDim.x/.y, blockIdx.z, gridDim.z
  __shared__ float red[16][64];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + lane;
  const int n = kF2 * C;
  float s0 = 0.f, s1 = 0.f;
  if (e < n) {
    int k = grp;
    for (; k + 16 < n_slabs; k += 32) {
      s0 += part[(int64_t)k * n + e];
      s1 += part[(int64_t)(k + 16) * n + e];
    }
    if (k < n_slabs) s0 += part[(int64_t)k * n + e];
  }
  red[grp][lane] = s0 + s1;
  __syncthreads();
  if (grp == 0 && e < n) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += red[i][lane];
    const int g = e / C, f = g >> 1;
    dWs[e] = co->s1[f] * t + co->o1[f] * (float)exact_get(&st->Sd[g]);
  }
}
ISD_ZONE_FN(eeg_bwd_dws_reduce_kernel, 1024)
__global__ __launch_bounds__(1024) void eeg_bwd_dws_reduce_kernel(const float* __restrict__ part, int n_slabs,
                                                                  const EegStats* __restrict__ st,
                                                                  const EegCoef* __restrict__ co,
                                                                  float* __restrict__ dWs, int C) {
  eeg_bwd_dws_reduce_kernel_body(part, n_slabs, st, co, dWs, C,
      gridDim.x, gridDim.y, blockIdx.z, gridDim.z);
}
ISD_ZONE_REGISTER(eeg_bwd_dws_reduce_kernel)

// Final assembly of the stage-1 gradients (dWt, dgamma1, dbeta1) and the separable weights.  One block.
__device__ __forceinline__ void eeg_bwd_final_kernel_body(const float* __restrict__ params,
                                                            float* __restrict__ dparams,
                                                            const EegStats* __restrict__ st,
                                                            const EegCoef* __restrict__ co, EegOff off, int C, int K,
                                                            double N1, int separable, double gs, int bn_train,
    unsigned zgx, unsigned zgy, unsigned zbz, unsigned zgz) {   // this zone's own gridDim.x/.y, blockIdx.z, gridDim.z
  const float* Wt = params + off.Wt;
  if (separable) {
    for (int e = threadIdx.x; e < kF2 * kF2; e += 256) dparams[off.Wp + e] = (float)st->dWp[e / kF2][e % kF2];
    for (int e = threadIdx.x; e < kF2 * kK2; e += 256) dparams[off.Wd + e] = (float)st->dWd[e / kK2][e % kK2];
  }
  __shared__ double SD[kF1], SU[kF1];
  if (threadIdx.x < kF1) {
    const int f = threadIdx.x;
    const double sd = (double)co->wsum[2 * f] * exact_get(&st->Sd[2 * f]) + (double)co->wsum[2 * f + 1] * exact_get(&st->Sd[2 * f + 1]);
    const double su = exact_get(&st->Su[2 * f]) + exact_get(&st->Su[2 * f + 1]);
    SD[f] = sd;
    SU[f] = su;
    dparams[off.b1 + f] = (float)(sd * gs);
    dparams[off.g1 + f] = (float)((su - co->mu1[f] * sd) / co->sig1[f] * gs);
  }
  __syncthreads();
  // dWt[f,k] = (g1/sig1) [ T1 - meanD N1 m[k] - (meanDa/sig1) (N1 (Wt G)[k] - mu1 N1 m[k]) ]
  for (int e = threadIdx.x; e < kF1 * K; e += 256) {
    const int f = e / K, k = e - f * K;
    const double sig = co->sig1[f], mu = co->mu1[f];
    const double t1 = exact_get(&st->T1[2 * f][k]) + exact_get(&st->T1[2 * f + 1][k]);
    const double meanD = bn_train ? SD[f] / N1 : 0.0;
    const double meanDa = bn_train ? (SU[f] - mu * SD[f]) / (sig * N1) : 0.0;
    double wg = 0.0;
    for (int k2 = 0; k2 < K; ++k2) wg += (double)Wt[f * K + k2] * co->G[k2][k];
    const double val = ((double)params[off.g1 + f] / sig) *
                       (t1 - meanD * N1 * co->m[k] - (meanDa / sig) * (N1 * wg - mu * N1 * co->m[k]));
    dparams[off.Wt + e] = (float)(val * gs);
  }
}
ISD_ZONE_FN(eeg_bwd_final_kernel, 256)
__global__ __launch_bounds__(256) void eeg_bwd_final_kernel(const float* __restrict__ params,
                                                            float* __restrict__ dparams,
                                                            const EegStats* __restrict__ st,
                                                            const EegCoef* __restrict__ co, EegOff off, int C, int K,
                                                            double N1, int separable, double gs, int bn_train) {
  eeg_bwd_final_kernel_body(params, dparams, st, co, off, C, K, N1, separable, gs, bn_train,
      gridDim.x, gridDim.y, blockIdx.z, gridDim.z);
}
ISD_ZONE_REGISTER(eeg_bwd_final_kernel)

// Gradient w.r.t. the input trials (what the attribution scripts differentiate).  With a1[f,c,t'] = (Wt_f * xpad_c)[t'],
// BN1 and the depthwise spatial layer:
//   dx[b,c,t] = sum_g s1[f] Ws[g,c] v[b,g,t]                                        (v: eeg_bwd_corr_kernel)
//             - sum_f sum_k Wt[f,k] ( c3_f a1[b,f,c,t-k+P] + c2_f )  over 0 <= t-k+P < Tp,      batch statistics only:
//   c3_f = g1_f meanDa_f / sig1_f^2,  c2_f = (g1_f / sig1_f) meanD_f - c3_f mu1_f   (BatchNorm's mean / variance paths;
//   meanD, meanDa as in eeg_bwd_final_kernel).  One workgroup per (row, 256-sample segment): the row segment with a
//   64-sample halo in LDS, per filter the a1 values the segment's outputs touch, then the transposed filter.
template <int KT>
__device__ __forceinline__ void eeg_bwd_dx_kernel_body(const float* __restrict__ x, const float* __restrict__ v,
                                                         const float* __restrict__ params, float* __restrict__ dx,
                                                         const EegStats* __restrict__ st, const EegCoef* __restrict__ co,
                                                         EegOff off, int C, int Krt, int T, int Tp, double N1,
                                                         int bn_train,
    unsigned zgx, unsigned zgy, unsigned zbz, unsigned zgz) {   // this zone's own gridDim.x/.y, blockIdx.z, gridDim.z
  const int K = KT ? KT : Krt;
  constexpr int KA = KT ? KT : kMaxK;
  __shared__ float xs[256 + 2 * kMaxK];                  // x[s0 - K + j]
  __shared__ float tmp[256 + kMaxK];                     // c3 a1[t'] + c2 for t' = s0 - (K - 1 - P) + u, 0 outside [0, Tp)
  __shared__ float wt[kMaxK];
  __shared__ float c2s[kF1], c3s[kF1];
  const int64_t row = blockIdx.x;                        // b * C + c
  const int64_t b = row / C;
  const int c = (int)(row - b * C);
  const int s0 = blockIdx.y * 256, tid = threadIdx.x, P = K / 2;
  const int t = s0 + tid;
  const float* Ws = params + off.Ws;
  float acc = 0.f;
  if (t < T) {
    const float* vr = v + b * kF2 * (int64_t)T + t;
#pragma unroll
    for (int g = 0; g < kF2; ++g) acc = fmaf(co->s1[g >> 1] * Ws[g * C + c], vr[(int64_t)g * T], acc);
  }
  if (bn_train) {
    if (tid < kF1) {
      const int f = tid;
      const double sd = (double)co->wsum[2 * f] * exact_get(&st->Sd[2 * f]) + (double)co->wsum[2 * f + 1] * exact_get(&st->Sd[2 * f + 1]);
      const double su = exact_get(&st->Su[2 * f]) + exact_get(&st->Su[2 * f + 1]);
      const double sig = co->sig1[f], mu = co->mu1[f], g1 = params[off.g1 + f];
      const double meanD = sd / N1, meanDa = (su - mu * sd) / (sig * N1);
      const double c3 = g1 * meanDa / (sig * sig);
      c3s[f] = (float)c3;
      c2s[f] = (float)(g1 / sig * meanD - c3 * mu);
    }
    const float* xr = x + row * (int64_t)T;
    for (int j = tid; j < 256 + 2 * K; j += 256) {
      const int tt = s0 - K + j;
      xs[j] = (tt >= 0 && tt < T) ? xr[tt] : 0.f;
    }
    const int u_lo = s0 - (K - 1 - P);                    // t' of tmp[0]
    for (int f = 0; f < kF1; ++f) {
      __syncthreads();                                   // xs / coefficients ready; previous filter's tmp consumed
      if (tid < K) wt[tid] = params[off.Wt + f * K + tid];
      __syncthreads();
      for (int u = tid; u < 256 + K; u += 256) {
        const int tp = u_lo + u;
        float a = 0.f;
        if (tp >= 0 && tp < Tp) {
          // a1[t'] = sum_k Wt[k] x[t' + k - P];  xs index of x[t' - P] is t' - P - s0 + K
          const float* xw = xs + (tp - P - s0 + K);
#pragma unroll
          for (int k = 0; k < KA; ++k)
            if (KT || k < K) a = fmaf(wt[k], xw[k], a);
          a = fmaf(c3s[f], a, c2s[f]);
        }
        tmp[u] = a;
      }
      __syncthreads();
      // t' = t - k + P  ->  u = t' - u_lo = tid - k + K - 1
      float sub = 0.f;
#pragma unroll
      for (int k = 0; k < KA; ++k)
        if (KT || k < K) sub = fmaf(wt[k], tmp[tid - k + K - 1], sub);
      acc -= sub;
    }
  }
  if (t < T) dx[row * (int64_t)T + t] = acc;
}
ISD_ZONE_FN_T(eeg_bwd_dx_kernel, 256, int)
template <int KT>
__global__ __launch_bounds__(256) void eeg_bwd_dx_kernel(const float* __restrict__ x, const float* __restrict__ v,
                                                         const float* __restrict__ params, float* __restrict__ dx,
                                                         const EegStats* __restrict__ st, const EegCoef* __restrict__ co,
                                                         EegOff off, int C, int Krt, int T, int Tp, double N1,
                                                         int bn_train) {
  eeg_bwd_dx_kernel_body<KT>(x, v, params, dx, st, co, off, C, Krt, T, Tp, N1, bn_train,
      gridDim.x, gridDim.y, blockIdx.z, gridDim.z);
}
ISD_ZONE_REGISTER_T(eeg_bwd_dx_kernel, 64)
ISD_ZONE_REGISTER_T(eeg_bwd_dx_kernel, 32)
ISD_ZONE_REGISTER_T(eeg_bwd_dx_kernel, 16)
ISD_ZONE_REGISTER_T(eeg_bwd_dx_kernel, 0)

// ------------------------------------------------------------------------------------------------
// CVBlock (reference: src/fast/models/fast.py:32-100).  Stage 1 (temporal conv, BN1, depthwise spatial
// conv, BN2, ELU) is the EEGNet one with AvgPool 8; stage 2 is a full 16->16 (1,16) convolution, BN3,
// ELU, AvgPool 2, and the projector is Linear(16*T3 -> F) over the flattened [16, T3] map.
// ------------------------------------------------------------------------------------------------
// W3 [h][g][k] -> fwd layout [g][k][h] and data-gradient layout [h][k][g] (16 contiguous scalars per tap)
__device__ __forceinline__ void cv_prep_kernel_body(const float* __restrict__ W3, float* __restrict__ Wf,
                                                      float* __restrict__ Wb,
    unsigned zgx, unsigned zgy, unsigned zbz, unsigned zgz) {   // this zone's own gridDim.x/.y, blockIdx.z, gridDim.z
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= kF2 * kF2 * kK2) return;
  const int h = e >> 8, g = (e >> 4) & 15, k = e & 15;
  const float w = W3[e];
  Wf[(g * kK2 + k) * kF2 + h] = w;
  Wb[(h * kK2 + k) * kF2 + g] = w;
}
ISD_ZONE_FN(cv_prep_kernel, 256)
__global__ __launch_bounds__(256) void cv_prep_kernel(const float* __restrict__ W3, float* __restrict__ Wf,
                                                      float* __restrict__ Wb) {
  cv_prep_kernel_body(W3, Wf, Wb,
      gridDim.x, gridDim.y, blockIdx.z, gridDim.z);
}
ISD_ZONE_REGISTER(cv_prep_kernel)

// a4[b,h,w] = sum_{g,k} W3[h,g,k] p2pad[b,g,w+k];  BN3 sums.  One thread per (b,w), 16 outputs in registers.
__device__ __forceinline__ void cv_conv3_kernel_body(const float* __restrict__ p2, const float* __restrict__ Wf,
                                                       float* __restrict__ a4, EegStats* __restrict__ st, int64_t B,
                                                       int T2, int T2p, int want_stats,
    unsigned zgx, unsigned zgy, unsigned zbz, unsigned zgz) {   // this zone's own gridDim.x/.y, blockIdx.z, gridDim.z
  __shared__ float red[4];
  __shared__ float tot[2 * kF2];
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = e < B * T2p;
  const int64_t b = live ? e / T2p : 0;
  const int w = live ? (int)(e - b * T2p) : 0;
  float o[kF2];
#pragma unroll
  for (int h = 0; h < kF2; ++h) o[h] = 0.f;
  if (live) {
    for (int g = 0; g < kF2; ++g) {
      const float* pr = p2 + (b * kF2 + g) * T2;
#pragma unroll
      for (int k = 0; k < kK2; ++k) {
        const int v = w + k - kP2;
        const float pv = (v >= 0 && v < T2) ? pr[v] : 0.f;
        const float* wv = Wf + (g * kK2 + k) * kF2;
#pragma unroll
        for (int h = 0; h < kF2; ++h) o[h] = fmaf(wv[h], pv, o[h]);
      }
    }
#pragma unroll
    for (int h = 0; h < kF2; ++h) a4[(b * kF2 + h) * T2p + w] = o[h];
  }
  if (want_stats) {
#pragma unroll
    for (int h = 0; h < kF2; ++h) {
      const float s1 = block_sum(o[h], red);
      const float s2 = block_sum(o[h] * o[h], red);
      if (threadIdx.x == 0) { tot[h] = s1; tot[kF2 + h] = s2; }
    }
    __syncthreads();
    if (threadIdx.x < kF2) exact_add(&st->a1[threadIdx.x], tot[threadIdx.x]);
    else if (threadIdx.x < 2 * kF2) exact_add(&st->a2[threadIdx.x - kF2], tot[threadIdx.x]);
  }
}
ISD_ZONE_FN(cv_conv3_kernel, 256)
__global__ __launch_bounds__(256) void cv_conv3_kernel(const float* __restrict__ p2, const float* __restrict__ Wf,
                                                       float* __restrict__ a4, EegStats* __restrict__ st, int64_t B,
                                                       int T2, int T2p, int want_stats) {
  cv_conv3_kernel_body(p2, Wf, a4, st, B, T2, T2p, want_stats,
      gridDim.x, gridDim.y, blockIdx.z, gridDim.z);
}
ISD_ZONE_REGISTER(cv_conv3_kernel)

// p3[b,h,v] = mean_{r<P2} ELU(A3 a4[b,h,P2 v+r] + B3) * dropout      ([B,16,T3] == the flattened projector input)
__device__ __forceinline__ void cv_pool3_kernel_body(const float* __restrict__ a4, const EegCoef* __restrict__ co,
                                                       float* __restrict__ p3, int64_t n, int T2p, int T3, int P2,
                                                       float dp, uint64_t seed,
    unsigned zgx, unsigned zgy, unsigned zbz, unsigned zgz) {   // this zone's own gridDim.x/.y, blockIdx.z, gridDim.z
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  const int64_t row = e / T3;
  const int v = (int)(e - row * T3), h = (int)(row & (kF2 - 1));
  const float A = co->A3[h], Bc = co->B3[h];
  const float* ar = a4 + row * T2p + P2 * v;
  float s = 0.f;
  for (int r = 0; r < P2; ++r) s += elu_f(fmaf(A, ar[r], Bc));
  p3[e] = s / (float)P2 * drop_scale((seed + co->seed_add) ^ 0x5bd1e995u, (uint64_t)e, dp);
}
ISD_ZONE_FN(cv_pool3_kernel, 256)
__global__ __launch_bounds__(256) void cv_pool3_kernel(const float* __restrict__ a4, const EegCoef* __restrict__ co,
                                                       float* __restrict__ p3, int64_t n, int T2p, int T3, int P2,
                                                       float dp, uint64_t seed) {
  cv_pool3_kernel_body(a4, co, p3, n, T2p, T3, P2, dp, seed,
      gridDim.x, gridDim.y, blockIdx.z, gridDim.z);
}
ISD_ZONE_REGISTER(cv_pool3_kernel)

__device__ __forceinline__ float cv_dy3(const float* __restrict__ dp3, const EegCoef* __restrict__ co, int64_t row,
                                        int h, int w, float av, int T3, int P2, float dp, uint64_t seed) {
  if (w >= P2 * T3) return 0.f;
  const int64_t e = row * T3 + w / P2;
  const float de = dp3[e] / (float)P2 * drop_scale((seed + co->seed_add) ^ 0x5bd1e995u, (uint64_t)e, dp);
  return de * elu_grad_f(fmaf(co->A3[h], av, co->B3[h]));
}

// BN3 backward sums: one wave per (b,h) row
__device__ __forceinline__ void cv_bwd3_sums_kernel_body(const float* __restrict__ a4, const float* __restrict__ dp3,
                                                           const EegCoef* __restrict__ co, EegStats* __restrict__ st,
                                                           int64_t rows, int T2p, int T3, int P2, float dp,
                                                           uint64_t seed,
    unsigned zgx, unsigned zgy, unsigned zbz, unsigned zgz) {   // this zone's own gridDim.x/.y, blockIdx.z, gridDim.z
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const int h = (int)(row & (kF2 - 1));
  const float mu = co->mu3[h], isg = 1.f / co->sig3[h];
  float s1 = 0.f, s2 = 0.f;
  for (int w = lane; w < P2 * T3; w += 64) {
    const float av = a4[row * T2p + w];
    const float dy = cv_dy3(dp3, co, row, h, w, av, T3, P2, dp, seed);
    s1 += dy;
    s2 += dy * (av - mu) * isg;
  }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  if (lane == 0) {
    exact_add(&st->dy3s[h], s1);
    exact_add(&st->dy3x[h], s2);
  }
}
ISD_ZONE_FN(cv_bwd3_sums_kernel, 256)
__global__ __launch_bounds__(256) void cv_bwd3_sums_kernel(const float* __restrict__ a4, const float* __restrict__ dp3,
                                                           const EegCoef* __restrict__ co, EegStats* __restrict__ st,
                                                           int64_t rows, int T2p, int T3, int P2, float dp,
                                                           uint64_t seed) {
  cv_bwd3_sums_kernel_body(a4, dp3, co, st, rows, T2p, T3, P2, dp, seed,
      gridDim.x, gridDim.y, blockIdx.z, gridDim.z);
}
ISD_ZONE_REGISTER(cv_bwd3_sums_kernel)

// da4 = cA3 (dy3 - cB3 - xhat3 cC3), elementwise over [B,16,T2p]
__device__ __forceinline__ void cv_bwd_da4_kernel_body(const float* __restrict__ a4, const float* __restrict__ dp3,
                                                         const EegCoef* __restrict__ co, float* __restrict__ da4,
                                                         int64_t n, int T2p, int T3, int P2, float dp, uint64_t seed,
    unsigned zgx, unsigned zgy, unsigned zbz, unsigned zgz) {   // this zone's own gridDim.x/.y, blockIdx.z, gridDim.z
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  const int64_t row = e / T2p;
  const int w = (int)(e - row * T2p), h = (int)(row & (kF2 - 1));
  const float av = a4[e];
  const float dy = cv_dy3(dp3, co, row, h, w, av, T3, P2, dp, seed);
  const float xh = (av - co->mu3[h]) / co->sig3[h];
  da4[e] = co->cA3[h] * (dy - co->cB3[h] - xh * co->cC3[h]);
}
ISD_ZONE_FN(cv_bwd_da4_kernel, 256)
__global__ __launch_bounds__(256) void cv_bwd_da4_kernel(const float* __restrict__ a4, const float* __restrict__ dp3,
                                                         const EegCoef* __restrict__ co, float* __restrict__ da4,
                                                         int64_t n, int T2p, int T3, int P2, float dp, uint64_t seed) {
  cv_bwd_da4_kernel_body(a4, dp3, co, da4, n, T2p, T3, P2, dp, seed,
      gridDim.x, gridDim.y, blockIdx.z, gridDim.z);
}
ISD_ZONE_REGISTER(cv_bwd_da4_kernel)

// dp2[b,g,v] = sum_{h,k} W3[h,g,k] da4[b,h,v-k+8].  One thread per (b,v), 16 outputs in registers.
__device__ __forceinline__ void cv_bwd_dp2_kernel_body(const float* __restrict__ da4, const float* __restrict__ Wb,
                                                         float* __restrict__ dp2, int64_t B, int T2, int T2p,
    unsigned zgx, unsigned zgy, unsigned zbz, unsigned zgz) {   // this zone's own gridDim.x/.y, blockIdx.z, gridDim.z
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= B * T2) return;
  const int64_t b = e / T2;
  const int v = (int)(e - b * T2);
  float o[kF2];
#pragma unroll
  for (int g = 0; g < kF2; ++g) o[g] = 0.f;
  for (int h = 0; h < kF2; ++h) {
    const float* dr = da4 + (b * kF2 + h) * T2p;
#pragma unroll
    for (int k = 0; k < kK2; ++k) {
      const int w = v - k + kP2;
      const float dv = (w >= 0 && w < T2p) ? dr[w] : 0.f;
      const float* wv = Wb + (h * kK2 + k) * kF2;
#pragma unroll
      for (int g = 0; g < kF2; ++g) o[g] = fmaf(wv[g], dv, o[g]);
    }
  }
#pragma unroll
  for (int g = 0; g < kF2; ++g) dp2[(b * kF2 + g) * T2 + v] = o[g];
}
ISD_ZONE_FN(cv_bwd_dp2_kernel, 256)
__global__ __launch_bounds__(256) void cv_bwd_dp2_kernel(const float* __restrict__ da4, const float* __restrict__ Wb,
                                                         float* __restrict__ dp2, int64_t B, int T2, int T2p) {
  cv_bwd_dp2_kernel_body(da4, Wb, dp2, B, T2, T2p,
      gridDim.x, gridDim.y, blockIdx.z, gridDim.z);
}
ISD_ZONE_REGISTER(cv_bwd_dp2_kernel)

// dW3[h,g,k] = sum_{b,w} da4[b,h,w] p2pad[b,g,w+k] on the matrix cores: M = h, N = k (tap), K = w, one
// 16x16 accumulator tile per input channel g.  Persistent waves over trials, partial slabs [h][g][k].
__device__ __forceinline__ void cv_bwd_w3_kernel_body(const float* __restrict__ da4, const float* __restrict__ p2,
                                                       float* __restrict__ part, int64_t B, int T2, int T2p,
    unsigned zgx, unsigned zgy, unsigned zbz, unsigned zgz) {   // this zone's own gridDim.x/.y, blockIdx.z, gridDim.z
  const int lane = threadIdx.x, q = lane >> 4, jl = lane & 15;
  f32x4 acc[kF2];
#pragma unroll
  for (int g = 0; g < kF2; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int64_t b = blockIdx.x; b < B; b += zgx) {
    const float* dr = da4 + (b * kF2 + jl) * T2p;
    const float* pb = p2 + b * kF2 * T2;
    for (int w0 = 0; w0 < T2p; w0 += 4) {
      const int w = w0 + q;
      const float af = w < T2p ? dr[w] : 0.f;
      const int v = w + jl - kP2;
      const bool ok = w < T2p && v >= 0 && v < T2;
#pragma unroll
      for (int g = 0; g < kF2; ++g) {
        const float bf = ok ? pb[g * T2 + v] : 0.f;
        acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, acc[g], 0, 0, 0);
      }
    }
  }
  float* slab = part + (int64_t)blockIdx.x * (kF2 * kF2 * kK2);
#pragma unroll
  for (int g = 0; g < kF2; ++g)
#pragma unroll
    for (int r = 0; r < 4; ++r) slab[(4 * q + r) * (kF2 * kK2) + g * kK2 + jl] = acc[g][r];
}
ISD_ZONE_FN(cv_bwd_w3_kernel, 64)
__global__ __launch_bounds__(64) void cv_bwd_w3_kernel(const float* __restrict__ da4, const float* __restrict__ p2,
                                                       float* __restrict__ part, int64_t B, int T2, int T2p) {
  cv_bwd_w3_kernel_body(da4, p2, part, B, T2, T2p,
      gridDim.x, gridDim.y, blockIdx.z, gridDim.z);
}
ISD_ZONE_REGISTER(cv_bwd_w3_kernel)

__device__ __forceinline__ void cv_w3_reduce_kernel_body(const float* __restrict__ part, int n_slabs,
                                                           float* __restrict__ dW3,
    unsigned zgx, unsigned zgy, unsigned zbz, unsigned zgz) {   // this zone's own gridDim.x/.y, blockIdx.z, gridDim.z
  // block = 64 elements x 4 slab groups (contiguous quarters of the slabs, combined in LDS in a fixed order): one
  // thread per element walking up to 1024 slabs was a chain of dependent loads
  __shared__ float red[4][64];
  constexpr int N = kF2 * kF2 * kK2;
  const int l = threadIdx.x & 63, gq = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + l;                      // N is a multiple of 64
  const int per = (n_slabs + 3) / 4;
  const int k_lo = gq * per, k_hi = k_lo + per < n_slabs ? k_lo + per : n_slabs;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int k = k_lo;
  for (; k + 3 < k_hi; k += 4) {
    s0 += part[(int64_t)k * N + e];
    s1 += part[(int64_t)(k + 1) * N + e];
    s2 += part[(int64_t)(k + 2) * N + e];
    s3 += part[(int64_t)(k + 3) * N + e];
  }
  for (; k < k_hi; ++k) s0 += part[(int64_t)k * N + e];
  red[gq][l] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (gq == 0) dW3[e] = (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]);
}
ISD_ZONE_FN(cv_w3_reduce_kernel, 256)
__global__ __launch_bounds__(256) void cv_w3_reduce_kernel(const float* __restrict__ part, int n_slabs,
                                                           float* __restrict__ dW3) {
  cv_w3_reduce_kernel_body(part, n_slabs, dW3,
      gridDim.x, gridDim.y, blockIdx.z, gridDim.z);
}
ISD_ZONE_REGISTER(cv_w3_reduce_kernel)

}  // namespace isd

using namespace isd;

struct isd_eegnet_plan {
  int C, F, K, T, Tp, T2, T2p, T3;
  int cv;          // 0: EEGNet_Encoder, 1: CVBlock
  int P1, P2;      // AvgPool widths of the two stages
  EegOff off;
  const unsigned long long* seed_dev;   // optional device-resident dropout step counter (isd_eegnet_plan_set_seed_counter)
};

static inline int64_t al64(int64_t v) { return (v + 63) / 64 * 64; }

static int eeg_plan_create(isd_eegnet_plan** out, int in_channels, int feature_dim, int kernel_length, int T, int cv);

extern "C" int isd_eegnet_plan_create(isd_eegnet_plan** out, int in_channels, int feature_dim, int kernel_length,
                                      int T) {
  return eeg_plan_create(out, in_channels, feature_dim, kernel_length, T, 0);
}
extern "C" int isd_cvblock_plan_create(isd_eegnet_plan** out, int in_channels, int dim_token, int T) {
  return eeg_plan_create(out, in_channels, dim_token, kMaxK, T, 1);
}
extern "C" int64_t isd_cvblock_flat_dim(const isd_eegnet_plan* p) {
  return (p && p->cv) ? (int64_t)kF2 * p->T3 : ISD_ERR_INVALID;
}

static int eeg_plan_create(isd_eegnet_plan** out, int in_channels, int feature_dim, int kernel_length, int T, int cv) {
  ISD_CHECK_ARG(out, "isd_eegnet_plan_create: null argument");
  ISD_CHECK_ARG(in_channels >= 1 && in_channels <= 16384, "isd_eegnet_plan_create: in_channels=%d", in_channels);
  ISD_CHECK_ARG(feature_dim >= 1 && feature_dim <= 1024, "isd_eegnet_plan_create: feature_dim=%d not in [1,1024]", feature_dim);
  ISD_CHECK_ARG(kernel_length >= 2 && kernel_length <= kMaxK && (kernel_length & 1) == 0,
                "isd_eegnet_plan_create: kernel_length=%d must be even and <= %d", kernel_length, kMaxK);
  isd_eegnet_plan* p = new isd_eegnet_plan();
  p->C = in_channels; p->F = feature_dim; p->K = kernel_length; p->T = T;
  p->cv = cv;
  p->P1 = cv ? 8 : 4;
  p->P2 = cv ? 2 : 8;
  p->Tp = T + 2 * (kernel_length / 2) - kernel_length + 1;
  p->T2 = p->Tp / p->P1;
  p->T2p = p->T2 + 2 * kP2 - kK2 + 1;
  p->T3 = p->T2p / p->P2;
  if (T < 1 || p->T2 < 1 || p->T3 < 1) {
    set_error("isd_eegnet_plan_create: T=%d is too short for the two pooling stages", T);
    delete p;
    return ISD_ERR_INVALID;
  }
  int o = 0;
  EegOff& f = p->off;
  f.Wt = o; o += kF1 * kernel_length;
  f.g1 = o; o += kF1;
  f.b1 = o; o += kF1;
  f.Ws = o; o += kF2 * in_channels;
  f.g2 = o; o += kF2;
  f.b2 = o; o += kF2;
  f.Wd = o; o += cv ? kF2 * kF2 * kK2 : kF2 * kK2;      // CVBlock: the full conv3 weight [16,16,16] sits here
  f.Wp = o; o += cv ? 0 : kF2 * kF2;
  f.g3 = o; o += kF2;
  f.b3 = o; o += kF2;
  f.Wl = o; o += feature_dim * kF2 * (cv ? p->T3 : 1);
  f.bl = o; o += feature_dim;
  f.total = o;
  *out = p;
  return ISD_OK;
}

extern "C" int isd_eegnet_plan_set_seed_counter(isd_eegnet_plan* p, const uint64_t* seed_dev) {
  ISD_CHECK_ARG(p, "isd_eegnet_plan_set_seed_counter: null plan");
  p->seed_dev = (const unsigned long long*)seed_dev;
  return ISD_OK;
}

extern "C" int isd_eegnet_plan_destroy(isd_eegnet_plan* p) {
  delete p;
  return ISD_OK;
}
extern "C" int64_t isd_eegnet_param_count(const isd_eegnet_plan* p) { return p ? p->off.total : ISD_ERR_INVALID; }
extern "C" int64_t isd_eegnet_buffer_count(const isd_eegnet_plan* p) { return p ? kBufTotal : ISD_ERR_INVALID; }

namespace {
struct EegWs {          // float offsets into the workspace
  int64_t stats, coef, z, u, p2, a4, pooled, dpooled, a3, da4, da3, dy2, v, part, lin, w3f, w3b, total;
  int n_slabs;
};
EegWs eeg_layout(const isd_eegnet_plan* p, int64_t B) {
  EegWs w;
  int64_t o = 0;
  w.stats = o; o += al64((int64_t)(sizeof(EegStats) + 3) / 4);
  w.coef = o; o += al64((int64_t)(sizeof(EegCoef) + 3) / 4);
  w.z = o; o += al64(B * kF2 * p->T);
  w.u = o; o += al64(B * kF2 * p->Tp);
  w.p2 = o; o += al64(B * kF2 * p->T2);
  w.a4 = o; o += al64(B * kF2 * p->T2p);
  const int64_t npool = B * kF2 * (p->cv ? p->T3 : 1);
  w.pooled = o; o += al64(npool);
  w.dpooled = o; o += al64(npool);
  w.a3 = o; o += al64(B * kF2 * p->T2p);
  w.da4 = o; o += al64(B * kF2 * p->T2p);
  w.da3 = o; o += al64(B * kF2 * p->T2p);
  w.dy2 = o; o += al64(B * kF2 * p->Tp);
  w.v = o; o += al64(B * kF2 * p->T);
  w.n_slabs = 1024;
  const int64_t slab = p->cv && kF2 * kK2 > p->C ? kF2 * kF2 * kK2 : kF2 * p->C;
  w.part = o; o += al64((int64_t)w.n_slabs * slab);
  w.lin = o; o += al64(isd_linear_workspace_bytes(B, kF2 * (p->cv ? p->T3 : 1), p->F) / 4 + 64);
  w.w3f = o; o += p->cv ? kF2 * kF2 * kK2 : 0;
  w.w3b = o; o += p->cv ? kF2 * kF2 * kK2 : 0;
  w.total = o;
  return w;
}
}  // namespace

extern "C" int64_t isd_eegnet_workspace_bytes(const isd_eegnet_plan* p, int64_t B) {
  if (!p || B < 0) return ISD_ERR_INVALID;
  return eeg_layout(p, B).total * 4;
}

// workgroups per row for the kernels that end in one fp64 atomic per workgroup and statistic: enough of them to
// fill the chip (~4 k workgroups), no more
// y-extent of the launches whose workgroups also walk rows: a multiple of 16 (the rows of a workgroup share the filter),
// about 1 k workgroups in all (64 atomics per accumulator address)
static unsigned row_groups(int64_t rows16, unsigned per_row) {
  int64_t want = 1024 / (per_row ? per_row : 1) / 16 * 16;
  if (want < 16) want = 16;
  return (unsigned)(rows16 < want ? rows16 : want);
}
static unsigned row_blocks(int row_len, int64_t n_rows) {
  const int64_t full = cdiv(row_len, 256);
  int64_t want = cdiv(4096, n_rows > 0 ? n_rows : 1);
  if (want < 1) want = 1;
  return (unsigned)(want < full ? want : full);
}

// ---------------------------------------------------------------------------------------
// Forward / backward in stages.  Between two stages a block of fp64 batch sums sits complete in the workspace
// (isd_eegnet_sync_block): under data parallelism the caller all-reduces it (SUM) there -- synchronised BatchNorm,
// SURVEY.md 8(e) -- and passes the world size so that counts are global and the gradients that are assembled from
// global sums are pre-divided (the gradient all-reduce that follows adds the ranks' copies).  world = 1 and the stages
// run back to back are the single-device call.
//   forward  0: x statistics            -> sync block F0 (autocorrelation sums of x: BN1)
//            1: BN1 fold, projection, temporal conv -> F1 (sum u, sum u^2: BN2)
//            2: BN2, ELU, pool, stage-2 conv        -> F2 (sum a4, sum a4^2: BN3)
//            3: BN3, ELU, pool, projector
//   backward 0: projector, BN3 sums      -> B0 (sum dy3, sum dy3 xhat3)
//            1: BN3 backward, stage-2 conv gradients, pool -> B1 (sum dy2, sum dy2 xhat2)
//            2: BN2 backward, correlations, dWs            -> B2 (sum da2, sum da2 u, sum da2 zpad)
//            3: stage-1 gradients (dWt, BN1 affine), separable weights
// ---------------------------------------------------------------------------------------
static int eeg_forward_stage(const isd_eegnet_plan* p, int stage, const float* x, const float* params, float* buffers,
                             float* out, float* ws, int64_t B, int training, float momentum, float eps, float dp,
                             uint64_t seed, int world, hipStream_t st, void* stream) {
  const EegWs w = eeg_layout(p, B);
  EegStats* S = (EegStats*)(ws + w.stats);
  EegCoef* Cf = (EegCoef*)(ws + w.coef);
  const int C = p->C, K = p->K, T = p->T, Tp = p->Tp, T2 = p->T2, T2p = p->T2p, T3 = p->T3;
  const int64_t rows = B * C;
  const double wd = (double)world;
  // `training` bit 0: BatchNorm uses batch statistics; bit 1: eval-mode statistics, but keep what a backward pass
  // needs (input attributions differentiate the eval-mode network)
  const int keep = training != 0;
  training &= 1;
  if (stage == 0) {
    ISD_HIP_TRY(zone_clear(S, sizeof(EegStats), st));
    if (training) {
      if (T <= 79) {
        const int MT = (T + 16) / 16;                              // 16 MT >= T + 1: room for the column of ones
        const int64_t want_g = cdiv(cdiv(rows, 4), (int64_t)kStatWaves * 8);   // >= 8 row groups per wave (see below)
        const int grid_g = want_g < 1024 ? (int)want_g : 1024;
        switch (MT) {
          case 1: ISD_ZLAUNCH(eeg_stats_gram_kernel<1>, dim3(grid_g), dim3(64 * kStatWaves), 0, st, x, S, rows, T); break;
          case 2: ISD_ZLAUNCH(eeg_stats_gram_kernel<2>, dim3(grid_g), dim3(64 * kStatWaves), 0, st, x, S, rows, T); break;
          case 3: ISD_ZLAUNCH(eeg_stats_gram_kernel<3>, dim3(grid_g), dim3(64 * kStatWaves), 0, st, x, S, rows, T); break;
          case 4: ISD_ZLAUNCH(eeg_stats_gram_kernel<4>, dim3(grid_g), dim3(64 * kStatWaves), 0, st, x, S, rows, T); break;
          default: ISD_ZLAUNCH(eeg_stats_gram_kernel<5>, dim3(grid_g), dim3(64 * kStatWaves), 0, st, x, S, rows, T); break;
        }
        ISD_ZLAUNCH(eeg_stats_gram_derive_kernel, dim3(1), dim3(64), 0, st, S, T);
      } else {
        // every workgroup ends with 1280 (bulk) / 3072 (edge) fp64 atomics on the plan's accumulators: give a wave at
        // least 16 rows (8 groups of 4) before that, or the atomics are the kernel (a workgroup per 4 rows of the FAST
        // heads' zone shape sent 6.5 M of them per launch: 0.2 ms for 20 MB of input)
        const int64_t want = cdiv(rows, (int64_t)kStatWaves * 16);
        const int grid = want < 2048 ? (int)want : 2048;             // bulk: 32 waves per CU
        ISD_ZLAUNCH(eeg_stats_kernel, dim3(grid), dim3(64 * kStatWaves), 0, st, x, S, rows, T);
        const int64_t want_e = cdiv(cdiv(rows, 4), (int64_t)kStatWaves * 8);
        const int grid_e = want_e < 512 ? (int)want_e : 512;         // edge: 4 rows per MFMA step, head and tail blocks
        ISD_ZLAUNCH(eeg_stats_edge_kernel, dim3(grid_e, 2), dim3(64 * kStatWaves), 0, st, x, S, rows, T);
        ISD_ZLAUNCH(eeg_stats_derive_kernel, dim3(1), dim3(64), 0, st, S);
      }
    }
    ISD_LAUNCH_CHECK();
    return ISD_OK;
  }
  if (stage == 1) {
    ISD_ZLAUNCH(eeg_finalize1_kernel, dim3(1), dim3(256), 0, st, params, buffers, S, Cf, p->off, C, K, T,
                       rows * world, training, momentum, eps, p->seed_dev);
    if (C >= 128)                                               // wide inputs: whole rows per workgroup
      ISD_ZLAUNCH(eeg_spatial_rows_kernel<5>, dim3((unsigned)cdiv(cdiv(T, 16), 5), (unsigned)B), dim3(256), 0, st,
                         x, params + p->off.Ws, ws + w.z, C, T);
    else
      ISD_ZLAUNCH(eeg_spatial_kernel, dim3((unsigned)cdiv(T, 16), (unsigned)B), dim3(256), 0, st, x,
                         params + p->off.Ws, ws + w.z, C, T, (int)cdiv(T, 16));
    {
      const unsigned gx = row_blocks(Tp, B * kF2);
#define ISD_EEG_K(KERNEL, ...)                          \
  do {                                                 \
    if (K == 64) ISD_ZLAUNCH(KERNEL<64>, __VA_ARGS__);      \
    else if (K == 32) ISD_ZLAUNCH(KERNEL<32>, __VA_ARGS__); \
    else if (K == 16) ISD_ZLAUNCH(KERNEL<16>, __VA_ARGS__); \
    else ISD_ZLAUNCH(KERNEL<0>, __VA_ARGS__);               \
  } while (0)
      ISD_EEG_K(eeg_tconv_kernel, dim3(gx, row_groups(B * kF2, gx)), dim3(256), 0, st, ws + w.z, params + p->off.Wt,
                ws + w.u, S, K, T, Tp, training, (int)(B * kF2));
    }
    ISD_LAUNCH_CHECK();
    return ISD_OK;
  }
  if (stage == 2) {
    ISD_ZLAUNCH(eeg_finalize2_kernel, dim3(1), dim3(64), 0, st, params, buffers, S, Cf, p->off,
                       (double)B * (double)Tp * wd, training, momentum, eps);
    ISD_ZLAUNCH(eeg_pool2_kernel, dim3((unsigned)cdiv(T2, 256), (unsigned)(B * kF2)), dim3(256), 0, st, ws + w.u,
                       Cf, ws + w.p2, Tp, T2, p->P1, dp, seed);
    if (p->cv) {
      ISD_ZLAUNCH(cv_prep_kernel, dim3(kF2 * kF2 * kK2 / 256), dim3(256), 0, st, params + p->off.Wd, ws + w.w3f,
                         ws + w.w3b);
      ISD_ZLAUNCH(cv_conv3_kernel, dim3((unsigned)cdiv(B * T2p, 256)), dim3(256), 0, st, ws + w.p2, ws + w.w3f,
                         ws + w.a4, S, B, T2, T2p, training);
    } else {
      const unsigned gxs = (unsigned)cdiv(T2p, 256);
      const int64_t gys = 2048 / gxs > 0 ? 2048 / gxs : 1;
      ISD_ZLAUNCH(eeg_sep_kernel, dim3(gxs, (unsigned)(B < gys ? B : gys)), dim3(256), 0, st, ws + w.p2,
                         params + p->off.Wd, params + p->off.Wp, keep ? ws + w.a3 : nullptr, ws + w.a4, S, T2, T2p,
                         training, (int)B);
    }
    ISD_LAUNCH_CHECK();
    return ISD_OK;
  }
  ISD_ZLAUNCH(eeg_finalize3_kernel, dim3(1), dim3(64), 0, st, params, buffers, S, Cf, p->off,
                     (double)B * (double)T2p * wd, training, momentum, eps);
  if (p->cv) {
    const int64_t n3 = B * kF2 * T3;
    ISD_ZLAUNCH(cv_pool3_kernel, dim3((unsigned)cdiv(n3, 256)), dim3(256), 0, st, ws + w.a4, Cf, ws + w.pooled, n3,
                       T2p, T3, p->P2, dp, seed);
    ISD_LAUNCH_CHECK();
    return isd_linear_forward(ws + w.pooled, params + p->off.Wl, params + p->off.bl, out, nullptr, B, kF2 * T3, p->F, 0,
                              stream);
  }
  ISD_ZLAUNCH(eeg_pool3_kernel, dim3((unsigned)cdiv(B * kF2, 4)), dim3(256), 0, st, ws + w.a4, Cf,
                     ws + w.pooled, B * kF2, T2p, T3, dp, seed);
  ISD_LAUNCH_CHECK();
  return isd_linear_forward(ws + w.pooled, params + p->off.Wl, params + p->off.bl, out, nullptr, B, kF2, p->F, 0,
                            stream);
}

static int eeg_forward_check(const isd_eegnet_plan* p, const float* x, const float* params, float* buffers, float* out,
                             void* workspace, int64_t B, float dropout_p, int world) {
  ISD_CHECK_ARG(p, "isd_eegnet_forward: null plan");
  ISD_CHECK_ARG(B >= 0 && B <= 0x7fffffff / kF2, "isd_eegnet_forward: B=%lld", (long long)B);
  ISD_CHECK_ARG(B == 0 || (x && params && buffers && out && workspace), "isd_eegnet_forward: null argument");
  ISD_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, "isd_eegnet_forward: dropout_p=%g not in [0,1)", dropout_p);
  ISD_CHECK_ARG(world >= 1 && world <= 65536, "isd_eegnet_forward: world=%d", world);
  return ISD_OK;
}

extern "C" int isd_eegnet_forward(const isd_eegnet_plan* p, const float* x, const float* params, float* buffers,
                                  float* out, void* workspace, int64_t B, int training, float momentum, float eps,
                                  float dropout_p, uint64_t seed, void* stream) {
  int rc = eeg_forward_check(p, x, params, buffers, out, workspace, B, dropout_p, 1);
  if (rc || B == 0) return rc;
  for (int stage = 0; stage < 4 && rc == ISD_OK; ++stage)
    rc = eeg_forward_stage(p, stage, x, params, buffers, out, (float*)workspace, B, training, momentum, eps,
                           (training & 1) ? dropout_p : 0.f, seed, 1, (hipStream_t)stream, stream);
  return rc;
}

extern "C" int isd_eegnet_forward_stage(const isd_eegnet_plan* p, int stage, const float* x, const float* params,
                                        float* buffers, float* out, void* workspace, int64_t B, int training,
                                        float momentum, float eps, float dropout_p, uint64_t seed, int world,
                                        void* stream) {
  int rc = eeg_forward_check(p, x, params, buffers, out, workspace, B, dropout_p, world);
  if (rc || B == 0) return rc;
  ISD_CHECK_ARG(stage >= 0 && stage < 4, "isd_eegnet_forward_stage: stage=%d not in [0,4)", stage);
  return eeg_forward_stage(p, stage, x, params, buffers, out, (float*)workspace, B, training, momentum, eps,
                           (training & 1) ? dropout_p : 0.f, seed, world, (hipStream_t)stream, stream);
}

static int eeg_backward_stage(const isd_eegnet_plan* p, int stage, const float* x, const float* params,
                              const float* dout, float* dparams, float* ws, int64_t B, float dropout_p, uint64_t seed,
                              int world, hipStream_t st, void* stream, int bn_train = 1) {
  const EegWs w = eeg_layout(p, B);
  EegStats* S = (EegStats*)(ws + w.stats);
  EegCoef* Cf = (EegCoef*)(ws + w.coef);
  const int C = p->C, K = p->K, T = p->T, Tp = p->Tp, T2 = p->T2, T2p = p->T2p, T3 = p->T3;
  const int64_t rows16 = B * kF2;
  const double wd = (double)world, gs = 1.0 / wd;
  if (stage == 0) {
    ISD_HIP_TRY(zone_clear((char*)S + offsetof(EegStats, dy3s), sizeof(EegStats) - offsetof(EegStats, dy3s), st));
    int rc = isd_linear_backward(ws + w.pooled, params + p->off.Wl, dout, nullptr, ws + w.dpooled, dparams + p->off.Wl,
                                 dparams + p->off.bl, ws + w.lin, B, kF2 * (p->cv ? T3 : 1), p->F, 0, stream);
    if (rc) return rc;
    if (p->cv)
      ISD_ZLAUNCH(cv_bwd3_sums_kernel, dim3((unsigned)cdiv(rows16, 4)), dim3(256), 0, st, ws + w.a4, ws + w.dpooled,
                         Cf, S, rows16, T2p, T3, p->P2, dropout_p, seed);
    else
      ISD_ZLAUNCH(eeg_bwd3_sums_kernel, dim3((unsigned)(rows16 / 4 < 256 ? rows16 / 4 : 256)), dim3(256), 0, st,
                         ws + w.a4, ws + w.dpooled, Cf, S, rows16, T2p, T3, dropout_p, seed);
    ISD_LAUNCH_CHECK();
    return ISD_OK;
  }
  if (stage == 1) {
    ISD_ZLAUNCH(eeg_bwd_bn_coef_kernel, dim3(1), dim3(64), 0, st, params, dparams, S, Cf, p->off,
                       (double)B * (double)T2p * wd, 3, gs, bn_train);
    if (p->cv) {
      ISD_ZLAUNCH(cv_bwd_da4_kernel, dim3((unsigned)cdiv(rows16 * T2p, 256)), dim3(256), 0, st, ws + w.a4,
                         ws + w.dpooled, Cf, ws + w.da4, rows16 * T2p, T2p, T3, p->P2, dropout_p, seed);
      const int slabs3 = B < w.n_slabs ? (int)B : w.n_slabs;
      ISD_ZLAUNCH(cv_bwd_w3_kernel, dim3(slabs3), dim3(64), 0, st, ws + w.da4, ws + w.p2, ws + w.part, B, T2, T2p);
      ISD_ZLAUNCH(cv_w3_reduce_kernel, dim3(kF2 * kF2 * kK2 / 64), dim3(256), 0, st, ws + w.part, slabs3,
                         dparams + p->off.Wd);
      ISD_ZLAUNCH(cv_bwd_dp2_kernel, dim3((unsigned)cdiv(B * T2, 256)), dim3(256), 0, st, ws + w.da4, ws + w.w3b,
                         ws + w.da3, B, T2, T2p);
      ISD_ZLAUNCH(eeg_bwd_pool2_kernel, dim3(row_blocks(Tp, rows16), row_groups(rows16, row_blocks(Tp, rows16))),
                         dim3(256), 0, st, ws + w.da3, (const float*)nullptr, ws + w.u, Cf, ws + w.dy2, S, Tp, T2, T2p,
                         p->P1, dropout_p, seed, (int)rows16);
    } else {
      ISD_ZLAUNCH(eeg_bwd_sep_kernel, dim3((unsigned)cdiv(T2p, 256), (unsigned)B), dim3(256), 0, st, ws + w.a4,
                         ws + w.dpooled, params + p->off.Wp, Cf, ws + w.da4, ws + w.da3, T2p, T3, dropout_p, seed);
      ISD_ZLAUNCH(eeg_bwd_sepw_kernel, dim3(kF2 * kF2 + kF2 * kK2), dim3(256), 0, st, ws + w.da4, ws + w.a3,
                         ws + w.da3, ws + w.p2, S, (int)B, T2, T2p);
      ISD_ZLAUNCH(eeg_bwd_pool2_kernel, dim3(row_blocks(Tp, rows16), row_groups(rows16, row_blocks(Tp, rows16))),
                         dim3(256), 0, st, ws + w.da3, params + p->off.Wd, ws + w.u, Cf, ws + w.dy2, S, Tp, T2, T2p, p->P1,
                         dropout_p, seed, (int)rows16);
    }
    ISD_LAUNCH_CHECK();
    return ISD_OK;
  }
  if (stage == 2) {
    ISD_ZLAUNCH(eeg_bwd_bn_coef_kernel, dim3(1), dim3(64), 0, st, params, dparams, S, Cf, p->off,
                       (double)B * (double)Tp * wd, 2, gs, bn_train);
    ISD_ZLAUNCH(eeg_bwd_bn2_kernel, dim3(row_blocks(Tp, rows16), row_groups(rows16, row_blocks(Tp, rows16))),
                       dim3(256), 0, st, ws + w.dy2, ws + w.u, Cf, S, Tp, (int)rows16);
    {
      const size_t lds = sizeof(float) * (size_t)((kCorrSeg + K + 64) + (kCorrSeg + 2 * K + 64) + kMaxK);
      const unsigned segs = (unsigned)cdiv(Tp, kCorrSeg);
      ISD_EEG_K(eeg_bwd_corr_kernel, dim3(row_groups(rows16, segs), segs), dim3(64), lds, st, ws + w.dy2, ws + w.z,
                params + p->off.Wt, ws + w.v, S, K, T, Tp, (int)rows16);
    }
    const int64_t n_chunks = B * ((T + kDwsPasses * 16 - 1) / (kDwsPasses * 16));
    const int slabs = n_chunks < w.n_slabs ? (int)n_chunks : w.n_slabs;
    ISD_ZLAUNCH(eeg_bwd_dws_kernel, dim3(slabs, (unsigned)cdiv(C, 256)), dim3(64), 0, st, ws + w.v, x, ws + w.part,
                       (int)B, C, T);
    // dWs takes the LOCAL sum of da2 (it is linear in the local batch): it runs in front of the all-reduce of B2
    ISD_ZLAUNCH(eeg_bwd_dws_reduce_kernel, dim3((unsigned)cdiv((int64_t)kF2 * C, 64)), dim3(1024), 0, st,
                       ws + w.part, slabs, S, Cf, dparams + p->off.Ws, C);
    ISD_LAUNCH_CHECK();
    return ISD_OK;
  }
  // the stage-1 gradients are assembled from GLOBAL sums (BatchNorm's backward couples the whole batch): every rank
  // computes the same global gradient, pre-divided by the world size; the separable weights are local sums
  ISD_ZLAUNCH(eeg_bwd_final_kernel, dim3(1), dim3(256), 0, st, params, dparams, S, Cf, p->off, C, K,
                     (double)(B * C) * (double)Tp * wd, !p->cv, gs, bn_train);
  ISD_LAUNCH_CHECK();
  return ISD_OK;
}

static int eeg_backward_check(const isd_eegnet_plan* p, const float* x, const float* params, const float* dout,
                              float* dparams, void* workspace, int64_t B, int world) {
  ISD_CHECK_ARG(p, "isd_eegnet_backward: null plan");
  ISD_CHECK_ARG(B >= 1, "isd_eegnet_backward: B=%lld", (long long)B);
  ISD_CHECK_ARG(x && params && dout && dparams && workspace, "isd_eegnet_backward: null argument");
  ISD_CHECK_ARG(world >= 1 && world <= 65536, "isd_eegnet_backward: world=%d", world);
  return ISD_OK;
}

extern "C" int isd_eegnet_backward(const isd_eegnet_plan* p, const float* x, const float* params, const float* dout,
                                   float* dparams, void* workspace, int64_t B, float dropout_p, uint64_t seed,
                                   void* stream) {
  int rc = eeg_backward_check(p, x, params, dout, dparams, workspace, B, 1);
  for (int stage = 0; stage < 4 && rc == ISD_OK; ++stage)
    rc = eeg_backward_stage(p, stage, x, params, dout, dparams, (float*)workspace, B, dropout_p, seed, 1,
                            (hipStream_t)stream, stream);
  return rc;
}

// Parameter gradients AND the gradient w.r.t. the input trials.  `training` as in the forward that preceded it:
// 1 = batch statistics (BatchNorm's mean / variance paths contribute to dx), 2 = eval-mode statistics with the
// activations kept (isd_eegnet_forward(..., training = 2, ...): what attribution methods differentiate).
extern "C" int isd_eegnet_backward_x(const isd_eegnet_plan* p, const float* x, const float* params, const float* dout,
                                     float* dparams, float* dx, void* workspace, int64_t B, int training,
                                     float dropout_p, uint64_t seed, void* stream) {
  int rc = eeg_backward_check(p, x, params, dout, dparams, workspace, B, 1);
  if (rc) return rc;
  ISD_CHECK_ARG(dx, "isd_eegnet_backward_x: null dx");
  ISD_CHECK_ARG(training == 1 || training == 2, "isd_eegnet_backward_x: training=%d (1: batch statistics, 2: eval + kept)",
                training);
  const int bn_train = training & 1;
  hipStream_t st = (hipStream_t)stream;
  for (int stage = 0; stage < 4 && rc == ISD_OK; ++stage)
    rc = eeg_backward_stage(p, stage, x, params, dout, dparams, (float*)workspace, B, bn_train ? dropout_p : 0.f, seed, 1,
                            st, stream, bn_train);
  if (rc) return rc;
  float* ws = (float*)workspace;
  const EegWs w = eeg_layout(p, B);
  ISD_CHECK_ARG(B * p->C <= 0x7fffffffLL, "isd_eegnet_backward_x: too many rows");
  const int K = p->K;
  ISD_EEG_K(eeg_bwd_dx_kernel, dim3((unsigned)(B * p->C), (unsigned)cdiv(p->T, 256)), dim3(256), 0, st, x, ws + w.v, params,
            dx, (const EegStats*)(ws + w.stats), (const EegCoef*)(ws + w.coef), p->off, p->C, p->K, p->T, p->Tp,
            (double)(B * p->C) * (double)p->Tp, bn_train);
  ISD_LAUNCH_CHECK();
  return ISD_OK;
}

extern "C" int isd_eegnet_backward_stage(const isd_eegnet_plan* p, int stage, const float* x, const float* params,
                                         const float* dout, float* dparams, void* workspace, int64_t B,
                                         float dropout_p, uint64_t seed, int world, void* stream) {
  int rc = eeg_backward_check(p, x, params, dout, dparams, workspace, B, world);
  if (rc) return rc;
  ISD_CHECK_ARG(stage >= 0 && stage < 4, "isd_eegnet_backward_stage: stage=%d not in [0,4)", stage);
  return eeg_backward_stage(p, stage, x, params, dout, dparams, (float*)workspace, B, dropout_p, seed, world,
                            (hipStream_t)stream, stream);
}

// The fp64 sums that are complete after `stage` (0..2) of the forward (backward = 0) or backward (backward = 1) pass:
// byte offset from the workspace base and number of doubles.  Summed over the ranks between stage and stage + 1 they
// make the BatchNorm statistics those of the global batch.
extern "C" int isd_eegnet_sync_block(const isd_eegnet_plan* p, int64_t B, int backward, int stage, int64_t* byte_offset,
                                     int64_t* n_doubles) {
  ISD_CHECK_ARG(p && byte_offset && n_doubles, "isd_eegnet_sync_block: null argument");
  ISD_CHECK_ARG(stage >= 0 && stage < 3 && B >= 0, "isd_eegnet_sync_block: stage=%d not in [0,3)", stage);
  const EegWs w = eeg_layout(p, B);
  size_t lo, hi;
  if (!backward) {
    // A, H, Tl, S, Hs, Ts: fp64 sums only -- the block ends in front of Sx, the first integer accumulator (its words
    // are the raw material S was derived from before the exchange; summed as doubles they would be garbage)
    if (stage == 0) { lo = offsetof(EegStats, A); hi = offsetof(EegStats, Sx); }
    else if (stage == 1) { lo = offsetof(EegStats, u1); hi = offsetof(EegStats, a1); }
    else { lo = offsetof(EegStats, a1); hi = offsetof(EegStats, dy3s); }
  } else {
    if (stage == 0) { lo = offsetof(EegStats, dy3s); hi = offsetof(EegStats, dy2s); }
    else if (stage == 1) { lo = offsetof(EegStats, dy2s); hi = offsetof(EegStats, Sd); }
    else { lo = offsetof(EegStats, Sd); hi = offsetof(EegStats, dWp); }                         // Sd, Su, T1
  }
  *byte_offset = w.stats * 4 + (int64_t)lo;
  *n_doubles = (int64_t)((hi - lo) / sizeof(double));
  return ISD_OK;
}

// 0: the block holds fp64 sums; 1: 64-bit integer words of exact accumulators (csrc/exact.h) -- all-reduce as int64
extern "C" int isd_eegnet_sync_block_kind(int backward, int stage) { return (backward || stage > 0) ? 1 : 0; }
