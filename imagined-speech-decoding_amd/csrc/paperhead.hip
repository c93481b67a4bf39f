// HeadConv_Paper_Version (reference: src/fast/models/fast.py:170-196) forward + backward on gfx950.
//
//   cnn1_t Conv2d(1->F1,(1,3),bias) -> cnn1_s Conv2d(F1->F1,(C,1)) -> BN -> GELU -> MaxPool(1,2)
//   3 x [ Conv2d((1,3), valid, no bias) -> BN -> GELU -> MaxPool(1,2) ]  widths F2 = F3 = F/3, F4 = F
//   mean over the remaining time steps.                                   F1 = F/2
//
// cnn1_t and cnn1_s are both linear with nothing between them, so layer 1 runs as one C -> F1 three-tap
// convolution with Weff[o,c,k] = sum_f Ws[o,f,c] Wt[f,k], beff[o] = sum_{f,c} Ws[o,f,c] bt[f]; the
// [B,F1,C,T-2] intermediate of the reference is never formed and the gradients of Wt, bt, Ws are assembled
// from dWeff / dbeff.  Every layer keeps its pre-BN output y_l (the batch statistics need a grid-wide
// reduction between the convolution and the normalisation), the pooled activations a_l are the next
// layer's input.  Convolutions: one thread per (trial, time step), 16 output channels in registers, weights
// through scalar loads; weight gradients: MFMA 16x16x4 (M = out channel, N = in channel, K = time), one
// accumulator per tap, persistent waves and per-wave partial slabs reduced in a fixed order.
// Statistics and cross-batch sums accumulate in fp64.
#include "common.h"
#include "zonebatch.h"
#include "exact.h"
#include <math.h>
#include <stddef.h>

namespace isd {

// every launch of this file goes through zone_launch: issued at once, or recorded for a zone-batched launch (zonebatch.h)
#define ISD_ZLAUNCH(...)                                    \
  do {                                                      \
    if (!zone_launch(__VA_ARGS__)) return ISD_ERR_INVALID;  \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kPhMaxF = 64;      // widest layer supported (feature_dim <= 64)
constexpr int kPhSlabs = 1024;

struct PhGeo {
  int C, T, F;
  int Fo[4], Ci[4], Ti[4], To[4], Tp[4], Fp[4], Cp[4];   // per layer: out/in channels, in/out/pooled length, paddings
  int wt, bt, ws, w[4], g[4], b[4], n_params;            // parameter offsets (w[0] unused)
  int rm[4], rv[4], n_bufs;                              // buffer offsets
};

struct PhStats {
  // per layer one contiguous block of batch sums (under data parallelism it is all-reduced before the layer's
  // normalisation / BatchNorm backward: synchronised BatchNorm)
  // (ExactAcc, csrc/exact.h: integer atomics -- the same bits whatever the order the workgroups arrive in)
  ExactAcc s[4][2][kPhMaxF];                // [layer][sum y | sum y^2]
  ExactAcc d[4][2][kPhMaxF];                // [layer][sum dyhat | sum dyhat*xhat]
};
struct PhCoef {
  float A[4][kPhMaxF], Bc[4][kPhMaxF], mu[4][kPhMaxF], isg[4][kPhMaxF];
  float cA[4][kPhMaxF], cB[4][kPhMaxF], cC[4][kPhMaxF];
};

__device__ __forceinline__ float ph_gelu(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float ph_gelu_grad(float x) {
  return 0.5f * (1.f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * expf(-0.5f * x * x);
}
__device__ __forceinline__ float ph_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float ph_block_sum(float v, float* red) {
  v = ph_wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float s = 0.f;
  if (threadIdx.x == 0)
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];
  return s;
}

// Weff / beff, and the two scalar-load layouts of every layer's weight:
//   Wf[l][c][k][Fp]  (forward: 16 consecutive output channels per tap)
//   Wb[l][o][k][Cp]  (data gradient: 16 consecutive input channels per tap)
__device__ __forceinline__ void ph_prep_kernel_body(const float* __restrict__ params, float* __restrict__ Wf,
                                                      float* __restrict__ Wb, float* __restrict__ beff, PhGeo g,
                                                      int l, int64_t wf_off, int64_t wb_off,
    unsigned zgx, unsigned zgy, unsigned zbz, unsigned zgz) {   // this zone's own gridDim.x/.y, blockIdx.z, gridDim.z
  const int Fo = g.Fo[l], Ci = g.Ci[l], Fp = g.Fp[l], Cp = g.Cp[l];
  const int e = blockIdx.x * 256 + threadIdx.x;
  const int n = (Fp > Fo ? Fp : Fo) * (Cp > Ci ? Cp : Ci) * 3;
  if (l == 0 && e < Fo) {
    float s = 0.f;
    for (int f = 0; f < Fo; ++f) {
      float ws = 0.f;
      for (int c = 0; c < Ci; ++c) ws += params[g.ws + (e * Fo + f) * Ci + c];
      s = fmaf(ws, params[g.bt + f], s);
    }
    beff[e] = s;
  }
  if (l == 0 && e >= Fo && e < g.Fp[0]) beff[e] = 0.f;
  if (e >= n) return;
  const int k = e % 3, c = (e / 3) % (Cp > Ci ? Cp : Ci), o = e / (3 * (Cp > Ci ? Cp : Ci));
  float w = 0.f;
  if (o < Fo && c < Ci) {
    if (l == 0) {
      for (int f = 0; f < Fo; ++f) w = fmaf(params[g.ws + (o * Fo + f) * Ci + c], params[g.wt + f * 3 + k], w);
    } else {
      w = params[g.w[l] + (o * Ci + c) * 3 + k];
    }
  }
  if (c < Ci && o < Fp) Wf[wf_off + ((int64_t)c * 3 + k) * Fp + o] = w;
  if (o < Fo && c < Cp) Wb[wb_off + ((int64_t)o * 3 + k) * Cp + c] = w;
}
ISD_ZONE_FN(ph_prep_kernel, 256)
__global__ __launch_bounds__(256) void ph_prep_kernel(const float* __restrict__ params, float* __restrict__ Wf,
                                                      float* __restrict__ Wb, float* __restrict__ beff, PhGeo g,
                                                      int l, int64_t wf_off, int64_t wb_off) {
  ph_prep_kernel_body(params, Wf, Wb, beff, g, l, wf_off, wb_off,
      gridDim.x, gridDim.y, blockIdx.z, gridDim.z);
}
ISD_ZONE_REGISTER(ph_prep_kernel)

// y[b,o,t] = bias[o] + sum_{c,k} W[o,c,k] in[b,c,t+k];  per-channel sums of y and y^2.
// Persistent blocks over (b,t); blockIdx.y selects a tile of 16 output channels.
__device__ __forceinline__ void ph_conv_kernel_body(const float* __restrict__ in, const float* __restrict__ Wf,
                                                      const float* __restrict__ bias, float* __restrict__ y,
                                                      ExactAcc* __restrict__ s1, ExactAcc* __restrict__ s2, int64_t B,
                                                      int Ci, int Ti, int To, int Fo, int Fp, int want_stats,
    unsigned zgx, unsigned zgy, unsigned zbz, unsigned zgz) {   // this zone's own gridDim.x/.y, blockIdx.z, gridDim.z
  __shared__ float red[4];
  __shared__ float tot[32];
  const int o0 = blockIdx.y * 16;
  float a1[16], a2[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a1[i] = a2[i] = 0.f;
  const int64_t n = B * To;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)zgx * 256) {
    const int64_t b = e / To;
    const int t = (int)(e - b * To);
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = bias ? bias[o0 + i] : 0.f;
    const float* ip = in + b * Ci * Ti + t;
    for (int c = 0; c < Ci; ++c) {
      const float v0 = ip[c * Ti], v1 = ip[c * Ti + 1], v2 = ip[c * Ti + 2];
      const float* w = Wf + (int64_t)c * 3 * Fp + o0;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = fmaf(w[i], v0, fmaf(w[Fp + i], v1, fmaf(w[2 * Fp + i], v2, acc[i])));
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (o0 + i < Fo) y[(b * Fo + o0 + i) * To + t] = acc[i];
      a1[i] += acc[i];
      a2[i] = fmaf(acc[i], acc[i], a2[i]);
    }
  }
  if (want_stats) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float r1 = ph_block_sum(a1[i], red);
      const float r2 = ph_block_sum(a2[i], red);
      if (threadIdx.x == 0) { tot[i] = r1; tot[16 + i] = r2; }
    }
    __syncthreads();
    const int i = threadIdx.x & 15;
    if (threadIdx.x < 16 && o0 + i < Fo) exact_add(&s1[o0 + i], tot[i]);
    else if (threadIdx.x >= 16 && threadIdx.x < 32 && o0 + i < Fo) exact_add(&s2[o0 + i], tot[16 + i]);
  }
}
ISD_ZONE_FN(ph_conv_kernel, 256)
__global__ __launch_bounds__(256) void ph_conv_kernel(const float* __restrict__ in, const float* __restrict__ Wf,
                                                      const float* __restrict__ bias, float* __restrict__ y,
                                                      ExactAcc* __restrict__ s1, ExactAcc* __restrict__ s2, int64_t B,
                                                      int Ci, int Ti, int To, int Fo, int Fp, int want_stats) {
  ph_conv_kernel_body(in, Wf, bias, y, s1, s2, B, Ci, Ti, To, Fo, Fp, want_stats,
      gridDim.x, gridDim.y, blockIdx.z, gridDim.z);
}
ISD_ZONE_REGISTER(ph_conv_kernel)

// BN coefficients of layer l: bn(y) = A y + Bc.  training: batch statistics (+ running update); eval: running buffers.
__device__ __forceinline__ void ph_finalize_kernel_body(const float* __restrict__ params, float* __restrict__ bufs,
                                   const PhStats* __restrict__ st, PhCoef* __restrict__ co, PhGeo g, int l, double N,
                                   int training, float momentum, float eps,
    unsigned zgx, unsigned zgy, unsigned zbz, unsigned zgz) {   // this zone's own gridDim.x/.y, blockIdx.z, gridDim.z
  const int o = threadIdx.x;
  if (o >= g.Fo[l]) return;
  double mu, var;
  if (training) {
    mu = exact_get(&st->s[l][0][o]) / N;
    var = exact_get(&st->s[l][1][o]) / N - mu * mu;
    if (var < 0.0) var = 0.0;
    bufs[g.rm[l] + o] = (1.f - momentum) * bufs[g.rm[l] + o] + momentum * (float)mu;
    bufs[g.rv[l] + o] = (1.f - momentum) * bufs[g.rv[l] + o] + momentum * (float)(var * N / (N > 1.0 ? N - 1.0 : 1.0));
  } else {
    mu = bufs[g.rm[l] + o];
    var = bufs[g.rv[l] + o];
  }
  const double isg = 1.0 / sqrt(var + (double)eps);
  const double gm = params[g.g[l] + o], bt = params[g.b[l] + o];
  co->A[l][o] = (float)(gm * isg);
  co->Bc[l][o] = (float)(bt - gm * mu * isg);
  co->mu[l][o] = (float)mu;
  co->isg[l][o] = (float)isg;
}
ISD_ZONE_FN(ph_finalize_kernel, 1024)
__global__ void ph_finalize_kernel(const float* __restrict__ params, float* __restrict__ bufs,
                                   const PhStats* __restrict__ st, PhCoef* __restrict__ co, PhGeo g, int l, double N,
                                   int training, float momentum, float eps) {
  ph_finalize_kernel_body(params, bufs, st, co, g, l, N, training, momentum, eps,
      gridDim.x, gridDim.y, blockIdx.z, gridDim.z);
}
ISD_ZONE_REGISTER(ph_finalize_kernel)

// a[b,o,p] = max(GELU(bn(y[2p])), GELU(bn(y[2p+1])))
__device__ __forceinline__ void ph_pool_kernel_body(const float* __restrict__ y, const PhCoef* __restrict__ co,
                                                      float* __restrict__ a, int64_t n, int l, int Fo, int To, int Tp,
    unsigned zgx, unsigned zgy, unsigned zbz, unsigned zgz) {   // this zone's own gridDim.x/.y, blockIdx.z, gridDim.z
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  const int64_t row = e / Tp;
  const int p = (int)(e - row * Tp), o = (int)(row % Fo);
  const float A = co->A[l][o], Bc = co->Bc[l][o];
  const float* yr = y + row * To + 2 * p;
  a[e] = fmaxf(ph_gelu(fmaf(A, yr[0], Bc)), ph_gelu(fmaf(A, yr[1], Bc)));
}
ISD_ZONE_FN(ph_pool_kernel, 256)
__global__ __launch_bounds__(256) void ph_pool_kernel(const float* __restrict__ y, const PhCoef* __restrict__ co,
                                                      float* __restrict__ a, int64_t n, int l, int Fo, int To, int Tp) {
  ph_pool_kernel_body(y, co, a, n, l, Fo, To, Tp,
      gridDim.x, gridDim.y, blockIdx.z, gridDim.z);
}
ISD_ZONE_REGISTER(ph_pool_kernel)

// out[b,o] = mean_p a4[b,o,p]
__device__ __forceinline__ void ph_mean_kernel_body(const float* __restrict__ a, float* __restrict__ out, int64_t rows,
                                                      int Tp,
    unsigned zgx, unsigned zgy, unsigned zbz, unsigned zgz) {   // this zone's own gridDim.x/.y, blockIdx.z, gridDim.z
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  float s = 0.f;
  for (int p = 0; p < Tp; ++p) s += a[r * Tp + p];
  out[r] = s / (float)Tp;
}
ISD_ZONE_FN(ph_mean_kernel, 256)
__global__ __launch_bounds__(256) void ph_mean_kernel(const float* __restrict__ a, float* __restrict__ out, int64_t rows,
                                                      int Tp) {
  ph_mean_kernel_body(a, out, rows, Tp,
      gridDim.x, gridDim.y, blockIdx.z, gridDim.z);
}
ISD_ZONE_REGISTER(ph_mean_kernel)

// ------------------------------------------------------------------------------------------------ backward
// Max-pool routing + GELU':  dyh[b,o,t] (gradient w.r.t. the BN output) from da[b,o,p] (or dout[b,o]/Tp for the
// last layer), and the BN backward sums.  blockIdx.y = channel; persistent over (b,p).
__device__ __forceinline__ void ph_bwd_pool_kernel_body(const float* __restrict__ y, const float* __restrict__ da,
                                                          const float* __restrict__ dout,
                                                          const PhCoef* __restrict__ co, float* __restrict__ dyh,
                                                          PhStats* __restrict__ st, int64_t B, int l, int Fo, int To,
                                                          int Tp,
    unsigned zgx, unsigned zgy, unsigned zbz, unsigned zgz) {   // this zone's own gridDim.x/.y, blockIdx.z, gridDim.z
  __shared__ float red[4];
  const int o = blockIdx.y;
  const float A = co->A[l][o], Bc = co->Bc[l][o], mu = co->mu[l][o], isg = co->isg[l][o];
  float s1 = 0.f, s2 = 0.f;
  const int64_t n = B * Tp;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)zgx * 256) {
    const int64_t b = e / Tp;
    const int p = (int)(e - b * Tp);
    const int64_t row = b * Fo + o;
    const float up = dout ? dout[row] / (float)Tp : da[row * Tp + p];
    const float y0 = y[row * To + 2 * p], y1 = y[row * To + 2 * p + 1];
    const float v0 = fmaf(A, y0, Bc), v1 = fmaf(A, y1, Bc);
    const bool second = ph_gelu(v1) > ph_gelu(v0);                  // ties -> first element, like max_pool2d
    const float d = up * ph_gelu_grad(second ? v1 : v0);
    dyh[row * To + 2 * p] = second ? 0.f : d;
    dyh[row * To + 2 * p + 1] = second ? d : 0.f;
    if ((To & 1) && p == Tp - 1) dyh[row * To + To - 1] = 0.f;      // the sample MaxPool drops
    s1 += d;
    s2 = fmaf(d, ((second ? y1 : y0) - mu) * isg, s2);
  }
  const float r1 = ph_block_sum(s1, red);
  const float r2 = ph_block_sum(s2, red);
  if (threadIdx.x == 0) {
    exact_add(&st->d[l][0][o], r1);
    exact_add(&st->d[l][1][o], r2);
  }
}
ISD_ZONE_FN(ph_bwd_pool_kernel, 256)
__global__ __launch_bounds__(256) void ph_bwd_pool_kernel(const float* __restrict__ y, const float* __restrict__ da,
                                                          const float* __restrict__ dout,
                                                          const PhCoef* __restrict__ co, float* __restrict__ dyh,
                                                          PhStats* __restrict__ st, int64_t B, int l, int Fo, int To,
                                                          int Tp) {
  ph_bwd_pool_kernel_body(y, da, dout, co, dyh, st, B, l, Fo, To, Tp,
      gridDim.x, gridDim.y, blockIdx.z, gridDim.z);
}
ISD_ZONE_REGISTER(ph_bwd_pool_kernel)

__device__ __forceinline__ void ph_bwd_coef_kernel_body(const float* __restrict__ params, float* __restrict__ dparams,
                                   const PhStats* __restrict__ st, PhCoef* __restrict__ co, PhGeo g, int l, double N,
                                   int bn_train, double gs,
    unsigned zgx, unsigned zgy, unsigned zbz, unsigned zgz) {   // this zone's own gridDim.x/.y, blockIdx.z, gridDim.z
  const int o = threadIdx.x;
  if (o >= g.Fo[l]) return;
  const double inv = bn_train ? 1.0 / N : 0.0;           // running statistics do not depend on the batch: no mean terms
  dparams[g.g[l] + o] = (float)(exact_get(&st->d[l][1][o]) * gs);    // global sums on every rank: pre-divided by the world size
  dparams[g.b[l] + o] = (float)(exact_get(&st->d[l][0][o]) * gs);
  co->cA[l][o] = params[g.g[l] + o] * co->isg[l][o];
  co->cB[l][o] = (float)(exact_get(&st->d[l][0][o]) * inv);
  co->cC[l][o] = (float)(exact_get(&st->d[l][1][o]) * inv);
}
ISD_ZONE_FN(ph_bwd_coef_kernel, 1024)
__global__ void ph_bwd_coef_kernel(const float* __restrict__ params, float* __restrict__ dparams,
                                   const PhStats* __restrict__ st, PhCoef* __restrict__ co, PhGeo g, int l, double N,
                                   int bn_train, double gs) {
  ph_bwd_coef_kernel_body(params, dparams, st, co, g, l, N, bn_train, gs,
      gridDim.x, gridDim.y, blockIdx.z, gridDim.z);
}
ISD_ZONE_REGISTER(ph_bwd_coef_kernel)

// dy = cA (dyh - cB - xhat cC) in place
__device__ __forceinline__ void ph_bwd_bn_kernel_body(float* __restrict__ dyh, const float* __restrict__ y,
                                                        const PhCoef* __restrict__ co, int64_t n, int l, int Fo,
                                                        int To,
    unsigned zgx, unsigned zgy, unsigned zbz, unsigned zgz) {   // this zone's own gridDim.x/.y, blockIdx.z, gridDim.z
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  const int o = (int)((e / To) % Fo);
  const float xh = (y[e] - co->mu[l][o]) * co->isg[l][o];
  dyh[e] = co->cA[l][o] * (dyh[e] - co->cB[l][o] - xh * co->cC[l][o]);
}
ISD_ZONE_FN(ph_bwd_bn_kernel, 256)
__global__ __launch_bounds__(256) void ph_bwd_bn_kernel(float* __restrict__ dyh, const float* __restrict__ y,
                                                        const PhCoef* __restrict__ co, int64_t n, int l, int Fo,
                                                        int To) {
  ph_bwd_bn_kernel_body(dyh, y, co, n, l, Fo, To,
      gridDim.x, gridDim.y, blockIdx.z, gridDim.z);
}
ISD_ZONE_REGISTER(ph_bwd_bn_kernel)

// da[b,c,s] = sum_{o,k} W[o,c,k] dy[b,o,s-k]   (gradient w.r.t. the layer input = the previous pooled activation)
__device__ __forceinline__ void ph_bwd_dgrad_kernel_body(const float* __restrict__ dy, const float* __restrict__ Wb,
                                                           float* __restrict__ da, int64_t B, int Ci, int Cp, int Ti,
                                                           int To, int Fo,
    unsigned zgx, unsigned zgy, unsigned zbz, unsigned zgz) {   // this zone's own gridDim.x/.y, blockIdx.z, gridDim.z
  const int c0 = blockIdx.y * 16;
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= B * Ti) return;
  const int64_t b = e / Ti;
  const int s = (int)(e - b * Ti);
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int o = 0; o < Fo; ++o) {
    const float* dr = dy + (b * Fo + o) * To;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int t = s - k;
      const float dv = (t >= 0 && t < To) ? dr[t] : 0.f;
      const float* w = Wb + ((int64_t)o * 3 + k) * Cp + c0;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = fmaf(w[i], dv, acc[i]);
    }
  }
#pragma unroll
  for (int i = 0; i < 16; ++i)
    if (c0 + i < Ci) da[(b * Ci + c0 + i) * Ti + s] = acc[i];
}
ISD_ZONE_FN(ph_bwd_dgrad_kernel, 256)
__global__ __launch_bounds__(256) void ph_bwd_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ Wb,
                                                           float* __restrict__ da, int64_t B, int Ci, int Cp, int Ti,
                                                           int To, int Fo) {
  ph_bwd_dgrad_kernel_body(dy, Wb, da, B, Ci, Cp, Ti, To, Fo,
      gridDim.x, gridDim.y, blockIdx.z, gridDim.z);
}
ISD_ZONE_REGISTER(ph_bwd_dgrad_kernel)

// dW[o,c,k] = sum_{b,t} dy[b,o,t] in[b,c,t+k] on the matrix cores (A = dy tile [16 o][4 t], B = in tile [4 t][16 c],
// one accumulator per tap) + the bias column sum_{b,t} dy[b,o,t].  blockIdx.y = (o tile, c tile); persistent over b.
// Partial slabs: part[block][o][c][k] and pbias[block][o].
__device__ __forceinline__ void ph_bwd_wgrad_kernel_body(const float* __restrict__ dy, const float* __restrict__ in,
                                                          float* __restrict__ part, float* __restrict__ pbias,
                                                          int64_t B, int Ci, int Ti, int To, int Fo, int n_ctile,
    unsigned zgx, unsigned zgy, unsigned zbz, unsigned zgz) {   // this zone's own gridDim.x/.y, blockIdx.z, gridDim.z
  const int lane = threadIdx.x, q = lane >> 4, jl = lane & 15;
  const int ot = blockIdx.y / n_ctile, ct = blockIdx.y - ot * n_ctile;
  const int o = ot * 16 + jl, c = ct * 16 + jl;
  f32x4 acc[3], accb = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 3; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int64_t b = blockIdx.x; b < B; b += zgx) {
    const float* dr = dy + (b * Fo + (o < Fo ? o : 0)) * To;
    const float* ir = in + (b * Ci + (c < Ci ? c : 0)) * Ti;
    for (int t0 = 0; t0 < To; t0 += 4) {
      const int t = t0 + q;
      const bool tv = t < To;
      const float af = (tv && o < Fo) ? dr[t] : 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float bf = (tv && c < Ci) ? ir[t + k] : 0.f;
        acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, acc[k], 0, 0, 0);
      }
      if (pbias && ct == 0) accb = __builtin_amdgcn_mfma_f32_16x16x4f32(af, tv ? 1.f : 0.f, accb, 0, 0, 0);
    }
  }
  float* slab = part + (int64_t)blockIdx.x * Fo * Ci * 3;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int oo = ot * 16 + 4 * q + r;
    if (oo < Fo && c < Ci) {
#pragma unroll
      for (int k = 0; k < 3; ++k) slab[((int64_t)oo * Ci + c) * 3 + k] = acc[k][r];
    }
    if (pbias && ct == 0 && jl == 0 && oo < Fo) pbias[(int64_t)blockIdx.x * Fo + oo] = accb[r];
  }
}
ISD_ZONE_FN(ph_bwd_wgrad_kernel, 64)
__global__ __launch_bounds__(64) void ph_bwd_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ in,
                                                          float* __restrict__ part, float* __restrict__ pbias,
                                                          int64_t B, int Ci, int Ti, int To, int Fo, int n_ctile) {
  ph_bwd_wgrad_kernel_body(dy, in, part, pbias, B, Ci, Ti, To, Fo, n_ctile,
      gridDim.x, gridDim.y, blockIdx.z, gridDim.z);
}
ISD_ZONE_REGISTER(ph_bwd_wgrad_kernel)

__device__ __forceinline__ void ph_reduce_kernel_body(const float* __restrict__ part, int n_slabs, int64_t n,
                                                        float* __restrict__ dst,
    unsigned zgx, unsigned zgy, unsigned zbz, unsigned zgz) {   // this zone's own gridDim.x/.y, blockIdx.z, gridDim.z
  // block = 64 elements x 4 slab groups (contiguous quarters of the slabs, four loads in flight each, combined in LDS in
  // a fixed order): one thread per element walking all the slabs was a chain of ~160 dependent load pairs, 42 us a launch
  __shared__ float red[4][64];
  const int l = threadIdx.x & 63, gq = threadIdx.x >> 6;
  const int64_t e = (int64_t)blockIdx.x * 64 + l;
  const int per = (n_slabs + 3) / 4;
  const int k_lo = gq * per, k_hi = k_lo + per < n_slabs ? k_lo + per : n_slabs;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (e < n) {
    int k = k_lo;
    for (; k + 3 < k_hi; k += 4) {
      s0 += part[(int64_t)k * n + e];
      s1 += part[(int64_t)(k + 1) * n + e];
      s2 += part[(int64_t)(k + 2) * n + e];
      s3 += part[(int64_t)(k + 3) * n + e];
    }
    for (; k < k_hi; ++k) s0 += part[(int64_t)k * n + e];
  }
  red[gq][l] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (gq == 0 && e < n) dst[e] = (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]);
}
ISD_ZONE_FN(ph_reduce_kernel, 256)
__global__ __launch_bounds__(256) void ph_reduce_kernel(const float* __restrict__ part, int n_slabs, int64_t n,
                                                        float* __restrict__ dst) {
  ph_reduce_kernel_body(part, n_slabs, n, dst,
      gridDim.x, gridDim.y, blockIdx.z, gridDim.z);
}
ISD_ZONE_REGISTER(ph_reduce_kernel)

// dWeff [F1][C][3], dbeff [F1] -> gradients of cnn1_t.weight, cnn1_t.bias, cnn1_s.weight.  One block.
__device__ __forceinline__ void ph_bwd_l1_kernel_body(const float* __restrict__ params, float* __restrict__ dparams,
                                                        const float* __restrict__ dWeff, const float* __restrict__ dbeff,
                                                        PhGeo g,
    unsigned zgx, unsigned zgy, unsigned zbz, unsigned zgz) {   // this zone's own gridDim.x/.y, blockIdx.z, gridDim.z
  const int F1 = g.Fo[0], C = g.C;
  const float* Wt = params + g.wt;
  const float* bt = params + g.bt;
  const float* Ws = params + g.ws;
  for (int e = threadIdx.x; e < F1 * F1 * C; e += 256) {          // dWs[o,f,c]
    const int c = e % C, f = (e / C) % F1, o = e / (C * F1);
    float s = dbeff[o] * bt[f];
#pragma unroll
    for (int k = 0; k < 3; ++k) s = fmaf(dWeff[(o * C + c) * 3 + k], Wt[f * 3 + k], s);
    dparams[g.ws + e] = s;
  }
  for (int e = threadIdx.x; e < F1 * 3; e += 256) {               // dWt[f,k]
    const int f = e / 3, k = e - f * 3;
    float s = 0.f;
    for (int o = 0; o < F1; ++o)
      for (int c = 0; c < C; ++c) s = fmaf(dWeff[(o * C + c) * 3 + k], Ws[(o * F1 + f) * C + c], s);
    dparams[g.wt + e] = s;
  }
  for (int f = threadIdx.x; f < F1; f += 256) {                   // dbt[f]
    float s = 0.f;
    for (int o = 0; o < F1; ++o) {
      float ws = 0.f;
      for (int c = 0; c < C; ++c) ws += Ws[(o * F1 + f) * C + c];
      s = fmaf(dbeff[o], ws, s);
    }
    dparams[g.bt + f] = s;
  }
}
ISD_ZONE_FN(ph_bwd_l1_kernel, 256)
__global__ __launch_bounds__(256) void ph_bwd_l1_kernel(const float* __restrict__ params, float* __restrict__ dparams,
                                                        const float* __restrict__ dWeff, const float* __restrict__ dbeff,
                                                        PhGeo g) {
  ph_bwd_l1_kernel_body(params, dparams, dWeff, dbeff, g,
      gridDim.x, gridDim.y, blockIdx.z, gridDim.z);
}
ISD_ZONE_REGISTER(ph_bwd_l1_kernel)

}  // namespace isd

using namespace isd;

struct isd_paperhead_plan {
  PhGeo g;
};

static inline int64_t ph_al64(int64_t v) { return (v + 63) / 64 * 64; }

extern "C" int isd_paperhead_plan_create(isd_paperhead_plan** out, int in_channels, int feature_dim, int T) {
  ISD_CHECK_ARG(out, "isd_paperhead_plan_create: null argument");
  ISD_CHECK_ARG(in_channels >= 1 && in_channels <= 4096, "isd_paperhead_plan_create: in_channels=%d", in_channels);
  ISD_CHECK_ARG(feature_dim >= 3 && feature_dim <= kPhMaxF, "isd_paperhead_plan_create: feature_dim=%d not in [3,%d]",
                feature_dim, kPhMaxF);
  isd_paperhead_plan* p = new isd_paperhead_plan();
  PhGeo& g = p->g;
  g.C = in_channels; g.T = T; g.F = feature_dim;
  g.Fo[0] = feature_dim / 2; g.Fo[1] = feature_dim / 3; g.Fo[2] = feature_dim / 3; g.Fo[3] = feature_dim;
  int Ti = T;
  for (int l = 0; l < 4; ++l) {
    g.Ci[l] = l == 0 ? in_channels : g.Fo[l - 1];
    g.Ti[l] = Ti;
    g.To[l] = Ti - 2;
    g.Tp[l] = g.To[l] / 2;
    g.Fp[l] = (g.Fo[l] + 15) / 16 * 16;
    g.Cp[l] = (g.Ci[l] + 15) / 16 * 16;
    Ti = g.Tp[l];
    if (g.To[l] < 2 || g.Tp[l] < 1) {
      set_error("isd_paperhead_plan_create: T=%d is too short for four conv(3)+MaxPool(2) stages (T >= 46)", T);
      delete p;
      return ISD_ERR_INVALID;
    }
  }
  int o = 0;
  const int F1 = g.Fo[0];
  g.wt = o; o += F1 * 3;
  g.bt = o; o += F1;
  g.ws = o; o += F1 * F1 * in_channels;
  g.w[0] = -1;
  g.g[0] = o; o += F1;
  g.b[0] = o; o += F1;
  for (int l = 1; l < 4; ++l) {
    g.w[l] = o; o += g.Fo[l] * g.Ci[l] * 3;
    g.g[l] = o; o += g.Fo[l];
    g.b[l] = o; o += g.Fo[l];
  }
  g.n_params = o;
  o = 0;
  for (int l = 0; l < 4; ++l) {
    g.rm[l] = o; o += g.Fo[l];
    g.rv[l] = o; o += g.Fo[l];
  }
  g.n_bufs = o;
  *out = p;
  return ISD_OK;
}

extern "C" int isd_paperhead_plan_destroy(isd_paperhead_plan* p) {
  delete p;
  return ISD_OK;
}
extern "C" int64_t isd_paperhead_param_count(const isd_paperhead_plan* p) { return p ? p->g.n_params : ISD_ERR_INVALID; }
extern "C" int64_t isd_paperhead_buffer_count(const isd_paperhead_plan* p) { return p ? p->g.n_bufs : ISD_ERR_INVALID; }

namespace {
struct PhWs {
  int64_t stats, coef, beff, wf[4], wb[4], y[4], a[4], dy[4], da[4], part, pbias, dweff, dbeff, total;
};
PhWs ph_layout(const PhGeo& g, int64_t B) {
  PhWs w;
  int64_t o = 0;
  w.stats = o; o += ph_al64((int64_t)(sizeof(PhStats) + 3) / 4);
  w.coef = o; o += ph_al64((int64_t)(sizeof(PhCoef) + 3) / 4);
  w.beff = o; o += 64;
  int64_t max_w = 0;
  for (int l = 0; l < 4; ++l) {
    w.wf[l] = o; o += ph_al64((int64_t)g.Ci[l] * 3 * g.Fp[l]);
    w.wb[l] = o; o += ph_al64((int64_t)g.Fo[l] * 3 * g.Cp[l]);
    w.y[l] = o; o += ph_al64(B * g.Fo[l] * g.To[l]);
    w.a[l] = o; o += ph_al64(B * g.Fo[l] * g.Tp[l]);
    w.dy[l] = o; o += ph_al64(B * g.Fo[l] * g.To[l]);
    w.da[l] = o; o += ph_al64(B * g.Fo[l] * g.Tp[l]);
    const int64_t nw = (int64_t)g.Fo[l] * g.Ci[l] * 3;
    if (nw > max_w) max_w = nw;
  }
  w.part = o; o += ph_al64((int64_t)kPhSlabs * max_w);
  w.pbias = o; o += ph_al64((int64_t)kPhSlabs * kPhMaxF);
  w.dweff = o; o += ph_al64((int64_t)g.Fo[0] * g.C * 3);
  w.dbeff = o; o += 64;
  w.total = o;
  return w;
}
}  // namespace

extern "C" int64_t isd_paperhead_workspace_bytes(const isd_paperhead_plan* p, int64_t B) {
  if (!p || B < 0) return ISD_ERR_INVALID;
  return ph_layout(p->g, B).total * 4;
}

// Forward / backward in stages (synchronised BatchNorm): between two stages one layer's block of fp64
// batch sums sits complete in the workspace (isd_paperhead_sync_block); under data parallelism the caller all-reduces
// it there and passes the world size.
//   forward  stage 0: weight preparation, conv 1                          -> sums of layer 1
//            stage s = 1..3: BN + GELU + pool of layer s, conv s + 1      -> sums of layer s + 1
//            stage 4: BN + GELU + pool of layer 4, mean over time
//   backward stage 0: pool routing / GELU' of layer 4                     -> BN backward sums of layer 4
//            stage k = 1..3: BN backward, weight + data gradient of layer 5 - k, pool routing of layer 4 - k -> its sums
//            stage 4: BN backward and gradients of layer 1 (and dx)
static int ph_forward_stage(const isd_paperhead_plan* p, int stage, const float* x, const float* params, float* buffers,
                            float* out, float* ws, int64_t B, int training, float momentum, float eps, int world,
                            hipStream_t st) {
  const PhGeo& g = p->g;
  const PhWs w = ph_layout(g, B);
  PhStats* S = (PhStats*)(ws + w.stats);
  PhCoef* Cf = (PhCoef*)(ws + w.coef);
  if (stage == 0) {
    ISD_HIP_TRY(zone_clear(S, sizeof(PhStats), st));
    for (int l = 0; l < 4; ++l) {
      const int nmax = (g.Fp[l] > g.Fo[l] ? g.Fp[l] : g.Fo[l]) * g.Cp[l] * 3;
      ISD_ZLAUNCH(ph_prep_kernel, dim3((unsigned)cdiv(nmax, 256)), dim3(256), 0, st, params, ws, ws, ws + w.beff,
                         g, l, w.wf[l], w.wb[l]);
    }
  }
  if (stage > 0) {
    const int l = stage - 1;
    ISD_ZLAUNCH(ph_finalize_kernel, dim3(1), dim3(64), 0, st, params, buffers, S, Cf, g, l,
                       (double)B * (double)g.To[l] * (double)world, training, momentum, eps);
    const int64_t np = B * g.Fo[l] * g.Tp[l];
    ISD_ZLAUNCH(ph_pool_kernel, dim3((unsigned)cdiv(np, 256)), dim3(256), 0, st, ws + w.y[l], Cf, ws + w.a[l], np,
                       l, g.Fo[l], g.To[l], g.Tp[l]);
  }
  if (stage < 4) {
    const int l = stage;
    const float* in = l == 0 ? x : ws + w.a[l - 1];
    const int64_t n = B * g.To[l];
    const unsigned gx = (unsigned)(cdiv(n, 256) < 2048 ? cdiv(n, 256) : 2048);
    ISD_ZLAUNCH(ph_conv_kernel, dim3(gx, (unsigned)(g.Fp[l] / 16)), dim3(256), 0, st, in, ws + w.wf[l],
                       l == 0 ? ws + w.beff : (const float*)nullptr, ws + w.y[l], S->s[l][0], S->s[l][1], B, g.Ci[l],
                       g.Ti[l], g.To[l], g.Fo[l], g.Fp[l], training);
  } else {
    ISD_ZLAUNCH(ph_mean_kernel, dim3((unsigned)cdiv(B * g.F, 256)), dim3(256), 0, st, ws + w.a[3], out, B * g.F,
                       g.Tp[3]);
  }
  ISD_LAUNCH_CHECK();
  return ISD_OK;
}

static int ph_forward_check(const isd_paperhead_plan* p, const float* x, const float* params, float* buffers, float* out,
                            void* workspace, int64_t B, int world) {
  ISD_CHECK_ARG(p, "isd_paperhead_forward: null plan");
  ISD_CHECK_ARG(B >= 0 && B <= (int64_t)1 << 26, "isd_paperhead_forward: B=%lld", (long long)B);
  ISD_CHECK_ARG(B == 0 || (x && params && buffers && out && workspace), "isd_paperhead_forward: null argument");
  ISD_CHECK_ARG(world >= 1 && world <= 65536, "isd_paperhead_forward: world=%d", world);
  return ISD_OK;
}

extern "C" int isd_paperhead_forward(const isd_paperhead_plan* p, const float* x, const float* params, float* buffers,
                                     float* out, void* workspace, int64_t B, int training, float momentum, float eps,
                                     void* stream) {
  int rc = ph_forward_check(p, x, params, buffers, out, workspace, B, 1);
  if (rc || B == 0) return rc;
  for (int stage = 0; stage < 5 && rc == ISD_OK; ++stage)
    rc = ph_forward_stage(p, stage, x, params, buffers, out, (float*)workspace, B, training, momentum, eps, 1,
                          (hipStream_t)stream);
  return rc;
}

extern "C" int isd_paperhead_forward_stage(const isd_paperhead_plan* p, int stage, const float* x, const float* params,
                                           float* buffers, float* out, void* workspace, int64_t B, int training,
                                           float momentum, float eps, int world, void* stream) {
  int rc = ph_forward_check(p, x, params, buffers, out, workspace, B, world);
  if (rc || B == 0) return rc;
  ISD_CHECK_ARG(stage >= 0 && stage < 5, "isd_paperhead_forward_stage: stage=%d not in [0,5)", stage);
  return ph_forward_stage(p, stage, x, params, buffers, out, (float*)workspace, B, training, momentum, eps, world,
                          (hipStream_t)stream);
}

// The batch sums that are complete after `stage` (0..3) of the forward (backward = 0: sums of layer stage + 1) or of the
// backward (backward = 1: sums of layer 4 - stage): byte offset from the workspace base and number of doubles.
extern "C" int isd_paperhead_sync_block(const isd_paperhead_plan* p, int64_t B, int backward, int stage,
                                        int64_t* byte_offset, int64_t* n_doubles) {
  ISD_CHECK_ARG(p && byte_offset && n_doubles, "isd_paperhead_sync_block: null argument");
  ISD_CHECK_ARG(stage >= 0 && stage < 4 && B >= 0, "isd_paperhead_sync_block: stage=%d not in [0,4)", stage);
  const PhWs w = ph_layout(p->g, B);
  const size_t lo = backward ? offsetof(PhStats, d) + sizeof(ExactAcc) * 2 * kPhMaxF * (3 - stage)
                             : offsetof(PhStats, s) + sizeof(ExactAcc) * 2 * kPhMaxF * stage;
  *byte_offset = w.stats * 4 + (int64_t)lo;
  *n_doubles = (int64_t)(sizeof(ExactAcc) / 8) * 2 * kPhMaxF;       // 8-byte words (int64: isd_paperhead_sync_block_kind)
  return ISD_OK;
}

// 1: every block holds 64-bit integer words of exact accumulators (csrc/exact.h) -- all-reduce as int64
extern "C" int isd_paperhead_sync_block_kind(int backward, int stage) { (void)backward; (void)stage; return 1; }

static int ph_backward_stage(const isd_paperhead_plan* p, int stage, const float* x, const float* params,
                             const float* dout, float* dparams, float* dx, float* ws, int64_t B, int bn_train, int world,
                             hipStream_t st) {
  const PhGeo& g = p->g;
  const PhWs w = ph_layout(g, B);
  PhStats* S = (PhStats*)(ws + w.stats);
  PhCoef* Cf = (PhCoef*)(ws + w.coef);
  const int slabs = B < kPhSlabs ? (int)B : kPhSlabs;
  if (stage == 0)
    ISD_HIP_TRY(zone_clear((char*)S + offsetof(PhStats, d), sizeof(PhStats) - offsetof(PhStats, d), st));
  if (stage > 0) {
    // the layer whose BatchNorm sums are complete: BN backward, weight gradient, data gradient
    const int l = 4 - stage;
    ISD_ZLAUNCH(ph_bwd_coef_kernel, dim3(1), dim3(64), 0, st, params, dparams, S, Cf, g, l,
                       (double)B * (double)g.To[l] * (double)world, bn_train, 1.0 / (double)world);
    const int64_t ny = B * g.Fo[l] * g.To[l];
    ISD_ZLAUNCH(ph_bwd_bn_kernel, dim3((unsigned)cdiv(ny, 256)), dim3(256), 0, st, ws + w.dy[l], ws + w.y[l], Cf,
                       ny, l, g.Fo[l], g.To[l]);
    const float* in = l == 0 ? x : ws + w.a[l - 1];
    const int n_ctile = g.Cp[l] / 16, n_otile = g.Fp[l] / 16;
    const int64_t nw = (int64_t)g.Fo[l] * g.Ci[l] * 3;
    ISD_ZLAUNCH(ph_bwd_wgrad_kernel, dim3(slabs, (unsigned)(n_ctile * n_otile)), dim3(64), 0, st, ws + w.dy[l], in,
                       ws + w.part, l == 0 ? ws + w.pbias : (float*)nullptr, B, g.Ci[l], g.Ti[l], g.To[l], g.Fo[l],
                       n_ctile);
    ISD_ZLAUNCH(ph_reduce_kernel, dim3((unsigned)cdiv(nw, 64)), dim3(256), 0, st, ws + w.part, slabs, nw,
                       l == 0 ? ws + w.dweff : dparams + g.w[l]);
    if (l == 0) {
      ISD_ZLAUNCH(ph_reduce_kernel, dim3(1), dim3(256), 0, st, ws + w.pbias, slabs, (int64_t)g.Fo[0],
                         ws + w.dbeff);
      ISD_ZLAUNCH(ph_bwd_l1_kernel, dim3(1), dim3(256), 0, st, params, dparams, ws + w.dweff, ws + w.dbeff, g);
      if (dx)   // the fused first layer is one C -> F1 three-tap convolution: its data gradient is the input gradient
        ISD_ZLAUNCH(ph_bwd_dgrad_kernel, dim3((unsigned)cdiv(B * g.Ti[0], 256), (unsigned)n_ctile), dim3(256), 0,
                           st, ws + w.dy[0], ws + w.wb[0], dx, B, g.Ci[0], g.Cp[0], g.Ti[0], g.To[0], g.Fo[0]);
    } else {
      ISD_ZLAUNCH(ph_bwd_dgrad_kernel, dim3((unsigned)cdiv(B * g.Ti[l], 256), (unsigned)n_ctile), dim3(256), 0, st,
                         ws + w.dy[l], ws + w.wb[l], ws + w.da[l - 1], B, g.Ci[l], g.Cp[l], g.Ti[l], g.To[l], g.Fo[l]);
    }
  }
  if (stage < 4) {
    // max-pool routing + GELU' of the next layer down, and its BatchNorm backward sums
    const int l = 3 - stage;
    const int64_t np = B * g.Tp[l];
    const unsigned gx = (unsigned)(cdiv(np, 256) < 256 ? cdiv(np, 256) : 256);
    ISD_ZLAUNCH(ph_bwd_pool_kernel, dim3(gx, (unsigned)g.Fo[l]), dim3(256), 0, st, ws + w.y[l],
                       l == 3 ? (const float*)nullptr : ws + w.da[l], l == 3 ? dout : (const float*)nullptr, Cf,
                       ws + w.dy[l], S, B, l, g.Fo[l], g.To[l], g.Tp[l]);
  }
  ISD_LAUNCH_CHECK();
  return ISD_OK;
}

static int ph_backward_impl(const isd_paperhead_plan* p, const float* x, const float* params, const float* dout,
                            float* dparams, float* dx, void* workspace, int64_t B, int bn_train, void* stream) {
  ISD_CHECK_ARG(p, "isd_paperhead_backward: null plan");
  ISD_CHECK_ARG(B >= 1, "isd_paperhead_backward: B=%lld", (long long)B);
  ISD_CHECK_ARG(x && params && dout && dparams && workspace, "isd_paperhead_backward: null argument");
  int rc = ISD_OK;
  for (int stage = 0; stage < 5 && rc == ISD_OK; ++stage)
    rc = ph_backward_stage(p, stage, x, params, dout, dparams, dx, (float*)workspace, B, bn_train, 1, (hipStream_t)stream);
  return rc;
}

extern "C" int isd_paperhead_backward_stage(const isd_paperhead_plan* p, int stage, const float* x, const float* params,
                                            const float* dout, float* dparams, void* workspace, int64_t B, int world,
                                            void* stream) {
  ISD_CHECK_ARG(p, "isd_paperhead_backward_stage: null plan");
  ISD_CHECK_ARG(B >= 1, "isd_paperhead_backward_stage: B=%lld", (long long)B);
  ISD_CHECK_ARG(x && params && dout && dparams && workspace, "isd_paperhead_backward_stage: null argument");
  ISD_CHECK_ARG(stage >= 0 && stage < 5, "isd_paperhead_backward_stage: stage=%d not in [0,5)", stage);
  ISD_CHECK_ARG(world >= 1 && world <= 65536, "isd_paperhead_backward_stage: world=%d", world);
  return ph_backward_stage(p, stage, x, params, dout, dparams, nullptr, (float*)workspace, B, 1, world,
                           (hipStream_t)stream);
}

extern "C" int isd_paperhead_backward(const isd_paperhead_plan* p, const float* x, const float* params,
                                      const float* dout, float* dparams, void* workspace, int64_t B, void* stream) {
  return ph_backward_impl(p, x, params, dout, dparams, nullptr, workspace, B, 1, stream);
}

// The same plus dx [B][C][T].  training: 1 = the forward used batch statistics, 0 = running statistics (eval mode: what
// attribution methods differentiate; every activation the backward needs is kept by the forward in both modes).
extern "C" int isd_paperhead_backward_x(const isd_paperhead_plan* p, const float* x, const float* params,
                                        const float* dout, float* dparams, float* dx, void* workspace, int64_t B,
                                        int training, void* stream) {
  ISD_CHECK_ARG(dx, "isd_paperhead_backward_x: null dx");
  return ph_backward_impl(p, x, params, dout, dparams, dx, workspace, B, training ? 1 : 0, stream);
}
