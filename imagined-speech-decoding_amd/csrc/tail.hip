// Transformer tail of FAST (reference: src/fast/models/fast.py:10-29 `AttentionBlock`,
// :260-268 `forward_transformer`) -- the small pieces around the MFMA linear layers of fc.hip:
// cls-token / positional embedding, LayerNorm, and the attention core (<= 8 tokens, head_dim <= 8).
// The dense projections (in_proj, out_proj, MLP) are isd_linear_forward/backward launches over the
// [B*S, D] token matrix.
#include "common.h"
#include <math.h>

namespace isd {

constexpr int kMaxS = 8, kMaxDh = 8;

__device__ __forceinline__ float tail_drop_scale(uint64_t seed, uint64_t idx, float p) {
  if (p <= 0.f) return 1.f;
  uint64_t v = (idx + 0x9E3779B97F4A7C15ull) ^ seed;
  v ^= v >> 30; v *= 0xBF58476D1CE4E5B9ull;
  v ^= v >> 27; v *= 0x94D049BB133111EBull;
  v ^= v >> 31;
  const float uu = (float)(v >> 40) * (1.f / 16777216.f);
  return uu >= p ? 1.f / (1.f - p) : 0.f;
}

// tok[b,0,:] = cls + pos[0];  tok[b,1+n,:] = x[b,n,:] + pos[1+n]          (fast.py:263-265)
__global__ void embed_fwd_kernel(const float* __restrict__ x, const float* __restrict__ cls,
                                 const float* __restrict__ pos, float* __restrict__ tok, int64_t B, int N, int D) {
  const int S = N + 1;
  const int64_t n = B * S * D;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int d = (int)(e % D);
    const int s = (int)((e / D) % S);
    const int64_t b = e / ((int64_t)D * S);
    const float base = s == 0 ? cls[d] : x[(b * N + (s - 1)) * D + d];
    tok[e] = base + pos[s * D + d];
  }
}

// dx[b,n,:] = dtok[b,1+n,:];  dpos[s,:] = sum_b dtok[b,s,:];  dcls = sum_b dtok[b,0,:].  One block per (s, d).
__global__ __launch_bounds__(256) void embed_bwd_kernel(const float* __restrict__ dtok, float* __restrict__ dx,
                                                        float* __restrict__ dcls, float* __restrict__ dpos, int64_t B,
                                                        int N, int D) {
  __shared__ float red[256];
  const int S = N + 1;
  const int s = blockIdx.x / D, d = blockIdx.x - s * D;
  float acc = 0.f;
  for (int64_t b = threadIdx.x; b < B; b += 256) {
    const float g = dtok[(b * S + s) * D + d];
    acc += g;
    if (s > 0) dx[(b * N + (s - 1)) * D + d] = g;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    dpos[s * D + d] = red[0];
    if (s == 0) dcls[d] = red[0];
  }
}

// LayerNorm over the last dim (D <= 64), eps inside the sqrt, biased variance (nn.LayerNorm).  One wave per row.
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ b, float* __restrict__ y,
                                                            float* __restrict__ stats, int64_t M, int D, float eps) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= M) return;
  const float v = lane < D ? x[row * D + lane] : 0.f;
  float s = v;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mu = s / (float)D;
  const float c = lane < D ? v - mu : 0.f;
  float q = c * c;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  const float rstd = rsqrtf(q / (float)D + eps);
  if (lane < D) y[row * D + lane] = c * rstd * w[lane] + b[lane];
  if (lane == 0) {
    stats[row * 2] = mu;
    stats[row * 2 + 1] = rstd;
  }
}

// dx = rstd (g - mean(g) - xhat mean(g xhat)),  g = dy * w
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ dy,
                                                            const float* __restrict__ stats, float* __restrict__ dx,
                                                            int64_t M, int D) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= M) return;
  const float mu = stats[row * 2], rstd = stats[row * 2 + 1];
  const bool in = lane < D;
  const float xh = in ? (x[row * D + lane] - mu) * rstd : 0.f;
  const float g = in ? dy[row * D + lane] * w[lane] : 0.f;
  float s1 = g, s2 = g * xh;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s1 += __shfl_xor(s1, o, 64);
    s2 += __shfl_xor(s2, o, 64);
  }
  if (in) dx[row * D + lane] = rstd * (g - s1 / (float)D - xh * s2 / (float)D);
}

// dw[d] = sum_m dy[m,d] xhat[m,d];  db[d] = sum_m dy[m,d].  One block per column d.
__global__ __launch_bounds__(256) void layernorm_wgrad_kernel(const float* __restrict__ x,
                                                              const float* __restrict__ dy,
                                                              const float* __restrict__ stats, float* __restrict__ dw,
                                                              float* __restrict__ db, int64_t M, int D) {
  __shared__ float r1[256], r2[256];
  const int d = blockIdx.x;
  float a = 0.f, c = 0.f;
  for (int64_t m = threadIdx.x; m < M; m += 256) {
    const float g = dy[m * D + d];
    a += g * (x[m * D + d] - stats[m * 2]) * stats[m * 2 + 1];
    c += g;
  }
  r1[threadIdx.x] = a;
  r2[threadIdx.x] = c;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) {
      r1[threadIdx.x] += r1[threadIdx.x + w];
      r2[threadIdx.x] += r2[threadIdx.x + w];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    dw[d] = r1[0];
    db[d] = r2[0];
  }
}

// Attention core: qkv [B,S,3D] (q | k | v, heads contiguous inside each) -> ctx [B,S,D]; probs [B,H,S,S] saved
// (after dropout scaling is NOT applied to the saved probs; the mask is regenerated in backward).
// One thread per (b, h, query).
__global__ void attn_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ ctx, float* __restrict__ probs,
                                int64_t B, int S, int H, int dh, float scale, float dropout_p, uint64_t seed) {
  const int D = H * dh;
  const int64_t n = B * H * S;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(e % S);
    const int h = (int)((e / S) % H);
    const int64_t b = e / ((int64_t)S * H);
    const float* q = qkv + (b * S + i) * 3 * D + h * dh;
    float sc[kMaxS];
    float mx = -INFINITY;
    for (int j = 0; j < S; ++j) {
      const float* k = qkv + (b * S + j) * 3 * D + D + h * dh;
      float a = 0.f;
      for (int t = 0; t < dh; ++t) a = fmaf(q[t], k[t], a);
      sc[j] = a * scale;
      mx = fmaxf(mx, sc[j]);
    }
    float den = 0.f;
    for (int j = 0; j < S; ++j) {
      sc[j] = expf(sc[j] - mx);
      den += sc[j];
    }
    float o[kMaxDh];
    for (int t = 0; t < dh; ++t) o[t] = 0.f;
    for (int j = 0; j < S; ++j) {
      const float p = sc[j] / den;
      probs[((b * H + h) * S + i) * S + j] = p;
      const float pd = p * tail_drop_scale(seed, (uint64_t)(((b * H + h) * S + i) * S + j), dropout_p);
      const float* v = qkv + (b * S + j) * 3 * D + 2 * D + h * dh;
      for (int t = 0; t < dh; ++t) o[t] = fmaf(pd, v[t], o[t]);
    }
    for (int t = 0; t < dh; ++t) ctx[(b * S + i) * D + h * dh + t] = o[t];
  }
}

// One thread per (b, h): dqkv from dctx, probs, qkv.
__global__ void attn_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ probs,
                                const float* __restrict__ dctx, float* __restrict__ dqkv, int64_t B, int S, int H,
                                int dh, float scale, float dropout_p, uint64_t seed) {
  const int D = H * dh;
  const int64_t n = B * H;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int h = (int)(e % H);
    const int64_t b = e / H;
    float dk[kMaxS][kMaxDh], dv[kMaxS][kMaxDh];
    for (int j = 0; j < S; ++j)
      for (int t = 0; t < dh; ++t) dk[j][t] = dv[j][t] = 0.f;
    for (int i = 0; i < S; ++i) {
      const float* dc = dctx + (b * S + i) * D + h * dh;
      const float* q = qkv + (b * S + i) * 3 * D + h * dh;
      float dp[kMaxS], p[kMaxS];
      float dot = 0.f;
      for (int j = 0; j < S; ++j) {
        const int64_t pi = ((b * H + h) * S + i) * S + j;
        const float m = tail_drop_scale(seed, (uint64_t)pi, dropout_p);
        p[j] = probs[pi];
        const float* v = qkv + (b * S + j) * 3 * D + 2 * D + h * dh;
        float a = 0.f;
        for (int t = 0; t < dh; ++t) {
          a = fmaf(dc[t], v[t], a);
          dv[j][t] = fmaf(p[j] * m, dc[t], dv[j][t]);
        }
        dp[j] = a * m;                                  // gradient w.r.t. the (pre-dropout) probability
        dot = fmaf(p[j], dp[j], dot);
      }
      float dq[kMaxDh];
      for (int t = 0; t < dh; ++t) dq[t] = 0.f;
      for (int j = 0; j < S; ++j) {
        const float ds = p[j] * (dp[j] - dot) * scale;
        const float* k = qkv + (b * S + j) * 3 * D + D + h * dh;
        for (int t = 0; t < dh; ++t) {
          dq[t] = fmaf(ds, k[t], dq[t]);
          dk[j][t] = fmaf(ds, q[t], dk[j][t]);
        }
      }
      for (int t = 0; t < dh; ++t) dqkv[(b * S + i) * 3 * D + h * dh + t] = dq[t];
    }
    for (int j = 0; j < S; ++j)
      for (int t = 0; t < dh; ++t) {
        dqkv[(b * S + j) * 3 * D + D + h * dh + t] = dk[j][t];
        dqkv[(b * S + j) * 3 * D + 2 * D + h * dh + t] = dv[j][t];
      }
  }
}

}  // namespace isd

using namespace isd;

static unsigned grid_for(int64_t n) {
  const int64_t g = cdiv(n, 256);
  return (unsigned)(g < 1 ? 1 : (g > 65535 ? 65535 : g));
}

extern "C" int isd_embed_forward(const float* x, const float* cls, const float* pos, float* tok, int64_t B, int N,
                                 int D, void* stream) {
  ISD_CHECK_ARG(B >= 0 && N >= 0 && D >= 1, "isd_embed_forward: bad shape");
  if (B == 0) return ISD_OK;
  ISD_CHECK_ARG(cls && pos && tok && (N == 0 || x), "isd_embed_forward: null argument");
  hipLaunchKernelGGL(embed_fwd_kernel, dim3(grid_for(B * (N + 1) * D)), dim3(256), 0, (hipStream_t)stream, x, cls, pos,
                     tok, B, N, D);
  ISD_LAUNCH_CHECK();
  return ISD_OK;
}

extern "C" int isd_embed_backward(const float* dtok, float* dx, float* dcls, float* dpos, int64_t B, int N, int D,
                                  void* stream) {
  ISD_CHECK_ARG(B >= 1 && N >= 0 && D >= 1, "isd_embed_backward: bad shape");
  ISD_CHECK_ARG(dtok && dcls && dpos && (N == 0 || dx), "isd_embed_backward: null argument");
  hipLaunchKernelGGL(embed_bwd_kernel, dim3((N + 1) * D), dim3(256), 0, (hipStream_t)stream, dtok, dx, dcls, dpos, B,
                     N, D);
  ISD_LAUNCH_CHECK();
  return ISD_OK;
}

extern "C" int isd_layernorm_forward(const float* x, const float* w, const float* b, float* y, float* stats, int64_t M,
                                     int D, float eps, void* stream) {
  ISD_CHECK_ARG(M >= 0 && D >= 1 && D <= 64, "isd_layernorm_forward: D=%d must be in [1,64]", D);
  if (M == 0) return ISD_OK;
  ISD_CHECK_ARG(x && w && b && y && stats, "isd_layernorm_forward: null argument");
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3((unsigned)cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, x, w, b, y,
                     stats, M, D, eps);
  ISD_LAUNCH_CHECK();
  return ISD_OK;
}

extern "C" int isd_layernorm_backward(const float* x, const float* w, const float* dy, const float* stats, float* dx,
                                      float* dw, float* db, int64_t M, int D, void* stream) {
  ISD_CHECK_ARG(M >= 1 && D >= 1 && D <= 64, "isd_layernorm_backward: bad shape");
  ISD_CHECK_ARG(x && w && dy && stats && dx && dw && db, "isd_layernorm_backward: null argument");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((unsigned)cdiv(M, 4)), dim3(256), 0, st, x, w, dy, stats, dx, M, D);
  hipLaunchKernelGGL(layernorm_wgrad_kernel, dim3(D), dim3(256), 0, st, x, dy, stats, dw, db, M, D);
  ISD_LAUNCH_CHECK();
  return ISD_OK;
}

extern "C" int isd_attention_forward(const float* qkv, float* ctx, float* probs, int64_t B, int S, int H, int head_dim,
                                     float dropout_p, uint64_t seed, void* stream) {
  ISD_CHECK_ARG(B >= 0 && S >= 1 && S <= kMaxS && H >= 1 && head_dim >= 1 && head_dim <= kMaxDh,
                "isd_attention_forward: S=%d (<=%d) H=%d head_dim=%d (<=%d)", S, kMaxS, H, head_dim, kMaxDh);
  ISD_CHECK_ARG(dropout_p >= 0.f && dropout_p < 1.f, "isd_attention_forward: dropout_p");
  if (B == 0) return ISD_OK;
  ISD_CHECK_ARG(qkv && ctx && probs, "isd_attention_forward: null argument");
  hipLaunchKernelGGL(attn_fwd_kernel, dim3(grid_for(B * H * S)), dim3(256), 0, (hipStream_t)stream, qkv, ctx, probs, B,
                     S, H, head_dim, 1.f / sqrtf((float)head_dim), dropout_p, seed);
  ISD_LAUNCH_CHECK();
  return ISD_OK;
}

extern "C" int isd_attention_backward(const float* qkv, const float* probs, const float* dctx, float* dqkv, int64_t B,
                                      int S, int H, int head_dim, float dropout_p, uint64_t seed, void* stream) {
  ISD_CHECK_ARG(B >= 1 && S >= 1 && S <= kMaxS && H >= 1 && head_dim >= 1 && head_dim <= kMaxDh,
                "isd_attention_backward: bad shape");
  ISD_CHECK_ARG(qkv && probs && dctx && dqkv, "isd_attention_backward: null argument");
  hipLaunchKernelGGL(attn_bwd_kernel, dim3(grid_for(B * H)), dim3(256), 0, (hipStream_t)stream, qkv, probs, dctx, dqkv,
                     B, S, H, head_dim, 1.f / sqrtf((float)head_dim), dropout_p, seed);
  ISD_LAUNCH_CHECK();
  return ISD_OK;
}
