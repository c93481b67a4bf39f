// Dense head and loss (reference: FAST.input_layer / last_layer / 'train_head' branch,
// src/fast/models/fast.py:235,239,273-278; nn.CrossEntropyLoss at src/fast/train/trainer.py:37,59;
// argmax predict at trainer.py:89).
//
// The linear layers run on v_mfma_f32_16x16x4_f32: M = output features (A = weight rows),
// N = 16 samples per wave (B = activation rows), K = input features in steps of 4, so the
// accumulator of a lane is 4 consecutive output features of one sample -> float4 stores.
#include "common.h"
#include "zonebatch.h"
#include <math.h>

namespace isd {

// every launch of this file goes through zone_launch: issued at once, or recorded for a zone-batched launch (zonebatch.h)
#define ISD_ZLAUNCH(...)                                    \
  do {                                                      \
    if (!zone_launch(__VA_ARGS__)) return ISD_ERR_INVALID;  \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kKC = 64;      // K chunk staged in LDS
constexpr int kRS = kKC + 2; // row stride: lanes (l&15) walk rows, == 2 (mod 32) is conflict free

__device__ __forceinline__ float gelu1(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu1_grad(float x) {
  return 0.5f * (1.f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * expf(-0.5f * x * x);
}

// y[m][o] = act( sum_i x[m][i] * W(o,i) + bias[o] ),  W(o,i) = w[o*so + i*si]
// grid: (ceil(M/64), ceil(Nout/64)); block 256 = 4 waves x 16 samples; up to 4 output tiles per wave.
__device__ __forceinline__ void linear_fwd_kernel_body(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ y,
                                                         float* __restrict__ pre, int64_t M, int K, int Nout,
                                                         int64_t so, int64_t si, int act,
                                                         const float* __restrict__ res,
    unsigned zgx, unsigned zgy, unsigned zbz, unsigned zgz) {   // this zone's own gridDim.x/.y, blockIdx.z, gridDim.z
  __shared__ float xs[64 * kRS];
  __shared__ float wsm[64 * kRS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t m0 = (int64_t)blockIdx.x * 64;
  const int o0 = blockIdx.y * 64;
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // The next K chunk is fetched into registers while the matrix cores work on the current one (and every load of a
  // chunk goes out behind clamped addresses, not bounds tests): with the fetch at the top of each chunk a
  // [320 x 256] x [256 x 32] product was four exposed memory latencies long, 34 us on 5 workgroups.
  constexpr int NL = 64 * kKC / 256;
  auto fetch = [&](int k0, float (&xv)[NL], float (&wv)[NL]) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int e = threadIdx.x + 256 * i;
      const int r = e / kKC, c = e - r * kKC;
      const int64_t m = m0 + r;
      const int k = k0 + c;
      const int kc = k < K ? k : K - 1;
      const int o = o0 + r;
      xv[i] = x[(m < M ? m : M - 1) * K + kc];
      wv[i] = w[(int64_t)(o < Nout ? o : Nout - 1) * so + kc * si];
    }
  };
  auto chunk = [&](int k0, const float (&xv)[NL], const float (&wv)[NL]) {
    __syncthreads();                                    // the previous chunk's readers are done
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int e = threadIdx.x + 256 * i;
      const int r = e / kKC, c = e - r * kKC;
      const bool kin = k0 + c < K;
      xs[r * kRS + c] = (m0 + r < M && kin) ? xv[i] : 0.f;
      wsm[r * kRS + c] = (o0 + r < Nout && kin) ? wv[i] : 0.f;
    }
    __syncthreads();
  };
  auto mma = [&]() {
    const float* xr = xs + (wave * 16 + (lane & 15)) * kRS + (lane >> 4);
    const float* wr = wsm + (lane & 15) * kRS + (lane >> 4);
#pragma unroll 4
    for (int kk = 0; kk < kKC; kk += 4) {
      const float bf = xr[kk];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (o0 + t * 16 < Nout) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[t * 16 * kRS + kk], bf, acc[t], 0, 0, 0);
      }
    }
  };
  if (K <= 4 * kKC) {
    // Short reductions (the token projection Linear(256, 32) of FAST: [320 x 256] x [256 x 32] at the reference's
    // batch): ALL chunks are fetched before the first is used -- 128 registers, ONE exposed memory latency instead
    // of one per chunk (the kernel was 20 us whatever the batch: four latencies with a few MFMAs between them).
    float xa[4][NL], wa[4][NL];
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (c * kKC < K) fetch(c * kKC, xa[c], wa[c]);
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (c * kKC < K) {
        chunk(c * kKC, xa[c], wa[c]);
        mma();
      }
  } else {
    float xv[NL], wv[NL];
    fetch(0, xv, wv);
    for (int k0 = 0; k0 < K; k0 += kKC) {
      chunk(k0, xv, wv);
      if (k0 + kKC < K) fetch(k0 + kKC, xv, wv);
      mma();
    }
  }
  const int64_t m = m0 + wave * 16 + (lane & 15);
  if (m >= M) return;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = o0 + t * 16 + 4 * (lane >> 4) + r;
      if (o < Nout) {
        float v = acc[t][r] + (bias ? bias[o] : 0.f);
        if (res) v += res[m * Nout + o];
        if (pre) pre[m * Nout + o] = v;
        y[m * Nout + o] = act ? gelu1(v) : v;
      }
    }
  }
}
ISD_ZONE_FN(linear_fwd_kernel, 256)
__global__ __launch_bounds__(256) void linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ y,
                                                         float* __restrict__ pre, int64_t M, int K, int Nout,
                                                         int64_t so, int64_t si, int act,
                                                         const float* __restrict__ res) {
  linear_fwd_kernel_body(x, w, bias, y, pre, M, K, Nout, so, si, act, res,
      gridDim.x, gridDim.y, blockIdx.z, gridDim.z);
}
ISD_ZONE_REGISTER(linear_fwd_kernel)

// dpre = dy * gelu'(pre)   (act) or a plain copy
__device__ __forceinline__ void act_bwd_kernel_body(const float* __restrict__ dy, const float* __restrict__ pre, float* __restrict__ dpre,
                               int64_t n, int act,
    unsigned zgx, unsigned zgy, unsigned zbz, unsigned zgz) {   // this zone's own gridDim.x/.y, blockIdx.z, gridDim.z
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)zgx * blockDim.x)
    dpre[e] = act ? dy[e] * gelu1_grad(pre[e]) : dy[e];
}
ISD_ZONE_FN(act_bwd_kernel, 1024)
__global__ void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ pre, float* __restrict__ dpre,
                               int64_t n, int act) {
  act_bwd_kernel_body(dy, pre, dpre, n, act,
      gridDim.x, gridDim.y, blockIdx.z, gridDim.z);
}
ISD_ZONE_REGISTER(act_bwd_kernel)

// dW[o][i] = sum_m dpre[m][o] * x[m][i];  column i == K is the bias gradient (x == 1).
// M = o (A = dpre^T), N = i (16 per wave), K = samples.  grid: (slabs, ceil((K+1)/64)); partial slabs.
__device__ __forceinline__ void linear_wgrad_kernel_body(const float* __restrict__ dpre, const float* __restrict__ x,
                                                           float* __restrict__ part, int64_t M, int K, int Nout,
                                                           int m_per_wg,
    unsigned zgx, unsigned zgy, unsigned zbz, unsigned zgz) {   // this zone's own gridDim.x/.y, blockIdx.z, gridDim.z
  __shared__ float ds[64 * 80];   // [64 samples][Nout<=64], stride 80 == 16 (mod 32)
  __shared__ float xs[64 * 80];   // [64 samples][64 inputs]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i0 = blockIdx.y * 64;
  const int n0 = zbz * 64;                       // output-feature chunk
  const int nn = (Nout - n0) < 64 ? (Nout - n0) : 64;
  const int64_t m_lo = (int64_t)blockIdx.x * m_per_wg;
  const int64_t m_hi = (m_lo + m_per_wg) < M ? (m_lo + m_per_wg) : M;
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int64_t ms = m_lo; ms < m_hi; ms += 64) {
    __syncthreads();
#pragma unroll
    for (int e = threadIdx.x; e < 64 * 64; e += 256) {   // clamped addresses + selects: all 32 loads go out together
      const int r = e >> 6, c = e & 63;
      const int64_t m = ms + r;
      const bool ok = m < m_hi;
      const int64_t mc = ok ? m : m_hi - 1;
      const float dv = dpre[mc * Nout + n0 + (c < nn ? c : nn - 1)];
      ds[r * 80 + c] = (ok && c < nn) ? dv : 0.f;
      const int i = i0 + c;
      const float xv = x[mc * K + (i < K ? i : K - 1)];
      xs[r * 80 + c] = !ok ? 0.f : (i < K ? xv : (i == K ? 1.f : 0.f));
    }
    __syncthreads();
    const float* ar = ds + (lane >> 4) * 80 + (lane & 15);
    const float* br = xs + (lane >> 4) * 80 + wave * 16 + (lane & 15);
#pragma unroll 4
    for (int mm = 0; mm < 64; mm += 4) {
      const float bf = br[mm * 80];
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (t * 16 < nn) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[mm * 80 + t * 16], bf, acc[t], 0, 0, 0);
    }
  }
  const int i = i0 + wave * 16 + (lane & 15);
  if (i > K) return;
  float* slab = part + (int64_t)blockIdx.x * Nout * (K + 1);
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = t * 16 + 4 * (lane >> 4) + r;
      if (o < nn) slab[(int64_t)(n0 + o) * (K + 1) + i] = acc[t][r];
    }
}
ISD_ZONE_FN(linear_wgrad_kernel, 256)
__global__ __launch_bounds__(256) void linear_wgrad_kernel(const float* __restrict__ dpre, const float* __restrict__ x,
                                                           float* __restrict__ part, int64_t M, int K, int Nout,
                                                           int m_per_wg) {
  linear_wgrad_kernel_body(dpre, x, part, M, K, Nout, m_per_wg,
      gridDim.x, gridDim.y, blockIdx.z, gridDim.z);
}
ISD_ZONE_REGISTER(linear_wgrad_kernel)

// Slab sums in a fixed order.  blockIdx.y selects a run of L slabs (index k * stride); with gridDim.y > 1 the
// run's sum replaces its first slab and a second launch (stride = L) adds the run sums and scatters to dw / db.
__device__ __forceinline__ void linear_wgrad_reduce_kernel_body(float* __restrict__ part, float* __restrict__ dw, float* __restrict__ db,
                                           int K, int Nout, int n_slabs, int L, int stride,
    unsigned zgx, unsigned zgy, unsigned zbz, unsigned zgz) {   // this zone's own gridDim.x/.y, blockIdx.z, gridDim.z
  const int n = Nout * (K + 1);
  const int k0 = blockIdx.y * L;
  const int k1 = k0 + L < n_slabs ? k0 + L : n_slabs;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += zgx * blockDim.x) {
    float s0 = 0.f, s1 = 0.f;
    int k = k0;
    for (; k + 1 < k1; k += 2) {
      s0 += part[(int64_t)k * stride * n + e];
      s1 += part[(int64_t)(k + 1) * stride * n + e];
    }
    if (k < k1) s0 += part[(int64_t)k * stride * n + e];
    const float s = s0 + s1;
    if (zgy > 1) {
      part[(int64_t)k0 * stride * n + e] = s;
      continue;
    }
    const int o = e / (K + 1), i = e - o * (K + 1);
    if (i < K) dw[o * K + i] = s;
    else if (db) db[o] = s;
  }
}
ISD_ZONE_FN(linear_wgrad_reduce_kernel, 1024)
__global__ void linear_wgrad_reduce_kernel(float* __restrict__ part, float* __restrict__ dw, float* __restrict__ db,
                                           int K, int Nout, int n_slabs, int L, int stride) {
  linear_wgrad_reduce_kernel_body(part, dw, db, K, Nout, n_slabs, L, stride,
      gridDim.x, gridDim.y, blockIdx.z, gridDim.z);
}
ISD_ZONE_REGISTER(linear_wgrad_reduce_kernel)

// Mean over tokens, softmax cross-entropy (mean over the global batch), gradient and argmax.
// One thread per trial; per-block partial loss sums go to `part` and the LAST block (agent-scope
// ticket) adds them in block order, so the loss is deterministic without a second launch.
__device__ __forceinline__ void softmax_ce_kernel_body(const float* __restrict__ lt, const void* __restrict__ labels,
                                                         int label_bytes, float* __restrict__ lmean,
                                                         float* __restrict__ loss, float* __restrict__ dlt,
                                                         int64_t* __restrict__ pred, int64_t B, int n_tok, int n_cls,
                                                         float grad_scale, float* __restrict__ part,
                                                         unsigned int* __restrict__ ticket,
    unsigned zgx, unsigned zgy, unsigned zbz, unsigned zgz) {   // this zone's own gridDim.x/.y, blockIdx.z, gridDim.z
  __shared__ float red[256];
  __shared__ bool last;
  float lsum = 0.f;
  const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (b < B) {
    float v[32];
    float mx = -INFINITY;
    int am = 0;
#pragma unroll 1
    for (int c = 0; c < n_cls; ++c) {
      float s = 0.f;
      for (int n = 0; n < n_tok; ++n) s += lt[(b * n_tok + n) * n_cls + c];
      s /= (float)n_tok;
      v[c] = s;
      if (lmean) lmean[b * n_cls + c] = s;
      if (s > mx) { mx = s; am = c; }          // strict '>' keeps the lowest index on ties (torch.argmax)
    }
    if (pred) pred[b] = am;
    if (labels) {
      float se = 0.f;
      for (int c = 0; c < n_cls; ++c) se += expf(v[c] - mx);
      const float lse = mx + logf(se);
      const int64_t yb = label_bytes == 1 ? (int64_t)((const unsigned char*)labels)[b] : ((const int64_t*)labels)[b];
      lsum = lse - v[yb];
      if (dlt) {
        for (int c = 0; c < n_cls; ++c) {
          const float g = (expf(v[c] - lse) - (c == yb ? 1.f : 0.f)) * grad_scale / (float)n_tok;
          for (int n = 0; n < n_tok; ++n) dlt[(b * n_tok + n) * n_cls + c] = g;
        }
      }
    }
  }
  if (!loss) return;
  red[threadIdx.x] = lsum;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0 && zgx == 1) {                   // one workgroup (B <= 256): nothing to combine, no ticket
    *loss = red[0] * grad_scale;
  } else if (threadIdx.x == 0) {
    __hip_atomic_store(&part[blockIdx.x], red[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last = (t == zgx - 1);
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      float tot = 0.f;
      for (unsigned int k = 0; k < zgx; ++k)
        tot += __hip_atomic_load(&part[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *loss = tot * grad_scale;
    }
  }
}
ISD_ZONE_FN(softmax_ce_kernel, 256)
__global__ __launch_bounds__(256) void softmax_ce_kernel(const float* __restrict__ lt, const void* __restrict__ labels,
                                                         int label_bytes, float* __restrict__ lmean,
                                                         float* __restrict__ loss, float* __restrict__ dlt,
                                                         int64_t* __restrict__ pred, int64_t B, int n_tok, int n_cls,
                                                         float grad_scale, float* __restrict__ part,
                                                         unsigned int* __restrict__ ticket) {
  softmax_ce_kernel_body(lt, labels, label_bytes, lmean, loss, dlt, pred, B, n_tok, n_cls, grad_scale, part, ticket,
      gridDim.x, gridDim.y, blockIdx.z, gridDim.z);
}
ISD_ZONE_REGISTER(softmax_ce_kernel)

}  // namespace isd

using namespace isd;

static int launch_linear(const float* x, const float* w, const float* bias, float* y, float* pre, int64_t M, int K,
                         int Nout, int64_t so, int64_t si, int act, hipStream_t st, const float* res = nullptr) {
  const int64_t gx = cdiv(M, 64);
  ISD_CHECK_ARG(gx <= 0x7fffffffLL, "linear: M too large");
  ISD_ZLAUNCH(linear_fwd_kernel, dim3((unsigned)gx, (unsigned)cdiv(Nout, 64)), dim3(256), 0, st, x, w, bias, y,
                     pre, M, K, Nout, so, si, act, res);
  ISD_LAUNCH_CHECK();
  return ISD_OK;
}

extern "C" int isd_linear_residual_forward(const float* x, const float* w, const float* bias, const float* res,
                                           float* y, int64_t M, int K, int N, void* stream) {
  ISD_CHECK_ARG(M >= 0 && K >= 1 && N >= 1 && K <= (1 << 20) && N <= (1 << 20), "isd_linear_residual_forward: bad shape");
  if (M == 0) return ISD_OK;
  ISD_CHECK_ARG(x && w && y && res, "isd_linear_residual_forward: null argument");
  return launch_linear(x, w, bias, y, nullptr, M, K, N, K, 1, 0, (hipStream_t)stream, res);
}

extern "C" int isd_linear_forward(const float* x, const float* w, const float* bias, float* y, float* pre, int64_t M,
                                  int K, int N, int act, void* stream) {
  ISD_CHECK_ARG(M >= 0 && K >= 1 && N >= 1 && K <= (1 << 20) && N <= (1 << 20), "isd_linear_forward: bad shape M=%lld K=%d N=%d",
                (long long)M, K, N);
  ISD_CHECK_ARG(act == 0 || act == 1, "isd_linear_forward: act must be 0 (none) or 1 (gelu)");
  if (M == 0) return ISD_OK;
  ISD_CHECK_ARG(x && w && y, "isd_linear_forward: null argument");
  return launch_linear(x, w, bias, y, pre, M, K, N, K, 1, act, (hipStream_t)stream);
}

static int wgrad_slabs(int64_t M, int* m_per_wg) {
  int64_t want = 256;
  int64_t per = cdiv(cdiv(M, want), 64) * 64;
  if (per < 64) per = 64;
  *m_per_wg = (int)per;
  return (int)cdiv(M, per);
}

extern "C" int64_t isd_linear_workspace_bytes(int64_t M, int K, int N) {
  if (M < 0 || K < 1 || N < 1) return ISD_ERR_INVALID;
  int mp;
  const int slabs = wgrad_slabs(M > 0 ? M : 1, &mp);
  return 4 * (M * N + (int64_t)slabs * N * (K + 1)) + 256;
}

extern "C" int isd_linear_backward(const float* x, const float* w, const float* dy, const float* pre, float* dx,
                                   float* dw, float* db, void* workspace, int64_t M, int K, int N, int act,
                                   void* stream) {
  ISD_CHECK_ARG(M >= 0 && K >= 1 && N >= 1, "isd_linear_backward: bad shape");
  ISD_CHECK_ARG(act == 0 || (act == 1 && pre), "isd_linear_backward: gelu needs the saved pre-activation");
  ISD_CHECK_ARG(dw, "isd_linear_backward: null dw");
  hipStream_t st = (hipStream_t)stream;
  if (M == 0) {
    ISD_HIP_TRY(zone_clear(dw, sizeof(float) * N * K, st));
    if (db) ISD_HIP_TRY(zone_clear(db, sizeof(float) * N, st));
    return ISD_OK;
  }
  ISD_CHECK_ARG(x && w && dy && workspace, "isd_linear_backward: null argument");
  float* dpre = (float*)workspace;
  float* part = dpre + ((M * N + 63) / 64) * 64;
  const float* dsrc = dy;
  if (act) {
    ISD_ZLAUNCH(act_bwd_kernel, dim3((unsigned)(cdiv(M * N, 256) < 4096 ? cdiv(M * N, 256) : 4096)), dim3(256), 0,
                       st, dy, pre, dpre, M * N, act);
    ISD_LAUNCH_CHECK();
    dsrc = dpre;
  }
  if (dx) {   // dx[m][i] = sum_o dpre[m][o] W[o][i]  ==  linear with W'(out=i, in=o) = w[o*K + i]
    int rc = launch_linear(dsrc, w, nullptr, dx, nullptr, M, N, K, 1, K, 0, st);
    if (rc) return rc;
  }
  int mp;
  const int slabs = wgrad_slabs(M, &mp);
  ISD_ZLAUNCH(linear_wgrad_kernel, dim3(slabs, (unsigned)cdiv(K + 1, 64), (unsigned)cdiv(N, 64)), dim3(256), 0,
                     st, dsrc, x, part, M, K, N, mp);
  {
    const unsigned bx = (unsigned)cdiv((int64_t)N * (K + 1), 256);
    if (slabs >= 64 && bx < 256) {
      int L = 8;
      while (L * L < slabs) L *= 2;
      const int S = (int)cdiv(slabs, L);
      ISD_ZLAUNCH(linear_wgrad_reduce_kernel, dim3(bx, S), dim3(256), 0, st, part, dw, db, K, N, slabs, L, 1);
      ISD_ZLAUNCH(linear_wgrad_reduce_kernel, dim3(bx, 1), dim3(256), 0, st, part, dw, db, K, N, S, S, L);
    } else {
      ISD_ZLAUNCH(linear_wgrad_reduce_kernel, dim3(bx, 1), dim3(256), 0, st, part, dw, db, K, N, slabs, slabs, 1);
    }
  }
  ISD_LAUNCH_CHECK();
  return ISD_OK;
}

extern "C" int64_t isd_softmax_ce_workspace_bytes(int64_t B) {
  if (B < 0) return ISD_ERR_INVALID;
  return 256 + 4 * cdiv(B > 0 ? B : 1, 256);
}

extern "C" int isd_softmax_ce(const float* logits_tok, const void* labels, int label_bytes, float* logits_mean,
                              float* loss, float* dlogits_tok, int64_t* pred, int64_t B, int n_tok, int n_cls,
                              float grad_scale, void* workspace, void* stream) {
  ISD_CHECK_ARG(B >= 0 && n_tok >= 1 && n_cls >= 1 && n_cls <= 32, "isd_softmax_ce: bad shape B=%lld n_tok=%d n_cls=%d",
                (long long)B, n_tok, n_cls);
  ISD_CHECK_ARG(!labels || label_bytes == 1 || label_bytes == 8, "isd_softmax_ce: labels must be uint8 or int64");
  ISD_CHECK_ARG(B == 0 || logits_tok, "isd_softmax_ce: null logits");
  ISD_CHECK_ARG(cdiv(B, 256) <= 0x7fffffffLL, "isd_softmax_ce: B too large");
  hipStream_t st = (hipStream_t)stream;
  const bool want_loss = labels && loss;
  ISD_CHECK_ARG(!want_loss || workspace, "isd_softmax_ce: the loss reduction needs isd_softmax_ce_workspace_bytes(B) bytes");
  if (B == 0) {
    if (want_loss) ISD_HIP_TRY(zone_clear(loss, sizeof(float), st));
    return ISD_OK;
  }
  float* scratch = want_loss ? (float*)workspace : nullptr;
  // arrival ticket of the cross-workgroup loss sum (a single workgroup needs none: at the reference's batch of 64 the
  // clear was a 5-us node of the replayed step)
  if (scratch && cdiv(B, 256) > 1) ISD_HIP_TRY(zone_clear(scratch, sizeof(unsigned int), st));
  ISD_ZLAUNCH(softmax_ce_kernel, dim3((unsigned)cdiv(B, 256)), dim3(256), 0, st, logits_tok, labels, label_bytes,
                     logits_mean, want_loss ? loss : nullptr, labels ? dlogits_tok : nullptr, pred, B, n_tok, n_cls,
                     grad_scale, scratch ? scratch + 64 : nullptr, reinterpret_cast<unsigned int*>(scratch));
  ISD_LAUNCH_CHECK();
  return ISD_OK;
}
