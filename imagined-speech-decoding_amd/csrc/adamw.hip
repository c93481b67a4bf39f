// AdamW on the flat parameter block (the optimizer of the reference's training loop: optim.AdamW(lr=0.0005),
// src/fast/train/trainer.py:49; torch defaults betas (0.9, 0.999), eps 1e-8, weight_decay 1e-2, decoupled decay).
//
// The classifier keeps every parameter, gradient and both moment estimates as ONE contiguous fp32 block each
// (classifier.py Trainer), so the step is a single elementwise pass: 16 bytes read and 12 written per parameter,
// 0.6 M parameters at BASELINE config 2 -- a few microseconds of HBM/L2 time.  torch's multi-tensor kernel took
// 29 us for the same block (one workgroup per 64 K-element chunk of each tensor in its list).
//
// Arithmetic (same operation order as torch/optim/adamw.py's single-tensor path and its fused kernel):
//   p  <- p (1 - lr wd)
//   m  <- m + (g - m)(1 - b1)
//   v  <- b2 v + (1 - b2) g g
//   p  <- p - (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// lr and t may come from device memory (lr_dev / step_dev) so that a captured HIP graph can replay the step.
#include "common.h"
#include <math.h>

namespace isd {

struct AdamArgs {
  float* p;
  const float* g;
  float* m;
  float* v;
  int64_t n;
  float lr, b1, b2, eps, wd;
  float c1, c2;                     // 1 - b1, 1 - b2 rounded from double (torch's lerp weight)
  float bc1, bc2_sqrt;              // 1 - b1^t, sqrt(1 - b2^t) (host step count)
  const float* lr_dev;              // optional: learning rate in device memory
  const int64_t* step_dev;          // optional: step count t >= 1 in device memory
};

__device__ __forceinline__ void adamw_one(float& p, float g, float& m, float& v, float decay, float c1, float b2,
                                          float c2, float step_size, float bc2_sqrt, float eps) {
  p *= decay;
  m = fmaf(g - m, c1, m);
  v = fmaf(b2, v, c2 * g * g);
  const float denom = sqrtf(v) / bc2_sqrt + eps;
  p -= step_size * (m / denom);
}

__global__ __launch_bounds__(256) void adamw_kernel(AdamArgs a) {
  __shared__ float sc[3];
  float lr = a.lr, bc1 = a.bc1, bc2s = a.bc2_sqrt;
  if (a.lr_dev || a.step_dev) {                          // wave-uniform: one lane derives the scalars of this replay
    if (threadIdx.x == 0) {
      if (a.lr_dev) lr = a.lr_dev[0];
      if (a.step_dev) {
        const double t = (double)a.step_dev[0];
        bc1 = (float)(1.0 - pow((double)a.b1, t));
        bc2s = (float)sqrt(1.0 - pow((double)a.b2, t));
      }
      sc[0] = lr; sc[1] = bc1; sc[2] = bc2s;
    }
    __syncthreads();
    lr = sc[0]; bc1 = sc[1]; bc2s = sc[2];
  }
  const float decay = 1.f - lr * a.wd, c1 = a.c1, c2 = a.c2, step_size = lr / bc1;
  const int64_t i4 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 >= a.n) return;
  if (i4 + 4 <= a.n) {                                   // the four blocks are 16-byte aligned (checked by the host)
    float4 p = *reinterpret_cast<float4*>(a.p + i4);
    const float4 g = *reinterpret_cast<const float4*>(a.g + i4);
    float4 m = *reinterpret_cast<float4*>(a.m + i4);
    float4 v = *reinterpret_cast<float4*>(a.v + i4);
    adamw_one(p.x, g.x, m.x, v.x, decay, c1, a.b2, c2, step_size, bc2s, a.eps);
    adamw_one(p.y, g.y, m.y, v.y, decay, c1, a.b2, c2, step_size, bc2s, a.eps);
    adamw_one(p.z, g.z, m.z, v.z, decay, c1, a.b2, c2, step_size, bc2s, a.eps);
    adamw_one(p.w, g.w, m.w, v.w, decay, c1, a.b2, c2, step_size, bc2s, a.eps);
    *reinterpret_cast<float4*>(a.p + i4) = p;
    *reinterpret_cast<float4*>(a.m + i4) = m;
    *reinterpret_cast<float4*>(a.v + i4) = v;
  } else {
    for (int64_t i = i4; i < a.n; ++i) {
      float p = a.p[i], m = a.m[i], v = a.v[i];
      adamw_one(p, a.g[i], m, v, decay, c1, a.b2, c2, step_size, bc2s, a.eps);
      a.p[i] = p; a.m[i] = m; a.v[i] = v;
    }
  }
}

// ---- the same update over a LIST of tensors in one launch (the autograd modules of isd_amd.nn keep one
// nn.Parameter per reference tensor and autograd hands every gradient its own buffer).  The pointers travel in the
// kernel arguments (4 KiB limit: kAdamMaxTensors per launch), every tensor owns a run of 1024-element blocks, and a
// workgroup finds its tensor by bisection of the run ends.  With a device-side step counter the kernel advances it
// itself: every workgroup reads t = counter + 1 and the running powers b1^t, b2^t, and the last one to finish stores
// them back.
constexpr int kAdamMaxTensors = 96;
// One record per tensor, 40 bytes: a workgroup reads ITS tensor's pointers and length from one cache line of the
// argument block.  (The argument block of a launch lives in host-visible memory and is not cached across workgroups'
// CUs: as five parallel arrays -- p[], g[], m[], v[], n[] -- every workgroup paid five of those reads, and the
// bisection of the one-dimensional grid seven dependent ones: the kernel took 19 us for 0.2 M parameters.)
struct AdamTensor {
  float* p;
  const float* g;
  float* m;
  float* v;
  int n;
  unsigned bend;                    // end (exclusive) of the tensor's run of workgroups (one-dimensional grid)
};
struct AdamMultiArgs {
  int count, advance;               // advance: this launch is the last of the step (stores the counter)
  int two_d;                        // grid (max runs, count) instead of the runs back to back along x
  unsigned n_blocks;                // workgroups of this launch that own elements (the counter's target)
  float lr, b1, b2, eps, wd, c1, c2, bc1, bc2_sqrt;
  double b1d, b2d;
  const float* lr_dev;
  long long* step_dev;              // [0] steps taken so far t, [1] workgroups finished (returns to 0),
                                    // [2], [3] b1^t, b2^t as doubles (valid when t > 0)
  AdamTensor t[kAdamMaxTensors];
};

__global__ __launch_bounds__(256) void adamw_multi_kernel(AdamMultiArgs a) {
  __shared__ float sc[3];
  __shared__ double pw[2];
  // this workgroup's tensor and its run of 1024 elements.  two_d: blockIdx.y IS the tensor (workgroups past the
  // tensor's end leave at once); otherwise the tensors' runs lie back to back along x and the tensor is found by
  // bisection of the run ends -- seven dependent scalar loads from the argument block, which is why the two-dimensional
  // grid is used whenever it is not mostly empty.
  int lo;
  unsigned bx;
  if (a.two_d) {
    lo = blockIdx.y;
    bx = blockIdx.x;
    if ((int64_t)bx * 1024 >= a.t[lo].n) return;
  } else {
    lo = 0;
    int hi = a.count - 1;                                  // first tensor whose run ends past this workgroup
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (blockIdx.x < a.t[mid].bend) hi = mid; else lo = mid + 1;
    }
    bx = blockIdx.x - (lo ? a.t[lo - 1].bend : 0u);
  }
  const AdamTensor T = a.t[lo];
  float* __restrict__ P = T.p;
  const float* __restrict__ G = T.g;
  float* __restrict__ M = T.m;
  float* __restrict__ V = T.v;
  const int n = T.n;
  const int i4 = ((int)bx * 256 + (int)threadIdx.x) * 4;
  const bool vec = ((((uintptr_t)P | (uintptr_t)G | (uintptr_t)M | (uintptr_t)V) & 15) == 0);   // wave-uniform
  const bool full = i4 + 4 <= n && vec;
  // the data loads go out BEFORE the step scalars are waited for (two memory latencies side by side, not in a row)
  float4 p = {0.f, 0.f, 0.f, 0.f}, g = p, m = p, v = p;
  if (full) {
    p = *reinterpret_cast<float4*>(P + i4);
    g = *reinterpret_cast<const float4*>(G + i4);
    m = *reinterpret_cast<float4*>(M + i4);
    v = *reinterpret_cast<float4*>(V + i4);
  }
  float lr = a.lr, bc1 = a.bc1, bc2s = a.bc2_sqrt;
  if (a.lr_dev || a.step_dev) {
    if (threadIdx.x == 0) {
      if (a.lr_dev) lr = a.lr_dev[0];
      if (a.step_dev) {
        // b^t by recurrence from the powers the previous step left (a double-precision pow() per workgroup is
        // hundreds of instructions); a zeroed counter block (t = 0) starts from b^0 = 1
        const bool first = a.step_dev[0] == 0;
        const double p1 = (first ? 1.0 : __longlong_as_double(a.step_dev[2])) * a.b1d;
        const double p2 = (first ? 1.0 : __longlong_as_double(a.step_dev[3])) * a.b2d;
        pw[0] = p1; pw[1] = p2;
        bc1 = (float)(1.0 - p1);
        bc2s = (float)sqrt(1.0 - p2);
      }
      sc[0] = lr; sc[1] = bc1; sc[2] = bc2s;
    }
    __syncthreads();
    lr = sc[0]; bc1 = sc[1]; bc2s = sc[2];
  }
  const float decay = 1.f - lr * a.wd, step_size = lr / bc1;
  if (full) {
    adamw_one(p.x, g.x, m.x, v.x, decay, a.c1, a.b2, a.c2, step_size, bc2s, a.eps);
    adamw_one(p.y, g.y, m.y, v.y, decay, a.c1, a.b2, a.c2, step_size, bc2s, a.eps);
    adamw_one(p.z, g.z, m.z, v.z, decay, a.c1, a.b2, a.c2, step_size, bc2s, a.eps);
    adamw_one(p.w, g.w, m.w, v.w, decay, a.c1, a.b2, a.c2, step_size, bc2s, a.eps);
    *reinterpret_cast<float4*>(P + i4) = p;
    *reinterpret_cast<float4*>(M + i4) = m;
    *reinterpret_cast<float4*>(V + i4) = v;
  } else {
    for (int i = i4; i < n && i < i4 + 4; ++i) {
      float pp = P[i], mm = M[i], vv = V[i];
      adamw_one(pp, G[i], mm, vv, decay, a.c1, a.b2, a.c2, step_size, bc2s, a.eps);
      P[i] = pp; M[i] = mm; V[i] = vv;
    }
  }
  if (a.step_dev && a.advance) {                           // the last workgroup to get here advances the counter
    __syncthreads();
    if (threadIdx.x == 0) {
      const long long t = a.step_dev[0] + 1;
      __threadfence();
      const unsigned long long done = atomicAdd(reinterpret_cast<unsigned long long*>(a.step_dev + 1), 1ull);
      if (done + 1 == a.n_blocks) {
        a.step_dev[1] = 0;
        a.step_dev[2] = __double_as_longlong(pw[0]);
        a.step_dev[3] = __double_as_longlong(pw[1]);
        a.step_dev[0] = t;
      }
    }
  }
}

}  // namespace isd

extern "C" int isd_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                              double lr, double beta1, double beta2, double eps, double weight_decay, int64_t step,
                              const float* lr_dev, const int64_t* step_dev, void* stream) {
  ISD_CHECK_ARG(params && grads && exp_avg && exp_avg_sq, "isd_adamw_step: null argument");
  ISD_CHECK_ARG(n >= 0 && n <= (int64_t)1 << 40, "isd_adamw_step: n=%lld", (long long)n);
  ISD_CHECK_ARG(step_dev || step >= 1, "isd_adamw_step: step=%lld (the first step is 1)", (long long)step);
  ISD_CHECK_ARG(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0 && weight_decay >= 0.0,
                "isd_adamw_step: betas=(%g, %g) eps=%g weight_decay=%g", beta1, beta2, eps, weight_decay);
  ISD_CHECK_ARG((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0,
                "isd_adamw_step: the four blocks must be 16-byte aligned");
  if (n == 0) return ISD_OK;
  isd::AdamArgs a;
  a.p = params; a.g = grads; a.m = exp_avg; a.v = exp_avg_sq; a.n = n;
  a.lr = (float)lr; a.b1 = (float)beta1; a.b2 = (float)beta2; a.eps = (float)eps; a.wd = (float)weight_decay;
  a.c1 = (float)(1.0 - beta1); a.c2 = (float)(1.0 - beta2);
  const double t = (double)(step >= 1 ? step : 1);
  a.bc1 = (float)(1.0 - pow(beta1, t));
  a.bc2_sqrt = (float)sqrt(1.0 - pow(beta2, t));
  a.lr_dev = lr_dev; a.step_dev = step_dev;
  hipLaunchKernelGGL(isd::adamw_kernel, dim3((unsigned)isd::cdiv(n, 1024)), dim3(256), 0, (hipStream_t)stream, a);
  ISD_LAUNCH_CHECK();
  return ISD_OK;
}

extern "C" int isd_adamw_multi_step(int n_tensors, float* const* params, const float* const* grads,
                                    float* const* exp_avg, float* const* exp_avg_sq, const int64_t* numel, double lr,
                                    double beta1, double beta2, double eps, double weight_decay, int64_t step,
                                    const float* lr_dev, int64_t* step_dev, void* stream) {
  ISD_CHECK_ARG(n_tensors >= 0, "isd_adamw_multi_step: n_tensors=%d", n_tensors);
  ISD_CHECK_ARG(n_tensors == 0 || (params && grads && exp_avg && exp_avg_sq && numel), "isd_adamw_multi_step: null argument");
  ISD_CHECK_ARG(step_dev || step >= 1, "isd_adamw_multi_step: step=%lld (the first step is 1)", (long long)step);
  ISD_CHECK_ARG(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0 && weight_decay >= 0.0,
                "isd_adamw_multi_step: betas=(%g, %g) eps=%g weight_decay=%g", beta1, beta2, eps, weight_decay);
  ISD_CHECK_ARG(((uintptr_t)step_dev & 7) == 0, "isd_adamw_multi_step: step_dev must be 8-byte aligned");
  for (int i = 0; i < n_tensors; ++i) {
    ISD_CHECK_ARG(numel[i] >= 0 && numel[i] <= 0x7fffffffLL, "isd_adamw_multi_step: tensor %d has %lld elements", i,
                  (long long)numel[i]);
    ISD_CHECK_ARG(numel[i] == 0 || (params[i] && grads[i] && exp_avg[i] && exp_avg_sq[i]),
                  "isd_adamw_multi_step: tensor %d has a null pointer", i);
  }
  isd::AdamMultiArgs a;
  a.lr = (float)lr; a.b1 = (float)beta1; a.b2 = (float)beta2; a.eps = (float)eps; a.wd = (float)weight_decay;
  a.c1 = (float)(1.0 - beta1); a.c2 = (float)(1.0 - beta2);
  const double t = (double)(step >= 1 ? step : 1);
  a.bc1 = (float)(1.0 - pow(beta1, t));
  a.bc2_sqrt = (float)sqrt(1.0 - pow(beta2, t));
  a.b1d = beta1; a.b2d = beta2;
  a.lr_dev = lr_dev; a.step_dev = (long long*)step_dev;
  int last = -1;                                             // the launch that holds the last non-empty tensor
  for (int i = 0; i < n_tensors; ++i)
    if (numel[i] > 0) last = i;
  if (last < 0) return ISD_OK;
  int i = 0;
  while (i <= last) {
    a.count = 0;
    unsigned blocks = 0;
    for (; i <= last && a.count < isd::kAdamMaxTensors; ++i) {
      if (numel[i] == 0) continue;
      const int c = a.count++;
      a.t[c].p = params[i]; a.t[c].g = grads[i]; a.t[c].m = exp_avg[i]; a.t[c].v = exp_avg_sq[i];
      a.t[c].n = (int)numel[i];
      blocks += (unsigned)isd::cdiv(numel[i], 1024);
      a.t[c].bend = blocks;
    }
    a.advance = i > last ? 1 : 0;
    a.n_blocks = blocks;
    unsigned gx = 1;
    for (int c = 0; c < a.count; ++c) {
      const unsigned b = (unsigned)isd::cdiv(a.t[c].n, 1024);
      if (b > gx) gx = b;
    }
    a.two_d = (uint64_t)gx * a.count <= 16ull * blocks + 1024 ? 1 : 0;   // empty workgroups read one line and leave
    const dim3 grid = a.two_d ? dim3(gx, (unsigned)a.count) : dim3(blocks);
    hipLaunchKernelGGL(isd::adamw_multi_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
    ISD_LAUNCH_CHECK();
  }
  return ISD_OK;
}
