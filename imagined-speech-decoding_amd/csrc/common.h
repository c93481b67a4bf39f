// Shared host/device helpers for libisd_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/isd_hip.h"

namespace isd {

void set_error(const char* fmt, ...);

#define ISD_CHECK_ARG(cond, ...)                 \
  do {                                           \
    if (!(cond)) {                               \
      isd::set_error(__VA_ARGS__);               \
      return ISD_ERR_INVALID;                    \
    }                                            \
  } while (0)

#define ISD_HIP_TRY(expr)                                                         \
  do {                                                                            \
    hipError_t _e = (expr);                                                       \
    if (_e != hipSuccess) {                                                       \
      isd::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),       \
                     __FILE__, __LINE__);                                         \
      return ISD_ERR_HIP;                                                         \
    }                                                                             \
  } while (0)

// checks the launch that was just enqueued (no host sync)
#define ISD_LAUNCH_CHECK() ISD_HIP_TRY(hipGetLastError())

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ------------------------------------------------------------------ device side
#if defined(__HIPCC__)

// DPP row shift right by N inside each 16-lane row; lanes with no source get 0 (bound_ctrl: no `old` operand to
// initialise, and the shift can fold into a consuming VOP2).
template <int N>
__device__ __forceinline__ float row_shr(float v) {
  int r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x110 + N, 0xf, 0xf, true);
  return __int_as_float(r);
}
template <int N>
__device__ __forceinline__ double row_shr(double v) {
  long long b = __double_as_longlong(v);
  int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffLL), 0x110 + N, 0xf, 0xf, true);
  int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x110 + N, 0xf, 0xf, true);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// DPP row shift left by N (lane i reads lane i+N of its 16-lane row; 0 past the end).
template <int N>
__device__ __forceinline__ float row_shl(float v) {
  int r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x100 + N, 0xf, 0xf, true);
  return __int_as_float(r);
}
// DPP rotate right by N inside the 16-lane row (lane i reads lane (i - N) mod 16): dpp_ctrl row_ror:N
template <int N>
__device__ __forceinline__ float row_ror(float v) {
  int r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + N, 0xf, 0xf, true);
  return __int_as_float(r);
}
// DPP shift right by one lane across the WHOLE wave (gfx9 wave_shr:1; lane 0 gets 0): joins neighbouring chunks of a
// row that spans two or four 16-lane DPP rows
__device__ __forceinline__ float wave_shr1(float v) {
  int r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true);
  return __int_as_float(r);
}
// value of lane `src` (wave-uniform index) broadcast to every lane
__device__ __forceinline__ double read_lane(double v, int src) {
  long long b = __double_as_longlong(v);
  int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), src);
  int hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// LDS hand-over between the lanes of ONE wave (workgroups of 64 threads): LDS operations of a wave execute in
// order, so it is enough to wait for the outstanding LDS operations and to keep the compiler from moving accesses
// across this point.  Unlike __syncthreads() this does not drain vmcnt: global stores and loads stay in flight.
// GELU (erf form, nn.GELU's default) and its derivative through  erfc(|z|) = t (a1 + t (a2 + ...)) exp(-z^2),
// t = 1 / (1 + p |z|)  (Abramowitz & Stegun 7.1.26, absolute error 1.5e-7 -- the fp32 rounding level of the
// activation).  libm's erff is ~100 instructions per element and, where an activation tile ends in GELU and GELU',
// was the larger part of the kernel's vector work.  cdf = 0.5 (1 + erf(x / sqrt 2)),  ez = exp(-x^2 / 2).
__device__ __forceinline__ void gelu_parts(float x, float& cdf, float& ez) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.f));
  ez = __expf(-z * z);
  const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f),
                              0.254829592f);
  const float c = 0.5f * poly * ez;                      // 0.5 erfc(|z|)
  cdf = x < 0.f ? c : 1.f - c;
}
__device__ __forceinline__ float gelu_fast(float x) {
  float cdf, ez;
  gelu_parts(x, cdf, ez);
  return x * cdf;
}
__device__ __forceinline__ float gelu_grad_fast(float x) {
  float cdf, ez;
  gelu_parts(x, cdf, ez);
  return fmaf(x * 0.39894228040143267794f, ez, cdf);
}

__device__ __forceinline__ void wave_lds_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

#endif  // __HIPCC__
}  // namespace isd
