// isd_exact_sum: the order-independent accumulator of csrc/exact.h as a C-ABI primitive -- the exact sum of n fp32
// values, rounded once to fp64, whatever the grid or the arrival order.  The BatchNorm heads use the accumulator inline;
// this entry point exists so that its arithmetic (carries across digits, negative totals, subnormals, cancellation,
// non-finite inputs) is tested in isolation against an exact host sum (tests/test_exact_gpu.py).
#include "common.h"
#include "exact.h"

namespace isd {

__global__ __launch_bounds__(256) void exact_sum_kernel(const float* __restrict__ x, int64_t n, ExactAcc* __restrict__ acc) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) exact_add(acc, x[i]);
}

__global__ void exact_read_kernel(const ExactAcc* __restrict__ acc, double* __restrict__ out) { *out = exact_get(acc); }

}  // namespace isd

extern "C" int64_t isd_exact_sum_workspace_bytes(void) { return (int64_t)sizeof(isd::ExactAcc); }

// x: n fp32 values (device); out: one double (device); workspace: isd_exact_sum_workspace_bytes() bytes (device)
extern "C" int isd_exact_sum(const float* x, int64_t n, double* out, void* workspace, void* stream) {
  ISD_CHECK_ARG(n >= 0 && out && workspace && (n == 0 || x), "isd_exact_sum: null argument or n=%lld", (long long)n);
  ISD_CHECK_ARG(n < (1ll << 31), "isd_exact_sum: at most 2^31 - 1 values per call (carry headroom of a digit)");
  hipStream_t st = (hipStream_t)stream;
  ISD_HIP_TRY(hipMemsetAsync(workspace, 0, sizeof(isd::ExactAcc), st));
  if (n > 0) {
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(isd::exact_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, n, (isd::ExactAcc*)workspace);
  }
  hipLaunchKernelGGL(isd::exact_read_kernel, dim3(1), dim3(1), 0, st, (const isd::ExactAcc*)workspace, out);
  ISD_LAUNCH_CHECK();
  return ISD_OK;
}
