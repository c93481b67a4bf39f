// Shader-clock probe for the measurements of bench.py (roofline_hbm reports the clock the passes ran at).
// One wave reads the shader-clock counter (s_memtime) and the constant-rate counter (s_memrealtime) around a spin of
// `spin_ticks` constant-rate ticks; the ratio of the two differences times the constant rate is the shader clock
// while whatever else is on the device runs.
#include "common.h"

namespace isd {

__global__ __launch_bounds__(64) void clock_probe_kernel(unsigned long long* __restrict__ out, unsigned spin_ticks) {
  if (threadIdx.x != 0) return;
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = r0;
  for (int i = 0; i < (1 << 22) && r1 - r0 < spin_ticks; ++i) {       // bounded: the counter advances on its own
    __builtin_amdgcn_s_sleep(8);
    r1 = __builtin_amdgcn_s_memrealtime();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[0] = t1 - t0;
  out[1] = r1 - r0;
}

}  // namespace isd

// out: two uint64 in device memory {shader-clock ticks, constant-rate ticks} over ~spin_us microseconds
extern "C" int isd_shader_clock_probe(uint64_t* out, int spin_us, void* stream) {
  ISD_CHECK_ARG(out && spin_us > 0 && spin_us <= 100000, "isd_shader_clock_probe: out=%p spin_us=%d", (void*)out, spin_us);
  int khz = 0, dev = 0;
  ISD_HIP_TRY(hipGetDevice(&dev));
  ISD_HIP_TRY(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev));
  if (khz <= 0) khz = 100000;
  hipLaunchKernelGGL(isd::clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long*)out,
                     (unsigned)((int64_t)spin_us * khz / 1000));
  ISD_LAUNCH_CHECK();
  return ISD_OK;
}

// rate of the constant counter in kHz (hipDeviceAttributeWallClockRate)
extern "C" int isd_wall_clock_khz(void) {
  int khz = 0, dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return ISD_ERR_INVALID;
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess) return ISD_ERR_INVALID;
  return khz > 0 ? khz : 100000;
}
