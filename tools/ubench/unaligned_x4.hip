// Does a dword-aligned (not 16-byte aligned) global_load_dwordx4 work on gfx950, and what does it cost?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/unaligned_x4.hip -o tools/ubench/unaligned_x4 && tools/ubench/unaligned_x4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
__global__ void k(const float* __restrict__ x, float* __restrict__ y, int n_rows, int T, int off) {
  const int64_t r = blockIdx.x * 256 + threadIdx.x;
  if (r >= n_rows) return;
  const float* p = x + r * T + off;
  float s = 0.f;
  for (int t = 0; t + 4 <= T - off; t += 4) {
    const f4u v = *reinterpret_cast<const f4u*>(p + t);
    s += v.x + 2.f * v.y + 3.f * v.z + 4.f * v.w;
  }
  y[r] = s;
}
int main() {
  const int n = 1 << 20, T = 65;
  std::vector<float> h((size_t)n * T);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(i % 97) * 0.25f;
  float *x, *y;
  hipMalloc(&x, h.size() * 4); hipMalloc(&y, n * 4);
  hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  for (int off = 0; off < 2; ++off) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, x, y, n, T, off);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, x, y, n, T, off);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<float> out(n);
    hipMemcpy(out.data(), y, n * 4, hipMemcpyDeviceToHost);
    double err = 0;
    for (int r = 0; r < n; r += 4097) {
      double s = 0;
      for (int t = 0; t + 4 <= T - off; t += 4) {
        const float* p = &h[(size_t)r * T + off + t];
        s += p[0] + 2.0 * p[1] + 3.0 * p[2] + 4.0 * p[3];
      }
      err = fmax(err, fabs(s - out[r]));
    }
    printf("off %d: %.3f ms, %.1f GB/s, max err %.3g (%s)\n", off, ms, h.size() * 4 / ms * 1e-6, err, hipGetErrorString(hipGetLastError()));
  }
  return 0;
}
