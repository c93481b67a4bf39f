// Microbenchmark: fp32 VALU issue rate on gfx950 (scalar v_fma_f32 vs packed v_pk_fma_f32), fp64 fma.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  if (MODE == 0) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; ++i) {
      x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
      x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
      asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  } else if (MODE == 1) {
    f2 x0 = {(float)threadIdx.x, 1.f}, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f;
    f2 av = {a, a}, bv = {b, b};
    for (int i = 0; i < iters; ++i) {
      x0 = __builtin_elementwise_fma(x0, av, bv); x1 = __builtin_elementwise_fma(x1, av, bv);
      x2 = __builtin_elementwise_fma(x2, av, bv); x3 = __builtin_elementwise_fma(x3, av, bv);
      asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
    }
    f2 s = x0 + x1 + x2 + x3;
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
  } else {
    double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    double ad = a, bd = b;
    for (int i = 0; i < iters; ++i) {
      x0 = __builtin_fma(x0, ad, bd); x1 = __builtin_fma(x1, ad, bd); x2 = __builtin_fma(x2, ad, bd); x3 = __builtin_fma(x3, ad, bd);
      x4 = __builtin_fma(x4, ad, bd); x5 = __builtin_fma(x5, ad, bd); x6 = __builtin_fma(x6, ad, bd); x7 = __builtin_fma(x7, ad, bd);
      asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
    }
    out[blockIdx.x * 256 + threadIdx.x] = (float)(x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7);
  }
}

template <int MODE>
void run(const char* name, int blocks, int iters) {
  float* out;
  hipMalloc(&out, sizeof(float) * blocks * 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.999f, 0.001f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.999f, 0.001f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double fmas = (double)blocks * 256 * iters * 8;
  printf("%-22s blocks=%5d  %.3f ms  %.1f TFLOP/s (fma=2)  %.2f lane-fma/clk/SIMD @2.4GHz\n", name, blocks, ms,
         2 * fmas / ms / 1e9, fmas / (ms * 1e-3) / (1024.0 * 2.4e9));
  hipFree(out);
}

int main() {
  for (int blocks : {256 * 2, 256 * 4, 256 * 8}) {
    run<0>("v_fma_f32 x8", blocks, 20000);
    run<1>("v_pk_fma_f32 x4", blocks, 20000);
    run<2>("v_fma_f64 x8", blocks, 20000);
  }
  return 0;
}
