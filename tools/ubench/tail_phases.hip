// Phase timing of the fused transformer-tail forward (shader-clock stamps of wave 0 in the second block):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DTF_TIMING -I imagined-speech-decoding_amd/csrc -I include tools/ubench/tail_phases.hip \
//         imagined-speech-decoding_amd/csrc/api.cpp -o tools/ubench/tail_phases && tools/ubench/tail_phases [B] [train]
#include "../../imagined-speech-decoding_amd/csrc/tailfused.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char** argv) {
  const int64_t B = argc > 1 ? atoll(argv[1]) : 64;
  const bool train = argc > 2;
  const int N = 5, D = 32, H = 8, L = 4, ncls = 5;
  const int64_t np = isd_tail_fused_param_count(N + 1, D, L, ncls);
  std::vector<float> hp(np), ht(B * N * D);
  srand(1);
  for (auto& v : hp) v = (rand() / (float)RAND_MAX - 0.5f) * 0.3f;
  for (int l = 0; l < L; ++l) {}
  for (auto& v : ht) v = rand() / (float)RAND_MAX - 0.5f;
  float *P, *T, *Lg, *Sv = nullptr, *Xf = nullptr;
  hipMalloc(&P, np * 4); hipMalloc(&T, ht.size() * 4); hipMalloc(&Lg, B * ncls * 4);
  if (train) { hipMalloc(&Sv, isd_tail_fused_save_floats(B, N + 1, D, L) * 4); hipMalloc(&Xf, B * D * 4); }
  hipMemcpy(P, hp.data(), np * 4, hipMemcpyHostToDevice);
  hipMemcpy(T, ht.data(), ht.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) {
    hipEventRecord(e0, 0);
    int rc = isd_tail_fused_forward(P, T, Lg, Sv, Xf, B, N, N + 1, D, H, L, 2 * D, ncls, 0.f, 0.f, 0.f, 1, nullptr, nullptr);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("rc %d launch %.1f us\n", rc, ms * 1e3);
  }
  long long t[32];
  hipMemcpyFromSymbol(t, HIP_SYMBOL(isd::tf_times), sizeof(t));
  const char* names[] = {"stage weights", "LN1", "gemm qkv", "save qkv", "attention", "gemm proj", "resid+LN2",
                         "gemm w1", "gelu", "gemm w2", "resid"};
  for (int k = 0; k < 11; ++k) printf("%-14s %8lld clk\n", names[k], t[k + 1] - t[k]);
  printf("layer total    %8lld clk (shader clock, 100 MHz units? see launch time)\n", t[11] - t[0]);
  return 0;
}
