// Zero-phase FIR band-pass: the band-pass of the SVM baseline,
//   mne.filter.filter_data(X, 250, l_freq=4, h_freq=40)      (notebooks/svm_baseline.ipynb:238-239, :968-969)
// = a linear-phase windowed-sinc FIR (413 taps at those arguments) applied with its delay compensated, on the
// row extended by odd reflection about its end points ('reflect_limited': 2 x[0] - x[d] for d up to
// min(n_taps, T) - 1 samples, zeros beyond).  The taps are designed on the host (isd_amd/filter_design.py
// fir_design) and handed over as doubles.
//
//   y[n] = sum_{k < n_taps} h[k] * xe[n - half + k],   half = (n_taps - 1) / 2
//
// Kernel: one wave per (row group, tile of 512 outputs).  The extended tile (512 + n_taps - 1 samples, reflection
// applied while staging) sits in LDS; a lane owns 8 consecutive outputs and walks the taps in blocks of 8 with a
// 16-sample register window, of which each block replaces one half by four ds_read_b128 (the LDS rows carry a
// 16-byte skew per lane so that these reads are bank-conflict free).  The taps are wave-uniform and arrive in
// SGPRs.  Per block: 64 FMAs per lane against 4 LDS reads -- the kernel is VALU-bound (826 flop per 8 bytes of
// HBM traffic at 413 taps, machine balance ~20 flop/B).
//   fp32 : TWO rows per wave packed in register pairs, so the FMAs issue as v_pk_fma_f32;
//   fp64 : one row per wave, v_fma_f64 (the notebook filters float64 arrays).
#include "common.h"

struct isd_fir_plan {
  int n_taps, n_blk;     // n_blk = ceil(n_taps / 8) blocks of 8 taps; device tables are zero-padded to 8 * (n_blk + 3)
  float* d_hf;           // fp32 taps, each stored twice
  double* d_hd;
};

namespace isd {

typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int kFirR = 8;                 // outputs per lane = taps per block
constexpr int kFirTile = 64 * kFirR;     // outputs per wave and row
constexpr int kFirSkew = kFirR + 2;      // LDS elements per 8-sample block (8-byte elements: +16 B per lane)

template <typename VT> struct FirOps;
template <> struct FirOps<f2> {
  using S = float;
  using Tap = f2;
  static constexpr int NR = 2;
  static __device__ __forceinline__ f2 make(float a, float b) { return (f2){a, b}; }
  // taps arrive as (h, h) pairs (the fp32 table stores every tap twice): an aligned SGPR pair is a v_pk_fma_f32
  // operand as it stands
  static __device__ __forceinline__ f2 fma_(f2 h, f2 w, f2 acc) { return __builtin_elementwise_fma(h, w, acc); }
  static __device__ __forceinline__ float get(f2 v, int r) { return r ? v.y : v.x; }
};
template <> struct FirOps<double> {
  using S = double;
  using Tap = double;
  static constexpr int NR = 1;
  static __device__ __forceinline__ double make(double a, double) { return a; }
  static __device__ __forceinline__ double fma_(double h, double w, double acc) { return fma(h, w, acc); }
  static __device__ __forceinline__ double get(double v, int) { return v; }
};

// sample m of the row extended by limited odd reflection (mne.filter._smart_pad 'reflect_limited'), branch-free:
// inside the row x[m]; outside 2 x[edge] - x[mirror] while the mirror lies within n_edge samples of the edge, else 0
template <typename S>
__device__ __forceinline__ S fir_ext(const S* __restrict__ row, int m, int T, int n_edge) {
  const bool left = m < 0, right = m >= T;
  const int d = left ? -m : m - (T - 1);                       // distance from the edge sample (outside the row)
  const bool out = left || right, far = out && d > n_edge;
  const int edge = left ? 0 : T - 1;
  int idx = left ? d : (right ? T - 1 - d : m);
  idx = far ? edge : idx;
  const S v = row[idx], e = row[edge];
  return far ? (S)0 : (out ? (S)2 * e - v : v);
}

template <typename VT>
__global__ __launch_bounds__(64) void fir_kernel(const typename FirOps<VT>::S* __restrict__ x,
                                                 typename FirOps<VT>::S* __restrict__ y,
                                                 const typename FirOps<VT>::Tap* __restrict__ taps, int64_t rows, int T,
                                                 int half, int n_blk, int n_edge) {
  using O = FirOps<VT>;
  using S = typename O::S;
  extern __shared__ __attribute__((aligned(16))) unsigned char fir_smem[];
  VT* xs = reinterpret_cast<VT*>(fir_smem);                    // [(64 + n_blk + 2) blocks][kFirSkew], the last 2 only prefetched
  const int lane = threadIdx.x;
  const int tile0 = blockIdx.x * kFirTile;
  const int64_t r0 = (int64_t)blockIdx.y * O::NR;
  const int64_t r1 = (O::NR == 2 && r0 + 1 < rows) ? r0 + 1 : r0;
  const S* rowa = x + r0 * T;
  const S* rowb = x + r1 * T;

  const int n_win = (64 + n_blk) * kFirR;                      // samples staged: tile + 8 * n_blk (>= tile + n_taps - 1)
  for (int i = lane; i < n_win; i += 64) {
    const int m = tile0 - half + i;
    const S a = fir_ext(rowa, m, T, n_edge);
    const S b = O::NR == 2 ? fir_ext(rowb, m, T, n_edge) : (S)0;
    xs[(i >> 3) * kFirSkew + (i & 7)] = O::make(a, b);
  }
  __syncthreads();

  // 24-sample register window as three 8-sample blocks that rotate roles (no register moves); the block needed
  // next and its taps are requested one block of FMAs ahead of their use
  VT acc[kFirR], w0[kFirR], w1[kFirR], w2[kFirR];
  const VT* nx = xs + lane * kFirSkew;
  auto fetch = [&](VT (&w)[kFirR], int blk) {
#pragma unroll
    for (int r = 0; r < kFirR; ++r) w[r] = nx[blk * kFirSkew + r];
  };
  using Tap = typename O::Tap;
  auto fetch_taps = [&](Tap (&h)[kFirR], int blk) {
#pragma unroll
    for (int j = 0; j < kFirR; ++j) h[j] = taps[blk * kFirR + j];        // wave-uniform: scalar loads
  };
  auto block = [&](const VT (&lo)[kFirR], const VT (&hi)[kFirR], const Tap (&h)[kFirR]) {
#pragma unroll
    for (int j = 0; j < kFirR; ++j) {
#pragma unroll
      for (int r = 0; r < kFirR; ++r) acc[r] = O::fma_(h[j], r + j < kFirR ? lo[r + j] : hi[r + j - kFirR], acc[r]);
    }
  };
#pragma unroll
  for (int r = 0; r < kFirR; ++r) acc[r] = O::make((S)0, (S)0);
  Tap ha[kFirR], hb[kFirR], hc[kFirR];
  fetch(w0, 0);
  fetch(w1, 1);
  fetch_taps(ha, 0);
  __builtin_amdgcn_s_waitcnt(0xC07F);                          // lgkmcnt(0): the loop starts with nothing in flight
  int kb = 0;
  for (; kb + 3 <= n_blk; kb += 3) {                           // taps table and LDS window are padded by 3 blocks
    // Each prefetch (LDS window block + scalar tap loads) is issued in front of a block of FMAs that does not
    // need it and awaited behind that block.  The wait is explicit: taps come through SMEM, whose results return
    // out of order, so a wait placed at their first use would be lgkmcnt(0) right after the next prefetch was issued.
#define ISD_FIR_PHASE(LO, HI, H, WN, BN, HN, TN)  \
    fetch(WN, BN); fetch_taps(HN, TN);            \
    __builtin_amdgcn_sched_barrier(0);            \
    block(LO, HI, H);                             \
    __builtin_amdgcn_sched_barrier(0);            \
    __builtin_amdgcn_s_waitcnt(0xC07F);           /* lgkmcnt(0) */ \
    __builtin_amdgcn_sched_barrier(0);
    ISD_FIR_PHASE(w0, w1, ha, w2, kb + 2, hb, kb + 1)
    ISD_FIR_PHASE(w1, w2, hb, w0, kb + 3, hc, kb + 2)
    ISD_FIR_PHASE(w2, w0, hc, w1, kb + 4, ha, kb + 3)
#undef ISD_FIR_PHASE
  }
  if (kb < n_blk) {                                            // one or two blocks left
    fetch(w2, kb + 2); fetch_taps(hb, kb + 1);
    block(w0, w1, ha);
    if (kb + 1 < n_blk) block(w1, w2, hb);
  }

  const int n0 = tile0 + lane * kFirR;
#pragma unroll
  for (int q = 0; q < O::NR; ++q) {
    if (q == 1 && r1 == r0) break;
    S* out = y + (r0 + q) * T + n0;
    if (n0 + kFirR <= T && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
      if constexpr (sizeof(S) == 4) {
        *reinterpret_cast<float4*>(out) = make_float4(O::get(acc[0], q), O::get(acc[1], q), O::get(acc[2], q), O::get(acc[3], q));
        *reinterpret_cast<float4*>(out + 4) = make_float4(O::get(acc[4], q), O::get(acc[5], q), O::get(acc[6], q), O::get(acc[7], q));
      } else {
#pragma unroll
        for (int r = 0; r < kFirR; r += 2)
          *reinterpret_cast<double2*>(out + r) = make_double2(O::get(acc[r], q), O::get(acc[r + 1], q));
      }
    } else {
#pragma unroll
      for (int r = 0; r < kFirR; ++r)
        if (n0 + r < T) out[r] = O::get(acc[r], q);
    }
  }
}

}  // namespace isd

using namespace isd;

extern "C" int isd_fir_plan_create(isd_fir_plan** out, int n_taps, const double* taps) {
  ISD_CHECK_ARG(out && taps, "isd_fir_plan_create: null argument");
  ISD_CHECK_ARG(n_taps >= 1 && (n_taps & 1) && n_taps <= 65535, "isd_fir_plan_create: n_taps=%d (need an odd count <= 65535)", n_taps);
  for (int k = 0; k < n_taps / 2; ++k)
    ISD_CHECK_ARG(taps[k] == taps[n_taps - 1 - k], "isd_fir_plan_create: taps are not symmetric (zero-phase needs a linear-phase type-I filter)");
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0) {
    set_error("isd_fir_plan_create: no HIP device");
    return ISD_ERR_NO_DEVICE;
  }
  isd_fir_plan* p = new isd_fir_plan();
  p->n_taps = n_taps;
  p->n_blk = (n_taps + kFirR - 1) / kFirR;
  const int n = (p->n_blk + 3) * kFirR;                         // the kernel prefetches up to 3 blocks past the end
  double* hd = new double[n]();
  float* hf = new float[2 * n]();                                 // every tap twice: (h, h)
  for (int k = 0; k < n_taps; ++k) { hd[k] = taps[k]; hf[2 * k] = hf[2 * k + 1] = (float)taps[k]; }
  hipError_t e = hipMalloc(&p->d_hd, n * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(&p->d_hf, 2 * n * sizeof(float));
  if (e == hipSuccess) e = hipMemcpy(p->d_hd, hd, n * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(p->d_hf, hf, 2 * n * sizeof(float), hipMemcpyHostToDevice);
  delete[] hd;
  delete[] hf;
  if (e != hipSuccess) {
    set_error("isd_fir_plan_create: %s", hipGetErrorString(e));
    if (p->d_hd) (void)hipFree(p->d_hd);
    if (p->d_hf) (void)hipFree(p->d_hf);
    delete p;
    return ISD_ERR_HIP;
  }
  *out = p;
  return ISD_OK;
}

extern "C" int isd_fir_plan_destroy(isd_fir_plan* p) {
  if (!p) return ISD_OK;
  if (p->d_hd) (void)hipFree(p->d_hd);
  if (p->d_hf) (void)hipFree(p->d_hf);
  delete p;
  return ISD_OK;
}

extern "C" int isd_fir_plan_taps(const isd_fir_plan* p) { return p ? p->n_taps : ISD_ERR_INVALID; }

template <typename VT>
static int fir_launch(const isd_fir_plan* p, const void* x, void* y, int64_t rows, int T, void* stream) {
  using S = typename FirOps<VT>::S;
  const int64_t groups = cdiv(rows, FirOps<VT>::NR);
  ISD_CHECK_ARG(groups <= 65535 * (int64_t)32768, "isd_fir_zero_phase: too many rows (%lld)", (long long)rows);
  const size_t lds = (size_t)(64 + p->n_blk + 2) * kFirSkew * sizeof(VT);
  ISD_CHECK_ARG(lds <= 160 * 1024, "isd_fir_zero_phase: %d taps need %zu bytes of LDS", p->n_taps, lds);
  if (lds > 64 * 1024)
    ISD_HIP_TRY(hipFuncSetAttribute((const void*)fir_kernel<VT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int n_edge = (p->n_taps < T ? p->n_taps : T) - 1;
  using Tap = typename FirOps<VT>::Tap;
  const Tap* taps = sizeof(S) == 4 ? (const Tap*)p->d_hf : (const Tap*)p->d_hd;
  const unsigned tiles = (unsigned)cdiv(T, kFirTile);
  // grid.y is limited to 65535: walk the row groups in slabs
  for (int64_t g0 = 0; g0 < groups; g0 += 65535) {
    const unsigned gy = (unsigned)((groups - g0) < 65535 ? (groups - g0) : 65535);
    const int64_t roff = g0 * FirOps<VT>::NR;
    hipLaunchKernelGGL((fir_kernel<VT>), dim3(tiles, gy), dim3(64), lds, (hipStream_t)stream, (const S*)x + roff * T,
                       (S*)y + roff * T, taps, rows - roff, T, (p->n_taps - 1) / 2, p->n_blk, n_edge);
    ISD_LAUNCH_CHECK();
  }
  return ISD_OK;
}

extern "C" int isd_fir_zero_phase_f32(const isd_fir_plan* p, const float* x, float* y, int64_t rows, int T, void* stream) {
  ISD_CHECK_ARG(p && x && y, "isd_fir_zero_phase_f32: null argument");
  ISD_CHECK_ARG(rows >= 0 && T >= 1, "isd_fir_zero_phase_f32: rows=%lld T=%d", (long long)rows, T);
  ISD_CHECK_ARG(x != y, "isd_fir_zero_phase_f32: in-place filtering is not supported");
  if (rows == 0) return ISD_OK;
  return fir_launch<f2>(p, x, y, rows, T, stream);
}

extern "C" int isd_fir_zero_phase_f64(const isd_fir_plan* p, const double* x, double* y, int64_t rows, int T, void* stream) {
  ISD_CHECK_ARG(p && x && y, "isd_fir_zero_phase_f64: null argument");
  ISD_CHECK_ARG(rows >= 0 && T >= 1, "isd_fir_zero_phase_f64: rows=%lld T=%d", (long long)rows, T);
  ISD_CHECK_ARG(x != y, "isd_fir_zero_phase_f64: in-place filtering is not supported");
  if (rows == 0) return ISD_OK;
  return fir_launch<double>(p, x, y, rows, T, stream);
}
