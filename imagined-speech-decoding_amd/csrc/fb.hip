// Filterbank: per-band Butterworth SOS cascade as a chunked linear-recurrence scan.
//
// Replaces scipy `sosfilt(butter(..., 'sos'), x, axis=-1)` (spec S steps 1-2; the
// reference's only band-pass is notebooks/svm_baseline.ipynb:238).
//
// Mapping (CDNA4, wave64): a 16-lane DPP row owns one 512-sample segment of one
// (trial, channel) row; each lane owns a contiguous 32-sample chunk in registers.
// Per band and per biquad section:
//   1. in-chunk DF2T recursion from zero state            (3 VALU ops / sample)
//   2. inclusive scan of the chunk-end states over the 16 lanes with the constant
//      2x2 transition matrix M = A^32 (Kogge-Stone over DPP row_shr, fp64)
//   3. zero-input correction  y[n] += h1[n]*s1_in + h2[n]*s2_in   (2 FMA / sample)
// Global traffic is coalesced float4; the chunk<->coalesced transposition goes
// through a padded LDS tile (stride 36 floats: conflict-free ds_read_b128).
// Rows longer than 512 samples chain 2 or 4 rows-of-16-lanes (and loop) with the
// group transition matrix P = M^16 and per-lane powers Q_i = M^i.
#include "common.h"
#include "stft_plan.h"
#include <math.h>
#include <vector>
#include <string.h>
#include <type_traits>

namespace isd {

constexpr int kL = 32;          // samples per lane
constexpr int kSeg = 16 * kL;   // samples per 16-lane group
constexpr int kPad = kL + 4;    // LDS chunk stride (floats)
constexpr int kMaxSec = 8;

struct FbSec {                  // constants of one (band, section); wave-uniform -> SMEM loads
  double Mp[4][4];              // M^(1,2,4,8), row-major 2x2, M = A^32
  double P[4];                  // M^16
  double a1d, a2d;
  double hd[kL][2];             // zero-input response seen at the output: row 0 of A^n
  float a1f, a2f;
  float hf[kL][2];
  float hq[2][kL / 2];          // first half of the table, structure-of-arrays (the packed fp32 cascade, below)
  float N16f[4];                // A^16: joins the two 16-sample halves a lane runs side by side in one register pair
};

struct FbBand {
  double gd;
  float gf;
  float pad;
};

}  // namespace isd

// The bands of a plan are split by arithmetic: set 0 runs the in-chunk recursion in fp32, set 1 in fp64
// (ISD_FB_AUTO decides per band; ISD_FB_F32 / ISD_FB_F64 put every band in one set).  `d_map` gives the
// position of a set's band in the caller's band order.
struct FbSet {
  int nb;
  isd::FbSec* d_sec;   // [nb][n_sections]
  isd::FbBand* d_band; // [nb]
  double* d_Q;         // [nb][n_sections][16][4]  per-lane M^i
  int* d_map;          // [nb] output band index
};
struct isd_fb_plan {
  int n_bands, n_sections, precision;   // precision: ISD_FB_F32, ISD_FB_F64 or ISD_FB_MIXED
  FbSet set[2];
  int* host_map[2];    // host copies of the output maps (band-bin tables are gathered through them)
};

namespace isd {

typedef float f2 __attribute__((ext_vector_type(2)));

// Value types: float (one row per lane group), double (fp64 FMA issues at the same rate as unpacked
// fp32 on gfx950), and f2 = TWO rows packed in a register pair so the cascade runs on
// v_pk_fma_f32 / v_pk_add_f32.  Measured at cfg2 (tools/ubench/valu_rate.hip: v_fma_f32 70 TF,
// v_pk_fma_f32 119 TF, v_fma_f64 62 TF) the packed variant halves the cascade's instruction count
// but its 64 extra VGPRs cost a wave per SIMD and it ran slower (2.10 vs 1.49 ms), so the dispatch
// uses float / double; f2 is kept for the kernels that can afford the registers.
template <typename VT> struct VOps;
template <> struct VOps<float> {
  using S = float;
  static constexpr int NR = 1;
  // chunk container: 16 register PAIRS {sample j, sample j + 16}.  The two 16-sample halves of the chunk are
  // independent recurrences (zero state each), so the whole cascade runs on v_pk_add/v_pk_fma_f32; the halves
  // are joined afterwards through A^16 (section<>, below).
  typedef f2 Arr[kL / 2];
  static __device__ __forceinline__ float at(const Arr& a, int n) { return (n & 16) ? a[n & 15].y : a[n & 15].x; }
  static __device__ __forceinline__ void put(Arr& a, int n, float s) { if (n & 16) a[n & 15].y = s; else a[n & 15].x = s; }
  static __device__ __forceinline__ float splat(float s) { return s; }
  static __device__ __forceinline__ float fma_(float a, float b, float c) { return fmaf(a, b, c); }
  static __device__ __forceinline__ float get(float v, int) { return v; }
  static __device__ __forceinline__ void set(float& v, int, float s) { v = s; }
  static __device__ __forceinline__ float a1(const FbSec& s) { return s.a1f; }
  static __device__ __forceinline__ float a2(const FbSec& s) { return s.a2f; }
  static __device__ __forceinline__ float h(const FbSec& s, int n, int j) { return s.hf[n][j]; }
  static __device__ __forceinline__ float g(const FbBand& b) { return b.gf; }
};
template <> struct VOps<f2> {
  using S = float;
  static constexpr int NR = 2;
  typedef f2 Arr[kL];
  static __device__ __forceinline__ f2 at(const Arr& a, int n) { return a[n]; }
  static __device__ __forceinline__ void put(Arr& a, int n, f2 s) { a[n] = s; }
  static __device__ __forceinline__ f2 splat(float s) { return (f2){s, s}; }
  static __device__ __forceinline__ f2 fma_(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
  static __device__ __forceinline__ float get(f2 v, int r) { return r ? v.y : v.x; }
  static __device__ __forceinline__ void set(f2& v, int r, float s) { if (r) v.y = s; else v.x = s; }
  static __device__ __forceinline__ float a1(const FbSec& s) { return s.a1f; }
  static __device__ __forceinline__ float a2(const FbSec& s) { return s.a2f; }
  static __device__ __forceinline__ float h(const FbSec& s, int n, int j) { return s.hf[n][j]; }
  static __device__ __forceinline__ float g(const FbBand& b) { return b.gf; }
};
template <> struct VOps<double> {
  using S = double;
  static constexpr int NR = 1;
  typedef double Arr[kL];
  static __device__ __forceinline__ double at(const Arr& a, int n) { return a[n]; }
  static __device__ __forceinline__ void put(Arr& a, int n, double s) { a[n] = s; }
  static __device__ __forceinline__ double splat(double s) { return s; }
  static __device__ __forceinline__ double fma_(double a, double b, double c) { return fma(a, b, c); }
  static __device__ __forceinline__ double get(double v, int) { return v; }
  static __device__ __forceinline__ void set(double& v, int, double s) { v = s; }
  static __device__ __forceinline__ double a1(const FbSec& s) { return s.a1d; }
  static __device__ __forceinline__ double a2(const FbSec& s) { return s.a2d; }
  static __device__ __forceinline__ double h(const FbSec& s, int n, int j) { return s.hd[n][j]; }
  static __device__ __forceinline__ double g(const FbBand& b) { return b.gd; }
};

template <int D>
__device__ __forceinline__ void scan_step(double& e1, double& e2, const double* M) {
  double p1 = row_shr<D>(e1), p2 = row_shr<D>(e2);
  e1 = fma(M[0], p1, fma(M[1], p2, e1));
  e2 = fma(M[2], p1, fma(M[3], p2, e2));
}

// One biquad section over the lane's chunk(s), including the cross-chunk state fix-up.
// c1/c2[r]: incoming state of the wave's first group for row-set r (GPR == 4 loop carry), updated.
template <typename VT, int GPR>
__device__ __forceinline__ void section(typename VOps<VT>::Arr& v, const FbSec& sc, const double* __restrict__ Qsec,
                                        int lane, double (&c1)[VOps<VT>::NR], double (&c2)[VOps<VT>::NR]) {
  using O = VOps<VT>;
  constexpr int NR = O::NR;
  const VT na1 = O::splat(-O::a1(sc)), na2 = O::splat(-O::a2(sc));
  VT s1 = O::splat(0), s2 = O::splat(0);
  [[maybe_unused]] float sA1 = 0.f, sA2 = 0.f;            // fp32 path: final state of the first half (zero start)
  if constexpr (std::is_same<VT, float>::value) {
    const f2 NA1 = {na1, na1}, NA2 = {na2, na2};
    f2 S1 = {0.f, 0.f}, S2 = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < kL / 2; ++j) {
      const f2 x = v[j];
      const f2 y = x + S1;
      S1 = __builtin_elementwise_fma(NA1, y, S2);
      S2 = __builtin_elementwise_fma(NA2, y, -x);
      v[j] = y;
    }
    sA1 = S1.x; sA2 = S2.x;
    // state after the whole chunk from a zero start: A^16 sA + sB
    s1 = fmaf(sc.N16f[0], S1.x, fmaf(sc.N16f[1], S2.x, S1.y));
    s2 = fmaf(sc.N16f[2], S1.x, fmaf(sc.N16f[3], S2.x, S2.y));
  } else {
#pragma unroll
    for (int n = 0; n < kL; ++n) {
      const VT x = O::at(v, n);
      const VT y = x + s1;
      s1 = O::fma_(na1, y, s2);
      s2 = O::fma_(na2, y, -x);
      O::put(v, n, y);
    }
  }
  double i1[NR], i2[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    double e1 = (double)O::get(s1, r), e2 = (double)O::get(s2, r);
    scan_step<1>(e1, e2, sc.Mp[0]);
    scan_step<2>(e1, e2, sc.Mp[1]);
    scan_step<4>(e1, e2, sc.Mp[2]);
    scan_step<8>(e1, e2, sc.Mp[3]);
    i1[r] = row_shr<1>(e1);                            // exclusive; lane 0 of each 16-lane row -> 0
    i2[r] = row_shr<1>(e2);
    if (GPR > 1) {
      // group totals (wave-uniform) and the serial chain over the 4 groups of the wave
      double E1[4], E2[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        E1[g] = read_lane(e1, 16 * g + 15);
        E2[g] = read_lane(e2, 16 * g + 15);
      }
      double C1[4], C2[4];
      C1[0] = (GPR == 4) ? c1[r] : 0.0;
      C2[0] = (GPR == 4) ? c2[r] : 0.0;
#pragma unroll
      for (int g = 1; g < 4; ++g) {
        if (g % GPR == 0) {
          C1[g] = 0.0;
          C2[g] = 0.0;
        } else {
          C1[g] = fma(sc.P[0], C1[g - 1], fma(sc.P[1], C2[g - 1], E1[g - 1]));
          C2[g] = fma(sc.P[2], C1[g - 1], fma(sc.P[3], C2[g - 1], E2[g - 1]));
        }
      }
      if (GPR == 4) {
        c1[r] = fma(sc.P[0], C1[3], fma(sc.P[1], C2[3], E1[3]));
        c2[r] = fma(sc.P[2], C1[3], fma(sc.P[3], C2[3], E2[3]));
      }
      const int q = lane >> 4;
      const double m1 = q == 0 ? C1[0] : q == 1 ? C1[1] : q == 2 ? C1[2] : C1[3];
      const double m2 = q == 0 ? C2[0] : q == 1 ? C2[1] : q == 2 ? C2[2] : C2[3];
      const double* Q = Qsec + (lane & 15) * 4;        // M^i of this lane
      i1[r] = fma(Q[0], m1, fma(Q[1], m2, i1[r]));
      i2[r] = fma(Q[2], m1, fma(Q[3], m2, i2[r]));
    }
  }
  VT t1, t2;
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    O::set(t1, r, (typename O::S)i1[r]);
    O::set(t2, r, (typename O::S)i2[r]);
  }
  if constexpr (std::is_same<VT, float>::value) {
    // incoming state of the second half: A^16 t + sA; both halves then take the same 16-entry table.
    // The table scalars are fetched here, after the scan released its matrix SGPRs: hoisted to the top of the
    // section they do not fit beside them and the compiler spills SGPRs into VGPR lanes.
    __builtin_amdgcn_sched_barrier(0);
    const float u1 = fmaf(sc.N16f[0], t1, fmaf(sc.N16f[1], t2, sA1));
    const float u2 = fmaf(sc.N16f[2], t1, fmaf(sc.N16f[3], t2, sA2));
    const f2 T1 = {t1, u1}, T2 = {t2, u2};
#pragma unroll
    for (int j = 0; j < kL / 2; ++j)
      v[j] = __builtin_elementwise_fma((f2){sc.hq[0][j], sc.hq[0][j]}, T1,
                                       __builtin_elementwise_fma((f2){sc.hq[1][j], sc.hq[1][j]}, T2, v[j]));
  } else {
#pragma unroll
    for (int n = 0; n < kL; ++n)
      v[n] = O::fma_(O::splat(O::h(sc, n, 0)), t1, O::fma_(O::splat(O::h(sc, n, 1)), t2, v[n]));
  }
}

// Cooperative (whole wave) coalesced load of the 4 groups' 512-sample segments into the
// padded chunk-major LDS tile.  gbase[g] < 0 marks an absent group.
__device__ __forceinline__ void tile_load(float* tile, const float* __restrict__ x, int lane,
                                          const int64_t (&gbase)[4], const int (&gt0)[4], int T, bool vec) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int e = (k * 64 + lane) * 4;              // element inside the 512-sample segment
      const int t = gt0[g] + e;
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gbase[g] >= 0) {
        const float* p = x + gbase[g] + t;
        if (vec && t + 3 < T) {
          val = *reinterpret_cast<const float4*>(p);
        } else {
          if (t + 0 < T) val.x = p[0];
          if (t + 1 < T) val.y = p[1];
          if (t + 2 < T) val.z = p[2];
          if (t + 3 < T) val.w = p[3];
        }
      }
      *reinterpret_cast<float4*>(tile + (g * 16 + (e >> 5)) * kPad + (e & 31)) = val;
    }
  }
}

__device__ __forceinline__ void tile_store(const float* tile, float* __restrict__ y, int lane,
                                           const int64_t (&gbase)[4], const int (&gt0)[4], int T, bool vec) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int e = (k * 64 + lane) * 4;
      const int t = gt0[g] + e;
      if (gbase[g] < 0 || t >= T) continue;
      const float4 val = *reinterpret_cast<const float4*>(tile + (g * 16 + (e >> 5)) * kPad + (e & 31));
      float* p = y + gbase[g] + t;
      if (vec && t + 3 < T) {
        *reinterpret_cast<float4*>(p) = val;
      } else {
        p[0] = val.x;
        if (t + 1 < T) p[1] = val.y;
        if (t + 2 < T) p[2] = val.z;
        if (t + 3 < T) p[3] = val.w;
      }
    }
  }
}

// Load the lane's chunk of every row-set through the LDS tile (coalesced global reads, chunk-major
// registers) and scale it by `gain`.
template <typename VT>
__device__ __forceinline__ void load_chunks(typename VOps<VT>::Arr& v, float* tile, const float* __restrict__ x,
                                            int lane, const int64_t (&xbase)[VOps<VT>::NR][4], const int (&gt0)[4],
                                            int T, bool vec) {
  using O = VOps<VT>;
#pragma unroll
  for (int r = 0; r < O::NR; ++r) {
    wave_lds_sync();
    tile_load(tile, x, lane, xbase[r], gt0, T, vec);
    wave_lds_sync();
    const float* src = tile + lane * kPad;            // (q*16 + i) == lane
#pragma unroll
    for (int n = 0; n < kL; n += 4) {
      const float4 f = *reinterpret_cast<const float4*>(src + n);
      if constexpr (O::NR == 1) {
        O::put(v, n, (typename O::S)f.x);
        O::put(v, n + 1, (typename O::S)f.y);
        O::put(v, n + 2, (typename O::S)f.z);
        O::put(v, n + 3, (typename O::S)f.w);
      } else {
        O::set(v[n], r, f.x);
        O::set(v[n + 1], r, f.y);
        O::set(v[n + 2], r, f.z);
        O::set(v[n + 3], r, f.w);
      }
    }
  }
}

// One wave per workgroup.  GPR = 16-lane groups per row (1: T<=512, 2: T<=1024, 4: any T);
// the wave serves NR * 4/GPR rows.
template <typename VT, int GPR>
__global__ __launch_bounds__(64) void fb_kernel(const FbSec* __restrict__ secs, const FbBand* __restrict__ bands,
                                                const double* __restrict__ Qtab, const float* __restrict__ x,
                                                float* __restrict__ y, int64_t R, int C, int T, int nb, int ns,
                                                int vec, const int* __restrict__ bmap, int nb_out) {
  using O = VOps<VT>;
  constexpr int NR = O::NR;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* tile = reinterpret_cast<float*>(smem_raw);                        // [4][16][kPad]
  double* carry = reinterpret_cast<double*>(smem_raw + 4 * 16 * kPad * 4);  // [NR][nb][ns][2]  (GPR == 4)
  const int lane = threadIdx.x;
  constexpr int RPS = 4 / GPR;                                             // rows per row-set
  const int64_t row0 = (int64_t)blockIdx.x * (RPS * NR);
  const int n_iter = (GPR == 4) ? (T + 4 * kSeg - 1) / (4 * kSeg) : 1;

  if (GPR == 4) {
    for (int j = lane; j < NR * nb * ns * 2; j += 64) carry[j] = 0.0;
  }

  for (int it = 0; it < n_iter; ++it) {
    int64_t xbase[NR][4];
    int64_t rowg[NR][4];
    int gt0[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      gt0[g] = (it * GPR + g % GPR) * kSeg;
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const int64_t row = row0 + r * RPS + g / GPR;
        rowg[r][g] = row;
        xbase[r][g] = (row < R && gt0[g] < T) ? row * (int64_t)T : -1;
      }
    }
    typename O::Arr xs;
    load_chunks<VT>(xs, tile, x, lane, xbase, gt0, T, vec != 0);
    constexpr int NA = sizeof(typename O::Arr) / sizeof(xs[0]);
    for (int b = 0; b < nb; ++b) {
      typename O::Arr v;
      const auto gain = O::g(bands[b]);
#pragma unroll
      for (int n = 0; n < NA; ++n) v[n] = xs[n] * gain;
      for (int s = 0; s < ns; ++s) {
        const int bs = b * ns + s;
        double c1[NR], c2[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          c1[r] = (GPR == 4) ? carry[(r * nb * ns + bs) * 2] : 0.0;
          c2[r] = (GPR == 4) ? carry[(r * nb * ns + bs) * 2 + 1] : 0.0;
        }
        section<VT, GPR>(v, secs[bs], Qtab + (int64_t)bs * 64, lane, c1, c2);
        if (GPR == 4 && n_iter > 1 && lane == 0) {
#pragma unroll
          for (int r = 0; r < NR; ++r) {
            carry[(r * nb * ns + bs) * 2] = c1[r];
            carry[(r * nb * ns + bs) * 2 + 1] = c2[r];
          }
        }
      }
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        wave_lds_sync();                              // earlier readers of the tile are done
        float* dst = tile + lane * kPad;
#pragma unroll
        for (int n = 0; n < kL; n += 4)
          *reinterpret_cast<float4*>(dst + n) =
              make_float4((float)O::get(O::at(v, n), r), (float)O::get(O::at(v, n + 1), r),
                          (float)O::get(O::at(v, n + 2), r), (float)O::get(O::at(v, n + 3), r));
        wave_lds_sync();
        int64_t ybase[4];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          if (xbase[r][gq] < 0) { ybase[gq] = -1; continue; }
          const int64_t row = rowg[r][gq];
          const int64_t bt = row / C, ch = row - bt * C;
          ybase[gq] = ((bt * nb_out + bmap[b]) * C + ch) * (int64_t)T;
        }
        tile_store(tile, y, lane, ybase, gt0, T, vec != 0);
      }
    }
  }
}

struct FusedBands {
  int klo[kMaxBands];
  int khi[kMaxBands];
};

// Windowed DFT of the band's own bins over the lane's chunk (register pairs) and the reduction to
// band magnitude / power.  Frame j of the row is chunk j-1 (first window half, table entries 0..31)
// followed by chunk j (second half, 32..63): every lane forms both partial sums, `row_shr:1` joins
// neighbours.  o0 = frame (lane & 15), o16 = frame 16 (meaningful on lane 15).
__device__ __forceinline__ void band_reduce_pairs(const f2 (&vf)[kL / 2], const float2* __restrict__ dft, int klo,
                                                  int khi, bool mine_lo_hi_valid, int my_klo, int my_khi, float scale2,
                                                  int mode, float& o0, float& o16) {
  float acc = 0.f, acc16 = 0.f;
  for (int k = klo; k <= khi; ++k) {
    const float2* __restrict__ tb = dft + k * 64;
    f2 p1 = {0.f, 0.f}, p2 = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < kL / 2; ++j) {
      // the 128 table scalars of a bin do not fit the SGPR file at once: fetch them in two batches of 64
      if (j == kL / 4) __builtin_amdgcn_sched_barrier(0);
      const f2 pr = vf[j];                                              // samples j, j + 16
      const f2 x0 = __builtin_shufflevector(pr, pr, 0, 0), x1 = __builtin_shufflevector(pr, pr, 1, 1);
      const float2 ca0 = tb[j], cb0 = tb[kL + j], ca1 = tb[j + 16], cb1 = tb[kL + j + 16];
      p1 = __builtin_elementwise_fma(x0, (f2){ca0.x, ca0.y}, p1);
      p2 = __builtin_elementwise_fma(x0, (f2){cb0.x, cb0.y}, p2);
      p1 = __builtin_elementwise_fma(x1, (f2){ca1.x, ca1.y}, p1);
      p2 = __builtin_elementwise_fma(x1, (f2){cb1.x, cb1.y}, p2);
    }
    const float zr = p2.x + row_shr<1>(p1.x), zi = p2.y + row_shr<1>(p1.y);
    float pw = (zr * zr + zi * zi) * scale2;
    float pw16 = (p1.x * p1.x + p1.y * p1.y) * scale2;
    if (mode == ISD_BP_MAGNITUDE) { pw = sqrtf(pw); pw16 = sqrtf(pw16); }
    const bool in = !mine_lo_hi_valid || (k >= my_klo && k <= my_khi);
    acc += in ? pw : 0.f;
    acc16 += in ? pw16 : 0.f;
  }
  o0 = acc;
  o16 = acc16;
}

// Band aggregation of materialised filtered signals y[B][nb][C][T] (nperseg 64 / hop 32, T <= 512):
// same chunk layout as the filterbank (coalesced float4 loads through the LDS tile), direct DFT of
// each row's own band bins.  HBM-bound: reads nb*C*T*4 bytes per trial, writes nb*C*J*4.
__global__ __launch_bounds__(64) void bandpower_direct_kernel(const float2* __restrict__ dft,
                                                              const float* __restrict__ y, float* __restrict__ feat,
                                                              int64_t R, int C, int T, int nb, int J, float scale2,
                                                              FusedBands fbnd, int mode, float eps, int vec) {
  using O = VOps<float>;
  __shared__ __attribute__((aligned(16))) float tile[4 * 16 * kPad];
  const int lane = threadIdx.x;
  const int i = lane & 15;
  const int64_t row0 = (int64_t)blockIdx.x * 4;
  int64_t xbase[1][4];
  int gt0[4];
  int wlo = 1 << 30, whi = -1;                                  // wave-uniform union of the 4 rows' bin ranges
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    gt0[g] = 0;
    const int64_t r = row0 + g;
    xbase[0][g] = (r < R) ? r * (int64_t)T : -1;
    if (r < R) {
      const int band = (int)((r / C) % nb);
      if (fbnd.khi[band] >= fbnd.klo[band]) {
        wlo = fbnd.klo[band] < wlo ? fbnd.klo[band] : wlo;
        whi = fbnd.khi[band] > whi ? fbnd.khi[band] : whi;
      }
    }
  }
  typename O::Arr vf;
  load_chunks<float>(vf, tile, y, lane, xbase, gt0, T, vec != 0);
  if (T < kSeg) {
#pragma unroll
    for (int n = 0; n < kL; ++n)
      if (i * kL + n >= T) O::put(vf, n, 0.f);
  }
  const int64_t row = row0 + (lane >> 4);
  const int band = (int)((row / C) % nb);
  const int klo = fbnd.klo[band], khi = fbnd.khi[band];
  float o0, o16;
  band_reduce_pairs(vf, dft, wlo, whi, true, klo, khi, scale2, mode, o0, o16);
  const float inv = khi >= klo ? 1.f / (float)(khi - klo + 1) : 0.f;
  o0 *= inv;
  o16 *= inv;
  if (mode == ISD_BP_LOGPOWER) { o0 = logf(o0 + eps); o16 = logf(o16 + eps); }
  if (row < R) {
    float* o = feat + row * (int64_t)J;                         // feat has the same [B][nb][C] row order as y
    if (i < J) o[i] = o0;
    if (i == 15 && J == 17) o[16] = o16;
  }
}

// Fused spec-S extractor for T <= 512, nperseg 64 / hop 32: after the cascade each lane
// holds chunk i of its row(s); STFT frame j is chunk j-1 (window first half) followed by
// chunk j (second half), so every lane forms two partial windowed DFT sums per bin and
// one DPP row_shr joins neighbours.  Only the band's own bins are evaluated.
template <typename VT>
__global__ __launch_bounds__(64) void fused_kernel(const FbSec* __restrict__ secs, const FbBand* __restrict__ bands,
                                                   const float2* __restrict__ dft, const float* __restrict__ x,
                                                   float* __restrict__ feat, int64_t R, int C, int T, int nb, int ns,
                                                   int J, float scale2, FusedBands fbnd, int mode, float eps,
                                                   int vec, const int* __restrict__ bmap, int nb_out) {
  using O = VOps<VT>;
  constexpr int NR = O::NR;
  using FT = typename std::conditional<NR == 2, f2, float>::type;      // DFT arithmetic is fp32
  using FO = VOps<FT>;
  __shared__ __attribute__((aligned(16))) float tile[4 * 16 * kPad];
  const int lane = threadIdx.x;
  const int i = lane & 15;
  const int64_t row0 = (int64_t)blockIdx.x * (4 * NR);
  int64_t xbase[NR][4];
  int gt0[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    gt0[g] = 0;
#pragma unroll
    for (int r = 0; r < NR; ++r) xbase[r][g] = (row0 + r * 4 + g < R) ? (row0 + r * 4 + g) * (int64_t)T : -1;
  }
  typename O::Arr xs;
  load_chunks<VT>(xs, tile, x, lane, xbase, gt0, T, vec != 0);
  constexpr int NA = sizeof(typename O::Arr) / sizeof(xs[0]);
  for (int b = 0; b < nb; ++b) {
    typename O::Arr v;
    const auto gain = O::g(bands[b]);
#pragma unroll
    for (int n = 0; n < NA; ++n) v[n] = xs[n] * gain;
    for (int s = 0; s < ns; ++s) {
      double c1[NR], c2[NR];
#pragma unroll
      for (int r = 0; r < NR; ++r) c1[r] = c2[r] = 0.0;
      section<VT, 1>(v, secs[b * ns + s], nullptr, lane, c1, c2);
    }
    typename FO::Arr vf;                              // fp32 copy for the DFT (aliases v when VT is fp32)
#pragma unroll
    for (int n = 0; n < kL; ++n) {
      if constexpr (NR == 1) {
        FO::put(vf, n, (float)O::at(v, n));
      } else {
        vf[n] = v[n];
      }
    }
    if (T < kSeg) {                                   // the STFT sees y[0..T) then zeros, not the filter's ringing
#pragma unroll
      for (int n = 0; n < kL; ++n)
        if (i * kL + n >= T) FO::put(vf, n, FO::splat(0.f));
    }
    const int klo = fbnd.klo[b], khi = fbnd.khi[b];
    float acc[NR], acc16[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) acc[r] = acc16[r] = 0.f;
    if constexpr (NR == 1) {
      band_reduce_pairs(vf, dft, klo, khi, false, 0, 0, scale2, mode, acc[0], acc16[0]);
    } else {
      for (int k = klo; k <= khi; ++k) {
        const float2* __restrict__ tb = dft + k * 64;
        FT p1r = FO::splat(0.f), p1i = FO::splat(0.f), p2r = FO::splat(0.f), p2i = FO::splat(0.f);
#pragma unroll
        for (int n = 0; n < kL; ++n) {
          const float2 ca = tb[n], cb = tb[kL + n];
          p1r = FO::fma_(vf[n], FO::splat(ca.x), p1r);
          p1i = FO::fma_(vf[n], FO::splat(ca.y), p1i);
          p2r = FO::fma_(vf[n], FO::splat(cb.x), p2r);
          p2i = FO::fma_(vf[n], FO::splat(cb.y), p2i);
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const float a1r = FO::get(p1r, r), a1i = FO::get(p1i, r);
          const float zr = FO::get(p2r, r) + row_shr<1>(a1r), zi = FO::get(p2i, r) + row_shr<1>(a1i);
          const float pw = (zr * zr + zi * zi) * scale2;
          const float pw16 = (a1r * a1r + a1i * a1i) * scale2;
          acc[r] += (mode == ISD_BP_MAGNITUDE) ? sqrtf(pw) : pw;
          acc16[r] += (mode == ISD_BP_MAGNITUDE) ? sqrtf(pw16) : pw16;
        }
      }
    }
    const float inv = khi >= klo ? 1.f / (float)(khi - klo + 1) : 0.f;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      float o0 = acc[r] * inv, o16 = acc16[r] * inv;
      if (mode == ISD_BP_LOGPOWER) { o0 = logf(o0 + eps); o16 = logf(o16 + eps); }
      const int64_t row = row0 + r * 4 + (lane >> 4);
      if (row < R) {
        const int64_t bt = row / C;
        const int ch = (int)(row - bt * C);
        float* o = feat + ((bt * nb_out + bmap[b]) * C + ch) * (int64_t)J;
        if (i < J) o[i] = o0;
        if (i == 15 && J == 17) o[16] = o16;
      }
    }
  }
}

// Fused spec-S extractor for long rows with heavily overlapped frames (stress configuration: 4096 samples,
// nperseg 1024, hop 64): the filterbank cascade of fb_kernel<VT, 4> (one row per wave, 2048 samples per pass, the
// section states carried across passes) followed, in registers, by the per-block DFT sums of the block-sum band
// power (stft.hip): every lane holds half of a 64-sample block, forms the half-block sums of the band's bins and
// their two neighbours, lane pairs are joined by one DPP shift and the 64 block sums of the row wait in LDS for
// blocksum_finish.  The filtered rows (86 MB per trial at the stress shape) are never written.
// The x row is re-read from global memory for every band and pass, and kLongShare one-wave workgroups share a row
// (each takes every kLongShare-th band).  Their ids are congruent mod 8 -- workgroups go to the 8 XCDs round-robin
// by id, and each XCD has its own L2 -- and lie within 64 consecutive ids, so the sharers run on one XCD at about
// the same time and the rows in flight there (~80 x 16 KiB) fit its 4 MiB L2.  With one workgroup per row walking
// all bands the rows in flight were ~10x the L2: the PMC counters showed 6.1 GB through the fabric for a 268 MB input.
constexpr int kLongShare = 8;                    // measured: 4 -> 4.58 ms, 8 -> 4.47 ms per 128 stress trials
template <typename VT, int KB>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4))) void fused_long_kernel(const FbSec* __restrict__ secs, const FbBand* __restrict__ bands,
                                                        const double* __restrict__ Qtab, const float2* __restrict__ blk,
                                                        const float* __restrict__ x, float* __restrict__ feat, int C,
                                                        int T, int nb, int ns, int J, int log2_nblk, int n_bins_max,
                                                        float scale2, FusedBands fbnd, int mode, float eps, int vec,
                                                        const int* __restrict__ bmap, int nb_out, int64_t n_rows) {
  using O = VOps<VT>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int lane = threadIdx.x;
  const int n_iter = (T + 4 * kSeg - 1) / (4 * kSeg);
  const int n_chunk = n_iter * 64;
  float2* Sblk = reinterpret_cast<float2*>(smem_raw);                      // [KB][64]
  float2* tw = Sblk + KB * 64;                                             // [64]  e^{-2 pi i u / nblk}
  double* carry = reinterpret_cast<double*>(tw + 64);                      // [ns][2]
  // id = 8 kLongShare q + 8 w + c  <->  row = 8 q + c, band subset w
  const int64_t id = blockIdx.x;
  const int64_t row = (id / (8 * kLongShare)) * 8 + (id & 7);
  const int share = (int)((id >> 3) % kLongShare);
  if (row >= n_rows) return;
  const int64_t bt = row / C;
  const int ch = (int)(row - bt * C);
  const int nblk = 1 << log2_nblk;
  const float* src = x + row * (int64_t)T;
  (void)n_chunk;
  if (lane < nblk) {
    float sn, cs;
    sincospif(2.f * (float)lane / (float)nblk, &sn, &cs);
    tw[lane] = make_float2(cs, -sn);
  }
  wave_lds_sync();
  for (int b = share; b < nb; b += kLongShare) {
    const int klo = fbnd.klo[b], khi = fbnd.khi[b];
    const int k0 = klo - 1, nbin = khi - klo + 1;
#pragma unroll
    for (int kk = 0; kk < KB; ++kk) Sblk[kk * 64 + lane] = make_float2(0.f, 0.f);
    if (lane < ns * 2) carry[lane] = 0.0;
    wave_lds_sync();
    const auto gain = O::g(bands[b]);
    for (int it = 0; it < n_iter; ++it) {
      typename O::Arr v;
      // the lane's 32 samples straight from global memory (128 contiguous bytes per lane, 8 KiB per wave; the row
      // is re-read once per band and stays in L2 / MALL): a row staged in LDS would cost 18 KiB per wave and hold
      // the CU at 7 waves
      const int e0 = (it * 64 + lane) * kL;
      const bool whole = vec && (it + 1) * 64 * kL <= T;          // wave-uniform: the pass lies inside the row
#pragma unroll
      for (int n = 0; n < kL; n += 4) {
        float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
        if (whole) {                                              // no per-element bounds arithmetic on the common path
          f = *reinterpret_cast<const float4*>(src + e0 + n);
        } else if (vec && e0 + n + 3 < T) {
          f = *reinterpret_cast<const float4*>(src + e0 + n);
        } else {
          if (e0 + n + 0 < T) f.x = src[e0 + n];
          if (e0 + n + 1 < T) f.y = src[e0 + n + 1];
          if (e0 + n + 2 < T) f.z = src[e0 + n + 2];
          if (e0 + n + 3 < T) f.w = src[e0 + n + 3];
        }
        O::put(v, n, (typename O::S)f.x * gain);
        O::put(v, n + 1, (typename O::S)f.y * gain);
        O::put(v, n + 2, (typename O::S)f.z * gain);
        O::put(v, n + 3, (typename O::S)f.w * gain);
      }
      for (int sct = 0; sct < ns; ++sct) {
        double c1[1] = {carry[sct * 2]}, c2[1] = {carry[sct * 2 + 1]};
        section<VT, 4>(v, secs[b * ns + sct], Qtab + (int64_t)(b * ns + sct) * 64, lane, c1, c2);
        wave_lds_sync();                                // every lane has read the incoming carry
        if (lane == 0) {
          carry[sct * 2] = c1[0];
          carry[sct * 2 + 1] = c2[0];
        }
      }
      wave_lds_sync();
      // the STFT sees y[0..T) then zeros, not the filter's ringing
      const int base = (it * 64 + lane) * kL;
      float vf[kL];
#pragma unroll
      for (int n = 0; n < kL; ++n) vf[n] = (float)O::at(v, n);
      if ((it + 1) * 64 * kL > T) {                     // only the last pass of a ragged row (wave-uniform)
#pragma unroll
        for (int n = 0; n < kL; ++n) vf[n] = base + n < T ? vf[n] : 0.f;
      }
      // half-block DFT sums with block-local phase (this lane's 32 samples start at offset 32*(lane&1) in the block)
      const float2 none = make_float2(0.f, 0.f);
      float2 P[KB];
#pragma unroll
      for (int kk = 0; kk < KB; ++kk) {
        __builtin_amdgcn_sched_barrier(0);              // one bin's 64 table scalars at a time in the SGPR file
        const int k = k0 + kk < n_bins_max ? k0 + kk : n_bins_max;
        const float2* tb = blk + (int64_t)k * 64;
        f2 acc = {0.f, 0.f};
#pragma unroll
        for (int n = 0; n < kL; ++n) acc = __builtin_elementwise_fma((f2){vf[n], vf[n]}, (f2){tb[n].x, tb[n].y}, acc);
        const float2 ph = tb[kL];                       // e^{-2 pi i k 32 / n}: the odd lane's offset in the block
        P[kk] = (lane & 1) ? make_float2(acc.x * ph.x - acc.y * ph.y, acc.x * ph.y + acc.y * ph.x)
                           : make_float2(acc.x, acc.y);
        (void)none;
      }
      const int m = it * 32 + (lane >> 1);
#pragma unroll
      for (int kk = 0; kk < KB; ++kk) {
        const float sx = P[kk].x + row_shl<1>(P[kk].x), sy = P[kk].y + row_shl<1>(P[kk].y);
        if (!(lane & 1)) Sblk[kk * 64 + m] = make_float2(sx, sy);
      }
    }
    wave_lds_sync();
    float2 S[KB];
#pragma unroll
    for (int kk = 0; kk < KB; ++kk) S[kk] = Sblk[kk * 64 + lane];
    blocksum_finish<KB>(S, tw, lane, k0, nbin, nblk, J, scale2, mode, eps,
                        feat + ((bt * nb_out + bmap[b]) * C + ch) * (int64_t)J);
    wave_lds_sync();
  }
}

static void mat2_mul(const double* a, const double* b, double* o) {
  double r[4] = {a[0] * b[0] + a[1] * b[2], a[0] * b[1] + a[1] * b[3], a[2] * b[0] + a[3] * b[2],
                 a[2] * b[1] + a[3] * b[3]};
  memcpy(o, r, sizeof(r));
}

}  // namespace isd

using namespace isd;

extern "C" int isd_fb_plan_destroy(isd_fb_plan* p);

extern "C" int isd_fb_plan_create(isd_fb_plan** out, int n_bands, int n_sections, const double* a12,
                                  const double* gain, int precision) {
  ISD_CHECK_ARG(out && a12 && gain, "isd_fb_plan_create: null argument");
  ISD_CHECK_ARG(n_bands >= 1 && n_bands <= 4096, "isd_fb_plan_create: n_bands=%d out of range", n_bands);
  ISD_CHECK_ARG(n_sections >= 1 && n_sections <= kMaxSec, "isd_fb_plan_create: n_sections=%d not in [1,%d]",
                n_sections, kMaxSec);
  ISD_CHECK_ARG(precision == ISD_FB_F32 || precision == ISD_FB_F64 || precision == ISD_FB_AUTO,
                "isd_fb_plan_create: bad precision %d", precision);
  const int n = n_bands * n_sections;
  std::vector<FbSec> secs(n);
  std::vector<FbBand> bands(n_bands);
  std::vector<double> Q((size_t)n * 64);
  std::vector<double> worst(n_bands, 0.0);
  for (int b = 0; b < n_bands; ++b) {
    bands[b].gd = gain[b];
    bands[b].gf = (float)gain[b];
    bands[b].pad = 0.f;
    for (int s = 0; s < n_sections; ++s) {
      const int bs = b * n_sections + s;
      const double a1 = a12[bs * 2], a2 = a12[bs * 2 + 1];
      // stability (poles strictly inside the unit circle)
      ISD_CHECK_ARG(a2 < 1.0 && a2 > -1.0 && fabs(a1) < 1.0 + a2,
                    "isd_fb_plan_create: band %d section %d is not stable (a1=%g a2=%g)", b, s, a1, a2);
      FbSec& sc = secs[bs];
      const double A[4] = {-a1, 1.0, -a2, 0.0};
      double An[4] = {1, 0, 0, 1};
      for (int k = 0; k < kL; ++k) {              // h[n] = row 0 of A^n ; afterwards An = A^32
        sc.hd[k][0] = An[0];
        sc.hd[k][1] = An[1];
        sc.hf[k][0] = (float)An[0];
        sc.hf[k][1] = (float)An[1];
        if (k < kL / 2) {
          sc.hq[0][k] = (float)An[0];
          sc.hq[1][k] = (float)An[1];
        }
        if (k == kL / 2)
          for (int e = 0; e < 4; ++e) sc.N16f[e] = (float)An[e];
        mat2_mul(A, An, An);
      }
      double Mk[4];
      memcpy(Mk, An, sizeof(Mk));
      double Qi[4] = {1, 0, 0, 1};
      for (int i = 0; i < 16; ++i) {              // Q_i = M^i
        memcpy(&Q[(size_t)bs * 64 + i * 4], Qi, sizeof(Qi));
        mat2_mul(Mk, Qi, Qi);
      }
      memcpy(sc.P, Qi, sizeof(Qi));               // M^16
      for (int k = 0; k < 4; ++k) {               // M^(1,2,4,8)
        memcpy(sc.Mp[k], Mk, sizeof(Mk));
        mat2_mul(Mk, Mk, Mk);
      }
      sc.a1d = a1; sc.a2d = a2; sc.a1f = (float)a1; sc.a2f = (float)a2;
      // fp32 round-off amplification of a resonator ~ 1 / ((1-r) sin(theta))
      if (a2 > 0.0) {
        const double r = sqrt(a2);
        double c = -a1 / (2.0 * r);
        c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
        const double st = sqrt(1.0 - c * c);
        const double g = 1.0 / ((1.0 - r) * (st > 1e-9 ? st : 1e-9));
        if (g > worst[b]) worst[b] = g;
      }
    }
  }
  isd_fb_plan* p = new isd_fb_plan();
  p->n_bands = n_bands; p->n_sections = n_sections;
  for (int k = 0; k < 2; ++k) {
    p->set[k] = FbSet{0, nullptr, nullptr, nullptr, nullptr};
    p->host_map[k] = nullptr;
  }
  // per-band arithmetic: AUTO sends a band to the fp64 set when one of its poles is too close to z = 1
  std::vector<int> idx[2];
  for (int b = 0; b < n_bands; ++b) {
    const int k = precision == ISD_FB_AUTO ? (worst[b] > 2000.0 ? 1 : 0) : (precision == ISD_FB_F64 ? 1 : 0);
    idx[k].push_back(b);
  }
  p->precision = idx[1].empty() ? ISD_FB_F32 : (idx[0].empty() ? ISD_FB_F64 : ISD_FB_MIXED);
  hipError_t e = hipSuccess;
  for (int k = 0; k < 2 && e == hipSuccess; ++k) {
    const int nb = (int)idx[k].size();
    if (!nb) continue;
    std::vector<FbSec> ss((size_t)nb * n_sections);
    std::vector<FbBand> bb(nb);
    std::vector<double> qq((size_t)nb * n_sections * 64);
    for (int i = 0; i < nb; ++i) {
      const int b = idx[k][i];
      bb[i] = bands[b];
      for (int sct = 0; sct < n_sections; ++sct) {
        ss[(size_t)i * n_sections + sct] = secs[(size_t)b * n_sections + sct];
        memcpy(&qq[((size_t)i * n_sections + sct) * 64], &Q[((size_t)b * n_sections + sct) * 64], sizeof(double) * 64);
      }
    }
    FbSet& fs = p->set[k];
    fs.nb = nb;
    p->host_map[k] = new int[nb];
    memcpy(p->host_map[k], idx[k].data(), sizeof(int) * nb);
    e = hipMalloc(&fs.d_sec, sizeof(FbSec) * ss.size());
    if (e == hipSuccess) e = hipMalloc(&fs.d_band, sizeof(FbBand) * nb);
    if (e == hipSuccess) e = hipMalloc(&fs.d_Q, sizeof(double) * qq.size());
    if (e == hipSuccess) e = hipMalloc(&fs.d_map, sizeof(int) * nb);
    if (e == hipSuccess) e = hipMemcpy(fs.d_sec, ss.data(), sizeof(FbSec) * ss.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(fs.d_band, bb.data(), sizeof(FbBand) * nb, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(fs.d_Q, qq.data(), sizeof(double) * qq.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(fs.d_map, idx[k].data(), sizeof(int) * nb, hipMemcpyHostToDevice);
  }
  if (e != hipSuccess) {
    set_error("isd_fb_plan_create: %s", hipGetErrorString(e));
    isd_fb_plan_destroy(p);
    return e == hipErrorNoDevice ? ISD_ERR_NO_DEVICE : ISD_ERR_HIP;
  }
  *out = p;
  return ISD_OK;
}

extern "C" int isd_fb_plan_destroy(isd_fb_plan* p) {
  if (!p) return ISD_OK;
  for (int k = 0; k < 2; ++k) {
    FbSet& fs = p->set[k];
    if (fs.d_sec) (void)hipFree(fs.d_sec);
    if (fs.d_band) (void)hipFree(fs.d_band);
    if (fs.d_Q) (void)hipFree(fs.d_Q);
    if (fs.d_map) (void)hipFree(fs.d_map);
    delete[] p->host_map[k];
  }
  delete p;
  return ISD_OK;
}

extern "C" int isd_fb_plan_precision(const isd_fb_plan* p) { return p ? p->precision : ISD_ERR_INVALID; }

template <typename VT, int GPR>
static int fb_launch(const isd_fb_plan* p, const FbSet& fs, const float* x, float* y, int64_t R, int C, int T,
                     hipStream_t st) {
  constexpr int NR = VOps<VT>::NR;
  const int64_t items = cdiv(R, (int64_t)NR * (4 / GPR));
  ISD_CHECK_ARG(items <= 0x7fffffffLL, "isd_fb_forward: too many rows (%lld)", (long long)R);
  const size_t lds = 4 * 16 * kPad * 4 + (GPR == 4 ? (size_t)NR * fs.nb * p->n_sections * 2 * 8 : 0);
  ISD_CHECK_ARG(lds <= 64 * 1024, "isd_fb_forward: n_bands*n_sections too large for the carry tile");
  const int vec = ((T & 3) == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) &&
                  ((reinterpret_cast<uintptr_t>(y) & 15) == 0);
  hipLaunchKernelGGL((fb_kernel<VT, GPR>), dim3((unsigned)items), dim3(64), lds, st, fs.d_sec, fs.d_band, fs.d_Q, x, y,
                     R, C, T, fs.nb, p->n_sections, vec, fs.d_map, p->n_bands);
  ISD_LAUNCH_CHECK();
  return ISD_OK;
}

template <typename VT>
static int fb_launch_t(const isd_fb_plan* p, const FbSet& fs, const float* x, float* y, int64_t R, int C, int T,
                       hipStream_t st) {
  if (T <= kSeg) return fb_launch<VT, 1>(p, fs, x, y, R, C, T, st);
  if (T <= 2 * kSeg) return fb_launch<VT, 2>(p, fs, x, y, R, C, T, st);
  return fb_launch<VT, 4>(p, fs, x, y, R, C, T, st);
}

extern "C" int isd_fb_forward(const isd_fb_plan* p, const float* x, float* y, int64_t B, int64_t C, int64_t T,
                              void* stream) {
  ISD_CHECK_ARG(p, "isd_fb_forward: null plan");
  ISD_CHECK_ARG(B >= 0 && C >= 1 && T >= 1 && T <= (1 << 24) && C <= (1 << 20), "isd_fb_forward: bad shape B=%lld C=%lld T=%lld",
                (long long)B, (long long)C, (long long)T);
  if (B == 0) return ISD_OK;
  ISD_CHECK_ARG(x && y, "isd_fb_forward: null argument");
  hipStream_t st = (hipStream_t)stream;
  const int64_t R = B * C;
  int rc = ISD_OK;
  if (p->set[0].nb) rc = fb_launch_t<float>(p, p->set[0], x, y, R, (int)C, (int)T, st);
  if (rc == ISD_OK && p->set[1].nb) rc = fb_launch_t<double>(p, p->set[1], x, y, R, (int)C, (int)T, st);
  return rc;
}

template <typename VT>
static int fused_launch(const isd_fb_plan* fb, const FbSet& fs, const isd_stft_plan* st, const float* x, float* feat,
                        int64_t R, int C, const FusedBands& fbnd, int mode, float eps, hipStream_t stream) {
  const int64_t items = cdiv(R, 4 * VOps<VT>::NR);
  ISD_CHECK_ARG(items <= 0x7fffffffLL, "isd_features_fused: too many rows (%lld)", (long long)R);
  const int vec = ((st->T & 3) == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  hipLaunchKernelGGL((fused_kernel<VT>), dim3((unsigned)items), dim3(64), 0, stream, fs.d_sec, fs.d_band, st->d_dft, x,
                     feat, R, C, st->T, fs.nb, fb->n_sections, st->J, st->scale * st->scale, fbnd, mode, eps, vec,
                     fs.d_map, fb->n_bands);
  ISD_LAUNCH_CHECK();
  return ISD_OK;
}

extern "C" int isd_features_fused(const isd_fb_plan* fb, const isd_stft_plan* st, const float* x, float* feat,
                                  int64_t B, int64_t C, const int* klo, const int* khi, int mode, float eps,
                                  void* stream) {
  ISD_CHECK_ARG(fb && st, "isd_features_fused: null plan");
  ISD_CHECK_ARG(B == 0 || (x && feat), "isd_features_fused: null argument");
  ISD_CHECK_ARG(B >= 0 && C >= 1 && C <= (1 << 20), "isd_features_fused: bad shape B=%lld C=%lld", (long long)B,
                (long long)C);
  ISD_CHECK_ARG(mode >= ISD_BP_MAGNITUDE && mode <= ISD_BP_LOGPOWER, "isd_features_fused: bad mode %d", mode);
  const bool short_rows = st->n == 64 && st->hop == 32 && st->T <= kSeg && st->d_dft;
  const bool long_rows = st->d_blk && st->hop == 64 && st->T <= 64 * 64;
  if (!short_rows && !long_rows) {
    set_error("isd_features_fused: needs nperseg=64/noverlap=32/T<=512, or hop 64 with nperseg = 2^a*64 and T<=4096 "
              "(got nperseg=%d hop=%d T=%d)", st->n, st->hop, st->T);
    return ISD_ERR_UNSUPPORTED;
  }
  FusedBands all = {};
  int rc = fill_band_args(st, fb->n_bands, klo, khi, all.klo, all.khi, "isd_features_fused");
  if (rc) return rc;
  if (B == 0) return ISD_OK;
  hipStream_t s = (hipStream_t)stream;
  if (!short_rows) {
    // long rows, heavily overlapped frames: filterbank + block sums in one kernel per band set
    int nbmax = 0;
    for (int b = 0; b < fb->n_bands; ++b) {
      const int nbin = all.khi[b] - all.klo[b] + 1;
      if (nbin < 1 || nbin > 6 || all.klo[b] < 1 || all.khi[b] > st->n / 2 - 1) {
        set_error("isd_features_fused: band %d needs 1..6 interior bins for the block-sum path (bins %d..%d)", b,
                  all.klo[b], all.khi[b]);
        return ISD_ERR_UNSUPPORTED;
      }
      if (nbin > nbmax) nbmax = nbin;
    }
    const int64_t rows = B * C;
    ISD_CHECK_ARG(rows <= 0x7fffffffLL / 16, "isd_features_fused: too many rows (%lld)", (long long)rows);
    int log2_nblk = 0;
    while ((64 << log2_nblk) < st->n) ++log2_nblk;
    const int KB = nbmax + 2 <= 4 ? 4 : nbmax + 2 <= 5 ? 5 : nbmax + 2 <= 6 ? 6 : 8;   // band bins + two neighbours
    const int n_iter = (st->T + 4 * kSeg - 1) / (4 * kSeg);
    const int vec = ((st->T & 3) == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
    for (int k = 0; k < 2; ++k) {
      const FbSet& fs = fb->set[k];
      if (!fs.nb) continue;
      FusedBands fbnd = {};
      for (int i = 0; i < fs.nb; ++i) {
        fbnd.klo[i] = all.klo[fb->host_map[k][i]];
        fbnd.khi[i] = all.khi[fb->host_map[k][i]];
      }
      const size_t lds = sizeof(float2) * ((size_t)KB * 64 + 64) + sizeof(double) * 2 * kMaxSec;
#define ISD_FL_LAUNCH(VT, K)                                                                                      \
  hipLaunchKernelGGL((fused_long_kernel<VT, K>), dim3((unsigned)(cdiv(rows, 8) * 8 * kLongShare)), dim3(64), lds, s, \
                     fs.d_sec, fs.d_band, fs.d_Q, st->d_blk, x, feat, (int)C, st->T, fs.nb, fb->n_sections, st->J,  \
                     log2_nblk, st->n / 2, st->scale * st->scale, fbnd, mode, eps, vec, fs.d_map, fb->n_bands, rows)
      if (k == 0) {
        if (KB == 4) ISD_FL_LAUNCH(float, 4); else if (KB == 5) ISD_FL_LAUNCH(float, 5);
        else if (KB == 6) ISD_FL_LAUNCH(float, 6); else ISD_FL_LAUNCH(float, 8);
      } else {
        if (KB == 4) ISD_FL_LAUNCH(double, 4); else if (KB == 5) ISD_FL_LAUNCH(double, 5);
        else if (KB == 6) ISD_FL_LAUNCH(double, 6); else ISD_FL_LAUNCH(double, 8);
      }
#undef ISD_FL_LAUNCH
      ISD_LAUNCH_CHECK();
    }
    return ISD_OK;
  }
  for (int k = 0; k < 2; ++k) {
    const FbSet& fs = fb->set[k];
    if (!fs.nb) continue;
    FusedBands fbnd = {};                                  // the set's bands, in the set's order
    for (int i = 0; i < fs.nb; ++i) {
      fbnd.klo[i] = all.klo[fb->host_map[k][i]];
      fbnd.khi[i] = all.khi[fb->host_map[k][i]];
    }
    rc = k ? fused_launch<double>(fb, fs, st, x, feat, B * C, (int)C, fbnd, mode, eps, s)
           : fused_launch<float>(fb, fs, st, x, feat, B * C, (int)C, fbnd, mode, eps, s);
    if (rc) return rc;
  }
  return ISD_OK;
}

int isd::bandpower_direct(const isd_stft_plan* st, const float* y, float* feat, int64_t R, int C, int nb,
                          const int* klo, const int* khi, int mode, float eps, hipStream_t stream) {
  FusedBands fbnd = {};
  for (int b = 0; b < nb; ++b) { fbnd.klo[b] = klo[b]; fbnd.khi[b] = khi[b]; }
  const int64_t items = cdiv(R, 4);
  ISD_CHECK_ARG(items <= 0x7fffffffLL, "isd_stft_bandpower: too many rows (%lld)", (long long)R);
  const int vec = ((st->T & 3) == 0) && ((reinterpret_cast<uintptr_t>(y) & 15) == 0);
  hipLaunchKernelGGL(bandpower_direct_kernel, dim3((unsigned)items), dim3(64), 0, stream, st->d_dft, y, feat, R, C,
                     st->T, nb, st->J, st->scale * st->scale, fbnd, mode, eps, vec);
  ISD_LAUNCH_CHECK();
  return ISD_OK;
}
