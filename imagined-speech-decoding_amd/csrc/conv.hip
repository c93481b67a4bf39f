// Zone-wise Conv4Layers stack (reference: src/fast/models/fast.py:103-119 `Conv4Layers`,
// :199-210 `Head`, :242-252 `FAST.forward_head`) forward + backward on gfx950.
//
//   cnn1 (1->F, 1x5, bias, valid) and cnn2 (F->F, Cz x 1) have no nonlinearity between
//   them, so they are applied as ONE (Cz x 5)-tap convolution with
//       Weff[g,c,k] = sum_f W2[g,f,c] W1[f,k],   beff[g] = sum_{f,c} W2[g,f,c] b1[f];
//   the [B',F,Cz,246] cnn1 activation (32x the input) is never formed.  The backward
//   pass produces dWeff/dbeff and chains them to cnn1.weight / cnn1.bias / cnn2.weight.
//   cnn3, cnn4: F->F, 5 taps, zero pad 2.  Then exact-erf GELU and the mean over time.
//
// All convolutions are GEMM-shaped (K = 5*Cin) and run on v_mfma_f32_16x16x4_f32
// (exact fp32 FMA chains).  M = output filters (16-row tiles), N = 16 time steps,
// K = 4 input channels per MFMA, one MFMA per tap.  Sliding windows and the zone
// gather are index arithmetic on the raw trial tensor; nothing is copied.
#include "common.h"
#include <math.h>
#include <string.h>
#include <type_traits>
#include <vector>

namespace isd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kTaps = 5;
constexpr int kCK = 32;          // input channels staged per chunk
constexpr int kMaxZones = 64;

struct ZoneDesc {
  int cin;          // channels of this zone (Cz)
  int idx_off;      // offset into the channel-index table
  int64_t p_off;    // offset of the zone's parameters in the flat parameter block
  int64_t eff_off;  // offset of Weff frag block (fwd) in the weight workspace
  int64_t wg_off;   // offset of the zone's dWeff block [F][Cz+1][5] in the wgrad result
};

// Activation storage type: float, or bf16 (config 3: bf16 activations / activation gradients, fp32
// accumulate).  Values are rounded to bf16 (RNE) where they are stored and where operands are staged, so
// the products equal those of a bf16 MFMA with fp32 accumulation; the MFMA itself stays the fp32 one.
struct bf16_t { unsigned short v; };
__device__ __forceinline__ float bf16_round(float x) {
  unsigned int u = __float_as_uint(x);
  if ((u & 0x7fffffffu) > 0x7f800000u) return x;                     // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return __uint_as_float(u & 0xffff0000u);
}
// two floats -> packed bf16 pair (RNE; v_cvt_pk_bf16_f32), low half = a
__device__ __forceinline__ unsigned int bf16_pack(float a, float b) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  typedef float fl2 __attribute__((ext_vector_type(2)));
  const bf2 r = __builtin_convertvector((fl2){a, b}, bf2);
  return __builtin_bit_cast(unsigned int, r);
}
__device__ __forceinline__ float bf16_lo(unsigned int u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned int u) { return __uint_as_float(u & 0xffff0000u); }
typedef short bf16x8 __attribute__((ext_vector_type(8)));   // MFMA operand: 8 bf16 in 4 VGPRs
template <typename AT> struct Act;
template <> struct Act<float> {
  static constexpr bool kBf16 = false;
  static __device__ __forceinline__ float ld(const void* p, int64_t i) { return ((const float*)p)[i]; }
  static __device__ __forceinline__ void st(void* p, int64_t i, float v) { ((float*)p)[i] = v; }
  static __device__ __forceinline__ float rnd(float v) { return v; }
};
template <> struct Act<bf16_t> {
  static constexpr bool kBf16 = true;
  static __device__ __forceinline__ float ld(const void* p, int64_t i) {
    return __uint_as_float((unsigned int)((const unsigned short*)p)[i] << 16);
  }
  static __device__ __forceinline__ void st(void* p, int64_t i, float v) {
    ((unsigned short*)p)[i] = (unsigned short)(__float_as_uint(bf16_round(v)) >> 16);
  }
  static __device__ __forceinline__ float rnd(float v) { return bf16_round(v); }
};

__device__ __forceinline__ float gelu_f(float x) { return gelu_fast(x); }               // common.h: A&S erf, 1.5e-7
__device__ __forceinline__ float gelu_grad_f(float x) { return gelu_grad_fast(x); }

// cnn3 / cnn4 weights [F][F][5] -> frag order, forward and transposed+flipped (dgrad) copies; `nbx` blocks of 256
// threads share one (zone, layer), this is block `bx` of them.
__device__ __forceinline__ void prep_conv_body(const float* __restrict__ params, const ZoneDesc& zd, int z,
                                               float* __restrict__ wf, float* __restrict__ wt, int F, int layer,
                                               int64_t zstride, int bf16, int bx, int nbx, int k32) {
  const int GT = F / 16;
  const int ncg = F / 4;
  const float* W = params + zd.p_off + F * kTaps + F + (int64_t)F * F * zd.cin + (int64_t)layer * F * F * kTaps;
  if (k32) {
    // bf16 matrix cores (F = 32): block (k, gt) holds, for lane l, the 8 bf16 W[gt*16 + (l&15)][8 (l>>4) + j][k] --
    // one v_mfma_f32_16x16x32_bf16 A fragment per tap covers all 32 input channels
    uint4* wf16 = reinterpret_cast<uint4*>(wf + z * zstride);
    uint4* wt16 = reinterpret_cast<uint4*>(wt + z * zstride);
    for (int e = bx * 256 + threadIdx.x; e < kTaps * GT * 64; e += nbx * 256) {
      const int lane = e & 63, blk = e >> 6;
      const int gt = blk % GT, k = blk / GT;
      const int g = gt * 16 + (lane & 15), c0 = 8 * (lane >> 4);
      float a[8], b[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        a[j] = W[(g * F + c0 + j) * kTaps + k];
        b[j] = W[((c0 + j) * F + g) * kTaps + (kTaps - 1 - k)];
      }
      wf16[e] = make_uint4(bf16_pack(a[0], a[1]), bf16_pack(a[2], a[3]), bf16_pack(a[4], a[5]), bf16_pack(a[6], a[7]));
      wt16[e] = make_uint4(bf16_pack(b[0], b[1]), bf16_pack(b[2], b[3]), bf16_pack(b[4], b[5]), bf16_pack(b[6], b[7]));
    }
    return;
  }
  const int total = ncg * kTaps * GT * 64;
  for (int e = bx * 256 + threadIdx.x; e < total; e += nbx * 256) {
    const int lane = e & 63;
    const int blk = e >> 6;
    const int gt = blk % GT;
    const int k = (blk / GT) % kTaps;
    const int cg = blk / (GT * kTaps);
    const int g = gt * 16 + (lane & 15);
    const int c = cg * 4 + (lane >> 4);
    const float a = W[(g * F + c) * kTaps + k];
    const float b = W[(c * F + g) * kTaps + (kTaps - 1 - k)];        // dIn[g] <- dOut[c], flipped taps
    wf[z * zstride + e] = bf16 ? bf16_round(a) : a;
    wt[z * zstride + e] = bf16 ? bf16_round(b) : b;
  }
}

// ---------------------------------------------------------------------------------------
// Weight preparation.  "frag order": block (cg, k, gt) holds, for lane l,
//   W[gt*16 + (l&15)][4*cg + (l>>4)][k]   (zero beyond Cin)   -> one coalesced A-fragment load.
// ---------------------------------------------------------------------------------------
// One launch prepares everything a step needs (short dependent launches cost ~4 us of hand-over each on top of
// their run time): blocks [0, nbw) the fused cnn1 o cnn2 weights, [nbw, nbw + F) its bias, and -- with
// n_layers == 4 -- 4 blocks per layer behind them the cnn3 / cnn4 fragment tables.
struct PrepConvArgs {
  float *w3, *w3t, *w4, *w4t;
  int64_t zstride;
  int n_layers;
  int k32;          // cnn3 / cnn4 as K = 32 bf16 fragments (the bf16 fused kernels) instead of fp32 fragment order
};
// `wfrag16` (nullable): the same fused first-layer weights as bf16 MFMA A fragments in "tap-window" order (the
// bf16 first-layer kernels below): block (cg, gt) holds, for lane l, the 8 bf16
//   W[gt*16 + (l&15)][4*cg + (l>>4)][tap j],  j = 0..4, then three zeros            (16 bytes per lane).
__global__ __launch_bounds__(256) void prep_fused_kernel(const float* __restrict__ params,
                                                         const ZoneDesc* __restrict__ zones,
                                                         float* __restrict__ wfrag, float* __restrict__ beff, int F,
                                                         int nbw, int bf16, PrepConvArgs pc,
                                                         uint4* __restrict__ wfrag16) {
  __shared__ float red[256];
  const int z = blockIdx.y;
  const ZoneDesc zd = zones[z];
  if ((int)blockIdx.x >= nbw + F) {
    const int r = blockIdx.x - (nbw + F), layer = r >> 2;
    prep_conv_body(params, zd, z, layer ? pc.w4 : pc.w3, layer ? pc.w4t : pc.w3t, F, layer, pc.zstride, bf16, r & 3, 4,
                   pc.k32);
    return;
  }
  const int GT = F / 16;
  const int ncg = (zd.cin + 3) / 4;
  const float* W1 = params + zd.p_off;                 // [F][5]
  const float* b1 = W1 + F * kTaps;                    // [F]
  const float* W2 = b1 + F;                            // [F][F][Cz]
  if ((int)blockIdx.x < nbw) {
    // one thread per (filter g, channel c), lanes along c: every W2 load of a wave is one coalesced row segment;
    // the thread forms all 5 taps and scatters them into the fragment order
    const int e = blockIdx.x * 256 + threadIdx.x;
    const int c4 = ncg * 4;
    if (e >= F * c4) return;
    const int g = e / c4, c = e - g * c4;
    float acc[kTaps];
#pragma unroll
    for (int k = 0; k < kTaps; ++k) acc[k] = 0.f;
    if (c < zd.cin) {
      const float* w2 = W2 + (int64_t)g * F * zd.cin + c;
#pragma unroll 16
      for (int f = 0; f < F; ++f) {
        const float w = w2[(int64_t)f * zd.cin];
#pragma unroll
        for (int k = 0; k < kTaps; ++k) acc[k] = fmaf(w, W1[f * kTaps + k], acc[k]);
      }
    }
    const int cg = c >> 2, lane = (c & 3) * 16 + (g & 15), gt = g >> 4;
#pragma unroll
    for (int k = 0; k < kTaps; ++k) {
      const float v = acc[k];
      wfrag[zd.eff_off + ((int64_t)(cg * kTaps + k) * GT + gt) * 64 + lane] = bf16 ? bf16_round(v) : v;
    }
    if (wfrag16) {
      uint4 w;
      w.x = bf16_pack(acc[0], acc[1]);
      w.y = bf16_pack(acc[2], acc[3]);
      w.z = bf16_pack(acc[4], 0.f);
      w.w = 0u;
      wfrag16[zd.eff_off / kTaps + ((int64_t)cg * GT + gt) * 64 + lane] = w;
    }
    return;
  }
  const int g = blockIdx.x - nbw;                      // one block per beff[g]
  if (g >= F) return;
  float s = 0.f;
  {
    const float* w2 = W2 + (int64_t)g * F * zd.cin;
    for (int c = threadIdx.x; c < zd.cin; c += 256) {      // independent loads, no division: they pipeline
#pragma unroll 8
      for (int f = 0; f < F; ++f) s = fmaf(w2[(int64_t)f * zd.cin + c], b1[f], s);
    }
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) beff[z * F + g] = red[0];
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

__device__ __forceinline__ void glds_copy16(const float* __restrict__ src, float* dst, int n4, int lane) {
  // wave-cooperative copy of n4 float4 (contiguous both sides); dst must be the wave-uniform base
  for (int e0 = 0; e0 < n4; e0 += 64) {
    if (e0 + lane < n4)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + (int64_t)(e0 + lane) * 4), (lds_ptr_t)(dst + e0 * 4), 16, 0, 0);
  }
}

// Same copy shared by `step/64` waves: a wave takes the 1 KiB pieces first, first + step, ... (first = 64 * wave
// index within the copying group).  All pieces are in flight at once; the next __syncthreads() retires them.
__device__ __forceinline__ void glds_copy16_strided(const float* __restrict__ src, float* dst, int n4, int first,
                                                    int step, int lane) {
  for (int e0 = first; e0 < n4; e0 += step) {
    if (e0 + lane < n4)
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + (int64_t)(e0 + lane) * 4), (lds_ptr_t)(dst + e0 * 4), 16, 0, 0);
  }
}

// ---------------------------------------------------------------------------------------
// Forward convolution (also used as dgrad with transposed weights).
// ---------------------------------------------------------------------------------------
struct ConvArgs {
  const void* in;            // MODE 0: fp32 trials; MODE 1: activations of type AT
  void* out;                 // activations of type AT
  const float* wfrag;        // frag-ordered weights (per-zone offset from ZoneDesc or z*wz_stride)
  const float* bias;         // [Z][F] or null
  const ZoneDesc* zones;
  const int* chan_idx;
  int64_t wz_stride;         // MODE 1: per-zone stride of wfrag
  int64_t items;             // B' = B * N
  int Z, F, Tin, Tout, pad, TT, IPW, RS;
  int CK;                    // conv5_fwd_kernel: channels staged per chunk (multiple of 4, <= 32)
  int lin;                   // 1: a chunk's rows are one contiguous block in global memory and RS == Tin
  int Ctot, Tx, N, S;        // MODE 0 only
};

// MODE 0: input gathered from the raw trials  x[b][chan_idx[c]][n*S + t]   (item = b*N + n)
// MODE 1: input is an activation tensor      in[((item*Z + z)*F + c)*Tin + t]
// LDS rows are unpadded copies of the source rows (stride RS >= Tin); zero padding, the ragged last
// time tile and the channel round-up are handled by masking the B fragment, so staging moves only
// real data (float4 when the block is contiguous and aligned).
template <int MODE, typename AT, int GT, int NT>
__global__ __launch_bounds__(256) void conv5_fwd_kernel(ConvArgs a) {
  using IT = typename std::conditional<MODE == 0, float, AT>::type;   // storage type of the input
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int z = blockIdx.y;
  const ZoneDesc zd = a.zones[z];
  const int cin = (MODE == 0) ? zd.cin : a.F;
  const int CK = a.CK;                                     // channels staged per chunk (32, or fewer for long rows)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t item0 = (int64_t)blockIdx.x * a.IPW;
  const int n_items = (int)((a.items - item0) < a.IPW ? (a.items - item0) : a.IPW);
  const int n_ct = n_items * a.TT;
  float* in_tile = smem + 4;                               // [IPW][CK][RS]; 4 floats of slack: masked reads may
                                                           // address up to `pad` elements before row 0
  float* w_tile = smem + 4 + ((a.IPW * CK * a.RS + 3) & ~3);  // [CK/4][5][GT][64], 16-byte aligned
  const float* wbase = a.wfrag + ((MODE == 0) ? zd.eff_off : (int64_t)z * a.wz_stride);
  const int n_chunks = (cin + CK - 1) / CK;
  const int q = lane >> 4, jl = lane & 15;

  for (int base = 0; base < n_ct; base += 4 * NT) {
    // every wave always carries NT column tiles: no branch stands between the MFMAs, slots past the end
    // recompute tile 0 and are not stored
    f32x4 acc[NT][GT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int g = 0; g < GT; ++g) acc[j][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    struct Frag {
      float af[kTaps][GT];
      float bf[NT][kTaps];
    };
    int t_ii[NT], t_t0[NT];
    bool t_ok[NT];
    int boff[NT];                                          // per-tile LDS offset of (row q, t0 + jl - pad)
    bool okk[NT][kTaps];                                   // per-tile, per-tap validity of the B element
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int ct = base + j * 4 + wave;                  // round-robin: few tiles still use every wave
      t_ok[j] = ct < n_ct;
      const int ctc = t_ok[j] ? ct : 0;
      t_ii[j] = ctc / a.TT;
      t_t0[j] = (ctc - t_ii[j] * a.TT) * 16;
      boff[j] = t_ii[j] * CK * a.RS + t_t0[j] + jl - a.pad + q * a.RS;
#pragma unroll
      for (int k = 0; k < kTaps; ++k) {
        const int idx = t_t0[j] + jl + k - a.pad;
        okk[j][k] = idx >= 0 && idx < a.Tin;
      }
    }
    for (int ch = 0; ch < n_chunks; ++ch) {
      const int c_lo = ch * CK;
      const int ckc = (cin - c_lo) < CK ? (cin - c_lo) : CK;
      const int ckc4 = (ckc + 3) & ~3;
      __syncthreads();
      if (a.lin) {
        // one contiguous block of ckc*Tin elements per item
        const int cnt = ckc * a.Tin;
        // few items: the whole workgroup copies each block; many items: one wave per item
        const int tid0 = n_items < 4 ? (int)threadIdx.x : lane;
        const int tstep = n_items < 4 ? 256 : 64;
        for (int ii = n_items < 4 ? 0 : wave; ii < n_items; ii += n_items < 4 ? 1 : 4) {
          const int64_t item = item0 + ii;
          const int64_t soff = (MODE == 0) ? (item * a.Ctot + a.chan_idx[zd.idx_off + c_lo]) * (int64_t)a.Tx
                                           : ((item * a.Z + z) * a.F + c_lo) * (int64_t)a.Tin;
          float* dst = in_tile + ii * CK * a.RS;
          if constexpr (std::is_same<IT, float>::value && !Act<AT>::kBf16) {
            const float* src = (const float*)a.in + soff;
            if ((((uintptr_t)src | (uintptr_t)dst) & 15) == 0 && (cnt & 3) == 0) {
              glds_copy16_strided(src, dst, cnt >> 2, tid0 - lane, tstep, lane);   // LDS-DMA: every piece in flight at once
            } else {
              for (int e = tid0; e < cnt; e += tstep) dst[e] = src[e];
            }
          } else {
            for (int e = tid0; e < cnt; e += tstep) dst[e] = Act<AT>::rnd(Act<IT>::ld(a.in, soff + e));
          }
        }
      } else {
        // one wave per row, lanes along time (coalesced segments)
        const int rows = n_items * ckc;
        for (int r = wave; r < rows; r += 4) {
          const int ii = r / ckc, cc = r - ii * ckc;
          const int64_t item = item0 + ii;
          int64_t soff;
          if (MODE == 0) {
            const int64_t b = item / a.N;
            const int n = (int)(item - b * a.N);
            soff = (b * a.Ctot + a.chan_idx[zd.idx_off + c_lo + cc]) * (int64_t)a.Tx + (int64_t)n * a.S;
          } else {
            soff = ((item * a.Z + z) * a.F + c_lo + cc) * (int64_t)a.Tin;
          }
          float* dst = in_tile + (ii * CK + cc) * a.RS;
          for (int t = lane; t < a.Tin; t += 64) dst[t] = Act<AT>::rnd(Act<IT>::ld(a.in, soff + t));
        }
      }
      if (ckc4 != ckc) {                                   // channel round-up rows read as zero
        for (int e = threadIdx.x; e < n_items * (ckc4 - ckc) * a.RS; e += 256) {
          const int ii = e / ((ckc4 - ckc) * a.RS), r = e - ii * (ckc4 - ckc) * a.RS;
          in_tile[(ii * CK + ckc) * a.RS + r] = 0.f;
        }
      }
      {
        const int wlen4 = (ckc4 / 4) * kTaps * GT * 16;
        glds_copy16_strided(wbase + (int64_t)ch * (CK / 4) * kTaps * GT * 64, w_tile, wlen4, wave * 64, 256, lane);
      }
      __syncthreads();
      // addresses and masks were hoisted out of this loop (boff / okk); the channel round-up rows are zero.
      // Fragments of channel group cg+1 are fetched from LDS while the MFMAs of group cg run.
      const float* wl = w_tile + lane;
      auto load = [&](int cg, Frag& f) {
        const float* rowp = in_tile + cg * 4 * a.RS;
#pragma unroll
        for (int k = 0; k < kTaps; ++k)
#pragma unroll
          for (int g = 0; g < GT; ++g) f.af[k][g] = wl[((cg * kTaps + k) * GT + g) * 64];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int k = 0; k < kTaps; ++k) {
            const float v = rowp[boff[j] + k];                 // may touch a neighbouring row: masked
            f.bf[j][k] = okk[j][k] ? v : 0.f;
          }
      };
      auto mma = [&](const Frag& f) {
#pragma unroll
        for (int k = 0; k < kTaps; ++k)
#pragma unroll
          for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int g = 0; g < GT; ++g)
              acc[j][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.af[k][g], f.bf[j][k], acc[j][g], 0, 0, 0);
      };
      const int ncg = ckc4 / 4;
      Frag f0, f1;
      load(0, f0);
      for (int cg = 0; cg < ncg; cg += 2) {
        if (cg + 1 < ncg) load(cg + 1, f1);
        mma(f0);
        if (cg + 1 < ncg) {
          if (cg + 2 < ncg) load(cg + 2, f0);
          mma(f1);
        }
      }
    }
    // epilogue: D[row g = 4*(lane>>4)+r][col t = lane&15]
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      if (!t_ok[j]) continue;
      const int64_t item = item0 + t_ii[j];
      const int t = t_t0[j] + jl;
      if (t >= a.Tout) continue;
#pragma unroll
      for (int gt = 0; gt < GT; ++gt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int g = gt * 16 + 4 * q + r;
          float v = acc[j][gt][r];
          if (a.bias) v += a.bias[z * a.F + g];
          Act<AT>::st(a.out, ((item * a.Z + z) * a.F + g) * (int64_t)a.Tout + t, v);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// First-layer forward for wide inputs (spec-S features: hundreds of channels, a handful of time steps):
// same tiling as conv5_fwd_kernel<0, float> on contiguous rows, but the input and weight chunks are
// double-buffered in LDS and filled by LDS-DMA (global_load_lds_dwordx4), so the copy of chunk c+1 runs under
// the MFMAs of chunk c.  One workgroup per CU (1 wave per SIMD, 4 column tiles x 2 filter tiles of
// accumulators per wave); one barrier per chunk: it retires this wave's DMA (vmcnt(0)) for chunk c and, being
// passed by every wave, frees the buffer chunk c-1 was read from.
// Preconditions (checked on the host): fp32, contiguous zone channels, whole-row windows, cin % 4 == 0,
// 16-byte aligned rows blocks.
// ---------------------------------------------------------------------------------------
template <int GT, int NT>
__global__ __launch_bounds__(256) void conv5_fwd_glds_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int z = blockIdx.y;
  const ZoneDesc zd = a.zones[z];
  const int cin = zd.cin;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t item0 = (int64_t)blockIdx.x * a.IPW;
  const int n_items = (int)((a.items - item0) < a.IPW ? (a.items - item0) : a.IPW);
  const int n_ct = n_items * a.TT;
  const int in_len = (a.IPW * kCK * a.RS + 3) & ~3;
  constexpr int w_len = (kCK / 4) * kTaps * GT * 64;
  const int buf_len = in_len + w_len + 32;                  // 32 floats of slack: junk columns read past the last row
  const float* wbase = a.wfrag + zd.eff_off;
  const int n_chunks = (cin + kCK - 1) / kCK;
  const int q = lane >> 4, jl = lane & 15;
  const int chan0 = a.chan_idx[zd.idx_off];

  auto stage = [&](int ch, int s) {
    float* in_tile = smem + 4 + s * buf_len;
    float* w_tile = in_tile + in_len;
    const int c_lo = ch * kCK;
    const int ckc = (cin - c_lo) < kCK ? (cin - c_lo) : kCK;
    const int cnt4 = (ckc * a.Tin) >> 2;
    for (int ii = wave; ii < n_items; ii += 4) {
      const int64_t soff = ((item0 + ii) * a.Ctot + chan0 + c_lo) * (int64_t)a.Tx;
      glds_copy16((const float*)a.in + soff, in_tile + ii * kCK * a.RS, cnt4, lane);
    }
    const int wlen4 = (ckc / 4) * kTaps * GT * 16;
    const int per = (((wlen4 + 3) / 4 + 63) / 64) * 64;     // the weight chunk is split over the 4 waves in 1 KiB pieces
    const int w0 = per * wave;
    const int w1 = w0 + per < wlen4 ? w0 + per : wlen4;
    if (w0 < wlen4)
      glds_copy16(wbase + (int64_t)ch * (kCK / 4) * kTaps * GT * 64 + (int64_t)w0 * 4, w_tile + w0 * 4, w1 - w0, lane);
  };

  // Every wave always carries NT column tiles (no branches around the MFMAs); tiles past the end compute on
  // whatever the LDS holds and are not stored.  The first layer is a valid convolution (pad 0): columns
  // beyond Tout are junk by construction and never stored, so the B operand needs no mask.
  f32x4 acc[NT][GT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int g = 0; g < GT; ++g) acc[j][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // Column tiles.  Rows of more than one tile (TT > 1): tile ct is 16 output steps of item ct / TT.  Rows of ONE tile
  // (spec-S features: 13 output steps): the workgroup's columns are its items' output steps back to back -- column
  // 16 ct + jl is step col % Tout of item col / Tout -- so a tile is full instead of 13/16 full (9 items in 8 tiles
  // instead of 8); every lane keeps its own (item, step).
  int t_ii[NT], t_t[NT], boff[NT];
  bool t_ok[NT];
  const bool packed = a.TT == 1;
  const int n_cols = n_items * a.Tout;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int ct = j * 4 + wave;
    if (packed) {
      const int col = ct * 16 + jl;
      t_ok[j] = col < n_cols;
      const int cc = t_ok[j] ? col : 0;                     // a column past the end computes on column 0, not stored
      t_ii[j] = cc / a.Tout;
      t_t[j] = cc - t_ii[j] * a.Tout;
    } else {
      const bool tile = ct < n_ct;
      const int ctc = tile ? ct : 0;
      t_ii[j] = ctc / a.TT;
      t_t[j] = (ctc - t_ii[j] * a.TT) * 16 + jl;
      t_ok[j] = tile && t_t[j] < a.Tout;
    }
    boff[j] = t_ii[j] * kCK * a.RS + t_t[j] + q * a.RS;
  }
  struct Frag {
    float af[kTaps][GT];
    float bf[NT][kTaps];
  };
  stage(0, 0);
  for (int ch = 0; ch < n_chunks; ++ch) {
    __syncthreads();                                        // drains this wave's DMA of chunk ch; all waves left chunk ch-1
    if (ch + 1 < n_chunks) stage(ch + 1, (ch + 1) & 1);
    const float* in_tile = smem + 4 + (ch & 1) * buf_len;
    const float* w_tile = in_tile + in_len + lane;
    const int c_lo = ch * kCK;
    const int ncg = ((cin - c_lo) < kCK ? (cin - c_lo) : kCK) / 4;
    auto load = [&](int cg, Frag& f) {
      const float* rowp = in_tile + cg * 4 * a.RS;
#pragma unroll
      for (int k = 0; k < kTaps; ++k)
#pragma unroll
        for (int g = 0; g < GT; ++g) f.af[k][g] = w_tile[((cg * kTaps + k) * GT + g) * 64];
#pragma unroll
      for (int jj = 0; jj < NT; ++jj)
#pragma unroll
        for (int k = 0; k < kTaps; ++k) f.bf[jj][k] = rowp[boff[jj] + k];
    };
    auto mma = [&](const Frag& f) {
#pragma unroll
      for (int k = 0; k < kTaps; ++k)
#pragma unroll
        for (int jj = 0; jj < NT; ++jj)
#pragma unroll
          for (int g = 0; g < GT; ++g)
            acc[jj][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.af[k][g], f.bf[jj][k], acc[jj][g], 0, 0, 0);
    };
    Frag f0, f1;
    load(0, f0);
    for (int cg = 0; cg < ncg; cg += 2) {
      if (cg + 1 < ncg) load(cg + 1, f1);
      mma(f0);
      if (cg + 1 < ncg) {
        if (cg + 2 < ncg) load(cg + 2, f0);
        mma(f1);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    if (!t_ok[j]) continue;
    const int64_t item = item0 + t_ii[j];
    const int t = t_t[j];
#pragma unroll
    for (int gt = 0; gt < GT; ++gt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int g = gt * 16 + 4 * q + r;
        float v = acc[j][gt][r];
        if (a.bias) v += a.bias[z * a.F + g];
        ((float*)a.out)[((item * a.Z + z) * a.F + g) * (int64_t)a.Tout + t] = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// BASELINE config 3 (bf16-mixed, scripts/train_fast.py:277): first-layer forward on the bf16 matrix cores.
//   v_mfma_f32_16x16x32_bf16, M = 16 filters, N = 16 time steps, K = 32 = 4 input channels x an 8-sample window.
// The K index of a channel runs over time: lane (t = l & 15, q = l >> 4) supplies the B operand
//   x[c = 4 cg + q][t + j],  j = 0..7
// and the A operand holds the 5 taps of that channel followed by three zeros (prep_fused_kernel, `wfrag16`), so one
// MFMA applies all five taps of four channels (5/8 of its K is used; at 16x the fp32 rate the matrix pipe is idle
// either way -- the kernel is bound by streaming x).  The window of a lane is built from ONE LDS dword per lane and
// DPP row shifts (lane t holds x[c][t]; x[c][t + j] is the value of lane t + j of its 16-lane row): no transposed
// staging, no per-lane gather.  Inputs stay fp32 in LDS (LDS-DMA, double-buffered as in conv5_fwd_glds_kernel) and are
// rounded to bf16 (RNE) as the fragment is packed; accumulation is fp32; the output is written as bf16 in
// [item][t][filter] order (8-byte stores of the lane's four consecutive filters), which is the layout the bf16 tail
// and the bf16 weight-gradient kernel read.
// Preconditions (host): one column tile (Tout <= 13 so that only sample 16 lies beyond the DPP row), Tin <= 17,
// contiguous zone channels, whole-row windows, cin % 4 == 0, 16-byte aligned row blocks.
// ---------------------------------------------------------------------------------------
// IN16 (round 3): the input itself is bf16 (the extractor's isd_features_fused_bf16 map: the same RNE rounding, done
// once by the producer) -- half the bytes to stream, the window is built from 16-bit LDS reads and packed with one
// v_lshl_or_b32 per pair instead of a rounding sequence.  Rows are then 2 Tin bytes: the chunk copies stay whole
// 16-byte pieces when the zone has a multiple of 8 channels (host check).
template <int GT, int NT, bool IN16 = false>
__global__ __launch_bounds__(256) void conv5_fwd_bf16_kernel(ConvArgs a, const uint4* __restrict__ wfrag16) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int z = blockIdx.y;
  const ZoneDesc zd = a.zones[z];
  const int cin = zd.cin;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t item0 = (int64_t)blockIdx.x * a.IPW;
  const int n_items = (int)((a.items - item0) < a.IPW ? (a.items - item0) : a.IPW);
  const int in_len = (a.IPW * kCK * a.RS + 3) & ~3;
  constexpr int w_len = (kCK / 4) * GT * 64 * 4;            // floats: 16 bytes per lane and fragment
  const int buf_len = in_len + w_len + 32;                  // 32 floats of slack: dead lanes read past the last row
  const uint4* wbase = wfrag16 + zd.eff_off / kTaps;
  const int n_chunks = (cin + kCK - 1) / kCK;
  const int q = lane >> 4, jl = lane & 15;
  const int chan0 = a.chan_idx[zd.idx_off];

  auto stage = [&](int ch, int s) {
    float* in_tile = smem + 4 + s * buf_len;
    float* w_tile = in_tile + in_len;
    const int c_lo = ch * kCK;
    const int ckc = (cin - c_lo) < kCK ? (cin - c_lo) : kCK;
    const int cnt4 = IN16 ? (ckc * a.Tin) >> 3 : (ckc * a.Tin) >> 2;
    for (int ii = wave; ii < n_items; ii += 4) {
      const int64_t soff = ((item0 + ii) * a.Ctot + chan0 + c_lo) * (int64_t)a.Tx;
      if (IN16)
        glds_copy16(reinterpret_cast<const float*>((const unsigned short*)a.in + soff), in_tile + ((ii * kCK * a.RS) >> 1),
                    cnt4, lane);
      else
        glds_copy16((const float*)a.in + soff, in_tile + ii * kCK * a.RS, cnt4, lane);
    }
    const int wlen4 = (ckc / 4) * GT * 64;                  // 16-byte pieces of the weight chunk
    glds_copy16_strided(reinterpret_cast<const float*>(wbase + (int64_t)ch * (kCK / 4) * GT * 64), w_tile, wlen4,
                        wave * 64, 256, lane);
  };

  f32x4 acc[NT][GT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int g = 0; g < GT; ++g) acc[j][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int t_ii[NT], boff[NT];
  bool t_ok[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int ct = j * 4 + wave;                            // one column tile per item (TT == 1)
    t_ok[j] = ct < n_items;
    t_ii[j] = t_ok[j] ? ct : 0;
    boff[j] = t_ii[j] * kCK * a.RS + q * a.RS;
  }
  const bool has16 = a.Tin > 16;
  const int j0 = jl < a.Tin ? jl : 0;                       // lane's own sample (rows shorter than 16: clamp, then zero)
  const bool own_ok = jl < a.Tin;
  stage(0, 0);
  for (int ch = 0; ch < n_chunks; ++ch) {
    __syncthreads();                                        // drains this wave's DMA of chunk ch; all waves left chunk ch-1
    if (ch + 1 < n_chunks) stage(ch + 1, (ch + 1) & 1);
    const float* in_tile = smem + 4 + (ch & 1) * buf_len;
    const uint4* w_tile = reinterpret_cast<const uint4*>(in_tile + in_len) + lane;
    const int c_lo = ch * kCK;
    const int ncg = ((cin - c_lo) < kCK ? (cin - c_lo) : kCK) / 4;
    // the LDS reads of channel group cg + 1 are issued before group cg is packed and multiplied (the chain LDS read ->
    // DPP shifts -> pack -> MFMA of one group is all latency): 51.5 -> 49.1 us at cfg 3.  (Measured and not kept: a
    // third chunk buffer with the DMA two chunks ahead, 52.4 us; sixteen items per workgroup, 78 us.)
    struct Raw {
      uint4 aw[GT];
      unsigned u0[NT], u16[NT];
      float e0[NT], s16[NT];
    };
    auto load = [&](int cg, Raw& r) {
#pragma unroll
      for (int g = 0; g < GT; ++g) r.aw[g] = w_tile[(cg * GT + g) * 64];
      const float* rowp = in_tile + cg * 4 * a.RS;
#pragma unroll
      for (int jj = 0; jj < NT; ++jj) {
        if constexpr (IN16) {
          const unsigned short* rr = reinterpret_cast<const unsigned short*>(in_tile) + cg * 4 * a.RS + boff[jj];
          r.u0[jj] = rr[j0];
          r.u16[jj] = has16 ? (unsigned)rr[16] : 0u;
        } else {
          const float* rr = rowp + boff[jj];
          r.e0[jj] = rr[j0];
          r.s16[jj] = has16 ? rr[16] : 0.f;              // the one sample beyond the DPP row that a stored output needs
        }
      }
    };
    auto compute = [&](const Raw& r) {
      bf16x8 af[GT];
#pragma unroll
      for (int g = 0; g < GT; ++g) af[g] = __builtin_bit_cast(bf16x8, r.aw[g]);
#pragma unroll
      for (int jj = 0; jj < NT; ++jj) {
        uint4 bw;
        if constexpr (IN16) {
          const unsigned u0 = own_ok ? r.u0[jj] : 0u;
          const unsigned u16 = jl == 12 ? r.u16[jj] : 0u;
          const float f0 = __uint_as_float(u0);             // bit patterns ride the DPP row shifts unchanged
          const unsigned u1 = __float_as_uint(row_shl<1>(f0)), u2 = __float_as_uint(row_shl<2>(f0)),
                         u3 = __float_as_uint(row_shl<3>(f0)), u4 = __float_as_uint(row_shl<4>(f0)) | u16;
          bw.x = u0 | (u1 << 16);
          bw.y = u2 | (u3 << 16);
          bw.z = u4;
          bw.w = 0u;
        } else {
          const float e0 = own_ok ? r.e0[jj] : 0.f;
          const float s16 = jl == 12 ? r.s16[jj] : 0.f;
          const float e1 = row_shl<1>(e0), e2 = row_shl<2>(e0), e3 = row_shl<3>(e0), e4 = row_shl<4>(e0) + s16;
          bw.x = bf16_pack(e0, e1);
          bw.y = bf16_pack(e2, e3);
          bw.z = bf16_pack(e4, 0.f);
          bw.w = 0u;
        }
        const bf16x8 bfr = __builtin_bit_cast(bf16x8, bw);
#pragma unroll
        for (int g = 0; g < GT; ++g)
          acc[jj][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[g], bfr, acc[jj][g], 0, 0, 0);
      }
    };
    Raw r0, r1;
    load(0, r0);
    for (int cg = 0; cg < ncg; cg += 2) {
      if (cg + 1 < ncg) load(cg + 1, r1);
      compute(r0);
      if (cg + 1 < ncg) {
        if (cg + 2 < ncg) load(cg + 2, r0);
        compute(r1);
      }
    }
  }
  // epilogue: D[row g = 4 q + r][col t = jl] -> out[item][t][g], four consecutive filters per lane
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    if (!t_ok[j] || jl >= a.Tout) continue;
    const int64_t item = item0 + t_ii[j];
#pragma unroll
    for (int gt = 0; gt < GT; ++gt) {
      const int g = gt * 16 + 4 * q;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[j][gt][r] + (a.bias ? a.bias[z * a.F + g + r] : 0.f);
      uint2 o;
      o.x = bf16_pack(v[0], v[1]);
      o.y = bf16_pack(v[2], v[3]);
      *reinterpret_cast<uint2*>((unsigned short*)a.out + ((item * a.Z + z) * a.Tout + jl) * (int64_t)a.F + g) = o;
    }
  }
}

// ---------------------------------------------------------------------------------------
// Fused forward of one (window, zone) item for the reference-native shape (F = 32, <= 16 channels per zone,
// <= 256 time steps): cnn1.cnn2 -> cnn3 -> cnn4 -> GELU -> mean, activations handed over in LDS.
// Persistent workgroups (one per CU): the weight fragments of the zone (Weff, W3, W4: 200 A-fragments,
// 50 KB) stay in LDS for the whole launch next to the x rows and the two activation tiles.
// A2/A3/A4 are still written to HBM (coalesced float4 from the LDS tile) when `store` is set, because the
// backward kernels consume them; inference skips the stores.
// ---------------------------------------------------------------------------------------
struct FusedFwdArgs {
  const float* x;            // raw trials [B][Ctot][Tx]
  const float* weff;         // frag-ordered Weff per zone (ZoneDesc::eff_off)
  const float* beff;         // [Z][F]
  const float* w3;           // frag-ordered cnn3 / cnn4 weights, [Z][conv_zstride]
  const float* w4;
  float* a2;                 // [items][Z][F][T1] (may be null when !store)
  float* a3;
  float* a4;
  float* feat;               // [items][Z][F]
  const ZoneDesc* zones;
  const int* chan_idx;
  int64_t wz_stride, items;
  int Z, W, T1, TT, store;   // store: 0 inference, 1 keep A2/A3/A4, 2 keep A2/A3 and GELU'(A4) (fused backward)
  int Ctot, Tx, N, S;
};

// One convolution layer of the fused forward for a wave's NJ column tiles: acc[j][gt] += W[gt] (x) in, over ncg
// groups of 4 input channels.  `wl` = frag-ordered weights + lane, `in` = LDS tile + q * RS (row of this lane's
// K index), off/ok = per-tile column offset and per-tap validity.  Fragments of group cg+1 are fetched while
// the MFMAs of group cg run; no branch stands between the MFMAs (dead tiles compute and are not stored).
template <int NJ>
__device__ __forceinline__ void fused_conv_mma(const float* __restrict__ wl, const float* __restrict__ in, int RS,
                                               int ncg, const int (&off)[NJ], const bool (&ok)[NJ][kTaps],
                                               f32x4 (&acc)[NJ][2]) {
  struct Frag {
    float af[kTaps][2];
    float bf[NJ][kTaps];
  };
  auto load = [&](int cg, Frag& f) {
    const float* rowp = in + cg * 4 * RS;
#pragma unroll
    for (int k = 0; k < kTaps; ++k) {
      f.af[k][0] = wl[((cg * kTaps + k) * 2 + 0) * 64];
      f.af[k][1] = wl[((cg * kTaps + k) * 2 + 1) * 64];
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int k = 0; k < kTaps; ++k) {
        const float v = rowp[off[j] + k];                    // may touch a neighbouring row: masked
        f.bf[j][k] = ok[j][k] ? v : 0.f;
      }
  };
  auto mma = [&](const Frag& f) {
#pragma unroll
    for (int k = 0; k < kTaps; ++k)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        acc[j][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.af[k][0], f.bf[j][k], acc[j][0], 0, 0, 0);
        acc[j][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.af[k][1], f.bf[j][k], acc[j][1], 0, 0, 0);
      }
  };
  Frag f0, f1;
  load(0, f0);
  for (int cg = 0; cg < ncg; cg += 2) {
    if (cg + 1 < ncg) load(cg + 1, f1);
    mma(f0);
    if (cg + 1 < ncg) {
      if (cg + 2 < ncg) load(cg + 2, f0);
      mma(f1);
    }
  }
}

template <int NW, bool DGELU = false>
__device__ __forceinline__ void fused_layer_store(const f32x4 (&acc)[16 / NW][2], const float* __restrict__ bias,
                                                  float* __restrict__ tile, int T1, int TT, int wave, int q, int jl) {
#pragma unroll
  for (int j = 0; j < 16 / NW; ++j) {
    const int tt = j * NW + wave;
    if (tt >= TT) continue;
    const int t = tt * 16 + jl;
    if (t >= T1) continue;
#pragma unroll
    for (int gt = 0; gt < 2; ++gt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int g = gt * 16 + 4 * q + r;
        const float v = acc[j][gt][r] + (bias ? bias[g] : 0.f);
        tile[g * T1 + t] = DGELU ? gelu_grad_f(v) : v;
      }
  }
}

template <int NW>
__global__ __launch_bounds__(NW * 64) void conv4_fused_fwd_kernel(FusedFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int F = 32;
  const int z = blockIdx.y;
  const ZoneDesc zd = a.zones[z];
  const int cz = zd.cin, ncg = (cz + 3) / 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, jl = lane & 15;
  const int T1 = a.T1, W = a.W, TT = a.TT;
  float* xz = smem + 4;                              // [16][W]  (4 floats of slack before every tile)
  float* t2 = smem + ((16 * W + 3) & ~3);            // [32][T1]
  float* t3 = t2 + ((F * T1 + 3) & ~3);              // [32][T1]
  float* red = t3 + ((F * T1 + 3) & ~3);             // [8][32]
  float* we = red + NW * F;                           // [4][5][2][64]   frag-ordered Weff of the zone
  float* w3s = we + 4 * kTaps * 2 * 64;              // [8][5][2][64]
  float* w4s = w3s + 8 * kTaps * 2 * 64;
  float* bls = w4s + 8 * kTaps * 2 * 64;             // [32] beff (in the 64 floats of slack behind the fragments)
  // ---- weight fragments into LDS once per (persistent) workgroup
  for (int e = threadIdx.x; e < 4 * kTaps * 2 * 64; e += NW * 64) we[e] = (e < ncg * kTaps * 2 * 64) ? a.weff[zd.eff_off + e] : 0.f;
  for (int e = threadIdx.x; e < 8 * kTaps * 2 * 64; e += NW * 64) {
    w3s[e] = a.w3[(int64_t)z * a.wz_stride + e];
    w4s[e] = a.w4[(int64_t)z * a.wz_stride + e];
  }
  for (int e = cz * W + threadIdx.x; e < 16 * W; e += NW * 64) xz[e] = 0.f;   // channel round-up rows stay zero
  if (threadIdx.x < F) bls[threadIdx.x] = a.beff[z * F + threadIdx.x];   // a global load per item would queue behind the x fetch
  constexpr int NJ = 16 / NW;
  int off0[NJ], off2[NJ];                             // tile column offsets for pad 0 / pad 2 reads
  bool ok0[NJ][kTaps], ok2[NJ][kTaps];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int tt = (j * NW + wave) < TT ? (j * NW + wave) : 0;   // dead slots recompute tile 0 (never stored)
    off0[j] = tt * 16 + jl;
    off2[j] = tt * 16 + jl - 2;
#pragma unroll
    for (int kk = 0; kk < kTaps; ++kk) {
      ok0[j][kk] = off0[j] + kk < W;
      ok2[j][kk] = off2[j] + kk >= 0 && off2[j] + kk < T1;
    }
  }

  // The wave's two x rows (channels wave, wave + NW) are fetched into registers ONE ITEM AHEAD (right after the
  // barrier that ends the first layer's reads of xz) and written to xz before the item's last stores are issued;
  // their channel indices are loaded once.  (Fetched at the top of the item -- behind a dependent channel-index
  // load -- the rows cost every item their whole latency: see the bf16 twin, tools/conv_phases.py.)
  static_assert(NW == 8, "16 x rows = two per wave");
  const bool has0 = wave < cz, has1 = wave + NW < cz;
  const int64_t ch0 = a.chan_idx[zd.idx_off + (has0 ? wave : 0)];
  const int64_t ch1 = a.chan_idx[zd.idx_off + (has1 ? wave + NW : 0)];
  const int nitems = (int)a.items, N = a.N;
  constexpr int XC = 5;                                 // 64-step chunks: W <= 16 * 16 + 4
  float xr[2][XC];
  int tcl[XC];
#pragma unroll
  for (int c = 0; c < XC; ++c) {
    tcl[c] = c * 64 + lane < W ? c * 64 + lane : W - 1;   // clamped: no test around a load
    xr[0][c] = xr[1][c] = 0.f;
  }
  auto fetch_x = [&](int item) {
    const int b = item / N, n = item - b * N;
    const float* s0 = a.x + ((int64_t)b * a.Ctot + ch0) * (int64_t)a.Tx + (int64_t)n * a.S;
    const float* s1 = a.x + ((int64_t)b * a.Ctot + ch1) * (int64_t)a.Tx + (int64_t)n * a.S;
#pragma unroll
    for (int c = 0; c < XC - 1; ++c) {
      xr[0][c] = s0[tcl[c]];
      xr[1][c] = s1[tcl[c]];
    }
    if (W > (XC - 1) * 64) {
      xr[0][XC - 1] = s0[tcl[XC - 1]];
      xr[1][XC - 1] = s1[tcl[XC - 1]];
    }
  };
  auto write_xz = [&]() {
#pragma unroll
    for (int c = 0; c < XC; ++c) {
      const int t = c * 64 + lane;
      if (t < W) {
        if (has0) xz[wave * W + t] = xr[0][c];
        if (has1) xz[(wave + NW) * W + t] = xr[1][c];
      }
    }
  };
  if ((int)blockIdx.x < nitems) fetch_x(blockIdx.x);
  __syncthreads();
  write_xz();

  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    __syncthreads();                                   // xz complete; previous item's tiles are no longer read
    f32x4 acc[16 / NW][2];
    // ---------------- cnn1 o cnn2 (valid, Cz x 5 taps)
#pragma unroll
    for (int j = 0; j < 16 / NW; ++j)
#pragma unroll
      for (int gt = 0; gt < 2; ++gt) acc[j][gt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    fused_conv_mma<NJ>(we + lane, xz + q * W, W, ncg, off0, ok0, acc);   // rows cz..4*ncg-1 of xz are zero
    fused_layer_store<NW>(acc, bls, t2, T1, TT, wave, q, jl);
    __syncthreads();                                   // t2 complete; every read of xz is done
    if (item + (int)gridDim.x < nitems) fetch_x(item + gridDim.x);
    if (a.store) {
      float4* dst = reinterpret_cast<float4*>(a.a2 + ((int64_t)item * a.Z + z) * (int64_t)(F * T1));
      const float4* src = reinterpret_cast<const float4*>(t2);
      for (int e = threadIdx.x; e < (F * T1) / 4; e += NW * 64) dst[e] = src[e];
    }
    // ---------------- cnn3 (pad 2)
#pragma unroll
    for (int j = 0; j < 16 / NW; ++j)
#pragma unroll
      for (int gt = 0; gt < 2; ++gt) acc[j][gt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    fused_conv_mma<NJ>(w3s + lane, t2 + q * T1, T1, 8, off2, ok2, acc);
    fused_layer_store<NW>(acc, nullptr, t3, T1, TT, wave, q, jl);
    __syncthreads();                                   // t3 complete; every read of t2 is done
    if (a.store) {
      float4* dst = reinterpret_cast<float4*>(a.a3 + ((int64_t)item * a.Z + z) * (int64_t)(F * T1));
      const float4* src = reinterpret_cast<const float4*>(t3);
      for (int e = threadIdx.x; e < (F * T1) / 4; e += NW * 64) dst[e] = src[e];
    }
    // ---------------- cnn4 (pad 2) -> GELU -> mean
#pragma unroll
    for (int j = 0; j < 16 / NW; ++j)
#pragma unroll
      for (int gt = 0; gt < 2; ++gt) acc[j][gt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    fused_conv_mma<NJ>(w4s + lane, t3 + q * T1, T1, 8, off2, ok2, acc);
    // t2 is free: stage A4 for the store -- or GELU'(A4), all the fused backward needs of it (store == 2);
    // GELU (row means) and GELU' share one erf evaluation
    float part[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int j = 0; j < 16 / NW; ++j) {
      const int tt = j * NW + wave, t = tt * 16 + jl;
      const bool live = tt < TT && t < T1;
      const int tc = live ? t : 0;
#pragma unroll
      for (int gt = 0; gt < 2; ++gt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float u = acc[j][gt][r];
          float cdf, ez;
          gelu_parts(u, cdf, ez);
          part[gt][r] += live ? u * cdf : 0.f;
          const float keep = a.store == 2 ? fmaf(u * 0.39894228040143267794f, ez, cdf) : u;
          if (live && a.store) t2[(gt * 16 + 4 * q + r) * T1 + tc] = keep;
        }
    }
#pragma unroll
    for (int gt = 0; gt < 2; ++gt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float sacc = part[gt][r];
        sacc += row_shr<8>(sacc);
        sacc += row_shr<4>(sacc);
        sacc += row_shr<2>(sacc);
        sacc += row_shr<1>(sacc);
        part[gt][r] = sacc;                            // lane 15 of each 16-lane row holds the row sum
      }
    if (jl == 15) {
#pragma unroll
      for (int gt = 0; gt < 2; ++gt)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave * F + gt * 16 + 4 * q + r] = part[gt][r];
    }
    __syncthreads();
    if (item + (int)gridDim.x < nitems) write_xz();       // before this item's last stores: the wait is for the fetch only
    if (threadIdx.x < F)
      a.feat[((int64_t)item * a.Z + z) * F + threadIdx.x] =
          [&] {
            float tot = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) tot += red[w * F + threadIdx.x];
            return tot;
          }() / (float)T1;
    if (a.store) {
      float4* dst = reinterpret_cast<float4*>(a.a4 + ((int64_t)item * a.Z + z) * (int64_t)(F * T1));
      const float4* src = reinterpret_cast<const float4*>(t2);
      for (int e = threadIdx.x; e < (F * T1) / 4; e += NW * 64) dst[e] = src[e];
    }
  }
}

// ---------------------------------------------------------------------------------------
// Fused backward of one (window, zone) item for the reference-native shape (companion of conv4_fused_fwd_kernel):
//   G4 = dfeat/T1 * GELU'(A4);  dW4 += G4 (*) A3;  G3 = W4^T (*) G4;  dW3 += G3 (*) A2;  G2 = W3^T (*) G3;
//   dWeff += G2 (*) x,  dbeff += sum_t G2.
// The gradient tiles G4/G3/G2 live only in LDS (two alternating [32][T1] tiles), A3 / A2 are staged by LDS-DMA
// into a third tile, A4 is consumed straight from global memory, and the transposed cnn3/cnn4 fragments stay in
// LDS for the whole (persistent) launch.  HBM traffic per item: three activation reads + the zone's x rows
// (the layer-wise backward moved 11 tile-sized reads / writes).  Every weight gradient accumulates in registers
// across all items of the workgroup (a wave owns one 16-channel tile x both filter tiles x 5 taps and a share
// of the time steps) and leaves as one partial slab per wave at the end.
// ---------------------------------------------------------------------------------------
struct FusedBwdArgs {
  const float* x;            // raw trials [B][Ctot][Tx]
  const float* dfeat;        // [items][Z][F]
  const float* a2;           // saved activations [items][Z][F][T1]
  const float* a3;
  const float* a4;           // GELU'(A4) (the forward ran with store == 2)
  const float* w3t;          // frag-ordered transposed + flipped cnn3 / cnn4 weights, [Z][conv_zstride]
  const float* w4t;
  float* part4;              // [gridDim.x * 4][slab1]   slab1 = Z*F*F*5, zone block at z*F*F*5, natural [g][c][k]
  float* part3;
  float* part0;              // [gridDim.x * NW][slab0]  dWeff blocks at ZoneDesc::wg_off, [g][cz+1][k]
  const ZoneDesc* zones;
  const int* chan_idx;
  int64_t wz_stride, items, slab1, slab0;
  int Z, W, T1, TT;
  int Ctot, Tx, N, S;
};

// acc[k] += sum over this wave's time steps of G[row][t] * In[col][t + k - pad]   (MFMA: M = the 16 filters of ONE
// filter tile -- G points at its first row --, N = channel, K = 4 time steps).  Steps first, first + stride, ... < nks;
// fragments of the next step are fetched from LDS while the MFMAs of the current one run.  `accb` (BIAS) takes an
// all-ones B operand: sum_t G.
// (Round 2 gave a wave BOTH filter tiles: 30 accumulator tiles = 120 registers across the three layers, 256 VGPRs,
// 132 bytes of scratch and 176 v_readlane / v_writelane.  A wave now owns one filter tile and twice the time steps:
// the same MFMAs, 15 accumulator tiles.)
template <bool BIAS>
__device__ __forceinline__ void fused_wgrad_mma(const float* __restrict__ G, const float* __restrict__ In, int T1,
                                                int RSi, int Tin, int pad, int first, int stride, int nks, int q,
                                                int jl, f32x4 (&acc)[kTaps], f32x4& accb) {
  struct Frag {
    float a;
    float b[kTaps];
    float one;
  };
  const float* gr = G + jl * T1 + q;
  const float* ir = In + jl * RSi + q - pad;
  // interior steps touch only valid samples for every lane and tap: plain reads at immediate offsets; the
  // (at most three) edge steps take the masked path.  The choice is wave-uniform and sits in the load phase.
  const int s_lo = (pad + 3) >> 2;                                   // first step with 4s - pad >= 0
  const int t_hi = (T1 < Tin + pad - kTaps + 1 ? T1 : Tin + pad - kTaps + 1);   // t + 4 - pad < Tin and t < T1
  const int s_hi = t_hi >> 2;                                        // steps s < s_hi have 4s + 3 < t_hi
  auto load = [&](int s, Frag& f) {
    const int t0 = s * 4;
    if (s >= s_lo && s < s_hi) {
      f.a = gr[t0];
#pragma unroll
      for (int k = 0; k < kTaps; ++k) f.b[k] = ir[t0 + k];
      f.one = 1.f;
      return;
    }
    const int t = t0 + q;
    const bool ok = t < T1;
    const float v0 = gr[ok ? t0 : 0];
    f.a = ok ? v0 : 0.f;
#pragma unroll
    for (int k = 0; k < kTaps; ++k) {
      const int idx = t + k - pad;
      const bool in = ok && idx >= 0 && idx < Tin;
      const float v = ir[in ? t0 + k : pad];
      f.b[k] = in ? v : 0.f;
    }
    f.one = ok ? 1.f : 0.f;
  };
  auto mma = [&](const Frag& f) {
#pragma unroll
    for (int k = 0; k < kTaps; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a, f.b[k], acc[k], 0, 0, 0);
    if (BIAS) accb = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a, f.one, accb, 0, 0, 0);
  };
  Frag f0, f1;
  if (first < nks) load(first, f0);
  for (int s = first; s < nks; s += 2 * stride) {
    if (s + stride < nks) load(s + stride, f1);
    mma(f0);
    if (s + stride < nks) {
      if (s + 2 * stride < nks) load(s + 2 * stride, f0);
      mma(f1);
    }
  }
}

template <int NW>
__global__ __launch_bounds__(NW * 64) void conv4_fused_bwd_kernel(FusedBwdArgs a) {
  static_assert(NW == 8, "wave roles below assume 8 waves");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int F = 32;
  const int z = blockIdx.y;
  const ZoneDesc zd = a.zones[z];
  const int cz = zd.cin, cin1 = cz + 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, jl = lane & 15;
  const int T1 = a.T1, W = a.W, TT = a.TT;
  const int tile = (F * T1 + 3) & ~3;
  float* xz = smem + 4;                              // [16][W]
  float* ga = smem + ((4 + 16 * W + 3) & ~3) + 4;    // [32][T1]  (4 floats of slack before each tile: pad-2 reads)
  float* gb = ga + tile + 4;                         // [32][T1]
  float* at = gb + tile + 4;                         // [32][T1]  staged activation (A3, then A2)
  float* w4s = at + tile + 4;                        // [8][5][2][64] transposed + flipped cnn4 fragments
  float* w3s = w4s + 8 * kTaps * 2 * 64;
  float* dfs = w3s + 8 * kTaps * 2 * 64;             // [32] dfeat of the item / T1
  for (int e = threadIdx.x; e < 8 * kTaps * 2 * 64; e += NW * 64) {
    w4s[e] = a.w4t[(int64_t)z * a.wz_stride + e];
    w3s[e] = a.w3t[(int64_t)z * a.wz_stride + e];
  }
  for (int e = cz * W + threadIdx.x; e < 16 * W; e += NW * 64) xz[e] = 0.f;   // channel round-up rows stay zero
  constexpr int NJ = 16 / NW;
  int off2[NJ];
  bool ok2[NJ][kTaps];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int tt = (j * NW + wave) < TT ? (j * NW + wave) : 0;
    off2[j] = tt * 16 + jl - 2;
#pragma unroll
    for (int kk = 0; kk < kTaps; ++kk) ok2[j][kk] = off2[j] + kk >= 0 && off2[j] + kk < T1;
  }
  // weight-gradient roles: cnn3 / cnn4: channel tile = wave & 1, filter tile = (wave >> 1) & 1, time share = wave >> 2
  // (of 2); Weff: filter tile = wave & 1, time share = wave >> 1 (of 4)
  const int ct = wave & 1, gw = (wave >> 1) & 1, ks = wave >> 2;
  const int g0w = wave & 1, ks0 = wave >> 1;
  const int nks = (T1 + 3) >> 2;
  f32x4 acc4[kTaps], acc3[kTaps], acc0[kTaps], accb = {0.f, 0.f, 0.f, 0.f}, nob = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < kTaps; ++k) {
    acc4[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc3[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc0[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const int n4 = (F * T1) >> 2;
  static_assert(NW == 8, "16 x rows = two per wave");
  const int64_t ch0 = a.chan_idx[zd.idx_off + (wave < cz ? wave : 0)];
  const int64_t ch1 = a.chan_idx[zd.idx_off + (wave + NW < cz ? wave + NW : 0)];

  for (int64_t item = blockIdx.x; item < a.items; item += gridDim.x) {
    const int64_t b = item / a.N;
    const int n = (int)(item - b * a.N);
    const int64_t abase = (item * a.Z + z) * (int64_t)(F * T1);
    __syncthreads();                                   // the previous item's tiles are no longer read
    glds_copy16_strided(a.a4 + abase, ga, n4, wave * 64, NW * 64, lane);      // GELU'(A4), stored by the forward
    glds_copy16_strided(a.a3 + abase, at, n4, wave * 64, NW * 64, lane);
    for (int r = wave, i = 0; r < cz; r += NW, ++i) {   // channel indices from registers: as a load here the DMA waited for it
      const float* src = a.x + (b * a.Ctot + (i ? ch1 : ch0)) * (int64_t)a.Tx + (int64_t)n * a.S;
      for (int t0 = 0; t0 < W; t0 += 64)
        if (t0 + lane < W)
          __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + t0 + lane), (lds_ptr_t)(xz + r * W + t0), 4, 0, 0);
    }
    if (threadIdx.x < F) dfs[threadIdx.x] = a.dfeat[(item * a.Z + z) * F + threadIdx.x] / (float)T1;
    __syncthreads();                                   // the three copies have landed, dfs is visible
    {
      // G4 = dfeat/T1 * GELU'(A4): scale the rows of the tile in place
      float4* gt4 = reinterpret_cast<float4*>(ga);
      for (int e = threadIdx.x; e < n4; e += NW * 64) {
        float4 v = gt4[e];
        const int g0 = (e * 4) / T1, g3 = (e * 4 + 3) / T1;
        if (g0 == g3) {
          const float d = dfs[g0];
          v.x *= d; v.y *= d; v.z *= d; v.w *= d;
        } else {
          v.x *= dfs[g0]; v.y *= dfs[(e * 4 + 1) / T1]; v.z *= dfs[(e * 4 + 2) / T1]; v.w *= dfs[g3];
        }
        gt4[e] = v;
      }
    }
    __syncthreads();                                   // G4, A3 and the x rows are in LDS
    // ---------------- cnn4: dW4 += G4 (*) A3 ; G3 = W4^T (*) G4 -> gb
    fused_wgrad_mma<false>(ga + gw * 16 * T1, at + ct * 16 * T1, T1, T1, T1, 2, ks, 2, nks, q, jl, acc4, nob);
    {
      f32x4 acc[NJ][2];
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int gt = 0; gt < 2; ++gt) acc[j][gt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      fused_conv_mma<NJ>(w4s + lane, ga + q * T1, T1, 8, off2, ok2, acc);
      fused_layer_store<NW>(acc, nullptr, gb, T1, TT, wave, q, jl);
    }
    __syncthreads();                                   // G3 complete; A3 and G4 are dead
    glds_copy16_strided(a.a2 + abase, at, n4, wave * 64, NW * 64, lane);
    // ---------------- cnn3 data gradient first (does not need A2): G2 = W3^T (*) G3 -> ga
    {
      f32x4 acc[NJ][2];
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int gt = 0; gt < 2; ++gt) acc[j][gt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      fused_conv_mma<NJ>(w3s + lane, gb + q * T1, T1, 8, off2, ok2, acc);
      fused_layer_store<NW>(acc, nullptr, ga, T1, TT, wave, q, jl);
    }
    __syncthreads();                                   // G2 complete, A2 landed
    fused_wgrad_mma<false>(gb + gw * 16 * T1, at + ct * 16 * T1, T1, T1, T1, 2, ks, 2, nks, q, jl, acc3, nob);
    // ---------------- cnn1 o cnn2: dWeff += G2 (*) x (valid convolution), dbeff += sum_t G2
    fused_wgrad_mma<true>(ga + g0w * 16 * T1, xz, T1, W, W, 0, ks0, 4, nks, q, jl, acc0, accb);
  }
  // ---------------- partial slabs
  // (two time shares -> two slabs per workgroup for cnn3 / cnn4, four for Weff: the host sums exactly those)
  {
    float* s4 = a.part4 + ((int64_t)blockIdx.x * 2 + ks) * a.slab1 + (int64_t)z * F * F * kTaps;
    float* s3 = a.part3 + ((int64_t)blockIdx.x * 2 + ks) * a.slab1 + (int64_t)z * F * F * kTaps;
    float* s0 = a.part0 + ((int64_t)blockIdx.x * 4 + ks0) * a.slab0 + zd.wg_off;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int g = gw * 16 + 4 * q + r, c = ct * 16 + jl;
      const int g0 = g0w * 16 + 4 * q + r;
#pragma unroll
      for (int k = 0; k < kTaps; ++k) {
        s4[((int64_t)g * F + c) * kTaps + k] = acc4[k][r];
        s3[((int64_t)g * F + c) * kTaps + k] = acc3[k][r];
        if (jl < cz) s0[((int64_t)g0 * cin1 + jl) * kTaps + k] = acc0[k][r];
      }
      if (jl < kTaps) s0[((int64_t)g0 * cin1 + cz) * kTaps + jl] = jl == 0 ? accb[r] : 0.f;
    }
  }
}

// ---------------------------------------------------------------------------------------
// BASELINE config 3 for the reference-native shape: the fused forward / backward above with bf16 activations on the
// bf16 matrix cores (scripts/train_fast.py:277 trains under bf16-mixed autocast).
// Activation and gradient tiles are bf16 [time][32 channels] (64-byte rows, two zero guard rows in front -- the
// pad-2 convolutions need no masks -- and zero rows behind T1 up to a multiple of 32 steps):
//  * cnn3 / cnn4 and their data gradients: K = 32 = all input channels of one tap, so a layer is 5 taps x 2 filter
//    tiles of v_mfma_f32_16x16x32_bf16 per 16-step column tile; the B fragment (8 consecutive channels of one time
//    step) is ONE 16-byte LDS read, the 64 lanes of a wave reading one contiguous KiB; the A fragments come
//    pre-packed from prep_fused_kernel (k32).  Results (4 consecutive filters of one step per lane) are packed and
//    stored as 8 bytes into the next tile.
//  * weight gradients: K = 32 time steps.  Both operands are columns of [time][channel] tiles, delivered transposed
//    by ds_read_b64_tr_b16; the input side reads a 12-row window once and the five tap windows are register subsets
//    (even pairs as read, odd pairs by v_alignbit).
//  * the first layer: the wave's two raw fp32 x channels are packed to bf16 pairs into a [time][16 channels] tile
//    (32-byte rows), K = 32 = 2 taps x 16 channels: 3 MFMAs per filter tile and column tile (the fp32 fragments of the
//    fp32 kernels took 80 % of the item's matrix-pipe time for an eighth of its flops); its weight gradient is the
//    same transposed-read contraction as the others.
// HBM traffic per item halves (A2, A3, GELU'(A4) as bf16); LDS per workgroup: forward 62 KB (two per CU), backward
// 147 KB (two sets of activation tiles: the next item streams in while this one computes).
// ---------------------------------------------------------------------------------------
#ifdef ISD_CF_TIMING                  // tools/conv_phases.py: shader-clock stamps of wave 0 of one workgroup, its 5th item
__device__ long long cf_times[32];
#define CF_MARK(k) do { if (blockIdx.x == 1 && blockIdx.y == 6 && threadIdx.x == 0 && item == blockIdx.x + 4 * (int64_t)gridDim.x) cf_times[k] = __builtin_readcyclecounter(); } while (0)
#else
#define CF_MARK(k) do { } while (0)
#endif
__host__ __device__ __forceinline__ int fused16_rows(int TT) { return 32 * ((TT + 1) / 2) + 8; }   // tile rows incl. guards

// acc[j][gt] = sum_k W_k[gt] x in(rows tt[j]*16 + . + k): `wfr` = LDS fragments [5][2][64] uint4, `in` = tile bytes
template <int NJ>
__device__ __forceinline__ void fused_conv_bf16(const uint4* __restrict__ wfr, const char* __restrict__ in,
                                                const int (&tt)[NJ], int lane, f32x4 (&acc)[NJ][2]) {
  const int q = lane >> 4, jl = lane & 15;
#pragma unroll
  for (int k = 0; k < kTaps; ++k) {      // tap-outer: two weight fragments live at a time, not ten
    const bf16x8 a0 = __builtin_bit_cast(bf16x8, wfr[(k * 2 + 0) * 64 + lane]);
    const bf16x8 a1 = __builtin_bit_cast(bf16x8, wfr[(k * 2 + 1) * 64 + lane]);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {       // output step t, tap k reads input step t + k - 2 = tile row t + k
      const bf16x8 b = *reinterpret_cast<const bf16x8*>(in + ((tt[j] * 16 + jl + k) * 64 + q * 16));
      acc[j][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b, acc[j][0], 0, 0, 0);
      acc[j][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b, acc[j][1], 0, 0, 0);
    }
  }
}

// accumulator tiles -> bf16 tile rows (4 consecutive filters of one step = 8 bytes per lane and filter tile)
// `bias`: the lane's 8 filter biases in registers ([filter tile][r]) or null
template <int NW, bool DGELU>
__device__ __forceinline__ void fused_store_bf16(const f32x4 (&acc)[16 / NW][2], const float (*bias)[4],
                                                 char* __restrict__ tile, int T1, int TT, int wave, int q, int jl) {
#pragma unroll
  for (int j = 0; j < 16 / NW; ++j) {
    const int tt = j * NW + wave;
    const int t = tt * 16 + jl;
    if (tt >= TT || t >= T1) continue;
#pragma unroll
    for (int gt = 0; gt < 2; ++gt) {
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[j][gt][r] + (bias ? bias[gt][r] : 0.f);
        if (DGELU) v[r] = gelu_grad_f(v[r]);
      }
      *reinterpret_cast<uint2*>(tile + ((t + 2) * 64 + (gt * 16 + 4 * q) * 2)) =
          make_uint2(bf16_pack(v[0], v[1]), bf16_pack(v[2], v[3]));
    }
  }
}

// cnn1 o cnn2 on the bf16 matrix cores: K = 32 = 2 taps x 16 channels.  `xt` = bf16 [time][16 channels] (32-byte rows),
// `wfr` = [3 tap pairs][2 filter tiles][64] fragments (tap 5 and channels >= cz are zero).  The B fragment of lane
// (step jl, k group q) is the 16-byte half (q & 1) of row step + 2 p + (q >> 1).
template <int NJ>
__device__ __forceinline__ void fused_conv1_bf16(const uint4* __restrict__ wfr, const char* __restrict__ xt,
                                                 const int (&tt)[NJ], int lane, f32x4 (&acc)[NJ][2]) {
  const int q = lane >> 4, jl = lane & 15;
  bf16x8 af[3][2];
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int gt = 0; gt < 2; ++gt) af[p][gt] = __builtin_bit_cast(bf16x8, wfr[(p * 2 + gt) * 64 + lane]);
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    bf16x8 bfr[3];
#pragma unroll
    for (int p = 0; p < 3; ++p)
      bfr[p] = *reinterpret_cast<const bf16x8*>(xt + ((tt[j] * 16 + jl + 2 * p + (q >> 1)) * 32 + (q & 1) * 16));
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      acc[j][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[p][0], bfr[p], acc[j][0], 0, 0, 0);
      acc[j][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[p][1], bfr[p], acc[j][1], 0, 0, 0);
    }
  }
}

// Per item: the wave's two x channels (2 wave, 2 wave + 1) are fetched into registers ONE ITEM AHEAD and written,
// packed to bf16 pairs, straight into the [time][16 ch] tile `xt` (it borrows the front of t3, idle between cnn4's
// reads and cnn3's stores) -> cnn1 o cnn2 -> t2 -> cnn3 -> t3 -> cnn4 -> GELU / GELU' (one erf evaluation for both)
// -> row means.  Phase stamps (tools/conv_phases.py) of the version that staged fp32 x rows by LDS-DMA at the top of
// the item and ran the first layer on the fp32 matrix cores: 12 200 of 37 300 cycles in the first layer (80 % of the
// item's matrix-pipe time for an eighth of its flops), 4 800 exposed on the x fetch (3 200 of them a dependent
// channel-index load), 5 600 in two erf passes.
template <int NW>
__global__ __launch_bounds__(NW * 64, 4) void conv4_fused_fwd_bf16_kernel(FusedFwdArgs a) {   // 2 workgroups per CU
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int F = 32;
  const int z = blockIdx.y;
  const ZoneDesc zd = a.zones[z];
  const int cz = zd.cin, ncg = (cz + 3) / 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, jl = lane & 15;
  const int T1 = a.T1, W = a.W, TT = a.TT;
  const int R = fused16_rows(TT), tile_f = R * 16;            // tile = R rows x 64 bytes
  char* t2 = reinterpret_cast<char*>(smem);
  char* t3 = t2 + tile_f * 4;
  char* xt = t3;                                              // [TT * 16 + 8][16] bf16
  float* red = reinterpret_cast<float*>(t3 + tile_f * 4);     // [NW][32]
  float* bls = red + NW * F;                                  // [32] beff: read per item from LDS -- as registers
                                                              // loaded before the loop the compiler waited for ALL
                                                              // vector memory (the x fetch included) at their use
  uint4* w1s = reinterpret_cast<uint4*>(bls + F);             // [3][2][64] bf16 K = 32 fragments of Weff
  uint4* w3s = w1s + 3 * 2 * 64;                              // [5][2][64]
  uint4* w4s = w3s + kTaps * 2 * 64;
  for (int e = threadIdx.x; e < 3 * 2 * 64; e += NW * 64) {
    const int l = e & 63, gt = (e >> 6) & 1, p = e >> 7, tap = 2 * p + (l >> 5);
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int ch = ((l >> 4) & 1) * 8 + i;
      v[i] = (ch < 4 * ncg && tap < kTaps)
                 ? a.weff[zd.eff_off + (((ch >> 2) * kTaps + tap) * 2 + gt) * 64 + (ch & 3) * 16 + (l & 15)] : 0.f;
    }
    w1s[e] = make_uint4(bf16_pack(v[0], v[1]), bf16_pack(v[2], v[3]), bf16_pack(v[4], v[5]), bf16_pack(v[6], v[7]));
  }
  for (int e = threadIdx.x; e < kTaps * 2 * 64; e += NW * 64) {
    w3s[e] = reinterpret_cast<const uint4*>(a.w3 + (int64_t)z * a.wz_stride)[e];
    w4s[e] = reinterpret_cast<const uint4*>(a.w4 + (int64_t)z * a.wz_stride)[e];
  }
  for (int e = threadIdx.x; e < 2 * tile_f; e += NW * 64) reinterpret_cast<float*>(t2)[e] = 0.f;   // guards, rows >= T1
  if (threadIdx.x < F) bls[threadIdx.x] = a.beff[z * F + threadIdx.x];
  constexpr int NJ = 16 / NW;
  int ttj[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) ttj[j] = (j * NW + wave) < TT ? (j * NW + wave) : 0;   // dead slots recompute tile 0
  const int n16 = T1 * 4;                                      // 16-byte pieces of a tile's real rows
  char* a2b = reinterpret_cast<char*>(a.a2);
  char* a3b = reinterpret_cast<char*>(a.a3);
  char* a4b = reinterpret_cast<char*>(a.a4);
  // this wave's x channels: their row indices once (as a load per item the fetch waited for it first)
  static_assert(NW == 8, "16 channels = two per wave");
  const bool has0 = 2 * wave < cz, has1 = 2 * wave + 1 < cz;
  const int64_t ch0 = a.chan_idx[zd.idx_off + (has0 ? 2 * wave : 0)];
  const int64_t ch1 = a.chan_idx[zd.idx_off + (has1 ? 2 * wave + 1 : 0)];
  const int nitems = (int)a.items, N = a.N;
  constexpr int XC = 5;                                        // 64-step chunks: W <= 16 * 16 + 4
  float xr[2][XC];
  int tcl[XC];
#pragma unroll
  for (int c = 0; c < XC; ++c) tcl[c] = c * 64 + lane < W ? c * 64 + lane : W - 1;   // clamped: no test around a load
  auto fetch_x = [&](int item) {
    const int b = item / N, n = item - b * N;
    const float* s0 = a.x + ((int64_t)b * a.Ctot + ch0) * (int64_t)a.Tx + (int64_t)n * a.S;
    const float* s1 = a.x + ((int64_t)b * a.Ctot + ch1) * (int64_t)a.Tx + (int64_t)n * a.S;
#pragma unroll
    for (int c = 0; c < XC - 1; ++c) {
      xr[0][c] = s0[tcl[c]];
      xr[1][c] = s1[tcl[c]];
    }
    if (W > (XC - 1) * 64) {
      xr[0][XC - 1] = s0[tcl[XC - 1]];
      xr[1][XC - 1] = s1[tcl[XC - 1]];
    }
  };
  // x -> xt: bf16 pairs (channels 2 wave, 2 wave + 1) of step t; absent channels and steps >= W: zero
  auto write_xt = [&]() {
#pragma unroll
    for (int c = 0; c < XC; ++c) {
      const int t = c * 64 + lane;
      const bool in = t < W;
      const unsigned pr = bf16_pack(has0 && in ? xr[0][c] : 0.f, has1 && in ? xr[1][c] : 0.f);
      if (t < TT * 16 + 8) *reinterpret_cast<unsigned*>(xt + t * 32 + wave * 4) = pr;
    }
  };
#pragma unroll
  for (int c = 0; c < XC; ++c) xr[0][c] = xr[1][c] = 0.f;
  if ((int)blockIdx.x < nitems) fetch_x(blockIdx.x);
  __syncthreads();
  write_xt();

  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    const int64_t abase = ((int64_t)item * a.Z + z) * (int64_t)T1 * 64;   // bytes: [item][zone][t][32] bf16
    CF_MARK(0);
    CF_MARK(1);
    __syncthreads();                                   // xt complete; the previous item's readers of t2 are done
    CF_MARK(2);
    if (item + (int)gridDim.x < nitems) fetch_x(item + gridDim.x);
    CF_MARK(3);
    f32x4 acc[NJ][2];
    // ---------------- cnn1 o cnn2 (valid, Cz x 5 taps)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int gt = 0; gt < 2; ++gt) acc[j][gt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    fused_conv1_bf16<NJ>(w1s, xt, ttj, lane, acc);
    CF_MARK(4);
    {
      const float4 b0 = *reinterpret_cast<const float4*>(bls + 4 * q), b1 = *reinterpret_cast<const float4*>(bls + 16 + 4 * q);
      const float bias[2][4] = {{b0.x, b0.y, b0.z, b0.w}, {b1.x, b1.y, b1.z, b1.w}};
      fused_store_bf16<NW, false>(acc, bias, t2, T1, TT, wave, q, jl);
    }
    CF_MARK(5);
    __syncthreads();                                   // t2 complete; every read of xt (= front of t3) is done
    CF_MARK(6);
    if (threadIdx.x < 8) reinterpret_cast<uint4*>(t3)[threadIdx.x] = make_uint4(0u, 0u, 0u, 0u);   // t3's guard rows
    if (a.store) {
      uint4* dst = reinterpret_cast<uint4*>(a2b + abase);
      const uint4* src = reinterpret_cast<const uint4*>(t2 + 2 * 64);
      for (int e = threadIdx.x; e < n16; e += NW * 64) dst[e] = src[e];
    }
    // ---------------- cnn3 (pad 2)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int gt = 0; gt < 2; ++gt) acc[j][gt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    CF_MARK(7);
    fused_conv_bf16<NJ>(w3s, t2, ttj, lane, acc);
    CF_MARK(8);
    fused_store_bf16<NW, false>(acc, nullptr, t3, T1, TT, wave, q, jl);
    CF_MARK(9);
    __syncthreads();                                   // t3 complete; every read of t2 is done
    CF_MARK(10);
    if (a.store) {
      uint4* dst = reinterpret_cast<uint4*>(a3b + abase);
      const uint4* src = reinterpret_cast<const uint4*>(t3 + 2 * 64);
      for (int e = threadIdx.x; e < n16; e += NW * 64) dst[e] = src[e];
    }
    // ---------------- cnn4 (pad 2) -> GELU -> mean; GELU'(A4) (or A4) into t2 for the backward
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int gt = 0; gt < 2; ++gt) acc[j][gt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    CF_MARK(11);
    fused_conv_bf16<NJ>(w4s, t3, ttj, lane, acc);
    CF_MARK(12);
    float part[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int tt = j * NW + wave, t = tt * 16 + jl;
      const bool live = tt < TT && t < T1;
#pragma unroll
      for (int gt = 0; gt < 2; ++gt) {
        float keep[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float u = acc[j][gt][r];
          float cdf, ez;
          gelu_parts(u, cdf, ez);                      // one erf for GELU and GELU'
          part[gt][r] += live ? u * cdf : 0.f;
          keep[r] = a.store == 2 ? fmaf(u * 0.39894228040143267794f, ez, cdf) : u;
        }
        if (live && a.store)
          *reinterpret_cast<uint2*>(t2 + ((t + 2) * 64 + (gt * 16 + 4 * q) * 2)) =
              make_uint2(bf16_pack(keep[0], keep[1]), bf16_pack(keep[2], keep[3]));
      }
    }
    CF_MARK(13);
#pragma unroll
    for (int gt = 0; gt < 2; ++gt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float sacc = part[gt][r];
        sacc += row_shr<8>(sacc);
        sacc += row_shr<4>(sacc);
        sacc += row_shr<2>(sacc);
        sacc += row_shr<1>(sacc);
        part[gt][r] = sacc;
      }
    if (jl == 15) {
#pragma unroll
      for (int gt = 0; gt < 2; ++gt)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave * F + gt * 16 + 4 * q + r] = part[gt][r];
    }
    CF_MARK(14);
    __syncthreads();                                   // row sums in `red`; every read of t3 is done
    CF_MARK(15);
    // the next item's x (fetched after this item's first barrier) into xt = the idle front of t3, BEFORE this item's
    // last stores are issued: the wait for the fetch is a wait for everything issued before it, too
    if (item + (int)gridDim.x < nitems) write_xt();
    if (threadIdx.x < F) {
      float tot = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) tot += red[w * F + threadIdx.x];
      a.feat[((int64_t)item * a.Z + z) * F + threadIdx.x] = tot / (float)T1;
    }
    if (a.store) {
      uint4* dst = reinterpret_cast<uint4*>(a4b + abase);
      const uint4* src = reinterpret_cast<const uint4*>(t2 + 2 * 64);
      for (int e = threadIdx.x; e < n16; e += NW * 64) dst[e] = src[e];
    }
    CF_MARK(16);
  }
}

// transposed 4-row x 16-column block read of a [row][32] bf16 tile: the lane of column c receives rows r0 .. r0 + 3 of
// that column, packed (lane 4 r + p of the 16-lane group supplies the address of row r, columns 4 p .. 4 p + 3)
__device__ __forceinline__ unsigned tr_addr(unsigned tile_base, int row0, int col0, int jl) {
  return tile_base + (unsigned)(((row0 + (jl >> 2)) * 32 + col0 + 4 * (jl & 3)) * 2);
}

// acc[gt][k] += sum_t G[t][gt*16 + m] * In[t + k - 2][ct*16 + n] over 32-step blocks s = first, first + stride, ...
__device__ __forceinline__ void fused_wgrad_bf16(unsigned g_base, unsigned in_base, int ct, int first, int stride,
                                                 int nkb, int q, int jl, f32x4 (&acc)[2][kTaps]) {
  for (int s = first; s < nkb; s += stride) {
    const unsigned ga0 = tr_addr(g_base, 2 + 32 * s + 8 * q, 0, jl);
    const unsigned ia0 = tr_addr(in_base, 32 * s + 8 * q, ct * 16, jl);
    uint2 a00, a01, a10, a11, w0, w1, w2;
    asm volatile(
        "ds_read_b64_tr_b16 %0, %7\n\tds_read_b64_tr_b16 %1, %7 offset:256\n\t"
        "ds_read_b64_tr_b16 %2, %7 offset:32\n\tds_read_b64_tr_b16 %3, %7 offset:288\n\t"
        "ds_read_b64_tr_b16 %4, %8\n\tds_read_b64_tr_b16 %5, %8 offset:256\n\tds_read_b64_tr_b16 %6, %8 offset:512\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(a00), "=&v"(a01), "=&v"(a10), "=&v"(a11), "=&v"(w0), "=&v"(w1), "=&v"(w2)
        : "v"(ga0), "v"(ia0)
        : "memory");
    const bf16x8 af0 = __builtin_bit_cast(bf16x8, make_uint4(a00.x, a00.y, a01.x, a01.y));
    const bf16x8 af1 = __builtin_bit_cast(bf16x8, make_uint4(a10.x, a10.y, a11.x, a11.y));
    const unsigned pe[6] = {w0.x, w0.y, w1.x, w1.y, w2.x, w2.y};     // window rows (0,1) (2,3) ... (10,11)
    unsigned po[5];
#pragma unroll
    for (int n = 0; n < 5; ++n) po[n] = __builtin_amdgcn_alignbit(pe[n + 1], pe[n], 16);   // rows (1,2) (3,4) ...
#pragma unroll
    for (int k = 0; k < kTaps; ++k) {
      const int h = k >> 1;
      const uint4 bw = (k & 1) ? make_uint4(po[h], po[h + 1], po[h + 2], po[h + 3])
                               : make_uint4(pe[h], pe[h + 1], pe[h + 2], pe[h + 3]);
      const bf16x8 bfr = __builtin_bit_cast(bf16x8, bw);
      acc[0][k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af0, bfr, acc[0][k], 0, 0, 0);
      acc[1][k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af1, bfr, acc[1][k], 0, 0, 0);
    }
  }
}

// first layer: acc[gt][k] += sum_t G2[t][gt*16 + m] * x[n][t + k], accb[gt] += sum_t G2[t][.]; x = the bf16
// [time][16 channels] tile (32-byte rows, row = step: a valid convolution has no guard rows)
__device__ __forceinline__ void fused_wgrad_x_bf16(unsigned g_base, unsigned xt_base, int first, int stride, int nkb,
                                                   int q, int jl, f32x4 (&acc)[2][kTaps], f32x4 (&accb)[2]) {
  const bf16x8 ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
  for (int s = first; s < nkb; s += stride) {
    const unsigned ga0 = tr_addr(g_base, 2 + 32 * s + 8 * q, 0, jl);
    const unsigned ia0 = xt_base + (unsigned)(((32 * s + 8 * q + (jl >> 2)) * 16 + 4 * (jl & 3)) * 2);
    uint2 a00, a01, a10, a11, w0, w1, w2;
    asm volatile(
        "ds_read_b64_tr_b16 %0, %7\n\tds_read_b64_tr_b16 %1, %7 offset:256\n\t"
        "ds_read_b64_tr_b16 %2, %7 offset:32\n\tds_read_b64_tr_b16 %3, %7 offset:288\n\t"
        "ds_read_b64_tr_b16 %4, %8\n\tds_read_b64_tr_b16 %5, %8 offset:128\n\tds_read_b64_tr_b16 %6, %8 offset:256\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(a00), "=&v"(a01), "=&v"(a10), "=&v"(a11), "=&v"(w0), "=&v"(w1), "=&v"(w2)
        : "v"(ga0), "v"(ia0)
        : "memory");
    const bf16x8 af0 = __builtin_bit_cast(bf16x8, make_uint4(a00.x, a00.y, a01.x, a01.y));
    const bf16x8 af1 = __builtin_bit_cast(bf16x8, make_uint4(a10.x, a10.y, a11.x, a11.y));
    const unsigned pe[6] = {w0.x, w0.y, w1.x, w1.y, w2.x, w2.y};     // window steps (0,1) (2,3) ... (10,11)
    unsigned po[5];
#pragma unroll
    for (int n = 0; n < 5; ++n) po[n] = __builtin_amdgcn_alignbit(pe[n + 1], pe[n], 16);   // steps (1,2) (3,4) ...
#pragma unroll
    for (int k = 0; k < kTaps; ++k) {
      const int h = k >> 1;
      const uint4 bw = (k & 1) ? make_uint4(po[h], po[h + 1], po[h + 2], po[h + 3])
                               : make_uint4(pe[h], pe[h + 1], pe[h + 2], pe[h + 3]);
      const bf16x8 bfr = __builtin_bit_cast(bf16x8, bw);
      acc[0][k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af0, bfr, acc[0][k], 0, 0, 0);
      acc[1][k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af1, bfr, acc[1][k], 0, 0, 0);
    }
    accb[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af0, ones, accb[0], 0, 0, 0);
    accb[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af1, ones, accb[1], 0, 0, 0);
  }
}

// 16 bytes per lane, global -> LDS (base + 16 lane), issued behind the compiler's back: as the builtin, every later
// ds_write of the kernel would first wait for it (the LDS-DMA hazard tracking cannot tell the tiles apart), and a
// fetch issued one item ahead would be waited for a few instructions later.  The consumer waits with
// __builtin_amdgcn_s_waitcnt (vmcnt 0) + a workgroup barrier.  `lds_base` must be wave-uniform.
__device__ __forceinline__ void dma16_async(const void* src_lane, unsigned lds_base) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
               :: "s"(__builtin_amdgcn_readfirstlane(lds_base)), "v"(src_lane) : "memory", "m0");   // (hipcc warns that m0 is reserved; the
}                                                                                   // clobber still records that it changes)

// One item ahead: GELU'(A4), A3 and A2 of the next item stream into the other tile set by LDS-DMA and its x channels
// and dfeat row into registers while this item computes (one workgroup per CU: the 120 weight-gradient accumulators
// leave two waves per SIMD, and 147 of the 160 KB of LDS pay for the second set).  Fetched at the top of the item the
// three tiles, the x rows behind a dependent channel-index load and dfeat cost the item their whole latency.
template <int NW>
__global__ __launch_bounds__(NW * 64) void conv4_fused_bwd_bf16_kernel(FusedBwdArgs a) {
  static_assert(NW == 8, "wave roles below assume 8 waves");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int F = 32;
  const int z = blockIdx.y;
  const ZoneDesc zd = a.zones[z];
  const int cz = zd.cin, cin1 = cz + 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, jl = lane & 15;
  const int T1 = a.T1, W = a.W, TT = a.TT;
  const int R = fused16_rows(TT), tile_f = R * 16, tile_b = tile_f * 4;
  // LDS: set 0 {tA, tB, tC}, set 1 {tA, tB, tC}, gb, xt, w4s, w3s.   tA: GELU'(A4) -> G4 -> G2;  tB: A3;  tC: A2
  char* sets = reinterpret_cast<char*>(smem);
  char* gb = sets + 6 * tile_b;
  const int xt_rows = 32 * ((TT + 1) / 2) + 16;               // the last 32-step block's 12-step window stays inside
  char* xt = gb + tile_b;                                     // [xt_rows][16] bf16 (rows past W: zeros)
  uint4* w4s = reinterpret_cast<uint4*>(xt + xt_rows * 32);   // transposed + flipped cnn4 / cnn3 fragments
  uint4* w3s = w4s + kTaps * 2 * 64;
  for (int e = threadIdx.x; e < kTaps * 2 * 64; e += NW * 64) {
    w4s[e] = reinterpret_cast<const uint4*>(a.w4t + (int64_t)z * a.wz_stride)[e];
    w3s[e] = reinterpret_cast<const uint4*>(a.w3t + (int64_t)z * a.wz_stride)[e];
  }
  for (int e = threadIdx.x; e < 7 * tile_f + xt_rows * 8; e += NW * 64) smem[e] = 0.f;   // guards, rows >= T1
  constexpr int NJ = 16 / NW;
  int ttj[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) ttj[j] = (j * NW + wave) < TT ? (j * NW + wave) : 0;
  const int ct = wave & 1, ks = wave >> 1;
  const int nkb = (T1 + 31) >> 5;                             // 32-step blocks of the weight gradients
  const unsigned smem_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)smem;
  const unsigned gb_base = smem_base + 6 * tile_b;
  const unsigned xt_base = gb_base + tile_b;
  f32x4 acc4[2][kTaps], acc3[2][kTaps], acc0[2][kTaps], accb[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    accb[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < kTaps; ++k) {
      acc4[g][k] = (f32x4){0.f, 0.f, 0.f, 0.f};
      acc3[g][k] = (f32x4){0.f, 0.f, 0.f, 0.f};
      acc0[g][k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  }
  const int n16 = T1 * 4;                                      // 16-byte pieces of a tile's real rows
  const char* a2b = reinterpret_cast<const char*>(a.a2);
  const char* a3b = reinterpret_cast<const char*>(a.a3);
  const char* a4b = reinterpret_cast<const char*>(a.a4);
  const bool has0 = 2 * wave < cz, has1 = 2 * wave + 1 < cz;   // this wave's x channels: 2 wave, 2 wave + 1
  const int64_t ch0 = a.chan_idx[zd.idx_off + (has0 ? 2 * wave : 0)];
  const int64_t ch1 = a.chan_idx[zd.idx_off + (has1 ? 2 * wave + 1 : 0)];
  const int nitems = (int)a.items, N = a.N;
  const float inv_t1 = 1.f / (float)T1;
  constexpr int XC = 5;                                        // 64-step chunks: W <= 16 * 16 + 4
  float xr[2][XC];
  float4 dr[2];                                                // dfeat[8 (tid & 3) .. + 8]: the filters of this thread's pieces
  int tcl[XC];
#pragma unroll
  for (int c = 0; c < XC; ++c) {
    tcl[c] = c * 64 + lane < W ? c * 64 + lane : W - 1;        // clamped: no test around a load
    xr[0][c] = xr[1][c] = 0.f;
  }
  auto prefetch = [&](int item, int set) {
    const int64_t abase = ((int64_t)item * a.Z + z) * (int64_t)T1 * 64;
    const unsigned tA = smem_base + (unsigned)(set * 3 * tile_b) + 128;          // past the two guard rows
#pragma unroll 1
    for (int e0 = wave * 64; e0 < n16; e0 += NW * 64)
      if (e0 + lane < n16) {
        const int64_t o = abase + (int64_t)(e0 + lane) * 16;
        dma16_async(a4b + o, tA + e0 * 16);
        dma16_async(a3b + o, tA + tile_b + e0 * 16);
        dma16_async(a2b + o, tA + 2 * tile_b + e0 * 16);
      }
    const int b = item / N, n = item - b * N;
    const float* s0 = a.x + ((int64_t)b * a.Ctot + ch0) * (int64_t)a.Tx + (int64_t)n * a.S;
    const float* s1 = a.x + ((int64_t)b * a.Ctot + ch1) * (int64_t)a.Tx + (int64_t)n * a.S;
#pragma unroll
    for (int c = 0; c < XC - 1; ++c) {
      xr[0][c] = s0[tcl[c]];
      xr[1][c] = s1[tcl[c]];
    }
    if (W > (XC - 1) * 64) {
      xr[0][XC - 1] = s0[tcl[XC - 1]];
      xr[1][XC - 1] = s1[tcl[XC - 1]];
    }
    const float4* dp = reinterpret_cast<const float4*>(a.dfeat + ((int64_t)item * a.Z + z) * F + (threadIdx.x & 3) * 8);
    dr[0] = dp[0];
    dr[1] = dp[1];
  };
  dr[0] = dr[1] = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();                                             // the zero fill is done before the first DMA lands
  if ((int)blockIdx.x < nitems) prefetch(blockIdx.x, 0);

  int set = 0;
  for (int item = blockIdx.x; item < nitems; item += gridDim.x, set ^= 1) {
    char* tA = sets + set * 3 * tile_b;
    char* tB = tA + tile_b;
    const unsigned tA_base = smem_base + (unsigned)(set * 3 * tile_b), tC_base = tA_base + 2 * tile_b;
    CF_MARK(17);
    __builtin_amdgcn_s_waitcnt(0x0F70);                // vmcnt(0): this wave's share of the item's fetch has landed
    __syncthreads();                                   // ... everyone's; the previous item's tiles are no longer read
    CF_MARK(18);
    // ---------------- x -> xt: bf16 pairs (channels 2 wave, 2 wave + 1) of step t; absent channels, steps >= W: zero
#pragma unroll
    for (int c = 0; c < XC; ++c) {
      const int t = c * 64 + lane;
      const bool in = t < W;
      const unsigned pr = bf16_pack(has0 && in ? xr[0][c] : 0.f, has1 && in ? xr[1][c] : 0.f);
      if (t < TT * 16 + 8) *reinterpret_cast<unsigned*>(xt + t * 32 + wave * 4) = pr;
    }
    CF_MARK(19);
    // ---------------- G4 = dfeat/T1 * GELU'(A4), in place: 16 bytes = 8 consecutive filters of one step
    {
      const float d[8] = {dr[0].x * inv_t1, dr[0].y * inv_t1, dr[0].z * inv_t1, dr[0].w * inv_t1,
                          dr[1].x * inv_t1, dr[1].y * inv_t1, dr[1].z * inv_t1, dr[1].w * inv_t1};
      for (int e = threadIdx.x; e < n16; e += NW * 64) {       // e & 3 == threadIdx.x & 3
        uint4* pv = reinterpret_cast<uint4*>(tA + 128) + e;
        const uint4 v = *pv;
        *pv = make_uint4(bf16_pack(bf16_lo(v.x) * d[0], bf16_hi(v.x) * d[1]), bf16_pack(bf16_lo(v.y) * d[2], bf16_hi(v.y) * d[3]),
                         bf16_pack(bf16_lo(v.z) * d[4], bf16_hi(v.z) * d[5]), bf16_pack(bf16_lo(v.w) * d[6], bf16_hi(v.w) * d[7]));
      }
    }
    CF_MARK(20);
    if (item + (int)gridDim.x < nitems) prefetch(item + gridDim.x, set ^ 1);
    CF_MARK(21);
    __syncthreads();                                   // G4 and xt are complete
    CF_MARK(22);
    // ---------------- cnn4: dW4 += G4 (*) A3 ; G3 = W4^T (*) G4 -> gb
    fused_wgrad_bf16(tA_base, tA_base + tile_b, ct, ks, 4, nkb, q, jl, acc4);
    CF_MARK(23);
    {
      f32x4 acc[NJ][2];
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int gt = 0; gt < 2; ++gt) acc[j][gt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      fused_conv_bf16<NJ>(w4s, tA, ttj, lane, acc);
      fused_store_bf16<NW, false>(acc, nullptr, gb, T1, TT, wave, q, jl);
    }
    CF_MARK(24);
    __syncthreads();                                   // G3 complete; G4 is dead
    CF_MARK(25);
    // ---------------- cnn3: G2 = W3^T (*) G3 -> tA
    {
      f32x4 acc[NJ][2];
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int gt = 0; gt < 2; ++gt) acc[j][gt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      fused_conv_bf16<NJ>(w3s, gb, ttj, lane, acc);
      fused_store_bf16<NW, false>(acc, nullptr, tA, T1, TT, wave, q, jl);
    }
    CF_MARK(26);
    __syncthreads();                                   // G2 complete
    CF_MARK(27);
    fused_wgrad_bf16(gb_base, tC_base, ct, ks, 4, nkb, q, jl, acc3);
    CF_MARK(28);
    // ---------------- cnn1 o cnn2: dWeff += G2 (*) x (valid convolution), dbeff += sum_t G2
    fused_wgrad_x_bf16(tA_base, xt_base, wave, NW, nkb, q, jl, acc0, accb);
    CF_MARK(29);
  }
  // ---------------- partial slabs (same layout as the fp32 kernel)
  {
    float* s4 = a.part4 + ((int64_t)blockIdx.x * 4 + ks) * a.slab1 + (int64_t)z * F * F * kTaps;
    float* s3 = a.part3 + ((int64_t)blockIdx.x * 4 + ks) * a.slab1 + (int64_t)z * F * F * kTaps;
    float* s0 = a.part0 + ((int64_t)blockIdx.x * NW + wave) * a.slab0 + zd.wg_off;
#pragma unroll
    for (int gt = 0; gt < 2; ++gt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int g = gt * 16 + 4 * q + r, c = ct * 16 + jl;
#pragma unroll
        for (int k = 0; k < kTaps; ++k) {
          s4[((int64_t)g * F + c) * kTaps + k] = acc4[gt][k][r];
          s3[((int64_t)g * F + c) * kTaps + k] = acc3[gt][k][r];
          if (jl < cz) s0[((int64_t)g * cin1 + jl) * kTaps + k] = acc0[gt][k][r];
        }
        if (jl < kTaps) s0[((int64_t)g * cin1 + cz) * kTaps + jl] = jl == 0 ? accb[gt][r] : 0.f;
      }
  }
}

// ---------------------------------------------------------------------------------------
// Tail of the feature classifier (spec-S features: a handful of time steps, one zone) in ONE launch:
//   A2 -> cnn3 -> cnn4 -> GELU -> mean -> Linear(F, n_cls) -> softmax cross-entropy / argmax
//   and back: dLogits -> dFeat -> G4 -> (dW4, G3) -> (dW3, G2), dW_fc, db_fc, loss.
// The layer-wise path spends 19 launches of 5-20 us on these [B, 32, 13] tensors.  Here one WAVE owns an item
// end to end (T1 <= 16: one 16-column MFMA tile), its tiles live in a wave-private LDS region (no workgroup
// barrier inside the item loop), the four weight fragment sets are shared by the workgroup in LDS, and the
// weight gradients accumulate in registers over the wave's items (40 MFMA accumulators), are combined across
// the 4 waves through LDS in wave order and leave as one slab per workgroup: [dW3 | dW4 | dW_fc | db_fc | loss].
// ---------------------------------------------------------------------------------------
struct TailArgs {
  const void* a2;            // fp32 [items][F][T1], or (BF) bf16 [items][T1][F]
  void* g2;                  // same layout: gradient w.r.t. A2 (training)
  const float* w3;           // frag-ordered cnn3 / cnn4 weights and their transposed + flipped copies (zone 0)
  const float* w4;
  const float* w3t;
  const float* w4t;
  const float* fc_w;         // [n_cls][F]
  const float* fc_b;         // [n_cls]
  const void* labels;        // uint8 or int64, null: inference
  float* logits;             // [items][n_cls]
  int64_t* pred;             // [items]
  float* part;               // [gridDim.x][slab]   slab = 2*F*F*5 + n_cls*(F+1) + 1
  int64_t items;
  int T1, n_cls, label_bytes, train, slab;
  float grad_scale;
};

// NI items side by side (independent MFMA chains hide the LDS and MFMA latencies of a single wave per SIMD):
// acc[i][gt] += W[gt] (x) in_i  for one 16-column tile per item (F = 32 input channels, pad 2); the A fragments
// are shared by the items
template <int NI>
__device__ __forceinline__ void tail_conv(const float* __restrict__ wl, const float* const (&in)[NI], int T1, int q,
                                          int jl, f32x4 (&acc)[NI][2]) {
  struct Frag {
    float af[kTaps][2];
    float bf[NI][kTaps];
  };
  bool ok[kTaps];
  int off[kTaps];
#pragma unroll
  for (int k = 0; k < kTaps; ++k) {
    const int idx = jl + k - 2;
    ok[k] = idx >= 0 && idx < T1;
    off[k] = ok[k] ? idx : 0;
  }
  auto load = [&](int cg, Frag& f) {
#pragma unroll
    for (int k = 0; k < kTaps; ++k) {
      f.af[k][0] = wl[((cg * kTaps + k) * 2 + 0) * 64];
      f.af[k][1] = wl[((cg * kTaps + k) * 2 + 1) * 64];
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const float v = in[i][(cg * 4 + q) * T1 + off[k]];
        f.bf[i][k] = ok[k] ? v : 0.f;
      }
    }
  };
  auto mma = [&](const Frag& f) {
#pragma unroll
    for (int k = 0; k < kTaps; ++k)
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.af[k][0], f.bf[i][k], acc[i][0], 0, 0, 0);
        acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.af[k][1], f.bf[i][k], acc[i][1], 0, 0, 0);
      }
  };
  Frag f0, f1;
  load(0, f0);
#pragma unroll 1
  for (int cg = 0; cg < 8; cg += 2) {
    load(cg + 1, f1);
    mma(f0);
    if (cg + 2 < 8) load(cg + 2, f0);
    mma(f1);
  }
}

// dW[gt][ct][k] += sum_i sum_t G_i[gt*16 + row][t] * In_i[ct*16 + col][t + k - 2]   (T1 <= 16: four K-steps per item)
template <int NI>
__device__ __forceinline__ void tail_wgrad(const float* const (&G)[NI], const float* const (&In)[NI], int T1, int q,
                                           int jl, f32x4 (&acc)[2][2][kTaps]) {
#pragma unroll 1
  for (int s = 0; s < 4; ++s) {
    const int t = s * 4 + q;
    const bool okA = t < T1;
    const int tc = okA ? t : 0;
    float a[NI][2], b[NI][2][kTaps];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
#pragma unroll
      for (int gt = 0; gt < 2; ++gt) {
        const float v = G[i][(gt * 16 + jl) * T1 + tc];
        a[i][gt] = okA ? v : 0.f;
      }
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int k = 0; k < kTaps; ++k) {
          const int idx = t + k - 2;
          const bool in = okA && idx >= 0 && idx < T1;
          const float v = In[i][(ct * 16 + jl) * T1 + (in ? idx : 0)];
          b[i][ct][k] = in ? v : 0.f;
        }
    }
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int gt = 0; gt < 2; ++gt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int k = 0; k < kTaps; ++k)
            acc[gt][ct][k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][gt], b[i][ct][k], acc[gt][ct][k], 0, 0, 0);
  }
}

template <bool BF = false>
__device__ __forceinline__ void tail_store_tile(const f32x4 (&acc)[2], float* __restrict__ tile, int T1, int q, int jl) {
  if (jl < T1) {
#pragma unroll
    for (int gt = 0; gt < 2; ++gt)
#pragma unroll
      for (int r = 0; r < 4; ++r) tile[(gt * 16 + 4 * q + r) * T1 + jl] = BF ? bf16_round(acc[gt][r]) : acc[gt][r];
  }
}

constexpr int kTailMaxCls = 16;
constexpr int kTailNI = 1;        // items a wave carries side by side (measured at cfg 2: 1 -> 107 us, 2 -> 116 us: 512 VGPRs + spills)

// Epilogue of the classifier-tail kernels: the waves' weight-gradient accumulators (cnn3 / cnn4: 40 f32x4 per lane),
// FC gradients and loss shares are summed across the workgroup and leave as ONE slab in natural order
// [dW3 [g][c][k] | dW4 | dW_fc [cls][F] | db_fc | loss].
// Until round 3 the waves took turns: each added its 160 + 10 values into one LDS image of the slab with scalar
// read-modify-writes at the slab's own (4-way bank-conflicting) addresses, a workgroup barrier per turn -- ~4 us per
// wave; measured by batch scaling (tools/tail_scaling.py) the bf16 tail spent 50 of its 66 us outside the item loop,
// the fp32 tail 36 of 104.  Now every copy of the accumulators lives in LDS in REGISTER layout ([value][lane] f32x4:
// b128 accesses, no conflicts, no index arithmetic), NC copies side by side: in phase p waves p NC .. p NC + NC - 1
// store (p = 0) or add (p > 0) their registers into their copy, and after ceil(NW / NC) phases every thread sums the
// copies element by element -- in copy order: the result does not depend on timing -- while it writes the slab, where
// the one permutation from register layout to natural order happens.
constexpr int kTailCombV = 43;                            // f32x4 per lane: 20 + 20 accumulators, 2 x FC, {bias, loss}
constexpr int kTailCombCopy = kTailCombV * 64 * 4;        // floats per copy
// W(layer, gt, ct, k): the lane's accumulator of (layer 0 = cnn3 / 1 = cnn4, filter tile, channel tile, tap)
template <int NW, int NC, typename WF>
__device__ __forceinline__ void tail_combine_store(float* __restrict__ smem, int wave, int lane, WF&& W,
                                                   const float (&accfc)[kTailMaxCls * 32 / 64], float accb,
                                                   float loss_acc, float* __restrict__ slab, int slab_len, int n_cls) {
  constexpr int F = 32;
  __syncthreads();                                        // the fragment sets and tiles are dead from here on
  f32x4* mine = reinterpret_cast<f32x4*>(smem + (wave % NC) * kTailCombCopy) + lane;
  constexpr int n_phase = (NW + NC - 1) / NC;
#pragma unroll
  for (int ph = 0; ph < n_phase; ++ph) {
    if (wave / NC == ph) {
      auto put = [&](int v, f32x4 x) {
        if (ph) {
          const f32x4 o = mine[v * 64];
          x[0] += o[0]; x[1] += o[1]; x[2] += o[2]; x[3] += o[3];
        }
        mine[v * 64] = x;
      };
#pragma unroll
      for (int layer = 0; layer < 2; ++layer)
#pragma unroll
        for (int gt = 0; gt < 2; ++gt)
#pragma unroll
          for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int k = 0; k < kTaps; ++k) put(((layer * 2 + gt) * 2 + ct) * kTaps + k, W(layer, gt, ct, k));
      put(40, (f32x4){accfc[0], accfc[1], accfc[2], accfc[3]});
      put(41, (f32x4){accfc[4], accfc[5], accfc[6], accfc[7]});
      put(42, (f32x4){accb, loss_acc, 0.f, 0.f});
    }
    __syncthreads();
  }
  constexpr int n_copy = NW < NC ? NW : NC;
  const int n34 = 2 * F * F * kTaps, o_b = n34 + n_cls * F, o_loss = o_b + n_cls;
  for (int e = threadIdx.x; e < slab_len; e += NW * 64) {
    int src;                                              // float index inside a copy: (v * 64 + lane) * 4 + r
    if (e < n34) {
      const int layer = e >= F * F * kTaps, w = e - layer * F * F * kTaps;
      const int g = w / (F * kTaps), rem = w - g * (F * kTaps), c = rem / kTaps, k = rem - c * kTaps;
      const int v = ((layer * 2 + (g >> 4)) * 2 + (c >> 4)) * kTaps + k;
      src = (v * 64 + ((g & 15) >> 2) * 16 + (c & 15)) * 4 + (g & 3);
    } else if (e < o_b) {
      const int f = e - n34, j = f >> 6;                  // flat FC element f = lane + 64 j
      src = ((40 + (j >> 2)) * 64 + (f & 63)) * 4 + (j & 3);
    } else if (e < o_loss) {
      src = (42 * 64 + (e - o_b)) * 4;                    // bias gradient of class e - o_b: lane = class
    } else {
      src = (42 * 64) * 4 + 1;                            // loss: lane 0
    }
    float v = smem[src];
#pragma unroll
    for (int c = 1; c < n_copy; ++c) v += smem[c * kTailCombCopy + src];
    slab[e] = v;
  }
}

// BF (BASELINE config 3): A2 arrives and G2 leaves as bf16 in [item][t][filter] order (the bf16 first-layer kernels),
// and every activation / activation gradient the tail hands from one layer to the next is rounded to bf16 (the
// cnn3 / cnn4 fragment copies already are); the products are those of a bf16 MFMA, the accumulation is fp32.
template <int NW, bool BF>
__global__ __launch_bounds__(NW * 64) void featcnn_tail_kernel(TailArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int F = 32, NI = kTailNI, WF = 8 * kTaps * 2 * 64;              // 5120 floats per fragment set
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, jl = lane & 15;
  const int T1 = a.T1, n_cls = a.n_cls;
  const int tile = F * T1, tpad = (tile + 3) & ~3;
  float* w3s = smem;                                          // fragment sets, shared by the workgroup
  float* w4s = w3s + WF;
  float* w3ts = w4s + WF;
  float* w4ts = w3ts + WF;
  float* fcs = w4ts + WF;                                     // [n_cls][F] then [n_cls]
  float* priv = fcs + kTailMaxCls * (F + 1) + wave * NI * (4 * tpad + 64);
  float *tA2[NI], *tA3[NI], *tG[NI], *tH[NI], *featL[NI], *logL[NI];   // wave-private tiles, per item slot
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    float* base = priv + i * (4 * tpad + 64);
    tA2[i] = base;
    tA3[i] = base + tpad;
    tG[i] = base + 2 * tpad;                                  // G4, later G2
    tH[i] = base + 3 * tpad;                                  // G3
    featL[i] = base + 4 * tpad;                               // [32] pooled features, then [16] logits, [16] dlogits
    logL[i] = featL[i] + 32;
  }
  for (int e = threadIdx.x; e < WF / 4; e += NW * 64) {      // float4: the sets are 16-byte aligned workspace blocks
    reinterpret_cast<float4*>(w3s)[e] = reinterpret_cast<const float4*>(a.w3)[e];
    reinterpret_cast<float4*>(w4s)[e] = reinterpret_cast<const float4*>(a.w4)[e];
    if (a.train) {
      reinterpret_cast<float4*>(w3ts)[e] = reinterpret_cast<const float4*>(a.w3t)[e];
      reinterpret_cast<float4*>(w4ts)[e] = reinterpret_cast<const float4*>(a.w4t)[e];
    }
  }
  for (int e = threadIdx.x; e < n_cls * F; e += NW * 64) fcs[e] = a.fc_w[e];
  for (int e = threadIdx.x; e < n_cls; e += NW * 64) fcs[n_cls * F + e] = a.fc_b[e];
  __syncthreads();

  f32x4 accW4[2][2][kTaps], accW3[2][2][kTaps];
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int k = 0; k < kTaps; ++k) {
        accW4[g][c][k] = (f32x4){0.f, 0.f, 0.f, 0.f};
        accW3[g][c][k] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
  float accfc[kTailMaxCls * F / 64], accb = 0.f, loss_acc = 0.f;   // lane l owns flat fc elements l, l+64, ...
#pragma unroll
  for (int i = 0; i < kTailMaxCls * F / 64; ++i) accfc[i] = 0.f;
  const int n4 = tile >> 2;                                   // <= 128: two float4 per lane
  const float inv_t = 1.f / (float)T1;
  const int64_t stride = (int64_t)gridDim.x * NW * NI;

  // item slot i of this wave walks items first + i, first + i + stride, ...; a slot past the end recomputes the
  // last valid item (uniform control flow for the MFMAs) and contributes nothing
  for (int64_t first = ((int64_t)blockIdx.x * NW + wave) * NI; first < a.items; first += stride) {
    int64_t item[NI];
    bool live[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      live[i] = first + i < a.items;
      item[i] = live[i] ? first + i : first;
      if constexpr (BF) {
        // [t][32] bf16: piece p = 8 consecutive filters of one step -> the fp32 [filter][t] tile
        if (lane < 4 * T1) {
          const uint4 w = reinterpret_cast<const uint4*>((const unsigned short*)a.a2 + item[i] * tile)[lane];
          const int t = lane >> 2, g0 = (lane & 3) * 8;
          float* d = tA2[i] + g0 * T1 + t;
          d[0] = bf16_lo(w.x); d[T1] = bf16_hi(w.x); d[2 * T1] = bf16_lo(w.y); d[3 * T1] = bf16_hi(w.y);
          d[4 * T1] = bf16_lo(w.z); d[5 * T1] = bf16_hi(w.z); d[6 * T1] = bf16_lo(w.w); d[7 * T1] = bf16_hi(w.w);
        }
      } else {
        const float4* src = reinterpret_cast<const float4*>((const float*)a.a2 + item[i] * tile);
        float4* dst = reinterpret_cast<float4*>(tA2[i]);
        if (lane < n4) dst[lane] = src[lane];
        if (lane + 64 < n4) dst[lane + 64] = src[lane + 64];
      }
    }
    wave_lds_sync();
    f32x4 acc[NI][2];
    auto clear = [&]() {
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        acc[i][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc[i][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    };
    clear();
    tail_conv<NI>(w3s + lane, tA2, T1, q, jl, acc);           // A3
#pragma unroll
    for (int i = 0; i < NI; ++i) tail_store_tile<BF>(acc[i], tA3[i], T1, q, jl);
    wave_lds_sync();
    clear();
    tail_conv<NI>(w4s + lane, tA3, T1, q, jl, acc);           // A4 stays in registers
    if constexpr (BF) {
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int gt = 0; gt < 2; ++gt)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][gt][r] = bf16_round(acc[i][gt][r]);
    }
    // GELU + mean over time: row sums inside each 16-lane row
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int gt = 0; gt < 2; ++gt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float sacc = jl < T1 ? gelu_f(acc[i][gt][r]) : 0.f;
          sacc += row_shr<8>(sacc);
          sacc += row_shr<4>(sacc);
          sacc += row_shr<2>(sacc);
          sacc += row_shr<1>(sacc);
          if (jl == 15) featL[i][gt * 16 + 4 * q + r] = sacc * inv_t;
        }
    wave_lds_sync();
#pragma unroll
    for (int i = 0; i < NI; ++i)
      if (lane < n_cls) {
        float l = fcs[n_cls * F + lane];
#pragma unroll
        for (int g = 0; g < F; ++g) l = fmaf(fcs[lane * F + g], featL[i][g], l);
        logL[i][lane] = l;
        if (live[i]) a.logits[item[i] * n_cls + lane] = l;
      }
    wave_lds_sync();
    float lse[NI];
    int yv[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      float mx = -INFINITY;
      int am = 0;
      for (int c = 0; c < n_cls; ++c) {
        const float v = logL[i][c];
        if (v > mx) { mx = v; am = c; }                       // strict '>' keeps the lowest index on ties (torch.argmax)
      }
      if (lane == 0 && live[i]) a.pred[item[i]] = am;
      lse[i] = 0.f;
      yv[i] = 0;
      if (a.labels) {
        float se = 0.f;
        for (int c = 0; c < n_cls; ++c) se += expf(logL[i][c] - mx);
        yv[i] = a.label_bytes == 1 ? (int)((const unsigned char*)a.labels)[item[i]]
                                   : (int)((const long long*)a.labels)[item[i]];
        lse[i] = mx + logf(se);
        if (lane == 0 && live[i]) loss_acc += (lse[i] - logL[i][yv[i]]) * a.grad_scale;
      }
    }
    if (!a.labels || !a.train) continue;
    // dlogits -> LDS (zero for a dead slot); dfeat for this lane's 8 filters; G4 = dfeat/T1 * GELU'(A4)
#pragma unroll
    for (int i = 0; i < NI; ++i)
      if (lane < n_cls)
        logL[i][16 + lane] = live[i] ? (expf(logL[i][lane] - lse[i]) - (lane == yv[i] ? 1.f : 0.f)) * a.grad_scale : 0.f;
    wave_lds_sync();
#pragma unroll
    for (int i = 0; i < NI; ++i) {
#pragma unroll
      for (int gt = 0; gt < 2; ++gt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int g = gt * 16 + 4 * q + r;
          float d = 0.f;
          for (int c = 0; c < n_cls; ++c) d = fmaf(fcs[c * F + g], logL[i][16 + c], d);
          acc[i][gt][r] = d * inv_t * gelu_grad_f(acc[i][gt][r]);
        }
      tail_store_tile<BF>(acc[i], tG[i], T1, q, jl);
      // FC gradients: flat element e = lane + 64 j of [n_cls][F]
#pragma unroll
      for (int j = 0; j < kTailMaxCls * F / 64; ++j) {
        const int e = lane + 64 * j;
        if (e < n_cls * F) accfc[j] = fmaf(logL[i][16 + (e >> 5)], featL[i][e & 31], accfc[j]);
      }
      if (lane < n_cls) accb += logL[i][16 + lane];
    }
    wave_lds_sync();
    tail_wgrad<NI>(tG, tA3, T1, q, jl, accW4);
    clear();
    tail_conv<NI>(w4ts + lane, tG, T1, q, jl, acc);           // G3
#pragma unroll
    for (int i = 0; i < NI; ++i) tail_store_tile<BF>(acc[i], tH[i], T1, q, jl);
    wave_lds_sync();
    tail_wgrad<NI>(tH, tA2, T1, q, jl, accW3);
    clear();
    tail_conv<NI>(w3ts + lane, tH, T1, q, jl, acc);           // G2
    wave_lds_sync();                                          // every read of tG (wgrad4, cnn4 data gradient) is done
#pragma unroll
    for (int i = 0; i < NI; ++i) tail_store_tile(acc[i], tG[i], T1, q, jl);
    wave_lds_sync();
#pragma unroll
    for (int i = 0; i < NI; ++i)
      if (live[i]) {
        if constexpr (BF) {
          if (lane < 4 * T1) {
            const int t = lane >> 2, g0 = (lane & 3) * 8;
            const float* sp = tG[i] + g0 * T1 + t;
            uint4 w;
            w.x = bf16_pack(sp[0], sp[T1]); w.y = bf16_pack(sp[2 * T1], sp[3 * T1]);
            w.z = bf16_pack(sp[4 * T1], sp[5 * T1]); w.w = bf16_pack(sp[6 * T1], sp[7 * T1]);
            reinterpret_cast<uint4*>((unsigned short*)a.g2 + item[i] * tile)[lane] = w;
          }
        } else {
          float4* dst = reinterpret_cast<float4*>((float*)a.g2 + item[i] * tile);
          const float4* src = reinterpret_cast<const float4*>(tG[i]);
          if (lane < n4) dst[lane] = src[lane];
          if (lane + 64 < n4) dst[lane + 64] = src[lane + 64];
        }
      }
    wave_lds_sync();                                          // tiles are reused by the next items
  }
  if (!a.labels) return;
  // the waves' accumulators -> one slab per workgroup (tail_combine_store: two copies, two phases)
  tail_combine_store<NW, 2>(smem, wave, lane,
                            [&](int layer, int gt, int ct, int k) -> f32x4 { return layer ? accW4[gt][ct][k] : accW3[gt][ct][k]; },
                            accfc, accb, loss_acc, a.part + (int64_t)blockIdx.x * a.slab, a.slab, n_cls);
}

// BASELINE config 3, round 3: the same launch on the bf16 matrix cores (v_mfma_f32_16x16x32_bf16).  The fp32-MFMA
// instance above (featcnn_tail_kernel<., true>: bf16 I/O and bf16-rounded intermediates, 480 v_mfma_f32_16x16x4_f32
// per item, 256 VGPRs + 172 AGPRs, one wave per SIMD) was the largest kernel of the bf16 classifier stage.  Here the
// item's tiles are bf16 [time][32 channels] rows (64 bytes; two zero guard rows in front, zero rows behind T1 -- the
// layout of the raw-EEG fused pair, whose fragment helpers it shares): K = 32 is all input channels of one tap, so
// cnn3 / cnn4 and their data gradients are 5 taps x 2 filter tiles = 10 MFMAs each, the weight gradients (K = 32 time
// steps, both operands through ds_read_b64_tr_b16) 20 each: 80 MFMAs per item.  A2 arrives and G2 leaves in exactly
// the tile's row format ([item][t][32] bf16): no conversion on either side.  Eight waves per workgroup (two per SIMD).
template <int NW>
__global__ __launch_bounds__(NW * 64) void featcnn_tail_bf16_kernel(TailArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int F = 32, FR = kTaps * 2 * 64;                   // uint4 per K = 32 fragment set (10 KiB)
  constexpr int R = 40, tile_b = R * 64;                       // fused16_rows(1) rows of 64 bytes
  constexpr int priv_b = 4 * tile_b + 256;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, q = lane >> 4, jl = lane & 15;
  const int T1 = a.T1, n_cls = a.n_cls;
  uint4* w3s = reinterpret_cast<uint4*>(smem);
  uint4* w4s = w3s + FR;
  uint4* w3ts = w4s + FR;
  uint4* w4ts = w3ts + FR;
  float* fcs = reinterpret_cast<float*>(w4ts + FR);            // [n_cls][F] then [n_cls]
  char* priv = reinterpret_cast<char*>(fcs + kTailMaxCls * (F + 1)) + wave * priv_b;
  char* tA2 = priv;
  char* tA3 = priv + tile_b;
  char* tG = priv + 2 * tile_b;                                // G4, later G2
  char* tH = priv + 3 * tile_b;                                // G3
  float* featL = reinterpret_cast<float*>(priv + 4 * tile_b);  // [32] pooled features
  float* logL = featL + 32;                                    // [16] logits, [16] dlogits
  const unsigned smem_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)smem;
  const unsigned priv_base = smem_base + (unsigned)(4 * FR * 16 + kTailMaxCls * (F + 1) * 4 + wave * priv_b);
  for (int e = threadIdx.x; e < FR; e += NW * 64) {
    w3s[e] = reinterpret_cast<const uint4*>(a.w3)[e];
    w4s[e] = reinterpret_cast<const uint4*>(a.w4)[e];
    if (a.train) {
      w3ts[e] = reinterpret_cast<const uint4*>(a.w3t)[e];
      w4ts[e] = reinterpret_cast<const uint4*>(a.w4t)[e];
    }
  }
  for (int e = threadIdx.x; e < n_cls * F; e += NW * 64) fcs[e] = a.fc_w[e];
  for (int e = threadIdx.x; e < n_cls; e += NW * 64) fcs[n_cls * F + e] = a.fc_b[e];
  for (int e = lane; e < priv_b / 16; e += 64) reinterpret_cast<uint4*>(priv)[e] = make_uint4(0u, 0u, 0u, 0u);   // guards, rows >= T1
  __syncthreads();

  f32x4 accW4[2][2][kTaps], accW3[2][2][kTaps];                // [channel tile][filter tile][tap]
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int k = 0; k < kTaps; ++k) {
        accW4[c][g][k] = (f32x4){0.f, 0.f, 0.f, 0.f};
        accW3[c][g][k] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
  float accfc[kTailMaxCls * F / 64], accb = 0.f, loss_acc = 0.f;   // lane l owns flat fc elements l, l+64, ...
#pragma unroll
  for (int i = 0; i < kTailMaxCls * F / 64; ++i) accfc[i] = 0.f;
  const int tile = F * T1;                                     // bf16 elements of an item
  const float inv_t = 1.f / (float)T1;
  const int64_t stride = (int64_t)gridDim.x * NW;
  const int tt0[1] = {0};
  auto store_tile = [&](const f32x4 (&v)[2], char* dst) {      // 4 consecutive filters of step jl = 8 bytes per tile
    if (jl < T1) {
#pragma unroll
      for (int gt = 0; gt < 2; ++gt)
        *reinterpret_cast<uint2*>(dst + ((jl + 2) * 64 + (gt * 16 + 4 * q) * 2)) =
            make_uint2(bf16_pack(v[gt][0], v[gt][1]), bf16_pack(v[gt][2], v[gt][3]));
    }
  };

  // a wave past the end recomputes the last valid item (uniform control flow for the MFMAs) and contributes nothing
  for (int64_t first = (int64_t)blockIdx.x * NW + wave; first - wave < a.items; first += stride) {
    const bool live = first < a.items;
    const int64_t item = live ? first : a.items - 1;
    if (lane < 4 * T1)
      *reinterpret_cast<uint4*>(tA2 + 128 + lane * 16) =
          reinterpret_cast<const uint4*>((const unsigned short*)a.a2 + item * tile)[lane];
    wave_lds_sync();
    f32x4 acc[1][2];
    auto clear = [&]() {
      acc[0][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
      acc[0][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    };
    clear();
    fused_conv_bf16<1>(w3s, tA2, tt0, lane, acc);              // A3
    store_tile(acc[0], tA3);
    wave_lds_sync();
    clear();
    fused_conv_bf16<1>(w4s, tA3, tt0, lane, acc);              // A4 stays in registers (a bf16 tensor under autocast)
#pragma unroll
    for (int gt = 0; gt < 2; ++gt)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[0][gt][r] = bf16_round(acc[0][gt][r]);
    // GELU + mean over time: row sums inside each 16-lane row
#pragma unroll
    for (int gt = 0; gt < 2; ++gt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float sacc = jl < T1 ? gelu_f(acc[0][gt][r]) : 0.f;
        sacc += row_shr<8>(sacc);
        sacc += row_shr<4>(sacc);
        sacc += row_shr<2>(sacc);
        sacc += row_shr<1>(sacc);
        if (jl == 15) featL[gt * 16 + 4 * q + r] = sacc * inv_t;
      }
    wave_lds_sync();
    if (lane < n_cls) {
      float l = fcs[n_cls * F + lane];
#pragma unroll
      for (int g = 0; g < F; ++g) l = fmaf(fcs[lane * F + g], featL[g], l);
      logL[lane] = l;
      if (live) a.logits[item * n_cls + lane] = l;
    }
    wave_lds_sync();
    float mx = -INFINITY, lse = 0.f;
    int am = 0, yv = 0;
    for (int c = 0; c < n_cls; ++c) {
      const float v = logL[c];
      if (v > mx) { mx = v; am = c; }                          // strict '>' keeps the lowest index on ties (torch.argmax)
    }
    if (lane == 0 && live) a.pred[item] = am;
    if (a.labels) {
      float se = 0.f;
      for (int c = 0; c < n_cls; ++c) se += expf(logL[c] - mx);
      yv = a.label_bytes == 1 ? (int)((const unsigned char*)a.labels)[item] : (int)((const long long*)a.labels)[item];
      lse = mx + logf(se);
      if (lane == 0 && live) loss_acc += (lse - logL[yv]) * a.grad_scale;
    }
    if (!a.labels || !a.train) continue;
    // dlogits -> LDS (zero for a dead slot); G4 = dfeat/T1 * GELU'(A4)
    if (lane < n_cls) logL[16 + lane] = live ? (expf(logL[lane] - lse) - (lane == yv ? 1.f : 0.f)) * a.grad_scale : 0.f;
    wave_lds_sync();
#pragma unroll
    for (int gt = 0; gt < 2; ++gt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int g = gt * 16 + 4 * q + r;
        float d = 0.f;
        for (int c = 0; c < n_cls; ++c) d = fmaf(fcs[c * F + g], logL[16 + c], d);
        acc[0][gt][r] = d * inv_t * gelu_grad_f(acc[0][gt][r]);
      }
    store_tile(acc[0], tG);
    // FC gradients: flat element e = lane + 64 j of [n_cls][F]
#pragma unroll
    for (int j = 0; j < kTailMaxCls * F / 64; ++j) {
      const int e = lane + 64 * j;
      if (e < n_cls * F) accfc[j] = fmaf(logL[16 + (e >> 5)], featL[e & 31], accfc[j]);
    }
    if (lane < n_cls) accb += logL[16 + lane];
    wave_lds_sync();
    fused_wgrad_bf16(priv_base + 2 * tile_b, priv_base + tile_b, 0, 0, 1, 1, q, jl, accW4[0]);   // dW4 += G4 (*) A3
    fused_wgrad_bf16(priv_base + 2 * tile_b, priv_base + tile_b, 1, 0, 1, 1, q, jl, accW4[1]);
    clear();
    fused_conv_bf16<1>(w4ts, tG, tt0, lane, acc);              // G3
    store_tile(acc[0], tH);
    wave_lds_sync();
    fused_wgrad_bf16(priv_base + 3 * tile_b, priv_base, 0, 0, 1, 1, q, jl, accW3[0]);            // dW3 += G3 (*) A2
    fused_wgrad_bf16(priv_base + 3 * tile_b, priv_base, 1, 0, 1, 1, q, jl, accW3[1]);
    clear();
    fused_conv_bf16<1>(w3ts, tH, tt0, lane, acc);              // G2
    wave_lds_sync();                                           // every read of tG (wgrad4, cnn4 data gradient) is done
    store_tile(acc[0], tG);
    wave_lds_sync();
    if (live && lane < 4 * T1)
      reinterpret_cast<uint4*>((unsigned short*)a.g2 + item * tile)[lane] = *reinterpret_cast<const uint4*>(tG + 128 + lane * 16);
    wave_lds_sync();                                           // tiles are reused by the next item
  }
  if (!a.labels) return;
  // the waves' accumulators -> one slab per workgroup (tail_combine_store: three copies, three phases for eight waves)
  tail_combine_store<NW, 3>(smem, wave, lane,
                            [&](int layer, int gt, int ct, int k) -> f32x4 { return layer ? accW4[ct][gt][k] : accW3[ct][gt][k]; },
                            accfc, accb, loss_acc, a.part + (int64_t)blockIdx.x * a.slab, a.slab, n_cls);
}

// ---------------------------------------------------------------------------------------
// GELU + mean over time (fast.py:117-118) and its backward (in place on the activation).
// one 16-lane row per (item, zone, filter) row of length T
// ---------------------------------------------------------------------------------------
template <typename AT>
__global__ __launch_bounds__(256) void gelu_mean_fwd_kernel(const void* __restrict__ a, float* __restrict__ feat,
                                                            int64_t rows, int T) {
  const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
  const int i = threadIdx.x & 15;
  float s = 0.f;
  if (row < rows) {
    for (int t = i; t < T; t += 16) s += gelu_f(Act<AT>::ld(a, row * T + t));
  }
  s += row_shr<8>(s);   // lanes >= 8 accumulate lanes - 8 ... finish with a butterfly via shifts
  s += row_shr<4>(s);
  s += row_shr<2>(s);
  s += row_shr<1>(s);
  if (row < rows && i == 15) feat[row] = s / (float)T;
}

// PRE: the buffer already holds GELU'(A) (the fused forward stored it): only the scaling is left
template <typename AT, bool PRE = false>
__global__ __launch_bounds__(256) void gelu_mean_bwd_kernel(void* __restrict__ a, const float* __restrict__ dfeat,
                                                            int64_t rows, int T) {
  const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
  const int i = threadIdx.x & 15;
  if (row >= rows) return;
  const float g = dfeat[row] / (float)T;
  for (int t = i; t < T; t += 16) {
    const float v = Act<AT>::ld(a, row * T + t);
    Act<AT>::st(a, row * T + t, g * (PRE ? v : gelu_grad_f(v)));
  }
}

// Gradient w.r.t. the input trials: dx[b][chan][n*S + t] += sum_{g,k} Weff[g][c][k] G2[item][g][t - k]  (full
// correlation of the first, fused layer; zones own disjoint channels, overlapping windows add).  One workgroup per
// (window, zone) item with the G2 tile in LDS; not a hot path (saliency / SHAP-style attributions), plain VALU.
// With at most two windows covering a sample (slide_step >= window_len / 2) the float atomics commute exactly.
__global__ __launch_bounds__(256) void conv5_dx_kernel(const float* __restrict__ g2, const float* __restrict__ wfrag,
                                                       const ZoneDesc* __restrict__ zones,
                                                       const int* __restrict__ chan_idx, float* __restrict__ dx,
                                                       int Z, int F, int T1, int W, int Ctot, int Tx, int N, int S) {
  extern __shared__ __attribute__((aligned(16))) float smem[];          // [F][T1]
  const int64_t item = blockIdx.x;
  const int z = blockIdx.y;
  const ZoneDesc zd = zones[z];
  const int GT = F / 16;
  const float* src = g2 + (item * Z + z) * (int64_t)(F * T1);
  for (int e = threadIdx.x; e < F * T1; e += 256) smem[e] = src[e];
  __syncthreads();
  const int64_t b = item / N;
  const int n = (int)(item - b * N);
  const float* wz = wfrag + zd.eff_off;
  for (int e = threadIdx.x; e < zd.cin * W; e += 256) {
    const int c = e / W, t = e - c * W;
    const int cg = c >> 2, cl = (c & 3) * 16;
    float acc = 0.f;
    for (int g = 0; g < F; ++g) {
      const float* gr = smem + g * T1;
      const int gt = g >> 4, gl = g & 15;
#pragma unroll
      for (int k = 0; k < kTaps; ++k) {
        const int tt = t - k;
        if (tt >= 0 && tt < T1) acc = fmaf(wz[((int64_t)(cg * kTaps + k) * GT + gt) * 64 + cl + gl], gr[tt], acc);
      }
    }
    atomicAdd(&dx[(b * Ctot + chan_idx[zd.idx_off + c]) * (int64_t)Tx + (int64_t)n * S + t], acc);
  }
}

// ---------------------------------------------------------------------------------------
// Weight gradient: dW[g][c][k] = sum_{item,t} dOut[item][g][t] * In[item][c][t + k - pad].
// M = g, N = c (16 per wave), K = t (4 per MFMA); 5 accumulators per (g-tile) for the taps.
// MODE 0 appends a virtual all-ones channel (index cin) whose tap-0 column is dbias.
// Each workgroup reduces a contiguous range of items and writes one partial slab.
// ---------------------------------------------------------------------------------------
struct WgradArgs {
  const void* dout;          // [items][Z][F][Tout], type AT
  const void* in;            // MODE 0: raw fp32 trials, MODE 1: [items][Z][F][Tin] of type AT
  float* part;               // [n_wg * n_grp][slab_size]
  const ZoneDesc* zones;
  const int* chan_idx;
  int64_t items, slab_size;
  int64_t wz_stride;         // MODE 1: per-zone stride inside a slab
  int items_per_wg, IPS;     // items per workgroup, items per LDS stage
  int seg_len;               // > 0: rows are staged in time segments of this many output samples (long windows)
  int CW;                    // input channels staged per workgroup (16, 32, 48 or 64)
  int lin;                   // 1: the CW rows of an item are one contiguous block and RSi == Tin
  int Z, F, Tin, Tout, pad, RSo, RSi;
  int Ctot, Tx, N, S;
};

// Wave roles: a wave owns one of the n_ct = CW/16 channel tiles and every filter tile (the B fragments are
// shared by both); the 4/n_ct wave groups split the time range of each item and write separate slabs
// (summed by reduce_slabs_kernel).
// LDS rows are unpadded copies (RSo >= Tout, RSi >= Tin); range / padding handled by masks.
template <int MODE, typename AT, bool BOTH>
__global__ __launch_bounds__(256) void conv5_wgrad_kernel(WgradArgs a) {
  using IT = typename std::conditional<MODE == 0, float, AT>::type;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int z = blockIdx.y;
  const ZoneDesc zd = a.zones[z];
  const int cin = (MODE == 0) ? zd.cin + 1 : a.F;          // + ones channel (dbias)
  const int c_base = blockIdx.z * a.CW;
  if (c_base >= cin) return;
  const int GT = a.F / 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane >> 4, jl = lane & 15;
  const int n_ct = a.CW / 16;                              // 1, 2 or 4 channel tiles
  constexpr bool both = BOTH;                              // host: BOTH == (F == 32): a wave owns both filter tiles
  const int n_grp = 4 / n_ct;                              // wave groups split the time range, one slab each
  const int ct = wave % n_ct;
  const int gsel = 0;
  const int grp = wave / n_ct;
  const int c_tile = c_base + ct * 16;
  const bool wave_live = c_tile < cin && grp < n_grp;
  const int n_real = (MODE == 0) ? cin - 1 : cin;           // channels that exist in memory
  const int do_sz = (a.IPS * a.F * a.RSo + 3) & ~3;
  float* do_tile = smem;                                    // [IPS][F][RSo]
  float* in_tile = smem + do_sz;                            // [IPS][CW][RSi]
  const int64_t i_lo = (int64_t)blockIdx.x * a.items_per_wg;
  const int64_t i_hi = (i_lo + a.items_per_wg) < a.items ? (i_lo + a.items_per_wg) : a.items;
  const int c_mine = c_tile + jl;                           // channel of this lane's B column
  const bool c_real = c_mine < n_real;
  const bool c_ones = (MODE == 0) && c_mine == cin - 1;

  f32x4 acc0[kTaps], acc1[kTaps];
#pragma unroll
  for (int k = 0; k < kTaps; ++k) {
    acc0[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc1[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const int cw_real = (n_real - c_base) < a.CW ? (n_real - c_base) : a.CW;   // real rows to stage (may be <= 0)

  // t_base / t_cnt: the staged time range of the rows (whole rows, or one segment of a row too long for the LDS
  // tile); org = global sample index of LDS column 0 of the input rows
  auto compute = [&](const float* do_tile, const float* in_tile, int n_it, int t_base, int t_cnt, int org) {
    if (wave_live) {
      const int row_b = c_real ? (ct * 16 + jl) : 0;
      const int Tk = (t_cnt + 3) & ~3;
      // wave groups split the reduction (time) range of every item, so one staged item keeps all waves busy.
      // The (item, K-step) pairs of the stage form one flat sequence; the fragments of step s+1 are fetched from
      // LDS while the MFMAs of step s run, and no branch stands between the MFMAs.
      const int nk = (Tk / 4 - grp + n_grp - 1) / n_grp;       // K-steps of this wave group per item
      const int n_step = n_it * nk;
      struct Frag {
        float a0, a1;
        float b[kTaps];
      };
      int l_ii = 0, l_kk = 0;                                  // loads are issued in step order: running (item, K-step)
      auto load = [&](int, Frag& f) {
        const int ii = l_ii;
        const int ta = (grp + l_kk * n_grp) * 4 + q;
        if (++l_kk == nk) { l_kk = 0; ++l_ii; }
        const float* dro = do_tile + (ii * a.F + gsel * 16 + jl) * a.RSo;
        const float* iro = in_tile + (ii * a.CW + row_b) * a.RSi;
        const bool a_ok = ta < t_cnt;
        const int tac = a_ok ? ta : 0;
        const float v0 = dro[tac];
        f.a0 = a_ok ? v0 : 0.f;
        if (BOTH) {
          const float v1 = dro[16 * a.RSo + tac];
          f.a1 = a_ok ? v1 : 0.f;
        }
#pragma unroll
        for (int k = 0; k < kTaps; ++k) {
          const int idx = t_base + ta + k - a.pad;               // global sample of the input row
          const bool in_rng = a_ok && idx >= 0 && idx < a.Tin;
          const float bf = iro[in_rng ? idx - org : 0];
          f.b[k] = (c_real && in_rng) ? bf : ((c_ones && a_ok && idx >= 0 && idx < a.Tout) ? 1.f : 0.f);
        }
      };
      auto mma = [&](const Frag& f) {
#pragma unroll
        for (int k = 0; k < kTaps; ++k) {
          acc0[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a0, f.b[k], acc0[k], 0, 0, 0);
          if (BOTH) acc1[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a1, f.b[k], acc1[k], 0, 0, 0);
        }
      };
      Frag f0, f1;
      if (n_step > 0) load(0, f0);
      for (int st = 0; st < n_step; st += 2) {
        if (st + 1 < n_step) load(st + 1, f1);
        mma(f0);
        if (st + 1 < n_step) {
          if (st + 2 < n_step) load(st + 2, f0);
          mma(f1);
        }
      }
    }
  };

  if (a.seg_len > 0) {
    // rows longer than the LDS tile: one item per stage, the time range in segments of seg_len output samples
    // (row-wise staging at the segment origin; the accumulators simply keep summing over the segments)
    for (int64_t item = i_lo; item < i_hi; ++item) {
      for (int t_base = 0; t_base < a.Tout; t_base += a.seg_len) {
        const int t_cnt = (a.Tout - t_base) < a.seg_len ? (a.Tout - t_base) : a.seg_len;
        const int org = t_base - a.pad, in_cnt = t_cnt + kTaps - 1;
        __syncthreads();
        for (int r = wave; r < a.F; r += 4) {
          const int64_t soff = (((item * a.Z + z) * a.F) + r) * (int64_t)a.Tout + t_base;
          for (int t = lane; t < t_cnt; t += 64) do_tile[r * a.RSo + t] = Act<AT>::ld(a.dout, soff + t);
        }
        for (int r = wave; r < cw_real; r += 4) {
          int64_t soff;
          if (MODE == 0) {
            const int64_t b = item / a.N;
            const int n = (int)(item - b * a.N);
            soff = (b * a.Ctot + a.chan_idx[zd.idx_off + c_base + r]) * (int64_t)a.Tx + (int64_t)n * a.S;
          } else {
            soff = ((item * a.Z + z) * a.F + c_base + r) * (int64_t)a.Tin;
          }
          for (int t = lane; t < in_cnt; t += 64) {
            const int idx = org + t;
            in_tile[r * a.RSi + t] = (idx >= 0 && idx < a.Tin) ? Act<AT>::rnd(Act<IT>::ld(a.in, soff + idx)) : 0.f;
          }
        }
        __syncthreads();
        compute(do_tile, in_tile, 1, t_base, t_cnt, org);
      }
    }
  }
  for (int64_t is = a.seg_len > 0 ? i_hi : i_lo; is < i_hi; is += a.IPS) {
    const int n_it = (int)((i_hi - is) < a.IPS ? (i_hi - is) : a.IPS);
    __syncthreads();
    // dOut: [F][Tout] of an item is contiguous; few items -> the whole workgroup copies each block
    const bool wg_copy = n_it < 4;
    const int tid0 = wg_copy ? (int)threadIdx.x : lane;
    const int tstep = wg_copy ? 256 : 64;
    {
      const int cnt = a.F * a.Tout;
      for (int ii = wg_copy ? 0 : wave; ii < n_it; ii += wg_copy ? 1 : 4) {
        const int64_t soff = ((is + ii) * a.Z + z) * (int64_t)cnt;
        float* dst = do_tile + ii * a.F * a.RSo;
        if constexpr (!Act<AT>::kBf16) {
          const float* src = (const float*)a.dout + soff;
          if (a.RSo == a.Tout && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0 && (cnt & 3) == 0) {
            glds_copy16_strided(src, dst, cnt >> 2, tid0 - lane, tstep, lane);     // LDS-DMA: every piece in flight at once
            continue;
          }
        }
        if (a.RSo == a.Tout) {
          for (int e = tid0; e < cnt; e += tstep) dst[e] = Act<AT>::ld(a.dout, soff + e);
        } else {
          for (int g = 0; g < a.F; ++g)
            for (int t = tid0; t < a.Tout; t += tstep) dst[g * a.RSo + t] = Act<AT>::ld(a.dout, soff + g * a.Tout + t);
        }
      }
    }
    if (cw_real > 0) {
      if (a.lin) {
        const int cnt = cw_real * a.Tin;
        for (int ii = wg_copy ? 0 : wave; ii < n_it; ii += wg_copy ? 1 : 4) {
          const int64_t item = is + ii;
          const int64_t soff = (MODE == 0) ? (item * a.Ctot + a.chan_idx[zd.idx_off + c_base]) * (int64_t)a.Tx
                                           : ((item * a.Z + z) * a.F + c_base) * (int64_t)a.Tin;
          float* dst = in_tile + ii * a.CW * a.RSi;
          if constexpr (std::is_same<IT, float>::value && !Act<AT>::kBf16) {
            const float* src = (const float*)a.in + soff;
            if ((((uintptr_t)src | (uintptr_t)dst) & 15) == 0 && (cnt & 3) == 0) {
              glds_copy16_strided(src, dst, cnt >> 2, tid0 - lane, tstep, lane);   // LDS-DMA: every piece in flight at once
            } else {
              for (int e = tid0; e < cnt; e += tstep) dst[e] = src[e];
            }
          } else {
            for (int e = tid0; e < cnt; e += tstep) dst[e] = Act<AT>::rnd(Act<IT>::ld(a.in, soff + e));
          }
        }
      } else {
        const int rows = n_it * cw_real;
        for (int r = wave; r < rows; r += 4) {
          const int ii = r / cw_real, cc = r - ii * cw_real;
          const int64_t item = is + ii;
          int64_t soff;
          if (MODE == 0) {
            const int64_t b = item / a.N;
            const int n = (int)(item - b * a.N);
            soff = (b * a.Ctot + a.chan_idx[zd.idx_off + c_base + cc]) * (int64_t)a.Tx + (int64_t)n * a.S;
          } else {
            soff = ((item * a.Z + z) * a.F + c_base + cc) * (int64_t)a.Tin;
          }
          float* dst = in_tile + (ii * a.CW + cc) * a.RSi;
          for (int t = lane; t < a.Tin; t += 64) dst[t] = Act<AT>::rnd(Act<IT>::ld(a.in, soff + t));
        }
      }
    }
    __syncthreads();
    compute(do_tile, in_tile, n_it, 0, a.Tout, 0);
  }
  if (!wave_live) return;
  float* slab = a.part + ((int64_t)blockIdx.x * n_grp + grp) * a.slab_size +
                ((MODE == 0) ? zd.wg_off : (int64_t)z * a.wz_stride);
  if (c_mine < cin) {
#pragma unroll
    for (int k = 0; k < kTaps; ++k)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int g = gsel * 16 + 4 * q + r;
        slab[((int64_t)g * cin + c_mine) * kTaps + k] = acc0[k][r];
        if (both) slab[((int64_t)(g + 16) * cin + c_mine) * kTaps + k] = acc1[k][r];
      }
  }
}

// First-layer weight gradient for wide inputs (companion of conv5_fwd_glds_kernel; same preconditions).
// A workgroup owns 64 input channels (one 16-channel tile per wave, all GT filter tiles) and a range of items;
// dOut [F][Tout] and the 64 input rows of IPS items are double-buffered in LDS by LDS-DMA; the MFMA loop is
// branch-free with register-prefetched fragments (K = 4 time steps, one accumulator per tap).  The dbias
// column (virtual all-ones channel `cin`) rides along in wave 0 of channel group 0.
template <int GT>
__global__ __launch_bounds__(256) void conv5_wgrad_wide_kernel(WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int z = blockIdx.y;
  const ZoneDesc zd = a.zones[z];
  const int cin = zd.cin, cin1 = zd.cin + 1;
  const int c_base = blockIdx.z * 64;
  if (c_base >= cin) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane >> 4, jl = lane & 15;
  const int cw = (cin - c_base) < 64 ? (cin - c_base) : 64;         // real rows of this channel group (multiple of 4)
  const int c_mine = c_base + wave * 16 + jl;
  const bool wave_live = c_base + wave * 16 < cin;
  const bool with_bias = blockIdx.z == 0 && wave == 0;
  const int do_len = a.F * a.Tout, in_len = 64 * a.Tin;             // per item (in_len: stride; cw*Tin are filled)
  const int item_len = (do_len + in_len + 3) & ~3;
  const int buf_len = a.IPS * item_len + 32;
  const int64_t i_lo = (int64_t)blockIdx.x * a.items_per_wg;
  const int64_t i_hi = (i_lo + a.items_per_wg) < a.items ? (i_lo + a.items_per_wg) : a.items;
  const int chan0 = a.chan_idx[zd.idx_off];
  const int nks = (a.Tout + 3) >> 2;
  // the tail of both buffers is read (masked) by the last rows' junk columns: keep it finite
  for (int e = threadIdx.x; e < 2 * buf_len + 8; e += 256) smem[e] = 0.f;
  __syncthreads();

  auto stage = [&](int64_t is, int s) {
    float* buf = smem + s * buf_len;
    const int n_it = (int)((i_hi - is) < a.IPS ? (i_hi - is) : a.IPS);
    for (int ii = wave; ii < n_it; ii += 4) {
      float* dst = buf + ii * item_len;
      glds_copy16((const float*)a.dout + ((is + ii) * a.Z + z) * (int64_t)do_len, dst, do_len >> 2, lane);
      glds_copy16((const float*)a.in + ((is + ii) * a.Ctot + chan0 + c_base) * (int64_t)a.Tx, dst + do_len,
                  (cw * a.Tin) >> 2, lane);
    }
  };

  f32x4 acc[GT][kTaps], accb[GT];
#pragma unroll
  for (int g = 0; g < GT; ++g) {
    accb[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < kTaps; ++k) acc[g][k] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  struct Frag {
    float af[GT];
    float bf[kTaps];
    float one;
  };
  const int a_off = jl * a.Tout + q;                                   // dOut[g = jl (+16 gt)][t0 + q]
  const int b_off = do_len + (wave * 16 + jl) * a.Tin + q;             // In[c][t0 + q + k]
  int s = 0;
  if (i_lo < i_hi) stage(i_lo, 0);
  for (int64_t is = i_lo; is < i_hi; is += a.IPS, s ^= 1) {
    __syncthreads();                                                   // DMA of this stage retired; previous stage consumed
    if (is + a.IPS < i_hi) stage(is + a.IPS, s ^ 1);
    if (!wave_live) continue;
    const float* buf = smem + s * buf_len;
    const int n_it = (int)((i_hi - is) < a.IPS ? (i_hi - is) : a.IPS);
    // The reduction runs over (item, output step) pairs four at a time.  With Tout >= 4 the stage's pairs are taken
    // back to back -- K index 4 st + q is step f % Tout of item f / Tout, kept per lane and advanced by four per load
    // (the loads are issued in step order) -- so 13 output steps cost 13/4 K steps per item instead of 4.
    const bool packed = a.Tout >= 4;
    const int n_step = packed ? (n_it * a.Tout + 3) >> 2 : n_it * nks;
    int p_ii = 0, p_t = q;                                              // this lane's next (item, step)
    auto load = [&](int st, Frag& f) {
      int ii, t0;
      bool ok;
      if (packed) {
        ii = p_ii; t0 = p_t - q;                                        // (a_off / b_off carry the + q)
        ok = p_ii < n_it;
        if (!ok) { ii = 0; t0 = -q; }                                   // past the stage's last pair: any valid address
        p_t += 4;
        if (p_t >= a.Tout) { p_t -= a.Tout; ++p_ii; }
      } else {
        ii = st / nks; t0 = (st - ii * nks) * 4;
        ok = t0 + q < a.Tout;
      }
      const float* ib = buf + ii * item_len + t0;
#pragma unroll
      for (int g = 0; g < GT; ++g) {
        const float v = ib[a_off + g * 16 * a.Tout];
        f.af[g] = ok ? v : 0.f;
      }
#pragma unroll
      for (int k = 0; k < kTaps; ++k) {
        const float v = ib[b_off + k];
        f.bf[k] = ok ? v : 0.f;
      }
      f.one = ok ? 1.f : 0.f;
    };
    auto mma = [&](const Frag& f) {
#pragma unroll
      for (int k = 0; k < kTaps; ++k)
#pragma unroll
        for (int g = 0; g < GT; ++g) acc[g][k] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.af[g], f.bf[k], acc[g][k], 0, 0, 0);
      if (with_bias) {
#pragma unroll
        for (int g = 0; g < GT; ++g) accb[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.af[g], f.one, accb[g], 0, 0, 0);
      }
    };
    Frag f0, f1;
    load(0, f0);
    for (int st = 0; st < n_step; st += 2) {
      if (st + 1 < n_step) load(st + 1, f1);
      mma(f0);
      if (st + 1 < n_step) {
        if (st + 2 < n_step) load(st + 2, f0);
        mma(f1);
      }
    }
  }
  if (!wave_live) return;
  float* slab = a.part + (int64_t)blockIdx.x * a.slab_size + zd.wg_off;
#pragma unroll
  for (int g = 0; g < GT; ++g)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gg = g * 16 + 4 * q + r;
      if (c_mine < cin) {
#pragma unroll
        for (int k = 0; k < kTaps; ++k) slab[((int64_t)gg * cin1 + c_mine) * kTaps + k] = acc[g][k][r];
      }
      if (with_bias && jl < kTaps) slab[((int64_t)gg * cin1 + cin) * kTaps + jl] = jl == 0 ? accb[g][r] : 0.f;
    }
}

// ---------------------------------------------------------------------------------------
// BASELINE config 3: first-layer weight gradient on the bf16 matrix cores (companion of conv5_fwd_bf16_kernel).
//   dW[g][c][k] = sum_items sum_t G[item][t][g] x[item][c][t + k],   v_mfma_f32_16x16x32_bf16 with
//   M = 16 filters, N = 16 input channels, K = 32 = 2 items x 16 output steps (rows t >= Tout are zero).
// A operand: the gradient arrives as bf16 [item][t][filter] (the bf16 tail writes it so); its LDS image keeps that
//   layout, padded to 16 zero-filled rows per item, and ds_read_b64_tr_b16 delivers the transposed 8-step fragment
//   (lane g <- column g of rows t0..t0+7: two 4-row blocks).
// B operand: lane (channel c, q) reads the 12 consecutive fp32 samples x[c][8 (q & 1) .. + 11] of item q >> 1 once
//   and packs them to bf16 twice (even and odd pairs): the five tap windows are register subsets, no shuffles.
//   Samples past the end of a row belong to the next row (finite) and meet zero rows of the A operand.
// Same workgroup roles, LDS-DMA double buffering, slab layout and dbias column as conv5_wgrad_wide_kernel.
// Preconditions (host): F = 32 (64-byte gradient rows), Tout <= 13, Tin <= 17, IPS even.
// ---------------------------------------------------------------------------------------
template <int GT, bool IN16 = false>
__global__ __launch_bounds__(256) void conv5_wgrad_wide_bf16_kernel(WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int z = blockIdx.y;
  const ZoneDesc zd = a.zones[z];
  const int cin = zd.cin, cin1 = zd.cin + 1;
  const int c_base = blockIdx.z * 64;
  if (c_base >= cin) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane >> 4, jl = lane & 15;
  const int cw = (cin - c_base) < 64 ? (cin - c_base) : 64;         // real rows of this channel group (multiple of 4)
  const int c_mine = c_base + wave * 16 + jl;
  const bool wave_live = c_base + wave * 16 < cin;
  const bool with_bias = blockIdx.z == 0 && wave == 0;
  const int do_len = 16 * a.F / 2;                                  // floats: 16 rows x F bf16 (rows >= Tout stay zero)
  const int do_real4 = (a.Tout * a.F * 2) >> 4;                     // 16-byte pieces actually copied
  const int in_len = 64 * a.Tin;                                    // per item (stride; cw * Tin are filled)
  const int item_len = (do_len + in_len + 3) & ~3;
  const int buf_len = a.IPS * item_len + 32;
  const int64_t i_lo = (int64_t)blockIdx.x * a.items_per_wg;
  const int64_t i_hi = (i_lo + a.items_per_wg) < a.items ? (i_lo + a.items_per_wg) : a.items;
  const int chan0 = a.chan_idx[zd.idx_off];
  // zero everything once: the pad rows of the gradient images are never written by the DMA, and the tail of both
  // buffers is read by the last rows' junk columns
  for (int e = threadIdx.x; e < 2 * buf_len + 8; e += 256) smem[e] = 0.f;
  __syncthreads();

  auto stage = [&](int64_t is, int s) {
    float* buf = smem + s * buf_len;
    const int n_it = (int)((i_hi - is) < a.IPS ? (i_hi - is) : a.IPS);
    for (int ii = wave; ii < a.IPS; ii += 4) {
      float* dst = buf + ii * item_len;
      if (ii < n_it) {
        glds_copy16(reinterpret_cast<const float*>((const unsigned short*)a.dout +
                                                   ((is + ii) * a.Z + z) * (int64_t)a.Tout * a.F),
                    dst, do_real4, lane);
        if (IN16)      // bf16 input rows (2 Tin bytes each): cw is a multiple of 8 (host check), whole 16-byte pieces
          glds_copy16(reinterpret_cast<const float*>((const unsigned short*)a.in +
                                                     ((is + ii) * a.Ctot + chan0 + c_base) * (int64_t)a.Tx),
                      dst + do_len, (cw * a.Tin) >> 3, lane);
        else
          glds_copy16((const float*)a.in + ((is + ii) * a.Ctot + chan0 + c_base) * (int64_t)a.Tx, dst + do_len,
                      (cw * a.Tin) >> 2, lane);
      } else {
        // a missing second item of the last pair: its gradient image must read as zero (the x rows may be stale)
        for (int e = lane; e < do_len; e += 64) dst[e] = 0.f;
      }
    }
  };

  f32x4 acc[GT][kTaps], accb[GT];
#pragma unroll
  for (int g = 0; g < GT; ++g) {
    accb[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < kTaps; ++k) acc[g][k] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const int ih = q >> 1, t0 = 8 * (q & 1);                            // this lane group's item of the pair, first step
  // transposed read: lane 4 r + p of a 16-lane group supplies row r, columns 4 p .. 4 p + 3 of a 4 x 16 block
  const unsigned a_byte = (unsigned)((t0 + (jl >> 2)) * a.F + 4 * (jl & 3)) * 2u;
  const int b_off = do_len + (wave * 16 + jl) * a.Tin + t0;          // x[c][t0 ...]
  const unsigned smem_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)smem;
  const bf16x8 ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
  int s = 0;
  if (i_lo < i_hi) stage(i_lo, 0);
  for (int64_t is = i_lo; is < i_hi; is += a.IPS, s ^= 1) {
    __syncthreads();                                                   // DMA of this stage retired; previous stage consumed
    if (is + a.IPS < i_hi) stage(is + a.IPS, s ^ 1);
    if (!wave_live) continue;
    const int n_it = (int)((i_hi - is) < a.IPS ? (i_hi - is) : a.IPS);
    for (int ip = 0; ip < n_it; ip += 2) {
      const int item_off = (s * buf_len + (ip + ih) * item_len);       // floats from smem
      // A: G^T fragments of both filter tiles (rows t0 .. t0 + 7 of this lane group's item)
      bf16x8 af[GT];
#pragma unroll
      for (int g = 0; g < GT; ++g) {
        const unsigned addr = smem_base + (unsigned)item_off * 4u + a_byte + (unsigned)(g * 16 * 2);
        uint2 lo, hi;
        asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:%3\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(lo), "=&v"(hi)
                     : "v"(addr), "i"(4 * 32 * 2)
                     : "memory");
        af[g] = __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
      }
      // B: 12 samples of this lane's channel, packed as even pairs (e0e1, e2e3, ...) and odd pairs (e1e2, e3e4, ...)
      unsigned int pe[6], po[5];
      if constexpr (IN16) {
        const unsigned short* xr = reinterpret_cast<const unsigned short*>(smem + item_off + do_len) +
                                   (wave * 16 + jl) * a.Tin + t0;
        unsigned int e[12];
#pragma unroll
        for (int n = 0; n < 12; ++n) e[n] = xr[n];
#pragma unroll
        for (int n = 0; n < 6; ++n) pe[n] = e[2 * n] | (e[2 * n + 1] << 16);
#pragma unroll
        for (int n = 0; n < 5; ++n) po[n] = e[2 * n + 1] | (e[2 * n + 2] << 16);
      } else {
      const float* xr = smem + item_off + b_off;
      float e[12];
#pragma unroll
      for (int n = 0; n < 12; ++n) e[n] = xr[n];
#pragma unroll
      for (int n = 0; n < 6; ++n) pe[n] = bf16_pack(e[2 * n], e[2 * n + 1]);
#pragma unroll
      for (int n = 0; n < 5; ++n) po[n] = bf16_pack(e[2 * n + 1], e[2 * n + 2]);
      }
#pragma unroll
      for (int k = 0; k < kTaps; ++k) {
        const int h = k >> 1;
        const uint4 bw = (k & 1) ? make_uint4(po[h], po[h + 1], po[h + 2], po[h + 3])
                                 : make_uint4(pe[h], pe[h + 1], pe[h + 2], pe[h + 3]);
        const bf16x8 bfr = __builtin_bit_cast(bf16x8, bw);
#pragma unroll
        for (int g = 0; g < GT; ++g) acc[g][k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[g], bfr, acc[g][k], 0, 0, 0);
      }
      if (with_bias) {
#pragma unroll
        for (int g = 0; g < GT; ++g) accb[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[g], ones, accb[g], 0, 0, 0);
      }
    }
  }
  if (!wave_live) return;
  float* slab = a.part + (int64_t)blockIdx.x * a.slab_size + zd.wg_off;
#pragma unroll
  for (int g = 0; g < GT; ++g)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gg = g * 16 + 4 * q + r;
      if (c_mine < cin) {
#pragma unroll
        for (int k = 0; k < kTaps; ++k) slab[((int64_t)gg * cin1 + c_mine) * kTaps + k] = acc[g][k][r];
      }
      if (with_bias && jl < kTaps) slab[((int64_t)gg * cin1 + cin) * kTaps + jl] = jl == 0 ? accb[g][r] : 0.f;
    }
}

// sum the per-workgroup slabs: out[e] = sum_s part[s][e].  Block = 64 elements x 4 slab groups
// (coalesced 256-B rows, 4 independent load streams per element, LDS combine).  blockIdx.y selects a run of
// L slabs (slab index = k * stride); with gridDim.y > 1 the run's sum is written over its own first slab, and
// a second launch (stride = L) adds the run sums -- a fixed order, so the result is deterministic.
// The final sums can land in up to three places (elements [0, n0) -> o0, [n0, n0 + n1) -> o1, the rest -> o2):
// the classifier tail's slab holds the cnn3/cnn4 gradients, the FC gradients and the loss back to back.
struct ReduceDst {
  float *o0, *o1, *o2;
  int64_t n0, n1;
};
__global__ __launch_bounds__(256) void reduce_slabs_kernel(float* __restrict__ part, ReduceDst out,
                                                           int64_t n, int n_slabs, int L, int stride) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int64_t e = (int64_t)blockIdx.x * 64 + lane;
  const int k0 = blockIdx.y * L;
  const int k1 = k0 + L < n_slabs ? k0 + L : n_slabs;
  float s0 = 0.f, s1 = 0.f;
  if (e < n) {
    int k = k0 + grp;
    for (; k + 4 < k1; k += 8) {
      s0 += part[(int64_t)k * stride * n + e];
      s1 += part[(int64_t)(k + 4) * stride * n + e];
    }
    if (k < k1) s0 += part[(int64_t)k * stride * n + e];
  }
  red[grp][lane] = s0 + s1;
  __syncthreads();
  if (grp == 0 && e < n) {
    const float v = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    if (gridDim.y > 1) part[(int64_t)k0 * stride * n + e] = v;
    else if (e < out.n0) out.o0[e] = v;
    else if (e < out.n0 + out.n1) out.o1[e - out.n0] = v;
    else out.o2[e - out.n0 - out.n1] = v;
  }
}

static inline void launch_reduce_slabs(float* part, ReduceDst out, int64_t n, int n_slabs, hipStream_t st) {
  const unsigned bx = (unsigned)cdiv(n, 64);
  if (n_slabs >= 64 && bx < 1024) {            // few elements, many slabs: two levels so the whole chip takes part
    int L = 8;
    while (L * L < n_slabs) L *= 2;
    const int S = (int)cdiv(n_slabs, L);
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(bx, S), dim3(256), 0, st, part, out, n, n_slabs, L, 1);
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(bx, 1), dim3(256), 0, st, part, out, n, S, S, L);
  } else {
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(bx, 1), dim3(256), 0, st, part, out, n, n_slabs, n_slabs, 1);
  }
}
static inline void launch_reduce_slabs(float* part, float* out, int64_t n, int n_slabs, hipStream_t st) {
  launch_reduce_slabs(part, ReduceDst{out, nullptr, nullptr, n, 0}, n, n_slabs, st);
}

// The three slab sets the fused backward kernels leave (cnn4, cnn3, Weff) summed by ONE launch (two when a set has 64
// slabs or more), the cnn3 / cnn4 sums written straight to their places in the flat gradient block.  At the
// reference's batch of 64 a step is ~40 kernels of a few microseconds: as three reductions of two launches each plus
// two scatter copies these sums were 8 of them (~40 us of a 0.87 ms step).  Same element order as
// reduce_slabs_kernel: the sums are bit for bit the ones the separate launches produced.
struct SlabJob {
  float* part;        // [n_slabs][n]
  float* out;         // layer < 0: contiguous destination [n]; else the flat gradient block (scattered per zone)
  int64_t n;
  int n_slabs, L;     // L: slabs per run of the first level (== n_slabs: one level)
  int layer;          // 0 / 1: cnn3 / cnn4 block of every zone (n = Z F F 5 in zone order); -1: contiguous
  unsigned bx0;       // first blockIdx.x of this job
};
struct SlabJobs {
  SlabJob j[3];
  const ZoneDesc* zones;
  int F;
};
// level 0: every job in one level (runs of L == n_slabs) -> destinations; level 1: run sums over each run's first slab;
// level 2: the run sums (stride L) -> destinations
__global__ __launch_bounds__(256) void reduce_jobs_kernel(SlabJobs js, int level) {
  __shared__ float red[4][64];
  const int ji = blockIdx.x >= js.j[2].bx0 ? 2 : (blockIdx.x >= js.j[1].bx0 ? 1 : 0);
  const SlabJob jb = js.j[ji];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int64_t e = (int64_t)(blockIdx.x - jb.bx0) * 64 + lane, n = jb.n;
  const int two = jb.L < jb.n_slabs;                       // this job needs two levels
  int k0, k1, stride;
  if (level == 2) {
    if (!two) return;                                      // finished by the level-1 launch
    k0 = 0; k1 = (jb.n_slabs + jb.L - 1) / jb.L; stride = jb.L;
  } else {
    k0 = blockIdx.y * jb.L;
    if (k0 >= jb.n_slabs) return;
    k1 = k0 + jb.L < jb.n_slabs ? k0 + jb.L : jb.n_slabs; stride = 1;
  }
  float s0 = 0.f, s1 = 0.f;
  if (e < n) {
    int k = k0 + grp;
    for (; k + 4 < k1; k += 8) {
      s0 += jb.part[(int64_t)k * stride * n + e];
      s1 += jb.part[(int64_t)(k + 4) * stride * n + e];
    }
    if (k < k1) s0 += jb.part[(int64_t)k * stride * n + e];
  }
  red[grp][lane] = s0 + s1;
  __syncthreads();
  if (grp == 0 && e < n) {
    const float v = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    if (level == 1 && two) {
      jb.part[(int64_t)k0 * n + e] = v;
    } else if (jb.layer < 0) {
      jb.out[e] = v;
    } else {
      const int nz = js.F * js.F * kTaps;
      const int z = (int)(e / nz);
      const ZoneDesc zd = js.zones[z];
      jb.out[zd.p_off + js.F * kTaps + js.F + (int64_t)js.F * js.F * zd.cin + (int64_t)jb.layer * nz + (e - (int64_t)z * nz)] = v;
    }
  }
}

// part4 / part3: n34 elements in n_slabs34 slabs -> cnn4 / cnn3 blocks of dparams; part0: n0 elements in n_slabs0
// slabs -> wg0 (contiguous)
static inline int launch_reduce_fused_bwd(float* part4, float* part3, float* part0, int64_t n34, int n_slabs34, int64_t n0,
                                          int n_slabs0, float* dparams, float* wg0, const ZoneDesc* zones, int F,
                                          hipStream_t st) {
  SlabJobs js = {};
  js.zones = zones; js.F = F;
  const int64_t ns[3] = {n34, n34, n0};
  const int slabs[3] = {n_slabs34, n_slabs34, n_slabs0};
  float* parts[3] = {part4, part3, part0};
  unsigned bx = 0;
  int s_max = 1;
  bool two = false;
  for (int i = 0; i < 3; ++i) {
    SlabJob& j = js.j[i];
    j.part = parts[i]; j.n = ns[i]; j.n_slabs = slabs[i]; j.bx0 = bx;
    j.layer = i == 0 ? 1 : (i == 1 ? 0 : -1);
    j.out = i == 2 ? wg0 : dparams;
    const unsigned b = (unsigned)cdiv(ns[i], 64);
    j.L = slabs[i];
    if (slabs[i] >= 64 && b < 1024) {                      // few elements, many slabs: two levels (launch_reduce_slabs's rule)
      int L = 8;
      while (L * L < slabs[i]) L *= 2;
      j.L = L;
      const int S = (int)cdiv(slabs[i], L);
      if (S > s_max) s_max = S;
      two = true;
    }
    bx += b;
  }
  hipLaunchKernelGGL(reduce_jobs_kernel, dim3(bx, s_max), dim3(256), 0, st, js, two ? 1 : 0);
  if (two) hipLaunchKernelGGL(reduce_jobs_kernel, dim3(bx, 1), dim3(256), 0, st, js, 2);
  ISD_LAUNCH_CHECK();
  return ISD_OK;
}

// Chain dWeff / dbeff back to cnn1.weight, cnn1.bias, cnn2.weight.  grid = (blocks, zones):
//   blocks [0, nb2)          : dW2[g,f,c] = sum_k dWeff[g,c,k] W1[f,k] + dbeff[g] b1[f]      (one thread per element)
//   blocks [nb2, nb2 + 5F)   : dW1[f,k]   = sum_{g,c} dWeff[g,c,k] W2[g,f,c]                 (one block per output)
//   blocks [nb2 + 5F, +F)    : db1[f]     = sum_g dbeff[g] sum_c W2[g,f,c]
// kFbwThreads: 1024 for wide inputs (the spec-S classifier: 576 channels), 256 for the zone shapes (<= 15 channels:
// 480 terms per output, where the longer reduction tree of a 1024-thread workgroup costs more than it saves)
template <int kFbwThreads>
__global__ __launch_bounds__(kFbwThreads) void fused_bwd_kernel(const float* __restrict__ params,
                                                                const ZoneDesc* __restrict__ zones,
                                                                const float* __restrict__ dweff,
                                                                float* __restrict__ dparams, int F, int nb2) {
  __shared__ float red[kFbwThreads];
  const ZoneDesc zd = zones[blockIdx.y];
  const int cz = zd.cin, cin1 = cz + 1;
  const float* W1 = params + zd.p_off;
  const float* b1 = W1 + F * kTaps;
  const float* W2 = b1 + F;
  const float* dWe = dweff + zd.wg_off;                 // [F][cz+1][5]; channel cz, tap 0 = dbeff
  float* dW1 = dparams + zd.p_off;
  float* db1 = dW1 + F * kTaps;
  float* dW2 = db1 + F;
  const int blk = blockIdx.x;
  if (blk < nb2) {
    const int e = blk * kFbwThreads + threadIdx.x;
    if (e >= F * F * cz) return;
    const int c = e % cz, f = (e / cz) % F, g = e / (cz * F);
    float s = dWe[(g * cin1 + cz) * kTaps] * b1[f];
#pragma unroll
    for (int k = 0; k < kTaps; ++k) s = fmaf(dWe[(g * cin1 + c) * kTaps + k], W1[f * kTaps + k], s);
    dW2[e] = s;
    return;
  }
  const int o = blk - nb2;                              // output index: [0, 5F) -> dW1, [5F, 6F) -> db1
  if (o >= F * kTaps + F) return;
  // one workgroup per output, its F x cz terms spread over ALL 1024 threads as (g, c) pairs (18 terms per thread at
  // 576 channels, 1 at a 15-channel zone): with 256 threads walking c and looping g, a 576-channel output was 72
  // dependent-latency-bound terms per thread and a 15-channel zone used 15 of its threads
  float s = 0.f;
  const int n_pair = F * cz;
  if (o < F * kTaps) {
    const int f = o / kTaps, k = o - f * kTaps;
#pragma unroll 4
    for (int pi = threadIdx.x; pi < n_pair; pi += kFbwThreads) {
      const int g = pi / cz, c = pi - g * cz;
      s = fmaf(dWe[(g * cin1 + c) * kTaps + k], W2[((int64_t)g * F + f) * cz + c], s);
    }
  } else {
    const int f = o - F * kTaps;
#pragma unroll 4
    for (int pi = threadIdx.x; pi < n_pair; pi += kFbwThreads) {
      const int g = pi / cz, c = pi - g * cz;
      s = fmaf(dWe[(g * cin1 + cz) * kTaps], W2[((int64_t)g * F + f) * cz + c], s);
    }
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = kFbwThreads / 2; w > 0; w >>= 1) {
    if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (o < F * kTaps) dW1[o] = red[0];
    else db1[o - F * kTaps] = red[0];
  }
}

}  // namespace isd

using namespace isd;

struct isd_conv4_plan {
  int Ctot, Z, F, n_layers, W, S;
  int cz[kMaxZones];
  int64_t p_off[kMaxZones];
  int64_t n_params;
  int64_t eff_size;        // floats of all fused frag blocks
  int64_t conv_zstride;    // floats of one zone's cnn3/cnn4 frag block
  int64_t wg_size;         // floats of all dWeff blocks
  int max_cz;
  int contiguous;          // every zone's channel list is consecutive -> rows of a zone are adjacent in memory
  int dma_ok;              // additionally every zone starts on, and spans, a multiple of 4 channels (16-byte row blocks)
  int act_bf16;            // activations / activation gradients stored as bf16 (config 3)
  ZoneDesc* d_zones;
  int* d_idx;
};

static inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }
static inline int row_threads(int row_len) {   // power of two in [16, 256] covering a staged row
  int tw = 16;
  while (tw < row_len && tw < 256) tw <<= 1;
  return tw;
}

extern "C" int isd_conv4_plan_create(isd_conv4_plan** out, int c_total, int n_zones, const int* zone_sizes,
                                     const int* zone_channels, int feature_dim, int n_layers, int window_len,
                                     int slide_step) {
  ISD_CHECK_ARG(out && zone_sizes && zone_channels, "isd_conv4_plan_create: null argument");
  ISD_CHECK_ARG(n_zones >= 1 && n_zones <= kMaxZones, "isd_conv4_plan_create: n_zones=%d not in [1,%d]", n_zones,
                kMaxZones);
  ISD_CHECK_ARG(feature_dim == 16 || feature_dim == 32, "isd_conv4_plan_create: feature_dim=%d must be 16 or 32",
                feature_dim);
  ISD_CHECK_ARG(n_layers == 2 || n_layers == 4, "isd_conv4_plan_create: n_layers must be 2 or 4");
  ISD_CHECK_ARG(window_len >= kTaps && slide_step >= 1, "isd_conv4_plan_create: window_len=%d slide_step=%d",
                window_len, slide_step);
  isd_conv4_plan* p = new isd_conv4_plan();
  p->Ctot = c_total; p->Z = n_zones; p->F = feature_dim; p->n_layers = n_layers; p->W = window_len; p->S = slide_step;
  p->d_zones = nullptr; p->d_idx = nullptr; p->act_bf16 = 0;
  const int F = feature_dim, GT = F / 16;
  std::vector<ZoneDesc> zd(n_zones);
  std::vector<int> idx;
  int64_t po = 0, eo = 0, wo = 0;
  p->max_cz = 0;
  for (int z = 0; z < n_zones; ++z) {
    const int cz = zone_sizes[z];
    if (cz < 1 || cz > 4096) { delete p; set_error("isd_conv4_plan_create: zone %d has %d channels", z, cz); return ISD_ERR_INVALID; }
    for (int c = 0; c < cz; ++c) {
      const int ch = zone_channels[idx.size()];
      if (ch < 0 || ch >= c_total) { delete p; set_error("isd_conv4_plan_create: channel index %d outside [0,%d)", ch, c_total); return ISD_ERR_INVALID; }
      idx.push_back(ch);
    }
    p->cz[z] = cz;
    p->p_off[z] = po;
    if (cz > p->max_cz) p->max_cz = cz;
    zd[z].cin = cz;
    zd[z].idx_off = (int)idx.size() - cz;
    zd[z].p_off = po;
    zd[z].eff_off = eo;
    zd[z].wg_off = wo;
    po += (int64_t)F * kTaps + F + (int64_t)F * F * cz + (n_layers == 4 ? 2LL * F * F * kTaps : 0);
    eo += (int64_t)align_up(cz, kCK) / 4 * kTaps * GT * 64;   // chunk-aligned so chunk offsets are uniform
    wo += (int64_t)F * (cz + 1) * kTaps;
  }
  p->n_params = po; p->eff_size = eo; p->wg_size = wo;
  p->contiguous = 1;
  for (int z = 0; z < n_zones; ++z)
    for (int c = 1; c < zone_sizes[z]; ++c)
      if (idx[zd[z].idx_off + c] != idx[zd[z].idx_off + c - 1] + 1) p->contiguous = 0;
  p->dma_ok = p->contiguous && (c_total % 4 == 0);
  for (int z = 0; z < n_zones; ++z)
    if (zone_sizes[z] % 4 != 0 || idx[zd[z].idx_off] % 4 != 0) p->dma_ok = 0;
  p->conv_zstride = (int64_t)(F / 4) * kTaps * GT * 64;
  hipError_t e = hipMalloc(&p->d_zones, sizeof(ZoneDesc) * n_zones);
  if (e == hipSuccess) e = hipMalloc(&p->d_idx, sizeof(int) * idx.size());
  if (e == hipSuccess) e = hipMemcpy(p->d_zones, zd.data(), sizeof(ZoneDesc) * n_zones, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(p->d_idx, idx.data(), sizeof(int) * idx.size(), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    set_error("isd_conv4_plan_create: %s", hipGetErrorString(e));
    isd_conv4_plan_destroy(p);
    return e == hipErrorNoDevice ? ISD_ERR_NO_DEVICE : ISD_ERR_HIP;
  }
  *out = p;
  return ISD_OK;
}

extern "C" int isd_conv4_plan_destroy(isd_conv4_plan* p) {
  if (!p) return ISD_OK;
  if (p->d_zones) (void)hipFree(p->d_zones);
  if (p->d_idx) (void)hipFree(p->d_idx);
  delete p;
  return ISD_OK;
}

extern "C" int isd_conv4_plan_set_activation_dtype(isd_conv4_plan* p, int dtype) {
  ISD_CHECK_ARG(p && (dtype == ISD_ACT_F32 || dtype == ISD_ACT_BF16), "isd_conv4_plan_set_activation_dtype: bad argument");
  p->act_bf16 = dtype == ISD_ACT_BF16;
  return ISD_OK;
}

extern "C" int64_t isd_conv4_param_count(const isd_conv4_plan* p) { return p ? p->n_params : ISD_ERR_INVALID; }

extern "C" int64_t isd_conv4_param_offset(const isd_conv4_plan* p, int zone, int which) {
  if (!p || zone < 0 || zone >= p->Z || which < 0 || which > 4 || (which > 2 && p->n_layers != 4)) return ISD_ERR_INVALID;
  const int64_t F = p->F;
  const int64_t offs[5] = {0, F * kTaps, F * kTaps + F, F * kTaps + F + F * F * p->cz[zone],
                           F * kTaps + F + F * F * p->cz[zone] + F * F * kTaps};
  return p->p_off[zone] + offs[which];
}

extern "C" int isd_conv4_windows(const isd_conv4_plan* p, int64_t T) {
  if (!p || T < p->W) return ISD_ERR_INVALID;
  return (int)((T - p->W) / p->S + 1);
}

namespace {
struct Geo {           // derived sizes for one call
  int N, T1, TT, IPW, RS_a, RS_b, lin0, CK;
  int64_t items, act;  // act = floats of one activation tensor
  // workspace layout (floats)
  int64_t o_eff, o_eff16, o_beff, o_w3, o_w3t, o_w4, o_w4t, o_a2, o_a3, o_a4, o_s, o_wg, o_wg34, o_part, total;
  int ipw0, ipw1, ns0, ns1, cw0, cw1, grp0, grp1;   // wgrad: items per wg, slabs (incl. wave groups), channels per wg
  int64_t slab0, slab1;
};

int row_stride_fwd(int need) {           // unpadded rows; B-fragment reads touch 4 rows x 16 consecutive t
  if (need <= 32) return need | 1;
  int rs = need;
  while (rs % 32 != 16) ++rs;
  return rs;
}

int make_geo(const isd_conv4_plan* p, int64_t B, int64_t T, Geo& g) {
  ISD_CHECK_ARG(T >= p->W, "conv4: T=%lld shorter than window_len=%d", (long long)T, p->W);
  g.N = (int)((T - p->W) / p->S + 1);
  g.T1 = p->W - (kTaps - 1);
  g.TT = (g.T1 + 15) / 16;
  g.IPW = g.TT >= 16 ? 1 : 16 / g.TT;
  g.items = B * g.N;
  g.act = g.items * p->Z * p->F * g.T1;
  // MODE 0 rows: one contiguous block per (item, chunk) when the zones are contiguous channel ranges and the
  // window is the whole row; otherwise gathered rows with a bank-friendly stride.  MODE 1 rows are always contiguous.
  g.lin0 = (p->contiguous && g.N == 1 && T == p->W) ? 1 : 0;
  g.RS_a = g.lin0 ? p->W : row_stride_fwd(p->W);
  g.RS_b = g.T1;
  {   // keep the staged tile (input rows + one weight chunk) inside 64 KiB of LDS; rows too long for that take the
      // large LDS allocation, and beyond ~1000 samples fewer channels per chunk
    const int GTf = p->F / 16;
    const int64_t row = g.RS_a > g.RS_b ? g.RS_a : g.RS_b;
    g.CK = kCK;
    int64_t fit = (64 * 1024 / 4 - (int64_t)(g.CK / 4) * kTaps * GTf * 64) / (g.CK * row);
    if (fit < 1) {
      while (g.CK > 4 && (150 * 1024 / 4 - 64 - (int64_t)(g.CK / 4) * kTaps * GTf * 64) / (g.CK * row) < 1) g.CK -= 4;
      fit = (150 * 1024 / 4 - 64 - (int64_t)(g.CK / 4) * kTaps * GTf * 64) / (g.CK * row);
    }
    ISD_CHECK_ARG(fit >= 1, "conv4: window_len=%d is too long for the LDS tile (about 9000 samples)", p->W);
    if (g.IPW > fit) g.IPW = (int)fit;
    // enough workgroups to co-schedule ~4 per CU (staging of one overlaps the MFMA phase of another)
    const int64_t occ = (g.items * p->Z) / 512;
    int min_ipw = (4 + g.TT - 1) / g.TT;                         // at least one column tile per wave
    if (min_ipw < 1) min_ipw = 1;
    int64_t want = occ < min_ipw ? min_ipw : occ;
    if (g.IPW > want) g.IPW = (int)want;
  }
  int64_t o = 0;
  g.o_eff = o;  o += align_up(p->eff_size, 64);
  g.o_eff16 = o; o += align_up(p->eff_size / kTaps * 4, 64);      // bf16 tap-window fragments: 16 bytes per (cg, gt, lane)
  g.o_beff = o; o += align_up((int64_t)p->Z * p->F, 64);
  const int64_t cw = align_up(p->conv_zstride * p->Z, 64);
  g.o_w3 = o; o += cw; g.o_w3t = o; o += cw; g.o_w4 = o; o += cw; g.o_w4t = o; o += cw;
  g.o_a2 = o; o += align_up(g.act, 64);
  g.o_a3 = o; o += align_up(g.act, 64);
  g.o_a4 = o; o += align_up(g.act, 64);
  g.o_s = o;  o += align_up(g.act, 64);
  g.o_wg = o; o += align_up(p->wg_size, 64);
  g.slab1 = (int64_t)p->Z * p->F * p->F * kTaps;
  g.o_wg34 = o; o += align_up(g.slab1, 64);
  // wgrad slabs: ~1024 workgroups over (item ranges) x zones x channel groups
  const int GT = p->F / 16;
  auto plan_wg = [&](int cin_max, int& cw, int& ipw, int& ns, int& grp) {
    cw = cin_max >= 64 ? 64 : (int)align_up(cin_max, 16);
    grp = 4 / (cw / 16);                                    // wave groups of the kernel (cw is 16, 32 or 64)
    const int zg = (cin_max + cw - 1) / cw;
    int64_t want = 1024 / ((int64_t)p->Z * zg);
    if (want < 1) want = 1;
    const int64_t it = g.items > 0 ? g.items : 1;
    if (want > it) want = it;
    ipw = (int)cdiv(it, want);
    ns = (int)cdiv(it, ipw) * grp;
  };
  plan_wg(p->max_cz + 1, g.cw0, g.ipw0, g.ns0, g.grp0);
  plan_wg(p->F, g.cw1, g.ipw1, g.ns1, g.grp1);
  g.slab0 = p->wg_size;
  const int64_t pa = (int64_t)g.ns0 * g.slab0, pb = (p->n_layers == 4) ? (int64_t)g.ns1 * g.slab1 : 0;
  // fused backward (reference-native shape): one workgroup row of max(1, 256/Z) per zone, 4 + 4 + 8 slabs each
  const int64_t rz = 256 / p->Z > 1 ? 256 / p->Z : 1;
  const int64_t pf = (p->n_layers == 4) ? rz * (8 * g.slab1 + 8 * g.slab0) : 0;
  const int64_t pmax = pa > pb ? pa : pb;
  g.o_part = o; o += align_up(pmax > pf ? pmax : pf, 64);
  g.total = o;
  return ISD_OK;
}
}  // namespace

extern "C" int64_t isd_conv4_workspace_bytes(const isd_conv4_plan* p, int64_t B, int64_t T) {
  if (!p || B < 0) return ISD_ERR_INVALID;
  Geo g;
  if (make_geo(p, B, T, g)) return ISD_ERR_INVALID;
  return g.total * 4;
}

static int launch_conv(int mode, int bf16, const ConvArgs& a, int n_zones, hipStream_t st) {
  const int64_t blocks = cdiv(a.items, a.IPW);
  ISD_CHECK_ARG(blocks <= 0x7fffffffLL, "conv4: too many items");
  const int GT = a.F / 16;
  const size_t lds = sizeof(float) * (4 + (((size_t)a.IPW * a.CK * a.RS + 3) & ~(size_t)3) + (size_t)(a.CK / 4) * kTaps * GT * 64 + 32);
  ISD_CHECK_ARG(lds <= 150 * 1024, "conv4: LDS tile of %zu bytes exceeds 150 KiB (window too long)", lds);
  const dim3 grid((unsigned)blocks, n_zones);
  const int tiles = a.IPW * a.TT;                     // column tiles per workgroup -> tiles per wave (1, 2 or 4)
  const int NT = tiles > 8 ? 4 : tiles > 4 ? 2 : 1;
#define ISD_CONV_LAUNCH_1(M, T, G, N)                                                                \
  do {                                                                                               \
    if (lds > 48 * 1024)                                                                             \
      ISD_HIP_TRY(hipFuncSetAttribute((const void*)conv5_fwd_kernel<M, T, G, N>,                     \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));        \
    hipLaunchKernelGGL((conv5_fwd_kernel<M, T, G, N>), grid, dim3(256), lds, st, a);                 \
  } while (0)
#define ISD_CONV_LAUNCH_N(M, T, G)                                                                   \
  do {                                                                                               \
    if (NT == 4) ISD_CONV_LAUNCH_1(M, T, G, 4);                                                      \
    else if (NT == 2) ISD_CONV_LAUNCH_1(M, T, G, 2);                                                 \
    else ISD_CONV_LAUNCH_1(M, T, G, 1);                                                              \
  } while (0)
#define ISD_CONV_LAUNCH(M, T)                                                                         \
  do {                                                                                               \
    if (GT == 2) ISD_CONV_LAUNCH_N(M, T, 2);                                                         \
    else ISD_CONV_LAUNCH_N(M, T, 1);                                                                 \
  } while (0)
  if (mode == 0 && !bf16) ISD_CONV_LAUNCH(0, float);
  else if (mode == 0) ISD_CONV_LAUNCH(0, bf16_t);
  else if (!bf16) ISD_CONV_LAUNCH(1, float);
  else ISD_CONV_LAUNCH(1, bf16_t);
#undef ISD_CONV_LAUNCH
#undef ISD_CONV_LAUNCH_N
#undef ISD_CONV_LAUNCH_1
  ISD_LAUNCH_CHECK();
  return ISD_OK;
}

// the reference-native shape with bf16 activations runs the bf16 fused pair (conv4_fused_*_bf16_kernel)
static bool fused16_ok(const isd_conv4_plan* p, const Geo& g) {
  return p->n_layers == 4 && p->F == 32 && p->act_bf16 && p->max_cz <= 16 && g.TT <= 16 && g.TT >= 4;
}
static size_t fused16_lds(const isd_conv4_plan* p, const Geo& g, bool bwd) {
  (void)p;
  const size_t tile_f = (size_t)fused16_rows(g.TT) * 16;
  const size_t w16 = (size_t)kTaps * 2 * 64 * 4;
  // backward: two sets of {GELU'(A4) / G4 / G2, A3, A2} + G3 + the bf16 x tile + two weight fragment sets
  // forward: t2, t3 (its front doubles as the bf16 x tile), row sums, bias, three weight fragment sets
  return sizeof(float) * (bwd ? 7 * tile_f + (size_t)(32 * ((g.TT + 1) / 2) + 16) * 8 + 2 * w16 + 16
                              : 2 * tile_f + 8 * 32 + 32 + 3 * 2 * 64 * 4 + 2 * w16 + 16);
}

static int launch_prep(const isd_conv4_plan* p, const Geo& g, const float* params, float* ws, hipStream_t st,
                       bool tap16 = false) {
  const int F = p->F;
  const int nbw = (int)cdiv(((int64_t)(p->max_cz + 3) / 4) * 4 * F, 256);
  // K = 32 bf16 fragments of cnn3 / cnn4: the raw-EEG fused pair and (tap16) the feature classifier's bf16 tail
  PrepConvArgs pc{ws + g.o_w3, ws + g.o_w3t, ws + g.o_w4, ws + g.o_w4t, p->conv_zstride, p->n_layers,
                  (tap16 && F == 32) || (fused16_ok(p, g) && fused16_lds(p, g, true) <= 160 * 1024) ? 1 : 0};
  const int extra = p->n_layers == 4 ? 8 : 0;
  hipLaunchKernelGGL(prep_fused_kernel, dim3(nbw + F + extra, p->Z), dim3(256), 0, st, params, p->d_zones,
                     ws + g.o_eff, ws + g.o_beff, F, nbw, p->act_bf16, pc,
                     tap16 ? reinterpret_cast<uint4*>(ws + g.o_eff16) : (uint4*)nullptr);
  ISD_LAUNCH_CHECK();
  return ISD_OK;
}

// BASELINE config 3 on the bf16 matrix cores: the classifier step (isd_featcnn_step) qualifies when the first layer is
// one column tile of at most 13 output steps over whole, contiguous, 16-byte aligned rows (the spec-S feature map:
// 17 frames) with 32 filters
static bool bf16_mfma_ok(const isd_conv4_plan* p, const Geo& g, const void* x) {
  return p->act_bf16 && p->F == 32 && g.lin0 && p->dma_ok && g.TT == 1 && g.T1 <= 13 && p->W <= 17 &&
         ((uintptr_t)x & 15) == 0;
}

// cnn1 o cnn2 forward into the A2 buffer; `a` comes back filled with the shared fields for the later layers
static int first_layer_forward(const isd_conv4_plan* p, const Geo& g, const float* x, int64_t T, float* ws,
                               hipStream_t st, ConvArgs& a, bool tap16 = false, bool in16 = false) {
  const int F = p->F;
  int rc;
  a = ConvArgs{};
  a.zones = p->d_zones; a.chan_idx = p->d_idx; a.items = g.items; a.Z = p->Z; a.F = F;
  a.TT = g.TT; a.IPW = g.IPW; a.Tout = g.T1; a.CK = g.CK;
  a.Ctot = p->Ctot; a.Tx = (int)T; a.N = g.N; a.S = p->S;
  // cnn1 o cnn2
  a.in = x; a.out = ws + g.o_a2; a.wfrag = ws + g.o_eff; a.bias = ws + g.o_beff; a.Tin = p->W; a.pad = 0; a.RS = g.RS_a;
  a.lin = g.lin0;
  if (tap16) {
    // bf16 matrix cores: 4 items per wave (NT), 16 per workgroup; fewer when the batch is small
    int NT = 4;
    int ipw = 16;
    const int64_t per_cu = cdiv(g.items * p->Z, 512);
    if (per_cu < ipw) { ipw = per_cu < 4 ? 4 : (int)per_cu; }
    NT = (ipw + 3) / 4;
    if (NT == 3) { NT = 4; }
    ipw = NT * 4;
    const size_t buf = (size_t)(((ipw * kCK * p->W + 3) & ~3) + (kCK / 4) * 2 * 64 * 4 + 32);
    const size_t lds = sizeof(float) * (4 + 2 * buf);
    a.IPW = ipw; a.RS = p->W;
    const dim3 grid((unsigned)cdiv(g.items, ipw), p->Z);
    const uint4* w16 = reinterpret_cast<const uint4*>(ws + g.o_eff16);
#define ISD_BF_LAUNCH(N, I16)                                                                                   \
  do {                                                                                                          \
    ISD_HIP_TRY(hipFuncSetAttribute((const void*)conv5_fwd_bf16_kernel<2, N, I16>,                              \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                     \
    hipLaunchKernelGGL((conv5_fwd_bf16_kernel<2, N, I16>), grid, dim3(256), lds, st, a, w16);                   \
  } while (0)
    if (in16) { if (NT == 4) ISD_BF_LAUNCH(4, true); else if (NT == 2) ISD_BF_LAUNCH(2, true); else ISD_BF_LAUNCH(1, true); }
    else { if (NT == 4) ISD_BF_LAUNCH(4, false); else if (NT == 2) ISD_BF_LAUNCH(2, false); else ISD_BF_LAUNCH(1, false); }
#undef ISD_BF_LAUNCH
    ISD_LAUNCH_CHECK();
    a.IPW = g.IPW;
    return ISD_OK;
  }
  {
    // wide inputs (>= 2 channel chunks): double-buffered LDS-DMA variant.  Two workgroups of 8 column tiles per CU
    // when there is enough work (the barrier / DMA wait of one runs under the MFMAs of the other), else one of 16.
    const int GT = F / 16;
    int NT = 2;
    const bool pack = g.TT == 1;                       // one-tile rows: the items' output steps share column tiles
    int ipw = pack ? (4 * NT * 16) / g.T1 : (4 * NT) / g.TT;
    // (two workgroups per CU only when there are more workgroups than CUs.  At 4096 items of 13 steps the packed tiles
    // are 3328 instead of 4096, but both are four per SIMD: the forward pass gains only beyond that batch)
    if (ipw < 1 || cdiv(g.items * p->Z, ipw) <= 256) {
      NT = 4;
      ipw = pack ? (16 * 16) / g.T1 : 16 / g.TT;
      const int64_t per_cu = cdiv(g.items * p->Z, 256);
      if (ipw > per_cu) ipw = (int)per_cu;
    }
    const size_t buf = (size_t)(((ipw * kCK * p->W + 3) & ~3) + (kCK / 4) * kTaps * GT * 64 + 32);
    const size_t lds = sizeof(float) * (4 + 2 * buf);
    if (g.lin0 && p->dma_ok && !p->act_bf16 && p->max_cz > kCK && g.TT <= 4 * NT && ipw >= 1 && lds <= 150 * 1024 &&
        ((uintptr_t)x & 15) == 0) {
      a.IPW = ipw; a.RS = p->W;
      const dim3 grid((unsigned)cdiv(g.items, ipw), p->Z);
#define ISD_GLDS_LAUNCH(G, N)                                                                                  \
  do {                                                                                                        \
    ISD_HIP_TRY(hipFuncSetAttribute((const void*)conv5_fwd_glds_kernel<G, N>,                                 \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                   \
    hipLaunchKernelGGL((conv5_fwd_glds_kernel<G, N>), grid, dim3(256), lds, st, a);                           \
  } while (0)
      if (GT == 2 && NT == 2) ISD_GLDS_LAUNCH(2, 2);
      else if (GT == 2) ISD_GLDS_LAUNCH(2, 4);
      else if (NT == 2) ISD_GLDS_LAUNCH(1, 2);
      else ISD_GLDS_LAUNCH(1, 4);
#undef ISD_GLDS_LAUNCH
      ISD_LAUNCH_CHECK();
      a.IPW = g.IPW;
      rc = ISD_OK;
    } else {
      rc = launch_conv(0, p->act_bf16, a, p->Z, st);
    }
  }
  return rc;
}

// LDS bytes of conv4_fused_bwd_kernel<8>; the forward keeps GELU'(A4) instead of A4 exactly when this fits
static size_t fused_bwd_lds(const isd_conv4_plan* p, const Geo& g) {
  const int tile = ((p->F * g.T1 + 3) & ~3) + 4;
  return sizeof(float) * (size_t)(((4 + 16 * p->W + 3) & ~3) + 4 + 3 * tile + 2 * 8 * kTaps * 2 * 64 + 32 + 16);
}

extern "C" int isd_conv4_forward(const isd_conv4_plan* p, const float* x, const float* params, float* feat,
                                 void* workspace, int64_t B, int64_t T, void* stream) {
  ISD_CHECK_ARG(p, "isd_conv4_forward: null plan");
  ISD_CHECK_ARG(B >= 0, "isd_conv4_forward: B=%lld", (long long)B);
  Geo g;
  int rc = make_geo(p, B, T, g);
  if (rc) return rc;
  if (B == 0) return ISD_OK;
  ISD_CHECK_ARG(x && params && feat && workspace, "isd_conv4_forward: null argument");
  hipStream_t st = (hipStream_t)stream;
  float* ws = (float*)workspace;
  const int F = p->F;
  rc = launch_prep(p, g, params, ws, st);
  if (rc) return rc;
  if (fused16_ok(p, g) && fused16_lds(p, g, true) <= 160 * 1024) {
    // reference-native shape, bf16 activations: the fused kernel on the bf16 matrix cores
    FusedFwdArgs fa = {};
    fa.x = x; fa.weff = ws + g.o_eff; fa.beff = ws + g.o_beff; fa.w3 = ws + g.o_w3; fa.w4 = ws + g.o_w4;
    fa.a2 = ws + g.o_a2; fa.a3 = ws + g.o_a3; fa.a4 = ws + g.o_a4; fa.feat = feat;
    fa.zones = p->d_zones; fa.chan_idx = p->d_idx; fa.wz_stride = p->conv_zstride; fa.items = g.items;
    fa.Z = p->Z; fa.W = p->W; fa.T1 = g.T1; fa.TT = g.TT; fa.store = 2;
    fa.Ctot = p->Ctot; fa.Tx = (int)T; fa.N = g.N; fa.S = p->S;
    const size_t lds = fused16_lds(p, g, false);
    int per_zone = (lds <= 80 * 1024 ? 512 : 256) / p->Z;     // two workgroups per CU when the tiles allow
    if (per_zone < 1) per_zone = 1;
    if (per_zone > g.items) per_zone = (int)g.items;
    ISD_HIP_TRY(hipFuncSetAttribute((const void*)conv4_fused_fwd_bf16_kernel<8>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((conv4_fused_fwd_bf16_kernel<8>), dim3(per_zone, p->Z), dim3(8 * 64), lds, st, fa);
    ISD_LAUNCH_CHECK();
    return ISD_OK;
  }
  if (p->n_layers == 4 && F == 32 && !p->act_bf16 && p->max_cz <= 16 && g.TT <= 16 && (F * g.T1) % 4 == 0 &&
      g.TT >= 4) {
    // reference-native shape: one persistent fused kernel (register-resident weights, activations through LDS)
    FusedFwdArgs fa = {};
    fa.x = x; fa.weff = ws + g.o_eff; fa.beff = ws + g.o_beff; fa.w3 = ws + g.o_w3; fa.w4 = ws + g.o_w4;
    fa.a2 = ws + g.o_a2; fa.a3 = ws + g.o_a3; fa.a4 = ws + g.o_a4; fa.feat = feat;
    fa.zones = p->d_zones; fa.chan_idx = p->d_idx; fa.wz_stride = p->conv_zstride; fa.items = g.items;
    fa.Z = p->Z; fa.W = p->W; fa.T1 = g.T1; fa.TT = g.TT; fa.store = fused_bwd_lds(p, g) <= 160 * 1024 ? 2 : 1;
    fa.Ctot = p->Ctot; fa.Tx = (int)T; fa.N = g.N; fa.S = p->S;
    constexpr int NW = 8;   // measured after the prefetch restructure (B=4096, T=512): 8 waves 25.6 ms/step, 16 waves 25.6, 4 waves 28.3
    const size_t lds = sizeof(float) * (size_t)(4 + ((16 * p->W + 3) & ~3) + 2 * ((F * g.T1 + 3) & ~3) + NW * F +
                                                 (4 + 8 + 8) * kTaps * 2 * 64 + 64);
    if (lds <= 150 * 1024) {
      int per_zone = 256 / p->Z;
      if (per_zone < 1) per_zone = 1;
      if (per_zone > g.items) per_zone = (int)g.items;
      ISD_HIP_TRY(hipFuncSetAttribute((const void*)conv4_fused_fwd_kernel<NW>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL((conv4_fused_fwd_kernel<NW>), dim3(per_zone, p->Z), dim3(NW * 64), lds, st, fa);
      ISD_LAUNCH_CHECK();
      return ISD_OK;
    }
  }
  ConvArgs a = {};
  rc = first_layer_forward(p, g, x, T, ws, st, a);
  if (rc) return rc;
  const float* last = ws + g.o_a2;
  if (p->n_layers == 4) {
    a.bias = nullptr; a.Tin = g.T1; a.pad = 2; a.RS = g.RS_b; a.wz_stride = p->conv_zstride; a.lin = 1;
    a.in = ws + g.o_a2; a.out = ws + g.o_a3; a.wfrag = ws + g.o_w3;
    rc = launch_conv(1, p->act_bf16, a, p->Z, st);
    if (rc) return rc;
    a.in = ws + g.o_a3; a.out = ws + g.o_a4; a.wfrag = ws + g.o_w4;
    rc = launch_conv(1, p->act_bf16, a, p->Z, st);
    if (rc) return rc;
    last = ws + g.o_a4;
  }
  const int64_t rows = g.items * p->Z * F;
  if (p->act_bf16)
    hipLaunchKernelGGL((gelu_mean_fwd_kernel<bf16_t>), dim3((unsigned)cdiv(rows * 16, 256)), dim3(256), 0, st, last,
                       feat, rows, g.T1);
  else
    hipLaunchKernelGGL((gelu_mean_fwd_kernel<float>), dim3((unsigned)cdiv(rows * 16, 256)), dim3(256), 0, st, last, feat,
                       rows, g.T1);
  ISD_LAUNCH_CHECK();
  return ISD_OK;
}

template <int MODE, typename AT, bool BOTH>
static int launch_wgrad_b(const WgradArgs& a, dim3 grid, size_t lds, hipStream_t st) {
  if (lds > 48 * 1024)
    ISD_HIP_TRY(hipFuncSetAttribute((const void*)conv5_wgrad_kernel<MODE, AT, BOTH>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL((conv5_wgrad_kernel<MODE, AT, BOTH>), grid, dim3(256), lds, st, a);
  ISD_LAUNCH_CHECK();
  return ISD_OK;
}
template <int MODE, typename AT>
static int launch_wgrad_t(const WgradArgs& a, dim3 grid, size_t lds, hipStream_t st) {
  return a.F == 32 ? launch_wgrad_b<MODE, AT, true>(a, grid, lds, st) : launch_wgrad_b<MODE, AT, false>(a, grid, lds, st);
}

static int launch_wgrad(int mode, int bf16, WgradArgs a, int n_zones, int cin_max, hipStream_t st) {   // `a` by value: the stage geometry chosen here must not leak into the caller's next launch
  int64_t per_item = (int64_t)a.F * a.RSo + (int64_t)a.CW * a.RSi;
  int ips = (int)((96 * 1024 / 4 - 4) / per_item);
  if (ips < 1) ips = (int)((150 * 1024 / 4 - 4) / per_item);   // long windows: one item in the large LDS allocation
  a.seg_len = 0;
  if (ips < 1) {
    // longer still: time segments (the reduction dimension of the weight gradient), one item per stage
    int seg = (int)((150 * 1024 / 4 - 64 - (int64_t)a.CW * (kTaps - 1)) / (a.F + a.CW)) & ~3;
    ISD_CHECK_ARG(seg >= 16, "conv4 wgrad: no LDS segment fits");
    a.seg_len = seg;
    a.RSo = seg;
    a.RSi = seg + kTaps - 1;
    a.lin = 0;
    per_item = (int64_t)a.F * a.RSo + (int64_t)a.CW * a.RSi;
    ips = 1;
  }
  if (ips > 1 && per_item * 4 > 32 * 1024) ips = 1;           // big items: one per stage, 2-3 workgroups per CU
  const int ips_occ = (int)((32 * 1024 / 4) / per_item);      // prefer <= 32 KiB so several workgroups share a CU
  if (ips_occ >= 1 && ips > ips_occ) ips = ips_occ;
  if (ips > 16) ips = 16;
  if (ips > a.items_per_wg) ips = a.items_per_wg;
  a.IPS = ips;
  // (two LDS stages filled by LDS-DMA were measured here: 28.9 vs 28.4 ms/step on FAST B=4096 -- one stage and two
  // co-resident workgroups per CU hide the copy better than one workgroup with two stages)
  const size_t lds = sizeof(float) * ((size_t)ips * per_item + 4);
  const int64_t wgs = cdiv(a.items, a.items_per_wg);
  const int zgroups = (cin_max + a.CW - 1) / a.CW;
  const dim3 grid((unsigned)wgs, n_zones, zgroups);
  if (mode == 0) return bf16 ? launch_wgrad_t<0, bf16_t>(a, grid, lds, st) : launch_wgrad_t<0, float>(a, grid, lds, st);
  return bf16 ? launch_wgrad_t<1, bf16_t>(a, grid, lds, st) : launch_wgrad_t<1, float>(a, grid, lds, st);
}

// cnn3 / cnn4 gradient in natural [F][F][5] layout -> flat gradient block
__global__ void scatter_conv_grad_kernel(const float* __restrict__ wg, const isd::ZoneDesc* __restrict__ zones,
                                         float* __restrict__ dparams, int F, int layer) {
  const int z = blockIdx.y;
  const isd::ZoneDesc zd = zones[z];
  const int n = F * F * isd::kTaps;
  float* dst = dparams + zd.p_off + F * isd::kTaps + F + (int64_t)F * F * zd.cin + (int64_t)layer * n;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) dst[e] = wg[(int64_t)z * n + e];
}

// the chain dWeff -> (dW1, db1, dW2): 1024-thread workgroups for wide inputs, 256 for the zone shapes
static inline void launch_fused_bwd(const isd_conv4_plan* p, const float* params, const float* dweff, float* dparams,
                                    hipStream_t st) {
  const int F = p->F;
  if (p->max_cz >= 64) {
    const int nb2 = (int)cdiv((int64_t)F * F * p->max_cz, 1024);
    hipLaunchKernelGGL(fused_bwd_kernel<1024>, dim3(nb2 + F * kTaps + F, p->Z), dim3(1024), 0, st, params, p->d_zones,
                       dweff, dparams, F, nb2);
  } else {
    const int nb2 = (int)cdiv((int64_t)F * F * p->max_cz, 256);
    hipLaunchKernelGGL(fused_bwd_kernel<256>, dim3(nb2 + F * kTaps + F, p->Z), dim3(256), 0, st, params, p->d_zones,
                       dweff, dparams, F, nb2);
  }
}

// cnn1 o cnn2 backward: dWeff (+ dbeff in the ones channel) from g2 = dL/dA2, then the chain to W1, b1, W2
static int first_layer_backward(const isd_conv4_plan* p, const Geo& g, const float* x, int64_t T, const float* params,
                                const float* g2, float* dparams, float* ws, hipStream_t st, bool tap16 = false,
                                bool in16 = false) {
  const int F = p->F;
  int rc;
  WgradArgs w = {};
  w.zones = p->d_zones; w.chan_idx = p->d_idx; w.items = g.items;
  w.Z = p->Z; w.F = F; w.Tout = g.T1; w.part = ws + g.o_part;
  w.Ctot = p->Ctot; w.Tx = (int)T; w.N = g.N; w.S = p->S;
  w.dout = g2; w.in = x; w.Tin = p->W; w.pad = 0;
  w.RSo = g.T1; w.lin = g.lin0;
  w.RSi = g.lin0 ? p->W : (p->W | 1);
  w.slab_size = g.slab0; w.wz_stride = 0; w.items_per_wg = g.ipw0; w.CW = g.cw0;
  int n_slabs0 = g.ns0;
  {
    // wide inputs: LDS-DMA double-buffered variant, ~3 workgroups per CU
    const int zg = (p->max_cz + 63) / 64;
    int64_t r_target = (256 * 3) / ((int64_t)zg * p->Z);
    if (r_target < 1) r_target = 1;
    if (r_target > g.ns0) r_target = g.ns0;
    const int ipw = (int)cdiv(g.items, r_target);
    const int R = (int)cdiv(g.items, ipw);
    const int item_len = (F * g.T1 + 64 * p->W + 3) & ~3;
    int ips = 4;
    while (ips > 1 && (size_t)(2 * (ips * item_len + 32) + 8) * 4 > 48 * 1024) --ips;
    const size_t lds = sizeof(float) * (size_t)(2 * (ips * item_len + 32) + 8);
    if (tap16) {
      // bf16 matrix cores: gradient images of 16 x 32 bf16 (1 KiB) + 64 input rows per item, item pairs per K step
      const int item16 = (16 * F / 2 + 64 * p->W + 3) & ~3;
      const int ips16 = 4;
      const size_t lds16 = sizeof(float) * (size_t)(2 * (ips16 * item16 + 32) + 8);
      w.items_per_wg = ipw; w.IPS = ips16;
      const dim3 grid((unsigned)R, p->Z, zg);
      if (in16) hipLaunchKernelGGL((conv5_wgrad_wide_bf16_kernel<2, true>), grid, dim3(256), lds16, st, w);
      else hipLaunchKernelGGL((conv5_wgrad_wide_bf16_kernel<2, false>), grid, dim3(256), lds16, st, w);
      ISD_LAUNCH_CHECK();
      n_slabs0 = R;
      rc = ISD_OK;
    } else if (g.lin0 && p->dma_ok && !p->act_bf16 && p->max_cz >= 64 && (F * g.T1) % 4 == 0 && lds <= 64 * 1024 &&
        ((uintptr_t)x & 15) == 0 && ((uintptr_t)g2 & 15) == 0) {
      w.items_per_wg = ipw; w.IPS = ips;
      const dim3 grid((unsigned)R, p->Z, zg);
      if (F == 32) hipLaunchKernelGGL(conv5_wgrad_wide_kernel<2>, grid, dim3(256), lds, st, w);
      else hipLaunchKernelGGL(conv5_wgrad_wide_kernel<1>, grid, dim3(256), lds, st, w);
      ISD_LAUNCH_CHECK();
      n_slabs0 = R;
      rc = ISD_OK;
    } else {
      rc = launch_wgrad(0, p->act_bf16, w, p->Z, p->max_cz + 1, st);
    }
  }
  if (rc) return rc;
  launch_reduce_slabs(ws + g.o_part, ws + g.o_wg, g.slab0, n_slabs0, st);
  {
    launch_fused_bwd(p, params, ws + g.o_wg, dparams, st);
  }
  ISD_LAUNCH_CHECK();
  return ISD_OK;
}

// ---------------------------------------------------------------------------------------
// One call for the whole classifier step on spec-S features: conv stack -> Linear -> softmax-CE forward and
// backward (featcnn_tail_kernel between the two first-layer kernels).
// ---------------------------------------------------------------------------------------
static bool featcnn_ok(const isd_conv4_plan* p, const Geo& g, int n_cls, const void* x = nullptr) {
  return p->Z == 1 && p->n_layers == 4 && p->F == 32 && (!p->act_bf16 || bf16_mfma_ok(p, g, x)) && g.TT == 1 &&
         g.N == 1 && n_cls >= 1 && n_cls <= kTailMaxCls;
}

extern "C" int isd_featcnn_supported(const isd_conv4_plan* p, int64_t B, int64_t T, int n_cls) {
  if (!p || B < 1) return 0;
  Geo g;
  if (make_geo(p, B, T, g)) return 0;
  return featcnn_ok(p, g, n_cls) ? 1 : 0;       // (bf16: additionally x must be 16-byte aligned; isd_featcnn_step checks)
}

static int featcnn_step_impl(const isd_conv4_plan* p, const float* x, const float* params, const float* fc_w,
                             const float* fc_b, const void* labels, int label_bytes, float* dparams, float* dfc,
                             float* logits, int64_t* pred, float* loss, void* workspace, int64_t B, int64_t T,
                             int n_cls, float grad_scale, void* stream, bool x_bf16) {
  ISD_CHECK_ARG(p, "isd_featcnn_step: null plan");
  ISD_CHECK_ARG(B >= 1, "isd_featcnn_step: B=%lld", (long long)B);
  Geo g;
  int rc = make_geo(p, B, T, g);
  if (rc) return rc;
  if (!featcnn_ok(p, g, n_cls, x)) {
    set_error("isd_featcnn_step: needs one zone, 4 layers, 32 filters, <= 16 output steps (bf16 activations: <= 13, over "
              "whole 16-byte aligned rows of at most 17 samples), <= %d classes", kTailMaxCls);
    return ISD_ERR_UNSUPPORTED;
  }
  const bool tap16 = p->act_bf16 != 0;
  if (x_bf16 && !(tap16 && p->max_cz % 8 == 0)) {
    set_error("isd_featcnn_step_bf16: a bf16 feature map needs a plan with bf16 activations and a multiple of 8 input "
              "channels (got act_bf16=%d, %d channels)", p->act_bf16, p->max_cz);
    return ISD_ERR_UNSUPPORTED;
  }
  ISD_CHECK_ARG(x && params && fc_w && fc_b && logits && pred && workspace, "isd_featcnn_step: null argument");
  ISD_CHECK_ARG(!labels || label_bytes == 1 || label_bytes == 8, "isd_featcnn_step: labels must be uint8 or int64");
  const bool train = labels && dparams && dfc;
  ISD_CHECK_ARG(!labels || loss, "isd_featcnn_step: labels without a loss pointer");
  hipStream_t st = (hipStream_t)stream;
  float* ws = (float*)workspace;
  const int F = p->F;
  rc = launch_prep(p, g, params, ws, st, tap16);
  if (rc) return rc;
  ConvArgs a = {};
  rc = first_layer_forward(p, g, x, T, ws, st, a, tap16, x_bf16);
  if (rc) return rc;
  TailArgs t = {};
  t.a2 = ws + g.o_a2; t.g2 = ws + g.o_a4;
  t.w3 = ws + g.o_w3; t.w4 = ws + g.o_w4; t.w3t = ws + g.o_w3t; t.w4t = ws + g.o_w4t;
  t.fc_w = fc_w; t.fc_b = fc_b; t.labels = labels; t.label_bytes = label_bytes;
  t.logits = logits; t.pred = pred; t.part = ws + g.o_part;
  t.items = g.items; t.T1 = g.T1; t.n_cls = n_cls; t.train = train ? 1 : 0; t.grad_scale = grad_scale;
  const int n34 = 2 * F * F * kTaps, nfc = n_cls * (F + 1);
  t.slab = n34 + nfc + 1;
  constexpr int TNW = 4;                                          // waves per workgroup (measured: 4 -> 107 us; 8 -> 135 us, 256 VGPRs + 133 spills)
  int blocks = (int)cdiv(g.items, TNW * kTailNI * 2);             // two rounds of NI items per wave
  if (blocks > 256) blocks = 256;
  if (blocks < 1) blocks = 1;
  const int tile = (F * g.T1 + 3) & ~3;
  size_t lds = sizeof(float) * (size_t)(4 * 8 * kTaps * 2 * 64 + kTailMaxCls * (F + 1) + TNW * kTailNI * (4 * tile + 64) + 16);
  if (lds < sizeof(float) * 2 * kTailCombCopy) lds = sizeof(float) * 2 * kTailCombCopy;           // two accumulator copies (epilogue)
  ISD_CHECK_ARG(lds <= 160 * 1024 && (int64_t)blocks * t.slab <= g.total - g.o_part, "isd_featcnn_step: workspace");
  if (tap16) {
    // bf16 matrix cores: eight waves per workgroup, an item per wave and round
    constexpr int BNW = 8;
    blocks = (int)cdiv(g.items, BNW * 2);
    if (blocks > 256) blocks = 256;
    if (blocks < 1) blocks = 1;
    size_t lds16 = (size_t)4 * kTaps * 2 * 64 * 16 + sizeof(float) * kTailMaxCls * (F + 1) + (size_t)BNW * (4 * 40 * 64 + 256) + 64;
    if (lds16 < sizeof(float) * 3 * kTailCombCopy) lds16 = sizeof(float) * 3 * kTailCombCopy;     // three accumulator copies (epilogue)
    ISD_CHECK_ARG(lds16 <= 160 * 1024 && (int64_t)blocks * t.slab <= g.total - g.o_part, "isd_featcnn_step: workspace");
    ISD_HIP_TRY(hipFuncSetAttribute((const void*)featcnn_tail_bf16_kernel<BNW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16));
    hipLaunchKernelGGL((featcnn_tail_bf16_kernel<BNW>), dim3(blocks), dim3(BNW * 64), lds16, st, t);
  } else {
    ISD_HIP_TRY(hipFuncSetAttribute((const void*)featcnn_tail_kernel<TNW, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((featcnn_tail_kernel<TNW, false>), dim3(blocks), dim3(TNW * 64), lds, st, t);
  }
  ISD_LAUNCH_CHECK();
  if (!labels) return ISD_OK;
  if (train) {
    // the slab sums go straight to their places: cnn3 / cnn4 gradients (back to back in the flat block), the FC
    // gradients, the loss
    float* dw34 = dparams + p->p_off[0] + F * kTaps + F + (int64_t)F * F * p->cz[0];
    launch_reduce_slabs(ws + g.o_part, ReduceDst{dw34, dfc, loss, n34, nfc}, t.slab, blocks, st);
  } else {
    // evaluation with labels: only the loss (last element of the slab) is wanted; the rest goes to scratch
    float* red = ws + g.o_s;                                      // free activation-sized scratch
    launch_reduce_slabs(ws + g.o_part, ReduceDst{red, red + n34, loss, n34, nfc}, t.slab, blocks, st);
  }
  ISD_LAUNCH_CHECK();
  if (!train) return ISD_OK;
  return first_layer_backward(p, g, x, T, params, ws + g.o_a4, dparams, ws, st, tap16, x_bf16);
}

extern "C" int isd_featcnn_step(const isd_conv4_plan* p, const float* x, const float* params, const float* fc_w,
                                const float* fc_b, const void* labels, int label_bytes, float* dparams, float* dfc,
                                float* logits, int64_t* pred, float* loss, void* workspace, int64_t B, int64_t T,
                                int n_cls, float grad_scale, void* stream) {
  return featcnn_step_impl(p, x, params, fc_w, fc_b, labels, label_bytes, dparams, dfc, logits, pred, loss, workspace, B,
                           T, n_cls, grad_scale, stream, false);
}

// The same step on a bf16 feature map x [B][C][T] (isd_features_fused_bf16): the plan must have bf16 activations
extern "C" int isd_featcnn_step_bf16(const isd_conv4_plan* p, const uint16_t* x, const float* params, const float* fc_w,
                                     const float* fc_b, const void* labels, int label_bytes, float* dparams, float* dfc,
                                     float* logits, int64_t* pred, float* loss, void* workspace, int64_t B, int64_t T,
                                     int n_cls, float grad_scale, void* stream) {
  return featcnn_step_impl(p, reinterpret_cast<const float*>(x), params, fc_w, fc_b, labels, label_bytes, dparams, dfc,
                           logits, pred, loss, workspace, B, T, n_cls, grad_scale, stream, true);
}

static int conv4_backward_impl(const isd_conv4_plan* p, const float* x, const float* params, const float* dfeat,
                               float* dparams, float* dx, void* workspace, int64_t B, int64_t T, void* stream) {
  ISD_CHECK_ARG(p, "isd_conv4_backward: null plan");
  ISD_CHECK_ARG(B >= 0, "isd_conv4_backward: B=%lld", (long long)B);
  Geo g;
  int rc = make_geo(p, B, T, g);
  if (rc) return rc;
  ISD_CHECK_ARG(dparams, "isd_conv4_backward: null dparams");
  hipStream_t st = (hipStream_t)stream;
  if (B == 0) {
    ISD_HIP_TRY(hipMemsetAsync(dparams, 0, sizeof(float) * p->n_params, st));
    return ISD_OK;
  }
  ISD_CHECK_ARG(x && params && dfeat && workspace, "isd_conv4_backward: null argument");
  ISD_CHECK_ARG(!dx || !p->act_bf16, "isd_conv4_backward_x: fp32 activations only");
  float* ws = (float*)workspace;
  const int F = p->F;
  const int64_t rows = g.items * p->Z * F;
  bool kept_dgelu = false;                                 // the forward of this step ran fused and kept GELU'(A4)
  if (fused16_ok(p, g) && fused16_lds(p, g, true) <= 160 * 1024) {
    ISD_CHECK_ARG(!dx, "isd_conv4_backward_x: fp32 activations only");
    constexpr int NW = 8;
    const size_t lds = fused16_lds(p, g, true);
    int per_zone = 256 / p->Z;
    if (per_zone < 1) per_zone = 1;
    if (per_zone > g.items) per_zone = (int)g.items;
    FusedBwdArgs fb = {};
    fb.x = x; fb.dfeat = dfeat; fb.a2 = ws + g.o_a2; fb.a3 = ws + g.o_a3; fb.a4 = ws + g.o_a4;
    fb.w3t = ws + g.o_w3t; fb.w4t = ws + g.o_w4t;
    fb.part4 = ws + g.o_part;
    fb.part3 = fb.part4 + (int64_t)per_zone * 4 * g.slab1;
    fb.part0 = fb.part3 + (int64_t)per_zone * 4 * g.slab1;
    fb.zones = p->d_zones; fb.chan_idx = p->d_idx; fb.wz_stride = p->conv_zstride; fb.items = g.items;
    fb.slab1 = g.slab1; fb.slab0 = g.slab0;
    fb.Z = p->Z; fb.W = p->W; fb.T1 = g.T1; fb.TT = g.TT;
    fb.Ctot = p->Ctot; fb.Tx = (int)T; fb.N = g.N; fb.S = p->S;
    ISD_HIP_TRY(hipFuncSetAttribute((const void*)conv4_fused_bwd_bf16_kernel<NW>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((conv4_fused_bwd_bf16_kernel<NW>), dim3(per_zone, p->Z), dim3(NW * 64), lds, st, fb);
    ISD_LAUNCH_CHECK();
    rc = launch_reduce_fused_bwd(fb.part4, fb.part3, fb.part0, g.slab1, per_zone * 4, g.slab0, per_zone * NW, dparams,
                                 ws + g.o_wg, p->d_zones, F, st);
    if (rc) return rc;
    launch_fused_bwd(p, params, ws + g.o_wg, dparams, st);
    ISD_LAUNCH_CHECK();
    return ISD_OK;
  }
  if (p->n_layers == 4 && F == 32 && !p->act_bf16 && p->max_cz <= 16 && g.TT <= 16 && (F * g.T1) % 4 == 0 &&
      g.TT >= 4) {
    // reference-native shape: one persistent fused kernel; gradient tiles stay in LDS, weight gradients in registers
    constexpr int NW = 8;
    const size_t lds = fused_bwd_lds(p, g);
    const size_t lds_fwd = sizeof(float) * (size_t)(4 + ((16 * p->W + 3) & ~3) + 2 * ((F * g.T1 + 3) & ~3) + 8 * F +
                                                     (4 + 8 + 8) * kTaps * 2 * 64 + 64);
    kept_dgelu = lds <= 160 * 1024 && lds_fwd <= 150 * 1024;
    if (kept_dgelu && !dx) {                              // (the input gradient needs G2 in memory: layer-wise path)
      int per_zone = 256 / p->Z;
      if (per_zone < 1) per_zone = 1;
      if (per_zone > g.items) per_zone = (int)g.items;
      FusedBwdArgs fb = {};
      fb.x = x; fb.dfeat = dfeat; fb.a2 = ws + g.o_a2; fb.a3 = ws + g.o_a3; fb.a4 = ws + g.o_a4;
      fb.w3t = ws + g.o_w3t; fb.w4t = ws + g.o_w4t;
      fb.part4 = ws + g.o_part;
      fb.part3 = fb.part4 + (int64_t)per_zone * 4 * g.slab1;
      fb.part0 = fb.part3 + (int64_t)per_zone * 4 * g.slab1;
      fb.zones = p->d_zones; fb.chan_idx = p->d_idx; fb.wz_stride = p->conv_zstride; fb.items = g.items;
      fb.slab1 = g.slab1; fb.slab0 = g.slab0;
      fb.Z = p->Z; fb.W = p->W; fb.T1 = g.T1; fb.TT = g.TT;
      fb.Ctot = p->Ctot; fb.Tx = (int)T; fb.N = g.N; fb.S = p->S;
      ISD_HIP_TRY(hipFuncSetAttribute((const void*)conv4_fused_bwd_kernel<NW>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lds));
      hipLaunchKernelGGL((conv4_fused_bwd_kernel<NW>), dim3(per_zone, p->Z), dim3(NW * 64), lds, st, fb);
      ISD_LAUNCH_CHECK();
      // the fp32 kernel leaves two slabs per workgroup for cnn3 / cnn4 and four for Weff (the buffers are sized for the
      // bf16 twin's four and eight)
      rc = launch_reduce_fused_bwd(fb.part4, fb.part3, fb.part0, g.slab1, per_zone * 2, g.slab0, per_zone * 4, dparams,
                                   ws + g.o_wg, p->d_zones, F, st);
      if (rc) return rc;
      launch_fused_bwd(p, params, ws + g.o_wg, dparams, st);
      ISD_LAUNCH_CHECK();
      return ISD_OK;
    }
  }
  float* top = ws + (p->n_layers == 4 ? g.o_a4 : g.o_a2);        // activation that fed GELU
  if (p->act_bf16)
    hipLaunchKernelGGL((gelu_mean_bwd_kernel<bf16_t>), dim3((unsigned)cdiv(rows * 16, 256)), dim3(256), 0, st, top,
                       dfeat, rows, g.T1);
  else if (kept_dgelu)
    hipLaunchKernelGGL((gelu_mean_bwd_kernel<float, true>), dim3((unsigned)cdiv(rows * 16, 256)), dim3(256), 0, st, top,
                       dfeat, rows, g.T1);
  else
    hipLaunchKernelGGL((gelu_mean_bwd_kernel<float>), dim3((unsigned)cdiv(rows * 16, 256)), dim3(256), 0, st, top, dfeat,
                       rows, g.T1);
  ISD_LAUNCH_CHECK();
  WgradArgs w = {};
  w.zones = p->d_zones; w.chan_idx = p->d_idx; w.items = g.items;
  w.Z = p->Z; w.F = F; w.Tout = g.T1; w.part = ws + g.o_part;
  w.Ctot = p->Ctot; w.Tx = (int)T; w.N = g.N; w.S = p->S;
  ConvArgs a = {};
  a.zones = p->d_zones; a.chan_idx = p->d_idx; a.items = g.items; a.Z = p->Z; a.F = F;
  a.TT = g.TT; a.IPW = g.IPW; a.Tout = g.T1; a.Tin = g.T1; a.pad = 2; a.RS = g.RS_b; a.wz_stride = p->conv_zstride;
  a.CK = g.CK;
  a.lin = 1;
  a.Ctot = p->Ctot; a.Tx = (int)T; a.N = g.N; a.S = p->S;
  const float* g2 = top;                                          // gradient w.r.t. the cnn2 output
  if (p->n_layers == 4) {
    w.dout = ws + g.o_a4; w.in = ws + g.o_a3; w.Tin = g.T1; w.pad = 2; w.RSo = g.T1; w.RSi = g.T1; w.lin = 1;
    w.slab_size = g.slab1; w.wz_stride = (int64_t)F * F * kTaps; w.items_per_wg = g.ipw1; w.CW = g.cw1;
    rc = launch_wgrad(1, p->act_bf16, w, p->Z, F, st);
    if (rc) return rc;
    launch_reduce_slabs(ws + g.o_part, ws + g.o_wg34, g.slab1, g.ns1, st);
    hipLaunchKernelGGL(scatter_conv_grad_kernel, dim3(4, p->Z), dim3(256), 0, st, ws + g.o_wg34, p->d_zones, dparams, F, 1);
    ISD_LAUNCH_CHECK();
    a.in = ws + g.o_a4; a.out = ws + g.o_s; a.wfrag = ws + g.o_w4t;
    rc = launch_conv(1, p->act_bf16, a, p->Z, st);
    if (rc) return rc;
    // cnn3: dW3 = wgrad(G3, A2); G2 = dgrad(G3) (into the A4 buffer, free now)
    w.dout = ws + g.o_s; w.in = ws + g.o_a2;
    rc = launch_wgrad(1, p->act_bf16, w, p->Z, F, st);
    if (rc) return rc;
    launch_reduce_slabs(ws + g.o_part, ws + g.o_wg34, g.slab1, g.ns1, st);
    hipLaunchKernelGGL(scatter_conv_grad_kernel, dim3(4, p->Z), dim3(256), 0, st, ws + g.o_wg34, p->d_zones, dparams, F, 0);
    ISD_LAUNCH_CHECK();
    a.in = ws + g.o_s; a.out = ws + g.o_a4; a.wfrag = ws + g.o_w3t;
    rc = launch_conv(1, p->act_bf16, a, p->Z, st);
    if (rc) return rc;
    g2 = ws + g.o_a4;
  }
  rc = first_layer_backward(p, g, x, T, params, g2, dparams, ws, st);
  if (rc || !dx) return rc;
  const size_t lds = sizeof(float) * (size_t)F * g.T1;
  ISD_CHECK_ARG(lds <= 150 * 1024, "isd_conv4_backward_x: window_len=%d is too long for the gradient tile", p->W);
  ISD_CHECK_ARG(g.items <= 0x7fffffffLL, "isd_conv4_backward_x: too many items");
  ISD_HIP_TRY(hipMemsetAsync(dx, 0, sizeof(float) * (size_t)B * p->Ctot * T, st));
  if (lds > 48 * 1024)
    ISD_HIP_TRY(hipFuncSetAttribute((const void*)conv5_dx_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(conv5_dx_kernel, dim3((unsigned)g.items, p->Z), dim3(256), lds, st, g2, ws + g.o_eff, p->d_zones,
                     p->d_idx, dx, p->Z, F, g.T1, p->W, p->Ctot, (int)T, g.N, p->S);
  ISD_LAUNCH_CHECK();
  return ISD_OK;
}

extern "C" int isd_conv4_backward(const isd_conv4_plan* p, const float* x, const float* params, const float* dfeat,
                                  float* dparams, void* workspace, int64_t B, int64_t T, void* stream) {
  return conv4_backward_impl(p, x, params, dfeat, dparams, nullptr, workspace, B, T, stream);
}

extern "C" int isd_conv4_backward_x(const isd_conv4_plan* p, const float* x, const float* params, const float* dfeat,
                                    float* dparams, float* dx, void* workspace, int64_t B, int64_t T, void* stream) {
  ISD_CHECK_ARG(dx || B == 0, "isd_conv4_backward_x: null dx");
  return conv4_backward_impl(p, x, params, dfeat, dparams, dx, workspace, B, T, stream);
}

#ifdef ISD_CF_TIMING
extern "C" int isd_debug_conv_marks(long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(isd::cf_times), sizeof(long long) * 32) == hipSuccess ? 0 : 1;
}
#endif
