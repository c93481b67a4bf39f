// Transformer tail of FAST in ONE launch per direction (reference: src/fast/models/fast.py:10-29 `AttentionBlock`,
// :260-268 `forward_transformer`; the mode the reference trains, src/fast/train/trainer.py:58, at batch 64,
// scripts/train_fast.py:274):
//   cls token + positional embedding -> L x [ x += MHA(LN1(x));  x += MLP(LN2(x)) ] -> last_layer(dropout(x[:, 0])).
// The sequence is <= 8 tokens of 32 features: per trial the whole tail is ~50 k multiply-adds per layer, and as
// separate launches (tail.hip + fc.hip: ~45 forward, ~60 backward launches) it is pure launch latency.
//
// Mapping: a wave holds the tokens of floor(64 / S) trials, ONE LANE PER TOKEN, and every activation tensor of the
// wave lives in LDS as [feature][token] columns (stride 68 dwords).  That one layout serves three access patterns
// without conflicts: the token's own lane walks its column (LayerNorm, GELU, softmax, the residual stream in
// registers), the lanes of a trial read one another's columns by address (attention), and the matrix cores take the
// columns directly as operands: a dense layer is  Y[64 tokens][N] = X[64][K] W[K][N]  on v_mfma_f32_16x16x4_f32 with
// lane l supplying feature 4 s + (l >> 4) of token 16 mt + (l & 15) in k-step s, and a weight gradient is the same
// product with the token index as K (lane l supplies feature l & 15 of token 4 s + (l >> 4)).  fp32 in, fp32
// accumulate: the sums equal the fmaf chain of the per-operator kernels.  The block's weights are staged in LDS once
// per layer ([K][N] images: transposed for the forward, as stored for the backward's W^T products).
// (Earlier versions of this file multiplied per lane with the weights in the scalar file, then as LDS broadcasts:
// 470 / 360 us per forward at any batch size -- one wave per SIMD waiting on every operand.  See DESIGN.md.)
// Dropout (attention probabilities, both MLP dropouts, the cls token) is counter-based: element e of site s of
// layer l draws from hash(seed, l, s, e), regenerated in the backward pass.
//
// The forward keeps what the backward needs, one record block per (layer, wave) laid out [field][lane] so that every
// store is one 256-byte row: LN statistics, qkv, attention probabilities, the MLP pre-activation, the two
// residual-stream snapshots.  The backward leaves each wave's partial parameter gradients in its own slab (the
// parameter block's layout) and tail_fused_reduce_kernel adds the slabs in a fixed order: deterministic.
#include "common.h"
#include <math.h>

namespace isd {

#ifdef TF_TIMING                      // tools/ubench/tail_phases.hip: shader-clock stamps of wave 0, first layer visited
__device__ long long tf_times[32];
#define TF_MARK(k) do { if (blockIdx.x == 0 && threadIdx.x == 0 && tf_first) tf_times[k] = __builtin_readcyclecounter(); } while (0)
#else
#define TF_MARK(k) do { } while (0)
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* tf_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* tf_gbl_ptr_t;

// D = dim_token (template parameter: 32 in production, 16 in the reference's small test configuration); the MLP hidden
// width is 2 D (fast.py:236)
constexpr int kTMaxS = 8;          // tokens per trial, cls included
constexpr int kTMaxL = 8;          // transformer blocks
constexpr int kTMaxCls = 16;
constexpr int kCS = 68;            // column stride in dwords: own-column, MFMA-operand and 16-byte tile stores all spread

// saved per (layer, token): field offsets (each field row is 64 lanes wide)
template <int D> struct Sv {
  static constexpr int xin = 0, ln1 = D, qkv = D + 2, prob = 4 * D + 2, ctx = prob + 8 * kTMaxS, xmid = ctx + D,
                       ln2 = xmid + D, hpre = ln2 + 2, total = hpre + 2 * D;
};

// The tail's parameters live in ONE flat block (the host packs them once; nn.FAST keeps them packed) and the kernels
// take it as a __restrict__ argument with integer offsets.  Order inside the block:
//   pos_embedding [n_pos][D] | cls_token [D] | per block: layer_norm_1.{weight,bias} |
//   attn.in_proj_{weight [3D, D], bias} | attn.out_proj.{weight, bias} | layer_norm_2.{weight,bias} |
//   linear.0.{weight [2D, D], bias} | linear.3.{weight [D, 2D], bias} | ... | last_layer.{weight [n_cls, D], bias}
struct TailLayerOff {
  int ln1w, ln1b, inw, inb, ow, ob, ln2w, ln2b, w1, b1, w2, b2;
};
struct TailMeta {
  int pos, cls, lastw, lastb;
  TailLayerOff layer[kTMaxL];
  long long B;
  int N, S, H, L, n_cls, n_pos;                          // n_pos = rows of the positional table (>= S)
  float p_attn, p_mlp, p_cls;
  unsigned long long seed;
  const unsigned long long* seed_dev;                    // optional device-resident step counter mixed into the seed
};

// the launch's dropout seed: a captured graph replays the same arguments, so its masks advance through *seed_dev
__device__ __forceinline__ unsigned long long tf_seed(const TailMeta& a) {
  return a.seed_dev ? a.seed + 0xD1342543DE82EF95ull * *a.seed_dev : a.seed;
}

// Counter-based dropout: element e of (layer, site) is kept when hash32(e, key(seed, layer, site)) / 2^24 >= p and then
// scaled by 1 / (1 - p); p = 0 keeps everything with scale 1 through the same arithmetic (no branch).  The key is
// wave-uniform (scalar ALU), the per-element part is murmur3's 32-bit finaliser.
struct TfDrop {
  unsigned key;
  float p, inv;
};
__device__ __forceinline__ TfDrop tf_drop(unsigned long long seed, int layer, int site, float p) {
  unsigned long long v = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(layer * 8 + site + 1);
  v ^= v >> 30; v *= 0xBF58476D1CE4E5B9ull;
  v ^= v >> 27; v *= 0x94D049BB133111EBull;
  v ^= v >> 31;
  return {(unsigned)v ^ (unsigned)(v >> 32), p, 1.f / (1.f - p)};
}
__device__ __forceinline__ float tf_keep(const TfDrop& d, unsigned long long e) {
  unsigned h = ((unsigned)e ^ ((unsigned)(e >> 32) * 0x27D4EB2Fu)) * 0x9E3779B1u ^ d.key;
  h ^= h >> 16; h *= 0x85EBCA6Bu;
  h ^= h >> 13; h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return (float)(h >> 8) * (1.f / 16777216.f) >= d.p ? d.inv : 0.f;
}

__device__ __forceinline__ float tf_gelu(float x) { return gelu_fast(x); }             // common.h: A&S erf, 1.5e-7
__device__ __forceinline__ float tf_gelu_grad(float x) { return gelu_grad_fast(x); }

// ------------------------------------------------------------------------------------------- staging the weights
// dst[i * STRIDE + o] = W[o][i]
template <int NO, int NI, int STRIDE>
__device__ __forceinline__ void stage_transposed(const float* __restrict__ W, float* dst, int tid) {
  constexpr int kQ = NI / 4;
#pragma unroll 4
  for (int idx = tid; idx < NO * kQ; idx += 256) {
    const int o = idx / kQ, i4 = (idx - o * kQ) * 4;
    const f32x4 v = *reinterpret_cast<const f32x4*>(W + o * NI + i4);
    dst[(i4 + 0) * STRIDE + o] = v[0];
    dst[(i4 + 1) * STRIDE + o] = v[1];
    dst[(i4 + 2) * STRIDE + o] = v[2];
    dst[(i4 + 3) * STRIDE + o] = v[3];
  }
}

__device__ __forceinline__ void stage_copy(const float* __restrict__ src, float* dst, int n, int tid) {
  for (int idx = tid * 4; idx < n; idx += 1024) *reinterpret_cast<f32x4*>(dst + idx) = *reinterpret_cast<const f32x4*>(src + idx);
}

// straight copy of n floats (multiple of 4, both sides 16-byte aligned) by LDS-DMA: every piece in flight at once,
// no registers; retire with s_waitcnt vmcnt(0)
__device__ __forceinline__ void stage_dma(const float* __restrict__ src, float* dst, int n, int tid) {
  const int n4 = n / 4, lane = tid & 63;
  for (int e0 = (tid >> 6) * 64; e0 < n4; e0 += 256) {    // a wave moves 1 KiB pieces; dst base is wave-uniform
    if (e0 + lane < n4)
      __builtin_amdgcn_global_load_lds((tf_gbl_ptr_t)(src + (int64_t)(e0 + lane) * 4), (tf_lds_ptr_t)(dst + e0 * 4), 16, 0, 0);
  }
}
__device__ __forceinline__ void stage_dma_wait() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// ------------------------------------------------------------------------------------------- the workgroup
// FOUR waves share the 64 tokens of a workgroup (one token per lane in every wave) and its LDS columns; the four SIMDs
// of a CU work on the same tokens: a dense layer gives each wave one 16-token row tile, elementwise work gives wave w
// the w-th quarter of the features, attention gives it every fourth head, a weight gradient every fourth tile.
// (One wave per workgroup ran every phase as one dependent chain on one SIMD: 120 / 385 us at any batch size.)
constexpr int kNW = 4;

// Y[tok][n] = bias[n] + sum_k X[tok][k] W[k * WS + n] for the 16 tokens of row tile `mt`: X = columns acol .. acol + K,
// Y -> columns ocol .. ocol + N, W and bias in LDS.  HOLD keeps the K / 4 A fragments in registers.
template <int K, int N, int WS, bool BIAS, bool HOLD>
__device__ __forceinline__ void gemm_cols(float* cols, int acol, const float* W, const float* bias, int ocol, int lane,
                                          int mt) {
  const int m = lane & 15, kq = lane >> 4;
  float af[HOLD ? K / 4 : 1];
  if (HOLD) {
#pragma unroll
    for (int t = 0; t < K / 4; ++t) af[t] = cols[(acol + 4 * t + kq) * kCS + 16 * mt + m];
  }
#pragma unroll 2
  for (int nb = 0; nb < N; nb += 16) {
    const float bv = BIAS ? bias[nb + m] : 0.f;
    f32x4 acc = {bv, bv, bv, bv};
#pragma unroll
    for (int t = 0; t < K / 4; ++t) {
      const float bw = W[(4 * t + kq) * WS + nb + m];
      const float av = HOLD ? af[t] : cols[(acol + 4 * t + kq) * kCS + 16 * mt + m];
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bw, acc, 0, 0, 0);
    }
    // D[row = token 4 kq + r][col = output m]: four consecutive tokens of one output column per lane
    *reinterpret_cast<f32x4*>(cols + (ocol + nb + m) * kCS + 16 * mt + 4 * kq) = acc;
  }
}

// per-token sum of one value over the four waves (each holds a feature quarter): scratch = 4 free columns
__device__ __forceinline__ float wg_sum4(float* cols, int c0, int wave, int lane, float v) {
  cols[(c0 + wave) * kCS + lane] = v;
  __syncthreads();
  return (cols[c0 * kCS + lane] + cols[(c0 + 1) * kCS + lane]) + (cols[(c0 + 2) * kCS + lane] + cols[(c0 + 3) * kCS + lane]);
}

// LayerNorm of the token's D features, this thread holding FS = D / 4 of them (eps inside the sqrt, biased variance:
// nn.LayerNorm); w, b = this thread's slice of the LDS copies; output to the thread's slots of columns hcol ..;
// scratch = 8 free columns.  Two barriers inside.
template <int D, int FS>
__device__ __forceinline__ void layer_norm_to_cols(const float (&x)[FS], const float* w, const float* b, float* hcol,
                                                   float* cols, int scratch, int wave, int lane, float& mu, float& rstd) {
  float s = 0.f;
#pragma unroll
  for (int d = 0; d < FS; ++d) s += x[d];
  mu = wg_sum4(cols, scratch, wave, lane, s) * (1.f / D);
  float q = 0.f;
#pragma unroll
  for (int d = 0; d < FS; ++d) {
    const float c = x[d] - mu;
    q = fmaf(c, c, q);
  }
  rstd = rsqrtf(wg_sum4(cols, scratch + 4, wave, lane, q) * (1.f / D) + 1e-5f);
#pragma unroll
  for (int d = 0; d < FS; ++d) hcol[d * kCS] = (x[d] - mu) * rstd * w[d] + b[d];
}

// LDS image of one block's weights for the forward: [K][N] = transposed matrices (rows padded by 4 floats to spread the
// staging stores), then the small parameters
template <int D> struct FwdImg {
  static constexpr int SI = 3 * D + 4, SO = D + 4, S1 = 2 * D + 4, S2 = D + 4;            // row strides
  static constexpr int win = 0, wo = win + D * SI, w1 = wo + D * SO, w2 = w1 + D * S1, small = w2 + 2 * D * S2;
  static constexpr int ln1w = small, ln1b = ln1w + D, inb = ln1b + D, ob = inb + 3 * D, ln2w = ob + D, ln2b = ln2w + D,
                       b1 = ln2b + D, b2 = b1 + 2 * D, total = b2 + D;
};

// tokin [B][N][D]: output of input_layer (Linear + GELU); logits [B][n_cls]; save [L][gridDim.x][Sv<D>::total][64]
// and xfinal [B][D] (the cls token entering last_layer, after dropout) with TRAIN
// kDH = head width (compile time: the attention loops carry no branches); TRAIN: dropout + the record
template <int kTD, int kDH, bool TRAIN>
__global__ __launch_bounds__(kNW * 64) void tail_fused_fwd_kernel(const float* __restrict__ P, TailMeta a,
                                                                  const float* __restrict__ tokin,
                                                                  float* __restrict__ logits, float* __restrict__ save,
                                                                  float* __restrict__ xfinal) {
  constexpr int kTHid = 2 * kTD, FS = kTD / kNW, HS = kTHid / kNW;
  using Img = FwdImg<kTD>;
  // column regions: q | k | v (the MLP's hidden vector and LayerNorm scratch reuse it), and two D-wide regions
  constexpr int RQ = 0, RA = 3 * kTD, RC = 4 * kTD, kColTotal = 5 * kTD;
  constexpr int kSvXin = Sv<kTD>::xin, kSvLn1 = Sv<kTD>::ln1, kSvQkv = Sv<kTD>::qkv, kSvProb = Sv<kTD>::prob,
                kSvCtx = Sv<kTD>::ctx, kSvXmid = Sv<kTD>::xmid, kSvLn2 = Sv<kTD>::ln2, kSvHpre = Sv<kTD>::hpre,
                kSvTotal = Sv<kTD>::total;
  __shared__ __attribute__((aligned(16))) float wl[Img::total];
  __shared__ __attribute__((aligned(16))) float cols[kColTotal * kCS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int S = a.S, G = 64 / S;                          // trials per workgroup
  const int g = lane / S, i = lane - g * S;
  const int64_t b = (int64_t)blockIdx.x * G + g;
  const bool live = g < G && b < a.B;
  const int gs = (g < G ? g : 0) * S;                     // first lane of this token's trial
  const int64_t bc = live ? b : 0;                        // dead lanes recompute trial 0 (finite values, never used)
  const int64_t tok = bc * S + i;                         // global token row (dropout counters)
  constexpr int kH = kTD / kDH;
  const float scale = 1.f / sqrtf((float)kDH);
  float* my = cols + lane;
  const unsigned long long seed = tf_seed(a);
  const int f0 = FS * wave, h0 = HS * wave;               // this thread's feature / hidden-feature quarter

  float x[FS];
#pragma unroll
  for (int d = 0; d < FS; ++d) {
    const float base = i == 0 ? P[a.cls + f0 + d] : tokin[(bc * a.N + (i - 1)) * kTD + f0 + d];
    x[d] = base + P[a.pos + i * kTD + f0 + d];
  }
  for (int l = 0; l < a.L; ++l) {
    const TailLayerOff& wo = a.layer[l];
#ifdef TF_TIMING
    const bool tf_first = l == 1;
#endif
    TF_MARK(0);
    stage_transposed<3 * kTD, kTD, Img::SI>(P + wo.inw, wl + Img::win, tid);
    stage_transposed<kTD, kTD, Img::SO>(P + wo.ow, wl + Img::wo, tid);
    stage_transposed<kTHid, kTD, Img::S1>(P + wo.w1, wl + Img::w1, tid);
    stage_transposed<kTD, kTHid, Img::S2>(P + wo.w2, wl + Img::w2, tid);
    stage_copy(P + wo.ln1w, wl + Img::ln1w, 2 * kTD, tid);
    stage_copy(P + wo.inb, wl + Img::inb, 3 * kTD, tid);
    stage_copy(P + wo.ob, wl + Img::ob, kTD, tid);
    stage_copy(P + wo.ln2w, wl + Img::ln2w, 2 * kTD, tid);
    stage_copy(P + wo.b1, wl + Img::b1, 2 * kTD, tid);
    stage_copy(P + wo.b2, wl + Img::b2, kTD, tid);
    // this workgroup's record block of layer l: field f of this lane at sv[f * 64]
    float* sv = save + (((int64_t)l * gridDim.x + blockIdx.x) * kSvTotal) * 64 + lane;
    const TfDrop d_attn = tf_drop(seed, l, 0, a.p_attn), d_mlp1 = tf_drop(seed, l, 1, a.p_mlp),
                 d_mlp2 = tf_drop(seed, l, 2, a.p_mlp);
    float mu, rstd;
    if (TRAIN) {
#pragma unroll
      for (int d = 0; d < FS; ++d) sv[(kSvXin + f0 + d) * 64] = x[d];
    }
    __syncthreads();                                      // weights staged; the previous block's columns are dead
    TF_MARK(1);
    layer_norm_to_cols<kTD, FS>(x, wl + Img::ln1w + f0, wl + Img::ln1b + f0, my + (RA + f0) * kCS, cols, RC, wave, lane, mu,
                                rstd);
    if (TRAIN && wave == 0) { sv[kSvLn1 * 64] = mu; sv[(kSvLn1 + 1) * 64] = rstd; }
    __syncthreads();
    TF_MARK(2);
    gemm_cols<kTD, 3 * kTD, Img::SI, true, true>(cols, RA, wl + Img::win, wl + Img::inb, RQ, lane, wave);   // q | k | v
    __syncthreads();
    TF_MARK(3);
    if (TRAIN) {
#pragma unroll
      for (int o = 0; o < 3 * FS; ++o) sv[(kSvQkv + 3 * f0 + o) * 64] = my[(RQ + 3 * f0 + o) * kCS];
    }
    // attention: this token's query against the keys / values of its trial (lanes gs .. gs + S - 1), every fourth head
    // per wave.  All kTMaxS positions are computed (positions past S read token 0 and are masked): no control flow
    // inside a head.
    TF_MARK(4);
    // the wave's heads side by side (unrolled: the LDS reads of one head's scores run under the other's exponentials;
    // one head at a time, the phase was a third of the forward pass at the reference's batch: 9 300 -> 6 000 cycles)
#pragma unroll
    for (int hi = 0; hi < (kH + kNW - 1) / kNW; ++hi) {
      const int hh = wave + hi * kNW;
      if (hh >= kH) break;
      const int qc = RQ + hh * kDH, kc = qc + kTD, vc = kc + kTD;
      float qh[kDH];
#pragma unroll
      for (int t = 0; t < kDH; ++t) qh[t] = my[(qc + t) * kCS];
      float sc[kTMaxS];
      float mx = -INFINITY;
#pragma unroll
      for (int j = 0; j < kTMaxS; ++j) {
        const int jj = gs + (j < S ? j : 0);
        float d0 = 0.f;
#pragma unroll
        for (int t = 0; t < kDH; ++t) d0 = fmaf(qh[t], cols[(kc + t) * kCS + jj], d0);
        sc[j] = j < S ? d0 * scale : -INFINITY;
        mx = fmaxf(mx, sc[j]);
      }
      float den = 0.f;
#pragma unroll
      for (int j = 0; j < kTMaxS; ++j) {
        sc[j] = expf(sc[j] - mx);                        // exp(-inf) = 0 past S
        den += sc[j];
      }
      float o[kDH];
#pragma unroll
      for (int t = 0; t < kDH; ++t) o[t] = 0.f;
#pragma unroll
      for (int j = 0; j < kTMaxS; ++j) {
        const int jj = gs + (j < S ? j : 0);
        const float p = sc[j] / den;
        float pd = p;
        if (TRAIN) {
          sv[(kSvProb + hh * kTMaxS + j) * 64] = p;
          pd *= tf_keep(d_attn, (unsigned long long)((tok * kH + hh) * kTMaxS + j));
        }
#pragma unroll
        for (int t = 0; t < kDH; ++t) o[t] = fmaf(pd, cols[(vc + t) * kCS + jj], o[t]);
      }
#pragma unroll
      for (int t = 0; t < kDH; ++t) my[(RC + hh * kDH + t) * kCS] = o[t];
    }
    __syncthreads();
    if (TRAIN) {
#pragma unroll
      for (int d = 0; d < FS; ++d) sv[(kSvCtx + f0 + d) * 64] = my[(RC + f0 + d) * kCS];
    }
    TF_MARK(5);
    gemm_cols<kTD, kTD, Img::SO, true, true>(cols, RC, wl + Img::wo, wl + Img::ob, RA, lane, wave);         // output projection
    __syncthreads();
    TF_MARK(6);
#pragma unroll
    for (int d = 0; d < FS; ++d) x[d] += my[(RA + f0 + d) * kCS];
    if (TRAIN) {
#pragma unroll
      for (int d = 0; d < FS; ++d) sv[(kSvXmid + f0 + d) * 64] = x[d];
    }
    layer_norm_to_cols<kTD, FS>(x, wl + Img::ln2w + f0, wl + Img::ln2b + f0, my + (RC + f0) * kCS, cols, RQ, wave, lane, mu,
                                rstd);
    if (TRAIN && wave == 0) { sv[kSvLn2 * 64] = mu; sv[(kSvLn2 + 1) * 64] = rstd; }
    __syncthreads();
    TF_MARK(7);
    gemm_cols<kTD, kTHid, Img::S1, true, true>(cols, RC, wl + Img::w1, wl + Img::b1, RQ, lane, wave);
    __syncthreads();
    TF_MARK(8);
    {                                                     // GELU + dropout in place on this thread's hidden quarter
      float pre[HS];
#pragma unroll
      for (int k = 0; k < HS; ++k) pre[k] = my[(RQ + h0 + k) * kCS];
#pragma unroll
      for (int k = 0; k < HS; ++k) {
        float v = tf_gelu(pre[k]);
        if (TRAIN) {
          sv[(kSvHpre + h0 + k) * 64] = pre[k];
          v *= tf_keep(d_mlp1, (unsigned long long)(tok * kTHid + h0 + k));
        }
        my[(RQ + h0 + k) * kCS] = v;
      }
    }
    __syncthreads();
    TF_MARK(9);
    gemm_cols<kTHid, kTD, Img::S2, true, true>(cols, RQ, wl + Img::w2, wl + Img::b2, RA, lane, wave);
    __syncthreads();
    TF_MARK(10);
#pragma unroll
    for (int d = 0; d < FS; ++d) {
      const float kp = TRAIN ? tf_keep(d_mlp2, (unsigned long long)(tok * kTD + f0 + d)) : 1.f;
      x[d] = fmaf(my[(RA + f0 + d) * kCS], kp, x[d]);
    }
    TF_MARK(11);
  }
  // ---- last_layer on the cls tokens: the four feature quarters meet in the RC columns
  __syncthreads();
  {
    const TfDrop d_cls = tf_drop(seed, kTMaxL, 0, a.p_cls);
#pragma unroll
    for (int d = 0; d < FS; ++d) {
      if (TRAIN) x[d] *= tf_keep(d_cls, (unsigned long long)(bc * kTD + f0 + d));
      my[(RC + f0 + d) * kCS] = x[d];
    }
  }
  __syncthreads();
  if (live && i == 0) {
    if (TRAIN) {
#pragma unroll
      for (int d = 0; d < FS; ++d) xfinal[bc * kTD + f0 + d] = x[d];
    }
    for (int c = wave; c < a.n_cls; c += kNW) {
      float acc = P[a.lastb + c];
#pragma unroll 8
      for (int d = 0; d < kTD; ++d) acc = fmaf(P[a.lastw + c * kTD + d], my[(RC + d) * kCS], acc);
      logits[bc * a.n_cls + c] = acc;
    }
  }
}

// ----------------------------------------------------------------------------------------------------- backward
// One launch for the whole tail, the thread <-> (token, feature quarter) mapping of the forward.  Data gradients
// dX = G W  are the same column products with the weights as stored ([out][in] = [K][N]); weight gradients are sums
// over tokens of outer products: 16 MFMAs per 16 x 16 tile with the token index as K, every fourth tile per wave, and
// the bias / LayerNorm-parameter sums are that product against a vector of ones.

// dW[o][i] = sum_tok g[tok][o] in[tok][i] (o < no_valid rows written), db[o] = sum_tok g[tok][o]; NI == 0: sums only
template <int NO, int NI>
__device__ __forceinline__ void wgrad_tiles(const float* cols, int gcol, int icol, float* __restrict__ dw,
                                            float* __restrict__ db, int lane, int wave, int no_valid) {
  const int m = lane & 15, kq = lane >> 4;
  constexpr int n_ib = NI / 16, n_ob = NO / 16;
  if (NI > 0) {
#pragma unroll 1
    for (int t = wave; t < n_ob * n_ib; t += kNW) {
      const int ob = (t / (n_ib > 0 ? n_ib : 1)) * 16, ib = (t % (n_ib > 0 ? n_ib : 1)) * 16;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 16; ++s)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(cols[(gcol + ob + m) * kCS + 4 * s + kq],
                                                   cols[(icol + ib + m) * kCS + 4 * s + kq], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (ob + 4 * kq + r < no_valid) dw[(ob + 4 * kq + r) * NI + ib + m] = acc[r];
    }
  }
#pragma unroll 1
  for (int t = (wave + n_ob * n_ib) % kNW; t < n_ob; t += kNW) {       // the sums start where the tiles left off
    const int ob = t * 16;
    f32x4 accb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 16; ++s)
      accb = __builtin_amdgcn_mfma_f32_16x16x4f32(cols[(gcol + ob + m) * kCS + 4 * s + kq], 1.f, accb, 0, 0, 0);
    if (m == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (ob + 4 * kq + r < no_valid) db[ob + 4 * kq + r] = accb[r];
    }
  }
}

// LayerNorm backward for one token, this thread holding FS of its D features: xs = the saved input (record rows, 64
// apart, this thread's slice), w = its slice of the LDS weight, dh read from its slots of columns dhcol.  Adds the
// input gradient to dx and leaves [dh * xhat | dh] in its slots of the 2 D columns at pcol (their sums over tokens
// are dweight | dbias).  scratch = 8 free columns; two barriers inside.
template <int D, int FS>
__device__ __forceinline__ void layer_norm_backward(const float* __restrict__ xs, float mu, float rstd, const float* w,
                                                    const float* dhcol, float (&dx)[FS], float* pcol, float* cols,
                                                    int scratch, int wave, int lane) {
  float dxh[FS], xh[FS];
  float c1 = 0.f, c2 = 0.f;
#pragma unroll
  for (int d = 0; d < FS; ++d) {
    const float dh = dhcol[d * kCS];
    xh[d] = (xs[d * 64] - mu) * rstd;
    dxh[d] = dh * w[d];
    c1 += dxh[d];
    c2 = fmaf(dxh[d], xh[d], c2);
    pcol[d * kCS] = dh * xh[d];
    pcol[(D + d) * kCS] = dh;
  }
  c1 = wg_sum4(cols, scratch, wave, lane, c1) * (1.f / D);
  c2 = wg_sum4(cols, scratch + 4, wave, lane, c2) * (1.f / D);
#pragma unroll
  for (int d = 0; d < FS; ++d) dx[d] += rstd * (dxh[d] - c1 - xh[d] * c2);
}

// dlogits [B][n_cls] (already scaled by the caller's loss weight); dtokin [B][N][D]; slab [gridDim.x][ptot]
template <int kTD, int kDH>
__global__ __launch_bounds__(kNW * 64) void tail_fused_bwd_kernel(const float* __restrict__ P, TailMeta a,
                                                                  const float* __restrict__ save,
                                                                  const float* __restrict__ xfinal,
                                                                  const float* __restrict__ dlogits,
                                                                  float* __restrict__ dtokin, float* __restrict__ slab,
                                                                  int ptot) {
  constexpr int kTHid = 2 * kTD, FS = kTD / kNW, HS = kTHid / kNW;
  // column regions: q | k | v (the MLP's hidden vector and the LayerNorm sums reuse it), two D-wide operand regions,
  // and 16 columns per wave for the attention's probability rows
  constexpr int RQ = 0, RA = 3 * kTD, RC = 4 * kTD, RB = 5 * kTD, RH = RQ, kColTotal = 5 * kTD + 16 * kNW;
  constexpr int kLayerFloats = 8 * kTD * kTD + 11 * kTD;   // one block's parameters, copied to LDS as they are
  constexpr int kSvXin = Sv<kTD>::xin, kSvLn1 = Sv<kTD>::ln1, kSvQkv = Sv<kTD>::qkv, kSvProb = Sv<kTD>::prob,
                kSvCtx = Sv<kTD>::ctx, kSvXmid = Sv<kTD>::xmid, kSvLn2 = Sv<kTD>::ln2, kSvHpre = Sv<kTD>::hpre,
                kSvTotal = Sv<kTD>::total;
  static_assert(kTD >= 16, "column regions");
  __shared__ __attribute__((aligned(16))) float wl[kLayerFloats];
  __shared__ __attribute__((aligned(16))) float cols[kColTotal * kCS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int S = a.S, G = 64 / S;
  const int g = lane / S, i = lane - g * S;
  const int64_t b = (int64_t)blockIdx.x * G + g;
  const bool live = g < G && b < a.B;
  const float lv = live ? 1.f : 0.f;                      // dead lanes carry the forward's (finite) records, weighted by 0
  const int gs = (g < G ? g : 0) * S;
  const int64_t bc = live ? b : 0;
  const int64_t tok = bc * S + i;
  constexpr int kH = kTD / kDH;
  const float scale = 1.f / sqrtf((float)kDH);
  float* my = cols + lane;
  float* slabw = slab + (int64_t)blockIdx.x * ptot;
  const unsigned long long seed = tf_seed(a);
  const int f0 = FS * wave, h0 = HS * wave;

  stage_dma(P + a.layer[a.L - 1].ln1w, wl, kLayerFloats, tid);           // the last block's weights: in flight
  // ---- last_layer: logits = W x0 + b on the cls token after dropout
  float dx[FS];
  {
    const bool cl = live && i == 0;
    if (wave == 0) {
#pragma unroll
      for (int c = 0; c < 16; ++c) my[(RA + c) * kCS] = (cl && c < a.n_cls) ? dlogits[bc * a.n_cls + c] : 0.f;
    }
#pragma unroll
    for (int d = 0; d < FS; ++d) my[(RC + f0 + d) * kCS] = cl ? xfinal[bc * kTD + f0 + d] : 0.f;
    __syncthreads();
    wgrad_tiles<16, kTD>(cols, RA, RC, slabw + a.lastw, slabw + a.lastb, lane, wave, a.n_cls);
#pragma unroll
    for (int d = 0; d < FS; ++d) dx[d] = 0.f;
    for (int c = 0; c < a.n_cls; ++c) {
      const float gl = my[(RA + c) * kCS];
      const float* w = P + a.lastw + c * kTD + f0;
#pragma unroll
      for (int d = 0; d < FS; ++d) dx[d] = fmaf(w[d], gl, dx[d]);
    }
    const TfDrop d_cls = tf_drop(seed, kTMaxL, 0, a.p_cls);
#pragma unroll
    for (int d = 0; d < FS; ++d) dx[d] *= tf_keep(d_cls, (unsigned long long)(bc * kTD + f0 + d));
    __syncthreads();
  }

  for (int l = a.L - 1; l >= 0; --l) {
    const TailLayerOff& wo = a.layer[l];
    const float* sv = save + (((int64_t)l * gridDim.x + blockIdx.x) * kSvTotal) * 64 + lane;
    const int l0 = wo.ln1w;                               // wl + (wo.<tensor> - l0) = that tensor's LDS copy
    const TfDrop d_attn = tf_drop(seed, l, 0, a.p_attn), d_mlp1 = tf_drop(seed, l, 1, a.p_mlp),
                 d_mlp2 = tf_drop(seed, l, 2, a.p_mlp);
    // ---- x_out = xmid + drop2(W2 m + b2),  m = drop1(gelu(hpre))
#pragma unroll
    for (int d = 0; d < FS; ++d)
      my[(RA + f0 + d) * kCS] = lv * dx[d] * tf_keep(d_mlp2, (unsigned long long)(tok * kTD + f0 + d));
    float pre[HS];
#pragma unroll
    for (int k = 0; k < HS; ++k) pre[k] = sv[(kSvHpre + h0 + k) * 64];
#pragma unroll
    for (int k = 0; k < HS; ++k)
      my[(RH + h0 + k) * kCS] = tf_gelu(pre[k]) * tf_keep(d_mlp1, (unsigned long long)(tok * kTHid + h0 + k));
    __syncthreads();
    wgrad_tiles<kTD, kTHid>(cols, RA, RH, slabw + wo.w2, slabw + wo.b2, lane, wave, kTD);
    stage_dma_wait();                                     // this thread's pieces of the block's weights have landed
    __syncthreads();                                      // ... everyone's; and m is no longer read
    gemm_cols<kTD, kTHid, kTHid, false, true>(cols, RA, wl + (wo.w2 - l0), nullptr, RH, lane, wave);    // dm = dy2 W2
    __syncthreads();
#pragma unroll
    for (int k = 0; k < HS; ++k)                          // hpre = W1 h2 + b1: gradient through dropout and GELU
      my[(RH + h0 + k) * kCS] *= tf_gelu_grad(pre[k]) * tf_keep(d_mlp1, (unsigned long long)(tok * kTHid + h0 + k));
    {
      const float mu = sv[kSvLn2 * 64], rstd = sv[(kSvLn2 + 1) * 64];
      const float* lw = wl + (wo.ln2w - l0) + f0;
      const float* lb = wl + (wo.ln2b - l0) + f0;
#pragma unroll
      for (int d = 0; d < FS; ++d) my[(RC + f0 + d) * kCS] = (sv[(kSvXmid + f0 + d) * 64] - mu) * rstd * lw[d] + lb[d];
      __syncthreads();
      wgrad_tiles<kTHid, kTD>(cols, RH, RC, slabw + wo.w1, slabw + wo.b1, lane, wave, kTHid);
      gemm_cols<kTHid, kTD, kTD, false, true>(cols, RH, wl + (wo.w1 - l0), nullptr, RA, lane, wave);    // dh2 = dhpre W1
      __syncthreads();
      layer_norm_backward<kTD, FS>(sv + (kSvXmid + f0) * 64, mu, rstd, lw, my + (RA + f0) * kCS, dx, my + (RQ + f0) * kCS,
                                   cols, RC, wave, lane);
      __syncthreads();
      wgrad_tiles<2 * kTD, 0>(cols, RQ, 0, nullptr, slabw + wo.ln2w, lane, wave, 2 * kTD);
    }
    // ---- xmid = xin + Wo ctx + bo
#pragma unroll
    for (int d = 0; d < FS; ++d) {
      my[(RA + f0 + d) * kCS] = dx[d];
      my[(RC + f0 + d) * kCS] = sv[(kSvCtx + f0 + d) * 64];
    }
    __syncthreads();
    wgrad_tiles<kTD, kTD>(cols, RA, RC, slabw + wo.ow, slabw + wo.ob, lane, wave, kTD);
    __syncthreads();                                      // ctx is dead, the LayerNorm sums are taken
    gemm_cols<kTD, kTD, kTD, false, true>(cols, RA, wl + (wo.ow - l0), nullptr, RC, lane, wave);        // dctx = dxmid Wo
#pragma unroll
    for (int o = 0; o < 3 * FS; ++o) my[(RQ + 3 * f0 + o) * kCS] = sv[(kSvQkv + 3 * f0 + o) * 64];
    __syncthreads();
    // ---- attention: ctx_i = sum_j drop(p_ij) v_j,  p = softmax(scale q k^T); every fourth head per wave, positions
    // past S computed on token 0 and masked (the saved probabilities there are 0): a head carries no control flow.
    // The hand-overs inside a head are between lanes of this wave only (its own 16 columns at rb).
    {
      const int rb = RB + 16 * wave;
#pragma unroll 1
      for (int hh = wave; hh < kH; hh += kNW) {
        const int qc = RQ + hh * kDH, kc = qc + kTD, vc = kc + kTD, cc = RC + hh * kDH;
        float dc[kDH], pr[kTMaxS], dp[kTMaxS];
#pragma unroll
        for (int t = 0; t < kDH; ++t) dc[t] = my[(cc + t) * kCS];
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < kTMaxS; ++j) {
          const int jj = gs + (j < S ? j : 0);
          pr[j] = sv[(kSvProb + hh * kTMaxS + j) * 64];
          const float kp = tf_keep(d_attn, (unsigned long long)((tok * kH + hh) * kTMaxS + j));
          float dpd = 0.f;
#pragma unroll
          for (int t = 0; t < kDH; ++t) dpd = fmaf(dc[t], cols[(vc + t) * kCS + jj], dpd);
          dp[j] = dpd * kp;
          dot = fmaf(dp[j], pr[j], dot);
          my[(rb + j) * kCS] = pr[j] * kp;
        }
        float dq[kDH];
#pragma unroll
        for (int t = 0; t < kDH; ++t) dq[t] = 0.f;
#pragma unroll
        for (int j = 0; j < kTMaxS; ++j) {
          const int jj = gs + (j < S ? j : 0);
          const float ds = pr[j] * (dp[j] - dot);
          my[(rb + 8 + j) * kCS] = ds;
#pragma unroll
          for (int t = 0; t < kDH; ++t) dq[t] = fmaf(ds, cols[(kc + t) * kCS + jj], dq[t]);
        }
        wave_lds_sync();
        float dk[kDH], dv[kDH];
#pragma unroll
        for (int t = 0; t < kDH; ++t) dk[t] = dv[t] = 0.f;
#pragma unroll
        for (int ii = 0; ii < kTMaxS; ++ii) {             // row ii of the trial's probability matrix, this token's column
          const int li = gs + (ii < S ? ii : 0);
          const float ok = ii < S ? 1.f : 0.f;
          const float pdv = ok * cols[(rb + i) * kCS + li], dsv = ok * cols[(rb + 8 + i) * kCS + li];
#pragma unroll
          for (int t = 0; t < kDH; ++t) {
            dv[t] = fmaf(pdv, cols[(cc + t) * kCS + li], dv[t]);
            dk[t] = fmaf(dsv, cols[(qc + t) * kCS + li], dk[t]);
          }
        }
        wave_lds_sync();
#pragma unroll
        for (int t = 0; t < kDH; ++t) {                   // the head's q / k / v slices become dq / dk / dv
          my[(qc + t) * kCS] = lv * scale * dq[t];
          my[(kc + t) * kCS] = lv * scale * dk[t];
          my[(vc + t) * kCS] = lv * dv[t];
        }
        wave_lds_sync();
      }
    }
    // ---- qkv = Win h1 + bin,  h1 = LN1(xin)
    {
      const float mu = sv[kSvLn1 * 64], rstd = sv[(kSvLn1 + 1) * 64];
      const float* lw = wl + (wo.ln1w - l0) + f0;
      const float* lb = wl + (wo.ln1b - l0) + f0;
#pragma unroll
      for (int d = 0; d < FS; ++d) my[(RA + f0 + d) * kCS] = (sv[(kSvXin + f0 + d) * 64] - mu) * rstd * lw[d] + lb[d];
      __syncthreads();                                    // dqkv and h1 complete
      wgrad_tiles<3 * kTD, kTD>(cols, RQ, RA, slabw + wo.inw, slabw + wo.inb, lane, wave, 3 * kTD);
      gemm_cols<3 * kTD, kTD, kTD, false, false>(cols, RQ, wl + (wo.inw - l0), nullptr, RC, lane, wave);   // dh1 = dqkv Win
      __syncthreads();
      layer_norm_backward<kTD, FS>(sv + (kSvXin + f0) * 64, mu, rstd, lw, my + (RC + f0) * kCS, dx, my + (RQ + f0) * kCS,
                                   cols, RA, wave, lane);
      __syncthreads();
      if (l > 0) stage_dma(P + a.layer[l - 1].ln1w, wl, kLayerFloats, tid);   // next block's weights under the sums
      wgrad_tiles<2 * kTD, 0>(cols, RQ, 0, nullptr, slabw + wo.ln1w, lane, wave, 2 * kTD);
    }
    __syncthreads();
  }
  // ---- tokens = cat(cls, tokin) + pos
  if (live && i > 0) {
#pragma unroll
    for (int d = 0; d < FS; ++d) dtokin[(bc * a.N + (i - 1)) * kTD + f0 + d] = dx[d];
  }
#pragma unroll
  for (int d = 0; d < FS; ++d) my[(RA + f0 + d) * kCS] = dx[d];
  __syncthreads();
  for (int idx = tid; idx < a.n_pos * kTD; idx += kNW * 64) {
    const int ii = idx / kTD, d = idx - ii * kTD;
    float sum = 0.f;
    if (ii < S)
      for (int gg = 0; gg < G; ++gg) sum += cols[(RA + d) * kCS + gg * S + ii];
    slabw[a.pos + idx] = sum;
    if (ii == 0) slabw[a.cls + d] = sum;
  }
}

// grad[p] = sum over the waves' slabs, in slab order
__global__ __launch_bounds__(256) void tail_fused_reduce_kernel(const float* __restrict__ slab, float* __restrict__ grad,
                                                                int nw, int ptot) {
  __shared__ float part[4][64];
  const int p = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (p < ptot) {
    const float* src = slab + p;
    int w = q;
    for (; w + 12 < nw; w += 16) {
      s0 += src[(int64_t)w * ptot];
      s1 += src[(int64_t)(w + 4) * ptot];
      s2 += src[(int64_t)(w + 8) * ptot];
      s3 += src[(int64_t)(w + 12) * ptot];
    }
    for (; w < nw; w += 4) s0 += src[(int64_t)w * ptot];
  }
  part[q][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (q == 0 && p < ptot) grad[p] = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
}

}  // namespace isd

using namespace isd;

extern "C" int64_t isd_tail_fused_save_floats(int64_t B, int S, int D, int L) {
  if (B < 0 || S < 1 || S > kTMaxS || L < 1 || L > kTMaxL || (D != 16 && D != 32)) return ISD_ERR_INVALID;
  return (int64_t)L * (D == 32 ? Sv<32>::total : Sv<16>::total) * cdiv(B, (int64_t)(64 / S)) * 64;   // [layer][wave][field][lane]
}

// offsets of the tail's tensors inside the flat block, state_dict order (see TailMeta)
static void tail_meta_offsets(TailMeta& m, int n_tokens_p1, int D, int L, int n_cls) {
  int o = 0;
  m.pos = o; o += n_tokens_p1 * D;
  m.cls = o; o += D;
  for (int l = 0; l < L; ++l) {
    TailLayerOff& t = m.layer[l];
    t.ln1w = o; o += D; t.ln1b = o; o += D;
    t.inw = o; o += 3 * D * D; t.inb = o; o += 3 * D;
    t.ow = o; o += D * D; t.ob = o; o += D;
    t.ln2w = o; o += D; t.ln2b = o; o += D;
    t.w1 = o; o += 2 * D * D; t.b1 = o; o += 2 * D;
    t.w2 = o; o += 2 * D * D; t.b2 = o; o += D;
  }
  m.lastw = o; o += n_cls * D;
  m.lastb = o;
}

extern "C" int64_t isd_tail_fused_param_count(int n_tokens_p1, int D, int L, int n_cls) {
  if (n_tokens_p1 < 1 || L < 1 || L > kTMaxL || n_cls < 1) return ISD_ERR_INVALID;
  TailMeta m = {};
  tail_meta_offsets(m, n_tokens_p1, D, L, n_cls);
  return (int64_t)m.lastb + n_cls;
}

static int tail_fused_check(const char* who, int N, int S_table, int D, int H, int L, int hidden, int n_cls, int64_t B) {
  ISD_CHECK_ARG((D == 32 || D == 16) && hidden == 2 * D,
                "%s: dim_token=%d hidden=%d (dim_token 16 or 32, hidden = 2 dim_token)", who, D, hidden);
  ISD_CHECK_ARG(N >= 0 && N + 1 <= kTMaxS && N + 1 <= S_table,
                "%s: %d tokens per trial (at most %d with the cls token, positional table of %d)", who, N + 1, kTMaxS,
                S_table);
  ISD_CHECK_ARG(H >= 1 && (D == 4 * H || D == 8 * H), "%s: num_heads=%d (head width 4 or 8)", who, H);
  ISD_CHECK_ARG(L >= 1 && L <= kTMaxL && n_cls >= 1 && n_cls <= kTMaxCls, "%s: L=%d n_cls=%d", who, L, n_cls);
  ISD_CHECK_ARG(B >= 0, "%s: B=%lld", who, (long long)B);
  return ISD_OK;
}

extern "C" int isd_tail_fused_supported(int N, int D, int H, int L, int hidden, int n_cls) {
  return (D == 32 || D == 16) && hidden == 2 * D && N >= 0 && N + 1 <= kTMaxS && H >= 1 && (D == 4 * H || D == 8 * H) &&
         L >= 1 && L <= kTMaxL && n_cls >= 1 && n_cls <= kTMaxCls;
}

extern "C" int isd_tail_fused_forward(const float* params, const float* tokin, float* logits, float* save,
                                      float* xfinal, int64_t B, int N, int n_tokens_p1, int D, int H, int L,
                                      int hidden, int n_cls, float p_attn, float p_mlp, float p_cls,
                                      uint64_t seed, const uint64_t* seed_dev, void* stream) {
  int rc = tail_fused_check("isd_tail_fused_forward", N, n_tokens_p1, D, H, L, hidden, n_cls, B);
  if (rc) return rc;
  ISD_CHECK_ARG(p_attn >= 0.f && p_attn < 1.f && p_mlp >= 0.f && p_mlp < 1.f && p_cls >= 0.f && p_cls < 1.f,
                "isd_tail_fused_forward: dropout probabilities must lie in [0, 1)");
  if (B == 0) return ISD_OK;
  ISD_CHECK_ARG(params && logits && (N == 0 || tokin), "isd_tail_fused_forward: null argument");
  ISD_CHECK_ARG(((uintptr_t)params & 15) == 0, "isd_tail_fused_forward: the parameter block must be 16-byte aligned");
  TailMeta a = {};
  tail_meta_offsets(a, n_tokens_p1, D, L, n_cls);
  a.B = B; a.N = N; a.S = N + 1; a.H = H; a.L = L; a.n_cls = n_cls; a.n_pos = n_tokens_p1;
  a.p_attn = p_attn; a.p_mlp = p_mlp; a.p_cls = p_cls; a.seed = seed;
  a.seed_dev = (const unsigned long long*)seed_dev;
  const int G = 64 / a.S;
  const dim3 grid((unsigned)cdiv(B, G));
  ISD_CHECK_ARG((save && xfinal) || (!save && !xfinal && p_attn == 0.f && p_mlp == 0.f && p_cls == 0.f),
                "isd_tail_fused_forward: dropout needs the training record (save and xfinal)");
  const int key = D * 100 + (D / H) * 10 + (save ? 1 : 0);
#define ISD_TF_FWD(DD, DH, TR)                                                                                  \
  case DD * 100 + DH * 10 + TR:                                                                                  \
    hipLaunchKernelGGL((tail_fused_fwd_kernel<DD, DH, TR != 0>), grid, dim3(256), 0, (hipStream_t)stream, params, \
                       a, tokin, logits, save, xfinal);                                                          \
    break;
  switch (key) {
    ISD_TF_FWD(32, 4, 0) ISD_TF_FWD(32, 4, 1) ISD_TF_FWD(32, 8, 0) ISD_TF_FWD(32, 8, 1)
    ISD_TF_FWD(16, 4, 0) ISD_TF_FWD(16, 4, 1) ISD_TF_FWD(16, 8, 0) ISD_TF_FWD(16, 8, 1)
  }
#undef ISD_TF_FWD
  ISD_LAUNCH_CHECK();
  return ISD_OK;
}

extern "C" int64_t isd_tail_fused_workspace_floats(int64_t B, int N, int n_tokens_p1, int D, int L, int n_cls) {
  if (B < 0 || N < 0 || N + 1 > kTMaxS) return ISD_ERR_INVALID;
  const int64_t pc = isd_tail_fused_param_count(n_tokens_p1, D, L, n_cls);
  if (pc < 0) return pc;
  return cdiv(B, (int64_t)(64 / (N + 1))) * pc;
}

extern "C" int isd_tail_fused_backward(const float* params, const float* save, const float* xfinal,
                                       const float* dlogits, float* dtokin, float* dparams, float* workspace, int64_t B,
                                       int N, int n_tokens_p1, int D, int H, int L, int hidden, int n_cls, float p_attn,
                                       float p_mlp, float p_cls, uint64_t seed, const uint64_t* seed_dev,
                                       void* stream) {
  int rc = tail_fused_check("isd_tail_fused_backward", N, n_tokens_p1, D, H, L, hidden, n_cls, B);
  if (rc) return rc;
  ISD_CHECK_ARG(B > 0, "isd_tail_fused_backward: empty batch");
  ISD_CHECK_ARG(params && save && xfinal && dlogits && dparams && workspace && (N == 0 || dtokin),
                "isd_tail_fused_backward: null argument");
  ISD_CHECK_ARG(((uintptr_t)params & 15) == 0, "isd_tail_fused_backward: the parameter block must be 16-byte aligned");
  TailMeta a = {};
  tail_meta_offsets(a, n_tokens_p1, D, L, n_cls);
  a.B = B; a.N = N; a.S = N + 1; a.H = H; a.L = L; a.n_cls = n_cls; a.n_pos = n_tokens_p1;
  a.p_attn = p_attn; a.p_mlp = p_mlp; a.p_cls = p_cls; a.seed = seed;
  a.seed_dev = (const unsigned long long*)seed_dev;
  const int ptot = a.lastb + n_cls;
  const int nw = (int)cdiv(B, (int64_t)(64 / a.S));
#define ISD_TF_BWD(DD, DH)                                                                                        \
  case DD * 10 + DH:                                                                                              \
    hipLaunchKernelGGL((tail_fused_bwd_kernel<DD, DH>), dim3(nw), dim3(256), 0, (hipStream_t)stream, params, a, save, \
                       xfinal, dlogits, dtokin, workspace, ptot);                                                 \
    break;
  switch (D * 10 + D / H) { ISD_TF_BWD(32, 4) ISD_TF_BWD(32, 8) ISD_TF_BWD(16, 4) ISD_TF_BWD(16, 8) }
#undef ISD_TF_BWD
  ISD_LAUNCH_CHECK();
  hipLaunchKernelGGL(tail_fused_reduce_kernel, dim3(cdiv(ptot, 64)), dim3(256), 0, (hipStream_t)stream, workspace, dparams, nw, ptot);
  ISD_LAUNCH_CHECK();
  return ISD_OK;
}
