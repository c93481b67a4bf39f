// Transformer tail of FAST in ONE launch per direction (reference: src/fast/models/fast.py:10-29 `AttentionBlock`,
// :260-268 `forward_transformer`; the mode the reference trains, src/fast/train/trainer.py:58):
//   cls token + positional embedding -> L x [ x += MHA(LN1(x));  x += MLP(LN2(x)) ] -> last_layer(dropout(x[:, 0])).
// The sequence is <= 8 tokens of 32 features: per trial the whole tail is ~50 k multiply-adds per layer, and as
// separate launches (tail.hip + fc.hip: ~45 forward, ~60 backward launches) it is pure launch latency.
//
// Mapping: ONE LANE PER TOKEN.  A wave holds the tokens of floor(64 / S) trials; a token's 32-feature residual stream
// lives in the lane's registers for the whole kernel, every dense layer is a per-lane matrix-vector product whose
// weights arrive through the scalar cache (all lanes use the same weights: one SGPR operand per FMA), LayerNorm and
// GELU are lane-local, and the only cross-lane step -- a query reading the keys / values of the other tokens of its
// trial -- goes through a [feature][lane] LDS tile (a row of 64 consecutive dwords: conflict-free, and the lanes of one
// trial read one another's column by address).  Vector registers cannot be indexed at run time, so a dense layer
// keeps its INPUT vector in registers (static indices) and streams its OUTPUTS, eight at a time, into such LDS columns;
// the next layer loads them back into registers.  Dropout (attention probabilities, both MLP dropouts, the cls token)
// is counter-based: element e of site s of layer l draws from hash(seed, l, s, e), regenerated in the backward pass.
//
// The forward keeps what the backward needs (one record per layer and token, [layer][token][field]): LN statistics, qkv, attention
// probabilities, the MLP pre-activation and the two residual-stream snapshots.  The backward is one launch per layer:
// lane-local data gradients the same way (transposed weights are just the other loop order), and the weight
// gradients -- sums over tokens of outer products -- on the matrix cores: the wave's 64 token rows of each
// (gradient, input) pair meet in LDS and v_mfma_f32_16x16x4_f32 contracts over the tokens; persistent waves keep
// the layer's accumulators in registers and leave one partial slab each.
#include "common.h"
#include <math.h>

namespace isd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// D = dim_token (template parameter: 32 in production, 16 in the reference's small test configuration); the MLP hidden
// width is 2 D (fast.py:236)
constexpr int kTMaxS = 8;          // tokens per trial, cls included
constexpr int kTMaxL = 8;          // transformer blocks
constexpr int kTMaxCls = 16;

// saved per (layer, token): field offsets in floats
template <int D> struct Sv {
  static constexpr int xin = 0, ln1 = D, qkv = D + 2, prob = 4 * D + 2, ctx = prob + 8 * kTMaxS, xmid = ctx + D,
                       ln2 = xmid + D, hpre = ln2 + 2, total = hpre + 2 * D;
};

// The tail's parameters live in ONE flat block (the host packs them once; nn.FAST keeps them packed) and the kernels
// take it as a __restrict__ argument with integer offsets: pointers fetched from an argument structure carry no alias
// information, and the compiler then loads every weight with per-lane vector loads instead of through the scalar cache.
// Order inside the block = the reference's state_dict order of the tail:
//   pos_embedding [1, n_tokens + 1, D] | cls_token [1, 1, D] | per block: layer_norm_1.{weight,bias} |
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
};
struct TailLayerW {
  const float *ln1w, *ln1b, *inw, *inb, *ow, *ob, *ln2w, *ln2b, *w1, *b1, *w2, *b2;
};

__device__ __forceinline__ float tf_keep(unsigned long long seed, int layer, int site, unsigned long long e, float p) {
  if (p <= 0.f) return 1.f;
  unsigned long long v = (e + 0x9E3779B97F4A7C15ull * (unsigned long long)(layer * 8 + site + 1)) ^ seed;
  v ^= v >> 30; v *= 0xBF58476D1CE4E5B9ull;
  v ^= v >> 27; v *= 0x94D049BB133111EBull;
  v ^= v >> 31;
  const float uu = (float)(v >> 40) * (1.f / 16777216.f);
  return uu >= p ? 1.f / (1.f - p) : 0.f;
}

__device__ __forceinline__ float tf_gelu(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float tf_gelu_grad(float x) {
  return 0.5f * (1.f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * expf(-0.5f * x * x);
}

// y[o] = b[o] + sum_i W[o][i] in[i] for o in [0, NO): outputs to the LDS column  out[o * 64]  (this lane's slot)
template <int NI, int NO>
__device__ __forceinline__ void matvec_to_lds(const float (&in)[NI], const float* __restrict__ W,
                                              const float* __restrict__ b, float* out) {
#pragma unroll 1
  for (int oc = 0; oc < NO; oc += 2) {                    // rolled: the loop body is 2 x NI FMAs of code
    // two weight rows (2 NI scalars) in the SGPR file at a time, two independent FMA chains
    const float* w0 = W + oc * NI;
    const float* w1 = w0 + NI;
    float a0 = b[oc], a1 = b[oc + 1];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      a0 = fmaf(w0[i], in[i], a0);
      a1 = fmaf(w1[i], in[i], a1);
    }
    out[oc * 64] = a0;
    out[(oc + 1) * 64] = a1;
    __builtin_amdgcn_sched_barrier(0);
  }
}

// x[d] += scale_d * (b[d] + sum_i W[d][i] in_lds[i * 64]) with the input in an LDS column; NI inputs are loaded into
// registers first (static indices), outputs are produced 4 at a time by a rolled loop over register-resident x --
// which a rolled loop cannot index, so the update goes through an LDS column of x as well
template <int NI>
__device__ __forceinline__ void load_col(float (&v)[NI], const float* col) {
#pragma unroll
  for (int i = 0; i < NI; ++i) v[i] = col[i * 64];
}

// LayerNorm over the lane's D features (eps inside the sqrt, biased variance: nn.LayerNorm)
template <int D>
__device__ __forceinline__ void layer_norm(const float (&x)[D], const float* __restrict__ w,
                                           const float* __restrict__ b, float (&h)[D], float& mu, float& rstd) {
  float s = 0.f;
#pragma unroll
  for (int d = 0; d < D; ++d) s += x[d];
  mu = s * (1.f / D);
  float q = 0.f;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const float c = x[d] - mu;
    q = fmaf(c, c, q);
  }
  rstd = rsqrtf(q * (1.f / D) + 1e-5f);
#pragma unroll
  for (int d = 0; d < D; ++d) h[d] = (x[d] - mu) * rstd * w[d] + b[d];
}

// tokin [B][N][D]: output of input_layer (Linear + GELU); logits [B][n_cls]; save [L][B*S][Sv<D>::total] or null
// (inference); xfinal [B][D]: the cls token entering last_layer, after dropout (training) or null
template <int kTD>
__global__ __launch_bounds__(64) void tail_fused_fwd_kernel(const float* __restrict__ P, TailMeta a,
                                                            const float* __restrict__ tokin,
                                                            float* __restrict__ logits, float* __restrict__ save,
                                                            float* __restrict__ xfinal) {
  constexpr int kTHid = 2 * kTD;
  // LDS columns of a wave: [feature][lane]
  constexpr int kColQkv = 0, kColCtx = 3 * kTD, kColMid = 4 * kTD, kColX = 6 * kTD, kColTotal = 7 * kTD;
  constexpr int kSvXin = Sv<kTD>::xin, kSvLn1 = Sv<kTD>::ln1, kSvQkv = Sv<kTD>::qkv, kSvProb = Sv<kTD>::prob,
                kSvCtx = Sv<kTD>::ctx, kSvXmid = Sv<kTD>::xmid, kSvLn2 = Sv<kTD>::ln2, kSvHpre = Sv<kTD>::hpre,
                kSvTotal = Sv<kTD>::total;
  __shared__ float cols[kColTotal * 64];
  const int lane = threadIdx.x;
  const int S = a.S, G = 64 / S;                          // trials per wave
  const int g = lane / S, i = lane - g * S;
  const int64_t b = (int64_t)blockIdx.x * G + g;
  const bool live = g < G && b < a.B;
  const int gs = (g < G ? g : 0) * S;                     // first lane of this token's trial
  const int64_t bc = live ? b : 0;
  const int64_t M = a.B * S;
  const int64_t tok = bc * S + i;                         // global token row
  const int dh = kTD / a.H;
  const float scale = 1.f / sqrtf((float)dh);
  float* my = cols + lane;

  float x[kTD];
#pragma unroll
  for (int d = 0; d < kTD; ++d) {
    const float base = i == 0 ? P[a.cls + d] : tokin[(bc * a.N + (i - 1)) * kTD + d];
    x[d] = base + P[a.pos + i * kTD + d];
  }
  for (int l = 0; l < a.L; ++l) {
    const TailLayerOff& wo = a.layer[l];
    const TailLayerW w = {P + wo.ln1w, P + wo.ln1b, P + wo.inw, P + wo.inb, P + wo.ow, P + wo.ob,
                          P + wo.ln2w, P + wo.ln2b, P + wo.w1, P + wo.b1, P + wo.w2, P + wo.b2};
    // this token's record of layer l: [layer][token][field] -- one base address per lane, every field an immediate offset
    float* sv = (save && live) ? save + ((int64_t)l * M + tok) * kSvTotal : nullptr;
    float h[kTD], mu, rstd;
    if (sv) {
#pragma unroll
      for (int d = 0; d < kTD; ++d) sv[kSvXin + d] = x[d];
    }
    layer_norm<kTD>(x, w.ln1w, w.ln1b, h, mu, rstd);
    if (sv) { sv[kSvLn1] = mu; sv[kSvLn1 + 1] = rstd; }
    matvec_to_lds<kTD, 3 * kTD>(h, w.inw, w.inb, my + kColQkv * 64);            // q | k | v columns
    wave_lds_sync();
    if (sv) {
      for (int o = 0; o < 3 * kTD; ++o) sv[kSvQkv + o] = my[(kColQkv + o) * 64];
    }
    // attention: this token's query against the keys / values of its trial (lanes gs .. gs + S - 1)
    for (int hh = 0; hh < a.H; ++hh) {
      float qh[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) qh[t] = t < dh ? my[(kColQkv + hh * dh + t) * 64] : 0.f;
      float sc[kTMaxS];
      float mx = -INFINITY;
#pragma unroll
      for (int j = 0; j < kTMaxS; ++j) {
        float d0 = 0.f;
        if (j < S) {
#pragma unroll
          for (int t = 0; t < 8; ++t)
            if (t < dh) d0 = fmaf(qh[t], cols[(kColQkv + kTD + hh * dh + t) * 64 + gs + j], d0);
        }
        sc[j] = j < S ? d0 * scale : -INFINITY;
        mx = fmaxf(mx, sc[j]);
      }
      float den = 0.f;
#pragma unroll
      for (int j = 0; j < kTMaxS; ++j) {
        sc[j] = j < S ? expf(sc[j] - mx) : 0.f;
        den += sc[j];
      }
      float o[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) o[t] = 0.f;
#pragma unroll
      for (int j = 0; j < kTMaxS; ++j) {
        if (j < S) {
          const float p = sc[j] / den;
          if (sv) sv[kSvProb + hh * kTMaxS + j] = p;
          const float pd = p * tf_keep(a.seed, l, 0, (unsigned long long)((tok * a.H + hh) * kTMaxS + j), a.p_attn);
#pragma unroll
          for (int t = 0; t < 8; ++t)
            if (t < dh) o[t] = fmaf(pd, cols[(kColQkv + 2 * kTD + hh * dh + t) * 64 + gs + j], o[t]);
        }
      }
#pragma unroll
      for (int t = 0; t < 8; ++t)
        if (t < dh) my[(kColCtx + hh * dh + t) * 64] = o[t];
    }
    wave_lds_sync();
    {
      float ctx[kTD];
      load_col<kTD>(ctx, my + kColCtx * 64);
      if (sv) {
#pragma unroll
        for (int d = 0; d < kTD; ++d) sv[kSvCtx + d] = ctx[d];
      }
      matvec_to_lds<kTD, kTD>(ctx, w.ow, w.ob, my + kColX * 64);               // attention output projection
      wave_lds_sync();
#pragma unroll
      for (int d = 0; d < kTD; ++d) x[d] += my[(kColX + d) * 64];
    }
    if (sv) {
#pragma unroll
      for (int d = 0; d < kTD; ++d) sv[kSvXmid + d] = x[d];
    }
    layer_norm<kTD>(x, w.ln2w, w.ln2b, h, mu, rstd);
    if (sv) { sv[kSvLn2] = mu; sv[kSvLn2 + 1] = rstd; }
    wave_lds_sync();
    matvec_to_lds<kTD, kTHid>(h, w.w1, w.b1, my + kColMid * 64);
    wave_lds_sync();
    {
      float m[kTHid];
      load_col<kTHid>(m, my + kColMid * 64);
#pragma unroll
      for (int o = 0; o < kTHid; ++o) {
        if (sv) sv[kSvHpre + o] = m[o];
        m[o] = tf_gelu(m[o]) * tf_keep(a.seed, l, 1, (unsigned long long)(tok * kTHid + o), a.p_mlp);
      }
      matvec_to_lds<kTHid, kTD>(m, w.w2, w.b2, my + kColX * 64);
      wave_lds_sync();
#pragma unroll
      for (int d = 0; d < kTD; ++d)
        x[d] = fmaf(my[(kColX + d) * 64], tf_keep(a.seed, l, 2, (unsigned long long)(tok * kTD + d), a.p_mlp), x[d]);
    }
    wave_lds_sync();
  }
  if (live && i == 0) {
#pragma unroll
    for (int d = 0; d < kTD; ++d) {
      x[d] *= tf_keep(a.seed, kTMaxL, 0, (unsigned long long)(bc * kTD + d), a.p_cls);
      if (xfinal) xfinal[bc * kTD + d] = x[d];
    }
    for (int c = 0; c < a.n_cls; ++c) {
      float acc = P[a.lastb + c];
#pragma unroll
      for (int d = 0; d < kTD; ++d) acc = fmaf(P[a.lastw + c * kTD + d], x[d], acc);
      logits[bc * a.n_cls + c] = acc;
    }
  }
}

// ----------------------------------------------------------------------------------------------------- backward
// One launch for the whole tail, the lane <-> token mapping of the forward.  Data gradients are lane-local: the
// transposed products  d_in[i] = sum_o W[o][i] g[o]  run as a rolled loop over o with g[o] read from the lane's LDS
// column and row o of W (contiguous) in the scalar file, accumulating all d_in[i] in registers.  Weight gradients
// are sums over tokens of outer products: the wave's 64 token rows of g and of the layer input already sit in
// [feature][lane] LDS columns, which IS the A / B operand layout of v_mfma_f32_16x16x4_f32 with the token index as K
// (lane l supplies feature l & 15 of token 4 s + (l >> 4) in k-step s), so each 16 x 16 tile of dW is 16 MFMAs and the
// bias / LayerNorm-parameter sums are the same product against a vector of ones.  Each wave leaves its partial sums
// in its own slab (same layout as the parameter block); tail_fused_reduce_kernel adds the slabs in a fixed order.
// The column stride is 68 dwords: own-column accesses (lane-contiguous) and the MFMA operand reads
// ((l & 15) * 68 + (l >> 4) -> bank 4 (l & 15) + (l >> 4)) are both conflict-free.
constexpr int kBS = 68;

// out[o] = sum_r col[r] * W[r][o], r < NR (rows of W contiguous: NO scalars per row)
template <int NR, int NO>
__device__ __forceinline__ void matvecT_from_lds(const float* __restrict__ W, const float* col, float (&out)[NO]) {
#pragma unroll
  for (int o = 0; o < NO; ++o) out[o] = 0.f;
  constexpr int kRows = NO <= 32 ? 2 : 1;                 // weight rows in the SGPR file per iteration
#pragma unroll 1
  for (int r = 0; r < NR; r += kRows) {
    float g[kRows];
#pragma unroll
    for (int q = 0; q < kRows; ++q) g[q] = col[(r + q) * kBS];
#pragma unroll
    for (int q = 0; q < kRows; ++q) {
      const float* w = W + (r + q) * NO;
#pragma unroll
      for (int o = 0; o < NO; ++o) out[o] = fmaf(w[o], g[q], out[o]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// dW[o][i] = sum_tok g[tok][o] in[tok][i] (o < no_valid rows written), db[o] = sum_tok g[tok][o]; NI == 0: sums only
template <int NO, int NI>
__device__ __forceinline__ void wgrad_tiles(const float* cols, int gcol, int icol, float* __restrict__ dw,
                                            float* __restrict__ db, int lane, int no_valid) {
  const int m = lane & 15, kq = lane >> 4;
#pragma unroll 1
  for (int ob = 0; ob < NO; ob += 16) {
    float av[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) av[t] = cols[(gcol + ob + m) * kBS + 4 * t + kq];
    if (NI > 0) {
#pragma unroll 1
      for (int ib = 0; ib < NI; ib += 16) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 16; ++t)
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], cols[(icol + ib + m) * kBS + 4 * t + kq], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (ob + 4 * kq + r < no_valid) dw[(ob + 4 * kq + r) * NI + ib + m] = acc[r];
      }
    }
    f32x4 accb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 16; ++t) accb = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], 1.f, accb, 0, 0, 0);
    if (m == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (ob + 4 * kq + r < no_valid) db[ob + 4 * kq + r] = accb[r];
    }
  }
}

// LayerNorm backward for one token: x (the saved input), statistics, dh = gradient of the output.  Adds the input
// gradient to dx and leaves [dh * xhat | dh] in the 2 D columns at `pcol` (their sums over tokens are dweight | dbias).
template <int D>
__device__ __forceinline__ void layer_norm_backward(const float* __restrict__ xs, float mu, float rstd,
                                                    const float* __restrict__ w, const float (&dh)[D], float (&dx)[D],
                                                    float* pcol) {
  float c1 = 0.f, c2 = 0.f;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const float xh = (xs[d] - mu) * rstd, dxh = dh[d] * w[d];
    c1 += dxh;
    c2 = fmaf(dxh, xh, c2);
    pcol[d * kBS] = dh[d] * xh;
    pcol[(D + d) * kBS] = dh[d];
  }
  c1 *= 1.f / D;
  c2 *= 1.f / D;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const float xh = (xs[d] - mu) * rstd;
    dx[d] += rstd * (dh[d] * w[d] - c1 - xh * c2);
  }
}

// dlogits [B][n_cls] (already scaled by the caller's loss weight); dtokin [B][N][D]; slab [gridDim.x][ptot]
template <int kTD>
__global__ __launch_bounds__(64) void tail_fused_bwd_kernel(const float* __restrict__ P, TailMeta a,
                                                            const float* __restrict__ save,
                                                            const float* __restrict__ xfinal,
                                                            const float* __restrict__ dlogits,
                                                            float* __restrict__ dtokin, float* __restrict__ slab,
                                                            int ptot) {
  constexpr int kTHid = 2 * kTD;
  constexpr int RQ = 0, RA = 3 * kTD, RB = 4 * kTD, RC = 6 * kTD, kColTotal = 7 * kTD;   // column regions
  constexpr int kSvXin = Sv<kTD>::xin, kSvLn1 = Sv<kTD>::ln1, kSvQkv = Sv<kTD>::qkv, kSvProb = Sv<kTD>::prob,
                kSvCtx = Sv<kTD>::ctx, kSvXmid = Sv<kTD>::xmid, kSvLn2 = Sv<kTD>::ln2, kSvHpre = Sv<kTD>::hpre,
                kSvTotal = Sv<kTD>::total;
  static_assert(kTD >= 16 && RB + 16 <= RC, "column regions");
  __shared__ float cols[kColTotal * kBS];
  const int lane = threadIdx.x;
  const int S = a.S, G = 64 / S;
  const int g = lane / S, i = lane - g * S;
  const int64_t b = (int64_t)blockIdx.x * G + g;
  const bool live = g < G && b < a.B;
  const float lv = live ? 1.f : 0.f;
  const int gs = (g < G ? g : 0) * S;
  const int64_t bc = live ? b : 0;                        // dead lanes read trial 0's (finite) records, weighted by 0
  const int64_t M = a.B * S;
  const int64_t tok = bc * S + i;
  const int dh = kTD / a.H;
  const float scale = 1.f / sqrtf((float)dh);
  float* my = cols + lane;
  float* slabw = slab + (int64_t)blockIdx.x * ptot;

  // ---- last_layer: logits = W x0 + b on the cls token after dropout
  float dx[kTD];
  {
    const bool cl = live && i == 0;
#pragma unroll
    for (int c = 0; c < 16; ++c) my[(RA + c) * kBS] = (cl && c < a.n_cls) ? dlogits[bc * a.n_cls + c] : 0.f;
#pragma unroll
    for (int d = 0; d < kTD; ++d) my[(RC + d) * kBS] = cl ? xfinal[bc * kTD + d] : 0.f;
    wave_lds_sync();
    wgrad_tiles<16, kTD>(cols, RA, RC, slabw + a.lastw, slabw + a.lastb, lane, a.n_cls);
#pragma unroll
    for (int d = 0; d < kTD; ++d) dx[d] = 0.f;
    for (int c = 0; c < a.n_cls; ++c) {
      const float gl = my[(RA + c) * kBS];
      const float* w = P + a.lastw + c * kTD;
#pragma unroll
      for (int d = 0; d < kTD; ++d) dx[d] = fmaf(w[d], gl, dx[d]);
    }
#pragma unroll
    for (int d = 0; d < kTD; ++d)
      dx[d] *= tf_keep(a.seed, kTMaxL, 0, (unsigned long long)(bc * kTD + d), a.p_cls);
    wave_lds_sync();
  }

  for (int l = a.L - 1; l >= 0; --l) {
    const TailLayerOff& wo = a.layer[l];
    const float* sv = save + ((int64_t)l * M + tok) * kSvTotal;
    // ---- x_out = xmid + drop2(W2 m + b2),  m = drop1(gelu(hpre))
#pragma unroll
    for (int d = 0; d < kTD; ++d)
      my[(RA + d) * kBS] = lv * dx[d] * tf_keep(a.seed, l, 2, (unsigned long long)(tok * kTD + d), a.p_mlp);
#pragma unroll 2
    for (int o = 0; o < kTHid; ++o)
      my[(RB + o) * kBS] = tf_gelu(sv[kSvHpre + o]) * tf_keep(a.seed, l, 1, (unsigned long long)(tok * kTHid + o), a.p_mlp);
    wave_lds_sync();
    wgrad_tiles<kTD, kTHid>(cols, RA, RB, slabw + wo.w2, slabw + wo.b2, lane, kTD);
    {
      float dm[kTHid];
      matvecT_from_lds<kTD, kTHid>(P + wo.w2, my + RA * kBS, dm);
      wave_lds_sync();
#pragma unroll
      for (int o = 0; o < kTHid; ++o) my[(RB + o) * kBS] = dm[o];
    }
    wave_lds_sync();
#pragma unroll 2
    for (int o = 0; o < kTHid; ++o)                       // hpre = W1 h2 + b1: gradient through dropout and GELU
      my[(RB + o) * kBS] *= tf_gelu_grad(sv[kSvHpre + o]) * tf_keep(a.seed, l, 1, (unsigned long long)(tok * kTHid + o), a.p_mlp);
    {
      const float mu = sv[kSvLn2], rstd = sv[kSvLn2 + 1];
      const float* lw = P + wo.ln2w;
      const float* lb = P + wo.ln2b;
#pragma unroll
      for (int d = 0; d < kTD; ++d) my[(RC + d) * kBS] = (sv[kSvXmid + d] - mu) * rstd * lw[d] + lb[d];
      wave_lds_sync();
      wgrad_tiles<kTHid, kTD>(cols, RB, RC, slabw + wo.w1, slabw + wo.b1, lane, kTHid);
      float dhv[kTD];
      matvecT_from_lds<kTHid, kTD>(P + wo.w1, my + RB * kBS, dhv);
      layer_norm_backward<kTD>(sv + kSvXmid, mu, rstd, lw, dhv, dx, my + RQ * kBS);
      wave_lds_sync();
      wgrad_tiles<2 * kTD, 0>(cols, RQ, 0, nullptr, slabw + wo.ln2w, lane, 2 * kTD);
    }
    // ---- xmid = xin + Wo ctx + bo
#pragma unroll
    for (int d = 0; d < kTD; ++d) {
      my[(RA + d) * kBS] = dx[d];
      my[(RC + d) * kBS] = sv[kSvCtx + d];
    }
    wave_lds_sync();
    wgrad_tiles<kTD, kTD>(cols, RA, RC, slabw + wo.ow, slabw + wo.ob, lane, kTD);
    {
      float dctx[kTD];
      matvecT_from_lds<kTD, kTD>(P + wo.ow, my + RA * kBS, dctx);
      wave_lds_sync();
#pragma unroll
      for (int d = 0; d < kTD; ++d) my[(RC + d) * kBS] = dctx[d];
    }
#pragma unroll 4
    for (int o = 0; o < 3 * kTD; ++o) my[(RQ + o) * kBS] = sv[kSvQkv + o];
    wave_lds_sync();
    // ---- attention: ctx_i = sum_j drop(p_ij) v_j,  p = softmax(scale q k^T)
    for (int hh = 0; hh < a.H; ++hh) {
      const int qc = RQ + hh * dh, kc = RQ + kTD + hh * dh, vc = RQ + 2 * kTD + hh * dh, cc = RC + hh * dh;
      float dc[8], pr[kTMaxS], dp[kTMaxS];
#pragma unroll
      for (int t = 0; t < 8; ++t) dc[t] = t < dh ? my[(cc + t) * kBS] : 0.f;
      float dot = 0.f;
#pragma unroll
      for (int j = 0; j < kTMaxS; ++j) {
        pr[j] = 0.f;
        dp[j] = 0.f;
        if (j < S) {
          pr[j] = sv[kSvProb + hh * kTMaxS + j];
          const float kp = tf_keep(a.seed, l, 0, (unsigned long long)((tok * a.H + hh) * kTMaxS + j), a.p_attn);
          float dpd = 0.f;
#pragma unroll
          for (int t = 0; t < 8; ++t)
            if (t < dh) dpd = fmaf(dc[t], cols[(vc + t) * kBS + gs + j], dpd);
          dp[j] = dpd * kp;
          dot = fmaf(dp[j], pr[j], dot);
          my[(RB + j) * kBS] = pr[j] * kp;
        }
      }
      float dq[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) dq[t] = 0.f;
#pragma unroll
      for (int j = 0; j < kTMaxS; ++j) {
        if (j < S) {
          const float ds = pr[j] * (dp[j] - dot);
          my[(RB + 8 + j) * kBS] = ds;
#pragma unroll
          for (int t = 0; t < 8; ++t)
            if (t < dh) dq[t] = fmaf(ds, cols[(kc + t) * kBS + gs + j], dq[t]);
        }
      }
      wave_lds_sync();
      float dk[8], dv[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) dk[t] = dv[t] = 0.f;
#pragma unroll
      for (int ii = 0; ii < kTMaxS; ++ii) {
        if (ii < S) {                                     // row ii of the trial's probability matrix, this token's column
          const float pdv = cols[(RB + i) * kBS + gs + ii], dsv = cols[(RB + 8 + i) * kBS + gs + ii];
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            if (t < dh) {
              dv[t] = fmaf(pdv, cols[(cc + t) * kBS + gs + ii], dv[t]);
              dk[t] = fmaf(dsv, cols[(qc + t) * kBS + gs + ii], dk[t]);
            }
          }
        }
      }
      wave_lds_sync();
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        if (t < dh) {                                     // the head's q / k / v slices become dq / dk / dv
          my[(qc + t) * kBS] = lv * scale * dq[t];
          my[(kc + t) * kBS] = lv * scale * dk[t];
          my[(vc + t) * kBS] = lv * dv[t];
        }
      }
      wave_lds_sync();
    }
    // ---- qkv = Win h1 + bin,  h1 = LN1(xin)
    {
      const float mu = sv[kSvLn1], rstd = sv[kSvLn1 + 1];
      const float* lw = P + wo.ln1w;
      const float* lb = P + wo.ln1b;
#pragma unroll
      for (int d = 0; d < kTD; ++d) my[(RA + d) * kBS] = (sv[kSvXin + d] - mu) * rstd * lw[d] + lb[d];
      wave_lds_sync();
      wgrad_tiles<3 * kTD, kTD>(cols, RQ, RA, slabw + wo.inw, slabw + wo.inb, lane, 3 * kTD);
      float dhv[kTD];
      matvecT_from_lds<3 * kTD, kTD>(P + wo.inw, my + RQ * kBS, dhv);
      layer_norm_backward<kTD>(sv + kSvXin, mu, rstd, lw, dhv, dx, my + RB * kBS);
      wave_lds_sync();
      wgrad_tiles<2 * kTD, 0>(cols, RB, 0, nullptr, slabw + wo.ln1w, lane, 2 * kTD);
    }
    wave_lds_sync();
  }
  // ---- tokens = cat(cls, tokin) + pos
  if (live && i > 0) {
#pragma unroll
    for (int d = 0; d < kTD; ++d) dtokin[(bc * a.N + (i - 1)) * kTD + d] = dx[d];
  }
#pragma unroll
  for (int d = 0; d < kTD; ++d) my[(RA + d) * kBS] = dx[d];
  wave_lds_sync();
  for (int idx = lane; idx < a.n_pos * kTD; idx += 64) {
    const int ii = idx / kTD, d = idx - ii * kTD;
    float sum = 0.f;
    if (ii < S)
      for (int gg = 0; gg < G; ++gg) sum += cols[(RA + d) * kBS + gg * S + ii];
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
  return (int64_t)L * (D == 32 ? Sv<32>::total : Sv<16>::total) * B * S;
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
  ISD_CHECK_ARG(H >= 1 && D % H == 0 && D / H <= 8, "%s: num_heads=%d", who, H);
  ISD_CHECK_ARG(L >= 1 && L <= kTMaxL && n_cls >= 1 && n_cls <= kTMaxCls, "%s: L=%d n_cls=%d", who, L, n_cls);
  ISD_CHECK_ARG(B >= 0, "%s: B=%lld", who, (long long)B);
  return ISD_OK;
}

extern "C" int isd_tail_fused_supported(int N, int D, int H, int L, int hidden, int n_cls) {
  return (D == 32 || D == 16) && hidden == 2 * D && N >= 0 && N + 1 <= kTMaxS && H >= 1 && D % H == 0 && D / H <= 8 &&
         L >= 1 && L <= kTMaxL && n_cls >= 1 && n_cls <= kTMaxCls;
}

extern "C" int isd_tail_fused_forward(const float* params, const float* tokin, float* logits, float* save,
                                      float* xfinal, int64_t B, int N, int n_tokens_p1, int D, int H, int L,
                                      int hidden, int n_cls, float p_attn, float p_mlp, float p_cls,
                                      uint64_t seed, void* stream) {
  int rc = tail_fused_check("isd_tail_fused_forward", N, n_tokens_p1, D, H, L, hidden, n_cls, B);
  if (rc) return rc;
  ISD_CHECK_ARG(p_attn >= 0.f && p_attn < 1.f && p_mlp >= 0.f && p_mlp < 1.f && p_cls >= 0.f && p_cls < 1.f,
                "isd_tail_fused_forward: dropout probabilities must lie in [0, 1)");
  if (B == 0) return ISD_OK;
  ISD_CHECK_ARG(params && logits && (N == 0 || tokin), "isd_tail_fused_forward: null argument");
  TailMeta a = {};
  tail_meta_offsets(a, n_tokens_p1, D, L, n_cls);
  a.B = B; a.N = N; a.S = N + 1; a.H = H; a.L = L; a.n_cls = n_cls; a.n_pos = n_tokens_p1;
  a.p_attn = p_attn; a.p_mlp = p_mlp; a.p_cls = p_cls; a.seed = seed;
  const int G = 64 / a.S;
  const dim3 grid((unsigned)cdiv(B, G));
  if (D == 32) hipLaunchKernelGGL(tail_fused_fwd_kernel<32>, grid, dim3(64), 0, (hipStream_t)stream, params, a, tokin, logits, save, xfinal);
  else hipLaunchKernelGGL(tail_fused_fwd_kernel<16>, grid, dim3(64), 0, (hipStream_t)stream, params, a, tokin, logits, save, xfinal);
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
                                       float p_mlp, float p_cls, uint64_t seed, void* stream) {
  int rc = tail_fused_check("isd_tail_fused_backward", N, n_tokens_p1, D, H, L, hidden, n_cls, B);
  if (rc) return rc;
  ISD_CHECK_ARG(B > 0, "isd_tail_fused_backward: empty batch");
  ISD_CHECK_ARG(params && save && xfinal && dlogits && dparams && workspace && (N == 0 || dtokin),
                "isd_tail_fused_backward: null argument");
  TailMeta a = {};
  tail_meta_offsets(a, n_tokens_p1, D, L, n_cls);
  a.B = B; a.N = N; a.S = N + 1; a.H = H; a.L = L; a.n_cls = n_cls; a.n_pos = n_tokens_p1;
  a.p_attn = p_attn; a.p_mlp = p_mlp; a.p_cls = p_cls; a.seed = seed;
  const int ptot = a.lastb + n_cls;
  const int nw = (int)cdiv(B, (int64_t)(64 / a.S));
  if (D == 32) hipLaunchKernelGGL(tail_fused_bwd_kernel<32>, dim3(nw), dim3(64), 0, (hipStream_t)stream, params, a, save, xfinal, dlogits, dtokin, workspace, ptot);
  else hipLaunchKernelGGL(tail_fused_bwd_kernel<16>, dim3(nw), dim3(64), 0, (hipStream_t)stream, params, a, save, xfinal, dlogits, dtokin, workspace, ptot);
  ISD_LAUNCH_CHECK();
  hipLaunchKernelGGL(tail_fused_reduce_kernel, dim3(cdiv(ptot, 64)), dim3(256), 0, (hipStream_t)stream, workspace, dparams, nw, ptot);
  ISD_LAUNCH_CHECK();
  return ISD_OK;
}
