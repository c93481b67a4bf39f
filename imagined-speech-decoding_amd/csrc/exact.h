// Order-independent accumulation of fp32 partial sums (BatchNorm statistics and the gradient sums of the BatchNorm
// heads, csrc/eegnet.hip and csrc/paperhead.hip).
//
// Thousands of workgroups each hold an fp32 partial of a batch sum.  Added with floating-point atomics the result
// depends on the order the workgroups arrive in: two runs on the same inputs differ in the last bits, and AdamW
// turns last bits of a near-zero gradient into full-size steps (the reference fixes its run-to-run behaviour with
// cudnn.deterministic, src/fast/utils.py:104-114).  Here every partial is added EXACTLY: an fp32 number is an
// integer multiple of 2^-149 below 2^128, so it is a 278-bit fixed-point number with at most 24 significant bits;
// the accumulator keeps that fixed-point value in nine base-2^32 digits held in 64-bit words (31 bits of carry
// headroom per digit: 2^31 additions) and a partial touches at most two neighbouring digits -- two 64-bit INTEGER
// atomics, which commute.  The sum is therefore the exact sum of the partials whatever the arrival order, and the
// same bits on every run; reading it rounds once to fp64.  Integer words also add exactly across ranks: the
// synchronised-BatchNorm exchange all-reduces them as int64.
#pragma once
#include <hip/hip_runtime.h>

namespace isd {

struct ExactAcc {
  long long w[10];                   // w[i], i < 9: digit of weight 2^(32 i - 149);  w[9]: count of non-finite partials
};

__device__ __forceinline__ void exact_add(ExactAcc* a, float v) {
  const int bits = __float_as_int(v);
  const int e = (bits >> 23) & 0xff;
  if (e == 0xff) {                                               // inf / nan: the sum reads back as nan
    atomicAdd((unsigned long long*)&a->w[9], 1ull);
    return;
  }
  long long m = (long long)((bits & 0x7fffff) | (e ? 0x800000 : 0));
  if (bits < 0) m = -m;
  const int p = (e ? e : 1) - 1;                                 // v = m 2^(p - 149), 0 <= p <= 253
  const int d = p >> 5;
  const long long t = m << (p & 31);                             // |t| < 2^55
  const long long lo = t & 0xffffffffll, hi = t >> 32;           // t = hi 2^32 + lo, 0 <= lo < 2^32 (floor split)
  if (lo) atomicAdd((unsigned long long*)&a->w[d], (unsigned long long)lo);
  if (hi) atomicAdd((unsigned long long*)&a->w[d + 1], (unsigned long long)hi);
}

// the accumulated value, rounded to fp64 (relative error < 2^-50; a pure function of the words)
__device__ __forceinline__ double exact_get(const ExactAcc* a) {
  if (a->w[9] != 0) return __longlong_as_double(0x7ff8000000000000ll);
  long long dgt[9];
  long long carry = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {                                  // carry-normalise: digits 0..7 into [0, 2^32)
    const long long v = a->w[i] + carry;
    carry = v >> 32;
    dgt[i] = v & 0xffffffffll;
  }
  dgt[8] = a->w[8] + carry;
  const bool neg = dgt[8] < 0;
  if (neg) {                                                     // magnitude of a negative total: negate digit-wise
    long long c = 1;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const long long v = (0xffffffffll - dgt[i]) + c;
      c = v >> 32;
      dgt[i] = v & 0xffffffffll;
    }
    dgt[8] = -dgt[8] - 1 + c;
  }
  double r = (double)dgt[8];
#pragma unroll
  for (int i = 7; i >= 0; --i) r = r * 4294967296.0 + (double)dgt[i];
  r = ldexp(r, -149);
  return neg ? -r : r;
}

}  // namespace isd
