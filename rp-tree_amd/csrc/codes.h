// codes.h — 16-bit order-preserving codes of projection values.
//
// The median split of a level needs, for almost every point, only the BIN of its key (which
// side of the node's pivot bin it falls on); the exact key matters for the ~1-2 % of points in
// the pivot bin and for the margins.  A projection kernel can therefore emit, next to the key
// P[c][i], a 2-byte code  C[c][i] = clamp(floor((P[c][i] - a_c) * s_c), 0, 65535)  per column c:
// the map is weakly monotone (floating-point subtract, multiply by s_c > 0 and floor never
// reverse an order; clamping neither), so  code(x) < code(y)  implies  x < y  and the streaming
// split may histogram and classify on codes (4 bytes per point and level instead of 10-12) and
// fetch exact keys only where they decide something.  (a_c, s_c) come from the minimum and
// maximum of the column over a SAMPLE of the rows: they only shape the bins — values outside the
// sample's range clamp into the edge codes — never the result.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rpt {

// ---- ordered-integer image of a floating key (for atomicMax / atomicMin) ----------------
__device__ inline unsigned long long ord_of(double v) {
  unsigned long long u = (unsigned long long)__double_as_longlong(v);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ULL);
}
__device__ inline double ord_to(unsigned long long u, double) {
  u = (u >> 63) ? (u & 0x7fffffffffffffffULL) : ~u;
  return __longlong_as_double((long long)u);
}
__device__ inline unsigned long long ord_of(float v) {
  unsigned int u = __float_as_uint(v);
  u = (u >> 31) ? ~u : (u | 0x80000000u);
  return (unsigned long long)u;
}
__device__ inline float ord_to(unsigned long long w, float) {
  unsigned int u = (unsigned int)w;
  u = (u >> 31) ? (u & 0x7fffffffu) : ~u;
  return __uint_as_float(u);
}

// Second, optional output of a projection batch (project_columns): codes[c][i] for the columns c
// of the batch whose level (c % L) is below Lc, row stride ld.  ONE geometry serves all columns:
// mm[0..1] = ordered images (ord_of) of the minimum and maximum of a sample of those columns'
// values.  (Per-column geometries would use the 16 bits a little better — hyperplanes of one
// forest have similar norms, the ranges differ by small factors — but cost an LDS fetch per
// value in kernels that run at their register limit: measured +18 % on the f64 MFMA kernel,
// against +3 % for the single pair, which lives in four registers.)
struct CodeOut {
  uint16_t* codes = nullptr;
  const unsigned long long* mm = nullptr;
  int64_t ld = 0;
  int L = 1, Lc = 0;
};

// code = clamp(floor(fma(v, s, b)), 0, 65535): weakly monotone in v (s > 0; one rounding of a
// monotone function; the conversion saturates)
template <class TC>
struct CodeGeo {
  TC s, b;
};

template <class TC>
__device__ inline CodeGeo<TC> code_geo(unsigned long long omin, unsigned long long omax) {
  CodeGeo<TC> g;
  g.s = (TC)1;
  g.b = (TC)0;
  if (omin <= omax) {  // the initial pair is (~0, 0): nothing sampled
    const TC lo = ord_to(omin, TC()), hi = ord_to(omax, TC());
    // one value in the sample (or a range that overflows): codes degenerate to 0 / 65535, the
    // split then sees one huge pivot bin and takes its general path
    const TC sc = (hi > lo && (hi - lo) < (TC)3e38) ? (TC)65535 / (hi - lo) : (TC)1;
    g.s = sc;
    g.b = -lo * sc;
    if (!(g.b == g.b) || g.b > (TC)3e38 || g.b < (TC)-3e38) {  // lo * sc overflowed
      g.s = (TC)1;
      g.b = (TC)0;
    }
  }
  return g;
}

// v_cvt_u32_f64 / v_cvt_u32_f32 saturate (negative and NaN -> 0, large -> 0xffffffff); the
// instruction is named explicitly because an out-of-range C++ conversion is undefined.  Three
// VALU instructions per code: the MFMA kernels' epilogues are not free (see CodeOut).
__device__ inline uint16_t code_of(double v, const CodeGeo<double>& g) {
  const double u = __builtin_fma(v, g.s, g.b);
  unsigned int c;
  asm("v_cvt_u32_f64 %0, %1" : "=v"(c) : "v"(u));
  return (uint16_t)(c < 65535u ? c : 65535u);
}
__device__ inline uint16_t code_of(float v, const CodeGeo<float>& g) {
  const float u = __builtin_fmaf(v, g.s, g.b);
  unsigned int c;
  asm("v_cvt_u32_f32 %0, %1" : "=v"(c) : "v"(u));
  return (uint16_t)(c < 65535u ? c : 65535u);
}

}  // namespace rpt
