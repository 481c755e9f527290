// dfx_numparse.hpp -- text -> number with Rust `str::parse::<T>()` semantics, usable on the device (hipcc) and on
// the host (g++: tests/numparse_fuzz.cpp checks it against glibc strtod, which is correctly rounded like Rust's
// dec2flt).  This is what arrow 0.12's csv reader runs per cell (`s.parse::<T::Native>()`), i.e. the arithmetic
// behind CsvDataSource (src/execution/datasource.rs:33-58).
//
// f64 / f32: grammar of core::num::dec2flt (Rust 2019): [+-]? ( digits [. digits?]? | . digits ) ( [eE] [+-]? digits )?
// or exactly "inf" / "NaN" after the optional sign.  Conversion: Eisel-Lemire (128-bit product with a truncated
// power of five, tools/gen_pow5_table.py) -- integer arithmetic only, so host and device agree bit for bit.
// The algorithm declines (returns NP_UNSUPPORTED) only when it cannot prove the rounding: > 19 significant digits
// whose truncation changes the result, or a product that sits within one unit of a rounding boundary outside
// the safe exponent range; Rust then runs an arbitrary-precision path that this library does not have.
// Integers: [+-]? digits (unsigned: '-' is an error even for "-0"), overflow is an error.
#pragma once
#include <stdint.h>

#include "dfx_pow5_table.hpp"

#if defined(__HIPCC__)
#define DFX_NP __device__ inline
#else
#define DFX_NP inline
#endif

namespace dfx {

enum : int { NP_OK = 0, NP_INVALID = 1, NP_UNSUPPORTED = 2 };

struct BiasedFp {
  uint64_t f;
  int32_t e;  // biased exponent; < 0: the algorithm declined
};

DFX_NP int np_clz64(uint64_t x) {
#if defined(__HIPCC__)
  return __clzll((long long)x);
#else
  return __builtin_clzll(x);
#endif
}

DFX_NP void np_mul64(uint64_t a, uint64_t b, uint64_t* lo, uint64_t* hi) {
#if defined(__HIPCC__)
  *lo = a * b;
  *hi = __umul64hi(a, b);
#else
  const unsigned __int128 p = (unsigned __int128)a * b;
  *lo = (uint64_t)p;
  *hi = (uint64_t)(p >> 64);
#endif
}

template <int MANT_BITS, int MIN_EXP, int INF_POWER, int MIN_RTE, int MAX_RTE, int SMALLEST_P10, int LARGEST_P10>
DFX_NP BiasedFp np_compute_float(int64_t q, uint64_t w) {
  BiasedFp zero = {0, 0}, inf = {0, INF_POWER}, err = {0, -1};
  if (w == 0 || q < SMALLEST_P10) return zero;
  if (q > LARGEST_P10) return inf;
  const int lz = np_clz64(w);
  w <<= lz;
  // w * 5^q, 128-bit approximation with (MANT_BITS + 3) bits of precision
  const uint64_t mask = 0xFFFFFFFFFFFFFFFFull >> (MANT_BITS + 3);
  const int idx = 2 * (int)(q - kPow5Smallest);
  uint64_t lo, hi;
  np_mul64(w, kPow5Table[idx], &lo, &hi);
  if ((hi & mask) == mask) {
    uint64_t lo2, hi2;
    np_mul64(w, kPow5Table[idx + 1], &lo2, &hi2);
    lo += hi2;
    if (hi2 > lo) ++hi;
  }
  if (lo == 0xFFFFFFFFFFFFFFFFull && !(q >= -27 && q <= 55)) return err;
  const int upperbit = (int)(hi >> 63);
  uint64_t mantissa = hi >> (upperbit + 64 - MANT_BITS - 3);
  int32_t power2 = (int32_t)((((int32_t)q * (152170 + 65536)) >> 16) + 63) + upperbit - lz - MIN_EXP;
  if (power2 <= 0) {  // subnormal
    if (-power2 + 1 >= 64) return zero;
    mantissa >>= -power2 + 1;
    mantissa += mantissa & 1;
    mantissa >>= 1;
    power2 = mantissa >= (1ull << MANT_BITS) ? 1 : 0;
    BiasedFp r = {mantissa, power2};
    return r;
  }
  // round to even when the product is exact and lands on a tie
  if (lo <= 1 && q >= MIN_RTE && q <= MAX_RTE && (mantissa & 3) == 1 &&
      (mantissa << (upperbit + 64 - MANT_BITS - 3)) == hi)
    mantissa &= ~1ull;
  mantissa += mantissa & 1;
  mantissa >>= 1;
  if (mantissa >= (2ull << MANT_BITS)) {
    mantissa = 1ull << MANT_BITS;
    ++power2;
  }
  mantissa &= ~(1ull << MANT_BITS);
  if (power2 >= INF_POWER) return inf;
  BiasedFp r = {mantissa, power2};
  return r;
}

DFX_NP bool np_is_digit(uint8_t c) { return (uint8_t)(c - '0') < 10; }

// decimal literal -> (negative, w, q, many_digits); special: 1 inf, 2 NaN.  Returns false on a grammar error.
DFX_NP bool np_scan_decimal(const uint8_t* s, int64_t n, bool* neg, uint64_t* w_out, int64_t* q_out, bool* many,
                            int* special) {
  *neg = false;
  *special = 0;
  *many = false;
  int64_t i = 0;
  if (n == 0) return false;
  if (s[0] == '-' || s[0] == '+') {
    *neg = s[0] == '-';
    i = 1;
    if (n == 1) return false;
  }
  if (n - i == 3 && s[i] == 'i' && s[i + 1] == 'n' && s[i + 2] == 'f') {
    *special = 1;
    return true;
  }
  if (n - i == 3 && s[i] == 'N' && s[i + 1] == 'a' && s[i + 2] == 'N') {
    *special = 2;
    return true;
  }
  uint64_t w = 0;
  int64_t digits = 0;        // significant digits accumulated in w (<= 19)
  int64_t dropped_int = 0;   // integer-part digits beyond the 19th
  bool nonzero_dropped = false;
  bool seen_nonzero = false;
  int64_t n_int = 0, n_frac = 0, frac_used = 0;
  for (; i < n && np_is_digit(s[i]); ++i, ++n_int) {
    const uint8_t d = (uint8_t)(s[i] - '0');
    if (d != 0) seen_nonzero = true;
    if (!seen_nonzero) continue;  // leading zeros carry no information
    if (digits < 19) {
      w = w * 10 + d;
      ++digits;
    } else {
      ++dropped_int;
      if (d != 0) nonzero_dropped = true;
    }
  }
  if (i < n && s[i] == '.') {
    ++i;
    for (; i < n && np_is_digit(s[i]); ++i, ++n_frac) {
      const uint8_t d = (uint8_t)(s[i] - '0');
      if (d != 0) seen_nonzero = true;
      if (!seen_nonzero) {  // 0.000ddd: zeros after the point shift the exponent
        ++frac_used;
        continue;
      }
      if (digits < 19) {
        w = w * 10 + d;
        ++digits;
        ++frac_used;
      } else if (d != 0) {
        nonzero_dropped = true;
      }
    }
  }
  if (n_int + n_frac == 0) return false;
  int64_t exp10 = 0;
  if (i < n && (s[i] == 'e' || s[i] == 'E')) {
    ++i;
    bool eneg = false;
    if (i < n && (s[i] == '-' || s[i] == '+')) {
      eneg = s[i] == '-';
      ++i;
    }
    if (i >= n || !np_is_digit(s[i])) return false;
    for (; i < n && np_is_digit(s[i]); ++i) {
      if (exp10 < 100000) exp10 = exp10 * 10 + (s[i] - '0');  // saturate: anything this large is 0 or inf
    }
    if (eneg) exp10 = -exp10;
  }
  if (i != n) return false;
  *w_out = w;
  *q_out = exp10 + dropped_int - frac_used;
  *many = nonzero_dropped;
  return true;
}

DFX_NP int np_parse_f64(const uint8_t* s, int64_t n, double* out) {
  bool neg, many;
  uint64_t w;
  int64_t q;
  int special;
  if (!np_scan_decimal(s, n, &neg, &w, &q, &many, &special)) return NP_INVALID;
  uint64_t bits;
  if (special == 1) {
    bits = 0x7FF0000000000000ull;
  } else if (special == 2) {
    bits = 0x7FF8000000000000ull;
  } else {
    BiasedFp fp = np_compute_float<52, -1023, 0x7FF, -4, 23, -342, 308>(q, w);
    if (many && fp.e >= 0) {
      const BiasedFp fp2 = np_compute_float<52, -1023, 0x7FF, -4, 23, -342, 308>(q, w + 1);
      if (fp2.e != fp.e || fp2.f != fp.f) fp.e = -1;
    }
    if (fp.e < 0) return NP_UNSUPPORTED;
    bits = fp.f | ((uint64_t)fp.e << 52);
  }
  if (neg) bits |= 0x8000000000000000ull;
  union {
    uint64_t u;
    double d;
  } c;
  c.u = bits;
  *out = c.d;
  return NP_OK;
}

DFX_NP int np_parse_f32(const uint8_t* s, int64_t n, float* out) {
  bool neg, many;
  uint64_t w;
  int64_t q;
  int special;
  if (!np_scan_decimal(s, n, &neg, &w, &q, &many, &special)) return NP_INVALID;
  uint32_t bits;
  if (special == 1) {
    bits = 0x7F800000u;
  } else if (special == 2) {
    bits = 0x7FC00000u;
  } else {
    BiasedFp fp = np_compute_float<23, -127, 0xFF, -17, 10, -65, 38>(q, w);
    if (many && fp.e >= 0) {
      const BiasedFp fp2 = np_compute_float<23, -127, 0xFF, -17, 10, -65, 38>(q, w + 1);
      if (fp2.e != fp.e || fp2.f != fp.f) fp.e = -1;
    }
    if (fp.e < 0) return NP_UNSUPPORTED;
    bits = (uint32_t)fp.f | ((uint32_t)fp.e << 23);
  }
  if (neg) bits |= 0x80000000u;
  union {
    uint32_t u;
    float f;
  } c;
  c.u = bits;
  *out = c.f;
  return NP_OK;
}

// [+-]? digits -> two's complement value of a signed type with the given number of bits (8..64)
DFX_NP int np_parse_int(const uint8_t* s, int64_t n, int bits, bool is_signed, uint64_t* out) {
  if (n == 0) return NP_INVALID;
  int64_t i = 0;
  bool neg = false;
  if (s[0] == '+' || s[0] == '-') {
    neg = s[0] == '-';
    if (neg && !is_signed) return NP_INVALID;  // u*::from_str rejects '-'
    i = 1;
    if (n == 1) return NP_INVALID;
  }
  const uint64_t maxpos = is_signed ? ((1ull << (bits - 1)) - 1) : (bits == 64 ? ~0ull : ((1ull << bits) - 1));
  const uint64_t limit = neg ? maxpos + 1 : maxpos;  // magnitude limit
  uint64_t v = 0;
  for (; i < n; ++i) {
    if (!np_is_digit(s[i])) return NP_INVALID;
    const uint64_t d = (uint64_t)(s[i] - '0');
    if (v > (limit - d) / 10) return NP_INVALID;  // overflow
    v = v * 10 + d;
  }
  *out = neg ? (uint64_t)(0 - v) : v;
  return NP_OK;
}

}  // namespace dfx
