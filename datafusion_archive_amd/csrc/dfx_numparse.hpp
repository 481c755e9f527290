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
    if (v > 1844674407370955161ull) return NP_INVALID;  // 10 v would not fit 64 bits
    const uint64_t t = v * 10;
    v = t + d;
    if (v < t) return NP_INVALID;
  }
  if (v > limit) return NP_INVALID;  // overflow: the magnitude only grows digit by digit, one check at the end decides
  *out = neg ? (uint64_t)(0 - v) : v;
  return NP_OK;
}

// ---- word-at-a-time variants ---------------------------------------------------------------------------
// Same functions, same results; the caller guarantees that the 8 bytes after the cell, s[n .. n + 8), may be READ (their
// values are ignored) -- the CSV cell kernel converts out of an LDS copy of the text, where that holds.  A cell is taken
// eight bytes at a time: the length of the digit run comes from a SWAR mask, its value from three multiplications.  The
// short forms cover [+-]? digits [. digits] ([eE] [+-]? 1-5 digits)? with at most 19 digits in all (then the integer of
// all digits fits 64 bits and nothing is dropped); everything else -- inf, NaN, longer literals, errors -- goes through
// the byte-at-a-time scanner above, so acceptance and values cannot differ (tests/native/numparse_fuzz.cpp holds both
// against each other on every input).
DFX_NP uint64_t np_load8(const uint8_t* p) {
  uint64_t w;
  __builtin_memcpy(&w, p, 8);
  return w;
}

// how many of the first (lowest-address) bytes of w are ASCII digits, 0..8.  A digit has high nibble 3 and, after adding 6,
// still high nibble 3; a carry out of a byte >= 0xFA only reaches LATER bytes, which a non-digit makes irrelevant.
DFX_NP int np_digit_run(uint64_t w) {
  const uint64_t hi = 0xF0F0F0F0F0F0F0F0ull, three = 0x3030303030303030ull;
  const uint64_t nd = ((w & hi) ^ three) | (((w + 0x0606060606060606ull) & hi) ^ three);
  if (nd == 0) return 8;
#if defined(__HIPCC__)
  return (__ffsll((long long)nd) - 1) >> 3;
#else
  return __builtin_ctzll(nd) >> 3;
#endif
}

DFX_NP uint32_t np_four_digits(uint32_t v) {  // bytes d0 d1 d2 d3 (d0 lowest, 0..9 or ASCII) -> d0 d1 d2 d3 as a number
  v = ((v & 0x0F0F0F0Fu) * 2561u) >> 8;
  return ((v & 0x00FF00FFu) * 6553601u) >> 16;
}

// value of the first len (1..8) bytes of w, all digits
DFX_NP uint32_t np_digits_value(uint64_t w, int len) {
  const uint64_t x = w << (8 * (8 - len));  // the digits move to the high end; zero bytes below them read as leading zeros
  return np_four_digits((uint32_t)x) * 10000u + np_four_digits((uint32_t)(x >> 32));
}

DFX_NP uint32_t np_pow10_small(int len) {  // 10^len, len 0..8
  return ((len & 1) ? 10u : 1u) * ((len & 2) ? 100u : 1u) * ((len & 4) ? 10000u : 1u) * ((len & 8) ? 100000000u : 1u);
}

// consumes a run of digits at s[*i ..): acc = acc 10^k + value, *nd += k.  Stops after 24 digits (more than any short form has).
DFX_NP void np_take_digits(const uint8_t* s, int64_t n, int64_t* i, uint64_t* acc, int* nd) {
  for (int k = 0; k < 3 && *i < n; ++k) {
    const uint64_t w = np_load8(s + *i);
    int len = np_digit_run(w);
    const int64_t left = n - *i;
    if ((int64_t)len > left) len = (int)left;
    if (len == 0) return;
    *acc = *acc * np_pow10_small(len) + np_digits_value(w, len);
    *nd += len;
    *i += len;
    if (len < 8) return;
  }
}

// the short forms of a decimal literal; false: not one of them (which says nothing about validity)
DFX_NP bool np_scan_decimal_short(const uint8_t* s, int64_t n, bool* neg, uint64_t* w_out, int64_t* q_out) {
  if (n <= 0) return false;
  int64_t i = 0;
  const uint8_t c0 = s[0];
  *neg = c0 == '-';
  if (c0 == '-' || c0 == '+') i = 1;
  uint64_t acc = 0;
  int nd = 0;
  np_take_digits(s, n, &i, &acc, &nd);
  const int n_int = nd;
  if (i < n && s[i] == '.') {
    ++i;
    np_take_digits(s, n, &i, &acc, &nd);
  }
  if (nd == 0 || nd > 19) return false;
  int64_t exp10 = 0;
  if (i < n) {
    const uint8_t c = s[i];
    if (c != 'e' && c != 'E') return false;
    ++i;
    bool eneg = false;
    if (i < n && (s[i] == '-' || s[i] == '+')) {
      eneg = s[i] == '-';
      ++i;
    }
    if (i >= n) return false;
    const uint64_t w = np_load8(s + i);
    int len = np_digit_run(w);
    if ((int64_t)len > n - i) len = (int)(n - i);
    if (len == 0 || len > 5 || i + len != n) return false;
    exp10 = (int64_t)np_digits_value(w, len);
    if (eneg) exp10 = -exp10;
  }
  *w_out = acc;
  *q_out = exp10 - (int64_t)(nd - n_int);
  return true;
}

DFX_NP int np_parse_f64_w(const uint8_t* s, int64_t n, double* out) {
  bool neg;
  uint64_t w;
  int64_t q;
  if (!np_scan_decimal_short(s, n, &neg, &w, &q)) return np_parse_f64(s, n, out);
  const BiasedFp fp = np_compute_float<52, -1023, 0x7FF, -4, 23, -342, 308>(q, w);
  if (fp.e < 0) return NP_UNSUPPORTED;
  uint64_t bits = fp.f | ((uint64_t)fp.e << 52);
  if (neg) bits |= 0x8000000000000000ull;
  union {
    uint64_t u;
    double d;
  } c;
  c.u = bits;
  *out = c.d;
  return NP_OK;
}

DFX_NP int np_parse_f32_w(const uint8_t* s, int64_t n, float* out) {
  bool neg;
  uint64_t w;
  int64_t q;
  if (!np_scan_decimal_short(s, n, &neg, &w, &q)) return np_parse_f32(s, n, out);
  const BiasedFp fp = np_compute_float<23, -127, 0xFF, -17, 10, -65, 38>(q, w);
  if (fp.e < 0) return NP_UNSUPPORTED;
  uint32_t bits = (uint32_t)fp.f | ((uint32_t)fp.e << 23);
  if (neg) bits |= 0x80000000u;
  union {
    uint32_t u;
    float f;
  } c;
  c.u = bits;
  *out = c.f;
  return NP_OK;
}

DFX_NP int np_parse_int_w(const uint8_t* s, int64_t n, int bits, bool is_signed, uint64_t* out) {
  if (n <= 0) return NP_INVALID;
  int64_t i = 0;
  const uint8_t c0 = s[0];
  const bool neg = c0 == '-';
  if (c0 == '-' || c0 == '+') i = 1;
  uint64_t v = 0;
  int nd = 0;
  np_take_digits(s, n, &i, &v, &nd);
  if (nd == 0 || nd > 19 || i != n) return np_parse_int(s, n, bits, is_signed, out);  // 20 digits, or not a number
  if (neg && !is_signed) return NP_INVALID;  // u*::from_str rejects '-'
  const uint64_t maxpos = is_signed ? ((1ull << (bits - 1)) - 1) : (bits == 64 ? ~0ull : ((1ull << bits) - 1));
  if (v > (neg ? maxpos + 1 : maxpos)) return NP_INVALID;
  *out = neg ? (uint64_t)(0 - v) : v;
  return NP_OK;
}

}  // namespace dfx
