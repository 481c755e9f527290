// dfx_csv_walk.hpp -- the record / field automaton of the reference's CSV source, shared by the device kernels
// (dfx_k_csv.hip) and the host (dfx_csv.cpp: error messages quote the offending cell).
//
// CsvDataSource (src/execution/datasource.rs:33-58) is arrow 0.12's csv::Reader over the `csv` crate with its
// defaults: delimiter ',', quote '"', doubled quotes inside a quoted field are one literal quote, a quote is
// special only as the FIRST byte of a field (inside an unquoted field it is a literal), bytes after a closing
// quote continue the field unquoted, records end at \n, \r or \r\n, empty lines are skipped.  The states:
//   0 StartRecord  1 StartField  2 InField  3 InQuotedField  4 QuoteInQuoted (closing quote or first of a pair)
// and the byte classes Q(uote) D(elimiter) T(erminator) O(ther).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define DFX_HD __host__ __device__ inline
#else
#define DFX_HD inline
#endif

namespace dfx {

// The first error of a batch, as ONE 64-bit word that the device reduces with atomicMin.  arrow's csv::Reader first
// reads every record of the batch (the csv crate reports UnequalLengths there: lowest record wins) and then converts
// column by column (lowest column, then lowest row), so: UnequalLengths (code 3) sorts below every cell error, cell
// errors sort by (column, record).  code: 1 value does not parse, 2 needs arbitrary precision, 3 UnequalLengths.
DFX_HD uint64_t csv_err_pack(int code, int col, int64_t record) {
  if (code == 3) return ((uint64_t)record << 8) | 3ull;
  return (1ull << 63) | ((uint64_t)(col & 0xFF) << 55) | (((uint64_t)record & ((1ull << 47) - 1)) << 8) | (uint64_t)(code & 0xFF);
}
DFX_HD void csv_err_unpack(uint64_t e, int* code, int* col, int64_t* record) {
  *code = (int)(e & 0xFF);
  if (e >> 63) {
    *col = (int)((e >> 55) & 0xFF);
    *record = (int64_t)((e >> 8) & ((1ull << 47) - 1));
  } else {
    *col = 0;
    *record = (int64_t)(e >> 8);
  }
}

enum : uint32_t { CSV_Q = 0, CSV_D = 1, CSV_T = 2, CSV_O = 3 };
DFX_HD uint32_t csv_class(uint8_t c) { return c == '"' ? CSV_Q : c == ',' ? CSV_D : (c == '\n' || c == '\r') ? CSV_T : CSV_O; }

// transition vectors: next state for each start state s in bits [3s, 3s + 3)
constexpr uint32_t csv_pack5(uint32_t s0, uint32_t s1, uint32_t s2, uint32_t s3, uint32_t s4) {
  return s0 | (s1 << 3) | (s2 << 6) | (s3 << 9) | (s4 << 12);
}
constexpr uint32_t kCsvTvQ = csv_pack5(3, 3, 2, 4, 3);
constexpr uint32_t kCsvTvD = csv_pack5(1, 1, 1, 3, 1);
constexpr uint32_t kCsvTvT = csv_pack5(0, 0, 0, 3, 0);
constexpr uint32_t kCsvTvO = csv_pack5(2, 2, 2, 3, 2);
constexpr uint32_t kCsvTvId = csv_pack5(0, 1, 2, 3, 4);
DFX_HD uint32_t csv_tv_of(uint32_t cls) { return cls == CSV_Q ? kCsvTvQ : cls == CSV_D ? kCsvTvD : cls == CSV_T ? kCsvTvT : kCsvTvO; }
DFX_HD uint32_t csv_tv_apply(uint32_t v, uint32_t s) { return (v >> (3 * s)) & 7u; }
DFX_HD uint32_t csv_tv_compose(uint32_t f, uint32_t g) {  // f first, then g
  uint32_t h = 0;
  for (uint32_t s = 0; s < 5; ++s) h |= csv_tv_apply(g, csv_tv_apply(f, s)) << (3 * s);
  return h;
}

// One field of a record as the walker reports it.
struct CsvField {
  uint64_t begin;   // first raw byte (the opening quote of a quoted field)
  uint64_t end;     // one past the last raw byte (the delimiter / terminator / end of input)
  uint64_t close;   // quoted: position of the closing quote (or `end` when the input ends inside the quotes)
  uint32_t ulen;    // length of the unescaped content
  bool quoted;      // started with a quote
  bool complex;     // doubled quotes, or bytes after the closing quote: content is not one contiguous span
};

// Walks the record that starts at `begin` (a non-terminator byte in state StartRecord) and calls
// field(index, CsvField) for every field; stops at the record terminator or at `limit`.  Returns the field count.
template <typename F>
DFX_HD int csv_walk_record(const uint8_t* buf, uint64_t begin, uint64_t limit, F&& field) {
  enum { START = 0, INF = 1, INQ = 2, QQ = 3 };
  int st = START;
  int nf = 0;
  CsvField f;
  f.begin = begin;
  f.end = begin;
  f.close = 0;
  f.ulen = 0;
  f.quoted = false;
  f.complex = false;
  uint64_t pos = begin;
  for (; pos < limit; ++pos) {
    const uint8_t c = buf[pos];
    const bool is_t = c == '\n' || c == '\r';
    bool end_field = false;
    if (st == START) {
      if (c == '"') {
        st = INQ;
        f.quoted = true;
      } else if (c == ',') {
        end_field = true;
      } else if (is_t) {
        break;
      } else {
        st = INF;
        f.ulen = 1;
      }
    } else if (st == INF) {
      if (c == ',') end_field = true;
      else if (is_t) break;
      else ++f.ulen;
    } else if (st == INQ) {
      if (c == '"') {
        st = QQ;
        f.close = pos;
      } else {
        ++f.ulen;
      }
    } else {  // QQ
      if (c == '"') {
        ++f.ulen;
        f.complex = true;
        st = INQ;
      } else if (c == ',') {
        end_field = true;
      } else if (is_t) {
        break;
      } else {
        st = INF;
        ++f.ulen;
        f.complex = true;
      }
    }
    if (end_field) {
      f.end = pos;
      field(nf++, f);
      f.begin = pos + 1;
      f.close = 0;
      f.ulen = 0;
      f.quoted = false;
      f.complex = false;
      st = START;
    }
  }
  f.end = pos;
  if (st == INQ) f.close = pos;  // input ended inside the quotes
  field(nf++, f);
  return nf;
}

// contiguous content span of a field (valid when !f.complex)
DFX_HD void csv_field_span(const CsvField& f, uint64_t* b, uint64_t* e) {
  if (f.quoted) {
    *b = f.begin + 1;
    *e = f.close;
  } else {
    *b = f.begin;
    *e = f.end;
  }
}

// unescaped content of a field -> out (f.ulen bytes)
DFX_HD void csv_copy_field(const uint8_t* buf, const CsvField& f, uint8_t* out) {
  if (!f.quoted) {
    for (uint64_t p = f.begin; p < f.end; ++p) *out++ = buf[p];
    return;
  }
  int st = 0;  // 0 in quotes, 1 just saw a quote, 2 unquoted tail
  for (uint64_t p = f.begin + 1; p < f.end; ++p) {
    const uint8_t c = buf[p];
    if (st == 0) {
      if (c == '"') st = 1;
      else *out++ = c;
    } else if (st == 1) {
      *out++ = c;  // a second quote is the literal quote; anything else starts the unquoted tail
      st = c == '"' ? 0 : 2;
    } else {
      *out++ = c;
    }
  }
}

// ---- quote-free text, word at a time (round 5) ------------------------------------------------------------
// Without a quote the automaton collapses: the state before a byte is a function of the byte before it (terminator ->
// StartRecord, delimiter -> StartField, other -> InField), a record is its bytes up to the first terminator, and every
// byte between that terminator and the next record's start is a terminator.  The device kernels use that for tiles
// (k_csv_tile_trans / k_csv_mark_write: record starts from terminator masks) and for records (k_csv_parse: cell
// boundaries from delimiter masks); the host build of these functions is held against the walker above by
// tests/native/csv_walk_fuzz.cpp.

// exact per-byte "== c" flags of a 32-bit word, gathered into 4 bits (bit k: byte k)
DFX_HD uint32_t csv_eq_nibble(uint32_t w, uint32_t c4) {
  const uint32_t x = w ^ c4;
  const uint32_t t = ((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x;  // bit 7 of every byte: byte != 0
  const uint32_t z = ~(t | 0x7F7F7F7Fu) >> 7;                // bits 0, 8, 16, 24: byte == 0
  return ((z * 0x00204081u) >> 21) & 0xFu;
}

// bit i: byte i of the 32 bytes w[0..8) is a record terminator (\n or \r)
DFX_HD uint32_t csv_tmask32(const uint32_t* w) {
  uint32_t t = 0;
  for (int i = 0; i < 8; ++i) t |= (csv_eq_nibble(w[i], 0x0A0A0A0Au) | csv_eq_nibble(w[i], 0x0D0D0D0Du)) << (4 * i);
  return t;
}

// record starts of 32 quote-free bytes outside a quoted field: a non-terminator right after a terminator
// (`after_t`: the byte before the 32 is a terminator, or the text starts here)
DFX_HD uint32_t csv_plain_starts(uint32_t t, bool after_t) { return ~t & ((t << 1) | (after_t ? 1u : 0u)); }

DFX_HD uint64_t csv_load8(const uint8_t* p) {
  uint64_t w;
  __builtin_memcpy(&w, p, 8);
  return w;
}

// bit 7 of every byte among the first `left` (>= 1) bytes of w that is a delimiter -- exact: no carry crosses a byte
DFX_HD uint64_t csv_delims8(uint64_t w, uint32_t left) {
  const uint64_t low7 = 0x7F7F7F7F7F7F7F7Full, high = 0x8080808080808080ull;
  const uint64_t keep = left >= 8u ? ~0ull : (1ull << (8u * left)) - 1ull;
  const uint64_t x = w ^ 0x2C2C2C2C2C2C2C2Cull;
  return ~(((x & low7) + low7) | x) & high & keep;
}

// is one of the first `left` (>= 1) bytes of w a quote?  (x - 1) & ~x & 0x80 flags every zero byte and, through the
// borrow, possibly bytes ABOVE one: existence is exact, and the bytes that `left` cuts off are the upper ones
DFX_HD bool csv_any_quote8(uint64_t w, uint32_t left) {
  const uint64_t ones = 0x0101010101010101ull, high = 0x8080808080808080ull;
  const uint64_t keep = left >= 8u ? ~0ull : (1ull << (8u * left)) - 1ull;
  const uint64_t x = w ^ 0x2222222222222222ull;
  return ((x - ones) & ~x & high & keep) != 0ull;
}

// Cell boundaries of the record text[b, limit) -- limit: the next record's start, or the end of the input -- when it holds
// no quote and exactly F fields: cells[i] (i < F - 1) = offset of the i-th delimiter, cells[F - 1] = the record's end (the
// terminators and blank lines before `limit` stripped).  false: a quote, or another field count -- the caller walks the
// record with csv_walk_record.  text[limit .. limit + 7] must be readable.  POS: an unsigned type that holds the offsets.
template <typename POS>
DFX_HD bool csv_plain_record(const uint8_t* text, uint32_t b, uint32_t limit, uint32_t F, POS* cells) {
  uint32_t e = limit;
  while (e > b && (text[e - 1u] == '\n' || text[e - 1u] == '\r')) --e;
  uint32_t found = 0;
  bool quote = false;
  for (uint32_t p = b; p < e; p += 8u) {
    const uint64_t w = csv_load8(text + p);
    uint64_t zc = csv_delims8(w, e - p);
    quote = quote || csv_any_quote8(w, e - p);
    while (zc) {
      const uint32_t k = (uint32_t)__builtin_ctzll(zc) >> 3;
      if (found + 1u < F) cells[found] = (POS)(p + k);
      ++found;
      zc &= zc - 1ull;
    }
  }
  cells[F - 1u] = (POS)e;
  return !quote && found + 1u == F;
}

}  // namespace dfx
