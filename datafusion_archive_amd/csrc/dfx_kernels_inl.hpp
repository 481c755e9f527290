// dfx_kernels_inl.hpp -- device-side building blocks shared by the kernel translation units:
// scalar helpers, the per-row expression interpreter, the accumulator algebra and the group-table
// find-or-insert.  Everything here is __device__ __forceinline__ (or a template), so each .hip
// translation unit gets its own copy and they compile in parallel.
#pragma once
#include <hip/hip_runtime.h>

#include "dfx_kernels.hpp"
#include "dfx_sigs.hpp"

namespace dfx {

typedef uint64_t u64x16 __attribute__((ext_vector_type(16)));
typedef uint64_t u64x8 __attribute__((ext_vector_type(8)));

#define DEV __device__ __forceinline__

// ---------------------------------------------------------------------------------------------
// scalar helpers
// ---------------------------------------------------------------------------------------------
DEV double as_f64(uint64_t x) { return __longlong_as_double((long long)x); }
DEV uint64_t f64_bits(double x) { return (uint64_t)__double_as_longlong(x); }
DEV float as_f32(uint64_t x) { return __uint_as_float((uint32_t)x); }
DEV uint64_t f32_bits(float x) { return (uint64_t)__float_as_uint(x); }
DEV bool get_bit(const uint8_t* bm, int64_t i) { return (bm[i >> 3] >> (i & 7)) & 1; }
DEV int lane_id() { return (int)(threadIdx.x & 63); }

__host__ __device__ inline uint64_t mix64(uint64_t z) {  // splitmix64 finaliser
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// Group-key hash.  Every scanned (or routed) row pays for it, and 64-bit multiplies are ~8 quarter-rate
// VALU ops each on CDNA, so the hash is built from three 32-bit multiplies per key word (murmur3-style
// two-block mix + fmix32 finaliser; ~2.5x cheaper than a splitmix64 finaliser).  The 32 hash bits are
// returned in the HIGH half: table slot = h >> (64 - log2 capacity), capacity <= 2^31.
__host__ __device__ inline uint32_t hash_word(uint64_t k, uint32_t seed) {
  uint32_t x = ((uint32_t)k ^ seed) * 0xCC9E2D51u;
  x ^= x >> 15;
  x ^= (uint32_t)(k >> 32);
  x *= 0x85EBCA6Bu;
  x ^= x >> 13;
  x *= 0xC2B2AE35u;
  return x ^ (x >> 16);
}
// hash_word restricted to keys below 2^32 is a bijection of the low word (odd multiplies and xor-shifts, the high
// word's xor is zero): its inverse, for the seed hash_keys<1> uses.  Multiplicative inverses modulo 2^32.
constexpr uint32_t kHashSeed0 = 0x9E3779B9u;
__host__ __device__ inline uint32_t unhash_word32(uint32_t h) {
  uint32_t x = h;
  x ^= x >> 16;
  x *= 0x7ED1B41Du;  // 0xC2B2AE35^-1
  x ^= x >> 13;
  x ^= x >> 26;
  x *= 0xA5CB9243u;  // 0x85EBCA6B^-1
  x ^= x >> 15;
  x ^= x >> 30;
  x *= 0xDEE13BB1u;  // 0xCC9E2D51^-1
  return x ^ kHashSeed0;
}
template <int KW>
__host__ __device__ inline uint64_t hash_keys(const uint64_t* key) {
  uint32_t h = hash_word(key[0], kHashSeed0);
#pragma unroll
  for (int w = 1; w < KW; ++w) h = hash_word(key[w], h);
  return (uint64_t)h << 32;
}
// owner rank of a group in the multi-GPU exchange: bits independent of the slot index
__host__ __device__ inline uint32_t hash_rank(uint64_t h, uint32_t world) {
  uint32_t x = (uint32_t)(h >> 32) ^ 0x5BD1E995u;
  x *= 0x2C1B3C6Du;
  x ^= x >> 15;
  x *= 0x297A2D39u;
  x ^= x >> 16;
  return x % world;
}

DEV bool is_signed_int(uint8_t t) { return t >= T_I8 && t <= T_I64; }
DEV bool is_int(uint8_t t) { return t >= T_I8 && t <= T_U64; }

// canonical 64-bit image of a value of dtype t: signed ints sign-extended, unsigned zero-extended,
// f32 as its bit pattern in the low dword, f64 as its bit pattern, Boolean 0/1
DEV uint64_t wrap_to(uint8_t t, uint64_t x) {
  switch (t) {
    case T_I8: return (uint64_t)(int64_t)(int8_t)x;
    case T_I16: return (uint64_t)(int64_t)(int16_t)x;
    case T_I32: return (uint64_t)(int64_t)(int32_t)x;
    case T_U8: return (uint64_t)(uint8_t)x;
    case T_U16: return (uint64_t)(uint16_t)x;
    case T_U32: return (uint64_t)(uint32_t)x;
    default: return x;
  }
}

DEV uint64_t load_canonical(uint8_t t, const void* base, int64_t row, int64_t bit_offset) {
  switch (t) {
    case T_F64: case T_I64: case T_U64: return ((const uint64_t*)base)[row];
    case T_I32: return (uint64_t)(int64_t)((const int32_t*)base)[row];
    case T_U32: case T_F32: return (uint64_t)((const uint32_t*)base)[row];
    case T_I16: return (uint64_t)(int64_t)((const int16_t*)base)[row];
    case T_U16: return (uint64_t)((const uint16_t*)base)[row];
    case T_I8: return (uint64_t)(int64_t)((const int8_t*)base)[row];
    case T_U8: return (uint64_t)((const uint8_t*)base)[row];
    case T_BOOL: return (uint64_t)get_bit((const uint8_t*)base, bit_offset + row);
    default: return 0;
  }
}

DEV void store_typed(uint8_t t, void* base, int64_t row, uint64_t v) {
  switch (t) {
    case T_F64: case T_I64: case T_U64: ((uint64_t*)base)[row] = v; break;
    case T_I32: case T_U32: case T_F32: ((uint32_t*)base)[row] = (uint32_t)v; break;
    case T_I16: case T_U16: ((uint16_t*)base)[row] = (uint16_t)v; break;
    case T_I8: case T_U8: ((uint8_t*)base)[row] = (uint8_t)v; break;
    default: break;
  }
}

// Rust `as` numeric casts (float -> int saturating, NaN -> 0); same table as oracle cast_val
DEV int64_t sat_to_i64(double x, int64_t lo, int64_t hi) {
  if (x != x) return 0;
  if (x <= (double)lo) return lo;
  if (x >= (double)hi) return hi;
  return (int64_t)x;
}
DEV uint64_t sat_to_u64(double x, uint64_t hi) {
  if (x != x) return 0;
  if (x <= 0.0) return 0;
  if (x >= (double)hi) return hi;
  return (uint64_t)x;
}

DEV uint64_t cast_value(uint8_t from, uint8_t to, uint64_t v) {
  if (from == to) return v;
  if (is_int(from)) {
    if (is_int(to)) return wrap_to(to, v);
    if (to == T_F64) return f64_bits(is_signed_int(from) ? (double)(int64_t)v : (double)v);
    return f32_bits(is_signed_int(from) ? (float)(int64_t)v : (float)v);
  }
  const double x = (from == T_F32) ? (double)as_f32(v) : as_f64(v);
  switch (to) {
    case T_F32: return (from == T_F32) ? v : f32_bits((float)as_f64(v));
    case T_F64: return f64_bits(x);
    case T_I8: return (uint64_t)sat_to_i64(x, -128, 127);
    case T_I16: return (uint64_t)sat_to_i64(x, -32768, 32767);
    case T_I32: return (uint64_t)sat_to_i64(x, -2147483648ll, 2147483647ll);
    case T_I64: return (uint64_t)sat_to_i64(x, (int64_t)0x8000000000000000ull, 0x7FFFFFFFFFFFFFFFll);
    case T_U8: return sat_to_u64(x, 255ull);
    case T_U16: return sat_to_u64(x, 65535ull);
    case T_U32: return sat_to_u64(x, 4294967295ull);
    case T_U64: return sat_to_u64(x, 0xFFFFFFFFFFFFFFFFull);
    default: return 0;
  }
}

// ... and its casts (the projection kernels inline cast_value; the interpreter's loop calls it)
__device__ __attribute__((noinline)) uint64_t cast_value_call(uint8_t from, uint8_t to, uint64_t v) { return cast_value(from, to, v); }

// integer Divide of the interpreter (arrow 0.12 array_ops::divide: DivideByZero; Rust's `/`: overflow panic).  Out of line: 64-bit
// integer division expands to ~150 instructions and a dozen live registers, which the interpreter's loop would carry in every path
struct IntDivResult {
  uint64_t value;
  uint32_t err;  // error bits to OR into the kernel's (returned by value: an `err` passed by reference would live in scratch memory)
};
__device__ __attribute__((noinline)) IntDivResult int_divide(uint8_t t, uint64_t x, uint64_t y, bool report) {
  IntDivResult r = {0, 0};
  if (y == 0) {
    if (report) r.err = 1u;
    return r;
  }
  if (is_signed_int(t)) {
    const uint64_t mn = wrap_to(t, 1ull << (t == T_I8 ? 7 : t == T_I16 ? 15 : t == T_I32 ? 31 : 63));
    if ((int64_t)y == -1 && x == mn) {
      if (report) r.err = 2u;
      r.value = x;
      return r;
    }
    r.value = (uint64_t)((int64_t)x / (int64_t)y);
    return r;
  }
  r.value = x / y;
  return r;
}

// ---------------------------------------------------------------------------------------------
// the per-row expression interpreter
// ---------------------------------------------------------------------------------------------
// NOTE: the register files are separate local vector VALUES (never members of a struct that is
// passed by reference): that is what lets SROA keep them in VGPRs and lower the wave-uniform
// dynamic indices to s_set_gpr_idx instead of scratch memory.
//
// Memory-level parallelism: a wave owns U consecutive 64-row groups per trip.  It first issues the
// column loads of ALL U groups (U x n_cols independent 512-byte requests in flight), then
// interprets the groups one after the other.  The column bank (values loaded per row) is sized to
// the program: BANK in {2,4,8} columns, U chosen so that BANK * U <= 16 values (32 VGPRs).
typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
typedef uint64_t u64x4 __attribute__((ext_vector_type(4)));
template <int BANK> struct Bank;
template <> struct Bank<2> { typedef u64x2 type; };
template <> struct Bank<4> { typedef u64x4 type; };
template <> struct Bank<8> { typedef u64x8 type; };

#define ROWSTATE_ARGS(s) s##_col, s##_reg, s##_colvalid, s##_regvalid
#define ROWSTATE_PARAMS COLV& s_col, u64x16& s_reg, uint32_t& s_colvalid, uint32_t& s_regvalid
#define ROWSTATE_CPARAMS const COLV& s_col, const u64x16& s_reg, const uint32_t& s_colvalid, const uint32_t& s_regvalid

// issue every column load of this row back to back (independent loads, all in flight together)
template <typename COLV>
DEV void load_columns(const DevProgram& P, const DevColumns& C, int64_t row, bool inb, COLV& s_col,
                      uint32_t& s_colvalid) {
  constexpr int BANK = (int)(sizeof(COLV) / 8);
  s_colvalid = 0xFFFFFFFFu;
  if (P.wide8) {  // wave-uniform.  8-byte columns without nulls: one unconditional load per bank slot (a slot past n_cols
                  // re-reads column 0 -- a cache hit), nothing between the loads, so all of them are in flight together.
                  // The dtype switch below waits for each load on its own (sub-word loads are widened at once): a
                  // wave then has ONE load in flight, and pass 1 of the partitioned GROUP BY ran at 1.7 TB/s.
#pragma unroll
    for (int c = 0; c < BANK; ++c) {
      const uint64_t* base = (const uint64_t*)C.c[c < P.n_cols ? c : 0].values;
      s_col[c] = __builtin_nontemporal_load(base + (inb ? row : 0));
    }
    return;
  }
#pragma unroll
  for (int c = 0; c < BANK; ++c) {
    if (c < P.n_cols) {
      uint64_t v = 0;
      if (inb) v = load_canonical(P.col_dtype[c], C.c[c].values, row, C.c[c].bit_offset);
      s_col[c] = v;
      if (P.has_nulls) {  // wave-uniform branch
        if (C.c[c].validity != nullptr) {
          const bool ok = inb ? get_bit(C.c[c].validity, C.c[c].bit_offset + row) : false;
          if (!ok) s_colvalid &= ~(1u << c);
        }
      }
    }
  }
}

// (Round 5 also tried the program itself in vector registers -- lane j holding SSA instruction j and literal j, decoded with
// v_readlane instead of scalar loads from the kernel arguments: 1.74 ms per 2^27-row launch of the headline through the interpreter
// against 1.63 without; the three registers it takes push the ring kernel's allocation over its 128, and the scalar loads were not
// what the waves waited for.  profiles/r05_interpreter_counters.txt)
// NULLS == false: the batch's referenced columns carry no nulls (DevProgram::has_nulls, decided at bind time) -- every operand is
// valid, and `valid` is a compile-time constant for the code around the call
template <typename COLV, bool NULLS = true>
DEV void fetch(const DevProgram& P, ROWSTATE_CPARAMS, uint8_t opnd, uint64_t& v, bool& valid) {
  const int idx = opnd & 63;
  const int kind = opnd >> 6;
  if (kind == OPK_REG) {
    v = s_reg[idx];
    valid = NULLS ? ((s_regvalid >> idx) & 1) : true;
  } else if (kind == OPK_COL) {
    v = s_col[idx & (int)(sizeof(COLV) / 8 - 1)];
    valid = NULLS ? ((s_colvalid >> idx) & 1) : true;
  } else {
    v = P.imm[idx & (kMaxImm - 1)];
    valid = true;
  }
}

// Executes the SSA program for one row.  Semantics per op are arrow 0.12 array_ops, the same
// table the oracle restates (oracle/dfx_oracle.c: compare_arrays / boolean_arrays / math_arrays).
// NULLS == false (round 5): with every operand valid the per-lane validity booleans -- and the divergent control flow the compiler
// builds around `if (vx && vy)` for arrow's null rules, lane masks saved and restored per SSA instruction -- fold away.
template <typename COLV, bool NULLS>
DEV void run_program_impl(const DevProgram& P, ROWSTATE_PARAMS, bool active, uint32_t& err) {
  s_regvalid = NULLS ? 0u : 0xFFFFFFFFu;
  for (int pc = 0; pc < P.n_ins; ++pc) {
    const DevIns in = P.ins[pc];
    const uint8_t t = in.t;
    uint64_t x, y = 0;
    bool vx, vy = true;
    fetch<COLV, NULLS>(P, ROWSTATE_ARGS(s), in.a, x, vx);
    if (in.op != DOP_CAST) fetch<COLV, NULLS>(P, ROWSTATE_ARGS(s), in.b, y, vy);
    uint64_t res = 0;
    bool v = vx && vy;
    if (in.op <= DOP_GE) {
      bool lt, eq, gt;
      if (t == T_F64) {
        const double a = as_f64(x), b = as_f64(y);
        lt = a < b; eq = a == b; gt = a > b;
      } else if (t == T_F32) {
        const float a = as_f32(x), b = as_f32(y);
        lt = a < b; eq = a == b; gt = a > b;
      } else if (t == T_U64) {
        lt = x < y; eq = x == y; gt = x > y;
      } else {
        const int64_t a = (int64_t)x, b = (int64_t)y;
        lt = a < b; eq = a == b; gt = a > b;
      }
      bool r;
      if (vx && vy) {
        switch (in.op) {
          case DOP_EQ: r = eq; break;
          case DOP_NE: r = !eq; break;
          case DOP_LT: r = lt; break;
          case DOP_LE: r = lt || eq; break;
          case DOP_GT: r = gt; break;
          default: r = gt || eq; break;
        }
      } else {  // arrow 0.12 bool_op over Option<T>: never null; None sorts below every value
        switch (in.op) {
          case DOP_EQ: r = (!vx && !vy); break;
          case DOP_NE: r = (vx != vy); break;
          case DOP_LT: r = (!vx && vy); break;
          case DOP_LE: r = !vx; break;
          case DOP_GT: r = (vx && !vy); break;
          default: r = !vy; break;
        }
      }
      res = r ? 1 : 0;
      v = true;
    } else if (in.op == DOP_AND) {
      res = x & y & 1;
    } else if (in.op == DOP_OR) {
      res = (x | y) & 1;
    } else if (in.op == DOP_CAST) {
      res = cast_value_call(t, in.b, x);
      v = vx;
    } else if (t == T_F64) {
      const double a = as_f64(x), b = as_f64(y);
      double o;
      switch (in.op) {
        case DOP_ADD: o = a + b; break;
        case DOP_SUB: o = a - b; break;
        case DOP_MUL: o = a * b; break;
        default:
          if (v && active && b == 0.0) err |= 1u;
          o = a / b;
          break;
      }
      res = f64_bits(o);
    } else if (t == T_F32) {
      const float a = as_f32(x), b = as_f32(y);
      float o;
      switch (in.op) {
        case DOP_ADD: o = a + b; break;
        case DOP_SUB: o = a - b; break;
        case DOP_MUL: o = a * b; break;
        default:
          if (v && active && b == 0.0f) err |= 1u;
          o = a / b;
          break;
      }
      res = f32_bits(o);
    } else {
      uint64_t o;
      switch (in.op) {
        case DOP_ADD: o = x + y; break;
        case DOP_SUB: o = x - y; break;
        case DOP_MUL: o = x * y; break;
        default: {
          const IntDivResult dr = int_divide(t, x, y, v && active);
          o = dr.value;
          err |= dr.err;
          break;
        }
      }
      res = wrap_to(t, o);
    }
    // a null result slot holds zero: arrow 0.12's builders append_null() over zero-initialised buffers, and the grouped
    // aggregates read value(row) of their argument WITHOUT a null check (aggregate.rs:561-603), so the slot's content
    // is observable (MAX(x + x) over a null x sees 0, not raw + raw)
    s_reg[pc] = (!NULLS || v) ? res : 0ull;
    if (NULLS) s_regvalid |= (v ? 1u : 0u) << pc;
  }
}

template <typename COLV>
DEV void run_program(const DevProgram& P, ROWSTATE_PARAMS, bool active, uint32_t& err) {
  if (P.has_nulls) run_program_impl<COLV, true>(P, ROWSTATE_ARGS(s), active, err);  // (wave-uniform: a property of the batch)
  else run_program_impl<COLV, false>(P, ROWSTATE_ARGS(s), active, err);
}

template <typename COLV>
DEV bool eval_predicate(const DevProgram& P, ROWSTATE_CPARAMS, uint8_t pred) {
  if (pred == kNoOperand) return true;
  uint64_t v;
  bool valid;
  fetch(P, ROWSTATE_ARGS(s), pred, v, valid);
  // FilterRelation reads filter.value(i): the raw value bit; a null slot holds false (filter.rs:86)
  return valid && (v & 1);
}

// ---------------------------------------------------------------------------------------------
// row-source policies: how one 64-row group turns into (pass, keys, aggregate arguments)
// ---------------------------------------------------------------------------------------------
// Kernels are templated on a policy.  All state lives in kernel-local variables (vector VALUES,
// see the note above); a policy is a bag of static inline functions.  Each kernel contains ONE
// copy of the evaluation code: the U prefetched row-groups are visited by a run-time loop that
// re-selects the group's column bank through a wave-uniform switch.
#define FOR_U _Pragma("unroll") for (int u = 0; u < U; ++u)
#define DFX_SELECT_BANK(uu, col, cv, cur, curv)                     \
  switch (uu) {                                                      \
    case 0: cur = col[0]; curv = cv[0]; break;                       \
    case 1: cur = col[U > 1 ? 1 : 0]; curv = cv[U > 1 ? 1 : 0]; break; \
    case 2: cur = col[U > 2 ? 2 : 0]; curv = cv[U > 2 ? 2 : 0]; break; \
    case 3: cur = col[U > 3 ? 3 : 0]; curv = cv[U > 3 ? 3 : 0]; break; \
    case 4: cur = col[U > 4 ? 4 : 0]; curv = cv[U > 4 ? 4 : 0]; break; \
    case 5: cur = col[U > 5 ? 5 : 0]; curv = cv[U > 5 ? 5 : 0]; break; \
    case 6: cur = col[U > 6 ? 6 : 0]; curv = cv[U > 6 ? 6 : 0]; break; \
    default: cur = col[U > 7 ? 7 : 0]; curv = cv[U > 7 ? 7 : 0]; break; \
  }

// generic: the SSA register program (any expression the compiler accepts, nulls included)
template <int BANK, int U_>
struct InterpPolicy {
  static constexpr bool kIsStatic = false;
  static constexpr bool kHasTripLoad = false;
  struct PREP {};  // per-wave state prepared before the scan loop (PlanPolicy: the plan words in vector registers)
  static DEV void prepare(const DevProgram&, const DevFastPlan&, PREP&) {}
  static constexpr int kPredTerms = -1;  // (run-time)
  static constexpr int U = U_;
  static constexpr int kStaticNa = 0;  // aggregates known at compile time (0: run-time)
  typedef typename Bank<BANK>::type COLV;
  // (defined after pass below)
  static DEV void load(const DevProgram& P, const DevColumns& C, int64_t row, bool inb, COLV& col, uint32_t& cv) {
    load_columns(P, C, row, inb, col, cv);
  }
  static DEV int na(const DevTable& T) { return T.na; }
  static DEV uint8_t acc_kind(const DevTable& T, int a) { return T.acc_kind[a]; }
  static DEV uint8_t xform(const DevTable& T, int a) { return T.val_xform[a]; }
  static DEV void eval(const DevProgram& P, const DevFastPlan&, const COLV& cur, uint32_t curv, u64x16& reg,
                       uint32_t& rv, bool inb, uint32_t& err, const PREP& = PREP()) {
    COLV c = cur;
    uint32_t cvv = curv;
    run_program(P, c, reg, cvv, rv, inb, err);
  }
  static DEV bool pass(const DevProgram& P, const DevFastPlan&, uint8_t pred, const COLV& cur, uint32_t curv,
                       const u64x16& reg, uint32_t rv, const PREP& = PREP()) {
    return eval_predicate(P, cur, reg, curv, rv, pred);
  }
  static DEV int form_of(const DevFastPlan&) { return 0; }
  template <int FORM>
  static DEV bool pass_form(const DevProgram& P, const DevFastPlan& F, uint8_t pred, const COLV& cur, uint32_t cv, const u64x16& reg, uint32_t rv, const PREP& prep = PREP()) {
    return pass(P, F, pred, cur, cv, reg, rv, prep);
  }
  // the wave's lane mask of pass_form (scalar): what the scan loops that keep their predicates in masks call
  template <int FORM>
  static DEV uint64_t pass_mask(const DevProgram& P, const DevFastPlan& F, uint8_t pred, const COLV& cur, uint32_t cv, const u64x16& reg, uint32_t rv, const PREP& prep = PREP()) {
    return __ballot(pass_form<FORM>(P, F, pred, cur, cv, reg, rv, prep));
  }
  static DEV void load_trip(const DevColumns&, int64_t, bool, int64_t, int, COLV (&)[U_], uint32_t (&)[U_]) {}  // (kIsStatic only)
  static DEV uint64_t key(const DevProgram& P, const DevFastPlan&, uint8_t opnd, int, const COLV& cur, uint32_t curv,
                          const u64x16& reg, uint32_t rv) {
    uint64_t v;
    bool valid;
    fetch(P, cur, reg, curv, rv, opnd, v, valid);
    return v;
  }
  static DEV void arg(const DevProgram& P, const DevFastPlan&, uint8_t opnd, int, const COLV& cur, uint32_t curv,
                      const u64x16& reg, uint32_t rv, uint64_t& v, bool& valid) {
    fetch(P, cur, reg, curv, rv, opnd, v, valid);
  }
};

// the interpreter for kernels that route ONE value per row (the narrow-row flavours of the partitioned pass 1): as FastPolicy1 --
// with the aggregate count a compile-time 1 the eight-wide value arrays of the routing code (16 VGPRs per row group, twice) drop
// out.  Round 5: the narrow ring kernel ran the interpreter at 128 VGPRs with 12 of them spilled to scratch -- and every scratch
// reload waits for the column loads in flight as well (one in-order counter)
template <int BANK, int U_>
struct InterpPolicy1 : InterpPolicy<BANK, U_> {
  static constexpr int kStaticNa = 1;
  static DEV int na(const DevTable&) { return 1; }
};

// three-way compare code of two canonical values of dtype t: 1 less, 2 equal, 4 greater, 0 unordered
DEV uint32_t cmp3(uint8_t t, uint64_t x, uint64_t y) {
  if (t == T_F64) {
    const double a = as_f64(x), b = as_f64(y);
    return (a < b ? 1u : 0u) | (a == b ? 2u : 0u) | (a > b ? 4u : 0u);
  }
  if (t == T_F32) {
    const float a = as_f32(x), b = as_f32(y);
    return (a < b ? 1u : 0u) | (a == b ? 2u : 0u) | (a > b ? 4u : 0u);
  }
  if (t == T_U64) return (x < y ? 1u : 0u) | (x == y ? 2u : 0u) | (x > y ? 4u : 0u);
  const int64_t a = (int64_t)x, b = (int64_t)y;
  return (a < b ? 1u : 0u) | (a == b ? 2u : 0u) | (a > b ? 4u : 0u);
}

// Wave-level comparison: the lane mask of `(cmp3(t, x, y) & m) != 0` for a wave-uniform m (it comes from the
// kernarg segment).  The three v_cmp write SGPR pairs and the selection by m runs on the scalar unit -- no
// per-lane select / or, no branch; the caller turns the final mask back into a per-lane predicate with
// inverse_ballot (a copy into VCC).  Bits of inactive lanes are meaningless.
template <typename TT>
DEV uint64_t cmp_mask_by(TT a, TT b, uint32_t m) {
  const uint64_t lt = __ballot(a < b), eq = __ballot(a == b), gt = __ballot(a > b);
  return ((m & 1u) ? lt : 0ull) | ((m & 2u) ? eq : 0ull) | ((m & 4u) ? gt : 0ull);
}
DEV uint64_t cmp_mask(uint8_t t, uint64_t x, uint64_t y, uint32_t m) {
  if (t == T_F64) return cmp_mask_by<double>(as_f64(x), as_f64(y), m);
  if (t == T_F32) return cmp_mask_by<float>(as_f32(x), as_f32(y), m);
  if (t == T_U64) return cmp_mask_by<uint64_t>(x, y, m);
  return cmp_mask_by<int64_t>((int64_t)x, (int64_t)y, m);
}
DEV bool lane_of_mask(uint64_t mask) { return __builtin_amdgcn_inverse_ballot_w64(mask); }

// shape-specialised (DevFastPlan): conjunction of `column <op> literal`, plain-column keys, column /
// short-product arguments; no nulls.  Straight-line code, the only scalar work is reading the plan.
template <int BANK, int U_>
struct FastPolicy {
  static constexpr bool kIsStatic = false;
  static constexpr bool kHasTripLoad = false;
  struct PREP {};
  static DEV void prepare(const DevProgram&, const DevFastPlan&, PREP&) {}
  static constexpr int kPredTerms = -1;  // (run-time)
  static constexpr int U = U_;
  static constexpr int kStaticNa = 0;  // aggregates known at compile time (0: run-time)
  typedef typename Bank<BANK>::type COLV;
  static DEV void load(const DevProgram& P, const DevColumns& C, int64_t row, bool inb, COLV& col, uint32_t& cv) {
    load_columns(P, C, row, inb, col, cv);
  }
  static DEV int na(const DevTable& T) { return T.na; }
  static DEV uint8_t acc_kind(const DevTable& T, int a) { return T.acc_kind[a]; }
  static DEV uint8_t xform(const DevTable& T, int a) { return T.val_xform[a]; }
  static DEV void eval(const DevProgram&, const DevFastPlan&, const COLV&, uint32_t, u64x16&, uint32_t&, bool,
                       uint32_t&, const PREP& = PREP()) {}
  static DEV bool pass(const DevProgram&, const DevFastPlan& F, uint8_t, const COLV& cur, uint32_t, const u64x16&,
                       uint32_t, const PREP& = PREP()) {
    uint64_t ok = ~0ull;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i < F.np) {
        const DevFastTerm t = F.term[i];
        ok &= cmp_mask(t.dtype, cur[t.col & (BANK - 1)], F.term_imm[i], (uint32_t)t.m) ^ (t.inv ? ~0ull : 0ull);
      }
    }
    return lane_of_mask(ok);
  }
  static DEV int form_of(const DevFastPlan&) { return 0; }
  template <int FORM>
  static DEV bool pass_form(const DevProgram& P, const DevFastPlan& F, uint8_t pred, const COLV& cur, uint32_t cv, const u64x16& reg, uint32_t rv, const PREP& prep = PREP()) {
    return pass(P, F, pred, cur, cv, reg, rv, prep);
  }
  // the wave's lane mask of pass_form (scalar): what the scan loops that keep their predicates in masks call
  template <int FORM>
  static DEV uint64_t pass_mask(const DevProgram& P, const DevFastPlan& F, uint8_t pred, const COLV& cur, uint32_t cv, const u64x16& reg, uint32_t rv, const PREP& prep = PREP()) {
    return __ballot(pass_form<FORM>(P, F, pred, cur, cv, reg, rv, prep));
  }
  static DEV void load_trip(const DevColumns&, int64_t, bool, int64_t, int, COLV (&)[U_], uint32_t (&)[U_]) {}  // (kIsStatic only)
  static DEV uint64_t key(const DevProgram&, const DevFastPlan& F, uint8_t, int k, const COLV& cur, uint32_t,
                          const u64x16&, uint32_t) {
    return cur[F.keycol[k] & (BANK - 1)];
  }
  static DEV double factor(const DevFastPlan& F, int a, int j, const COLV& cur) {
    const DevFastFactor f = F.arg[a].f[j];
    const double x = as_f64(cur[f.col & (BANK - 1)]);
    const double imm = as_f64(F.arg_imm[a][j]);
    switch (f.kind) {
      case FF_IMM_MINUS_COL: return imm - x;
      case FF_COL_PLUS_IMM: return x + imm;
      case FF_COL_MINUS_IMM: return x - imm;
      case FF_COL_TIMES_IMM: return x * imm;
      default: return x;
    }
  }
  static DEV void arg(const DevProgram&, const DevFastPlan& F, uint8_t, int a, const COLV& cur, uint32_t,
                      const u64x16&, uint32_t, uint64_t& v, bool& valid) {
    valid = true;
    const DevFastArg A = F.arg[a];
    if (A.nf == 1 && A.f[0].kind == FF_COL) {
      v = cur[A.f[0].col & (BANK - 1)];
      return;
    }
    double acc = factor(F, a, 0, cur);
    if (A.nf > 1) acc = acc * factor(F, a, 1, cur);
    if (A.nf > 2) acc = acc * factor(F, a, 2, cur);
    v = f64_bits(acc);
  }
};

// FastPolicy for scans that route ONE value per row (one aggregate, or PTF_SHARED's common operand): the per-aggregate
// loops and their plan words (8 aggregates x 3 factors) drop out of the kernel
template <int BANK, int U_>
struct FastPolicy1 : FastPolicy<BANK, U_> {
  static constexpr int kStaticNa = 1;
  static DEV int na(const DevTable&) { return 1; }
};

// compile-time shape signature (dfx_sigs.hpp): column slots, term types, accumulator kinds and
// operand transforms are constants, every referenced column is 8 bytes wide, no nulls.  The only
// run-time inputs are the comparison masks and the literals.
template <int BANK, int U_, typename SIG>
struct StaticPolicy {
  static constexpr bool kIsStatic = true;
  static constexpr bool kHasTripLoad = true;
  struct PREP {};
  static DEV void prepare(const DevProgram&, const DevFastPlan&, PREP&) {}
  static constexpr int kPredTerms = SIG::NP;  // 0: the signature has no predicate (every in-range row passes)
  static constexpr int U = U_;
  static constexpr int kStaticNa = SIG::NA;
  typedef typename Bank<BANK>::type COLV;
  static DEV void load(const DevProgram&, const DevColumns& C, int64_t row, bool inb, COLV& col, uint32_t& cv) {
    cv = 0xFFFFFFFFu;
#pragma unroll
    for (int c = 0; c < BANK; ++c)
      if (c < SIG::NCOL)  // streamed once: non-temporal, so the table blocks / open region lines keep the L2.
        // Out-of-range lanes re-read row 0 (no branch around the load: the compiler can then count the
        // loads in flight and software-pipelined kernels wait with vmcnt(N) instead of vmcnt(0)).
#ifdef DFX_COND_LOAD
        col[c] = inb ? __builtin_nontemporal_load((const uint64_t*)C.c[c].values + row) : 0ull;
#else
        col[c] = __builtin_nontemporal_load((const uint64_t*)C.c[c].values + (inb ? row : 0));
#endif
  }
  // The U row groups of one trip, starting at group w0 (wave-uniform): ONE scalar base per column and a 32-bit lane index
  // clamped to the last row of the batch (v_min_u32 + shift per load; the per-row form clamps a 64-bit index: six vector
  // instructions per load).  Lanes past the end re-read the last row (their rows are masked out by `row < n` later); an
  // inactive trip (w0 past the end: the software pipeline's last prefetch) reads row 0.  Unconditional loads: the compiler
  // can count them (vmcnt(N), not vmcnt(0)).
  static DEV void load_trip(const DevColumns& C, int64_t w0, bool active, int64_t n, int lane, COLV (&col)[U], uint32_t (&cv)[U]) {
    const int64_t r0 = active ? w0 * 64 : 0;
    const int64_t left = n - r0;
    const uint32_t last = (!active || left <= 0) ? 0u : (uint32_t)((left < (int64_t)U * 64 ? left : (int64_t)U * 64) - 1);
#pragma unroll
    for (int u = 0; u < U; ++u) cv[u] = 0xFFFFFFFFu;
#pragma unroll
    for (int c = 0; c < BANK; ++c) {
      if (c < SIG::NCOL) {
        const uint64_t* p = (const uint64_t*)C.c[c].values + ((!active || left <= 0) ? 0 : r0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint32_t i = (uint32_t)(u * 64 + lane);
          col[u][c] = __builtin_nontemporal_load(p + (i < last ? i : last));
        }
      }
    }
  }
  static DEV int na(const DevTable&) { return SIG::NA; }
  static DEV uint8_t acc_kind(const DevTable&, int a) { return SIG::acc(a); }
  static DEV uint8_t xform(const DevTable&, int a) { return SIG::xf(a); }
  static DEV void eval(const DevProgram&, const DevFastPlan&, const COLV&, uint32_t, u64x16&, uint32_t&, bool,
                       uint32_t&, const PREP& = PREP()) {}
  static DEV bool pass(const DevProgram&, const DevFastPlan& F, uint8_t, const COLV& cur, uint32_t, const u64x16&,
                       uint32_t, const PREP& = PREP()) {
    uint64_t ok = ~0ull;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i < SIG::NP)
        ok &= cmp_mask(SIG::term_cls(i), cur[SIG::term_col(i)], F.term_imm[i], (uint32_t)F.term[i].m) ^
              (F.term[i].inv ? ~0ull : 0ull);
    }
    return lane_of_mask(ok);
  }
  // Comparison operators as COMPILE-TIME constants.  With run-time masks a term costs three v_cmp (less / equal / greater)
  // and a dozen scalar selects per row group; the operators of a query do not change while it runs, so the scan kernels
  // pick one of a few loop bodies once per wave: FORM = m0 | m1 << 3 | ... (three-way masks of the terms, none inverted),
  // 0 = the run-time form above.  form_of() names the FORM a plan matches (0 if none is instantiated).
  static __host__ __device__ inline int form_of(const DevFastPlan& F) {
    int form = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i < SIG::NP) {
        if (F.term[i].inv || F.term[i].m < 1 || F.term[i].m > 6) return 0;
        form |= (int)F.term[i].m << (3 * i);
      }
    return form;
  }
  template <int M, typename TT>
  static DEV bool cmp_ct(TT a, TT b) {
    if constexpr (M == 1) return a < b;
    else if constexpr (M == 2) return a == b;
    else if constexpr (M == 3) return a <= b;
    else if constexpr (M == 4) return a > b;
    else if constexpr (M == 5) return a < b || a > b;
    else return a >= b;
  }
  template <int M>
  static DEV bool term_ct(uint8_t t, uint64_t x, uint64_t y) {
    if (t == T_F64) return cmp_ct<M, double>(as_f64(x), as_f64(y));
    if (t == T_F32) return cmp_ct<M, float>(as_f32(x), as_f32(y));
    if (t == T_U64) return cmp_ct<M, uint64_t>(x, y);
    return cmp_ct<M, int64_t>((int64_t)x, (int64_t)y);
  }
  template <int FORM>
  static DEV bool pass_form(const DevProgram& P, const DevFastPlan& F, uint8_t pred, const COLV& cur, uint32_t cv, const u64x16& reg, uint32_t rv, const PREP& = PREP()) {
    if constexpr (FORM == 0) {
      return pass(P, F, pred, cur, cv, reg, rv);
    } else {
      bool ok = true;
      if constexpr (SIG::NP > 0) ok = ok && term_ct<((FORM >> 0) & 7) ? ((FORM >> 0) & 7) : 1>(SIG::term_cls(0), cur[SIG::term_col(0)], F.term_imm[0]);
      if constexpr (SIG::NP > 1) ok = ok && term_ct<((FORM >> 3) & 7) ? ((FORM >> 3) & 7) : 1>(SIG::term_cls(1), cur[SIG::term_col(1)], F.term_imm[1]);
      if constexpr (SIG::NP > 2) ok = ok && term_ct<((FORM >> 6) & 7) ? ((FORM >> 6) & 7) : 1>(SIG::term_cls(2), cur[SIG::term_col(2)], F.term_imm[2]);
      if constexpr (SIG::NP > 3) ok = ok && term_ct<((FORM >> 9) & 7) ? ((FORM >> 9) & 7) : 1>(SIG::term_cls(3), cur[SIG::term_col(3)], F.term_imm[3]);
      return ok;
    }
  }
  // ... as the wave's lane mask: one ballot per TERM, the conjunction on the scalar unit.  (The ballot of `a && b` is lowered as
  // v_cndmask + v_cmp_ne on top of the two compares: each compare can write its SGPR pair directly.)
  template <int FORM>
  static DEV uint64_t pass_mask(const DevProgram& P, const DevFastPlan& F, uint8_t pred, const COLV& cur, uint32_t cv, const u64x16& reg, uint32_t rv, const PREP& = PREP()) {
    if constexpr (FORM == 0) {
      return __ballot(pass(P, F, pred, cur, cv, reg, rv));
    } else {
      uint64_t ok = ~0ull;
      if constexpr (SIG::NP > 0) ok &= __ballot(term_ct<((FORM >> 0) & 7) ? ((FORM >> 0) & 7) : 1>(SIG::term_cls(0), cur[SIG::term_col(0)], F.term_imm[0]));
      if constexpr (SIG::NP > 1) ok &= __ballot(term_ct<((FORM >> 3) & 7) ? ((FORM >> 3) & 7) : 1>(SIG::term_cls(1), cur[SIG::term_col(1)], F.term_imm[1]));
      if constexpr (SIG::NP > 2) ok &= __ballot(term_ct<((FORM >> 6) & 7) ? ((FORM >> 6) & 7) : 1>(SIG::term_cls(2), cur[SIG::term_col(2)], F.term_imm[2]));
      if constexpr (SIG::NP > 3) ok &= __ballot(term_ct<((FORM >> 9) & 7) ? ((FORM >> 9) & 7) : 1>(SIG::term_cls(3), cur[SIG::term_col(3)], F.term_imm[3]));
      return ok;
    }
  }
  static DEV uint64_t key(const DevProgram&, const DevFastPlan&, uint8_t, int k, const COLV& cur, uint32_t,
                          const u64x16&, uint32_t) {
    uint64_t v = 0;
#pragma unroll
    for (int j = 0; j < kMaxKeys; ++j)  // k is a compile-time constant after unrolling at the call site
      if (j < SIG::KW && j == k) v = cur[SIG::key_col(j)];
    return v;
  }
  // factor j of aggregate argument a: kind and column are compile-time constants after unrolling
  static DEV double sfactor(const DevFastPlan& F, int a, int j, const COLV& cur) {
    const double x = as_f64(cur[SIG::fc(a, j) & (BANK - 1)]);
    const double imm = as_f64(F.arg_imm[a][j]);
    switch (SIG::fk(a, j)) {
      case FF_RT:  // the operator is the plan's: wave-uniform, read from the kernel arguments
        switch (F.arg[a].f[j].kind) {
          case FF_IMM_MINUS_COL: return imm - x;
          case FF_COL_PLUS_IMM: return x + imm;
          case FF_COL_MINUS_IMM: return x - imm;
          default: return x * imm;
        }
      case FF_IMM_MINUS_COL: return imm - x;
      case FF_COL_PLUS_IMM: return x + imm;
      case FF_COL_MINUS_IMM: return x - imm;
      case FF_COL_TIMES_IMM: return x * imm;
      default: return x;
    }
  }
  static DEV void arg(const DevProgram&, const DevFastPlan& F, uint8_t, int a, const COLV& cur, uint32_t,
                      const u64x16&, uint32_t, uint64_t& v, bool& valid) {
    valid = true;
    v = 0;
#pragma unroll
    for (int j = 0; j < kMaxAggs; ++j) {
      if (j < SIG::NA && j == a) {
        if (SIG::arg_dyn(j)) {  // product of signature-defined factors, multiplied left to right
          double acc = sfactor(F, j, 0, cur);
          if (SIG::nf(j) > 1) acc = acc * sfactor(F, j, 1, cur);
          if (SIG::nf(j) > 2) acc = acc * sfactor(F, j, 2, cur);
          v = f64_bits(acc);
        } else {
          v = cur[SIG::arg_col(j)];
        }
      }
    }
  }
};

// ---------------------------------------------------------------------------------------------
// scan plan (DevScanPlan): the run-time shape family evaluated as DATA -- see dfx_device.hpp
// ---------------------------------------------------------------------------------------------
// A wave-uniform plan word in a VECTOR register.  Plan words read from the kernel arguments are scalar values, and a
// loop that keeps thirty of them alive runs out of scalar registers (the run-time decoded kernels carried 230-300 SGPR
// spills: a v_readlane per use); the vector file has room (128 registers per lane at sixteen waves per CU, the scan
// loops use ~60).  volatile: the copy stays where PlanPolicy::prepare puts it, before the scan loop (a pure asm was
// re-executed inside the loop, with its scalar sources alive across it: nothing gained).
DEV uint32_t plan_word(uint32_t s) {
  uint32_t v;
  asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(s));
  return v;
}
DEV uint64_t plan_word64(uint64_t s) { return (uint64_t)plan_word((uint32_t)s) | ((uint64_t)plan_word((uint32_t)(s >> 32)) << 32); }

template <int NCOL, bool NULLS>
struct PlanBank {
  uint64_t v[NCOL];               // the aligned 8 bytes that hold the row's element of every plan column
  uint32_t vb[NULLS ? NCOL : 1];  // NULLS: validity bytes -- the row's own byte (per-row loads), or lane j = byte j of the TRIP's
                                  // bitmap slice (trip loads: one load per column and trip, every row group picks its byte
                                  // with ds_bpermute)
  DEV uint64_t operator[](int c) const { return v[c]; }
};

template <int NCOL>
DEV uint64_t plan_sel(const u64x16& reg, uint32_t slot) {  // slot is wave-uniform: compare-and-select, no indexing
  uint64_t x = reg[0];
  if (NCOL > 1) x = slot == 1u ? reg[1] : x;
  if (NCOL > 2) x = slot == 2u ? reg[2] : x;
  if (NCOL > 3) x = slot == 3u ? reg[3] : x;
  return x;
}

// NCOL: plan column slots loaded per row (slots past the plan's n_cols repeat slot 0: a cache hit, no branch around a load);
// GENK bit 0 (W4): 4-byte columns (read as the aligned 8 bytes that hold the element, widened afterwards), bit 1 (NULLS):
// validity bitmaps (without either every column is 8 bytes wide and null-free: no widening, no validity loads);
// FIXED: one key in slot 0, one routed argument in slot 1 (the partitioned GROUP BY's one-value kernels).
// bit 2 (KEY4, FIXED kernels only): slot 0 -- the key -- is a 4-byte integer column and every other slot is 8 bytes wide: the
// key is read with a real 4-byte load (half the bytes of the aligned-pair form, no lane select) and extended in one operation.
// The reference's own GROUP BY fixtures have Int32 keys (aggregate.rs:1033-1127).
constexpr int kPlanW4 = 1, kPlanNulls = 2, kPlanKey4 = 4;
template <int NCOL, int U_, int GENK, bool FIXED>
struct PlanPolicy {
  static constexpr bool W4 = (GENK & kPlanW4) != 0, NULLS = (GENK & kPlanNulls) != 0, KEY4 = (GENK & kPlanKey4) != 0;
  static_assert(!(W4 && KEY4), "KEY4 is the special case of W4");
  static constexpr bool kIsStatic = false;
  static constexpr bool kHasTripLoad = true;
  static constexpr int kPredTerms = -1;
  static constexpr int U = U_;
  static constexpr int kStaticNa = FIXED ? 1 : 0;
  static_assert(!NULLS || U_ * 8 + 1 <= 64, "one lane per validity byte of a trip");
  typedef PlanBank<NCOL, NULLS> COLV;
  // The terms' plan words, one copy per lane (see plan_word): everything a term needs is a vector operand, the predicate is
  // evaluated without one scalar instruction besides the `t < np` guards.  Five registers per term: the range, and one word
  // of flags that v_bfe takes apart per use (a register per flag cost the ring kernels their 128-register budget):
  //   bit 0  image: negative values are complemented (f64)       bit 1  image: the sign bit is flipped (f64, i64)
  //   bit 2  complement the range test (NotEq, impossible terms) bit 3, 4  the term's column slot
  //   bit 5  the term's value for a null column value
  // ... and what the columns need, as per-LANE constants (the row groups of a trip start at multiples of 64 rows, so a lane's
  // half of an aligned pair, its validity bit and its byte of the trip's bitmap slice never change): widening and validity are
  // vector selects, no column word is decoded inside the loop.
  struct PREP {
    uint64_t nlo[kPlanTerms];   // -lo
    uint64_t span[kPlanTerms];
    uint32_t flags[kPlanTerms];
    uint32_t key_sext;                // KEY4: all ones: the key is a signed integer (sign-extended), else zero-extended
    uint32_t hi_half[W4 ? NCOL : 1];  // W4: all ones where this lane's 4-byte element is the HIGH word of its aligned pair
    uint32_t is4[W4 ? NCOL : 1];      //     all ones: a 4-byte column
    uint32_t sext[W4 ? NCOL : 1];     //     all ones: ... of signed integers
    uint32_t vshift[NULLS ? NCOL : 1];  // NULLS: (lane + bit_offset) & 7: the row's bit inside its validity byte
    uint32_t vlane[NULLS ? NCOL : 1];   //        ((lane + bit_offset) >> 3) * 4: ds_bpermute address of the row's byte in group 0 of a trip
  };
  static DEV void prepare(const DevProgram&, const DevFastPlan& F, PREP& W) {
#pragma unroll
    for (int t = 0; t < kPlanTerms; ++t) {
      const DevPlanTerm& T = F.scan.term[t];
      W.nlo[t] = plan_word64(0ull - T.lo);
      W.span[t] = plan_word64(T.span);
      W.flags[t] = plan_word((T.a != 0ull ? 1u : 0u) | ((T.b >> 63) ? 2u : 0u) | (T.inv ? 4u : 0u) | ((T.col & 3u) << 3) | (T.if_null ? 32u : 0u));
    }
    const uint32_t lane = (uint32_t)lane_id();
#pragma unroll
    for (int c = 0; c < NCOL; ++c) {
      const uint32_t meta = F.scan.col_meta[c];
      if constexpr (W4) {
        const bool four = (meta & 15u) == 2u;
        W.is4[c] = plan_word(four ? 0xFFFFFFFFu : 0u);
        W.sext[c] = plan_word((four && ((meta >> 8) & 3u) == PX_SEXT32) ? 0xFFFFFFFFu : 0u);
        W.hi_half[c] = (four && ((lane + ((meta >> 4) & 1u)) & 1u)) ? 0xFFFFFFFFu : 0u;
      }
      if constexpr (NULLS) {
        const uint32_t at = lane + ((meta >> 16) & 7u);
        W.vshift[c] = at & 7u;
        W.vlane[c] = (at >> 3) << 2;
      }
    }
    W.key_sext = KEY4 ? plan_word(((F.scan.col_meta[0] >> 8) & 3u) == PX_SEXT32 ? 0xFFFFFFFFu : 0u) : 0u;
    if constexpr (!W4) W.hi_half[0] = W.is4[0] = W.sext[0] = 0;
    if constexpr (!NULLS) W.vshift[0] = W.vlane[0] = 0;
  }
  static DEV int na(const DevTable& T) { return FIXED ? 1 : T.na; }
  static DEV uint8_t acc_kind(const DevTable& T, int a) { return T.acc_kind[a]; }
  static DEV uint8_t xform(const DevTable& T, int a) { return T.val_xform[a]; }

  // C is the plan's own binding (bind_scan_plan): values = the column base rounded down to 8 bytes, validity = the byte
  // that holds row 0's bit (or the block of 0xFF bytes), bit_offset = the column word (plan_col_meta)
  static DEV void load(const DevProgram&, const DevColumns& C, int64_t row, bool inb, COLV& col, uint32_t& cv) {
    cv = 0xFFFFFFFFu;  // (vb holds every row's own byte)
    const uint64_t r = inb ? (uint64_t)row : 0ull;
    if constexpr (!NULLS) col.vb[0] = 0;
#pragma unroll
    for (int c = 0; c < NCOL; ++c) {
      const uint32_t meta = (uint32_t)C.c[c].bit_offset;
      if constexpr (W4) {
        const uint64_t off = ((r + ((meta >> 4) & 1u)) << (meta & 15u)) & ~7ull;
        col.v[c] = __builtin_nontemporal_load((const uint64_t*)((const uint8_t*)C.c[c].values + off));
      } else if constexpr (KEY4) {
        if (c == 0)  // (values = the column base rounded down to 8 bytes; bit 4 of the column word: row 0 is the pair's second element)
          col.v[c] = (uint64_t)__builtin_nontemporal_load((const uint32_t*)C.c[c].values + ((meta >> 4) & 1u) + r);
        else
          col.v[c] = __builtin_nontemporal_load((const uint64_t*)C.c[c].values + r);
      } else {
        col.v[c] = __builtin_nontemporal_load((const uint64_t*)C.c[c].values + r);
      }
      if constexpr (NULLS) col.vb[c] = (uint32_t)C.c[c].validity[((r + ((meta >> 16) & 7u)) >> 3) & (0ull - (uint64_t)((meta >> 20) & 1u))];
    }
  }
  // one trip of U row groups from group w0: a scalar base per column, 32-bit lane offsets clamped to the batch's last row
  // (StaticPolicy::load_trip), every load unconditional.  Validity: ONE byte load per column and trip -- lane j takes byte j
  // of the trip's slice of the bitmap (U x 8 + 1 bytes at most) -- instead of one per row group: vector-memory instructions,
  // not bytes, are what a second load per group costs.  cv[u] = u tells eval which rows of the slice are its own.
  static DEV void load_trip(const DevColumns& C, int64_t w0, bool active, int64_t n, int lane, COLV (&col)[U], uint32_t (&cv)[U]) {
    const int64_t r0 = active ? w0 * 64 : 0;
    const int64_t left = n - r0;
    const bool none = !active || left <= 0;
    const uint32_t last = none ? 0u : (uint32_t)((left < (int64_t)U * 64 ? left : (int64_t)U * 64) - 1);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      cv[u] = (uint32_t)u;
      if constexpr (!NULLS) col[u].vb[0] = 0;
    }
#pragma unroll
    for (int c = 0; c < NCOL; ++c) {
      const uint32_t meta = (uint32_t)C.c[c].bit_offset;
      if constexpr (W4) {
        const uint32_t shift = meta & 15u, delta = (meta >> 4) & 1u;
        const uint8_t* p = (const uint8_t*)C.c[c].values + (none ? 0ull : ((uint64_t)r0 << shift));  // r0 is a multiple of 64: still 8-byte aligned
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint32_t i = (uint32_t)(u * 64 + lane);
          const uint32_t ic = i < last ? i : last;
          col[u].v[c] = __builtin_nontemporal_load((const uint64_t*)(p + (((ic + delta) << shift) & ~7u)));
        }
      } else if (KEY4 && c == 0) {
        const uint32_t* p = (const uint32_t*)C.c[c].values + ((meta >> 4) & 1u) + (none ? 0 : r0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint32_t i = (uint32_t)(u * 64 + lane);
          col[u].v[c] = (uint64_t)__builtin_nontemporal_load(p + (i < last ? i : last));
        }
      } else {
        const uint64_t* p = (const uint64_t*)C.c[c].values + (none ? 0 : r0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint32_t i = (uint32_t)(u * 64 + lane);
          col[u].v[c] = __builtin_nontemporal_load(p + (i < last ? i : last));
        }
      }
      if constexpr (NULLS) {
        const uint32_t vbit0 = (meta >> 16) & 7u;
        const uint32_t vmask = 0u - ((meta >> 20) & 1u);  // no bitmap: byte 0 of the ones block for every lane
        const uint8_t* pv = C.c[c].validity + ((none || vmask == 0u) ? 0ull : ((uint64_t)r0 >> 3));
        const uint32_t last_byte = (last + vbit0) >> 3;
        const uint32_t j = (uint32_t)lane < last_byte ? (uint32_t)lane : last_byte;
        const uint32_t bytes = (uint32_t)pv[j & vmask];
#pragma unroll
        for (int u = 0; u < U; ++u) col[u].vb[c] = bytes;
      }
    }
  }
  // widened values into reg[0 .. NCOL), validity bits into rv (bit c = slot c).  curv: the row group's index inside its trip
  // (load_trip), or ~0 (per-row loads).  Kernels that prepare() pass W; without it (never in the scan loops) the column words
  // are decoded here.
  static DEV void eval(const DevProgram& P, const DevFastPlan& F, const COLV& cur, uint32_t curv, u64x16& reg, uint32_t& rv, bool inb,
                       uint32_t& err) {
    PREP W;
    prepare(P, F, W);
    eval(P, F, cur, curv, reg, rv, inb, err, W);
  }
  static DEV void eval(const DevProgram&, const DevFastPlan& F, const COLV& cur, uint32_t curv, u64x16& reg, uint32_t& rv, bool,
                       uint32_t&, const PREP& W) {
    rv = 0xFFFFFFFFu;
    uint32_t valid = 0;
#pragma unroll
    for (int c = 0; c < NCOL; ++c) {
      uint64_t x = cur.v[c];
      if constexpr (W4) {
        const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
        const uint32_t w = (hi & W.hi_half[c]) | (lo & ~W.hi_half[c]);  // (an 8-byte column: hi_half = 0, w = lo)
        const uint32_t xh = (hi & ~W.is4[c]) | ((uint32_t)((int32_t)w >> 31) & W.sext[c]);
        x = ((uint64_t)xh << 32) | w;
        if (((F.scan.col_meta[c] >> 8) & 3u) == PX_F32) x = f64_bits((double)__uint_as_float(w));  // (wave-uniform, rare; exact)
      }
      if constexpr (KEY4) {
        if (c == 0) x = ((uint64_t)((uint32_t)((int32_t)(uint32_t)x >> 31) & W.key_sext) << 32) | (uint32_t)x;
      }
      if constexpr (NULLS) {
        uint32_t byte = cur.vb[c];
        if (curv != 0xFFFFFFFFu)  // the trip's slice: this row's byte sits in lane (group x 64 + lane + bit_offset) / 8
          byte = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(W.vlane[c] + (curv << 5)), (int)byte);
        valid |= ((byte >> W.vshift[c]) & 1u) << c;
      }
      reg[c] = x;
    }
    if constexpr (NULLS) rv = valid;
  }
  // value of the term's column: a bitwise select by the slot's bits (vector masks: no scalar compare, no SGPR pair per term)
  static DEV uint64_t term_value(const u64x16& reg, uint32_t flags) {
    const uint32_t m0 = (uint32_t)__builtin_amdgcn_sbfe((int)flags, 3, 1);  // all ones: slot bit 0
    const uint64_t M0 = ((uint64_t)m0 << 32) | m0;
    if constexpr (NCOL <= 2) {
      return reg[0] ^ ((reg[0] ^ reg[NCOL > 1 ? 1 : 0]) & M0);
    } else {
      const uint32_t m1 = (uint32_t)__builtin_amdgcn_sbfe((int)flags, 4, 1);
      const uint64_t M1 = ((uint64_t)m1 << 32) | m1;
      const uint64_t y0 = reg[0] ^ ((reg[0] ^ reg[1]) & M0);
      const uint64_t y1 = reg[2] ^ ((reg[2] ^ reg[NCOL > 3 ? 3 : 2]) & M0);
      return y0 ^ ((y0 ^ y1) & M1);
    }
  }
  static DEV bool pass(const DevProgram&, const DevFastPlan& F, uint8_t, const COLV&, uint32_t, const u64x16& reg, uint32_t rv,
                       const PREP& W) {
    uint32_t ok = 1u;
#pragma unroll
    for (int t = 0; t < kPlanTerms; ++t) {
      if (t < F.scan.np) {  // (wave-uniform; nothing but vector arithmetic inside)
        const uint32_t f = W.flags[t];
        const uint64_t x = term_value(reg, f);
        const uint32_t hi = (uint32_t)(x >> 32);
        const uint32_t neg = (uint32_t)((int32_t)hi >> 31) & (uint32_t)__builtin_amdgcn_sbfe((int)f, 0, 1);  // all ones: a negative f64
        const uint32_t img_hi = hi ^ (neg & 0x7FFFFFFFu) ^ ((f << 30) & 0x80000000u);
        const uint64_t img = ((uint64_t)img_hi << 32) | ((uint32_t)x ^ neg);
        uint32_t r = ((img + W.nlo[t]) <= W.span[t] ? 1u : 0u) ^ ((f >> 2) & 1u);
        if constexpr (NULLS) r = ((rv >> ((f >> 3) & 3u)) & 1u) ? r : ((f >> 5) & 1u);
        ok &= r;
      }
    }
    return ok != 0u;
  }
  static DEV int form_of(const DevFastPlan&) { return 0; }
  template <int FORM>
  static DEV bool pass_form(const DevProgram& P, const DevFastPlan& F, uint8_t pred, const COLV& cur, uint32_t cv, const u64x16& reg, uint32_t rv, const PREP& W) {
    return pass(P, F, pred, cur, cv, reg, rv, W);
  }
  template <int FORM>
  static DEV uint64_t pass_mask(const DevProgram& P, const DevFastPlan& F, uint8_t pred, const COLV& cur, uint32_t cv, const u64x16& reg, uint32_t rv, const PREP& W) {
    return __ballot(pass(P, F, pred, cur, cv, reg, rv, W));
  }
  static DEV uint64_t key(const DevProgram&, const DevFastPlan& F, uint8_t, int k, const COLV&, uint32_t, const u64x16& reg, uint32_t) {
    if constexpr (FIXED) return reg[0];
    uint64_t v = 0;
#pragma unroll
    for (int j = 0; j < kMaxKeys; ++j)  // k is a compile-time constant after unrolling at the call site
      if (j == k) v = plan_sel<NCOL>(reg, F.scan.keyslot[j]);
    return v;
  }
  static DEV void arg(const DevProgram&, const DevFastPlan& F, uint8_t, int a, const COLV&, uint32_t, const u64x16& reg, uint32_t rv,
                      uint64_t& v, bool& valid) {
    if constexpr (FIXED) {
      v = reg[NCOL > 1 ? 1 : 0];
      valid = (NULLS && F.scan.count_valid) ? ((rv >> 1) & 1u) != 0u : true;
    } else {
      v = 0;
      valid = true;
#pragma unroll
      for (int j = 0; j < kMaxAggs; ++j) {
        if (j == a) {
          const uint32_t slot = F.scan.argslot[j];
          v = plan_sel<NCOL>(reg, slot);
          if (NULLS && F.scan.count_valid) valid = ((rv >> slot) & 1u) != 0u;
        }
      }
    }
  }
  // the argument in plan column `slot` (wave-uniform, run-time: the pair scan's second operand, DevPartition::pair_slot1)
  static DEV void arg_slot(const DevFastPlan& F, uint32_t slot, const u64x16& reg, uint32_t rv, uint64_t& v, bool& valid) {
    v = plan_sel<NCOL>(reg, slot);
    valid = (NULLS && F.scan.count_valid) ? ((rv >> slot) & 1u) != 0u : true;
  }
};
template <int NCOL, int U, int GENK> using PlanPolicyN = PlanPolicy<NCOL, U, GENK, false>;  // any keys / arguments
template <int NCOL, int U, int GENK> using PlanPolicy1 = PlanPolicy<NCOL, U, GENK, true>;   // key in slot 0, one routed value in slot 1

// the column loads of one trip of U row groups starting at group w0 (wave-uniform); `active` false: nothing is needed (the
// software pipeline's prefetch past the end), the loads still happen (row 0) so that their count stays fixed
template <typename POL>
DEV void load_trip(const DevProgram& P, const DevColumns& C, int64_t w0, bool active, int64_t n, int lane,
                   typename POL::COLV (&col)[POL::U], uint32_t (&cv)[POL::U]) {
  if constexpr (POL::kHasTripLoad) {
    POL::load_trip(C, w0, active, n, lane, col, cv);
  } else {
#pragma unroll
    for (int u = 0; u < POL::U; ++u) {
      const int64_t row = (w0 + u) * 64 + lane;
      POL::load(P, C, row, active && row < n, col[u], cv[u]);
    }
  }
}

// signature list: index == sig id handed to the launchers; -1: none
template <int BANK, int U> using PolPred2F64 = StaticPolicy<BANK, U, SigPred2F64>;

// ---------------------------------------------------------------------------------------------
// accumulator algebra (one 64-bit word per (group, aggregate))
// ---------------------------------------------------------------------------------------------
// order-preserving u64 image of an f64; NaN is canonicalised so that MIN and MAX ignore it unless
// every value is NaN (f64::min / f64::max, aggregate.rs:136-141 / :205-210)
DEV uint64_t f64_ordered(double d, bool for_min) {
  uint64_t b = f64_bits(d);
  if (d != d) b = for_min ? 0x7FF8000000000000ull : 0xFFF8000000000000ull;
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__host__ __device__ inline double f64_from_ordered(uint64_t u) {
  const uint64_t b = (u >> 63) ? (u & 0x7FFFFFFFFFFFFFFFull) : ~u;
  union { uint64_t u; double d; } c;
  c.u = b;
  return c.d;
}

DEV uint64_t transform_value(uint8_t xf, uint64_t v, bool valid) {
  switch (xf) {
    case VT_F64_ORD_MIN: return f64_ordered(as_f64(v), true);
    case VT_F64_ORD_MAX: return f64_ordered(as_f64(v), false);
    case VT_F32_ORD_MIN: return f64_ordered((double)as_f32(v), true);
    case VT_F32_ORD_MAX: return f64_ordered((double)as_f32(v), false);
    case VT_COUNT_VALID: return valid ? 1ull : 0ull;
    default: return v;
  }
}

// non-atomic combine (thread-private / shuffle reductions)
DEV uint64_t acc_combine(uint8_t kind, uint64_t a, uint64_t b) {
  switch (kind) {
    case ACC_ADD_F64: return f64_bits(as_f64(a) + as_f64(b));
    case ACC_ADD_F32: return f32_bits(as_f32(a) + as_f32(b));
    case ACC_ADD_U64: return a + b;
    case ACC_MIN_S64: return (uint64_t)(((int64_t)a < (int64_t)b) ? (int64_t)a : (int64_t)b);
    case ACC_MAX_S64: return (uint64_t)(((int64_t)a > (int64_t)b) ? (int64_t)a : (int64_t)b);
    case ACC_MIN_U64: return a < b ? a : b;
    default: return a > b ? a : b;
  }
}

// one hardware atomic, result unused (no-return form); works on global and LDS addresses
DEV void acc_atomic(uint8_t kind, uint64_t* p, uint64_t v) {
  switch (kind) {
    case ACC_ADD_F64: unsafeAtomicAdd((double*)p, as_f64(v)); break;
    case ACC_ADD_F32: unsafeAtomicAdd((float*)p, as_f32(v)); break;
    case ACC_ADD_U64: atomicAdd((unsigned long long*)p, (unsigned long long)v); break;
    case ACC_MIN_S64: atomicMin((long long*)p, (long long)v); break;
    case ACC_MAX_S64: atomicMax((long long*)p, (long long)v); break;
    case ACC_MIN_U64: atomicMin((unsigned long long*)p, (unsigned long long)v); break;
    default: atomicMax((unsigned long long*)p, (unsigned long long)v); break;
  }
}

DEV uint64_t shfl_xor_u64(uint64_t v, int m) { return (uint64_t)__shfl_xor((unsigned long long)v, m, 64); }

// ---------------------------------------------------------------------------------------------
// group table: find-or-insert + atomic update
// ---------------------------------------------------------------------------------------------
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// returns false when the bounded probe sequence found neither the key nor a free slot
template <int KW>
DEV bool table_upsert_slot(const DevTable& T, const uint64_t (&key)[KW], uint64_t h, uint64_t& slot_out,
                           bool& inserted) {
  inserted = false;
  // The home slot is the base of the ALIGNED GROUP OF FOUR the hash points into (probing stays linear from there).
  // Pass 2 of the partitioned strategy examines a whole group per step (two 16-byte LDS reads) and a wave pays for its
  // slowest lane: with homes anywhere in the group a key often spills into the next group, with group-base homes a
  // 4-slot bucket at load <= 0.5 rarely overflows (measured: -10 % VALU, -6.5 % LDS instructions in pass 2).  The
  // global kernels pay ~1.5 more single-slot probes per lookup, which their atomic rate (24 G/s) hides.
  uint64_t slot = ((h >> T.shift) & T.mask) & ~3ull;
  if (KW == 1) {
    if (key[0] == kEmptyKey) {  // the one key that collides with the claim sentinel owns slot `cap`
      slot_out = T.mask + 1;
      if (__hip_atomic_load(&T.ctrl[CTRL_SENTINEL], RLX_AGENT) == 0u) {
        if (atomicExch(&T.ctrl[CTRL_SENTINEL], 1u) == 0u) inserted = true;
      }
      return true;
    }
    for (int p = 0; p < T.max_probe; ++p) {
      const uint64_t k = __hip_atomic_load(&T.keys[slot], RLX_AGENT);
      if (k == key[0]) {
        slot_out = slot;
        return true;
      }
      if (k == kEmptyKey) {
        const uint64_t old = atomicCAS((unsigned long long*)&T.keys[slot], (unsigned long long)kEmptyKey,
                                       (unsigned long long)key[0]);
        if (old == kEmptyKey) {
          inserted = true;
          slot_out = slot;
          return true;
        }
        if (old == key[0]) {
          slot_out = slot;
          return true;
        }
      }
      slot = (slot & ~(uint64_t)T.block_mask) | ((slot + 1) & (uint64_t)T.block_mask);
    }
    return false;
  } else {
    // multi-word keys: claim the slot's state word (0 empty -> 1 busy), publish the key words
    // write-through, drain, then state = 2.  A lane that meets a busy slot re-reads it on its next
    // loop trip (structured loop: the claimer never waits on anybody, so no SIMT deadlock).
    int spins = 0;
    for (int p = 0; p < T.max_probe;) {
      uint32_t st = __hip_atomic_load(&T.state[slot], RLX_AGENT);
      if (st == 0u) {
        const uint32_t old = atomicCAS(&T.state[slot], 0u, 1u);
        if (old == 0u) {
#pragma unroll
          for (int w = 0; w < KW; ++w) __hip_atomic_store(&T.keys[(uint64_t)w * T.stride + slot], key[w], RLX_AGENT);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __hip_atomic_store(&T.state[slot], 2u, RLX_AGENT);
          inserted = true;
          slot_out = slot;
          return true;
        }
        st = old;
      }
      if (st == 1u) {
        if (++spins > (1 << 20)) return false;
        continue;
      }
      bool same = true;
#pragma unroll
      for (int w = 0; w < KW; ++w)
        same = same && (__hip_atomic_load(&T.keys[(uint64_t)w * T.stride + slot], RLX_AGENT) == key[w]);
      if (same) {
        slot_out = slot;
        return true;
      }
      slot = (slot & ~(uint64_t)T.block_mask) | ((slot + 1) & (uint64_t)T.block_mask);
      ++p;
    }
    return false;
  }
}

// append one row (keys + accumulator operands) to the spill list; wave-aggregated cursor bump
template <int KW>
DEV void spill_row(const DevTable& T, const DevRows& spill, bool do_spill, const uint64_t (&key)[KW],
                   const uint64_t (&val)[kMaxAggs]) {
  const uint64_t m = __ballot(do_spill);
  if (m == 0) return;
  const int lane = lane_id();
  const int leader = __ffsll((unsigned long long)m) - 1;
  uint64_t base = 0;
  if (lane == leader)
    base = atomicAdd((unsigned long long*)&T.ctrl[CTRL_SPILL_LO], (unsigned long long)__popcll(m));
  base = __shfl(base, leader, 64);
  if (do_spill) {
    const uint64_t pos = base + (uint64_t)__popcll(m & ((1ull << lane) - 1ull));
    if (pos < spill.capacity) {
#pragma unroll
      for (int w = 0; w < KW; ++w) spill.words[(uint64_t)w * spill.capacity + pos] = key[w];
#pragma unroll
      for (int a = 0; a < kMaxAggs; ++a)  // static indices: val[] must stay in registers
        if (a < T.na) spill.words[(uint64_t)(KW + a) * spill.capacity + pos] = val[a];
    }
  }
}

// statistics counter of this workgroup's stripe (see DevTable::stats)
DEV void stat_add(const DevTable& T, int which, uint64_t v) {
  if (T.stats && v) atomicAdd((unsigned long long*)&T.stats[(size_t)(blockIdx.x % kStatStripes) * STAT_WORDS + which], (unsigned long long)v);
}

// the group whose (single-word) key equals the claim sentinel lives in slot `cap`: the slice of table_apply<1>
// that such a row takes, as a small function of its own (the partition kernels call it from unrolled loops)
DEV void sentinel_apply(const DevTable& T, const uint64_t (&val)[kMaxAggs]) {
  bool inserted = false;
  if (__hip_atomic_load(&T.ctrl[CTRL_SENTINEL], RLX_AGENT) == 0u)
    inserted = atomicExch(&T.ctrl[CTRL_SENTINEL], 1u) == 0u;
  const uint64_t mi = __ballot(inserted);
  if (mi != 0 && lane_id() == __ffsll((unsigned long long)mi) - 1) atomicAdd(&T.ctrl[CTRL_OCCUPIED], (uint32_t)__popcll(mi));
  const uint64_t slot = T.mask + 1;
#pragma unroll
  for (int a = 0; a < kMaxAggs; ++a)
    if (a < T.na) acc_atomic(T.acc_kind[a], &T.accs[(uint64_t)a * T.stride + slot], val[a]);
}

template <int KW>
DEV bool table_apply(const DevTable& T, const uint64_t (&key)[KW], const uint64_t (&val)[kMaxAggs]) {
  uint64_t slot = 0;
  bool inserted = false;
  const bool ok = table_upsert_slot<KW>(T, key, hash_keys<KW>(key), slot, inserted);
  // new groups are counted ONCE PER WAVE: thousands of lanes bumping the same word serialise at
  // ~11 ns per atomic (MI355X_MICROARCH.md, fanin) -- 1 M inserts would cost 11 ms
  const uint64_t mi = __ballot(ok && inserted);
  if (mi != 0 && lane_id() == __ffsll((unsigned long long)mi) - 1) atomicAdd(&T.ctrl[CTRL_OCCUPIED], (uint32_t)__popcll(mi));
  if (!ok) return false;
#pragma unroll
  for (int a = 0; a < kMaxAggs; ++a)
    if (a < T.na) acc_atomic(T.acc_kind[a], &T.accs[(uint64_t)a * T.stride + slot], val[a]);
  return true;
}

}  // namespace dfx
