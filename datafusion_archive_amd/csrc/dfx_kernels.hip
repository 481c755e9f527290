// dfx_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the filter / projection /
// aggregate path.  Memory-bound integer/f64 work: no MFMA.  Design rules used throughout:
//   * one wave owns 64 consecutive rows at a time, so every global access is a fully coalesced
//     512-byte (8 B/lane) request and `__ballot` of a per-row predicate IS the Arrow LSB-first
//     bitmap word of those rows;
//   * expression intermediates live in VGPR register files indexed by wave-uniform indices
//     (s_set_gpr_idx), never in memory; literals come from the kernarg segment by scalar load;
//   * all inter-workgroup state (group table, counters) is touched only with agent-scope atomics:
//     per-XCD L2s are not coherent, atomics are (MI355X_MICROARCH.md, inter-workgroup visibility);
//   * grids are sized to a few resident workgroups per CU on 256 CUs and stride over tiles.
// Compile with -ffp-contract=off (the reference never fuses a*b+c) and -munsafe-fp-atomics
// (hardware global_atomic_add_f64 / ds_add_f64 instead of CAS loops).
#include "dfx_kernels.hpp"

#include <hip/hip_runtime.h>

#include <mutex>
#include <string>
#include <vector>

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

template <int KW>
__host__ __device__ inline uint64_t hash_keys(const uint64_t* key) {
  uint64_t h = mix64(key[0] + 0x9E3779B97F4A7C15ull);
#pragma unroll
  for (int w = 1; w < KW; ++w) h = mix64(h ^ (key[w] + 0x9E3779B97F4A7C15ull * (uint64_t)(w + 1)));
  return h;
}

uint64_t host_hash_keys(const uint64_t* key, int kw) {
  switch (kw) {
    case 1: return hash_keys<1>(key);
    case 2: return hash_keys<2>(key);
    case 3: return hash_keys<3>(key);
    default: return hash_keys<4>(key);
  }
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

// ---------------------------------------------------------------------------------------------
// the per-row expression interpreter
// ---------------------------------------------------------------------------------------------
// NOTE: the register files are separate local vector VALUES (never members of a struct that is
// passed by reference): that is what lets SROA keep them in VGPRs and lower the wave-uniform
// dynamic indices to s_set_gpr_idx instead of scratch memory.
#define ROWSTATE_DECL(s) u64x8 s##_col; u64x16 s##_reg; uint32_t s##_colvalid, s##_regvalid
#define ROWSTATE_ARGS(s) s##_col, s##_reg, s##_colvalid, s##_regvalid
#define ROWSTATE_PARAMS u64x8& s_col, u64x16& s_reg, uint32_t& s_colvalid, uint32_t& s_regvalid
#define ROWSTATE_CPARAMS const u64x8& s_col, const u64x16& s_reg, const uint32_t& s_colvalid, const uint32_t& s_regvalid

// issue every column load of this row back to back (independent loads, all in flight together)
template <bool NULLS>
DEV void load_columns(const DevProgram& P, const DevColumns& C, int64_t row, bool inb, ROWSTATE_PARAMS) {
  s_colvalid = 0xFFFFFFFFu;
#pragma unroll
  for (int c = 0; c < kMaxCols; ++c) {
    if (c < P.n_cols) {
      uint64_t v = 0;
      if (inb) v = load_canonical(P.col_dtype[c], C.c[c].values, row, C.c[c].bit_offset);
      s_col[c] = v;
      if (NULLS) {
        if (C.c[c].validity != nullptr) {
          const bool ok = inb ? get_bit(C.c[c].validity, C.c[c].bit_offset + row) : false;
          if (!ok) s_colvalid &= ~(1u << c);
        }
      }
    }
  }
}

DEV void fetch(const DevProgram& P, ROWSTATE_CPARAMS, uint8_t opnd, uint64_t& v, bool& valid) {
  const int idx = opnd & 63;
  const int kind = opnd >> 6;
  if (kind == OPK_REG) {
    v = s_reg[idx];
    valid = (s_regvalid >> idx) & 1;
  } else if (kind == OPK_COL) {
    v = s_col[idx & 7];
    valid = (s_colvalid >> idx) & 1;
  } else {
    v = P.imm[idx & (kMaxImm - 1)];
    valid = true;
  }
}

// Executes the SSA program for one row.  Semantics per op are arrow 0.12 array_ops, the same
// table the oracle restates (oracle/dfx_oracle.c: compare_arrays / boolean_arrays / math_arrays).
DEV void run_program(const DevProgram& P, ROWSTATE_PARAMS, bool active, uint32_t& err) {
  s_regvalid = 0;
  for (int pc = 0; pc < P.n_ins; ++pc) {
    const DevIns in = P.ins[pc];
    const uint8_t t = in.t;
    uint64_t x, y = 0;
    bool vx, vy = true;
    fetch(P, ROWSTATE_ARGS(s), in.a, x, vx);
    if (in.op != DOP_CAST) fetch(P, ROWSTATE_ARGS(s), in.b, y, vy);
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
      res = cast_value(t, in.b, x);
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
        default:
          if (y == 0) {
            if (v && active) err |= 1u;
            o = 0;
          } else if (is_signed_int(t)) {
            const uint64_t mn = wrap_to(t, 1ull << (t == T_I8 ? 7 : t == T_I16 ? 15 : t == T_I32 ? 31 : 63));
            if ((int64_t)y == -1 && x == mn) {
              if (v && active) err |= 2u;
              o = x;
            } else {
              o = (uint64_t)((int64_t)x / (int64_t)y);
            }
          } else {
            o = x / y;
          }
          break;
      }
      res = wrap_to(t, o);
    }
    s_reg[pc] = res;
    s_regvalid |= (v ? 1u : 0u) << pc;
  }
}

DEV bool eval_predicate(const DevProgram& P, ROWSTATE_CPARAMS, uint8_t pred) {
  if (pred == kNoOperand) return true;
  uint64_t v;
  bool valid;
  fetch(P, ROWSTATE_ARGS(s), pred, v, valid);
  // FilterRelation reads filter.value(i): the raw value bit; a null slot holds false (filter.rs:86)
  return valid && (v & 1);
}

// ---------------------------------------------------------------------------------------------
// K1 predicate_mask
// ---------------------------------------------------------------------------------------------
template <bool NULLS>
__global__ __launch_bounds__(kBlock) void k_predicate_mask(const DevProgram P, const DevColumns C,
                                                           const uint8_t pred, const int64_t n,
                                                           uint64_t* __restrict__ mask_words,
                                                           uint32_t* __restrict__ tile_counts,
                                                           uint32_t* __restrict__ ctrl) {
  __shared__ uint32_t wave_cnt[kBlock / 64];
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int64_t n_words = (n + 63) >> 6;
  const int64_t n_tiles = (n + kTileRows - 1) / kTileRows;
  uint32_t err = 0;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    uint32_t cnt = 0;
#pragma unroll 2
    for (int i = 0; i < 16; ++i) {
      const int64_t w = tile * 64 + wave * 16 + i;
      const int64_t row = w * 64 + lane;
      const bool inb = row < n;
      ROWSTATE_DECL(s);
      load_columns<NULLS>(P, C, row, inb, ROWSTATE_ARGS(s));
      run_program(P, ROWSTATE_ARGS(s), inb, err);
      const bool pass = inb && eval_predicate(P, ROWSTATE_ARGS(s), pred);
      const uint64_t word = __ballot(pass);
      if (lane == 0 && w < n_words) mask_words[w] = word;
      cnt += (uint32_t)__popcll(word);
    }
    if (tile_counts != nullptr) {
      if (lane == 0) wave_cnt[wave] = cnt;
      __syncthreads();
      if (threadIdx.x == 0) tile_counts[tile] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
      __syncthreads();
    }
  }
  if (err) atomicOr(&ctrl[CTRL_ERROR], err);
}

// ---------------------------------------------------------------------------------------------
// exclusive scans (tile counts -> offsets; string lengths -> offsets)
// ---------------------------------------------------------------------------------------------
constexpr int kScanChunk = 4096;  // elements per block (256 threads x 16)

DEV uint64_t wave_inclusive_scan(uint64_t v) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint64_t o = __shfl_up(v, d, 64);
    if (lane_id() >= d) v += o;
  }
  return v;
}

// block-wide exclusive scan of one value per thread; returns the exclusive prefix, *total = block sum
DEV uint64_t block_exclusive_scan(uint64_t v, uint64_t* total, uint64_t* lds4) {
  const uint64_t inc = wave_inclusive_scan(v);
  const int wave = threadIdx.x >> 6;
  if (lane_id() == 63) lds4[wave] = inc;
  __syncthreads();
  uint64_t base = 0;
  for (int w = 0; w < wave; ++w) base += lds4[w];
  *total = lds4[0] + lds4[1] + lds4[2] + lds4[3];
  __syncthreads();
  return base + inc - v;
}

template <typename TIN>
__global__ __launch_bounds__(kBlock) void k_scan_local(const TIN* __restrict__ in, int64_t n,
                                                       uint64_t* __restrict__ block_sums) {
  __shared__ uint64_t lds4[4];
  const int64_t base = (int64_t)blockIdx.x * kScanChunk + (int64_t)threadIdx.x * 16;
  uint64_t sum = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i)
    if (base + i < n) sum += (uint64_t)in[base + i];
  uint64_t total;
  block_exclusive_scan(sum, &total, lds4);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(kBlock) void k_scan_sums(uint64_t* __restrict__ block_sums, int64_t nb) {
  // single block: exclusive scan of block_sums in place; block_sums[nb] = grand total
  __shared__ uint64_t lds4[4];
  uint64_t carry = 0;
  for (int64_t b0 = 0; b0 < nb; b0 += kBlock) {
    const int64_t i = b0 + threadIdx.x;
    const uint64_t v = i < nb ? block_sums[i] : 0;
    uint64_t total;
    const uint64_t ex = block_exclusive_scan(v, &total, lds4);
    if (i < nb) block_sums[i] = carry + ex;
    carry += total;
  }
  if (threadIdx.x == 0) block_sums[nb] = carry;
}

template <typename TIN, typename TOUT>
__global__ __launch_bounds__(kBlock) void k_scan_apply(const TIN* __restrict__ in, int64_t n,
                                                       const uint64_t* __restrict__ block_sums,
                                                       int64_t nb, TOUT* __restrict__ out) {
  __shared__ uint64_t lds4[4];
  const int64_t base = (int64_t)blockIdx.x * kScanChunk + (int64_t)threadIdx.x * 16;
  uint64_t v[16];
  uint64_t sum = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    v[i] = (base + i < n) ? (uint64_t)in[base + i] : 0;
    sum += v[i];
  }
  uint64_t total;
  uint64_t run = block_sums[blockIdx.x] + block_exclusive_scan(sum, &total, lds4);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    if (base + i < n) out[base + i] = (TOUT)run;
    run += v[i];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = (TOUT)block_sums[nb];
}

// ---------------------------------------------------------------------------------------------
// K4 compact
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void k_compact(const T* __restrict__ in,
                                                    const uint64_t* __restrict__ mask_words,
                                                    const uint64_t* __restrict__ tile_offsets,
                                                    const int64_t n, T* __restrict__ out) {
  __shared__ uint32_t word_off[64];
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int64_t n_words = (n + 63) >> 6;
  const int64_t n_tiles = (n + kTileRows - 1) / kTileRows;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    if (wave == 0) {  // popcount of each of the tile's 64 words, exclusive-scanned by one wave
      const int64_t w = tile * 64 + lane;
      const uint32_t c = w < n_words ? (uint32_t)__popcll(mask_words[w]) : 0u;
      const uint64_t inc = wave_inclusive_scan((uint64_t)c);
      word_off[lane] = (uint32_t)(inc - c);
    }
    __syncthreads();
    const uint64_t tile_base = tile_offsets[tile];
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
      const int wi = wave * 16 + i;
      const int64_t w = tile * 64 + wi;
      if (w < n_words) {
        const uint64_t word = mask_words[w];  // wave-uniform address: one request
        const int64_t row = w * 64 + lane;
        if ((word >> lane) & 1) {
          const uint32_t rank = (uint32_t)__popcll(word & ((1ull << lane) - 1ull));
          out[tile_base + word_off[wi] + rank] = in[row];
        }
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(kBlock) void k_utf8_lengths(const int32_t* __restrict__ offsets, int64_t n,
                                                         int32_t* __restrict__ lengths,
                                                         int32_t* __restrict__ starts) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const int32_t a = offsets[i], b = offsets[i + 1];
    lengths[i] = b - a;
    starts[i] = a;
  }
}

__global__ __launch_bounds__(kBlock) void k_utf8_gather(const uint8_t* __restrict__ data,
                                                        const int32_t* __restrict__ src_starts,
                                                        const int32_t* __restrict__ dst_offsets,
                                                        int64_t m, uint8_t* __restrict__ out) {
  // one 16-lane group per output string: lanes stride over the bytes
  const int64_t gid = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 4;
  const int sub = threadIdx.x & 15;
  const int64_t stride = ((int64_t)gridDim.x * kBlock) >> 4;
  for (int64_t i = gid; i < m; i += stride) {
    const int32_t s = src_starts[i], d = dst_offsets[i], len = dst_offsets[i + 1] - d;
    for (int32_t b = sub; b < len; b += 16) out[d + b] = data[s + b];
  }
}

// ---------------------------------------------------------------------------------------------
// K2/K3 project
// ---------------------------------------------------------------------------------------------
template <bool NULLS>
__global__ __launch_bounds__(kBlock) void k_project(const DevProgram P, const DevColumns C,
                                                    const DevProjectPlan plan, const int64_t n,
                                                    uint32_t* __restrict__ ctrl) {
  const int lane = lane_id();
  const int64_t n_words = (n + 63) >> 6;
  const int64_t wave_global = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * kBlock) >> 6;
  uint32_t err = 0;
  for (int64_t w = wave_global; w < n_words; w += n_waves) {
    const int64_t row = w * 64 + lane;
    const bool inb = row < n;
    ROWSTATE_DECL(s);
    load_columns<NULLS>(P, C, row, inb, ROWSTATE_ARGS(s));
    run_program(P, ROWSTATE_ARGS(s), inb, err);
#pragma unroll
    for (int o = 0; o < kMaxOut; ++o) {
      if (o < plan.n_out) {
        uint64_t v;
        bool valid;
        fetch(P, ROWSTATE_ARGS(s), plan.out[o], v, valid);
        const uint8_t t = plan.out_dtype[o];
        if (t == T_BOOL) {
          const uint64_t bits = __ballot(inb && (v & 1));
          if (lane == 0) ((uint64_t*)plan.out_values[o])[w] = bits;
        } else if (inb) {
          store_typed(t, plan.out_values[o], row, v);
        }
        if (plan.out_validity[o] != nullptr) {
          const uint64_t vb = __ballot(inb && valid);
          if (lane == 0) plan.out_validity[o][w] = vb;
        }
      }
    }
  }
  if (err) atomicOr(&ctrl[CTRL_ERROR], err);
}

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
// K5 reduce_all (ungrouped aggregates of one batch)
// ---------------------------------------------------------------------------------------------
// partial layout per aggregate a: partial[4a+0] accumulator word (pre-filled with the identity),
// [4a+1] number of valid arguments, [4a+2] min over (row << 1 | is_nan) of valid rows (u64::MAX
// when none): arrow 0.12 min/max scan with `<` / `>`, so a NaN in the first valid slot sticks.
template <bool NULLS>
__global__ __launch_bounds__(kBlock) void k_reduce(const DevProgram P, const DevColumns C,
                                                   const DevAggPlan plan, const DevTable T,
                                                   const int64_t n, uint64_t* __restrict__ partial,
                                                   uint32_t* __restrict__ ctrl) {
  __shared__ uint64_t lds[kBlock / 64][kMaxAggs * 3];
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int64_t n_words = (n + 63) >> 6;
  const int64_t wave_global = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * kBlock) >> 6;
  uint64_t acc[kMaxAggs], cnt[kMaxAggs], first[kMaxAggs];
#pragma unroll
  for (int a = 0; a < kMaxAggs; ++a) {
    acc[a] = T.acc_init[a];
    cnt[a] = 0;
    first[a] = ~0ull;
  }
  uint32_t err = 0;
  uint64_t passed = 0;
  for (int64_t w = wave_global; w < n_words; w += n_waves) {
    const int64_t row = w * 64 + lane;
    const bool inb = row < n;
    ROWSTATE_DECL(s);
    load_columns<NULLS>(P, C, row, inb, ROWSTATE_ARGS(s));
    run_program(P, ROWSTATE_ARGS(s), inb, err);
    const bool pass = inb && eval_predicate(P, ROWSTATE_ARGS(s), plan.pred);
    if (pass) {
      ++passed;
#pragma unroll
      for (int a = 0; a < kMaxAggs; ++a) {
        if (a < T.na) {
          uint64_t v;
          bool valid;
          fetch(P, ROWSTATE_ARGS(s), plan.arg[a], v, valid);
          const uint8_t xf = T.val_xform[a];
          if (xf == VT_COUNT_VALID) {
            acc[a] += valid ? 1ull : 0ull;
            cnt[a] += 1;
          } else if (valid) {  // array_ops::{min,max,sum} skip nulls
            if (xf != VT_RAW) {
              const double d = (xf == VT_F32_ORD_MIN || xf == VT_F32_ORD_MAX) ? (double)as_f32(v) : as_f64(v);
              const uint64_t tag = ((uint64_t)row << 1) | (d != d ? 1ull : 0ull);
              first[a] = tag < first[a] ? tag : first[a];
            }
            acc[a] = acc_combine(T.acc_kind[a], acc[a], transform_value(xf, v, valid));
            cnt[a] += 1;
          }
        }
      }
    }
  }
  // wave tree (xor butterfly), then one atomic per workgroup per word
#pragma unroll
  for (int a = 0; a < kMaxAggs; ++a) {
    if (a < T.na) {
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) {
        acc[a] = acc_combine(T.acc_kind[a], acc[a], shfl_xor_u64(acc[a], m));
        cnt[a] += shfl_xor_u64(cnt[a], m);
        const uint64_t of = shfl_xor_u64(first[a], m);
        first[a] = of < first[a] ? of : first[a];
      }
      if (lane == 0) {
        lds[wave][a * 3 + 0] = acc[a];
        lds[wave][a * 3 + 1] = cnt[a];
        lds[wave][a * 3 + 2] = first[a];
      }
    }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) passed += shfl_xor_u64(passed, m);
  __syncthreads();
  if (threadIdx.x < T.na) {
    const int a = threadIdx.x;
    uint64_t x = lds[0][a * 3], c = lds[0][a * 3 + 1], f = lds[0][a * 3 + 2];
    for (int w = 1; w < kBlock / 64; ++w) {
      x = acc_combine(T.acc_kind[a], x, lds[w][a * 3]);
      c += lds[w][a * 3 + 1];
      f = lds[w][a * 3 + 2] < f ? lds[w][a * 3 + 2] : f;
    }
    if (c) {
      acc_atomic(T.acc_kind[a], &partial[4 * a + 0], x);
      atomicAdd((unsigned long long*)&partial[4 * a + 1], (unsigned long long)c);
      atomicMin((unsigned long long*)&partial[4 * a + 2], (unsigned long long)f);
    }
  }
  if (lane == 0 && passed) atomicAdd((unsigned long long*)&ctrl[CTRL_PASSED_LO], (unsigned long long)passed);
  if (err) atomicOr(&ctrl[CTRL_ERROR], err);
}

// AccumulatorSet::accumulate_scalar for the batch scalars (aggregate.rs:107-145/:176-214/:245-283),
// executed by one thread; then re-arms the batch partials with their identities.
// func: 0 min, 1 max, 2 sum, 3 count.  state[2a] = has, state[2a+1] = value bits (canonical).
__global__ void k_reduce_fold(const DevTable T, const uint8_t* __restrict__ arg_dtype,
                              const uint8_t* __restrict__ func, uint64_t* __restrict__ partial,
                              uint64_t* __restrict__ state) {
  const int a = threadIdx.x;
  if (a >= T.na) return;
  const uint64_t accw = partial[4 * a + 0], cnt = partial[4 * a + 1], first = partial[4 * a + 2];
  partial[4 * a + 0] = T.acc_init[a];
  partial[4 * a + 1] = 0;
  partial[4 * a + 2] = ~0ull;
  const uint8_t t = arg_dtype[a], f = func[a];
  bool has = cnt != 0;
  uint64_t val = accw;
  if (f == 3) {
    has = true;  // deviation D3: COUNT of a batch is always Some(n)
  } else if (has && (t == T_F64 || t == T_F32) && f != 2) {
    double d = (first & 1) ? __longlong_as_double(0x7FF8000000000000ll) : f64_from_ordered(accw);
    val = (t == T_F64) ? f64_bits(d) : f32_bits((float)d);
  } else if (has && f == 2 && is_int(t)) {
    val = wrap_to(t, accw);
  }
  if (!has) return;  // Option::None: accumulator unchanged (or stays None)
  if (!state[2 * a]) {
    state[2 * a] = 1;
    state[2 * a + 1] = val;
    return;
  }
  const uint64_t cur = state[2 * a + 1];
  uint64_t out;
  if (f == 3) {
    out = cur + val;
  } else if (t == T_F64) {
    const double x = as_f64(cur), y = as_f64(val);
    out = f64_bits(f == 0 ? fmin(x, y) : f == 1 ? fmax(x, y) : x + y);
  } else if (t == T_F32) {
    const float x = as_f32(cur), y = as_f32(val);
    out = f32_bits(f == 0 ? fminf(x, y) : f == 1 ? fmaxf(x, y) : x + y);
  } else if (is_signed_int(t)) {
    const int64_t x = (int64_t)cur, y = (int64_t)val;
    out = f == 0 ? (uint64_t)(x < y ? x : y) : f == 1 ? (uint64_t)(x > y ? x : y) : wrap_to(t, cur + val);
  } else {
    out = f == 0 ? (cur < val ? cur : val) : f == 1 ? (cur > val ? cur : val) : wrap_to(t, cur + val);
  }
  state[2 * a + 1] = out;
}

// ---------------------------------------------------------------------------------------------
// group table: find-or-insert + atomic update
// ---------------------------------------------------------------------------------------------
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// returns false when the bounded probe sequence found neither the key nor a free slot
template <int KW>
DEV bool table_upsert_slot(const DevTable& T, const uint64_t (&key)[KW], uint64_t h, uint64_t& slot_out,
                           bool& inserted) {
  inserted = false;
  uint64_t slot = (h >> T.shift) & T.mask;
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
      slot = (slot + 1) & T.mask;
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
      slot = (slot + 1) & T.mask;
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
      for (int a = 0; a < T.na; ++a) spill.words[(uint64_t)(KW + a) * spill.capacity + pos] = val[a];
    }
  }
}

template <int KW>
DEV bool table_apply(const DevTable& T, const uint64_t (&key)[KW], const uint64_t (&val)[kMaxAggs]) {
  uint64_t slot;
  bool inserted;
  if (!table_upsert_slot<KW>(T, key, hash_keys<KW>(key), slot, inserted)) return false;
  if (inserted) atomicAdd(&T.ctrl[CTRL_OCCUPIED], 1u);
#pragma unroll
  for (int a = 0; a < kMaxAggs; ++a)
    if (a < T.na) acc_atomic(T.acc_kind[a], &T.accs[(uint64_t)a * T.stride + slot], val[a]);
  return true;
}

// ---------------------------------------------------------------------------------------------
// K6/K7 hash_agg with an LDS front cache
// ---------------------------------------------------------------------------------------------
// Dynamic LDS layout (all 8-byte words, base 16-byte aligned): keys[KW][S], accs[na][S].
// For KW > 1 an extra state[S] (uint32) follows.  S = plan.lds_slots (power of two), split into
// plan.lds_copies lane-replicated sub-tables so that few-group inputs (TPC-H Q1: <= 6 groups) do not
// serialise 64 lanes on one LDS address.
template <int KW, bool NULLS>
__global__ __launch_bounds__(kBlock) void k_hash_agg(const DevProgram P, const DevColumns C,
                                                     const DevAggPlan plan, const DevTable T,
                                                     const DevRows spill, const int64_t n) {
  extern __shared__ __attribute__((aligned(16))) uint64_t lds[];
  const int S = plan.lds_slots;
  uint64_t* lkeys = lds;
  uint64_t* laccs = lds + (size_t)KW * S;
  uint32_t* lstate = (uint32_t*)(lds + (size_t)(KW + T.na) * S);
  const int lane = lane_id();
  if (S > 0) {
    for (int i = threadIdx.x; i < S; i += kBlock) {
      lkeys[i] = kEmptyKey;
      if (KW > 1) lstate[i] = 0u;
      for (int a = 0; a < T.na; ++a) laccs[a * S + i] = T.acc_init[a];
    }
    __syncthreads();
  }
  const int sub_slots = S > 0 ? S / plan.lds_copies : 0;
  const int sub_base = S > 0 ? (lane & (plan.lds_copies - 1)) * sub_slots : 0;

  const int64_t n_words = (n + 63) >> 6;
  const int64_t wave_global = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * kBlock) >> 6;
  uint32_t err = 0;
  uint32_t lds_hit = 0, lds_miss = 0;
  uint64_t passed = 0;
  int iter = 0;
  bool saturated = false;
  for (int64_t w = wave_global; w < n_words; w += n_waves, ++iter) {
    if ((iter & 15) == 0) {  // wave-uniform, one request: has the table passed its load limit?
      saturated = __hip_atomic_load(&T.ctrl[CTRL_SATURATED], RLX_AGENT) != 0u;
      if (!saturated && (uint64_t)__hip_atomic_load(&T.ctrl[CTRL_OCCUPIED], RLX_AGENT) > T.load_limit) {
        saturated = true;
        if (lane == 0) __hip_atomic_store(&T.ctrl[CTRL_SATURATED], 1u, RLX_AGENT);
      }
    }
    const int64_t row = w * 64 + lane;
    const bool inb = row < n;
    ROWSTATE_DECL(s);
    load_columns<NULLS>(P, C, row, inb, ROWSTATE_ARGS(s));
    run_program(P, ROWSTATE_ARGS(s), inb, err);
    const bool pass = inb && eval_predicate(P, ROWSTATE_ARGS(s), plan.pred);
    uint64_t key[KW];
    uint64_t val[kMaxAggs];
#pragma unroll
    for (int k = 0; k < KW; ++k) {
      bool kvalid;
      fetch(P, ROWSTATE_ARGS(s), plan.key[k], key[k], kvalid);  // key nulls are not checked (aggregate.rs:807-852)
    }
#pragma unroll
    for (int a = 0; a < kMaxAggs; ++a) {
      val[a] = 0;
      if (a < T.na) {
        uint64_t v;
        bool valid;
        fetch(P, ROWSTATE_ARGS(s), plan.arg[a], v, valid);  // value(row) read blindly (aggregate.rs:561-603)
        val[a] = transform_value(T.val_xform[a], v, valid);
      }
    }
    passed += pass ? 1 : 0;
    bool todo = pass;
    // ---- LDS front cache ----
    if (S > 0 && todo && !(KW == 1 && key[0] == kEmptyKey)) {
      const uint64_t h = hash_keys<KW>(key);
      int slot = sub_base + (int)(h & (uint64_t)(sub_slots - 1));
      int found = -1;
      if (KW == 1) {
        for (int p = 0; p < 4 && found < 0; ++p) {
          const uint64_t k = lkeys[slot];
          if (k == key[0]) {
            found = slot;
          } else if (k == kEmptyKey) {
            const uint64_t old = atomicCAS((unsigned long long*)&lkeys[slot], (unsigned long long)kEmptyKey,
                                           (unsigned long long)key[0]);
            if (old == kEmptyKey || old == key[0]) found = slot;
          }
          if (found < 0) slot = sub_base + ((slot - sub_base + 1) & (sub_slots - 1));
        }
      } else {
        int spins = 0;
        for (int p = 0; p < 4 && found < 0;) {
          uint32_t st = __hip_atomic_load(&lstate[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (st == 0u) {
            const uint32_t old = atomicCAS(&lstate[slot], 0u, 1u);
            if (old == 0u) {
#pragma unroll
              for (int k = 0; k < KW; ++k) lkeys[k * S + slot] = key[k];
              __threadfence_block();
              __hip_atomic_store(&lstate[slot], 2u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
              found = slot;
              break;
            }
            st = old;
          }
          if (st == 1u) {
            if (++spins > 4096) break;
            continue;
          }
          __threadfence_block();
          bool same = true;
#pragma unroll
          for (int k = 0; k < KW; ++k) same = same && (((volatile uint64_t*)lkeys)[k * S + slot] == key[k]);
          if (same) found = slot;
          else slot = sub_base + ((slot - sub_base + 1) & (sub_slots - 1));
          ++p;
        }
      }
      if (found >= 0) {
#pragma unroll
        for (int a = 0; a < kMaxAggs; ++a)
          if (a < T.na) acc_atomic(T.acc_kind[a], &laccs[a * S + found], val[a]);
        todo = false;
        ++lds_hit;
      } else {
        ++lds_miss;
      }
    }
    // ---- global table ----
    if (todo && !saturated) {
      if (table_apply<KW>(T, key, val)) todo = false;
    }
    // ---- spill (table saturated or probe sequence exhausted) ----
    spill_row<KW>(T, spill, todo, key, val);
  }
  // flush the LDS cache: every occupied slot becomes one merge into the global table
  if (S > 0) {
    __syncthreads();
    for (int i = threadIdx.x; i < S; i += kBlock) {
      bool occ;
      uint64_t key[KW];
      if (KW == 1) {
        key[0] = lkeys[i];
        occ = key[0] != kEmptyKey;
      } else {
        occ = lstate[i] == 2u;
#pragma unroll
        for (int k = 0; k < KW; ++k) key[k] = lkeys[k * S + i];
      }
      uint64_t val[kMaxAggs];
#pragma unroll
      for (int a = 0; a < kMaxAggs; ++a) val[a] = (a < T.na) ? laccs[a * S + i] : 0;
      bool todo = occ;
      const bool sat = __hip_atomic_load(&T.ctrl[CTRL_SATURATED], RLX_AGENT) != 0u;
      if (todo && !sat) {
        if (table_apply<KW>(T, key, val)) todo = false;
      }
      spill_row<KW>(T, spill, todo, key, val);
    }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    passed += shfl_xor_u64(passed, m);
    lds_hit += __shfl_xor(lds_hit, m, 64);
    lds_miss += __shfl_xor(lds_miss, m, 64);
  }
  if (lane == 0) {
    if (passed) atomicAdd((unsigned long long*)&T.ctrl[CTRL_PASSED_LO], (unsigned long long)passed);
    if (lds_hit) atomicAdd(&T.ctrl[CTRL_LDS_HIT], lds_hit);
    if (lds_miss) atomicAdd(&T.ctrl[CTRL_LDS_MISS], lds_miss);
  }
  if (err) atomicOr(&T.ctrl[CTRL_ERROR], err);
}

// pre-evaluated rows -> table (spill replay, rehash, partial import).  The source is `rows` planes
// of capacity rows.capacity; rows [row_begin, row_begin + n_rows).
template <int KW>
__global__ __launch_bounds__(kBlock) void k_merge_rows(const DevRows rows, const int64_t row_begin,
                                                       const int64_t n_rows, const DevTable T,
                                                       const DevRows spill) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  const int64_t n_pad = (n_rows + 63) & ~63ll;  // whole waves stay in the loop for the ballots
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n_pad; i += stride) {
    const bool inb = i < n_rows;
    uint64_t key[KW];
    uint64_t val[kMaxAggs];
#pragma unroll
    for (int k = 0; k < KW; ++k) key[k] = inb ? rows.words[(uint64_t)k * rows.capacity + row_begin + i] : 0;
#pragma unroll
    for (int a = 0; a < kMaxAggs; ++a)
      val[a] = (inb && a < T.na) ? rows.words[(uint64_t)(KW + a) * rows.capacity + row_begin + i] : 0;
    bool todo = inb;
    if (todo && table_apply<KW>(T, key, val)) todo = false;
    spill_row<KW>(T, spill, todo, key, val);
  }
}

template <int KW>
DEV bool slot_occupied(const DevTable& T, uint64_t slot) {
  if (slot == T.mask + 1) return KW == 1 && T.ctrl[CTRL_SENTINEL] != 0u;
  if (KW == 1) return T.keys[slot] != kEmptyKey;
  return T.state[slot] == 2u;
}

template <int KW>
__global__ __launch_bounds__(kBlock) void k_rehash(const DevTable from, const DevTable to, const DevRows spill) {
  const int64_t n_slots = (int64_t)from.mask + 2;  // + the sentinel slot
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  const int64_t n_pad = (n_slots + 63) & ~63ll;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n_pad; i += stride) {
    const bool occ = i < n_slots && slot_occupied<KW>(from, (uint64_t)i);
    uint64_t key[KW];
    uint64_t val[kMaxAggs];
#pragma unroll
    for (int k = 0; k < KW; ++k) key[k] = occ ? from.keys[(uint64_t)k * from.stride + i] : 0;
    if (KW == 1 && occ && (uint64_t)i == from.mask + 1) key[0] = kEmptyKey;
#pragma unroll
    for (int a = 0; a < kMaxAggs; ++a) val[a] = (occ && a < from.na) ? from.accs[(uint64_t)a * from.stride + i] : 0;
    bool todo = occ;
    if (todo && table_apply<KW>(to, key, val)) todo = false;
    spill_row<KW>(to, spill, todo, key, val);
  }
}

__global__ __launch_bounds__(kBlock) void k_fill_u64(uint64_t* __restrict__ p, uint64_t v, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) p[i] = v;
}
__global__ __launch_bounds__(kBlock) void k_fill_u32(uint32_t* __restrict__ p, uint32_t v, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) p[i] = v;
}

// ---------------------------------------------------------------------------------------------
// K8 emit_groups
// ---------------------------------------------------------------------------------------------
template <int KW>
__global__ __launch_bounds__(kBlock) void k_table_mask(const DevTable T, uint64_t* __restrict__ mask_words,
                                                       uint32_t* __restrict__ tile_counts) {
  __shared__ uint32_t wave_cnt[kBlock / 64];
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int64_t n = (int64_t)T.mask + 2;
  const int64_t n_words = (n + 63) >> 6;
  const int64_t n_tiles = (n + kTileRows - 1) / kTileRows;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    uint32_t cnt = 0;
    for (int i = 0; i < 16; ++i) {
      const int64_t w = tile * 64 + wave * 16 + i;
      const int64_t slot = w * 64 + lane;
      const bool occ = slot < n && slot_occupied<KW>(T, (uint64_t)slot);
      const uint64_t word = __ballot(occ);
      if (lane == 0 && w < n_words) mask_words[w] = word;
      cnt += (uint32_t)__popcll(word);
    }
    if (lane == 0) wave_cnt[wave] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) tile_counts[tile] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    __syncthreads();
  }
}

// dense u64 plane -> typed output column (keys: narrow; aggregates: undo the accumulator image)
__global__ __launch_bounds__(kBlock) void k_finalize(const uint64_t* __restrict__ in, int64_t n,
                                                     uint8_t out_dtype, uint8_t val_xform,
                                                     void* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    uint64_t v = in[i];
    if (val_xform == VT_F64_ORD_MIN || val_xform == VT_F64_ORD_MAX) {
      v = f64_bits(f64_from_ordered(v));
    } else if (val_xform == VT_F32_ORD_MIN || val_xform == VT_F32_ORD_MAX) {
      v = f32_bits((float)f64_from_ordered(v));
    }
    store_typed(out_dtype, out, i, v);
  }
}

// ---------------------------------------------------------------------------------------------
// multi-GPU partial export
// ---------------------------------------------------------------------------------------------
template <int KW>
DEV uint64_t slot_key_hash(const DevTable& T, uint64_t slot, uint64_t (&key)[KW]) {
#pragma unroll
  for (int k = 0; k < KW; ++k) key[k] = T.keys[(uint64_t)k * T.stride + slot];
  if (KW == 1 && slot == T.mask + 1) key[0] = kEmptyKey;
  return hash_keys<KW>(key);
}

template <int KW>
__global__ __launch_bounds__(kBlock) void k_partial_count(const DevTable T, int world, uint64_t* counts) {
  const int64_t n = (int64_t)T.mask + 2;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    if (slot_occupied<KW>(T, (uint64_t)i)) {
      uint64_t key[KW];
      const uint64_t h = slot_key_hash<KW>(T, (uint64_t)i, key);
      atomicAdd((unsigned long long*)&counts[(h >> 7) % (uint64_t)world], 1ull);
    }
  }
}

template <int KW>
__global__ __launch_bounds__(kBlock) void k_partial_scatter(const DevTable T, int world,
                                                            const uint64_t* __restrict__ bucket_base,
                                                            const uint64_t* __restrict__ bucket_count,
                                                            uint64_t* cursors, uint64_t* __restrict__ dst) {
  const int64_t n = (int64_t)T.mask + 2;
  const int nw = KW + T.na;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    if (slot_occupied<KW>(T, (uint64_t)i)) {
      uint64_t key[KW];
      const uint64_t h = slot_key_hash<KW>(T, (uint64_t)i, key);
      const uint64_t r = (h >> 7) % (uint64_t)world;
      const uint64_t g = atomicAdd((unsigned long long*)&cursors[r], 1ull);
      uint64_t* b = dst + (uint64_t)nw * bucket_base[r];
      const uint64_t cnt = bucket_count[r];
#pragma unroll
      for (int k = 0; k < KW; ++k) b[(uint64_t)k * cnt + g] = key[k];
      for (int a = 0; a < T.na; ++a) b[(uint64_t)(KW + a) * cnt + g] = T.accs[(uint64_t)a * T.stride + i];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// synthetic columns -- same definition as orc_synth_fill (oracle/dfx_oracle.c)
// ---------------------------------------------------------------------------------------------
DEV uint64_t synth_u64(uint64_t seed, int column_id, int64_t row) {
  const uint64_t s = seed ^ ((uint64_t)(uint32_t)column_id * 0xA0761D6478BD642Full);
  return mix64(s + ((uint64_t)row + 1ull) * 0x9E3779B97F4A7C15ull);
}

__global__ __launch_bounds__(kBlock) void k_synth(int kind, int column_id, double p0, double p1, uint64_t seed,
                                                  int64_t row_begin, int64_t n, void* __restrict__ out) {
  int zipf_bits = 0;
  const uint64_t G = (uint64_t)(int64_t)p0;
  if (kind == 3) {
    while ((1ull << zipf_bits) < G && zipf_bits < 62) ++zipf_bits;
  }
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const uint64_t r = synth_u64(seed, column_id, row_begin + i);
    if (kind == 0) {
      const double u = (double)(r >> 11) * 0x1.0p-53;
      const double t = p1 * u;
      ((double*)out)[i] = p0 + t;
    } else if (kind == 1) {
      ((double*)out)[i] = (double)(r >> 44) * 0x1.0p-10;
    } else if (kind == 2) {
      ((int64_t*)out)[i] = (int64_t)__umul64hi(r, G);
    } else {
      const uint64_t b = __umul64hi(r, (uint64_t)zipf_bits + 1ull);
      const uint64_t r2 = mix64(r ^ 0xD6E8FEB86659FD93ull);
      uint64_t k = (b == 0) ? 0 : ((1ull << (b - 1)) + __umul64hi(r2, 1ull << (b - 1)));
      if (k >= G) k = G - 1;
      ((int64_t*)out)[i] = (int64_t)k;
    }
  }
}

// =============================================================================================
// host side: launch helpers + profiler
// =============================================================================================
static const char* kKernelNames[KID_COUNT_] = {
    "predicate_mask", "compact", "project", "reduce_all", "hash_agg", "merge_rows", "rehash",
    "emit_mask", "finalize", "scan", "synth", "fill", "gather_utf8", "partial", "partition"};
const char* kernel_name(int kid) { return (kid >= 0 && kid < KID_COUNT_) ? kKernelNames[kid] : "?"; }

namespace {
struct ProfEntry {
  int64_t launches = 0;
  double total_ms = 0.0;
  double algo_bytes = 0.0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};
std::mutex g_prof_mu;
bool g_prof_on = false;
ProfEntry g_prof[KID_COUNT_];
std::vector<hipEvent_t> g_event_pool;

hipEvent_t get_event() {
  if (!g_event_pool.empty()) {
    hipEvent_t e = g_event_pool.back();
    g_event_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

void drain(ProfEntry& p) {
  for (auto& pr : p.pending) {
    float ms = 0.f;
    if (hipEventSynchronize(pr.second) == hipSuccess && hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess)
      p.total_ms += (double)ms;
    g_event_pool.push_back(pr.first);
    g_event_pool.push_back(pr.second);
  }
  p.pending.clear();
}

struct Scope {  // brackets one launch with events on the launch stream when profiling is on
  int kid;
  hipStream_t s;
  hipEvent_t a = nullptr, b = nullptr;
  Scope(int kid_, hipStream_t s_, double bytes) : kid(kid_), s(s_) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_prof_on) return;
    a = get_event();
    b = get_event();
    g_prof[kid].launches += 1;
    g_prof[kid].algo_bytes += bytes;
    if (a) hipEventRecord(a, s);
  }
  ~Scope() {
    if (!a || !b) return;
    hipEventRecord(b, s);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof[kid].pending.emplace_back(a, b);
  }
};

int g_cu_count = 0;
}  // namespace

int device_cu_count() {
  if (g_cu_count == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      g_cu_count = prop.multiProcessorCount;
    if (g_cu_count <= 0) g_cu_count = 256;
  }
  return g_cu_count;
}

void profile_enable(bool on) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_on = on;
}
void profile_reset() {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& p : g_prof) {
    drain(p);
    p.launches = 0;
    p.total_ms = 0;
    p.algo_bytes = 0;
  }
}
int profile_count() { return KID_COUNT_; }
bool profile_get(int index, const char** name, int64_t* launches, double* total_ms, double* algo_bytes) {
  if (index < 0 || index >= KID_COUNT_) return false;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  drain(g_prof[index]);
  *name = kKernelNames[index];
  *launches = g_prof[index].launches;
  *total_ms = g_prof[index].total_ms;
  *algo_bytes = g_prof[index].algo_bytes;
  return true;
}

// grid for a streaming kernel over `units` block-sized units: enough workgroups to fill 256 CUs
// several times over (>> 256 WGs; blocks land round-robin on the 8 XCDs), capped so the
// grid-stride loop amortises launch and tail effects.
static int stream_grid(int64_t units, int per_cu) {
  int64_t cap = (int64_t)device_cu_count() * per_cu;
  if (units < 1) units = 1;
  return (int)(units < cap ? units : cap);
}

hipError_t launch_predicate_mask(const DevProgram& P, const DevColumns& C, uint8_t pred, int64_t n,
                                 uint64_t* mask_words, uint32_t* tile_counts, uint32_t* ctrl,
                                 double algo_bytes, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  Scope sc(KID_PREDICATE_MASK, s, algo_bytes);
  const int64_t tiles = (n + kTileRows - 1) / kTileRows;
  const int grid = stream_grid(tiles, 8);
  if (P.has_nulls)
    hipLaunchKernelGGL(k_predicate_mask<true>, dim3(grid), dim3(kBlock), 0, s, P, C, pred, n, mask_words, tile_counts, ctrl);
  else
    hipLaunchKernelGGL(k_predicate_mask<false>, dim3(grid), dim3(kBlock), 0, s, P, C, pred, n, mask_words, tile_counts, ctrl);
  return hipGetLastError();
}

template <typename TIN, typename TOUT>
static hipError_t scan_impl(const TIN* in, TOUT* out, int64_t n, uint64_t* tmp, hipStream_t s) {
  Scope sc(KID_SCAN, s, 0);
  const int64_t nb = n > 0 ? (n + kScanChunk - 1) / kScanChunk : 1;
  hipLaunchKernelGGL((k_scan_local<TIN>), dim3((unsigned)nb), dim3(kBlock), 0, s, in, n, tmp);
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(kBlock), 0, s, tmp, nb);
  hipLaunchKernelGGL((k_scan_apply<TIN, TOUT>), dim3((unsigned)nb), dim3(kBlock), 0, s, in, n, tmp, nb, out);
  return hipGetLastError();
}
hipError_t launch_scan_u32(const uint32_t* in, uint64_t* out, int64_t n, uint64_t* tmp, hipStream_t s) {
  return scan_impl<uint32_t, uint64_t>(in, out, n, tmp, s);
}
hipError_t launch_scan_i32(const int32_t* in, int32_t* out, int64_t n, uint64_t* tmp, hipStream_t s) {
  return scan_impl<int32_t, int32_t>(in, out, n, tmp, s);
}

hipError_t launch_compact(const void* in, int width, const uint64_t* mask_words, const uint64_t* tile_offsets,
                          int64_t n, void* out, double algo_bytes, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  Scope sc(KID_COMPACT, s, algo_bytes);
  const int64_t tiles = (n + kTileRows - 1) / kTileRows;
  const int grid = stream_grid(tiles, 8);
  switch (width) {
    case 8: hipLaunchKernelGGL(k_compact<uint64_t>, dim3(grid), dim3(kBlock), 0, s, (const uint64_t*)in, mask_words, tile_offsets, n, (uint64_t*)out); break;
    case 4: hipLaunchKernelGGL(k_compact<uint32_t>, dim3(grid), dim3(kBlock), 0, s, (const uint32_t*)in, mask_words, tile_offsets, n, (uint32_t*)out); break;
    case 2: hipLaunchKernelGGL(k_compact<uint16_t>, dim3(grid), dim3(kBlock), 0, s, (const uint16_t*)in, mask_words, tile_offsets, n, (uint16_t*)out); break;
    case 1: hipLaunchKernelGGL(k_compact<uint8_t>, dim3(grid), dim3(kBlock), 0, s, (const uint8_t*)in, mask_words, tile_offsets, n, (uint8_t*)out); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_utf8_lengths(const int32_t* offsets, int64_t n, int32_t* lengths, int32_t* starts, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  Scope sc(KID_GATHER_UTF8, s, 0);
  const int grid = stream_grid((n + kBlock - 1) / kBlock, 8);
  hipLaunchKernelGGL(k_utf8_lengths, dim3(grid), dim3(kBlock), 0, s, offsets, n, lengths, starts);
  return hipGetLastError();
}
hipError_t launch_utf8_gather(const uint8_t* data, const int32_t* src_starts, const int32_t* dst_offsets,
                              int64_t m, uint8_t* out, hipStream_t s) {
  if (m <= 0) return hipSuccess;
  Scope sc(KID_GATHER_UTF8, s, 0);
  const int grid = stream_grid((m * 16 + kBlock - 1) / kBlock, 8);
  hipLaunchKernelGGL(k_utf8_gather, dim3(grid), dim3(kBlock), 0, s, data, src_starts, dst_offsets, m, out);
  return hipGetLastError();
}

hipError_t launch_project(const DevProgram& P, const DevColumns& C, const DevProjectPlan& plan, int64_t n,
                          uint32_t* ctrl, double algo_bytes, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  Scope sc(KID_PROJECT, s, algo_bytes);
  const int grid = stream_grid((n + kBlock - 1) / kBlock, 8);
  if (P.has_nulls)
    hipLaunchKernelGGL(k_project<true>, dim3(grid), dim3(kBlock), 0, s, P, C, plan, n, ctrl);
  else
    hipLaunchKernelGGL(k_project<false>, dim3(grid), dim3(kBlock), 0, s, P, C, plan, n, ctrl);
  return hipGetLastError();
}

hipError_t launch_reduce(const DevProgram& P, const DevColumns& C, const DevAggPlan& plan, const DevTable& T,
                         int64_t n, uint64_t* partial, uint32_t* ctrl, double algo_bytes, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  Scope sc(KID_REDUCE, s, algo_bytes);
  const int grid = stream_grid((n + kBlock - 1) / kBlock, 8);
  if (P.has_nulls)
    hipLaunchKernelGGL(k_reduce<true>, dim3(grid), dim3(kBlock), 0, s, P, C, plan, T, n, partial, ctrl);
  else
    hipLaunchKernelGGL(k_reduce<false>, dim3(grid), dim3(kBlock), 0, s, P, C, plan, T, n, partial, ctrl);
  return hipGetLastError();
}

hipError_t launch_reduce_fold(const DevTable& T, const uint8_t* arg_dtype, const uint8_t* func,
                              uint64_t* partial, uint64_t* state, hipStream_t s) {
  hipLaunchKernelGGL(k_reduce_fold, dim3(1), dim3(64), 0, s, T, arg_dtype, func, partial, state);
  return hipGetLastError();
}

template <int KW>
static hipError_t hash_agg_impl(const DevProgram& P, const DevColumns& C, const DevAggPlan& plan,
                                const DevTable& T, const DevRows& spill, int64_t n, hipStream_t s) {
  const int64_t n_blocks = (n + kBlock - 1) / kBlock;
  const size_t lds_bytes = plan.lds_slots > 0
                               ? (size_t)plan.lds_slots * ((size_t)(KW + T.na) * 8 + (KW > 1 ? 4 : 0))
                               : 0;
  // LDS-heavy blocks: fewer, longer-lived workgroups amortise the cache init + flush
  const int per_cu = lds_bytes > 0 ? (lds_bytes > 40 * 1024 ? 2 : 4) : 8;
  const int grid = stream_grid(n_blocks, per_cu);
  if (P.has_nulls)
    hipLaunchKernelGGL((k_hash_agg<KW, true>), dim3(grid), dim3(kBlock), lds_bytes, s, P, C, plan, T, spill, n);
  else
    hipLaunchKernelGGL((k_hash_agg<KW, false>), dim3(grid), dim3(kBlock), lds_bytes, s, P, C, plan, T, spill, n);
  return hipGetLastError();
}

hipError_t launch_hash_agg(const DevProgram& P, const DevColumns& C, const DevAggPlan& plan, const DevTable& T,
                           const DevRows& spill, int64_t n, double algo_bytes, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  Scope sc(KID_HASH_AGG, s, algo_bytes);
  switch (T.kw) {
    case 1: return hash_agg_impl<1>(P, C, plan, T, spill, n, s);
    case 2: return hash_agg_impl<2>(P, C, plan, T, spill, n, s);
    case 3: return hash_agg_impl<3>(P, C, plan, T, spill, n, s);
    case 4: return hash_agg_impl<4>(P, C, plan, T, spill, n, s);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_merge_rows(const DevRows& rows, int64_t row_begin, int64_t n_rows, const DevTable& T,
                             const DevRows& spill, hipStream_t s) {
  if (n_rows <= 0) return hipSuccess;
  Scope sc(KID_MERGE_ROWS, s, 0);
  const int grid = stream_grid((n_rows + kBlock - 1) / kBlock, 8);
  switch (T.kw) {
    case 1: hipLaunchKernelGGL(k_merge_rows<1>, dim3(grid), dim3(kBlock), 0, s, rows, row_begin, n_rows, T, spill); break;
    case 2: hipLaunchKernelGGL(k_merge_rows<2>, dim3(grid), dim3(kBlock), 0, s, rows, row_begin, n_rows, T, spill); break;
    case 3: hipLaunchKernelGGL(k_merge_rows<3>, dim3(grid), dim3(kBlock), 0, s, rows, row_begin, n_rows, T, spill); break;
    case 4: hipLaunchKernelGGL(k_merge_rows<4>, dim3(grid), dim3(kBlock), 0, s, rows, row_begin, n_rows, T, spill); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_merge_bucket(const uint64_t* bucket, uint64_t count, const DevTable& T, const DevRows& spill,
                               hipStream_t s) {
  DevRows rows;
  rows.words = const_cast<uint64_t*>(bucket);
  rows.capacity = count;
  return launch_merge_rows(rows, 0, (int64_t)count, T, spill, s);
}

hipError_t launch_rehash(const DevTable& from, const DevTable& to, const DevRows& spill, hipStream_t s) {
  Scope sc(KID_REHASH, s, 0);
  const int64_t n = (int64_t)from.mask + 2;
  const int grid = stream_grid((n + kBlock - 1) / kBlock, 8);
  switch (from.kw) {
    case 1: hipLaunchKernelGGL(k_rehash<1>, dim3(grid), dim3(kBlock), 0, s, from, to, spill); break;
    case 2: hipLaunchKernelGGL(k_rehash<2>, dim3(grid), dim3(kBlock), 0, s, from, to, spill); break;
    case 3: hipLaunchKernelGGL(k_rehash<3>, dim3(grid), dim3(kBlock), 0, s, from, to, spill); break;
    case 4: hipLaunchKernelGGL(k_rehash<4>, dim3(grid), dim3(kBlock), 0, s, from, to, spill); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_fill_u64(uint64_t* p, uint64_t v, int64_t n, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  Scope sc(KID_FILL, s, 0);
  hipLaunchKernelGGL(k_fill_u64, dim3(stream_grid((n + kBlock - 1) / kBlock, 8)), dim3(kBlock), 0, s, p, v, n);
  return hipGetLastError();
}
hipError_t launch_fill_u32(uint32_t* p, uint32_t v, int64_t n, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  Scope sc(KID_FILL, s, 0);
  hipLaunchKernelGGL(k_fill_u32, dim3(stream_grid((n + kBlock - 1) / kBlock, 8)), dim3(kBlock), 0, s, p, v, n);
  return hipGetLastError();
}

hipError_t launch_table_mask(const DevTable& T, uint64_t* mask_words, uint32_t* tile_counts, hipStream_t s) {
  Scope sc(KID_EMIT_MASK, s, 0);
  const int64_t n = (int64_t)T.mask + 2;
  const int64_t tiles = (n + kTileRows - 1) / kTileRows;
  const int grid = stream_grid(tiles, 8);
  switch (T.kw) {
    case 1: hipLaunchKernelGGL(k_table_mask<1>, dim3(grid), dim3(kBlock), 0, s, T, mask_words, tile_counts); break;
    case 2: hipLaunchKernelGGL(k_table_mask<2>, dim3(grid), dim3(kBlock), 0, s, T, mask_words, tile_counts); break;
    case 3: hipLaunchKernelGGL(k_table_mask<3>, dim3(grid), dim3(kBlock), 0, s, T, mask_words, tile_counts); break;
    case 4: hipLaunchKernelGGL(k_table_mask<4>, dim3(grid), dim3(kBlock), 0, s, T, mask_words, tile_counts); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_finalize(const uint64_t* in, int64_t n, uint8_t out_dtype, uint8_t val_xform, void* out,
                           hipStream_t s) {
  if (n <= 0) return hipSuccess;
  Scope sc(KID_FINALIZE, s, 0);
  hipLaunchKernelGGL(k_finalize, dim3(stream_grid((n + kBlock - 1) / kBlock, 8)), dim3(kBlock), 0, s, in, n, out_dtype, val_xform, out);
  return hipGetLastError();
}

hipError_t launch_partial_count(const DevTable& T, int world, uint64_t* counts, hipStream_t s) {
  Scope sc(KID_PARTIAL, s, 0);
  const int64_t n = (int64_t)T.mask + 2;
  const int grid = stream_grid((n + kBlock - 1) / kBlock, 8);
  switch (T.kw) {
    case 1: hipLaunchKernelGGL(k_partial_count<1>, dim3(grid), dim3(kBlock), 0, s, T, world, counts); break;
    case 2: hipLaunchKernelGGL(k_partial_count<2>, dim3(grid), dim3(kBlock), 0, s, T, world, counts); break;
    case 3: hipLaunchKernelGGL(k_partial_count<3>, dim3(grid), dim3(kBlock), 0, s, T, world, counts); break;
    case 4: hipLaunchKernelGGL(k_partial_count<4>, dim3(grid), dim3(kBlock), 0, s, T, world, counts); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_partial_scatter(const DevTable& T, int world, const uint64_t* bucket_base,
                                  const uint64_t* bucket_count, uint64_t* cursors, uint64_t* dst, hipStream_t s) {
  Scope sc(KID_PARTIAL, s, 0);
  const int64_t n = (int64_t)T.mask + 2;
  const int grid = stream_grid((n + kBlock - 1) / kBlock, 8);
  switch (T.kw) {
    case 1: hipLaunchKernelGGL(k_partial_scatter<1>, dim3(grid), dim3(kBlock), 0, s, T, world, bucket_base, bucket_count, cursors, dst); break;
    case 2: hipLaunchKernelGGL(k_partial_scatter<2>, dim3(grid), dim3(kBlock), 0, s, T, world, bucket_base, bucket_count, cursors, dst); break;
    case 3: hipLaunchKernelGGL(k_partial_scatter<3>, dim3(grid), dim3(kBlock), 0, s, T, world, bucket_base, bucket_count, cursors, dst); break;
    case 4: hipLaunchKernelGGL(k_partial_scatter<4>, dim3(grid), dim3(kBlock), 0, s, T, world, bucket_base, bucket_count, cursors, dst); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_synth(int kind, int column_id, double p0, double p1, uint64_t seed, int64_t row_begin, int64_t n,
                        void* out, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  Scope sc(KID_SYNTH, s, 0);
  hipLaunchKernelGGL(k_synth, dim3(stream_grid((n + kBlock - 1) / kBlock, 16)), dim3(kBlock), 0, s, kind, column_id, p0, p1, seed, row_begin, n, out);
  return hipGetLastError();
}

}  // namespace dfx
