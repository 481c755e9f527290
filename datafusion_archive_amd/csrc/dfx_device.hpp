// dfx_device.hpp -- structs shared by host code and gfx950 kernels.
//
// The RecordBatch expression evaluator of the reference (src/execution/expression.rs) composes
// Rust closures, one full materialised array per node.  Here an expression forest (predicate +
// group keys + aggregate arguments, or projection outputs) is compiled ONCE, at operator
// creation, into a tiny SSA register program that one fused kernel interprets per row with all
// intermediates in VGPRs (register files are ext_vector_type values indexed by wave-uniform
// indices -> s_set_gpr_idx, no scratch).  Literals never become arrays: they live in the kernarg
// segment and are read with scalar loads.
#pragma once
#include <stdint.h>

namespace dfx {

constexpr int kMaxRegs = 16;  // computed values per fused program
constexpr int kMaxCols = 8;   // distinct input columns per fused program
constexpr int kMaxImm = 16;   // distinct literals per fused program
constexpr int kMaxKeys = 8;   // GROUP BY key words (the table kernels are built for 1, 2, 3, 4 and 8: five to seven keys run as eight, padded with constant words)
constexpr int kMaxAggs = 8;   // aggregates per AggregateRelation
constexpr int kMaxOut = 8;    // projection outputs per launch

// dtype codes == dfx_dtype (include/dfx.h)
enum : uint8_t {
  T_NONE = 0, T_BOOL = 1, T_I8 = 2, T_I16 = 3, T_I32 = 4, T_I64 = 5, T_U8 = 6, T_U16 = 7,
  T_U32 = 8, T_U64 = 9, T_F32 = 10, T_F64 = 11, T_UTF8 = 12
};

// operand = (kind << 6) | index
enum : uint8_t { OPK_REG = 0, OPK_COL = 1, OPK_IMM = 2, OPK_NONE = 3 };
constexpr uint8_t kNoOperand = 0xFF;
inline constexpr uint8_t make_operand(int kind, int idx) { return (uint8_t)((kind << 6) | (idx & 63)); }

enum : uint8_t {
  DOP_EQ = 0, DOP_NE = 1, DOP_LT = 2, DOP_LE = 3, DOP_GT = 4, DOP_GE = 5,  // == dfx_operator 0..5
  DOP_AND = 6, DOP_OR = 7, DOP_ADD = 8, DOP_SUB = 9, DOP_MUL = 10, DOP_DIV = 11, DOP_CAST = 12
};

struct DevIns {
  uint8_t op;  // DOP_*
  uint8_t t;   // operand dtype (CAST: source dtype)
  uint8_t a;   // operand
  uint8_t b;   // operand (CAST: target dtype, raw)
};

struct DevProgram {
  int32_t n_ins;
  int32_t n_cols;
  int32_t n_imm;
  int32_t has_nulls;  // any referenced column carries a validity bitmap in this batch
  int32_t wide8;      // every referenced column is 8 bytes wide (Int64 / UInt64 / Float64) and none has nulls in this batch:
                      // the generic policies load them branch-free, all loads of a trip in flight together
  DevIns ins[kMaxRegs];
  uint64_t imm[kMaxImm];
  uint8_t col_dtype[kMaxCols];
};

struct DevColumn {
  const void* values;       // element 0 of the (offset-adjusted) values buffer; Boolean: bitmap base
  const uint8_t* validity;  // bitmap base or nullptr
  int64_t bit_offset;       // arrow `offset`: applies to validity bits and Boolean value bits
};

struct DevColumns {
  DevColumn c[kMaxCols];
};

// ---- shape-specialised fast plan ------------------------------------------------------------
// Most scans have a fixed shape: a conjunction of `column <op> literal` comparisons, plain-column
// GROUP BY keys, and aggregate arguments that are a column or a short product of
// (column | literal +- column) factors (TPC-H Q1: price * (1 - disc) * (1 + tax)).  For these the
// host derives a DevFastPlan from the SSA program; the kernels then evaluate it with straight-line
// code (no bytecode loop, almost no scalar work).  Anything else, and any batch with nulls, runs
// the generic interpreter.  Both produce identical results (tests compare them).
enum : uint8_t {
  FF_NONE = 0, FF_COL = 1, FF_IMM_MINUS_COL = 2, FF_COL_PLUS_IMM = 3, FF_COL_MINUS_IMM = 4, FF_COL_TIMES_IMM = 5,
  FF_RT = 15  // only in compile-time signatures (dfx_sigs.hpp): "column <op> literal, the op is the plan's" -- never in a DevFastPlan
};
struct DevFastTerm {
  uint8_t col;    // column slot
  uint8_t dtype;  // column dtype (compare class)
  uint8_t m;      // three-way mask: 1 less, 2 equal, 4 greater
  uint8_t inv;    // 1: NotEq
};
struct DevFastFactor {
  uint8_t kind;  // FF_*
  uint8_t col;
};
struct DevFastArg {
  uint8_t nf;  // 1..3 factors multiplied left to right; nf == 1 && FF_COL: the plain column
  uint8_t pad;
  DevFastFactor f[3];
};
// ---- scan plan: the same shape family as DATA, not as control flow ------------------------------
// The run-time decoded shapes above are bound by the CU's one scalar unit: operator masks chosen with s_cselect, dtype
// branches, plan words read inside the loop (126 scalar instructions per 64-row group against 43 for
// a compile-time signature).  A scan plan removes the decisions instead of compiling them in:
//   * every `column <op> literal` term is an unsigned RANGE test on an order-preserving 64-bit image of the value:
//       image(x) = x ^ (sign(x) & a) ^ b        f64: a = 0x7FF..F, b = 0x80..0   i64: a = 0, b = 0x80..0   u64: a = b = 0
//       pass     = ((image(x) - lo) <=u span) != inv
//     Eq is a one-value range (two for +-0.0: their images are neighbours), NotEq its complement, the ordered operators
//     half-lines that end at the image of -inf / +inf, so NaN fails every ordered term and Eq and passes NotEq, as IEEE
//     says; an impossible term (x < -inf, a NaN literal) is the complement of the full range.  One code path for six
//     operators and six column types;
//   * 4-byte columns (Int32 / UInt32 / Float32) are read as the ALIGNED 8 bytes that hold the element -- the same
//     unconditional global_load_dwordx2 as an 8-byte column, so all loads of a trip are in flight together -- and widened
//     afterwards (sign / zero extension, f32 -> f64 is exact) by lane parity;
//   * validity bitmaps are read one BYTE per lane (the 64 lanes of a row group share 8 or 9 bytes: one cache line), the bit
//     is (byte >> ((bit_offset + lane) & 7)) & 1; a column without a bitmap points at a block of 0xFF bytes.  A null value
//     gives the term arrow 0.12's answer for None (it sorts below every value): DevPlanTerm::if_null;
//   * the plan words live in VECTOR registers (dfx_kernels_inl.hpp, PlanPolicy): no scalar-register pressure, no selects.
// Column slots are the plan's own: PlanPolicy1 kernels (one key, one routed value) find the key in slot 0 and the
// aggregate's argument in slot 1, further predicate columns behind them.
constexpr int kPlanCols = 4;
constexpr int kPlanTerms = 4;
enum : uint32_t { PX_NONE = 0, PX_SEXT32 = 1, PX_ZEXT32 = 2, PX_F32 = 3 };
// column word (DevScanPlan::col_meta, also DevColumn::bit_offset of a plan-bound column):
//   bits 0..3 log2(value bytes) (2 / 3), bit 4 element index of row 0 inside its aligned pair (4-byte values),
//   bits 8..9 PX_*, bits 16..18 bit_offset & 7 of the validity bitmap, bit 20 the column HAS a bitmap (without one every lane
//   reads byte 0 of the block of 0xFF bytes, whatever its row)
inline constexpr uint32_t plan_col_meta(uint32_t shift, uint32_t delta, uint32_t ext, uint32_t vbit0, bool bitmap) {
  return shift | (delta << 4) | (ext << 8) | (vbit0 << 16) | (bitmap ? 1u << 20 : 0u);
}
struct DevPlanTerm {
  uint32_t col;      // plan column slot
  uint32_t inv;      // 1: complement (NotEq, impossible terms)
  uint32_t if_null;  // the term's value for a null column value
  uint32_t pad;
  uint64_t a, b;     // image transform
  uint64_t lo, span; // range
};
struct DevScanPlan {
  int32_t valid;        // 0: shape not covered (then nothing below is meaningful)
  int32_t gen;          // what this batch needs of the kernels: bit 0 a 4-byte column (widening loads), bit 1 a validity bitmap,
                        // bit 2 (instead of bit 0; one-key kernels) the key is the ONLY 4-byte column: a real 4-byte load
  int32_t n_cols;       // plan column slots in use
  int32_t np;           // terms
  int32_t count_valid;  // 1: COUNT(x) looks at x's validity (no Filter below: fn filter's output is all-valid, filter.rs:83-92)
  uint32_t col_meta[kPlanCols];
  uint8_t keyslot[kMaxKeys];
  uint8_t argslot[kMaxAggs];
  DevPlanTerm term[kPlanTerms];
};

struct DevFastPlan {
  int32_t valid;  // 0: shape not covered
  int32_t np;     // predicate terms (0: no predicate)
  DevFastTerm term[4];
  uint64_t term_imm[4];
  uint8_t keycol[kMaxKeys];
  DevFastArg arg[kMaxAggs];
  uint64_t arg_imm[kMaxAggs][3];
  int32_t synth;     // bit i: term i was ADDED by build_fast (the open side of a one-sided range): true for every value the query's
                     // own term passes, so a scan plan leaves it out (and a null value must not be judged by it)
  int32_t plan_mode; // AggOptions::plan as the launchers see it (scan.plan), bits 0..1: 0 never bind a scan plan, 1 after the
                     // signatures, 2 before them; bit 2: the host RELIES on the plan (a fused predicate over a batch with nulls, which
                     // only a plan evaluates by the reference's rules): a launcher that cannot bind one fails instead of falling back
  DevScanPlan scan;  // filled per launch by bind_scan_plan() (dfx_expr.cpp) when a PlanPolicy kernel runs
};

// ---- aggregation ----------------------------------------------------------------------------
// Every accumulator is one 64-bit word updated with ONE hardware atomic per row.
enum : uint8_t {
  ACC_ADD_F64 = 0,  // SUM(f64): global_atomic_add_f64 / ds_add_f64
  ACC_ADD_F32 = 1,  // SUM(f32): f32 add on the low dword
  ACC_ADD_U64 = 2,  // SUM(int*) wrapping, COUNT
  ACC_MIN_S64 = 3,
  ACC_MAX_S64 = 4,
  ACC_MIN_U64 = 5,  // also MIN(f32/f64) on the order-preserving integer image
  ACC_MAX_U64 = 6   // also MAX(f32/f64)
};
// how the evaluated argument becomes the accumulator operand
enum : uint8_t {
  VT_RAW = 0,           // canonical 64-bit value as is
  VT_F64_ORD_MIN = 1,   // f64 -> sortable u64, NaN canonicalised ABOVE +inf (f64::min ignores NaN)
  VT_F64_ORD_MAX = 2,   // f64 -> sortable u64, NaN canonicalised BELOW -inf (f64::max ignores NaN)
  VT_F32_ORD_MIN = 3,   // f32 widened to f64 (exact), then as above
  VT_F32_ORD_MAX = 4,
  VT_COUNT_VALID = 5    // 1 if the argument is valid else 0
};

constexpr int kSynthNullStream = 0x4E554C4C;  // == DFX_SYNTH_NULL_STREAM (include/dfx.h): the draw that decides a synthetic row's validity
constexpr uint64_t kEmptyKey = 0x8000000000000000ull;  // single-word key claim sentinel
// PTF_NARROW: 32-bit hash images stand for keys; two images are reserved (keys that hash to them are not "narrow")
constexpr uint32_t kTagEmpty = 0xFFFFFFFFu;    // empty slot of the LDS tag plane / padding row of a region
constexpr uint32_t kTagForeign = 0xFFFFFFFEu;  // slot that holds a key without an image (>= 2^32): occupied, never matches

// ungrouped aggregates: every workgroup of K5 adds its result to one of kReduceSlots copies of the batch
// partial (slot = workgroup index mod kReduceSlots) -- thousands of agent-scope atomics on ONE address
// serialise at ~11 ns each; the fold kernel combines the copies
constexpr int kReduceSlots = 64;
constexpr int kReduceSlotWords = 4 * kMaxAggs;  // per aggregate: accumulator, valid count, first-valid tag, (a == 0: rows passed)

// control block words (uint32) of a group table / reduction
enum : int {
  CTRL_OCCUPIED = 0,    // groups in the table
  CTRL_ERROR = 1,       // bit0 DivideByZero, bit1 signed-division overflow
  CTRL_SPILL_LO = 2,    // spill cursor (64-bit, two words)
  CTRL_SPILL_HI = 3,
  CTRL_SENTINEL = 4,    // 1: the key equal to kEmptyKey has been seen (its slot is index cap)
  CTRL_SATURATED = 5,   // table past its load limit: blocks spill instead of probing
  CTRL_LDS_HIT = 6,     // rows absorbed by the LDS front cache (sampled statistic)
  CTRL_LDS_MISS = 7,
  CTRL_PASSED_LO = 8,   // rows that passed the predicate (64-bit)
  CTRL_PASSED_HI = 9,
  CTRL_WIDE_KEYS = 11,  // nonzero: a key >= 2^32 (or with a reserved hash image) has been seen -- no 12-byte rows (PTF_NARROW)
  CTRL_MAX_FILL = 10,   // partitioned strategy: largest region fill any producer has seen since the last pass 2 (which resets it)
  CTRL_WORDS = 16
};

constexpr int kStatStripes = 64;
enum : int { STAT_PASSED = 0, STAT_LDS_HIT = 1, STAT_LDS_MISS = 2, STAT_WORDS = 8 };

struct DevTable {
  uint64_t* keys;      // kw planes of `stride` words; plane 0 doubles as the claim word when kw == 1
  uint64_t* accs;      // na planes of `stride` words, pre-filled with the identity
  uint32_t* state;     // kw > 1: claim state per slot (0 empty, 1 busy, 2 ready); else nullptr
  uint32_t* ctrl;      // CTRL_WORDS words
  uint64_t* stats;     // [kStatStripes][STAT_WORDS] statistics counters (may be null), striped by workgroup:
                       // thousands of atomics on ONE address serialise at ~11 ns each on MI355X
  uint64_t stride;     // cap + 64 (slot `cap` is reserved for the kEmptyKey group)
  uint64_t mask;       // cap - 1
  int32_t shift;       // 64 - log2(cap): slot = hash >> shift
  int32_t kw;
  int32_t na;
  int32_t max_probe;
  uint32_t block_mask; // probing is confined to aligned blocks of block_mask + 1 slots (a block is
                       // also the LDS sub-table of one partition in the partitioned strategy)
  uint32_t pad0;
  uint64_t load_limit; // occupied above this => saturated
  uint8_t acc_kind[kMaxAggs];
  uint8_t val_xform[kMaxAggs];
  uint64_t acc_init[kMaxAggs];
};

// spill / partial rows: word-major planes [kw + na][capacity]
struct DevRows {
  uint64_t* words;
  uint64_t capacity;
};

struct DevAggPlan {
  uint8_t pred;                // operand or kNoOperand
  uint8_t key[kMaxKeys];       // operands
  uint8_t key_dtype[kMaxKeys];
  uint8_t arg[kMaxAggs];       // operands
  uint8_t arg_dtype[kMaxAggs];
  int32_t lds_slots;           // 0: no LDS front cache; else power of two
  int32_t lds_copies;          // power of two: lane-replicated sub-tables (few-group inputs)
};

// ---- partitioned strategy (high-cardinality GROUP BY) -------------------------------------------
// Global atomics top out at ~24 G/s on MI355X whatever the table size; LDS atomics run at > 1 T/s.
// So for many groups the passing rows are first routed (pass 1) into per-(workgroup, partition)
// private regions -- partition = the table block the key hashes to -- and then (pass 2) each
// partition is aggregated by one workgroup inside an LDS copy of its table block.
struct DevPartition {
  uint64_t* rows;      // [partition][producer][cap_rows][n_words]  (row-major regions)
  uint64_t part_stride;// words between the regions of consecutive partitions of one producer
  uint64_t prod_stride;// words between the regions of consecutive producers of one partition.  Producer-major (default):
                       // part_stride = cap_rows * n_words, prod_stride = n_parts * part_stride (+ pad) -- the 256 streams a
                       // pass-1 workgroup appends to lie within ONE contiguous piece of the scratch (a handful of TLB
                       // entries per workgroup), pass 2 walks its 256 regions one after the other.  Partition-major: the
                       // other way round (round 1's layout; pass 1 then touches a region every part_stride words)
  uint64_t win_stride; // words between consecutive 64-row WINDOWS of one region.  Layouts 0 / 1: 64 rows' worth (a region is
                       // contiguous).  Layout 2 (windowed, default): n_parts * part_stride -- window w of EVERY partition of
                       // a producer lies side by side, so the 256 append positions of a pass-1 workgroup (fills are near
                       // uniform) stay within a megabyte however large the regions are: a handful of TLB entries instead
                       // of one page per region, and the region capacity (deferred pass 2) no longer costs pass 1 anything
  uint32_t* counts;    // [partition][producer]
  uint32_t n_parts;    // table blocks
  uint32_t n_producers;// pass-1 workgroups
  uint32_t cap_rows;   // rows per (producer, partition) region
  uint32_t n_words;    // kw + na
  uint32_t part_shift; // partition = slot >> part_shift
  uint32_t stage_rows; // mode 1: rows of the pass-1 workgroup's LDS write-combining buffer
  uint32_t mode;       // pass 1: 0 direct routing (one 16-byte store per row), 1 LDS counting sort + coalesced copy-out
  uint32_t block;      // pass-1 workgroup size (mode 1: 512 or 1024)
  uint32_t flags;      // PTF_*
  uint32_t ws_scanners;// PTF_WS: scanner waves of the 16: 8 (+ 8 routers: selective scans) or 4 (+ 12 routers: dense scans)
  uint32_t pair_plane; // PTF_PAIR / PTF_PLANES, pass 2: the accumulator plane this launch aggregates
  uint32_t pair_operand;  // PTF_PAIR, pass 2: which of the row's two operands that plane takes (0 or 1)
  uint32_t plane_xf;   // pass 2 of a RAW operand (PTF_PLANES): that accumulator's operand transform (VT_*), filled in by the launcher
  uint32_t plane_spills;  // pass 2, one launch per plane: 1 = this launch is the LAST plane of its operand -- it alone puts the rows whose key
                          // finds no slot in a full block into the spill list, with every accumulator of that operand (the block is as
                          // full for every plane: the earlier launches fail on exactly the same rows and drop them)
  uint32_t pair_ops;   // PTF_PAIR: bit a = the operand (0 / 1) of accumulator a.  Two aggregates: 0b10, each operand transformed by the scan.
                       // PTF_PAIR | PTF_PLANES: three and more aggregates over the two columns -- the operands travel RAW (null-free
                       // batches), every accumulator's pass 2 applies its own transform
  uint32_t pair_arg1;  // PTF_PAIR: an accumulator whose argument is operand 1 (pass 1 evaluates that argument; operand 0 is accumulator 0's)
  uint32_t pair_slot1; // ... its plan column slot and its operand transform, filled in by the launcher once the scan plan is bound (the kernel
  uint32_t pair_xf1;   //     indexes nothing by a run-time accumulator number)
  // Control-block snapshot written BY THE KERNEL (null: none): the last workgroup to finish copies T.ctrl into this
  // host-mapped pinned buffer.  The host reads it after the launch's completion event -- no copy engine, no blit kernel
  // that would have to find room next to 256 persistent 1024-lane workgroups, nothing on a side stream.
  uint32_t* snap_host; // [CTRL_WORDS], host memory (hipHostMalloc)
  uint32_t* snap_done; // device word: workgroups of this launch that have finished (reset by the last one)
};
enum : uint32_t {
  PTF_RESUME = 1u,        // pass 1 appends to the regions as `counts` left them (pass 2 of earlier batches is still pending)
  PTF_STREAM_PASS2 = 2u,  // pass 2: region-streaming kernel (one aggregate, 16-byte rows)
  PTF_HOT = 4u,           // pass 1 (ring flavour, one aggregate): hot-key pairs in LDS (skewed keys)
  PTF_SHARED = 32u,       // with PTF_NARROW, 2..3 aggregates that all take the SAME null-free operand (AVG = SUM + COUNT,
                          // SUM + MIN + MAX of one column ...): routed rows stay {hash image, RAW operand}; pass 2 applies every
                          // aggregate's own transform and atomic to it (n_words is 2 whatever the aggregate count)
  PTF_CHUNK16 = 16u,      // narrow rows, no hot keys: the large-chunk geometry (kNarrowChunkRows / kNarrowRingRows below) -- round 6:
                          // LINE chunks, ten 12-byte rows + 8 bytes of padding = ONE whole 128-byte line per chunk, three-slot rings
  PTF_WS = 64u,           // pass 1, wave-specialised flavour (dfx_k_partition_ws_inl.hpp): DevPartition::ws_scanners of the 16 waves scan,
                          // the others route; needs PTF_NARROW | PTF_CHUNK16, no PTF_HOT / PTF_SHARED
  PTF_PLANES = 256u,      // with PTF_SHARED | PTF_NARROW | PTF_CHUNK16 | PTF_WS: the aggregates' common RAW operand travels through the one-value
                          // pass 1 (LINE chunks, table blocks of 8192 slots) and pass 2 runs once per accumulator plane, applying that
                          // aggregate's transform to the operand (DevPartition::pair_plane) -- instead of 4096-slot blocks with every plane
  PTF_PAIR = 128u,        // with PTF_NARROW | PTF_CHUNK16 | PTF_WS (LINE chunks only): aggregates over TWO operand columns, one scan -- routed
                          // rows are 20 bytes {operand 0, hash image, operand 1} (kPair* below), pass 2 runs once per accumulator plane
                          // (DevPartition::pair_plane / pair_operand) over the same regions: each launch is the one-value kernel with its
                          // 96 KB block.  Two aggregates: the scan transforms each operand for its accumulator; more (+ PTF_PLANES): raw
                          // operands, the transform in pass 2 (DevPartition::pair_ops)
  PTF_NARROW = 8u         // keys below 2^32 (seen by the calibration slice): 12-byte routed rows {hash image, operand}, pass 2
                          // works on 32-bit images; needs ring flavour, one key word, one aggregate
};

// Geometry of the large narrow chunks (PTF_CHUNK16).  tools/ubench5.hip (profiles/r06_ubench5_mi355x.jsonl): the memory system
// takes appended chunks at 4.86 TB/s when every chunk is a whole number of 128-byte lines (128, 256, 384, 768 bytes alike) and
// at 3.44 TB/s when it is not (64 and 192 bytes alike: rounds 2-5 wrote 192-byte chunks of sixteen rows, i.e. a line and a HALF),
// and a scan's reads and its appends do not overlap in time (t = read bytes / 7.2 TB/s + written bytes / that rate, within a few
// percent of every measured launch).  Twelve-byte rows do not tile a line, so a chunk is TEN rows + 8 bytes of padding:
//   row r of a region lives at byte (r / 10) * 128 + (r % 10) * 12; a region's rows are counted in row SLOTS (a multiple of ten);
//   pass 2 reads trips of SIX chunks = 60 rows = 768 bytes (the same trip stride as 64 contiguous rows);
//   the LDS ring of a partition is three 128-byte slots (96 KB for 256 partitions, as the two 16-row slots were).
// -DDFX_LINE_CHUNKS=0 builds the round-5 geometry (sixteen contiguous rows, two slots) for A/B runs (tools/build_variant.py).
#ifndef DFX_LINE_CHUNKS
#define DFX_LINE_CHUNKS 1
#endif
constexpr bool kNarrowLine = DFX_LINE_CHUNKS != 0;
constexpr int kNarrowChunkRows = kNarrowLine ? 10 : 16;  // rows per chunk
constexpr int kNarrowRingSlots = kNarrowLine ? 3 : 2;    // chunk slots per partition ring
constexpr int kNarrowRingRows = kNarrowChunkRows * kNarrowRingSlots;
constexpr int kNarrowSlotBytes = kNarrowLine ? 128 : 192;  // LDS bytes (and region bytes) per chunk
constexpr int kNarrowTripRows = kNarrowLine ? 60 : 64;     // rows of one 768-byte pass-2 trip
constexpr uint32_t kNarrowCapQuantum = kNarrowLine ? 320u : 64u;  // cap_rows of a PTF_CHUNK16 layout is a multiple of this

// PTF_PAIR: a routed row is five dwords {operand 0 lo, operand 0 hi, hash image, operand 1 lo, operand 1 hi}: pass 2 of plane 0
// reads {operand 0, image} at byte 0, pass 2 of plane 1 reads {image, operand 1} at byte 8 -- twelve bytes either way, as of a
// one-value row.  SIX rows + 8 bytes of padding are one 128-byte line (a chunk); row r of a region lives at byte
// (r / 6) * 128 + (r % 6) * 20; pass 2 reads trips of TEN lines = 60 rows = 1280 bytes; three ring slots per partition as above.
constexpr int kPairRowDwords = 5;
constexpr int kPairChunkRows = 6;
constexpr int kPairRingRows = kPairChunkRows * kNarrowRingSlots;
constexpr int kPairTripRows = 60;
constexpr uint32_t kPairTripBytes = 1280u;
constexpr uint32_t kPairCapQuantum = 60u;  // cap_rows of a PTF_PAIR layout: whole chunks and whole trips

// ---- Utf8 GROUP BY keys: device string dictionary (dfx_k_dict.hip) ---------------------------------
enum : int { DICT_POOL = 0, DICT_IDS = 1, DICT_OVERFLOW = 2, DICT_WORDS = 4 };
struct DevDict {
  uint32_t* state;    // per slot: 0 empty, 1 being filled, 2 ready, 3 abandoned (overflow; cleared by the rebuild)
  uint64_t* hash;     // per slot
  uint64_t* sid;      // per slot: string id
  uint64_t* str_off;  // per id: offset of the bytes in `pool`
  uint32_t* str_len;  // per id
  uint8_t* pool;
  uint64_t* cursors;  // DICT_WORDS words: pool bytes used, ids used, overflow flag
  uint64_t mask;      // slots - 1
  int32_t shift;      // 64 - log2(slots)
  int32_t pad;
  uint64_t id_cap;
  uint64_t pool_cap;
};

// ---- CSV source (dfx_k_csv.hip) -------------------------------------------------------------------------
constexpr int kCsvMaxCols = 32;
struct DevCsvCol {
  void* values;        // fixed width: nb values; Boolean: (nb + 63) / 64 bitmap words; Utf8: u64 per record, where the gather finds the cell
  uint64_t* validity;  // (nb + 63) / 64 words (not used for Utf8: csv cells of a Utf8 column are never null)
  int32_t* lens;       // Utf8: unescaped length per record
  uint8_t dtype;
};
struct DevCsvPlan {
  int32_t n_cols;            // schema columns (cells beyond are ignored)
  uint32_t expected_fields;  // fields of the first record: every record must have as many (csv crate, flexible = false)
  uint64_t* null_counts;     // [n_cols]
  uint64_t* general_tiles;   // tiles that took the per-lane walk (a counter)
  uint64_t* err;             // min over failing cells of (record << 16 | column << 8 | code); ~0: none
  DevCsvCol col[kCsvMaxCols];
};

// single-pass FilterRelation (k_filter_fused): bank columns (columns the predicate reads anyway) that are compacted by the
// kernel that evaluates the predicate -- they are read from HBM once
constexpr int kFusedOutCols = 2;
struct DevFusedOut {
  int32_t n;                     // 0..kFusedOutCols
  uint8_t slot[kFusedOutCols];   // column slot of the fused program
  uint8_t dtype[kFusedOutCols];
  void* out[kFusedOutCols];      // room for cap_rows rows
  uint32_t dense;                // the host expects dense tiles (it has seen this stream keep more than a wave can park in LDS)
  uint64_t cap_rows;             // rows past it are not stored (the host sized the buffers from the selectivity it has seen; when
                                 // a batch keeps more, it compacts those columns again from the bitmap: k_compact)
};

struct DevProjectPlan {
  int32_t n_out;
  uint8_t out[kMaxOut];        // operands
  uint8_t out_dtype[kMaxOut];
  void* out_values[kMaxOut];
  uint64_t* out_validity[kMaxOut];  // nullptr: not wanted
};

}  // namespace dfx
