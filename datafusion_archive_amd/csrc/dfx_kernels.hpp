// dfx_kernels.hpp -- host-callable launchers of the gfx950 kernels (implemented in dfx_kernels.hip).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "dfx_device.hpp"

namespace dfx {

constexpr int kTileRows = 4096;  // one compaction tile = 64 bitmap words
constexpr int kBlock = 256;

// kernel ids for the built-in profiler
enum KernelId : int {
  KID_PREDICATE_MASK = 0,
  KID_COMPACT,
  KID_PROJECT,
  KID_REDUCE,
  KID_HASH_AGG,
  KID_MERGE_ROWS,
  KID_REHASH,
  KID_EMIT_MASK,
  KID_FINALIZE,
  KID_SCAN,
  KID_SYNTH,
  KID_FILL,
  KID_GATHER_UTF8,
  KID_PARTIAL,
  KID_PARTITION,
  KID_PARTITION_AGG,
  KID_CSV,
  KID_SORT,
  KID_COUNT_
};
const char* kernel_name(int kid);

// ---- profiler (HIP events around tracked launches, on the launch stream) --------------------
void profile_enable(bool on);
void profile_reset();
int profile_count();
bool profile_get(int index, const char** name, int64_t* launches, double* total_ms, double* algo_bytes);

int device_cu_count();

// ---- scan plans (DevScanPlan, dfx_device.hpp; built in dfx_expr.cpp) -------------------------------------------------
// Shape check at operator-creation time (no batch needed): <= 4 columns of 4 / 8-byte numeric types, a conjunction of
// `column <op> literal` terms, plain-column keys, plain-column arguments.
bool scan_plan_shape_ok(const DevProgram& P, const DevFastPlan& F, int kw, int na, const uint8_t* val_xform);
// Per launch: the plan-order column binding (Cout) and the plan itself (Fout->scan) for the bound batch.  fixed: one key in
// slot 0 and the (first) argument in slot 1 (PlanPolicy1).  False: the shape or this batch's buffers are not covered.
bool bind_scan_plan(const DevProgram& P, const DevFastPlan& F, const DevColumns& C, int kw, int na, const uint8_t* val_xform,
                    bool fixed, DevFastPlan* Fout, DevColumns* Cout);

// K1 predicate_mask: fused compare/AND/OR expression -> Arrow LSB bitmap (one __ballot per 64 rows)
//   replaces comparison_ops!/boolean_ops!/literal_array! closures (expression.rs:171-243, :410-465)
// tile_counts (may be null): popcount per 4096-row tile, for the compaction offsets.
// fast: shape-specialised plan (valid == 0 or nulls present: the generic interpreter runs)
hipError_t launch_mask_and_count(uint64_t* mask, const uint64_t* other, uint32_t* tile_counts, int64_t n, hipStream_t s);
hipError_t launch_predicate_mask(const DevProgram& P, const DevFastPlan& fast, const DevColumns& C, uint8_t pred,
                                 int64_t n, uint64_t* mask_words, uint32_t* tile_counts, uint32_t* ctrl,
                                 double algo_bytes, hipStream_t s);

// K1 + K4 in ONE pass (single-pass FilterRelation, filter.rs:46-110): evaluates the predicate, writes the Arrow bitmap
// words, the per-tile exclusive offsets (decoupled look-back over the tiles' kept counts: what launch_scan_u32 of the tile
// counts would give, tile_offsets[n_tiles] = kept rows) and compacts up to kFusedOutCols of the predicate's own columns
// from the values already in registers.  sync: filter_fused_sync_words(n) words, zeroed before the launch; sync[1] = kept.
size_t filter_fused_sync_words(int64_t n);
hipError_t launch_filter_fused(const DevProgram& P, const DevFastPlan& fast, const DevColumns& C, uint8_t pred, int64_t n,
                               uint64_t* mask_words, uint64_t* tile_offsets, uint64_t* sync, const DevFusedOut& O,
                               uint32_t* ctrl, double algo_bytes, hipStream_t s);

// exclusive scan of uint32 counts into uint64 offsets (out[n] = total); tmp: >= (n/4096 + 2) u64
hipError_t launch_scan_u32(const uint32_t* in, uint64_t* out, int64_t n, uint64_t* tmp, hipStream_t s);
// exclusive scan of int32 lengths into int32 offsets (out[n] = total); tmp as above
hipError_t launch_scan_i32(const int32_t* in, int32_t* out, int64_t n, uint64_t* tmp, hipStream_t s);

// K4 compact: order-preserving stream compaction of one fixed-width column by bitmap
//   replaces fn filter (filter.rs:79-110).  width in {1,2,4,8} bytes.
hipError_t launch_compact(const void* in, int width, const uint64_t* mask_words,
                          const uint64_t* tile_offsets, int64_t n, void* out, double algo_bytes,
                          hipStream_t s, uint64_t out_limit = ~0ull);
// Utf8 support for K4: lengths from offsets, byte gather
hipError_t launch_utf8_lengths(const int32_t* offsets, int64_t n, int32_t* lengths, int32_t* starts, hipStream_t s);
hipError_t launch_utf8_gather(const uint8_t* data, const int32_t* src_starts, const int32_t* dst_offsets,
                              int64_t m, uint8_t* out, hipStream_t s);

// K2/K3 project: fused arithmetic / cast expression outputs (typed values + validity bitmaps)
//   replaces math_ops!/cast_column! closures (expression.rs:131-169, :246-280, :466-493)
hipError_t launch_project(const DevProgram& P, const DevColumns& C, const DevProjectPlan& plan,
                          int64_t n, uint32_t* ctrl, double algo_bytes, hipStream_t s);

// K5 reduce_all: ungrouped aggregates of one batch into per-aggregate batch partials, then the
// scalar fold (accumulate_scalar) into the running state.
//   replaces array_min/max/sum + without_group_by (aggregate.rs:344-546, :703-785)
// partial: na * 4 u64 words {acc, valid_count, first_valid_tag, unused}; state: na * 2 words {has, bits}
hipError_t launch_reduce(const DevProgram& P, const DevFastPlan& fast, const DevColumns& C, const DevAggPlan& plan,
                         const DevTable& T /* kinds / xforms / inits only */, int64_t n,
                         uint64_t* partial, uint32_t* ctrl, double algo_bytes, hipStream_t s);
hipError_t launch_reduce_fold(const DevTable& T, const uint8_t* arg_dtype, const uint8_t* func,
                              uint64_t* partial, uint64_t* state, uint32_t* ctrl, hipStream_t s);

// K6/K7 hash_agg: (optional predicate) + group keys + aggregate arguments -> table updates, with
// an LDS front cache per workgroup and a spill list for rows the table cannot take.
//   replaces with_group_by + update_accumulators (aggregate.rs:787-875, :548-612)
hipError_t launch_hash_agg(const DevProgram& P, const DevFastPlan& fast, const DevColumns& C,
                           const DevAggPlan& plan, const DevTable& T, const DevRows& spill, int64_t n,
                           double algo_bytes, hipStream_t s);
// K7 for a handful of groups (dfx_k_fewgroup.hip): per-lane register accumulators against a wave-uniform key
// dictionary; same table / spill contract as launch_hash_agg.  fewgroup_supported: fast plan, no nulls,
// <= 2 key words, <= 4 aggregates.
bool fewgroup_supported(const DevProgram& P, const DevFastPlan& fast, const DevTable& T);
hipError_t launch_fewgroup_agg(const DevProgram& P, const DevFastPlan& fast, const DevColumns& C,
                               const DevAggPlan& plan, const DevTable& T, const DevRows& spill, int64_t n,
                               double algo_bytes, hipStream_t s);
// insert pre-evaluated rows (spill replays, LDS flushes of other ranks, all-to-all imports)
hipError_t launch_merge_rows(const DevRows& rows, int64_t row_begin, int64_t n_rows, const DevTable& T,
                             const DevRows& spill, hipStream_t s);
hipError_t launch_rehash(const DevTable& from, const DevTable& to, const DevRows& spill, hipStream_t s);
hipError_t launch_fill_u64(uint64_t* p, uint64_t v, int64_t n, hipStream_t s);
// device -> host-mapped pinned memory by a kernel on `s` (falls back to hipMemcpyAsync for unaligned pointers)
hipError_t launch_copy_to_host(const void* src_device, void* dst_pinned, size_t bytes, hipStream_t s);
hipError_t launch_fill_u32(uint32_t* p, uint32_t v, int64_t n, hipStream_t s);

// K8 emit_groups: occupancy bitmap of the table (then launch_scan_u32 + launch_compact on each
// plane), and the typed finalisation of one output column.
//   replaces the result macros (aggregate.rs:633-699, :877-951)
hipError_t launch_table_mask(const DevTable& T, uint64_t* mask_words, uint32_t* tile_counts, hipStream_t s);
// in: dense u64 plane; out: typed column. is_key: narrow only. xform/kind/func describe the agg.
hipError_t launch_finalize(const uint64_t* in, int64_t n, uint8_t out_dtype, uint8_t val_xform,
                           void* out, hipStream_t s);
// AVG columns: sum / count in the argument's type, validity bitmap (count == 0 -> null), null count
hipError_t launch_finalize_avg(const uint64_t* sum, const uint64_t* cnt, int64_t n, uint8_t out_dtype, void* out,
                               uint64_t* validity, uint64_t* null_count, hipStream_t s);
uint64_t host_avg_value(uint8_t t, uint64_t sum, uint64_t cnt);

// multi-GPU partial export: count per destination rank, then scatter into bucketed planes
hipError_t launch_partial_count(const DevTable& T, int world, uint64_t* counts, hipStream_t s);
hipError_t launch_partial_scatter(const DevTable& T, int world, const uint64_t* bucket_base,
                                  const uint64_t* bucket_count, uint64_t* cursors, uint64_t* dst,
                                  hipStream_t s);
// merge bucketed planes (layout of dfx_aggregate_partial_export) into a table
hipError_t launch_merge_bucket(const uint64_t* bucket, uint64_t count, const DevTable& T,
                               const DevRows& spill, hipStream_t s);

// Utf8 GROUP BY keys (dfx_k_dict.hip): strings -> stable 64-bit ids and back
hipError_t launch_dict_encode(const int32_t* offsets, const uint8_t* data, int64_t n, const DevDict& D, uint64_t n_known, uint64_t* ids,
                              hipStream_t s);  // n_known: ids the dictionary already holds (complete: written by earlier launches)
hipError_t launch_dict_rebuild(const DevDict& D, uint64_t n_ids, hipStream_t s);
hipError_t launch_dict_remap_plane(uint64_t* plane, uint64_t n_slots, const uint64_t* remap, uint64_t n_ids, hipStream_t s);
hipError_t launch_dict_lengths(const uint64_t* ids, int64_t g, const DevDict& D, uint32_t* lens, hipStream_t s);
hipError_t launch_dict_gather(const uint64_t* ids, int64_t g, const DevDict& D, const uint64_t* starts, int32_t* offsets,
                              uint8_t* out, hipStream_t s);

// partitioned GROUP BY (dfx_k_partition.hip): pass 1 routes passing rows to per-(producer, partition)
// regions, pass 2 aggregates every partition in an LDS copy of its table block.  Single-word keys.
size_t partition_stage_bytes(const DevPartition& PT);
bool partition_planes_supported(const DevProgram& P, const DevFastPlan& fast, const DevColumns& C, const DevTable& T);  // PTF_PLANES: ... the one-value kernels with the raw operand
bool partition_pair_supported(const DevProgram& P, const DevFastPlan& fast, const DevColumns& C, const DevTable& T);  // PTF_PAIR: this bound batch fits the pair kernels
hipError_t launch_probe_wide_keys(const DevTable& T, hipStream_t s);
size_t partition_ring_bytes(uint32_t n_words, uint32_t n_parts, int ring_rows, bool hot = false, bool narrow = false, int queue_rows = 0);
size_t partition_ws_bytes(uint32_t n_parts, int scanner_waves, int operands = 1);  // LDS of the wave-specialised pass-1 kernel (PTF_WS)
uint32_t partition_sort_capacity(uint32_t n_words, uint32_t n_parts, uint32_t block, size_t lds_budget);
hipError_t launch_partition(const DevProgram& P, const DevFastPlan& fast, const DevColumns& C, const DevAggPlan& plan,
                            const DevTable& T, const DevPartition& PT, const DevRows& spill, int64_t n,
                            double algo_bytes, hipStream_t s);
hipError_t launch_partition_agg(const DevTable& T, const DevPartition& PT, const DevRows& spill, double algo_bytes,
                                hipStream_t s);

// CSV text -> Arrow columns (dfx_k_csv.hip): record boundaries by parallel simulation of the csv automaton, then one
// thread per record converts the cells.  buf must be readable up to the next multiple of 32 bytes past n.
//   boundaries_count: tile_trans / tile_state / tile_counts have one entry per csv_tile_bytes() of text;
//   (scan tile_counts into tile_offsets, total = number of records) then boundaries_write fills row_start[0..total).
int64_t csv_tile_bytes();
hipError_t launch_csv_boundaries_count(const uint8_t* buf, uint64_t n, uint32_t* tile_trans, uint32_t* block_vec, uint8_t* tile_state,
                                       uint32_t* tile_counts, hipStream_t s);
hipError_t launch_csv_boundaries_write(const uint8_t* buf, uint64_t n, const uint32_t* tile_trans, const uint8_t* tile_state,
                                       const uint64_t* tile_offsets, uint64_t* row_start, hipStream_t s);
hipError_t launch_csv_count_fields(const uint8_t* buf, const uint64_t* row_start, int64_t row, uint32_t* out,
                                   hipStream_t s);
hipError_t launch_csv_parse(const uint8_t* buf, const uint64_t* row_start, int64_t r0, int64_t nb,
                            const DevCsvPlan& plan, double avg_record_bytes, int wave_tiles, double algo_bytes, hipStream_t s);
hipError_t launch_csv_utf8_gather(const uint8_t* buf, const uint64_t* row_start, int64_t r0, int64_t nb, int field,
                                  const int32_t* offsets, const uint64_t* starts, uint8_t* out, hipStream_t s);

// ORDER BY (dfx_k_sort.hip): order-preserving key images, stable LSD radix sort of (image, row) pairs, gathers
hipError_t launch_sort_image(const void* values, const uint8_t* validity, int64_t bit_offset, uint8_t dtype, int asc, int64_t n,
                             uint64_t* image, uint64_t* null_image, hipStream_t s);
// Utf8 key: chunk >= 0: big-endian image of bytes [8 * chunk, 8 * chunk + 8) zero padded; chunk < 0: the length.  max_len (may be
// null): atomicMax of the string lengths
hipError_t launch_sort_image_utf8(const int32_t* offsets, const uint8_t* data, const uint8_t* validity, int64_t bit_offset, int chunk,
                                  int asc, int64_t n, uint64_t* image, uint64_t* null_image, uint32_t* max_len, hipStream_t s);
hipError_t launch_sort_iota(uint32_t* idx, int64_t n, hipStream_t s);
hipError_t launch_sort_gather_u64(const uint64_t* src, const uint32_t* idx, int64_t n, uint64_t* dst, hipStream_t s);
hipError_t launch_radix_hist8(const uint64_t* img, int64_t n, uint64_t* hist /* [8][256], zeroed */, hipStream_t s);
// top-k: histogram of one digit over the elements matching a prefix (radix select), bitmap of img <= threshold
hipError_t launch_select_hist(const uint64_t* img, int64_t n, uint64_t prefix, uint64_t mask, int shift, uint64_t* hist /* [256], zeroed */,
                              hipStream_t s);
hipError_t launch_select_mask(const uint64_t* img, int64_t n, uint64_t threshold, uint64_t* mask_words, uint32_t* tile_counts,
                              hipStream_t s);
int64_t radix_tiles(int64_t n);  // counts / offsets hold 256 * radix_tiles(n) entries, digit-major
hipError_t launch_radix_count(const uint64_t* img, int64_t n, int shift, uint32_t* counts, hipStream_t s);
hipError_t launch_radix_scatter(const uint64_t* img_in, const uint32_t* idx_in, int64_t n, int shift, const uint64_t* offsets,
                                uint64_t* img_out, uint32_t* idx_out, hipStream_t s);
hipError_t launch_sort_locate(const uint32_t* idx, int64_t n, const uint64_t* starts, int nb, uint64_t* loc, hipStream_t s);
hipError_t launch_gather_fixed(const void* const* bases, const uint64_t* loc, int64_t n, int width, void* out, hipStream_t s);
hipError_t launch_gather_bits(const uint8_t* const* bases, const int64_t* bit_offsets, const uint64_t* loc, int64_t n,
                              uint64_t* out, uint64_t* zero_count, hipStream_t s);
hipError_t launch_gather_utf8_lens(const int32_t* const* offsets, const uint64_t* loc, int64_t n, int32_t* lens, hipStream_t s);
hipError_t launch_gather_utf8_copy(const int32_t* const* offsets, const uint8_t* const* data, const uint64_t* loc, int64_t n,
                                   const int32_t* dst_offsets, uint8_t* out, hipStream_t s);

// synthetic columns (definition shared with oracle/dfx_oracle.c: orc_synth_fill)
hipError_t launch_synth(int kind, int column_id, double p0, double p1, uint64_t seed, int64_t row_begin,
                        int64_t n, void* out, hipStream_t s);
// validity bitmap of a synthetic column with nulls (DFX_SYNTH_NULL_PERMILLE): (n + 63) / 64 words; *nulls += null rows
hipError_t launch_synth_validity(int column_id, uint32_t permille, uint64_t seed, int64_t row_begin, int64_t n, uint64_t* words,
                                 uint64_t* nulls, hipStream_t s);

// host mirror of the device hash (rank ownership in tests)
uint64_t host_hash_keys(const uint64_t* key, int kw);
uint32_t host_unhash_word32(uint32_t image);

}  // namespace dfx
