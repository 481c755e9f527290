/*
 * dfx_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A single-threaded plain-C restatement of the reference algorithm for the hot path
 * (andygrove/datafusion-archive src/execution::{expression,filter,projection,aggregate}).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library,
 * and only as the checker.  The product (datafusion_archive_amd/csrc) never links or calls it.
 *
 * Parity pinning: the reference cannot be built here (no Rust toolchain) and the arithmetic it
 * delegates to lives in the un-vendored crate `arrow = "0.12.0"` (Cargo.toml:28).  The oracle is
 * pinned against every golden vector the reference's own tests hold for this path
 * (tests/sql.rs:29-77, src/execution/aggregate.rs:965-1127, src/execution/projection.rs:83-103)
 * in tests/test_oracle_golden.py.  Behaviour the reference tests do not exercise is restated
 * from arrow 0.12's published array_ops semantics and marked "unpinned" at the function.
 */
#ifndef DFX_ORACLE_H
#define DFX_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#include "../include/dfx.h" /* dfx_expr_node, dfx_dtype, dfx_operator, dfx_status: shared vocabulary only */

#ifdef __cplusplus
extern "C" {
#endif

/* An Arrow-like array. Primitive: `values` = length * sizeof(T). Boolean: `values` = LSB-first
 * bitmap. Utf8: offsets (length+1 int32) + data. validity: LSB-first bitmap or NULL (= all valid). */
typedef struct orc_array {
  int32_t dtype;
  int32_t owned; /* 1: buffers malloc'd by the oracle (orc_array_free frees them) */
  int64_t length;
  void* values;
  uint8_t* validity;
  int32_t* offsets;
  uint8_t* data;
} orc_array;

typedef struct orc_batch {
  int64_t num_rows;
  int32_t num_columns;
  int32_t owned;
  orc_array** columns;
} orc_batch;

void orc_array_free(orc_array* a);
void orc_batch_free(orc_batch* b);

/* compile_scalar_expr + closure evaluation (expression.rs:283-505): evaluate node `root` over
 * the batch; every node materialises a full array exactly like the reference closures do. */
int32_t orc_eval(const dfx_expr_node* nodes, int32_t n_nodes, int32_t root, const orc_batch* batch,
                 orc_array** out, char* err, size_t errlen);

/* FilterRelation::next body (filter.rs:46-71) for one batch: evaluate predicate, require
 * Boolean, compact every column with fn filter (filter.rs:79-110). */
int32_t orc_filter_next(const dfx_expr_node* nodes, int32_t n_nodes, int32_t root,
                        const orc_batch* batch, orc_batch** out, char* err, size_t errlen);

/* ProjectRelation::next body (projection.rs:46-66) for one batch. */
int32_t orc_project_next(const dfx_expr_node* nodes, int32_t n_nodes, const int32_t* roots,
                         int32_t n_roots, const orc_batch* batch, orc_batch** out, char* err,
                         size_t errlen);

/* AggregateRelation (aggregate.rs:614-952): push input batches in order, then finish. */
typedef struct orc_agg orc_agg;
int32_t orc_agg_new(const dfx_expr_node* nodes, int32_t n_nodes, const int32_t* group_roots,
                    int32_t n_group, const int32_t* aggr_roots, int32_t n_aggr, orc_agg** out,
                    char* err, size_t errlen);
int32_t orc_agg_push(orc_agg* agg, const orc_batch* batch, char* err, size_t errlen);
int32_t orc_agg_finish(orc_agg* agg, orc_batch** out, char* err, size_t errlen);
void orc_agg_free(orc_agg* agg);

/* Synthetic data generator shared (by definition, not by code) with the device generator. */
uint64_t orc_synth_u64(uint64_t seed, int32_t column_id, int64_t row);
int32_t orc_synth_fill(int32_t kind, int32_t column_id, double p0, double p1, uint64_t seed,
                       int64_t row_begin, int64_t n, void* out);
/* validity bitmap of a synthetic column with nulls (include/dfx.h: DFX_SYNTH_NULL_PERMILLE); returns the null count */
int64_t orc_synth_validity(int32_t column_id, int32_t permille, uint64_t seed, int64_t row_begin, int64_t n, uint8_t* bits);

/* CPU baseline: run `[filter ->] aggregate` reference-shaped (1024-row batches, materialised
 * literal arrays, per-row hash map) over synthetic columns; returns wall seconds via *seconds and
 * the result batch (may be NULL to discard). filter_root < 0: no filter. */
int32_t orc_run_synth_query(const dfx_synth_column* cols, int32_t n_cols, uint64_t seed,
                            int64_t row_begin, int64_t n_rows, int64_t batch_rows,
                            const dfx_expr_node* nodes, int32_t n_nodes, int32_t filter_root,
                            const int32_t* group_roots, int32_t n_group, const int32_t* aggr_roots,
                            int32_t n_aggr, int32_t mask_only, double* seconds, orc_batch** out,
                            int64_t* rows_out, char* err, size_t errlen);

/* FilterRelation reference-shaped over synthetic columns: the compacted batches concatenated into out_values[c] (room for
 * n_rows 8-byte elements each; entries / the array may be NULL) and the predicate's BooleanArray concatenated into
 * mask_bits (LSB first, (n_rows + 7) / 8 bytes; may be NULL). */
int32_t orc_run_synth_filter(const dfx_synth_column* cols, int32_t n_cols, uint64_t seed, int64_t row_begin,
                             int64_t n_rows, int64_t batch_rows, const dfx_expr_node* nodes, int32_t n_nodes,
                             int32_t filter_root, void** out_values, uint8_t* mask_bits, int64_t* rows_out,
                             double* seconds, char* err, size_t errlen);

/* CsvDataSource::new(filename, schema, batch_size) + next() (datasource.rs:33-58): arrow 0.12 csv::Reader with
 * has_headers = true over the csv crate's defaults.  *out == NULL at end of input (Ok(None)). */
typedef struct orc_csv orc_csv;
int32_t orc_csv_open(const char* filename, const int32_t* dtypes, int32_t n_cols, int64_t batch_size, orc_csv** out,
                     char* err, size_t errlen);
int32_t orc_csv_next(orc_csv* c, orc_batch** out, char* err, size_t errlen);
void orc_csv_close(orc_csv* c);

#ifdef __cplusplus
}
#endif
#endif
