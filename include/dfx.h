/*
 * dfx.h -- C ABI of the MI355X-native filter / projection / aggregate execution path.
 *
 * This is the drop-in boundary for andygrove/datafusion-archive's src/execution::{filter,
 * projection,aggregate} and its RecordBatch expression evaluator.  Every entry point below
 * names the reference interface it replaces (paths relative to the reference repo root).
 *
 * Conventions
 *  - plain C, no C++/HIP/torch types in any signature;
 *  - RecordBatches cross the boundary as Arrow C Data Interface structs; the pull-based
 *    `trait Relation { next(); schema() }` (src/execution/relation.rs:27-32) crosses it as the
 *    Arrow C Stream Interface: get_next()==Relation::next() (a released array == Ok(None)),
 *    get_schema()==Relation::schema();
 *  - every function returns a dfx_status (0 = OK); the code mirrors the variant of
 *    `ExecutionError` (src/execution/error.rs:26-36) the reference would have produced.  Message
 *    text goes to the caller-provided `err` buffer (or ArrowArrayStream.get_last_error for
 *    stream callbacks).  No C++ exception, HIP error or panic crosses this ABI;
 *  - handles are thread-confined (the reference is Rc<RefCell<..>>, i.e. !Send + !Sync).
 */
#ifndef DFX_H
#define DFX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------
 * Arrow C Data / C Stream / C Device interfaces (standard ABI, verbatim struct layouts).
 * ---------------------------------------------------------------------------------------- */
#ifndef ARROW_C_DATA_INTERFACE
#define ARROW_C_DATA_INTERFACE
#define ARROW_FLAG_DICTIONARY_ORDERED 1
#define ARROW_FLAG_NULLABLE 2
#define ARROW_FLAG_MAP_KEYS_SORTED 4
struct ArrowSchema {
  const char* format;
  const char* name;
  const char* metadata;
  int64_t flags;
  int64_t n_children;
  struct ArrowSchema** children;
  struct ArrowSchema* dictionary;
  void (*release)(struct ArrowSchema*);
  void* private_data;
};
struct ArrowArray {
  int64_t length;
  int64_t null_count;
  int64_t offset;
  int64_t n_buffers;
  int64_t n_children;
  const void** buffers;
  struct ArrowArray** children;
  struct ArrowArray* dictionary;
  void (*release)(struct ArrowArray*);
  void* private_data;
};
#endif

#ifndef ARROW_C_STREAM_INTERFACE
#define ARROW_C_STREAM_INTERFACE
struct ArrowArrayStream {
  int (*get_schema)(struct ArrowArrayStream*, struct ArrowSchema* out);
  int (*get_next)(struct ArrowArrayStream*, struct ArrowArray* out);
  const char* (*get_last_error)(struct ArrowArrayStream*);
  void (*release)(struct ArrowArrayStream*);
  void* private_data;
};
#endif

/* ------------------------------------------------------------------------------------------
 * Status codes == ExecutionError variants (src/execution/error.rs:26-36), same order.
 * ---------------------------------------------------------------------------------------- */
typedef enum dfx_status {
  DFX_OK = 0,
  DFX_IO_ERROR = 1,        /* ExecutionError::IoError        */
  DFX_PARSER_ERROR = 2,    /* ExecutionError::ParserError    */
  DFX_GENERAL = 3,         /* ExecutionError::General        */
  DFX_INVALID_COLUMN = 4,  /* ExecutionError::InvalidColumn  */
  DFX_NOT_IMPLEMENTED = 5, /* ExecutionError::NotImplemented (also the reference's unimplemented!()) */
  DFX_INTERNAL_ERROR = 6,  /* ExecutionError::InternalError  (also the reference's panic!()/unwrap()) */
  DFX_ARROW_ERROR = 7,     /* ExecutionError::ArrowError     (e.g. arrow DivideByZero) */
  DFX_EXECUTION_ERROR = 8  /* ExecutionError::ExecutionError (also HIP / RCCL failures) */
} dfx_status;

/* arrow::datatypes::DataType subset used by the path (expression.rs:135-166). */
typedef enum dfx_dtype {
  DFX_TYPE_NONE = 0,
  DFX_BOOLEAN = 1,
  DFX_INT8 = 2,
  DFX_INT16 = 3,
  DFX_INT32 = 4,
  DFX_INT64 = 5,
  DFX_UINT8 = 6,
  DFX_UINT16 = 7,
  DFX_UINT32 = 8,
  DFX_UINT64 = 9,
  DFX_FLOAT32 = 10,
  DFX_FLOAT64 = 11,
  DFX_UTF8 = 12
} dfx_dtype;

/* logicalplan::Operator (src/logicalplan.rs:67-84), same order. */
typedef enum dfx_operator {
  DFX_OP_EQ = 0,
  DFX_OP_NOT_EQ = 1,
  DFX_OP_LT = 2,
  DFX_OP_LT_EQ = 3,
  DFX_OP_GT = 4,
  DFX_OP_GT_EQ = 5,
  DFX_OP_PLUS = 6,
  DFX_OP_MINUS = 7,
  DFX_OP_MULTIPLY = 8,
  DFX_OP_DIVIDE = 9,
  DFX_OP_MODULUS = 10,
  DFX_OP_AND = 11,
  DFX_OP_OR = 12,
  DFX_OP_NOT = 13,
  DFX_OP_LIKE = 14,
  DFX_OP_NOT_LIKE = 15
} dfx_operator;

/* logicalplan::Expr variants (src/logicalplan.rs:136-167), same order. */
typedef enum dfx_expr_kind {
  DFX_EXPR_COLUMN = 0,
  DFX_EXPR_LITERAL = 1,
  DFX_EXPR_BINARY = 2,
  DFX_EXPR_IS_NOT_NULL = 3,
  DFX_EXPR_IS_NULL = 4,
  DFX_EXPR_CAST = 5,
  DFX_EXPR_SORT = 6,
  DFX_EXPR_SCALAR_FUNCTION = 7,
  DFX_EXPR_AGGREGATE_FUNCTION = 8
} dfx_expr_kind;

/*
 * One node of a serialised logicalplan::Expr tree.  A tree is an array of nodes in any order
 * with child links by index; `root` selects the top node.  This is what a Rust shim derives
 * from the `&Expr` it is handed by ExecutionContext::execute (src/execution/context.rs:132,
 * :155,:174,:180) -- the reference's RuntimeExpr holds opaque closures, so the tree itself has
 * to cross the boundary.
 */
typedef struct dfx_expr_node {
  int32_t kind;   /* dfx_expr_kind */
  int32_t op;     /* BINARY: dfx_operator */
  int32_t dtype;  /* LITERAL: literal's type (DFX_TYPE_NONE == ScalarValue::Null);
                     CAST: target type; AGGREGATE_FUNCTION / SCALAR_FUNCTION: return_type */
  int32_t left;   /* BINARY: left child; CAST / AGGREGATE / IS_NULL / IS_NOT_NULL / SORT: the child; else -1 */
  int32_t right;  /* BINARY: right child; else -1 */
  int32_t column; /* COLUMN: column index */
  int32_t n_args; /* AGGREGATE_FUNCTION / SCALAR_FUNCTION: number of arguments (reference asserts 1,
                     expression.rs:91); the first argument is `left` */
  int32_t reserved;
  union {
    int64_t i64;  /* Int8..Int64 literals, sign-extended */
    uint64_t u64; /* UInt8..UInt64 literals, zero-extended; Boolean: 0/1 */
    double f64;   /* Float64 literal */
    float f32;    /* Float32 literal */
  } lit;
  const char* name; /* AGGREGATE_FUNCTION / SCALAR_FUNCTION: function name (matched case-insensitively,
                       expression.rs:98); Utf8 literal: the string; else NULL */
} dfx_expr_node;

/* ------------------------------------------------------------------------------------------
 * Library / device
 * ---------------------------------------------------------------------------------------- */
/* ABI version of this header. */
#define DFX_ABI_VERSION 1
int32_t dfx_abi_version(void);

/* Bind the calling process to one GPU (one process per GPU).  Must precede any other call that
 * touches the device; default device is 0.  No reference equivalent (reference is CPU-only). */
int32_t dfx_init(int32_t device_ordinal, char* err, size_t errlen);
/* Device facts for reports: name, CU count, HBM bytes, wavefront size. */
int32_t dfx_device_info(char* name, size_t namelen, int32_t* n_cu, int64_t* hbm_bytes,
                        int32_t* wavefront, char* err, size_t errlen);
/* Block until every stream of the library is idle (bench bracketing). */
int32_t dfx_synchronize(char* err, size_t errlen);

/* ------------------------------------------------------------------------------------------
 * Expression compiler.
 *   replaces: compile_scalar_expr (src/execution/expression.rs:283-505)
 *             compile_expr        (src/execution/expression.rs:80-121)
 *             RuntimeExpr::{get_name,get_type} (expression.rs:56-77)
 * Validation errors map 1:1 on the reference's Err cases (unsupported literal type :306-309,
 * unsupported operator :494-497, unsupported expression :500-503, cast rules :316-379,
 * unsupported aggregate :103-106).
 * ---------------------------------------------------------------------------------------- */
typedef struct dfx_runtime_expr dfx_runtime_expr;

int32_t dfx_compile_scalar_expr(const dfx_expr_node* nodes, int32_t n_nodes, int32_t root,
                                const struct ArrowSchema* input_schema, dfx_runtime_expr** out,
                                char* err, size_t errlen);
int32_t dfx_compile_expr(const dfx_expr_node* nodes, int32_t n_nodes, int32_t root,
                         const struct ArrowSchema* input_schema, dfx_runtime_expr** out, char* err,
                         size_t errlen);
const char* dfx_runtime_expr_name(const dfx_runtime_expr* e); /* RuntimeExpr::get_name */
int32_t dfx_runtime_expr_type(const dfx_runtime_expr* e);     /* RuntimeExpr::get_type -> dfx_dtype */
int32_t dfx_runtime_expr_is_aggregate(const dfx_runtime_expr* e);
void dfx_runtime_expr_free(dfx_runtime_expr* e);

/* ------------------------------------------------------------------------------------------
 * Operators.  Each consumes (takes ownership of) an input stream and fills `out` with the
 * operator's own stream; out->release frees the whole sub-tree, like dropping the
 * Rc<RefCell<Relation>>.  When `input` is itself a stream produced by this library the operators
 * are chained on the device (no host round trip) and Filter feeding Aggregate is fused into one
 * kernel.  The runtime exprs are borrowed for the duration of the call only.
 * ---------------------------------------------------------------------------------------- */

/* replaces FilterRelation::new(input, expr, schema) + impl Relation (src/execution/filter.rs:36-77)
 * and fn filter (filter.rs:79-110). */
int32_t dfx_filter_relation_new(struct ArrowArrayStream* input, const dfx_runtime_expr* expr,
                                const struct ArrowSchema* schema, struct ArrowArrayStream* out,
                                char* err, size_t errlen);

/* replaces ProjectRelation::new(input, expr, schema) + impl Relation (src/execution/projection.rs:36-71). */
int32_t dfx_project_relation_new(struct ArrowArrayStream* input, const dfx_runtime_expr* const* exprs,
                                 int32_t n_exprs, const struct ArrowSchema* schema,
                                 struct ArrowArrayStream* out, char* err, size_t errlen);

/* replaces AggregateRelation::new(schema, input, group_expr, aggr_expr) + impl Relation
 * (src/execution/aggregate.rs:47-61, :614-631, :703-952).  `schema` may be NULL or empty
 * (context.rs:185 passes Schema::empty()). */
int32_t dfx_aggregate_relation_new(const struct ArrowSchema* schema, struct ArrowArrayStream* input,
                                   const dfx_runtime_expr* const* group_exprs, int32_t n_group,
                                   const dfx_runtime_expr* const* aggr_exprs, int32_t n_aggr,
                                   struct ArrowArrayStream* out, char* err, size_t errlen);

/* Per-operator options.  The reference's operators take no tuning knobs; this library's strategy switches (dfx_set_option
 * below, process-wide defaults kept for benchmarks and debugging) can also be given to ONE operator: it starts from the
 * process defaults as they are when it first runs, applies `options` on top, and is not affected by later dfx_set_option
 * calls.  Keys are the ones dfx_set_option documents ("agg.strategy", "scan.fast", "filter.single_pass", ...); an unknown
 * key is DFX_GENERAL.  No global mutable state is involved (thread-confined handles). */
typedef struct dfx_option {
  const char* key;
  int64_t value;
} dfx_option;
int32_t dfx_filter_relation_new_with_options(struct ArrowArrayStream* input, const dfx_runtime_expr* expr,
                                             const struct ArrowSchema* schema, const dfx_option* options, int32_t n_options,
                                             struct ArrowArrayStream* out, char* err, size_t errlen);
int32_t dfx_aggregate_relation_new_with_options(const struct ArrowSchema* schema, struct ArrowArrayStream* input,
                                                const dfx_runtime_expr* const* group_exprs, int32_t n_group,
                                                const dfx_runtime_expr* const* aggr_exprs, int32_t n_aggr,
                                                const dfx_option* options, int32_t n_options,
                                                struct ArrowArrayStream* out, char* err, size_t errlen);

/* ------------------------------------------------------------------------------------------
 * HBM-resident tables: the in-memory DataSource (src/execution/datasource.rs:27-30 trait
 * DataSource; relation.rs:34-54 DataSourceRelation).  A table is uploaded (or generated) once
 * into HBM and can be scanned many times; the scan is a library stream, so operators stacked on
 * it never leave the device.
 * ---------------------------------------------------------------------------------------- */
typedef struct dfx_table dfx_table;

/* Drain `input` (host Arrow batches), copy every column to HBM. Consumes the stream. */
int32_t dfx_table_from_stream(struct ArrowArrayStream* input, dfx_table** out, char* err,
                              size_t errlen);

/* Synthetic column generators (deterministic per (seed, column id, global row index); the CPU
 * oracle reproduces any slice -- SURVEY.md section 8(d)). */
typedef enum dfx_synth_kind {
  DFX_SYNTH_F64_UNIFORM = 0, /* p0 + p1 * u,  u in [0,1) 53-bit          (dtype Float64) */
  DFX_SYNTH_F64_EXACT = 1,   /* m * 2^-S, m uniform integer in [0,2^B): B = p0 (0: 20), S = p1 (0: 10)  (dtype Float64) */
  DFX_SYNTH_I64_UNIFORM = 2, /* uniform integer in [0, (int64)p0)          (dtype Int64)   */
  DFX_SYNTH_I64_ZIPF = 3,    /* floor(p0 ^ u) - 1 clipped to [0,p0): log-uniform skew (dtype Int64) */
  DFX_SYNTH_I32_UNIFORM = 4, /* uniform integer in [0, (int32)p0): the same draw as I64_UNIFORM, stored in 4 bytes (dtype Int32) */
  DFX_SYNTH_I64_WIDE = 5     /* (u + 1) * 0x9E3779B97F4A7C15 mod 2^64 with u the I64_UNIFORM draw in [0, (int64)p0): p0 distinct keys
                              * spread over all of Int64 -- hashed ids, both signs, (almost) none below 2^32 (dtype Int64).  The
                              * reference takes any Int64 key (aggregate.rs:807-852) */
} dfx_synth_kind;
/* Nulls: `kind | (permille << 8)` gives the column a validity bitmap in which a row is NULL with probability permille / 1000
 * (1 .. 1000), decided by a draw of its own -- DFX_SYNTH_NULL_STREAM mixed into the column id -- so the values under the null
 * slots are ordinary values: grouped aggregates read value(row) without a null check (aggregate.rs:561-603). */
#define DFX_SYNTH_KIND(k) ((k) & 0xFF)
#define DFX_SYNTH_NULL_PERMILLE(k) (((k) >> 8) & 0x3FF)
#define DFX_SYNTH_NULL_STREAM 0x4E554C4C
typedef struct dfx_synth_column {
  const char* name;
  int32_t kind;      /* dfx_synth_kind, optionally | (null permille << 8) */
  int32_t column_id; /* stream id mixed into the generator */
  double p0, p1;
} dfx_synth_column;
int32_t dfx_table_synth(const dfx_synth_column* cols, int32_t n_cols, uint64_t seed,
                        int64_t row_begin, int64_t n_rows, dfx_table** out, char* err,
                        size_t errlen);

int64_t dfx_table_num_rows(const dfx_table* t);
int32_t dfx_table_num_columns(const dfx_table* t);
/* Raw device pointer of a column's values buffer (for torch/RCCL plumbing and debugging). */
const void* dfx_table_column_device_ptr(const dfx_table* t, int32_t column);

/* DataSourceRelation over a resident table: yields slices of `batch_rows` rows (<=0: one batch).
 * The table must outlive the stream. */
int32_t dfx_table_scan_new(const dfx_table* t, int64_t batch_rows, struct ArrowArrayStream* out,
                           char* err, size_t errlen);
/* ... over the rows [row_begin, row_begin + n_rows) of the table only (row_begin a multiple of 64: slices stay
 * byte-aligned in every bitmap; n_rows < 0: to the end).  A partition of a resident table as a DataSource of its own
 * (relation.rs:34-54 wraps whatever DataSource it is given): what bench.py uses to check the last rows of the
 * 10^10-row table -- row indices beyond 2^32 -- against the oracle. */
int32_t dfx_table_scan_range_new(const dfx_table* t, int64_t row_begin, int64_t n_rows, int64_t batch_rows,
                                 struct ArrowArrayStream* out, char* err, size_t errlen);
void dfx_table_free(dfx_table* t);

/* ------------------------------------------------------------------------------------------
 * CSV data source.  replaces CsvDataSource::new(filename, schema, batch_size) + impl DataSource
 * (src/execution/datasource.rs:33-58; wrapped by DataSourceRelation, relation.rs:34-54).  As in the
 * reference the arrow csv reader is created with has_headers = true: the FIRST RECORD IS ALWAYS
 * CONSUMED AS A HEADER.  The text is copied to HBM once; record boundaries, cell conversion
 * (Rust `str::parse` semantics, correctly rounded floats) and Utf8 extraction run on the device,
 * batch_size records per get_next().  Errors: a missing file is the reference's unwrap() panic
 * (DFX_INTERNAL_ERROR); a cell that does not parse is DFX_ARROW_ERROR "Error while parsing value
 * {cell} at line {n}"; a record with a different field count than the first is DFX_ARROW_ERROR.
 * The stream is a library stream: operators stacked on it never leave the device.
 * ---------------------------------------------------------------------------------------- */
int32_t dfx_csv_datasource_new(const char* filename, const struct ArrowSchema* schema, int64_t batch_size,
                               struct ArrowArrayStream* out, char* err, size_t errlen);

/* ------------------------------------------------------------------------------------------
 * ORDER BY / LIMIT.  The operators behind LogicalPlan::Sort { expr: [Expr::Sort { expr, asc }], input, schema } and
 * LogicalPlan::Limit { limit, input, schema } (src/logicalplan.rs:313-338), which the reference's planner emits
 * (sqlplanner.rs:142-183) and its executor leaves at unimplemented!() (context.rs:113,194): there is no reference
 * behaviour, the semantics are this library's (parity unpinned): stable sort, NULL larger than every value (last
 * when ascending, first when descending), NaN larger than every number, ONE result batch.  `exprs` are the sort
 * expressions compiled with dfx_compile_scalar_expr (the inner `expr` of Expr::Sort; compile_scalar_expr itself
 * rejects Expr::Sort like the reference, expression.rs:380-399), `ascending[i]` their direction.  Keys: any
 * fixed-width scalar expression, or a Utf8 column (byte-wise lexicographic).  LIMIT keeps the first `limit` rows.
 * ---------------------------------------------------------------------------------------- */
int32_t dfx_sort_relation_new(struct ArrowArrayStream* input, const dfx_runtime_expr* const* exprs,
                              const int32_t* ascending, int32_t n_exprs, const struct ArrowSchema* schema,
                              struct ArrowArrayStream* out, char* err, size_t errlen);
int32_t dfx_limit_relation_new(struct ArrowArrayStream* input, int64_t limit, const struct ArrowSchema* schema,
                               struct ArrowArrayStream* out, char* err, size_t errlen);

/* ------------------------------------------------------------------------------------------
 * Multi-GPU GROUP BY exchange (no reference equivalent: the reference is single-process).
 * One process per GPU.  After draining its local input, an aggregate stream exports its partial
 * groups bucketed by hash(key) % world into one contiguous device buffer per payload word; the
 * host plumbing (torch.distributed / RCCL all-to-all) moves the buckets; the receiving side
 * imports them and merges.  Then get_next() on the stream emits the groups this rank owns.
 * ---------------------------------------------------------------------------------------- */
/* Drain the input and build the local partial table.  Fills n_words = key words + accumulator
 * words per group, and counts[world] = groups destined to each rank. */
int32_t dfx_aggregate_partial_build(struct ArrowArrayStream* agg, int32_t world, int32_t* n_words,
                                    int64_t* counts, char* err, size_t errlen);
/* Write the bucketed partials: `dst` is a device buffer of n_words * total int64 words laid out
 * word-major within each destination bucket: for rank r, bucket base = n_words * prefix(r), and
 * word w of group g of that bucket at base + w * counts[r] + g. */
int32_t dfx_aggregate_partial_export(struct ArrowArrayStream* agg, void* dst_device, int64_t dst_words,
                                     char* err, size_t errlen);
/* Replace the stream's state by the merge of `n_buckets` received buckets (same layout as
 * export: bucket b holds counts[b] groups starting at word offset n_words * prefix(b)). */
int32_t dfx_aggregate_partial_import(struct ArrowArrayStream* agg, const void* src_device,
                                     const int64_t* counts, int32_t n_buckets, char* err,
                                     size_t errlen);

/* The same exchange as ONE library call over RCCL (counts, then variable-length buckets, as
 * grouped ncclSend/ncclRecv on the library's stream; xGMI between the GPUs of a node).  One process per GPU:
 *   rank 0:      dfx_comm_unique_id(id)           -- ncclGetUniqueId; the host plumbing hands `id` to every rank
 *   every rank:  dfx_comm_init(id, world, rank)   -- ncclCommInitRank on the library's device (dfx_init)
 *   per query:   dfx_aggregate_exchange(agg, comm, stats)  then get_next() emits the groups this rank owns
 * dfx_aggregate_exchange drains the input, counts the groups per owner rank, and replaces the stream's table by the merge
 * of what it received in TWO collective rounds: one all-gather of world + 3 words per rank (its state and query shape, the
 * groups it can receive / send without allocating more, its group count per destination rank: every buffer of the second
 * round is allocated before the first), then the buckets, the last of them with a trailer word (the sender's state) --
 * one host read-back (the count matrix) plus the final synchronisation.  Only if some rank receives or holds more groups
 * than it announced do the ranks allocate again and agree in a round of their own.  Ungrouped aggregates are combined with an all-gather of the per-rank scalars (every rank then emits
 * the global row).  stats (may be NULL): [0] groups sent, [1] groups received, [2] bytes sent, [3] host synchronisations.
 * RCCL is bound at run time (dlopen librccl.so.1); without it these calls return ExecutionError.  Utf8 keys:
 * NotImplemented (dictionary ids are rank-local). */
#define DFX_COMM_ID_BYTES 128
typedef struct dfx_comm dfx_comm;
int32_t dfx_comm_unique_id(uint8_t* id /* [DFX_COMM_ID_BYTES] */, char* err, size_t errlen);
int32_t dfx_comm_init(const uint8_t* id, int32_t world, int32_t rank, dfx_comm** out, char* err, size_t errlen);
void dfx_comm_destroy(dfx_comm* comm);
/* The number of ranks the communicator ITSELF reports (ncclCommCount): what RCCL saw, not what the host asked for.
 * -1: not a communicator, or RCCL refused.  (bench.py puts it into the line's config.rccl_ranks.) */
int32_t dfx_comm_ranks(const dfx_comm* comm);
int32_t dfx_aggregate_exchange(struct ArrowArrayStream* agg, dfx_comm* comm, int64_t* stats, char* err, size_t errlen);

/* ------------------------------------------------------------------------------------------
 * Measurement hooks (used by bench.py; not part of the drop-in surface).
 * ---------------------------------------------------------------------------------------- */
/* The group hash of a single-word key (its 32 bits sit in the HIGH half: table slot = hash >> (64 - log2 capacity)) and,
 * for keys below 2^32, its inverse: the hash restricted to such keys is a bijection of the low word, which is what lets
 * the partitioned GROUP BY route 12-byte rows {hash image, operand} and turn claimed images back into keys.  Host code,
 * no GPU needed; used by the CPU tests. */
uint64_t dfx_debug_group_hash(uint64_t key);
uint32_t dfx_debug_unhash32(uint32_t image);
/* One `column <op> literal` term of a scan plan (csrc/dfx_device.hpp: DevScanPlan) evaluated on the host with the arithmetic
 * the kernels use: the range test on the order-preserving image of `value`.  dtype: the column's dfx_dtype (Int32, UInt32,
 * Float32, Int64, UInt64, Float64); op: dfx_operator 0..5 (Eq NotEq Lt LtEq Gt GtEq); literal / value: canonical 64-bit
 * forms (signed ints sign-extended, unsigned zero-extended, Float32 bits in the low word, Float64 bits); is_null: the value
 * is null (arrow 0.12's rule for None decides).  Returns 0 / 1, or -1 for a type the plans do not cover.  Host code, no
 * GPU needed: the CPU tests compare it with the comparison it restates over NaN, +-0.0, +-inf and the integer extremes. */
int32_t dfx_debug_plan_term(int32_t dtype, int32_t op, uint64_t literal, uint64_t value, int32_t is_null);
/* Pulls every batch of a library stream and drops it on the device: no host RecordBatch, no D2H copy (what a stacked
 * operator would see).  rows / batches (may be NULL): what came out. */
int32_t dfx_relation_drain_device(struct ArrowArrayStream* stream, int64_t* rows, int64_t* batches, char* err, size_t errlen);
/* Test hook: the Arrow bitmap (LSB first) a FilterRelation computed for its most recent input batch -- the BooleanArray of
 * the reference's predicate closure (filter.rs:53-66), which the operator itself never hands out.  out == NULL: start
 * keeping it (call before next()); else copies (rows + 7) / 8 bytes to `out` (host memory). */
int32_t dfx_filter_debug_mask(struct ArrowArrayStream* filter_stream, uint8_t* out, int64_t out_bytes, int64_t* rows, char* err, size_t errlen);
/* When enabled every tracked kernel launch is bracketed by HIP events on its launch stream. */
int32_t dfx_profile_enable(int32_t on);
int32_t dfx_profile_reset(void);
/* Number of distinct kernels recorded so far; then per index: name, launches, total ms,
 * algorithmic bytes the launches covered. */
int32_t dfx_profile_count(void);
int32_t dfx_profile_get(int32_t index, char* name, size_t namelen, int64_t* launches,
                        double* total_ms, double* algo_bytes);

/* EXPLAIN of an operator tree created by this library: one line per operator, children indented by two spaces --
 * what was fused (a Filter under an Aggregate), how large the fused program is and which kernel family will run it
 * (a compile-time shape signature, the run-time decoded shape family, the SSA interpreter).  Host state only: works
 * before the first get_next and without a GPU.  (The reference prints its logical plan upstream of this boundary,
 * context.rs:105; this is the physical counterpart.)  Writes at most buflen - 1 bytes + NUL; returns the full length,
 * or -1 if `stream` was not produced by this library. */
int64_t dfx_relation_explain(struct ArrowArrayStream* stream, char* buf, size_t buflen);

/* Tunables (bench/test only; process-wide, read when an operator is created or a batch is launched).
 * Returns DFX_GENERAL for an unknown key.  Keys:
 *   "agg.strategy"           0 auto (calibrated on the first rows of a stream), 1 global-atomic table only,
 *                            2 LDS front cache, 3 partitioned (route rows to table blocks, aggregate blocks in LDS)
 *   "agg.capacity_log2"      initial GROUP BY table slots (0: 2^21)
 *   "agg.lds_slots" / "agg.lds_copies"   LDS front cache geometry (-1 auto)
 *   "agg.fewgroup"           1: <= 8 groups run on register accumulators (default), 0: LDS front cache
 *   "agg.partition_mode"     pass 1 of the partitioned strategy: 2 lock-free LDS rings (default), 1 LDS counting sort,
 *                            0 direct routing; "agg.partition_block", "agg.partition_pad", "agg.partition_cap_rows"
 *   "agg.replay_in_place"    experimental, default 0
 *   "agg.dict_capacity_log2" initial slots of a Utf8 key dictionary (0: 2^16)
 *   "filter.single_pass"     FilterRelation: 1 one kernel per batch (predicate, bitmap, look-back, compaction; default), 0 mask ->
 *                            scan -> compaction;  "filter.dense": the single-pass kernel's flavour that keeps a tile in registers
 *                            (one read of the column however many rows pass): -1 once the stream has kept > 22 % of a batch
 *                            (default), 0 never, 1 whenever the shape allows (one Float64 predicate column)
 *   "scan.fast"              0: always the generic SSA interpreter instead of the shape-specialised kernels
 *   "scan.plan"              scan plans (run-time query shapes evaluated as data: range tests on value images, 4-byte columns
 *                            widened, nulls by arrow's comparison rule): 1 wherever the shape is covered and no compile-time
 *                            signature matches (default), 0 never (round 3's dispatch), 2 also instead of the signatures
 *   "host.stream"            how HOST Arrow batches reach HBM (also a per-operator option: the first operator above a host
 *                            source decides): 0 pageable copies in order on the library's stream, the producer's array
 *                            released after the stream has passed them (default: 0.85 of the link, the fastest form measured);
 *                            1 pinned staging ring filled by library threads, DMA on a copy stream, the array released when its
 *                            bytes have been copied out; 2 one batch ahead on a copy stream; 3 = 2 + large buffers page-locked
 *                            in place.  "host.stage_threads" (8), "host.stage_mb" (16: bytes per pinned slot),
 *                            "host.stage_slots" (6) size the ring of form 1
 *   "pool.trim"              (any value) return the cached device buffers to the driver */
int32_t dfx_set_option(const char* key, int64_t value);
/* Measurement counters: "h2d_bytes" (column bytes the uploaders copied host -> device), "h2d_staged_bytes" (of which through
 * the pinned staging ring), "csv_cells" (cells the CSV source converted) -- what projection push-down saves, "csv_tiles" / "csv_general_tiles"
 * (64-record tiles converted by the CSV source / those that took the per-lane walk instead of the wave-cooperative path).  -1: unknown name. */
int64_t dfx_counter_get(const char* name);
void dfx_counter_reset(void);

#ifdef __cplusplus
}
#endif
#endif /* DFX_H */
