// dfx_relation.hpp -- internal operator classes (device-side mirror of the reference's operator tree).
#pragma once
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "dfx_host.hpp"
#include "dfx_kernels.hpp"

struct dfx_comm;

namespace dfx {

// ---- stream adapters --------------------------------------------------------------------------
// Takes ownership of *input (moves the struct). If the stream was produced by this library the
// inner device relation is unwrapped, otherwise the host stream is wrapped in an uploader.
Status adopt_input_stream(struct ArrowArrayStream* input, std::unique_ptr<Relation>* out);
// Exposes a device relation as an Arrow C stream (downloads each batch to host Arrow memory).
void export_relation(std::unique_ptr<Relation> rel, struct ArrowArrayStream* out);
// The relation behind one of our exported streams (nullptr if foreign). Not owning.
Relation* peek_exported(struct ArrowArrayStream* s);

// ---- table scan (dfx_table.cpp) -----------------------------------------------------------------
struct TableData {
  SchemaInfo schema;
  int64_t num_rows = 0;
  std::vector<DeviceColumn> columns;  // whole-table columns
  mutable ScanMemo memo;              // Relation::scan_memo()
};
class TableScanRelation : public Relation {
 public:
  TableScanRelation(std::shared_ptr<const TableData> t, int64_t batch_rows, int64_t row_begin = 0, int64_t n_rows = -1);
  RelationKind kind() const override { return REL_TABLE_SCAN; }
  Status next(DeviceBatch* out, bool* has) override;
  const SchemaInfo& schema() const override { return table_->schema; }
  void explain(std::string* out, int depth) const override;
  ScanMemo* scan_memo() override { return begin_ == 0 ? &table_->memo : nullptr; }  // (the memo describes the table's FIRST rows: a scan that starts elsewhere calibrates for itself)
  void prefer_batch_rows(int64_t rows) override {
    if (!emitted_any_ && rows > batch_rows_) batch_rows_ = rows & ~(int64_t)63;
  }

 private:
  std::shared_ptr<const TableData> table_;
  int64_t batch_rows_;
  int64_t begin_ = 0, end_ = 0;  // the rows this scan hands out
  int64_t pos_ = 0;
  bool emitted_any_ = false;
};

// ---- option sets ---------------------------------------------------------------------------------
struct AggOptions {
  int strategy = 0;         // 0 auto, 1 global table only, 2 LDS front cache
  int capacity_log2 = 0;    // 0: default
  int lds_slots = -1;       // -1 auto
  int lds_copies = -1;      // -1 auto
  int fast = 1;             // 0: always run the generic interpreter (tests compare both paths)
  int plan = 1;             // scan plans (DevScanPlan: range tests on value images, 4-byte columns and validity bitmaps without the
                            // interpreter): 1 wherever the shape is covered and no compile-time signature matches, 0 never (round 3's
                            // dispatch: run-time decoded shapes / interpreter), 2 also INSTEAD of the compile-time signatures (A/B)
  int partition_mode = 2;   // pass 1 of the partitioned strategy: 2 lock-free LDS rings, 1 LDS counting sort, 0 direct routing
  int dict_capacity_log2 = 0;  // initial slots of a Utf8 key dictionary (0: 2^16); tests use tiny values to force growth
  int partition_cap_rows = 0;  // rows per (producer, partition) region; 0: sized from the batch
  int partition_pad = 0;       // bytes of padding between partitions in the routing scratch
  int partition_block = 1024;  // pass-1 workgroup size in mode 1 (512: two workgroups per CU)
  int partition_defer = 0;     // routing regions hold this many worst-case batches (1: pass 2 after every batch).  0 (default):
                               // as many batches as make up ~2^27 rows, at most 8 -- a pass 2 per 2^27 scanned rows is what
                               // measured best for selective scans whatever the batch size (2^26-row batches: 2 saves 3 %,
                               // 4 and more lose it again; 2^27-row batches: 1); scans that route most of their rows get 1
  int partition_split_rows = 1 << 26;  // a batch of a scan that routes most of its rows is routed in launches of at most this
                               // many rows (0: never split): the regions of a 2^27-row launch cost pass 1 +20 %
  int partition_defer_batches = 8;  // at most this many pass-1 launches share one pass 2
  int export_kernel_copy = 1;  // large result columns reach the host by a copy kernel writing pinned memory (0: hipMemcpyAsync / copy engines)
  int ctrl_snapshot = 1;       // partitioned strategy: 1 the batch's last kernel writes the control-block snapshot to pinned host
                               // memory itself, 0 asynchronous copy on the side stream (round 1)
  int narrow_chunk16 = 1;      // narrow rows are written in 16-row chunks (sector-aligned) when no hot-key pairs need the LDS
  int shared_operand = 1;      // 2..3 aggregates of one null-free operand over narrow keys: 12-byte routed rows {image, raw operand}
  int narrow_keys = -1;        // 12-byte routed rows when the calibration slice saw only keys below 2^32: -1 auto, 0 never
  int partition_layout = 1;    // routing scratch: 1 producer-major, 0 partition-major (round 1), 2 windowed (DevPartition::win_stride;
                               // measured no better for the filtered query and 17 % worse in pass 1 when every row is routed)
  int partition_producers = 0; // pass-1 workgroups of the ring flavour (0: one per CU)
  int hot_keys = -1;           // pass 1 hot-key pairs in LDS: -1 when the calibration slice saw skew, 0 never, 1 always
  int early_keys = 1;          // 1: once the group count has stopped changing between two batches, the key column is compacted and copied
                               // to the host on the side stream while the scan goes on; emit hands that copy out if no group was added since
  int emit_async = 1;          // 1: emit queues its compaction kernels with the host's group count and checks the table's afterwards
  int calibration_memo = 1;    // 1: a resident table remembers the outcome of an aggregate's calibration slice per program shape
  int pass2_stream = 1;        // pass 2 of one-aggregate queries: region-streaming kernel (0: the flattened-index kernel)
  int fewgroup = 1;            // <= 8 groups after calibration: register accumulators (dfx_k_fewgroup.hip); 0: LDS front cache
  int pass1_ws = 8;            // pass 1 of selective scans over narrow keys: non-zero = the wave-specialised kernel (8 scanner +
                               // 8 router waves: the split DESIGN.md section 4 measured best of 6 / 8 / 10 / 12 / 14); 0: the ring
                               // kernel, every wave scans and routes
  int pass1_ws_dense = 0;      // the wave-specialised pass 1 when more than half of the rows are routed: 0 / 1 yes (default), -1 no (round 4's
                               // rule: the symmetric ring kernel)
  int pass1_ws_dense_scanners = 4;  // ... its split once more than two thirds are routed: 4 scanner + 12 router waves (8: always 8 + 8)
  int merge_scan_batches = 1;  // an aggregate over a scan of a resident table asks for slices of >= 2^27 rows (one per routing window)
                               // whatever batch width the scan was created with; 0: the caller's batch width is kept
  int host_stream = 0;         // host Arrow batches -> HBM (HostStreamOptions::mode): 0 in-order pageable copies (default: measured fastest),
                               // 1 pinned staging ring filled by library threads, 2 one batch ahead on a copy stream, 3 = 2 + page-locking in place
  int host_stage_threads = 8;  // ... threads that fill the ring
  int host_stage_mb = 16;      // ... bytes per slot
  int host_stage_slots = 6;    // ... slots
  int chunk_hold = 4;          // several chunks of accumulators over one table: batches held so that every chunk scans them in a row (one host check per
                               // chunk and hold; 1: per chunk and batch, rounds 3-5)
  int shared_planes = 1;       // two and more aggregates of ONE operand (AVG = SUM + COUNT, SUM + MIN + MAX of a column ...), narrow key, many groups, keys
                               // not skewed: the raw operand
                               // goes through the one-value pass 1 (whole-line chunks, 8192-slot table blocks) and pass 2 runs once per accumulator
                               // plane (PTF_PLANES); 0: rounds 3-6 -- 4096-slot blocks that hold every plane, 8-row chunks
  int pair_scan = 1;           // aggregates over TWO plain columns (two or more accumulators: SUM(v), MIN(w); AVG(v), MAX(w) ...), one narrow key, many
                               // groups, keys not skewed: ONE scan routes both operands (20-byte rows, PTF_PAIR) and pass 2 runs once per
                               // accumulator plane, instead of a scan per aggregate (0: always the scans)
  int split_aggregates = 1;    // one key, several aggregates of different operands, many groups: a scan per aggregate through the one-value
                               // kernels of the partitioned strategy (0: one scan that routes a row with every operand)
  int filter_dense = -1;       // single-pass FilterRelation, tiles kept in registers (k_filter_fused_dense): -1 when the stream has
                               // been keeping more than a wave can park in LDS (> 22 % of its rows), 0 never, 1 whenever the shape allows
  int filter_single_pass = 1;  // FilterRelation: predicate + bitmap + tile offsets (decoupled look-back) + compaction of the predicate's own
                               // columns in ONE kernel (0: k_predicate_mask -> scan -> k_compact, the column is read twice)
  int csv_wave_tiles = 1;      // CsvDataSource cells: one wave per tile of 64 records converts from an LDS copy of the tile's text (0: every
                               // lane walks its own record in global memory -- the path tiles with quotes or ragged records take anyway)
  int replay_in_place = 1;     // 1: rows spilled by a table that is NOT full (region overflow of a heavy key) are replayed into the
                               // table as it is, and only what it cannot take makes it grow (0: every spill quadruples the table)
};
AggOptions& agg_options();  // the process-wide defaults (dfx_set_option)
inline HostStreamOptions host_stream_options_of(const AggOptions& o) {
  HostStreamOptions h;
  h.mode = o.host_stream;
  h.threads = o.host_stage_threads;
  h.piece_mb = o.host_stage_mb;
  h.slots = o.host_stage_slots;
  return h;
}
// one key of dfx_set_option applied to an option set; false: unknown key
bool set_option_in(AggOptions& o, const char* key, int64_t value);
typedef std::vector<std::pair<std::string, int64_t>> OptionOverrides;
// An operator's own option set: the process defaults as they are when the operator first asks, with its overrides on top;
// frozen from then on (later dfx_set_option calls do not reach a running operator).
struct OperatorOptions {
  OptionOverrides overrides;
  const AggOptions& get() {
    if (!frozen_) {
      opt_ = agg_options();
      for (const auto& kv : overrides) (void)set_option_in(opt_, kv.first.c_str(), kv.second);
      frozen_ = true;
    }
    return opt_;
  }

 private:
  AggOptions opt_;
  bool frozen_ = false;
};

// ---- FilterRelation (src/execution/filter.rs) ---------------------------------------------------
class FilterRelation : public Relation {
 public:
  FilterRelation(std::unique_ptr<Relation> input, const dfx_runtime_expr& expr, SchemaInfo schema, OptionOverrides options = OptionOverrides());
  RelationKind kind() const override { return REL_FILTER; }
  Status next(DeviceBatch* out, bool* has) override;
  const SchemaInfo& schema() const override { return schema_; }
  void require_columns(const std::vector<char>& needed) override;
  void host_stream_options(const HostStreamOptions& o) override { if (input_) input_->host_stream_options(o); }
  void explain(std::string* out, int depth) const override;
  // for Filter -> Aggregate fusion
  std::unique_ptr<Relation> release_input() { return std::move(input_); }
  const dfx_runtime_expr& predicate() const { return expr_; }
  Relation* input() { return input_.get(); }
  bool single_program() const { return more_.empty(); }  // false: the predicate is evaluated as several conjuncts
  // test hook (dfx_filter_debug_mask): keep the bitmap of the most recent input batch
  void keep_mask(bool on) { keep_mask_ = on; }
  const std::shared_ptr<void>& last_mask() const { return last_mask_; }
  int64_t last_mask_rows() const { return last_mask_rows_; }

 private:
  std::unique_ptr<Relation> input_;
  dfx_runtime_expr expr_;
  SchemaInfo schema_;
  std::unique_ptr<ProgramBuilder> builder_;
  uint8_t pred_operand_ = kNoOperand;
  DevFastPlan fast_;
  Status deferred_;  // evaluation-time type errors of the reference surface on next()
  std::shared_ptr<void> ctrl_;
  mutable OperatorOptions opt_;
  double sel_seen_ = -1.0;    // largest kept / rows of a batch so far (-1: none yet): sizes the single-pass kernel's output buffers
  bool source_told_ = false;  // this operator's option set has reached the host source below (before the first batch is pulled)
  std::shared_ptr<void> ctrl_host_;  // pinned copy of the control block (kept count + error bits of a batch)
  std::vector<char> out_needed_;  // empty: every column is compacted
  bool keep_mask_ = false;
  std::shared_ptr<void> last_mask_;
  int64_t last_mask_rows_ = 0;
  // A conjunction that exceeds the limits of ONE fused program (kMaxCols columns, kMaxRegs computed values, kMaxImm
  // literals -- the reference has none, expression.rs:171-243 builds closures of any size): its top-level AND chain is
  // packed greedily into several programs; builder_ / pred_operand_ / fast_ are the first, these the others, and the
  // masks are ANDed (kept(A AND B) == kept(A) && kept(B): a null conjunct keeps nothing on either side)
  struct Part {
    std::unique_ptr<ProgramBuilder> builder;
    uint8_t operand = kNoOperand;
    DevFastPlan fast;
  };
  std::vector<Part> more_;
  Status build_parts();
};

// ---- ProjectRelation (src/execution/projection.rs) ----------------------------------------------
class ProjectRelation : public Relation {
 public:
  ProjectRelation(std::unique_ptr<Relation> input, std::vector<dfx_runtime_expr> exprs, SchemaInfo schema);
  RelationKind kind() const override { return REL_PROJECT; }
  Status next(DeviceBatch* out, bool* has) override;
  const SchemaInfo& schema() const override { return schema_; }
  void host_stream_options(const HostStreamOptions& o) override { if (input_) input_->host_stream_options(o); }
  void explain(std::string* out, int depth) const override;

 private:
  std::unique_ptr<Relation> input_;
  std::vector<dfx_runtime_expr> exprs_;
  SchemaInfo schema_;
  // computed outputs are packed greedily into fused programs (each within the device program
  // limits: kMaxRegs computed values, kMaxCols columns, kMaxOut outputs)
  struct Group {
    std::unique_ptr<ProgramBuilder> builder;
    std::vector<size_t> outputs;  // indices into exprs_
  };
  std::vector<Group> groups_;
  std::vector<int> passthrough_;      // >= 0: plain column reference (Arc clone in the reference)
  std::vector<uint8_t> operands_;     // computed outputs
  std::vector<int> out_dtype_;
  Status deferred_;
  std::shared_ptr<void> ctrl_;
};

// ---- AggregateRelation (src/execution/aggregate.rs) ---------------------------------------------
class AggregateRelation : public Relation {
 public:
  AggregateRelation(SchemaInfo schema, std::unique_ptr<Relation> input, std::vector<dfx_runtime_expr> group,
                    std::vector<dfx_runtime_expr> aggr, OptionOverrides options = OptionOverrides());
  ~AggregateRelation() override;
  RelationKind kind() const override { return REL_AGGREGATE; }
  Status next(DeviceBatch* out, bool* has) override;
  const SchemaInfo& schema() const override { return schema_; }
  void explain(std::string* out, int depth) const override;

  // multi-GPU partial exchange (include/dfx.h: dfx_aggregate_partial_*)
  Status partial_build(int world, int* n_words, int64_t* counts);
  Status partial_export(void* dst_device, int64_t dst_words);
  Status partial_import(const void* src_device, const int64_t* counts, int n_buckets);
  // the whole exchange inside the library over RCCL (dfx_exchange.cpp; include/dfx.h: dfx_aggregate_exchange)
  Status exchange(struct ::dfx_comm* comm, int64_t* stats);
  // pieces of the exchange that need the operator's state
  bool is_ungrouped() const;
  Status ungrouped_state_begin();                    // drains the input
  int ungrouped_state_words() const;                 // 2 words per accumulator slot: has-value flag, value bits
  const void* ungrouped_state_device() const;
  Status ungrouped_state_merge(const uint64_t* all_states, int world, int rank);  // fold the ranks' states, in rank order
  Status partial_count_device(int world, int* n_words, uint64_t** d_counts, std::shared_ptr<void>* owner);  // counts stay on the device
  Status partial_export_with(const std::vector<int64_t>& counts, void* dst_device, int64_t dst_words, bool sync, bool all_planes = false);
  // the pieces of the in-library exchange (dfx_exchange.cpp).  Accumulators beyond kMaxAggs live in several chunks of planes
  // over the same keys: groups are counted once, every chunk is exported / sent / merged with its own planes.  Utf8 keys:
  // the rank-local dictionary ids are turned into ids of a dictionary every rank builds identically (all ranks' strings
  // in rank order), the key planes are rewritten, and the groups travel like integer keys.
  int exchange_chunks() const;
  int exchange_chunk_words(int c) const;           // key words + accumulators of chunk c
  Status exchange_drain();                          // drains the input (grouped), no device work otherwise
  uint64_t exchange_group_bound() const;            // after the drain: an upper bound of the groups the count kernel will find
  Status exchange_count(int world, uint64_t* d_counts);  // groups per destination rank into d_counts[0, world) (zeroed here)
  Status exchange_export_chunk(int c, const std::vector<int64_t>& counts, void* dst_device, int64_t dst_words);
  Status exchange_import_begin(uint64_t total_groups);
  Status exchange_import_chunk(int c, const void* src_device, const int64_t* counts, int n_buckets);
  Status exchange_import_finish();
  int exchange_dicts() const;                       // Utf8 GROUP BY keys
  Status exchange_dict_local(int d, std::vector<uint32_t>* lens, std::vector<uint8_t>* pool);  // strings by local id
  Status exchange_dict_globalise(int d, const std::vector<uint32_t>& lens, const std::vector<uint8_t>& pool, const std::vector<uint64_t>& remap);
  Status ungrouped_select_chunk(int c);

  struct Impl;

 private:
  SchemaInfo schema_;
  std::unique_ptr<Impl> impl_;
};

// shared by filter / aggregate: CTRL_ERROR bits -> the reference's error
Status error_from_ctrl(uint32_t bits);
void set_exchange_test_failure(int64_t v);  // dfx_exchange.cpp: dfx_set_option("test.exchange_fail", rank << 8 | stage)

}  // namespace dfx
