// dfx_aggregate.cpp -- AggregateRelation (src/execution/aggregate.rs) on the device.
//
//   without_group_by (aggregate.rs:703-785): per input batch ONE fused kernel (predicate +
//     argument expressions + per-aggregate reduction, K5) produces the batch scalars, a one-thread
//     fold kernel applies AccumulatorSet::accumulate_scalar; one row comes back at end of input.
//   with_group_by (aggregate.rs:787-952): per input batch ONE fused kernel (K7 = predicate + key
//     and argument expressions + hash aggregation into an HBM-resident open-addressing table with
//     an LDS front cache); the table grows by rehash when it passes its load limit (rows that do
//     not fit are spilled and replayed); K8 emits dense arrays at end of input.
//   A FilterRelation feeding the aggregate (context.rs:126-139,162-192) is absorbed: its predicate
//   becomes part of the fused program and no filtered batch is ever materialised.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "dfx_relation.hpp"
#include "dfx_sigs.hpp"

namespace dfx {

AggOptions& agg_options() {
  static AggOptions o;
  return o;
}

namespace {
int ceil_log2(uint64_t v) {
  int l = 0;
  while ((1ull << l) < v && l < 62) ++l;
  return l;
}
}  // namespace

struct AggregateRelation::Impl {
  std::unique_ptr<Relation> input;
  bool has_pred = false;
  dfx_runtime_expr pred;
  std::vector<dfx_runtime_expr> group, aggr;
  // Utf8 GROUP BY keys are dictionary-encoded on the device into UInt64 ids that live in extra ("virtual")
  // columns appended to every input batch; the fused program sees an ordinary integer key (dfx_k_dict.hip)
  struct DictKey {
    int key = 0;       // index among the GROUP BY expressions
    int src_col = 0;   // the Utf8 column of the input schema
    int virt_col = 0;  // its UInt64 id column in `bind_schema`
    DevDict D;
    std::shared_ptr<void> state, hash, sid, str_off, str_len, pool, cursors;
    uint64_t ids_used = 0, pool_used = 0;  // as of the last completed batch
    bool allocated = false;
  };
  std::vector<DictKey> dicts;
  SchemaInfo bind_schema;                  // input schema + the virtual id columns (what the program binds to)
  std::vector<dfx_runtime_expr> group_rw;  // GROUP BY expressions with Utf8 columns redirected to their id columns
  std::vector<int> key_out_dtype;          // result type of each key column (DFX_UTF8 for dictionary keys)
  // result aggregates -> accumulators: AVG(x) is the pair SUM(x), COUNT(x) of consecutive accumulators, divided at
  // emit time (deviation D7).  `aggr` holds the ACCUMULATOR expressions (AVG already expanded), `outs` the results.
  struct OutAgg {
    int acc = 0;
    bool avg = false;
    int dtype = 0;
    std::string name;
  };
  std::vector<OutAgg> outs;
  // More accumulators than one fused program takes (the reference has no limit: create_accumulators builds any number,
  // aggregate.rs:319-342; real TPC-H Q1 needs 11) are split into CHUNKS -- <= kMaxAggs accumulators whose arguments fit the
  // program's limits on columns / computed values / literals: one fused program per chunk
  // (predicate + keys + that chunk's arguments), all chunks updating their own accumulator planes of the SAME table --
  // the second chunk's kernels find the key the first one inserted.  `builder / plan / fast / na / acc_kind ...` and the
  // table view T always describe the ACTIVE chunk; the others rest in `chunks`.  One chunk (the common case) never
  // touches any of this.
  static constexpr int kMaxAccsTotal = 32;
  struct Chunk {
    int a0 = 0, n = 0;  // accumulators [a0, a0 + n)
    std::unique_ptr<ProgramBuilder> builder, builder_np;
    DevAggPlan plan, plan_np;
    DevFastPlan fast, fast_np;
    std::shared_ptr<void> partial, state, dev_arg_dtype, dev_func;  // ungrouped state of this chunk
  };
  std::vector<Chunk> chunks;
  int cur_chunk = 0;
  // ONE key, several aggregates of DIFFERENT operands (SUM(v), MIN(w) ...): under the partitioned strategy a scan per aggregate --
  // each through the one-value kernels (12-byte routed rows, 256 partitions, the wave-specialised pass 1, the lean pass 2) -- beats
  // one scan that routes a row per key with every operand (24-byte rows: 512 partitions, 4-row chunks: pass 1 alone 1.42 ms per 2^27
  // rows against 2 x 0.43).  `single_chunks` holds that chunking, built at set-up; it replaces `chunks` when the strategy decision
  // (calibration slice or the resident table's memo) says "partitioned" -- few groups keep the one scan for all aggregates.
  std::vector<Chunk> single_chunks;
  // Round 6, late: the PAIR scan.  Two aggregates, narrow keys, a program the scan plan binds with three columns: the all-aggregates
  // program keeps running -- ONE scan routes {operand 0, image, operand 1} (20-byte rows, six per 128-byte line: PTF_PAIR) and pass 2
  // runs once per accumulator plane over the same regions, each launch the one-value kernel with its 96 KB block.  16 + 24 bytes read
  // per row become 24.  `single_chunks` stays in reserve: the stream falls back to it at a batch boundary when the pair kernels no
  // longer apply (the table outgrew 256 partitions, a batch the plan cannot bind).
  // The same host logic serves 2..3 aggregates of ONE operand (split_is_shared; PTF_PLANES, agg.shared_planes): the raw operand goes
  // through the one-value pass 1 exactly as the headline's does, pass 2 runs once per accumulator plane with that aggregate's
  // transform.  Rounds 3-6 gave such queries 4096-slot blocks holding every plane (twice the partitions, 8-row chunks of 96 bytes).
  bool pair_mode = false;
  bool pair_wide_seen = false;   // a key without a 32-bit image turned up under pair_mode: its rows take the spill list until the next batch
                                 // boundary, where the stream leaves for the scans per aggregate (they have a wide routed form)
  bool pair_is_planes = false;   // pair_mode: the shared-operand flavour
  bool split_is_shared = false;  // the aggregates single_chunks splits all take the same operand
  // ... or exactly TWO different operands between them (SUM(v), COUNT(v), MAX(w) ...): split_ops bit a = the operand (0 / 1) of
  // accumulator a, split_arg1 = the first accumulator of operand 1.  Two aggregates: the pair scan as described; three and more: the
  // operands travel RAW in the pair row (null-free batches only) and every accumulator gets its own pass 2 with its transform
  int split_distinct = 0;        // distinct operands among the aggregates (3: more than two)
  uint32_t split_ops = 0, split_arg1 = 0;
  // (four and more aggregates of one operand never had the all-planes block -- shared_operand() stops at three --: they keep the scans per
  // aggregate when the planes are switched off)
  bool split_applies() const { return split_ready && opt().split_aggregates && (!split_is_shared || opt().shared_planes || na_total > 3); }
  bool same_operand_all() const {  // shared_operand() without its limit of three
    if (kw != 1 || na < 2 || !opt().shared_operand) return false;
    for (int a = 1; a < na; ++a)
      if (plan.arg[a] != plan.arg[0]) return false;
    return true;
  }
  bool pair_batch_ok(const DeviceBatch& b);
  Status pair_fall_back();
  bool split_ready = false;     // single_chunks is built (used if agg.split_aggregates allows it when the operator runs)
  bool split_done = false;      // ... and installed
  bool split_decided = false;   // the strategy decision has been taken (whichever way)
  bool stop_after_decision = false;  // consume_batch_chunk returns as soon as the strategy is decided (rows before `decided_rows` are done)
  int64_t decided_rows = 0;
  void install_chunks(std::vector<Chunk>&& next);
  int na_total = 0;
  uint8_t acc_kind_all[kMaxAccsTotal], val_xform_all[kMaxAccsTotal];
  uint64_t acc_init_all[kMaxAccsTotal];
  uint64_t* accs_full = nullptr;  // plane 0 of the table's accumulators (T.accs is the active chunk's first plane)
  DevTable import_T;              // multi-GPU exchange: the table the received group partials are merged into
  uint64_t* import_accs_full = nullptr;
  std::vector<std::shared_ptr<void>> import_owners, import_keep;
  void activate(int c);
  DevTable view_of(const DevTable& any_view, uint64_t* full_accs, int c) const;
  DevTable full_view(const DevTable& any_view, uint64_t* full_accs) const;
  Status partial_view_check() const;
  Status build_chunk_programs(Chunk& ch);
  std::unique_ptr<ProgramBuilder> builder;
  DevAggPlan plan;
  DevFastPlan fast;
  // The same program WITHOUT the absorbed predicate, for batches whose referenced columns carry nulls: the reference's
  // FilterRelation emits all-valid arrays (fn filter ignores value nulls, filter.rs:83-92), so an aggregate over a Filter
  // sees every surviving slot as valid -- COUNT counts them, SUM adds whatever the slot holds.  A fused evaluation would
  // apply the ORIGINAL validity to the aggregate arguments; such batches are therefore filtered for real
  // (FilterRelation's kernels) and then aggregated without a predicate.  Null-free batches stay fused.
  std::unique_ptr<ProgramBuilder> builder_np;
  DevAggPlan plan_np;
  DevFastPlan fast_np;
  bool unfused_now = false;
  bool plan_required = false;  // the batch in hand has nulls under the fused predicate and was left fused for a scan plan
  Status deferred;
  bool done = false;
  bool built = false;
  int kw = 0, na = 0;
  int kw_out = 0;  // GROUP BY expressions of the query = key columns of the result (kw: key WORDS the kernels see -- five to
                   // seven keys are padded to eight with constant zero words, the table kernels being built for 1, 2, 3, 4, 8)
  std::vector<int> key_dtype, arg_dtype, out_dtype, func;
  uint8_t acc_kind[kMaxAggs], val_xform[kMaxAggs];
  uint64_t acc_init[kMaxAggs];
  // grouped state
  DevTable T;
  std::vector<std::shared_ptr<void>> table_owners;
  std::shared_ptr<void> ctrl;
  std::shared_ptr<void> stats;  // DevTable::stats
  DevRows spill;
  std::shared_ptr<void> spill_owner;
  bool lds_enabled = true;
  bool lds_calibrated = false;
  bool calibrating = false;     // the launch in progress is the calibration slice
  bool use_partition = false;   // strategy 3: route rows to table blocks, aggregate blocks in LDS
  bool narrow = false;          // every key the calibration slice saw is below 2^32: 12-byte routed rows (PTF_NARROW)
  int64_t launch_rows_hint = 0;  // > 0: the current batch is routed in launches of at most this many rows
  bool dense_seen = false;      // more than half of the calibration slice's rows passed the predicate: pass 2 after every batch
  bool mostly_seen = false;     // ... more than two thirds: pass 1's wave-specialised kernel runs 4 scanner + 12 router waves instead of 8 + 8
  bool skew_seen = false;       // the calibration slice's front cache absorbed a sizeable share of its rows: heavy keys
  DevPartition PT;
  std::shared_ptr<void> pt_rows, pt_counts;
  size_t pt_rows_bytes = 0, pt_cnt_bytes = 0;
  // Pass 2 is DEFERRED: pass 1 of several batches appends to the same routing regions (their fill counters live in
  // PT.counts between launches) and one pass 2 aggregates them all -- its table-block load/store, its launch and its
  // short-region tails are paid once per window instead of once per batch.  The window closes when the regions could
  // overflow: pt_fill_bound is an upper bound of the largest region fill, from the control-block snapshots
  // (CTRL_MAX_FILL, one batch behind) plus pt_worst rows for every batch launched since.
  bool pt_layout_valid = false;
  int64_t pt_layout_rows = 0;     // batch length the region layout was sized for
  uint32_t pt_worst = 0;          // rows one batch of that length adds to a region in the expected worst case (2 x average + 64)
  int pt_pending = 0;             // pass-1 launches waiting for their pass 2
  uint64_t pt_fill_bound = 0;
  int64_t pt_last_p2_seq = -1;    // batch_seq at the last pass-2 launch: older snapshots say nothing about the current fills
  int64_t pt_rows_in_flight = 0;  // input rows of the pending launches (all of them may still end up in the spill list)
  std::shared_ptr<void> snap_done;  // device word of DevPartition::snap_done
  bool snap_armed = false;          // the batch just launched writes its own control-block snapshot (no copy on the side stream)
  int64_t rows_seen = 0;
  // Several chunks of accumulators over ONE table (more than 8 aggregates, or one scan per aggregate): every chunk's scan of a batch
  // ends with a host check of the control block -- rows spilled under chunk c must be replayed while chunk c is active -- i.e. with
  // an idle device for a host round trip.  Round 6: up to chunk_hold batches are HELD and each chunk scans all of them in a row
  // (between batches of one chunk the checks run one batch behind, as in a single-chunk stream): one round trip per chunk and
  // hold, not per chunk and batch (two aggregates of different operands over 10^9 rows: 16 -> 4).
  std::vector<DeviceBatch> held;
  size_t held_bytes = 0;
  Status run_held();
  uint64_t occupied_known = 0;
  // control block checks run ONE BATCH BEHIND the launches: after batch i its control block is copied to
  // pinned memory asynchronously, batch i + 1 is launched, and only then is batch i's copy examined, so
  // the device never idles on a host round trip between batches
  std::shared_ptr<void> ctrl_host;          // pinned, 2 x CTRL_WORDS
  hipEvent_t ctrl_ev[2] = {nullptr, nullptr};
  hipEvent_t main_ev[2] = {nullptr, nullptr};  // "batch i launched" markers on the main stream
  bool ctrl_pending[2] = {false, false};
  int64_t ctrl_rows[2] = {0, 0};
  int64_t ctrl_seq[2] = {0, 0};             // batch_seq of the launch each snapshot follows
  int64_t batch_seq = 0;
  uint64_t unconfirmed_rows = 0;            // rows of launched batches whose control block is not examined yet
  // ungrouped state
  std::shared_ptr<void> partial, state, dev_arg_dtype, dev_func;
  // export
  std::vector<uint64_t> export_counts;
  mutable OperatorOptions options;  // this operator's option set (process defaults + its own overrides, frozen at first use)
  const AggOptions& opt() const { return options.get(); }

  Status setup(const SchemaInfo& input_schema);
  Status alloc_table(int cap_log2, DevTable* T, std::vector<std::shared_ptr<void>>* owners, bool new_ctrl, uint64_t** full_accs_out);
  Status ensure_spill(int64_t rows);
  Status ensure_partition(int64_t rows, bool nulls_now = false);
  bool shared_operand() const;
  Status flush_pass2();
  uint64_t program_fingerprint() const;
  Status grow_and_replay(uint64_t occupied, uint64_t spilled, uint64_t replay_from = 0);
  Status consume_batch(const DeviceBatch& b);
  Status consume_batch_chunk(const DeviceBatch& b);
  Status launch_rows(const DeviceBatch& b, const DevProgram& prog, const DevColumns& cols, int64_t row0, int64_t n);
  Status drain();
  Status emit_grouped(DeviceBatch* out, int64_t expected);
  std::shared_ptr<void> emit_total;  // pinned: the scan's group count
  // The key column ahead of time (agg.early_keys).  The result download is the one part of a query that cannot start before its
  // last kernel -- except for the keys: once every group exists, the key column is final.  When the group count has not changed
  // between two consecutive control-block snapshots, the compaction of the key plane and its copy to pinned memory are queued on
  // the side stream while the scan goes on.  At emit the copy is valid iff no group was added since (groups are never removed: the
  // count then differs) and the table was not replaced; it is attached to the key column and the exporter hands it out.
  // Round 6: NOBODY WAITS for the copy.  Round 2 found the DMA engine's device-to-host copies stalling for 6-150 ms once in a few
  // hundred calls (that is why the result itself is downloaded by a kernel), and emit used to sit in hipEventSynchronize behind
  // this one.  Now emit asks (hipEventQuery): a copy that has not finished is RETIRED -- its event, its buffers and the table
  // it reads (`keep`) move to a list that is emptied as the events complete -- and the step takes the path it would have taken
  // without the copy (+0.15 ms, not +100).
  struct EarlyKeys {
    bool armed = false;
    uint64_t occupied = 0;    // group count it was made for
    uint64_t generation = 0;  // table generation it was made from
    size_t bytes = 0;
    std::shared_ptr<void> host, total;          // pinned: the column, the compaction's own group count
    std::vector<std::shared_ptr<void>> scratch;  // device buffers the side stream is still using
    std::vector<std::shared_ptr<void>> keep;     // the table planes its kernels read (alive until they have run)
    hipEvent_t done = nullptr, start = nullptr;
    struct Retired {
      hipEvent_t done;
      std::vector<std::shared_ptr<void>> buffers;
    };
    std::vector<Retired> retired;
    bool ready() const { return !armed || !done || hipEventQuery(done) == hipSuccess; }
    void reap(bool block) {  // retired copies whose side-stream work has finished give their buffers back
      for (size_t i = 0; i < retired.size();) {
        if (block) (void)hipEventSynchronize(retired[i].done);
        if (block || hipEventQuery(retired[i].done) == hipSuccess) {
          (void)hipEventDestroy(retired[i].done);
          retired.erase(retired.begin() + (long)i);
        } else {
          ++i;
        }
      }
    }
    void drop() {  // forget the copy without waiting for it (before the table it reads is replaced, or when emit finds it unfinished)
      if (armed && done && hipEventQuery(done) != hipSuccess) {
        Retired r;
        r.done = done;
        done = nullptr;  // (a new event next time)
        r.buffers = std::move(scratch);
        r.buffers.insert(r.buffers.end(), keep.begin(), keep.end());
        r.buffers.push_back(host);
        r.buffers.push_back(total);  // (the pending copies write both)
        total.reset();
        retired.push_back(std::move(r));
      }
      armed = false;
      scratch.clear();
      keep.clear();
      host.reset();
      reap(false);
    }
    void cancel() { drop(); }
    ~EarlyKeys() {
      drop();
      reap(true);
      if (done) (void)hipEventDestroy(done);
      if (start) (void)hipEventDestroy(start);
    }
  } early;
  uint64_t table_generation = 0;
  uint64_t early_last_occupied = ~0ull;  // the group count of the previous snapshot
  Status early_keys_maybe();
  Status emit_ungrouped(DeviceBatch* out);
  Status read_ctrl(uint32_t* host_ctrl);
  Status post_ctrl(int64_t rows);
  Status alloc_ctrl_host();
  Status examine_ctrl(int slot);
  Status settle_ctrl();
  Status handle_ctrl(const uint32_t* hc, int64_t n);
  Status dict_alloc(DictKey& d, int slots_log2, uint64_t pool_cap, bool keep);
  Status dict_encode(DictKey& d, const DeviceColumn& src, int64_t n, DeviceColumn* ids_col);
  Status dict_emit(const DictKey& d, const uint64_t* ids, int64_t g, DeviceColumn* out);
  ~Impl() {
    if (ctrl_pending[0] || ctrl_pending[1]) (void)hipStreamSynchronize(ctx().aux);  // snapshots still in flight
    for (int i = 0; i < 2; ++i) {
      if (ctrl_ev[i]) (void)hipEventDestroy(ctrl_ev[i]);
      if (main_ev[i]) (void)hipEventDestroy(main_ev[i]);
    }
  }
};

// ---- setup ---------------------------------------------------------------------------------------
Status AggregateRelation::Impl::setup(const SchemaInfo& input_schema) {
  bind_schema = input_schema;
  group_rw = group;
  key_out_dtype.assign(group.size(), 0);
  for (size_t k = 0; k < group.size(); ++k) {  // GroupByScalar::Utf8 (aggregate.rs:838-846)
    if (group[k].is_aggregate || group[k].root < 0) continue;
    const dfx_expr_node& r = group[k].nodes[group[k].root];
    if (r.kind != DFX_EXPR_COLUMN || r.column < 0 || r.column >= (int)input_schema.fields.size()) continue;
    if (input_schema.fields[r.column].dtype != DFX_UTF8) continue;
    DictKey d;
    d.key = (int)k;
    d.src_col = r.column;
    d.virt_col = (int)bind_schema.fields.size();
    memset(&d.D, 0, sizeof(d.D));
    Field f;
    f.name = "__dict_ids_" + std::to_string(k);
    f.dtype = DFX_UINT64;
    f.nullable = false;
    bind_schema.fields.push_back(f);
    group_rw[k].nodes[group_rw[k].root].column = d.virt_col;
    group_rw[k].dtype = DFX_UINT64;
    key_out_dtype[k] = DFX_UTF8;
    dicts.push_back(std::move(d));
  }
  kw_out = (int)group.size();
  na_total = (int)aggr.size();
  if (kw_out > kMaxKeys) return Status::Err(DFX_NOT_IMPLEMENTED, strfmt("more than %d GROUP BY expressions", kMaxKeys));
  kw = kw_out;
  if (kw_out > 4) {  // (the reference builds a Vec<GroupByScalar> of any length, aggregate.rs:807-852)
    dfx_runtime_expr zero;
    dfx_expr_node n;
    memset(&n, 0, sizeof(n));
    n.kind = DFX_EXPR_LITERAL;
    n.dtype = DFX_INT64;
    n.left = n.right = n.column = -1;
    n.lit.i64 = 0;
    zero.nodes.push_back(n);
    zero.strings.emplace_back();
    zero.has_name.push_back(0);
    zero.root = 0;
    zero.dtype = DFX_INT64;
    zero.name = "0";
    while ((int)group_rw.size() < kMaxKeys) {
      group_rw.push_back(zero);
      key_out_dtype.push_back(DFX_INT64);
    }
    kw = kMaxKeys;
  }
  if (na_total > kMaxAccsTotal) return Status::Err(DFX_NOT_IMPLEMENTED, strfmt("more than %d accumulators", kMaxAccsTotal));
  key_dtype.assign(kw, 0);
  arg_dtype.assign(na_total, 0);
  out_dtype.assign(na_total, 0);
  func.assign(na_total, 0);
  chunks.clear();
  for (int a0 = 0; a0 == 0 || a0 < na_total;) {
    // as many of the next accumulators as ONE fused program takes: <= kMaxAggs, and within the program's limits on
    // distinct columns, computed values and literals (a chunk that does not fit is rebuilt one accumulator shorter)
    int n = std::min(kMaxAggs, na_total - a0);
    for (;;) {
      Chunk ch;
      ch.a0 = a0;
      ch.n = n;
      Status st = build_chunk_programs(ch);
      if (st.ok()) {
        chunks.push_back(std::move(ch));
        break;
      }
      if (!program_limit_error(st) || n <= 1) return st;  // (a genuinely unsupported aggregate is not rebuilt kMaxAggs times)
      --n;
    }
    a0 += std::max(n, 1);
  }
  // chunk 0 becomes the active one
  cur_chunk = 0;
  Chunk& c0 = chunks[0];
  builder = std::move(c0.builder);
  builder_np = std::move(c0.builder_np);
  plan = c0.plan;
  plan_np = c0.plan_np;
  fast = c0.fast;
  fast_np = c0.fast_np;
  na = c0.n;
  for (int a = 0; a < na; ++a) {
    acc_kind[a] = acc_kind_all[a];
    val_xform[a] = val_xform_all[a];
    acc_init[a] = acc_init_all[a];
  }
  // one key, two or more aggregates that do not all take the same operand: the per-aggregate chunking for the partitioned
  // strategy (see single_chunks).  Built now so that a shape the one-aggregate programs cannot take shows up here, not mid-stream.
  if (kw == 1 && kw_out == 1 && na_total >= 2 && chunks.size() == 1) {  // (agg.split_aggregates / agg.shared_planes are read when the operator runs: options freeze at first use)
    std::vector<Chunk> singles;
    bool ok = true;
    for (int a = 0; a < na_total && ok; ++a) {
      Chunk ch;
      ch.a0 = a;
      ch.n = 1;
      ok = build_chunk_programs(ch).ok();
      if (ok) singles.push_back(std::move(ch));
    }
    if (ok) {
      single_chunks = std::move(singles);
      split_ready = true;
      {  // (shared_operand() without the option: options are not frozen yet)
        split_is_shared = na >= 2;  // (any number of aggregates of one operand: a pass 2 per plane has no limit of three)
        for (int a = 1; a < na; ++a)
          if (plan.arg[a] != plan.arg[0]) split_is_shared = false;
        split_distinct = 1;
        split_ops = 0;
        for (int a = 1; a < na; ++a) {
          if (plan.arg[a] == plan.arg[0]) continue;
          if (split_distinct == 1) {
            split_distinct = 2;
            split_arg1 = (uint32_t)a;
          }
          if (plan.arg[a] == plan.arg[split_arg1]) split_ops |= 1u << a;
          else split_distinct = 3;
        }
      }
    }
  }
  return Status::OK();
}

// the fused programs of one chunk: predicate + keys + arguments [a0, a0 + n), and the predicate-free twin
Status AggregateRelation::Impl::build_chunk_programs(Chunk& ch) {
  ch.builder.reset(new ProgramBuilder(bind_schema));
  ProgramBuilder* builder = ch.builder.get();
  DevAggPlan& plan = ch.plan;
  DevFastPlan& fast = ch.fast;
  memset(&plan, 0, sizeof(plan));
  memset(&fast, 0, sizeof(fast));
  plan.pred = kNoOperand;
  if (has_pred) {
    int dt = 0;
    DFX_RETURN_IF_ERROR(builder->add(pred, pred.root, &plan.pred, &dt));
    if (dt != DFX_BOOLEAN) return Status::Err(DFX_EXECUTION_ERROR, "Filter expression did not evaluate to boolean");
  }
  for (int k = 0; k < kw; ++k) {
    if (k < kw_out && group[k].is_aggregate) return Status::Err(DFX_INTERNAL_ERROR, "explicit panic: get_func() on an aggregate expression");
    int dt = 0;
    DFX_RETURN_IF_ERROR(builder->add(group_rw[k], group_rw[k].root, &plan.key[k], &dt));
    if (!dtype_is_int(dt))  // aggregate.rs:848-850 (floats and booleans are rejected)
      return Status::Err(DFX_EXECUTION_ERROR, "Unsupported GROUP BY data type");
    key_dtype[k] = dt;
    if (!key_out_dtype[k]) key_out_dtype[k] = dt;
    plan.key_dtype[k] = (uint8_t)dt;
  }
  for (int a = ch.a0; a < ch.a0 + ch.n; ++a) {
    const int la = a - ch.a0;  // index inside the chunk
    const dfx_runtime_expr& e = aggr[a];
    if (!e.is_aggregate)  // create_accumulators (aggregate.rs:335-337)
      return Status::Err(DFX_EXECUTION_ERROR, "invalid aggregate expression");
    int dt = 0;
    DFX_RETURN_IF_ERROR(builder->add(e, e.agg_arg, &plan.arg[la], &dt));
    arg_dtype[a] = dt;
    plan.arg_dtype[la] = (uint8_t)dt;
    func[a] = e.agg_func;
    const int t = e.agg_type;
    uint8_t& acc_kind_a = acc_kind_all[a];
    uint8_t& val_xform_a = val_xform_all[a];
    uint64_t& acc_init_a = acc_init_all[a];
    if (e.agg_func == AGG_COUNT) {  // deviation D3 (reference: "unsupported aggregate function")
      out_dtype[a] = DFX_UINT64;
      acc_kind_a = ACC_ADD_U64;
      val_xform_a = VT_COUNT_VALID;
      acc_init_a = 0;
      continue;
    }
    if (!dtype_is_numeric(t)) {  // array_min/max/sum `_ =>` arms (aggregate.rs:406-408 ...)
      const char* fn = e.agg_func == AGG_MIN ? "MIN" : e.agg_func == AGG_MAX ? "MAX" : "SUM";
      return Status::Err(DFX_EXECUTION_ERROR, std::string("Unsupported data type for ") + fn);
    }
    if (dt != t)  // downcast_ref::<T>().unwrap() by the declared type (aggregate.rs:347, :563)
      return Status::Err(DFX_INTERNAL_ERROR, strfmt("called `Option::unwrap()` on a `None` value (aggregate argument is %s, declared %s)",
                                                    dtype_name(dt), dtype_name(t)));
    out_dtype[a] = t;
    const bool grouped = kw > 0;
    if (e.agg_func == AGG_SUM) {
      val_xform_a = VT_RAW;
      if (t == DFX_FLOAT64) {
        acc_kind_a = ACC_ADD_F64;
        // grouped: the first value initialises the accumulator => identity is -0.0 (x + -0.0 == x
        // bit for bit); ungrouped: array_ops::sum starts from 0.0 (aggregate.rs:480-546)
        acc_init_a = grouped ? 0x8000000000000000ull : 0ull;
      } else if (t == DFX_FLOAT32) {
        acc_kind_a = ACC_ADD_F32;
        acc_init_a = grouped ? 0x80000000ull : 0ull;
      } else {
        acc_kind_a = ACC_ADD_U64;
        acc_init_a = 0;
      }
    } else {
      const bool is_min = e.agg_func == AGG_MIN;
      if (t == DFX_FLOAT64 || t == DFX_FLOAT32) {
        val_xform_a = t == DFX_FLOAT64 ? (is_min ? VT_F64_ORD_MIN : VT_F64_ORD_MAX) : (is_min ? VT_F32_ORD_MIN : VT_F32_ORD_MAX);
        acc_kind_a = is_min ? ACC_MIN_U64 : ACC_MAX_U64;
        acc_init_a = is_min ? ~0ull : 0ull;
      } else if (dtype_is_signed(t)) {
        val_xform_a = VT_RAW;
        acc_kind_a = is_min ? ACC_MIN_S64 : ACC_MAX_S64;
        acc_init_a = is_min ? 0x7FFFFFFFFFFFFFFFull : 0x8000000000000000ull;
      } else {
        val_xform_a = VT_RAW;
        acc_kind_a = is_min ? ACC_MIN_U64 : ACC_MAX_U64;
        acc_init_a = is_min ? ~0ull : 0ull;
      }
    }
  }
  builder->build_fast(plan.pred, plan.key, kw, plan.arg, ch.n, &fast);
  if (has_pred) {  // predicate-free twin (operands are numbered differently: its own plan)
    ch.builder_np.reset(new ProgramBuilder(bind_schema));
    ch.plan_np = plan;
    ch.plan_np.pred = kNoOperand;
    int dt = 0;
    for (int k = 0; k < kw; ++k) DFX_RETURN_IF_ERROR(ch.builder_np->add(group_rw[k], group_rw[k].root, &ch.plan_np.key[k], &dt));
    for (int a = ch.a0; a < ch.a0 + ch.n; ++a) DFX_RETURN_IF_ERROR(ch.builder_np->add(aggr[a], aggr[a].agg_arg, &ch.plan_np.arg[a - ch.a0], &dt));
    ch.builder_np->build_fast(ch.plan_np.pred, ch.plan_np.key, kw, ch.plan_np.arg, ch.n, &ch.fast_np);
  }
  return Status::OK();
}

// Replace the chunking (grouped aggregates only): the active chunk's members go back to their chunk, `next` becomes the
// chunk list, its first chunk the active one.  The accumulator planes are per ACCUMULATOR, not per chunk: nothing moves.
void AggregateRelation::Impl::install_chunks(std::vector<Chunk>&& next) {
  auto swap_with = [&](Chunk& ch) {
    std::swap(builder, ch.builder);
    std::swap(builder_np, ch.builder_np);
    std::swap(plan, ch.plan);
    std::swap(plan_np, ch.plan_np);
    std::swap(fast, ch.fast);
    std::swap(fast_np, ch.fast_np);
  };
  swap_with(chunks[(size_t)cur_chunk]);
  chunks = std::move(next);
  cur_chunk = 0;
  swap_with(chunks[0]);
  const Chunk& ch = chunks[0];
  na = ch.n;
  for (int a = 0; a < na; ++a) {
    acc_kind[a] = acc_kind_all[ch.a0 + a];
    val_xform[a] = val_xform_all[ch.a0 + a];
    acc_init[a] = acc_init_all[ch.a0 + a];
  }
  if (kw > 0 && accs_full) T = view_of(T, accs_full, 0);
  pt_layout_valid = false;  // (routed rows change width)
}

// Make chunk c the active one: programs, accumulator algebra, ungrouped buffers and the table view.
void AggregateRelation::Impl::activate(int c) {
  if (c == cur_chunk) return;
  auto swap_with = [&](Chunk& ch) {
    std::swap(builder, ch.builder);
    std::swap(builder_np, ch.builder_np);
    std::swap(plan, ch.plan);
    std::swap(plan_np, ch.plan_np);
    std::swap(fast, ch.fast);
    std::swap(fast_np, ch.fast_np);
    std::swap(partial, ch.partial);
    std::swap(state, ch.state);
    std::swap(dev_arg_dtype, ch.dev_arg_dtype);
    std::swap(dev_func, ch.dev_func);
  };
  swap_with(chunks[(size_t)cur_chunk]);  // the active members go back to their chunk
  swap_with(chunks[(size_t)c]);          // chunk c's become active
  cur_chunk = c;
  const Chunk& ch = chunks[(size_t)c];
  na = ch.n;
  for (int a = 0; a < na; ++a) {
    acc_kind[a] = acc_kind_all[ch.a0 + a];
    val_xform[a] = val_xform_all[ch.a0 + a];
    acc_init[a] = acc_init_all[ch.a0 + a];
  }
  if (kw > 0 && accs_full) T = view_of(T, accs_full, c);
  else if (kw == 0) {
    T.na = na;
    for (int a = 0; a < na; ++a) {
      T.acc_kind[a] = acc_kind[a];
      T.val_xform[a] = val_xform[a];
      T.acc_init[a] = acc_init[a];
    }
  }
}

// chunk c's view of a table: same keys / control block, accumulator planes [a0, a0 + n)
DevTable AggregateRelation::Impl::view_of(const DevTable& any_view, uint64_t* full_accs, int c) const {
  DevTable v = any_view;
  const Chunk& ch = chunks[(size_t)c];
  v.accs = full_accs + (uint64_t)ch.a0 * v.stride;
  v.na = ch.n;
  for (int a = 0; a < kMaxAggs; ++a) {
    v.acc_kind[a] = a < ch.n ? acc_kind_all[ch.a0 + a] : 0;
    v.val_xform[a] = a < ch.n ? val_xform_all[ch.a0 + a] : 0;
    v.acc_init[a] = a < ch.n ? acc_init_all[ch.a0 + a] : 0;
  }
  return v;
}

// EVERY accumulator plane of a table as one view (the planes are contiguous by accumulator index, whatever the chunking):
// what the public partial_* entry points move.  The drain may have replaced the chunk list (one scan per aggregate,
// agg.split_aggregates): a chunk's view would then carry one plane of several and the others would be emitted with their
// init values (round-4 advisor finding).  Valid while all accumulators fit one kernel's view (<= kMaxAggs).
DevTable AggregateRelation::Impl::full_view(const DevTable& any_view, uint64_t* full_accs) const {
  DevTable v = any_view;
  v.accs = full_accs;
  v.na = na_total;
  for (int a = 0; a < kMaxAggs; ++a) {
    v.acc_kind[a] = a < na_total ? acc_kind_all[a] : 0;
    v.val_xform[a] = a < na_total ? val_xform_all[a] : 0;
    v.acc_init[a] = a < na_total ? acc_init_all[a] : 0;
  }
  return v;
}

// (after the drain: the chunk list is final)
Status AggregateRelation::Impl::partial_view_check() const {
  if (na_total > kMaxAggs) return Status::Err(DFX_NOT_IMPLEMENTED, strfmt("multi-GPU exchange of more than %d accumulators", kMaxAggs));
  return Status::OK();
}

// what the calibration slice's outcome depends on: the fused program (predicate, key and argument expressions with
// their literals), the input columns it binds and the plan's operands (FNV-1a over the bytes)
uint64_t AggregateRelation::Impl::program_fingerprint() const {
  uint64_t h = 0xCBF29CE484222325ull;
  auto mix = [&](const void* p, size_t n) {
    const uint8_t* b = (const uint8_t*)p;
    for (size_t i = 0; i < n; ++i) h = (h ^ b[i]) * 0x100000001B3ull;
  };
  const DevProgram& P = builder->program();
  mix(&P.n_ins, sizeof(P.n_ins));
  mix(&P.n_cols, sizeof(P.n_cols));
  mix(&P.n_imm, sizeof(P.n_imm));
  mix(P.ins, sizeof(DevIns) * (size_t)std::max(0, std::min<int>(P.n_ins, kMaxRegs)));
  mix(P.imm, sizeof(uint64_t) * (size_t)std::max(0, std::min<int>(P.n_imm, kMaxImm)));
  mix(P.col_dtype, sizeof(P.col_dtype));
  for (int ci : builder->columns()) mix(&ci, sizeof(ci));
  mix(&plan.pred, 1);
  mix(plan.key, sizeof(plan.key));
  mix(plan.arg, sizeof(plan.arg));
  mix(&kw, sizeof(kw));
  mix(&na, sizeof(na));
  return h;
}

// ---- table management ------------------------------------------------------------------------------
Status AggregateRelation::Impl::alloc_table(int cap_log2, DevTable* Tn, std::vector<std::shared_ptr<void>>* owners,
                                            bool new_ctrl, uint64_t** full_accs_out) {
  hipStream_t s = ctx().stream;
  memset(Tn, 0, sizeof(*Tn));
  const uint64_t cap = 1ull << cap_log2;
  Tn->stride = cap + 64;
  Tn->mask = cap - 1;
  Tn->shift = 64 - cap_log2;
  Tn->kw = kw;
  Tn->na = na;
  Tn->load_limit = cap / 2;
  Tn->max_probe = (int)std::min<uint64_t>(cap, 1u << 30);
  {  // probing block = what one workgroup can hold in 128 KB of LDS (keys + the accumulators of the widest chunk)
    int widest = 1;
    for (const Chunk& ch : chunks) widest = std::max(widest, ch.n);
    if (split_applies()) widest = 1;  // (the partitioned strategy will run one accumulator per scan: blocks of 8192 slots, 256 partitions)
    uint64_t blk = 16384 / (uint64_t)(std::max(kw, 1) + widest);  // 128 KB of LDS per block (pass 2: one workgroup per CU)
    uint64_t p2 = 64;
    while (p2 * 2 <= blk) p2 *= 2;
    if (p2 > cap) p2 = cap;
    Tn->block_mask = (uint32_t)(p2 - 1);
  }
  for (int a = 0; a < na; ++a) {
    Tn->acc_kind[a] = acc_kind[a];
    Tn->val_xform[a] = val_xform[a];
    Tn->acc_init[a] = acc_init[a];
  }
  Status st;
  auto keys = device_alloc(sizeof(uint64_t) * Tn->stride * (size_t)std::max(kw, 1), &st);
  if (!keys) return st;
  auto accs = device_alloc(sizeof(uint64_t) * Tn->stride * (size_t)std::max(na_total, 1), &st);  // every chunk's planes
  if (!accs) return st;
  Tn->keys = (uint64_t*)keys.get();
  Tn->accs = (uint64_t*)accs.get();
  owners->clear();
  owners->push_back(keys);
  owners->push_back(accs);
  if (kw > 1) {
    auto slot_state = device_alloc(sizeof(uint32_t) * Tn->stride, &st);
    if (!slot_state) return st;
    Tn->state = (uint32_t*)slot_state.get();
    owners->push_back(slot_state);
    DFX_HIP(hipMemsetAsync(Tn->state, 0, sizeof(uint32_t) * Tn->stride, s));
  } else {
    DFX_HIP(launch_fill_u64(Tn->keys, kEmptyKey, (int64_t)Tn->stride, s));
  }
  for (int a = 0; a < na_total; ++a) DFX_HIP(launch_fill_u64(Tn->accs + (size_t)a * Tn->stride, acc_init_all[a], (int64_t)Tn->stride, s));
  if (full_accs_out) *full_accs_out = Tn->accs;
  Tn->accs += (uint64_t)chunks[(size_t)cur_chunk].a0 * Tn->stride;  // the caller gets the ACTIVE chunk's view
  if (new_ctrl) {
    ctrl = device_alloc(sizeof(uint32_t) * CTRL_WORDS, &st);
    if (!ctrl) return st;
    DFX_HIP(hipMemsetAsync(ctrl.get(), 0, sizeof(uint32_t) * CTRL_WORDS, s));
    stats = device_alloc(sizeof(uint64_t) * kStatStripes * STAT_WORDS, &st);
    if (!stats) return st;
    DFX_HIP(hipMemsetAsync(stats.get(), 0, sizeof(uint64_t) * kStatStripes * STAT_WORDS, s));
  }
  Tn->ctrl = (uint32_t*)ctrl.get();
  Tn->stats = (uint64_t*)stats.get();
  return Status::OK();
}

Status AggregateRelation::Impl::ensure_spill(int64_t rows) {
  if (rows <= 0) {
    return Status::OK();
  }
  if (spill.words && spill.capacity >= (uint64_t)rows) return Status::OK();
  DFX_RETURN_IF_ERROR(settle_ctrl());  // rows spilled by batches still in flight live in the old list
  ScopedUs t_alloc(&counters().agg_alloc_us);
  Status st;
  int widest = na;  // the list is shared by every chunk of accumulators: planes for the widest one
  for (const Chunk& ch : chunks) widest = std::max(widest, ch.n);
  spill_owner = device_alloc(sizeof(uint64_t) * (size_t)rows * (size_t)(kw + widest), &st);
  if (!spill_owner) return st;
  spill.words = (uint64_t*)spill_owner.get();
  spill.capacity = (uint64_t)rows;
  return Status::OK();
}

// scratch for the partitioned strategy, sized for a batch of `rows` rows (worst case: all pass)
// the active chunk's 2..3 aggregates all take the same operand (AVG's SUM and COUNT, SUM + MIN + MAX of one column ...):
// with narrow keys and no nulls in this batch, routed rows carry that one operand (PTF_SHARED)
bool AggregateRelation::Impl::shared_operand() const {
  if (kw != 1 || na < 2 || na > 3 || !opt().shared_operand) return false;
  for (int a = 1; a < na; ++a)
    if (plan.arg[a] != plan.arg[0]) return false;
  return true;
}

Status AggregateRelation::Impl::ensure_partition(int64_t rows, bool nulls_now) {
  const uint64_t S = (uint64_t)T.block_mask + 1;
  const bool raw_ok = !nulls_now || (has_pred && !unfused_now);  // (a raw operand has no validity: fine under an absorbed predicate -- every surviving slot is valid)
  const bool want_planes = pair_mode && pair_is_planes && narrow && opt().narrow_keys != 0 && raw_ok && same_operand_all() && kNarrowLine && opt().narrow_chunk16 &&
                           opt().pass1_ws > 0 && opt().partition_layout != 2 && ((uint32_t)opt().partition_mode & 0x8Fu) == 2u &&
                           partition_ws_bytes((uint32_t)((T.mask + 1) / S), 4, 1) <= (size_t)158 * 1024;
  const bool want_shared = !pair_mode && narrow && opt().narrow_keys != 0 && !nulls_now && shared_operand() &&
                           ((uint32_t)opt().partition_mode & 0x8Fu) == 2u &&
                           partition_ring_bytes(2, (uint32_t)((T.mask + 1) / S), 16, false, true, 128) <= (size_t)158 * 1024;
  const bool want_pair = pair_mode && !pair_is_planes && !want_shared && narrow && kw == 1 && na >= 2 && split_distinct == 2 && (na == 2 || raw_ok) && kNarrowLine && opt().narrow_keys != 0 && opt().narrow_chunk16 &&
                         opt().pass1_ws > 0 && opt().partition_layout != 2 && ((uint32_t)opt().partition_mode & 0x8Fu) == 2u &&
                         partition_ws_bytes((uint32_t)((T.mask + 1) / S), 8, 2) <= (size_t)158 * 1024;
  if (pair_mode && !want_pair && !want_planes)  // (a table block holds ONE accumulator plane in this mode: no other routed form fits; until the
    return Status::Err(DFX_NOT_IMPLEMENTED, "pair scan: not for this table");  // next batch boundary the rows go through the global table)
  const bool want_narrow = narrow && kw == 1 && (na == 1 || want_shared || want_pair || want_planes) && opt().narrow_keys != 0;
  const uint32_t n_words = (want_shared || want_planes) ? 2u : (uint32_t)(kw + na);
  if (pt_layout_valid && rows <= pt_layout_rows && PT.n_parts == (uint32_t)((T.mask + 1) / S) && PT.n_words == n_words &&
      ((PT.flags & PTF_NARROW) != 0) == want_narrow && ((PT.flags & PTF_SHARED) != 0) == (want_shared || want_planes) && ((PT.flags & PTF_PAIR) != 0) == want_pair &&
      ((PT.flags & PTF_PLANES) != 0) == (want_planes || (want_pair && na > 2)))
    return Status::OK();  // same table, a batch the regions were sized for: keep appending
  DFX_RETURN_IF_ERROR(flush_pass2());  // rows routed under the old layout
  pt_layout_valid = false;
  memset(&PT, 0, sizeof(PT));
  PT.n_parts = (uint32_t)((T.mask + 1) / S);
  PT.n_words = n_words;
  int ps = 0;
  while ((1ull << ps) < S) ++ps;
  PT.part_shift = (uint32_t)ps;
  if (PT.n_parts > 4096) return Status::Err(DFX_NOT_IMPLEMENTED, "partitioned strategy: too many table blocks");
  // pass-1 flavour (agg.partition_mode).  Scattered 16-byte stores are transaction-bound at ~87 G rows/s on
  // MI355X while runs of >= 64 bytes reach > 400 G rows/s (tools/ubench2.hip), so routed rows are write-combined
  // in LDS whenever the partition count allows it:
  //   2 (default)  lock-free per-partition LDS rings, 128-byte chunks, no barrier in the scan loop
  //   1            workgroup-wide LDS counting sort (also for partition counts whose rings do not fit LDS)
  //   0            one 16-byte store per row straight from registers (very many partitions)
  const AggOptions& o = opt();
  const uint32_t block = o.partition_block == 512 ? 512u : 1024u;
  const size_t budget = block == 512 ? (size_t)79 * 1024 : (size_t)156 * 1024;
  const uint32_t sort_cap = partition_sort_capacity(PT.n_words, PT.n_parts, block, budget);
  const int want = o.partition_mode & 15;
  if (want_shared) {
    PT.flags |= PTF_NARROW | PTF_SHARED;
    PT.mode = 2u;
    PT.block = 1024;
    PT.stage_rows = 0;
    PT.n_producers = (uint32_t)std::min(1024, device_cu_count());
    if (o.partition_producers > 0) PT.n_producers = (uint32_t)std::min(1024, o.partition_producers);
  } else if (want_planes) {
    PT.flags |= PTF_NARROW | PTF_CHUNK16 | PTF_WS | PTF_SHARED | PTF_PLANES;
    PT.ws_scanners = (o.pass1_ws == 4 || (mostly_seen && o.pass1_ws_dense_scanners == 4)) ? 4u : 8u;  // (as for one aggregate)
    PT.mode = 2u;
    PT.block = 1024;
    PT.stage_rows = 0;
    PT.n_producers = (uint32_t)std::min(1024, device_cu_count());
    if (o.partition_producers > 0) PT.n_producers = (uint32_t)std::min(1024, o.partition_producers);
  } else if (want_pair) {
    PT.flags |= PTF_NARROW | PTF_CHUNK16 | PTF_WS | PTF_PAIR;
    if (na > 2) PT.flags |= PTF_PLANES;  // raw operands, a transform per accumulator in pass 2
    PT.pair_ops = split_ops;
    PT.pair_arg1 = split_arg1;
    PT.ws_scanners = 8u;
    PT.mode = 2u;
    PT.block = 1024;
    PT.stage_rows = 0;
    PT.n_producers = (uint32_t)std::min(1024, device_cu_count());
    if (o.partition_producers > 0) PT.n_producers = (uint32_t)std::min(1024, o.partition_producers);
  } else if (want == 2 && partition_ring_bytes(PT.n_words, PT.n_parts, 16) <= (size_t)158 * 1024) {
    const bool hot = o.hot_keys > 0 || (o.hot_keys < 0 && skew_seen);
    const bool chunks8 = !((uint32_t)o.partition_mode & 0x80u);
    if (want_narrow && chunks8) PT.flags |= PTF_NARROW;
    if (hot && na == 1 && chunks8 && partition_ring_bytes(PT.n_words, PT.n_parts, 16, true, (PT.flags & PTF_NARROW) != 0) <= (size_t)158 * 1024)
      PT.flags |= PTF_HOT;
    if ((PT.flags & PTF_NARROW) && !(PT.flags & PTF_HOT) && o.narrow_chunk16 && !(kNarrowLine && o.partition_layout == 2) /* LINE chunks: contiguous regions */ &&
        partition_ring_bytes(PT.n_words, PT.n_parts, kNarrowRingRows, false, true) <= (size_t)158 * 1024)
      PT.flags |= PTF_CHUNK16;
    // selective scans: the scanning and the routing belong to different waves (dfx_k_partition_ws_inl.hpp).  When most rows
    // pass, every wave has rows to route all the time and the ring kernel's symmetric waves are the better fit
    // (round 5, 2^26-row launches, us per launch: ring kernel / 8 + 8 waves / 4 + 12 waves -- selectivity 0.2: - / 415 / 535; 0.5: 326 /
    // 261 / 292; 0.8: 403 / 371 / 353; every row routed: 434 / 438 / 417-424 -- profiles/r05_pass1_ws_by_selectivity.txt.  So: always
    // the wave-specialised kernel, four scanners once more than two thirds of the rows are routed.  agg.pass1_ws_dense = -1: never
    // above one half, round 4's rule)
    const bool ws_fits = o.pass1_ws_dense >= 0 || !dense_seen;
    if ((PT.flags & PTF_CHUNK16) && o.pass1_ws > 0 && ws_fits && !(((uint32_t)o.partition_mode) & ~15u)) {
      // the split: 8 scanners + 8 routers; dense scans (more than half of the rows routed): 4 + 12 (agg.pass1_ws_dense_scanners)
      PT.ws_scanners = (o.pass1_ws == 4 || (mostly_seen && o.pass1_ws_dense_scanners == 4)) ? 4u : 8u;  // (agg.pass1_ws = 4: that split whatever the selectivity -- tests)
      if (partition_ws_bytes(PT.n_parts, (int)PT.ws_scanners) <= (size_t)158 * 1024) PT.flags |= PTF_WS;
    }
    PT.mode = 2u | ((uint32_t)o.partition_mode & ~15u);
    PT.block = 1024;
    PT.stage_rows = 0;
    PT.n_producers = (uint32_t)std::min(1024, device_cu_count());
    if (o.partition_producers > 0) PT.n_producers = (uint32_t)std::min(1024, o.partition_producers);
  } else if (want == 2 && partition_ring_bytes(PT.n_words, PT.n_parts, 8) <= (size_t)158 * 1024) {
    // several aggregates: rows of 3+ words.  8-row rings (two 4-row chunks) still fit where 16-row ones do not
    PT.mode = 2u | 0x100u;
    PT.block = 1024;
    PT.stage_rows = 0;
    PT.n_producers = (uint32_t)std::min(1024, device_cu_count());
    if (o.partition_producers > 0) PT.n_producers = (uint32_t)std::min(1024, o.partition_producers);
  } else if (want != 0 && PT.n_parts <= 1024 && sort_cap >= 4 * PT.n_parts) {
    PT.mode = 1u | ((uint32_t)o.partition_mode & ~15u);
    PT.block = block;
    PT.stage_rows = sort_cap;
    PT.n_producers = (uint32_t)std::min(1024, device_cu_count() * (int)(1024 / block));
  } else {
    // one producer workgroup (1024 lanes) per CU: producers x partitions x 128 B of open region lines
    PT.mode = 0;
    PT.block = 1024;
    PT.stage_rows = 0;
    PT.n_producers = (uint32_t)std::min(1024, device_cu_count());
  }
  const uint64_t avg = (uint64_t)rows / ((uint64_t)PT.n_producers * PT.n_parts) + 1;
  // capacities are whole 64-row trips; LINE chunks (ten rows per 128-byte line, PTF_CHUNK16): whole lines as well
  const uint64_t capq = (PT.flags & PTF_PAIR) ? (uint64_t)kPairCapQuantum : ((PT.flags & PTF_CHUNK16) && kNarrowLine) ? (uint64_t)kNarrowCapQuantum : 64ull;
  pt_worst = (uint32_t)((2 * avg + 64 + capq - 1) / capq * capq);
  // regions hold `window` worst-case batches.  Deferral pays when few rows are routed (headline, 20 %: 2 batches per
  // pass 2 = -3 % per query); when most rows are, the twice-as-long regions cost pass 1 more than the saved launches
  // give back (config 3, 1e9 rows: 11.05 ms at 2, 10.08 ms at 1)
  int window = o.partition_defer > 0 ? std::min(o.partition_defer, 16) : (int)std::max<int64_t>(1, std::min<int64_t>(8, ((int64_t)1 << 27) / std::max<int64_t>(rows, 1)));
  if (dense_seen) window = 1;
  PT.cap_rows = pt_worst * (uint32_t)window;
  if (o.partition_cap_rows > 0) {  // tests: tiny regions (overflow -> spill list); no deferral
    PT.cap_rows = (uint32_t)(((uint64_t)o.partition_cap_rows + capq - 1) / capq * capq);
    pt_worst = PT.cap_rows;
  }
  if (o.pass2_stream && na == 1 && kw == 1) PT.flags |= PTF_STREAM_PASS2;
  uint64_t pad_words = (uint64_t)(o.partition_pad >= 0 ? o.partition_pad : 0) / 8;
  if ((PT.flags & PTF_CHUNK16) && kNarrowLine) pad_words = (pad_words + 15) / 16 * 16;  // (every region starts on a 128-byte line)
  size_t row_bytes;
  const bool line_chunks = (PT.flags & PTF_CHUNK16) && kNarrowLine;  // a region is cap_rows / 10 lines of 128 bytes
  const bool pair_rows = (PT.flags & PTF_PAIR) != 0;  // ... cap_rows / 6 lines
  const uint64_t region_words = pair_rows ? (uint64_t)(PT.cap_rows / (uint32_t)kPairChunkRows) * 16u
                                : line_chunks ? (uint64_t)(PT.cap_rows / (uint32_t)kNarrowChunkRows) * (kNarrowSlotBytes / 8)
                                : (PT.flags & PTF_NARROW) ? (uint64_t)PT.cap_rows * 12 / 8 : (uint64_t)PT.cap_rows * PT.n_words;
  // one pass-2 trip's worth (64 contiguous rows, or six LINE chunks = 60 rows: 768 bytes either way): regions are contiguous (layouts 0 and 1)
  PT.win_stride = pair_rows ? (uint64_t)(kPairTripBytes / 8u) : line_chunks ? 96u : region_words / (PT.cap_rows / 64);
  if (o.partition_layout == 2) {  // windowed: window w of every partition of a producer side by side
    PT.part_stride = PT.win_stride;
    PT.win_stride = (uint64_t)PT.n_parts * PT.part_stride;
    PT.prod_stride = (uint64_t)(PT.cap_rows / 64) * PT.win_stride + pad_words;
    row_bytes = sizeof(uint64_t) * (size_t)PT.n_producers * PT.prod_stride;
  } else if (o.partition_layout == 0) {  // partition-major (round 1)
    PT.prod_stride = region_words;
    PT.part_stride = (uint64_t)PT.n_producers * PT.prod_stride + pad_words;
    row_bytes = sizeof(uint64_t) * (size_t)PT.n_parts * PT.part_stride;
  } else {  // producer-major
    PT.part_stride = region_words;
    PT.prod_stride = (uint64_t)PT.n_parts * PT.part_stride + pad_words;
    row_bytes = sizeof(uint64_t) * (size_t)PT.n_producers * PT.prod_stride;
  }
  const size_t cnt_bytes = sizeof(uint32_t) * (size_t)PT.n_parts * PT.n_producers;
  Status st;
  ScopedUs t_alloc(&counters().agg_alloc_us);
  if (!pt_rows || pt_rows_bytes < row_bytes) {
    pt_rows.reset();
    pt_rows = device_alloc(row_bytes, &st);
    if (!pt_rows) return st;
    pt_rows_bytes = row_bytes;
  }
  if (!pt_counts || pt_cnt_bytes < cnt_bytes) {
    pt_counts.reset();
    pt_counts = device_alloc(cnt_bytes, &st);
    if (!pt_counts) return st;
    pt_cnt_bytes = cnt_bytes;
  }
  PT.rows = (uint64_t*)pt_rows.get();
  PT.counts = (uint32_t*)pt_counts.get();
  pt_layout_valid = true;
  pt_layout_rows = rows;
  pt_pending = 0;
  pt_fill_bound = 0;
  pt_rows_in_flight = 0;
  return Status::OK();
}

// pass 2 over everything the pending pass-1 launches routed (no-op when nothing is pending)
Status AggregateRelation::Impl::flush_pass2() {
  if (pt_pending == 0) return Status::OK();
  ++counters().agg_pass2_launches;
  DFX_HIP(launch_partition_agg(T, PT, spill, 0, ctx().stream));
  pt_pending = 0;
  pt_fill_bound = 0;
  pt_rows_in_flight = 0;
  pt_last_p2_seq = batch_seq;
  return Status::OK();
}

Status AggregateRelation::Impl::read_ctrl(uint32_t* host_ctrl) {
  ScopedUs t(&counters().agg_sync_us);
  hipStream_t s = ctx().stream;
  DFX_HIP(hipMemcpyAsync(host_ctrl, ctrl.get(), sizeof(uint32_t) * CTRL_WORDS, hipMemcpyDeviceToHost, s));
  DFX_HIP(hipStreamSynchronize(s));
  return Status::OK();
}

// queue an asynchronous snapshot of the control block after the batch just launched.  The copy runs on the side
// stream behind an event, so the next batch's kernels follow this batch's directly (an in-stream D2H copy costs
// ~10 us of idle device per batch: rocprofv3 timeline).  The snapshot may already contain counts of the NEXT batch;
// every word is monotone (errors, occupancy, spill cursor), so that only makes the check earlier.
Status AggregateRelation::Impl::alloc_ctrl_host() {
  Status st;
  ctrl_host = pinned_alloc(sizeof(uint32_t) * CTRL_WORDS * 2, &st);
  if (!ctrl_host) return st;
  for (int i = 0; i < 2; ++i) {
    DFX_HIP(hipEventCreateWithFlags(&ctrl_ev[i], hipEventDisableTiming));
    DFX_HIP(hipEventCreateWithFlags(&main_ev[i], hipEventDisableTiming));
  }
  return Status::OK();
}

Status AggregateRelation::Impl::post_ctrl(int64_t rows) {
  hipStream_t s = ctx().stream;
  hipStream_t aux = ctx().aux;
  if (!ctrl_host) DFX_RETURN_IF_ERROR(alloc_ctrl_host());
  const int slot = (int)(batch_seq & 1);
  if (snap_armed) {
    // the batch's last kernel writes the snapshot into this slot of the pinned buffer: all there is to wait for is the
    // kernel itself.  (The previous occupant of the slot, two batches back, was examined after the previous launch.)
    snap_armed = false;
    DFX_HIP(hipEventRecord(ctrl_ev[slot], s));
  } else {
    if (ctrl_pending[slot]) DFX_RETURN_IF_ERROR(examine_ctrl(slot));
    DFX_HIP(hipEventRecord(main_ev[slot], s));
    DFX_HIP(hipStreamWaitEvent(aux, main_ev[slot], 0));
    DFX_HIP(hipMemcpyAsync((uint32_t*)ctrl_host.get() + slot * CTRL_WORDS, ctrl.get(), sizeof(uint32_t) * CTRL_WORDS,
                           hipMemcpyDeviceToHost, aux));
    DFX_HIP(hipEventRecord(ctrl_ev[slot], aux));
  }
  ctrl_pending[slot] = true;
  ctrl_rows[slot] = rows;
  ctrl_seq[slot] = batch_seq;
  unconfirmed_rows += (uint64_t)rows;
  ++batch_seq;
  return Status::OK();
}

// errors, growth: what the per-batch check has always done, on a (possibly one batch old) snapshot
Status AggregateRelation::Impl::handle_ctrl(const uint32_t* hc, int64_t n) {
  if (hc[CTRL_ERROR]) return error_from_ctrl(hc[CTRL_ERROR]);
  if (narrow && hc[CTRL_WIDE_KEYS] && pair_mode) pair_wide_seen = true;  // (no wide form fits a block that holds one plane: consume_batch falls back)
  if (narrow && hc[CTRL_WIDE_KEYS] && !pair_mode) {
    // a key without a 32-bit image turned up (it went to the spill list): 16-byte rows from now on
    DFX_RETURN_IF_ERROR(flush_pass2());
    narrow = false;
    pt_layout_valid = false;
  }
  occupied_known = hc[CTRL_OCCUPIED];
  const uint64_t spilled = ((uint64_t)hc[CTRL_SPILL_HI] << 32) | hc[CTRL_SPILL_LO];
  uint64_t passed_total = 0;
  if (getenv("DFX_DEBUG") && stats) {  // statistics stripes (debug only: one more synchronous copy)
    std::vector<uint64_t> hs((size_t)kStatStripes * STAT_WORDS);
    (void)hipMemcpy(hs.data(), stats.get(), sizeof(uint64_t) * hs.size(), hipMemcpyDeviceToHost);
    for (int i = 0; i < kStatStripes; ++i) passed_total += hs[(size_t)i * STAT_WORDS + STAT_PASSED];
  }
  if (getenv("DFX_DEBUG"))
    fprintf(stderr, "[dfx] batch n=%lld partition=%d lds=%d occupied=%u spilled=%llu saturated=%u passed=%llu cap=%llu "
            "parts=%u cap_rows=%u stage=%u spillcap=%llu\n", (long long)n, (int)use_partition, (int)lds_enabled,
            hc[CTRL_OCCUPIED], (unsigned long long)spilled, hc[CTRL_SATURATED],
            (unsigned long long)passed_total,
            (unsigned long long)(T.mask + 1), PT.n_parts, PT.cap_rows, PT.stage_rows, (unsigned long long)spill.capacity);
  if (spilled > 0 || hc[CTRL_SATURATED] || occupied_known > T.load_limit) {
    // later batches may already be running against the saturated table: let them finish, then rebuild.  Rows still
    // waiting in the routing regions belong to the blocks of THIS table: aggregate them first.
    DFX_RETURN_IF_ERROR(flush_pass2());
    uint32_t now[CTRL_WORDS];
    DFX_RETURN_IF_ERROR(read_ctrl(now));
    ctrl_pending[0] = ctrl_pending[1] = false;
    unconfirmed_rows = 0;
    if (now[CTRL_ERROR]) return error_from_ctrl(now[CTRL_ERROR]);
    uint64_t spilled_now = ((uint64_t)now[CTRL_SPILL_HI] << 32) | now[CTRL_SPILL_LO];
    uint64_t replay_from = 0;
    if (opt().replay_in_place && spilled_now > 0 && !now[CTRL_SATURATED] && now[CTRL_OCCUPIED] <= T.load_limit &&
        2 * spilled_now <= spill.capacity) {
      // The table is not full: the rows were spilled by overflowing routing regions (a hot key).  Put them into the
      // table as it is; a row it cannot take is appended to the list BEHIND the rows being replayed (the cursor is not
      // reset), and only those make the table grow.
      hipStream_t s = ctx().stream;
      DFX_HIP(launch_merge_rows(spill, 0, (int64_t)spilled_now, T, spill, s));
      uint32_t after[CTRL_WORDS];
      DFX_RETURN_IF_ERROR(read_ctrl(after));
      if (after[CTRL_ERROR]) return error_from_ctrl(after[CTRL_ERROR]);
      const uint64_t cursor = ((uint64_t)after[CTRL_SPILL_HI] << 32) | after[CTRL_SPILL_LO];
      if (cursor == spilled_now && !after[CTRL_SATURATED] && after[CTRL_OCCUPIED] <= T.load_limit) {
        after[CTRL_SPILL_LO] = after[CTRL_SPILL_HI] = 0;
        DFX_HIP(hipMemcpyAsync(ctrl.get(), after, sizeof(uint32_t) * CTRL_WORDS, hipMemcpyHostToDevice, s));
        DFX_HIP(hipStreamSynchronize(s));  // `after` is a stack buffer
        occupied_known = after[CTRL_OCCUPIED];
        return Status::OK();
      }
      replay_from = spilled_now;
      spilled_now = cursor;
      memcpy(now, after, sizeof(now));
    }
    DFX_RETURN_IF_ERROR(grow_and_replay(now[CTRL_OCCUPIED], spilled_now, replay_from));
    DFX_RETURN_IF_ERROR(read_ctrl(now));
    occupied_known = now[CTRL_OCCUPIED];
  }
  return Status::OK();
}

Status AggregateRelation::Impl::examine_ctrl(int slot) {
  if (!ctrl_pending[slot]) return Status::OK();
  {
    ScopedUs t(&counters().agg_ctrl_wait_us);
    DFX_HIP(hipEventSynchronize(ctrl_ev[slot]));
  }
  ctrl_pending[slot] = false;
  unconfirmed_rows -= std::min<uint64_t>(unconfirmed_rows, (uint64_t)ctrl_rows[slot]);
  uint32_t hc[CTRL_WORDS];
  memcpy(hc, (const uint32_t*)ctrl_host.get() + slot * CTRL_WORDS, sizeof(hc));
  if (use_partition && ctrl_seq[slot] > pt_last_p2_seq) {
    // the snapshot was taken after the launch with sequence number ctrl_seq[slot] (it may already show later launches:
    // only larger); every launch since then adds at most pt_worst rows to a region
    const uint64_t later = (uint64_t)std::max<int64_t>(0, batch_seq - 1 - ctrl_seq[slot]);
    pt_fill_bound = std::min<uint64_t>(pt_fill_bound, (uint64_t)hc[CTRL_MAX_FILL] + later * pt_worst);
  }
  return handle_ctrl(hc, ctrl_rows[slot]);
}

// everything launched so far has been checked (end of input, or before the spill list is replaced)
Status AggregateRelation::Impl::settle_ctrl() {
  if (!ctrl_pending[0] && !ctrl_pending[1]) return Status::OK();
  const int older = (int)(batch_seq & 1);  // the slot the NEXT batch would use holds the older snapshot
  DFX_RETURN_IF_ERROR(examine_ctrl(older));
  DFX_RETURN_IF_ERROR(examine_ctrl(older ^ 1));
  return Status::OK();
}

// The table passed its load limit (or a probe sequence was exhausted): build a table at least 4x
// larger, rehash, then replay the spilled rows into it.  Afterwards occupancy <= 1/4.
Status AggregateRelation::Impl::grow_and_replay(uint64_t occupied, uint64_t spilled, uint64_t replay_from) {
  ++counters().agg_growths;
  hipStream_t s = ctx().stream;
  if (spilled > spill.capacity)
    return Status::Err(DFX_INTERNAL_ERROR, strfmt("group spill list overflow (%llu rows > capacity %llu)",
                                                  (unsigned long long)spilled, (unsigned long long)spill.capacity));
  const int cur_log2 = 64 - T.shift;
  const int need_log2 = ceil_log2(4 * (occupied + (spilled - replay_from) + 1));
  const int new_log2 = std::max(cur_log2 + 2, need_log2);
  if (new_log2 > 31) return Status::Err(DFX_EXECUTION_ERROR, "GROUP BY table would exceed 2^31 slots");
  DevTable Tn;
  std::vector<std::shared_ptr<void>> owners;
  uint64_t* accs_full_new = nullptr;
  DFX_RETURN_IF_ERROR(alloc_table(new_log2, &Tn, &owners, false, &accs_full_new));
  // reset the shared control words that describe the (new) table
  uint32_t zeros[CTRL_WORDS];
  memset(zeros, 0, sizeof(zeros));
  uint32_t host_ctrl[CTRL_WORDS];
  DFX_RETURN_IF_ERROR(read_ctrl(host_ctrl));
  host_ctrl[CTRL_OCCUPIED] = 0;
  host_ctrl[CTRL_SPILL_LO] = host_ctrl[CTRL_SPILL_HI] = 0;
  host_ctrl[CTRL_SENTINEL] = 0;
  host_ctrl[CTRL_SATURATED] = 0;
  const uint32_t had_sentinel = 0;  // rehash re-raises it when it meets the sentinel slot
  (void)had_sentinel;
  DevRows no_spill;
  no_spill.words = nullptr;
  no_spill.capacity = 0;
  // `from` still needs the old CTRL_SENTINEL to know whether slot `cap` is occupied: give the old
  // table a private copy of the control block for the duration of the rehash
  Status st;
  auto old_ctrl = device_alloc(sizeof(uint32_t) * CTRL_WORDS, &st);
  if (!old_ctrl) return st;
  DFX_HIP(hipMemcpyAsync(old_ctrl.get(), ctrl.get(), sizeof(uint32_t) * CTRL_WORDS, hipMemcpyDeviceToDevice, s));
  DFX_HIP(hipMemcpyAsync(ctrl.get(), host_ctrl, sizeof(uint32_t) * CTRL_WORDS, hipMemcpyHostToDevice, s));
  DFX_HIP(hipStreamSynchronize(s));  // host_ctrl is a stack buffer
  pt_layout_valid = false;  // the routing regions are per table block
  DevTable Told = T;
  Told.ctrl = (uint32_t*)old_ctrl.get();
  DFX_HIP(launch_rehash(Told, Tn, no_spill, s));
  for (int c = 0; c < (int)chunks.size(); ++c) {  // the other chunks' planes move the same way (their keys are already in place)
    if (c == cur_chunk) continue;
    DFX_HIP(launch_rehash(view_of(Told, accs_full, c), view_of(Tn, accs_full_new, c), no_spill, s));
  }
  if (spilled > replay_from) DFX_HIP(launch_merge_rows(spill, (int64_t)replay_from, (int64_t)(spilled - replay_from), Tn, no_spill, s));
  early.cancel();  // (its kernels read the old table)
  ++table_generation;
  T = Tn;
  accs_full = accs_full_new;
  table_owners = owners;  // old buffers return to the pool once the stream has passed them
  DFX_HIP(hipStreamSynchronize(s));
  return Status::OK();
}

Status AggregateRelation::Impl::launch_rows(const DeviceBatch& b, const DevProgram& prog_in, const DevColumns& cols_in,
                                            int64_t row0, int64_t n) {
  hipStream_t s = ctx().stream;
  snap_armed = false;
  DevProgram prog = prog_in;
  DevColumns cols = cols_in;
  double bytes = 0;
  for (int i = 0; i < prog.n_cols; ++i) {  // advance the bound columns to row0 (row0 is a multiple of 64)
    const int w = prog.col_dtype[i] == T_BOOL ? 0 : dtype_width(prog.col_dtype[i]);
    if (w) cols.c[i].values = (const uint8_t*)cols.c[i].values + (size_t)row0 * w;
    else cols.c[i].bit_offset += row0;
    if (cols.c[i].validity && w) cols.c[i].bit_offset += row0;
    bytes += (double)n * (w ? w : 0.125);
  }
  DevAggPlan p = plan;
  // (while the per-aggregate chunking is pending -- the table's blocks are sized for one accumulator per scan -- the all-aggregates
  // program never takes the partitioned strategy: its pass 2 would not fit a block into LDS)
  bool partition_now = use_partition && (pair_mode || !(split_applies() && !split_done));
  if (partition_now) {
    Status pst = ensure_partition(launch_rows_hint > 0 ? std::max<int64_t>(n, std::min<int64_t>(launch_rows_hint, b.num_rows)) : std::max<int64_t>(n, b.num_rows), prog.has_nulls != 0);  // (the slice after the calibration rows: size for the whole batch)
    if (!pst.ok() && pst.code == DFX_NOT_IMPLEMENTED) partition_now = false;  // global-atomic path instead
    else if (!pst.ok()) return pst;
  }
  if (partition_now) {
    DevFastPlan fpp = fast;
    if (!opt().fast) fpp.valid = 0;
    fpp.plan_mode = opt().plan | (plan_required ? 4 : 0);
    DevPartition pt = PT;
    if (pt_pending > 0) pt.flags |= PTF_RESUME;
    // close the window when one more batch could overflow a region (or the batch budget is used up; the calibration
    // slice is aggregated at once: the strategy decision reads the group count)
    const int max_batches = pair_mode ? std::min(2, std::max(1, opt().partition_defer_batches)) : std::max(1, opt().partition_defer_batches);  // (pair_mode: the spill list's sizing)
    const bool close_window = calibrating || pt_pending + 1 >= max_batches || pt_fill_bound + 2 * (uint64_t)pt_worst > PT.cap_rows;
    // the LAST kernel of this batch publishes the control block itself (examined one batch later, see post_ctrl)
    snap_armed = false;
    uint32_t* snap_to = nullptr;
    if (opt().ctrl_snapshot == 1 && lds_calibrated && !calibrating) {
      if (!ctrl_host) DFX_RETURN_IF_ERROR(alloc_ctrl_host());
      if (!snap_done) {
        Status st;
        snap_done = device_alloc(sizeof(uint32_t) * 16, &st);
        if (!snap_done) return st;
        DFX_HIP(hipMemsetAsync(snap_done.get(), 0, sizeof(uint32_t) * 16, s));
      }
      snap_to = (uint32_t*)ctrl_host.get() + (size_t)(batch_seq & 1) * CTRL_WORDS;
      snap_armed = true;
    }
    if (!close_window) {
      pt.snap_host = snap_to;
      pt.snap_done = (uint32_t*)snap_done.get();
    }
    DFX_HIP(launch_partition(prog, fpp, cols, p, T, pt, spill, n, bytes, s));
    if (pt.flags & PTF_SHARED) ++counters().agg_shared_operand_launches;
    if (pt.flags & PTF_PAIR) ++counters().agg_pair_launches;
    if (pt.flags & PTF_PLANES) ++counters().agg_plane_launches;
    ++pt_pending;
    pt_fill_bound += pt_worst;
    pt_rows_in_flight += n;
    if (close_window) {
      PT.snap_host = snap_to;
      PT.snap_done = (uint32_t*)snap_done.get();
      Status fst = flush_pass2();
      PT.snap_host = nullptr;
      PT.snap_done = nullptr;
      DFX_RETURN_IF_ERROR(fst);
    }
    return Status::OK();
  }
  DevFastPlan fp = fast;
  if (!opt().fast) fp.valid = 0;
  fp.plan_mode = opt().plan | (plan_required ? 4 : 0);
  // a handful of groups (the calibration slice / earlier batches saw <= 8): register accumulators.  Should more
  // groups turn up later the kernel still handles them (through the table), and the next batch goes back to K7.
  if (lds_enabled && lds_calibrated && !calibrating && opt().strategy != 1 && opt().fewgroup &&
      occupied_known > 0 && occupied_known <= 8 && fewgroup_supported(prog, fp, T)) {
    DFX_HIP(launch_fewgroup_agg(prog, fp, cols, p, T, spill, n, bytes, s));
    return Status::OK();
  }
  if (lds_enabled && opt().strategy != 1) {
    const AggOptions& o = opt();
    int slots = o.lds_slots >= 0 ? o.lds_slots : 4096;
    if (calibrating && o.lds_slots < 0) slots = 512;  // calibration slice: the cache only has to tell few groups from many
    while (slots > 64 && (size_t)slots * ((size_t)(kw + na) * 8 + (kw > 1 ? 4 : 0)) > 64 * 1024) slots >>= 1;
    int copies = o.lds_copies > 0 ? o.lds_copies : 1;
    if (o.lds_copies <= 0 && lds_calibrated) {  // few groups: lane-replicated sub-tables
      if (occupied_known <= 16) copies = 16;
      else if (occupied_known <= 128) copies = 4;
    }
    while (copies > 1 && slots / copies < 64) copies >>= 1;
    p.lds_slots = slots;
    p.lds_copies = copies;
  } else {
    p.lds_slots = 0;
    p.lds_copies = 1;
  }
  DFX_HIP(launch_hash_agg(prog, fp, cols, p, T, spill, n, bytes, s));
  (void)b;
  return Status::OK();
}

namespace {
// one batch as a relation (input of the per-batch FilterRelation of the unfused path)
struct OneBatchRelation : Relation {
  DeviceBatch batch;
  SchemaInfo schema_;
  bool done = false;
  RelationKind kind() const override { return REL_TABLE_SCAN; }
  const SchemaInfo& schema() const override { return schema_; }
  Status next(DeviceBatch* out, bool* has) override {
    *has = !done;
    if (!done) *out = batch;
    done = true;
    return Status::OK();
  }
};
}  // namespace

Status AggregateRelation::Impl::consume_batch_chunk(const DeviceBatch& b) {
  plan_required = false;
  if (has_pred && !unfused_now && b.num_rows > 0) {
    bool nulls = false;
    for (int ci : builder->columns())
      if (ci >= 0 && ci < (int)b.columns.size() && b.columns[(size_t)ci].validity && b.columns[(size_t)ci].null_count != 0) nulls = true;
    // A scan plan evaluates the fused form with exactly those rules -- a null judged by arrow's comparison rule, every
    // surviving slot valid, value(row) read regardless (DevScanPlan::count_valid) -- in one pass: no materialised filter.
    if (nulls && opt().plan != 0 && opt().fast != 0 && scan_plan_shape_ok(builder->program(), fast, kw, na, val_xform)) {
      nulls = false;
      plan_required = true;  // (this batch's launchers must bind the plan: nothing else evaluates the fused form correctly)
    }
    if (nulls) {  // FilterRelation for real (its output is all-valid), then the predicate-free program
      std::unique_ptr<OneBatchRelation> one(new OneBatchRelation());
      one->batch = b;
      one->schema_ = input->schema();
      FilterRelation filter(std::move(one), pred, input->schema());
      std::vector<char> needed(input->schema().fields.size(), 0);
      for (int ci : builder_np->columns())
        if (ci >= 0 && ci < (int)needed.size()) needed[(size_t)ci] = 1;
      for (const DictKey& d : dicts)
        if (d.src_col >= 0 && d.src_col < (int)needed.size()) needed[(size_t)d.src_col] = 1;
      filter.require_columns(needed);
      DeviceBatch fb;
      bool got = false;
      DFX_RETURN_IF_ERROR(filter.next(&fb, &got));
      if (!got) return Status::OK();
      std::swap(builder, builder_np);
      std::swap(plan, plan_np);
      std::swap(fast, fast_np);
      unfused_now = true;
      const bool stop = stop_after_decision;  // (the filtered batch is this call's own: it is consumed whole, whatever is decided on the way)
      stop_after_decision = false;
      Status st = consume_batch_chunk(fb);
      stop_after_decision = stop;
      if (stop) decided_rows = b.num_rows;
      unfused_now = false;
      std::swap(builder, builder_np);
      std::swap(plan, plan_np);
      std::swap(fast, fast_np);
      return st;
    }
  }
  // The absorbed FilterRelation's batch-level error survives the fusion: fn filter has no arm for Boolean
  // (filter.rs:105-108), so a batch with a Boolean column fails under a Filter whether or not anybody reads that column
  // and whether this batch runs fused (no nulls) or through a real FilterRelation (nulls in the program's columns)
  if (has_pred && !unfused_now)
    for (size_t c = 0; c < b.columns.size(); ++c)
      if (b.columns[c].dtype == DFX_BOOLEAN) return Status::Err(DFX_EXECUTION_ERROR, "filter not supported for Boolean");
  const int64_t n = b.num_rows;
  if (n == 0 && kw > 0) return Status::OK();  // (ungrouped: an empty batch still folds Some(0) into COUNT)
  hipStream_t s = ctx().stream;
  DevProgram prog;
  DevColumns cols;
  if (dicts.empty()) {
    DFX_RETURN_IF_ERROR(builder->bind(b, &prog, &cols));
  } else {  // append the id column of every Utf8 key
    DeviceBatch ab = b;
    ab.columns.resize(bind_schema.fields.size());
    for (DictKey& d : dicts) {
      if (d.src_col >= (int)b.columns.size() || b.columns[d.src_col].dtype != DFX_UTF8)
        return Status::Err(DFX_INTERNAL_ERROR, "GROUP BY key column is not Utf8 in this batch");
      DFX_RETURN_IF_ERROR(dict_encode(d, b.columns[d.src_col], n, &ab.columns[d.virt_col]));
    }
    DFX_RETURN_IF_ERROR(builder->bind(ab, &prog, &cols));
  }
  if (kw == 0) {
    double bytes = 0;
    for (int i = 0; i < prog.n_cols; ++i) bytes += (double)n * (prog.col_dtype[i] == T_BOOL ? 0.125 : dtype_width(prog.col_dtype[i]));
    DevFastPlan fp = fast;
    if (!opt().fast) fp.valid = 0;
    fp.plan_mode = opt().plan | (plan_required ? 4 : 0);
    DFX_HIP(launch_reduce(prog, fp, cols, plan, T, n, (uint64_t*)partial.get(), (uint32_t*)ctrl.get(), bytes, s));
    DFX_HIP(launch_reduce_fold(T, (const uint8_t*)dev_arg_dtype.get(), (const uint8_t*)dev_func.get(),
                               (uint64_t*)partial.get(), (uint64_t*)state.get(), (uint32_t*)ctrl.get(), s));
    rows_seen += n;
    return Status::OK();
  }
  // grouped: can this batch overflow the table in the worst case (every row a new group)?
  const AggOptions& oo = opt();
  if (oo.strategy == 3 && kw == 1) {
    if (!use_partition && oo.narrow_keys > 0) narrow = true;  // forced strategy: no calibration slice -- optimistic (tests); ensure_partition looks at the shape
    use_partition = true;
  }
  const bool may_spill = use_partition || occupied_known + unconfirmed_rows + (uint64_t)n > T.load_limit;
  // two batches can be in flight unchecked; with a deferred pass 2 every row of the window may still be spilled (by pass 2
  // itself, when its block is full)
  // (pair scan: a row whose key finds no slot in its block is spilled once per OPERAND -- by the last plane of each, dfx_k_partition.hip --;
  // planes of a shared operand: once.  Their windows hold at most two batches, launch_rows)
  const int64_t window_rows = use_partition ? (pair_mode ? (int64_t)std::min(2, std::max(1, opt().partition_defer_batches)) * (pair_is_planes ? 1 : 2)
                                                         : (int64_t)std::max(1, opt().partition_defer_batches)) * std::max(n, pt_layout_rows) : 0;
  if (may_spill) DFX_RETURN_IF_ERROR(ensure_spill(2 * n + window_rows + 65536));
  T.max_probe = may_spill ? 128 : (int)std::min<uint64_t>(T.mask + 1, 1u << 30);
  int64_t row0 = 0;
  const AggOptions& o = opt();
  // (a batch that went through a real FilterRelation is not the table's first rows: its calibration says nothing about them)
  ScanMemo* memo = (o.calibration_memo && !unfused_now) ? input->scan_memo() : nullptr;
  uint64_t remembered = 0;
  if (!lds_calibrated && o.strategy == 0 && n > (1 << 21) && memo && memo->lookup(program_fingerprint(), &remembered)) {
    // an earlier query of this shape over the same resident table already ran the calibration slice: same decision,
    // no slice, no synchronous read-back (the real group count arrives with the control-block snapshots as always)
    skew_seen = (remembered >> 63) != 0;
    narrow = ((remembered >> 62) & 1) != 0;
    dense_seen = ((remembered >> 61) & 1) != 0;
    mostly_seen = ((remembered >> 60) & 1) != 0;
    remembered &= ~(15ull << 60);
    occupied_known = remembered;
    lds_calibrated = true;
    lds_enabled = remembered <= 8192;
    if (!lds_enabled && kw == 1 && remembered >= 16384) {
      use_partition = true;
      DFX_RETURN_IF_ERROR(ensure_spill(2 * n + (int64_t)std::max(1, opt().partition_defer_batches) * n + 65536));
    }
  }
  if (!lds_calibrated && o.strategy == 0 && n > (1 << 21)) {
    // calibration slice: measure the LDS front-cache hit rate and the group count on the first
    // 2^18 rows before committing the rest of the stream to a strategy
    const int64_t n0 = 1 << 18;
    calibrating = true;
    Status cst = launch_rows(b, prog, cols, 0, n0);
    calibrating = false;
    DFX_RETURN_IF_ERROR(cst);
    if (kw == 1) DFX_HIP(launch_probe_wide_keys(T, ctx().stream));  // does any key of the slice lack a 32-bit image?
    uint32_t hc[CTRL_WORDS];
    DFX_RETURN_IF_ERROR(read_ctrl(hc));
    if (hc[CTRL_ERROR]) return error_from_ctrl(hc[CTRL_ERROR]);
    if ((((uint64_t)hc[CTRL_SPILL_HI] << 32) | hc[CTRL_SPILL_LO]) > 0 || hc[CTRL_SATURATED]) {
      // The slice did not fit the table (a table that starts very small: agg.capacity_log2): its spilled rows sit in the spill list
      // that the strategy decision below is about to REPLACE by a larger one.  Round 6, found by a test of the pair scan at 2^14
      // slots: nothing replayed them first -- the spill cursor went on counting them, the rebuild after the batch replayed whatever
      // the new list's memory held in their place (60-80 of 200 000 groups missing, or keys that never were in the data).  Grow /
      // replay now; the decision then reads the real group count of the slice.
      DFX_RETURN_IF_ERROR(handle_ctrl(hc, n0));
      DFX_RETURN_IF_ERROR(read_ctrl(hc));
      if (hc[CTRL_ERROR]) return error_from_ctrl(hc[CTRL_ERROR]);
    }
    // (a property of the KEYS: whether a launch routes 12-byte rows also depends on the aggregates of the chunk it serves --
    // ensure_partition -- and a query that is split into one scan per aggregate has one-aggregate chunks after this point)
    narrow = kw == 1 && hc[CTRL_WIDE_KEYS] == 0;
    // strategy from the number of groups the calibration slice produced: the LDS front cache pays
    // when the groups fit it (every later row is an LDS atomic); for many groups per-row global
    // atomics would cap the query near 24 G rows/s, so rows are routed to their table blocks
    // instead (dfx_k_partition.hip); in between, the global table alone.
    occupied_known = hc[CTRL_OCCUPIED];
    {  // share of the slice's rows that a 512-slot front cache absorbed: ~0 for a million uniform keys, a third and
       // more under a Zipf-like distribution (statistics stripes of K7; one more small synchronous copy, once per stream)
      std::vector<uint64_t> hs((size_t)kStatStripes * STAT_WORDS, 0);
      DFX_HIP(hipMemcpy(hs.data(), stats.get(), sizeof(uint64_t) * hs.size(), hipMemcpyDeviceToHost));
      uint64_t hit = 0, miss = 0, passed = 0;
      for (int i = 0; i < kStatStripes; ++i) {
        hit += hs[(size_t)i * STAT_WORDS + STAT_LDS_HIT];
        miss += hs[(size_t)i * STAT_WORDS + STAT_LDS_MISS];
        passed += hs[(size_t)i * STAT_WORDS + STAT_PASSED];
      }
      dense_seen = passed * 2 > (uint64_t)n0;
      mostly_seen = passed * 3 > (uint64_t)n0 * 2;
      skew_seen = occupied_known >= 16384 && miss > 0 && hit * 8 >= miss;  // (`miss` counts every row that went through the cache) >= 12.5 % reused although the groups do not fit
    }
    if (memo) memo->remember(program_fingerprint(), occupied_known | (skew_seen ? 1ull << 63 : 0ull) | (narrow ? 1ull << 62 : 0ull) | (dense_seen ? 1ull << 61 : 0ull) | (mostly_seen ? 1ull << 60 : 0ull));
    lds_calibrated = true;
    lds_enabled = occupied_known <= 8192;
    if (!lds_enabled && kw == 1 && occupied_known >= 16384) {
      use_partition = true;
      DFX_RETURN_IF_ERROR(ensure_spill(2 * n + (int64_t)std::max(1, opt().partition_defer_batches) * n + 65536));
    }
    row0 = n0;
  } else if (!lds_calibrated) {
    if (o.strategy == 1) lds_enabled = false;
  }
  if (stop_after_decision && lds_calibrated) {  // (consume_batch: the decision is what was asked for; rows [0, row0) are done)
    decided_rows = row0;
    rows_seen += row0;
    return Status::OK();
  }
  {  // a scan that routes most of its rows: launches of at most partition_split_rows rows (regions sized for that many)
    // (selective scans: twice that -- 2^27-row launches measured best, 2^28-row ones 7 % slower)
    const int64_t split = (use_partition && o.partition_split_rows >= (1 << 20)) ? (((int64_t)o.partition_split_rows * (dense_seen ? 1 : 2)) & ~(int64_t)63) : 0;
    launch_rows_hint = split;
    Status lst = Status::OK();
    if (split > 0 && n - row0 > split) {
      for (int64_t at = row0; at < n && lst.ok(); at += split) lst = launch_rows(b, prog, cols, at, std::min(split, n - at));
    } else {
      lst = launch_rows(b, prog, cols, row0, n - row0);
    }
    launch_rows_hint = 0;
    DFX_RETURN_IF_ERROR(lst);
  }
  if (!lds_calibrated) {  // first batch of a stream that skipped the calibration slice: decide now
    uint32_t hc[CTRL_WORDS];
    DFX_RETURN_IF_ERROR(read_ctrl(hc));
    DFX_RETURN_IF_ERROR(handle_ctrl(hc, n));
    if (o.strategy == 0) lds_enabled = occupied_known <= 8192;
    lds_calibrated = true;
  } else {
    const int prev = (int)((batch_seq & 1) ^ 1);
    DFX_RETURN_IF_ERROR(post_ctrl(n));       // snapshot of THIS batch, examined after the next launch
    DFX_RETURN_IF_ERROR(examine_ctrl(prev)); // the previous batch's snapshot (normally complete by now)
    DFX_RETURN_IF_ERROR(early_keys_maybe());
  }
  rows_seen += n;
  if (stop_after_decision) decided_rows = n;  // (a batch too small for a calibration slice: it ran whole, decided afterwards)
  return Status::OK();
}

// One input batch through every chunk of accumulators.  With several chunks each chunk's kernels are checked
// synchronously (errors, spilled rows, growth) before the next chunk runs: the spill list and the routing scratch carry
// rows of ONE chunk's width at a time.
static DeviceBatch rows_from(const DeviceBatch& b, int64_t row0) {  // rows [row0, end) of a batch, zero copy (row0: a multiple of 64)
  DeviceBatch r;
  r.num_rows = b.num_rows - row0;
  r.columns.reserve(b.columns.size());
  for (const DeviceColumn& c : b.columns) {
    DeviceColumn s = c;
    s.length = r.num_rows;
    if (!c.absent) {
      if (c.dtype == DFX_UTF8) {
        if (c.offsets) s.offsets = c.offsets + row0;
        s.data_bytes = 0;
      } else if (c.dtype == DFX_BOOLEAN) {
        if (c.values) s.values = (const uint8_t*)c.values + (row0 >> 3);
      } else if (c.values) {
        s.values = (const uint8_t*)c.values + (size_t)row0 * dtype_width(c.dtype);
      }
      if (c.validity) s.validity = c.validity + (row0 >> 3);
      if (c.null_count != 0) s.null_count = -1;
    }
    r.columns.push_back(std::move(s));
  }
  return r;
}

Status AggregateRelation::Impl::consume_batch(const DeviceBatch& b) {
  if (split_applies() && !split_done && !split_decided) {
    // The strategy decision first (calibration slice, the resident table's memo, a forced strategy), with the all-aggregates
    // program and nothing else of the batch; if it says "partitioned", the per-aggregate chunking takes over from there.
    const bool forced = opt().strategy == 3;  // (no decision to wait for: the chunk loop below turns the strategy on itself)
    if (forced) {
      decided_rows = 0;
    } else {
      stop_after_decision = true;
      decided_rows = 0;
      Status st = consume_batch_chunk(b);
      stop_after_decision = false;
      DFX_RETURN_IF_ERROR(st);
    }
    // decided = a strategy has been chosen.  An empty first batch (or one a real FilterRelation emptied) chooses nothing:
    // the next batch comes back here (round-4 advisor finding: a decision recorded before any calibration left a later
    // "partitioned" verdict with the split pending for good -- every launch on per-row global atomics)
    split_decided = forced || lds_calibrated || (decided_rows == 0 && b.num_rows > 0);
    if (forced && kw == 1) {  // (what consume_batch_chunk does for a forced strategy, before the choice below looks at `narrow`)
      if (!use_partition && opt().narrow_keys > 0) narrow = true;
      use_partition = true;
    }
    if (split_decided && (use_partition || forced)) {
      DFX_RETURN_IF_ERROR(flush_pass2());
      DFX_RETURN_IF_ERROR(settle_ctrl());
      if ((split_is_shared ? opt().shared_planes : opt().pair_scan) && use_partition && pair_batch_ok(b)) {
        pair_mode = true;  // the all-aggregates program goes on: one scan for both operands (single_chunks stays in reserve)
      } else {
        install_chunks(std::move(single_chunks));
        single_chunks.clear();
        split_done = true;
      }
    }
    if (decided_rows >= b.num_rows) return Status::OK();
    if (decided_rows == 0) return consume_batch(b);
    return consume_batch(rows_from(b, decided_rows));
  }
  if (pair_mode && (pair_wide_seen || !pair_batch_ok(b))) DFX_RETURN_IF_ERROR(pair_fall_back());
  if (chunks.size() <= 1) return consume_batch_chunk(b);
  if (kw > 0 && opt().chunk_hold > 1) {  // grouped, several chunks: hold the batch (see `held`)
    size_t bytes = 0;
    for (const DeviceColumn& c : b.columns) bytes += (size_t)std::max<int64_t>(c.length, 0) * (size_t)std::max(1, dtype_width(c.dtype));
    held.push_back(b);
    held_bytes += bytes;
    if ((int)held.size() < opt().chunk_hold && held_bytes < ((size_t)8 << 30)) return Status::OK();
    return run_held();
  }
  for (int c = 0; c < (int)chunks.size(); ++c) {
    activate(c);
    const int64_t seen = rows_seen;
    DFX_RETURN_IF_ERROR(consume_batch_chunk(b));
    rows_seen = seen;
    if (kw > 0) {
      DFX_RETURN_IF_ERROR(flush_pass2());
      DFX_RETURN_IF_ERROR(settle_ctrl());
      uint32_t hc[CTRL_WORDS];
      DFX_RETURN_IF_ERROR(read_ctrl(hc));
      DFX_RETURN_IF_ERROR(handle_ctrl(hc, b.num_rows));
    }
  }
  activate(0);
  rows_seen += b.num_rows;
  return Status::OK();
}

// Can this batch go through the pair scan (see pair_mode)?  Host work only: the program is bound to the batch and the scan plan to that.
bool AggregateRelation::Impl::pair_batch_ok(const DeviceBatch& b) {
  const bool dbg = getenv("DFX_DEBUG") != nullptr;
  auto no = [&](const char* why) {
    if (dbg) fprintf(stderr, "[dfx] pair scan: no (%s)\n", why);
    return false;
  };
  if (!kNarrowLine || !narrow || kw != 1 || chunks.size() != 1 || (int)single_chunks.size() != na || !dicts.empty() || unfused_now) return no("shape");
  if (split_is_shared ? !(na >= 2 && na <= kMaxAggs && same_operand_all() && opt().shared_planes) : !(split_distinct == 2 && na >= 2 && na <= kMaxAggs)) return no("aggregates");
  const AggOptions& o = opt();
  // Skewed keys (the calibration slice's front cache absorbed a sizeable share of its rows): the one-value scans keep the heavy keys
  // in LDS (PTF_HOT) -- the pair rows and the planes have no such thing, a heavy key overflows its regions into the spill list and
  // the replay queues on a few addresses (Zipf(1.0), 10^9 rows: SUM(v), MIN(w) 51 ms against 12.2 for a scan per aggregate;
  // SUM(v), MIN(v) 43 ms, 96 with the all-planes blocks).  One scan per aggregate then.
  if (o.hot_keys > 0 || (o.hot_keys < 0 && skew_seen)) return no("skewed keys: the one-value scans have the hot-key pairs");
  if ((!split_is_shared && (!o.plan || !o.fast)) || o.narrow_keys == 0 || !o.narrow_chunk16 || o.pass1_ws <= 0 || o.partition_layout == 2 || ((uint32_t)o.partition_mode & 0x8Fu) != 2u) return no("options");
  if (!split_is_shared && !scan_plan_shape_ok(builder->program(), fast, kw, na, val_xform)) return no("scan plan shape");  // (also: a predicate over nulls stays fused, consume_batch_chunk)
  const uint64_t S = (uint64_t)T.block_mask + 1;
  if (S != 8192 || partition_ws_bytes((uint32_t)((T.mask + 1) / S), split_is_shared ? 4 : 8, split_is_shared ? 1 : 2) > (size_t)158 * 1024) return no("table blocks");
  if (b.num_rows <= 0) {
    pair_is_planes = split_is_shared;
    return true;
  }
  DevProgram prog;
  DevColumns cols;
  if (!builder->bind(b, &prog, &cols).ok()) return no("bind");
  DevFastPlan fp = fast;
  if (!o.fast) fp.valid = 0;
  fp.plan_mode = o.plan;
  if (split_is_shared) {  // the raw operand through the one-value kernels: a null-free batch, a signature or the plan's fixed-slot binding
    if (!partition_planes_supported(prog, fp, cols, T)) return no("one-value binding of the shared operand");
  } else if (!partition_pair_supported(prog, fp, cols, T)) {
    return no("plan binding");
  }
  pair_is_planes = split_is_shared;
  return true;
}

// the pair scan no longer applies: one scan per aggregate from the next batch on (a batch boundary: nothing is half launched)
Status AggregateRelation::Impl::pair_fall_back() {
  const bool flushed = pt_pending > 0;
  DFX_RETURN_IF_ERROR(flush_pass2());
  DFX_RETURN_IF_ERROR(settle_ctrl());
  if (flushed) {  // the pass 2 just launched has no snapshot of its own: rows it spilled carry EVERY accumulator -- replay them under this view
    uint32_t hc[CTRL_WORDS];
    DFX_RETURN_IF_ERROR(read_ctrl(hc));
    DFX_RETURN_IF_ERROR(handle_ctrl(hc, 0));
  }
  ++counters().agg_pair_fallbacks;
  pair_mode = false;
  install_chunks(std::move(single_chunks));
  single_chunks.clear();
  split_done = true;
  if (pair_wide_seen) narrow = false;  // (16-byte routed rows from here on; install_chunks has invalidated the layout)
  return Status::OK();
}

// every chunk over every held batch, one control-block check per chunk
Status AggregateRelation::Impl::run_held() {
  if (held.empty()) return Status::OK();
  std::vector<DeviceBatch> hb;
  hb.swap(held);
  held_bytes = 0;
  const int64_t seen = rows_seen;
  int64_t total = 0;
  for (const DeviceBatch& b : hb) total += b.num_rows;
  for (int c = 0; c < (int)chunks.size(); ++c) {
    activate(c);
    rows_seen = seen;
    for (const DeviceBatch& b : hb) DFX_RETURN_IF_ERROR(consume_batch_chunk(b));
    DFX_RETURN_IF_ERROR(flush_pass2());
    DFX_RETURN_IF_ERROR(settle_ctrl());
    uint32_t hc[CTRL_WORDS];
    DFX_RETURN_IF_ERROR(read_ctrl(hc));
    DFX_RETURN_IF_ERROR(handle_ctrl(hc, total));
  }
  activate(0);
  rows_seen = seen + total;
  return Status::OK();
}

// ---- Utf8 key dictionary (host side of dfx_k_dict.hip) ------------------------------------------------
// (re)allocate a dictionary with 2^slots_log2 slots (ids capacity = half of that) and `pool_cap` pool bytes;
// keep == true carries the strings of completed batches over and rebuilds the slot table from them
Status AggregateRelation::Impl::dict_alloc(DictKey& d, int slots_log2, uint64_t pool_cap, bool keep) {
  hipStream_t s = ctx().stream;
  const uint64_t slots = 1ull << slots_log2, id_cap = slots / 2;
  Status st;
  auto dstate = device_alloc(sizeof(uint32_t) * slots, &st);
  if (!dstate) return st;
  auto hash = device_alloc(sizeof(uint64_t) * slots, &st);
  if (!hash) return st;
  auto sid = device_alloc(sizeof(uint64_t) * slots, &st);
  if (!sid) return st;
  auto str_off = device_alloc(sizeof(uint64_t) * id_cap, &st);
  if (!str_off) return st;
  auto str_len = device_alloc(sizeof(uint32_t) * id_cap, &st);
  if (!str_len) return st;
  auto pool = device_alloc(std::max<uint64_t>(pool_cap, 64), &st);
  if (!pool) return st;
  auto cursors = device_alloc(sizeof(uint64_t) * DICT_WORDS, &st);
  if (!cursors) return st;
  DFX_HIP(hipMemsetAsync(dstate.get(), 0, sizeof(uint32_t) * slots, s));
  if (keep && d.allocated) {
    if (d.pool_used) DFX_HIP(hipMemcpyAsync(pool.get(), d.pool.get(), d.pool_used, hipMemcpyDeviceToDevice, s));
    if (d.ids_used) {
      DFX_HIP(hipMemcpyAsync(str_off.get(), d.str_off.get(), sizeof(uint64_t) * d.ids_used, hipMemcpyDeviceToDevice, s));
      DFX_HIP(hipMemcpyAsync(str_len.get(), d.str_len.get(), sizeof(uint32_t) * d.ids_used, hipMemcpyDeviceToDevice, s));
    }
  } else {
    d.ids_used = d.pool_used = 0;
  }
  const uint64_t hc[DICT_WORDS] = {d.pool_used, d.ids_used, 0, 0};
  DFX_HIP(hipMemcpyAsync(cursors.get(), hc, sizeof(hc), hipMemcpyHostToDevice, s));
  DFX_HIP(hipStreamSynchronize(s));  // hc is a stack buffer; the old arrays are released below
  d.state = dstate; d.hash = hash; d.sid = sid; d.str_off = str_off; d.str_len = str_len; d.pool = pool; d.cursors = cursors;
  d.D.state = (uint32_t*)dstate.get();
  d.D.hash = (uint64_t*)hash.get();
  d.D.sid = (uint64_t*)sid.get();
  d.D.str_off = (uint64_t*)str_off.get();
  d.D.str_len = (uint32_t*)str_len.get();
  d.D.pool = (uint8_t*)pool.get();
  d.D.cursors = (uint64_t*)cursors.get();
  d.D.mask = slots - 1;
  d.D.shift = 64 - slots_log2;
  d.D.id_cap = id_cap;
  d.D.pool_cap = std::max<uint64_t>(pool_cap, 64);
  d.allocated = true;
  if (d.ids_used) DFX_HIP(launch_dict_rebuild(d.D, d.ids_used, s));
  return Status::OK();
}

// ids of one batch's strings; grows the dictionary (ids stay stable) and re-encodes when it overflows
Status AggregateRelation::Impl::dict_encode(DictKey& d, const DeviceColumn& src, int64_t n, DeviceColumn* ids_col) {
  hipStream_t s = ctx().stream;
  Status st;
  auto ids = device_alloc(sizeof(uint64_t) * (size_t)std::max<int64_t>(n, 1), &st);
  if (!ids) return st;
  if (!d.allocated) {
    int lg = opt().dict_capacity_log2 > 0 ? opt().dict_capacity_log2 : 16;
    lg = std::max(4, std::min(lg, 30));
    DFX_RETURN_IF_ERROR(dict_alloc(d, lg, std::max<uint64_t>((uint64_t)src.data_bytes * 2, 1u << 16), false));
  }
  for (int attempt = 0; n > 0; ++attempt) {
    if (attempt > 16) return Status::Err(DFX_INTERNAL_ERROR, "Utf8 key dictionary does not converge");
    DFX_HIP(launch_dict_encode(src.offsets, src.data, n, d.D, d.ids_used, (uint64_t*)ids.get(), s));
    uint64_t hc[DICT_WORDS];
    DFX_HIP(hipMemcpyAsync(hc, d.D.cursors, sizeof(hc), hipMemcpyDeviceToHost, s));
    DFX_HIP(hipStreamSynchronize(s));
    if (hc[DICT_OVERFLOW] == 2) return Status::Err(DFX_INTERNAL_ERROR, "Utf8 key dictionary: slot claim timed out");
    if (hc[DICT_OVERFLOW] == 0) {
      d.ids_used = hc[DICT_IDS];
      d.pool_used = hc[DICT_POOL];
      break;
    }
    // overflow: forget this attempt (its ids were not used yet), grow x4 (slots / ids) and to fit the batch (pool)
    int lg = 64 - d.D.shift;
    const uint64_t want_ids = std::max<uint64_t>(hc[DICT_IDS], d.ids_used + 1);
    while ((1ull << lg) / 2 < want_ids * 2 && lg < 31) ++lg;
    lg = std::min(31, std::max(lg, 64 - d.D.shift + 2));
    const uint64_t want_pool = std::max<uint64_t>(hc[DICT_POOL], d.pool_used + (uint64_t)src.data_bytes) * 2;
    DFX_RETURN_IF_ERROR(dict_alloc(d, lg, std::max<uint64_t>(want_pool, d.D.pool_cap), true));
  }
  ids_col->dtype = DFX_UINT64;
  ids_col->length = n;
  ids_col->null_count = 0;
  ids_col->values = ids.get();
  ids_col->validity = nullptr;
  ids_col->bit_offset = 0;
  ids_col->owners.clear();
  ids_col->owners.push_back(ids);
  return Status::OK();
}

// group ids -> Arrow Utf8 column (offsets + data) on the device
Status AggregateRelation::Impl::dict_emit(const DictKey& d, const uint64_t* ids, int64_t g, DeviceColumn* out) {
  hipStream_t s = ctx().stream;
  Status st;
  auto lens = device_alloc(sizeof(uint32_t) * (size_t)std::max<int64_t>(g, 1), &st);
  if (!lens) return st;
  auto starts = device_alloc(sizeof(uint64_t) * (size_t)(g + 1), &st);
  if (!starts) return st;
  auto tmp = device_alloc(sizeof(uint64_t) * (size_t)(g / 4096 + 4), &st);
  if (!tmp) return st;
  auto offs = device_alloc(sizeof(int32_t) * (size_t)(g + 1), &st);
  if (!offs) return st;
  uint64_t total = 0;
  if (g > 0) {
    DFX_HIP(launch_dict_lengths(ids, g, d.D, (uint32_t*)lens.get(), s));
    DFX_HIP(launch_scan_u32((const uint32_t*)lens.get(), (uint64_t*)starts.get(), g, (uint64_t*)tmp.get(), s));
    DFX_HIP(hipMemcpyAsync(&total, (uint64_t*)starts.get() + g, sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    DFX_HIP(hipStreamSynchronize(s));
  } else {
    DFX_HIP(hipMemsetAsync(starts.get(), 0, sizeof(uint64_t), s));
  }
  if (total > 0x7FFFFFFFull) return Status::Err(DFX_EXECUTION_ERROR, "Utf8 group keys exceed 2 GB (Arrow Utf8 offsets are 32-bit)");
  auto data = device_alloc((size_t)std::max<uint64_t>(total, 8), &st);
  if (!data) return st;
  DFX_HIP(launch_dict_gather(ids, g, d.D, (const uint64_t*)starts.get(), (int32_t*)offs.get(), (uint8_t*)data.get(), s));
  out->dtype = DFX_UTF8;
  out->length = g;
  out->null_count = 0;
  out->values = nullptr;
  out->offsets = (const int32_t*)offs.get();
  out->data = (const uint8_t*)data.get();
  out->data_bytes = (int64_t)total;
  out->owners.clear();
  out->owners.push_back(offs);
  out->owners.push_back(data);
  return Status::OK();
}

Status AggregateRelation::Impl::drain() {
  if (built) return Status::OK();
  ScopedUs t_drain(&counters().agg_drain_us);
  DFX_RETURN_IF_ERROR(ensure_init());
  if (!options.overrides.empty() && input) input->host_stream_options(host_stream_options_of(opt()));  // (its own option set: how a host source below moves batches)
  hipStream_t s = ctx().stream;
  Status st;
  if (kw == 0) {
    memset(&T, 0, sizeof(T));
    ctrl = device_alloc(sizeof(uint32_t) * CTRL_WORDS, &st);
    if (!ctrl) return st;
    DFX_HIP(hipMemsetAsync(ctrl.get(), 0, sizeof(uint32_t) * CTRL_WORDS, s));
    for (int c = (int)chunks.size() - 1; c >= 0; --c) {  // every chunk: batch partials, running state, type tables (chunk 0 last: it stays active)
      activate(c);
      partial = device_alloc(sizeof(uint64_t) * kReduceSlots * kReduceSlotWords, &st);
      if (!partial) return st;
      state = device_alloc(sizeof(uint64_t) * 2 * kMaxAggs, &st);
      if (!state) return st;
      dev_arg_dtype = device_alloc(kMaxAggs, &st);
      if (!dev_arg_dtype) return st;
      dev_func = device_alloc(kMaxAggs, &st);
      if (!dev_func) return st;
      std::vector<uint64_t> hpv((size_t)kReduceSlots * kReduceSlotWords, 0);
      uint64_t* hp = hpv.data();
      uint8_t hd[kMaxAggs], hf[kMaxAggs];
      memset(hd, 0, sizeof(hd));
      memset(hf, 0, sizeof(hf));
      const int a0 = chunks[(size_t)c].a0;
      for (int a = 0; a < na; ++a) {
        for (int sl = 0; sl < kReduceSlots; ++sl) {
          hp[(size_t)sl * kReduceSlotWords + 4 * a] = acc_init[a];
          hp[(size_t)sl * kReduceSlotWords + 4 * a + 2] = ~0ull;
        }
        hd[a] = (uint8_t)arg_dtype[a0 + a];
        hf[a] = (uint8_t)func[a0 + a];
      }
      DFX_HIP(hipMemcpy(partial.get(), hp, sizeof(uint64_t) * hpv.size(), hipMemcpyHostToDevice));  // (blocking: stack / loop-local sources)
      DFX_HIP(hipMemcpy(dev_arg_dtype.get(), hd, sizeof(hd), hipMemcpyHostToDevice));
      DFX_HIP(hipMemcpy(dev_func.get(), hf, sizeof(hf), hipMemcpyHostToDevice));
      DFX_HIP(hipMemsetAsync(state.get(), 0, sizeof(uint64_t) * 2 * kMaxAggs, s));
    }
    T.na = na;
    for (int a = 0; a < na; ++a) {
      T.acc_kind[a] = acc_kind[a];
      T.val_xform[a] = val_xform[a];
      T.acc_init[a] = acc_init[a];
    }
    DFX_HIP(hipStreamSynchronize(s));
  } else {
    int cap_log2 = opt().capacity_log2 > 0 ? opt().capacity_log2 : 21;
    cap_log2 = std::max(6, std::min(cap_log2, 31));
    DFX_RETURN_IF_ERROR(alloc_table(cap_log2, &T, &table_owners, true, &accs_full));
    spill.words = nullptr;
    spill.capacity = 0;
  }
  // one slice per routing window from a resident table, whatever batch width its scan was created with (the result of an
  // aggregate does not depend on it; every slice costs a pass-1 launch)
  if (opt().merge_scan_batches) input->prefer_batch_rows((int64_t)1 << 27);
  for (;;) {
    DeviceBatch b;
    bool has = false;
    DFX_RETURN_IF_ERROR(input->next(&b, &has));
    if (!has) break;
    DFX_RETURN_IF_ERROR(consume_batch(b));
  }
  DFX_RETURN_IF_ERROR(run_held());  // (batches a multi-chunk aggregate was still holding)
  if (kw == 0) {
    uint32_t hc[CTRL_WORDS];
    DFX_RETURN_IF_ERROR(read_ctrl(hc));
    if (hc[CTRL_ERROR]) return error_from_ctrl(hc[CTRL_ERROR]);
  } else {
    DFX_RETURN_IF_ERROR(flush_pass2());
    DFX_RETURN_IF_ERROR(settle_ctrl());
    if (use_partition) {  // the last pass 2 ran after the last snapshot: errors, spilled rows, growth
      uint32_t hc[CTRL_WORDS];
      DFX_RETURN_IF_ERROR(read_ctrl(hc));
      DFX_RETURN_IF_ERROR(handle_ctrl(hc, 0));
    }
  }
  built = true;
  return Status::OK();
}

// ---- output ------------------------------------------------------------------------------------------
static Status upload_small(const void* host, size_t bytes, std::shared_ptr<void>* dev) {
  Status st;
  *dev = device_alloc(bytes ? bytes : 8, &st);
  if (!*dev) return st;
  if (bytes) DFX_HIP(hipMemcpy(dev->get(), host, bytes, hipMemcpyHostToDevice));
  return Status::OK();
}

Status AggregateRelation::Impl::emit_ungrouped(DeviceBatch* out) {  // aggregate.rs:745-784
  uint64_t hs[2 * kMaxAccsTotal];
  memset(hs, 0, sizeof(hs));
  for (int c = 0; c < (int)chunks.size(); ++c) {  // every chunk keeps its own (has-value, bits) pairs
    const void* st_c = c == cur_chunk ? state.get() : chunks[(size_t)c].state.get();
    DFX_HIP(hipMemcpy(hs + 2 * chunks[(size_t)c].a0, st_c, sizeof(uint64_t) * 2 * (size_t)chunks[(size_t)c].n, hipMemcpyDeviceToHost));
  }
  out->num_rows = 1;
  out->columns.clear();
  out->columns.resize(outs.size());
  for (size_t j = 0; j < outs.size(); ++j) {
    const int a = outs[j].acc;
    DeviceColumn& c = out->columns[j];
    c.dtype = outs[j].avg ? outs[j].dtype : out_dtype[a];
    c.length = 1;
    uint64_t bits = hs[2 * a + 1];
    bool has_value = hs[2 * a] != 0;
    if (outs[j].avg) {  // SUM / COUNT (deviation D7); None when nothing was counted
      const uint64_t cntv = hs[2 * (a + 1)] ? hs[2 * (a + 1) + 1] : 0;
      has_value = has_value && cntv != 0;
      bits = has_value ? host_avg_value((uint8_t)outs[j].dtype, bits, cntv) : 0;
    }
    uint8_t raw[8];
    memcpy(raw, &bits, 8);  // little endian: the low bytes are the narrow value
    std::shared_ptr<void> dv, dn;
    DFX_RETURN_IF_ERROR(upload_small(raw, 8, &dv));
    c.values = dv.get();
    c.owners.push_back(dv);
    const bool has = has_value;
    uint8_t vb[8] = {(uint8_t)(has ? 1 : 0), 0, 0, 0, 0, 0, 0, 0};
    DFX_RETURN_IF_ERROR(upload_small(vb, 8, &dn));
    c.validity = (const uint8_t*)dn.get();
    c.null_count = has ? 0 : 1;
    if (!has) c.null_count = 1;
    else c.validity = nullptr;
    c.owners.push_back(dn);
  }
  return Status::OK();
}

// Queues the key column's compaction and download on the side stream when the group count has stopped changing (see EarlyKeys).
Status AggregateRelation::Impl::early_keys_maybe() {
  const uint64_t prev = early_last_occupied;
  early_last_occupied = occupied_known;
  if (early.armed && early.generation == table_generation && early.occupied == occupied_known) return Status::OK();  // still good
  if (!opt().early_keys || !opt().emit_async || !use_partition || kw != 1 || kw_out != 1 || !dicts.empty() || chunks.size() != 1)
    return Status::OK();
  if (occupied_known < 32768 || occupied_known != prev) return Status::OK();  // small results are not worth it; still growing
  early.cancel();
  hipStream_t aux = ctx().aux;
  Status st;
  const int64_t g = (int64_t)occupied_known;
  const int64_t n_slots = (int64_t)T.mask + 2;
  const int64_t n_words = (n_slots + 63) / 64;
  const int64_t n_tiles = (n_slots + kTileRows - 1) / kTileRows;
  const int dt = key_dtype[0];
  // Speculative work: a buffer that cannot be had (memory pressure, the tests' allocation-failure injection) drops the
  // attempt -- the query itself does not need it and must not fail because of it.
  auto mask = device_alloc(sizeof(uint64_t) * (size_t)n_words, &st);
  auto counts = mask ? device_alloc(sizeof(uint32_t) * (size_t)n_tiles, &st) : nullptr;
  auto offsets = counts ? device_alloc(sizeof(uint64_t) * (size_t)(n_tiles + 1), &st) : nullptr;
  auto tmp = offsets ? device_alloc(sizeof(uint64_t) * (size_t)(n_tiles / 4096 + 4), &st) : nullptr;
  auto vals = tmp ? device_alloc((size_t)g * dtype_width(dt), &st) : nullptr;
  early.bytes = (size_t)g * dtype_width(dt);
  if (vals) early.host = pinned_alloc(early.bytes, &st);
  if (vals && early.host && !early.total) early.total = pinned_alloc(sizeof(uint64_t), &st);
  std::shared_ptr<void> dense;  // 4-byte keys: the compacted 8-byte key words before narrowing
  if (vals && dtype_width(dt) != 8) dense = device_alloc(sizeof(uint64_t) * (size_t)g, &st);
  if (!vals || !early.host || !early.total || (dtype_width(dt) != 8 && !dense)) {
    early.host.reset();
    return Status::OK();
  }
  *(uint64_t*)early.total.get() = ~0ull;
  if (!early.done) DFX_HIP(hipEventCreateWithFlags(&early.done, hipEventDisableTiming));
  if (!early.start) DFX_HIP(hipEventCreateWithFlags(&early.start, hipEventDisableTiming));
  early.scratch = {mask, counts, offsets, tmp, vals};
  if (dense) early.scratch.push_back(dense);
  early.keep = table_owners;  // (the side stream reads the key plane: it stays allocated until that has happened, whatever replaces the table)
  early.keep.push_back(ctrl);
  // The side stream starts behind everything queued on the main stream so far: the pool hands out blocks whose previous users may
  // still be queued there.  It is not ordered against what comes LATER: whatever those kernels add to the table makes the final
  // group count differ from `g`, and the copy is dropped.
  DFX_HIP(hipEventRecord(early.start, ctx().stream));
  DFX_HIP(hipStreamWaitEvent(aux, early.start, 0));
  DFX_HIP(launch_table_mask(T, (uint64_t*)mask.get(), (uint32_t*)counts.get(), aux));
  DFX_HIP(launch_scan_u32((const uint32_t*)counts.get(), (uint64_t*)offsets.get(), n_tiles, (uint64_t*)tmp.get(), aux));
  DFX_HIP(hipMemcpyAsync(early.total.get(), (uint64_t*)offsets.get() + n_tiles, sizeof(uint64_t), hipMemcpyDeviceToHost, aux));
  DFX_HIP(launch_fill_u64(T.keys + T.mask + 1, kEmptyKey, 1, aux));  // (as emit_grouped: the sentinel group's key word; always this constant)
  if (dtype_width(dt) == 8) {
    DFX_HIP(launch_compact(T.keys, 8, (const uint64_t*)mask.get(), (const uint64_t*)offsets.get(), n_slots, vals.get(), 0, aux, (uint64_t)g));
  } else {
    DFX_HIP(launch_compact(T.keys, 8, (const uint64_t*)mask.get(), (const uint64_t*)offsets.get(), n_slots, dense.get(), 0, aux, (uint64_t)g));
    DFX_HIP(launch_finalize((const uint64_t*)dense.get(), g, (uint8_t)dt, (uint8_t)VT_RAW, vals.get(), aux));
  }
  // by the copy engine, not by a kernel: pass 1's workgroups take a CU's whole register file, so a copy kernel's waves and a pass-1
  // workgroup cannot share a CU -- measured: the kernel copy made the pass-1 launches it met 0.2 ms longer, more than it saved
  DFX_HIP(hipMemcpyAsync(early.host.get(), vals.get(), early.bytes, hipMemcpyDeviceToHost, aux));
  DFX_HIP(hipEventRecord(early.done, aux));
  early.armed = true;
  early.occupied = occupied_known;
  early.generation = table_generation;
  ++counters().agg_early_keys;
  return Status::OK();
}

Status AggregateRelation::Impl::emit_grouped(DeviceBatch* out, int64_t expected) {  // aggregate.rs:877-951
  ScopedUs t_emit(&counters().agg_emit_us);
  hipStream_t s = ctx().stream;
  const int64_t n_slots = (int64_t)T.mask + 2;
  const int64_t n_words = (n_slots + 63) / 64;
  const int64_t n_tiles = (n_slots + kTileRows - 1) / kTileRows;
  Status st;
  if (!emit_total) {
    emit_total = pinned_alloc(sizeof(uint64_t), &st);
    if (!emit_total) return st;
  }
  uint64_t* total = (uint64_t*)emit_total.get();
  // Round 6: when the key column was copied ahead of time (agg.early_keys) and is still valid -- the same table, the group count it
  // was made for, its own scan's total equal to it: groups are never removed, so the occupancy mask it compacted with IS the
  // table's -- that mask, its tile offsets and the compacted key column on the device are what emit would compute again: reuse
  // them (three kernels and their boundaries less behind the query's last pass 2: ~0.1 ms of a 4 ms step).
  std::shared_ptr<void> mask, counts, offsets, tmp, early_keys_dev;
  bool reuse_early = false;
  if (expected >= 0 && early.armed && early.generation == table_generation && early.occupied == (uint64_t)expected &&
      early.scratch.size() >= 5 && kw_out == 1 && dicts.empty()) {
    // (its kernels and copies ran on the side stream while the scan went on: long finished -- unless the copy engine stalled)
    if (early.ready() && *(const uint64_t*)early.total.get() == (uint64_t)expected && early.bytes == (size_t)expected * dtype_width(key_dtype[0])) {
      mask = early.scratch[0];
      offsets = early.scratch[2];
      early_keys_dev = early.scratch[4];
      reuse_early = true;
      *total = (uint64_t)expected;
      ++counters().agg_emit_reused_early;
    }
  }
  if (!reuse_early) {
    mask = device_alloc(sizeof(uint64_t) * (size_t)n_words, &st);
    if (!mask) return st;
    counts = device_alloc(sizeof(uint32_t) * (size_t)n_tiles, &st);
    if (!counts) return st;
    offsets = device_alloc(sizeof(uint64_t) * (size_t)(n_tiles + 1), &st);
    if (!offsets) return st;
    tmp = device_alloc(sizeof(uint64_t) * (size_t)(n_tiles / 4096 + 4), &st);
    if (!tmp) return st;
    DFX_HIP(launch_table_mask(T, (uint64_t*)mask.get(), (uint32_t*)counts.get(), s));
    DFX_HIP(launch_scan_u32((const uint32_t*)counts.get(), (uint64_t*)offsets.get(), n_tiles, (uint64_t*)tmp.get(), s));
    // The group count is already on the host (CTRL_OCCUPIED of the last control-block check), so the compaction kernels
    // are queued without waiting for the scan's total; the total comes back with the final synchronisation and must
    // agree.  `expected < 0`: second attempt after a disagreement, with the scan's own count (one extra round trip).
    *total = ~0ull;
    DFX_HIP(hipMemcpyAsync(total, (uint64_t*)offsets.get() + n_tiles, sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    if (expected < 0) DFX_HIP(hipStreamSynchronize(s));
  }
  const int64_t g = expected < 0 ? (int64_t)*total : expected;
  out->num_rows = g;
  out->columns.clear();
  out->columns.resize((size_t)kw_out + outs.size());
  auto dense = device_alloc(sizeof(uint64_t) * (size_t)std::max<int64_t>(g, 1), &st);
  if (!dense) return st;
  // the sentinel group's key word is not stored in the table: patch slot `cap` before compaction
  if (kw == 1 && !reuse_early) DFX_HIP(launch_fill_u64(T.keys + T.mask + 1, kEmptyKey, 1, s));
  for (int k = 0; k < kw_out; ++k) {  // (padding words beyond kw_out are constants: not part of the result)
    const uint64_t* plane = T.keys + (size_t)k * T.stride;
    const int dt = key_dtype[k];
    DeviceColumn& c = out->columns[k];
    c.dtype = dt;
    c.length = g;
    if (reuse_early) {  // (one key column, no dictionary: the side stream compacted -- and narrowed -- it already)
      c.values = early_keys_dev.get();
      c.owners.push_back(early_keys_dev);
      continue;
    }
    const DictKey* dk = nullptr;
    for (const DictKey& d : dicts)
      if (d.key == k) dk = &d;
    if (dk || dtype_width(dt) != 8)
      DFX_HIP(launch_compact(plane, 8, (const uint64_t*)mask.get(), (const uint64_t*)offsets.get(), n_slots, dense.get(), 0, s, (uint64_t)g));
    if (dk) {  // ids -> Arrow Utf8
      DFX_RETURN_IF_ERROR(dict_emit(*dk, (const uint64_t*)dense.get(), g, &c));
      continue;
    }
    auto vals = device_alloc((size_t)std::max<int64_t>(g, 1) * dtype_width(dt), &st);
    if (!vals) return st;
    if (dtype_width(dt) == 8) {  // the plane's words ARE the column: compact straight into it (one kernel and 16 bytes per group less)
      DFX_HIP(launch_compact(plane, 8, (const uint64_t*)mask.get(), (const uint64_t*)offsets.get(), n_slots, vals.get(), 0, s, (uint64_t)g));
    } else {
      DFX_HIP(launch_finalize((const uint64_t*)dense.get(), g, (uint8_t)dt, (uint8_t)VT_RAW, vals.get(), s));
    }
    c.values = vals.get();
    c.owners.push_back(vals);
  }
  for (size_t j = 0; j < outs.size(); ++j) {
    const int a = outs[j].acc;
    const int dt = outs[j].avg ? outs[j].dtype : out_dtype[a];
    DeviceColumn& c = out->columns[(size_t)kw_out + j];
    c.dtype = dt;
    c.length = g;
    auto vals = device_alloc((size_t)std::max<int64_t>(g, 1) * dtype_width(dt), &st);
    if (!vals) return st;
    const bool raw8 = !outs[j].avg && dtype_width(dt) == 8 && (val_xform_all[a] == VT_RAW || val_xform_all[a] == VT_COUNT_VALID);  // SUM(f64 / i64), COUNT: no image to undo
    DFX_HIP(launch_compact(accs_full + (size_t)a * T.stride, 8, (const uint64_t*)mask.get(), (const uint64_t*)offsets.get(), n_slots,
                           raw8 ? vals.get() : dense.get(), 0, s, (uint64_t)g));
    if (raw8) {
    } else if (!outs[j].avg) {
      DFX_HIP(launch_finalize((const uint64_t*)dense.get(), g, (uint8_t)dt, val_xform_all[a], vals.get(), s));
    } else {  // SUM plane / COUNT plane (deviation D7); groups that counted nothing are null
      auto dense_cnt = device_alloc(sizeof(uint64_t) * (size_t)std::max<int64_t>(g, 1), &st);
      if (!dense_cnt) return st;
      auto valid = device_alloc(sizeof(uint64_t) * (size_t)((g + 63) / 64 + 1), &st);
      if (!valid) return st;
      auto nulls = device_alloc(sizeof(uint64_t), &st);
      if (!nulls) return st;
      DFX_HIP(hipMemsetAsync(nulls.get(), 0, sizeof(uint64_t), s));
      DFX_HIP(launch_compact(accs_full + (size_t)(a + 1) * T.stride, 8, (const uint64_t*)mask.get(), (const uint64_t*)offsets.get(),
                             n_slots, dense_cnt.get(), 0, s, (uint64_t)g));
      DFX_HIP(launch_finalize_avg((const uint64_t*)dense.get(), (const uint64_t*)dense_cnt.get(), g, (uint8_t)dt, vals.get(),
                                  (uint64_t*)valid.get(), (uint64_t*)nulls.get(), s));
      uint64_t n_null = 0;
      DFX_HIP(hipMemcpyAsync(&n_null, nulls.get(), sizeof(uint64_t), hipMemcpyDeviceToHost, s));
      DFX_HIP(hipStreamSynchronize(s));
      if (n_null) {
        c.validity = (const uint8_t*)valid.get();
        c.null_count = (int64_t)n_null;
        c.owners.push_back(valid);
      }
    }
    c.values = vals.get();
    c.owners.push_back(vals);
  }
  DFX_HIP(hipStreamSynchronize(s));
  if ((int64_t)*total != g) {
    if (expected < 0) return Status::Err(DFX_INTERNAL_ERROR, "group count changed during emit");
    return emit_grouped(out, -1);  // the host's count was stale: redo with the table's own
  }
  if (early.armed) {  // the key column copied ahead of time: valid iff it was made from this table with this many groups -- and has arrived
    DeviceColumn& kc = out->columns[0];
    if (!early.ready()) {
      ++counters().agg_early_keys_late;
    } else if (early.generation == table_generation && early.occupied == (uint64_t)g && *(const uint64_t*)early.total.get() == (uint64_t)g &&
        early.bytes == (size_t)g * dtype_width(kc.dtype) && kc.values != nullptr) {
      kc.host_values = early.host;
      kc.host_values_of = kc.values;
      kc.host_bytes = early.bytes;
      ++counters().agg_early_keys_used;
    }
    early.drop();
  }
  return Status::OK();
}

// ---- public class -----------------------------------------------------------------------------------
AggregateRelation::AggregateRelation(SchemaInfo schema, std::unique_ptr<Relation> input,
                                     std::vector<dfx_runtime_expr> group, std::vector<dfx_runtime_expr> aggr, OptionOverrides options)
    : schema_(std::move(schema)), impl_(new Impl()) {
  Impl& m = *impl_;
  m.options.overrides = std::move(options);
  m.group = std::move(group);
  for (const dfx_runtime_expr& e : aggr) {  // AVG(x) -> SUM(x), COUNT(x)
    Impl::OutAgg o;
    o.acc = (int)m.aggr.size();
    o.avg = e.is_aggregate && e.agg_func == AGG_AVG;
    o.dtype = e.agg_func == AGG_COUNT ? (int)DFX_UINT64 : e.agg_type;
    o.name = e.name;
    m.outs.push_back(o);
    if (o.avg) {
      dfx_runtime_expr sum = e, cnt = e;
      sum.agg_func = AGG_SUM;
      cnt.agg_func = AGG_COUNT;
      cnt.agg_type = DFX_UINT64;
      m.aggr.push_back(sum);
      m.aggr.push_back(cnt);
    } else {
      m.aggr.push_back(e);
    }
  }
  // Filter -> Aggregate fusion (K7)
  if (input->kind() == REL_FILTER) {
    FilterRelation* f = static_cast<FilterRelation*>(input.get());
    // the predicate joins the scan's fused program only if it fits next to the keys and at least one argument; otherwise
    // (or when the Filter itself needed several programs) the Filter stays a relation of its own below the aggregate
    bool fits = f->single_program();
    if (fits && !f->predicate().is_aggregate) {
      // EVERY accumulator must fit beside the predicate and the keys on its own (setup() splits the accumulators into
      // chunks down to one per program, never below): one that does not would fail the whole query with NotImplemented
      // where the reference -- which has no such limit -- runs it; un-fused, its program holds keys + argument only
      for (size_t a = 0; a < std::max<size_t>(m.aggr.size(), 1) && fits; ++a) {
        ProgramBuilder trial(f->input()->schema());
        uint8_t opnd = kNoOperand;
        int dt = 0;
        Status tst = trial.add(f->predicate(), f->predicate().root, &opnd, &dt);
        for (size_t k = 0; k < m.group.size() && tst.ok(); ++k)
          if (!m.group[k].is_aggregate) tst = trial.add(m.group[k], m.group[k].root, &opnd, &dt);
        if (tst.ok() && a < m.aggr.size() && m.aggr[a].is_aggregate && m.aggr[a].agg_arg >= 0) tst = trial.add(m.aggr[a], m.aggr[a].agg_arg, &opnd, &dt);
        if (program_limit_error(tst)) fits = false;
      }
    }
    if (fits && !f->predicate().is_aggregate) {
      m.has_pred = true;
      m.pred = f->predicate();
      std::unique_ptr<Relation> inner = f->release_input();
      input = std::move(inner);
    }
  }
  m.input = std::move(input);
  m.deferred = m.setup(m.input->schema());
  if (m.deferred.ok()) {  // projection push-down: predicate, key and argument columns only (incl. the Utf8 key sources)
    std::vector<char> needed(m.input->schema().fields.size(), 0);
    for (int ci : m.builder->columns())
      if (ci >= 0 && ci < (int)needed.size()) needed[ci] = 1;
    for (const Impl::Chunk& ch : m.chunks)  // every chunk of accumulators reads its own columns from the same batches
      if (ch.builder)
        for (int ci : ch.builder->columns())
          if (ci >= 0 && ci < (int)needed.size()) needed[ci] = 1;
    for (const Impl::DictKey& d : m.dicts)
      if (d.src_col >= 0 && d.src_col < (int)needed.size()) needed[d.src_col] = 1;
    m.input->require_columns(needed);
  }
  // output schema: group columns then aggregates (aggregate.rs:894-949); context.rs:185 passes
  // Schema::empty(), so derive names/types from the expressions when none is given
  SchemaInfo derived;
  for (size_t k = 0; k < m.group.size(); ++k) {
    Field f;
    f.name = m.group[k].name;
    f.dtype = k < m.key_out_dtype.size() && m.key_out_dtype[k] ? m.key_out_dtype[k] : m.group[k].dtype;
    f.nullable = false;
    derived.fields.push_back(f);
  }
  for (const Impl::OutAgg& o : m.outs) {
    Field f;
    f.name = o.name;
    f.dtype = o.dtype;
    f.nullable = true;
    derived.fields.push_back(f);
  }
  if (schema_.fields.size() == derived.fields.size()) {
    for (size_t i = 0; i < derived.fields.size(); ++i) {
      derived.fields[i].name = schema_.fields[i].name;
    }
  }
  schema_ = derived;
}

AggregateRelation::~AggregateRelation() {}

void AggregateRelation::explain(std::string* out, int depth) const {
  const Impl& m = *impl_;
  if (!m.deferred.ok()) {
    explain_line(out, depth, "Aggregate: error deferred to next(): " + m.deferred.msg);
  } else {
    const DevProgram& P = m.builder->program();
    const char* shape = "SSA interpreter";
    if (m.kw == 0) {
      if (sig_matches<SigCountPred2F64>(P, m.fast, 0, m.na, m.acc_kind, m.val_xform)) shape = "static shape CountPred2F64";
      else if (sig_matches<SigSumCountPred2F64>(P, m.fast, 0, m.na, m.acc_kind, m.val_xform)) shape = "static shape SumCountPred2F64";
      else if (m.opt().plan != 0 && scan_plan_shape_ok(P, m.fast, 0, m.na, m.val_xform)) shape = "scan plan where a batch has nulls or 4-byte columns (PlanPolicy: range tests on value images), else column-op-literal shape (FastPolicy)";
      else if (m.fast.valid) shape = "column-op-literal shape (FastPolicy; interpreter when a batch has nulls)";
    } else {
      if (m.kw == 1 && sig_matches<SigKeySumPred2F64>(P, m.fast, 1, m.na, m.acc_kind, m.val_xform)) shape = "static shape KeySumPred2F64";
      else if (m.kw == 1 && sig_matches<SigKeyAffSumPred2F64>(P, m.fast, 1, m.na, m.acc_kind, m.val_xform)) shape = "static shape KeyAffSumPred2F64 (pass 1 of the partitioned strategy)";
      else if (m.kw == 1 && sig_matches<SigKeySum>(P, m.fast, 1, m.na, m.acc_kind, m.val_xform)) shape = "static shape KeySum";
      else if (m.kw == 2 && sig_matches<SigQ1>(P, m.fast, 2, m.na, m.acc_kind, m.val_xform)) shape = "static shape Q1";
      else if (m.opt().plan != 0 && scan_plan_shape_ok(P, m.fast, m.kw, m.na, m.val_xform))
        shape = m.kw == 1 ? "scan plan (PlanPolicy: range tests on value images, 4-byte columns widened, nulls by arrow's rule; every kernel of the partitioned strategy, the other strategies where a batch has nulls or 4-byte columns)"
                          : "scan plan where a batch has nulls or 4-byte columns (PlanPolicy), else column-op-literal shape (FastPolicy)";
      else if (m.fast.valid) shape = "column-op-literal shape (FastPolicy; interpreter when a batch has nulls)";
    }
    std::string text = strfmt("Aggregate: %d keys%s, %d accumulators", m.kw_out, m.kw != m.kw_out ? " (as 8 key words)" : "", m.na_total);
    if (m.chunks.size() > 1) text += strfmt(" in %d chunks of <= %d (one fused program each, the same table)", (int)m.chunks.size(), kMaxAggs);
    const bool plan_fuses = m.opt().plan != 0 && m.opt().fast != 0 && scan_plan_shape_ok(P, m.fast, m.kw, m.na, m.val_xform);
    text += !m.has_pred ? ", no predicate" : plan_fuses ? ", Filter below fused into the scan (batches with nulls too: the scan plan judges a null by arrow's comparison rule and counts every surviving slot as valid)"
                                                        : ", Filter below fused into the scan (un-fused for batches with nulls in its columns)";
    text += ", " + explain_program(P) + ", " + shape;
    if (m.kw == 0) text += ", ungrouped reduce (64 partial copies + fold)";
    else if (m.kw == 1) text += ", strategy chosen on the first 2^18 rows: register accumulators (<= 8 groups) / LDS front cache (<= 8192) / table / "
                               "partitioned (>= 16384 groups: pass 1 routes rows to table blocks, pass 2 aggregates blocks in LDS)";
    else text += ", strategy chosen on the first 2^18 rows: register accumulators (<= 8 groups) / LDS front cache (<= 8192) / table";
    if (m.shared_operand())
      text += strfmt("; %d aggregates of ONE operand: the partitioned strategy routes 12-byte rows {hash image, raw operand} while keys are "
                     "narrow and batches have no nulls; pass 2 runs once per accumulator plane with that aggregate's transform "
                     "(agg.shared_planes; 0: one pass 2 over 4096-slot blocks that hold every plane)", m.na);
    if (m.pair_mode) text += m.pair_is_planes ? "; ran the one-value pass 1 with a pass 2 per accumulator plane" : "; ran the pair scan (both operands routed by one scan, a pass 2 per accumulator plane: agg.pair_scan)";
    if (m.split_ready && !m.split_done && !m.split_is_shared && !m.pair_mode)
      text += strfmt("; %d aggregates of different operands: if the calibration slice chooses the partitioned strategy, one scan per "
                     "aggregate (its own fused program and accumulator plane over the same keys: 12-byte routed rows, the one-aggregate "
                     "kernels; agg.split_aggregates)", m.na_total);
    if (m.split_done) text += strfmt("; ran one scan per aggregate (%d scans per batch: agg.split_aggregates)", (int)m.chunks.size());
    if (!m.dicts.empty()) text += strfmt(", %d Utf8 keys dictionary-encoded on the device", (int)m.dicts.size());
    if (m.built && m.kw > 0)  // after the input was drained: what actually ran
      text += strfmt("; ran %lld rows: %s, %llu of 2^%d table slots occupied", (long long)m.rows_seen,
                     m.use_partition ? "partitioned" : (m.lds_enabled && m.occupied_known <= 8 && m.opt().fewgroup) ? "few groups (register accumulators or LDS front cache)"
                                     : m.lds_enabled ? "LDS front cache + table" : "table (global atomics)",
                     (unsigned long long)m.occupied_known, 64 - m.T.shift);
    else if (m.built)
      text += strfmt("; ran %lld rows", (long long)m.rows_seen);
    explain_line(out, depth, text);
  }
  if (m.input) m.input->explain(out, depth + 1);
}

Status AggregateRelation::next(DeviceBatch* out, bool* has) {
  *has = false;
  Impl& m = *impl_;
  if (m.done) return Status::OK();  // end_of_results (aggregate.rs:616-618)
  m.done = true;
  if (!m.deferred.ok()) return m.deferred;
  if (m.group.empty() && m.aggr.empty())
    return Status::Err(DFX_INTERNAL_ERROR, "assertion failed: record batch needs at least one column");
  DFX_RETURN_IF_ERROR(m.drain());
  if (m.kw == 0) DFX_RETURN_IF_ERROR(m.emit_ungrouped(out));
  // (Utf8 keys: dict_emit indexes the dictionary with the compacted ids before the scan's total could contradict the host's
  // count -- the table's own count first, one round trip more)
  else DFX_RETURN_IF_ERROR(m.emit_grouped(out, (m.opt().emit_async && m.dicts.empty()) ? (int64_t)m.occupied_known : -1));
  *has = true;
  return Status::OK();
}

// ---- multi-GPU partial exchange ---------------------------------------------------------------------
Status AggregateRelation::partial_build(int world, int* n_words, int64_t* counts) {
  if (!impl_->dicts.empty())
    return Status::Err(DFX_NOT_IMPLEMENTED, "multi-GPU exchange of Utf8 GROUP BY keys (dictionary ids are rank-local)");
  Impl& m = *impl_;
  if (!m.deferred.ok()) return m.deferred;
  if (m.kw == 0) return Status::Err(DFX_NOT_IMPLEMENTED, "partial exchange is for GROUP BY aggregates");
  if (world < 1 || world > 1024) return Status::Err(DFX_GENERAL, "world must be in 1..1024");
  DFX_RETURN_IF_ERROR(m.partial_view_check());  // (by accumulators, not by chunks: the drain may re-chunk -- one scan per aggregate)
  DFX_RETURN_IF_ERROR(m.drain());
  hipStream_t s = ctx().stream;
  Status st;
  auto dc = device_alloc(sizeof(uint64_t) * (size_t)world, &st);
  if (!dc) return st;
  DFX_HIP(hipMemsetAsync(dc.get(), 0, sizeof(uint64_t) * (size_t)world, s));
  DFX_HIP(launch_partial_count(m.T, world, (uint64_t*)dc.get(), s));
  m.export_counts.assign((size_t)world, 0);
  DFX_HIP(hipMemcpyAsync(m.export_counts.data(), dc.get(), sizeof(uint64_t) * (size_t)world, hipMemcpyDeviceToHost, s));
  DFX_HIP(hipStreamSynchronize(s));
  for (int r = 0; r < world; ++r) counts[r] = (int64_t)m.export_counts[r];
  *n_words = m.kw + m.na_total;
  return Status::OK();
}

Status AggregateRelation::partial_export(void* dst_device, int64_t dst_words) {
  Impl& m = *impl_;
  if (m.export_counts.empty()) return Status::Err(DFX_GENERAL, "partial_build must precede partial_export");
  std::vector<int64_t> counts(m.export_counts.begin(), m.export_counts.end());
  return partial_export_with(counts, dst_device, dst_words, true, /*all_planes=*/true);
}

// the count step of partial_build with the counts left on the device: d_counts[0, world) = groups per destination
// rank, d_counts[world, 2 world) = scratch for the counts received from the peers
Status AggregateRelation::partial_count_device(int world, int* n_words, uint64_t** d_counts, std::shared_ptr<void>* owner) {
  if (!impl_->dicts.empty())
    return Status::Err(DFX_NOT_IMPLEMENTED, "multi-GPU exchange of Utf8 GROUP BY keys (dictionary ids are rank-local)");
  Impl& m = *impl_;
  if (!m.deferred.ok()) return m.deferred;
  if (m.kw == 0) return Status::Err(DFX_INTERNAL_ERROR, "partial_count_device is for GROUP BY aggregates");
  if (world < 1 || world > 1024) return Status::Err(DFX_GENERAL, "world must be in 1..1024");
  DFX_RETURN_IF_ERROR(m.partial_view_check());
  DFX_RETURN_IF_ERROR(m.drain());
  hipStream_t s = ctx().stream;
  Status st;
  *owner = device_alloc(sizeof(uint64_t) * (size_t)world * 2, &st);
  if (!*owner) return st;
  DFX_HIP(hipMemsetAsync(owner->get(), 0, sizeof(uint64_t) * (size_t)world * 2, s));
  DFX_HIP(launch_partial_count(m.T, world, (uint64_t*)owner->get(), s));
  *d_counts = (uint64_t*)owner->get();
  *n_words = m.kw + m.na_total;
  m.export_counts.assign((size_t)world, 0);  // (filled by partial_export_with)
  return Status::OK();
}

Status AggregateRelation::partial_export_with(const std::vector<int64_t>& counts, void* dst_device, int64_t dst_words, bool sync, bool all_planes) {
  Impl& m = *impl_;
  const int world = (int)counts.size();
  // all_planes: the public partial_* path -- every accumulator in one row, whatever chunking the drain installed;
  // otherwise the ACTIVE chunk's planes (the in-library exchange walks the chunks itself)
  const DevTable Tv = all_planes ? m.full_view(m.T, m.accs_full) : m.T;
  const int na_v = all_planes ? m.na_total : m.na;
  m.export_counts.assign(counts.begin(), counts.end());
  std::vector<uint64_t> base((size_t)world, 0);
  uint64_t total = 0;
  for (int r = 0; r < world; ++r) {
    base[r] = total;
    total += m.export_counts[r];
  }
  if ((uint64_t)dst_words < total * (uint64_t)(m.kw + na_v))
    return Status::Err(DFX_GENERAL, "partial export buffer too small");
  hipStream_t s = ctx().stream;
  Status st;
  auto dbase = device_alloc(sizeof(uint64_t) * (size_t)world * 3, &st);
  if (!dbase) return st;
  uint64_t* d = (uint64_t*)dbase.get();
  {  // bucket bases, bucket counts, zeroed cursors: one blocking copy of 3 x world words (the host vector dies with this scope)
    std::vector<uint64_t> hw((size_t)world * 3, 0);
    for (int r = 0; r < world; ++r) {
      hw[(size_t)r] = base[(size_t)r];
      hw[(size_t)world + r] = m.export_counts[(size_t)r];
    }
    DFX_HIP(hipMemcpy(d, hw.data(), sizeof(uint64_t) * hw.size(), hipMemcpyHostToDevice));
  }
  if (m.kw == 1) DFX_HIP(launch_fill_u64(m.T.keys + m.T.mask + 1, kEmptyKey, 1, s));
  DFX_HIP(launch_partial_scatter(Tv, world, d, d + world, d + 2 * world, (uint64_t*)dst_device, s));
  if (sync) {
    DFX_HIP(hipStreamSynchronize(s));
  } else {
    m.table_owners.push_back(dbase);  // the scatter kernel is still queued: keep its base / count words alive
  }
  return Status::OK();
}

// ---- the in-library exchange, piece by piece (dfx_exchange.cpp drives the collectives between them) ---------------------
int AggregateRelation::exchange_chunks() const { return std::max<int>(1, (int)impl_->chunks.size()); }
int AggregateRelation::exchange_chunk_words(int c) const {
  const Impl& m = *impl_;
  return m.kw + (m.chunks.empty() ? m.na : m.chunks[(size_t)c].n);
}
int AggregateRelation::exchange_dicts() const { return (int)impl_->dicts.size(); }

Status AggregateRelation::exchange_drain() {
  Impl& m = *impl_;
  if (!m.deferred.ok()) return m.deferred;
  if (m.kw == 0) return Status::Err(DFX_INTERNAL_ERROR, "exchange_drain is for GROUP BY aggregates");
  return m.drain();
}

// the groups of the drained table as the host knows them (the control block's count after the drain's last check) + the
// sentinel group's slot: what partial_count can find at most
uint64_t AggregateRelation::exchange_group_bound() const { return impl_->occupied_known + 1; }

Status AggregateRelation::exchange_count(int world, uint64_t* d_counts) {
  Impl& m = *impl_;
  if (world < 1 || world > 1024) return Status::Err(DFX_GENERAL, "world must be in 1..1024");
  hipStream_t s = ctx().stream;
  DFX_HIP(hipMemsetAsync(d_counts, 0, sizeof(uint64_t) * (size_t)world, s));
  DFX_HIP(launch_partial_count(m.T, world, d_counts, s));
  return Status::OK();
}

Status AggregateRelation::exchange_export_chunk(int c, const std::vector<int64_t>& counts, void* dst_device, int64_t dst_words) {
  Impl& m = *impl_;
  if (m.chunks.size() > 1) m.activate(c);
  return partial_export_with(counts, dst_device, dst_words, /*sync=*/false, /*all_planes=*/false);
}

Status AggregateRelation::exchange_import_begin(uint64_t total_groups) {
  Impl& m = *impl_;
  if (!m.built) return Status::Err(DFX_GENERAL, "the input must be drained before the import");
  const int cap_log2 = std::max(10, ceil_log2(4 * (total_groups + 1)));
  if (cap_log2 > 31) return Status::Err(DFX_EXECUTION_ERROR, "GROUP BY table would exceed 2^31 slots");
  m.import_owners.clear();
  m.import_accs_full = nullptr;
  if (m.chunks.size() > 1) m.activate(0);
  m.import_keep = {m.ctrl, m.stats};  // the OLD table's control block stays readable: later chunks are still exported from it
  DFX_RETURN_IF_ERROR(m.alloc_table(cap_log2, &m.import_T, &m.import_owners, true, &m.import_accs_full));
  return Status::OK();
}

Status AggregateRelation::exchange_import_chunk(int c, const void* src_device, const int64_t* counts, int n_buckets) {
  Impl& m = *impl_;
  hipStream_t s = ctx().stream;
  DevRows no_spill;
  no_spill.words = nullptr;
  no_spill.capacity = 0;
  // chunk c's planes of the NEW table; the first chunk inserts the keys, the others find them
  const DevTable Tc = m.chunks.size() > 1 ? m.view_of(m.import_T, m.import_accs_full, c) : m.import_T;
  const int nw = exchange_chunk_words(c);
  uint64_t off = 0;
  for (int b = 0; b < n_buckets; ++b) {
    if (counts[b] > 0)
      DFX_HIP(launch_merge_bucket((const uint64_t*)src_device + (size_t)nw * off, (uint64_t)counts[b], Tc, no_spill, s));
    off += (uint64_t)counts[b];
  }
  return Status::OK();
}

Status AggregateRelation::exchange_import_finish() {
  Impl& m = *impl_;
  DFX_HIP(hipStreamSynchronize(ctx().stream));
  // the table is replaced: a key column copied ahead of time (agg.early_keys) was made from the OLD table's slot order and its
  // side-stream kernels may still read the old planes -- wait for them, drop the copy, and make any later validity check fail
  m.early.cancel();
  ++m.table_generation;
  m.early_last_occupied = ~0ull;
  m.T = m.import_T;
  m.accs_full = m.import_accs_full;
  m.table_owners = m.import_owners;
  m.import_owners.clear();
  m.import_keep.clear();
  m.export_counts.clear();
  if (m.chunks.size() > 1) m.activate(0);
  uint32_t hc[CTRL_WORDS];
  DFX_RETURN_IF_ERROR(m.read_ctrl(hc));
  if (hc[CTRL_ERROR]) return error_from_ctrl(hc[CTRL_ERROR]);
  m.occupied_known = hc[CTRL_OCCUPIED];
  return Status::OK();
}

// the strings of dictionary d in local-id order (lengths + bytes back to back)
Status AggregateRelation::exchange_dict_local(int d, std::vector<uint32_t>* lens, std::vector<uint8_t>* pool) {
  Impl& m = *impl_;
  const Impl::DictKey& k = m.dicts[(size_t)d];
  lens->assign((size_t)k.ids_used, 0);
  pool->clear();
  if (!k.allocated || k.ids_used == 0) return Status::OK();
  std::vector<uint64_t> offs((size_t)k.ids_used);
  std::vector<uint8_t> raw((size_t)k.pool_used);
  DFX_HIP(hipStreamSynchronize(ctx().stream));
  DFX_HIP(hipMemcpy(lens->data(), k.D.str_len, sizeof(uint32_t) * lens->size(), hipMemcpyDeviceToHost));
  DFX_HIP(hipMemcpy(offs.data(), k.D.str_off, sizeof(uint64_t) * offs.size(), hipMemcpyDeviceToHost));
  if (!raw.empty()) DFX_HIP(hipMemcpy(raw.data(), k.D.pool, raw.size(), hipMemcpyDeviceToHost));
  size_t total = 0;
  for (uint32_t l : *lens) total += l;
  pool->reserve(total);
  for (size_t i = 0; i < lens->size(); ++i) {  // the pool is filled by atomics: put the strings in id order
    if (offs[i] + (*lens)[i] > raw.size()) return Status::Err(DFX_INTERNAL_ERROR, "Utf8 key dictionary: string outside the pool");
    pool->insert(pool->end(), raw.begin() + (ptrdiff_t)offs[i], raw.begin() + (ptrdiff_t)(offs[i] + (*lens)[i]));
  }
  return Status::OK();
}

// installs the GLOBAL dictionary (strings by global id: lens + bytes back to back) as dictionary d and rewrites the key
// plane of that GROUP BY column: local id -> remap[local id].  The table is not probed again before the exchange scatters
// it (count / scatter walk the slots), and what the import builds is keyed by global ids from the start.
Status AggregateRelation::exchange_dict_globalise(int d, const std::vector<uint32_t>& lens, const std::vector<uint8_t>& pool,
                                                   const std::vector<uint64_t>& remap) {
  Impl& m = *impl_;
  Impl::DictKey& k = m.dicts[(size_t)d];
  hipStream_t s = ctx().stream;
  Status st;
  if (!remap.empty()) {
    auto dremap = device_alloc(sizeof(uint64_t) * remap.size(), &st);
    if (!dremap) return st;
    DFX_HIP(hipMemcpy(dremap.get(), remap.data(), sizeof(uint64_t) * remap.size(), hipMemcpyHostToDevice));
    uint64_t* plane = m.T.keys + (uint64_t)k.key * m.T.stride;
    DFX_HIP(launch_dict_remap_plane(plane, m.T.mask + 2, (const uint64_t*)dremap.get(), (uint64_t)remap.size(), s));
    DFX_HIP(hipStreamSynchronize(s));  // dremap dies with this scope
  }
  const uint64_t g = lens.size();
  int lg = 4;
  while ((1ull << lg) / 2 < std::max<uint64_t>(g, 1) && lg < 31) ++lg;
  k.ids_used = k.pool_used = 0;
  DFX_RETURN_IF_ERROR(m.dict_alloc(k, lg, std::max<uint64_t>(pool.size(), 64), false));
  std::vector<uint64_t> offs((size_t)g);
  uint64_t at = 0;
  for (size_t i = 0; i < (size_t)g; ++i) {
    offs[i] = at;
    at += lens[i];
  }
  if (g) {
    DFX_HIP(hipMemcpy(k.D.str_len, lens.data(), sizeof(uint32_t) * (size_t)g, hipMemcpyHostToDevice));
    DFX_HIP(hipMemcpy(k.D.str_off, offs.data(), sizeof(uint64_t) * (size_t)g, hipMemcpyHostToDevice));
    if (!pool.empty()) DFX_HIP(hipMemcpy(k.D.pool, pool.data(), pool.size(), hipMemcpyHostToDevice));
  }
  k.ids_used = g;
  k.pool_used = pool.size();
  const uint64_t hc[DICT_WORDS] = {k.pool_used, k.ids_used, 0, 0};
  DFX_HIP(hipMemcpy(k.D.cursors, hc, sizeof(hc), hipMemcpyHostToDevice));
  return Status::OK();
}

Status AggregateRelation::ungrouped_select_chunk(int c) {
  Impl& m = *impl_;
  if (c < 0 || c >= exchange_chunks()) return Status::Err(DFX_GENERAL, "no such chunk");
  if (m.chunks.size() > 1) m.activate(c);
  return Status::OK();
}

static uint64_t host_wrap_to(uint8_t t, uint64_t x) {  // == wrap_to (dfx_kernels_inl.hpp)
  switch (t) {
    case DFX_INT8: return (uint64_t)(int64_t)(int8_t)x;
    case DFX_INT16: return (uint64_t)(int64_t)(int16_t)x;
    case DFX_INT32: return (uint64_t)(int64_t)(int32_t)x;
    case DFX_UINT8: return (uint64_t)(uint8_t)x;
    case DFX_UINT16: return (uint64_t)(uint16_t)x;
    case DFX_UINT32: return (uint64_t)(uint32_t)x;
    default: return x;
  }
}

// ---- ungrouped aggregates across ranks ---------------------------------------------------------------
bool AggregateRelation::is_ungrouped() const { return impl_->deferred.ok() && impl_->group.empty(); }
Status AggregateRelation::ungrouped_state_begin() {
  Impl& m = *impl_;
  if (!m.deferred.ok()) return m.deferred;
  if (m.group.empty() && m.aggr.empty())
    return Status::Err(DFX_INTERNAL_ERROR, "assertion failed: record batch needs at least one column");
  return m.drain();
}
int AggregateRelation::ungrouped_state_words() const { return 2 * kMaxAggs; }
const void* AggregateRelation::ungrouped_state_device() const { return impl_->state.get(); }

// AccumulatorSet::accumulate_scalar (aggregate.rs:107-145,176-214,245-283) between the ranks' scalars, folded in rank
// order: the same arms as the device's batch fold (k_reduce_fold)
Status AggregateRelation::ungrouped_state_merge(const uint64_t* all, int world, int rank) {
  (void)rank;
  Impl& m = *impl_;
  uint64_t out[2 * kMaxAggs];
  memset(out, 0, sizeof(out));
  for (int a = 0; a < m.na; ++a) {
    const int a0 = m.chunks[(size_t)m.cur_chunk].a0;  // arg_dtype / func are indexed over ALL accumulators, the state block over the active chunk's
    const int t = m.arg_dtype[a0 + a], f = m.func[a0 + a];
    bool has = false;
    uint64_t cur = 0;
    for (int r = 0; r < world; ++r) {
      const uint64_t* st = all + (size_t)r * 2 * kMaxAggs;
      if (!st[2 * a]) continue;
      const uint64_t val = st[2 * a + 1];
      if (!has) {
        has = true;
        cur = val;
        continue;
      }
      if (f == AGG_COUNT) {
        cur += val;
      } else if (t == DFX_FLOAT64) {
        double x, y;
        memcpy(&x, &cur, 8);
        memcpy(&y, &val, 8);
        const double o = f == AGG_MIN ? fmin(x, y) : f == AGG_MAX ? fmax(x, y) : x + y;
        memcpy(&cur, &o, 8);
      } else if (t == DFX_FLOAT32) {
        float x, y;
        const uint32_t cx = (uint32_t)cur, cy = (uint32_t)val;
        memcpy(&x, &cx, 4);
        memcpy(&y, &cy, 4);
        const float o = f == AGG_MIN ? fminf(x, y) : f == AGG_MAX ? fmaxf(x, y) : x + y;
        uint32_t ob;
        memcpy(&ob, &o, 4);
        cur = ob;
      } else if (dtype_is_signed(t)) {
        const int64_t x = (int64_t)cur, y = (int64_t)val;
        cur = f == AGG_MIN ? (uint64_t)std::min(x, y) : f == AGG_MAX ? (uint64_t)std::max(x, y) : host_wrap_to((uint8_t)t, cur + val);
      } else {
        cur = f == AGG_MIN ? std::min(cur, val) : f == AGG_MAX ? std::max(cur, val) : host_wrap_to((uint8_t)t, cur + val);
      }
    }
    out[2 * a] = has ? 1 : 0;
    out[2 * a + 1] = cur;
  }
  DFX_HIP(hipMemcpy(m.state.get(), out, sizeof(out), hipMemcpyHostToDevice));
  return Status::OK();
}

Status AggregateRelation::partial_import(const void* src_device, const int64_t* counts, int n_buckets) {
  Impl& m = *impl_;
  if (!m.built) return Status::Err(DFX_GENERAL, "partial_build must precede partial_import");
  hipStream_t s = ctx().stream;
  uint64_t total = 0;
  for (int b = 0; b < n_buckets; ++b) total += (uint64_t)counts[b];
  const int cap_log2 = std::max(10, ceil_log2(4 * (total + 1)));
  if (cap_log2 > 31) return Status::Err(DFX_EXECUTION_ERROR, "GROUP BY table would exceed 2^31 slots");
  DevTable Tn;
  std::vector<std::shared_ptr<void>> owners;
  uint64_t* accs_full_new = nullptr;
  DFX_RETURN_IF_ERROR(m.partial_view_check());
  DFX_RETURN_IF_ERROR(m.alloc_table(cap_log2, &Tn, &owners, true, &accs_full_new));
  DevRows no_spill;
  no_spill.words = nullptr;
  no_spill.capacity = 0;
  const int nw = m.kw + m.na_total;  // rows as partial_export wrote them: every accumulator
  const DevTable Tall = m.full_view(Tn, accs_full_new);
  uint64_t off = 0;
  for (int b = 0; b < n_buckets; ++b) {
    if (counts[b] > 0)
      DFX_HIP(launch_merge_bucket((const uint64_t*)src_device + (size_t)nw * off, (uint64_t)counts[b], Tall, no_spill, s));
    off += (uint64_t)counts[b];
  }
  DFX_HIP(hipStreamSynchronize(s));
  m.early.cancel();
  ++m.table_generation;
  m.T = Tn;
  m.accs_full = accs_full_new;
  m.table_owners = owners;
  m.export_counts.clear();
  uint32_t hc[CTRL_WORDS];
  DFX_RETURN_IF_ERROR(m.read_ctrl(hc));
  m.occupied_known = hc[CTRL_OCCUPIED];
  return Status::OK();
}

}  // namespace dfx

using namespace dfx;

extern "C" {

int32_t dfx_aggregate_relation_new(const struct ArrowSchema* schema, struct ArrowArrayStream* input,
                                   const dfx_runtime_expr* const* group_exprs, int32_t n_group,
                                   const dfx_runtime_expr* const* aggr_exprs, int32_t n_aggr,
                                   struct ArrowArrayStream* out, char* err, size_t errlen) {
  return dfx_aggregate_relation_new_with_options(schema, input, group_exprs, n_group, aggr_exprs, n_aggr, nullptr, 0, out, err, errlen);
}

int32_t dfx_aggregate_relation_new_with_options(const struct ArrowSchema* schema, struct ArrowArrayStream* input,
                                                const dfx_runtime_expr* const* group_exprs, int32_t n_group,
                                                const dfx_runtime_expr* const* aggr_exprs, int32_t n_aggr,
                                                const dfx_option* options, int32_t n_options,
                                                struct ArrowArrayStream* out, char* err, size_t errlen) {
  return c_abi_guard(err, errlen, [&]() -> int32_t {
    if (!out || (n_options > 0 && !options)) return to_c(Status::Err(DFX_GENERAL, "null argument"), err, errlen);
    OptionOverrides ov;
    {
      AggOptions probe = agg_options();
      for (int i = 0; i < n_options; ++i) {
        if (!options[i].key || !set_option_in(probe, options[i].key, options[i].value))
          return to_c(Status::Err(DFX_GENERAL, std::string("unknown option ") + (options[i].key ? options[i].key : "(null)")), err, errlen);
        ov.emplace_back(options[i].key, options[i].value);
      }
    }
    std::unique_ptr<Relation> in;
    Status st = adopt_input_stream(input, &in);
    if (!st.ok()) return to_c(st, err, errlen);
    SchemaInfo si;
    st = schema_from_arrow(schema, &si);
    if (!st.ok()) return to_c(st, err, errlen);
    std::vector<dfx_runtime_expr> g, a;
    for (int i = 0; i < n_group; ++i) g.push_back(*group_exprs[i]);
    for (int i = 0; i < n_aggr; ++i) a.push_back(*aggr_exprs[i]);
    std::unique_ptr<Relation> rel(new AggregateRelation(si, std::move(in), std::move(g), std::move(a), std::move(ov)));
    export_relation(std::move(rel), out);
    return DFX_OK;
  });
}

static AggregateRelation* as_aggregate(struct ArrowArrayStream* s) {
  Relation* r = peek_exported(s);
  if (!r || r->kind() != REL_AGGREGATE) return nullptr;
  return static_cast<AggregateRelation*>(r);
}

int32_t dfx_aggregate_partial_build(struct ArrowArrayStream* agg, int32_t world, int32_t* n_words, int64_t* counts,
                                    char* err, size_t errlen) {
  return c_abi_guard(err, errlen, [&]() -> int32_t {
    AggregateRelation* a = as_aggregate(agg);
    if (!a) return to_c(Status::Err(DFX_GENERAL, "not an aggregate stream of this library"), err, errlen);
    int nw = 0;
    Status st = a->partial_build(world, &nw, counts);
    if (n_words) *n_words = nw;
    return to_c(st, err, errlen);
  });
}

int32_t dfx_aggregate_partial_export(struct ArrowArrayStream* agg, void* dst_device, int64_t dst_words, char* err,
                                     size_t errlen) {
  return c_abi_guard(err, errlen, [&]() -> int32_t {
    AggregateRelation* a = as_aggregate(agg);
    if (!a) return to_c(Status::Err(DFX_GENERAL, "not an aggregate stream of this library"), err, errlen);
    return to_c(a->partial_export(dst_device, dst_words), err, errlen);
  });
}

int32_t dfx_aggregate_partial_import(struct ArrowArrayStream* agg, const void* src_device, const int64_t* counts,
                                     int32_t n_buckets, char* err, size_t errlen) {
  return c_abi_guard(err, errlen, [&]() -> int32_t {
    AggregateRelation* a = as_aggregate(agg);
    if (!a) return to_c(Status::Err(DFX_GENERAL, "not an aggregate stream of this library"), err, errlen);
    return to_c(a->partial_import(src_device, counts, n_buckets), err, errlen);
  });
}

}  // extern "C"
