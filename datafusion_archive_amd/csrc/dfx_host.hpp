// dfx_host.hpp -- host-side runtime shared by the C-ABI translation units: status type, schema
// model, device/pinned memory pools, device-resident batches and the internal operator interface.
//
// The internal operator tree mirrors the reference's: `struct Relation` below is
// src/execution/relation.rs:27-32 with RecordBatch replaced by a device-resident batch.  The
// Arrow C Stream adapters at the library edge (dfx_relation.cpp) convert to / from host Arrow.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include <exception>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/dfx.h"
#include "dfx_device.hpp"

namespace dfx {

// ---- status -----------------------------------------------------------------------------------
struct Status {
  int32_t code = DFX_OK;
  std::string msg;
  bool ok() const { return code == DFX_OK; }
  static Status OK() { return Status(); }
  static Status Err(int32_t code, std::string m) {
    Status s;
    s.code = code;
    s.msg = std::move(m);
    return s;
  }
};
std::string strfmt(const char* fmt, ...);
int32_t to_c(const Status& s, char* err, size_t errlen);

// Every extern "C" entry point runs its body through this: nothing unwinds across the C ABI (a Rust / C caller cannot catch a
// C++ exception), the caller gets DFX_INTERNAL_ERROR and the message instead.
template <class Body>
int32_t c_abi_guard(char* err, size_t errlen, Body&& body) noexcept {
  try {
    return body();
  } catch (const std::bad_alloc&) {
    return to_c(Status::Err(DFX_EXECUTION_ERROR, "out of host memory"), err, errlen);
  } catch (const std::exception& e) {
    return to_c(Status::Err(DFX_INTERNAL_ERROR, e.what()), err, errlen);
  } catch (...) {
    return to_c(Status::Err(DFX_INTERNAL_ERROR, "unknown C++ exception"), err, errlen);
  }
}

#define DFX_RETURN_IF_ERROR(expr)   \
  do {                              \
    ::dfx::Status _st = (expr);     \
    if (!_st.ok()) return _st;      \
  } while (0)

#define DFX_HIP(expr)                                                                          \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess)                                                                      \
      return ::dfx::Status::Err(DFX_EXECUTION_ERROR,                                           \
                                ::dfx::strfmt("HIP error %s at %s:%d (%s)", hipGetErrorString(_e), \
                                              __FILE__, __LINE__, #expr));                     \
  } while (0)

// ---- types / schema ---------------------------------------------------------------------------
const char* dtype_name(int dt);  // Rust {:?} of the DataType
int dtype_width(int dt);         // bytes of a fixed-width value (0: Boolean/Utf8)
bool dtype_is_numeric(int dt);
bool dtype_is_int(int dt);
bool dtype_is_signed(int dt);
const char* dtype_arrow_format(int dt);
int dtype_from_arrow_format(const char* fmt);  // DFX_TYPE_NONE if unsupported

struct Field {
  std::string name;
  int dtype = DFX_TYPE_NONE;
  bool nullable = true;
};
struct SchemaInfo {
  std::vector<Field> fields;
};
Status schema_from_arrow(const struct ArrowSchema* s, SchemaInfo* out);
// Schema of an operator whose arrays are its input's (Filter, Sort, Limit): the reference hands such operators an
// arbitrary Arc<Schema> (often Schema::empty()) that nothing checks against the arrays.  Here the exported schema must
// describe the arrays (consumers import buffers by it), so only the NAMES of a caller-supplied schema with the right
// field count are kept; types and nullability are the input's.
inline SchemaInfo schema_names_over(const SchemaInfo& caller, const SchemaInfo& input) {
  SchemaInfo out = input;
  if (caller.fields.size() == input.fields.size())
    for (size_t i = 0; i < out.fields.size(); ++i) out.fields[i].name = caller.fields[i].name;
  return out;
}
// exports a struct-typed ArrowSchema ("+s") owning all its memory
void schema_to_arrow(const SchemaInfo& s, struct ArrowSchema* out);

// ---- device context ---------------------------------------------------------------------------
struct Context {
  int device = 0;
  bool initialised = false;
  hipStream_t stream = nullptr;  // all kernels and copies of this process: one in-order stream
  hipStream_t aux = nullptr;     // side stream for control-block snapshots (keeps D2H copies out of the kernel chain)
};
Context& ctx();
Status ensure_init();
// 256 bytes of 0xFF on the device, allocated once per process: the "validity bitmap" of a column that has none, for the
// kernels that read every plan column's validity unconditionally (scan plans, dfx_device.hpp).  nullptr without a device.
const uint8_t* device_ones_block();
// measurement: bytes of column data copied host -> device by the uploaders (dfx_counter_get("h2d_bytes"))
struct Counters {
  long long h2d_bytes = 0;
  long long h2d_staged_bytes = 0;
  long long filter_output_regrows = 0;      // single-pass FilterRelation: batches that kept more rows than their output buffers were sized for
  long long filter_lookback_fallbacks = 0;  // ... batches redone in two passes because the look-back gave up waiting (a shared GPU)  // ... of which through the pinned staging ring (HostStreamOptions::mode 1)
  long long csv_cells = 0;  // cells converted by the CSV source
  long long csv_tiles = 0;          // 64-record tiles the CSV cell kernel converted ...
  long long csv_general_tiles = 0;  // ... of them through the per-lane walk (quotes, ragged records, a tile longer than the LDS window)
  // host-side time accounting of the aggregate (microseconds; tools/kprobe.py): where a query's wall time goes beyond its kernels
  long long agg_ctrl_wait_us = 0;   // blocked on control-block snapshots (one batch behind the launches)
  long long agg_sync_us = 0;        // other synchronous read-backs (calibration, growth, end of input)
  long long agg_emit_us = 0;        // emit_grouped / emit_ungrouped, launch to final synchronisation
  long long agg_drain_us = 0;       // drain(): every batch consumed and settled
  long long agg_alloc_us = 0;       // routing scratch / spill list / table allocation
  long long export_us = 0;          // device batch -> host Arrow (allocation of the pinned result buffers, D2H, synchronisation)
  long long export_alloc_us = 0;    // ... of which: result buffer allocation (pinned pool)
  long long export_host_ready = 0;  // result columns the exporter found already on the host (early key download)
  long long agg_early_keys = 0;     // speculative key-column downloads started ...
  long long agg_early_keys_used = 0;  // ... and still valid at emit
  long long agg_early_keys_late = 0;        // ... copies that had not arrived when emit asked (a stalled copy engine): retired, not waited for
  long long agg_emit_reused_early = 0;     // ... and emits that also reused its occupancy mask, tile offsets and device key column
  long long agg_pass2_launches = 0;
  long long agg_plane_launches = 0;   // pass-1 launches whose pass 2 runs once per accumulator plane of a shared operand (PTF_PLANES)
  long long agg_pair_launches = 0;    // pass-1 launches that routed two operands per row (PTF_PAIR)
  long long agg_pair_fallbacks = 0;   // streams that left the pair scan for one scan per aggregate (the table outgrew the pair kernels' partitions, a batch the plan cannot bind)
  long long agg_growths = 0;
  // dfx_aggregate_exchange, per rank (bench.py --gpus N: extra.phases_ms): where a multi-GPU step's wall time goes
  long long xchg_calls = 0;
  long long xchg_local_us = 0;       // the rank's own scan + local aggregation (drain), until its stream is idle
  long long xchg_wait_peers_us = 0;  // the first agreement round: mostly waiting for the slowest rank's local phase
  long long xchg_rounds = 0;         // collective rounds of the grouped exchanges (2 per query: the all-gather of states + counts, the buckets)
  long long xchg_host_syncs = 0;     // ... and their host synchronisations (2 per query)
  long long xchg_exchange_us = 0;    // counts, buffers, payload rounds, merge kernels, the closing agreement
  long long agg_shared_operand_launches = 0;  // pass-1 launches that routed {image, shared raw operand} rows (PTF_SHARED)
};
struct ScopedUs {  // adds the scope's wall time to a counter
  long long* acc;
  long long t0;
  static long long now();
  explicit ScopedUs(long long* a) : acc(a), t0(now()) {}
  ~ScopedUs() { *acc += now() - t0; }
};
Counters& counters();

// pooled device memory (hipMalloc is ~100 us; operators allocate per batch)
std::shared_ptr<void> device_alloc(size_t bytes, Status* st);
// pooled pinned host memory for H2D / D2H staging
std::shared_ptr<void> pinned_alloc(size_t bytes, Status* st);
void pool_trim();
void pool_inject_oom(int n);  // test hook (dfx_set_option "pool.inject_oom"): the next n device allocations fail their first attempt for real

// ---- device-resident data ---------------------------------------------------------------------
struct DeviceColumn {
  int dtype = DFX_TYPE_NONE;
  int64_t length = 0;
  int64_t null_count = 0;       // 0: validity may be absent
  const void* values = nullptr; // fixed width: element 0; Boolean: bitmap base (see bit_offset)
  const uint8_t* validity = nullptr;
  int64_t bit_offset = 0;       // for validity and Boolean values
  const int32_t* offsets = nullptr;  // Utf8: length + 1 entries
  const uint8_t* data = nullptr;     // Utf8: indexed by the raw offsets
  int64_t data_bytes = 0;            // Utf8: bytes referenced (offsets[length] - offsets[0])
  std::vector<std::shared_ptr<void>> owners;  // keeps the buffers alive
  bool absent = false;  // projection push-down: the consumer declared it never reads this column (no buffers)
  // A COMPLETE pinned host copy of `values` that its producer already made (the aggregate's key column, downloaded while
  // the scan was still running): the exporter hands it out instead of copying.  Honoured only while `values` still is the
  // pointer the copy was made from (a slice of the column moves `values` and so drops it).
  std::shared_ptr<void> host_values;
  const void* host_values_of = nullptr;
  size_t host_bytes = 0;
};

struct DeviceBatch {
  int64_t num_rows = 0;
  std::vector<DeviceColumn> columns;
};

// How a source of HOST Arrow batches moves them to HBM (AggOptions::host_*: "host.stream", "host.stage_threads",
// "host.stage_mb", "host.stage_slots"; an operator's own option set reaches its source through Relation::host_stream_options)
struct HostStreamOptions {
  int mode = 0;      // 0 (default) in order: hipMemcpyAsync of the pageable buffers on the library's stream (HIP pins chunk-wise
                     //   inside the runtime) and one stream synchronisation per batch: 53.6 GB/s = 0.85 of the link, the best of the
                     //   four forms on this platform (profiles/r04_host_stream_matrix.txt);
                     //   1 staged: library threads copy the producer's buffers into a ring of pinned slots while the DMA engine
                     //   drains the slots filled before, the producer's array released when its bytes have been copied out -- built in
                     //   round 4 to reach the engine's 57 GB/s from pinned memory; measured 38-46 GB/s whatever the thread count (2-16),
                     //   slot size (2-32 MB) or ring depth: the staging copy triples the host-memory traffic per byte moved;
                     //   2 one batch ahead on a copy stream (51.6 GB/s); 3 = 2 + the producer's large buffers page-locked in place
  int threads = 8;   // staged: threads filling slots
  int piece_mb = 16; // staged: bytes per slot (a column buffer travels in pieces of this size)
  int slots = 6;     // staged: pinned slots (slots x piece_mb of pinned memory per source)
};

enum RelationKind { REL_HOST_STREAM, REL_TABLE_SCAN, REL_FILTER, REL_PROJECT, REL_AGGREGATE, REL_CSV, REL_SORT, REL_LIMIT };

// trait Relation (src/execution/relation.rs:27-32)
struct Relation {
  virtual ~Relation() {}
  virtual RelationKind kind() const = 0;
  // Ok(Some(batch)) -> *has = true; Ok(None) -> *has = false
  virtual Status next(DeviceBatch* out, bool* has) = 0;
  virtual const SchemaInfo& schema() const = 0;
  // Projection push-down (the rule the reference has written but switched off: sqlplanner.rs:433-539, context.rs:89;
  // TableScan.projection logicalplan.rs:340-345).  A consumer tells its input which of the input's columns it will
  // ever read; the producer may then leave the others `absent` (not uploaded / not parsed / not compacted).  Without
  // a call every column is produced.  needed.size() == schema().fields.size().
  virtual void require_columns(const std::vector<char>& needed) { (void)needed; }
  // One line per operator, children indented below (dfx_relation_explain): what was fused, which kernel family a
  // program will run on.  Host state only -- never touches the device, so it also works without one.
  virtual void explain(std::string* out, int depth) const;
  // What earlier queries learned about this input and may reuse (null: nothing is remembered).  A resident table keeps
  // the number of groups an aggregate's calibration slice produced, keyed by a fingerprint of the fused program, so that
  // the second query of the same shape does not pay for the slice and its synchronous read-back again.
  virtual struct ScanMemo* scan_memo() { return nullptr; }
  // A consumer whose result does not depend on the batch width (an aggregate) may ask a source that slices RESIDENT data
  // for batches of at least `rows` rows: a scan of an HBM table then hands out one slice per routing window instead of
  // many small ones (each costs a launch).  Sources that produce batches (host streams, CSV) ignore it.
  virtual void prefer_batch_rows(int64_t /*rows*/) {}
  // An operator with its own option set tells the host source below it how to move batches (before its first next());
  // operators forward it, device sources ignore it.  Without it a host source uses the process defaults.
  virtual void host_stream_options(const HostStreamOptions& /*o*/) {}
};
struct ScanMemo {  // shared by every scan of a resident table (a `mutable` member of a const TableData): guarded
  mutable std::mutex mu;
  std::vector<std::pair<uint64_t, uint64_t>> calibrated_groups;  // (program fingerprint, groups in the first 2^18 rows)
  bool lookup(uint64_t fp, uint64_t* groups) const {
    std::lock_guard<std::mutex> lk(mu);
    for (const auto& e : calibrated_groups)
      if (e.first == fp) {
        *groups = e.second;
        return true;
      }
    return false;
  }
  void remember(uint64_t fp, uint64_t groups) {
    std::lock_guard<std::mutex> lk(mu);
    for (auto& e : calibrated_groups)
      if (e.first == fp) {
        e.second = groups;
        return;
      }
    if (calibrated_groups.size() < 64) calibrated_groups.emplace_back(fp, groups);
  }
};
// a fused device program ran out of columns / computed values / literals (ProgramBuilder::emit): callers that can split
// their work into several programs do so on this error
inline bool program_limit_error(const Status& st) {
  return !st.ok() && st.code == DFX_NOT_IMPLEMENTED && st.msg.compare(0, 16, "fused expression") == 0;
}
// "<indent><text>\n"
void explain_line(std::string* out, int depth, const std::string& text);
// "program: c columns, i instructions, l literals"
std::string explain_program(const DevProgram& P);

// ---- expressions (dfx_expr.cpp) ---------------------------------------------------------------
enum AggregateType { AGG_MIN = 0, AGG_MAX = 1, AGG_SUM = 2, AGG_COUNT = 3, AGG_AVG = 4 };

}  // namespace dfx

// RuntimeExpr (src/execution/expression.rs:42-54): a validated copy of the Expr tree + name + type.
struct dfx_runtime_expr {
  std::vector<dfx_expr_node> nodes;  // own copy (names point into `strings`)
  std::vector<std::string> strings;
  std::vector<char> has_name;
  dfx_runtime_expr() = default;
  dfx_runtime_expr(const dfx_runtime_expr& o) { *this = o; }
  dfx_runtime_expr& operator=(const dfx_runtime_expr& o) {
    if (this == &o) return *this;
    nodes = o.nodes;
    strings = o.strings;
    has_name = o.has_name;
    root = o.root;
    name = o.name;
    dtype = o.dtype;
    is_aggregate = o.is_aggregate;
    agg_func = o.agg_func;
    agg_arg = o.agg_arg;
    agg_type = o.agg_type;
    rebind();
    return *this;
  }
  void rebind() {  // node names must point into THIS object's strings
    for (size_t i = 0; i < nodes.size(); ++i) nodes[i].name = has_name[i] ? strings[i].c_str() : nullptr;
  }
  int32_t root = -1;
  std::string name;
  int32_t dtype = DFX_TYPE_NONE;  // type of the evaluated array
  bool is_aggregate = false;
  int32_t agg_func = -1;   // dfx::AggregateType
  int32_t agg_arg = -1;    // node index of args[0]
  int32_t agg_type = DFX_TYPE_NONE;  // declared return_type `t`
};

namespace dfx {

// Builds ONE fused device program for a set of expression roots over one input schema, with
// common-subexpression elimination.  Type errors the reference raises at evaluation time
// (comparison_ops / math_ops / boolean_ops downcasts) are returned by add() as a Status the
// operator stores and raises on its first next().
class ProgramBuilder {
 public:
  explicit ProgramBuilder(const SchemaInfo& schema);
  // returns the operand naming the value of node `root` of `e`, and its dtype
  Status add(const dfx_runtime_expr& e, int32_t root, uint8_t* operand, int* dtype);
  const DevProgram& program() const { return prog_; }
  // column slot -> schema column index
  const std::vector<int>& columns() const { return cols_; }
  // bind the batch's buffers; sets has_nulls
  Status bind(const DeviceBatch& batch, DevProgram* prog, DevColumns* cols) const;
  // Derive the shape-specialised plan (dfx_device.hpp: DevFastPlan) from the SSA program: pred is a
  // conjunction of `column <op> literal`, keys are plain columns, arguments are a column or a product
  // of up to three (column | literal +- column | column * literal) factors.  F->valid = 0 otherwise.
  void build_fast(uint8_t pred, const uint8_t* keys, int kw, const uint8_t* args, int na, DevFastPlan* F) const;

 private:
  Status emit(const dfx_runtime_expr& e, int32_t idx, uint8_t* operand, int* dtype);
  const SchemaInfo& schema_;
  DevProgram prog_;
  std::vector<int> cols_;
};

}  // namespace dfx
