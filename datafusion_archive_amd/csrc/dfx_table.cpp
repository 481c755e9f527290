// dfx_table.cpp -- HBM-resident tables (the in-memory DataSource: src/execution/datasource.rs:27-30,
// relation.rs:34-54), synthetic generators, and the library-level C ABI (init, info, profiling,
// options).
#include <string.h>

#include "dfx_relation.hpp"

struct dfx_table {
  std::shared_ptr<dfx::TableData> data;
};

namespace dfx {

TableScanRelation::TableScanRelation(std::shared_ptr<const TableData> t, int64_t batch_rows, int64_t row_begin, int64_t n_rows)
    : table_(std::move(t)), batch_rows_(batch_rows) {
  begin_ = std::max<int64_t>(0, std::min(row_begin, table_->num_rows));
  end_ = (n_rows < 0 || n_rows > table_->num_rows - begin_) ? table_->num_rows : begin_ + n_rows;  // (no begin_ + n_rows overflow for an `everything` n_rows)
  pos_ = begin_;
  if (batch_rows_ <= 0) batch_rows_ = end_ > begin_ ? end_ - begin_ : 1;
  batch_rows_ = (batch_rows_ + 63) / 64 * 64;  // slices stay byte-aligned in every bitmap
}

void TableScanRelation::explain(std::string* out, int depth) const {
  std::string range = (begin_ != 0 || end_ != table_->num_rows) ? strfmt(", rows [%lld, %lld) of them", (long long)begin_, (long long)end_) : std::string();
  explain_line(out, depth, strfmt("TableScan: %lld rows resident in HBM%s, %d columns, batches of %lld rows (zero-copy slices)",
                                  (long long)table_->num_rows, range.c_str(), (int)table_->columns.size(), (long long)batch_rows_));
}

Status TableScanRelation::next(DeviceBatch* out, bool* has) {
  *has = false;
  if (pos_ >= end_) return Status::OK();  // Ok(None)
  const int64_t n = std::min(batch_rows_, end_ - pos_);
  out->num_rows = n > 0 ? n : 0;
  out->columns.clear();
  out->columns.reserve(table_->columns.size());
  for (const DeviceColumn& c : table_->columns) {
    DeviceColumn s = c;  // shares the owners: zero copy
    s.length = out->num_rows;
    if (c.dtype == DFX_UTF8) {
      s.offsets = c.offsets + pos_;
      s.data_bytes = 0;  // resolved lazily by the exporter from the offsets
    } else if (c.dtype == DFX_BOOLEAN) {
      s.values = (const uint8_t*)c.values + (pos_ >> 3);
    } else {
      s.values = (const uint8_t*)c.values + (size_t)pos_ * dtype_width(c.dtype);
    }
    if (c.validity) s.validity = c.validity + (pos_ >> 3);
    if (c.null_count != 0) s.null_count = -1;
    out->columns.push_back(std::move(s));
  }
  pos_ += batch_rows_;
  emitted_any_ = true;
  *has = true;
  return Status::OK();
}

namespace {

// concatenates device batches column-wise into one resident table.  One batch is adopted as it is; several batches are
// copied: fixed-width values with D2D copies, validity / Boolean bits and Utf8 cells with the multi-batch gathers of
// dfx_k_sort.hip driven by the identity permutation (global row -> (batch, row)).
Status build_table(Relation* rel, std::shared_ptr<TableData>* out) {
  std::shared_ptr<TableData> t(new TableData());
  t->schema = rel->schema();
  std::vector<DeviceBatch> batches;
  int64_t total = 0;
  for (;;) {
    DeviceBatch b;
    bool has = false;
    DFX_RETURN_IF_ERROR(rel->next(&b, &has));
    if (!has) break;
    total += b.num_rows;
    if (b.num_rows > 0 || batches.empty()) batches.push_back(std::move(b));
  }
  t->num_rows = total;
  const size_t nc = t->schema.fields.size();
  t->columns.resize(nc);
  hipStream_t s = ctx().stream;
  if (batches.empty()) {
    for (size_t c = 0; c < nc; ++c) {
      t->columns[c].dtype = t->schema.fields[c].dtype;
      t->columns[c].length = 0;
    }
    *out = t;
    return Status::OK();
  }
  const bool single = batches.size() == 1;
  Status st;
  std::shared_ptr<void> loc;  // identity permutation resolved to (batch, row), built on first use
  auto need_loc = [&]() -> Status {
    if (loc) return Status::OK();
    if (total >= (1ll << 32)) return Status::Err(DFX_NOT_IMPLEMENTED, "multi-batch upload of 2^32 or more rows with nullable / Utf8 / Boolean columns");
    std::vector<uint64_t> starts(batches.size() + 1, 0);
    for (size_t b = 0; b < batches.size(); ++b) starts[b + 1] = starts[b] + (uint64_t)batches[b].num_rows;
    auto dstarts = device_alloc(sizeof(uint64_t) * starts.size(), &st);
    if (!dstarts) return st;
    DFX_HIP(hipMemcpyAsync(dstarts.get(), starts.data(), sizeof(uint64_t) * starts.size(), hipMemcpyHostToDevice, s));
    auto idx = device_alloc(sizeof(uint32_t) * (size_t)std::max<int64_t>(total, 1), &st);
    if (!idx) return st;
    loc = device_alloc(sizeof(uint64_t) * (size_t)std::max<int64_t>(total, 1), &st);
    if (!loc) return st;
    DFX_HIP(launch_sort_iota((uint32_t*)idx.get(), total, s));
    DFX_HIP(launch_sort_locate((const uint32_t*)idx.get(), total, (const uint64_t*)dstarts.get(), (int)batches.size(), (uint64_t*)loc.get(), s));
    DFX_HIP(hipStreamSynchronize(s));  // `starts` is a stack vector
    return Status::OK();
  };
  auto upload_ptrs = [&](const std::vector<const void*>& v, std::shared_ptr<void>* dev) -> Status {
    *dev = device_alloc(sizeof(void*) * std::max<size_t>(v.size(), 1), &st);
    if (!*dev) return st;
    DFX_HIP(hipMemcpyAsync(dev->get(), v.data(), sizeof(void*) * v.size(), hipMemcpyHostToDevice, s));
    DFX_HIP(hipStreamSynchronize(s));
    return Status::OK();
  };
  auto upload_i64 = [&](const std::vector<int64_t>& v, std::shared_ptr<void>* dev) -> Status {
    *dev = device_alloc(sizeof(int64_t) * std::max<size_t>(v.size(), 1), &st);
    if (!*dev) return st;
    DFX_HIP(hipMemcpyAsync(dev->get(), v.data(), sizeof(int64_t) * v.size(), hipMemcpyHostToDevice, s));
    DFX_HIP(hipStreamSynchronize(s));
    return Status::OK();
  };
  const size_t words = (size_t)(total + 63) / 64 + 1;
  for (size_t c = 0; c < nc; ++c) {
    DeviceColumn& col = t->columns[c];
    if (single && batches[0].columns[c].bit_offset == 0) {  // common case: adopt
      col = batches[0].columns[c];
      continue;
    }
    col.dtype = t->schema.fields[c].dtype;
    col.length = total;
    bool any_nulls = false;
    for (auto& b : batches)
      if (b.columns[c].validity && b.columns[c].null_count != 0) any_nulls = true;
    if (any_nulls) {
      DFX_RETURN_IF_ERROR(need_loc());
      std::vector<const void*> vb;
      std::vector<int64_t> vo;
      for (auto& b : batches) {
        const DeviceColumn& bc = b.columns[c];
        vb.push_back((bc.validity && bc.null_count != 0) ? bc.validity : nullptr);
        vo.push_back(bc.bit_offset);
      }
      std::shared_ptr<void> dvb, dvo;
      DFX_RETURN_IF_ERROR(upload_ptrs(vb, &dvb));
      DFX_RETURN_IF_ERROR(upload_i64(vo, &dvo));
      auto valid = device_alloc(words * 8, &st);
      if (!valid) return st;
      auto zeros = device_alloc(sizeof(uint64_t), &st);
      if (!zeros) return st;
      DFX_HIP(hipMemsetAsync(zeros.get(), 0, sizeof(uint64_t), s));
      DFX_HIP(launch_gather_bits((const uint8_t* const*)dvb.get(), (const int64_t*)dvo.get(), (const uint64_t*)loc.get(), total,
                                 (uint64_t*)valid.get(), (uint64_t*)zeros.get(), s));
      uint64_t nz = 0;
      DFX_HIP(hipMemcpyAsync(&nz, zeros.get(), sizeof(uint64_t), hipMemcpyDeviceToHost, s));
      DFX_HIP(hipStreamSynchronize(s));
      col.validity = (const uint8_t*)valid.get();
      col.null_count = (int64_t)nz;
      col.owners.push_back(valid);
    }
    if (col.dtype == DFX_UTF8) {
      DFX_RETURN_IF_ERROR(need_loc());
      std::vector<const void*> ob, db;
      for (auto& b : batches) {
        ob.push_back(b.columns[c].offsets);
        db.push_back(b.columns[c].data);
      }
      std::shared_ptr<void> dob, ddb;
      DFX_RETURN_IF_ERROR(upload_ptrs(ob, &dob));
      DFX_RETURN_IF_ERROR(upload_ptrs(db, &ddb));
      auto lens = device_alloc(sizeof(int32_t) * (size_t)(total + 1), &st);
      if (!lens) return st;
      auto offs = device_alloc(sizeof(int32_t) * (size_t)(total + 1), &st);
      if (!offs) return st;
      auto tmp = device_alloc(sizeof(uint64_t) * (size_t)(total / 4096 + 4), &st);
      if (!tmp) return st;
      DFX_HIP(launch_gather_utf8_lens((const int32_t* const*)dob.get(), (const uint64_t*)loc.get(), total, (int32_t*)lens.get(), s));
      DFX_HIP(launch_scan_i32((const int32_t*)lens.get(), (int32_t*)offs.get(), total, (uint64_t*)tmp.get(), s));
      int32_t bytes = 0;
      DFX_HIP(hipMemcpyAsync(&bytes, (int32_t*)offs.get() + total, sizeof(int32_t), hipMemcpyDeviceToHost, s));
      DFX_HIP(hipStreamSynchronize(s));
      if (bytes < 0) return Status::Err(DFX_EXECUTION_ERROR, "Utf8 column of a resident table exceeds 2 GB (Arrow Utf8 offsets are 32-bit)");
      auto data = device_alloc((size_t)std::max<int32_t>(bytes, 8), &st);
      if (!data) return st;
      DFX_HIP(launch_gather_utf8_copy((const int32_t* const*)dob.get(), (const uint8_t* const*)ddb.get(), (const uint64_t*)loc.get(), total,
                                      (const int32_t*)offs.get(), (uint8_t*)data.get(), s));
      col.offsets = (const int32_t*)offs.get();
      col.data = (const uint8_t*)data.get();
      col.data_bytes = bytes;
      col.owners.push_back(offs);
      col.owners.push_back(data);
    } else if (col.dtype == DFX_BOOLEAN) {
      DFX_RETURN_IF_ERROR(need_loc());
      std::vector<const void*> vb;
      std::vector<int64_t> vo;
      for (auto& b : batches) {
        vb.push_back(b.columns[c].values);
        vo.push_back(b.columns[c].bit_offset);
      }
      std::shared_ptr<void> dvb, dvo;
      DFX_RETURN_IF_ERROR(upload_ptrs(vb, &dvb));
      DFX_RETURN_IF_ERROR(upload_i64(vo, &dvo));
      auto vals = device_alloc(words * 8, &st);
      if (!vals) return st;
      DFX_HIP(launch_gather_bits((const uint8_t* const*)dvb.get(), (const int64_t*)dvo.get(), (const uint64_t*)loc.get(), total,
                                 (uint64_t*)vals.get(), nullptr, s));
      col.values = vals.get();
      col.owners.push_back(vals);
    } else {
      const int w = dtype_width(col.dtype);
      auto vals = device_alloc((size_t)std::max<int64_t>(total, 1) * w, &st);
      if (!vals) return st;
      int64_t pos = 0;
      for (auto& b : batches) {
        const DeviceColumn& bc = b.columns[c];
        if (bc.length) DFX_HIP(hipMemcpyAsync((uint8_t*)vals.get() + (size_t)pos * w, bc.values, (size_t)bc.length * w, hipMemcpyDeviceToDevice, s));
        pos += bc.length;
      }
      col.values = vals.get();
      col.owners.push_back(vals);
    }
  }
  DFX_HIP(hipStreamSynchronize(s));
  *out = t;
  return Status::OK();
}

}  // namespace
}  // namespace dfx

using namespace dfx;

namespace dfx {
bool set_option_in(AggOptions& o, const char* key, int64_t value) {
  if (!key) return false;
  if (!strcmp(key, "agg.strategy")) o.strategy = (int)value;
  else if (!strcmp(key, "agg.capacity_log2")) o.capacity_log2 = (int)value;
  else if (!strcmp(key, "agg.lds_slots")) o.lds_slots = (int)value;
  else if (!strcmp(key, "agg.lds_copies")) o.lds_copies = (int)value;
  else if (!strcmp(key, "scan.fast")) o.fast = (int)value;
  else if (!strcmp(key, "scan.plan")) o.plan = (int)value;
  else if (!strcmp(key, "agg.partition_mode")) o.partition_mode = (int)value;
  else if (!strcmp(key, "agg.partition_block")) o.partition_block = (int)value;
  else if (!strcmp(key, "agg.fewgroup")) o.fewgroup = (int)value;
  else if (!strcmp(key, "agg.replay_in_place")) o.replay_in_place = (int)value;
  else if (!strcmp(key, "agg.partition_pad")) o.partition_pad = (int)value;
  else if (!strcmp(key, "agg.dict_capacity_log2")) o.dict_capacity_log2 = (int)value;
  else if (!strcmp(key, "agg.partition_cap_rows")) o.partition_cap_rows = (int)value;
  else if (!strcmp(key, "agg.partition_defer")) o.partition_defer = (int)value;
  else if (!strcmp(key, "agg.partition_defer_batches")) o.partition_defer_batches = (int)value;
  else if (!strcmp(key, "agg.partition_split_rows")) o.partition_split_rows = (int)value;
  else if (!strcmp(key, "agg.pass2_stream")) o.pass2_stream = (int)value;
  else if (!strcmp(key, "agg.calibration_memo")) o.calibration_memo = (int)value;
  else if (!strcmp(key, "agg.emit_async")) o.emit_async = (int)value;
  else if (!strcmp(key, "agg.early_keys")) o.early_keys = (int)value;
  else if (!strcmp(key, "agg.hot_keys")) o.hot_keys = (int)value;
  else if (!strcmp(key, "agg.partition_layout")) o.partition_layout = (int)value;
  else if (!strcmp(key, "agg.narrow_keys")) o.narrow_keys = (int)value;
  else if (!strcmp(key, "agg.narrow_chunk16")) o.narrow_chunk16 = (int)value;
  else if (!strcmp(key, "agg.shared_operand")) o.shared_operand = (int)value;
  else if (!strcmp(key, "agg.ctrl_snapshot")) o.ctrl_snapshot = (int)value;
  else if (!strcmp(key, "export.kernel_copy")) o.export_kernel_copy = (int)value;
  else if (!strcmp(key, "agg.partition_producers")) o.partition_producers = (int)value;
  else if (!strcmp(key, "agg.pass1_ws")) o.pass1_ws = (int)value;
  else if (!strcmp(key, "agg.pass1_ws_dense")) o.pass1_ws_dense = (int)value;
  else if (!strcmp(key, "agg.pass1_ws_dense_scanners")) o.pass1_ws_dense_scanners = (int)value;
  else if (!strcmp(key, "agg.merge_scan_batches")) o.merge_scan_batches = (int)value;
  else if (!strcmp(key, "filter.single_pass")) o.filter_single_pass = (int)value;
  else if (!strcmp(key, "filter.dense")) o.filter_dense = (int)value;
  else if (!strcmp(key, "agg.split_aggregates")) o.split_aggregates = (int)value;
  else if (!strcmp(key, "agg.chunk_hold")) o.chunk_hold = (int)value;
  else if (!strcmp(key, "agg.pair_scan")) o.pair_scan = (int)value;
  else if (!strcmp(key, "agg.shared_planes")) o.shared_planes = (int)value;
  else if (!strcmp(key, "csv.wave_tiles")) o.csv_wave_tiles = (int)value;
  else if (!strcmp(key, "host.stream")) o.host_stream = (int)value;
  else if (!strcmp(key, "host.stage_threads")) o.host_stage_threads = (int)value;
  else if (!strcmp(key, "host.stage_mb")) o.host_stage_mb = (int)value;
  else if (!strcmp(key, "host.stage_slots")) o.host_stage_slots = (int)value;
  else return false;
  return true;
}
}  // namespace dfx

extern "C" {

int32_t dfx_init(int32_t device_ordinal, char* err, size_t errlen) {
  return c_abi_guard(err, errlen, [&]() -> int32_t {
    Context& c = ctx();
    if (c.initialised && c.device != device_ordinal)
      return to_c(Status::Err(DFX_GENERAL, "dfx_init: the library is already bound to another device"), err, errlen);
    c.device = device_ordinal;
    return to_c(ensure_init(), err, errlen);
  });
}

int32_t dfx_device_info(char* name, size_t namelen, int32_t* n_cu, int64_t* hbm_bytes, int32_t* wavefront, char* err,
                        size_t errlen) {
  return c_abi_guard(err, errlen, [&]() -> int32_t {
    Status st = ensure_init();
    if (!st.ok()) return to_c(st, err, errlen);
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, ctx().device);
    if (e != hipSuccess) return to_c(Status::Err(DFX_EXECUTION_ERROR, hipGetErrorString(e)), err, errlen);
    if (name && namelen) snprintf(name, namelen, "%s (%s)", prop.name, prop.gcnArchName);
    if (n_cu) *n_cu = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    if (wavefront) *wavefront = prop.warpSize;
    return DFX_OK;
  });
}

int32_t dfx_synchronize(char* err, size_t errlen) {
  return c_abi_guard(err, errlen, [&]() -> int32_t {
    Status st = ensure_init();
    if (!st.ok()) return to_c(st, err, errlen);
    hipError_t e = hipStreamSynchronize(ctx().stream);
    if (e != hipSuccess) return to_c(Status::Err(DFX_EXECUTION_ERROR, hipGetErrorString(e)), err, errlen);
    return DFX_OK;
  });
}

int32_t dfx_table_from_stream(struct ArrowArrayStream* input, dfx_table** out, char* err, size_t errlen) {
  return c_abi_guard(err, errlen, [&]() -> int32_t {
    if (!out) return to_c(Status::Err(DFX_GENERAL, "null argument"), err, errlen);
    *out = nullptr;
    Status st = ensure_init();
    if (!st.ok()) return to_c(st, err, errlen);
    std::unique_ptr<Relation> in;
    st = adopt_input_stream(input, &in);
    if (!st.ok()) return to_c(st, err, errlen);
    std::shared_ptr<TableData> t;
    st = build_table(in.get(), &t);
    if (!st.ok()) return to_c(st, err, errlen);
    *out = new dfx_table{t};
    return DFX_OK;
  });
}

static_assert(dfx::kSynthNullStream == DFX_SYNTH_NULL_STREAM, "device and header agree on the validity stream");
int32_t dfx_table_synth(const dfx_synth_column* cols, int32_t n_cols, uint64_t seed, int64_t row_begin, int64_t n_rows,
                        dfx_table** out, char* err, size_t errlen) {
  return c_abi_guard(err, errlen, [&]() -> int32_t {
    if (!out || !cols || n_cols < 1 || n_rows < 0) return to_c(Status::Err(DFX_GENERAL, "invalid argument"), err, errlen);
    *out = nullptr;
    Status st = ensure_init();
    if (!st.ok()) return to_c(st, err, errlen);
    std::shared_ptr<TableData> t(new TableData());
    t->num_rows = n_rows;
    hipStream_t s = ctx().stream;
    for (int c = 0; c < n_cols; ++c) {
      Field f;
      f.name = cols[c].name ? cols[c].name : strfmt("c%d", c);
      const int kind = DFX_SYNTH_KIND(cols[c].kind);
      const uint32_t permille = (uint32_t)DFX_SYNTH_NULL_PERMILLE(cols[c].kind);
      f.dtype = (kind == DFX_SYNTH_I64_UNIFORM || kind == DFX_SYNTH_I64_ZIPF || kind == DFX_SYNTH_I64_WIDE) ? DFX_INT64 : kind == DFX_SYNTH_I32_UNIFORM ? DFX_INT32 : DFX_FLOAT64;
      f.nullable = permille != 0;
      if (cols[c].kind < 0 || kind > DFX_SYNTH_I64_WIDE || permille > 1000 || (cols[c].kind >> 18) != 0)
        return to_c(Status::Err(DFX_NOT_IMPLEMENTED, "unknown synthetic column kind"), err, errlen);
      t->schema.fields.push_back(f);
      DeviceColumn col;
      col.dtype = f.dtype;
      col.length = n_rows;
      auto vals = device_alloc((size_t)std::max<int64_t>(n_rows, 1) * 8, &st);  // (4-byte kinds: half used)
      if (!vals) return to_c(st, err, errlen);
      hipError_t e = launch_synth(kind, cols[c].column_id, cols[c].p0, cols[c].p1, seed, row_begin, n_rows, vals.get(), s);
      if (e != hipSuccess) return to_c(Status::Err(DFX_EXECUTION_ERROR, hipGetErrorString(e)), err, errlen);
      col.values = vals.get();
      col.owners.push_back(vals);
      if (permille != 0 && n_rows > 0) {
        auto bits = device_alloc((size_t)((n_rows + 63) / 64) * 8 + 8, &st);
        if (!bits) return to_c(st, err, errlen);
        uint64_t* count = (uint64_t*)bits.get() + (n_rows + 63) / 64;  // (the word behind the bitmap)
        e = hipMemsetAsync(count, 0, 8, s);
        if (e == hipSuccess) e = launch_synth_validity(cols[c].column_id, permille, seed, row_begin, n_rows, (uint64_t*)bits.get(), count, s);
        uint64_t nulls = 0;
        if (e == hipSuccess) e = hipMemcpyAsync(&nulls, count, 8, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return to_c(Status::Err(DFX_EXECUTION_ERROR, hipGetErrorString(e)), err, errlen);
        col.validity = (const uint8_t*)bits.get();
        col.bit_offset = 0;
        col.null_count = (int64_t)nulls;
        col.owners.push_back(bits);
      }
      t->columns.push_back(std::move(col));
    }
    hipError_t e = hipStreamSynchronize(s);
    if (e != hipSuccess) return to_c(Status::Err(DFX_EXECUTION_ERROR, hipGetErrorString(e)), err, errlen);
    *out = new dfx_table{t};
    return DFX_OK;
  });
}

int64_t dfx_table_num_rows(const dfx_table* t) { return t ? t->data->num_rows : 0; }
int32_t dfx_table_num_columns(const dfx_table* t) { return t ? (int32_t)t->data->columns.size() : 0; }
const void* dfx_table_column_device_ptr(const dfx_table* t, int32_t column) {
  if (!t || column < 0 || column >= (int32_t)t->data->columns.size()) return nullptr;
  return t->data->columns[column].values;
}

int32_t dfx_table_scan_new(const dfx_table* t, int64_t batch_rows, struct ArrowArrayStream* out, char* err, size_t errlen) {
  return c_abi_guard(err, errlen, [&]() -> int32_t {
    if (!t || !out) return to_c(Status::Err(DFX_GENERAL, "null argument"), err, errlen);
    std::unique_ptr<Relation> rel(new TableScanRelation(t->data, batch_rows));
    export_relation(std::move(rel), out);
    return DFX_OK;
  });
}

int32_t dfx_table_scan_range_new(const dfx_table* t, int64_t row_begin, int64_t n_rows, int64_t batch_rows, struct ArrowArrayStream* out,
                                 char* err, size_t errlen) {
  return c_abi_guard(err, errlen, [&]() -> int32_t {
    if (!t || !out) return to_c(Status::Err(DFX_GENERAL, "null argument"), err, errlen);
    if (row_begin < 0 || (row_begin & 63) != 0 || row_begin > t->data->num_rows)
      return to_c(Status::Err(DFX_GENERAL, "row_begin must be a multiple of 64 inside the table"), err, errlen);
    std::unique_ptr<Relation> rel(new TableScanRelation(t->data, batch_rows, row_begin, n_rows));
    export_relation(std::move(rel), out);
    return DFX_OK;
  });
}

void dfx_table_free(dfx_table* t) { delete t; }

// ---- measurement hooks ---------------------------------------------------------------------------
int32_t dfx_profile_enable(int32_t on) {
  profile_enable(on != 0);
  return DFX_OK;
}
int32_t dfx_profile_reset(void) {
  profile_reset();
  return DFX_OK;
}
int32_t dfx_profile_count(void) { return profile_count(); }
int32_t dfx_profile_get(int32_t index, char* name, size_t namelen, int64_t* launches, double* total_ms, double* algo_bytes) {
  const char* nm = nullptr;
  int64_t l = 0;
  double ms = 0, b = 0;
  if (!profile_get(index, &nm, &l, &ms, &b)) return DFX_GENERAL;
  if (name && namelen) snprintf(name, namelen, "%s", nm);
  if (launches) *launches = l;
  if (total_ms) *total_ms = ms;
  if (algo_bytes) *algo_bytes = b;
  return DFX_OK;
}

uint64_t dfx_debug_group_hash(uint64_t key) { return host_hash_keys(&key, 1); }
uint32_t dfx_debug_unhash32(uint32_t image) { return host_unhash_word32(image); }

int64_t dfx_counter_get(const char* name) {
  if (!name) return -1;
  if (!strcmp(name, "h2d_bytes")) return counters().h2d_bytes;
  if (!strcmp(name, "h2d_staged_bytes")) return counters().h2d_staged_bytes;
  if (!strcmp(name, "filter_output_regrows")) return counters().filter_output_regrows;
  if (!strcmp(name, "filter_lookback_fallbacks")) return counters().filter_lookback_fallbacks;
  if (!strcmp(name, "csv_cells")) return counters().csv_cells;
  if (!strcmp(name, "export_host_ready")) return counters().export_host_ready;
  if (!strcmp(name, "agg_early_keys")) return counters().agg_early_keys;
  if (!strcmp(name, "agg_early_keys_used")) return counters().agg_early_keys_used;
  if (!strcmp(name, "agg_emit_reused_early")) return counters().agg_emit_reused_early;
  if (!strcmp(name, "agg_early_keys_late")) return counters().agg_early_keys_late;
  if (!strcmp(name, "csv_tiles")) return counters().csv_tiles;
  if (!strcmp(name, "csv_general_tiles")) return counters().csv_general_tiles;
  if (!strcmp(name, "agg_ctrl_wait_us")) return counters().agg_ctrl_wait_us;
  if (!strcmp(name, "agg_sync_us")) return counters().agg_sync_us;
  if (!strcmp(name, "agg_emit_us")) return counters().agg_emit_us;
  if (!strcmp(name, "agg_drain_us")) return counters().agg_drain_us;
  if (!strcmp(name, "agg_alloc_us")) return counters().agg_alloc_us;
  if (!strcmp(name, "agg_pass2_launches")) return counters().agg_pass2_launches;
  if (!strcmp(name, "agg_pair_launches")) return counters().agg_pair_launches;
  if (!strcmp(name, "agg_plane_launches")) return counters().agg_plane_launches;
  if (!strcmp(name, "agg_pair_fallbacks")) return counters().agg_pair_fallbacks;
  if (!strcmp(name, "agg_growths")) return counters().agg_growths;
  if (!strcmp(name, "agg_shared_operand_launches")) return counters().agg_shared_operand_launches;
  if (!strcmp(name, "xchg_calls")) return counters().xchg_calls;
  if (!strcmp(name, "xchg_local_us")) return counters().xchg_local_us;
  if (!strcmp(name, "xchg_wait_peers_us")) return counters().xchg_wait_peers_us;
  if (!strcmp(name, "xchg_exchange_us")) return counters().xchg_exchange_us;
  if (!strcmp(name, "xchg_rounds")) return counters().xchg_rounds;
  if (!strcmp(name, "xchg_host_syncs")) return counters().xchg_host_syncs;
  if (!strcmp(name, "export_us")) return counters().export_us;
  if (!strcmp(name, "export_alloc_us")) return counters().export_alloc_us;
  return -1;
}
void dfx_counter_reset(void) { counters() = Counters(); }

int32_t dfx_set_option(const char* key, int64_t value) {
  if (!key) return DFX_GENERAL;
  if (!strcmp(key, "pool.trim")) {
    pool_trim();
    return DFX_OK;
  }
  if (!strcmp(key, "pool.inject_oom")) {  // test hook: the next `value` device allocations that miss the pool fail their first attempt
    pool_inject_oom((int)value);
    return DFX_OK;
  }
  if (!strcmp(key, "test.exchange_fail")) {  // test hook: rank << 8 | stage fails locally at that stage of dfx_aggregate_exchange (0: off)
    set_exchange_test_failure(value);
    return DFX_OK;
  }
  return set_option_in(agg_options(), key, value) ? DFX_OK : DFX_GENERAL;
}

}  // extern "C"
