// dfx_sort.cpp -- SortRelation and LimitRelation: LogicalPlan::Sort / LogicalPlan::Limit (logicalplan.rs:313-338), which
// the reference's planner produces for ORDER BY / LIMIT (sqlplanner.rs:142-183) and its executor leaves at
// `unimplemented!()` (context.rs:113,194).  SURVEY.md section 8(f) rank 4.  No reference behaviour exists, so the
// semantics are defined here and in the oracle (tests/oracle.py sort_batches / limit_batches), PARITY UNPINNED:
//   * ORDER BY e1 [ASC|DESC], e2 ...: stable; NULL is larger than every value (last when ascending, first when
//     descending); NaN is larger than every number; the result is ONE batch (like the aggregate's);
//   * sort keys: any scalar expression of a fixed-width type, or a Utf8 COLUMN (byte-wise lexicographic order, like Rust's
//     `str` Ord); payload: any column type;
//   * LIMIT n: the first n rows of the input stream, batch boundaries kept.
#include <string.h>

#include <algorithm>

#include "dfx_relation.hpp"

namespace dfx {

namespace {

// Expr::Column(i) as a RuntimeExpr (what compile_scalar_expr would return)
dfx_runtime_expr column_expr(int i, const SchemaInfo& schema) {
  dfx_runtime_expr e;
  dfx_expr_node n;
  memset(&n, 0, sizeof(n));
  n.kind = DFX_EXPR_COLUMN;
  n.left = n.right = -1;
  n.column = i;
  e.nodes.push_back(n);
  e.strings.push_back("");
  e.has_name.push_back(0);
  e.root = 0;
  e.name = schema.fields[(size_t)i].name;
  e.dtype = schema.fields[(size_t)i].dtype;
  e.rebind();
  return e;
}

template <typename T>
Status upload_vec(const std::vector<T>& v, std::shared_ptr<void>* dev) {
  Status st;
  *dev = device_alloc(std::max<size_t>(sizeof(T) * v.size(), 8), &st);
  if (!*dev) return st;
  if (!v.empty()) DFX_HIP(hipMemcpyAsync(dev->get(), v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice, ctx().stream));
  DFX_HIP(hipStreamSynchronize(ctx().stream));  // v may be a temporary
  return Status::OK();
}

}  // namespace

class SortRelation : public Relation {
 public:
  SortRelation(std::unique_ptr<Relation> input, std::vector<dfx_runtime_expr> keys, std::vector<int> asc, SchemaInfo schema)
      : keys_(std::move(keys)), asc_(std::move(asc)), schema_(std::move(schema)) {
    n_payload_ = (int)input->schema().fields.size();
    schema_ = schema_names_over(schema_, input->schema());
    // payload columns pass through (zero copy), the keys are evaluated next to them by the projection machinery
    std::vector<dfx_runtime_expr> exprs;
    for (int i = 0; i < n_payload_; ++i) exprs.push_back(column_expr(i, input->schema()));
    for (const dfx_runtime_expr& k : keys_) exprs.push_back(k);
    for (const dfx_runtime_expr& k : keys_) {
      if (k.is_aggregate) deferred_ = Status::Err(DFX_INTERNAL_ERROR, "explicit panic: get_func() on an aggregate expression");
    }
    if (deferred_.ok()) projected_.reset(new ProjectRelation(std::move(input), exprs, SchemaInfo()));
  }
  RelationKind kind() const override { return REL_SORT; }
  const SchemaInfo& schema() const override { return schema_; }
  Status next(DeviceBatch* out, bool* has) override;
  // every single-input operator forwards the hint (rule: the FIRST operator with its own option set above a host source
  // decides how that source moves its batches -- HostStreamRelation::host_stream_options ignores later callers)
  void host_stream_options(const HostStreamOptions& o) override { if (projected_) projected_->host_stream_options(o); }
  // ORDER BY ... LIMIT k: a LimitRelation directly above tells the sort that only the first k rows will be read
  void set_limit(int64_t k) { limit_ = k; }
  void explain(std::string* out, int depth) const override {
    std::string text = strfmt("Sort: %d keys (", (int)keys_.size());
    for (size_t i = 0; i < keys_.size(); ++i) text += std::string(i ? ", " : "") + keys_[i].name + (asc_[i] ? " ASC" : " DESC");
    text += "), stable LSD radix sort of key images";
    if (limit_ >= 0) text += strfmt(", top-%lld by radix select", (long long)limit_);
    if (!deferred_.ok()) text += ", error deferred to next(): " + deferred_.msg;
    explain_line(out, depth, text);
    if (projected_) projected_->explain(out, depth + 1);
  }

 private:
  // sorts the m row indices of *idx (rows of the n-row input) by sort key `key`
  Status sort_by_key(const std::vector<DeviceBatch>& batches, int key, int64_t n, int64_t m, std::shared_ptr<void>* idx);
  // top-k: the rows whose FIRST key is <= the k-th smallest first key (>= k of them, input order kept)
  Status select_candidates(const std::vector<DeviceBatch>& batches, int64_t n, int64_t k, std::shared_ptr<void>* idx, int64_t* m);
  int64_t limit_ = -1;
  std::unique_ptr<Relation> projected_;
  std::vector<dfx_runtime_expr> keys_;
  std::vector<int> asc_;
  SchemaInfo schema_;
  int n_payload_ = 0;
  Status deferred_;
  bool done_ = false;
};

// one stable sort of the current permutation by sort key `key` (column n_payload_ + key of every batch).  A fixed-width
// key is one 64-bit image; a Utf8 key is the sequence (length, last 8-byte chunk, ..., first chunk), least significant first.
Status SortRelation::sort_by_key(const std::vector<DeviceBatch>& batches, int key, int64_t n, int64_t m,
                                 std::shared_ptr<void>* idx) {
  hipStream_t s = ctx().stream;
  Status st;
  const int col = n_payload_ + key;
  const bool utf8 = batches[0].columns[(size_t)col].dtype == DFX_UTF8;
  bool any_nulls = false;
  for (const DeviceBatch& b : batches)
    if (b.columns[(size_t)col].validity && b.columns[(size_t)col].null_count != 0) any_nulls = true;
  auto image = device_alloc(sizeof(uint64_t) * (size_t)n, &st);
  if (!image) return st;
  std::shared_ptr<void> null_image;
  if (any_nulls) {
    null_image = device_alloc(sizeof(uint64_t) * (size_t)n, &st);
    if (!null_image) return st;
  }
  auto dmax = device_alloc(sizeof(uint32_t) * 2, &st);
  if (!dmax) return st;
  DFX_HIP(hipMemsetAsync(dmax.get(), 0, sizeof(uint32_t) * 2, s));
  // fills `image` (and, once, `null_image`) with part `chunk` of the key: fixed width: chunk 0 only; Utf8: -1 = length
  auto make_image = [&](int chunk, bool with_nulls, bool want_max) -> Status {
    int64_t at = 0;
    for (const DeviceBatch& b : batches) {
      const DeviceColumn& c = b.columns[(size_t)col];
      const uint8_t* validity = (c.validity && c.null_count != 0) ? c.validity : nullptr;
      uint64_t* ni = (any_nulls && with_nulls) ? (uint64_t*)null_image.get() + at : nullptr;
      if (utf8)
        DFX_HIP(launch_sort_image_utf8(c.offsets, c.data, validity, c.bit_offset, chunk, asc_[(size_t)key], b.num_rows,
                                       (uint64_t*)image.get() + at, ni, want_max ? (uint32_t*)dmax.get() : nullptr, s));
      else
        DFX_HIP(launch_sort_image(c.values, validity, c.bit_offset, (uint8_t)c.dtype, asc_[(size_t)key], b.num_rows,
                                  (uint64_t*)image.get() + at, ni, s));
      at += b.num_rows;
    }
    return Status::OK();
  };
  const int64_t tiles = radix_tiles(m);
  auto counts = device_alloc(sizeof(uint32_t) * (size_t)(256 * tiles), &st);
  if (!counts) return st;
  auto offsets = device_alloc(sizeof(uint64_t) * (size_t)(256 * tiles + 1), &st);
  if (!offsets) return st;
  auto tmp = device_alloc(sizeof(uint64_t) * (size_t)(256 * tiles / 4096 + 4), &st);
  if (!tmp) return st;
  auto img_a = device_alloc(sizeof(uint64_t) * (size_t)m, &st);
  if (!img_a) return st;
  auto img_b = device_alloc(sizeof(uint64_t) * (size_t)m, &st);
  if (!img_b) return st;
  auto idx_b = device_alloc(sizeof(uint32_t) * (size_t)m, &st);
  if (!idx_b) return st;
  auto hist = device_alloc(sizeof(uint64_t) * 8 * 256, &st);
  if (!hist) return st;
  std::shared_ptr<void> idx_a = *idx;
  // one stable 64-bit radix sort of the permutation by `src` (digits that are constant over the input are skipped)
  auto sort_by_image = [&](const uint64_t* src) -> Status {
    // the image of the m rows in their CURRENT order, then which of its 8 digits vary at all
    DFX_HIP(launch_sort_gather_u64(src, (const uint32_t*)idx_a.get(), m, (uint64_t*)img_a.get(), s));
    DFX_HIP(hipMemsetAsync(hist.get(), 0, sizeof(uint64_t) * 8 * 256, s));
    DFX_HIP(launch_radix_hist8((const uint64_t*)img_a.get(), m, (uint64_t*)hist.get(), s));
    uint64_t hh[8 * 256];
    DFX_HIP(hipMemcpyAsync(hh, hist.get(), sizeof(hh), hipMemcpyDeviceToHost, s));
    DFX_HIP(hipStreamSynchronize(s));
    for (int d = 0; d < 8; ++d) {
      bool constant = false;
      for (int b = 0; b < 256; ++b)
        if (hh[d * 256 + b] == (uint64_t)m) constant = true;
      if (constant) continue;  // every element has the same digit: the pass would be the identity
      DFX_HIP(launch_radix_count((const uint64_t*)img_a.get(), m, 8 * d, (uint32_t*)counts.get(), s));
      DFX_HIP(launch_scan_u32((const uint32_t*)counts.get(), (uint64_t*)offsets.get(), 256 * tiles, (uint64_t*)tmp.get(), s));
      DFX_HIP(launch_radix_scatter((const uint64_t*)img_a.get(), (const uint32_t*)idx_a.get(), m, 8 * d,
                                   (const uint64_t*)offsets.get(), (uint64_t*)img_b.get(), (uint32_t*)idx_b.get(), s));
      std::swap(img_a, img_b);
      std::swap(idx_a, idx_b);
    }
    return Status::OK();
  };
  if (!utf8) {
    DFX_RETURN_IF_ERROR(make_image(0, true, false));
    DFX_RETURN_IF_ERROR(sort_by_image((const uint64_t*)image.get()));
  } else {
    DFX_RETURN_IF_ERROR(make_image(-1, true, true));  // length: the least significant part
    uint32_t max_len = 0;
    DFX_HIP(hipMemcpyAsync(&max_len, dmax.get(), sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    DFX_HIP(hipStreamSynchronize(s));
    DFX_RETURN_IF_ERROR(sort_by_image((const uint64_t*)image.get()));
    for (int chunk = (int)((max_len + 7) / 8) - 1; chunk >= 0; --chunk) {
      DFX_RETURN_IF_ERROR(make_image(chunk, false, false));
      DFX_RETURN_IF_ERROR(sort_by_image((const uint64_t*)image.get()));
    }
  }
  if (any_nulls) DFX_RETURN_IF_ERROR(sort_by_image((const uint64_t*)null_image.get()));  // most significant: NULLs last / first
  *idx = idx_a;
  DFX_HIP(hipStreamSynchronize(s));
  return Status::OK();
}

// ORDER BY ... LIMIT k without sorting everything: radix-select the k-th smallest image T of the first key (8 rounds of
// one 256-bin histogram over the rows that still match the prefix), keep the rows with image <= T (at least k; ties
// included, input order kept), and sort only those.  Not applicable (full sort instead): Utf8 or nullable first key.
Status SortRelation::select_candidates(const std::vector<DeviceBatch>& batches, int64_t n, int64_t k, std::shared_ptr<void>* idx,
                                       int64_t* m) {
  hipStream_t s = ctx().stream;
  Status st;
  const int col = n_payload_;
  for (const DeviceBatch& b : batches) {
    const DeviceColumn& c = b.columns[(size_t)col];
    if (c.dtype == DFX_UTF8 || (c.validity && c.null_count != 0)) return Status::OK();  // *m stays n: full sort
  }
  auto image = device_alloc(sizeof(uint64_t) * (size_t)n, &st);
  if (!image) return st;
  int64_t at = 0;
  for (const DeviceBatch& b : batches) {
    const DeviceColumn& c = b.columns[(size_t)col];
    DFX_HIP(launch_sort_image(c.values, nullptr, c.bit_offset, (uint8_t)c.dtype, asc_[0], b.num_rows, (uint64_t*)image.get() + at, nullptr, s));
    at += b.num_rows;
  }
  auto hist = device_alloc(sizeof(uint64_t) * 256, &st);
  if (!hist) return st;
  uint64_t prefix = 0, mask = 0;
  int64_t rank = k;  // 1-based rank of the wanted element among the rows matching the prefix
  for (int d = 7; d >= 0; --d) {
    DFX_HIP(hipMemsetAsync(hist.get(), 0, sizeof(uint64_t) * 256, s));
    DFX_HIP(launch_select_hist((const uint64_t*)image.get(), n, prefix, mask, 8 * d, (uint64_t*)hist.get(), s));
    uint64_t hh[256];
    DFX_HIP(hipMemcpyAsync(hh, hist.get(), sizeof(hh), hipMemcpyDeviceToHost, s));
    DFX_HIP(hipStreamSynchronize(s));
    int b = 0;
    for (; b < 256; ++b) {
      if ((int64_t)hh[b] >= rank) break;
      rank -= (int64_t)hh[b];
    }
    if (b == 256) return Status::Err(DFX_INTERNAL_ERROR, "top-k selection lost its element");
    prefix |= (uint64_t)b << (8 * d);
    mask |= 0xFFull << (8 * d);
  }
  // rows with image <= prefix, in input order
  const int64_t n_words = (n + 63) / 64, n_tiles = (n + kTileRows - 1) / kTileRows;
  auto mw = device_alloc(sizeof(uint64_t) * (size_t)n_words, &st);
  if (!mw) return st;
  auto tc = device_alloc(sizeof(uint32_t) * (size_t)n_tiles, &st);
  if (!tc) return st;
  auto to = device_alloc(sizeof(uint64_t) * (size_t)(n_tiles + 1), &st);
  if (!to) return st;
  auto tmp = device_alloc(sizeof(uint64_t) * (size_t)(n_tiles / 4096 + 4), &st);
  if (!tmp) return st;
  DFX_HIP(launch_select_mask((const uint64_t*)image.get(), n, prefix, (uint64_t*)mw.get(), (uint32_t*)tc.get(), s));
  DFX_HIP(launch_scan_u32((const uint32_t*)tc.get(), (uint64_t*)to.get(), n_tiles, (uint64_t*)tmp.get(), s));
  uint64_t kept = 0;
  DFX_HIP(hipMemcpyAsync(&kept, (uint64_t*)to.get() + n_tiles, sizeof(uint64_t), hipMemcpyDeviceToHost, s));
  DFX_HIP(hipStreamSynchronize(s));
  if ((int64_t)kept < k) return Status::Err(DFX_INTERNAL_ERROR, "top-k selection kept too few rows");
  auto cand = device_alloc(sizeof(uint32_t) * (size_t)kept, &st);
  if (!cand) return st;
  DFX_HIP(launch_compact(idx->get(), 4, (const uint64_t*)mw.get(), (const uint64_t*)to.get(), n, cand.get(), 0, s));
  DFX_HIP(hipStreamSynchronize(s));
  *idx = cand;
  *m = (int64_t)kept;
  return Status::OK();
}

Status SortRelation::next(DeviceBatch* out, bool* has) {
  *has = false;
  if (!deferred_.ok()) return deferred_;
  if (done_) return Status::OK();
  done_ = true;
  DFX_RETURN_IF_ERROR(ensure_init());
  hipStream_t s = ctx().stream;
  std::vector<DeviceBatch> batches;
  int64_t n = 0;
  for (;;) {
    DeviceBatch b;
    bool got = false;
    DFX_RETURN_IF_ERROR(projected_->next(&b, &got));
    if (!got) break;
    if (b.num_rows > 0) {
      n += b.num_rows;
      batches.push_back(std::move(b));
    }
  }
  if (n == 0) return Status::OK();  // nothing to sort: Ok(None)
  if (n >= (1ll << 32)) return Status::Err(DFX_NOT_IMPLEMENTED, "ORDER BY over 2^32 or more rows");
  Status st;
  auto idx0 = device_alloc(sizeof(uint32_t) * (size_t)n, &st);
  if (!idx0) return st;
  std::shared_ptr<void> idx = idx0;
  DFX_HIP(launch_sort_iota((uint32_t*)idx.get(), n, s));
  int64_t m = n;         // rows that take part in the sort
  int64_t n_out = n;     // rows that are emitted
  if (limit_ >= 0 && limit_ < n) {
    n_out = limit_;
    if (n_out == 0) return Status::OK();
    if (limit_ <= n / 8) DFX_RETURN_IF_ERROR(select_candidates(batches, n, limit_, &idx, &m));  // else: sort everything
  }
  for (int k = (int)keys_.size() - 1; k >= 0; --k) DFX_RETURN_IF_ERROR(sort_by_key(batches, k, n, m, &idx));
  n = n_out;  // from here on: the rows that are emitted (the first n_out of the sorted permutation)
  // (batch, row) of every output row, then gather the payload columns from the input batches
  const int nb = (int)batches.size();
  std::vector<uint64_t> starts((size_t)nb + 1, 0);
  for (int b = 0; b < nb; ++b) starts[(size_t)b + 1] = starts[(size_t)b] + (uint64_t)batches[(size_t)b].num_rows;
  std::shared_ptr<void> dstarts;
  DFX_RETURN_IF_ERROR(upload_vec(starts, &dstarts));
  auto loc = device_alloc(sizeof(uint64_t) * (size_t)n, &st);
  if (!loc) return st;
  DFX_HIP(launch_sort_locate((const uint32_t*)idx.get(), n, (const uint64_t*)dstarts.get(), nb, (uint64_t*)loc.get(), s));
  out->num_rows = n;
  out->columns.clear();
  out->columns.resize((size_t)n_payload_);
  const size_t words = (size_t)(n + 63) / 64;
  for (int c = 0; c < n_payload_; ++c) {
    DeviceColumn& oc = out->columns[(size_t)c];
    const int dt = batches[0].columns[(size_t)c].dtype;
    oc.dtype = dt;
    oc.length = n;
    bool any_nulls = false;
    for (const DeviceBatch& b : batches)
      if (b.columns[(size_t)c].validity && b.columns[(size_t)c].null_count != 0) any_nulls = true;
    if (any_nulls) {
      std::vector<const uint8_t*> vb;
      std::vector<int64_t> vo;
      for (const DeviceBatch& b : batches) {
        const DeviceColumn& ic = b.columns[(size_t)c];
        vb.push_back((ic.validity && ic.null_count != 0) ? ic.validity : nullptr);
        vo.push_back(ic.bit_offset);
      }
      std::shared_ptr<void> dvb, dvo;
      DFX_RETURN_IF_ERROR(upload_vec(vb, &dvb));
      DFX_RETURN_IF_ERROR(upload_vec(vo, &dvo));
      auto valid = device_alloc(words * 8 + 8, &st);
      if (!valid) return st;
      auto zeros = device_alloc(sizeof(uint64_t), &st);
      if (!zeros) return st;
      DFX_HIP(hipMemsetAsync(zeros.get(), 0, sizeof(uint64_t), s));
      DFX_HIP(launch_gather_bits((const uint8_t* const*)dvb.get(), (const int64_t*)dvo.get(), (const uint64_t*)loc.get(), n,
                                 (uint64_t*)valid.get(), (uint64_t*)zeros.get(), s));
      uint64_t nz = 0;
      DFX_HIP(hipMemcpyAsync(&nz, zeros.get(), sizeof(uint64_t), hipMemcpyDeviceToHost, s));
      DFX_HIP(hipStreamSynchronize(s));
      if (nz) {
        oc.validity = (const uint8_t*)valid.get();
        oc.null_count = (int64_t)nz;
        oc.owners.push_back(valid);
      }
    }
    if (dt == DFX_UTF8) {
      std::vector<const int32_t*> ob;
      std::vector<const uint8_t*> db;
      for (const DeviceBatch& b : batches) {
        ob.push_back(b.columns[(size_t)c].offsets);
        db.push_back(b.columns[(size_t)c].data);
      }
      std::shared_ptr<void> dob, ddb;
      DFX_RETURN_IF_ERROR(upload_vec(ob, &dob));
      DFX_RETURN_IF_ERROR(upload_vec(db, &ddb));
      auto lens = device_alloc(sizeof(int32_t) * (size_t)(n + 1), &st);
      if (!lens) return st;
      auto offs = device_alloc(sizeof(int32_t) * (size_t)(n + 1), &st);
      if (!offs) return st;
      auto tmp = device_alloc(sizeof(uint64_t) * (size_t)(n / 4096 + 4), &st);
      if (!tmp) return st;
      DFX_HIP(launch_gather_utf8_lens((const int32_t* const*)dob.get(), (const uint64_t*)loc.get(), n, (int32_t*)lens.get(), s));
      DFX_HIP(launch_scan_i32((const int32_t*)lens.get(), (int32_t*)offs.get(), n, (uint64_t*)tmp.get(), s));
      int32_t total = 0;
      DFX_HIP(hipMemcpyAsync(&total, (int32_t*)offs.get() + n, sizeof(int32_t), hipMemcpyDeviceToHost, s));
      DFX_HIP(hipStreamSynchronize(s));
      if (total < 0) return Status::Err(DFX_EXECUTION_ERROR, "sorted Utf8 column exceeds 2 GB (Arrow Utf8 offsets are 32-bit)");
      auto data = device_alloc((size_t)std::max<int32_t>(total, 8), &st);
      if (!data) return st;
      DFX_HIP(launch_gather_utf8_copy((const int32_t* const*)dob.get(), (const uint8_t* const*)ddb.get(), (const uint64_t*)loc.get(), n,
                                      (const int32_t*)offs.get(), (uint8_t*)data.get(), s));
      oc.offsets = (const int32_t*)offs.get();
      oc.data = (const uint8_t*)data.get();
      oc.data_bytes = total;
      oc.owners.push_back(offs);
      oc.owners.push_back(data);
    } else if (dt == DFX_BOOLEAN) {
      std::vector<const uint8_t*> vb;
      std::vector<int64_t> vo;
      for (const DeviceBatch& b : batches) {
        vb.push_back((const uint8_t*)b.columns[(size_t)c].values);
        vo.push_back(b.columns[(size_t)c].bit_offset);
      }
      std::shared_ptr<void> dvb, dvo;
      DFX_RETURN_IF_ERROR(upload_vec(vb, &dvb));
      DFX_RETURN_IF_ERROR(upload_vec(vo, &dvo));
      auto vals = device_alloc(words * 8 + 8, &st);
      if (!vals) return st;
      DFX_HIP(launch_gather_bits((const uint8_t* const*)dvb.get(), (const int64_t*)dvo.get(), (const uint64_t*)loc.get(), n,
                                 (uint64_t*)vals.get(), nullptr, s));
      oc.values = vals.get();
      oc.owners.push_back(vals);
    } else {
      std::vector<const void*> vb;
      for (const DeviceBatch& b : batches) vb.push_back(b.columns[(size_t)c].values);
      std::shared_ptr<void> dvb;
      DFX_RETURN_IF_ERROR(upload_vec(vb, &dvb));
      const int w = dtype_width(dt);
      auto vals = device_alloc((size_t)n * (size_t)w, &st);
      if (!vals) return st;
      DFX_HIP(launch_gather_fixed((const void* const*)dvb.get(), (const uint64_t*)loc.get(), n, w, vals.get(), s));
      oc.values = vals.get();
      oc.owners.push_back(vals);
    }
  }
  DFX_HIP(hipStreamSynchronize(s));  // the input batches (and the pointer tables) are released after this
  *has = true;
  return Status::OK();
}

// ---- LIMIT -------------------------------------------------------------------------------------------
class LimitRelation : public Relation {
 public:
  LimitRelation(std::unique_ptr<Relation> input, int64_t limit, SchemaInfo schema)
      : input_(std::move(input)), left_(limit), schema_(std::move(schema)) {
    schema_ = schema_names_over(schema_, input_->schema());
  }
  RelationKind kind() const override { return REL_LIMIT; }
  const SchemaInfo& schema() const override { return schema_; }
  void require_columns(const std::vector<char>& needed) override { input_->require_columns(needed); }
  void host_stream_options(const HostStreamOptions& o) override { input_->host_stream_options(o); }
  void explain(std::string* out, int depth) const override {
    explain_line(out, depth, strfmt("Limit: %lld rows left", (long long)left_));
    input_->explain(out, depth + 1);
  }
  Status next(DeviceBatch* out, bool* has) override {
    *has = false;
    if (left_ <= 0) return Status::OK();
    bool got = false;
    DFX_RETURN_IF_ERROR(input_->next(out, &got));
    if (!got) return Status::OK();
    if (out->num_rows > left_) {  // a prefix of a batch: same buffers, shorter length
      out->num_rows = left_;
      for (DeviceColumn& c : out->columns) {
        c.length = left_;
        if (c.null_count != 0) c.null_count = -1;
        if (c.dtype == DFX_UTF8) c.data_bytes = 0;  // resolved lazily by the exporter from the offsets
      }
    }
    left_ -= out->num_rows;
    *has = true;
    return Status::OK();
  }

 private:
  std::unique_ptr<Relation> input_;
  int64_t left_;
  SchemaInfo schema_;
};

}  // namespace dfx

using namespace dfx;

extern "C" {

int32_t dfx_sort_relation_new(struct ArrowArrayStream* input, const dfx_runtime_expr* const* exprs, const int32_t* ascending,
                              int32_t n_exprs, const struct ArrowSchema* schema, struct ArrowArrayStream* out, char* err,
                              size_t errlen) {
  return c_abi_guard(err, errlen, [&]() -> int32_t {
    if (!out || n_exprs < 1 || !exprs || !ascending) return to_c(Status::Err(DFX_GENERAL, "invalid argument"), err, errlen);
    std::unique_ptr<Relation> in;
    Status st = adopt_input_stream(input, &in);
    if (!st.ok()) return to_c(st, err, errlen);
    SchemaInfo si;
    st = schema_from_arrow(schema, &si);
    if (!st.ok()) return to_c(st, err, errlen);
    std::vector<dfx_runtime_expr> keys;
    std::vector<int> asc;
    for (int i = 0; i < n_exprs; ++i) {
      if (!exprs[i]) return to_c(Status::Err(DFX_GENERAL, "null sort expression"), err, errlen);
      keys.push_back(*exprs[i]);
      asc.push_back(ascending[i] != 0);
    }
    std::unique_ptr<Relation> rel(new SortRelation(std::move(in), std::move(keys), std::move(asc), si));
    export_relation(std::move(rel), out);
    return DFX_OK;
  });
}

int32_t dfx_limit_relation_new(struct ArrowArrayStream* input, int64_t limit, const struct ArrowSchema* schema,
                               struct ArrowArrayStream* out, char* err, size_t errlen) {
  return c_abi_guard(err, errlen, [&]() -> int32_t {
    if (!out || limit < 0) return to_c(Status::Err(DFX_GENERAL, "invalid argument"), err, errlen);
    std::unique_ptr<Relation> in;
    Status st = adopt_input_stream(input, &in);
    if (!st.ok()) return to_c(st, err, errlen);
    SchemaInfo si;
    st = schema_from_arrow(schema, &si);
    if (!st.ok()) return to_c(st, err, errlen);
    if (in->kind() == REL_SORT) static_cast<SortRelation*>(in.get())->set_limit(limit);  // ORDER BY ... LIMIT k: top-k
    std::unique_ptr<Relation> rel(new LimitRelation(std::move(in), limit, si));
    export_relation(std::move(rel), out);
    return DFX_OK;
  });
}

}  // extern "C"
