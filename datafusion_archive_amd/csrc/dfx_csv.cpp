// dfx_csv.cpp -- CsvDataSource (src/execution/datasource.rs:33-58) as a device relation.
//
//   CsvDataSource::new(filename, schema, batch_size) opens the file with arrow 0.12's csv::Reader::new(file, schema,
//   has_headers = true, batch_size, projection = None): the FIRST RECORD IS ALWAYS CONSUMED AS A HEADER (even when the
//   file has none -- uk_cities.csv: 37 lines -> 36 rows), every next() converts up to batch_size records, a cell that
//   does not parse is ArrowError::ParseError("Error while parsing value {cell} at line {n}"), empty cells of primitive
//   columns are null, Utf8 cells never are, and a record with a different number of fields than the first is the csv
//   crate's UnequalLengths error.  End of file is Ok(None).
//
// Here the text is copied to HBM once (pinned staging, one H2D copy), record boundaries are found by a parallel
// simulation of the csv automaton, and every next() converts one batch of records on the device (dfx_k_csv.hip); the
// batches never return to the host unless the consumer is the host.
#include <stdio.h>
#include <string.h>

#include <algorithm>

#include "dfx_csv_walk.hpp"
#include "dfx_relation.hpp"

namespace dfx {

class CsvRelation : public Relation {
 public:
  CsvRelation(std::string filename, SchemaInfo schema, int64_t batch_size)
      : filename_(std::move(filename)), schema_(std::move(schema)), batch_size_(batch_size) {}
  RelationKind kind() const override { return REL_CSV; }
  const SchemaInfo& schema() const override { return schema_; }
  Status next(DeviceBatch* out, bool* has) override;
  void require_columns(const std::vector<char>& needed) override { needed_ = needed; }
  void explain(std::string* out, int depth) const override {
    int n = 0;
    for (size_t i = 0; i < schema_.fields.size(); ++i) n += (needed_.empty() || needed_[i]) ? 1 : 0;
    explain_line(out, depth, strfmt("CsvDataSource: %s, text parsed on the device, %d of %d columns converted, batches of %lld rows",
                                    filename_.c_str(), n, (int)schema_.fields.size(), (long long)batch_size_));
  }
  Status open();  // File::open(filename).unwrap() happens in the constructor of the reference: so does this

 private:
  Status index_records();
  Status cell_error(uint64_t packed);
  std::string filename_;
  SchemaInfo schema_;
  int64_t batch_size_;
  std::shared_ptr<void> text_;       // device copy of the file, padded
  uint64_t n_bytes_ = 0;
  std::shared_ptr<void> row_start_;  // u64[records + 1]
  int64_t n_records_ = 0;            // including the header record
  uint32_t expected_fields_ = 0;
  int64_t next_record_ = 1;          // record 0 is the header
  bool indexed_ = false;
  std::vector<char> needed_;  // projection push-down: cells of other columns are located but not converted
};

Status CsvRelation::open() {
  DFX_RETURN_IF_ERROR(ensure_init());
  if (schema_.fields.size() > (size_t)kCsvMaxCols)
    return Status::Err(DFX_NOT_IMPLEMENTED, strfmt("CSV source with more than %d columns", kCsvMaxCols));
  if (schema_.fields.empty()) return Status::Err(DFX_GENERAL, "CSV source needs a schema with at least one column");
  for (const Field& f : schema_.fields)
    if (f.dtype < DFX_BOOLEAN || f.dtype > DFX_UTF8)
      return Status::Err(DFX_NOT_IMPLEMENTED, std::string("CSV column of type ") + dtype_name(f.dtype));
  FILE* fp = fopen(filename_.c_str(), "rb");
  if (!fp)  // the reference panics: File::open(filename).unwrap()
    return Status::Err(DFX_INTERNAL_ERROR, strfmt("called `Result::unwrap()` on an `Err` value: could not open %s", filename_.c_str()));
  fseek(fp, 0, SEEK_END);
  const long long size = ftell(fp);
  fseek(fp, 0, SEEK_SET);
  if (size < 0) {
    fclose(fp);
    return Status::Err(DFX_IO_ERROR, "cannot determine the size of " + filename_);
  }
  n_bytes_ = (uint64_t)size;
  Status st;
  const size_t padded = ((size_t)n_bytes_ + 64 + 63) / 64 * 64;
  text_ = device_alloc(padded, &st);
  if (!text_) {
    fclose(fp);
    return st;
  }
  hipStream_t s = ctx().stream;
  // pinned staging in 64 MB pieces: read() of piece i + 1 overlaps the H2D copy of piece i
  const size_t piece = 64u << 20;
  std::shared_ptr<void> stage[2];
  hipEvent_t done[2] = {nullptr, nullptr};
  for (int i = 0; i < 2; ++i) {
    stage[i] = pinned_alloc(std::min<size_t>(piece, std::max<size_t>((size_t)n_bytes_, 64)), &st);
    if (!stage[i]) {
      fclose(fp);
      return st;
    }
    if (hipEventCreateWithFlags(&done[i], hipEventDisableTiming) != hipSuccess) {
      fclose(fp);
      return Status::Err(DFX_EXECUTION_ERROR, "hipEventCreate failed");
    }
  }
  Status result = Status::OK();
  uint64_t off = 0;
  for (int i = 0; off < n_bytes_; ++i) {
    const int b = i & 1;
    if (i >= 2 && hipEventSynchronize(done[b]) != hipSuccess) {
      result = Status::Err(DFX_EXECUTION_ERROR, "hipEventSynchronize failed");
      break;
    }
    const size_t want = (size_t)std::min<uint64_t>(piece, n_bytes_ - off);
    const size_t got = fread(stage[b].get(), 1, want, fp);
    if (got != want) {
      result = Status::Err(DFX_IO_ERROR, "short read from " + filename_);
      break;
    }
    if (hipMemcpyAsync((uint8_t*)text_.get() + off, stage[b].get(), want, hipMemcpyHostToDevice, s) != hipSuccess ||
        hipEventRecord(done[b], s) != hipSuccess) {
      result = Status::Err(DFX_EXECUTION_ERROR, "H2D copy of the CSV text failed");
      break;
    }
    off += want;
  }
  fclose(fp);
  (void)hipMemsetAsync((uint8_t*)text_.get() + n_bytes_, 0, padded - (size_t)n_bytes_, s);
  (void)hipStreamSynchronize(s);
  for (int i = 0; i < 2; ++i) (void)hipEventDestroy(done[i]);
  return result;
}

Status CsvRelation::index_records() {
  indexed_ = true;
  hipStream_t s = ctx().stream;
  Status st;
  const int64_t tile = csv_tile_bytes();
  const int64_t n_tiles = (int64_t)((n_bytes_ + (uint64_t)tile - 1) / (uint64_t)tile);
  if (n_tiles == 0) {
    n_records_ = 0;
    return Status::OK();
  }
  auto trans = device_alloc(sizeof(uint32_t) * (size_t)n_tiles, &st);
  if (!trans) return st;
  auto state = device_alloc((size_t)n_tiles + 8, &st);
  if (!state) return st;
  auto block_vec = device_alloc(sizeof(uint32_t) * (size_t)(n_tiles / 1024 + 1), &st);  // one word per 1024 tiles
  if (!block_vec) return st;
  auto counts = device_alloc(sizeof(uint32_t) * (size_t)n_tiles, &st);
  if (!counts) return st;
  auto offsets = device_alloc(sizeof(uint64_t) * (size_t)(n_tiles + 1), &st);
  if (!offsets) return st;
  auto tmp = device_alloc(sizeof(uint64_t) * (size_t)(n_tiles / 4096 + 4), &st);
  if (!tmp) return st;
  const uint8_t* buf = (const uint8_t*)text_.get();
  DFX_HIP(launch_csv_boundaries_count(buf, n_bytes_, (uint32_t*)trans.get(), (uint32_t*)block_vec.get(), (uint8_t*)state.get(), (uint32_t*)counts.get(), s));
  DFX_HIP(launch_scan_u32((const uint32_t*)counts.get(), (uint64_t*)offsets.get(), n_tiles, (uint64_t*)tmp.get(), s));
  uint64_t total = 0;
  DFX_HIP(hipMemcpyAsync(&total, (uint64_t*)offsets.get() + n_tiles, sizeof(uint64_t), hipMemcpyDeviceToHost, s));
  DFX_HIP(hipStreamSynchronize(s));
  n_records_ = (int64_t)total;
  row_start_ = device_alloc(sizeof(uint64_t) * (size_t)(total + 1), &st);
  if (!row_start_) return st;
  DFX_HIP(launch_csv_boundaries_write(buf, n_bytes_, (const uint32_t*)trans.get(), (const uint8_t*)state.get(), (const uint64_t*)offsets.get(),
                                      (uint64_t*)row_start_.get(), s));
  DFX_HIP(hipMemcpyAsync((uint64_t*)row_start_.get() + total, &n_bytes_, sizeof(uint64_t), hipMemcpyHostToDevice, s));
  if (total > 0) {
    auto nf = device_alloc(sizeof(uint32_t) * 2, &st);
    if (!nf) return st;
    DFX_HIP(launch_csv_count_fields(buf, (const uint64_t*)row_start_.get(), 0, (uint32_t*)nf.get(), s));
    DFX_HIP(hipMemcpyAsync(&expected_fields_, nf.get(), sizeof(uint32_t), hipMemcpyDeviceToHost, s));
  }
  DFX_HIP(hipStreamSynchronize(s));
  return Status::OK();
}

// the failing cell's text, for the reference's message
Status CsvRelation::cell_error(uint64_t packed) {
  int64_t record = 0;
  int col = 0, code = 0;
  csv_err_unpack(packed, &code, &col, &record);
  const int64_t line = record;  // arrow: line_number starts at 1 with a header, + index of the record in the file
  uint64_t span[2] = {0, 0};
  hipStream_t s = ctx().stream;
  DFX_HIP(hipMemcpyAsync(span, (const uint64_t*)row_start_.get() + record, sizeof(span), hipMemcpyDeviceToHost, s));
  DFX_HIP(hipStreamSynchronize(s));
  std::vector<uint8_t> row((size_t)(span[1] - span[0]) + 1);
  if (span[1] > span[0])
    DFX_HIP(hipMemcpy(row.data(), (const uint8_t*)text_.get() + span[0], (size_t)(span[1] - span[0]), hipMemcpyDeviceToHost));
  if (code == 3) {
    const int nf = csv_walk_record(row.data(), 0, span[1] - span[0], [](int, const CsvField&) {});
    return Status::Err(DFX_ARROW_ERROR, strfmt("Error parsing line %lld: UnequalLengths { expected_len: %u, len: %d }",
                                               (long long)line, expected_fields_, nf));
  }
  std::string cell;
  csv_walk_record(row.data(), 0, span[1] - span[0], [&](int fi, const CsvField& f) {
    if (fi != col) return;
    cell.resize(f.ulen);
    if (f.ulen) csv_copy_field(row.data(), f, (uint8_t*)&cell[0]);
  });
  if (code == 2)
    return Status::Err(DFX_NOT_IMPLEMENTED, strfmt("value %s at line %lld needs arbitrary-precision decimal conversion "
                                                   "(more than 19 significant digits on a rounding boundary)", cell.c_str(), (long long)line));
  return Status::Err(DFX_ARROW_ERROR, strfmt("Error while parsing value %s at line %lld", cell.c_str(), (long long)line));
}

Status CsvRelation::next(DeviceBatch* out, bool* has) {
  *has = false;
  if (!indexed_) DFX_RETURN_IF_ERROR(index_records());
  const int64_t left = n_records_ - next_record_;
  if (left <= 0) return Status::OK();  // Ok(None)
  const int64_t nb = std::min(left, batch_size_ > 0 ? batch_size_ : left);
  hipStream_t s = ctx().stream;
  Status st;
  const int nc = (int)schema_.fields.size();
  DevCsvPlan plan;
  memset(&plan, 0, sizeof(plan));
  plan.n_cols = nc;
  plan.expected_fields = expected_fields_;
  auto ctrl = device_alloc(sizeof(uint64_t) * (size_t)(nc + 2), &st);  // null counts, the first error, general-path tiles
  if (!ctrl) return st;
  DFX_HIP(hipMemsetAsync(ctrl.get(), 0, sizeof(uint64_t) * (size_t)(nc + 2), s));
  DFX_HIP(hipMemsetAsync((uint64_t*)ctrl.get() + nc, 0xFF, sizeof(uint64_t), s));
  plan.null_counts = (uint64_t*)ctrl.get();
  plan.err = (uint64_t*)ctrl.get() + nc;
  plan.general_tiles = (uint64_t*)ctrl.get() + nc + 1;
  const size_t words = (size_t)(nb + 63) / 64;
  out->num_rows = nb;
  out->columns.clear();
  out->columns.resize((size_t)nc);
  std::vector<std::shared_ptr<void>> lens((size_t)nc), starts((size_t)nc);  // Utf8: cell lengths, cell positions in the text
  for (int c = 0; c < nc; ++c) {
    DeviceColumn& col = out->columns[c];
    col.dtype = schema_.fields[c].dtype;
    col.length = nb;
    if ((size_t)c < needed_.size() && !needed_[c]) {
      col.absent = true;
      plan.col[c].dtype = T_NONE;
      continue;
    }
    plan.col[c].dtype = (uint8_t)col.dtype;
    counters().csv_cells += nb;
    if (col.dtype == DFX_UTF8) {
      lens[c] = device_alloc(sizeof(int32_t) * (size_t)(nb + 1), &st);
      if (!lens[c]) return st;
      plan.col[c].lens = (int32_t*)lens[c].get();
      starts[c] = device_alloc(sizeof(uint64_t) * (size_t)nb, &st);
      if (!starts[c]) return st;
      plan.col[c].values = starts[c].get();
      continue;
    }
    const size_t vbytes = col.dtype == DFX_BOOLEAN ? words * 8 : (size_t)nb * dtype_width(col.dtype);
    auto vals = device_alloc(std::max<size_t>(vbytes, 8), &st);
    if (!vals) return st;
    auto valid = device_alloc(words * 8, &st);
    if (!valid) return st;
    plan.col[c].values = vals.get();
    plan.col[c].validity = (uint64_t*)valid.get();
    col.values = vals.get();
    col.validity = (const uint8_t*)valid.get();
    col.owners.push_back(vals);
    col.owners.push_back(valid);
  }
  const uint8_t* buf = (const uint8_t*)text_.get();
  const uint64_t* rs = (const uint64_t*)row_start_.get();
  const double avg_record = n_records_ > 0 ? (double)n_bytes_ / (double)n_records_ : 0.0;
  DFX_HIP(launch_csv_parse(buf, rs, next_record_, nb, plan, avg_record, agg_options().csv_wave_tiles, 0.0, s));
  std::vector<uint64_t> hc((size_t)nc + 2);
  DFX_HIP(hipMemcpyAsync(hc.data(), ctrl.get(), sizeof(uint64_t) * hc.size(), hipMemcpyDeviceToHost, s));
  DFX_HIP(hipStreamSynchronize(s));
  counters().csv_tiles += (nb + 63) / 64;
  counters().csv_general_tiles += (long long)hc[(size_t)nc + 1];
  if (hc[(size_t)nc] != ~0ull) return cell_error(hc[(size_t)nc]);
  for (int c = 0; c < nc; ++c) {
    DeviceColumn& col = out->columns[c];
    if (col.absent) continue;
    if (col.dtype != DFX_UTF8) {
      col.null_count = (int64_t)hc[(size_t)c];
      if (col.null_count == 0) col.validity = nullptr;
      continue;
    }
    auto offs = device_alloc(sizeof(int32_t) * (size_t)(nb + 1), &st);
    if (!offs) return st;
    auto tmp = device_alloc(sizeof(uint64_t) * (size_t)(nb / 4096 + 4), &st);
    if (!tmp) return st;
    DFX_HIP(launch_scan_i32((const int32_t*)lens[c].get(), (int32_t*)offs.get(), nb, (uint64_t*)tmp.get(), s));
    int32_t total = 0;
    DFX_HIP(hipMemcpyAsync(&total, (int32_t*)offs.get() + nb, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    DFX_HIP(hipStreamSynchronize(s));
    if (total < 0) return Status::Err(DFX_EXECUTION_ERROR, "Utf8 column of a CSV batch exceeds 2 GB (Arrow Utf8 offsets are 32-bit)");
    auto data = device_alloc((size_t)std::max<int32_t>(total, 8), &st);
    if (!data) return st;
    DFX_HIP(launch_csv_utf8_gather(buf, rs, next_record_, nb, c, (const int32_t*)offs.get(), (const uint64_t*)starts[c].get(), (uint8_t*)data.get(), s));
    col.offsets = (const int32_t*)offs.get();
    col.data = (const uint8_t*)data.get();
    col.data_bytes = total;
    col.null_count = 0;
    col.owners.push_back(offs);
    col.owners.push_back(data);
  }
  next_record_ += nb;
  *has = true;
  return Status::OK();
}

}  // namespace dfx

using namespace dfx;

extern "C" {

int32_t dfx_csv_datasource_new(const char* filename, const struct ArrowSchema* schema, int64_t batch_size,
                               struct ArrowArrayStream* out, char* err, size_t errlen) {
  return c_abi_guard(err, errlen, [&]() -> int32_t {
    if (!filename || !schema || !out) return to_c(Status::Err(DFX_GENERAL, "null argument"), err, errlen);
    SchemaInfo si;
    Status st = schema_from_arrow(schema, &si);
    if (!st.ok()) return to_c(st, err, errlen);
    std::unique_ptr<CsvRelation> rel(new CsvRelation(filename, si, batch_size));
    st = rel->open();
    if (!st.ok()) return to_c(st, err, errlen);
    export_relation(std::move(rel), out);
    return DFX_OK;
  });
}

}  // extern "C"
