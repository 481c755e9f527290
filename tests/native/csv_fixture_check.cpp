// Host emulation of what the device CSV source computes for one file -- with the SAME headers the kernels compile
// (csrc/dfx_csv_walk.hpp: the record / field automaton, csrc/dfx_numparse.hpp: the number parsers) and the same
// per-cell rules as k_csv_parse (csrc/dfx_k_csv.hip: empty cell -> null, quoted number -> invalid, "true"/"false",
// integer width checks, Utf8 never null) -- compared cell by cell, bit for bit, with the oracle's reader
// (oracle/dfx_oracle.c: orc_csv_*).  The GPU tests read the same committed fixtures (tests/test_gpu_csv.py); this runs
// without a GPU and covers the data-dependent logic, not the kernels' indexing.
// usage: csv_fixture_check <file.csv> <dtype,dtype,...>     (dfx_dtype codes of include/dfx.h)
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../datafusion_archive_amd/csrc/dfx_csv_walk.hpp"
#include "../../datafusion_archive_amd/csrc/dfx_numparse.hpp"
#include "../../oracle/dfx_oracle.h"

using namespace dfx;

static bool bytes_equal(const uint8_t* p, uint64_t n, const char* lit) { return n == strlen(lit) && memcmp(p, lit, n) == 0; }

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  std::vector<int32_t> dts;
  for (char* tok = strtok(argv[2], ","); tok; tok = strtok(nullptr, ",")) dts.push_back(atoi(tok));
  const int n_cols = (int)dts.size();
  FILE* fp = fopen(argv[1], "rb");
  if (!fp) return 2;
  std::string text;
  char chunk[65536];
  for (size_t got; (got = fread(chunk, 1, sizeof chunk, fp)) > 0;) text.append(chunk, got);
  fclose(fp);
  const uint64_t n = text.size();
  text.append(64, '\0');  // the device buffer is padded too
  const uint8_t* buf = (const uint8_t*)text.data();

  // record starts: the 5-state automaton (the kernels compose the same transitions per chunk; csv_walk_fuzz checks that)
  std::vector<uint64_t> start;
  uint32_t s = 0;
  for (uint64_t i = 0; i < n; ++i) {
    const uint32_t cls = csv_class(buf[i]);
    if (s == 0 && cls != CSV_T) start.push_back(i);
    s = csv_tv_apply(csv_tv_of(cls), s);
  }
  start.push_back(n);
  const int64_t n_records = (int64_t)start.size() - 1;  // including the header record

  orc_csv* c = nullptr;
  char err[512] = {0};
  if (orc_csv_open(argv[1], dts.data(), n_cols, 1 << 20, &c, err, sizeof err) != 0) {
    printf("oracle open failed: %s\n", err);
    return 1;
  }
  orc_batch* b = nullptr;
  if (orc_csv_next(c, &b, err, sizeof err) != 0 || !b) {
    printf("oracle read failed: %s\n", err);
    return 1;
  }
  if (b->num_rows != n_records - 1 || b->num_columns != n_cols) {
    printf("MISMATCH: %lld data records found by the automaton, oracle has %lld rows x %d columns\n", (long long)(n_records - 1),
           (long long)b->num_rows, b->num_columns);
    return 1;
  }
  long cells = 0, nulls = 0;
  for (int64_t r = 1; r < n_records; ++r) {
    const int64_t row = r - 1;
    std::vector<CsvField> fields;
    const int nf = csv_walk_record(buf, start[(size_t)r], start[(size_t)r + 1], [&](int, const CsvField& f) { fields.push_back(f); });
    if (nf != n_cols) {
      printf("MISMATCH: record %lld has %d fields, schema has %d (the oracle accepted the file)\n", (long long)r, nf, n_cols);
      return 1;
    }
    for (int col = 0; col < n_cols; ++col) {
      const CsvField& f = fields[(size_t)col];
      const orc_array* a = b->columns[col];
      const bool o_valid = !a->validity || ((a->validity[row >> 3] >> (row & 7)) & 1);
      uint64_t cb, ce;
      csv_field_span(f, &cb, &ce);
      ++cells;
      if (dts[(size_t)col] == DFX_UTF8) {
        std::string cell(f.ulen, '\0');
        if (f.ulen) csv_copy_field(buf, f, (uint8_t*)&cell[0]);
        const std::string want((const char*)a->data + a->offsets[row], (size_t)(a->offsets[row + 1] - a->offsets[row]));
        if (!o_valid || want != cell) {
          printf("MISMATCH row %lld col %d: device '%s' oracle '%s' (valid %d)\n", (long long)row, col, cell.c_str(), want.c_str(), (int)o_valid);
          return 1;
        }
        continue;
      }
      bool valid = false;
      uint64_t bits = 0;  // the value's bit pattern, zero-extended
      if (f.ulen != 0) {
        int rc = f.complex ? NP_INVALID : NP_OK;
        const uint8_t* p = buf + cb;
        const int64_t len = (int64_t)f.ulen;
        if (rc == NP_OK) {
          switch (dts[(size_t)col]) {
            case DFX_FLOAT64: {
              double d = 0;
              rc = np_parse_f64(p, len, &d);
              memcpy(&bits, &d, 8);
              break;
            }
            case DFX_FLOAT32: {
              float d = 0;
              rc = np_parse_f32(p, len, &d);
              uint32_t w;
              memcpy(&w, &d, 4);
              bits = w;
              break;
            }
            case DFX_BOOLEAN:
              if (bytes_equal(p, (uint64_t)len, "true")) bits = 1;
              else if (!bytes_equal(p, (uint64_t)len, "false")) rc = NP_INVALID;
              break;
            default: {
              const int dt = dts[(size_t)col];
              const int width = (dt == DFX_INT8 || dt == DFX_UINT8) ? 8 : (dt == DFX_INT16 || dt == DFX_UINT16) ? 16
                                : (dt == DFX_INT32 || dt == DFX_UINT32) ? 32 : 64;
              rc = np_parse_int(p, len, width, dt >= DFX_INT8 && dt <= DFX_INT64, &bits);
              if (width < 64) bits &= (1ull << width) - 1;
              break;
            }
          }
        }
        if (rc != NP_OK) {
          printf("MISMATCH row %lld col %d: the device parser rejects '%.*s' (rc %d), the oracle accepted the file\n", (long long)row, col,
                 (int)len, (const char*)p, rc);
          return 1;
        }
        valid = true;
      }
      if (valid != o_valid) {
        printf("MISMATCH row %lld col %d: device valid %d, oracle valid %d\n", (long long)row, col, (int)valid, (int)o_valid);
        return 1;
      }
      if (!valid) {
        ++nulls;
        continue;
      }
      uint64_t want = 0;
      switch (dts[(size_t)col]) {
        case DFX_BOOLEAN: want = (((const uint8_t*)a->values)[row >> 3] >> (row & 7)) & 1; break;
        case DFX_INT8: case DFX_UINT8: want = ((const uint8_t*)a->values)[row]; break;
        case DFX_INT16: case DFX_UINT16: want = ((const uint16_t*)a->values)[row]; break;
        case DFX_INT32: case DFX_UINT32: case DFX_FLOAT32: want = ((const uint32_t*)a->values)[row]; break;
        default: want = ((const uint64_t*)a->values)[row]; break;
      }
      if (want != bits) {
        printf("MISMATCH row %lld col %d ('%.*s'): device bits %016llx oracle bits %016llx\n", (long long)row, col, (int)f.ulen,
               (const char*)(buf + cb), (unsigned long long)bits, (unsigned long long)want);
        return 1;
      }
    }
  }
  printf("ok: %lld rows, %ld cells (%ld nulls) agree bit for bit with the oracle\n", (long long)(n_records - 1), cells, nulls);
  orc_batch_free(b);
  orc_csv_close(c);
  return 0;
}
