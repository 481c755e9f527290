// Host build of csrc/dfx_csv_walk.hpp (the record / field automaton the device kernels run) checked against the oracle's
// independent byte-at-a-time reader (oracle/dfx_oracle.c: orc_csv_*), on random text over the bytes that matter to the
// automaton.  Also checks the PARALLEL formulation: composing the per-chunk transition vectors of random chunkings gives
// the same record starts as the sequential walk.
// usage: csv_walk_fuzz <iterations> <seed> <tmpdir>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <random>
#include <string>
#include <vector>

#include "../../datafusion_archive_amd/csrc/dfx_csv_walk.hpp"
#include "../../oracle/dfx_oracle.h"

using namespace dfx;

// record starts by sequential simulation of the 5-state automaton
static std::vector<uint64_t> starts_sequential(const std::string& t) {
  std::vector<uint64_t> out;
  uint32_t s = 0;
  for (size_t i = 0; i < t.size(); ++i) {
    const uint32_t cls = csv_class((uint8_t)t[i]);
    if (s == 0 && cls != CSV_T) out.push_back(i);
    s = csv_tv_apply(csv_tv_of(cls), s);
  }
  return out;
}

// the same through transition vectors of random chunks (what k_csv_tile_trans / k_csv_mark do)
static std::vector<uint64_t> starts_parallel(const std::string& t, std::mt19937_64& rng) {
  std::vector<size_t> cuts = {0};
  while (cuts.back() < t.size()) cuts.push_back(std::min(t.size(), cuts.back() + 1 + (size_t)(rng() % 40)));
  std::vector<uint32_t> vec(cuts.size() - 1, kCsvTvId);
  for (size_t c = 0; c + 1 < cuts.size(); ++c)
    for (size_t i = cuts[c]; i < cuts[c + 1]; ++i) vec[c] = csv_tv_compose(vec[c], csv_tv_of(csv_class((uint8_t)t[i])));
  std::vector<uint64_t> out;
  uint32_t prefix = kCsvTvId;
  for (size_t c = 0; c + 1 < cuts.size(); ++c) {
    uint32_t s = csv_tv_apply(prefix, 0u);
    for (size_t i = cuts[c]; i < cuts[c + 1]; ++i) {
      const uint32_t cls = csv_class((uint8_t)t[i]);
      if (s == 0 && cls != CSV_T) out.push_back(i);
      s = csv_tv_apply(csv_tv_of(cls), s);
    }
    prefix = csv_tv_compose(prefix, vec[c]);
  }
  return out;
}

int main(int argc, char** argv) {
  const long iters = argc > 1 ? atol(argv[1]) : 2000;
  std::mt19937_64 rng(argc > 2 ? strtoull(argv[2], nullptr, 10) : 1);
  const std::string dir = argc > 3 ? argv[3] : "/tmp";
  const char* pieces[] = {"\"", "\"\"", ",", "\n", "\r", "\r\n", "a", "bc", " ", "1", "\"x\"", "\",\"", "\"\n\"", "'"};
  long records = 0, fields = 0;
  for (long it = 0; it < iters; ++it) {
    std::string t = "h\n";  // a one-field header: every record of the fuzz then has its own field count (flexible walk)
    const int np = 1 + (int)(rng() % 60);
    for (int i = 0; i < np; ++i) t += pieces[rng() % (sizeof(pieces) / sizeof(pieces[0]))];
    const std::vector<uint64_t> seq = starts_sequential(t), par = starts_parallel(t, rng);
    if (seq != par) {
      printf("MISMATCH parallel vs sequential record starts on %s\n", t.c_str());
      return 1;
    }
    // walk every record and compare its fields with the oracle reading the same bytes as a 1..N-column Utf8 file:
    // the oracle enforces equal field counts, so feed it one record at a time behind a header of the same width
    for (size_t r = 1; r < seq.size(); ++r) {
      const uint64_t begin = seq[r], limit = r + 1 < seq.size() ? seq[r + 1] : t.size();
      std::vector<std::string> got;
      const int nf = csv_walk_record((const uint8_t*)t.data(), begin, limit, [&](int, const CsvField& f) {
        std::string cell(f.ulen, '\0');
        if (f.ulen) csv_copy_field((const uint8_t*)t.data(), f, (uint8_t*)&cell[0]);
        got.push_back(cell);
      });
      std::string file;
      for (int i = 0; i < nf; ++i) file += i ? ",h" : "h";
      file += "\n";
      file.append(t, begin, limit - begin);
      const std::string path = dir + "/csv_walk_fuzz.csv";
      FILE* fp = fopen(path.c_str(), "wb");
      fwrite(file.data(), 1, file.size(), fp);
      fclose(fp);
      std::vector<int32_t> dts((size_t)nf, DFX_UTF8);
      orc_csv* c = nullptr;
      char err[256] = {0};
      if (orc_csv_open(path.c_str(), dts.data(), nf, 16, &c, err, sizeof err) != 0) {
        printf("oracle open failed: %s\n", err);
        return 1;
      }
      orc_batch* b = nullptr;
      const int32_t rc = orc_csv_next(c, &b, err, sizeof err);
      if (rc != 0 || !b || b->num_rows < 1 || b->num_columns != nf) {
        printf("MISMATCH: oracle rc=%d (%s) rows=%lld on record '%s' (walker found %d fields)\n", rc, err, b ? (long long)b->num_rows : -1,
               file.c_str(), nf);
        return 1;
      }
      for (int i = 0; i < nf; ++i) {
        const orc_array* a = b->columns[i];
        const std::string want((const char*)a->data + a->offsets[0], (size_t)(a->offsets[1] - a->offsets[0]));
        if (want != got[(size_t)i]) {
          printf("MISMATCH field %d of record '%s': walker '%s' oracle '%s'\n", i, file.c_str(), got[(size_t)i].c_str(), want.c_str());
          return 1;
        }
      }
      orc_batch_free(b);
      orc_csv_close(c);
      ++records;
      fields += nf;
    }
  }
  printf("ok: %ld records, %ld fields agree with the oracle; parallel == sequential boundaries on %ld texts\n", records, fields, iters);
  return 0;
}
