// Host build of csrc/dfx_csv_walk.hpp (the record / field automaton the device kernels run) checked against the oracle's
// independent byte-at-a-time reader (oracle/dfx_oracle.c: orc_csv_*), on random text over the bytes that matter to the
// automaton.  Also checks the PARALLEL formulation: composing the per-chunk transition vectors of random chunkings gives
// the same record starts as the sequential walk.
// Round 5: the word-at-a-time functions for quote-free text (csv_plain_starts over 32-byte terminator masks, csv_plain_record's
// delimiter masks -- what k_csv_tile_trans / k_csv_mark_write / k_csv_parse run) against the same sequential walk.
// usage: csv_walk_fuzz <iterations> <seed> <tmpdir>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
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

// record starts of a quote-free text from 32-byte terminator masks (k_csv_tile_trans / k_csv_mark_write without the scan)
static std::vector<uint64_t> starts_plain_masks(const std::string& t) {
  std::string pad = t;
  pad.append(64 - pad.size() % 64 + 64, '\n');  // whole chunks; the padding is terminators: no start in it
  std::vector<uint64_t> out;
  for (size_t c = 0; c + 32 <= pad.size(); c += 32) {
    uint32_t w[8];
    memcpy(w, pad.data() + c, 32);
    const bool after_t = c == 0 || pad[c - 1] == '\n' || pad[c - 1] == '\r';
    uint32_t st = csv_plain_starts(csv_tmask32(w), after_t);
    for (; st; st &= st - 1) out.push_back(c + (size_t)__builtin_ctz(st));
  }
  while (!out.empty() && out.back() >= t.size()) out.pop_back();
  return out;
}

// every record of a text through csv_plain_record, against the walker
static bool check_plain_records(const std::string& t, const std::vector<uint64_t>& seq, long* n_plain, long* n_declined) {
  std::string pad = t + std::string(8, '"');  // readable past the end; quotes there must not be seen
  for (size_t r = 0; r < seq.size(); ++r) {
    const uint64_t begin = seq[r], limit = r + 1 < seq.size() ? seq[r + 1] : t.size();
    std::vector<CsvField> fields;
    bool any_quote = false;
    const int nf = csv_walk_record((const uint8_t*)t.data(), begin, limit, [&](int, const CsvField& f) { fields.push_back(f); });
    uint64_t rec_end = fields.back().end;
    for (uint64_t i = begin; i < rec_end; ++i) any_quote = any_quote || t[i] == '"';
    for (int F = std::max(1, nf - 1); F <= nf + 1; ++F) {
      std::vector<uint16_t> cells((size_t)F + 1, 0xFFFF);
      const bool ok = csv_plain_record((const uint8_t*)pad.data(), (uint32_t)begin, (uint32_t)limit, (uint32_t)F, cells.data());
      if (cells[(size_t)F] != 0xFFFF) {
        printf("MISMATCH csv_plain_record wrote past its %d entries on '%s'\n", F, t.c_str());
        return false;
      }
      // (a quote inside the trailing terminator run cannot be: those bytes are terminators)
      const bool want = !any_quote && F == nf;
      if (ok != want) {
        printf("MISMATCH csv_plain_record accepts=%d want=%d (F=%d, walker %d fields, quote=%d) record at %llu of '%s'\n", (int)ok, (int)want, F, nf,
               (int)any_quote, (unsigned long long)begin, t.c_str());
        return false;
      }
      if (!ok) {
        ++*n_declined;
        continue;
      }
      for (int i = 0; i < nf; ++i) {
        if ((uint64_t)cells[(size_t)i] != fields[(size_t)i].end) {
          printf("MISMATCH csv_plain_record cell %d ends at %u, walker at %llu, in '%s'\n", i, (unsigned)cells[(size_t)i],
                 (unsigned long long)fields[(size_t)i].end, t.c_str());
          return false;
        }
      }
      ++*n_plain;
    }
  }
  return true;
}

int main(int argc, char** argv) {
  const long iters = argc > 1 ? atol(argv[1]) : 2000;
  std::mt19937_64 rng(argc > 2 ? strtoull(argv[2], nullptr, 10) : 1);
  const std::string dir = argc > 3 ? argv[3] : "/tmp";
  const char* pieces[] = {"\"", "\"\"", ",", "\n", "\r", "\r\n", "a", "bc", " ", "1", "\"x\"", "\",\"", "\"\n\"", "'"};
  long records = 0, fields = 0, n_plain = 0, n_declined = 0, plain_texts = 0;
  // quote-free texts (and a share with a few quotes sprinkled in, which csv_plain_record must decline record by record)
  const char* plain_pieces[] = {",", ",", "\n", "\r", "\r\n", "a", "bc", " ", "1", "-2.5e3", "12345678", "'", ",,", "\n\n", "\xfa", "+"};
  for (long it = 0; it < iters * 4; ++it) {
    std::string t = "h\n";
    const int np = 1 + (int)(rng() % 200);
    for (int i = 0; i < np; ++i) t += plain_pieces[rng() % (sizeof(plain_pieces) / sizeof(plain_pieces[0]))];
    const std::vector<uint64_t> seq = starts_sequential(t);
    if (seq != starts_plain_masks(t)) {
      printf("MISMATCH terminator-mask record starts on quote-free text '%s'\n", t.c_str());
      return 1;
    }
    ++plain_texts;
    if (it % 4 == 3) {  // quotes in the middle of fields are literals for the automaton's boundaries only when not at a field start:
      std::string q = t;  // put them after a letter, so the record structure the walker sees stays what the masks assume
      for (size_t i = 2; i + 1 < q.size(); ++i)
        if (q[i] == 'a' && rng() % 6 == 0) q.insert(i + 1, "\"");
      if (!check_plain_records(q, starts_sequential(q), &n_plain, &n_declined)) return 1;
    } else if (!check_plain_records(t, seq, &n_plain, &n_declined)) {
      return 1;
    }
  }
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
  printf("ok: %ld records, %ld fields agree with the oracle; parallel == sequential boundaries on %ld texts; quote-free: mask starts == sequential on "
         "%ld texts, csv_plain_record == walker on %ld records (%ld declined as it must)\n", records, fields, iters, plain_texts, n_plain, n_declined);
  return 0;
}
