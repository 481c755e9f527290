// dfx_k_csv.hip -- CSV text -> Arrow columns on the device (the CsvDataSource that
// feeds every reference test and example, src/execution/datasource.rs:33-58).
//
//   record boundaries   The CSV automaton (dfx_csv_walk.hpp) is not a quote-parity problem: a quote is special only
//                       as the first byte of a field.  So boundaries come from an exact PARALLEL simulation: every
//                       32-byte chunk is summarised as a transition vector (end state for each of the 5 start
//                       states, 15 bits), vectors compose associatively, a scan over chunks / tiles gives every
//                       chunk its true start state, and a replay marks the bytes where a record starts
//                       (k_csv_tile_trans -> k_csv_tile_scan -> k_csv_mark<count> -> scan -> k_csv_mark<write>).
//   cells               one thread per record walks its bytes once and converts every cell: integers and floats with
//                       Rust's `str::parse` semantics (dfx_numparse.hpp: Eisel-Lemire, correctly rounded), booleans,
//                       Utf8 lengths; validity bitmaps are wave ballots (k_csv_parse); Utf8 bytes are gathered by a
//                       second walk after the offset scan (k_csv_utf8_gather).
// Bound: HBM reads of the text (3 boundary passes + 1..2 cell passes); the cell walk is byte-serial per record, so
// it is latency- rather than bandwidth-bound for now -- ingest is PCIe-bound (63 GB/s) long before that matters.
#include "dfx_csv_walk.hpp"
#include "dfx_kernels_inl.hpp"
#include "dfx_launch.hpp"
#include "dfx_numparse.hpp"

namespace dfx {

constexpr int kCsvChunk = 32;                      // bytes per thread
constexpr int kCsvBlock = 256;                     // threads per tile
constexpr int kCsvTile = kCsvChunk * kCsvBlock;    // 8192 bytes

// one thread's 32 bytes
struct CsvChunk {
  uint32_t w[8];
  int m;           // valid bytes (0..32)
  bool has_quote;  // any '"' among the valid bytes (false also when m < 32: those chunks take the general path)
};

// exact per-byte "== c" flags of a 32-bit word, gathered into 4 bits (bit k: byte k)
DEV uint32_t csv_eq_nibble(uint32_t w, uint32_t c4) {
  const uint32_t x = w ^ c4;
  const uint32_t t = ((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x;  // bit 7 of every byte: byte != 0
  const uint32_t z = ~(t | 0x7F7F7F7Fu) >> 7;                // bits 0, 8, 16, 24: byte == 0
  return ((z * 0x00204081u) >> 21) & 0xFu;
}

DEV CsvChunk csv_load_chunk(const uint8_t* __restrict__ buf, uint64_t pos, uint64_t n) {
  CsvChunk c;
  c.m = pos >= n ? 0 : (int)((n - pos) < (uint64_t)kCsvChunk ? (n - pos) : (uint64_t)kCsvChunk);
  c.has_quote = false;
  if (c.m == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) c.w[i] = 0;
    return c;
  }
  const uint4 a = *(const uint4*)(buf + pos);  // the buffer is padded to a multiple of 64 bytes
  const uint4 b = *(const uint4*)(buf + pos + 16);
  c.w[0] = a.x; c.w[1] = a.y; c.w[2] = a.z; c.w[3] = a.w; c.w[4] = b.x; c.w[5] = b.y; c.w[6] = b.z; c.w[7] = b.w;
  uint32_t q = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) q |= csv_eq_nibble(c.w[i], 0x22222222u);
  c.has_quote = q != 0;
  return c;
}

DEV uint8_t csv_chunk_byte(const CsvChunk& c, int i) { return (uint8_t)(c.w[i >> 2] >> ((i & 3) * 8)); }

// bit i: byte i is a record terminator (\n or \r) -- only meaningful for the general bytes of a full chunk
DEV uint32_t csv_chunk_tmask(const CsvChunk& c) {
  uint32_t t = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) t |= (csv_eq_nibble(c.w[i], 0x0A0A0A0Au) | csv_eq_nibble(c.w[i], 0x0D0D0D0Du)) << (4 * i);
  return t;
}

// transition vector of the chunk
DEV uint32_t csv_chunk_vector(const CsvChunk& c) {
  if (c.m == kCsvChunk && !c.has_quote) {
    // no quote in the chunk: every start state but InQuoted ends in the state the LAST byte dictates
    // (D -> StartField, T -> StartRecord, other -> InField); InQuoted stays InQuoted
    const uint32_t e = csv_tv_apply(csv_tv_of(csv_class(csv_chunk_byte(c, kCsvChunk - 1))), 2u);
    return csv_pack5(e, e, e, 3u, e);
  }
  uint32_t v = kCsvTvId;
#pragma unroll
  for (int i = 0; i < kCsvChunk; ++i)
    if (i < c.m) v = csv_tv_compose(v, csv_tv_of(csv_class(csv_chunk_byte(c, i))));
  return v;
}

// bit i: byte i starts a record, given the state before the chunk
DEV uint32_t csv_chunk_starts(const CsvChunk& c, uint32_t s) {
  if (c.m == kCsvChunk && !c.has_quote) {
    if (s == 3u) return 0u;  // the whole chunk is inside a quoted field
    const uint32_t t = csv_chunk_tmask(c);
    return ~t & ((t << 1) | (s == 0u ? 1u : 0u));  // a non-terminator right after a terminator / in StartRecord
  }
  uint32_t starts = 0;
  for (int i = 0; i < c.m; ++i) {
    const uint32_t cls = csv_class(csv_chunk_byte(c, i));
    if (s == 0u && cls != CSV_T) starts |= 1u << i;
    s = csv_tv_apply(csv_tv_of(cls), s);
  }
  return starts;
}

// exclusive scan of the transition vectors of a 256-thread block; returns the prefix of this thread and the total
DEV uint32_t csv_block_scan(uint32_t v, uint32_t* lds_wave_tot, uint32_t* total) {
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(inc, d, 64);
    if (lane >= d) inc = csv_tv_compose(o, inc);
  }
  if (lane == 63) lds_wave_tot[wave] = inc;
  __syncthreads();
  uint32_t excl = __shfl_up(inc, 1, 64);
  if (lane == 0) excl = kCsvTvId;
  uint32_t pre = kCsvTvId, tot = kCsvTvId;
#pragma unroll
  for (int w = 0; w < kCsvBlock / 64; ++w) {
    if (w < wave) pre = csv_tv_compose(pre, lds_wave_tot[w]);
    tot = csv_tv_compose(tot, lds_wave_tot[w]);
  }
  *total = tot;
  __syncthreads();
  return csv_tv_compose(pre, excl);
}

__global__ __launch_bounds__(kCsvBlock) void k_csv_tile_trans(const uint8_t* __restrict__ buf, uint64_t n,
                                                             uint32_t* __restrict__ tile_trans) {
  __shared__ uint32_t wave_tot[kCsvBlock / 64];
  const uint64_t pos = (uint64_t)blockIdx.x * kCsvTile + (uint64_t)threadIdx.x * kCsvChunk;
  const uint32_t v = csv_chunk_vector(csv_load_chunk(buf, pos, n));
  uint32_t total;
  (void)csv_block_scan(v, wave_tot, &total);
  if (threadIdx.x == 0) tile_trans[blockIdx.x] = total;
}

// one workgroup: state at the start of every tile (the file starts in StartRecord)
__global__ __launch_bounds__(1024) void k_csv_tile_scan(const uint32_t* __restrict__ tile_trans, int64_t n_tiles,
                                                         uint8_t* __restrict__ tile_state) {
  __shared__ uint32_t part[1024];
  const int64_t per = (n_tiles + 1023) / 1024;
  const int64_t t0 = (int64_t)threadIdx.x * per;
  const int64_t t1 = t0 + per < n_tiles ? t0 + per : n_tiles;
  uint32_t v = kCsvTvId;
  for (int64_t t = t0; t < t1; ++t) v = csv_tv_compose(v, tile_trans[t]);
  part[threadIdx.x] = v;
  __syncthreads();
  if (threadIdx.x == 0) {  // 1024 sequential compositions: ~10 us
    uint32_t acc = kCsvTvId;
    for (int i = 0; i < 1024; ++i) {
      const uint32_t x = part[i];
      part[i] = acc;
      acc = csv_tv_compose(acc, x);
    }
  }
  __syncthreads();
  uint32_t s = csv_tv_apply(part[threadIdx.x], 0u);
  for (int64_t t = t0; t < t1; ++t) {
    tile_state[t] = (uint8_t)s;
    s = csv_tv_apply(tile_trans[t], s);
  }
}

// WRITE == false: record starts per tile -> tile_counts.  WRITE == true: their byte positions -> row_start.
template <bool WRITE>
__global__ __launch_bounds__(kCsvBlock) void k_csv_mark(const uint8_t* __restrict__ buf, uint64_t n,
                                                       const uint8_t* __restrict__ tile_state,
                                                       uint32_t* __restrict__ tile_counts,
                                                       const uint64_t* __restrict__ tile_offsets,
                                                       uint64_t* __restrict__ row_start) {
  __shared__ uint32_t wave_tot[kCsvBlock / 64];
  __shared__ uint32_t wave_cnt[kCsvBlock / 64];
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const uint64_t pos = (uint64_t)blockIdx.x * kCsvTile + (uint64_t)threadIdx.x * kCsvChunk;
  const CsvChunk chunk = csv_load_chunk(buf, pos, n);
  const uint32_t v = csv_chunk_vector(chunk);
  uint32_t total;
  const uint32_t pre = csv_block_scan(v, wave_tot, &total);
  // replay with the true start state: a record starts at a non-terminator byte met in state StartRecord
  const uint32_t starts = csv_chunk_starts(chunk, csv_tv_apply(pre, (uint32_t)tile_state[blockIdx.x]));
  const uint32_t cnt = (uint32_t)__popc(starts);
  uint32_t inc = cnt;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(inc, d, 64);
    if (lane >= d) inc += o;
  }
  if (lane == 63) wave_cnt[wave] = inc;
  __syncthreads();
  uint32_t base = 0, tile_total = 0;
#pragma unroll
  for (int w = 0; w < kCsvBlock / 64; ++w) {
    if (w < wave) base += wave_cnt[w];
    tile_total += wave_cnt[w];
  }
  if (!WRITE) {
    if (threadIdx.x == 0) tile_counts[blockIdx.x] = tile_total;
  } else {
    uint64_t at = tile_offsets[blockIdx.x] + base + inc - cnt;
    uint32_t b = starts;
    while (b) {
      const int i = __ffs((int)b) - 1;
      row_start[at++] = pos + (uint64_t)i;
      b &= b - 1;
    }
  }
}

// ---- cells -----------------------------------------------------------------------------------------
DEV bool csv_bytes_equal(const uint8_t* p, uint64_t n, const char* lit, uint64_t m) {
  if (n != m) return false;
  for (uint64_t i = 0; i < m; ++i)
    if (p[i] != (uint8_t)lit[i]) return false;
  return true;
}

// fields of record 0 (the header the reference always consumes): the count every later record must match
__global__ void k_csv_count_fields(const uint8_t* __restrict__ buf, const uint64_t* __restrict__ row_start,
                                   int64_t row, uint32_t* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  out[0] = (uint32_t)csv_walk_record(buf, row_start[row], row_start[row + 1], [](int, const CsvField&) {});
}

// Two phases, so that the expensive part runs convergently:
//   A  every lane walks its record once and only RECORDS where its cells are (LDS: content offset + length/flags per
//      column) -- lanes reach their cell ends at different bytes, but the walk itself is a few instructions per byte;
//   B  a wave-uniform loop over the columns: all 64 lanes convert the cell of the SAME column together (one dtype,
//      one code path).  Converting inside the walk ran the conversions one lane at a time (measured 6.7x slower).
__global__ __launch_bounds__(kBlock) void k_csv_parse(const uint8_t* __restrict__ buf,
                                                     const uint64_t* __restrict__ row_start, int64_t r0, int64_t nb,
                                                     const DevCsvPlan plan) {
  extern __shared__ uint32_t cell_lds[];  // [n_cols][kBlock] content offset, then [n_cols][kBlock] ulen | quoted << 30 | complex << 31
  uint32_t* cell_off = cell_lds;
  uint32_t* cell_len = cell_lds + (size_t)plan.n_cols * kBlock;
  const int64_t tid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const bool inb = tid < nb;
  const int lane = lane_id();
  uint64_t err = ~0ull;
  uint64_t begin = 0;
  int nf = 0;
  if (inb) {
    begin = row_start[r0 + tid];
    const uint64_t limit = row_start[r0 + tid + 1];
    const int n_cols = plan.n_cols;
    nf = csv_walk_record(buf, begin, limit, [&](int fi, const CsvField& f) {
      if (fi >= n_cols) return;
      uint64_t cb, ce;
      csv_field_span(f, &cb, &ce);
      (void)ce;
      cell_off[fi * kBlock + threadIdx.x] = (uint32_t)(cb - begin);
      cell_len[fi * kBlock + threadIdx.x] = (f.ulen & 0x3FFFFFFFu) | (f.quoted ? 0x40000000u : 0u) | (f.complex ? 0x80000000u : 0u);
    });
    if ((uint32_t)nf != plan.expected_fields) {  // csv crate, flexible == false: UnequalLengths
      const uint64_t e = csv_err_pack(3, 0, r0 + tid);
      err = e < err ? e : err;
    }
  }
  for (int c = 0; c < plan.n_cols; ++c) {  // wave-uniform: one column, one dtype for all lanes
    const DevCsvCol col = plan.col[c];
    if (col.dtype == T_NONE) continue;  // projection push-down: nobody reads this column
    const bool have = inb && c < nf;  // a record shorter than the schema: `rows[i].get(col)` is None
    const uint32_t off = have ? cell_off[c * kBlock + threadIdx.x] : 0u;
    const uint32_t lw = have ? cell_len[c * kBlock + threadIdx.x] : 0u;
    const uint32_t ulen = lw & 0x3FFFFFFFu;
    if (col.dtype == T_UTF8) {  // Some(s) => append_string(s): never null ("" when the record is short)
      if (inb) col.lens[tid] = (int32_t)ulen;
      continue;
    }
    bool valid = false;
    bool bval = false;
    if (have && ulen != 0) {  // `Some(s) if s.len() > 0` else append_null
      // a number cannot contain a quote: "12"3 (-> 123 in the csv crate) is rejected here
      int rc = (lw & 0x80000000u) ? NP_INVALID : NP_OK;
      const uint8_t* s = buf + begin + off;
      const int64_t sl = (int64_t)ulen;  // contiguous content: its length is the unescaped length
      if (rc == NP_OK) {
        switch (col.dtype) {
          case T_F64: {
            double d = 0;
            rc = np_parse_f64(s, sl, &d);
            ((double*)col.values)[tid] = d;
            break;
          }
          case T_F32: {
            float d = 0;
            rc = np_parse_f32(s, sl, &d);
            ((float*)col.values)[tid] = d;
            break;
          }
          case T_BOOL: {
            if (csv_bytes_equal(s, (uint64_t)sl, "true", 4)) bval = true;
            else if (!csv_bytes_equal(s, (uint64_t)sl, "false", 5)) rc = NP_INVALID;
            break;
          }
          default: {
            uint64_t v = 0;
            const int bits = (col.dtype == T_I8 || col.dtype == T_U8) ? 8 : (col.dtype == T_I16 || col.dtype == T_U16) ? 16
                             : (col.dtype == T_I32 || col.dtype == T_U32) ? 32 : 64;
            rc = np_parse_int(s, sl, bits, is_signed_int(col.dtype), &v);
            store_typed(col.dtype, col.values, tid, v);
            break;
          }
        }
      }
      if (rc == NP_OK) valid = true;
      else {
        const uint64_t e = csv_err_pack(rc == NP_INVALID ? 1 : 2, c, r0 + tid);
        err = e < err ? e : err;
      }
    }
    if (inb && !valid && col.dtype != T_BOOL) store_typed(col.dtype, col.values, tid, 0ull);
    const uint64_t vm = __ballot(valid);
    const bool wave_inb = (tid & ~63ll) < nb;  // waves of the last workgroup that lie wholly past nb own no bitmap word
    if (lane == 0 && wave_inb) {
      col.validity[tid >> 6] = vm;
      const int64_t rows_here = nb - (tid & ~63ll) < 64 ? nb - (tid & ~63ll) : 64;
      const int nulls = (int)rows_here - __popcll(vm);
      if (nulls > 0) atomicAdd((unsigned long long*)&plan.null_counts[c], (unsigned long long)nulls);
    }
    if (col.dtype == T_BOOL) {
      const uint64_t bm = __ballot(valid && bval);
      if (lane == 0 && wave_inb) ((uint64_t*)col.values)[tid >> 6] = bm;
    }
  }
  if (err != ~0ull) atomicMin((unsigned long long*)plan.err, (unsigned long long)err);
}

// Utf8 column `field`: unescaped bytes of every record's cell -> out + offsets[row]
__global__ __launch_bounds__(kBlock) void k_csv_utf8_gather(const uint8_t* __restrict__ buf,
                                                           const uint64_t* __restrict__ row_start, int64_t r0,
                                                           int64_t nb, int field, const int32_t* __restrict__ offsets,
                                                           uint8_t* __restrict__ out) {
  const int64_t tid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (tid >= nb) return;
  uint8_t* dst = out + offsets[tid];
  csv_walk_record(buf, row_start[r0 + tid], row_start[r0 + tid + 1], [&](int fi, const CsvField& f) {
    if (fi == field) csv_copy_field(buf, f, dst);
  });
}

// ---- host launchers -----------------------------------------------------------------------------------
hipError_t launch_csv_boundaries_count(const uint8_t* buf, uint64_t n, uint32_t* tile_trans, uint8_t* tile_state,
                                       uint32_t* tile_counts, hipStream_t s) {
  const int64_t n_tiles = (int64_t)((n + kCsvTile - 1) / kCsvTile);
  if (n_tiles <= 0) return hipSuccess;
  Scope sc(KID_CSV, s, (double)n * 2);
  hipLaunchKernelGGL(k_csv_tile_trans, dim3((unsigned)n_tiles), dim3(kCsvBlock), 0, s, buf, n, tile_trans);
  hipLaunchKernelGGL(k_csv_tile_scan, dim3(1), dim3(1024), 0, s, tile_trans, n_tiles, tile_state);
  hipLaunchKernelGGL(k_csv_mark<false>, dim3((unsigned)n_tiles), dim3(kCsvBlock), 0, s, buf, n, tile_state, tile_counts,
                     (const uint64_t*)nullptr, (uint64_t*)nullptr);
  return hipGetLastError();
}

hipError_t launch_csv_boundaries_write(const uint8_t* buf, uint64_t n, const uint8_t* tile_state,
                                       const uint64_t* tile_offsets, uint64_t* row_start, hipStream_t s) {
  const int64_t n_tiles = (int64_t)((n + kCsvTile - 1) / kCsvTile);
  if (n_tiles <= 0) return hipSuccess;
  Scope sc(KID_CSV, s, (double)n);
  hipLaunchKernelGGL(k_csv_mark<true>, dim3((unsigned)n_tiles), dim3(kCsvBlock), 0, s, buf, n, tile_state,
                     (uint32_t*)nullptr, tile_offsets, row_start);
  return hipGetLastError();
}

int64_t csv_tile_bytes() { return kCsvTile; }

hipError_t launch_csv_count_fields(const uint8_t* buf, const uint64_t* row_start, int64_t row, uint32_t* out,
                                   hipStream_t s) {
  hipLaunchKernelGGL(k_csv_count_fields, dim3(1), dim3(64), 0, s, buf, row_start, row, out);
  return hipGetLastError();
}

hipError_t launch_csv_parse(const uint8_t* buf, const uint64_t* row_start, int64_t r0, int64_t nb,
                            const DevCsvPlan& plan, double algo_bytes, hipStream_t s) {
  if (nb <= 0) return hipSuccess;
  Scope sc(KID_CSV, s, algo_bytes);
  const int64_t blocks = (nb + kBlock - 1) / kBlock;
  const size_t lds = (size_t)plan.n_cols * kBlock * 2 * sizeof(uint32_t);  // <= 32 columns: 64 KB
  hipLaunchKernelGGL(k_csv_parse, dim3((unsigned)blocks), dim3(kBlock), lds, s, buf, row_start, r0, nb, plan);
  return hipGetLastError();
}

hipError_t launch_csv_utf8_gather(const uint8_t* buf, const uint64_t* row_start, int64_t r0, int64_t nb, int field,
                                  const int32_t* offsets, uint8_t* out, hipStream_t s) {
  if (nb <= 0) return hipSuccess;
  Scope sc(KID_CSV, s, 0);
  const int64_t blocks = (nb + kBlock - 1) / kBlock;
  hipLaunchKernelGGL(k_csv_utf8_gather, dim3((unsigned)blocks), dim3(kBlock), 0, s, buf, row_start, r0, nb, field,
                     offsets, out);
  return hipGetLastError();
}

}  // namespace dfx
