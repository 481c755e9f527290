// dfx_k_csv.hip -- CSV text -> Arrow columns on the device (the CsvDataSource that
// feeds every reference test and example, src/execution/datasource.rs:33-58).
//
//   record boundaries   The CSV automaton (dfx_csv_walk.hpp) is not a quote-parity problem: a quote is special only
//                       as the first byte of a field.  So boundaries come from an exact PARALLEL simulation: every
//                       32-byte chunk is summarised as a transition vector (end state for each of the 5 start
//                       states, 15 bits), vectors compose associatively, a scan over chunks / tiles gives every
//                       chunk its true start state, and a replay marks the bytes where a record starts
//                       (k_csv_tile_trans -> k_csv_tile_scan_{totals,blocks,apply} -> k_csv_count_unknown -> scan ->
//                       k_csv_mark_write).  A tile without a quote needs no replay: its record count follows from its own
//                       bytes and the state it starts in, so only tiles with quotes are read a second time before the write.
//   cells               one wave per tile of 64 records (k_csv_parse): the tile's text goes to LDS with coalesced loads, SWAR
//                       masks number its delimiters and record ends, and the wave converts one column at a time out of
//                       LDS: integers and floats with Rust's `str::parse` semantics (dfx_numparse.hpp: eight bytes at a
//                       time, Eisel-Lemire, correctly rounded), booleans, Utf8 lengths; validity bitmaps are wave ballots.
//                       Tiles with quotes or ragged records take a per-lane walk.  Utf8 bytes are gathered by a second
//                       walk after the offset scan (k_csv_utf8_gather).
// Bound: HBM reads of the text (2 boundary passes, 3 where there are quotes, + 1..2 cell passes); ingest is PCIe-bound
// (63 GB/s) long before that matters.
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
DEV uint32_t csv_chunk_tmask(const CsvChunk& c) { return csv_tmask32(c.w); }

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

// exclusive scan of the transition vectors of a BLOCK-thread workgroup; returns the prefix of this thread and the total
template <int BLOCK = kCsvBlock>
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
  for (int w = 0; w < BLOCK / 64; ++w) {
    if (w < wave) pre = csv_tv_compose(pre, lds_wave_tot[w]);
    tot = csv_tv_compose(tot, lds_wave_tot[w]);
  }
  *total = tot;
  __syncthreads();
  return csv_tv_compose(pre, excl);
}

// The word a tile leaves for the scan: its transition vector (15 bits) and -- when no chunk of the tile holds a quote and the
// tile is whole -- the number of records that start in it, counted as if its first byte did not follow a terminator, plus
// whether that first byte could start a record.  Without a quote the state before a byte is a function of the byte before it,
// so the count needs no scan; kCsvTileUnknown marks the tiles whose starts k_csv_count_unknown has to count by replay.
constexpr uint32_t kCsvTileVec = 0x7FFFu;
constexpr uint32_t kCsvTileUnknown = 1u << 15;
constexpr uint32_t kCsvTileFirstNonT = 1u << 16;
constexpr int kCsvTileCountShift = 17;

__global__ __launch_bounds__(kCsvBlock) void k_csv_tile_trans(const uint8_t* __restrict__ buf, uint64_t n,
                                                             uint32_t* __restrict__ tile_trans) {
  __shared__ uint32_t wave_tot[kCsvBlock / 64];
  __shared__ uint32_t wave_last[kCsvBlock / 64];
  __shared__ uint32_t wave_cnt[kCsvBlock / 64];
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const uint64_t pos = (uint64_t)blockIdx.x * kCsvTile + (uint64_t)threadIdx.x * kCsvChunk;
  const CsvChunk chunk = csv_load_chunk(buf, pos, n);
  const bool plain = chunk.m == kCsvChunk && !chunk.has_quote;
  const uint32_t t = csv_chunk_tmask(chunk);
  if (lane == 63) wave_last[wave] = t >> 31;
  if (__syncthreads_and(plain ? 1 : 0)) {
    uint32_t after = (uint32_t)__shfl_up((int)(t >> 31), 1, 64);
    if (lane == 0) after = wave == 0 ? 0u : wave_last[wave - 1];
    uint32_t cnt = (uint32_t)__popc(csv_plain_starts(t, after != 0u));
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) cnt += (uint32_t)__shfl_xor((int)cnt, d, 64);
    if (lane == 0) wave_cnt[wave] = cnt;
    if (threadIdx.x == 0) wave_tot[0] = (t & 1u) ? 0u : kCsvTileFirstNonT;
    __syncthreads();
    if (threadIdx.x == kCsvBlock - 1) {
      uint32_t total = 0;
#pragma unroll
      for (int w = 0; w < kCsvBlock / 64; ++w) total += wave_cnt[w];
      // every start state but InQuoted ends in the state the LAST byte dictates; InQuoted stays InQuoted
      const uint32_t e = csv_tv_apply(csv_tv_of(csv_class(csv_chunk_byte(chunk, kCsvChunk - 1))), 2u);
      tile_trans[blockIdx.x] = csv_pack5(e, e, e, 3u, e) | wave_tot[0] | (total << kCsvTileCountShift);
    }
    return;
  }
  const uint32_t v = csv_chunk_vector(chunk);
  uint32_t total;
  (void)csv_block_scan(v, wave_tot, &total);
  if (threadIdx.x == 0) tile_trans[blockIdx.x] = total | kCsvTileUnknown;
}

// State at the start of every tile (the file starts in StartRecord), three launches over blocks of 1024 tiles: the composed
// vector of every block, one workgroup that scans those (the states at the block starts), and the blocks again -- the state of
// every tile, and with it the record count of every quote-free tile.
__global__ __launch_bounds__(1024) void k_csv_tile_scan_totals(const uint32_t* __restrict__ tile_trans, int64_t n_tiles,
                                                                uint32_t* __restrict__ block_vec) {
  __shared__ uint32_t wave_tot[16];
  const int64_t t = (int64_t)blockIdx.x * 1024 + threadIdx.x;
  uint32_t total;
  (void)csv_block_scan<1024>(t < n_tiles ? tile_trans[t] & kCsvTileVec : kCsvTvId, wave_tot, &total);
  if (threadIdx.x == 0) block_vec[blockIdx.x] = total;
}

__global__ __launch_bounds__(1024) void k_csv_tile_scan_blocks(uint32_t* __restrict__ block_vec, int64_t n_blocks) {  // vectors in, states out
  __shared__ uint32_t wave_tot[16];
  uint32_t s = 0u;
  for (int64_t b0 = 0; b0 < n_blocks; b0 += 1024) {
    const int64_t b = b0 + threadIdx.x;
    uint32_t total;
    const uint32_t pre = csv_block_scan<1024>(b < n_blocks ? block_vec[b] : kCsvTvId, wave_tot, &total);
    if (b < n_blocks) block_vec[b] = csv_tv_apply(pre, s);
    s = csv_tv_apply(total, s);
  }
}

__global__ __launch_bounds__(1024) void k_csv_tile_scan_apply(const uint32_t* __restrict__ tile_trans, int64_t n_tiles,
                                                               const uint32_t* __restrict__ block_state,
                                                               uint8_t* __restrict__ tile_state, uint32_t* __restrict__ tile_counts) {
  __shared__ uint32_t wave_tot[16];
  const int64_t t = (int64_t)blockIdx.x * 1024 + threadIdx.x;
  const uint32_t x = t < n_tiles ? tile_trans[t] : kCsvTvId;
  uint32_t total;
  const uint32_t pre = csv_block_scan<1024>(x & kCsvTileVec, wave_tot, &total);
  if (t >= n_tiles) return;
  const uint32_t s = csv_tv_apply(pre, block_state[blockIdx.x]);
  tile_state[t] = (uint8_t)s;
  // records that start in a quote-free tile: none inside a quoted field, else the tile's own count (+ its first byte when the
  // tile begins in StartRecord); tiles with quotes are counted by k_csv_count_unknown
  if (!(x & kCsvTileUnknown))
    tile_counts[t] = s == 3u ? 0u : (x >> kCsvTileCountShift) + ((s == 0u && (x & kCsvTileFirstNonT)) ? 1u : 0u);
}

// Record starts of one tile, by the whole workgroup.  WRITE == false: their number -> tile_counts (the tiles the scan could
// not count: quotes, the ragged last tile).  WRITE == true: their byte positions -> row_start.  Quote-free tiles skip the
// transition vectors and their scan: a record starts at a non-terminator that follows a terminator (or the tile's StartRecord).
template <bool WRITE>
DEV void csv_mark_tile(int64_t tile, bool known, const uint8_t* __restrict__ buf, uint64_t n, const uint8_t* __restrict__ tile_state,
                       uint32_t* __restrict__ tile_counts, const uint64_t* __restrict__ tile_offsets,
                       uint64_t* __restrict__ row_start, uint32_t* wave_tot, uint32_t* wave_cnt) {
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const uint32_t s_tile = (uint32_t)tile_state[tile];
  if (known && s_tile == 3u) return;  // a quote-free tile inside a quoted field: no record starts here
  const uint64_t pos = (uint64_t)tile * kCsvTile + (uint64_t)threadIdx.x * kCsvChunk;
  const CsvChunk chunk = csv_load_chunk(buf, pos, n);
  uint32_t starts;
  if (known) {
    const uint32_t t = csv_chunk_tmask(chunk);
    uint32_t after = (uint32_t)__shfl_up((int)(t >> 31), 1, 64);
    if (lane == 0) {
      if (threadIdx.x == 0) after = s_tile == 0u ? 1u : 0u;
      else {
        const uint8_t c = buf[pos - 1];
        after = (c == '\n' || c == '\r') ? 1u : 0u;
      }
    }
    starts = csv_plain_starts(t, after != 0u);
  } else {
    const uint32_t v = csv_chunk_vector(chunk);
    uint32_t total;
    const uint32_t pre = csv_block_scan(v, wave_tot, &total);
    // replay with the true start state: a record starts at a non-terminator byte met in state StartRecord
    starts = csv_chunk_starts(chunk, csv_tv_apply(pre, s_tile));
  }
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
    if (threadIdx.x == 0) tile_counts[tile] = tile_total;
    __syncthreads();  // wave_cnt is written again for the workgroup's next tile
  } else {
    uint64_t at = tile_offsets[tile] + base + inc - cnt;
    uint32_t b = starts;
    while (b) {
      const int i = __ffs((int)b) - 1;
      row_start[at++] = pos + (uint64_t)i;
      b &= b - 1;
    }
  }
}

// every workgroup looks at kCsvBlock consecutive tiles (one word each) and counts the starts of those the scan left open
__global__ __launch_bounds__(kCsvBlock) void k_csv_count_unknown(const uint8_t* __restrict__ buf, uint64_t n,
                                                                const uint32_t* __restrict__ tile_trans, int64_t n_tiles,
                                                                const uint8_t* __restrict__ tile_state,
                                                                uint32_t* __restrict__ tile_counts) {
  __shared__ uint32_t wave_tot[kCsvBlock / 64];
  __shared__ uint32_t wave_cnt[kCsvBlock / 64];
  __shared__ uint64_t open_mask[kCsvBlock / 64];
  const int64_t first = (int64_t)blockIdx.x * kCsvBlock;
  const int64_t t = first + threadIdx.x;
  const uint64_t m = __ballot(t < n_tiles && (tile_trans[t < n_tiles ? t : 0] & kCsvTileUnknown) != 0u);
  if (lane_id() == 0) open_mask[threadIdx.x >> 6] = m;
  __syncthreads();
  for (int w = 0; w < kCsvBlock / 64; ++w) {
    uint64_t open = open_mask[w];  // (workgroup-uniform)
    while (open) {
      const int j = __ffsll((long long)open) - 1;
      open &= open - 1;
      csv_mark_tile<false>(first + w * 64 + j, false, buf, n, tile_state, tile_counts, nullptr, nullptr, wave_tot, wave_cnt);
    }
  }
}

__global__ __launch_bounds__(kCsvBlock) void k_csv_mark_write(const uint8_t* __restrict__ buf, uint64_t n,
                                                             const uint32_t* __restrict__ tile_trans,
                                                             const uint8_t* __restrict__ tile_state,
                                                             const uint64_t* __restrict__ tile_offsets,
                                                             uint64_t* __restrict__ row_start) {
  __shared__ uint32_t wave_tot[kCsvBlock / 64];
  __shared__ uint32_t wave_cnt[kCsvBlock / 64];
  csv_mark_tile<true>((int64_t)blockIdx.x, !(tile_trans[blockIdx.x] & kCsvTileUnknown), buf, n, tile_state, nullptr, tile_offsets,
                      row_start, wave_tot, wave_cnt);
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

// ---- k_csv_parse: one wave per tile of 64 consecutive records ---------------------------------------------
// The records of a tile are contiguous text [row_start[first], row_start[first + 64]).  The wave-cooperative path
// (every tile whose text has no quote, fits the wave's LDS window and has the header's field count in every record):
//   A  the wave copies the tile's text into LDS with coalesced 16-byte loads.  Then every lane reads ITS record out of LDS
//      eight bytes at a time: a SWAR mask of the delimiters gives the cell boundaries (to an LDS list, F entries per record
//      for `expected_fields` = F: F - 1 delimiters and the record's end), another says whether there is a quote.  Without a
//      quote a record is its bytes up to the terminators that precede the next record's start, so its end needs no search.
//      A lane whose record has a quote or not exactly F - 1 delimiters sends the tile down the general path below.
//      (Round 5 first numbered the structural bytes cooperatively -- per-16-byte masks, a wave prefix sum, a list in text
//      order: 1 390 vector instructions per tile, the kernel was VALU-bound; the per-lane walk over LDS words needs a third.)
//   B  a wave-uniform loop over the columns: all 64 lanes convert the cell of the SAME column together (one dtype, one
//      code path), reading its bytes from LDS.
// The general path (quotes, a tile longer than the window, a record with another field count -- which is the error the
// reference reports): every lane walks its own record once from global memory and only RECORDS where its cells are
// (LDS: content offset + length / flags per column), then phase B runs on those, reading global memory.  Converting
// inside the walk ran the conversions one lane at a time (measured 6.7x slower, round 1).
constexpr int kCsvTextCapMax = 16384;  // list entries are 16-bit offsets into the window

// phase B.  FAST: cells come from the structural list, bytes from the LDS window; else from cell_off / cell_len + global memory
template <bool FAST>
DEV void csv_convert_cells(const uint8_t* __restrict__ buf, const DevCsvPlan& plan, int64_t r0, int64_t nb, int64_t tid,
                           bool inb, int nf, uint64_t begin, uint32_t first_off, const uint8_t* text, const uint16_t* spos,
                           const uint32_t* cell_off, const uint32_t* cell_len, uint64_t& err) {
  const int lane = lane_id();
  const uint32_t F = plan.expected_fields;
  for (int c = 0; c < plan.n_cols; ++c) {  // wave-uniform: one column, one dtype for all lanes
    const DevCsvCol col = plan.col[c];
    if (col.dtype == T_NONE) continue;  // projection push-down: nobody reads this column
    const bool have = inb && c < nf;    // a record shorter than the schema: `rows[i].get(col)` is None
    uint32_t off = 0, lw = 0;
    if (have) {
      if (FAST) {
        off = c == 0 ? first_off : (uint32_t)spos[(uint32_t)lane * F + (uint32_t)c - 1u] + 1u;
        lw = (uint32_t)spos[(uint32_t)lane * F + (uint32_t)c] - off;
      } else {
        off = cell_off[c * 64 + lane];
        lw = cell_len[c * 64 + lane];
      }
    }
    const uint32_t ulen = lw & 0x3FFFFFFFu;
    if (col.dtype == T_UTF8) {  // Some(s) => append_string(s): never null ("" when the record is short)
      if (inb) {
        col.lens[tid] = (int32_t)ulen;
        // where the gather finds the bytes: the content's position in the text, or (bit 63) "walk the record": doubled quotes
        // or bytes after the closing quote make the content something other than a span of the text
        ((uint64_t*)col.values)[tid] = (lw & 0x80000000u) ? (1ull << 63) : (FAST ? begin - first_off : begin) + off;
      }
      continue;
    }
    bool valid = false;
    bool bval = false;
    if (have && ulen != 0) {  // `Some(s) if s.len() > 0` else append_null
      // a number cannot contain a quote: "12"3 (-> 123 in the csv crate) is rejected here
      int rc = (lw & 0x80000000u) ? NP_INVALID : NP_OK;
      const uint8_t* s = FAST ? text + off : buf + begin + off;
      const int64_t sl = (int64_t)ulen;  // contiguous content: its length is the unescaped length
      if (rc == NP_OK) {
        switch (col.dtype) {
          case T_F64: {
            double d = 0;
            rc = FAST ? np_parse_f64_w(s, sl, &d) : np_parse_f64(s, sl, &d);  // the window may be read 8 bytes past a cell
            ((double*)col.values)[tid] = d;
            break;
          }
          case T_F32: {
            float d = 0;
            rc = FAST ? np_parse_f32_w(s, sl, &d) : np_parse_f32(s, sl, &d);
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
            rc = FAST ? np_parse_int_w(s, sl, bits, is_signed_int(col.dtype), &v) : np_parse_int(s, sl, bits, is_signed_int(col.dtype), &v);
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
    if (lane == 0) {  // lane 0 of a wave that runs this is always a record of the batch
      col.validity[tid >> 6] = vm;
      const int64_t rows_here = nb - tid < 64 ? nb - tid : 64;
      const int nulls = (int)rows_here - __popcll(vm);
      if (nulls > 0) atomicAdd((unsigned long long*)&plan.null_counts[c], (unsigned long long)nulls);
    }
    if (col.dtype == T_BOOL) {
      const uint64_t bm = __ballot(valid && bval);
      if (lane == 0) ((uint64_t*)col.values)[tid >> 6] = bm;
    }
  }
}

__global__ __launch_bounds__(kBlock) void k_csv_parse(const uint8_t* __restrict__ buf,
                                                     const uint64_t* __restrict__ row_start, int64_t r0, int64_t nb,
                                                     const DevCsvPlan plan, uint32_t text_cap, uint32_t wave_lds) {
  extern __shared__ __attribute__((aligned(16))) uint8_t csv_lds[];
  const int lane = lane_id();
  uint8_t* wl = csv_lds + (size_t)(threadIdx.x >> 6) * wave_lds;  // this wave's room: no workgroup barrier below
  const int64_t tid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if ((tid & ~63ll) >= nb) return;  // a wave of the last workgroup that lies wholly past nb
  const bool inb = tid < nb;
  const int nrec = (int)(nb - (tid & ~63ll) < 64 ? nb - (tid & ~63ll) : 64);
  uint64_t err = ~0ull;
  uint64_t begin = 0, limit = 0;
  if (inb) {
    begin = row_start[r0 + tid];
    limit = row_start[r0 + tid + 1];
  }
  const uint32_t F = plan.expected_fields;
  const uint64_t tb = __shfl(begin, 0, 64);
  const uint64_t te = __shfl(limit, nrec - 1, 64);
  const uint64_t base = tb & ~15ull;  // the window starts 16-byte aligned
  const uint64_t span64 = te - base;
  bool fast = text_cap != 0 && span64 <= (uint64_t)text_cap;
  if (fast) {
    uint8_t* text = wl;
    uint16_t* spos = (uint16_t*)(wl + text_cap);
    const uint32_t span = (uint32_t)span64;
    for (uint32_t my = (uint32_t)lane * 16u; my < span; my += 1024u)
      *(uint4*)(text + my) = *(const uint4*)(buf + base + my);  // the text buffer is padded to a multiple of 64 bytes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    bool ok = true;
    if (inb) ok = csv_plain_record(text, (uint32_t)(begin - base), (uint32_t)(limit - base), F, spos + (uint32_t)lane * F);
    fast = __ballot(!ok) == 0ull;
    if (fast) {
      csv_convert_cells<true>(buf, plan, r0, nb, tid, inb, (int)F, begin, (uint32_t)(begin - base), text, spos, nullptr,
                              nullptr, err);
      if (err != ~0ull) atomicMin((unsigned long long*)plan.err, (unsigned long long)err);
      return;
    }
    __builtin_amdgcn_wave_barrier();  // the general path reuses the window
  }
  if (lane == 0) atomicAdd((unsigned long long*)plan.general_tiles, 1ull);
  uint32_t* cell_off = (uint32_t*)wl;  // [n_cols][64] content offset, then [n_cols][64] ulen | quoted << 30 | complex << 31
  uint32_t* cell_len = cell_off + (size_t)plan.n_cols * 64;
  int nf = 0;
  if (inb) {
    const int n_cols = plan.n_cols;
    nf = csv_walk_record(buf, begin, limit, [&](int fi, const CsvField& f) {
      if (fi >= n_cols) return;
      uint64_t cb, ce;
      csv_field_span(f, &cb, &ce);
      (void)ce;
      cell_off[fi * 64 + lane] = (uint32_t)(cb - begin);
      cell_len[fi * 64 + lane] = (f.ulen & 0x3FFFFFFFu) | (f.quoted ? 0x40000000u : 0u) | (f.complex ? 0x80000000u : 0u);
    });
    if ((uint32_t)nf != F) {  // csv crate, flexible == false: UnequalLengths
      const uint64_t e = csv_err_pack(3, 0, r0 + tid);
      err = e < err ? e : err;
    }
  }
  csv_convert_cells<false>(buf, plan, r0, nb, tid, inb, nf, begin, 0u, nullptr, nullptr, cell_off, cell_len, err);
  if (err != ~0ull) atomicMin((unsigned long long*)plan.err, (unsigned long long)err);
}

// Utf8 column `field`: unescaped bytes of every record's cell -> out + offsets[row].  k_csv_parse left the position of every
// cell whose content is a span of the text (`starts`): those are copied eight bytes at a time; the others are walked.
__global__ __launch_bounds__(kBlock) void k_csv_utf8_gather(const uint8_t* __restrict__ buf,
                                                           const uint64_t* __restrict__ row_start, int64_t r0,
                                                           int64_t nb, int field, const int32_t* __restrict__ offsets,
                                                           const uint64_t* __restrict__ starts, uint8_t* __restrict__ out) {
  const int64_t tid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (tid >= nb) return;
  const int32_t o = offsets[tid];
  uint8_t* dst = out + o;
  const uint64_t st = starts[tid];
  if (st >> 63) {
    csv_walk_record(buf, row_start[r0 + tid], row_start[r0 + tid + 1], [&](int fi, const CsvField& f) {
      if (fi == field) csv_copy_field(buf, f, dst);
    });
    return;
  }
  const uint8_t* src = buf + st;
  const int32_t len = offsets[tid + 1] - o;
  int32_t i = 0;
  for (; i + 8 <= len; i += 8) {
    uint64_t w;
    __builtin_memcpy(&w, src + i, 8);
    __builtin_memcpy(dst + i, &w, 8);
  }
  if (i + 4 <= len) {
    uint32_t w;
    __builtin_memcpy(&w, src + i, 4);
    __builtin_memcpy(dst + i, &w, 4);
    i += 4;
  }
  for (; i < len; ++i) dst[i] = src[i];
}

// ---- host launchers -----------------------------------------------------------------------------------
hipError_t launch_csv_boundaries_count(const uint8_t* buf, uint64_t n, uint32_t* tile_trans, uint32_t* block_vec, uint8_t* tile_state,
                                       uint32_t* tile_counts, hipStream_t s) {
  const int64_t n_tiles = (int64_t)((n + kCsvTile - 1) / kCsvTile);
  if (n_tiles <= 0) return hipSuccess;
  Scope sc(KID_CSV, s, (double)n);
  const int64_t n_blocks = (n_tiles + 1023) / 1024;  // block_vec: one word per 1024 tiles
  hipLaunchKernelGGL(k_csv_tile_trans, dim3((unsigned)n_tiles), dim3(kCsvBlock), 0, s, buf, n, tile_trans);
  hipLaunchKernelGGL(k_csv_tile_scan_totals, dim3((unsigned)n_blocks), dim3(1024), 0, s, (const uint32_t*)tile_trans, n_tiles, block_vec);
  hipLaunchKernelGGL(k_csv_tile_scan_blocks, dim3(1), dim3(1024), 0, s, block_vec, n_blocks);
  hipLaunchKernelGGL(k_csv_tile_scan_apply, dim3((unsigned)n_blocks), dim3(1024), 0, s, (const uint32_t*)tile_trans, n_tiles,
                     (const uint32_t*)block_vec, tile_state, tile_counts);
  hipLaunchKernelGGL(k_csv_count_unknown, dim3((unsigned)((n_tiles + kCsvBlock - 1) / kCsvBlock)), dim3(kCsvBlock), 0, s, buf, n,
                     (const uint32_t*)tile_trans, n_tiles, (const uint8_t*)tile_state, tile_counts);
  return hipGetLastError();
}

hipError_t launch_csv_boundaries_write(const uint8_t* buf, uint64_t n, const uint32_t* tile_trans, const uint8_t* tile_state,
                                       const uint64_t* tile_offsets, uint64_t* row_start, hipStream_t s) {
  const int64_t n_tiles = (int64_t)((n + kCsvTile - 1) / kCsvTile);
  if (n_tiles <= 0) return hipSuccess;
  Scope sc(KID_CSV, s, (double)n);
  hipLaunchKernelGGL(k_csv_mark_write, dim3((unsigned)n_tiles), dim3(kCsvBlock), 0, s, buf, n, tile_trans, tile_state, tile_offsets,
                     row_start);
  return hipGetLastError();
}

int64_t csv_tile_bytes() { return kCsvTile; }

hipError_t launch_csv_count_fields(const uint8_t* buf, const uint64_t* row_start, int64_t row, uint32_t* out,
                                   hipStream_t s) {
  hipLaunchKernelGGL(k_csv_count_fields, dim3(1), dim3(64), 0, s, buf, row_start, row, out);
  return hipGetLastError();
}

// per-wave LDS of k_csv_parse: the text window (1.5 x the average tile, 2..16 KB) + the structural list (64 F 16-bit entries),
// or the general path's cell table when that is larger.  wave_tiles == 0 (csv.wave_tiles, tests): the general path only.
hipError_t launch_csv_parse(const uint8_t* buf, const uint64_t* row_start, int64_t r0, int64_t nb,
                            const DevCsvPlan& plan, double avg_record_bytes, int wave_tiles, double algo_bytes, hipStream_t s) {
  if (nb <= 0) return hipSuccess;
  Scope sc(KID_CSV, s, algo_bytes);
  const int64_t blocks = (nb + kBlock - 1) / kBlock;
  uint32_t text_cap = 0;
  if (wave_tiles && plan.expected_fields >= 1 && plan.expected_fields <= 64) {
    const double want = avg_record_bytes * 64.0 * 1.5 + 64.0;
    text_cap = want > (double)kCsvTextCapMax ? (uint32_t)kCsvTextCapMax : (uint32_t)want;
    text_cap = (text_cap + 1023u) & ~1023u;
    if (text_cap < 2048u) text_cap = 2048u;
    const uint32_t room = (16384u - 128u * plan.expected_fields) & ~1023u;  // four waves within 64 KB of dynamic LDS
    if (text_cap > room) text_cap = room;
  }
  uint32_t wave_lds = text_cap ? text_cap + 128u * plan.expected_fields : 0u;
  const uint32_t general = (uint32_t)plan.n_cols * 64u * 2u * (uint32_t)sizeof(uint32_t);  // <= 32 columns: 16 KB
  if (wave_lds < general) wave_lds = general;
  wave_lds = (wave_lds + 15u) & ~15u;
  hipLaunchKernelGGL(k_csv_parse, dim3((unsigned)blocks), dim3(kBlock), (size_t)wave_lds * (kBlock / 64), s, buf, row_start, r0,
                     nb, plan, text_cap, wave_lds);
  return hipGetLastError();
}

hipError_t launch_csv_utf8_gather(const uint8_t* buf, const uint64_t* row_start, int64_t r0, int64_t nb, int field,
                                  const int32_t* offsets, const uint64_t* starts, uint8_t* out, hipStream_t s) {
  if (nb <= 0) return hipSuccess;
  Scope sc(KID_CSV, s, 0);
  const int64_t blocks = (nb + kBlock - 1) / kBlock;
  hipLaunchKernelGGL(k_csv_utf8_gather, dim3((unsigned)blocks), dim3(kBlock), 0, s, buf, row_start, r0, nb, field,
                     offsets, starts, out);
  return hipGetLastError();
}

}  // namespace dfx
