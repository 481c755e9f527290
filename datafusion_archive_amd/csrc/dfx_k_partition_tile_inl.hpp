// dfx_k_partition_tile_inl.hpp -- pass 1 of the partitioned GROUP BY, TILE-SORTED flavour (PTF_TILE): dense scans.
//
// The ring kernels (dfx_k_partition_inl.hpp, dfx_k_partition_ws_inl.hpp) write-combine routed rows through per-partition LDS
// rings that any wave may append to at any time; the price is a protocol of dependent LDS round trips per routed batch of 64
// rows -- fill atomic -> generation word -> ring row -> commit atomic -> job list -> ring read -> store, ~16 LDS
// instructions, spin loops on busy chunk slots -- which a SELECTIVE scan hides behind its column loads (a fifth of the rows
// is routed) and a DENSE scan does not: config 3 (no predicate, every row routed) spent 414 us per 2^26-row launch where its
// traffic (1.07 GB read + 0.81 GB written) asks for 300.
//
// When most rows are routed there is no need for a protocol: a workgroup takes a TILE of rows (16 waves x U row groups x 64
// = 8192 rows at U = 8), counting-sorts it by partition in LDS and copies it out in sorted order -- adjacent lanes write
// adjacent rows of the same region, runs of tile / n_parts rows (32 rows = 384 bytes at 256 partitions).  Per routed row:
// ONE LDS atomic (its rank inside the tile's partition), one LDS write and one LDS read of the row, two small table reads;
// two workgroup barriers per tile, none of which waits for global memory (s_waitcnt lgkmcnt(0) + s_barrier: HIP's
// __syncthreads() would also wait for the NEXT tile's column loads, which are in flight across the whole tile):
//
//   P1(t)   evaluate the rows of tile t (predicate, key -> hash image -> partition, operand); rank = hist[t & 1][part]++
//   -- B1 -- (every rank of tile t is taken; the sorted buffer of tile t - 1 has been copied out)
//   P2(t)   every wave, redundantly (identical values, so no barrier between this and its own later reads): exclusive scan
//           of the tile's histogram -> off[]; delta[] = fill - off (region row = sorted position + delta);
//           fill[(t + 1) & 1] = fill[t & 1] + count; wave 0 clears hist[(t + 1) & 1]
//   P3(t)   sorted[off[part] + rank] = row
//   -- B2 -- (tile t is sorted)
//   hand-over: the columns of tile t + 1 (loaded one tile ago) become current, the loads of tile t + 2 are issued
//   P4(t)   lane-contiguous copy-out: sorted position s -> region row s + delta[part of the row]; 12-byte rows {hash image,
//           operand} (PTF_NARROW) or 16-byte rows {key, operand}
//   P1(t+1) ...   (the loop is rotated: hand-over | P4(t - 1) | P1(t) | B1 | P2 | P3 | B2 -- one copy of P1, see below)
// Same regions, counts and padding as the ring flavours (a region is padded to a whole 16-row chunk at the end of the launch),
// so pass 2 -- and a later launch of another flavour that resumes the window -- cannot tell.  Rows the routed form cannot
// carry (narrow mode: a key >= 2^32 or a reserved image; the claim sentinel; region overflow) take the spill list as everywhere.
// One key word, one routed value; no hot-key pairs (skewed streams keep the ring kernel with PTF_HOT).
#pragma once
#include "dfx_k_partition_inl.hpp"

namespace dfx {

constexpr int kTileBlock = 1024;
constexpr uint32_t kTileMaxParts = 1024;

// LDS barrier that does not wait for global memory (see above)
DEV void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

inline size_t partition_tile_bytes(uint32_t n_parts, int u, bool wide) {
  const size_t rows = (size_t)(kTileBlock / 64) * (size_t)u * 64;
  return rows * (wide ? 18 : 12) + (size_t)n_parts * 4 * 6 + 64;
}

template <typename POL, int WIDE>
__global__ __launch_bounds__(kTileBlock) void k_partition_tile(const DevProgram P, const DevFastPlan F, const DevColumns C,
                                                              const DevAggPlan plan, const DevTable T,
                                                              const DevPartition PT, const DevRows spill, const int64_t n) {
  typedef typename POL::COLV COLV;
  constexpr int U = POL::U;
  constexpr int NWAVES = kTileBlock / 64;
  constexpr uint32_t TILE = (uint32_t)NWAVES * U * 64;
  extern __shared__ __attribute__((aligned(16))) uint64_t lds[];
  // sorted buffer (structure of arrays: every plane is written at random positions, read lane-contiguously)
  uint64_t* const s_val = lds;                                                  // [TILE]
  uint64_t* const s_key = lds + TILE;                                           // [TILE]   (WIDE)
  uint32_t* const s_tag = (uint32_t*)(lds + TILE);                              // [TILE]   (narrow: the hash image)
  uint16_t* const s_part = (uint16_t*)(lds + 2 * (size_t)TILE);                 // [TILE]   (WIDE: the partition, not derivable without re-hashing)
  uint32_t* const tab = WIDE ? (uint32_t*)(lds + 2 * (size_t)TILE + TILE / 4) : s_tag + TILE;
  const uint32_t NP = PT.n_parts;
  uint32_t* const hist = tab;            // [2][NP]
  uint32_t* const fill = tab + 2 * NP;   // [2][NP]
  uint32_t* const off = tab + 4 * NP;    // [NP]
  uint32_t* const delta = tab + 5 * NP;  // [NP]
  const int lane = lane_id();
  const int wave = (int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t producer = blockIdx.x;
  for (uint32_t p = threadIdx.x; p < NP; p += kTileBlock) {
    hist[p] = 0;
    hist[NP + p] = 0;
    fill[p] = (PT.flags & PTF_RESUME) ? PT.counts[(uint64_t)p * PT.n_producers + producer] : 0u;
    fill[NP + p] = 0;
  }
  __syncthreads();
  // this producer's regions: region p starts at prod_base + p * part_bytes, row r at + r * kRowBytes (layouts 0 and 1: regions are contiguous)
  constexpr uint32_t kRowBytes = WIDE ? 16u : 12u;
  uint8_t* const prod_base = (uint8_t*)(PT.rows + (uint64_t)producer * PT.prod_stride);
  const uint64_t part_bytes = PT.part_stride * 8ull;
  const int tag_shift = T.shift - 32;  // slot = image >> tag_shift (the image is the hash's high half)
  const int64_t n_groups = (n + 63) >> 6;
  const int64_t tile_groups = (int64_t)NWAVES * U;
  const int64_t n_tiles = (n_groups + tile_groups - 1) / tile_groups;
  uint32_t err = 0;
  uint64_t passed = 0;  // wave-uniform
  typename POL::PREP prep;
  POL::prepare(F, prep);
  // the current tile's rows between P1 and P3
  uint32_t r_img[U], r_rank[U];  // rank: 0xFFFFFFFF = no routed row in this lane
  uint64_t r_val[U], r_key[WIDE ? U : 1];
  COLV col[U], ncol[U];
  uint32_t cv[U], ncv[U];
  auto part_of_img = [&](uint32_t img) -> uint32_t { return ((img >> tag_shift) & (uint32_t)T.mask) >> PT.part_shift; };
  // P4 of the tile sorted last (total_rows of them lie in the sorted buffer): lane-contiguous copy-out
  auto copy_out = [&](uint32_t total_rows) {
    FOR_U {
      const uint32_t s = (uint32_t)u * kTileBlock + threadIdx.x;
      const bool have = s < total_rows;
      const uint32_t sc = have ? s : 0u;
      const uint64_t val = s_val[sc];
      uint32_t part, tag = 0;
      uint64_t kk = 0;
      if (WIDE) {
        kk = s_key[sc];
        part = s_part[sc];
      } else {
        tag = s_tag[sc];
        part = part_of_img(tag);
      }
      if (!have) part = 0;  // (a stale row of an earlier tile: any table index will do)
      const uint32_t row = s + delta[part];
      const bool fits = row < PT.cap_rows;
      if (have && fits) {
        uint8_t* const dst = prod_base + (uint64_t)part * part_bytes + (uint64_t)row * kRowBytes;
        if (WIDE) {
          *(ulonglong2*)dst = make_ulonglong2(kk, val);
        } else {
          uint32_t* const o32 = (uint32_t*)dst;
          o32[0] = tag;
          o32[1] = (uint32_t)val;
          o32[2] = (uint32_t)(val >> 32);
        }
      }
      if (__ballot(have && !fits) != 0) {  // region overflow (skewed keys): the general path takes the row -- as a key again
        uint64_t key[1] = {WIDE ? kk : (uint64_t)unhash_word32(tag)};
        uint64_t sv[kMaxAggs];
#pragma unroll
        for (int a = 0; a < kMaxAggs; ++a) sv[a] = a == 0 ? val : 0ull;
        spill_row<1>(T, spill, have && !fits, key, sv);
      }
    }
  };
  // The loop is rotated so that it holds ONE copy of P1 and the column hand-over sits BEFORE the previous tile's stores:
  //   hand-over(t): tile t's columns (issued a whole tile ago) become current, tile t + 1's loads are issued
  //   P4(t - 1) | P1(t) | B1 | P2(t) | P3(t) | B2
  // The wait of the next hand-over then covers loads issued a whole tile earlier and stores issued before P1 -- vmcnt counts
  // both, in order, and the compiler waits for everything after conditional code with memory operations (the rare spill paths).
  int64_t tile = blockIdx.x;
  {
    const int64_t w0 = tile * tile_groups + (int64_t)wave * U;
    load_trip<POL>(P, C, w0, tile < n_tiles && w0 < n_groups, n, lane, ncol, ncv);
  }
  int cur = 0;
  uint32_t total = 0;  // rows of the previous tile still to be copied out
  for (; tile < n_tiles; tile += gridDim.x) {
    FOR_U {
      col[u] = ncol[u];
      cv[u] = ncv[u];
    }
    {
      const int64_t next = tile + gridDim.x;
      const int64_t w1 = next * tile_groups + (int64_t)wave * U;
      load_trip<POL>(P, C, w1, next < n_tiles && w1 < n_groups, n, lane, ncol, ncv);
    }
    copy_out(total);  // P4(t - 1)
    uint32_t* const h = hist + (size_t)cur * NP;
    uint32_t* const f_cur = fill + (size_t)cur * NP;
    uint32_t* const f_nxt = fill + (size_t)(cur ^ 1) * NP;
    {  // P1(t)
      const int64_t w0 = tile * tile_groups + (int64_t)wave * U;
      FOR_U {
        const int64_t row = (w0 + u) * 64 + lane;
        const bool inb = row < n;
        u64x16 reg;
        uint32_t rv = 0;
        POL::eval(P, F, col[u], cv[u], reg, rv, inb, err, prep);
        bool pass = inb && POL::pass(P, F, plan.pred, col[u], cv[u], reg, rv, prep);
        uint64_t key[1];
        key[0] = POL::key(P, F, plan.key[0], 0, col[u], cv[u], reg, rv);
        uint64_t v;
        bool valid;
        POL::arg(P, F, plan.arg[0], 0, col[u], cv[u], reg, rv, v, valid);
        const uint64_t val0 = transform_value(POL::xform(T, 0), v, valid);
        passed += (uint64_t)__popcll(__ballot(pass));
        const uint64_t hk = hash_keys<1>(key);
        const uint32_t img = (uint32_t)(hk >> 32);
        // rows the routed form cannot carry go to the spill list and nowhere else (the replay -- launch_merge_rows -> table_apply --
        // knows the sentinel key's slot): narrow rows without a 32-bit image (the host then leaves narrow mode), the claim sentinel
        const bool slow = pass && (WIDE ? key[0] == kEmptyKey : ((key[0] >> 32) != 0 || img >= kTagForeign));
        if (__ballot(slow) != 0) {
          if (!WIDE && __hip_atomic_load(&T.ctrl[CTRL_WIDE_KEYS], RLX_AGENT) == 0u) __hip_atomic_store(&T.ctrl[CTRL_WIDE_KEYS], 1u, RLX_AGENT);
          uint64_t sv[kMaxAggs];
#pragma unroll
          for (int a = 0; a < kMaxAggs; ++a) sv[a] = a == 0 ? val0 : 0ull;
          spill_row<1>(T, spill, slow, key, sv);
          pass = pass && !slow;
        }
        r_img[u] = img;
        r_val[u] = val0;
        if (WIDE) r_key[u] = key[0];
        r_rank[u] = pass ? atomicAdd(&h[part_of_img(img)], 1u) : 0xFFFFFFFFu;
      }
    }
    lds_barrier();  // B1
    // P2: per-wave redundant scan of the histogram (lane l owns partitions [l * PPL, (l + 1) * PPL))
    {
      const uint32_t PPL = (NP + 63u) / 64u;
      const uint32_t p0 = (uint32_t)lane * PPL;
      uint32_t sum = 0;
      for (uint32_t q = 0; q < PPL; ++q)
        if (p0 + q < NP) sum += h[p0 + q];
      uint32_t inc = sum;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)inc, d, 64);
        if (lane >= d) inc += o;
      }
      total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
      uint32_t excl = inc - sum;
      for (uint32_t q = 0; q < PPL; ++q) {
        const uint32_t p = p0 + q;
        if (p < NP) {
          const uint32_t c = h[p];
          const uint32_t f = f_cur[p];
          off[p] = excl;
          delta[p] = f - excl;
          f_nxt[p] = f + c;
          excl += c;
        }
      }
      if (wave == 0) {
        uint32_t* const hn = hist + (size_t)(cur ^ 1) * NP;
        for (uint32_t q = 0; q < PPL; ++q)
          if (p0 + q < NP) hn[p0 + q] = 0;
      }
    }
    // P3: scatter into the sorted buffer
    FOR_U {
      if (r_rank[u] != 0xFFFFFFFFu) {
        const uint32_t part = part_of_img(r_img[u]);
        const uint32_t pos = off[part] + r_rank[u];
        s_val[pos] = r_val[u];
        if (WIDE) {
          s_key[pos] = r_key[u];
          s_part[pos] = (uint16_t)part;
        } else {
          s_tag[pos] = r_img[u];
        }
      }
    }
    lds_barrier();  // B2
    cur ^= 1;
  }
  copy_out(total);  // P4 of the last tile
  __syncthreads();
  // region counts; a region is padded to a whole 16-row chunk with rows pass 2 skips (the ring flavours resume at chunk boundaries)
  const uint32_t* const f_end = fill + (size_t)cur * NP;
  uint32_t max_fill = 0;
  for (uint32_t p = threadIdx.x; p < NP; p += kTileBlock) {
    uint32_t f = f_end[p];
    if (f > PT.cap_rows) f = PT.cap_rows;
    const uint32_t rem = f % 16u;
    if (rem != 0) {
      for (uint32_t r = f; r < f - rem + 16u; ++r) {  // (cap_rows is a multiple of 64: the padding never leaves the region)
        uint8_t* const dst = prod_base + (uint64_t)p * part_bytes + (uint64_t)r * kRowBytes;
        if (WIDE) {
          *(ulonglong2*)dst = make_ulonglong2(kEmptyKey, 0ull);
        } else {
          uint32_t* const o32 = (uint32_t*)dst;
          o32[0] = kTagEmpty;
          o32[1] = o32[2] = 0;
        }
      }
      f = f - rem + 16u;
    }
    PT.counts[(uint64_t)p * PT.n_producers + producer] = f;
    max_fill = f > max_fill ? f : max_fill;
  }
  publish_max_fill(T, max_fill);
  if (lane == 0) stat_add(T, STAT_PASSED, passed);
  if (err) atomicOr(&T.ctrl[CTRL_ERROR], err);
  snapshot_ctrl_if_last(T, PT);
}

template <typename POLT>
void launch_partition_tile(const DevProgram& P, const DevFastPlan& fast, const DevColumns& C, const DevAggPlan& plan, const DevTable& T,
                           const DevPartition& PT, const DevRows& spill, int64_t n, hipStream_t s) {
  const int grid = (int)PT.n_producers;  // every producer writes its counts, even with no rows
  if (PT.flags & PTF_NARROW)
    hipLaunchKernelGGL((k_partition_tile<POLT, 0>), dim3(grid), dim3(kTileBlock), partition_tile_bytes(PT.n_parts, POLT::U, false), s, P, fast, C, plan, T, PT, spill, n);
  else
    hipLaunchKernelGGL((k_partition_tile<POLT, 1>), dim3(grid), dim3(kTileBlock), partition_tile_bytes(PT.n_parts, POLT::U, true), s, P, fast, C, plan, T, PT, spill, n);
}

}  // namespace dfx
