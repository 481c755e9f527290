// dfx_k_partition_tile_inl.hpp -- pass 1 of the partitioned GROUP BY, TILE-BUCKETED flavour (PTF_TILE): dense scans.
//
// The ring kernels (dfx_k_partition_inl.hpp, dfx_k_partition_ws_inl.hpp) write-combine routed rows through per-partition LDS
// rings that any wave may append to at any time; the price is a protocol of dependent LDS round trips per routed batch of 64
// rows -- fill atomic -> generation word -> ring row -> commit atomic -> job list -> ring read -> store, ~16 LDS
// instructions, spin loops on busy chunk slots -- which a SELECTIVE scan hides behind its column loads (a fifth of the rows
// is routed) and a DENSE scan does not: config 3 (no predicate, every row routed) spent 414 us per 2^26-row launch where its
// traffic (1.07 GB read + 0.81 GB written) asks for 300.
//
// When most rows are routed there is no need for a protocol.  A workgroup takes a TILE of rows (16 waves x U row groups x 64
// = 8192 rows at U = 8) and works in two phases separated by workgroup barriers that do not wait for global memory
// (s_waitcnt lgkmcnt(0) + s_barrier: HIP's __syncthreads() would also wait for the NEXT tile's column loads, which are in
// flight across the whole tile):
//
//   scatter(t)   every row: predicate, key -> hash image -> partition p, operand; rank = hist[p]++ (ONE returning LDS atomic);
//                the row's place in its region is fill[p] + rank, known at once.  rank < CAP: the row is parked in partition
//                p's LDS bucket, slot rank (CAP ~ 1.5 x the expected rows per partition and tile: 48 at 8192 / 256);
//                rank >= CAP (one row in ~5000 on uniform keys): stored straight to the region from registers.
//   -- B1 --
//   hand-over    the columns of tile t + 1 (loaded one tile ago) become current, the loads of tile t + 2 are issued -- BEFORE
//                this tile's stores: the wait then covers loads only (vmcnt counts loads and stores, in order)
//   copy-out(t)  wave w owns partitions w, w + 16, ...: lane r < min(hist[p], CAP) reads bucket slot r and writes region row
//                fill[p] + r -- adjacent lanes, adjacent rows: one run of ~32 rows = 384 bytes per partition and tile, the
//                partition wave-uniform (a scalar base, no per-row partition arithmetic); then fill[p] += hist[p], hist[p] = 0
//   -- B2 --
// Per routed row: one LDS atomic, one LDS write and one LDS read of the row; no scan, no permutation, nothing carried in
// registers across a barrier.  (The first version of this file counting-SORTED the tile -- histogram, barrier, a 256-entry
// scan by every wave, scatter to the compact position, barrier, copy-out with a per-row partition look-up: correct, and at
// 452 us per 2^26-row launch slower than the ring kernel's 414: ~470 SIMD cycles of vector ALU per row group, of which the
// scan and the per-row bookkeeping were half; profiles/r05_call1_kprobe.txt.)
// Same regions, counts and padding as the ring flavours (a region is padded to a whole 16-row chunk at the end of the launch),
// so pass 2 -- and a later launch of another flavour that resumes the window -- cannot tell.  Rows the routed form cannot
// carry (narrow mode: a key >= 2^32 or a reserved image; the claim sentinel; region overflow) take the spill list as everywhere.
// One key word, one routed value; no hot-key pairs (skewed streams keep the ring kernel with PTF_HOT; here a hot key's rows
// beyond CAP per tile are single stores, correct but slow).
#pragma once
#include "dfx_k_partition_inl.hpp"

namespace dfx {

constexpr int kTileBlock = 1024;  // one workgroup per CU (tiles of 16 waves x U row groups); 512: two per CU, out of phase with each other

// LDS barrier that does not wait for global memory (see above)
DEV void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <typename POL, int WIDE, int BLOCK = kTileBlock>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_partition_tile(const DevProgram P, const DevFastPlan F, const DevColumns C,
                                                              const DevAggPlan plan, const DevTable T,
                                                              const DevPartition PT, const DevRows spill, const int64_t n) {
  typedef typename POL::COLV COLV;
  constexpr int U = POL::U;
  constexpr int NWAVES = BLOCK / 64;
  extern __shared__ __attribute__((aligned(16))) uint64_t lds[];
  const uint32_t NP = PT.n_parts;
  const uint32_t CAP = PT.stage_rows;  // bucket slots per partition
  // buckets, structure of arrays: slot (p, r) at p * CAP + r
  uint64_t* const b_val = lds;                                                   // [NP * CAP]
  uint64_t* const b_key = lds + (size_t)NP * CAP;                                // [NP * CAP]  (WIDE)
  uint32_t* const b_tag = (uint32_t*)(lds + (size_t)NP * CAP);                   // [NP * CAP]  (narrow: the hash image)
  uint32_t* const hist = WIDE ? (uint32_t*)(lds + 2 * (size_t)NP * CAP) : b_tag + (size_t)NP * CAP;  // [NP] rows of the tile per partition
  uint32_t* const fill = hist + NP;                                              // [NP] rows already in this producer's regions
  const int lane = lane_id();
  const int wave = (int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t producer = blockIdx.x;
  for (uint32_t p = threadIdx.x; p < NP; p += BLOCK) {
    hist[p] = 0;
    fill[p] = (PT.flags & PTF_RESUME) ? PT.counts[(uint64_t)p * PT.n_producers + producer] : 0u;
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
  POL::prepare(P, F, prep);
  COLV col[U], ncol[U];
  uint32_t cv[U], ncv[U];
  auto store_row = [&](uint32_t part, uint32_t row, uint32_t tag, uint64_t kk, uint64_t val) {
    uint8_t* const dst = prod_base + (uint64_t)part * part_bytes + (uint64_t)row * kRowBytes;
    if (WIDE) {
      *(ulonglong2*)dst = make_ulonglong2(kk, val);
    } else {
      uint32_t* const o32 = (uint32_t*)dst;
      o32[0] = tag;
      o32[1] = (uint32_t)val;
      o32[2] = (uint32_t)(val >> 32);
    }
  };
  auto spill_one = [&](bool todo, uint32_t tag, uint64_t kk, uint64_t val) {  // (wave-convergent)
    uint64_t key[1] = {WIDE ? kk : (uint64_t)unhash_word32(tag)};
    uint64_t sv[kMaxAggs];
#pragma unroll
    for (int a = 0; a < kMaxAggs; ++a) sv[a] = a == 0 ? val : 0ull;
    spill_row<1>(T, spill, todo, key, sv);
  };
  int64_t tile = blockIdx.x;
  {
    const int64_t w0 = tile * tile_groups + (int64_t)wave * U;
    load_trip<POL>(P, C, w0, tile < n_tiles && w0 < n_groups, n, lane, col, cv);
    const int64_t next = tile + gridDim.x;
    const int64_t w1 = next * tile_groups + (int64_t)wave * U;
    load_trip<POL>(P, C, w1, next < n_tiles && w1 < n_groups, n, lane, ncol, ncv);
  }
  for (; tile < n_tiles; tile += gridDim.x) {
    // scatter(t).  Straight-line over the U row groups of the wave -- hashes, then the U rank atomics back to back, then the U
    // bucket writes -- so that the groups' LDS round trips overlap; the rare rows (no routed form, bucket full) are collected
    // in lane bit masks and handled behind ONE wave-uniform test each, outside the common path.
    {
      const int64_t w0 = tile * tile_groups + (int64_t)wave * U;
      const bool full = (w0 + U) * 64 <= n;  // (wave-uniform: no lane of this wave's trip lies past the end)
      uint32_t s_img[U], s_part[U], s_rank[U];
      uint64_t s_val[U], s_keyw[WIDE ? U : 1];
      uint32_t passbits = 0, slowbits = 0;
      FOR_U {
        const int64_t row = (w0 + u) * 64 + lane;
        const bool inb = full || row < n;
        u64x16 reg;
        uint32_t rv = 0;
        POL::eval(P, F, col[u], cv[u], reg, rv, inb, err, prep);
        const bool pass = inb && POL::pass(P, F, plan.pred, col[u], cv[u], reg, rv, prep);
        uint64_t key[1];
        key[0] = POL::key(P, F, plan.key[0], 0, col[u], cv[u], reg, rv);
        uint64_t v;
        bool valid;
        POL::arg(P, F, plan.arg[0], 0, col[u], cv[u], reg, rv, v, valid);
        s_val[u] = transform_value(POL::xform(T, 0), v, valid);
        const uint32_t img = (uint32_t)(hash_keys<1>(key) >> 32);
        s_img[u] = img;
        if (WIDE) s_keyw[u] = key[0];
        s_part[u] = ((img >> tag_shift) & (uint32_t)T.mask) >> PT.part_shift;
        // rows the routed form cannot carry: narrow rows without a 32-bit image (the host then leaves narrow mode), the claim sentinel
        const bool slow = pass && (WIDE ? key[0] == kEmptyKey : ((key[0] >> 32) != 0 || img >= kTagForeign));
        passbits |= (pass && !slow ? 1u : 0u) << u;
        slowbits |= (slow ? 1u : 0u) << u;
      }
      if (__ballot(slowbits != 0) != 0) {
        // they go to the spill list and nowhere else (the replay -- launch_merge_rows -> table_apply -- knows the sentinel key's slot).
        // Narrow rows: the key is re-read from the column (the image does not stand for it)
        if (!WIDE && __hip_atomic_load(&T.ctrl[CTRL_WIDE_KEYS], RLX_AGENT) == 0u) __hip_atomic_store(&T.ctrl[CTRL_WIDE_KEYS], 1u, RLX_AGENT);
        FOR_U {
          const bool slow = (slowbits >> u) & 1u;
          if (__ballot(slow) != 0) {
            u64x16 reg;
            uint32_t rv = 0;
            uint32_t e2 = 0;
            POL::eval(P, F, col[u], cv[u], reg, rv, true, e2, prep);
            uint64_t key[1];
            key[0] = POL::key(P, F, plan.key[0], 0, col[u], cv[u], reg, rv);
            uint64_t sv[kMaxAggs];
#pragma unroll
            for (int a = 0; a < kMaxAggs; ++a) sv[a] = a == 0 ? s_val[u] : 0ull;
            spill_row<1>(T, spill, slow, key, sv);
          }
        }
      }
      FOR_U passed += (uint64_t)__popcll(__ballot(((passbits | slowbits) >> u) & 1u));
      FOR_U {
        s_rank[u] = 0xFFFFFFFFu;
        if ((passbits >> u) & 1u) s_rank[u] = atomicAdd(&hist[s_part[u]], 1u);
      }
      uint32_t directbits = 0;
      FOR_U {
        const bool pass = (passbits >> u) & 1u;
        const bool staged = pass && s_rank[u] < CAP;
        if (staged) {
          const uint32_t at = s_part[u] * CAP + s_rank[u];
          b_val[at] = s_val[u];
          if (WIDE) b_key[at] = s_keyw[u];
          else b_tag[at] = s_img[u];
        }
        directbits |= (pass && !staged ? 1u : 0u) << u;
      }
      if (__ballot(directbits != 0) != 0) {  // a full bucket: the row's place is known all the same (fill[] is stable until B1)
        FOR_U {
          const bool direct = (directbits >> u) & 1u;
          if (__ballot(direct) != 0) {
            uint32_t rrow = 0;
            if (direct) rrow = fill[s_part[u]] + s_rank[u];
            const bool fits = direct && rrow < PT.cap_rows;
            if (fits) store_row(s_part[u], rrow, s_img[u], WIDE ? s_keyw[u] : 0ull, s_val[u]);
            if (__ballot(direct && !fits) != 0) spill_one(direct && !fits, s_img[u], WIDE ? s_keyw[u] : 0ull, s_val[u]);
          }
        }
      }
    }
    lds_barrier();  // B1: every row of tile t is in its bucket (or in its region)
    // hand-over: tile t + 1's columns become current, tile t + 2's loads are issued
    {
      const int64_t next = tile + gridDim.x;
      FOR_U {
        col[u] = ncol[u];
        cv[u] = ncv[u];
      }
      const int64_t w2 = (next + gridDim.x) * tile_groups + (int64_t)wave * U;
      load_trip<POL>(P, C, w2, next + gridDim.x < n_tiles && w2 < n_groups, n, lane, ncol, ncv);
    }
    // copy-out(t): wave w owns partitions w, w + 16, ...  Everything about a partition is wave-uniform (scalar registers):
    // its row count, its region's next free row, the store's base address; a lane adds its own 12 (16) bytes.
    if (CAP <= 32u) {
      // small buckets (two workgroups per CU, tiles of 4096 rows): a wave's visit serves TWO partitions, one per half-wave
      const uint32_t half = (uint32_t)lane >> 5, r = (uint32_t)lane & 31u;
      for (uint32_t p0 = (uint32_t)wave * 2u; p0 < NP; p0 += 2u * (uint32_t)NWAVES) {
        const uint32_t p = p0 + half;
        const bool live = p < NP;
        const uint32_t pc = live ? p : 0u;
        const uint32_t cnt = live ? hist[pc] : 0u;
        const uint32_t f = fill[pc];
        const uint32_t m = cnt < CAP ? cnt : CAP;
        const bool have = r < m;
        const uint32_t at = pc * CAP + (have ? r : 0u);
        const uint64_t val = b_val[at];
        uint64_t kk = 0;
        uint32_t tag = 0;
        if (WIDE) kk = b_key[at];
        else tag = b_tag[at];
        const uint32_t rrow = f + r;
        const bool fits = have && rrow < PT.cap_rows;
        if (fits) store_row(pc, rrow, tag, kk, val);
        if (__ballot(have && !fits) != 0) spill_one(have && !fits, tag, kk, val);  // region overflow (skewed keys)
        if (live && r == 0) {
          fill[pc] = f + cnt;
          hist[pc] = 0;
        }
      }
    } else {
      // ONE LDS round trip gives the wave the counts and fills of all its partitions (lane i <-> partition wave + NWAVES * i; a
      // visit reads them back with v_readlane), and the bucket reads of four visits are in flight together: counters showed the
      // first version -- two dependent LDS round trips per visit -- spending 43 % of its wave cycles in s_waitcnt and 35 % at
      // the barriers behind them (profiles/r05_tile_counters.txt).
      const uint32_t lane_bytes = (uint32_t)lane * kRowBytes;
      const uint32_t myp = (uint32_t)wave + (uint32_t)NWAVES * (uint32_t)lane;
      uint32_t v_cnt = 0, v_fill = 0;
      if (myp < NP) {
        v_cnt = hist[myp];
        v_fill = fill[myp];
        fill[myp] = v_fill + v_cnt;
        hist[myp] = 0;
      }
      const uint32_t n_visits = (NP - (uint32_t)wave + (uint32_t)NWAVES - 1u) / (uint32_t)NWAVES;
      constexpr int VB = 4;  // visits per batch
      for (uint32_t i0 = 0; i0 < n_visits; i0 += VB) {
        uint32_t cnt[VB], f[VB], m[VB], pp[VB];
        uint64_t bv[VB], bk[VB];
        uint32_t bt[VB];
        bool okv[VB];
#pragma unroll
        for (int j = 0; j < VB; ++j) {
          const uint32_t i = i0 + (uint32_t)j;
          const uint32_t ic = i < n_visits ? i : 0u;
          cnt[j] = (uint32_t)__builtin_amdgcn_readlane((int)v_cnt, (int)ic);
          f[j] = (uint32_t)__builtin_amdgcn_readlane((int)v_fill, (int)ic);
          if (i >= n_visits) cnt[j] = 0;
          m[j] = cnt[j] < CAP ? cnt[j] : CAP;
          pp[j] = (uint32_t)wave + (uint32_t)NWAVES * ic;
          okv[j] = (uint32_t)lane < m[j];  // (CAP <= 64 on this path: one step per visit)
          const uint32_t at = pp[j] * CAP + (okv[j] ? (uint32_t)lane : 0u);
          bv[j] = b_val[at];
          if (WIDE) bk[j] = b_key[at];
          else bt[j] = b_tag[at];
        }
#pragma unroll
        for (int j = 0; j < VB; ++j) {
          uint8_t* const region = prod_base + (uint64_t)pp[j] * part_bytes;  // (scalar)
          if (f[j] + m[j] <= PT.cap_rows) {  // the common case: no row of this bucket can leave the region
            if (okv[j]) {
              uint8_t* const dst = region + (uint64_t)f[j] * kRowBytes + (size_t)lane_bytes;
              if (WIDE) {
                *(ulonglong2*)dst = make_ulonglong2(bk[j], bv[j]);
              } else {
                uint32_t* const o32 = (uint32_t*)dst;
                o32[0] = bt[j];
                o32[1] = (uint32_t)bv[j];
                o32[2] = (uint32_t)(bv[j] >> 32);
              }
            }
          } else {  // region overflow (skewed keys): row by row, the spill list takes what does not fit
            const uint32_t rrow = f[j] + (uint32_t)lane;
            const bool fits = okv[j] && rrow < PT.cap_rows;
            if (fits) store_row(pp[j], rrow, WIDE ? 0u : bt[j], WIDE ? bk[j] : 0ull, bv[j]);
            if (__ballot(okv[j] && !fits) != 0) spill_one(okv[j] && !fits, WIDE ? 0u : bt[j], WIDE ? bk[j] : 0ull, bv[j]);
          }
        }
      }
    }
    lds_barrier();  // B2: the buckets are free, fill[] is current
  }
  __syncthreads();
  // region counts; a region is padded to a whole 16-row chunk with rows pass 2 skips (the ring flavours resume at chunk boundaries)
  uint32_t max_fill = 0;
  for (uint32_t p = threadIdx.x; p < NP; p += BLOCK) {
    uint32_t f = fill[p];
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

// POLT: the policy of the narrow flavour (12-byte rows: eight row groups per wave and tile for the signatures), POLTW: of the wide
// one (16-byte rows: four -- its buckets are a third larger per row)
template <typename POLT, typename POLTW>
void launch_partition_tile(const DevProgram& P, const DevFastPlan& fast, const DevColumns& C, const DevAggPlan& plan, const DevTable& T,
                           const DevPartition& PT, const DevRows& spill, int64_t n, hipStream_t s) {
  const int grid = (int)PT.n_producers;  // every producer writes its counts, even with no rows
  const bool narrow = (PT.flags & PTF_NARROW) != 0;
  const size_t lds_bytes = partition_tile_bytes(PT.n_parts, PT.stage_rows, !narrow);
  if (PT.block == 512) {  // two workgroups per CU (agg.tile_block = 512): half the tile, half the bucket, the phases of the two interleave
    if (narrow) hipLaunchKernelGGL((k_partition_tile<POLT, 0, 512>), dim3(grid), dim3(512), lds_bytes, s, P, fast, C, plan, T, PT, spill, n);
    else hipLaunchKernelGGL((k_partition_tile<POLTW, 1, 512>), dim3(grid), dim3(512), lds_bytes, s, P, fast, C, plan, T, PT, spill, n);
  } else {
    if (narrow) hipLaunchKernelGGL((k_partition_tile<POLT, 0, 1024>), dim3(grid), dim3(1024), lds_bytes, s, P, fast, C, plan, T, PT, spill, n);
    else hipLaunchKernelGGL((k_partition_tile<POLTW, 1, 1024>), dim3(grid), dim3(1024), lds_bytes, s, P, fast, C, plan, T, PT, spill, n);
  }
}

}  // namespace dfx
