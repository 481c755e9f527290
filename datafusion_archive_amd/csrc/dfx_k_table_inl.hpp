// dfx_k_table_inl.hpp -- the group-table kernels, templated on the number of key words KW.
// Instantiated once per KW in dfx_k_table{1,2,3,4}.hip so the four variants build in parallel.
#pragma once
#include "dfx_kernels_inl.hpp"
#include "dfx_launch.hpp"

namespace dfx {
// ---------------------------------------------------------------------------------------------
// K6/K7 hash_agg with an LDS front cache
// ---------------------------------------------------------------------------------------------
// Dynamic LDS layout (all 8-byte words, base 16-byte aligned): keys[KW][S], accs[na][S].
// For KW > 1 an extra state[S] (uint32) follows.  S = plan.lds_slots (power of two), split into
// plan.lds_copies lane-replicated sub-tables so that few-group inputs (TPC-H Q1: <= 6 groups) do not
// serialise 64 lanes on one LDS address.
template <int KW, typename POL>
__global__ __launch_bounds__(kBlock) void k_hash_agg(const DevProgram P, const DevFastPlan F, const DevColumns C,
                                                     const DevAggPlan plan, const DevTable T,
                                                     const DevRows spill, const int64_t n) {
  typedef typename POL::COLV COLV;
  constexpr int U = POL::U;
  extern __shared__ __attribute__((aligned(16))) uint64_t lds[];
  const int S = plan.lds_slots;
  uint64_t* lkeys = lds;
  uint64_t* laccs = lds + (size_t)KW * S;
  uint32_t* lstate = (uint32_t*)(lds + (size_t)(KW + POL::na(T)) * S);
  const int lane = lane_id();
  if (S > 0) {
    for (int i = threadIdx.x; i < S; i += kBlock) {
      lkeys[i] = kEmptyKey;
      if (KW > 1) lstate[i] = 0u;
      for (int a = 0; a < POL::na(T); ++a) laccs[a * S + i] = T.acc_init[a];
    }
    __syncthreads();
  }
  const int sub_slots = S > 0 ? S / plan.lds_copies : 0;
  const int sub_base = S > 0 ? (lane & (plan.lds_copies - 1)) * sub_slots : 0;

  const int64_t n_words = (n + 63) >> 6;
  const int64_t wave_global = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * kBlock) >> 6;
  uint32_t err = 0;
  uint32_t lds_hit = 0, lds_miss = 0;
  uint64_t passed = 0;
  int iter = 0;
  bool saturated = false;
  typename POL::PREP prep;  // (PlanPolicy: the plan words in vector registers; empty otherwise)
  POL::prepare(P, F, prep);
  for (int64_t w0 = wave_global * U; w0 < n_words; w0 += n_waves * U, ++iter) {
    if ((iter & 7) == 0) {  // wave-uniform, one request: has the table passed its load limit?
      saturated = __hip_atomic_load(&T.ctrl[CTRL_SATURATED], RLX_AGENT) != 0u;
      if (!saturated && (uint64_t)__hip_atomic_load(&T.ctrl[CTRL_OCCUPIED], RLX_AGENT) > T.load_limit) {
        saturated = true;
        if (lane == 0) __hip_atomic_store(&T.ctrl[CTRL_SATURATED], 1u, RLX_AGENT);
      }
    }
    COLV col[U];
    uint32_t cv[U];
    FOR_U {
      const int64_t row = (w0 + u) * 64 + lane;
      POL::load(P, C, row, row < n, col[u], cv[u]);
    }
    // ONE copy of the evaluation + table code: a run-time loop over the U prefetched row-groups
#pragma nounroll
    for (int uu = 0; uu < U; ++uu) {
      COLV cur;
      uint32_t curv;
      DFX_SELECT_BANK(uu, col, cv, cur, curv)
      const int64_t row = (w0 + uu) * 64 + lane;
      const bool inb = row < n;
      u64x16 reg;
      uint32_t rv = 0;
      POL::eval(P, F, cur, curv, reg, rv, inb, err, prep);
      const bool pass = inb && POL::pass(P, F, plan.pred, cur, curv, reg, rv, prep);
      uint64_t key[KW];
      uint64_t val[kMaxAggs];
#pragma unroll
      for (int k = 0; k < KW; ++k)  // key nulls are not checked (aggregate.rs:807-852)
        key[k] = POL::key(P, F, plan.key[k], k, cur, curv, reg, rv);
#pragma unroll
      for (int a = 0; a < kMaxAggs; ++a) {
        val[a] = 0;
        if (a < POL::na(T)) {
          uint64_t v;
          bool valid;  // value(row) read blindly (aggregate.rs:561-603)
          POL::arg(P, F, plan.arg[a], a, cur, curv, reg, rv, v, valid);
          val[a] = transform_value(POL::xform(T, a), v, valid);
        }
      }
      passed += pass ? 1 : 0;
      bool todo = pass;
      // ---- LDS front cache ----
      if (S > 0 && todo && !(KW == 1 && key[0] == kEmptyKey)) {
        const uint64_t h = hash_keys<KW>(key);
        // the 32 hash bits live in the HIGH half of h; the table slot takes its top bits, the cache its low ones
        int slot = sub_base + (int)((h >> 32) & (uint64_t)(sub_slots - 1));
        int found = -1;
        if (KW == 1) {
          for (int p = 0; p < 4 && found < 0; ++p) {
            const uint64_t k = lkeys[slot];
            if (k == key[0]) {
              found = slot;
              ++lds_hit;  // a REUSED cache entry: the only case in which the cache saves a global atomic
            } else if (k == kEmptyKey) {
              const uint64_t old = atomicCAS((unsigned long long*)&lkeys[slot], (unsigned long long)kEmptyKey,
                                             (unsigned long long)key[0]);
              if (old == kEmptyKey || old == key[0]) found = slot;
              if (old == key[0]) ++lds_hit;
            }
            if (found < 0) slot = sub_base + ((slot - sub_base + 1) & (sub_slots - 1));
          }
        } else {
          int spins = 0;
          for (int p = 0; p < 4 && found < 0;) {
            uint32_t st = __hip_atomic_load(&lstate[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (st == 0u) {
              const uint32_t old = atomicCAS(&lstate[slot], 0u, 1u);
              if (old == 0u) {
  #pragma unroll
                for (int k = 0; k < KW; ++k) lkeys[k * S + slot] = key[k];
                __threadfence_block();
                __hip_atomic_store(&lstate[slot], 2u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                found = slot;
                break;
              }
              st = old;
            }
            if (st == 1u) {
              if (++spins > 4096) break;
              continue;
            }
            __threadfence_block();
            bool same = true;
  #pragma unroll
            for (int k = 0; k < KW; ++k) same = same && (((volatile uint64_t*)lkeys)[k * S + slot] == key[k]);
            if (same) {
              found = slot;
              ++lds_hit;
            }
            else slot = sub_base + ((slot - sub_base + 1) & (sub_slots - 1));
            ++p;
          }
        }
        if (found >= 0) {
  #pragma unroll
          for (int a = 0; a < kMaxAggs; ++a)
            if (a < POL::na(T)) acc_atomic(POL::acc_kind(T, a), &laccs[a * S + found], val[a]);
          todo = false;
        }
        ++lds_miss;  // rows that went through the cache (hit / rows = reuse rate)
      }
      // ---- global table ----
      if (todo && !saturated) {
        if (table_apply<KW>(T, key, val)) todo = false;
      }
      // ---- spill (table saturated or probe sequence exhausted) ----
      spill_row<KW>(T, spill, todo, key, val);
    }
  }
  // flush the LDS cache: every occupied slot becomes one merge into the global table
  if (S > 0) {
    __syncthreads();
    // ONE poll per thread: an agent-scope load of the same word from every lane of every slot
    // iteration serialises on one L2 channel (measured: 2 M loads ~ 1 ms)
    const bool sat = __hip_atomic_load(&T.ctrl[CTRL_SATURATED], RLX_AGENT) != 0u;
    for (int i = threadIdx.x; i < S; i += kBlock) {
      bool occ;
      uint64_t key[KW];
      if (KW == 1) {
        key[0] = lkeys[i];
        occ = key[0] != kEmptyKey;
      } else {
        occ = lstate[i] == 2u;
#pragma unroll
        for (int k = 0; k < KW; ++k) key[k] = lkeys[k * S + i];
      }
      uint64_t val[kMaxAggs];
#pragma unroll
      for (int a = 0; a < kMaxAggs; ++a) val[a] = (a < POL::na(T)) ? laccs[a * S + i] : 0;
      bool todo = occ;
      if (todo && !sat) {
        if (table_apply<KW>(T, key, val)) todo = false;
      }
      spill_row<KW>(T, spill, todo, key, val);
    }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    passed += shfl_xor_u64(passed, m);
    lds_hit += __shfl_xor(lds_hit, m, 64);
    lds_miss += __shfl_xor(lds_miss, m, 64);
  }
  if (lane == 0) {
    stat_add(T, STAT_PASSED, passed);
    stat_add(T, STAT_LDS_HIT, lds_hit);
    stat_add(T, STAT_LDS_MISS, lds_miss);
  }
  if (err) atomicOr(&T.ctrl[CTRL_ERROR], err);
}

// pre-evaluated rows -> table (spill replay, rehash, partial import).  The source is `rows` planes
// of capacity rows.capacity; rows [row_begin, row_begin + n_rows).
template <int KW>
__global__ __launch_bounds__(kBlock) void k_merge_rows(const DevRows rows, const int64_t row_begin,
                                                       const int64_t n_rows, const DevTable T,
                                                       const DevRows spill) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  const int64_t n_pad = (n_rows + 63) & ~63ll;  // whole waves stay in the loop for the ballots
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n_pad; i += stride) {
    const bool inb = i < n_rows;
    uint64_t key[KW];
    uint64_t val[kMaxAggs];
#pragma unroll
    for (int k = 0; k < KW; ++k) key[k] = inb ? rows.words[(uint64_t)k * rows.capacity + row_begin + i] : 0;
#pragma unroll
    for (int a = 0; a < kMaxAggs; ++a)
      val[a] = (inb && a < T.na) ? rows.words[(uint64_t)(KW + a) * rows.capacity + row_begin + i] : 0;
    bool todo = inb;
    if (todo && table_apply<KW>(T, key, val)) todo = false;
    spill_row<KW>(T, spill, todo, key, val);
  }
}

template <int KW>
DEV bool slot_occupied(const DevTable& T, uint64_t slot) {
  if (slot == T.mask + 1) return KW == 1 && T.ctrl[CTRL_SENTINEL] != 0u;
  if (KW == 1) return T.keys[slot] != kEmptyKey;
  return T.state[slot] == 2u;
}

template <int KW>
__global__ __launch_bounds__(kBlock) void k_rehash(const DevTable from, const DevTable to, const DevRows spill) {
  const int64_t n_slots = (int64_t)from.mask + 2;  // + the sentinel slot
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  const int64_t n_pad = (n_slots + 63) & ~63ll;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n_pad; i += stride) {
    const bool occ = i < n_slots && slot_occupied<KW>(from, (uint64_t)i);
    uint64_t key[KW];
    uint64_t val[kMaxAggs];
#pragma unroll
    for (int k = 0; k < KW; ++k) key[k] = occ ? from.keys[(uint64_t)k * from.stride + i] : 0;
    if (KW == 1 && occ && (uint64_t)i == from.mask + 1) key[0] = kEmptyKey;
#pragma unroll
    for (int a = 0; a < kMaxAggs; ++a) val[a] = (occ && a < from.na) ? from.accs[(uint64_t)a * from.stride + i] : 0;
    bool todo = occ;
    if (todo && table_apply<KW>(to, key, val)) todo = false;
    spill_row<KW>(to, spill, todo, key, val);
  }
}

// ---------------------------------------------------------------------------------------------
// K8 emit_groups
// ---------------------------------------------------------------------------------------------
template <int KW>
__global__ __launch_bounds__(kBlock) void k_table_mask(const DevTable T, uint64_t* __restrict__ mask_words,
                                                       uint32_t* __restrict__ tile_counts) {
  __shared__ uint32_t wave_cnt[kBlock / 64];
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int64_t n = (int64_t)T.mask + 2;
  const int64_t n_words = (n + 63) >> 6;
  const int64_t n_tiles = (n + kTileRows - 1) / kTileRows;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    uint32_t cnt = 0;
    for (int i = 0; i < 16; ++i) {
      const int64_t w = tile * 64 + wave * 16 + i;
      const int64_t slot = w * 64 + lane;
      const bool occ = slot < n && slot_occupied<KW>(T, (uint64_t)slot);
      const uint64_t word = __ballot(occ);
      if (lane == 0 && w < n_words) mask_words[w] = word;
      cnt += (uint32_t)__popcll(word);
    }
    if (lane == 0) wave_cnt[wave] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) tile_counts[tile] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// multi-GPU partial export
// ---------------------------------------------------------------------------------------------
template <int KW>
DEV uint64_t slot_key_hash(const DevTable& T, uint64_t slot, uint64_t (&key)[KW]) {
#pragma unroll
  for (int k = 0; k < KW; ++k) key[k] = T.keys[(uint64_t)k * T.stride + slot];
  if (KW == 1 && slot == T.mask + 1) key[0] = kEmptyKey;
  return hash_keys<KW>(key);
}

// Both kernels aggregate per WORKGROUP in LDS and touch the `world` global counters once per workgroup (tile):
// agent-scope atomics on one address serialise at ~11 ns each on MI355X, so one atomic per group would cost
// ~11 ms per million groups -- more than the whole scan of 1e9 rows.
constexpr int kMaxWorld = 1024;
constexpr int kPartialItems = 8;  // table slots per thread and tile

template <int KW>
__global__ __launch_bounds__(kBlock) void k_partial_count(const DevTable T, int world, uint64_t* counts) {
  __shared__ uint32_t hist[kMaxWorld];
  for (int r = threadIdx.x; r < world; r += kBlock) hist[r] = 0;
  __syncthreads();
  const int64_t n = (int64_t)T.mask + 2;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    if (slot_occupied<KW>(T, (uint64_t)i)) {
      uint64_t key[KW];
      const uint64_t h = slot_key_hash<KW>(T, (uint64_t)i, key);
      atomicAdd(&hist[hash_rank(h, (uint32_t)world)], 1u);
    }
  }
  __syncthreads();
  for (int r = threadIdx.x; r < world; r += kBlock)
    if (hist[r]) atomicAdd((unsigned long long*)&counts[r], (unsigned long long)hist[r]);
}

template <int KW>
__global__ __launch_bounds__(kBlock) void k_partial_scatter(const DevTable T, int world,
                                                            const uint64_t* __restrict__ bucket_base,
                                                            const uint64_t* __restrict__ bucket_count,
                                                            uint64_t* cursors, uint64_t* __restrict__ dst) {
  __shared__ uint32_t hist[kMaxWorld];   // groups of this tile per destination rank
  __shared__ uint64_t tbase[kMaxWorld];  // where this tile's groups start inside each bucket
  const int64_t n = (int64_t)T.mask + 2;
  const int nw = KW + T.na;
  const int64_t tile_slots = (int64_t)kBlock * kPartialItems;
  for (int64_t t0 = (int64_t)blockIdx.x * tile_slots; t0 < n; t0 += (int64_t)gridDim.x * tile_slots) {
    for (int r = threadIdx.x; r < world; r += kBlock) hist[r] = 0;
    __syncthreads();
    uint32_t rank[kPartialItems], lpos[kPartialItems];
#pragma unroll
    for (int it = 0; it < kPartialItems; ++it) {
      const int64_t i = t0 + (int64_t)it * kBlock + threadIdx.x;
      rank[it] = 0xFFFFFFFFu;
      lpos[it] = 0;
      if (i < n && slot_occupied<KW>(T, (uint64_t)i)) {
        uint64_t key[KW];
        const uint64_t h = slot_key_hash<KW>(T, (uint64_t)i, key);
        rank[it] = hash_rank(h, (uint32_t)world);
        lpos[it] = atomicAdd(&hist[rank[it]], 1u);
      }
    }
    __syncthreads();
    for (int r = threadIdx.x; r < world; r += kBlock)
      tbase[r] = hist[r] ? atomicAdd((unsigned long long*)&cursors[r], (unsigned long long)hist[r]) : 0ull;
    __syncthreads();
#pragma unroll
    for (int it = 0; it < kPartialItems; ++it) {
      if (rank[it] != 0xFFFFFFFFu) {
        const int64_t i = t0 + (int64_t)it * kBlock + threadIdx.x;
        const uint32_t r = rank[it];
        const uint64_t g = tbase[r] + lpos[it];
        uint64_t* b = dst + (uint64_t)nw * bucket_base[r];
        const uint64_t cnt = bucket_count[r];
        uint64_t key[KW];
        (void)slot_key_hash<KW>(T, (uint64_t)i, key);
#pragma unroll
        for (int k = 0; k < KW; ++k) b[(uint64_t)k * cnt + g] = key[k];
        for (int a = 0; a < T.na; ++a) b[(uint64_t)(KW + a) * cnt + g] = T.accs[(uint64_t)a * T.stride + i];
      }
    }
    __syncthreads();
  }
}


// ---------------------------------------------------------------------------------------------
// host launchers (one instantiation per KW)
// ---------------------------------------------------------------------------------------------
template <int KW>
hipError_t table_hash_agg(const DevProgram& P, const DevFastPlan& fast, const DevColumns& C, const DevAggPlan& plan,
                          const DevTable& T, const DevRows& spill, int64_t n, hipStream_t s) {
  const int64_t n_blocks = (n + kBlock - 1) / kBlock;
  const size_t lds_bytes = plan.lds_slots > 0
                               ? (size_t)plan.lds_slots * ((size_t)(KW + T.na) * 8 + (KW > 1 ? 4 : 0))
                               : 0;
  // LDS-heavy blocks: fewer, longer-lived workgroups amortise the cache init + flush
  const int per_cu = lds_bytes > 0 ? (lds_bytes > 40 * 1024 ? 2 : 4) : 8;
  const int grid = stream_grid(n_blocks, per_cu);
#define DFX_HA(POL) hipLaunchKernelGGL((k_hash_agg<KW, POL>), dim3(grid), dim3(kBlock), lds_bytes, s, P, fast, C, plan, T, spill, n)
  if (KW == 1 && sig_matches<SigKeySumPred2F64>(P, fast, KW, T.na, T.acc_kind, T.val_xform)) {
    DFX_HA(DFX_ARG(StaticPolicy<2, 4, SigKeySumPred2F64>));
    return hipGetLastError();
  }
  if (KW == 1 && sig_matches<SigKeySum>(P, fast, KW, T.na, T.acc_kind, T.val_xform)) {
    DFX_HA(DFX_ARG(StaticPolicy<2, 4, SigKeySum>));
    return hipGetLastError();
  }
  if (KW == 2 && sig_matches<SigQ1>(P, fast, KW, T.na, T.acc_kind, T.val_xform)) {
    DFX_HA(DFX_ARG(StaticPolicy<8, 2, SigQ1>));
    return hipGetLastError();
  }
  // A scan plan where the decoded shapes would not run (validity bitmaps) or would run their slow loader (4-byte columns):
  // the same shape family with nulls by arrow's rules and widening loads.  (8-byte null-free scans keep FastPolicy here: it
  // also takes product arguments; the partitioned strategy is where the rows are, and it runs plans for everything.)
  if (P.has_nulls || !P.wide8 || (fast.plan_mode & 3) == 2) {
    DevFastPlan fp;
    DevColumns cp;
    if (bind_scan_plan(P, fast, C, KW, T.na, T.val_xform, false, &fp, &cp)) {
      if (fp.scan.n_cols <= 2) hipLaunchKernelGGL((k_hash_agg<KW, PlanPolicyN<2, 4, kPlanW4 | kPlanNulls>>), dim3(grid), dim3(kBlock), lds_bytes, s, P, fp, cp, plan, T, spill, n);
      else hipLaunchKernelGGL((k_hash_agg<KW, PlanPolicyN<4, 2, kPlanW4 | kPlanNulls>>), dim3(grid), dim3(kBlock), lds_bytes, s, P, fp, cp, plan, T, spill, n);
      return hipGetLastError();
    }
  }
  if (fast.plan_mode & 4) return hipErrorNotSupported;  // (the host fused a predicate over nulls counting on a plan)
  const bool use_fast = fast.valid && !P.has_nulls;
  if (P.n_cols <= 2) { if (use_fast) DFX_HA(DFX_ARG(FastPolicy<2, 4>)); else DFX_HA(DFX_ARG(InterpPolicy<2, 4>)); }
  else if (P.n_cols <= 4) { if (use_fast) DFX_HA(DFX_ARG(FastPolicy<4, 4>)); else DFX_HA(DFX_ARG(InterpPolicy<4, 4>)); }
  else { if (use_fast) DFX_HA(DFX_ARG(FastPolicy<8, 2>)); else DFX_HA(DFX_ARG(InterpPolicy<8, 2>)); }
#undef DFX_HA
  return hipGetLastError();
}

template <int KW>
hipError_t table_merge_rows(const DevRows& rows, int64_t row_begin, int64_t n_rows, const DevTable& T,
                            const DevRows& spill, hipStream_t s) {
  const int grid = stream_grid((n_rows + kBlock - 1) / kBlock, 8);
  hipLaunchKernelGGL(k_merge_rows<KW>, dim3(grid), dim3(kBlock), 0, s, rows, row_begin, n_rows, T, spill);
  return hipGetLastError();
}

template <int KW>
hipError_t table_rehash(const DevTable& from, const DevTable& to, const DevRows& spill, hipStream_t s) {
  const int64_t n = (int64_t)from.mask + 2;
  const int grid = stream_grid((n + kBlock - 1) / kBlock, 8);
  hipLaunchKernelGGL(k_rehash<KW>, dim3(grid), dim3(kBlock), 0, s, from, to, spill);
  return hipGetLastError();
}

template <int KW>
hipError_t table_mask(const DevTable& T, uint64_t* mask_words, uint32_t* tile_counts, hipStream_t s) {
  const int64_t n = (int64_t)T.mask + 2;
  const int64_t tiles = (n + kTileRows - 1) / kTileRows;
  const int grid = stream_grid(tiles, 8);
  hipLaunchKernelGGL(k_table_mask<KW>, dim3(grid), dim3(kBlock), 0, s, T, mask_words, tile_counts);
  return hipGetLastError();
}

template <int KW>
hipError_t table_partial_count(const DevTable& T, int world, uint64_t* counts, hipStream_t s) {
  const int64_t n = (int64_t)T.mask + 2;
  if (world > kMaxWorld) return hipErrorInvalidValue;
  const int grid = stream_grid((n + kBlock - 1) / kBlock, 8);
  hipLaunchKernelGGL(k_partial_count<KW>, dim3(grid), dim3(kBlock), 0, s, T, world, counts);
  return hipGetLastError();
}

template <int KW>
hipError_t table_partial_scatter(const DevTable& T, int world, const uint64_t* bucket_base,
                                 const uint64_t* bucket_count, uint64_t* cursors, uint64_t* dst, hipStream_t s) {
  const int64_t n = (int64_t)T.mask + 2;
  if (world > kMaxWorld) return hipErrorInvalidValue;
  const int grid = stream_grid((n + (int64_t)kBlock * kPartialItems - 1) / ((int64_t)kBlock * kPartialItems), 8);
  hipLaunchKernelGGL(k_partial_scatter<KW>, dim3(grid), dim3(kBlock), 0, s, T, world, bucket_base, bucket_count, cursors, dst);
  return hipGetLastError();
}

}  // namespace dfx
