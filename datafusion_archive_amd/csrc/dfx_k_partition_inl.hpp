// dfx_k_partition_inl.hpp -- the partitioned GROUP BY strategy for high-cardinality single-key
// aggregates (BASELINE config 3: 1 M Int64 keys).
//
// Why: global atomics on MI355X top out near 24 G updates/s whatever the table size or scope
// (profiles/r01_ubench_mi355x.jsonl), i.e. ~42 ms for 1e9 rows, while LDS atomics run above 1 T/s.
// With uniformly distributed keys a per-workgroup LDS cache never sees a key twice, so rows are
// first ROUTED to the workgroup that owns their slice of the table:
//
//   pass 1  k_partition      every workgroup ("producer") scans its row tiles (predicate + key +
//           argument expressions, same row-source policies as K7), stages the passing rows in LDS
//           and appends them to per-(producer, partition) private regions of a scratch buffer.
//           No global atomics: a region has exactly one writer.  partition = table block index.
//   pass 2  k_partition_agg  one workgroup per partition copies its table block (keys + accumulator
//           planes, 64 KB) into LDS, folds the partition's rows in with LDS CAS / LDS atomics
//           (ds_cmpst_rtn_b64, ds_add_f64, ...), and writes the block back.
//
// Rows that do not fit (a region overflows: heavy skew; a block is full) go to the ordinary spill
// list and are merged by the global-atomic path, so the strategy is correct for any distribution.
//
// Translation units: the pass-1 kernels below are templates over the row-source policy; every (policy, write-combining
// policy) pair is instantiated in its own file (dfx_k_partition_v0.hip ... _v7.hip, one DFX_PARTITION_VARIANT line each)
// so that they compile in parallel; dfx_k_partition.hip holds the dispatcher (launch_partition), pass 2 and the sizing
// helpers.
#pragma once
#include "dfx_kernels_inl.hpp"
#include "dfx_launch.hpp"

namespace dfx {

constexpr int kPBlock = 1024;  // pass-1 workgroup (one per CU): 16 waves share one set of fill counters
constexpr int kABlock = 1024;  // pass-2 workgroup (one per CU): 16 waves share a 128 KB LDS copy of a table block

// address of row `row` of the region that producer `producer` fills for partition `part`
DEV uint64_t* region_row(const DevPartition& PT, uint32_t part, uint32_t producer, uint32_t row) {
  return PT.rows + (uint64_t)part * PT.part_stride + (uint64_t)producer * PT.prod_stride + (uint64_t)(row >> 6) * PT.win_stride +
         (uint64_t)(row & 63u) * PT.n_words;
}

// 12-byte rows (PTF_NARROW): the strides stay in 8-byte words (cap_rows is a multiple of 64), rows are 3 dwords
DEV uint32_t* region_row12(const DevPartition& PT, uint32_t part, uint32_t producer, uint32_t row) {
  return (uint32_t*)(PT.rows + (uint64_t)part * PT.part_stride + (uint64_t)producer * PT.prod_stride + (uint64_t)(row >> 6) * PT.win_stride) +
         (uint64_t)(row & 63u) * 3u;
}

// Narrow rows of the large-chunk geometry (PTF_CHUNK16; dfx_device.hpp: kNarrowLine): with LINE chunks a chunk of CH = 10 rows
// is one 128-byte line, in the region and in the LDS ring alike.  Regions of this geometry are contiguous (layouts 0 and 1).
template <int CH, int NARROW, int DW = 3>
constexpr bool ring_is_line() {
  return kNarrowLine && NARROW != 0 && ((DW == 3 && CH == kNarrowChunkRows) || (DW == kPairRowDwords && CH == kPairChunkRows));
}
// dword index (from the ring's first dword) of row r of chunk slot sl of partition `part`  (DW: dwords per row -- 3, or 5: PTF_PAIR)
template <int CH, int RP, int NARROW, int DW = 3>
DEV uint32_t ring_dword12(uint32_t part, uint32_t sl, uint32_t r) {
  if constexpr (ring_is_line<CH, NARROW, DW>()) return (part * (uint32_t)(RP / CH) + sl) * 32u + r * (uint32_t)DW;
  else return (part * (uint32_t)RP + sl * (uint32_t)CH + r) * 3u;
}
// dwords of a narrow ring
template <int CH, int RP, int NARROW, int DW = 3>
DEV size_t ring_dwords12(uint32_t n_parts) {
  if constexpr (ring_is_line<CH, NARROW, DW>()) return (size_t)n_parts * (size_t)(RP / CH) * 32u;
  else return (size_t)n_parts * RP * 3u;
}
// row slot `row` of a narrow region
template <int CH, int NARROW, int DW = 3>
DEV uint32_t* region_row12g(const DevPartition& PT, uint32_t part, uint32_t producer, uint32_t row) {
  if constexpr (ring_is_line<CH, NARROW, DW>()) {
    const uint32_t c = row / (uint32_t)CH;
    return (uint32_t*)(PT.rows + (uint64_t)part * PT.part_stride + (uint64_t)producer * PT.prod_stride) + (uint64_t)c * 32u + (uint64_t)(row - c * (uint32_t)CH) * (uint32_t)DW;
  } else {
    return region_row12(PT, part, producer, row);
  }
}

DEV uint32_t partition_of(const DevTable& T, const DevPartition& PT, uint64_t h) {
  return (uint32_t)(((h >> T.shift) & T.mask) >> PT.part_shift);
}

// largest region fill of this launch -> T.ctrl[CTRL_MAX_FILL] (the host decides from it when pass 2 has to run).  One
// candidate per workgroup, and only if it beats what is already there: same-address agent atomics serialise at
// ~11 ns each, a thousand of them at the end of every launch would be 5 % of pass 1.
DEV void publish_max_fill(const DevTable& T, uint32_t max_fill) {
  __shared__ uint32_t wg_max_fill;
  if (threadIdx.x == 0) wg_max_fill = 0;
  __syncthreads();
#pragma unroll
  for (int mm = 32; mm >= 1; mm >>= 1) {
    const uint32_t o = (uint32_t)__shfl_xor((int)max_fill, mm, 64);
    max_fill = o > max_fill ? o : max_fill;
  }
  if (lane_id() == 0 && max_fill != 0) atomicMax(&wg_max_fill, max_fill);
  __syncthreads();
  if (threadIdx.x == 0 && wg_max_fill > __hip_atomic_load(&T.ctrl[CTRL_MAX_FILL], RLX_AGENT)) atomicMax(&T.ctrl[CTRL_MAX_FILL], wg_max_fill);
}

// DevPartition::snap_host: the last workgroup of the launch publishes the control block to the host.  Every update of
// T.ctrl is an agent-scope atomic issued before the workgroup's increment of snap_done (release), so the workgroup that
// sees the full count (acquire) reads final values.  Called by every thread at the very end of the kernel.
DEV void snapshot_ctrl_if_last(const DevTable& T, const DevPartition& PT) {
  if (PT.snap_host == nullptr) return;
  __syncthreads();  // this workgroup's ctrl updates have been issued
  if (threadIdx.x >= 64) return;  // wave 0 only from here
  uint32_t done = 0;
  if (threadIdx.x == 0) done = __hip_atomic_fetch_add(PT.snap_done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
  done = (uint32_t)__builtin_amdgcn_readfirstlane((int)done);
  if (done != gridDim.x - 1) return;
  // the last workgroup: lanes 0..15 copy one control word each -- ONE 64-byte write over PCIe (sixteen stores by one lane
  // cost ~20 us at the tail of every launch: measured as +25 us of pass 1 when it carried the snapshot)
  if (threadIdx.x == 0) __hip_atomic_store(PT.snap_done, 0u, RLX_AGENT);  // ready for the next launch (stream order)
  static_assert(CTRL_WORDS == 16, "one lane per control word");
  if (threadIdx.x < CTRL_WORDS)
    __hip_atomic_store(&PT.snap_host[threadIdx.x], __hip_atomic_load(&T.ctrl[threadIdx.x], RLX_AGENT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// pass 1.  No staging: a passing row is routed straight from registers.  Its position inside the
// (producer, partition) region comes from an LDS atomic on the workgroup's per-partition fill
// counter; the U row-groups of a trip issue their LDS atomics back to back, then their 16-byte row
// stores.  The regions have exactly one writing workgroup, so there is no global atomic and no
// barrier in the loop, and the only LDS is the counter array (occupancy is register-bound).
template <typename POL>
__global__ __launch_bounds__(kPBlock) void k_partition(const DevProgram P, const DevFastPlan F, const DevColumns C,
                                                      const DevAggPlan plan, const DevTable T,
                                                      const DevPartition PT, const DevRows spill, const int64_t n) {
  typedef typename POL::COLV COLV;
  constexpr int U = POL::U;
  extern __shared__ __attribute__((aligned(16))) uint64_t lds[];
  uint32_t* fill = (uint32_t*)lds;  // [n_parts] rows appended to each of this producer's regions
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int NW = (int)PT.n_words;
  const uint32_t producer = blockIdx.x;
  for (uint32_t p = threadIdx.x; p < PT.n_parts; p += kPBlock)
    fill[p] = (PT.flags & PTF_RESUME) ? PT.counts[(uint64_t)p * PT.n_producers + producer] : 0u;
  __syncthreads();
  const int64_t n_groups = (n + 63) >> 6;
  const int64_t wave_global = (int64_t)blockIdx.x * (kPBlock / 64) + wave;
  const int64_t n_waves = (int64_t)gridDim.x * (kPBlock / 64);
  uint32_t err = 0;
  uint64_t passed = 0;
  typename POL::PREP prep;  // (PlanPolicy: the plan words in vector registers; empty otherwise)
  POL::prepare(P, F, prep);
  for (int64_t w0 = wave_global * U; w0 < n_groups; w0 += n_waves * U) {
    COLV col[U];
    uint32_t cv[U];
    FOR_U {
      const int64_t row = (w0 + u) * 64 + lane;
      POL::load(P, C, row, row < n, col[u], cv[u]);
    }
    uint64_t key[U][1];
    uint64_t val[U][kMaxAggs];
    uint32_t part[U], pos[U];
    uint32_t passbits = 0;
    FOR_U {
      const int64_t row = (w0 + u) * 64 + lane;
      const bool inb = row < n;
      u64x16 reg;
      uint32_t rv = 0;
      POL::eval(P, F, col[u], cv[u], reg, rv, inb, err, prep);
      bool pass = inb && POL::pass(P, F, plan.pred, col[u], cv[u], reg, rv, prep);
      key[u][0] = POL::key(P, F, plan.key[0], 0, col[u], cv[u], reg, rv);
#pragma unroll
      for (int a = 0; a < kMaxAggs; ++a) {
        val[u][a] = 0;
        if (a < POL::na(T)) {
          uint64_t v;
          bool valid;
          POL::arg(P, F, plan.arg[a], a, col[u], cv[u], reg, rv, v, valid);
          val[u][a] = transform_value(POL::xform(T, a), v, valid);
        }
      }
      passed += pass ? 1 : 0;
      if (pass && key[u][0] == kEmptyKey) {  // the claim-sentinel key lives outside the blocks
        const bool ok = table_apply<1>(T, key[u], val[u]);
        (void)ok;
        pass = false;
      }
      part[u] = partition_of(T, PT, hash_keys<1>(key[u]));
      passbits |= (pass ? 1u : 0u) << u;
    }
    FOR_U pos[u] = ((passbits >> u) & 1u) ? atomicAdd(&fill[part[u]], 1u) : 0xFFFFFFFFu;  // LDS atomics in flight together
    FOR_U {
      const bool pass = (passbits >> u) & 1u;
      bool todo = pass;
      if (pass && pos[u] < PT.cap_rows) {
        uint64_t* dst = region_row(PT, part[u], producer, pos[u]);
        if (POL::na(T) == 1) {  // 16-byte row: one store
          *(ulonglong2*)dst = make_ulonglong2(key[u][0], val[u][0]);
        } else {
          dst[0] = key[u][0];
#pragma unroll
          for (int a = 0; a < kMaxAggs; ++a)
            if (a < POL::na(T)) dst[1 + a] = val[u][a];
        }
        todo = false;
      }
      spill_row<1>(T, spill, todo, key[u], val[u]);  // region overflow (skewed keys): the general path takes it
    }
  }
  __syncthreads();
  uint32_t max_fill = 0;
  for (uint32_t p = threadIdx.x; p < PT.n_parts; p += kPBlock) {
    const uint32_t f = fill[p] < PT.cap_rows ? fill[p] : PT.cap_rows;
    PT.counts[(uint64_t)p * PT.n_producers + producer] = f;
    max_fill = f > max_fill ? f : max_fill;
  }
  publish_max_fill(T, max_fill);
#pragma unroll
  for (int mm = 32; mm >= 1; mm >>= 1) passed += shfl_xor_u64(passed, mm);
  if (lane == 0) stat_add(T, STAT_PASSED, passed);
  if (err) atomicOr(&T.ctrl[CTRL_ERROR], err);
  snapshot_ctrl_if_last(T, PT);
}


// ---- pass 1, write-combining variant --------------------------------------------------------------
// Measured on MI355X (tools/ubench2.hip, profiles/r01_ubench_mi355x.jsonl): scattered 16-byte stores
// run at ~87 G/s chip-wide however L2-resident their lines are (transaction-bound), while runs of
// >= 64 contiguous bytes written by adjacent lanes run at > 400 G rows/s.  So the workgroup first
// COLLECTS passing rows in an LDS buffer (one wave-aggregated LDS atomic per trip), and when the
// buffer is full the whole workgroup counting-sorts it by partition (LDS histogram -> scan ->
// permutation) and copies it out in sorted order: adjacent lanes then write adjacent rows of the
// same region.  The hash is computed once per PASSING row at full lane utilisation, not once per
// scanned row.  A flush round is six barriers; all waves take part in every round (a wave that has
// finished its input keeps joining rounds until every wave has finished).
constexpr int kSortMaxCap = 8192;  // LDS buffer rows (upper bound: 13-bit row index in the permutation word)

DEV uint32_t mbcnt64(uint64_t m) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

struct SortLds {
  uint64_t* buf;    // [n_words][cap] word-major row buffer
  uint32_t* perm;   // [cap] sorted position -> (partition << 16 | buffer row)
  uint32_t* hist;   // [n_parts] rows of this round per partition; after the scan: exclusive offsets
  uint32_t* fill;   // [n_parts] rows already written to this producer's region
  uint32_t* delta;  // [n_parts] region row = sorted position + delta (mod 2^32)
  uint32_t* misc;   // [0] claimed rows, [1] finished waves, [2..] wave totals of the scan
  uint32_t cap;     // rows of the buffer (plane stride)
  uint32_t limit;   // rows accepted before the next flush (<= cap)
};

template <int BLOCK>
DEV bool partition_flush(const DevTable& T, const DevPartition& PT, const DevRows& spill, const SortLds& L,
                         uint32_t producer, int na) {
  constexpr int NWAVES = BLOCK / 64;
  constexpr int ITEMS = kSortMaxCap / BLOCK;
  const uint32_t tid = threadIdx.x;
  const int lane = lane_id();
  const uint32_t items = L.cap / BLOCK;
  __syncthreads();  // B0: every append of this round is in the buffer
  uint32_t R = L.misc[0];
  if (R > L.limit) R = L.limit;
  const bool last = L.misc[1] == (uint32_t)NWAVES;
  uint32_t packed[ITEMS];
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    packed[it] = 0xFFFFFFFFu;
    const uint32_t i = tid + (uint32_t)it * BLOCK;
    if ((uint32_t)it < items && i < R) {
      uint64_t key[1] = {L.buf[i]};
      const uint32_t part = partition_of(T, PT, hash_keys<1>(key));
      const uint32_t rank = atomicAdd(&L.hist[part], 1u);
      packed[it] = (part << 16) | rank;
    }
  }
  __syncthreads();  // B1: histogram complete
  // exclusive scan over partitions; PP consecutive partitions per thread
  const uint32_t NPT = PT.n_parts;
  const uint32_t PP = (NPT + BLOCK - 1) / BLOCK;
  const uint32_t p0 = tid * PP;
  uint32_t sum = 0;
  for (uint32_t q = 0; q < PP; ++q)
    if (p0 + q < NPT) sum += L.hist[p0 + q];
  uint32_t inc = sum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(inc, d, 64);
    if (lane >= d) inc += o;
  }
  if (lane == 63) L.misc[2 + (tid >> 6)] = inc;
  __syncthreads();  // B2
  uint32_t excl = inc - sum;
#pragma unroll
  for (int w = 0; w < NWAVES; ++w)
    if (w < (int)(tid >> 6)) excl += L.misc[2 + w];
  for (uint32_t q = 0; q < PP; ++q) {
    const uint32_t p = p0 + q;
    if (p < NPT) {
      const uint32_t h = L.hist[p];
      const uint32_t f = L.fill[p];
      L.hist[p] = excl;
      L.delta[p] = f - excl;
      L.fill[p] = f + h;
      excl += h;
    }
  }
  __syncthreads();  // B3: offsets ready
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    if (packed[it] != 0xFFFFFFFFu) {
      const uint32_t part = packed[it] >> 16, rank = packed[it] & 0xFFFFu;
      L.perm[L.hist[part] + rank] = (part << 16) | (tid + (uint32_t)it * BLOCK);
    }
  }
  __syncthreads();  // B4: permutation complete
  for (uint32_t p = tid; p < NPT; p += BLOCK) L.hist[p] = 0;
  if (tid == 0) L.misc[0] = 0;
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    if ((uint32_t)it < items && (uint32_t)it * BLOCK < R) {  // wave-uniform
      const uint32_t j = tid + (uint32_t)it * BLOCK;
      const bool inb = j < R;
      uint64_t key[1] = {0};
      uint64_t val[kMaxAggs];
#pragma unroll
      for (int a = 0; a < kMaxAggs; ++a) val[a] = 0;
      bool todo = false;
      if (inb) {
        const uint32_t e = L.perm[j];
        const uint32_t part = e >> 16, i = e & 0xFFFFu;
        const uint32_t row = j + L.delta[part];
        key[0] = L.buf[i];
#pragma unroll
        for (int a = 0; a < kMaxAggs; ++a)
          if (a < na) val[a] = L.buf[(size_t)(1 + a) * L.cap + i];
        if (row < PT.cap_rows) {
          uint64_t* dst = region_row(PT, part, producer, row);
          if (na == 1) {
            *(ulonglong2*)dst = make_ulonglong2(key[0], val[0]);
          } else {
            dst[0] = key[0];
#pragma unroll
            for (int a = 0; a < kMaxAggs; ++a)
              if (a < na) dst[1 + a] = val[a];
          }
        } else {
          todo = true;  // region overflow (skewed keys): the general path takes the row
        }
      }
      spill_row<1>(T, spill, todo, key, val);
    }
  }
  __syncthreads();  // B5: the buffer may be overwritten
  return last;
}

template <typename POL, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_partition_sorted(const DevProgram P, const DevFastPlan F, const DevColumns C,
                                                           const DevAggPlan plan, const DevTable T,
                                                           const DevPartition PT, const DevRows spill, const int64_t n) {
  typedef typename POL::COLV COLV;
  constexpr int U = POL::U;
  constexpr int NWAVES = BLOCK / 64;
  extern __shared__ __attribute__((aligned(16))) uint64_t lds[];
  SortLds L;
  L.cap = PT.stage_rows;
  // co-resident workgroups must not flush in lockstep (a flush leaves the CU's memory pipe idle unless the
  // other workgroup is scanning): odd producers take a short first round
  L.limit = (blockIdx.x & 1u) ? L.cap / 2 : L.cap;
  L.buf = lds;
  L.perm = (uint32_t*)(lds + (size_t)PT.n_words * L.cap);
  L.hist = L.perm + L.cap;
  L.fill = L.hist + PT.n_parts;
  L.delta = L.fill + PT.n_parts;
  L.misc = L.delta + PT.n_parts;
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int na = POL::na(T);
  const uint32_t producer = blockIdx.x;
  for (uint32_t p = threadIdx.x; p < PT.n_parts; p += BLOCK) {
    L.hist[p] = 0;
    L.fill[p] = (PT.flags & PTF_RESUME) ? PT.counts[(uint64_t)p * PT.n_producers + producer] : 0u;
  }
  if (threadIdx.x < 2 + NWAVES) L.misc[threadIdx.x] = 0;
  __syncthreads();
  const int64_t n_groups = (n + 63) >> 6;
  const int64_t wave_global = (int64_t)blockIdx.x * NWAVES + wave;
  const int64_t n_waves = (int64_t)gridDim.x * NWAVES;
  uint32_t err = 0;
  uint64_t passed = 0;
  typename POL::PREP prep;  // (PlanPolicy: the plan words in vector registers; empty otherwise)
  POL::prepare(P, F, prep);
  for (int64_t w0 = wave_global * U; w0 < n_groups; w0 += n_waves * U) {
    COLV col[U];
    uint32_t cv[U];
    FOR_U {
      const int64_t row = (w0 + u) * 64 + lane;
      POL::load(P, C, row, row < n, col[u], cv[u]);
    }
    uint64_t key[U][1];
    uint64_t val[U][kMaxAggs];
    uint32_t pend = 0;
    FOR_U {
      const int64_t row = (w0 + u) * 64 + lane;
      const bool inb = row < n;
      u64x16 reg;
      uint32_t rv = 0;
      POL::eval(P, F, col[u], cv[u], reg, rv, inb, err, prep);
      bool pass = inb && POL::pass(P, F, plan.pred, col[u], cv[u], reg, rv, prep);
      key[u][0] = POL::key(P, F, plan.key[0], 0, col[u], cv[u], reg, rv);
#pragma unroll
      for (int a = 0; a < kMaxAggs; ++a) {
        val[u][a] = 0;
        if (a < POL::na(T)) {
          uint64_t v;
          bool valid;
          POL::arg(P, F, plan.arg[a], a, col[u], cv[u], reg, rv, v, valid);
          val[u][a] = transform_value(POL::xform(T, a), v, valid);
        }
      }
      passed += pass ? 1 : 0;
      if (pass && key[u][0] == kEmptyKey) {  // the claim-sentinel key lives outside the blocks
        const bool ok = table_apply<1>(T, key[u], val[u]);
        (void)ok;
        pass = false;
      }
      pend |= (pass ? 1u : 0u) << u;
    }
    while (true) {
      uint64_t m[U];
      uint32_t tot = 0;
      FOR_U {
        m[u] = __ballot((pend >> u) & 1u);
        tot += (uint32_t)__popcll(m[u]);
      }
      if (tot == 0) break;
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(&L.misc[0], tot);
      base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
      uint32_t b = base;
      FOR_U {
        const uint32_t pos = b + mbcnt64(m[u]);
        if (((pend >> u) & 1u) && pos < L.limit) {
          L.buf[pos] = key[u][0];
#pragma unroll
          for (int a = 0; a < kMaxAggs; ++a)
            if (a < POL::na(T)) L.buf[(size_t)(1 + a) * L.cap + pos] = val[u][a];
          pend &= ~(1u << u);
        }
        b += (uint32_t)__popcll(m[u]);
      }
      if (base + tot <= L.limit) break;
      partition_flush<BLOCK>(T, PT, spill, L, producer, na);
      L.limit = L.cap;
    }
  }
  if (lane == 0) atomicAdd(&L.misc[1], 1u);
  while (!partition_flush<BLOCK>(T, PT, spill, L, producer, na)) {
  }
  uint32_t max_fill = 0;
  for (uint32_t p = threadIdx.x; p < PT.n_parts; p += BLOCK) {
    const uint32_t f = L.fill[p] < PT.cap_rows ? L.fill[p] : PT.cap_rows;
    PT.counts[(uint64_t)p * PT.n_producers + producer] = f;
    max_fill = f > max_fill ? f : max_fill;
  }
  publish_max_fill(T, max_fill);
#pragma unroll
  for (int mm = 32; mm >= 1; mm >>= 1) passed += shfl_xor_u64(passed, mm);
  if (lane == 0) stat_add(T, STAT_PASSED, passed);
  if (err) atomicOr(&T.ctrl[CTRL_ERROR], err);
  snapshot_ctrl_if_last(T, PT);
}

// ---- pass 1, lock-free write-combining variant ("ring") ----------------------------------------------
// The counting-sort variant above pays for its barriers: while a workgroup sorts, its CU issues no
// loads.  Here no wave ever waits for another one in the common case:
//   * each wave compacts its passing rows into a wave-private LDS queue (ballot + mbcnt); whenever 64
//     rows are queued it pops them and routes them at full lane utilisation;
//   * routing a row: hash -> partition; pos = LDS atomic on the workgroup's fill[partition] = the row's
//     final index in this producer's region; the row is parked in the partition's LDS ring
//     (kRingNCH chunks of kRingCH rows); the lane that parks the last row of a chunk (per-chunk commit
//     counter) owns the flush;
//   * flush jobs of a wave are executed cooperatively: kRingCH adjacent lanes store one chunk, i.e. one
//     64-byte run (>= 64-byte runs reach the coalesced store rate, tools/ubench2.hip);
//   * a chunk slot is reused only after its previous generation was flushed (per-slot generation word).
//     A lane whose slot is still busy keeps its row, helps with this wave's flush jobs and retries; the
//     lowest incomplete chunk of a partition never depends on anything, so every retry loop terminates.
// Leftover partial chunks are written row by row at the end.
// ring rows per partition kRingRP = rows per chunk (CH: 4 or 8) x chunks (NCH).  (An 8-row ring with two
// workgroups per CU -- 32 waves -- was measured ~20 % slower than 16 rows and one workgroup.)
constexpr int ring_queue_rows(int rp) { return rp == 16 ? 192 : 128; }  // (32- and 30-row rings: 96 KB of ring, 128-row queues)  // (32-row rings: 16-row chunks of 12-byte rows need the LDS)
constexpr int kRingBlock = 1024;
constexpr int kHotSlots = 1024;        // hot-key pairs per pass-1 workgroup, 16-byte rows
constexpr int kHotSlotsNarrow = 2048;  // ... with 12-byte rows (the ring is 16 KB smaller)
#ifndef DFX_RING_DEPTH
#define DFX_RING_DEPTH 1
#endif

struct RingLds {
  uint64_t* ring;    // [n_parts][kRingRP][n_words]
  uint64_t* queue;   // [waves][n_words][kRingQ]
  uint32_t* jobs;    // [waves][64] partition << 20 | chunk
  uint32_t* fill;    // [n_parts]
  uint32_t* commit;  // [n_parts][kRingNCH]
  uint32_t* gen;     // [n_parts][kRingNCH]
};

#ifdef DFX_PARTITION_MAIN_TU  // a plain function: defined once, in dfx_k_partition.hip
size_t partition_ring_bytes(uint32_t n_words, uint32_t n_parts, int kRingRP, bool hot, bool narrow, int queue_rows) {
  const int kRingQ = queue_rows > 0 ? queue_rows : ring_queue_rows(kRingRP);
  const size_t ring = (narrow && kNarrowLine && kRingRP == kNarrowRingRows) ? (size_t)n_parts * kNarrowRingSlots * kNarrowSlotBytes
                                                                            : (size_t)n_parts * kRingRP * (narrow ? 12 : n_words * 8);
  return ring + (size_t)(kRingBlock / 64) * kRingQ * n_words * 8 +
         (size_t)(kRingBlock / 64) * 64 * (kRingRP >= 16 ? 8 : 4) + (size_t)n_parts * 4 * (1 + 2 * 4) + 64 +
         (hot ? (size_t)(narrow ? kHotSlotsNarrow : kHotSlots) * 16 : 0);
}
#endif

#define WG_SCOPE __HIP_MEMORY_SCOPE_WORKGROUP

// route up to 64 rows (one per lane with have == true)
// NARROW (PTF_NARROW): 12-byte routed rows {32-bit hash image, 64-bit operand} instead of {key, operand}.  For a key
// below 2^32 the group hash (hash_word) is a BIJECTION of the key's low word -- three odd multiplies and three
// xor-shifts -- so the image identifies the key and pass 2 turns it back (unhash_word32) when it claims a slot.  The
// partition is implied by the region, the slot by the image's top bits: pass 2 neither re-hashes nor compares 64-bit
// keys.  A row whose key is not narrow (or whose image is one of the two reserved tags) takes the spill list.
// PTF_SHARED: the routed value is the aggregates' common RAW operand; a row that leaves the routed path (spill list,
// sentinel key) needs every aggregate's own accumulator operand again
DEV void expand_shared_operand(const DevTable& T, const uint64_t raw, uint64_t (&out)[kMaxAggs]) {
#pragma unroll
  for (int a = 0; a < kMaxAggs; ++a) out[a] = a < T.na ? transform_value(T.val_xform[a], raw, true) : 0ull;
}

template <int NV, int kRingCH, int kRingRP, int NARROW = 0>
DEV void ring_route(const DevTable& T, const DevPartition& PT, const DevRows& spill, const RingLds& L, uint32_t producer,
                    int na, bool have, const uint64_t (&key)[1], const uint64_t (&val)[kMaxAggs], uint64_t h, uint32_t& err) {
  constexpr int kRingNCH = kRingRP / kRingCH;
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int NW = (int)PT.n_words;
  uint32_t part = 0, pos = 0;
  bool pending = false, todo = false;
  const uint32_t img = (uint32_t)(h >> 32);
  if (have) {
    if (NARROW && ((key[0] >> 32) != 0 || img >= kTagForeign)) {
      todo = true;  // not representable as an image: the general path takes the row, and the host leaves narrow mode
      if (__hip_atomic_load(&T.ctrl[CTRL_WIDE_KEYS], RLX_AGENT) == 0u) __hip_atomic_store(&T.ctrl[CTRL_WIDE_KEYS], 1u, RLX_AGENT);
    } else {
      part = partition_of(T, PT, h);
      pos = atomicAdd(&L.fill[part], 1u);
      pending = pos < PT.cap_rows;
      todo = !pending;  // region overflow (skewed keys): the general path takes the row
    }
  }
  if (NARROW && (PT.flags & PTF_SHARED)) {
    if (__ballot(todo) != 0) {
      uint64_t sv[kMaxAggs];
      expand_shared_operand(T, val[0], sv);
      spill_row<1>(T, spill, todo, key, sv);
    }
  } else {
    spill_row<1>(T, spill, todo, key, val);
  }
  const uint32_t c = pos / kRingCH, sl = c % kRingNCH, g = c / kRingNCH, r = pos % kRingCH;
  const uint32_t cs = part * kRingNCH + sl;
  uint32_t* jobs = L.jobs + wave * 64;
  uint32_t spins = 0;
  while (true) {
    bool job = false;
    if (pending && __hip_atomic_load(&L.gen[cs], __ATOMIC_ACQUIRE, WG_SCOPE) == g) {
      uint64_t* dst = L.ring + ((size_t)part * kRingRP + sl * kRingCH + r) * NW;
      if (NARROW) {
        uint32_t* d32 = (uint32_t*)L.ring + ring_dword12<kRingCH, kRingRP, NARROW>(part, sl, r);
        d32[0] = img;
        d32[1] = (uint32_t)val[0];
        d32[2] = (uint32_t)(val[0] >> 32);
      } else if (NV == 1) {
        *(ulonglong2*)dst = make_ulonglong2(key[0], val[0]);
      } else {
        dst[0] = key[0];
#pragma unroll
        for (int a = 0; a < NV; ++a)
          if (a < na) dst[1 + a] = val[a];
      }
      const uint32_t old = __hip_atomic_fetch_add(&L.commit[cs], 1u, __ATOMIC_ACQ_REL, WG_SCOPE);
      job = old == (uint32_t)kRingCH - 1u;
      pending = false;
    }
    const uint64_t jm = __ballot(job);
    if (jm != 0) {
      const uint32_t njobs = (uint32_t)__popcll(jm);
      if (job) jobs[mbcnt64(jm)] = (part << 20) | c;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      // lanes that copy one chunk out: a 128-byte LINE chunk is eight 16-byte pieces (ten rows and the padding, as they lie in the
      // ring slot), any other chunk one lane per row
      constexpr bool LINE = ring_is_line<kRingCH, NARROW>();
      constexpr uint32_t kJobLanes = LINE ? 8u : (uint32_t)kRingCH;
      for (uint32_t j0 = 0; j0 < njobs; j0 += 64 / kJobLanes) {
        const uint32_t j = j0 + (uint32_t)lane / kJobLanes;
        if (j < njobs) {
          const uint32_t jw = jobs[j];
          const uint2 jb = make_uint2(jw >> 20, jw & 0xFFFFFu);
          const uint32_t rr = (uint32_t)lane % kJobLanes;
          if constexpr (LINE) {
            const uint4 piece = *(const uint4*)((const uint32_t*)L.ring + ring_dword12<kRingCH, kRingRP, NARROW>(jb.x, jb.y % kRingNCH, 0) + rr * 4u);
            *(uint4*)(region_row12g<kRingCH, NARROW>(PT, jb.x, producer, jb.y * kRingCH) + rr * 4u) = piece;
            continue;
          }
          const uint64_t* src = L.ring + ((size_t)jb.x * kRingRP + (jb.y % kRingNCH) * kRingCH + rr) * NW;
          uint64_t* out = region_row(PT, jb.x, producer, jb.y * kRingCH + rr);
          if (NARROW) {
            const uint32_t* s32 = (const uint32_t*)L.ring + ((size_t)jb.x * kRingRP + (jb.y % kRingNCH) * kRingCH + rr) * 3;
            uint32_t* o32 = region_row12(PT, jb.x, producer, jb.y * kRingCH + rr);
            const uint32_t a = s32[0], b = s32[1], c3 = s32[2];
            o32[0] = a;
            o32[1] = b;
            o32[2] = c3;
          } else if (NV == 1) {
            *(ulonglong2*)out = *(const ulonglong2*)src;
          } else {
            for (int w = 0; w < NW; ++w) out[w] = src[w];
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      if (job) {  // the slot belongs to the next generation
        __hip_atomic_store(&L.commit[cs], 0u, __ATOMIC_RELAXED, WG_SCOPE);
        __hip_atomic_fetch_add(&L.gen[cs], 1u, __ATOMIC_RELEASE, WG_SCOPE);
      }
    }
    if (__ballot(pending) == 0) break;
    if (++spins > (1u << 22)) {  // cannot happen (see above); never hang the device
      err |= 4u;
      break;
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

// Rows that leave the routed path for the spill list, as a real function (one copy per kernel, called from the wave-specialised scan
// loop's unrolled bodies and from the pair flavour of ring_route2): inlined, its per-accumulator transforms cost
// the headline's kernel 3 000 instructions and 23 more spilled scalar registers for rows that do not occur (round 6: found by
// tools/kernel_meta.py after a bench line came out 5 % slow).  Scalars only -- a struct by reference would have to live in scratch.
// mode: 0 one value as it is; 1 one RAW value, every accumulator's own transform of it (PTF_SHARED); 2 two values as they are
// (PTF_PAIR); 3 two RAW values, accumulator a takes operand (ops >> a) & 1 (PTF_PAIR | PTF_PLANES).  xf: val_xform[0..7], a byte each.
__device__ __attribute__((noinline)) void ws_slow_rows_call(uint32_t* ctrl, uint64_t* words, uint64_t capacity, uint32_t mode, uint32_t na, uint64_t xf,
                                                            uint32_t ops, bool mark_wide, bool slow, uint64_t key0, uint64_t val0, uint64_t val1) {
  if (mark_wide && __hip_atomic_load(&ctrl[CTRL_WIDE_KEYS], RLX_AGENT) == 0u) __hip_atomic_store(&ctrl[CTRL_WIDE_KEYS], 1u, RLX_AGENT);
  const uint64_t m = __ballot(slow);
  const int lane = lane_id();
  const int leader = __ffsll((unsigned long long)m) - 1;
  uint64_t base = 0;
  if (lane == leader) base = atomicAdd((unsigned long long*)&ctrl[CTRL_SPILL_LO], (unsigned long long)__popcll(m));
  base = __shfl(base, leader, 64);
  if (slow) {
    const uint64_t pos = base + (uint64_t)__popcll(m & ((1ull << lane) - 1ull));
    if (pos < capacity) {
      words[pos] = key0;
      if (mode == 0u) {
        words[capacity + pos] = val0;
      } else if (mode == 2u) {
        words[capacity + pos] = val0;
        words[2 * capacity + pos] = val1;
      } else {
#pragma unroll 1
        for (uint32_t a = 0; a < na; ++a)
          words[(uint64_t)(1u + a) * capacity + pos] = transform_value((uint8_t)(xf >> (8u * a)), (mode == 3u && ((ops >> a) & 1u)) ? val1 : val0, true);
      }
    }
  }
}
// ring_route for TWO batches of up to 64 rows at once (one routed value per row).  A call of ring_route is a chain of
// dependent LDS round trips -- fill atomic -> generation word -> ring row -> commit atomic -> job list -> ring read ->
// store: ~1 us per call whatever the row count, and a wave executes it alone.  When every wave has rows to route all the
// time (dense scans: config 3) or only a few waves route (the routers of the wave-specialised kernel) that latency IS
// the throughput.  Here the two batches' LDS operations are issued side by side: one round trip serves both.
// Same protocol, same invariants: positions come from the fill atomics (distinct for all 128 rows), a lane may own up to
// two flush jobs per turn, every job of the wave is executed before anybody retries, so a row that waits for a chunk
// slot of this wave's own other batch still gets it.
// KEY_IS_IMG: `key` holds the row's 32-bit hash IMAGE, not its key (the dense split of the wave-specialised kernel: the scanner has
// hashed the row and turned away what has no image); a row that leaves the routed path gets its key back (unhash_word32)
// DW = 5 (PTF_PAIR): the row carries a second operand, val2 (20-byte rows, six per 128-byte line: dfx_device.hpp)
template <int kRingCH, int kRingRP, int NARROW, bool KEY_IS_IMG = false, int DW = 3>
DEV void ring_route2(const DevTable& T, const DevPartition& PT, const DevRows& spill, const RingLds& L, uint32_t producer,
                     const bool (&have)[2], const uint64_t (&key)[2], const uint64_t (&val)[2], const uint64_t (&h)[2], uint32_t& err,
                     const uint64_t (&val2)[2]) {
  static_assert(DW == 3 || (DW == kPairRowDwords && NARROW != 0), "rows of three dwords, or the narrow pair rows");
  constexpr int kRingNCH = kRingRP / kRingCH;
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int NW = (int)PT.n_words;
  uint32_t part[2] = {0, 0}, pos[2] = {0, 0};
  bool pending[2] = {false, false}, todo[2] = {false, false}, narrow_ok[2];
  uint32_t img[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    img[b] = (uint32_t)(h[b] >> 32);
    narrow_ok[b] = KEY_IS_IMG || !(NARROW && ((key[b] >> 32) != 0 || img[b] >= kTagForeign));
    part[b] = partition_of(T, PT, h[b]);
  }
  {  // both fill atomics in flight together
    uint32_t got[2] = {0, 0};
#pragma unroll
    for (int b = 0; b < 2; ++b)
      if (have[b] && narrow_ok[b]) got[b] = atomicAdd(&L.fill[part[b]], 1u);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      if (have[b] && narrow_ok[b]) {
        pos[b] = got[b];
        pending[b] = pos[b] < PT.cap_rows;
        todo[b] = !pending[b];  // region overflow (skewed keys): the general path takes the row
      } else if (have[b]) {
        todo[b] = true;  // not representable as an image: the general path takes the row, and the host leaves narrow mode
        if (__hip_atomic_load(&T.ctrl[CTRL_WIDE_KEYS], RLX_AGENT) == 0u) __hip_atomic_store(&T.ctrl[CTRL_WIDE_KEYS], 1u, RLX_AGENT);
      }
    }
  }
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    if (__ballot(todo[b]) != 0) {  // (rare)
      uint64_t k1[1] = {KEY_IS_IMG ? (uint64_t)unhash_word32(img[b]) : key[b]};
      if constexpr (DW == kPairRowDwords) {  // two operands: the spill function (every accumulator's transform when the operands are raw)
        uint64_t xf = 0;
#pragma unroll
        for (int a = 0; a < kMaxAggs; ++a) xf |= (uint64_t)T.val_xform[a] << (8 * a);
        ws_slow_rows_call(T.ctrl, spill.words, spill.capacity, (PT.flags & PTF_PLANES) ? 3u : 2u, (uint32_t)T.na, xf, PT.pair_ops, false, todo[b], k1[0], val[b], val2[b]);
        continue;
      }
      uint64_t sv[kMaxAggs];
      if (NARROW && (PT.flags & PTF_SHARED)) {
        expand_shared_operand(T, val[b], sv);
      } else {
#pragma unroll
        for (int a = 0; a < kMaxAggs; ++a) sv[a] = a == 0 ? val[b] : 0ull;
      }
      spill_row<1>(T, spill, todo[b], k1, sv);
    }
  }
  uint32_t c[2], sl[2], g[2], r[2], cs[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    c[b] = pos[b] / kRingCH;
    sl[b] = c[b] % kRingNCH;
    g[b] = c[b] / kRingNCH;
    r[b] = pos[b] % kRingCH;
    cs[b] = part[b] * kRingNCH + sl[b];
  }
  uint32_t* jobs = L.jobs + wave * 128;  // (two batches: up to 128 jobs per turn; the job area holds 128 words per wave)
  uint32_t spins = 0;
  while (true) {
    bool job[2] = {false, false};
    uint32_t gen_now[2] = {0, 0};
#pragma unroll
    for (int b = 0; b < 2; ++b)
      if (pending[b]) gen_now[b] = __hip_atomic_load(&L.gen[cs[b]], __ATOMIC_ACQUIRE, WG_SCOPE);
    bool park[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      park[b] = pending[b] && gen_now[b] == g[b];
      if (park[b]) {
        if (NARROW) {
          uint32_t* d32 = (uint32_t*)L.ring + ring_dword12<kRingCH, kRingRP, NARROW, DW>(part[b], sl[b], r[b]);
          if constexpr (DW == kPairRowDwords) {
            d32[0] = (uint32_t)val[b];
            d32[1] = (uint32_t)(val[b] >> 32);
            d32[2] = img[b];
            d32[3] = (uint32_t)val2[b];
            d32[4] = (uint32_t)(val2[b] >> 32);
          } else {
            d32[0] = img[b];
            d32[1] = (uint32_t)val[b];
            d32[2] = (uint32_t)(val[b] >> 32);
          }
        } else {
          uint64_t* dst = L.ring + ((size_t)part[b] * kRingRP + sl[b] * kRingCH + r[b]) * NW;
          *(ulonglong2*)dst = make_ulonglong2(key[b], val[b]);
        }
      }
    }
    {  // both commit atomics in flight together
      uint32_t old[2] = {0, 0};
#pragma unroll
      for (int b = 0; b < 2; ++b)
        if (park[b]) old[b] = __hip_atomic_fetch_add(&L.commit[cs[b]], 1u, __ATOMIC_ACQ_REL, WG_SCOPE);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        if (park[b]) {
          job[b] = old[b] == (uint32_t)kRingCH - 1u;
          pending[b] = false;
        }
      }
    }
    const uint64_t jm0 = __ballot(job[0]), jm1 = __ballot(job[1]);
    if ((jm0 | jm1) != 0) {
      const uint32_t n0 = (uint32_t)__popcll(jm0), njobs = n0 + (uint32_t)__popcll(jm1);
      if (job[0]) jobs[mbcnt64(jm0)] = (part[0] << 20) | c[0];
      if (job[1]) jobs[n0 + mbcnt64(jm1)] = (part[1] << 20) | c[1];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      constexpr bool LINE = ring_is_line<kRingCH, NARROW, DW>();  // (see ring_route)
      constexpr uint32_t kJobLanes = LINE ? 8u : (uint32_t)kRingCH;
      for (uint32_t j0 = 0; j0 < njobs; j0 += 64 / kJobLanes) {
        const uint32_t j = j0 + (uint32_t)lane / kJobLanes;
        if (j < njobs) {
          const uint32_t jw = jobs[j];
          const uint2 jb = make_uint2(jw >> 20, jw & 0xFFFFFu);
          const uint32_t rr = (uint32_t)lane % kJobLanes;
          if constexpr (LINE) {
            // one global_store_dwordx4 per lane, eight adjacent lanes = one whole 128-byte line
            const uint4 piece = *(const uint4*)((const uint32_t*)L.ring + ring_dword12<kRingCH, kRingRP, NARROW, DW>(jb.x, jb.y % kRingNCH, 0) + rr * 4u);
            *(uint4*)(region_row12g<kRingCH, NARROW, DW>(PT, jb.x, producer, jb.y * kRingCH) + rr * 4u) = piece;
          } else if (NARROW) {
            const uint32_t* s32 = (const uint32_t*)L.ring + ((size_t)jb.x * kRingRP + (jb.y % kRingNCH) * kRingCH + rr) * 3;
            uint32_t* o32 = region_row12(PT, jb.x, producer, jb.y * kRingCH + rr);
            const uint32_t a = s32[0], b2 = s32[1], c3 = s32[2];
            // one global_store_dwordx3 per lane, 16 adjacent lanes = one 192-byte chunk.  (With the non-temporal hint the launch
            // took 417-480 us against 394-423 without, three alternating processes on one box -- probably because
            // consecutive 192-byte chunks of a region share every other 128-byte line, which L2 merges into whole-line writes.)
            o32[0] = a;
            o32[1] = b2;
            o32[2] = c3;
          } else {
            const uint64_t* src = L.ring + ((size_t)jb.x * kRingRP + (jb.y % kRingNCH) * kRingCH + rr) * NW;
            uint64_t* out = region_row(PT, jb.x, producer, jb.y * kRingCH + rr);
            *(ulonglong2*)out = *(const ulonglong2*)src;
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        if (job[b]) {  // the slot belongs to the next generation
          __hip_atomic_store(&L.commit[cs[b]], 0u, __ATOMIC_RELAXED, WG_SCOPE);
          __hip_atomic_fetch_add(&L.gen[cs[b]], 1u, __ATOMIC_RELEASE, WG_SCOPE);
        }
      }
    }
    if (__ballot(pending[0] || pending[1]) == 0) break;
    if (++spins > (1u << 22)) {  // cannot happen (see ring_route); never hang the device
      err |= 4u;
      break;
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

template <int kRingCH, int kRingRP, int NARROW, bool KEY_IS_IMG = false>
DEV void ring_route2(const DevTable& T, const DevPartition& PT, const DevRows& spill, const RingLds& L, uint32_t producer,
                     const bool (&have)[2], const uint64_t (&key)[2], const uint64_t (&val)[2], const uint64_t (&h)[2], uint32_t& err) {
  ring_route2<kRingCH, kRingRP, NARROW, KEY_IS_IMG, 3>(T, PT, spill, L, producer, have, key, val, h, err, val);
}

// ---- hot keys (skewed inputs) --------------------------------------------------------------------------
// A key that owns percent of the rows overflows its (producer, partition) regions (the overflow goes through the spill
// list and global same-address atomics at ~11 ns each) and makes 64 lanes of pass 2 queue on one LDS address.  With
// PTF_HOT every pass-1 workgroup keeps kHotSlots direct-mapped (key, accumulator) pairs in LDS, claimed once by the
// first key that hashes there -- under a Zipf-like distribution the heavy keys turn up within the first rows and take
// their slots -- and rows of a cached key are added there instead of being routed.  At the end every claimed slot is
// routed as ONE row (key, partial accumulator): the accumulator algebra is associative, so pass 2 cannot tell.  The
// slot index comes from the LOW hash bits (the partition from the high ones).  Switched on by the calibration slice
// (a front cache that absorbs a sizeable share of the rows means skew); uniform keys never pay for it.
// Up to kHotWays consecutive slots are tried (first free one is claimed): with one slot per hash two of the top-64 keys
// collide with probability ~0.9 in 1024 slots, and a heavy key that loses its slot overflows its regions again (measured:
// Zipf without a filter 4.7x slower in pass 1 and 12 ms of spill replay per query).
constexpr int kHotWays = 4;
template <int SLOTS>
DEV bool hot_absorb(uint64_t* hot_keys, uint64_t* hot_accs, uint8_t kind, uint64_t h, uint64_t key, uint64_t val) {
  uint32_t hs = (uint32_t)(h >> 32) & (uint32_t)(SLOTS - 1);
#pragma unroll
  for (int w = 0; w < kHotWays; ++w) {
    uint64_t cur = hot_keys[hs];
    if (cur == kEmptyKey) {
      const uint64_t old = atomicCAS((unsigned long long*)&hot_keys[hs], (unsigned long long)kEmptyKey, (unsigned long long)key);
      cur = old == kEmptyKey ? key : old;
    }
    if (cur == key) {
      acc_atomic(kind, &hot_accs[hs], val);
      return true;
    }
    hs = (hs + 1u) & (uint32_t)(SLOTS - 1);
  }
  return false;
}

template <typename POL, int kRingCH, int kRingRP, bool HOT = false, int NARROW = 0, int QROWS = 0>
__global__ __launch_bounds__(kRingBlock) void k_partition_ring(const DevProgram P, const DevFastPlan F, const DevColumns C,
                                                              const DevAggPlan plan, const DevTable T,
                                                              const DevPartition PT, const DevRows spill, const int64_t n) {
  typedef typename POL::COLV COLV;
  constexpr int U = POL::U;
  static_assert(U == 1 || U == 2 || U == 4, "row-groups per trip");
  constexpr int NWAVES = kRingBlock / 64;
  constexpr int NV = POL::kStaticNa == 1 ? 1 : kMaxAggs;
  constexpr int kRingNCH = kRingRP / kRingCH;
  constexpr int kRingQ = QROWS > 0 ? QROWS : ring_queue_rows(kRingRP);
  extern __shared__ __attribute__((aligned(16))) uint64_t lds[];
  const int NW = (int)PT.n_words;
  RingLds L;
  L.ring = lds;
  const size_t ring_words = NARROW ? ring_dwords12<kRingCH, kRingRP, NARROW>(PT.n_parts) / 2 : (size_t)PT.n_parts * kRingRP * NW;  // 12-byte rows
  L.queue = L.ring + ring_words;
  L.jobs = (uint32_t*)(L.queue + (size_t)NWAVES * kRingQ * NW);
  L.fill = (uint32_t*)(L.jobs + NWAVES * 64 * (kRingRP >= 16 ? 2 : 1));
  L.commit = L.fill + PT.n_parts;
  L.gen = L.commit + (size_t)PT.n_parts * 4;
  // hot-key pairs behind everything else, 16-byte aligned (as an offset from `lds`: keeps the LDS address space)
  const size_t hot_word0 = (ring_words + (size_t)NWAVES * kRingQ * NW) +
                           ((size_t)(NWAVES * 64 * (kRingRP >= 16 ? 2 : 1) + PT.n_parts * (1 + 2 * 4)) * 4 + 15) / 16 * 2;
  constexpr int kHot = NARROW ? kHotSlotsNarrow : kHotSlots;
  uint64_t* hot_keys = lds + hot_word0;
  uint64_t* hot_accs = hot_keys + kHot;
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  // PTF_SHARED: one routed value (the aggregates' common raw operand) whatever the aggregate count
  const bool shared = NARROW && (PT.flags & PTF_SHARED) != 0;
  const int na = (POL::kStaticNa == 1 || shared) ? 1 : POL::na(T);
  if (HOT) {
    for (uint32_t i = threadIdx.x; i < (uint32_t)kHot; i += kRingBlock) {
      hot_keys[i] = kEmptyKey;
      hot_accs[i] = T.acc_init[0];
    }
  }
  const uint32_t producer = blockIdx.x;
  // fill, commit, gen.  PTF_RESUME: pass 2 of the earlier batches is still pending and this producer's regions already
  // hold counts[] rows -- a whole number of chunks, the epilogue below pads -- so chunk numbering goes on from there:
  // ring slot sl first serves the lowest chunk >= c0 that maps to it
  for (uint32_t p = threadIdx.x; p < PT.n_parts; p += kRingBlock) {
    const uint32_t f0 = (PT.flags & PTF_RESUME) ? PT.counts[(uint64_t)p * PT.n_producers + producer] : 0u;
    const uint32_t c0 = f0 / kRingCH;
    L.fill[p] = f0;
#pragma unroll
    for (int sl = 0; sl < kRingNCH; ++sl) {
      L.commit[p * kRingNCH + sl] = 0;
      L.gen[p * kRingNCH + sl] = (c0 + (uint32_t)(kRingNCH - 1 - sl)) / (uint32_t)kRingNCH;
    }
  }
  __syncthreads();
  uint64_t* q = L.queue + (size_t)wave * kRingQ * NW;  // word-major planes [NW][kRingQ]
  uint32_t qn = 0;                                     // queued rows (wave-uniform)
  const int64_t n_groups = (n + 63) >> 6;
  const int64_t wave_global = (int64_t)blockIdx.x * NWAVES + wave;
  const int64_t n_waves = (int64_t)gridDim.x * NWAVES;
  uint32_t err = 0;
  uint64_t passed = 0;
  // software pipeline, kRingDepth trips deep: the columns of trips t + 1 .. t + kRingDepth are in flight while trip t
  // is evaluated and routed.  Depth 1 = 16 waves per CU x 4 KB = 16 MB in flight chip-wide.  Depths 2 and 3 (92 / 104
  // VGPRs, -DDFX_RING_DEPTH) were measured: 3.46-3.66 ms per 1e9 rows against 3.24-3.62 ms at depth 1 -- no gain, the
  // kernel is not short of bytes in flight; the run-to-run spread (+-6 % on one box) is larger than any difference.
  constexpr int kRingDepth = DFX_RING_DEPTH;
  typename POL::PREP prep;  // (PlanPolicy: the plan words in vector registers; empty otherwise)
  POL::prepare(P, F, prep);
  COLV ncol[kRingDepth][U];
  uint32_t ncv[kRingDepth][U];
#pragma unroll
  for (int d = 0; d < kRingDepth; ++d) {
    const int64_t w0 = wave_global * U + (int64_t)d * n_waves * U;
    load_trip<POL>(P, C, w0, w0 < n_groups, n, lane, ncol[d], ncv[d]);  // (static signatures: one scalar base per column, clamped 32-bit lane indices)
  }
  for (int64_t w0 = wave_global * U; w0 < n_groups; w0 += n_waves * U) {
    COLV col[U];
    uint32_t cv[U];
    FOR_U {
      col[u] = ncol[0][u];
      cv[u] = ncv[0][u];
    }
#pragma unroll
    for (int d = 0; d + 1 < kRingDepth; ++d) {
      FOR_U {
        ncol[d][u] = ncol[d + 1][u];
        ncv[d][u] = ncv[d + 1][u];
      }
    }
    {
      const int64_t w1 = w0 + (int64_t)kRingDepth * n_waves * U;
      load_trip<POL>(P, C, w1, w1 < n_groups, n, lane, ncol[kRingDepth - 1], ncv[kRingDepth - 1]);
    }
#ifdef DFX_RING_WAIT_ALL
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    // (Routing two full groups side by side straight from registers -- ring_route2, no compaction queue -- was tried for
    // dense scans without a predicate, config 3's signature: 424-429 us per 2^26-row launch against 373-414 for the queue +
    // one batch per call below.  128 rows in flight per wave x 16 waves outrun the 32-row rings: lanes wait for chunk
    // slots and hold BOTH batches up.  The routers of the wave-specialised kernel -- eight waves -- use it: no measurable
    // difference there, 388-408 us per launch with it, 383-413 without.)
    FOR_U {
      const int64_t row = (w0 + u) * 64 + lane;
      const bool inb = row < n;
      u64x16 reg;
      uint32_t rv = 0;
      POL::eval(P, F, col[u], cv[u], reg, rv, inb, err, prep);
      bool pass = inb && POL::pass(P, F, plan.pred, col[u], cv[u], reg, rv, prep);
      uint64_t key[1];
      uint64_t val[kMaxAggs];
      key[0] = POL::key(P, F, plan.key[0], 0, col[u], cv[u], reg, rv);
#pragma unroll
      for (int a = 0; a < kMaxAggs; ++a) {
        val[a] = 0;
        if (a < na) {
          uint64_t v;
          bool valid;
          POL::arg(P, F, plan.arg[a], a, col[u], cv[u], reg, rv, v, valid);
          val[a] = transform_value(shared ? (uint8_t)VT_RAW : POL::xform(T, a), v, valid);
        }
      }
      passed += pass ? 1 : 0;
      if (__ballot(pass && key[0] == kEmptyKey) != 0) {  // the claim-sentinel key lives outside the blocks
        if (pass && key[0] == kEmptyKey) {
          if (shared) {
            uint64_t sv[kMaxAggs];
            expand_shared_operand(T, val[0], sv);
            sentinel_apply(T, sv);
          } else {
            sentinel_apply(T, val);
          }
          pass = false;
        }
      }
      const uint64_t m = __ballot(pass);
      // A row group whose 64 rows ALL pass, met with an empty queue (dense scans: config 3 has no predicate at all), is routed
      // straight from registers: the compaction queue would only copy it (two LDS writes, two LDS reads, the rank arithmetic
      // per row group).  Same call site as the queue's batches, so no second copy of the ring protocol.
      const bool direct = m == ~0ull && qn == 0;
      if (pass && !direct) {
        const uint32_t at = qn + mbcnt64(m);
        q[at] = key[0];
#pragma unroll
        for (int a = 0; a < NV; ++a)
          if (a < na) q[(size_t)(1 + a) * kRingQ + at] = val[a];
      }
      if (!direct) qn += (uint32_t)__popcll(m);
      if (direct || kRingQ < 192 || (u & 1) == 1 || u == U - 1) {  // the queue holds < 64 + (kRingQ - 64) rows
        while (direct || qn >= 64) {
          uint64_t k2[1];
          uint64_t v2[kMaxAggs];
          if (direct) {
            k2[0] = key[0];
#pragma unroll
            for (int a = 0; a < kMaxAggs; ++a) v2[a] = (a < NV && a < na) ? val[a] : 0;
          } else {
            qn -= 64;
            k2[0] = q[qn + lane];
#pragma unroll
            for (int a = 0; a < kMaxAggs; ++a) v2[a] = (a < NV && a < na) ? q[(size_t)(1 + a) * kRingQ + qn + lane] : 0;
          }
          const uint64_t h2 = hash_keys<1>(k2);
          bool have2 = true;
          if (HOT && NV == 1) have2 = !hot_absorb<kHot>(hot_keys, hot_accs, POL::acc_kind(T, 0), h2, k2[0], v2[0]);
          ring_route<NV, kRingCH, kRingRP, NARROW>(T, PT, spill, L, producer, na, have2, k2, v2, h2, err);
          if (direct) break;
        }
      }
    }
  }
  {  // the wave's last < 64 rows
    uint64_t k2[1];
    uint64_t v2[kMaxAggs];
    bool have = (uint32_t)lane < qn;
    k2[0] = have ? q[lane] : 0;
#pragma unroll
    for (int a = 0; a < kMaxAggs; ++a) v2[a] = (have && a < NV && a < na) ? q[(size_t)(1 + a) * kRingQ + lane] : 0;
    const uint64_t h2 = hash_keys<1>(k2);
    if (HOT && NV == 1 && have) have = !hot_absorb<kHot>(hot_keys, hot_accs, POL::acc_kind(T, 0), h2, k2[0], v2[0]);
    if (qn != 0) ring_route<NV, kRingCH, kRingRP, NARROW>(T, PT, spill, L, producer, na, have, k2, v2, h2, err);
  }
  if (HOT && NV == 1) {  // every claimed hot slot becomes one routed row (key, partial accumulator)
    __syncthreads();     // all waves have finished absorbing
    for (int i0 = 0; i0 < kHot; i0 += kRingBlock) {
      uint64_t k2[1];
      uint64_t v2[kMaxAggs];
#pragma unroll
      for (int a = 0; a < kMaxAggs; ++a) v2[a] = 0;
      k2[0] = hot_keys[i0 + threadIdx.x];
      v2[0] = hot_accs[i0 + threadIdx.x];
      const bool have = k2[0] != kEmptyKey;
      if (__ballot(have) != 0) ring_route<NV, kRingCH, kRingRP, NARROW>(T, PT, spill, L, producer, na, have, k2, v2, hash_keys<1>(k2), err);
    }
  }
  __syncthreads();
  // partial chunks + region counts.  A partial chunk is padded to a whole one with rows whose key is kEmptyKey (pass 2
  // skips them; a real row never carries that key, it lives in slot `cap`), so that a later launch can go on appending
  // at a chunk boundary (PTF_RESUME).  cap_rows is a multiple of 64, so the padding never leaves the region.
  uint32_t max_fill = 0;
  for (uint32_t p = threadIdx.x; p < PT.n_parts; p += kRingBlock) {
    uint32_t f = L.fill[p];
    if (f > PT.cap_rows) f = PT.cap_rows;
    const uint32_t c = f / kRingCH;
    const uint32_t rem = f % kRingCH;
    for (uint32_t r = 0; r < rem; ++r) {
      if (NARROW) {
        const uint32_t* s32 = (const uint32_t*)L.ring + ring_dword12<kRingCH, kRingRP, NARROW>(p, c % kRingNCH, r);
        uint32_t* o32 = region_row12g<kRingCH, NARROW>(PT, p, producer, c * kRingCH + r);
        for (int w = 0; w < 3; ++w) o32[w] = s32[w];
      } else {
        const uint64_t* src = L.ring + ((size_t)p * kRingRP + (c % kRingNCH) * kRingCH + r) * NW;
        uint64_t* out = region_row(PT, p, producer, c * kRingCH + r);
        for (int w = 0; w < NW; ++w) out[w] = src[w];
      }
    }
    if (rem != 0) {
      for (uint32_t r = rem; r < (uint32_t)kRingCH; ++r) {
        if (NARROW) {
          uint32_t* o32 = region_row12g<kRingCH, NARROW>(PT, p, producer, c * kRingCH + r);
          o32[0] = kTagEmpty;
          o32[1] = o32[2] = 0;
        } else {
          uint64_t* out = region_row(PT, p, producer, c * kRingCH + r);
          out[0] = kEmptyKey;
          for (int w = 1; w < NW; ++w) out[w] = 0;
        }
      }
      f = (c + 1) * kRingCH;
    }
    PT.counts[(uint64_t)p * PT.n_producers + producer] = f;
    max_fill = f > max_fill ? f : max_fill;
  }
  publish_max_fill(T, max_fill);
#pragma unroll
  for (int mm = 32; mm >= 1; mm >>= 1) passed += shfl_xor_u64(passed, mm);
  if (lane == 0) stat_add(T, STAT_PASSED, passed);
  if (err) atomicOr(&T.ctrl[CTRL_ERROR], err);
  snapshot_ctrl_if_last(T, PT);
}

template <typename POL, typename POLS, typename POLN = POLS>  // POLS: the policy flavour used by the write-combining kernel; POLN: by its narrow-row flavours (one routed value)
void launch_partition_pol(const DevProgram& P, const DevFastPlan& fast, const DevColumns& C, const DevAggPlan& plan,
                                 const DevTable& T, const DevPartition& PT, const DevRows& spill, int64_t n,
                                 size_t lds_bytes, hipStream_t s) {
  const int grid = (int)PT.n_producers;  // every producer writes its counts, even with no rows
  if ((PT.mode & 15u) == 0)
    hipLaunchKernelGGL((k_partition<POL>), dim3(grid), dim3(kPBlock), lds_bytes, s, P, fast, C, plan, T, PT, spill, n);
  else if ((PT.mode & 15u) == 2 && (PT.mode & 0x100u))  // rows of 3+ words: 8-row rings of 4-row chunks (a 16-row ring of
                                                       // 24-byte rows + its queues do not fit LDS; the counting sort is 2-3 x slower)
    hipLaunchKernelGGL((k_partition_ring<POLS, 4, 8>), dim3(grid), dim3(kRingBlock), lds_bytes, s, P, fast, C, plan, T, PT, spill, n);
  else if ((PT.mode & 15u) == 2 && (PT.mode & 0x80u))  // 4-row chunks (64-byte runs)
    hipLaunchKernelGGL((k_partition_ring<POLS, 4, 16>), dim3(grid), dim3(kRingBlock), lds_bytes, s, P, fast, C, plan, T, PT, spill, n);
  else if ((PT.mode & 15u) == 2 && (PT.flags & PTF_NARROW) && (PT.flags & PTF_SHARED))
    // 2..3 aggregates of one operand: 4096-slot blocks, so twice the partitions -- 128-row wave queues make room for their rings
    hipLaunchKernelGGL((k_partition_ring<POLN, 8, 16, false, 1, 128>), dim3(grid), dim3(kRingBlock), lds_bytes, s, P, fast, C, plan, T, PT, spill, n);
  else if ((PT.mode & 15u) == 2 && (PT.flags & PTF_NARROW) && (PT.flags & PTF_HOT))
    hipLaunchKernelGGL((k_partition_ring<POLN, 8, 16, true, 1>), dim3(grid), dim3(kRingBlock), lds_bytes, s, P, fast, C, plan, T, PT, spill, n);
  else if ((PT.mode & 15u) == 2 && (PT.flags & PTF_NARROW) && (PT.flags & PTF_CHUNK16))
    // the large-chunk geometry (dfx_device.hpp: kNarrowLine): whole 128-byte lines of ten rows; rounds 3-5: 16-row chunks of 192 bytes
    // (three whole 64-byte sectors; a 96-byte chunk of 8 rows straddles sectors: WRITE_SIZE was 1.26 x the routed bytes)
    hipLaunchKernelGGL((k_partition_ring<POLN, kNarrowChunkRows, kNarrowRingRows, false, 1>), dim3(grid), dim3(kRingBlock), lds_bytes, s, P, fast, C, plan, T, PT, spill, n);
  else if ((PT.mode & 15u) == 2 && (PT.flags & PTF_NARROW))  // narrow keys: 12-byte rows
    hipLaunchKernelGGL((k_partition_ring<POLN, 8, 16, false, 1>), dim3(grid), dim3(kRingBlock), lds_bytes, s, P, fast, C, plan, T, PT, spill, n);
  else if ((PT.mode & 15u) == 2 && (PT.flags & PTF_HOT))  // skewed keys: hot-key pairs in LDS
    hipLaunchKernelGGL((k_partition_ring<POLS, 8, 16, true>), dim3(grid), dim3(kRingBlock), lds_bytes, s, P, fast, C, plan, T, PT, spill, n);
  else if ((PT.mode & 15u) == 2)  // 8-row chunks (full 128-byte lines): ~4 % faster on MI355X
    hipLaunchKernelGGL((k_partition_ring<POLS, 8, 16>), dim3(grid), dim3(kRingBlock), lds_bytes, s, P, fast, C, plan, T, PT, spill, n);
  else if (PT.block == 512)
    hipLaunchKernelGGL((k_partition_sorted<POLS, 512>), dim3(grid), dim3(512), lds_bytes, s, P, fast, C, plan, T, PT, spill, n);
  else
    hipLaunchKernelGGL((k_partition_sorted<POLS, 1024>), dim3(grid), dim3(1024), lds_bytes, s, P, fast, C, plan, T, PT, spill, n);
}

// one pass-1 variant = one translation unit
#define DFX_PARTITION_VARIANT_ARGS                                                                                         \
  const DevProgram &P, const DevFastPlan &fast, const DevColumns &C, const DevAggPlan &plan, const DevTable &T,            \
      const DevPartition &PT, const DevRows &spill, int64_t n, size_t lds_bytes, hipStream_t s
#define DFX_PARTITION_VARIANT(ID, POL, POLS, ...)                                                                          \
  void launch_partition_variant##ID(DFX_PARTITION_VARIANT_ARGS) {                                                          \
    launch_partition_pol<POL, POLS, ##__VA_ARGS__>(P, fast, C, plan, T, PT, spill, n, lds_bytes, s);                      \
  }

}  // namespace dfx
