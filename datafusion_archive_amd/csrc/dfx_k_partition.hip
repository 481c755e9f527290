// dfx_k_partition.hip -- the partitioned GROUP BY strategy for high-cardinality single-key
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
#include "dfx_kernels_inl.hpp"
#include "dfx_launch.hpp"

namespace dfx {

constexpr int kPBlock = 512;  // pass-1 workgroup: 8 waves share one set of fill counters and 64 KB of LDS

DEV uint32_t partition_of(const DevTable& T, const DevPartition& PT, uint64_t h) {
  return (uint32_t)(((h >> T.shift) & T.mask) >> PT.part_shift);
}

// One WAVE flushes its own staging area: every staged row is appended to its (producer,
// partition) region; the per-partition fill counters are shared by the workgroup's four waves
// (LDS atomics), the regions have no other writer, so no global atomic and no barrier is needed.
template <typename POL>
DEV void partition_flush_wave(const DevTable& T, const DevPartition& PT, const DevRows& spill,
                              const uint64_t* stage, uint32_t* fill, uint32_t cnt, uint32_t producer) {
  const int NW = (int)PT.n_words;
  const int lane = lane_id();
  const uint32_t cnt_pad = (cnt + 63u) & ~63u;  // the whole wave stays in the loop (ballots in spill_row)
  for (uint32_t i = (uint32_t)lane; i < cnt_pad; i += 64u) {
    const bool inb = i < cnt;
    uint64_t key[1];
    uint64_t val[kMaxAggs];
    key[0] = inb ? stage[(size_t)i * NW] : 0;
#pragma unroll
    for (int a = 0; a < kMaxAggs; ++a) val[a] = (inb && a < POL::na(T)) ? stage[(size_t)i * NW + 1 + a] : 0;
    bool todo = inb;
    if (inb) {
      const uint32_t p = partition_of(T, PT, hash_keys<1>(key));
      const uint32_t pos = atomicAdd(&fill[p], 1u);  // LDS atomic: arrival rank inside the region
      if (pos < PT.cap_rows) {
        uint64_t* dst = PT.rows + (((uint64_t)p * PT.n_producers + producer) * PT.cap_rows + pos) * NW;
        dst[0] = key[0];
        for (int a = 0; a < POL::na(T); ++a) dst[1 + a] = val[a];
        todo = false;
      }
    }
    spill_row<1>(T, spill, todo, key, val);  // region overflow (skewed keys): the general path takes it
  }
}

template <typename POL>
__global__ __launch_bounds__(kPBlock) void k_partition(const DevProgram P, const DevFastPlan F, const DevColumns C,
                                                      const DevAggPlan plan, const DevTable T,
                                                      const DevPartition PT, const DevRows spill, const int64_t n) {
  typedef typename POL::COLV COLV;
  constexpr int U = POL::U;
  extern __shared__ __attribute__((aligned(16))) uint64_t lds[];
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int NW = (int)PT.n_words;
  const uint32_t RSW = PT.stage_rows;
  uint64_t* stage = lds + (size_t)wave * RSW * NW;                          // this wave's rows [RSW][NW]
  uint32_t* fill = (uint32_t*)(lds + (size_t)(kPBlock / 64) * RSW * NW);     // [n_parts], shared
  for (uint32_t p = threadIdx.x; p < PT.n_parts; p += kPBlock) fill[p] = 0;
  __syncthreads();
  const uint32_t producer = blockIdx.x;
  const int64_t n_groups = (n + 63) >> 6;
  const int64_t wave_global = (int64_t)blockIdx.x * (kPBlock / 64) + wave;
  const int64_t n_waves = (int64_t)gridDim.x * (kPBlock / 64);
  uint32_t err = 0;
  uint64_t passed = 0;
  uint32_t scnt = 0;  // rows staged by this wave (wave-uniform)
  for (int64_t w0 = wave_global * U; w0 < n_groups; w0 += n_waves * U) {
    COLV col[U];
    uint32_t cv[U];
    FOR_U {
      const int64_t row = (w0 + u) * 64 + lane;
      POL::load(P, C, row, row < n, col[u], cv[u]);
    }
#pragma nounroll
    for (int uu = 0; uu < U; ++uu) {
      COLV cur;
      uint32_t curv;
      DFX_SELECT_BANK(uu, col, cv, cur, curv)
      const int64_t row = (w0 + uu) * 64 + lane;
      const bool inb = row < n;
      u64x16 reg;
      uint32_t rv = 0;
      POL::eval(P, F, cur, curv, reg, rv, inb, err);
      const bool pass = inb && POL::pass(P, F, plan.pred, cur, curv, reg, rv);
      uint64_t key[1];
      uint64_t val[kMaxAggs];
      key[0] = POL::key(P, F, plan.key[0], 0, cur, curv, reg, rv);
#pragma unroll
      for (int a = 0; a < kMaxAggs; ++a) {
        val[a] = 0;
        if (a < POL::na(T)) {
          uint64_t v;
          bool valid;
          POL::arg(P, F, plan.arg[a], a, cur, curv, reg, rv, v, valid);
          val[a] = transform_value(POL::xform(T, a), v, valid);
        }
      }
      passed += pass ? 1 : 0;
      bool stage_it = pass;
      if (pass && key[0] == kEmptyKey) {  // the claim-sentinel key lives outside the blocks
        stage_it = false;
        const bool ok = table_apply<1>(T, key, val);
        (void)ok;
      }
      // wave-private staging: position = running count + rank among the passing lanes
      const uint64_t m = __ballot(stage_it);
      if (stage_it) {
        const uint32_t pos = scnt + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        stage[(size_t)pos * NW] = key[0];
#pragma unroll
        for (int a = 0; a < kMaxAggs; ++a)
          if (a < POL::na(T)) stage[(size_t)pos * NW + 1 + a] = val[a];
      }
      scnt += (uint32_t)__popcll(m);
    }
    if (scnt + (uint32_t)(U * 64) > RSW) {  // the next trip might not fit (wave-uniform)
      partition_flush_wave<POL>(T, PT, spill, stage, fill, scnt, producer);
      scnt = 0;
    }
  }
  if (scnt > 0) partition_flush_wave<POL>(T, PT, spill, stage, fill, scnt, producer);
  __syncthreads();
  for (uint32_t p = threadIdx.x; p < PT.n_parts; p += kPBlock) {
    const uint32_t f = fill[p];
    PT.counts[(uint64_t)p * PT.n_producers + producer] = f < PT.cap_rows ? f : PT.cap_rows;
  }
#pragma unroll
  for (int mm = 32; mm >= 1; mm >>= 1) passed += shfl_xor_u64(passed, mm);
  if (lane == 0 && passed) atomicAdd((unsigned long long*)&T.ctrl[CTRL_PASSED_LO], (unsigned long long)passed);
  if (err) atomicOr(&T.ctrl[CTRL_ERROR], err);
}

// pass 2: one workgroup per partition (= table block).  The rows of all producers are visited as
// ONE flattened index space (prefix sums of the per-producer counts live in LDS), so all 256
// lanes stay busy however small the individual regions are.
__global__ __launch_bounds__(kBlock) void k_partition_agg(const DevTable T, const DevPartition PT, const DevRows spill) {
  extern __shared__ __attribute__((aligned(16))) uint64_t lds[];
  __shared__ uint32_t wave_tot[kBlock / 64];
  const uint32_t S = T.block_mask + 1;
  const int NW = (int)PT.n_words;
  uint64_t* lkeys = lds;
  uint64_t* laccs = lds + S;
  uint32_t* pre = (uint32_t*)(lds + (size_t)S * NW);  // [n_producers + 1] exclusive prefix of counts
  const uint32_t p = blockIdx.x;
  const uint64_t slot0 = (uint64_t)p * S;
  const int lane = lane_id();
  for (uint32_t i = threadIdx.x; i < S; i += kBlock) {
    lkeys[i] = T.keys[slot0 + i];
    for (int a = 0; a < T.na; ++a) laccs[(size_t)a * S + i] = T.accs[(uint64_t)a * T.stride + slot0 + i];
  }
  // exclusive scan of the producer counts (n_producers <= 1024: up to 4 per thread)
  const uint32_t NP = PT.n_producers;
  uint32_t c[4], tsum = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t w = threadIdx.x * 4 + j;
    c[j] = w < NP ? PT.counts[(uint64_t)p * NP + w] : 0u;
    tsum += c[j];
  }
  uint32_t inc = tsum;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(inc, d, 64);
    if (lane >= d) inc += o;
  }
  if (lane == 63) wave_tot[threadIdx.x >> 6] = inc;
  __syncthreads();
  uint32_t base = 0;
  for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += wave_tot[w];
  const uint32_t total = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
  uint32_t run = base + inc - tsum;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t w = threadIdx.x * 4 + j;
    if (w < NP) pre[w] = run;
    run += c[j];
  }
  if (threadIdx.x == 0) pre[NP] = total;
  __syncthreads();
  uint32_t new_keys = 0;
  const uint32_t total_pad = (total + 63u) & ~63u;
  for (uint32_t i = threadIdx.x; i < total_pad; i += kBlock) {
    const bool inb = i < total;
    uint64_t key[1];
    uint64_t val[kMaxAggs];
    key[0] = 0;
#pragma unroll
    for (int a = 0; a < kMaxAggs; ++a) val[a] = 0;
    bool todo = inb;
    if (inb) {
      uint32_t lo = 0, hi = NP;  // largest w with pre[w] <= i
      while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (pre[mid] <= i) lo = mid; else hi = mid;
      }
      const uint64_t* src = PT.rows + (((uint64_t)p * NP + lo) * PT.cap_rows + (i - pre[lo])) * NW;
      key[0] = src[0];
#pragma unroll
      for (int a = 0; a < kMaxAggs; ++a)
        if (a < T.na) val[a] = src[1 + a];
      const uint64_t h = hash_keys<1>(key);
      uint32_t slot = (uint32_t)((h >> T.shift) & T.mask) & T.block_mask;
      int found = -1;
      for (uint32_t pr = 0; pr < S && found < 0; ++pr) {
        const uint64_t k = lkeys[slot];
        if (k == key[0]) {
          found = (int)slot;
        } else if (k == kEmptyKey) {
          const uint64_t old = atomicCAS((unsigned long long*)&lkeys[slot], (unsigned long long)kEmptyKey,
                                         (unsigned long long)key[0]);
          if (old == kEmptyKey) {
            found = (int)slot;
            ++new_keys;
          } else if (old == key[0]) {
            found = (int)slot;
          }
        }
        if (found < 0) slot = (slot + 1) & T.block_mask;
      }
      if (found >= 0) {
#pragma unroll
        for (int a = 0; a < kMaxAggs; ++a)
          if (a < T.na) acc_atomic(T.acc_kind[a], &laccs[(size_t)a * S + found], val[a]);
        todo = false;
      }
    }
    spill_row<1>(T, spill, todo, key, val);  // block full: grow-and-replay takes it
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < S; i += kBlock) {
    T.keys[slot0 + i] = lkeys[i];
    for (int a = 0; a < T.na; ++a) T.accs[(uint64_t)a * T.stride + slot0 + i] = laccs[(size_t)a * S + i];
  }
#pragma unroll
  for (int mm = 32; mm >= 1; mm >>= 1) new_keys += __shfl_xor(new_keys, mm, 64);
  if (lane == 0 && new_keys) atomicAdd(&T.ctrl[CTRL_OCCUPIED], new_keys);
}

size_t partition_stage_bytes(const DevPartition& PT) {
  return (size_t)(kPBlock / 64) * PT.n_words * PT.stage_rows * 8 + (size_t)PT.n_parts * 4 + 16;
}

hipError_t launch_partition(const DevProgram& P, const DevFastPlan& fast, const DevColumns& C, const DevAggPlan& plan,
                            const DevTable& T, const DevPartition& PT, const DevRows& spill, int64_t n,
                            double algo_bytes, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  Scope sc(KID_PARTITION, s, algo_bytes);
  const size_t lds_bytes = partition_stage_bytes(PT);
  const int grid = (int)PT.n_producers;  // every producer writes its counts, even with no rows
  if (lds_bytes > 65536) return hipErrorInvalidValue;  // the host sizes the plan to fit (ensure_partition)
#define DFX_PT(POL) hipLaunchKernelGGL((k_partition<POL>), dim3(grid), dim3(kPBlock), lds_bytes, s, P, fast, C, plan, T, PT, spill, n)
  if (PT.stage_rows >= 256 && sig_matches<SigKeySumPred2F64>(P, fast, 1, T.na, T.acc_kind, T.val_xform)) {
    DFX_PT(DFX_ARG(StaticPolicy<2, 4, SigKeySumPred2F64>));
    return hipGetLastError();
  }
  if (PT.stage_rows >= 256 && sig_matches<SigKeySum>(P, fast, 1, T.na, T.acc_kind, T.val_xform)) {
    DFX_PT(DFX_ARG(StaticPolicy<2, 4, SigKeySum>));
    return hipGetLastError();
  }
  const bool use_fast = fast.valid && !P.has_nulls;
  // a wave trip is U x 64 rows and must fit the wave's staging area: wide rows use U = 2
  if (PT.stage_rows >= 256) {
    if (P.n_cols <= 2) { if (use_fast) DFX_PT(DFX_ARG(FastPolicy<2, 4>)); else DFX_PT(DFX_ARG(InterpPolicy<2, 4>)); }
    else if (P.n_cols <= 4) { if (use_fast) DFX_PT(DFX_ARG(FastPolicy<4, 4>)); else DFX_PT(DFX_ARG(InterpPolicy<4, 4>)); }
    else { if (use_fast) DFX_PT(DFX_ARG(FastPolicy<8, 4>)); else DFX_PT(DFX_ARG(InterpPolicy<8, 4>)); }
  } else {
    if (P.n_cols <= 4) { if (use_fast) DFX_PT(DFX_ARG(FastPolicy<4, 2>)); else DFX_PT(DFX_ARG(InterpPolicy<4, 2>)); }
    else { if (use_fast) DFX_PT(DFX_ARG(FastPolicy<8, 2>)); else DFX_PT(DFX_ARG(InterpPolicy<8, 2>)); }
  }
#undef DFX_PT
  return hipGetLastError();
}

hipError_t launch_partition_agg(const DevTable& T, const DevPartition& PT, const DevRows& spill, double algo_bytes,
                                hipStream_t s) {
  Scope sc(KID_PARTITION_AGG, s, algo_bytes);
  const size_t lds_bytes = (size_t)(T.block_mask + 1) * (size_t)(1 + T.na) * 8 + (size_t)(PT.n_producers + 1) * 4 + 16;
  if (lds_bytes > 65536 || PT.n_producers > 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_partition_agg, dim3(PT.n_parts), dim3(kBlock), lds_bytes, s, T, PT, spill);
  return hipGetLastError();
}

}  // namespace dfx
