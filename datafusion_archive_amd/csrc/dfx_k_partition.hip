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

constexpr int kPBlock = 1024;  // pass-1 workgroup (one per CU): 16 waves share one set of fill counters
constexpr int kABlock = 1024;  // pass-2 workgroup (one per CU): 16 waves share a 128 KB LDS copy of a table block

DEV uint32_t partition_of(const DevTable& T, const DevPartition& PT, uint64_t h) {
  return (uint32_t)(((h >> T.shift) & T.mask) >> PT.part_shift);
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
  for (uint32_t p = threadIdx.x; p < PT.n_parts; p += kPBlock) fill[p] = 0;
  __syncthreads();
  const uint32_t producer = blockIdx.x;
  const int64_t n_groups = (n + 63) >> 6;
  const int64_t wave_global = (int64_t)blockIdx.x * (kPBlock / 64) + wave;
  const int64_t n_waves = (int64_t)gridDim.x * (kPBlock / 64);
  uint32_t err = 0;
  uint64_t passed = 0;
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
      POL::eval(P, F, col[u], cv[u], reg, rv, inb, err);
      bool pass = inb && POL::pass(P, F, plan.pred, col[u], cv[u], reg, rv);
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
        uint64_t* dst = PT.rows + (((uint64_t)part[u] * PT.n_producers + producer) * PT.cap_rows + pos[u]) * NW;
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
__global__ __launch_bounds__(kABlock) void k_partition_agg(const DevTable T, const DevPartition PT, const DevRows spill) {
  extern __shared__ __attribute__((aligned(16))) uint64_t lds[];
  __shared__ uint32_t wave_tot[kABlock / 64];
  const uint32_t S = T.block_mask + 1;
  const int NW = (int)PT.n_words;
  uint64_t* lkeys = lds;
  uint64_t* laccs = lds + S;
  uint32_t* pre = (uint32_t*)(lds + (size_t)S * NW);  // [n_producers + 1] exclusive prefix of counts
  const uint32_t p = blockIdx.x;
  const uint64_t slot0 = (uint64_t)p * S;
  const int lane = lane_id();
  for (uint32_t i = threadIdx.x; i < S; i += kABlock) {
    lkeys[i] = T.keys[slot0 + i];
    for (int a = 0; a < T.na; ++a) laccs[(size_t)a * S + i] = T.accs[(uint64_t)a * T.stride + slot0 + i];
  }
  // exclusive scan of the producer counts (n_producers <= 1024: one per thread)
  const uint32_t NP = PT.n_producers;
  const uint32_t c0 = threadIdx.x < NP ? PT.counts[(uint64_t)p * NP + threadIdx.x] : 0u;
  uint32_t inc = c0;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(inc, d, 64);
    if (lane >= d) inc += o;
  }
  if (lane == 63) wave_tot[threadIdx.x >> 6] = inc;
  __syncthreads();
  uint32_t base = 0, total = 0;
#pragma unroll
  for (int w = 0; w < kABlock / 64; ++w) {
    if (w < (int)(threadIdx.x >> 6)) base += wave_tot[w];
    total += wave_tot[w];
  }
  if (threadIdx.x < NP) pre[threadIdx.x] = base + inc - c0;
  if (threadIdx.x == 0) pre[NP] = total;
  __syncthreads();
  uint32_t new_keys = 0;
  const uint32_t total_pad = (total + 63u) & ~63u;
  for (uint32_t i = threadIdx.x; i < total_pad; i += kABlock) {
    const bool inb = i < total;
    uint64_t key[1];
    uint64_t val[kMaxAggs];
    key[0] = 0;
#pragma unroll
    for (int a = 0; a < kMaxAggs; ++a) val[a] = 0;
    bool todo = inb;
    if (inb) {
      // producer of flattened row i: the counts are near-uniform, so interpolate and correct
      uint32_t lo = (uint32_t)(((uint64_t)i * NP) / total);
      if (lo >= NP) lo = NP - 1;
      while (pre[lo] > i) --lo;
      while (pre[lo + 1] <= i) ++lo;
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
  for (uint32_t i = threadIdx.x; i < S; i += kABlock) {
    T.keys[slot0 + i] = lkeys[i];
    for (int a = 0; a < T.na; ++a) T.accs[(uint64_t)a * T.stride + slot0 + i] = laccs[(size_t)a * S + i];
  }
#pragma unroll
  for (int mm = 32; mm >= 1; mm >>= 1) new_keys += __shfl_xor(new_keys, mm, 64);
  if (lane == 0 && new_keys) atomicAdd(&T.ctrl[CTRL_OCCUPIED], new_keys);
}

size_t partition_stage_bytes(const DevPartition& PT) {
  return (size_t)PT.n_parts * 4 + 16;
}

hipError_t launch_partition(const DevProgram& P, const DevFastPlan& fast, const DevColumns& C, const DevAggPlan& plan,
                            const DevTable& T, const DevPartition& PT, const DevRows& spill, int64_t n,
                            double algo_bytes, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  Scope sc(KID_PARTITION, s, algo_bytes);
  const size_t lds_bytes = partition_stage_bytes(PT);
  const int grid = (int)PT.n_producers;  // every producer writes its counts, even with no rows
  if (lds_bytes > 65536) return hipErrorInvalidValue;
#define DFX_PT(POL) hipLaunchKernelGGL((k_partition<POL>), dim3(grid), dim3(kPBlock), lds_bytes, s, P, fast, C, plan, T, PT, spill, n)
  if (sig_matches<SigKeySumPred2F64>(P, fast, 1, T.na, T.acc_kind, T.val_xform)) {
    DFX_PT(DFX_ARG(StaticPolicy<2, 4, SigKeySumPred2F64>));
    return hipGetLastError();
  }
  if (sig_matches<SigKeySum>(P, fast, 1, T.na, T.acc_kind, T.val_xform)) {
    DFX_PT(DFX_ARG(StaticPolicy<2, 4, SigKeySum>));
    return hipGetLastError();
  }
  const bool use_fast = fast.valid && !P.has_nulls;
  if (P.n_cols <= 2) { if (use_fast) DFX_PT(DFX_ARG(FastPolicy<2, 4>)); else DFX_PT(DFX_ARG(InterpPolicy<2, 2>)); }
  else if (P.n_cols <= 4) { if (use_fast) DFX_PT(DFX_ARG(FastPolicy<4, 2>)); else DFX_PT(DFX_ARG(InterpPolicy<4, 2>)); }
  else { if (use_fast) DFX_PT(DFX_ARG(FastPolicy<8, 2>)); else DFX_PT(DFX_ARG(InterpPolicy<8, 1>)); }
#undef DFX_PT
  return hipGetLastError();
}

hipError_t launch_partition_agg(const DevTable& T, const DevPartition& PT, const DevRows& spill, double algo_bytes,
                                hipStream_t s) {
  Scope sc(KID_PARTITION_AGG, s, algo_bytes);
  const size_t lds_bytes = (size_t)(T.block_mask + 1) * (size_t)(1 + T.na) * 8 + (size_t)(PT.n_producers + 1) * 4 + 16;
  if (lds_bytes > 160 * 1024 - 256 || PT.n_producers > 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_partition_agg, dim3(PT.n_parts), dim3(kABlock), lds_bytes, s, T, PT, spill);
  return hipGetLastError();
}

}  // namespace dfx
