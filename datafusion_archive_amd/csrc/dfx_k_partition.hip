// dfx_k_partition.hip -- partitioned GROUP BY: the pass-1 dispatcher, pass 2 (k_partition_agg) and the sizing helpers.
// The pass-1 kernels and the description of the strategy are in dfx_k_partition_inl.hpp; their instantiations are in
// dfx_k_partition_v0.hip ... _v7.hip.
#define DFX_PARTITION_MAIN_TU
#include "dfx_k_partition_ws_inl.hpp"

namespace dfx {


// pass 2: one workgroup per partition (= table block).  The rows of all producers are visited as
// ONE flattened index space (prefix sums of the per-producer counts live in LDS), so all lanes stay
// busy however small the individual regions are.  Global loads are issued RU rows ahead of the LDS
// probing (the kernel is latency-bound otherwise: 16 waves, one dependent load each).
template <int NA1>  // NA1 == 1: one aggregate (16-byte rows); 0: any
__global__ __launch_bounds__(kABlock) void k_partition_agg(const DevTable T, const DevPartition PT, const DevRows spill) {
  extern __shared__ __attribute__((aligned(16))) uint64_t lds[];
  __shared__ uint32_t wave_tot[kABlock / 64];
  constexpr int RU = NA1 ? 4 : 2;
  constexpr int NV = NA1 ? 1 : kMaxAggs;
  const uint32_t S = T.block_mask + 1;
  const int NW = (int)PT.n_words;
  uint64_t* lkeys = lds;
  uint64_t* laccs = lds + S;
  uint32_t* pre = (uint32_t*)(lds + (size_t)S * NW);  // [n_producers + 1] exclusive prefix of counts
  const uint32_t p = blockIdx.x;
  const uint64_t slot0 = (uint64_t)p * S;
  const int lane = lane_id();
#ifdef DFX_PA_TIMING
  const long long tt0 = wall_clock64();
#endif
  // block -> LDS, 16-byte loads, all loads of a plane in flight before the LDS writes (S is a power of two >= 2)
  for (int w = 0; w < NW; ++w) {
    const uint64_t* src = (w == 0 ? T.keys : T.accs + (uint64_t)(w - 1) * T.stride) + slot0;
    uint64_t* dst = lds + (size_t)w * S;
    for (uint32_t i0 = threadIdx.x * 2; i0 < S; i0 += kABlock * 2 * 4) {
      const uint32_t ia = i0, ib = i0 + kABlock * 2, ic = i0 + kABlock * 4, id = i0 + kABlock * 6;
      const ulonglong2 ta = *(const ulonglong2*)(src + (ia < S ? ia : 0));  // clamped: all four loads in flight together
      const ulonglong2 tb = *(const ulonglong2*)(src + (ib < S ? ib : 0));
      const ulonglong2 tc = *(const ulonglong2*)(src + (ic < S ? ic : 0));
      const ulonglong2 td = *(const ulonglong2*)(src + (id < S ? id : 0));
      if (ia < S) *(ulonglong2*)(dst + ia) = ta;
      if (ib < S) *(ulonglong2*)(dst + ib) = tb;
      if (ic < S) *(ulonglong2*)(dst + ic) = tc;
      if (id < S) *(ulonglong2*)(dst + id) = td;
    }
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
#ifdef DFX_PA_TIMING
  const long long tt1 = wall_clock64();
#endif
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
  // software pipeline: the rows of trip t + 1 are loaded while trip t probes the LDS block
  const float inv = total ? (float)NP / (float)total : 0.f;
  uint64_t key[RU][1], nkey[RU][1];
  uint64_t val[RU][NV], nval[RU][NV];
  bool inb[RU], ninb[RU];
  auto fetch = [&](uint32_t i0, uint64_t (&k)[RU][1], uint64_t (&v)[RU][NV], bool (&ib)[RU]) {
#pragma unroll
    for (int r = 0; r < RU; ++r) {
      const uint32_t i = i0 + (uint32_t)r * kABlock + threadIdx.x;
      ib[r] = i < total;
      k[r][0] = 0;
#pragma unroll
      for (int a = 0; a < NV; ++a) v[r][a] = 0;
      if (NA1) {
        // unconditional load (idle lanes re-read the partition's first row): a branch around it makes the compiler
        // wait with vmcnt(0) -- a full memory round trip per trip -- instead of counting the loads in flight
        const uint64_t* src = PT.rows + (uint64_t)p * PT.part_stride;
        if (ib[r]) {
          // producer of flattened row i: the counts are near-uniform, so interpolate and correct
          uint32_t lo = (uint32_t)((float)i * inv);
          if (lo >= NP) lo = NP - 1;
          while (pre[lo] > i) --lo;
          while (pre[lo + 1] <= i) ++lo;
          src = region_row(PT, p, lo, i - pre[lo]);
        }
        typedef uint64_t u64x2_t __attribute__((ext_vector_type(2)));
        const u64x2_t kv = __builtin_nontemporal_load((const u64x2_t*)src);
        ib[r] = ib[r] && kv.x != kEmptyKey;  // padding rows (pass 1 rounds every region up to whole chunks)
        k[r][0] = ib[r] ? kv.x : 0ull;
        v[r][0] = ib[r] ? kv.y : 0ull;
      } else if (ib[r]) {
        // producer of flattened row i: the counts are near-uniform, so interpolate and correct
        uint32_t lo = (uint32_t)((float)i * inv);
        if (lo >= NP) lo = NP - 1;
        while (pre[lo] > i) --lo;
        while (pre[lo + 1] <= i) ++lo;
        const uint64_t* src = region_row(PT, p, lo, i - pre[lo]);
        {
          k[r][0] = src[0];
#pragma unroll
          for (int a = 0; a < NV; ++a)
            if (a < T.na) v[r][a] = src[1 + a];
          if (k[r][0] == kEmptyKey) ib[r] = false;  // padding row
        }
      }
    }
  };
#ifdef DFX_PA_TIMING
  const long long tt2 = wall_clock64();
#endif
  fetch(0, nkey, nval, ninb);
  for (uint32_t i0 = 0; i0 < total; i0 += kABlock * RU) {  // wave-uniform trip count
#pragma unroll
    for (int r = 0; r < RU; ++r) {
      key[r][0] = nkey[r][0];
      inb[r] = ninb[r];
#pragma unroll
      for (int a = 0; a < NV; ++a) val[r][a] = nval[r][a];
    }
    if (i0 + kABlock * RU < total) fetch(i0 + kABlock * RU, nkey, nval, ninb);
#pragma unroll
    for (int r = 0; r < RU; ++r) {
      bool todo = inb[r];
      if (inb[r]) {
        const uint64_t h = hash_keys<1>(key[r]);
        const uint32_t slot = (uint32_t)((h >> T.shift) & T.mask) & T.block_mask;
        int found = -1;
        // Linear probing, FOUR slots per step (two 16-byte LDS reads of the aligned group): a wave pays for
        // the longest probe sequence among its 64 lanes.  The home slot is the group base (table_upsert_slot),
        // so the probe ORDER is exactly home, home + 1, ... and the block stays a valid linear-probing table for
        // the global kernels; most lookups end in the first group.
        uint32_t g = slot >> 2;
        uint32_t vm = 0xFu;
        for (uint32_t it = 0; it <= (S >> 2) && found < 0;) {
          const ulonglong2 ka = *(const ulonglong2*)&lkeys[g * 4];
          const ulonglong2 kb = *(const ulonglong2*)&lkeys[g * 4 + 2];
          const uint64_t kk = key[r][0];
          const uint32_t mm = ((ka.x == kk ? 1u : 0u) | (ka.y == kk ? 2u : 0u) | (kb.x == kk ? 4u : 0u) | (kb.y == kk ? 8u : 0u)) & vm;
          const uint32_t em = ((ka.x == kEmptyKey ? 1u : 0u) | (ka.y == kEmptyKey ? 2u : 0u) | (kb.x == kEmptyKey ? 4u : 0u) |
                               (kb.y == kEmptyKey ? 8u : 0u)) & vm;
          if (mm) {
            found = (int)(g * 4 + (uint32_t)__ffs((int)mm) - 1u);
          } else if (em) {
            const uint32_t at = g * 4 + (uint32_t)__ffs((int)em) - 1u;
            const uint64_t old = atomicCAS((unsigned long long*)&lkeys[at], (unsigned long long)kEmptyKey, (unsigned long long)kk);
            if (old == kEmptyKey) {
              found = (int)at;
              ++new_keys;
            } else if (old == kk) {
              found = (int)at;
            }  // else: another key claimed it meanwhile -- look at the same group again
          } else {
            g = (g + 1) & ((S >> 2) - 1);
            vm = 0xFu;
            ++it;
          }
        }
        if (found >= 0) {
#pragma unroll
          for (int a = 0; a < NV; ++a)
            if (a < T.na) acc_atomic(T.acc_kind[a], &laccs[(size_t)a * S + found], val[r][a]);
          todo = false;
        }
      }
      if (__ballot(todo) != 0) {  // block full: grow-and-replay takes the row
        uint64_t sv[kMaxAggs];
#pragma unroll
        for (int a = 0; a < kMaxAggs; ++a) sv[a] = a < NV ? val[r][a < NV ? a : 0] : 0;
        spill_row<1>(T, spill, todo, key[r], sv);
      }
    }
  }
#ifdef DFX_PA_TIMING
  const long long tt3 = wall_clock64();
#endif
  __syncthreads();
#ifdef DFX_PA_TIMING
  const long long tt4 = wall_clock64();
#endif
  for (int w = 0; w < NW; ++w) {
    uint64_t* dst = (w == 0 ? T.keys : T.accs + (uint64_t)(w - 1) * T.stride) + slot0;
    const uint64_t* src = lds + (size_t)w * S;
    for (uint32_t i = threadIdx.x * 2; i < S; i += kABlock * 2) *(ulonglong2*)(dst + i) = *(const ulonglong2*)(src + i);
  }
#pragma unroll
  for (int mm = 32; mm >= 1; mm >>= 1) new_keys += __shfl_xor(new_keys, mm, 64);
  if (lane == 0 && new_keys) atomicAdd(&T.ctrl[CTRL_OCCUPIED], new_keys);
  if (p == 0 && threadIdx.x == 0) __hip_atomic_store(&T.ctrl[CTRL_MAX_FILL], 0u, RLX_AGENT);  // the regions are empty again
  snapshot_ctrl_if_last(T, PT);
#ifdef DFX_PA_TIMING
  if ((threadIdx.x == 0 || threadIdx.x == 1023) && (p == 0 || p == 100 || p == 255))
    printf("PA p=%u t=%u total=%u: load %lld prefix %lld loop %lld barrier %lld store %lld (100MHz ticks)\n", p, threadIdx.x, total,
           tt1 - tt0, tt2 - tt1, tt3 - tt2, tt4 - tt3, wall_clock64() - tt4);
  if (threadIdx.x == 0 && (p % 16) == 0) {  // are the 256 workgroups resident together?
    unsigned xcc = 0, hwid = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    printf("PAWG p=%u start %lld end %lld xcc %u cu %u se %u\n", p, tt0 % 100000000ll, wall_clock64() % 100000000ll, xcc & 15u,
           (hwid >> 8) & 15u, (hwid >> 13) & 7u);
  }
#endif
}

// ---- pass 2, streaming form --------------------------------------------------------------------------------
// What round 2 measured about the kernel above (and two rewrites of it): 4.9 us per million routed rows + ~20 us per
// launch WHATEVER the row width (16 / 12 bytes), the hash (3 multiplies / none), the LDS traffic per probe (32 / 16
// bytes) or the depth of the row prefetch -- 1.3 us per 64-row trip and wave.  The disassembly says why: ~110 vector,
// ~180 scalar and ~45 branch instructions per trip (divergent find-or-claim with its EXEC bookkeeping, a run-time switch
// over the accumulator kind, spill checks, 64-bit address arithmetic).  A wave issues one instruction at a time and a
// SIMD one scalar instruction per cycle across its four waves: the loop is bound by instruction ISSUE, mostly scalar.
//
// k_partition_agg_lean is the same algorithm with a short common path:
//   * a WAVE walks whole regions (wave w owns the regions of producers w, w + 16, ...; their row counts sit in one VGPR,
//     lane j = producer w + 16 j, read back with v_readlane), so a row's address is a scalar base + lane * row bytes;
//   * the accumulator kind is a template parameter;
//   * FAST PATH, branch-free: the row's home group of four slots is read (one 16-byte LDS read of 32-bit tags, or two of
//     64-bit keys), four compares, the matching slot picked with v_cndmask, one LDS atomic under the match mask.  In a
//     table that has seen its keys (every batch but the first) 96 % of the rows end here;
//   * a row whose home group does not hold its key (it lives further along the probe sequence, or is new, or the block is
//     full) is parked in a per-wave LDS queue; up to 64 parked rows at a time take the general find-or-claim (same probe
//     order as the global kernels: the block stays a valid linear-probing table) at full lane utilisation;
//   * the row loads of kPF trips are in flight per wave.  The compiler tracks vector-memory results with ONE in-order
//     counter (vmcnt) and is pessimistic after conditional code that contains memory operations, so the loads are
//     issued by inline assembly into VGPRs v88..v119, which the kernel withholds from the register allocator
//     (amdgpu_num_vgpr(88)): nothing the compiler generates can read or reuse them while a load is in flight, and
//     `s_waitcnt vmcnt(kPF - 1)` + v_mov hands a landed row over.  (A first version let the compiler allocate those
//     registers and tied them through the wait with a "+v" constraint; it copied half of an in-flight row BEFORE the wait.)
// NARROW (PTF_NARROW): 12-byte rows {hash image, operand}; the LDS block holds 32-bit TAGS instead of keys: tag = hash
// image of the slot's key (a bijection for keys below 2^32, see ring_route), kTagEmpty for an empty slot, kTagForeign for
// a slot whose key has no image (>= 2^32, inserted by the general path: occupied, never equal to a row's image).  12
// bytes per slot: 96 KB of LDS instead of 128; claimed tags become keys again at write-back.
// Padding rows (pass 1 rounds every region up to whole chunks: key kEmptyKey / image kTagEmpty) are skipped.
constexpr int kP2RetryRows = 128;     // per-wave queue of rows that missed their home group (narrow rows: 12 bytes each)
constexpr int kP2RetryRowsWide = 96;  // ... 16-byte rows: 128 KB of block + 16 x 96 x 16 B = 152 KB of LDS
constexpr int kSharedMaxAggs = 3;  // PTF_SHARED: 4096 slots x (4-byte tag + 3 accumulators) = 112 KB of LDS
constexpr int kPF = 8;
constexpr int kP2Vgprs = 88;  // v88..v119: kPF x 4 registers of in-flight rows, outside the register allocator's reach
struct Row4 { uint32_t x, y, z, w; };
template <int D, int NARROW>
DEV void p2_issue(uint32_t voff, const void* base) {
  if constexpr (NARROW != 0) {
  if constexpr (D == 0) asm volatile("global_load_dwordx3 v[88:90], %0, %1 nt" ::"v"(voff), "s"(base) : "memory", "v88", "v89", "v90");
  else if constexpr (D == 1) asm volatile("global_load_dwordx3 v[92:94], %0, %1 nt" ::"v"(voff), "s"(base) : "memory", "v92", "v93", "v94");
  else if constexpr (D == 2) asm volatile("global_load_dwordx3 v[96:98], %0, %1 nt" ::"v"(voff), "s"(base) : "memory", "v96", "v97", "v98");
  else if constexpr (D == 3) asm volatile("global_load_dwordx3 v[100:102], %0, %1 nt" ::"v"(voff), "s"(base) : "memory", "v100", "v101", "v102");
  else if constexpr (D == 4) asm volatile("global_load_dwordx3 v[104:106], %0, %1 nt" ::"v"(voff), "s"(base) : "memory", "v104", "v105", "v106");
  else if constexpr (D == 5) asm volatile("global_load_dwordx3 v[108:110], %0, %1 nt" ::"v"(voff), "s"(base) : "memory", "v108", "v109", "v110");
  else if constexpr (D == 6) asm volatile("global_load_dwordx3 v[112:114], %0, %1 nt" ::"v"(voff), "s"(base) : "memory", "v112", "v113", "v114");
  else if constexpr (D == 7) asm volatile("global_load_dwordx3 v[116:118], %0, %1 nt" ::"v"(voff), "s"(base) : "memory", "v116", "v117", "v118");
  } else {
  if constexpr (D == 0) asm volatile("global_load_dwordx4 v[88:91], %0, %1 nt" ::"v"(voff), "s"(base) : "memory", "v88", "v89", "v90", "v91");
  else if constexpr (D == 1) asm volatile("global_load_dwordx4 v[92:95], %0, %1 nt" ::"v"(voff), "s"(base) : "memory", "v92", "v93", "v94", "v95");
  else if constexpr (D == 2) asm volatile("global_load_dwordx4 v[96:99], %0, %1 nt" ::"v"(voff), "s"(base) : "memory", "v96", "v97", "v98", "v99");
  else if constexpr (D == 3) asm volatile("global_load_dwordx4 v[100:103], %0, %1 nt" ::"v"(voff), "s"(base) : "memory", "v100", "v101", "v102", "v103");
  else if constexpr (D == 4) asm volatile("global_load_dwordx4 v[104:107], %0, %1 nt" ::"v"(voff), "s"(base) : "memory", "v104", "v105", "v106", "v107");
  else if constexpr (D == 5) asm volatile("global_load_dwordx4 v[108:111], %0, %1 nt" ::"v"(voff), "s"(base) : "memory", "v108", "v109", "v110", "v111");
  else if constexpr (D == 6) asm volatile("global_load_dwordx4 v[112:115], %0, %1 nt" ::"v"(voff), "s"(base) : "memory", "v112", "v113", "v114", "v115");
  else if constexpr (D == 7) asm volatile("global_load_dwordx4 v[116:119], %0, %1 nt" ::"v"(voff), "s"(base) : "memory", "v116", "v117", "v118", "v119");
  }
}
template <int D, int NARROW, int N>
DEV void p2_take(Row4& r) {  // waits until at most N younger loads are in flight, then copies slot D out
  if constexpr (NARROW != 0) {
    r.w = 0;
  if constexpr (D == 0) asm volatile("s_waitcnt vmcnt(%3)\n\tv_mov_b32 %0, v88\n\tv_mov_b32 %1, v89\n\tv_mov_b32 %2, v90" : "=v"(r.x), "=v"(r.y), "=v"(r.z) : "n"(N) : "memory");
  else if constexpr (D == 1) asm volatile("s_waitcnt vmcnt(%3)\n\tv_mov_b32 %0, v92\n\tv_mov_b32 %1, v93\n\tv_mov_b32 %2, v94" : "=v"(r.x), "=v"(r.y), "=v"(r.z) : "n"(N) : "memory");
  else if constexpr (D == 2) asm volatile("s_waitcnt vmcnt(%3)\n\tv_mov_b32 %0, v96\n\tv_mov_b32 %1, v97\n\tv_mov_b32 %2, v98" : "=v"(r.x), "=v"(r.y), "=v"(r.z) : "n"(N) : "memory");
  else if constexpr (D == 3) asm volatile("s_waitcnt vmcnt(%3)\n\tv_mov_b32 %0, v100\n\tv_mov_b32 %1, v101\n\tv_mov_b32 %2, v102" : "=v"(r.x), "=v"(r.y), "=v"(r.z) : "n"(N) : "memory");
  else if constexpr (D == 4) asm volatile("s_waitcnt vmcnt(%3)\n\tv_mov_b32 %0, v104\n\tv_mov_b32 %1, v105\n\tv_mov_b32 %2, v106" : "=v"(r.x), "=v"(r.y), "=v"(r.z) : "n"(N) : "memory");
  else if constexpr (D == 5) asm volatile("s_waitcnt vmcnt(%3)\n\tv_mov_b32 %0, v108\n\tv_mov_b32 %1, v109\n\tv_mov_b32 %2, v110" : "=v"(r.x), "=v"(r.y), "=v"(r.z) : "n"(N) : "memory");
  else if constexpr (D == 6) asm volatile("s_waitcnt vmcnt(%3)\n\tv_mov_b32 %0, v112\n\tv_mov_b32 %1, v113\n\tv_mov_b32 %2, v114" : "=v"(r.x), "=v"(r.y), "=v"(r.z) : "n"(N) : "memory");
  else if constexpr (D == 7) asm volatile("s_waitcnt vmcnt(%3)\n\tv_mov_b32 %0, v116\n\tv_mov_b32 %1, v117\n\tv_mov_b32 %2, v118" : "=v"(r.x), "=v"(r.y), "=v"(r.z) : "n"(N) : "memory");
  } else {
  if constexpr (D == 0) asm volatile("s_waitcnt vmcnt(%4)\n\tv_mov_b32 %0, v88\n\tv_mov_b32 %1, v89\n\tv_mov_b32 %2, v90\n\tv_mov_b32 %3, v91" : "=v"(r.x), "=v"(r.y), "=v"(r.z), "=v"(r.w) : "n"(N) : "memory");
  else if constexpr (D == 1) asm volatile("s_waitcnt vmcnt(%4)\n\tv_mov_b32 %0, v92\n\tv_mov_b32 %1, v93\n\tv_mov_b32 %2, v94\n\tv_mov_b32 %3, v95" : "=v"(r.x), "=v"(r.y), "=v"(r.z), "=v"(r.w) : "n"(N) : "memory");
  else if constexpr (D == 2) asm volatile("s_waitcnt vmcnt(%4)\n\tv_mov_b32 %0, v96\n\tv_mov_b32 %1, v97\n\tv_mov_b32 %2, v98\n\tv_mov_b32 %3, v99" : "=v"(r.x), "=v"(r.y), "=v"(r.z), "=v"(r.w) : "n"(N) : "memory");
  else if constexpr (D == 3) asm volatile("s_waitcnt vmcnt(%4)\n\tv_mov_b32 %0, v100\n\tv_mov_b32 %1, v101\n\tv_mov_b32 %2, v102\n\tv_mov_b32 %3, v103" : "=v"(r.x), "=v"(r.y), "=v"(r.z), "=v"(r.w) : "n"(N) : "memory");
  else if constexpr (D == 4) asm volatile("s_waitcnt vmcnt(%4)\n\tv_mov_b32 %0, v104\n\tv_mov_b32 %1, v105\n\tv_mov_b32 %2, v106\n\tv_mov_b32 %3, v107" : "=v"(r.x), "=v"(r.y), "=v"(r.z), "=v"(r.w) : "n"(N) : "memory");
  else if constexpr (D == 5) asm volatile("s_waitcnt vmcnt(%4)\n\tv_mov_b32 %0, v108\n\tv_mov_b32 %1, v109\n\tv_mov_b32 %2, v110\n\tv_mov_b32 %3, v111" : "=v"(r.x), "=v"(r.y), "=v"(r.z), "=v"(r.w) : "n"(N) : "memory");
  else if constexpr (D == 6) asm volatile("s_waitcnt vmcnt(%4)\n\tv_mov_b32 %0, v112\n\tv_mov_b32 %1, v113\n\tv_mov_b32 %2, v114\n\tv_mov_b32 %3, v115" : "=v"(r.x), "=v"(r.y), "=v"(r.z), "=v"(r.w) : "n"(N) : "memory");
  else if constexpr (D == 7) asm volatile("s_waitcnt vmcnt(%4)\n\tv_mov_b32 %0, v116\n\tv_mov_b32 %1, v117\n\tv_mov_b32 %2, v118\n\tv_mov_b32 %3, v119" : "=v"(r.x), "=v"(r.y), "=v"(r.z), "=v"(r.w) : "n"(N) : "memory");
  }
}
DEV void p2_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

DEV int pa2_lookup(uint64_t* lkeys, uint32_t S, uint32_t slot, uint64_t kk, uint32_t& new_keys) {
  uint32_t g = slot >> 2;
  for (uint32_t it = 0; it <= (S >> 2);) {
    const ulonglong2 ka = *(const ulonglong2*)&lkeys[g * 4];
    const ulonglong2 kb = *(const ulonglong2*)&lkeys[g * 4 + 2];
    const uint32_t mm = (ka.x == kk ? 1u : 0u) | (ka.y == kk ? 2u : 0u) | (kb.x == kk ? 4u : 0u) | (kb.y == kk ? 8u : 0u);
    if (mm) return (int)(g * 4 + (uint32_t)__ffs((int)mm) - 1u);
    const uint32_t em = (ka.x == kEmptyKey ? 1u : 0u) | (ka.y == kEmptyKey ? 2u : 0u) | (kb.x == kEmptyKey ? 4u : 0u) |
                        (kb.y == kEmptyKey ? 8u : 0u);
    if (em) {
      const uint32_t at = g * 4 + (uint32_t)__ffs((int)em) - 1u;
      const uint64_t old = atomicCAS((unsigned long long*)&lkeys[at], (unsigned long long)kEmptyKey, (unsigned long long)kk);
      if (old == kEmptyKey) {
        ++new_keys;
        return (int)at;
      }
      if (old == kk) return (int)at;
      continue;  // another key claimed it meanwhile -- look at the same group again
    }
    g = (g + 1) & ((S >> 2) - 1);
    ++it;
  }
  return -1;
}
DEV int pa2n_lookup(uint32_t* ltags, uint32_t S, uint32_t slot, uint32_t img, uint32_t& new_keys) {
  uint32_t g = slot >> 2;
  for (uint32_t it = 0; it <= (S >> 2);) {
    const uint4 t = *(const uint4*)&ltags[g * 4];
    const uint32_t mm = (t.x == img ? 1u : 0u) | (t.y == img ? 2u : 0u) | (t.z == img ? 4u : 0u) | (t.w == img ? 8u : 0u);
    if (mm) return (int)(g * 4 + (uint32_t)__ffs((int)mm) - 1u);
    const uint32_t em = (t.x == kTagEmpty ? 1u : 0u) | (t.y == kTagEmpty ? 2u : 0u) | (t.z == kTagEmpty ? 4u : 0u) | (t.w == kTagEmpty ? 8u : 0u);
    if (em) {
      const uint32_t at = g * 4 + (uint32_t)__ffs((int)em) - 1u;
      const uint32_t old = atomicCAS(&ltags[at], kTagEmpty, img);
      if (old == kTagEmpty) {
        ++new_keys;
        return (int)at;
      }
      if (old == img) return (int)at;
      continue;  // another image claimed it meanwhile -- look at the same group again
    }
    g = (g + 1) & ((S >> 2) - 1);
    ++it;
  }
  return -1;
}

// branch-free look at ONE aligned group of four slots.  The LDS read and the compare are separate steps so that the
// reads of two rows can be issued together; p2_pin keeps the compiler from sinking a read under the condition its
// result is used in (it would then wait for each read on its own).
struct Group4 {
  uint4 t;           // narrow: four tags
  ulonglong2 ka, kb; // wide: four keys
};
template <int NARROW>
DEV void p2_read_group(const uint32_t* ltags, const uint64_t* lkeys, uint32_t slot4, Group4& g) {
  if (NARROW) {
    g.t = *(const uint4*)&ltags[slot4];
  } else {
    g.ka = *(const ulonglong2*)&lkeys[slot4];
    g.kb = *(const ulonglong2*)&lkeys[slot4 + 2];
  }
}
template <int NARROW>
DEV void p2_pin(Group4& a, Group4& b) {
  if (NARROW) {
    asm volatile("" : "+v"(a.t.x), "+v"(a.t.y), "+v"(a.t.z), "+v"(a.t.w), "+v"(b.t.x), "+v"(b.t.y), "+v"(b.t.z), "+v"(b.t.w));
  } else {
    asm volatile("" : "+v"(a.ka.x), "+v"(a.ka.y), "+v"(a.kb.x), "+v"(a.kb.y), "+v"(b.ka.x), "+v"(b.ka.y), "+v"(b.kb.x), "+v"(b.kb.y));
  }
}
template <int NARROW>
DEV int p2_match(const Group4& g, uint32_t img, uint64_t kk) {
  int idx = -1;
  if (NARROW) {
    idx = g.t.w == img ? 3 : idx;
    idx = g.t.z == img ? 2 : idx;
    idx = g.t.y == img ? 1 : idx;
    idx = g.t.x == img ? 0 : idx;
  } else {
    idx = g.kb.y == kk ? 3 : idx;
    idx = g.kb.x == kk ? 2 : idx;
    idx = g.ka.y == kk ? 1 : idx;
    idx = g.ka.x == kk ? 0 : idx;
  }
  return idx;
}

// The two rows' tag compares of look2 as ONE block (narrow rows).  The compiler's form is v_cmp -> vcc, s_nop 1, v_cndmask, four
// times per row and row after row: a VALU read of a lane mask needs two wait states behind the VALU that wrote it, so eight
// s_nop per pair of trips (5 % of the loop's issue slots).  Here the two rows' compares alternate and go to four scalar pairs,
// every select reads a mask written at least three instructions earlier: sixteen instructions, no wait states.
// idx = the first of the group's four tags that equals the row's image, -1 if none.
#ifndef DFX_P2_ASM_MATCH
#define DFX_P2_ASM_MATCH 1
#endif
DEV void p2_match2_narrow(const uint4& a, uint32_t img0, const uint4& b, uint32_t img1, int& i0, int& i1) {
  uint64_t mA, mB, mC, mD;
  int x0, x1;
  asm volatile(
      "v_cmp_eq_u32_e64 %[mA], %[aw], %[g0]\n\t"
      "v_cmp_eq_u32_e64 %[mB], %[bw], %[g1]\n\t"
      "v_cmp_eq_u32_e64 %[mC], %[az], %[g0]\n\t"
      "v_cmp_eq_u32_e64 %[mD], %[bz], %[g1]\n\t"
      "v_cndmask_b32_e64 %[x0], -1, 3, %[mA]\n\t"
      "v_cndmask_b32_e64 %[x1], -1, 3, %[mB]\n\t"
      "v_cmp_eq_u32_e64 %[mA], %[ay], %[g0]\n\t"
      "v_cmp_eq_u32_e64 %[mB], %[by], %[g1]\n\t"
      "v_cndmask_b32_e64 %[x0], %[x0], 2, %[mC]\n\t"
      "v_cndmask_b32_e64 %[x1], %[x1], 2, %[mD]\n\t"
      "v_cmp_eq_u32_e64 %[mC], %[ax], %[g0]\n\t"
      "v_cmp_eq_u32_e64 %[mD], %[bx], %[g1]\n\t"
      "v_cndmask_b32_e64 %[x0], %[x0], 1, %[mA]\n\t"
      "v_cndmask_b32_e64 %[x1], %[x1], 1, %[mB]\n\t"
      "v_cndmask_b32_e64 %[x0], %[x0], 0, %[mC]\n\t"
      "v_cndmask_b32_e64 %[x1], %[x1], 0, %[mD]"
      : [mA] "=&s"(mA), [mB] "=&s"(mB), [mC] "=&s"(mC), [mD] "=&s"(mD), [x0] "=&v"(x0), [x1] "=&v"(x1)
      : [ax] "v"(a.x), [ay] "v"(a.y), [az] "v"(a.z), [aw] "v"(a.w), [bx] "v"(b.x), [by] "v"(b.y), [bz] "v"(b.z), [bw] "v"(b.w),
        [g0] "v"(img0), [g1] "v"(img1));
  i0 = x0;
  i1 = x1;
}

template <int NARROW, int KIND>
__global__ __launch_bounds__(kABlock) __attribute__((amdgpu_num_vgpr(kP2Vgprs))) void k_partition_agg_lean(const DevTable T, const DevPartition PT,
                                                                                                             const DevRows spill) {
  extern __shared__ __attribute__((aligned(16))) uint64_t lds[];
  constexpr uint32_t kRowBytes = NARROW ? 12u : 16u;
  // KIND < 0 (PTF_SHARED): T.na (2..kSharedMaxAggs) aggregates of ONE operand -- the row carries the raw operand, every
  // aggregate applies its own transform and atomic to it (kinds and transforms are wave-uniform kernel arguments)
  constexpr bool MULTI = KIND < 0;
  static_assert(!MULTI || NARROW != 0, "shared-operand rows are narrow rows");
  const uint32_t NA = MULTI ? (uint32_t)T.na : 1u;
  const uint32_t S = T.block_mask + 1;
  // PTF_PAIR (two aggregates of different operands, 20-byte routed rows): this launch aggregates operand / accumulator plane
  // PT.pair_plane -- the block holds the tags and THAT plane, the row's other operand is not even loaded.  NARROW == 2: the
  // twelve bytes a lane reads are {operand lo, operand hi, image} (plane 0); plane 1 reads {image, operand} like any narrow row.
  // PTF_PLANES (NARROW == 3: aggregates of ONE operand, one launch per accumulator plane): ordinary narrow rows {image, RAW operand};
  // this launch applies plane PT.pair_plane's transform to the operand and aggregates into that plane.
  // NARROW == 4: as 2 with the plane's transform (PTF_PAIR | PTF_PLANES: three and more aggregates over two RAW operands).
  const bool pair = NARROW != 0 && !MULTI && (PT.flags & PTF_PAIR) != 0;  // (20-byte rows)
  constexpr bool PLANES = NARROW == 3 || NARROW == 4;                      // (apply the plane's transform to the operand)
  constexpr bool FMT2 = NARROW == 2 || NARROW == 4;                        // ({operand lo, operand hi, image})
  const bool per_plane = pair || PLANES;
  const uint32_t plane = per_plane ? PT.pair_plane : 0u;
  const uint32_t last_plane = per_plane ? (uint32_t)T.na - 1u : 0u;
  const uint8_t plane_xf = PLANES ? (uint8_t)PT.plane_xf : (uint8_t)VT_RAW;  // (a scalar from the launcher: no kernel-argument array is indexed at run time)
  uint64_t* const t_accs = T.accs + (uint64_t)plane * T.stride;
  // wide: keys[S] accs[S]; narrow: accs[NA][S] tags[S]
  uint64_t* lkeys = lds;
  uint64_t* laccs = NARROW ? lds : lds + S;
  uint32_t* ltags = (uint32_t*)(lds + (size_t)(NARROW ? NA : 1u) * S);
  auto apply = [&](uint32_t at, uint64_t val) {  // (PLANES: `val` is the RAW operand -- this plane's transform here, so that a row that ends in the spill list still has it raw)
    if constexpr (PLANES) val = transform_value(plane_xf, val, true);
    if constexpr (MULTI) {
#pragma unroll
      for (uint32_t a = 0; a < (uint32_t)kSharedMaxAggs; ++a)
        if (a < NA) acc_atomic(T.acc_kind[a], &laccs[(size_t)a * S + at], transform_value(T.val_xform[a], val, true));
    } else {
      acc_atomic((uint8_t)KIND, &laccs[at], val);
    }
  };
  const uint32_t p = blockIdx.x;
  const uint64_t slot0 = (uint64_t)p * S;
  const int lane = lane_id();
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t NP = PT.n_producers;
  // this wave's regions: producer wave + 16 j lives in lane j
  const uint32_t my_prod = wave + (uint32_t)(kABlock / 64) * (uint32_t)lane;
  const uint32_t v_cnt = my_prod < NP ? PT.counts[(uint64_t)p * NP + my_prod] : 0u;
  // table block -> LDS
  for (uint32_t i0 = threadIdx.x * 2; i0 < S; i0 += kABlock * 2) {
    const ulonglong2 kk = *(const ulonglong2*)(T.keys + slot0 + i0);
    for (uint32_t a = 0; a < NA; ++a)
      *(ulonglong2*)(laccs + (size_t)a * S + i0) = *(const ulonglong2*)(t_accs + (uint64_t)a * T.stride + slot0 + i0);
    if (NARROW) {
      uint32_t tg[2];
      const uint64_t k2[2] = {kk.x, kk.y};
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        uint64_t key1[1] = {k2[j]};
        const uint32_t im = (uint32_t)(hash_keys<1>(key1) >> 32);
        tg[j] = k2[j] == kEmptyKey ? kTagEmpty : (((k2[j] >> 32) != 0 || im >= kTagForeign) ? kTagForeign : im);
      }
      *(uint2*)(ltags + i0) = make_uint2(tg[0], tg[1]);
    } else {
      *(ulonglong2*)(lkeys + i0) = kk;
    }
  }
  p2_drain();  // (the block's loads have landed anyway; from here on vmcnt belongs to the row loads below)
  // wave-uniform cursor: region ordinal, rows left in it, address of the next trip's first row
  const uint8_t* const part_bytes = (const uint8_t*)(PT.rows + (uint64_t)p * PT.part_stride);
  const uint64_t prod_bytes = PT.prod_stride * 8ull;
  const uint64_t win_bytes = PT.win_stride * 8ull;  // from one 64-row trip of a region to the next
  const uint8_t* const safe_ptr = part_bytes + (uint64_t)wave * prod_bytes;  // always a readable trip (>= 64 rows per region)
  const uint32_t n_mine = (NP + (uint32_t)(kABlock / 64) - 1u - wave) / (uint32_t)(kABlock / 64);  // regions of this wave
  uint32_t s_j = 0;
  uint32_t s_rem = (uint32_t)__builtin_amdgcn_readlane((int)v_cnt, 0);
  const uint8_t* s_ptr = safe_ptr;
  // LINE chunks (PTF_CHUNK16 with kNarrowLine, dfx_device.hpp): a trip is six 128-byte lines of ten rows -- lane L < 60 reads row
  // L % 10 of line L / 10 --, else 64 contiguous rows; 768 bytes either way
  const bool line_chunks = NARROW != 0 && kNarrowLine && (PT.flags & PTF_CHUNK16) != 0;
  const uint32_t trip_rows = line_chunks ? (uint32_t)kNarrowTripRows : 64u;  // (PTF_PAIR: ten lines of six rows = 60 as well)
  static_assert(kPairTripRows == kNarrowTripRows || !kNarrowLine, "one trip length for both line geometries");
  const uint32_t pair_off = (pair && PT.pair_operand != 0u) ? 8u : 0u;  // operand 1's twelve bytes start at the image
  const uint32_t voff = pair ? ((uint32_t)lane / (uint32_t)kPairChunkRows) * 128u + ((uint32_t)lane % (uint32_t)kPairChunkRows) * (4u * kPairRowDwords) + pair_off
                        : line_chunks ? ((uint32_t)lane / 10u) * 128u + ((uint32_t)lane % 10u) * 12u
                                      : (uint32_t)lane * kRowBytes;  // (lanes past the trip's rows re-read its first row: measured 11 % over-fetch otherwise)
  uint32_t take[kPF];
  uint64_t s_win = win_bytes;  // (0 once the wave's regions are exhausted: the remaining loads of the pipeline re-read safe_ptr)
  auto advance = [&]() -> uint32_t {  // rows of the next trip (0: exhausted); leaves its address in s_ptr
    if (s_rem == 0) {  // the next region that holds rows -- off the common path
      while (s_rem == 0 && s_j < n_mine) {
        ++s_j;
        if (s_j < n_mine) {
          s_rem = (uint32_t)__builtin_amdgcn_readlane((int)v_cnt, (int)(s_j & 63u));
          s_ptr = part_bytes + (uint64_t)(wave + (uint32_t)(kABlock / 64) * s_j) * prod_bytes;
        }
      }
      if (s_rem == 0) {
        s_ptr = safe_ptr;
        s_win = 0;
      }
    }
    return s_rem < trip_rows ? s_rem : trip_rows;
  };
#define DFX_P2_FETCH(D)                                                   \
  {                                                                       \
    const uint32_t tk = advance();                                        \
    take[D] = tk;                                                         \
    p2_issue<D, NARROW>((uint32_t)lane < tk ? voff : pair_off, (const void*)s_ptr);                         \
    s_ptr += s_win;                                                       \
    s_rem -= tk;                                                          \
  }
  DFX_P2_FETCH(0) DFX_P2_FETCH(1) DFX_P2_FETCH(2) DFX_P2_FETCH(3) DFX_P2_FETCH(4) DFX_P2_FETCH(5) DFX_P2_FETCH(6) DFX_P2_FETCH(7)
  static_assert(kPF == 8, "eight row slots");
  __syncthreads();  // the block is in LDS
  uint32_t new_keys = 0;
  const int tag_shift = T.shift - 32;                 // narrow: slot = image >> tag_shift (the image is the hash's high half)
  const uint32_t mask4 = T.block_mask & ~3u;          // home slot = base of the aligned group of four
  bool more = true;
  // TWO trips are probed together: their home-group reads (and, when needed, their next-group reads) are in flight at
  // the same time, so a wave exposes one LDS round trip per pair of trips instead of one per trip (with 4 waves per SIMD
  // the probe loop was bound by that latency: 3.9 us per million rows whatever the instruction count).
  struct Probe {
    uint64_t val, kk;
    uint32_t home4, img;
    int j;  // after look2: the row's slot within its home group (0..3), -1: a row whose key is not there (parked), 4: not a row
    bool real;
  };
  auto decode = [&](const Row4& r, uint32_t tk, Probe& q) {
    const bool inb = (uint32_t)lane < tk;
    if (FMT2) {  // {operand lo, operand hi, image}
      q.val = ((uint64_t)r.y << 32) | r.x;
      q.img = r.z;
      q.kk = 0;
      q.real = inb && r.z != kTagEmpty;
      q.home4 = (uint32_t)(r.z >> tag_shift) & mask4;
    } else if (NARROW) {
      q.val = ((uint64_t)r.z << 32) | r.y;
      q.img = r.x;
      q.kk = 0;
      q.real = inb && r.x != kTagEmpty;
      q.home4 = (uint32_t)(r.x >> tag_shift) & mask4;
    } else {
      q.kk = ((uint64_t)r.y << 32) | r.x;
      q.val = ((uint64_t)r.w << 32) | r.z;
      q.img = 0;
      q.real = inb && q.kk != kEmptyKey;
      uint64_t key1[1] = {q.kk};
      q.home4 = (uint32_t)(hash_keys<1>(key1) >> T.shift) & mask4;
    }
    q.j = 4;
  };
  auto look2 = [&](Probe& q0, uint32_t g0, Probe& q1, uint32_t g1) {  // both rows' groups: two LDS reads in flight, then the compares
    Group4 a, b;
    p2_read_group<NARROW>(ltags, lkeys, g0, a);
    p2_read_group<NARROW>(ltags, lkeys, g1, b);
    int i0, i1;
    if constexpr (NARROW != 0 && DFX_P2_ASM_MATCH != 0) {
      p2_match2_narrow(a.t, q0.img, b.t, q1.img, i0, i1);  // (also pins both reads in front of the compares)
    } else {
      p2_pin<NARROW>(a, b);
      i0 = p2_match<NARROW>(a, q0.img, q0.kk);
      i1 = p2_match<NARROW>(b, q1.img, q1.kk);
    }
    // hit and miss are then ONE compare each on j (a bool that is the AND of two lane masks costs a v_cndmask + v_cmp to ballot)
    q0.j = q0.real ? i0 : 4;
    q1.j = q1.real ? i1 : 4;
  };
  // Rows whose home group does not hold their key (4 % at load 0.5, every row of a block's first batch) are parked in a
  // per-wave LDS queue and go through the general find-or-claim up to 64 at a time.  Handling them in place costs the
  // WHOLE wave a second group look on 99.5 % of the trip pairs (1 - 0.96^128) and the divergent general path on a third
  // of them: more than half of the kernel's vector instructions for 4 % of the rows (pass 2: -20 %).
  constexpr uint32_t kRQ = NARROW ? (uint32_t)kP2RetryRows : (uint32_t)kP2RetryRowsWide;  // rows per wave
  constexpr uint32_t kRW = NARROW ? 3u : 4u;  // words per parked row: {image | key lo, key hi} + {operand lo, operand hi}
  uint32_t* const rq = (NARROW ? (uint32_t*)(ltags + S) : (uint32_t*)(lds + 2 * (size_t)S)) + (size_t)wave * kRQ * kRW;
  uint32_t rqn = 0;  // queued rows (wave-uniform)
  auto retry = [&](bool active, uint32_t at) {  // one queued row per active lane
    uint32_t img = 0;
    uint64_t kk = 0, val = 0;
    if (active) {
      if (NARROW) {
        img = rq[at * kRW];
      } else {
        kk = ((uint64_t)rq[at * kRW + 1] << 32) | rq[at * kRW];
      }
      val = ((uint64_t)rq[at * kRW + kRW - 1] << 32) | rq[at * kRW + kRW - 2];
    }
    bool todo = active;
    if (active) {
      int found;
      if (NARROW) {
        found = pa2n_lookup(ltags, S, (uint32_t)(img >> tag_shift) & mask4, img, new_keys);
      } else {
        uint64_t key1[1] = {kk};
        found = pa2_lookup(lkeys, S, (uint32_t)(hash_keys<1>(key1) >> T.shift) & mask4, kk, new_keys);
      }
      if (found >= 0) {
        apply((uint32_t)found, val);
        todo = false;
      }
    }
    if (__ballot(todo) != 0 && (!per_plane || PT.plane_spills != 0u)) {  // block full: grow-and-replay takes the row (as a key again)
      // one launch per plane: only the LAST plane of an operand spills -- the block is as full for every plane, so the others fail on
      // exactly these rows -- and it writes every accumulator of its operand (raw operand: each one's transform; the pair of two
      // aggregates: the one value the scan transformed), the other operand's accumulators get their identity (adding it is a no-op)
      uint64_t key[1] = {NARROW ? (uint64_t)unhash_word32(img) : kk};
      uint64_t sv[kMaxAggs];
#pragma unroll
      for (int j = 0; j < kMaxAggs; ++j) {
        if (MULTI) {
          sv[j] = (uint32_t)j < NA ? transform_value(T.val_xform[j], val, true) : 0ull;
        } else if (per_plane) {
          const bool mine = pair ? ((PT.pair_ops >> j) & 1u) == PT.pair_operand : true;  // (planes of a shared operand: every accumulator)
          sv[j] = j >= T.na ? 0ull : !mine ? T.acc_init[j] : PLANES ? transform_value(T.val_xform[j], val, true) : val;
        } else {
          sv[j] = j == 0 ? val : 0ull;
        }
      }
      spill_row<1>(T, spill, todo, key, sv);
    }
  };
  auto park = [&](const Probe& q) {
    const uint64_t m = __ballot(q.j < 0);
    if (m != 0) {
      if (q.j < 0) {
        const uint32_t at = rqn + mbcnt64(m);
        if (NARROW) {
          rq[at * kRW] = q.img;
        } else {
          rq[at * kRW] = (uint32_t)q.kk;
          rq[at * kRW + 1] = (uint32_t)(q.kk >> 32);
        }
        rq[at * kRW + kRW - 2] = (uint32_t)q.val;
        rq[at * kRW + kRW - 1] = (uint32_t)(q.val >> 32);
      }
      rqn += (uint32_t)__popcll(m);
      // at most kRQ - 64 rows may stay queued: the next trip parks up to 64 more
      while (rqn > kRQ - 64u) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const uint32_t take_n = rqn < 64u ? rqn : 64u;
        rqn -= take_n;
        retry((uint32_t)lane < take_n, rqn + (uint32_t)lane);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      }
    }
  };
  auto process2 = [&](const Row4& r0, uint32_t tk0, const Row4& r1, uint32_t tk1) {
    Probe q0, q1;
    decode(r0, tk0, q0);
    decode(r1, tk1, q1);
    look2(q0, q0.home4, q1, q1.home4);
    if ((uint32_t)q0.j < 4u) apply(q0.home4 + (uint32_t)q0.j, q0.val);
    if ((uint32_t)q1.j < 4u) apply(q1.home4 + (uint32_t)q1.j, q1.val);
    park(q0);
    park(q1);
  };
#define DFX_P2_PAIR(D0, D1)                                                                                       \
  if (more) {                                                                                                     \
    const uint32_t tk0 = take[D0], tk1 = take[D1];                                                                \
    if (tk0 == 0) {                                                                                               \
      more = false;                                                                                               \
    } else {                                                                                                      \
      Row4 r0, r1;                                                                                                \
      p2_take<D0, NARROW, kPF - 1>(r0);                                                                           \
      p2_take<D1, NARROW, kPF - 2>(r1);                                                                           \
      DFX_P2_FETCH(D0)                                                                                            \
      DFX_P2_FETCH(D1)                                                                                            \
      process2(r0, tk0, r1, tk1);                                                                                 \
      if (tk1 == 0) more = false;                                                                                 \
    }                                                                                                             \
  }
  while (more) {
    DFX_P2_PAIR(0, 1) DFX_P2_PAIR(2, 3) DFX_P2_PAIR(4, 5) DFX_P2_PAIR(6, 7)
  }
#undef DFX_P2_PAIR
#undef DFX_P2_FETCH
  if (rqn != 0) {  // the wave's last parked rows (at most kRQ - 64 <= 64)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    retry((uint32_t)lane < rqn, (uint32_t)lane);
  }
  p2_drain();  // loads still in flight own v88..v119 until they land
  // The key plane goes back only if this block claimed a slot: after a table's first window every key is there already, and
  // the narrow form's write-back is 8-byte stores into every other slot (load 0.5) -- partial lines, 8 MB per launch at 10^6 groups.
  const bool claimed = __syncthreads_or(new_keys != 0) != 0;
  for (uint32_t i0 = threadIdx.x * 2; i0 < S; i0 += kABlock * 2) {
    for (uint32_t a = 0; a < NA; ++a)
      *(ulonglong2*)(t_accs + (uint64_t)a * T.stride + slot0 + i0) = *(const ulonglong2*)(laccs + (size_t)a * S + i0);
    if (!claimed) continue;
    if (NARROW) {
      const uint2 tg = *(const uint2*)(ltags + i0);
      // a tag that is an image is written back as its key (unchanged for slots that held it before, new for claimed
      // ones); empty and foreign slots keep what the table holds
      if (tg.x < kTagForeign) T.keys[slot0 + i0] = (uint64_t)unhash_word32(tg.x);
      if (tg.y < kTagForeign) T.keys[slot0 + i0 + 1] = (uint64_t)unhash_word32(tg.y);
    } else {
      *(ulonglong2*)(T.keys + slot0 + i0) = *(const ulonglong2*)(lkeys + i0);
    }
  }
#pragma unroll
  for (int mm = 32; mm >= 1; mm >>= 1) new_keys += __shfl_xor(new_keys, mm, 64);
  if (lane == 0 && new_keys) atomicAdd(&T.ctrl[CTRL_OCCUPIED], new_keys);
  if (per_plane && plane != last_plane) return;  // (the last plane's launch reads the same regions: it is the one that ends the window)
  if (p == 0 && threadIdx.x == 0) __hip_atomic_store(&T.ctrl[CTRL_MAX_FILL], 0u, RLX_AGENT);  // the regions are empty again
  snapshot_ctrl_if_last(T, PT);
}

template <int NARROW>
static void launch_agg_lean(const DevTable& T, const DevPartition& PT, const DevRows& spill, size_t lds_bytes, hipStream_t s, int plane = 0) {
#define DFX_LEAN(K) hipLaunchKernelGGL((k_partition_agg_lean<NARROW, K>), dim3(PT.n_parts), dim3(kABlock), lds_bytes, s, T, PT, spill)
  switch (T.acc_kind[plane]) {
    case ACC_ADD_F64: DFX_LEAN(ACC_ADD_F64); break;
    case ACC_ADD_F32: DFX_LEAN(ACC_ADD_F32); break;
    case ACC_ADD_U64: DFX_LEAN(ACC_ADD_U64); break;
    case ACC_MIN_S64: DFX_LEAN(ACC_MIN_S64); break;
    case ACC_MAX_S64: DFX_LEAN(ACC_MAX_S64); break;
    case ACC_MIN_U64: DFX_LEAN(ACC_MIN_U64); break;
    default: DFX_LEAN(ACC_MAX_U64); break;
  }
#undef DFX_LEAN
}

// does the (single-word-key) table hold a key that has no 32-bit image?  Run once, after the calibration slice.
__global__ __launch_bounds__(256) void k_probe_wide_keys(const DevTable T) {
  const uint64_t n = T.mask + 1;
  bool wide = false;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
    const uint64_t k = T.keys[i];
    if (k != kEmptyKey) {
      uint64_t key1[1] = {k};
      wide = wide || (k >> 32) != 0 || (uint32_t)(hash_keys<1>(key1) >> 32) >= kTagForeign;
    }
  }
  if (__ballot(wide) != 0 && lane_id() == 0) __hip_atomic_store(&T.ctrl[CTRL_WIDE_KEYS], 1u, RLX_AGENT);
  if (blockIdx.x == 0 && threadIdx.x == 0 && __hip_atomic_load(&T.ctrl[CTRL_SENTINEL], RLX_AGENT) != 0u)
    __hip_atomic_store(&T.ctrl[CTRL_WIDE_KEYS], 1u, RLX_AGENT);  // the key i64::MIN lives outside the slots
}
hipError_t launch_probe_wide_keys(const DevTable& T, hipStream_t s) {
  if (T.kw != 1) return hipSuccess;
  hipLaunchKernelGGL(k_probe_wide_keys, dim3(1024), dim3(256), 0, s, T);
  return hipGetLastError();
}

size_t partition_stage_bytes(const DevPartition& PT) {
  if ((PT.mode & 15u) == 0) return (size_t)PT.n_parts * 4 + 16;
  if ((PT.mode & 15u) == 2 && (PT.flags & PTF_WS)) return partition_ws_bytes(PT.n_parts, PT.ws_scanners == 4 ? 4 : 8, (PT.flags & PTF_PAIR) ? 2 : 1);
  if ((PT.mode & 15u) == 2)
    return partition_ring_bytes(PT.n_words, PT.n_parts, (PT.flags & PTF_CHUNK16) ? kNarrowRingRows : (PT.mode & 0x100u) ? 8 : 16, (PT.flags & PTF_HOT) != 0,
                                (PT.flags & PTF_NARROW) != 0, (PT.flags & PTF_SHARED) ? 128 : 0);
  return (size_t)PT.stage_rows * ((size_t)PT.n_words * 8 + 4) + (size_t)PT.n_parts * 12 + (2 + 16) * 4 + 16;
}

// LDS rows of the write-combining buffer for a workgroup of `block` lanes and an LDS budget
uint32_t partition_sort_capacity(uint32_t n_words, uint32_t n_parts, uint32_t block, size_t lds_budget) {
  const size_t fixed = (size_t)n_parts * 12 + (2 + 16) * 4 + 16;
  if (lds_budget <= fixed) return 0;
  size_t cap = (lds_budget - fixed) / ((size_t)n_words * 8 + 4);
  cap = cap / block * block;
  if (cap > (size_t)kSortMaxCap) cap = kSortMaxCap;
  return (uint32_t)cap;
}

// pass-1 variants (dfx_k_partition_v*.hip)
void launch_partition_variant0(DFX_PARTITION_VARIANT_ARGS);
void launch_partition_variant1(DFX_PARTITION_VARIANT_ARGS);
void launch_partition_variant2(DFX_PARTITION_VARIANT_ARGS);
void launch_partition_variant3(DFX_PARTITION_VARIANT_ARGS);
void launch_partition_variant4(DFX_PARTITION_VARIANT_ARGS);
void launch_partition_variant5(DFX_PARTITION_VARIANT_ARGS);
void launch_partition_variant6(DFX_PARTITION_VARIANT_ARGS);
void launch_partition_variant7(DFX_PARTITION_VARIANT_ARGS);
void launch_partition_variant8(DFX_PARTITION_VARIANT_ARGS);
void launch_partition_variant9(DFX_PARTITION_VARIANT_ARGS);   // PlanPolicy, <= 2 columns, 8-byte null-free
void launch_partition_variant10(DFX_PARTITION_VARIANT_ARGS);  // ... 4-byte columns AND validity bitmaps
void launch_partition_variant11(DFX_PARTITION_VARIANT_ARGS);  // PlanPolicy, <= 4 columns, 8-byte null-free
void launch_partition_variant12(DFX_PARTITION_VARIANT_ARGS);  // ... 4-byte columns AND validity bitmaps
void launch_partition_variant13(DFX_PARTITION_VARIANT_ARGS);  // <= 2 columns, 4-byte columns (no bitmap in the batch)
void launch_partition_variant14(DFX_PARTITION_VARIANT_ARGS);  // <= 2 columns, validity bitmaps (8-byte columns)
void launch_partition_variant15(DFX_PARTITION_VARIANT_ARGS);  // <= 4 columns, 4-byte columns
void launch_partition_variant16(DFX_PARTITION_VARIANT_ARGS);  // <= 4 columns, validity bitmaps
void launch_partition_variant17(DFX_PARTITION_VARIANT_ARGS);  // <= 2 columns, a 4-byte KEY (the only 4-byte column)
void launch_partition_variant18(DFX_PARTITION_VARIANT_ARGS);  // <= 2 columns, a 4-byte key + validity bitmaps
void launch_partition_variant19(DFX_PARTITION_VARIANT_ARGS);  // <= 4 columns, a 4-byte key
void launch_partition_variant20(DFX_PARTITION_VARIANT_ARGS);  // <= 4 columns, a 4-byte key + validity bitmaps
void launch_partition_variant21(DFX_PARTITION_VARIANT_ARGS);  // 3 columns
void launch_partition_variant22(DFX_PARTITION_VARIANT_ARGS);  // 3 columns, 4-byte columns
void launch_partition_variant23(DFX_PARTITION_VARIANT_ARGS);  // 3 columns, validity bitmaps
void launch_partition_variant24(DFX_PARTITION_VARIANT_ARGS);  // 3 columns, both

// The scan plan (DevScanPlan): run-time shapes as data.  Which binding a launch needs follows from the kernel flavour that
// will run: the one-value flavours (narrow rows, the wave-specialised kernel) find the key in slot 0 and the routed value in
// slot 1 (PlanPolicy1), the others look their slots up.
static bool launch_partition_plan(const DevProgram& P, const DevFastPlan& fast, const DevColumns& C, const DevAggPlan& plan,
                                  const DevTable& T, const DevPartition& PT, const DevRows& spill, int64_t n, size_t lds_bytes,
                                  hipStream_t s) {
  const bool shared = (PT.flags & PTF_SHARED) != 0;
  const bool pair = (PT.flags & PTF_PAIR) != 0;  // two operands per routed row: the multi-value binding (slot look-ups) in the wave-specialised kernel
  const bool one_value = !pair && ((PT.flags & PTF_WS) || ((PT.mode & 15u) == 2 && (PT.flags & PTF_NARROW)));
  if (one_value && !shared && T.na != 1) return false;
  if (pair && (T.na < 2 || T.kw != 1 || !(PT.flags & PTF_WS))) return false;
  const uint8_t raw_xf[kMaxAggs] = {VT_RAW};
  DevFastPlan fp;
  DevColumns cp;
  if (!bind_scan_plan(P, fast, C, 1, shared ? 1 : T.na, shared ? raw_xf : T.val_xform, one_value, &fp, &cp)) return false;
  typedef void (*Variant)(DFX_PARTITION_VARIANT_ARGS);
  // [> 2 columns][DevScanPlan::gen: bit 0 4-byte columns, bit 1 validity bitmaps, bit 2 (instead of bit 0) a 4-byte key only]
  static const Variant by_need[3][8] = {
      {launch_partition_variant9, launch_partition_variant13, launch_partition_variant14, launch_partition_variant10,
       launch_partition_variant17, launch_partition_variant13, launch_partition_variant18, launch_partition_variant10},
      {launch_partition_variant21, launch_partition_variant22, launch_partition_variant23, launch_partition_variant24,
       launch_partition_variant19, launch_partition_variant22, launch_partition_variant20, launch_partition_variant24},
      {launch_partition_variant11, launch_partition_variant15, launch_partition_variant16, launch_partition_variant12,
       launch_partition_variant19, launch_partition_variant15, launch_partition_variant20, launch_partition_variant12}};
  if (!one_value && (fp.scan.gen & 4)) return false;  // (cannot happen: bind_scan_plan gives bit 2 to fixed-slot bindings only)
  if (pair && fp.scan.n_cols != 3 && fp.scan.n_cols != 4) return false;  // (the pair kernels: key + two operand columns, and one more predicate column -- variants 21-24, 11 / 15 / 16 / 12)
  DevPartition PTp = PT;
  if (pair) {  // operand 1: the plan slot and the transform of its first accumulator
    const uint32_t a1 = PT.pair_arg1 < (uint32_t)kMaxAggs ? PT.pair_arg1 : 1u;
    PTp.pair_slot1 = fp.scan.argslot[a1];
    PTp.pair_xf1 = T.val_xform[a1];
  }
  by_need[fp.scan.n_cols <= 2 ? 0 : fp.scan.n_cols == 3 ? 1 : 2][fp.scan.gen & 7](P, fp, cp, plan, T, PTp, spill, n, lds_bytes, s);
  return true;
}

// PTF_PAIR: can THIS bound batch go through the pair kernels?  (the plan binds it, three plan columns; host only, no launch)
bool partition_pair_supported(const DevProgram& P, const DevFastPlan& fast, const DevColumns& C, const DevTable& T) {
  if (!kNarrowLine || T.na < 2 || T.kw != 1) return false;
  if (T.na > 2 && P.has_nulls && fast.np == 0) return false;  // (three and more aggregates: the operands travel raw -- see partition_planes_supported)
  DevFastPlan fp;
  DevColumns cp;
  if (!bind_scan_plan(P, fast, C, 1, T.na, T.val_xform, false, &fp, &cp)) return false;
  return (fp.scan.n_cols == 3 || fp.scan.n_cols == 4) && !(fp.scan.gen & 4);
}

// PTF_PLANES: does THIS bound batch take the wave-specialised one-value pass 1 with the aggregates' common raw operand?
// (a compile-time signature of the raw shape, or the scan plan's fixed-slot binding; host only, no launch)
bool partition_planes_supported(const DevProgram& P, const DevFastPlan& fast, const DevColumns& C, const DevTable& T) {
  // (nulls: a raw operand carries no validity.  Under an absorbed predicate nobody asks for it -- every surviving slot is valid,
  // filter.rs:83-92, DevScanPlan::count_valid --; without one COUNT would)
  if (!kNarrowLine || T.na < 2 || T.na > kMaxAggs || T.kw != 1 || (P.has_nulls && fast.np == 0)) return false;
  const uint8_t raw_kind0[1] = {SigKeySumPred2F64::acc(0)}, raw_kind1[1] = {SigKeySum::acc(0)}, raw_xf[kMaxAggs] = {VT_RAW};
  if ((fast.plan_mode & 3) != 2 && (sig_matches<SigKeySumPred2F64>(P, fast, 1, 1, raw_kind0, raw_xf) || sig_matches<SigKeySum>(P, fast, 1, 1, raw_kind1, raw_xf) ||
                                    sig_matches<SigKeyAffSumPred2F64>(P, fast, 1, 1, raw_kind0, raw_xf)))
    return true;
  DevFastPlan fp;
  DevColumns cp;
  return bind_scan_plan(P, fast, C, 1, 1, raw_xf, true, &fp, &cp);
}

hipError_t launch_partition(const DevProgram& P, const DevFastPlan& fast, const DevColumns& C, const DevAggPlan& plan,
                            const DevTable& T, const DevPartition& PT, const DevRows& spill, int64_t n,
                            double algo_bytes, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  Scope sc(KID_PARTITION, s, algo_bytes);
  const size_t lds_bytes = partition_stage_bytes(PT);
  if ((PT.mode & 15u) == 0 && lds_bytes > 65536) return hipErrorInvalidValue;
  if ((PT.mode & 15u) == 2 && lds_bytes > 160 * 1024) return hipErrorInvalidValue;
  if ((PT.mode & 15u) == 1 && (lds_bytes > 160 * 1024 || PT.stage_rows == 0 || PT.stage_rows > (uint32_t)kSortMaxCap ||
                       PT.n_parts > 4096 || (PT.block != 512 && PT.block != 1024)))
    return hipErrorInvalidValue;
  // PTF_SHARED: pass 1 routes the aggregates' common RAW operand -- the shape of a one-aggregate query whose operand is
  // that column (what the accumulators do with it is pass 2's business)
  const bool shared = (PT.flags & PTF_SHARED) != 0;
  const uint8_t raw_kind0[1] = {SigKeySumPred2F64::acc(0)}, raw_kind1[1] = {SigKeySum::acc(0)}, raw_xf[1] = {VT_RAW};
  static_assert(SigKeySumPred2F64::xf(0) == VT_RAW && SigKeySum::xf(0) == VT_RAW, "the one-aggregate signatures route the raw operand");
  if (PT.flags & PTF_PAIR)  // (the host asked partition_pair_supported first: a batch the plan cannot bind is its error to handle)
    return launch_partition_plan(P, fast, C, plan, T, PT, spill, n, lds_bytes, s) ? hipGetLastError() : hipErrorNotSupported;
  if ((fast.plan_mode & 3) == 2 && launch_partition_plan(P, fast, C, plan, T, PT, spill, n, lds_bytes, s)) return hipGetLastError();  // (A/B: scan.plan = 2)
  if (shared ? sig_matches<SigKeySumPred2F64>(P, fast, 1, 1, raw_kind0, raw_xf) : sig_matches<SigKeySumPred2F64>(P, fast, 1, T.na, T.acc_kind, T.val_xform)) {
    launch_partition_variant0(P, fast, C, plan, T, PT, spill, n, lds_bytes, s);
    return hipGetLastError();
  }
  if (shared ? sig_matches<SigKeySum>(P, fast, 1, 1, raw_kind1, raw_xf) : sig_matches<SigKeySum>(P, fast, 1, T.na, T.acc_kind, T.val_xform)) {
    launch_partition_variant1(P, fast, C, plan, T, PT, spill, n, lds_bytes, s);
    return hipGetLastError();
  }
  if (shared ? sig_matches<SigKeyAffSumPred2F64>(P, fast, 1, 1, raw_kind0, raw_xf) : sig_matches<SigKeyAffSumPred2F64>(P, fast, 1, T.na, T.acc_kind, T.val_xform)) {
    launch_partition_variant8(P, fast, C, plan, T, PT, spill, n, lds_bytes, s);
    return hipGetLastError();
  }
  // everything else the plan covers: any single MIN / MAX / COUNT / SUM, one to four terms over Int32 ... Float64 columns,
  // nullable columns -- same kernels, the query is data (the wave-specialised flavour included: its scan loop stays short)
  if (launch_partition_plan(P, fast, C, plan, T, PT, spill, n, lds_bytes, s)) return hipGetLastError();
  if (PT.flags & PTF_PLANES) return hipErrorNotSupported;  // (the host asked partition_planes_supported first; the ring kernels below write another geometry)
  if (fast.plan_mode & 4) return hipErrorNotSupported;  // (the host fused a predicate over nulls counting on a plan)
  const bool use_fast = fast.valid && !P.has_nulls;
  // The wave-specialised flavour pays when the scan loop is short (compile-time signatures).  The run-time decoded shapes and
  // the interpreter spend several times as many instructions per row group: eight scanner waves cannot keep up, sixteen
  // symmetric ones can (measured with PTF_WS on every policy: FastPolicy queries -25 %, the interpreter -28 %).  Same regions,
  // counts and padding either way, so the choice is made per launch.
  DevPartition PTg = PT;
  PTg.flags &= ~PTF_WS;
  const size_t lds_g = partition_stage_bytes(PTg);
  if (P.n_cols <= 2) { if (use_fast) launch_partition_variant2(P, fast, C, plan, T, PTg, spill, n, lds_g, s); else launch_partition_variant3(P, fast, C, plan, T, PTg, spill, n, lds_g, s); }
  else if (P.n_cols <= 4) { if (use_fast) launch_partition_variant4(P, fast, C, plan, T, PTg, spill, n, lds_g, s); else launch_partition_variant5(P, fast, C, plan, T, PTg, spill, n, lds_g, s); }
  else { if (use_fast) launch_partition_variant6(P, fast, C, plan, T, PTg, spill, n, lds_g, s); else launch_partition_variant7(P, fast, C, plan, T, PTg, spill, n, lds_g, s); }
  return hipGetLastError();
}

hipError_t launch_partition_agg(const DevTable& T, const DevPartition& PT, const DevRows& spill, double algo_bytes,
                                hipStream_t s) {
  Scope sc(KID_PARTITION_AGG, s, algo_bytes);
  size_t lds_bytes = (size_t)(T.block_mask + 1) * (size_t)(1 + T.na) * 8 + (size_t)(PT.n_producers + 1) * 4 + 16;
  if (((lds_bytes > 160 * 1024 - 256) && !(PT.flags & (PTF_PAIR | PTF_PLANES))) || PT.n_producers > 1024) return hipErrorInvalidValue;
  if ((PT.flags & PTF_NARROW) && (PT.flags & PTF_SHARED) && !(PT.flags & PTF_PLANES)) {
    const size_t shared_lds = (size_t)(T.block_mask + 1) * (size_t)(4 + 8 * T.na) + (size_t)(kABlock / 64) * kP2RetryRows * 12;
    if (T.na < 2 || T.na > kSharedMaxAggs || T.kw != 1 || shared_lds > 160 * 1024 - 256) return hipErrorInvalidValue;
    hipLaunchKernelGGL((k_partition_agg_lean<1, -1>), dim3(PT.n_parts), dim3(kABlock), shared_lds, s, T, PT, spill);
  } else if ((PT.flags & PTF_NARROW) && (PT.flags & PTF_SHARED) && (PT.flags & PTF_PLANES)) {
    // aggregates of one operand, one launch of the one-value kernel per accumulator plane (the plane's own transform and atomic)
    if (T.na < 2 || T.na > kMaxAggs || T.kw != 1) return hipErrorInvalidValue;
    const size_t plane_lds = (size_t)(T.block_mask + 1) * 12 + (size_t)(kABlock / 64) * kP2RetryRows * 12;
    for (int a = 0; a < T.na; ++a) {
      DevPartition Pa = PT;
      Pa.pair_plane = (uint32_t)a;
      Pa.plane_xf = T.val_xform[a];
      Pa.plane_spills = a + 1 == T.na ? 1u : 0u;
      if (a + 1 < T.na) Pa.snap_host = nullptr;
      launch_agg_lean<3>(T, Pa, spill, plane_lds, s, a);
    }
  } else if ((PT.flags & PTF_NARROW) && (PT.flags & PTF_PAIR)) {
    // two launches of the one-value kernel over the same regions, one per operand / accumulator plane; the first claims the
    // window's new keys, the second finds them, resets CTRL_MAX_FILL and publishes the control block
    if (T.na < 2 || T.kw != 1 || !kNarrowLine) return hipErrorInvalidValue;
    const size_t pair_lds = (size_t)(T.block_mask + 1) * 12 + (size_t)(kABlock / 64) * kP2RetryRows * 12;
    const bool raw_ops = (PT.flags & PTF_PLANES) != 0;  // (three and more aggregates: the launch applies the accumulator's transform)
    for (int a = 0; a < T.na; ++a) {
      DevPartition Pa = PT;
      Pa.pair_plane = (uint32_t)a;
      Pa.pair_operand = (PT.pair_ops >> a) & 1u;
      Pa.plane_xf = T.val_xform[a];
      Pa.plane_spills = 1u;  // the last accumulator of this operand?
      for (int b = a + 1; b < T.na; ++b)
        if (((PT.pair_ops >> b) & 1u) == Pa.pair_operand) Pa.plane_spills = 0u;
      if (a + 1 < T.na) Pa.snap_host = nullptr;
      if (Pa.pair_operand == 0u) {
        if (raw_ops) launch_agg_lean<4>(T, Pa, spill, pair_lds, s, a); else launch_agg_lean<2>(T, Pa, spill, pair_lds, s, a);
      } else {
        if (raw_ops) launch_agg_lean<3>(T, Pa, spill, pair_lds, s, a); else launch_agg_lean<1>(T, Pa, spill, pair_lds, s, a);
      }
    }
  } else if (PT.flags & PTF_NARROW) {
    if (T.na != 1 || T.kw != 1) return hipErrorInvalidValue;
    launch_agg_lean<1>(T, PT, spill, (size_t)(T.block_mask + 1) * 12 + (size_t)(kABlock / 64) * kP2RetryRows * 12, s);
  } else if (T.na == 1 && (PT.flags & PTF_STREAM_PASS2) && PT.n_words == 2)
    launch_agg_lean<0>(T, PT, spill, (size_t)(T.block_mask + 1) * 16 + (size_t)(kABlock / 64) * kP2RetryRowsWide * 16, s);
  else if (T.na == 1) hipLaunchKernelGGL(k_partition_agg<1>, dim3(PT.n_parts), dim3(kABlock), lds_bytes, s, T, PT, spill);
  else hipLaunchKernelGGL(k_partition_agg<0>, dim3(PT.n_parts), dim3(kABlock), lds_bytes, s, T, PT, spill);
  return hipGetLastError();
}

}  // namespace dfx
