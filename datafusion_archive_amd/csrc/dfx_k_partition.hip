// dfx_k_partition.hip -- partitioned GROUP BY: the pass-1 dispatcher, pass 2 (k_partition_agg) and the sizing helpers.
// The pass-1 kernels and the description of the strategy are in dfx_k_partition_inl.hpp; their instantiations are in
// dfx_k_partition_v0.hip ... _v7.hip.
#define DFX_PARTITION_MAIN_TU
#include <type_traits>

#include "dfx_k_partition_inl.hpp"

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
// The kernel above finds the producer region of every flattened row index per lane (float interpolation + LDS prefix
// reads + 64-bit address arithmetic) and keeps ONE trip of row loads in flight.  Measured (round 2): 4.9 us per million
// routed rows + ~20 us per launch whatever the row width, the hash or the LDS traffic -- 1.3 us per 64-row trip and
// wave, i.e. one HBM round trip per trip: the kernel is LATENCY bound, its software pipeline does not pipeline.  (The
// compiler counts vector-memory operations in ONE in-order counter, vmcnt; the rare spill path's stores and atomics sit
// in conditional code inside the loop, and after such a join the wait it inserts for the row load is pessimistic.)
//
// Here a WAVE walks whole regions -- wave w owns the regions of producers w, w + 16, ... (their row counts sit in one
// VGPR, lane j = producer w + 16 j, read back with v_readlane), so a row's address is a scalar base plus lane * row
// bytes -- and the row loads are issued by inline assembly with explicit `s_waitcnt vmcnt(kPF - 1)`: kPF trips (kPF KB
// per wave, 16 waves per CU) are in flight while a trip is probed, whatever else the loop body contains.  (The compiler
// does not know these are loads: every register they write is passed through the wait that covers it before it is read,
// and through a final vmcnt(0) before it dies.)
// NARROW (PTF_NARROW): 12-byte rows {hash image, operand}.  The LDS copy of the table block then holds a plane of 32-bit
// TAGS instead of 64-bit keys: tag = hash image of the slot's key (a bijection for keys below 2^32, see ring_route),
// kTagEmpty for an empty slot, kTagForeign for a slot whose key has no image (>= 2^32, inserted by the general path:
// occupied, never equal to a row's image).  One 16-byte LDS read shows a whole 4-slot group, the compare is four 32-bit
// compares, a claim is a 32-bit LDS CAS, the row's slot is its image's top bits (no re-hash); 12 bytes per slot: 96 KB
// of LDS instead of 128.  Claimed tags become keys again at write-back.
// Padding rows (pass 1 rounds every region up to whole chunks: key kEmptyKey / image kTagEmpty) are skipped.
constexpr int kPF = 8;
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x3_t __attribute__((ext_vector_type(3)));
DEV void row_load_issue(u32x4_t& r, const void* p) { asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(r) : "v"(p) : "memory"); }
DEV void row_load_issue(u32x3_t& r, const void* p) { asm volatile("global_load_dwordx3 %0, %1, off nt" : "=v"(r) : "v"(p) : "memory"); }
template <int N, typename R>
DEV void row_load_wait(R& r) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(r) : "n"(N) : "memory"); }

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

template <int NARROW>
__global__ __launch_bounds__(kABlock) void k_partition_agg_pipe(const DevTable T, const DevPartition PT, const DevRows spill) {
  extern __shared__ __attribute__((aligned(16))) uint64_t lds[];
  typedef typename std::conditional<NARROW != 0, u32x3_t, u32x4_t>::type ROW;
  constexpr uint32_t kRowDwords = NARROW ? 3u : 4u;
  const uint32_t S = T.block_mask + 1;
  // wide: keys[S] accs[S]; narrow: accs[S] tags[S]
  uint64_t* lkeys = lds;
  uint64_t* laccs = NARROW ? lds : lds + S;
  uint32_t* ltags = (uint32_t*)(lds + S);
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
    const ulonglong2 aa = *(const ulonglong2*)(T.accs + slot0 + i0);
    *(ulonglong2*)(laccs + i0) = aa;
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
  const uint64_t* const part_rows = PT.rows + (uint64_t)p * PT.part_stride;
  // wave-uniform cursor over (region ordinal j, row offset i0)
  uint32_t s_j = 0, s_i0 = 0;
  uint32_t s_cnt = (uint32_t)__builtin_amdgcn_readlane((int)v_cnt, 0);
  const uint32_t n_mine = (NP + (uint32_t)(kABlock / 64) - 1u - wave) / (uint32_t)(kABlock / 64);  // regions of this wave
  auto fetch = [&](ROW& r, bool& act) {
    while (s_i0 >= s_cnt && s_j < n_mine) {  // scalar loop: next non-empty region
      ++s_j;
      s_i0 = 0;
      s_cnt = s_j < n_mine ? (uint32_t)__builtin_amdgcn_readlane((int)v_cnt, (int)(s_j & 63u)) : 0u;
    }
    const bool live = s_j < n_mine;
    act = live && s_i0 + (uint32_t)lane < s_cnt;
    const uint32_t* base = (const uint32_t*)(part_rows + (uint64_t)(wave + (uint32_t)(kABlock / 64) * (live ? s_j : 0u)) * PT.prod_stride) +
                           (uint64_t)(live ? s_i0 : 0u) * kRowDwords;
    row_load_issue(r, base + (act ? (uint32_t)lane * kRowDwords : 0u));  // idle lanes re-read the region's first row
    s_i0 += 64;
  };
  ROW rows[kPF];
  bool act[kPF];
#pragma unroll
  for (int d = 0; d < kPF; ++d) fetch(rows[d], act[d]);
  __syncthreads();  // the block is in LDS
  uint32_t new_keys = 0;
  const uint8_t kind = T.acc_kind[0];
  const int tag_shift = T.shift - 32;  // narrow: slot = image >> tag_shift (the image is the hash's high half)
  bool more = true;
  while (more) {
#pragma unroll
    for (int d = 0; d < kPF; ++d) {
      const bool a = act[d];
      if (__ballot(a) == 0) {  // wave-uniform: the cursor is exhausted (trips are handed out in order)
        more = false;
        break;
      }
      row_load_wait<kPF - 1>(rows[d]);  // the oldest of the kPF loads in flight has landed
      const ROW cur = rows[d];
      fetch(rows[d], act[d]);
      uint64_t key[1], val;
      bool have;
      if (NARROW) {
        have = a && cur[0] != kTagEmpty;
        val = ((uint64_t)cur[2] << 32) | cur[1];
        key[0] = 0;
      } else {
        key[0] = ((uint64_t)cur[1] << 32) | cur[0];
        val = ((uint64_t)cur[NARROW ? 0 : 3] << 32) | cur[2];
        have = a && key[0] != kEmptyKey;
      }
      bool todo = have;
      if (have) {
        int found;
        if (NARROW) {
          const uint32_t slot = (uint32_t)(((uint64_t)cur[0] >> tag_shift) & T.mask) & T.block_mask;
          found = pa2n_lookup(ltags, S, slot, cur[0], new_keys);
        } else {
          const uint64_t h = hash_keys<1>(key);
          const uint32_t slot = (uint32_t)((h >> T.shift) & T.mask) & T.block_mask;
          found = pa2_lookup(lkeys, S, slot, key[0], new_keys);
        }
        if (found >= 0) {
          acc_atomic(kind, &laccs[found], val);
          todo = false;
        }
      }
      if (__ballot(todo) != 0) {  // block full: grow-and-replay takes the row (as a key again)
        if (NARROW) key[0] = (uint64_t)unhash_word32(cur[0]);
        uint64_t sv[kMaxAggs];
#pragma unroll
        for (int q = 0; q < kMaxAggs; ++q) sv[q] = q == 0 ? val : 0ull;
        spill_row<1>(T, spill, todo, key, sv);
      }
    }
  }
#pragma unroll
  for (int d = 0; d < kPF; ++d) row_load_wait<0>(rows[d]);  // loads still in flight own these registers until they land
  __syncthreads();
  for (uint32_t i0 = threadIdx.x * 2; i0 < S; i0 += kABlock * 2) {
    *(ulonglong2*)(T.accs + slot0 + i0) = *(const ulonglong2*)(laccs + i0);
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
  if (p == 0 && threadIdx.x == 0) __hip_atomic_store(&T.ctrl[CTRL_MAX_FILL], 0u, RLX_AGENT);  // the regions are empty again
  snapshot_ctrl_if_last(T, PT);
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
  if ((PT.mode & 15u) == 2) return partition_ring_bytes(PT.n_words, PT.n_parts, 16, (PT.flags & PTF_HOT) != 0, (PT.flags & PTF_NARROW) != 0);
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
  if (sig_matches<SigKeySumPred2F64>(P, fast, 1, T.na, T.acc_kind, T.val_xform)) {
    launch_partition_variant0(P, fast, C, plan, T, PT, spill, n, lds_bytes, s);
    return hipGetLastError();
  }
  if (sig_matches<SigKeySum>(P, fast, 1, T.na, T.acc_kind, T.val_xform)) {
    launch_partition_variant1(P, fast, C, plan, T, PT, spill, n, lds_bytes, s);
    return hipGetLastError();
  }
  const bool use_fast = fast.valid && !P.has_nulls;
  if (P.n_cols <= 2) { if (use_fast) launch_partition_variant2(P, fast, C, plan, T, PT, spill, n, lds_bytes, s); else launch_partition_variant3(P, fast, C, plan, T, PT, spill, n, lds_bytes, s); }
  else if (P.n_cols <= 4) { if (use_fast) launch_partition_variant4(P, fast, C, plan, T, PT, spill, n, lds_bytes, s); else launch_partition_variant5(P, fast, C, plan, T, PT, spill, n, lds_bytes, s); }
  else { if (use_fast) launch_partition_variant6(P, fast, C, plan, T, PT, spill, n, lds_bytes, s); else launch_partition_variant7(P, fast, C, plan, T, PT, spill, n, lds_bytes, s); }
  return hipGetLastError();
}

hipError_t launch_partition_agg(const DevTable& T, const DevPartition& PT, const DevRows& spill, double algo_bytes,
                                hipStream_t s) {
  Scope sc(KID_PARTITION_AGG, s, algo_bytes);
  size_t lds_bytes = (size_t)(T.block_mask + 1) * (size_t)(1 + T.na) * 8 + (size_t)(PT.n_producers + 1) * 4 + 16;
  if (lds_bytes > 160 * 1024 - 256 || PT.n_producers > 1024) return hipErrorInvalidValue;
  if (PT.flags & PTF_NARROW) {
    if (T.na != 1 || T.kw != 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_partition_agg_pipe<1>, dim3(PT.n_parts), dim3(kABlock), (size_t)(T.block_mask + 1) * 12, s, T, PT, spill);
  } else if (T.na == 1 && (PT.flags & PTF_STREAM_PASS2) && PT.n_words == 2)
    hipLaunchKernelGGL(k_partition_agg_pipe<0>, dim3(PT.n_parts), dim3(kABlock), (size_t)(T.block_mask + 1) * 16, s, T, PT, spill);
  else if (T.na == 1) hipLaunchKernelGGL(k_partition_agg<1>, dim3(PT.n_parts), dim3(kABlock), lds_bytes, s, T, PT, spill);
  else hipLaunchKernelGGL(k_partition_agg<0>, dim3(PT.n_parts), dim3(kABlock), lds_bytes, s, T, PT, spill);
  return hipGetLastError();
}

}  // namespace dfx
