// dfx_k_core.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the filter / projection /
// aggregate path.  Memory-bound integer/f64 work: no MFMA.  Design rules used throughout:
//   * one wave owns 64 consecutive rows at a time, so every global access is a fully coalesced
//     512-byte (8 B/lane) request and `__ballot` of a per-row predicate IS the Arrow LSB-first
//     bitmap word of those rows;
//   * expression intermediates live in VGPR register files indexed by wave-uniform indices
//     (s_set_gpr_idx), never in memory; literals come from the kernarg segment by scalar load;
//   * all inter-workgroup state (group table, counters) is touched only with agent-scope atomics:
//     per-XCD L2s are not coherent, atomics are (MI355X_MICROARCH.md, inter-workgroup visibility);
//   * grids are sized to a few resident workgroups per CU on 256 CUs and stride over tiles.
// Compile with -ffp-contract=off (the reference never fuses a*b+c) and -munsafe-fp-atomics
// (hardware global_atomic_add_f64 / ds_add_f64 instead of CAS loops).
#include "dfx_kernels.hpp"

#include <hip/hip_runtime.h>

#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

#include <algorithm>

#include <atomic>

#include "dfx_kernels_inl.hpp"
#include "dfx_launch.hpp"

namespace dfx {

uint32_t host_unhash_word32(uint32_t image) { return unhash_word32(image); }
uint64_t host_hash_keys(const uint64_t* key, int kw) {
  switch (kw) {
    case 1: return hash_keys<1>(key);
    case 2: return hash_keys<2>(key);
    case 3: return hash_keys<3>(key);
    case 4: return hash_keys<4>(key);
    default: return hash_keys<8>(key);
  }
}

// ---------------------------------------------------------------------------------------------
// K1 predicate_mask
// ---------------------------------------------------------------------------------------------
template <typename POL>
__global__ __launch_bounds__(kBlock) void k_predicate_mask(const DevProgram P, const DevFastPlan F, const DevColumns C,
                                                           const uint8_t pred, const int64_t n,
                                                           uint64_t* __restrict__ mask_words,
                                                           uint32_t* __restrict__ tile_counts,
                                                           uint32_t* __restrict__ ctrl) {
  typedef typename POL::COLV COLV;
  constexpr int U = POL::U;
  __shared__ uint32_t wave_cnt[kBlock / 64];
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int64_t n_words = (n + 63) >> 6;
  const int64_t n_tiles = (n + kTileRows - 1) / kTileRows;
  uint32_t err = 0;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    uint32_t cnt = 0;
    for (int i0 = 0; i0 < 16; i0 += U) {
      const int64_t w0 = tile * 64 + wave * 16 + i0;
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
        const int64_t w = w0 + uu;
        const bool inb = w * 64 + lane < n;
        u64x16 reg;
        uint32_t rv = 0;
        POL::eval(P, F, cur, curv, reg, rv, inb, err);
        const bool pass = inb && POL::pass(P, F, pred, cur, curv, reg, rv);
        const uint64_t word = __ballot(pass);
        if (lane == 0 && w < n_words) mask_words[w] = word;
        cnt += (uint32_t)__popcll(word);
      }
    }
    if (tile_counts != nullptr) {
      if (lane == 0) wave_cnt[wave] = cnt;
      __syncthreads();
      if (threadIdx.x == 0) tile_counts[tile] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
      __syncthreads();
    }
  }
  if (err) atomicOr(&ctrl[CTRL_ERROR], err);
}

// ---------------------------------------------------------------------------------------------
// K1 + K4 in one pass: single-pass FilterRelation
// ---------------------------------------------------------------------------------------------
// The two-pass form (k_predicate_mask -> scan of the tile counts -> k_compact) reads a predicate column twice: once to
// evaluate it, once to compact it.  Here a workgroup evaluates a SUPER-TILE of four 4096-row tiles (one tile per wave),
// parks the passing rows' values of up to kFusedOutCols of the predicate's own columns in LDS (wave-private segments:
// position = rows the wave kept so far + rank inside the ballot word), learns where its super-tile starts in the output
// from a LOOK-BACK over the kept counts of the super-tiles before it, and copies its segments out in whole coalesced
// runs.  Row order is preserved (filter.rs:86-90).  The Arrow bitmap and the per-tile exclusive offsets are written as
// well: every OTHER column of the batch is compacted by k_compact from them, reading it once, too.
//
// Look-back, two levels.  Super-tiles are dealt round-robin to a co-resident grid (G ~ 1000 workgroups), which therefore
// runs in lockstep: the G super-tiles of a round publish their counts at about the same time and every one of them needs
// the sum of all earlier ones.  A flat decoupled look-back (64 predecessors per step) walks G/128 windows on average, and
// a status round trip under a streaming load costs microseconds (first version of this kernel, 4096-row units: 12 us of
// look-back per 5 us of streaming; two levels: still ~10 us, three dependent round trips).  So: (a) one status word per
// SUPER-TILE {ready, count} and one per GROUP of 64 consecutive super-tiles {aggregate | inclusive prefix}: a super-tile
// adds up the counts of the super-tiles before it in its own group -- one 64-wide load -- and the prefix before its group
// from the group words -- one more 64-wide load that covers 4096 units; both loads are in flight together; the last
// super-tile of a group publishes the group's aggregate as soon as it has the former, and the group's inclusive prefix
// when it has the latter; (b) 128 KB of column per synchronisation instead of 32; (c) the column loads are
// software-pipelined ACROSS super-tiles: a workgroup's first loads of its next super-tile are in flight while it sits in
// the two barriers and the look-back of the current one.
// LDS: a wave parks at most kFusedStage values per column (a quarter of its tile: selectivities up to ~24 %); a wave that
// keeps more reads its tile's passing rows AGAIN after the look-back (the bitmap words are in LDS) -- dense filters fall
// back to two reads of the column for those tiles, never to a wrong result.
// Forward progress: every workgroup takes its super-tiles in increasing order, publishes a count BEFORE it waits, a group
// aggregate needs counts only, a group prefix needs aggregates and the nearest earlier prefix (group 0 needs none): the
// smallest unpublished word never waits for anything unpublished.  The grid must be co-resident (launch_filter_fused
// sizes it from the occupancy API); the spin is bounded all the same (error bit 8).
constexpr uint64_t kLbAggregate = 1ull << 62, kLbInclusive = 2ull << 62, kLbFlags = 3ull << 62;
constexpr int kLbGroup = 64;            // super-tiles per second-level word
constexpr int kFusedTiles = kBlock / 64;  // tiles per super-tile: one per wave
constexpr int kFusedStage = 1024;       // values a wave can park per column

size_t filter_fused_sync_words(int64_t n) {
  const size_t tiles = (size_t)((n + kTileRows - 1) / kTileRows);
  const size_t units = (tiles + kFusedTiles - 1) / kFusedTiles;
  return 2 + units + (units + kLbGroup - 1) / kLbGroup;
}

DEV uint32_t mbcnt_u64(uint64_t m) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
DEV uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += shfl_xor_u64(v, m);
  return v;
}

// The look-back of one super-tile (wave 0 of its workgroup): publish the super-tile's kept count A, add up what the
// super-tiles before it kept (see above), publish the group words this super-tile owes, leave the exclusive prefix in *s_base.
DEV void filter_lookback(const int64_t unit, const int64_t n_units, const int64_t n_tiles, const uint32_t A, uint64_t* __restrict__ state,
                         uint64_t* __restrict__ gstate, uint64_t* __restrict__ sync, uint64_t* __restrict__ tile_offsets,
                         uint32_t* __restrict__ ctrl, const int lane, uint64_t* s_base, uint32_t& err) {
  if (lane == 0) __hip_atomic_store(&state[unit], kLbAggregate | (uint64_t)A, RLX_AGENT);
  const int64_t g = unit / kLbGroup;
  const int q = (int)(unit % kLbGroup);
  const bool last_of_group = q == kLbGroup - 1 || unit == n_units - 1;
  bool need_w = q > 0, need_g = g > 0, agg_pending = last_of_group && g > 0;
  uint64_t within = 0, before = 0;
  int64_t ghi = g - 1;  // lane l looks at group ghi - l
  uint32_t spins = 0;
  while (need_w || need_g) {
    uint64_t st = 0, gs = 0;
    if (need_w) st = __hip_atomic_load(&state[lane < q ? unit - 1 - lane : unit], RLX_AGENT);
    if (need_g) {
      const int64_t gj = ghi - lane;
      gs = __hip_atomic_load(&gstate[gj >= 0 ? gj : 0], RLX_AGENT);
      if (gj < 0) gs = kLbInclusive;  // before group 0: inclusive prefix 0
    }
    bool moved = false;
    if (need_w) {
      if (__ballot(lane < q && (st & kLbFlags) == 0) == 0) {  // every unit before this one in its group has published
        within = wave_sum_u64(lane < q ? (st & ~kLbFlags) : 0ull);
        need_w = false;
      }
    }
    if (!need_w && agg_pending) {  // the group's aggregate: everybody after this group waits for it
      if (lane == 0) __hip_atomic_store(&gstate[g], kLbAggregate | (within + (uint64_t)A), RLX_AGENT);
      agg_pending = false;
    }
    if (need_g) {
      const uint64_t fl = gs & kLbFlags;
      const uint64_t not_ready = __ballot(fl == 0);
      const uint64_t incl = __ballot(fl == kLbInclusive);
      if (incl != 0) {
        const int pl = __ffsll((unsigned long long)incl) - 1;  // nearest group with an inclusive prefix
        const uint64_t upto = pl == 63 ? ~0ull : ((1ull << (pl + 1)) - 1ull);
        if ((not_ready & upto) == 0) {
          before += wave_sum_u64(((upto >> lane) & 1ull) ? (gs & ~kLbFlags) : 0ull);
          need_g = false;
        }
      } else if (not_ready == 0) {  // 64 aggregates, no prefix among them: add them all, look 64 groups further back
        before += wave_sum_u64(gs & ~kLbFlags);
        ghi -= 64;
        moved = true;
      }
    }
    if (!(need_w || need_g) || moved) continue;
    if (++spins > (1u << 22)) {  // cannot happen (see above); never hang the device
      err |= 8u;
      break;
    }
    __builtin_amdgcn_s_sleep(1);
  }
  const uint64_t base = before + within;
  if (lane == 0) {
    if (last_of_group) __hip_atomic_store(&gstate[g], kLbInclusive | (base + (uint64_t)A), RLX_AGENT);
    *s_base = base;
    if (unit == n_units - 1) {
      tile_offsets[n_tiles] = base + (uint64_t)A;
      sync[1] = base + (uint64_t)A;  // kept rows of the batch
      ctrl[CTRL_PASSED_LO] = (uint32_t)(base + (uint64_t)A);  // ... and next to the error word: ONE read-back per batch
      ctrl[CTRL_PASSED_HI] = (uint32_t)((base + (uint64_t)A) >> 32);
    }
  }
}

// (a compile-time signature fits 128 VGPRs -- four workgroups per CU --, the generic policies take what they need)
template <typename POL>
__global__ __launch_bounds__(kBlock, (POL::kStaticNa == 0 && sizeof(typename POL::COLV) == 16 && POL::U == 8 && POL::kIsStatic) ? 4 : 1) void k_filter_fused(const DevProgram P, const DevFastPlan F, const DevColumns C,
                                                         const uint8_t pred, const int64_t n,
                                                         uint64_t* __restrict__ mask_words,
                                                         uint64_t* __restrict__ tile_offsets,
                                                         uint64_t* __restrict__ sync, const DevFusedOut O,
                                                         uint32_t* __restrict__ ctrl) {
  typedef typename POL::COLV COLV;
  constexpr int U = POL::U;
  constexpr int BANK = (int)(sizeof(COLV) / 8);
  constexpr int NW = kFusedTiles;              // waves per workgroup = tiles per super-tile
  constexpr int kTileWords = kTileRows / 64;   // 64 bitmap words per tile
  static_assert(kLbGroup == 64 && kTileWords == 64, "one lane per unit of a group / per bitmap word of a tile");
  extern __shared__ __attribute__((aligned(16))) uint64_t stage[];  // [O.n][NW][kFusedStage] kept values
  __shared__ uint64_t s_words[NW * kTileWords];
  __shared__ uint32_t s_wave_cnt[NW];
  __shared__ uint64_t s_base;
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int64_t n_words = (n + 63) >> 6;
  const int64_t n_tiles = (n + kTileRows - 1) / kTileRows;
  const int64_t n_units = (n_tiles + NW - 1) / NW;
  uint64_t* const state = sync + 2;          // per super-tile
  uint64_t* const gstate = state + n_units;  // per group of kLbGroup super-tiles
  uint32_t err = 0;
  // the whole loop once per comparison FORM (StaticPolicy::pass_form: compile-time operators; 0: run-time masks)
  auto run = [&](auto form_tag) {
  constexpr int FORM = decltype(form_tag)::value;
  int64_t unit = blockIdx.x;
  COLV ncol[U];
  uint32_t ncv[U];
  load_trip<POL>(P, C, (unit * NW + wave) * kTileWords, unit < n_units, n, lane, ncol, ncv);  // the very first loads of this workgroup
  for (; unit < n_units; unit += gridDim.x) {
    const int64_t tile = unit * NW + wave;  // this wave's tile
    uint32_t cnt = 0;                       // rows this wave has kept in its tile (wave-uniform)
    for (int i0 = 0; i0 < kTileWords; i0 += U) {
      const int64_t w0 = tile * kTileWords + i0;
      COLV col[U];
      uint32_t cv[U];
      FOR_U {
        col[u] = ncol[u];
        cv[u] = ncv[u];
      }
      {  // the next trip's loads (of this tile, or the first ones of the workgroup's next super-tile) before this trip is evaluated
        const bool same = i0 + U < kTileWords;
        const int64_t nu = same ? unit : unit + gridDim.x;
        load_trip<POL>(P, C, (nu * NW + wave) * kTileWords + (same ? i0 + U : 0), nu < n_units, n, lane, ncol, ncv);
      }
      auto one_group = [&](const COLV& cur, uint32_t curv, int uu) {
        const int64_t w = w0 + uu;
        const bool inb = w * 64 + lane < n;
        u64x16 reg;
        uint32_t rv = 0;
        POL::eval(P, F, cur, curv, reg, rv, inb, err);
        const bool pass = inb && POL::template pass_form<FORM>(P, F, pred, cur, curv, reg, rv);
        const uint64_t word = __ballot(pass);
        if (lane == 0) s_words[wave * kTileWords + i0 + uu] = word;
        const uint32_t at = cnt + mbcnt_u64(word);
        if (pass && at < (uint32_t)kFusedStage) {
#pragma unroll
          for (int o = 0; o < kFusedOutCols; ++o) {
            if (o < O.n) {
              uint64_t v = cur[0];
#pragma unroll
              for (int c = 1; c < BANK; ++c) v = (O.slot[o] == c) ? cur[c] : v;
              stage[(size_t)(o * NW + wave) * kFusedStage + at] = v;
            }
          }
        }
        cnt += (uint32_t)__popcll(word);
      };
      if constexpr (POL::kIsStatic) {  // straight-line code is short here: unrolled (no scalar branch chain to pick the bank)
        FOR_U one_group(col[u], cv[u], u);
      } else {  // the interpreter's body is long: ONE copy, the group's bank re-selected at run time
#pragma nounroll
        for (int uu = 0; uu < U; ++uu) {
          COLV cur;
          uint32_t curv;
          DFX_SELECT_BANK(uu, col, cv, cur, curv)
          one_group(cur, curv, uu);
        }
      }
    }
    {  // this tile's 64 bitmap words: one coalesced 512-byte store per wave
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      const int64_t w = tile * kTileWords + lane;
      if (w < n_words) mask_words[w] = s_words[wave * kTileWords + lane];
    }
    if (lane == 0) s_wave_cnt[wave] = cnt;
    __syncthreads();
    uint32_t wc[NW];
    uint32_t A = 0;
#pragma unroll
    for (int k = 0; k < NW; ++k) {
      wc[k] = s_wave_cnt[k];
      A += wc[k];
    }
    if (wave == 0) filter_lookback(unit, n_units, n_tiles, A, state, gstate, sync, tile_offsets, ctrl, lane, &s_base, err);
    __syncthreads();
    uint64_t my_base = s_base;
#pragma unroll
    for (int k = 0; k < NW; ++k)
      if (k < wave) my_base += wc[k];
    if (lane == 0 && tile < n_tiles) tile_offsets[tile] = my_base;
#pragma unroll
    for (int o = 0; o < kFusedOutCols; ++o) {
      if (o < O.n) {
        const uint8_t t = O.dtype[o];
        // (the output buffers hold O.cap_rows rows: what lies beyond is the host's to compact again)
        const uint32_t room = my_base >= O.cap_rows ? 0u : (O.cap_rows - my_base < (uint64_t)cnt ? (uint32_t)(O.cap_rows - my_base) : cnt);
        if (cnt <= (uint32_t)kFusedStage) {  // (wave-uniform) everything this wave kept is parked in LDS
          const uint64_t* src = stage + (size_t)(o * NW + wave) * kFusedStage;
          if (t == T_F64 || t == T_I64 || t == T_U64) {
            uint64_t* dst = (uint64_t*)O.out[o] + my_base;
            for (uint32_t j = (uint32_t)lane; j < room; j += 64) dst[j] = src[j];
          } else {
            for (uint32_t j = (uint32_t)lane; j < room; j += 64) store_typed(t, O.out[o], (int64_t)(my_base + j), src[j]);
          }
        } else {  // a dense tile: its passing rows are read again (bitmap words from LDS), as k_compact would -- the tile was
                  // streamed through this CU microseconds ago, so the second read is an L2 / Infinity Cache hit, not HBM traffic.
                  // Eight row groups per step, their loads issued together: one dependent load -> store round trip per row group
                  // (64 per tile) made selectivities above a quarter cost 2.5 x the selective case (cfg2_filter_dense_sel50: 0.29).
          const int slot = O.slot[o];
          constexpr int kStep = 8;  // (16 was measured: the selective path lost 13 % -- 0.588 -> 0.510 -- to the registers the dense path asked for)
          static_assert(kTileWords % kStep == 0, "whole steps");
          uint32_t run = 0;
          const int64_t n_rows_m1 = n - 1;
          // (the width is decided once per tile, outside the steps: a dtype switch around every load would put a join --
          // and a conservative wait -- between them)
          auto steps = [&](auto load_one, auto store_one) {
            for (int i0 = 0; i0 < kTileWords; i0 += kStep) {
              uint64_t word[kStep], v[kStep];
#pragma unroll
              for (int j = 0; j < kStep; ++j) {
                word[j] = s_words[wave * kTileWords + i0 + j];
                int64_t row = (tile * kTileWords + i0 + j) * 64 + lane;
                row = row < n_rows_m1 ? row : n_rows_m1;  // (unconditional loads: rows past the end are never selected)
                v[j] = load_one(row);
              }
#pragma unroll
              for (int j = 0; j < kStep; ++j) {
                if ((word[j] >> lane) & 1ull) {
                  const uint32_t at = run + mbcnt_u64(word[j]);
                  if (at < room) store_one(my_base + at, v[j]);
                }
                run += (uint32_t)__popcll(word[j]);
              }
            }
          };
          if (t == T_F64 || t == T_I64 || t == T_U64) {
            const uint64_t* in8 = (const uint64_t*)C.c[slot].values;
            uint64_t* out8 = (uint64_t*)O.out[o];
            steps([&](int64_t row) { return in8[row]; }, [&](uint64_t at, uint64_t v) { out8[at] = v; });
          } else if (t == T_I32 || t == T_U32 || t == T_F32) {
            const uint32_t* in4 = (const uint32_t*)C.c[slot].values;
            uint32_t* out4 = (uint32_t*)O.out[o];
            steps([&](int64_t row) { return (uint64_t)in4[row]; }, [&](uint64_t at, uint64_t v) { out4[at] = (uint32_t)v; });
          } else {
            steps([&](int64_t row) { return load_canonical(t, C.c[slot].values, row, C.c[slot].bit_offset); },
                  [&](uint64_t at, uint64_t v) { store_typed(t, O.out[o], (int64_t)at, v); });
          }
        }
      }
    }
    // (the next super-tile's first barrier separates these reads of s_base / s_words / stage from their next writes)
  }
  };
  switch (POL::form_of(F)) {  // (wave-uniform: the plan sits in the kernarg segment)
    case 4 | (1 << 3): run(std::integral_constant<int, (POL::kIsStatic ? (4 | (1 << 3)) : 0)>{}); break;  // x >  a AND x <  b
    case 6 | (1 << 3): run(std::integral_constant<int, (POL::kIsStatic ? (6 | (1 << 3)) : 0)>{}); break;  // x >= a AND x <  b
    case 4 | (3 << 3): run(std::integral_constant<int, (POL::kIsStatic ? (4 | (3 << 3)) : 0)>{}); break;  // x >  a AND x <= b
    case 6 | (3 << 3): run(std::integral_constant<int, (POL::kIsStatic ? (6 | (3 << 3)) : 0)>{}); break;  // x >= a AND x <= b
    default: run(std::integral_constant<int, 0>{}); break;
  }
  if (err) atomicOr(&ctrl[CTRL_ERROR], err);
}

// DENSE flavour (round 4): the tile stays in REGISTERS.  k_filter_fused parks a wave's kept values in LDS -- at most a quarter of
// its tile -- and a wave that keeps more reads its passing rows a second time after the look-back: 8 + 8 sel + 8 sel bytes of
// traffic per row, which is all that selectivities of 0.5 / 0.9 could ever reach (0.40 / 0.44 of the roofline).  Here a wave loads
// its whole 4096-row tile up front -- 64 independent 512-byte loads, 128 vector registers --, evaluates the predicate from the
// registers, keeps bitmap word i in lane i (one coalesced 512-byte store per tile, `v_readlane` hands the words back after the
// look-back) and stores the kept values from the same registers: the column is read ONCE whatever the selectivity, nothing is
// staged in LDS.  Two workgroups per CU (256 registers each): while one waits in its look-back the other streams.
// For the shape config 2 is written in: a compile-time signature over ONE 8-byte column, which is also the one column the kernel
// compacts (launch_filter_fused checks); chosen by the host once a stream has kept more than a wave can park (DevFusedOut::dense).
template <typename POL, int FORM>
__global__ __launch_bounds__(kBlock, 2) void k_filter_fused_dense(const DevProgram P, const DevFastPlan F, const DevColumns C,
                                                                  const uint8_t pred, const int64_t n,
                                                                  uint64_t* __restrict__ mask_words,
                                                                  uint64_t* __restrict__ tile_offsets,
                                                                  uint64_t* __restrict__ sync, const DevFusedOut O,
                                                                  uint32_t* __restrict__ ctrl) {
  typedef typename POL::COLV COLV;
  constexpr int BANK = (int)(sizeof(COLV) / 8);
  constexpr int NW = kFusedTiles;
  constexpr int kTileWords = kTileRows / 64;
  static_assert(POL::kIsStatic && kTileWords == 64, "one lane per bitmap word of a tile; straight-line predicate code");
  __shared__ uint32_t s_wave_cnt[NW];
  __shared__ uint64_t s_base;
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int64_t n_words = (n + 63) >> 6;
  const int64_t n_tiles = (n + kTileRows - 1) / kTileRows;
  const int64_t n_units = (n_tiles + NW - 1) / NW;
  uint64_t* const state = sync + 2;
  uint64_t* const gstate = state + n_units;
  const uint64_t* __restrict__ in = (const uint64_t*)C.c[0].values;
  uint64_t* __restrict__ out = (uint64_t*)O.out[0];
  uint32_t err = 0;
  {
    for (int64_t unit = blockIdx.x; unit < n_units; unit += gridDim.x) {
      const int64_t tile = unit * NW + wave;
      const int64_t row0 = tile * kTileRows;
      const int64_t left = n - row0;  // rows of this tile (<= 0: a wave past the end of the batch: it reads row 0 and keeps nothing)
      const uint32_t last = left <= 0 ? 0u : (uint32_t)((left < (int64_t)kTileRows ? left : (int64_t)kTileRows) - 1);
      const uint64_t* base = in + (left <= 0 ? 0 : row0);
      // (opaque per super-tile: everything derived from the lane number alone -- 64 row indices, 64 lane == i masks -- is loop
      // invariant, and hoisted out of this loop it occupies 128 vector and 128 scalar registers for the kernel's whole life)
      uint32_t lv = (uint32_t)lane;
      asm volatile("" : "+v"(lv));
      const int32_t left_lane = (int32_t)(left >= (int64_t)kTileRows ? (int64_t)kTileRows : (left < 0 ? 0 : left)) - (int32_t)lv;
      uint64_t keep[kTileWords];
      if (left >= (int64_t)kTileRows) {  // (wave-uniform) a whole tile: one lane offset, the row group in the instruction's immediate
        const uint64_t* mine = base + lv;
#pragma unroll
        for (int i = 0; i < kTileWords; ++i) keep[i] = __builtin_nontemporal_load(mine + i * 64);
      } else {  // the batch's last tile (or none of it): unconditional loads of clamped rows, masked below
#pragma unroll
        for (int i = 0; i < kTileWords; ++i) {
          const uint32_t idx = (uint32_t)(i * 64) + lv;
          keep[i] = __builtin_nontemporal_load(base + (idx < last ? idx : last));
          __builtin_amdgcn_sched_barrier(0);  // (one index register at a time: 64 of them at once spill)
        }
      }
      uint32_t cnt = 0;
      uint32_t my_lo = 0, my_hi = 0;  // lane i: bitmap word i of the tile
#pragma unroll
      for (int i = 0; i < kTileWords; ++i) {
        const bool inb = (int32_t)(i * 64) < left_lane;
        COLV cur;
#pragma unroll
        for (int c = 0; c < BANK; ++c) cur[c] = keep[i];
        u64x16 reg;
        uint32_t rv = 0;
        POL::eval(P, F, cur, 0xFFFFFFFFu, reg, rv, inb, err);
        const bool pass = inb && POL::template pass_form<FORM>(P, F, pred, cur, 0xFFFFFFFFu, reg, rv);
        const uint64_t word = __ballot(pass);
        my_lo = lv == (uint32_t)i ? (uint32_t)word : my_lo;
        my_hi = lv == (uint32_t)i ? (uint32_t)(word >> 32) : my_hi;
        cnt += (uint32_t)__popcll(word);
        __builtin_amdgcn_sched_barrier(0);  // (one ballot at a time: the scheduler would keep all 64 SGPR pairs alive)
      }
      {
        const int64_t w = tile * kTileWords + lane;
        if (w < n_words) mask_words[w] = ((uint64_t)my_hi << 32) | my_lo;
      }
      if (lane == 0) s_wave_cnt[wave] = cnt;
      __syncthreads();
      uint32_t wc[NW];
      uint32_t A = 0;
#pragma unroll
      for (int k = 0; k < NW; ++k) {
        wc[k] = s_wave_cnt[k];
        A += wc[k];
      }
      if (wave == 0) filter_lookback(unit, n_units, n_tiles, A, state, gstate, sync, tile_offsets, ctrl, lane, &s_base, err);
      __syncthreads();
      uint64_t my_base = s_base;
#pragma unroll
      for (int k = 0; k < NW; ++k)
        if (k < wave) my_base += wc[k];
      if (lane == 0 && tile < n_tiles) tile_offsets[tile] = my_base;
      // (the output buffer holds O.cap_rows rows: what lies beyond is the host's to compact again)
      const uint32_t room = my_base >= O.cap_rows ? 0u : (O.cap_rows - my_base < (uint64_t)cnt ? (uint32_t)(O.cap_rows - my_base) : cnt);
      uint64_t* dst = out + my_base;
      uint32_t run_at = 0;  // (wave-uniform)
      const uint32_t lo = my_lo, hi = my_hi;
#pragma unroll
      for (int i = 0; i < kTileWords; ++i) {
        const uint64_t word = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)hi, i) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)lo, i);
        const uint32_t at = run_at + mbcnt_u64(word);
        if (((word >> lane) & 1ull) && at < room) dst[at] = keep[i];
        run_at += (uint32_t)__popcll(word);
        __builtin_amdgcn_sched_barrier(0);
      }
      // (the next super-tile's first barrier separates these reads of s_base / s_wave_cnt from their next writes)
    }
  }
  if (err) atomicOr(&ctrl[CTRL_ERROR], err);
}

// ---------------------------------------------------------------------------------------------
// exclusive scans (tile counts -> offsets; string lengths -> offsets)
// ---------------------------------------------------------------------------------------------
constexpr int kScanChunk = 4096;  // elements per block (256 threads x 16)

DEV uint64_t wave_inclusive_scan(uint64_t v) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint64_t o = __shfl_up(v, d, 64);
    if (lane_id() >= d) v += o;
  }
  return v;
}

// block-wide exclusive scan of one value per thread; returns the exclusive prefix, *total = block sum
DEV uint64_t block_exclusive_scan(uint64_t v, uint64_t* total, uint64_t* lds4) {
  const uint64_t inc = wave_inclusive_scan(v);
  const int wave = threadIdx.x >> 6;
  if (lane_id() == 63) lds4[wave] = inc;
  __syncthreads();
  uint64_t base = 0;
  for (int w = 0; w < wave; ++w) base += lds4[w];
  *total = lds4[0] + lds4[1] + lds4[2] + lds4[3];
  __syncthreads();
  return base + inc - v;
}

template <typename TIN>
__global__ __launch_bounds__(kBlock) void k_scan_local(const TIN* __restrict__ in, int64_t n,
                                                       uint64_t* __restrict__ block_sums) {
  __shared__ uint64_t lds4[4];
  const int64_t base = (int64_t)blockIdx.x * kScanChunk + (int64_t)threadIdx.x * 16;
  uint64_t sum = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i)
    if (base + i < n) sum += (uint64_t)in[base + i];
  uint64_t total;
  block_exclusive_scan(sum, &total, lds4);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(kBlock) void k_scan_sums(uint64_t* __restrict__ block_sums, int64_t nb) {
  // single block: exclusive scan of block_sums in place; block_sums[nb] = grand total
  __shared__ uint64_t lds4[4];
  uint64_t carry = 0;
  for (int64_t b0 = 0; b0 < nb; b0 += kBlock) {
    const int64_t i = b0 + threadIdx.x;
    const uint64_t v = i < nb ? block_sums[i] : 0;
    uint64_t total;
    const uint64_t ex = block_exclusive_scan(v, &total, lds4);
    if (i < nb) block_sums[i] = carry + ex;
    carry += total;
  }
  if (threadIdx.x == 0) block_sums[nb] = carry;
}

template <typename TIN, typename TOUT>
__global__ __launch_bounds__(kBlock) void k_scan_apply(const TIN* __restrict__ in, int64_t n,
                                                       const uint64_t* __restrict__ block_sums,
                                                       int64_t nb, TOUT* __restrict__ out) {
  __shared__ uint64_t lds4[4];
  const int64_t base = (int64_t)blockIdx.x * kScanChunk + (int64_t)threadIdx.x * 16;
  uint64_t v[16];
  uint64_t sum = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    v[i] = (base + i < n) ? (uint64_t)in[base + i] : 0;
    sum += v[i];
  }
  uint64_t total;
  uint64_t run = block_sums[blockIdx.x] + block_exclusive_scan(sum, &total, lds4);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    if (base + i < n) out[base + i] = (TOUT)run;
    run += v[i];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = (TOUT)block_sums[nb];
}

// ---------------------------------------------------------------------------------------------
// K4 compact
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void k_compact(const T* __restrict__ in,
                                                    const uint64_t* __restrict__ mask_words,
                                                    const uint64_t* __restrict__ tile_offsets,
                                                    const int64_t n, T* __restrict__ out, const uint64_t out_limit) {
  __shared__ uint32_t word_off[64];
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int64_t n_words = (n + 63) >> 6;
  const int64_t n_tiles = (n + kTileRows - 1) / kTileRows;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    if (wave == 0) {  // popcount of each of the tile's 64 words, exclusive-scanned by one wave
      const int64_t w = tile * 64 + lane;
      const uint32_t c = w < n_words ? (uint32_t)__popcll(mask_words[w]) : 0u;
      const uint64_t inc = wave_inclusive_scan((uint64_t)c);
      word_off[lane] = (uint32_t)(inc - c);
    }
    __syncthreads();
    const uint64_t tile_base = tile_offsets[tile];
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
      const int wi = wave * 16 + i;
      const int64_t w = tile * 64 + wi;
      if (w < n_words) {
        const uint64_t word = mask_words[w];  // wave-uniform address: one request
        const int64_t row = w * 64 + lane;
        if ((word >> lane) & 1) {
          const uint32_t rank = (uint32_t)__popcll(word & ((1ull << lane) - 1ull));
          const uint64_t at = tile_base + word_off[wi] + rank;
          if (at < out_limit) out[at] = in[row];  // (emit sizes `out` from the host's group count before the scan's total is back)
        }
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(kBlock) void k_utf8_lengths(const int32_t* __restrict__ offsets, int64_t n,
                                                         int32_t* __restrict__ lengths,
                                                         int32_t* __restrict__ starts) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const int32_t a = offsets[i], b = offsets[i + 1];
    lengths[i] = b - a;
    starts[i] = a;
  }
}

__global__ __launch_bounds__(kBlock) void k_utf8_gather(const uint8_t* __restrict__ data,
                                                        const int32_t* __restrict__ src_starts,
                                                        const int32_t* __restrict__ dst_offsets,
                                                        int64_t m, uint8_t* __restrict__ out) {
  // one 16-lane group per output string: lanes stride over the bytes
  const int64_t gid = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 4;
  const int sub = threadIdx.x & 15;
  const int64_t stride = ((int64_t)gridDim.x * kBlock) >> 4;
  for (int64_t i = gid; i < m; i += stride) {
    const int32_t s = src_starts[i], d = dst_offsets[i], len = dst_offsets[i + 1] - d;
    for (int32_t b = sub; b < len; b += 16) out[d + b] = data[s + b];
  }
}

// ---------------------------------------------------------------------------------------------
// K2/K3 project
// ---------------------------------------------------------------------------------------------
template <typename POL>
__global__ __launch_bounds__(kBlock) void k_project(const DevProgram P, const DevColumns C,
                                                    const DevProjectPlan plan, const int64_t n,
                                                    uint32_t* __restrict__ ctrl) {
  typedef typename POL::COLV COLV;
  constexpr int U = POL::U;
  const int lane = lane_id();
  const int64_t n_words = (n + 63) >> 6;
  const int64_t wave_global = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * kBlock) >> 6;
  DevFastPlan F_unused;
  F_unused.valid = 0;
  uint32_t err = 0;
  for (int64_t w0 = wave_global * U; w0 < n_words; w0 += n_waves * U) {
    COLV col[U];
    uint32_t cv[U];
    FOR_U {
      const int64_t row = (w0 + u) * 64 + lane;
      POL::load(P, C, row, row < n, col[u], cv[u]);
    }
#pragma nounroll
    for (int uu = 0; uu < U; ++uu) {
      const int64_t w = w0 + uu;
      if (w >= n_words) break;  // wave-uniform
      COLV cur;
      uint32_t curv;
      DFX_SELECT_BANK(uu, col, cv, cur, curv)
      const int64_t row = w * 64 + lane;
      const bool inb = row < n;
      u64x16 reg;
      uint32_t rv = 0;
      POL::eval(P, F_unused, cur, curv, reg, rv, inb, err);
#pragma unroll
      for (int o = 0; o < kMaxOut; ++o) {
        if (o < plan.n_out) {
          uint64_t v;
          bool valid;
          fetch(P, cur, reg, curv, rv, plan.out[o], v, valid);
          const uint8_t t = plan.out_dtype[o];
          if (t == T_BOOL) {
            const uint64_t bits = __ballot(inb && (v & 1));
            if (lane == 0) ((uint64_t*)plan.out_values[o])[w] = bits;
          } else if (inb) {
            store_typed(t, plan.out_values[o], row, v);
          }
          if (plan.out_validity[o] != nullptr) {
            const uint64_t vb = __ballot(inb && valid);
            if (lane == 0) plan.out_validity[o][w] = vb;
          }
        }
      }
    }
  }
  if (err) atomicOr(&ctrl[CTRL_ERROR], err);
}

// AccumulatorSet::accumulate_scalar for the batch scalars (aggregate.rs:107-145/:176-214/:245-283),
// executed by one thread; then re-arms the batch partials with their identities.
// func: 0 min, 1 max, 2 sum, 3 count.  state[2a] = has, state[2a+1] = value bits (canonical).
__global__ void k_reduce_fold(const DevTable T, const uint8_t* __restrict__ arg_dtype,
                              const uint8_t* __restrict__ func, uint64_t* __restrict__ partial,
                              uint64_t* __restrict__ state, uint32_t* __restrict__ ctrl) {
  // one wave: lane = copy of the batch partial (kReduceSlots == 64).  Combine the copies per aggregate with a
  // shuffle tree (all three words are associative and commutative), re-arm them; lane a then folds aggregate a.
  static_assert(kReduceSlots == 64, "one lane per partial copy");
  const int lane = threadIdx.x;
  uint64_t* mine = partial + (size_t)lane * kReduceSlotWords;
  uint64_t passed = mine[3];
  mine[3] = 0;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) passed += shfl_xor_u64(passed, m);
  if (lane == 0 && passed && T.stats) atomicAdd((unsigned long long*)&T.stats[STAT_PASSED], (unsigned long long)passed);
  (void)ctrl;
  uint64_t accw = 0, cnt = 0, first = ~0ull;
  for (int a = 0; a < T.na; ++a) {
    uint64_t x = mine[4 * a + 0], c = mine[4 * a + 1], f1 = mine[4 * a + 2];
    mine[4 * a + 0] = T.acc_init[a];
    mine[4 * a + 1] = 0;
    mine[4 * a + 2] = ~0ull;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      const uint64_t ox = shfl_xor_u64(x, m), oc = shfl_xor_u64(c, m), of = shfl_xor_u64(f1, m);
      if (oc) x = c ? acc_combine(T.acc_kind[a], x, ox) : ox;  // copies without valid rows hold the identity
      c += oc;
      f1 = of < f1 ? of : f1;
    }
    if (lane == a) {
      accw = x;
      cnt = c;
      first = f1;
    }
  }
  const int a = lane;
  if (a >= T.na) return;
  const uint8_t t = arg_dtype[a], f = func[a];
  bool has = cnt != 0;
  uint64_t val = accw;
  if (f == 3) {
    has = true;  // deviation D3: COUNT of a batch is always Some(n)
  } else if (has && (t == T_F64 || t == T_F32) && f != 2) {
    double d = (first & 1) ? __longlong_as_double(0x7FF8000000000000ll) : f64_from_ordered(accw);
    val = (t == T_F64) ? f64_bits(d) : f32_bits((float)d);
  } else if (has && f == 2 && is_int(t)) {
    val = wrap_to(t, accw);
  }
  if (!has) return;  // Option::None: accumulator unchanged (or stays None)
  if (!state[2 * a]) {
    state[2 * a] = 1;
    state[2 * a + 1] = val;
    return;
  }
  const uint64_t cur = state[2 * a + 1];
  uint64_t out;
  if (f == 3) {
    out = cur + val;
  } else if (t == T_F64) {
    const double x = as_f64(cur), y = as_f64(val);
    out = f64_bits(f == 0 ? fmin(x, y) : f == 1 ? fmax(x, y) : x + y);
  } else if (t == T_F32) {
    const float x = as_f32(cur), y = as_f32(val);
    out = f32_bits(f == 0 ? fminf(x, y) : f == 1 ? fmaxf(x, y) : x + y);
  } else if (is_signed_int(t)) {
    const int64_t x = (int64_t)cur, y = (int64_t)val;
    out = f == 0 ? (uint64_t)(x < y ? x : y) : f == 1 ? (uint64_t)(x > y ? x : y) : wrap_to(t, cur + val);
  } else {
    out = f == 0 ? (cur < val ? cur : val) : f == 1 ? (cur > val ? cur : val) : wrap_to(t, cur + val);
  }
  state[2 * a + 1] = out;
}

__global__ __launch_bounds__(kBlock) void k_fill_u64(uint64_t* __restrict__ p, uint64_t v, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) p[i] = v;
}
__global__ __launch_bounds__(kBlock) void k_fill_u32(uint32_t* __restrict__ p, uint32_t v, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) p[i] = v;
}

// dense u64 plane -> typed output column (keys: narrow; aggregates: undo the accumulator image)
__global__ __launch_bounds__(kBlock) void k_finalize(const uint64_t* __restrict__ in, int64_t n,
                                                     uint8_t out_dtype, uint8_t val_xform,
                                                     void* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    uint64_t v = in[i];
    if (val_xform == VT_F64_ORD_MIN || val_xform == VT_F64_ORD_MAX) {
      v = f64_bits(f64_from_ordered(v));
    } else if (val_xform == VT_F32_ORD_MIN || val_xform == VT_F32_ORD_MAX) {
      v = f32_bits((float)f64_from_ordered(v));
    }
    store_typed(out_dtype, out, i, v);
  }
}

// AVG = SUM / COUNT of the same argument (deviation D7: the reference's planner types AVG, its executor has none).
// sum: the SUM accumulator word, cnt: the COUNT word; result in the argument's type: IEEE division for floats,
// truncating division of the wrapped sum for integers; null when cnt == 0.  validity: Arrow bitmap words.
__host__ __device__ inline uint64_t avg_value(uint8_t t, uint64_t sum, uint64_t cnt) {
  union { uint64_t u; double d; } c64;
  union { uint32_t u; float f; } c32;
  if (t == T_F64) {
    c64.u = sum;
    c64.d = c64.d / (double)cnt;
    return c64.u;
  }
  if (t == T_F32) {
    c32.u = (uint32_t)sum;
    c32.f = c32.f / (float)cnt;
    return (uint64_t)c32.u;
  }
  // integers: the wrapped sum in the argument's width, truncating division
  int bits = 64;
  bool is_signed = false;
  switch (t) {
    case T_I8: bits = 8; is_signed = true; break;
    case T_I16: bits = 16; is_signed = true; break;
    case T_I32: bits = 32; is_signed = true; break;
    case T_I64: bits = 64; is_signed = true; break;
    case T_U8: bits = 8; break;
    case T_U16: bits = 16; break;
    case T_U32: bits = 32; break;
    default: break;
  }
  if (is_signed) {
    const int64_t x = bits == 64 ? (int64_t)sum : (int64_t)(sum << (64 - bits)) >> (64 - bits);
    return (uint64_t)(x / (int64_t)cnt);
  }
  const uint64_t x = bits == 64 ? sum : sum & ((1ull << bits) - 1ull);
  return x / cnt;
}
uint64_t host_avg_value(uint8_t t, uint64_t sum, uint64_t cnt) { return avg_value(t, sum, cnt); }

__global__ __launch_bounds__(kBlock) void k_finalize_avg(const uint64_t* __restrict__ sum, const uint64_t* __restrict__ cnt,
                                                         int64_t n, uint8_t out_dtype, void* __restrict__ out,
                                                         uint64_t* __restrict__ validity, uint64_t* __restrict__ null_count) {
  const int64_t n_pad = (n + 63) & ~63ll;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n_pad; i += (int64_t)gridDim.x * kBlock) {
    const bool inb = i < n;
    const uint64_t c = inb ? cnt[i] : 0;
    const bool valid = inb && c != 0;
    if (inb) store_typed(out_dtype, out, i, valid ? avg_value(out_dtype, sum[i], c) : 0ull);
    const uint64_t vm = __ballot(valid), im = __ballot(inb);
    if (lane_id() == 0) {
      validity[i >> 6] = vm;
      const int nulls = __popcll(im & ~vm);
      if (nulls) atomicAdd((unsigned long long*)null_count, (unsigned long long)nulls);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// synthetic columns -- same definition as orc_synth_fill (oracle/dfx_oracle.c)
// ---------------------------------------------------------------------------------------------
DEV uint64_t synth_u64(uint64_t seed, int column_id, int64_t row) {
  const uint64_t s = seed ^ ((uint64_t)(uint32_t)column_id * 0xA0761D6478BD642Full);
  return mix64(s + ((uint64_t)row + 1ull) * 0x9E3779B97F4A7C15ull);
}

__global__ __launch_bounds__(kBlock) void k_synth(int kind, int column_id, double p0, double p1, uint64_t seed,
                                                  int64_t row_begin, int64_t n, void* __restrict__ out) {
  int zipf_bits = 0;
  const uint64_t G = (uint64_t)(int64_t)p0;
  const int exact_shift = 64 - (p0 > 0.0 ? (int)p0 : 20);
  const double exact_scale = __longlong_as_double((long long)(1023 - (p1 > 0.0 ? (int)p1 : 10)) << 52);  // 2^-S
  if (kind == 3) {
    while ((1ull << zipf_bits) < G && zipf_bits < 62) ++zipf_bits;
  }
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const uint64_t r = synth_u64(seed, column_id, row_begin + i);
    if (kind == 0) {
      const double u = (double)(r >> 11) * 0x1.0p-53;
      const double t = p1 * u;
      ((double*)out)[i] = p0 + t;
    } else if (kind == 1) {  // m * 2^-S, m below 2^B (B = p0 or 20, S = p1 or 10): exact products / sums within the bit budget
      ((double*)out)[i] = (double)(r >> exact_shift) * exact_scale;
    } else if (kind == 2) {
      ((int64_t*)out)[i] = (int64_t)__umul64hi(r, G);
    } else if (kind == 4) {
      ((int32_t*)out)[i] = (int32_t)__umul64hi(r, G);
    } else if (kind == 5) {  // DFX_SYNTH_I64_WIDE
      ((int64_t*)out)[i] = (int64_t)((__umul64hi(r, G) + 1ull) * 0x9E3779B97F4A7C15ull);
    } else {
      const uint64_t b = __umul64hi(r, (uint64_t)zipf_bits + 1ull);
      const uint64_t r2 = mix64(r ^ 0xD6E8FEB86659FD93ull);
      uint64_t k = (b == 0) ? 0 : ((1ull << (b - 1)) + __umul64hi(r2, 1ull << (b - 1)));
      if (k >= G) k = G - 1;
      ((int64_t*)out)[i] = (int64_t)k;
    }
  }
}

// validity bitmap of a synthetic column: row NULL with probability permille / 1000 (include/dfx.h: DFX_SYNTH_NULL_PERMILLE;
// same draw as orc_synth_validity).  One ballot word per 64 rows; nulls += the number of null rows.
__global__ __launch_bounds__(kBlock) void k_synth_validity(int column_id, uint32_t permille, uint64_t seed, int64_t row_begin,
                                                           int64_t n, uint64_t* __restrict__ words, unsigned long long* nulls) {
  const int64_t n_words = (n + 63) >> 6;
  const int lane = lane_id();
  unsigned long long mine = 0;
  for (int64_t w = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6; w < n_words; w += ((int64_t)gridDim.x * kBlock) >> 6) {
    const int64_t i = w * 64 + lane;
    const bool inb = i < n;
    const bool valid = inb && __umul64hi(synth_u64(seed, column_id ^ kSynthNullStream, row_begin + i), 1000ull) >= (uint64_t)permille;
    const uint64_t m = __ballot(valid);
    if (lane == 0) {
      words[w] = m;
      const int64_t rows = n - w * 64 < 64 ? n - w * 64 : 64;
      mine += (unsigned long long)(rows - __popcll(m));
    }
  }
  if (lane == 0 && mine) atomicAdd(nulls, mine);
}

// =============================================================================================
// host side: launch helpers + profiler
// =============================================================================================
static const char* kKernelNames[KID_COUNT_] = {
    "predicate_mask", "compact", "project", "reduce_all", "hash_agg", "merge_rows", "rehash",
    "emit_mask", "finalize", "scan", "synth", "fill", "gather_utf8", "partial", "partition", "partition_agg", "csv", "sort"};
const char* kernel_name(int kid) { return (kid >= 0 && kid < KID_COUNT_) ? kKernelNames[kid] : "?"; }

namespace {
struct ProfEntry {
  int64_t launches = 0;
  double total_ms = 0.0;
  double algo_bytes = 0.0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};
std::mutex g_prof_mu;
bool g_prof_on = false;
ProfEntry g_prof[KID_COUNT_];
std::vector<hipEvent_t> g_event_pool;

hipEvent_t get_event() {
  if (!g_event_pool.empty()) {
    hipEvent_t e = g_event_pool.back();
    g_event_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

void drain(ProfEntry& p) {
  for (auto& pr : p.pending) {
    float ms = 0.f;
    if (hipEventSynchronize(pr.second) == hipSuccess && hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess)
      p.total_ms += (double)ms;
    g_event_pool.push_back(pr.first);
    g_event_pool.push_back(pr.second);
  }
  p.pending.clear();
}

int g_cu_count = 0;
}  // namespace

Scope::Scope(int kid_, hipStream_t s_, double bytes) : kid(kid_), s(s_) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (!g_prof_on) return;
  a = get_event();
  b = get_event();
  g_prof[kid].launches += 1;
  g_prof[kid].algo_bytes += bytes;
  if (a) (void)hipEventRecord(a, s);
}
Scope::~Scope() {
  if (!a || !b) return;
  (void)hipEventRecord(b, s);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof[kid].pending.emplace_back(a, b);
}

int device_cu_count() {
  if (g_cu_count == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      g_cu_count = prop.multiProcessorCount;
    if (g_cu_count <= 0) g_cu_count = 256;
  }
  return g_cu_count;
}

void profile_enable(bool on) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_on = on;
}
void profile_reset() {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& p : g_prof) {
    drain(p);
    p.launches = 0;
    p.total_ms = 0;
    p.algo_bytes = 0;
  }
}
int profile_count() { return KID_COUNT_; }
bool profile_get(int index, const char** name, int64_t* launches, double* total_ms, double* algo_bytes) {
  if (index < 0 || index >= KID_COUNT_) return false;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  drain(g_prof[index]);
  *name = kKernelNames[index];
  *launches = g_prof[index].launches;
  *total_ms = g_prof[index].total_ms;
  *algo_bytes = g_prof[index].algo_bytes;
  return true;
}

// grid for a streaming kernel over `units` block-sized units: enough workgroups to fill 256 CUs
// several times over (>> 256 WGs; blocks land round-robin on the 8 XCDs), capped so the
// grid-stride loop amortises launch and tail effects.
int stream_grid(int64_t units, int per_cu) {
  int64_t cap = (int64_t)device_cu_count() * per_cu;
  if (units < 1) units = 1;
  return (int)(units < cap ? units : cap);
}

hipError_t launch_predicate_mask(const DevProgram& P, const DevFastPlan& fast, const DevColumns& C, uint8_t pred,
                                 int64_t n, uint64_t* mask_words, uint32_t* tile_counts, uint32_t* ctrl,
                                 double algo_bytes, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  Scope sc(KID_PREDICATE_MASK, s, algo_bytes);
  const int64_t tiles = (n + kTileRows - 1) / kTileRows;
  const int grid = stream_grid(tiles, 8);
#define DFX_MASK(POL) hipLaunchKernelGGL((k_predicate_mask<POL>), dim3(grid), dim3(kBlock), 0, s, P, fast, C, pred, n, mask_words, tile_counts, ctrl)
  {
    const uint8_t none[kMaxAggs] = {0};
    if (sig_matches<SigPred2F64>(P, fast, 0, 0, none, none)) {
      DFX_MASK(DFX_ARG(StaticPolicy<2, 8, SigPred2F64>));
      return hipGetLastError();
    }
  }
  const bool use_fast = fast.valid && !P.has_nulls;
  if (P.n_cols <= 2) { if (use_fast) DFX_MASK(DFX_ARG(FastPolicy<2, 8>)); else DFX_MASK(DFX_ARG(InterpPolicy<2, 8>)); }
  else if (P.n_cols <= 4) { if (use_fast) DFX_MASK(DFX_ARG(FastPolicy<4, 4>)); else DFX_MASK(DFX_ARG(InterpPolicy<4, 4>)); }
  else { if (use_fast) DFX_MASK(DFX_ARG(FastPolicy<8, 2>)); else DFX_MASK(DFX_ARG(InterpPolicy<8, 2>)); }
#undef DFX_MASK
  return hipGetLastError();
}

template <typename POL>
static hipError_t filter_fused_launch(const DevProgram& P, const DevFastPlan& fast, const DevColumns& C, uint8_t pred, int64_t n,
                                      uint64_t* mask_words, uint64_t* tile_offsets, uint64_t* sync, const DevFusedOut& O,
                                      uint32_t* ctrl, hipStream_t s) {
  const size_t lds = (size_t)O.n * kFusedTiles * kFusedStage * sizeof(uint64_t);
  // the grid must be co-resident (the look-back waits for other workgroups): workgroups per CU from the occupancy API,
  // asked once per (policy, LDS size)
  // (one device per process -- ctx().device --; the cache is atomic: two threads that both miss compute the same answer)
  static std::atomic<int> per_cu[kFusedOutCols + 1];
  int wg_per_cu = per_cu[O.n].load(std::memory_order_relaxed);
  if (wg_per_cu == 0) {
    int nb = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_filter_fused<POL>, kBlock, lds);
    if (e != hipSuccess) return e;
    wg_per_cu = nb > 0 ? nb : 1;
    per_cu[O.n].store(wg_per_cu, std::memory_order_relaxed);
  }
  const int64_t tiles = (n + kTileRows - 1) / kTileRows;
  const int64_t units = (tiles + kFusedTiles - 1) / kFusedTiles;
  const int64_t cap = (int64_t)device_cu_count() * wg_per_cu;
  const int grid = (int)(units < cap ? units : cap);
  hipLaunchKernelGGL((k_filter_fused<POL>), dim3(grid), dim3(kBlock), lds, s, P, fast, C, pred, n, mask_words, tile_offsets, sync, O, ctrl);
  return hipGetLastError();
}

template <typename POL, int FORM>
static hipError_t filter_fused_dense_launch(const DevProgram& P, const DevFastPlan& fast, const DevColumns& C, uint8_t pred, int64_t n,
                                            uint64_t* mask_words, uint64_t* tile_offsets, uint64_t* sync, const DevFusedOut& O,
                                            uint32_t* ctrl, hipStream_t s) {
  static std::atomic<int> per_cu{0};  // (co-resident grid, as above)
  int wg_per_cu = per_cu.load(std::memory_order_relaxed);
  if (wg_per_cu == 0) {
    int nb = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_filter_fused_dense<POL, FORM>, kBlock, 0);
    if (e != hipSuccess) return e;
    wg_per_cu = nb > 0 ? nb : 1;
    per_cu.store(wg_per_cu, std::memory_order_relaxed);
  }
  const int64_t tiles = (n + kTileRows - 1) / kTileRows;
  const int64_t units = (tiles + kFusedTiles - 1) / kFusedTiles;
  const int64_t cap = (int64_t)device_cu_count() * wg_per_cu;
  const int grid = (int)(units < cap ? units : cap);
  hipLaunchKernelGGL((k_filter_fused_dense<POL, FORM>), dim3(grid), dim3(kBlock), 0, s, P, fast, C, pred, n, mask_words, tile_offsets, sync, O, ctrl);
  return hipGetLastError();
}

hipError_t launch_filter_fused(const DevProgram& P, const DevFastPlan& fast, const DevColumns& C, uint8_t pred, int64_t n,
                               uint64_t* mask_words, uint64_t* tile_offsets, uint64_t* sync, const DevFusedOut& O,
                               uint32_t* ctrl, double algo_bytes, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  if (O.n < 0 || O.n > kFusedOutCols) return hipErrorInvalidValue;
  Scope sc(KID_PREDICATE_MASK, s, algo_bytes);
#define DFX_FUSED(POL) return filter_fused_launch<POL>(P, fast, C, pred, n, mask_words, tile_offsets, sync, O, ctrl, s)
  {
    const uint8_t none[kMaxAggs] = {0};
    if (sig_matches<SigPred2F64>(P, fast, 0, 0, none, none)) {
      // dense streams: the tile in registers, the column read once (the signature's one column is the one column compacted)
      const bool one_wide_column = P.n_cols == 1 && O.n == 1 && O.slot[0] == 0 && (O.dtype[0] == T_F64 || O.dtype[0] == T_I64 || O.dtype[0] == T_U64);
      if (O.dense && one_wide_column) {
        typedef StaticPolicy<2, 8, SigPred2F64> POLD;
#define DFX_DENSE(FORM) return filter_fused_dense_launch<POLD, FORM>(P, fast, C, pred, n, mask_words, tile_offsets, sync, O, ctrl, s)
        switch (POLD::form_of(fast)) {  // the comparison form is a template argument: one loop per kernel
          case 4 | (1 << 3): DFX_DENSE(4 | (1 << 3));  // x >  a AND x <  b
          case 6 | (1 << 3): DFX_DENSE(6 | (1 << 3));  // x >= a AND x <  b
          case 4 | (3 << 3): DFX_DENSE(4 | (3 << 3));  // x >  a AND x <= b
          case 6 | (3 << 3): DFX_DENSE(6 | (3 << 3));  // x >= a AND x <= b
          default: DFX_DENSE(0);
        }
#undef DFX_DENSE
      }
      DFX_FUSED(DFX_ARG(StaticPolicy<2, 8, SigPred2F64>));
    }
  }
  const bool use_fast = fast.valid && !P.has_nulls;
  if (P.n_cols <= 2) { if (use_fast) DFX_FUSED(DFX_ARG(FastPolicy<2, 8>)); else DFX_FUSED(DFX_ARG(InterpPolicy<2, 8>)); }
  else if (P.n_cols <= 4) { if (use_fast) DFX_FUSED(DFX_ARG(FastPolicy<4, 4>)); else DFX_FUSED(DFX_ARG(InterpPolicy<4, 4>)); }
  else { if (use_fast) DFX_FUSED(DFX_ARG(FastPolicy<8, 2>)); else DFX_FUSED(DFX_ARG(InterpPolicy<8, 2>)); }
#undef DFX_FUSED
}

// mask &= other, per-tile counts of the result: a predicate too large for one fused program is evaluated as several
// conjuncts (FilterRelation, dfx_relation.cpp); one wave per 64-word tile
__global__ __launch_bounds__(256) void k_mask_and_count(uint64_t* __restrict__ mask, const uint64_t* __restrict__ other,
                                                        uint32_t* __restrict__ tile_counts, const int64_t n_words,
                                                        const int64_t n_tiles) {
  const int lane = lane_id();
  const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tile >= n_tiles) return;
  const int64_t w = tile * 64 + lane;
  uint64_t word = 0;
  if (w < n_words) {
    word = mask[w] & other[w];
    mask[w] = word;
  }
  uint32_t cnt = (uint32_t)__popcll(word);
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) cnt += (uint32_t)__shfl_xor((int)cnt, m, 64);
  if (lane == 0) tile_counts[tile] = cnt;
}
hipError_t launch_mask_and_count(uint64_t* mask, const uint64_t* other, uint32_t* tile_counts, int64_t n, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  const int64_t n_words = (n + 63) / 64, n_tiles = (n + kTileRows - 1) / kTileRows;
  hipLaunchKernelGGL(k_mask_and_count, dim3((unsigned)((n_tiles + 3) / 4)), dim3(256), 0, s, mask, other, tile_counts, n_words, n_tiles);
  return hipGetLastError();
}

template <typename TIN, typename TOUT>
static hipError_t scan_impl(const TIN* in, TOUT* out, int64_t n, uint64_t* tmp, hipStream_t s) {
  Scope sc(KID_SCAN, s, 0);
  const int64_t nb = n > 0 ? (n + kScanChunk - 1) / kScanChunk : 1;
  hipLaunchKernelGGL((k_scan_local<TIN>), dim3((unsigned)nb), dim3(kBlock), 0, s, in, n, tmp);
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(kBlock), 0, s, tmp, nb);
  hipLaunchKernelGGL((k_scan_apply<TIN, TOUT>), dim3((unsigned)nb), dim3(kBlock), 0, s, in, n, tmp, nb, out);
  return hipGetLastError();
}
hipError_t launch_scan_u32(const uint32_t* in, uint64_t* out, int64_t n, uint64_t* tmp, hipStream_t s) {
  return scan_impl<uint32_t, uint64_t>(in, out, n, tmp, s);
}
hipError_t launch_scan_i32(const int32_t* in, int32_t* out, int64_t n, uint64_t* tmp, hipStream_t s) {
  return scan_impl<int32_t, int32_t>(in, out, n, tmp, s);
}

hipError_t launch_compact(const void* in, int width, const uint64_t* mask_words, const uint64_t* tile_offsets,
                          int64_t n, void* out, double algo_bytes, hipStream_t s, uint64_t out_limit) {
  if (n <= 0) return hipSuccess;
  Scope sc(KID_COMPACT, s, algo_bytes);
  const int64_t tiles = (n + kTileRows - 1) / kTileRows;
  const int grid = stream_grid(tiles, 8);
  switch (width) {
    case 8: hipLaunchKernelGGL(k_compact<uint64_t>, dim3(grid), dim3(kBlock), 0, s, (const uint64_t*)in, mask_words, tile_offsets, n, (uint64_t*)out, out_limit); break;
    case 4: hipLaunchKernelGGL(k_compact<uint32_t>, dim3(grid), dim3(kBlock), 0, s, (const uint32_t*)in, mask_words, tile_offsets, n, (uint32_t*)out, out_limit); break;
    case 2: hipLaunchKernelGGL(k_compact<uint16_t>, dim3(grid), dim3(kBlock), 0, s, (const uint16_t*)in, mask_words, tile_offsets, n, (uint16_t*)out, out_limit); break;
    case 1: hipLaunchKernelGGL(k_compact<uint8_t>, dim3(grid), dim3(kBlock), 0, s, (const uint8_t*)in, mask_words, tile_offsets, n, (uint8_t*)out, out_limit); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_utf8_lengths(const int32_t* offsets, int64_t n, int32_t* lengths, int32_t* starts, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  Scope sc(KID_GATHER_UTF8, s, 0);
  const int grid = stream_grid((n + kBlock - 1) / kBlock, 8);
  hipLaunchKernelGGL(k_utf8_lengths, dim3(grid), dim3(kBlock), 0, s, offsets, n, lengths, starts);
  return hipGetLastError();
}
hipError_t launch_utf8_gather(const uint8_t* data, const int32_t* src_starts, const int32_t* dst_offsets,
                              int64_t m, uint8_t* out, hipStream_t s) {
  if (m <= 0) return hipSuccess;
  Scope sc(KID_GATHER_UTF8, s, 0);
  const int grid = stream_grid((m * 16 + kBlock - 1) / kBlock, 8);
  hipLaunchKernelGGL(k_utf8_gather, dim3(grid), dim3(kBlock), 0, s, data, src_starts, dst_offsets, m, out);
  return hipGetLastError();
}

hipError_t launch_project(const DevProgram& P, const DevColumns& C, const DevProjectPlan& plan, int64_t n,
                          uint32_t* ctrl, double algo_bytes, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  Scope sc(KID_PROJECT, s, algo_bytes);
  const int grid = stream_grid((n + kBlock - 1) / kBlock, 8);
  if (P.n_cols <= 2)
    hipLaunchKernelGGL((k_project<InterpPolicy<2, 8>>), dim3(grid), dim3(kBlock), 0, s, P, C, plan, n, ctrl);
  else if (P.n_cols <= 4)
    hipLaunchKernelGGL((k_project<InterpPolicy<4, 4>>), dim3(grid), dim3(kBlock), 0, s, P, C, plan, n, ctrl);
  else
    hipLaunchKernelGGL((k_project<InterpPolicy<8, 2>>), dim3(grid), dim3(kBlock), 0, s, P, C, plan, n, ctrl);
  return hipGetLastError();
}

hipError_t launch_reduce_fold(const DevTable& T, const uint8_t* arg_dtype, const uint8_t* func,
                              uint64_t* partial, uint64_t* state, uint32_t* ctrl, hipStream_t s) {
  hipLaunchKernelGGL(k_reduce_fold, dim3(1), dim3(64), 0, s, T, arg_dtype, func, partial, state, ctrl);
  return hipGetLastError();
}


DFX_DECLARE_TABLE_KW(1)
DFX_DECLARE_TABLE_KW(2)
DFX_DECLARE_TABLE_KW(3)
DFX_DECLARE_TABLE_KW(4)
DFX_DECLARE_TABLE_KW(8)

#define DFX_KW_DISPATCH(kw, CALL)            \
  switch (kw) {                              \
    case 1: return CALL(1);                  \
    case 2: return CALL(2);                  \
    case 3: return CALL(3);                  \
    case 4: return CALL(4);                  \
    case 8: return CALL(8);                  \
    default: return hipErrorInvalidValue;    \
  }

hipError_t launch_hash_agg(const DevProgram& P, const DevFastPlan& fast, const DevColumns& C, const DevAggPlan& plan,
                           const DevTable& T, const DevRows& spill, int64_t n, double algo_bytes, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  Scope sc(KID_HASH_AGG, s, algo_bytes);
#define CALL(K) table_hash_agg<K>(P, fast, C, plan, T, spill, n, s)
  DFX_KW_DISPATCH(T.kw, CALL)
#undef CALL
}

hipError_t launch_merge_rows(const DevRows& rows, int64_t row_begin, int64_t n_rows, const DevTable& T,
                             const DevRows& spill, hipStream_t s) {
  if (n_rows <= 0) return hipSuccess;
  Scope sc(KID_MERGE_ROWS, s, 0);
#define CALL(K) table_merge_rows<K>(rows, row_begin, n_rows, T, spill, s)
  DFX_KW_DISPATCH(T.kw, CALL)
#undef CALL
}

hipError_t launch_merge_bucket(const uint64_t* bucket, uint64_t count, const DevTable& T, const DevRows& spill,
                               hipStream_t s) {
  DevRows rows;
  rows.words = const_cast<uint64_t*>(bucket);
  rows.capacity = count;
  return launch_merge_rows(rows, 0, (int64_t)count, T, spill, s);
}

hipError_t launch_rehash(const DevTable& from, const DevTable& to, const DevRows& spill, hipStream_t s) {
  Scope sc(KID_REHASH, s, 0);
#define CALL(K) table_rehash<K>(from, to, spill, s)
  DFX_KW_DISPATCH(from.kw, CALL)
#undef CALL
}

hipError_t launch_table_mask(const DevTable& T, uint64_t* mask_words, uint32_t* tile_counts, hipStream_t s) {
  Scope sc(KID_EMIT_MASK, s, 0);
#define CALL(K) table_mask<K>(T, mask_words, tile_counts, s)
  DFX_KW_DISPATCH(T.kw, CALL)
#undef CALL
}

hipError_t launch_partial_count(const DevTable& T, int world, uint64_t* counts, hipStream_t s) {
  Scope sc(KID_PARTIAL, s, 0);
#define CALL(K) table_partial_count<K>(T, world, counts, s)
  DFX_KW_DISPATCH(T.kw, CALL)
#undef CALL
}

hipError_t launch_partial_scatter(const DevTable& T, int world, const uint64_t* bucket_base,
                                  const uint64_t* bucket_count, uint64_t* cursors, uint64_t* dst, hipStream_t s) {
  Scope sc(KID_PARTIAL, s, 0);
#define CALL(K) table_partial_scatter<K>(T, world, bucket_base, bucket_count, cursors, dst, s)
  DFX_KW_DISPATCH(T.kw, CALL)
#undef CALL
}

// Result download by the shader: 16-byte loads from HBM, 16-byte stores to host-mapped pinned memory (the destination
// is hipHostMalloc'd, device-accessible).  The copy engines' path showed sporadic 6 - 150 ms stalls on these boxes for an
// 8 MB device-to-host copy (tools/d2h_probe.py: 1 in 300 with torch's own copy, too); a kernel on the query's stream has
// no such outliers and needs no engine hand-over.  `bytes` any value; src / dst 16-byte aligned (pool allocations are).
__global__ __launch_bounds__(256) void k_copy_to_host(const uint4* __restrict__ src, uint4* __restrict__ dst, int64_t n16,
                                                      const uint8_t* __restrict__ src_tail, uint8_t* __restrict__ dst_tail, int tail) {
  // four 16-byte pieces per lane in flight (the PCIe writes are posted; the loads are what a lane waits for)
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
    dst[i] = a;
    dst[i + stride] = b;
    dst[i + 2 * stride] = c;
    dst[i + 3 * stride] = d;
  }
  for (; i < n16; i += stride) dst[i] = src[i];
  if (blockIdx.x == 0 && (int)threadIdx.x < tail) dst_tail[threadIdx.x] = src_tail[threadIdx.x];
}
hipError_t launch_copy_to_host(const void* src_device, void* dst_pinned, size_t bytes, hipStream_t s) {
  if (bytes == 0) return hipSuccess;
  if ((((uintptr_t)src_device) | ((uintptr_t)dst_pinned)) & 15u) return hipMemcpyAsync(dst_pinned, src_device, bytes, hipMemcpyDeviceToHost, s);
  const int64_t n16 = (int64_t)(bytes / 16);
  const int tail = (int)(bytes % 16);
  const int grid = (int)std::min<int64_t>(512, std::max<int64_t>(1, (n16 + 1023) / 1024));
  hipLaunchKernelGGL(k_copy_to_host, dim3(grid), dim3(256), 0, s, (const uint4*)src_device, (uint4*)dst_pinned, n16,
                     (const uint8_t*)src_device + n16 * 16, (uint8_t*)dst_pinned + n16 * 16, tail);
  return hipGetLastError();
}

hipError_t launch_fill_u64(uint64_t* p, uint64_t v, int64_t n, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  Scope sc(KID_FILL, s, 0);
  hipLaunchKernelGGL(k_fill_u64, dim3(stream_grid((n + kBlock - 1) / kBlock, 8)), dim3(kBlock), 0, s, p, v, n);
  return hipGetLastError();
}
hipError_t launch_fill_u32(uint32_t* p, uint32_t v, int64_t n, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  Scope sc(KID_FILL, s, 0);
  hipLaunchKernelGGL(k_fill_u32, dim3(stream_grid((n + kBlock - 1) / kBlock, 8)), dim3(kBlock), 0, s, p, v, n);
  return hipGetLastError();
}

hipError_t launch_finalize(const uint64_t* in, int64_t n, uint8_t out_dtype, uint8_t val_xform, void* out,
                           hipStream_t s) {
  if (n <= 0) return hipSuccess;
  Scope sc(KID_FINALIZE, s, 0);
  hipLaunchKernelGGL(k_finalize, dim3(stream_grid((n + kBlock - 1) / kBlock, 8)), dim3(kBlock), 0, s, in, n, out_dtype, val_xform, out);
  return hipGetLastError();
}

hipError_t launch_finalize_avg(const uint64_t* sum, const uint64_t* cnt, int64_t n, uint8_t out_dtype, void* out,
                               uint64_t* validity, uint64_t* null_count, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  Scope sc(KID_FINALIZE, s, 0);
  hipLaunchKernelGGL(k_finalize_avg, dim3(stream_grid((n + kBlock - 1) / kBlock, 8)), dim3(kBlock), 0, s, sum, cnt, n, out_dtype, out,
                     validity, null_count);
  return hipGetLastError();
}

hipError_t launch_synth_validity(int column_id, uint32_t permille, uint64_t seed, int64_t row_begin, int64_t n, uint64_t* words,
                                 uint64_t* nulls, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  Scope sc(KID_SYNTH, s, 0);
  hipLaunchKernelGGL(k_synth_validity, dim3(stream_grid((n + kBlock - 1) / kBlock, 16)), dim3(kBlock), 0, s, column_id, permille, seed, row_begin, n,
                     words, (unsigned long long*)nulls);
  return hipGetLastError();
}

hipError_t launch_synth(int kind, int column_id, double p0, double p1, uint64_t seed, int64_t row_begin, int64_t n,
                        void* out, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  Scope sc(KID_SYNTH, s, 0);
  hipLaunchKernelGGL(k_synth, dim3(stream_grid((n + kBlock - 1) / kBlock, 16)), dim3(kBlock), 0, s, kind, column_id, p0, p1, seed, row_begin, n, out);
  return hipGetLastError();
}

}  // namespace dfx
