// dfx_k_fewgroup.hip -- GROUP BY with a handful of groups (BASELINE config 5, the TPC-H-Q1 shape: <= 6
// groups, 56 B/row): accumulators live in REGISTERS.
//
// Why: with <= 8 groups the LDS front cache of k_hash_agg still pays, per row, a key probe plus one LDS
// atomic per aggregate, and lanes of a wave that hit the same group serialise on the LDS address
// (measured 3.3 TB/s on the Q1 shape while the ungrouped K5 reduction streams at 5.5 TB/s).  Here every
// wave keeps a dictionary of the group keys it has met in SGPRs (wave-uniform) and every lane a private
// accumulator per (dictionary entry, aggregate) in VGPRs.  A row costs, per dictionary entry, KW v_cmp
// into an SGPR mask and -- under that mask as EXEC -- one ALU op per aggregate: no LDS, no atomics, no
// probing.  A key that is not in the dictionary yet takes a wave-uniform slow path (readlane of the first
// such lane -> new entry); when the dictionary is full the row goes to the global table like in K7.
// At the end a wave folds its lanes with a shuffle tree, the workgroup combines its 16 waves in a small
// LDS table (agent-scope atomics on ONE address serialise at ~11 ns, so per-wave merges would cost
// ~90 us per launch) and every distinct group of the workgroup becomes ONE table_apply.
//
// Replaces (for this case) the per-row work of with_group_by + update_accumulators
// (aggregate.rs:787-875, :548-612).  Chosen by the host after the calibration slice reported <= 8 groups.
#include <algorithm>

#include "dfx_kernels_inl.hpp"
#include "dfx_launch.hpp"

namespace dfx {

constexpr int kFGBlock = 512;   // 8 waves share one LDS combine table
constexpr int kFGGroups = 8;    // dictionary entries per wave
constexpr int kFGSlots = 64;    // slots of the workgroup's LDS combine table (power of two)

DEV uint64_t readlane_u64(uint64_t v, int src) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src);
  return ((uint64_t)hi << 32) | lo;
}

template <int KW, int NA, typename POL>
__global__ __launch_bounds__(kFGBlock) void k_fewgroup_agg(const DevProgram P, const DevFastPlan F, const DevColumns C,
                                                          const DevAggPlan plan, const DevTable T, const DevRows spill,
                                                          const int64_t n) {
  typedef typename POL::COLV COLV;
  constexpr int U = POL::U;
  constexpr int G = kFGGroups;
  constexpr int S = kFGSlots;
  __shared__ uint64_t lkeys[KW * S];
  __shared__ uint64_t laccs[NA * S];
  __shared__ uint32_t lstate[S];
  const int lane = lane_id();
  const int na = POL::na(T);
  for (int i = threadIdx.x; i < S; i += kFGBlock) {
    lstate[i] = 0u;
#pragma unroll
    for (int a = 0; a < NA; ++a) laccs[a * S + i] = T.acc_init[a];
  }
  __syncthreads();

  uint64_t dk[G][KW];   // wave-uniform: the group keys this wave has met (SGPRs)
  uint64_t acc[G][NA];  // per lane
#pragma unroll
  for (int g = 0; g < G; ++g) {
#pragma unroll
    for (int k = 0; k < KW; ++k) dk[g][k] = 0;
#pragma unroll
    for (int a = 0; a < NA; ++a) acc[g][a] = T.acc_init[a];
  }
  int ng = 0;  // wave-uniform

  const int64_t n_words = (n + 63) >> 6;
  const int64_t wave_global = ((int64_t)blockIdx.x * kFGBlock + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * kFGBlock) >> 6;
  uint32_t err = 0;
  uint64_t passed = 0;
  typename POL::PREP prep;  // (PlanPolicy: the plan words in vector registers; empty otherwise)
  POL::prepare(P, F, prep);
  // Software pipeline, one trip deep (round 6): the columns of trip t + 1 are in flight while trip t is evaluated.  One
  // 512-lane workgroup per CU (the accumulators take the registers: two waves per SIMD) issued its 14 loads and waited for
  // all of them before it evaluated anything -- 57 KB per CU in flight at best, nothing while it computed: 4.8-5.0 TB/s on the
  // Q1 shape against the 7.2 TB/s two nt streams reach (tools/ubench4.hip).
#ifndef DFX_FG_PREFETCH
#define DFX_FG_PREFETCH 1  // trips in flight ahead of the one being evaluated (0: round 5's form -- load, wait, evaluate -- for A/B builds)
#endif
#if DFX_FG_PREFETCH
  // Software pipeline (round 6): the columns of trips t + 1 .. t + D are in flight while trip t is evaluated.  One 512-lane
  // workgroup per CU (the accumulators take the registers: two waves per SIMD) issued its 14 loads and waited for all of them
  // before it evaluated anything -- 57 KB per CU in flight at best, nothing while it computed: 4.85 TB/s on the Q1 shape
  // against the 7.2 TB/s two nt streams reach (tools/ubench4.hip).  D = 1: 5.7-5.8 TB/s; D = 2 (206 VGPRs) measured 5.6: not short of bytes in flight any more.
  constexpr int D = DFX_FG_PREFETCH;
  COLV ncol[D][U];
  uint32_t ncv[D][U];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const int64_t wd = wave_global * U + (int64_t)d * n_waves * U;
    load_trip<POL>(P, C, wd, wd < n_words, n, lane, ncol[d], ncv[d]);
  }
  for (int64_t w0 = wave_global * U; w0 < n_words; w0 += n_waves * U) {
    COLV col[U];
    uint32_t cv[U];
    FOR_U {
      col[u] = ncol[0][u];
      cv[u] = ncv[0][u];
    }
#pragma unroll
    for (int d = 0; d + 1 < D; ++d) {
      FOR_U {
        ncol[d][u] = ncol[d + 1][u];
        ncv[d][u] = ncv[d + 1][u];
      }
    }
    {
      const int64_t w1 = w0 + (int64_t)D * n_waves * U;
      load_trip<POL>(P, C, w1, w1 < n_words, n, lane, ncol[D - 1], ncv[D - 1]);
    }
#else
  for (int64_t w0 = wave_global * U; w0 < n_words; w0 += n_waves * U) {
    COLV col[U];
    uint32_t cv[U];
    load_trip<POL>(P, C, w0, true, n, lane, col, cv);
#endif
#pragma nounroll
    for (int uu = 0; uu < U; ++uu) {  // ONE copy of the evaluation + accumulation code
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
      for (int k = 0; k < KW; ++k) key[k] = POL::key(P, F, plan.key[k], k, cur, curv, reg, rv);
#pragma unroll
      for (int a = 0; a < kMaxAggs; ++a) {
        val[a] = 0;
        if (a < NA && a < na) {
          uint64_t v;
          bool valid;  // value(row) read blindly (aggregate.rs:561-603)
          POL::arg(P, F, plan.arg[a], a, cur, curv, reg, rv, v, valid);
          val[a] = transform_value(POL::xform(T, a), v, valid);
        }
      }
      passed += pass ? 1 : 0;
      // Match the rows against the dictionary: per entry KW v_cmp against SGPR keys give a lane mask, and under
      // that mask (as EXEC) one ALU op per aggregate updates the lane's private accumulator.  `todo` (rows not
      // accounted for yet) lives on the scalar unit.  A key the wave has not met yet becomes a new entry and the
      // match runs again -- 6 times per wave on the Q1 shape, then never again.
      uint64_t todo = __ballot(pass);
      int g0 = 0;  // entries below g0 have been matched against this row-group already
      while (todo != 0) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
          if (g >= g0 && g < ng) {  // wave-uniform
            bool eq = true;
#pragma unroll
            for (int k = 0; k < KW; ++k) eq = eq && key[k] == dk[g][k];
            const uint64_t hm = __ballot(eq) & todo;
            todo &= ~hm;
            if (hm != 0) {
              if (lane_of_mask(hm)) {
#pragma unroll
                for (int a = 0; a < NA; ++a)
                  if (a < na) acc[g][a] = acc_combine(POL::acc_kind(T, a), acc[g][a], val[a]);
              }
            }
          }
        }
        if (todo == 0) break;
        g0 = ng;
        const int leader = __ffsll((unsigned long long)todo) - 1;
        uint64_t kk[KW];
#pragma unroll
        for (int k = 0; k < KW; ++k) kk[k] = readlane_u64(key[k], leader);
        if (ng < G) {
#pragma unroll
          for (int g = 0; g < G; ++g) {
            if (g == ng) {  // wave-uniform
#pragma unroll
              for (int k = 0; k < KW; ++k) dk[g][k] = kk[k];
            }
          }
          ++ng;
        } else {  // dictionary full: the rows with this key go to the table themselves
          bool same = lane_of_mask(todo);
#pragma unroll
          for (int k = 0; k < KW; ++k) same = same && key[k] == kk[k];
          todo &= ~__ballot(same);
          bool t2 = same;
          if (t2 && table_apply<KW>(T, key, val)) t2 = false;
          spill_row<KW>(T, spill, t2, key, val);
        }
      }
    }
  }

  // ---- fold the lanes of the wave: lane g ends up with dictionary entry g (run-time loop: one code copy) ----
  uint64_t mykey[KW];
  uint64_t myval[kMaxAggs];
#pragma unroll
  for (int k = 0; k < KW; ++k) mykey[k] = 0;
#pragma unroll
  for (int a = 0; a < kMaxAggs; ++a) myval[a] = 0;
  bool mine = false;
#pragma nounroll
  for (int g = 0; g < ng; ++g) {
    uint64_t t[NA];
    uint64_t tk[KW];
#pragma unroll
    for (int a = 0; a < NA; ++a) t[a] = acc[0][a];
#pragma unroll
    for (int k = 0; k < KW; ++k) tk[k] = dk[0][k];
#pragma unroll
    for (int j = 1; j < G; ++j) {
      if (g == j) {  // wave-uniform select
#pragma unroll
        for (int a = 0; a < NA; ++a) t[a] = acc[j][a];
#pragma unroll
        for (int k = 0; k < KW; ++k) tk[k] = dk[j][k];
      }
    }
#pragma unroll
    for (int a = 0; a < NA; ++a) {
      if (a < na) {
        uint64_t v = t[a];
        const uint8_t kind = POL::acc_kind(T, a);
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) v = acc_combine(kind, v, shfl_xor_u64(v, m));
        if (lane == g) myval[a] = v;
      }
    }
    if (lane == g) {
      mine = true;
#pragma unroll
      for (int k = 0; k < KW; ++k) mykey[k] = tk[k];
    }
  }
  // ---- combine the waves of the workgroup in LDS (state word: 0 empty, 1 busy, 2 ready) ----
  bool todo = mine;
  if (mine) {
    const uint64_t h = hash_keys<KW>(mykey);
    int slot = (int)((h >> 32) & (uint64_t)(S - 1));
    int found = -1;
    int spins = 0;
    for (int p = 0; p < S && found < 0;) {
      uint32_t st = __hip_atomic_load(&lstate[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (st == 0u) {
        const uint32_t old = atomicCAS(&lstate[slot], 0u, 1u);
        if (old == 0u) {
#pragma unroll
          for (int k = 0; k < KW; ++k) lkeys[k * S + slot] = mykey[k];
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
      for (int k = 0; k < KW; ++k) same = same && (((volatile uint64_t*)lkeys)[k * S + slot] == mykey[k]);
      if (same) found = slot;
      else slot = (slot + 1) & (S - 1);
      ++p;
    }
    if (found >= 0) {
#pragma unroll
      for (int a = 0; a < NA; ++a)
        if (a < na) acc_atomic(POL::acc_kind(T, a), &laccs[a * S + found], myval[a]);
      todo = false;
    }
  }
  if (__ballot(todo) != 0) {  // combine table full (> 64 distinct groups in one workgroup): straight to the table
    if (todo && table_apply<KW>(T, mykey, myval)) todo = false;
    spill_row<KW>(T, spill, todo, mykey, myval);
  }
  __syncthreads();
  // ---- one table update per distinct group of the workgroup ----
  if (threadIdx.x < 64) {  // whole first wave (spill_row ballots)
    const bool sat = __hip_atomic_load(&T.ctrl[CTRL_SATURATED], RLX_AGENT) != 0u;
    const int i = threadIdx.x;  // S == 64
    const bool occ = lstate[i] == 2u;
    uint64_t key[KW];
    uint64_t val[kMaxAggs];
#pragma unroll
    for (int k = 0; k < KW; ++k) key[k] = lkeys[k * S + i];
#pragma unroll
    for (int a = 0; a < kMaxAggs; ++a) val[a] = (a < NA && a < na) ? laccs[a * S + i] : 0;
    bool t3 = occ;
    if (t3 && !sat) {
      if (table_apply<KW>(T, key, val)) t3 = false;
    }
    spill_row<KW>(T, spill, t3, key, val);
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) passed += shfl_xor_u64(passed, m);
  if (lane == 0) stat_add(T, STAT_PASSED, passed);
  if (err) atomicOr(&T.ctrl[CTRL_ERROR], err);
}

static_assert(kFGSlots == 64, "the flush uses one wave");

// host: can the few-group kernel run this scan?
bool fewgroup_supported(const DevProgram& P, const DevFastPlan& fast, const DevTable& T) {
  if (!(T.kw >= 1 && T.kw <= 2 && T.na >= 1 && T.na <= 4)) return false;
  if (fast.valid && !P.has_nulls) return true;
  return (fast.plan_mode & 3) != 0 && scan_plan_shape_ok(P, fast, T.kw, T.na, T.val_xform);  // (validity bitmaps: a scan plan)
}

hipError_t launch_fewgroup_agg(const DevProgram& P, const DevFastPlan& fast, const DevColumns& C, const DevAggPlan& plan,
                               const DevTable& T, const DevRows& spill, int64_t n, double algo_bytes, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  if (!fewgroup_supported(P, fast, T)) return hipErrorInvalidValue;
  Scope sc(KID_HASH_AGG, s, algo_bytes);
  const int64_t n_blocks = (n + kFGBlock - 1) / kFGBlock;
  const int grid = (int)std::min<int64_t>(n_blocks, device_cu_count());
#define DFX_FG(KW, POL) hipLaunchKernelGGL((k_fewgroup_agg<KW, 4, POL>), dim3(grid), dim3(kFGBlock), 0, s, P, fast, C, plan, T, spill, n)
  if (T.kw == 2 && sig_matches<SigQ1>(P, fast, T.kw, T.na, T.acc_kind, T.val_xform)) {
    DFX_FG(2, DFX_ARG(StaticPolicy<8, 2, SigQ1>));
    return hipGetLastError();
  }
  if (P.has_nulls || !P.wide8 || (fast.plan_mode & 3) == 2) {  // validity bitmaps / 4-byte columns: the scan plan (see table_hash_agg)
    DevFastPlan fp;
    DevColumns cp;
    if (bind_scan_plan(P, fast, C, T.kw, T.na, T.val_xform, false, &fp, &cp)) {
#define DFX_FGP(KW, POL) hipLaunchKernelGGL((k_fewgroup_agg<KW, 4, POL>), dim3(grid), dim3(kFGBlock), 0, s, P, fp, cp, plan, T, spill, n)
      if (T.kw == 1) {
        if (fp.scan.n_cols <= 2) DFX_FGP(1, DFX_ARG(PlanPolicyN<2, 4, kPlanW4 | kPlanNulls>));
        else DFX_FGP(1, DFX_ARG(PlanPolicyN<4, 2, kPlanW4 | kPlanNulls>));
      } else {
        if (fp.scan.n_cols <= 2) DFX_FGP(2, DFX_ARG(PlanPolicyN<2, 4, kPlanW4 | kPlanNulls>));
        else DFX_FGP(2, DFX_ARG(PlanPolicyN<4, 2, kPlanW4 | kPlanNulls>));
      }
#undef DFX_FGP
      return hipGetLastError();
    }
    if (P.has_nulls || (fast.plan_mode & 4)) return hipErrorNotSupported;  // (fewgroup_supported said yes for a plan only)
  }
  if (T.kw == 1) {
    if (P.n_cols <= 2) DFX_FG(1, DFX_ARG(FastPolicy<2, 4>));
    else if (P.n_cols <= 4) DFX_FG(1, DFX_ARG(FastPolicy<4, 4>));
    else DFX_FG(1, DFX_ARG(FastPolicy<8, 2>));
  } else {
    if (P.n_cols <= 2) DFX_FG(2, DFX_ARG(FastPolicy<2, 4>));
    else if (P.n_cols <= 4) DFX_FG(2, DFX_ARG(FastPolicy<4, 4>));
    else DFX_FG(2, DFX_ARG(FastPolicy<8, 2>));
  }
#undef DFX_FG
  return hipGetLastError();
}

}  // namespace dfx
