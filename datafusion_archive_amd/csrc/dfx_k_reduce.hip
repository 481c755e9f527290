// dfx_k_reduce.hip -- K5 reduce_all (ungrouped aggregates), its own translation unit.
#include <type_traits>

#include "dfx_kernels_inl.hpp"
#include "dfx_launch.hpp"

namespace dfx {
// ---------------------------------------------------------------------------------------------
// K5 reduce_all (ungrouped aggregates of one batch)
// ---------------------------------------------------------------------------------------------
// partial layout (kReduceSlots copies, see dfx_device.hpp) per aggregate a: [4a+0] accumulator word (pre-filled
// with the identity), [4a+1] number of valid arguments, [4a+2] min over (row << 1 | is_nan) of valid rows
// (u64::MAX when none): arrow 0.12 min/max scan with `<` / `>`, so a NaN in the first valid slot sticks;
// word [3] of a copy counts the rows that passed the predicate.
template <typename POL, int NAMAX>
__global__ __launch_bounds__(kBlock) void k_reduce(const DevProgram P, const DevFastPlan F, const DevColumns C,
                                                   const DevAggPlan plan, const DevTable T,
                                                   const int64_t n, uint64_t* __restrict__ partial,
                                                   uint32_t* __restrict__ ctrl) {
  typedef typename POL::COLV COLV;
  constexpr int U = POL::U;
  __shared__ uint64_t lds[kBlock / 64][kMaxAggs * 3];
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  const int64_t n_words = (n + 63) >> 6;
  const int64_t wave_global = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * kBlock) >> 6;
  uint64_t acc[NAMAX], cnt[NAMAX], first[NAMAX];
#pragma unroll
  for (int a = 0; a < NAMAX; ++a) {
    acc[a] = T.acc_init[a];
    cnt[a] = 0;
    first[a] = ~0ull;
  }
  uint32_t err = 0;
  uint64_t passed = 0;
  typename POL::PREP prep;  // (PlanPolicy: the plan words in vector registers; empty otherwise)
  POL::prepare(P, F, prep);
  // the scan loop once per comparison FORM (StaticPolicy::pass_form: the operators as compile-time constants; 0: run-time masks)
  auto scan = [&](auto form_tag) {
  constexpr int FORM = decltype(form_tag)::value;
  // software pipeline, one trip deep (round 6, as the partitioned scans): the next trip's columns are in flight while this one is folded
#ifndef DFX_REDUCE_PREFETCH
#define DFX_REDUCE_PREFETCH 1  // (0: round 5's form -- load, wait, evaluate -- for A/B builds)
#endif
#if DFX_REDUCE_PREFETCH
  COLV ncol[U];
  uint32_t ncv[U];
  load_trip<POL>(P, C, wave_global * U, wave_global * U < n_words, n, lane, ncol, ncv);
  for (int64_t w0 = wave_global * U; w0 < n_words; w0 += n_waves * U) {
    COLV col[U];
    uint32_t cv[U];
    FOR_U {
      col[u] = ncol[u];
      cv[u] = ncv[u];
    }
    load_trip<POL>(P, C, w0 + n_waves * U, w0 + n_waves * U < n_words, n, lane, ncol, ncv);
#else
  for (int64_t w0 = wave_global * U; w0 < n_words; w0 += n_waves * U) {
    COLV col[U];
    uint32_t cv[U];
    load_trip<POL>(P, C, w0, true, n, lane, col, cv);
#endif
    auto body = [&](const COLV& cur, const uint32_t curv, const int64_t row) {
      const bool inb = row < n;
      u64x16 reg;
      uint32_t rv = 0;
      POL::eval(P, F, cur, curv, reg, rv, inb, err, prep);
      const bool pass = inb && POL::template pass_form<FORM>(P, F, plan.pred, cur, curv, reg, rv, prep);
      if (pass) {
        ++passed;
#pragma unroll
        for (int a = 0; a < NAMAX; ++a) {
          if (a < POL::na(T)) {
            uint64_t v;
            bool valid;
            POL::arg(P, F, plan.arg[a], a, cur, curv, reg, rv, v, valid);
            const uint8_t xf = POL::xform(T, a);
            if (xf == VT_COUNT_VALID) {
              acc[a] += valid ? 1ull : 0ull;
              cnt[a] += 1;
            } else if (valid) {  // array_ops::{min,max,sum} skip nulls
              if (xf != VT_RAW) {
                const double d = (xf == VT_F32_ORD_MIN || xf == VT_F32_ORD_MAX) ? (double)as_f32(v) : as_f64(v);
                const uint64_t tag = ((uint64_t)row << 1) | (d != d ? 1ull : 0ull);
                first[a] = tag < first[a] ? tag : first[a];
              }
              acc[a] = acc_combine(POL::acc_kind(T, a), acc[a], transform_value(xf, v, valid));
              cnt[a] += 1;
            }
          }
        }
      }
    };
    if constexpr (POL::kStaticNa > 0) {
      // compile-time shapes: the per-group code is a handful of instructions -- unroll it (the run-time
      // bank select below costs more scalar branches than the work itself)
      FOR_U body(col[u], cv[u], (w0 + u) * 64 + lane);
    } else {
      // ONE copy of the (large) generic evaluation code: a run-time loop over the U prefetched row-groups
#pragma nounroll
      for (int uu = 0; uu < U; ++uu) {
        COLV cur;
        uint32_t curv;
        DFX_SELECT_BANK(uu, col, cv, cur, curv)
        body(cur, curv, (w0 + uu) * 64 + lane);
      }
    }
  }
  };
  switch (POL::form_of(F)) {  // (wave-uniform: the plan sits in the kernarg segment)
    case 4 | (1 << 3): scan(std::integral_constant<int, (POL::kIsStatic ? (4 | (1 << 3)) : 0)>{}); break;  // x >  a AND x <  b
    case 6 | (1 << 3): scan(std::integral_constant<int, (POL::kIsStatic ? (6 | (1 << 3)) : 0)>{}); break;  // x >= a AND x <  b
    case 4 | (3 << 3): scan(std::integral_constant<int, (POL::kIsStatic ? (4 | (3 << 3)) : 0)>{}); break;  // x >  a AND x <= b
    case 6 | (3 << 3): scan(std::integral_constant<int, (POL::kIsStatic ? (6 | (3 << 3)) : 0)>{}); break;  // x >= a AND x <= b
    default: scan(std::integral_constant<int, 0>{}); break;
  }
  // wave tree (xor butterfly), then one atomic per workgroup per word
#pragma unroll
  for (int a = 0; a < NAMAX; ++a) {
    if (a < POL::na(T)) {
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) {
        acc[a] = acc_combine(POL::acc_kind(T, a), acc[a], shfl_xor_u64(acc[a], m));
        cnt[a] += shfl_xor_u64(cnt[a], m);
        const uint64_t of = shfl_xor_u64(first[a], m);
        first[a] = of < first[a] ? of : first[a];
      }
      if (lane == 0) {
        lds[wave][a * 3 + 0] = acc[a];
        lds[wave][a * 3 + 1] = cnt[a];
        lds[wave][a * 3 + 2] = first[a];
      }
    }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) passed += shfl_xor_u64(passed, m);
  __syncthreads();
  if (threadIdx.x < T.na) {
    const int a = threadIdx.x;
    uint64_t x = lds[0][a * 3], c = lds[0][a * 3 + 1], f = lds[0][a * 3 + 2];
    for (int w = 1; w < kBlock / 64; ++w) {
      x = acc_combine(T.acc_kind[a], x, lds[w][a * 3]);
      c += lds[w][a * 3 + 1];
      f = lds[w][a * 3 + 2] < f ? lds[w][a * 3 + 2] : f;
    }
    uint64_t* mine = partial + (size_t)(blockIdx.x % kReduceSlots) * kReduceSlotWords;
    if (c) {
      acc_atomic(T.acc_kind[a], &mine[4 * a + 0], x);
      atomicAdd((unsigned long long*)&mine[4 * a + 1], (unsigned long long)c);
      atomicMin((unsigned long long*)&mine[4 * a + 2], (unsigned long long)f);
    }
  }
  if (lane == 0 && passed)
    atomicAdd((unsigned long long*)&partial[(size_t)(blockIdx.x % kReduceSlots) * kReduceSlotWords + 3], (unsigned long long)passed);
  if (err) atomicOr(&ctrl[CTRL_ERROR], err);
}

hipError_t launch_reduce(const DevProgram& P, const DevFastPlan& fast, const DevColumns& C, const DevAggPlan& plan,
                         const DevTable& T, int64_t n, uint64_t* partial, uint32_t* ctrl, double algo_bytes,
                         hipStream_t s) {
  if (n <= 0) return hipSuccess;
  Scope sc(KID_REDUCE, s, algo_bytes);
  const int grid = stream_grid((n + kBlock - 1) / kBlock, 8);
#define DFX_REDUCE(POL, NM) hipLaunchKernelGGL((k_reduce<POL, NM>), dim3(grid), dim3(kBlock), 0, s, P, fast, C, plan, T, n, partial, ctrl)
#define DFX_REDUCE_NA(POL) do { if (T.na <= 2) DFX_REDUCE(DFX_ARG(POL), 2); else DFX_REDUCE(DFX_ARG(POL), 8); } while (0)
  // compile-time shape signatures first (dfx_sigs.hpp), then the run-time decoded fast plan, then
  // the generic interpreter
  if (sig_matches<SigCountPred2F64>(P, fast, 0, T.na, T.acc_kind, T.val_xform)) {
    DFX_REDUCE(DFX_ARG(StaticPolicy<2, 8, SigCountPred2F64>), 2);
    return hipGetLastError();
  }
  if (sig_matches<SigSumCountPred2F64>(P, fast, 0, T.na, T.acc_kind, T.val_xform)) {
    DFX_REDUCE(DFX_ARG(StaticPolicy<2, 8, SigSumCountPred2F64>), 2);
    return hipGetLastError();
  }
  if (P.has_nulls || !P.wide8 || (fast.plan_mode & 3) == 2) {  // validity bitmaps / 4-byte columns: the scan plan (see table_hash_agg)
    DevFastPlan fp;
    DevColumns cp;
    if (bind_scan_plan(P, fast, C, 0, T.na, T.val_xform, false, &fp, &cp)) {
#define DFX_REDUCE_P(POL, NM) hipLaunchKernelGGL((k_reduce<POL, NM>), dim3(grid), dim3(kBlock), 0, s, P, fp, cp, plan, T, n, partial, ctrl)
      if (fp.scan.n_cols <= 2) { if (T.na <= 2) DFX_REDUCE_P(DFX_ARG(PlanPolicyN<2, 4, kPlanW4 | kPlanNulls>), 2); else DFX_REDUCE_P(DFX_ARG(PlanPolicyN<2, 4, kPlanW4 | kPlanNulls>), 8); }
      else { if (T.na <= 2) DFX_REDUCE_P(DFX_ARG(PlanPolicyN<4, 2, kPlanW4 | kPlanNulls>), 2); else DFX_REDUCE_P(DFX_ARG(PlanPolicyN<4, 2, kPlanW4 | kPlanNulls>), 8); }
#undef DFX_REDUCE_P
      return hipGetLastError();
    }
  }
  if (fast.plan_mode & 4) return hipErrorNotSupported;  // (the host fused a predicate over nulls counting on a plan)
  const bool use_fast = fast.valid && !P.has_nulls;
  if (P.n_cols <= 2) { if (use_fast) DFX_REDUCE_NA(DFX_ARG(FastPolicy<2, 8>)); else DFX_REDUCE_NA(DFX_ARG(InterpPolicy<2, 8>)); }
  else if (P.n_cols <= 4) { if (use_fast) DFX_REDUCE_NA(DFX_ARG(FastPolicy<4, 4>)); else DFX_REDUCE_NA(DFX_ARG(InterpPolicy<4, 4>)); }
  else { if (use_fast) DFX_REDUCE_NA(DFX_ARG(FastPolicy<8, 2>)); else DFX_REDUCE_NA(DFX_ARG(InterpPolicy<8, 2>)); }
#undef DFX_REDUCE_NA
#undef DFX_REDUCE
  return hipGetLastError();
}

}  // namespace dfx
