// dfx_launch.hpp -- host-side helpers shared by the kernel translation units (profiler scope,
// grid sizing) and the per-KW entry points of the group-table kernels.
#pragma once
#include <hip/hip_runtime_api.h>

#include "dfx_kernels.hpp"

namespace dfx {

// Brackets one tracked launch with HIP events on the launch stream while profiling is enabled.
struct Scope {
  int kid;
  hipStream_t s;
  hipEvent_t a = nullptr, b = nullptr;
  Scope(int kid, hipStream_t s, double algo_bytes);
  ~Scope();
  Scope(const Scope&) = delete;
  Scope& operator=(const Scope&) = delete;
};

// grid for a streaming kernel over `units` block-sized units: enough workgroups to fill 256 CUs
// several times over (>> 256 WGs; blocks land round-robin on the 8 XCDs), capped so the
// grid-stride loop amortises launch and tail effects.
int stream_grid(int64_t units, int per_cu);
#define DFX_ARG(...) __VA_ARGS__  // protects template commas inside macro arguments

// group-table kernels, one explicit instantiation per key width (dfx_k_table{1,2,3,4}.hip)
template <int KW>
hipError_t table_hash_agg(const DevProgram& P, const DevFastPlan& fast, const DevColumns& C, const DevAggPlan& plan,
                          const DevTable& T, const DevRows& spill, int64_t n, hipStream_t s);
template <int KW>
hipError_t table_merge_rows(const DevRows& rows, int64_t row_begin, int64_t n_rows, const DevTable& T,
                            const DevRows& spill, hipStream_t s);
template <int KW>
hipError_t table_rehash(const DevTable& from, const DevTable& to, const DevRows& spill, hipStream_t s);
template <int KW>
hipError_t table_mask(const DevTable& T, uint64_t* mask_words, uint32_t* tile_counts, hipStream_t s);
template <int KW>
hipError_t table_partial_count(const DevTable& T, int world, uint64_t* counts, hipStream_t s);
template <int KW>
hipError_t table_partial_scatter(const DevTable& T, int world, const uint64_t* bucket_base,
                                 const uint64_t* bucket_count, uint64_t* cursors, uint64_t* dst, hipStream_t s);

#define DFX_DECLARE_TABLE_KW(KW)                                                                                   \
  extern template hipError_t table_hash_agg<KW>(const DevProgram&, const DevFastPlan&, const DevColumns&,         \
                                                const DevAggPlan&, const DevTable&, const DevRows&, int64_t,      \
                                                hipStream_t);                                                     \
  extern template hipError_t table_merge_rows<KW>(const DevRows&, int64_t, int64_t, const DevTable&,              \
                                                  const DevRows&, hipStream_t);                                   \
  extern template hipError_t table_rehash<KW>(const DevTable&, const DevTable&, const DevRows&, hipStream_t);      \
  extern template hipError_t table_mask<KW>(const DevTable&, uint64_t*, uint32_t*, hipStream_t);                   \
  extern template hipError_t table_partial_count<KW>(const DevTable&, int, uint64_t*, hipStream_t);                \
  extern template hipError_t table_partial_scatter<KW>(const DevTable&, int, const uint64_t*, const uint64_t*,    \
                                                       uint64_t*, uint64_t*, hipStream_t);

#define DFX_INSTANTIATE_TABLE_KW(KW)                                                                               \
  template hipError_t table_hash_agg<KW>(const DevProgram&, const DevFastPlan&, const DevColumns&,                \
                                         const DevAggPlan&, const DevTable&, const DevRows&, int64_t,             \
                                         hipStream_t);                                                            \
  template hipError_t table_merge_rows<KW>(const DevRows&, int64_t, int64_t, const DevTable&, const DevRows&,     \
                                           hipStream_t);                                                          \
  template hipError_t table_rehash<KW>(const DevTable&, const DevTable&, const DevRows&, hipStream_t);             \
  template hipError_t table_mask<KW>(const DevTable&, uint64_t*, uint32_t*, hipStream_t);                          \
  template hipError_t table_partial_count<KW>(const DevTable&, int, uint64_t*, hipStream_t);                       \
  template hipError_t table_partial_scatter<KW>(const DevTable&, int, const uint64_t*, const uint64_t*,           \
                                                uint64_t*, uint64_t*, hipStream_t);

}  // namespace dfx
