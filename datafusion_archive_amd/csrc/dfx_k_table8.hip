// dfx_k_table8.hip -- group-table kernels for 8-word GROUP BY keys: five to eight key columns (fewer than eight are padded
// with constant zero words by the host, dfx_aggregate.cpp) -- see dfx_k_table_inl.hpp.
#include "dfx_k_table_inl.hpp"

namespace dfx {
DFX_INSTANTIATE_TABLE_KW(8)
}  // namespace dfx
