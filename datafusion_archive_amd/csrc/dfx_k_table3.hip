// dfx_k_table3.hip -- group-table kernels for 3-word GROUP BY keys (see dfx_k_table_inl.hpp).
#include "dfx_k_table_inl.hpp"

namespace dfx {
DFX_INSTANTIATE_TABLE_KW(3)
}  // namespace dfx
