// dfx_k_table4.hip -- group-table kernels for 4-word GROUP BY keys (see dfx_k_table_inl.hpp).
#include "dfx_k_table_inl.hpp"

namespace dfx {
DFX_INSTANTIATE_TABLE_KW(4)
}  // namespace dfx
