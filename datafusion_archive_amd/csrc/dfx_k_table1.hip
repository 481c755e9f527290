// dfx_k_table1.hip -- group-table kernels for 1-word GROUP BY keys (see dfx_k_table_inl.hpp).
#include "dfx_k_table_inl.hpp"

namespace dfx {
DFX_INSTANTIATE_TABLE_KW(1)
}  // namespace dfx
