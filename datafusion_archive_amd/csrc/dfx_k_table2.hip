// dfx_k_table2.hip -- group-table kernels for 2-word GROUP BY keys (see dfx_k_table_inl.hpp).
#include "dfx_k_table_inl.hpp"

namespace dfx {
DFX_INSTANTIATE_TABLE_KW(2)
}  // namespace dfx
