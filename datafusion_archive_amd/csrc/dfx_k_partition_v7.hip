// dfx_k_partition_v7.hip -- pass 1 of the partitioned GROUP BY for one row-source policy: InterpPolicy, <= 8 columns.
#include "dfx_k_partition_inl.hpp"
namespace dfx {
DFX_PARTITION_VARIANT(7, DFX_ARG(InterpPolicy<8, 1>), DFX_ARG(InterpPolicy<8, 1>), DFX_ARG(InterpPolicy1<8, 1>))
}  // namespace dfx
