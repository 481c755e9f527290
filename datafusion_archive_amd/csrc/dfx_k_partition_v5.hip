// dfx_k_partition_v5.hip -- pass 1 of the partitioned GROUP BY for one row-source policy: InterpPolicy, <= 4 columns.
#include "dfx_k_partition_inl.hpp"
namespace dfx {
DFX_PARTITION_VARIANT(5, DFX_ARG(InterpPolicy<4, 2>), DFX_ARG(InterpPolicy<4, 1>), DFX_ARG(InterpPolicy1<4, 1>))
}  // namespace dfx
