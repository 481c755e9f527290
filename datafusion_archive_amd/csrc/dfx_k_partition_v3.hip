// dfx_k_partition_v3.hip -- pass 1 of the partitioned GROUP BY for one row-source policy: InterpPolicy, <= 2 columns.
#include "dfx_k_partition_inl.hpp"
namespace dfx {
DFX_PARTITION_VARIANT(3, DFX_ARG(InterpPolicy<2, 2>), DFX_ARG(InterpPolicy<2, 1>), DFX_ARG(InterpPolicy1<2, 1>))
}  // namespace dfx
