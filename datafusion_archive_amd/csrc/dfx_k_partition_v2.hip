// dfx_k_partition_v2.hip -- pass 1 of the partitioned GROUP BY for one row-source policy: FastPolicy, <= 2 columns.
#include "dfx_k_partition_inl.hpp"
namespace dfx {
DFX_PARTITION_VARIANT(2, DFX_ARG(FastPolicy<2, 4>), DFX_ARG(FastPolicy<2, 2>), DFX_ARG(FastPolicy1<2, 4>))
}  // namespace dfx
