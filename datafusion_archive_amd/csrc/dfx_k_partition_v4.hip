// dfx_k_partition_v4.hip -- pass 1 of the partitioned GROUP BY for one row-source policy: FastPolicy, <= 4 columns.
#include "dfx_k_partition_inl.hpp"
namespace dfx {
DFX_PARTITION_VARIANT(4, DFX_ARG(FastPolicy<4, 2>), DFX_ARG(FastPolicy<4, 2>), DFX_ARG(FastPolicy1<4, 2>))
}  // namespace dfx
