// dfx_k_partition_v6.hip -- pass 1 of the partitioned GROUP BY for one row-source policy: FastPolicy, <= 8 columns.
#include "dfx_k_partition_inl.hpp"
namespace dfx {
DFX_PARTITION_VARIANT(6, DFX_ARG(FastPolicy<8, 2>), DFX_ARG(FastPolicy<8, 1>), DFX_ARG(FastPolicy1<8, 1>))
}  // namespace dfx
