// dfx_k_partition_v4.hip -- pass 1 of the partitioned GROUP BY for one row-source policy: FastPolicy, <= 4 columns.
#include "dfx_k_partition_ws_inl.hpp"
namespace dfx {
DFX_PARTITION_VARIANT_WS(4, false, DFX_ARG(FastPolicy<4, 2>), DFX_ARG(FastPolicy<4, 2>), DFX_ARG(FastPolicy1<4, 2>), DFX_ARG(FastPolicy1<4, 2>))
}  // namespace dfx
