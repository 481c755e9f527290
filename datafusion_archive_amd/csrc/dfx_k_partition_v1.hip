// dfx_k_partition_v1.hip -- pass 1 of the partitioned GROUP BY for one row-source policy: Static SigKeySum (key + SUM, no predicate: BASELINE config 3).
#include "dfx_k_partition_ws_inl.hpp"
namespace dfx {
DFX_PARTITION_VARIANT_WS(1, DFX_ARG(StaticPolicy<2, 4, SigKeySum>), DFX_ARG(StaticPolicy<2, 4, SigKeySum>), DFX_ARG(StaticPolicy<2, 4, SigKeySum>), DFX_ARG(StaticPolicy<2, 4, SigKeySum>), DFX_ARG(StaticPolicy<2, 8, SigKeySum>))
}  // namespace dfx
