// dfx_k_partition_v9.hip -- pass 1 of the partitioned GROUP BY for one row-source policy: PlanPolicy (the scan plan: range tests on value images, plan words in vector registers), <= 2 columns, 8-byte null-free columns.
#include "dfx_k_partition_ws_inl.hpp"
namespace dfx {
DFX_PARTITION_VARIANT_WS(9, DFX_ARG(PlanPolicyN<2, 2, false>), DFX_ARG(PlanPolicyN<2, 2, false>), DFX_ARG(PlanPolicy1<2, 2, false>), DFX_ARG(PlanPolicy1<2, 4, false>))
}  // namespace dfx
