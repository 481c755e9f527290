// dfx_k_partition_v10.hip -- pass 1 of the partitioned GROUP BY for one row-source policy: PlanPolicy (the scan plan: range tests on value images, plan words in vector registers), <= 2 columns, general form (4-byte columns, validity bitmaps).
#include "dfx_k_partition_ws_inl.hpp"
namespace dfx {
DFX_PARTITION_VARIANT_WS(10, DFX_ARG(PlanPolicyN<2, 2, true>), DFX_ARG(PlanPolicyN<2, 2, true>), DFX_ARG(PlanPolicy1<2, 2, true>), DFX_ARG(PlanPolicy1<2, 4, true>))
}  // namespace dfx
