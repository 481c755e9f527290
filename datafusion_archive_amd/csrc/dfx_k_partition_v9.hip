// dfx_k_partition_v9.hip -- pass 1 of the partitioned GROUP BY for one row-source policy: PlanPolicy (the scan plan: range tests on
// value images, plan words in vector registers), <= 2 columns, GENK = 0 (8-byte-null-free: bit 0 4-byte columns widened, bit 1 validity bitmaps).
#include "dfx_k_partition_ws_inl.hpp"
namespace dfx {
DFX_PARTITION_VARIANT_WS(9, DFX_ARG(PlanPolicyN<2, 2, 0>), DFX_ARG(PlanPolicyN<2, 2, 0>), DFX_ARG(PlanPolicy1<2, 2, 0>), DFX_ARG(PlanPolicy1<2, 4, 0>), DFX_ARG(PlanPolicy1<2, 4, 0>))
}  // namespace dfx
