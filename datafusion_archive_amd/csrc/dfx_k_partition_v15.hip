// dfx_k_partition_v15.hip -- pass 1 of the partitioned GROUP BY for one row-source policy: PlanPolicy (the scan plan: range tests on
// value images, plan words in vector registers), <= 4 columns, GENK = 1 (4-byte: bit 0 4-byte columns widened, bit 1 validity bitmaps).
#include "dfx_k_partition_ws_inl.hpp"
namespace dfx {
// (+ the PAIR flavour: two aggregates of different operands routed by ONE scan -- PTF_PAIR, dfx_device.hpp)
DFX_PARTITION_VARIANT_WS_PAIR(15, DFX_ARG(PlanPolicyN<4, 1, 1>), DFX_ARG(PlanPolicyN<4, 1, 1>), DFX_ARG(PlanPolicy1<4, 1, 1>), DFX_ARG(PlanPolicy1<4, 2, 1>), DFX_ARG(PlanPolicy1<4, 2, 1>), DFX_ARG(PlanPolicyN<4, 2, 1>))
}  // namespace dfx
