// dfx_k_partition_v23.hip -- pass 1 of the partitioned GROUP BY for one row-source policy: PlanPolicy (the scan plan: range tests on
// value images, plan words in vector registers), exactly 3 columns (key, routed value, one more predicate column -- e.g.
// SUM(w) WHERE v ... GROUP BY k, or the second scan of SUM(v), MIN(w)), GENK = 2 (bitmaps: bit 0 4-byte columns widened, bit 1 validity bitmaps).
#include "dfx_k_partition_ws_inl.hpp"
namespace dfx {
// (+ the PAIR flavour: two aggregates of different operands routed by ONE scan -- PTF_PAIR, dfx_device.hpp)
DFX_PARTITION_VARIANT_WS_PAIR(23, DFX_ARG(PlanPolicyN<3, 1, 2>), DFX_ARG(PlanPolicyN<3, 1, 2>), DFX_ARG(PlanPolicy1<3, 1, 2>), DFX_ARG(PlanPolicy1<3, 3, 2>), DFX_ARG(PlanPolicy1<3, 3, 2>), DFX_ARG(PlanPolicyN<3, 3, 2>))
}  // namespace dfx
