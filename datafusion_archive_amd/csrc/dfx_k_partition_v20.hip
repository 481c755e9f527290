// dfx_k_partition_v20.hip -- pass 1 of the partitioned GROUP BY for one row-source policy: PlanPolicy (the scan plan: range tests on
// value images, plan words in vector registers), <= 4 columns, GENK = 6 (4-byte-key+bitmaps: bit 1 validity bitmaps, bit 2 the key is the only 4-byte column).
// (the multi-value flavours of this unit are never launched: bit 2 belongs to the one-key, one-value binding)
#include "dfx_k_partition_ws_inl.hpp"
namespace dfx {
DFX_PARTITION_VARIANT_WS(20, DFX_ARG(PlanPolicy1<4, 1, 6>), DFX_ARG(PlanPolicy1<4, 1, 6>), DFX_ARG(PlanPolicy1<4, 1, 6>), DFX_ARG(PlanPolicy1<4, 2, 6>), DFX_ARG(PlanPolicy1<4, 2, 6>))
}  // namespace dfx
