// dfx_k_partition_v17.hip -- pass 1 of the partitioned GROUP BY for one row-source policy: PlanPolicy (the scan plan: range tests on
// value images, plan words in vector registers), <= 2 columns, GENK = 4 (4-byte-key: bit 1 validity bitmaps, bit 2 the key is the only 4-byte column).
// (the multi-value flavours of this unit are never launched: bit 2 belongs to the one-key, one-value binding)
// Eight row groups per trip in the wave-specialised flavour: 12 bytes per row instead of 16 leave this kernel below the traffic
// ceiling the 8-byte kernels sit at, and it is the bytes in flight that bound it (431 -> 407 us per 2^27-row launch).
#include "dfx_k_partition_ws_inl.hpp"
namespace dfx {
DFX_PARTITION_VARIANT_WS(17, DFX_ARG(PlanPolicy1<2, 2, 4>), DFX_ARG(PlanPolicy1<2, 2, 4>), DFX_ARG(PlanPolicy1<2, 2, 4>), DFX_ARG(PlanPolicy1<2, 8, 4>), DFX_ARG(PlanPolicy1<2, 4, 4>))
}  // namespace dfx
