// dfx_k_partition_v0.hip -- pass 1 of the partitioned GROUP BY for one row-source policy: Static SigKeySumPred2F64 (key + SUM under a two-sided f64 predicate: the headline query).
#include "dfx_k_partition_ws_inl.hpp"
namespace dfx {
DFX_PARTITION_VARIANT_WS(0, DFX_ARG(StaticPolicy<2, 4, SigKeySumPred2F64>), DFX_ARG(StaticPolicy<2, 4, SigKeySumPred2F64>), DFX_ARG(StaticPolicy<2, 4, SigKeySumPred2F64>), DFX_ARG(StaticPolicy<2, 4, SigKeySumPred2F64>), DFX_ARG(StaticPolicy<2, 8, SigKeySumPred2F64>))
}  // namespace dfx
