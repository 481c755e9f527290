// dfx_k_partition_v8.hip -- pass 1 of the partitioned GROUP BY for one row-source policy: Static SigKeyAffSumPred2F64 (the headline's shape with SUM(v <+ - *> literal) as the argument).
#include "dfx_k_partition_ws_inl.hpp"
namespace dfx {
DFX_PARTITION_VARIANT_WS(8, DFX_ARG(StaticPolicy<2, 4, SigKeyAffSumPred2F64>), DFX_ARG(StaticPolicy<2, 4, SigKeyAffSumPred2F64>), DFX_ARG(StaticPolicy<2, 4, SigKeyAffSumPred2F64>), DFX_ARG(StaticPolicy<2, 4, SigKeyAffSumPred2F64>), DFX_ARG(StaticPolicy<2, 8, SigKeyAffSumPred2F64>))
}  // namespace dfx
