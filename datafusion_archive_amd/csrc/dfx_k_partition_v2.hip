// dfx_k_partition_v2.hip -- pass 1 of the partitioned GROUP BY for one row-source policy: FastPolicy, <= 2 columns.
#include "dfx_k_partition_inl.hpp"
namespace dfx {
#ifndef DFX_V2_UN
#define DFX_V2_UN 4  // row groups per trip of the one-value flavour (A/B: tools/build_variant.py)
#endif
DFX_PARTITION_VARIANT(2, DFX_ARG(FastPolicy<2, 4>), DFX_ARG(FastPolicy<2, 2>), DFX_ARG(FastPolicy1<2, DFX_V2_UN>))
}  // namespace dfx
