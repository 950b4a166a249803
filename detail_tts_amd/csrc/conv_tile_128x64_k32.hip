#include "conv_gemm_kernel.h"
namespace dtts {
DTTS_INSTANTIATE_CONV_TILE(128, 64, 2, 2, 32, false)
}  // namespace dtts
