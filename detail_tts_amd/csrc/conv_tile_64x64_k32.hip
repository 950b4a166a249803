#include "conv_gemm_kernel.h"
namespace dtts {
DTTS_INSTANTIATE_CONV_TILE(64, 64, 2, 2, 32, true)
}  // namespace dtts
