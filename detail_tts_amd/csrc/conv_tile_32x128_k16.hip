#include "conv_gemm_kernel.h"
namespace dtts {
DTTS_INSTANTIATE_CONV_TILE(32, 128, 1, 4, 16, false)
}  // namespace dtts
