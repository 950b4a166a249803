#include "conv_gemm_kernel.h"
namespace dtts {
DTTS_INSTANTIATE_CONV_TILE(128, 128, 2, 2, 32, false)
}  // namespace dtts
