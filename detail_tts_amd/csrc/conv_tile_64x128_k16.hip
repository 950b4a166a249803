#include "conv_gemm_kernel.h"
namespace dtts {
DTTS_INSTANTIATE_CONV_TILE(64, 128, 2, 2, 16, false)
}  // namespace dtts
