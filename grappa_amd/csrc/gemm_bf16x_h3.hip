// split-in-kernel GEMM, arithmetic f32_f16x3 (the default)
#include "gemm_bf16x_impl.h"
GRAPPA_BF16X_MODE_FUNCS(H3, h3)
