// split-in-kernel GEMM, arithmetic f32_bf16x6
#include "gemm_bf16x_impl.h"
GRAPPA_BF16X_MODE_FUNCS(X6, x6)
