// split-in-kernel GEMM, arithmetic f32_bf16x9
#include "gemm_bf16x_impl.h"
GRAPPA_BF16X_MODE_FUNCS(X9, x9)
