// split-in-kernel GEMM, arithmetic bf16x3
#include "gemm_bf16x_impl.h"
GRAPPA_BF16X_MODE_FUNCS(X3, x3)
