// split-in-kernel GEMM, arithmetic bf16
#include "gemm_bf16x_impl.h"
GRAPPA_BF16X_MODE_FUNCS(X1, x1)
