// Dispatch of the split-in-kernel GEMM (gemm_bf16x_impl.h) over its arithmetics; each lives in a translation unit of its own
// (gemm_bf16x_<mode>.hip).
#include "gemm_common.h"

using namespace grappa_gemm;

#define GRAPPA_DECL(NAME)                                                                   \
    int grappa_bf16x_launch_##NAME(hipStream_t st, GemmParams& p, bool vec_kcontig);        \
    int grappa_bf16x_launch_group4_##NAME(hipStream_t st, const GemmGroup4& g, bool b_kcontig); \
    int grappa_bf16x_launch_grouped_##NAME(hipStream_t st, const GemmParams* d_ps, const int* d_wg_begin, int nprob, int total_wgs, bool vec, int psrc)
GRAPPA_DECL(x9);
GRAPPA_DECL(x6);
GRAPPA_DECL(x3);
GRAPPA_DECL(x1);
GRAPPA_DECL(h3);
#undef GRAPPA_DECL

// grouped weight-gradient products (layout a_kcontig = b_kcontig = 0, tile 256 x 128): called by grappa_gemm_f32_grouped
// psrc: bit 0 / bit 1 = A / B of every product in the pair format (ABI 8, F32_F16X3 only)
int grappa_launch_gemm_bf16x_grouped(hipStream_t st, const GemmParams* d_ps, const int* d_wg_begin, int nprob, int total_wgs, int precision, bool vec, int psrc) {
    switch (precision) {
        case GRAPPA_GEMM_F32_BF16X9: return grappa_bf16x_launch_grouped_x9(st, d_ps, d_wg_begin, nprob, total_wgs, vec, psrc);
        case GRAPPA_GEMM_F32_BF16X6: return grappa_bf16x_launch_grouped_x6(st, d_ps, d_wg_begin, nprob, total_wgs, vec, psrc);
        case GRAPPA_GEMM_BF16X3: return grappa_bf16x_launch_grouped_x3(st, d_ps, d_wg_begin, nprob, total_wgs, vec, psrc);
        case GRAPPA_GEMM_BF16: return grappa_bf16x_launch_grouped_x1(st, d_ps, d_wg_begin, nprob, total_wgs, vec, psrc);
        case GRAPPA_GEMM_F32_F16X3: return grappa_bf16x_launch_grouped_h3(st, d_ps, d_wg_begin, nprob, total_wgs, vec, psrc);
        default: return GRAPPA_ERR_ARG;
    }
}

// called by grappa_gemm_f32 (gemm_f32.hip) for precision != GRAPPA_GEMM_F32_MFMA; (p.bm, p.bn) is 256x128 or 128x128
int grappa_launch_gemm_bf16x(hipStream_t st, GemmParams& p, int precision, bool vec_kcontig) {
    switch (precision) {
        case GRAPPA_GEMM_F32_BF16X9: return grappa_bf16x_launch_x9(st, p, vec_kcontig);
        case GRAPPA_GEMM_F32_BF16X6: return grappa_bf16x_launch_x6(st, p, vec_kcontig);
        case GRAPPA_GEMM_BF16X3: return grappa_bf16x_launch_x3(st, p, vec_kcontig);
        case GRAPPA_GEMM_BF16: return grappa_bf16x_launch_x1(st, p, vec_kcontig);
        case GRAPPA_GEMM_F32_F16X3: return grappa_bf16x_launch_h3(st, p, vec_kcontig);
        default: return GRAPPA_ERR_ARG;
    }
}


// up to four forward / input-gradient products of one layout in one launch (grappa_gemm_f32_group), fp32 operands, tile 256 x 128
int grappa_launch_gemm_bf16x_group4(hipStream_t st, const GemmGroup4& g, int precision, bool b_kcontig) {
    switch (precision) {
        case GRAPPA_GEMM_F32_BF16X9: return grappa_bf16x_launch_group4_x9(st, g, b_kcontig);
        case GRAPPA_GEMM_F32_BF16X6: return grappa_bf16x_launch_group4_x6(st, g, b_kcontig);
        case GRAPPA_GEMM_BF16X3: return grappa_bf16x_launch_group4_x3(st, g, b_kcontig);
        case GRAPPA_GEMM_BF16: return grappa_bf16x_launch_group4_x1(st, g, b_kcontig);
        case GRAPPA_GEMM_F32_F16X3: return grappa_bf16x_launch_group4_h3(st, g, b_kcontig);
        default: return GRAPPA_ERR_ARG;
    }
}
