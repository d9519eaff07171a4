/* grappa_hip.h -- C ABI of libgrappa_hip.so: the MI355X (gfx950) kernels of the Grappa hot path.
 *
 * The reference (hits-mbm-dev/grappa) has no native layer: every function below replaces a piece of
 * Python that today expands into torch/DGL device kernels.  Each entry point cites the reference
 * lines (relative to /root/reference/src/grappa/) whose arithmetic it computes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc'ed / torch.cuda tensor storage) unless named h_*;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises;
 *   - no allocation, no mutable global state, re-entrant per stream and per host thread: every option of a call travels in its arguments
 *     (ABI 10 removed the four process-wide setters: plan override, tail launches, split-K reduction, dropout salt); scratch comes from the
 *     caller (`ws`, `ws_bytes`; sizes from the *_workspace_bytes() queries).  Environment variables read once at first use (GRAPPA_PLAN_TAILS,
 *     GRAPPA_PAIRS_TILE ...) only choose DEFAULTS and never change afterwards;
 *   - return 0 on success, a negative GRAPPA_ERR_* otherwise (never throws / aborts);
 *   - fp32 activations, int32 indices, row-major, leading dimensions in ELEMENTS.
 */
#ifndef GRAPPA_HIP_H
#define GRAPPA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GRAPPA_OK 0
#define GRAPPA_ERR_ARG (-1)        /* unsupported shape / null pointer / bad enum */
#define GRAPPA_ERR_LAUNCH (-2)     /* hipGetLastError() != hipSuccess after the launch */
#define GRAPPA_ERR_WORKSPACE (-3)  /* ws_bytes too small */

#define GRAPPA_ABI_VERSION 11
int grappa_abi_version(void);
/* name of the offload arch the library was compiled for ("gfx950") */
const char* grappa_build_arch(void);
/* kernels this library has launched in this process since the last reset (reset != 0: returns the count and clears it); measurement only */
long long grappa_launch_count(int reset);

/* ------------------------------------------------------------------------------------------------
 * Dense blocks.  C = epilogue(opA(A) * opB(B)); the arithmetic of the product is grappa_gemm_desc.precision (below): the engine's default is
 * F32_F16X3 (fp32 operands as two fp16 pieces, three partial products on v_mfma_f32_32x32x16_f16), the native fp32 matrix instruction
 * (v_mfma_f32_32x32x2_f32) serves products with M or N <= 32 and precision F32_MFMA.
 *   A(m,k) = a_kcontig ? A[m*lda + k] : A[k*lda + m]
 *   B(n,k) = b_kcontig ? B[n*ldb + k] : B[k*ldb + n]
 *   forward  Y = X W^T      : a_kcontig=1 (X[M,K]),  b_kcontig=1 (W[N,K])   nn.Linear
 *   dgrad    dX = dY W      : a_kcontig=1 (dY[M,N']), b_kcontig=0 (W[N',K'])
 *   wgrad    dW = dY^T X    : a_kcontig=0 (dY[tok,N']), b_kcontig=0 (X[tok,K'])
 * epilogue, in this order, per element v = acc(m,n):
 *   v += pre[m,n]                          (pre != NULL: pre-activation addend, e.g. a second GEMM's result)
 *   v += bias[n]                           (bias != NULL)
 *   v  = elu(v)                            (act == GRAPPA_ACT_ELU)
 *   v *= (aux[m,n] > 0 ? 1 : aux[m,n]+1)   (aux != NULL: ELU'(z) from the saved ELU output)
 *   if (C2) { C[m,n] = v; }                (pre-dropout copy kept for the backward pass)
 *   v  = keep(seed, m*N+n) ? v/(1-p) : 0   (drop_p > 0; counter based, see grappa_dropout_keep)
 *   v += res[m,n]                          (res != NULL)
 *   v += OUT[m,n]                          (accumulate != 0)
 *   OUT[m,n] = v                           OUT = C2 ? C2 : C
 * Replaces: torch.nn.Linear / ELU / Dropout / residual adds in models/graph_attention.py:98-101,
 * :125-127, :261-272, :286-308; models/network_utils.py:44-54, :112-126 (in_proj/out_proj of
 * nn.MultiheadAttention); models/interaction_parameters.py:148-161; perm_equiv_transformer.py:231-264.
 */
#define GRAPPA_ACT_NONE 0
#define GRAPPA_ACT_ELU 1

/* Arithmetic of the product (grappa_gemm_desc.precision).  Inputs, outputs, epilogue and accumulation are fp32 in every mode.
 *   F32_MFMA    native fp32 matrix instruction (157 TFLOP/s peak; the value 0 of a zeroed descriptor -- the Python engine asks for F32_F16X3)
 *   F32_BF16X9  each fp32 operand split exactly into 3 bf16 pieces, all 9 partial products on the bf16 matrix cores: every
 *               partial product is exact, the result differs from an fp32 FMA chain only by accumulation order
 *   F32_BF16X6  the 6 largest partial products (drops terms <= 2^-24 |a||b|): fp32-grade, ~2 ulp per product
 *   BF16X3      2 pieces / 3 products (~2^-16 relative);  BF16: operands rounded to bf16 (the "bf16" trainer precision of
 *               the reference's configs, experiment/trainrun.py / Lightning `precision`)
 *   F32_F16X3   (ABI 4) each fp32 operand, scaled by a power of two per row of the product so that its largest magnitude lands in
 *               [2^14, 2^15), is split into 2 fp16 pieces (11 + 11 significant bits and a sign: a - hi - lo <= 2^-24 |a|); the 3
 *               products hi*hi + hi*lo + lo*hi run on the fp16 matrix cores (every product exact in fp32), the dropped lo*lo is
 *               <= 2^-24 |a||b|: fp32-grade like F32_BF16X6 at half the matrix instructions.  Needs a_amax / b_amax (below): the
 *               largest |element| of every row of the product's A (M values) and B (N values), from grappa_amax_f32.  Elements
 *               more than 2^16 below their row's maximum lose relative (never absolute: <= 2^-39 of the maximum) precision.
 * Shapes with M <= 32 or N <= 32 always take the native fp32 path. */
#define GRAPPA_GEMM_F32_MFMA 0
#define GRAPPA_GEMM_F32_BF16X9 1
#define GRAPPA_GEMM_F32_BF16X6 2
#define GRAPPA_GEMM_BF16X3 3
#define GRAPPA_GEMM_BF16 4
#define GRAPPA_GEMM_F32_F16X3 5

typedef struct grappa_gemm_desc {
    int M, N, K;
    int a_kcontig, b_kcontig;
    const float* A; int lda;
    const float* B; int ldb;
    float* C; int ldc;
    float* C2; int ldc2;          /* optional second output (see above) */
    const float* bias;            /* [N] or NULL */
    const float* res; int ldres;  /* [M,N] or NULL */
    const float* aux; int ldaux;  /* [M,N] or NULL */
    const float* pre; int ldpre;  /* [M,N] or NULL */
    float* a_colsum;              /* [M] or NULL; a_kcontig == 0 only: a_colsum[m] += sum_k A(m,k)  (bias gradient of a wgrad GEMM) */
    int act;
    float drop_p;
    uint64_t drop_seed;
    int accumulate;
    int precision;                /* GRAPPA_GEMM_* */
    /* ---- plane format (ABI 3; everything below may be 0 / NULL).  A matrix X "in planes" is three bf16 matrices P0, P1, P2 of
     * X's shape, P0 = bf16(X), P1 = bf16(X - P0), P2 = bf16(X - P0 - P1), X == P0 + P1 + P2 exactly; plane p starts
     * `*_plane_stride` ELEMENTS after plane 0, leading dimensions are in bf16 elements and multiples of 8, bases 16-byte aligned.
     * a_planes / b_planes != 0: A / B point to plane 0 (the pointer types above are then nominal).  Supported: b_planes alone
     * ("weight planes": A = fp32 activations [M][K] with K % 32 == 0, a_kcontig = b_kcontig = 1 -- forward with the planes of W,
     * dgrad with the planes of W^T; results are bit-identical to the split-in-kernel product) or both (a_kcontig = b_kcontig = 1,
     * or both 0 = the wgrad layout).  K-contiguous planes must be zero beyond K up to the next multiple of 32 (ld >= round_up(K, 32));
     * k-major operands need no padding (rows beyond K are taken from a page of zeros); rows * ld * element size < 2^32; M, N > 32.  The operands are
     * split ONCE by their producer (grappa_split_planes_f32, Cp below) instead of by every GEMM that reads them. */
    int a_planes, b_planes;
    size_t a_plane_stride, b_plane_stride;
    uint16_t* Cp; int ldcp; size_t cp_plane_stride;               /* planes of the FINAL value (what OUT receives); C may be NULL then */
    const uint16_t* resp; int ldresp; size_t resp_plane_stride;   /* residual given in planes (instead of res) */
    const uint16_t* auxp; int ldauxp; size_t auxp_plane_stride;   /* saved ELU output given in planes (instead of aux) */
    /* number of planes behind Cp / resp / auxp: 3 (0 means 3) = the exact fp32 split; 1 = a plain bf16 tensor (the bf16 storage
     * configuration: activations are kept in bf16 in HBM, precision GRAPPA_GEMM_BF16 reads plane 0 of both operands) */
    int cp_nplanes, resp_nplanes, auxp_nplanes;                   /* (Cp / C1p / resp / auxp are not available with F32_MFMA or M, N <= 32) */
    uint16_t* C1p; int ldc1p;                                     /* bf16 copy of the value C receives when C2 is used (before dropout / residual) */
    /* ---- ABI 4, precision F32_F16X3 only: bit patterns of max_k |A(m, k)| (M values) and max_k |B(n, k)| (N values); fp32 operands.
     * amax_bcast bit 0 / bit 1: a_amax / b_amax is ONE value for all rows (an upper bound of the whole operand; the weight-gradient
     * product, whose reduction runs over the tokens, takes max over the token maxima: elements more than 2^16 below the tensor's
     * largest lose relative precision gradually, absolute error <= 2^-39 of that largest) */
    const uint32_t* a_amax; const uint32_t* b_amax; int amax_bcast;
    /* any precision, optional: receives max_n |OUT(m, n)| (M values, fp32 bit patterns) of the final fp32 output, for a following
     * F32_F16X3 product that reads OUT as its A operand.  The row epilogue leaves per-segment maxima in the workspace (behind the
     * split-K slabs: grappa_gemm_f32_workspace_bytes includes them) and one small launch combines them; not with the grouped entry */
    uint32_t* out_amax;
    /* ---- ABI 5, the PAIR format: an fp32 matrix X[R][C] as fp16 (HI, LO) pairs plus one fp32 bit pattern per row, amax_r >= max_c
     * |X[r][c]| (the row's largest magnitude or an upper bound of it):
     *     s_r = 141 - exponent_field(amax_r),  HI = f16(X * 2^s_r),  LO = f16(X * 2^s_r - HI),  X = (HI + LO) * 2^-s_r
     * i.e. exactly the operand precision F32_F16X3 builds from fp32 rows inside every workgroup, built once by the tensor's producer
     * instead (grappa_split_pairs_f32; the *_pairs_* producers below) at the same 4 bytes per element.  Layout: a row is a sequence
     * of blocks of 16 consecutive k, each [16 x HI | 16 x LO] (one 64-byte granule): element (r, k) has HI at fp16 index
     * r * ld + 32 * (k / 16) + k % 16 and LO 16 further; ld (fp16 elements) >= 2 * round_up(C, 32) and a multiple of 8, base 16-byte
     * aligned, zeros beyond C up to the next multiple of 32.  precision F32_F16X3 with a_planes != 0 and b_planes != 0
     * (a_kcontig = b_kcontig = 1: forward with the pairs of W, dgrad with the pairs of W^T; lda / ldb = the rows' ld, plane strides
     * unused) reads both operands as pairs by LDS-DMA, a_amax / b_amax being the rows' amax_r; the result is bit-identical to the
     * fp32-operand F32_F16X3 product of the same scales and K split.  M, N > 32; rows * ld * 2 < 2^32.
     * b_planes != 0 alone ("weight pairs": A = fp32 activations [M][K], K % 16 == 0, lda % 4 == 0, 16-byte aligned, a_kcontig =
     * b_kcontig = 1) takes A as it is -- raw fp32 rows by LDS-DMA, each wavefront splits the fragments of its own 64 rows -- and gives
     * the same bits as the all-pairs product.  a_planes != 0 alone is refused. */
    /* ---- ABI 7: the residual given as the INPUT of a LayerNorm.  res_ln_mean != NULL: `res` holds x, the rows BEFORE normalisation, and the
     * epilogue adds LayerNorm(x)(m, n) = fma((x(m, n) - mean[m]) * rstd[m], gamma[n], beta[n]) -- the very bits grappa_layernorm_fwd_*
     * writes -- so that a producer that hands its normalised rows on in the pair format need not write them in fp32 as well (inference).
     * fp32 `res` (not resp), N % 4 == 0, gamma / beta 16-byte aligned, the split kernels only (M, N > 32, precision other than F32_MFMA; fp32
     * operands or both in the pair format): every epilogue form of those (straight-line classes, the generic row walk with C2 / activation,
     * the split-K reduction) adds the recomputed rows; anything else is refused (GRAPPA_ERR_ARG). */
    const float* res_ln_mean; const float* res_ln_rstd; const float* res_ln_gamma; const float* res_ln_beta;
    /* ---- ABI 8: operands of a WEIGHT-GRADIENT product (a_kcontig = b_kcontig = 0, precision F32_F16X3, grappa_gemm_f32_grouped only) in
     * the pair format.  a_planes != 0: A is the [K tokens][M features] matrix as its producer wrote it for the forward / input-gradient
     * products -- pairs, token row k scaled by its own maximum a_rowmax[k] (K values) -- lda in fp16 elements (>= 2 * M, % 8 == 0), M % 32
     * == 0, 16-byte aligned; a_amax (with amax_bcast bit 0) = the maximum over the token maxima.  The kernel moves every token row onto the
     * tensor's scale with exact fp16 multiplications by powers of two (the halves a fresh split under that scale would give, up to the
     * rounding of fp16 denormals).  b_planes / b_rowmax / bit 1: the same for B ([K tokens][N features]).  One operand may stay fp32.
     * The products of one grouped call may mix formats (one launch either way; a group of one combination runs the kernel specialised for it). */
    const uint32_t* a_rowmax; const uint32_t* b_rowmax;
    /* ---- ABI 8: row maxima of OUT left as the per-segment partials the epilogue writes anyway, combined by the CONSUMER.
     * out_amax_parts != NULL (instead of out_amax; the split kernels only: M, N > 32, precision other than F32_MFMA): receives
     * nseg = ceil(N / 32) arrays of M maxima, parts[seg * M + m].  A following product whose A operand is OUT (forward / input-gradient
     * layout, fp32 operands) passes a_amax = parts and a_amax_nseg = nseg and takes the maximum over the segments itself;
     * grappa_amax_reduce over all nseg * M values gives the whole-tensor maximum of a weight-gradient product. */
    uint32_t* out_amax_parts; int a_amax_nseg;
    /* ---- ABI 10: the options that used to be process-wide setters, per call (all 0 / NULL = the library's defaults).
     * plan_cfg: tuning and tests -- force the tile configuration index plan_cfg - 1 (0: the plan's own choice); plan_nsplit: force the split-K
     * factor (0: own choice).  plan_tail: the "tail launch" (when the tile grid is 256 q + rem workgroups, the rem tiles run as a second launch
     * with their K range cut): 0 = allowed unless the environment says GRAPPA_PLAN_TAILS=0, 1 = allowed, 2 = never (a caller that keeps several
     * streams busy: the partial last round then runs beside another stream's kernels), 3 = forced where possible (tests).  Results differ only
     * in the summation order of the tail tiles' K slices.
     * splitk_reduce: where a split-K product of the fp32-operand split kernels sums its K slices -- 0 = default (a launch of the reduction kernel
     * behind the product, unless the environment says GRAPPA_SPLITK_IN_KERNEL=1), 1 = the reduction launch, 2 = inside the product's own launch
     * (the last workgroup of a tile to arrive adds the slabs in the fixed order 0, 1, ...).  Same bits either way.
     * drop_salt: one uint64 in DEVICE memory (or NULL: none) mixed into drop_seed when the kernel runs: seed + word * 0x9E3779B97F4A7C15.  A
     * train step captured in a hipGraph increments the word inside the graph: every replay draws fresh masks while forward and backward of
     * one replay agree.  (The row-wise kernels that draw the same masks take the same pointer as their last argument.) */
    int plan_cfg, plan_nsplit, plan_tail, splitk_reduce;
    const uint64_t* drop_salt;
} grappa_gemm_desc;

/* Largest magnitudes of an fp32 matrix x[R][C] (leading dimension ldx), as fp32 bit patterns: row_amax[r] = max_c |x[r][c]|,
 * col_amax[c] = max_r |x[r][c]| (either may be NULL), one pass over x.  A product's operand uses the array ALONG which it is
 * not reduced: forward x[tok][in] -> row_amax, W[out][in] -> row_amax; dgrad dY[tok][out] -> row_amax, W -> col_amax;
 * wgrad dY -> col_amax, x -> col_amax (or, with amax_bcast, the single maximum of each: grappa_amax_reduce over the row maxima).
 * NaN counts as larger than everything (the product is then NaN / Inf as in fp32).
 * ws: grappa_amax_f32_workspace_bytes(R, C) bytes (partial column maxima), needed only with col_amax. */
size_t grappa_amax_f32_workspace_bytes(int R, int C);
int grappa_amax_f32(void* stream, int R, int C, const float* x, int ldx, uint32_t* row_amax, uint32_t* col_amax, void* ws, size_t ws_bytes);
/* Row and column maxima of many small matrices in ONE launch (the weights of a model, refreshed once per optimiser step): matrix i is
 * handled by workgroup i.  `descs` is a DEVICE array; every matrix needs C % 4 == 0, C <= 2048, ld % 4 == 0 and a 16-byte aligned base
 * (the caller checks: the library cannot read the table). */
typedef struct grappa_amax_item {
    const float* x; int R, C, ld, pad_;
    uint32_t* row_amax; uint32_t* col_amax;
} grappa_amax_item;
int grappa_amax_f32_batched(void* stream, int count, const grappa_amax_item* descs);
/* out[b] = max_i in[b][i], i < n[b], for `count` arrays in one launch per 32 (bit patterns of magnitudes: unsigned order): whole-tensor
 * maxima from row maxima.  `in` and `n` are HOST arrays (of device pointers / lengths). */
int grappa_amax_reduce(void* stream, int count, const uint32_t* const* in, const int* n, uint32_t* out);
/* out[m] = max over seg of parts[seg * M + m]: the row maxima from a product's per-segment partials (grappa_gemm_desc.out_amax_parts), for a
 * consumer that needs them as one array */
int grappa_amax_combine(void* stream, int M, int nseg, const uint32_t* parts, uint32_t* out);

/* fp32 X[R][C] -> plane format: planes[p][r][c] (transpose == 0) or planes[p][c][r] (transpose != 0), p = 0..2, leading
 * dimension ldp, `plane_stride` elements between planes.  Only the R x C (C x R) block is written: padding the GEMM relies on
 * (zeros up to the next multiple of 32 along k) is the caller's (allocate zeroed).  Weights are split once per optimiser step
 * (grappa_amd/optim.py WeightPlanes), both orientations: W for the forward product, W^T for dgrad. */
int grappa_split_planes_f32(void* stream, int R, int C, const float* x, int ldx, uint16_t* planes, int ldp, size_t plane_stride,
                            int transpose);

/* ABI 5: fp32 X[R][C] -> pair format (see grappa_gemm_desc): row r of the output is row r of X (transpose == 0) or column r of X
 * (transpose != 0), leading dimension ldp fp16 elements.  amax: the bit patterns that define the scale of every OUTPUT row (R values,
 * or C values when transposing: grappa_amax_f32's row_amax / col_amax).  Only the elements of X are written; the zero padding along k
 * is the caller's (allocate zeroed).  Weights are split once per optimiser step, both orientations; activations by their producers. */
int grappa_split_pairs_f32(void* stream, int R, int C, const float* x, int ldx, const uint32_t* amax, uint16_t* pairs, int ldp, int transpose);
/* The same for many matrices in ONE launch (the weights of a model, both orientations, once per optimiser step).  `items` lives in DEVICE
 * memory; tile_begin = number of 32 x 32 tiles ((R + 31) / 32 * ((C + 31) / 32)) of the items before this one, total_tiles their sum;
 * amax = the row maxima of X (transpose == 0) or its column maxima (transpose != 0: the pairs of X^T, [C][R]). */
typedef struct grappa_split_pairs_item {
    const float* x; const uint32_t* amax; uint16_t* pairs;
    int R, C, ldx, ldp, transpose, tile_begin;
} grappa_split_pairs_item;
int grappa_split_pairs_f32_batched(void* stream, int count, int total_tiles, const grappa_split_pairs_item* items);

/* workspace of a product: enough for every plan the library may choose for this shape whatever plan_tail / splitk_reduce say (plan_cfg and
 * plan_nsplit = 0); a descriptor that forces a configuration or a split asks grappa_gemm_f32_workspace_bytes_desc */
size_t grappa_gemm_f32_workspace_bytes(int M, int N, int K);
size_t grappa_gemm_f32_workspace_bytes_desc(const grappa_gemm_desc* d);
/* host-only: the tile (tile_m x tile_n x 32) and split-K factor the launcher will use for this shape (default options), and the "tail": when
 * the tile grid is 256*q + rem workgroups, the rem tiles run as a second launch with their K range split tail_nsplit ways */
int grappa_gemm_f32_plan(int M, int N, int K, int precision, int* tile_m, int* tile_n, int* nsplit, int* tail_tiles, int* tail_nsplit);
/* ABI 10: the same for a descriptor's M, N, K, precision and plan options (operands taken as fp32) */
int grappa_gemm_f32_plan_desc(const grappa_gemm_desc* d, int* tile_m, int* tile_n, int* nsplit, int* tail_tiles, int* tail_nsplit);
int grappa_gemm_f32(void* stream, const grappa_gemm_desc* d, void* ws, size_t ws_bytes);

/* Grouped weight gradients: n <= GRAPPA_GEMM_GROUP_MAX independent products in the wgrad layout (a_kcontig = b_kcontig = 0, fp32
 * operands, M, N > 32, one common precision other than F32_MFMA) as ONE grid + one reduction.  A weight gradient alone has 8 - 24
 * output tiles and must cut its K (= tokens) 10 - 32 ways to fill the chip, i.e. write and re-read that many partial tiles per
 * output tile; a backward pass's gradients launched together need 3 - 11 cuts each.  Results per product are those of
 * grappa_gemm_f32 up to the summation order of the K chunks (fixed, reproducible).  The workspace also carries the device copies of
 * the descriptors.  Replaces nothing in the reference (torch.autograd runs one addmm per weight); used by ops._WgradQueue. */
#define GRAPPA_GEMM_GROUP_MAX 16
size_t grappa_gemm_f32_grouped_workspace_bytes(const grappa_gemm_desc* descs, int n);
int grappa_gemm_f32_grouped(void* stream, const grappa_gemm_desc* descs, int n, void* ws, size_t ws_bytes);
/* ABI 8: up to 4 FORWARD or INPUT-GRADIENT products in ONE launch -- the same product of the four writer heads (bond / angle / proper /
 * improper), which at small batches leaves most of the chip idle when launched head by head.  All descriptors: a_kcontig = 1, one
 * b_kcontig, one precision (not F32_MFMA), operands all fp32 or all in the pair format (both operands, b_kcontig = 1), M, N > 32, fp32 C
 * (no bf16 tensors, no a_colsum), operands that allow 16-byte loads; every product keeps its own epilogue (bias, activation, dropout,
 * residual incl. res_ln_*, C2, out_amax).  No split-K.  Anything else: GRAPPA_ERR_ARG (launch the products one by one). */
size_t grappa_gemm_f32_group_workspace_bytes(const grappa_gemm_desc* descs, int n);
int grappa_gemm_f32_group(void* stream, const grappa_gemm_desc* descs, int n, void* ws, size_t ws_bytes);

/* out[n] (+)= sum_m x[m*ldx + n]   (bias gradients) */
size_t grappa_colsum_workspace_bytes(int M, int N);
int grappa_colsum_f32(void* stream, int M, int N, const float* x, int ldx, float* out, int accumulate,
                      void* ws, size_t ws_bytes);

/* dz = dy * keep(seed,idx)/(1-p) * (y ? elu'(y) : 1)   elementwise over an [M,N] view (backward of the
 * act+dropout epilogue).  y == NULL: no activation.  dz may alias dy. */
int grappa_act_dropout_bwd_f32(void* stream, int M, int N, const float* dy, int lddy, const float* y, int ldy,
                               float drop_p, uint64_t drop_seed, float* dz, int lddz, const uint64_t* drop_salt);
/* ABI 4, *_amax_f32 variants of the producers of dense-product operands: the same kernel also writes the largest magnitude of every
 * row of its output (M fp32 bit patterns, as grappa_amax_f32's row_amax) -- the scales of a following F32_F16X3 product, without a
 * pass of their own.  A NULL array gives the plain kernel. */
int grappa_act_dropout_bwd_amax_f32(void* stream, int M, int N, const float* dy, int lddy, const float* y, int ldy,
                                    float drop_p, uint64_t drop_seed, float* dz, int lddz, uint32_t* dz_amax, const uint64_t* drop_salt);
/* (drop_salt, ABI 10: see grappa_gemm_desc.drop_salt; NULL = none)
 * ---- ABI 8: batched row-wise kernels.  The writer heads run layer-locked (one autograd node per transformer layer over all heads): their
 * products go out as one grouped launch (grappa_gemm_f32_group) and their row-wise kernels as ONE launch over up to
 * GRAPPA_ROW_BATCH_MAX independent tensors.  Same arithmetic per tensor as the single-tensor entry points (fp32, row maxima written). */
#define GRAPPA_ROW_BATCH_MAX 4
typedef struct grappa_ln_fwd_item {          /* = grappa_layernorm_fwd_amax_f32's arguments; y_amax may be NULL */
    int M, W; const float* x; int ldx; const float* gamma; const float* beta; float* y; int ldy; float* mean; float* rstd; uint32_t* y_amax;
} grappa_ln_fwd_item;
int grappa_layernorm_fwd_batched_f32(void* stream, const grappa_ln_fwd_item* items, int count);
typedef struct grappa_ln_bwd_item {          /* grappa_layernorm_bwd_amax_f32 with accumulate = 2: `part` receives grappa_layernorm_bwd_partial_rows(M)
                                              * rows of [dgamma | dbeta] partials (2 W floats each) for grappa_colsum_partials_batched; dx_amax may be NULL */
    int M, W; const float* dy; int lddy; const float* x; int ldx; const float* mean; const float* rstd; const float* gamma; float* dx; int lddx;
    float* part; uint32_t* dx_amax;
} grappa_ln_bwd_item;
int grappa_layernorm_bwd_batched_f32(void* stream, const grappa_ln_bwd_item* items, int count);
typedef struct grappa_act_dropout_item {     /* = grappa_act_dropout_bwd_amax_f32's arguments (N % 4 == 0, N <= 2048, 16-byte aligned rows); y may be NULL */
    int M, N; const float* dy; int lddy; const float* y; int ldy; float drop_p; uint64_t drop_seed; float* dz; int lddz; uint32_t* dz_amax;
    const uint64_t* drop_salt;               /* ABI 10 */
} grappa_act_dropout_item;
int grappa_act_dropout_bwd_batched_f32(void* stream, const grappa_act_dropout_item* items, int count);
typedef struct grappa_seqattn_item {         /* = grappa_seqattn_fwd_amax_f32 / grappa_seqattn_bwd_amax_f32's arguments; amax may be NULL */
    int s, T, nheads, dh; const float* qkv; float* out; const float* dout; float* dqkv; uint32_t* amax;
} grappa_seqattn_item;
int grappa_seqattn_fwd_batched_f32(void* stream, const grappa_seqattn_item* items, int count);      /* uses qkv, out, amax (of out) */
int grappa_seqattn_bwd_batched_f32(void* stream, const grappa_seqattn_item* items, int count);      /* uses qkv, dout, dqkv, amax (of dqkv) */
/* ABI 8: the same rows written in the PAIR format (pairs, ldp >= 2 * N fp16 elements, % 8 == 0; N % 32 == 0, N <= 2048; dz_amax = the
 * rows' scales, required); dz (fp32) may be NULL: the products behind -- the input gradient through grappa_gemm_f32's pair operands, the
 * weight gradient through ABI 8's -- read the pairs. */
int grappa_act_dropout_bwd_pairs_f32(void* stream, int M, int N, const float* dy, int lddy, const float* y, int ldy, float drop_p,
                                     uint64_t drop_seed, float* dz, int lddz, uint32_t* dz_amax, uint16_t* pairs, int ldp, const uint64_t* drop_salt);

/* y = x + z elementwise (used to merge gradient branches); y may alias x */
int grappa_add_f32(void* stream, size_t n, const float* x, const float* z, float* y);

/* ------------------------------------------------------------------------------------------------
 * LayerNorm over the last dimension (eps = 1e-5, biased variance), one wavefront per row.
 * W % 4 == 0 and W <= 2048.  Replaces torch.nn.LayerNorm in graph_attention.py:258,:265;
 * network_utils.py:38,:98. */
int grappa_layernorm_fwd_f32(void* stream, int M, int W, const float* x, int ldx, const float* gamma, const float* beta,
                             float* y, int ldy, float* mean, float* rstd);
int grappa_layernorm_fwd_amax_f32(void* stream, int M, int W, const float* x, int ldx, const float* gamma, const float* beta,
                                  float* y, int ldy, float* mean, float* rstd, uint32_t* y_amax);
/* ABI 7: the same rows ALSO written in the pair format (grappa_gemm_desc, ABI 5: the A operand of a following F32_F16X3 product, split once
 * by the kernel that holds the whole row and its maximum): pairs[M][ldp] fp16, ldp >= 2 * W, ldp % 8 == 0, 16-byte aligned; W % 32 == 0.
 * y (fp32; NULL: not written), mean, rstd (NULL: not written) as above and bit-identical to them; y_amax is required (the pairs' scales).
 * Inference: LayerNorm -> product chains without a second pass over the activation and without a split in the product. */
int grappa_layernorm_fwd_pairs_f32(void* stream, int M, int W, const float* x, int ldx, const float* gamma, const float* beta,
                                   float* y, int ldy, float* mean, float* rstd, uint32_t* y_amax, uint16_t* pairs, int ldp);
size_t grappa_layernorm_bwd_workspace_bytes(int M, int W);
/* dx may alias dy.  dgamma/dbeta: accumulate = 1 adds to the existing values, 0 overwrites.  accumulate = 2 defers them: the
 * kernel leaves its per-block partial sums at the start of `ws` -- grappa_layernorm_bwd_partial_rows(M) rows of 2 W floats,
 * [dgamma | dbeta] -- and writes neither vector; the caller keeps that `ws` alive and later hands the partials of many LayerNorms to
 * ONE launch of grappa_colsum_partials_batched (a backward pass has ~50 LayerNorms: 100 small reduction launches otherwise). */
int grappa_layernorm_bwd_partial_rows(int M);
typedef struct {
    const float* part;   /* nrows x n, row-major, contiguous */
    int nrows, n;
    float* out;          /* columns [0, n_first) (all n when out2 is NULL) */
    float* out2;         /* columns [n_first, n), or NULL */
    int n_first;
    int accumulate;      /* != 0: add to the existing values */
} grappa_colsum_item;
/* out (+)= column sums of each item's partial rows, summed in a fixed order (the same bits every run); `items` is a HOST array, any
 * count; two items of one call must not share an output */
int grappa_colsum_partials_batched(void* stream, const grappa_colsum_item* items, int count);
int grappa_layernorm_bwd_f32(void* stream, int M, int W, const float* dy, int lddy, const float* x, int ldx,
                             const float* mean, const float* rstd, const float* gamma,
                             float* dx, int lddx, float* dgamma, float* dbeta, int accumulate,
                             void* ws, size_t ws_bytes);
int grappa_layernorm_bwd_amax_f32(void* stream, int M, int W, const float* dy, int lddy, const float* x, int ldx,
                                  const float* mean, const float* rstd, const float* gamma,
                                  float* dx, int lddx, float* dgamma, float* dbeta, int accumulate,
                                  void* ws, size_t ws_bytes, uint32_t* dx_amax);
/* ABI 9: the same with the dropout backward of the tensor this LayerNorm read fused in -- the reference's layers are
 * y = LN(x) + drop(f(LN(x))) chains (perm_equiv_transformer.py:127-151), so the gradient a LayerNorm backward produces is what the dropout
 * of the layer in front masks next: dz[r][c] = keep(drop_seed, r * W + c) ? dx[r][c] / (1 - drop_p) : 0 (the mask of grappa_act_dropout_bwd_f32
 * on an [M x W] tensor) and dz's row maxima, written while the row is in registers instead of by a launch of its own.  0 < drop_p < 1;
 * dx_amax may be NULL */
int grappa_layernorm_bwd_drop_f32(void* stream, int M, int W, const float* dy, int lddy, const float* x, int ldx,
                                  const float* mean, const float* rstd, const float* gamma,
                                  float* dx, int lddx, float* dgamma, float* dbeta, int accumulate,
                                  void* ws, size_t ws_bytes, uint32_t* dx_amax,
                                  float drop_p, uint64_t drop_seed, float* dz, int lddz, uint32_t* dz_amax, const uint64_t* drop_salt);

/* ------------------------------------------------------------------------------------------------
 * Graph attention message passing (DGL DotGatConv, graph_attention.py:249/:283 -> DGL u_dot_v +
 * edge_softmax + u_mul_e/sum): for every destination v and head h
 *     alpha_uv = softmax_{u in N(v)} <ft_u,ft_v>/sqrt(D),  out_v = sum_u alpha_uv ft_u.
 * CSR by destination: indptr[N+1], indices[E] (source ids).  F = H*D floats per row, D % 4 == 0,
 * D/4 a power of two <= 64, F <= 2048.  alpha[E,H] is written for the backward pass.
 * Backward (dft = dL/dft) uses rev[E] (slot of the reverse edge; the bond graph is symmetric) and
 * delta[N,H] scratch; it gathers -- no atomics, bitwise reproducible. */
int grappa_gat_fwd_f32(void* stream, int N, int E, int H, int D, const int* indptr, const int* indices,
                       const float* ft, float* out, float* alpha);
int grappa_gat_bwd_f32(void* stream, int N, int E, int H, int D, const int* indptr, const int* indices, const int* rev,
                       const float* ft, const float* out, const float* alpha, const float* dout,
                       float* dft, float* delta /* [N,H] scratch */);
/* DGL SAGEConv 'mean' aggregation (graph_attention.py:360-363,:391):
 * out_v = sum_{u in N(v)} x_u * (scale_by_neighbor ? 1/deg(u) : 1/deg(v)).
 * forward: scale_by_neighbor = 0; backward (symmetric graph): scale_by_neighbor = 1 on d(out). */
int grappa_neighbor_mean_f32(void* stream, int N, int F, const int* indptr, const int* indices,
                             const float* x, float* out, int scale_by_neighbor);
/* ABI 5, the bf16 storage configuration of the SAGE block: bf16 rows in (8-byte aligned), out bf16 (out_f32 == 0) or fp32 (16-byte aligned) */
int grappa_neighbor_mean_bf16(void* stream, int N, int F, const int* indptr, const int* indices,
                              const uint16_t* x, void* out, int out_f32, int scale_by_neighbor);

/* ------------------------------------------------------------------------------------------------
 * Input featurisation: sinusoidal encoding of the partial charge, graph_attention.py:428-444.
 * enc[n, 2j] = sin(s f_j), enc[n, 2j+1] = cos(s f_j), s = (clamp(q,lo,hi)+hi)/(hi-lo), f_j = 1e4^(-j/(dim/2)).
 * Written into out[n*ldo + col0 ...]. */
int grappa_charge_encoding_f32(void* stream, int N, const float* q, int dim, float lo, float hi, float* out, int ldo, int col0);

/* ------------------------------------------------------------------------------------------------
 * Tuple (n-body) stage.  Token table layout everywhere: row = pos*T + t  (the reference's
 * (n_seq, n_batch, n_feats) tensors, interaction_parameters.py:173-178).
 * gather : x[pos*T+t, 0:W] = a[idx[t*s+pos], 0:W]; if pe: x[.., W-1] = pe[pos]  (RepProjector gather
 *          + positional-encoding concat, perm_equiv_transformer.py:134-141).
 * gather_bwd: da[n, 0:W] = sum over rows r in inv_rows[inv_ptr[n] : inv_ptr[n+1]] of dx[r, 0:W]
 *          (column W-1 zeroed when has_pe). */
int grappa_tuple_gather_fwd_f32(void* stream, int T, int s, int W, const float* a, int lda, const int* idx,
                                const float* pe, float* x, int ldx);
int grappa_tuple_gather_bwd_f32(void* stream, int N, int W, const int* inv_ptr, const int* inv_rows,
                                const float* dx, int lddx, float* da, int ldda, int has_pe, int accumulate);

/* nn.MultiheadAttention over the s <= 4 tokens of each tuple (network_utils.py:105,:122):
 * qkv[pos*T+t, 3F] = [q|k|v], head h = columns h*dh.. of each third; out[pos*T+t, F].
 * dh % 4 == 0, dh/4 a power of two, F = nheads*dh <= 1024. */
int grappa_seqattn_fwd_f32(void* stream, int s, int T, int nheads, int dh, const float* qkv, float* out);
int grappa_seqattn_bwd_f32(void* stream, int s, int T, int nheads, int dh, const float* qkv, const float* dout, float* dqkv);
/* ABI 4: the same kernels, also writing the largest magnitude of every row of out (s*T values) / of dqkv (s*T values) */
int grappa_seqattn_fwd_amax_f32(void* stream, int s, int T, int nheads, int dh, const float* qkv, float* out, uint32_t* out_amax);
/* ABI 7: the attention output in the pair format only (inference: nothing but the out-projection product reads it): pairs[s*T][ldp],
 * F = nheads * dh <= 512, F % 32 == 0, ldp >= 2 * F; out_amax (s*T values, required) = the rows' scales; values bit-identical to
 * grappa_seqattn_fwd_amax_f32 followed by grappa_split_pairs_f32 */
int grappa_seqattn_fwd_pairs_f32(void* stream, int s, int T, int nheads, int dh, const float* qkv, uint16_t* pairs, int ldp, uint32_t* out_amax);
int grappa_seqattn_bwd_amax_f32(void* stream, int s, int T, int nheads, int dh, const float* qkv, const float* dout, float* dqkv,
                                uint32_t* dqkv_amax);

/* Symmetriser input (perm_equiv_transformer.py:248-261): z[p*T+t, j*F+f] = x[perm[p*s+j]*T+t, f].
 * bwd: dx[i*T+t, f] = sum_p dz[p*T+t, inv_p(i)*F+f].  h_perm is a HOST array (P*s ints). */
int grappa_perm_concat_fwd_f32(void* stream, int s, int T, int F, int P, const int* h_perm, const float* x, float* z);
int grappa_perm_concat_bwd_f32(void* stream, int s, int T, int F, int P, const int* h_perm, const float* dz, float* dx);

/* Output maps (interaction_parameters.py:252-266, :347-362, :536-560; final_layer.py:52,:91-97;
 * network_utils.py:144-145).  o[P*T, ldo] is the symmetriser MLP output for all P permuted copies;
 * c = sum_p o[p*T+t, :].
 *   kind BOND : eq = std_e*(elu(mos_e + c0 - 1)+1)+min_e ; k = std_k*(elu(mos_k + c1 - 1)+1)+min_k
 *   kind ANGLE: eq = max*sigmoid(som*c0)                 ; k as above
 *   kind TORSION: gated ? k_n = c_n*sigmoid(c_{np+n})*k_std[n] : k_n = c_n*k_std[n]+k_mean[n];
 *                 k_n = |k_n| > cutoff ? k_n : 0
 * consts (device, float): BOND  [mos_e,std_e,min_e,mos_k,std_k,min_k];
 *                         ANGLE [som,max,0,mos_k,std_k,min_k];  TORSION [k_std[np] | k_mean[np]]. */
#define GRAPPA_OUT_BOND 0
#define GRAPPA_OUT_ANGLE 1
#define GRAPPA_OUT_TORSION 2
int grappa_param_out_fwd_f32(void* stream, int kind, int T, int P, int n_per, int gated, float cutoff,
                             const float* o, int ldo, const float* consts, float* k, float* eq);
int grappa_param_out_bwd_f32(void* stream, int kind, int T, int P, int n_per, int gated, float cutoff,
                             const float* o, int ldo, const float* consts, const float* dk, const float* deq,
                             float* d_o);
/* ABI 4, learnable_statistics=True (reference models/final_layer.py:21-44, :64-88; interaction_parameters.py:463-470): d_consts[i] =
 * dL/d consts[i] of the same map (6 values for bonds / angles, 2 * n_per for torsions; entries of constants that are buffers in the
 * reference -- min_, max -- are computed but meaningless).  Fixed-order two-stage sum: reproducible. */
size_t grappa_param_out_stats_workspace_bytes(int T);
int grappa_param_out_bwd_stats_f32(void* stream, int kind, int T, int P, int n_per, int gated, float cutoff, const float* o, int ldo,
                                   const float* consts, const float* dk, const float* deq, float* d_consts, void* ws, size_t ws_bytes);

/* ------------------------------------------------------------------------------------------------
 * Molecular-mechanics energy / force and its backward (models/internal_coordinates.py:15-125,:150-210;
 * models/energy.py:8-71,:99-145).  xyz[N,C,3].  Per level: idx[T,s], k (T or T*n_per), eq (T), mol_ptr[B+1].
 *   energy  : E[b,c] = sum of 1/2 k (x-eq)^2 (bonds, angles) + sum_n k_n cos(n phi) (+|k_n| if offset)
 *             per-term totals term_energy[4,B,C]; optional per-tuple energies tuple_e[l] (T_l,C).
 *   gradient: G[a,c,:] = dE/dxyz via the atom->tuple incidence (inc_ptr[N+1], inc_code = t<<4|level<<2|pos).
 *   backward: given gE[B,C] and gG[N,C,3] (either may be NULL): gk, geq per tuple (closed form of the
 *             double backward through autograd.grad(create_graph=True), energy.py:139).
 * conf_mask[B,C] (NULL = all ones) is NOT applied here; dummy conformations are handled by the loss. */
typedef struct grappa_mm_desc {
    int N, C, B;
    const float* xyz;
    int T[4];                 /* n2, n3, n4, n4_improper */
    const int* idx[4];
    const float* k[4];
    const float* eq[4];       /* [2],[3] unused */
    const int* mol_ptr[4];
    int n_per[4];             /* [0],[1] unused */
    int offset_torsion;
    const int* inc_ptr;
    const int* inc_code;
    const int* atom_molptr;   /* [B+1] */
} grappa_mm_desc;

int grappa_mm_energy_fwd_f32(void* stream, const grappa_mm_desc* d, float* energy, float* term_energy,
                             float* const tuple_e[4], float* const tuple_x[4]);
int grappa_mm_gradient_fwd_f32(void* stream, const grappa_mm_desc* d, float* grad);
int grappa_mm_bwd_f32(void* stream, const grappa_mm_desc* d, const float* gE, const float* gG,
                      float* const gk[4], float* const geq[4]);

/* ------------------------------------------------------------------------------------------------
 * MolwiseLoss (training/loss.py:45-167 with utils/graph_utils.py:35-86), one workgroup per molecule:
 *  l_m = wE*mean_c((E-<E>)-(Eref-<Eref>))^2 + wG*mean_{a,c,xyz}(G-Gref)^2   over real conformations
 *  loss_mol[b] = l_m ; gE = d(sum_m l_m * inv_B)/dE ; gG likewise.  is_dummy may be NULL. */
int grappa_loss_ef_fwd_bwd_f32(void* stream, int B, int C, int N, const int* atom_molptr,
                               const float* energy, const float* energy_ref, const float* is_dummy,
                               const float* grad, const float* grad_ref, float wE, float wG, float inv_B,
                               float* loss_mol, float* gE, float* gG);
/* parameter MSE vs classical parameters (NaN references masked) + L2 on torsion k, per molecule:
 *  loss_mol[b] += pw[b] * sum_levels fac_l^2 sum (p-pref)^2 / (#entries) + reg terms; d/dp written to gp[l]
 *  (every entry of gp[l] is overwritten).  levels: 0 n2_k, 1 n2_eq, 2 n3_k, 3 n3_eq, 4 n4_k, 5 n4_improper_k.
 *  ref[l] may be NULL (level skipped in the MSE); width[l] = columns; ref_width[l] = columns of ref. */
typedef struct grappa_ploss_desc {
    int B;
    const int* mol_ptr[6];
    const float* p[6];
    const float* ref[6];
    int width[6];
    int ref_width[6];
    float fac[6];
    float reg[6];             /* L2 prefactor per level (only [4],[5] non-zero in the reference) */
    const float* pw;          /* [B] parameter weight per molecule, or NULL = no MSE term */
    float inv_B;
} grappa_ploss_desc;
int grappa_loss_param_fwd_bwd_f32(void* stream, const grappa_ploss_desc* d, float* loss_mol, float* const gp[6]);

/* Device-side collate (replaces the per-batch host work of data/GraphDataLoader.py:23-73 collate_fn,
 * utils/dgl_utils.py:11-60 batch and :132-171 set_number_confs for a dataset that is resident in HBM).
 * A table is an array of 4-byte elements (int32 / float32; an int64 column counts as two) with `width` elements per row.
 * Workgroup (j, t) copies the rows of batch slot j of table t:
 *   rows  dst_row[j] .. dst_row[j+1]-1 of dst   <-   rows src_row[j] .. of src          (all arrays are DEVICE pointers)
 * and transforms them according to `mode`:
 *   COPY      verbatim (features, charges, reference parameters)
 *   ADD       int32 + p0[j]                       (CSR columns / tuple indices + atom offset, reverse-edge slots + edge offset)
 *   INV_ROWS  local token row pos*p0[j] + t  ->  pos*c0 + p1[j] + t     (p0 = tuples of the molecule, p1 = tuple offset of the
 *             slot, c0 = tuples of the batch at this level; width 1)
 *   INC_CODE  packed incidence code (tuple << 4 | level << 2 | pos) + (p0[4*j + level] << 4)
 *   CONF      conformation selection: a source row holds (p0[j], width) values, a batch row (c0, width);
 *             dst[row][c][e] = src[src_row[j] + (row*p0[j] + p1[j*c0 + c])*width + e]  -- src_row[j] is an ELEMENT offset here;
 *             p1 = selected conformation per (slot, output conformation): sub-sampling and dummy padding alike. */
#define GRAPPA_COLLATE_COPY 0
#define GRAPPA_COLLATE_ADD 1
#define GRAPPA_COLLATE_INV_ROWS 2
#define GRAPPA_COLLATE_INC_CODE 3
#define GRAPPA_COLLATE_CONF 4
typedef struct grappa_collate_desc {
    const int32_t* src;
    int32_t* dst;
    const int64_t* src_row;   /* [B] */
    const int64_t* dst_row;   /* [B+1] */
    const int32_t* p0;
    const int32_t* p1;
    int64_t c0;
    int32_t width;
    int32_t mode;
} grappa_collate_desc;
/* descs_device: the n_tables descriptors in device memory (read by the kernel); descs_host: the same array in host memory
 * (validated before the launch).  One launch, grid (B, n_tables). */
int grappa_collate_batch(void* stream, const grappa_collate_desc* descs_device, const grappa_collate_desc* descs_host, int n_tables, int B);

/* FastEvaluator.step (training/evaluation.py:53-113; get_energies / get_gradients utils/graph_utils.py:35-86), one workgroup
 * per molecule instead of dgl.unbatch + a Python loop:  out[b*4 + {0,1,2,3}] = { sum over real conformations of the squared
 * difference of the per-molecule-centred energies, number of real conformations, sum over atoms x real conformations x xyz of
 * (G - G_ref)^2, atoms x real conformations }.  grad / grad_ref may both be NULL (FastEvaluator(gradients=False)). */
int grappa_eval_se_f32(void* stream, int B, int C, int N, const int* atom_molptr, const float* energy, const float* energy_ref,
                       const float* is_dummy, const float* grad, const float* grad_ref, float* out);

/* ------------------------------------------------------------------------------------------------
 * Optimiser (training/lightning_model.py:297-299 Adam; lightning_trainer.py:92 gradient_clip_val=10):
 * sumsq: out[0] (+)= sum x^2.  adam: p -= lr * mhat/(sqrt(vhat)+eps) with g scaled by
 * clip = min(1, max_norm/(sqrt(*sumsq)+1e-6)) when sumsq != NULL; grad_scale multiplies g first. */
size_t grappa_sumsq_workspace_bytes(size_t n);
int grappa_sumsq_f32(void* stream, size_t n, const float* x, float* out, int accumulate, void* ws, size_t ws_bytes);
int grappa_adam_step_f32(void* stream, size_t n, float* p, const float* g, float* m, float* v,
                         float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                         float grad_scale, const float* sumsq, float max_norm);
/* ABI 8, for a train step captured in a hipGraph: the same update with the learning rate (lr_dev[0]) and the step count (step_dev[0] >= 1,
 * for the bias corrections) read from DEVICE memory at execution time. */
int grappa_adam_step_dyn_f32(void* stream, size_t n, float* p, const float* g, float* m, float* v, const float* lr_dev, float beta1, float beta2,
                             float eps, float weight_decay, const int* step_dev, float grad_scale, const float* sumsq, float max_norm);
/* ------------------------------------------------------------------------------------------------
 * bf16 storage configuration (BASELINE configs[2]: "bf16, MFMA dense heads"; the reference's Lightning `precision` / its
 * torch.set_float32_matmul_precision('medium'), training/trainrun.py:3): activations and activation gradients are kept as bf16
 * in HBM (uint16_t bit patterns, round to nearest even on every store), half the bytes of every HBM-bound kernel; statistics,
 * softmax, accumulation, parameters, parameter gradients, energies and forces stay fp32.  Same semantics and argument meaning
 * as the *_f32 entry points above; rows 8-byte aligned.  The dense products take precision GRAPPA_GEMM_BF16 with a_planes /
 * b_planes = 1 and *_nplanes = 1 (grappa_gemm_desc): one plane = the bf16 tensor itself. */
int grappa_layernorm_fwd_bf16(void* stream, int M, int W, const uint16_t* x, int ldx, const float* gamma, const float* beta,
                              uint16_t* y, int ldy, float* mean, float* rstd);
int grappa_layernorm_bwd_bf16(void* stream, int M, int W, const uint16_t* dy, int lddy, const uint16_t* x, int ldx,
                              const float* mean, const float* rstd, const float* gamma, uint16_t* dx, int lddx,
                              float* dgamma, float* dbeta, int accumulate, void* ws, size_t ws_bytes);
int grappa_act_dropout_bwd_bf16(void* stream, int M, int N, const uint16_t* dy, int lddy, const uint16_t* y, int ldy,
                                float drop_p, uint64_t drop_seed, uint16_t* dz, int lddz, const uint64_t* drop_salt);
int grappa_gat_fwd_bf16(void* stream, int N, int E, int H, int D, const int* indptr, const int* indices,
                        const uint16_t* ft, uint16_t* out, float* alpha);
int grappa_gat_bwd_bf16(void* stream, int N, int E, int H, int D, const int* indptr, const int* indices, const int* rev,
                        const uint16_t* ft, const uint16_t* out, const float* alpha, const uint16_t* dout,
                        uint16_t* dft, float* delta);
int grappa_tuple_gather_fwd_bf16(void* stream, int T, int s, int W, const uint16_t* a, int lda, const int* idx,
                                 const float* pe, uint16_t* x, int ldx);
int grappa_tuple_gather_bwd_bf16(void* stream, int N, int W, const int* inv_ptr, const int* inv_rows,
                                 const uint16_t* dx, int lddx, uint16_t* da, int ldda, int has_pe, int accumulate);
int grappa_seqattn_fwd_bf16(void* stream, int s, int T, int nheads, int dh, const uint16_t* qkv, uint16_t* out);
int grappa_seqattn_bwd_bf16(void* stream, int s, int T, int nheads, int dh, const uint16_t* qkv, const uint16_t* dout, uint16_t* dqkv);
int grappa_perm_concat_fwd_bf16(void* stream, int s, int T, int F, int P, const int* h_perm, const uint16_t* x, uint16_t* z);
int grappa_perm_concat_bwd_bf16(void* stream, int s, int T, int F, int P, const int* h_perm, const uint16_t* dz, uint16_t* dx);
/* element-type conversion of a [M,N] view (round to nearest even), for the few places where the two configurations meet */
int grappa_convert_f32_to_bf16(void* stream, int M, int N, const float* x, int ldx, uint16_t* y, int ldy);
int grappa_convert_bf16_to_f32(void* stream, int M, int N, const uint16_t* x, int ldx, float* y, int ldy);

/* ------------------------------------------------------------------------------------------------
 * ABI 11: the FUSED writer-head layer (SURVEY section 8(b) "grappa_writer_head_fwd/bwd"): one kernel per transformer layer of a writer head,
 * one workgroup per tile of 64 token rows (= every token of 32 / 21 / 16 tuples), the 512-wide activations resident in LDS:
 *     x1 = LN(x; n1);  qkv = x1 W_in^T + b_in;  a = multi-head attention over the s tokens of each tuple (nheads x 64, softmax over the s keys);
 *     x2 = drop(a W_o^T + b_o; seed1) + x1;  x3 = LN(x2; nf);  u = ELU(x3 W_1^T + b_1);  out = drop(u W_2^T + b_2; seed2) + x3
 * Replaces: models/network_utils.py:112-133 (DottedAttWithMLP.forward) with :44-54 (FeedForwardLayer.forward), one layer of the stack of
 * models/perm_equiv_transformer.py:121-151; nn.MultiheadAttention's in_proj / out_proj as in grappa_gemm_f32 above.
 * Token rows as everywhere: row = pos * T + t, tensors contiguous with F = 512 columns (qkv: 1536).  Dropout as grappa_gemm_f32 (element index
 * row * 512 + column).  dtype GRAPPA_WRITER_BF16: x, out and the saved activations are bf16 (the bf16 storage configuration), products on
 * v_mfma_f32_16x16x32_bf16 with fp32 accumulation, LayerNorm / softmax / ELU in fp32; every stored tensor is rounded exactly where the
 * unfused kernels round it.  Weights travel PACKED in MFMA fragment order (grappa_writer_pack_weight, once per optimiser step).
 * save_* (all or none; training): the tensors the backward pass reads -- x1, qkv, att, x2, x3, u and the statistics of both LayerNorms -- are
 * written as by-products.  Supported: F = 512, nheads = 8, s = 2, 3, 4; anything else GRAPPA_ERR_ARG (the caller runs the unfused sequence). */
#define GRAPPA_WRITER_BF16 1
typedef struct grappa_writer_layer_desc {
    int s, T, F, nheads, dtype;
    const void* x;                                        /* (s*T, F) */
    void* out;                                            /* (s*T, F) */
    const void *w_in_pk, *w_o_pk, *w1_pk, *w2_pk;         /* packed (3F x F), (F x F), (F x F), (F x F) */
    const float *b_in, *b_o, *b1, *b2;                    /* 3F, F, F, F */
    const float *n1_gamma, *n1_beta, *nf_gamma, *nf_beta; /* F each */
    float drop_p;
    uint64_t seed1, seed2;
    const uint64_t* drop_salt;                            /* device word mixed into both seeds, or NULL (grappa_gemm_desc.drop_salt) */
    float *save_mean1, *save_rstd1, *save_meanf, *save_rstdf;      /* (s*T) each, or NULL */
    void *save_x1, *save_qkv, *save_att, *save_x2, *save_x3, *save_u;      /* (s*T, F), qkv (s*T, 3F), or NULL */
    /* GATHER mode (gather_idx != NULL; the first layer of a head whose LayerNorm and q | k | v were computed once per (atom, position) TABLE row,
       reference models/interaction_parameters.py:155-180 + network_utils.py:112-133 on tokens that depend on their tuple only through (atom, position)):
       x1 and q | k | v are GATHERED -- token (pos, t) takes row gather_idx[t * s + pos] of x1_tab (rows x F, already normalised) and of qkv_tab
       (rows x 3F, bias included) -- instead of computed; x, n1_*, w_in_pk, save_x1 / save_qkv / save_mean1 / save_rstd1 are not used */
    const int* gather_idx;
    const void *x1_tab, *qkv_tab;
    int x2_tiled;      /* != 0: save_x2 holds grappa_writer_head_tiles(s, T) * 64 rows in the TILE order of grappa_writer_head_bwd (x2_tiled there too):
                          per tile and lane the 16 accumulator quads behind one another; 0: plain rows (what grappa_layernorm_bwd_bf16 reads) */
} grappa_writer_layer_desc;
int grappa_writer_head_fwd(void* stream, const grappa_writer_layer_desc* d);
/* The backward pass of the same layer: the whole input-gradient chain in one kernel per tile of 64 token rows
 *     dz2 = mask2(dout);  dz1 = (dz2 W_2) * ELU'(u);  dx3 = dz1 W_1 + dout;  dx2 = LN'(dx3; x2, nf);  dzo = mask1(dx2);  datt = dzo W_o;
 *     dqkv = attention'(qkv, datt);  dx1 = dqkv W_in + dx2;  dx = LN'(dx1; x, n1)
 * (the derivative of models/network_utils.py:112-133 with :44-54; mask1 / mask2: the forward's dropout masks, regenerated from seed1 / seed2).
 * The four weight (and bias) gradients are ordinary grouped weight-gradient products (grappa_gemm_f32_grouped) over the operands this kernel
 * writes as by-products -- dz2, dz1, dzo (s*T, F), dqkv (s*T, 3F) -- and the activations the forward saved (u, x3, att, x1).  LayerNorm parameter
 * gradients: per-tile partial sums, ln*_part[tile][0][F] = dgamma, [tile][1][F] = dbeta, grappa_writer_head_tiles(s, T) tiles (the layout
 * grappa_colsum_partials_batched reduces).  Weights packed TRANSPOSED (grappa_writer_pack_weight(F, 3F or F, W, ld, 1, ...)). */
typedef struct grappa_writer_layer_bwd_desc {
    int s, T, F, nheads, dtype;
    const void* dout;                                     /* (s*T, F): gradient of the layer's output */
    const void *x, *qkv, *x2, *u;                         /* the layer's input and what grappa_writer_head_fwd saved */
    const float *mean1, *rstd1, *meanf, *rstdf;
    const float *n1_gamma, *nf_gamma;
    const void *w_in_tpk, *w_o_tpk, *w1_tpk, *w2_tpk;     /* packed W_in^T (F x 3F), W_o^T, W_1^T, W_2^T (F x F) */
    float drop_p;
    uint64_t seed1, seed2;
    const uint64_t* drop_salt;
    void* dx;                                             /* (s*T, F): gradient of the layer's input */
    void *dz2, *dz1, *dzo, *dqkv;
    float *ln1_part, *lnf_part;                           /* (tiles, 2, F) each */
    int x2_tiled;                                         /* the layout of x2: see grappa_writer_layer_desc */
    const int* gather_idx;                                /* GATHER mode: `qkv` is the (rows x 3F) TABLE read through gather_idx; the chain ends behind the attention:
                                                             `dqkv` (token rows) and `dx` = the skip branch's gradient dx2 (token rows) are the outputs, x / mean1 / rstd1 /
                                                             n1_gamma / w_in_tpk / ln1_part are not used */
} grappa_writer_layer_bwd_desc;
int grappa_writer_head_bwd(void* stream, const grappa_writer_layer_bwd_desc* d);
int grappa_writer_head_tiles(int s, int T);
/* W (N x K fp32, rows ldw elements apart; transpose != 0: the operand is W^T, i.e. W is K x N) in the fragment order of the fused layer: block
 * (n / 16, k / 32) is one contiguous KB, 16 bytes per lane.  N % 16 == 0, K % 32 == 0; `out` holds grappa_writer_pack_bytes(N, K, dtype) bytes. */
size_t grappa_writer_pack_bytes(int N, int K, int dtype);
int grappa_writer_pack_weight(void* stream, int N, int K, const float* W, int ldw, int transpose, int dtype, void* out);

/* the dropout decision used by every kernel above, exposed for tests: 1 = keep */
int grappa_dropout_keep(uint64_t seed, uint64_t index, float p);

#ifdef __cplusplus
}
#endif
#endif /* GRAPPA_HIP_H */
