// (included by gemm_bf16x_<mode>.hip: one translation unit per arithmetic, so that `make -j` compiles them side by side)
// fp32 GEMM emulated on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, 16x the rate of the native fp32 MFMA) by operand
// splitting: every fp32 operand is written as a sum of bf16 pieces, a = a0 + a1 + a2 with a0 = bf16(a), a1 = bf16(a - a0),
// a2 = bf16(a - a0 - a1) (each subtraction exact in fp32; three round-to-nearest pieces carry 24 significant bits), and the
// product is assembled from bf16 x bf16 partial products accumulated in fp32:
//   X9 : all 9 a_p*b_q -- every partial product is exact in fp32, so only the accumulation order differs from an fp32 FMA chain
//   X6 : drops a1*b2, a2*b1, a2*b2 (each <= 2^-24 |a||b|): ~2 ulp of fp32 per product
//   X3 : two pieces, a0*b0 + a0*b1 + a1*b0 (~2^-16 relative);  X1: plain bf16 operands (bf16 compute configs)
//   H3 : TWO fp16 pieces (11 + 11 bits and a sign carry 24), hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16: fp32-grade at half
//        the matrix instructions of X6.  fp16 has 5 exponent bits, so every row of A and of B is scaled by a power of two that
//        puts its largest magnitude (a_amax / b_amax, one pass over the operand ahead of the product) into [2^14, 2^15); the
//        accumulators are scaled back (v_ldexp_f32, exact) before the epilogue.  Small elements of a row become fp16 denormals
//        (kept by the gfx950 matrix cores): absolute error <= 2^-39 of the row's maximum.
//
// What bounds this kernel is the LDS, not the matrix pipe: three pieces triple every fragment read and every store, so the
// tile is shaped to minimise LDS bytes per MFMA.  256x128 macro tile, 8 wavefronts as 4 x 2, each a 64x64 block of four 32x32
// accumulators: 12 ds_read_b128 feed 24 MFMAs per 16-deep k-slab (a 64x32 wavefront tile needs 9 reads per 12 MFMAs and
// saturates the LDS array at two wavefronts per SIMD).  K advances in slabs of 16: global fp32 -> registers (four slabs ahead)
// -> split to bf16 pieces -> LDS [piece][row][16 bf16 + 16 B pad] in two stages of 54 KB; MFMA operand fragments are single
// ds_read_b128 (8 consecutive k of one row).  Row-contiguous ("k-major") sources are transposed on the way in: a thread owns
// one row and reads its 4 k values with 4 row-coalesced dword loads.  Per slab and wavefront: the first two thirds of the MFMAs run
// interleaved with the split + LDS stores of the next slab, then one barrier, then the next slab's fragment reads are issued
// between the remaining MFMAs.
// Epilogue, split-K slabs, XCD-aware order and the tail launch are shared with the native kernel (gemm_common.h).
#include "gemm_common.h"

using namespace grappa_gemm;

// timing experiments only (tools/gemm_f16x3_check.py --build-variants): GB_KNOCK = 1 no MFMAs, 2 no split arithmetic (raw halves
// stored), 3 no global loads in the steady state, 4 no LDS traffic in the steady state, 5 no barrier in the steady state, 6 no epilogue
#ifndef GB_KNOCK
#define GB_KNOCK 0
#endif
#ifndef GB_STAGGER
#define GB_STAGGER 0
#endif

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int SK = 16;                  // k-slab per pipeline step (one 32x32x16 MFMA deep)
constexpr int ROWB = 48;                // LDS row: 16 bf16 (32 B) + 16 B pad -> conflict-free ds_read_b128 per 16-lane group
#ifndef GB_AHEAD
#define GB_AHEAD 4
#endif
constexpr int AHEAD = GB_AHEAD;         // register sets (even): slab s+AHEAD is loaded while slab s is multiplied

enum Mode { X1 = 1, X3 = 3, X6 = 6, X9 = 9, H3 = 103 };
template <int MODE> struct Pieces {
    static constexpr bool HALF = MODE == H3;
    static constexpr int NP = MODE == X1 ? 1 : (MODE == X3 || MODE == H3 ? 2 : 3);
    static constexpr int NPROD = MODE == H3 ? 3 : MODE;
};

// power of two that moves a row's largest magnitude (fp32 bit pattern) into [2^14, 2^15): exponent field E -> 2^(141 - E)
// (zero / denormal rows: 2^141, still finite after scaling; Inf / NaN rows stay Inf / NaN)
__device__ inline int amax_shift(unsigned bits) { return 141 - (int)((bits >> 23) & 0xffu); }

// largest magnitude of row m of A: one array, or the maximum over the per-segment partials its producer left (grappa_gemm_desc.a_amax_nseg)
__device__ inline unsigned a_row_amax(const grappa_gemm_desc& d, int m) {
    unsigned v = d.a_amax[m];
    // (eight independent loads in flight per trip: one load per trip is one L2 round trip per segment, 16 in a row for a 512-wide producer,
    //  in the prologue of a kernel that has nothing else to do yet -- measured +1.1 ms per C2 step)
    for (int s0 = 1; s0 < d.a_amax_nseg; s0 += 8) {
        unsigned t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = d.a_amax[(size_t)min(s0 + u, d.a_amax_nseg - 1) * d.M + m];
#pragma unroll
        for (int u = 0; u < 8; ++u) v = max(v, t[u]);
    }
    return v;
}

struct Quad { float x[4]; unsigned rm; };      // rm: pair-format sources only (the token row's largest magnitude); dead otherwise

// slab element (row, kq..kq+3) owned by this thread for quad slot j of an operand with ROWS rows.
// K-contiguous source: 4 lanes cover the 64 bytes of one row; k-major source: 64 lanes cover 64 consecutive rows of one k.
// KMV (k-major source read with 16-byte loads ALONG ITS ROWS): the quad is (rows row..row+3, ONE k = kq); 64 lanes cover 1 KB of one k.
template <int NT, int ROWS, bool KCONT, bool KMV = false>
__device__ inline void quad_coords(int j, int& row, int& kq) {
    if (KCONT) {
        row = (threadIdx.x >> 2) + (NT / 4) * j;
        kq = (threadIdx.x & 3) << 2;
    } else if (KMV) {
        const int f = threadIdx.x + j * NT;
        row = (f % (ROWS / 4)) << 2;
        kq = f / (ROWS / 4);
    } else {
        const int f = threadIdx.x + j * NT;
        row = f & (ROWS - 1);
        kq = (f / ROWS) << 2;
    }
}

// bytes of one piece of an operand in an LDS stage: [row][16 k + pad] (ROWB per row), or -- KMV -- [k][row] (transposing reads)
template <int ROWS, bool KMV> struct PieceBytes { static constexpr int value = KMV ? SK * ROWS * 2 : ROWS * ROWB; };

// Loads never wait for their data: out-of-range rows and k are only CLAMPED here (the addresses stay inside the operand);
// the k tail is zeroed when the slab is split and stored (store_quads<MASK>), AHEAD - 1 steps later.
template <int NT, int ROWS, bool KCONT, bool VEC, bool KMV = false>
__device__ inline void load_quads(const float* __restrict__ src, int ld, int row0, int k0, int R, int Kend, Quad (&q)[ROWS * 4 / NT]) {
#pragma unroll
    for (int j = 0; j < ROWS * 4 / NT; ++j) {
        int row, kq;
        quad_coords<NT, ROWS, KCONT, KMV>(j, row, kq);
        const int gk = k0 + kq;
        if (!KCONT && KMV) {
            // four consecutive rows of one k: the (16-byte padded, host-checked) source row covers round_up(R, 4)
            const int gr = min(row0 + row, ((R + 3) & ~3) - 4);
            const float4 v = *reinterpret_cast<const float4*>(src + (size_t)(gk < Kend ? gk : 0) * ld + gr);
            q[j].x[0] = v.x; q[j].x[1] = v.y; q[j].x[2] = v.z; q[j].x[3] = v.w;
            continue;
        }
        const int gr = min(row0 + row, R - 1);          // out-of-range rows are clamped: their results are never stored
        if (KCONT && VEC) {
            const float4 v = *reinterpret_cast<const float4*>(src + (size_t)gr * ld + (gk < Kend ? gk : 0));   // K % 4 == 0 here
            q[j].x[0] = v.x; q[j].x[1] = v.y; q[j].x[2] = v.z; q[j].x[3] = v.w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ck = gk + e < Kend ? gk + e : 0;
                q[j].x[e] = KCONT ? src[(size_t)gr * ld + ck] : src[(size_t)ck * ld + gr];
            }
        }
    }
}

__device__ inline float u2f(unsigned x) { return __uint_as_float(x); }

// ---- ABI 8: a k-major operand of the weight-gradient product given in the PAIR format.  The operand is [K tokens][ROWS features]; token
// row k was split under ITS OWN scale 2^s_k (s_k = 141 - exponent of rowmax[k]), but the product reduces over the tokens, so the kernel
// needs every element under ONE scale, the tensor's (2^s_T, s_T <= s_k).  Multiplying the fp16 halves by 2^-(s_k - s_T) is exact while the
// result stays a normal fp16 number -- the very halves a fresh split under the tensor's scale would give -- and rounds to the fp16
// denormal grid below that, like the fresh split does.  Two factors (2^-14 at most the first) reach down to 2^-39 of the tensor's
// largest magnitude, the resolution of the denormal grid itself.
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
__device__ inline unsigned pow2_neg_h2(int d) {               // 2^-d as two fp16 (d >= 0): normal, denormal, or 0 below 2^-24
    const unsigned h = d <= 14 ? (unsigned)(15 - d) << 10 : (d <= 24 ? 1u << (24 - d) : 0u);
    return h | (h << 16);
}
__device__ inline unsigned rescale_h2(unsigned v, unsigned f1, unsigned f2) {
    const half2_t r = (__builtin_bit_cast(half2_t, v) * __builtin_bit_cast(half2_t, f1)) * __builtin_bit_cast(half2_t, f2);
    return __builtin_bit_cast(unsigned, r);
}
// "quad" j of a pair-format source = ONE 16-byte chunk of one token row k = kq (the KMV coordinates: chunk c = row / 4): a token's row
// segment is a sequence of 64-byte granules [8 HI | 8 HI | 8 LO | 8 LO] of 16 features, so chunk c holds ONE piece (c & 2: LO) of the 8
// features 16 (c / 4) + 8 (c & 1) .. + 7 -- the 64 lanes of a wavefront read 1 KB of one token row in one instruction (the fp32 path's
// access pattern; 8-byte HI / LO loads per 4 features measured 10 % slower than the fp32-operand kernel) -- plus the token's maximum
template <int NT, int ROWS>
__device__ inline void load_quads_pairs(const uint16_t* __restrict__ src, int ld, const unsigned* __restrict__ rowmax, int row0, int k0, int R, int Kend,
                                        Quad (&q)[ROWS * 4 / NT]) {
#pragma unroll
    for (int j = 0; j < ROWS * 4 / NT; ++j) {
        int row, kq;
        quad_coords<NT, ROWS, false, true>(j, row, kq);
        const int c = row >> 2;
        const int gk = k0 + kq < Kend ? k0 + kq : 0;
        const int gran = min((row0 >> 4) + (c >> 2), (R >> 4) - 1);               // granules beyond the operand's rows are clamped (never stored)
        const uint4 v = *reinterpret_cast<const uint4*>(src + (size_t)gk * ld + 32 * gran + 8 * (c & 3));
        q[j].x[0] = u2f(v.x); q[j].x[1] = u2f(v.y); q[j].x[2] = u2f(v.z); q[j].x[3] = u2f(v.w);
        q[j].rm = rowmax[gk];
    }
}
// do_cs (the A operand of the tile_n == 0 workgroups): the bias gradient rides on the rescaled halves -- cs[0 .. 7] += this thread's piece of
// its 8 features UNDER THE TENSOR'S SCALE, one v_dot2_f32_f16 per element (HI and LO sums and the scale 2^-s_T meet in the reduction behind
// the main loop).  (Converting every half back to fp32 under its row's scale -- 3 vector instructions per element on a quarter of the
// workgroups -- made the whole launch 11 % slower than the fp32-operand kernel: the grid waits for its slowest workgroups.)
template <int NT, int ROWS, bool MASK>
__device__ inline void store_quads_pairs(char* __restrict__ opbase, const Quad (&q)[ROWS * 4 / NT], int krem, int e_tensor, bool do_cs, float (&cs)[8]) {
    constexpr int PIECE = PieceBytes<ROWS, true>::value;
    half2_t one_zero, zero_one;
    one_zero[0] = (_Float16)1.0f; one_zero[1] = (_Float16)0.0f;
    zero_one[0] = (_Float16)0.0f; zero_one[1] = (_Float16)1.0f;
#pragma unroll
    for (int j = 0; j < ROWS * 4 / NT; ++j) {
        int row, kq;
        quad_coords<NT, ROWS, false, true>(j, row, kq);
        const int c = row >> 2, piece = (c >> 1) & 1, row8 = 16 * (c >> 2) + 8 * (c & 1);
        const int d = max(e_tensor - (int)((q[j].rm >> 23) & 0xffu), 0), d1 = min(d, 14);
        const unsigned f1 = pow2_neg_h2(d1), f2 = pow2_neg_h2(d - d1);
        unsigned w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = rescale_h2(__float_as_uint(q[j].x[e]), f1, f2);
        if (MASK && kq >= krem) w[0] = w[1] = w[2] = w[3] = 0u;
        if (do_cs) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const half2_t h = __builtin_bit_cast(half2_t, w[e]);
                cs[2 * e] = __builtin_amdgcn_fdot2(h, one_zero, cs[2 * e], false);
                cs[2 * e + 1] = __builtin_amdgcn_fdot2(h, zero_one, cs[2 * e + 1], false);
            }
        }
        // [k][row] with the 16-byte chunk (8 rows) swizzled by the k-row, as store_quads: one 16-byte write
        char* dst = opbase + piece * PIECE + kq * (ROWS * 2) + ((((row8 >> 3) ^ ((kq & 3) << 2))) << 4);
        *reinterpret_cast<uint4*>(dst) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}
// split 4 consecutive-k fp32 values into NP bf16 pieces (round to nearest even; the residual r - float(piece) is exact in
// fp32) and store each piece's 4 values as one 8-byte LDS write.  krem = valid k of this slab counted from its first column
// (MASK: a slab at the end of a K range, whose tail is zero-filled here).
// HALF: the row's power-of-two scale first (shift[j], exact), then fp16 pieces hi = f16(r), lo = f16(r - hi).
template <int NT, int NP, int ROWS, bool KCONT, bool MASK, bool HALF, bool KMV = false>
__device__ inline void store_quads(char* __restrict__ opbase, const Quad (&q)[ROWS * 4 / NT], int krem, const int (&shift)[ROWS * 4 / NT][KMV ? 4 : 1]) {
    constexpr int PIECE = PieceBytes<ROWS, KMV>::value;
#pragma unroll
    for (int j = 0; j < ROWS * 4 / NT; ++j) {
        int row, kq;
        quad_coords<NT, ROWS, KCONT, KMV>(j, row, kq);
        float r[4] = {q[j].x[0], q[j].x[1], q[j].x[2], q[j].x[3]};
        if (MASK) {
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = (KMV ? kq : kq + e) < krem ? r[e] : 0.f;
        }
        // KMV: [k][row] with the 16-byte chunk (8 rows) swizzled by the k-row, chunk ^ ((k & 3) << 2): the transposing fragment reads
        // (read_frags) then meet no bank conflicts; otherwise [row][k] with a padded row
        char* dst = KMV ? opbase + kq * (ROWS * 2) + ((((row >> 3) ^ ((kq & 3) << 2))) << 4) + ((row & 7) << 1) : opbase + row * ROWB + kq * 2;
        if (HALF) {
            // (packed multiplies and v_fma_mix_f32 residuals -- 40 instead of 67 vector instructions per slab -- measured no faster:
            // the vector unit is not what this kernel waits for; the plain form keeps the exact v_ldexp_f32 for any shift)
            if (GB_KNOCK == 2) {
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    *reinterpret_cast<uint2*>(dst + p * PIECE) = make_uint2(__float_as_uint(r[p]) & 0x3bff3bffu, __float_as_uint(r[p + 2]) & 0x3bff3bffu);
                continue;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = __builtin_ldexpf(r[e], shift[j][KMV ? e : 0]);
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                f16x2 h01, h23;
                h01[0] = (_Float16)r[0]; h01[1] = (_Float16)r[1];        // round to nearest even; |r| < 2^15 never overflows
                h23[0] = (_Float16)r[2]; h23[1] = (_Float16)r[3];
                *reinterpret_cast<uint2*>(dst + p * PIECE) = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
                if (p + 1 < NP) {
                    r[0] -= (float)h01[0]; r[1] -= (float)h01[1];
                    r[2] -= (float)h23[0]; r[3] -= (float)h23[1];
                }
            }
            continue;
        }
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            bf16x2 h01, h23;
            h01[0] = (__bf16)r[0]; h01[1] = (__bf16)r[1];
            h23[0] = (__bf16)r[2]; h23[1] = (__bf16)r[3];
            const unsigned u01 = __builtin_bit_cast(unsigned, h01), u23 = __builtin_bit_cast(unsigned, h23);
            *reinterpret_cast<uint2*>(dst + p * PIECE) = make_uint2(u01, u23);
            if (p + 1 < NP) {                       // float(bf16) is the 16 bits moved to the top half of the word
                r[0] -= u2f(u01 << 16); r[1] -= u2f(u01 & 0xffff0000u);
                r[2] -= u2f(u23 << 16); r[3] -= u2f(u23 & 0xffff0000u);
            }
        }
    }
}

// bias gradient: sum over k of A(row, k) for the rows a thread stages.  Row-contiguous A, dword loads: one row per thread (cs[0]);
// KMV: four consecutive rows per thread (cs[0..3])
template <int NT, int ROWS, bool KMV, bool MASK>
__device__ inline void quad_rowsum(const Quad (&q)[ROWS * 4 / NT], int krem, float (&cs)[8]) {
#pragma unroll
    for (int j = 0; j < ROWS * 4 / NT; ++j) {
        int row, kq;
        quad_coords<NT, ROWS, false, KMV>(j, row, kq);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float v = (!MASK || (KMV ? kq : kq + e) < krem) ? q[j].x[e] : 0.f;
            cs[KMV ? e : 0] += v;
        }
    }
}

template <int NP, int TM, int TN> struct Frags { bf16x8 a[TM][NP], b[TN][NP]; };

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

// KMV operands lie [k][row] in the stage; ds_read_tr16_b64 transposes on the way out: a 16-lane group reads a block of 4 k-rows x 16
// rows -- lane 4q + p supplies &(k-row q, rows 4p .. 4p + 3) and receives row (lane & 15) of the four k-rows.  An MFMA operand lane
// (lr, lh) wants row lr, k = 8 lh + 0..7: two reads (k-rows 8 lh + 4 rd + q, rd = 0, 1)
template <int ROWS>
__device__ inline unsigned tr_offset(int row0, int lane, int rd) {
    const int g16 = lane >> 4, blk = (g16 & 1) * 16, lh = g16 >> 1, q = (lane & 15) >> 2, pp = lane & 3;
    const int k = 8 * lh + 4 * rd + q, m = row0 + blk + 4 * pp;
    return (unsigned)(k * (ROWS * 2) + ((((m >> 3) ^ (q << 2))) << 4) + ((m & 7) << 1));
}
__device__ inline bf16x8 tr_pair(const char* p0, const char* p1) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p1);
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

// MFMA operand fragments of one staged slab ((TM + TN) * NP ds_read_b128, or two transposing 8-byte reads each for KMV operands)
template <int NP, int BM, int BN, int TM, int TN, bool AKMV, bool BKMV>
__device__ inline void read_frags(const char* __restrict__ stage, int wm0, int wn0, int lane, Frags<NP, TM, TN>& f) {
    const int lr = lane & 31, lh = lane >> 5;
    constexpr int PA = PieceBytes<BM, AKMV>::value, PB = PieceBytes<BN, BKMV>::value;
    const char* a_s = stage;
    const char* b_s = stage + NP * PA;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if (AKMV) f.a[i][p] = tr_pair(a_s + p * PA + tr_offset<BM>(wm0 + i * 32, lane, 0), a_s + p * PA + tr_offset<BM>(wm0 + i * 32, lane, 1));
            else f.a[i][p] = *reinterpret_cast<const bf16x8*>(a_s + p * PA + (wm0 + i * 32 + lr) * ROWB + lh * 16);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if (BKMV) f.b[j][p] = tr_pair(b_s + p * PB + tr_offset<BN>(wn0 + j * 32, lane, 0), b_s + p * PB + tr_offset<BN>(wn0 + j * 32, lane, 1));
            else f.b[j][p] = *reinterpret_cast<const bf16x8*>(b_s + p * PB + (wn0 + j * 32 + lr) * ROWB + lh * 16);
        }
    }
}

// MFMAs [LO, HI) of one slab: products smallest first (pa + pb descending), the TM x TN accumulators innermost so that
// consecutive MFMAs never depend on each other
template <int MODE, int TM, int TN, int LO, int HI>
__device__ inline void mfma_range(const Frags<Pieces<MODE>::NP, TM, TN>& f, f32x16 (&acc)[TM][TN]) {
    constexpr int NP = Pieces<MODE>::NP;
    int idx = 0;
#pragma unroll
    for (int s = 2 * (NP - 1); s >= 0; --s) {
#pragma unroll
        for (int pa = 0; pa < NP; ++pa) {
            const int pb = s - pa;
            if (pb < 0 || pb >= NP) continue;
            if (MODE == X6 && s > 2) continue;                  // X6 keeps pa + pb <= 2
            if ((MODE == X3 || MODE == H3) && s > 1) continue;  // X3 / H3 keep pa + pb <= 1
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    // B fragment first: the accumulator holds the transposed tile (4 consecutive n per lane, tile_epilogue_rows)
                    if (GB_KNOCK != 1 && idx >= LO && idx < HI) {
                        if (MODE == H3)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, f.b[j][pb]), __builtin_bit_cast(f16x8, f.a[i][pa]),
                                                                               acc[i][j], 0, 0, 0);
                        else
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.b[j][pb], f.a[i][pa], acc[i][j], 0, 0, 0);
                    }
                    ++idx;
                }
        }
    }
}

// sched_group_barrier sequence for NMFMA x { 1 MFMA, a share of NVALU vector ops, a share of NWRITE LDS stores }
template <int G, int NMFMA, int NWRITE, int NVALU>
struct PhaseOrder {
    static __device__ inline void emit() {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, (NVALU + NMFMA - 1) / NMFMA, 0);
        constexpr int W = (G + 1) * NWRITE / NMFMA - G * NWRITE / NMFMA;
        if (W > 0) __builtin_amdgcn_sched_group_barrier(0x200, W > 0 ? W : 1, 0);
        PhaseOrder<G + 1, NMFMA, NWRITE, NVALU>::emit();
    }
};
template <int NMFMA, int NWRITE, int NVALU>
struct PhaseOrder<NMFMA, NMFMA, NWRITE, NVALU> {
    static __device__ inline void emit() {}
};

// sched_group_barrier sequence for NMFMA x { 1 MFMA, a share of NREAD LDS reads }
template <int G, int NMFMA, int NREAD>
struct ReadOrder {
    static __device__ inline void emit() {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        constexpr int R = (G + 1) * NREAD / NMFMA - G * NREAD / NMFMA;
        if (R > 0) __builtin_amdgcn_sched_group_barrier(0x100, R > 0 ? R : 1, 0);
        ReadOrder<G + 1, NMFMA, NREAD>::emit();
    }
};
template <int NMFMA, int NREAD>
struct ReadOrder<NMFMA, NMFMA, NREAD> {
    static __device__ inline void emit() {}
};

struct KRange {
    int kbeg, kend, nsteps;
    __device__ inline int k_of(int s) const { return kbeg + s * SK; }
};

// One slab.  On entry fc holds the fragments of slab s (LDS stage s & 1); on exit fn holds those of slab s + 1.
//   global loads of slab s+AHEAD -> L | first two thirds of the MFMAs on fc, interleaved with the bf16 split + LDS store of slab s+1
//   (registers S) into the other stage | barrier | fragment reads of slab s+1 -> fn, one per MFMA of the last third
// TAIL = one of the last steps of the K range: loads / stores happen only while slabs remain and the stored slab is masked to
// the valid k; main-loop steps do both unconditionally.
template <int NT, int MODE, int BM, int BN, int WN, bool AK, bool BKC, bool VEC, bool KMV, bool TAIL, int PSRC>
__device__ inline void pipeline_step(const grappa_gemm_desc& d, char* __restrict__ smem, f32x16 (&acc)[2][BN / WN / 32], int m0, int n0,
                                     const KRange& kr, int s, int wm0, int wn0, int lane, Quad (&la)[BM * 4 / NT], Quad (&lb)[BN * 4 / NT],
                                     const Quad (&sa)[BM * 4 / NT], const Quad (&sb)[BN * 4 / NT],
                                     const Frags<Pieces<MODE>::NP, 2, BN / WN / 32>& fc, Frags<Pieces<MODE>::NP, 2, BN / WN / 32>& fn,
                                     float (&cs)[8], bool do_cs, const int (&sha)[BM * 4 / NT][(!AK && KMV) ? 4 : 1],
                                     const int (&shb)[BN * 4 / NT][(!BKC && KMV) ? 4 : 1]) {
    constexpr int NP = Pieces<MODE>::NP, TM = 2, TN = BN / WN / 32;
    constexpr bool AV = !AK && KMV, BV = !BKC && KMV;          // row-contiguous operands read by 16-byte loads along their rows
    constexpr int PA = PieceBytes<BM, AV>::value, PB = PieceBytes<BN, BV>::value;
    constexpr int STAGE = NP * (PA + PB);
    constexpr int NM = Pieces<MODE>::NPROD * TM * TN;
    // MFMAs issued before the barrier (they carry the split + LDS stores of the next slab); the rest cover its fragment reads.
    // Two thirds / one third measured 1 % faster than halves (the vector work is spread thinner), three quarters no better
    constexpr int NFIRST = NM >= 12 ? NM * 2 / 3 : NM / 2;
    const bool do_store = !TAIL || s + 1 < kr.nsteps;
    const bool do_load = !TAIL || s + AHEAD < kr.nsteps;
    // the operand is in the pair format (wgrad layout, KMV, H3 only)
    constexpr bool pa_ = (PSRC & 1) != 0, pb_ = (PSRC & 2) != 0;
    if (do_load && !(GB_KNOCK == 3 && !TAIL)) {
        if constexpr (pa_) load_quads_pairs<NT, BM>(reinterpret_cast<const uint16_t*>(d.A), d.lda, d.a_rowmax, m0, kr.k_of(s + AHEAD), d.M, kr.kend, la);
        else load_quads<NT, BM, AK, VEC, AV>(d.A, d.lda, m0, kr.k_of(s + AHEAD), d.M, kr.kend, la);
        if constexpr (pb_) load_quads_pairs<NT, BN>(reinterpret_cast<const uint16_t*>(d.B), d.ldb, d.b_rowmax, n0, kr.k_of(s + AHEAD), d.N, kr.kend, lb);
        else load_quads<NT, BN, BKC, VEC, BV>(d.B, d.ldb, n0, kr.k_of(s + AHEAD), d.N, kr.kend, lb);
    }
    __builtin_amdgcn_sched_barrier(0);
    char* nxt = smem + ((s + 1) & 1) * STAGE;
    mfma_range<MODE, TM, TN, 0, NFIRST>(fc, acc);
    if (do_store && !(GB_KNOCK == 4 && !TAIL)) {
        const int krem = TAIL ? kr.kend - kr.k_of(s + 1) : SK;
        if constexpr (pa_) store_quads_pairs<NT, BM, TAIL>(nxt, sa, krem, 141 - sha[0][0], !AK && do_cs, cs);
        else store_quads<NT, NP, BM, AK, TAIL, Pieces<MODE>::HALF, AV>(nxt, sa, krem, sha);
        if constexpr (pb_) store_quads_pairs<NT, BN, TAIL>(nxt + NP * PA, sb, krem, 141 - shb[0][0], false, cs);
        else store_quads<NT, NP, BN, BKC, TAIL, Pieces<MODE>::HALF, BV>(nxt + NP * PA, sb, krem, shb);
        if (!AK && do_cs) {
            if constexpr (!pa_) quad_rowsum<NT, BM, AV, TAIL>(sa, krem, cs);
        }
    }
    if (!TAIL) {
        // issue order of this phase: one MFMA, then a slice of the split arithmetic and of the LDS stores, so that the matrix
        // pipe runs under the vector work instead of before it (left alone the compiler bunches all MFMAs after the barrier)
        constexpr int NQ = (BM + BN) * 4 / NT;
        PhaseOrder<0, NFIRST, NQ * NP, NQ * (NP == 3 ? 26 : NP == 2 ? (Pieces<MODE>::HALF ? 20 : 16) : 6)>::emit();      // vector ops per quad as counted in the ISA (an over-estimate leaves the last MFMAs bare)
    }
    __builtin_amdgcn_sched_barrier(0);
    if (!(GB_KNOCK == 5 && !TAIL)) __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    if (do_store && !(GB_KNOCK == 4 && !TAIL)) read_frags<NP, BM, BN, TM, TN, AV, BV>(nxt, wm0, wn0, lane, fn);
    if (GB_KNOCK == 4 && !TAIL) fn = fc;
    mfma_range<MODE, TM, TN, NFIRST, NM>(fc, acc);
    if (!TAIL) {
        // the next slab's fragment reads ride between these MFMAs instead of all eight wavefronts bursting them at the LDS
        // right after the barrier (an MFMA issues only once its wavefront's reads are queued): +1.7 % on the workload's shapes
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        ReadOrder<0, NM - NFIRST, (TM + TN) * NP + (AV ? TM * NP : 0) + (BV ? TN * NP : 0)>::emit();
    }
    __builtin_amdgcn_sched_barrier(0);
}

// one workgroup's tile; (nwg, wgid) = size of the problem's workgroup grid and this workgroup's place in it (a launch of its own:
// gridDim / blockIdx; a grouped launch: the problem's share of the grid)
template <int NT, int MODE, int BM, int BN, int WN, bool AK, bool BKC, bool VEC, bool KMV, int PSRC = 0>
__device__ __forceinline__ void gemm_bf16x_body(const GemmParams& p, int nwg, int wgid) {
    static_assert(PSRC == 0 || (MODE == H3 && KMV && !AK && !BKC), "pair-format sources: the wgrad layout of the fp16-split arithmetic only");
    static_assert(PSRC >= 0 && PSRC <= 3, "PSRC 4 (formats per product) is resolved by the grouped kernel");
    constexpr bool pa_ = (PSRC & 1) != 0, pb_ = (PSRC & 2) != 0;
    constexpr int NP = Pieces<MODE>::NP, TM = 2, TN = BN / WN / 32;
    constexpr int NQA = BM * 4 / NT, NQB = BN * 4 / NT;
    constexpr bool AV = !AK && KMV, BV = !BKC && KMV;
    constexpr int PA = PieceBytes<BM, AV>::value;
    static_assert((NT / 64 / WN) * 64 == BM, "wavefront grid must cover the tile");
    extern __shared__ char smem[];
    const grappa_gemm_desc& d = p.d;
    if (GB_STAGGER > 0 && wgid < 256) {                      // experiment: the first workgroup of a CU starts late by 0 .. 3 units
        const int ph = (wgid >> 3) & 3;
        for (int i = 0; i < ph; ++i) __builtin_amdgcn_s_sleep(GB_STAGGER);
    }
    const TileCoord tc = map_logical(p, nwg, wgid);
    const int split = tc.split, tile_local = tc.tile_local, tile_n = tc.tile_n;
    const int m0 = tc.tile_m * BM, n0 = tile_n * BN;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm0 = (wave / WN) * 64, wn0 = (wave % WN) * (BN / WN);
    const int lr = lane & 31, lh = lane >> 5;
    KRange kr;
    kr.kbeg = split * p.k_per_split;
    kr.kend = min(d.K, kr.kbeg + p.k_per_split);
    kr.nsteps = (kr.kend - kr.kbeg + SK - 1) / SK;
    const int nsteps = kr.nsteps;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    int sha[NQA][AV ? 4 : 1], shb[NQB][BV ? 4 : 1];            // H3: power-of-two scale of the row(s) each staged quad belongs to
#pragma unroll
    for (int j = 0; j < NQA; ++j) {
        int row, kq;
        quad_coords<NT, BM, AK, AV>(j, row, kq);
#pragma unroll
        for (int e = 0; e < (AV ? 4 : 1); ++e)
            sha[j][e] = Pieces<MODE>::HALF ? amax_shift((d.amax_bcast & 1) ? d.a_amax[0] : a_row_amax(d, min(m0 + row + e, d.M - 1))) : 0;
    }
#pragma unroll
    for (int j = 0; j < NQB; ++j) {
        int row, kq;
        quad_coords<NT, BN, BKC, BV>(j, row, kq);
#pragma unroll
        for (int e = 0; e < (BV ? 4 : 1); ++e)
            shb[j][e] = Pieces<MODE>::HALF ? amax_shift(d.b_amax[(d.amax_bcast & 2) ? 0 : min(n0 + row + e, d.N - 1)]) : 0;
    }
    const bool do_cs = !AK && d.a_colsum != nullptr && tile_n == 0;
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // sums over k of A(row, k) for the row (KMV: the four rows; pair format: the eight rows of one piece) this thread stages -- the same for all its quads
    if (nsteps > 0) {
        static_assert(AHEAD % 2 == 0, "the fragment sets alternate with the slabs");
        Quad qa[AHEAD][NQA], qb[AHEAD][NQB];      // register sets: slab t lives in set t % AHEAD from its load to its split
        Frags<NP, TM, TN> fr[2];                  // fragments of even / odd slabs
#pragma unroll
        for (int u = 0; u < AHEAD; ++u) {
            if constexpr (pa_) load_quads_pairs<NT, BM>(reinterpret_cast<const uint16_t*>(d.A), d.lda, d.a_rowmax, m0, kr.k_of(u < nsteps ? u : 0), d.M, kr.kend, qa[u]);
            else load_quads<NT, BM, AK, VEC, AV>(d.A, d.lda, m0, kr.k_of(u < nsteps ? u : 0), d.M, kr.kend, qa[u]);
            if constexpr (pb_) load_quads_pairs<NT, BN>(reinterpret_cast<const uint16_t*>(d.B), d.ldb, d.b_rowmax, n0, kr.k_of(u < nsteps ? u : 0), d.N, kr.kend, qb[u]);
            else load_quads<NT, BN, BKC, VEC, BV>(d.B, d.ldb, n0, kr.k_of(u < nsteps ? u : 0), d.N, kr.kend, qb[u]);
        }
        if constexpr (pa_) store_quads_pairs<NT, BM, true>(smem, qa[0], kr.kend - kr.kbeg, 141 - sha[0][0], !AK && do_cs, cs);
        else store_quads<NT, NP, BM, AK, true, Pieces<MODE>::HALF, AV>(smem, qa[0], kr.kend - kr.kbeg, sha);
        if constexpr (pb_) store_quads_pairs<NT, BN, true>(smem + NP * PA, qb[0], kr.kend - kr.kbeg, 141 - shb[0][0], false, cs);
        else store_quads<NT, NP, BN, BKC, true, Pieces<MODE>::HALF, BV>(smem + NP * PA, qb[0], kr.kend - kr.kbeg, shb);
        if (!AK && do_cs) {
            if constexpr (!pa_) quad_rowsum<NT, BM, AV, true>(qa[0], kr.kend - kr.kbeg, cs);
        }
        __syncthreads();
        read_frags<NP, BM, BN, TM, TN, AV, BV>(smem, wm0, wn0, lane, fr[0]);
        int s = 0;
        // step s stores slab s+1 (register set (s+1) % AHEAD) and loads slab s+AHEAD into the set slab s occupied
#define GRAPPA_STEP(TAIL, U) \
    pipeline_step<NT, MODE, BM, BN, WN, AK, BKC, VEC, KMV, TAIL, PSRC>(d, smem, acc, m0, n0, kr, s + (U), wm0, wn0, lane, qa[U], qb[U], qa[((U) + 1) % AHEAD], \
                                                                 qb[((U) + 1) % AHEAD], fr[(U) & 1], fr[((U) + 1) & 1], cs, do_cs, sha, shb)
        // main loop: AHEAD steps per trip (the register sets rotate); never stores the last slab of the range and every load
        // it issues is for an existing slab
        for (; s + 2 * AHEAD - 1 < nsteps; s += AHEAD) {
#pragma unroll
            for (int u = 0; u < AHEAD; ++u) { GRAPPA_STEP(false, u); }
        }
        for (; s < nsteps; s += AHEAD) {              // the last (up to 2 AHEAD - 1) steps
#pragma unroll
            for (int u = 0; u < AHEAD; ++u)
                if (s + u < nsteps) { GRAPPA_STEP(true, u); }
        }
#undef GRAPPA_STEP
    }

    if (!AK && do_cs) {
        __syncthreads();                                     // every wavefront is past its last fragment read
        float* red = reinterpret_cast<float*>(smem);
        if constexpr (pa_) {
            // thread t staged ONE piece of rows 16 (c / 4) + 8 (c & 1) .. + 7, c = t % (BM / 4), for the tokens of its k group t / (BM / 4):
            // red[2 * group + piece][row]
            const int c = threadIdx.x % (BM / 4), grp = threadIdx.x / (BM / 4);
#pragma unroll
            for (int e = 0; e < 8; ++e) red[(grp * 2 + ((c >> 1) & 1)) * BM + 16 * (c >> 2) + 8 * (c & 1) + e] = cs[e];
        } else if (AV) {
            // thread t staged rows 4 (t % 64) .. + 3 (k = t / 64 and t / 64 + 8): red[t / 64][row]
#pragma unroll
            for (int e = 0; e < 4; ++e) red[(threadIdx.x / (BM / 4)) * BM + ((threadIdx.x % (BM / 4)) << 2) + e] = cs[e];
        } else {
            red[threadIdx.x] = cs[0];                        // threads t, t + BM, ... staged row t
        }
        __syncthreads();
        if (threadIdx.x < BM) {
            float sum = 0.f;
            constexpr int NG = pa_ ? 2 * (NT / (BM / 4)) : (AV ? NT / (BM / 4) : NT / BM);
#pragma unroll
            for (int t = 0; t < NG; ++t) sum += red[threadIdx.x + t * BM];
            if constexpr (pa_) sum = __builtin_ldexpf(sum, -sha[0][0]);          // the halves were summed under the tensor's scale 2^s_T
            const int m = m0 + threadIdx.x;
            if (m < d.M) {
                if (p.nsplit > 1) p.cs_slab[(size_t)split * d.M + m] = sum;
                else d.a_colsum[m] += sum;
            }
        }
    }
    if (Pieces<MODE>::HALF && GB_KNOCK != 8) {
        // undo the row scales: accumulator element e of block (i, j) is (m, n) = (wm0 + 32 i + lr, wn0 + 32 j + 8 (e / 4) + 4 lh + e % 4)
        int ea[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) ea[i] = amax_shift((d.amax_bcast & 1) ? d.a_amax[0] : a_row_amax(d, min(m0 + wm0 + i * 32 + lr, d.M - 1)));
        const bool b_rows = (d.amax_bcast & 2) == 0 && (reinterpret_cast<uintptr_t>(d.b_amax) & 15) == 0;
        const int eb_all = (d.amax_bcast & 2) ? amax_shift(d.b_amax[0]) : 0;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn0 + j * 32 + g * 8 + lh * 4;                 // four consecutive columns, n % 4 == 0
                int eb[4];
                if (b_rows && n + 3 < d.N) {
                    const uint4 u = *reinterpret_cast<const uint4*>(d.b_amax + n);
                    eb[0] = amax_shift(u.x); eb[1] = amax_shift(u.y); eb[2] = amax_shift(u.z); eb[3] = amax_shift(u.w);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) eb[q] = (d.amax_bcast & 2) ? eb_all : amax_shift(d.b_amax[min(n + q, d.N - 1)]);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int i = 0; i < TM; ++i) acc[i][j][4 * g + q] = __builtin_ldexpf(acc[i][j][4 * g + q], -(ea[i] + eb[q]));
            }
    }
    __syncthreads();                                         // the stages (and the column-sum scratch) are dead: reuse as staging
    if (GB_KNOCK == 6) {
        float t = 0.f;                                       // keep every accumulator alive
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) t += acc[i][j][e];
        if (t == 123.456f) p.d.C[0] = t;
        return;
    }
    tile_epilogue_rows<BM, BN, TM, TN>(p, acc, reinterpret_cast<float*>(smem + wave * EPI_WAVE_BYTES), m0, n0, wm0, wn0, lane, split, tile_local,
                                       p.vec_io != 0);
    if (p.nsplit > 1 && p.tickets) splitk_finish_tile<NT>(p, tile_local, reinterpret_cast<volatile int*>(smem));
}

template <int NT, int MODE, int BM, int BN, int WN, bool AK, bool BKC, bool VEC, bool KMV>
__global__ __launch_bounds__(NT, 2) void gemm_bf16x_kernel(GemmParams p) {
    gemm_bf16x_body<NT, MODE, BM, BN, WN, AK, BKC, VEC, KMV>(p, gridDim.x, blockIdx.x);
}

// Grouped launch: ONE grid over several independent products (the weight gradients of a backward pass, deferred and launched
// together).  A wgrad alone has 8 .. 24 tiles and must cut its K (= tokens) 10 .. 32 ways to fill 256 CUs, i.e. write and re-read
// 10 .. 32 partial tiles per output tile; sixteen of them together fill the chip with 3 .. 11 cuts each.  Problem descriptors and
// the prefix of workgroups per problem live in device memory (copied ahead of the launch on the same stream).
template <int NT, int MODE, int BM, int BN, int WN, bool AK, bool BKC, bool VEC, bool KMV, int PSRC = 0>
__global__ __launch_bounds__(NT, 2) void gemm_bf16x_grouped_kernel(const GemmParams* __restrict__ ps, const int* __restrict__ wg_begin, int nprob) {
    const int wg = blockIdx.x;
    int g = 0;
    while (g + 1 < nprob && wg >= wg_begin[g + 1]) ++g;
    if constexpr (PSRC == 4) {
        // products of mixed operand formats in one launch: every workgroup runs the body SPECIALISED for its product's formats (one uniform
        // branch per workgroup; a body that decides per load measured 13 % slower: the pinned MFMA / VALU interleave assumes one instruction mix)
        const int f = (ps[g].d.a_planes ? 1 : 0) | (ps[g].d.b_planes ? 2 : 0);
        const int nwg = wg_begin[g + 1] - wg_begin[g], id = wg - wg_begin[g];
        if (f == 0) gemm_bf16x_body<NT, MODE, BM, BN, WN, AK, BKC, VEC, KMV, 0>(ps[g], nwg, id);
        else if (f == 1) gemm_bf16x_body<NT, MODE, BM, BN, WN, AK, BKC, VEC, KMV, 1>(ps[g], nwg, id);
        else if (f == 2) gemm_bf16x_body<NT, MODE, BM, BN, WN, AK, BKC, VEC, KMV, 2>(ps[g], nwg, id);
        else gemm_bf16x_body<NT, MODE, BM, BN, WN, AK, BKC, VEC, KMV, 3>(ps[g], nwg, id);
    } else {
        gemm_bf16x_body<NT, MODE, BM, BN, WN, AK, BKC, VEC, KMV, PSRC>(ps[g], wg_begin[g + 1] - wg_begin[g], wg - wg_begin[g]);
    }
}

// the same product of up to four independent problems (the four writer heads) in one grid: at small batches one head's product leaves
// most of the chip idle for a whole tile time (batch 32: 44 .. 176 workgroups on 256 CUs)
template <int NT, int MODE, int BM, int BN, int WN, bool AK, bool BKC, bool VEC, bool KMV>
__global__ __launch_bounds__(NT, 2) void gemm_bf16x_group4_kernel(GemmGroup4 g) {
    const int i = group4_find(g, blockIdx.x);
    gemm_bf16x_body<NT, MODE, BM, BN, WN, AK, BKC, VEC, KMV>(g.p[i], g.wg_begin[i + 1] - g.wg_begin[i], blockIdx.x - g.wg_begin[i]);
}

template <int NT, int MODE, int BM, int BN, int WN, bool AK, bool BKC, bool VEC, bool KMV>
int launch_group4_mode(hipStream_t st, const GemmGroup4& g) {
    constexpr size_t stages = 2 * (size_t)Pieces<MODE>::NP * (PieceBytes<BM, !AK && KMV>::value + PieceBytes<BN, !BKC && KMV>::value);
    constexpr size_t staging = (NT / 64) * (size_t)EPI_WAVE_BYTES;
    constexpr size_t smem = stages > staging ? stages : staging;
    auto kern = gemm_bf16x_group4_kernel<NT, MODE, BM, BN, WN, AK, BKC, VEC, KMV>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return GRAPPA_ERR_LAUNCH;
        attr_set = true;
    }
    GRAPPA_LAUNCH(kern, dim3(g.wg_begin[g.count]), dim3(NT), smem, st, g);
    return grappa_launch_status();
}

template <int NT, int MODE, int BM, int BN, int WN, bool AK, bool BKC, bool VEC, bool KMV>
int launch_mode(hipStream_t st, GemmParams& p) {
    constexpr size_t stages = 2 * (size_t)Pieces<MODE>::NP * (PieceBytes<BM, !AK && KMV>::value + PieceBytes<BN, !BKC && KMV>::value);
    constexpr size_t staging = (NT / 64) * (size_t)EPI_WAVE_BYTES;
    constexpr size_t smem = stages > staging ? stages : staging;
    auto kern = gemm_bf16x_kernel<NT, MODE, BM, BN, WN, AK, BKC, VEC, KMV>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return GRAPPA_ERR_LAUNCH;
        attr_set = true;
    }
    GRAPPA_LAUNCH(kern, dim3(p.ntiles_launch * p.nsplit), dim3(NT), smem, st, p);
    return grappa_launch_status();
}

// vec: every operand can be read with 16-byte loads (K-contiguous: along k; row-contiguous: along its rows, which then lie [k][row] in
// the LDS and reach the MFMA through transposing reads)
template <int NT, int MODE, int BM, int BN, int WN>
int launch_layout(hipStream_t st, GemmParams& p, bool vec) {
    const grappa_gemm_desc& d = p.d;
    if (d.a_kcontig && d.b_kcontig) return vec ? launch_mode<NT, MODE, BM, BN, WN, true, true, true, false>(st, p) : launch_mode<NT, MODE, BM, BN, WN, true, true, false, false>(st, p);
    if (d.a_kcontig) return vec ? launch_mode<NT, MODE, BM, BN, WN, true, false, true, true>(st, p) : launch_mode<NT, MODE, BM, BN, WN, true, false, false, false>(st, p);
    return vec ? launch_mode<NT, MODE, BM, BN, WN, false, false, false, true>(st, p) : launch_mode<NT, MODE, BM, BN, WN, false, false, false, false>(st, p);
}

template <int MODE>
int launch_tile(hipStream_t st, GemmParams& p, bool vec) {
    if (p.bm == 256 && p.bn == 128) return launch_layout<512, MODE, 256, 128, 2>(st, p, vec);                         // 4 x 2 wavefronts of 64 x 64
    if (p.bm == 128 && p.bn == 128) return launch_layout<512, MODE, 128, 128, 4>(st, p, vec);                         // 2 x 4 wavefronts of 64 x 32
    return GRAPPA_ERR_ARG;
}

// PSRC: bit 0 / bit 1 = the A / B operand of every product of the group is in the pair format (ABI 8; H3 and KMV only)
template <int MODE, bool KMV, int PSRC = 0>
int launch_grouped_wgrad(hipStream_t st, const GemmParams* d_ps, const int* d_wg_begin, int nprob, int total_wgs) {
    constexpr int NT = 512, BM = 256, BN = 128;
    constexpr size_t stages = 2 * (size_t)Pieces<MODE>::NP * (PieceBytes<BM, KMV>::value + PieceBytes<BN, KMV>::value);
    constexpr size_t staging = (NT / 64) * (size_t)EPI_WAVE_BYTES;
    constexpr size_t smem = stages > staging ? stages : staging;
    auto kern = gemm_bf16x_grouped_kernel<NT, MODE, BM, BN, 2, false, false, false, KMV, PSRC>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return GRAPPA_ERR_LAUNCH;
        attr_set = true;
    }
    GRAPPA_LAUNCH(kern, dim3(total_wgs), dim3(NT), smem, st, d_ps, d_wg_begin, nprob);
    return grappa_launch_status();
}

}  // namespace

// one arithmetic per translation unit: GRAPPA_BF16X_MODE_FUNCS(X6, x6) defines grappa_bf16x_launch_x6 / grappa_bf16x_launch_grouped_x6
// forward (b_kcontig) or input-gradient layout, 16-byte loads, tile 256 x 128
template <int MODE>
int launch_group4(hipStream_t st, const GemmGroup4& g, bool b_kcontig) {
    if (b_kcontig) return launch_group4_mode<512, MODE, 256, 128, 2, true, true, true, false>(st, g);
    return launch_group4_mode<512, MODE, 256, 128, 2, true, false, true, true>(st, g);
}

#define GRAPPA_BF16X_MODE_FUNCS(MODE, NAME)                                                                                                  \
    int grappa_bf16x_launch_##NAME(hipStream_t st, GemmParams& p, bool vec_kcontig) { return launch_tile<MODE>(st, p, vec_kcontig); }       \
    int grappa_bf16x_launch_group4_##NAME(hipStream_t st, const GemmGroup4& g, bool b_kcontig) { return launch_group4<MODE>(st, g, b_kcontig); } \
    int grappa_bf16x_launch_grouped_##NAME(hipStream_t st, const GemmParams* d_ps, const int* d_wg_begin, int nprob, int total_wgs, bool vec, \
                                           int psrc) {                                                                                     \
        if constexpr (MODE == H3) {                                                                                                        \
            if (psrc != 0 && !vec) return GRAPPA_ERR_ARG;                                                                                  \
            if (psrc == 1) return launch_grouped_wgrad<MODE, true, 1>(st, d_ps, d_wg_begin, nprob, total_wgs);                             \
            if (psrc == 2) return launch_grouped_wgrad<MODE, true, 2>(st, d_ps, d_wg_begin, nprob, total_wgs);                             \
            if (psrc == 3) return launch_grouped_wgrad<MODE, true, 3>(st, d_ps, d_wg_begin, nprob, total_wgs);                             \
            if (psrc == 4) return launch_grouped_wgrad<MODE, true, 4>(st, d_ps, d_wg_begin, nprob, total_wgs);                             \
        } else if (psrc != 0) {                                                                                                            \
            return GRAPPA_ERR_ARG;                                                                                                         \
        }                                                                                                                                  \
        return vec ? launch_grouped_wgrad<MODE, true>(st, d_ps, d_wg_begin, nprob, total_wgs)                                              \
                   : launch_grouped_wgrad<MODE, false>(st, d_ps, d_wg_begin, nprob, total_wgs);                                            \
    }

