"""Builds build/variants/libgrappa_hip_tworow.so: the shipped library with ONE kernel changed -- LayerNorm forward takes two rows per
wavefront and trip (rows <= 512 wide), the round-1 kernel that returned deviating rows when this library's GEMMs ran on other queues
("Multi-queue deviation").  Test material for tools/stream_order_probe.py only; nothing ships from here.  The
variant is compiled with the compiler's default feature set -- packed fp32 instructions ON, unlike csrc/Makefile: they are what deviates.

    python tools/ln_two_rows_variant.py [gemm knock-out name ...]     # also links one library per named GEMM knock-out (GB_KNOCK)
"""
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
CSRC = os.path.join(ROOT, "grappa_amd", "csrc")
OUT = os.path.join(ROOT, "build", "variants")
HIPCC = ["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "-std=c++17", "--offload-arch=gfx950", f"-I{os.path.join(ROOT, 'include')}", f"-I{CSRC}"]

MARK = "    // one row per wavefront and trip."
TWO_ROWS = r'''
    if constexpr (NCH <= 2) {
        // two rows per trip: the second row's loads and shuffle reductions fill the latency of the first's
        for (int row0 = wave_global; row0 < M; row0 += 2 * nwaves) {
            const int row1 = row0 + nwaves;
            const bool two = row1 < M;
            const T* xr0 = x + (size_t)row0 * ldx;
            const T* xr1 = x + (size_t)(two ? row1 : row0) * ldx;
            float4 v0[NCH], v1[NCH];
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int c = lane + 64 * i;
                if (c < nvec) {
                    v0[i] = ld4(xr0, c);
                    v1[i] = ld4(xr1, c);
                    s0 += (v0[i].x + v0[i].y) + (v0[i].z + v0[i].w);
                    s1 += (v1[i].x + v1[i].y) + (v1[i].z + v1[i].w);
                }
            }
            const float mean0 = wave_sum(s0) / (float)W, mean1 = wave_sum(s1) / (float)W;
            float q0 = 0.f, q1 = 0.f;
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int c = lane + 64 * i;
                if (c < nvec) {
                    const float a = v0[i].x - mean0, b = v0[i].y - mean0, cc = v0[i].z - mean0, d = v0[i].w - mean0;
                    q0 += (a * a + b * b) + (cc * cc + d * d);
                    const float e = v1[i].x - mean1, f = v1[i].y - mean1, g = v1[i].z - mean1, h = v1[i].w - mean1;
                    q1 += (e * e + f * f) + (g * g + h * h);
                }
            }
            const float rstd0 = 1.0f / sqrtf(wave_sum(q0) / (float)W + 1e-5f), rstd1 = 1.0f / sqrtf(wave_sum(q1) / (float)W + 1e-5f);
            T* yr0 = y + (size_t)row0 * ldy;
            T* yr1 = y + (size_t)(two ? row1 : row0) * ldy;
            unsigned am0 = 0u, am1 = 0u;
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int c = lane + 64 * i;
                if (c < nvec) {
                    const float4 g = reinterpret_cast<const float4*>(gamma)[c];
                    const float4 b = reinterpret_cast<const float4*>(beta)[c];
                    float4 o;
                    o.x = (v0[i].x - mean0) * rstd0 * g.x + b.x;
                    o.y = (v0[i].y - mean0) * rstd0 * g.y + b.y;
                    o.z = (v0[i].z - mean0) * rstd0 * g.z + b.z;
                    o.w = (v0[i].w - mean0) * rstd0 * g.w + b.w;
                    st4(yr0, c, o);
                    am0 = max(am0, mag4(o));
                    if (two) {
                        o.x = (v1[i].x - mean1) * rstd1 * g.x + b.x;
                        o.y = (v1[i].y - mean1) * rstd1 * g.y + b.y;
                        o.z = (v1[i].z - mean1) * rstd1 * g.z + b.z;
                        o.w = (v1[i].w - mean1) * rstd1 * g.w + b.w;
                        st4(yr1, c, o);
                        am1 = max(am1, mag4(o));
                    }
                }
            }
            if (STORE_STATS && lane == 0) {
                mean_out[row0] = mean0;
                rstd_out[row0] = rstd0;
                if (two) {
                    mean_out[row1] = mean1;
                    rstd_out[row1] = rstd1;
                }
            }
            if (y_amax) {
                const unsigned m0 = wave_umax(am0), m1 = wave_umax(am1);
                if (lane == 0) {
                    y_amax[row0] = m0;
                    if (two) y_amax[row1] = m1;
                }
            }
        }
        return;
    }
'''


DPP_SUM = r'''
__device__ inline float wave_sum_dpp(float v) {
#define GRAPPA_DPP_ADD(CTRL) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true))
    GRAPPA_DPP_ADD(0xB1);       // quad_perm [1, 0, 3, 2]
    GRAPPA_DPP_ADD(0x4E);       // quad_perm [2, 3, 0, 1]
    GRAPPA_DPP_ADD(0x141);      // row_half_mirror
    GRAPPA_DPP_ADD(0x140);      // row_mirror: every lane of a row of 16 holds the row's sum
#undef GRAPPA_DPP_ADD
    const int b = __float_as_int(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(b, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(b, 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(b, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(b, 48));
    return (r0 + r1) + (r2 + r3);
}

'''


def without_maxima(two):
    """the same kernel without the row-maxima code (round 1's form: 58 registers; with it 62)"""
    import re
    two = re.sub(r"\n\s*am[01] = max\(am[01], mag4\(o\)\);", "", two)
    two = two.replace("            unsigned am0 = 0u, am1 = 0u;\n", "")
    i, j = two.index("            if (y_amax) {"), two.index("        }\n        return;")
    return two[:i] + two[j:]


def main():
    os.makedirs(OUT, exist_ok=True)
    src = open(os.path.join(CSRC, "rowwise.hip")).read()
    assert src.count(MARK) == 1, "rowwise.hip: the one-row loop's comment moved"
    var = os.path.join(OUT, "rowwise_tworow.hip")
    two = TWO_ROWS
    if "--no-maxima" in sys.argv:
        sys.argv.remove("--no-maxima")
        two = without_maxima(two)
    for opt, ins in (("--vmcnt-after-store", 'asm volatile("s_waitcnt vmcnt(0)" ::: "memory");'),
                     ("--nops-after-store", 'asm volatile("s_nop 15\\n s_nop 15\\n s_nop 15\\n s_nop 15" ::: "memory");')):
        if opt in sys.argv:
            # between the first row's 16-byte store and the second row's arithmetic (which reuses the store's data registers)
            sys.argv.remove(opt)
            assert two.count("st4(yr0, c, o);") == 1
            two = two.replace("st4(yr0, c, o);", "st4(yr0, c, o);\n                    " + ins)
    if "--nops-after-load" in sys.argv:
        # gamma / beta have arrived (vmcnt 0), then 16 idle cycles before the first instruction that reads them
        sys.argv.remove("--nops-after-load")
        kg = "                    const float4 g = reinterpret_cast<const float4*>(gamma)[c];\n"
        kb = "                    const float4 b = reinterpret_cast<const float4*>(beta)[c];\n"
        assert two.count(kg) == 1 and two.count(kb) == 1
        two = two.replace(kg, kg.replace("const float4 g", "float4 g"))
        two = two.replace(kb, kb.replace("const float4 b", "float4 b") +
                          '                    asm volatile("s_waitcnt vmcnt(0)\\n s_nop 15" : "+v"(g.x), "+v"(g.y), "+v"(g.z), "+v"(g.w), '
                          '"+v"(b.x), "+v"(b.y), "+v"(b.z), "+v"(b.w) :: "memory");\n')
    if "--dpp-sums" in sys.argv:
        # the two-row kernel's wave sums on the DPP data path + v_readlane instead of __shfl_xor (= ds_bpermute, the LDS crossbar)
        sys.argv.remove("--dpp-sums")
        two = two.replace("wave_sum(", "wave_sum_dpp(")
        src = src.replace("template <int NCH, bool STORE_STATS, typename T>\n__global__ __launch_bounds__(256) void layernorm_fwd_kernel(", DPP_SUM +
                          "template <int NCH, bool STORE_STATS, typename T>\n__global__ __launch_bounds__(256) void layernorm_fwd_kernel(", 1)
        assert "wave_sum_dpp(float" in src
    open(var, "w").write(src.replace(MARK, two + MARK))
    obj = os.path.join(OUT, "rowwise_tworow.o")
    subprocess.run(HIPCC + ["-c", var, "-o", obj], check=True)
    base = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".o") and f != "rowwise.o"]
    so = os.path.join(OUT, "libgrappa_hip_tworow.so")
    subprocess.run(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-o", so, obj, *base], check=True)
    print(so)
    knock = {"nomfma": 1, "nosplit": 2, "noglobal": 3, "nolds": 4, "nobarrier": 5, "noepi": 6, "nostore": 7, "nodescale": 8}
    units = ("gemm_bf16x_h3", "gemm_bf16x_x6", "gemm_bf16x_x3")
    for name in sys.argv[1:]:
        objs = []
        for u in units:
            o = os.path.join(OUT, f"{u}_{name}.o")
            subprocess.run(HIPCC + ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops",      # the GEMM neighbours: compiled as csrc/Makefile does
                                    f"-DGB_KNOCK={knock[name]}", "-c", os.path.join(CSRC, u + ".hip"), "-o", o], check=True)
            objs.append(o)
        rest = [b for b in base if os.path.basename(b)[:-2] not in units]
        so = os.path.join(OUT, f"libgrappa_hip_tworow_{name}.so")
        subprocess.run(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-o", so, obj, *objs, *rest], check=True)
        print(so)


if __name__ == "__main__":
    main()
