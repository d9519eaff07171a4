// Which CUs does bit i of a hipExtStreamCreateWithCUMask mask enable on an MI355X (8 XCDs x 32 CUs)?  Launches a grid on masked streams and
// histograms the XCD (HW_REG_XCC_ID) and the (SE, CU) of HW_REG_HW_ID that each workgroup ran on.
//     hipcc --offload-arch=gfx950 -O2 -o /tmp/cu_mask_probe tools/cu_mask_probe.hip && /tmp/cu_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <set>

__global__ void where_kernel(uint32_t* out, int spin) {
    uint32_t xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    float v = threadIdx.x;
    for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;          // long enough that every enabled CU takes workgroups
    if (threadIdx.x == 0) out[blockIdx.x] = ((xcc & 0xfu) << 16) | (hw & 0xffffu) | (v == 1.25f ? 1u << 31 : 0u);
}

static void probe(const char* name, const std::vector<uint32_t>& mask) {
    hipStream_t st;
    if (hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("%s: create failed\n", name); return; }
    const int n = 4096;
    uint32_t* d;
    hipMalloc(&d, n * 4);
    where_kernel<<<n, 256, 0, st>>>(d, 20000);
    hipStreamSynchronize(st);
    std::vector<uint32_t> h(n);
    hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
    int per_xcc[16] = {0};
    std::set<uint32_t> cus;
    for (uint32_t v : h) { per_xcc[(v >> 16) & 15]++; cus.insert((((v >> 16) & 15) << 16) | (v & 0xff00u)); }
    printf("%-28s distinct (xcd, se/sh/cu) = %3zu  workgroups per xcd:", name, cus.size());
    for (int x = 0; x < 8; ++x) printf(" %4d", per_xcc[x]);
    printf("\n");
    hipFree(d);
    hipStreamDestroy(st);
}

int main() {
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    printf("%s: %d CUs\n", pr.name, pr.multiProcessorCount);
    const int words = 8;                                   // 256 bits
    auto mk = [&](auto pred) { std::vector<uint32_t> m(words, 0u); for (int i = 0; i < 32 * words; ++i) if (pred(i)) m[i / 32] |= 1u << (i % 32); return m; };
    probe("all", mk([](int) { return true; }));
    probe("bits 0..31", mk([](int i) { return i < 32; }));
    probe("bits 0..63", mk([](int i) { return i < 64; }));
    probe("bits 128..255", mk([](int i) { return i >= 128; }));
    probe("i % 8 == 0", mk([](int i) { return i % 8 == 0; }));
    probe("i % 8 < 2", mk([](int i) { return i % 8 < 2; }));
    probe("i % 8 >= 4", mk([](int i) { return i % 8 >= 4; }));
    probe("i % 2 == 0", mk([](int i) { return i % 2 == 0; }));
    probe("(i / 8) % 2 == 0", mk([](int i) { return (i / 8) % 2 == 0; }));
    probe("i / 8 < 8 (CUs 0..7 of each)", mk([](int i) { return i / 8 < 8; }));
    probe("i % 8 == 0 && i / 8 < 8", mk([](int i) { return i % 8 == 0 && i / 8 < 8; }));
    return 0;
}
