// Where do the waves of a workgroup land?  (run the binary on the GPU box)
// k_coarse and other kernels give one wave of a 256-thread workgroup the sequential part of the work.  If the dispatcher put wave 0
// of every workgroup on the same SIMD of its CU, those waves would share one issue port while three SIMDs idle.  This records
// HW_ID (SIMD, CU, SH, SE) and XCC_ID per wave of a grid shaped like k_coarse's (1024 workgroups x 256 threads, LDS_KB of LDS per
// workgroup, all resident together) and prints, per workgroup-local wave number, how the waves spread over the four SIMDs, and --
// second table -- how many wave-0s share a SIMD of one CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <map>
#include <vector>
#ifndef LDS_KB
#define LDS_KB 24
#endif
__global__ __launch_bounds__(256) void k(uint2* out, int spin) {
    __shared__ uint32_t lds[LDS_KB * 256];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    float a = (float)lds[(threadIdx.x * 7) & 255];
    for (int i = 0; i < spin; i++) a = a * 1.0001f + 0.5f;  // keeps the workgroups resident together
    if ((threadIdx.x & 63u) == 0u) out[blockIdx.x * 4u + (threadIdx.x >> 6)] = make_uint2(hw, xcc + (a == 12345.0f ? 1u : 0u));
}
int main() {
    const int G = 1024;
    uint2* d;
    hipMalloc(&d, G * 4 * sizeof(uint2));
    hipLaunchKernelGGL(k, dim3(G), dim3(256), 0, 0, d, 200000);
    hipDeviceSynchronize();
    std::vector<uint2> h(G * 4);
    hipMemcpy(h.data(), d, G * 4 * sizeof(uint2), hipMemcpyDeviceToHost);
    int per_wave_simd[4][4] = {};
    std::map<uint32_t, int> wave0_per_simd;  // key: (xcc, se, sh, cu, simd)
    std::map<uint32_t, int> wg_per_cu;
    std::map<uint32_t, std::vector<int>> wgs_of_cu;
    for (int b = 0; b < G; b++)
        for (int w = 0; w < 4; w++) {
            const uint32_t hw = h[b * 4 + w].x, xcc = h[b * 4 + w].y & 15u;
            const uint32_t simd = (hw >> 4) & 3u, cu = (hw >> 8) & 15u, sh = (hw >> 12) & 1u, se = (hw >> 13) & 7u;
            per_wave_simd[w][simd]++;
            const uint32_t cukey = (xcc << 12) | (se << 8) | (sh << 4) | cu;
            if (w == 0) { wave0_per_simd[(cukey << 2) | simd]++; wg_per_cu[cukey]++; wgs_of_cu[cukey].push_back(b * 4 + (int)simd); }
        }
    printf("LDS %d KB per workgroup, %d workgroups of 256 threads\nwave of the workgroup -> SIMD 0..3:\n", LDS_KB, G);
    for (int w = 0; w < 4; w++) printf("  wave %d: %5d %5d %5d %5d\n", w, per_wave_simd[w][0], per_wave_simd[w][1], per_wave_simd[w][2], per_wave_simd[w][3]);
    int hist[17] = {};
    for (auto& kv : wave0_per_simd) hist[kv.second > 16 ? 16 : kv.second]++;
    printf("CUs used: %zu; SIMDs holding n wave-0s: ", wg_per_cu.size());
    for (int n = 1; n <= 16; n++) if (hist[n]) printf(" n=%d: %d", n, hist[n]);
    printf("\n(first 16 workgroups: xcc se sh cu | simd of waves 0..3)\n");
    for (int b = 0; b < 16; b++) {
        const uint32_t hw = h[b * 4].x;
        printf("  wg %2d: %u %u %u %2u |", b, h[b * 4].y & 15u, (hw >> 13) & 7u, (hw >> 12) & 1u, (hw >> 8) & 15u);
        for (int w = 0; w < 4; w++) printf(" %u", (h[b * 4 + w].x >> 4) & 3u);
        printf("\n");
    }
    printf("(workgroups of the first 6 CUs: id/simd-of-wave-0)\n");
    int shown = 0;
    for (auto& kv : wgs_of_cu) {
        if (shown++ >= 6) break;
        printf("  cu %05x:", kv.first);
        for (int v : kv.second) printf(" %d/%d", v >> 2, v & 3);
        printf("\n");
    }
    return 0;
}
