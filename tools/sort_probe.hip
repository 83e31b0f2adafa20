// sort_probe.hip -- how fast is a stable device radix sort of (tile index, crossing index) pairs on MI355X?
// hipcc --offload-arch=gfx950 -O3 tools/sort_probe.hip -o tools/sort_probe
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <stdio.h>
#include <vector>
int main() {
    const size_t n = 4617444;
    const unsigned bits = 21;
    std::vector<unsigned> hk(n), hv(n);
    unsigned long long s = 88172645463325252ull;
    for (size_t i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; hk[i] = (unsigned)(i / 4 + (s % 9) * 256) & ((1u << bits) - 1); hv[i] = (unsigned)i; }
    unsigned *k0, *k1, *v0, *v1;
    hipMalloc(&k0, n * 4); hipMalloc(&k1, n * 4); hipMalloc(&v0, n * 4); hipMalloc(&v1, n * 4);
    hipMemcpy(k0, hk.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(v0, hv.data(), n * 4, hipMemcpyHostToDevice);
    size_t tmp_bytes = 0;
    rocprim::radix_sort_pairs(nullptr, tmp_bytes, k0, k1, v0, v1, n, 0, bits, 0);
    void* tmp; hipMalloc(&tmp, tmp_bytes);
    printf("temp storage %zu bytes\n", tmp_bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        rocprim::radix_sort_pairs(tmp, tmp_bytes, k0, k1, v0, v1, n, 0, bits, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("radix_sort_pairs %zu pairs, %u bits: %.1f us\n", n, bits, ms * 1000);
    }
    std::vector<unsigned> ok(n), ov(n);
    hipMemcpy(ok.data(), k1, n * 4, hipMemcpyDeviceToHost); hipMemcpy(ov.data(), v1, n * 4, hipMemcpyDeviceToHost);
    bool good = true;
    for (size_t i = 1; i < n && good; i++) good = ok[i - 1] < ok[i] || (ok[i - 1] == ok[i] && ov[i - 1] < ov[i]);
    printf("sorted and stable: %s\n", good ? "yes" : "NO");
    return 0;
}
