// Write-pattern microbenchmark for gfx950 (run on the GPU box): 3.1 M records of 24 bytes (three 8-byte words) written once, one wave
// per 64 records, by the patterns k_flatten_lines could use.  Prints us per pass and GB/s of payload.
//   stride3   lane writes its record's three words with three stores (word 3 * rec + j): every store touches the whole 1.5 KB
//   dense     store j writes words 64 j ... 64 j + 63 of the wave's 1.5 KB (records in lane order = memory order)
//   runs R    as dense, but the records are scattered in runs of R records inside a window of 576 records (a batch of 64 jobs)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/wr.hip -o tools/ubench/wr
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define NREC (3u << 20)
#define WIN 576u
__device__ __forceinline__ uint32_t place(uint32_t rec, uint32_t R) {  // record -> position: runs of R records shuffled inside a window
    if (R == 0u) return rec;
    const uint32_t w = rec / WIN, o = rec % WIN, run = o / R, in = o % R, nruns = WIN / R;
    const uint32_t prun = (run * 37u + 11u) % nruns;  // 37 is coprime with every nruns used (576 / R for R = 1, 2, 3, 4, 9: 576, 288, 192, 144, 64)
    return w * WIN + prun * R + in;
}
template <int MODE>
__global__ __launch_bounds__(256) void k(uint2* __restrict__ out, uint32_t R) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
    const uint32_t rec0 = wave * 64u;
    if (rec0 >= NREC) return;
    if (MODE == 0) {
        const uint32_t pos = place(rec0 + lane, R);
#pragma unroll
        for (uint32_t j = 0; j < 3; j++) out[3u * pos + j] = make_uint2(pos, j);
    } else {
#pragma unroll
        for (uint32_t j = 0; j < 3; j++) {
            const uint32_t e = 64u * j + lane, rec = rec0 + e / 3u, part = e % 3u;
            const uint32_t pos = place(rec, R);
            out[3u * pos + part] = make_uint2(pos, part);
        }
    }
}
int main() {
    uint2* out; (void)hipMalloc(&out, (size_t)NREC * 24 + 4096);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const uint32_t blocks = NREC / 256u;
    struct { const char* name; int mode; uint32_t R; } cases[] = {{"stride3 (lane = record, 3 stores)", 0, 0}, {"dense (transposed)", 1, 0},
        {"stride3, runs of 2 records", 0, 2}, {"transposed, runs of 1", 1, 1}, {"transposed, runs of 2", 1, 2}, {"transposed, runs of 3", 1, 3},
        {"transposed, runs of 4", 1, 4}, {"transposed, runs of 9", 1, 9}};
    for (auto& c : cases) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; rep++) {
            (void)hipEventRecord(e0);
            if (c.mode == 0) k<0><<<blocks, 256>>>(out, c.R); else k<1><<<blocks, 256>>>(out, c.R);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < best) best = ms;
        }
        printf("%-36s %7.1f us  %7.1f GB/s\n", c.name, best * 1e3, (double)NREC * 24 / (best * 1e-3) / 1e9);
    }
    return 0;
}
