// kernels_selftest.hip -- diagnostics: evaluates the dmath.h scalar routines on the device so that
// tests can compare them bit-for-bit with the CPU checker (tests/test_gpu_math.py).  Not part of
// the render path.
#include "kcommon.h"
#include <algorithm>
#include <utility>
#include <vector>

using namespace jd;

__global__ void k_selftest_math(int op, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = a[i], y = b ? b[i] : 0.0f, r = 0.0f;
    switch (op) {
        case 0: r = sin_(x); break;
        case 1: r = cos_(x); break;
        case 2: r = atan2_(x, y); break;
        case 3: r = acos_(x); break;
        case 4: r = asin_(x); break;
        case 5: r = pow23_abs_(x); break;
        case 6: r = x / y; break;
        case 7: r = sqrt_(x); break;
        case 8: r = round_(x); break;
        case 9: r = u2f(to_u32(x)); break;
        case 10: r = u2f((uint32_t)to_i32(x)); break;
        case 11: r = u2f((uint32_t)f32_to_f16(x)); break;
        case 12: r = x * y + x; break;      // must NOT be contracted to an FMA
        case 13: r = floor_(x * y + 0.5f); break;
        case 14: r = fmin_(x, y); break;
        case 15: r = fmax_(x, y); break;
        case 16: r = clamp_(x, 0.0f, 1.0f); break;
        case 17: r = clamp_(x * y, 0.0f, 1.0f); break;
        default: break;
    }
    out[i] = r;
}

extern "C" int jh_selftest_math_launch(hipStream_t stream, int op, const float* a, const float* b, float* out, uint32_t n) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_selftest_math, dim3((n + 255) / 256), dim3(256), 0, stream, op, a, b, out, n);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ------------------------------------------------------------------------------------------------
// Atomics self-test (round 6).  The kernels allocate with returning atomics whose VALUE differs from lane to lane on an address
// that is the same for the wave; LLVM's atomic optimizer rewrites such a call into one atomic per wave plus lane offsets, and in
// round 5 one of its strategies (DPP) returned overlapping ranges when the call sat in a loop that lanes leave at different trips.
// The product no longer asks for that strategy, and flatten's hot call sites aggregate by hand (wave_bump); this test holds
// BOTH forms -- whatever the compiler of the day makes of them -- to what a serial execution gives, so that a toolchain update
// that breaks either fails `pytest -m gpu` instead of a frame:
//   form 0  plain per-lane `atomicAdd(ctr, n)` in a loop with divergent skips and exits (the compiler's business)
//   form 1  the same through wave_bump (ours)
//   form 2  wave-private LDS: XOR of a per-lane value into ONE word (fine_msaa's even-odd row word) and 64-bit OR into a word chosen per
//           lane (fine_area's row masks), under the same divergent control flow
// Forms 0 / 1: every (p, n) handed out must tile [0, counter) exactly.  Form 2: the words must equal the serial XOR / OR.
// ------------------------------------------------------------------------------------------------
__host__ __device__ static inline uint32_t st_hash(uint32_t seed, uint32_t tid, uint32_t i) {
    uint32_t x = seed * 0x9E3779B9u + tid * 0x85EBCA6Bu + i * 0xC2B2AE35u + 0x27D4EB2Fu;
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}
#define ST_ITERS 12u
__host__ __device__ static inline uint32_t st_trips(uint32_t seed, uint32_t tid) { return 1u + st_hash(seed, tid, 0xffffu) % ST_ITERS; }
__host__ __device__ static inline uint32_t st_n(uint32_t seed, uint32_t tid, uint32_t i) {
    const uint32_t h = st_hash(seed, tid, i);
    return (h & 3u) == 0u ? 0u : ((h >> 2) & 7u) == 7u ? 1u + ((h >> 5) % 100u) : 1u + ((h >> 5) & 3u);  // a quarter skips; mostly 1..4, some up to 100
}

__global__ __launch_bounds__(256) void k_selftest_atomics(int form, uint32_t seed, uint32_t* __restrict__ ctr, uint2* __restrict__ ranges,
                                                         unsigned long long* __restrict__ lds_out) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t trips = st_trips(seed, tid);
    __shared__ uint32_t sh_x[4];
    __shared__ unsigned long long sh_or[4][16];
    const uint32_t w = threadIdx.x >> 6;
    if (form == 2) {
        if (jk::lane_id() == 0u) sh_x[w] = 0u;
        if (jk::lane_id() < 16u) sh_or[w][jk::lane_id()] = 0ull;
        jk::wave_sync();
    }
    for (uint32_t i = 0u; i < ST_ITERS; i++) {
        if (i >= trips) break;  // lanes leave at different trips
        const uint32_t n = st_n(seed, tid, i);
        uint2 r = make_uint2(0u, 0u);
        if (n != 0u) {  // divergent skip
            if (form == 0) r = make_uint2(atomicAdd(ctr, n), n);
            else if (form == 1) r = make_uint2(jk::wave_bump(ctr, n), n);
            else {
                atomicXor(&sh_x[w], 1u << (n & 31u));
                atomicOr(&sh_or[w][st_hash(seed, tid, i + 100u) & 15u], 1ull << jk::lane_id());
            }
        }
        if (form != 2) ranges[(size_t)tid * ST_ITERS + i] = r;
    }
    if (form != 2) {
        for (uint32_t i = trips; i < ST_ITERS; i++) ranges[(size_t)tid * ST_ITERS + i] = make_uint2(0u, 0u);
    } else {
        jk::wave_sync();
        const uint32_t wave = tid >> 6;
        if (jk::lane_id() == 0u) lds_out[(size_t)wave * 17u + 16u] = sh_x[w];
        if (jk::lane_id() < 16u) lds_out[(size_t)wave * 17u + jk::lane_id()] = sh_or[w][jk::lane_id()];
    }
}

// returns 0 when the form behaves like a serial execution, else the number of violations (capped), < 0 on a runtime error
extern "C" int jh_selftest_atomics_launch(hipStream_t stream, int form, uint32_t seed, uint32_t n_waves) {
    if (form < 0 || form > 2 || n_waves == 0u || n_waves > 4096u) return -1;
    const uint32_t blocks = (n_waves + 3u) / 4u, threads = blocks * 256u, waves = blocks * 4u;
    uint32_t* d_ctr = nullptr; uint2* d_r = nullptr; unsigned long long* d_l = nullptr;
    const size_t rb = (size_t)threads * ST_ITERS * sizeof(uint2), lb = (size_t)waves * 17u * 8u;
    if (hipMalloc(&d_ctr, 4) != hipSuccess || hipMalloc(&d_r, rb) != hipSuccess || hipMalloc(&d_l, lb) != hipSuccess) return -2;
    int bad = 0;
    if (hipMemsetAsync(d_ctr, 0, 4, stream) != hipSuccess) bad = -2;
    if (!bad) {
        hipLaunchKernelGGL(k_selftest_atomics, dim3(blocks), dim3(256), 0, stream, form, seed, d_ctr, d_r, d_l);
        if (hipGetLastError() != hipSuccess) bad = -2;
    }
    std::vector<uint2> r((size_t)threads * ST_ITERS);
    std::vector<unsigned long long> l((size_t)waves * 17u);
    uint32_t total = 0u;
    if (!bad && (hipMemcpyAsync(r.data(), d_r, rb, hipMemcpyDeviceToHost, stream) != hipSuccess ||
                 hipMemcpyAsync(l.data(), d_l, lb, hipMemcpyDeviceToHost, stream) != hipSuccess ||
                 hipMemcpyAsync(&total, d_ctr, 4, hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess))
        bad = -2;
    (void)hipFree(d_ctr); (void)hipFree(d_r); (void)hipFree(d_l);
    if (bad) return bad;
    if (form != 2) {
        uint64_t want = 0;
        std::vector<std::pair<uint32_t, uint32_t>> got;
        for (uint32_t t = 0; t < threads; t++)
            for (uint32_t i = 0; i < ST_ITERS; i++) {
                const uint32_t n = i < st_trips(seed, t) ? st_n(seed, t, i) : 0u;
                const uint2 g = r[(size_t)t * ST_ITERS + i];
                want += n;
                if (g.y != n) bad++;
                if (n) got.push_back({g.x, n});
            }
        if (want != total) bad++;
        std::sort(got.begin(), got.end());
        uint64_t at = 0;
        for (auto& g : got) { if (g.first != at) bad++; at = (uint64_t)g.first + g.second; }
        if (at != total) bad++;
    } else {
        for (uint32_t wv = 0; wv < waves; wv++) {
            uint32_t x = 0u; unsigned long long o[16] = {0};
            for (uint32_t ln = 0; ln < 64u; ln++) {
                const uint32_t t = wv * 64u + ln;
                for (uint32_t i = 0; i < st_trips(seed, t); i++) {
                    const uint32_t n = st_n(seed, t, i);
                    if (n) { x ^= 1u << (n & 31u); o[st_hash(seed, t, i + 100u) & 15u] |= 1ull << ln; }
                }
            }
            if (l[(size_t)wv * 17u + 16u] != x) bad++;
            for (int k = 0; k < 16; k++) if (l[(size_t)wv * 17u + k] != o[k]) bad++;
        }
    }
    return bad > 1000000 ? 1000000 : bad;
}
