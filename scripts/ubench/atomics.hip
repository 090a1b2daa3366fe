// Microbenchmark: fp32 global atomic throughput on gfx950 for three lane->address patterns.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// mode 0: each lane: 12 atomics = 4 random texels x 3 channels (today's K7 pattern: per-lane taps)
// mode 1: groups of 6 adjacent lanes write 6 adjacent dwords (one row of 2 taps), random rows
// mode 2: 64 lanes -> 64 consecutive dwords at a random base (coalesced flush)
// mode 3: like 0 but texels drawn from a small 64 KB window (hot set)
template <int MODE>
__global__ void k(float* buf, uint32_t ndw, int iters, uint32_t seed) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63, wave = gid >> 6;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 3) {
            uint32_t h = hash(gid * 977u + it * 131071u + seed);
            const uint32_t ntex = (MODE == 3 ? 16384u : ndw / 3u) - 4u;
            for (int t = 0; t < 4; ++t) {
                const uint32_t tx = (hash(h + t) % ntex) * 3u;
                for (int c = 0; c < 3; ++c) unsafeAtomicAdd(buf + tx + c, 1.0f);
            }
        } else if (MODE == 1) {
            // 12 instructions, each: lane -> (group = lane/6, dword = lane%6); groups hit random rows
            for (int q = 0; q < 12; ++q) {
                const uint32_t grp = lane / 6u;
                const uint32_t row = hash(wave * 7919u + it * 131u + q * 17u + grp + seed) % (ndw / 3u - 8u);
                if (lane < 60) unsafeAtomicAdd(buf + row * 3u + (lane % 6u), 1.0f);
            }
        } else {
            for (int q = 0; q < 12; ++q) {
                const uint32_t base = hash(wave * 7919u + it * 131u + q * 17u + seed) % (ndw - 64u);
                unsafeAtomicAdd(buf + base + lane, 1.0f);
            }
        }
    }
}

template <int MODE>
void run(float* d, uint32_t ndw, const char* name) {
    const int blocks = 4096, threads = 256, iters = 8;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, ndw, 1, 1u);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, ndw, iters, 7u);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    double lanes = (double)blocks * threads * iters * 12.0 * (MODE == 1 ? 60.0 / 64.0 : 1.0);
    printf("%-44s %8.3f ms  %8.2f G lane-atomics/s  %8.2f G wave-instr/s\n", name, ms, lanes / ms / 1e6,
           (double)blocks * threads / 64 * iters * 12.0 / ms / 1e6);
}

int main() {
    const uint32_t ndw = 6u * 1024u * 1024u * 3u;     // the C3 texture-grad buffer (75.5 MB)
    float* d; CK(hipMalloc(&d, (size_t)ndw * 4)); CK(hipMemset(d, 0, (size_t)ndw * 4));
    run<0>(d, ndw, "per-lane 4 random texels x3ch (scatter)");
    run<3>(d, ndw, "same, texels from a 192 KB hot window");
    run<1>(d, ndw, "6 adjacent lanes -> 6 adjacent dwords");
    run<2>(d, ndw, "64 lanes -> 64 consecutive dwords");
    return 0;
}
