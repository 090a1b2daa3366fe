// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of the blend kernels:
// kernels that move a KNOWN number of bytes past the L2, one pattern each.  Run under `rocprofv3 --pmc FETCH_SIZE` and again
// under `--pmc WRITE_SIZE` (scripts/traffic_cal.sh); scripts/traffic_cal_summary.py divides counter by known bytes.
//   k_stream_read16    wide coalesced streaming read, 16 B per lane (the guide's reference pattern: FETCH_SIZE reports 1/2)
//   k_gather12         12-byte (dwordx3) gathers at random texel offsets of a 3 GiB array, every gather in its own 128-B line:
//                      the texture taps of K6 / K7 when they miss (each miss must bring in at least one 64-B sector)
//   k_stream_write4    coalesced 4-byte stores (the record planes of K7)
//   k_atomic4_scatter  4-byte fire-and-forget fp32 atomics to random, distinct lines (accumulator rows / border footprints)
//   k_atomic_rows      16 lanes x 4 B atomics on one 64-B run, two runs per 128-B row (K7's accumulator-row update)
// The arrays are far larger than L2 (32 MiB) + Infinity Cache (256 MiB) and every line is touched once, so every access misses.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct __attribute__((packed, aligned(4))) F3 { float x, y, z; };

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16; return x; }

__global__ void k_stream_read16(const float4* __restrict__ src, size_t n4, float* __restrict__ sink) {
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) { const float4 v = src[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) sink[0] = acc;
}
// gather g reads 12 bytes at the start (+ a random multiple of 12 below 116) of line perm(g): lines are visited once, in a scrambled order
__global__ void k_gather12(const char* __restrict__ src, uint32_t nlines, uint32_t ngather, float* __restrict__ sink) {
    float acc = 0.f;
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < ngather; g += gridDim.x * blockDim.x) {
        const uint32_t line = (uint32_t)(((uint64_t)g * 2654435761ull) % nlines);          // odd multiplier: a permutation when nlines is a power of two
        const uint32_t off = (hash32(g) % 9u) * 12u;
        const F3 v = *reinterpret_cast<const F3*>(src + (size_t)line * 128u + off);
        acc += v.x + v.y + v.z;
    }
    if (acc == 123.456f) sink[0] = acc;
}
__global__ void k_stream_write4(float* __restrict__ dst, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = (float)i;
}
__global__ void k_atomic4_scatter(float* __restrict__ dst, uint32_t nlines, uint32_t natomic) {
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < natomic; g += gridDim.x * blockDim.x) {
        const uint32_t line = (uint32_t)(((uint64_t)g * 2654435761ull) % nlines);
        unsafeAtomicAdd(dst + (size_t)line * 32u + (hash32(g) & 31u), 1.0f);
    }
}
__global__ void k_atomic_rows(float* __restrict__ dst, uint32_t nrows, uint32_t ntask) {
    // one task = 16 lanes adding 2 x 16 consecutive floats of one 128-B row (K7 stage C2); 4 tasks per wave
    const uint32_t lane = threadIdx.x & 63u, sub = lane & 15u;
    for (uint32_t t = (blockIdx.x * blockDim.x + threadIdx.x) >> 4; t < ntask; t += (gridDim.x * blockDim.x) >> 4) {
        const uint32_t row = (uint32_t)(((uint64_t)t * 2654435761ull) % nrows);
        float* p = dst + (size_t)row * 32u + sub;
        unsafeAtomicAdd(p, 1.0f);
        unsafeAtomicAdd(p + 16, 1.0f);
    }
}

int main() {
    const size_t BYTES = 3ull << 30;                       // 3 GiB: 12x the Infinity Cache
    const uint32_t NLINES = (uint32_t)(BYTES / 128);       // 25 165 824 lines (not a power of two: the multiplier is still coprime to it)
    char* buf; float* sink;
    CHECK(hipMalloc(&buf, BYTES)); CHECK(hipMalloc(&sink, 256));
    CHECK(hipMemset(buf, 0, BYTES));
    const uint32_t NG = 16u << 20;                          // 16.8 M gathers / atomics: each in its own line
    const dim3 grid(256 * 16), blk(256);
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto timed = [&](const char* name, double known_bytes, auto launch) {
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0)); launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms = 0.f; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("{\"kernel\": \"%s\", \"known_MB\": %.3f, \"ms\": %.4f, \"GBps\": %.1f}\n", name, known_bytes / 1e6, ms, known_bytes / (ms * 1e-3) / 1e9);
    };
    timed("k_stream_read16", (double)BYTES, [&] { hipLaunchKernelGGL(k_stream_read16, grid, blk, 0, 0, (const float4*)buf, BYTES / 16, sink); });
    timed("k_gather12", 12.0 * NG, [&] { hipLaunchKernelGGL(k_gather12, grid, blk, 0, 0, (const char*)buf, NLINES, NG, sink); });
    timed("k_stream_write4", (double)(1ull << 30), [&] { hipLaunchKernelGGL(k_stream_write4, grid, blk, 0, 0, (float*)buf, (size_t)(1ull << 28)); });
    CHECK(hipMemset(buf, 0, BYTES));
    timed("k_atomic4_scatter", 4.0 * NG, [&] { hipLaunchKernelGGL(k_atomic4_scatter, grid, blk, 0, 0, (float*)buf, NLINES, NG); });
    timed("k_atomic_rows", 128.0 * (NG / 4), [&] { hipLaunchKernelGGL(k_atomic_rows, grid, blk, 0, 0, (float*)buf, NLINES, NG / 4); });
    CHECK(hipDeviceSynchronize());
    return 0;
}
