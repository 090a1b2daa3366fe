// What a streaming read reaches on this part, by load width, loads in flight per thread and waves per CU (the 3.2 TB/s of
// traffic_cal.hip is ONE 16-byte load per thread per trip).  hipcc --offload-arch=gfx950 -O3 stream_bw.hip -o stream_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int U>
__global__ void k_read16(const float4* __restrict__ src, size_t n4, float* __restrict__ sink) {
    float acc = 0.f;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = src[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (acc == 123.456f) sink[0] = acc;
}
template <int U>
__global__ void k_read4(const float* __restrict__ src, size_t n, float* __restrict__ sink) {
    float acc = 0.f;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = src[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    if (acc == 123.456f) sink[0] = acc;
}
// block-contiguous: every workgroup reads its own contiguous slab (what k_texgrad_reduce / K1 / K8 do), 4 B per lane
template <int U>
__global__ void k_read4_slab(const float* __restrict__ src, size_t n, float* __restrict__ sink) {
    float acc = 0.f;
    const size_t per = n / gridDim.x;
    const float* p = src + per * blockIdx.x;
    for (size_t i = threadIdx.x; i + (U - 1) * blockDim.x < per; i += U * blockDim.x) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = p[i + u * blockDim.x];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    if (acc == 123.456f) sink[0] = acc;
}
#define RUN(NAME, KERNEL, GRID, BLOCK, ...) do { \
    hipEventRecord(e0); for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(KERNEL, dim3(GRID), dim3(BLOCK), 0, 0, __VA_ARGS__); hipEventRecord(e1); hipEventSynchronize(e1); \
    float ms; hipEventElapsedTime(&ms, e0, e1); printf("%-34s grid %6d x %4d : %7.1f GB/s\n", NAME, (int)(GRID), (int)(BLOCK), 3.0 * BYTES / (ms * 1e6)); } while (0)
int main() {
    const size_t BYTES = 3ull << 30;
    char* buf; float* sink;
    hipMalloc(&buf, BYTES); hipMalloc(&sink, 4); hipMemset(buf, 0, BYTES);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t n4 = BYTES / 16, n = BYTES / 4;
    for (int g : {2048, 8192, 32768}) {
        RUN("16 B x1", k_read16<1>, g, 256, (const float4*)buf, n4, sink);
        RUN("16 B x4", k_read16<4>, g, 256, (const float4*)buf, n4, sink);
        RUN("16 B x8", k_read16<8>, g, 256, (const float4*)buf, n4, sink);
        RUN("4 B x1", k_read4<1>, g, 256, (const float*)buf, n, sink);
        RUN("4 B x4", k_read4<4>, g, 256, (const float*)buf, n, sink);
        RUN("4 B x10", k_read4<10>, g, 256, (const float*)buf, n, sink);
        RUN("4 B x10 slab", k_read4_slab<10>, g, 256, (const float*)buf, n, sink);
        RUN("4 B x10 slab 512 thr", k_read4_slab<10>, g, 512, (const float*)buf, n, sink);
    }
    return 0;
}
