// LDS atomic throughput vs same-address multiplicity (gfx950).  One workgroup per CU-ish, 256 threads; every lane issues
// ITER x 12 ds_add_f32 to address (lane / K) * STRIDE (+ offset k): K lanes of a wave-instruction share each address.
// Prints cycles per wave-instruction per CU (wall time * clock / instructions per CU).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int K, int INTEGER>
__global__ void __launch_bounds__(256) k(float* out, int iters, int spread) {
    __shared__ float s[64 * 33 * 3];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 64 * 33 * 3; i += 256) s[i] = 0.f;
    __syncthreads();
    // address group: lanes {g*K .. g*K+K-1} share a base; distinct groups land on distinct banks when spread == 1
    const int g = lane / K;
    float* p = s + (g * spread) % (64 * 33 * 3 - 32);
    unsigned* pu = reinterpret_cast<unsigned*>(p);
    float v = 1.0f + tid;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            if (INTEGER == 2) atomicAdd(reinterpret_cast<unsigned long long*>(pu) + k, (unsigned long long)tid << 20);
            else if (INTEGER == 1) atomicAdd(pu + k, (unsigned)tid);
            else __hip_atomic_fetch_add(p + k, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    if (tid == 0) out[blockIdx.x] = s[0];
}

template <int K, int INTEGER>
void run(const char* name, int spread) {
    float* d; hipMalloc(&d, 4096 * 4);
    const int blocks = 256 * 4, iters = 2000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<K, INTEGER>), dim3(blocks), dim3(256), 0, 0, d, 10, spread);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<K, INTEGER>), dim3(blocks), dim3(256), 0, 0, d, iters, spread);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double instr_per_cu = (double)blocks * 4 /*waves*/ * iters * 12 / 256.0;
    printf("%-28s K=%2d spread=%3d : %8.3f ms  -> %6.1f cycles / wave-instruction / CU (2.4 GHz)\n", name, K, spread, ms,
           ms * 1e-3 * 2.4e9 / instr_per_cu);
    hipFree(d);
}

int main() {
    run<1, 0>("ds_add_f32 no sharing", 13);
    run<2, 0>("ds_add_f32", 13);
    run<4, 0>("ds_add_f32", 13);
    run<8, 0>("ds_add_f32", 13);
    run<16, 0>("ds_add_f32", 13);
    run<64, 0>("ds_add_f32 all lanes one addr", 13);
    run<1, 0>("ds_add_f32 bank-conflict x2", 64);
    run<1, 0>("ds_add_f32 stride 32 dwords", 32);
    run<1, 1>("ds_add_u32 no sharing", 13);
    run<8, 1>("ds_add_u32", 13);
    run<64, 1>("ds_add_u32 all lanes one addr", 13);
    run<1, 2>("ds_add_u64 no sharing", 26);
    run<2, 2>("ds_add_u64", 26);
    run<8, 2>("ds_add_u64", 26);
    run<64, 2>("ds_add_u64 all lanes one addr", 26);
    return 0;
}
