#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include "../../texture-gs_amd/csrc/wave_ops.h"
__global__ void k(float* out) {
    const int l = threadIdx.x;
    const float v = (float)(l * l % 17) + 0.25f * l;
    out[l] = lane_xor4(v) - __shfl_xor(v, 4, 64);
    out[64 + l] = lane_xor8(v) - __shfl_xor(v, 8, 64);
    out[128 + l] = sum_xor16(v) - (v + __shfl_xor(v, 16, 64));
    out[192 + l] = sum_xor32(v) - (v + __shfl_xor(v, 32, 64));
    float s = v; for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    out[256 + l] = wave_sum(v) - s;
    float a16[16], a32[32];
    for (int i = 0; i < 16; ++i) a16[i] = (float)((l * 7 + i * 13) % 11) - 3.0f;
    for (int i = 0; i < 32; ++i) a32[i] = (float)((l * 5 + i * 3) % 19) - 7.0f;
    float ref16 = 0, ref32 = 0;
    {   // reference: full reduction of element (l & 15) / (l & 31)
        for (int i = 0; i < 16; ++i) { float t = a16[i]; for (int m = 32; m >= 1; m >>= 1) t += __shfl_xor(t, m, 64); if (i == (l & 15)) ref16 = t; }
        for (int i = 0; i < 32; ++i) { float t = a32[i]; for (int m = 32; m >= 1; m >>= 1) t += __shfl_xor(t, m, 64); if (i == (l & 31)) ref32 = t; }
    }
    out[320 + l] = reduce_transposed<16>(a16, l) - ref16;
    out[384 + l] = reduce_transposed<32>(a32, l) - ref32;
    {   float t32[32]; for (int i = 0; i < 32; ++i) t32[i] = a32[i];
        float refb = 0; const int want = transposed_index(l);
        for (int i = 0; i < 32; ++i) { float t = a32[i]; for (int m = 32; m >= 1; m >>= 1) t += __shfl_xor(t, m, 64); if (i == want) refb = t; }
        out[448 + l] = reduce32_bankfirst(t32, l) - refb; }
}
int main() {
    float* d; hipMalloc(&d, 512 * 4); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[8] = {"xor4", "xor8", "sum_xor16", "sum_xor32", "wave_sum", "reduce16T", "reduce32T", "reduce32BF"};
    int bad = 0;
    for (int t = 0; t < 8; ++t) { float m = 0; for (int l = 0; l < 64; ++l) m = fmaxf(m, fabsf(h[t * 64 + l])); printf("%-10s max|diff| %g\n", names[t], m); bad += m != 0.f; }
    printf(bad ? "FAIL\n" : "ALL OK\n");
    return bad;
}
