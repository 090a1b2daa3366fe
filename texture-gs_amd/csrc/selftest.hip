// On-device self-test of csrc/wave_ops.h (the DPP / permlane primitives K7's reductions depend on), exported through
// the C ABI so that a `-m gpu` test pins it on real hardware: every primitive against the plain __shfl_xor formulation.
#include "common.h"
#include "wave_ops.h"

namespace {

__global__ void __launch_bounds__(64)
k_selftest_waveops(const float* __restrict__ seed, float* __restrict__ out) {
    const int l = threadIdx.x;
    const float v = seed[l];
    out[l] = lane_xor4(v) - __shfl_xor(v, 4, 64);
    out[64 + l] = lane_xor8(v) - __shfl_xor(v, 8, 64);
    out[128 + l] = sum_xor16(v) - (v + __shfl_xor(v, 16, 64));
    out[192 + l] = sum_xor32(v) - (v + __shfl_xor(v, 32, 64));
    float s = v;
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    {   // wave_sum adds in a different association order than the xor ladder: compare against the exact integer sum
        out[256 + l] = wave_sum(rintf(v)) - [&] { float t = rintf(v); for (int m = 32; m >= 1; m >>= 1) t += __shfl_xor(t, m, 64); return t; }();
    }
    float a16[16], a32[32], b32[32];
    for (int i = 0; i < 16; ++i) a16[i] = rintf(seed[64 + ((l * 7 + i * 13) & 63)] * 8.0f);      // small integers: sums are exact
    for (int i = 0; i < 32; ++i) { a32[i] = rintf(seed[64 + ((l * 5 + i * 3) & 63)] * 8.0f); b32[i] = a32[i]; }
    float ref16 = 0.f, ref32 = 0.f, refb = 0.f;
    const int want = transposed_index(l);
    for (int i = 0; i < 16; ++i) {
        float t = a16[i];
        for (int m = 32; m >= 1; m >>= 1) t += __shfl_xor(t, m, 64);
        if (i == (l & 15)) ref16 = t;
    }
    for (int i = 0; i < 32; ++i) {
        float t = a32[i];
        for (int m = 32; m >= 1; m >>= 1) t += __shfl_xor(t, m, 64);
        if (i == (l & 31)) ref32 = t;
        if (i == want) refb = t;
    }
    out[320 + l] = reduce_transposed<16>(a16, l) - ref16;
    out[384 + l] = reduce_transposed<32>(a32, l) - ref32;
    out[448 + l] = reduce32_bankfirst(b32, l) - refb;
    out[512 + l] = (float)wave_max_i((int)rintf(seed[l] * 100.0f)) -
                   [&] { int t = (int)rintf(seed[l] * 100.0f); for (int m = 32; m >= 1; m >>= 1) t = max(t, __shfl_xor(t, m, 64)); return (float)t; }();
}

}  // namespace

void launch_selftest_waveops(const float* seed128, float* out576, hipStream_t s) {
    hipLaunchKernelGGL(k_selftest_waveops, dim3(1), dim3(64), 0, s, seed128, out576);
}
