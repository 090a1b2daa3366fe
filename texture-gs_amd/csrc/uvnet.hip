// Fused UV-Taylor producer: uvs = phi(mu) and the 3x3 Jacobian d phi_i / d x_j for every Gaussian
// in ONE kernel -- the two operator inputs the reference obtains from UVNet.forward (models/modules/uv_net.py:19-36) and
// from three extra backward passes through it (torch.autograd.functional.jacobian, models/texture_gaussian3d.py:216-227).
//
// Network (the shipped configs' shape, nn.Linear semantics of models/modules/utils.py:43-54):
//     h1 = relu(W1 x + b1)                 3 -> 128
//     a  = relu(W2 h1 + b2 + emb)          128 -> 128, then the geometry embedding is added (uv_net.py:31-33)
//     h2 = relu(W3 a + b3);  h3 = relu(W4 h2 + b4)      128 -> 128 twice
//     o  = W5 h3 + b5;  uv = o / max(|o|, 1e-12)        128 -> 3, F.normalize
// Jacobian by FORWARD mode: the three tangents d/dx_j ride along as three more columns per point, so every 128x128
// layer is one GEMM  Y[128 x 4P] = W[128 x 128] X[128 x 4P]  (value | d/dx0 | d/dx1 | d/dx2), ReLU masks taken from the value
// column.  This is the one dense contraction next to the hot path, so it runs on the matrix cores: fp32-in / fp32-accumulate
// v_mfma_f32_32x32x2_f32 (exact f32; a bf16 MFMA would put 1e-2 into J).  One workgroup = 32 points = a 128 x 128
// activation tile in LDS; wave w owns rows [32w, 32w+32) of every layer: its 32 x 128 slice of W stays in 64 VGPRs
// (pre-packed in MFMA A-operand order by k_uv_pack, so the loads are coalesced), the four 32x32 accumulators
// (value, three tangents) in 64 more.  3 layers x 4 tiles x 64 k-steps = 768 MFMAs per wave, 64 cycles each.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int UV_H = 128;          // hidden width
constexpr int UV_P = 32;           // points per workgroup

struct UVArgs {
    const float *W1, *b1, *b2, *emb, *b3, *b4, *W5, *b5, *off, *scale;
    const float* packed;           // [3 layers][4 bands][64 steps][64 lanes]
};

// W (row-major [128][128]) -> A-operand order of mfma_f32_32x32x2f32: lane l of band b at step s holds W[32b + (l & 31)][2s + (l >> 5)]
__global__ void __launch_bounds__(256)
k_uv_pack(const float* __restrict__ W2, const float* __restrict__ W3, const float* __restrict__ W4, float* __restrict__ packed) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= 3 * 4 * 64 * 64) return;
    const int lane = idx & 63, s = (idx >> 6) & 63, band = (idx >> 12) & 3, layer = idx >> 14;
    const float* W = layer == 0 ? W2 : (layer == 1 ? W3 : W4);
    packed[idx] = W[(band * 32 + (lane & 31)) * UV_H + 2 * s + (lane >> 5)];
}

__global__ void __launch_bounds__(256, 2)
k_uv_taylor(UVArgs a, const float* __restrict__ xyz, int N, float* __restrict__ uvs, float* __restrict__ J) {
    __shared__ float sX[UV_H][4 * UV_P];          // activations: row = neuron, col = plane * 32 + point (plane 0 = value)
    __shared__ float sO[3][4 * UV_P];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p0 = blockIdx.x * UV_P;
    // ---- layer 1 (3 -> 128) on the VALU, with the optional input normalisation (uv_net.py:22-25)
    for (int e = tid; e < UV_H * UV_P; e += 256) {
        const int i = e >> 5, p = e & 31, n = min(p0 + p, N - 1);
        float x[3], inv[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            inv[c] = a.scale ? 1.0f / a.scale[c] : 1.0f;
            x[c] = (xyz[3 * n + c] - (a.off ? a.off[c] : 0.0f)) * inv[c];
        }
        const float w0 = a.W1[3 * i], w1 = a.W1[3 * i + 1], w2 = a.W1[3 * i + 2];
        const float pre = w0 * x[0] + w1 * x[1] + w2 * x[2] + (a.b1 ? a.b1[i] : 0.0f);
        const bool on = pre > 0.0f;
        sX[i][p] = on ? pre : 0.0f;
        sX[i][UV_P + p] = on ? w0 * inv[0] : 0.0f;
        sX[i][2 * UV_P + p] = on ? w1 * inv[1] : 0.0f;
        sX[i][3 * UV_P + p] = on ? w2 * inv[2] : 0.0f;
    }
    __syncthreads();
    // ---- three 128 x 128 layers on the matrix cores
    const int bn = lane & 31, bk = lane >> 5;
    for (int layer = 0; layer < 3; ++layer) {
        float areg[64];
        const float* __restrict__ pk = a.packed + ((size_t)(layer * 4 + wave) * 64) * 64 + lane;
#pragma unroll
        for (int s = 0; s < 64; ++s) areg[s] = pk[s * 64];
        f32x16 acc0 = {0.f}, acc1 = {0.f}, acc2 = {0.f}, acc3 = {0.f};
#pragma unroll
        for (int s = 0; s < 64; ++s) {
            const float* row = &sX[2 * s + bk][bn];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[s], row[0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[s], row[UV_P], acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[s], row[2 * UV_P], acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[s], row[3 * UV_P], acc3, 0, 0, 0);
        }
        __syncthreads();                           // every wave has read the layer's input
        const float* bias = layer == 0 ? a.b2 : (layer == 1 ? a.b3 : a.b4);
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int i = wave * 32 + (v & 3) + 8 * (v >> 2) + 4 * bk;          // C/D layout: col = lane & 31
            float val = acc0[v] + (bias ? bias[i] : 0.0f);
            if (layer == 0) val += a.emb[i];
            const bool on = val > 0.0f;
            sX[i][bn] = on ? val : 0.0f;
            sX[i][UV_P + bn] = on ? acc1[v] : 0.0f;
            sX[i][2 * UV_P + bn] = on ? acc2[v] : 0.0f;
            sX[i][3 * UV_P + bn] = on ? acc3[v] : 0.0f;
        }
        __syncthreads();
    }
    // ---- output layer (128 -> 3) for the value and the three tangents
    for (int e = tid; e < 3 * 4 * UV_P; e += 256) {
        const int c = e >> 7, col = e & 127;
        float o = (col < UV_P && a.b5) ? a.b5[c] : 0.0f;
        for (int i = 0; i < UV_H; ++i) o += a.W5[c * UV_H + i] * sX[i][col];
        sO[c][col] = o;
    }
    __syncthreads();
    if (tid < UV_P && p0 + tid < N) {
        const int p = tid, n = p0 + p;
        const float o0 = sO[0][p], o1 = sO[1][p], o2 = sO[2][p];
        const float rn = 1.0f / fmaxf(sqrtf(o0 * o0 + o1 * o1 + o2 * o2), 1e-12f);
        const float u0 = o0 * rn, u1 = o1 * rn, u2 = o2 * rn;
        uvs[3 * n] = u0; uvs[3 * n + 1] = u1; uvs[3 * n + 2] = u2;
#pragma unroll
        for (int j = 0; j < 3; ++j) {                 // d(o/|o|) = (I - u u^T) do / |o|;  J[n][3i + j] = d uv_i / d x_j
            const float d0 = sO[0][(j + 1) * UV_P + p], d1 = sO[1][(j + 1) * UV_P + p], d2 = sO[2][(j + 1) * UV_P + p];
            const float ud = u0 * d0 + u1 * d1 + u2 * d2;
            J[9 * n + j] = (d0 - u0 * ud) * rn;
            J[9 * n + 3 + j] = (d1 - u1 * ud) * rn;
            J[9 * n + 6 + j] = (d2 - u2 * ud) * rn;
        }
    }
}


// ------------------------------------------------------------------------------------------------ split-bf16 variant (round 5)
// The same network on the bf16 matrix-core rate (16x the f32-input MFMA rate on gfx950) at near-f32 accuracy: every operand is
// split x = hi + lo into two bf16 (8 + 8 significand bits) and a product is taken as  Wh Xh + Wh Xl + Wl Xh  -- three
// v_mfma_f32_32x32x16_bf16 with f32 accumulation; the dropped Wl Xl term and the residual of the split are ~2^-17 relative.
// Opt-in (TexGSUVNet callers pick texgs_uv_taylor_packed_bf16x3): the f32 kernel above stays the checked default.
//   * activations live in LDS ALREADY SPLIT and k-contiguous, sH / sL[column][neuron] (bf16): a B operand (8 consecutive k of one
//     column) is one 16-byte read; the epilogue of a layer converts each output once and writes four neurons (8 bytes) at a time;
//   * rows are padded to 272 bytes: the 32 lanes of a half-wave start 17 x 16 bytes apart -> conflict-free 16-byte reads;
//   * which k the hardware assigns to (lane >> 5, element j) does not matter: A and B are loaded with the SAME assignment
//     (k = 16 s + 8 (lane >> 5) + j), and the instruction pairs equal (lane >> 5, j) of A and B.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int UV_PITCH = 136;      // bf16 per activation row: 128 neurons + 8 pad (272 bytes)

struct UVArgsB {
    const float *W1, *b1, *b2, *emb, *b3, *b4, *W5, *b5, *off, *scale;
    const uint4* packed;           // [3 layers][4 bands][8 k-steps][2: hi, lo][64 lanes] x 8 bf16
};

__device__ __forceinline__ void split_bf16(float x, __bf16& hi, __bf16& lo) {
    hi = (__bf16)x;
    lo = (__bf16)(x - (float)hi);
}

// W (row-major [128][128]) -> A-operand order of mfma_f32_32x32x16_bf16, split: lane l of band b at k-step s holds
// W[32 b + (l & 31)][16 s + 8 (l >> 5) + j], j = 0..7
__global__ void __launch_bounds__(256)
k_uv_pack_bf16x3(const float* __restrict__ W2, const float* __restrict__ W3, const float* __restrict__ W4, uint4* __restrict__ packed) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= 3 * 4 * 8 * 64) return;
    const int lane = idx & 63, s = (idx >> 6) & 7, band = (idx >> 9) & 3, layer = idx >> 11;
    const float* W = layer == 0 ? W2 : (layer == 1 ? W3 : W4);
    const float* row = W + (band * 32 + (lane & 31)) * UV_H + 16 * s + 8 * (lane >> 5);
    bf16x8 h, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) { __bf16 a, b; split_bf16(row[j], a, b); h[j] = a; l[j] = b; }
    uint4* o = packed + ((size_t)((layer * 4 + band) * 8 + s) * 2) * 64 + lane;
    o[0] = __builtin_bit_cast(uint4, h);
    o[64] = __builtin_bit_cast(uint4, l);
}

__global__ void __launch_bounds__(256, 2)
k_uv_taylor_bf16x3(UVArgsB a, const float* __restrict__ xyz, int N, float* __restrict__ uvs, float* __restrict__ J) {
    __shared__ __attribute__((aligned(16))) __bf16 sH[4 * UV_P][UV_PITCH];     // activations, high halves: [column = plane * 32 + point][neuron]
    __shared__ __attribute__((aligned(16))) __bf16 sL[4 * UV_P][UV_PITCH];     // low halves
    __shared__ float sO[3][4 * UV_P];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p0 = blockIdx.x * UV_P;
    auto put = [&](int col, int i, float v) { __bf16 h, l; split_bf16(v, h, l); sH[col][i] = h; sL[col][i] = l; };
    // ---- layer 1 (3 -> 128) on the VALU, with the optional input normalisation (uv_net.py:22-25)
    for (int e = tid; e < UV_H * UV_P; e += 256) {
        const int p = e >> 7, i = e & 127, n = min(p0 + p, N - 1);        // (neuron fastest: consecutive lanes write consecutive bf16)
        float x[3], inv[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            inv[c] = a.scale ? 1.0f / a.scale[c] : 1.0f;
            x[c] = (xyz[3 * n + c] - (a.off ? a.off[c] : 0.0f)) * inv[c];
        }
        const float w0 = a.W1[3 * i], w1 = a.W1[3 * i + 1], w2 = a.W1[3 * i + 2];
        const float pre = w0 * x[0] + w1 * x[1] + w2 * x[2] + (a.b1 ? a.b1[i] : 0.0f);
        const bool on = pre > 0.0f;
        put(p, i, on ? pre : 0.0f);
        put(UV_P + p, i, on ? w0 * inv[0] : 0.0f);
        put(2 * UV_P + p, i, on ? w1 * inv[1] : 0.0f);
        put(3 * UV_P + p, i, on ? w2 * inv[2] : 0.0f);
    }
    __syncthreads();
    // ---- three 128 x 128 layers on the matrix cores, three bf16 products per f32 product
    const int bn = lane & 31, bk = lane >> 5;
    for (int layer = 0; layer < 3; ++layer) {
        bf16x8 ah[8], al[8];
        const uint4* __restrict__ pk = a.packed + ((size_t)((layer * 4 + wave) * 8) * 2) * 64 + lane;
#pragma unroll
        for (int s = 0; s < 8; ++s) { ah[s] = __builtin_bit_cast(bf16x8, pk[(2 * s) * 64]); al[s] = __builtin_bit_cast(bf16x8, pk[(2 * s + 1) * 64]); }
        f32x16 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = f32x16{0.f};
#pragma unroll
        for (int s = 0; s < 8; ++s) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(&sH[t * UV_P + bn][16 * s + 8 * bk]);
                const bf16x8 bl = *reinterpret_cast<const bf16x8*>(&sL[t * UV_P + bn][16 * s + 8 * bk]);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s], bh, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s], bl, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[s], bh, acc[t], 0, 0, 0);
            }
        }
        __syncthreads();                           // every wave has read the layer's input
        const float* bias = layer == 0 ? a.b2 : (layer == 1 ? a.b3 : a.b4);
#pragma unroll
        for (int q = 0; q < 4; ++q) {              // C/D layout: col = lane & 31, rows (v & 3) + 8 (v >> 2) + 4 (lane >> 5): four consecutive per q
            const int i0 = wave * 32 + 8 * q + 4 * bk;
            bool on[4];
            bf16x4 h, l;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float val = acc[0][4 * q + r] + (bias ? bias[i0 + r] : 0.0f);
                if (layer == 0) val += a.emb[i0 + r];
                on[r] = val > 0.0f;
                __bf16 x, y; split_bf16(on[r] ? val : 0.0f, x, y); h[r] = x; l[r] = y;
            }
            *reinterpret_cast<bf16x4*>(&sH[bn][i0]) = h; *reinterpret_cast<bf16x4*>(&sL[bn][i0]) = l;
#pragma unroll
            for (int t = 1; t < 4; ++t) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { __bf16 x, y; split_bf16(on[r] ? acc[t][4 * q + r] : 0.0f, x, y); h[r] = x; l[r] = y; }
                *reinterpret_cast<bf16x4*>(&sH[t * UV_P + bn][i0]) = h; *reinterpret_cast<bf16x4*>(&sL[t * UV_P + bn][i0]) = l;
            }
        }
        __syncthreads();
    }
    // ---- output layer (128 -> 3) for the value and the three tangents, f32 on the re-joined halves
    for (int e = tid; e < 3 * 4 * UV_P; e += 256) {
        const int c = e >> 7, col = e & 127;
        float o = (col < UV_P && a.b5) ? a.b5[c] : 0.0f;
        for (int i = 0; i < UV_H; i += 8) {
            const bf16x8 h = *reinterpret_cast<const bf16x8*>(&sH[col][i]), l = *reinterpret_cast<const bf16x8*>(&sL[col][i]);
#pragma unroll
            for (int j = 0; j < 8; ++j) o += a.W5[c * UV_H + i + j] * ((float)h[j] + (float)l[j]);
        }
        sO[c][col] = o;
    }
    __syncthreads();
    if (tid < UV_P && p0 + tid < N) {
        const int p = tid, n = p0 + p;
        const float o0 = sO[0][p], o1 = sO[1][p], o2 = sO[2][p];
        const float rn = 1.0f / fmaxf(sqrtf(o0 * o0 + o1 * o1 + o2 * o2), 1e-12f);
        const float u0 = o0 * rn, u1 = o1 * rn, u2 = o2 * rn;
        uvs[3 * n] = u0; uvs[3 * n + 1] = u1; uvs[3 * n + 2] = u2;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float d0 = sO[0][(j + 1) * UV_P + p], d1 = sO[1][(j + 1) * UV_P + p], d2 = sO[2][(j + 1) * UV_P + p];
            const float ud = u0 * d0 + u1 * d1 + u2 * d2;
            J[9 * n + j] = (d0 - u0 * ud) * rn;
            J[9 * n + 3 + j] = (d1 - u1 * ud) * rn;
            J[9 * n + 6 + j] = (d2 - u2 * ud) * rn;
        }
    }
}

}  // namespace

size_t uv_taylor_temp_bytes() { return (size_t)3 * UV_H * UV_H * sizeof(float); }

// W2, W3, W4 -> MFMA A-operand order.  The result depends on the weights only: callers that evaluate the same network again
// (every view of a retexture / viewer session; every step between two optimizer updates) pack once and reuse `packed`.
int launch_uv_pack(const TexGSUVNet* net, void* packed, hipStream_t s) {
    hipLaunchKernelGGL(k_uv_pack, dim3(3 * 4 * 64 * 64 / 256), dim3(256), 0, s, net->W2, net->W3, net->W4, reinterpret_cast<float*>(packed));
    return (int)hipGetLastError();
}

int launch_uv_taylor_packed(const TexGSUVNet* net, const void* packed, const float* xyz, int N, float* uvs, float* grad_uvs,
                            hipStream_t s) {
    if (N <= 0) return 0;
    UVArgs a;
    a.W1 = net->W1; a.b1 = net->b1; a.b2 = net->b2; a.emb = net->emb; a.b3 = net->b3; a.b4 = net->b4; a.W5 = net->W5; a.b5 = net->b5;
    a.off = net->xyz_offset; a.scale = net->xyz_scale; a.packed = reinterpret_cast<const float*>(packed);
    hipLaunchKernelGGL(k_uv_taylor, dim3((N + UV_P - 1) / UV_P), dim3(256), 0, s, a, xyz, N, uvs, grad_uvs);
    return (int)hipGetLastError();
}

int launch_uv_taylor(const TexGSUVNet* net, const float* xyz, int N, float* uvs, float* grad_uvs, void* temp, hipStream_t s) {
    if (N <= 0) return 0;
    if (int r = launch_uv_pack(net, temp, s)) return r;
    return launch_uv_taylor_packed(net, temp, xyz, N, uvs, grad_uvs, s);
}

// split-bf16 variant: the same two steps (the packed buffer has the same size, a different layout)
int launch_uv_pack_bf16x3(const TexGSUVNet* net, void* packed, hipStream_t s) {
    hipLaunchKernelGGL(k_uv_pack_bf16x3, dim3(3 * 4 * 8 * 64 / 256), dim3(256), 0, s, net->W2, net->W3, net->W4, reinterpret_cast<uint4*>(packed));
    return (int)hipGetLastError();
}

int launch_uv_taylor_packed_bf16x3(const TexGSUVNet* net, const void* packed, const float* xyz, int N, float* uvs, float* grad_uvs,
                                   hipStream_t s) {
    if (N <= 0) return 0;
    UVArgsB a;
    a.W1 = net->W1; a.b1 = net->b1; a.b2 = net->b2; a.emb = net->emb; a.b3 = net->b3; a.b4 = net->b4; a.W5 = net->W5; a.b5 = net->b5;
    a.off = net->xyz_offset; a.scale = net->xyz_scale; a.packed = reinterpret_cast<const uint4*>(packed);
    hipLaunchKernelGGL(k_uv_taylor_bf16x3, dim3((N + UV_P - 1) / UV_P), dim3(256), 0, s, a, xyz, N, uvs, grad_uvs);
    return (int)hipGetLastError();
}
