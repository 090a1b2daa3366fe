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

// The thin first layer, written with explicit roundings: every kernel of this file (f32, split-bf16, mixed, backward) forms the SAME
// f32 pre-activation from the same inputs, whatever the compiler would have contracted -- so the f32-MFMA value columns of the f32
// kernel, of the mixed kernel and of the backward's recomputation agree bit for bit, and with them the ReLU masks.
__device__ __forceinline__ float uv_norm_in(float x, float off, float inv) { return __fmul_rn(__fsub_rn(x, off), inv); }
__device__ __forceinline__ float uv_layer1_pre(float w0, float w1, float w2, float b, float x0, float x1, float x2) {
    return __fmaf_rn(w2, x2, __fmaf_rn(w1, x1, __fmaf_rn(w0, x0, b)));
}
// (bias, then the embedding: the order the f32 kernel's epilogue has always used)
__device__ __forceinline__ float uv_bias_emb(float acc, float bias, float emb) { return __fadd_rn(__fadd_rn(acc, bias), emb); }

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
            x[c] = uv_norm_in(xyz[3 * n + c], a.off ? a.off[c] : 0.0f, inv[c]);
        }
        const float w0 = a.W1[3 * i], w1 = a.W1[3 * i + 1], w2 = a.W1[3 * i + 2];
        const float pre = uv_layer1_pre(w0, w1, w2, a.b1 ? a.b1[i] : 0.0f, x[0], x[1], x[2]);
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
            const float val = uv_bias_emb(acc0[v], bias ? bias[i] : 0.0f, layer == 0 ? a.emb[i] : 0.0f);
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
            x[c] = uv_norm_in(xyz[3 * n + c], a.off ? a.off[c] : 0.0f, inv[c]);
        }
        const float w0 = a.W1[3 * i], w1 = a.W1[3 * i + 1], w2 = a.W1[3 * i + 2];
        const float pre = uv_layer1_pre(w0, w1, w2, a.b1 ? a.b1[i] : 0.0f, x[0], x[1], x[2]);
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
                const float val = uv_bias_emb(acc[0][4 * q + r], bias ? bias[i0 + r] : 0.0f, layer == 0 ? a.emb[i0 + r] : 0.0f);
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


// ------------------------------------------------------------------------------------------------ mixed variant (round 5)
// VALUE in f32, TANGENTS in split bf16.  The value column decides everything discrete -- the ReLU masks, and through uvs the
// texel a pixel samples -- so it stays on the f32-input MFMA exactly as in k_uv_taylor (same masks, same uvs to the last bit of
// the accumulation order); the three tangent columns only ever multiply pixel offsets of a few pixels, where 1e-5 relative is two
// orders below the operator's tolerance, and they are 3/4 of the work: they take the three-product bf16 path of
// k_uv_taylor_bf16x3.  MFMA cycles per wave and layer: 64 x 64 (value) + 72 x 32 (tangents) = 6 400 against 16 384 all-f32.
struct UVArgsM {
    const float *W1, *b1, *b2, *emb, *b3, *b4, *W5, *b5, *off, *scale;
    const float* packed_f32;       // k_uv_pack layout
    const uint4* packed_b16;       // k_uv_pack_bf16x3 layout
};

__global__ void __launch_bounds__(256, 2)
k_uv_taylor_mixed(UVArgsM a, const float* __restrict__ xyz, int N, float* __restrict__ uvs, float* __restrict__ J) {
    __shared__ float sV[UV_H][UV_P];                                            // value activations [neuron][point]
    __shared__ __attribute__((aligned(16))) __bf16 sH[3 * UV_P][UV_PITCH];     // tangent activations, high halves [plane * 32 + point][neuron]
    __shared__ __attribute__((aligned(16))) __bf16 sL[3 * UV_P][UV_PITCH];     // low halves
    __shared__ float sO[3][4 * UV_P];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p0 = blockIdx.x * UV_P;
    auto put = [&](int col, int i, float v) { __bf16 h, l; split_bf16(v, h, l); sH[col][i] = h; sL[col][i] = l; };
    auto layer1 = [&](int i, int p, float& pre, float (&t)[3]) {
        const int n = min(p0 + p, N - 1);
        float x[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float inv = a.scale ? 1.0f / a.scale[c] : 1.0f;
            x[c] = uv_norm_in(xyz[3 * n + c], a.off ? a.off[c] : 0.0f, inv);
            t[c] = a.W1[3 * i + c] * inv;
        }
        pre = uv_layer1_pre(a.W1[3 * i], a.W1[3 * i + 1], a.W1[3 * i + 2], a.b1 ? a.b1[i] : 0.0f, x[0], x[1], x[2]);
    };
    // ---- layer 1 (3 -> 128) on the VALU, twice: point-fastest for the f32 plane, neuron-fastest for the bf16 planes (each
    // layout wants its own lane order; the recomputation is five FMAs)
    for (int e = tid; e < UV_H * UV_P; e += 256) {
        const int i = e >> 5, p = e & 31;
        float pre, t[3];
        layer1(i, p, pre, t);
        sV[i][p] = pre > 0.0f ? pre : 0.0f;
    }
    for (int e = tid; e < UV_H * UV_P; e += 256) {
        const int p = e >> 7, i = e & 127;
        float pre, t[3];
        layer1(i, p, pre, t);
        const bool on = pre > 0.0f;
        put(p, i, on ? t[0] : 0.0f); put(UV_P + p, i, on ? t[1] : 0.0f); put(2 * UV_P + p, i, on ? t[2] : 0.0f);
    }
    __syncthreads();
    const int bn = lane & 31, bk = lane >> 5;
    for (int layer = 0; layer < 3; ++layer) {
        f32x16 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = f32x16{0.f};
        {   // value: f32-input MFMA, as k_uv_taylor
            float areg[64];
            const float* __restrict__ pk = a.packed_f32 + ((size_t)(layer * 4 + wave) * 64) * 64 + lane;
#pragma unroll
            for (int s = 0; s < 64; ++s) areg[s] = pk[s * 64];
#pragma unroll
            for (int s = 0; s < 64; ++s) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[s], sV[2 * s + bk][bn], acc[0], 0, 0, 0);
        }
        {   // tangents: three bf16 products per f32 product, as k_uv_taylor_bf16x3
            bf16x8 ah[8], al[8];
            const uint4* __restrict__ pk = a.packed_b16 + ((size_t)((layer * 4 + wave) * 8) * 2) * 64 + lane;
#pragma unroll
            for (int s = 0; s < 8; ++s) { ah[s] = __builtin_bit_cast(bf16x8, pk[(2 * s) * 64]); al[s] = __builtin_bit_cast(bf16x8, pk[(2 * s + 1) * 64]); }
#pragma unroll
            for (int s = 0; s < 8; ++s) {
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const bf16x8 bh = *reinterpret_cast<const bf16x8*>(&sH[t * UV_P + bn][16 * s + 8 * bk]);
                    const bf16x8 bl = *reinterpret_cast<const bf16x8*>(&sL[t * UV_P + bn][16 * s + 8 * bk]);
                    acc[1 + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s], bh, acc[1 + t], 0, 0, 0);
                    acc[1 + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s], bl, acc[1 + t], 0, 0, 0);
                    acc[1 + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[s], bh, acc[1 + t], 0, 0, 0);
                }
            }
        }
        __syncthreads();                           // every wave has read the layer's input
        const float* bias = layer == 0 ? a.b2 : (layer == 1 ? a.b3 : a.b4);
#pragma unroll
        for (int q = 0; q < 4; ++q) {              // C/D layout: col = lane & 31, rows (v & 3) + 8 (v >> 2) + 4 (lane >> 5): four consecutive per q
            const int i0 = wave * 32 + 8 * q + 4 * bk;
            bool on[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float val = uv_bias_emb(acc[0][4 * q + r], bias ? bias[i0 + r] : 0.0f, layer == 0 ? a.emb[i0 + r] : 0.0f);
                on[r] = val > 0.0f;
                sV[i0 + r][bn] = on[r] ? val : 0.0f;
            }
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                bf16x4 h, l;
#pragma unroll
                for (int r = 0; r < 4; ++r) { __bf16 x, y; split_bf16(on[r] ? acc[1 + t][4 * q + r] : 0.0f, x, y); h[r] = x; l[r] = y; }
                *reinterpret_cast<bf16x4*>(&sH[t * UV_P + bn][i0]) = h; *reinterpret_cast<bf16x4*>(&sL[t * UV_P + bn][i0]) = l;
            }
        }
        __syncthreads();
    }
    // ---- output layer (128 -> 3): the value column from the f32 plane, the tangents from the re-joined halves
    for (int e = tid; e < 3 * 4 * UV_P; e += 256) {
        const int c = e >> 7, col = e & 127;
        float o;
        if (col < UV_P) {
            o = a.b5 ? a.b5[c] : 0.0f;
            for (int i = 0; i < UV_H; ++i) o += a.W5[c * UV_H + i] * sV[i][col];
        } else {
            o = 0.0f;
            const int tc = col - UV_P;
            for (int i = 0; i < UV_H; i += 8) {
                const bf16x8 h = *reinterpret_cast<const bf16x8*>(&sH[tc][i]), l = *reinterpret_cast<const bf16x8*>(&sL[tc][i]);
#pragma unroll
                for (int j = 0; j < 8; ++j) o += a.W5[c * UV_H + i + j] * ((float)h[j] + (float)l[j]);
            }
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

// ------------------------------------------------------------------------------------------------ backward (round 5)
// Gradients of uvs = UVNet(xyz) w.r.t. every weight, bias and the embedding for an upstream gradient g [N,3]: what
// loss.backward() does through models/modules/uv_net.py:19-36 in the reference (autograd over five Linear layers + F.normalize).
// As library calls this is 14 tall-skinny GEMMs and ~30 elementwise passes over [N,128] tensors (5.5 ms of a 10 ms iteration at
// N = 300 000); here it is ONE persistent kernel that keeps a tile of 64 points in LDS from the recomputed forward to the last
// gradient and the three 128x128 weight gradients in registers across all of a workgroup's tiles:
//     forward   h1 = relu(W1 x + b1) (VALU) | a = relu(W2 h1 + b2 + emb) | h2 = relu(W3 a + b3) | h3 = relu(W4 h2 + b4)   3 GEMMs
//               o = W5 h3 + b5, u = o / |o|, do = (g - u (u.g)) / |o|                                                       VALU
//     backward  d4 = (W5^T do) . [h3 > 0]  -> overwrites h3 in LDS            dW5 += do h3^T, db5 += do, db4 += d4           VALU
//               d3 = (W4^T d4) . [h2 > 0]  -> overwrites h2                   dW4 += d4 h2^T                              2 GEMMs
//               d2 = (W3^T d3) . [a  > 0]  -> overwrites a                    dW3 += d3 a^T,  db3 += d3                   2 GEMMs
//               d1 = (W2^T d2) . [h1 > 0]  -> overwrites h1                   dW2 += d2 h1^T, db2 (= d emb) += d2         2 GEMMs
//               dW1 += d1 x^T, db1 += d1                                                                                    VALU
// (d phi / d xyz is not produced: the caller has the Jacobian from the forward launch, d xyz = J^T g.)
// All nine GEMMs are v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate), 1152 per wave and tile.  Wave w owns output rows
// [32 w, 32 w + 32): for W x and W^T d the A operand is its slice of the pre-packed (transposed) weights, 64 VGPRs, loaded one
// GEMM AHEAD into a second register set (one wave per SIMD: nothing else would hide the L2 round trip); for d h^T the A operand
// is the wave's rows of d and B the rows of h, both read point-major from LDS (row pitch 65 floats: conflict-free both ways).
// LDS: four activation planes [128][65] f32 = 133 KB -> one workgroup per CU, launched as a persistent grid of <= 256 workgroups;
// each writes its partial sums once (k_uv_backward_reduce adds the <= 256 partials: deterministic, no atomics).
constexpr int BW_P = 64;             // points per tile
constexpr int BW_PITCH = BW_P + 1;
constexpr int BW_SMALL = 10;         // per-neuron vectors a thread accumulates: dW1[.][0..2], db1, db2, db3, db4, dW5[0..2][.]

struct UVBwdArgs {
    const float *W1, *b1, *b2, *emb, *b3, *b4, *W5, *b5, *off, *scale;
    const float4* pk4;               // A operands of W2, W3, W4, W2^T, W3^T, W4^T (k_uv_pack_bwd)
    const float* xyz;
    const float* g;
    int N, n_tiles;
    float* partW;                    // [grid][3][128][128]   dW2, dW3, dW4
    float* partS;                    // [grid][2][BW_SMALL][128]
    float* partB5;                   // [grid][4]
};

// W2, W3, W4, W2^T, W3^T, W4^T in A-operand order, four k-steps per 16-byte load:
//   pk4[m][band][g][lane] = (M_m[32 band + (lane & 31)][2 (4 g + j) + (lane >> 5)], j = 0..3),  M = W for m < 3, W^T for m >= 3
__global__ void __launch_bounds__(256)
k_uv_pack_bwd(const float* __restrict__ W2, const float* __restrict__ W3, const float* __restrict__ W4, float* __restrict__ pk, int n_mat) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_mat * 4 * 64 * 64) return;
    const int j = idx & 3, lane = (idx >> 2) & 63, g = (idx >> 8) & 15, band = (idx >> 12) & 3, m = idx >> 14;
    const int layer = m % 3;
    const float* W = layer == 0 ? W2 : (layer == 1 ? W3 : W4);
    const int r = band * 32 + (lane & 31), k = 2 * (4 * g + j) + (lane >> 5);
    pk[idx] = m < 3 ? W[r * UV_H + k] : W[k * UV_H + r];
}

// The mixed backward's W2^T, W3^T, W4^T: the same 64 KB per matrix and 16 KB per wave slice, as split bf16 in the A-operand order of
// mfma_f32_32x32x16_bf16 -- chunk (2 s + {0: hi, 1: lo}) of a slice holds lane l's M^T[32 band + (l & 31)][16 s + 8 (l >> 5) + j], j = 0..7
__global__ void __launch_bounds__(256)
k_uv_pack_bwd_b16(const float* __restrict__ W2, const float* __restrict__ W3, const float* __restrict__ W4, uint4* __restrict__ pkT) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= 3 * 4 * 8 * 64) return;
    const int lane = idx & 63, s = (idx >> 6) & 7, band = (idx >> 9) & 3, layer = idx >> 11;
    const float* W = layer == 0 ? W2 : (layer == 1 ? W3 : W4);
    const int r = band * 32 + (lane & 31), k0 = 16 * s + 8 * (lane >> 5);
    bf16x8 h, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) { __bf16 x, y; split_bf16(W[(k0 + j) * UV_H + r], x, y); h[j] = x; l[j] = y; }
    uint4* o = pkT + ((size_t)((layer * 4 + band) * 16 + 2 * s)) * 64 + lane;
    o[0] = __builtin_bit_cast(uint4, h);
    o[64] = __builtin_bit_cast(uint4, l);
}

typedef float BwPlane[BW_PITCH];

// One wave's 32 x 128 slice of a packed matrix: buffer loads -- the descriptor and the slice's byte offset are scalars, the only
// per-lane part is lane * 16 (as flat loads the compiler kept 6 x 16 64-bit per-lane addresses alive across the tile loop and spilled)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void bw_load_a(float (&A)[64], __amdgpu_buffer_rsrc_t rsrc, int slice_bytes, int lane) {
#pragma unroll
    for (int g = 0; g < 16; ++g) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, slice_bytes + g * 1024, 0);
        A[4 * g] = __uint_as_float(v.x); A[4 * g + 1] = __uint_as_float(v.y); A[4 * g + 2] = __uint_as_float(v.z); A[4 * g + 3] = __uint_as_float(v.w);
    }
}

// c[128 x 64 points] rows of this wave = A (the wave's 32 x 128 slice, in registers) x sIn[128][64]
__device__ __forceinline__ void bw_gemm(const float (&A)[64], const BwPlane* sIn, int bn, int bk, f32x16& c0, f32x16& c1) {
#pragma unroll
    for (int s = 0; s < 64; ++s) {
        const float* row = &sIn[2 * s + bk][bn];
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(A[s], row[0], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(A[s], row[32], c1, 0, 0, 0);
        if ((s & 7) == 7) __builtin_amdgcn_sched_barrier(0);       // (keeps the scheduler from hoisting all 128 LDS reads: 512 VGPRs are spoken for)
    }
}

// dW[32 rows of this wave x 128] += sD[rows][64 points] x sH[128][64 points]^T
__device__ __forceinline__ void bw_outer(const BwPlane* sD, const BwPlane* sH, int wave, int bn, int bk, f32x16 (&dW)[4]) {
#pragma unroll
    for (int s = 0; s < BW_P / 2; ++s) {
        const float av = sD[32 * wave + bn][2 * s + bk];
#pragma unroll
        for (int t = 0; t < 4; ++t)
            dW[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, sH[32 * t + bn][2 * s + bk], dW[t], 0, 0, 0);
        if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
}

// ---- the mixed backward's forms of the two (round 6): the operands split into two bf16 each on the fly, three
// v_mfma_f32_32x32x16_bf16 per f32 product (Ah Bh + Ah Bl + Al Bh, f32 accumulation: ~2^-17 relative, as in k_uv_taylor_bf16x3).  The
// activation planes stay f32 and neuron-major in LDS (the forward recomputation needs them exact: its ReLU masks are the forward
// launch's); a B operand -- eight consecutive k of one column -- is eight 4-byte reads, conflict-free at the 65-float pitch either way.
// One wave per SIMD: nothing else hides an LDS round trip or fills the matrix pipe's shadow, and per MFMA (32 cycles of the pipe)
// there are ~8 VALU instructions of conversion to place.  Left to the compiler, instruction selection sinks every read next to its
// conversion (one exposed LDS round trip per pair of elements; scheduling fences only bind the machine scheduler that runs later):
// so the reads are inline asm with hand-counted s_waitcnt, and both loops are software-pipelined three deep by hand -- a stage issues
// the reads of step s + 2, the MFMAs of step s and, between them, the conversions of step s + 1 (read one stage ago).
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned bw_lds_addr(const void* p) { return (unsigned)(unsigned long long)p; }      // generic -> LDS byte offset
template <int O0, int O1> __device__ __forceinline__ f32x2v bw_lds_rd2(unsigned addr) {                          // offsets in dwords, < 256
    f32x2v v;
    asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(addr), "n"(O0), "n"(O1));
    return v;
}
// base + a compile-time constant, formed where it is used (left to the compiler, the 40 stage addresses of a tile are hoisted out of the
// tile loop as invariants, spilled, and re-read from scratch in front of every group of reads)
template <int OFF> __device__ __forceinline__ unsigned bw_addr_plus(unsigned base) {
    unsigned a;
    asm volatile("v_add_u32_e32 %0, %2, %1" : "=v"(a) : "v"(base), "n"(OFF));
    return a;
}
#define BW_FENCE() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ bf16x8 bw_a_frag(const float (&A)[64], int q) {
    const f32x4v v = {A[4 * q], A[4 * q + 1], A[4 * q + 2], A[4 * q + 3]};
    return __builtin_bit_cast(bf16x8, v);
}
// two consecutive k of one operand -> elements 2 i, 2 i + 1 of its high and low halves (round-to-nearest both: |x - hi - lo| <= 2^-16 |x|).
// One wave per SIMD issues a VALU instruction every 8 clocks (v_cvt_pk_bf16_f32: 16; measured), so these five instructions per pair,
// not the MFMAs, bound the loops below.  (A truncated high half -- v_perm instead of the first convert -- measured no faster and
// doubles the error.)
__device__ __forceinline__ void bw_split_pair(const f32x2v x, int i, bf16x8& h, bf16x8& l) {
    __bf16 h0, l0, h1, l1;
    split_bf16(x[0], h0, l0); split_bf16(x[1], h1, l1);
    h[2 * i] = h0; h[2 * i + 1] = h1; l[2 * i] = l0; l[2 * i + 1] = l1;
}
#define BW_MF16(A_, B_, C_) C_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_, B_, C_, 0, 0, 0)
// c[wave's 32 rows x 64 points] = M^T slice (split bf16, k_uv_pack_bwd_b16 order, in registers) x sIn[128][64]
// reads of step S: x[tile][i] = {sIn[16 S + 8 bk + 2 i][bn + 32 tile], the row below}: one ds_read2 per pair of k
template <int S> __device__ __forceinline__ void bw_gemm_ld(unsigned base, f32x2v (&u)[2][4]) {
    const unsigned b0 = bw_addr_plus<16 * S * BW_PITCH * 4>(base), b1 = bw_addr_plus<(16 * S + 4) * BW_PITCH * 4>(base);
    u[0][0] = bw_lds_rd2<0, BW_PITCH>(b0); u[0][1] = bw_lds_rd2<2 * BW_PITCH, 3 * BW_PITCH>(b0);
    u[1][0] = bw_lds_rd2<32, BW_PITCH + 32>(b0); u[1][1] = bw_lds_rd2<2 * BW_PITCH + 32, 3 * BW_PITCH + 32>(b0);
    u[0][2] = bw_lds_rd2<0, BW_PITCH>(b1); u[0][3] = bw_lds_rd2<2 * BW_PITCH, 3 * BW_PITCH>(b1);
    u[1][2] = bw_lds_rd2<32, BW_PITCH + 32>(b1); u[1][3] = bw_lds_rd2<2 * BW_PITCH + 32, 3 * BW_PITCH + 32>(b1);
}
#define BW_WAIT8(N_, U_) asm volatile("s_waitcnt lgkmcnt(" #N_ ")" : "+v"(U_[0][0]), "+v"(U_[0][1]), "+v"(U_[0][2]), "+v"(U_[0][3]), \
                                      "+v"(U_[1][0]), "+v"(U_[1][1]), "+v"(U_[1][2]), "+v"(U_[1][3]))
template <int S> __device__ __forceinline__ void bw_gemm_stage(const float (&A)[64], unsigned base, f32x2v (&x)[3][2][4], bf16x8 (&hb)[2][2],
                                                               bf16x8 (&lb)[2][2], f32x16& c0, f32x16& c1) {
    constexpr int c = S & 1, n = (S + 1) % 3;
    constexpr bool more = S + 1 < 8, load = S + 2 < 8;
    const bf16x8 ah = bw_a_frag(A, 2 * S), al = bw_a_frag(A, 2 * S + 1);
    auto cvt = [&](int i) { bw_split_pair(x[n][0][i], i, hb[c ^ 1][0], lb[c ^ 1][0]); bw_split_pair(x[n][1][i], i, hb[c ^ 1][1], lb[c ^ 1][1]); };
    BW_FENCE();
    if constexpr (load) bw_gemm_ld<S + 2 < 8 ? S + 2 : 0>(base, x[(S + 2) % 3]);
    BW_FENCE();
    BW_MF16(ah, hb[c][0], c0);
    BW_FENCE();
    if constexpr (more) {                   // the reads of step S + 1 have had a whole stage; only this stage's eight may be outstanding
        if constexpr (load) BW_WAIT8(8, x[n]); else BW_WAIT8(0, x[n]);
    }
    BW_FENCE();
    BW_MF16(ah, hb[c][1], c1);
    BW_FENCE();
    if constexpr (more) cvt(0);
    BW_FENCE();
    BW_MF16(ah, lb[c][0], c0);
    BW_FENCE();
    if constexpr (more) cvt(1);
    BW_FENCE();
    BW_MF16(ah, lb[c][1], c1);
    BW_FENCE();
    if constexpr (more) cvt(2);
    BW_FENCE();
    BW_MF16(al, hb[c][0], c0);
    BW_FENCE();
    if constexpr (more) cvt(3);
    BW_FENCE();
    BW_MF16(al, hb[c][1], c1);
    if constexpr (more) bw_gemm_stage<more ? S + 1 : 0>(A, base, x, hb, lb, c0, c1);
}
__device__ __forceinline__ void bw_gemm_b16(const float (&A)[64], const BwPlane* sIn, int bn, int bk, f32x16& c0, f32x16& c1) {
    static_assert(3 * BW_PITCH + 32 < 256, "ds_read2_b32 offsets are 8 bits");
    f32x2v x[3][2][4];                      // [step % 3][tile][pair of k]
    bf16x8 hb[2][2], lb[2][2];              // [step & 1][tile]
    const unsigned base = bw_lds_addr(&sIn[8 * bk][bn]);
    bw_gemm_ld<0>(base, x[0]); bw_gemm_ld<1>(base, x[1]);
    BW_WAIT8(8, x[0]);
#pragma unroll
    for (int i = 0; i < 4; ++i) { bw_split_pair(x[0][0][i], i, hb[0][0], lb[0][0]); bw_split_pair(x[0][1][i], i, hb[0][1], lb[0][1]); }
    bw_gemm_stage<0>(A, base, x, hb, lb, c0, c1);
    BW_FENCE();
}
// dW[32 rows of this wave x 128] += sD[rows][64 points] x sH[128][64 points]^T, k = the point; a stage = (k-step sk, column block t)
template <int OFF> __device__ __forceinline__ void bw_outer_ld(unsigned base, f32x2v (&u)[4]) {       // eight consecutive points as four pairs
    const unsigned a = bw_addr_plus<OFF>(base);
    u[0] = bw_lds_rd2<0, 1>(a); u[1] = bw_lds_rd2<2, 3>(a); u[2] = bw_lds_rd2<4, 5>(a); u[3] = bw_lds_rd2<6, 7>(a);
}
constexpr int BW_NS = (BW_P / 16) * 4;      // 16 stages, q = 4 sk + t
constexpr int bw_b_off(int q) { return ((q & 3) * 32 * BW_PITCH + 16 * (q >> 2)) * 4; }
template <int Q> __device__ __forceinline__ void bw_outer_stage(unsigned base_a, unsigned base_b, f32x2v (&xb)[3][4], f32x2v (&xa)[2][4],
                                                                bf16x8 (&bh)[2], bf16x8 (&bl)[2], bf16x8 (&ah)[2], bf16x8 (&al)[2], f32x16 (&dW)[4]) {
    constexpr int sk = Q >> 2, t = Q & 3, c = Q & 1, n = (Q + 1) % 3, an = (sk + 1) & 1;
    constexpr bool more_a = sk + 1 < BW_P / 16, more = Q + 1 < BW_NS, load = Q + 2 < BW_NS;
    BW_FENCE();
    if constexpr (load) bw_outer_ld<bw_b_off(load ? Q + 2 : 0)>(base_b, xb[(Q + 2) % 3]);
    if constexpr (t == 0 && more_a) bw_outer_ld<16 * (sk + 1) * 4>(base_a, xa[an]);       // the next k-step's rows of d: converted at t = 1, 2
    BW_FENCE();
    BW_MF16(ah[sk & 1], bh[c], dW[t]);
    BW_FENCE();
    if constexpr (more) {                   // the reads of the previous stage are done; this stage's four (+ four of d at t = 0) may be outstanding
        if constexpr (!load) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xb[n][0]), "+v"(xb[n][1]), "+v"(xb[n][2]), "+v"(xb[n][3]));
        else if constexpr (t == 0 && more_a) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(xb[n][0]), "+v"(xb[n][1]), "+v"(xb[n][2]), "+v"(xb[n][3]));
        else asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(xb[n][0]), "+v"(xb[n][1]), "+v"(xb[n][2]), "+v"(xb[n][3]),
                          "+v"(xa[an][0]), "+v"(xa[an][1]), "+v"(xa[an][2]), "+v"(xa[an][3]));
        bw_split_pair(xb[n][0], 0, bh[c ^ 1], bl[c ^ 1]); bw_split_pair(xb[n][1], 1, bh[c ^ 1], bl[c ^ 1]);
    }
    BW_FENCE();
    BW_MF16(ah[sk & 1], bl[c], dW[t]);
    BW_FENCE();
    if constexpr (more) { bw_split_pair(xb[n][2], 2, bh[c ^ 1], bl[c ^ 1]); bw_split_pair(xb[n][3], 3, bh[c ^ 1], bl[c ^ 1]); }
    BW_FENCE();
    BW_MF16(al[sk & 1], bh[c], dW[t]);
    BW_FENCE();
    if constexpr (more_a && (t == 1 || t == 2)) {          // (read at t = 0, waited for at t = 1)
        bw_split_pair(xa[an][2 * (t - 1)], 2 * (t - 1), ah[an], al[an]); bw_split_pair(xa[an][2 * (t - 1) + 1], 2 * (t - 1) + 1, ah[an], al[an]);
    }
    if constexpr (more) bw_outer_stage<more ? Q + 1 : 0>(base_a, base_b, xb, xa, bh, bl, ah, al, dW);
}
__device__ __forceinline__ void bw_outer_b16(const BwPlane* sD, const BwPlane* sH, int wave, int bn, int bk, f32x16 (&dW)[4]) {
    f32x2v xb[3][4], xa[2][4];
    bf16x8 bh[2], bl[2], ah[2], al[2];
    const unsigned base_b = bw_lds_addr(&sH[bn][8 * bk]), base_a = bw_lds_addr(&sD[32 * wave + bn][8 * bk]);
    bw_outer_ld<bw_b_off(0)>(base_b, xb[0]); bw_outer_ld<0>(base_a, xa[0]); bw_outer_ld<bw_b_off(1)>(base_b, xb[1]);
    asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(xb[0][0]), "+v"(xb[0][1]), "+v"(xb[0][2]), "+v"(xb[0][3]), "+v"(xa[0][0]), "+v"(xa[0][1]), "+v"(xa[0][2]), "+v"(xa[0][3]));
#pragma unroll
    for (int i = 0; i < 4; ++i) { bw_split_pair(xb[0][i], i, bh[0], bl[0]); bw_split_pair(xa[0][i], i, ah[0], al[0]); }
    bw_outer_stage<0>(base_a, base_b, xb, xa, bh, bl, ah, al, dW);
    BW_FENCE();
}
#undef BW_WAIT8
#undef BW_MF16
#undef BW_FENCE

// B16 = the mixed form: forward recomputation on the f32-input MFMA (bit-identical masks), the six backward GEMMs on split bf16
template <bool B16>
__global__ void __launch_bounds__(256, 1)
k_uv_backward(UVBwdArgs a) {
    __shared__ float sH1[UV_H][BW_PITCH], sA[UV_H][BW_PITCH], sH2[UV_H][BW_PITCH], sH3[UV_H][BW_PITCH];
    __shared__ float sX[BW_P][4], sG[BW_P][4];            // the tile's (normalised) inputs and upstream gradients
    __shared__ float sW1[UV_H][4];                        // W1 | b1
    __shared__ float sW5[3][UV_H];
    __shared__ float sBias[3][UV_H];                      // b2, b3, b4
    __shared__ float sEmb[UV_H];
    __shared__ float sO[4][3][BW_P];
    __shared__ float sDo[3][BW_P];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bn = lane & 31, bk = lane >> 5;
    const int ni = tid & 127, hf = tid >> 7;              // the neuron / half of the tile's points this thread owns in the VALU phases
    for (int e = tid; e < UV_H; e += 256) {
        sW1[e][0] = a.W1[3 * e]; sW1[e][1] = a.W1[3 * e + 1]; sW1[e][2] = a.W1[3 * e + 2]; sW1[e][3] = a.b1 ? a.b1[e] : 0.0f;
        sBias[0][e] = a.b2 ? a.b2[e] : 0.0f;
        sEmb[e] = a.emb[e];
        sBias[1][e] = a.b3 ? a.b3[e] : 0.0f;
        sBias[2][e] = a.b4 ? a.b4[e] : 0.0f;
        sW5[0][e] = a.W5[e]; sW5[1][e] = a.W5[UV_H + e]; sW5[2][e] = a.W5[2 * UV_H + e];
    }
    float inv[3], off[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { inv[c] = a.scale ? 1.0f / a.scale[c] : 1.0f; off[c] = a.off ? a.off[c] : 0.0f; }
    const float b5[3] = {a.b5 ? a.b5[0] : 0.0f, a.b5 ? a.b5[1] : 0.0f, a.b5 ? a.b5[2] : 0.0f};

    f32x16 dW2[4], dW3[4], dW4[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { dW2[t] = f32x16{0.f}; dW3[t] = f32x16{0.f}; dW4[t] = f32x16{0.f}; }
    float acc_s[BW_SMALL];
#pragma unroll
    for (int q = 0; q < BW_SMALL; ++q) acc_s[q] = 0.0f;
    float acc_b5[3] = {0.0f, 0.0f, 0.0f};

    const __amdgpu_buffer_rsrc_t pk = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(a.pk4), 0, 6 * UV_H * UV_H * 4, 0x00020000);
    constexpr int LSTR = 4 * 16 * 64 * 16;                                                        // bytes per packed matrix
    const int pF = __builtin_amdgcn_readfirstlane(wave) * 16 * 64 * 16;                          // + m * LSTR: W2, W3, W4
    const int pT = pF + 3 * LSTR;                                                                 //             W2^T, W3^T, W4^T
    float RA[64], RB[64];
    bw_load_a(RA, pk, pF, lane);                                    // W2
    auto relu_out = [&](BwPlane* sOut, const float* bias, const float* emb_or_null, const f32x16& c0, const f32x16& c1) {
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int i = wave * 32 + (v & 3) + 8 * (v >> 2) + 4 * bk;             // C/D layout: col = lane & 31
            const float b = bias[i], em = emb_or_null ? emb_or_null[i] : 0.0f;      // (the forward kernels' rounding order: same masks)
            sOut[i][bn] = fmaxf(uv_bias_emb(c0[v], b, em), 0.0f);
            sOut[i][32 + bn] = fmaxf(uv_bias_emb(c1[v], b, em), 0.0f);
        }
    };
    auto mask_into = [&](BwPlane* sH, const f32x16& c0, const f32x16& c1) {       // d_in = c . [h > 0], in place of h
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int i = wave * 32 + (v & 3) + 8 * (v >> 2) + 4 * bk;
            sH[i][bn] = sH[i][bn] > 0.0f ? c0[v] : 0.0f;
            sH[i][32 + bn] = sH[i][32 + bn] > 0.0f ? c1[v] : 0.0f;
        }
    };
    auto row_sum = [&](const BwPlane* sD) {
        float r = 0.0f;
#pragma unroll 8
        for (int p = 0; p < BW_P / 2; ++p) r += sD[ni][32 * hf + p];
        return r;
    };
    __syncthreads();

    for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
        const int p0 = tile * BW_P;
        // ---- inputs of the tile (points past N: g = 0, so they add nothing anywhere)
        if (tid < BW_P) {
            const int n = min(p0 + tid, a.N - 1);
            const bool ok = p0 + tid < a.N;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                sX[tid][c] = uv_norm_in(a.xyz[3 * n + c], off[c], inv[c]);
                sG[tid][c] = ok ? a.g[3 * n + c] : 0.0f;
            }
        }
        __syncthreads();
        // ---- layer 1 (3 -> 128), VALU
        for (int e = tid; e < UV_H * BW_P; e += 256) {
            const int i = e >> 6, p = e & 63;
            sH1[i][p] = fmaxf(uv_layer1_pre(sW1[i][0], sW1[i][1], sW1[i][2], sW1[i][3], sX[p][0], sX[p][1], sX[p][2]), 0.0f);
        }
        __syncthreads();
        // ---- forward, three GEMMs; the next GEMM's A slice is loaded while this one runs
        {
            bw_load_a(RB, pk, pF + LSTR, lane);                     // W3
            f32x16 c0 = {0.f}, c1 = {0.f};
            bw_gemm(RA, sH1, bn, bk, c0, c1);
            relu_out(sA, sBias[0], sEmb, c0, c1);
        }
        __syncthreads();
        {
            bw_load_a(RA, pk, pF + 2 * LSTR, lane);                 // W4
            f32x16 c0 = {0.f}, c1 = {0.f};
            bw_gemm(RB, sA, bn, bk, c0, c1);
            relu_out(sH2, sBias[1], nullptr, c0, c1);
        }
        __syncthreads();
        {
            bw_load_a(RB, pk, pT + 2 * LSTR, lane);                 // W4^T
            f32x16 c0 = {0.f}, c1 = {0.f};
            bw_gemm(RA, sH2, bn, bk, c0, c1);
            relu_out(sH3, sBias[2], nullptr, c0, c1);
        }
        __syncthreads();
        // ---- output layer + F.normalize and its backward
        {
            float o0 = 0.0f, o1 = 0.0f, o2 = 0.0f;
#pragma unroll 8
            for (int ii = 0; ii < 32; ++ii) {
                const int i = 32 * wave + ii;
                const float h = sH3[i][lane];
                o0 += sW5[0][i] * h; o1 += sW5[1][i] * h; o2 += sW5[2][i] * h;
            }
            sO[wave][0][lane] = o0; sO[wave][1][lane] = o1; sO[wave][2][lane] = o2;
        }
        __syncthreads();
        if (tid < BW_P) {
            float o[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) o[c] = b5[c] + ((sO[0][c][tid] + sO[1][c][tid]) + (sO[2][c][tid] + sO[3][c][tid]));
            const float rn = 1.0f / fmaxf(sqrtf(o[0] * o[0] + o[1] * o[1] + o[2] * o[2]), 1e-12f);
            const float u0 = o[0] * rn, u1 = o[1] * rn, u2 = o[2] * rn;
            const float g0 = sG[tid][0], g1 = sG[tid][1], g2 = sG[tid][2];
            const float ug = u0 * g0 + u1 * g1 + u2 * g2;
            const float d0 = (g0 - u0 * ug) * rn, d1 = (g1 - u1 * ug) * rn, d2 = (g2 - u2 * ug) * rn;
            sDo[0][tid] = d0; sDo[1][tid] = d1; sDo[2][tid] = d2;
            acc_b5[0] += d0; acc_b5[1] += d1; acc_b5[2] += d2;
        }
        __syncthreads();
        // ---- dW5, d4 (in place of h3), db4
        {
            const float w0 = sW5[0][ni], w1 = sW5[1][ni], w2 = sW5[2][ni];
#pragma unroll 8
            for (int pp = 0; pp < BW_P / 2; ++pp) {
                const int p = 32 * hf + pp;
                const float h = sH3[ni][p], d0 = sDo[0][p], d1 = sDo[1][p], d2 = sDo[2][p];
                acc_s[7] += d0 * h; acc_s[8] += d1 * h; acc_s[9] += d2 * h;
                const float d4 = h > 0.0f ? w0 * d0 + w1 * d1 + w2 * d2 : 0.0f;
                acc_s[6] += d4;
                sH3[ni][p] = d4;
            }
        }
        __syncthreads();
        // ---- layer 4: d3 = W4^T d4 . [h2 > 0], dW4 += d4 h2^T
        if constexpr (B16) {
            // ONE A buffer through the backward chain: the outer product needs no A operand, so the next layer's slice is loaded into
            // the registers the GEMM has just finished with and lands while the outer product runs (RA is dead until the last load:
            // 64 registers for the hand-pipelined operand conversions)
            f32x16 c0 = {0.f}, c1 = {0.f};
            bw_gemm_b16(RB, sH3, bn, bk, c0, c1);
            bw_load_a(RB, pk, pT + LSTR, lane);                     // W3^T
            bw_outer_b16(sH3, sH2, wave, bn, bk, dW4);
            __syncthreads();
            mask_into(sH2, c0, c1);
        } else {
            bw_load_a(RA, pk, pT + LSTR, lane);                     // W3^T
            f32x16 c0 = {0.f}, c1 = {0.f};
            bw_gemm(RB, sH3, bn, bk, c0, c1);
            bw_outer(sH3, sH2, wave, bn, bk, dW4);
            __syncthreads();
            mask_into(sH2, c0, c1);
        }
        __syncthreads();
        // ---- layer 3
        if constexpr (B16) {
            f32x16 c0 = {0.f}, c1 = {0.f};
            bw_gemm_b16(RB, sH2, bn, bk, c0, c1);
            bw_load_a(RB, pk, pT, lane);                            // W2^T
            bw_outer_b16(sH2, sA, wave, bn, bk, dW3);
            acc_s[5] += row_sum(sH2);
            __syncthreads();
            mask_into(sA, c0, c1);
        } else {
            bw_load_a(RB, pk, pT, lane);                            // W2^T
            f32x16 c0 = {0.f}, c1 = {0.f};
            bw_gemm(RA, sH2, bn, bk, c0, c1);
            bw_outer(sH2, sA, wave, bn, bk, dW3);
            acc_s[5] += row_sum(sH2);
            __syncthreads();
            mask_into(sA, c0, c1);
        }
        __syncthreads();
        // ---- layer 2
        {
            f32x16 c0 = {0.f}, c1 = {0.f};
            if constexpr (B16) {
                bw_gemm_b16(RB, sA, bn, bk, c0, c1);
                bw_load_a(RA, pk, pF, lane);                        // W2, for the next tile
                bw_outer_b16(sA, sH1, wave, bn, bk, dW2);
            } else {
                bw_load_a(RA, pk, pF, lane);                        // W2, for the next tile
                bw_gemm(RB, sA, bn, bk, c0, c1);
                bw_outer(sA, sH1, wave, bn, bk, dW2);
            }
            acc_s[4] += row_sum(sA);
            __syncthreads();
            mask_into(sH1, c0, c1);
        }
        __syncthreads();
        // ---- layer 1: dW1 += d1 x^T, db1 += d1
#pragma unroll 8
        for (int pp = 0; pp < BW_P / 2; ++pp) {
            const int p = 32 * hf + pp;
            const float d = sH1[ni][p];
            acc_s[0] += d * sX[p][0]; acc_s[1] += d * sX[p][1]; acc_s[2] += d * sX[p][2];
            acc_s[3] += d;
        }
        __syncthreads();
    }
    // ---- this workgroup's partial sums
    float* __restrict__ pw = a.partW + (size_t)blockIdx.x * 3 * UV_H * UV_H;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int r = wave * 32 + (v & 3) + 8 * (v >> 2) + 4 * bk, col = 32 * t + bn;
            pw[r * UV_H + col] = dW2[t][v];
            pw[UV_H * UV_H + r * UV_H + col] = dW3[t][v];
            pw[2 * UV_H * UV_H + r * UV_H + col] = dW4[t][v];
        }
    float* __restrict__ ps = a.partS + ((size_t)blockIdx.x * 2 + hf) * BW_SMALL * UV_H;
#pragma unroll
    for (int q = 0; q < BW_SMALL; ++q) ps[q * UV_H + ni] = acc_s[q];
    if (wave == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = acc_b5[c];
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
            if (lane == 0) a.partB5[blockIdx.x * 4 + c] = v;
        }
    }
}

struct UVGradOut {
    float *dW1, *db1, *dW2, *db2, *dW3, *db3, *dW4, *db4, *dW5, *db5;
};

// partial sums of the <= 256 workgroups -> the gradients.  One thread per output element adds its partials in workgroup order
// (deterministic); the loads of 16 partials are issued together -- one dependent load per partial was 256 serial L2 / HBM round
// trips per thread (118 us for 50 MB).
__global__ void __launch_bounds__(256)
k_uv_backward_reduce(const float* __restrict__ partW, const float* __restrict__ partS, const float* __restrict__ partB5, int G,
                     UVGradOut o) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    constexpr int NW = 3 * UV_H * UV_H, NS = BW_SMALL * UV_H;
    auto sum_strided = [](const float* __restrict__ p, int n, size_t stride) {
        float v = 0.0f;
        int g = 0;
        for (; g + 16 <= n; g += 16) {
            float t[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) t[k] = p[(size_t)(g + k) * stride];
#pragma unroll
            for (int k = 0; k < 16; ++k) v += t[k];
        }
        for (; g < n; ++g) v += p[(size_t)g * stride];
        return v;
    };
    if (e < NW) {
        float* dst = e < UV_H * UV_H ? o.dW2 : (e < 2 * UV_H * UV_H ? o.dW3 : o.dW4);
        if (!dst) return;
        dst[e & (UV_H * UV_H - 1)] = sum_strided(partW + e, G, NW);
    } else if (e < NW + NS) {
        const int k = e - NW, q = k >> 7, i = k & 127;
        const float v = sum_strided(partS + k, 2 * G, NS);
        if (q < 3) { if (o.dW1) o.dW1[3 * i + q] = v; }
        else if (q == 3) { if (o.db1) o.db1[i] = v; }
        else if (q == 4) { if (o.db2) o.db2[i] = v; }
        else if (q == 5) { if (o.db3) o.db3[i] = v; }
        else if (q == 6) { if (o.db4) o.db4[i] = v; }
        else { if (o.dW5) o.dW5[(q - 7) * UV_H + i] = v; }
    } else if (e < NW + NS + 3) {
        const int c = e - NW - NS;
        if (o.db5) o.db5[c] = sum_strided(partB5 + c, G, 4);
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

// ---- backward: temp = [packed W2..W4][packed W2^T..W4^T][partials of `uv_backward_blocks(N)` workgroups]
static int uv_backward_blocks(int N) { const int t = (N + BW_P - 1) / BW_P; return t < 256 ? (t < 1 ? 1 : t) : 256; }
static size_t uv_backward_part_floats() { return (size_t)3 * UV_H * UV_H + 2 * BW_SMALL * UV_H + 4; }
size_t uv_backward_temp_bytes(int N) {
    return 2 * uv_taylor_temp_bytes() + (size_t)uv_backward_blocks(N) * uv_backward_part_floats() * sizeof(float);
}

int launch_uv_backward(const TexGSUVNet* net, const float* xyz, const float* g, int N, const TexGSUVNetGrad* out, void* temp, int mixed,
                       hipStream_t s) {
    const int G = uv_backward_blocks(N);
    float* packed = reinterpret_cast<float*>(temp);
    float* partW = packed + 6 * UV_H * UV_H;
    float* partS = partW + (size_t)G * 3 * UV_H * UV_H;
    float* partB5 = partS + (size_t)G * 2 * BW_SMALL * UV_H;
    UVGradOut o{out->dW1, out->db1, out->dW2, out->db2, out->dW3, out->db3, out->dW4, out->db4, out->dW5, out->db5};
    if (N <= 0) {                                          // no points: every gradient is zero
        (void)hipMemsetAsync(partW, 0, (size_t)G * uv_backward_part_floats() * sizeof(float), s);
    } else {
        if (mixed) {        // W2..W4 in f32 A-operand order (forward recomputation), W2^T..W4^T as split bf16
            hipLaunchKernelGGL(k_uv_pack_bwd, dim3(3 * 4 * 64 * 64 / 256), dim3(256), 0, s, net->W2, net->W3, net->W4, packed, 3);
            hipLaunchKernelGGL(k_uv_pack_bwd_b16, dim3(3 * 4 * 8 * 64 / 256), dim3(256), 0, s, net->W2, net->W3, net->W4,
                               reinterpret_cast<uint4*>(packed + 3 * UV_H * UV_H));
        } else {
            hipLaunchKernelGGL(k_uv_pack_bwd, dim3(6 * 4 * 64 * 64 / 256), dim3(256), 0, s, net->W2, net->W3, net->W4, packed, 6);
        }
        UVBwdArgs a;
        a.W1 = net->W1; a.b1 = net->b1; a.b2 = net->b2; a.emb = net->emb; a.b3 = net->b3; a.b4 = net->b4; a.W5 = net->W5; a.b5 = net->b5;
        a.off = net->xyz_offset; a.scale = net->xyz_scale; a.pk4 = reinterpret_cast<const float4*>(packed); a.xyz = xyz; a.g = g;
        a.N = N; a.n_tiles = (N + BW_P - 1) / BW_P; a.partW = partW; a.partS = partS; a.partB5 = partB5;
        if (mixed) hipLaunchKernelGGL(k_uv_backward<true>, dim3(G), dim3(256), 0, s, a);
        else hipLaunchKernelGGL(k_uv_backward<false>, dim3(G), dim3(256), 0, s, a);
    }
    const int n_out = 3 * UV_H * UV_H + BW_SMALL * UV_H + 3;
    hipLaunchKernelGGL(k_uv_backward_reduce, dim3((n_out + 255) / 256), dim3(256), 0, s, partW, partS, partB5, G, o);
    return (int)hipGetLastError();
}

// mixed variant: `packed` holds BOTH layouts back to back (2 x uv_taylor_temp_bytes(): f32 pack, then split-bf16 pack)
int launch_uv_pack_mixed(const TexGSUVNet* net, void* packed, hipStream_t s) {
    if (int r = launch_uv_pack(net, packed, s)) return r;
    return launch_uv_pack_bf16x3(net, reinterpret_cast<char*>(packed) + uv_taylor_temp_bytes(), s);
}

int launch_uv_taylor_packed_mixed(const TexGSUVNet* net, const void* packed, const float* xyz, int N, float* uvs, float* grad_uvs,
                                  hipStream_t s) {
    if (N <= 0) return 0;
    UVArgsM a;
    a.W1 = net->W1; a.b1 = net->b1; a.b2 = net->b2; a.emb = net->emb; a.b3 = net->b3; a.b4 = net->b4; a.W5 = net->W5; a.b5 = net->b5;
    a.off = net->xyz_offset; a.scale = net->xyz_scale;
    a.packed_f32 = reinterpret_cast<const float*>(packed);
    a.packed_b16 = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(packed) + uv_taylor_temp_bytes());
    hipLaunchKernelGGL(k_uv_taylor_mixed, dim3((N + UV_P - 1) / UV_P), dim3(256), 0, s, a, xyz, N, uvs, grad_uvs);
    return (int)hipGetLastError();
}
