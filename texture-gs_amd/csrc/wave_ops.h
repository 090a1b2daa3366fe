// Wave64 cross-lane primitives for gfx950 built on DPP and the CDNA4 permlane swaps (no LDS crossbar traffic).
#pragma once
#include <hip/hip_runtime.h>

#define DPP_QUAD_XOR1   0xB1     // quad_perm [1,0,3,2]
#define DPP_QUAD_XOR2   0x4E     // quad_perm [2,3,0,1]
#define DPP_ROW_HMIRROR 0x141
#define DPP_ROW_MIRROR  0x140

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}

// value of lane (l ^ 4): row_shl:4 feeds lanes 0-3/8-11 of each row, row_shr:4 feeds lanes 4-7/12-15
__device__ __forceinline__ float lane_xor4(float v) {
    const int x = __float_as_int(v);
    int t = __builtin_amdgcn_update_dpp(x, x, 0x104, 0xF, 0x5, false);
    t = __builtin_amdgcn_update_dpp(t, x, 0x114, 0xF, 0xA, false);
    return __int_as_float(t);
}
// value of lane (l ^ 8)
__device__ __forceinline__ float lane_xor8(float v) {
    const int x = __float_as_int(v);
    int t = __builtin_amdgcn_update_dpp(x, x, 0x108, 0xF, 0x3, false);
    t = __builtin_amdgcn_update_dpp(t, x, 0x118, 0xF, 0xC, false);
    return __int_as_float(t);
}
// v[l] + v[l ^ 16] and v[l] + v[l ^ 32] in every lane
__device__ __forceinline__ float sum_xor16(float v) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_int(v), __float_as_int(v), false, false);
    return __int_as_float(r[0]) + __int_as_float(r[1]);
}
__device__ __forceinline__ float sum_xor32(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_int(v), __float_as_int(v), false, false);
    return __int_as_float(r[0]) + __int_as_float(r[1]);
}

// Sum over the 64 lanes of a wave; result in every lane.
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_mov<DPP_QUAD_XOR1>(v);
    v += dpp_mov<DPP_QUAD_XOR2>(v);
    v += dpp_mov<DPP_ROW_HMIRROR>(v);
    v += dpp_mov<DPP_ROW_MIRROR>(v);
    v = sum_xor16(v);
    return sum_xor32(v);
}

__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = max(v, __shfl_xor(v, m, 64));
    return v;
}

// Transposing butterfly over NV (= 16 or 32) per-lane values: on return every lane l holds the sum over the 64 lanes
// of v[l & (NV-1)].  Each halving step keeps one of a pair and sends the other to the partner lane, so the live
// value count halves per step: 3*NV/2 + ... VALU instead of NV full reductions.
template <int NV>
__device__ __forceinline__ float reduce_transposed(float (&v)[NV], int lane) {
    static_assert(NV == 16 || NV == 32, "NV");
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8, b4 = lane & 16;
    float a[NV / 2];
#pragma unroll
    for (int i = 0; i < NV / 2; ++i) {
        const float keep = b0 ? v[2 * i + 1] : v[2 * i], send = b0 ? v[2 * i] : v[2 * i + 1];
        a[i] = keep + dpp_mov<DPP_QUAD_XOR1>(send);
    }
    float b[NV / 4];
#pragma unroll
    for (int i = 0; i < NV / 4; ++i) {
        const float keep = b1 ? a[2 * i + 1] : a[2 * i], send = b1 ? a[2 * i] : a[2 * i + 1];
        b[i] = keep + dpp_mov<DPP_QUAD_XOR2>(send);
    }
    float c[NV / 8];
#pragma unroll
    for (int i = 0; i < NV / 8; ++i) {
        const float keep = b2 ? b[2 * i + 1] : b[2 * i], send = b2 ? b[2 * i] : b[2 * i + 1];
        c[i] = keep + lane_xor4(send);
    }
    float d[NV / 16];
#pragma unroll
    for (int i = 0; i < NV / 16; ++i) {
        const float keep = b3 ? c[2 * i + 1] : c[2 * i], send = b3 ? c[2 * i] : c[2 * i + 1];
        d[i] = keep + lane_xor8(send);
    }
    float r;
    if (NV == 32) {
        const float keep = b4 ? d[NV / 16 - 1] : d[0], send = b4 ? d[0] : d[NV / 16 - 1];
        const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_int(send), __float_as_int(send), false, false);
        // lane l needs send[l ^ 16]: rows 0,2 take it from sw[1] (= odd rows), rows 1,3 from sw[0] (= even rows)
        r = keep + (b4 ? __int_as_float(sw[0]) : __int_as_float(sw[1]));
    } else {
        r = sum_xor16(d[0]);
    }
    return sum_xor32(r);
}


// ---- bank-first variant ------------------------------------------------------------------------------------------
// Same reduction, steps reordered so that the two widest ones (16 and 8 pairs) are the lane^4 and lane^8 exchanges:
// their keep/send split coincides with DPP *banks* (groups of 4 lanes), so one pair costs two bank-masked
// v_add_f32_dpp (banks {0,2} <- a + a[l+4], banks {1,3} <- b + b[l-4]) instead of 2 selects + 2 DPP moves + 1 add.
// On return lane l (< 32; lanes 32..63 mirror them) holds the wave total of v[transposed_index(l)].
__device__ __forceinline__ int transposed_index(int lane) {
    return ((lane >> 2) & 1) | (((lane >> 3) & 1) << 1) | ((lane & 1) << 2) | (((lane >> 1) & 1) << 3) | (((lane >> 4) & 1) << 4);
}

__device__ __forceinline__ float pair_xor4(float a, float b) {
    float r;
    asm volatile("s_nop 1\n\t"
                 "v_add_f32_dpp %0, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
                 "v_add_f32_dpp %0, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xa"
                 : "=&v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float pair_xor8(float a, float b) {
    float r;
    asm volatile("s_nop 1\n\t"
                 "v_add_f32_dpp %0, %1, %1 row_shl:8 row_mask:0xf bank_mask:0x3\n\t"
                 "v_add_f32_dpp %0, %2, %2 row_shr:8 row_mask:0xf bank_mask:0xc"
                 : "=&v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ float reduce32_bankfirst(float (&v)[32], int lane) {
    const bool b0 = lane & 1, b1 = lane & 2, b4 = lane & 16;
    float a[16], b[8], c[4], d[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = pair_xor4(v[2 * i], v[2 * i + 1]);
#pragma unroll
    for (int i = 0; i < 8; ++i) b[i] = pair_xor8(a[2 * i], a[2 * i + 1]);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float keep = b0 ? b[2 * i + 1] : b[2 * i], send = b0 ? b[2 * i] : b[2 * i + 1];
        c[i] = keep + dpp_mov<DPP_QUAD_XOR1>(send);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float keep = b1 ? c[2 * i + 1] : c[2 * i], send = b1 ? c[2 * i] : c[2 * i + 1];
        d[i] = keep + dpp_mov<DPP_QUAD_XOR2>(send);
    }
    const float keep = b4 ? d[1] : d[0], send = b4 ? d[0] : d[1];
    const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_int(send), __float_as_int(send), false, false);
    const float r = keep + (b4 ? __int_as_float(sw[0]) : __int_as_float(sw[1]));
    return sum_xor32(r);
}

// 16-lane version: sums v[0..31] over each DPP row (16 lanes) separately.  On return lane l holds, for its row, the total of
// slot transposed_index(l & 15) in `lo` and of slot 16 + transposed_index(l & 15) in `hi` (the four steps of
// reduce32_bankfirst before its cross-row exchanges: lane^4, lane^8 bank-masked, lane^1, lane^2 quad_perm).
// LIVE: bit k set = slot k may be non-zero in some lane; an exchange whose slots are all dead is not issued (the result is the
// constant 0).  The 28 moments of the full backward: 0x0FFFFFFF; without the UV chain 13 slots are live and 15 of the 29 exchanges go.
template <uint32_t LIVE = 0xFFFFFFFFu>
__device__ __forceinline__ void reduce32_rows16_masked(float (&v)[32], int lane, float& lo, float& hi) {
    const bool b0 = lane & 1, b1 = lane & 2;
    float a[16], b[8], c[4];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = ((LIVE >> (2 * i)) & 3u) ? pair_xor4(v[2 * i], v[2 * i + 1]) : 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) b[i] = ((LIVE >> (4 * i)) & 15u) ? pair_xor8(a[2 * i], a[2 * i + 1]) : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if ((LIVE >> (8 * i)) & 255u) {
            const float keep = b0 ? b[2 * i + 1] : b[2 * i], send = b0 ? b[2 * i] : b[2 * i + 1];
            c[i] = keep + dpp_mov<DPP_QUAD_XOR1>(send);
        } else c[i] = 0.f;
    }
    lo = 0.f; hi = 0.f;
    if (LIVE & 0xFFFFu) {
        const float keep = b1 ? c[1] : c[0], send = b1 ? c[0] : c[1];
        lo = keep + dpp_mov<DPP_QUAD_XOR2>(send);
    }
    if (LIVE >> 16) {
        const float keep = b1 ? c[3] : c[2], send = b1 ? c[2] : c[3];
        hi = keep + dpp_mov<DPP_QUAD_XOR2>(send);
    }
}
// NLIVE (a multiple of 4): slots >= NLIVE are known to be zero in every lane.
template <int NLIVE = 32>
__device__ __forceinline__ void reduce32_rows16(float (&v)[32], int lane, float& lo, float& hi) {
    static_assert(NLIVE % 4 == 0 && NLIVE > 0 && NLIVE <= 32, "NLIVE");
    reduce32_rows16_masked<(NLIVE == 32) ? 0xFFFFFFFFu : ((1u << (NLIVE & 31)) - 1u)>(v, lane, lo, hi);
}
