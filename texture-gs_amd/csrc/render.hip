// K6 render forward and K7 render backward -- gfx950 (CDNA4), wave64.
//
// One 256-thread workgroup (4 wave64) per 16x16 tile; each wave owns an 8x8 pixel sub-block so that the
// wave-level early-out (`__ballot`) and the "does this Gaussian touch my pixels" test are spatially tight.
// The tile's depth-sorted instance list is consumed in batches of 256: the workgroup gathers the 96-byte
// hot part of each instance's 128-byte record (xy, conic, opacity, UV Taylor fold g/G/phi, view-dependent
// colour, depth, normal) into LDS as six float4 planes (conflict-free staging writes, broadcast reads), then
// every wave walks the batch independently -- no barrier inside a batch.
//
// No MFMA: there is no dense contraction on this path.  Bound: HBM / L2 gather + fp32 atomics (backward).
#include "common.h"

namespace {

#define DPP_QUAD_XOR1   0xB1     // quad_perm [1,0,3,2]
#define DPP_QUAD_XOR2   0x4E     // quad_perm [2,3,0,1]
#define DPP_ROW_HMIRROR 0x141
#define DPP_ROW_MIRROR  0x140

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}

// Sum over the 64 lanes of a wave; result is wave-uniform.
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_mov<DPP_QUAD_XOR1>(v);
    v += dpp_mov<DPP_QUAD_XOR2>(v);
    v += dpp_mov<DPP_ROW_HMIRROR>(v);
    v += dpp_mov<DPP_ROW_MIRROR>(v);
    const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (a + b) + (c + d);
}

__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = max(v, __shfl_xor(v, m, 64));
    return v;
}

// blockIdx -> tile.  Workgroup b is observed to run on XCD b % 8; give each XCD a contiguous band of tiles so
// neighbouring tiles (which share most of their Gaussians' records) hit the same 4 MiB L2.  Speed only.
__device__ __forceinline__ int tile_of_block(int b, int num_tiles) {
    const int chunk = (num_tiles + 7) >> 3;
    return (b & 7) * chunk + (b >> 3);
}

// Cubemap address of direction u (not necessarily unit): face (+x,-x,+y,-y,+z,-z; NVDIFFREC/util.py:94-101
// inverted), bilinear taps with clamp-to-edge inside the face, texel centres at (i+0.5)/R.
struct CubeTap {
    int   o00, o01, o10, o11;   // float offsets of the 4 taps' first channel
    float fx, fy;
    // for the backward: sc/tc numerators, 0.5*R/ma, axis bookkeeping
    float sc, tc, h, rma, sm, su, sv;
    int   axis;
};

__device__ __forceinline__ CubeTap cube_address(float u0, float u1, float u2, int R) {
    CubeTap t;
    const float a0 = fabsf(u0), a1 = fabsf(u1), a2 = fabsf(u2);
    float m, ua, ub;
    if (a0 >= a1 && a0 >= a2) { t.axis = 0; m = u0; t.sm = (u0 >= 0.f) ? 1.f : -1.f; ua = u2; t.su = -t.sm; ub = u1; t.sv = -1.f; }
    else if (a1 >= a2)        { t.axis = 1; m = u1; t.sm = (u1 >= 0.f) ? 1.f : -1.f; ua = u0; t.su = 1.f;   ub = u2; t.sv = t.sm; }
    else                      { t.axis = 2; m = u2; t.sm = (u2 >= 0.f) ? 1.f : -1.f; ua = u0; t.su = t.sm;  ub = u1; t.sv = -1.f; }
    const int face = 2 * t.axis + (t.sm > 0.f ? 0 : 1);
    const float ma = fmaxf(fabsf(m), TG_MA_MIN);
    t.rma = __builtin_amdgcn_rcpf(ma);
    t.sc = t.su * ua; t.tc = t.sv * ub;
    const float halfR = 0.5f * (float)R;
    t.h = halfR * t.rma;
    const float col = (t.sc * t.rma + 1.0f) * halfR - 0.5f;
    const float row = (t.tc * t.rma + 1.0f) * halfR - 0.5f;
    const float x0f = floorf(col), y0f = floorf(row);
    t.fx = col - x0f; t.fy = row - y0f;
    const int x0 = (int)x0f, y0 = (int)y0f;
    const int x0c = min(max(x0, 0), R - 1), x1c = min(max(x0 + 1, 0), R - 1);
    const int y0c = min(max(y0, 0), R - 1), y1c = min(max(y0 + 1, 0), R - 1);
    const int fb = face * R;
    t.o00 = ((fb + y0c) * R + x0c) * 3; t.o01 = ((fb + y0c) * R + x1c) * 3;
    t.o10 = ((fb + y1c) * R + x0c) * 3; t.o11 = ((fb + y1c) * R + x1c) * 3;
    return t;
}

struct PixArgs {
    int W, H, tiles_x, num_tiles, R;
    const uint2* ranges;
    const uint32_t* point_list;
    const float4* rec;
    const float* texture;
    const float* bg;
};

// ------------------------------------------------------------------------------------------------ K6
__global__ void __launch_bounds__(TG_BLOCK)
k_render_fwd(PixArgs a, float* __restrict__ out_color, float* __restrict__ out_depth, float* __restrict__ out_norm,
             float* __restrict__ out_alpha, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib) {
    __shared__ float4 s_rec[6][TG_BLOCK];
    __shared__ int s_alive[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = tile_of_block(blockIdx.x, a.num_tiles);
    if (tile >= a.num_tiles) return;
    const int tile_x = tile % a.tiles_x, tile_y = tile / a.tiles_x;
    const int px = tile_x * TEXGS_TILE + ((wave & 1) << 3) + (lane & 7);
    const int py = tile_y * TEXGS_TILE + ((wave >> 1) << 3) + (lane >> 3);
    const bool inside = (px < a.W) && (py < a.H);
    const float pxf = (float)px, pyf = (float)py;
    const uint2 range = a.ranges[tile];
    const int todo = (int)(range.y - range.x);
    const float* __restrict__ tex = a.texture;

    bool done = !inside;
    float T = 1.0f;
    float C0 = 0.f, C1 = 0.f, C2 = 0.f, Dp = 0.f, N0 = 0.f, N1 = 0.f, N2 = 0.f, Al = 0.f;
    uint32_t last = 0;

    for (int base = 0; base < todo; base += TG_BLOCK) {
        const unsigned long long alive = __ballot(!done);
        if (lane == 0) s_alive[wave] = (alive != 0ull);
        __syncthreads();
        if (!(s_alive[0] | s_alive[1] | s_alive[2] | s_alive[3])) break;
        const int cnt = min(TG_BLOCK, todo - base);
        if (tid < cnt) {
            const uint32_t id = a.point_list[range.x + base + tid];
            const float4* __restrict__ r = a.rec + (size_t)id * (TEXGS_REC_FLOATS / 4);
            const float4 v0 = r[0], v1 = r[1], v2 = r[2], v3 = r[3], v4 = r[4], v5 = r[5];
            s_rec[0][tid] = v0; s_rec[1][tid] = v1; s_rec[2][tid] = v2;
            s_rec[3][tid] = v3; s_rec[4][tid] = v4; s_rec[5][tid] = v5;
        }
        __syncthreads();
        if (alive == 0ull) continue;                    // this wave is finished; it only helps staging
        for (int j = 0; j < cnt; ++j) {
            const float4 r0 = s_rec[0][j];              // xy.x xy.y conic.a conic.b
            const float4 r1 = s_rec[1][j];              // conic.c opacity g.x g.y
            const float dx = r0.x - pxf, dy = r0.y - pyf;
            const float power = -0.5f * (r0.z * dx * dx + r1.x * dy * dy) - r0.w * dx * dy;
            const float alpha = fminf(TG_ALPHA_MAX, r1.y * __expf(power));
            bool ok = (!done) && (power <= 0.0f) && (alpha >= TG_ALPHA_MIN);
            const float Tn = T * (1.0f - alpha);
            if (ok && Tn < TG_T_EPS) { done = true; ok = false; }
            if (ok) {
                const float4 r2 = s_rec[2][j];          // G00 G01 G10 G11
                const float4 r3 = s_rec[3][j];          // G20 G21 phi0 phi1
                const float4 r4 = s_rec[4][j];          // phi2 vd0 vd1 vd2
                const float4 r5 = s_rec[5][j];          // depth n0 n1 n2
                const float dpx = -dx, dpy = -dy;
                const float den = 1.0f + r1.z * dpx + r1.w * dpy;
                const float inv = (den >= TG_DEN_MIN) ? __builtin_amdgcn_rcpf(den) : 0.0f;
                const float u0 = r3.z + (r2.x * dpx + r2.y * dpy) * inv;
                const float u1 = r3.w + (r2.z * dpx + r2.w * dpy) * inv;
                const float u2 = r4.x + (r3.x * dpx + r3.y * dpy) * inv;
                const CubeTap ct = cube_address(u0, u1, u2, a.R);
                const float w00 = (1.f - ct.fx) * (1.f - ct.fy), w01 = ct.fx * (1.f - ct.fy);
                const float w10 = (1.f - ct.fx) * ct.fy,         w11 = ct.fx * ct.fy;
                const float* p00 = tex + ct.o00; const float* p01 = tex + ct.o01;
                const float* p10 = tex + ct.o10; const float* p11 = tex + ct.o11;
                const float t0 = w00 * p00[0] + w01 * p01[0] + w10 * p10[0] + w11 * p11[0];
                const float t1 = w00 * p00[1] + w01 * p01[1] + w10 * p10[1] + w11 * p11[1];
                const float t2 = w00 * p00[2] + w01 * p01[2] + w10 * p10[2] + w11 * p11[2];
                const float c0 = fmaxf(0.f, TG_SH_C0 * t0 + r4.y + 0.5f);
                const float c1 = fmaxf(0.f, TG_SH_C0 * t1 + r4.z + 0.5f);
                const float c2 = fmaxf(0.f, TG_SH_C0 * t2 + r4.w + 0.5f);
                const float w = alpha * T;
                C0 += w * c0; C1 += w * c1; C2 += w * c2;
                Dp += w * r5.x; N0 += w * r5.y; N1 += w * r5.z; N2 += w * r5.w; Al += w;
                T = Tn;
                last = (uint32_t)(base + j + 1);
            }
            if (__ballot(!done) == 0ull) break;
        }
    }
    if (inside) {
        const int HW = a.W * a.H, pix = py * a.W + px;
        out_color[pix] = C0 + T * a.bg[0];
        out_color[HW + pix] = C1 + T * a.bg[1];
        out_color[2 * HW + pix] = C2 + T * a.bg[2];
        out_depth[pix] = Dp;
        out_norm[pix] = N0; out_norm[HW + pix] = N1; out_norm[2 * HW + pix] = N2;
        out_alpha[pix] = Al;
        final_T[pix] = T;
        n_contrib[pix] = last;
    }
}

// ------------------------------------------------------------------------------------------------ K7
__global__ void __launch_bounds__(TG_BLOCK)
k_render_bwd(PixArgs a, const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
             const float* __restrict__ dL_dcolor, const float* __restrict__ dL_ddepth,
             const float* __restrict__ dL_dnorm, const float* __restrict__ dL_dalpha,
             float* __restrict__ acc, float* __restrict__ dtex) {
    __shared__ float4 s_rec[6][TG_BLOCK];
    __shared__ float s_grad[TG_BLOCK][TEXGS_ACC_FLOATS];
    __shared__ uint32_t s_id[TG_BLOCK];
    __shared__ int s_max[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = tile_of_block(blockIdx.x, a.num_tiles);
    if (tile >= a.num_tiles) return;
    const int tile_x = tile % a.tiles_x, tile_y = tile / a.tiles_x;
    const int px = tile_x * TEXGS_TILE + ((wave & 1) << 3) + (lane & 7);
    const int py = tile_y * TEXGS_TILE + ((wave >> 1) << 3) + (lane >> 3);
    const bool inside = (px < a.W) && (py < a.H);
    const float pxf = (float)px, pyf = (float)py;
    const uint2 range = a.ranges[tile];
    const int HW = a.W * a.H, pix = py * a.W + px;
    const float* __restrict__ tex = a.texture;

    float Tfin = 1.f; int last = 0;
    float dpix[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // dL/d (r,g,b,depth,nx,ny,nz,alpha)
    if (inside) {
        Tfin = final_T[pix]; last = (int)n_contrib[pix];
        if (dL_dcolor) { dpix[0] = dL_dcolor[pix]; dpix[1] = dL_dcolor[HW + pix]; dpix[2] = dL_dcolor[2 * HW + pix]; }
        if (dL_ddepth) dpix[3] = dL_ddepth[pix];
        if (dL_dnorm) { dpix[4] = dL_dnorm[pix]; dpix[5] = dL_dnorm[HW + pix]; dpix[6] = dL_dnorm[2 * HW + pix]; }
        if (dL_dalpha) dpix[7] = dL_dalpha[pix];
    }
    const float bgdot = a.bg[0] * dpix[0] + a.bg[1] * dpix[1] + a.bg[2] * dpix[2];
    {
        const int wm = wave_max_i(last);
        if (lane == 0) s_max[wave] = wm;
    }
    __syncthreads();
    const int max_last = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
    const int wave_last = s_max[wave];

    float T = Tfin;
    float accum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float last_alpha = 0.f;
    float last_f[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    const int nb = (max_last + TG_BLOCK - 1) / TG_BLOCK;
    for (int b = nb - 1; b >= 0; --b) {
        const int base = b * TG_BLOCK;
        const int cnt = min(TG_BLOCK, max_last - base);
        __syncthreads();
        if (tid < cnt) {
            const uint32_t id = a.point_list[range.x + base + tid];
            s_id[tid] = id;
            const float4* __restrict__ r = a.rec + (size_t)id * (TEXGS_REC_FLOATS / 4);
            const float4 v0 = r[0], v1 = r[1], v2 = r[2], v3 = r[3], v4 = r[4], v5 = r[5];
            s_rec[0][tid] = v0; s_rec[1][tid] = v1; s_rec[2][tid] = v2;
            s_rec[3][tid] = v3; s_rec[4][tid] = v4; s_rec[5][tid] = v5;
        }
#pragma unroll
        for (int k = 0; k < TEXGS_ACC_FLOATS; ++k) s_grad[tid][k] = 0.f;
        __syncthreads();
        const int jhi = min(cnt, wave_last - base) - 1;          // nothing in this wave contributed beyond wave_last
        for (int j = jhi; j >= 0; --j) {
            const int pos = base + j;
            const float4 r0 = s_rec[0][j];
            const float4 r1 = s_rec[1][j];
            const float dx = r0.x - pxf, dy = r0.y - pyf;
            const float power = -0.5f * (r0.z * dx * dx + r1.x * dy * dy) - r0.w * dx * dy;
            const float Gs = __expf(power);
            const float araw = r1.y * Gs;
            const float alpha = fminf(TG_ALPHA_MAX, araw);
            const bool ok = inside && (pos < last) && (power <= 0.0f) && (alpha >= TG_ALPHA_MIN);
            if (__ballot(ok) == 0ull) continue;
            float part[TEXGS_ACC_FLOATS];
#pragma unroll
            for (int k = 0; k < TEXGS_ACC_FLOATS; ++k) part[k] = 0.f;
            if (ok) {
                const float4 r2 = s_rec[2][j];
                const float4 r3 = s_rec[3][j];
                const float4 r4 = s_rec[4][j];
                const float4 r5 = s_rec[5][j];
                const float one_m_a = 1.0f - alpha;
                T = T / one_m_a;
                const float w = alpha * T;
                // ---- recompute the texture branch
                const float dpx = -dx, dpy = -dy;
                const float den = 1.0f + r1.z * dpx + r1.w * dpy;
                const bool good = den >= TG_DEN_MIN;
                const float inv = good ? __builtin_amdgcn_rcpf(den) : 0.0f;
                const float nu0 = r2.x * dpx + r2.y * dpy, nu1 = r2.z * dpx + r2.w * dpy, nu2 = r3.x * dpx + r3.y * dpy;
                const float u0 = r3.z + nu0 * inv, u1 = r3.w + nu1 * inv, u2 = r4.x + nu2 * inv;
                const CubeTap ct = cube_address(u0, u1, u2, a.R);
                const float w00 = (1.f - ct.fx) * (1.f - ct.fy), w01 = ct.fx * (1.f - ct.fy);
                const float w10 = (1.f - ct.fx) * ct.fy,         w11 = ct.fx * ct.fy;
                float t00[3], t01[3], t10[3], t11[3], f[8];
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    t00[ch] = tex[ct.o00 + ch]; t01[ch] = tex[ct.o01 + ch];
                    t10[ch] = tex[ct.o10 + ch]; t11[ch] = tex[ct.o11 + ch];
                }
                const float vd[3] = {r4.y, r4.z, r4.w};
                float pre[3];
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    const float tv = w00 * t00[ch] + w01 * t01[ch] + w10 * t10[ch] + w11 * t11[ch];
                    pre[ch] = TG_SH_C0 * tv + vd[ch] + 0.5f;
                    f[ch] = fmaxf(0.f, pre[ch]);
                }
                f[3] = r5.x; f[4] = r5.y; f[5] = r5.z; f[6] = r5.w; f[7] = 1.0f;
                // ---- alpha gradient: suffix accumulation behind this Gaussian (lineage back-to-front replay)
                float dL_dalpha_ = 0.f;
#pragma unroll
                for (int ch = 0; ch < 8; ++ch) {
                    accum[ch] = last_alpha * last_f[ch] + (1.f - last_alpha) * accum[ch];
                    last_f[ch] = f[ch];
                    dL_dalpha_ += (f[ch] - accum[ch]) * dpix[ch];
                }
                last_alpha = alpha;
                dL_dalpha_ *= T;
                dL_dalpha_ += (-Tfin / one_m_a) * bgdot;
                // ---- 2D Gaussian: straight-through the 0.99 clamp (lineage)
                const float dL_dpower = araw * dL_dalpha_;
                const float gdx = -(r0.z * dx + r0.w * dy), gdy = -(r1.x * dy + r0.w * dx);
                part[R_XY]        = dL_dpower * gdx;
                part[R_XY + 1]    = dL_dpower * gdy;
                part[R_CONIC]     = -0.5f * dx * dx * dL_dpower;
                part[R_CONIC + 1] = -dx * dy * dL_dpower;
                part[R_CONIC + 2] = -0.5f * dy * dy * dL_dpower;
                part[R_OP]        = Gs * dL_dalpha_;
                // ---- per-Gaussian blended features
                part[R_DEPTH] = w * dpix[3];
                part[R_N] = w * dpix[4]; part[R_N + 1] = w * dpix[5]; part[R_N + 2] = w * dpix[6];
                // ---- colour -> view-dependent term, texture, uv
                float dtexv[3];
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    const float dcol = (pre[ch] > 0.f) ? w * dpix[ch] : 0.f;
                    part[R_VD + ch] = dcol;
                    dtexv[ch] = TG_SH_C0 * dcol;
                }
                float dLdcol = 0.f, dLdrow = 0.f;
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    if (dtexv[ch] != 0.f) {
                        unsafeAtomicAdd(dtex + ct.o00 + ch, w00 * dtexv[ch]);
                        unsafeAtomicAdd(dtex + ct.o01 + ch, w01 * dtexv[ch]);
                        unsafeAtomicAdd(dtex + ct.o10 + ch, w10 * dtexv[ch]);
                        unsafeAtomicAdd(dtex + ct.o11 + ch, w11 * dtexv[ch]);
                    }
                    dLdcol += dtexv[ch] * ((1.f - ct.fy) * (t01[ch] - t00[ch]) + ct.fy * (t11[ch] - t10[ch]));
                    dLdrow += dtexv[ch] * ((1.f - ct.fx) * (t10[ch] - t00[ch]) + ct.fx * (t11[ch] - t01[ch]));
                }
                // (col,row) -> uv
                const float dua = dLdcol * ct.su * ct.h, dub = dLdrow * ct.sv * ct.h;
                const float dum = -(dLdcol * ct.sc + dLdrow * ct.tc) * ct.h * ct.rma * ct.sm;
                float du0, du1, du2;
                if (ct.axis == 0)      { du0 = dum; du2 = dua; du1 = dub; }
                else if (ct.axis == 1) { du1 = dum; du0 = dua; du2 = dub; }
                else                   { du2 = dum; du0 = dua; du1 = dub; }
                part[R_PHI] = du0; part[R_PHI + 1] = du1; part[R_PHI + 2] = du2;
                if (good) {
                    const float dn0 = du0 * inv, dn1 = du1 * inv, dn2 = du2 * inv;     // dL/d num
                    const float dden = -(du0 * nu0 + du1 * nu1 + du2 * nu2) * inv * inv;
                    part[R_GM + 0] = dn0 * dpx; part[R_GM + 1] = dn0 * dpy;
                    part[R_GM + 2] = dn1 * dpx; part[R_GM + 3] = dn1 * dpy;
                    part[R_GM + 4] = dn2 * dpx; part[R_GM + 5] = dn2 * dpy;
                    part[R_G2] = dden * dpx; part[R_G2 + 1] = dden * dpy;
                    // dp = pix - xy
                    part[R_XY]     -= (r2.x * dn0 + r2.z * dn1 + r3.x * dn2) + r1.z * dden;
                    part[R_XY + 1] -= (r2.y * dn0 + r2.w * dn1 + r3.y * dn2) + r1.w * dden;
                }
            }
            // ---- reduce the 24 partials over the wave, accumulate per instance in LDS
#pragma unroll
            for (int k = 0; k < TEXGS_ACC_FLOATS; ++k) {
                const float s = wave_sum(part[k]);
                if (lane == 0) __hip_atomic_fetch_add(&s_grad[j][k], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        __syncthreads();
        // flush: 32 lanes per instance (24 used) -> consecutive addresses inside one 96-B accumulator row
        for (int idx = tid; idx < cnt * 32; idx += TG_BLOCK) {
            const int j = idx >> 5, k = idx & 31;
            if (k < TEXGS_ACC_FLOATS) {
                const float v = s_grad[j][k];
                if (v != 0.f) unsafeAtomicAdd(acc + (size_t)s_id[j] * TEXGS_ACC_FLOATS + k, v);
            }
        }
    }
}

inline PixArgs make_pix(const CamConst& c, const TexGSFrame* f, const TexGSInputs* in, const TexGSGeom* g,
                        const TexGSBinning* b) {
    PixArgs a;
    a.W = c.W; a.H = c.H; a.tiles_x = c.tiles_x; a.num_tiles = c.tiles_x * c.tiles_y; a.R = c.R;
    a.ranges = reinterpret_cast<const uint2*>(b->ranges);
    a.point_list = b->point_list;
    a.rec = reinterpret_cast<const float4*>(g->rec);
    a.texture = in->texture;
    a.bg = f->bg;
    return a;
}

}  // namespace

void launch_render_fwd(const CamConst& c, const TexGSFrame* f, const TexGSInputs* in, const TexGSGeom* g,
                       const TexGSBinning* b, TexGSImage* img, hipStream_t s) {
    const PixArgs a = make_pix(c, f, in, g, b);
    const int grid = ((a.num_tiles + 7) / 8) * 8;
    hipLaunchKernelGGL(k_render_fwd, dim3(grid), dim3(TG_BLOCK), 0, s, a, img->out_color, img->out_depth, img->out_norm,
                       img->out_alpha, img->final_T, img->n_contrib);
}

void launch_render_bwd(const CamConst& c, const TexGSFrame* f, const TexGSInputs* in, const TexGSGeom* g,
                       const TexGSBinning* b, const TexGSImage* img, TexGSGrads* gr, hipStream_t s) {
    const PixArgs a = make_pix(c, f, in, g, b);
    const int grid = ((a.num_tiles + 7) / 8) * 8;
    hipLaunchKernelGGL(k_render_bwd, dim3(grid), dim3(TG_BLOCK), 0, s, a, img->final_T, img->n_contrib,
                       gr->dL_dcolor, gr->dL_ddepth, gr->dL_dnorm, gr->dL_dalpha, gr->acc, gr->dL_dtexture);
}
