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
#include "wave_ops.h"
#include <stdlib.h>

namespace {

// blockIdx -> tile.  Workgroup b is observed to run on XCD b % 8 (speed only, never correctness).  Tiles are dealt to
// XCDs in groups of 8 row-adjacent tiles, cyclically: neighbours inside a group share most of their Gaussians'
// records (same 4 MiB L2), while every XCD still gets the same mix of light (border) and heavy (centre) tiles --
// contiguous per-XCD bands left the centre XCDs with ~1.6x the mean work.
__device__ __forceinline__ int tile_of_block(int b, int num_tiles) {
    const int xcd = b & 7, idx = b >> 3;
    return (((idx >> 3) << 3) + xcd) * 8 + (idx & 7);
}

struct __attribute__((packed, aligned(4))) Texel3 { float x, y, z; };   // one global_load_dwordx3 per tap
__device__ __forceinline__ Texel3 load_texel(const float* __restrict__ tex, int off) {
    return *reinterpret_cast<const Texel3*>(tex + off);
}

// Cubemap address of direction u (not necessarily unit): face (+x,-x,+y,-y,+z,-z; NVDIFFREC/util.py:94-101
// inverted), bilinear taps with clamp-to-edge inside the face, texel centres at (i+0.5)/R.
struct CubeTap {
    int   o00, o01, o10, o11;   // float offsets of the 4 taps' first channel
    int   x0, x1, y0, y1;       // clamped tap coordinates inside the face
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
    t.x0 = x0c; t.x1 = x1c; t.y0 = y0c; t.y1 = y1c;
    const int fb = face * R;
    t.o00 = ((fb + y0c) * R + x0c) * 3; t.o01 = ((fb + y0c) * R + x1c) * 3;
    t.o10 = ((fb + y1c) * R + x0c) * 3; t.o11 = ((fb + y1c) * R + x1c) * 3;
    return t;
}

#ifdef TEXGS_STATS
__device__ unsigned long long g_stats[16];
#define STAT(i, v) atomicAdd(&g_stats[i], (unsigned long long)(v))
#else
#define STAT(i, v) do {} while (0)
#endif

struct PixArgs {
    int W, H, tiles_x, num_tiles, R;
    uint32_t quad_dirty_index;      // float index of the dirty-face word inside the quad buffer
    const uint2* ranges;
    const uint32_t* tile_order;
    const uint32_t* point_list;
    const float4* rec;
    const float* texture;
    const float* bg;
};

// ------------------------------------------------------------------------------------------------ K6
// Forward blend.  One wave = one 8x8 pixel block; the 4 waves of a tile are independent (no block barrier, no LDS
// staging of records).  Per chunk of 64 instances lane l keeps instance l's whole 96-byte record in registers.
//   (sequential) every lane walks the chunk for its own pixel; the tested Gaussian's (xy, conic, opacity) arrive by
//       v_readlane broadcast -- no LDS traffic or LDS latency in the dependent chain; ~30 VALU per test.  Depth,
//       normal and alpha accumulate here (w = alpha*T needs no texture).
//   (dense)      contributing (pixel, j, w) triples are compacted with ballot + mbcnt into a 128-entry per-wave LDS
//       ring; whenever 64 are queued all 64 lanes pop one each, fetch the item's Gaussian fields from lane j's
//       registers through the LDS crossbar (ds_bpermute), do the UV Taylor step, cubemap addressing and 4 dwordx3
//       tap loads with full lane occupancy and 64 fetches in flight, then add w*colour into the pixel's LDS
//       accumulator.  The colour sum is order-independent, so this equals the in-order blend.
// (Only ~8 of 64 pixels of a wave contribute to a given Gaussian: with the texture path inside the sequential loop it
//  ran at ~12 % lane efficiency; with LDS-staged records the LDS was 50 % busy and 22 % of wave time was LDS issue stall.)
#define FQ_CAP 128

#ifndef FWD_WAVES_PER_SIMD
#define FWD_WAVES_PER_SIMD 8
#endif
template <int FABL>     // timing experiments only (0 = product): 1 skip the dense phase, 2 dense phase without loads
__global__ void __launch_bounds__(TG_BLOCK, FWD_WAVES_PER_SIMD)
k_render_fwd(PixArgs a, float* __restrict__ out_color, float* __restrict__ out_depth, float* __restrict__ out_norm,
             float* __restrict__ out_alpha, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib) {
    __shared__ uint2 s_qall[4][FQ_CAP];
    __shared__ float s_col[TG_BLOCK * 3];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if ((int)blockIdx.x >= a.num_tiles) return;
    const int tile = (int)a.tile_order[blockIdx.x];      // longest list first
    const int tile_x = tile % a.tiles_x, tile_y = tile / a.tiles_x;
    const int wave_px = tile_x * TEXGS_TILE + ((wave & 1) << 3), wave_py = tile_y * TEXGS_TILE + ((wave >> 1) << 3);
    const int px = wave_px + (lane & 7), py = wave_py + (lane >> 3);
    const bool inside = (px < a.W) && (py < a.H);
    const float pxf = (float)px, pyf = (float)py;
    const uint2 range = a.ranges[tile];
    const int todo = (int)(range.y - range.x);
    const float* __restrict__ tex = a.texture;
    uint2* s_q = s_qall[wave];
    float* s_c = s_col + wave * 192;

    s_col[tid * 3 + 0] = 0.f; s_col[tid * 3 + 1] = 0.f; s_col[tid * 3 + 2] = 0.f;   // own pixel; only this wave touches it
    __builtin_amdgcn_wave_barrier();

    bool done = !inside;
    float T = 1.0f;
    float Dp = 0.f, N0 = 0.f, N1 = 0.f, N2 = 0.f, Al = 0.f;
    uint32_t last = 0;
    int qhead = 0, qtail = 0;                                  // wave-uniform

#define FWD_DRAIN(NITEMS)                                                                                              \
    do {                                                                                                               \
        const int n_ = (NITEMS);                                                                                       \
        uint2 e_ = make_uint2(0u, 0u);                                                                                 \
        if (lane < n_) e_ = s_q[(qhead + lane) & (FQ_CAP - 1)];                                                        \
        const int pl_ = (int)(e_.y >> 8) & 63, jj_ = (int)(e_.y & 63u);                                                \
        const float gx_ = BP(r0.x, jj_), gy_ = BP(r0.y, jj_), g0_ = BP(r1.z, jj_), g1_ = BP(r1.w, jj_);                \
        const float G00 = BP(r2.x, jj_), G01 = BP(r2.y, jj_), G10 = BP(r2.z, jj_), G11 = BP(r2.w, jj_);                \
        const float G20 = BP(r3.x, jj_), G21 = BP(r3.y, jj_), ph0 = BP(r3.z, jj_), ph1 = BP(r3.w, jj_);                \
        const float ph2 = BP(r4.x, jj_), vd0 = BP(r4.y, jj_), vd1 = BP(r4.z, jj_), vd2 = BP(r4.w, jj_);                \
        if (lane < n_) {                                                                                               \
            const float w_ = __uint_as_float(e_.x);                                                                    \
            const float dpx = (float)(wave_px + (pl_ & 7)) - gx_, dpy = (float)(wave_py + (pl_ >> 3)) - gy_;           \
            const float den = 1.0f + g0_ * dpx + g1_ * dpy;                                                            \
            const float inv = (den >= TG_DEN_MIN) ? __builtin_amdgcn_rcpf(den) : 0.0f;                                 \
            const float u0 = ph0 + (G00 * dpx + G01 * dpy) * inv;                                                      \
            const float u1 = ph1 + (G10 * dpx + G11 * dpy) * inv;                                                      \
            const float u2 = ph2 + (G20 * dpx + G21 * dpy) * inv;                                                      \
            const CubeTap ct = cube_address(u0, u1, u2, a.R);                                                          \
            const float w00 = (1.f - ct.fx) * (1.f - ct.fy), w01 = ct.fx * (1.f - ct.fy);                              \
            const float w10 = (1.f - ct.fx) * ct.fy,         w11 = ct.fx * ct.fy;                                      \
            Texel3 q00 = {0.1f, 0.2f, 0.3f}, q01 = q00, q10 = q00, q11 = q00;                                          \
            if (FABL != 2) { q00 = load_texel(tex, ct.o00); q01 = load_texel(tex, ct.o01);                             \
                             q10 = load_texel(tex, ct.o10); q11 = load_texel(tex, ct.o11); }                           \
            const float t0 = w00 * q00.x + w01 * q01.x + w10 * q10.x + w11 * q11.x;                                    \
            const float t1 = w00 * q00.y + w01 * q01.y + w10 * q10.y + w11 * q11.y;                                    \
            const float t2 = w00 * q00.z + w01 * q01.z + w10 * q10.z + w11 * q11.z;                                    \
            float* cp = s_c + pl_ * 3;                                                                                 \
            __hip_atomic_fetch_add(cp + 0, w_ * fmaxf(0.f, TG_SH_C0 * t0 + vd0 + 0.5f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
            __hip_atomic_fetch_add(cp + 1, w_ * fmaxf(0.f, TG_SH_C0 * t1 + vd1 + 0.5f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
            __hip_atomic_fetch_add(cp + 2, w_ * fmaxf(0.f, TG_SH_C0 * t2 + vd2 + 0.5f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
        }                                                                                                              \
    } while (0)
#define BP(V, J) __int_as_float(__builtin_amdgcn_ds_bpermute((J) << 2, __float_as_int(V)))
#define RLF(V, J) __int_as_float(__builtin_amdgcn_readlane(__float_as_int(V), (J)))

    for (int base = 0; base < todo; base += 64) {
        if (__ballot(!done) == 0ull) break;
        const int cnt = min(64, todo - base);
        float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0, r3 = r0, r4 = r0, r5 = r0, r6 = make_float4(-1.f, 1.f, 0.f, 0.f);
        if (lane < cnt) {
            const uint32_t id = a.point_list[range.x + base + lane];
            const float4* __restrict__ r = a.rec + (size_t)id * (TEXGS_REC_FLOATS / 4);
            r0 = r[0]; r1 = r[1]; r2 = r[2]; r3 = r[3]; r4 = r[4]; r5 = r[5]; r6 = r[6];
        }
        // per-wave cull, lane-parallel: can instance `lane` reach alpha >= 1/255 anywhere in this wave's 8x8 block?
        unsigned long long todo_mask = __ballot((r0.x + r6.x >= (float)wave_px) && (r0.x - r6.x <= (float)(wave_px + 7)) &&
                                                (r0.y + r6.x >= (float)wave_py) && (r0.y - r6.x <= (float)(wave_py + 7)));
        while (todo_mask != 0ull) {
            const int j = __ffsll((long long)todo_mask) - 1;
            todo_mask &= todo_mask - 1ull;
            const float gx_ = RLF(r0.x, j), gy_ = RLF(r0.y, j), ca = RLF(r0.z, j), cb = RLF(r0.w, j);
            const float cc = RLF(r1.x, j), thr = RLF(r6.y, j);
            const float dx = gx_ - pxf, dy = gy_ - pyf;
            const float power = -0.5f * (ca * dx * dx + cc * dy * dy) - cb * dx * dy;
            if (__ballot((!done) && (power <= 0.0f) && (power >= thr)) == 0ull) continue;   // conservative prefilter
            const float op = RLF(r1.y, j);
            const float alpha = fminf(TG_ALPHA_MAX, op * __expf(power));
            bool ok = (!done) && (power <= 0.0f) && (alpha >= TG_ALPHA_MIN);
            const float Tn = T * (1.0f - alpha);
            if (ok && Tn < TG_T_EPS) { done = true; ok = false; }
            const unsigned long long bal = __ballot(ok);
            if (bal != 0ull) {
                const float dep = RLF(r5.x, j), n0 = RLF(r5.y, j), n1 = RLF(r5.z, j), n2 = RLF(r5.w, j);
                if (ok) {
                    const float w = alpha * T;
                    Dp += w * dep; N0 += w * n0; N1 += w * n1; N2 += w * n2; Al += w;
                    T = Tn;
                    last = (uint32_t)(base + j + 1);
                    const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32),
                                          __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                    s_q[(qtail + rank) & (FQ_CAP - 1)] = make_uint2(__float_as_uint(w), ((uint32_t)lane << 8) | (uint32_t)j);
                }
                qtail += __popcll(bal);
                if (qtail - qhead >= 64) {
                    __builtin_amdgcn_wave_barrier();
                    if (FABL != 1) FWD_DRAIN(64);
                    qhead += 64;
                }
                if (__ballot(!done) == 0ull) break;
            }
        }
        // items reference this chunk's registers: finish them before the next chunk is loaded
        if (qtail - qhead > 0) {
            __builtin_amdgcn_wave_barrier();
            if (FABL != 1) FWD_DRAIN(qtail - qhead);
            qhead = qtail;
        }
    }
#undef FWD_DRAIN
#undef BP
#undef RLF
    __builtin_amdgcn_wave_barrier();
    if (inside) {
        const int HW = a.W * a.H, pix = py * a.W + px;
        out_color[pix] = s_col[tid * 3 + 0] + T * a.bg[0];
        out_color[HW + pix] = s_col[tid * 3 + 1] + T * a.bg[1];
        out_color[2 * HW + pix] = s_col[tid * 3 + 2] + T * a.bg[2];
        out_depth[pix] = Dp;
        out_norm[pix] = N0; out_norm[HW + pix] = N1; out_norm[2 * HW + pix] = N2;
        out_alpha[pix] = Al;
        final_T[pix] = T;
        n_contrib[pix] = last;
    }
}

// ------------------------------------------------------------------------------------------------ K7
// Backward replay.  What the counters said about the first version (rocprofv3, profiles/r01_*):
//   * 124 M L2 misses / 4.5 GB written per launch: on this multi-XCD part every global fp32 atomic executes
//     memory-side, and a wave instruction whose 64 lanes hit 64 unrelated dwords is 64 requests (~21 G req/s chip
//     wide); 6 adjacent lanes -> 6 adjacent dwords is 4.6x cheaper, 64 consecutive dwords 12.8x (scripts/ubench);
//   * only ~8 of 64 pixels of a wave contribute to a given Gaussian, so texture math inside the per-pixel loop ran
//     at ~12 % lane efficiency and the 24-value wave reduction ran for every (wave, Gaussian).
// Structure now (one wave = one 8x8 pixel block, the 4 waves of a tile are independent; no block barrier in the loop):
//   per chunk of 64 instances (back to front) lane l keeps instance l's (xy, conic, opacity, depth, normal) in
//   registers; the sequential loops broadcast them with v_readlane -- no LDS traffic, no LDS latency in the chain.
//   stage A  sequential, ~30 VALU / test: falloff, alpha, T /= (1-alpha); contributing (pixel, j) pairs are
//            compacted (ballot + mbcnt) into an LDS item list {T, alpha_raw, q, key}; per-j ballots stay in VGPRs.
//   stage B  dense, 64 items per round: UV Taylor step, cubemap address, 4 dwordx3 tap loads, colour; stores per item
//            q = colour . dL/dpixel (for the suffix recurrence) and dL/dcolour (3), dL/duv (3), 1/den, dL/dden;
//            the 12 texture-gradient updates of each pair are transposed through LDS so adjacent lanes issue
//            adjacent dwords of a tap row.  (A first cut added the uv-path partials to LDS accumulators with
//            ds_add_f32: ~8 lanes per address serialise, 1.5 ms per launch.)
//   stage C  sequential, scalar suffix recurrence dL/dalpha = T (s - suffix) + bg term with s = q + geometry
//            channels; all 24 per-Gaussian partials are formed by the owning pixel lane, reduced over the wave with
//            ONE transposing butterfly (value k ends in lane k, DPP + permlane swaps only) and lanes 0..23 add
//            24 consecutive dwords of the accumulator row: one coalesced memory-side request.
#ifndef BQ_CAP
#define BQ_CAP 128
#endif
// Tile-local texture-gradient cache: direct-mapped, toroidal spatial hash slot = (x mod 64) + 64 * (y mod TC_H), tag =
// texel offset.  A footprint narrower than 64 x TC_H texels is collision-free wherever it sits; aliasing updates (other
// face, far side, parallax spread) fall through to the transposed global atomics.  Flushed once per tile, coalesced.
#ifndef TC_ENABLE
#define TC_ENABLE 0      // measured: with the quad layout a miss is ONE request; the cache's LDS ops cost more than they save
#endif
#define TC_W 64
#ifndef TC_H
#define TC_H 32
#endif
#define TC_SLOTS (TC_W * TC_H)
#define TC_EMPTY 0xFFFFFFFFu
#ifndef TC_SECOND_CHANCE
#define TC_SECOND_CHANCE 0
#endif

#ifndef BWD_WAVES_PER_SIMD
#define BWD_WAVES_PER_SIMD 2
#endif
template <int ABL>      // timing experiments only (0 = product): 1 no texture atomics, 2 no stage-C reduce, 8 no tap loads
__global__ void __launch_bounds__(TG_BLOCK, BWD_WAVES_PER_SIMD)
k_render_bwd(PixArgs a, const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
             const float* __restrict__ dL_dcolor, const float* __restrict__ dL_ddepth,
             const float* __restrict__ dL_dnorm, const float* __restrict__ dL_dalpha,
             float* __restrict__ acc, float* __restrict__ dtex, float* __restrict__ quads) {
    __shared__ float4 s_items_all[4][BQ_CAP * 3];             // 3 float4 per item: {T, araw, q, key} {dc, du0} {du1, du2, inv, dden}
    __shared__ float s_sval_all[4][64 * 13];                  // 13 KB: 12 texture-gradient dwords per pair (+1 pad)
    __shared__ uint32_t s_sbase_all[4][64];                   // 1 KB: their base offset
    __shared__ float s_dpix[TG_BLOCK * 3];                    // 3 KB
    __shared__ float4 s_recs_all[4][6 * 64];                  // 24 KB: the chunk's records, plane-major [k][lane]
#if TC_ENABLE
    __shared__ uint32_t s_ttag[TC_SLOTS];                     // 8 KB
    __shared__ float s_tval[TC_SLOTS * 3];                    // 24 KB
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if ((int)blockIdx.x >= a.num_tiles) return;
    const int tile = (int)a.tile_order[blockIdx.x];      // longest list first
    const int tile_x = tile % a.tiles_x, tile_y = tile / a.tiles_x;
    const int wave_px = tile_x * TEXGS_TILE + ((wave & 1) << 3), wave_py = tile_y * TEXGS_TILE + ((wave >> 1) << 3);
    const int px = wave_px + (lane & 7), py = wave_py + (lane >> 3);
    const bool inside = (px < a.W) && (py < a.H);
    const float pxf = (float)px, pyf = (float)py;
    const uint2 range = a.ranges[tile];
    const int todo = (int)(range.y - range.x);
    const int HW = a.W * a.H, pix = py * a.W + px;
    const float* __restrict__ tex = a.texture;
    float4* s_items = s_items_all[wave];
    float* s_sval = s_sval_all[wave];
    float4* s_recs = s_recs_all[wave];
    uint32_t* s_sbase = s_sbase_all[wave];

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
    s_dpix[tid * 3 + 0] = dpix[0]; s_dpix[tid * 3 + 1] = dpix[1]; s_dpix[tid * 3 + 2] = dpix[2];
#if TC_ENABLE
    for (int k = tid; k < TC_SLOTS; k += TG_BLOCK) { s_ttag[k] = TC_EMPTY; s_tval[3 * k] = 0.f; s_tval[3 * k + 1] = 0.f; s_tval[3 * k + 2] = 0.f; }
    __syncthreads();
#endif
    const int wave_last = min(wave_max_i(last), todo);
    __builtin_amdgcn_wave_barrier();

    float T = Tfin;
    float suffix = 0.f, last_alpha = 0.f, last_s = 0.f;
    uint32_t faces_marked = 0u;                                // wave-uniform: faces already flagged dirty

    const int nchunks = (wave_last + 63) >> 6;
    for (int c = nchunks - 1; c >= 0; --c) {
        const int base = c << 6;
        const int jtop = min(64, wave_last - base);           // instances [0, jtop) of this chunk matter
        // ---- lane l <- instance l of the chunk
        uint32_t id = 0;
        float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2v = r0, r3v = r0, r4v = r0, r5 = r0, r6 = make_float4(-1.f, 1.f, 0.f, 0.f);
        if (lane < jtop) {
            id = a.point_list[range.x + base + lane];
            const float4* __restrict__ r = a.rec + (size_t)id * (TEXGS_REC_FLOATS / 4);
            r0 = r[0]; r1 = r[1]; r2v = r[2]; r3v = r[3]; r4v = r[4]; r5 = r[5]; r6 = r[6];
        }
        // stage A broadcasts from registers (v_readlane: no LDS latency in its dependent chain); stages B and C fetch the
        // per-Gaussian fields from this LDS copy (stage C: 6 broadcast ds_read_b128 instead of ~29 v_readlane whose SGPR
        // results collide with gfx9's one-SGPR-per-VALU constant-bus limit; stage B: per-lane gather instead of 16 ds_bpermute)
        __builtin_amdgcn_wave_barrier();
        s_recs[0 * 64 + lane] = r0; s_recs[1 * 64 + lane] = r1; s_recs[2 * 64 + lane] = r2v;
        s_recs[3 * 64 + lane] = r3v; s_recs[4 * 64 + lane] = r4v; s_recs[5 * 64 + lane] = r5;
        __builtin_amdgcn_wave_barrier();
        // per-wave cull (see K6): instances that cannot reach alpha >= 1/255 inside this wave's 8x8 block are never visited
        if (lane == 0) STAT(8, 1);                                      /* wave-chunks */
        const unsigned long long cull_mask = __ballot((r0.x + r6.x >= (float)wave_px) && (r0.x - r6.x <= (float)(wave_px + 7)) &&
                                                      (r0.y + r6.x >= (float)wave_py) && (r0.y - r6.x <= (float)(wave_py + 7)));
        uint32_t touched_lo = 0u, touched_hi = 0u;           // lane j keeps the stage-A ballot of instance j
        if (lane == 0) STAT(0, __popcll(cull_mask));                     /* tests after per-wave cull */
        unsigned long long amask = cull_mask;                 // instances still to be tested (stage A), high to low
        while (amask != 0ull) {
            // ================================================================ stage A
            int n_items = 0;
            unsigned long long seg_mask = 0ull;                 // instances of this segment that produced items
            while (amask != 0ull) {
                const int j = 63 - __clzll((long long)amask);
                const unsigned long long jbit = 1ull << j;
                const float gx_ = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r0.x), j));
                const float gy_ = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r0.y), j));
                const float ca = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r0.z), j));
                const float cb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r0.w), j));
                const float cc = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r1.x), j));
                const float thr = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r6.y), j));
                const float dx = gx_ - pxf, dy = gy_ - pyf;
                const float power = -0.5f * (ca * dx * dx + cc * dy * dy) - cb * dx * dy;
                if (__ballot(inside && (base + j < last) && (power <= 0.0f) && (power >= thr)) == 0ull) { amask &= ~jbit; continue; }
                const float op = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r1.y), j));
                const float araw = op * __expf(power);
                const float alpha = fminf(TG_ALPHA_MAX, araw);
                const bool ok = inside && (base + j < last) && (power <= 0.0f) && (alpha >= TG_ALPHA_MIN);
                const unsigned long long bal = __ballot(ok);
                const int nb = __popcll(bal);
                if (lane == 0) STAT(1, 1);                              /* full tests (post prefilter) */
                if (nb == 0) { amask &= ~jbit; continue; }
                if (n_items + nb > BQ_CAP) break;                   // segment full; j is re-tested in the next one
                amask &= ~jbit;
                seg_mask |= jbit;
                if (lane == j) { touched_lo = (uint32_t)bal; touched_hi = (uint32_t)(bal >> 32); }
                if (ok) {
                    T = T * __builtin_amdgcn_rcpf(1.0f - alpha);     // v_rcp_f32 (1 ulp): an IEEE divide is ~10 VALU
                    const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32),
                                          __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                    s_items[(n_items + rank) * 3] = make_float4(T, araw, 0.f, __uint_as_float(((uint32_t)lane << 8) | (uint32_t)j));
                }
                n_items += nb;
            }
            __builtin_amdgcn_wave_barrier();
            // ================================================================ stage B
            if (lane == 0) { STAT(2, n_items); STAT(3, (n_items + 63) / 64); STAT(4, 1); }   /* items, rounds, segments */
            for (int r = 0; r < n_items; r += 64) {
                const int e = r + lane;
                const bool have = e < n_items;
                uint32_t toff[12]; float tval[12];
#pragma unroll
                for (int k = 0; k < 12; ++k) { toff[k] = 0u; tval[k] = 0.f; }
                float4 it = make_float4(1.f, 0.f, 0.f, 0.f);
                if (have) it = s_items[e * 3];
                const uint32_t key = __float_as_uint(it.w);
                const int pl = (int)(key >> 8) & 63, jj = (int)(key & 63u);
                const float4 q0 = s_recs[0 * 64 + jj], q1 = s_recs[1 * 64 + jj], r2 = s_recs[2 * 64 + jj],
                             r3 = s_recs[3 * 64 + jj], r4 = s_recs[4 * 64 + jj];
                if (have) {
                    const float alpha = fminf(TG_ALPHA_MAX, it.y);
                    const float w = alpha * it.x;
                    const float ipx = (float)(wave_px + (pl & 7)), ipy = (float)(wave_py + (pl >> 3));
                    const float dpx = ipx - q0.x, dpy = ipy - q0.y;
                    const float den = 1.0f + q1.z * dpx + q1.w * dpy;
                    const bool good = den >= TG_DEN_MIN;
                    const float inv = good ? __builtin_amdgcn_rcpf(den) : 0.0f;
                    const float nu0 = r2.x * dpx + r2.y * dpy, nu1 = r2.z * dpx + r2.w * dpy, nu2 = r3.x * dpx + r3.y * dpy;
                    const float u0 = r3.z + nu0 * inv, u1 = r3.w + nu1 * inv, u2 = r4.x + nu2 * inv;
                    const CubeTap ct = cube_address(u0, u1, u2, a.R);
                    const float w00 = (1.f - ct.fx) * (1.f - ct.fy), w01 = ct.fx * (1.f - ct.fy);
                    const float w10 = (1.f - ct.fx) * ct.fy,         w11 = ct.fx * ct.fy;
                    Texel3 t00 = {0.1f, 0.2f, 0.3f}, t01 = t00, t10 = t00, t11 = t00;
                    if (!(ABL & 8)) {
                        t00 = load_texel(tex, ct.o00); t01 = load_texel(tex, ct.o01);
                        t10 = load_texel(tex, ct.o10); t11 = load_texel(tex, ct.o11);
                    }
                    const float d0 = s_dpix[(wave * 64 + pl) * 3 + 0], d1 = s_dpix[(wave * 64 + pl) * 3 + 1],
                                d2 = s_dpix[(wave * 64 + pl) * 3 + 2];
                    const float pre0 = TG_SH_C0 * (w00 * t00.x + w01 * t01.x + w10 * t10.x + w11 * t11.x) + r4.y + 0.5f;
                    const float pre1 = TG_SH_C0 * (w00 * t00.y + w01 * t01.y + w10 * t10.y + w11 * t11.y) + r4.z + 0.5f;
                    const float pre2 = TG_SH_C0 * (w00 * t00.z + w01 * t01.z + w10 * t10.z + w11 * t11.z) + r4.w + 0.5f;
                    const float qv = fmaxf(0.f, pre0) * d0 + fmaxf(0.f, pre1) * d1 + fmaxf(0.f, pre2) * d2;
                    // colour -> view-dependent term and texture
                    const float dc0 = (pre0 > 0.f) ? w * d0 : 0.f, dc1 = (pre1 > 0.f) ? w * d1 : 0.f, dc2 = (pre2 > 0.f) ? w * d2 : 0.f;
                    const float x0 = TG_SH_C0 * dc0, x1 = TG_SH_C0 * dc1, x2 = TG_SH_C0 * dc2;
                    toff[0] = ct.o00; toff[1] = ct.o00 + 1; toff[2] = ct.o00 + 2; toff[3] = ct.o01; toff[4] = ct.o01 + 1; toff[5] = ct.o01 + 2;
                    toff[6] = ct.o10; toff[7] = ct.o10 + 1; toff[8] = ct.o10 + 2; toff[9] = ct.o11; toff[10] = ct.o11 + 1; toff[11] = ct.o11 + 2;
                    tval[0] = w00 * x0; tval[1] = w00 * x1; tval[2] = w00 * x2; tval[3] = w01 * x0; tval[4] = w01 * x1; tval[5] = w01 * x2;
                    tval[6] = w10 * x0; tval[7] = w10 * x1; tval[8] = w10 * x2; tval[9] = w11 * x0; tval[10] = w11 * x1; tval[11] = w11 * x2;
#if TC_ENABLE
                    if (!(ABL & 4)) {
                        // cache probe per tap: hit -> LDS accumulate and drop the global update.  Tags are read
                        // first (4 independent ds_read in flight); the CAS runs only on a tap's first touch.
                        const int tx_[4] = {ct.x0, ct.x1, ct.x0, ct.x1}, ty_[4] = {ct.y0, ct.y0, ct.y1, ct.y1};
                        int slot_[4]; uint32_t cur_[4];
#pragma unroll
                        for (int tp = 0; tp < 4; ++tp) {
                            slot_[tp] = (tx_[tp] & (TC_W - 1)) | ((ty_[tp] & (TC_H - 1)) << 6);
                            cur_[tp] = s_ttag[slot_[tp]];
                        }
#pragma unroll
                        for (int tp = 0; tp < 4; ++tp) {
                            const uint32_t tag = toff[tp * 3];
                            uint32_t old = cur_[tp];
                            if (old == TC_EMPTY) old = atomicCAS(&s_ttag[slot_[tp]], TC_EMPTY, tag);
#if TC_SECOND_CHANCE
                            if (old != TC_EMPTY && old != tag) {     // taken by another texel: one more try at a hashed slot
                                slot_[tp] = (int)((tag * 2654435761u) >> (32 - 11)) & (TC_SLOTS - 1);
                                old = s_ttag[slot_[tp]];
                                if (old == TC_EMPTY) old = atomicCAS(&s_ttag[slot_[tp]], TC_EMPTY, tag);
                            }
#endif
                            STAT((old == TC_EMPTY || old == tag) ? 5 : 6, 1);           /* cache hits / misses (taps) */
                            if (old == TC_EMPTY || old == tag) {
#pragma unroll
                                for (int ch = 0; ch < 3; ++ch) {
                                    const float v_ = tval[tp * 3 + ch];
                                    if (v_ != 0.f) __hip_atomic_fetch_add(&s_tval[slot_[tp] * 3 + ch], v_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                    tval[tp * 3 + ch] = 0.f;
                                }
                            }
                        }
                    }
#endif
                    const float dLdcol = x0 * ((1.f - ct.fy) * (t01.x - t00.x) + ct.fy * (t11.x - t10.x))
                                       + x1 * ((1.f - ct.fy) * (t01.y - t00.y) + ct.fy * (t11.y - t10.y))
                                       + x2 * ((1.f - ct.fy) * (t01.z - t00.z) + ct.fy * (t11.z - t10.z));
                    const float dLdrow = x0 * ((1.f - ct.fx) * (t10.x - t00.x) + ct.fx * (t11.x - t01.x))
                                       + x1 * ((1.f - ct.fx) * (t10.y - t00.y) + ct.fx * (t11.y - t01.y))
                                       + x2 * ((1.f - ct.fx) * (t10.z - t00.z) + ct.fx * (t11.z - t01.z));
                    const float dua = dLdcol * ct.su * ct.h, dub = dLdrow * ct.sv * ct.h;
                    const float dum = -(dLdcol * ct.sc + dLdrow * ct.tc) * ct.h * ct.rma * ct.sm;
                    float du0, du1, du2;
                    if (ct.axis == 0)      { du0 = dum; du2 = dua; du1 = dub; }
                    else if (ct.axis == 1) { du1 = dum; du0 = dua; du2 = dub; }
                    else                   { du2 = dum; du0 = dua; du1 = dub; }
                    const float dden = -(du0 * nu0 + du1 * nu1 + du2 * nu2) * inv * inv;   // inv = 0 when !good
                    s_items[e * 3].z = qv;
                    s_items[e * 3 + 1] = make_float4(dc0, dc1, dc2, du0);
                    s_items[e * 3 + 2] = make_float4(du1, du2, inv, dden);
                }
                if (!(ABL & 1)) {
                    // Texture-gradient scatter of the cache misses.  Regular footprints (x1 = x0+1, y1 = y0+1) go to the
                    // quad arrays: the 12 dwords of a pair are ONE aligned 48-byte run, and the (pair, k) -> lane transpose
                    // below makes 12 adjacent lanes carry it, i.e. one memory-side request per footprint.  Clamped
                    // footprints at face borders (rare) are scattered straight into dL_dtexture by the owning lane.
                    uint32_t qbase = 0u;
                    bool regular = false;
                    int qface = -1;
                    if (have && quads != nullptr) {
                        const uint32_t o00 = toff[0] / 3u;                       // texel index (face*R + y0)*R + x0
                        const int Rr = a.R;
                        const int x0 = (int)(o00 % (uint32_t)Rr), yf = (int)(o00 / (uint32_t)Rr);
                        const int y0 = yf % Rr, face = yf / Rr;
                        regular = (toff[3] == toff[0] + 3u) && (toff[6] == toff[0] + 3u * (uint32_t)Rr) && (toff[9] == toff[6] + 3u);
                        const int QW = (Rr >> 1) + 1;
                        const int phase = (x0 & 1) | ((y0 & 1) << 1);
                        qbase = (uint32_t)((((phase * 6 + face) * QW + (y0 >> 1)) * QW + (x0 >> 1)) * 16);
                        if (regular) qface = face;
                    }
                    if (quads != nullptr) {        // dirty-face word (last slot of the quad buffer): <= 6 atomics per wave overall
                        uint32_t fm = 0u;
#pragma unroll
                        for (int f = 0; f < 6; ++f) fm |= (__ballot(qface == f) != 0ull) ? (1u << f) : 0u;
                        const uint32_t fresh = fm & ~faces_marked;
                        if (fresh != 0u && lane == 0) atomicOr(reinterpret_cast<uint32_t*>(quads) + a.quad_dirty_index, fresh);
                        faces_marked |= fm;
                    }
                    if (have && !regular) {
#pragma unroll
                        for (int k = 0; k < 12; ++k) { if (tval[k] != 0.f) unsafeAtomicAdd(dtex + toff[k], tval[k]); tval[k] = 0.f; }
                    }
                    s_sbase[lane] = qbase;
#pragma unroll
                    for (int k = 0; k < 12; ++k) s_sval[lane * 13 + k] = tval[k];
                    __builtin_amdgcn_wave_barrier();
                    const int nent = min(64, n_items - r) * 12;
#pragma unroll
                    for (int it2 = 0; it2 < 12; ++it2) {
                        const int ee = it2 * 64 + lane;
                        if (ee < nent) {
                            const int pr = ee / 12, k = ee - pr * 12;
                            const float v = s_sval[pr * 13 + k];
                            if (v != 0.f) unsafeAtomicAdd(quads + s_sbase[pr] + k, v);
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
            __builtin_amdgcn_wave_barrier();
            // ================================================================ stage C
            int it0 = 0;
            while (seg_mask != 0ull) {
                const int jj = 63 - __clzll((long long)seg_mask);
                seg_mask &= ~(1ull << jj);
                const uint32_t blo = (uint32_t)__builtin_amdgcn_readlane((int)touched_lo, jj);
                const uint32_t bhi = (uint32_t)__builtin_amdgcn_readlane((int)touched_hi, jj);
                const unsigned long long bal = ((unsigned long long)bhi << 32) | blo;
                const bool ok = (bal >> lane) & 1ull;
                float part[32];
#pragma unroll
                for (int k = 0; k < 32; ++k) part[k] = 0.f;
                const float4 c0 = s_recs[0 * 64 + jj], c1 = s_recs[1 * 64 + jj], c2 = s_recs[2 * 64 + jj],
                             c3 = s_recs[3 * 64 + jj], c5 = s_recs[5 * 64 + jj];             // uniform address: LDS broadcast
                if (ok) {
                    const int rank = (int)__builtin_amdgcn_mbcnt_hi(bhi, __builtin_amdgcn_mbcnt_lo(blo, 0u));
                    const float4 i0 = s_items[(it0 + rank) * 3], i1 = s_items[(it0 + rank) * 3 + 1], i2 = s_items[(it0 + rank) * 3 + 2];
                    const float Ti = i0.x, araw = i0.y, qv = i0.z;
                    const float alpha = fminf(TG_ALPHA_MAX, araw);
                    const float gx_ = c0.x, gy_ = c0.y, ca = c0.z, cb = c0.w, cc = c1.x, op = c1.y;
                    const float dep = c5.x, n0 = c5.y, n1 = c5.z, n2 = c5.w;
                    const float dx = gx_ - pxf, dy = gy_ - pyf;
                    const float w = alpha * Ti;
                    const float s_i = qv + dep * dpix[3] + n0 * dpix[4] + n1 * dpix[5] + n2 * dpix[6] + dpix[7];
                    suffix = last_alpha * last_s + (1.f - last_alpha) * suffix;
                    last_s = s_i; last_alpha = alpha;
                    const float dL_dalpha_ = (s_i - suffix) * Ti - Tfin * __builtin_amdgcn_rcpf(1.0f - alpha) * bgdot;
                    const float dL_dpower = araw * dL_dalpha_;        // straight through the 0.99 clamp (lineage)
                    const float gdx = -(ca * dx + cb * dy), gdy = -(cc * dy + cb * dx);
                    // uv path (stage B results): dn = du * inv, dp = pix - xy
                    const float du0 = i1.w, du1 = i2.x, du2 = i2.y, inv = i2.z, dden = i2.w;
                    const float dn0 = du0 * inv, dn1 = du1 * inv, dn2 = du2 * inv;
                    const float dpx = -dx, dpy = -dy;
                    const float ggx = c1.z, ggy = c1.w;
                    const float G00 = c2.x, G01 = c2.y, G10 = c2.z, G11 = c2.w, G20 = c3.x, G21 = c3.y;
                    part[R_XY]        = dL_dpower * gdx - ((G00 * dn0 + G10 * dn1 + G20 * dn2) + ggx * dden);
                    part[R_XY + 1]    = dL_dpower * gdy - ((G01 * dn0 + G11 * dn1 + G21 * dn2) + ggy * dden);
                    part[R_CONIC]     = -0.5f * dx * dx * dL_dpower;
                    part[R_CONIC + 1] = -dx * dy * dL_dpower;
                    part[R_CONIC + 2] = -0.5f * dy * dy * dL_dpower;
                    part[R_OP]        = araw * __builtin_amdgcn_rcpf(op) * dL_dalpha_;
                    part[R_G2] = dden * dpx; part[R_G2 + 1] = dden * dpy;
                    part[R_GM + 0] = dn0 * dpx; part[R_GM + 1] = dn0 * dpy;
                    part[R_GM + 2] = dn1 * dpx; part[R_GM + 3] = dn1 * dpy;
                    part[R_GM + 4] = dn2 * dpx; part[R_GM + 5] = dn2 * dpy;
                    part[R_PHI] = du0; part[R_PHI + 1] = du1; part[R_PHI + 2] = du2;
                    part[R_VD] = i1.x; part[R_VD + 1] = i1.y; part[R_VD + 2] = i1.z;
                    part[R_DEPTH] = w * dpix[3];
                    part[R_N] = w * dpix[4]; part[R_N + 1] = w * dpix[5]; part[R_N + 2] = w * dpix[6];
                }
#undef RL
                if (lane == 0) STAT(7, 1);                              /* stage C heavy iterations */
                it0 += __popcll(bal);
                // bank-first transposing butterfly (wave_ops.h): lane l < 32 ends with the wave total of slot transposed_index(l);
                // the 24 slots are 24 consecutive dwords of one accumulator row -> one coalesced memory-side request
                const int slot = transposed_index(lane);
                const float tot = (ABL & 2) ? part[slot] : reduce32_bankfirst(part, lane);
                const uint32_t idj = (uint32_t)__builtin_amdgcn_readlane((int)id, jj);
                if (lane < 32 && slot < TEXGS_ACC_FLOATS && tot != 0.f) unsafeAtomicAdd(acc + (size_t)idj * TEXGS_ACC_FLOATS + slot, tot);
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
#if TC_ENABLE
    // flush the texel cache: thread -> dword, consecutive slots are consecutive texels of a row (coalesced atomics)
    __syncthreads();
    if (!(ABL & 1)) {
        for (int k = tid; k < TC_SLOTS * 3; k += TG_BLOCK) {
            const int slot = k / 3, ch = k - slot * 3;
            const uint32_t tag = s_ttag[slot];
            const float v = s_tval[k];
            if (tag != TC_EMPTY && v != 0.f) unsafeAtomicAdd(dtex + tag + ch, v);
        }
    }
#endif
}

// Sum the four phase-shifted quad arrays into dL_dtexture[6,R,R,3] (+=: the LDS-cache flush and the clamped border
// footprints are already there) and leave the quad arrays all-zero again, so a persistent scratch needs no memset.  Texel x receives slot px = (x - a) & 1 of quad (x - a) >> 1 of phase array a, a = 0,1.
__global__ void __launch_bounds__(TG_BLOCK)
k_texgrad_gather(int R, uint32_t dirty_index, float* __restrict__ quads, float* __restrict__ dtex) {
    const int QW = (R >> 1) + 1;
    // one workgroup = one 16x16-texel block of a face: every 64-byte quad line is consumed inside one workgroup
    const int bpr = (R + 15) >> 4;                              // blocks per row
    const int face = (int)(blockIdx.x / (uint32_t)(bpr * bpr));
    const int brem = (int)(blockIdx.x % (uint32_t)(bpr * bpr));
    const int x = ((brem % bpr) << 4) + (threadIdx.x & 15), y = ((brem / bpr) << 4) + (threadIdx.x >> 4);
    if (x >= R || y >= R) return;
    const size_t texel = ((size_t)face * R + y) * R + x;
    const uint32_t dirty = reinterpret_cast<const uint32_t*>(quads)[dirty_index];
    if (!((dirty >> face) & 1u)) return;                      // K7 never touched this face: its quads are still all-zero
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int dy = y - b;
        if (dy < 0) continue;
#pragma unroll
        for (int a_ = 0; a_ < 2; ++a_) {
            const int dx = x - a_;
            if (dx < 0) continue;
            const size_t q = ((((size_t)(a_ | (b << 1)) * 6 + face) * QW + (dy >> 1)) * QW + (dx >> 1)) * 16
                           + (size_t)(((dy & 1) * 2 + (dx & 1)) * 3);
            const float v0 = quads[q], v1 = quads[q + 1], v2 = quads[q + 2];
            if (v0 != 0.f || v1 != 0.f || v2 != 0.f) {      // read-and-clear: each slot belongs to exactly one texel thread
                s0 += v0; s1 += v1; s2 += v2;
                quads[q] = 0.f; quads[q + 1] = 0.f; quads[q + 2] = 0.f;
            }
        }
    }
    if (s0 != 0.f || s1 != 0.f || s2 != 0.f) {
        float* o = dtex + texel * 3;
        o[0] += s0; o[1] += s1; o[2] += s2;
    }
}

inline PixArgs make_pix(const CamConst& c, const TexGSFrame* f, const TexGSInputs* in, const TexGSGeom* g,
                        const TexGSBinning* b) {
    PixArgs a;
    a.W = c.W; a.H = c.H; a.tiles_x = c.tiles_x; a.num_tiles = c.tiles_x * c.tiles_y; a.R = c.R;
    a.ranges = reinterpret_cast<const uint2*>(b->ranges);
    a.point_list = b->point_list;
    a.tile_order = b->tile_order;
    a.quad_dirty_index = (uint32_t)(tex_quads_floats(c.R) - 16);
    a.rec = reinterpret_cast<const float4*>(g->rec);
    a.texture = in->texture;
    a.bg = f->bg;
    return a;
}

}  // namespace

void launch_render_fwd(const CamConst& c, const TexGSFrame* f, const TexGSInputs* in, const TexGSGeom* g,
                       const TexGSBinning* b, TexGSImage* img, hipStream_t s) {
    PixArgs a = make_pix(c, f, in, g, b);
#ifdef TEXGS_EXPERIMENTS     // timing experiments (ablations, heaviest-K-tiles runs); never in the product build
    static const int maxt_f = getenv("TEXGS_MAXTILES") ? atoi(getenv("TEXGS_MAXTILES")) : 0;
    if (maxt_f > 0 && maxt_f < a.num_tiles) a.num_tiles = maxt_f;
#endif
    const int grid = a.num_tiles;
#ifdef TEXGS_EXPERIMENTS
    static const int fabl = getenv("TEXGS_FWD_ABLATE") ? atoi(getenv("TEXGS_FWD_ABLATE")) : 0;
    if (fabl == 1) { hipLaunchKernelGGL(k_render_fwd<1>, dim3(grid), dim3(TG_BLOCK), 0, s, a, img->out_color, img->out_depth, img->out_norm, img->out_alpha, img->final_T, img->n_contrib); return; }
    if (fabl == 2) { hipLaunchKernelGGL(k_render_fwd<2>, dim3(grid), dim3(TG_BLOCK), 0, s, a, img->out_color, img->out_depth, img->out_norm, img->out_alpha, img->final_T, img->n_contrib); return; }
#endif
    hipLaunchKernelGGL(k_render_fwd<0>, dim3(grid), dim3(TG_BLOCK), 0, s, a, img->out_color, img->out_depth, img->out_norm,
                       img->out_alpha, img->final_T, img->n_contrib);
}

void launch_render_bwd(const CamConst& c, const TexGSFrame* f, const TexGSInputs* in, const TexGSGeom* g,
                       const TexGSBinning* b, const TexGSImage* img, TexGSGrads* gr, hipStream_t s) {
    PixArgs a = make_pix(c, f, in, g, b);
#ifdef TEXGS_EXPERIMENTS
    static const int maxt_b = getenv("TEXGS_MAXTILES") ? atoi(getenv("TEXGS_MAXTILES")) : 0;
    if (maxt_b > 0 && maxt_b < a.num_tiles) a.num_tiles = maxt_b;
#endif
    const int grid = a.num_tiles;
#define LAUNCH_BWD(A) hipLaunchKernelGGL(k_render_bwd<A>, dim3(grid), dim3(TG_BLOCK), 0, s, a, img->final_T, img->n_contrib, \
                       gr->dL_dcolor, gr->dL_ddepth, gr->dL_dnorm, gr->dL_dalpha, gr->acc, gr->dL_dtexture, gr->tex_quads)
#ifdef TEXGS_EXPERIMENTS
    static const int abl = getenv("TEXGS_ABLATE") ? atoi(getenv("TEXGS_ABLATE")) : 0;
    switch (abl) {
        case 1: LAUNCH_BWD(1); break;
        case 2: LAUNCH_BWD(2); break;
        case 3: LAUNCH_BWD(3); break;
        case 4: LAUNCH_BWD(4); break;
        case 9: LAUNCH_BWD(9); break;
        case 11: LAUNCH_BWD(11); break;
        default: LAUNCH_BWD(0); break;
    }
#else
    LAUNCH_BWD(0);
#endif
#undef LAUNCH_BWD
}

#ifdef TEXGS_STATS
extern "C" int texgs_debug_stats(unsigned long long* host16, int reset) {
    if (hipMemcpyFromSymbol(host16, HIP_SYMBOL(g_stats), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stats), z, sizeof(z)); }
    return 0;
}
#endif

size_t tex_quads_floats(int R) {
    const size_t QW = (size_t)(R >> 1) + 1;
    return (size_t)4 * 6 * QW * QW * 16 + 16;      // + one 64-byte slot holding the dirty-face word
}

void launch_texgrad_gather(const CamConst& c, TexGSGrads* gr, hipStream_t s) {
    const int bpr = (c.R + 15) >> 4;
    const int blocks = 6 * bpr * bpr;
    hipLaunchKernelGGL(k_texgrad_gather, dim3(blocks), dim3(TG_BLOCK), 0, s, c.R, (uint32_t)(tex_quads_floats(c.R) - 16),
                       gr->tex_quads, gr->dL_dtexture);
    (void)hipMemsetAsync(gr->tex_quads + (tex_quads_floats(c.R) - 16), 0, 64, s);      // leave the scratch all-zero
}
