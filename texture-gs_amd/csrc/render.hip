// K6 render forward, K7 render backward, and the texture-gradient bin reduce -- gfx950 (CDNA4), wave64.
//
// One wave = one 8x8 pixel block of a 16x16 tile; the waves of a tile share nothing (no block barrier anywhere), so the
// workgroup size is a pure scheduling choice (TG_WAVES_PER_WG).  Per chunk of 64 depth-sorted instances lane l keeps
// instance l's record in registers: the per-pixel *sequential* loops (alpha test, transmittance) get the tested Gaussian by
// v_readlane broadcast; contributing (pixel, Gaussian) pairs are compacted (ballot + mbcnt) into a per-wave LDS list and
// the *dense* texture work (UV Taylor step, cubemap address, 4 taps, colour / gradients) runs 64 pairs at a time with
// every lane busy, reading the pair's Gaussian fields from a per-wave LDS copy of the chunk's records.
//
// Texture gradient (K7): every fp32 global atomic on this part executes memory-side at ~20 G requests/s and the ~19 M
// bilinear footprints of a C3 view cost 0.69 ms that way (profiles/r02_ablation.md).  Instead K7 APPENDS a 24-byte
// record {cell, fx, fy, dL/dtexel-colour (3)} per footprint with plain coalesced stores to the list of the 32x32-texel
// texture block ("bin") the footprint is anchored in -- one returning atomic per (wave round, distinct bin) on the
// bin's cursor -- and k_texgrad_reduce then sums each bin's list in LDS and adds every texel of the block to
// dL_dtexture once.  No MFMA: there is no dense contraction on this path.
#include "common.h"
#include "wave_ops.h"
#include <stdlib.h>

namespace {

#ifndef TG_WAVES_PER_WG
#define TG_WAVES_PER_WG 1          // 1: one workgroup per 8x8 block (finest dispatch; K7 keeps 11 waves/CU by LDS); 4: one per tile
#endif
#define TG_WG_THREADS (64 * TG_WAVES_PER_WG)

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
    int   axis, face;
};

__device__ __forceinline__ CubeTap cube_address(float u0, float u1, float u2, int R) {
    CubeTap t;
    const float a0 = fabsf(u0), a1 = fabsf(u1), a2 = fabsf(u2);
    float m, ua, ub;
    if (a0 >= a1 && a0 >= a2) { t.axis = 0; m = u0; t.sm = (u0 >= 0.f) ? 1.f : -1.f; ua = u2; t.su = -t.sm; ub = u1; t.sv = -1.f; }
    else if (a1 >= a2)        { t.axis = 1; m = u1; t.sm = (u1 >= 0.f) ? 1.f : -1.f; ua = u0; t.su = 1.f;   ub = u2; t.sv = t.sm; }
    else                      { t.axis = 2; m = u2; t.sm = (u2 >= 0.f) ? 1.f : -1.f; ua = u0; t.su = t.sm;  ub = u1; t.sv = -1.f; }
    t.face = 2 * t.axis + (t.sm > 0.f ? 0 : 1);
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
    const int fb = t.face * R;
    t.o00 = ((fb + y0c) * R + x0c) * 3; t.o01 = ((fb + y0c) * R + x1c) * 3;
    t.o10 = ((fb + y1c) * R + x0c) * 3; t.o11 = ((fb + y1c) * R + x1c) * 3;
    return t;
}

// The falloff exponent, shared by K6 and K7: K7 must reproduce K6's contributor decisions (power <= 0, alpha >= 1/255)
// bit for bit, so both evaluate this one explicitly ordered sequence of fp32 operations.  K1 stores the conic pre-scaled,
// (ah, bh, ch) = (-a/2, -b, -c/2), so power = ah dx^2 + bh dx dy + ch dy^2 is 3 mul + 1 mul + 2 fma.
__device__ __forceinline__ float gauss_power(float ah, float bh, float ch, float dx, float dy) {
    return __fmaf_rn(bh, __fmul_rn(dx, dy), __fmaf_rn(ch, __fmul_rn(dy, dy), __fmul_rn(ah, __fmul_rn(dx, dx))));
}
__device__ __forceinline__ float gauss_alpha_raw(float op, float power) { return op * __expf(power); }

// Per-wave cull, lane-parallel (lane = one instance of the chunk): can the instance reach alpha >= 1/255 at ANY point of the
// wave's 8x8 pixel block?  power(d) = ah dx^2 + bh dx dy + ch dy^2 is concave with its maximum 0 at the splat centre, so its
// maximum over the block's rectangle is 0 if the centre is inside, else it sits on one of the four edges, where it is a 1-D
// concave parabola maximised at the clamped stationary point.  Exact for the continuous rectangle, hence conservative for
// the pixel centres; `thr` already carries a margin far above fp32 rounding.  (The bounding-disc test alone let through
// twice as many instances as ever produced an item: edge-on splats are needles, not discs.)
__device__ __forceinline__ bool block_reachable(float gx, float gy, float ah, float bh, float ch, float thr, float rcull,
                                                float x0, float y0) {
    const float x1 = x0 + 7.0f, y1 = y0 + 7.0f;
    if (!(rcull >= 0.f && (gx + rcull >= x0) && (gx - rcull <= x1) && (gy + rcull >= y0) && (gy - rcull <= y1))) return false;
    const float dx0 = gx - x1, dx1 = gx - x0, dy0 = gy - y1, dy1 = gy - y0;     // d = centre - pixel ranges over [dx0,dx1] x [dy0,dy1]
    if (dx0 <= 0.f && dx1 >= 0.f && dy0 <= 0.f && dy1 >= 0.f) return true;     // centre inside the block
    const float ihc = -0.5f * __builtin_amdgcn_rcpf(ch), iha = -0.5f * __builtin_amdgcn_rcpf(ah);
    auto edge_x = [&](float dx) {       // dx fixed, dy free in [dy0, dy1]
        const float dy = fminf(dy1, fmaxf(dy0, bh * dx * ihc));
        return ah * dx * dx + bh * dx * dy + ch * dy * dy;
    };
    auto edge_y = [&](float dy) {
        const float dx = fminf(dx1, fmaxf(dx0, bh * dy * iha));
        return ah * dx * dx + bh * dx * dy + ch * dy * dy;
    };
    const float best = fmaxf(fmaxf(edge_x(dx0), edge_x(dx1)), fmaxf(edge_y(dy0), edge_y(dy1)));
    return best >= thr - 1e-3f * fabsf(thr) - 1e-4f;
}

struct PixArgs {
    int W, H, tiles_x, num_tiles, R;
    const uint2* ranges;
    const uint32_t* tile_order;
    const uint32_t* point_list;
    const float4* rec;
    const float* texture;
    const float* bg;
};

// workgroup -> (tile, 8x8 block).  Tiles are launched longest-list-first (tile_order).  With one wave per workgroup the
// four blocks of a tile get ids that are equal mod 8, so they run on the same XCD (workgroup b is observed on XCD b % 8:
// speed only) and share its L2 for the tile's records.
__device__ __forceinline__ bool wave_block(const PixArgs& a, int& tile, int& wave) {
#if TG_WAVES_PER_WG == 4
    if ((int)blockIdx.x >= a.num_tiles) return false;
    const int rank = (int)blockIdx.x;
    tile = (int)a.tile_order[blockIdx.x];
    wave = (int)(threadIdx.x >> 6);
#else
    const int b = (int)blockIdx.x, k = b >> 3;
    wave = k & 3;
    const int rank = ((k >> 2) << 3) | (b & 7);
    if (rank >= a.num_tiles) return false;
    tile = (int)a.tile_order[rank];
#endif
    (void)rank;
    return true;
}
inline int blend_grid(int num_tiles) {
#if TG_WAVES_PER_WG == 4
    return num_tiles;
#else
    return 4 * ((num_tiles + 7) & ~7);
#endif
}

#define RLF(V, J) __int_as_float(__builtin_amdgcn_readlane(__float_as_int(V), (J)))
// HIP's __ballot(int) materialises the predicate as 0/1 in a VGPR and compares it again (v_cndmask + v_cmp per ballot);
// the builtin takes the lane mask the compares already produced.
#define TG_BALLOT(P) __builtin_amdgcn_ballot_w64((bool)(P))

// ------------------------------------------------------------------------------------------------ K6
// Forward blend.  Per chunk of 64 instances:
//   (sequential) every lane walks the chunk for its own pixel; the tested Gaussian's (xy, conic, opacity) arrive by
//       v_readlane broadcast -- no LDS traffic or LDS latency in the dependent chain; ~30 VALU per test.  Depth,
//       normal and alpha accumulate here (w = alpha*T needs no texture).
//   (dense)      contributing (pixel, j, w) triples are compacted with ballot + mbcnt into a 128-entry per-wave LDS
//       ring; whenever 64 are queued all 64 lanes pop one each, read the item's Gaussian fields (xy, g, G, phi, viewdep:
//       four float4) from the per-wave LDS copy of the chunk's records, do the UV Taylor step, cubemap addressing and
//       4 dwordx3 tap loads with full lane occupancy and 64 fetches in flight, then add w*colour into the pixel's LDS
//       accumulator (fixed point, integer atomics).  The colour sum is order-independent, so this equals the in-order blend.
// (Only ~8 of 64 pixels of a wave contribute to a given Gaussian: with the texture path inside the sequential loop it
//  ran at ~12 % lane efficiency.  Fetching the item's 16 fields with ds_bpermute from lane j's registers made 42 % of
//  the wave time an LDS-issue stall: profiles/r01_v7_summary.txt.)
#define FQ_CAP 128
#ifndef FWD_WAVES_PER_SIMD
#define FWD_WAVES_PER_SIMD 5       // 81 VGPRs.  (6 = 80 VGPRs was measured: same solo time, slower next to another view's K7)
#endif

__global__ void __launch_bounds__(TG_WG_THREADS, FWD_WAVES_PER_SIMD)
k_render_fwd(PixArgs a, float* __restrict__ out_color, float* __restrict__ out_depth, float* __restrict__ out_norm,
             float* __restrict__ out_alpha, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib) {
    __shared__ uint2 s_qall[TG_WAVES_PER_WG][FQ_CAP];
    __shared__ unsigned long long s_colall[TG_WAVES_PER_WG][64 * 3];   // Q32.32 colour sums (see drain)
    __shared__ float4 s_recall[TG_WAVES_PER_WG][4 * 64];     // (xy, g) | G0-3 | G4-5, phi0-1 | phi2, viewdep
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int tile, wave;
    if (!wave_block(a, tile, wave)) return;
    const int tile_x = tile % a.tiles_x, tile_y = tile / a.tiles_x;
    const int wave_px = tile_x * TEXGS_TILE + ((wave & 1) << 3), wave_py = tile_y * TEXGS_TILE + ((wave >> 1) << 3);
    const int px = wave_px + (lane & 7), py = wave_py + (lane >> 3);
    const bool inside = (px < a.W) && (py < a.H);
    const float pxf = (float)px, pyf = (float)py;
    const uint2 range = a.ranges[tile];
    const int todo = (int)(range.y - range.x);
    const float* __restrict__ tex = a.texture;
    uint2* s_q = s_qall[wv];
    unsigned long long* s_c = s_colall[wv];
    float4* s_rec = s_recall[wv];

    s_c[lane * 3 + 0] = 0ull; s_c[lane * 3 + 1] = 0ull; s_c[lane * 3 + 2] = 0ull;   // own pixel; only this wave touches it
    __builtin_amdgcn_wave_barrier();

    bool done = !inside;
    unsigned long long done_mask = TG_BALLOT(!inside);        // the same, as a wave-level lane mask (scalar registers)
    float T = 1.0f;
    float Dp = 0.f, N0 = 0.f, N1 = 0.f, N2 = 0.f, Al = 0.f;
    uint32_t last = 0;
    int qhead = 0, qtail = 0;                                  // wave-uniform

    // The dense phase is software-pipelined by one batch: drain(n) first FINISHES the previous batch (its 4 taps were
    // loaded a whole batch interval ago: colour, Q32.32 accumulate), then STARTS the new one (queue pop, record fields, UV
    // Taylor step, cubemap address, tap loads issued) and returns without waiting for them.
    int pn = 0;                                    // lanes of the batch in flight (wave-uniform)
    int p_pl = 0;
    float p_w = 0.f, p_fx = 0.f, p_fy = 0.f, p_vd0 = 0.f, p_vd1 = 0.f, p_vd2 = 0.f;
    Texel3 p00 = {0.f, 0.f, 0.f}, p01 = p00, p10 = p00, p11 = p00;
    auto finish = [&]() {
        if (lane < pn) {
            const float w00 = (1.f - p_fx) * (1.f - p_fy), w01 = p_fx * (1.f - p_fy);
            const float w10 = (1.f - p_fx) * p_fy,         w11 = p_fx * p_fy;
            const float t0 = w00 * p00.x + w01 * p01.x + w10 * p10.x + w11 * p11.x;
            const float t1 = w00 * p00.y + w01 * p01.y + w10 * p10.y + w11 * p11.y;
            const float t2 = w00 * p00.z + w01 * p01.z + w10 * p10.z + w11 * p11.z;
            // w * colour (>= 0) goes to the owning pixel's accumulator as Q32.32 fixed point with an INTEGER LDS atomic:
            // ds_add_f32 retires ~3 cycles per LANE on gfx950 (193 cycles per wave instruction, scripts/ubench/lds_atomics.hip),
            // ds_add_u64 6 cycles per instruction.  2^-32 resolution (45 items: < 1e-8), exact and order-independent below 2^31.
            unsigned long long* cp = s_c + p_pl * 3;
            atomicAdd(cp + 0, (unsigned long long)(fminf(p_w * fmaxf(0.f, TG_SH_C0 * t0 + p_vd0 + 0.5f), 2.0e9f) * 4294967296.0f));
            atomicAdd(cp + 1, (unsigned long long)(fminf(p_w * fmaxf(0.f, TG_SH_C0 * t1 + p_vd1 + 0.5f), 2.0e9f) * 4294967296.0f));
            atomicAdd(cp + 2, (unsigned long long)(fminf(p_w * fmaxf(0.f, TG_SH_C0 * t2 + p_vd2 + 0.5f), 2.0e9f) * 4294967296.0f));
        }
        pn = 0;
    };
    auto drain = [&](int n_) {
        finish();
        uint2 e_ = make_uint2(0u, 0u);
        if (lane < n_) e_ = s_q[(qhead + lane) & (FQ_CAP - 1)];
        const int pl_ = (int)(e_.y >> 8) & 63, jj_ = (int)(e_.y & 63u);
        const float4 p0 = s_rec[jj_], p1 = s_rec[64 + jj_], p2 = s_rec[128 + jj_], p3 = s_rec[192 + jj_];
        const float dpx = (float)(wave_px + (pl_ & 7)) - p0.x, dpy = (float)(wave_py + (pl_ >> 3)) - p0.y;
        const float den = 1.0f + p0.z * dpx + p0.w * dpy;
        const float inv = (den >= TG_DEN_MIN) ? __builtin_amdgcn_rcpf(den) : 0.0f;
        const float u0 = p2.z + (p1.x * dpx + p1.y * dpy) * inv;
        const float u1 = p2.w + (p1.z * dpx + p1.w * dpy) * inv;
        const float u2 = p3.x + (p2.x * dpx + p2.y * dpy) * inv;
        const CubeTap ct = cube_address(u0, u1, u2, a.R);
        if (lane < n_) {
            p00 = load_texel(tex, ct.o00); p01 = load_texel(tex, ct.o01);
            p10 = load_texel(tex, ct.o10); p11 = load_texel(tex, ct.o11);
        }
        p_w = __uint_as_float(e_.x); p_pl = pl_; p_fx = ct.fx; p_fy = ct.fy; p_vd0 = p3.y; p_vd1 = p3.z; p_vd2 = p3.w;
        pn = n_;
    };

    // (Double-buffering the records in registers -- chunk c+1 in flight while chunk c is blended -- was measured: no gain,
    // +28 VGPRs.  The kernel is bound by VALU issue at its occupancy, not by the index -> record load latency.)
    for (int base = 0; base < todo; base += 64) {
        if (~done_mask == 0ull) break;
        const int cnt = min(64, todo - base);
        float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0, r3 = r0, r4 = r0, r5 = r0, r6 = make_float4(-1.f, 1.f, 0.f, 0.f);
        if (lane < cnt) {
            const uint32_t id = a.point_list[range.x + base + lane];
            const float4* __restrict__ r = a.rec + (size_t)id * (TEXGS_REC_FLOATS / 4);
            r0 = r[0]; r1 = r[1]; r2 = r[2]; r3 = r[3]; r4 = r[4]; r5 = r[5]; r6 = r[6];
        }
        // dense-phase copy of the chunk (the previous chunk's items were all drained before this point)
        s_rec[lane] = make_float4(r0.x, r0.y, r1.z, r1.w); s_rec[64 + lane] = r2; s_rec[128 + lane] = r3; s_rec[192 + lane] = r4;
        __builtin_amdgcn_wave_barrier();
        // per-wave cull, lane-parallel: can instance `lane` reach alpha >= 1/255 anywhere in this wave's 8x8 block?
        unsigned long long todo_mask = TG_BALLOT(block_reachable(r0.x, r0.y, r0.z, r0.w, r1.x, r6.y, r6.x, (float)wave_px, (float)wave_py));
        // One exit per tested instance: alpha is evaluated for every candidate that survived the block cull (84 % of them blend
        // somewhere in the block; a power-threshold prefilter ahead of the alpha test was slower).  Wave-level decisions are
        // PRODUCTS of single-compare ballots: a ballot of one compare is the v_cmp's own lane mask and the combination is scalar
        // ALU; a ballot of a compound predicate costs a v_cndmask + v_cmp round trip.
        auto test = [&](int j, float& power, float& alpha, float4& geo) {
            const float gx_ = RLF(r0.x, j), gy_ = RLF(r0.y, j), ca = RLF(r0.z, j), cb = RLF(r0.w, j);
            const float cc = RLF(r1.x, j), op = RLF(r1.y, j);
            geo = make_float4(RLF(r5.x, j), RLF(r5.y, j), RLF(r5.z, j), RLF(r5.w, j));       // depth, normal
            power = gauss_power(ca, cb, cc, gx_ - pxf, gy_ - pyf);
            alpha = fminf(TG_ALPHA_MAX, gauss_alpha_raw(op, power));
        };
        auto blend = [&](int j, float power, float alpha, const float4& geo) -> bool {          // true: every pixel is done
            const unsigned long long m_ok = TG_BALLOT(power <= 0.0f) & ~done_mask & TG_BALLOT(alpha >= TG_ALPHA_MIN);
            if (m_ok == 0ull) return false;
            bool ok = (!done) && (power <= 0.0f) && (alpha >= TG_ALPHA_MIN);
            const float Tn = T * (1.0f - alpha);
            const unsigned long long m_kill = m_ok & TG_BALLOT(Tn < TG_T_EPS);
            if (ok && Tn < TG_T_EPS) { done = true; ok = false; }
            done_mask |= m_kill;
            const unsigned long long bal = m_ok & ~m_kill;
            if (bal != 0ull) {
                if (ok) {
                    const float w = alpha * T;
                    Dp += w * geo.x; N0 += w * geo.y; N1 += w * geo.z; N2 += w * geo.w; Al += w;
                    T = Tn;
                    last = (uint32_t)(base + j + 1);
                    const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32),
                                          __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                    s_q[(qtail + rank) & (FQ_CAP - 1)] = make_uint2(__float_as_uint(w), ((uint32_t)lane << 8) | (uint32_t)j);
                }
                qtail += __popcll(bal);
                if (qtail - qhead >= 64) {
                    __builtin_amdgcn_wave_barrier();
                    drain(64);
                    qhead += 64;
                }
            }
            return ~done_mask == 0ull;
        };
        while (todo_mask != 0ull) {
            const int j = __ffsll((long long)todo_mask) - 1;
            todo_mask &= todo_mask - 1ull;
            float p, al;
            float4 g;
            test(j, p, al, g);
            if (blend(j, p, al, g)) break;
        }
        // items reference this chunk's LDS copy: finish them before the next chunk is loaded
        if (qtail - qhead > 0) {
            __builtin_amdgcn_wave_barrier();
            drain(qtail - qhead);
            qhead = qtail;
        }
        __builtin_amdgcn_wave_barrier();
    }
    finish();
    __builtin_amdgcn_wave_barrier();
    if (inside) {
        const int HW = a.W * a.H, pix = py * a.W + px;
        const double q = 1.0 / 4294967296.0;
        out_color[pix] = (float)((double)s_c[lane * 3 + 0] * q) + T * a.bg[0];
        out_color[HW + pix] = (float)((double)s_c[lane * 3 + 1] * q) + T * a.bg[1];
        out_color[2 * HW + pix] = (float)((double)s_c[lane * 3 + 2] * q) + T * a.bg[2];
        out_depth[pix] = Dp;
        out_norm[pix] = N0; out_norm[HW + pix] = N1; out_norm[2 * HW + pix] = N2;
        out_alpha[pix] = Al;
        final_T[pix] = T;
        n_contrib[pix] = last;
    }
}

// ------------------------------------------------------------------------------------------------ K7
// Backward replay, per chunk of 64 instances, back to front:
//   stage A  sequential, ~30 VALU / test: falloff, alpha, T /= (1-alpha); contributing (pixel, j) pairs are
//            compacted (ballot + mbcnt) into an LDS item list {T, alpha_raw, q, key}; per-j ballots stay in VGPRs.
//   stage B  dense, 64 items per round: UV Taylor step, cubemap address, 4 dwordx3 tap loads, colour; stores per item
//            s = colour . dL/dcolour + geometry channels . their gradients (what stage C1 sums) and dL/dcolour (3),
//            dL/duv (3), 1/den, dL/dden; appends the item's texture-gradient record to its texture bin (see the file header).
//   stage C1 sequential, per pixel: dL/dalpha_i = T_i s_i - (sum of s_k alpha_k T_k behind i + bg term) / (1 - alpha_i), one
//            running sum per pixel; leaves w and dL/dpower in the item.
//   stage C2 dense over TASKS (<= 16 consecutive items of one Gaussian, 4 tasks per round): the 28 per-Gaussian moment
//            terms of every item, a 16-lane transposing butterfly (DPP only, wave_ops.h), and the 16 lanes add the
//            Gaussian's 128-byte accumulator row as two 64-byte runs.
#ifndef BQ_CAP
#define BQ_CAP 128
#endif
#ifndef BWD_WAVES_PER_SIMD
#define BWD_WAVES_PER_SIMD 2
#endif

#ifdef K7_STATS        // diagnostics build (scripts/k7_stats.py): dynamic work counters of K7, summed over the launch
__device__ unsigned long long g_k7_stats[16];
#define K7_COUNT(i, n) (k7s[i] += (unsigned)(n))
#else
#define K7_COUNT(i, n) ((void)0)
#endif
#ifndef K7_ABL
#define K7_ABL 0      // timing-only ablations (scripts/bench_variants.sh): 1 = no stage C1, 2 = no stage C2, 4 = no stage B
#endif
struct TexBinArgs {
    float*    rec;         // [nbins][6][cap]: cell | fx | fy | dL/dtexel-colour r, g, b   (plane-major inside a bin)
    uint32_t* cursor;      // [nbins] records appended so far (may exceed cap: the excess went to dL_dtexture directly)
    uint32_t* stats;       // [0] max list length (for the host), [1] bits of max |dL/dpixel colour| of this call
    uint32_t  cap;
    int       nb;          // bins per face row = ceil(R / 32)
};

// footprints that cannot be binned (clamped at a face border, or the bin is full): straight into dL_dtexture
__device__ __forceinline__ void scatter_direct(float* __restrict__ dtex, const CubeTap& ct, float w00, float w01, float w10,
                                               float w11, float x0, float x1, float x2) {
    unsafeAtomicAdd(dtex + ct.o00, w00 * x0); unsafeAtomicAdd(dtex + ct.o00 + 1, w00 * x1); unsafeAtomicAdd(dtex + ct.o00 + 2, w00 * x2);
    unsafeAtomicAdd(dtex + ct.o01, w01 * x0); unsafeAtomicAdd(dtex + ct.o01 + 1, w01 * x1); unsafeAtomicAdd(dtex + ct.o01 + 2, w01 * x2);
    unsafeAtomicAdd(dtex + ct.o10, w10 * x0); unsafeAtomicAdd(dtex + ct.o10 + 1, w10 * x1); unsafeAtomicAdd(dtex + ct.o10 + 2, w10 * x2);
    unsafeAtomicAdd(dtex + ct.o11, w11 * x0); unsafeAtomicAdd(dtex + ct.o11 + 1, w11 * x1); unsafeAtomicAdd(dtex + ct.o11 + 2, w11 * x2);
}

__global__ void __launch_bounds__(TG_WG_THREADS, BWD_WAVES_PER_SIMD)
k_render_bwd(PixArgs a, TexBinArgs tb, const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
             const float* __restrict__ dL_dcolor, const float* __restrict__ dL_ddepth,
             const float* __restrict__ dL_dnorm, const float* __restrict__ dL_dalpha,
             float* __restrict__ acc, float* __restrict__ dtex) {
    // 3 float4 per item: {T -> w, s -> dL/dpower, alpha_raw, key} {dc, du0} {du1, du2, inv, dden}; + one all-zero item (stage C2's idle lanes)
    __shared__ float4 s_items_all[TG_WAVES_PER_WG][BQ_CAP * 3 + 3];
    __shared__ float4 s_dpix_all[TG_WAVES_PER_WG][64];            // dL/d(r, g, b, alpha) of the wave's pixels
    __shared__ float4 s_recs_all[TG_WAVES_PER_WG][6 * 64];        // the chunk's records, plane-major [k][lane]
    __shared__ float s_dgeo_all[TG_WAVES_PER_WG][64 * 4];         // dL/d(depth, normal) of the wave's pixels
    __shared__ uint32_t s_task_all[TG_WAVES_PER_WG][BQ_CAP / 16 + 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int tile, wave;
    if (!wave_block(a, tile, wave)) return;
    const int tile_x = tile % a.tiles_x, tile_y = tile / a.tiles_x;
    const int wave_px = tile_x * TEXGS_TILE + ((wave & 1) << 3), wave_py = tile_y * TEXGS_TILE + ((wave >> 1) << 3);
    const int px = wave_px + (lane & 7), py = wave_py + (lane >> 3);
    const bool inside = (px < a.W) && (py < a.H);
    const float pxf = (float)px, pyf = (float)py;
    const uint2 range = a.ranges[tile];
    const int todo = (int)(range.y - range.x);
    const int HW = a.W * a.H, pix = py * a.W + px;
    const float* __restrict__ tex = a.texture;
    float4* s_items = s_items_all[wv];
    float4* s_recs = s_recs_all[wv];
    float4* s_dpix = s_dpix_all[wv];
    float* s_dgeo = s_dgeo_all[wv];
    uint32_t* s_task = s_task_all[wv];

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
    if (lane < 3) s_items[BQ_CAP * 3 + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
    s_dpix[lane] = make_float4(dpix[0], dpix[1], dpix[2], dpix[7]);
    *reinterpret_cast<float4*>(&s_dgeo[lane * 4]) = make_float4(dpix[3], dpix[4], dpix[5], dpix[6]);
    if (tb.rec != nullptr) {
        // image-wide bound on the texture-gradient records (|x| <= C0 |dL/dpixel|): the reduce kernel's fixed-point scale.
        // Positive floats order like their bit patterns; the plain read first keeps 10^4 waves off one hot word.
        const int mbits = wave_max_i(__float_as_int(fmaxf(fabsf(dpix[0]), fmaxf(fabsf(dpix[1]), fabsf(dpix[2])))));
        if (lane == 0 && (uint32_t)mbits > __hip_atomic_load(tb.stats + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(tb.stats + 1, (uint32_t)mbits);
    }
    const int wave_last = min(wave_max_i(last), todo);
    __builtin_amdgcn_wave_barrier();

    float T = Tfin;
    float behind = Tfin * bgdot;      // sum of s_k alpha_k T_k over the contributors BEHIND the current one, + the background term
    uint32_t my_it0_sink = 0u;
#ifdef K7_STATS
    unsigned k7s[16] = {0};
#endif

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
        // results collide with gfx9's one-SGPR-per-VALU constant-bus limit; stage B: per-lane gather)
        __builtin_amdgcn_wave_barrier();
        s_recs[0 * 64 + lane] = r0; s_recs[1 * 64 + lane] = r1; s_recs[2 * 64 + lane] = r2v;
        s_recs[3 * 64 + lane] = r3v; s_recs[4 * 64 + lane] = r4v; s_recs[5 * 64 + lane] = r5;
        __builtin_amdgcn_wave_barrier();
        // per-wave cull (see K6): instances that cannot reach alpha >= 1/255 inside this wave's 8x8 block are never visited
        const unsigned long long cull_mask = TG_BALLOT(block_reachable(r0.x, r0.y, r0.z, r0.w, r1.x, r6.y, r6.x, (float)wave_px, (float)wave_py));
        K7_COUNT(0, 1); K7_COUNT(1, jtop); K7_COUNT(2, __popcll(cull_mask));      // chunks, instances, instances after the cull
        uint32_t touched_lo = 0u, touched_hi = 0u;           // lane j keeps the stage-A ballot of instance j
        unsigned long long amask = cull_mask;                 // instances still to be tested (stage A), high to low
        while (amask != 0ull) {
            // ================================================================ stage A
            int n_items = 0;
            unsigned long long seg_mask = 0ull;                 // instances of this segment that produced items
            while (amask != 0ull) {
                const int j = 63 - __clzll((long long)amask);
                const unsigned long long jbit = 1ull << j;
                K7_COUNT(3, 1);                                   // stage-A iterations
                // one exit: alpha for every candidate that survived the block cull (84 % of them produce items anyway; a
                // power-threshold prefilter before the alpha test was slower), all six broadcasts up front
                const float gx_ = RLF(r0.x, j), gy_ = RLF(r0.y, j), ca = RLF(r0.z, j), cb = RLF(r0.w, j);
                const float cc = RLF(r1.x, j), op = RLF(r1.y, j);
                const float power = gauss_power(ca, cb, cc, gx_ - pxf, gy_ - pyf);
                const float araw = gauss_alpha_raw(op, power);
                const float alpha = fminf(TG_ALPHA_MAX, araw);
                const bool ok = (base + j < last) && (power <= 0.0f) && (alpha >= TG_ALPHA_MIN);
                const unsigned long long bal = TG_BALLOT(power <= 0.0f) & TG_BALLOT(base + j < last) & TG_BALLOT(alpha >= TG_ALPHA_MIN);
                const int nb = __popcll(bal);
                if (nb == 0) { amask &= ~jbit; continue; }
                if (n_items + nb > BQ_CAP) break;                   // segment full; j is re-tested in the next one
                amask &= ~jbit;
                seg_mask |= jbit;
                K7_COUNT(4, 1); K7_COUNT(5, nb); K7_COUNT(6, (nb + 15) >> 4);      // instances with items, items, C2 tasks
                if (lane == j) { touched_lo = (uint32_t)bal; touched_hi = (uint32_t)(bal >> 32); }
                if (ok) {
                    T = T * __builtin_amdgcn_rcpf(1.0f - alpha);     // v_rcp_f32 (1 ulp): an IEEE divide is ~10 VALU
                    const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32),
                                          __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                    s_items[(n_items + rank) * 3] = make_float4(T, 0.f, araw, __uint_as_float(((uint32_t)lane << 8) | (uint32_t)j));
                }
                n_items += nb;
            }
            __builtin_amdgcn_wave_barrier();
            // ================================================================ stage B
            // A segment holds at most two rounds of 64 items.  Both rounds' FRONT halves run first (addresses, the 4 tap
            // loads, bin grouping, the cursor atomic: 8 loads + 2 returning atomics in flight), then both BACK halves
            // (colour / gradient math, record stores).  K7's time falls as a + b / (waves per CU) with a large b: its waves
            // mostly wait on memory, so the loads of round 1 are issued before anything waits on those of round 0.
            struct Round {                                        // what the back half needs, as few registers as possible
                bool have, binned;
                int e, pl, jj, my_leader, my_rank, axis;
                uint32_t bin, cell, slot0;
                int o00, dox, doy;                                // tap offsets: o01 = o00 + dox, o10 = o00 + doy, o11 = o00 + dox + doy
                float w, vd0, vd1, vd2, nu0, nu1, nu2, inv;
                float fx, fy, ka, kb, kc, kd;                     // d(col,row)/d(ua,ub,m) factors of the cube projection
                Texel3 t00, t01, t10, t11;
            };
            auto front = [&](int r, Round& R) {
                R.e = r + lane;
                R.have = R.e < n_items;
                float4 it = make_float4(1.f, 0.f, 0.f, 0.f);
                if (R.have) it = s_items[R.e * 3];
                const uint32_t key = __float_as_uint(it.w);
                R.pl = (int)(key >> 8) & 63;
                const int jj = (int)(key & 63u);
                R.jj = jj;
                const float4 q0 = s_recs[0 * 64 + jj], q1 = s_recs[1 * 64 + jj], r2 = s_recs[2 * 64 + jj],
                             r3 = s_recs[3 * 64 + jj], r4 = s_recs[4 * 64 + jj];
                R.w = fminf(TG_ALPHA_MAX, it.z) * it.x;
                R.vd0 = r4.y; R.vd1 = r4.z; R.vd2 = r4.w;
                // UV Taylor step, cubemap address, tap loads
                const float dpx = (float)(wave_px + (R.pl & 7)) - q0.x, dpy = (float)(wave_py + (R.pl >> 3)) - q0.y;
                const float den = 1.0f + q1.z * dpx + q1.w * dpy;
                R.inv = (den >= TG_DEN_MIN) ? __builtin_amdgcn_rcpf(den) : 0.0f;
                R.nu0 = r2.x * dpx + r2.y * dpy; R.nu1 = r2.z * dpx + r2.w * dpy; R.nu2 = r3.x * dpx + r3.y * dpy;
                const CubeTap ct = cube_address(r3.z + R.nu0 * R.inv, r3.w + R.nu1 * R.inv, r4.x + R.nu2 * R.inv, a.R);
                const Texel3 tz = {0.f, 0.f, 0.f};
                R.t00 = tz; R.t01 = tz; R.t10 = tz; R.t11 = tz;
                if (R.have) {
                    R.t00 = load_texel(tex, ct.o00); R.t01 = load_texel(tex, ct.o01);
                    R.t10 = load_texel(tex, ct.o10); R.t11 = load_texel(tex, ct.o11);
                }
                R.axis = ct.axis; R.fx = ct.fx; R.fy = ct.fy;
                R.ka = ct.su * ct.h; R.kb = ct.sv * ct.h;
                const float km = ct.h * ct.rma * ct.sm;
                R.kc = ct.sc * km; R.kd = ct.tc * km;
                R.o00 = ct.o00; R.dox = ct.o01 - ct.o00; R.doy = ct.o10 - ct.o00;
                R.cell = (uint32_t)(((ct.y0 & 31) << 8) | (ct.x0 & 31));
                // slot in the texture bin's record list: one returning atomic per distinct bin of the round.  Group the
                // lanes by bin (ballots only), then every group leader bumps its bin's cursor in ONE instruction.
                R.binned = R.have && tb.rec != nullptr && ct.x1 == ct.x0 + 1 && ct.y1 == ct.y0 + 1;   // not clamped at a face border
                R.bin = (uint32_t)((ct.face * tb.nb + (ct.y0 >> 5)) * tb.nb + (ct.x0 >> 5));
                bool leader = false;
                R.my_leader = lane; R.my_rank = 0;
                int my_n = 0;
                unsigned long long pend = TG_BALLOT(R.binned);
                while (pend != 0ull) {
                    const int l0 = __ffsll((long long)pend) - 1;
                    const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)R.bin, l0);
                    const unsigned long long m = TG_BALLOT(R.binned && R.bin == b0);
                    K7_COUNT(9, 1);                               // bin-grouping iterations
                    if ((m >> lane) & 1ull) {
                        R.my_leader = l0;
                        R.my_rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                        if (lane == l0) { leader = true; my_n = __popcll(m); }
                    }
                    pend &= ~m;
                }
                R.slot0 = 0u;
                if (leader) R.slot0 = atomicAdd(tb.cursor + R.bin, (uint32_t)my_n);
            };
            auto back = [&](Round& R) {
                const Texel3 &t00 = R.t00, &t01 = R.t01, &t10 = R.t10, &t11 = R.t11;
                const float w00 = (1.f - R.fx) * (1.f - R.fy), w01 = R.fx * (1.f - R.fy);
                const float w10 = (1.f - R.fx) * R.fy,         w11 = R.fx * R.fy;
                float x0 = 0.f, x1 = 0.f, x2 = 0.f;
                if (R.have) {
                    const float w = R.w;
                    const float4 dp = s_dpix[R.pl];
                    const float d0 = dp.x, d1 = dp.y, d2 = dp.z;
                    const float pre0 = TG_SH_C0 * (w00 * t00.x + w01 * t01.x + w10 * t10.x + w11 * t11.x) + R.vd0 + 0.5f;
                    const float pre1 = TG_SH_C0 * (w00 * t00.y + w01 * t01.y + w10 * t10.y + w11 * t11.y) + R.vd1 + 0.5f;
                    const float pre2 = TG_SH_C0 * (w00 * t00.z + w01 * t01.z + w10 * t10.z + w11 * t11.z) + R.vd2 + 0.5f;
                    const float qv = fmaxf(0.f, pre0) * d0 + fmaxf(0.f, pre1) * d1 + fmaxf(0.f, pre2) * d2;
                    // colour -> view-dependent term and texture
                    const float dc0 = (pre0 > 0.f) ? w * d0 : 0.f, dc1 = (pre1 > 0.f) ? w * d1 : 0.f, dc2 = (pre2 > 0.f) ? w * d2 : 0.f;
                    x0 = TG_SH_C0 * dc0; x1 = TG_SH_C0 * dc1; x2 = TG_SH_C0 * dc2;
                    const float dLdcol = x0 * ((1.f - R.fy) * (t01.x - t00.x) + R.fy * (t11.x - t10.x))
                                       + x1 * ((1.f - R.fy) * (t01.y - t00.y) + R.fy * (t11.y - t10.y))
                                       + x2 * ((1.f - R.fy) * (t01.z - t00.z) + R.fy * (t11.z - t10.z));
                    const float dLdrow = x0 * ((1.f - R.fx) * (t10.x - t00.x) + R.fx * (t11.x - t01.x))
                                       + x1 * ((1.f - R.fx) * (t10.y - t00.y) + R.fx * (t11.y - t01.y))
                                       + x2 * ((1.f - R.fx) * (t10.z - t00.z) + R.fx * (t11.z - t01.z));
                    const float dua = dLdcol * R.ka, dub = dLdrow * R.kb;
                    const float dum = -(dLdcol * R.kc + dLdrow * R.kd);
                    float du0, du1, du2;
                    if (R.axis == 0)      { du0 = dum; du2 = dua; du1 = dub; }
                    else if (R.axis == 1) { du1 = dum; du0 = dua; du2 = dub; }
                    else                  { du2 = dum; du0 = dua; du1 = dub; }
                    const float dden = -(du0 * R.nu0 + du1 * R.nu1 + du2 * R.nu2) * R.inv * R.inv;   // inv = 0 when den < DEN_MIN
                    // s = colour . dL/dcolour + (depth, normal) . dL/d(depth, normal) + dL/dalpha: everything stage C1's
                    // recurrence needs from this pair, formed here where all 64 lanes work
                    const float4 c5 = s_recs[5 * 64 + R.jj];
                    const float4 dg = *reinterpret_cast<const float4*>(&s_dgeo[R.pl * 4]);
                    s_items[R.e * 3].y = qv + c5.x * dg.x + c5.y * dg.y + c5.z * dg.z + c5.w * dg.w + dp.w;
                    s_items[R.e * 3 + 1] = make_float4(dc0, dc1, dc2, du0);
                    s_items[R.e * 3 + 2] = make_float4(du1, du2, R.inv, dden);
                }
                // texture gradient of this pair: append the record, or straight to dL_dtexture when the footprint is clamped at
                // a face border / the bin is full (still correct, just slow)
                const uint32_t slot = (uint32_t)__builtin_amdgcn_ds_bpermute(R.my_leader << 2, (int)R.slot0) + (uint32_t)R.my_rank;
                if (R.binned && slot < tb.cap) {
                    float* __restrict__ rp = tb.rec + (size_t)R.bin * tb.cap * 6 + slot;
                    rp[0] = __uint_as_float(R.cell);
                    rp[tb.cap] = R.fx; rp[2 * (size_t)tb.cap] = R.fy;
                    rp[3 * (size_t)tb.cap] = x0; rp[4 * (size_t)tb.cap] = x1; rp[5 * (size_t)tb.cap] = x2;
                } else if (R.have && (x0 != 0.f || x1 != 0.f || x2 != 0.f)) {
                    CubeTap ct;
                    ct.o00 = R.o00; ct.o01 = R.o00 + R.dox; ct.o10 = R.o00 + R.doy; ct.o11 = R.o00 + R.dox + R.doy;
                    scatter_direct(dtex, ct, w00, w01, w10, w11, x0, x1, x2);
                }
            };
            static_assert(BQ_CAP <= 128, "stage B is unrolled for at most two rounds per segment");
            K7_COUNT(7, 1); K7_COUNT(8, (n_items + 63) >> 6);                       // segments, stage-B rounds
            if (n_items > 0 && !(K7_ABL & 4)) {
                Round R0, R1;
                front(0, R0);
                if (n_items > 64) front(64, R1);
                back(R0);
                if (n_items > 64) back(R1);
            }
            __builtin_amdgcn_wave_barrier();
            // ================================================================ stage C1: per-pixel recurrence (sequential in j)
            // dL/dalpha_i = T_i s_i - (B_i + T_final bg . dL/dcolour) / (1 - alpha_i),  B_i = sum over the contributors k BEHIND i
            // of s_k alpha_k T_k: one running sum per pixel (`behind`), back to front.  s_i (colour . dL/dcolour + geometry
            // channels) comes ready-made from stage B; this loop leaves w = alpha T and P = dL/dpower in the item.
            int it0 = 0;
            uint32_t my_it0 = 0u;                                  // lane j: first item of Gaussian j in this segment
            if (K7_ABL != 0) my_it0_sink += (uint32_t)n_items + touched_lo;
            if (!(K7_ABL & 1)) {
                unsigned long long sm = seg_mask;
                while (sm != 0ull) {
                    const int jj = 63 - __clzll((long long)sm);
                    sm &= ~(1ull << jj);
                    const uint32_t blo = (uint32_t)__builtin_amdgcn_readlane((int)touched_lo, jj);
                    const uint32_t bhi = (uint32_t)__builtin_amdgcn_readlane((int)touched_hi, jj);
                    if (lane == jj) my_it0 = (uint32_t)it0;
                    if ((((unsigned long long)bhi << 32 | blo) >> lane) & 1ull) {
                        const int it = it0 + (int)__builtin_amdgcn_mbcnt_hi(bhi, __builtin_amdgcn_mbcnt_lo(blo, 0u));
                        const float4 i0 = s_items[it * 3];
                        const float Ti = i0.x, s_i = i0.y, araw = i0.z;
                        const float alpha = fminf(TG_ALPHA_MAX, araw);
                        const float w = alpha * Ti;
                        const float dL_dalpha_ = Ti * s_i - behind * __builtin_amdgcn_rcpf(1.0f - alpha);
                        behind = __fmaf_rn(s_i, w, behind);
                        // {w, P}: P = dL/dpower straight through the 0.99 clamp (lineage)
                        *reinterpret_cast<float2*>(&s_items[it * 3]) = make_float2(w, araw * dL_dalpha_);
                    }
                    it0 += __popcll(((unsigned long long)bhi << 32) | blo);
                }
            }
            // ================================================================ stage C2: per-Gaussian moment sums, 16 lanes per task
            // A task = up to 16 consecutive items of ONE Gaussian (its items are contiguous in the list).  Four tasks per
            // round: every lane forms the 28 moment terms of its item, a transposing butterfly over the 16 lanes (bank-masked
            // DPP for lane^4 / lane^8, quad_perm for lane^1 / lane^2) leaves two of the 32 row slots in each lane, and the
            // 16 lanes add the Gaussian's 128-byte accumulator row as two 64-byte runs.  (The wave-wide version of this --
            // one 64-lane butterfly per (wave, Gaussian) with ~17 of 64 lanes live -- was 42 % of K7's instructions.)
            if (!(K7_ABL & 2)) {
                const uint32_t nb_mine = ((seg_mask >> lane) & 1ull) ? (uint32_t)(__popc(touched_lo) + __popc(touched_hi)) : 0u;
                const uint32_t ntask = (nb_mine + 15u) >> 4;
                uint32_t incl = ntask;                            // inclusive prefix over the lanes (any fixed order works)
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64); if (lane >= d) incl += o; }
                const int total = __builtin_amdgcn_readlane((int)incl, 63);
                __builtin_amdgcn_wave_barrier();
                for (uint32_t t = 0; t < ntask; ++t)             // task word: j | first item << 6 | item count << 14
                    s_task[incl - ntask + t] = (uint32_t)lane | ((my_it0 + 16u * t) << 6) | (min(16u, nb_mine - 16u * t) << 14);
                __builtin_amdgcn_wave_barrier();
                const int sub = lane & 15;
                for (int q0 = 0; q0 < total; q0 += 4) {
                    K7_COUNT(10, 1);                              // C2 rounds (4 tasks each)
                    const int q = q0 + (lane >> 4);
                    const uint32_t task = (q < total) ? s_task[q] : 0u;
                    const int jt = (int)(task & 63u);
                    const bool have = (uint32_t)sub < (task >> 14);
                    // lanes without an item read the all-zero item behind the list: every moment below comes out 0 with no
                    // branch and no 32-register clear (the butterfly needs all 64 lanes anyway)
                    const int item = have ? (int)((task >> 6) & 255u) + sub : BQ_CAP;
                    // Gaussian index of the task: lane jt holds instance jt's (all 64 lanes are active here: bpermute reads 0
                    // from an inactive source lane)
                    const uint32_t gid = (uint32_t)__builtin_amdgcn_ds_bpermute(jt << 2, (int)id);
                    float part[32];
                    const float2 gxy = *reinterpret_cast<const float2*>(&s_recs[jt]);
                    {
                        const float4 i0 = s_items[item * 3], i1 = s_items[item * 3 + 1], i2 = s_items[item * 3 + 2];
                        const uint32_t key = __float_as_uint(i0.w);
                        const int pl = (int)(key >> 8) & 63;
                        const float w = i0.x, P = i0.y;
                        const float dx = gxy.x - (float)(wave_px + (pl & 7)), dy = gxy.y - (float)(wave_py + (pl >> 3));   // xy - pixel
                        // RAW MOMENTS about the splat centre (TexGSGrads.acc layout, texgs.h); K8, which has conic / opacity /
                        // G / g in registers anyway, turns them into dL/d(xy, conic, opacity, G, g, ...)
                        const float du0 = i1.w, du1 = i2.x, du2 = i2.y, inv = i2.z, dden = i2.w;
                        const float dn0 = du0 * inv, dn1 = du1 * inv, dn2 = du2 * inv;
                        const float dpx = -dx, dpy = -dy;                       // pixel - xy
                        const float Pdx = P * dx, Pdy = P * dy;
                        const float4 dg = *reinterpret_cast<const float4*>(&s_dgeo[pl * 4]);
                        part[M_P] = P; part[M_P + 1] = Pdx; part[M_P + 2] = Pdy;
                        part[M_P + 3] = Pdx * dx; part[M_P + 4] = Pdx * dy; part[M_P + 5] = Pdy * dy;
                        part[M_DEN] = dden; part[M_DEN + 1] = dden * dpx; part[M_DEN + 2] = dden * dpy;
                        part[M_DN + 0] = dn0; part[M_DN + 1] = dn0 * dpx; part[M_DN + 2] = dn0 * dpy;
                        part[M_DN + 3] = dn1; part[M_DN + 4] = dn1 * dpx; part[M_DN + 5] = dn1 * dpy;
                        part[M_DN + 6] = dn2; part[M_DN + 7] = dn2 * dpx; part[M_DN + 8] = dn2 * dpy;
                        part[M_PHI] = du0; part[M_PHI + 1] = du1; part[M_PHI + 2] = du2;
                        part[M_VD] = i1.x; part[M_VD + 1] = i1.y; part[M_VD + 2] = i1.z;
                        part[M_DEPTH] = w * dg.x;
                        part[M_N] = w * dg.y; part[M_N + 1] = w * dg.z; part[M_N + 2] = w * dg.w;
#pragma unroll
                        for (int k = M_N + 3; k < 32; ++k) part[k] = 0.f;
                    }
                    float lo, hi;
                    reduce32_rows16(part, lane, lo, hi);          // lane holds slots transposed_index(lane & 15) and 16 + that
                    if (q < total) {
                        float* row = acc + (size_t)gid * TEXGS_ACC_FLOATS + transposed_index(sub);
                        if (lo != 0.f) unsafeAtomicAdd(row, lo);
                        if (hi != 0.f) unsafeAtomicAdd(row + 16, hi);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
#ifdef K7_STATS
    K7_COUNT(11, 1);
    if (lane == 0) for (int i = 0; i < 16; ++i) if (k7s[i]) atomicAdd(&g_k7_stats[i], (unsigned long long)k7s[i]);
#endif
    if (K7_ABL != 0 && s_items[lane * 3].x == 12345.678f) acc[0] = (float)my_it0_sink;    // ablation builds: keep the LDS traffic alive
}

// ------------------------------------------------------------------------------------------------ texture-gradient reduce
// One workgroup per 32x32-texel bin: sum the bin's records into a 33x33-texel LDS tile (footprints anchored in the bin
// reach one texel past its right / bottom edge, still inside the face), then add every non-zero texel of the tile to
// dL_dtexture[6,R,R,3] once -- 99 consecutive dwords per tile row, i.e. coalesced memory-side requests; neighbouring
// bins overlap in that one-texel seam, hence atomics.  Leaves the cursor at 0 for the next call.
// The tile is 64-bit FIXED POINT: LDS float atomics retire ~3 cycles per lane on gfx950 (ds_add_f32: 193 cycles per wave
// instruction, ds_add_u64: 6; scripts/ubench/lds_atomics.hip), which made the first version of this kernel 1.8 ms.  Scale:
// every record value is bounded by C0 * max|dL/dpixel colour| (K7 leaves that maximum in stats[1]) and is mapped to
// < 2^42, so 2^20 records per bin cannot overflow; resolution 2^-42 of the image-wide bound, sums exact and
// order-independent (the texture gradient of the binned path is bit-reproducible run to run).
#define TB_EDGE 33
__global__ void __launch_bounds__(256)
k_texgrad_reduce(int R, TexBinArgs tb, float* __restrict__ dtex) {
    __shared__ long long s_tile[TB_EDGE * TB_EDGE * 3];          // [row][col][channel], 2^42-scaled fixed point
    const int b = (int)blockIdx.x, tid = (int)threadIdx.x;
    const uint32_t filled = tb.cursor[b];
    if (filled == 0u) return;                                  // uniform per workgroup
    for (int k = tid; k < TB_EDGE * TB_EDGE * 3; k += 256) s_tile[k] = 0ll;
    __syncthreads();
    const float bound = TG_SH_C0 * __uint_as_float(tb.stats[1]);
    int e = 0;
    (void)frexpf(bound, &e);                                   // bound < 2^e
    const double up = (double)ldexpf(1.0f, 42 - e);
    const float down = ldexpf(1.0f, e - 42);
    const uint32_t cnt = min(filled, tb.cap);
    const float* __restrict__ rp = tb.rec + (size_t)b * tb.cap * 6;
    const size_t cap = tb.cap;
    // float -> int64 without the 11-instruction generic conversion: |v| < 2^42, so v + 1.5 * 2^52 (exact in double) carries
    // round(v) in its mantissa; subtracting the bias as integers leaves the two's-complement value
    const double magic = 6755399441055744.0;
    const long long magic_bits = __double_as_longlong(magic);
    auto add_record = [&](uint32_t cell, float fx, float fy, float x0, float x1, float x2) {
        const double dx0 = (double)x0 * up, dx1 = (double)x1 * up, dx2 = (double)x2 * up;
        const double w00 = (double)((1.f - fx) * (1.f - fy)), w01 = (double)(fx * (1.f - fy));
        const double w10 = (double)((1.f - fx) * fy), w11 = (double)(fx * fy);
        unsigned long long* t = reinterpret_cast<unsigned long long*>(s_tile) + ((cell >> 8) * TB_EDGE + (cell & 0xFFu)) * 3;
#define TB_ADD(P, V) atomicAdd((P), (unsigned long long)(__double_as_longlong((V) + magic) - magic_bits))
        TB_ADD(t + 0, w00 * dx0); TB_ADD(t + 1, w00 * dx1); TB_ADD(t + 2, w00 * dx2);
        TB_ADD(t + 3, w01 * dx0); TB_ADD(t + 4, w01 * dx1); TB_ADD(t + 5, w01 * dx2);
        TB_ADD(t + TB_EDGE * 3 + 0, w10 * dx0); TB_ADD(t + TB_EDGE * 3 + 1, w10 * dx1); TB_ADD(t + TB_EDGE * 3 + 2, w10 * dx2);
        TB_ADD(t + TB_EDGE * 3 + 3, w11 * dx0); TB_ADD(t + TB_EDGE * 3 + 4, w11 * dx1); TB_ADD(t + TB_EDGE * 3 + 5, w11 * dx2);
#undef TB_ADD
    };
    // two records per thread in flight (12 loads).  Measured and dropped: 4 / 8 in flight, two tile copies for even / odd lanes, a
    // planar [channel][row][col] tile, plain read-modify-write of the tile interior on the way out (profiles/README.md): the
    // kernel sits at 2x its HBM floor (the records, 0.45 GB at C3) with 66 % LDS bank-conflict cycles on the random atomics
    uint32_t i = (uint32_t)tid;
    for (; i + 256u < cnt; i += 512u) {
        const uint32_t i2 = i + 256u;
        const uint32_t ca = __float_as_uint(rp[i]), cb = __float_as_uint(rp[i2]);
        const float fxa = rp[cap + i], fya = rp[2 * cap + i], xa0 = rp[3 * cap + i], xa1 = rp[4 * cap + i], xa2 = rp[5 * cap + i];
        const float fxb = rp[cap + i2], fyb = rp[2 * cap + i2], xb0 = rp[3 * cap + i2], xb1 = rp[4 * cap + i2], xb2 = rp[5 * cap + i2];
        add_record(ca, fxa, fya, xa0, xa1, xa2);
        add_record(cb, fxb, fyb, xb0, xb1, xb2);
    }
    if (i < cnt) add_record(__float_as_uint(rp[i]), rp[cap + i], rp[2 * cap + i], rp[3 * cap + i], rp[4 * cap + i], rp[5 * cap + i]);
    __syncthreads();
    const int face = b / (tb.nb * tb.nb), by = (b / tb.nb) % tb.nb, bx = b % tb.nb;
    for (int k = tid; k < TB_EDGE * TB_EDGE * 3; k += 256) {
        const long long q = s_tile[k];
        if (q == 0ll) continue;
        const int row = k / (TB_EDGE * 3), c = k - row * (TB_EDGE * 3);
        const int y = by * 32 + row, xq = bx * 96 + c;
        if (y >= R || xq >= R * 3) continue;
        // rows / columns 0 and 32 of the tile are shared with the neighbouring bins' tiles, hence atomics
        unsafeAtomicAdd(dtex + ((size_t)(face * R + y) * R) * 3 + xq, (float)q * down);
    }
    if (tid == 0) {
        tb.cursor[b] = 0u;
        if (filled > tb.cap) atomicMax(tb.stats, filled);      // overflowed: tell the host how long the list wanted to be
    }
}

inline PixArgs make_pix(const CamConst& c, const TexGSFrame* f, const TexGSInputs* in, const TexGSGeom* g,
                        const TexGSBinning* b) {
    PixArgs a;
    a.W = c.W; a.H = c.H; a.tiles_x = c.tiles_x; a.num_tiles = c.tiles_x * c.tiles_y; a.R = c.R;
    a.ranges = reinterpret_cast<const uint2*>(b->ranges);
    a.point_list = b->point_list;
    a.tile_order = b->tile_order;
    a.rec = reinterpret_cast<const float4*>(g->rec);
    a.texture = in->texture;
    a.bg = f->bg;
    return a;
}

inline TexBinArgs make_bins(const CamConst& c, const TexGSGrads* gr) {
    TexBinArgs tb;
    tb.nb = (c.R + 31) >> 5;
    const bool on = gr->tex_bins != nullptr && gr->tex_bin_cursor != nullptr && gr->tex_bin_cap > 0;
    tb.rec = on ? gr->tex_bins : nullptr;
    tb.cursor = on ? gr->tex_bin_cursor : nullptr;
    tb.stats = on ? gr->tex_bin_cursor + tex_bin_count(c.R) : nullptr;
    tb.cap = on ? gr->tex_bin_cap : 0u;
    return tb;
}

}  // namespace

size_t tex_bin_count(int R) {
    const size_t nb = (size_t)((R + 31) >> 5);
    return 6 * nb * nb;
}

void launch_render_fwd(const CamConst& c, const TexGSFrame* f, const TexGSInputs* in, const TexGSGeom* g,
                       const TexGSBinning* b, TexGSImage* img, hipStream_t s) {
    PixArgs a = make_pix(c, f, in, g, b);
#ifdef TEXGS_EXPERIMENTS     // critical-path experiments: blend only the K longest tile lists (results are wrong by construction)
    if (getenv("TEXGS_MAXTILES_FWD")) a.num_tiles = min(a.num_tiles, atoi(getenv("TEXGS_MAXTILES_FWD")));
#endif
    hipLaunchKernelGGL(k_render_fwd, dim3(blend_grid(a.num_tiles)), dim3(TG_WG_THREADS), 0, s, a, img->out_color, img->out_depth,
                       img->out_norm, img->out_alpha, img->final_T, img->n_contrib);
}

void launch_render_bwd(const CamConst& c, const TexGSFrame* f, const TexGSInputs* in, const TexGSGeom* g,
                       const TexGSBinning* b, const TexGSImage* img, TexGSGrads* gr, hipStream_t s) {
    PixArgs a = make_pix(c, f, in, g, b);
#ifdef TEXGS_EXPERIMENTS
    if (getenv("TEXGS_MAXTILES_BWD")) a.num_tiles = min(a.num_tiles, atoi(getenv("TEXGS_MAXTILES_BWD")));
#endif
    hipLaunchKernelGGL(k_render_bwd, dim3(blend_grid(a.num_tiles)), dim3(TG_WG_THREADS), 0, s, a, make_bins(c, gr), img->final_T,
                       img->n_contrib, gr->dL_dcolor, gr->dL_ddepth, gr->dL_dnorm, gr->dL_dalpha, gr->acc, gr->dL_dtexture);
}

void launch_texgrad_reduce(const CamConst& c, TexGSGrads* gr, hipStream_t s) {
    const TexBinArgs tb = make_bins(c, gr);
    if (!tb.rec) return;
    hipLaunchKernelGGL(k_texgrad_reduce, dim3((unsigned)tex_bin_count(c.R)), dim3(256), 0, s, c.R, tb, gr->dL_dtexture);
}

#ifdef K7_STATS
extern "C" int texgs_debug_k7_stats(unsigned long long* out16, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_k7_stats), 16 * sizeof(unsigned long long));
    if (e == hipSuccess && reset) { unsigned long long z[16] = {0}; e = hipMemcpyToSymbol(HIP_SYMBOL(g_k7_stats), z, sizeof(z)); }
    return (int)e;
}
#endif
