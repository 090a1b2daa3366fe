// K6 render forward, K7 render backward, and the texture-gradient bin reduce -- gfx950 (CDNA4), wave64.
//
// One wave = one 8x8 pixel block of a 16x16 tile = FOUR 4x4 quadrants, one per DPP row of 16 lanes; the waves of a tile
// share nothing (no block barrier anywhere; one wave per workgroup = finest dispatch).
//
// Work flows through three levels, each denser than the one before (SURVEY.md A.4 / A.5 is the contract):
//   raw batches    64 instances of the tile's depth-sorted list at a time, lane = instance: only the 32-byte TEST record
//                  (xy, conic, opacity, cull radius / threshold) is fetched, and an exact concave-maximum test decides
//                  whether the instance can reach alpha >= 1/255 anywhere in the 8x8 block.  Survivors (about a third) are
//                  queued in depth order.
//   chunks         64 SURVIVORS at a time, lane = survivor: the 80-byte SHADING record is fetched only now, the chunk is
//                  laid out as planes in LDS, and the same exact test per 4x4 quadrant builds four per-quadrant lists.
//   lock-step test every quadrant (16 lanes, lane = pixel) walks ITS OWN list: one iteration tests up to four different
//                  Gaussians, one per quadrant (a Gaussian covers ~17 of the 64 pixels of a block: one Gaussian per
//                  iteration ran with 27 % of the lanes).  The tested Gaussian's parameters come from the LDS planes with
//                  row-broadcast reads, software-prefetched one iteration ahead; transmittance stays in the pixel's lane.
//   dense phase    contributing (pixel, Gaussian) pairs are compacted (ballot + mbcnt) into a per-wave LDS list and the
//                  texture work (UV Taylor step, cubemap address, 4 taps, colour / gradients) runs 64 pairs at a time.
//
// Texture gradient (K7): every fp32 global atomic on this part executes memory-side at ~20 G requests/s and the ~19 M
// bilinear footprints of a C3 view cost 0.69 ms that way (profiles/r02_ablation.md).  Instead K7 WRITES one 16-byte
// record {fx, fy, cell, dL/dtexel-colour (3)} per footprint with ONE non-temporal store into the list of the 32x32-texel
// texture block ("bin") the footprint is anchored in, and k_texgrad_reduce then sums each bin's list in LDS and adds every
// texel of the block to dL_dtexture once.  The lists are EXACTLY sized and every slot has an owner before K7 starts: K6
// counts the footprints per (8x8 pixel block, bin) in a 16-entry LDS table while it renders and, at block end, RESERVES the
// block's range of each list with one returning atomic per entry on the bin's total; a one-workgroup scan turns the totals
// into list offsets; K7 reads its block's table back and hands out slots with an LDS atomic (round 5; until then: a grouping
// loop and one returning GLOBAL atomic per (wave round, distinct bin) on a cursor -- 100 of K7's 700 us).  The records of a
// view are one contiguous array (~0.30 GB at C3) -- no per-bin capacity, no chunk tables, nothing to wait for.
// What bounds these kernels (round 6): the path behind the L2 -- gather / scatter requests into the Infinity
// Cache -- not instruction issue; write-once / read-once streams are therefore non-temporal (nt_load / nt_store).
// No MFMA: there is no dense contraction on this path.
#include "common.h"
#include "wave_ops.h"

namespace {

typedef unsigned long long ull;

struct __attribute__((packed, aligned(4))) Texel3 { float x, y, z; };   // one global_load_dwordx3 per tap
// `boff` = BYTE offset into the texture (< 2^32: abi.hip rejects tex_res > 7168): base in SGPRs + a zero-extended 32-bit lane
// offset is the `global_load v, v_off, s[base]` form -- no 64-bit address arithmetic per tap
__device__ __forceinline__ Texel3 load_texel(const float* __restrict__ tex, uint32_t boff) {
    return *reinterpret_cast<const Texel3*>(reinterpret_cast<const char*>(tex) + boff);
}

// records are written once (K7) and read once (the reduce): non-temporal on both sides, they should not displace texel lines and
// shading records from the L2 (A/B against plain accesses: K7 FETCH_SIZE 1.42 -> 1.03 GB, 589 -> 578 us; profiles/r06_ablation.md)
typedef uint32_t tg_u4 __attribute__((ext_vector_type(4)));
// the same for the other read-once / write-once streams of the blend kernels (survivor lists, per-pixel inputs and outputs, the
// reservation tables): whatever is touched once should not take an L2 line from a texel or a shading record
template <class T> __device__ __forceinline__ T nt_load(const T* p) {
    return __builtin_nontemporal_load(p);
}
template <class T> __device__ __forceinline__ void nt_store(T* p, T v) {
    __builtin_nontemporal_store(v, p);
}
__device__ __forceinline__ uint2 nt_load2(const uint2* p) {
    const unsigned long long v = nt_load(reinterpret_cast<const unsigned long long*>(p));
    return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
}
__device__ __forceinline__ void nt_store2(uint2* p, uint2 v) {
    nt_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)v.x | ((unsigned long long)v.y << 32));
}

// Cubemap address of direction u (not necessarily unit): face (+x,-x,+y,-y,+z,-z; NVDIFFREC/util.py:94-101
// inverted), bilinear taps with clamp-to-edge inside the face, texel centres at (i+0.5)/R.
struct CubeTap {
    uint32_t o00, dox, doy;     // BYTE offset of tap 00's first channel; o01 = o00 + dox, o10 = o00 + doy, o11 = o00 + dox + doy
                                // (dox = 12 or 0, doy = 12 R or 0: 0 when the footprint is clamped at the face border)
    int   x0, y0;               // clamped coordinates of tap 00 inside the face
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
    t.x0 = x0c; t.y0 = y0c;
    // 24-bit multiplies (full rate; v_mul_lo_u32 is quarter rate): face * R + y < 6 R < 2^24, R < 2^24
    const uint32_t texel = __umul24((uint32_t)(t.face * R + y0c), (uint32_t)R) + (uint32_t)x0c;
    t.o00 = (texel * 3u) << 2;
    t.dox = (x1c != x0c) ? 12u : 0u;
    t.doy = (y1c != y0c) ? 12u * (uint32_t)R : 0u;
    return t;
}

// The falloff exponent, shared by K6 and K7: K7 must reproduce K6's contributor decisions (power <= 0, alpha >= 1/255)
// bit for bit, so both evaluate this one explicitly ordered sequence of fp32 operations.  K1 stores the conic pre-scaled,
// (ah, bh, ch) = (-a/2, -b, -c/2), so power = ah dx^2 + bh dx dy + ch dy^2 is 3 mul + 1 mul + 2 fma.
__device__ __forceinline__ float gauss_power(float ah, float bh, float ch, float dx, float dy) {
    return __fmaf_rn(bh, __fmul_rn(dx, dy), __fmaf_rn(ch, __fmul_rn(dy, dy), __fmul_rn(ah, __fmul_rn(dx, dx))));
}
__device__ __forceinline__ float gauss_alpha_raw(float op, float power) { return op * __expf(power); }

// Cull, lane-parallel (lane = one instance): can the instance reach alpha >= 1/255 at ANY point of the pixel rectangle
// [x0, x0 + ext] x [y0, y0 + ext] (ext = 7: the wave's 8x8 block, ext = 3: one 4x4 quadrant)?  power(d) = ah dx^2 + bh dx dy +
// ch dy^2 is concave with its maximum 0 at the splat centre, so its maximum over the rectangle is 0 if the centre is inside,
// else it sits on one of the four edges, where it is a 1-D concave parabola maximised at the clamped stationary point.
// Exact for the continuous rectangle, hence conservative for the pixel centres; `thr` already carries a margin far above
// fp32 rounding.  (The bounding-disc test alone let through twice as many instances as ever produced an item: edge-on
// splats are needles, not discs.)  A false positive costs a test, never a result.
__device__ __forceinline__ bool block_reachable(float gx, float gy, float ah, float bh, float ch, float thr, float rcull,
                                                float x0, float y0, float ext) {
    const float x1 = x0 + ext, y1 = y0 + ext;
    if (!(rcull >= 0.f && (gx + rcull >= x0) && (gx - rcull <= x1) && (gy + rcull >= y0) && (gy - rcull <= y1))) return false;
    const float dx0 = gx - x1, dx1 = gx - x0, dy0 = gy - y1, dy1 = gy - y0;     // d = centre - pixel ranges over [dx0,dx1] x [dy0,dy1]
    if (dx0 <= 0.f && dx1 >= 0.f && dy0 <= 0.f && dy1 >= 0.f) return true;     // centre inside the rectangle
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
    const float4* rec_test;     // [N][2]: (xy, ah, bh) | (ch, opacity, rcull, thr)
    const float4* rec_shade;    // [N][5]: (g, G0, G1) | (G2..G5) | (phi, vd0) | (vd1, vd2, depth, n0) | (n1, n2, -, -)
    const float* texture;
    const float* bg;
    // K6 -> K7 hand-off (NULL in a forward-only call): the survivors of every 8x8 block's cull, in list order.  Block (tile, w)
    // owns entries [4 * range.x + w * len, ... + len) (len = the tile's list length: an upper bound of its survivors).
    uint2*    surv;          // {Gaussian id, list position}
    uint16_t* surv_qm;       // bit q: the survivor can reach quadrant q
    uint32_t* surv_cnt;      // [4 * tiles] survivors written per block
    uint32_t* resv;          // [4 * tiles][3][64] per-block reservation table {bin | offset inside the bin's list | count} (NULL: no binned texture gradient)
    // K6 -> K7 ITEM STREAM (texgs.h v15; item_pages NULL: off): one {T, alpha_raw, Gaussian id << 6 | pixel lane} per contributing pair
    uint32_t* item_pages;    // [cap][3][TG_PAGE]
    uint32_t* item_link;     // [cap] previous page of the same block
    uint32_t* item_tail;     // [4 * tiles][2] {last page, items}
    uint32_t* item_ctl;      // sub-pool cursors (every 16th word) + overflow flag
    uint32_t  item_sub_cap;  // pages per sub-pool
    uint32_t  item_sub_mask; // sub-pools - 1
    const uint32_t* run_if;  // the survivor-replay K7 only: do nothing unless this word is non-zero (NULL: always run)
};
#define TG_PAGE TEXGS_ITEM_PAGE
#define TG_PAGE_SHIFT 8
static_assert((1 << TG_PAGE_SHIFT) == TG_PAGE, "page size");
#define TG_NOPAGE 0xFFFFFFFFu
#define TG_PAGE_UNSET 0xFFFFFFFEu

// workgroup (= one wave) -> (tile, 8x8 block).  Tiles are launched longest-list-first (tile_order).  The four blocks of a
// tile get ids that are equal mod 8, so they run on the same XCD (workgroup b is observed on XCD b % 8: speed only) and
// share its L2 for the tile's records.
__device__ __forceinline__ bool wave_block(const PixArgs& a, int& tile, int& wave) {
    const int b = (int)blockIdx.x, k = b >> 3;
    wave = k & 3;
    const int rank = ((k >> 2) << 3) | (b & 7);
    if (rank >= a.num_tiles) return false;
    tile = (int)a.tile_order[rank];
    return true;
}
inline int blend_grid(int num_tiles) { return 4 * ((num_tiles + 7) & ~7); }

// HIP's __ballot(int) materialises the predicate as 0/1 in a VGPR and compares it again (v_cndmask + v_cmp per ballot);
// the builtin takes the lane mask the compares already produced.
#define TG_BALLOT(P) __builtin_amdgcn_ballot_w64((bool)(P))
__device__ __forceinline__ int mbcnt64(ull m) {
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// Texture bin (32x32-texel block) of a bilinear footprint, and whether the footprint is binned at all (not clamped at a face
// border: such footprints go straight to dL_dtexture).  K6 counts with this, K7 appends with this: same inputs, same answer.
__device__ __forceinline__ uint32_t tap_bin(const CubeTap& ct, int nb) { return (uint32_t)((ct.face * nb + (ct.y0 >> 5)) * nb + (ct.x0 >> 5)); }
__device__ __forceinline__ bool tap_binned(const CubeTap& ct) { return ct.dox != 0u && ct.doy != 0u; }
// Per-block RESERVATION table (K6 -> K7), round 5.  The bilinear footprints of an 8x8 pixel block fall into 11 texture bins at
// the median and up to ~45 (C3; measured with oracle/texgs_ref.c texgs_ref_block_bin_stats: front and back shell, needle-shaped
// splats whose Taylor term sweeps the cube face) -- a DIRECT-MAPPED table of 64 entries indexed by the low bits of the bin's (x, y)
// inside the face holds 97.8 % of the footprints (16 entries: 87 %, 32: 96 %), the first bin to arrive owns a slot.  K6 counts per
// entry and, at block end, takes the block's range of every owned bin's record list with ONE returning atomic per entry; K7 reads
// the table back and hands out slots with one LDS atomic per lane -- no grouping loop, no global cursor.  Footprints whose slot
// belongs to another bin take the round-4 path: counted per (round, bin) by a global atomic in K6, appended behind the reserved
// part of the list through a global cursor in K7.  The bins live in the spare fourth word of plane B (entries 0..63 of
// Planes::B[].w: LDS the layout had and did not use), the counts as packed 16-bit pairs: not one byte of LDS more than round 4.
// (A count beyond 65 535 wraps into its neighbour: the reservation is then too small / too large by that much, K7 sends what does
// not fit to dL_dtexture directly and zero-fills what stays empty -- slower, still exact.)
#define TG_RESV 64
#define TG_RESV_EMPTY 0xFFFFFFFFu
#define TG_SLOT_NONE 0xFFFFFFFFu
#define TG_SLOT_OVF  0x80000000u          // | leader lane << 8 | rank: an overflow footprint's slot, resolved in the back half
__device__ __forceinline__ int tap_home(const CubeTap& ct) { return ((ct.x0 >> 5) & 7) | (((ct.y0 >> 5) & 7) << 3); }
template <class P> __device__ __forceinline__ uint32_t& resv_bin(P& p, int h) { return reinterpret_cast<uint32_t*>(&p.B[h])[3]; }

// ---- per-wave LDS layout shared by K6 and K7 ----
#define TG_RING 128          // survivor queue (raw list positions), power of two >= 127
#define TG_DUMMY 64          // plane slot of the all-zero dummy survivor (alpha 0: never contributes); list padding points here
struct Planes {              // the chunk's 64 survivors, plane-major [field group][survivor]: row-broadcast reads are conflict-free
    float4 A[65];            // xy, ah, bh                      (test)
    float4 B[65];            // ch, opacity, list position, [.w of entries 0..63: the block's reservation table, resv_bin()]    (test)
    float4 C[65];            // depth, normal                    (K6 blend, K7 stage B)
    float4 D[65];            // g, G0, G1                        (dense phase)
    float4 E[65];            // G2..G5
    float4 F[65];            // phi, vd0
    float2 G[66];            // vd1, vd2
};

// item key: survivor slot (7 bits) | pixel lane << 8 | pixel x offset in the block << 14 | y offset << 17
#define KEY_J(K)   ((int)((K) & 127u))
#define KEY_PL(K)  ((int)(((K) >> 8) & 63u))
#define KEY_OX(K)  ((int)(((K) >> 14) & 7u))
#define KEY_OY(K)  ((int)(((K) >> 17) & 7u))

// lane -> pixel of the 8x8 block: DPP row q = lane >> 4 is the 4x4 quadrant (q & 1, q >> 1), lanes of a row in row-major order
__device__ __forceinline__ void lane_pixel(int lane, int& ox, int& oy) {
    ox = (((lane >> 4) & 1) << 2) | (lane & 3);
    oy = (((lane >> 5) & 1) << 2) | ((lane >> 2) & 3);
}

// Chunk load, lane = survivor: shading record from HBM (the test record again: an L2 hit), planes to LDS.
__device__ __forceinline__ void load_chunk(const PixArgs& a, Planes& P, int lane, bool live, uint32_t id, uint32_t pos,
                                           float4& T0, float4& T1) {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 S0 = z, S1 = z, S2 = z, S3 = z, S4 = z;
    T0 = z; T1 = make_float4(0.f, 0.f, -1.f, 1.f);
    if (live) {
        const float4* __restrict__ tp = a.rec_test + 2 * (size_t)id;
        const float4* __restrict__ sp = a.rec_shade + (TEXGS_REC_SHADE_FLOATS / 4) * (size_t)id;
        T0 = tp[0]; T1 = tp[1]; S0 = sp[0]; S1 = sp[1]; S2 = sp[2]; S3 = sp[3]; S4 = sp[4];
    }
    P.A[lane] = T0;
    { float* pb = reinterpret_cast<float*>(&P.B[lane]); pb[0] = T1.x; pb[1] = T1.y; pb[2] = __uint_as_float(pos); }   // (.w: the reservation table)
    P.C[lane] = make_float4(S3.z, S3.w, S4.x, S4.y);
    P.D[lane] = S0; P.E[lane] = S1; P.F[lane] = S2;
    P.G[lane] = make_float2(S3.x, S3.y);
}
__device__ __forceinline__ void init_dummy(Planes& P, int lane) {
    if (lane == 0) {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        P.A[TG_DUMMY] = z; P.B[TG_DUMMY] = make_float4(0.f, 0.f, __uint_as_float(0xFFFFFFFFu), 0.f); P.C[TG_DUMMY] = z;
        P.D[TG_DUMMY] = z; P.E[TG_DUMMY] = z; P.F[TG_DUMMY] = z; P.G[TG_DUMMY] = make_float2(0.f, 0.f);
    }
}

// ------------------------------------------------------------------------------------------------ K6
// Forward blend.  Depth, normal and alpha accumulate in the lock-step test loop (w = alpha*T needs no texture); the colour
// of every contributing (pixel, Gaussian) pair is added by the dense phase into the pixel's LDS accumulator as Q32.32
// fixed point with integer atomics -- the sum is order-independent, so it equals the in-order blend and is bit-reproducible.
#define FQ_CAP 128
struct __attribute__((aligned(16))) FwdLds {
    Planes p;                           // 6768 B
    uint2 q[FQ_CAP];                    // 1024: dense-phase queue {w, key}
    ull col[64 * 3];                    // 1536: Q32.32 colour sums of the wave's pixels
    uint32_t ring[TG_RING];             //  512
    uint8_t list[4][64];                //  256: per-quadrant survivor lists, padded with TG_DUMMY
    uint32_t ccnt[TG_RESV / 2];         //  128: footprints counted per reservation entry, two 16-bit counts per word (the bins: p.B[].w)
};                                      // 10224 B -> 16 waves per CU

// TAPS = false: the untextured surface (TexGSInputs.texture == NULL; render/render.py:75-84 through `diff_gauss`): the colour of a
// pair is max(0, viewdep + 0.5), no UV step, no cubemap address, no taps.
template <bool TAPS>
__global__ void __launch_bounds__(64, 4)
k_render_fwd(PixArgs a, float* __restrict__ out_color, float* __restrict__ out_depth, float* __restrict__ out_norm,
             float* __restrict__ out_alpha, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
             uint32_t* __restrict__ bin_count) {
    __shared__ FwdLds L;
    const int lane = (int)threadIdx.x;
    int tile, wave;
    if (!wave_block(a, tile, wave)) return;
    const int tile_x = tile % a.tiles_x, tile_y = tile / a.tiles_x;
    const int wave_px = tile_x * TEXGS_TILE + ((wave & 1) << 3), wave_py = tile_y * TEXGS_TILE + ((wave >> 1) << 3);
    int ox, oy;
    lane_pixel(lane, ox, oy);
    const int px = wave_px + ox, py = wave_py + oy;
    const bool inside = (px < a.W) && (py < a.H);
    const float pxf = (float)px, pyf = (float)py;
    const float wpx = (float)wave_px, wpy = (float)wave_py;
    const uint2 range = a.ranges[tile];
    const int todo = (int)(range.y - range.x);
    const float* __restrict__ tex = a.texture;
    const uint32_t keybase = ((uint32_t)lane << 8) | ((uint32_t)ox << 14) | ((uint32_t)oy << 17);
    const uint8_t* mylist = L.list[lane >> 4];
    const int nbins_row = (a.R + 31) >> 5;

    L.col[lane * 3 + 0] = 0ull; L.col[lane * 3 + 1] = 0ull; L.col[lane * 3 + 2] = 0ull;   // own pixel; only this wave touches it
    resv_bin(L.p, lane) = TG_RESV_EMPTY;
    if (lane < TG_RESV / 2) L.ccnt[lane] = 0u;
    init_dummy(L.p, lane);
    __builtin_amdgcn_wave_barrier();

    bool done = !inside;
    ull done_mask = TG_BALLOT(!inside);        // the same, as a wave-level lane mask (scalar registers)
    float T = 1.0f;
    float Dp = 0.f, N0 = 0.f, N1 = 0.f, N2 = 0.f, Al = 0.f;
    uint32_t last = 0;
    int qhead = 0, qtail = 0;                  // wave-uniform

    // The dense phase is software-pipelined by one batch: drain(n) first FINISHES the previous batch (its 4 taps were
    // loaded a whole batch interval ago: colour, Q32.32 accumulate), then STARTS the new one (queue pop, record fields, UV
    // Taylor step, cubemap address, tap loads issued) and returns without waiting for them.
    // K6 -> K7 item stream (see PixArgs): pages from this block's sub-pool, one returning atomic per page, issued a page ahead
    const bool streaming = a.item_pages != nullptr;
    const uint32_t sub = (uint32_t)blockIdx.x & a.item_sub_mask;
    uint32_t cur_page = TG_PAGE_UNSET, prev_page = TG_NOPAGE;      // wave-uniform
    uint32_t pend_v = 0u;                      // lane 0: the pending allocation's result
    int nitems = 0;
    uint32_t cid = 0u;                         // lane = survivor of the current chunk: its Gaussian id
    auto alloc_issue = [&]() { if (lane == 0) pend_v = atomicAdd(a.item_ctl + 16u * sub, 1u); };
    auto alloc_take = [&](uint32_t prev) -> uint32_t {      // the pending allocation: its page (chained behind `prev`), or TG_NOPAGE
        const uint32_t local = (uint32_t)__builtin_amdgcn_readfirstlane((int)pend_v);
        if (local < a.item_sub_cap) {
            const uint32_t page = sub * a.item_sub_cap + local;
            if (lane == 0) a.item_link[page] = prev;
            return page;
        }
        if (lane == 0) a.item_ctl[TEXGS_ITEM_CTL_FLAG] = 1u;     // the buffer is too small: the survivor-replay K7 runs instead
        return TG_NOPAGE;
    };
    if (streaming && todo > 0) alloc_issue();
    int pn = 0;                                // lanes of the batch in flight (wave-uniform)
    int p_pl = 0;
    float p_w = 0.f, p_fx = 0.f, p_fy = 0.f, p_vd0 = 0.f, p_vd1 = 0.f, p_vd2 = 0.f;
    Texel3 p00 = {0.f, 0.f, 0.f}, p01 = p00, p10 = p00, p11 = p00;
    auto finish = [&]() {
        if (lane < pn) {
            float t0 = 0.f, t1 = 0.f, t2 = 0.f;
            if constexpr (TAPS) {
                const float w00 = (1.f - p_fx) * (1.f - p_fy), w01 = p_fx * (1.f - p_fy);
                const float w10 = (1.f - p_fx) * p_fy,         w11 = p_fx * p_fy;
                t0 = w00 * p00.x + w01 * p01.x + w10 * p10.x + w11 * p11.x;
                t1 = w00 * p00.y + w01 * p01.y + w10 * p10.y + w11 * p11.y;
                t2 = w00 * p00.z + w01 * p01.z + w10 * p10.z + w11 * p11.z;
            }
            // w * colour (>= 0) goes to the owning pixel's accumulator as Q32.32 fixed point with an INTEGER LDS atomic:
            // ds_add_f32 retires ~3 cycles per LANE on gfx950 (193 cycles per wave instruction, scripts/ubench/lds_atomics.hip),
            // ds_add_u64 6 cycles per instruction.  2^-32 resolution (45 items: < 1e-8), exact and order-independent below 2^31.
            ull* cp = L.col + p_pl * 3;
            atomicAdd(cp + 0, (ull)(fminf(p_w * fmaxf(0.f, TG_SH_C0 * t0 + p_vd0 + 0.5f), 2.0e9f) * 4294967296.0f));
            atomicAdd(cp + 1, (ull)(fminf(p_w * fmaxf(0.f, TG_SH_C0 * t1 + p_vd1 + 0.5f), 2.0e9f) * 4294967296.0f));
            atomicAdd(cp + 2, (ull)(fminf(p_w * fmaxf(0.f, TG_SH_C0 * t2 + p_vd2 + 0.5f), 2.0e9f) * 4294967296.0f));
        }
        pn = 0;
    };
    auto drain = [&](int n_) {
        finish();
        uint2 e_ = make_uint2(0u, 0u);
        if (lane < n_) e_ = L.q[(qhead + lane) & (FQ_CAP - 1)];
        const int jj_ = KEY_J(e_.y);
        const float4 f_ = L.p.F[jj_];
        const float2 g2 = L.p.G[jj_];
        // The queue carries T (what K7 needs), not w: alpha_raw is evaluated again here from the same planes with the same
        // explicitly ordered operations as in the test loop -- bit-identical -- and w = min(0.99, alpha_raw) * T as there.
        const float4 a4_ = L.p.A[jj_];
        const float2 b2_ = *reinterpret_cast<const float2*>(&L.p.B[jj_]);
        const float ipx_ = (float)(wave_px + KEY_OX(e_.y)), ipy_ = (float)(wave_py + KEY_OY(e_.y));
        const float araw_ = gauss_alpha_raw(b2_.y, gauss_power(a4_.z, a4_.w, b2_.x, a4_.x - ipx_, a4_.y - ipy_));
        const float T_ = __uint_as_float(e_.x);
        const float w_ = fminf(TG_ALPHA_MAX, araw_) * T_;
        if (streaming) {
            if (cur_page == TG_PAGE_UNSET) { cur_page = alloc_take(TG_NOPAGE); alloc_issue(); }
            const bool cross = ((nitems + n_) >> TG_PAGE_SHIFT) != (nitems >> TG_PAGE_SHIFT);
            uint32_t nxt = TG_NOPAGE;
            if (cross) { nxt = alloc_take(cur_page); alloc_issue(); }
            const uint32_t v = (uint32_t)nitems + (uint32_t)lane;
            const uint32_t page = ((v >> TG_PAGE_SHIFT) == ((uint32_t)nitems >> TG_PAGE_SHIFT)) ? cur_page : nxt;
            const uint32_t gid = (uint32_t)__builtin_amdgcn_ds_bpermute(jj_ << 2, (int)cid);
            if (lane < n_ && page != TG_NOPAGE) {
                uint32_t* __restrict__ pb = a.item_pages + (size_t)page * (3 * TG_PAGE) + (v & (TG_PAGE - 1));
                nt_store(pb, e_.x); nt_store(pb + TG_PAGE, __float_as_uint(araw_)); nt_store(pb + 2 * TG_PAGE, (gid << 6) | (uint32_t)KEY_PL(e_.y));
            }
            if (cross) { prev_page = cur_page; cur_page = nxt; }
            nitems += n_;
        }
        if constexpr (TAPS) {
            const float2 xy = make_float2(a4_.x, a4_.y);
            const float4 d_ = L.p.D[jj_], e4 = L.p.E[jj_];
            const float dpx = ipx_ - xy.x, dpy = ipy_ - xy.y;
            const float den = 1.0f + d_.x * dpx + d_.y * dpy;
            const float inv = (den >= TG_DEN_MIN) ? __builtin_amdgcn_rcpf(den) : 0.0f;
            const float u0 = f_.x + (d_.z * dpx + d_.w * dpy) * inv;
            const float u1 = f_.y + (e4.x * dpx + e4.y * dpy) * inv;
            const float u2 = f_.z + (e4.z * dpx + e4.w * dpy) * inv;
            const CubeTap ct = cube_address(u0, u1, u2, a.R);
            // every lane loads: a lane without an item decoded entry (0, 0) -> survivor slot 0 of the chunk (a real record or
            // zeros), whose address is as valid as any (clamped coordinates); predicating the loads cost 12 register clears
            p00 = load_texel(tex, ct.o00); p01 = load_texel(tex, ct.o00 + ct.dox);
            p10 = load_texel(tex, ct.o00 + ct.doy); p11 = load_texel(tex, ct.o00 + ct.doy + ct.dox);
            p_fx = ct.fx; p_fy = ct.fy;
            if (bin_count != nullptr) {
                // a backward will follow: count this round's footprints per texture bin in the block's reservation table (sizes of
                // K7's record lists).  Hit: ONE integer LDS atomic per lane.  Otherwise the lanes are grouped by bin with ballots:
                // the first bin to arrive at a free entry claims it (a dozen times per block); a bin whose entry belongs to another
                // one is counted on the global overflow counter, one atomic per (round, bin) -- 2 % of the footprints.
                const bool binned = (lane < n_) && tap_binned(ct);
                const uint32_t bin = tap_bin(ct, nbins_row);
                const int home = tap_home(ct);
                const bool hit = binned && resv_bin(L.p, home) == bin;
                if (hit) atomicAdd(&L.ccnt[home >> 1], 1u << ((home & 1) << 4));
                ull pend = TG_BALLOT(binned) & ~TG_BALLOT(hit);
                while (pend != 0ull) {
                    const int l0 = __ffsll((long long)pend) - 1;
                    const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)bin, l0);
                    const int h0 = __builtin_amdgcn_readlane(home, l0);
                    const ull m = pend & TG_BALLOT(bin == b0);
                    const uint32_t e0 = resv_bin(L.p, h0);          // (wave-uniform address: a broadcast read)
                    if (lane == l0) {
                        const uint32_t n = (uint32_t)__popcll(m);
                        if (e0 == TG_RESV_EMPTY) { resv_bin(L.p, h0) = b0; atomicAdd(&L.ccnt[h0 >> 1], n << ((h0 & 1) << 4)); }
                        else atomicAdd(bin_count + 6 * nbins_row * nbins_row + b0, n);
                    }
                    __builtin_amdgcn_wave_barrier();                // (the next group's read must see a claimed entry)
                    pend &= ~m;
                }
            }
        }
        p_w = w_; p_pl = KEY_PL(e_.y); p_vd0 = f_.w; p_vd1 = g2.x; p_vd2 = g2.y;
        pn = n_;
    };

    const size_t sbase = 4 * (size_t)range.x + (size_t)wave * (size_t)todo;       // this block's survivor entries (K6 -> K7)
    int nsurv = 0;
    int r = 0, nq = 0, qh = 0;                 // next raw list position, survivors queued, queue head (all wave-uniform)
    // the raw batch at r is loaded ONE BATCH AHEAD (ids two ahead): its test records are in flight while the previous batch is culled
    // and while a chunk is blended
    const float4 zt0 = make_float4(0.f, 0.f, 0.f, 0.f), zt1 = make_float4(0.f, 0.f, -1.f, 1.f);      // rcull < 0: never reachable
    float4 nt0 = zt0, nt1 = zt1;
    if (lane < todo) {
        const float4* __restrict__ tp = a.rec_test + 2 * (size_t)a.point_list[range.x + lane];
        nt0 = tp[0]; nt1 = tp[1];
    }
    uint32_t nid = (64 + lane < todo) ? a.point_list[range.x + 64 + lane] : 0u;      // Gaussian ids of the batch at r + 64
    bool all_done = (~done_mask == 0ull);
    while (!all_done) {
        // ---- raw batches: 8x8 cull on the 32-byte test records, survivors queued in list order
        while (nq < 64 && r < todo) {
            const int idx = r + lane;
            const float4 t0 = nt0, t1 = nt1;
            r += 64;
            nt0 = zt0; nt1 = zt1;
            if (r + lane < todo) {
                const float4* __restrict__ tp = a.rec_test + 2 * (size_t)nid;
                nt0 = tp[0]; nt1 = tp[1];
            }
            nid = (r + 64 + lane < todo) ? a.point_list[range.x + r + 64 + lane] : 0u;
            const bool reach = block_reachable(t0.x, t0.y, t0.z, t0.w, t1.x, t1.w, t1.z, wpx, wpy, 7.0f);
            const ull m = TG_BALLOT(reach);
            if (reach) L.ring[(qh + nq + mbcnt64(m)) & (TG_RING - 1)] = (uint32_t)idx;
            nq += __popcll(m);
        }
        if (nq == 0) break;
        // ---- chunk: up to 64 survivors, lane = survivor
        const int take = min(64, nq);
        __builtin_amdgcn_wave_barrier();
        uint32_t pos = 0xFFFFFFFFu, id = 0u;
        if (lane < take) {
            pos = L.ring[(qh + lane) & (TG_RING - 1)];
            id = a.point_list[range.x + pos];
        }
        qh = (qh + take) & (TG_RING - 1); nq -= take;
        float4 T0, T1;
        cid = id;
        load_chunk(a, L.p, lane, lane < take, id, pos, T0, T1);     // (every item of the previous chunk was started by a drain: its fields are in registers)
        reinterpret_cast<uint32_t*>(&L.list[0][0])[lane] = 0x40404040u;      // pad all four lists with TG_DUMMY
        __builtin_amdgcn_wave_barrier();
        int len[4];
        uint32_t qbits = 0u;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool rowdone = ((done_mask >> (16 * q)) & 0xFFFFull) == 0xFFFFull;
            const bool geo = (lane < take) &&
                block_reachable(T0.x, T0.y, T0.z, T0.w, T1.x, T1.w, T1.z, wpx + (float)((q & 1) << 2), wpy + (float)((q >> 1) << 2), 3.0f);
            qbits |= geo ? (1u << q) : 0u;
            const bool rq = geo && !rowdone;
            const ull m = TG_BALLOT(rq);
            len[q] = __popcll(m);
            if (rq) L.list[q][mbcnt64(m)] = (uint8_t)lane;
        }
        if (a.surv != nullptr) {                // a backward will follow: it replays exactly these survivors, back to front
            if (lane < take) { nt_store2(a.surv + sbase + nsurv + lane, make_uint2(id, pos)); nt_store(a.surv_qm + sbase + nsurv + lane, (uint16_t)qbits); }
            nsurv += take;
        }
        __builtin_amdgcn_wave_barrier();
        // ---- lock-step test loop: iteration t tests list entry t of every quadrant.  Parameters of iteration t + 1 and the
        // list entry of t + 2 are in flight while t is evaluated.
        int tmax = max(max(len[0], len[1]), max(len[2], len[3]));
        // one tested list entry per quadrant; returns true when every pixel of the block is done
        auto test_one = [&](const float4& A, const float4& B, const float4& Cc, const int j) -> bool {
                // One exit per tested instance: alpha is evaluated for every candidate that survived the culls.  Wave-level decisions
                // are PRODUCTS of single-compare ballots: a ballot of one compare is the v_cmp's own lane mask and the combination is
                // scalar ALU; a ballot of a compound predicate costs a v_cndmask + v_cmp round trip.
                const float power = gauss_power(A.z, A.w, B.x, A.x - pxf, A.y - pyf);
                const float alpha = fminf(TG_ALPHA_MAX, gauss_alpha_raw(B.y, power));
                const ull m_ok = TG_BALLOT(power <= 0.0f) & ~done_mask & TG_BALLOT(alpha >= TG_ALPHA_MIN);
                if (m_ok == 0ull) return false;
                bool ok = (!done) && (power <= 0.0f) && (alpha >= TG_ALPHA_MIN);
                const float Tn = T * (1.0f - alpha);
                const ull m_kill = m_ok & TG_BALLOT(Tn < TG_T_EPS);
                if (ok && Tn < TG_T_EPS) { done = true; ok = false; }
                done_mask |= m_kill;
                const ull bal = m_ok & ~m_kill;
                if (bal != 0ull) {
                    if (ok) {
                        const float w = alpha * T;
                        Dp += w * Cc.x; N0 += w * Cc.y; N1 += w * Cc.z; N2 += w * Cc.w; Al += w;
                        L.q[(qtail + mbcnt64(bal)) & (FQ_CAP - 1)] = make_uint2(__float_as_uint(T), keybase | (uint32_t)j);      // T BEFORE this pair
                        T = Tn;
                        last = __float_as_uint(B.z) + 1u;
                    }
                    qtail += __popcll(bal);
                    if (qtail - qhead >= 64) {
                        __builtin_amdgcn_wave_barrier();
                        drain(64);
                        qhead += 64;
                    }
                }
                if (m_kill != 0ull) {               // some pixels finished: quadrants whose 16 pixels are all done stop walking their lists
                    if (~done_mask == 0ull) { all_done = true; return true; }
                    int tm = 0;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (((done_mask >> (16 * q)) & 0xFFFFull) != 0xFFFFull) tm = max(tm, len[q]);
                    tmax = tm;
                }
            return false;
        };
        // Two register sets, used and refilled alternately: set X holds the even entries, set Y the odd ones; while one is evaluated
        // the other is loaded (one iteration ahead), the list index two ahead.  (One set rotated through a "next" copy cost eight
        // 64-bit register moves per iteration: a fifth of the loop's VALU instructions.)
        int jx = mylist[0], jy = mylist[1];
        float4 xA = L.p.A[jx], xB = L.p.B[jx], xC = L.p.C[jx];
        float4 yA = xA, yB = xB, yC = xC;
        for (int t = 0; t < tmax; t += 2) {
            yA = L.p.A[jy]; yB = L.p.B[jy]; yC = L.p.C[jy];
            const int j0 = jx;
            jx = mylist[min(t + 2, 63)];
            if (test_one(xA, xB, xC, j0)) break;
            if (t + 1 >= tmax) break;
            xA = L.p.A[jx]; xB = L.p.B[jx]; xC = L.p.C[jx];
            const int j1 = jy;
            jy = mylist[min(t + 3, 63)];
            if (test_one(yA, yB, yC, j1)) break;
        }
        // items reference this chunk's LDS planes: start them before the next chunk is loaded
        if (qtail - qhead > 0) {
            __builtin_amdgcn_wave_barrier();
            drain(qtail - qhead);
            qhead = qtail;
        }
        __builtin_amdgcn_wave_barrier();
    }
    finish();
    __builtin_amdgcn_wave_barrier();
    if (TAPS && bin_count != nullptr) {
        // reserve: this block's footprints of bin b occupy [off, off + n) of b's record list (offsets inside the list: the lists'
        // bases are known only after k_bin_offsets has scanned the totals this very atomic builds); lane = table entry
        const uint32_t b = resv_bin(L.p, lane), n = (L.ccnt[lane >> 1] >> ((lane & 1) << 4)) & 0xFFFFu;
        uint32_t off = 0u;
        if (b != TG_RESV_EMPTY) off = atomicAdd(bin_count + b, n);
        uint32_t* __restrict__ rv = a.resv + (size_t)(4 * tile + wave) * (3 * TG_RESV);
        nt_store(rv + lane, b); nt_store(rv + TG_RESV + lane, off); nt_store(rv + 2 * TG_RESV + lane, n);
    }
    if (a.surv_cnt != nullptr && lane == 0) a.surv_cnt[4 * tile + wave] = (uint32_t)nsurv;
    if (streaming && lane == 0) {
        // the page that holds the block's LAST item (a stream that ends exactly on a page boundary has already moved on)
        const uint32_t tailp = (nitems > 0 && (nitems & (TG_PAGE - 1)) == 0) ? prev_page : cur_page;
        reinterpret_cast<uint2*>(a.item_tail)[4 * tile + wave] = make_uint2(tailp, (uint32_t)nitems);
    }
    if (inside) {
        const int HW = a.W * a.H, pix = py * a.W + px;
        const double q = 1.0 / 4294967296.0;
        nt_store(out_color + pix, (float)((double)L.col[lane * 3 + 0] * q) + T * a.bg[0]);
        nt_store(out_color + HW + pix, (float)((double)L.col[lane * 3 + 1] * q) + T * a.bg[1]);
        nt_store(out_color + 2 * HW + pix, (float)((double)L.col[lane * 3 + 2] * q) + T * a.bg[2]);
        nt_store(out_depth + pix, Dp);
        nt_store(out_norm + pix, N0); nt_store(out_norm + HW + pix, N1); nt_store(out_norm + 2 * HW + pix, N2);
        nt_store(out_alpha + pix, Al);
        nt_store(final_T + pix, T);
        nt_store(n_contrib + pix, last);
    }
}

// ------------------------------------------------------------------------------------------------ K7
// Backward replay, back to front, over the survivor lists K6 left for every 8x8 block (no culling here: K6 culled exactly these
// lists).  Per chunk of 64 survivors, in segments of at most BQ_CAP items / BWD_MAX_IT iterations:
//   stage A  lock-step over the four quadrant lists (see the file header), ~45 VALU per iteration: falloff, alpha,
//            T /= (1 - alpha); contributing (pixel, Gaussian) pairs are compacted (ballot + mbcnt) into the LDS item list
//            {T, -, alpha_raw, key}; the iteration's ballot and first item stay in the registers of lane <iteration>.
//   stage B  dense, 64 items per round: UV Taylor step, cubemap address, 4 dwordx3 tap loads, colour; stores per item
//            s = colour . dL/dcolour + geometry channels . their gradients (what stage C1 sums) and dL/dcolour (3),
//            dL/duv (3), 1/den, dL/dden; appends the item's texture-gradient record to its texture bin (see the file header).
//   stage C1 per pixel, iteration by iteration: dL/dalpha_i = T_i s_i - (sum of s_k alpha_k T_k behind i + bg term) / (1 - alpha_i),
//            one running sum per pixel; leaves w and dL/dpower in the item.
//   stage C2 dense over TASKS = the items of one (iteration, quadrant) = up to 16 consecutive items of ONE Gaussian, four
//            tasks per round (one per DPP row): the 28 per-Gaussian moment terms of every item, a 16-lane transposing butterfly
//            (DPP only, wave_ops.h), and the 16 lanes add the Gaussian's 128-byte accumulator row as two 64-byte runs.
#define BWD_MAX_IT 16
struct TexBinArgs {
    uint32_t* rec;         // [cap][4] 16-byte records (rec_pack); bin b owns [base[b], base[b+1])
    uint32_t* cursor;      // [nbins] next free record of each list's OVERFLOW part, absolute (k_bin_offsets sets it to the end of the reserved part)
    const uint32_t* base;  // [nbins + 1] exclusive scan of K6's per-bin counts
    const uint32_t* order; // [nbins] the reduce kernel's launch order: bins by falling list length (k_bin_offsets)
    uint32_t* stats;       // [0] max records a call wanted (for the host), [1] bits of max |dL/dpixel colour| of this call
    uint32_t  cap;         // records the buffer holds; what does not fit goes to dL_dtexture directly
    int       nb;          // bins per face row = ceil(R / 32)
};

// footprints that cannot be binned (clamped at a face border, beyond the buffer): straight into dL_dtexture.  Offsets in BYTES.
__device__ __forceinline__ void scatter_direct(float* __restrict__ dtex, uint32_t o00, uint32_t dox, uint32_t doy, float fx, float fy,
                                               float x0, float x1, float x2) {
    const float w00 = (1.f - fx) * (1.f - fy), w01 = fx * (1.f - fy), w10 = (1.f - fx) * fy, w11 = fx * fy;
    float* p00 = dtex + (o00 >> 2);
    float* p01 = dtex + ((o00 + dox) >> 2);
    float* p10 = dtex + ((o00 + doy) >> 2);
    float* p11 = dtex + ((o00 + doy + dox) >> 2);
    unsafeAtomicAdd(p00, w00 * x0); unsafeAtomicAdd(p00 + 1, w00 * x1); unsafeAtomicAdd(p00 + 2, w00 * x2);
    unsafeAtomicAdd(p01, w01 * x0); unsafeAtomicAdd(p01 + 1, w01 * x1); unsafeAtomicAdd(p01 + 2, w01 * x2);
    unsafeAtomicAdd(p10, w10 * x0); unsafeAtomicAdd(p10 + 1, w10 * x1); unsafeAtomicAdd(p10 + 2, w10 * x2);
    unsafeAtomicAdd(p11, w11 * x0); unsafeAtomicAdd(p11 + 1, w11 * x1); unsafeAtomicAdd(p11 + 2, w11 * x2);
}

// Texture-gradient RECORD, 16 bytes, one per bilinear footprint, written with ONE 16-byte store (round 6; until then 20 bytes as
// five planes: five scattered 4-byte stores per footprint, whose partially written lines the L2 had to give up before they were
// full -- ablation: K7 -90 us without the stores):
//   w0 = low 16 bits of fx18 | low 16 bits of fy18 << 16          fx18 = round(fx * 2^18), 18 bits (as the 20-byte record kept)
//   w1 = r, its 5 low mantissa bits replaced by the footprint's cell x inside its 32x32-texel bin (r rounded to 18 mantissa bits)
//   w2 = g, likewise with cell y          w3 = b, its 4 low mantissa bits = the two high bits of fx18 and of fy18
// inf / NaN survive the rounding (an inf / NaN upstream gradient still reaches exactly the texels it touches).
struct __attribute__((aligned(16))) Rec4 { uint32_t a, b, c, d; };
__device__ __forceinline__ void rec_store(uint32_t* __restrict__ base, uint32_t slot, const Rec4 r) {
    tg_u4 v; v.x = r.a; v.y = r.b; v.z = r.c; v.w = r.d;
    __builtin_nontemporal_store(v, reinterpret_cast<tg_u4*>(base) + slot);
}
__device__ __forceinline__ Rec4 rec_load(const Rec4* __restrict__ p) {
    const tg_u4 v = __builtin_nontemporal_load(reinterpret_cast<const tg_u4*>(p));
    Rec4 r; r.a = v.x; r.b = v.y; r.c = v.z; r.d = v.w;
    return r;
}
__device__ __forceinline__ uint32_t rec_word0(float fx, float fy, uint32_t& hi) {
    const uint32_t qx = min((uint32_t)(fx * 262144.0f + 0.5f), 262143u), qy = min((uint32_t)(fy * 262144.0f + 0.5f), 262143u);
    hi = (qx >> 16) | ((qy >> 16) << 2);
    return (qx & 0xFFFFu) | (qy << 16);
}
__device__ __forceinline__ Rec4 rec_pack(uint32_t w0, uint32_t hi, int cx, int cy, float x0, float x1, float x2) {
    Rec4 r;
    r.a = w0;
    r.b = ((__float_as_uint(x0) + 16u) & ~31u) | (uint32_t)(cx & 31);
    r.c = ((__float_as_uint(x1) + 16u) & ~31u) | (uint32_t)(cy & 31);
    r.d = ((__float_as_uint(x2) + 8u) & ~15u) | hi;
    return r;
}
struct RecVal { int cx, cy; float fx, fy, x0, x1, x2; };
__device__ __forceinline__ RecVal rec_unpack(const Rec4 w) {
    RecVal r;
    r.fx = (float)((w.a & 0xFFFFu) | ((w.d & 3u) << 16)) * (1.0f / 262144.0f);
    r.fy = (float)((w.a >> 16) | (((w.d >> 2) & 3u) << 16)) * (1.0f / 262144.0f);
    r.cx = (int)(w.b & 31u); r.cy = (int)(w.c & 31u);
    r.x0 = __uint_as_float(w.b & ~31u); r.x1 = __uint_as_float(w.c & ~31u); r.x2 = __uint_as_float(w.d & ~15u);
    return r;
}

// live accumulator-row slots (common.h M_*) of the two C2 flavours: all 28 moments / without the UV chain (M_DEN, M_DN, M_PHI)
#define TG_MOMENTS_ALL  0x0FFFFFFFu
#define TG_MOMENTS_NOUV (0x3Fu | (0x7Fu << 21))

// K7 is compiled in two configurations (csrc/render_bwd_body.h, measured on MI355X, profiles/r05_ablation.md):
//   occ  -- shading records gathered from global memory per item (L2 hits), segments of 64 items, registers held to 4 waves per
//           SIMD: 9.3 KB of LDS and ~106 VGPRs per wave -> 16 waves per CU (round 4: 17.6 KB, 144 VGPRs, 9 waves).  K7 698 -> 643-659 us,
//           C3 pipelined 856 -> 903-926 views/s.  Used by every flavour that runs the per-Gaussian stages.
//   lds  -- the round-4 shape (records in LDS planes, segments of 128 items, two rounds of taps in flight, 9 waves per CU): the
//           texture-only flavour has no stage C to hide a dependent L2 round trip behind and is faster this way (405 vs 512 us).
#ifdef K7_TRACE
// experiment builds only (scripts/exp_build.sh trace "-DK7_TRACE"; scripts/k7_trace.py): per block of the last K7 launch
// {start, end (100 MHz wall clock), XCC | HW_ID << 8, survivors | list length << 32}
#define K7_TRACE_BLOCKS 32768
__device__ unsigned long long k7_trace[4 * K7_TRACE_BLOCKS];
#endif
namespace k7_occ {
#define BQ_CAP 64
#define K7_GATHER 1
#define K7_WAVES_PER_SIMD 4
#include "render_bwd_body.h"
#undef BQ_CAP
#undef K7_GATHER
#undef K7_WAVES_PER_SIMD
}  // namespace k7_occ
namespace k7_lds {
#define BQ_CAP 128
#define K7_GATHER 0
#define K7_WAVES_PER_SIMD 2
#include "render_bwd_body.h"
#undef BQ_CAP
#undef K7_GATHER
#undef K7_WAVES_PER_SIMD
}  // namespace k7_lds
// K7 over K6's item stream (texgs.h v15): the flavours with the per-Gaussian stages.  Same registers-per-wave target as k7_occ.
#ifdef K7S_TRACE
#ifndef K7_TRACE_BLOCKS
#define K7_TRACE_BLOCKS 32768
#endif
__device__ unsigned long long k7s_trace[8 * K7_TRACE_BLOCKS];
#endif
#ifndef K7S_WAVES_PER_SIMD
#define K7S_WAVES_PER_SIMD 4
#endif
namespace k7_stream {
#include "render_bwd_stream.h"
}  // namespace k7_stream

// ------------------------------------------------------------------------------------------------ texture-gradient lists
// Exclusive scan of K6's per-bin footprint counts -> list offsets (one workgroup; nbins = 6144 at R = 1024).  count[b] = the
// footprints the blocks RESERVED in bin b's list (K7 adds base[b] to its blocks' reservations, which K6 made relative to the start
// of the list), count[nbins + b] = the overflow footprints, appended behind them through cursor[b] (absolute, starts at the end of
// the reserved part).
__global__ void __launch_bounds__(1024)
k_bin_offsets(int nbins, const uint32_t* __restrict__ count, uint32_t* __restrict__ base, uint32_t* __restrict__ cursor,
              uint32_t* __restrict__ order, uint32_t* __restrict__ stats) {
    // one pass: thread t owns the `per` consecutive counts [t * per, (t + 1) * per) -- serial inside the thread, one wave scan,
    // one cross-wave step (6 144 bins at R = 1024: 6 per thread).  Chunks of 1 024 x BO_MAX bins if there are more.
    constexpr int BO_MAX = 32;
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_carry;
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) s_carry = 0u;
    __syncthreads();
    for (int c0 = 0; c0 < nbins; c0 += 1024 * BO_MAX) {
        const int n = min(nbins - c0, 1024 * BO_MAX);
        const int per = (n + 1023) >> 10;
        const int i0 = c0 + tid * per;
        uint32_t v[BO_MAX];
        uint32_t sum = 0u;
#pragma unroll
        for (int k = 0; k < BO_MAX; ++k) { v[k] = (k < per && i0 + k < c0 + n) ? count[i0 + k] + count[nbins + i0 + k] : 0u; sum += v[k]; }
        uint32_t incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64); if (lane >= d) incl += o; }
        if (lane == 63) s_w[wv] = incl;
        __syncthreads();
        uint32_t run = s_carry + incl - sum;
        for (int w = 0; w < wv; ++w) run += s_w[w];
#pragma unroll
        for (int k = 0; k < BO_MAX; ++k) { if (k < per && i0 + k < c0 + n) { base[i0 + k] = run; cursor[i0 + k] = run + count[i0 + k]; } run += v[k]; }
        __syncthreads();
        if (tid == 1023) s_carry = run;
        __syncthreads();
    }
    if (tid == 0) {
        base[nbins] = s_carry;
        if (s_carry > stats[0]) stats[0] = s_carry;      // what the lists of this view need: the host sizes the buffer from it
        stats[1] = 0u;                                   // max |dL/dpixel colour| of THIS call: K7 raises it, the reduce reads it
    }
    // Launch order of the reduce kernel: longest list first (a C3 view: 6 144 lists, mean 3 000 records, the longest 33 000 --
    // in index order the launch ends on whichever long list started late), as a counting sort on a 1024-level log-ish length
    // class (speed only: any order is correct; ties keep no particular order).
    __shared__ uint32_t s_cnt[1024];
    auto length_class = [](uint32_t len) -> uint32_t {     // monotone decreasing in len: 1023 = empty, 0 = longest
        if (len == 0u) return 1023u;
        const uint32_t e = 31u - (uint32_t)__clz((int)len);
        const uint32_t frac = (e >= 5u) ? ((len >> (e - 5u)) & 31u) : ((len << (5u - e)) & 31u);
        return 1022u - min(e * 32u + frac, 1022u);
    };
    s_cnt[tid] = 0u;
    __syncthreads();
    for (int i = tid; i < nbins; i += 1024) atomicAdd(&s_cnt[length_class(count[i] + count[nbins + i])], 1u);
    __syncthreads();
    {
        const uint32_t c = s_cnt[tid];
        uint32_t incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64); if (lane >= d) incl += o; }
        if (lane == 63) s_w[wv] = incl;
        __syncthreads();
        uint32_t run = incl - c;
        for (int w = 0; w < wv; ++w) run += s_w[w];
        s_cnt[tid] = run;
    }
    __syncthreads();
    for (int i = tid; i < nbins; i += 1024) order[atomicAdd(&s_cnt[length_class(count[i] + count[nbins + i])], 1u)] = (uint32_t)i;
}

// One workgroup per 32x32-texel bin: sum the bin's records into a 33x33-texel LDS tile (footprints anchored in the bin
// reach one texel past its right / bottom edge, still inside the face), then add every non-zero texel of the tile to
// dL_dtexture[6,R,R,3] once -- 99 consecutive dwords per tile row, i.e. coalesced memory-side requests; neighbouring
// bins overlap in that one-texel seam, hence atomics.
// The tile is 64-bit FIXED POINT: LDS float atomics retire ~3 cycles per lane on gfx950 (ds_add_f32: 193 cycles per wave
// instruction, ds_add_u64: 6; scripts/ubench/lds_atomics.hip), which made the first version of this kernel 1.8 ms.  Scale:
// every record value is bounded by C0 * max|dL/dpixel colour| OF THIS CALL (k_bin_offsets clears stats[1], K7 raises it)
// and is mapped to < 2^42, so 2^20 records cannot overflow; a longer list is summed in segments of 2^20 records with the tile
// written out in between; resolution 2^-42 of the image-wide bound, sums exact and order-independent.
// A NON-FINITE bound (an inf / NaN upstream gradient, e.g. under a GradScaler overflow) has no fixed-point scale: the bin's
// records are then added with float atomics, which carry inf / NaN to exactly the texels the atomic path would have reached.
#define TB_EDGE 33
#define TB_ROW 37       // texels per tile row in LDS: the pad puts rows y and y + 1 (taps 00 / 10 of one record) 30 banks apart; measured 180 -> 162 us
#define TB_SEG (1u << 20)
#define TB_THREADS 512  // 256: the same; 1024: 185 us (two workgroups per CU)
__global__ void __launch_bounds__(TB_THREADS)
k_texgrad_reduce(int R, TexBinArgs tb, float* __restrict__ dtex) {
    __shared__ long long s_tile[TB_EDGE * TB_ROW * 3];           // [row][col (padded)][channel], 2^42-scaled fixed point
    const int b = (int)tb.order[blockIdx.x], tid = (int)threadIdx.x;
    const uint32_t b0 = tb.base[b], b1 = tb.base[b + 1];
    const uint32_t filled = tb.cursor[b] - b0;                   // the reserved part (every slot has an owner block, which fills it) + what K7
                                                               // appended behind it (the cursor is absolute and started at the end of the reserved part)
    if (filled == 0u) return;                                  // uniform per workgroup
    const uint32_t room = (b0 < tb.cap) ? min(b1, tb.cap) - b0 : 0u;      // records of this list that exist (K7's own test)
    const uint32_t cnt = min(filled, room);
    const Rec4* __restrict__ rp = reinterpret_cast<const Rec4*>(tb.rec) + b0;
    const int face = b / (tb.nb * tb.nb), by = (b / tb.nb) % tb.nb, bx = b % tb.nb;
    const uint32_t bbits = tb.stats[1];
    if (bbits >= 0x7F800000u) {                                // inf / NaN upstream gradient: float atomics, record by record
        for (uint32_t i = (uint32_t)tid; i < cnt; i += (uint32_t)TB_THREADS) {
            const RecVal r = rec_unpack(rec_load(rp + i));
            const uint32_t y = (uint32_t)by * 32u + (uint32_t)r.cy, x = (uint32_t)bx * 32u + (uint32_t)r.cx;
            const uint32_t o00 = ((((uint32_t)face * (uint32_t)R + y) * (uint32_t)R + x) * 3u) << 2;
            scatter_direct(dtex, o00, 12u, 12u * (uint32_t)R, r.fx, r.fy, r.x0, r.x1, r.x2);
        }
        return;
    }
    const float bound = TG_SH_C0 * __uint_as_float(bbits);
    int e = 0;
    (void)frexpf(bound, &e);                                   // bound < 2^e
    const double up = (double)ldexpf(1.0f, 42 - e);
    const float down = ldexpf(1.0f, e - 42);
    // float -> int64 without the 11-instruction generic conversion: |v| < 2^42, so v + 1.5 * 2^52 (exact in double) carries
    // round(v) in its mantissa; subtracting the bias as integers leaves the two's-complement value
    const double magic = 6755399441055744.0;
    const long long magic_bits = __double_as_longlong(magic);
    auto add_record = [&](const Rec4 w) {
        const RecVal r = rec_unpack(w);
        const float fx = r.fx, fy = r.fy;
        const double dx0 = (double)r.x0 * up, dx1 = (double)r.x1 * up, dx2 = (double)r.x2 * up;
        const double w00 = (double)((1.f - fx) * (1.f - fy)), w01 = (double)(fx * (1.f - fy));
        const double w10 = (double)((1.f - fx) * fy), w11 = (double)(fx * fy);
        ull* t = reinterpret_cast<ull*>(s_tile) + (r.cy * TB_ROW + r.cx) * 3;
#define TB_ADD(P, V) atomicAdd((P), (ull)(__double_as_longlong((V) + magic) - magic_bits))
        TB_ADD(t + 0, w00 * dx0); TB_ADD(t + 1, w00 * dx1); TB_ADD(t + 2, w00 * dx2);
        TB_ADD(t + 3, w01 * dx0); TB_ADD(t + 4, w01 * dx1); TB_ADD(t + 5, w01 * dx2);
        TB_ADD(t + TB_ROW * 3 + 0, w10 * dx0); TB_ADD(t + TB_ROW * 3 + 1, w10 * dx1); TB_ADD(t + TB_ROW * 3 + 2, w10 * dx2);
        TB_ADD(t + TB_ROW * 3 + 3, w11 * dx0); TB_ADD(t + TB_ROW * 3 + 4, w11 * dx1); TB_ADD(t + TB_ROW * 3 + 5, w11 * dx2);
#undef TB_ADD
    };
    for (uint32_t s0 = 0u; s0 < cnt; s0 += TB_SEG) {             // one trip unless the list is longer than 2^20 records
        const uint32_t s1 = min(cnt, s0 + TB_SEG);
        for (int k = tid; k < TB_EDGE * TB_ROW * 3; k += TB_THREADS) s_tile[k] = 0ll;
        __syncthreads();
        // two records per thread in flight.  (More in flight, the next trip's loads issued ahead of the LDS atomics, lanes spread over
        // distinct cells: no change.)
        uint32_t i = s0 + (uint32_t)tid;
        for (; i + (uint32_t)TB_THREADS < s1; i += 2u * (uint32_t)TB_THREADS) {
            const Rec4 ra = rec_load(rp + i), rb = rec_load(rp + i + (uint32_t)TB_THREADS);
            add_record(ra);
            add_record(rb);
        }
        if (i < s1) add_record(rec_load(rp + i));
        __syncthreads();
        for (int k = tid; k < TB_EDGE * TB_EDGE * 3; k += TB_THREADS) {
            const int row = k / (TB_EDGE * 3), c = k - row * (TB_EDGE * 3);
            const long long q = s_tile[row * (TB_ROW * 3) + c];
            if (q == 0ll) continue;
            const int y = by * 32 + row, xq = bx * 96 + c;
            if (y >= R || xq >= R * 3) continue;
            // rows / columns 0 and 32 of the tile are shared with the neighbouring bins' tiles, hence atomics
            unsafeAtomicAdd(dtex + ((size_t)(face * R + y) * R) * 3 + xq, (float)q * down);
        }
        __syncthreads();
    }
}

inline PixArgs make_pix(const CamConst& c, const TexGSFrame* f, const TexGSInputs* in, const TexGSGeom* g,
                        const TexGSBinning* b, const TexGSImage* img) {
    PixArgs a;
    a.W = c.W; a.H = c.H; a.tiles_x = c.tiles_x; a.num_tiles = c.tiles_x * c.tiles_y; a.R = c.R;
    a.ranges = reinterpret_cast<const uint2*>(b->ranges);
    a.point_list = b->point_list;
    a.tile_order = b->tile_order;
    a.rec_test = reinterpret_cast<const float4*>(g->rec_test);
    a.rec_shade = reinterpret_cast<const float4*>(g->rec_shade);
    a.texture = in->texture;
    a.bg = f->bg;
    const bool hand = img->survivors != nullptr && img->surv_qmask != nullptr && img->surv_count != nullptr;
    a.surv = hand ? reinterpret_cast<uint2*>(img->survivors) : nullptr;
    a.surv_qm = hand ? img->surv_qmask : nullptr;
    a.surv_cnt = hand ? img->surv_count : nullptr;
    a.resv = img->tex_bin_resv;
    const bool items = hand && img->item_pages != nullptr && img->item_link != nullptr && img->item_tail != nullptr &&
                       img->item_ctl != nullptr && img->item_sub_pools != 0u;
    a.item_pages = items ? img->item_pages : nullptr;
    a.item_link = items ? img->item_link : nullptr;
    a.item_tail = items ? img->item_tail : nullptr;
    a.item_ctl = items ? img->item_ctl : nullptr;
    a.item_sub_cap = items ? img->item_page_cap / img->item_sub_pools : 0u;
    a.item_sub_mask = items ? img->item_sub_pools - 1u : 0u;
    a.run_if = nullptr;
    return a;
}

inline TexBinArgs make_bins(const CamConst& c, const TexGSImage* img, const TexGSGrads* gr) {
    TexBinArgs tb;
    tb.nb = (c.R + 31) >> 5;
    const bool on = (gr->want & TEXGS_WANT_TEXTURE) && img->tex_bin_count != nullptr && img->tex_bin_resv != nullptr && gr->tex_bins != nullptr &&
                    gr->tex_bin_cursor != nullptr && gr->tex_bin_base != nullptr && gr->tex_rec_cap > 0;
    tb.rec = on ? reinterpret_cast<uint32_t*>(gr->tex_bins) : nullptr;
    tb.cursor = on ? gr->tex_bin_cursor : nullptr;
    tb.base = on ? gr->tex_bin_base : nullptr;
    tb.order = on ? gr->tex_bin_base + tex_bin_count(c.R) + 1 : nullptr;
    tb.stats = on ? gr->tex_bin_cursor + tex_bin_count(c.R) : nullptr;
    tb.cap = on ? gr->tex_rec_cap : 0u;
    return tb;
}

}  // namespace

size_t tex_bin_count(int R) {
    const size_t nb = (size_t)((R + 31) >> 5);
    return 6 * nb * nb;
}

void launch_render_fwd(const CamConst& c, const TexGSFrame* f, const TexGSInputs* in, const TexGSGeom* g,
                       const TexGSBinning* b, TexGSImage* img, hipStream_t s) {
    const PixArgs a = make_pix(c, f, in, g, b, img);
    if (in->texture)
        hipLaunchKernelGGL(k_render_fwd<true>, dim3(blend_grid(a.num_tiles)), dim3(64), 0, s, a, img->out_color, img->out_depth,
                           img->out_norm, img->out_alpha, img->final_T, img->n_contrib, img->tex_bin_resv ? img->tex_bin_count : (uint32_t*)nullptr);
    else
        hipLaunchKernelGGL(k_render_fwd<false>, dim3(blend_grid(a.num_tiles)), dim3(64), 0, s, a, img->out_color, img->out_depth,
                           img->out_norm, img->out_alpha, img->final_T, img->n_contrib, (uint32_t*)nullptr);
}

// Which flavour of K7 a call gets (see the template's comment): the texture gradient only when it is wanted and there is a texture,
// the per-Gaussian stages only when a Gaussian gradient is wanted; the UV chain rides with the per-Gaussian stages of the textured
// operator.
void launch_render_bwd(const CamConst& c, const TexGSFrame* f, const TexGSInputs* in, const TexGSGeom* g,
                       const TexGSBinning* b, const TexGSImage* img, TexGSGrads* gr, hipStream_t s) {
    const PixArgs a0 = make_pix(c, f, in, g, b, img);
    const TexBinArgs tb = make_bins(c, img, gr);
    const bool taps = in->texture != nullptr;
    const bool tex = taps && (gr->want & TEXGS_WANT_TEXTURE) && gr->dL_dtexture != nullptr;
    const bool geo = (gr->want & TEXGS_WANT_GAUSSIANS) != 0;
    if (!tex && !geo) return;
    if (tex && tb.rec)      // list offsets + cursors from the counts the forward left (one small workgroup)
        hipLaunchKernelGGL(k_bin_offsets, dim3(1), dim3(1024), 0, s, (int)tex_bin_count(c.R), (const uint32_t*)img->tex_bin_count,
                           gr->tex_bin_base, gr->tex_bin_cursor, gr->tex_bin_base + tex_bin_count(c.R) + 1, tb.stats);
    const dim3 grid(blend_grid(a0.num_tiles)), blk(64);
    PixArgs a = a0;
    if (geo && a.item_pages != nullptr) {
        // the forward left its item stream: the dense kernel over it; the survivor-replay kernel behind it runs only if K6 ran out of
        // pages (one word decides for the whole view; ~10^4 empty workgroups otherwise)
#define K7S_LAUNCH(TEX, UVG, TAPS) hipLaunchKernelGGL((k7_stream::k_render_bwd_stream<TEX, UVG, TAPS>), grid, blk, 0, s, a, tb, img->final_T, \
        gr->dL_dcolor, gr->dL_ddepth, gr->dL_dnorm, gr->dL_dalpha, gr->acc, gr->dL_dtexture)
        if (!taps)      K7S_LAUNCH(false, false, false);
        else if (tex)   K7S_LAUNCH(true, true, true);
        else            K7S_LAUNCH(false, true, true);
#undef K7S_LAUNCH
        a.run_if = a.item_ctl + TEXGS_ITEM_CTL_FLAG;
    }
#define K7_LAUNCH(NS, TEX, GEO, UVG, TAPS) hipLaunchKernelGGL((NS::k_render_bwd<TEX, GEO, UVG, TAPS>), grid, blk, 0, s, a, tb, img->final_T, \
        img->n_contrib, gr->dL_dcolor, gr->dL_ddepth, gr->dL_dnorm, gr->dL_dalpha, gr->acc, gr->dL_dtexture)
    if (!taps)            K7_LAUNCH(k7_occ, false, true, false, false);
    else if (tex && geo)  K7_LAUNCH(k7_occ, true, true, true, true);
    else if (tex)         K7_LAUNCH(k7_lds, true, false, false, true);
    else                  K7_LAUNCH(k7_occ, false, true, true, true);
#undef K7_LAUNCH
}

bool tex_bins_enabled(const CamConst& c, const TexGSInputs* in, const TexGSImage* img, const TexGSGrads* gr) {
    return in->texture != nullptr && gr->dL_dtexture != nullptr && make_bins(c, img, gr).rec != nullptr;
}

void launch_texgrad_reduce(const CamConst& c, const TexGSImage* img, TexGSGrads* gr, hipStream_t s) {
    const TexBinArgs tb = make_bins(c, img, gr);
    if (!tb.rec) return;
    hipLaunchKernelGGL(k_texgrad_reduce, dim3((unsigned)tex_bin_count(c.R)), dim3(TB_THREADS), 0, s, c.R, tb, gr->dL_dtexture);
}

#ifdef K7S_TRACE
extern "C" __attribute__((visibility("default"))) int texgs_debug_k7s_trace(void* host_dst) {
    return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(k7s_trace), sizeof(unsigned long long) * 8 * K7_TRACE_BLOCKS);
}
#endif
#ifdef K7_TRACE
extern "C" __attribute__((visibility("default"))) int texgs_debug_k7_trace(void* host_dst) {
    return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(k7_trace), sizeof(unsigned long long) * 4 * K7_TRACE_BLOCKS);
}
#endif
