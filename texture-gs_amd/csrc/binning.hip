// K2-K5: tile binning -- hand-written radix sorts, scan, duplicate-with-keys, tile ranges, tile launch order.  gfx950.
//
// Bit-exact contract: the tile lists are the (tile, depth-bits, Gaussian-index)-lexicographic
// order of all (Gaussian, touched tile) instances; keys_sorted[k] = (tile << 32) | bits(depth), point_list[k] = index.
// The lineage gets there with ONE stable LSD sort of D 64-bit keys over 32 + log2(T) bits (6 onesweep passes with
// decoupled look-back; rocPRIM's took 210 us at C3 -- every look-back hop is a ~1 us cross-XCD round trip here).
// This file sorts in two levels instead:
//   1. the N GAUSSIANS by depth bits (4 x 8-bit stable passes over 8-byte pairs; culled ones carry 0xFFFFFFFF),
//   2. an exclusive scan of tiles_touched in that depth-rank order,
//   3. K3 emits the instances in rank order, element = (tile << 32) | rank -- already sorted by rank,
//   4. the D INSTANCES by tile only (2 stable passes over ceil(log2 T) bits); the last pass writes keys_sorted /
//      point_list from the rank-ordered depth / index arrays.
// Stable passes in this order give exactly the order above (ties in depth keep index order, as the lineage's do).
// Every pass is two launches, no spinning: k_radix_count (per-block digit histograms in LDS, integer LDS atomics are
// full rate on gfx950) and k_radix_scatter (hierarchical prefix over the count table, stable ballot-match ranking,
// scatter).  All integer / byte work, HBM- and launch-latency-bound; no MFMA.
#include "common.h"
#include <cstring>

namespace {

constexpr int RS_THREADS = 256;         // 4 waves
constexpr int RS_MAX_BLOCKS = 1024;     // count-table rows per pass
constexpr int RS_GROUP = 32;            // rows per group partial
// elements per block, lower bound: a block is one dependent chain (loads -> table prefix -> 64-element rows ranked one after the
// other), so at the sizes of this path smaller blocks finish sooner: 2048 / 1024 / 512 / 256 -> K2 + K4 = 92 / 78 / 73 / 75 us at C2
#ifndef RS_PER_MIN
#define RS_PER_MIN 512u
#endif

struct RadixTables {                    // one per pass
    uint32_t* table;                    // [RS_MAX_BLOCKS][256] per-block digit counts (written, not accumulated)
    uint32_t* gtable;                   // [RS_MAX_BLOCKS / RS_GROUP][256] group partial sums (atomics: zero before the pass)
};                                      // (digit totals = column sums of gtable, formed by the scatter kernel: a `total[256]` kept by
                                        //  atomics took one atomic per (block, digit) on 256 hot words -- 548 per word in K4 at C3)
constexpr size_t RS_ZERO_WORDS = (size_t)(RS_MAX_BLOCKS / RS_GROUP) * 256;           // gtable of one pass

__device__ __forceinline__ uint32_t digit_of(uint32_t k, int shift, uint32_t mask) { return (k >> shift) & mask; }
__device__ __forceinline__ uint32_t digit_of(uint64_t k, int shift, uint32_t mask) { return (uint32_t)(k >> shift) & mask; }

// contiguous element range of block b: [b * per, min(n, (b + 1) * per)), per a multiple of 64
__device__ __forceinline__ void block_range(uint32_t n, uint32_t per, uint32_t& lo, uint32_t& hi) {
    lo = min(n, blockIdx.x * per);
    hi = min(n, lo + per);
}

template <typename KEY>
__global__ void __launch_bounds__(RS_THREADS)
k_radix_count(const KEY* __restrict__ keys, const uint32_t* __restrict__ n_ptr, uint32_t n_max, uint32_t per, int shift,
              uint32_t mask, RadixTables t) {
    __shared__ uint32_t s_hist[256];
    const uint32_t n = n_ptr ? min(*n_ptr, n_max) : n_max;
    uint32_t lo, hi;
    block_range(n, per, lo, hi);
    s_hist[threadIdx.x] = 0u;
    __syncthreads();
    // 8 keys per thread per trip, all loads in flight together (the kernel is one dependent chain: its time is its round trips)
    constexpr int CB = 8;
    for (uint32_t b0 = lo + threadIdx.x; b0 < hi; b0 += RS_THREADS * CB) {
        KEY kk[CB];
#pragma unroll
        for (int u = 0; u < CB; ++u) { const uint32_t i = b0 + (uint32_t)u * RS_THREADS; kk[u] = (i < hi) ? keys[i] : (KEY)0; }
#pragma unroll
        for (int u = 0; u < CB; ++u) if (b0 + (uint32_t)u * RS_THREADS < hi) atomicAdd(&s_hist[digit_of(kk[u], shift, mask)], 1u);
    }
    __syncthreads();
    const uint32_t c = s_hist[threadIdx.x];
    t.table[blockIdx.x * 256 + threadIdx.x] = c;
    if (c != 0u) atomicAdd(&t.gtable[(blockIdx.x / RS_GROUP) * 256 + threadIdx.x], c);
}

// MODE 0: pairs (u32 key, u32 val) -> pairs; val_in == nullptr means val = element index (first depth pass)
// MODE 1: u64 elements -> u64 elements
// MODE 2: u64 elements (tile << 32 | rank) -> keys_sorted = (tile << 32) | depth_rank[rank], point_list = id_rank[rank]
template <typename KEY, int MODE>
__global__ void __launch_bounds__(RS_THREADS)
k_radix_scatter(const KEY* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, KEY* __restrict__ keys_out,
                uint32_t* __restrict__ vals_out, const uint32_t* __restrict__ n_ptr, uint32_t n_max, uint32_t per, int shift,
                uint32_t mask, int bits, RadixTables t, const uint32_t* __restrict__ depth_rank,
                const uint32_t* __restrict__ id_rank) {
    __shared__ uint32_t s_wcnt[4][256];          // per-wave digit counts, then running output cursors
    __shared__ uint32_t s_scan[256];
    const uint32_t n = n_ptr ? min(*n_ptr, n_max) : n_max;
    uint32_t lo, hi;
    block_range(n, per, lo, hi);
    if (lo >= hi) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // wave w owns the w-th quarter of the block's range, in 64-element rows; RB rows (keys + values) are loaded together.  With
    // per = 2048 that is the wave's whole quarter: the keys are read ONCE, and their loads are in flight together with the
    // count-table loads below (the kernel is one dependent chain: its time is its round trips).
    constexpr int RB = 8;
    const uint32_t rows = (hi - lo + 63u) >> 6, rows_per_wave = (rows + 3u) >> 2;
    const uint32_t wlo = min(hi, lo + (uint32_t)wv * rows_per_wave * 64u), whi = min(hi, wlo + rows_per_wave * 64u);
    const bool single = (whi - wlo) <= 64u * RB;
    KEY kk[RB];
    uint32_t vv[RB];
    auto load_batch = [&](uint32_t b0) {
#pragma unroll
        for (int rr = 0; rr < RB; ++rr) {
            const uint32_t i = b0 + 64u * rr + lane;
            kk[rr] = (i < whi) ? keys_in[i] : (KEY)0;
            vv[rr] = (MODE == 0 && vals_in != nullptr && i < whi) ? vals_in[i] : i;
        }
    };
    load_batch(wlo);
    // ---- where this block's digit-d elements start: digit base + blocks before this one
    uint32_t before = 0u;
    {
        // fixed trip counts, predicated: all (at most 32 + 31) loads are in flight together instead of one round trip each.
        // Every group row is read: the rows before this block's group count towards `before`, all of them towards the digit total.
        static_assert(RS_MAX_BLOCKS / RS_GROUP <= RS_GROUP, "one loop covers the group rows and the rows inside a group");
        const int g = blockIdx.x / RS_GROUP, inb = (int)blockIdx.x - g * RS_GROUP;
        const int ngroups = ((int)gridDim.x + RS_GROUP - 1) / RS_GROUP;
        uint32_t part[2 * RS_GROUP];
#pragma unroll
        for (int k = 0; k < RS_GROUP; ++k) {
            part[k] = (k < ngroups) ? t.gtable[k * 256 + tid] : 0u;
            part[RS_GROUP + k] = (k < inb) ? t.table[(g * RS_GROUP + k) * 256 + tid] : 0u;
        }
        uint32_t tot = 0u;
#pragma unroll
        for (int k = 0; k < RS_GROUP; ++k) { tot += part[k]; before += (k < g) ? part[k] : 0u; }
#pragma unroll
        for (int k = RS_GROUP; k < 2 * RS_GROUP; ++k) before += part[k];
        // exclusive scan of the digit totals over the 256 digits (thread = digit)
        uint32_t incl = tot;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
        if (lane == 63) s_scan[wv] = incl;
        __syncthreads();
        uint32_t wbase = 0u;
        for (int w = 0; w < wv; ++w) wbase += s_scan[w];
        before += wbase + incl - tot;
        __syncthreads();
    }
    // ---- per-wave digit counts of this block
    s_wcnt[0][tid] = 0u; s_wcnt[1][tid] = 0u; s_wcnt[2][tid] = 0u; s_wcnt[3][tid] = 0u;
    __syncthreads();
    for (uint32_t b0 = wlo; b0 < whi; b0 += 64u * RB) {
        if (b0 != wlo) load_batch(b0);
#pragma unroll
        for (int rr = 0; rr < RB; ++rr)
            if (b0 + 64u * rr + lane < whi) atomicAdd(&s_wcnt[wv][digit_of(kk[rr], shift, mask)], 1u);
    }
    __syncthreads();
    {   // thread = digit: turn the four wave counts into the four waves' first output slots
        uint32_t run = before;
#pragma unroll
        for (int w = 0; w < 4; ++w) { const uint32_t c = s_wcnt[w][tid]; s_wcnt[w][tid] = run; run += c; }
    }
    __syncthreads();
    // ---- stable ranking + scatter, one 64-element row at a time (rows in order, lanes in order)
    uint32_t* cur = s_wcnt[wv];
    for (uint32_t b0 = wlo; b0 < whi; b0 += 64u * RB) {
        if (!single) load_batch(b0);            // (a single batch is still in registers from the count pass)
        uint32_t dr[RB];
        if (MODE == 2) {        // the final pass's two gathers by depth rank, also batched
#pragma unroll
            for (int rr = 0; rr < RB; ++rr) {
                const bool have = b0 + 64u * rr + lane < whi;
                dr[rr] = have ? depth_rank[(uint32_t)kk[rr]] : 0u;
                vv[rr] = have ? id_rank[(uint32_t)kk[rr]] : 0u;
            }
        }
#pragma unroll
        for (int rr = 0; rr < RB; ++rr) {
            const uint32_t i = b0 + 64u * rr + lane;
            if (b0 + 64u * rr >= whi) break;                      // wave-uniform
            const bool have = i < whi;
            const KEY k = kk[rr];
            const uint32_t d = have ? digit_of(k, shift, mask) : 0xFFFFFFFFu;
            // peers = lanes of this row with my digit (one ballot per digit bit)
            unsigned long long peers = __ballot(have);
            for (int bt = 0; bt < bits; ++bt) {
                const unsigned long long m = __ballot((d >> bt) & 1u);
                peers &= ((d >> bt) & 1u) ? m : ~m;
            }
            uint32_t pos = 0u;
            if (have) {
                const uint32_t below = (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0u));
                pos = cur[d] + below;
            }
            __builtin_amdgcn_wave_barrier();
            if (have && (peers >> lane) >> 1 == 0ull) cur[d] = pos + 1u;       // highest peer advances the cursor past the row
            __builtin_amdgcn_wave_barrier();
            if (have) {
                if (MODE == 0) {
                    keys_out[pos] = k;
                    vals_out[pos] = vv[rr];
                } else if (MODE == 1) {
                    keys_out[pos] = k;
                } else {
                    keys_out[pos] = (KEY)(((uint64_t)k & 0xFFFFFFFF00000000ull) | (uint64_t)dr[rr]);
                    vals_out[pos] = vv[rr];
                }
            }
        }
    }
}

// ---- exclusive scan of tiles_touched in depth-rank order (two launches, 2048 elements per block)
#ifndef SC_PER_DEF
#define SC_PER_DEF 512          // 2048 / 1024 / 512: K2 48.6 / 46.5 / 45.2 us at C2 (same reason as RS_PER_MIN)
#endif
constexpr int SC_PER = SC_PER_DEF;
constexpr int SC_ELEMS = SC_PER / RS_THREADS;      // consecutive elements per thread in k_scan_apply
__global__ void __launch_bounds__(RS_THREADS)
k_scan_sums(int N, const uint32_t* __restrict__ id_rank, const uint32_t* __restrict__ tiles_touched, uint32_t* __restrict__ bsum) {
    __shared__ uint32_t s_w[4];
    const int base = blockIdx.x * SC_PER;
    uint32_t s = 0u;
    {   // both gathers batched: 8 index loads in flight, then 8 value loads
        uint32_t ids[SC_PER / RS_THREADS];
#pragma unroll
        for (int k = 0; k < SC_PER / RS_THREADS; ++k) { const int r = base + (int)threadIdx.x + k * RS_THREADS; ids[k] = (r < N) ? id_rank[r] : 0u; }
#pragma unroll
        for (int k = 0; k < SC_PER / RS_THREADS; ++k) { const int r = base + (int)threadIdx.x + k * RS_THREADS; s += (r < N) ? tiles_touched[ids[k]] : 0u; }
    }
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) bsum[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
__global__ void __launch_bounds__(RS_THREADS)
k_scan_apply(int N, int nblocks, const uint32_t* __restrict__ id_rank, const uint32_t* __restrict__ tiles_touched,
             const uint32_t* __restrict__ bsum, uint32_t* __restrict__ offs_rank) {
    __shared__ uint32_t s_w[4];
    __shared__ uint32_t s_base;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    {   // sum of the block sums before this block
        uint32_t s = 0u;
        for (int k = tid; k < (int)blockIdx.x; k += RS_THREADS) s += bsum[k];
        for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
        if (lane == 0) s_w[wv] = s;
        __syncthreads();
        if (tid == 0) s_base = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        __syncthreads();
    }
    // thread t owns elements [t * SC_ELEMS, (t + 1) * SC_ELEMS) of the block (rank order)
    const int r0 = blockIdx.x * SC_PER + tid * SC_ELEMS;
    uint32_t v[SC_ELEMS];
    uint32_t s = 0u;
#pragma unroll
    for (int k = 0; k < SC_ELEMS; ++k) { const int r = r0 + k; v[k] = (r < N) ? tiles_touched[id_rank[r]] : 0u; s += v[k]; }
    uint32_t incl = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
    __syncthreads();
    if (lane == 63) s_w[wv] = incl;
    __syncthreads();
    uint32_t run = s_base + incl - s;
    for (int w = 0; w < wv; ++w) run += s_w[w];
#pragma unroll
    for (int k = 0; k < SC_ELEMS; ++k) { const int r = r0 + k; if (r < N) offs_rank[r] = run; run += v[k]; }
}

// K3: one thread per depth rank; emits that Gaussian's instances, element = (tile << 32) | rank, at offs_rank[rank]...
// Also zero-fills `ranges` (empty tiles keep (0, 0); k_ranges runs later on the same stream).
__global__ void __launch_bounds__(TG_BLOCK)
k_duplicate(int N, int tiles_x, int T, const uint32_t* __restrict__ id_rank, const uint32_t* __restrict__ offs_rank,
            const uint32_t* __restrict__ tiles_touched, const uint2* __restrict__ rect, uint64_t* __restrict__ elems,
            uint2* __restrict__ ranges, uint32_t* __restrict__ zero_words, int num_zero_words) {
    const int r = blockIdx.x * TG_BLOCK + threadIdx.x;
    for (int k = r; k < T; k += (int)gridDim.x * TG_BLOCK) ranges[k] = make_uint2(0u, 0u);
    for (int k = r; k < num_zero_words; k += (int)gridDim.x * TG_BLOCK) zero_words[k] = 0u;      // group count tables of the tile sort
    if (r >= N) return;
    const uint32_t id = id_rank[r];
    if (tiles_touched[id] == 0u) return;
    uint32_t off = offs_rank[r];
    const uint2 rc = rect[id];
    const uint32_t x0 = rc.x & 0xffffu, y0 = rc.x >> 16, x1 = rc.y & 0xffffu, y1 = rc.y >> 16;
    for (uint32_t y = y0; y < y1; ++y)
        for (uint32_t x = x0; x < x1; ++x)
            elems[off++] = ((uint64_t)(y * (uint32_t)tiles_x + x) << 32) | (uint64_t)(uint32_t)r;
}

// K5: ranges[tile] = [first, last) by boundary detection on the sorted keys' high word (K3 zero-filled `ranges`).
__global__ void __launch_bounds__(TG_BLOCK)
k_ranges(uint32_t D, const uint64_t* __restrict__ keys, uint2* __restrict__ ranges) {
    const uint32_t i = blockIdx.x * TG_BLOCK + threadIdx.x;
    if (i >= D) return;
    const uint32_t cur = (uint32_t)(keys[i] >> 32);
    if (i == 0) ranges[cur].x = 0;
    else {
        const uint32_t prev = (uint32_t)(keys[i - 1] >> 32);
        if (cur != prev) { ranges[prev].y = i; ranges[cur].x = i; }
    }
    if (i == D - 1) ranges[cur].y = D;
}

// Launch order of the blend kernels: longest tile list first (LPT), as a counting sort on a 1024-level log-ish length
// bucket in ONE workgroup (speed only: any order is correct; ties keep no particular order).
__global__ void __launch_bounds__(1024)
k_tile_order(uint32_t T, const uint2* __restrict__ ranges, uint32_t* __restrict__ order, uint32_t* __restrict__ zero_words,
             int num_zero_words) {
    __shared__ uint32_t s_cnt[1024];
    __shared__ uint32_t s_w[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int k = tid; k < num_zero_words; k += 1024) zero_words[k] = 0u;       // K6's per-bin footprint counters (texgs.h tex_bin_count)
    auto bucket = [](uint32_t len) -> uint32_t {            // monotone decreasing in len: 1023 = empty, 0 = longest
        if (len == 0u) return 1023u;
        const uint32_t e = 31u - (uint32_t)__clz((int)len);              // floor(log2 len), 0..31
        const uint32_t frac = (e >= 5u) ? ((len >> (e - 5u)) & 31u) : ((len << (5u - e)) & 31u);
        const uint32_t key = min(e * 32u + frac, 1022u);
        return 1022u - key;
    };
    s_cnt[tid] = 0u;
    __syncthreads();
    for (uint32_t i = tid; i < T; i += 1024u) { const uint2 r = ranges[i]; atomicAdd(&s_cnt[bucket(r.y - r.x)], 1u); }
    __syncthreads();
    const uint32_t c = s_cnt[tid];
    uint32_t incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
    if (lane == 63) s_w[wv] = incl;
    __syncthreads();
    uint32_t base = incl - c;
    for (int w = 0; w < wv; ++w) base += s_w[w];
    __syncthreads();
    s_cnt[tid] = base;
    __syncthreads();
    for (uint32_t i = tid; i < T; i += 1024u) {
        const uint2 r = ranges[i];
        order[atomicAdd(&s_cnt[bucket(r.y - r.x)], 1u)] = i;
    }
}

inline int tile_bits(uint32_t T) {
    int bits = 0;
    while ((1u << bits) < T && bits < 31) ++bits;     // ceil(log2 T)
    return bits == 0 ? 1 : bits;
}
inline size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

// ---- scratch layouts ------------------------------------------------------------------------------------------------
// scan_temp (Gaussian level, sized by scan_temp_bytes(N)):
//   [0]        header zero-filled by K1: 4 x gtable of the depth passes
//   then       4 x table, key_a, key_b, val_a, val_b (u32[N] each), bsum (u32[ceil(N / 2048)]), block_D (u32[ceil(N / 256)][3])
// sort_temp (instance level, sized by sort_temp_bytes(capacity, T)):
//   [0]        header zero-filled by K3: 3 x gtable of the tile passes
//   then       3 x table, elem_tmp (u64[capacity])
struct GaussScratch {
    uint32_t* tables;       // 4 passes
    uint32_t* block_D;      // [ceil(N / 256)][3] K1's per-workgroup sums {tiles_touched, fingerprint lo, hi} (the host adds them up)
    uint32_t *key_a, *key_b, *val_a, *val_b, *bsum;
    size_t header_bytes;
};
inline size_t zero_header_bytes(int passes, size_t extra) { return align256((size_t)passes * RS_ZERO_WORDS * 4 + extra); }
inline GaussScratch gauss_scratch(void* base, int N) {
    GaussScratch g;
    char* p = (char*)base;
    g.tables = (uint32_t*)p;
    g.header_bytes = zero_header_bytes(4, 0);
    p += g.header_bytes + align256(4 * (size_t)RS_MAX_BLOCKS * 256 * 4);
    const size_t nb = align256((size_t)(N > 0 ? N : 1) * 4);
    g.key_a = (uint32_t*)p; p += nb;
    g.key_b = (uint32_t*)p; p += nb;
    g.val_a = (uint32_t*)p; p += nb;
    g.val_b = (uint32_t*)p; p += nb;
    g.bsum = (uint32_t*)p; p += align256(((size_t)(N > 0 ? N : 1) + SC_PER - 1) / SC_PER * 4 + 256);
    g.block_D = (uint32_t*)p;
    return g;
}
// pass `pass` of `passes`: its gtable sits in the zeroed header at `base`, its table after the header
inline RadixTables tables_at(uint32_t* base, int pass, int passes, size_t extra) {
    RadixTables t;
    t.gtable = base + (size_t)pass * RS_ZERO_WORDS;
    t.table = reinterpret_cast<uint32_t*>((char*)base + zero_header_bytes(passes, extra)) + (size_t)pass * RS_MAX_BLOCKS * 256;
    return t;
}
inline void pass_geometry(uint32_t n, uint32_t& blocks, uint32_t& per) {
    per = ((n + RS_MAX_BLOCKS - 1) / RS_MAX_BLOCKS + 63u) & ~63u;       // elements per block, multiple of 64
    if (per < RS_PER_MIN) per = RS_PER_MIN;
    blocks = (n + per - 1) / per;
    if (blocks == 0u) blocks = 1u;
}

}  // namespace

size_t scan_temp_bytes(int N) {
    const size_t n = (size_t)(N > 0 ? N : 1);
    return zero_header_bytes(4, 0) + align256(4 * (size_t)RS_MAX_BLOCKS * 256 * 4) + 4 * align256(n * 4)
         + align256(((n + SC_PER - 1) / SC_PER) * 4 + 256) + align256(((n + TG_BLOCK - 1) / TG_BLOCK) * 12);
}

constexpr int TILE_PASSES_MAX = 3;      // tile ids up to 2^24 in digits of at most 8 bits (the count / scatter kernels index 256-entry LDS tables)
inline size_t sort_tables_bytes() {
    return zero_header_bytes(TILE_PASSES_MAX, 0) + align256(TILE_PASSES_MAX * (size_t)RS_MAX_BLOCKS * 256 * 4);
}
size_t sort_temp_bytes(uint32_t D, uint32_t T) {
    (void)T;
    return sort_tables_bytes() + align256((size_t)(D > 0 ? D : 1) * 8);
}

uint32_t* bin_block_sums_ptr(const TexGSGeom* g, int N) { return gauss_scratch(g->scan_temp, N).block_D; }
uint32_t* bin_header_ptr(const TexGSGeom* g, int N, int* words) {       // group count tables of the depth passes: K1 zero-fills them
    const GaussScratch gs = gauss_scratch(g->scan_temp, N);
    *words = (int)(gs.header_bytes / 4);
    return gs.tables;
}

// Gaussian level: depth sort (4 passes) + exclusive scan of tiles_touched in rank order.  Needs K1's depth keys
// (bits of view z; 0xFFFFFFFF for culled) in g->depth.  Independent of D: runs while the host waits for the D readback.
int launch_depth_sort_scan(const TexGSGeom* g, int N, hipStream_t s) {
    if (N <= 0) return 0;
    const GaussScratch gs = gauss_scratch(g->scan_temp, N);
    uint32_t blocks, per;
    pass_geometry((uint32_t)N, blocks, per);
    const uint32_t* kin = reinterpret_cast<const uint32_t*>(g->depth);
    const uint32_t* vin = nullptr;                        // first pass: value = index
    uint32_t* kout = gs.key_a; uint32_t* vout = gs.val_a;
    for (int pass = 0; pass < 4; ++pass) {
        const RadixTables t = tables_at(gs.tables, pass, 4, 0);
        hipLaunchKernelGGL(k_radix_count<uint32_t>, dim3(blocks), dim3(RS_THREADS), 0, s, kin, (const uint32_t*)nullptr, (uint32_t)N,
                           per, pass * 8, 255u, t);
        hipLaunchKernelGGL((k_radix_scatter<uint32_t, 0>), dim3(blocks), dim3(RS_THREADS), 0, s, kin, vin, kout, vout,
                           (const uint32_t*)nullptr, (uint32_t)N, per, pass * 8, 255u, 8, t, (const uint32_t*)nullptr,
                           (const uint32_t*)nullptr);
        kin = kout; vin = vout;
        kout = (kout == gs.key_a) ? gs.key_b : gs.key_a;
        vout = (vout == gs.val_a) ? gs.val_b : gs.val_a;
    }
    // after 4 passes the sorted pairs are in (key_b, val_b)
    const int nsb = (N + SC_PER - 1) / SC_PER;
    hipLaunchKernelGGL(k_scan_sums, dim3(nsb), dim3(RS_THREADS), 0, s, N, gs.val_b, g->tiles_touched, gs.bsum);
    hipLaunchKernelGGL(k_scan_apply, dim3(nsb), dim3(RS_THREADS), 0, s, N, nsb, gs.val_b, g->tiles_touched, gs.bsum, g->offsets);
    hipError_t e = hipGetLastError();
    return (int)e;
}

void launch_duplicate(const CamConst& c, const TexGSGeom* g, TexGSBinning* b, hipStream_t s) {
    if (c.N <= 0 || b->num_rendered == 0) return;
    const GaussScratch gs = gauss_scratch(g->scan_temp, c.N);
    const int blocks = (c.N + TG_BLOCK - 1) / TG_BLOCK;
    hipLaunchKernelGGL(k_duplicate, dim3(blocks), dim3(TG_BLOCK), 0, s, c.N, c.tiles_x, c.tiles_x * c.tiles_y, gs.val_b, g->offsets,
                       g->tiles_touched, reinterpret_cast<const uint2*>(g->rect), b->keys_unsorted, reinterpret_cast<uint2*>(b->ranges),
                       reinterpret_cast<uint32_t*>(b->sort_temp), (int)(zero_header_bytes(TILE_PASSES_MAX, 0) / 4));
}

// Instance level: stable LSD sort by tile id in 1-3 digits of at most 8 bits (two for up to 65 536 tiles); the last pass
// writes keys_sorted / point_list.  With three passes (more than 65 536 tiles, i.e. images beyond ~16 Mpixel) the middle pass
// uses keys_unsorted as its output buffer: its K3 contents are then gone after the forward.
int launch_sort(const CamConst& c, const TexGSGeom* g, TexGSBinning* b, hipStream_t s) {
    const uint32_t D = b->num_rendered;
    if (D == 0) return 0;
    const GaussScratch gs = gauss_scratch(g->scan_temp, c.N);
    uint32_t* tbl = reinterpret_cast<uint32_t*>(b->sort_temp);
    uint64_t* tmp = reinterpret_cast<uint64_t*>((char*)b->sort_temp + sort_tables_bytes());
    const int tb = tile_bits((uint32_t)(c.tiles_x * c.tiles_y));
    const int npass = (tb + 7) / 8;
    if (npass > TILE_PASSES_MAX) return (int)hipErrorInvalidValue;       // validate_frame rejects such images first
    uint32_t blocks, per;
    pass_geometry(D, blocks, per);
    const uint64_t* src = b->keys_unsorted;
    int shift = 32, left = tb;
    for (int pass = 0; pass < npass; ++pass) {
        const int bits = (left + (npass - pass) - 1) / (npass - pass);      // even split, low digits first
        const RadixTables t = tables_at(tbl, pass, TILE_PASSES_MAX, 0);
        const uint32_t mask = (1u << bits) - 1u;
        hipLaunchKernelGGL(k_radix_count<uint64_t>, dim3(blocks), dim3(RS_THREADS), 0, s, src, (const uint32_t*)nullptr, D, per,
                           shift, mask, t);
        if (pass == npass - 1) {
            hipLaunchKernelGGL((k_radix_scatter<uint64_t, 2>), dim3(blocks), dim3(RS_THREADS), 0, s, src, (const uint32_t*)nullptr,
                               b->keys_sorted, b->point_list, (const uint32_t*)nullptr, D, per, shift, mask, bits, t,
                               (const uint32_t*)gs.key_b, (const uint32_t*)gs.val_b);
        } else {
            uint64_t* dst = (src == tmp) ? b->keys_unsorted : tmp;
            hipLaunchKernelGGL((k_radix_scatter<uint64_t, 1>), dim3(blocks), dim3(RS_THREADS), 0, s, src, (const uint32_t*)nullptr, dst,
                               (uint32_t*)nullptr, (const uint32_t*)nullptr, D, per, shift, mask, bits, t, (const uint32_t*)nullptr,
                               (const uint32_t*)nullptr);
            src = dst;
        }
        shift += bits; left -= bits;
    }
    return (int)hipGetLastError();
}

void launch_ranges(const CamConst& c, TexGSBinning* b, uint32_t* zero_words, int num_zero_words, hipStream_t s) {
    const uint32_t T = (uint32_t)(c.tiles_x * c.tiles_y);
    if (b->num_rendered == 0) (void)hipMemsetAsync(b->ranges, 0, sizeof(uint32_t) * 2 * T, s);     // no K3 ran: every tile is empty
    if (b->num_rendered > 0) {
        const int blocks = (int)((b->num_rendered + TG_BLOCK - 1) / TG_BLOCK);
        hipLaunchKernelGGL(k_ranges, dim3(blocks), dim3(TG_BLOCK), 0, s, b->num_rendered, b->keys_sorted,
                           reinterpret_cast<uint2*>(b->ranges));
    }
    hipLaunchKernelGGL(k_tile_order, dim3(1), dim3(1024), 0, s, T, reinterpret_cast<const uint2*>(b->ranges), b->tile_order,
                       zero_words, num_zero_words);
}
