// K2-K5: tile binning -- hand-written radix sorts, scan, duplicate-with-keys, tile ranges, tile launch order.  gfx950.
//
// Bit-exact contract: the tile lists are the (tile, depth-bits, Gaussian-index)-lexicographic
// order of all (Gaussian, touched tile) instances; keys_sorted[k] = (tile << 32) | bits(depth), point_list[k] = index.
// The lineage gets there with ONE stable LSD sort of D 64-bit keys over 32 + log2(T) bits (6 onesweep passes with
// decoupled look-back; rocPRIM's took 210 us at C3 -- every look-back hop is a ~1 us cross-XCD round trip here).
// This file sorts in two levels instead:
//   1. the N GAUSSIANS by (depth bits, index) (culled ones carry 0xFFFFFFFF): a partition into depth bins, cut into balanced
//      groups of consecutive bins, and a bitonic sort of every group (round 4; rounds 2-3: four stable 8-bit passes),
//   2. an exclusive scan of tiles_touched in that depth-rank order (inside the group sort + one add of the groups in front),
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

// ---- K2, round 4: depth sort of the N Gaussians + exclusive scan of tiles_touched in rank order in THREE launches -------------
// (round 3: four stable 8-bit radix passes = 8 launches, + 2 for the scan; every launch is one or two dependent round trips over
// 2.4 MB on a chip it cannot fill: 64 us at C3, 46 us at C2.)  Now:
//   k_depth_count    partition, step 1: K1 left (min, max) of the valid depth keys per workgroup; bin = (key - min) * NB / (max -
//                    min + 1) (NB = a power of two, 128 - 256 Gaussians per bin on average; culled Gaussians, key
//                    0xFFFFFFFF: bin NB); per-block LDS histogram -> global bin totals (atomics);
//   k_depth_scatter  step 2: exclusive scan of the totals (every block, redundantly), one returning atomic per (block, non-empty
//                    bin) reserves the block's slots, elements go to their bin's region in ARBITRARY order.  Block 0 also cuts
//                    the bins into GROUPS of consecutive bins: a new group starts where the running total crosses a multiple of
//                    DS_GROUP -- the sort units are balanced whatever the depth distribution (a far outlier stretches the key
//                    range and leaves most bins empty; equal-width sort units would then hold thousands);
//   k_depth_group_sort  one workgroup per group: its (key, index) pairs are sorted by (key, index) -- a total order, so the
//                    arbitrary order of step 2 does not matter and the result is exactly the stable order by (depth bits, index)
//                    of the lineage -- in registers (<= 1024 pairs: bitonic network over lane shuffles, LDS only for the two
//                    cross-wave distances), in LDS (<= 2048), or by an LSD radix in global memory (a single bin holds > ~1900
//                    Gaussians, i.e. 1/NB of the depth range does; one wave, slow, correct); tiles_touched is gathered in that
//                    order and scanned inside the group;
//                    K3 scans the groups' tile sums (every workgroup, redundantly) and adds a rank's group prefix when it reads
//                    its group-local offset from scratch (and writes the sum to geom->offsets, an output of the contract).
// (Tried and dropped, profiles/r04_ablation.md: equal-width sort units; one wave per unit; the scans done by "the last block to
//  finish" -- a ticket word takes 13 ns per workgroup, same address.)
constexpr int DS_NB_MAX = 8192;        // depth bins (+ 1 for culled Gaussians)
constexpr int DS_THREADS = 256;
constexpr int DS_PER = 2048;           // elements per block of the count / scatter kernels
constexpr int DS_CAP = 2048;           // pairs a group may hold to be sorted in LDS
constexpr uint32_t DS_GROUP = 160u;    // a group closes at the first bin boundary past a multiple of this many Gaussians
constexpr int DS_FUSED_GROUPS = 2048;  // up to this many groups K3 scans the group sums itself; beyond: k_depth_prefix
constexpr uint32_t DS_COPY = 4096u;    // culled pairs one workgroup of the group-sort kernel copies
constexpr int DS_BLK_WORDS = 5;        // K1's per-workgroup words: tiles_touched, fingerprint lo / hi, min / max valid depth key
// drange words (written by block 0 of k_depth_scatter)
enum { DR_LO = 0, DR_SCALE = 1, DR_NB = 2, DR_GROUPS = 3, DR_BITS = 4 };

struct DepthRange { uint32_t lo, scale, nb, bits; };

inline int depth_log2_bins(int N) {                 // NB = the power of two in [256, DS_NB_MAX] with 128 < N / NB <= 256.  Measured at
    int lb = 8;                                     // C3 (N = 300 k), count + scatter + sort + K3, us: NB 512: 76, 1024: 54.5, 2048:
    while ((N >> lb) > 256 && (1 << lb) < DS_NB_MAX) ++lb;      // 54.6, 4096: 58, 8192: 65 -- finer bins make the sort cheaper and the
    return lb;                                      // partition dearer (one returning atomic per (block, bin) it touches)
}
inline int depth_max_groups(int N, int nb) { const int g = N / (int)DS_GROUP + 2; return g < nb ? g : nb; }

// every block derives the same (lo, scale) from K1's per-workgroup (min, max)
__device__ __forceinline__ DepthRange depth_range(const uint32_t* __restrict__ block_D, int nblk, int lb, uint32_t* s_red) {
    uint32_t mn = 0xFFFFFFFFu, mx = 0u;
    const uint32_t *bmin = block_D + 3 * (size_t)nblk, *bmax = block_D + 4 * (size_t)nblk;
#pragma unroll 4
    for (int k = threadIdx.x; k < nblk; k += (int)blockDim.x) { mn = min(mn, bmin[k]); mx = max(mx, bmax[k]); }
    for (int d = 32; d >= 1; d >>= 1) { mn = min(mn, (uint32_t)__shfl_xor((int)mn, d, 64)); mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, 64)); }
    const int nw = (int)blockDim.x >> 6;
    if (nw > 1) {
        if ((threadIdx.x & 63) == 0) { s_red[threadIdx.x >> 6] = mn; s_red[4 + (threadIdx.x >> 6)] = mx; }
        __syncthreads();
        mn = s_red[0]; mx = s_red[4];
        for (int w = 1; w < nw; ++w) { mn = min(mn, s_red[w]); mx = max(mx, s_red[4 + w]); }
        __syncthreads();
    }
    const uint32_t range = (mx >= mn) ? mx - mn : 0u;
    DepthRange r; r.lo = mn; r.nb = 1u << lb;
    r.bits = range ? (uint32_t)(32 - __clz((int)range)) : 0u;
    // bin = floor((key - lo) * scale / 2^32), scale = floor(2^32 * nb / (range + 1)) -> bin <= range * nb / (range + 1) < nb;
    // a range narrower than nb keys: bin = key - lo (scale 0)
    r.scale = (range < r.nb) ? 0u : (uint32_t)(((unsigned long long)r.nb << 32) / ((unsigned long long)range + 1ull));
    return r;
}
__device__ __forceinline__ uint32_t depth_bin(uint32_t key, uint32_t lo, uint32_t scale, uint32_t nb) {
    return key == 0xFFFFFFFFu ? nb : (scale ? __umulhi(key - lo, scale) : key - lo);
}
__device__ __forceinline__ uint32_t depth_bin(uint32_t key, const DepthRange& r) { return depth_bin(key, r.lo, r.scale, r.nb); }

__global__ void __launch_bounds__(DS_THREADS)
k_depth_count(const uint32_t* __restrict__ keys, int N, const uint32_t* __restrict__ block_D, int nblk, int lb, uint32_t* __restrict__ btot) {
    __shared__ uint32_t s_hist[DS_NB_MAX + 1];
    __shared__ uint32_t s_red[8];
    const int tid = threadIdx.x;
    const int base = blockIdx.x * DS_PER;
    uint32_t kk[DS_PER / DS_THREADS];                   // issued before the (min, max) reduction: one round trip for both
#pragma unroll
    for (int u = 0; u < DS_PER / DS_THREADS; ++u) { const int i = base + tid + u * DS_THREADS; kk[u] = (i < N) ? keys[i] : 0u; }
    const int nb1 = (1 << lb) + 1;
    for (int k = tid; k < nb1; k += DS_THREADS) s_hist[k] = 0u;
    const DepthRange r = depth_range(block_D, nblk, lb, s_red);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < DS_PER / DS_THREADS; ++u) if (base + tid + u * DS_THREADS < N) atomicAdd(&s_hist[depth_bin(kk[u], r)], 1u);
    __syncthreads();
    // consecutive lanes -> consecutive words: atomics on runs of addresses go through at 13x the rate of scattered ones (scripts/ubench)
    for (int k = tid; k < nb1; k += DS_THREADS) { const uint32_t c = s_hist[k]; if (c) atomicAdd(&btot[k], c); }
}

__global__ void __launch_bounds__(DS_THREADS)
k_depth_scatter(const uint32_t* __restrict__ keys, int N, const uint32_t* __restrict__ block_D, int nblk, int lb,
                const uint32_t* __restrict__ btot, uint32_t* __restrict__ bcur, uint32_t* __restrict__ gpos, uint32_t* __restrict__ gmap,
                uint32_t* __restrict__ drange, unsigned long long* __restrict__ pair_out) {
    __shared__ uint32_t s_hist[DS_NB_MAX + 1];          // this block's counts, then its next free slot per bin
    __shared__ uint32_t s_base[DS_NB_MAX + 2];
    __shared__ uint32_t s_red[8];
    __shared__ uint32_t s_w[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int base = blockIdx.x * DS_PER;
    const int nb = 1 << lb, nb1 = nb + 1;
    uint32_t kk[DS_PER / DS_THREADS];                   // keys, bin totals and K1's (min, max): all loads in flight together
#pragma unroll
    for (int u = 0; u < DS_PER / DS_THREADS; ++u) { const int i = base + tid + u * DS_THREADS; kk[u] = (i < N) ? keys[i] : 0u; }
    for (int k = tid; k < nb1; k += DS_THREADS) { s_base[k] = btot[k]; s_hist[k] = 0u; }
    const DepthRange r = depth_range(block_D, nblk, lb, s_red);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < DS_PER / DS_THREADS; ++u) if (base + tid + u * DS_THREADS < N) atomicAdd(&s_hist[depth_bin(kk[u], r)], 1u);
    const int per = (nb1 + DS_THREADS - 1) / DS_THREADS;
    {   // exclusive scan of the nb + 1 bin totals in LDS (every block, redundantly): thread t owns `per` consecutive bins
        uint32_t sum = 0u;
        for (int k = 0; k < per; ++k) { const int q = tid * per + k; sum += (q < nb1) ? s_base[q] : 0u; }
        uint32_t incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64); if (lane >= d) incl += o; }
        if (lane == 63) s_w[wv] = incl;
        __syncthreads();
        uint32_t run = incl - sum;
        for (int w = 0; w < wv; ++w) run += s_w[w];
        for (int k = 0; k < per; ++k) { const int q = tid * per + k; if (q < nb1) { const uint32_t c = s_base[q]; s_base[q] = run; run += c; } }
        if (tid == DS_THREADS - 1) s_base[nb1] = run;
    }
    __syncthreads();
    // this block's slots in every bin it has elements for: one returning atomic per (block, bin), consecutive lanes on consecutive
    // words, eight in flight per thread (the LDS store needs the atomic's result: one at a time is one round trip each)
    for (int k0 = 0; k0 < nb1; k0 += 8 * DS_THREADS) {
        uint32_t c[8], o[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int k = k0 + u * DS_THREADS + tid; c[u] = (k < nb1) ? s_hist[k] : 0u; }
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int k = k0 + u * DS_THREADS + tid; o[u] = c[u] ? atomicAdd(&bcur[k], c[u]) : 0u; }
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int k = k0 + u * DS_THREADS + tid; if (k < nb1) s_hist[k] = s_base[k] + o[u]; }
    }
    if (blockIdx.x == 0) {
        // groups of consecutive bins [0, nb): bin q opens a group when its first slot lies in another DS_GROUP-block of slots than
        // the previous bin's.  gpos[g] = first slot of group g (gpos[G] = number of valid Gaussians), gmap[q] = group of bin q;
        // the culled bin nb is "group" G with tile sum 0.
        unsigned long long flags = 0ull;                // thread t owns bins [t * per, t * per + per), per <= 33
        for (int k = 0; k < per; ++k) {
            const int q = tid * per + k;
            if (q < nb && (q == 0 || s_base[q] / DS_GROUP != s_base[q - 1] / DS_GROUP)) flags |= 1ull << k;
        }
        const uint32_t mine = (uint32_t)__popcll(flags);
        uint32_t incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64); if (lane >= d) incl += o; }
        __syncthreads();                                // s_w is reused
        if (lane == 63) s_w[wv] = incl;
        __syncthreads();
        uint32_t g = incl - mine;                       // groups opened before this thread's bins
        for (int w = 0; w < wv; ++w) g += s_w[w];
        const uint32_t G = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        for (int k = 0; k < per; ++k) {
            const int q = tid * per + k;
            if (q < nb) {
                if ((flags >> k) & 1ull) { gpos[g] = s_base[q]; ++g; }
                gmap[q] = g - 1u;                       // bin 0 always opens group 0
            }
        }
        if (tid == 0) {
            gpos[G] = s_base[nb]; gpos[G + 1] = s_base[nb1];        // culled "group": [valid, N)
            gmap[nb] = G;
            drange[DR_LO] = r.lo; drange[DR_SCALE] = r.scale; drange[DR_NB] = r.nb; drange[DR_GROUPS] = G; drange[DR_BITS] = r.bits;
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < DS_PER / DS_THREADS; ++u) {
        const int i = base + tid + u * DS_THREADS;
        if (i < N) { const uint32_t p = atomicAdd(&s_hist[depth_bin(kk[u], r)], 1u); pair_out[p] = ((unsigned long long)kk[u] << 32) | (unsigned long long)(uint32_t)i; }
    }
}

// loads that must see what this wave (or another) wrote to global memory earlier (the fallback's ping-pong passes, the last wave's scan)
__device__ __forceinline__ uint32_t ld_coherent(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// One stable 8-bit LSD radix pass over n (key, val) pairs in global memory by ONE wave (rows of 64 ranked with ballot matches exactly
// as k_radix_scatter does).  digit = ((by_val ? val : key - lo) >> shift) & 255.
__device__ void wave_radix_pass(const uint32_t* kin, const uint32_t* vin, uint32_t* kout, uint32_t* vout, uint32_t n, uint32_t lo,
                                bool by_val, int shift, uint32_t* s_cnt /*[256]*/) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 4; ++k) s_cnt[lane * 4 + k] = 0u;
    __builtin_amdgcn_wave_barrier();
    for (uint32_t i = lane; i < n; i += 64u) {
        const uint32_t x = by_val ? ld_coherent(vin + i) : ld_coherent(kin + i) - lo;
        atomicAdd(&s_cnt[(x >> shift) & 255u], 1u);
    }
    __builtin_amdgcn_wave_barrier();
    {   // exclusive scan over the 256 digits: lane owns 4 consecutive digits
        uint32_t c[4], sum = 0u;
#pragma unroll
        for (int k = 0; k < 4; ++k) { c[k] = s_cnt[lane * 4 + k]; sum += c[k]; }
        uint32_t incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64); if (lane >= d) incl += o; }
        uint32_t run = incl - sum;
#pragma unroll
        for (int k = 0; k < 4; ++k) { s_cnt[lane * 4 + k] = run; run += c[k]; }
    }
    __builtin_amdgcn_wave_barrier();
    for (uint32_t b0 = 0u; b0 < n; b0 += 64u) {
        const uint32_t i = b0 + lane;
        const bool have = i < n;
        const uint32_t k = have ? ld_coherent(kin + i) : 0u, v = have ? ld_coherent(vin + i) : 0u;
        const uint32_t d = have ? (((by_val ? v : k - lo) >> shift) & 255u) : 0xFFFFFFFFu;
        unsigned long long peers = __ballot(have);
        for (int bt = 0; bt < 8; ++bt) { const unsigned long long m = __ballot((d >> bt) & 1u); peers &= ((d >> bt) & 1u) ? m : ~m; }
        uint32_t pos = 0u;
        if (have) pos = s_cnt[d] + (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0u));
        __builtin_amdgcn_wave_barrier();
        if (have && (peers >> lane) >> 1 == 0ull) s_cnt[d] = pos + 1u;
        __builtin_amdgcn_wave_barrier();
        if (have) { kout[pos] = k; vout[pos] = v; }
    }
    __threadfence();
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long x, int m) {
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)x, m, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)(x >> 32), m, 64);
    return ((unsigned long long)hi << 32) | lo;
}

// Bitonic sort of P = 256 * E pairs held E per thread (element i = e * 256 + tid) by one workgroup of 256: compare-exchange
// distances j < 64 go over lane shuffles, j = 64 / 128 through LDS (the partner sits in another wave), j >= 256 stay inside the
// thread's registers.  P = 256: 33 of the 36 steps never touch LDS or a barrier.
template <int E>
__device__ __forceinline__ void bitonic_regs(unsigned long long (&x)[E], unsigned long long* s_pair, int tid) {
    constexpr uint32_t P = 256u * E;
#pragma unroll
    for (uint32_t k = 2u; k <= P; k <<= 1) {
#pragma unroll
        for (uint32_t j = k >> 1; j > 0u; j >>= 1) {
            if (j >= 256u) {
                constexpr int dummy = 0; (void)dummy;
                const uint32_t je = j >> 8;
#pragma unroll
                for (uint32_t e = 0; e < (uint32_t)E; ++e) {
                    if ((e & je) == 0u) {
                        const uint32_t i = e * 256u + (uint32_t)tid;
                        const bool up = (i & k) == 0u;
                        const unsigned long long a = x[e], c = x[e | je];
                        const bool sw = (a > c) == up;
                        x[e] = sw ? c : a; x[e | je] = sw ? a : c;
                    }
                }
            } else if (j >= 64u) {
                __syncthreads();
#pragma unroll
                for (uint32_t e = 0; e < (uint32_t)E; ++e) s_pair[e * 256u + (uint32_t)tid] = x[e];
                __syncthreads();
#pragma unroll
                for (uint32_t e = 0; e < (uint32_t)E; ++e) {
                    const uint32_t i = e * 256u + (uint32_t)tid;
                    const unsigned long long y = s_pair[i ^ j];
                    const bool take_min = ((i & k) == 0u) == ((i & j) == 0u);
                    x[e] = take_min ? (x[e] < y ? x[e] : y) : (x[e] > y ? x[e] : y);
                }
            } else {
#pragma unroll
                for (uint32_t e = 0; e < (uint32_t)E; ++e) {
                    const uint32_t i = e * 256u + (uint32_t)tid;
                    const unsigned long long y = shfl_xor_u64(x[e], (int)j);
                    const bool take_min = ((i & k) == 0u) == ((i & j) == 0u);
                    x[e] = take_min ? (x[e] < y ? x[e] : y) : (x[e] > y ? x[e] : y);
                }
            }
        }
    }
}

// sorts the group [start, start + n), n <= 256 * E, writes (key, index) and the in-group exclusive scan of tiles_touched in
// rank order; returns the group's tile sum (in every thread)
template <int E>
__device__ __forceinline__ uint32_t group_sort_regs(const unsigned long long* __restrict__ pair_a, uint32_t start, uint32_t n,
                                                     uint32_t* __restrict__ key_b, uint32_t* __restrict__ val_b,
                                                     const uint32_t* __restrict__ tiles_touched, uint32_t* __restrict__ offs_rank,
                                                     unsigned long long* s_pair, uint32_t* s_w, int tid) {
    const int lane = tid & 63, wv = tid >> 6;
    unsigned long long x[E];
#pragma unroll
    for (int e = 0; e < E; ++e) { const uint32_t i = (uint32_t)e * 256u + (uint32_t)tid; x[e] = (i < n) ? pair_a[start + i] : ~0ull; }
    bitonic_regs<E>(x, s_pair, tid);
    uint32_t tt[E], incl[E];
#pragma unroll
    for (int e = 0; e < E; ++e) { const uint32_t i = (uint32_t)e * 256u + (uint32_t)tid; tt[e] = (i < n) ? tiles_touched[(uint32_t)x[e]] : 0u; }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        uint32_t v = tt[e];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)v, d, 64); if (lane >= d) v += o; }
        incl[e] = v;
        if (lane == 63) s_w[e * 4 + wv] = v;
    }
    __syncthreads();
    uint32_t run = 0u, total = 0u;
#pragma unroll
    for (int q = 0; q < 4 * E; ++q) { const uint32_t w = s_w[q]; total += w; }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const uint32_t i = (uint32_t)e * 256u + (uint32_t)tid;
        uint32_t base = run;
#pragma unroll
        for (int w = 0; w < 4; ++w) { const uint32_t sw = s_w[e * 4 + w]; if (w < wv) base += sw; run += sw; }
        if (i < n) { key_b[start + i] = (uint32_t)(x[e] >> 32); val_b[start + i] = (uint32_t)x[e]; offs_rank[start + i] = base + incl[e] - tt[e]; }
    }
    return total;
}

__global__ void __launch_bounds__(DS_THREADS)
k_depth_group_sort(int N, const uint32_t* __restrict__ gpos, const uint32_t* __restrict__ drange,
                   unsigned long long* pair_a, uint32_t* key_b, uint32_t* val_b,
                   const uint32_t* __restrict__ tiles_touched, uint32_t* __restrict__ offs_rank, uint32_t* __restrict__ bsum, int gmax) {
    __shared__ unsigned long long s_pair[DS_CAP];          // 16 KB: the cross-wave exchanges (the fallback's digit table aliases it)
    __shared__ uint32_t s_w[4 * (DS_CAP / DS_THREADS)];
    const int b = (int)blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t G = drange[DR_GROUPS];                  // <= gmax (depth_max_groups)
    if (b >= gmax) {            // culled Gaussians: they emit nothing (tiles_touched = 0), their mutual order is irrelevant; the copy is
        const uint32_t start = gpos[G], n = gpos[G + 1] - start;        // spread over N / DS_COPY workgroups (one would be the kernel's tail)
        const uint32_t lo = (uint32_t)(b - gmax) * DS_COPY, hi = min(n, lo + DS_COPY);
        for (uint32_t i = lo + tid; i < hi; i += DS_THREADS) {
            const unsigned long long x = pair_a[start + i];
            key_b[start + i] = (uint32_t)(x >> 32); val_b[start + i] = (uint32_t)x; offs_rank[start + i] = 0u;
        }
        if (b == gmax && tid == 0) bsum[G] = 0u;
        return;
    }
    const uint32_t start = gpos[b], n = gpos[b + 1] - start;     // (loaded before G is known: b + 1 <= gmax lies inside gpos)
    if ((uint32_t)b >= G) return;
    // n is workgroup-uniform: every branch below is taken by all 256 threads or none
    uint32_t total = 0u;                                     // this group's sum of tiles_touched (valid in thread 0 at the end)
    if (n == 0u) {
        // nothing
    } else if (n <= 64u) {
        if (wv != 0) return;
        // ---- one wave: bitonic network over the 64 lanes (21 compare-exchange steps, two shuffles each)
        unsigned long long x = (lane < (int)n) ? pair_a[start + lane] : ~0ull;
#pragma unroll
        for (int k = 2; k <= 64; k <<= 1)
#pragma unroll
            for (int j = k >> 1; j > 0; j >>= 1) {
                const unsigned long long y = shfl_xor_u64(x, j);
                const bool take_min = ((lane & k) == 0) == ((lane & j) == 0);
                x = take_min ? (x < y ? x : y) : (x > y ? x : y);
            }
        const uint32_t tt = (lane < (int)n) ? tiles_touched[(uint32_t)x] : 0u;
        uint32_t incl = tt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64); if (lane >= d) incl += o; }
        if (lane < (int)n) { key_b[start + lane] = (uint32_t)(x >> 32); val_b[start + lane] = (uint32_t)x; offs_rank[start + lane] = incl - tt; }
        total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    } else if (n <= 256u) {
        total = group_sort_regs<1>(pair_a, start, n, key_b, val_b, tiles_touched, offs_rank, s_pair, s_w, tid);
    } else if (n <= 512u) {
        total = group_sort_regs<2>(pair_a, start, n, key_b, val_b, tiles_touched, offs_rank, s_pair, s_w, tid);
    } else if (n <= 1024u) {
        total = group_sort_regs<4>(pair_a, start, n, key_b, val_b, tiles_touched, offs_rank, s_pair, s_w, tid);
    } else if (n <= (uint32_t)DS_CAP) {
        // ---- 1025 .. 2048 pairs (one bin holds most of them): bitonic sort in LDS, every step behind a barrier; 8 pairs
        // per thread in registers would cost every group of the launch its occupancy
        for (uint32_t i = tid; i < (uint32_t)DS_CAP; i += DS_THREADS) s_pair[i] = (i < n) ? pair_a[start + i] : ~0ull;
        for (uint32_t k = 2u; k <= (uint32_t)DS_CAP; k <<= 1)
            for (uint32_t j = k >> 1; j > 0u; j >>= 1) {
                __syncthreads();
                for (uint32_t t = tid; t < (uint32_t)DS_CAP / 2u; t += DS_THREADS) {
                    const uint32_t i = ((t & ~(j - 1u)) << 1) | (t & (j - 1u)), l = i | j;
                    const unsigned long long x = s_pair[i], y = s_pair[l];
                    if ((x > y) == ((i & k) == 0u)) { s_pair[i] = y; s_pair[l] = x; }
                }
            }
        __syncthreads();
        constexpr uint32_t E = DS_CAP / DS_THREADS;          // consecutive ranks per thread
        uint32_t sum = 0u;
        for (uint32_t u = 0; u < E; ++u) {
            const uint32_t i = tid * E + u;
            if (i < n) {
                const unsigned long long x = s_pair[i];
                key_b[start + i] = (uint32_t)(x >> 32); val_b[start + i] = (uint32_t)x;
                sum += tiles_touched[(uint32_t)x];
            }
        }
        uint32_t incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64); if (lane >= d) incl += o; }
        if (lane == 63) s_w[wv] = incl;
        __syncthreads();
        uint32_t run = incl - sum;
        for (int w = 0; w < wv; ++w) run += s_w[w];
        for (uint32_t u = 0; u < E; ++u) {
            const uint32_t i = tid * E + u;
            if (i < n) { offs_rank[start + i] = run; run += tiles_touched[(uint32_t)s_pair[i]]; }
        }
        total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    } else {
        if (wv != 0) return;
        // ---- fallback (one wave): LSD radix by (key, index) in global memory.  The pairs are unpacked into the group's region of
        // (b); the passes ping-pong between it and the group's own region of pair_a seen as two u32 arrays of n
        uint32_t* s_cnt = reinterpret_cast<uint32_t*>(s_pair);
        const uint32_t lo = drange[DR_LO];
        int nbv = 0; while ((1u << nbv) < (uint32_t)N && nbv < 32) ++nbv;          // index bits
        const int pv = (nbv + 7) / 8, pk = ((int)drange[DR_BITS] + 7) / 8;          // key bits that vary at all
        uint32_t *ki = key_b + start, *vi = val_b + start, *ko = reinterpret_cast<uint32_t*>(pair_a + start), *vo = ko + n;
        for (uint32_t c0 = 0u; c0 < n; c0 += 64u) {         // unpack 64 at a time (the scratch half is the pairs' own storage: all 64
            const uint32_t i = c0 + lane;                   // loads of a row are done before its stores, rows ahead are untouched)
            if (i < n) { const unsigned long long x = pair_a[start + i]; ki[i] = (uint32_t)(x >> 32); vi[i] = (uint32_t)x; }
        }
        __threadfence();
        __builtin_amdgcn_wave_barrier();
        for (int p = 0; p < pv + pk; ++p) {
            wave_radix_pass(ki, vi, ko, vo, n, lo, p < pv, (p < pv ? p : p - pv) * 8, s_cnt);
            uint32_t* t = ki; ki = ko; ko = t; t = vi; vi = vo; vo = t;
        }
        if (ki != key_b + start) {          // an odd number of passes left the result in the scratch half
            for (uint32_t i = lane; i < n; i += 64u) { key_b[start + i] = ld_coherent(ki + i); val_b[start + i] = ld_coherent(vi + i); }
            __threadfence();
            __builtin_amdgcn_wave_barrier();
        }
        uint32_t carry = 0u;                // tiles_touched in rank order, scanned 64 at a time with a running carry
        for (uint32_t c0 = 0u; c0 < n; c0 += 64u) {
            const uint32_t i = c0 + lane;
            const uint32_t tt = (i < n) ? tiles_touched[ld_coherent(val_b + start + i)] : 0u;
            uint32_t incl = tt;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64); if (lane >= d) incl += o; }
            if (i < n) offs_rank[start + i] = carry + incl - tt;
            carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
        total = carry;
    }
    if (tid == 0) bsum[b] = total;          // the group's tile sum; K3 scans them
}

// exclusive scan, in place, of the groups' tile sums (one workgroup; <= DS_NB_MAX + 1 values): only for N beyond DS_FUSED_GROUPS
// groups, see k_duplicate
__global__ void __launch_bounds__(1024)
k_depth_prefix(const uint32_t* __restrict__ drange, uint32_t* __restrict__ bsum) {
    __shared__ uint32_t s_w[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nb1 = (int)drange[DR_GROUPS] + 1;
    const int per = (nb1 + 1023) / 1024;            // <= 9
    uint32_t v[9], sum = 0u;
#pragma unroll
    for (int k = 0; k < 9; ++k) { const int q = tid * per + k; v[k] = (k < per && q < nb1) ? bsum[q] : 0u; sum += v[k]; }
    uint32_t incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64); if (lane >= d) incl += o; }
    if (lane == 63) s_w[wv] = incl;
    __syncthreads();
    uint32_t run = incl - sum;
    for (int w = 0; w < wv; ++w) run += s_w[w];
#pragma unroll
    for (int k = 0; k < 9; ++k) { const int q = tid * per + k; if (k < per && q < nb1) bsum[q] = run; run += v[k]; }
}

// K3: one thread per depth rank; emits that Gaussian's instances, element = (tile << 32) | rank, at offs_rank[rank]...
// loc_rank[rank] is the prefix INSIDE the rank's sort group (k_depth_group_sort, scratch); the group's own prefix (scan of bsum) is
// added here and the sum written to offs_rank (geom->offsets = exclusive scan of tiles_touched in rank order is an output of the
// contract).  Reads only what K2 left and writes only outputs: running K3 again on the same K2 result gives the same result.
// Also zero-fills `ranges` (empty tiles keep (0, 0); k_ranges runs later on the same stream).
// PRE: `bsum` already holds the exclusive scan of the groups' tile sums (k_depth_prefix: large N, where every workgroup redoing
// the scan is N^2 / 40 960 loads) -- otherwise the workgroup scans the <= 2 049 sums itself (cheaper than a launch).
template <bool PRE>
__global__ void __launch_bounds__(TG_BLOCK)
k_duplicate(int N, int tiles_x, int T, const uint32_t* __restrict__ id_rank, const uint32_t* __restrict__ key_rank,
            const uint32_t* __restrict__ loc_rank, uint32_t* __restrict__ offs_rank, const uint32_t* __restrict__ drange, const uint32_t* __restrict__ gmap, const uint32_t* __restrict__ bsum,
            const uint32_t* __restrict__ tiles_touched, const uint2* __restrict__ rect, uint64_t* __restrict__ elems,
            uint2* __restrict__ ranges, uint32_t* __restrict__ zero_words, int num_zero_words) {
    __shared__ uint32_t s_pre[PRE ? 1 : DS_FUSED_GROUPS + 1];
    __shared__ uint32_t s_w[TG_BLOCK / 64];
    const int r = blockIdx.x * TG_BLOCK + threadIdx.x;
    for (int k = r; k < T; k += (int)gridDim.x * TG_BLOCK) ranges[k] = make_uint2(0u, 0u);
    for (int k = r; k < num_zero_words; k += (int)gridDim.x * TG_BLOCK) zero_words[k] = 0u;      // group count tables of the tile sort
    const uint32_t d_lo = drange[DR_LO], d_scale = drange[DR_SCALE], d_nb = drange[DR_NB];
    uint32_t my_group = 0u, my_off = 0u;                // issued ahead of the scan below: key -> bin -> group is two dependent loads
    if (r < N) { my_group = gmap[depth_bin(key_rank[r], d_lo, d_scale, d_nb)]; my_off = loc_rank[r]; }
    uint32_t my_pre = 0u;
    if constexpr (PRE) {
        if (r < N) my_pre = bsum[my_group];
    } else {   // exclusive scan of the groups' tile sums (every block, redundantly: <= 8 KB of L2 reads; a kernel of its own costs more)
        const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nb1 = (int)drange[DR_GROUPS] + 1;
        for (int q = tid; q < nb1; q += TG_BLOCK) s_pre[q] = bsum[q];
        __syncthreads();
        const int per = (nb1 + TG_BLOCK - 1) / TG_BLOCK;
        uint32_t sum = 0u;
        for (int k = 0; k < per; ++k) { const int q = tid * per + k; sum += (q < nb1) ? s_pre[q] : 0u; }
        uint32_t incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl_up((int)incl, d, 64); if (lane >= d) incl += o; }
        if (lane == 63) s_w[wv] = incl;
        __syncthreads();
        uint32_t run = incl - sum;
        for (int w = 0; w < wv; ++w) run += s_w[w];
        for (int k = 0; k < per; ++k) { const int q = tid * per + k; if (q < nb1) { const uint32_t c = s_pre[q]; s_pre[q] = run; run += c; } }
        __syncthreads();
        my_pre = s_pre[my_group];
    }
    if (r >= N) return;
    uint32_t off = my_off + my_pre;
    offs_rank[r] = off;
    const uint32_t id = id_rank[r];
    if (tiles_touched[id] == 0u) return;
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
             int num_zero_words, uint32_t* __restrict__ zero_words2, int num_zero_words2) {
    __shared__ uint32_t s_cnt[1024];
    __shared__ uint32_t s_w[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int k = tid; k < num_zero_words; k += 1024) zero_words[k] = 0u;       // K6's per-bin footprint counters (texgs.h tex_bin_count)
    for (int k = tid; k < num_zero_words2; k += 1024) zero_words2[k] = 0u;     // K6's item-stream page cursors + flag (texgs.h item_ctl)
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
//   [0]        header zero-filled by K1: btot[DS_NB_MAX + 1] bin totals, bcur[DS_NB_MAX + 1] bin fill counters
//   then       gpos[DS_NB_MAX + 2], gmap / bsum[DS_NB_MAX + 1], drange[8], pair_a (u64[N]), key_b, val_b (u32[N] each),
//              block_D (u32[5][ceil(N / 256)])
// sort_temp (instance level, sized by sort_temp_bytes(capacity, T)):
//   [0]        header zero-filled by K3: 3 x gtable of the tile passes
//   then       3 x table, elem_tmp (u64[capacity])
struct GaussScratch {
    uint32_t *btot, *bcur;  // header (K1 zero-fills it)
    uint32_t *gpos, *gmap, *bsum, *drange;
    uint32_t* block_D;      // [5][ceil(N / 256)] K1's per-workgroup sums of tiles_touched, fingerprint lo, hi, min / max valid depth key
    unsigned long long* pair_a;     // [N] (key << 32 | index) partitioned into depth bins
    uint32_t *key_b, *val_b;       // [N] depth bits / Gaussian index in rank order
    uint32_t *loc_b;               // [N] exclusive scan of tiles_touched in rank order INSIDE the rank's sort group (K2); K3 adds the
                                   // group's prefix and writes geom->offsets -- from this read-only source, so K3 can run again
    size_t header_bytes;
};
inline size_t zero_header_bytes(int passes, size_t extra) { return align256((size_t)passes * RS_ZERO_WORDS * 4 + extra); }
inline size_t depth_header_bytes() { return align256(((size_t)2 * (DS_NB_MAX + 1) + 2) * 4); }
inline GaussScratch gauss_scratch(void* base, int N) {
    GaussScratch g;
    char* p = (char*)base;
    g.header_bytes = depth_header_bytes();
    g.btot = (uint32_t*)p; g.bcur = g.btot + (DS_NB_MAX + 1);
    p += g.header_bytes;
    g.gpos = (uint32_t*)p; p += align256((size_t)(DS_NB_MAX + 2) * 4);
    g.gmap = (uint32_t*)p; p += align256((size_t)(DS_NB_MAX + 1) * 4);
    g.bsum = (uint32_t*)p; p += align256((size_t)(DS_NB_MAX + 1) * 4);
    g.drange = (uint32_t*)p; p += 256;
    const size_t nb = align256((size_t)(N > 0 ? N : 1) * 4);
    g.pair_a = (unsigned long long*)p; p += 2 * nb;
    g.key_b = (uint32_t*)p; p += nb;
    g.val_b = (uint32_t*)p; p += nb;
    g.loc_b = (uint32_t*)p; p += nb;
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
    return depth_header_bytes() + align256((size_t)(DS_NB_MAX + 2) * 4) + 2 * align256((size_t)(DS_NB_MAX + 1) * 4) + 256
         + 5 * align256(n * 4) + align256(((n + TG_BLOCK - 1) / TG_BLOCK) * (size_t)DS_BLK_WORDS * 4);
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
uint32_t* bin_header_ptr(const TexGSGeom* g, int N, int* words) {       // bin totals / fill counters of the depth sort: K1 zero-fills them
    const GaussScratch gs = gauss_scratch(g->scan_temp, N);
    *words = (int)(gs.header_bytes / 4);
    return gs.btot;
}

// Gaussian level: depth sort + exclusive scan of tiles_touched in rank order (three launches, see above).  Needs K1's depth keys
// (bits of view z; 0xFFFFFFFF for culled) in g->depth and its per-workgroup (min, max).  Independent of D: runs while the host
// waits for the D readback.  Result (all in scan_temp): (key_b, val_b) = depth bits / Gaussian index in rank order, loc_b = the
// prefix inside each rank's sort group; K3 completes it with the scan of the groups' tile sums and writes g->offsets.
int launch_depth_sort_scan(const TexGSGeom* g, int N, hipStream_t s) {
    if (N <= 0) return 0;
    const GaussScratch gs = gauss_scratch(g->scan_temp, N);
    const int nblk = (N + TG_BLOCK - 1) / TG_BLOCK;
    const int blocks = (N + DS_PER - 1) / DS_PER;
    const int lb = depth_log2_bins(N);
    const int gmax = depth_max_groups(N, 1 << lb);
    const uint32_t* keys = reinterpret_cast<const uint32_t*>(g->depth);
    hipLaunchKernelGGL(k_depth_count, dim3(blocks), dim3(DS_THREADS), 0, s, keys, N, (const uint32_t*)gs.block_D, nblk, lb, gs.btot);
    hipLaunchKernelGGL(k_depth_scatter, dim3(blocks), dim3(DS_THREADS), 0, s, keys, N, (const uint32_t*)gs.block_D, nblk, lb,
                       (const uint32_t*)gs.btot, gs.bcur, gs.gpos, gs.gmap, gs.drange, gs.pair_a);
    hipLaunchKernelGGL(k_depth_group_sort, dim3(gmax + (N + (int)DS_COPY - 1) / (int)DS_COPY), dim3(DS_THREADS), 0, s, N,
                       (const uint32_t*)gs.gpos, (const uint32_t*)gs.drange, gs.pair_a, gs.key_b, gs.val_b,
                       (const uint32_t*)g->tiles_touched, gs.loc_b, gs.bsum, gmax);
    if (gmax > DS_FUSED_GROUPS) hipLaunchKernelGGL(k_depth_prefix, dim3(1), dim3(1024), 0, s, (const uint32_t*)gs.drange, gs.bsum);
    hipError_t e = hipGetLastError();
    return (int)e;
}

void launch_duplicate(const CamConst& c, const TexGSGeom* g, TexGSBinning* b, hipStream_t s) {
    if (c.N <= 0 || b->num_rendered == 0) return;
    const GaussScratch gs = gauss_scratch(g->scan_temp, c.N);
    const int blocks = (c.N + TG_BLOCK - 1) / TG_BLOCK;
    const bool pre = depth_max_groups(c.N, 1 << depth_log2_bins(c.N)) > DS_FUSED_GROUPS;       // (launch_depth_sort_scan ran k_depth_prefix)
#define K3_LAUNCH(PRE) hipLaunchKernelGGL(k_duplicate<PRE>, dim3(blocks), dim3(TG_BLOCK), 0, s, c.N, c.tiles_x, c.tiles_x * c.tiles_y, (const uint32_t*)gs.val_b, \
                       (const uint32_t*)gs.key_b, (const uint32_t*)gs.loc_b, g->offsets, (const uint32_t*)gs.drange, (const uint32_t*)gs.gmap, (const uint32_t*)gs.bsum, \
                       (const uint32_t*)g->tiles_touched, reinterpret_cast<const uint2*>(g->rect), b->keys_unsorted, reinterpret_cast<uint2*>(b->ranges), \
                       reinterpret_cast<uint32_t*>(b->sort_temp), (int)(zero_header_bytes(TILE_PASSES_MAX, 0) / 4))
    if (pre) K3_LAUNCH(true); else K3_LAUNCH(false);
#undef K3_LAUNCH
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

void launch_ranges(const CamConst& c, TexGSBinning* b, uint32_t* zero_words, int num_zero_words, uint32_t* zero_words2,
                   int num_zero_words2, hipStream_t s) {
    const uint32_t T = (uint32_t)(c.tiles_x * c.tiles_y);
    if (b->num_rendered == 0) (void)hipMemsetAsync(b->ranges, 0, sizeof(uint32_t) * 2 * T, s);     // no K3 ran: every tile is empty
    if (b->num_rendered > 0) {
        const int blocks = (int)((b->num_rendered + TG_BLOCK - 1) / TG_BLOCK);
        hipLaunchKernelGGL(k_ranges, dim3(blocks), dim3(TG_BLOCK), 0, s, b->num_rendered, b->keys_sorted,
                           reinterpret_cast<uint2*>(b->ranges));
    }
    hipLaunchKernelGGL(k_tile_order, dim3(1), dim3(1024), 0, s, T, reinterpret_cast<const uint2*>(b->ranges), b->tile_order,
                       zero_words, num_zero_words, zero_words2, num_zero_words2);
}
