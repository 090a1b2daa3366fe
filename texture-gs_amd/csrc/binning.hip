// K2 inclusive scan, K3 duplicate-with-keys, K4 radix sort, K5 tile ranges -- gfx950.
// Integer / byte work, HBM-bound.  Bit-exact contract: instance k of Gaussian i
// lives at offsets[i-1]+k and covers rect cell k in row-major order; key = (tile_id << 32) | bits(depth);
// the sort is a stable LSD radix sort over bits [0, 32 + bits(T)), so equal keys keep Gaussian-index order.
#include "common.h"
#include <cstring>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_radix_sort.hpp>

namespace {

// K3: one wave-lane per Gaussian; each lane walks its tile rectangle.  Writes are 8 B + 4 B per instance.
__global__ void __launch_bounds__(TG_BLOCK)
k_duplicate(int N, int tiles_x, const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ tiles_touched,
            const uint2* __restrict__ rect, const float* __restrict__ depth,
            uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const int i = blockIdx.x * TG_BLOCK + threadIdx.x;
    if (i >= N) return;
    const uint32_t cnt = tiles_touched[i];
    if (cnt == 0) return;
    uint32_t off = offsets[i] - cnt;                 // exclusive offset
    const uint2 rc = rect[i];
    const uint32_t x0 = rc.x & 0xffffu, y0 = rc.x >> 16, x1 = rc.y & 0xffffu, y1 = rc.y >> 16;
    const uint64_t dbits = (uint64_t)__float_as_uint(depth[i]);
    for (uint32_t y = y0; y < y1; ++y)
        for (uint32_t x = x0; x < x1; ++x) {
            keys[off] = ((uint64_t)(y * (uint32_t)tiles_x + x) << 32) | dbits;
            vals[off] = (uint32_t)i;
            ++off;
        }
}

// K5: ranges[tile] = [first, last) by boundary detection on the sorted keys' high word.
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

// Launch order of the blend kernels: longest tile list first (LPT).  key = 0xFFFF - min(len / 4, 0xFFFF).
__global__ void __launch_bounds__(TG_BLOCK)
k_order_keys(uint32_t T, const uint2* __restrict__ ranges, uint32_t* __restrict__ keys, uint32_t* __restrict__ ids) {
    const uint32_t i = blockIdx.x * TG_BLOCK + threadIdx.x;
    if (i >= T) return;
    const uint2 r = ranges[i];
    const uint32_t len = r.y - r.x;
    keys[i] = 0xFFFFu - min(len >> 2, 0xFFFFu);
    ids[i] = i;
}

inline int key_end_bit(uint32_t T) {
    int bits = 0;
    while ((1u << bits) < T && bits < 31) ++bits;     // ceil(log2 T)
    if (bits == 0) bits = 1;
    return 32 + bits;
}

}  // namespace

size_t scan_temp_bytes(int N) {
    size_t bytes = 0;
    if (N <= 0) return 256;
    (void)rocprim::inclusive_scan(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)N,
                                  rocprim::plus<uint32_t>(), (hipStream_t)0);
    return bytes < 256 ? 256 : bytes;
}

int launch_scan(const TexGSGeom* g, int N, hipStream_t s) {
    if (N <= 0) return 0;
    size_t bytes = g->scan_temp_bytes;
    hipError_t e = rocprim::inclusive_scan(g->scan_temp, bytes, (const uint32_t*)g->tiles_touched, g->offsets,
                                           (size_t)N, rocprim::plus<uint32_t>(), s);
    return e == hipSuccess ? 0 : (int)e;
}

static size_t order_temp_bytes(uint32_t T) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr,
                                    (uint32_t*)nullptr, (size_t)T, 0u, 16u, (hipStream_t)0);
    return bytes < 256 ? 256 : bytes;
}

size_t sort_temp_bytes(uint32_t D, uint32_t T) {
    size_t bytes = 0;
    if (D == 0) return order_temp_bytes(T);
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr,
                                    (const uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)D, 0u,
                                    (unsigned)key_end_bit(T), (hipStream_t)0);
    const size_t ob = order_temp_bytes(T);
    if (bytes < ob) bytes = ob;
    return bytes < 256 ? 256 : bytes;
}

void launch_duplicate(const CamConst& c, const TexGSGeom* g, TexGSBinning* b, hipStream_t s) {
    if (c.N <= 0 || b->num_rendered == 0) return;
    const int blocks = (c.N + TG_BLOCK - 1) / TG_BLOCK;
    hipLaunchKernelGGL(k_duplicate, dim3(blocks), dim3(TG_BLOCK), 0, s, c.N, c.tiles_x, g->offsets, g->tiles_touched,
                       reinterpret_cast<const uint2*>(g->rect), g->depth, b->keys_unsorted, b->vals_unsorted);
}

int launch_sort(const CamConst& c, TexGSBinning* b, hipStream_t s) {
    if (b->num_rendered == 0) return 0;
    size_t bytes = b->sort_temp_bytes;
    const uint32_t T = (uint32_t)(c.tiles_x * c.tiles_y);
    hipError_t e = rocprim::radix_sort_pairs(b->sort_temp, bytes, (const uint64_t*)b->keys_unsorted, b->keys_sorted,
                                             (const uint32_t*)b->vals_unsorted, b->point_list,
                                             (size_t)b->num_rendered, 0u, (unsigned)key_end_bit(T), s);
    return e == hipSuccess ? 0 : (int)e;
}

void launch_ranges(const CamConst& c, TexGSBinning* b, hipStream_t s) {
    const uint32_t T = (uint32_t)(c.tiles_x * c.tiles_y);
    if (b->num_rendered > 0) {
        const int blocks = (int)((b->num_rendered + TG_BLOCK - 1) / TG_BLOCK);
        hipLaunchKernelGGL(k_ranges, dim3(blocks), dim3(TG_BLOCK), 0, s, b->num_rendered, b->keys_sorted,
                           reinterpret_cast<uint2*>(b->ranges));
    }
    // tile launch order (stable: equal lengths keep tile-index order)
    uint32_t* keys_in = b->order_keys; uint32_t* keys_out = b->order_keys + T;
    uint32_t* ids_in = b->order_keys + 2 * T;                             // sorted ids land here, then copied back
    hipLaunchKernelGGL(k_order_keys, dim3((T + TG_BLOCK - 1) / TG_BLOCK), dim3(TG_BLOCK), 0, s, T,
                       reinterpret_cast<const uint2*>(b->ranges), keys_in, b->tile_order);
    (void)ids_in;
    size_t bytes = b->sort_temp_bytes;
    // in-place on values is not allowed: sort (keys_in, tile_order) -> (keys_out, order_keys scratch) then copy back
    (void)rocprim::radix_sort_pairs(b->sort_temp, bytes, (const uint32_t*)keys_in, keys_out, (const uint32_t*)b->tile_order,
                                    ids_in, (size_t)T, 0u, 16u, s);
    (void)hipMemcpyAsync(b->tile_order, ids_in, sizeof(uint32_t) * T, hipMemcpyDeviceToDevice, s);
}
