// C-ABI entry points of libtexgs.so (declared in include/texgs.h).  No C++ types or exceptions cross this
// boundary: int return codes + a thread-local error string.
#include "common.h"
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#include <mutex>
#include <vector>

namespace {

thread_local char g_err[512] = "";

// ---- optional HIP-event kernel timing (profiling only; the data path keeps no global state) ----
struct EvRec { int kid; hipEvent_t a, b; };
std::mutex g_prof_mu;
bool g_prof_on = false;
uint32_t g_prof_mask = 0xFFFFFFFFu;       // kernel ids to bracket (an event pair costs ~5 us of stream time: bench.py brackets
                                          // only the dominant kernel inside its timed region)
std::vector<EvRec> g_prof_log;
std::vector<hipEvent_t> g_prof_free;

hipEvent_t prof_event() {
    if (!g_prof_free.empty()) { hipEvent_t e = g_prof_free.back(); g_prof_free.pop_back(); return e; }
    hipEvent_t e; (void)hipEventCreate(&e); return e;
}

struct ProfScope {          // brackets the launches issued inside its lifetime
    int kid; hipStream_t s; hipEvent_t a; bool on;
    ProfScope(int k, hipStream_t st) : kid(k), s(st), a(nullptr), on(false) {
        std::lock_guard<std::mutex> l(g_prof_mu);
        if (g_prof_on && ((g_prof_mask >> k) & 1u)) { on = true; a = prof_event(); (void)hipEventRecord(a, s); }
    }
    ~ProfScope() {
        if (!on) return;
        std::lock_guard<std::mutex> l(g_prof_mu);
        hipEvent_t b = prof_event(); (void)hipEventRecord(b, s);
        g_prof_log.push_back({kid, a, b});
    }
};

int fail(const char* where, hipError_t e) {
    snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(e));
    return (int)e ? (int)e : -1;
}

int fail_msg(const char* msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return -1;
}

// After every launch: cheap async check; with debug=1 also a stream sync (lineage `debug` semantics,
// render/uv_tex_render.py:37).
int check(const TexGSFrame* f, hipStream_t s, const char* where) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(where, e);
    if (f->debug) {
        e = hipStreamSynchronize(s);
        if (e != hipSuccess) return fail(where, e);
    }
    return 0;
}

int validate_frame(const TexGSFrame* f) {
    if (!f) return fail_msg("frame is NULL");
    if (f->image_height <= 0 || f->image_width <= 0) return fail_msg("image size must be positive");
    if (f->image_width > 65535 * TEXGS_TILE || f->image_height > 65535 * TEXGS_TILE) return fail_msg("image too large");
    {   // tile ids are sorted in at most three 8-bit digits and live in the high word of a 64-bit key
        const long long tx = (f->image_width + TEXGS_TILE - 1) / TEXGS_TILE, ty = (f->image_height + TEXGS_TILE - 1) / TEXGS_TILE;
        if (tx * ty > (1ll << 24)) return fail_msg("image too large: more than 2^24 tiles");
    }
    if (f->sh_degree < 0 || f->sh_degree > 3) return fail_msg("sh_degree must be in [0,3]");
    if (f->num_gaussians < 0) return fail_msg("num_gaussians < 0");
    if (f->tex_res <= 0) return fail_msg("tex_res must be positive");
    if (f->tex_res > 7168) return fail_msg("tex_res > 7168: the blend kernels address texels with 32-bit byte offsets (6 R^2 x 12 B < 2^32)");
    if (!f->bg || !f->viewmatrix || !f->projmatrix || !f->campos) return fail_msg("frame device pointers must be non-NULL");
    return 0;
}

}  // namespace

extern "C" {

int texgs_abi_version(void) { return TEXGS_ABI_VERSION; }

#ifndef TEXGS_BUILD_ID
#define TEXGS_BUILD_ID "unknown"
#endif
const char* texgs_build_id(void) { return TEXGS_BUILD_ID; }

const char* texgs_last_error(void) { return g_err; }

size_t texgs_scan_temp_bytes(int32_t num_gaussians) { return scan_temp_bytes(num_gaussians); }

size_t texgs_sort_temp_bytes(uint32_t num_rendered, uint32_t num_tiles) { return sort_temp_bytes(num_rendered, num_tiles); }

size_t texgs_tex_bin_count(int32_t tex_res) { return tex_bin_count(tex_res); }

int texgs_preprocess_forward(const TexGSFrame* frame, const TexGSInputs* in, TexGSGeom* geom, void* stream) {
    if (int r = validate_frame(frame)) return r;
    if (!in || !geom) return fail_msg("NULL argument");
    if (frame->num_gaussians == 0) return 0;
    if (!in->means3D || !in->opacities) return fail_msg("per-Gaussian input pointer is NULL");
    if (in->texture && (!in->uvs || !in->gradient_uvs)) return fail_msg("uvs / gradient_uvs are required with a texture");
    if (in->cov3D_precomp && in->texture) return fail_msg("cov3D_precomp is an input of the untextured surface only (texture must be NULL)");
    if (!in->cov3D_precomp && (!in->scales || !in->rotations)) return fail_msg("scales and rotations (or cov3D_precomp) are required");
    if (geom->scan_temp_bytes < scan_temp_bytes(frame->num_gaussians)) return fail_msg("scan_temp too small");
    hipStream_t s = (hipStream_t)stream;
    const CamConst c = make_cam(frame);
    { ProfScope p(TEXGS_K_PREPROCESS_FWD, s); launch_preprocess_fwd(c, frame, in, geom, s); }
    if (int r = check(frame, s, "preprocess_fwd")) return r;
    return 0;
}

// K2: depth sort of the Gaussians + exclusive scan of tiles_touched in depth-rank order (independent of D)
static int depth_sort_scan(const TexGSFrame* frame, TexGSGeom* geom, hipStream_t s) {
    if (frame->num_gaussians == 0) return 0;
    { ProfScope p(TEXGS_K_SCAN, s);
      if (int r = launch_depth_sort_scan(geom, frame->num_gaussians, s)) return fail("depth sort / scan", (hipError_t)r); }
    return check(frame, s, "depth sort / scan");
}

namespace {
struct Readback { uint32_t* host = nullptr; size_t words = 0; hipEvent_t ev = nullptr; };
constexpr int RB_MAX_DEVICES = 64;
thread_local Readback g_rb_dev[RB_MAX_DEVICES];   // one pinned word + event per (host thread, device): HIP events belong to the
                                                  // device that was current when they were created
}

// D = sum of tiles_touched and the geometry fingerprint (K1 leaves per-workgroup partial sums of both): asynchronous copy into
// pinned memory; with sort_first, K2 -- depth sort + scan, which WRITE geom->offsets and geom->scan_temp -- is launched before
// the host waits, so the device stays busy during the one unavoidable device->host sync.
int texgs_read_num_rendered2(const TexGSGeom* geom, int32_t num_gaussians, uint32_t* host_out, uint64_t* fingerprint_out,
                             int32_t sort_first, void* stream) {
    if (!geom || !host_out) return fail_msg("NULL argument");
    *host_out = 0;
    if (fingerprint_out) *fingerprint_out = 0;
    if (num_gaussians <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= RB_MAX_DEVICES) return fail_msg("hipGetDevice failed");
    Readback& g_rb = g_rb_dev[dev];
    // K1 left five words per workgroup (three partial sums + the depth-key range, which only the device-side sort reads; no atomics,
    // nothing to zero-fill): copy the three sums (3 * nblk words), add them up here
    const size_t nblk = ((size_t)num_gaussians + TG_BLOCK - 1) / TG_BLOCK, nw = 3 * nblk;
    if (g_rb.words < nw) {
        if (g_rb.host) (void)hipHostFree(g_rb.host);
        g_rb.host = nullptr;
        g_rb.words = nw < 4096 ? 4096 : nw * 2;
        if (hipHostMalloc((void**)&g_rb.host, g_rb.words * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) { g_rb.words = 0; return fail_msg("hipHostMalloc failed"); }
    }
    if (!g_rb.ev && hipEventCreateWithFlags(&g_rb.ev, hipEventDisableTiming) != hipSuccess) return fail_msg("hipEventCreate failed");
    hipError_t e = hipMemcpyAsync(g_rb.host, bin_block_sums_ptr(geom, num_gaussians), nw * sizeof(uint32_t), hipMemcpyDeviceToHost, s);
    if (e != hipSuccess) return fail("num_rendered readback", e);
    e = hipEventRecord(g_rb.ev, s);
    if (e != hipSuccess) return fail("num_rendered event", e);
    if (sort_first) {
        TexGSFrame f0; memset(&f0, 0, sizeof(f0)); f0.num_gaussians = num_gaussians;
        if (int r = depth_sort_scan(&f0, const_cast<TexGSGeom*>(geom), s)) return r;
    }
    e = hipEventSynchronize(g_rb.ev);
    if (e != hipSuccess) return fail("num_rendered sync", e);
    unsigned long long total = 0ull;
    uint32_t fa = 0u, fb = 0u;
    for (size_t k = 0; k < nblk; ++k) { total += g_rb.host[k]; fa += g_rb.host[nblk + k]; fb += g_rb.host[2 * nblk + k]; }
    if (total > 0xFFFFFFFFull) return fail_msg("num_rendered exceeds 2^32 - 1 instances");
    *host_out = (uint32_t)total;
    if (fingerprint_out) *fingerprint_out = ((uint64_t)fb << 32) | (uint64_t)fa;
    return 0;
}

// The same readback in two steps, for a caller that issues K1 of a LATER view early (texgs.rasterizer forward prefetch): `begin` copies
// K1's per-workgroup partial sums into the caller's PINNED host buffer asynchronously (and launches K2 with sort_first), the caller
// records an event of its own behind it and goes on; `reduce` -- host only, after that event has completed -- adds them up.
size_t texgs_num_rendered_words(int32_t num_gaussians) {
    return num_gaussians <= 0 ? 0 : 3 * (((size_t)num_gaussians + TG_BLOCK - 1) / TG_BLOCK);
}
int texgs_num_rendered_begin(const TexGSGeom* geom, int32_t num_gaussians, uint32_t* host_pinned, size_t host_words, int32_t sort_first,
                             void* stream) {
    if (!geom || !host_pinned) return fail_msg("NULL argument");
    if (num_gaussians <= 0) return 0;
    const size_t nw = texgs_num_rendered_words(num_gaussians);
    if (host_words < nw) return fail_msg("host buffer too small (texgs_num_rendered_words)");
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemcpyAsync(host_pinned, bin_block_sums_ptr(geom, num_gaussians), nw * sizeof(uint32_t), hipMemcpyDeviceToHost, s);
    if (e != hipSuccess) return fail("num_rendered readback", e);
    if (sort_first) {
        TexGSFrame f0; memset(&f0, 0, sizeof(f0)); f0.num_gaussians = num_gaussians;
        if (int r = depth_sort_scan(&f0, const_cast<TexGSGeom*>(geom), s)) return r;
    }
    return 0;
}
int texgs_num_rendered_reduce(const uint32_t* host_pinned, int32_t num_gaussians, uint32_t* host_out, uint64_t* fingerprint_out) {
    if (!host_pinned || !host_out) return fail_msg("NULL argument");
    *host_out = 0;
    if (fingerprint_out) *fingerprint_out = 0;
    if (num_gaussians <= 0) return 0;
    const size_t nblk = ((size_t)num_gaussians + TG_BLOCK - 1) / TG_BLOCK;
    unsigned long long total = 0ull;
    uint32_t fa = 0u, fb = 0u;
    for (size_t k = 0; k < nblk; ++k) { total += host_pinned[k]; fa += host_pinned[nblk + k]; fb += host_pinned[2 * nblk + k]; }
    if (total > 0xFFFFFFFFull) return fail_msg("num_rendered exceeds 2^32 - 1 instances");
    *host_out = (uint32_t)total;
    if (fingerprint_out) *fingerprint_out = ((uint64_t)fb << 32) | (uint64_t)fa;
    return 0;
}

int texgs_read_num_rendered(const TexGSGeom* geom, int32_t num_gaussians, uint32_t* host_out, void* stream) {
    return texgs_read_num_rendered2(geom, num_gaussians, host_out, nullptr, 1, stream);
}

int texgs_depth_sort_scan(TexGSGeom* geom, int32_t num_gaussians, void* stream) {
    if (!geom) return fail_msg("NULL argument");
    TexGSFrame f0; memset(&f0, 0, sizeof(f0)); f0.num_gaussians = num_gaussians;
    return depth_sort_scan(&f0, geom, (hipStream_t)stream);
}

static int render_forward_impl(const TexGSFrame* frame, const TexGSInputs* in, const TexGSGeom* geom,
                               const TexGSBinning* bin, TexGSImage* img, void* stream, bool counters_zeroed) {
    if (int r = validate_frame(frame)) return r;
    if (!in || !geom || !bin || !img) return fail_msg("NULL argument");
    hipStream_t s = (hipStream_t)stream;
    const CamConst c = make_cam(frame);
    if (img->tex_bin_count && in->texture && !counters_zeroed) {       // K6 counts the texture-gradient footprints per bin into it
        hipError_t e = hipMemsetAsync(img->tex_bin_count, 0, sizeof(uint32_t) * 2 * tex_bin_count(c.R), s);
        if (e != hipSuccess) return fail("tex_bin_count memset", e);
    }
    if (img->item_pages || img->item_link || img->item_tail || img->item_ctl) {       // the K6 -> K7 item stream: all or nothing
        if (!img->item_pages || !img->item_link || !img->item_tail || !img->item_ctl)
            return fail_msg("TexGSImage.item_pages / item_link / item_tail / item_ctl must be all NULL or all set");
        if (!img->survivors) return fail_msg("the item stream needs the survivor hand-off beside it (its fallback)");
        const uint32_t np = img->item_sub_pools;
        if (np == 0u || np > TEXGS_ITEM_MAX_POOLS || (np & (np - 1u)) != 0u) return fail_msg("item_sub_pools must be a power of two in [1, 64]");
        if (img->item_page_cap / np < 2u) return fail_msg("item_page_cap must hold at least two pages per sub-pool");
        if (!counters_zeroed) {
            hipError_t e = hipMemsetAsync(img->item_ctl, 0, sizeof(uint32_t) * TEXGS_ITEM_CTL_WORDS, s);
            if (e != hipSuccess) return fail("item_ctl memset", e);
        }
    }
    { ProfScope p(TEXGS_K_RENDER_FWD, s); launch_render_fwd(c, frame, in, geom, bin, img, s); }
    return check(frame, s, "render_fwd");
}

int texgs_render_forward(const TexGSFrame* frame, const TexGSInputs* in, const TexGSGeom* geom,
                         const TexGSBinning* bin, TexGSImage* img, void* stream) {
    return render_forward_impl(frame, in, geom, bin, img, stream, false);
}

int texgs_bin_sort_render_forward(const TexGSFrame* frame, const TexGSInputs* in, const TexGSGeom* geom,
                                  TexGSBinning* bin, TexGSImage* img, void* stream) {
    if (int r = validate_frame(frame)) return r;
    if (!in || !geom || !bin || !img) return fail_msg("NULL argument");
    hipStream_t s = (hipStream_t)stream;
    const CamConst c = make_cam(frame);
    if (bin->num_rendered > 0) {
        if (bin->sort_temp_bytes < sort_temp_bytes(bin->num_rendered, (uint32_t)(c.tiles_x * c.tiles_y))) return fail_msg("sort_temp too small");
        { ProfScope p(TEXGS_K_DUPLICATE, s); launch_duplicate(c, geom, bin, s); }
        if (int r = check(frame, s, "duplicate_with_keys")) return r;
        { ProfScope p(TEXGS_K_SORT, s);
          if (int r = launch_sort(c, geom, bin, s)) return fail("tile sort", (hipError_t)r); }
        if (int r = check(frame, s, "tile sort")) return r;
    }
    // (the one-workgroup tile-order kernel also zero-fills the per-bin footprint counters K6 adds into)
    { ProfScope p(TEXGS_K_RANGES, s);
      launch_ranges(c, bin, img->tex_bin_count, img->tex_bin_count ? 2 * (int)tex_bin_count(c.R) : 0,
                    img->item_ctl, img->item_ctl ? TEXGS_ITEM_CTL_WORDS : 0, s); }
    if (int r = check(frame, s, "tile_ranges")) return r;
    return render_forward_impl(frame, in, geom, bin, img, stream, true);
}

int texgs_forward(const TexGSFrame* frame, const TexGSInputs* in, TexGSGeom* geom, TexGSBinning* bin, uint32_t capacity,
                  TexGSImage* img, uint32_t* num_rendered_out, void* stream) {
    if (!num_rendered_out || !bin) return fail_msg("NULL argument");
    if (int r = texgs_preprocess_forward(frame, in, geom, stream)) return r;
    if (int r = texgs_read_num_rendered(geom, frame->num_gaussians, num_rendered_out, stream)) return r;
    bin->num_rendered = *num_rendered_out;
    if (*num_rendered_out > capacity) {
        snprintf(g_err, sizeof(g_err), "num_rendered %u exceeds binning capacity %u", *num_rendered_out, capacity);
        return TEXGS_ERR_CAPACITY;
    }
    return texgs_bin_sort_render_forward(frame, in, geom, bin, img, stream);
}

int texgs_backward_render(const TexGSFrame* frame, const TexGSInputs* in, const TexGSGeom* geom,
                          const TexGSBinning* bin, const TexGSImage* img, TexGSGrads* grads, void* stream) {
    if (int r = validate_frame(frame)) return r;
    if (!in || !geom || !bin || !img || !grads) return fail_msg("NULL argument");
    if (!(grads->want & TEXGS_WANT_ALL)) return fail_msg("TexGSGrads.want is empty: ask for TEXGS_WANT_TEXTURE and / or TEXGS_WANT_GAUSSIANS");
    if ((grads->want & TEXGS_WANT_GAUSSIANS) && !grads->acc) return fail_msg("acc must be allocated (zero-filled) when Gaussian gradients are wanted");
    if ((grads->want & TEXGS_WANT_TEXTURE) && in->texture && !grads->dL_dtexture) return fail_msg("dL_dtexture must be allocated (zero-filled) when the texture gradient is wanted");
    if (!img->survivors || !img->surv_qmask || !img->surv_count)
        return fail_msg("the forward of this call left no survivor lists (TexGSImage.survivors / surv_qmask / surv_count were NULL): "
                        "run the forward with them to be able to run its backward");
    hipStream_t s = (hipStream_t)stream;
    const CamConst c = make_cam(frame);
    if (bin->num_rendered > 0) {
        { ProfScope p(TEXGS_K_RENDER_BWD, s); launch_render_bwd(c, frame, in, geom, bin, img, grads, s); }
        if (int r = check(frame, s, "render_bwd")) return r;
        if (tex_bins_enabled(c, in, img, grads)) {
            { ProfScope p(TEXGS_K_TEXGRAD_REDUCE, s); launch_texgrad_reduce(c, img, grads, s); }
            if (int r = check(frame, s, "texgrad_reduce")) return r;
        }
    }
    return 0;
}

int texgs_backward_preprocess(const TexGSFrame* frame, const TexGSInputs* in, const TexGSGeom* geom, TexGSGrads* grads,
                              void* stream) {
    if (int r = validate_frame(frame)) return r;
    if (!in || !geom || !grads) return fail_msg("NULL argument");
    if (!(grads->want & TEXGS_WANT_GAUSSIANS)) return 0;          // nobody reads a per-Gaussian gradient: K7 summed no moments either
    if (!grads->acc) return fail_msg("acc must be allocated");
    if (in->cov3D_precomp && !grads->dL_dcov3D) return fail_msg("dL_dcov3D is required with cov3D_precomp");
    hipStream_t s = (hipStream_t)stream;
    const CamConst c = make_cam(frame);
    { ProfScope p(TEXGS_K_PREPROCESS_BWD, s); launch_preprocess_bwd(c, frame, in, geom, grads, s); }
    return check(frame, s, "preprocess_bwd");
}

// (Round 5, measured null: the texture-gradient reduce on a side stream next to K8 -- both wait only for K7 -- changed nothing:
// reference call pattern 728.0 vs 728.2 and 720.7 vs 712.5 views/s, iteration leg 5.172 vs 5.170 ms, profiles/r05_ablation.md.)
int texgs_backward(const TexGSFrame* frame, const TexGSInputs* in, const TexGSGeom* geom,
                   const TexGSBinning* bin, const TexGSImage* img, TexGSGrads* grads, void* stream) {
    if (int r = texgs_backward_render(frame, in, geom, bin, img, grads, stream)) return r;
    return texgs_backward_preprocess(frame, in, geom, grads, stream);
}

int texgs_profile_enable(int on) {
    std::lock_guard<std::mutex> l(g_prof_mu);
    g_prof_on = on != 0;
    if (!g_prof_on) {
        for (auto& r : g_prof_log) { g_prof_free.push_back(r.a); g_prof_free.push_back(r.b); }
        g_prof_log.clear();
    }
    return 0;
}

int texgs_profile_select(uint32_t kernel_mask) {
    std::lock_guard<std::mutex> l(g_prof_mu);
    g_prof_mask = kernel_mask;
    return 0;
}

int texgs_profile_read(float* ms_sum_host, uint32_t* launches_host) {
    if (!ms_sum_host || !launches_host) return fail_msg("NULL argument");
    std::lock_guard<std::mutex> l(g_prof_mu);
    for (auto& r : g_prof_log) {
        hipError_t e = hipEventSynchronize(r.b);
        if (e != hipSuccess) return fail("profile event sync", e);
        float ms = 0.f;
        e = hipEventElapsedTime(&ms, r.a, r.b);
        if (e != hipSuccess) return fail("profile elapsed", e);
        if (r.kid >= 0 && r.kid < TEXGS_NUM_KERNELS) { ms_sum_host[r.kid] += ms; launches_host[r.kid] += 1; }
        g_prof_free.push_back(r.a); g_prof_free.push_back(r.b);
    }
    g_prof_log.clear();
    return 0;
}

int texgs_rgb_alpha_loss(const float* image, const float* gt_image, const float* alpha, const float* gt_alpha,
                         int32_t H, int32_t W, float lambda_dssim, float lambda_alpha, float* scratch, float* sums,
                         float* dL_dimage, float* dL_dalpha, void* stream) {
    if (!image || !gt_image || !scratch || !sums || !dL_dimage) return fail_msg("NULL argument");
    if (H <= 0 || W <= 0) return fail_msg("image size must be positive");
    hipStream_t s = (hipStream_t)stream;
    launch_rgb_alpha_loss(image, gt_image, alpha, gt_alpha, H, W, lambda_dssim, lambda_alpha, scratch, sums, dL_dimage,
                          dL_dalpha, s);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail("rgb_alpha_loss", e);
}

int texgs_geom_losses(const float* norm, const float* gt_norm, const float* gt_image, const float* mask, const float* depth,
                      const float* gt_depth, int32_t H, int32_t W, float lambda_norm, float lambda_smooth, float gamma,
                      float lambda_depth, float* sums, float* dL_dnorm, float* dL_ddepth, void* stream) {
    if (!sums) return fail_msg("NULL argument");
    if (H <= 0 || W <= 0 || !(gamma > 0.f)) return fail_msg("image size and gamma must be positive");
    if (lambda_norm != 0.f && (!norm || !gt_norm || !dL_dnorm)) return fail_msg("norm term needs norm, gt_norm, dL_dnorm");
    if (lambda_smooth != 0.f && (!norm || !gt_image || !dL_dnorm)) return fail_msg("smoothness term needs norm, gt_image, dL_dnorm");
    if (lambda_depth != 0.f && (!depth || !gt_depth || !dL_ddepth)) return fail_msg("depth term needs depth, gt_depth, dL_ddepth");
    launch_geom_losses(norm, gt_norm, gt_image, mask, depth, gt_depth, H, W, lambda_norm, lambda_smooth, gamma, lambda_depth, sums,
                       dL_dnorm, dL_ddepth, (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail("geom_losses", e);
}

int texgs_norm_from_depth(const float* depth, const float* viewmatrix, float tanfovx, float tanfovy, int32_t H, int32_t W,
                          float threshold, float* out_norm, float* out_mask, void* stream) {
    if (!depth || !viewmatrix || !out_norm || !out_mask) return fail_msg("NULL argument");
    if (H <= 0 || W <= 0) return fail_msg("image size must be positive");
    launch_norm_from_depth(depth, viewmatrix, tanfovx, tanfovy, H, W, threshold, out_norm, out_mask, (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail("norm_from_depth", e);
}

size_t texgs_uv_taylor_temp_bytes(void) { return uv_taylor_temp_bytes(); }

int texgs_uv_taylor(const TexGSUVNet* net, const float* xyz, int32_t N, float* uvs, float* grad_uvs, void* temp, void* stream) {
    if (!net || !xyz || !uvs || !grad_uvs || !temp) return fail_msg("NULL argument");
    if (net->hidden != 128) return fail_msg("texgs_uv_taylor supports the shipped UVNet shape only (hidden width 128)");
    if (!net->W1 || !net->W2 || !net->W3 || !net->W4 || !net->W5 || !net->emb) return fail_msg("weight pointer is NULL");
    if (N < 0) return fail_msg("N < 0");
    if (int r = launch_uv_taylor(net, xyz, N, uvs, grad_uvs, temp, (hipStream_t)stream)) return fail("uv_taylor", (hipError_t)r);
    return 0;
}

static int check_uvnet(const TexGSUVNet* net) {
    if (net->hidden != 128) return fail_msg("texgs_uv_taylor supports the shipped UVNet shape only (hidden width 128)");
    if (!net->W1 || !net->W2 || !net->W3 || !net->W4 || !net->W5 || !net->emb) return fail_msg("weight pointer is NULL");
    return 0;
}

int texgs_uv_pack(const TexGSUVNet* net, void* packed, void* stream) {
    if (!net || !packed) return fail_msg("NULL argument");
    if (int r = check_uvnet(net)) return r;
    if (int r = launch_uv_pack(net, packed, (hipStream_t)stream)) return fail("uv_pack", (hipError_t)r);
    return 0;
}

int texgs_uv_taylor_packed(const TexGSUVNet* net, const void* packed, const float* xyz, int32_t N, float* uvs, float* grad_uvs,
                           void* stream) {
    if (!net || !packed || !xyz || !uvs || !grad_uvs) return fail_msg("NULL argument");
    if (int r = check_uvnet(net)) return r;
    if (N < 0) return fail_msg("N < 0");
    if (int r = launch_uv_taylor_packed(net, packed, xyz, N, uvs, grad_uvs, (hipStream_t)stream)) return fail("uv_taylor", (hipError_t)r);
    return 0;
}

int texgs_uv_pack_bf16x3(const TexGSUVNet* net, void* packed, void* stream) {
    if (!net || !packed) return fail_msg("NULL argument");
    if (int r = check_uvnet(net)) return r;
    if (int r = launch_uv_pack_bf16x3(net, packed, (hipStream_t)stream)) return fail("uv_pack_bf16x3", (hipError_t)r);
    return 0;
}

int texgs_uv_taylor_packed_bf16x3(const TexGSUVNet* net, const void* packed, const float* xyz, int32_t N, float* uvs, float* grad_uvs,
                                  void* stream) {
    if (!net || !packed || !xyz || !uvs || !grad_uvs) return fail_msg("NULL argument");
    if (int r = check_uvnet(net)) return r;
    if (N < 0) return fail_msg("N < 0");
    if (int r = launch_uv_taylor_packed_bf16x3(net, packed, xyz, N, uvs, grad_uvs, (hipStream_t)stream)) return fail("uv_taylor_bf16x3", (hipError_t)r);
    return 0;
}

// (declared here, not in common.h: csrc/uvnet.hip's mixed-precision launchers -- C++ linkage like the rest of common.h's)
extern "C++" {
int launch_uv_pack_mixed(const TexGSUVNet* net, void* packed, hipStream_t s);
int launch_uv_taylor_packed_mixed(const TexGSUVNet* net, const void* packed, const float* xyz, int N, float* uvs, float* grad_uvs, hipStream_t s);
}

int texgs_uv_pack_mixed(const TexGSUVNet* net, void* packed, void* stream) {
    if (!net || !packed) return fail_msg("NULL argument");
    if (int r = check_uvnet(net)) return r;
    if (int r = launch_uv_pack_mixed(net, packed, (hipStream_t)stream)) return fail("uv_pack_mixed", (hipError_t)r);
    return 0;
}

int texgs_uv_taylor_packed_mixed(const TexGSUVNet* net, const void* packed, const float* xyz, int32_t N, float* uvs, float* grad_uvs,
                                 void* stream) {
    if (!net || !packed || !xyz || !uvs || !grad_uvs) return fail_msg("NULL argument");
    if (int r = check_uvnet(net)) return r;
    if (N < 0) return fail_msg("N < 0");
    if (int r = launch_uv_taylor_packed_mixed(net, packed, xyz, N, uvs, grad_uvs, (hipStream_t)stream)) return fail("uv_taylor_mixed", (hipError_t)r);
    return 0;
}

size_t texgs_uv_backward_temp_bytes(int32_t N) { return uv_backward_temp_bytes(N < 0 ? 0 : N); }

static int uv_backward_impl(const TexGSUVNet* net, const float* xyz, const float* g_uvs, int32_t N, const TexGSUVNetGrad* out, void* temp,
                            int mixed, void* stream) {
    if (!net || !out || !temp) return fail_msg("NULL argument");
    if (int r = check_uvnet(net)) return r;
    if (N < 0) return fail_msg("N < 0");
    if (N > 0 && (!xyz || !g_uvs)) return fail_msg("NULL argument");
    if (int r = launch_uv_backward(net, xyz, g_uvs, N, out, temp, mixed, (hipStream_t)stream)) return fail("uv_backward", (hipError_t)r);
    return 0;
}
int texgs_uv_backward(const TexGSUVNet* net, const float* xyz, const float* g_uvs, int32_t N, const TexGSUVNetGrad* out, void* temp,
                      void* stream) {
    return uv_backward_impl(net, xyz, g_uvs, N, out, temp, 0, stream);
}
int texgs_uv_backward_mixed(const TexGSUVNet* net, const float* xyz, const float* g_uvs, int32_t N, const TexGSUVNetGrad* out, void* temp,
                            void* stream) {
    return uv_backward_impl(net, xyz, g_uvs, N, out, temp, 1, stream);
}

int texgs_selftest_waveops(const float* seed128, float* out576, void* stream) {
    if (!seed128 || !out576) return fail_msg("NULL argument");
    launch_selftest_waveops(seed128, out576, (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail("selftest_waveops", e);
}

int texgs_mark_visible(const TexGSFrame* frame, const float* means3D, uint8_t* visible, void* stream) {
    if (int r = validate_frame(frame)) return r;
    if (!means3D || !visible) return fail_msg("NULL argument");
    hipStream_t s = (hipStream_t)stream;
    launch_mark_visible(frame, means3D, visible, s);
    return check(frame, s, "mark_visible");
}

}  // extern "C"
