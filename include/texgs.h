/*
 * texgs.h -- C ABI of libtexgs.so, the MI355X (gfx950) textured Gaussian rasterizer.
 *
 * This is the drop-in boundary for the ONE hot path of slothfulxtx/Texture-GS: the operator the
 * reference imports at render/uv_tex_render.py:4 (`from diff_gauss_uv_tex import
 * GaussianRasterizationSettings, GaussianRasterizer`) and calls at render/uv_tex_render.py:56-66,
 * plus its autograd backward (triggered at models/texture_gaussian3d.py:410).  In the reference that
 * module is a pybind11 extension (`_C.rasterize_gaussians` / `_C.rasterize_gaussians_backward`, not in
 * the reference tree -- un-vendored pip dependency, requirements.txt:15); the entry points below are
 * what a binding for this path binds instead.  Plain pointers and sizes only: no torch / C++ types.
 *
 * All pointers are DEVICE pointers unless a field says "host".  All buffers are caller-allocated
 * (the Python host layer allocates them through torch's caching allocator so stream semantics hold);
 * the library keeps no global or static scratch, so several forwards may be alive before a backward
 * (models/texture_gaussian3d.py:318 and :378 both precede :410).  Kernels are enqueued on the
 * `stream` argument (a hipStream_t passed as void*).  Every function returns 0 on success, non-zero
 * on failure; texgs_last_error() then returns a thread-local message.
 */
#ifndef TEXGS_H
#define TEXGS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TEXGS_ABI_VERSION 15
#define TEXGS_TILE 16          /* 16x16 pixel tiles, one 256-thread workgroup (4 wave64) per tile     */
#define TEXGS_REC_TEST_FLOATS 8    /* per-Gaussian TEST record (32 B): what the per-block culls and the alpha test read  */
#define TEXGS_REC_SHADE_FLOATS 20  /* per-Gaussian SHADING record (80 B): fetched only for Gaussians that survive a cull (one record per
                                      128-byte line, 32 floats, was measured: K6 -6 us, K1 +3, K7 +-0 -- not kept) */
#define TEXGS_ACC_FLOATS 32    /* per-Gaussian moment accumulators of the backward: one 128-byte line  */

/* Per-call configuration = GaussianRasterizationSettings (render/uv_tex_render.py:25-38). */
typedef struct TexGSFrame {
    int32_t image_height;      /* :26 */
    int32_t image_width;       /* :27 */
    float   tanfovx;           /* :28 */
    float   tanfovy;           /* :29 */
    float   scale_modifier;    /* :31 */
    int32_t sh_degree;         /* :35 active degree 0..3                                               */
    int32_t sh_coeffs;         /* shs.shape[1] (15 for max degree 3); 0 when shs is NULL               */
    int32_t tex_res;           /* R of texture[6,R,R,3] (models/texture_gaussian3d.py:51)              */
    int32_t num_gaussians;     /* N                                                                    */
    int32_t debug;             /* :37 -- sync + error check after every launch                         */
    const float* bg;           /* :30 f32[3]                                                           */
    const float* viewmatrix;   /* :32 f32[16], row-vector (transposed) form of utils/cameras.py:62     */
    const float* projmatrix;   /* :33 f32[16], full world->clip, utils/cameras.py:64                   */
    const float* campos;       /* :36 f32[3]                                                           */
} TexGSFrame;

/* Operator inputs = kwargs of render/uv_tex_render.py:56-66 (means2D is a zero grad-carrier, never read). */
typedef struct TexGSInputs {
    const float* means3D;      /* f32[N,3]                                                             */
    const float* shs;          /* f32[N,K,3] view-dependent SH coefficients 1..K, or NULL              */
    const float* opacities;    /* f32[N]   (sigmoid-activated, models/texture_gaussian3d.py:239)       */
    const float* scales;       /* f32[N,3] (exp-activated, :197)                                       */
    const float* rotations;    /* f32[N,4] (w,x,y,z), unit (:201)                                      */
    const float* uvs;          /* f32[N,3] phi(mu) on the unit sphere (:229-236)                       */
    const float* gradient_uvs; /* f32[N,9] [3*i+j] = d uv_i / d x_j (:216-227)                         */
    const float* texture;      /* f32[6,R,R,3] SH-DC valued cubemap (:51, :16-21).  NULL = the UNTEXTURED surface (`diff_gauss`,
                                  render/render.py:75-84): colour = max(0, viewdep + color_offset + 0.5); uvs / gradient_uvs may
                                  then be NULL too, the blend kernels skip the UV step, the cubemap address and the taps, and
                                  TexGSFrame.tex_res is ignored                                                             */
    const float* color_offset; /* f32[N,3] or NULL: added to the view-dependent colour term.  Used by the
                                  untextured `diff_gauss` surface (render/render.py:75-84): C0*SH_DC or
                                  colors_precomp - 0.5                                                    */
    const float* cov3D_precomp;/* f32[N,6] or NULL: world-space covariance (xx,xy,xz,yy,yz,zz -- strip_lowerdiag,
                                  utils/general.py:73-82) used INSTEAD of scales / rotations (render/render.py:52-53,
                                  `cov3Ds_precomp`; scales / rotations may then be NULL and scale_modifier is not applied).
                                  Untextured surface only.  The splat normal is then the eigenvector of the smallest eigenvalue
                                  (a selection: no gradient flows through it)                                           */
} TexGSInputs;

/* Per-Gaussian state written by texgs_preprocess_forward (K1) and texgs_read_num_rendered (K2). */
typedef struct TexGSGeom {
    float*    rec_test;        /* f32[N,8]:  xy(2) conic(-a/2,-b,-c/2) opacity rcull thr -- read for EVERY (8x8 block, instance) pair */
    float*    rec_shade;       /* f32[N,20]: g(2) G(6) | phi(3) viewdep(3) | depth normal(3) xy(2) -- read only for instances
                                  that can reach alpha >= 1/255 somewhere in the block (about a third of them); the last two
                                  words repeat the test record's xy (v15) so that K7's per-item gather needs one record    */
    float*    depth;           /* f32[N] view-space z = the depth sort key (its bit pattern); 0xFFFFFFFF for culled Gaussians */
    int32_t*  radii;           /* i32[N] screen radius in px; 0 = culled (operator output `radii`)     */
    uint32_t* rect;            /* u32[N,2]: (minx | miny<<16), (maxx | maxy<<16) tile rectangle        */
    uint32_t* tiles_touched;   /* u32[N]                                                               */
    uint32_t* offsets;         /* u32[N] EXCLUSIVE prefix sum of tiles_touched in depth-rank order: offsets[r] = first
                                  instance slot of the r-th Gaussian of the (depth bits, index) order.  Written by K3
                                  (texgs_bin_sort_render_forward) from what K2 left in scan_temp; undefined before that.  K3
                                  only reads K2's result, so texgs_bin_sort_render_forward may be called again on it (a retry
                                  with larger buffers, a timing loop)                                                   */
    void*     scan_temp;       /* >= texgs_scan_temp_bytes(N): K1's per-workgroup words, depth bins, depth-sorted (key, index) */
    size_t    scan_temp_bytes;
} TexGSGeom;

/* Tile binning buffers, sized from num_rendered (D). */
typedef struct TexGSBinning {
    uint32_t  num_rendered;    /* D = sum of tiles_touched (host value)                                */
    uint64_t* keys_unsorted;   /* u64[D]  (tile_id << 32) | depth rank, emitted in depth-rank order (K3) */
    uint64_t* keys_sorted;     /* u64[D]  (tile_id << 32) | float_bits(depth), sorted: the lineage's sorted key list */
    uint32_t* point_list;      /* u32[D]  Gaussian index, sorted by (tile, depth, index)               */
    uint32_t* ranges;          /* u32[T,2] [first,last) into point_list per tile (K3 zero-fills it)    */
    uint32_t* tile_order;      /* u32[T] tile ids, longest list first (launch order of the blend kernels) */
    void*     sort_temp;       /* >= texgs_sort_temp_bytes(D, T): tile-pass count tables + u64[D] ping-pong buffer */
    size_t    sort_temp_bytes;
} TexGSBinning;

/* Operator outputs (render/uv_tex_render.py:56) + the per-pixel state the backward replays. */
typedef struct TexGSImage {
    float*    out_color;       /* f32[3,H,W]                                                           */
    float*    out_depth;       /* f32[1,H,W]                                                           */
    float*    out_norm;        /* f32[3,H,W]                                                           */
    float*    out_alpha;       /* f32[1,H,W]                                                           */
    float*    final_T;         /* f32[H,W]                                                             */
    uint32_t* n_contrib;       /* u32[H,W] 1-based position of the last contributor in the tile list   */
    uint32_t* tex_bin_count;   /* u32[2 * texgs_tex_bin_count(R)] or NULL.  Non-NULL = "a backward will follow": K6 (which zero-fills it
                                  first) counts the bilinear footprints per 32x32-texel texture bin, the exact sizes of the
                                  record lists texgs_backward_render builds: [b] = records the 8x8 pixel blocks RESERVED in bin
                                  b's list (tex_bin_resv), [count + b] = overflow footprints, appended behind them (v12).
                                  NULL: forward-only call, or a backward that sends every texture-gradient footprint through
                                  atomics.                                                                              */
    /* K6 -> K7 hand-off, all three non-NULL when a backward will follow (else all NULL): K6 culls every tile list against each
       of the tile's four 8x8 pixel blocks anyway; it leaves the survivors so that K7 replays them instead of culling again. */
    uint32_t* survivors;       /* u32[2 * 4 * capacity] {Gaussian id, list position} pairs; block (tile, w) owns entries
                                  [4 * ranges[tile].first + w * len, ... + len), len = the tile's list length; capacity = the
                                  element count keys_sorted / point_list were sized for                                     */
    uint16_t* surv_qmask;      /* u16[4 * capacity] bit q: the survivor reaches 4x4 quadrant q of its block                  */
    uint32_t* surv_count;      /* u32[4 * T] survivors written per block                                                     */
    uint32_t* tex_bin_resv;    /* u32[4 * T * TEXGS_RESV_WORDS] or NULL (v12; non-NULL together with tex_bin_count).  Per 8x8 pixel
                                  block, K6's RESERVATIONS in the texture-gradient record lists: a direct-mapped table of 64
                                  entries {texture bin (0xFFFFFFFF: free), first record of the block inside that bin's list,
                                  records} as three planes of 64 words, entry = low 3 bits of the bin's x | y inside its face.
                                  K6 takes each range with one returning atomic per (block, bin) on tex_bin_count[bin]; K7
                                  hands the slots out block-locally (an LDS atomic per footprint: no global cursor, no
                                  grouping).  Footprints whose entry belongs to another bin (~2 %) are counted in the second half
                                  of tex_bin_count and appended through TexGSGrads.tex_bin_cursor.  No initialisation; written
                                  for every block.                                                                       */
    /* K6 -> K7 ITEM STREAM (v15), all NULL / 0 or all set.  Set = "a backward that wants per-Gaussian gradients will follow": K6
       appends, per 8x8 pixel block and in blend order (front to back), one 12-byte item {T before the pair, alpha_raw = opacity *
       exp(power), Gaussian id << 6 | pixel lane} for every CONTRIBUTING (pixel, Gaussian) pair -- what the lineage's backward
       recomputes per pair by replaying the tile list -- and K7 walks the block's items back to front instead of
       re-testing the survivor lists: no test loop, no transmittance recurrence (T /= 1 - alpha), decisions identical by
       construction.  Pages of TEXGS_ITEM_PAGE items (three planes of TEXGS_ITEM_PAGE words: T, alpha_raw, key) are taken from
       item_sub_pools sub-pools of the buffer with one returning atomic per page; a block's pages are chained backwards through
       item_link.  A buffer that is too small is not an error: K6 raises item_ctl[TEXGS_ITEM_CTL_FLAG], the stream kernel does
       nothing and the survivor-replay kernel (the hand-off above, always written) runs instead; the cursors keep counting, so
       max(item_ctl[16 * s]) * item_sub_pools is what the view needed (the caller sizes the next buffer from it). */
    uint32_t* item_pages;      /* u32[item_page_cap * 3 * TEXGS_ITEM_PAGE]; no initialisation                                     */
    uint32_t* item_link;       /* u32[item_page_cap]: the block's previous page (0xFFFFFFFF: none); no initialisation              */
    uint32_t* item_tail;       /* u32[4 * T * 2]: per block {last page, items}; written for every block                            */
    uint32_t* item_ctl;        /* u32[TEXGS_ITEM_CTL_WORDS]: sub-pool cursors (one per 64-byte line) + the overflow flag; zero-filled
                                  by the library before K6                                                                         */
    uint32_t  item_page_cap;   /* pages item_pages / item_link hold; sub-pool s owns pages [s * cap / pools, (s + 1) * cap / pools) */
    uint32_t  item_sub_pools;  /* power of two, 1..TEXGS_ITEM_MAX_POOLS (one hot atomic word serialises at ~13 ns per request)    */
} TexGSImage;
#define TEXGS_RESV_WORDS 192
#define TEXGS_ITEM_PAGE 256
#define TEXGS_ITEM_MAX_POOLS 64
#define TEXGS_ITEM_CTL_FLAG (16 * TEXGS_ITEM_MAX_POOLS)
#define TEXGS_ITEM_CTL_WORDS (16 * TEXGS_ITEM_MAX_POOLS + 16)

#define TEXGS_ACC_MEANS3D 1
#define TEXGS_ACC_MEANS2D 2
#define TEXGS_ACC_SHS 4
#define TEXGS_ACC_OPACITIES 8
#define TEXGS_ACC_SCALES 16
#define TEXGS_ACC_ROTATIONS 32
#define TEXGS_ACC_UVS 64
#define TEXGS_ACC_COLOR_OFFSET 128
#define TEXGS_ACC_COV3D 256
#define TEXGS_ACC_ALL 511

/* TexGSGrads.want: which gradients the caller will read.  The backward compiles out what nobody asked for. */
#define TEXGS_WANT_TEXTURE   1  /* dL_dtexture (records + bin reduce, or atomics).  Without it K6 need not count footprints
                                   (TexGSImage.tex_bin_count NULL) and K7 appends nothing                                    */
#define TEXGS_WANT_GAUSSIANS 2  /* every per-Gaussian output (means3D .. color_offset): K7's per-pixel recurrence and moment sums,
                                   and K8.  Without it texgs_backward_preprocess is a no-op and the per-Gaussian outputs are
                                   left untouched (a texture-only optimisation step, models/texture_gaussian3d.py:439-440)  */
#define TEXGS_WANT_ALL       3

/* Backward: upstream grads in, input grads out.  NULL dL_dout pointers mean "zero". */
typedef struct TexGSGrads {
    const float* dL_dcolor;    /* f32[3,H,W] or NULL */
    const float* dL_ddepth;    /* f32[1,H,W] or NULL */
    const float* dL_dnorm;     /* f32[3,H,W] or NULL */
    const float* dL_dalpha;    /* f32[1,H,W] or NULL */
    float* acc;                /* (needed with TEXGS_WANT_GAUSSIANS) f32[N,32] ALL-ZERO on entry and all-zero again on return (K8 clears the rows it read):
                                  per-Gaussian raw moment sums, scratch between the two backward kernels              */
    float* dL_dmeans3D;        /* f32[N,3]                                                             */
    float* dL_dmeans2D;        /* f32[N,3] dL/d(ndc xy), z = 0 (lineage convention)                    */
    float* dL_dshs;            /* f32[N,K,3] or NULL                                                   */
    float* dL_dopacities;      /* f32[N]                                                               */
    float* dL_dscales;         /* f32[N,3]                                                             */
    float* dL_drotations;      /* f32[N,4]                                                             */
    float* dL_duvs;            /* f32[N,3]                                                             */
    float* dL_dtexture;        /* (needed with TEXGS_WANT_TEXTURE) f32[6,R,R,3] caller zero-filled; accumulated into with fp32
                                  atomics.  An inf / NaN upstream colour gradient reaches exactly the texels it touches     */
    float* dL_dcolor_offset;   /* f32[N,3] or NULL                                                     */
    float* dL_dcov3D;          /* f32[N,6] or NULL (required with TexGSInputs.cov3D_precomp when Gaussian gradients are wanted;
                                  off-diagonal entries carry both symmetric halves, as the lineage's do)                 */
    uint32_t want;             /* TEXGS_WANT_* bit mask, non-zero                                      */
    float*    tex_bins;        /* texture-gradient records, 16 bytes each: u32[4 * tex_rec_cap] (TEXGS_TEXBIN_RECORD_FLOATS words per
                                  record), 16-byte aligned, or NULL.  The texture is cut into 32x32-texel blocks ("bins", 6 *
                                  ceil(R/32)^2 of them).  K7 writes one record {fx, fy (18 bits each), tap-00 cell inside the bin
                                  (5 + 5 bits, in the low mantissa bits of r and g), dL/dtexel-colour rgb} per bilinear footprint
                                  with one 16-byte store into the list of the bin the footprint is anchored in (v15; 20 bytes in
                                  five planes until v14); the lists are contiguous and exactly sized from
                                  TexGSImage.tex_bin_count (~0.3 GB for a C3 view).  The reduce kernel at the end of
                                  texgs_backward_render sums each list in LDS and adds every texel to dL_dtexture once.  NULL (or no
                                  counts, or cap 0) = fp32 atomics straight into dL_dtexture (~20 G requests/s memory-side: 0.7 ms
                                  per C3 view).  Contents need no initialisation.                                           */
    uint32_t* tex_bin_cursor;  /* u32[texgs_tex_bin_count(R) + 2]: the fill cursors of the lists' OVERFLOW parts (scratch, set at the
                                  start of every backward: no initialisation) + two status words: [count] receives
                                  max(records a call needed) -- zero it once; never cleared by the library: the caller sizes
                                  tex_rec_cap from it --, [count+1] = bits of max |dL/dpixel colour| of the call in flight
                                  (reset by every backward).                                                              */
    uint32_t* tex_bin_base;    /* u32[2 * texgs_tex_bin_count(R) + 1]: scratch of this call (no initialisation): the list
                                  offsets [count + 1], then the reduce kernel's launch order [count] (v11)                  */
    uint32_t  tex_rec_cap;     /* records tex_bins holds.  Too small is not an error: footprints that do not fit fall back
                                  to atomics.                                                                            */
    int32_t accumulate;        /* bit mask (TEXGS_ACC_*): K8 ADDS into the per-Gaussian outputs whose bit is set (fused gradient
                                  accumulation of a multi-view step, or straight into a leaf's existing .grad; culled Gaussians
                                  write nothing there) and overwrites the others.  dL_dtexture is always accumulated into.   */
} TexGSGrads;

int         texgs_abi_version(void);
/* Identity of the sources this library was built from (v15): 16 hex digits of sha256 over the files of csrc/, include/texgs.h and the compile
 * flags, baked in by texture-gs_amd/build.py.  The Python host layer recomputes it from the tree and refuses a stale library. */
const char* texgs_build_id(void);
const char* texgs_last_error(void);

size_t texgs_scan_temp_bytes(int32_t num_gaussians);
size_t texgs_sort_temp_bytes(uint32_t num_rendered, uint32_t num_tiles);
size_t texgs_tex_bin_count(int32_t tex_res);
#define TEXGS_TEXBIN_RECORD_FLOATS 4

/* K1: frustum cull, EWA projection, radius, tile rect, SH view term, normal, UV Taylor pre-fold, depth sort key, and
 * D = sum of tiles_touched (device word).  Replaces the first half of _C.rasterize_gaussians. */
int texgs_preprocess_forward(const TexGSFrame* frame, const TexGSInputs* in, TexGSGeom* geom, void* stream);

/* The one device->host sync of the forward (the lineage has the same one): starts the asynchronous readback of D,
 * launches K2 -- the sort of the N Gaussians by (depth bits, index) and the exclusive scan of tiles_touched in
 * that order, neither of which depends on D -- and only then waits for D, so the device is busy during the sync. */
int texgs_read_num_rendered(const TexGSGeom* geom, int32_t num_gaussians, uint32_t* host_out, void* stream);

/* The same readback, also returning K1's 64-bit GEOMETRY FINGERPRINT: a hash of everything the tile binning (K2-K5), K6's
 * survivor lists and its per-bin footprint counts are functions of (depth bits, tile rect, test record and the UV-Taylor part of
 * the shading record of every Gaussian).  Two forwards with equal fingerprints, image size and texture resolution have identical
 * TexGSBinning contents, survivor lists and tex_bin_count: the second may run texgs_render_forward on the first one's buffers
 * (the reference renders every training view twice, models/texture_gaussian3d.py:318 and :375-389 -- same camera, same
 * Gaussians, sh_degree 0 the second time).  Equality of a 64-bit hash, i.e. a 2^-64 chance of a false match per comparison.
 * sort_first != 0: launch K2 before waiting (as texgs_read_num_rendered does); 0: only wait -- the caller expects to re-use
 * existing lists and calls texgs_depth_sort_scan itself if the fingerprint turns out different. */
int texgs_read_num_rendered2(const TexGSGeom* geom, int32_t num_gaussians, uint32_t* host_out, uint64_t* fingerprint_out,
                             int32_t sort_first, void* stream);
/* The same readback in two steps (v15), for a caller that issues K1 of a later view early so that no forward ever waits for its own
 * K1: texgs_num_rendered_begin copies K1's partial sums into the caller's PINNED host buffer (>= texgs_num_rendered_words(N) words)
 * asynchronously on `stream` and, with sort_first, launches K2; the caller records an event behind it.  texgs_num_rendered_reduce --
 * host only, after that event completed -- returns D and the geometry fingerprint. */
size_t texgs_num_rendered_words(int32_t num_gaussians);
int texgs_num_rendered_begin(const TexGSGeom* geom, int32_t num_gaussians, uint32_t* host_pinned, size_t host_words, int32_t sort_first,
                             void* stream);
int texgs_num_rendered_reduce(const uint32_t* host_pinned, int32_t num_gaussians, uint32_t* host_out, uint64_t* fingerprint_out);
/* K2 alone: depth sort of the Gaussians + exclusive scan of tiles_touched in depth-rank order (writes geom->offsets, scan_temp). */
int texgs_depth_sort_scan(TexGSGeom* geom, int32_t num_gaussians, void* stream);

/* K3 duplicate-with-keys (in depth-rank order), K4 stable 2-pass radix sort of the D instances by tile id (the list is then
 * ordered by (tile, depth bits, index) exactly like the lineage's one 32+ceil(log2 T)-bit sort), K5 tile ranges + tile
 * launch order, K6 16x16-tile alpha-blend with cubemap fetch.  Second half of _C.rasterize_gaussians. */
int texgs_bin_sort_render_forward(const TexGSFrame* frame, const TexGSInputs* in, const TexGSGeom* geom,
                                  TexGSBinning* bin, TexGSImage* img, void* stream);

/* Whole forward in one call: texgs_preprocess_forward, the num_rendered readback, and -- when D fits `capacity`
 * (the element count the caller sized keys/vals/point_list for; sort_temp_bytes >= texgs_sort_temp_bytes(capacity, T))
 * -- texgs_bin_sort_render_forward, with no host work between the sync and the next launches.  Returns
 * TEXGS_ERR_CAPACITY with *num_rendered_out = D when the buffers are too small: the caller grows them, sets
 * bin->num_rendered = D and calls texgs_bin_sort_render_forward itself. */
#define TEXGS_ERR_CAPACITY 1000
int texgs_forward(const TexGSFrame* frame, const TexGSInputs* in, TexGSGeom* geom, TexGSBinning* bin, uint32_t capacity,
                  TexGSImage* img, uint32_t* num_rendered_out, void* stream);

/* K6 alone on existing binning (re-render with a different texture / sh_degree-independent state). */
int texgs_render_forward(const TexGSFrame* frame, const TexGSInputs* in, const TexGSGeom* geom,
                         const TexGSBinning* bin, TexGSImage* img, void* stream);

/* K7 (back-to-front replay, per-Gaussian partials + texture-gradient records) + bin reduce + K8 (chain to the operator's
 * inputs).  Replaces _C.rasterize_gaussians_backward. */
int texgs_backward(const TexGSFrame* frame, const TexGSInputs* in, const TexGSGeom* geom,
                   const TexGSBinning* bin, const TexGSImage* img, TexGSGrads* grads, void* stream);

/* The two halves of texgs_backward, for callers that pipeline several views over HIP streams: the first (K7 + bin reduce)
 * only touches grads->acc, the texture bins and dL_dtexture (atomics); the second (K8) is the one that writes -- or, with
 * grads->accumulate, read-modify-writes -- the per-Gaussian outputs, so it alone has to be ordered between two views that
 * share a gradient buffer (hipStreamWaitEvent between the two calls).  texgs_backward == render, then preprocess. */
int texgs_backward_render(const TexGSFrame* frame, const TexGSInputs* in, const TexGSGeom* geom,
                          const TexGSBinning* bin, const TexGSImage* img, TexGSGrads* grads, void* stream);
int texgs_backward_preprocess(const TexGSFrame* frame, const TexGSInputs* in, const TexGSGeom* geom, TexGSGrads* grads,
                              void* stream);

/* Optional per-kernel HIP-event timing (used by bench.py for the live roofline figure).  While enabled,
 * every kernel launch of this library is bracketed by hipEventRecord on the launch stream.
 * texgs_profile_read synchronises the recorded events, adds elapsed ms / launch counts per kernel id into
 * the caller's host arrays (length TEXGS_NUM_KERNELS) and clears the log. */
enum {
    TEXGS_K_PREPROCESS_FWD = 0, TEXGS_K_SCAN = 1, TEXGS_K_DUPLICATE = 2, TEXGS_K_SORT = 3, TEXGS_K_RANGES = 4,
    TEXGS_K_RENDER_FWD = 5, TEXGS_K_RENDER_BWD = 6, TEXGS_K_PREPROCESS_BWD = 7, TEXGS_K_TEXGRAD_REDUCE = 8,
    TEXGS_NUM_KERNELS = 9
};
int texgs_profile_enable(int on);
int texgs_profile_select(uint32_t kernel_mask);      /* bit k = bracket kernel id k (default: all) */
int texgs_profile_read(float* ms_sum_host, uint32_t* launches_host);

/* Fused loss front-end for the operator's outputs -- the always-on terms of TextureGaussian3D.compute_loss
 * (models/texture_gaussian3d.py:333-345; losses/pixelwise_loss.py l1_loss, losses/ssim_loss.py:16-54):
 *   loss = (1-l)*mean|I-Igt| + l*(1 - mean SSIM_11x11(I,Igt)) + la*mean|A-Agt|.
 * image/gt_image f32[3,H,W]; alpha/gt_alpha f32[1,H,W] or NULL (then dL_dalpha is untouched); scratch f32[9*H*W];
 * sums f32[4] receives {sum|I-Igt|, sum SSIM, sum|A-Agt|, 0}; dL_dimage / dL_dalpha are written for d(loss) = 1. */
int texgs_rgb_alpha_loss(const float* image, const float* gt_image, const float* alpha, const float* gt_alpha,
                         int32_t H, int32_t W, float lambda_dssim, float lambda_alpha, float* scratch, float* sums,
                         float* dL_dimage, float* dL_dalpha, void* stream);

/* Geometric regularisers of TextureGaussian3D.compute_loss on the operator's normal / depth outputs
 * (models/texture_gaussian3d.py:347-368), the producers of dL/dnorm and dL/ddepth for texgs_backward:
 *   lambda_norm   * norm_loss(norm, gt_norm, mask)        losses/norm_reg_loss.py:66-71
 *   lambda_smooth * smooth_loss(gt_image, norm, mask)     losses/smooth_loss.py:4-27 (bilateral weight exp(-|d rgb|_1 / gamma))
 *   lambda_depth  * l1_loss(depth, gt_depth)              losses/pixelwise_loss.py
 * norm, gt_norm, gt_image f32[3,H,W]; mask f32[1,H,W] or NULL (= ones); depth, gt_depth f32[1,H,W]; a term with lambda 0
 * is skipped and its pointers may be NULL.  sums f32[12] receives {sum m, sum (1-<n,g>) m, sum w_k (4), sum w_k|dn| (4),
 * sum |d-d_gt|, 0}; dL_dnorm / dL_ddepth are written for d(loss) = 1. */
int texgs_geom_losses(const float* norm, const float* gt_norm, const float* gt_image, const float* mask, const float* depth,
                      const float* gt_depth, int32_t H, int32_t W, float lambda_norm, float lambda_smooth, float gamma,
                      float lambda_depth, float* sums, float* dL_dnorm, float* dL_ddepth, void* stream);

/* Pseudo-normal and validity mask from the operator's depth output -- norm_from_depth of losses/norm_reg_loss.py:16-63, the
 * producer of (gt_norm, mask) for the normal-regularisation term norm_reg_loss (:73-78, models/texture_gaussian3d.py:360-363;
 * the reference detaches depth there, so there is no backward).  depth f32[1,H,W]; viewmatrix = DEVICE f32[16], the camera's
 * world_view_transform as the reference stores it (row-vector form, the same tensor TexGSFrame.viewmatrix points to): the
 * kernel inverts it itself (camera -> world), so the loss needs neither a host copy nor a sync; out_norm f32[3,H,W];
 * out_mask f32[1,H,W]. */
int texgs_norm_from_depth(const float* depth, const float* viewmatrix, float tanfovx, float tanfovy, int32_t H, int32_t W,
                          float threshold, float* out_norm, float* out_mask, void* stream);

/* Fused UV-Taylor producer: the operator inputs `uvs` and `gradient_uvs` straight from the Gaussian centres, replacing
 * UVNet.forward (models/modules/uv_net.py:19-36) + torch.autograd.functional.jacobian (models/texture_gaussian3d.py:216-227).
 * Weights are nn.Linear tensors (row-major [out, in]) of the shipped architecture (hidden width 128): pre_mlp = W1, W2;
 * mlp = W3, W4, W5; biases may be NULL (tiny-cuda-nn networks have none); emb f32[128] is geo_emb.weight[0]; xyz_offset /
 * xyz_scale f32[3] or NULL.  uvs f32[N,3] (unit), grad_uvs f32[N,9] with [3*i+j] = d uv_i / d x_j.  fp32 MFMA. */
typedef struct TexGSUVNet {
    const float *W1, *b1;      /* [128,3], [128]   */
    const float *W2, *b2;      /* [128,128], [128] */
    const float *emb;          /* [128]            */
    const float *W3, *b3, *W4, *b4;
    const float *W5, *b5;      /* [3,128], [3]     */
    const float *xyz_offset, *xyz_scale;
    int32_t hidden;            /* must be 128      */
} TexGSUVNet;
size_t texgs_uv_taylor_temp_bytes(void);
int texgs_uv_taylor(const TexGSUVNet* net, const float* xyz, int32_t N, float* uvs, float* grad_uvs, void* temp, void* stream);
/* The same in two steps, for callers that evaluate one set of weights many times (every view of a retexture / viewer session,
 * every forward between two optimizer steps): texgs_uv_pack re-orders W2..W4 for the matrix cores into `packed`
 * (texgs_uv_taylor_temp_bytes() bytes) once, texgs_uv_taylor_packed evaluates with it. */
int texgs_uv_pack(const TexGSUVNet* net, void* packed, void* stream);
int texgs_uv_taylor_packed(const TexGSUVNet* net, const void* packed, const float* xyz, int32_t N, float* uvs, float* grad_uvs,
                           void* stream);
/* The same two steps at SPLIT-bf16 precision (v12, opt-in): every operand of the three 128x128 layers is split into two bf16
 * halves and a product taken as hi*hi + hi*lo + lo*hi on the bf16 matrix cores (f32 accumulation) -- ~2.5x faster than the f32-
 * input MFMA of texgs_uv_taylor_packed, uvs / Jacobian within ~2e-5 of it.  `packed` has the same size, a different layout: a
 * buffer packed by one variant must not be handed to the other. */
int texgs_uv_pack_bf16x3(const TexGSUVNet* net, void* packed, void* stream);
int texgs_uv_taylor_packed_bf16x3(const TexGSUVNet* net, const void* packed, const float* xyz, int32_t N, float* uvs, float* grad_uvs,
                                  void* stream);

/* The same two steps with the VALUE column on the f32-input MFMA and the three TANGENT columns at split-bf16 precision (v14,
 * opt-in): uvs and every ReLU mask are those of texgs_uv_taylor_packed (the value column decides everything discrete and runs
 * exactly as there); the Jacobian is within ~1e-5 relative of it; ~2x faster.  `packed` holds BOTH layouts back to back:
 * 2 * texgs_uv_taylor_temp_bytes() bytes. */
int texgs_uv_pack_mixed(const TexGSUVNet* net, void* packed, void* stream);
int texgs_uv_taylor_packed_mixed(const TexGSUVNet* net, const void* packed, const float* xyz, int32_t N, float* uvs, float* grad_uvs,
                                 void* stream);

/* Backward of the UV map (v13): gradients of uvs = UVNet(xyz) w.r.t. every weight, bias and the embedding for an upstream
 * gradient g_uvs f32[N,3] -- what loss.backward() does through models/modules/uv_net.py:19-36 in the reference (autograd over
 * five nn.Linear + F.normalize; reached from models/texture_gaussian3d.py:410 via `uvs`, :229-236).  ONE persistent kernel
 * (activations recomputed per tile of 64 points in LDS, the three 128x128 weight gradients in registers, fp32 MFMA) + a
 * deterministic reduction of its <= 256 partial sums.  Outputs are nn.Linear-shaped, OVERWRITTEN (not accumulated); a NULL
 * pointer skips that gradient; d emb = db2 (the embedding is added where b2 is).  d xyz is not produced here: it is J^T g with
 * the Jacobian texgs_uv_taylor already returned.  temp: texgs_uv_backward_temp_bytes(N) bytes. */
typedef struct TexGSUVNetGrad {
    float *dW1, *db1;          /* [128,3], [128]   */
    float *dW2, *db2;          /* [128,128], [128] */
    float *dW3, *db3, *dW4, *db4;
    float *dW5, *db5;          /* [3,128], [3]     */
} TexGSUVNetGrad;
size_t texgs_uv_backward_temp_bytes(int32_t N);
int texgs_uv_backward(const TexGSUVNet* net, const float* xyz, const float* g_uvs, int32_t N, const TexGSUVNetGrad* out, void* temp,
                      void* stream);
/* The same gradients with the six GEMMs of the backward chain (W^T d, d h^T) as split-bf16 products (three bf16 MFMAs per f32
 * product, f32 accumulation: ~1e-5 relative); the forward recomputation stays on the f32-input MFMA, so the ReLU masks are those
 * of the forward launch bit for bit (v15; same arguments and temp size). */
int texgs_uv_backward_mixed(const TexGSUVNet* net, const float* xyz, const float* g_uvs, int32_t N, const TexGSUVNetGrad* out, void* temp,
                            void* stream);

/* Hardware self-test of the wave64 cross-lane primitives the backward's reductions use (csrc/wave_ops.h: DPP lane^4 /
 * lane^8 exchanges, permlane16/32 swaps, both transposing butterflies).  seed: f32[128] device; out: f32[576] device,
 * nine blocks of 64 differences against the __shfl_xor formulation -- all exactly 0 on gfx950. */
int texgs_selftest_waveops(const float* seed128, float* out576, void* stream);

/* Frustum test only (upstream API `markVisible`; unused by the reference). visible: u8[N]. */
int texgs_mark_visible(const TexGSFrame* frame, const float* means3D, uint8_t* visible, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TEXGS_H */
