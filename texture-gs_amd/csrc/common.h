// Shared declarations for the gfx950 kernels of libtexgs.so.  CDNA4 only: wave64, 160 KiB LDS/CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "texgs.h"

// ---- operator constants (restated in oracle/texgs_torch.py) ----
#define TG_NEAR_Z        0.2f
#define TG_LOWPASS       0.3f
#define TG_FRUSTUM_CLAMP 1.3f
#define TG_ALPHA_MAX     0.99f
#define TG_ALPHA_MIN     (1.0f / 255.0f)
#define TG_T_EPS         1e-4f
#define TG_PLANE_EPS     5e-2f
#define TG_DEN_MIN       0.2f
#define TG_MA_MIN        1e-20f
#define TG_SH_C0         0.28209479177387814f

// slots of the per-Gaussian fields whose gradients K8 assembles (float index; K1 stores them split into a test and a
// shading record, texgs.h TexGSGeom)
enum {
    R_XY = 0, R_CONIC = 2, R_OP = 5, R_G2 = 6, R_GM = 8, R_PHI = 14, R_VD = 17, R_DEPTH = 20, R_N = 21
};
// accumulator-row slots (TexGSGrads.acc, 32 floats = one 128-byte line per Gaussian): raw moments about the splat centre,
// summed by K7 over the Gaussian's contributing pixels.  dx = xy - pixel, dp = pixel - xy.
//   M_P   : sum P {1, dx, dy, dx^2, dx dy, dy^2}          P = dL/dpower
//   M_DEN : sum dden {1, dpx, dpy}                         dden = dL/d(1 + g.dp)
//   M_DN  : for c = 0..2: sum dn_c {1, dpx, dpy}           dn = dL/duv / den
//   M_PHI : sum dL/duv (3)   M_VD: sum dL/dcolour (3)   M_DEPTH: sum w dL/ddepth   M_N: sum w dL/dnormal (3)
enum {
    M_P = 0, M_DEN = 6, M_DN = 9, M_PHI = 18, M_VD = 21, M_DEPTH = 24, M_N = 25
};

#define TG_BLOCK 256

struct CamConst {       // small per-frame constants passed by value in kernel args (SGPRs)
    int W, H, tiles_x, tiles_y;
    float fx, fy, tanfovx, tanfovy;
    float scale_modifier;
    int sh_degree, sh_coeffs, R, N;
};

static inline CamConst make_cam(const TexGSFrame* f) {
    CamConst c;
    c.W = f->image_width; c.H = f->image_height;
    c.tiles_x = (c.W + TEXGS_TILE - 1) / TEXGS_TILE;
    c.tiles_y = (c.H + TEXGS_TILE - 1) / TEXGS_TILE;
    c.tanfovx = f->tanfovx; c.tanfovy = f->tanfovy;
    c.fx = (float)c.W / (2.0f * f->tanfovx);
    c.fy = (float)c.H / (2.0f * f->tanfovy);
    c.scale_modifier = f->scale_modifier;
    c.sh_degree = f->sh_degree; c.sh_coeffs = f->sh_coeffs; c.R = f->tex_res; c.N = f->num_gaussians;
    return c;
}

// launchers implemented in the .hip files (host side, called from abi.hip)
void launch_preprocess_fwd(const CamConst& c, const TexGSFrame* f, const TexGSInputs* in, TexGSGeom* g, hipStream_t s);
void launch_preprocess_bwd(const CamConst& c, const TexGSFrame* f, const TexGSInputs* in, const TexGSGeom* g,
                           TexGSGrads* gr, hipStream_t s);
void launch_mark_visible(const TexGSFrame* f, const float* means3D, uint8_t* visible, hipStream_t s);
int  launch_depth_sort_scan(const TexGSGeom* g, int N, hipStream_t s);
uint32_t* bin_block_sums_ptr(const TexGSGeom* g, int N);
uint32_t* bin_header_ptr(const TexGSGeom* g, int N, int* words);
size_t scan_temp_bytes(int N);
size_t sort_temp_bytes(uint32_t D, uint32_t T);
void launch_duplicate(const CamConst& c, const TexGSGeom* g, TexGSBinning* b, hipStream_t s);
int  launch_sort(const CamConst& c, const TexGSGeom* g, TexGSBinning* b, hipStream_t s);
void launch_ranges(const CamConst& c, TexGSBinning* b, uint32_t* zero_words, int num_zero_words, uint32_t* zero_words2,
                   int num_zero_words2, hipStream_t s);
void launch_render_fwd(const CamConst& c, const TexGSFrame* f, const TexGSInputs* in, const TexGSGeom* g,
                       const TexGSBinning* b, TexGSImage* img, hipStream_t s);
size_t tex_bin_count(int R);
bool tex_bins_enabled(const CamConst& c, const TexGSInputs* in, const TexGSImage* img, const TexGSGrads* gr);
void launch_texgrad_reduce(const CamConst& c, const TexGSImage* img, TexGSGrads* gr, hipStream_t s);
void launch_render_bwd(const CamConst& c, const TexGSFrame* f, const TexGSInputs* in, const TexGSGeom* g,
                       const TexGSBinning* b, const TexGSImage* img, TexGSGrads* gr, hipStream_t s);
int launch_rgb_alpha_loss(const float* image, const float* gt_image, const float* alpha, const float* gt_alpha, int H,
                          int W, float lambda_dssim, float lambda_alpha, float* scratch, float* sums, float* d_image,
                          float* d_alpha, hipStream_t s);
void launch_selftest_waveops(const float* seed128, float* out576, hipStream_t s);
int launch_geom_losses(const float* norm, const float* gt_norm, const float* gt_image, const float* mask, const float* depth,
                       const float* gt_depth, int H, int W, float lambda_norm, float lambda_smooth, float gamma,
                       float lambda_depth, float* sums, float* d_norm, float* d_depth, hipStream_t s);
int launch_norm_from_depth(const float* depth, const float* viewmatrix, float tanfovx, float tanfovy, int H, int W, float threshold,
                           float* out_norm, float* out_mask, hipStream_t s);
size_t uv_taylor_temp_bytes();
int launch_uv_pack(const TexGSUVNet* net, void* packed, hipStream_t s);
int launch_uv_taylor_packed(const TexGSUVNet* net, const void* packed, const float* xyz, int N, float* uvs, float* grad_uvs, hipStream_t s);
int launch_uv_taylor(const TexGSUVNet* net, const float* xyz, int N, float* uvs, float* grad_uvs, void* temp, hipStream_t s);
int launch_uv_pack_bf16x3(const TexGSUVNet* net, void* packed, hipStream_t s);
size_t uv_backward_temp_bytes(int N);
int launch_uv_backward(const TexGSUVNet* net, const float* xyz, const float* g, int N, const TexGSUVNetGrad* out, void* temp, int mixed,
                       hipStream_t s);
int launch_uv_taylor_packed_bf16x3(const TexGSUVNet* net, const void* packed, const float* xyz, int N, float* uvs, float* grad_uvs, hipStream_t s);
