// Fused loss front-end for the operator's outputs: the always-on terms of
// TextureGaussian3D.compute_loss (models/texture_gaussian3d.py:333-345):
//     loss = (1 - l) * mean|I - I_gt|  +  l * (1 - mean SSIM(I, I_gt))  +  la * mean|A - A_gt|
// SSIM as losses/ssim_loss.py:16-54: 11x11 Gaussian window (sigma 1.5, separable), zero padding 5, per channel,
// C1 = 0.01^2, C2 = 0.03^2.  Two HBM-bound kernels, 16x16-pixel tiles with a 5-pixel halo staged in LDS:
//   k_ssim_fwd   5 windowed moments -> SSIM value and its partials w.r.t. (mu1, E[x^2], E[xy]); block-reduced sums
//   k_ssim_bwd   the same (symmetric) window applied to the 3 partial maps -> dL/dI; + the L1 sign terms
// The reference does this with 5 depthwise conv2d calls + ~25 elementwise kernels and their autograd.
#include "common.h"

namespace {

#define LT 16                   // tile
#define LH 5                    // halo
#define LW (LT + 2 * LH)        // 26

struct Win { float w[11]; };

__device__ __forceinline__ float block_sum(float v, float* s_red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    if (lane == 0) s_red[wave] = v;
    __syncthreads();
    const float t = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    __syncthreads();
    return t;
}

// (Same-address global atomics execute memory-side, ~13 ns each and serialised: one pair per 16x16 tile -- 15 000 of them at
// 800x800 -- made this kernel 201 us, 3 % of its HBM bound.  Blocks therefore loop over tiles and add their sums ONCE.)
__global__ void __launch_bounds__(256)
k_ssim_fwd(int H, int W, int tiles_x, int tiles_y, Win win, const float* __restrict__ img, const float* __restrict__ gt,
           float* __restrict__ part /* [3 maps][3 ch][H][W] */, float* __restrict__ sums) {
    __shared__ float s_x[LW][LW + 1], s_y[LW][LW + 1];
    __shared__ float s_h[5][LW][LT + 1];
    __shared__ float s_red[4];
    float acc_ssim = 0.f, acc_l1 = 0.f;
    const int nwork = tiles_x * tiles_y * 3;
    for (int work = blockIdx.x; work < nwork; work += gridDim.x) {
        const int c = work / (tiles_x * tiles_y), tt = work - c * (tiles_x * tiles_y);
        const int tx0 = (tt % tiles_x) * LT, ty0 = (tt / tiles_x) * LT;
        const float* __restrict__ X = img + (size_t)c * H * W;
        const float* __restrict__ Y = gt + (size_t)c * H * W;
        __syncthreads();                                        // the previous tile's LDS reads are done
        for (int k = threadIdx.x; k < LW * LW; k += 256) {
            const int r = k / LW, q = k % LW, yy = ty0 + r - LH, xx = tx0 + q - LH;
            const bool in = yy >= 0 && yy < H && xx >= 0 && xx < W;
            s_x[r][q] = in ? X[(size_t)yy * W + xx] : 0.f;
            s_y[r][q] = in ? Y[(size_t)yy * W + xx] : 0.f;
        }
        __syncthreads();
        for (int k = threadIdx.x; k < LW * LT; k += 256) {        // horizontal pass
            const int r = k / LT, q = k % LT;
            float a = 0.f, b = 0.f, aa = 0.f, bb = 0.f, ab = 0.f;
#pragma unroll
            for (int t = 0; t < 11; ++t) {
                const float x = s_x[r][q + t], y = s_y[r][q + t], w = win.w[t];
                a += w * x; b += w * y; aa += w * x * x; bb += w * y * y; ab += w * x * y;
            }
            s_h[0][r][q] = a; s_h[1][r][q] = b; s_h[2][r][q] = aa; s_h[3][r][q] = bb; s_h[4][r][q] = ab;
        }
        __syncthreads();
        const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4, px = tx0 + lx, py = ty0 + ly;
        float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
        for (int t = 0; t < 11; ++t) {
            const float w = win.w[t];
            mu1 += w * s_h[0][ly + t][lx]; mu2 += w * s_h[1][ly + t][lx]; e11 += w * s_h[2][ly + t][lx];
            e22 += w * s_h[3][ly + t][lx]; e12 += w * s_h[4][ly + t][lx];
        }
        if (px < W && py < H) {
            const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
            const float s1 = e11 - mu1 * mu1, s2 = e22 - mu2 * mu2, s12 = e12 - mu1 * mu2;
            const float A1 = 2.f * mu1 * mu2 + C1, A2 = 2.f * s12 + C2, B1 = mu1 * mu1 + mu2 * mu2 + C1, B2 = s1 + s2 + C2;
            const float inv = 1.f / (B1 * B2);
            const float ssim = A1 * A2 * inv;
            const size_t o = ((size_t)c * H + py) * W + px, plane = (size_t)3 * H * W;
            part[o]             = 2.f * mu2 * (A2 - A1) * inv - ssim * 2.f * mu1 * (B2 - B1) * inv;   // dS/dmu1
            part[plane + o]     = -ssim / B2;                                                        // dS/dE[x^2]
            part[2 * plane + o] = 2.f * A1 * inv;                                                    // dS/dE[xy]
            acc_ssim += ssim;
            acc_l1 += fabsf(s_x[ly + LH][lx + LH] - s_y[ly + LH][lx + LH]);
        }
    }
    __syncthreads();
    const float ts = block_sum(acc_ssim, s_red), tl = block_sum(acc_l1, s_red);
    if (threadIdx.x == 0) { atomicAdd(sums + 0, tl); atomicAdd(sums + 1, ts); }
}

__global__ void __launch_bounds__(256)
k_ssim_bwd(int H, int W, Win win, const float* __restrict__ img, const float* __restrict__ gt,
           const float* __restrict__ part, float g_ssim, float g_l1, float* __restrict__ d_img) {
    __shared__ float s_p[3][LW][LW + 1];
    __shared__ float s_h[3][LW][LT + 1];
    const int c = blockIdx.z, tx0 = blockIdx.x * LT, ty0 = blockIdx.y * LT;
    const size_t plane = (size_t)3 * H * W;
    for (int k = threadIdx.x; k < LW * LW; k += 256) {
        const int r = k / LW, q = k % LW, yy = ty0 + r - LH, xx = tx0 + q - LH;
        const bool in = yy >= 0 && yy < H && xx >= 0 && xx < W;
        const size_t o = ((size_t)c * H + yy) * W + xx;
#pragma unroll
        for (int m = 0; m < 3; ++m) s_p[m][r][q] = in ? part[m * plane + o] : 0.f;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < LW * LT; k += 256) {
        const int r = k / LT, q = k % LT;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int t = 0; t < 11; ++t) { const float w = win.w[t]; a0 += w * s_p[0][r][q + t]; a1 += w * s_p[1][r][q + t]; a2 += w * s_p[2][r][q + t]; }
        s_h[0][r][q] = a0; s_h[1][r][q] = a1; s_h[2][r][q] = a2;
    }
    __syncthreads();
    const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4, px = tx0 + lx, py = ty0 + ly;
    if (px >= W || py >= H) return;
    float c0 = 0.f, c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int t = 0; t < 11; ++t) { const float w = win.w[t]; c0 += w * s_h[0][ly + t][lx]; c1 += w * s_h[1][ly + t][lx]; c2 += w * s_h[2][ly + t][lx]; }
    const size_t o = ((size_t)c * H + py) * W + px;
    const float x = img[o], y = gt[o];
    const float sgn = (x > y) ? 1.f : ((x < y) ? -1.f : 0.f);
    d_img[o] = g_ssim * (c0 + 2.f * x * c1 + y * c2) + g_l1 * sgn;
}

__global__ void __launch_bounds__(256)
k_alpha_l1(int P, const float* __restrict__ a, const float* __restrict__ gt, float g, float* __restrict__ d_a,
           float* __restrict__ sums) {
    __shared__ float s_red[4];
    float l = 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < P; i += gridDim.x * 256) {
        const float d = a[i] - gt[i];
        l += fabsf(d);
        d_a[i] = g * ((d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f));
    }
    const float t = block_sum(l, s_red);
    if (threadIdx.x == 0) atomicAdd(sums + 2, t);          // one atomic per block (grid-stride: a few hundred blocks)
}

// ---- geometric regularisers of TextureGaussian3D.compute_loss (models/texture_gaussian3d.py:347-368) on the operator's
// normal / depth outputs -- the terms that feed dL/dnorm and dL/ddepth into K7:
//   norm_loss(norm, gt_norm, mask)   losses/norm_reg_loss.py:66-71   sum((1 - <n, g>) m) / (sum m + 1e-6)
//   smooth_loss(gt_image, norm, mask) losses/smooth_loss.py:4-27     bilateral first-order smoothness over the 4 neighbour
//        directions k (right, down, down-right, anti-diagonal): w_k = exp(-sum_c |d rgb| / gamma) m_a m_b,
//        L = 1/4 sum_k  sum(w_k sum_c |d n_c|) / (sum w_k + 1e-6)
//   l1_loss(depth, gt_depth)         losses/pixelwise_loss.py        mean |d - d_gt|
// Two HBM-bound kernels: the normalisers sum m and sum w_k depend only on the ground truth, so k_geom_sums reduces them
// (and the loss numerators) and k_geom_grad then writes dL/dnorm, dL/ddepth in one pass.
// sums: [0] sum m  [1] sum (1-<n,g>) m  [2..5] sum w_k  [6..9] sum w_k |dn|  [10] sum |d - d_gt|
struct GeomArgs {
    int H, W;
    const float *norm, *gt_norm, *gt_image, *mask, *depth, *gt_depth;
    float inv_gamma;
};

__device__ __forceinline__ float pair_weight(const GeomArgs& a, int ya, int xa, int yb, int xb) {
    const size_t P = (size_t)a.H * a.W, ia = (size_t)ya * a.W + xa, ib = (size_t)yb * a.W + xb;
    const float d = fabsf(a.gt_image[ia] - a.gt_image[ib]) + fabsf(a.gt_image[P + ia] - a.gt_image[P + ib])
                  + fabsf(a.gt_image[2 * P + ia] - a.gt_image[2 * P + ib]);
    float w = __expf(-d * a.inv_gamma);
    if (a.mask) w *= a.mask[ia] * a.mask[ib];
    return w;
}
// pair k anchored at (y, x): (ya, xa) - (yb, xb); false when it leaves the image
__device__ __forceinline__ bool pair_ends(int k, int y, int x, int H, int W, int& ya, int& xa, int& yb, int& xb) {
    ya = y; xa = x;
    if (k == 0) { yb = y; xb = x + 1; }
    else if (k == 1) { yb = y + 1; xb = x; }
    else if (k == 2) { yb = y + 1; xb = x + 1; }
    else { ya = y + 1; xa = x; yb = y; xb = x + 1; }
    return ya >= 0 && yb >= 0 && xa >= 0 && xb >= 0 && ya < H && yb < H && xa < W && xb < W;
}

// (Eleven same-address atomics per 256 pixels -- 27 500 at 800x800, serialised memory-side -- made this kernel 359 us; blocks
// now stride over the image, keep the eleven sums in registers and add them once.)
__global__ void __launch_bounds__(256)
k_geom_sums(GeomArgs a, int do_norm, int do_smooth, int do_depth, float* __restrict__ sums) {
    __shared__ float s_red[4];
    const int P = a.H * a.W;
    float v[11];
#pragma unroll
    for (int k = 0; k < 11; ++k) v[k] = 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < P; i += gridDim.x * 256) {
        const int y = i / a.W, x = i % a.W;
        const float n0 = a.norm ? a.norm[i] : 0.f, n1 = a.norm ? a.norm[P + i] : 0.f, n2 = a.norm ? a.norm[2 * P + i] : 0.f;
        if (do_norm) {
            const float m = a.mask ? a.mask[i] : 1.f;
            v[0] += m;
            v[1] += (1.f - (n0 * a.gt_norm[i] + n1 * a.gt_norm[P + i] + n2 * a.gt_norm[2 * P + i])) * m;
        }
        if (do_smooth) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int ya, xa, yb, xb;
                if (!pair_ends(k, y, x, a.H, a.W, ya, xa, yb, xb)) continue;
                const float w = pair_weight(a, ya, xa, yb, xb);
                const size_t ia = (size_t)ya * a.W + xa, ib = (size_t)yb * a.W + xb;
                v[2 + k] += w;
                v[6 + k] += fabsf(w * (a.norm[ia] - a.norm[ib])) + fabsf(w * (a.norm[P + ia] - a.norm[P + ib]))
                          + fabsf(w * (a.norm[2 * P + ia] - a.norm[2 * P + ib]));
            }
        }
        if (do_depth) v[10] += fabsf(a.depth[i] - a.gt_depth[i]);
    }
#pragma unroll
    for (int k = 0; k < 11; ++k) {
        const float t = block_sum(v[k], s_red);
        if (threadIdx.x == 0 && t != 0.f) atomicAdd(sums + k, t);
    }
}

__global__ void __launch_bounds__(256)
k_geom_grad(GeomArgs a, float l_norm, float l_smooth, float l_depth, const float* __restrict__ sums,
            float* __restrict__ d_norm, float* __restrict__ d_depth) {
    const int i = blockIdx.x * 256 + threadIdx.x, P = a.H * a.W;
    if (i >= P) return;
    const int y = i / a.W, x = i % a.W;
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    if (l_norm != 0.f) {
        const float c = -l_norm * (a.mask ? a.mask[i] : 1.f) / (sums[0] + 1e-6f);
        g0 = c * a.gt_norm[i]; g1 = c * a.gt_norm[P + i]; g2 = c * a.gt_norm[2 * P + i];
    }
    if (l_smooth != 0.f) {
        // this pixel is end `a` of the pairs anchored at ... and end `b` of the pairs anchored at ...
        const int ay[4] = {y, y, y, y - 1}, ax[4] = {x, x, x, x};             // anchors where (y, x) is end a
        const int by[4] = {y, y - 1, y - 1, y}, bx[4] = {x - 1, x, x - 1, x - 1};   // anchors where (y, x) is end b
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float sc = 0.25f * l_smooth / (sums[2 + k] + 1e-6f);
#pragma unroll
            for (int side = 0; side < 2; ++side) {
                const int yy = side ? by[k] : ay[k], xx = side ? bx[k] : ax[k];
                int ya, xa, yb, xb;
                if (yy < 0 || xx < 0 || !pair_ends(k, yy, xx, a.H, a.W, ya, xa, yb, xb)) continue;
                const float w = pair_weight(a, ya, xa, yb, xb) * sc * (side ? -1.f : 1.f);
                const size_t ia = (size_t)ya * a.W + xa, ib = (size_t)yb * a.W + xb;
                const float e0 = a.norm[ia] - a.norm[ib], e1 = a.norm[P + ia] - a.norm[P + ib], e2 = a.norm[2 * P + ia] - a.norm[2 * P + ib];
                g0 += w * ((e0 > 0.f) ? 1.f : ((e0 < 0.f) ? -1.f : 0.f));
                g1 += w * ((e1 > 0.f) ? 1.f : ((e1 < 0.f) ? -1.f : 0.f));
                g2 += w * ((e2 > 0.f) ? 1.f : ((e2 < 0.f) ? -1.f : 0.f));
            }
        }
    }
    if (d_norm) { d_norm[i] = g0; d_norm[P + i] = g1; d_norm[2 * P + i] = g2; }
    if (d_depth && l_depth != 0.f) {
        const float e = a.depth[i] - a.gt_depth[i];
        d_depth[i] = (l_depth / (float)P) * ((e > 0.f) ? 1.f : ((e < 0.f) ? -1.f : 0.f));
    }
}

// ------------------------------------------------------------------------------------------------ pseudo-normal from depth
// losses/norm_reg_loss.py:16-63 (norm_from_depth): back-project every pixel's depth to a world-space point, one-sided
// differences to the four neighbours (replicate border), normal = normalise(cross(grad_y, grad_x)) with grad_x / grad_y the
// means of the two one-sided differences, mask = all four differences shorter than `threshold`.  The camera-to-world matrix
// (inverse of the column-convention view matrix world_view_transform^T, rows 0..2) is formed IN the kernel from the device-resident
// view matrix (adjugate of its 3x3 part: wave-uniform scalar work) -- no host copy, no host-side inverse, no sync in the loss.
// One thread per pixel: 5 depth reads (L1 / L2 hits), 16 bytes written -- elementwise HBM work, no LDS.
struct DepthNormArgs { int H, W; float tx, ty, thr; float m[12]; };

// vm = world_view_transform as the reference stores it (row-vector convention, [4,4] row-major): p_view = [p, 1] @ vm, i.e.
// p_view = A p + t with A[r][c] = vm[c*4 + r], t[r] = vm[12 + r].  Returns rows 0..2 of [A^-1 | -A^-1 t].
__device__ __forceinline__ void cam_to_world(const float* __restrict__ vm, float (&m)[12]) {
    const float a00 = vm[0], a01 = vm[4], a02 = vm[8], a10 = vm[1], a11 = vm[5], a12 = vm[9], a20 = vm[2], a21 = vm[6], a22 = vm[10];
    const float t0 = vm[12], t1 = vm[13], t2 = vm[14];
    const float c00 = a11 * a22 - a12 * a21, c01 = a02 * a21 - a01 * a22, c02 = a01 * a12 - a02 * a11;
    const float c10 = a12 * a20 - a10 * a22, c11 = a00 * a22 - a02 * a20, c12 = a02 * a10 - a00 * a12;
    const float c20 = a10 * a21 - a11 * a20, c21 = a01 * a20 - a00 * a21, c22 = a00 * a11 - a01 * a10;
    const float idet = 1.0f / (a00 * c00 + a01 * c10 + a02 * c20);
    const float i[9] = {c00 * idet, c01 * idet, c02 * idet, c10 * idet, c11 * idet, c12 * idet, c20 * idet, c21 * idet, c22 * idet};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        m[4 * r] = i[3 * r]; m[4 * r + 1] = i[3 * r + 1]; m[4 * r + 2] = i[3 * r + 2];
        m[4 * r + 3] = -(i[3 * r] * t0 + i[3 * r + 1] * t1 + i[3 * r + 2] * t2);
    }
}

__device__ __forceinline__ void backproject(const DepthNormArgs& a, const float* __restrict__ depth, int x, int y, float (&p)[3]) {
    const float d = depth[y * a.W + x];
    const float nx = (2.0f * (float)x + 1.0f) / (float)a.W - 1.0f, ny = (2.0f * (float)y + 1.0f) / (float)a.H - 1.0f;
    const float cx = nx * a.tx * d, cy = ny * a.ty * d;
#pragma unroll
    for (int k = 0; k < 3; ++k) p[k] = a.m[4 * k] * cx + a.m[4 * k + 1] * cy + a.m[4 * k + 2] * d + a.m[4 * k + 3];
}

__global__ void __launch_bounds__(256)
k_norm_from_depth(DepthNormArgs a, const float* __restrict__ viewmatrix, const float* __restrict__ depth,
                  float* __restrict__ out_norm, float* __restrict__ out_mask) {
    const int i = blockIdx.x * 256 + threadIdx.x, P = a.H * a.W;
    if (i >= P) return;
    cam_to_world(viewmatrix, a.m);
    const int y = i / a.W, x = i - y * a.W;
    float c[3], l[3], r[3], u[3], dn[3];
    backproject(a, depth, x, y, c);
    backproject(a, depth, max(x - 1, 0), y, l);
    backproject(a, depth, min(x + 1, a.W - 1), y, r);
    backproject(a, depth, x, max(y - 1, 0), u);
    backproject(a, depth, x, min(y + 1, a.H - 1), dn);
    float gl[3], gr[3], gu[3], gd[3], gx[3], gy[3];
    float nl = 0.f, nr = 0.f, nu = 0.f, nd = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        gl[k] = c[k] - l[k]; gr[k] = r[k] - c[k]; gu[k] = c[k] - u[k]; gd[k] = dn[k] - c[k];
        nl += gl[k] * gl[k]; nr += gr[k] * gr[k]; nu += gu[k] * gu[k]; nd += gd[k] * gd[k];
        gx[k] = (gr[k] + gl[k]) * 0.5f; gy[k] = (gd[k] + gu[k]) * 0.5f;
    }
    const float t2 = a.thr;
    const bool ok = sqrtf(nl) < t2 && sqrtf(nr) < t2 && sqrtf(nu) < t2 && sqrtf(nd) < t2;
    // cross(grad_y, grad_x)
    const float n0 = gy[1] * gx[2] - gy[2] * gx[1], n1 = gy[2] * gx[0] - gy[0] * gx[2], n2 = gy[0] * gx[1] - gy[1] * gx[0];
    const float len = fmaxf(sqrtf(n0 * n0 + n1 * n1 + n2 * n2), 1e-6f);      // F.normalize(eps = 1e-6)
    out_norm[i] = n0 / len; out_norm[P + i] = n1 / len; out_norm[2 * P + i] = n2 / len;
    out_mask[i] = ok ? 1.0f : 0.0f;
}

}  // namespace

int launch_geom_losses(const float* norm, const float* gt_norm, const float* gt_image, const float* mask, const float* depth,
                       const float* gt_depth, int H, int W, float lambda_norm, float lambda_smooth, float gamma,
                       float lambda_depth, float* sums, float* d_norm, float* d_depth, hipStream_t s) {
    GeomArgs a;
    a.H = H; a.W = W; a.norm = norm; a.gt_norm = gt_norm; a.gt_image = gt_image; a.mask = mask; a.depth = depth;
    a.gt_depth = gt_depth; a.inv_gamma = 1.f / gamma;
    (void)hipMemsetAsync(sums, 0, 12 * sizeof(float), s);
    const int P = H * W, blocks = (P + 255) / 256;
    // 512 persistent blocks (2 per CU) stride over the image: 11 atomics per BLOCK, not per 256 pixels
    hipLaunchKernelGGL(k_geom_sums, dim3(blocks < 512 ? blocks : 512), dim3(256), 0, s, a, lambda_norm != 0.f, lambda_smooth != 0.f,
                       lambda_depth != 0.f, sums);
    hipLaunchKernelGGL(k_geom_grad, dim3(blocks), dim3(256), 0, s, a, lambda_norm, lambda_smooth, lambda_depth,
                       (const float*)sums, d_norm, d_depth);
    return 0;
}

// returns 0; sums[0..2] = sum|I-Igt|, sum SSIM, sum|A-Agt| (device); dL/dI and dL/dA are for d(loss) = 1
int launch_rgb_alpha_loss(const float* image, const float* gt_image, const float* alpha, const float* gt_alpha, int H,
                          int W, float lambda_dssim, float lambda_alpha, float* scratch, float* sums, float* d_image,
                          float* d_alpha, hipStream_t s) {
    Win win;
    float tot = 0.f;
    for (int k = 0; k < 11; ++k) { win.w[k] = expf(-(float)((k - 5) * (k - 5)) / (2.f * 1.5f * 1.5f)); tot += win.w[k]; }
    for (int k = 0; k < 11; ++k) win.w[k] /= tot;
    (void)hipMemsetAsync(sums, 0, 4 * sizeof(float), s);
    const dim3 grid((W + LT - 1) / LT, (H + LT - 1) / LT, 3);
    const float n = (float)(3 * (size_t)H * W);
    const int nwork = (int)(grid.x * grid.y * 3);
    hipLaunchKernelGGL(k_ssim_fwd, dim3(nwork < 1024 ? nwork : 1024), dim3(256), 0, s, H, W, (int)grid.x, (int)grid.y, win, image,
                       gt_image, scratch, sums);
    hipLaunchKernelGGL(k_ssim_bwd, grid, dim3(256), 0, s, H, W, win, image, gt_image, (const float*)scratch,
                       -lambda_dssim / n, (1.f - lambda_dssim) / n, d_image);
    if (alpha && gt_alpha && d_alpha) {
        const int P = H * W;
        const int ab = (P + 255) / 256;
        hipLaunchKernelGGL(k_alpha_l1, dim3(ab < 512 ? ab : 512), dim3(256), 0, s, P, alpha, gt_alpha, lambda_alpha / (float)P,
                           d_alpha, sums);
    }
    return 0;
}

int launch_norm_from_depth(const float* depth, const float* viewmatrix, float tanfovx, float tanfovy, int H, int W, float threshold,
                           float* out_norm, float* out_mask, hipStream_t s) {
    DepthNormArgs a;
    a.H = H; a.W = W; a.tx = tanfovx; a.ty = tanfovy; a.thr = threshold;
    for (int k = 0; k < 12; ++k) a.m[k] = 0.f;
    const int P = H * W;
    hipLaunchKernelGGL(k_norm_from_depth, dim3((P + 255) / 256), dim3(256), 0, s, a, viewmatrix, depth, out_norm, out_mask);
    return 0;
}
