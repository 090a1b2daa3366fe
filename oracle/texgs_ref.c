/*
 * TEST INFRASTRUCTURE ONLY -- plain-C (fp32) restatement of the textured Gaussian rasterizer operator.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; never the product path.
 *
 * PARITY UNPINNED: the reference's arithmetic for this path lives in the un-vendored, un-pinned pip dependency
 * `diff_gauss_uv_tex` (reference requirements.txt:15); /root/reference holds no source, test or golden vector for
 * it.  This file restates the published algorithm (3DGS tile rasterizer lineage, reference README.md:170;
 * Texture-GS paper arXiv 2403.10050) on the reference's own call-site contract and in-tree conventions:
 *   operator arguments           render/uv_tex_render.py:25-38,56-66
 *   row-vector matrices          utils/cameras.py:62-65, utils/graphics.py:22-29,51-71
 *   pixel<->ndc, depth = view z  models/texture_gaussian3d.py:299-309
 *   quaternion, cov = RS(RS)^T   utils/general.py:87-119, models/gaussian3d.py:17-21
 *   SH constants / signs         utils/sh.py:26-112; colour clamp render/render.py:68
 *   texel = SH-DC                models/texture_gaussian3d.py:16-21
 *   cubemap faces                models/modules/NVDIFFREC/util.py:94-101
 * It is validated (tests/test_c_oracle.py, CPU) against oracle/texgs_torch.py: forward to fp32 rounding, and its
 * hand-written backward against torch autograd of the float64 restatement.
 *
 * Role: (1) bit-exact contract for the integer stages -- built with -ffp-contract=off and written in the same fp32
 * operation order as the HIP preprocess kernel, so radii, tile rects, tiles_touched, offsets, 64-bit keys, the
 * sorted point list and the tile ranges must be IDENTICAL to the GPU's; (2) full-size parity (BASELINE configs at
 * 800x800 / 300k) in seconds; (3) bench.py's CPU baseline ("port", OpenMP over Gaussians / tiles).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define TILE 16
#define NEAR_Z 0.2f
#define LOWPASS 0.3f
#define FRUSTUM_CLAMP 1.3f
#define ALPHA_MAX 0.99f
#define ALPHA_MIN (1.0f / 255.0f)
#define T_EPS 1e-4f
#define PLANE_EPS 5e-2f
#define DEN_MIN 0.2f
#define MA_MIN 1e-20f
#define SH_C0 0.28209479177387814f
#define REC 24
/* -DTEXGS_REF_VARIANT (tests only): the blend loops with a differently-rounded exp and reciprocal -- a second "fp32
 * implementation" of the same contract, used on the CPU to check that texgs_ref_ambiguity explains every difference between
 * two such implementations (tests/test_c_oracle.py) before that is asked of the HIP kernels. */
#ifdef TEXGS_REF_VARIANT
#define EXPF(x) exp2f((x) * 1.44269504088896f)
#define RCPF(x) ((float)(1.0 / (double)(x)) * (1.0f + 5.9604645e-8f))
#else
#define EXPF(x) expf(x)
#define RCPF(x) (1.0f / (x))
#endif

typedef struct {
    int H, W, N, K, R, sh_degree;
    float tanfovx, tanfovy, scale_modifier;
    const float *bg, *V, *P, *cam;               /* host pointers: bg[3], viewmatrix[16], projmatrix[16], campos[3] */
    const float *means, *shs, *opac, *scales, *rots, *uvs, *juv, *tex;
    /* the untextured surface `diff_gauss` (reference render/render.py:75-84): tex == NULL (uvs / juv may then be NULL too) --
       colour = max(0, viewdep + coff + 0.5) with coff = C0 * SH_DC or colors_precomp - 0.5 (render/render.py:66-68);
       cov3d (render/render.py:52-53, layout of utils/general.py:73-82: xx xy xz yy yz zz) replaces scales / rots, the scale
       modifier is then not applied (lineage) */
    const float *coff, *cov3d;
} RefIn;

typedef struct {           /* per-Gaussian forward intermediates (same fields as the HIP kernel's) */
    int valid;
    float m[3], t[3], hx, hy, hw, pw, xy[2], q[4], s[3], R[9], M[9], S[6], txc, tyc;
    int clx, cly;
    float J00, J02, J11, J12, T0[3], T1[3], a, b, c, det, inv, conic[3];
    int radius, kmin;
    float sign, n[3], nv[3], sdot;
    int degen;
    float gx, gy, G[6], Km[9], dir[3], dlen;
} Geo;

#define WR(V, r, c) ((V)[(c) * 4 + (r)])

static int sh_active(int deg, int K) { int w = (deg + 1) * (deg + 1) - 1; return w < K ? w : K; }

static void sh_basis(int deg, float x, float y, float z, float *b) {
    const float C1 = 0.4886025119029199f;
    b[0] = -C1 * y; b[1] = C1 * z; b[2] = -C1 * x;
    if (deg > 1) {
        float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        b[3] = 1.0925484305920792f * xy; b[4] = -1.0925484305920792f * yz;
        b[5] = 0.31539156525252005f * (2.0f * zz - xx - yy); b[6] = -1.0925484305920792f * xz;
        b[7] = 0.5462742152960396f * (xx - yy);
        if (deg > 2) {
            b[8] = -0.5900435899266435f * y * (3.0f * xx - yy); b[9] = 2.890611442640554f * xy * z;
            b[10] = -0.4570457994644658f * y * (4.0f * zz - xx - yy);
            b[11] = 0.3731763325901154f * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
            b[12] = -0.4570457994644658f * x * (4.0f * zz - xx - yy);
            b[13] = 1.445305721320277f * z * (xx - yy); b[14] = -0.5900435899266435f * x * (xx - 3.0f * yy);
        }
    }
}

static void sh_basis_grad(int deg, float x, float y, float z, float *bx, float *by, float *bz) {
    const float C1 = 0.4886025119029199f;
    bx[0] = 0; by[0] = -C1; bz[0] = 0; bx[1] = 0; by[1] = 0; bz[1] = C1; bx[2] = -C1; by[2] = 0; bz[2] = 0;
    if (deg > 1) {
        const float c20 = 1.0925484305920792f, c22 = 0.31539156525252005f, c24 = 0.5462742152960396f;
        bx[3] = c20 * y; by[3] = c20 * x; bz[3] = 0;
        bx[4] = 0; by[4] = -c20 * z; bz[4] = -c20 * y;
        bx[5] = c22 * (-2.0f * x); by[5] = c22 * (-2.0f * y); bz[5] = c22 * 4.0f * z;
        bx[6] = -c20 * z; by[6] = 0; bz[6] = -c20 * x;
        bx[7] = c24 * 2.0f * x; by[7] = -c24 * 2.0f * y; bz[7] = 0;
        if (deg > 2) {
            float xx = x * x, yy = y * y, zz = z * z;
            const float c30 = -0.5900435899266435f, c31 = 2.890611442640554f, c32 = -0.4570457994644658f,
                        c33 = 0.3731763325901154f, c35 = 1.445305721320277f;
            bx[8] = c30 * 6.0f * x * y; by[8] = c30 * (3.0f * xx - 3.0f * yy); bz[8] = 0;
            bx[9] = c31 * y * z; by[9] = c31 * x * z; bz[9] = c31 * x * y;
            bx[10] = c32 * (-2.0f * x * y); by[10] = c32 * (4.0f * zz - xx - 3.0f * yy); bz[10] = c32 * 8.0f * y * z;
            bx[11] = c33 * (-6.0f * x * z); by[11] = c33 * (-6.0f * y * z); bz[11] = c33 * (6.0f * zz - 3.0f * xx - 3.0f * yy);
            bx[12] = c32 * (4.0f * zz - 3.0f * xx - yy); by[12] = c32 * (-2.0f * x * y); bz[12] = c32 * 8.0f * x * z;
            bx[13] = c35 * 2.0f * x * z; by[13] = -c35 * 2.0f * y * z; bz[13] = c35 * (xx - yy);
            bx[14] = c30 * (3.0f * xx - 3.0f * yy); by[14] = c30 * (-6.0f * x * y); bz[14] = 0;
        }
    }
}

/* Unit eigenvector of the smallest eigenvalue of the symmetric matrix (xx,xy,xz,yy,yz,zz): cyclic Jacobi, six sweeps in the fixed
 * order (0,1), (0,2), (1,2), on the matrix scaled by 1 / trace.  "The shortest axis" of a splat given by its covariance
 * (decision 8): a selection -- no gradient flows through it.  Same operations as the HIP kernel's. */
static void jacobi_rot(float *app, float *aqq, float *apq, float *arp, float *arq, float *v0p, float *v0q, float *v1p, float *v1q,
                       float *v2p, float *v2q) {
    if (fabsf(*apq) < 1e-30f) return;
    const float theta = (*aqq - *app) / (2.0f * *apq);
    const float t = ((theta >= 0.f) ? 1.0f : -1.0f) / (fabsf(theta) + sqrtf(theta * theta + 1.0f));
    const float c = 1.0f / sqrtf(t * t + 1.0f), sn = t * c;
    *app -= t * *apq; *aqq += t * *apq; *apq = 0.f;
    const float rp = *arp, rq = *arq;
    *arp = c * rp - sn * rq; *arq = sn * rp + c * rq;
    float x;
    x = *v0p; *v0p = c * x - sn * *v0q; *v0q = sn * x + c * *v0q;
    x = *v1p; *v1p = c * x - sn * *v1q; *v1q = sn * x + c * *v1q;
    x = *v2p; *v2p = c * x - sn * *v2q; *v2q = sn * x + c * *v2q;
}
static void smallest_eigvec(const float *S6, float *n) {
    const float tr = S6[0] + S6[3] + S6[5];
    const float sc = (tr > 0.f) ? 1.0f / tr : 1.0f;
    float a00 = S6[0] * sc, a01 = S6[1] * sc, a02 = S6[2] * sc, a11 = S6[3] * sc, a12 = S6[4] * sc, a22 = S6[5] * sc;
    float v00 = 1.f, v01 = 0.f, v02 = 0.f, v10 = 0.f, v11 = 1.f, v12 = 0.f, v20 = 0.f, v21 = 0.f, v22 = 1.f;
    for (int sweep = 0; sweep < 6; ++sweep) {
        jacobi_rot(&a00, &a11, &a01, &a02, &a12, &v00, &v01, &v10, &v11, &v20, &v21);
        jacobi_rot(&a00, &a22, &a02, &a01, &a12, &v00, &v02, &v10, &v12, &v20, &v22);
        jacobi_rot(&a11, &a22, &a12, &a01, &a02, &v01, &v02, &v11, &v12, &v21, &v22);
    }
    int k = 0; float m = a00;
    if (a11 < m) { m = a11; k = 1; }
    if (a22 < m) { m = a22; k = 2; }
    n[0] = (k == 0) ? v00 : ((k == 1) ? v01 : v02);
    n[1] = (k == 0) ? v10 : ((k == 1) ? v11 : v12);
    n[2] = (k == 0) ? v20 : ((k == 1) ? v21 : v22);
}

/* Same fp32 operation order as geo_forward() in the HIP preprocess kernel (bit-exact contract). */
static void geo_forward(Geo *g, const RefIn *in, int i) {
    const float *V = in->V, *P = in->P;
    const int W = in->W, H = in->H;
    const float fx = (float)W / (2.0f * in->tanfovx), fy = (float)H / (2.0f * in->tanfovy);
    g->m[0] = in->means[3 * i]; g->m[1] = in->means[3 * i + 1]; g->m[2] = in->means[3 * i + 2];
    const float mx = g->m[0], my = g->m[1], mz = g->m[2];
    g->t[0] = V[0] * mx + V[4] * my + V[8] * mz + V[12];
    g->t[1] = V[1] * mx + V[5] * my + V[9] * mz + V[13];
    g->t[2] = V[2] * mx + V[6] * my + V[10] * mz + V[14];
    g->valid = g->t[2] > NEAR_Z;
    g->radius = 0;
    if (!g->valid) return;
    const float tx = g->t[0], ty = g->t[1], tz = g->t[2];
    g->hx = P[0] * mx + P[4] * my + P[8] * mz + P[12];
    g->hy = P[1] * mx + P[5] * my + P[9] * mz + P[13];
    g->hw = P[3] * mx + P[7] * my + P[11] * mz + P[15];
    g->pw = 1.0f / (g->hw + 1e-7f);
    const float ndcx = g->hx * g->pw, ndcy = g->hy * g->pw;
    g->xy[0] = ((ndcx + 1.0f) * (float)W - 1.0f) * 0.5f;
    g->xy[1] = ((ndcy + 1.0f) * (float)H - 1.0f) * 0.5f;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    if (in->cov3d) {        /* the world covariance is an input (render/render.py:52-53): used as it is */
        for (int k = 0; k < 6; ++k) g->S[k] = in->cov3d[6 * i + k];
        for (int k = 0; k < 9; ++k) { g->R[k] = 0.f; g->M[k] = 0.f; }
        g->q[0] = g->q[1] = g->q[2] = g->q[3] = 0.f; g->s[0] = g->s[1] = g->s[2] = 0.f;
    } else {
    for (int k = 0; k < 4; ++k) g->q[k] = in->rots[4 * i + k];
    const float r = g->q[0], x = g->q[1], y = g->q[2], z = g->q[3];
    g->R[0] = 1.0f - 2.0f * (y * y + z * z); g->R[1] = 2.0f * (x * y - r * z); g->R[2] = 2.0f * (x * z + r * y);
    g->R[3] = 2.0f * (x * y + r * z); g->R[4] = 1.0f - 2.0f * (x * x + z * z); g->R[5] = 2.0f * (y * z - r * x);
    g->R[6] = 2.0f * (x * z - r * y); g->R[7] = 2.0f * (y * z + r * x); g->R[8] = 1.0f - 2.0f * (x * x + y * y);
    s0 = in->scales[3 * i]; s1 = in->scales[3 * i + 1]; s2 = in->scales[3 * i + 2];
    g->s[0] = in->scale_modifier * s0; g->s[1] = in->scale_modifier * s1; g->s[2] = in->scale_modifier * s2;
    for (int rr = 0; rr < 3; ++rr) for (int cc = 0; cc < 3; ++cc) g->M[rr * 3 + cc] = g->R[rr * 3 + cc] * g->s[cc];
    const float *M = g->M;
    g->S[0] = M[0] * M[0] + M[1] * M[1] + M[2] * M[2];
    g->S[1] = M[0] * M[3] + M[1] * M[4] + M[2] * M[5];
    g->S[2] = M[0] * M[6] + M[1] * M[7] + M[2] * M[8];
    g->S[3] = M[3] * M[3] + M[4] * M[4] + M[5] * M[5];
    g->S[4] = M[3] * M[6] + M[4] * M[7] + M[5] * M[8];
    g->S[5] = M[6] * M[6] + M[7] * M[7] + M[8] * M[8];
    }
    const float limx = FRUSTUM_CLAMP * in->tanfovx, limy = FRUSTUM_CLAMP * in->tanfovy;
    const float txtz = tx / tz, tytz = ty / tz;
    g->clx = (txtz < -limx) || (txtz > limx);
    g->cly = (tytz < -limy) || (tytz > limy);
    g->txc = fminf(limx, fmaxf(-limx, txtz)) * tz;
    g->tyc = fminf(limy, fmaxf(-limy, tytz)) * tz;
    if (!g->clx) g->txc = tx;
    if (!g->cly) g->tyc = ty;
    const float tz2 = tz * tz;
    g->J00 = fx / tz; g->J02 = -(fx * g->txc) / tz2;
    g->J11 = fy / tz; g->J12 = -(fy * g->tyc) / tz2;
    for (int k = 0; k < 3; ++k) {
        g->T0[k] = g->J00 * WR(V, 0, k) + g->J02 * WR(V, 2, k);
        g->T1[k] = g->J11 * WR(V, 1, k) + g->J12 * WR(V, 2, k);
    }
    const float *S = g->S;
    const float v00 = S[0] * g->T0[0] + S[1] * g->T0[1] + S[2] * g->T0[2];
    const float v01 = S[1] * g->T0[0] + S[3] * g->T0[1] + S[4] * g->T0[2];
    const float v02 = S[2] * g->T0[0] + S[4] * g->T0[1] + S[5] * g->T0[2];
    const float v10 = S[0] * g->T1[0] + S[1] * g->T1[1] + S[2] * g->T1[2];
    const float v11 = S[1] * g->T1[0] + S[3] * g->T1[1] + S[4] * g->T1[2];
    const float v12 = S[2] * g->T1[0] + S[4] * g->T1[1] + S[5] * g->T1[2];
    g->a = (g->T0[0] * v00 + g->T0[1] * v01 + g->T0[2] * v02) + LOWPASS;
    g->b = g->T1[0] * v00 + g->T1[1] * v01 + g->T1[2] * v02;
    g->c = (g->T1[0] * v10 + g->T1[1] * v11 + g->T1[2] * v12) + LOWPASS;
    g->det = g->a * g->c - g->b * g->b;
    if (g->det == 0.0f) { g->valid = 0; return; }
    g->inv = 1.0f / g->det;
    g->conic[0] = g->c * g->inv; g->conic[1] = -g->b * g->inv; g->conic[2] = g->a * g->inv;
    const float mid = 0.5f * (g->a + g->c);
    const float lam = mid + sqrtf(fmaxf(0.1f, mid * mid - g->det));
    g->radius = (int)ceilf(3.0f * sqrtf(lam));
    g->dir[0] = mx - in->cam[0]; g->dir[1] = my - in->cam[1]; g->dir[2] = mz - in->cam[2];
    float n0, n1, n2;
    if (in->cov3d) {
        float ev[3];
        smallest_eigvec(g->S, ev);
        n0 = ev[0]; n1 = ev[1]; n2 = ev[2];
        g->kmin = -1;       /* no rotation-matrix column: the normal's gradient goes nowhere */
    } else {
        g->kmin = 0; float smin = s0;
        if (s1 < smin) { smin = s1; g->kmin = 1; }
        if (s2 < smin) { smin = s2; g->kmin = 2; }
        n0 = g->R[0 + g->kmin]; n1 = g->R[3 + g->kmin]; n2 = g->R[6 + g->kmin];
    }
    g->sign = ((n0 * g->dir[0] + n1 * g->dir[1] + n2 * g->dir[2]) > 0.0f) ? -1.0f : 1.0f;
    g->n[0] = g->sign * n0; g->n[1] = g->sign * n1; g->n[2] = g->sign * n2;
    g->dlen = sqrtf(g->dir[0] * g->dir[0] + g->dir[1] * g->dir[1] + g->dir[2] * g->dir[2]);
    g->dir[0] /= g->dlen; g->dir[1] /= g->dlen; g->dir[2] /= g->dlen;
    for (int k = 0; k < 3; ++k) g->nv[k] = WR(V, k, 0) * g->n[0] + WR(V, k, 1) * g->n[1] + WR(V, k, 2) * g->n[2];
    g->sdot = g->nv[0] * tx + g->nv[1] * ty + g->nv[2] * tz;
    const float tn = sqrtf(tx * tx + ty * ty + tz * tz);
    const float *juv = in->tex ? in->juv : NULL;       /* untextured surface: no UV plane at all */
    g->degen = (juv == NULL) || fabsf(g->sdot) <= PLANE_EPS * tn;
    for (int rr = 0; rr < 3; ++rr) for (int cc = 0; cc < 3; ++cc)
        g->Km[rr * 3 + cc] = (juv == NULL) ? 0.0f
                           : juv[9 * i + rr * 3 + 0] * WR(V, cc, 0) + juv[9 * i + rr * 3 + 1] * WR(V, cc, 1)
                           + juv[9 * i + rr * 3 + 2] * WR(V, cc, 2);
    if (g->degen) {
        g->gx = 0; g->gy = 0; for (int k = 0; k < 6; ++k) g->G[k] = 0;
    } else {
        const float ax = tz * g->nv[0] / g->sdot, ay = tz * g->nv[1] / g->sdot;
        g->gx = ax / fx; g->gy = ay / fy;
        const float B00 = tz / fx - tx * g->gx, B01 = -tx * g->gy;
        const float B10 = -ty * g->gx, B11 = tz / fy - ty * g->gy;
        const float B20 = -tz * g->gx, B21 = -tz * g->gy;
        for (int rr = 0; rr < 3; ++rr) {
            g->G[rr * 2 + 0] = g->Km[rr * 3 + 0] * B00 + g->Km[rr * 3 + 1] * B10 + g->Km[rr * 3 + 2] * B20;
            g->G[rr * 2 + 1] = g->Km[rr * 3 + 0] * B01 + g->Km[rr * 3 + 1] * B11 + g->Km[rr * 3 + 2] * B21;
        }
    }
}

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

static void tile_rect(const Geo *g, int gxn, int gyn, int *x0, int *y0, int *x1, int *y1) {
    const float rf = (float)g->radius;
    *x0 = imin(gxn, imax(0, (int)((g->xy[0] - rf) / (float)TILE)));
    *y0 = imin(gyn, imax(0, (int)((g->xy[1] - rf) / (float)TILE)));
    *x1 = imin(gxn, imax(0, (int)((g->xy[0] + rf + (float)(TILE - 1)) / (float)TILE)));
    *y1 = imin(gyn, imax(0, (int)((g->xy[1] + rf + (float)(TILE - 1)) / (float)TILE)));
}

/* K1+K2: rec[N,24], depth[N], radii[N], rect[N,4], tiles[N], offsets[N]; returns D */
uint32_t texgs_ref_preprocess(const RefIn *in, float *rec, float *depth, int32_t *radii, int32_t *rect,
                              uint32_t *tiles, uint32_t *offsets) {
    const int N = in->N, gxn = (in->W + TILE - 1) / TILE, gyn = (in->H + TILE - 1) / TILE;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i) {
        Geo g;
        geo_forward(&g, in, i);
        int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
        if (g.valid) { tile_rect(&g, gxn, gyn, &x0, &y0, &x1, &y1); if ((x1 - x0) * (y1 - y0) == 0) g.valid = 0; }
        float *r = rec + (size_t)i * REC;
        if (!g.valid) {
            radii[i] = 0; tiles[i] = 0; depth[i] = 0; rect[4 * i] = rect[4 * i + 1] = rect[4 * i + 2] = rect[4 * i + 3] = 0;
            memset(r, 0, sizeof(float) * REC);
            continue;
        }
        float vd[3] = {0, 0, 0};
        const int na = (in->shs && in->sh_degree > 0) ? sh_active(in->sh_degree, in->K) : 0;
        if (na > 0) {
            float b[15];
            sh_basis(in->sh_degree, g.dir[0], g.dir[1], g.dir[2], b);
            const float *sp = in->shs + (size_t)i * in->K * 3;
            for (int k = 0; k < na; ++k) { vd[0] += b[k] * sp[3 * k]; vd[1] += b[k] * sp[3 * k + 1]; vd[2] += b[k] * sp[3 * k + 2]; }
        }
        if (in->coff) { vd[0] += in->coff[3 * i]; vd[1] += in->coff[3 * i + 1]; vd[2] += in->coff[3 * i + 2]; }
        radii[i] = g.radius; tiles[i] = (uint32_t)((x1 - x0) * (y1 - y0)); depth[i] = g.t[2];
        rect[4 * i] = x0; rect[4 * i + 1] = y0; rect[4 * i + 2] = x1; rect[4 * i + 3] = y1;
        r[0] = g.xy[0]; r[1] = g.xy[1]; r[2] = g.conic[0]; r[3] = g.conic[1]; r[4] = g.conic[2]; r[5] = in->opac[i];
        r[6] = g.gx; r[7] = g.gy; for (int k = 0; k < 6; ++k) r[8 + k] = g.G[k];
        if (in->tex && in->uvs) { r[14] = in->uvs[3 * i]; r[15] = in->uvs[3 * i + 1]; r[16] = in->uvs[3 * i + 2]; }
        else { r[14] = 0.f; r[15] = 0.f; r[16] = 1.f; }
        r[17] = vd[0]; r[18] = vd[1]; r[19] = vd[2]; r[20] = g.t[2]; r[21] = g.n[0]; r[22] = g.n[1]; r[23] = g.n[2];
    }
    uint32_t run = 0;
    for (int i = 0; i < N; ++i) { run += tiles[i]; offsets[i] = run; }
    return run;
}

/* K3-K5: keys (tile<<32 | depth bits), stable LSD radix sort (8-bit digits), ranges[T,2] */
void texgs_ref_bin(const RefIn *in, uint32_t D, const float *depth, const int32_t *rect, const uint32_t *tiles,
                   const uint32_t *offsets, uint64_t *keys_unsorted, uint32_t *vals_unsorted, uint64_t *keys_sorted,
                   uint32_t *point_list, uint32_t *ranges) {
    const int N = in->N, gxn = (in->W + TILE - 1) / TILE, gyn = (in->H + TILE - 1) / TILE;
    const int T = gxn * gyn;
    memset(ranges, 0, sizeof(uint32_t) * 2 * (size_t)T);
    if (D == 0) return;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i) {
        if (tiles[i] == 0) continue;
        uint32_t off = offsets[i] - tiles[i];
        uint32_t db; memcpy(&db, &depth[i], 4);
        for (int y = rect[4 * i + 1]; y < rect[4 * i + 3]; ++y)
            for (int x = rect[4 * i]; x < rect[4 * i + 2]; ++x) {
                keys_unsorted[off] = ((uint64_t)(uint32_t)(y * gxn + x) << 32) | db;
                vals_unsorted[off] = (uint32_t)i; ++off;
            }
    }
    int bits = 0; while ((1u << bits) < (uint32_t)T && bits < 31) ++bits; if (bits == 0) bits = 1;
    const int end_bit = 32 + bits;
    uint64_t *ka = (uint64_t *)malloc(sizeof(uint64_t) * D), *kb = (uint64_t *)malloc(sizeof(uint64_t) * D);
    uint32_t *va = (uint32_t *)malloc(sizeof(uint32_t) * D), *vb = (uint32_t *)malloc(sizeof(uint32_t) * D);
    memcpy(ka, keys_unsorted, sizeof(uint64_t) * D); memcpy(va, vals_unsorted, sizeof(uint32_t) * D);
    for (int shift = 0; shift < end_bit; shift += 8) {
        size_t cnt[257]; memset(cnt, 0, sizeof(cnt));
        const int nb = (end_bit - shift) < 8 ? (end_bit - shift) : 8;
        const uint64_t mask = (1ull << nb) - 1ull;
        for (uint32_t i = 0; i < D; ++i) cnt[((ka[i] >> shift) & mask) + 1]++;
        for (int d = 0; d < 256; ++d) cnt[d + 1] += cnt[d];
        for (uint32_t i = 0; i < D; ++i) { size_t p = cnt[(ka[i] >> shift) & mask]++; kb[p] = ka[i]; vb[p] = va[i]; }
        uint64_t *tk = ka; ka = kb; kb = tk; uint32_t *tv = va; va = vb; vb = tv;
    }
    memcpy(keys_sorted, ka, sizeof(uint64_t) * D); memcpy(point_list, va, sizeof(uint32_t) * D);
    free(ka); free(kb); free(va); free(vb);
    for (uint32_t i = 0; i < D; ++i) {
        const uint32_t cur = (uint32_t)(keys_sorted[i] >> 32);
        if (i == 0) ranges[2 * cur] = 0;
        else { const uint32_t prev = (uint32_t)(keys_sorted[i - 1] >> 32); if (cur != prev) { ranges[2 * prev + 1] = i; ranges[2 * cur] = i; } }
        if (i == D - 1) ranges[2 * cur + 1] = D;
    }
}

typedef struct { int o00, o01, o10, o11, axis; float fx, fy, sc, tc, h, rma, sm, su, sv; int x0, y0, face; } Tap;   /* x0, y0: unclamped cell */

static Tap cube_address(float u0, float u1, float u2, int R) {
    Tap t; float m, ua, ub;
    const float a0 = fabsf(u0), a1 = fabsf(u1), a2 = fabsf(u2);
    if (a0 >= a1 && a0 >= a2) { t.axis = 0; m = u0; t.sm = (u0 >= 0.f) ? 1.f : -1.f; ua = u2; t.su = -t.sm; ub = u1; t.sv = -1.f; }
    else if (a1 >= a2)        { t.axis = 1; m = u1; t.sm = (u1 >= 0.f) ? 1.f : -1.f; ua = u0; t.su = 1.f;   ub = u2; t.sv = t.sm; }
    else                      { t.axis = 2; m = u2; t.sm = (u2 >= 0.f) ? 1.f : -1.f; ua = u0; t.su = t.sm;  ub = u1; t.sv = -1.f; }
    const int face = 2 * t.axis + (t.sm > 0.f ? 0 : 1);
    const float ma = fmaxf(fabsf(m), MA_MIN);
    t.rma = RCPF(ma); t.sc = t.su * ua; t.tc = t.sv * ub;
    const float halfR = 0.5f * (float)R;
    t.h = halfR * t.rma;
    const float col = (t.sc * t.rma + 1.0f) * halfR - 0.5f, row = (t.tc * t.rma + 1.0f) * halfR - 0.5f;
    const float x0f = floorf(col), y0f = floorf(row);
    t.fx = col - x0f; t.fy = row - y0f;
    const int x0 = (int)x0f, y0 = (int)y0f;
    const int x0c = imin(imax(x0, 0), R - 1), x1c = imin(imax(x0 + 1, 0), R - 1);
    const int y0c = imin(imax(y0, 0), R - 1), y1c = imin(imax(y0 + 1, 0), R - 1);
    const int fb = face * R;
    t.x0 = x0; t.y0 = y0; t.face = face;
    t.o00 = ((fb + y0c) * R + x0c) * 3; t.o01 = ((fb + y0c) * R + x1c) * 3;
    t.o10 = ((fb + y1c) * R + x0c) * 3; t.o11 = ((fb + y1c) * R + x1c) * 3;
    return t;
}

/* K6: outputs out[8,H,W] (r,g,b,depth,nx,ny,nz,alpha), final_T[H,W], n_contrib[H,W] */
void texgs_ref_render_fwd(const RefIn *in, const float *rec, const uint32_t *point_list, const uint32_t *ranges,
                          float *out, float *final_T, uint32_t *n_contrib) {
    const int W = in->W, H = in->H, gxn = (W + TILE - 1) / TILE, gyn = (H + TILE - 1) / TILE, HW = W * H;
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < gxn * gyn; ++tile) {
        const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        const int tx0 = (tile % gxn) * TILE, ty0 = (tile / gxn) * TILE;
        for (int ly = 0; ly < TILE; ++ly) for (int lx = 0; lx < TILE; ++lx) {
            const int px = tx0 + lx, py = ty0 + ly;
            if (px >= W || py >= H) continue;
            const float pxf = (float)px, pyf = (float)py;
            float T = 1.0f, A[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            uint32_t last = 0;
            for (uint32_t k = r0; k < r1; ++k) {
                const float *r = rec + (size_t)point_list[k] * REC;
                const float dx = r[0] - pxf, dy = r[1] - pyf;
                const float power = -0.5f * (r[2] * dx * dx + r[4] * dy * dy) - r[3] * dx * dy;
                if (power > 0.0f) continue;
                const float alpha = fminf(ALPHA_MAX, r[5] * EXPF(power));
                if (alpha < ALPHA_MIN) continue;
                const float Tn = T * (1.0f - alpha);
                if (Tn < T_EPS) break;
                const float w = alpha * T;
                if (in->tex) {
                    const float dpx = -dx, dpy = -dy;
                    const float den = 1.0f + r[6] * dpx + r[7] * dpy;
                    const float inv = (den >= DEN_MIN) ? RCPF(den) : 0.0f;
                    const float u0 = r[14] + (r[8] * dpx + r[9] * dpy) * inv;
                    const float u1 = r[15] + (r[10] * dpx + r[11] * dpy) * inv;
                    const float u2 = r[16] + (r[12] * dpx + r[13] * dpy) * inv;
                    const Tap ct = cube_address(u0, u1, u2, in->R);
                    const float w00 = (1.f - ct.fx) * (1.f - ct.fy), w01 = ct.fx * (1.f - ct.fy);
                    const float w10 = (1.f - ct.fx) * ct.fy, w11 = ct.fx * ct.fy;
                    for (int ch = 0; ch < 3; ++ch) {
                        const float tv = w00 * in->tex[ct.o00 + ch] + w01 * in->tex[ct.o01 + ch] + w10 * in->tex[ct.o10 + ch]
                                       + w11 * in->tex[ct.o11 + ch];
                        A[ch] += w * fmaxf(0.f, SH_C0 * tv + r[17 + ch] + 0.5f);
                    }
                } else {        /* untextured surface: the colour is the (offset) view-dependent term alone */
                    for (int ch = 0; ch < 3; ++ch) A[ch] += w * fmaxf(0.f, r[17 + ch] + 0.5f);
                }
                A[3] += w * r[20]; A[4] += w * r[21]; A[5] += w * r[22]; A[6] += w * r[23]; A[7] += w;
                T = Tn; last = k - r0 + 1;
            }
            const int pix = py * W + px;
            out[pix] = A[0] + T * in->bg[0]; out[HW + pix] = A[1] + T * in->bg[1]; out[2 * HW + pix] = A[2] + T * in->bg[2];
            for (int ch = 3; ch < 8; ++ch) out[ch * HW + pix] = A[ch];
            final_T[pix] = T; n_contrib[pix] = last;
        }
    }
}

/* Ambiguity map of one forward (TEST ATTRIBUTION ONLY; same loops as texgs_ref_render_fwd, twice per pixel).
 * Two fp32 implementations of this operator can only differ by more than rounding where a DISCRETE decision sits within
 * rounding of its threshold.  margin[H*W] = the smallest relative distance of any such decision taken for the pixel:
 *   alpha vs 1/255 (every tested instance), T (1 - alpha) vs 1e-4 (every instance that passed), power vs 0,
 *   cubemap face (largest vs second largest |uv|) and den vs DEN_MIN (every contributor)
 * -- the same list as oracle/texgs_torch.py render().  A pixel is "forward-ambiguous" when margin < tau_fwd.
 * Gradients have two more discrete selections per contributing (pixel, Gaussian) pair: the bilinear cell (the sample is
 * continuous across a cell edge, its uv-derivative is not) and the max(0, .) of the colour.  Flags (bytes, caller zero-fills):
 *   gflag[N]     the Gaussian's gradient row may legitimately differ: it contributes to (or is within tau_fwd of contributing
 *                to) a forward-ambiguous pixel, or one of its pairs is within tau_cell texels of a cell edge, or within
 *                tau_relu of the colour clamp;
 *   tflag[6RR]   same for a texel: tapped by a contributor of a forward-ambiguous pixel or by a pair within tau_relu of the clamp. */
void texgs_ref_ambiguity_ex(const RefIn *in, const float *rec, const uint32_t *point_list, const uint32_t *ranges,
                            float tau_fwd, float tau_cell, float tau_relu, float *margin, uint8_t *gflag, uint8_t *tflag, float *cond,
                            int own_only);
void texgs_ref_ambiguity(const RefIn *in, const float *rec, const uint32_t *point_list, const uint32_t *ranges,
                         float tau_fwd, float tau_cell, float tau_relu, float *margin, uint8_t *gflag, uint8_t *tflag) {
    texgs_ref_ambiguity_ex(in, rec, point_list, ranges, tau_fwd, tau_cell, tau_relu, margin, gflag, tflag, NULL, 0);
}
/* _ex: also cond[H*W] (may be NULL) = sum over the pixel's contributors of w_i * (|A dx^2| / 2 + |C dy^2| / 2 + |B dx dy|) * 2^-23 --
 * the rounding error of the falloff exponent `power` in fp32, carried to the blend weights.  `power` is a difference of terms that
 * grow with the square of the pixel's distance from the splat centre; for splats hundreds of pixels wide (a 4 112 x 4 112 frame of
 * 3 000 Gaussians, the stress scene's screen-filling ones) those terms reach 10-100 and alpha itself is only known to ~1e-5..1e-4
 * relative, whatever the implementation.  The tests widen a pixel's tolerance by a small multiple of cond (negligible -- ~1e-7 --
 * on the benchmark scenes, whose splats are ~10 px), and the alpha / T decisions below count as ambiguous when their margin is
 * inside that same uncertainty.  Rows (gflag / tflag) are flagged for pixels whose 1/255, power, face or den decisions are
 * ambiguous -- NOT for a T-stop alone: what a marginal stop adds or removes carries a weight below 2e-4 of the pixel.
 * own_only (the pair-level checks): a marginal alpha >= 1/255 decision flags only the row of the Gaussian whose decision it is --
 * the pixel's OTHER contributors see their transmittance change by 1/255 when it flips, which texgs_ref_render_bwd_ex books as
 * mass (2/255 of each of their terms) instead of excusing their rows; power ~ 0, face and den decisions still flag every
 * contributor of the pixel (rare). */
void texgs_ref_ambiguity_ex(const RefIn *in, const float *rec, const uint32_t *point_list, const uint32_t *ranges,
                            float tau_fwd, float tau_cell, float tau_relu, float *margin, uint8_t *gflag, uint8_t *tflag, float *cond,
                            int own_only) {
    const int W = in->W, H = in->H, gxn = (W + TILE - 1) / TILE, gyn = (H + TILE - 1) / TILE;
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < gxn * gyn; ++tile) {
        const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        const int tx0 = (tile % gxn) * TILE, ty0 = (tile / gxn) * TILE;
        for (int ly = 0; ly < TILE; ++ly) for (int lx = 0; lx < TILE; ++lx) {
            const int px = tx0 + lx, py = ty0 + ly;
            if (px >= W || py >= H) continue;
            const float pxf = (float)px, pyf = (float)py;
            float mpix = INFINITY, mpix_hard = INFINITY, mpix_strict = INFINITY, cpix = 0.0f;
            for (int pass = 0; pass < 2; ++pass) {
                /* pass 0: the pixel's margin.  pass 1: flags; in a forward-ambiguous pixel everything that contributes -- or is
                   within tau_fwd of contributing, or comes after a T-threshold stop that is within tau_fwd of not happening --
                   may carry a different gradient */
                const int pix_amb = (pass == 1) && (mpix_hard < tau_fwd);       /* (a T-stop alone does not flag rows) */
                const int pix_strict = (pass == 1) && (mpix_strict < tau_fwd);  /* a power ~ 0, face or den decision */
                int past_marginal_stop = 0;
                float T = 1.0f, tunc = 0.0f;        /* tunc: relative uncertainty of T so far (sum of its factors' uncertainties) */
                for (uint32_t k = r0; k < r1; ++k) {
                    const uint32_t id = point_list[k];
                    const float *r = rec + (size_t)id * REC;
                    const float dx = r[0] - pxf, dy = r[1] - pyf;
                    const float power = -0.5f * (r[2] * dx * dx + r[4] * dy * dy) - r[3] * dx * dy;
                    float m = INFINITY;
                    /* uncertainty of `power` itself (absolute) = of alpha (relative): a few roundings of its largest term */
                    const float pu = 4.0f * 5.9604645e-8f * (0.5f * fabsf(r[2] * dx * dx) + 0.5f * fabsf(r[4] * dy * dy) + fabsf(r[3] * dx * dy));
                    /* power > 0 skips the instance; mathematically power <= 0, so the decision can only differ where |power| is inside
                       its own rounding error (until round 5 a blanket 1e-5: every pixel within ~0.01 px of a splat centre) */
                    if (fabsf(power) <= 4.0f * pu + 1e-20f) m = 0.0f;
                    float m_strict = m;
                    const float araw = r[5] * expf(power);
                    if (power <= 0.0f || m == 0.0f) m = fminf(m, fmaxf(0.0f, fabsf(araw - ALPHA_MIN) / ALPHA_MIN - pu));
                    const float m_own = m;                  /* this instance's own power / alpha decisions */
                    float m_hard = m;
                    const int pass_alpha = (power <= 0.0f) && (fminf(ALPHA_MAX, araw) >= ALPHA_MIN);
                    int contributes = pass_alpha;
                    float Tn = T;
                    if (pass_alpha) {
                        const float alpha = fminf(ALPHA_MAX, araw);
                        Tn = T * (1.0f - alpha);
                        tunc += pu * alpha / fmaxf(1.0f - alpha, 0.01f);
                        const float m_T = fmaxf(0.0f, (fabsf(Tn - T_EPS) / T_EPS - tunc) * 0.2f);      /* T is a long product: 5x wider band than alpha's */
                        m = fminf(m, m_T);
                        if (pass == 0) cpix += alpha * T * pu * 0.25f;          /* (pu carries the factor 4; cond is in units of one rounding) */
                        if (Tn < T_EPS && !past_marginal_stop) {
                            if (pass == 0 || m_T >= tau_fwd) { if (pass == 0) mpix = fminf(mpix, m); break; }
                            past_marginal_stop = 1;          /* pass 1, marginal stop: the other implementation may blend on */
                        }
                    }
                    if (pass == 0) {
                        if (contributes && in->tex) {
                            const float dpx = -dx, dpy = -dy;
                            const float den = 1.0f + r[6] * dpx + r[7] * dpy;
                            const float inv = (den >= DEN_MIN) ? 1.0f / den : 0.0f;
                            float a0 = fabsf(r[14] + (r[8] * dpx + r[9] * dpy) * inv), a1 = fabsf(r[15] + (r[10] * dpx + r[11] * dpy) * inv);
                            float a2 = fabsf(r[16] + (r[12] * dpx + r[13] * dpy) * inv), t;
                            if (a0 < a1) { t = a0; a0 = a1; a1 = t; }
                            if (a1 < a2) { t = a1; a1 = a2; a2 = t; }
                            if (a0 < a1) { t = a0; a0 = a1; a1 = t; }
                            m = fminf(m, (a0 - a1) / fmaxf(a0, MA_MIN));
                            m = fminf(m, fabsf(den - DEN_MIN));
                            m_hard = fminf(m_hard, fminf((a0 - a1) / fmaxf(a0, MA_MIN), fabsf(den - DEN_MIN)));
                            m_strict = fminf(m_strict, fminf((a0 - a1) / fmaxf(a0, MA_MIN), fabsf(den - DEN_MIN)));
                        }
                        mpix = fminf(mpix, m);
                        mpix_hard = fminf(mpix_hard, m_hard);
                        mpix_strict = fminf(mpix_strict, m_strict);
                    } else {
                        const int marginal = pix_amb && (m < tau_fwd || past_marginal_stop);      /* may contribute on the other side */
                        if ((contributes || marginal) && !in->tex) {
                            float m_relu = INFINITY;
                            for (int ch = 0; ch < 3; ++ch) m_relu = fminf(m_relu, fabsf(r[17 + ch] + 0.5f));
                            if ((own_only ? (pix_strict || (pix_amb && m_own < tau_fwd)) : pix_amb) || m_relu < tau_relu) gflag[id] = 1;
                        } else if (contributes || marginal) {
                            const float dpx = -dx, dpy = -dy;
                            const float den = 1.0f + r[6] * dpx + r[7] * dpy;
                            const float inv = (den >= DEN_MIN) ? 1.0f / den : 0.0f;
                            const float u0 = r[14] + (r[8] * dpx + r[9] * dpy) * inv;
                            const float u1 = r[15] + (r[10] * dpx + r[11] * dpy) * inv;
                            const float u2 = r[16] + (r[12] * dpx + r[13] * dpy) * inv;
                            const Tap ct = cube_address(u0, u1, u2, in->R);
                            const float w00 = (1.f - ct.fx) * (1.f - ct.fy), w01 = ct.fx * (1.f - ct.fy);
                            const float w10 = (1.f - ct.fx) * ct.fy, w11 = ct.fx * ct.fy;
                            float m_relu = INFINITY;
                            for (int ch = 0; ch < 3; ++ch) {
                                const float tv = w00 * in->tex[ct.o00 + ch] + w01 * in->tex[ct.o01 + ch] + w10 * in->tex[ct.o10 + ch]
                                               + w11 * in->tex[ct.o11 + ch];
                                m_relu = fminf(m_relu, fabsf(SH_C0 * tv + r[17 + ch] + 0.5f));
                            }
                            const float m_cell = fminf(fminf(ct.fx, 1.0f - ct.fx), fminf(ct.fy, 1.0f - ct.fy));
                            if ((own_only ? (pix_strict || (pix_amb && m_own < tau_fwd)) : pix_amb) || m_cell < tau_cell || m_relu < tau_relu) gflag[id] = 1;
                            if (pix_amb || m_relu < tau_relu) {
                                tflag[ct.o00 / 3] = 1; tflag[ct.o01 / 3] = 1; tflag[ct.o10 / 3] = 1; tflag[ct.o11 / 3] = 1;
                            }
                        }
                    }
                    if (contributes) T = Tn;
                }
            }
            margin[py * W + px] = mpix;
            if (cond) cond[py * W + px] = cpix;
        }
    }
}

static void atomic_addf(float *p, float v) {
#pragma omp atomic
    *p += v;
}
static void atomic_addd(double *p, double v) {
#pragma omp atomic
    *p += v;
}

/* K7: dout[8,H,W] upstream grads; acc[N,24] (double, zero-filled by caller), dtex (float, zero-filled).
 * TEST ATTRIBUTION (optional, fmass != NULL; caller zero-fills fmass[N,24]): the part of every Gaussian's sums that hangs on a
 * bilinear CELL choice within tau_cell texels of flipping -- for each such (pixel, Gaussian) pair, cell_weight x |term| of the
 * terms that carry the texture's uv-derivative (slots 6..16 and the uv share of slots 0, 1).  Across a cell edge the sample is
 * continuous and so are the texture gradient and everything else; only dL/duv of the pair changes -- by O(1) of itself on a
 * white-noise texture (cell_weight 1), by a few per cent on a band-limited one.  Pushed through texgs_ref_preprocess_bwd (which is
 * linear in the sums) this bounds, row by row, how far two fp32 implementations may be apart (tests/helpers.py grad_mass_attributed):
 * the tolerance of EVERY row is widened by what its own near-edge pairs can contribute, instead of excusing a row altogether for
 * having one such pair among its hundreds. */
void texgs_ref_render_bwd_ex(const RefIn *in, const float *rec, const uint32_t *point_list, const uint32_t *ranges,
                             const float *final_T, const uint32_t *n_contrib, const float *dout, double *acc, float *dtex,
                             float tau_cell, float cell_weight, float tex_slope, double *fmass, const float *margin, float tau_fwd);
/* CONDITIONING MASS (round 6; with fmass): the falloff exponent power = -(A dx^2 + C dy^2)/2 - B dx dy is a sum of terms that grow with
 * the square of the distance from the splat centre; under splats hundreds of pixels wide they reach 10..100 and alpha is known only to
 * pu = 4 roundings of the largest term, RELATIVE, in any fp32 implementation (texgs_ref_ambiguity_ex uses the same pu for its decision
 * margins).  Everything the backward sums is a smooth function of the alphas, so to first order a pair's terms move by
 *   its own alpha:           pu_i                                   (w_i, alpha_raw and the Gaussian factor of dL/dpower)
 *   the transmittance:       tunc_i = sum over pairs in front of pu_j alpha_j / (1 - alpha_j)
 *   the blend behind it:     u_all * sum_ch |dL/dout_ch| * (blend of |feature_ch| behind the pair)     (dL/dalpha is a difference
 *                            (f - accum) that may cancel while accum's own uncertainty does not), u_all = max pu + the pixel's total tunc
 * and the background term by the pixel's total.  cond_weight x those bounds are added to fmass, slot by slot, like the cell-edge
 * masses; pushed through the (linear) last stage they widen a row's tolerance by what the conditioning of ITS OWN pairs allows --
 * negligible on the benchmark scenes (pu ~ 1e-7), the whole story in the stress scene.  0 = off. */
static float g_cond_weight = 0.0f;
void texgs_ref_set_cond_weight(float w) { g_cond_weight = w; }
void texgs_ref_render_bwd(const RefIn *in, const float *rec, const uint32_t *point_list, const uint32_t *ranges,
                          const float *final_T, const uint32_t *n_contrib, const float *dout, double *acc, float *dtex) {
    texgs_ref_render_bwd_ex(in, rec, point_list, ranges, final_T, n_contrib, dout, acc, dtex, 0.0f, 0.0f, 0.0f, NULL, NULL, 0.0f);
}
/* tex_slope: (re-used argument) tau_relu for the pair-level treatment of the colour clamp, 0 = off.
 * margin / tau_fwd (optional, with fmass): in a pixel whose forward decisions are within tau_fwd of flipping (texgs_ref_ambiguity's
 * map) a contributor may appear or vanish with alpha ~ 1/255 -- every OTHER pair of the pixel then sees its transmittance, and with
 * it all of its terms, move by at most 1/255: 2/255 of each term is booked as mass.  (The row of the marginal Gaussian itself is
 * the one texgs_ref_ambiguity_ex(own_only) flags.) */
void texgs_ref_render_bwd_ex(const RefIn *in, const float *rec, const uint32_t *point_list, const uint32_t *ranges,
                             const float *final_T, const uint32_t *n_contrib, const float *dout, double *acc, float *dtex,
                             float tau_cell, float cell_weight, float tex_slope, double *fmass, const float *margin, float tau_fwd) {
    const int W = in->W, H = in->H, gxn = (W + TILE - 1) / TILE, gyn = (H + TILE - 1) / TILE, HW = W * H;
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < gxn * gyn; ++tile) {
        const uint32_t r0 = ranges[2 * tile];
        const int tx0 = (tile % gxn) * TILE, ty0 = (tile / gxn) * TILE;
        for (int ly = 0; ly < TILE; ++ly) for (int lx = 0; lx < TILE; ++lx) {
            const int px = tx0 + lx, py = ty0 + ly;
            if (px >= W || py >= H) continue;
            const int pix = py * W + px;
            const float pxf = (float)px, pyf = (float)py;
            const float Tfin = final_T[pix];
            const int last = (int)n_contrib[pix];
            float dpix[8];
            for (int ch = 0; ch < 8; ++ch) dpix[ch] = dout[ch * HW + pix];
            const float bgdot = in->bg[0] * dpix[0] + in->bg[1] * dpix[1] + in->bg[2] * dpix[2];
            float T = Tfin, accum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_alpha = 0, last_f[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            /* conditioning mass: the pixel's totals, front to back (same decisions as the loop below) */
            const float condw = fmass ? g_cond_weight : 0.0f;
            float tunc_total = 0.f, pu_max = 0.f, tsum_back = 0.f, accabs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (condw > 0.0f) {
                for (int pos = 0; pos < last; ++pos) {
                    const float *r = rec + (size_t)point_list[r0 + pos] * REC;
                    const float dx = r[0] - pxf, dy = r[1] - pyf;
                    const float power = -0.5f * (r[2] * dx * dx + r[4] * dy * dy) - r[3] * dx * dy;
                    if (power > 0.0f) continue;
                    const float alpha = fminf(ALPHA_MAX, r[5] * EXPF(power));
                    if (alpha < ALPHA_MIN) continue;
                    const float pu = 4.0f * 5.9604645e-8f * (0.5f * fabsf(r[2] * dx * dx) + 0.5f * fabsf(r[4] * dy * dy) + fabsf(r[3] * dx * dy));
                    tunc_total += pu * alpha / fmaxf(1.0f - alpha, 0.01f);
                    pu_max = fmaxf(pu_max, pu);
                }
            }
            for (int pos = last - 1; pos >= 0; --pos) {
                const uint32_t id = point_list[r0 + pos];
                const float *r = rec + (size_t)id * REC;
                const float dx = r[0] - pxf, dy = r[1] - pyf;
                const float power = -0.5f * (r[2] * dx * dx + r[4] * dy * dy) - r[3] * dx * dy;
                if (power > 0.0f) continue;
                const float Gs = EXPF(power), araw = r[5] * Gs, alpha = fminf(ALPHA_MAX, araw);
                if (alpha < ALPHA_MIN) continue;
                T = T / (1.0f - alpha);
                const float w = alpha * T;
                const float dpx = -dx, dpy = -dy;
                const int textured = in->tex != NULL;
                const float den = 1.0f + r[6] * dpx + r[7] * dpy;
                const int good = textured && den >= DEN_MIN;
                const float inv = good ? RCPF(den) : 0.0f;
                const float nu0 = r[8] * dpx + r[9] * dpy, nu1 = r[10] * dpx + r[11] * dpy, nu2 = r[12] * dpx + r[13] * dpy;
                const float u0 = r[14] + nu0 * inv, u1 = r[15] + nu1 * inv, u2 = r[16] + nu2 * inv;
                Tap ct; memset(&ct, 0, sizeof(ct));
                if (textured) ct = cube_address(u0, u1, u2, in->R);
                const float w00 = (1.f - ct.fx) * (1.f - ct.fy), w01 = ct.fx * (1.f - ct.fy);
                const float w10 = (1.f - ct.fx) * ct.fy, w11 = ct.fx * ct.fy;
                float f[8], pre[3], t00[3] = {0, 0, 0}, t01[3] = {0, 0, 0}, t10[3] = {0, 0, 0}, t11[3] = {0, 0, 0};
                for (int ch = 0; ch < 3; ++ch) {
                    if (textured) {
                        t00[ch] = in->tex[ct.o00 + ch]; t01[ch] = in->tex[ct.o01 + ch];
                        t10[ch] = in->tex[ct.o10 + ch]; t11[ch] = in->tex[ct.o11 + ch];
                        pre[ch] = SH_C0 * (w00 * t00[ch] + w01 * t01[ch] + w10 * t10[ch] + w11 * t11[ch]) + r[17 + ch] + 0.5f;
                    } else pre[ch] = r[17 + ch] + 0.5f;       /* untextured surface (render/render.py:68) */
                    f[ch] = fmaxf(0.f, pre[ch]);
                }
                f[3] = r[20]; f[4] = r[21]; f[5] = r[22]; f[6] = r[23]; f[7] = 1.0f;
                float dLda = 0.f, behind_abs = 0.f;
                for (int ch = 0; ch < 8; ++ch) {
                    accabs[ch] = last_alpha * fabsf(last_f[ch]) + (1.f - last_alpha) * accabs[ch];     /* (uses last_f before it is replaced) */
                    accum[ch] = last_alpha * last_f[ch] + (1.f - last_alpha) * accum[ch];
                    last_f[ch] = f[ch];
                    dLda += (f[ch] - accum[ch]) * dpix[ch];
                    behind_abs += accabs[ch] * fabsf(dpix[ch]);
                }
                last_alpha = alpha;
                const float dLda_sum = dLda;            /* sum_ch (f - accum) dL/dout, before the factor T */
                dLda *= T;
                dLda += (-Tfin / (1.0f - alpha)) * bgdot;
                const float dLdp = araw * dLda;
                const float gdx = -(r[2] * dx + r[3] * dy), gdy = -(r[4] * dy + r[3] * dx);
                float part[REC];
                memset(part, 0, sizeof(part));
                part[0] = dLdp * gdx; part[1] = dLdp * gdy;
                part[2] = -0.5f * dx * dx * dLdp; part[3] = -dx * dy * dLdp; part[4] = -0.5f * dy * dy * dLdp;
                part[5] = Gs * dLda;
                part[20] = w * dpix[3]; part[21] = w * dpix[4]; part[22] = w * dpix[5]; part[23] = w * dpix[6];
                float dtexv[3], dLdcol = 0.f, dLdrow = 0.f;
                for (int ch = 0; ch < 3; ++ch) {
                    const float dc = (pre[ch] > 0.f) ? w * dpix[ch] : 0.f;
                    part[17 + ch] = dc; dtexv[ch] = textured ? SH_C0 * dc : 0.f;
                    if (dtexv[ch] != 0.f) {
                        atomic_addf(dtex + ct.o00 + ch, w00 * dtexv[ch]); atomic_addf(dtex + ct.o01 + ch, w01 * dtexv[ch]);
                        atomic_addf(dtex + ct.o10 + ch, w10 * dtexv[ch]); atomic_addf(dtex + ct.o11 + ch, w11 * dtexv[ch]);
                    }
                    dLdcol += dtexv[ch] * ((1.f - ct.fy) * (t01[ch] - t00[ch]) + ct.fy * (t11[ch] - t10[ch]));
                    dLdrow += dtexv[ch] * ((1.f - ct.fx) * (t10[ch] - t00[ch]) + ct.fx * (t11[ch] - t01[ch]));
                }
                const float dua = dLdcol * ct.su * ct.h, dub = dLdrow * ct.sv * ct.h;
                const float dum = -(dLdcol * ct.sc + dLdrow * ct.tc) * ct.h * ct.rma * ct.sm;
                float du0, du1, du2;
                if (ct.axis == 0) { du0 = dum; du2 = dua; du1 = dub; }
                else if (ct.axis == 1) { du1 = dum; du0 = dua; du2 = dub; }
                else { du2 = dum; du0 = dua; du1 = dub; }
                part[14] = du0; part[15] = du1; part[16] = du2;
                float uv0 = 0.f, uv1 = 0.f;        /* the uv share of the xy slots */
                if (good) {
                    const float dn0 = du0 * inv, dn1 = du1 * inv, dn2 = du2 * inv;
                    const float dden = -(du0 * nu0 + du1 * nu1 + du2 * nu2) * inv * inv;
                    part[8] = dn0 * dpx; part[9] = dn0 * dpy; part[10] = dn1 * dpx; part[11] = dn1 * dpy;
                    part[12] = dn2 * dpx; part[13] = dn2 * dpy; part[6] = dden * dpx; part[7] = dden * dpy;
                    uv0 = (r[8] * dn0 + r[10] * dn1 + r[12] * dn2) + r[6] * dden;
                    uv1 = (r[9] * dn0 + r[11] * dn1 + r[13] * dn2) + r[7] * dden;
                    part[0] -= uv0;
                    part[1] -= uv1;
                }
                double *ap = acc + (size_t)id * REC;
                for (int k = 0; k < REC; ++k) if (part[k] != 0.f) atomic_addd(ap + k, (double)part[k]);
                if (condw > 0.0f) {
                    const float pu = 4.0f * 5.9604645e-8f * (0.5f * fabsf(r[2] * dx * dx) + 0.5f * fabsf(r[4] * dy * dy) + fabsf(r[3] * dx * dy));
                    const float term = pu * alpha / fmaxf(1.0f - alpha, 0.01f);
                    const float tunc_i = fmaxf(0.0f, tunc_total - tsum_back - term);      /* pairs strictly in front of this one */
                    tsum_back += term;
                    const float u_all = pu_max + tunc_total;
                    const float rel_w = pu + tunc_i;
                    const float dLda_unc = T * (tunc_i * fabsf(dLda_sum) + u_all * behind_abs)
                                         + (Tfin / fmaxf(1.0f - alpha, 0.01f)) * fabsf(bgdot) * (tunc_total + term);
                    const float cdl[6] = {araw * gdx, araw * gdy, -0.5f * dx * dx * araw, -dx * dy * araw, -0.5f * dy * dy * araw, Gs};
                    double *fp = fmass + (size_t)id * REC;
                    for (int k = 0; k < 6; ++k) {
                        const float m = fabsf(cdl[k]) * (dLda_unc + fabsf(dLda) * pu);
                        if (m != 0.f) atomic_addd(fp + k, (double)(condw * m));
                    }
                    if (uv0 != 0.f) atomic_addd(fp + 0, (double)(condw * fabsf(uv0) * rel_w));
                    if (uv1 != 0.f) atomic_addd(fp + 1, (double)(condw * fabsf(uv1) * rel_w));
                    for (int k = 6; k < REC; ++k) if (part[k] != 0.f) atomic_addd(fp + k, (double)(condw * fabsf(part[k]) * rel_w));
                }
                if (fmass && tex_slope > 0.0f) {
                    /* colour clamp max(0, pre) within tex_slope (= tau_relu) of switching for a channel: the pair's colour gradient of
                       that channel is all or nothing -- w dL/dcolour for the view-dependent slots, and through C0 the texture's
                       derivative for the uv slots (bounded with absolute values) */
                    float dcol_b = 0.f, drow_b = 0.f; int any = 0;
                    double *fp = fmass + (size_t)id * REC;
                    for (int ch = 0; ch < 3; ++ch) {
                        if (fabsf(pre[ch]) >= tex_slope) continue;
                        any = 1;
                        const float pot = fabsf(w * dpix[ch]);
                        if (pot != 0.f) atomic_addd(fp + 17 + ch, (double)pot);
                        if (textured) {
                            dcol_b += SH_C0 * pot * fabsf((1.f - ct.fy) * (t01[ch] - t00[ch]) + ct.fy * (t11[ch] - t10[ch]));
                            drow_b += SH_C0 * pot * fabsf((1.f - ct.fx) * (t10[ch] - t00[ch]) + ct.fx * (t11[ch] - t01[ch]));
                        }
                    }
                    if (any && textured) {
                        const float ea = dcol_b * ct.h, eb = drow_b * ct.h, em = (dcol_b * fabsf(ct.sc) + drow_b * fabsf(ct.tc)) * ct.h * ct.rma;
                        float e0, e1, e2;
                        if (ct.axis == 0) { e0 = em; e2 = ea; e1 = eb; } else if (ct.axis == 1) { e1 = em; e0 = ea; e2 = eb; } else { e2 = em; e0 = ea; e1 = eb; }
                        float q[REC]; memset(q, 0, sizeof(q)); float q0 = 0.f, q1 = 0.f;
                        q[14] = e0; q[15] = e1; q[16] = e2;
                        if (good) {
                            const float n0 = e0 * inv, n1 = e1 * inv, n2 = e2 * inv;
                            const float dd = (e0 * fabsf(nu0) + e1 * fabsf(nu1) + e2 * fabsf(nu2)) * inv * inv;
                            q[8] = n0 * fabsf(dpx); q[9] = n0 * fabsf(dpy); q[10] = n1 * fabsf(dpx); q[11] = n1 * fabsf(dpy);
                            q[12] = n2 * fabsf(dpx); q[13] = n2 * fabsf(dpy); q[6] = dd * fabsf(dpx); q[7] = dd * fabsf(dpy);
                            q0 = (fabsf(r[8]) * n0 + fabsf(r[10]) * n1 + fabsf(r[12]) * n2) + fabsf(r[6]) * dd;
                            q1 = (fabsf(r[9]) * n0 + fabsf(r[11]) * n1 + fabsf(r[13]) * n2) + fabsf(r[7]) * dd;
                        }
                        for (int k = 6; k < 17; ++k) if (q[k] != 0.f) atomic_addd(fp + k, (double)q[k]);
                        if (q0 != 0.f) atomic_addd(fp + 0, (double)q0);
                        if (q1 != 0.f) atomic_addd(fp + 1, (double)q1);
                    }
                }
                if (fmass && margin && margin[pix] < tau_fwd) {
                    double *fp = fmass + (size_t)id * REC;
                    for (int k = 0; k < REC; ++k) if (part[k] != 0.f) atomic_addd(fp + k, (double)((2.0f / 255.0f) * fabsf(part[k])));
                }
                if (fmass && textured && fminf(fminf(ct.fx, 1.0f - ct.fx), fminf(ct.fy, 1.0f - ct.fy)) < tau_cell) {
                    /* What would this pair contribute had the sample been taken in the NEIGHBOURING bilinear cell (the sample value
                       is the same there -- bilinear interpolation is continuous -- its uv-derivative is not)?  The uv terms are
                       linear in dL/duv, so the change is exactly the terms evaluated on the DIFFERENCE of the two cells' derivatives.
                       Up to three neighbours when the sample sits near a corner; the largest change per term is the pair's mass. */
                    const int nx = fminf(ct.fx, 1.0f - ct.fx) < tau_cell, ny = fminf(ct.fy, 1.0f - ct.fy) < tau_cell;
                    const int sx = (ct.fx < 0.5f) ? -1 : 1, sy = (ct.fy < 0.5f) ? -1 : 1, R_ = in->R, fb = ct.face * R_;
                    float mass[REC]; memset(mass, 0, sizeof(mass)); float m_uv0 = 0.f, m_uv1 = 0.f;
                    for (int alt = 0; alt < 3; ++alt) {
                        const int ax = (alt == 0 || alt == 2) ? sx : 0, ay = (alt == 1 || alt == 2) ? sy : 0;
                        if ((ax && !nx) || (ay && !ny) || (!ax && !ay)) continue;
                        const float fxa = ct.fx - (float)ax, fya = ct.fy - (float)ay;          /* the same point in the neighbour's frame */
                        const int xa0 = imin(imax(ct.x0 + ax, 0), R_ - 1), xa1 = imin(imax(ct.x0 + ax + 1, 0), R_ - 1);
                        const int ya0 = imin(imax(ct.y0 + ay, 0), R_ - 1), ya1 = imin(imax(ct.y0 + ay + 1, 0), R_ - 1);
                        float dcol = 0.f, drow = 0.f;
                        for (int ch = 0; ch < 3; ++ch) {
                            const float a00 = in->tex[((fb + ya0) * R_ + xa0) * 3 + ch], a01 = in->tex[((fb + ya0) * R_ + xa1) * 3 + ch];
                            const float a10 = in->tex[((fb + ya1) * R_ + xa0) * 3 + ch], a11 = in->tex[((fb + ya1) * R_ + xa1) * 3 + ch];
                            const float gc_alt = (1.f - fya) * (a01 - a00) + fya * (a11 - a10), gr_alt = (1.f - fxa) * (a10 - a00) + fxa * (a11 - a01);
                            const float gc_cur = (1.f - ct.fy) * (t01[ch] - t00[ch]) + ct.fy * (t11[ch] - t10[ch]);
                            const float gr_cur = (1.f - ct.fx) * (t10[ch] - t00[ch]) + ct.fx * (t11[ch] - t01[ch]);
                            dcol += dtexv[ch] * (gc_alt - gc_cur); drow += dtexv[ch] * (gr_alt - gr_cur);
                        }
                        const float ea = dcol * ct.su * ct.h, eb = drow * ct.sv * ct.h;
                        const float em = -(dcol * ct.sc + drow * ct.tc) * ct.h * ct.rma * ct.sm;
                        float e0, e1, e2;
                        if (ct.axis == 0) { e0 = em; e2 = ea; e1 = eb; }
                        else if (ct.axis == 1) { e1 = em; e0 = ea; e2 = eb; }
                        else { e2 = em; e0 = ea; e1 = eb; }
                        float q[REC]; memset(q, 0, sizeof(q)); float q0 = 0.f, q1 = 0.f;
                        q[14] = e0; q[15] = e1; q[16] = e2;
                        if (good) {
                            const float n0 = e0 * inv, n1 = e1 * inv, n2 = e2 * inv;
                            const float dd = -(e0 * nu0 + e1 * nu1 + e2 * nu2) * inv * inv;
                            q[8] = n0 * dpx; q[9] = n0 * dpy; q[10] = n1 * dpx; q[11] = n1 * dpy; q[12] = n2 * dpx; q[13] = n2 * dpy;
                            q[6] = dd * dpx; q[7] = dd * dpy;
                            q0 = (r[8] * n0 + r[10] * n1 + r[12] * n2) + r[6] * dd; q1 = (r[9] * n0 + r[11] * n1 + r[13] * n2) + r[7] * dd;
                        }
                        for (int k = 6; k < 17; ++k) mass[k] = fmaxf(mass[k], fabsf(q[k]));
                        m_uv0 = fmaxf(m_uv0, fabsf(q0)); m_uv1 = fmaxf(m_uv1, fabsf(q1));
                    }
                    double *fp = fmass + (size_t)id * REC;
                    for (int k = 6; k < 17; ++k) if (mass[k] != 0.f) atomic_addd(fp + k, (double)(cell_weight * mass[k]));
                    if (m_uv0 != 0.f) atomic_addd(fp + 0, (double)(cell_weight * m_uv0));
                    if (m_uv1 != 0.f) atomic_addd(fp + 1, (double)(cell_weight * m_uv1));
                    (void)uv0; (void)uv1;
                }
            }
        }
    }
}

/* K8: acc[N,24] -> input gradients.  d_coff [N,3] (dL/dcolor_offset) and d_cov [N,6] (dL/dcov3D, off-diagonal entries carrying
 * both symmetric halves: the lineage's layout) may be NULL; with in->cov3d, d_scales / d_rots may be NULL and receive nothing. */
void texgs_ref_preprocess_bwd(const RefIn *in, const int32_t *radii, const double *acc, float *d_means, float *d_means2D,
                              float *d_shs, float *d_op, float *d_scales, float *d_rots, float *d_uvs, float *d_coff, float *d_cov) {
    const int N = in->N, K = in->K;
    const float fx = (float)in->W / (2.0f * in->tanfovx), fy = (float)in->H / (2.0f * in->tanfovy);
    const float *V = in->V, *P = in->P;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < N; ++i) {
        for (int k = 0; k < 3; ++k) {
            d_means[3 * i + k] = 0; d_means2D[3 * i + k] = 0;
            if (d_scales) d_scales[3 * i + k] = 0;
            if (d_uvs) d_uvs[3 * i + k] = 0;
            if (d_coff) d_coff[3 * i + k] = 0;
        }
        if (d_rots) for (int k = 0; k < 4; ++k) d_rots[4 * i + k] = 0;
        if (d_cov) for (int k = 0; k < 6; ++k) d_cov[6 * i + k] = 0;
        d_op[i] = 0;
        if (d_shs) for (int k = 0; k < 3 * K; ++k) d_shs[(size_t)i * 3 * K + k] = 0;
        if (radii[i] <= 0) continue;
        Geo g; geo_forward(&g, in, i);
        float A[REC]; for (int k = 0; k < REC; ++k) A[k] = (float)acc[(size_t)i * REC + k];
        const float tx = g.t[0], ty = g.t[1], tz = g.t[2];
        float dt[3] = {0, 0, 0}, dm[3] = {0, 0, 0}, dR[9]; for (int k = 0; k < 9; ++k) dR[k] = 0;
        d_op[i] = A[5];
        if (d_uvs) { d_uvs[3 * i] = A[14]; d_uvs[3 * i + 1] = A[15]; d_uvs[3 * i + 2] = A[16]; }
        if (d_coff) { d_coff[3 * i] = A[17]; d_coff[3 * i + 1] = A[18]; d_coff[3 * i + 2] = A[19]; }
        const float dA = A[2], dB = A[3], dC = A[4], inv = g.inv, inv2 = inv * inv;
        const float da = dA * (-g.c * g.c * inv2) + dB * (g.b * g.c * inv2) + dC * (inv - g.a * g.c * inv2);
        const float db = dA * (2.0f * g.b * g.c * inv2) + dB * (-inv - 2.0f * g.b * g.b * inv2) + dC * (2.0f * g.a * g.b * inv2);
        const float dc = dA * (inv - g.a * g.c * inv2) + dB * (g.a * g.b * inv2) + dC * (-g.a * g.a * inv2);
        const float hb = 0.5f * db; const float *S = g.S;
        float TS0[3], TS1[3], dT0[3], dT1[3];
        TS0[0] = g.T0[0] * S[0] + g.T0[1] * S[1] + g.T0[2] * S[2]; TS0[1] = g.T0[0] * S[1] + g.T0[1] * S[3] + g.T0[2] * S[4];
        TS0[2] = g.T0[0] * S[2] + g.T0[1] * S[4] + g.T0[2] * S[5];
        TS1[0] = g.T1[0] * S[0] + g.T1[1] * S[1] + g.T1[2] * S[2]; TS1[1] = g.T1[0] * S[1] + g.T1[1] * S[3] + g.T1[2] * S[4];
        TS1[2] = g.T1[0] * S[2] + g.T1[1] * S[4] + g.T1[2] * S[5];
        for (int k = 0; k < 3; ++k) { dT0[k] = 2.0f * (da * TS0[k] + hb * TS1[k]); dT1[k] = 2.0f * (hb * TS0[k] + dc * TS1[k]); }
        float dS[9], dM[9], dscale[3];
        for (int k = 0; k < 3; ++k) for (int l = 0; l < 3; ++l)
            dS[k * 3 + l] = g.T0[k] * g.T0[l] * da + (g.T0[k] * g.T1[l] + g.T1[k] * g.T0[l]) * hb + g.T1[k] * g.T1[l] * dc;
        if (in->cov3d && d_cov) {
            d_cov[6 * i] = dS[0]; d_cov[6 * i + 1] = dS[1] + dS[3]; d_cov[6 * i + 2] = dS[2] + dS[6];
            d_cov[6 * i + 3] = dS[4]; d_cov[6 * i + 4] = dS[5] + dS[7]; d_cov[6 * i + 5] = dS[8];
        }
        for (int rr = 0; rr < 3; ++rr) for (int cc = 0; cc < 3; ++cc)
            dM[rr * 3 + cc] = 2.0f * (dS[rr * 3] * g.M[cc] + dS[rr * 3 + 1] * g.M[3 + cc] + dS[rr * 3 + 2] * g.M[6 + cc]);
        for (int cc = 0; cc < 3; ++cc) {
            dscale[cc] = (dM[cc] * g.R[cc] + dM[3 + cc] * g.R[3 + cc] + dM[6 + cc] * g.R[6 + cc]) * in->scale_modifier;
            for (int rr = 0; rr < 3; ++rr) dR[rr * 3 + cc] += dM[rr * 3 + cc] * g.s[cc];
        }
        const float dJ00 = dT0[0] * WR(V, 0, 0) + dT0[1] * WR(V, 0, 1) + dT0[2] * WR(V, 0, 2);
        const float dJ02 = dT0[0] * WR(V, 2, 0) + dT0[1] * WR(V, 2, 1) + dT0[2] * WR(V, 2, 2);
        const float dJ11 = dT1[0] * WR(V, 1, 0) + dT1[1] * WR(V, 1, 1) + dT1[2] * WR(V, 1, 2);
        const float dJ12 = dT1[0] * WR(V, 2, 0) + dT1[1] * WR(V, 2, 1) + dT1[2] * WR(V, 2, 2);
        const float tz2 = tz * tz, tz3 = tz2 * tz;
        dt[2] += -dJ00 * fx / tz2 - dJ11 * fy / tz2 + 2.0f * dJ02 * fx * g.txc / tz3 + 2.0f * dJ12 * fy * g.tyc / tz3;
        if (!g.clx) dt[0] += -dJ02 * fx / tz2;
        if (!g.cly) dt[1] += -dJ12 * fy / tz2;
        const float dndx = A[0] * 0.5f * (float)in->W, dndy = A[1] * 0.5f * (float)in->H;
        d_means2D[3 * i] = dndx; d_means2D[3 * i + 1] = dndy;
        {
            const float dhx = dndx * g.pw, dhy = dndy * g.pw, dhw = -(dndx * g.hx + dndy * g.hy) * g.pw * g.pw;
            for (int k = 0; k < 3; ++k) dm[k] += dhx * P[k * 4] + dhy * P[k * 4 + 1] + dhw * P[k * 4 + 3];
        }
        dt[2] += A[20];
        float dn[3] = {A[21], A[22], A[23]};
        if (!g.degen) {
            const float *dG = &A[8]; float dBm[6];
            for (int k = 0; k < 3; ++k) for (int cc = 0; cc < 2; ++cc)
                dBm[k * 2 + cc] = g.Km[k] * dG[cc] + g.Km[3 + k] * dG[2 + cc] + g.Km[6 + k] * dG[4 + cc];
            float dtz = dBm[0] / fx + dBm[3] / fy, dgt[2];
            for (int cc = 0; cc < 2; ++cc) dgt[cc] = A[6 + cc] - (dBm[cc] * tx + dBm[2 + cc] * ty + dBm[4 + cc] * tz);
            for (int k = 0; k < 3; ++k) dt[k] -= dBm[k * 2] * g.gx + dBm[k * 2 + 1] * g.gy;
            const float dax = dgt[0] / fx, day = dgt[1] / fy, dot_a_nv = dax * g.nv[0] + day * g.nv[1];
            dtz += dot_a_nv / g.sdot;
            float dnv[3] = {tz * dax / g.sdot, tz * day / g.sdot, 0.f};
            const float ds = -tz * dot_a_nv / (g.sdot * g.sdot);
            dnv[0] += ds * tx; dnv[1] += ds * ty; dnv[2] += ds * tz;
            dt[0] += ds * g.nv[0]; dt[1] += ds * g.nv[1]; dt[2] += ds * g.nv[2] + dtz;
            for (int k = 0; k < 3; ++k) dn[k] += dnv[0] * WR(V, 0, k) + dnv[1] * WR(V, 1, k) + dnv[2] * WR(V, 2, k);
        }
        if (g.kmin >= 0) { dR[g.kmin] += g.sign * dn[0]; dR[3 + g.kmin] += g.sign * dn[1]; dR[6 + g.kmin] += g.sign * dn[2]; }
        const int na = (in->shs && in->sh_degree > 0) ? sh_active(in->sh_degree, K) : 0;
        if (d_shs && na > 0) {
            float b[15], bx[15], by[15], bz[15];
            sh_basis(in->sh_degree, g.dir[0], g.dir[1], g.dir[2], b);
            sh_basis_grad(in->sh_degree, g.dir[0], g.dir[1], g.dir[2], bx, by, bz);
            const float *sp = in->shs + (size_t)i * K * 3; float *dsp = d_shs + (size_t)i * K * 3;
            const float v0 = A[17], v1 = A[18], v2 = A[19];
            float ddx = 0, ddy = 0, ddz = 0;
            for (int k = 0; k < na; ++k) {
                dsp[3 * k] = b[k] * v0; dsp[3 * k + 1] = b[k] * v1; dsp[3 * k + 2] = b[k] * v2;
                const float w = sp[3 * k] * v0 + sp[3 * k + 1] * v1 + sp[3 * k + 2] * v2;
                ddx += bx[k] * w; ddy += by[k] * w; ddz += bz[k] * w;
            }
            const float dd = g.dir[0] * ddx + g.dir[1] * ddy + g.dir[2] * ddz;
            dm[0] += (ddx - g.dir[0] * dd) / g.dlen; dm[1] += (ddy - g.dir[1] * dd) / g.dlen; dm[2] += (ddz - g.dir[2] * dd) / g.dlen;
        }
        for (int k = 0; k < 3; ++k) dm[k] += dt[0] * V[k * 4] + dt[1] * V[k * 4 + 1] + dt[2] * V[k * 4 + 2];
        for (int k = 0; k < 3; ++k) d_means[3 * i + k] = dm[k];
        if (in->cov3d || !d_scales || !d_rots) continue;
        for (int k = 0; k < 3; ++k) d_scales[3 * i + k] = dscale[k];
        const float r = g.q[0], x = g.q[1], y = g.q[2], z = g.q[3];
        d_rots[4 * i] = 2.0f * (-z * dR[1] + y * dR[2] + z * dR[3] - x * dR[5] - y * dR[6] + x * dR[7]);
        d_rots[4 * i + 1] = 2.0f * (y * dR[1] + z * dR[2] + y * dR[3] - 2.0f * x * dR[4] - r * dR[5] + z * dR[6] + r * dR[7] - 2.0f * x * dR[8]);
        d_rots[4 * i + 2] = 2.0f * (-2.0f * y * dR[0] + x * dR[1] + r * dR[2] + x * dR[3] + z * dR[5] - r * dR[6] + z * dR[7] - 2.0f * y * dR[8]);
        d_rots[4 * i + 3] = 2.0f * (-2.0f * z * dR[0] - r * dR[1] + x * dR[2] + r * dR[3] - 2.0f * z * dR[4] + y * dR[5] + x * dR[6] + y * dR[7]);
    }
}

int texgs_ref_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void texgs_ref_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* DIAGNOSTIC (tests / sizing only): per 8x8 pixel block, how many distinct 32x32-texel texture bins do the bilinear footprints of
 * its contributing (pixel, Gaussian) pairs fall into, and how many of the footprints would miss a DIRECT-MAPPED table of
 * `slots` (16 / 32 / 64) entries (first bin to arrive owns its slot; hash 0: low bits of the bin's (x, y) inside the face,
 * hash 1: the same xor-folded with the face)?  The product's K6 keeps such a table per block (csrc/render.hip); this is the sizing
 * evidence for it.  out_blocks[4 * T][3] = {distinct bins (capped at 1024), footprints, footprints that miss}. */
void texgs_ref_block_bin_stats(const RefIn *in, const float *rec, const uint32_t *point_list, const uint32_t *ranges,
                               int slots, int hash, uint32_t *out_blocks) {
    const int W = in->W, H = in->H, gxn = (W + TILE - 1) / TILE, gyn = (H + TILE - 1) / TILE;
    const int nb = (in->R + 31) >> 5;
    const int lx2 = (slots >= 64) ? 3 : 2, ly2 = (slots >= 32) ? 3 : 2;        /* 16: 2+2 bits, 32: 2+3, 64: 3+3 */
#pragma omp parallel for schedule(dynamic, 1)
    for (int tile = 0; tile < gxn * gyn; ++tile) {
        const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
        const int tx0 = (tile % gxn) * TILE, ty0 = (tile / gxn) * TILE;
        for (int blk = 0; blk < 4; ++blk) {
            uint32_t seen[1024]; int nseen = 0;
            uint32_t slot[64]; for (int k = 0; k < 64; ++k) slot[k] = 0xFFFFFFFFu;
            uint32_t nfoot = 0, nmiss = 0;
            /* K6 walks the list in depth order for the whole block; here pixel by pixel -- the OWNER of a slot may differ from
               the kernel's (first to arrive in list order), the counts are representative, not identical */
            for (int ly = 0; ly < 8; ++ly) for (int lx = 0; lx < 8; ++lx) {
                const int px = tx0 + ((blk & 1) << 3) + lx, py = ty0 + ((blk >> 1) << 3) + ly;
                if (px >= W || py >= H) continue;
                const float pxf = (float)px, pyf = (float)py;
                float T = 1.0f;
                for (uint32_t k = r0; k < r1; ++k) {
                    const float *r = rec + (size_t)point_list[k] * REC;
                    const float dx = r[0] - pxf, dy = r[1] - pyf;
                    const float power = -0.5f * (r[2] * dx * dx + r[4] * dy * dy) - r[3] * dx * dy;
                    if (power > 0.0f) continue;
                    const float alpha = fminf(ALPHA_MAX, r[5] * expf(power));
                    if (alpha < ALPHA_MIN) continue;
                    const float Tn = T * (1.0f - alpha);
                    if (Tn < T_EPS) break;
                    const float dpx = -dx, dpy = -dy;
                    const float den = 1.0f + r[6] * dpx + r[7] * dpy;
                    const float inv = (den >= DEN_MIN) ? 1.0f / den : 0.0f;
                    const Tap ct = cube_address(r[14] + (r[8] * dpx + r[9] * dpy) * inv, r[15] + (r[10] * dpx + r[11] * dpy) * inv,
                                                r[16] + (r[12] * dpx + r[13] * dpy) * inv, in->R);
                    T = Tn;
                    if (ct.o01 == ct.o00 || ct.o10 == ct.o00) continue;            /* clamped at a face border: never binned */
                    const int texel = ct.o00 / 3, x0 = texel % in->R, y0 = (texel / in->R) % in->R, face = texel / (in->R * in->R);
                    const uint32_t bin = (uint32_t)((face * nb + (y0 >> 5)) * nb + (x0 >> 5));
                    ++nfoot;
                    int f = 0; for (; f < nseen; ++f) if (seen[f] == bin) break;
                    if (f == nseen && nseen < 1024) seen[nseen++] = bin;
                    int h = ((x0 >> 5) & ((1 << lx2) - 1)) | (((y0 >> 5) & ((1 << ly2) - 1)) << lx2);
                    if (hash == 1) h = (h ^ (face * 11)) & (slots - 1);
                    if (slot[h] == 0xFFFFFFFFu) slot[h] = bin;
                    if (slot[h] != bin) ++nmiss;
                }
            }
            uint32_t *o = out_blocks + 3 * (size_t)(4 * tile + blk);
            o[0] = (uint32_t)nseen; o[1] = nfoot; o[2] = nmiss;
        }
    }
}
