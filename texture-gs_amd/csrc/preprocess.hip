// K1 preprocess forward, K8 preprocess backward, K9 mark_visible -- gfx950.
//
// Built with -ffp-contract=off: every quantity that feeds an integer stage (depth bits of the sort key,
// pixel centre -> tile rectangle, radius) is an explicitly ordered sequence of IEEE fp32 operations with
// correctly rounded '/' and sqrt, so the C oracle (oracle/texgs_ref.c, same operation order, also built
// without contraction) reproduces keys / rects / radii bit for bit.
//
// One thread per Gaussian, 256 threads (4 wave64) per workgroup.  HBM-bound: reads 92+12K B, writes a 32-B test record
// + an 80-B shading record + 24 B of SoA state per Gaussian.
#include "common.h"

namespace {

struct Frame {          // per-frame matrices, read through wave-uniform (scalar) loads
    float V[16];        // row-vector view matrix: t = [m,1] @ V
    float P[16];        // row-vector full projection
    float cam[3];
};

__device__ __forceinline__ Frame load_frame(const float* vm, const float* pm, const float* cp) {
    Frame f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { f.V[i] = vm[i]; f.P[i] = pm[i]; }
    f.cam[0] = cp[0]; f.cam[1] = cp[1]; f.cam[2] = cp[2];
    return f;
}

// Forward intermediates of one Gaussian, shared by K1 and K8 (K8 recomputes instead of saving ~200 B).
struct Geo {
    bool  valid;
    float m[3], t[3];
    float hx, hy, hw, pw, xy[2];
    float q[4], s[3], R[9], M[9], S[6];           // S: xx,xy,xz,yy,yz,zz
    float txc, tyc; bool clx, cly;
    float J00, J02, J11, J12, T0[3], T1[3];
    float a, b, c, det, inv, conic[3];
    int   radius;
    int   kmin; float sign; float n[3];
    float nv[3], sdot; bool degen; float gx, gy, G[6], K[9];
    float dir[3], dlen;
};

// Wr[r][c] (world->view rotation, column-vector form) = V[c*4 + r]
#define WR(F, r, c) ((F).V[(c) * 4 + (r)])

// Unit eigenvector of the SMALLEST eigenvalue of the symmetric 3x3 matrix S6 = (xx,xy,xz,yy,yz,zz): cyclic Jacobi, 6 sweeps,
// everything in registers (fixed rotation order (0,1), (0,2), (1,2); the matrix is first scaled by 1 / trace so that the
// products of a 1e-5-sized covariance stay far from underflow).  Used only with cov3D_precomp, where there is no rotation matrix
// whose column could be taken: "the shortest axis" of the splat is this eigenvector.
__device__ __forceinline__ void smallest_eigvec(const float* S6, float* n) {
    const float tr = S6[0] + S6[3] + S6[5];
    const float sc = (tr > 0.f) ? 1.0f / tr : 1.0f;
    float a00 = S6[0] * sc, a01 = S6[1] * sc, a02 = S6[2] * sc, a11 = S6[3] * sc, a12 = S6[4] * sc, a22 = S6[5] * sc;
    float v00 = 1.f, v01 = 0.f, v02 = 0.f, v10 = 0.f, v11 = 1.f, v12 = 0.f, v20 = 0.f, v21 = 0.f, v22 = 1.f;   // v[r][c], columns = eigenvectors
    // one Jacobi rotation in the (p, q) plane; r is the third index.  app, aqq, apq: the 2x2 block; arp, arq: the third row's entries
    auto rot = [](float& app, float& aqq, float& apq, float& arp, float& arq, float& v0p, float& v0q, float& v1p, float& v1q,
                  float& v2p, float& v2q) {
        if (fabsf(apq) < 1e-30f) return;
        const float theta = (aqq - app) / (2.0f * apq);
        const float t = ((theta >= 0.f) ? 1.0f : -1.0f) / (fabsf(theta) + sqrtf(theta * theta + 1.0f));
        const float c = 1.0f / sqrtf(t * t + 1.0f), sn = t * c;
        app -= t * apq; aqq += t * apq; apq = 0.f;
        const float rp = arp, rq = arq;
        arp = c * rp - sn * rq; arq = sn * rp + c * rq;
        float x;
        x = v0p; v0p = c * x - sn * v0q; v0q = sn * x + c * v0q;
        x = v1p; v1p = c * x - sn * v1q; v1q = sn * x + c * v1q;
        x = v2p; v2p = c * x - sn * v2q; v2q = sn * x + c * v2q;
    };
#pragma unroll 1
    for (int sweep = 0; sweep < 6; ++sweep) {
        rot(a00, a11, a01, a02, a12, v00, v01, v10, v11, v20, v21);
        rot(a00, a22, a02, a01, a12, v00, v02, v10, v12, v20, v22);
        rot(a11, a22, a12, a01, a02, v01, v02, v11, v12, v21, v22);
    }
    int k = 0; float m = a00;
    if (a11 < m) { m = a11; k = 1; }
    if (a22 < m) { m = a22; k = 2; }
    n[0] = (k == 0) ? v00 : ((k == 1) ? v01 : v02);
    n[1] = (k == 0) ? v10 : ((k == 1) ? v11 : v12);
    n[2] = (k == 0) ? v20 : ((k == 1) ? v21 : v22);
}

// COV: the world covariance is an input (cov6 = TexGSInputs.cov3D_precomp) -- a compile-time flavour so that the common path
// carries neither the branch nor the eigenvector code (K1 went 36 -> 64 us when it did)
template <bool COV = false>
__device__ __forceinline__ void geo_forward(Geo& g, const Frame& F, const CamConst& C, int i,
                                            const float* __restrict__ means, const float* __restrict__ scales,
                                            const float* __restrict__ rots, const float* __restrict__ juv,
                                            const float* __restrict__ cov6 = nullptr) {
    g.m[0] = means[3 * i + 0]; g.m[1] = means[3 * i + 1]; g.m[2] = means[3 * i + 2];
    const float mx = g.m[0], my = g.m[1], mz = g.m[2];
    g.t[0] = F.V[0] * mx + F.V[4] * my + F.V[8] * mz + F.V[12];
    g.t[1] = F.V[1] * mx + F.V[5] * my + F.V[9] * mz + F.V[13];
    g.t[2] = F.V[2] * mx + F.V[6] * my + F.V[10] * mz + F.V[14];
    g.valid = g.t[2] > TG_NEAR_Z;
    g.radius = 0;
    if (!g.valid) return;
    const float tx = g.t[0], ty = g.t[1], tz = g.t[2];

    g.hx = F.P[0] * mx + F.P[4] * my + F.P[8] * mz + F.P[12];
    g.hy = F.P[1] * mx + F.P[5] * my + F.P[9] * mz + F.P[13];
    g.hw = F.P[3] * mx + F.P[7] * my + F.P[11] * mz + F.P[15];
    g.pw = 1.0f / (g.hw + 1e-7f);
    const float ndcx = g.hx * g.pw, ndcy = g.hy * g.pw;
    g.xy[0] = ((ndcx + 1.0f) * (float)C.W - 1.0f) * 0.5f;
    g.xy[1] = ((ndcy + 1.0f) * (float)C.H - 1.0f) * 0.5f;

    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    if constexpr (COV) {
        // world-space covariance given (render/render.py:52-53): used as it is, scale_modifier not applied (lineage)
#pragma unroll
        for (int k = 0; k < 6; ++k) g.S[k] = cov6[6 * i + k];
#pragma unroll
        for (int k = 0; k < 9; ++k) { g.R[k] = 0.f; g.M[k] = 0.f; }
        g.q[0] = g.q[1] = g.q[2] = g.q[3] = 0.f; g.s[0] = g.s[1] = g.s[2] = 0.f;
    } else {
    // cov3D = (R diag(s)) (R diag(s))^T      (models/gaussian3d.py:17-21, utils/general.py:87-119)
    g.q[0] = rots[4 * i + 0]; g.q[1] = rots[4 * i + 1]; g.q[2] = rots[4 * i + 2]; g.q[3] = rots[4 * i + 3];
    const float r = g.q[0], x = g.q[1], y = g.q[2], z = g.q[3];
    g.R[0] = 1.0f - 2.0f * (y * y + z * z); g.R[1] = 2.0f * (x * y - r * z); g.R[2] = 2.0f * (x * z + r * y);
    g.R[3] = 2.0f * (x * y + r * z); g.R[4] = 1.0f - 2.0f * (x * x + z * z); g.R[5] = 2.0f * (y * z - r * x);
    g.R[6] = 2.0f * (x * z - r * y); g.R[7] = 2.0f * (y * z + r * x); g.R[8] = 1.0f - 2.0f * (x * x + y * y);
    s0 = scales[3 * i + 0]; s1 = scales[3 * i + 1]; s2 = scales[3 * i + 2];
    g.s[0] = C.scale_modifier * s0; g.s[1] = C.scale_modifier * s1; g.s[2] = C.scale_modifier * s2;
#pragma unroll
    for (int rr = 0; rr < 3; ++rr)
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) g.M[rr * 3 + cc] = g.R[rr * 3 + cc] * g.s[cc];
    const float* M = g.M;
    g.S[0] = M[0] * M[0] + M[1] * M[1] + M[2] * M[2];
    g.S[1] = M[0] * M[3] + M[1] * M[4] + M[2] * M[5];
    g.S[2] = M[0] * M[6] + M[1] * M[7] + M[2] * M[8];
    g.S[3] = M[3] * M[3] + M[4] * M[4] + M[5] * M[5];
    g.S[4] = M[3] * M[6] + M[4] * M[7] + M[5] * M[8];
    g.S[5] = M[6] * M[6] + M[7] * M[7] + M[8] * M[8];
    }

    // EWA cov2D = (J Wr) S (J Wr)^T + 0.3 I
    const float limx = TG_FRUSTUM_CLAMP * C.tanfovx, limy = TG_FRUSTUM_CLAMP * C.tanfovy;
    const float txtz = tx / tz, tytz = ty / tz;
    g.clx = (txtz < -limx) || (txtz > limx);
    g.cly = (tytz < -limy) || (tytz > limy);
    g.txc = fminf(limx, fmaxf(-limx, txtz)) * tz;
    g.tyc = fminf(limy, fmaxf(-limy, tytz)) * tz;
    if (!g.clx) g.txc = tx;
    if (!g.cly) g.tyc = ty;
    const float tz2 = tz * tz;
    g.J00 = C.fx / tz; g.J02 = -(C.fx * g.txc) / tz2;
    g.J11 = C.fy / tz; g.J12 = -(C.fy * g.tyc) / tz2;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        g.T0[k] = g.J00 * WR(F, 0, k) + g.J02 * WR(F, 2, k);
        g.T1[k] = g.J11 * WR(F, 1, k) + g.J12 * WR(F, 2, k);
    }
    const float* S = g.S;
    const float v00 = S[0] * g.T0[0] + S[1] * g.T0[1] + S[2] * g.T0[2];
    const float v01 = S[1] * g.T0[0] + S[3] * g.T0[1] + S[4] * g.T0[2];
    const float v02 = S[2] * g.T0[0] + S[4] * g.T0[1] + S[5] * g.T0[2];
    const float v10 = S[0] * g.T1[0] + S[1] * g.T1[1] + S[2] * g.T1[2];
    const float v11 = S[1] * g.T1[0] + S[3] * g.T1[1] + S[4] * g.T1[2];
    const float v12 = S[2] * g.T1[0] + S[4] * g.T1[1] + S[5] * g.T1[2];
    g.a = (g.T0[0] * v00 + g.T0[1] * v01 + g.T0[2] * v02) + TG_LOWPASS;
    g.b = g.T1[0] * v00 + g.T1[1] * v01 + g.T1[2] * v02;
    g.c = (g.T1[0] * v10 + g.T1[1] * v11 + g.T1[2] * v12) + TG_LOWPASS;
    g.det = g.a * g.c - g.b * g.b;
    if (g.det == 0.0f) { g.valid = false; return; }
    g.inv = 1.0f / g.det;
    g.conic[0] = g.c * g.inv; g.conic[1] = -g.b * g.inv; g.conic[2] = g.a * g.inv;
    const float mid = 0.5f * (g.a + g.c);
    const float lam = mid + sqrtf(fmaxf(0.1f, mid * mid - g.det));
    g.radius = (int)ceilf(3.0f * sqrtf(lam));

    // normal: shortest axis (first minimum), flipped to face the camera, world space
    g.dir[0] = mx - F.cam[0]; g.dir[1] = my - F.cam[1]; g.dir[2] = mz - F.cam[2];
    float n0, n1, n2;
    if constexpr (COV) {
        float ev[3];
        smallest_eigvec(g.S, ev);
        n0 = ev[0]; n1 = ev[1]; n2 = ev[2];
        g.kmin = -1;                                // no rotation-matrix column: K8 sends no normal gradient anywhere
    } else {
        g.kmin = 0; float smin = s0;
        if (s1 < smin) { smin = s1; g.kmin = 1; }
        if (s2 < smin) { smin = s2; g.kmin = 2; }
        n0 = (g.kmin == 0) ? g.R[0] : ((g.kmin == 1) ? g.R[1] : g.R[2]);      // selects, not a runtime index:
        n1 = (g.kmin == 0) ? g.R[3] : ((g.kmin == 1) ? g.R[4] : g.R[5]);      // keeps R[] in registers
        n2 = (g.kmin == 0) ? g.R[6] : ((g.kmin == 1) ? g.R[7] : g.R[8]);
    }
    g.sign = ((n0 * g.dir[0] + n1 * g.dir[1] + n2 * g.dir[2]) > 0.0f) ? -1.0f : 1.0f;
    g.n[0] = g.sign * n0; g.n[1] = g.sign * n1; g.n[2] = g.sign * n2;
    g.dlen = sqrtf(g.dir[0] * g.dir[0] + g.dir[1] * g.dir[1] + g.dir[2] * g.dir[2]);
    g.dir[0] /= g.dlen; g.dir[1] /= g.dlen; g.dir[2] /= g.dlen;

    // UV Taylor pre-fold: uv(p) = phi + G dp / (1 + g.dp), dp = pix - xy   
#pragma unroll
    for (int k = 0; k < 3; ++k)
        g.nv[k] = WR(F, k, 0) * g.n[0] + WR(F, k, 1) * g.n[1] + WR(F, k, 2) * g.n[2];
    g.sdot = g.nv[0] * tx + g.nv[1] * ty + g.nv[2] * tz;
    const float tn = sqrtf(tx * tx + ty * ty + tz * tz);
    g.degen = (juv == nullptr) || fabsf(g.sdot) <= TG_PLANE_EPS * tn;      // untextured surface: no UV plane at all
    // K = Jphi * R_c2w ; R_c2w[k][c] = Wr[c][k]
#pragma unroll
    for (int rr = 0; rr < 3; ++rr)
#pragma unroll
        for (int cc = 0; cc < 3; ++cc)
            g.K[rr * 3 + cc] = (juv == nullptr) ? 0.0f
                             : juv[9 * i + rr * 3 + 0] * WR(F, cc, 0) + juv[9 * i + rr * 3 + 1] * WR(F, cc, 1)
                             + juv[9 * i + rr * 3 + 2] * WR(F, cc, 2);
    if (g.degen) {
        g.gx = 0.0f; g.gy = 0.0f;
#pragma unroll
        for (int k = 0; k < 6; ++k) g.G[k] = 0.0f;
    } else {
        const float ax = tz * g.nv[0] / g.sdot, ay = tz * g.nv[1] / g.sdot;
        g.gx = ax / C.fx; g.gy = ay / C.fy;
        const float B00 = tz / C.fx - tx * g.gx, B01 = -tx * g.gy;
        const float B10 = -ty * g.gx,            B11 = tz / C.fy - ty * g.gy;
        const float B20 = -tz * g.gx,            B21 = -tz * g.gy;
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
            g.G[rr * 2 + 0] = g.K[rr * 3 + 0] * B00 + g.K[rr * 3 + 1] * B10 + g.K[rr * 3 + 2] * B20;
            g.G[rr * 2 + 1] = g.K[rr * 3 + 0] * B01 + g.K[rr * 3 + 1] * B11 + g.K[rr * 3 + 2] * B21;
        }
    }
}

__device__ __forceinline__ void tile_rect(const Geo& g, const CamConst& C, int& x0, int& y0, int& x1, int& y1) {
    const float rf = (float)g.radius;
    x0 = min(C.tiles_x, max(0, (int)((g.xy[0] - rf) / (float)TEXGS_TILE)));
    y0 = min(C.tiles_y, max(0, (int)((g.xy[1] - rf) / (float)TEXGS_TILE)));
    x1 = min(C.tiles_x, max(0, (int)((g.xy[0] + rf + (float)(TEXGS_TILE - 1)) / (float)TEXGS_TILE)));
    y1 = min(C.tiles_y, max(0, (int)((g.xy[1] + rf + (float)(TEXGS_TILE - 1)) / (float)TEXGS_TILE)));
}

// SH basis (bands 1..3) of utils/sh.py:74-100 for coefficient k = 1..15 -> b[k-1]
__device__ __forceinline__ void sh_basis(int deg, float x, float y, float z, float* b) {
    const float C1 = 0.4886025119029199f;
    b[0] = -C1 * y; b[1] = C1 * z; b[2] = -C1 * x;
    if (deg > 1) {
        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        b[3] = 1.0925484305920792f * xy;
        b[4] = -1.0925484305920792f * yz;
        b[5] = 0.31539156525252005f * (2.0f * zz - xx - yy);
        b[6] = -1.0925484305920792f * xz;
        b[7] = 0.5462742152960396f * (xx - yy);
        if (deg > 2) {
            b[8]  = -0.5900435899266435f * y * (3.0f * xx - yy);
            b[9]  = 2.890611442640554f * xy * z;
            b[10] = -0.4570457994644658f * y * (4.0f * zz - xx - yy);
            b[11] = 0.3731763325901154f * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
            b[12] = -0.4570457994644658f * x * (4.0f * zz - xx - yy);
            b[13] = 1.445305721320277f * z * (xx - yy);
            b[14] = -0.5900435899266435f * x * (xx - 3.0f * yy);
        }
    }
}

// d basis_k / d (x,y,z)
__device__ __forceinline__ void sh_basis_grad(int deg, float x, float y, float z, float* bx, float* by, float* bz) {
    const float C1 = 0.4886025119029199f;
    bx[0] = 0.f; by[0] = -C1; bz[0] = 0.f;
    bx[1] = 0.f; by[1] = 0.f; bz[1] = C1;
    bx[2] = -C1; by[2] = 0.f; bz[2] = 0.f;
    if (deg > 1) {
        const float c20 = 1.0925484305920792f, c22 = 0.31539156525252005f, c24 = 0.5462742152960396f;
        bx[3] = c20 * y;   by[3] = c20 * x;   bz[3] = 0.f;
        bx[4] = 0.f;       by[4] = -c20 * z;  bz[4] = -c20 * y;
        bx[5] = c22 * (-2.0f * x); by[5] = c22 * (-2.0f * y); bz[5] = c22 * 4.0f * z;
        bx[6] = -c20 * z;  by[6] = 0.f;       bz[6] = -c20 * x;
        bx[7] = c24 * 2.0f * x; by[7] = -c24 * 2.0f * y; bz[7] = 0.f;
        if (deg > 2) {
            const float xx = x * x, yy = y * y, zz = z * z;
            const float c30 = -0.5900435899266435f, c31 = 2.890611442640554f, c32 = -0.4570457994644658f,
                        c33 = 0.3731763325901154f, c35 = 1.445305721320277f;
            bx[8]  = c30 * 6.0f * x * y;            by[8]  = c30 * (3.0f * xx - 3.0f * yy);        bz[8]  = 0.f;
            bx[9]  = c31 * y * z;                   by[9]  = c31 * x * z;                          bz[9]  = c31 * x * y;
            bx[10] = c32 * (-2.0f * x * y);         by[10] = c32 * (4.0f * zz - xx - 3.0f * yy);   bz[10] = c32 * 8.0f * y * z;
            bx[11] = c33 * (-6.0f * x * z);         by[11] = c33 * (-6.0f * y * z);                bz[11] = c33 * (6.0f * zz - 3.0f * xx - 3.0f * yy);
            bx[12] = c32 * (4.0f * zz - 3.0f * xx - yy); by[12] = c32 * (-2.0f * x * y);           bz[12] = c32 * 8.0f * x * z;
            bx[13] = c35 * 2.0f * x * z;            by[13] = -c35 * 2.0f * y * z;                  bz[13] = c35 * (xx - yy);
            bx[14] = c30 * (3.0f * xx - 3.0f * yy); by[14] = c30 * (-6.0f * x * y);                bz[14] = 0.f;
        }
    }
}

__device__ __forceinline__ int sh_active(int deg, int K) {
    const int want = (deg + 1) * (deg + 1) - 1;
    return want < K ? want : K;
}

// The staged dL/dSH rows of a workgroup, LDS -> global (store, or add into a gradient buffer).  Four 16-byte accesses per
// thread are issued before the first is waited for: one element per iteration was 48 dependent read-modify-write round trips per
// wave (3K = 48) and 25 of K8's 116 us.  `g` must be 16-byte aligned for the vector path (checked by the caller, uniform).
template <int BLOCK, bool ADD>
__device__ __forceinline__ void sh_rows_out(float* __restrict__ g, const float* __restrict__ s, int count, bool vec) {
    const int tid = (int)threadIdx.x;
    int done = 0;
    if (vec) {
        const int n4 = count >> 2;
        float4* __restrict__ g4 = reinterpret_cast<float4*>(g);
        const float4* s4 = reinterpret_cast<const float4*>(s);
        for (int k = tid; k < n4; k += 4 * BLOCK) {
            float4 v[4], o[4];
            bool nz[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool in = k + u * BLOCK < n4;
                v[u] = in ? s4[k + u * BLOCK] : make_float4(0.f, 0.f, 0.f, 0.f);
                nz[u] = in && (!ADD || v[u].x != 0.f || v[u].y != 0.f || v[u].z != 0.f || v[u].w != 0.f);      // rows of culled Gaussians: nothing to add
            }
            if (ADD) {
#pragma unroll
                for (int u = 0; u < 4; ++u) if (nz[u]) o[u] = g4[k + u * BLOCK];
#pragma unroll
                for (int u = 0; u < 4; ++u) if (nz[u]) g4[k + u * BLOCK] = make_float4(o[u].x + v[u].x, o[u].y + v[u].y, o[u].z + v[u].z, o[u].w + v[u].w);
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) if (nz[u]) g4[k + u * BLOCK] = v[u];
            }
        }
        done = n4 << 2;
    }
    for (int k = done + tid; k < count; k += BLOCK) {
        const float v = s[k];
        if (ADD) { if (v != 0.f) g[k] += v; } else g[k] = v;
    }
}

// ------------------------------------------------------------------------------------------------ K1
template <bool COV>
__global__ void __launch_bounds__(TG_BLOCK)
k_preprocess_fwd(CamConst C, const float* __restrict__ vm, const float* __restrict__ pm, const float* __restrict__ cp,
                 const float* __restrict__ means, const float* __restrict__ shs, const float* __restrict__ opac,
                 const float* __restrict__ scales, const float* __restrict__ rots, const float* __restrict__ uvs,
                 const float* __restrict__ juv, const float* __restrict__ coff, const float* __restrict__ cov6,
                 float4* __restrict__ rec_test, float4* __restrict__ rec_shade, float* __restrict__ depth, int32_t* __restrict__ radii,
                 uint2* __restrict__ rect, uint32_t* __restrict__ tiles_touched, uint32_t* __restrict__ block_D,
                 uint32_t* __restrict__ zero_words, int num_zero_words) {
    extern __shared__ __attribute__((aligned(16))) float s_sh[];          // [256][3K] staged SH rows (coalesced load)
    __shared__ uint32_t s_tt[5 * (TG_BLOCK / 64)];
    const int i = blockIdx.x * TG_BLOCK + threadIdx.x;
    const bool live = i < C.N;
    for (int k = i; k < num_zero_words; k += (int)gridDim.x * TG_BLOCK) zero_words[k] = 0u;      // count tables of the depth sort (K2)
    const Frame F = load_frame(vm, pm, cp);
    Geo g;
    g.valid = false;
    if (live) geo_forward<COV>(g, F, C, i, means, scales, rots, juv, cov6);
    int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
    if (g.valid) {
        tile_rect(g, C, x0, y0, x1, y1);
        if ((x1 - x0) * (y1 - y0) == 0) g.valid = false;
    }
    const int na = (shs != nullptr && C.sh_degree > 0) ? sh_active(C.sh_degree, C.sh_coeffs) : 0;
    const int row = 3 * C.sh_coeffs;
    if (na > 0) {           // block-cooperative, contiguous read of this block's SH rows (180 B per Gaussian at K = 15)
        const size_t first = (size_t)blockIdx.x * TG_BLOCK * row;
        const size_t total = (size_t)C.N * row;
        const int count = (int)min((size_t)TG_BLOCK * row, total - first);
        for (int k = threadIdx.x; k < count; k += TG_BLOCK) s_sh[k] = shs[first + k];
        __syncthreads();
    }
    // Two things leave this kernel as one partial sum per workgroup (no atomics, nothing to zero; the host reads them back while
    // the depth sort runs and adds them up): D = sum of tiles_touched, and a 64-bit FINGERPRINT of what the tile binning, K6's
    // survivor lists and its per-bin footprint counts are functions of -- depth bits, tile rect, the whole test record and the
    // UV-Taylor part of the shading record of every Gaussian (view-dependent colour, depth and normal of the shading record do not
    // enter).  A second forward of the same geometry (models/texture_gaussian3d.py:375-389: same camera, sh_degree 0) finds the
    // same fingerprint and re-uses the first one's lists (texgs/rasterizer.py); two wrapping 32-bit sums of per-Gaussian hashes.
    uint32_t tt = 0u, ha = 0u, hb = 0u;
    if (live) {
        ha = (uint32_t)i * 0x9E3779B1u + 0x85EBCA77u;
        hb = ((uint32_t)i * 0xC2B2AE3Du) ^ 0x27D4EB2Fu;
    }
    auto mix = [&](uint32_t w) {
        ha = (ha ^ w) * 0x01000193u;
        hb = __builtin_rotateleft32(hb ^ w, 13) * 5u + 0xE6546B64u;
    };
    if (live && !g.valid) {
        radii[i] = 0; tiles_touched[i] = 0; rect[i] = make_uint2(0u, 0u);
        depth[i] = __uint_as_float(0xFFFFFFFFu);   // sort key of a culled Gaussian: after every visible one
        mix(0xFFFFFFFFu);                          // record left unwritten: never gathered (no instances)
    }
    if (live && g.valid) {
    // view-dependent colour: SH bands 1..deg at the (unit) view direction
    float vd[3] = {0.f, 0.f, 0.f};
    if (na > 0) {
        float b[15];
#pragma unroll
        for (int k = 0; k < 15; ++k) b[k] = 0.f;
        sh_basis(C.sh_degree, g.dir[0], g.dir[1], g.dir[2], b);
        const float* sp = s_sh + threadIdx.x * row;
#pragma unroll
        for (int k = 0; k < 15; ++k) {
            if (k < na) { vd[0] += b[k] * sp[3 * k + 0]; vd[1] += b[k] * sp[3 * k + 1]; vd[2] += b[k] * sp[3 * k + 2]; }
        }
    }
    if (coff) { vd[0] += coff[3 * i + 0]; vd[1] += coff[3 * i + 1]; vd[2] += coff[3 * i + 2]; }
    radii[i] = g.radius;
    tt = (uint32_t)((x1 - x0) * (y1 - y0));
    tiles_touched[i] = tt;
    depth[i] = g.t[2];
    const uint2 rc = make_uint2((uint32_t)x0 | ((uint32_t)y0 << 16), (uint32_t)x1 | ((uint32_t)y1 << 16));
    rect[i] = rc;
    // The record is split by who reads it (render.hip): the TEST part (32 B) is fetched for every (8x8 block, instance) pair,
    // the SHADING part (80 B) only for instances that survive the block cull.
    // conic pre-scaled for the blend kernels' falloff exponent: power = ah dx^2 + bh dx dy + ch dy^2 (render.hip gauss_power)
    // culling aids (never change a decision, only skip sure misses):
    //   thr   : alpha >= 1/255  <=>  power >= ln(1/(255*opacity)); lowered by a margin far above fp32 rounding
    //   rcull : pixel distance beyond which power < thr for every direction (largest eigenvalue of cov2D)
    const float op = opac[i];
    const float thr = (op > 0.0f) ? (-logf(255.0f * op) - 1e-3f) : 1.0f;          // > 0: nothing can pass
    const float cmid = 0.5f * (g.a + g.c);
    const float clam = cmid + sqrtf(fmaxf(0.1f, cmid * cmid - g.det));
    const float rcull = (thr < 0.0f) ? (sqrtf(-2.0f * thr * clam) * 1.001f + 0.01f) : -1.0f;
    const float4 t0 = make_float4(g.xy[0], g.xy[1], -0.5f * g.conic[0], -g.conic[1]);
    const float4 t1 = make_float4(-0.5f * g.conic[2], op, rcull, thr);
    const float4 s0 = make_float4(g.gx, g.gy, g.G[0], g.G[1]);
    const float4 s1 = make_float4(g.G[2], g.G[3], g.G[4], g.G[5]);
    const float4 s2 = uvs ? make_float4(uvs[3 * i + 0], uvs[3 * i + 1], uvs[3 * i + 2], vd[0]) : make_float4(0.f, 0.f, 1.f, vd[0]);
    float4* rt = rec_test + (size_t)i * (TEXGS_REC_TEST_FLOATS / 4);
    rt[0] = t0; rt[1] = t1;
    float4* rs = rec_shade + (size_t)i * (TEXGS_REC_SHADE_FLOATS / 4);
    rs[0] = s0; rs[1] = s1; rs[2] = s2;
    rs[3] = make_float4(vd[1], vd[2], g.t[2], g.n[0]);
    rs[4] = make_float4(g.n[1], g.n[2], g.xy[0], g.xy[1]);      // (xy again: K7's per-item gather of this record needs no second one)
    mix(__float_as_uint(g.t[2])); mix(rc.x); mix(rc.y);
    mix(__float_as_uint(t0.x)); mix(__float_as_uint(t0.y)); mix(__float_as_uint(t0.z)); mix(__float_as_uint(t0.w));
    mix(__float_as_uint(t1.x)); mix(__float_as_uint(t1.y)); mix(__float_as_uint(t1.z)); mix(__float_as_uint(t1.w));
    mix(__float_as_uint(s0.x)); mix(__float_as_uint(s0.y)); mix(__float_as_uint(s0.z)); mix(__float_as_uint(s0.w));
    mix(__float_as_uint(s1.x)); mix(__float_as_uint(s1.y)); mix(__float_as_uint(s1.z)); mix(__float_as_uint(s1.w));
    mix(__float_as_uint(s2.x)); mix(__float_as_uint(s2.y)); mix(__float_as_uint(s2.z));
    }
    // avalanche, then the workgroup's three sums and the (min, max) of its valid depth keys (the depth sort's bucket range)
    ha ^= ha >> 15; ha *= 0x2C1B3C6Du; ha ^= ha >> 12;
    hb ^= hb >> 16; hb *= 0x85EBCA6Bu; hb ^= hb >> 13;
    if (!live) { ha = 0u; hb = 0u; }
    uint32_t kmin = (live && g.valid) ? __float_as_uint(g.t[2]) : 0xFFFFFFFFu, kmax = (live && g.valid) ? __float_as_uint(g.t[2]) : 0u;
    for (int d = 32; d >= 1; d >>= 1) {
        tt += __shfl_xor(tt, d, 64); ha += __shfl_xor(ha, d, 64); hb += __shfl_xor(hb, d, 64);
        kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, d, 64)); kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, d, 64));
    }
    constexpr int NW = TG_BLOCK / 64;
    if ((threadIdx.x & 63) == 0) {
        const int w = threadIdx.x >> 6;
        s_tt[w] = tt; s_tt[NW + w] = ha; s_tt[2 * NW + w] = hb; s_tt[3 * NW + w] = kmin; s_tt[4 * NW + w] = kmax;
    }
    __syncthreads();
    if (threadIdx.x < 5) {
        uint32_t r = s_tt[threadIdx.x * NW];
        for (int w = 1; w < NW; ++w) {
            const uint32_t v = s_tt[threadIdx.x * NW + w];
            r = (threadIdx.x < 3) ? r + v : ((threadIdx.x == 3) ? min(r, v) : max(r, v));
        }
        block_D[threadIdx.x * gridDim.x + blockIdx.x] = r;  // five arrays of gridDim.x: tiles_touched, fingerprint lo, hi, min / max valid depth key
    }
}

// ------------------------------------------------------------------------------------------------ K8
#ifndef K8_BLOCK
#define K8_BLOCK 256
#endif
template <bool COV>
__global__ void __launch_bounds__(K8_BLOCK)
k_preprocess_bwd(CamConst C, const float* __restrict__ vm, const float* __restrict__ pm, const float* __restrict__ cp,
                 const float* __restrict__ means, const float* __restrict__ shs, const float* __restrict__ opac,
                 const float* __restrict__ scales,
                 const float* __restrict__ rots, const float* __restrict__ juv, const float* __restrict__ cov6,
                 const int32_t* __restrict__ radii, float* __restrict__ acc,
                 float* __restrict__ d_means, float* __restrict__ d_means2D, float* __restrict__ d_shs,
                 float* __restrict__ d_op, float* __restrict__ d_scales, float* __restrict__ d_rots,
                 float* __restrict__ d_uvs, float* __restrict__ d_coff, float* __restrict__ d_cov6, int accumulate) {
    extern __shared__ __attribute__((aligned(16))) float s_sh[];          // [256][3K]: SH rows in, dL/dSH rows out
    const int i = blockIdx.x * K8_BLOCK + threadIdx.x;
    const int K = C.sh_coeffs;
    const bool live = i < C.N;
    const int row = 3 * K;
    const int na = (shs != nullptr && C.sh_degree > 0) ? sh_active(C.sh_degree, K) : 0;
    const size_t first = (size_t)blockIdx.x * K8_BLOCK * row;
    const int count = d_shs ? (int)min((size_t)K8_BLOCK * row, (size_t)C.N * row - first) : 0;
    const bool vec = d_shs && (((size_t)d_shs) & 15) == 0;
    if (na > 0 && d_shs) {
        for (int k = threadIdx.x; k < count; k += K8_BLOCK) s_sh[k] = shs[first + k];      // plain loads: the compiler batches them
        __syncthreads();
    }
    const bool visible = live && radii[i] > 0;
    if (live && !visible) {          // culled: zero rows for the outputs that are written, nothing for those that are added into
        if (!(accumulate & TEXGS_ACC_MEANS3D)) d_means[3 * i] = d_means[3 * i + 1] = d_means[3 * i + 2] = 0.f;
        if (!(accumulate & TEXGS_ACC_MEANS2D)) d_means2D[3 * i] = d_means2D[3 * i + 1] = d_means2D[3 * i + 2] = 0.f;
        if (!(accumulate & TEXGS_ACC_OPACITIES)) d_op[i] = 0.f;
        if constexpr (COV) {
            if (!(accumulate & TEXGS_ACC_COV3D)) { for (int k = 0; k < 6; ++k) d_cov6[6 * i + k] = 0.f; }
        } else {
            if (!(accumulate & TEXGS_ACC_SCALES)) d_scales[3 * i] = d_scales[3 * i + 1] = d_scales[3 * i + 2] = 0.f;
            if (!(accumulate & TEXGS_ACC_ROTATIONS)) d_rots[4 * i] = d_rots[4 * i + 1] = d_rots[4 * i + 2] = d_rots[4 * i + 3] = 0.f;
        }
        if (d_uvs && !(accumulate & TEXGS_ACC_UVS)) d_uvs[3 * i] = d_uvs[3 * i + 1] = d_uvs[3 * i + 2] = 0.f;
        if (d_coff && !(accumulate & TEXGS_ACC_COLOR_OFFSET)) d_coff[3 * i] = d_coff[3 * i + 1] = d_coff[3 * i + 2] = 0.f;
    }
    if (visible) {
#define OUT(BIT, P, V) do { if (accumulate & (BIT)) (P) += (V); else (P) = (V); } while (0)
    const Frame F = load_frame(vm, pm, cp);
    Geo g;
    geo_forward<COV>(g, F, C, i, means, scales, rots, juv, cov6);
    // K7 left raw moment sums (common.h M_*) in this Gaussian's accumulator row; turn them into the gradients of the
    // record fields (R_* slots) here, where conic / opacity / G / g are in registers anyway, and hand the row back zeroed
    // (the scratch is all-zero between calls: no 38 MB memset per backward).
    float A[24];
    {
        float Mo[TEXGS_ACC_FLOATS];
        float4* ap = reinterpret_cast<float4*>(acc + (size_t)i * TEXGS_ACC_FLOATS);
#pragma unroll
        for (int k = 0; k < TEXGS_ACC_FLOATS / 4; ++k) {
            const float4 v = ap[k];
            Mo[4 * k] = v.x; Mo[4 * k + 1] = v.y; Mo[4 * k + 2] = v.z; Mo[4 * k + 3] = v.w;
            ap[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const float ca = g.conic[0], cb = g.conic[1], cc = g.conic[2];
        const float m00 = Mo[M_P], m10 = Mo[M_P + 1], m01 = Mo[M_P + 2];
        const float N00 = Mo[M_DN], N10 = Mo[M_DN + 3], N20 = Mo[M_DN + 6], D0 = Mo[M_DEN];
        A[R_XY]     = -(ca * m10 + cb * m01) - ((g.G[0] * N00 + g.G[2] * N10 + g.G[4] * N20) + g.gx * D0);
        A[R_XY + 1] = -(cc * m01 + cb * m10) - ((g.G[1] * N00 + g.G[3] * N10 + g.G[5] * N20) + g.gy * D0);
        A[R_CONIC] = -0.5f * Mo[M_P + 3]; A[R_CONIC + 1] = -Mo[M_P + 4]; A[R_CONIC + 2] = -0.5f * Mo[M_P + 5];
        const float op = opac[i];
        A[R_OP] = (op > 0.0f) ? m00 / op : 0.0f;                 // d alpha_raw / d opacity = alpha_raw / opacity
        A[R_G2] = Mo[M_DEN + 1]; A[R_G2 + 1] = Mo[M_DEN + 2];
#pragma unroll
        for (int cidx = 0; cidx < 3; ++cidx) {
            A[R_GM + 2 * cidx] = Mo[M_DN + 3 * cidx + 1]; A[R_GM + 2 * cidx + 1] = Mo[M_DN + 3 * cidx + 2];
            A[R_PHI + cidx] = Mo[M_PHI + cidx]; A[R_VD + cidx] = Mo[M_VD + cidx]; A[R_N + cidx] = Mo[M_N + cidx];
        }
        A[R_DEPTH] = Mo[M_DEPTH];
    }
    const float tx = g.t[0], ty = g.t[1], tz = g.t[2];
    float dt[3] = {0.f, 0.f, 0.f};      // dL/d t (view-space mean)
    float dm[3] = {0.f, 0.f, 0.f};      // dL/d mean (world)
    float dR[9];                        // dL/d R
#pragma unroll
    for (int k = 0; k < 9; ++k) dR[k] = 0.f;

    // (1,2) pass-through
    OUT(TEXGS_ACC_OPACITIES, d_op[i], A[R_OP]);
    if (d_uvs) { OUT(TEXGS_ACC_UVS, d_uvs[3 * i + 0], A[R_PHI]); OUT(TEXGS_ACC_UVS, d_uvs[3 * i + 1], A[R_PHI + 1]); OUT(TEXGS_ACC_UVS, d_uvs[3 * i + 2], A[R_PHI + 2]); }
    if (d_coff) { OUT(TEXGS_ACC_COLOR_OFFSET, d_coff[3 * i + 0], A[R_VD]); OUT(TEXGS_ACC_COLOR_OFFSET, d_coff[3 * i + 1], A[R_VD + 1]); OUT(TEXGS_ACC_COLOR_OFFSET, d_coff[3 * i + 2], A[R_VD + 2]); }

    // (3) conic -> cov2D (a,b,c)
    const float dA = A[R_CONIC], dB = A[R_CONIC + 1], dC = A[R_CONIC + 2];
    const float inv = g.inv, inv2 = inv * inv;
    const float da = dA * (-g.c * g.c * inv2) + dB * (g.b * g.c * inv2) + dC * (inv - g.a * g.c * inv2);
    const float db = dA * (2.0f * g.b * g.c * inv2) + dB * (-inv - 2.0f * g.b * g.b * inv2) + dC * (2.0f * g.a * g.b * inv2);
    const float dc = dA * (inv - g.a * g.c * inv2) + dB * (g.a * g.b * inv2) + dC * (-g.a * g.a * inv2);
    // (4) cov2D = T S T^T
    const float hb = 0.5f * db;
    const float* S = g.S;
    float TS0[3], TS1[3];
    TS0[0] = g.T0[0] * S[0] + g.T0[1] * S[1] + g.T0[2] * S[2];
    TS0[1] = g.T0[0] * S[1] + g.T0[1] * S[3] + g.T0[2] * S[4];
    TS0[2] = g.T0[0] * S[2] + g.T0[1] * S[4] + g.T0[2] * S[5];
    TS1[0] = g.T1[0] * S[0] + g.T1[1] * S[1] + g.T1[2] * S[2];
    TS1[1] = g.T1[0] * S[1] + g.T1[1] * S[3] + g.T1[2] * S[4];
    TS1[2] = g.T1[0] * S[2] + g.T1[1] * S[4] + g.T1[2] * S[5];
    float dT0[3], dT1[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        dT0[k] = 2.0f * (da * TS0[k] + hb * TS1[k]);
        dT1[k] = 2.0f * (hb * TS0[k] + dc * TS1[k]);
    }
    // dL/dS (symmetric 3x3, full matrix)
    float dS[9];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int l = 0; l < 3; ++l)
            dS[k * 3 + l] = g.T0[k] * g.T0[l] * da + (g.T0[k] * g.T1[l] + g.T1[k] * g.T0[l]) * hb + g.T1[k] * g.T1[l] * dc;
    // S = M M^T -> dM = 2 dS M
    float dM[9];
#pragma unroll
    for (int rr = 0; rr < 3; ++rr)
#pragma unroll
        for (int cc = 0; cc < 3; ++cc)
            dM[rr * 3 + cc] = 2.0f * (dS[rr * 3 + 0] * g.M[0 * 3 + cc] + dS[rr * 3 + 1] * g.M[1 * 3 + cc] + dS[rr * 3 + 2] * g.M[2 * 3 + cc]);
    float dscale[3];
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) {
        dscale[cc] = (dM[0 + cc] * g.R[0 + cc] + dM[3 + cc] * g.R[3 + cc] + dM[6 + cc] * g.R[6 + cc]) * C.scale_modifier;
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) dR[rr * 3 + cc] += dM[rr * 3 + cc] * g.s[cc];
    }
    // T = J Wr -> dJ
    const float dJ00 = dT0[0] * WR(F, 0, 0) + dT0[1] * WR(F, 0, 1) + dT0[2] * WR(F, 0, 2);
    const float dJ02 = dT0[0] * WR(F, 2, 0) + dT0[1] * WR(F, 2, 1) + dT0[2] * WR(F, 2, 2);
    const float dJ11 = dT1[0] * WR(F, 1, 0) + dT1[1] * WR(F, 1, 1) + dT1[2] * WR(F, 1, 2);
    const float dJ12 = dT1[0] * WR(F, 2, 0) + dT1[1] * WR(F, 2, 1) + dT1[2] * WR(F, 2, 2);
    const float tz2 = tz * tz, tz3 = tz2 * tz;
    dt[2] += -dJ00 * C.fx / tz2 - dJ11 * C.fy / tz2 + 2.0f * dJ02 * C.fx * g.txc / tz3 + 2.0f * dJ12 * C.fy * g.tyc / tz3;
    if (!g.clx) dt[0] += -dJ02 * C.fx / tz2;      // clamped: no gradient (lineage)
    if (!g.cly) dt[1] += -dJ12 * C.fy / tz2;

    // (6) mean2D: record slot is dL/d(pixel xy); operator returns dL/d(ndc xy)
    const float dndx = A[R_XY] * 0.5f * (float)C.W, dndy = A[R_XY + 1] * 0.5f * (float)C.H;
    OUT(TEXGS_ACC_MEANS2D, d_means2D[3 * i + 0], dndx); OUT(TEXGS_ACC_MEANS2D, d_means2D[3 * i + 1], dndy); if (!(accumulate & TEXGS_ACC_MEANS2D)) d_means2D[3 * i + 2] = 0.f;
    {
        const float dhx = dndx * g.pw, dhy = dndy * g.pw;
        const float dhw = -(dndx * g.hx + dndy * g.hy) * g.pw * g.pw;
#pragma unroll
        for (int k = 0; k < 3; ++k) dm[k] += dhx * F.P[k * 4 + 0] + dhy * F.P[k * 4 + 1] + dhw * F.P[k * 4 + 3];
    }
    // (7) depth
    dt[2] += A[R_DEPTH];
    // (9) normal (world) from the blend
    float dn[3] = {A[R_N], A[R_N + 1], A[R_N + 2]};

    // (11) UV Taylor pre-fold
    if (!g.degen) {
        const float* dG = &A[R_GM];
        float dBm[6];                            // dB[k][c] = sum_r K[r][k] dG[r][c]
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
                dBm[k * 2 + cc] = g.K[0 * 3 + k] * dG[0 * 2 + cc] + g.K[1 * 3 + k] * dG[1 * 2 + cc] + g.K[2 * 3 + k] * dG[2 * 2 + cc];
        float dtz = dBm[0] / C.fx + dBm[3] / C.fy;
        float dgt[2];
#pragma unroll
        for (int cc = 0; cc < 2; ++cc)
            dgt[cc] = A[R_G2 + cc] - (dBm[0 * 2 + cc] * tx + dBm[1 * 2 + cc] * ty + dBm[2 * 2 + cc] * tz);
#pragma unroll
        for (int k = 0; k < 3; ++k) dt[k] -= dBm[k * 2 + 0] * g.gx + dBm[k * 2 + 1] * g.gy;
        const float dax = dgt[0] / C.fx, day = dgt[1] / C.fy;
        const float dot_a_nv = dax * g.nv[0] + day * g.nv[1];
        dtz += dot_a_nv / g.sdot;
        float dnv[3] = {tz * dax / g.sdot, tz * day / g.sdot, 0.f};
        const float ds = -tz * dot_a_nv / (g.sdot * g.sdot);
        dnv[0] += ds * tx; dnv[1] += ds * ty; dnv[2] += ds * tz;
        dt[0] += ds * g.nv[0]; dt[1] += ds * g.nv[1]; dt[2] += ds * g.nv[2] + dtz;
#pragma unroll
        for (int k = 0; k < 3; ++k)
            dn[k] += dnv[0] * WR(F, 0, k) + dnv[1] * WR(F, 1, k) + dnv[2] * WR(F, 2, k);
    }
    // n = sign * R[:, kmin]
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float on = (g.kmin == k) ? g.sign : 0.0f;
        dR[0 + k] += on * dn[0]; dR[3 + k] += on * dn[1]; dR[6 + k] += on * dn[2];
    }

    // (10) view-dependent colour -> shs, view direction (rows staged in LDS; dL/dSH overwrites the row in place)
    if (d_shs && na > 0) {
        float b[15], bx[15], by[15], bz[15];
#pragma unroll
        for (int k = 0; k < 15; ++k) { b[k] = 0.f; bx[k] = 0.f; by[k] = 0.f; bz[k] = 0.f; }
        sh_basis(C.sh_degree, g.dir[0], g.dir[1], g.dir[2], b);
        sh_basis_grad(C.sh_degree, g.dir[0], g.dir[1], g.dir[2], bx, by, bz);
        float* sp = s_sh + threadIdx.x * row;
        const float v0 = A[R_VD], v1 = A[R_VD + 1], v2 = A[R_VD + 2];
        float ddx = 0.f, ddy = 0.f, ddz = 0.f;
#pragma unroll
        for (int k = 0; k < 15; ++k) {
            if (k < na) {
                const float w = sp[3 * k + 0] * v0 + sp[3 * k + 1] * v1 + sp[3 * k + 2] * v2;
                ddx += bx[k] * w; ddy += by[k] * w; ddz += bz[k] * w;
                sp[3 * k + 0] = b[k] * v0; sp[3 * k + 1] = b[k] * v1; sp[3 * k + 2] = b[k] * v2;
            }
        }
        for (int k = 3 * na; k < row; ++k) sp[k] = 0.f;
        // dir = d/|d|
        const float dd = g.dir[0] * ddx + g.dir[1] * ddy + g.dir[2] * ddz;
        dm[0] += (ddx - g.dir[0] * dd) / g.dlen;
        dm[1] += (ddy - g.dir[1] * dd) / g.dlen;
        dm[2] += (ddz - g.dir[2] * dd) / g.dlen;
    }

    // (8) t = [m,1] @ V
#pragma unroll
    for (int k = 0; k < 3; ++k) dm[k] += dt[0] * F.V[k * 4 + 0] + dt[1] * F.V[k * 4 + 1] + dt[2] * F.V[k * 4 + 2];
    OUT(TEXGS_ACC_MEANS3D, d_means[3 * i + 0], dm[0]); OUT(TEXGS_ACC_MEANS3D, d_means[3 * i + 1], dm[1]); OUT(TEXGS_ACC_MEANS3D, d_means[3 * i + 2], dm[2]);
    if constexpr (COV) {       // the given covariance: dL/dS, off-diagonal entries carrying both symmetric halves (lineage layout)
        OUT(TEXGS_ACC_COV3D, d_cov6[6 * i + 0], dS[0]); OUT(TEXGS_ACC_COV3D, d_cov6[6 * i + 1], dS[1] + dS[3]);
        OUT(TEXGS_ACC_COV3D, d_cov6[6 * i + 2], dS[2] + dS[6]); OUT(TEXGS_ACC_COV3D, d_cov6[6 * i + 3], dS[4]);
        OUT(TEXGS_ACC_COV3D, d_cov6[6 * i + 4], dS[5] + dS[7]); OUT(TEXGS_ACC_COV3D, d_cov6[6 * i + 5], dS[8]);
    }
    if constexpr (!COV) { OUT(TEXGS_ACC_SCALES, d_scales[3 * i + 0], dscale[0]); OUT(TEXGS_ACC_SCALES, d_scales[3 * i + 1], dscale[1]); OUT(TEXGS_ACC_SCALES, d_scales[3 * i + 2], dscale[2]); }

    // (5) R(q) -> q
    if constexpr (!COV) {
    const float r = g.q[0], x = g.q[1], y = g.q[2], z = g.q[3];
    OUT(TEXGS_ACC_ROTATIONS, d_rots[4 * i + 0], 2.0f * (-z * dR[1] + y * dR[2] + z * dR[3] - x * dR[5] - y * dR[6] + x * dR[7]));
    OUT(TEXGS_ACC_ROTATIONS, d_rots[4 * i + 1], 2.0f * (y * dR[1] + z * dR[2] + y * dR[3] - 2.0f * x * dR[4] - r * dR[5] + z * dR[6] + r * dR[7] - 2.0f * x * dR[8]));
    OUT(TEXGS_ACC_ROTATIONS, d_rots[4 * i + 2], 2.0f * (-2.0f * y * dR[0] + x * dR[1] + r * dR[2] + x * dR[3] + z * dR[5] - r * dR[6] + z * dR[7] - 2.0f * y * dR[8]));
    OUT(TEXGS_ACC_ROTATIONS, d_rots[4 * i + 3], 2.0f * (-2.0f * z * dR[0] - r * dR[1] + x * dR[2] + r * dR[3] - 2.0f * z * dR[4] + y * dR[5] + x * dR[6] + y * dR[7]));
    }
#undef OUT
    }   // visible
    if (d_shs) {            // dL/dSH: zero rows for culled Gaussians / inactive degree, then one coalesced block store
        if (na == 0 || (live && !visible)) {
            if (live) for (int k = 0; k < row; ++k) s_sh[threadIdx.x * row + k] = 0.f;
        }
        __syncthreads();
        if (accumulate & TEXGS_ACC_SHS) sh_rows_out<K8_BLOCK, true>(d_shs + first, s_sh, count, vec);
        else sh_rows_out<K8_BLOCK, false>(d_shs + first, s_sh, count, vec);
    }
}

// ------------------------------------------------------------------------------------------------ K9
__global__ void __launch_bounds__(TG_BLOCK)
k_mark_visible(int N, const float* __restrict__ vm, const float* __restrict__ means, uint8_t* __restrict__ visible) {
    const int i = blockIdx.x * TG_BLOCK + threadIdx.x;
    if (i >= N) return;
    const float tz = vm[2] * means[3 * i] + vm[6] * means[3 * i + 1] + vm[10] * means[3 * i + 2] + vm[14];
    visible[i] = tz > TG_NEAR_Z ? 1 : 0;
}

}  // namespace

void launch_preprocess_fwd(const CamConst& c, const TexGSFrame* f, const TexGSInputs* in, TexGSGeom* g, hipStream_t s) {
    if (c.N <= 0) return;
    const int blocks = (c.N + TG_BLOCK - 1) / TG_BLOCK;
    const size_t lds = (in->shs && c.sh_degree > 0) ? (size_t)TG_BLOCK * 3 * c.sh_coeffs * sizeof(float) : 0;
    int hdr_words = 0;
    uint32_t* hdr = bin_header_ptr(g, c.N, &hdr_words);
#define K1_LAUNCH(COV) hipLaunchKernelGGL(k_preprocess_fwd<COV>, dim3(blocks), dim3(TG_BLOCK), lds, s, c, f->viewmatrix, f->projmatrix, f->campos, \
                       in->means3D, in->shs, in->opacities, in->scales, in->rotations, in->uvs, in->gradient_uvs, in->color_offset, \
                       in->cov3D_precomp, reinterpret_cast<float4*>(g->rec_test), reinterpret_cast<float4*>(g->rec_shade), g->depth, g->radii, \
                       reinterpret_cast<uint2*>(g->rect), g->tiles_touched, bin_block_sums_ptr(g, c.N), hdr, hdr_words)
    if (in->cov3D_precomp) K1_LAUNCH(true); else K1_LAUNCH(false);
#undef K1_LAUNCH
}

void launch_preprocess_bwd(const CamConst& c, const TexGSFrame* f, const TexGSInputs* in, const TexGSGeom* g,
                           TexGSGrads* gr, hipStream_t s) {
    if (c.N <= 0) return;
    const int blocks = (c.N + K8_BLOCK - 1) / K8_BLOCK;
    const size_t lds = gr->dL_dshs ? (size_t)K8_BLOCK * 3 * c.sh_coeffs * sizeof(float) : 0;
#define K8_LAUNCH(COV) hipLaunchKernelGGL(k_preprocess_bwd<COV>, dim3(blocks), dim3(K8_BLOCK), lds, s, c, f->viewmatrix, f->projmatrix, f->campos, \
                       in->means3D, in->shs, in->opacities, in->scales, in->rotations, in->gradient_uvs, in->cov3D_precomp, \
                       g->radii, gr->acc, gr->dL_dmeans3D, gr->dL_dmeans2D, gr->dL_dshs, gr->dL_dopacities, gr->dL_dscales, \
                       gr->dL_drotations, gr->dL_duvs, gr->dL_dcolor_offset, gr->dL_dcov3D, gr->accumulate)
    if (in->cov3D_precomp) K8_LAUNCH(true); else K8_LAUNCH(false);
#undef K8_LAUNCH
}

void launch_mark_visible(const TexGSFrame* f, const float* means3D, uint8_t* visible, hipStream_t s) {
    if (f->num_gaussians <= 0) return;
    const int blocks = (f->num_gaussians + TG_BLOCK - 1) / TG_BLOCK;
    hipLaunchKernelGGL(k_mark_visible, dim3(blocks), dim3(TG_BLOCK), 0, s, f->num_gaussians, f->viewmatrix, means3D, visible);
}
