// K7 over the K6 -> K7 ITEM STREAM (k_render_bwd_stream) -- included by render.hip once, inside namespace k7_stream.
//
// K6 left, per 8x8 pixel block and in blend order, one item {T before the pair, alpha_raw, Gaussian id << 6 | pixel lane} for every
// CONTRIBUTING (pixel, Gaussian) pair (texgs.h TexGSImage.item_*).  This kernel walks the block's items back to front, 64 at a time
// (lane = item), and runs only the dense work of the backward:
//   stage B  as in the survivor-replay kernel (render_bwd_body.h): the Gaussian's shading record gathered per item (an L2 hit),
//            UV Taylor step, cubemap address, 4 dwordx3 taps, colour, s, dL/dcolour, dL/duv, dL/dden, the texture-gradient record;
//   stage C1 per pixel, back to front: dL/dalpha_i from the running sum of s_k alpha_k T_k behind i;
//   stage C2 per-Gaussian moment sums over TASKS of up to 16 consecutive items of one Gaussian, four per round.
// What is gone: the chunk phase, the lock-step test loop (stage A: 2.7 tests per contributing pair), the transmittance recurrence
// T /= (1 - alpha) (T is the forward's own value), the per-quadrant lists, and every partially filled stage-B round (a segment is 64
// items except the block's last one).  The structure stage C needs is DERIVED from the 64 keys of the segment with two ballots:
//   * a new pseudo-iteration starts where the pixel lane does not increase (K6 queued the items of one lock-step iteration in
//     lane order; iterations whose pixel sets happen to be ordered merge -- harmless, a pixel appears once per pseudo-iteration);
//   * a new task starts where (Gaussian, 4x4 quadrant) changes.
// Software pipeline: front(g) [gathers + taps issued] -> stage C of the PREVIOUS segment -> derive(g) -> back(g): the taps of a
// segment travel while the previous segment's sums are formed.
struct __attribute__((aligned(16))) StreamLds {
    float4 items[64 * 3 + 3];          // 3120: {T -> w, s -> dL/dpower, alpha_raw, key} {dc, du0} {du1, du2, inv, dden}; + one all-zero item
    float2 dxy[64 + 1];                //  520: splat centre - pixel of the item (stage C2); + zero
    float4 dpix[64];                   // 1024: dL/d(r, g, b, alpha) of the wave's pixels
    float4 dgeo[64];                   // 1024: dL/d(depth, normal)
    ull    itb[64];                    //  512: pixel-lane ballot of each pseudo-iteration of the segment
    uint32_t task[64];                 //  256: first item | items << 8
    uint32_t rbin[TG_RESV], tpos[TG_RESV], tend[TG_RESV];   // 768: the block's record-list reservations {bin, next free, end}
    uint8_t itf[64];                   //   64: first item of each pseudo-iteration
};                                     // 7288 B

template <bool TEX, bool UVG, bool TAPS>
__global__ void __launch_bounds__(64, K7S_WAVES_PER_SIMD)
k_render_bwd_stream(PixArgs a, TexBinArgs tb, const float* __restrict__ final_T,
                    const float* __restrict__ dL_dcolor, const float* __restrict__ dL_ddepth,
                    const float* __restrict__ dL_dnorm, const float* __restrict__ dL_dalpha,
                    float* __restrict__ acc, float* __restrict__ dtex) {
    static_assert(TAPS || (!TEX && !UVG), "no texture: no texture gradient and no UV chain");
    __shared__ StreamLds L;
    const int lane = (int)threadIdx.x;
    int tile, wave;
    if (!wave_block(a, tile, wave)) return;
    if (a.item_ctl[TEXGS_ITEM_CTL_FLAG] != 0u) return;      // K6 ran out of pages: the survivor-replay kernel does this view
    const uint2 tail = reinterpret_cast<const uint2*>(a.item_tail)[4 * tile + wave];
    const int n = (int)tail.y;
    if (n == 0) return;                                     // no contributing pair: nothing reserved, nothing to add
    const int tile_x = tile % a.tiles_x, tile_y = tile / a.tiles_x;
    const int wave_px = tile_x * TEXGS_TILE + ((wave & 1) << 3), wave_py = tile_y * TEXGS_TILE + ((wave >> 1) << 3);
    int ox, oy;
    lane_pixel(lane, ox, oy);
    const int px = wave_px + ox, py = wave_py + oy;
    const bool inside = (px < a.W) && (py < a.H);
    const int HW = a.W * a.H, pix = py * a.W + px;
    const float* __restrict__ tex = a.texture;

    float Tfin = 1.f;
    float dpix[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // dL/d (r,g,b,depth,nx,ny,nz,alpha)
    if (inside) {
        Tfin = nt_load(final_T + pix);
        if (dL_dcolor) { dpix[0] = nt_load(dL_dcolor + pix); dpix[1] = nt_load(dL_dcolor + HW + pix); dpix[2] = nt_load(dL_dcolor + 2 * HW + pix); }
        if (dL_ddepth) dpix[3] = nt_load(dL_ddepth + pix);
        if (dL_dnorm) { dpix[4] = nt_load(dL_dnorm + pix); dpix[5] = nt_load(dL_dnorm + HW + pix); dpix[6] = nt_load(dL_dnorm + 2 * HW + pix); }
        if (dL_dalpha) dpix[7] = nt_load(dL_dalpha + pix);
    }
    const float bgdot = a.bg[0] * dpix[0] + a.bg[1] * dpix[1] + a.bg[2] * dpix[2];
    if (lane < 3) L.items[64 * 3 + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane == 0) L.dxy[64] = make_float2(0.f, 0.f);
    L.dpix[lane] = make_float4(dpix[0], dpix[1], dpix[2], dpix[7]);
    L.dgeo[lane] = make_float4(dpix[3], dpix[4], dpix[5], dpix[6]);
    if (TEX && tb.rec != nullptr) {
        // image-wide bound on the texture-gradient records: the reduce kernel's fixed-point scale of THIS call (see render_bwd_body.h)
        const int mbits = wave_max_i(max(max(__float_as_int(fabsf(dpix[0])), __float_as_int(fabsf(dpix[1]))), __float_as_int(fabsf(dpix[2]))));
        if (lane == 0 && (uint32_t)mbits > __hip_atomic_load(tb.stats + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(tb.stats + 1, (uint32_t)mbits);
    }
    if constexpr (TEX) {
        // the block's reservations in the record lists (K6: {bin, offset inside the bin's list, count} per table entry)
        uint32_t b = TG_RESV_EMPTY, p0 = 0u, cnt = 0u;
        if (tb.rec != nullptr) {
            const uint32_t* __restrict__ rv = a.resv + (size_t)(4 * tile + wave) * (3 * TG_RESV);
            b = rv[lane];
            if (b != TG_RESV_EMPTY) { p0 = tb.base[b] + rv[TG_RESV + lane]; cnt = rv[2 * TG_RESV + lane]; }
        }
        L.rbin[lane] = b; L.tpos[lane] = p0; L.tend[lane] = p0 + cnt;
    }
    __builtin_amdgcn_wave_barrier();

    float behind = Tfin * bgdot;      // sum of s_k alpha_k T_k over the contributors BEHIND the current one, + the background term

    // the block's pages, walked backwards; a segment = the 64 items [64 g, 64 g + 64) of the stream lies inside one page
    int g = (n - 1) >> 6;
    int cur_pi = g >> 2;
    uint32_t cur_page = tail.x;
    uint32_t prev_page = a.item_link[cur_page];
    auto seg_ptr = [&](int gg) -> const uint32_t* {
        const uint32_t page = ((gg >> 2) == cur_pi) ? cur_page : prev_page;
        return a.item_pages + (size_t)page * (3 * TG_PAGE) + ((gg & 3) << 6) + lane;
    };
    uint32_t nT = 0u, nA = 0u, nK = 0u;                       // the NEXT segment's items, loaded one segment ahead
    {
        const uint32_t* __restrict__ sp = seg_ptr(g);
        if (64 * g + lane < n) { nT = nt_load(sp); nA = nt_load(sp + TG_PAGE); nK = nt_load(sp + 2 * TG_PAGE); }
    }

    struct Seg { int n_items, n_it, ntask; uint32_t it_lo, it_hi, it_first; };      // it_*: lane k = pixel ballot / first item of pseudo-iteration k
    struct Round {                                        // what the back half needs
        bool have;
        int pl, axis;
        uint32_t key;
        uint32_t slot, ovf0, ovf1;                        // record slot (see render_bwd_body.h)
        uint32_t fxw, fyw;
        uint32_t o00, dox, doy;
        float T, araw, w, vd0, vd1, vd2, nu0, nu1, nu2, inv;
        float fx, fy, ka, kb, kc, kd;
        float dx, dy;                                     // splat centre - pixel
        Texel3 t00, t01, t10, t11;
        float4 c5;                                        // depth, normal of the item's Gaussian
    };

    auto front = [&](Round& R, int cnt) {
        R.have = lane < cnt;
        R.key = R.have ? nK : 0u;
        R.T = R.have ? __uint_as_float(nT) : 1.f;
        R.araw = R.have ? __uint_as_float(nA) : 0.f;
        R.pl = (int)(R.key & 63u);
        R.w = fminf(TG_ALPHA_MAX, R.araw) * R.T;
        // the shading record of the item's Gaussian (lanes of one task read the same 80 bytes; a lane without an item reads Gaussian 0)
        const float4* __restrict__ sp = a.rec_shade + (TEXGS_REC_SHADE_FLOATS / 4) * (size_t)(R.key >> 6);
        float4 sd = make_float4(0.f, 0.f, 0.f, 0.f), se = sd;
        if constexpr (TAPS) { sd = sp[0]; se = sp[1]; }
        const float4 sf = sp[2];
        const float4 s3 = sp[3];
        const float4 s4 = sp[4];
        R.vd0 = sf.w; R.vd1 = s3.x; R.vd2 = s3.y;
        R.c5 = make_float4(s3.z, s3.w, s4.x, s4.y);
        int iox, ioy;
        lane_pixel(R.pl, iox, ioy);
        R.dx = s4.z - (float)(wave_px + iox); R.dy = s4.w - (float)(wave_py + ioy);
        if constexpr (TAPS) {
            const float dpx = -R.dx, dpy = -R.dy;
            const float den = 1.0f + sd.x * dpx + sd.y * dpy;
            R.inv = (den >= TG_DEN_MIN) ? __builtin_amdgcn_rcpf(den) : 0.0f;
            R.nu0 = sd.z * dpx + sd.w * dpy; R.nu1 = se.x * dpx + se.y * dpy; R.nu2 = se.z * dpx + se.w * dpy;
            const CubeTap ct = cube_address(sf.x + R.nu0 * R.inv, sf.y + R.nu1 * R.inv, sf.z + R.nu2 * R.inv, a.R);
            R.t00 = load_texel(tex, ct.o00); R.t01 = load_texel(tex, ct.o00 + ct.dox);
            R.t10 = load_texel(tex, ct.o00 + ct.doy); R.t11 = load_texel(tex, ct.o00 + ct.doy + ct.dox);
            R.fx = ct.fx; R.fy = ct.fy;
            if constexpr (UVG) {
                R.axis = ct.axis;
                R.ka = ct.su * ct.h; R.kb = ct.sv * ct.h;
                const float km = ct.h * ct.rma * ct.sm;
                R.kc = ct.sc * km; R.kd = ct.tc * km;
            }
            if constexpr (TEX) {
                R.o00 = ct.o00; R.dox = ct.dox; R.doy = ct.doy;
                {   // the record's first word; fyw = cell x | cell y << 5 | the high bits of fx18 / fy18 << 10 (rec_pack)
                    uint32_t hi;
                    R.fxw = rec_word0(ct.fx, ct.fy, hi);
                    R.fyw = (uint32_t)(ct.x0 & 31) | ((uint32_t)(ct.y0 & 31) << 5) | (hi << 10);
                }
                const bool binned = R.have && tb.rec != nullptr && tap_binned(ct);
                const uint32_t bin = tap_bin(ct, tb.nb);
                const int home = tap_home(ct);
                const bool hit = binned && L.rbin[home] == bin;
                R.slot = TG_SLOT_NONE; R.ovf0 = 0u; R.ovf1 = 0u;
                if (hit) {
                    const uint32_t pos = atomicAdd(&L.tpos[home], 1u);
                    if (pos < L.tend[home]) R.slot = pos;
                }
                ull pend = TG_BALLOT(binned) & ~TG_BALLOT(hit);
                while (pend != 0ull) {
                    const int l0 = __ffsll((long long)pend) - 1;
                    const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)bin, l0);
                    const ull m = pend & TG_BALLOT(bin == b0);
                    if ((m >> lane) & 1ull) R.slot = TG_SLOT_OVF | ((uint32_t)l0 << 8) | (uint32_t)mbcnt64(m);
                    if (lane == l0) { R.ovf0 = atomicAdd(tb.cursor + b0, (uint32_t)__popcll(m)); R.ovf1 = tb.base[b0 + 1u]; }
                    pend &= ~m;
                }
            }
        }
    };

    // pseudo-iterations and tasks of the segment from its 64 keys (see the file header); everything lands in LDS / lane registers
    auto derive = [&](const Round& R, int cnt) -> Seg {
        Seg sg;
        sg.n_items = cnt;
        const uint32_t key = R.key;
        const uint32_t kp = (uint32_t)__builtin_amdgcn_ds_bpermute(((lane - 1) & 63) << 2, (int)key);      // the item in front of this one
        const bool first = lane == 0;
        const bool is = R.have && (first || (key & 63u) <= (kp & 63u));
        const bool ts = R.have && (first || (key >> 4) != (kp >> 4));
        const ull S = TG_BALLOT(is), TS = TG_BALLOT(ts);
        sg.n_it = __popcll(S); sg.ntask = __popcll(TS);
        L.itb[lane] = 0ull;
        __builtin_amdgcn_wave_barrier();
        const int k_e = mbcnt64(S) + (is ? 1 : 0) - 1;          // pseudo-iteration of this item
        if (R.have) atomicOr(&L.itb[k_e], 1ull << (key & 63u));
        if (is) L.itf[k_e] = (uint8_t)lane;
        if (ts) {
            const ull above = TS & ~((2ull << lane) - 1ull);   // task starts behind this one (lane 63: the shift wraps to 0, nothing above)
            const int next = (above != 0ull) ? (__ffsll((long long)above) - 1) : cnt;
            L.task[mbcnt64(TS)] = (uint32_t)lane | ((uint32_t)(next - lane) << 8);
        }
        __builtin_amdgcn_wave_barrier();
        const ull bk = L.itb[lane];
        sg.it_lo = (uint32_t)bk; sg.it_hi = (uint32_t)(bk >> 32); sg.it_first = (uint32_t)L.itf[lane];
        return sg;
    };

    auto back = [&](Round& R) {
        float x0 = 0.f, x1 = 0.f, x2 = 0.f;
        if (R.have) {
            const float w = R.w;
            const float4 dp = L.dpix[R.pl];
            const float d0 = dp.x, d1 = dp.y, d2 = dp.z;
            float tv0 = 0.f, tv1 = 0.f, tv2 = 0.f, e0 = 0.f, e1 = 0.f, e2 = 0.f, f0 = 0.f, f1 = 0.f, f2 = 0.f;
            if constexpr (TAPS) {
                const float a0 = R.t01.x - R.t00.x, b0 = R.t10.x - R.t00.x, q0 = (R.t11.x - R.t10.x) - a0;
                const float a1 = R.t01.y - R.t00.y, b1 = R.t10.y - R.t00.y, q1 = (R.t11.y - R.t10.y) - a1;
                const float a2 = R.t01.z - R.t00.z, b2 = R.t10.z - R.t00.z, q2 = (R.t11.z - R.t10.z) - a2;
                e0 = __fmaf_rn(R.fy, q0, a0); e1 = __fmaf_rn(R.fy, q1, a1); e2 = __fmaf_rn(R.fy, q2, a2);
                f0 = __fmaf_rn(R.fx, q0, b0); f1 = __fmaf_rn(R.fx, q1, b1); f2 = __fmaf_rn(R.fx, q2, b2);
                tv0 = __fmaf_rn(R.fy, b0, __fmaf_rn(R.fx, e0, R.t00.x));
                tv1 = __fmaf_rn(R.fy, b1, __fmaf_rn(R.fx, e1, R.t00.y));
                tv2 = __fmaf_rn(R.fy, b2, __fmaf_rn(R.fx, e2, R.t00.z));
            }
            const float pre0 = TG_SH_C0 * tv0 + R.vd0 + 0.5f;
            const float pre1 = TG_SH_C0 * tv1 + R.vd1 + 0.5f;
            const float pre2 = TG_SH_C0 * tv2 + R.vd2 + 0.5f;
            const float dc0 = (pre0 > 0.f) ? w * d0 : 0.f, dc1 = (pre1 > 0.f) ? w * d1 : 0.f, dc2 = (pre2 > 0.f) ? w * d2 : 0.f;
            x0 = TG_SH_C0 * dc0; x1 = TG_SH_C0 * dc1; x2 = TG_SH_C0 * dc2;
            const float qv = fmaxf(0.f, pre0) * d0 + fmaxf(0.f, pre1) * d1 + fmaxf(0.f, pre2) * d2;
            const float4 c5 = R.c5;
            const float4 dg = L.dgeo[R.pl];
            const float s_ = qv + c5.x * dg.x + c5.y * dg.y + c5.z * dg.z + c5.w * dg.w + dp.w;
            L.items[lane * 3] = make_float4(R.T, s_, R.araw, __uint_as_float(R.key));
            L.dxy[lane] = make_float2(R.dx, R.dy);
            float du0 = 0.f;
            if constexpr (UVG) {
                const float dLdcol = x0 * e0 + x1 * e1 + x2 * e2;
                const float dLdrow = x0 * f0 + x1 * f1 + x2 * f2;
                const float dua = dLdcol * R.ka, dub = dLdrow * R.kb;
                const float dum = -(dLdcol * R.kc + dLdrow * R.kd);
                float du1, du2;
                if (R.axis == 0)      { du0 = dum; du2 = dua; du1 = dub; }
                else if (R.axis == 1) { du1 = dum; du0 = dua; du2 = dub; }
                else                  { du2 = dum; du0 = dua; du1 = dub; }
                const float dden = -(du0 * R.nu0 + du1 * R.nu1 + du2 * R.nu2) * R.inv * R.inv;   // inv = 0 when den < DEN_MIN
                L.items[lane * 3 + 2] = make_float4(du1, du2, R.inv, dden);
            }
            L.items[lane * 3 + 1] = make_float4(dc0, dc1, dc2, du0);
        }
        if constexpr (TEX) {
            uint32_t slot = R.slot;
            const bool ovf = (slot & TG_SLOT_OVF) != 0u && slot != TG_SLOT_NONE;
            if (TG_BALLOT(ovf) != 0ull) {                 // (wave-uniform: most rounds have no overflow footprint)
                const int ldr = (int)((slot >> 8) & 63u) << 2;
                const uint32_t p0 = (uint32_t)__builtin_amdgcn_ds_bpermute(ldr, (int)R.ovf0);
                const uint32_t p1 = (uint32_t)__builtin_amdgcn_ds_bpermute(ldr, (int)R.ovf1);
                if (ovf) { const uint32_t pos = p0 + (slot & 63u); slot = (pos < p1) ? pos : TG_SLOT_NONE; }
            }
            if (slot < tb.cap) {
                rec_store(tb.rec, slot, rec_pack(R.fxw, R.fyw >> 10, (int)(R.fyw & 31u), (int)((R.fyw >> 5) & 31u), x0, x1, x2));
            } else if (R.have && (x0 != 0.f || x1 != 0.f || x2 != 0.f)) {
                scatter_direct(dtex, R.o00, R.dox, R.doy, R.fx, R.fy, x0, x1, x2);
            }
        }
    };

    auto stage_c = [&](const Seg& sg) {
        // ================================================================ stage C1: per-pixel recurrence, BACK TO FRONT (the items of a
        // segment are in blend order: the last pseudo-iteration is the rearmost)
        for (int k = sg.n_it - 1; k >= 0; --k) {
            const uint32_t blo = (uint32_t)__builtin_amdgcn_readlane((int)sg.it_lo, k);
            const uint32_t bhi = (uint32_t)__builtin_amdgcn_readlane((int)sg.it_hi, k);
            const int it0 = __builtin_amdgcn_readlane((int)sg.it_first, k);
            if (((((ull)bhi << 32) | blo) >> lane) & 1ull) {
                const int it = it0 + (int)__builtin_amdgcn_mbcnt_hi(bhi, __builtin_amdgcn_mbcnt_lo(blo, 0u));
                const float4 i0 = L.items[it * 3];
                const float Ti = i0.x, s_i = i0.y, araw = i0.z;
                const float alpha = fminf(TG_ALPHA_MAX, araw);
                const float w = alpha * Ti;
                const float dL_dalpha_ = Ti * s_i - behind * __builtin_amdgcn_rcpf(1.0f - alpha);
                behind = __fmaf_rn(s_i, w, behind);
                *reinterpret_cast<float2*>(&L.items[it * 3]) = make_float2(w, araw * dL_dalpha_);      // {w, dL/dpower}
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ================================================================ stage C2: per-Gaussian moment sums, 16 lanes per task
        {
            const int sub = lane & 15;
            for (int q0 = 0; q0 < sg.ntask; q0 += 4) {
                const int qi = q0 + (lane >> 4);
                const bool live = qi < sg.ntask;
                const uint32_t task = live ? L.task[qi] : 0u;
                const int first = live ? (int)(task & 255u) : 64;
                const bool have = (uint32_t)sub < (task >> 8);
                const int item = have ? first + sub : 64;            // (64: the all-zero item)
                const uint32_t gid = __float_as_uint(L.items[first * 3].w) >> 6;
                float part[32];
                {
                    const float4 i0 = L.items[item * 3], i1 = L.items[item * 3 + 1];
                    const int pl = (int)(__float_as_uint(i0.w) & 63u);
                    const float w = i0.x, P = i0.y;
                    const float2 d = L.dxy[item];
                    const float dx = d.x, dy = d.y;                     // xy - pixel
                    const float Pdx = P * dx, Pdy = P * dy;
                    const float4 dg = L.dgeo[pl];
#pragma unroll
                    for (int k = 0; k < 32; ++k) part[k] = 0.f;
                    part[M_P] = P; part[M_P + 1] = Pdx; part[M_P + 2] = Pdy;
                    part[M_P + 3] = Pdx * dx; part[M_P + 4] = Pdx * dy; part[M_P + 5] = Pdy * dy;
                    if constexpr (UVG) {
                        const float4 i2 = L.items[item * 3 + 2];
                        const float du0 = i1.w, du1 = i2.x, du2 = i2.y, inv = i2.z, dden = i2.w;
                        const float dn0 = du0 * inv, dn1 = du1 * inv, dn2 = du2 * inv;
                        const float dpx = -dx, dpy = -dy;                       // pixel - xy
                        part[M_DEN] = dden; part[M_DEN + 1] = dden * dpx; part[M_DEN + 2] = dden * dpy;
                        part[M_DN + 0] = dn0; part[M_DN + 1] = dn0 * dpx; part[M_DN + 2] = dn0 * dpy;
                        part[M_DN + 3] = dn1; part[M_DN + 4] = dn1 * dpx; part[M_DN + 5] = dn1 * dpy;
                        part[M_DN + 6] = dn2; part[M_DN + 7] = dn2 * dpx; part[M_DN + 8] = dn2 * dpy;
                        part[M_PHI] = du0; part[M_PHI + 1] = du1; part[M_PHI + 2] = du2;
                    }
                    part[M_VD] = i1.x; part[M_VD + 1] = i1.y; part[M_VD + 2] = i1.z;
                    part[M_DEPTH] = w * dg.x;
                    part[M_N] = w * dg.y; part[M_N + 1] = w * dg.z; part[M_N + 2] = w * dg.w;
                }
                float lo, hi;
                static_assert(M_N + 3 == 28, "the live-slot masks cover slots 0..27");
                reduce32_rows16_masked<UVG ? TG_MOMENTS_ALL : TG_MOMENTS_NOUV>(part, lane, lo, hi);
                if (live) {
                    float* row = acc + (size_t)gid * TEXGS_ACC_FLOATS + transposed_index(sub);
                    if (lo != 0.f) unsafeAtomicAdd(row, lo);
                    if (hi != 0.f) unsafeAtomicAdd(row + 16, hi);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    };

    Seg prev;
    prev.n_items = 0; prev.n_it = 0; prev.ntask = 0; prev.it_lo = 0u; prev.it_hi = 0u; prev.it_first = 0u;
#ifdef K7S_TRACE
    // experiment builds only (scripts/k7s_trace.py): where a block's time goes, in core clocks.  Every stamp is preceded by a full
    // s_waitcnt so that the wait lands in the phase that caused it.
    const unsigned long long tr_t0 = wall_clock64();
    unsigned long long tr_acc[5] = {0ull, 0ull, 0ull, 0ull, 0ull};
#define TR_STAMP(K) do { __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = clock64(); \
                         tr_acc[K] += now_ - tr_last; tr_last = now_; __builtin_amdgcn_sched_barrier(0); } while (0)
    unsigned long long tr_last = clock64();
#else
#define TR_STAMP(K) do { } while (0)
#endif
    for (; g >= 0; --g) {
        // Issue priority by what the block still has to do (longest remaining stream first), as in the survivor-replay kernel
        // (a survivor is ~13 items: the thresholds are the replay kernel's 128 / 64 survivors)
        if (g > 26) __builtin_amdgcn_s_setprio(3); else if (g > 13) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);
        const int cnt = min(64, n - 64 * g);
        Round R;
        front(R, cnt);
        // the next segment's items (and, at a page boundary, the page before the one they are in)
        nT = 0u; nA = 0u; nK = 0u;
        if (g > 0) {
            const uint32_t* __restrict__ sp = seg_ptr(g - 1);
            nT = nt_load(sp); nA = nt_load(sp + TG_PAGE); nK = nt_load(sp + 2 * TG_PAGE);
            if (((g - 1) >> 2) != cur_pi) {            // entered the previous page: it becomes the current one, its link is loaded a page ahead
                cur_page = prev_page; --cur_pi;
                prev_page = (cur_pi > 0) ? a.item_link[cur_page] : TG_NOPAGE;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#ifdef K7S_TRACE
        {   // phase 0 = everything up to here WITHOUT the waits for memory (issue + the dependent record round trip inside front)
            __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = clock64(); tr_acc[0] += now_ - tr_last; tr_last = now_;
            __builtin_amdgcn_sched_barrier(0);
        }
#endif
        if (prev.n_items > 0) stage_c(prev);
        __builtin_amdgcn_sched_barrier(0);
        TR_STAMP(1);                                     // stage C of the previous segment + whatever memory wait it did not cover
        const Seg cur = derive(R, cnt);
        TR_STAMP(2);
        back(R);
        __builtin_amdgcn_wave_barrier();
        TR_STAMP(3);
        prev = cur;
    }
    stage_c(prev);
    TR_STAMP(4);
#ifdef K7S_TRACE
    if (lane == 0 && blockIdx.x < K7_TRACE_BLOCKS) {
        unsigned long long* tr = k7s_trace + 8 * (size_t)blockIdx.x;
        tr[0] = tr_t0; tr[1] = wall_clock64(); tr[2] = (unsigned long long)(uint32_t)n;
        tr[3] = tr_acc[0]; tr[4] = tr_acc[1]; tr[5] = tr_acc[2]; tr[6] = tr_acc[3]; tr[7] = tr_acc[4];
    }
#endif
    if constexpr (TEX) {
        // a reservation this block did not use up -- impossible while K6 and K7 agree on every footprint; should they ever not,
        // the reduce must not sum what an earlier call left in the unused slots
        __builtin_amdgcn_wave_barrier();
        const uint32_t q1 = min(L.tend[lane], tb.cap);
        for (uint32_t q = L.tpos[lane]; q < q1; ++q) { Rec4 z; z.a = 0u; z.b = 0u; z.c = 0u; z.d = 0u; reinterpret_cast<Rec4*>(tb.rec)[q] = z; }
    }
}
