// K7 (k_render_bwd) and its per-wave LDS layout -- included by render.hip once per CONFIGURATION, inside a namespace of its own,
// with BQ_CAP (items per segment), K7_GATHER (shading records from global memory instead of LDS planes) and K7_WAVES_PER_SIMD
// (the occupancy the register allocator is held to) defined by the includer.  No include guard on purpose.
#if K7_GATHER
struct PlanesT { float4 A[65]; float4 B[65]; };      // test planes only (stage A, and the splat centre for stages B / C2)
#else
typedef Planes PlanesT;
#endif
struct __attribute__((aligned(16))) BwdLds {      // bytes: configuration lds (BQ_CAP 128, planes in LDS) / occ (BQ_CAP 64, K7_GATHER)
    float4 items[BQ_CAP * 3 + 3];       // 6192 / 3120: 3 float4 per item {T -> w, s -> dL/dpower, alpha_raw, key} {dc, du0} {du1, du2, inv, dden}; + one all-zero item
    float4 abuf[BQ_CAP];                // 2048 / 1024: {T, -, alpha_raw, key} of the NEXT segment's items (stage A runs one segment ahead of B / C)
    PlanesT p;                          // 6768 / 2080
    float4 dpix[64];                    // 1024: dL/d(r, g, b, alpha) of the wave's pixels
    float4 dgeo[64];                    // 1024: dL/d(depth, normal)
    uint32_t task[64];                  //  256
    uint8_t list[4][64];                //  256
    uint32_t tpos[TG_RESV], tend[TG_RESV];   // 512: the block's reservations: next free record (absolute), end of the range (the bins: p.B[].w)
};                                      // 18080 B -> 9 waves per CU / 9296 B -> 16 waves per CU (17 by LDS; the registers are held to 4 per SIMD)

// What a caller wants decides what is compiled in (TexGSGrads.want, texgs.h):
//   TEX   the texture gradient (records / atomics).              false: frozen texture, or the untextured surface
//   GEO   the per-Gaussian moments (stages C1, C2, K8 after it).  false: only the texture is trained (every Gaussian input frozen)
//   UVG   the UV chain (dL/duv, dL/dden -> M_DEN, M_DN, M_PHI).   false: nothing upstream of uv wants a gradient, or no texture
//   TAPS  the texture is sampled at all.                          false: untextured surface (texture == NULL)
template <bool TEX, bool GEO, bool UVG, bool TAPS>
__global__ void __launch_bounds__(64, K7_WAVES_PER_SIMD)
k_render_bwd(PixArgs a, TexBinArgs tb, const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
             const float* __restrict__ dL_dcolor, const float* __restrict__ dL_ddepth,
             const float* __restrict__ dL_dnorm, const float* __restrict__ dL_dalpha,
             float* __restrict__ acc, float* __restrict__ dtex) {
    static_assert(TAPS || (!TEX && !UVG), "no texture: no texture gradient and no UV chain");
    static_assert(GEO || !UVG, "the UV chain ends in the per-Gaussian moments");
    static_assert(TEX || GEO, "nothing to compute");
    __shared__ BwdLds L;
    const int lane = (int)threadIdx.x;
    int tile, wave;
    if (!wave_block(a, tile, wave)) return;
    if (a.run_if != nullptr && *a.run_if == 0u) return;      // launched as the item-stream kernel's fallback, and K6 did not run out of pages
#ifdef K7_TRACE
    const unsigned long long trace_t0 = wall_clock64();   // experiment builds only (scripts/k7_trace.py): when each block ran, and where
#endif
    const int tile_x = tile % a.tiles_x, tile_y = tile / a.tiles_x;
    const int wave_px = tile_x * TEXGS_TILE + ((wave & 1) << 3), wave_py = tile_y * TEXGS_TILE + ((wave >> 1) << 3);
    int ox, oy;
    lane_pixel(lane, ox, oy);
    const int px = wave_px + ox, py = wave_py + oy;
    const bool inside = (px < a.W) && (py < a.H);
    const float pxf = (float)px, pyf = (float)py;
    const uint2 range = a.ranges[tile];
    const int todo = (int)(range.y - range.x);
    const int HW = a.W * a.H, pix = py * a.W + px;
    const float* __restrict__ tex = a.texture;
    const uint32_t keybase = ((uint32_t)lane << 8) | ((uint32_t)ox << 14) | ((uint32_t)oy << 17);
    const uint8_t* mylist = L.list[lane >> 4];

    float Tfin = 1.f; int last = 0;
    float dpix[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // dL/d (r,g,b,depth,nx,ny,nz,alpha)
    if (inside) {
        Tfin = nt_load(final_T + pix); last = (int)nt_load(n_contrib + pix);
        if (dL_dcolor) { dpix[0] = nt_load(dL_dcolor + pix); dpix[1] = nt_load(dL_dcolor + HW + pix); dpix[2] = nt_load(dL_dcolor + 2 * HW + pix); }
        if (dL_ddepth) dpix[3] = nt_load(dL_ddepth + pix);
        if (dL_dnorm) { dpix[4] = nt_load(dL_dnorm + pix); dpix[5] = nt_load(dL_dnorm + HW + pix); dpix[6] = nt_load(dL_dnorm + 2 * HW + pix); }
        if (dL_dalpha) dpix[7] = nt_load(dL_dalpha + pix);
    }
    const float bgdot = a.bg[0] * dpix[0] + a.bg[1] * dpix[1] + a.bg[2] * dpix[2];
    if (lane < 3) L.items[BQ_CAP * 3 + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
    L.dpix[lane] = make_float4(dpix[0], dpix[1], dpix[2], dpix[7]);
    L.dgeo[lane] = make_float4(dpix[3], dpix[4], dpix[5], dpix[6]);
#if K7_GATHER
    if (lane == 0) { L.p.A[TG_DUMMY] = make_float4(0.f, 0.f, 0.f, 0.f); L.p.B[TG_DUMMY] = make_float4(0.f, 0.f, __uint_as_float(0xFFFFFFFFu), 0.f); }
#else
    init_dummy(L.p, lane);
#endif
    if (TEX && tb.rec != nullptr) {
        // image-wide bound on the texture-gradient records (|x| <= C0 |dL/dpixel|): the reduce kernel's fixed-point scale of THIS
        // call (k_bin_offsets, launched just before this kernel, cleared the word).  Non-negative floats order like their bit
        // patterns, and so do +inf (0x7F800000) and NaN (above it): an INTEGER maximum carries a non-finite upstream gradient to the
        // reduce, which then adds the records with float atomics instead of fixed point (fmaxf would have dropped the NaN).  The
        // plain read first keeps 10^4 waves off one hot word.
        const int mbits = wave_max_i(max(max(__float_as_int(fabsf(dpix[0])), __float_as_int(fabsf(dpix[1]))), __float_as_int(fabsf(dpix[2]))));
        if (lane == 0 && (uint32_t)mbits > __hip_atomic_load(tb.stats + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(tb.stats + 1, (uint32_t)mbits);
    }
    if constexpr (TEX) {
        // the block's reservations (K6 left {bin, offset inside the bin's list, count} per table entry; k_bin_offsets has since
        // scanned the totals); lane = table entry
        uint32_t b = TG_RESV_EMPTY, p0 = 0u, n = 0u;
        if (tb.rec != nullptr) {
            const uint32_t* __restrict__ rv = a.resv + (size_t)(4 * tile + wave) * (3 * TG_RESV);
            b = nt_load(rv + lane);
            if (b != TG_RESV_EMPTY) { p0 = tb.base[b] + nt_load(rv + TG_RESV + lane); n = nt_load(rv + 2 * TG_RESV + lane); }
        }
        resv_bin(L.p, lane) = b; L.tpos[lane] = p0; L.tend[lane] = p0 + n;
    }
    // last contributor of each quadrant (row maximum) and of the block
    int rl = last;
    rl = max(rl, __shfl_xor(rl, 1, 64)); rl = max(rl, __shfl_xor(rl, 2, 64));
    rl = max(rl, __shfl_xor(rl, 4, 64)); rl = max(rl, __shfl_xor(rl, 8, 64));
    int rowlast[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) rowlast[q] = min(__builtin_amdgcn_readlane(rl, 16 * q), todo);
    const int wave_last = max(max(rowlast[0], rowlast[1]), max(rowlast[2], rowlast[3]));
    __builtin_amdgcn_wave_barrier();

    float T = Tfin;
    float behind = Tfin * bgdot;      // sum of s_k alpha_k T_k over the contributors BEHIND the current one, + the background term

    // The survivors of this block's cull come from K6 (it culled exactly this list, at least as far as the last contributor):
    // chunks of 64 from the back, lane = survivor in descending list position.  The next chunk's entries are loaded one chunk ahead.
    const size_t sbase = 4 * (size_t)range.x + (size_t)wave * (size_t)todo;
    const int ns = min((int)a.surv_cnt[4 * tile + wave], todo);      // (a count beyond the block's region can only be a stale buffer)
    uint2 nsv = make_uint2(0u, 0xFFFFFFFFu);
    uint32_t nqm = 0u;
    if (ns - 1 - lane >= 0) { nsv = nt_load2(a.surv + sbase + (ns - 1 - lane)); nqm = nt_load(a.surv_qm + sbase + (ns - 1 - lane)); }
    // ---- stage B, as two halves per round of 64 items (see the kernel header): FRONT issues everything that goes to memory,
    // BACK consumes it.  A segment has at most two rounds: both fronts first (8 tap loads + 2 returning atomics in flight), then
    // both backs.  (Starting round 0's front inside stage A, as soon as 64 items exist, was measured: K7 732 -> 771 us.)
    struct Seg { int n_items, n_it; uint32_t it_lo, it_hi, it_first; };      // it_*: lane k = ballot / first item of the k-th productive iteration
    struct Round {                                        // what the back half needs, as few registers as possible
        bool have;
        int e, pl, jj, axis;
        uint32_t key;
        uint32_t slot;                                    // the item's record slot (absolute); TG_SLOT_NONE: none (not binned / no room);
                                                          // TG_SLOT_OVF | leader << 8 | rank: an overflow footprint (see front_b)
        uint32_t ovf0, ovf1;                              // group leaders of overflow footprints: first slot of the group, end of the list
        uint32_t fxw, fyw;                                // the record's first word; cell + high fraction bits (rec_word0 / rec_pack)
        uint32_t o00, dox, doy;                           // tap byte offsets: o01 = o00 + dox, o10 = o00 + doy, o11 = o00 + dox + doy
        float w, vd0, vd1, vd2, nu0, nu1, nu2, inv;
        float fx, fy, ka, kb, kc, kd;                     // d(col,row)/d(ua,ub,m) factors of the cube projection
        Texel3 t00, t01, t10, t11;
#if K7_GATHER
        float4 sd, se, sf;                                // g, G (8 floats), phi + vd0: between the two parts of the front half only
        float4 c5;                                        // depth, normal of the item's Gaussian
#endif
    };
    uint32_t cid = 0u;                                    // lane = survivor of the current chunk: its Gaussian id
    (void)cid;
    // front half, part A: the item, and (K7_GATHER) the loads of its Gaussian's shading record.  Part A of EVERY round of the segment
    // runs before any part B: vmcnt retires in order, a record load issued behind another round's taps would wait for those taps.
    auto front_a = [&](int rbase, Round& R, int n_items) {
        R.e = rbase + lane;
        R.have = R.e < n_items;
        float4 it = make_float4(1.f, 0.f, 0.f, 0.f);
        if (R.have) {
            it = L.abuf[R.e];
            if constexpr (GEO) L.items[R.e * 3] = it;         // (the previous segment's stage C has read its items by now)
        }
        R.key = __float_as_uint(it.w);
        R.pl = KEY_PL(R.key);
        R.jj = KEY_J(R.key);
        R.w = fminf(TG_ALPHA_MAX, it.z) * it.x;
#if K7_GATHER
        // the shading record of the item's Gaussian from global memory (an L2 hit; lanes of one task read the same 80 bytes)
        const float4* __restrict__ sp = a.rec_shade + (TEXGS_REC_SHADE_FLOATS / 4) * (size_t)(uint32_t)__builtin_amdgcn_ds_bpermute(R.jj << 2, (int)cid);
        if constexpr (TAPS) { R.sd = sp[0]; R.se = sp[1]; }
        R.sf = sp[2];
        const float4 s3 = sp[3];
        const float2 s4 = *reinterpret_cast<const float2*>(sp + 4);
        R.vd1 = s3.x; R.vd2 = s3.y;
        R.c5 = make_float4(s3.z, s3.w, s4.x, s4.y);
#endif
    };
    auto front_b = [&](Round& R) {
        const uint32_t key = R.key;
        const int jj = R.jj;
#if K7_GATHER
        const float4 f_ = R.sf;
        R.vd0 = f_.w;
#else
        const float4 f_ = L.p.F[jj];
        const float2 g2 = L.p.G[jj];
        R.vd0 = f_.w; R.vd1 = g2.x; R.vd2 = g2.y;
#endif
        if constexpr (TAPS) {
            // UV Taylor step, cubemap address, tap loads
            const float2 xy = *reinterpret_cast<const float2*>(&L.p.A[jj]);
#if K7_GATHER
            const float4 d_ = R.sd, e4 = R.se;
#else
            const float4 d_ = L.p.D[jj], e4 = L.p.E[jj];
#endif
            const float dpx = (float)(wave_px + KEY_OX(key)) - xy.x, dpy = (float)(wave_py + KEY_OY(key)) - xy.y;
            const float den = 1.0f + d_.x * dpx + d_.y * dpy;
            R.inv = (den >= TG_DEN_MIN) ? __builtin_amdgcn_rcpf(den) : 0.0f;
            R.nu0 = d_.z * dpx + d_.w * dpy; R.nu1 = e4.x * dpx + e4.y * dpy; R.nu2 = e4.z * dpx + e4.w * dpy;
            const CubeTap ct = cube_address(f_.x + R.nu0 * R.inv, f_.y + R.nu1 * R.inv, f_.z + R.nu2 * R.inv, a.R);
            // every lane loads (a lane without an item decoded key 0 -> survivor slot 0, pixel (0, 0) of the block: a valid address)
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
                // (rounded to 18 mantissa bits, not truncated: no bias; fx in [0, 1) may round up to exactly 1)
                {   // the record's first word; fyw = cell x | cell y << 5 | the high bits of fx18 / fy18 << 10 (rec_pack)
                    uint32_t hi;
                    R.fxw = rec_word0(ct.fx, ct.fy, hi);
                    R.fyw = (uint32_t)(ct.x0 & 31) | ((uint32_t)(ct.y0 & 31) << 5) | (hi << 10);
                }
                // slot in the texture bin's record list: from the block's reservation of that bin (K6 counted exactly these
                // footprints), one returning LDS atomic per lane.  A footprint whose table entry belongs to another bin (2 %) goes
                // behind the reserved part of the list: lanes grouped by bin, one returning GLOBAL atomic per group on the bin's
                // overflow cursor, resolved in the back half (the round-4 path).
                const bool binned = R.have && tb.rec != nullptr && tap_binned(ct);
                const uint32_t bin = tap_bin(ct, tb.nb);
                const int home = tap_home(ct);
                const bool hit = binned && resv_bin(L.p, home) == bin;
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
    auto back = [&](Round& R) {
        float x0 = 0.f, x1 = 0.f, x2 = 0.f;
        if (R.have) {
            const float w = R.w;
            const float4 dp = L.dpix[R.pl];
            const float d0 = dp.x, d1 = dp.y, d2 = dp.z;
            // bilinear sample and its two derivatives in the nested form c = t00 + fx e + fy b, e = a + fy d = dc/dfx,
            // f = b + fx d = dc/dfy with a = t01 - t00, b = t10 - t00, d = (t11 - t10) - a: 8 operations per channel for all
            // three (the weighted-sum form took 17)
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
            // colour -> view-dependent term and texture
            const float dc0 = (pre0 > 0.f) ? w * d0 : 0.f, dc1 = (pre1 > 0.f) ? w * d1 : 0.f, dc2 = (pre2 > 0.f) ? w * d2 : 0.f;
            x0 = TG_SH_C0 * dc0; x1 = TG_SH_C0 * dc1; x2 = TG_SH_C0 * dc2;
            if constexpr (GEO) {
                const float qv = fmaxf(0.f, pre0) * d0 + fmaxf(0.f, pre1) * d1 + fmaxf(0.f, pre2) * d2;
                // s = colour . dL/dcolour + (depth, normal) . dL/d(depth, normal) + dL/dalpha: everything stage C1's
                // recurrence needs from this pair, formed here where all 64 lanes work
#if K7_GATHER
                const float4 c5 = R.c5;
#else
                const float4 c5 = L.p.C[R.jj];
#endif
                const float4 dg = L.dgeo[R.pl];
                L.items[R.e * 3].y = qv + c5.x * dg.x + c5.y * dg.y + c5.z * dg.z + c5.w * dg.w + dp.w;
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
                    L.items[R.e * 3 + 2] = make_float4(du1, du2, R.inv, dden);
                }
                L.items[R.e * 3 + 1] = make_float4(dc0, dc1, dc2, du0);
            }
        }
        if constexpr (TEX) {
            // texture gradient of this pair: append the record, or straight to dL_dtexture when the footprint is clamped at
            // a face border / does not fit the buffer (still correct, just slow)
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
    for (int hi = ns; hi > 0; hi -= 64) {
        // Issue priority by what the block still has to do -- longest REMAINING list first.  The launch is 10 000 one-wave blocks on
        // 4 096 wave slots and a block is long (median a third of the kernel, profiles/r05_k7_block_trace.json): with equal shares
        // the longest blocks end the kernel alone.  K7 645 -> 626 us (661 -> 646 on another box); priority by the block's TOTAL
        // length, four levels instead of three, the same in K6: no change (profiles/r05_ablation.md).
        if (hi > 128) __builtin_amdgcn_s_setprio(3); else if (hi > 64) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);
        const bool live = hi - 1 - lane >= 0;
        const uint32_t id = nsv.x, pos = nsv.y, qm = nqm;
        nsv = make_uint2(0u, 0xFFFFFFFFu); nqm = 0u;
        if (hi - 65 - lane >= 0) { nsv = nt_load2(a.surv + sbase + (hi - 65 - lane)); nqm = nt_load(a.surv_qm + sbase + (hi - 65 - lane)); }
        // survivors behind the block's last contributor (K6 tested them, nothing blended): skip whole chunks of them
        const int take = min(64, hi);
        if ((int)__builtin_amdgcn_readlane((int)pos, take - 1) >= wave_last) continue;
        __builtin_amdgcn_wave_barrier();
        cid = id;
#if K7_GATHER
        {   // lane = survivor: only the test record goes to LDS (a dead lane: alpha 0, position beyond every list)
            float4 T0 = make_float4(0.f, 0.f, 0.f, 0.f), T1 = make_float4(0.f, 0.f, -1.f, 1.f);
            if (live) { const float4* __restrict__ tp = a.rec_test + 2 * (size_t)id; T0 = tp[0]; T1 = tp[1]; }
            L.p.A[lane] = T0;
            float* pb = reinterpret_cast<float*>(&L.p.B[lane]); pb[0] = T1.x; pb[1] = T1.y; pb[2] = __uint_as_float(pos);
        }
#else
        float4 T0, T1;
        load_chunk(a, L.p, lane, live, id, pos, T0, T1);
#endif
        reinterpret_cast<uint32_t*>(&L.list[0][0])[lane] = 0x40404040u;
        __builtin_amdgcn_wave_barrier();
        int len[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool rq = live && ((int)pos < rowlast[q]) && ((qm >> q) & 1u);
            const ull m = TG_BALLOT(rq);
            len[q] = __popcll(m);
            if (rq) L.list[q][mbcnt64(m)] = (uint8_t)lane;
        }
        __builtin_amdgcn_wave_barrier();
        const int tmax = max(max(len[0], len[1]), max(len[2], len[3]));
        int t = 0;
        // ================================================================ stage A (lock-step test loop) of ONE segment, items -> L.abuf
        auto stage_a = [&](Seg& sg) {
            sg.n_it = 0; sg.n_items = 0; sg.it_lo = 0u; sg.it_hi = 0u; sg.it_first = 0u;
            // one tested list entry per quadrant; returns true when the segment is full (the entry is tested again in the next one)
            auto test_one = [&](const float4& cA, const float4& cB, const int cj) -> bool {
                const float power = gauss_power(cA.z, cA.w, cB.x, cA.x - pxf, cA.y - pyf);
                const float araw = gauss_alpha_raw(cB.y, power);
                const float alpha = fminf(TG_ALPHA_MAX, araw);
                const bool before = __float_as_uint(cB.z) < (uint32_t)last;
                const bool ok = before && (power <= 0.0f) && (alpha >= TG_ALPHA_MIN);
                const ull bal = TG_BALLOT(power <= 0.0f) & TG_BALLOT(before) & TG_BALLOT(alpha >= TG_ALPHA_MIN);
                const int nb = __popcll(bal);
                if (nb != 0) {
                    if (sg.n_items + nb > BQ_CAP) return true;         // segment full; this iteration is re-tested in the next one
                    if (lane == sg.n_it) { sg.it_lo = (uint32_t)bal; sg.it_hi = (uint32_t)(bal >> 32); sg.it_first = (uint32_t)sg.n_items; }
                    if (ok) {
                        T = T * __builtin_amdgcn_rcpf(1.0f - alpha);     // v_rcp_f32 (1 ulp): an IEEE divide is ~10 VALU
                        L.abuf[sg.n_items + mbcnt64(bal)] = make_float4(T, 0.f, araw, __uint_as_float(keybase | (uint32_t)cj));
                    }
                    sg.n_items += nb; ++sg.n_it;
                }
                return false;
            };
            // two register sets used and refilled alternately (no register rotation: see K6's loop)
            int jx = mylist[min(t, 63)], jy = mylist[min(t + 1, 63)];
            float4 xA = L.p.A[jx], xB = L.p.B[jx];
            float4 yA = xA, yB = xB;
            while (t < tmax && sg.n_it < BWD_MAX_IT) {
                yA = L.p.A[jy]; yB = L.p.B[jy];
                const int j0 = jx;
                jx = mylist[min(t + 2, 63)];
                if (test_one(xA, xB, j0)) break;
                ++t;
                if (!(t < tmax && sg.n_it < BWD_MAX_IT)) break;
                xA = L.p.A[jx]; xB = L.p.B[jx];
                const int j1 = jy;
                jy = mylist[min(t + 2, 63)];
                if (test_one(yA, yB, j1)) break;
                ++t;
            }
        };
        auto stage_c = [&](const Seg& sg) {
            // ================================================================ stage C1: per-pixel recurrence, iteration by iteration
            // dL/dalpha_i = T_i s_i - (B_i + T_final bg . dL/dcolour) / (1 - alpha_i),  B_i = sum over the contributors k BEHIND i
            // of s_k alpha_k T_k: one running sum per pixel (`behind`), back to front.  s_i (colour . dL/dcolour + geometry
            // channels) comes ready-made from stage B; this loop leaves w = alpha T and P = dL/dpower in the item.
            for (int k = 0; k < sg.n_it; ++k) {
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
                    // {w, P}: P = dL/dpower straight through the 0.99 clamp (lineage)
                    *reinterpret_cast<float2*>(&L.items[it * 3]) = make_float2(w, araw * dL_dalpha_);
                }
            }
            // ================================================================ stage C2: per-Gaussian moment sums, 16 lanes per task
            // task list: lane (k, q) = (iteration k, quadrant q) of the segment looks at its row of the iteration's ballot
            int ntask;
            {
                const int itk = lane >> 2, qq = lane & 3;
                const uint32_t blo = (uint32_t)__builtin_amdgcn_ds_bpermute(itk << 2, (int)sg.it_lo);
                const uint32_t bhi = (uint32_t)__builtin_amdgcn_ds_bpermute(itk << 2, (int)sg.it_hi);
                const uint32_t f0 = (uint32_t)__builtin_amdgcn_ds_bpermute(itk << 2, (int)sg.it_first);
                const ull b = ((ull)bhi << 32) | blo;
                const int c = __popc((uint32_t)(b >> (16 * qq)) & 0xFFFFu);
                const int below = __popcll(b & ((1ull << (16 * qq)) - 1ull));
                const bool havet = (itk < sg.n_it) && (c > 0);
                const ull tm = TG_BALLOT(havet);
                ntask = __popcll(tm);
                if (havet) L.task[mbcnt64(tm)] = (f0 + (uint32_t)below) | ((uint32_t)c << 8);       // first item | item count << 8
            }
            __builtin_amdgcn_wave_barrier();
            // Four tasks per round: every lane forms the 28 moment terms of its item, a transposing butterfly over the 16 lanes
            // (bank-masked DPP for lane^4 / lane^8, quad_perm for lane^1 / lane^2) leaves two of the 32 row slots in each lane,
            // and the 16 lanes add the Gaussian's 128-byte accumulator row as two 64-byte runs.
            {
                const int sub = lane & 15;
                for (int q0 = 0; q0 < ntask; q0 += 4) {
                    const int qi = q0 + (lane >> 4);
                    const bool live = qi < ntask;
                    const uint32_t task = live ? L.task[qi] : 0u;
                    const int first = live ? (int)(task & 255u) : BQ_CAP;
                    const bool have = (uint32_t)sub < (task >> 8);
                    // lanes without an item read the all-zero item behind the list: every moment below comes out 0 with no
                    // branch and no 32-register clear (the butterfly needs all 64 lanes anyway)
                    const int item = have ? first + sub : BQ_CAP;
                    // the task's Gaussian: survivor slot from its first item's key, index from the lane that holds that survivor
                    // (all 64 lanes are active here: bpermute reads 0 from an inactive source lane)
                    const int jt = KEY_J(__float_as_uint(L.items[first * 3].w));
                    const uint32_t gid = (uint32_t)__builtin_amdgcn_ds_bpermute(jt << 2, (int)id);
                    float part[32];
                    const float2 gxy = *reinterpret_cast<const float2*>(&L.p.A[jt]);
                    {
                        const float4 i0 = L.items[item * 3], i1 = L.items[item * 3 + 1];
                        const uint32_t key = __float_as_uint(i0.w);
                        const int pl = KEY_PL(key);
                        const float w = i0.x, P = i0.y;
                        const float dx = gxy.x - (float)(wave_px + KEY_OX(key)), dy = gxy.y - (float)(wave_py + KEY_OY(key));   // xy - pixel
                        // RAW MOMENTS about the splat centre (TexGSGrads.acc layout, texgs.h); K8, which has conic / opacity /
                        // G / g in registers anyway, turns them into dL/d(xy, conic, opacity, G, g, ...)
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
                    static_assert(M_N + 3 == 28, "the live-slot masks above cover slots 0..27");
                    // lane holds slots transposed_index(lane & 15) and 16 + that
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
        // Segment pipeline: A(n + 1) runs BETWEEN the two halves of B(n).  The taps of segment n are gathers that miss the L2 (1-2 us
        // each; 43 % of this kernel's wave-cycles were s_waitcnt on them) and stage A is a few hundred instructions without one
        // vector-memory operation -- so every tap of segment n is in flight while the next segment's lists are walked, and the
        // in-order vmcnt the back half waits on counts nothing that was issued in between.  (Scheduling barriers: left alone, the
        // machine scheduler interleaved both back halves and waited for all eight taps first.)  No look-ahead across chunks: the
        // next chunk's planes are not in LDS yet.
        constexpr int NR = (BQ_CAP + 63) / 64;          // rounds of 64 items per segment, all of them in flight together
        static_assert(BQ_CAP <= 192, "item indices travel in 8 bits (task words), and index BQ_CAP is the all-zero item");
        Seg nxt;
        stage_a(nxt);
        while (nxt.n_items > 0) {
            const Seg cur = nxt;
            __builtin_amdgcn_wave_barrier();
            // ================================================================ stage B, front halves
            Round R[NR];
#if K7_GATHER
#pragma unroll
            for (int r = 0; r < NR; ++r)
                if (r == 0 || r * 64 < cur.n_items) front_a(r * 64, R[r], cur.n_items);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < NR; ++r)
                if (r == 0 || r * 64 < cur.n_items) front_b(R[r]);
#else
#pragma unroll
            for (int r = 0; r < NR; ++r)
                if (r == 0 || r * 64 < cur.n_items) { front_a(r * 64, R[r], cur.n_items); front_b(R[r]); }
#endif
            __builtin_amdgcn_sched_barrier(0);
            nxt.n_items = 0; nxt.n_it = 0;
            if (t < tmax) {
                stage_a(nxt);
            }
            // ================================================================ stage B, back halves
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                __builtin_amdgcn_sched_barrier(0);      // back(r) waits for round r's loads only
                if (r == 0 || r * 64 < cur.n_items) back(R[r]);
            }
            __builtin_amdgcn_wave_barrier();
            if constexpr (GEO) stage_c(cur);
            __builtin_amdgcn_wave_barrier();
        }
    }
    if constexpr (TEX) {
        // A reservation this block did not use up -- impossible while K6 and K7 agree on every footprint (same decisions, same uv
        // arithmetic); should they ever not, the reduce must not sum whatever an earlier call left in the unused slots.
        __builtin_amdgcn_wave_barrier();
        const uint32_t q1 = min(L.tend[lane], tb.cap);
        for (uint32_t q = L.tpos[lane]; q < q1; ++q) { Rec4 z; z.a = 0u; z.b = 0u; z.c = 0u; z.d = 0u; reinterpret_cast<Rec4*>(tb.rec)[q] = z; }
    }
#ifdef K7_TRACE
    if (lane == 0 && blockIdx.x < K7_TRACE_BLOCKS) {
        unsigned long long* tr = k7_trace + 4 * (size_t)blockIdx.x;
        tr[0] = trace_t0; tr[1] = wall_clock64();
        tr[2] = (unsigned long long)__builtin_amdgcn_s_getreg(6164) | ((unsigned long long)__builtin_amdgcn_s_getreg(63492) << 8);     // XCC_ID[3:0] | HW_ID << 8
        tr[3] = (unsigned long long)(uint32_t)ns | ((unsigned long long)(uint32_t)todo << 32);
    }
#endif
}

