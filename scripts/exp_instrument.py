#!/usr/bin/env python3
"""Experiments only (nothing here ships): make an INSTRUMENTED copy of csrc/render.hip under build_exp/exp/ with
  -DTG_STATS   dynamic work counters of K6 / K7 (read by scripts/exp_stats.py through texgs_debug_stats), and
  -DABL=n      timing-only ablations of K7 (1 no C1, 2 no C2, 4 no stage B, 8 no record stores / scatter, 16 no bin
               bookkeeping at all), -DFABL=1 K6 without its dense phase
so that the product translation unit carries no experiment scaffolding.  Every anchor must match exactly once."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(ROOT, "texture-gs_amd/csrc/render.hip")).read()


def rep(old, new, count=1):
    global src
    assert src.count(old) == count, (src.count(old), old[:90])
    src = src.replace(old, new)


rep('typedef unsigned long long ull;', '''typedef unsigned long long ull;
#ifdef TG_STATS
__device__ unsigned long long g_stats[32];
#define CNT(i, n) (st_[i] += (unsigned)(n))
#define CNT_DECL unsigned st_[16] = {0}
#define CNT_FLUSH(base) do { if (lane == 0) for (int i_ = 0; i_ < 16; ++i_) if (st_[i_]) atomicAdd(&g_stats[(base) + i_], (unsigned long long)st_[i_]); } while (0)
#else
#define CNT(i, n) ((void)0)
#define CNT_DECL
#define CNT_FLUSH(base) ((void)0)
#endif
#ifndef ABL
#define ABL 0
#endif
#ifndef FABL
#define FABL 0
#endif
''')
# ---- K6
rep('    int r = 0, nq = 0, qh = 0;                 // next raw list position',
    '    CNT_DECL;\n    CNT(0, 1);\n    int r = 0, nq = 0, qh = 0;                 // next raw list position')
rep('''            nq += __popcll(m);
        }
        if (nq == 0) break;
        // ---- chunk: up to 64 survivors, lane = survivor
        const int take = min(64, nq);''', '''            nq += __popcll(m);
            CNT(1, 1); CNT(2, __popcll(TG_BALLOT(idx < todo))); CNT(3, __popcll(m));
        }
        if (nq == 0) break;
        // ---- chunk: up to 64 survivors, lane = survivor
        const int take = min(64, nq);
        CNT(4, 1);''')
rep('        if (a.surv != nullptr) {                // a backward will follow: it replays exactly these survivors, back to front',
    '''#ifdef TG_STATS
        {   // what 16 per-2x2-quad lists would give: first-order upper bound of the concave falloff at the quad centre
            int mx = 0, sm = 0, mx_exact = 0;
            for (int qd = 0; qd < 16; ++qd) {
                const float cx = wpx + 2.f * (float)(qd & 3) + 0.5f, cy = wpy + 2.f * (float)(qd >> 2) + 0.5f;
                const float dx = T0.x - cx, dy = T0.y - cy;
                const float p = T0.z * dx * dx + T0.w * dx * dy + T1.x * dy * dy;
                const float g1 = 2.f * T0.z * dx + T0.w * dy, g2 = T0.w * dx + 2.f * T1.x * dy;
                const bool ok = (lane < take) && T1.z >= 0.f && (p + 0.5f * (fabsf(g1) + fabsf(g2)) >= T1.w - 1e-3f * fabsf(T1.w) - 1e-4f);
                const int n = __popcll(TG_BALLOT(ok));
                mx = max(mx, n); sm += n;
                const bool ex = (lane < take) && block_reachable(T0.x, T0.y, T0.z, T0.w, T1.x, T1.w, T1.z, cx - 0.5f, cy - 0.5f, 1.0f);
                mx_exact = max(mx_exact, __popcll(TG_BALLOT(ex)));
            }
            CNT(11, mx); CNT(12, sm); CNT(13, mx_exact);
        }
#endif
        if (a.surv != nullptr) {                // a backward will follow: it replays exactly these survivors, back to front''')
rep('        int tmax = max(max(len[0], len[1]), max(len[2], len[3]));',
    '        int tmax = max(max(len[0], len[1]), max(len[2], len[3]));\n        CNT(5, len[0] + len[1] + len[2] + len[3]); CNT(6, tmax);\n        if (FABL & 2) tmax = 0;')
rep('            if (m_ok == 0ull) continue;', '            CNT(7, 1);\n            if (m_ok == 0ull) continue;\n            CNT(8, 1);')
rep('''                qtail += __popcll(bal);
                if (qtail - qhead >= 64) {
                    __builtin_amdgcn_wave_barrier();''', '''                qtail += __popcll(bal);
                CNT(9, __popcll(bal));
                if (FABL & 1) qhead = qtail;
                if (qtail - qhead >= 64) {
                    CNT(10, 1);
                    __builtin_amdgcn_wave_barrier();''')
rep('''        if (qtail - qhead > 0) {
            __builtin_amdgcn_wave_barrier();''', '''        if (qtail - qhead > 0) {
            CNT(10, 1);
            __builtin_amdgcn_wave_barrier();''')
rep('''    finish();
    __builtin_amdgcn_wave_barrier();
    if (bin_count != nullptr && lane < 8''', '''    finish();
    CNT_FLUSH(0);
    __builtin_amdgcn_wave_barrier();
    if (bin_count != nullptr && lane < 8''')
# ---- K7
rep('    const int ns = (int)a.surv_cnt[4 * tile + wave];', '    CNT_DECL;\n    CNT(0, 1);\n    const int ns = (int)a.surv_cnt[4 * tile + wave];\n    CNT(3, ns);')
rep('        __builtin_amdgcn_wave_barrier();\n        float4 T0, T1;\n        load_chunk(a, L.p, lane, live, id, pos, T0, T1);',
    '        CNT(4, 1);\n        __builtin_amdgcn_wave_barrier();\n        float4 T0, T1;\n        load_chunk(a, L.p, lane, live, id, pos, T0, T1);')
rep('        const int tmax = max(max(len[0], len[1]), max(len[2], len[3]));\n        int t = 0;',
    '        const int tmax = (ABL & 32) ? 0 : max(max(len[0], len[1]), max(len[2], len[3]));\n        CNT(5, len[0] + len[1] + len[2] + len[3]); CNT(6, tmax);\n        int t = 0;')
rep('''                    const int nb = __popcll(bal);
                    if (nb != 0) {
                        if (n_items + nb > BQ_CAP) break;''', '''                    const int nb = __popcll(bal);
                    CNT(7, 1);
                    if (nb != 0) {
                        if (n_items + nb > BQ_CAP) break;
                        CNT(8, 1); CNT(9, nb);''')
rep('            if (n_items == 0) continue;\n', '            if (n_items == 0) continue;\n            CNT(10, 1); CNT(11, (n_items + 63) >> 6);\n')
rep('''            {
                Round R0, R1;
                front(0, R0);''', '''            if (!(ABL & 4)) {
                Round R0, R1;
                front(0, R0);''')
rep('        R.binned = R.have && tb.rec != nullptr && tap_binned(ct);',
    '        R.binned = R.have && tb.rec != nullptr && tap_binned(ct) && !(ABL & 16);')
rep('        if (R.binned && slot < TB_LIST_MAX && pos < R.b1 && pos < tb.cap) {',
    '        if (ABL & 24) {\n        } else if (R.binned && slot < TB_LIST_MAX && pos < R.b1 && pos < tb.cap) {')
rep('            for (int k = 0; k < n_it; ++k) {\n                const uint32_t blo = (uint32_t)__builtin_amdgcn_readlane((int)it_lo, k);',
    '            if (!(ABL & 1)) for (int k = 0; k < n_it; ++k) {\n                const uint32_t blo = (uint32_t)__builtin_amdgcn_readlane((int)it_lo, k);')
rep('                        if (lo != 0.f) unsafeAtomicAdd(row, lo);\n                        if (hi != 0.f) unsafeAtomicAdd(row + 16, hi);',
    '                        if (ABL & 64) { if (lo == 12345.f && hi == 54321.f) row[0] = lo; } else {\n                        if (lo != 0.f) unsafeAtomicAdd(row, lo);\n                        if (hi != 0.f) unsafeAtomicAdd(row + 16, hi); }')
rep('                const bool havet = (itk < n_it) && (c > 0);\n',
    '''                const bool havet = (itk < n_it) && (c > 0);
#ifdef TG_STATS
                {   // what per-Gaussian task merging inside a segment would give
                    const int jmine = KEY_J(__float_as_uint(L.items[(havet ? (int)(f0 + (uint32_t)below) : BQ_CAP) * 3].w));
                    int distinct = 0, merged = 0;
                    for (int jj = 0; jj < 64; ++jj) {
                        const ull mm = TG_BALLOT(havet && jmine == jj);
                        if (mm == 0ull) continue;
                        int n = 0;
                        ull m2 = mm;
                        while (m2) { const int l = __ffsll((long long)m2) - 1; m2 &= m2 - 1; n += __builtin_amdgcn_readlane(c, l); }
                        ++distinct; merged += (n + 15) >> 4;
                    }
                    CNT(14, distinct); CNT(15, merged);
                }
#endif
''')
rep('                ntask = __popcll(tm);\n', '                ntask = __popcll(tm);\n                CNT(12, ntask); CNT(13, (ntask + 3) >> 2);\n                { const int n8_ = __popcll(TG_BALLOT(havet && c <= 8)), n4_ = __popcll(TG_BALLOT(havet && c <= 4)); CNT(1, n8_); CNT(2, n4_); CNT(0, ((ntask - n8_ + 3) >> 2) + ((n8_ + 7) >> 3)); }\n                if (ABL & 2) ntask = 0;\n')
rep('''            __builtin_amdgcn_wave_barrier();
        }
    }
}
''', '''            __builtin_amdgcn_wave_barrier();
        }
    }
    CNT_FLUSH(16);
    if (ABL != 0 && L.items[lane * 3].x == 12345.678f) acc[0] = behind + T;    // ablation builds: keep the chains alive
}
''')
# ABL & 128: K7's record slots without the returning global atomic and without the per-lane base loads (wrong lists, same stores)
rep('        if (leader) R.slot0 = atomicAdd(tb.cursor + R.bin, (uint32_t)my_n);\n        if (R.binned) { R.b0 = tb.base[R.bin]; R.b1 = tb.base[R.bin + 1u]; }',
    '''        if (ABL & 128) {
            if (leader) R.slot0 = (uint32_t)((rbase * 7 + n_items * 13) & 2047);
            if (R.binned) { R.b0 = R.bin * 3000u; R.b1 = R.b0 + 3000u; }
        } else {
        if (leader) R.slot0 = atomicAdd(tb.cursor + R.bin, (uint32_t)my_n);
        if (R.binned) { R.b0 = tb.base[R.bin]; R.b1 = tb.base[R.bin + 1u]; }
        }''')
# ABL & 256: the ballot grouping loop replaced by one group per round (wrong lists, same atomics and stores)
rep('        while (pend != 0ull) {\n            const int l0 = __ffsll((long long)pend) - 1;\n            const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)R.bin, l0);\n            const ull m = TG_BALLOT(R.binned && R.bin == b0);',
    '''        while (pend != 0ull) {
            const int l0 = __ffsll((long long)pend) - 1;
            const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)R.bin, l0);
            const ull m = (ABL & 256) ? pend : TG_BALLOT(R.binned && R.bin == b0);''')
src += '''
#ifdef TG_STATS
extern "C" int texgs_debug_stats(unsigned long long* out32, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_stats), 32 * sizeof(unsigned long long));
    if (e == hipSuccess && reset) { unsigned long long z[32] = {0}; e = hipMemcpyToSymbol(HIP_SYMBOL(g_stats), z, sizeof(z)); }
    return (int)e;
}
#endif
'''
out = os.path.join(ROOT, "build_exp/exp/render_exp.hip")
os.makedirs(os.path.dirname(out), exist_ok=True)
open(out, "w").write(src)
print("wrote", out)
