// dp_traceback.hpp -- the banded DP in the reference's traceback mode (src/c/align.c:96,345-365,494-515), its
// backtrace (align.c:518-577) and calculateFlankScore (align.c:593-644), for --calculateFlankScore=1
// (calign.pyx:235-245,261-264).
//
// In this mode the low 2 bits of every M/I/D value carry the state the value came from (M=0, I=1, D=3): they take part
// in the 16-bit min() tie-breaks, so the traceback -- and therefore the flank score -- is only reproduced by carrying
// them exactly as the reference does.  One lane per DP as in dp_core.hpp, but one reference SSE lane per VGPR (plain
// 16-bit wrapping adds / signed mins): this is the optional slow path, clarity wins over the last 20 %.
//
// Back-pointers: per half-step the reference stores 8 x int16 of which 6 bits per lane are ever read; they are packed
// here into one 64-bit word per half-step (lane k in bits 6k..6k+5: m | i<<2 | d<<4), laid out [half-step][job] so
// a wavefront's stores are contiguous.  The backtrace never materialises the aln1/aln2 strings: it walks the
// back-pointers once and evaluates calculateFlankScore's terms on the way; the only coupling between neighbouring
// alignment columns (gap-open vs gap-extend depends on the PREVIOUS column, i.e. the next one met when walking
// backwards) is handled with a one-column delay.
#pragma once
#include "dp_unpacked.hpp"

namespace plat {

struct TbView {                      // back-pointer storage of one job
    unsigned long long* base;        // &bp[0][job]
    size_t stride;                   // jobs per half-step row
    __device__ __forceinline__ void put(int s, unsigned long long v) const { base[(size_t)s * stride] = v; }
    __device__ __forceinline__ unsigned get(int s, int lane) const {
        return (unsigned)(base[(size_t)(s < 0 ? 0 : s) * stride] >> (6 * lane)) & 63u;
    }
};

// Forward pass with labels.  w0: haplotype words of slice positions 0..7; rw(h)/hw(h) as in dp_run().
// Returns the score; *min_idx = half-step index of the minimum (align.c:261-288,416-443).
template <class RW, class HW>
__device__ __forceinline__ int dp_forward_tb(const uint32_t (&w0)[8], int len2, RW rw, HW hw, const TbView& bp, int* min_idx)
{
    constexpr u16 GE = 12, NP = 8, NEG = 0x8000u, INF = (u16)INF16;      // gapextend 3, nucprior 2 (chaplotype.pyx:607-608), x4
    u16 m1[8], i1[8], d1[8], m2[8], i2[8], d2[8], s1[8], nq[8], gp[8], s2[8], q2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        m1[k] = i1[k] = d1[k] = m2[k] = i2[k] = d2[k] = INF;             // align.c:139-144
        s1[k] = (u16)(w0[k] & 0xFFFFu);
        nq[k] = s1[k] == (u16)CODE_N ? (u16)0 : INF;                     // align.c:175-178
        gp[k] = (u16)(w0[k] >> 16);
        s2[k] = 0x01FFu;                                                  // never equals a base code
        q2[k] = 0x0100u;                                                  // 64*4, align.c:159
    }
    int minscore = 0x7800, midx = -1;
    const int nsteps = len2 + 8;
    for (int h = 0; h < nsteps; ++h) {                                    // align.c:199
        const uint32_t rwd = rw(h);
        // ---------------- even half-step
#pragma unroll
        for (int k = 7; k > 0; --k) { s2[k] = s2[k - 1]; q2[k] = q2[k - 1]; }
        s2[0] = (u16)(rwd & 0xFFFFu);
        q2[0] = (u16)(rwd >> 16);
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k == h) { m1[k] = NEG; m2[k] = NEG; }                     // free start, align.c:244-250
#pragma unroll
        for (int k = 0; k < 8; ++k) m1[k] = mn16(m1[k], mn16(i1[k], d1[k]));
        if (h >= len2) {
            const int e = h - len2;
            u16 v = m1[0];
#pragma unroll
            for (int k = 1; k < 8; ++k) v = (k == e) ? m1[k] : v;
            if ((int)(short)v < minscore) { minscore = (int)(short)v; midx = 2 * h; }
        }
        u16 t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const u16 sub = mn16(mnu16((u16)(s1[k] ^ s2[k]), q2[k]), nq[k]);   // align.c:314-318
            m1[k] = a16(m1[k], sub);
            t[k] = mn16(a16(d2[k], GE), a16(mn16(m2[k], i2[k]), k < 7 ? gp[k + 1] : (u16)0));   // align.c:320-324
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) i1[k] = a16(mn16(a16(i2[k], GE), a16(m2[k], gp[k])), NP);     // align.c:331-335
#pragma unroll
        for (int k = 7; k > 0; --k) d1[k] = t[k - 1];                    // align.c:326-329
        d1[0] = INF;
        {
            unsigned long long w = 0;                                     // align.c:345-365
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const unsigned bits = (m1[k] & 3u) | ((i1[k] & 3u) << 2) | ((d1[k] & 3u) << 4);
                w |= (unsigned long long)bits << (6 * k);
                m1[k] = (u16)(m1[k] & ~3u);
                i1[k] = (u16)((i1[k] & ~3u) | 1u);
                d1[k] = (u16)((d1[k] & ~3u) | 3u);
            }
            bp.put(2 * h, w);
        }
        // ---------------- odd half-step
        {
            uint32_t hwd = hw(h);
            if (h == nsteps - 1) hwd = (uint32_t)CODE_N | ((uint32_t)gp[7] << 16);   // past the slice: 'N', go[len1-1] (align.c:376,387)
#pragma unroll
            for (int k = 0; k < 7; ++k) { s1[k] = s1[k + 1]; nq[k] = nq[k + 1]; gp[k] = gp[k + 1]; }
            s1[7] = (u16)(hwd & 0xFFFFu);
            nq[7] = s1[7] == (u16)CODE_N ? (u16)0 : INF;
            gp[7] = (u16)(hwd >> 16);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) m2[k] = mn16(m2[k], mn16(i2[k], d2[k]));
        if (h >= len2) {
            const int e = h - len2;
            u16 v = m2[0];
#pragma unroll
            for (int k = 1; k < 8; ++k) v = (k == e) ? m2[k] : v;
            if ((int)(short)v < minscore) { minscore = (int)(short)v; midx = 2 * h + 1; }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const u16 sub = mn16(mnu16((u16)(s1[k] ^ s2[k]), q2[k]), nq[k]);   // align.c:466-470
            m2[k] = a16(m2[k], sub);
            d2[k] = mn16(a16(d1[k], GE), a16(mn16(m1[k], i1[k]), gp[k]));      // align.c:472-476
        }
#pragma unroll
        for (int k = 0; k < 7; ++k) i2[k] = a16(mn16(a16(i1[k + 1], GE), a16(m1[k + 1], gp[k])), NP);   // align.c:478-484
        i2[7] = INF;
        {
            unsigned long long w = 0;                                     // align.c:494-515
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const unsigned bits = (m2[k] & 3u) | ((i2[k] & 3u) << 2) | ((d2[k] & 3u) << 4);
                w |= (unsigned long long)bits << (6 * k);
                m2[k] = (u16)(m2[k] & ~3u);
                i2[k] = (u16)((i2[k] & ~3u) | 1u);
                d2[k] = (u16)((d2[k] & ~3u) | 3u);
            }
            bp.put(2 * h + 1, w);
        }
    }
    *min_idx = midx;
    return (minscore + 0x8000) >> 2;                                      // align.c:520
}

// Backtrace (align.c:523-577) fused with calculateFlankScore (align.c:593-644).
//   hwf(x)  : haplotype word (base code | 4*gapopen << 16) of position x of the WHOLE haplotype, st = slice start in it
//   rwf(y)  : read word of read position y
// Returns the flank score to subtract.
template <class HWF, class RWF>
__device__ __forceinline__ int tb_flank_score(const TbView& bp, int min_idx, int len2, HWF hwf, int st, int hapLen, int hapFlank, RWF rwf)
{
    int s = min_idx;
    int i = s / 2 - len2;
    int y = len2;
    int x = s - y;
    int state = bp.get(s, i) & 3;
    s -= 2;
    int flank = 0;
    // the column met last (= the next one in forward order); its gap cost waits for the state of the column before it
    int pend_state = 0, pend_open = 0, pend_ext = 0;
    while (y > 0) {
        const unsigned e = bp.get(s, i);
        const int newstate = (state == 0 ? e : state == 1 ? (e >> 2) : (e >> 4)) & 3;
        int cur, open = 0, ext = 0, mcost = 0;
        if (state == 0) {                                                 // match column: hap[x] over read[y]
            cur = 0;
            s -= 2; --x; --y;
            const int xg = st + x;
            const uint32_t hwd = hwf(xg < 0 ? 0 : xg), rwd = rwf(y);
            const bool in = xg < hapFlank || xg >= hapLen - hapFlank;
            if (in && (hwd & 0xFFFFu) != (rwd & 0xFFFFu) && (hwd & 0xFFFFu) != CODE_N) mcost = (int)(rwd >> 18);   // quals[y]
        } else if (state == 1) {                                          // insertion: '-' over read[y]
            cur = 1;
            i += s & 1; s -= 1; --y;
            const int xg = st + x;
            if (xg < hapFlank || xg >= hapLen - hapFlank) {
                open = (int)(hwf(xg - 1 < 0 ? 0 : xg - 1) >> 18) + 2;     // localGapOpen[x-1] + nucprior
                ext = 3 + 2;                                              // gapextend + nucprior
            }
        } else {                                                          // deletion: hap[x] over '-'
            cur = 3;
            s -= 1; i -= s & 1; --x;
            const int xg = st + x;
            if (xg < hapFlank || xg >= hapLen - hapFlank) {
                open = (int)(hwf(xg < 0 ? 0 : xg) >> 18);                  // localGapOpen[x]
                ext = 3;
            }
        }
        if (pend_state != 0) flank += (pend_state == cur) ? pend_ext : pend_open;
        pend_state = cur; pend_open = open; pend_ext = ext;
        flank += mcost;
        state = newstate;
    }
    if (pend_state != 0) flank += pend_open;                              // first column: prevstate starts as 'M' (align.c:601)
    return flank;
}

}  // namespace plat
