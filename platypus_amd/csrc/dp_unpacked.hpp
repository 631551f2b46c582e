// dp_unpacked.hpp -- the same banded DP as dp_core.hpp (reference: src/c/align.c:77-586, score only), in the
// formulation that fits gfx950's VALU issue rules best.
//
// Measured (tools/ubench/valu_rate.hip, profiles/r01_valu_issue_rates.txt): only 32-bit-encoded VOP2 ops
// (v_add_u16, v_min_i16, v_min_u16, v_xor_b32 ...) issue at full rate; every VOP3/VOP3P/SDWA/DPP op -- including the
// packed v_pk_add_u16 / v_pk_min_i16 and the v_alignbit/v_perm lane shifts of dp_core.hpp -- issues at half rate.  A
// packed op therefore buys nothing over two scalar 16-bit ops, and the lane shifts are pure overhead.  Here every one
// of the reference's 8 int16 lanes lives in its own VGPR (low half), all arithmetic is scalar 16-bit VOP2 (still
// genuinely wrapping 16-bit adds / signed 16-bit mins, i.e. the reference's values by construction), and the lane
// shifts of the sliding windows cost nothing: the windows are ring buffers whose phase (h mod 8) is a template
// parameter, so a "shift" is a compile-time renaming of registers.
//
// State carried between steps (everything else is transient):
//   mi1 = min(M,I) of the even parity, d1;  mi2 = min(M,I) of the odd parity, d2;
//   un  = min(I2 + ge, M2 + go) -- the even I of the NEXT step before "+ nucprior" (it only needs the odd state and
//         the odd half-step's gap-open window, which is the next step's even window).
#pragma once
#include "dp_core.hpp"

namespace plat {

typedef unsigned short u16;

__device__ __forceinline__ u16 a16(u16 a, u16 b) { return (u16)(a + b); }                              // v_add_u16 (wraps)
__device__ __forceinline__ u16 mn16(u16 a, u16 b) { return (short)a < (short)b ? a : b; }              // v_min_i16
__device__ __forceinline__ u16 mnu16(u16 a, u16 b) { return a < b ? a : b; }                           // v_min_u16

template <bool HAS_N>
struct DPU {
    u16 mi1[8], d1[8], mi2[8], d2[8], un[8];
    u16 m2[8], i2[8];                        // live across steps only during the 8 forced steps (free-start fix-up)
    u16 s1[8], gp[8], nq[8], s2[8], q2[8];   // ring buffers: haplotype side indexed by x & 7, read side by y & 7
    int minscore;

    __device__ __forceinline__ void init(const uint32_t (&hw)[8]) {
        minscore = 0x7800;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            mi1[k] = d1[k] = mi2[k] = d2[k] = m2[k] = i2[k] = (u16)INF16;
            s2[k] = 0x01FFu;                              // never equals a base code
            q2[k] = 0x0100u;                              // 64*4, align.c:159
            s1[k] = (u16)(hw[k] & 0xFFFFu);               // x = k -> slot k
            gp[k] = (u16)(hw[k] >> 16);
            if (HAS_N) nq[k] = (hw[k] & 0xFFFFu) == CODE_N ? (u16)0 : (u16)INF16;
        }
        // un for step 0: min(i2 + ge, m2 + go[k]) with i2 = m2 = pos_inf
#pragma unroll
        for (int k = 0; k < 8; ++k) un[k] = mn16(a16((u16)INF16, 12), a16((u16)INF16, gp[k]));
    }

    // One full step h with phase PH = h & 7.  FL >= 0: free start on lane FL (h < 8).  EL >= 0: extraction lane.
    template <int PH, int FL, int EL>
    __device__ __forceinline__ void step(uint32_t rw, uint32_t hwd, bool ext_rt = false) {
        constexpr u16 GE = 12, NP = 8, NEG = 0x8000u;      // gapextend*4, nucprior*4 (chaplotype.pyx:607-608), -0x8000
        u16 m1[8], i1[8], S[8];
        // ---------------- even half-step: lane k works on x = h + k (slot (PH+k)&7), y = h - k (slot (PH-k)&7)
        s2[PH] = (u16)(rw & 0xFFFFu);
        q2[PH] = (u16)(rw >> 16);
        if (FL >= 0) {
            constexpr int f = FL >= 0 ? FL : 0;
            mi1[f] = NEG; mi2[f] = NEG;
            un[f] = mn16(a16(i2[f], GE), a16(NEG, gp[(PH + f) & 7]));          // align.c:331-335 with the forced m2
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) S[k] = mn16(mi1[k], d1[k]);
        if (EL >= 0) take(S[EL >= 0 ? EL : 0]);
        else if (FL == 7) { if (ext_rt) take(S[0]); }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            u16 c = mnu16((u16)(s1[(PH + k) & 7] ^ s2[(PH - k) & 7]), q2[(PH - k) & 7]);
            if (HAS_N) c = mn16(c, nq[(PH + k) & 7]);
            m1[k] = a16(S[k], c);
            i1[k] = a16(un[k], NP);
        }
        // haplotype word 8+h enters slot PH (it replaces x = h, which only the even half used)
        s1[PH] = (u16)(hwd & 0xFFFFu);
        gp[PH] = (u16)(hwd >> 16);
        if (HAS_N) nq[PH] = (hwd & 0xFFFFu) == CODE_N ? (u16)0 : (u16)INF16;
        // d1[k+1] = min(d2[k] + ge, mi2[k] + go[h+1+k]);  d1[0] = inf          (align.c:320-329)
        {
            u16 t[7];
#pragma unroll
            for (int k = 0; k < 7; ++k) t[k] = mn16(a16(d2[k], GE), a16(mi2[k], gp[(PH + 1 + k) & 7]));
#pragma unroll
            for (int k = 0; k < 7; ++k) d1[k + 1] = t[k];
            d1[0] = (u16)INF16;
        }
        u16 i2n[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) mi1[k] = mn16(m1[k], i1[k]);
        // i2[k] = min(i1[k+1] + ge, m1[k+1] + go[h+1+k]) + np;  i2[7] = inf     (align.c:478-484)
#pragma unroll
        for (int k = 0; k < 7; ++k) i2n[k] = a16(mn16(a16(i1[k + 1], GE), a16(m1[k + 1], gp[(PH + 1 + k) & 7])), NP);
        i2n[7] = (u16)INF16;
        // ---------------- odd half-step: lane k works on x = h + 1 + k (slot (PH+1+k)&7), y = h - k
#pragma unroll
        for (int k = 0; k < 8; ++k) S[k] = mn16(mi2[k], d2[k]);
        if (EL >= 0) take(S[EL >= 0 ? EL : 0]);
        else if (FL == 7) { if (ext_rt) take(S[0]); }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            u16 c = mnu16((u16)(s1[(PH + 1 + k) & 7] ^ s2[(PH - k) & 7]), q2[(PH - k) & 7]);
            if (HAS_N) c = mn16(c, nq[(PH + 1 + k) & 7]);
            const u16 m2k = a16(S[k], c);
            const u16 g = gp[(PH + 1 + k) & 7];
            d2[k] = mn16(a16(d1[k], GE), a16(mi1[k], g));                       // align.c:472-476
            mi2[k] = mn16(m2k, i2n[k]);
            un[k] = mn16(a16(i2n[k], GE), a16(m2k, g));                         // next step's even I before + np
            if (FL >= 0) { m2[k] = m2k; i2[k] = i2n[k]; }
        }
    }

    __device__ __forceinline__ void take(u16 v) {
        const int sc = (int)(short)v;
        minscore = sc < minscore ? sc : minscore;
    }

    // rotate the ring buffers so that the current phase r becomes phase 0: new[p] = old[(p + r) & 7]
    __device__ __forceinline__ void rotate(int r) {
        rot(s1, r); rot(gp, r); rot(s2, r); rot(q2, r);
        if (HAS_N) rot(nq, r);
    }
    static __device__ __forceinline__ void rot(u16 (&a)[8], int r) {
        u16 t[8];
#pragma unroll
        for (int sft = 1; sft <= 4; sft <<= 1) {
            const bool on = (r & sft) != 0;
#pragma unroll
            for (int p = 0; p < 8; ++p) t[p] = on ? a[(p + sft) & 7] : a[p];
#pragma unroll
            for (int p = 0; p < 8; ++p) a[p] = t[p];
        }
    }

    __device__ __forceinline__ int result() const { return (minscore + 0x8000) >> 2; }   // align.c:520
};

template <bool HAS_N, class RW, class HW>
__device__ __forceinline__ int dp_run_u(DPU<HAS_N>& dp, int len2, RW rw, HW hw)
{
    const bool l7 = (len2 == 7);
    dp.template step<0, 0, -1>(rw(0), hw(0));
    dp.template step<1, 1, -1>(rw(1), hw(1));
    dp.template step<2, 2, -1>(rw(2), hw(2));
    dp.template step<3, 3, -1>(rw(3), hw(3));
    dp.template step<4, 4, -1>(rw(4), hw(4));
    dp.template step<5, 5, -1>(rw(5), hw(5));
    dp.template step<6, 6, -1>(rw(6), hw(6));
    dp.template step<7, 7, -1>(rw(7), hw(7), l7);
    int h = 8;
    if (l7) {      // h = 7 was both the last forced and the first extra step; extra steps 1..7 follow at phases 0..6
        dp.template step<0, -1, 1>(rw(8), hw(8));
        dp.template step<1, -1, 2>(rw(9), hw(9));
        dp.template step<2, -1, 3>(rw(10), hw(10));
        dp.template step<3, -1, 4>(rw(11), hw(11));
        dp.template step<4, -1, 5>(rw(12), hw(12));
        dp.template step<5, -1, 6>(rw(13), hw(13));
        dp.template step<6, -1, 7>(rw(14), hw(14));
        return dp.result();
    }
    for (; h + 8 <= len2; h += 8) {
        dp.template step<0, -1, -1>(rw(h), hw(h));
        dp.template step<1, -1, -1>(rw(h + 1), hw(h + 1));
        dp.template step<2, -1, -1>(rw(h + 2), hw(h + 2));
        dp.template step<3, -1, -1>(rw(h + 3), hw(h + 3));
        dp.template step<4, -1, -1>(rw(h + 4), hw(h + 4));
        dp.template step<5, -1, -1>(rw(h + 5), hw(h + 5));
        dp.template step<6, -1, -1>(rw(h + 6), hw(h + 6));
        dp.template step<7, -1, -1>(rw(h + 7), hw(h + 7));
    }
    const int r = len2 - h;                         // 0..7 in-read steps left, at phases 0..r-1
    if (r > 0) dp.template step<0, -1, -1>(rw(h), hw(h));
    if (r > 1) dp.template step<1, -1, -1>(rw(h + 1), hw(h + 1));
    if (r > 2) dp.template step<2, -1, -1>(rw(h + 2), hw(h + 2));
    if (r > 3) dp.template step<3, -1, -1>(rw(h + 3), hw(h + 3));
    if (r > 4) dp.template step<4, -1, -1>(rw(h + 4), hw(h + 4));
    if (r > 5) dp.template step<5, -1, -1>(rw(h + 5), hw(h + 5));
    if (r > 6) dp.template step<6, -1, -1>(rw(h + 6), hw(h + 6));
    dp.rotate(r);
    h = len2;                                        // the 8 extra steps (align.c:199), phases 0..7 after the rotation
    dp.template step<0, -1, 0>(rw(h), hw(h));
    dp.template step<1, -1, 1>(rw(h + 1), hw(h + 1));
    dp.template step<2, -1, 2>(rw(h + 2), hw(h + 2));
    dp.template step<3, -1, 3>(rw(h + 3), hw(h + 3));
    dp.template step<4, -1, 4>(rw(h + 4), hw(h + 4));
    dp.template step<5, -1, 5>(rw(h + 5), hw(h + 5));
    dp.template step<6, -1, 6>(rw(h + 6), hw(h + 6));
    dp.template step<7, -1, 7>(rw(h + 7), hw(h + 7));
    return dp.result();
}

}  // namespace plat
