// dp_core.hpp -- banded affine-gap DP of Platypus' fastAlignmentRoutine for gfx950 (CDNA4).
//
// Reference behaviour: src/c/align.c:77-586 (score-only mode, aln1 == NULL).  The reference keeps
// 8 x int16 lanes in one SSE register per DP state; here ONE GPU LANE owns ONE whole alignment and
// keeps the same 8 int16 lanes packed two-per-VGPR (4 VGPRs per state vector, lanes k and k+4 together), so a wave64
// advances 64 alignments per instruction with v_pk_add_u16 / v_pk_min_i16 / v_pk_min_u16 -- genuinely
// 16-bit wrapping arithmetic, i.e. the same values as _mm_add_epi16/_mm_min_epi16 by construction.
// The reference's lane shifts (_mm_slli/_mm_srli_si128 by one lane) are a register renaming plus one v_perm_b32
// (see V8).  No MFMA: this is a min-plus recurrence, not a contraction.
//
// Inputs arrive as pre-converted 32-bit WORDS (built once per batch by k_prep_reads / k_seed and
// then re-used by every DP that touches the base):
//     read word  rw = (byte << 9) | (4*qual) << 16      pad rows past the read end: '0' / 64*4 (align.c:223-226)
//     hap  word  hw = (byte << 9) | (4*gapopen) << 16
// so that entering a base into the sliding 8-lane windows is one v_perm/v_alignbit per vector.
//
// Differences in FORM (not in values) from the SSE formulation, each verified against the oracle:
//  * bases are held pre-shifted (byte << 9) so that "mismatch ? qual : 0" is
//    v_pk_min_u16(hap ^ read, qual4): the XOR of two different 7-bit bytes is >= 512 > 4*127.
//  * only min(M,I) and D of each parity and the next step's "I before + nucprior" are carried between steps
//    (M and I are transient): one packed min less per register per half-step and < 80 VGPRs.
//  * I of the odd parity is computed as shift_down(min(I1+ge, M1+go_even)) + np: one lane shift
//    instead of two (go_odd[k] == go_even[k+1]).
//  * HAS_N = false drops the "haplotype base is N -> cost 0" vector (align.c:175-178,385) when the
//    haplotype holds no 'N' at all (a per-haplotype flag computed while it is staged in LDS).
//  * the last extra step reads the haplotype word one past the slice instead of 'N'
//    (align.c:376): that lane only feeds cells x >= len1, which never reach the result.
//  * values are held RE-BIASED: v' = v + 0x8000 (mod 2^16), so the reference's "zero" -0x8000 is 0, pos_inf 0x7800 is 0xF800
//    and every signed 16-bit min is an unsigned one.  The map is an order-preserving bijection that commutes with
//    wrapping addition, so all values correspond one to one, wrap-arounds included.
//  * SWAR = true adds both 16-bit halves with ONE 32-bit v_add_u32 (a full-rate VOP2 op; v_pk_add_u16 issues at half
//    rate, profiles/r01_valu_issue_rates.txt).  That is the same result unless a low half carries out, i.e. unless the
//    reference's own 16-bit add wraps.  Every finite value of the DP is at most 4*(sum of the read's qualities) + 1100
//    (the cost of the all-mismatch path on the cell's diagonal, plus one gap open / extend / nucprior / substitution
//    pending in a sum), and values derived from pos_inf stay within 0xF800 + 1100: the kernel picks SWAR only for
//    reads whose quality sum is at most DP_SWAR_MAX_QSUM, for which no add can reach 0x10000.  All other reads take the
//    packed adds, which wrap exactly like _mm_add_epi16.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace plat {

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, a) + __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ uint32_t pk_min_i(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b)));
}
__device__ __forceinline__ uint32_t pk_min_u(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ uint32_t splat16(uint32_t v) { return (v & 0xFFFFu) | (v << 16); }

// 8 x int16 lanes: lane k and lane k + 4 share a VGPR, v[k & 3] = lane k (k < 4, low half) | lane k + 4 (high half).
// A one-lane shift then moves whole registers -- v[j] <- v[j -+ 1] is a renaming in unrolled code -- and only the register that
// receives the fill value and the lane crossing from one half to the other is BUILT, with one v_perm_b32 (round 4; rounds 1-3 kept
// lanes 2j, 2j+1 together and paid four v_alignbit_b32 per shift, 24 half-rate instructions of the 118 per step).
struct V8 { uint32_t v[4]; };

// __builtin_amdgcn_perm(a, b, sel): result byte i = byte (sel_i & 3) of b for sel_i in 0..3, of a for sel_i in 4..7
// lane k <- lane k-1, lane 0 <- low half of fill
__device__ __forceinline__ void shift_up(V8& a, uint32_t fill) {
    const uint32_t n0 = __builtin_amdgcn_perm(a.v[3], fill, 0x05040100u);       // fill.lo | lane 3 << 16
    a.v[3] = a.v[2]; a.v[2] = a.v[1]; a.v[1] = a.v[0]; a.v[0] = n0;
}
// lane k <- lane k-1, lane 0 <- HIGH half of fill
__device__ __forceinline__ void shift_up_hi(V8& a, uint32_t fill) {
    const uint32_t n0 = __builtin_amdgcn_perm(a.v[3], fill, 0x05040302u);       // fill.hi | lane 3 << 16
    a.v[3] = a.v[2]; a.v[2] = a.v[1]; a.v[1] = a.v[0]; a.v[0] = n0;
}
// lane k <- lane k+1, lane 7 <- low half of fill
__device__ __forceinline__ void shift_down(V8& a, uint32_t fill) {
    const uint32_t n3 = __builtin_amdgcn_perm(fill, a.v[0], 0x05040302u);       // lane 4 | fill.lo << 16
    a.v[0] = a.v[1]; a.v[1] = a.v[2]; a.v[2] = a.v[3]; a.v[3] = n3;
}
// lane k <- lane k+1, lane 7 <- HIGH half of fill
__device__ __forceinline__ void shift_down_hi(V8& a, uint32_t fill) {
    const uint32_t n3 = __builtin_amdgcn_perm(fill, a.v[0], 0x07060302u);       // lane 4 | fill.hi << 16
    a.v[0] = a.v[1]; a.v[1] = a.v[2]; a.v[2] = a.v[3]; a.v[3] = n3;
}

constexpr uint32_t INF16 = 0x7800u;          // pos_inf, align.c:97 (as a cost; haplotype-N masks, unpacked/traceback variants)
constexpr uint32_t INF2 = 0x78007800u;
constexpr uint32_t NEG2 = 0x80008000u;       // -0x8000 in both halves (free start, align.c:249)
constexpr uint32_t INFB = 0xF800u;           // pos_inf re-biased (+0x8000)
constexpr uint32_t INFB2 = 0xF800F800u;
constexpr int DP_SWAR_MAX_QSUM = 15000;      // 4*15000 + 1100 < 0xF800 - 0x200
constexpr uint32_t CODE_N = ((uint32_t)'N') << 9;

__device__ __forceinline__ uint32_t code9(uint32_t byte) { return (byte & 0x7Fu) << 9; }
__device__ __forceinline__ uint32_t read_word(uint32_t base, uint32_t qual) { return code9(base) | ((qual * 4u) << 16); }
__device__ __forceinline__ uint32_t hap_word(uint32_t base, uint32_t gapopen) { return code9(base) | ((gapopen * 4u) << 16); }
constexpr uint32_t READ_PAD_WORD = ((((uint32_t)'0') & 0x7Fu) << 9) | ((64u * 4u) << 16);   // align.c:224-225

template <bool HAS_N, bool SWAR = false>
struct DP {
    // carried between steps: min(M,I) and D of both parities, and un = min(I2 + ge, M2 + go) = the even I of the NEXT
    // step before "+ nucprior" (the odd half-step's gap-open window is the next step's even window).  M and I
    // themselves are transient.  All of them re-biased (see the header of this file).
    V8 mi1, d1, mi2, d2, un, i2p, s1w, s1n, gop, gopn, s2w, q2w;
    uint32_t GE, NP, GENP, FILLI;
    unsigned minscore;
    // SWAR (no add of this read can wrap, see the header): min(a, b) + np == min(a + np, b + np), so "+ nucprior" is folded into
    // the two sums of the insertion state -- un / U are carried WITH it (gopn = gop + np rides along as one more window) and the
    // two adds per register and step that applied it are gone.
    static constexpr bool FOLD = SWAR;

    static __device__ __forceinline__ uint32_t add(uint32_t a, uint32_t b) { return SWAR ? a + b : pk_add(a, b); }

    // hw[0..7]: haplotype words of slice positions 0..7 (align.c:157,181)
    __device__ __forceinline__ void init(const uint32_t (&hw)[8], int gapextend, int nucprior) {
        GE = splat16((uint32_t)(gapextend * 4));
        NP = splat16((uint32_t)(nucprior * 4));
        GENP = GE + NP;
        FILLI = FOLD ? INFB : (INFB - (uint32_t)(nucprior * 4)) & 0xFFFFu;    // (+ NP ==) pos_inf (align.c:483)
        minscore = INFB;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            d1.v[j] = d2.v[j] = mi1.v[j] = mi2.v[j] = i2p.v[j] = INFB2;
            s2w.v[j] = 0x01FF01FFu;   // never equals a base code; XOR with any code is >= 511
            q2w.v[j] = 0x01000100u;   // 64*4, align.c:159
            s1w.v[j] = (hw[j] & 0xFFFFu) | (hw[j + 4] << 16);                  // lanes j and j + 4
            gop.v[j] = (hw[j] >> 16) | (hw[j + 4] & 0xFFFF0000u);
            if (HAS_N)
                s1n.v[j] = ((hw[j] & 0xFFFFu) == CODE_N ? 0u : INF16) |
                           ((hw[j + 4] & 0xFFFFu) == CODE_N ? 0u : (INF16 << 16));
            if (FOLD) gopn.v[j] = gop.v[j] + NP;
            un.v[j] = FOLD ? pk_min_u(add(INFB2, GENP), add(INFB2, gopn.v[j]))
                           : pk_min_u(add(INFB2, GE), add(INFB2, gop.v[j]));  // i2 = m2 = pos_inf before step 0
        }
    }

    // One full step h (even half-step then odd half-step), align.c:199-516.
    //  rw : read word entering lane 0 (seq2[h], qual2[h]);  hw : haplotype word entering lane 7 (slice pos 8+h)
    //  FL >= 0 : h < 8  -> free start on lane FL = h (align.c:244-250)
    //  EL >= 0 : h >= len2 -> candidate final score in lane EL = h - len2 (align.c:261-288,416-443)
    template <int FL, int EL>
    __device__ __forceinline__ void step(uint32_t rw, uint32_t hw, bool ext_rt = false) {
        V8 S, T, U, m1, i1;
        // ---------------- even half-step
        shift_up(s2w, rw);
        shift_up_hi(q2w, rw);
        if (FL >= 0) {
            constexpr uint32_t msk = (FL & 4) ? 0xFFFF0000u : 0x0000FFFFu;
            constexpr int j = (FL >= 0 ? FL : 0) & 3;
            mi1.v[j] = mi1.v[j] & ~msk;                          // -0x8000 re-biased = 0
            mi2.v[j] = mi2.v[j] & ~msk;
            // the forced m2 also feeds this step's I (align.c:331-335): redo that lane of un with m2 = -0x8000
            const uint32_t uf = FOLD ? pk_min_u(add(i2p.v[j], GENP), gopn.v[j]) : pk_min_u(add(i2p.v[j], GE), gop.v[j]);
            un.v[j] = (un.v[j] & ~msk) | (uf & msk);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) S.v[j] = pk_min_u(mi1.v[j], d1.v[j]);
        if (EL >= 0) take<(EL >= 0 ? EL : 0)>(S);
        else if (FL == 7) { if (ext_rt) take<0>(S); }          // len2 == 7: h = 7 is both the last forced and the first extra step
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t c = pk_min_u(s1w.v[j] ^ s2w.v[j], q2w.v[j]);
            if (HAS_N) c = pk_min_u(c, s1n.v[j]);
            m1.v[j] = add(S.v[j], c);
            i1.v[j] = FOLD ? un.v[j] : add(un.v[j], NP);
        }
        // gap-open vector of the odd half-step (== srli(gap_open) of the even one, lane 7 unused)
        V8 gopE = FOLD ? gopn : gop;                                             // (the one U adds: gop + np when folded)
        shift_down_hi(gop, hw);
        if (FOLD) shift_down_hi(gopn, hw + (NP & 0xFFFF0000u));
#pragma unroll
        for (int j = 0; j < 4; ++j)
            T.v[j] = pk_min_u(add(d2.v[j], GE), add(mi2.v[j], gop.v[j]));
        shift_up(T, INFB);
        d1 = T;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            mi1.v[j] = pk_min_u(m1.v[j], i1.v[j]);
            U.v[j] = pk_min_u(add(i1.v[j], FOLD ? GENP : GE), add(m1.v[j], gopE.v[j]));      // -> i2 after shift
        }
        // ---------------- odd half-step
        shift_down(s1w, hw);
        if (HAS_N) shift_down(s1n, (hw & 0xFFFFu) == CODE_N ? 0u : INF16);
#pragma unroll
        for (int j = 0; j < 4; ++j) S.v[j] = pk_min_u(mi2.v[j], d2.v[j]);
        if (EL >= 0) take<(EL >= 0 ? EL : 0)>(S);
        else if (FL == 7) { if (ext_rt) take<0>(S); }
        shift_down(U, FILLI);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t c = pk_min_u(s1w.v[j] ^ s2w.v[j], q2w.v[j]);
            if (HAS_N) c = pk_min_u(c, s1n.v[j]);
            const uint32_t m2 = add(S.v[j], c);
            const uint32_t i2 = FOLD ? U.v[j] : add(U.v[j], NP);
            d2.v[j] = pk_min_u(add(d1.v[j], GE), add(mi1.v[j], gop.v[j]));
            mi2.v[j] = pk_min_u(m2, i2);
            un.v[j] = FOLD ? pk_min_u(add(i2, GENP), add(m2, gopn.v[j]))
                           : pk_min_u(add(i2, GE), add(m2, gop.v[j]));        // next step's even I before + np (folded: with it)
            if (FL >= 0) i2p.v[j] = i2;
        }
    }

    template <int E>
    __device__ __forceinline__ void take(const V8& S) {
        const uint32_t r = S.v[E & 3];
        const unsigned sc = (E & 4) ? (r >> 16) : (r & 0xFFFFu);
        minscore = sc < minscore ? sc : minscore;
    }

    __device__ __forceinline__ int result() const { return (int)(minscore >> 2); }   // (minscore + 0x8000) >> 2, align.c:520
};

// ---- words built on the fly from the caller's bytes ---------------------------------------------------------------------
// (round 3: the DP no longer reads a pre-converted tile -- 12 % of the pairs reach it, the other reads' words were built and
//  written for nothing.)  Eight bases of a read / eight haplotype positions arrive as two 64-bit loads each (bases, qualities or
//  gap-open penalties; any alignment), and one word costs two instructions: v_perm_b32 puts byte k of the two sources into the
//  two 16-bit halves, v_pk_lshlrev_b16 shifts them by 9 and by 2 -- (base << 9) | (4 * quality) << 16.
__device__ __forceinline__ uint64_t load_u64_unaligned(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
template <int K>
__device__ __forceinline__ uint32_t word_of_bytes(uint32_t base4, uint32_t cost4) {     // K = 0..3: byte K of both
    const uint32_t t = __builtin_amdgcn_perm(cost4, base4, 0x0c000c00u | ((4u + K) << 16) | (uint32_t)K);   // base | cost << 16
    return __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, t) << (u16x2){9, 2}));
}
__device__ __forceinline__ void words_of_8(uint64_t bases, uint64_t costs, uint32_t (&out)[8]) {
    const uint32_t b0 = (uint32_t)bases, b1 = (uint32_t)(bases >> 32), c0 = (uint32_t)costs, c1 = (uint32_t)(costs >> 32);
    out[0] = word_of_bytes<0>(b0, c0); out[1] = word_of_bytes<1>(b0, c0); out[2] = word_of_bytes<2>(b0, c0); out[3] = word_of_bytes<3>(b0, c0);
    out[4] = word_of_bytes<0>(b1, c1); out[5] = word_of_bytes<1>(b1, c1); out[6] = word_of_bytes<2>(b1, c1); out[7] = word_of_bytes<3>(b1, c1);
}

// The 8 forced steps h = 0..7 and the 8 extra steps h = len2..len2+7 are fully unrolled with
// compile-time lane indices; RW(h) / HW(h) are callables returning the read / haplotype word of step h, RAWR(h) / RAWW(h)
// the bytes of the eight steps h..h+7 of the main loop (all of them inside the read).
struct Raw8 { uint64_t a, b; };                  // eight bases + eight qualities (or gap-open penalties) as they lie in memory
struct Raw16 { uint64_t a0, a1, b0, b1; };       // sixteen of each
typedef uint32_t u32x4_unaligned __attribute__((ext_vector_type(4), aligned(1)));
__device__ __forceinline__ void load_16_unaligned(const uint8_t* p, uint64_t& lo, uint64_t& hi) {
    const u32x4_unaligned v = *(const u32x4_unaligned*)p;                  // one global_load_dwordx4
    lo = v.x | ((uint64_t)v.y << 32); hi = v.z | ((uint64_t)v.w << 32);
}
template <bool HAS_N, bool SWAR, class RW, class HW, class RAWR, class RAWW, class RAW16R, class RAW16W>
__device__ __forceinline__ int dp_run8(DP<HAS_N, SWAR>& dp, int len2, RW rw, HW hw, RAWR raw_r, RAWW raw_w, RAW16R raw16_r, RAW16W raw16_w)
{
    const bool l7 = (len2 == 7);
    {
        uint32_t r[8], w[8];
        if (len2 >= 8) { const Raw8 x = raw_r(0), y = raw_w(0); words_of_8(x.a, x.b, r); words_of_8(y.a, y.b, w); }
        else {
#pragma unroll
            for (int k = 0; k < 8; ++k) { r[k] = rw(k); w[k] = hw(k); }
        }
        dp.template step<0, -1>(r[0], w[0]);
        dp.template step<1, -1>(r[1], w[1]);
        dp.template step<2, -1>(r[2], w[2]);
        dp.template step<3, -1>(r[3], w[3]);
        dp.template step<4, -1>(r[4], w[4]);
        dp.template step<5, -1>(r[5], w[5]);
        dp.template step<6, -1>(r[6], w[6]);
        dp.template step<7, -1>(r[7], w[7], l7);
    }
    int h = 8;
    if (h + 15 < len2) {
        // 16 steps per trip (the loop-carried register copies are paid once per trip): a lane's bytes lie where no other lane's do, so
        // a wave's load touches 64 cache lines whatever its width -- 16 bytes per array and trip keep the texture addresser off the
        // critical path.  The bytes of the NEXT trip are requested before this trip's steps.
        Raw16 nr = raw16_r(h), nw = raw16_w(h);
        for (; h + 15 < len2; h += 16) {
            const Raw16 cr = nr, cw = nw;
            if (h + 31 < len2) { nr = raw16_r(h + 16); nw = raw16_w(h + 16); }
            {
                uint32_t r[8], w[8];
                words_of_8(cr.a0, cr.b0, r); words_of_8(cw.a0, cw.b0, w);
#pragma unroll
                for (int k = 0; k < 8; ++k) dp.template step<-1, -1>(r[k], w[k]);
            }
            {
                uint32_t r[8], w[8];
                words_of_8(cr.a1, cr.b1, r); words_of_8(cw.a1, cw.b1, w);
#pragma unroll
                for (int k = 0; k < 8; ++k) dp.template step<-1, -1>(r[k], w[k]);
            }
        }
    }
    if (h + 7 < len2) {
        const Raw8 x = raw_r(h), y = raw_w(h);
        uint32_t r[8], w[8];
        words_of_8(x.a, x.b, r); words_of_8(y.a, y.b, w);
#pragma unroll
        for (int k = 0; k < 8; ++k) dp.template step<-1, -1>(r[k], w[k]);
        h += 8;
    }
    for (; h < len2; ++h) dp.template step<-1, -1>(rw(h), hw(h));
    // h == max(len2, 8) here
    if (!l7) { dp.template step<-1, 0>(rw(h), hw(h)); ++h; }
    dp.template step<-1, 1>(rw(h), hw(h)); ++h;
    dp.template step<-1, 2>(rw(h), hw(h)); ++h;
    dp.template step<-1, 3>(rw(h), hw(h)); ++h;
    dp.template step<-1, 4>(rw(h), hw(h)); ++h;
    dp.template step<-1, 5>(rw(h), hw(h)); ++h;
    dp.template step<-1, 6>(rw(h), hw(h)); ++h;
    dp.template step<-1, 7>(rw(h), hw(h));
    return dp.result();
}

template <bool HAS_N, bool SWAR, class RW, class HW>
__device__ __forceinline__ int dp_run(DP<HAS_N, SWAR>& dp, int len2, RW rw, HW hw)
{
    const bool l7 = (len2 == 7);
    dp.template step<0, -1>(rw(0), hw(0));
    dp.template step<1, -1>(rw(1), hw(1));
    dp.template step<2, -1>(rw(2), hw(2));
    dp.template step<3, -1>(rw(3), hw(3));
    dp.template step<4, -1>(rw(4), hw(4));
    dp.template step<5, -1>(rw(5), hw(5));
    dp.template step<6, -1>(rw(6), hw(6));
    dp.template step<7, -1>(rw(7), hw(7), l7);
    int h = 8;
    for (; h + 7 < len2; h += 8) {               // 8 steps per trip: the loop-carried register copies are paid once per trip
        uint32_t r[8], w[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { r[k] = rw(h + k); w[k] = hw(h + k); }
#pragma unroll
        for (int k = 0; k < 8; ++k) dp.template step<-1, -1>(r[k], w[k]);
    }
    for (; h < len2; ++h) dp.template step<-1, -1>(rw(h), hw(h));
    // h == max(len2, 8) here
    if (!l7) { dp.template step<-1, 0>(rw(h), hw(h)); ++h; }
    dp.template step<-1, 1>(rw(h), hw(h)); ++h;
    dp.template step<-1, 2>(rw(h), hw(h)); ++h;
    dp.template step<-1, 3>(rw(h), hw(h)); ++h;
    dp.template step<-1, 4>(rw(h), hw(h)); ++h;
    dp.template step<-1, 5>(rw(h), hw(h)); ++h;
    dp.template step<-1, 6>(rw(h), hw(h)); ++h;
    dp.template step<-1, 7>(rw(h), hw(h));
    return dp.result();
}

// DP over raw byte rows (plat_dp_batch, a1): words are built on the fly, pads as align.c:223-226,376,387.
__device__ __forceinline__ int dp_score_bytes(const uint8_t* __restrict__ hap, const uint8_t* __restrict__ go,
                                              const uint8_t* __restrict__ read, const uint8_t* __restrict__ qual,
                                              int len2, int gapextend, int nucprior)
{
    const int len1 = len2 + 15;
    DP<true, false> dp;
    uint32_t w0[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) w0[k] = hap_word(hap[k], go[k]);
    dp.init(w0, gapextend, nucprior);
    auto rw = [&](int h) -> uint32_t { return h < len2 ? read_word(read[h], qual[h]) : READ_PAD_WORD; };
    auto hw = [&](int h) -> uint32_t {
        return 8 + h < len1 ? hap_word(hap[8 + h], go[8 + h]) : hap_word((uint32_t)'N', go[len1 - 1]);
    };
    return dp_run<true, false>(dp, len2, rw, hw);
}

}  // namespace plat
