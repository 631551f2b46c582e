// dp_core.hpp -- banded affine-gap DP of Platypus' fastAlignmentRoutine for gfx950 (CDNA4).
//
// Reference behaviour: src/c/align.c:77-586 (score-only mode, aln1 == NULL).  The reference keeps
// 8 x int16 lanes in one SSE register per DP state; here ONE GPU LANE owns ONE whole alignment and
// keeps the same 8 int16 lanes packed two-per-VGPR (4 VGPRs per state vector), so a wave64 advances
// 64 alignments per instruction with v_pk_add_u16 / v_pk_min_i16 / v_pk_min_u16 -- genuinely
// 16-bit wrapping arithmetic, i.e. the same values as _mm_add_epi16/_mm_min_epi16 by construction.
// The reference's lane shifts (_mm_slli/_mm_srli_si128 by one lane) become v_alignbit_b32 across
// the 4 VGPRs.  No MFMA: this is a min-plus recurrence, not a contraction.
//
// Differences in FORM (not in values) from the SSE formulation, each verified against the oracle:
//  * bases are held pre-shifted (byte << 9) so that "mismatch ? qual : 0" is
//    v_pk_min_u16(hap ^ read, qual4): the XOR of two different 7-bit bytes is >= 512 > 4*93.
//  * min(M,I) of each parity is cached (it is needed by the D update of the other parity and by
//    the next min(M,I,D) of its own), saving one packed min per register per half-step.
//  * I of the odd parity is computed as shift_down(min(I1+ge, M1+go_even)) + np: one lane shift
//    instead of two (go_odd[k] == go_even[k+1]).
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

// 8 x int16 lanes, lane k in v[k>>1], even lanes in the low half.
struct V8 { uint32_t v[4]; };

// lane k <- lane k-1, lane 0 <- fill (low 16 bits of fill)
__device__ __forceinline__ void shift_up(V8& a, uint32_t fill) {
    a.v[3] = __builtin_amdgcn_alignbit(a.v[3], a.v[2], 16);
    a.v[2] = __builtin_amdgcn_alignbit(a.v[2], a.v[1], 16);
    a.v[1] = __builtin_amdgcn_alignbit(a.v[1], a.v[0], 16);
    a.v[0] = (a.v[0] << 16) | (fill & 0xFFFFu);
}
// lane k <- lane k+1, lane 7 <- fill
__device__ __forceinline__ void shift_down(V8& a, uint32_t fill) {
    a.v[0] = __builtin_amdgcn_alignbit(a.v[1], a.v[0], 16);
    a.v[1] = __builtin_amdgcn_alignbit(a.v[2], a.v[1], 16);
    a.v[2] = __builtin_amdgcn_alignbit(a.v[3], a.v[2], 16);
    a.v[3] = (a.v[3] >> 16) | (fill << 16);
}

constexpr uint32_t INF16 = 0x7800u;          // pos_inf, align.c:97
constexpr uint32_t INF2 = 0x78007800u;
constexpr uint32_t NEG2 = 0x80008000u;       // -0x8000 in both halves (free start, align.c:249)

__device__ __forceinline__ uint32_t code9(uint32_t byte) { return (byte & 0x7Fu) << 9; }
__device__ __forceinline__ uint32_t nqual(uint32_t byte) { return byte == (uint32_t)'N' ? 0u : INF16; }

// Sequential reader of a byte string in global memory: one aligned dword per 4 bytes plus
// v_alignbyte for arbitrary start alignment.  Reads at most 8 bytes past the last byte consumed
// (hence PLAT_BLOB_PAD = 32).
struct ByteStream {
    const uint32_t* w;
    uint32_t cur, nxt, sh;
    __device__ __forceinline__ void init(const uint8_t* p) {
        sh = (uint32_t)((uintptr_t)p & 3u);
        w = (const uint32_t*)(p - sh);
        cur = w[0];
        nxt = w[1];
        w += 2;
    }
    __device__ __forceinline__ uint32_t next4() {
        uint32_t r = __builtin_amdgcn_alignbyte(nxt, cur, sh);
        cur = nxt;
        nxt = *w++;
        return r;
    }
};

struct DP {
    V8 m1, i1, d1, m2, i2, d2, mi1, mi2, s1w, s1n, gop, s2w, q2w;
    uint32_t GE, NP, FILLI;
    int minscore;

    __device__ __forceinline__ void init(uint32_t hap0, uint32_t hap1, uint32_t go0, uint32_t go1,
                                         int gapextend, int nucprior) {
        GE = splat16((uint32_t)(gapextend * 4));
        NP = splat16((uint32_t)(nucprior * 4));
        FILLI = (INF16 - (uint32_t)(nucprior * 4)) & 0xFFFFu;   // + NP == pos_inf (align.c:483)
        minscore = 0x7800;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            m1.v[j] = i1.v[j] = d1.v[j] = m2.v[j] = i2.v[j] = d2.v[j] = mi1.v[j] = mi2.v[j] = INF2;
            s2w.v[j] = 0x01FF01FFu;   // never equals a base code; XOR with any code is >= 511
            q2w.v[j] = 0x01000100u;   // 64*4, align.c:159
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            uint32_t hb = ((k < 4 ? hap0 : hap1) >> (8 * (k & 3))) & 0xFFu;
            uint32_t gb = ((k < 4 ? go0 : go1) >> (8 * (k & 3))) & 0xFFu;
            uint32_t sh = 16 * (k & 1);
            if ((k & 1) == 0) { s1w.v[k >> 1] = 0; s1n.v[k >> 1] = 0; gop.v[k >> 1] = 0; }
            s1w.v[k >> 1] |= code9(hb) << sh;
            s1n.v[k >> 1] |= nqual(hb) << sh;
            gop.v[k >> 1] |= ((gb * 4u) & 0xFFFFu) << sh;
        }
    }

    // One full step h (even half-step then odd half-step), align.c:199-516.
    //  rb/qb : read base / quality entering lane 0 (seq2[h], qual2[h]; '0'/64 past the read end)
    //  hb/gb : haplotype base / gap-open entering lane 7 (seq1[8+h], localgapopen[8+h])
    //  FORCE : h < 8  -> free start on lane h (align.c:244-250); lane index = fl
    //  EXTRACT: h >= len2 -> candidate final score in lane el = h - len2 (align.c:261-288,416-443)
    template <bool FORCE, bool EXTRACT>
    __device__ __forceinline__ void step(uint32_t rb, uint32_t qb, uint32_t hb, uint32_t gb, int fl, int el) {
        V8 S, T, U;
        // ---------------- even half-step
        shift_up(s2w, code9(rb));
        shift_up(q2w, qb * 4u);
        if (FORCE) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint32_t msk = (fl == 2 * j ? 0xFFFFu : 0u) | (fl == 2 * j + 1 ? 0xFFFF0000u : 0u);
                m1.v[j] = (m1.v[j] & ~msk) | (NEG2 & msk);
                m2.v[j] = (m2.v[j] & ~msk) | (NEG2 & msk);
                mi1.v[j] = (mi1.v[j] & ~msk) | (NEG2 & msk);
                mi2.v[j] = (mi2.v[j] & ~msk) | (NEG2 & msk);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) S.v[j] = pk_min_i(mi1.v[j], d1.v[j]);
        if (EXTRACT) take(S, el);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t c = pk_min_i(pk_min_u(s1w.v[j] ^ s2w.v[j], q2w.v[j]), s1n.v[j]);
            m1.v[j] = pk_add(S.v[j], c);
            U.v[j] = pk_min_i(pk_add(i2.v[j], GE), pk_add(m2.v[j], gop.v[j]));   // i1 before +NP
        }
        // gap-open vector of the odd half-step (== srli(gap_open) of the even one, lane 7 unused)
        V8 gopE = gop;
        shift_down(gop, gb * 4u);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            T.v[j] = pk_min_i(pk_add(d2.v[j], GE), pk_add(mi2.v[j], gop.v[j]));
        shift_up(T, INF16);
        d1 = T;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            i1.v[j] = pk_add(U.v[j], NP);
            mi1.v[j] = pk_min_i(m1.v[j], i1.v[j]);
            U.v[j] = pk_min_i(pk_add(i1.v[j], GE), pk_add(m1.v[j], gopE.v[j]));   // -> i2 after shift
        }
        // ---------------- odd half-step
        shift_down(s1w, code9(hb));
        shift_down(s1n, nqual(hb));
#pragma unroll
        for (int j = 0; j < 4; ++j) S.v[j] = pk_min_i(mi2.v[j], d2.v[j]);
        if (EXTRACT) take(S, el);
        shift_down(U, FILLI);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t c = pk_min_i(pk_min_u(s1w.v[j] ^ s2w.v[j], q2w.v[j]), s1n.v[j]);
            m2.v[j] = pk_add(S.v[j], c);
            d2.v[j] = pk_min_i(pk_add(d1.v[j], GE), pk_add(mi1.v[j], gop.v[j]));
            i2.v[j] = pk_add(U.v[j], NP);
            mi2.v[j] = pk_min_i(m2.v[j], i2.v[j]);
        }
    }

    __device__ __forceinline__ void take(const V8& S, int el) {
        uint32_t r = el < 4 ? (el < 2 ? S.v[0] : S.v[1]) : (el < 6 ? S.v[2] : S.v[3]);
        int sc = (int)(short)((el & 1) ? (r >> 16) : (r & 0xFFFFu));
        minscore = sc < minscore ? sc : minscore;
    }

    __device__ __forceinline__ int result() const { return (minscore + 0x8000) >> 2; }   // align.c:520
};

// Full DP for one alignment.  hap/go point at the slice start (len2+15 valid bytes each).
__device__ __forceinline__ int dp_score(const uint8_t* __restrict__ hap, const uint8_t* __restrict__ go,
                                        const uint8_t* __restrict__ read, const uint8_t* __restrict__ qual,
                                        int len2, int gapextend, int nucprior)
{
    const int len1 = len2 + 15;
    ByteStream sh, sg, sr, sq;
    sh.init(hap); sg.init(go); sr.init(read); sq.init(qual);
    DP dp;
    {
        uint32_t h0 = sh.next4(), h1 = sh.next4(), g0 = sg.next4(), g1 = sg.next4();
        dp.init(h0, h1, g0, g1, gapextend, nucprior);
    }
    const uint32_t golast = go[len1 - 1];
    int h = 0;
    if (len2 >= 8) {
        // prologue: h = 0..7, free start on lane h; all inputs in range
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            uint32_t R = sr.next4(), Q = sq.next4(), H = sh.next4(), G = sg.next4();
#pragma unroll
            for (int i = 0; i < 4; ++i)
                dp.step<true, false>((R >> (8 * i)) & 0xFFu, (Q >> (8 * i)) & 0xFFu, (H >> (8 * i)) & 0xFFu,
                                     (G >> (8 * i)) & 0xFFu, 4 * b + i, 0);
        }
        h = 8;
        // main loop: whole blocks of 4 steps with h+3 < len2 (all reads in range, no checks)
        const int nmain = (len2 >> 2) - 2;
        for (int b = 0; b < nmain; ++b) {
            uint32_t R = sr.next4(), Q = sq.next4(), H = sh.next4(), G = sg.next4();
#pragma unroll
            for (int i = 0; i < 4; ++i)
                dp.step<false, false>((R >> (8 * i)) & 0xFFu, (Q >> (8 * i)) & 0xFFu, (H >> (8 * i)) & 0xFFu,
                                      (G >> (8 * i)) & 0xFFu, 0, 0);
        }
        h += 4 * (nmain > 0 ? nmain : 0);
    }
    // generic tail: remaining in-read steps, then the 8 extra steps (align.c:199 "one extra iteration")
    {
        uint32_t R = 0, Q = 0, H = 0, G = 0;
        int i = 0;
        for (; h < len2 + 8; ++h, i = (i + 1) & 3) {
            if (i == 0) {
                if (h < len2) { R = sr.next4(); Q = sq.next4(); }
                H = sh.next4(); G = sg.next4();
            }
            uint32_t rb = (R >> (8 * i)) & 0xFFu, qb = (Q >> (8 * i)) & 0xFFu;
            uint32_t hb = (H >> (8 * i)) & 0xFFu, gb = (G >> (8 * i)) & 0xFFu;
            if (h >= len2) { rb = (uint32_t)'0'; qb = 64u; }                 // align.c:223-226
            if (8 + h >= len1) { hb = (uint32_t)'N'; gb = golast; }          // align.c:376,387
            const bool force = h < 8, ext = h >= len2;
            if (force) {
                if (ext) dp.step<true, true>(rb, qb, hb, gb, h, h - len2);
                else dp.step<true, false>(rb, qb, hb, gb, h, 0);
            } else {
                if (ext) dp.step<false, true>(rb, qb, hb, gb, 0, h - len2);
                else dp.step<false, false>(rb, qb, hb, gb, 0, 0);
            }
        }
    }
    return dp.result();
}

}  // namespace plat
