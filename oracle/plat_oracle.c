/*
 * plat_oracle.c -- CPU restatement of the Platypus read->haplotype likelihood and
 * local-assembly hot path.
 *
 * THIS FILE IS TEST INFRASTRUCTURE.  It is the parity oracle for the HIP path in
 * platypus_amd/csrc.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product path never does (it fails loudly
 * when the HIP library is missing).
 *
 * Every function cites the reference file:line it restates (paths relative to
 * the reference tree, andyrimmer/Platypus v0.8.1.1).  Nothing here is copied from
 * the reference: the DP is written from the behavioural description in
 * SURVEY.md App. A (8 x int16 lanes, wrapping adds, signed mins), as plain
 * scalar C over int16_t[8] arrays -- no SSE intrinsics.
 *
 * Pinning (see tests/golden/README.md):
 *   orc_dp_*          pinned against the UNMODIFIED reference src/c/align.c built
 *                     into oracle/_ref/libalign_ref.so (tests/test_oracle.py)
 *                     and against tests/golden/dp_cases.npz.
 *   orc_map_align,    pinned against golden vectors generated in this container
 *   orc_assemble      from the reference's own Cython sources (tests/golden/README.md).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------
 * a1: fastAlignmentRoutine  (src/c/align.c:77-586)
 * ------------------------------------------------------------------------------------------ */

#define POS_INF ((int16_t)0x7800)

static inline int16_t w16(int v) { return (int16_t)(uint16_t)v; }          /* wrapping narrow */
static inline int16_t add16(int16_t a, int16_t b) { return w16((int)a + (int)b); }
static inline int16_t min16(int16_t a, int16_t b) { return a < b ? a : b; }

/* lane k <- lane k-1, lane 0 <- fill   (the reference's _mm_slli_si128(v,2) + insert lane 0) */
static inline void shift_up(int16_t* v, int16_t fill) {
    for (int k = 7; k > 0; --k) v[k] = v[k - 1];
    v[0] = fill;
}
/* lane k <- lane k+1, lane 7 <- fill   (_mm_srli_si128(v,2) + insert lane 7) */
static inline void shift_down(int16_t* v, int16_t fill) {
    for (int k = 0; k < 7; ++k) v[k] = v[k + 1];
    v[7] = fill;
}

/*
 * Lane-exact restatement of align.c:94-586.  seq1 = haplotype slice (len2+15 bytes),
 * seq2/qual2 = read (len2 bytes), go = local gap-open slice (len2+15 bytes).
 * If aln1 != NULL the traceback (align.c:345-365,494-515,518-577) is also produced.
 * Returns the alignment score; -1 on allocation failure.
 */
ORC_API int orc_dp_align(const char* seq1, const char* seq2, const char* qual2, int len2,
                         int gapextend, int nucprior, const char* go,
                         char* aln1, char* aln2, int* firstpos)
{
    const int len1 = len2 + 15;                      /* align.c:88 */
    const int16_t GE = w16(gapextend * 4);           /* align.c:94 */
    const int16_t NP = w16(nucprior * 4);            /* align.c:95 */
    const int traceback = (aln1 != NULL);            /* align.c:96 */
    int16_t m1[8], i1[8], d1[8], m2[8], i2[8], d2[8];
    int16_t s1w[8], s1n[8], gop[8], s2w[8], q2w[8];
    int mask[8];
    int16_t (*bp)[8] = NULL;                         /* backpointers, align.c:126 */

    if (traceback) {
        bp = (int16_t (*)[8])malloc(sizeof(int16_t[8]) * 2 * (size_t)(len1 + 8));
        if (!bp) return -1;
    }
    for (int k = 0; k < 8; ++k) {
        m1[k] = i1[k] = d1[k] = m2[k] = i2[k] = d2[k] = POS_INF;      /* :139-144 */
        s1w[k] = (int16_t)seq1[k];                                    /* :157 */
        s2w[k] = POS_INF;                                             /* :158 */
        q2w[k] = 64 * 4;                                              /* :159 */
        s1n[k] = (s1w[k] == 'N') ? 0 : POS_INF;                       /* :175-178, n_score=0 :17 */
        gop[k] = w16(4 * (int)go[k]);                                 /* :181 */
        mask[k] = (k == 0);                                           /* :124 */
    }
    int16_t minscore = POS_INF;                                       /* :186 */
    int minscoreidx = -1;

    for (int h = 0; h < len2 + 8; ++h) {                              /* :199, s = 2h */
        int16_t t[8];
        /* ---- even half-step ------------------------------------------------ */
        if (h < len2) {                                               /* :218-226 */
            shift_up(s2w, (int16_t)seq2[h]);
            shift_up(q2w, w16(4 * (int)qual2[h]));
        } else {
            shift_up(s2w, (int16_t)'0');
            shift_up(q2w, 64 * 4);
        }
        for (int k = 0; k < 8; ++k)                                   /* :249-250 free start */
            if (mask[k]) { m1[k] = (int16_t)-0x8000; m2[k] = (int16_t)-0x8000; }
        for (int k = 0; k < 8; ++k) m1[k] = min16(m1[k], min16(i1[k], d1[k]));   /* :251 */
        if (h >= len2) {                                              /* :261-288 */
            int16_t sc = m1[h - len2];
            if (sc < minscore) { minscore = sc; minscoreidx = 2 * h; }
        }
        for (int k = 0; k < 8; ++k) {                                 /* :314-318 */
            int16_t sub = (s2w[k] == s1w[k]) ? 0 : q2w[k];
            m1[k] = add16(m1[k], min16(sub, s1n[k]));
        }
        for (int k = 0; k < 8; ++k) {                                 /* :320-324 */
            int16_t g = (k < 7) ? gop[k + 1] : 0;                     /* _mm_srli_si128(gap_open,2) */
            t[k] = min16(add16(d2[k], GE), add16(min16(m2[k], i2[k]), g));
        }
        for (int k = 7; k > 0; --k) d1[k] = t[k - 1];                 /* :326-329 */
        d1[0] = POS_INF;
        for (int k = 0; k < 8; ++k)                                   /* :331-335 */
            i1[k] = add16(min16(add16(i2[k], GE), add16(m2[k], gop[k])), NP);
        if (traceback) {                                              /* :345-365 */
            for (int k = 0; k < 8; ++k) {
                bp[2 * h][k] = w16((m1[k] & 3) | ((i1[k] & 3) << 2) | ((d1[k] & 3) << 6));
                m1[k] = w16(m1[k] & ~3);
                i1[k] = w16((i1[k] & ~3) | 1);
                d1[k] = w16((d1[k] & ~3) | 3);
            }
        }
        /* ---- odd half-step ------------------------------------------------- */
        {
            char c = (8 + h < len1) ? seq1[8 + h] : 'N';              /* :376 */
            int gi = (8 + h < len1) ? 8 + h : len1 - 1;               /* :387 */
            shift_down(s1w, (int16_t)c);                              /* :384 */
            shift_down(s1n, (c == 'N') ? 0 : POS_INF);                /* :385 */
            shift_down(gop, w16(4 * (int)go[gi]));                    /* :386-388 */
        }
        for (int k = 7; k > 0; --k) mask[k] = mask[k - 1];            /* :405-406 */
        mask[0] = 0;
        for (int k = 0; k < 8; ++k) m2[k] = min16(m2[k], min16(i2[k], d2[k]));   /* :407 */
        if (h >= len2) {                                              /* :416-443 */
            int16_t sc = m2[h - len2];
            if (sc < minscore) { minscore = sc; minscoreidx = 2 * h + 1; }
        }
        for (int k = 0; k < 8; ++k) {                                 /* :466-470 */
            int16_t sub = (s2w[k] == s1w[k]) ? 0 : q2w[k];
            m2[k] = add16(m2[k], min16(sub, s1n[k]));
        }
        for (int k = 0; k < 8; ++k)                                   /* :472-476 */
            d2[k] = min16(add16(d1[k], GE), add16(min16(m1[k], i1[k]), gop[k]));
        for (int k = 0; k < 7; ++k)                                   /* :478-484 */
            i2[k] = add16(min16(add16(i1[k + 1], GE), add16(m1[k + 1], gop[k])), NP);
        i2[7] = POS_INF;
        if (traceback) {                                              /* :494-515 */
            for (int k = 0; k < 8; ++k) {
                bp[2 * h + 1][k] = w16((m2[k] & 3) | ((i2[k] & 3) << 2) | ((d2[k] & 3) << 6));
                m2[k] = w16(m2[k] & ~3);
                i2[k] = w16((i2[k] & ~3) | 1);
                d2[k] = w16((d2[k] & ~3) | 3);
            }
        }
    }
    const int score = ((int)minscore + 0x8000) >> 2;                  /* :520,585 */
    if (!traceback) return score;

    /* Backtrace, align.c:523-577.  States: match 0, insert 1, delete 3. */
    {
        int s = minscoreidx;
        int i = s / 2 - len2;
        int y = len2;
        int x = s - y;
        int alnidx = 0;
        int state = (bp[s][i] >> 0) & 3;
        s -= 2;
        while (y > 0) {
            int newstate = (bp[s][i] >> (2 * state)) & 3;
            if (state == 0) {
                s -= 2;
                aln1[alnidx] = seq1[--x];
                aln2[alnidx] = seq2[--y];
            } else if (state == 1) {
                i += s & 1;
                s -= 1;
                aln1[alnidx] = '-';
                aln2[alnidx] = seq2[--y];
            } else {
                s -= 1;
                i -= s & 1;
                aln1[alnidx] = seq1[--x];
                aln2[alnidx] = '-';
            }
            state = newstate;
            alnidx++;
        }
        aln1[alnidx] = 0;
        aln2[alnidx] = 0;
        if (firstpos) *firstpos = x;
        for (int a = 0, b = alnidx - 1; a < b; ++a, --b) {
            char ta = aln1[a], tb = aln2[a];
            aln1[a] = aln1[b]; aln2[a] = aln2[b];
            aln1[b] = ta;      aln2[b] = tb;
        }
    }
    free(bp);
    return score;
}

/* score-only convenience */
ORC_API int orc_dp_score(const char* seq1, const char* seq2, const char* qual2, int len2,
                         int gapextend, int nucprior, const char* go)
{
    return orc_dp_align(seq1, seq2, qual2, len2, gapextend, nucprior, go, NULL, NULL, NULL);
}

/* batched score-only DP over padded rows (mirrors plat_dp_batch's layout) */
ORC_API void orc_dp_batch(int n, int lmax, const char* haps, const char* reads, const char* quals,
                          const char* gos, const int* len2, int gapextend, int nucprior, int* out)
{
    for (int j = 0; j < n; ++j)
        out[j] = orc_dp_score(haps + (size_t)j * (lmax + 15), reads + (size_t)j * lmax,
                              quals + (size_t)j * lmax, len2[j], gapextend, nucprior,
                              gos + (size_t)j * (lmax + 15));
}

/* ------------------------------------------------------------------------------------------
 * a2: calculateFlankScore  (src/c/align.c:593-644)
 * ------------------------------------------------------------------------------------------ */
ORC_API int orc_flank_score(int hapLen, int hapFlank, const char* quals, const char* go,
                            int gapextend, int nucprior, int firstpos,
                            const char* aln1, const char* aln2)
{
    char prev = 'M';
    int x = firstpos, y = 0, score = 0;
    for (int i = 0; aln1[i]; ++i) {
        char st = 'M';
        if (aln1[i] == '-') st = 'I';
        if (aln2[i] == '-') st = 'D';
        const int inflank = (x < hapFlank || x >= hapLen - hapFlank);
        if (st == 'M') {
            if (aln1[i] != aln2[i] && inflank) score += (aln1[i] == 'N') ? 0 : quals[y];
            ++x; ++y;
        } else if (st == 'I') {
            if (inflank) score += (prev == 'I') ? gapextend + nucprior : go[x - 1] + nucprior;
            ++y;
        } else {
            if (inflank) score += (prev == 'D') ? gapextend : go[x];
            ++x;
        }
        prev = st;
    }
    return score;
}

/* ------------------------------------------------------------------------------------------
 * a3-a5: 7-mer hashing  (src/cython/calign.pyx:61-165)
 * ------------------------------------------------------------------------------------------ */
#define HASH_NUCS 7
#define HASH_SIZE 16384

static inline unsigned base_code(char ch) {          /* calign.pyx:69-74 */
    int c = ch & 7;
    if (c == 7) c = 2;
    return (unsigned)(c & 3);
}

ORC_API unsigned orc_kmer_code(const char* seq) {     /* my_hash, calign.pyx:61-76 */
    unsigned h = 0;
    for (int i = 0; i < HASH_NUCS; ++i) h = (h << 2) + base_code(seq[i]);
    return h;
}

/* hash_sequence_multihit, calign.pyx:94-124.  table/next: HASH_SIZE shorts each, zeroed here. */
ORC_API void orc_hash_haplotype(const char* seq, int n, int16_t* table, int16_t* next)
{
    memset(table, 0, sizeof(int16_t) * HASH_SIZE);
    memset(next, 0, sizeof(int16_t) * HASH_SIZE);
    if (n < HASH_NUCS) return;
    for (int i = 0; i < n - HASH_NUCS; ++i) {
        unsigned h = orc_kmer_code(seq + i);
        int slot = i + 1;
        if (table[h] == 0) table[h] = (int16_t)slot;
        else {
            int j = table[h];
            while (next[j] != 0) j = next[j];
            next[j] = (int16_t)slot;
        }
    }
}

/* hashReadForMapping, calign.pyx:155-165.  out: rlen-7 shorts. */
ORC_API void orc_hash_read(const char* seq, int rlen, int16_t* out)
{
    for (int i = 0; i < rlen - HASH_NUCS; ++i) out[i] = (int16_t)orc_kmer_code(seq + i);
}

/* ------------------------------------------------------------------------------------------
 * a6: mapAndAlignReadToHaplotype  (src/cython/calign.pyx:170-272)
 *
 * The strncmp shortcut at calign.pyx:196 reads one byte before the haplotype buffer
 * (indexOfReadIntoHap is still -1 there); it is treated as never taken.
 * n_dp (optional) receives the number of fastAlignmentRoutine calls the reference makes.
 * ------------------------------------------------------------------------------------------ */
ORC_API int orc_map_align(const char* read, const char* quals, int readStart, int hapStart,
                          int readLen, int hapLen, const int16_t* hapHash, const int16_t* hapNext,
                          const int16_t* readHash, const char* hap, int gapExtend, int nucprior,
                          const char* go, int hapFlank, int doFlank, int* n_dp)
{
    if (n_dp) *n_dp = 0;
    if (readLen < HASH_NUCS) return 0;                                /* :179-180 */
    const int nCounts = hapLen + readLen;
    int* counts = (int*)calloc((size_t)nCounts + 1, sizeof(int));     /* :206 */
    char* aln1 = NULL; char* aln2 = NULL;
    int firstpos = 0, maxcount = 0, best = 1000000, bestPos = -1;
    if (hapFlank > 0) {                                               /* :199-202 */
        aln1 = (char*)malloc((size_t)2 * readLen + 16);
        aln2 = (char*)malloc((size_t)2 * readLen + 16);
    }
    for (int i = 0; i < readLen - HASH_NUCS; ++i) {                   /* :209-220 */
        int hidx = hapHash[(uint16_t)readHash[i]];
        while (hidx != 0) {
            int pos = hidx - i - 1;
            int c = ++counts[pos + readLen];
            if (c > maxcount) maxcount = c;
            hidx = hapNext[hidx];
        }
    }
    if (maxcount > 0) {                                               /* :222-247 */
        for (int j = 0; j < nCounts; ++j) {
            if (counts[j] != maxcount) continue;
            int idx = j - readLen;
            if (idx >= -readLen && idx + readLen + 15 < hapLen) {
                int st = idx - 8 > 0 ? idx - 8 : 0;
                int sc = orc_dp_align(hap + st, read, quals, readLen, gapExtend, nucprior,
                                      go + st, aln1, aln2, &firstpos);
                if (n_dp) ++*n_dp;
                if (doFlank == 1 && sc > 0 && hapFlank > 0)
                    sc -= orc_flank_score(hapLen, hapFlank, quals, go, gapExtend, nucprior,
                                          firstpos + st, aln1, aln2);
                if (sc < best) {
                    best = sc; bestPos = idx;
                    if (best == 0) { free(aln1); free(aln2); free(counts); return 0; }
                }
            }
        }
    }
    {                                                                 /* :252-267 */
        int idx = readStart - hapStart;
        if (hapLen - readLen - 15 < idx) idx = hapLen - readLen - 15;
        if (idx != bestPos) {
            int st = idx - 8 > 0 ? idx - 8 : 0;
            int sc = orc_dp_align(hap + st, read, quals, readLen, gapExtend, nucprior,
                                  go + st, aln1, aln2, &firstpos);
            if (n_dp) ++*n_dp;
            if (doFlank == 1 && sc > 0 && hapLen > 0)
                sc -= orc_flank_score(hapLen, hapFlank, quals, go, gapExtend, nucprior,
                                      firstpos + st, aln1, aln2);
            if (sc < best) best = sc;
        }
    }
    free(aln1); free(aln2); free(counts);
    return best;
}

/* ------------------------------------------------------------------------------------------
 * a7: Haplotype.annotateWithGapOpen  (src/cython/chaplotype.pyx:552-590; table :64-67)
 * The 49-entry table is homopolq[i]-'!' for the reference's per_base_indel_errors model;
 * tests/test_oracle.py recomputes it from the formula at chaplotype.pyx:64-67.
 * ------------------------------------------------------------------------------------------ */
static const signed char ORC_HOMOPOL_GO[50] = {
    45,42,41,39,37,32,28,23,20,19,17,16,15,14,13,12,11,11,10,9,9,8,8,7,7,7,6,6,6,5,5,5,
    4,4,4,3,3,3,3,2,2,2,2,2,1,1,1,1,1, /* NUL terminator of the byte string: */ -33 };

ORC_API void orc_gap_open(const char* seq, int hapLen, char* out /* hapLen+1 */)
{
    int homopol = -1, homopollen = 0;
    out[hapLen] = 0;
    for (int index = hapLen - 1; index >= 0; --index) {
        if ((int)seq[index] == homopol) {
            if (homopollen + 1 < 49) homopollen += 1;     /* errorModel[homopollen+1] != 0 */
        } else homopollen = 0;
        out[index] = (char)ORC_HOMOPOL_GO[homopollen];
        homopol = seq[index];
        if (homopol == 'N') homopol = 0;
    }
}

/* ------------------------------------------------------------------------------------------
 * a8: alignReadToHaplotype's score -> log-likelihood  (src/cython/chaplotype.pyx:594-676,
 * standard mode useMapQualCap=0).
 * ------------------------------------------------------------------------------------------ */
static const double mLTOT = -0.23025850929940459;                     /* chaplotype.pyx / calign.pyx:31 */

ORC_API double orc_loglik(int score, int mapq)
{
    double probMapRight = log(1.0 - exp(mLTOT * mapq));               /* :621 */
    double v = mLTOT * score + probMapRight;                          /* :676 */
    return v > -300.0 ? v : -300.0;
}

/* ------------------------------------------------------------------------------------------
 * a8-a10 for one (read, haplotype): hashes + gap-open are built here for convenience.
 * ------------------------------------------------------------------------------------------ */
ORC_API int orc_align_read_to_hap(const char* read, const char* quals, int readLen, int readStart,
                                  const char* hap, int hapLen, int hapStart, int hapFlank,
                                  int doFlank, int* n_dp)
{
    int16_t* table = (int16_t*)malloc(sizeof(int16_t) * HASH_SIZE);
    int16_t* next = (int16_t*)malloc(sizeof(int16_t) * HASH_SIZE);
    int16_t* rh = (int16_t*)malloc(sizeof(int16_t) * (size_t)(readLen > 7 ? readLen : 8));
    char* go = (char*)malloc((size_t)hapLen + 1);
    orc_hash_haplotype(hap, hapLen, table, next);
    if (readLen >= HASH_NUCS) orc_hash_read(read, readLen, rh);
    orc_gap_open(hap, hapLen, go);
    int sc = orc_map_align(read, quals, readStart, hapStart, readLen, hapLen, table, next, rh, hap,
                           3, 2, go, hapFlank, doFlank, n_dp);
    free(table); free(next); free(rh); free(go);
    return sc;
}

/* ------------------------------------------------------------------------------------------
 * a9: Haplotype.alignReads over a whole window  (src/cython/chaplotype.pyx:306-377)
 *
 * Reads are given as one concatenated list in the reference's order good -> bad -> broken
 * (chaplotype.pyx:341-373).  kind[r]: 0 good, 1 bad, 2 brokenMate.  flags = BAM bitFlag
 * (QCFail = 512, htslibWrapper.pxd:243).  out_ll[h*(nReads)+r]; the caller appends the 999
 * sentinel.  out_score (optional) gets the raw integer score (-1 for skipped reads).
 * ------------------------------------------------------------------------------------------ */
ORC_API void orc_align_window(int nHaps, const char* hapBlob, const int* hapOff, const int* hapLen,
                              int hapStartPos /* Haplotype.startPos */, int hapEndPos, int endBuffer,
                              int nReads, const char* seqBlob, const char* qualBlob, const int* readOff,
                              const int* readLen, const int* readPos, const int* readEnd,
                              const unsigned char* mapq, const int* flags, const unsigned char* kind,
                              int doFlank, double* out_ll, int* out_score, long long* n_dp_total)
{
    int16_t* table = (int16_t*)malloc(sizeof(int16_t) * HASH_SIZE);
    int16_t* next = (int16_t*)malloc(sizeof(int16_t) * HASH_SIZE);
    long long ndp = 0;
    for (int h = 0; h < nHaps; ++h) {
        const char* hap = hapBlob + hapOff[h];
        const int hl = hapLen[h];
        char* go = (char*)malloc((size_t)hl + 1);
        orc_hash_haplotype(hap, hl, table, next);
        orc_gap_open(hap, hl, go);
        for (int r = 0; r < nReads; ++r) {
            const size_t o = (size_t)h * nReads + r;
            int skip = 0;
            if (kind[r] != 2) {                                      /* :343-346, :358-361 */
                int os = hapStartPos > readPos[r] ? hapStartPos : readPos[r];
                int oe = hapEndPos < readEnd[r] ? hapEndPos : readEnd[r];
                int ov = oe > os ? oe - os : -1;                     /* chaplotype.pyx:103-115 */
                if ((flags[r] & 512) || ov < HASH_NUCS) skip = 1;
            }
            if (skip) { out_ll[o] = 0.0; if (out_score) out_score[o] = -1; continue; }
            const int rl = readLen[r];
            int16_t* rh = (int16_t*)malloc(sizeof(int16_t) * (size_t)(rl > 7 ? rl : 8));
            if (rl >= HASH_NUCS) orc_hash_read(seqBlob + readOff[r], rl, rh);
            int nd = 0;
            int sc = orc_map_align(seqBlob + readOff[r], qualBlob + readOff[r], readPos[r],
                                   hapStartPos - endBuffer, rl, hl, table, next, rh, hap, 3, 2, go,
                                   endBuffer, doFlank, &nd);
            ndp += nd;
            free(rh);
            out_ll[o] = orc_loglik(sc, mapq[r]);
            if (out_score) out_score[o] = sc;
        }
        free(go);
    }
    if (n_dp_total) *n_dp_total = ndp;
    free(table); free(next);
}

/* ------------------------------------------------------------------------------------------
 * a11: DiploidGenotype.calculateDataLikelihood  (src/cython/cgenotype.pyx:131-189)
 * arr1/arr2: per-read log-likelihood arrays terminated by 999; same_hap = (hap1 is hap2).
 * ------------------------------------------------------------------------------------------ */
ORC_API double orc_genotype_loglik(const double* arr1, const double* arr2, int same_hap,
                                   int totalReads, int nGoodReads, double* gof,
                                   double* hap1Like, double* hap2Like)
{
    const double log10E = 0.43429448190325182;      /* cgenotype.pyx:24 */
    const double logHalf = -0.69314718055994529;    /* cgenotype.pyx:28 */
    double likelihood = 0.0, g = 0.0, h1 = 0.0, h2 = 0.0;
    for (int r = 0; r <= totalReads; ++r) {
        double l1 = arr1[r], l2 = arr2[r];
        if (l1 == 999 && l2 == 999) break;
        double ll1 = log10E * l1, ll2 = log10E * l2;
        h1 += ll1; h2 += ll2;
        g += (ll1 > ll2 ? ll1 : ll2);
        if (same_hap) likelihood += l1;
        else if (fabs(l1 - l2) >= 3) likelihood += (logHalf + (l1 > l2 ? l1 : l2));
        else if (fabs(l1 - l2) <= 1e-3) likelihood += l1;
        else likelihood += log(0.5 * (exp(l1) + exp(l2)));
    }
    if (gof) *gof = nGoodReads > 0 ? (-10 * g) / nGoodReads : 0.0;
    if (hap1Like) *hap1Like = h1;
    if (hap2Like) *hap2Like = h2;
    return likelihood;
}

/* ------------------------------------------------------------------------------------------
 * a12: Population.setup for one individual  (src/cython/cpopulation.pyx:283-309)
 * ll: [nHaps][totalReads+1] (999-terminated rows).  Genotypes in the order of
 * generateAllGenotypesFromHaplotypeList (cgenotype.pyx:193-218): (i,j), i<=j.
 * out_gl[g] = rescaled likelihood, out_logl[g] = raw log-likelihood, out_gof[g].
 * ------------------------------------------------------------------------------------------ */
ORC_API void orc_population_setup_ind(int nHaps, const double* ll, int totalReads, int nGoodReads,
                                      double* out_logl, double* out_gl, double* out_gof)
{
    int g = 0;
    double maxLL = -1e7;                                              /* cpopulation.pyx:288 */
    const int stride = totalReads + 1;
    for (int a = 0; a < nHaps; ++a)
        for (int b = a; b < nHaps; ++b, ++g) {
            if (nGoodReads == 0) { out_logl[g] = 1.0; out_gl[g] = 1.0; out_gof[g] = 0.0; continue; }
            double L = orc_genotype_loglik(ll + (size_t)a * stride, ll + (size_t)b * stride, a == b,
                                           totalReads, nGoodReads, &out_gof[g], NULL, NULL);
            if (L > maxLL) maxLL = L;
            out_logl[g] = L;
        }
    if (nGoodReads != 0)
        for (int k = 0; k < g; ++k) {                                 /* :304-309 */
            double v = exp(out_logl[k] - maxLL);
            out_gl[k] = v > 1e-300 ? v : 1e-300;
        }
}

/* ------------------------------------------------------------------------------------------
 * EM haplotype frequencies, genotype calls, variant posteriors, per-position
 * genotype marginalisation.  Genotype g <-> haplotype pair (i,j), i<=j, in the order of
 * generateAllGenotypesFromHaplotypeList.  gl: [nInd][nGen] rescaled genotype likelihoods
 * (a12); nReads[i] = number of good reads of individual i (cpopulation.pyx:286-287).
 * Pinned by tests/golden/population_cases.json.gz (outputs of the reference's own method texts).
 * ------------------------------------------------------------------------------------------ */
static void hap_pair_of(int g, int nHap, int* a, int* b) {
    int i = 0, rowlen = nHap;
    while (g >= rowlen) { g -= rowlen; --rowlen; ++i; }
    *a = i; *b = i + g;
}

/* Population.EMiteration, cpopulation.pyx:384-457 */
static double em_iteration(int nInd, int nHap, int nGen, const int* nReads, const double* gl,
                           const int* hidx, double* freq, double* newFreqs, double* em)
{
    double maxChange = 0.0;
    int nIndWithData = 0;
    for (int i = 0; i < nInd; ++i) {
        if (nReads[i] == 0) continue;
        const double* L = gl + (size_t)i * nGen;
        double* csr = em + (size_t)i * nGen;
        double csrSum = 0.0;
        ++nIndWithData;
        for (int j = 0; j < nGen; ++j) {
            const int s = hidx[2 * j], r = hidx[2 * j + 1];
            const double thisCSR = L[j] * freq[s] * freq[r] * (1 + (r != s));       /* :421 */
            csr[j] = thisCSR;
            csrSum += thisCSR;
        }
        if (csrSum > 0.0)
            for (int j = 0; j < nGen; ++j) csr[j] /= csrSum;
    }
    for (int k = 0; k < nHap; ++k) newFreqs[k] = 0.0;
    for (int i = 0; i < nInd; ++i) {
        if (nReads[i] == 0) continue;
        const double* csr = em + (size_t)i * nGen;
        for (int j = 0; j < nGen; ++j) {
            newFreqs[hidx[2 * j]] += csr[j];                                        /* :443-444 */
            newFreqs[hidx[2 * j + 1]] += csr[j];
        }
    }
    for (int k = 0; k < nHap; ++k) {
        newFreqs[k] = newFreqs[k] / (2 * nIndWithData);
        const double freqChange = fabs(freq[k] - newFreqs[k]);
        if (freqChange > maxChange) maxChange = freqChange;
        freq[k] = newFreqs[k];
    }
    return maxChange;
}

/* Population.call (cpopulation.pyx:678-703) + callGenotypes (:623-676).
 * freq[nHap], em[nInd][nGen] (rows of individuals without reads are left as passed in), calls[nInd]
 * (-1 = None).  Returns the number of EM iterations; *maxChangeOut = last maximum frequency change. */
ORC_API int orc_em_call(int nInd, int nHap, const int* nReads, const double* gl, int maxIters, int useEM,
                        double* freq, double* em, int* calls, double* maxChangeOut)
{
    const int nGen = nHap * (nHap + 1) / 2;
    int* hidx = (int*)malloc(sizeof(int) * 2 * (size_t)nGen);
    double* newFreqs = (double*)malloc(sizeof(double) * (size_t)nHap);
    for (int g = 0; g < nGen; ++g) hap_pair_of(g, nHap, &hidx[2 * g], &hidx[2 * g + 1]);
    double eps = 1.0 / (nInd * 2 * 2);                                              /* :684 */
    if (1e-3 < eps) eps = 1e-3;
    double maxChange = eps + 1;
    const double uniformFreq = 1.0 / nHap;
    int iters = 0;
    for (int k = 0; k < nHap; ++k) freq[k] = uniformFreq;
    while (maxChange > eps && iters < maxIters) {                                   /* :700-702 */
        maxChange = em_iteration(nInd, nHap, nGen, nReads, gl, hidx, freq, newFreqs, em);
        ++iters;
    }
    for (int i = 0; i < nInd; ++i) {                                                /* callGenotypes */
        if (nReads[i] == 0) { calls[i] = -1; continue; }
        int best = -1;
        double maxL = 0.0;
        for (int g = 0; g < nGen; ++g) {
            const double v = useEM == 1 ? em[(size_t)i * nGen + g] : gl[(size_t)i * nGen + g];
            if (best == -1 || v > maxL) { maxL = v; best = g; }
        }
        calls[i] = best;
    }
    if (maxChangeOut) *maxChangeOut = maxChange;
    free(hidx); free(newFreqs);
    return iters;
}

/* Population.calculatePosterior, cpopulation.pyx:459-594.  hapHasVar[h] != 0 iff the variant is one
 * of haplotype h's variants; prior = var.calculatePrior(refFile) (or 0.5 for flatPrior).  Returns the
 * rounded phred-scaled posterior. */
ORC_API double orc_variant_posterior(int nInd, int nHap, const int* nReads, const double* gl, const double* freq,
                                     const unsigned char* hapHasVar, double prior)
{
    const int nGen = nHap * (nHap + 1) / 2;
    const int logOfMinFloat = -708;
    double* fprime = (double*)malloc(sizeof(double) * (size_t)nHap);
    double sumFreqs = 0.0, sumLogProbVariant = 0.0, sumLogProbNoVariant = 0.0;
    for (int i = 0; i < nHap; ++i) {                                                /* :509-518 */
        if (!hapHasVar[i]) { fprime[i] = freq[i]; sumFreqs += freq[i]; }
        else fprime[i] = 0.0;
    }
    if (sumFreqs > 0)
        for (int i = 0; i < nHap; ++i) fprime[i] /= sumFreqs;                       /* :527-534 */
    for (int i = 0; i < nInd; ++i) {
        if (nReads[i] == 0) continue;
        const double* L = gl + (size_t)i * nGen;
        double sumVar = 0.0, sumNoVar = 0.0;
        int g = 0;
        for (int r = 0; r < nHap; ++r)
            for (int s2 = r; s2 < nHap; ++s2, ++g) {
                const double factor = r != s2 ? 2.0 : 1.0;
                sumVar += (factor * freq[r] * freq[s2] * L[g]);                      /* :564 */
                sumNoVar += (factor * fprime[r] * fprime[s2] * L[g]);               /* :569 */
            }
        sumLogProbVariant += sumVar > 0 ? log(sumVar) : logOfMinFloat;
        sumLogProbNoVariant += sumNoVar > 0 ? log(sumNoVar) : logOfMinFloat;
    }
    free(fprime);
    double ratio = exp(sumLogProbNoVariant - sumLogProbVariant);                    /* :586 */
    if (!(ratio > 1e-300)) ratio = 1e-300;
    return round(-10.0 * (log10(ratio * (1.0 - prior)) - log10(prior + ratio * (1.0 - prior))));
}

/* computeGenotypeCallAndLikelihoods, vcfutils.pyx:163-334, for one sample.
 * varInHap: [nHap][nVar]; isRef[nHap]; gl, gof: this sample's [nGen] rows.
 * phased[2]; likelihoods[(nVar+1)(nVar+2)/2] in (index1, index2<=index1) order;
 * out4 = {bestLikelihood/sum, nonRefPosterior/sum, refPosterior/sum, bestGoodnessOfFitValue}. */
ORC_API void orc_genotype_call(int nHap, int nVar, int nIndividuals, const double* freq, const double* gl,
                               const double* gof, const int* varInHap, const int* isRef,
                               int* phased, double* likelihoods, double* out4)
{
    double sumLikelihoods = 0.0, bestGof = 1e6, bestLikelihood = -1.0, nonRefPosterior = 0.0, refPosterior = 0.0;
    double phasedMaxLike = -1e6;
    int phasedIndex1 = -1, phasedIndex2 = -1, nl = 0;
    for (int index1 = 0; index1 <= nVar; ++index1)
        for (int index2 = 0; index2 <= index1; ++index2) {
            double marginal = 0.0;
            int g = 0;
            for (int h1 = 0; h1 < nHap; ++h1)
                for (int h2 = h1; h2 < nHap; ++h2, ++g) {
                    const int ref1 = isRef[h1], ref2 = isRef[h2];
                    const double factor = h1 != h2 ? 2.0 : 1.0;
                    int matching = 0, v1h1 = 0, v1h2 = 0, v2h1 = 0, v2h2 = 0;
                    if (index1 == 0 && index2 == 0) {
                        if (ref1 && ref2) matching = 1;
                    } else if (index2 == 0) {
                        v1h1 = varInHap[h1 * nVar + index1 - 1]; v1h2 = varInHap[h2 * nVar + index1 - 1];
                        if ((ref2 && v1h1) || (ref1 && v1h2)) matching = 1;
                    } else {
                        v1h1 = varInHap[h1 * nVar + index1 - 1]; v1h2 = varInHap[h2 * nVar + index1 - 1];
                        v2h1 = varInHap[h1 * nVar + index2 - 1]; v2h2 = varInHap[h2 * nVar + index2 - 1];
                        if ((v1h1 && v2h2) || (v2h1 && v1h2)) matching = 1;
                    }
                    if (!matching) continue;
                    double cur;
                    if (nIndividuals > 25) cur = (factor * freq[h1] * freq[h2] * gl[g]);       /* :252-255 */
                    else cur = (factor * gl[g]);
                    marginal += cur;
                    if (cur > phasedMaxLike) {                                                    /* :260-303 */
                        phasedMaxLike = cur;
                        if (index1 == 0 && index2 == 0) { phasedIndex1 = index1; phasedIndex2 = index2; }
                        else if (index2 == 0 && index1 != 0) {
                            if (v1h1) { phasedIndex1 = index1; phasedIndex2 = index2; }
                            else if (v1h2) { phasedIndex1 = index2; phasedIndex2 = index1; }
                        } else if (index2 == index1 && index1 > 0) { phasedIndex1 = index1; phasedIndex2 = index2; }
                        else if (index2 > 0 && index1 > 0 && index2 != index1) {
                            if (v1h1 && v2h2) { phasedIndex1 = index1; phasedIndex2 = index2; }
                            else if (v1h2 && v2h1) { phasedIndex1 = index2; phasedIndex2 = index1; }
                        }
                    }
                    if (gof[g] < bestGof) bestGof = gof[g];
                }
            if (marginal > bestLikelihood) bestLikelihood = marginal;
            if ((index1 == 1 && index2 == 0) || (index1 == 1 && index2 == 1)) nonRefPosterior += marginal;
            else if (index1 == 0 && index2 == 0) refPosterior += marginal;
            sumLikelihoods += marginal;
            likelihoods[nl++] = marginal;
        }
    phased[0] = phasedIndex1; phased[1] = phasedIndex2;
    out4[0] = bestLikelihood / sumLikelihoods;
    out4[1] = nonRefPosterior / sumLikelihoods;
    out4[2] = refPosterior / sumLikelihoods;
    out4[3] = bestGof;
}

/* ------------------------------------------------------------------------------------------
 * VariantCandidateGenerator (src/cython/variant.pyx:459-751) -- variant candidates
 * from the CIGARs and mismatches of a set of reads.  One record per candidate occurrence, in the
 * reference's emission order; merging equal variants (addVariantToList, :510-527) is left to the
 * caller.  ref = contig[refSeqStart .. refSeqStart+refLen), contigLen = FastaFile SeqLength.
 * cigar: (op,len) int16 pairs per read, cig_off[r] .. cig_off[r+1] pairs.
 * rec: 6 ints per record {pos, nRemoved, nAdded, removed offset in ref (or -1), added offset in the
 * read blob (or -1), read index}.  Returns the number of records, or -1 if maxRec was too small, or -2
 * if a deletion reaches outside the reference window that was handed over.
 * ------------------------------------------------------------------------------------------ */
typedef struct { int* rec; int n, max, err; } CandOut;

static void cand_emit(CandOut* o, int pos, int nrem, int nadd, int remOff, int addOff, int read) {
    if (o->n >= o->max) { o->err = -1; return; }
    int* r = o->rec + 6 * (size_t)o->n++;
    r[0] = pos < 0 ? 0 : pos;                                          /* Variant.__init__: max(0, refPos), variant.pyx:118 */
    r[1] = nrem; r[2] = nadd; r[3] = remOff; r[4] = addOff; r[5] = read;
}

/* getSnpCandidatesFromReadSegment, variant.pyx:529-612 */
static void cand_snps(CandOut* o, const char* refSeq, int refSeqStart, const char* readSeq, const char* readQual,
                      long long seqBlobOff, int rlen, int readStart, int readOffset, int refOffset, int lenSeqToCheck,
                      int minFlank, int minBaseQual, int read)
{
    int msr = -1, mer = -1, msd = -1, med = -1;       /* misMatchStartRef/EndRef/StartRead/EndRead */
    for (int index = 0; index < lenSeqToCheck; ++index) {
        if (readOffset == 0 && index < minFlank) continue;
        if (index + readOffset >= rlen - minFlank) continue;
        const int readIndex = index + readOffset;
        const int refIndex = (index + refOffset + readStart) - refSeqStart;
        const char readChar = readSeq[readIndex], refChar = refSeq[refIndex];
        const int baseQual = readQual[readIndex];
        if (readChar != refChar) {
            if (readChar != 'N' && refChar != 'N' && baseQual >= minBaseQual) {
                if (msr == -1) { msr = mer = refIndex; msd = med = readIndex; }
                else if (refIndex - mer <= minFlank) { mer = refIndex; med = readIndex; }
                else {
                    cand_emit(o, msr + refSeqStart, mer - msr + 1, med - msd + 1, msr, (int)(seqBlobOff + msd), read);
                    msr = mer = refIndex; msd = med = readIndex;
                }
            }
        } else if (msr != -1 && refIndex - mer > minFlank) {
            cand_emit(o, msr + refSeqStart, mer - msr + 1, med - msd + 1, msr, (int)(seqBlobOff + msd), read);
            msr = mer = msd = med = -1;
        }
    }
    if (msr != -1) cand_emit(o, msr + refSeqStart, mer - msr + 1, med - msd + 1, msr, (int)(seqBlobOff + msd), read);
}

static int count_n(const char* s, int n) { int c = 0; for (int i = 0; i < n; ++i) c += s[i] == 'N'; return c; }

ORC_API int orc_variant_candidates(const char* ref, int refLen, int refSeqStart, int contigLen,
                                   int nReads, const char* seq, const char* qual, const long long* read_off,
                                   const int* pos, const int* flags, const short* cigar, const int* cig_off,
                                   int minFlank, int minBaseQual, int genSNPs, int genIndels,
                                   int* rec, int maxRec)
{
    CandOut o = { rec, 0, maxRec, 0 };
    for (int r = 0; r < nReads; ++r) {                                 /* addCandidatesFromReads, :722-743 */
        if (flags[r] & 512) continue;                                  /* Read_IsQCFail */
        const char* readSeq = seq + read_off[r];
        const char* readQual = qual + read_off[r];
        const int rlen = (int)(read_off[r + 1] - read_off[r]);
        const int readStart = pos[r];
        const short* ops = cigar + 2 * (size_t)cig_off[r];
        const int cigarLength = cig_off[r + 1] - cig_off[r];
        int refOffset = 0, readOffset = 0;
        for (int ci = 0; ci < cigarLength; ++ci) {                     /* getVariantCandidatesFromSingleRead, :614-720 */
            const int flag = ops[2 * ci], length = ops[2 * ci + 1];
            if (flag == 1 || flag == 2) {                              /* insertion / deletion */
                int flanked = 0;
                if (ci > 0 && ops[2 * ci - 2] == 0 && ops[2 * ci - 1] >= minFlank) flanked = 1;
                else if (ci < cigarLength - 1 && ops[2 * ci + 2] == 0 && ops[2 * ci + 3] >= minFlank) flanked = 1;
                if (flag == 1) {
                    if (flanked && count_n(readSeq + readOffset, length) == 0 && genIndels)
                        cand_emit(&o, readStart + refOffset - 1, 0, length, -1, (int)(read_off[r] + readOffset), r);
                    readOffset += length;
                } else {
                    if (flanked) {
                        /* refFile.getSequence(rname, a, a+length): clamped to [0, contigLen-1], fastafile.pyx:186-187 */
                        int a = readStart + refOffset, b = a + length;
                        if (a < 0) a = 0;
                        if (b > contigLen - 1) b = contigLen - 1;
                        const int n = b > a ? b - a : 0;
                        if (a - refSeqStart < 0 || a - refSeqStart + n > refLen) o.err = -2;
                        else if (count_n(ref + (a - refSeqStart), n) == 0 && genIndels)
                            cand_emit(&o, readStart + refOffset - 1, n, 0, a - refSeqStart, -1, r);
                    }
                    refOffset += length;
                }
            } else if (flag == 0 || flag == 7 || flag == 8) {          /* M, =, X */
                if (!(flag == 7 || (length < minFlank && flag == 0)) && genSNPs)
                    cand_snps(&o, ref, refSeqStart, readSeq, readQual, read_off[r], rlen, readStart, readOffset, refOffset,
                              length, minFlank, minBaseQual, r);
                readOffset += length;
                refOffset += length;
            } else if (flag == 3) {                                    /* N: skipped reference */
                refOffset += length;
            } else if (flag == 4) {                                    /* soft clip */
                readOffset += length;
                if (ci == 0) refOffset += length;
            }                                                          /* H, P, others: nothing */
        }
    }
    return o.err ? o.err : o.n;
}

/* ------------------------------------------------------------------------------------------
 * Read QC / trimming: checkAndTrimRead (src/cython/cwindow.pyx:332-481) for one stream of reads
 * (bamReadBuffer.addReadToBuffer, :560-595: `theLastRead` is the previous read of the stream).
 * qual is modified in place (trimmed bases get quality 0), flags get BAM_FQCFAIL where the reference
 * sets it; ok[r] = return value; reason[r] = filter type that rejected the read (0..6 as
 * cwindow.pyx:40-46, 7 = secondary alignment, -1 = accepted).  enabled[4]: MATE_UNMAPPED, MATE_DISTANT,
 * SMALL_INSERT, DUPLICATE filters on/off (a counter of -1 switches the filter off in the reference).
 * ------------------------------------------------------------------------------------------ */
ORC_API void orc_check_and_trim(int nReads, char* qual, const long long* read_off, const int* pos, const unsigned char* mapq,
                                int* flags, const short* chromID, const short* mateChromID, const int* insertSize,
                                const int* matePos, const short* cigar, const int* cig_off,
                                int minGoodQualBases, int minMapQual, int minBaseQual, int trimOverlapping, int trimAdapter,
                                int trimReadFlank, int trimSoftClipped, const int* enabled, int* ok, int* reason)
{
    for (int r = 0; r < nReads; ++r) {
        char* q = qual + read_off[r];
        const int rlen = (int)(read_off[r + 1] - read_off[r]);
        const int f = flags[r];
        const int paired = f & 1, proper = f & 2, unmapped = f & 4, mateUnmapped = f & 8, reverse = f & 16, mateReverse = f & 32;
        ok[r] = 0; reason[r] = -1;
        if (f & 256) { flags[r] |= 512; reason[r] = 7; continue; }                               /* :337-339 */
        if (mapq[r] < minMapQual) { flags[r] |= 512; reason[r] = 6; continue; }                  /* :341-344 */
        int nBelow = 0;
        for (int i = 0; i < rlen; ++i) nBelow += q[i] < minBaseQual;
        if (rlen - nBelow < minGoodQualBases) { flags[r] |= 512; reason[r] = 0; continue; }      /* :354-357 */
        if (unmapped) { flags[r] |= 512; reason[r] = 1; continue; }                              /* :360-363 */
        if (enabled[0] && paired && mateUnmapped) { reason[r] = 2; continue; }                   /* :367-371 (no QCFail) */
        if (enabled[1] && paired && (chromID[r] != mateChromID[r] || !proper)) { reason[r] = 3; continue; }
        if (enabled[2] && paired && insertSize[r] != 0 && abs(insertSize[r]) < rlen) { flags[r] |= 512; reason[r] = 4; continue; }
        if (enabled[3]) {                                                                        /* :389-409 */
            if (f & 1024) { flags[r] |= 512; reason[r] = 5; continue; }
            if (r > 0 && pos[r] == pos[r - 1] && rlen == (int)(read_off[r] - read_off[r - 1])) {
                if (paired) {
                    if (matePos[r - 1] == matePos[r]) { flags[r] |= 512; reason[r] = 5; continue; }
                } else { flags[r] |= 512; reason[r] = 5; continue; }
            }
        }
        if (!reverse) {                                                                          /* :415-421 */
            for (int i = 1; i <= rlen; ++i) {
                if (i < trimReadFlank || q[rlen - i] < 5) q[rlen - i] = 0; else break;
            }
        } else {
            for (int i = 0; i < rlen; ++i) {
                if (i < trimReadFlank || q[i] < 5) q[i] = 0; else break;
            }
        }
        const int absIns = abs(insertSize[r]);
        if (trimOverlapping == 1 && paired && absIns > 0 && !reverse && mateReverse && absIns < 2 * rlen) {   /* :438-440 */
            int lim = (2 * rlen - insertSize[r]) + 1;
            if (lim > rlen) lim = rlen;
            for (int i = 1; i <= lim; ++i) q[rlen - i] = 0;
        }
        if (trimAdapter == 1 && paired && absIns > 0 && absIns < rlen) {                          /* :445-452 */
            if (reverse) { for (int i = 1; i < rlen - absIns + 1; ++i) q[rlen - i] = 0; }
            else { for (int i = absIns; i < rlen; ++i) q[i] = 0; }
        }
        if (trimSoftClipped == 1) {                                                               /* :462-479 */
            int index = 0;
            for (int c = cig_off[r]; c < cig_off[r + 1]; ++c) {
                const int op = cigar[2 * c], len = cigar[2 * c + 1];
                if (op == 0 || op == 1) index += len;
                else if (op == 4) for (int j = 0; j < len; ++j) q[index++] = 0;
            }
        }
        ok[r] = 1;
    }
}

/* ------------------------------------------------------------------------------------------
 * SURVEY 8(f) rank 3 (read statistics of the VCF INFO field): the per-variant loop over the reads of a
 * window in vcfINFO (src/cython/vcfutils.pyx:1300-1390) with readOverlapsVariant (:901-913),
 * readQualIsGoodVariantPosition (:917-943) and variantSupportedByRead (:961-1072).
 * Reads: one table; sample i's good reads are [good_begin[i], good_end[i]), its bad reads
 * [bad_begin[i], bad_end[i]).  out[v*16 + k]: 0 TC, 1 TC_bad, 2 TR, 3 TC_ab, 4 TR_ab, 5 NR_sb, 6 NF_sb,
 * 7 TCR, 8 TCF, 9 TCR_sb, 10 TCF_sb, 11 NR, 12 NF, 13 nGoodReads, 14 nBadReads, 15 sum(mapq^2);
 * per_sample[(v*nInd + i)*2 + {0,1}] = nReadsThisSample, nVarReadsThisSample; minq[v*maxq + k] (k <
 * nminq[v]) = the MMLQ window minima in read order.
 * ------------------------------------------------------------------------------------------ */
static int read_overlaps_variant(int readStart, int readEnd, int vmin, int vmax) { return readStart <= vmax && readEnd > vmin; }

static int read_qual_good(const char* qual, int rlen, int readPos, int vmin, int vmax) {
    int a = vmin - readPos, b = vmax - readPos;
    if (a > rlen) a = rlen; if (a < 0) a = 0;
    if (b > rlen) b = rlen; if (b < 0) b = 0;
    for (int i = a; i != b; ++i) {                                     /* pointer walk qualStart != qualEnd, :931-941 */
        if (i > rlen + 64) return 1;                                   /* (a > b never happens: vmin <= vmax) */
        if (qual[i] < 5) return 0;
    }
    return 1;
}

static int variant_supported_by_read(const char* seq, int rlen, int readStart, const short* ops, int cigarLength,
                                     int varPos, int nAdded, int nRemoved, const char* added, int exactIndel)
{
    int refOffset = 0, readOffset = 0;
    for (int ci = 0; ci < cigarLength; ++ci) {
        const int flag = ops[2 * ci], length = ops[2 * ci + 1];
        if (flag == 1) {
            if (nAdded != nRemoved) {
                if (exactIndel) {
                    if (nAdded - nRemoved == length) {
                        /* pRead.seq[start:start+nAdded] == varAdded (Python slice, clipped at the NUL-terminated length) */
                        int n = nAdded;
                        if (readOffset + n > rlen) n = rlen - readOffset > 0 ? rlen - readOffset : 0;
                        if (n == nAdded && memcmp(seq + readOffset, added, (size_t)nAdded) == 0) return 1;
                    }
                    return 0;
                }
                return 1;
            }
            readOffset += length;
        } else if (flag == 2) {
            if (nAdded != nRemoved) {
                if (exactIndel) return nRemoved - nAdded == length;
                return 1;
            }
            refOffset += length;
        } else if (flag == 0 || flag == 7 || flag == 8) {
            const int start = varPos - readStart + readOffset - refOffset;
            if (refOffset + readStart <= varPos && refOffset + readStart + length > varPos && nAdded == nRemoved)
                if (start + nAdded <= rlen && start >= 0 && memcmp(seq + start, added, (size_t)nAdded) == 0) return 1;
            readOffset += length;
            refOffset += length;
        } else if (flag == 3) {
            readOffset += length;                                      /* (the reference advances both here, :1056-1058) */
            refOffset += length;
        } else if (flag == 4) {
            readOffset += length;
            if (ci == 0) refOffset += length;
        }
    }
    return 0;
}

ORC_API void orc_variant_read_stats(int nVars, const int* varPos, const int* bamMin, const int* bamMax, const int* nAdded,
                                    const int* nRemoved, const char* addedBlob, const int* addedOff,
                                    int nInd, const int* good_begin, const int* good_end, const int* bad_begin, const int* bad_end,
                                    const unsigned char* varInGenotype /* [nVars][nInd] */,
                                    const char* seq, const char* qual, const long long* read_off, const int* pos, const int* end,
                                    const unsigned char* mapq, const int* flags, const short* cigar, const int* cig_off,
                                    int minBaseQual, int badReadsWindow, int exactIndel,
                                    long long* out, int* per_sample, int* minq, int maxq, int* nminq)
{
    (void)minBaseQual;
    for (int v = 0; v < nVars; ++v) {
        long long* o = out + 16 * (size_t)v;
        for (int k = 0; k < 16; ++k) o[k] = 0;
        nminq[v] = 0;
        const int vmin = bamMin[v], vmax = bamMax[v];
        for (int i = 0; i < nInd; ++i) {
            const int inGt = varInGenotype[(size_t)v * nInd + i];
            int nReads = 0, nVarReads = 0;
            o[13] += good_end[i] - good_begin[i];
            o[14] += bad_end[i] - bad_begin[i];
            for (int r = bad_begin[i]; r < bad_end[i]; ++r) {          /* :1316-1333 */
                const int rlen = (int)(read_off[r + 1] - read_off[r]);
                if (!read_overlaps_variant(pos[r], end[r], vmin, vmax)) continue;
                if (!read_qual_good(qual + read_off[r], rlen, pos[r], vmin, vmax)) continue;
                o[1] += 1; o[15] += (long long)mapq[r] * mapq[r];
            }
            for (int r = good_begin[i]; r < good_end[i]; ++r) {        /* :1335-1387 */
                const int rlen = (int)(read_off[r + 1] - read_off[r]);
                const char* q = qual + read_off[r];
                if (!read_overlaps_variant(pos[r], end[r], vmin, vmax)) continue;
                if (!read_qual_good(q, rlen, pos[r], vmin, vmax)) continue;
                const int rev = (flags[r] & 16) != 0;
                ++nReads; o[0] += 1; o[15] += (long long)mapq[r] * mapq[r];
                if (inGt) { o[3] += 1; if (rev) o[9] += 1; else o[10] += 1; }
                if (rev) o[7] += 1; else o[8] += 1;
                if (variant_supported_by_read(seq + read_off[r], rlen, pos[r], cigar + 2 * (size_t)cig_off[r], cig_off[r + 1] - cig_off[r],
                                              varPos[v], nAdded[v], nRemoved[v], addedBlob + addedOff[v], exactIndel)) {
                    o[2] += 1; ++nVarReads;
                    if (inGt) { o[4] += 1; if (rev) o[5] += 1; else o[6] += 1; }
                    if (rev) o[11] += 1; else o[12] += 1;
                    if (inGt) {
                        int ws = vmin - pos[r] - (badReadsWindow - 1) / 2, we = vmax - pos[r] + (badReadsWindow - 1) / 2;
                        if (ws < 0) ws = 0;
                        if (we > rlen) we = rlen;
                        int m = 0;
                        for (int k = ws; k < we; ++k) m = (k == ws) ? q[k] : (q[k] < m ? q[k] : m);
                        if (nminq[v] < maxq) minq[(size_t)v * maxq + nminq[v]] = m;
                        nminq[v] += 1;
                    }
                }
            }
            per_sample[((size_t)v * nInd + i) * 2] = nReads;
            per_sample[((size_t)v * nInd + i) * 2 + 1] = nVarReads;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * a14-a18: coloured de-Bruijn assembler  (src/cython/assembler.pyx:73-1476)
 * ------------------------------------------------------------------------------------------ */
#define COL_REF 1
#define COL_READ 2
#define COL_BOTH 3

typedef struct {
    const char* seq;        /* k-mer bytes (points into ref / read blob), assembler.pyx:73-81 */
    int colours, position;
    int nEdges;
    int edgeEnd[4];
    double edgeW[4];
    double weight;
    char dfs;
    int hnext;              /* hash chain */
} ONode;

typedef struct {
    ONode* nodes; int nNodes, cap;      /* allNodes, insertion order (assembler.pyx:788) */
    int* heads; int nHeads; int k;
} OGraph;

static unsigned kmer_hash(const char* s, int k) {
    unsigned h = 2166136261u;
    for (int i = 0; i < k; ++i) { h ^= (unsigned char)s[i]; h *= 16777619u; }
    return h;
}

static OGraph* og_new(int k) {
    OGraph* g = (OGraph*)calloc(1, sizeof(OGraph));
    g->k = k; g->cap = 8192; g->nodes = (ONode*)malloc(sizeof(ONode) * g->cap);
    g->nHeads = 1 << 16; g->heads = (int*)malloc(sizeof(int) * g->nHeads);
    for (int i = 0; i < g->nHeads; ++i) g->heads[i] = -1;
    return g;
}
static void og_free(OGraph* g) { free(g->nodes); free(g->heads); free(g); }

/* DeBruijnGraph_InsertOrUpdateNode + NodeDict_FindOrInsert (assembler.pyx:668-797):
 * identity = exact k bytes; new -> appended to allNodes; existing -> colours |=, weight += */
static int og_touch(OGraph* g, const char* seq, int colour, int position, double weight) {
    unsigned b = kmer_hash(seq, g->k) & (unsigned)(g->nHeads - 1);
    for (int n = g->heads[b]; n >= 0; n = g->nodes[n].hnext)
        if (g->nodes[n].seq == seq || strncmp(g->nodes[n].seq, seq, (size_t)g->k) == 0) {
            g->nodes[n].colours |= colour;
            g->nodes[n].weight += weight;
            return n;
        }
    if (g->nNodes == g->cap) { g->cap *= 2; g->nodes = (ONode*)realloc(g->nodes, sizeof(ONode) * g->cap); }
    ONode* nd = &g->nodes[g->nNodes];
    memset(nd, 0, sizeof(*nd));
    nd->seq = seq; nd->colours = colour; nd->position = position; nd->weight = weight; nd->dfs = 'N';
    nd->hnext = g->heads[b]; g->heads[b] = g->nNodes;
    return g->nNodes++;
}

/* DeBruijnGraph_AddEdge, assembler.pyx:801-827 */
static void og_add_edge(OGraph* g, const char* s, const char* e, int colour, int ps, int pe, double w) {
    int a = og_touch(g, s, colour, ps, w);
    int b = og_touch(g, e, colour, pe, w);
    ONode* A = &g->nodes[a];
    for (int i = 0; i < 4; ++i) {
        if (i >= A->nEdges) { A->edgeEnd[i] = b; A->edgeW[i] = w; A->nEdges++; return; }
        if (A->edgeEnd[i] == b) { A->edgeW[i] += w; return; }
    }
    /* >4 distinct successors (non-ACGT bases only): silently dropped, :823-824 */
}

static void og_load(OGraph* g, const char* ref, int refLen, int refStart,
                    int nReads, const char* seqBlob, const char* qualBlob, const int* off,
                    const int* len, int minQual)
{
    const int k = g->k;
    for (int i = 0; i < refLen - k - 1; ++i)                          /* loadReferenceIntoGraph :1295-1319 */
        og_add_edge(g, ref + i, ref + i + 1, COL_REF, refStart + i, refStart + i + 1, 1.0);
    for (int r = 0; r < nReads; ++r) {                                /* loadReadIntoGraph :1348-1387 */
        const char* s = seqBlob + off[r]; const char* q = qualBlob + off[r];
        for (int i = 0; i < len[r] - k - 1; ++i) {
            int mq = 100000000, hasN = 0;
            for (int j = i; j < i + k + 1; ++j) { if (q[j] < mq) mq = q[j]; if (s[j] == 'N') hasN = 1; }
            if (mq >= minQual && !hasN) og_add_edge(g, s + i, s + i + 1, COL_READ, -1, -1, (double)mq);
        }
    }
}

/* dfsVisit / detectCyclesInGraph_Recursive, assembler.pyx:831-898 */
static int og_dfs(OGraph* g, int n, double minWeight) {
    ONode* nd = &g->nodes[n];
    nd->dfs = 'g';
    for (int i = 0; i < nd->nEdges; ++i) {
        ONode* e = &g->nodes[nd->edgeEnd[i]];
        if (e->colours == COL_READ && nd->edgeW[i] < minWeight) continue;
        if (e->dfs == 'w') { if (og_dfs(g, nd->edgeEnd[i], minWeight)) return 1; }
        else if (e->dfs == 'g') return 1;
    }
    nd->dfs = 'b';
    return 0;
}
static int og_has_cycle(OGraph* g, double minWeight) {
    for (int i = 0; i < g->nNodes; ++i) g->nodes[i].dfs = 'w';
    for (int i = 0; i < g->nNodes; ++i)
        if (g->nodes[i].dfs == 'w' && og_dfs(g, i, minWeight)) return 1;
    return 0;
}

typedef struct { int* n; int len; } OPath;
static OPath path_ext(const OPath* p, int node) {
    OPath q; q.len = (p ? p->len : 0) + 1; q.n = (int*)malloc(sizeof(int) * q.len);
    if (p) memcpy(q.n, p->n, sizeof(int) * p->len);
    q.n[q.len - 1] = node; return q;
}

typedef struct { int pos, nrem, nadd, type; int roff, aoff; int seq; } OVar;

/*
 * orc_assemble: assembleReadsAndDetectVariants (assembler.pyx:1429-1476) for reads already
 * selected/ordered as loadBAMDataIntoGraph does (:1391-1425; QCFail reads removed by the caller).
 * Outputs, sorted as sorted(theVars) (Variant.__richcmp__ '<', variant.pyx:282-363):
 *   out_pos[i], out_nrem[i], out_nadd[i], and out_blob holding removed||added per variant at
 *   out_off[i].  Returns the number of variants, or -(needed) if capacity is too small.
 * out_nnodes (optional) = node count of the final graph.
 */
ORC_API int orc_assemble(const char* ref, int refLen, int refStart, int assemStart, int assemEnd,
                         int nReads, const char* seqBlob, const char* qualBlob, const int* off,
                         const int* len, int kmerSize, int minQual, int minWeightI, int noCycles,
                         int maxVars, int* out_pos, int* out_nrem, int* out_nadd, int* out_off,
                         char* out_blob, int blobCap, int* out_nnodes)
{
    const double minWeight = (double)minWeightI;
    int k = kmerSize;
    OGraph* g = og_new(k);
    og_load(g, ref, refLen, refStart, nReads, seqBlob, qualBlob, off, len, minQual);
    int find = 1;
    if (noCycles) {                                                   /* :1453-1471 */
        while (og_has_cycle(g, minWeight)) {
            if (k > 50) { find = 0; break; }
            k += 5; og_free(g); g = og_new(k);
            og_load(g, ref, refLen, refStart, nReads, seqBlob, qualBlob, off, len, minQual);
        }
    }
    if (out_nnodes) *out_nnodes = g->nNodes;

    int nv = 0, vcap = 64, blobUsed = 0, blobNeed = 0;
    OVar* vars = (OVar*)malloc(sizeof(OVar) * vcap);
    char* tmpblob = NULL; int tmpcap = 0;

    /* findBubblesInGraph, assembler.pyx:1116-1177 */
    for (int ni = 0; find && ni < g->nNodes; ++ni) {
        ONode* nd = &g->nodes[ni];
        if (!(nd->colours == COL_BOTH && nd->position >= assemStart && nd->position < assemEnd)) continue;
        for (int ej = 0; ej < nd->nEdges; ++ej) {
            if (g->nodes[nd->edgeEnd[ej]].colours != COL_READ) continue;
            /* getVariantPathsThroughGraphFromNode, :1027-1112 */
            OPath stack[64]; int top = 0;
            OPath fin[64]; int nfin = 0;
            int aborted = 0;
            { OPath p0 = path_ext(NULL, ni); OPath p1 = path_ext(&p0, nd->edgeEnd[ej]); free(p0.n); stack[top++] = p1; }
            while (top > 0) {
                OPath p = stack[--top];
                if (top > 20 || nfin > 20) { free(p.n); aborted = 1; break; }          /* :1052-1057 */
                int cyc = 0;                                                           /* :999-1023 */
                for (int a = 0; a < p.len; ++a) g->nodes[p.n[a]].dfs = 'w';
                for (int a = 0; a < p.len; ++a) {
                    if (g->nodes[p.n[a]].dfs == 'w') g->nodes[p.n[a]].dfs = 'g'; else { cyc = 1; break; }
                }
                ONode* end = &g->nodes[p.n[p.len - 1]];
                if (cyc) { free(p.n); }
                else if (end->colours == COL_BOTH) { fin[nfin++] = p; }
                else if (end->colours == COL_REF) { free(p.n); }
                else {
                    for (int i = 0; i < end->nEdges; ++i) {                            /* :1091-1107 */
                        int ne = end->edgeEnd[i];
                        int c = g->nodes[ne].colours;
                        if (end->edgeW[i] >= minWeight || c == COL_BOTH || c == COL_REF)
                            stack[top++] = path_ext(&p, ne);
                    }
                    free(p.n);
                }
            }
            if (aborted) {
                for (int a = 0; a < top; ++a) free(stack[a].n);
                for (int a = 0; a < nfin; ++a) free(fin[a].n);
                continue;
            }
            for (int f = 0; f < nfin; ++f) {                          /* extractVarFromBubblePath :1196-1291 */
                OPath p = fin[f];
                int s = g->nodes[p.n[0]].position, t = g->nodes[p.n[p.len - 1]].position;
                if (t >= s) {
                    int rl = t - s + 1, al = p.len;
                    const char* r = ref + (s - refStart);
                    char* a = (char*)malloc((size_t)al + 1);
                    for (int i = 0; i < al; ++i) a[i] = g->nodes[p.n[i]].seq[0];
                    while (al > 0 && rl > 0 && r[rl - 1] == a[al - 1]) { --rl; --al; }   /* suffix first */
                    int ao = 0;
                    while (al > 0 && rl > 0 && r[0] == a[ao]) { ++r; ++ao; --rl; --al; ++s; }
                    if (nv == vcap) { vcap *= 2; vars = (OVar*)realloc(vars, sizeof(OVar) * vcap); }
                    if (blobUsed + rl + al > tmpcap) { tmpcap = (blobUsed + rl + al) * 2 + 256; tmpblob = (char*)realloc(tmpblob, (size_t)tmpcap); }
                    OVar* v = &vars[nv];
                    v->pos = s > 0 ? s : 0;                           /* variant.pyx:121 */
                    v->nrem = rl; v->nadd = al; v->roff = blobUsed; v->aoff = blobUsed + rl; v->seq = nv;
                    memcpy(tmpblob + blobUsed, r, (size_t)rl);
                    memcpy(tmpblob + blobUsed + rl, a + ao, (size_t)al);
                    blobUsed += rl + al;
                    if (rl == al) v->type = (al == 1) ? 0 : 1;        /* variant.pyx:136-144 */
                    else if (rl == 0) v->type = 2; else if (al == 0) v->type = 3; else v->type = 4;
                    ++nv;
                    free(a);
                }
                free(p.n);
            }
        }
    }
    og_free(g);

    /* stable insertion sort by (pos, type, nRemoved)  -- sorted(), assembler.pyx:1476 */
    for (int i = 1; i < nv; ++i) {
        OVar v = vars[i]; int j = i - 1;
        while (j >= 0 && (vars[j].pos > v.pos || (vars[j].pos == v.pos && (vars[j].type > v.type ||
               (vars[j].type == v.type && vars[j].nrem > v.nrem))))) { vars[j + 1] = vars[j]; --j; }
        vars[j + 1] = v;
    }
    blobNeed = blobUsed;
    int ret = nv;
    if (nv > maxVars || blobNeed > blobCap) ret = -(nv > 0 ? nv : 1);
    else {
        int o = 0;
        for (int i = 0; i < nv; ++i) {
            out_pos[i] = vars[i].pos; out_nrem[i] = vars[i].nrem; out_nadd[i] = vars[i].nadd; out_off[i] = o;
            memcpy(out_blob + o, tmpblob + vars[i].roff, (size_t)(vars[i].nrem + vars[i].nadd));
            o += vars[i].nrem + vars[i].nadd;
        }
    }
    free(vars); free(tmpblob);
    return ret;
}


/* ------------------------------------------------------------------------------------------
 * computeHaplotypeScore  (src/cython/vcfutils.pyx:1076-1114)
 * hapLike[h] = DiploidGenotype.hap1Like of haplotype h (orc_genotype_loglik's hap1Like for the
 * last individual with reads).  Sort the negated values; the first cluster ends at the first gap
 * > 20; if that gap is < 50 the second cluster (up to the next gap > 20) is added.
 * Pinned by tests/golden/vcf_cases.json.gz (HapScore of the reference's vcfINFO text).
 * ------------------------------------------------------------------------------------------ */
static int orc_cmp_double(const void* a, const void* b)
{
    const double x = *(const double*)a, y = *(const double*)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

ORC_API int orc_haplotype_score(int nHaps, const double* hapLike)
{
    if (nHaps <= 0) return 0;
    double* v = (double*)malloc((size_t)nHaps * sizeof(double));
    for (int h = 0; h < nHaps; ++h) v[h] = -hapLike[h];
    qsort(v, (size_t)nHaps, sizeof(double), orc_cmp_double);
    int sizes[2] = {1, 0}, nClusters = 1;
    double dist = 0;
    for (int i = 1; i < nHaps; ++i) {
        if (v[i] - v[i - 1] > 20) {
            if (nClusters == 1) dist = v[i] - v[i - 1];
            if (nClusters == 2) break;
            ++nClusters;
            sizes[1] = 1;
        } else
            ++sizes[nClusters - 1];
    }
    free(v);
    return sizes[0] + ((dist < 50 && dist > 0) ? sizes[1] : 0);
}
