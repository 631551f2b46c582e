// plat_stageb.hip -- what callVariantsInRegion does between the candidate generator and Population.setup, on the device
// (SURVEY.md 8(f) ranks 1-2): sorted candidates -> leftNormaliseIndel -> filterVariants -> calling windows -> window pointers ->
// every valid combination of a window's variants as a haplotype -> haplotype bytes, sorted -> the window batch the likelihood
// kernels take.  One sample per region (the cohort case stays with the caller).  Reference:
//   variantcaller.pyx:456-470,523-531    sorted(getCandidates()), leftNormaliseIndel, sorted, filterVariants
//   variant.pyx:282-363                  Variant.__richcmp__: (refPos, varType, nRemoved)
//   platypusutils.pyx:806-931            leftNormaliseIndel
//   variantFilter.pyx:98-171             filterVariants
//   window.py:49-127,140-238             getBunchesOfVariants, WindowsAndVariants
//   cwindow.pyx:176-264,655-689          ReadArray.setWindowPointers / setWindowPointersBasedOnMatePos
//   variantFilter.pyx:377-441            getFilteredHaplotypes (the branch that enumerates), platypusutils.pyx:735-802 isHaplotypeValid
//   chaplotype.pyx:127-191,397-449       Haplotype.__init__, getMutatedSequence
//   variantcaller.pyx:325-383            mergeHaplotypes: sorted(haplotypes) (equal sequences are left to the caller)
//
// Seven small kernels, no host round trip between them (windows keep the order (region, window) in the batch):
//   k_sb_variants  one workgroup per region: the region's candidates in LDS; rank sort by the reference's key; one WAVE per indel
//                  walks the reference for its leftmost / rightmost placement (64 positions per step, ballot); second sort; runs of
//                  equal variants merged (supports summed); the filter; then one lane bunches the survivors into windows (the
//                  bunching is a sequential rule over ~100 positions, on LDS)
//   k_sb_windows   one workgroup per region, one thread per window: window pointers by binary search in the read table, the decision
//                  (call / skip / the caller's greedy filter), the valid combinations counted and their lengths summed; then the
//                  prefix sums of the region's windows
//   k_sb_haps_rank one workgroup per window: the bytes behind the haplotypes' common prefix staged in LDS (every lane finds the segment its
//                  byte comes from), lexicographic ranks by wave-wide comparisons, haplotypes with one sequence merged by prior product
//   k_sb_prefix    one workgroup per region: prefix sums of its windows' counts
//   k_sb_scan      one wave: exclusive scan over the REGIONS -> where each region's windows, haplotypes, reads, pairs and bytes go
//   k_sb_haps_write  one workgroup per window: every haplotype's bytes written once, in rank order
//   k_sb_reads     one wave per window: read indices, kinds and offsets of the window's reads
// Anything the reference would raise on, anything whose order depends on a Python dictionary and anything beyond the capacities is
// flagged for the caller (region or window) instead of being guessed.
#include <math.h>

#include "plat_internal.hpp"

namespace plat {

constexpr int SB_CAP = 1024;                      // candidates of a region held in LDS (more: the caller's own code)
constexpr int SB_THREADS = 1024;                  // k_sb_variants: 16 waves (16 indels walk the reference side by side)
constexpr int SB_DICT_CAP = 5400;               // distinct records of a scan whose dictionary fits 8192 slots (a resize at 5462 keys would need 32768)
constexpr int SB_MAXCOMB = 5;                     // a window with more variants than this goes through the greedy filter (the caller's)

struct SbIn {
    plat_stage_b_in b;
    plat_stage_b_options o;
    double pow01[16];                              // pow(0.1, k) of the HOST's libm (variant.pyx:243: the prior of a multi-nucleotide variant)
};

__device__ __forceinline__ int sb_type(int nrem, int nadd) {          // variant.pyx:49-53,127-140: SNP 0, MNP 1, INS 2, DEL 3, REP 4
    if (nrem == nadd) return nadd == 1 ? 0 : 1;
    if (nrem == 0) return 2;
    if (nadd == 0) return 3;
    return 4;
}

// exclusive scan over the threads of a workgroup (up to 16 waves); total in *tot (LDS scratch of 16 ints)
__device__ __forceinline__ int sb_block_scan(int v, int* wsum, int* tot) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
    if (lane == 63) wsum[w] = x;
    __syncthreads();
    int base = 0;
    for (int k = 0; k < w; ++k) base += wsum[k];
    if (threadIdx.x == blockDim.x - 1) *tot = base + x;
    __syncthreads();
    return base + x - v;
}

struct SbRegion {                                  // per-region LDS state of k_sb_variants
    unsigned long long key[SB_CAP];                // pos << 32 | type << 28 | nrem
    int id[SB_CAP];                                // first-record id (the dictionary's insertion order), then: rank in the first sort
    int pos[SB_CAP], nrem[SB_CAP], nadd[SB_CAP], supp[SB_CAP], remc[SB_CAP], addo[SB_CAP], bmin[SB_CAP], bmax[SB_CAP];
    unsigned short perm[SB_CAP];                   // sorted position -> element
    unsigned char head[SB_CAP], keep[SB_CAP];
    int list[SB_CAP];
    int wsum[16], tot, nIndel, addedUsed, status, nKept;
    // the Python-2 dictionaries of the candidate generator, replayed when an order depends on them (sb_dict_order)
    int dId[SB_DICT_CAP];                          // first-record ids of ALL distinct records of the scan, ascending = insertion order
    int dHash[SB_DICT_CAP];                        // hash(Variant) narrowed to a C int (variant.pxd:31); before that: scratch of the sort
    unsigned short dCand[SB_DICT_CAP];             // candidate index + 1 of the record (0: it did not pass the support filter)
    unsigned short dOrd[SB_DICT_CAP];              // keys in the order the dictionary yields them
    unsigned short dScr[SB_DICT_CAP];              // a table's keys in slot order while it is rebuilt
    unsigned short dP2[SB_CAP], dO2[SB_CAP], dS2[SB_CAP];   // the second dictionary: its keys (distinct-record indices), its order, rebuild scratch
    int dN, dN2;
};

// the added bases of element e: in the read table (addo >= 0) or, once normalised, in the region's own blob (addo = -(offset + 1))
__device__ __forceinline__ const uint8_t* sb_added(const SbRegion& R, int e, const uint8_t* read_seq, const uint8_t* blob) {
    const int a = R.addo[e];
    return a >= 0 ? read_seq + a : blob + (-(a + 1));
}
__device__ __forceinline__ bool sb_same(const SbRegion& R, int a, int b, const uint8_t* read_seq, const uint8_t* blob) {   // Variant.__richcmp__ == (variant.pyx:282-300)
    if (R.pos[a] != R.pos[b] || R.nrem[a] != R.nrem[b] || R.nadd[a] != R.nadd[b]) return false;
    const uint8_t* x = sb_added(R, a, read_seq, blob);
    const uint8_t* y = sb_added(R, b, read_seq, blob);
    for (int i = 0; i < R.nadd[a]; ++i) if (x[i] != y[i]) return false;
    return true;                                                       // (removed bases: the reference's own at pos, equal when pos and nrem are)
}

// ---- CPython 2.7 on the device: hash(Variant) and the iteration order of a dict (what variantcaller.pyx:456-470 walks) ---------------------
// hash(str), stringobject.c (64-bit build)
__device__ __forceinline__ unsigned long long sb_py2_string_hash(const uint8_t* p, int n) {
    if (n == 0) return 0ull;
    unsigned long long x = (unsigned long long)p[0] << 7;
    for (int i = 0; i < n; ++i) x = (1000003ull * x) ^ p[i];
    x ^= (unsigned long long)n;
    return x == ~0ull ? ~0ull - 1 : x;
}
// hash((refName, refPos, removed, added)) (tupleobject.c) kept in a C int (variant.pyx:270-280, variant.pxd:31): the dictionary probes
// with the sign-extended low 32 bits, -1 -> -2
__device__ __forceinline__ int sb_py2_variant_hash(unsigned long long nameHash, int refPos, const uint8_t* rem, int nrem, const uint8_t* add, int nadd) {
    const unsigned long long h[4] = {nameHash, (unsigned long long)(long long)refPos, sb_py2_string_hash(rem, nrem), sb_py2_string_hash(add, nadd)};
    unsigned long long x = 0x345678ull, mult = 1000003ull;
    for (int i = 0; i < 4; ++i) { x = (x ^ h[i]) * mult; mult += (unsigned long long)(82520ll + 2 * (3 - i)); }
    x += 97531ull;
    if (x == ~0ull) x = ~0ull - 1;
    int narrowed = (int)(unsigned)x;
    return narrowed == -1 ? -2 : narrowed;
}
// Iteration order of a Python-2 dict into which `n` distinct keys (0 .. n-1, hashes H[key] as the dictionary sees them) were inserted in
// that order (dictobject.c: a new key takes the first empty slot of its probe sequence i = 5 i + perturb + 1, perturb >>= 5; the table of
// 8 slots is rebuilt 4 x used slots large, in slot order, when two thirds full).  The WHOLE workgroup runs it, exactly:
//  * between two rebuilds the table has one size and the keys enter it in a known order -- the keys of the old table in its slot order,
//    then the new ones: key of priority p;
//  * the slot a key ends in is the first of its probe sequence that no key of HIGHER priority ends in.  So every key claims its current
//    slot with an atomicMin of its priority; who does not hold its slot afterwards moves one probe on; a key, once beaten for a slot, is
//    beaten for good (a slot only ever passes to higher priorities), so nobody has to go back.  Rounds repeat until nobody moves:
//    as many as the longest probe chain, a few dozen, instead of one insertion after the other;
//  * the slot order of the finished table (a scan) is the next table's insertion order.
// owner: 8192 words; step: one byte per key; ord: out (and the old table's order in between); tmp: n entries.  n <= SB_DICT_CAP.
__device__ void sb_py2_dict_order(int n, const int* __restrict__ H, unsigned* owner, unsigned char* step, unsigned short* ord, unsigned short* tmp,
                                  int* wsum, int* tot, int* flag)
{
    const int tid = threadIdx.x, nthr = blockDim.x;
    int size = 8, m = 0;                                               // m keys are in the table (ord[0..m): its slot order)
    for (;;) {
        const int lim = (2 * size + 2) / 3;                            // the insertion that makes used * 3 >= size * 2
        const int e1 = n < lim ? n : lim;                              // keys m .. e1-1 enter this table; priority p -> key: p < m ? ord[p] : p
        const unsigned long long mask = (unsigned long long)size - 1;
        for (int i = tid; i < size; i += nthr) owner[i] = 0xFFFFFFFFu;
        for (int p = tid; p < e1; p += nthr) step[p] = 0;
        __syncthreads();
        auto slotOf = [&](int p) -> unsigned {
            const int key = p < m ? (int)ord[p] : p;
            const unsigned long long h = (unsigned long long)(long long)H[key];
            unsigned long long i = h & mask, perturb = h;
            for (int t = step[p]; t > 0; --t) { i = 5 * i + perturb + 1; perturb >>= 5; }
            return (unsigned)(i & mask);
        };
        for (;;) {
            if (tid == 0) *flag = 0;
            // claim: a key walks its probe sequence past every slot a higher priority holds (beaten once, beaten for good) and takes the
            // first it can; whoever it displaces finds out below and walks on in the next round
            for (int p = tid; p < e1; p += nthr) {
                const int key = p < m ? (int)ord[p] : p;
                const unsigned long long h = (unsigned long long)(long long)H[key];
                unsigned long long i = h & mask, perturb = h;
                int t = step[p];
                for (int q = t; q > 0; --q) { i = 5 * i + perturb + 1; perturb >>= 5; }
                while (atomicMin(&owner[(unsigned)(i & mask)], (unsigned)p) < (unsigned)p) { i = 5 * i + perturb + 1; perturb >>= 5; ++t; }
                step[p] = (unsigned char)t;
            }
            __syncthreads();
            bool moved = false;
            for (int p = tid; p < e1; p += nthr) if (owner[slotOf(p)] != (unsigned)p) { step[p] = (unsigned char)(step[p] + 1); moved = true; }
            if (moved) *flag = 1;
            __syncthreads();
            if (!*flag) break;
            __syncthreads();
        }
        // the table's slot order
        {
            const int per = (size + nthr - 1) / nthr, s0 = tid * per, s1 = min(size, s0 + per);
            int cnt = 0;
            for (int i = s0; i < s1; ++i) cnt += owner[i] != 0xFFFFFFFFu;
            int at = sb_block_scan(cnt, wsum, tot);
            for (int i = s0; i < s1; ++i) if (owner[i] != 0xFFFFFFFFu) { const int p = (int)owner[i]; tmp[at++] = (unsigned short)(p < m ? (int)ord[p] : p); }
        }
        __syncthreads();
        for (int i = tid; i < e1; i += nthr) ord[i] = tmp[i];
        __syncthreads();
        m = e1;
        if (e1 * 3 >= size * 2) {                                      // rebuilt 4 x used large (also behind the LAST key: the final table is the new one)
            int ns = 8;
            while (ns <= e1 * 4) ns <<= 1;
            size = ns;
            continue;
        }
        if (e1 == n) break;
    }
}

__global__ void __launch_bounds__(1024)
k_sb_variants(SbIn in, plat_stage_b_out out, const int32_t* __restrict__ mtab)
{
    extern __shared__ __align__(16) unsigned char sb_lds[];
    SbRegion& R = *(SbRegion*)sb_lds;
    const plat_stage_b_in& b = in.b;
    const plat_stage_b_options& o = in.o;
    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int32_t* hdr = out.hdr + 8 * g;
    const int n = b.cand_n[2 * g];
#ifdef PLAT_SB_TIMING
    long long tm_[12]; int tmn_ = 0;
#define SB_MARK() do { if (tmn_ < 12) tm_[tmn_++] = wall_clock64(); } while (0)
    SB_MARK();
#else
#define SB_MARK() do { } while (0)
#endif
    if (tid == 0) { R.status = 0; R.nIndel = 0; R.addedUsed = 0; R.nKept = 0; R.dN2 = 0; }
    __syncthreads();
    if (b.cand_n[2 * g + 1] != 0 || n > SB_CAP || n > b.cap_per_scan) {                  // (the merge kernel's own verdict is the caller's to read)
        if (tid == 0) { hdr[0] = PLAT_SB_HOST; hdr[1] = hdr[2] = hdr[3] = hdr[4] = hdr[6] = hdr[7] = 0; hdr[5] = 5; }
        return;
    }
    const long long roff = b.ref_off[g];
    const int refLen = (int)(b.ref_off[g + 1] - roff), rss = b.ref_seq_start[g], contigLen = b.contig_len[g];
    const uint8_t* ref = b.ref_seq + roff;                             // ref[x - rss] = contig base x for rss <= x < rss + refLen
    uint8_t* blob = out.added + (long long)g * b.cap_added;
    const int rlen = b.region_rlen[g];
    int nk = 0;
    // Pass 0 orders candidates of one key by their first records (the order a dictionary yields them in unless two of its keys met in a
    // slot).  Where the result can depend on the real order -- see (a), (b) below -- the dictionaries are replayed (sb_replay) and pass 1
    // runs with their order as the tie-break, which is what the reference did in the first place.
    for (int pass = 0; pass < 2; ++pass) {
    long long nrec = 0;
    // ---- load; key of the reference's order
    for (int i = tid; i < n; i += SB_THREADS) {
        const int32_t* c = b.cand + 8ll * ((long long)g * b.cap_per_scan + i);
        const int pos = c[3] < 0 ? 0 : c[3], nrem = c[4], nadd = c[5];
        R.id[i] = pass ? (int)R.dOrd[i] : c[0]; R.supp[i] = c[1]; R.pos[i] = pos; R.nrem[i] = nrem; R.nadd[i] = nadd;
        R.remc[i] = nrem ? rss + c[6] - (int)roff : pos;             // contig coordinate of the removed bases (c[6]: offset in the reference blob)
        R.addo[i] = nadd ? c[7] : 0;
        R.bmin[i] = pos; R.bmax[i] = pos;
        R.key[i] = (unsigned long long)(unsigned)pos << 32 | (unsigned long long)sb_type(nrem, nadd) << 28 | (unsigned)(nrem & 0x0FFFFFFF);
        nrec += c[1];
        if (nrem > 0x0FFFFFF || nadd > 0xFFFF) atomicOr(&R.status, 1);
    }
    {   // candidate records of the region (a statistic of the caller)
        for (int d = 32; d; d >>= 1) nrec += __shfl_down(nrec, d, 64);
        __syncthreads();
        if (lane == 0) R.wsum[wv] = (int)nrec;
        __syncthreads();
        if (tid == 0) { int t = 0; for (int k = 0; k < SB_THREADS / 64; ++k) t += R.wsum[k]; hdr[3] = t; }
        __syncthreads();
    }
    SB_MARK();
    // ---- first sort: (refPos, varType, nRemoved), equal keys in the dictionary's insertion order (sorted() is stable)
    for (int i = tid; i < n; i += SB_THREADS) {
        const unsigned long long k = R.key[i]; const int id = R.id[i];
        int r = 0;
        for (int j = 0; j < n; ++j) { const unsigned long long kj = R.key[j]; r += (kj < k) || (kj == k && R.id[j] < id); }
        R.list[i] = r;
    }
    __syncthreads();
    for (int i = tid; i < n; i += SB_THREADS) R.id[i] = R.list[i];          // id := rank in the first sort (the tie-break of the second)
    __syncthreads();
    SB_MARK();
    // ---- leftNormaliseIndel: pure insertions / deletions at refPos >= 100
    for (int i = tid; i < n; i += SB_THREADS) {
        const int nrem = R.nrem[i], nadd = R.nadd[i];
        if (nrem != nadd && !(nrem > 0 && nadd > 0) && R.pos[i] >= 100) R.list[atomicAdd(&R.nIndel, 1)] = i;
    }
    __syncthreads();
    for (int q = wv; q < R.nIndel; q += SB_THREADS / 64) {                           // one wave per indel
        const int e = R.list[q];
        const int pos = R.pos[e], nrem = R.nrem[e], nadd = R.nadd[e];
        const int window = (nadd > nrem ? nadd : nrem) + rlen;
        const int seqMax = contigLen - 1;
        const int wmin = pos - window > 1 ? pos - window : 1, wmax = pos + window < seqMax ? pos + window : seqMax;
        const int Lr = wmax - wmin, cut = pos - wmin;
        // the reference window handed over must hold [wmin, wmax); an indel whose tail ref[cut + nrem + 1:] would be empty sits at the
        // contig's end: the caller's code
        if (wmin < rss || wmax > rss + refLen || Lr < 1 || cut + nrem + 1 >= Lr) { if (lane == 0) atomicOr(&R.status, 1); continue; }
        const uint8_t* rw = ref + (wmin - rss);                        // ref string of the function: rw[0 .. Lr)
        const uint8_t* add = b.read_seq + R.addo[e];
        const int Lh = Lr - nrem + nadd, nmin = Lr < Lh ? Lr : Lh;
        auto hapAt = [&](int i) -> int { return i <= cut ? rw[i] : (i < cut + 1 + nadd ? add[i - cut - 1] : rw[i - nadd + nrem]); };
        // rightmost placement: first mismatch from the left (hap == ref up to cut)
        int fwd = nmin;
        for (int i0 = cut + 1; i0 < nmin; i0 += 64) {
            const int i = i0 + lane;
            const bool mm = i < nmin && hapAt(i) != rw[i];
            const unsigned long long m = __ballot(mm);
            if (m) { fwd = i0 + __builtin_ctzll(m); break; }
        }
        const int maxPos = wmin + fwd + nrem;
        // leftmost placement: first mismatch from the right (the tails behind the indel are the same bytes)
        int back = -1;
        for (int k0 = Lr - (cut + nrem + 1); k0 < nmin; k0 += 64) {
            const int k = k0 + lane;
            const bool mm = k < nmin && hapAt(Lh - 1 - k) != rw[Lr - 1 - k];
            const unsigned long long m = __ballot(mm);
            if (m) { back = k0 + __builtin_ctzll(m); break; }
        }
        if (back < 0) continue;                                        // no mismatch at all: the variant stays as it is
        const int first = Lr - back - nrem, newPos = wmin + first - 1;
        if (first < 0) { if (lane == 0) atomicOr(&R.status, 1); continue; }   // "Error in variant conversion to standard format"
        int off = 0;
        if (nadd) {
            if (lane == 0) off = atomicAdd(&R.addedUsed, nadd);
            off = __shfl(off, 0, 64);
            if (off + nadd > b.cap_added) { if (lane == 0) atomicOr(&R.status, 5); continue; }
            for (int i = lane; i < nadd; i += 64) blob[off + i] = (uint8_t)hapAt(first + i);
        }
        if (lane == 0) {
            R.pos[e] = newPos; R.bmin[e] = newPos; R.bmax[e] = maxPos;
            if (nrem) R.remc[e] = wmin + first;
            if (nadd) R.addo[e] = -(off + 1);
            R.key[e] = (unsigned long long)(unsigned)newPos << 32 | (R.key[e] & 0xFFFFFFFFull);
        }
    }
    __threadfence_block();
    __syncthreads();
    SB_MARK();
    // ---- second sort (stable on the first)
    for (int i = tid; i < n; i += SB_THREADS) {
        const unsigned long long k = R.key[i]; const int id = R.id[i];
        int r = 0;
        for (int j = 0; j < n; ++j) { const unsigned long long kj = R.key[j]; r += (kj < k) || (kj == k && R.id[j] < id); }
        R.perm[r] = (unsigned short)i;
    }
    __syncthreads();
    SB_MARK();
    // ---- filterVariants: runs of equal variants (each compared with the run's first: equality is transitive) are merged into the first
    for (int r = tid; r < n; r += SB_THREADS) R.head[r] = r == 0 || !sb_same(R, R.perm[r], R.perm[r - 1], b.read_seq, blob);
    __syncthreads();
    for (int r = tid; r < n; r += SB_THREADS) {
        R.keep[r] = 0;
        if (!R.head[r]) continue;
        const int e = R.perm[r];
        int supp = R.supp[e], mn = R.bmin[e], mx = R.bmax[e], q = r + 1;
        for (; q < n && !R.head[q]; ++q) { const int f = R.perm[q]; supp += R.supp[f]; mn = min(mn, R.bmin[f]); mx = max(mx, R.bmax[f]); }
        R.supp[e] = supp; R.bmin[e] = mn; R.bmax[e] = mx;              // (Variant.addVariant, variant.pyx:261-268)
        const int size = max(R.nadd[e], R.nrem[e]);
        // every candidate here comes from the reads alone (varSource == PLATYPUS_VAR); minSupport == options.minReads (variantcaller.pyx:526)
        if (q == n) R.keep[r] = supp >= o.minReads;                     // the last run: no size test (variantFilter.pyx:160-169)
        else R.keep[r] = supp >= o.minReads && size <= o.maxSize;
    }
    __syncthreads();
    // ---- orders that depend on how a Python-2 dictionary iterates: not decided here
    //  (a) three or more candidates with one key among which one variant occurs twice next to a different one
    for (int r = tid; r < n; r += SB_THREADS) {
        const unsigned long long k = R.key[R.perm[r]];
        if (r > 0 && R.key[R.perm[r - 1]] == k) continue;
        int m = 1;
        while (r + m < n && R.key[R.perm[r + m]] == k) ++m;
        if (m < 3) continue;
        bool twice = false, other = false;
        if (m > 48) twice = other = true;
        for (int x = 0; x < m && m <= 48; ++x)
            for (int y = x + 1; y < m; ++y) { if (sb_same(R, R.perm[r + x], R.perm[r + y], b.read_seq, blob)) twice = true; else other = true; }
        if (twice && other && pass == 0) atomicOr(&R.status, 2);
    }
    // ---- the survivors, in order
    {
        const int per = (n + SB_THREADS - 1) / SB_THREADS, r0 = tid * per, r1 = min(n, r0 + per);
        int cnt = 0;
        for (int r = r0; r < r1; ++r) cnt += R.keep[r];
        int at = sb_block_scan(cnt, R.wsum, &R.tot);
        for (int r = r0; r < r1; ++r) if (R.keep[r]) R.list[at++] = R.perm[r];
    }
    __syncthreads();
    nk = R.tot;
    //  (b) two survivors that compare equal
    for (int k = tid + 1; k < nk; k += SB_THREADS) if (pass == 0 && R.key[R.list[k]] == R.key[R.list[k - 1]]) atomicOr(&R.status, 2);
    if (nk > b.cap_vars) atomicOr(&R.status, 5);                      // (bit 2: a capacity of the caller, not an exception)
    __syncthreads();
    const int st0 = R.status;
    __syncthreads();
    if (st0 & 1) { if (tid == 0) { hdr[0] = PLAT_SB_HOST; hdr[1] = hdr[2] = 0; hdr[5] = (st0 & 4) ? 6 : 1; } return; }
    if (!(st0 & 2)) break;
    // ---- the order depends on the dictionaries: replay them (variantcaller.pyx:456-470: the sample's variantHeap walked with iteritems(),
    // what passes the support filter put into the all-samples generator's variantHeap, its values() sorted)
    {
        const int32_t* tab = mtab + (size_t)g * 2 * 8192;
#ifdef PLAT_SB_TIMING
        const long long tq0 = wall_clock64();
#endif
        if (tid == 0) { R.dN = 0; R.dN2 = 0; }
        __syncthreads();
        for (int sl = tid; sl < 8192; sl += SB_THREADS) {                // every distinct record of the scan (the merge kernel's table)
            const int id = tab[sl] - 1;
            if (id >= 0) { const int k = atomicAdd(&R.dN, 1); if (k < SB_DICT_CAP) R.dHash[k] = id; }
        }
        __syncthreads();
        const int n1 = R.dN;
        if (n1 > SB_DICT_CAP || !b.cand_rec || !b.region_name_hash) { if (tid == 0) { hdr[0] = PLAT_SB_HOST; hdr[1] = hdr[2] = 0; hdr[5] = 2; } return; }
        // ascending first-record id = the order they entered the dictionary.  The ids of a scan spread evenly over its reads' range: 1024
        // buckets of equal width hold two or three each -- count, scan, scatter, then rank inside the bucket
        {
            int* bcnt = (int*)R.key;                                     // 1024 counters, 1024 starts (the pass-0 arrays are dead)
            int* bstart = bcnt + 1024;
            int lo = 0x7FFFFFFF, hi = 0;
            for (int i = tid; i < n1; i += SB_THREADS) { lo = min(lo, R.dHash[i]); hi = max(hi, R.dHash[i]); }
            for (int d = 32; d; d >>= 1) { lo = min(lo, __shfl_xor(lo, d, 64)); hi = max(hi, __shfl_xor(hi, d, 64)); }
            if (lane == 0) { R.wsum[wv] = lo; }
            bcnt[tid] = 0;
            __syncthreads();
            lo = R.wsum[0];
            for (int k = 1; k < SB_THREADS / 64; ++k) lo = min(lo, R.wsum[k]);
            __syncthreads();
            if (lane == 0) R.wsum[wv] = hi;
            __syncthreads();
            hi = R.wsum[0];
            for (int k = 1; k < SB_THREADS / 64; ++k) hi = max(hi, R.wsum[k]);
            __syncthreads();
            int sh = 0;
            while (((hi - lo) >> sh) >= 1024) ++sh;
            for (int i = tid; i < n1; i += SB_THREADS) atomicAdd(&bcnt[(R.dHash[i] - lo) >> sh], 1);
            __syncthreads();
            const int mine = bcnt[tid];
            const int ex = sb_block_scan(mine, R.wsum, &R.tot);
            bstart[tid] = ex;
            bcnt[tid] = 0;
            __syncthreads();
            for (int i = tid; i < n1; i += SB_THREADS) {                 // scatter into the bucket's stretch of dId (any order inside it)
                const int id = R.dHash[i], bk = (id - lo) >> sh;
                R.dId[bstart[bk] + atomicAdd(&bcnt[bk], 1)] = id;
            }
            __syncthreads();
            for (int i = tid; i < n1; i += SB_THREADS) {                 // rank inside the bucket -> final place (through dHash: dId is being read)
                const int id = R.dId[i], bk = (id - lo) >> sh, b0 = bstart[bk], b1 = b0 + bcnt[bk];
                int r = 0;
                for (int j = b0; j < b1; ++j) r += R.dId[j] < id;
                R.dHash[b0 + r] = id;
            }
            __syncthreads();
            for (int i = tid; i < n1; i += SB_THREADS) R.dId[i] = R.dHash[i];
        }
        __syncthreads();
#ifdef PLAT_SB_TIMING
        const long long tq1 = wall_clock64();
#endif
        const unsigned long long nameHash = (unsigned long long)b.region_name_hash[g];
        for (int i = tid; i < n1; i += SB_THREADS) {
            const int32_t* me = b.cand_rec + 5ll * R.dId[i];
            R.dHash[i] = sb_py2_variant_hash(nameHash, me[0] < 0 ? 0 : me[0], b.ref_seq + me[3], me[1], b.read_seq + me[4], me[2]);
            R.dCand[i] = 0;
        }
        __syncthreads();
        for (int c = tid; c < n; c += SB_THREADS) {                      // which of them are candidates (passed the support filter)
            const int id = b.cand[8ll * ((long long)g * b.cap_per_scan + c)];
            int lo = 0, hi = n1;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (R.dId[mid] < id) lo = mid + 1; else hi = mid; }
            if (lo < n1 && R.dId[lo] == id) R.dCand[lo] = (unsigned short)(c + 1);
        }
        __syncthreads();
        // (scratch of the replay: the candidate arrays of pass 0 are dead -- pass 1 loads them again)
#ifdef PLAT_SB_TIMING
        const long long tq2 = wall_clock64();
#endif
        unsigned* owner = (unsigned*)R.key;                               // 8192 words = key, id, pos, nrem, nadd, supp, remc
        unsigned char* step = (unsigned char*)R.addo;                    // one byte per key (addo, bmin: 8 KB)
        sb_py2_dict_order(n1, R.dHash, owner, step, R.dOrd, R.dScr, R.wsum, &R.tot, &R.dN2);
#ifdef PLAT_SB_TIMING
        const long long tq3 = wall_clock64();
#endif
        // the sample's dictionary walked in its order: the keys that pass enter the second dictionary in that order
        {
            const int per = (n1 + SB_THREADS - 1) / SB_THREADS, r0 = tid * per, r1 = min(n1, r0 + per);
            int cnt = 0;
            for (int r = r0; r < r1; ++r) cnt += R.dCand[R.dOrd[r]] != 0;
            int at = sb_block_scan(cnt, R.wsum, &R.tot);
            for (int r = r0; r < r1; ++r) if (R.dCand[R.dOrd[r]] && at < SB_CAP) { R.dP2[at] = R.dOrd[r]; R.dId[at] = R.dHash[R.dOrd[r]]; ++at; }
        }
        __syncthreads();
        const int m2 = min(R.tot, SB_CAP);
        sb_py2_dict_order(m2, R.dId, owner, step, R.dO2, R.dS2, R.wsum, &R.tot, &R.dN2);
        // rank of every candidate in the second dictionary's order: candidate -> rank in dOrd (free now)
        for (int r = tid; r < m2; r += SB_THREADS) R.dOrd[R.dCand[R.dP2[R.dO2[r]]] - 1] = (unsigned short)r;
        __syncthreads();
#ifdef PLAT_SB_TIMING
        if (tid == 0) printf("[sb replay] region %d: %d distinct, %d candidates; gather+sort %lld us, hash+mark %lld us, dict1 %lld us, dict2+ranks %lld us\n", g, n1, m2,
                             (tq1 - tq0) / 100, (tq2 - tq1) / 100, (tq3 - tq2) / 100, (wall_clock64() - tq3) / 100);
#endif
        if (tid == 0) { R.dN2 = m2; R.status = 0; R.nIndel = 0; R.addedUsed = 0; }
        __syncthreads();
        if (R.dN2 != n) { if (tid == 0) { hdr[0] = PLAT_SB_HOST; hdr[1] = hdr[2] = 0; hdr[5] = 3; } return; }   // (every candidate is a distinct record: cannot happen)
    }
    }   // pass
    SB_MARK();
    if (tid == 0) hdr[5] = 0;
    // ---- the region's variants out (+ the added bases of those that still live in the read table)
    for (int k = tid; k < nk; k += SB_THREADS) {
        const int e = R.list[k];
        const long long v = (long long)g * b.cap_vars + k;
        out.var_pos[v] = R.pos[e]; out.var_nrem[v] = R.nrem[e]; out.var_nadd[v] = R.nadd[e]; out.var_support[v] = R.supp[e];
        out.var_bam_min[v] = R.bmin[e]; out.var_bam_max[v] = R.bmax[e]; out.var_rem_pos[v] = R.remc[e];
        int ao = 0;
        if (R.nadd[e]) {
            if (R.addo[e] < 0) ao = -(R.addo[e] + 1);
            else {
                ao = atomicAdd(&R.addedUsed, R.nadd[e]);
                if (ao + R.nadd[e] > b.cap_added) { atomicOr(&R.status, 5); ao = 0; }
                else for (int i = 0; i < R.nadd[e]; ++i) blob[ao + i] = b.read_seq[R.addo[e] + i];
            }
        }
        out.var_add_off[v] = ao;
    }
    __syncthreads();
    SB_MARK();
    // ---- windows: WindowGenerator.getBunchesOfVariants over the variants inside [start, end), one lane (a sequential rule)
    if (tid == 0) {
        const int start = b.region_start[g], end = b.region_end[g], maxContigPos = contigLen - 1;
        const int maxSpan = o.largeWindows == 1 ? o.maxSize : rlen;
        int nw = 0, st = R.status;
        bool haveBunch = false;
        int bMin = 0, bMax = 0, bCount = 0, bFirst = 0;
        auto emit = [&]() {
            const int ws = max(bMin - o.minVarDist, start), we = min(bMax + o.minVarDist, maxContigPos);
            if (we - ws > o.maxSize) return;                           // variantcaller.pyx:566-568
            if (nw >= b.cap_windows) { st = 5; return; }
            const long long w = (long long)g * b.cap_windows + nw;
            out.win_start[w] = ws; out.win_end[w] = we; out.win_var_first[w] = bFirst; out.win_var_n[w] = bCount;
            ++nw;
        };
        int k = 0;
        while (k < nk && R.pos[R.list[k]] < start) ++k;
        while (k < nk) {
            const int p = R.pos[R.list[k]];
            if (p >= end) break;
            int gMin = p, gMax = p, gCount = 0;                        // the variants at one position (getVariantsByPos)
            const int gFirst = k;
            for (; k < nk && R.pos[R.list[k]] == p; ++k) { const int e = R.list[k]; gMax = max(gMax, max(p, p + R.nrem[e] - 1)); ++gCount; }
            if (!haveBunch) { haveBunch = true; bMin = gMin; bMax = gMax; bCount = gCount; bFirst = gFirst; continue; }
            const int gap = gMin - bMax;
            bool merge;
            if (bMax >= gMin) merge = true;                            // overlapping variants always share a window
            else if (!o.mergeClusteredVariants || gap >= o.maxVarDist) merge = false;
            else if (gMax - bMin > maxSpan) merge = false;
            else if (bCount + gCount <= o.maxVariants) merge = true;
            else merge = gap < o.minVarDist;
            if (merge) { bMin = min(bMin, gMin); bMax = max(bMax, gMax); bCount += gCount; }
            else { emit(); bMin = gMin; bMax = gMax; bCount = gCount; bFirst = gFirst; }
        }
        if (haveBunch) emit();
        if (st) { hdr[0] = PLAT_SB_HOST; hdr[1] = hdr[2] = 0; hdr[5] = (st & 4) ? 6 : 4; }
        else { hdr[0] = 0; hdr[1] = nk; hdr[2] = nw; hdr[4] = R.addedUsed; hdr[5] = 0; hdr[6] = R.dN2 > 0; hdr[7] = 0; }
#ifdef PLAT_SB_TIMING
        SB_MARK();
        if (g < 2) { printf("[sb variants] region %d: %d candidates, %d variants, %d windows, replay %d; us between marks:", g, n, nk, nw, R.dN2 > 0);
                     for (int i = 1; i < tmn_; ++i) printf(" %lld", (tm_[i] - tm_[i - 1]) / 100); printf("\n"); }
#endif
    }
#undef SB_MARK
}

// ---- a haplotype of a window as segments of the reference and added bases (getMutatedSequence, chaplotype.pyx:397-449) -------------
// Walks the variants of `mask` (bit i = variant first + i of the region's list); calls seg(kind, source, length): kind 0 = contig bases
// [source, source + length), kind 1 = added bases of variant `source`.  Returns the length, or -1 where the reference raises
// ("Cannot have beginPos > endPos in getSequence").
struct SbWin { int hapStart, hapEnd, endBuf, contigLen; };
template <class F>
__device__ __forceinline__ int sb_walk_hap(const SbWin& w, unsigned mask, int nv, const int32_t* vpos, const int32_t* vnrem, const int32_t* vnadd, F&& seg)
{
    int L = 0;
    bool bad = false;
    auto refSeg = [&](int bpos, int epos) {                           // FastaFile.getSequence(bpos, epos), fastafile.pyx:173-207
        if (bpos < 0) bpos = 0;
        if (epos > w.contigLen - 1) epos = w.contigLen - 1;
        if (epos < bpos) { bad = true; return; }
        if (epos > bpos) { seg(0, bpos, epos - bpos); L += epos - bpos; }
    };
    refSeg(w.hapStart - w.endBuf, w.hapStart);
    int cur = w.hapStart;
    bool firstSeen = false;
    for (int i = 0; i < nv; ++i) {
        if (!(mask >> i & 1u)) continue;
        const int p = vpos[i], nrem = vnrem[i], nadd = vnadd[i];
        if (!firstSeen) { firstSeen = true; if (p != cur) { refSeg(cur, p); cur = p; } }
        if (p > cur) { refSeg(cur, p); cur = p; }
        if (nadd == nrem) { seg(1, i, nadd); L += nadd; cur += nrem; }
        else {
            if (nadd == 0 || nrem == 0) { if (p == cur) { if (p >= 0 && p < w.contigLen) seg(0, p, 1); else seg(2, '-', 1); L += 1; cur += 1; } }
            cur += nrem;
            if (nadd) { seg(1, i, nadd); L += nadd; }
        }
    }
    if (cur < w.hapEnd) refSeg(cur, w.hapEnd);
    refSeg(w.hapEnd, w.hapEnd + w.endBuf);
    return bad ? -1 : L;
}

// isHaplotypeValid (platypusutils.pyx:735-802): consecutive variants must not overlap; a SNP / MNP may end where an indel starts
__device__ __forceinline__ bool sb_valid(unsigned mask, int nv, const int32_t* vpos, const int32_t* vnrem, const int32_t* vnadd) {
    int last = -1;
    for (int i = 0; i < nv; ++i) {
        if (!(mask >> i & 1u)) continue;
        if (last >= 0) {
            const int amax = max(vpos[last], vpos[last] + vnrem[last] - 1), bmin = vpos[i];
            if (amax > bmin) return false;
            if (amax == bmin && !(vnadd[last] == vnrem[last] && vnadd[i] != vnrem[i])) return false;
        }
        last = i;
    }
    return true;
}
// the k-th subset of nv variants in itertools.combinations order over sizes 1, 2, ... (k from 0); 0 when there is none
__device__ __forceinline__ unsigned sb_next_comb(unsigned prev, int nv) {
    // combinations of one size ascend lexicographically by index tuple = descend by the bit-reversed mask
    auto rev = [nv](unsigned m) { unsigned r = 0; for (int i = 0; i < nv; ++i) if (m >> i & 1u) r |= 1u << (nv - 1 - i); return r; };
    const int size = prev ? __popc(prev) : 0;
    if (prev) {
        for (unsigned r = rev(prev); r-- > 0;) if (__popc(r) == size) return rev(r);
    }
    if (size + 1 > nv) return 0;
    unsigned r = ((1u << (size + 1)) - 1u) << (nv - size - 1);        // the first of the next size: indices 0 .. size
    return rev(r);
}

// ReadArray.setWindowPointers (cwindow.pyx:176-234): first read that may overlap [start, end) / first read starting at or behind end
__device__ __forceinline__ int sb_lower_bound(const int32_t* a, int n, long long key) {
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if ((long long)a[mid] < key) lo = mid + 1; else hi = mid; }
    return lo;
}

// scratch of the stage (ints): per window slot t eight ints {flags, haplotypes, reads, haplotype bytes, read bytes, longest haplotype, -, -} at
// 8 t; eight int64 at 8 S + 16 t: the exclusive prefix INSIDE its region of {windows, haplotypes, reads, pairs, haplotype bytes, read bytes,
// genotype likelihoods}; per region g 24 int64 at 24 S + 48 g: [0..6] the region's totals of the same, [7..9] maxima (haplotype length, reads,
// haplotypes of a window), [10..16] the region's base in the batch (k_sb_scan).  S = n_regions x cap_windows.
__device__ __forceinline__ int32_t* sb_slot(const plat_stage_b_out& out, long long t) { return out.scratch + 8 * t; }
__device__ __forceinline__ long long* sb_prefix(const plat_stage_b_in& b, const plat_stage_b_out& out, long long t) {
    return (long long*)(out.scratch + 8ll * b.n_regions * b.cap_windows) + 8 * t;
}
__device__ __forceinline__ long long* sb_region(const plat_stage_b_in& b, const plat_stage_b_out& out, int g) {
    return (long long*)(out.scratch + 24ll * b.n_regions * b.cap_windows) + 24 * g;
}
// ... and per window slot 32 words behind the regions' block: the masks of its haplotypes in their final (sorted) order
__device__ __forceinline__ uint32_t* sb_masks(const plat_stage_b_in& b, const plat_stage_b_out& out, long long t) {
    return (uint32_t*)(out.scratch + 24ll * b.n_regions * b.cap_windows + 48ll * b.n_regions) + 32 * t;
}

// one window: window pointers, the decision, the valid combinations counted
static __device__ void sb_one_window(const plat_stage_b_in& b, const plat_stage_b_options& o, const plat_stage_b_out& out, int g, int k, int32_t* sc)
{
    const long long w = (long long)g * b.cap_windows + k;
    const int ws = out.win_start[w], we = out.win_end[w], nv = out.win_var_n[w];
    const int contigLen = b.contig_len[g], rlen = b.region_rlen[g];
    int flags = 0, ptr[6] = {0, 0, 0, 0, 0, 0};
    // window pointers of the three read arrays
    for (int a = 0; a < 3; ++a) {
        const int N = b.tab_n[3 * g + a], base = b.tab_begin[3 * g + a], longest = b.tab_longest[3 * g + a];
        int s = 0, e = 0;
        if (N > 0) {
            const long long keyS = (long long)ws - longest > 1 ? (long long)ws - longest : 1;
            if (a < 2) {
                s = sb_lower_bound(b.read_pos + base, N, keyS);
                e = sb_lower_bound(b.read_pos + base, N, we);
                while (s < N && b.read_end[base + s] <= ws) ++s;
            } else {                                                  // setWindowPointersBasedOnMatePos (:236-264)
                const int32_t* mp = b.broken_mate_pos + (base - b.broken_base);
                s = sb_lower_bound(mp, N, keyS);
                e = sb_lower_bound(mp, N, we);
            }
            if (s > e) flags = PLAT_SBW_HOST;                          // "Read start pointer > read end pointer": the reference raises
            if (e > N) e = N;
        }
        ptr[2 * a] = s; ptr[2 * a + 1] = e;
    }
    for (int q = 0; q < 6; ++q) out.win_ptrs[6 * w + q] = ptr[q];
    const int nGood = ptr[1] - ptr[0];
    SbWin W;
    W.hapStart = ws > 0 ? ws : 0; W.hapEnd = we < contigLen - 1 ? we : contigLen - 1; W.endBuf = 2 * rlen < 500 ? 2 * rlen : 500; W.contigLen = contigLen;
    // the reference window handed over must hold every byte a haplotype can take
    const int rss = b.ref_seq_start[g], refLen = (int)(b.ref_off[g + 1] - b.ref_off[g]);
    const int lo = W.hapStart - W.endBuf > 0 ? W.hapStart - W.endBuf : 0, hi = min(W.hapEnd + W.endBuf, contigLen - 1);
    if (lo < rss || hi > rss + refLen) flags = PLAT_SBW_HOST;
    for (int i = 0; i < nv; ++i) {                                     // (an indel's anchor base is read at its position: getCharacter reaches the contig's last base, getSequence never does)
        const int p = out.var_pos[(long long)g * b.cap_vars + out.win_var_first[w] + i];
        if (p < rss || p >= rss + refLen) flags = PLAT_SBW_HOST;
    }
    int nHaps = 0, maxLen = 0;
    long long hapBytes = 0;
    if (!flags) {
        // (the fields of the window's first variants in registers: the walks below read them dozens of times)
        int32_t vpos[SB_MAXCOMB], vnrem[SB_MAXCOMB], vnadd[SB_MAXCOMB];
        {
            const long long v0 = (long long)g * b.cap_vars + out.win_var_first[w];
#pragma unroll
            for (int i = 0; i < SB_MAXCOMB; ++i) {
                const bool in = i < nv;
                vpos[i] = in ? out.var_pos[v0 + i] : 0; vnrem[i] = in ? out.var_nrem[v0 + i] : 0; vnadd[i] = in ? out.var_nadd[v0 + i] : 0;
            }
        }
        auto none = [](int, int, int) {};
        const int refL = sb_walk_hap(W, 0u, 0, vpos, vnrem, vnadd, none);
        if (refL < 0 || refL > 16384) flags = PLAT_SBW_HOST;          // (raises there: logged, the window is skipped -- the caller reproduces it)
        else if (nGood == 0 || (double)nGood > o.maxReads) flags = PLAT_SBW_SKIP;
        else if (nv > o.maxVariants) flags = o.skipDifficultWindows ? PLAT_SBW_SKIP : PLAT_SBW_HOST;   // filterVariantsByCoverage: the caller's
        else {
            const double lg = log2((double)(o.maxHaplotypes - 1));
            if (!((double)nv <= lg || (o.filterVarsByCoverage && (double)o.maxVariants <= lg)) || nv > SB_MAXCOMB) flags = PLAT_SBW_HOST;   // the greedy filter
        }
        if (!flags) {
            nHaps = 1; hapBytes = refL; maxLen = refL;
            for (unsigned m = sb_next_comb(0u, nv); m; m = sb_next_comb(m, nv)) {
                if (!sb_valid(m, nv, vpos, vnrem, vnadd)) continue;
                const int L = sb_walk_hap(W, m, nv, vpos, vnrem, vnadd, none);
                if (L < 0 || L > 16384) { flags = PLAT_SBW_HOST; break; }
                ++nHaps; hapBytes += L; maxLen = max(maxLen, L);
            }
        }
    }
    long long readBytes = 0;
    int nReads = 0;
    if (!flags) {
        for (int a = 0; a < 3; ++a) {
            const int base = b.tab_begin[3 * g + a];
            nReads += ptr[2 * a + 1] - ptr[2 * a];
            readBytes += b.read_off[base + ptr[2 * a + 1]] - b.read_off[base + ptr[2 * a]];
        }
        if (readBytes > 0x7FFFFFFF || hapBytes > 0x7FFFFFFF) flags = PLAT_SBW_HOST;
    }
    out.win_flags[w] = flags; out.win_n_haps[w] = flags ? 0 : nHaps; out.win_batch[w] = -1;
    sc[0] = flags; sc[1] = nHaps; sc[2] = nReads; sc[3] = (int)hapBytes; sc[4] = (int)readBytes; sc[5] = maxLen; sc[6] = 0; sc[7] = 0;
}


// exclusive scan of a 64-bit value over the 256 threads of a workgroup, total in *tot
__device__ __forceinline__ long long sb_block_scan64(long long v, long long* wsum, long long* tot) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    long long x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const long long y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
    if (lane == 63) wsum[w] = x;
    __syncthreads();
    long long base = 0;
    for (int k = 0; k < w; ++k) base += wsum[k];
    if (threadIdx.x == blockDim.x - 1) *tot = base + x;
    __syncthreads();
    return base + x - v;
}

// one workgroup per region: its windows, one per thread
__global__ void __launch_bounds__(256)
k_sb_windows(SbIn in, plat_stage_b_out out)
{
    const plat_stage_b_in& b = in.b;
    const int g = blockIdx.x, tid = threadIdx.x;
    const int32_t* hdr = out.hdr + 8 * g;
    const int nW = hdr[0] != 0 ? 0 : hdr[2];
    for (int k = tid; k < nW; k += 256) sb_one_window(b, in.o, out, g, k, sb_slot(out, (long long)g * b.cap_windows + k));
}

// one workgroup per region, behind k_sb_haps_rank (which knows how many haplotypes a window keeps): prefix sums of the region's windows
__global__ void __launch_bounds__(256)
k_sb_prefix(SbIn in, plat_stage_b_out out)
{
    const plat_stage_b_in& b = in.b;
    __shared__ long long wsum[4], tot, carry[7];
    __shared__ int mx[3];
    const int g = blockIdx.x, tid = threadIdx.x;
    const int32_t* hdr = out.hdr + 8 * g;
    const int nW = hdr[0] != 0 ? 0 : hdr[2];
    if (tid < 7) carry[tid] = 0;
    if (tid < 3) mx[tid] = 0;
    __syncthreads();
    for (int k0 = 0; k0 < nW; k0 += 256) {
        const int k = k0 + tid;
        const long long t = (long long)g * b.cap_windows + k;
        long long v[7] = {0, 0, 0, 0, 0, 0, 0};
        if (k < nW) {
            const int32_t* sc = sb_slot(out, t);
            if (sc[0] == 0) {
                v[0] = 1; v[1] = sc[1]; v[2] = sc[2]; v[3] = (long long)sc[1] * sc[2]; v[4] = sc[3]; v[5] = sc[4]; v[6] = (long long)sc[1] * (sc[1] + 1) / 2;
                atomicMax(&mx[0], sc[5]); atomicMax(&mx[1], sc[2]); atomicMax(&mx[2], sc[1]);
            }
        }
        long long* pf = k < nW ? sb_prefix(b, out, t) : nullptr;
        for (int q = 0; q < 7; ++q) {
            const long long ex = sb_block_scan64(v[q], wsum, &tot);
            if (pf) pf[q] = carry[q] + ex;
            __syncthreads();
            if (tid == 0) carry[q] += tot;
            __syncthreads();
        }
    }
    if (tid == 0) {
        long long* rg = sb_region(b, out, g);
        for (int q = 0; q < 7; ++q) rg[q] = carry[q];
        rg[7] = mx[0]; rg[8] = mx[1]; rg[9] = mx[2];
    }
}

// ---- where every region's windows go in the batch: exclusive scan over the regions, one wave -------------------------------------------
__global__ void __launch_bounds__(64)
k_sb_scan(SbIn in, plat_stage_b_out out)
{
    const plat_stage_b_in& b = in.b;
    const int lane = threadIdx.x;
    long long carry[7] = {0, 0, 0, 0, 0, 0, 0};
    int m0 = 0, m1 = 0, m2 = 0;
    for (int g0 = 0; g0 < b.n_regions; g0 += 64) {
        const int g = g0 + lane;
        long long* rg = g < b.n_regions ? sb_region(b, out, g) : nullptr;
        for (int q = 0; q < 7; ++q) {
            const long long v = rg ? rg[q] : 0;
            long long x = v;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const long long y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
            if (rg) rg[10 + q] = carry[q] + x - v;
            carry[q] += __shfl(x, 63, 64);
        }
        if (rg) { m0 = max(m0, (int)rg[7]); m1 = max(m1, (int)rg[8]); m2 = max(m2, (int)rg[9]); }
    }
#pragma unroll
    for (int d = 32; d; d >>= 1) { m0 = max(m0, __shfl_xor(m0, d, 64)); m1 = max(m1, __shfl_xor(m1, d, 64)); m2 = max(m2, __shfl_xor(m2, d, 64)); }
    if (lane == 0) {
        const long long* run = carry;
        const bool over = run[0] > b.cap_batch_windows || run[1] > b.cap_batch_haps || run[2] > b.cap_batch_reads || run[4] > b.cap_hap_bytes;
        out.totals[0] = run[0]; out.totals[1] = run[1]; out.totals[2] = run[2]; out.totals[3] = run[3]; out.totals[4] = run[6];
        out.totals[5] = run[4]; out.totals[6] = run[5]; out.totals[7] = m0; out.totals[8] = m1; out.totals[9] = m2; out.totals[10] = over ? 1 : 0;
        for (int q = 11; q < 16; ++q) out.totals[q] = 0;
        if (!over) {                                                   // the arrays' closing entries
            const int nw = (int)run[0];
            out.b_hap_begin[nw] = (int)run[1]; out.b_read_begin[nw] = (int)run[2]; out.b_pair_off[nw] = run[3]; out.b_seg_begin[nw] = (int)run[2];
            out.b_gl_off[nw] = run[6];
            out.b_hap_off[run[1]] = run[4]; out.b_read_off[run[2]] = run[5];
        }
    }
}

// ---- haplotypes of a window: sorted, equal ones merged (k_sb_haps_rank), then their bytes (k_sb_haps_write) ----------------------------
// One workgroup (four waves) per window, haplotype h handled by wave h % 4.  Every haplotype of a window is the reference up to the
// window's first variant: the SB_STAGE bytes behind that point are staged in LDS and sorted(haplotypes) is decided there (two strings that
// agree on all of them and go on are left to the caller: a tandem repeat longer than the stage).  mergeHaplotypes (variantcaller.pyx:
// 325-383): of haplotypes with one sequence the one with the largest product of its variants' priors stays (the first of them on a tie)
// -- decided here when all their variants are SNPs / multi-nucleotide variants (variant.pyx:219-259: 1e-3 / 3; 5e-5 x 0.1^(nDiffs-1) x 0.9
// with the host's pow), left to the caller when an indel is among them (its prior needs the repeat annotation).
constexpr int SB_STAGE = 384;
__global__ void __launch_bounds__(256)
k_sb_haps_rank(SbIn in, plat_stage_b_out out)
{
    const plat_stage_b_in& b = in.b;
    __shared__ unsigned s_mask[32];
    __shared__ int s_len[32], s_rank[32], s_dup, s_eq[32], s_drop[32];
    __shared__ double s_prior[32];
    __shared__ int s_seg[4][24][3];                                    // segments of the haplotype a wave is staging: kind, source, length
    __shared__ int32_t s_vpos[8], s_vnrem[8], s_vnadd[8], s_vadd[8], s_vrem[8];
    __shared__ __align__(16) uint8_t s_stage[32][SB_STAGE];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, tid = threadIdx.x;
    const int g = blockIdx.y;
    const int nWin = out.hdr[8 * g] != 0 ? 0 : out.hdr[8 * g + 2];
    const uint8_t* added = out.added + (long long)g * b.cap_added;
    const uint8_t* ref = b.ref_seq + b.ref_off[g];
    const int contigLen = b.contig_len[g], rlen = b.region_rlen[g], rss = b.ref_seq_start[g];
    for (int kw = blockIdx.x; kw < nWin; kw += gridDim.x) {
        __syncthreads();
        const long long t = (long long)g * b.cap_windows + kw;
        int32_t* sc = sb_slot(out, t);
        if (sc[0] != 0) continue;
        const long long w = t;
        const int nv = out.win_var_n[w], nH = sc[1];
        const int ws = out.win_start[w], we = out.win_end[w];
        const long long v0 = (long long)g * b.cap_vars + out.win_var_first[w];
        if (tid < 8) {                                                 // the window's variants in LDS: the walks read them dozens of times
            const bool in_ = tid < nv;
            s_vpos[tid] = in_ ? out.var_pos[v0 + tid] : 0; s_vnrem[tid] = in_ ? out.var_nrem[v0 + tid] : 0; s_vnadd[tid] = in_ ? out.var_nadd[v0 + tid] : 0;
            s_vadd[tid] = in_ ? out.var_add_off[v0 + tid] : 0; s_vrem[tid] = in_ ? out.var_rem_pos[v0 + tid] : 0;
        }
        if (tid < 32) { s_rank[tid] = 0; s_eq[tid] = -1; s_drop[tid] = 0; }
        __syncthreads();
        const int32_t* vpos = s_vpos;
        const int32_t* vnrem = s_vnrem;
        const int32_t* vnadd = s_vnadd;
        const int32_t* vadd = s_vadd;
        SbWin W;
        W.hapStart = ws > 0 ? ws : 0; W.hapEnd = we < contigLen - 1 ? we : contigLen - 1; W.endBuf = 2 * rlen < 500 ? 2 * rlen : 500; W.contigLen = contigLen;
        // the valid combinations in the reference's order (haplotype 0 = the reference) and their lengths
        if (tid == 0) {
            auto none = [](int, int, int) {};
            int h = 0;
            s_mask[0] = 0u; s_len[0] = sb_walk_hap(W, 0u, 0, vpos, vnrem, vnadd, none); h = 1;
            for (unsigned m = sb_next_comb(0u, nv); m && h < 32; m = sb_next_comb(m, nv)) {
                if (!sb_valid(m, nv, vpos, vnrem, vnadd)) continue;
                s_mask[h] = m; s_len[h] = sb_walk_hap(W, m, nv, vpos, vnrem, vnadd, none); ++h;
            }
            s_dup = 0;
        }
        __syncthreads();
        int common = 0;
        if (nv > 0) { const int lo = W.hapStart - W.endBuf > 0 ? W.hapStart - W.endBuf : 0; common = max(0, min(vpos[0], W.hapEnd) - lo); }
        for (int h = wv; h < nH; h += 4) {                             // the bytes behind the common prefix into the stage
            int nseg = 0;
            if (lane == 0) {
                auto put = [&](int kind, int src, int len) { if (nseg < 24) { s_seg[wv][nseg][0] = kind; s_seg[wv][nseg][1] = src; s_seg[wv][nseg][2] = len; } ++nseg; };
                sb_walk_hap(W, s_mask[h], nv, vpos, vnrem, vnadd, put);
            }
            nseg = __shfl(nseg, 0, 64);
            __builtin_amdgcn_wave_barrier();
            __threadfence_block();
            const int L = s_len[h], n = min(SB_STAGE, L - common);
            for (int p = lane; p < n; p += 64) {
                int q = common + p, k = 0;
                while (k < nseg - 1 && q >= s_seg[wv][k][2]) { q -= s_seg[wv][k][2]; ++k; }
                const int kind = s_seg[wv][k][0], src = s_seg[wv][k][1];
                s_stage[h][p] = kind == 0 ? ref[src + q - rss] : (kind == 1 ? added[vadd[src] + q] : (uint8_t)src);
            }
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
        // ranks: sorted(haplotypes) compares the byte strings; equal strings keep their order
        int pairNo = 0;
        for (int i = 0; i < nH; ++i)
            for (int j = i + 1; j < nH; ++j, ++pairNo) {
                if ((pairNo & 3) != wv) continue;
                const int Li = s_len[i] - common, Lj = s_len[j] - common, Lm = min(min(Li, Lj), SB_STAGE);
                int cmp = 0;
                for (int p0 = 0; p0 < Lm && cmp == 0; p0 += 64) {
                    const int p = p0 + lane;
                    const int cx = p < Lm ? s_stage[i][p] : 0, cy = p < Lm ? s_stage[j][p] : 0;
                    const unsigned long long mm = __ballot(cx != cy);
                    if (mm) { const int f = __builtin_ctzll(mm); const int ax = __shfl(cx, f, 64), ay = __shfl(cy, f, 64); cmp = ax < ay ? -1 : 1; }
                }
                bool undecided = false;
                if (cmp == 0) {
                    if (Lm == SB_STAGE && (Li > SB_STAGE || Lj > SB_STAGE)) undecided = true;   // they agree on the whole stage and go on
                    else cmp = Li < Lj ? -1 : (Li > Lj ? 1 : 0);
                }
                if (lane == 0) {
                    if (undecided) s_dup = 1;
                    else if (cmp == 0) atomicMax(&s_eq[j], i);          // j has the sequence of an earlier haplotype: mergeHaplotypes
                    atomicAdd(&s_rank[cmp <= 0 ? j : i], 1);
                }
            }
        __syncthreads();
        // mergeHaplotypes over the runs of equal sequences (in sorted order = insertion order inside a run): the largest prior product stays
        if (tid == 0 && !s_dup) {
            bool any = false;
            for (int h = 0; h < nH; ++h) any = any || s_eq[h] >= 0;
            if (any) {
                for (int h = 0; h < nH && !s_dup; ++h) {               // prior product of every haplotype that has a twin
                    bool twin = s_eq[h] >= 0;
                    for (int k = 0; k < nH && !twin; ++k) twin = s_eq[k] == h;
                    if (!twin) continue;
                    double pr = 1.0;
                    for (int i = 0; i < nv; ++i) {
                        if (!(s_mask[h] >> i & 1u)) continue;
                        const int na = vnadd[i], nr = vnrem[i];
                        double p1;
                        if (na == 1 && nr == 1) p1 = 1e-3 / 3;
                        else if (na == nr) {
                            int nd = 0;
                            for (int q = 0; q < na; ++q) nd += added[vadd[i] + q] != ref[s_vrem[i] + q - rss];
                            if (nd < 1 || nd > 16) { s_dup = 1; break; }
                            p1 = 5e-5 * in.pow01[nd - 1] * (1.0 - 0.1);
                        } else { s_dup = 1; break; }                     // an indel: its prior needs the repeat annotation -- the caller's
                        pr *= p1 < 1e-10 ? 1e-10 : p1;
                    }
                    s_prior[h] = pr;
                }
                if (!s_dup) {
                    // runs: haplotypes with one sequence share the smallest index among them (s_eq chains point to earlier twins)
                    for (int h = 0; h < nH; ++h) {
                        int root = h;
                        while (s_eq[root] >= 0) root = s_eq[root];
                        if (root != h) continue;
                        int best = h;                                    // `last` of the reference's loop: replaced only by a LARGER product
                        for (int k = h + 1; k < nH; ++k) {
                            int rk = k;
                            while (s_eq[rk] >= 0) rk = s_eq[rk];
                            if (rk != h) continue;
                            // (sorted order inside a run = index order: equal strings keep their order)
                            if (s_prior[k] > s_prior[best]) { s_drop[best] = 1; best = k; } else s_drop[k] = 1;
                        }
                    }
                }
            }
        }
        __syncthreads();
        if (s_dup) { if (tid == 0) out.win_flags[w] = PLAT_SBW_DUPLICATE; }   // (stays in the batch with every haplotype; the caller prepares it itself)
        // final order: rank among the haplotypes that stay
        if (tid < nH) {
            int r = 0;
            for (int k = 0; k < nH; ++k) if (!s_drop[k] && (s_rank[k] < s_rank[tid] || (s_rank[k] == s_rank[tid] && k < tid))) ++r;
            if (!s_drop[tid]) sb_masks(b, out, t)[r] = s_mask[tid];
        }
        if (tid == 0) {
            int keep = 0, bytes = 0, mxl = 0;
            for (int k = 0; k < nH; ++k) if (!s_drop[k]) { ++keep; bytes += s_len[k]; mxl = max(mxl, s_len[k]); }
            if (keep <= 1) { sc[0] = PLAT_SBW_SKIP; out.win_flags[w] = PLAT_SBW_SKIP; out.win_n_haps[w] = 0; }   // one haplotype left: the loop does not call the window
            else { sc[1] = keep; sc[3] = bytes; sc[5] = mxl; out.win_n_haps[w] = keep; }
        }
    }
}

// the bytes of every haplotype, once, straight to their place: one workgroup per window, wave per haplotype
__global__ void __launch_bounds__(256)
k_sb_haps_write(SbIn in, plat_stage_b_out out)
{
    const plat_stage_b_in& b = in.b;
    __shared__ int s_len[32], s_off[32];
    __shared__ int s_seg[4][24][3];
    __shared__ int32_t s_vpos[8], s_vnrem[8], s_vnadd[8], s_vadd[8];
    if (out.totals[10]) return;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, tid = threadIdx.x;
    const int g = blockIdx.y;
    const int nWin = out.hdr[8 * g] != 0 ? 0 : out.hdr[8 * g + 2];
    const long long* rg = sb_region(b, out, g);
    const uint8_t* added = out.added + (long long)g * b.cap_added;
    const uint8_t* ref = b.ref_seq + b.ref_off[g];
    const int contigLen = b.contig_len[g], rlen = b.region_rlen[g], rss = b.ref_seq_start[g];
    for (int kw = blockIdx.x; kw < nWin; kw += gridDim.x) {
        __syncthreads();
        const long long t = (long long)g * b.cap_windows + kw;
        const int32_t* sc = sb_slot(out, t);
        if (sc[0] != 0) continue;
        const long long* pf = sb_prefix(b, out, t);
        const long long w = t;
        const int bw = (int)(rg[10] + pf[0]), nv = out.win_var_n[w], nH = sc[1], hb = (int)(rg[11] + pf[1]);
        const long long byte0 = rg[14] + pf[4];
        const int ws = out.win_start[w], we = out.win_end[w];
        const uint32_t* masks = sb_masks(b, out, t);
        if (tid == 0) {                                                // the window's entries in the batch arrays
            const int rb = (int)(rg[12] + pf[2]);
            out.win_batch[w] = bw;
            out.b_hap_begin[bw] = hb; out.b_read_begin[bw] = rb; out.b_pair_off[bw] = rg[13] + pf[3]; out.b_seg_begin[bw] = rb; out.b_gl_off[bw] = rg[16] + pf[6];
            out.b_n_good[bw] = out.win_ptrs[6 * w + 1] - out.win_ptrs[6 * w];
            out.b_start[bw] = ws > 0 ? ws : 0; out.b_end[bw] = we < contigLen - 1 ? we : contigLen - 1; out.b_flank[bw] = 2 * rlen < 500 ? 2 * rlen : 500;
        }
        const long long v0 = (long long)g * b.cap_vars + out.win_var_first[w];
        if (tid < 8) {
            const bool in_ = tid < nv;
            s_vpos[tid] = in_ ? out.var_pos[v0 + tid] : 0; s_vnrem[tid] = in_ ? out.var_nrem[v0 + tid] : 0; s_vnadd[tid] = in_ ? out.var_nadd[v0 + tid] : 0;
            s_vadd[tid] = in_ ? out.var_add_off[v0 + tid] : 0;
        }
        __syncthreads();
        SbWin W;
        W.hapStart = ws > 0 ? ws : 0; W.hapEnd = we < contigLen - 1 ? we : contigLen - 1; W.endBuf = 2 * rlen < 500 ? 2 * rlen : 500; W.contigLen = contigLen;
        if (tid < nH) { auto none = [](int, int, int) {}; s_len[tid] = sb_walk_hap(W, masks[tid], nv, s_vpos, s_vnrem, s_vnadd, none); }
        __syncthreads();
        if (tid < nH) {
            int off = 0;
            for (int k = 0; k < tid; ++k) off += s_len[k];
            s_off[tid] = off;
            out.b_hap_off[hb + tid] = byte0 + off;
            out.b_hap_mask[hb + tid] = masks[tid];
        }
        __syncthreads();
        uint8_t* dst = out.b_hap_seq + byte0;
        for (int h = wv; h < nH; h += 4) {
            int nseg = 0;
            if (lane == 0) {
                auto put = [&](int kind, int src, int len) { if (nseg < 24) { s_seg[wv][nseg][0] = kind; s_seg[wv][nseg][1] = src; s_seg[wv][nseg][2] = len; } ++nseg; };
                sb_walk_hap(W, masks[h], nv, s_vpos, s_vnrem, s_vnadd, put);
            }
            nseg = __shfl(nseg, 0, 64);
            __builtin_amdgcn_wave_barrier();
            __threadfence_block();
            auto byteAt = [&](int p) -> uint8_t {
                int q = p, k = 0;
                while (k < nseg - 1 && q >= s_seg[wv][k][2]) { q -= s_seg[wv][k][2]; ++k; }
                const int kind = s_seg[wv][k][0], src = s_seg[wv][k][1];
                return kind == 0 ? ref[src + q - rss] : (kind == 1 ? added[s_vadd[src] + q] : (uint8_t)src);
            };
            const int L = s_len[h], to = s_off[h];
            int p = lane;
            for (; p + 192 < L; p += 256) {                            // (four loads in flight per lane)
                const uint8_t c0 = byteAt(p), c1 = byteAt(p + 64), c2 = byteAt(p + 128), c3 = byteAt(p + 192);
                dst[to + p] = c0; dst[to + p + 64] = c1; dst[to + p + 128] = c2; dst[to + p + 192] = c3;
            }
            for (; p < L; p += 64) dst[to + p] = byteAt(p);
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ---- the reads of every window: indices into the read table, kinds, offsets ---------------------------------------------------------
__global__ void __launch_bounds__(256)
k_sb_reads(SbIn in, plat_stage_b_out out)
{
    const plat_stage_b_in& b = in.b;
    if (out.totals[10]) return;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = blockIdx.y;
    const int nWin = out.hdr[8 * g] != 0 ? 0 : out.hdr[8 * g + 2];
    const long long* rg = sb_region(b, out, g);
    for (int kw = blockIdx.x * 4 + wv; kw < nWin; kw += gridDim.x * 4) {
        const long long t = (long long)g * b.cap_windows + kw;
        const int32_t* sc = sb_slot(out, t);
        if (sc[0] != 0) continue;
        const long long* pf = sb_prefix(b, out, t);
        const long long w = t;
        int at = (int)(rg[12] + pf[2]);
        long long bytes = rg[15] + pf[5];
        for (int a = 0; a < 3; ++a) {
            const int base = b.tab_begin[3 * g + a], s = out.win_ptrs[6 * w + 2 * a], e = out.win_ptrs[6 * w + 2 * a + 1];
            if (e <= s) continue;
            const long long o0 = b.read_off[base + s];
            for (int i = s + lane; i < e; i += 64) {
                out.b_read_src[at + (i - s)] = base + i;
                out.b_read_kind[at + (i - s)] = (uint8_t)a;
                out.b_read_off[at + (i - s)] = bytes + (b.read_off[base + i] - o0);
            }
            bytes += b.read_off[base + e] - o0;
            at += e - s;
        }
    }
}

}  // namespace plat

PLAT_EXPORT int plat_stage_b_batch(plat_ctx* ctx, const plat_stage_b_in* batch, const plat_stage_b_options* options,
                                   const plat_stage_b_out* outp, void* stream)
{
    if (!ctx || !batch || !options || !outp) return PLAT_ERR_INVALID;
    const plat_stage_b_in& b = *batch;
    const plat_stage_b_out& o = *outp;
    if (b.n_regions < 0 || b.cap_per_scan < 1 || b.cap_vars < 1 || b.cap_windows < 1 || b.cap_added < 1 || b.cap_batch_windows < 1 || b.cap_batch_haps < 1 ||
        b.cap_batch_reads < 1 || b.cap_hap_bytes < 1)
        return PLAT_ERR_INVALID;
    if (b.n_regions == 0) return PLAT_OK;
    if (!b.cand || !b.cand_n || !b.ref_seq || !b.ref_off || !b.ref_seq_start || !b.contig_len || !b.region_start || !b.region_end || !b.region_rlen ||
        !b.read_seq || !b.read_off || !b.read_pos || !b.read_end || !b.tab_begin || !b.tab_n || !b.tab_longest || !b.broken_mate_pos)
        return PLAT_ERR_INVALID;
    if (!o.hdr || !o.var_pos || !o.var_nrem || !o.var_nadd || !o.var_support || !o.var_bam_min || !o.var_bam_max || !o.var_rem_pos || !o.var_add_off ||
        !o.added || !o.win_start || !o.win_end || !o.win_var_first || !o.win_var_n || !o.win_flags || !o.win_ptrs || !o.win_n_haps || !o.win_batch ||
        !o.b_hap_begin || !o.b_read_begin || !o.b_start || !o.b_end || !o.b_flank || !o.b_pair_off || !o.b_gl_off || !o.b_seg_begin || !o.b_n_good ||
        !o.b_hap_off || !o.b_hap_mask || !o.b_hap_seq || !o.b_read_off || !o.b_read_src || !o.b_read_kind || !o.totals || !o.scratch)
        return PLAT_ERR_INVALID;
    if (options->maxHaplotypes < 3) return PLAT_ERR_UNSUPPORTED;
    if (!ctx->merge_tab.ptr) return PLAT_ERR_INVALID;                  // (plat_candidates_merge_batch of THIS context comes first: its table of distinct records is read)
    PLAT_HIP(ctx, hipSetDevice(ctx->device));
    plat::SbIn in;
    in.b = b; in.o = *options;
    for (int k = 0; k < 16; ++k) in.pow01[k] = pow(0.1, (double)k);    // (host libm: what Variant.calculatePrior multiplies with)
    hipStream_t st = (hipStream_t)stream;
    if (!ctx->sb_attr_set) {                                            // (the attribute is per device: remembered per context, as k_assemble's launch sets its own)
        PLAT_HIP(ctx, hipFuncSetAttribute((const void*)plat::k_sb_variants, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(plat::SbRegion)));
        ctx->sb_attr_set = true;
    }
    { PLAT_KT_BEGIN(ctx, PLAT_KT_SB_VARIANTS, st); hipLaunchKernelGGL(plat::k_sb_variants, dim3((unsigned)b.n_regions), dim3(plat::SB_THREADS), sizeof(plat::SbRegion), st, in, o,
                       (const int32_t*)ctx->merge_tab.ptr); PLAT_KT_END(ctx, PLAT_KT_SB_VARIANTS, st); }
    { PLAT_KT_BEGIN(ctx, PLAT_KT_SB_WINDOWS, st); hipLaunchKernelGGL(plat::k_sb_windows, dim3((unsigned)b.n_regions), dim3(256), 0, st, in, o); PLAT_KT_END(ctx, PLAT_KT_SB_WINDOWS, st); }
    { PLAT_KT_BEGIN(ctx, PLAT_KT_SB_HAPS_RANK, st); hipLaunchKernelGGL(plat::k_sb_haps_rank, dim3(48, (unsigned)b.n_regions), dim3(256), 0, st, in, o); PLAT_KT_END(ctx, PLAT_KT_SB_HAPS_RANK, st); }
    { PLAT_KT_BEGIN(ctx, PLAT_KT_SB_PREFIX, st); hipLaunchKernelGGL(plat::k_sb_prefix, dim3((unsigned)b.n_regions), dim3(256), 0, st, in, o); PLAT_KT_END(ctx, PLAT_KT_SB_PREFIX, st); }
    { PLAT_KT_BEGIN(ctx, PLAT_KT_SB_SCAN, st); hipLaunchKernelGGL(plat::k_sb_scan, dim3(1), dim3(64), 0, st, in, o); PLAT_KT_END(ctx, PLAT_KT_SB_SCAN, st); }
    { PLAT_KT_BEGIN(ctx, PLAT_KT_SB_HAPS_WRITE, st); hipLaunchKernelGGL(plat::k_sb_haps_write, dim3(48, (unsigned)b.n_regions), dim3(256), 0, st, in, o); PLAT_KT_END(ctx, PLAT_KT_SB_HAPS_WRITE, st); }
    { PLAT_KT_BEGIN(ctx, PLAT_KT_SB_READS, st); hipLaunchKernelGGL(plat::k_sb_reads, dim3(12, (unsigned)b.n_regions), dim3(256), 0, st, in, o); PLAT_KT_END(ctx, PLAT_KT_SB_READS, st); }
    PLAT_HIP(ctx, hipGetLastError());
    return PLAT_OK;
}
