// plat_align.hip -- device path for Haplotype.alignReads / alignSingleRead.
//
// Pipeline of plat_align_window_batch (one stream, two small host read-backs):
//   k_validate    input checks + maxima (chaplotype.pyx:180-183 length rule), per-window read-tile shape
//   k_tile_scan   exclusive scan of the per-window tile sizes
//   k_prep_reads  one workgroup per (window, 64 reads): what EVERY read needs and nothing else -- its 2-bit base codes as
//                 transposed bit planes (= the rolling 7-mer codes of a5 hashReadForMapping, calign.pyx:155-165) and a
//                 descriptor (length, skip rule, mapq, plain-ACGT flag, quality sum / minimum / count below Q20).
//                 (Rounds 1-2 also wrote a tile of pre-converted 4-byte DP words per base of every read; since the
//                 seeding proofs only one pair in eight reaches the DP, which now builds its words from the bytes.)
//   k_seed        one workgroup per haplotype: bytes staged in LDS, gap-open annotation (a7, chaplotype.pyx:552-590)
//                 written as one byte per haplotype position, 7-mer index in LDS (a4, calign.pyx:94-124),
//                 then one wave per read: diagonal vote (calign.pyx:206-220) with 16-bit LDS counters and the
//                 arg-max candidate list in ascending order (calign.pyx:222-233) -> DP jobs
//   k_dp_jobs     one lane per banded DP (a1, align.c:77-586), see dp_core.hpp
//   k_finalize    per (read, haplotype): the reference's candidate selection replayed on the job scores
//                 (calign.pyx:235-267), score -> log-likelihood (a8, chaplotype.pyx:621-676)
#include "dp_core.hpp"
#include <algorithm>
#include "dp_unpacked.hpp"
#include "dp_traceback.hpp"
#include <stdio.h>

#include "plat_internal.hpp"

namespace plat {

// Job slots: pair p owns slot p for its first DP (99% of pairs need exactly one); further candidate DPs
// (tandem repeats: many arg-max diagonals) go to an overflow area behind the npairs primary slots, reserved
// with one global atomic per such pair.  (A single job counter bumped by every pair saturates one L2
// atomic unit: ~90 atomics/us, i.e. ~20 ms for 2M pairs -- measured in round 1.)
struct PairRec { int32_t extra_base, idx0; int16_t ncand, orig_k; uint8_t mapq, pad[3]; };   // ncand: -1 skipped, -2 read < 7 bp, -3 exact match (idx0 = read length), -4 ungapped alignment proven optimal (idx0 = read length, extra_base = score)
struct Job { uint32_t col; int32_t hap, idx, len; };     // col = index of the read in the batch; len 0 = empty slot;
                                                         // hap bit 30 (JOB_BIGQ): the read's quality sum forbids the 32-bit SWAR adds
constexpr int32_t JOB_BIGQ = 1 << 30;
__device__ __forceinline__ int job_hap(const Job& j) { return j.hap & (JOB_BIGQ - 1); }
// per-read descriptor: index of the read in the batch, mapping position, len | flags<<16 | mapq<<24
// (flags bit0: skipped by the QCFail / overlap < 7 rule; bit1: the read holds a byte other than A, C, G, T;
//  bit2: quality sum above DP_SWAR_MAX_QSUM -> its DPs use the packed 16-bit adds)
struct ReadInfo { uint32_t col, aux; int32_t pos; uint32_t lfm; };      // aux bits 0..15: number of bases with quality < LOWQ (the ungapped proof)
constexpr unsigned LOWQ = 20u;
enum { SHORTCUT_UNGAPPED = 1, SHORTCUT_EXACT = 2, SHORTCUT_NLOW = 4, SHORTCUT_BIGQ = 8, SEED_LEAN = 1024, SEED_XCD = 2048 };   // what k_seed may finish without a DP (PLAT_NO_UNGAPPED / PLAT_NO_EXACT
                                                                         // switch them off; PLAT_NO_NLOW values unique windows by the smallest quality only)
__device__ __forceinline__ long long job_slot(long long pair, long long npairs, int extra_base, int k) {
    return k == 0 ? pair : npairs + extra_base + (k - 1);
}

enum { CNT_ERR = 0, CNT_MAXHAP, CNT_MAXREAD, CNT_NEXTRA, CNT_PAIRS_ALIGNED, CNT_NDP_REF, CNT_CELLS_REF, CNT_CELLS_RUN,
       CNT_NJOBS_RUN, CNT_TILE_TOTAL, CNT_SLOW_SEED, CNT_MAXH, CNT_NDENSE, CNT_HAPBLOB, CNT_NPAIRS, CNT_READBLOB, CNT_T0, CNT_T1, CNT_T2, CNT_T3, CNT_T4, CNT_NWAVES, CNT_N };
static_assert(CNT_N <= 64, "the pinned read-back area holds 64 words");

// The dense list of live DP job slots, built where the jobs are made (k_seed / k_seed_slow): DENSE_SEGS segments of `segcap`
// entries, one counter each, a whole window's jobs in one segment (w % DENSE_SEGS); a wave reserves room for its live jobs
// with ONE atomic.  The counters sit 4 KB apart (different L2 channels: one address takes ~90 atomics/us) behind the
// other counters; k_dense_total copies them into cnt[] for the host and sums them.
constexpr int DENSE_SEGS = 8;
constexpr int DENSE_CNT_STRIDE = 512;                    // in long longs
constexpr int CNT_DP_TILE = 60;                           // k_dp_jobs' tile counter (dynamic tiles)
constexpr int CNT_AREA = 64 + DENSE_SEGS * DENSE_CNT_STRIDE;   // long longs reserved (and zeroed) for counters per batch
__device__ __forceinline__ long long* dense_counter(long long* cnt, int seg) { return cnt + 64 + seg * DENSE_CNT_STRIDE; }
__device__ __forceinline__ long long dense_count(const long long* cnt, int seg, long long segcap) {
    const long long c = cnt[64 + seg * DENSE_CNT_STRIDE];
    return c < segcap ? c : segcap;
}
// flat index g over the concatenation of the segments -> job slot, -1 past the end
__device__ __forceinline__ long long dense_slot(const int32_t* __restrict__ dense, long long segcap, const long long* __restrict__ cnt, long long g) {
#pragma unroll
    for (int k = 0; k < DENSE_SEGS; ++k) {
        const long long c = dense_count(cnt, k, segcap);
        if (g < c) return (long long)dense[k * segcap + g];
        g -= c;
    }
    return -1;
}

__constant__ signed char c_homopol_go[49] = {   // homopolq[i]-'!' (chaplotype.pyx:64-67); see tests/test_oracle.py
    45, 42, 41, 39, 37, 32, 28, 23, 20, 19, 17, 16, 15, 14, 13, 12, 11, 11, 10, 9, 9, 8, 8, 7, 7, 7, 6, 6, 6, 5, 5, 5,
    4, 4, 4, 3, 3, 3, 3, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1};

__device__ __forceinline__ double loglik_of(int score, const double* __restrict__ mapq_lut, int mapq) {
    const double v = -0.23025850929940459 * (double)score + mapq_lut[mapq];   // chaplotype.pyx:676 (no FMA: -ffp-contract=off)
    return v > -300.0 ? v : -300.0;
}

__device__ __forceinline__ void set_err(long long* cnt, int code) {
    atomicCAS((unsigned long long*)&cnt[CNT_ERR], 0ull, (unsigned long long)(long long)code);
}

// ------------------------------------------------------------------------------------------------
// win_rows[w] = max read length in window w + 8 (rows of its read tile); hap_win[h] = window of haplotype h
__global__ void __launch_bounds__(256)
k_validate(plat_window_batch b, long long* cnt, int32_t* __restrict__ hap_win, int32_t* __restrict__ win_rows, int calc_flank)
{
    __shared__ int s_max[3];
    if (threadIdx.x < 3) s_max[threadIdx.x] = 0;
    __syncthreads();
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    const int lane = threadIdx.x & 63, gwave = tid >> 6, nwaves = nt >> 6;
    int maxhap = 0, maxread = 0, maxR = 0;
    for (int h = tid; h < b.n_haps; h += nt) {
        long long len = b.hap_off[h + 1] - b.hap_off[h];
        if (len > 16384) set_err(cnt, PLAT_ERR_HAP_TOO_LONG);
        if (len < 0) set_err(cnt, PLAT_ERR_BAD_INPUT);
        maxhap = max(maxhap, (int)min(len, 1ll << 20));
    }
    // one wave per window: shape checks, haplotype -> window map, longest read
    for (int w = gwave; w < b.n_windows; w += nwaves) {
        const int h0 = b.win_hap_begin[w], h1 = b.win_hap_begin[w + 1];
        const int r0 = b.win_read_begin[w], r1 = b.win_read_begin[w + 1];
        const long long H = h1 - h0, R = r1 - r0;
        if (H < 0 || R < 0 || b.pair_off[w + 1] - b.pair_off[w] != H * R) {
            if (lane == 0) { set_err(cnt, PLAT_ERR_BAD_INPUT); win_rows[w] = 8; }
            continue;
        }
        if (H * R >= 0x7FFFFF00ll && lane == 0) set_err(cnt, PLAT_ERR_OVERFLOW);       // (k_pairs numbers a window's pairs in 31 bits)
        // --calculateFlankScore=1 with hapFlank == 0 dereferences a NULL alignment buffer in the reference (calign.pyx:199-202,261-264)
        if (calc_flank && b.win_flank[w] <= 0 && lane == 0) set_err(cnt, PLAT_ERR_UNSUPPORTED);
        for (int h = h0 + lane; h < h1; h += 64) hap_win[h] = w;
        int lm = 0;
        for (int r = r0 + lane; r < r1; r += 64) {
            long long len = b.read_off[r + 1] - b.read_off[r];
            if (len < 0 || len > 32767) { set_err(cnt, PLAT_ERR_BAD_INPUT); len = 0; }      // cAlignedRead.rlen is a short
            lm = max(lm, (int)len);
        }
#pragma unroll
        for (int s2 = 32; s2 > 0; s2 >>= 1) lm = max(lm, __shfl_xor(lm, s2));
        if (lane == 0) win_rows[w] = lm + 8;
        maxread = max(maxread, lm);
        maxR = max(maxR, (int)min(R, 1ll << 30));
    }
    // (the 7-bit ASCII check of the blobs is done where the bytes are read anyway: k_prep_reads and k_seed)
    atomicMax(&s_max[0], maxhap); atomicMax(&s_max[1], maxread); atomicMax(&s_max[2], maxR);
    __syncthreads();
    if (threadIdx.x == 0) {
        // thousands of workgroups hitting three words of one cache line serialise in the L2 (~90 atomics/us): look first,
        // most workgroups find their maxima already there
        const int idx[3] = {CNT_MAXHAP, CNT_MAXREAD, CNT_MAXH};
        for (int k = 0; k < 3; ++k)
            if ((long long)s_max[k] > __atomic_load_n(&cnt[idx[k]], __ATOMIC_RELAXED))
                atomicMax((unsigned long long*)&cnt[idx[k]], (unsigned long long)s_max[k]);
    }
}

// exclusive scan of rows*R per window -> tile_off (dwords); single workgroup
__global__ void __launch_bounds__(1024)
k_tile_scan(plat_window_batch b, const int32_t* __restrict__ win_rows, long long* __restrict__ tile_off, long long* cnt,
            plat_batch_hints hints, int check_hints, int32_t* __restrict__ wave_win, int32_t* __restrict__ wave_first, long long wave_cap)
{
    __shared__ long long part[1024];
    __shared__ long long part2[1024];
    const int t = threadIdx.x, nt = blockDim.x;
    const int per = (b.n_windows + nt - 1) / nt;
    const int w0 = min(b.n_windows, t * per), w1 = min(b.n_windows, w0 + per);
    // waves of k_pairs per window: ceil(H R / pairs per wave) (plat_align.hip, "the seeding stage as TWO kernels")
    auto nwaves_of = [&](int w) -> long long {
        const long long R = b.win_read_begin[w + 1] - b.win_read_begin[w], H = b.win_hap_begin[w + 1] - b.win_hap_begin[w];
        if (R <= 0 || H <= 0) return 0;
        return R >= 13 ? (H * R + 63) >> 6 : (long long)(((unsigned)H + 4u) / 5u);     // 64 pairs per wave, or 5 whole haplotypes (5 R pairs)
    };
    // Up to TS_KEEP windows per thread (batches of up to 16 k windows): their figures are loaded ONCE, all loads in flight together, and kept
    // in registers across the scan -- this single workgroup sits on the batch's critical path and its time was two chains of `per`
    // dependent round trips.  Larger batches take the loops.
    constexpr int TS_KEEP = 16;
    const bool keep = per <= TS_KEEP;
    long long ts[TS_KEEP];
    int nws[TS_KEEP];
    long long s = 0, s2 = 0;
    if (keep) {
#pragma unroll
        for (int k = 0; k < TS_KEEP; ++k) {
            const int w = w0 + k;
            ts[k] = 0; nws[k] = 0;
            if (w < w1) {
                ts[k] = ((long long)win_rows[w] * (b.win_read_begin[w + 1] - b.win_read_begin[w]) + 3) & ~3ll;
                nws[k] = (int)min(nwaves_of(w), 0x7FFFFFFFll);
            }
        }
#pragma unroll
        for (int k = 0; k < TS_KEEP; ++k) { s += ts[k]; s2 += nws[k]; }
    } else {
        for (int w = w0; w < w1; ++w) {
            s += ((long long)win_rows[w] * (b.win_read_begin[w + 1] - b.win_read_begin[w]) + 3) & ~3ll;
            s2 += nwaves_of(w);
        }
    }
    part[t] = s; part2[t] = s2;
    __syncthreads();
    for (int d = 1; d < nt; d <<= 1) {
        long long v = t >= d ? part[t - d] : 0, v2 = t >= d ? part2[t - d] : 0;
        __syncthreads();
        part[t] += v; part2[t] += v2;
        __syncthreads();
    }
    long long run = part[t] - s, run2 = part2[t] - s2;
    auto emit = [&](int w, long long tsz, long long n) {
        tile_off[w] = run;
        run += tsz;
        if (wave_win) {
            wave_first[w] = (int32_t)min(run2, 0x7FFFFFFFll);
            for (long long k = 0; k < n && run2 + k < wave_cap; ++k) wave_win[run2 + k] = w;
            run2 += n;
        }
    };
    if (keep) {
#pragma unroll
        for (int k = 0; k < TS_KEEP; ++k) if (w0 + k < w1) emit(w0 + k, ts[k], nws[k]);
    } else {
        for (int w = w0; w < w1; ++w)
            emit(w, ((long long)win_rows[w] * (b.win_read_begin[w + 1] - b.win_read_begin[w]) + 3) & ~3ll, nwaves_of(w));
    }
    if (t == nt - 1) {
        cnt[CNT_NWAVES] = part2[t] <= wave_cap ? part2[t] : 0;
        if (wave_win && part2[t] > wave_cap) set_err(cnt, PLAT_ERR_OVERFLOW);       // (cannot happen: the host's bound covers every window shape)
        cnt[CNT_TILE_TOTAL] = part[t];
        // everything the host needs before it can size the scratch buffers travels in ONE read-back of cnt[]
        cnt[CNT_HAPBLOB] = b.hap_off[b.n_haps];
        cnt[CNT_NPAIRS] = b.pair_off[b.n_windows];
        cnt[CNT_READBLOB] = b.read_off[b.n_reads];
        // asynchronous mode: the host sized every buffer and launch from the caller's hints; anything they do not cover
        // stops the pipeline here (every later kernel returns at once when the error word is set)
        if (check_hints && (cnt[CNT_MAXHAP] > hints.max_hap_len || cnt[CNT_MAXREAD] > hints.max_read_len ||
                            cnt[CNT_MAXH] > hints.max_reads_per_window || cnt[CNT_NPAIRS] != hints.n_pairs ||
                            cnt[CNT_HAPBLOB] > hints.hap_blob_len || cnt[CNT_READBLOB] > hints.read_blob_len))
            set_err(cnt, PLAT_ERR_BAD_HINTS);
    }
}

// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned base2(unsigned ch) {      // calign.pyx:69-74
    unsigned c = ch & 7u;
    if (c == 7u) c = 2u;
    return c & 3u;
}

constexpr int PREP_LMAX = 448;          // reads up to this length are staged through LDS (64 reads x seq+qual <= 56 KB)

__global__ void __launch_bounds__(256)
k_prep_reads(plat_window_batch b, const int32_t* __restrict__ win_rows, const long long* __restrict__ tile_off,
             uint16_t* __restrict__ codes, ReadInfo* __restrict__ rinfo, long long* cnt, int qoff, int xcd)
// qoff = byte offset of the quality image in the dynamic LDS (= 64 * min(longest read, PREP_LMAX) + 32); the bit-plane
// accumulators of staged windows follow at 2 * qoff (1 KB per 64-base chunk).
// `codes` holds, per window and in the tile's footprint (2 bytes per tile element), the reads' 2-bit base codes
// (calign.pyx:69-74 coding: A=1 C=3 G=2 T=0, N=2) as two BIT PLANES, 64 bases per 64-bit word, transposed (see below).
// A 7-mer code (a5, hashReadForMapping calign.pyx:155-165) is 7 consecutive bits of plane 0 and 7 of plane 1; only
// equality of codes matters to the vote, so this bit order is as good as the reference's.
// grid = (windows, groups of 64 reads).  The 64 reads of a group are contiguous in the blobs: they are copied to LDS
// with coalesced loads and transposed from there (a direct strided gather thrashes the L1 for windows with
// thousands of reads).
{
    extern __shared__ __attribute__((aligned(16))) unsigned char psm[];
    __shared__ int s_off[65];
    __shared__ unsigned s_dirty[2];                      // bit rl: read rl of the group holds a byte other than A, C, G, T
    __shared__ unsigned s_qsum[64];                      // sum of the base qualities of read rl (picks the DP's add flavour)
    __shared__ unsigned s_qmin[64];                      // smallest base quality of read rl (k_seed's ungapped-alignment proof)
    __shared__ unsigned s_nlow[64];                      // number of bases of read rl with quality < LOWQ (same proof)
    // (grid.x a multiple of 8: XCD x -- workgroups go to the XCDs round robin -- takes the x-th eighth of the windows, the eighth whose
    // haplotypes k_seed gives to the same XCD: what this kernel writes is what that one reads, and some of it is still in that L2)
    int w = blockIdx.x;
    if (xcd) {
        w = (int)(blockIdx.x & 7u) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
        if (w >= b.n_windows) return;
    }
    const int rb = b.win_read_begin[w], R = b.win_read_begin[w + 1] - rb;
    if (cnt[CNT_ERR] != 0) return;                       // an earlier stage refused the batch
    const int c0 = (int)blockIdx.y * 64;
    if (c0 >= R) return;
    const int nr = min(64, R - c0);
    const int rows = win_rows[w];
    const long long toff = tile_off[w];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const long long blob0 = b.read_off[rb + c0];
    const int nbytes = (int)(b.read_off[rb + c0 + nr] - blob0);
    const bool staged = rows - 8 <= PREP_LMAX;
    // the group's bytes are copied as ALIGNED 16-byte vectors; the LDS image keeps the blob's misalignment (mis = address & 15)
    const int misS = (int)((uintptr_t)(b.read_seq + blob0) & 15), misQ = (int)((uintptr_t)(b.read_qual + blob0) & 15);
    ReadInfo my_ri = ReadInfo{0u, 0u, 0, 0u};            // loaded now, stored at the end together with the "dirty" bit
    if (tid < nr) {
        const int wstart = b.win_start[w], wend = b.win_end[w];
        const int r = rb + c0 + tid;
        const int L = (int)(b.read_off[r + 1] - b.read_off[r]);
        // skip rule, chaplotype.pyx:343-346 / 358-361 (brokenMates are always aligned, :366-373)
        int skip = 0;
        if (b.read_kind[r] != 2) {
            const int os = max(wstart, b.read_pos[r]), oe = min(wend, b.read_end[r]);
            const int ov = oe > os ? oe - os : -1;                       // chaplotype.pyx:103-115
            skip = (b.read_flags[r] & 512) || ov < 7;
        }
        my_ri = ReadInfo{(uint32_t)r, 0u, b.read_pos[r],
                         (uint32_t)L | ((uint32_t)skip << 16) | ((uint32_t)b.read_mapq[r] << 24)};
    }
    if (tid <= nr) s_off[tid] = (int)(b.read_off[rb + c0 + tid] - blob0);
    {   // copy to LDS + 7-bit ASCII check (the DP packs bases as byte << 9 and qualities as 4*q in 16 bits).  The bytes before blob0 /
        // after the group inside the first / last vector belong to neighbouring reads or to nobody (same 16-byte granule, hence same
        // page, as a byte of the group): they are copied and not looked at.  A thread has up to four vectors of each blob in flight
        // before it waits for the first (the copy is one round trip to memory for a group of 64 x 150 bases, not one per vector).
        const uint4* gs16 = (const uint4*)(b.read_seq + blob0 - misS);
        const uint4* gq16 = (const uint4*)(b.read_qual + blob0 - misQ);
        const int nvS = (misS + nbytes + 15) >> 4, nvQ = (misQ + nbytes + 15) >> 4, nvM = max(nvS, nvQ);
        auto edge = [](uint4 v, int i, int nv, int mis, int nb) -> unsigned {      // the vector's bytes that belong to the group, OR-ed
            const int lo = i == 0 ? mis : 0, hi = i == nv - 1 ? ((mis + nb - 1) & 15) + 1 : 16;      // bytes [lo, hi) of this vector
            const unsigned c[4] = {v.x, v.y, v.z, v.w};
            unsigned acc = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int a = max(lo - 4 * k, 0), e = min(hi - 4 * k, 4);
                if (e > a) acc |= c[k] & (unsigned)(((1ull << (8 * e)) - 1ull) & ~((1ull << (8 * a)) - 1ull));
            }
            return acc;
        };
        unsigned bad = 0;
        for (int i0 = tid; i0 < nvM; i0 += 4 * nthr) {
            uint4 vs[4], vq[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = i0 + k * nthr;
                vs[k] = i < nvS ? gs16[i] : make_uint4(0u, 0u, 0u, 0u);
                vq[k] = i < nvQ ? gq16[i] : make_uint4(0u, 0u, 0u, 0u);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = i0 + k * nthr;
                if (i < nvS) {
                    bad |= (i == 0 || i == nvS - 1) ? edge(vs[k], i, nvS, misS, nbytes) : (vs[k].x | vs[k].y | vs[k].z | vs[k].w);
                    if (staged) ((uint4*)psm)[i] = vs[k];
                }
                if (i < nvQ) {
                    bad |= (i == 0 || i == nvQ - 1) ? edge(vq[k], i, nvQ, misQ, nbytes) : (vq[k].x | vq[k].y | vq[k].z | vq[k].w);
                    if (staged) ((uint4*)(psm + qoff))[i] = vq[k];
                }
            }
        }
        if (bad & 0x80808080u) set_err(cnt, PLAT_ERR_BAD_INPUT);
    }
    if (tid < 2) s_dirty[tid] = 0u;
    if (tid < 64) { s_qsum[tid] = 0u; s_qmin[tid] = 255u; s_nlow[tid] = 0u; }
    unsigned* s_pl = (unsigned*)(psm + 2 * qoff);        // [chunk][plane][half][64 reads] bit-plane accumulators (staged windows)
    const int nchunks = (rows - 8 + 63) >> 6;
    if (staged)
        for (int i = tid; i < nchunks * 256; i += nthr) s_pl[i] = 0u;
    __syncthreads();
    const unsigned char* gs = b.read_seq + blob0;
    const unsigned char* gq = b.read_qual + blob0;
    // a thread takes 4 consecutive bases of one read rl (lanes run over rl).  e / nr by multiply-shift (exact for e < 16384, nr <= 64).
    {
        const int ngrp = (rows - 8 + 3) >> 2;            // groups of 4 bases of the window's longest read
        const int ne = ngrp * nr;
        const unsigned M = (1u << 22) / (unsigned)nr + 1u;
        for (int e = tid; e < ne; e += nthr) {
            const int g = ne <= 16384 ? (int)(((unsigned)e * M) >> 22) : e / nr;
            const int rl = e - g * nr;
            const int o = s_off[rl], L = s_off[rl + 1] - o;
            if (4 * g >= L) continue;
            bool dirty = false;
            unsigned qs = 0, qm = 255u, n0 = 0, n1 = 0, nl = 0;
            if (staged) {
                // the thread's 4 bases and 4 qualities as two dwords (unaligned reads of the LDS images), then 4 at a time
                const int nval = min(L - 4 * g, 4);
                const unsigned as_ = (unsigned)(misS + o + 4 * g), aq_ = (unsigned)(qoff + misQ + o + 4 * g);
                const uint32_t* P = (const uint32_t*)psm;
                const uint32_t keep = nval >= 4 ? 0xFFFFFFFFu : ((1u << (8 * nval)) - 1u);
                const uint32_t vs = __builtin_amdgcn_alignbyte(P[(as_ >> 2) + 1], P[as_ >> 2], as_ & 3u) & keep;
                const uint32_t vq = __builtin_amdgcn_alignbyte(P[(aq_ >> 2) + 1], P[aq_ >> 2], aq_ & 3u) & keep;
                // 2-bit codes (calign.pyx:69-74: c = ch & 7; 7 -> 2; c & 3), one per byte
                uint32_t x = vs & 0x07070707u;
                const uint32_t t7 = x & (x >> 1) & (x >> 2) & 0x01010101u;       // bytes that are 7
                x = (x ^ (t7 | (t7 << 2))) & 0x03030303u;                        // 7 ^ 5 = 2
                const uint32_t y0 = x & 0x01010101u, y1 = (x >> 1) & 0x01010101u;
                n0 = (y0 | (y0 >> 7) | (y0 >> 14) | (y0 >> 21)) & 0xFu;
                n1 = (y1 | (y1 >> 7) | (y1 >> 14) | (y1 >> 21)) & 0xFu;
                // plain base <=> the byte is the letter of its own code (code 0..3 = T, A, G, C)
                const uint32_t expect = __builtin_amdgcn_perm(0u, 0x43474154u, x);
                dirty = ((expect ^ vs) & keep) != 0u;
                qs = __builtin_amdgcn_sad_u8(vq, 0u, 0u);
                const uint32_t vqm = vq | ~keep;
                qm = min(min(vqm & 0xFFu, (vqm >> 8) & 0xFFu), min((vqm >> 16) & 0xFFu, vqm >> 24));
                // bytes >= LOWQ get bit 7 (7-bit qualities; the bytes past the read are 0xFF): the others are the low ones
                nl = 4u - (unsigned)__popc((((vqm & 0x7F7F7F7Fu) + 0x01010101u * (128u - LOWQ)) | vqm) & 0x80808080u);
            } else
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = 4 * g + j;
                if (i < L) {
                    const unsigned ch = gs[o + i];
                    const unsigned ql = gq[o + i];
                    qs += ql;
                    qm = min(qm, ql);
                    nl += ql < LOWQ;
                    const unsigned b2 = base2(ch);
                    n0 |= (b2 & 1u) << j;
                    n1 |= (b2 >> 1) << j;
                    const unsigned dch = ch - 65u;                               // 'A' 'C' 'G' 'T' = 65 + {0, 2, 6, 19}
                    dirty |= dch > 19u || !((0x80045u >> dch) & 1u);
                }
            }
            if (dirty) atomicOr(&s_dirty[rl >> 5], 1u << (rl & 31));
            if (qs) atomicAdd(&s_qsum[rl], qs);
            if (qm < s_qmin[rl]) atomicMin(&s_qmin[rl], qm);   // look first: most threads do not lower the minimum
            if (nl) atomicAdd(&s_nlow[rl], nl);
            if (staged) {                                   // the 4 bases of this thread: 4 bits of each plane, inside one 32-bit half
                const int c = (4 * g) >> 6, half = ((4 * g) >> 5) & 1, sh = (4 * g) & 31;
                if (n0) atomicOr(&s_pl[((c * 2 + 0) * 2 + half) * 64 + rl], n0 << sh);
                if (n1) atomicOr(&s_pl[((c * 2 + 1) * 2 + half) * 64 + rl], n1 << sh);
            }
        }
    }
    // bit planes: for every read and every chunk c of 64 bases, plane0 = bit 0 and plane1 = bit 1 of the 2-bit base code,
    // one bit per base; word (2c+plane) of read rl at rd2[(2c+plane)*R + rl].  Staged windows: the tile loop above has
    // OR-ed every thread's 4 bits into LDS; otherwise (reads too long for the LDS image) ballots over the 64 lanes.
    unsigned long long* rd2 = (unsigned long long*)(codes + toff);
    if (staged) {
        __syncthreads();
        for (int e = tid; e < nchunks * 2 * nr; e += nthr) {
            const int cp = e / nr, rl = e - cp * nr;      // cp = 2*chunk + plane
            const unsigned long long lo = s_pl[(cp * 2 + 0) * 64 + rl], hi = s_pl[(cp * 2 + 1) * 64 + rl];
            rd2[(long long)cp * R + c0 + rl] = lo | (hi << 32);
        }
    } else {
        const int lane = tid & 63, wv = tid >> 6, nwv = nthr >> 6;
        for (int c = wv; c < nchunks; c += nwv) {        // one wave per chunk; lane rl keeps read rl's two words
            unsigned long long my0 = 0, my1 = 0;
            const int i = 64 * c + lane;
            for (int rl = 0; rl < nr; ++rl) {
                const int o = s_off[rl], L = s_off[rl + 1] - o;
                const unsigned b2 = i < L ? base2(gs[o + i]) : 0u;
                const unsigned long long m0 = __ballot(b2 & 1u), m1 = __ballot(b2 & 2u);
                if (lane == rl) { my0 = m0; my1 = m1; }
            }
            if (lane < nr) {
                rd2[(long long)(2 * c) * R + c0 + lane] = my0;
                rd2[(long long)(2 * c + 1) * R + c0 + lane] = my1;
            }
        }
    }
    __syncthreads();
    if (tid < nr) {
        my_ri.lfm |= ((s_dirty[tid >> 5] >> (tid & 31)) & 1u) << 17;
        my_ri.lfm |= (s_qsum[tid] > (unsigned)DP_SWAR_MAX_QSUM ? 1u : 0u) << 18;
        my_ri.lfm |= min(s_qmin[tid], 31u) << 19;        // flags bits 3..7
        my_ri.aux = min(s_nlow[tid], 0xFFFFu);
        rinfo[rb + c0 + tid] = my_ri;
    }
}

// ------------------------------------------------------------------------------------------------
#define CNT16(c, j) (((c)[(j) >> 1] >> (16 * ((j) & 1))) & 0x7FFFu)
__device__ __forceinline__ unsigned tbl_slot(unsigned code, unsigned mask) { return (code * 40503u + (code >> 5)) & mask; }

// k-mer lookup: first haplotype position (+1) holding this code, 0 if none
__device__ __forceinline__ unsigned kmer_head(const unsigned* table, unsigned code, bool direct, unsigned tmask) {
    if (direct) return ((const unsigned short*)table)[code];           // direct mode: u16 head per 14-bit code
    const unsigned key = (code + 1u) << 16;
    unsigned slot = tbl_slot(code, tmask);
    unsigned e = table[slot];
    while (e != 0u && (e & 0xFFFF0000u) != key) { slot = (slot + 1u) & tmask; e = table[slot]; }
    return e & 0xFFFFu;
}

// k_seed: one workgroup per haplotype; ONE LANE PER (read, haplotype) PAIR.
// LDS carve (dynamic):  table u32[tsize] | next u16[maxhap+2] | planes u64[4][nw64] (h0, h1, eq, nu) |
//                       counts u16[nwaves][cw] | scalars
// The k-mer index has two modes: haplotypes up to 4096 bp use a small open-addressing table (load factor <= 0.8);
// longer ones (up to the reference's cap of 16384) index all 4^7 codes directly, as the reference does
// (calign.pyx:98-99).
//
// Everything is bit-parallel on BIT PLANES (one bit per base, 64 bases per word, built with wave ballots):
//   h0/h1 = the two bits of the haplotype's 2-bit base codes, eq = "byte equals its right neighbour" (gives the
//   homopolymer run lengths of annotateWithGapOpen by count-trailing-ones), nu = "the 7-mer starting here occurs more
//   than once in this haplotype".
// Every lane tries to PROVE that one diagonal d* is the unique arg-max of the reference's diagonal vote
// (calign.pyx:206-233) without counting votes: the read's planes (<= 256 bp: 4 x 2 words in registers) are XORed
// with the haplotype's planes shifted to the hypothesis diagonal; a shift-AND ladder marks every read position where 7
// consecutive bases match = a k-mer that votes for d*;  C = popcount of those marks.  Votes for any OTHER diagonal are
// at most  X = (#matching k-mers flagged nu) * (maxmult-1) + (#non-matching k-mers) * maxmult  (maxmult = largest
// 7-mer multiplicity in the haplotype), so X < C  =>  d* is the only candidate of calign.pyx:222-233.
// Hypothesis A = the read's mapping offset (calign.pyx:252); B = the diagonal of the read's first haplotype-unique k-mer.
// Pairs that cannot be decided (tandem repeats, ties, reads longer than 256 bp) fall back to the exact vote: the whole
// wave counts that pair's diagonals in 16-bit LDS counters (two per dword, 32-bit LDS atomics; bit 15 = claim flag
// that picks one representative lane per arg-max diagonal) and emits the candidates in ascending order.
typedef unsigned long long u64;
constexpr int SEED_CHUNKS = 4;
constexpr int UNG_KMAX = 4;           // mismatches on the candidate diagonal the ungapped-alignment proof takes on

__device__ __forceinline__ u64 funnel(u64 lo, u64 hi, int sh) { return sh ? (lo >> sh) | (hi << (64 - sh)) : lo; }

// 14-bit k-mer code at position i from two bit planes given as (lo, hi) word pairs
__device__ __forceinline__ unsigned plane_code(u64 a_lo, u64 a_hi, u64 b_lo, u64 b_hi, int sh) {
    return (unsigned)(funnel(a_lo, a_hi, sh) & 0x7Full) | ((unsigned)(funnel(b_lo, b_hi, sh) & 0x7Full) << 7);
}
// k-mer code of read position i from the transposed bit planes of one read (column pointer, row stride R)
__device__ __forceinline__ unsigned read_code(const u64* __restrict__ col, int R, int i) {
    const int c = i >> 6, sh = i & 63;
    const bool cross = sh > 57;
    const u64 a_lo = col[(long long)(2 * c) * R], b_lo = col[(long long)(2 * c + 1) * R];
    const u64 a_hi = cross ? col[(long long)(2 * c + 2) * R] : 0ull, b_hi = cross ? col[(long long)(2 * c + 3) * R] : 0ull;
    return plane_code(a_lo, a_hi, b_lo, b_hi, sh);
}

struct SlowRec { int32_t hap, rl; };     // a (haplotype, read-in-window) pair the bit-parallel proof left undecided

// The reference's diagonal vote for ONE pair (calign.pyx:206-233,252-267), worked by a whole wave: counts in 16-bit LDS
// counters (two per dword, 32-bit LDS atomics), candidates emitted in ascending diagonal order.  `counts` must be zero on entry and is zero again on exit.
__device__ __forceinline__ void seed_exact_vote(const unsigned* table, const unsigned short* nxt, unsigned* counts, bool direct,
                                                unsigned tmask, int hapLen, int h, const u64* scp, int R, int sL, int sidx0,
                                                unsigned scol, int smapq, long long spidx, long long npairs, int extra_cap,
                                                Job* __restrict__ jobs, PairRec* __restrict__ pairs, long long* cnt,
                                                int32_t* __restrict__ dense, long long segcap, int seg)
{
    const int lane = threadIdx.x & 63;
    const int n = hapLen + sL, snk = sL - 7, j0i = sidx0 + sL;
    // pass 1: diagonal vote, calign.pyx:209-220
    unsigned mymax = 0;
    for (int i = lane; i < snk; i += 64) {
        unsigned hidx = kmer_head(table, read_code(scp, R, i), direct, tmask);
        while (hidx != 0u) {
            const int j = (int)hidx - i - 1 + sL;
            const unsigned sh = 16u * (unsigned)(j & 1);
            const unsigned c = ((atomicAdd(&counts[j >> 1], 1u << sh) >> sh) & 0x7FFFu) + 1u;
            mymax = max(mymax, c);
            hidx = nxt[hidx];
        }
    }
#pragma unroll
    for (int s2 = 32; s2 > 0; s2 >>= 1) mymax = max(mymax, (unsigned)__shfl_xor((int)mymax, s2));
    const unsigned maxcount = mymax;
    const bool s_orig_in = maxcount > 0 && j0i >= 0 && j0i < n && CNT16(counts, j0i) == maxcount && sidx0 + sL + 15 < hapLen;
    // the arg-max diagonals that may be aligned (calign.pyx:228), found by scanning this read's counters
    int sncand = 0, myidx = 0x7FFFFFFF;
    for (int j0 = 0; j0 < n; j0 += 64) {
        const int j = j0 + lane;
        const bool is = maxcount > 0 && j < n && CNT16(counts, j) == maxcount && (j - sL) + sL + 15 < hapLen;
        const unsigned long long bal = __ballot(is);
        if (bal && myidx == 0x7FFFFFFF) myidx = j0 + (int)__ffsll((long long)bal) - 1 - sL;
        sncand += __popcll(bal);
    }
    const int njobs = sncand + (s_orig_in ? 0 : 1);
    int sbase = 0;
    if (njobs > 1) {
        if (lane == 0) sbase = (int)atomicAdd((unsigned long long*)&cnt[CNT_NEXTRA], (unsigned long long)(njobs - 1));
        sbase = __shfl(sbase, 0);
    }
    const bool fits = njobs == 1 || (long long)sbase + (njobs - 1) <= (long long)extra_cap;
    int orig_k = sncand;
    if (sncand == 1) {
        if (lane == 0) jobs[spidx] = Job{scol, h, myidx, sL};
        if (s_orig_in) orig_k = 0;
    } else if (sncand > 1) {
        // ordered emission (ascending diagonal, calign.pyx:223) by scanning this read's counters
        int k = 0;
        for (int j0 = 0; j0 < n; j0 += 64) {
            const int j = j0 + lane;
            const bool is = j < n && CNT16(counts, j) == maxcount && (j - sL) + sL + 15 < hapLen;
            const unsigned long long bal = __ballot(is);
            if (is) {
                const int mypos = k + __popcll(bal & ((1ull << lane) - 1ull));
                if (fits || mypos == 0) jobs[job_slot(spidx, npairs, sbase, mypos)] = Job{scol, h, j - sL, sL};
            }
            if (s_orig_in && j0i >= j0 && j0i < j0 + 64) orig_k = k + __popcll(bal & ((1ull << (j0i - j0)) - 1ull));
            k += __popcll(bal);
        }
    }
    if (lane == 0) {
        if (!s_orig_in && (fits || sncand == 0)) jobs[job_slot(spidx, npairs, sbase, sncand)] = Job{scol, h, sidx0, sL};
        pairs[spidx] = PairRec{sbase, sidx0, (int16_t)sncand, (int16_t)orig_k, (uint8_t)smapq, {0, 0, 0}};
    }
    // the pair's job slots join the dense list (when they do not fit the batch is re-run or refused anyway)
    if (fits) {
        long long db = 0;
        if (lane == 0) db = (long long)atomicAdd((unsigned long long*)dense_counter(cnt, seg), (unsigned long long)njobs);
        db = ((long long)(unsigned)__shfl((int)db, 0)) | ((long long)__shfl((int)(db >> 32), 0) << 32);
        if (db + njobs <= segcap)
            for (int k = lane; k < njobs; k += 64) dense[seg * segcap + db + k] = (int32_t)job_slot(spidx, npairs, sbase, k);
    }
    // all counters back to zero (cheaper than walking the chains again)
    for (int j = lane; j < ((n + 2) >> 1); j += 64) counts[j] = 0u;
}

// The vote of seed_exact_vote WITHOUT its reservations (round 6): the arg-max diagonals of one pair are left, ascending, in an LDS record
// (at most SLOW_MAXC of them; `sncand` says how many there are), so that a workgroup can reserve the job slots and dense-list entries of a
// whole group of pairs with ONE returning atomic per counter -- ~11 k pairs per launch of the WGS job each asked the same two addresses,
// and the L2 serves ~90 of those per microsecond.  `counts` zero on entry and on exit.
constexpr int SLOW_MAXC = 32;
struct SlowOut {
    long long pidx; int32_t h, sL, sidx0; uint32_t scol; int32_t smapq, sncand, orig_in, orig_k, seg, njobs, sbase, fits; long long db;
    uint16_t cand[SLOW_MAXC];
};
__device__ __forceinline__ void seed_vote_collect(const unsigned* table, const unsigned short* nxt, unsigned* counts, bool direct,
                                                  unsigned tmask, int hapLen, const u64* scp, int R, int sL, int sidx0, SlowOut* o)
{
    const int lane = threadIdx.x & 63;
    const int n = hapLen + sL, snk = sL - 7, j0i = sidx0 + sL;
    unsigned mymax = 0;
    for (int i = lane; i < snk; i += 64) {                                 // pass 1: diagonal vote, calign.pyx:209-220
        unsigned hidx = kmer_head(table, read_code(scp, R, i), direct, tmask);
        while (hidx != 0u) {
            const int j = (int)hidx - i - 1 + sL;
            const unsigned sh = 16u * (unsigned)(j & 1);
            const unsigned c = ((atomicAdd(&counts[j >> 1], 1u << sh) >> sh) & 0x7FFFu) + 1u;
            mymax = max(mymax, c);
            hidx = nxt[hidx];
        }
    }
#pragma unroll
    for (int s2 = 32; s2 > 0; s2 >>= 1) mymax = max(mymax, (unsigned)__shfl_xor((int)mymax, s2));
    const unsigned maxcount = mymax;
    const bool s_orig_in = maxcount > 0 && j0i >= 0 && j0i < n && CNT16(counts, j0i) == maxcount && sidx0 + sL + 15 < hapLen;
    int k = 0, orig_k = -1;
    for (int j0 = 0; j0 < n; j0 += 64) {                                   // the arg-max diagonals that may be aligned (calign.pyx:228), ascending (:223)
        const int j = j0 + lane;
        const bool is = maxcount > 0 && j < n && CNT16(counts, j) == maxcount && (j - sL) + sL + 15 < hapLen;
        const unsigned long long bal = __ballot(is);
        if (is) { const int at = k + __popcll(bal & ((1ull << lane) - 1ull)); if (at < SLOW_MAXC) o->cand[at] = (uint16_t)j; }
        if (s_orig_in && j0i >= j0 && j0i < j0 + 64) orig_k = k + __popcll(bal & ((1ull << (j0i - j0)) - 1ull));
        k += __popcll(bal);
    }
    for (int j = lane; j < ((n + 2) >> 1); j += 64) counts[j] = 0u;        // all counters back to zero
    if (lane == 0) { o->sncand = k; o->orig_in = s_orig_in ? 1 : 0; o->orig_k = s_orig_in ? orig_k : k; o->njobs = k + (s_orig_in ? 0 : 1); }
}
// ... and what seed_exact_vote writes for the pair, from the record: sbase / fits / db were reserved for the whole group by the caller
__device__ __forceinline__ void seed_vote_emit(const SlowOut* o, long long npairs, Job* __restrict__ jobs, PairRec* __restrict__ pairs,
                                               int32_t* __restrict__ dense, long long segcap)
{
    const int lane = threadIdx.x & 63;
    const int sncand = o->sncand, sL = o->sL, h = o->h, sbase = o->sbase, njobs = o->njobs;
    const bool fits = o->fits != 0, s_orig_in = o->orig_in != 0;
    const long long spidx = o->pidx;
    const uint32_t scol = o->scol;
    for (int k = lane; k < sncand; k += 64)
        if (fits || k == 0) jobs[job_slot(spidx, npairs, sbase, k)] = Job{scol, h, (int)o->cand[k] - sL, sL};
    if (lane == 0) {
        if (!s_orig_in && (fits || sncand == 0)) jobs[job_slot(spidx, npairs, sbase, sncand)] = Job{scol, h, o->sidx0, sL};
        pairs[spidx] = PairRec{sbase, o->sidx0, (int16_t)sncand, (int16_t)o->orig_k, (uint8_t)o->smapq, {0, 0, 0}};
    }
    if (fits && o->db >= 0 && o->db + njobs <= segcap)
        for (int k = lane; k < njobs; k += 64) dense[o->seg * segcap + o->db + k] = (int32_t)job_slot(spidx, npairs, sbase, k);
}

// a4: the haplotype's k-mer index (hash_sequence_multihit, calign.pyx:94-124) in LDS: positions 0..hapLen-8, entry =
// (code+1)<<16 | (pos+1) in an open-addressing table (or u16 heads indexed by code when direct), equal codes chained
// through nxt[] from the LAST position to the first.  exact_mult: also walk the chains for the largest multiplicity.
__device__ __forceinline__ void seed_build_index(unsigned* table, unsigned short* nxt, const u64* h0, const u64* h1, u64* nup,
                                                 int* s_scal, int hapLen, int nch, bool direct, int tsize, unsigned tmask,
                                                 bool exact_mult)
{
    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = nthr >> 6;
    for (int i = tid; i < (direct ? tsize / 2 : tsize); i += nthr) table[i] = 0u;
    __syncthreads();
    for (int t = wave; t < nch; t += nw) {
        const int p = 64 * t + lane;
        if (p < hapLen - 7) {
            const unsigned code = plane_code(h0[t], h0[t + 1], h1[t], h1[t + 1], lane);
            if (direct) {                               // u16 heads, two per dword: exchange one half with a CAS loop
                const unsigned sh = 16u * (code & 1u);
                unsigned cur = table[code >> 1], seen;
                do {
                    seen = cur;
                    cur = atomicCAS(&table[code >> 1], seen, (seen & ~(0xFFFFu << sh)) | ((unsigned)(p + 1) << sh));
                } while (cur != seen);
                nxt[p + 1] = (unsigned short)((seen >> sh) & 0xFFFFu);
            } else {
                const unsigned key = (code + 1u) << 16;
                unsigned slot = tbl_slot(code, tmask);
                unsigned e = table[slot];
                for (;;) {
                    if (e == 0u) {
                        unsigned old = atomicCAS(&table[slot], 0u, key | (unsigned)(p + 1));
                        if (old == 0u) { nxt[p + 1] = 0; break; }
                        e = old;
                    }
                    if ((e & 0xFFFF0000u) == key) {
                        unsigned old = atomicCAS(&table[slot], e, key | (unsigned)(p + 1));
                        if (old == e) { nxt[p + 1] = (unsigned short)(e & 0xFFFFu); break; }
                        e = old;
                    } else {
                        slot = (slot + 1u) & tmask;
                        e = table[slot];
                    }
                }
            }
        }
    }
    __syncthreads();
    if (!exact_mult) return;
    // only a chain HEAD walks its chain; the longest chain gives maxmult
    for (int t = wave; t < nch; t += nw) {
        const int p = 64 * t + lane;
        if (p < hapLen - 7) {
            const unsigned code = plane_code(h0[t], h0[t + 1], h1[t], h1[t + 1], lane);
            const unsigned hd = kmer_head(table, code, direct, tmask);
            if (hd == (unsigned)(p + 1) && nxt[hd] != 0u) {
                int c = 0;
                for (unsigned hh = hd; hh != 0u; hh = nxt[hh]) ++c;
                atomicMax(&s_scal[1], c);
            }
        }
    }
    __syncthreads();
}

struct SeedPlanes { u64 m0, m1, me; };
__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }

// One single-wave workgroup sweeps one haplotype: planes, gap-open bytes, multiplicity maps, nu plane, per-chunk gap-open minima, flags.
// Returns have_index (the counting table overflowed: the exact maximum came from the index, which then overwrote the maps).
__device__ __forceinline__ bool seed_sweep(unsigned* table, unsigned short* nxt, u64* h0, u64* h1, u64* eqp, u64* nup, int* s_scal, int nw64,
                                           const uint8_t* __restrict__ hs, int hapLen, long long hoff, bool first_group, uint8_t* __restrict__ gob,
                                           long long* cnt, int shortcuts, bool direct, int tsize, unsigned tmask, int nch, bool& stop)
{
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nw = nthr >> 6;
    stop = false;
    unsigned* seen1 = table;
    unsigned* seen2 = table + 512;
    if (tid < 3) s_scal[tid] = tid == 1;                 // has_n = 0, maxmult = 1, other-than-ACGTN = 0
    signed char* s_go = (signed char*)(s_scal + 4);      // LDS copy of the gap-open table
    unsigned* trip = (unsigned*)(s_scal + 4 + 16);       // [32] (code + 1) | (occurrences beyond the second) << 16
    unsigned char* s_gmin = (unsigned char*)(trip + 32); // [nw64] smallest gap-open penalty of each chunk of 64 haplotype positions
    if (tid < 49) s_go[tid] = c_homopol_go[tid];
    if (tid < 32) trip[tid] = 0u;
    for (int i = tid; i < 256; i += nthr) ((uint4*)table)[i] = make_uint4(0u, 0u, 0u, 0u);
    for (int i = tid; i < 4 * nw64; i += nthr) h0[i] = 0ull;            // h0, h1, eqp, nup are contiguous
    __syncthreads();
    // ---- passes A + B, one sweep over the haplotype in chunks of 64 bases (nw == 1: this wave sees every chunk).
    // A: planes by ballot.  B: a7 gap-open annotation (chaplotype.pyx:552-590): table[min(48, #following bytes equal to
    // this one)], 'N' -> table[0], written together with the base as the DP's haplotype word; multiplicity maps of the
    // k-mers at positions 0..hapLen-8 (the positions hash_sequence_multihit indexes, calign.pyx:109).
    // B of chunk t needs the planes of chunks t and t+1 (and the first byte of t+2): the bytes are loaded three chunks
    // ahead so that the global-load latency hides behind a whole iteration of work.
    int level = 1;
    {
        // (the loop is written without divergent branches: every `if` on a lane condition costs half a dozen scalar
        //  instructions of exec-mask bookkeeping, and this sweep had three scalar instructions for every vector one)
        auto ldb = [&](int t) -> unsigned { const int p = 64 * t + lane; return p < hapLen ? (unsigned)hs[p] : 0u; };
        typedef SeedPlanes Planes;
        u64 anyN = 0ull, anyOther = 0ull;                // wave-uniform: a byte 'N' / a byte other than A, C, G, T, N was seen
        unsigned accBits = 0u;                           // per lane: OR of its bytes (7-bit ASCII check after the loop)
        auto mk = [&](int t, unsigned c, unsigned cnext_chunk) -> Planes {
            const int p = 64 * t + lane;
            accBits |= c;
            unsigned cn = (unsigned)__shfl_down((int)c, 1);
            const unsigned first_next = (unsigned)__shfl((int)cnext_chunk, 0);
            cn = lane == 63 ? first_next : cn;
            const unsigned b2 = base2(c);                                // bytes past the end are 0 -> code 0
            Planes P;
            P.m0 = __ballot(p < hapLen && (b2 & 1u));
            P.m1 = __ballot(p < hapLen && (b2 & 2u));
            P.me = __ballot(p + 1 < hapLen && c == cn && c != (unsigned)'N');
            anyN |= __ballot(c == (unsigned)'N');
            anyOther |= __ballot(p < hapLen && c != 'A' && c != 'C' && c != 'G' && c != 'T' && c != 'N');
            return P;
        };
        unsigned b0 = ldb(0), b1 = ldb(1), b2_ = ldb(2);
        Planes P0 = mk(0, b0, b1), P1 = mk(1, b1, b2_);
        for (int t = 0; t < nch; ++t) {
            const unsigned b3 = ldb(t + 3);
            const int p = 64 * t + lane;
            {   // the chunk's three plane words: lanes 0..2 store one each (h0, h1, eqp lie nw64 words apart)
                const u64 val = lane == 0 ? P0.m0 : (lane == 1 ? P0.m1 : P0.me);
                if (lane < 3) h0[lane * nw64 + t] = val;
            }
            {
                const u64 v = funnel(P0.me, P1.me, lane);
                const int run = min(48, (int)__ffsll((long long)~v) - 1);     // trailing ones of v (v never has 64 ones beyond the cap)
                const int go = s_go[run < 0 ? 48 : run];
                if (p < hapLen && first_group) gob[hoff + p] = (uint8_t)go;       // localGapOpen[p]: the DP builds its haplotype words from it
            }
            {   // multiplicity maps of the k-mers at positions 0..hapLen-8: "seen", "seen twice"; lanes past the last k-mer OR in nothing
                const unsigned code = plane_code(P0.m0, P1.m0, P0.m1, P1.m1, lane);
                const unsigned wd = code >> 5, bit = p < hapLen - 7 ? 1u << (code & 31u) : 0u;
                const unsigned dup = atomicOr(&seen1[wd], bit) & bit;
                const unsigned third = atomicOr(&seen2[wd], dup) & dup;
                level = max(level, dup ? 2 : 1);
                if (__any(third != 0u)) {                                   // third or later occurrence (rare): count it
                    if (third) {
                        const unsigned key = code + 1u;
                        unsigned slot = code & 31u;
                        int probes = 0;
                        for (;;) {
                            unsigned e = trip[slot];
                            if (e == 0u) {
                                e = atomicCAS(&trip[slot], 0u, key | (1u << 16));
                                if (e == 0u) break;
                            }
                            if ((e & 0xFFFFu) == key) { atomicAdd(&trip[slot], 1u << 16); break; }
                            slot = (slot + 1u) & 31u;
                            if (++probes == 32) { level = 0x7FFF; break; }     // table full: the exact maximum comes from the index
                        }
                    }
                }
            }
            b0 = b1; b1 = b2_; b2_ = b3;
            P0 = P1;
            P1 = mk(t + 2, b1, b2_);
        }
        if (accBits & 0x80u) set_err(cnt, PLAT_ERR_BAD_INPUT);             // 7-bit ASCII only (the DP packs bases as byte << 9)
        if (lane == 0) { if (anyN) s_scal[0] = 1; if (anyOther) s_scal[2] = 1; }
    }
    __syncthreads();
    // smallest gap-open penalty of every chunk of 64 positions = the table entry of the chunk's LONGEST run (the table never rises
    // with the run length): one lane per chunk, on the run plane: positions where k consecutive bits are set, k = 1, 2, ...
    for (int t = tid; t < nch; t += nthr) {
        const u64 e0 = eqp[t], e1 = eqp[t + 1];
        const int nvalid = min(64, hapLen - 64 * t);
        u64 left = nvalid >= 64 ? ~0ull : ((1ull << nvalid) - 1ull);
        int longest = 0;
        while (longest < 48) {
            left &= funnel(e0, e1, longest);
            if (left == 0ull) break;
            ++longest;
        }
        s_gmin[t] = (unsigned char)s_go[longest];
    }
    __syncthreads();
    if (shortcuts & 256) { stop = true; return false; }  // (measurement only, PLAT_SEED_DEBUG: the haplotype sweep alone)
    if (lane < 32) level = max(level, trip[lane] ? 2 + (int)(trip[lane] >> 16) : 0);
#pragma unroll
    for (int s2 = 32; s2 > 0; s2 >>= 1) level = max(level, __shfl_xor(level, s2));
    if (lane == 0 && level > 1) atomicMax(&s_scal[1], level);
    __syncthreads();
    // ---- pass C: nu = "this k-mer occurs more than once"
    for (int t = wave; t < nch; t += nw) {
        const int p = 64 * t + lane;
        bool dup = false;
        if (p < hapLen - 7) {
            const unsigned code = plane_code(h0[t], h0[t + 1], h1[t], h1[t + 1], lane);
            dup = (seen2[code >> 5] >> (code & 31u)) & 1u;
        }
        const u64 md = __ballot(dup);
        if (lane == 0) nup[t] = md;
    }
    __syncthreads();
    bool have_index = false;
    if (s_scal[1] >= 0x7FFF) {                           // counting table overflowed: the exact maximum comes from the index chains
        __syncthreads();
        if (tid == 0) s_scal[1] = 1;
        __syncthreads();
        seed_build_index(table, nxt, h0, h1, nup, s_scal, hapLen, nch, direct, tsize, tmask, true);
        have_index = true;
    }
    return have_index;
}

// ---- the haplotypes of a window share almost everything -----------------------------------------------------------------------
// They are the window's reference sequence with a few variants applied, so sweeping each of them repeats most of the work.
// k_seed_base sweeps the FIRST haplotype of every window once and leaves the result in global memory (per window, SEED_BASE_*
// below: flags, the "seen at least once" map, the four planes, the per-chunk gap-open minima; its gap-open bytes are in `gob`).
// k_seed then DERIVES a haplotype of the same length that differs from that base in at most SEED_MAXDIFF bases instead of sweeping it:
// the base's planes, with the chunks the differing bases touch rebuilt; gap-open bytes recomputed within 48 positions of them; and a
// CONSERVATIVE nu plane instead of multiplicity maps of its own:
//   * a 7-mer that does not contain a differing base keeps the base's flag.  If it is unique in the base it is unique here as long as
//     no new 7-mer equals it (checked: a new 7-mer whose code the base has seen at all sends the haplotype to the full sweep); if it is
//     flagged in the base it stays flagged, even when the copies that made it so are among the <= 7 per difference that disappeared;
//   * the <= 7 new 7-mers per differing base are unique (none is in the base, and they are checked against each other);
//   * the largest multiplicity is at most the base's.
// A flag too many only weakens the proofs (more votes granted to other diagonals, fewer unique windows): a pair may go to the DP that
// a full sweep would have finished, never the other way round -- the scores are the same.  The base haplotype itself is "derived" with
// no difference, i.e. loaded.
constexpr int SEED_MAXDIFF = 8;
constexpr int SEED_BASE_SCAL = 0, SEED_BASE_SEEN = 16, SEED_BASE_PLANES = 16 + 2048;      // byte offsets inside a window's record
__host__ __device__ __forceinline__ size_t seed_base_stride(int maxhap) {
    const size_t nw64 = (((size_t)maxhap + 63) >> 6) + 8;
    return (size_t)SEED_BASE_PLANES + 32 * nw64 + ((nw64 + 15) & ~(size_t)15);
}

__device__ __forceinline__ SeedPlanes seed_chunk_planes(int t, unsigned c, unsigned cnext_chunk, int hapLen, int lane) {
    const int p = 64 * t + lane;
    unsigned cn = (unsigned)__shfl_down((int)c, 1);
    const unsigned first_next = (unsigned)__shfl((int)cnext_chunk, 0);
    cn = lane == 63 ? first_next : cn;
    const unsigned b2 = base2(c);                                        // bytes past the end are 0 -> code 0
    SeedPlanes P;
    P.m0 = __ballot(p < hapLen && (b2 & 1u));
    P.m1 = __ballot(p < hapLen && (b2 & 2u));
    P.me = __ballot(p + 1 < hapLen && c == cn && c != (unsigned)'N');
    return P;
}

// Returns false when the haplotype has to be swept in full.  base: the window's record; hb / gob_base: the base haplotype's bytes and
// gap-open bytes; table .. s_gmin: this workgroup's LDS working set (as seed_sweep leaves it); write_gob: store the gap-open bytes.
__device__ __forceinline__ bool seed_derive(const unsigned char* __restrict__ base, const uint8_t* __restrict__ hb, const uint8_t* __restrict__ gob_base,
                                            unsigned* table, u64* h0, u64* h1, u64* eqp, u64* nup, int* s_scal, unsigned char* s_gmin, int nw64,
                                            const uint8_t* __restrict__ hs, int hapLen, bool is_base, bool write_gob, uint8_t* __restrict__ gob,
                                            long long* cnt)
{
    const int lane = threadIdx.x & 63;
    const int nch = (hapLen + 63) >> 6;                  // <= 64: haplotypes up to 4096 bases are derived
    const int* bscal = (const int*)(base + SEED_BASE_SCAL);
    const unsigned* bseen = (const unsigned*)(base + SEED_BASE_SEEN);
    const u64* bplanes = (const u64*)(base + SEED_BASE_PLANES);
    const unsigned char* bgmin = base + SEED_BASE_PLANES + 32 * (size_t)nw64;
    signed char* s_go = (signed char*)(s_scal + 4);
    auto ldb = [&](int t) -> unsigned { const int p = 64 * t + lane; return p < hapLen ? (unsigned)hs[p] : 0u; };
    auto ld4 = [&](const uint8_t* q) -> uint32_t { uint32_t v; __builtin_memcpy(&v, q, 4); return v; };
    // (everything that does not depend on the differences is requested first: the base's planes, minima and flags)
    u64 pl[2];
    pl[0] = lane < 4 * nw64 ? bplanes[lane] : 0ull;
    pl[1] = lane + 64 < 4 * nw64 ? bplanes[lane + 64] : 0ull;
    const unsigned char gm0 = lane < nch ? bgmin[lane] : (unsigned char)0;
    const int bs0 = bscal[0], bs1 = bscal[1], bs2 = bscal[2];
    // a. where it differs from the base, four bases per lane and trip (blobs are followed by PLAT_BLOB_PAD readable bytes); lane j gets
    // the position of the j-th differing base
    int ndiff = 0, mypos = -1;
    uint32_t acc = 0u;
    if (!is_base)
        for (int o0 = 0; o0 < hapLen; o0 += 1024) {      // four trips' loads in flight
            uint32_t xa[4], xd[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int o = o0 + 256 * k + 4 * lane;
                const bool in = o < hapLen;
                const uint32_t va = in ? ld4(hs + o) : 0u, vb = in ? ld4(hb + o) : 0u;
                const int nv = hapLen - o;                // bytes of this dword inside the haplotype
                const uint32_t keep = nv >= 4 ? 0xFFFFFFFFu : (nv <= 0 ? 0u : ((1u << (8 * nv)) - 1u));
                xa[k] = va & keep; xd[k] = (va ^ vb) & keep;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                acc |= xa[k];
                u64 m = __ballot(xd[k] != 0u);
                while (m) {                              // (wave-uniform; a handful of trips in the whole haplotype)
                    const int L = (int)__ffsll((long long)m) - 1;
                    m &= m - 1ull;
                    const uint32_t x = (uint32_t)__shfl((int)xd[k], L);
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if ((x >> (8 * q)) & 0xFFu) {
                            if (lane == ndiff) mypos = o0 + 256 * k + 4 * L + q;
                            ++ndiff;
                        }
                    if (ndiff > SEED_MAXDIFF) return false;
                }
            }
        }
    if (acc & 0x80808080u) set_err(cnt, PLAT_ERR_BAD_INPUT);               // 7-bit ASCII only (the DP packs bases as byte << 9)
    bool newN = false, newOther = false;
    if (ndiff) {
        const unsigned a = (lane < ndiff) ? (unsigned)hs[mypos] : (unsigned)'A';
        newN = __any(a == (unsigned)'N');
        newOther = __any(a != 'A' && a != 'C' && a != 'G' && a != 'T' && a != 'N');
    }
    // b. the base's planes (h0, h1, eq, nu are contiguous), the gap-open table
    if (lane < 4 * nw64) h0[lane] = pl[0];
    if (lane + 64 < 4 * nw64) h0[lane + 64] = pl[1];
    for (int i = lane + 128; i < 4 * nw64; i += 64) h0[i] = bplanes[i];
    if (lane < 49) s_go[lane] = c_homopol_go[lane];
    wave_sync();
    // c. plane words that change: the chunk of a differing base, and the chunk before it when the base is its first (eq looks one base
    // ahead); gap-open bytes that may change: the 48 positions in front of a changed run bit
    u64 redo = 0ull, godirty = 0ull;
    for (int j = 0; j < ndiff; ++j) {
        const int p = __shfl(mypos, j);
        redo |= 1ull << (p >> 6);
        if (p > 0) redo |= 1ull << ((p - 1) >> 6);
        for (int t = max(p - 49, 0) >> 6; t <= (p >> 6); ++t) godirty |= 1ull << t;
    }
    for (u64 m = redo; m; m &= m - 1ull) {
        const int t = (int)__ffsll((long long)m) - 1;
        const SeedPlanes Q = seed_chunk_planes(t, ldb(t), ldb(t + 1), hapLen, lane);
        const u64 val = lane == 0 ? Q.m0 : (lane == 1 ? Q.m1 : Q.me);
        if (lane < 3) h0[lane * nw64 + t] = val;
    }
    wave_sync();
    // d. gap-open bytes (a7): the base's, four at a time; then the chunks marked above recomputed from the run plane
    if (write_gob) {
        for (int o = 4 * lane; o < hapLen; o += 256) {
            const uint32_t v = ld4(gob_base + o);
            if (o + 4 <= hapLen) __builtin_memcpy(gob + o, &v, 4);
            else for (int q = 0; o + q < hapLen; ++q) gob[o + q] = (uint8_t)(v >> (8 * q));
        }
        for (u64 m = godirty; m; m &= m - 1ull) {
            const int t = (int)__ffsll((long long)m) - 1, p = 64 * t + lane;
            const u64 v = funnel(eqp[t], eqp[t + 1], lane);
            const int run = min(48, (int)__ffsll((long long)~v) - 1);
            if (p < hapLen) gob[p] = (uint8_t)s_go[run < 0 ? 48 : run];
        }
    }
    for (int t = lane; t < nch; t += 64) {               // per-chunk minima: the base's, recomputed where a run may have changed
        unsigned char gm = t < 64 ? gm0 : bgmin[t];
        if ((godirty >> t) & 1ull) {
            const u64 e0 = eqp[t], e1 = eqp[t + 1];
            const int nvalid = min(64, hapLen - 64 * t);
            u64 left = nvalid >= 64 ? ~0ull : ((1ull << nvalid) - 1ull);
            int longest = 0;
            while (longest < 48) {
                left &= funnel(e0, e1, longest);
                if (left == 0ull) break;
                ++longest;
            }
            gm = (unsigned char)s_go[longest];
        }
        s_gmin[t] = gm;
    }
    // e. the 7-mers that contain a differing base are new: none of them may be a code the base has seen, nor equal another new one
    int prev = -1;
    for (int j = 0; j < ndiff; ++j) {
        const int p = __shfl(mypos, j);
        const int sidx = p - 6 + lane;                   // lanes 0..6: the 7-mers starting at p-6 .. p (those not taken by the difference before)
        const bool valid = lane < 7 && sidx >= 0 && sidx > prev && sidx < hapLen - 7;
        unsigned code = 0xFFFFFFFFu;
        if (valid) { const int t = sidx >> 6; code = plane_code(h0[t], h0[t + 1], h1[t], h1[t + 1], sidx & 63); }
        if (lane < 7) table[7 * j + lane] = code;        // (the index area is free: no index of this haplotype exists yet)
        if (valid) atomicAnd(&((unsigned*)nup)[sidx >> 5], ~(1u << (sidx & 31)));
        prev = p;
    }
    wave_sync();
    const int n = 7 * ndiff;
    const unsigned mycode = lane < n ? table[lane] : 0xFFFFFFFFu;
    const bool hit = mycode != 0xFFFFFFFFu && ((bseen[mycode >> 5] >> (mycode & 31u)) & 1u);      // the base's "seen at least once" map: one look-up per lane
    if (__any(hit)) return false;
    for (int j = 0; j < n; ++j) {
        const unsigned cj = table[j];
        if (cj != 0xFFFFFFFFu && __any(lane != j && mycode == cj)) return false;
    }
    // f. flags: N / other bytes may only have been added (a flag too many sends pairs to the DP's general path, nothing else)
    if (lane == 0) { s_scal[0] = bs0 | (newN ? 1 : 0); s_scal[1] = bs1; s_scal[2] = bs2 | (newOther ? 1 : 0); }
    wave_sync();
    return true;
}

// One single-wave workgroup per window: the full sweep of the window's first haplotype (its gap-open bytes and has_n flag are final),
// left in the window's record for k_seed.  ok = 0 when nothing can be derived from it (counting table overflowed: the index overwrote
// the maps; haplotype shorter than 16 bases).
__global__ void __launch_bounds__(64)
k_seed_base(plat_window_batch b, uint8_t* __restrict__ gob, uint8_t* __restrict__ hap_has_n, long long* cnt, unsigned char* __restrict__ basebuf,
            int tsize_max, int maxhap, int shortcuts)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nw64 = ((maxhap + 63) >> 6) + 8;
    unsigned* table = (unsigned*)smem;
    unsigned short* nxt = (unsigned short*)(smem + (size_t)tsize_max * 4);
    u64* h0 = (u64*)(smem + (size_t)tsize_max * 4 + (((size_t)maxhap + 2) * 2 + 7 & ~(size_t)7));
    u64* h1 = h0 + nw64;
    u64* eqp = h1 + nw64;
    u64* nup = eqp + nw64;
    int* s_scal = (int*)(nup + nw64);
    unsigned char* s_gmin = (unsigned char*)((unsigned*)(s_scal + 4 + 16) + 32);
    if (cnt[CNT_ERR] != 0) return;
    const int w = blockIdx.x, lane = threadIdx.x;
    unsigned char* rec = basebuf + (size_t)w * seed_base_stride(maxhap);
    const int h = b.win_hap_begin[w];
    if (b.win_hap_begin[w + 1] <= h) return;
    const long long hoff = b.hap_off[h];
    const int hapLen = (int)(b.hap_off[h + 1] - hoff);
    const bool direct = hapLen > 4096;
    int tsize = 64;
    if (direct) tsize = 16384;
    else while (tsize < hapLen + hapLen / 4) tsize <<= 1;
    const unsigned tmask = (unsigned)tsize - 1u;
    const int nch = (hapLen + 63) >> 6;
    bool stop = false;
    const bool have_index = seed_sweep(table, nxt, h0, h1, eqp, nup, s_scal, nw64, b.hap_seq + hoff, hapLen, hoff, true, gob, cnt, shortcuts, direct, tsize,
                                       tmask, nch, stop);
    const bool ok = !have_index && !stop && hapLen >= 16 && hapLen <= 4096;
    if (lane == 0) {
        hap_has_n[h] = (uint8_t)s_scal[0];
        int* o = (int*)(rec + SEED_BASE_SCAL);
        o[0] = s_scal[0]; o[1] = s_scal[1]; o[2] = s_scal[2]; o[3] = ok ? 1 : 0;
    }
    if (!ok) return;
    unsigned* oseen = (unsigned*)(rec + SEED_BASE_SEEN);
    for (int i = lane; i < 512; i += 64) oseen[i] = table[i];
    u64* oplanes = (u64*)(rec + SEED_BASE_PLANES);
    for (int i = lane; i < 4 * nw64; i += 64) oplanes[i] = h0[i];
    unsigned char* ogmin = rec + SEED_BASE_PLANES + 32 * (size_t)nw64;
    for (int t = lane; t < nch; t += 64) ogmin[t] = s_gmin[t];
}

__global__ void __launch_bounds__(64)
k_seed(plat_window_batch b, const int32_t* __restrict__ hap_win, const int32_t* __restrict__ win_rows,
       const long long* __restrict__ tile_off, const ReadInfo* __restrict__ rinfo,
       const uint16_t* __restrict__ codes, uint8_t* __restrict__ gob, uint8_t* __restrict__ hap_has_n,
       PairRec* __restrict__ pairs, Job* __restrict__ jobs, long long npairs, int extra_cap, long long* cnt,
       SlowRec* __restrict__ slow_list, int tsize_max, int maxhap, int shortcuts,
       int32_t* __restrict__ dense, long long segcap, const double* __restrict__ mapq_lut, double* __restrict__ out_ll,
       int32_t* __restrict__ out_score, const unsigned char* __restrict__ basebuf)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nw64 = ((maxhap + 63) >> 6) + 8;           // plane words incl. slack for the shifted window of a hypothesis
    unsigned* table = (unsigned*)smem;
    unsigned short* nxt = (unsigned short*)(smem + (size_t)tsize_max * 4);      // tsize_max = carve size >= 1024 dwords: the multiplicity maps overlay the table
    u64* h0 = (u64*)(smem + (size_t)tsize_max * 4 + (((size_t)maxhap + 2) * 2 + 7 & ~(size_t)7));
    u64* h1 = h0 + nw64;
    u64* eqp = h1 + nw64;
    u64* nup = eqp + nw64;
    int* s_scal = (int*)(nup + nw64);                    // [0] has_n  [1] maxmult, then the gap-open table

    // Workgroups go to the 8 XCDs round robin by their linear id, and each XCD has its own L2: with SEED_XCD (grid.x a multiple of 8) XCD x
    // takes the x-th eighth of the haplotypes in order, so that the haplotypes of a window -- which all read the window's read planes and
    // ReadInfo -- run on ONE XCD at about the same time and those bytes leave HBM once, not once per XCD.
    int h = blockIdx.x;
    if (shortcuts & SEED_XCD) {
        const int per = (int)(gridDim.x >> 3);
        h = (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
        if (h >= b.n_haps) return;
    }
    if (cnt[CNT_ERR] != 0) return;                       // an earlier stage refused the batch
    const int w = hap_win[h];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nw = nthr >> 6;
    {   // this workgroup's group of read chunks lies beyond the window's reads: nothing to do
        const int Rw = b.win_read_begin[w + 1] - b.win_read_begin[w];
        if (blockIdx.y > 0 && (int)blockIdx.y * SEED_CHUNKS * nw * 64 >= Rw) return;
    }
    const bool first_group = blockIdx.y == 0;            // writes the per-haplotype outputs (hapw, has_n)
    const long long hoff = b.hap_off[h];
    const int hapLen = (int)(b.hap_off[h + 1] - hoff);
    const uint8_t* hs = b.hap_seq + hoff;

    const bool direct = hapLen > 4096;
    int tsize = 64;
    if (direct) tsize = 16384;
    else while (tsize < hapLen + hapLen / 4) tsize <<= 1;
    const unsigned tmask = (unsigned)tsize - 1u;
    const int nch = (hapLen + 63) >> 6;                  // chunks of 64 haplotype positions

    // Setup, fast part: the proof of hypothesis A only needs the planes, nu and maxmult.  Multiplicities come from two
    // 16384-bit maps over the 14-bit k-mer codes ("seen at least once / twice"), filled with one LDS atomicOr per k-mer,
    // plus a 32-entry counting table for the few codes seen three times or more.  The maps overlay the k-mer index, which
    // is only built (seed_build_index) when some pair of this workgroup needs a look-up: hypothesis B, the no-vote test,
    // the exact vote, or more than 32 distinct codes of multiplicity >= 3.  (LDS per workgroup decides how many
    // haplotypes a CU works on at once: 6 KB instead of 9.4 KB with a third map.)
    unsigned char* s_gmin = (unsigned char*)((unsigned*)(s_scal + 4 + 16) + 32);     // [nw64] smallest gap-open penalty per chunk (seed_sweep)
    bool stop = false, have_index = false, derived = false;
    if (basebuf) {                                       // the window's first haplotype was swept by k_seed_base: derive this one from it if possible
        const unsigned char* rec = basebuf + (size_t)w * seed_base_stride(maxhap);
        const int hB = b.win_hap_begin[w];
        const long long hoffB = b.hap_off[hB];
        if (((const int*)(rec + SEED_BASE_SCAL))[3] != 0 && (int)(b.hap_off[hB + 1] - hoffB) == hapLen)
            derived = seed_derive(rec, b.hap_seq + hoffB, gob + hoffB, table, h0, h1, eqp, nup, s_scal, s_gmin, nw64, hs, hapLen, h == hB,
                                  first_group && h != hB, gob + hoff, cnt);
    }
    if (!derived) have_index = seed_sweep(table, nxt, h0, h1, eqp, nup, s_scal, nw64, hs, hapLen, hoff, first_group, gob, cnt, shortcuts, direct, tsize, tmask, nch, stop);
    if (stop) return;
    if (tid == 0 && first_group) hap_has_n[h] = (uint8_t)s_scal[0];
    const int maxmult = s_scal[1];
    const bool hap_plain = s_scal[2] == 0;               // only A, C, G, T, N: equal 2-bit codes of plain read bases mean equal bytes or a haplotype N

    const int rb = b.win_read_begin[w], R = b.win_read_begin[w + 1] - rb;
    const int hl = h - b.win_hap_begin[w];
    const int hapStart = b.win_start[w] - b.win_flank[w];                   // chaplotype.pyx:606
    const long long pbase = b.pair_off[w] + (long long)hl * R;
    const u64* rd2 = (const u64*)(codes + tile_off[w]);
    const int nkp = hapLen - 7;                          // haplotype k-mer positions 0..hapLen-8 (calign.pyx:109)

    // reads are processed in chunks of 64 (one lane per read); blockIdx.y selects a group of SEED_CHUNKS chunks so that
    // windows with thousands of reads (population mode) spread over many workgroups (each rebuilds the small index)
    const int cbeg = (int)blockIdx.y * SEED_CHUNKS * nw * 64;
    const int cend = min(R, cbeg + SEED_CHUNKS * nw * 64);
    for (int c0 = cbeg + wave * 64; c0 < cend; c0 += nw * 64) {
        const int rl = c0 + lane;
        const bool valid = rl < R;
        ReadInfo ri = ReadInfo{0, 0, 0, 0};
        if (valid) ri = rinfo[rb + rl];
        const int L = (int)(ri.lfm & 0xFFFFu);
        const int rflags = (int)((ri.lfm >> 16) & 0xFFu);
        const uint8_t mapq = (uint8_t)(ri.lfm >> 24);
        const long long pidx = pbase + rl;
        const bool skipped = (rflags & 1) != 0, tooshort = L < 7;
        const int hq = h | ((rflags & 4) ? JOB_BIGQ : 0);                   // haplotype index + the read's add-flavour flag
        const bool hapshort = valid && !skipped && !tooshort && hapLen < L + 15;
        if (hapshort) set_err(cnt, PLAT_ERR_HAP_TOO_SHORT);
        const bool live = valid && !skipped && !tooshort && !hapshort;
        const int nk = live ? L - 7 : 0;
        const int idx0 = min(ri.pos - hapStart, hapLen - L - 15);           // calign.pyx:252
        const u64* col = rd2 + (valid ? rl : 0);

        // ---- bit-parallel proof
        const bool canfast = live && L <= 256;
        int nCl = canfast ? (L + 63) >> 6 : 0, nCmax = nCl;
#pragma unroll
        for (int s2 = 32; s2 > 0; s2 >>= 1) nCmax = max(nCmax, __shfl_xor(nCmax, s2));
        u64 r0[4], r1[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            r0[c] = c < nCl ? col[(long long)(2 * c) * R] : 0ull;
            r1[c] = c < nCl ? col[(long long)(2 * c + 1) * R] : 0ull;
        }
        int dstar = idx0;
        bool proven = false, triedB = false, exact = false;
        u64 missA[4] = {0ull, 0ull, 0ull, 0ull}, uniqA[4] = {0ull, 0ull, 0ull, 0ull};   // of hypothesis A, when it is proven (see "ungapped" below)
        bool provenA = false;
        for (int attempt = 0; attempt < 2; ++attempt) {
            const bool run = canfast && !proven && dstar >= 0 && (attempt == 0 || triedB);
            if (__any(run)) {
                const int wq = run ? (dstar >> 6) : 0, sb = dstar & 63;
                const int nvalid = min(nk, nkp - dstar);                 // k-mers i < nvalid lie on haplotype positions
                u64 Z[5], NU[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (c < nCmax) {
                        const u64 x = (funnel(h0[wq + c], h0[wq + c + 1], sb) ^ r0[c]) | (funnel(h1[wq + c], h1[wq + c + 1], sb) ^ r1[c]);
                        Z[c] = ~x;
                        NU[c] = funnel(nup[wq + c], nup[wq + c + 1], sb);
                    } else { Z[c] = 0ull; NU[c] = 0ull; }
                }
                Z[4] = 0ull;
                u64 P2[5];
#pragma unroll
                for (int c = 0; c < 4; ++c) P2[c] = Z[c] & ((Z[c] >> 1) | (Z[c + 1] << 63));
                P2[4] = 0ull;
                int C = 0, NUc = 0;
                u64 U7[4];                               // k-mer i of the read equals the haplotype's at d*+i AND that k-mer is unique in the haplotype
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    U7[c] = 0ull;
                    if (c < nCmax) {
                        const u64 P4 = P2[c] & ((P2[c] >> 2) | (P2[c + 1] << 62));
                        u64 P7 = P4 & ((P2[c] >> 4) | (P2[c + 1] << 60)) & ((Z[c] >> 6) | (Z[c + 1] << 58));
                        const int nb = nvalid - 64 * c;
                        const u64 msk = nb >= 64 ? ~0ull : (nb <= 0 ? 0ull : ((1ull << nb) - 1ull));
                        P7 &= msk;
                        C += __popcll(P7);
                        NUc += __popcll(P7 & NU[c]);
                        U7[c] = P7 & ~NU[c];
                    }
                }
                const int X = NUc * (maxmult - 1) + (nk - C) * maxmult;
                if (run && X < C) {
                    proven = true;
                    // does the whole read match the haplotype on d*?  (Z: one bit per base, 1 = equal codes)
                    u64 miss = 0ull;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int nb = L - 64 * c;
                        const u64 mc = ~Z[c] & (nb >= 64 ? ~0ull : (nb <= 0 ? 0ull : ((1ull << nb) - 1ull)));
                        miss |= mc;
                        if (attempt == 0) { missA[c] = mc; uniqA[c] = U7[c]; }
                    }
                    exact = miss == 0ull;
                    provenA = attempt == 0;
                }
            }
            if (attempt == 0) {
                // everything below needs k-mer look-ups: build the index now if this workgroup (= one wave) has not yet
                if (!have_index && __any(live && !proven)) {
                    seed_build_index(table, nxt, h0, h1, nup, s_scal, hapLen, nch, direct, tsize, tmask, false);
                    have_index = true;
                }
                // hypothesis B for the lanes A could not prove: diagonal of the first haplotype-unique k-mer
                if (canfast && !proven) {
                    for (int i = 0; i < nk; ++i) {
                        const unsigned hd = kmer_head(table, read_code(col, R, i), direct, tmask);
                        if (hd != 0u && nxt[hd] == 0u) {
                            const int d = (int)hd - i - 1;
                            if (d != idx0 && d >= 0) { dstar = d; triedB = true; }
                            break;
                        }
                    }
                }
                if (!__any(triedB)) break;
            }
        }
        // no k-mer of the read occurs in the haplotype <=> maxcount == 0 (calign.pyx:222): decided, no candidate.
        // (tested only for pairs the proof left open)
        bool novote = false;
        if (live && !proven) {
            novote = true;
            for (int i = 0; i < nk && novote; ++i)
                if (kmer_head(table, read_code(col, R, i), direct, tmask) != 0u) novote = false;
        }
        const bool decided = !live || novote || proven;
        int ncand = 0, cidx = idx0;
        bool orig_in = false;
        if (live && proven && dstar + L + 15 < hapLen) { ncand = 1; cidx = dstar; orig_in = (idx0 == dstar); }   // calign.pyx:228
        // The read equals the haplotype on the one candidate diagonal: that DP scores 0 (no cost is negative and the
        // all-match path costs 0; a haplotype N costs 0 as well, align.c:17,314-318) and calign.pyx:242-247 returns it
        // at once.  No DP is launched for the pair.
        const bool zero = (shortcuts & SHORTCUT_EXACT) && ncand == 1 && exact && hap_plain && !((rflags >> 1) & 1);
        // ---- "ungapped": the read differs from the haplotype in 1..UNG_KMAX bases on the one candidate diagonal d*, which is also
        // the mapping position, and NO other path of the band can be cheaper than paying those mismatches.  Then the single DP
        // of the pair returns U = sum of the mismatching bases' qualities and is not launched.  Cost model (align.c:314-335,
        // 466-484): a mismatch costs qual[y]; a deletion of l bases go[x] + 3 (l - 1); an insertion of l bases go[x] + 2 + 5 (l - 1);
        // nothing is negative; the band is d*-8 .. d*+7 (needs d* >= 8), the path may start and end on any diagonal.
        // The one fact every bound uses: where 7-mer i of the read equals the haplotype's 7-mer at d*+i and that 7-mer occurs
        // ONCE in the haplotype, the read has a mismatch in [i, i+7) on every other diagonal; n such starts inside a stretch the
        // path spends on ONE other diagonal give ceil(n/7) disjoint windows, together worth V(.) (each holds a mismatching base,
        // a base costs >= the read's smallest quality m, all but n_low of its bases cost >= 20).  A gap opening costs >= G, the
        // smallest gap-open penalty of the slice, and an insertion skips read bases: <= 8 when it leaves or rejoins d*, <= 15
        // otherwise, so a gap in the middle of a stretch spoils <= 21 starts = 3 windows.
        // A path is a chain of stretches ON d*, which pay exactly the mismatches p_j inside them, and EXCURSIONS, each of which
        // dodges a run of mismatches j..j'.  If every possible excursion costs at least the qualities it dodges, and a path that
        // never touches d* costs >= U, no path beats U.  Windows are counted per stretch between two mismatches (they cannot
        // overlap across a mismatching base): W(a,b) = sum over those stretches of ceil(#unique-matching starts inside [a,b) / 7).
        // A FURTHER gap inside an excursion spoils the windows it cuts or skips, at a price: a deletion or a one-base insertion
        // one window for >= G, an insertion of 2..8 two for >= G+7, of 9..15 three for >= G+42; so spoiling windows costs
        // >= v'' = min((G+7)/2, (G+42)/3) apiece, and windows are worth phi(n) = V(n) with its slopes capped at v''
        // (= V itself once G >= 33):
        //   never on d*                                     phi(W(0, L-6)) >= U
        //   elsewhere, then on d* from after p_j            G + min(phi(W), 7 + phi(W-1)) >= q_1 + .. + q_j,  W = W(0, p_j-6): the gap
        //                                                   that joins d* is a deletion or a one-base insertion, or a longer
        //                                                   insertion that costs >= 7 more and skips <= 7 more starts
        //   on d* up to a gap before p_j, then elsewhere    the same with W = W(p_j+1, L-6) and q_j + .. + q_k
        //   leaves d* before p_j, rejoins after p_j'        two gaps = an insertion and a deletion of l bases each:
        //                                                   2G + 8l - 6 + windows, i.e. >= 2G + min(2 + phi(W), 10 + phi(W-1));
        //                                                   more gaps: >= 2G + max(G, phi(W-2));  W = W(p_j+1, p_j'-6);
        //                                                   all >= q_j + .. + q_j'
        // (haplotype without N, read of plain A/C/G/T: equal codes = equal bytes.)
        int ung_score = -1;
        int why = 0;                                     // (PLAT_SEED_DEBUG=512: why the pair reached the DP; counted below)
        {
            const int mq = (rflags >> 3) & 31;
            // The proof's cost model is exact arithmetic; align.c adds in wrapping int16 ("no overflow checks", align.c:81).  A read whose
            // quality sum allows a band cell to pass 0x7FFF (rflags bit 2, the flag that also picks the DP's add flavour, dp_core.hpp)
            // is left to the DP, which wraps as the reference does.  (The exact-match shortcut above needs no such guard: re-biased
            // values are unsigned, nothing is below 0, and the all-match path stays at 0 whatever the other cells do.)
            const bool wrapfree = !(rflags & 4) || (shortcuts & SHORTCUT_BIGQ);      // (SHORTCUT_BIGQ: measurement only, PLAT_UNGAPPED_BIGQ=1)
            const bool cand = (shortcuts & SHORTCUT_UNGAPPED) && ncand == 1 && orig_in && provenA && !exact && hap_plain && s_scal[0] == 0 &&
                              !((rflags >> 1) & 1) && cidx >= 8 && L >= 32 && wrapfree;
            int k = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) k += __popcll(missA[c]);
            const bool part = cand && k >= 1 && k <= UNG_KMAX;
            why = !(ncand == 1) ? 1 : !orig_in ? 2 : !provenA ? 3 : exact ? 4 : k > UNG_KMAX ? 5 : !cand ? 8 : 0;
            int kmw = part ? k : 0;                          // most mismatches any lane of the wave has to look at
#pragma unroll
            for (int s2 = 32; s2 > 0; s2 >>= 1) kmw = max(kmw, __shfl_xor(kmw, s2));
            if (kmw > 0) {
                int cw[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) cw[c] = __popcll(uniqA[c]);
                auto Cpre = [&](int x) -> int {              // unique-matching k-mer starts in [0, x)
                    x = min(max(x, 0), 256);
                    const int wi = x >> 6, sh = x & 63;
                    const u64 wsel = wi == 0 ? uniqA[0] : wi == 1 ? uniqA[1] : wi == 2 ? uniqA[2] : wi == 3 ? uniqA[3] : 0ull;
                    const int below = (wi > 0 ? cw[0] : 0) + (wi > 1 ? cw[1] : 0) + (wi > 2 ? cw[2] : 0) + (wi > 3 ? cw[3] : 0);
                    return below + __popcll(wsel & ((1ull << sh) - 1ull));
                };
                auto Wof = [&](int n) -> int { return (max(n, 0) + 6) / 7; };
                // n_low: how many of the read's bases may cost less than LOWQ (n disjoint windows are worth V(n) = mq min(n, n_low) + LOWQ max(n - n_low, 0))
                const int nlow = (shortcuts & SHORTCUT_NLOW) ? (int)(ri.aux & 0xFFFFu) : 0x7FFF;
                // the mismatches in read order (lanes with fewer than kmw repeat their last one with quality 0: its tests repeat too)
                const uint8_t* rq = b.read_qual + b.read_off[rb + (valid ? rl : 0)];
                u64 mm[4] = {missA[0], missA[1], missA[2], missA[3]};
                int pp[UNG_KMAX], qq[UNG_KMAX];
#pragma unroll
                for (int j = 0; j < UNG_KMAX; ++j) {
                    pp[j] = j ? pp[j - 1] : 0; qq[j] = 0;
                    if (j < kmw) {                           // wave-uniform
                        int pos = -1;
#pragma unroll
                        for (int c = 3; c >= 0; --c) if (mm[c]) pos = 64 * c + (int)__ffsll((long long)mm[c]) - 1;
                        if (part && pos >= 0) { pp[j] = pos; qq[j] = rq[pos]; }
#pragma unroll
                        for (int c = 0; c < 4; ++c) if (pos >= 0 && (pos >> 6) == c) mm[c] &= mm[c] - 1ull;
                    }
                }
                int U = 0;
#pragma unroll
                for (int j = 0; j < UNG_KMAX; ++j) U += qq[j];
                const int st = cidx - 8;
                int G = 127;
                if (part) for (int t = st >> 6; t <= (st + L + 14) >> 6; ++t) G = min(G, (int)s_gmin[t]);
                // everything below in units of 1/6 (v'' has a half and a third in it)
                const int v6 = min(3 * (G + 7), 2 * (G + 42));
                const int c_lo = min(6 * mq, v6), c_hi = min(6 * max(mq, (int)LOWQ), v6);
                auto phi6 = [&](int n) -> int { n = max(n, 0); return c_lo * min(n, nlow) + c_hi * max(n - nlow, 0); };
                // unique-matching starts before p_j - 6 and before p_j + 1 (none start in between: those 7-mers hold the mismatch)
                int cm6[UNG_KMAX], cp1[UNG_KMAX];
#pragma unroll
                for (int j = 0; j < UNG_KMAX; ++j) {
                    if (j < kmw) { cm6[j] = Cpre(pp[j] - 6); cp1[j] = Cpre(pp[j] + 1); }
                    else { cm6[j] = cm6[j - (j > 0)]; cp1[j] = cp1[j - (j > 0)]; }
                }
                const int Ctot = Cpre(L - 6);
                int Iw[UNG_KMAX];                            // windows between mismatch j and the next
#pragma unroll
                for (int j = 0; j + 1 < UNG_KMAX; ++j) Iw[j] = Wof(cm6[j + 1] - cp1[j]);
                Iw[UNG_KMAX - 1] = 0;
                int Wall = Wof(cm6[0]) + Wof(Ctot - cp1[UNG_KMAX - 1]);
#pragma unroll
                for (int j = 0; j + 1 < UNG_KMAX; ++j) Wall += Iw[j];
                int bad = phi6(Wall) < 6 * U ? 10 : 0;       // never on d*
                // an excursion at the head or the tail of the read: its gap next to d* is a deletion or a one-base insertion
                // (no window lost) or a longer insertion (>= 7 more, one window lost)
                auto edge = [&](int W, int T) -> bool { return 6 * (G - T) + min(phi6(W), 42 + phi6(W - 1)) >= 0; };
                int Rs = 0, Qs = U, Wbefore = Wof(cm6[0]);   // windows before mismatch j
#pragma unroll
                for (int j = 0; j < UNG_KMAX; ++j) {
                    if (j < kmw) {
                        Rs += qq[j];
                        if (!bad && !edge(Wbefore, Rs)) bad = 11;                     // elsewhere, then on d* from after p_j
                        if (!bad && !edge(Wall - Wbefore, Qs)) bad = 12;              // on d* up to a gap before p_j, then elsewhere
                        Wbefore += Iw[j];
                        Qs -= qq[j];
                        int T = qq[j], Wm = 0;
                        if (!bad && 2 * G + 2 < T) bad = 13;                          // an excursion around p_j alone
#pragma unroll
                        for (int j2 = j + 1; j2 < UNG_KMAX; ++j2) {
                            if (j2 < kmw) {
                                T += qq[j2];
                                Wm += Iw[j2 - 1];
                                const int slack = 6 * (2 * G - T);
                                const bool two = slack + 12 + phi6(Wm) >= 0 && slack + 60 + phi6(Wm - 1) >= 0;
                                const bool more = slack + max(6 * G, phi6(Wm - 2)) >= 0;
                                if (!bad && !(two && more)) bad = 14;
                            }
                        }
                    }
                }
                if (part && !bad) ung_score = U;
                else if (part) why = bad;
            }
        }
        const bool ungapped = ung_score >= 0;
        // extra job slot for (one candidate that is not the mapping position): one atomic per wave
        int base = 0;
        {
            const bool need = decided && live && ncand == 1 && !orig_in && !zero;
            const unsigned long long m = __ballot(need);
            if (m) {
                int wb = 0;
                if (lane == 0) wb = (int)atomicAdd((unsigned long long*)&cnt[CNT_NEXTRA], (unsigned long long)__popcll(m));
                wb = __shfl(wb, 0);
                base = wb + __popcll(m & ((1ull << lane) - 1ull));
                if (need && (long long)base + 1 <= (long long)extra_cap) jobs[npairs + base] = Job{ri.col, hq, idx0, L};
            }
        }
        // Decided pairs: the ones that need no DP are finished here (skipped read: 0.0, chaplotype.pyx:345-346; read < 7 bp or exact
        // match: score 0; ungapped alignment proven optimal: its score), the others leave a job in their slot.
        // (SEED_LEAN, the asynchronous entry point: nobody asks for statistics afterwards, and k_finalize_dense only looks at pairs with
        // a DP -- the 32 bytes of records of a finished pair, 7 pairs in 8 of a clean batch, are not written at all.)
        bool prim = false;                                   // the pair's primary slot holds a DP
        const bool recs = !(shortcuts & SEED_LEAN);
        if (valid && decided) {
            if (!live) {
                const bool sk = skipped || hapshort;
                if (recs) {
                    pairs[pidx] = PairRec{0, 0, (int16_t)(sk ? -1 : -2), 0, mapq, {0, 0, 0}};
                    jobs[pidx] = Job{ri.col, h, 0, 0};
                }
                out_ll[pidx] = sk ? 0.0 : loglik_of(0, mapq_lut, mapq);
                if (out_score) out_score[pidx] = sk ? -1 : 0;
            } else if (zero) {
                if (recs) {
                    pairs[pidx] = PairRec{0, L, (int16_t)-3, 0, mapq, {0, 0, 0}};
                    jobs[pidx] = Job{ri.col, h, cidx, 0};
                }
                out_ll[pidx] = loglik_of(0, mapq_lut, mapq);
                if (out_score) out_score[pidx] = 0;
            } else if (ungapped) {
                if (recs) {
                    pairs[pidx] = PairRec{ung_score, L, (int16_t)-4, 0, mapq, {0, 0, 0}};
                    jobs[pidx] = Job{ri.col, h, cidx, 0};
                }
                out_ll[pidx] = loglik_of(ung_score, mapq_lut, mapq);
                if (out_score) out_score[pidx] = ung_score;
            } else {
                jobs[pidx] = Job{ri.col, hq, cidx, L};
                pairs[pidx] = PairRec{base, idx0, (int16_t)ncand, (int16_t)(orig_in ? 0 : ncand), mapq, {0, 0, 0}};
                prim = true;
            }
        }
        if (shortcuts & 512) {                              // measurement only
            for (int r = 0; r < 16; ++r) {
                const unsigned long long m = __ballot(prim && why == r);
                if (m && lane == 0) atomicAdd((unsigned long long*)&cnt[32 + r], (unsigned long long)__popcll(m));
            }
        }
        {   // the wave's live job slots join the dense list: primary slots, then the extra ones, room reserved with one atomic
            const bool extra = decided && live && ncand == 1 && !orig_in && !zero && valid && (long long)base + 1 <= (long long)extra_cap;
            const unsigned long long m1 = __ballot(prim), m2 = __ballot(extra);
            const int n1 = __popcll(m1), n2 = __popcll(m2);
            if (n1 + n2) {
                const int seg = w % DENSE_SEGS;
                long long db = 0;
                if (lane == 0) db = (long long)atomicAdd((unsigned long long*)dense_counter(cnt, seg), (unsigned long long)(n1 + n2));
                db = ((long long)(unsigned)__shfl((int)db, 0)) | ((long long)__shfl((int)(db >> 32), 0) << 32);
                if (db + n1 + n2 <= segcap) {
                    const unsigned long long below = (1ull << lane) - 1ull;
                    if (prim) dense[seg * segcap + db + __popcll(m1 & below)] = (int32_t)pidx;
                    if (extra) dense[seg * segcap + db + n1 + __popcll(m2 & below)] = (int32_t)(npairs + base);
                }
            }
        }
        // ---- pairs that could not be decided go to the exact vote in k_seed_slow (one wave per pair, spread over the
        // whole device: a tandem-repeat window would otherwise serialise all its reads on this one wave)
        const unsigned long long todo = __ballot(valid && !decided);
        if (todo) {
            long long sb = 0;
            if (lane == 0) sb = (long long)atomicAdd((unsigned long long*)&cnt[CNT_SLOW_SEED], (unsigned long long)__popcll(todo));
            sb = ((long long)(unsigned)__shfl((int)sb, 0)) | ((long long)__shfl((int)(sb >> 32), 0) << 32);
            if (valid && !decided) slow_list[sb + __popcll(todo & ((1ull << lane) - 1ull))] = SlowRec{h, rl};
        }
    }
}


// ---- round 4: the seeding stage as TWO kernels -----------------------------------------------------------------------------------------
// k_seed did two things per haplotype in one single-wave workgroup: the haplotype sweep (planes, gap-open bytes, multiplicity maps) and
// the per-pair proofs with ONE LANE PER READ of the window -- 34..40 lanes of 64 on a 30x window of 150 bp reads, and that part is two
// thirds of its vector instructions.  k_sweep keeps the first half and leaves what the proofs read in global memory (per haplotype:
// flags, largest multiplicity, the planes h0 / h1 / nu, per-chunk gap-open minima: ~0.5 KB); k_pairs packs the window's (haplotype, read)
// pairs DENSELY, 64 per wave whatever the number of reads, stages the <= SEED_NST haplotype records its pairs touch in LDS and runs the
// same proofs.  The k-mer index of a haplotype is built in k_pairs only when a pair needs look-ups (hypothesis B, no-vote test), as before.
constexpr int SEED_NST = 6;            // haplotype records staged per wave of k_pairs: 64 consecutive pairs of a window with R >= 13 reads
                                       // span at most 6 haplotypes; windows with fewer reads give a wave 5 whole haplotypes (5 R <= 60 pairs)
__host__ __device__ __forceinline__ int seed_pairs_per_wave(int R) { return R >= 13 ? 64 : 5 * (R > 0 ? R : 1); }
__host__ __device__ __forceinline__ size_t seed_state_stride(int maxhap) {
    const size_t nw64 = (((size_t)maxhap + 63) >> 6) + 8;
    return 16 + 24 * nw64 + ((nw64 + 15) & ~(size_t)15);                   // int scal[4] | u64 h0[nw64], h1[nw64], nu[nw64] | u8 gmin[nw64]
}

__global__ void __launch_bounds__(64)
k_sweep(plat_window_batch b, uint8_t* __restrict__ gob, uint8_t* __restrict__ hap_has_n, long long* cnt, int tsize_max, int maxhap, int shortcuts,
        const unsigned char* __restrict__ basebuf, const int32_t* __restrict__ hap_win, unsigned char* __restrict__ state)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nw64 = ((maxhap + 63) >> 6) + 8;
    unsigned* table = (unsigned*)smem;
    unsigned short* nxt = (unsigned short*)(smem + (size_t)tsize_max * 4);
    u64* h0 = (u64*)(smem + (size_t)tsize_max * 4 + (((size_t)maxhap + 2) * 2 + 7 & ~(size_t)7));
    u64* h1 = h0 + nw64;
    u64* eqp = h1 + nw64;
    u64* nup = eqp + nw64;
    int* s_scal = (int*)(nup + nw64);
    unsigned char* s_gmin = (unsigned char*)((unsigned*)(s_scal + 4 + 16) + 32);
    int h = blockIdx.x;
    if (shortcuts & SEED_XCD) {                          // haplotype h on XCD floor(8 h / n): see k_seed
        const int per = (int)(gridDim.x >> 3);
        h = (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
        if (h >= b.n_haps) return;
    }
    if (cnt[CNT_ERR] != 0) return;
    const int lane = threadIdx.x;
    const long long hoff = b.hap_off[h];
    const int hapLen = (int)(b.hap_off[h + 1] - hoff);
    const uint8_t* hs = b.hap_seq + hoff;
    const bool direct = hapLen > 4096;
    int tsize = 64;
    if (direct) tsize = 16384;
    else while (tsize < hapLen + hapLen / 4) tsize <<= 1;
    const unsigned tmask = (unsigned)tsize - 1u;
    const int nch = (hapLen + 63) >> 6;
    bool stop = false, derived = false;
    if (basebuf) {                                       // PLAT_SEED_SHARE=1: derive from the window's first haplotype where possible (see seed_derive)
        const int w = hap_win[h];
        const unsigned char* rec = basebuf + (size_t)w * seed_base_stride(maxhap);
        const int hB = b.win_hap_begin[w];
        const long long hoffB = b.hap_off[hB];
        if (((const int*)(rec + SEED_BASE_SCAL))[3] != 0 && (int)(b.hap_off[hB + 1] - hoffB) == hapLen)
            derived = seed_derive(rec, b.hap_seq + hoffB, gob + hoffB, table, h0, h1, eqp, nup, s_scal, s_gmin, nw64, hs, hapLen, h == hB,
                                  h != hB, gob + hoff, cnt);
    }
    if (!derived) seed_sweep(table, nxt, h0, h1, eqp, nup, s_scal, nw64, hs, hapLen, hoff, true, gob, cnt, shortcuts, direct, tsize, tmask, nch, stop);
    if (stop) return;
    unsigned char* rec = state + (size_t)h * seed_state_stride(maxhap);
    if (lane == 0) hap_has_n[h] = (uint8_t)s_scal[0];
    if (lane < 4) ((int*)rec)[lane] = lane < 3 ? s_scal[lane] : 0;
    u64* op = (u64*)(rec + 16);
    for (int i = lane; i < nw64; i += 64) { op[i] = h0[i]; op[nw64 + i] = h1[i]; op[2 * nw64 + i] = nup[i]; }
    unsigned char* og = rec + 16 + 24 * (size_t)nw64;
    for (int t = lane; t < ((nw64 + 15) & ~15); t += 64) og[t] = t < nch ? s_gmin[t] : (unsigned char)0;
}

// One wave = up to 64 consecutive (haplotype, read) pairs of ONE window, in the order of the window's likelihood block
// (pair q = hl * R + rl).  wave_win / wave_first (k_tile_scan): wave -> window, window -> its first wave.
__global__ void __launch_bounds__(64)
k_pairs(plat_window_batch b, const int32_t* __restrict__ wave_win, const int32_t* __restrict__ wave_first,
        const long long* __restrict__ tile_off, const ReadInfo* __restrict__ rinfo, const uint16_t* __restrict__ codes,
        PairRec* __restrict__ pairs, Job* __restrict__ jobs, long long npairs, int extra_cap, long long* cnt,
        SlowRec* __restrict__ slow_list, int tsize_max, int maxhap, int shortcuts,
        int32_t* __restrict__ dense, long long segcap, const double* __restrict__ mapq_lut, double* __restrict__ out_ll,
        int32_t* __restrict__ out_score, const unsigned char* __restrict__ state)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nw64 = ((maxhap + 63) >> 6) + 8;
    unsigned* table = (unsigned*)smem;                   // the k-mer index of ONE staged haplotype at a time, built on demand
    unsigned short* nxt = (unsigned short*)(smem + (size_t)tsize_max * 4);
    unsigned char* lstate = smem + (size_t)tsize_max * 4 + (((size_t)maxhap + 2) * 2 + 15 & ~(size_t)15);    // SEED_NST haplotype records
    const size_t stride = seed_state_stride(maxhap);
    const long long nwaves = cnt[CNT_NWAVES];            // (the grid is the host's upper bound)
    long long v = blockIdx.x;
    if (shortcuts & SEED_XCD) {                          // the waves of a window on ONE XCD (they share its read planes and haplotype records):
        const long long per = (nwaves + 7) >> 3, idx = (long long)(blockIdx.x >> 3);      // XCD x takes the x-th eighth of the waves there ARE
        if (idx >= per) return;
        v = (long long)(blockIdx.x & 7u) * per + idx;
    }
    if (cnt[CNT_ERR] != 0 || v >= nwaves) return;
    const int lane = threadIdx.x;
    const int w = wave_win[v];
    const int rb = b.win_read_begin[w], R = b.win_read_begin[w + 1] - rb;
    const int hb = b.win_hap_begin[w], H = b.win_hap_begin[w + 1] - hb;
    const int PW = seed_pairs_per_wave(R);
    // (a window's pairs fit 31 bits: k_validate refuses H R >= 2^31)
    const unsigned q0 = (unsigned)(v - (long long)wave_first[w]) * (unsigned)PW, QT = (unsigned)H * (unsigned)R;
    if (q0 >= QT) return;
    const int nq = (int)min((unsigned)PW, QT - q0);
    const int hl_lo = (int)(q0 / (unsigned)R), hl_hi = (int)((q0 + (unsigned)nq - 1u) / (unsigned)R), nst = hl_hi - hl_lo + 1;      // <= SEED_NST
    const int r_lo = (int)(q0 - (unsigned)hl_lo * (unsigned)R);             // the first pair's read
    const int hapStart = b.win_start[w] - b.win_flank[w];                   // chaplotype.pyx:606
    const u64* rd2 = (const u64*)(codes + tile_off[w]);
    const bool valid = lane < nq;
    // lane -> (haplotype, read): r_lo + lane < R + 64, i.e. at most SEED_NST - 1 whole rows further
    int slot = 0, rl = r_lo + (valid ? lane : 0);
#pragma unroll
    for (int k = 0; k < SEED_NST - 1; ++k) if (rl >= R) { rl -= R; ++slot; }
    ReadInfo ri = ReadInfo{0, 0, 0, 0};
    if (valid) ri = rinfo[rb + rl];                      // (requested before the records are staged: the two do not depend on each other)
    {   // the records of haplotypes hb + hl_lo .. hb + hl_hi are contiguous in `state`
        const u64* src = (const u64*)(state + (size_t)(hb + hl_lo) * stride);
        u64* dst = (u64*)lstate;
        const int nwords = (int)((size_t)nst * stride / 8);
        for (int i = lane; i < nwords; i += 64) dst[i] = src[i];
    }
    wave_sync();
    {
        const int hl = hl_lo + slot;
        const int h = hb + hl;
        const unsigned char* rec = lstate + (size_t)slot * stride;
        const u64* h0 = (const u64*)(rec + 16);
        const u64* h1 = h0 + nw64;
        const u64* nup = h1 + nw64;
        const unsigned char* s_gmin = (const unsigned char*)(nup + nw64);
        const int has_n = ((const int*)rec)[0], maxmult = ((const int*)rec)[1];
        const bool hap_plain = ((const int*)rec)[2] == 0;
        const int hapLen = (int)(b.hap_off[h + 1] - b.hap_off[h]);
        const int nkp = hapLen - 7;                      // haplotype k-mer positions 0..hapLen-8 (calign.pyx:109)
        const int L = (int)(ri.lfm & 0xFFFFu);
        const int rflags = (int)((ri.lfm >> 16) & 0xFFu);
        const uint8_t mapq = (uint8_t)(ri.lfm >> 24);
        const long long pidx = b.pair_off[w] + (long long)hl * R + rl;
        const bool skipped = (rflags & 1) != 0, tooshort = L < 7;
        const int hq = h | ((rflags & 4) ? JOB_BIGQ : 0);                   // haplotype index + the read's add-flavour flag
        const bool hapshort = valid && !skipped && !tooshort && hapLen < L + 15;
        if (hapshort) set_err(cnt, PLAT_ERR_HAP_TOO_SHORT);
        const bool live = valid && !skipped && !tooshort && !hapshort;
        const int nk = live ? L - 7 : 0;
        const int idx0 = min(ri.pos - hapStart, hapLen - L - 15);           // calign.pyx:252
        const u64* col = rd2 + (valid ? rl : 0);

        // ---- bit-parallel proof
        const bool canfast = live && L <= 256;
        int nCl = canfast ? (L + 63) >> 6 : 0, nCmax = nCl;
#pragma unroll
        for (int s2 = 32; s2 > 0; s2 >>= 1) nCmax = max(nCmax, __shfl_xor(nCmax, s2));
        u64 r0[4], r1[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            r0[c] = c < nCl ? col[(long long)(2 * c) * R] : 0ull;
            r1[c] = c < nCl ? col[(long long)(2 * c + 1) * R] : 0ull;
        }
        int dstar = idx0;
        bool proven = false, triedB = false, exact = false, direct = false;
        unsigned tmask = 0u;
        u64 missA[4] = {0ull, 0ull, 0ull, 0ull}, uniqA[4] = {0ull, 0ull, 0ull, 0ull};   // of hypothesis A, when it is proven (see "ungapped" below)
        bool provenA = false;
        // pass 0: hypothesis A for every lane (planes only).  Pass s + 1: the lanes of staged haplotype s that A left open need k-mer
        // look-ups -- that haplotype's index is built (the wave's one index area), hypothesis B is tried, then the no-vote test.
        bool novote = false;
        for (int pass = 0; pass <= nst; ++pass) {
            const int attempt = pass == 0 ? 0 : 1;
            bool mine = false;
            if (pass > 0) {
                mine = live && !proven && slot == pass - 1;
                if (!__any(mine)) continue;
                const unsigned char* recs = lstate + (size_t)(pass - 1) * stride;
                const int hs_ = hb + hl_lo + pass - 1;
                const int hapLenS = (int)(b.hap_off[hs_ + 1] - b.hap_off[hs_]);
                direct = hapLenS > 4096;
                int tsize = 64;
                if (direct) tsize = 16384;
                else while (tsize < hapLenS + hapLenS / 4) tsize <<= 1;
                tmask = (unsigned)tsize - 1u;
                u64* p0 = (u64*)(recs + 16);
                seed_build_index(table, nxt, p0, p0 + nw64, p0 + 2 * nw64, (int*)recs, hapLenS, (hapLenS + 63) >> 6, direct, tsize, tmask, false);
                // hypothesis B for the lanes A could not prove: diagonal of the first haplotype-unique k-mer
                triedB = false;
                if (mine && canfast) {
                    for (int i = 0; i < nk; ++i) {
                        const unsigned hd = kmer_head(table, read_code(col, R, i), direct, tmask);
                        if (hd != 0u && nxt[hd] == 0u) {
                            const int d = (int)hd - i - 1;
                            if (d != idx0 && d >= 0) { dstar = d; triedB = true; }
                            break;
                        }
                    }
                }
            }
            const bool run = canfast && !proven && dstar >= 0 && (pass == 0 || triedB);
            if (__any(run)) {
                const int wq = run ? (dstar >> 6) : 0, sb = dstar & 63;
                const int nvalid = min(nk, nkp - dstar);                 // k-mers i < nvalid lie on haplotype positions
                u64 Z[5], NU[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (c < nCmax) {
                        const u64 x = (funnel(h0[wq + c], h0[wq + c + 1], sb) ^ r0[c]) | (funnel(h1[wq + c], h1[wq + c + 1], sb) ^ r1[c]);
                        Z[c] = ~x;
                        NU[c] = funnel(nup[wq + c], nup[wq + c + 1], sb);
                    } else { Z[c] = 0ull; NU[c] = 0ull; }
                }
                Z[4] = 0ull;
                u64 P2[5];
#pragma unroll
                for (int c = 0; c < 4; ++c) P2[c] = Z[c] & ((Z[c] >> 1) | (Z[c + 1] << 63));
                P2[4] = 0ull;
                int C = 0, NUc = 0;
                u64 U7[4];                               // k-mer i of the read equals the haplotype's at d*+i AND that k-mer is unique in the haplotype
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    U7[c] = 0ull;
                    if (c < nCmax) {
                        const u64 P4 = P2[c] & ((P2[c] >> 2) | (P2[c + 1] << 62));
                        u64 P7 = P4 & ((P2[c] >> 4) | (P2[c + 1] << 60)) & ((Z[c] >> 6) | (Z[c + 1] << 58));
                        const int nb = nvalid - 64 * c;
                        const u64 msk = nb >= 64 ? ~0ull : (nb <= 0 ? 0ull : ((1ull << nb) - 1ull));
                        P7 &= msk;
                        C += __popcll(P7);
                        NUc += __popcll(P7 & NU[c]);
                        U7[c] = P7 & ~NU[c];
                    }
                }
                const int X = NUc * (maxmult - 1) + (nk - C) * maxmult;
                if (run && X < C) {
                    proven = true;
                    // does the whole read match the haplotype on d*?  (Z: one bit per base, 1 = equal codes)
                    u64 miss = 0ull;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int nb = L - 64 * c;
                        const u64 mc = ~Z[c] & (nb >= 64 ? ~0ull : (nb <= 0 ? 0ull : ((1ull << nb) - 1ull)));
                        miss |= mc;
                        if (attempt == 0) { missA[c] = mc; uniqA[c] = U7[c]; }
                    }
                    exact = miss == 0ull;
                    provenA = attempt == 0;
                }
            }
            // no k-mer of the read occurs in the haplotype <=> maxcount == 0 (calign.pyx:222): decided, no candidate.
            // (tested only for pairs the proof left open)
            if (pass > 0 && mine && !proven) {
                novote = true;
                for (int i = 0; i < nk && novote; ++i)
                    if (kmer_head(table, read_code(col, R, i), direct, tmask) != 0u) novote = false;
            }
            if (pass == 0 && !__any(live && !proven)) break;
        }
        const bool decided = !live || novote || proven;
        int ncand = 0, cidx = idx0;
        bool orig_in = false;
        if (live && proven && dstar + L + 15 < hapLen) { ncand = 1; cidx = dstar; orig_in = (idx0 == dstar); }   // calign.pyx:228
        // The read equals the haplotype on the one candidate diagonal: that DP scores 0 (no cost is negative and the
        // all-match path costs 0; a haplotype N costs 0 as well, align.c:17,314-318) and calign.pyx:242-247 returns it
        // at once.  No DP is launched for the pair.
        const bool zero = (shortcuts & SHORTCUT_EXACT) && ncand == 1 && exact && hap_plain && !((rflags >> 1) & 1);
        // ---- "ungapped": the read differs from the haplotype in 1..UNG_KMAX bases on the one candidate diagonal d*, which is also
        // the mapping position, and NO other path of the band can be cheaper than paying those mismatches.  Then the single DP
        // of the pair returns U = sum of the mismatching bases' qualities and is not launched.  Cost model (align.c:314-335,
        // 466-484): a mismatch costs qual[y]; a deletion of l bases go[x] + 3 (l - 1); an insertion of l bases go[x] + 2 + 5 (l - 1);
        // nothing is negative; the band is d*-8 .. d*+7 (needs d* >= 8), the path may start and end on any diagonal.
        // The one fact every bound uses: where 7-mer i of the read equals the haplotype's 7-mer at d*+i and that 7-mer occurs
        // ONCE in the haplotype, the read has a mismatch in [i, i+7) on every other diagonal; n such starts inside a stretch the
        // path spends on ONE other diagonal give ceil(n/7) disjoint windows, together worth V(.) (each holds a mismatching base,
        // a base costs >= the read's smallest quality m, all but n_low of its bases cost >= 20).  A gap opening costs >= G, the
        // smallest gap-open penalty of the slice, and an insertion skips read bases: <= 8 when it leaves or rejoins d*, <= 15
        // otherwise, so a gap in the middle of a stretch spoils <= 21 starts = 3 windows.
        // A path is a chain of stretches ON d*, which pay exactly the mismatches p_j inside them, and EXCURSIONS, each of which
        // dodges a run of mismatches j..j'.  If every possible excursion costs at least the qualities it dodges, and a path that
        // never touches d* costs >= U, no path beats U.  Windows are counted per stretch between two mismatches (they cannot
        // overlap across a mismatching base): W(a,b) = sum over those stretches of ceil(#unique-matching starts inside [a,b) / 7).
        // A FURTHER gap inside an excursion spoils the windows it cuts or skips, at a price: a deletion or a one-base insertion
        // one window for >= G, an insertion of 2..8 two for >= G+7, of 9..15 three for >= G+42; so spoiling windows costs
        // >= v'' = min((G+7)/2, (G+42)/3) apiece, and windows are worth phi(n) = V(n) with its slopes capped at v''
        // (= V itself once G >= 33):
        //   never on d*                                     phi(W(0, L-6)) >= U
        //   elsewhere, then on d* from after p_j            G + min(phi(W), 7 + phi(W-1)) >= q_1 + .. + q_j,  W = W(0, p_j-6): the gap
        //                                                   that joins d* is a deletion or a one-base insertion, or a longer
        //                                                   insertion that costs >= 7 more and skips <= 7 more starts
        //   on d* up to a gap before p_j, then elsewhere    the same with W = W(p_j+1, L-6) and q_j + .. + q_k
        //   leaves d* before p_j, rejoins after p_j'        two gaps = an insertion and a deletion of l bases each:
        //                                                   2G + 8l - 6 + windows, i.e. >= 2G + min(2 + phi(W), 10 + phi(W-1));
        //                                                   more gaps: >= 2G + max(G, phi(W-2));  W = W(p_j+1, p_j'-6);
        //                                                   all >= q_j + .. + q_j'
        // (haplotype without N, read of plain A/C/G/T: equal codes = equal bytes.)
        int ung_score = -1;
        int why = 0;                                     // (PLAT_SEED_DEBUG=512: why the pair reached the DP; counted below)
        {
            const int mq = (rflags >> 3) & 31;
            // The proof's cost model is exact arithmetic; align.c adds in wrapping int16 ("no overflow checks", align.c:81).  A read whose
            // quality sum allows a band cell to pass 0x7FFF (rflags bit 2, the flag that also picks the DP's add flavour, dp_core.hpp)
            // is left to the DP, which wraps as the reference does.  (The exact-match shortcut above needs no such guard: re-biased
            // values are unsigned, nothing is below 0, and the all-match path stays at 0 whatever the other cells do.)
            const bool wrapfree = !(rflags & 4) || (shortcuts & SHORTCUT_BIGQ);      // (SHORTCUT_BIGQ: measurement only, PLAT_UNGAPPED_BIGQ=1)
            const bool cand = (shortcuts & SHORTCUT_UNGAPPED) && ncand == 1 && orig_in && provenA && !exact && hap_plain && has_n == 0 &&
                              !((rflags >> 1) & 1) && cidx >= 8 && L >= 32 && wrapfree;
            int k = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) k += __popcll(missA[c]);
            const bool part = cand && k >= 1 && k <= UNG_KMAX;
            why = !(ncand == 1) ? 1 : !orig_in ? 2 : !provenA ? 3 : exact ? 4 : k > UNG_KMAX ? 5 : !cand ? 8 : 0;
            int kmw = part ? k : 0;                          // most mismatches any lane of the wave has to look at
#pragma unroll
            for (int s2 = 32; s2 > 0; s2 >>= 1) kmw = max(kmw, __shfl_xor(kmw, s2));
            if (kmw > 0) {
                int cw[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) cw[c] = __popcll(uniqA[c]);
                auto Cpre = [&](int x) -> int {              // unique-matching k-mer starts in [0, x)
                    x = min(max(x, 0), 256);
                    const int wi = x >> 6, sh = x & 63;
                    const u64 wsel = wi == 0 ? uniqA[0] : wi == 1 ? uniqA[1] : wi == 2 ? uniqA[2] : wi == 3 ? uniqA[3] : 0ull;
                    const int below = (wi > 0 ? cw[0] : 0) + (wi > 1 ? cw[1] : 0) + (wi > 2 ? cw[2] : 0) + (wi > 3 ? cw[3] : 0);
                    return below + __popcll(wsel & ((1ull << sh) - 1ull));
                };
                auto Wof = [&](int n) -> int { return (max(n, 0) + 6) / 7; };
                // n_low: how many of the read's bases may cost less than LOWQ (n disjoint windows are worth V(n) = mq min(n, n_low) + LOWQ max(n - n_low, 0))
                const int nlow = (shortcuts & SHORTCUT_NLOW) ? (int)(ri.aux & 0xFFFFu) : 0x7FFF;
                // the mismatches in read order (lanes with fewer than kmw repeat their last one with quality 0: its tests repeat too)
                const uint8_t* rq = b.read_qual + b.read_off[rb + (valid ? rl : 0)];
                u64 mm[4] = {missA[0], missA[1], missA[2], missA[3]};
                int pp[UNG_KMAX], qq[UNG_KMAX];
#pragma unroll
                for (int j = 0; j < UNG_KMAX; ++j) {
                    pp[j] = j ? pp[j - 1] : 0; qq[j] = 0;
                    if (j < kmw) {                           // wave-uniform
                        int pos = -1;
#pragma unroll
                        for (int c = 3; c >= 0; --c) if (mm[c]) pos = 64 * c + (int)__ffsll((long long)mm[c]) - 1;
                        if (part && pos >= 0) { pp[j] = pos; qq[j] = rq[pos]; }
#pragma unroll
                        for (int c = 0; c < 4; ++c) if (pos >= 0 && (pos >> 6) == c) mm[c] &= mm[c] - 1ull;
                    }
                }
                int U = 0;
#pragma unroll
                for (int j = 0; j < UNG_KMAX; ++j) U += qq[j];
                const int st = cidx - 8;
                int G = 127;
                if (part) for (int t = st >> 6; t <= (st + L + 14) >> 6; ++t) G = min(G, (int)s_gmin[t]);
                // everything below in units of 1/6 (v'' has a half and a third in it)
                const int v6 = min(3 * (G + 7), 2 * (G + 42));
                const int c_lo = min(6 * mq, v6), c_hi = min(6 * max(mq, (int)LOWQ), v6);
                auto phi6 = [&](int n) -> int { n = max(n, 0); return c_lo * min(n, nlow) + c_hi * max(n - nlow, 0); };
                // unique-matching starts before p_j - 6 and before p_j + 1 (none start in between: those 7-mers hold the mismatch)
                int cm6[UNG_KMAX], cp1[UNG_KMAX];
#pragma unroll
                for (int j = 0; j < UNG_KMAX; ++j) {
                    if (j < kmw) { cm6[j] = Cpre(pp[j] - 6); cp1[j] = Cpre(pp[j] + 1); }
                    else { cm6[j] = cm6[j - (j > 0)]; cp1[j] = cp1[j - (j > 0)]; }
                }
                const int Ctot = Cpre(L - 6);
                int Iw[UNG_KMAX];                            // windows between mismatch j and the next
#pragma unroll
                for (int j = 0; j + 1 < UNG_KMAX; ++j) Iw[j] = Wof(cm6[j + 1] - cp1[j]);
                Iw[UNG_KMAX - 1] = 0;
                int Wall = Wof(cm6[0]) + Wof(Ctot - cp1[UNG_KMAX - 1]);
#pragma unroll
                for (int j = 0; j + 1 < UNG_KMAX; ++j) Wall += Iw[j];
                int bad = phi6(Wall) < 6 * U ? 10 : 0;       // never on d*
                // an excursion at the head or the tail of the read: its gap next to d* is a deletion or a one-base insertion
                // (no window lost) or a longer insertion (>= 7 more, one window lost)
                auto edge = [&](int W, int T) -> bool { return 6 * (G - T) + min(phi6(W), 42 + phi6(W - 1)) >= 0; };
                int Rs = 0, Qs = U, Wbefore = Wof(cm6[0]);   // windows before mismatch j
#pragma unroll
                for (int j = 0; j < UNG_KMAX; ++j) {
                    if (j < kmw) {
                        Rs += qq[j];
                        if (!bad && !edge(Wbefore, Rs)) bad = 11;                     // elsewhere, then on d* from after p_j
                        if (!bad && !edge(Wall - Wbefore, Qs)) bad = 12;              // on d* up to a gap before p_j, then elsewhere
                        Wbefore += Iw[j];
                        Qs -= qq[j];
                        int T = qq[j], Wm = 0;
                        if (!bad && 2 * G + 2 < T) bad = 13;                          // an excursion around p_j alone
#pragma unroll
                        for (int j2 = j + 1; j2 < UNG_KMAX; ++j2) {
                            if (j2 < kmw) {
                                T += qq[j2];
                                Wm += Iw[j2 - 1];
                                const int slack = 6 * (2 * G - T);
                                const bool two = slack + 12 + phi6(Wm) >= 0 && slack + 60 + phi6(Wm - 1) >= 0;
                                const bool more = slack + max(6 * G, phi6(Wm - 2)) >= 0;
                                if (!bad && !(two && more)) bad = 14;
                            }
                        }
                    }
                }
                if (part && !bad) ung_score = U;
                else if (part) why = bad;
            }
        }
        const bool ungapped = ung_score >= 0;
        // extra job slot for (one candidate that is not the mapping position): one atomic per wave
        int base = 0;
        {
            const bool need = decided && live && ncand == 1 && !orig_in && !zero;
            const unsigned long long m = __ballot(need);
            if (m) {
                int wb = 0;
                if (lane == 0) wb = (int)atomicAdd((unsigned long long*)&cnt[CNT_NEXTRA], (unsigned long long)__popcll(m));
                wb = __shfl(wb, 0);
                base = wb + __popcll(m & ((1ull << lane) - 1ull));
                if (need && (long long)base + 1 <= (long long)extra_cap) jobs[npairs + base] = Job{ri.col, hq, idx0, L};
            }
        }
        // Decided pairs: the ones that need no DP are finished here (skipped read: 0.0, chaplotype.pyx:345-346; read < 7 bp or exact
        // match: score 0; ungapped alignment proven optimal: its score), the others leave a job in their slot.
        // (SEED_LEAN, the asynchronous entry point: nobody asks for statistics afterwards, and k_finalize_dense only looks at pairs with
        // a DP -- the 32 bytes of records of a finished pair, 7 pairs in 8 of a clean batch, are not written at all.)
        bool prim = false;                                   // the pair's primary slot holds a DP
        const bool recs = !(shortcuts & SEED_LEAN);
        if (valid && decided) {
            if (!live) {
                const bool sk = skipped || hapshort;
                if (recs) {
                    pairs[pidx] = PairRec{0, 0, (int16_t)(sk ? -1 : -2), 0, mapq, {0, 0, 0}};
                    jobs[pidx] = Job{ri.col, h, 0, 0};
                }
                out_ll[pidx] = sk ? 0.0 : loglik_of(0, mapq_lut, mapq);
                if (out_score) out_score[pidx] = sk ? -1 : 0;
            } else if (zero) {
                if (recs) {
                    pairs[pidx] = PairRec{0, L, (int16_t)-3, 0, mapq, {0, 0, 0}};
                    jobs[pidx] = Job{ri.col, h, cidx, 0};
                }
                out_ll[pidx] = loglik_of(0, mapq_lut, mapq);
                if (out_score) out_score[pidx] = 0;
            } else if (ungapped) {
                if (recs) {
                    pairs[pidx] = PairRec{ung_score, L, (int16_t)-4, 0, mapq, {0, 0, 0}};
                    jobs[pidx] = Job{ri.col, h, cidx, 0};
                }
                out_ll[pidx] = loglik_of(ung_score, mapq_lut, mapq);
                if (out_score) out_score[pidx] = ung_score;
            } else {
                jobs[pidx] = Job{ri.col, hq, cidx, L};
                pairs[pidx] = PairRec{base, idx0, (int16_t)ncand, (int16_t)(orig_in ? 0 : ncand), mapq, {0, 0, 0}};
                prim = true;
            }
        }
        if (shortcuts & 512) {                              // measurement only
            for (int r = 0; r < 16; ++r) {
                const unsigned long long m = __ballot(prim && why == r);
                if (m && lane == 0) atomicAdd((unsigned long long*)&cnt[32 + r], (unsigned long long)__popcll(m));
            }
        }
        {   // the wave's live job slots join the dense list: primary slots, then the extra ones, room reserved with one atomic
            const bool extra = decided && live && ncand == 1 && !orig_in && !zero && valid && (long long)base + 1 <= (long long)extra_cap;
            const unsigned long long m1 = __ballot(prim), m2 = __ballot(extra);
            const int n1 = __popcll(m1), n2 = __popcll(m2);
            if (n1 + n2) {
                const int seg = w % DENSE_SEGS;
                long long db = 0;
                if (lane == 0) db = (long long)atomicAdd((unsigned long long*)dense_counter(cnt, seg), (unsigned long long)(n1 + n2));
                db = ((long long)(unsigned)__shfl((int)db, 0)) | ((long long)__shfl((int)(db >> 32), 0) << 32);
                if (db + n1 + n2 <= segcap) {
                    const unsigned long long below = (1ull << lane) - 1ull;
                    if (prim) dense[seg * segcap + db + __popcll(m1 & below)] = (int32_t)pidx;
                    if (extra) dense[seg * segcap + db + n1 + __popcll(m2 & below)] = (int32_t)(npairs + base);
                }
            }
        }
        // ---- pairs that could not be decided go to the exact vote in k_seed_slow (one wave per pair, spread over the
        // whole device: a tandem-repeat window would otherwise serialise all its reads on this one wave)
        const unsigned long long todo = __ballot(valid && !decided);
        if (todo) {
            long long sb = 0;
            if (lane == 0) sb = (long long)atomicAdd((unsigned long long*)&cnt[CNT_SLOW_SEED], (unsigned long long)__popcll(todo));
            sb = ((long long)(unsigned)__shfl((int)sb, 0)) | ((long long)__shfl((int)(sb >> 32), 0) << 32);
            if (valid && !decided) slow_list[sb + __popcll(todo & ((1ull << lane) - 1ull))] = SlowRec{h, rl};
        }
    }
}

__device__ __forceinline__ void wave_lds_sync() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }

// k_seed_slow: the exact vote for the pairs the seeding kernels queued (a few hundred per launch on config 2, ~11 k per launch of the WGS
// job: every read over a tandem repeat, for every haplotype of its window).  Round 6: a workgroup is SLOW_WAVES waves and takes groups of
// SLOW_GROUP consecutive queue entries (the entries one seeding wave queued are consecutive and mostly share their haplotype).  Inside a
// group every run of entries of one haplotype costs ONE index build, made by all the threads together (measured: a build by one wave is 18
// of the 27 us a pair cost -- linear probing at load 0.65, the slowest of 64 lanes sets each round's time); its waves then vote for one entry
// each, each wave in its own diagonal counters, leaving the pair's arg-max diagonals in an LDS record.  Then the job slots and dense-list
// entries of the whole group are reserved with ONE returning atomic per counter (before: two per pair on the same two addresses, ~23 k per
// launch against the L2's ~90 per microsecond and address) and the waves write their pairs' jobs.  Same LDS carve as k_seed up to the
// counters, of which there is one set per wave.
constexpr int SLOW_WAVES = 4, SLOW_GROUP = 32;
__device__ unsigned long long g_slow_ticks[8];            // PLAT_SLOW_TIMING=1 (measurement): thread 0's 100 MHz ticks per phase, summed over groups
__global__ void __launch_bounds__(64 * SLOW_WAVES)
k_seed_slow(plat_window_batch b, const int32_t* __restrict__ hap_win, const long long* __restrict__ tile_off,
            const ReadInfo* __restrict__ rinfo, const uint16_t* __restrict__ codes, PairRec* __restrict__ pairs,
            Job* __restrict__ jobs, long long npairs, int extra_cap, long long* cnt, const SlowRec* __restrict__ slow_list,
            int tsize_max, int maxhap, int cw, int32_t* __restrict__ dense, long long segcap, int timing, int group)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ SlowOut s_out[SLOW_GROUP];
    __shared__ SlowRec s_ent[SLOW_GROUP];
    __shared__ long long s_dbase[DENSE_SEGS];
    const int nw64 = ((maxhap + 63) >> 6) + 8;
    unsigned* table = (unsigned*)smem;
    unsigned short* nxt = (unsigned short*)(smem + (size_t)tsize_max * 4);
    u64* h0 = (u64*)(smem + (size_t)tsize_max * 4 + (((size_t)maxhap + 2) * 2 + 7 & ~(size_t)7));
    u64* h1 = h0 + nw64;
    u64* nup = h1 + 2 * nw64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwv = (int)blockDim.x >> 6;   // (SLOW_WAVES waves, fewer when the counters of a long haplotype would not fit)
    unsigned* counts = (unsigned*)(nup + nw64) + (size_t)wave * (size_t)(cw >> 1);      // this wave's diagonal counters
    int* s_scal = (int*)((unsigned*)(nup + nw64) + (size_t)nwv * (size_t)(cw >> 1));
    long long nslow = cnt[CNT_SLOW_SEED];
    if (cnt[CNT_ERR] != 0) return;                       // an earlier stage refused the batch
    if (nslow > npairs) nslow = npairs;
    // entries per workgroup round: as few as fill the grid once -- a launch with a few hundred entries (config 2) gives every entry its own workgroup
    // (its entries share no haplotype and a group is worked through run by run), one with ten thousand (a chunk of the WGS job) takes `group` at a time
    group = (int)min((long long)group, max(1ll, (nslow + (long long)gridDim.x - 1) / (long long)gridDim.x));
    if ((long long)group * blockIdx.x >= nslow) return;                              // (most workgroups of a launch: nothing queued for them)
    for (int j = lane; j < (cw >> 1); j += 64) counts[j] = 0u;
    unsigned long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0}, acc[4] = {0, 0, 0, 0};
#define SLOW_TICK(i) do { if (timing) tk[i] = wall_clock64(); } while (0)
    for (long long g = blockIdx.x; group * g < nslow; g += gridDim.x) {
        const long long e0 = group * g, gEnd = min(nslow, e0 + group);
        const int nent = (int)(gEnd - e0);
        long long e = e0;
        acc[0] = acc[1] = acc[2] = 0;
        __syncthreads();
        if (tid < nent) s_ent[tid] = slow_list[e0 + tid];                                 // the group's entries in one round trip (a load per entry and thread before)
        __syncthreads();
        while (e < gEnd) {                                                                // (uniform over the workgroup: everyone reads the same entries)
            SLOW_TICK(0);
            const int h = s_ent[e - e0].hap, w = hap_win[h];
            long long run = e + 1;
            while (run < gEnd && s_ent[run - e0].hap == h) ++run;
            const long long hoff = b.hap_off[h];
            const int hapLen = (int)(b.hap_off[h + 1] - hoff);
            const uint8_t* hs = b.hap_seq + hoff;
            const bool direct = hapLen > 4096;
            int tsize = 64;
            if (direct) tsize = 16384;
            else while (tsize < hapLen + hapLen / 4) tsize <<= 1;
            const unsigned tmask = (unsigned)tsize - 1u;
            const int nch = (hapLen + 63) >> 6;
            __syncthreads();                                                             // (the votes of the run before are over: the table may go)
            for (int i = tid; i < 2 * nw64; i += 64 * nwv) h0[i] = 0ull;                 // h0, h1 contiguous
            __syncthreads();
            for (int t0 = 16 * wave; t0 < nch; t0 += 16 * nwv) {                          // 16 chunks of bytes per memory round trip and wave
                unsigned by[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int p = 64 * (t0 + k) + lane;
                    by[k] = p < hapLen ? (unsigned)hs[p] : 0u;
                }
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int p = 64 * (t0 + k) + lane;
                    const unsigned b2 = p < hapLen ? base2(by[k]) : 0u;
                    const u64 m0 = __ballot(b2 & 1u), m1 = __ballot(b2 & 2u);
                    if (lane == 0 && t0 + k < nch) { h0[t0 + k] = m0; h1[t0 + k] = m1; }
                }
            }
            __syncthreads();
            SLOW_TICK(1);
            seed_build_index(table, nxt, h0, h1, nup, s_scal, hapLen, nch, direct, tsize, tmask, false);
            SLOW_TICK(2);
            const int rb = b.win_read_begin[w], R = b.win_read_begin[w + 1] - rb;
            const int hapStart = b.win_start[w] - b.win_flank[w];                        // chaplotype.pyx:606
            for (long long q = e + wave; q < run; q += nwv) {                             // one entry per wave at a time: the vote, its arg-max diagonals into the record
                const int rl = s_ent[q - e0].rl;
                const ReadInfo ri = rinfo[rb + rl];
                const int L = (int)(ri.lfm & 0xFFFFu);
                const int idx0 = min(ri.pos - hapStart, hapLen - L - 15);                // calign.pyx:252
                const long long pidx = b.pair_off[w] + (long long)(h - b.win_hap_begin[w]) * R + rl;
                const u64* scp = (const u64*)(codes + tile_off[w]) + rl;
                SlowOut* o = &s_out[q - e0];
                const int hflag = h | (((ri.lfm >> 18) & 1u) ? JOB_BIGQ : 0);
                seed_vote_collect(table, nxt, counts, direct, tmask, hapLen, scp, R, L, idx0, o);
                if (lane == 0) { o->pidx = pidx; o->h = hflag; o->sL = L; o->sidx0 = idx0; o->scol = ri.col; o->smapq = (int)(ri.lfm >> 24); o->seg = w % DENSE_SEGS; }
                wave_lds_sync();
                if (o->sncand > SLOW_MAXC) {
                    // more arg-max diagonals than the record holds (long tandem repeats): this pair alone, with its own reservations
                    seed_exact_vote(table, nxt, counts, direct, tmask, hapLen, hflag, scp, R, L, idx0, ri.col, (int)(ri.lfm >> 24), pidx,
                                    npairs, extra_cap, jobs, pairs, cnt, dense, segcap, w % DENSE_SEGS);
                    if (lane == 0) o->njobs = -1;
                }
            }
            SLOW_TICK(3);
            if (timing) { acc[0] += tk[1] - tk[0]; acc[1] += tk[2] - tk[1]; acc[2] += tk[3] - tk[2]; }
            e = run;
        }
        __syncthreads();
        SLOW_TICK(4);
        // ONE reservation per counter for the group, by the first wave (a lane per entry; SLOW_GROUP <= 64): extra job slots (pairs with more than
        // one job) first -- whether a pair's jobs fit decides whether it joins the dense list --, then the dense list per segment
        if (wave == 0) {
            const bool mine = lane < nent && s_out[lane < nent ? lane : 0].njobs >= 0;
            const int nj = mine ? s_out[lane].njobs : 0, seg = mine ? s_out[lane].seg : -1;
            int ex = nj > 1 ? nj - 1 : 0, exIncl = ex;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(exIncl, d); if (lane >= d) exIncl += v; }
            const int exTotal = __shfl(exIncl, 63);
            long long base = 0;
            if (lane == 0 && exTotal > 0) base = (long long)atomicAdd((unsigned long long*)&cnt[CNT_NEXTRA], (unsigned long long)exTotal);
            base = ((long long)(unsigned)__shfl((int)base, 0)) | ((long long)__shfl((int)(base >> 32), 0) << 32);
            const int sbase = nj > 1 ? (int)(base + exIncl - ex) : 0;
            const bool fits = nj <= 1 || (long long)sbase + (nj - 1) <= (long long)extra_cap;
            long long db = -1;
            int mytot = 0;
#pragma unroll
            for (int k = 0; k < DENSE_SEGS; ++k) {
                const int v = (mine && fits && seg == k) ? nj : 0;
                int incl = v;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const int u = __shfl_up(incl, d); if (lane >= d) incl += u; }
                const int tot = __shfl(incl, 63);
                if (seg == k && mine && fits) db = incl - v;
                if (lane == k) mytot = tot;
            }
            if (lane < DENSE_SEGS) s_dbase[lane] = mytot > 0 ? (long long)atomicAdd((unsigned long long*)dense_counter(cnt, lane), (unsigned long long)mytot) : 0;
            wave_lds_sync();
            if (mine) { SlowOut& o = s_out[lane]; o.sbase = sbase; o.fits = fits ? 1 : 0; o.db = db >= 0 ? db + s_dbase[seg] : -1; }
        }
        __syncthreads();
        SLOW_TICK(5);
        for (int t = wave; t < nent; t += nwv)
            if (s_out[t].njobs >= 0) seed_vote_emit(&s_out[t], npairs, jobs, pairs, dense, segcap);
        __syncthreads();
        SLOW_TICK(6);
        if (timing && tid == 0) {
            atomicAdd(&g_slow_ticks[0], acc[0]); atomicAdd(&g_slow_ticks[1], acc[1]); atomicAdd(&g_slow_ticks[2], acc[2]);
            atomicAdd(&g_slow_ticks[3], tk[5] - tk[4]); atomicAdd(&g_slow_ticks[4], tk[6] - tk[5]); atomicAdd(&g_slow_ticks[7], 1ull);
        }
    }
#undef SLOW_TICK
}


// ------------------------------------------------------------------------------------------------
// number of live job slots for the host (synchronous entry point, statistics) and for the traceback kernel's slabs
__global__ void k_dense_total(long long* cnt, long long segcap)
{
    long long t = 0;
    for (int k = 0; k < DENSE_SEGS; ++k) t += dense_count(cnt, k, segcap);
    cnt[CNT_NDENSE] = t;
}

// ------------------------------------------------------------------------------------------------
// One DP from the caller's bytes: rs / rq = the read's bases and qualities, hs / gs = the haplotype's bases and gap-open penalties
// at the slice start (calign.pyx:229,256).  Rows past the read's end are the reference's pads ('0', 64: align.c:223-226); the last
// extra step reads one haplotype position past the slice (a lane that never reaches the result, dp_core.hpp).
template <bool HAS_N, bool SWAR, bool UNPACKED>
__device__ __forceinline__ int dp_job(const uint8_t* __restrict__ rs, const uint8_t* __restrict__ rq, const uint8_t* __restrict__ hs,
                                      const uint8_t* __restrict__ gs, int len2)
{
    uint32_t w0[8];
    words_of_8(load_u64_unaligned(hs), load_u64_unaligned(gs), w0);
    hs += 8; gs += 8;
    auto rw = [&](int h) -> uint32_t { return h < len2 ? read_word(rs[h], rq[h]) : READ_PAD_WORD; };
    auto hw = [&](int h) -> uint32_t { return hap_word(hs[h], gs[h]); };
    auto rw8 = [&](int h) -> Raw8 { return Raw8{load_u64_unaligned(rs + h), load_u64_unaligned(rq + h)}; };
    auto hw8 = [&](int h) -> Raw8 { return Raw8{load_u64_unaligned(hs + h), load_u64_unaligned(gs + h)}; };
    if (UNPACKED) {
        DPU<HAS_N> dp;
        dp.init(w0);                                                         // gapextend 3, nucprior 2: chaplotype.pyx:607-608
        return dp_run_u<HAS_N>(dp, len2, rw, hw);
    } else {
        DP<HAS_N, SWAR> dp;
        dp.init(w0, 3, 2);
        auto rw16 = [&](int h) -> Raw16 { Raw16 x; load_16_unaligned(rs + h, x.a0, x.a1); load_16_unaligned(rq + h, x.b0, x.b1); return x; };
        auto hw16 = [&](int h) -> Raw16 { Raw16 x; load_16_unaligned(hs + h, x.a0, x.a1); load_16_unaligned(gs + h, x.b0, x.b1); return x; };
        return dp_run8<HAS_N, SWAR>(dp, len2, rw, hw, rw8, hw8, rw16, hw16);
    }
}

// One lane per live job slot (dense list).  Slot j < npairs is the primary DP of pair j.  Pairs that need a single DP (one
// candidate that is also the mapping position, or no candidate at all) are finished right here: score -> log-likelihood
// (a8).  Only pairs with several candidate DPs go through k_finalize_multi.
// Occupancy is pinned at 4 waves/SIMD: the kernel is VALU-issue bound (5 or 6 waves measured no faster), and the registers it
// leaves free let the latency-bound kernels of ANOTHER batch (other plat_ctx / stream) run next to it (bench.py --streams).
template <bool UNPACKED>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_dp_jobs(plat_window_batch b, const uint8_t* __restrict__ gob, const uint8_t* __restrict__ hap_has_n, const Job* __restrict__ jobs,
          const PairRec* __restrict__ pairs, const double* __restrict__ mapq_lut, long long npairs,
          const int32_t* __restrict__ dense, long long segcap, long long* __restrict__ cnt, long long extra_cap,
          int32_t* __restrict__ job_score, double* __restrict__ out_ll, int32_t* __restrict__ out_score, int dyn)
{
    if (cnt[CNT_ERR] != 0 || cnt[CNT_NEXTRA] > extra_cap) return;   // refused batch / job overflow (reported by the host)
    // a fixed grid walks the dense list in tiles of 256 jobs (the list's length is only known on the device)
    long long ndense = 0;
#pragma unroll
    for (int k = 0; k < DENSE_SEGS; ++k) ndense += dense_count(cnt, k, segcap);
    // dyn: every WAVE pulls tiles of 64 jobs with one atomic (cnt[CNT_DP_TILE], zeroed with the batch's counters) until the list is empty --
    // the waves of a launch that is several rounds deep finish together instead of in the order the fixed stride dealt them their tiles
    const int lane = threadIdx.x & 63;
    long long t0 = dyn ? 0 : (long long)blockIdx.x * blockDim.x;
    for (;;) {
        if (dyn) {
            unsigned long long x = 0;
            if (lane == 0) x = atomicAdd((unsigned long long*)&cnt[CNT_DP_TILE], 1ull);
            t0 = 64ll * (long long)(((unsigned long long)(unsigned)__shfl((int)x, 0)) | ((unsigned long long)(unsigned)__shfl((int)(x >> 32), 0) << 32));
        }
        if (t0 >= ndense) break;
        const long long t = t0 + (dyn ? lane : (int)threadIdx.x);
        const long long t0next = t0 + (long long)gridDim.x * blockDim.x;
        const bool active = t < ndense;
        const long long j = active ? dense_slot(dense, segcap, cnt, t) : 0;
        Job jb = Job{0, 0, 0, 0};
        if (active) jb = jobs[j];
        int has_n = 0;
        const int hap = job_hap(jb);
        const int bigq = active && (jb.hap & JOB_BIGQ) != 0;
        const int st = max(0, jb.idx - 8);                                       // calign.pyx:229,256
        long long hoff = 0, roff = 0;
        if (active) {
            has_n = hap_has_n[hap];
            hoff = b.hap_off[hap] + st;
            roff = b.read_off[jb.col];
        }
        const uint8_t* hs = b.hap_seq + hoff;
        const uint8_t* gs = gob + hoff;
        const uint8_t* rs = b.read_seq + roff;
        const uint8_t* rq = b.read_qual + roff;
        int sc = 0;
        // wave-uniform choice of the code path: haplotype N's need the extra mask; the 32-bit SWAR adds are only taken when
        // every read of the wave has a quality sum that rules out a carry between the packed halves (dp_core.hpp)
        const bool anyN = __any(has_n), anyBig = UNPACKED || __any(bigq);
        if (anyN) {
            if (anyBig) { if (active) sc = dp_job<true, false, UNPACKED>(rs, rq, hs, gs, jb.len); }
            else        { if (active) sc = dp_job<true, true, UNPACKED>(rs, rq, hs, gs, jb.len); }
        } else {
            if (anyBig) { if (active) sc = dp_job<false, false, UNPACKED>(rs, rq, hs, gs, jb.len); }
            else        { if (active) sc = dp_job<false, true, UNPACKED>(rs, rq, hs, gs, jb.len); }
        }
        if (active) {
            if (j >= npairs) job_score[j] = sc;
            else {
                const PairRec pr = pairs[j];                                     // read after the DP: nothing of it is live across the loop
                if (pr.ncand == 0 || (pr.ncand == 1 && pr.orig_k == 0)) {
                    out_ll[j] = loglik_of(sc, mapq_lut, pr.mapq);
                    if (out_score) out_score[j] = sc;
                } else job_score[j] = sc;
            }
        }
        if (!dyn) t0 = t0next;
    }
}

// --calculateFlankScore=1 (a2): the same job list, but every DP runs in the reference's traceback mode and its score is
// reduced by the part of the alignment that lies in the haplotype's flanks (calign.pyx:235-245,261-264).  Jobs
// [j0, j0+jn) of the list; bpbuf holds 2*(maxread+8) rows of `bstride` 64-bit back-pointer words.
__global__ void __launch_bounds__(256)
k_dp_tb_jobs(plat_window_batch b, const int32_t* __restrict__ hap_win, const uint8_t* __restrict__ gob,
             const Job* __restrict__ jobs, const PairRec* __restrict__ pairs,
             const double* __restrict__ mapq_lut, long long npairs, const int32_t* __restrict__ dense, long long segcap, long long j0, long long jn,
             const long long* __restrict__ cnt, long long extra_cap, unsigned long long* __restrict__ bpbuf, long long bstride, int32_t* __restrict__ job_score,
             double* __restrict__ out_ll, int32_t* __restrict__ out_score)
{
    if (cnt[CNT_ERR] != 0 || cnt[CNT_NEXTRA] > extra_cap) return;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= jn) return;
    const long long j = dense_slot(dense, segcap, cnt, j0 + t);
    if (j < 0) return;
    const Job jb = jobs[j];
    int sc;
    {
        const int hap = job_hap(jb);
        const int w = hap_win[hap];
        const long long hoff = b.hap_off[hap];
        const int hapLen = (int)(b.hap_off[hap + 1] - hoff), hapFlank = b.win_flank[w];
        const int st = max(0, jb.idx - 8);                                   // calign.pyx:229,256
        const uint8_t* hfull = b.hap_seq + hoff;
        const uint8_t* gfull = gob + hoff;
        const uint8_t* rs = b.read_seq + b.read_off[jb.col];
        const uint8_t* rq = b.read_qual + b.read_off[jb.col];
        const int len2 = jb.len;
        uint32_t w0[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) w0[k] = hap_word(hfull[st + k], gfull[st + k]);
        auto rw = [&](int h) -> uint32_t { return h < len2 ? read_word(rs[h], rq[h]) : READ_PAD_WORD; };
        auto hw = [&](int h) -> uint32_t { return hap_word(hfull[st + 8 + h], gfull[st + 8 + h]); };
        const TbView bp{bpbuf + t, (size_t)bstride};
        int midx;
        sc = dp_forward_tb(w0, len2, rw, hw, bp, &midx);
        if (sc > 0) {
            auto hwf = [&](int xg) -> uint32_t { return hap_word(hfull[xg], gfull[xg]); };      // any position of the whole haplotype
            auto rwf = [&](int y) -> uint32_t { return read_word(rs[y], rq[y]); };
            sc -= tb_flank_score(bp, midx, len2, hwf, st, hapLen, hapFlank, rwf);
        }
    }
    if (j >= npairs) { job_score[j] = sc; return; }
    const PairRec pr = pairs[j];
    if (pr.ncand == 0 || (pr.ncand == 1 && pr.orig_k == 0)) {
        out_ll[j] = loglik_of(sc, mapq_lut, pr.mapq);
        if (out_score) out_score[j] = sc;
    } else job_score[j] = sc;
}

__global__ void __launch_bounds__(256)
k_dp_rows(int n, int lmax, const uint8_t* __restrict__ haps, const uint8_t* __restrict__ reads,
          const uint8_t* __restrict__ quals, const uint8_t* __restrict__ gos, const int32_t* __restrict__ len2,
          int gapextend, int nucprior, int32_t* __restrict__ out)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const size_t ho = (size_t)j * (lmax + 15), ro = (size_t)j * lmax;
    // The production constants go through the job kernel's own core (16 bytes per array and trip, dp_run8): its loads reach up to 16
    // bytes past a row's end, i.e. into the next row -- every row but the last takes it; the last row (and other constants) goes byte by byte.
    if (gapextend == 3 && nucprior == 2 && lmax >= 16 && j + 1 < n) out[j] = dp_job<true, false, false>(reads + ro, quals + ro, haps + ho, gos + ho, len2[j]);
    else out[j] = dp_score_bytes(haps + ho, gos + ho, reads + ro, quals + ro, len2[j], gapextend, nucprior);
}

// ------------------------------------------------------------------------------------------------
// The reference's candidate selection (calign.pyx:223-267) replayed on the job scores of one pair.
__device__ __forceinline__ int select_best(const PairRec& pr, long long p, long long npairs, const Job* __restrict__ jobs,
                                           const int32_t* __restrict__ job_score, int* ndp)
{
    int best = 1000000, bestPos = -1, n = 0;                                 // calign.pyx:190
    bool done = false;
    for (int k = 0; k < pr.ncand; ++k) {                                    // calign.pyx:223-247
        const long long js = job_slot(p, npairs, pr.extra_base, k);
        const int sc = job_score[js];
        ++n;
        if (sc < best) {
            best = sc; bestPos = jobs[js].idx;
            if (best == 0) { done = true; break; }
        }
    }
    if (!done && pr.idx0 != bestPos) {                                      // calign.pyx:255-267
        const int sc = job_score[job_slot(p, npairs, pr.extra_base, pr.orig_k)];
        ++n;
        if (sc < best) best = sc;
    }
    *ndp = n;
    return best;
}

__global__ void __launch_bounds__(256)
k_finalize_multi(const PairRec* __restrict__ pairs, const Job* __restrict__ jobs, const int32_t* __restrict__ job_score,
                 const double* __restrict__ mapq_lut, long long npairs, const long long* __restrict__ cnt, long long extra_cap,
                 double* __restrict__ out_ll, int32_t* __restrict__ out_score)
{
    if (cnt[CNT_ERR] != 0 || cnt[CNT_NEXTRA] > extra_cap) return;
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npairs) return;
    const PairRec pr = pairs[p];
    if (pr.ncand < 0 || pr.ncand == 0 || (pr.ncand == 1 && pr.orig_k == 0)) return;     // finished by k_dp_jobs
    int ndp;
    const int best = select_best(pr, p, npairs, jobs, job_score, &ndp);
    out_ll[p] = loglik_of(best, mapq_lut, pr.mapq);
    if (out_score) out_score[p] = best;
}

// The same over the dense list of live job slots (asynchronous entry point: k_seed left no record of the pairs it finished, SEED_LEAN):
// a pair with several candidate DPs has its primary slot in the list exactly once.
__global__ void __launch_bounds__(256)
k_finalize_dense(const PairRec* __restrict__ pairs, const Job* __restrict__ jobs, const int32_t* __restrict__ job_score,
                 const double* __restrict__ mapq_lut, long long npairs, const int32_t* __restrict__ dense, long long segcap,
                 const long long* __restrict__ cnt, long long extra_cap, double* __restrict__ out_ll, int32_t* __restrict__ out_score, long long* sticky)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        // what the host would have checked after its read-backs is recorded in a pinned host word that plat_stream_sync returns
        // (first error wins until it is read)
        long long e = cnt[CNT_ERR];
        if (e == 0 && cnt[CNT_NEXTRA] > extra_cap) e = PLAT_ERR_OVERFLOW;
        if (e != 0 && *sticky == 0) *sticky = e;
    }
    if (cnt[CNT_ERR] != 0 || cnt[CNT_NEXTRA] > extra_cap) return;
    long long ndense = 0;
#pragma unroll
    for (int k = 0; k < DENSE_SEGS; ++k) ndense += dense_count(cnt, k, segcap);
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < ndense; t += (long long)gridDim.x * blockDim.x) {
        const long long p = dense_slot(dense, segcap, cnt, t);
        if (p < 0 || p >= npairs) continue;
        const PairRec pr = pairs[p];
        if (pr.ncand < 0 || pr.ncand == 0 || (pr.ncand == 1 && pr.orig_k == 0)) continue;  // finished by k_dp_jobs
        int ndp;
        const int best = select_best(pr, p, npairs, jobs, job_score, &ndp);
        out_ll[p] = loglik_of(best, mapq_lut, pr.mapq);
        if (out_score) out_score[p] = best;
    }
}

// statistics for plat_align_stats (only launched when the caller asks for them)
__global__ void __launch_bounds__(256)
k_stats(const PairRec* __restrict__ pairs, const Job* __restrict__ jobs, const int32_t* __restrict__ job_score,
        long long npairs, long long* cnt)
{
    __shared__ unsigned long long s_acc[4];
    if (threadIdx.x < 4) s_acc[threadIdx.x] = 0ull;
    __syncthreads();
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long aligned = 0, ndp = 0, cells = 0;
    if (p < npairs) {
        const PairRec pr = pairs[p];
        if (pr.ncand != -1) {
            aligned = 1;
            if (pr.ncand == -3 || pr.ncand == -4) { ndp = 1; cells = 16ull * (unsigned long long)pr.idx0; }   // exact / ungapped: the reference runs one DP
            if (pr.ncand >= 0) {
                int n = 1;
                if (!(pr.ncand == 0 || (pr.ncand == 1 && pr.orig_k == 0))) select_best(pr, p, npairs, jobs, job_score, &n);
                ndp = (unsigned long long)n;
                cells = ndp * 16ull * (unsigned long long)jobs[p].len;
            }
        }
    }
    for (int s = 32; s > 0; s >>= 1) {
        aligned += __shfl_xor((long long)aligned, s);
        ndp += __shfl_xor((long long)ndp, s);
        cells += __shfl_xor((long long)cells, s);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&s_acc[0], aligned); atomicAdd(&s_acc[1], ndp); atomicAdd(&s_acc[2], cells);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd((unsigned long long*)&cnt[CNT_PAIRS_ALIGNED], s_acc[0]);
        atomicAdd((unsigned long long*)&cnt[CNT_NDP_REF], s_acc[1]);
        atomicAdd((unsigned long long*)&cnt[CNT_CELLS_REF], s_acc[2]);
    }
}

__global__ void k_sum_job_cells(const Job* __restrict__ jobs, long long njobs, long long* cnt)
{
    unsigned long long c = 0, n = 0;
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < njobs; j += (long long)gridDim.x * blockDim.x) {
        c += 16ull * jobs[j].len;
        n += jobs[j].len != 0;
    }
    for (int s = 32; s > 0; s >>= 1) { c += __shfl_xor((long long)c, s); n += __shfl_xor((long long)n, s); }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd((unsigned long long*)&cnt[CNT_CELLS_RUN], c);
        atomicAdd((unsigned long long*)&cnt[CNT_NJOBS_RUN], n);
    }
}

// asynchronous mode: what the host would have checked after its read-backs is recorded in a pinned host word that
// plat_stream_sync returns (first error wins until it is read)
__global__ void k_async_epilogue(const long long* __restrict__ cnt, long long extra_cap, long long* sticky)
{
    long long e = cnt[CNT_ERR];
    if (e == 0 && cnt[CNT_NEXTRA] > extra_cap) e = PLAT_ERR_OVERFLOW;
    if (e != 0 && *sticky == 0) *sticky = e;
}

}  // namespace plat

using namespace plat;

// =================================================================================================
PLAT_EXPORT int plat_dp_batch(plat_ctx* ctx, int n, int lmax, const uint8_t* hap_slices, const uint8_t* reads,
                              const uint8_t* quals, const uint8_t* gapopen, const int32_t* len2, int gapextend,
                              int nucprior, int32_t* out_score, void* stream)
{
    if (!ctx || n < 0 || lmax < 7) return PLAT_ERR_INVALID;
    if (n == 0) return PLAT_OK;
    if (!hap_slices || !reads || !quals || !gapopen || !len2 || !out_score) return PLAT_ERR_INVALID;
    PLAT_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_dp_rows, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, lmax, hap_slices,
                       reads, quals, gapopen, len2, gapextend, nucprior, out_score);
    PLAT_HIP(ctx, hipGetLastError());
    return PLAT_OK;
}

static int align_seed_launch(plat_ctx* ctx, const plat_window_batch& b, hipStream_t st, long long* cnt, int maxhap,
                             int maxread, int maxR, long long npairs, int extra_cap, const int32_t* hap_win, const int32_t* win_rows,
                             const long long* tile_off, int shortcuts, int32_t* dense, long long segcap, double* out_ll, int32_t* out_score,
                             const int32_t* wave_win, const int32_t* wave_first, long long wave_cap)
{
    int tsize_max = 64;                                        // in dwords
    if (maxhap > 4096) tsize_max = 8192;                       // direct mode: 16384 u16 heads
    else while (tsize_max < maxhap + maxhap / 4) tsize_max <<= 1;
    if (tsize_max < 1024) tsize_max = 1024;                    // the 2 x 512 dwords of the multiplicity maps overlay the table
    const int cw = (maxhap + maxread + 8 + 1) & ~1;            // 16-bit diagonal counters of the exact vote, even count
    const size_t nw64 = (((size_t)maxhap + 63) >> 6) + 8;
    const size_t lds0 = (size_t)tsize_max * 4 + ((((size_t)maxhap + 2) * 2 + 7) & ~(size_t)7) + 4 * nw64 * 8 + 16 + 64 + 128;
    const size_t lds = lds0 + ((nw64 + 15) & ~(size_t)15);    // k_seed: + one byte per chunk of 64 positions (gap-open minima)
    int slow_group = 8;                                         // entries per workgroup round (<= SLOW_GROUP); PLAT_SLOW_GROUP / PLAT_SLOW_WAVES: measurements
    if (const char* eg = getenv("PLAT_SLOW_GROUP")) slow_group = atoi(eg) > 0 && atoi(eg) <= SLOW_GROUP ? atoi(eg) : slow_group;
    int slow_waves = SLOW_WAVES;                                // (a set of diagonal counters per wave; fewer waves when long haplotypes make the sets large)
    if (const char* ew = getenv("PLAT_SLOW_WAVES")) slow_waves = atoi(ew) > 0 && atoi(ew) <= SLOW_WAVES ? atoi(ew) : slow_waves;
    while (slow_waves > 1 && lds0 + (size_t)slow_waves * (size_t)cw * 2 > 64 * 1024) slow_waves >>= 1;
    const size_t lds_slow = lds0 + (size_t)slow_waves * (size_t)cw * 2;
    if (lds_slow > 160 * 1024 || lds > 160 * 1024) return PLAT_ERR_HAP_TOO_LONG;
    if (lds > 48 * 1024) {
        PLAT_HIP(ctx, hipFuncSetAttribute((const void*)k_seed, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        PLAT_HIP(ctx, hipFuncSetAttribute((const void*)k_seed_base, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    if (lds_slow > 48 * 1024)
        PLAT_HIP(ctx, hipFuncSetAttribute((const void*)k_seed_slow, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_slow));
    // PLAT_SEED_SHARE=1: the windows' first haplotypes are swept once (k_seed_base) and the others derived from them where they can be.
    // Built, verified and measured in round 3: a quarter fewer vector instructions in k_seed and the same time -- a wave's time is its
    // chain of latencies, and deriving has as many round trips as sweeping; with k_seed_base's launch on top it is OFF by default.
    const char* e_sh = getenv("PLAT_SEED_SHARE");          // (read per call)
    const bool share = maxhap <= 4096 && !(shortcuts & 0x300) && e_sh && e_sh[0] == '1';
    unsigned char* basebuf = nullptr;
    if (share) {
        int rcb = plat_reserve(ctx, ctx->seedbase, (size_t)b.n_windows * seed_base_stride(maxhap) + 64);
        if (rcb) return rcb;
        basebuf = (unsigned char*)ctx->seedbase.ptr;
        hipLaunchKernelGGL(k_seed_base, dim3(b.n_windows), dim3(64), lds, st, b, (uint8_t*)ctx->hapw.ptr, (uint8_t*)ctx->hap_flags.ptr, cnt, basebuf,
                           tsize_max, maxhap, shortcuts);
    }
    // one wave per workgroup (the lazy index build is wave-local); blockIdx.y = group of SEED_CHUNKS x 64 reads
    const int ngroups = (maxR + SEED_CHUNKS * 64 - 1) / (SEED_CHUNKS * 64);
    const char* e_x = getenv("PLAT_SEED_XCD");             // (read per call; 0 = haplotype h on workgroup h)
    const bool xcd = !(e_x && e_x[0] == '0') && b.n_haps >= 64;
    if (xcd) shortcuts |= SEED_XCD;
    const unsigned gx = xcd ? (unsigned)((b.n_haps + 7) / 8) * 8u : (unsigned)b.n_haps;
    ctx->ev_split = 0;
    if (!wave_win)
        hipLaunchKernelGGL(k_seed, dim3(gx, ngroups > 0 ? ngroups : 1), dim3(64), lds, st, b, hap_win, win_rows, tile_off,
                           (const ReadInfo*)ctx->rinfo.ptr, (const uint16_t*)ctx->codes.ptr, (uint8_t*)ctx->hapw.ptr,
                           (uint8_t*)ctx->hap_flags.ptr, (PairRec*)ctx->pair_rec.ptr, (Job*)ctx->jobs.ptr, npairs, extra_cap, cnt,
                           (SlowRec*)ctx->slow.ptr, tsize_max, maxhap, shortcuts, dense, segcap,
                           (const double*)ctx->d_mapq_lut, out_ll, out_score, (const unsigned char*)basebuf);
    else {
        // the seeding stage as two kernels: the haplotype sweeps, then the (haplotype, read) pairs packed 64 to a wave
        const size_t stride = seed_state_stride(maxhap);
        int rcs = plat_reserve(ctx, ctx->seedstate, (size_t)(b.n_haps + SEED_NST + 1) * stride + 64);
        if (rcs) return rcs;
        const size_t lds_pairs = (size_t)tsize_max * 4 + ((((size_t)maxhap + 2) * 2 + 15) & ~(size_t)15) + (size_t)SEED_NST * stride + 64;
        if (lds_pairs > 160 * 1024) return PLAT_ERR_HAP_TOO_LONG;
        if (lds > 48 * 1024) PLAT_HIP(ctx, hipFuncSetAttribute((const void*)k_sweep, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (lds_pairs > 48 * 1024) PLAT_HIP(ctx, hipFuncSetAttribute((const void*)k_pairs, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pairs));
        { PLAT_KT_BEGIN(ctx, PLAT_KT_SWEEP, st); hipLaunchKernelGGL(k_sweep, dim3(gx), dim3(64), lds, st, b, (uint8_t*)ctx->hapw.ptr, (uint8_t*)ctx->hap_flags.ptr, cnt, tsize_max, maxhap, shortcuts,
                           (const unsigned char*)basebuf, hap_win, (unsigned char*)ctx->seedstate.ptr); PLAT_KT_END(ctx, PLAT_KT_SWEEP, st); }
        PLAT_EV(ctx, 8, st);
        ctx->ev_split = 1;
        if (!(shortcuts & 256)) {                              // (PLAT_SEED_DEBUG=256: the sweeps alone)
            const bool xw = (shortcuts & SEED_XCD) != 0;
            const unsigned gp = (unsigned)(xw ? ((wave_cap + 7) / 8) * 8 : wave_cap);
            { PLAT_KT_BEGIN(ctx, PLAT_KT_PAIRS, st); hipLaunchKernelGGL(k_pairs, dim3(gp > 0 ? gp : 1), dim3(64), lds_pairs, st, b, wave_win, wave_first, tile_off, (const ReadInfo*)ctx->rinfo.ptr,
                               (const uint16_t*)ctx->codes.ptr, (PairRec*)ctx->pair_rec.ptr, (Job*)ctx->jobs.ptr, npairs, extra_cap, cnt,
                               (SlowRec*)ctx->slow.ptr, tsize_max, maxhap, shortcuts, dense, segcap, (const double*)ctx->d_mapq_lut, out_ll, out_score,
                               (const unsigned char*)ctx->seedstate.ptr); PLAT_KT_END(ctx, PLAT_KT_PAIRS, st); }
        }
    }
    PLAT_EV(ctx, 5, st);                                       // k_seed alone: ev[1] .. ev[5]
    { PLAT_KT_BEGIN(ctx, PLAT_KT_SEED_SLOW, st); hipLaunchKernelGGL(k_seed_slow, dim3(2048), dim3(64 * slow_waves), lds_slow, st, b, hap_win, tile_off, (const ReadInfo*)ctx->rinfo.ptr,
                       (const uint16_t*)ctx->codes.ptr, (PairRec*)ctx->pair_rec.ptr, (Job*)ctx->jobs.ptr, npairs, extra_cap, cnt,
                       (const SlowRec*)ctx->slow.ptr, tsize_max, maxhap, cw, dense, segcap, getenv("PLAT_SLOW_TIMING") ? 1 : 0, slow_group); PLAT_KT_END(ctx, PLAT_KT_SEED_SLOW, st); }
    if (getenv("PLAT_SLOW_TIMING")) {
        unsigned long long t[8];
        PLAT_HIP(ctx, hipStreamSynchronize(st));
        PLAT_HIP(ctx, hipMemcpyFromSymbol(t, HIP_SYMBOL(g_slow_ticks), sizeof t));
        fprintf(stderr, "k_seed_slow: %llu groups; thread 0's mean us per group: loads + planes %.1f builds %.1f votes %.1f reserve %.1f emit %.1f\n", t[7],
                0.01 * t[0] / (t[7] ? t[7] : 1), 0.01 * t[1] / (t[7] ? t[7] : 1), 0.01 * t[2] / (t[7] ? t[7] : 1), 0.01 * t[3] / (t[7] ? t[7] : 1), 0.01 * t[4] / (t[7] ? t[7] : 1));
        memset(t, 0, sizeof t);
        PLAT_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(g_slow_ticks), t, sizeof t));
    }
    if (!(shortcuts & SEED_LEAN))                              // (the asynchronous entry point reads nothing back: every kernel sums the segments itself)
        hipLaunchKernelGGL(k_dense_total, dim3(1), dim3(1), 0, st, cnt, segcap);
    PLAT_HIP(ctx, hipGetLastError());
    return PLAT_OK;
}

// Shared body of the synchronous and the asynchronous entry point.  hints == NULL: the sizes come from two internal
// read-backs (after validation, after seeding); otherwise from the caller, nothing is read back and the device refuses
// the batch (PLAT_ERR_BAD_HINTS via plat_stream_sync) if the hints do not cover it.
static int align_impl(plat_ctx* ctx, const plat_window_batch* batch, const plat_batch_hints* hints, int calc_flank_score,
                      int use_mapq_cap, double* out_loglik, int32_t* out_score, plat_align_stats* out_stats, void* stream)
{
    if (!ctx || !batch) return PLAT_ERR_INVALID;
    if (use_mapq_cap) return PLAT_ERR_UNSUPPORTED;
    calc_flank_score = calc_flank_score != 0;
    const plat_window_batch b = *batch;
    if (b.n_windows < 0 || b.n_haps < 0 || b.n_reads < 0) return PLAT_ERR_INVALID;
    if (out_stats) memset(out_stats, 0, sizeof(*out_stats));
    if (b.n_windows == 0 || b.n_haps == 0) return PLAT_OK;
    if (!b.win_hap_begin || !b.win_read_begin || !b.win_start || !b.win_end || !b.win_flank || !b.pair_off ||
        !b.hap_seq || !b.hap_off || !b.read_off || !out_loglik)
        return PLAT_ERR_INVALID;
    if (b.n_reads > 0 && (!b.read_seq || !b.read_qual || !b.read_pos || !b.read_end || !b.read_mapq ||
                          !b.read_flags || !b.read_kind))
        return PLAT_ERR_INVALID;
    const bool async = hints != NULL;
    if (async && (hints->max_hap_len < 0 || hints->max_read_len < 0 || hints->max_reads_per_window < 0 || hints->n_pairs < 0 ||
                  hints->hap_blob_len < 0 || hints->read_blob_len < 0 || hints->extra_jobs_cap < 0))
        return PLAT_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    PLAT_HIP(ctx, hipSetDevice(ctx->device));

    // small per-batch arrays: counters | hap_win[n_haps] | win_rows[n_windows] | tile_off[n_windows]
    const size_t o_hapwin = (size_t)CNT_AREA * sizeof(long long);
    const size_t o_rows = o_hapwin + (((size_t)b.n_haps * 4 + 7) & ~(size_t)7);
    const size_t o_toff = o_rows + (((size_t)b.n_windows * 4 + 7) & ~(size_t)7);
    int rc = plat_reserve(ctx, ctx->counters, o_toff + (size_t)b.n_windows * 8 + 64);
    if (rc) return rc;
    char* cbase = (char*)ctx->counters.ptr;
    long long* cnt = (long long*)cbase;
    int32_t* hap_win = (int32_t*)(cbase + o_hapwin);
    int32_t* win_rows = (int32_t*)(cbase + o_rows);
    long long* tile_off = (long long*)(cbase + o_toff);
    if ((rc = plat_reserve(ctx, ctx->rinfo, (size_t)(b.n_reads + 1) * sizeof(ReadInfo)))) return rc;
    if ((rc = plat_reserve(ctx, ctx->hap_flags, (size_t)b.n_haps + 64))) return rc;

    ctx->ev_valid_align = 0;
    PLAT_EV(ctx, 0, st);
    PLAT_HIP(ctx, hipMemsetAsync(cnt, 0, (size_t)CNT_AREA * sizeof(long long), st));
    plat_batch_hints hv = {};
    if (async) hv = *hints;
    {   // a wave per window, a thread per haplotype; a small batch (a chunk of the region loop) does not pay for 2048 workgroups
        const long long want = std::max<long long>(((long long)b.n_windows * 64 + 255) / 256, ((long long)b.n_haps + 255) / 256);
        { PLAT_KT_BEGIN(ctx, PLAT_KT_VALIDATE, st); hipLaunchKernelGGL(k_validate, dim3((unsigned)std::min<long long>(2048, std::max<long long>(1, want))), dim3(256), 0, st, b, cnt, hap_win,
                           win_rows, calc_flank_score); PLAT_KT_END(ctx, PLAT_KT_VALIDATE, st); }
    }
    // waves of k_pairs: <= n_pairs / 64 + n_haps / 5 + n_windows (64 pairs per wave; windows with < 13 reads give a wave 5 whole haplotypes)
    // (asynchronous: the caller stated n_pairs, so the wave map is sized and built right here; synchronous: after the read-back below)
    const char* e_fused = getenv("PLAT_SEED_FUSED");           // =1: rounds 1-3's single seeding kernel (k_seed) instead of k_sweep + k_pairs (read per call)
    const bool seed_fused = e_fused && e_fused[0] == '1';
    auto wave_cap_for = [&](long long np) { return std::min<long long>(np / 64 + b.n_haps / 5 + b.n_windows + 8, 0x7FFFFF00ll); };
    long long wave_cap = 0;
    int32_t* wave_win = nullptr;
    int32_t* wave_first = nullptr;
    auto reserve_wave_map = [&](long long np) -> int {
        wave_cap = wave_cap_for(np);
        const int rcw = plat_reserve(ctx, ctx->seedmap, ((size_t)wave_cap + (size_t)b.n_windows + 16) * sizeof(int32_t));
        if (rcw) return rcw;
        wave_first = (int32_t*)ctx->seedmap.ptr;
        wave_win = wave_first + b.n_windows + 8;
        return PLAT_OK;
    };
    if (async && !seed_fused && (rc = reserve_wave_map(hv.n_pairs))) return rc;
    { PLAT_KT_BEGIN(ctx, PLAT_KT_TILE_SCAN, st); hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(1024), 0, st, b, win_rows, tile_off, cnt, hv, async ? 1 : 0, wave_win, wave_first, wave_cap); PLAT_KT_END(ctx, PLAT_KT_TILE_SCAN, st); }
    PLAT_HIP(ctx, hipGetLastError());
    int64_t* hb = ctx->h_readback;
    long long tile_total;
    if (!async) {
        // read back: error, maxima, blob lengths, number of pairs, tile size
        PLAT_HIP(ctx, hipMemcpyAsync(hb, cnt, CNT_N * sizeof(long long), hipMemcpyDeviceToHost, st));
        PLAT_HIP(ctx, hipStreamSynchronize(st));
        if (hb[CNT_ERR] != 0) return (int)hb[CNT_ERR];
        hv.max_hap_len = (int)hb[CNT_MAXHAP]; hv.max_read_len = (int)hb[CNT_MAXREAD]; hv.max_reads_per_window = (int)hb[CNT_MAXH];
        hv.hap_blob_len = hb[CNT_HAPBLOB]; hv.n_pairs = hb[CNT_NPAIRS]; hv.read_blob_len = hb[CNT_READBLOB];
        tile_total = hb[CNT_TILE_TOTAL];
        if (!seed_fused && hv.n_pairs > 0) {                   // the wave map of k_pairs, now that the number of pairs is known (the scan again: same offsets)
            if ((rc = reserve_wave_map(hv.n_pairs))) return rc;
            { PLAT_KT_BEGIN(ctx, PLAT_KT_TILE_SCAN, st); hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(1024), 0, st, b, win_rows, tile_off, cnt, hv, 0, wave_win, wave_first, wave_cap); PLAT_KT_END(ctx, PLAT_KT_TILE_SCAN, st); }
        }
    } else {
        tile_total = (long long)(hv.max_read_len + 8) * b.n_reads + 4ll * b.n_windows;     // upper bound of k_tile_scan's total
    }
    const int maxhap = hv.max_hap_len, maxread = hv.max_read_len, maxR = hv.max_reads_per_window;
    const long long hapblob = hv.hap_blob_len, npairs = hv.n_pairs;
    if (npairs == 0) return PLAT_OK;
    if (tile_total > 0x7FFFFFFFF0ll) return PLAT_ERR_OVERFLOW;
    if ((rc = plat_reserve(ctx, ctx->hapw, (size_t)hapblob + 256))) return rc;           // one gap-open byte per haplotype position
    if ((rc = plat_reserve(ctx, ctx->codes, ((size_t)tile_total + 64) * 2))) return rc;
    if ((rc = plat_reserve(ctx, ctx->pair_rec, (size_t)npairs * sizeof(PairRec)))) return rc;
    if ((rc = plat_reserve(ctx, ctx->slow, (size_t)npairs * sizeof(SlowRec)))) return rc;
    long long extra_cap = npairs / 4 + 4096;
    if (async && hv.extra_jobs_cap > 0) extra_cap = hv.extra_jobs_cap;
    if ((long long)(ctx->jobs.cap / sizeof(Job)) - npairs > extra_cap) extra_cap = (long long)(ctx->jobs.cap / sizeof(Job)) - npairs;
    if (extra_cap > 0x7FFFFF00ll) extra_cap = 0x7FFFFF00ll;

    const int prep_groups = maxR > 0 ? (maxR + 63) / 64 : 1;
    // LDS image of a group of 64 reads, sized by the batch's longest read: occupancy of this kernel is LDS-limited
    const int prep_qoff = 64 * ((std::min(maxread, PREP_LMAX) + 3) & ~3) + 32;   // + the 16-byte copy's slack at both ends
    const size_t prep_lds = (size_t)2 * prep_qoff + (size_t)((std::min(maxread, PREP_LMAX) + 63) >> 6) * 1024;
    if (prep_lds > 48 * 1024)
        PLAT_HIP(ctx, hipFuncSetAttribute((const void*)k_prep_reads, hipFuncAttributeMaxDynamicSharedMemorySize, (int)prep_lds));
    {
        const char* e_x = getenv("PLAT_SEED_XCD");           // (read per call; 0 = window w on workgroup w)
        const bool xcd = !(e_x && e_x[0] == '0') && b.n_windows >= 64;
        const unsigned gx = xcd ? (unsigned)((b.n_windows + 7) / 8) * 8u : (unsigned)b.n_windows;
        { PLAT_KT_BEGIN(ctx, PLAT_KT_PREP_READS, st); hipLaunchKernelGGL(k_prep_reads, dim3(gx, prep_groups), dim3(256), prep_lds, st, b, win_rows, tile_off,
                           (uint16_t*)ctx->codes.ptr, (ReadInfo*)ctx->rinfo.ptr, cnt, prep_qoff, xcd ? 1 : 0); PLAT_KT_END(ctx, PLAT_KT_PREP_READS, st); }
    }
    long long njobs = 0;
    PLAT_EV(ctx, 1, st);
    for (int attempt = 0; attempt < 2; ++attempt) {
        if ((rc = plat_reserve(ctx, ctx->jobs, (size_t)(npairs + extra_cap) * sizeof(Job)))) return rc;
        if (attempt > 0) {                     // (the first pass starts from the memset of all counters above)
            PLAT_HIP(ctx, hipMemsetAsync(&cnt[CNT_NEXTRA], 0, sizeof(long long), st));
            PLAT_HIP(ctx, hipMemsetAsync(&cnt[CNT_SLOW_SEED], 0, sizeof(long long), st));
            PLAT_HIP(ctx, hipMemsetAsync(cnt + 64, 0, (size_t)DENSE_SEGS * DENSE_CNT_STRIDE * sizeof(long long), st));
        }
        // the ungapped-alignment shortcut applies to plain scores only (the flank score needs the traceback);
        // PLAT_NO_UNGAPPED=1 sends every such pair through the DP instead (cross-check in tests/test_gpu_parity.py)
        const char* e_ung = getenv("PLAT_NO_UNGAPPED");    // (read per call: the test flips it inside one process)
        const int no_ungapped = e_ung && e_ung[0] == '1';
        const char* e_ex = getenv("PLAT_NO_EXACT");        // every reference DP is then run (bench.py's gcups_all_dp)
        const int no_exact = e_ex && e_ex[0] == '1';
        const char* e_dbg = getenv("PLAT_SEED_DEBUG");     // measurement only: 256 = k_seed stops after the haplotype sweep (results are garbage)
        const char* e_nl = getenv("PLAT_NO_NLOW");
        const char* e_bq = getenv("PLAT_UNGAPPED_BIGQ");   // measurement only: lets the ungapped proof take reads in the wrap regime too (tools/ungapped_crosscheck.py --bigq)
        const int shortcuts = ((!calc_flank_score && !no_ungapped) ? SHORTCUT_UNGAPPED : 0) | (no_exact ? 0 : SHORTCUT_EXACT) |
                              ((e_nl && e_nl[0] == '1') ? 0 : SHORTCUT_NLOW) | ((e_bq && e_bq[0] == '1') ? SHORTCUT_BIGQ : 0) |
                              (e_dbg ? (atoi(e_dbg) & 0x300) : 0) | (async ? SEED_LEAN : 0);
        // the dense list of live job slots is built by the seeding kernels themselves (DENSE_SEGS segments, each able to hold every slot)
        const long long segcap = npairs + extra_cap;
        if ((rc = plat_reserve(ctx, ctx->dense, ((size_t)segcap * DENSE_SEGS + 64) * sizeof(int32_t)))) return rc;
        if ((rc = align_seed_launch(ctx, b, st, cnt, maxhap, maxread, maxR, npairs, (int)extra_cap, hap_win, win_rows, tile_off,
                                    shortcuts, (int32_t*)ctx->dense.ptr, segcap, out_loglik, out_score, wave_win, wave_first, wave_cap))) return rc;
        njobs = npairs + extra_cap;
        if (async) break;                      // job overflow is caught on the device and reported by plat_stream_sync
        PLAT_HIP(ctx, hipMemcpyAsync(hb, cnt, CNT_N * sizeof(long long), hipMemcpyDeviceToHost, st));
        PLAT_HIP(ctx, hipStreamSynchronize(st));
        if (hb[CNT_ERR] != 0) return (int)hb[CNT_ERR];
        const long long nextra = hb[CNT_NEXTRA];
        njobs = npairs + nextra;
        if (nextra <= extra_cap) break;
        if (nextra > 0x7FFFFF00ll || attempt == 1) return PLAT_ERR_OVERFLOW;
        extra_cap = nextra;                    // tandem-rich batch: re-run the seeding with the exact capacity
    }
    // synchronous mode: exact number of live slots; asynchronous: only the device knows it
    const long long segcap = npairs + extra_cap;
    const long long ngrid = async ? npairs + extra_cap : hb[CNT_NDENSE];
    const int32_t* dense = (const int32_t*)ctx->dense.ptr;
    if ((rc = plat_reserve(ctx, ctx->job_score, (size_t)(njobs + 1) * sizeof(int32_t)))) return rc;
    PLAT_EV(ctx, 2, st);
    if (ngrid == 0) {
        // nothing to align
    } else if (calc_flank_score) {
        // traceback mode: 2*(maxread+8) back-pointer words per job; the job list is processed in slabs of bounded size
        const long long rows = 2ll * (maxread + 8);
        long long slab = (long long)((6ull << 30) / ((unsigned long long)rows * 8ull));
        if (slab > ngrid) slab = ngrid;
        slab = (slab + 255) & ~255ll;
        if ((rc = plat_reserve(ctx, ctx->tb, (size_t)rows * (size_t)slab * 8))) return rc;
        for (long long j0 = 0; j0 < ngrid; j0 += slab) {
            const long long jn = ngrid - j0 < slab ? ngrid - j0 : slab;
            hipLaunchKernelGGL(k_dp_tb_jobs, dim3((unsigned)((jn + 255) / 256)), dim3(256), 0, st, b, hap_win,
                               (const uint8_t*)ctx->hapw.ptr, (const Job*)ctx->jobs.ptr,
                               (const PairRec*)ctx->pair_rec.ptr, ctx->d_mapq_lut, npairs, dense, segcap, j0, jn, cnt, extra_cap,
                               (unsigned long long*)ctx->tb.ptr, slab, (int32_t*)ctx->job_score.ptr, out_loglik, out_score);
        }
    } else {
        static int dp_impl = -1;                 // 1 = one int16 lane per VGPR (dp_unpacked.hpp), 0 = packed (dp_core.hpp)
        if (dp_impl < 0) { const char* e = getenv("PLAT_DP_IMPL"); dp_impl = e ? (strcmp(e, "unpacked") == 0) : 0; }   // packed measured faster
        // a fixed grid walks the list (its length lives on the device): two rounds of the blocks a device holds at 4 waves/SIMD
        static int dp_mult = -1;                                   // workgroups of the fixed grid per CU (PLAT_DP_GRID_PER_CU: measurements)
        if (dp_mult < 0) { const char* e = getenv("PLAT_DP_GRID_PER_CU"); dp_mult = e && atoi(e) > 0 ? atoi(e) : 8; }
        const long long want = (ngrid + 255) / 256, fixed = (long long)dp_mult * ctx->n_cu;
        // PLAT_DP_TILES=1 (measurement, round 5): every wave pulls 64-job tiles with one atomic instead of walking the list with a fixed
        // stride.  Measured SLOWER on both shapes -- config 2 (3 328 tiles, one per wave) k_dp_jobs 137 -> 180 us, every reference DP
        // executed (24.5 k tiles) 4 008 -> 3 971 GCUPS: thousands of atomics on one L2 address (~90 per us) cost more than the uneven
        // last round they remove.  Off by default.
        static int dp_dyn = -1;
        if (dp_dyn < 0) { const char* e = getenv("PLAT_DP_TILES"); dp_dyn = e ? (e[0] == '1') : 0; }
        // dynamic tiles: exactly the blocks the device holds at 4 waves/SIMD (4 per CU), each wave pulling tiles until none is left
        const long long resident = 4ll * ctx->n_cu;
        const dim3 grid((unsigned)(dp_dyn ? (want < resident ? want : resident) : (want < fixed ? want : fixed)));
        if (dp_impl)
            { PLAT_KT_BEGIN(ctx, PLAT_KT_DP_JOBS, st); hipLaunchKernelGGL(k_dp_jobs<true>, grid, dim3(256), 0, st, b, (const uint8_t*)ctx->hapw.ptr,
                               (const uint8_t*)ctx->hap_flags.ptr, (const Job*)ctx->jobs.ptr,
                               (const PairRec*)ctx->pair_rec.ptr, ctx->d_mapq_lut, npairs, dense, segcap, cnt, extra_cap,
                               (int32_t*)ctx->job_score.ptr, out_loglik, out_score, dp_dyn); PLAT_KT_END(ctx, PLAT_KT_DP_JOBS, st); }
        else
            { PLAT_KT_BEGIN(ctx, PLAT_KT_DP_JOBS, st); hipLaunchKernelGGL(k_dp_jobs<false>, grid, dim3(256), 0, st, b, (const uint8_t*)ctx->hapw.ptr,
                               (const uint8_t*)ctx->hap_flags.ptr, (const Job*)ctx->jobs.ptr,
                               (const PairRec*)ctx->pair_rec.ptr, ctx->d_mapq_lut, npairs, dense, segcap, cnt, extra_cap,
                               (int32_t*)ctx->job_score.ptr, out_loglik, out_score, dp_dyn); PLAT_KT_END(ctx, PLAT_KT_DP_JOBS, st); }
    }
    PLAT_EV(ctx, 3, st);
    if (async) {
        const long long want = (ngrid + 255) / 256, fixed = 8ll * ctx->n_cu;
        { PLAT_KT_BEGIN(ctx, PLAT_KT_FINALIZE, st); hipLaunchKernelGGL(k_finalize_dense, dim3((unsigned)(want < fixed ? std::max(want, 1ll) : fixed)), dim3(256), 0, st,
                           (const PairRec*)ctx->pair_rec.ptr, (const Job*)ctx->jobs.ptr, (const int32_t*)ctx->job_score.ptr,
                           ctx->d_mapq_lut, npairs, dense, segcap, cnt, extra_cap, out_loglik, out_score, (long long*)ctx->d_sticky); PLAT_KT_END(ctx, PLAT_KT_FINALIZE, st); }
    } else
        { PLAT_KT_BEGIN(ctx, PLAT_KT_FINALIZE, st); hipLaunchKernelGGL(k_finalize_multi, dim3((unsigned)((npairs + 255) / 256)), dim3(256), 0, st,
                           (const PairRec*)ctx->pair_rec.ptr, (const Job*)ctx->jobs.ptr,
                           (const int32_t*)ctx->job_score.ptr, ctx->d_mapq_lut, npairs, cnt, extra_cap, out_loglik, out_score); PLAT_KT_END(ctx, PLAT_KT_FINALIZE, st); }
    PLAT_EV(ctx, 4, st);
    if (async) {                                               // (k_finalize_dense left the batch's verdict in the sticky word)
        PLAT_HIP(ctx, hipGetLastError());
        ctx->ev_valid_align = ctx->profile;
        ctx->prof_dp_jobs = 0; ctx->prof_dp_bytes = 0;
        return PLAT_OK;
    }
    if (out_stats || ctx->profile)
        hipLaunchKernelGGL(k_sum_job_cells, dim3(256), dim3(256), 0, st, (const Job*)ctx->jobs.ptr, njobs, cnt);
    if (out_stats)
        hipLaunchKernelGGL(k_stats, dim3((unsigned)((npairs + 255) / 256)), dim3(256), 0, st,
                           (const PairRec*)ctx->pair_rec.ptr, (const Job*)ctx->jobs.ptr,
                           (const int32_t*)ctx->job_score.ptr, npairs, cnt);
    PLAT_HIP(ctx, hipGetLastError());
    if (out_stats || ctx->profile) {
        PLAT_HIP(ctx, hipMemcpyAsync(hb, cnt, 64 * sizeof(long long), hipMemcpyDeviceToHost, st));
        PLAT_HIP(ctx, hipStreamSynchronize(st));
        if (getenv("PLAT_SEED_DEBUG") && (atoi(getenv("PLAT_SEED_DEBUG")) & 512)) {
            fprintf(stderr, "k_seed, pairs that left a DP job, by reason:");
            for (int r = 0; r < 16; ++r) fprintf(stderr, " %lld", (long long)hb[32 + r]);
            fprintf(stderr, "\n");
        }
        ctx->ev_valid_align = ctx->profile;
        ctx->prof_dp_jobs = hb[CNT_NJOBS_RUN];
        ctx->prof_dp_bytes = hb[CNT_CELLS_RUN] / 4 + 34 * hb[CNT_NJOBS_RUN];     // sum(4*len2 + 34); cells = 16*len2
    }
    if (out_stats) {
        out_stats->n_pairs = npairs;
        out_stats->n_pairs_aligned = hb[CNT_PAIRS_ALIGNED];
        out_stats->n_dp_launched = hb[CNT_NJOBS_RUN];
        out_stats->n_dp_reference = hb[CNT_NDP_REF];
        out_stats->cells_reference = hb[CNT_CELLS_REF];
        out_stats->cells_launched = hb[CNT_CELLS_RUN];
        out_stats->n_seed_fallback = hb[CNT_SLOW_SEED];
    }
    return PLAT_OK;
}

PLAT_EXPORT int plat_align_window_batch(plat_ctx* ctx, const plat_window_batch* batch, int calc_flank_score,
                                        int use_mapq_cap, double* out_loglik, int32_t* out_score,
                                        plat_align_stats* out_stats, void* stream)
{
    return align_impl(ctx, batch, NULL, calc_flank_score, use_mapq_cap, out_loglik, out_score, out_stats, stream);
}

PLAT_EXPORT int plat_align_window_batch_async(plat_ctx* ctx, const plat_window_batch* batch, const plat_batch_hints* hints,
                                              int calc_flank_score, int use_mapq_cap, double* out_loglik,
                                              int32_t* out_score, void* stream)
{
    if (!hints) return PLAT_ERR_INVALID;
    return align_impl(ctx, batch, hints, calc_flank_score, use_mapq_cap, out_loglik, out_score, NULL, stream);
}
