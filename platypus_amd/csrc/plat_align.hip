// plat_align.hip -- device path for Haplotype.alignReads / alignSingleRead.
//
// Pipeline of plat_align_window_batch (one stream, one host read-back):
//   k_validate   : input checks + maxima                     (chaplotype.pyx:180-183 length rule)
//   k_hap_window : haplotype -> window map
//   k_seed       : one workgroup per haplotype: haplotype bytes staged in LDS, gap-open annotation
//                  (a7, chaplotype.pyx:552-590), 7-mer index in LDS (a4, calign.pyx:94-124), then one
//                  wave per read: diagonal vote (calign.pyx:206-220), arg-max candidate list in
//                  ascending order (calign.pyx:222-233) -> DP job list in HBM
//   k_dp_jobs    : one lane per banded DP (a1, align.c:77-586), see dp_core.hpp
//   k_finalize   : per (read, haplotype): the reference's candidate selection replayed on the job
//                  scores (calign.pyx:235-267), score -> log-likelihood (a8, chaplotype.pyx:621-676)
#include "dp_core.hpp"
#include "plat_internal.hpp"

namespace plat {

// Job slots: pair p owns slot p for its first DP (99% of pairs need exactly one); further candidate DPs
// (tandem repeats: many arg-max diagonals) go to an overflow area behind the npairs primary slots, reserved
// with one global atomic per such pair.  (A single job counter bumped by every pair saturates one L2
// atomic unit: ~90 atomics/us, i.e. ~20 ms for 2M pairs -- measured in round 1.)
struct PairRec { int32_t extra_base, ncand, orig_k, idx0; };   // ncand: -1 skipped read, -2 read shorter than 7
__device__ __forceinline__ long long job_slot(long long pair, long long npairs, int extra_base, int k) {
    return k == 0 ? pair : npairs + extra_base + (k - 1);
}
struct Job { int32_t read, hap, idx, len; };

enum { CNT_ERR = 0, CNT_MAXHAP, CNT_MAXREAD, CNT_NEXTRA, CNT_PAIRS_ALIGNED, CNT_NDP_REF, CNT_CELLS_REF, CNT_CELLS_RUN, CNT_NJOBS_RUN, CNT_N };

__constant__ signed char c_homopol_go[49] = {   // homopolq[i]-'!' (chaplotype.pyx:64-67); see tests/test_oracle.py
    45, 42, 41, 39, 37, 32, 28, 23, 20, 19, 17, 16, 15, 14, 13, 12, 11, 11, 10, 9, 9, 8, 8, 7, 7, 7, 6, 6, 6, 5, 5, 5,
    4, 4, 4, 3, 3, 3, 3, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1};

__device__ __forceinline__ void set_err(long long* cnt, int code) {
    atomicCAS((unsigned long long*)&cnt[CNT_ERR], 0ull, (unsigned long long)(long long)code);
}

// ------------------------------------------------------------------------------------------------
__global__ void k_validate(plat_window_batch b, long long* cnt)
{
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    int maxhap = 0, maxread = 0;
    for (int h = tid; h < b.n_haps; h += nt) {
        long long len = b.hap_off[h + 1] - b.hap_off[h];
        if (len > 16384) set_err(cnt, PLAT_ERR_HAP_TOO_LONG);
        if (len < 0) set_err(cnt, PLAT_ERR_BAD_INPUT);
        maxhap = max(maxhap, (int)min(len, 1ll << 20));
    }
    for (int r = tid; r < b.n_reads; r += nt) {
        long long len = b.read_off[r + 1] - b.read_off[r];
        if (len < 0 || len > 32767) set_err(cnt, PLAT_ERR_BAD_INPUT);      // cAlignedRead.rlen is a short
        maxread = max(maxread, (int)min(max(len, 0ll), 1ll << 20));
    }
    for (int w = tid; w < b.n_windows; w += nt) {
        long long H = b.win_hap_begin[w + 1] - b.win_hap_begin[w], R = b.win_read_begin[w + 1] - b.win_read_begin[w];
        if (H < 0 || R < 0 || b.pair_off[w + 1] - b.pair_off[w] != H * R) set_err(cnt, PLAT_ERR_BAD_INPUT);
    }
    // 7-bit ASCII check over the blobs (the DP packs bases as byte << 9)
    {
        const long long nh = b.n_haps ? b.hap_off[b.n_haps] : 0, nr = b.n_reads ? b.read_off[b.n_reads] : 0;
        unsigned bad = 0;
        for (long long i = tid; i < nh; i += nt) bad |= b.hap_seq[i];
        for (long long i = tid; i < nr; i += nt) bad |= b.read_seq[i] | b.read_qual[i];
        if (bad & 0x80u) set_err(cnt, PLAT_ERR_BAD_INPUT);
    }
    atomicMax((unsigned long long*)&cnt[CNT_MAXHAP], (unsigned long long)maxhap);
    atomicMax((unsigned long long*)&cnt[CNT_MAXREAD], (unsigned long long)maxread);
}

__global__ void k_hap_window(plat_window_batch b, int32_t* hap_win)
{
    int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= b.n_windows) return;
    for (int h = b.win_hap_begin[w]; h < b.win_hap_begin[w + 1]; ++h) hap_win[h] = w;
}

// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned base2(unsigned ch) {      // calign.pyx:69-74
    unsigned c = ch & 7u;
    if (c == 7u) c = 2u;
    return c & 3u;
}
#define CNT16(c, j) (((c)[(j) >> 1] >> (16 * ((j) & 1))) & 0xFFFFu)
__device__ __forceinline__ unsigned tbl_slot(unsigned code, unsigned mask) { return (code * 40503u + (code >> 5)) & mask; }

// LDS carve (dynamic):  table u32[tsize_max] | next u16[maxhap+2] | hapb u8[maxhap+16] | counts u16[nw][cw]
// (counts are 16-bit, two per dword, updated with 32-bit LDS atomics: a count never exceeds readLen-7 < 65536)
// The k-mer index has two modes: haplotypes up to 4096 bp use a small open-addressing table (>= 2*hapLen
// entries, more workgroups per CU); longer ones (up to the reference's cap of 16384) index all 4^7 codes
// directly, as the reference does (calign.pyx:98-99).
__global__ void __launch_bounds__(256)
k_seed(plat_window_batch b, const int32_t* __restrict__ hap_win, uint8_t* __restrict__ go_blob,
       PairRec* __restrict__ pairs, Job* __restrict__ jobs, long long npairs, int extra_cap, long long* cnt,
       int tsize_max, int maxhap, int cw)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned* table = (unsigned*)smem;
    unsigned short* nxt = (unsigned short*)(smem + (size_t)tsize_max * 4);
    unsigned char* hapb = smem + (size_t)tsize_max * 4 + (((size_t)maxhap + 2) * 2 + 3 & ~(size_t)3);
    unsigned* counts_all = (unsigned*)(hapb + (((size_t)maxhap + 16) + 3 & ~(size_t)3));

    const int h = blockIdx.x;
    const int w = hap_win[h];
    const long long hoff = b.hap_off[h];
    const int hapLen = (int)(b.hap_off[h + 1] - hoff);
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nw = nthr >> 6;
    unsigned* counts = counts_all + (size_t)wave * (cw >> 1);

    // table size: power of two >= 2*hapLen (open addressing) or all 16384 codes (direct)
    const bool direct = hapLen > 4096;
    int tsize = 64;
    if (direct) tsize = 16384;
    else while (tsize < 2 * hapLen) tsize <<= 1;
    const unsigned tmask = (unsigned)tsize - 1u;

    for (int i = tid; i < tsize; i += nthr) table[i] = 0u;
    for (int i = tid; i < hapLen; i += nthr) hapb[i] = b.hap_seq[hoff + i];
    __syncthreads();

    // a7: gap-open annotation (chaplotype.pyx:552-590): table[min(48, #following bytes equal to this one)], 'N' -> table[0]
    for (int p = tid; p < hapLen; p += nthr) {
        unsigned char c = hapb[p];
        int run = 0;
        if (c != 'N') {
            for (int q = p + 1; q < hapLen && run < 48 && hapb[q] == c; ++q) ++run;
        }
        go_blob[hoff + p] = (uint8_t)c_homopol_go[run];
    }
    // a4: k-mer index (positions 0..hapLen-8; calign.pyx:109): open addressing on the 14-bit code,
    // entry = (code+1)<<16 | (pos+1); equal codes are chained through nxt[] (order is irrelevant to the vote)
    for (int p = tid; p < hapLen - 7; p += nthr) {
        unsigned code = 0;
#pragma unroll
        for (int k = 0; k < 7; ++k) code = (code << 2) + base2(hapb[p + k]);
        if (direct) {
            unsigned old = atomicExch(&table[code], (unsigned)(p + 1));
            nxt[p + 1] = (unsigned short)old;
            continue;
        }
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
    __syncthreads();

    const int rb = b.win_read_begin[w], re = b.win_read_begin[w + 1];
    const int R = re - rb;
    const int hl = h - b.win_hap_begin[w];
    const int wstart = b.win_start[w], wend = b.win_end[w], flank = b.win_flank[w];
    const int hapStart = wstart - flank;                                    // chaplotype.pyx:606

    for (int rl = wave; rl < R; rl += nw) {
        const int r = rb + rl;
        const long long pidx = b.pair_off[w] + (long long)hl * R + rl;
        const long long roff = b.read_off[r];
        const int L = (int)(b.read_off[r + 1] - roff);
        const int rpos = b.read_pos[r];
        // skip rule, chaplotype.pyx:343-346 / 358-361 (brokenMates are always aligned, :366-373)
        bool skip = false;
        if (b.read_kind[r] != 2) {
            int os = max(wstart, rpos), oe = min(wend, b.read_end[r]);
            int ov = oe > os ? oe - os : -1;
            skip = (b.read_flags[r] & 512) || ov < 7;
        }
        if (skip || L < 7) {                                                // calign.pyx:179-180
            if (lane == 0) { pairs[pidx] = PairRec{0, skip ? -1 : -2, 0, 0}; jobs[pidx] = Job{r, h, 0, 0}; }
            continue;
        }
        if (hapLen < L + 15) {
            if (lane == 0) { set_err(cnt, PLAT_ERR_HAP_TOO_SHORT); pairs[pidx] = PairRec{0, -1, 0, 0}; jobs[pidx] = Job{r, h, 0, 0}; }
            continue;
        }
        const int n = hapLen + L;
        for (int j = lane; j < ((n + 1) >> 1); j += 64) counts[j] = 0u;
        // diagonal vote, calign.pyx:209-220
        unsigned mymax = 0;
        const uint8_t* rs = b.read_seq + roff;
        for (int i = lane; i < L - 7; i += 64) {
            unsigned code = 0;
#pragma unroll
            for (int k = 0; k < 7; ++k) code = (code << 2) + base2(rs[i + k]);
            unsigned hidx;
            if (direct) hidx = table[code];
            else {
                const unsigned key = (code + 1u) << 16;
                unsigned slot = tbl_slot(code, tmask);
                unsigned e = table[slot];
                while (e != 0u && (e & 0xFFFF0000u) != key) { slot = (slot + 1u) & tmask; e = table[slot]; }
                hidx = e & 0xFFFFu;
            }
            while (hidx != 0u) {
                const int j = (int)hidx - i - 1 + L;
                const unsigned sh = 16u * (unsigned)(j & 1);
                unsigned c = ((atomicAdd(&counts[j >> 1], 1u << sh) >> sh) & 0xFFFFu) + 1u;
                mymax = max(mymax, c);
                hidx = nxt[hidx];
            }
        }
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) mymax = max(mymax, (unsigned)__shfl_xor((int)mymax, s));
        const unsigned maxcount = mymax;
        // candidates: counts == maxcount, ascending, idx + L + 15 < hapLen  (calign.pyx:222-228)
        int ncand = 0;
        if (maxcount > 0)
            for (int j0 = 0; j0 < n; j0 += 64) {
                int j = j0 + lane;
                bool is = j < n && CNT16(counts, j) == maxcount && (j - L) + L + 15 < hapLen;
                ncand += __popcll(__ballot(is));
            }
        int idx0 = min(rpos - hapStart, hapLen - L - 15);                   // calign.pyx:252
        const int j0i = idx0 + L;
        const bool orig_in = maxcount > 0 && j0i >= 0 && j0i < n && CNT16(counts, j0i) == maxcount && idx0 + L + 15 < hapLen;
        const int njobs = ncand + (orig_in ? 0 : 1);
        int base = 0;
        if (njobs > 1) {
            if (lane == 0) base = (int)atomicAdd((unsigned long long*)&cnt[CNT_NEXTRA], (unsigned long long)(njobs - 1));
            base = __shfl(base, 0);
        }
        const bool fits = njobs == 1 || (long long)base + (njobs - 1) <= (long long)extra_cap;
        int orig_k = ncand;
        if (maxcount > 0) {
            int k = 0;
            for (int j0 = 0; j0 < n; j0 += 64) {
                int j = j0 + lane;
                bool is = j < n && CNT16(counts, j) == maxcount && (j - L) + L + 15 < hapLen;
                unsigned long long bal = __ballot(is);
                if (is) {
                    int mypos = k + __popcll(bal & ((1ull << lane) - 1ull));
                    if (fits || mypos == 0) jobs[job_slot(pidx, npairs, base, mypos)] = Job{r, h, j - L, L};
                }
                if (orig_in && j0i >= j0 && j0i < j0 + 64)
                    orig_k = k + __popcll(bal & ((1ull << (j0i - j0)) - 1ull));
                k += __popcll(bal);
            }
        }
        if (lane == 0) {
            if (!orig_in && (fits || ncand == 0)) jobs[job_slot(pidx, npairs, base, ncand)] = Job{r, h, idx0, L};
            pairs[pidx] = PairRec{base, ncand, orig_k, idx0};
        }
    }
}

// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_dp_jobs(plat_window_batch b, const uint8_t* __restrict__ go_blob, const Job* __restrict__ jobs,
          long long njobs, int32_t* __restrict__ job_score)
{
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= njobs) return;
    const Job jb = jobs[j];
    if (jb.len == 0) return;                                                // slot of a skipped pair
    const int st = max(0, jb.idx - 8);                                      // calign.pyx:229,256
    const long long hoff = b.hap_off[jb.hap] + st;
    const long long roff = b.read_off[jb.read];
    job_score[j] = dp_score(b.hap_seq + hoff, go_blob + hoff, b.read_seq + roff, b.read_qual + roff, jb.len, 3, 2);
}

__global__ void __launch_bounds__(256)
k_dp_rows(int n, int lmax, const uint8_t* __restrict__ haps, const uint8_t* __restrict__ reads,
          const uint8_t* __restrict__ quals, const uint8_t* __restrict__ gos, const int32_t* __restrict__ len2,
          int gapextend, int nucprior, int32_t* __restrict__ out)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const size_t ho = (size_t)j * (lmax + 15), ro = (size_t)j * lmax;
    out[j] = dp_score(haps + ho, gos + ho, reads + ro, quals + ro, len2[j], gapextend, nucprior);
}

// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_finalize(plat_window_batch b, const PairRec* __restrict__ pairs, const Job* __restrict__ jobs,
           const int32_t* __restrict__ job_score, const double* __restrict__ mapq_lut, long long npairs,
           double* __restrict__ out_ll, int32_t* __restrict__ out_score, long long* cnt)
{
    __shared__ unsigned long long s_acc[4];
    if (threadIdx.x < 4) s_acc[threadIdx.x] = 0ull;
    __syncthreads();
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long aligned = 0, ndp = 0, cells = 0;
    if (p < npairs) {
        // window of this pair: largest w with pair_off[w] <= p
        int lo = 0, hi = b.n_windows;
        while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (b.pair_off[mid] <= p) lo = mid; else hi = mid; }
        // windows with zero pairs share offsets: step to the last window whose offset <= p and that is non-empty
        while (lo + 1 < b.n_windows && b.pair_off[lo + 1] <= p) ++lo;
        const int w = lo;
        const int rb = b.win_read_begin[w], R = b.win_read_begin[w + 1] - rb;
        const int rl = (int)((p - b.pair_off[w]) % R);
        const int r = rb + rl;
        const PairRec pr = pairs[p];
        double ll = 0.0;
        int score = -1;
        if (pr.ncand != -1) {
            int best = 0;
            if (pr.ncand >= 0) {
                best = 1000000;                                              // calign.pyx:190
                int bestPos = -1;
                bool done = false;
                const int L = jobs[p].len;
                for (int k = 0; k < pr.ncand; ++k) {                        // calign.pyx:223-247
                    const long long js = job_slot(p, npairs, pr.extra_base, k);
                    int sc = job_score[js];
                    ++ndp;
                    if (sc < best) {
                        best = sc; bestPos = jobs[js].idx;
                        if (best == 0) { done = true; break; }
                    }
                }
                if (!done && pr.idx0 != bestPos) {                          // calign.pyx:255-267
                    int sc = job_score[job_slot(p, npairs, pr.extra_base, pr.orig_k)];
                    ++ndp;
                    if (sc < best) best = sc;
                }
                cells = ndp * 16ull * (unsigned long long)L;
            }
            score = best;
            const double v = -0.23025850929940459 * (double)best + mapq_lut[b.read_mapq[r]];   // chaplotype.pyx:676
            ll = v > -300.0 ? v : -300.0;
            aligned = 1;
        }
        out_ll[p] = ll;
        if (out_score) out_score[p] = score;
    }
    // block reduction of the statistics
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
    for (int s = 32; s > 0; s >>= 1) n += __shfl_xor((long long)n, s);
    if ((threadIdx.x & 63) == 0) atomicAdd((unsigned long long*)&cnt[CNT_NJOBS_RUN], n);
    for (int s = 32; s > 0; s >>= 1) c += __shfl_xor((long long)c, s);
    if ((threadIdx.x & 63) == 0) atomicAdd((unsigned long long*)&cnt[CNT_CELLS_RUN], c);
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
                             int maxread, long long npairs, int extra_cap, const int32_t* hap_win)
{
    int tsize_max = 64;
    if (maxhap > 4096) tsize_max = 16384;
    else while (tsize_max < 2 * maxhap) tsize_max <<= 1;
    const int cw = (maxhap + maxread + 8 + 1) & ~1;            // 16-bit counters, even count
    const size_t fixed = (size_t)tsize_max * 4 + ((((size_t)maxhap + 2) * 2 + 3) & ~(size_t)3) +
                         ((((size_t)maxhap + 16) + 3) & ~(size_t)3);
    const size_t lds_cap = 160 * 1024;
    int nw = 4;
    while (nw > 1 && fixed + (size_t)nw * cw * 2 > lds_cap) nw >>= 1;
    const size_t lds = fixed + (size_t)nw * cw * 2;
    if (lds > lds_cap) return PLAT_ERR_HAP_TOO_LONG;
    if (lds > 64 * 1024)
        PLAT_HIP(ctx, hipFuncSetAttribute((const void*)k_seed, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_seed, dim3(b.n_haps), dim3(64 * nw), lds, st, b, hap_win, (uint8_t*)ctx->go_blob.ptr,
                       (PairRec*)ctx->pair_rec.ptr, (Job*)ctx->jobs.ptr, npairs, extra_cap, cnt, tsize_max, maxhap, cw);
    PLAT_HIP(ctx, hipGetLastError());
    return PLAT_OK;
}

PLAT_EXPORT int plat_align_window_batch(plat_ctx* ctx, const plat_window_batch* batch, int calc_flank_score,
                                        int use_mapq_cap, double* out_loglik, int32_t* out_score,
                                        plat_align_stats* out_stats, void* stream)
{
    if (!ctx || !batch) return PLAT_ERR_INVALID;
    if (calc_flank_score || use_mapq_cap) return PLAT_ERR_UNSUPPORTED;
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
    hipStream_t st = (hipStream_t)stream;
    PLAT_HIP(ctx, hipSetDevice(ctx->device));

    int rc = plat_reserve(ctx, ctx->counters, (CNT_N + 8) * sizeof(long long) + (size_t)b.n_haps * sizeof(int32_t));
    if (rc) return rc;
    long long* cnt = (long long*)ctx->counters.ptr;
    int32_t* hap_win = (int32_t*)(cnt + CNT_N + 8);
    ctx->ev_valid_align = 0;
    PLAT_EV(ctx, 0, st);
    PLAT_HIP(ctx, hipMemsetAsync(cnt, 0, (CNT_N + 8) * sizeof(long long), st));
    hipLaunchKernelGGL(k_validate, dim3(1024), dim3(256), 0, st, b, cnt);
    hipLaunchKernelGGL(k_hap_window, dim3((b.n_windows + 255) / 256), dim3(256), 0, st, b, hap_win);
    PLAT_HIP(ctx, hipGetLastError());
    // read back: error, maxima, blob length, number of pairs
    int64_t* hb = ctx->h_readback;
    PLAT_HIP(ctx, hipMemcpyAsync(hb, cnt, CNT_N * sizeof(long long), hipMemcpyDeviceToHost, st));
    PLAT_HIP(ctx, hipMemcpyAsync(hb + 16, b.hap_off + b.n_haps, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    PLAT_HIP(ctx, hipMemcpyAsync(hb + 17, b.pair_off + b.n_windows, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    PLAT_HIP(ctx, hipStreamSynchronize(st));
    if (hb[CNT_ERR] != 0) return (int)hb[CNT_ERR];
    const int maxhap = (int)hb[CNT_MAXHAP], maxread = (int)hb[CNT_MAXREAD];
    const long long hapblob = hb[16], npairs = hb[17];
    if (npairs == 0) return PLAT_OK;
    if ((rc = plat_reserve(ctx, ctx->go_blob, (size_t)hapblob + 64))) return rc;
    if ((rc = plat_reserve(ctx, ctx->pair_rec, (size_t)npairs * sizeof(PairRec)))) return rc;
    long long extra_cap = npairs / 4 + 4096;
    if ((long long)(ctx->jobs.cap / sizeof(Job)) - npairs > extra_cap) extra_cap = (long long)(ctx->jobs.cap / sizeof(Job)) - npairs;
    if (extra_cap > 0x7FFFFF00ll) extra_cap = 0x7FFFFF00ll;

    long long njobs = 0;
    PLAT_EV(ctx, 1, st);
    for (int attempt = 0; attempt < 2; ++attempt) {
        if ((rc = plat_reserve(ctx, ctx->jobs, (size_t)(npairs + extra_cap) * sizeof(Job)))) return rc;
        PLAT_HIP(ctx, hipMemsetAsync(&cnt[CNT_NEXTRA], 0, sizeof(long long), st));
        if ((rc = align_seed_launch(ctx, b, st, cnt, maxhap, maxread, npairs, (int)extra_cap, hap_win))) return rc;
        PLAT_HIP(ctx, hipMemcpyAsync(hb, cnt, CNT_N * sizeof(long long), hipMemcpyDeviceToHost, st));
        PLAT_HIP(ctx, hipStreamSynchronize(st));
        if (hb[CNT_ERR] != 0) return (int)hb[CNT_ERR];
        const long long nextra = hb[CNT_NEXTRA];
        njobs = npairs + nextra;
        if (nextra <= extra_cap) break;
        if (nextra > 0x7FFFFF00ll || attempt == 1) return PLAT_ERR_OVERFLOW;
        extra_cap = nextra;                    // tandem-rich batch: re-run the seeding with the exact capacity
    }
    if ((rc = plat_reserve(ctx, ctx->job_score, (size_t)(njobs + 1) * sizeof(int32_t)))) return rc;
    PLAT_EV(ctx, 2, st);
    if (njobs > 0) {
        hipLaunchKernelGGL(k_dp_jobs, dim3((unsigned)((njobs + 255) / 256)), dim3(256), 0, st, b,
                           (const uint8_t*)ctx->go_blob.ptr, (const Job*)ctx->jobs.ptr, njobs,
                           (int32_t*)ctx->job_score.ptr);
    }
    PLAT_EV(ctx, 3, st);
    if (njobs > 0 && (out_stats || ctx->profile))
        hipLaunchKernelGGL(k_sum_job_cells, dim3(256), dim3(256), 0, st, (const Job*)ctx->jobs.ptr, njobs, cnt);
    hipLaunchKernelGGL(k_finalize, dim3((unsigned)((npairs + 255) / 256)), dim3(256), 0, st, b,
                       (const PairRec*)ctx->pair_rec.ptr, (const Job*)ctx->jobs.ptr,
                       (const int32_t*)ctx->job_score.ptr, ctx->d_mapq_lut, npairs, out_loglik, out_score, cnt);
    PLAT_HIP(ctx, hipGetLastError());
    PLAT_EV(ctx, 4, st);
    if (out_stats || ctx->profile) {
        PLAT_HIP(ctx, hipMemcpyAsync(hb, cnt, CNT_N * sizeof(long long), hipMemcpyDeviceToHost, st));
        PLAT_HIP(ctx, hipStreamSynchronize(st));
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
    }
    return PLAT_OK;
}
