// plat_candidates.hip -- VariantCandidateGenerator (src/cython/variant.pyx:459-751): variant candidates
// from the CIGAR strings and the mismatches of the reads of a region.
//
// One lane walks one read exactly as getVariantCandidatesFromSingleRead / getSnpCandidatesFromReadSegment do (the
// mismatch-run state machine is sequential in the read), and writes one record per candidate OCCURRENCE into the read's
// own slice of the output -- so records come out in the reference's emission order (read by read, left to right) without
// any ordering step.  Merging equal variants and counting their supporting reads (addVariantToList, :499-527) is a
// dictionary operation over the few records that come back; it stays on the host (hostapi.VariantCandidateGenerator).
#include "plat_internal.hpp"
#include <mutex>
#include <math.h>

namespace plat {

struct CandEmit {
    int32_t* rec; int n, cap;
    __device__ __forceinline__ void put(int pos, int nrem, int nadd, long long rem_off, long long add_off) {
        if (n < cap) {
            int32_t* r = rec + 5 * (long long)n;
            r[0] = pos < 0 ? 0 : pos;                    // Variant.__init__: max(0, refPos), variant.pyx:118
            r[1] = nrem; r[2] = nadd; r[3] = (int32_t)rem_off; r[4] = (int32_t)add_off;
        }
        ++n;
    }
};

// 8 bytes from any address (global memory takes unaligned 64-bit loads)
__device__ __forceinline__ unsigned long long load_u64_bytes(const uint8_t* p) {
    typedef unsigned long long __attribute__((aligned(1))) u64u;
    return *(const u64u*)p;
}

__device__ __forceinline__ bool has_n(const uint8_t* s, int n) {
    for (int i = 0; i < n; ++i) if (s[i] == 'N') return true;
    return false;
}

__global__ void __launch_bounds__(64)
k_candidates(plat_candidate_batch b, int min_flank, int min_base_qual, int gen_snps, int gen_indels, int max_per_read,
             const int32_t* __restrict__ read_region, int32_t* __restrict__ rec, int32_t* __restrict__ count, int32_t* __restrict__ status)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= b.n_reads) return;
    int st = 0;
    CandEmit out{rec + 5ll * (long long)r * max_per_read, 0, max_per_read};
    if (!(b.read_flags[r] & 512)) {                      // Read_IsQCFail reads are skipped, variant.pyx:729-731
        const int g = read_region[r];
        const long long roff = b.ref_off[g];
        const int refLen = (int)(b.ref_off[g + 1] - roff), refSeqStart = b.ref_seq_start[g], contigLen = b.contig_len[g];
        const uint8_t* ref = b.ref_seq + roff;
        const long long soff = b.read_off[r];
        const uint8_t* readSeq = b.read_seq + soff;
        const uint8_t* readQual = b.read_qual + soff;
        const int rlen = (int)(b.read_off[r + 1] - soff);
        const int readStart = b.read_pos[r];
        const int16_t* ops = b.cigar + 2ll * b.cig_off[r];
        const int cigarLength = b.cig_off[r + 1] - b.cig_off[r];
        int refOffset = 0, readOffset = 0;
        for (int ci = 0; ci < cigarLength; ++ci) {       // getVariantCandidatesFromSingleRead, :614-720
            const int flag = ops[2 * ci], length = ops[2 * ci + 1];
            if (flag == 1 || flag == 2) {                // insertion / deletion: needs a match of >= minFlank on one side
                bool flanked = false;
                if (ci > 0 && ops[2 * ci - 2] == 0 && ops[2 * ci - 1] >= min_flank) flanked = true;
                else if (ci < cigarLength - 1 && ops[2 * ci + 2] == 0 && ops[2 * ci + 3] >= min_flank) flanked = true;
                if (flag == 1) {
                    if (flanked && gen_indels && !has_n(readSeq + readOffset, length))
                        out.put(readStart + refOffset - 1, 0, length, -1, soff + readOffset);
                    readOffset += length;
                } else {
                    if (flanked) {
                        // refFile.getSequence(rname, a, a + length): clamped to [0, contigLen - 1], fastafile.pyx:186-187
                        int a = readStart + refOffset, e = a + length;
                        if (a < 0) a = 0;
                        if (e > contigLen - 1) e = contigLen - 1;
                        const int n = e > a ? e - a : 0;
                        if (a - refSeqStart < 0 || a - refSeqStart + n > refLen) st = PLAT_ERR_BAD_INPUT;   // outside the window handed over
                        else if (gen_indels && !has_n(ref + (a - refSeqStart), n))
                            out.put(readStart + refOffset - 1, n, 0, roff + (a - refSeqStart), -1);
                    }
                    refOffset += length;
                }
            } else if (flag == 0 || flag == 7 || flag == 8) {    // M, =, X
                if (!(flag == 7 || (length < min_flank && flag == 0)) && gen_snps) {
                    // getSnpCandidatesFromReadSegment, :529-612.  The loop there visits every base; only mismatches change its state
                    // (a run of mismatches is closed at the first MATCH more than minFlank behind its end -- or by the next mismatch that
                    // far behind it, which finds the same run and writes the same record), so whole words of equal bases are stepped over:
                    // 8 bases of the read against 8 of the reference per load.
                    int msr = -1, mer = -1, msd = -1, med = -1;
                    int lo = (readOffset == 0) ? (min_flank < length ? min_flank : length) : 0;                  // first index looked at
                    int hi = rlen - min_flank - readOffset;                                                       // one past the last
                    if (hi > length) hi = length;
                    // the reference reads past its buffer where refIndex leaves [0, refLen): it stops there (after the bases before)
                    const int refIndex0 = refOffset + readStart - refSeqStart;                                    // refIndex of index 0
                    if (lo < hi) {
                        if (refIndex0 + lo < 0) { st = PLAT_ERR_BAD_INPUT; hi = lo; }
                        else if (refIndex0 + hi > refLen) { st = PLAT_ERR_BAD_INPUT; hi = refLen - refIndex0; }
                    }
                    const uint8_t* rp = readSeq + readOffset;
                    const uint8_t* fp = ref + refIndex0;
                    auto mismatch = [&](int index) {
                        const int readIndex = index + readOffset, refIndex = refIndex0 + index;
                        const uint8_t readChar = rp[index], refChar = fp[index];
                        if (readChar != 'N' && refChar != 'N' && (int)readQual[readIndex] >= min_base_qual) {
                            if (msr == -1) { msr = mer = refIndex; msd = med = readIndex; }
                            else if (refIndex - mer <= min_flank) { mer = refIndex; med = readIndex; }
                            else {
                                out.put(msr + refSeqStart, mer - msr + 1, med - msd + 1, roff + msr, soff + msd);
                                msr = mer = refIndex; msd = med = readIndex;
                            }
                        }
                    };
                    int index = lo;
                    // (round 5: 32 bases per trip -- the eight loads of a trip do not depend on one another, so a trip waits for memory once
                    //  where four trips of 8 bases waited four times: the walk is a chain of such waits, not a stream.  The last trip reads
                    //  past `hi` -- blobs are followed by PLAT_BLOB_PAD readable bytes -- and masks what it read there.)
                    constexpr int CAND_TRIP = 4;                                     // words of 8 bases per trip
                    static_assert(PLAT_BLOB_PAD >= 8 * CAND_TRIP, "the last trip of the mismatch scan reads past a blob's end");
                    for (; index < hi; index += 8 * CAND_TRIP) {
                        const int nleft = hi - index;
                        unsigned long long x4[CAND_TRIP];
#pragma unroll
                        for (int q = 0; q < CAND_TRIP; ++q) x4[q] = load_u64_bytes(rp + index + 8 * q) ^ load_u64_bytes(fp + index + 8 * q);
                        if (nleft < 8 * CAND_TRIP) {
#pragma unroll
                            for (int q = 0; q < CAND_TRIP; ++q) {
                                const int nb = nleft - 8 * q;
                                if (nb <= 0) x4[q] = 0ull;
                                else if (nb < 8) x4[q] &= (1ull << (8 * nb)) - 1ull;
                            }
                        }
                        unsigned long long any = 0ull;
#pragma unroll
                        for (int q = 0; q < CAND_TRIP; ++q) any |= x4[q];
                        if (any == 0ull) continue;
#pragma unroll
                        for (int q = 0; q < CAND_TRIP; ++q) {
                            unsigned long long x = x4[q];
                            while (x) {
                                const int j = (__ffsll((long long)x) - 1) >> 3;
                                mismatch(index + 8 * q + j);
                                x &= ~(0xFFull << (8 * j));
                            }
                        }
                    }
                    if (msr != -1) out.put(msr + refSeqStart, mer - msr + 1, med - msd + 1, roff + msr, soff + msd);
                }
                readOffset += length;
                refOffset += length;
            } else if (flag == 3) {                      // N: skipped reference
                refOffset += length;
            } else if (flag == 4) {                      // soft clip: bases present in the read, positions were moved back
                readOffset += length;
                if (ci == 0) refOffset += length;
            }                                            // H, P, anything else: nothing
        }
    }
    if (out.n > max_per_read && st == 0) st = PLAT_ERR_OVERFLOW;
    count[r] = out.n;
    status[r] = st;
}

// ---- the scan on 2-bit base codes (round 6) --------------------------------------------------------------------------------------------
// k_candidates walks a read 8 bases per 64-bit word, 40 scattered loads for 150 bases and the same again for the reference: 1.07 GB of traffic
// per chunk of 128 regions against the 0.42 GB it has to read.  Equal bases have equal codes, so the mismatch scan can run on the codes -- 32 bases
// per word, one aligned load each for read and reference, the whole read in ONE round of loads -- and look at the bytes only where the codes
// differ: a position whose codes differ is a mismatch of the bytes or holds a byte that is not A, C, G, T (then the byte test of the
// reference's loop decides, as before); a position whose codes are equal while its bytes differ can only hold such a byte -- an N (which the loop
// ignores anyway: variant.pyx:560-567) or, in an "irregular" reference region (any other byte), anything: those regions take the byte scan.
__global__ void __launch_bounds__(256)
k_ref_codes(const uint8_t* __restrict__ ref, const int64_t* __restrict__ ref_off, int n_regions, long long n_bytes, uint32_t* __restrict__ codes,
            int32_t* __restrict__ irregular)
{
    const long long d = (long long)blockIdx.x * blockDim.x + threadIdx.x;        // one dword of codes = 16 bases
    if (16 * d >= n_bytes) return;
    uint32_t cw = 0;
    bool odd = false;
    for (int k = 0; k < 16; ++k) {
        const long long i = 16 * d + k;
        const unsigned b = i < n_bytes ? ref[i] : (unsigned)'A';
        cw |= ((b >> 1) & 3u) << (2 * k);
        odd = odd || !(b == 'A' || b == 'C' || b == 'G' || b == 'T' || b == 'N');
    }
    codes[d] = cw;
    if (odd)
        for (int k = 0; k < 16; ++k) {
            const long long i = 16 * d + k;
            if (i >= n_bytes) break;
            const unsigned b = ref[i];
            if (b == 'A' || b == 'C' || b == 'G' || b == 'T' || b == 'N') continue;
            int lo = 0, hi = n_regions;                                           // the region this byte belongs to: ref_off[g] <= i < ref_off[g + 1]
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (ref_off[mid] <= i) lo = mid; else hi = mid; }
            irregular[lo] = 1;
        }
}

// 32 bases of a code stream from base index i (any alignment): two aligned words and a funnel shift
__device__ __forceinline__ unsigned long long codes_at(const unsigned long long* __restrict__ c, long long i) {
    const long long w = i >> 5;
    const int sh = 2 * (int)(i & 31);
    const unsigned long long lo = c[w];
    if (sh == 0) return lo;
    return (lo >> sh) | (c[w + 1] << (64 - sh));
}

__global__ void __launch_bounds__(64)
k_candidates_codes(plat_candidate_batch b, const unsigned long long* __restrict__ read_codes, const unsigned long long* __restrict__ ref_codes,
                   const int32_t* __restrict__ ref_irregular, int min_flank, int min_base_qual, int gen_snps, int gen_indels, int max_per_read,
                   const int32_t* __restrict__ read_region, int32_t* __restrict__ rec, int32_t* __restrict__ count, int32_t* __restrict__ status)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= b.n_reads) return;
    int st = 0;
    CandEmit out{rec + 5ll * (long long)r * max_per_read, 0, max_per_read};
    if (!(b.read_flags[r] & 512)) {                      // Read_IsQCFail reads are skipped, variant.pyx:729-731
        const int g = read_region[r];
        const long long roff = b.ref_off[g];
        const int refLen = (int)(b.ref_off[g + 1] - roff), refSeqStart = b.ref_seq_start[g], contigLen = b.contig_len[g];
        const bool byBytes = ref_irregular[g] != 0;      // a reference byte other than A, C, G, T, N in this region: the byte scan
        const uint8_t* ref = b.ref_seq + roff;
        const long long soff = b.read_off[r];
        const uint8_t* readSeq = b.read_seq + soff;
        const uint8_t* readQual = b.read_qual + soff;
        const int rlen = (int)(b.read_off[r + 1] - soff);
        const int readStart = b.read_pos[r];
        const int16_t* ops = b.cigar + 2ll * b.cig_off[r];
        const int cigarLength = b.cig_off[r + 1] - b.cig_off[r];
        int refOffset = 0, readOffset = 0;
        for (int ci = 0; ci < cigarLength; ++ci) {       // getVariantCandidatesFromSingleRead, :614-720
            const int flag = ops[2 * ci], length = ops[2 * ci + 1];
            if (flag == 1 || flag == 2) {                // insertion / deletion: needs a match of >= minFlank on one side
                bool flanked = false;
                if (ci > 0 && ops[2 * ci - 2] == 0 && ops[2 * ci - 1] >= min_flank) flanked = true;
                else if (ci < cigarLength - 1 && ops[2 * ci + 2] == 0 && ops[2 * ci + 3] >= min_flank) flanked = true;
                if (flag == 1) {
                    if (flanked && gen_indels && !has_n(readSeq + readOffset, length))
                        out.put(readStart + refOffset - 1, 0, length, -1, soff + readOffset);
                    readOffset += length;
                } else {
                    if (flanked) {
                        int a = readStart + refOffset, e = a + length;   // refFile.getSequence(rname, a, a + length): clamped to [0, contigLen - 1], fastafile.pyx:186-187
                        if (a < 0) a = 0;
                        if (e > contigLen - 1) e = contigLen - 1;
                        const int n = e > a ? e - a : 0;
                        if (a - refSeqStart < 0 || a - refSeqStart + n > refLen) st = PLAT_ERR_BAD_INPUT;   // outside the window handed over
                        else if (gen_indels && !has_n(ref + (a - refSeqStart), n))
                            out.put(readStart + refOffset - 1, n, 0, roff + (a - refSeqStart), -1);
                    }
                    refOffset += length;
                }
            } else if (flag == 0 || flag == 7 || flag == 8) {    // M, =, X
                if (!(flag == 7 || (length < min_flank && flag == 0)) && gen_snps) {
                    // getSnpCandidatesFromReadSegment, :529-612: only mismatches change the loop's state (k_candidates has the argument)
                    int msr = -1, mer = -1, msd = -1, med = -1;
                    int lo = (readOffset == 0) ? (min_flank < length ? min_flank : length) : 0;                  // first index looked at
                    int hi = rlen - min_flank - readOffset;                                                       // one past the last
                    if (hi > length) hi = length;
                    const int refIndex0 = refOffset + readStart - refSeqStart;                                    // refIndex of index 0
                    if (lo < hi) {
                        if (refIndex0 + lo < 0) { st = PLAT_ERR_BAD_INPUT; hi = lo; }
                        else if (refIndex0 + hi > refLen) { st = PLAT_ERR_BAD_INPUT; hi = refLen - refIndex0; }
                    }
                    const uint8_t* rp = readSeq + readOffset;
                    const uint8_t* fp = ref + refIndex0;
                    auto mismatch = [&](int index) {
                        const int readIndex = index + readOffset, refIndex = refIndex0 + index;
                        const uint8_t readChar = rp[index], refChar = fp[index];
                        if (readChar != refChar && readChar != 'N' && refChar != 'N' && (int)readQual[readIndex] >= min_base_qual) {
                            if (msr == -1) { msr = mer = refIndex; msd = med = readIndex; }
                            else if (refIndex - mer <= min_flank) { mer = refIndex; med = readIndex; }
                            else {
                                out.put(msr + refSeqStart, mer - msr + 1, med - msd + 1, roff + msr, soff + msd);
                                msr = mer = refIndex; msd = med = readIndex;
                            }
                        }
                    };
                    if (byBytes) {
                        for (int index = lo; index < hi; ++index) if (rp[index] != fp[index]) mismatch(index);
                    } else {
                        // 32 bases per word; CODE_TRIP words (160 bases: a whole 150-base read) are requested together
                        constexpr int CODE_TRIP = 5;
                        const long long rb = soff + readOffset, fb = roff + refIndex0;                           // blob index of index 0, read and reference
                        for (int index = lo; index < hi; index += 32 * CODE_TRIP) {
                            unsigned long long x[CODE_TRIP];
#pragma unroll
                            for (int q = 0; q < CODE_TRIP; ++q) {
                                const int at = index + 32 * q;
                                x[q] = at < hi ? codes_at(read_codes, rb + at) ^ codes_at(ref_codes, fb + at) : 0ull;
                            }
#pragma unroll
                            for (int q = 0; q < CODE_TRIP; ++q) {
                                const int at = index + 32 * q, nb = hi - at;
                                if (nb <= 0) continue;
                                unsigned long long y = x[q];
                                if (nb < 32) y &= (1ull << (2 * nb)) - 1ull;
                                y = (y | (y >> 1)) & 0x5555555555555555ull;                                         // one bit per base whose codes differ
                                while (y) {
                                    const int j = (__ffsll((long long)y) - 1) >> 1;
                                    mismatch(at + j);
                                    y &= y - 1ull;
                                }
                            }
                        }
                    }
                    if (msr != -1) out.put(msr + refSeqStart, mer - msr + 1, med - msd + 1, roff + msr, soff + msd);
                }
                readOffset += length;
                refOffset += length;
            } else if (flag == 3) {                      // N: skipped reference
                refOffset += length;
            } else if (flag == 4) {                      // soft clip: bases present in the read, positions were moved back
                readOffset += length;
                if (ci == 0) refOffset += length;
            }                                            // H, P, anything else: nothing
        }
    }
    if (out.n > max_per_read && st == 0) st = PLAT_ERR_OVERFLOW;
    count[r] = out.n;
    status[r] = st;
}

}  // namespace plat

PLAT_EXPORT int plat_ref_codes(plat_ctx* ctx, int n_regions, const uint8_t* ref_seq, const int64_t* ref_off, int64_t n_bytes, uint32_t* out_codes,
                               int32_t* out_irregular, void* stream)
{
    if (!ctx || n_regions < 0 || n_bytes < 0) return PLAT_ERR_INVALID;
    if (n_regions == 0 || n_bytes == 0) return PLAT_OK;
    if (!ref_seq || !ref_off || !out_codes || !out_irregular) return PLAT_ERR_INVALID;
    PLAT_HIP(ctx, hipSetDevice(ctx->device));
    PLAT_HIP(ctx, hipMemsetAsync(out_irregular, 0, (size_t)n_regions * sizeof(int32_t), (hipStream_t)stream));
    const long long nd = (n_bytes + 15) / 16;
    PLAT_HIP(ctx, hipMemsetAsync(out_codes + nd, 0, 8 * sizeof(uint32_t), (hipStream_t)stream));      // (the scan reads a 64-bit word past the last base)
    hipLaunchKernelGGL(plat::k_ref_codes, dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ref_seq, ref_off, n_regions, (long long)n_bytes,
                       out_codes, out_irregular);
    PLAT_HIP(ctx, hipGetLastError());
    return PLAT_OK;
}

PLAT_EXPORT int plat_candidates_batch_codes(plat_ctx* ctx, const plat_candidate_batch* batch, const uint32_t* read_codes, const uint32_t* ref_codes,
                                            const int32_t* ref_irregular, int min_flank, int min_base_qual, int gen_snps, int gen_indels, int max_per_read,
                                            const int32_t* read_region, int32_t* out_rec, int32_t* out_count, int32_t* out_status, void* stream)
{
    if (!ctx || !batch || max_per_read < 1 || min_flank < 0 || !read_codes || !ref_codes || !ref_irregular) return PLAT_ERR_INVALID;
    if (((uintptr_t)read_codes & 7) || ((uintptr_t)ref_codes & 7)) return PLAT_ERR_INVALID;             // (read as 64-bit words)
    const plat_candidate_batch b = *batch;
    if (b.n_regions < 0 || b.n_reads < 0) return PLAT_ERR_INVALID;
    if (b.n_reads == 0) return PLAT_OK;
    if (!b.ref_seq || !b.ref_off || !b.ref_seq_start || !b.contig_len || !b.read_seq || !b.read_qual || !b.read_off ||
        !b.read_pos || !b.read_flags || !b.cigar || !b.cig_off || !read_region || !out_rec || !out_count || !out_status)
        return PLAT_ERR_INVALID;
    PLAT_HIP(ctx, hipSetDevice(ctx->device));
    PLAT_EV_TAB(ctx, 2, (hipStream_t)stream);
    { PLAT_KT_BEGIN(ctx, PLAT_KT_CANDIDATES, (hipStream_t)stream);
      hipLaunchKernelGGL(plat::k_candidates_codes, dim3((unsigned)((b.n_reads + 63) / 64)), dim3(64), 0, (hipStream_t)stream, b, (const unsigned long long*)read_codes,
                         (const unsigned long long*)ref_codes, ref_irregular, min_flank, min_base_qual, gen_snps, gen_indels, max_per_read, read_region, out_rec, out_count,
                         out_status);
      PLAT_KT_END(ctx, PLAT_KT_CANDIDATES, (hipStream_t)stream); }
    PLAT_EV_TAB(ctx, 3, (hipStream_t)stream);
    ctx->ev_valid_cand = ctx->profile;
    PLAT_HIP(ctx, hipGetLastError());
    return PLAT_OK;
}

namespace plat {
}  // namespace plat

PLAT_EXPORT int plat_candidates_batch(plat_ctx* ctx, const plat_candidate_batch* batch, int min_flank, int min_base_qual,
                                      int gen_snps, int gen_indels, int max_per_read, const int32_t* read_region,
                                      int32_t* out_rec, int32_t* out_count, int32_t* out_status, void* stream)
{
    if (!ctx || !batch || max_per_read < 1 || min_flank < 0) return PLAT_ERR_INVALID;
    const plat_candidate_batch b = *batch;
    if (b.n_regions < 0 || b.n_reads < 0) return PLAT_ERR_INVALID;
    if (b.n_reads == 0) return PLAT_OK;
    if (!b.ref_seq || !b.ref_off || !b.ref_seq_start || !b.contig_len || !b.read_seq || !b.read_qual || !b.read_off ||
        !b.read_pos || !b.read_flags || !b.cigar || !b.cig_off || !read_region || !out_rec || !out_count || !out_status)
        return PLAT_ERR_INVALID;
    PLAT_HIP(ctx, hipSetDevice(ctx->device));
    PLAT_EV_TAB(ctx, 2, (hipStream_t)stream);
    { PLAT_KT_BEGIN(ctx, PLAT_KT_CANDIDATES, (hipStream_t)stream); hipLaunchKernelGGL(plat::k_candidates, dim3((unsigned)((b.n_reads + 63) / 64)), dim3(64), 0, (hipStream_t)stream, b,
                       min_flank, min_base_qual, gen_snps, gen_indels, max_per_read, read_region, out_rec, out_count, out_status); PLAT_KT_END(ctx, PLAT_KT_CANDIDATES, (hipStream_t)stream); }
    PLAT_EV_TAB(ctx, 3, (hipStream_t)stream);
    ctx->ev_valid_cand = ctx->profile;
    PLAT_HIP(ctx, hipGetLastError());
    return PLAT_OK;
}

// ------------------------------------------------------------------------------------------------
// The dictionary step behind the scan (VariantCandidateGenerator.addVariantToList, variant.pyx:499-527) and the per-sample support
// filter of generateVariantsInRegion (variantcaller.pyx:456-467) for every scan (= region x sample) of a candidate batch.
// k_candidates_merge: ONE THREAD PER READ puts the read's records into the scan's hash table in global memory (atomics in L2) -- slot =
// the record with the smallest id among those of equal content (first occurrence: the reference's dictionary order) + the number of reads
// showing it.  (Round 2 kept the table in LDS, one workgroup per scan: twenty reads per thread one after the other, each a chain of
// dependent loads -- 140 us for four scans of 20 000 reads, the longest kernel of the region pipeline.)
// k_candidates_filter: per distinct record, the reads covering its position (ReadArray.countReadsCoveringRegion, cwindow.pyx:176-206)
// and the filter.
namespace plat {
constexpr int MERGE_SLOTS = 8192, MERGE_LIMIT = 6144, MERGE_PROBES = 1024;
// per scan: rep + 1 [MERGE_SLOTS] (0 = empty) | count [MERGE_SLOTS]; after the tables of all scans, 4 words per scan: distinct, status, need, -
__device__ __forceinline__ int32_t* merge_flags(int32_t* mtab, int n_scans, int g) { return mtab + (size_t)n_scans * 2 * MERGE_SLOTS + 4 * (size_t)g; }

__device__ __forceinline__ bool rec_same(const plat_candidate_batch& b, const int32_t* x, const int32_t* y) {
    if (x[0] != y[0] || x[1] != y[1] || x[2] != y[2]) return false;
    for (int i = 0; i < x[1]; ++i) if (b.ref_seq[(long long)x[3] + i] != b.ref_seq[(long long)y[3] + i]) return false;
    for (int i = 0; i < x[2]; ++i) if (b.read_seq[(long long)x[4] + i] != b.read_seq[(long long)y[4] + i]) return false;
    return true;
}

__global__ void __launch_bounds__(256)
k_candidates_merge(plat_candidate_batch b, const int32_t* __restrict__ scan_read_begin, int n_scans, int max_per_read,
                   const int32_t* __restrict__ rec, const int32_t* __restrict__ count, const int32_t* __restrict__ status,
                   int32_t* __restrict__ mtab, int32_t* __restrict__ out_n)
{
    const int g = blockIdx.y;
    const int r0 = scan_read_begin[g], N = scan_read_begin[g + 1] - r0;
    int32_t* tab = mtab + (size_t)g * 2 * MERGE_SLOTS;
    int32_t* flags = merge_flags(mtab, n_scans, g);
    if (blockIdx.x == 0 && threadIdx.x == 0) { out_n[2 * g] = 0; out_n[2 * g + 1] = 0; }     // (the filter kernel counts into them)
    // distinct records are counted per workgroup and added once (most records of a scan are sequencing errors seen once: an atomic per
    // new record on one word of L2 was most of this kernel); a table filling up meanwhile shows as a probe sequence that does not end
    __shared__ int s_new;
    if (threadIdx.x == 0) s_new = 0;
    __syncthreads();
    const bool full = flags[0] > MERGE_LIMIT;
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < N && !full; q += gridDim.x * blockDim.x) {
        const int r = r0 + q, c = count[r];
        if (status[r] == PLAT_ERR_BAD_INPUT) flags[1] = PLAT_ERR_BAD_INPUT;
        if (c > max_per_read) { atomicMax(&flags[2], c); continue; }
        for (int k = 0; k < c; ++k) {
            const int id = r * max_per_read + k;
            const int32_t* me = rec + 5ll * id;
            unsigned h = (unsigned)me[0] * 2654435761u + (unsigned)me[1] * 40503u + (unsigned)me[2] * 97u;
            for (int i = 0; i < me[1]; ++i) h = h * 31u + b.ref_seq[(long long)me[3] + i];
            for (int i = 0; i < me[2]; ++i) h = h * 37u + b.read_seq[(long long)me[4] + i];
            unsigned sl = (h ^ (h >> 15)) & (MERGE_SLOTS - 1);
            bool placed = false;
            for (int tries = 0; tries < MERGE_PROBES && !placed; ++tries) {
                int cur = *(volatile int32_t*)&tab[sl];
                if (cur == 0) {
                    const int old = atomicCAS(&tab[sl], 0, id + 1);
                    if (old == 0) { atomicAdd(&s_new, 1); atomicAdd(&tab[MERGE_SLOTS + sl], 1); placed = true; break; }
                    cur = old;
                }
                if (cur - 1 == id || rec_same(b, rec + 5ll * (cur - 1), me)) { atomicMin(&tab[sl], id + 1); atomicAdd(&tab[MERGE_SLOTS + sl], 1); placed = true; break; }
                sl = (sl + 1u) & (MERGE_SLOTS - 1);
            }
            if (!placed) atomicMax(&flags[0], MERGE_LIMIT + 1);          // the table is (nearly) full
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_new) atomicAdd(&flags[0], s_new);
}

// per distinct record of a scan: the reads covering its position (ReadArray.countReadsCoveringRegion, cwindow.pyx:176-206) and the
// per-sample support filter of generateVariantsInRegion (variantcaller.pyx:456-467).  grid = (MERGE_SLOTS / 256, scans).
__global__ void __launch_bounds__(256)
k_candidates_filter(plat_candidate_batch b, const int32_t* __restrict__ read_end, const int32_t* __restrict__ scan_read_begin,
                    const int32_t* __restrict__ scan_longest, int n_scans, const int32_t* __restrict__ rec, int32_t* __restrict__ mtab, double min_var_freq,
                    int cap, int32_t* __restrict__ out_cand, int32_t* __restrict__ out_n)
{
    const int g = blockIdx.y, sl = blockIdx.x * blockDim.x + threadIdx.x;
    const int32_t* flags = merge_flags(mtab, n_scans, g);
    const int distinct = flags[0], f_status = flags[1], f_need = flags[2];
    if (f_need > 0 || f_status != 0 || distinct > MERGE_LIMIT) {
        // -(2^20 + needed records per read) | overflow of the table | a read the scan refused
        if (sl == 0) out_n[2 * g + 1] = f_status != 0 ? f_status : (f_need > 0 ? -(1 << 20) - f_need : PLAT_ERR_OVERFLOW);
        return;
    }
    const int id = mtab[(size_t)g * 2 * MERGE_SLOTS + sl] - 1;
    if (id < 0) return;
    const int c = mtab[(size_t)g * 2 * MERGE_SLOTS + MERGE_SLOTS + sl];
    const int r0 = scan_read_begin[g], N = scan_read_begin[g + 1] - r0;
    const int32_t* pos = b.read_pos + r0;
    const int32_t* endp = read_end + r0;
    const int longest = scan_longest[g];
    const int32_t* me = rec + 5ll * id;
    const int start = me[0];
    // countReadsCoveringRegion(start, start + 1), cwindow.pyx:176-206
    int total = 0;
    if (N > 0) {
        const long long key = (long long)start - longest > 1 ? (long long)start - longest : 1;
        int lo = 0, hi = N;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if ((long long)pos[mid] < key) lo = mid + 1; else hi = mid; }
        int s = lo;
        lo = 0; hi = N;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (pos[mid] < start + 1) lo = mid + 1; else hi = mid; }
        const int e = lo;
        while (s < N && endp[s] <= start) ++s;
        if (s > e) { atomicCAS(&out_n[2 * g + 1], 0, PLAT_ERR_BAD_INPUT); return; }    // "Read start pointer > read end pointer": the reference raises
        total = e - s;
    }
    const double frac = total == 0 ? 0.0 : (double)c / (double)total;
    if (frac >= min_var_freq || me[1] != me[2]) {
        const int at = atomicAdd(&out_n[2 * g], 1);
        if (at < cap) {
            int32_t* o = out_cand + 8ll * ((long long)g * cap + at);
            o[0] = id; o[1] = c; o[2] = total; o[3] = me[0]; o[4] = me[1]; o[5] = me[2]; o[6] = me[3]; o[7] = me[4];
        } else atomicCAS(&out_n[2 * g + 1], 0, PLAT_ERR_OVERFLOW);                      // more candidates than the caller's room: its status says so
    }
}
}  // namespace plat

PLAT_EXPORT int plat_candidates_merge_batch(plat_ctx* ctx, const plat_candidate_batch* batch, const int32_t* read_end, int n_scans,
                                            const int32_t* scan_read_begin, const int32_t* scan_longest, int max_per_read,
                                            const int32_t* rec, const int32_t* count, const int32_t* status, double min_var_freq,
                                            int cap_per_scan, int32_t* out_cand, int32_t* out_n, void* stream)
{
    if (!ctx || !batch || n_scans < 0 || max_per_read < 1 || cap_per_scan < 1) return PLAT_ERR_INVALID;
    if (n_scans == 0) return PLAT_OK;
    if (n_scans > PLAT_GRID_Y_MAX) return PLAT_ERR_OVERFLOW;       // (one scan = one region of a chunk; the merge table is per scan)
    const plat_candidate_batch b = *batch;
    if (!b.ref_seq || !b.read_seq || !b.read_pos || !read_end || !scan_read_begin || !scan_longest || !rec || !count || !status || !out_cand || !out_n)
        return PLAT_ERR_INVALID;
    PLAT_HIP(ctx, hipSetDevice(ctx->device));
    const size_t tab_bytes = ((size_t)n_scans * 2 * plat::MERGE_SLOTS + 4 * (size_t)n_scans) * sizeof(int32_t);
    int rcm = plat_reserve(ctx, ctx->merge_tab, tab_bytes);
    if (rcm) return rcm;
    int32_t* mtab = (int32_t*)ctx->merge_tab.ptr;
    PLAT_HIP(ctx, hipMemsetAsync(mtab, 0, tab_bytes, (hipStream_t)stream));
    // a thread per read when the scans are of one size; a scan with more than its share is walked in strides
    long long per = ((long long)b.n_reads + n_scans - 1) / n_scans;
    unsigned gx = (unsigned)((per + 255) / 256);
    gx = gx < 1 ? 1 : (gx > 4096 ? 4096 : gx);
    { PLAT_KT_BEGIN(ctx, PLAT_KT_CAND_MERGE, (hipStream_t)stream); hipLaunchKernelGGL(plat::k_candidates_merge, dim3(gx, (unsigned)n_scans), dim3(256), 0, (hipStream_t)stream, b,
                       scan_read_begin, n_scans, max_per_read, rec, count, status, mtab, out_n); PLAT_KT_END(ctx, PLAT_KT_CAND_MERGE, (hipStream_t)stream); }
    { PLAT_KT_BEGIN(ctx, PLAT_KT_CAND_FILTER, (hipStream_t)stream); hipLaunchKernelGGL(plat::k_candidates_filter, dim3(plat::MERGE_SLOTS / 256, (unsigned)n_scans), dim3(256), 0, (hipStream_t)stream, b, read_end,
                       scan_read_begin, scan_longest, n_scans, rec, mtab, min_var_freq, cap_per_scan, out_cand, out_n); PLAT_KT_END(ctx, PLAT_KT_CAND_FILTER, (hipStream_t)stream); }
    PLAT_HIP(ctx, hipGetLastError());
    return PLAT_OK;
}

// ------------------------------------------------------------------------------------------------
// Read QC / trimming: checkAndTrimRead (cwindow.pyx:332-481).  One lane per read; `theLastRead` of
// bamReadBuffer.addReadToBuffer (:560-595) is simply the previous read of the same stream and only its position, length
// and mate position are looked at, so the reads of a stream are independent of each other.
namespace plat {

__global__ void __launch_bounds__(64)
k_read_qc(plat_readqc_batch b, plat_readqc_options o, int32_t* __restrict__ ok, int32_t* __restrict__ reason)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= b.n_reads) return;
    int8_t* q = (int8_t*)b.read_qual + b.read_off[r];
    const int rlen = (int)(b.read_off[r + 1] - b.read_off[r]);
    const int f = b.read_flags[r];
    const bool paired = f & 1, proper = f & 2, unmapped = f & 4, mateUnmapped = f & 8, reverse = f & 16, mateReverse = f & 32;
    const int ins = b.insert_size[r];
    const int absIns = ins < 0 ? -ins : ins;
    int why = -1, qcfail = 1;
    if (f & 256) why = 7;                                                            // secondary alignment, :337-339
    else if ((int)b.read_mapq[r] < o.min_map_qual) why = 6;                          // :341-344
    else {
        int nBelow = 0;
        for (int i = 0; i < rlen; ++i) nBelow += q[i] < o.min_base_qual;
        if (rlen - nBelow < o.min_good_qual_bases) why = 0;                          // :354-357
        else if (unmapped) why = 1;                                                  // :360-363
        else if (o.filter_mate_unmapped && paired && mateUnmapped) { why = 2; qcfail = 0; }          // :367-371 (no QCFail flag)
        else if (o.filter_mate_distant && paired && (b.chrom_id[r] != b.mate_chrom_id[r] || !proper)) { why = 3; qcfail = 0; }
        else if (o.filter_small_insert && paired && ins != 0 && absIns < rlen) why = 4;
        else if (o.filter_duplicates) {                                              // :389-409
            if (f & 1024) why = 5;
            else if (r > 0 && b.stream_of[r - 1] == b.stream_of[r] && b.read_pos[r] == b.read_pos[r - 1] &&
                     rlen == (int)(b.read_off[r] - b.read_off[r - 1])) {
                if (!paired || b.mate_pos[r - 1] == b.mate_pos[r]) why = 5;
            }
        }
    }
    if (why >= 0) {
        if (qcfail) b.read_flags[r] = f | 512;
        ok[r] = 0; reason[r] = why;
        return;
    }
    if (!reverse) {                                                                  // low-quality tail, :415-421
        for (int i = 1; i <= rlen; ++i) {
            if (i < o.trim_read_flank || q[rlen - i] < 5) q[rlen - i] = 0; else break;
        }
    } else {
        for (int i = 0; i < rlen; ++i) {
            if (i < o.trim_read_flank || q[i] < 5) q[i] = 0; else break;
        }
    }
    if (o.trim_overlapping == 1 && paired && absIns > 0 && !reverse && mateReverse && absIns < 2 * rlen) {   // :438-440
        int lim = (2 * rlen - ins) + 1;
        if (lim > rlen) lim = rlen;
        for (int i = 1; i <= lim; ++i) q[rlen - i] = 0;
    }
    if (o.trim_adapter == 1 && paired && absIns > 0 && absIns < rlen) {               // :445-452
        if (reverse) { for (int i = 1; i < rlen - absIns + 1; ++i) q[rlen - i] = 0; }
        else { for (int i = absIns; i < rlen; ++i) q[i] = 0; }
    }
    if (o.trim_soft_clipped == 1) {                                                   // :462-479 (only M and I advance the cursor)
        int index = 0;
        for (int c = b.cig_off[r]; c < b.cig_off[r + 1]; ++c) {
            const int op = b.cigar[2 * c], len = b.cigar[2 * c + 1];
            if (op == 0 || op == 1) index += len;
            else if (op == 4) for (int j = 0; j < len && index < rlen; ++j) q[index++] = 0;
        }
    }
    ok[r] = 1; reason[r] = -1;
}

}  // namespace plat

PLAT_EXPORT int plat_read_qc_batch(plat_ctx* ctx, const plat_readqc_batch* batch, const plat_readqc_options* options,
                                   int32_t* out_ok, int32_t* out_reason, void* stream)
{
    if (!ctx || !batch || !options) return PLAT_ERR_INVALID;
    const plat_readqc_batch b = *batch;
    if (b.n_reads < 0) return PLAT_ERR_INVALID;
    if (b.n_reads == 0) return PLAT_OK;
    if (!b.read_qual || !b.read_off || !b.read_pos || !b.read_mapq || !b.read_flags || !b.chrom_id || !b.mate_chrom_id ||
        !b.insert_size || !b.mate_pos || !b.cigar || !b.cig_off || !b.stream_of || !out_ok || !out_reason)
        return PLAT_ERR_INVALID;
    PLAT_HIP(ctx, hipSetDevice(ctx->device));
    { PLAT_KT_BEGIN(ctx, PLAT_KT_READ_QC, (hipStream_t)stream); hipLaunchKernelGGL(plat::k_read_qc, dim3((unsigned)((b.n_reads + 63) / 64)), dim3(64), 0, (hipStream_t)stream, b, *options,
                       out_ok, out_reason); PLAT_KT_END(ctx, PLAT_KT_READ_QC, (hipStream_t)stream); }
    PLAT_HIP(ctx, hipGetLastError());
    return PLAT_OK;
}

// ------------------------------------------------------------------------------------------------
// Read statistics of the VCF INFO field: vcfINFO's loop over the reads of a window
// (vcfutils.pyx:1300-1390) with readOverlapsVariant (:901-913), readQualIsGoodVariantPosition (:917-943) and
// variantSupportedByRead (:961-1072).  One wave per variant; lanes over the reads of one sample at a time; every count is a
// ballot popcount, the MMLQ window minima are written in read order (prefix popcount).
namespace plat {

__device__ __forceinline__ bool qual_good_at(const int8_t* q, int rlen, int readPos, int vmin, int vmax) {
    int a = max(0, min(rlen, vmin - readPos)), e = max(0, min(rlen, vmax - readPos));
    for (int i = a; i < e; ++i) if (q[i] < 5) return false;
    return true;
}

__device__ __forceinline__ bool bytes_equal(const uint8_t* a, const uint8_t* b, int n) {
    for (int i = 0; i < n; ++i) if (a[i] != b[i]) return false;
    return true;
}

__device__ __forceinline__ bool read_supports(const uint8_t* seq, int rlen, int readStart, const int16_t* ops, int ncig, int varPos,
                                              int nAdded, int nRemoved, const uint8_t* added, int exact)
{
    int refOffset = 0, readOffset = 0;
    for (int ci = 0; ci < ncig; ++ci) {
        const int flag = ops[2 * ci], length = ops[2 * ci + 1];
        if (flag == 1) {                                                     // insertion, :984-1003
            if (nAdded != nRemoved) {
                if (!exact) return true;
                if (nAdded - nRemoved == length && readOffset + nAdded <= rlen && bytes_equal(seq + readOffset, added, nAdded)) return true;
                return false;
            }
            readOffset += length;
        } else if (flag == 2) {                                              // deletion, :1005-1024
            if (nAdded != nRemoved) return exact ? (nRemoved - nAdded == length) : true;
            refOffset += length;
        } else if (flag == 0 || flag == 7 || flag == 8) {                    // M, =, X, :1027-1048
            const int start = varPos - readStart + readOffset - refOffset;
            if (refOffset + readStart <= varPos && refOffset + readStart + length > varPos && nAdded == nRemoved &&
                start >= 0 && start + nAdded <= rlen && bytes_equal(seq + start, added, nAdded))
                return true;
            readOffset += length;
            refOffset += length;
        } else if (flag == 3) {                                              // N: the reference advances both, :1051-1053
            readOffset += length;
            refOffset += length;
        } else if (flag == 4) {
            readOffset += length;
            if (ci == 0) refOffset += length;
        }
    }
    return false;
}

__global__ void __launch_bounds__(64)
k_variant_read_stats(plat_infostats_batch b, int bad_reads_window, int exact, int64_t* __restrict__ out, int32_t* __restrict__ per_sample,
                     int32_t* __restrict__ minq, int32_t* __restrict__ nminq)
{
    const int v = blockIdx.x, lane = threadIdx.x;
    const int w = b.var_window[v];
    const int vmin = b.var_bam_min[v], vmax = b.var_bam_max[v], varPos = b.var_pos[v];
    const int nAdded = b.var_n_added[v], nRemoved = b.var_n_removed[v];
    const uint8_t* added = b.var_added + b.var_added_off[v];
    int32_t* mq = minq + b.minq_off[v];
    long long c[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) c[k] = 0;
    int nmq = 0;
    const int half = (bad_reads_window - 1) / 2;
    for (int i = 0; i < b.n_ind; ++i) {
        const long long seg = (long long)w * b.n_ind + i;
        const bool inGt = b.var_in_genotype[(long long)v * b.n_ind + i] != 0;
        const int gb = b.good_begin[seg], ge = b.good_end[seg], bb = b.bad_begin[seg], be = b.bad_end[seg];
        c[13] += ge - gb; c[14] += be - bb;
        for (int r0 = bb; r0 < be; r0 += 64) {                               // bad reads: coverage and mapping quality only
            const int r = r0 + lane;
            bool hit = false; long long m2 = 0;
            if (r < be) {
                const int rlen = (int)(b.read_off[r + 1] - b.read_off[r]);
                hit = b.read_pos[r] <= vmax && b.read_end[r] > vmin &&
                      qual_good_at((const int8_t*)b.read_qual + b.read_off[r], rlen, b.read_pos[r], vmin, vmax);
                if (hit) m2 = (long long)b.read_mapq[r] * b.read_mapq[r];
            }
            c[1] += __popcll(__ballot(hit));
#pragma unroll
            for (int s = 32; s > 0; s >>= 1) m2 += __shfl_xor(m2, s);
            c[15] += m2;
        }
        int nReads = 0, nVarReads = 0;
        for (int r0 = gb; r0 < ge; r0 += 64) {
            const int r = r0 + lane;
            bool hit = false, rev = false, sup = false; long long m2 = 0; int wmin = 0;
            if (r < ge) {
                const long long so = b.read_off[r];
                const int rlen = (int)(b.read_off[r + 1] - so);
                const int8_t* q = (const int8_t*)b.read_qual + so;
                const int rp = b.read_pos[r];
                hit = rp <= vmax && b.read_end[r] > vmin && qual_good_at(q, rlen, rp, vmin, vmax);
                if (hit) {
                    rev = (b.read_flags[r] & 16) != 0;
                    m2 = (long long)b.read_mapq[r] * b.read_mapq[r];
                    sup = read_supports(b.read_seq + so, rlen, rp, b.cigar + 2ll * b.cig_off[r], b.cig_off[r + 1] - b.cig_off[r], varPos,
                                        nAdded, nRemoved, added, exact);
                    if (sup && inGt) {                                       // MMLQ window, :1372-1383
                        const int ws = max(0, vmin - rp - half), we = min(rlen, vmax - rp + half);
                        for (int k = ws; k < we; ++k) wmin = (k == ws) ? (int)q[k] : min(wmin, (int)q[k]);
                    }
                }
            }
            const unsigned long long mh = __ballot(hit), mr = __ballot(hit && rev), ms = __ballot(sup), msr = __ballot(sup && rev);
            const int nh = __popcll(mh), nhr = __popcll(mr), ns = __popcll(ms), nsr = __popcll(msr);
            nReads += nh; c[0] += nh; c[7] += nhr; c[8] += nh - nhr;
            c[2] += ns; nVarReads += ns; c[11] += nsr; c[12] += ns - nsr;
            if (inGt) {
                c[3] += nh; c[9] += nhr; c[10] += nh - nhr;
                c[4] += ns; c[5] += nsr; c[6] += ns - nsr;
                if (sup) mq[nmq + __popcll(ms & ((1ull << lane) - 1ull))] = wmin;
                nmq += ns;
            }
#pragma unroll
            for (int s = 32; s > 0; s >>= 1) m2 += __shfl_xor(m2, s);
            c[15] += m2;
        }
        if (lane == 0) {
            per_sample[((long long)v * b.n_ind + i) * 2] = nReads;
            per_sample[((long long)v * b.n_ind + i) * 2 + 1] = nVarReads;
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 16; ++k) out[16ll * v + k] = c[k];
        nminq[v] = nmq;
    }
}

}  // namespace plat

PLAT_EXPORT int plat_variant_read_stats_batch(plat_ctx* ctx, const plat_infostats_batch* batch, int bad_reads_window,
                                              int count_only_exact_indel_matches, int64_t* out_counts, int32_t* out_per_sample,
                                              int32_t* out_minq, int32_t* out_nminq, void* stream)
{
    if (!ctx || !batch || bad_reads_window < 1) return PLAT_ERR_INVALID;
    const plat_infostats_batch b = *batch;
    if (b.n_vars < 0 || b.n_ind < 1) return PLAT_ERR_INVALID;
    if (b.n_vars == 0) return PLAT_OK;
    if (!b.var_window || !b.var_pos || !b.var_bam_min || !b.var_bam_max || !b.var_n_added || !b.var_n_removed || !b.var_added ||
        !b.var_added_off || !b.var_in_genotype || !b.minq_off || !b.good_begin || !b.good_end || !b.bad_begin || !b.bad_end ||
        !b.read_seq || !b.read_qual || !b.read_off || !b.read_pos || !b.read_end || !b.read_mapq || !b.read_flags || !b.cigar ||
        !b.cig_off || !out_counts || !out_per_sample || !out_minq || !out_nminq)
        return PLAT_ERR_INVALID;
    PLAT_HIP(ctx, hipSetDevice(ctx->device));
    { PLAT_KT_BEGIN(ctx, PLAT_KT_VARIANT_READ_STATS, (hipStream_t)stream); hipLaunchKernelGGL(plat::k_variant_read_stats, dim3(b.n_vars), dim3(64), 0, (hipStream_t)stream, b, bad_reads_window,
                       count_only_exact_indel_matches, out_counts, out_per_sample, out_minq, out_nminq); PLAT_KT_END(ctx, PLAT_KT_VARIANT_READ_STATS, (hipStream_t)stream); }
    PLAT_HIP(ctx, hipGetLastError());
    return PLAT_OK;
}

// ---- the loops of INFO ABPV / SbPval / MMLQ on the device (plat_variant_info_batch) -----------------------------------------------------
namespace plat {
constexpr int LOGFACT_N = 4096;
// betaBinomialCDF(k, n, alpha, beta) as its three terms (platypusutils.pyx:267-315): t[1] = logBeta(beta + n - k - 1, alpha + k + 1),
// t[2] = threeFTwo(k, n, alpha, beta), t[3] = logBeta(alpha, beta) + logBeta(n - k, k + 2) + log(n + 1).  Returns the state word: 1, or 2
// when a log-factorial lies beyond the table.  (k == n is the caller's: the function returns 1.0 there.)
__device__ __forceinline__ double cdf_terms(long long k, long long n, long long alpha, long long beta, const double* __restrict__ LF, double* t)
{
    const double* LOGI = LF + LOGFACT_N;                                  // LOGI[m - 1] = log((double)m), m = 1 .. 4096
    auto in = [](long long x) { return x >= 0 && x < LOGFACT_N; };
    const long long x1 = beta + n - k - 1, y1 = alpha + k + 1, x3 = n - k, y3 = k + 2;
    if (!in(x1 - 1) || !in(y1 - 1) || !in(x1 + y1 - 1) || !in(alpha - 1) || !in(beta - 1) || !in(alpha + beta - 1) || !in(x3 - 1) || !in(y3 - 1) ||
        !in(x3 + y3 - 1) || n + 1 < 1 || n + 1 > LOGFACT_N)
        return 2.0;
    auto logBeta = [&](long long x, long long y) { return (LF[x - 1] + LF[y - 1]) - LF[x + y - 1]; };
    const double a_2 = (double)alpha + (double)k + 1.0, a_3 = (double)k - (double)n + 1.0, b_1 = (double)k + 2.0,
                 b_2 = -(double)beta - (double)n + (double)k + 2.0;
    double theSum = 1.0, lastTerm = 1.0;
    const long long m = k - n + 1 < 0 ? -(k - n + 1) : k - n + 1;
    for (long long i = 1; i <= m; ++i) {
        const double di = (double)i;
        const double newTerm = lastTerm * (a_2 + di - 1) * (a_3 + di - 1) / ((b_1 + di - 1) * (b_2 + di - 1));
        theSum += newTerm;
        lastTerm = newTerm;
    }
    t[1] = logBeta(x1, y1);
    t[2] = theSum;
    t[3] = logBeta(alpha, beta) + logBeta(x3, y3) + LOGI[n];              // log((double)(n + 1))
    return 1.0;
}

// one wave per variant: lane 0 the two p-values' terms, the wave the median
__global__ void __launch_bounds__(64)
k_variant_info(int n_vars, const int64_t* __restrict__ counts, const int64_t* __restrict__ minq_off, const int32_t* __restrict__ minq,
               const int32_t* __restrict__ n_minq, const double* __restrict__ LF, double* __restrict__ out_terms, int32_t* __restrict__ out_mmlq)
{
    const int v = blockIdx.x, lane = threadIdx.x;
    if (v >= n_vars) return;
    const int64_t* c = counts + 16ll * v;
    if (lane == 0) {
        double* t = out_terms + 8ll * v;
        // computeAlleleBiasPValue(totalReads = TC_ab, variantReads = TR_ab), vcfutils.pyx:1156-1173
        const long long total = c[3], var = c[4];
        t[0] = 0.0; t[1] = 1.0; t[2] = 0.0; t[3] = 0.0;
        if (!(total > 0 && (double)var / (double)total >= 0.5) && total != 0) {
            if (var == total) { t[1] = 0.0; }                            // betaBinomialCDF == 1.0: min(1.0, 0.0)
            else t[0] = cdf_terms(var, total, 20, 20, LF, t);
        }
        // computeStrandBiasPValue(nFwdReads = TCF_sb, nRevReads = TCR_sb, nFwdVarReads = NF_sb, nRevVarReads = NR_sb), :1177-1222
        double* u = t + 4;
        const long long nF = c[10], nR = c[9], vF = c[6], vR = c[5];
        u[0] = 0.0; u[1] = 1.0; u[2] = 0.0; u[3] = 0.0;
        if (!(nF == 0 || nR == 0) && nF + nR > 0 && vF + vR > 0) {
            const bool useForward = !(nF < nR);
            const double freq = (double)(useForward ? nF : nR) / (double)(nF + nR);
            long long alpha, beta;
            if (freq < 0.5) { alpha = 20; beta = (long long)((double)alpha / freq - (double)alpha); }
            else if (freq > 0.5) { beta = 20; alpha = (long long)((double)beta * freq / (1.0 - freq)); }
            else alpha = beta = 20;
            const long long k = useForward ? vF : vR, n = vF + vR;
            if (k == n) u[1] = 1.0;                                      // betaBinomialCDF's own early exit
            else u[0] = cdf_terms(k, n, alpha, beta, LF, u);
        }
    }
    // sorted(values)[n // 2]: the element with n // 2 smaller-or-earlier-equal ones in front of it
    const int n = n_minq[v];
    const int32_t* q = minq + minq_off[v];
    int med = 100;
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        const int mine = i < n ? q[i] : 0;
        int rank = 0;
        for (int j = 0; j < n; ++j) { const int o = q[j]; rank += (o < mine) || (o == mine && j < i); }
        const unsigned long long hit = __ballot(i < n && rank == n / 2);
        if (hit) med = __shfl(mine, (int)__builtin_ctzll(hit), 64);
    }
    if (lane == 0) out_mmlq[v] = med;
}
}  // namespace plat

PLAT_EXPORT int plat_variant_info_batch(plat_ctx* ctx, int n_vars, const int64_t* counts, const int64_t* minq_off, const int32_t* minq,
                                        const int32_t* n_minq, double* out_terms, int32_t* out_mmlq, void* stream)
{
    if (!ctx || n_vars < 0) return PLAT_ERR_INVALID;
    if (n_vars == 0) return PLAT_OK;
    if (!counts || !minq_off || !minq || !n_minq || !out_terms || !out_mmlq) return PLAT_ERR_INVALID;
    PLAT_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->d_logfact) {
        // logFactorial as the reference computes it (platypusutils.pyx:178-191: a sum of logs below 15, Stirling's series with pow() above) and
        // log(m), both with the HOST's libm: the device adds table entries, it never calls a transcendental function for these fields
        static double lut[2 * plat::LOGFACT_N];
        static bool made = false;
        static std::mutex mk;
        {
            std::lock_guard<std::mutex> g(mk);
            if (!made) {
                for (int x = 0; x < plat::LOGFACT_N; ++x) {
                    double ans = 0.0;
                    if (x < 15) for (int i = 1; i <= x; ++i) ans += log((double)i);
                    else {
                        const double y = (double)x;
                        ans = (y * log(y) + log(2.0 * M_PI * y) / 2 - y + (pow(y, -1)) / 12 - (pow(y, -3)) / 360 + (pow(y, -5)) / 1260 - (pow(y, -7)) / 1680 + (pow(y, -9)) / 1188);
                    }
                    lut[x] = ans;
                    lut[plat::LOGFACT_N + x] = log((double)(x + 1));
                }
                made = true;
            }
        }
        PLAT_HIP(ctx, hipMalloc(&ctx->d_logfact, sizeof(lut)));
        PLAT_HIP(ctx, hipMemcpy(ctx->d_logfact, lut, sizeof(lut), hipMemcpyHostToDevice));
    }
    { PLAT_KT_BEGIN(ctx, PLAT_KT_VARIANT_INFO, (hipStream_t)stream); hipLaunchKernelGGL(plat::k_variant_info, dim3((unsigned)n_vars), dim3(64), 0, (hipStream_t)stream, n_vars, counts, minq_off, minq, n_minq,
                       (const double*)ctx->d_logfact, out_terms, out_mmlq); PLAT_KT_END(ctx, PLAT_KT_VARIANT_INFO, (hipStream_t)stream); }
    PLAT_HIP(ctx, hipGetLastError());
    return PLAT_OK;
}


// ---- window read slices out of a resident read table (plat_gather_reads) --------------------------------------------------
namespace plat {
// Four destination reads per wave, sixteen lanes each: lanes 0-7 of the sixteen move the bases, 8-15 the qualities, 16 bytes at a time
// from and to any address (global memory takes unaligned 64-bit loads and stores).  A read's copy is a chain of three dependent loads
// (index -> offsets -> bytes) that a wave waits out whatever its width: with one read per wave (rounds 3-4) the 230 k reads of a chunk of
// 64 regions were 28 rounds of waves on the chip, 76-85 us for 86 MB; four per wave share each wait.
__global__ void __launch_bounds__(256)
k_gather_reads(long long n_dst, const int32_t* __restrict__ src_index, const int64_t* __restrict__ dst_off,
               const uint8_t* __restrict__ src_seq, const uint8_t* __restrict__ src_qual, const int64_t* __restrict__ src_off,
               const int32_t* __restrict__ src_pos, const int32_t* __restrict__ src_end, const uint8_t* __restrict__ src_mapq,
               const int32_t* __restrict__ src_flags, uint8_t* __restrict__ dst_seq, uint8_t* __restrict__ dst_qual,
               int32_t* __restrict__ dst_pos, int32_t* __restrict__ dst_end, uint8_t* __restrict__ dst_mapq, int32_t* __restrict__ dst_flags)
{
    typedef unsigned long long __attribute__((aligned(1))) u64u;
    const int l16 = threadIdx.x & 15, l8 = l16 & 7;
    const long long ng = ((long long)gridDim.x * blockDim.x) >> 4;
    for (long long d = (((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 4); d < n_dst; d += ng) {
        const int s = src_index[d];
        const long long so = src_off[s], n = src_off[s + 1] - so, to = dst_off[d];
        const uint8_t* src = l16 < 8 ? src_seq + so : src_qual + so;
        uint8_t* dst = l16 < 8 ? dst_seq + to : dst_qual + to;
        if (n >= 16) {                                                  // pieces of 16 bytes; the last one ends AT the read's end (it may overlap the one before)
            const long long np = (n + 15) >> 4;
            for (long long k = l8; k < np; k += 8) {
                const long long i = k + 1 < np ? 16 * k : n - 16;
                const unsigned long long a = *(const u64u*)(src + i), c = *(const u64u*)(src + i + 8);
                *(u64u*)(dst + i) = a; *(u64u*)(dst + i + 8) = c;
            }
        } else for (long long i = l8; i < n; i += 8) dst[i] = src[i];
        if (l16 == 0) { dst_pos[d] = src_pos[s]; dst_end[d] = src_end[s]; dst_mapq[d] = src_mapq[s]; dst_flags[d] = src_flags[s]; }
    }
}
}  // namespace plat

PLAT_EXPORT int plat_gather_reads(plat_ctx* ctx, int64_t n_dst, const int32_t* src_index, const int64_t* dst_off,
                                  const uint8_t* src_seq, const uint8_t* src_qual, const int64_t* src_off, const int32_t* src_pos,
                                  const int32_t* src_end, const uint8_t* src_mapq, const int32_t* src_flags, uint8_t* dst_seq,
                                  uint8_t* dst_qual, int32_t* dst_pos, int32_t* dst_end, uint8_t* dst_mapq, int32_t* dst_flags,
                                  void* stream)
{
    if (!ctx || n_dst < 0) return PLAT_ERR_INVALID;
    if (n_dst == 0) return PLAT_OK;
    if (!src_index || !dst_off || !src_seq || !src_qual || !src_off || !src_pos || !src_end || !src_mapq || !src_flags || !dst_seq ||
        !dst_qual || !dst_pos || !dst_end || !dst_mapq || !dst_flags)
        return PLAT_ERR_INVALID;
    PLAT_HIP(ctx, hipSetDevice(ctx->device));
    const long long nblk = (n_dst + 15) / 16;                         // 256 threads = 16 reads
    { PLAT_KT_BEGIN(ctx, PLAT_KT_GATHER_READS, (hipStream_t)stream); hipLaunchKernelGGL(plat::k_gather_reads, dim3((unsigned)(nblk < 65535 * 8 ? nblk : 65535 * 8)), dim3(256), 0, (hipStream_t)stream, (long long)n_dst,
                       src_index, dst_off, src_seq, src_qual, src_off, src_pos, src_end, src_mapq, src_flags, dst_seq, dst_qual, dst_pos,
                       dst_end, dst_mapq, dst_flags); PLAT_KT_END(ctx, PLAT_KT_GATHER_READS, (hipStream_t)stream); }
    PLAT_HIP(ctx, hipGetLastError());
    return PLAT_OK;
}


// ---- read tables packed at one byte per base (plat_unpack_reads) ----------------------------------------------------------
namespace plat {
// 16 packed bytes per thread: one 128-bit load (two 64-bit ones from any address when the source does not share the outputs' alignment:
// global memory takes unaligned loads), two 128-bit stores.  The two OUTPUT arrays may start anywhere as long as they share their
// misalignment `mis` (a read table inside a chunk's blob does): thread t takes the 16-byte line [16 t - mis, 16 t - mis + 16) of them,
// clipped to [0, n) at the two ends.
__global__ void __launch_bounds__(256)
k_unpack_reads(long long n, int mis, int vec, int src_aligned, const uint8_t* __restrict__ packed, uint8_t* __restrict__ out_seq, uint8_t* __restrict__ out_qual)
{
    const long long i0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 16 - mis;
    if (i0 >= n) return;
    if (vec && i0 >= 0 && i0 + 16 <= n) {
        uint32_t w[4];
        if (src_aligned) { const uint4 v = *(const uint4*)(packed + i0); w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w; }
        else {
            const unsigned long long lo = load_u64_bytes(packed + i0), hi = load_u64_bytes(packed + i0 + 8);
            w[0] = (uint32_t)lo; w[1] = (uint32_t)(lo >> 32); w[2] = (uint32_t)hi; w[3] = (uint32_t)(hi >> 32);
        }
        uint32_t sq[4], ql[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t code = w[k] & 0x03030303u;
            sq[k] = __builtin_amdgcn_perm(0u, 0x47544341u, code);           // byte selector 0..3 -> 'A' 'C' 'T' 'G'
            ql[k] = (w[k] >> 2) & 0x3F3F3F3Fu;
        }
        *(uint4*)(out_seq + i0) = make_uint4(sq[0], sq[1], sq[2], sq[3]);
        *(uint4*)(out_qual + i0) = make_uint4(ql[0], ql[1], ql[2], ql[3]);
    } else {
        for (long long i = i0 < 0 ? 0 : i0; i < n && i < i0 + 16; ++i) {
            const unsigned b = packed[i];
            out_seq[i] = (uint8_t)((0x47544341u >> (8u * (b & 3u))) & 0xFFu);
            out_qual[i] = (uint8_t)(b >> 2);
        }
    }
}
__global__ void __launch_bounds__(256)
k_unpack_exceptions(long long n_exc, long long n, const int64_t* __restrict__ idx, const uint8_t* __restrict__ eb, const uint8_t* __restrict__ eq,
                    uint8_t* __restrict__ out_seq, uint8_t* __restrict__ out_qual, uint32_t* __restrict__ codes = nullptr)
{
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_exc) return;
    const long long i = idx[k];
    if (i < 0 || i >= n) return;
    out_seq[i] = eb[k]; out_qual[i] = eq[k];
    if (codes) {                                                           // the base's code follows its byte: (ASCII >> 1) & 3
        const unsigned sh = 2u * (unsigned)(i & 15);
        atomicAnd(&codes[i >> 4], ~(3u << sh));
        atomicOr(&codes[i >> 4], (((unsigned)eb[k] >> 1) & 3u) << sh);
    }
}
}  // namespace plat

PLAT_EXPORT int plat_unpack_reads(plat_ctx* ctx, int64_t n_bytes, const uint8_t* packed, uint8_t* out_seq, uint8_t* out_qual,
                                  int64_t n_exc, const int64_t* exc_index, const uint8_t* exc_base, const uint8_t* exc_qual, void* stream)
{
    if (!ctx || n_bytes < 0 || n_exc < 0) return PLAT_ERR_INVALID;
    if (n_bytes == 0) return PLAT_OK;
    if (!packed || !out_seq || !out_qual || (n_exc > 0 && (!exc_index || !exc_base || !exc_qual))) return PLAT_ERR_INVALID;
    PLAT_HIP(ctx, hipSetDevice(ctx->device));
    // lanes work on 16-byte lines of the OUTPUT arrays: possible when the two share their misalignment (else byte by byte); a source
    // with another misalignment (a resident table expanded into its place in a chunk's blob) is read with unaligned loads
    const int m0 = (int)((uintptr_t)packed & 15), m1 = (int)((uintptr_t)out_seq & 15), m2 = (int)((uintptr_t)out_qual & 15);
    const int vec = m1 == m2, mis = vec ? m1 : 0;                          // (not shared: every thread walks its own 16 bytes)
    const long long nthr = (n_bytes + mis + 15) / 16;
    hipLaunchKernelGGL(plat::k_unpack_reads, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (long long)n_bytes,
                       mis, vec, (int)(m0 == mis), packed, out_seq, out_qual);
    if (n_exc > 0)
        hipLaunchKernelGGL(plat::k_unpack_exceptions, dim3((unsigned)((n_exc + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (long long)n_exc,
                           (long long)n_bytes, exc_index, exc_base, exc_qual, out_seq, out_qual);
    PLAT_HIP(ctx, hipGetLastError());
    return PLAT_OK;
}


namespace plat {
// k_unpack_reads for a list of pieces: blockIdx.y = piece; lanes work on 16-byte lines of the piece's place in the output
__global__ void __launch_bounds__(256)
k_unpack_pieces(const plat_unpack_piece* __restrict__ pieces, uint8_t* __restrict__ out_seq, uint8_t* __restrict__ out_qual, int same_base_alignment,
                uint32_t* __restrict__ codes)
{
    // codes (round 6, may be null): the bases again as 2 bits each (A 0, C 1, T 2, G 3 = the packed byte's low bits = (ASCII >> 1) & 3), base i of the
    // blob at bits 2 (i & 15) of dword i >> 4 -- what plat_candidates_batch_codes compares 32 bases per 64-bit word.  A whole line of 16 bases is one
    // plain store; the bases of a partial line (a piece's first and last) are OR-ed in, the buffer having been zeroed by the caller.
    const plat_unpack_piece pc = pieces[blockIdx.y];
    const uint8_t* packed = pc.src;
    uint8_t* oseq = out_seq + pc.dst;
    uint8_t* oqual = out_qual + pc.dst;
    const long long n = pc.n;
    const int mis = same_base_alignment ? (int)((uintptr_t)oseq & 15) : 0;
    const bool srcAligned = (((uintptr_t)packed - (uintptr_t)mis) & 15) == 0 || (((uintptr_t)packed & 15) == (unsigned)mis);
    const long long lines = (n + mis + 15) / 16;
    for (long long ln = (long long)blockIdx.x * blockDim.x + threadIdx.x; ln < lines; ln += (long long)gridDim.x * blockDim.x) {
        const long long i0 = ln * 16 - mis;
        if (same_base_alignment && i0 >= 0 && i0 + 16 <= n) {
            uint32_t w[4];
            if (srcAligned) { const uint4 v = *(const uint4*)(packed + i0); w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w; }
            else {
                const unsigned long long lo = load_u64_bytes(packed + i0), hi = load_u64_bytes(packed + i0 + 8);
                w[0] = (uint32_t)lo; w[1] = (uint32_t)(lo >> 32); w[2] = (uint32_t)hi; w[3] = (uint32_t)(hi >> 32);
            }
            uint32_t sq[4], ql[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t code = w[k] & 0x03030303u;
                sq[k] = __builtin_amdgcn_perm(0u, 0x47544341u, code);
                ql[k] = (w[k] >> 2) & 0x3F3F3F3Fu;
            }
            *(uint4*)(oseq + i0) = make_uint4(sq[0], sq[1], sq[2], sq[3]);
            *(uint4*)(oqual + i0) = make_uint4(ql[0], ql[1], ql[2], ql[3]);
            if (codes) {
                uint32_t cw = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    uint32_t t = w[k] & 0x03030303u;
                    t = (t | (t >> 6)) & 0x000F000Fu;
                    cw |= ((t | (t >> 12)) & 0xFFu) << (8 * k);
                }
                const long long at = pc.dst + i0;                          // blob index of the line's first base
                if ((at & 15) == 0) codes[at >> 4] = cw;
                else for (int k = 0; k < 16; ++k) atomicOr(&codes[(at + k) >> 4], ((cw >> (2 * k)) & 3u) << (2 * ((at + k) & 15)));
            }
        } else {
            for (long long i = i0 < 0 ? 0 : i0; i < n && i < i0 + 16; ++i) {
                const unsigned bb = packed[i];
                oseq[i] = (uint8_t)((0x47544341u >> (8u * (bb & 3u))) & 0xFFu);
                oqual[i] = (uint8_t)(bb >> 2);
                if (codes) atomicOr(&codes[(pc.dst + i) >> 4], (bb & 3u) << (2 * ((pc.dst + i) & 15)));
            }
        }
    }
}
}  // namespace plat

static int unpack_pieces(plat_ctx* ctx, int n_pieces, int64_t max_piece_bytes, const plat_unpack_piece* pieces, uint8_t* out_seq, uint8_t* out_qual, uint32_t* out_codes,
                         int64_t total_bytes, int64_t n_exc, const int64_t* exc_index, const uint8_t* exc_base, const uint8_t* exc_qual, void* stream);

PLAT_EXPORT int plat_unpack_reads_pieces(plat_ctx* ctx, int n_pieces, int64_t max_piece_bytes, const plat_unpack_piece* pieces, uint8_t* out_seq, uint8_t* out_qual,
                                         int64_t total_bytes, int64_t n_exc, const int64_t* exc_index, const uint8_t* exc_base, const uint8_t* exc_qual, void* stream)
{
    return unpack_pieces(ctx, n_pieces, max_piece_bytes, pieces, out_seq, out_qual, nullptr, total_bytes, n_exc, exc_index, exc_base, exc_qual, stream);
}

PLAT_EXPORT int plat_unpack_reads_pieces_codes(plat_ctx* ctx, int n_pieces, int64_t max_piece_bytes, const plat_unpack_piece* pieces, uint8_t* out_seq, uint8_t* out_qual,
                                               uint32_t* out_codes, int64_t total_bytes, int64_t n_exc, const int64_t* exc_index, const uint8_t* exc_base,
                                               const uint8_t* exc_qual, void* stream)
{
    if (!out_codes) return PLAT_ERR_INVALID;
    return unpack_pieces(ctx, n_pieces, max_piece_bytes, pieces, out_seq, out_qual, out_codes, total_bytes, n_exc, exc_index, exc_base, exc_qual, stream);
}

static int unpack_pieces(plat_ctx* ctx, int n_pieces, int64_t max_piece_bytes, const plat_unpack_piece* pieces, uint8_t* out_seq, uint8_t* out_qual, uint32_t* out_codes,
                         int64_t total_bytes, int64_t n_exc, const int64_t* exc_index, const uint8_t* exc_base, const uint8_t* exc_qual, void* stream)
{
    if (!ctx || n_pieces < 0 || max_piece_bytes < 0 || total_bytes < 0 || n_exc < 0) return PLAT_ERR_INVALID;
    if (n_pieces == 0) return PLAT_OK;
    if (!pieces || !out_seq || !out_qual || (n_exc > 0 && (!exc_index || !exc_base || !exc_qual))) return PLAT_ERR_INVALID;
    PLAT_HIP(ctx, hipSetDevice(ctx->device));
    // (the code words of partial lines are OR-ed in: the buffer starts as zeros; + 8 words: plat_candidates_batch_codes reads a 64-bit word past the last base)
    if (out_codes) PLAT_HIP(ctx, hipMemsetAsync(out_codes, 0, ((size_t)(total_bytes + 15) / 16 + 8) * sizeof(uint32_t), (hipStream_t)stream));
    const int same = ((uintptr_t)out_seq & 15) == ((uintptr_t)out_qual & 15);
    long long gx = (max_piece_bytes / 16 + 256) / 256;
    gx = gx < 1 ? 1 : (gx > 1024 ? 1024 : gx);
    PLAT_EV_TAB(ctx, 0, (hipStream_t)stream);
    for (int p0 = 0; p0 < n_pieces; p0 += PLAT_GRID_Y_MAX) {      // (gridDim.y holds at most 65 535 pieces: one launch per batch of them)
        const int np = n_pieces - p0 < PLAT_GRID_Y_MAX ? n_pieces - p0 : PLAT_GRID_Y_MAX;
        { PLAT_KT_BEGIN(ctx, PLAT_KT_UNPACK_PIECES, (hipStream_t)stream); hipLaunchKernelGGL(plat::k_unpack_pieces, dim3((unsigned)gx, (unsigned)np), dim3(256), 0, (hipStream_t)stream, pieces + p0, out_seq, out_qual, same, out_codes); PLAT_KT_END(ctx, PLAT_KT_UNPACK_PIECES, (hipStream_t)stream); }
    }
    PLAT_EV_TAB(ctx, 1, (hipStream_t)stream);
    ctx->ev_valid_unpack = ctx->profile;
    if (n_exc > 0)
        hipLaunchKernelGGL(plat::k_unpack_exceptions, dim3((unsigned)((n_exc + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (long long)n_exc,
                           (long long)total_bytes, exc_index, exc_base, exc_qual, out_seq, out_qual, out_codes);
    PLAT_HIP(ctx, hipGetLastError());
    return PLAT_OK;
}

// ---- a chunk's read table put together on the device from resident tables (plat_concat_read_tables) ----------------------------------
namespace plat {
__global__ void __launch_bounds__(256)
k_concat_tables(const plat_table_desc* __restrict__ desc, int64_t* __restrict__ dst_off, int32_t* __restrict__ dst_pos, int32_t* __restrict__ dst_end,
                uint8_t* __restrict__ dst_mapq, int32_t* __restrict__ dst_flags, int32_t* __restrict__ dst_cig_off, int16_t* __restrict__ dst_cigar,
                int32_t* __restrict__ dst_region, long long n_total, long long total_bytes, long long total_pairs)
{
    const plat_table_desc d = desc[blockIdx.y];
    const int stride = gridDim.x * blockDim.x;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < d.n; i += stride) {
        const long long r = d.first_read + i;
        dst_off[r] = d.first_byte + d.off[i];
        dst_pos[r] = d.pos[i]; dst_end[r] = d.end[i]; dst_mapq[r] = d.mapq[i]; dst_flags[r] = d.flags[i];
        const int c0 = d.cig_off[i], c1 = d.cig_off[i + 1];
        dst_cig_off[r] = (int32_t)(d.first_pair + c0);
        for (int c = c0; c < c1; ++c) {
            dst_cigar[2 * (d.first_pair + c)] = d.cigar[2 * c];
            dst_cigar[2 * (d.first_pair + c) + 1] = d.cigar[2 * c + 1];
        }
        if (d.scan >= 0) dst_region[r] = d.scan;
    }
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        dst_off[n_total] = total_bytes; dst_cig_off[n_total] = (int32_t)total_pairs;
        dst_cigar[2 * total_pairs] = 0; dst_cigar[2 * total_pairs + 1] = 0;
    }
}
}  // namespace plat

PLAT_EXPORT int plat_concat_read_tables(plat_ctx* ctx, int n_tables, int max_reads_per_table, const plat_table_desc* desc, int64_t* dst_off, int32_t* dst_pos,
                                        int32_t* dst_end, uint8_t* dst_mapq, int32_t* dst_flags, int32_t* dst_cig_off, int16_t* dst_cigar,
                                        int32_t* dst_region, int64_t n_total_reads, int64_t total_bytes, int64_t total_pairs, void* stream)
{
    if (!ctx || n_tables < 0 || max_reads_per_table < 0 || n_total_reads < 0) return PLAT_ERR_INVALID;
    if (!dst_off || !dst_pos || !dst_end || !dst_mapq || !dst_flags || !dst_cig_off || !dst_cigar || !dst_region) return PLAT_ERR_INVALID;
    if (n_tables < 1 || !desc) return PLAT_ERR_INVALID;
    PLAT_HIP(ctx, hipSetDevice(ctx->device));
    unsigned gx = (unsigned)((max_reads_per_table + 255) / 256);
    gx = gx < 1 ? 1 : (gx > 64 ? 64 : gx);
    for (int t0 = 0; t0 < n_tables; t0 += PLAT_GRID_Y_MAX) {      // (every batch's first block also writes the table's closing entries: the same values)
        const int nt = n_tables - t0 < PLAT_GRID_Y_MAX ? n_tables - t0 : PLAT_GRID_Y_MAX;
        { PLAT_KT_BEGIN(ctx, PLAT_KT_CONCAT_TABLES, (hipStream_t)stream); hipLaunchKernelGGL(plat::k_concat_tables, dim3(gx, (unsigned)nt), dim3(256), 0, (hipStream_t)stream, desc + t0, dst_off, dst_pos, dst_end,
                           dst_mapq, dst_flags, dst_cig_off, dst_cigar, dst_region, (long long)n_total_reads, (long long)total_bytes, (long long)total_pairs); PLAT_KT_END(ctx, PLAT_KT_CONCAT_TABLES, (hipStream_t)stream); }
    }
    PLAT_HIP(ctx, hipGetLastError());
    return PLAT_OK;
}


// ---- pieces of device memory into one blob (plat_copy_pieces) -----------------------------------------------------------------------------
namespace plat {
__global__ void __launch_bounds__(256)
k_copy_pieces(const plat_unpack_piece* __restrict__ pieces, uint8_t* __restrict__ dst_blob)
{
    typedef unsigned long long __attribute__((aligned(1))) u64u;
    const plat_unpack_piece pc = pieces[blockIdx.y];
    const uint8_t* src = pc.src;
    uint8_t* dst = dst_blob + pc.dst;
    const long long n8 = pc.n & ~7ll;
    const long long stride = 8ll * gridDim.x * blockDim.x;
    for (long long i = 8ll * ((long long)blockIdx.x * blockDim.x + threadIdx.x); i < n8; i += stride) *(u64u*)(dst + i) = *(const u64u*)(src + i);
    if (blockIdx.x == 0) for (long long i = n8 + threadIdx.x; i < pc.n; i += blockDim.x) dst[i] = src[i];
}
}  // namespace plat

PLAT_EXPORT int plat_copy_pieces(plat_ctx* ctx, int n_pieces, int64_t max_piece_bytes, const plat_unpack_piece* pieces, uint8_t* dst_blob, void* stream)
{
    if (!ctx || n_pieces < 0 || max_piece_bytes < 0) return PLAT_ERR_INVALID;
    if (n_pieces == 0) return PLAT_OK;
    if (!pieces || !dst_blob) return PLAT_ERR_INVALID;
    PLAT_HIP(ctx, hipSetDevice(ctx->device));
    long long gx = (max_piece_bytes / 8 + 256) / 256;
    gx = gx < 1 ? 1 : (gx > 256 ? 256 : gx);
    for (int p0 = 0; p0 < n_pieces; p0 += PLAT_GRID_Y_MAX) {      // (a whole job's regions are one piece each in the exchange: more than gridDim.y holds)
        const int np = n_pieces - p0 < PLAT_GRID_Y_MAX ? n_pieces - p0 : PLAT_GRID_Y_MAX;
        { PLAT_KT_BEGIN(ctx, PLAT_KT_COPY_PIECES, (hipStream_t)stream); hipLaunchKernelGGL(plat::k_copy_pieces, dim3((unsigned)gx, (unsigned)np), dim3(256), 0, (hipStream_t)stream, pieces + p0, dst_blob); PLAT_KT_END(ctx, PLAT_KT_COPY_PIECES, (hipStream_t)stream); }
    }
    PLAT_HIP(ctx, hipGetLastError());
    return PLAT_OK;
}
