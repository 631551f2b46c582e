// plat_candidates.hip -- VariantCandidateGenerator (src/cython/variant.pyx:459-751): variant candidates
// from the CIGAR strings and the mismatches of the reads of a region.
//
// One lane walks one read exactly as getVariantCandidatesFromSingleRead / getSnpCandidatesFromReadSegment do (the
// mismatch-run state machine is sequential in the read), and writes one record per candidate OCCURRENCE into the read's
// own slice of the output -- so records come out in the reference's emission order (read by read, left to right) without
// any ordering step.  Merging equal variants and counting their supporting reads (addVariantToList, :499-527) is a
// dictionary operation over the few records that come back; it stays on the host (hostapi.VariantCandidateGenerator).
#include "plat_internal.hpp"

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
                    // getSnpCandidatesFromReadSegment, :529-612
                    int msr = -1, mer = -1, msd = -1, med = -1;
                    for (int index = 0; index < length; ++index) {
                        if (readOffset == 0 && index < min_flank) continue;
                        if (index + readOffset >= rlen - min_flank) continue;
                        const int readIndex = index + readOffset;
                        const int refIndex = (index + refOffset + readStart) - refSeqStart;
                        if (refIndex < 0 || refIndex >= refLen) { st = PLAT_ERR_BAD_INPUT; break; }   // the reference reads past its buffer here
                        const uint8_t readChar = readSeq[readIndex], refChar = ref[refIndex];
                        if (readChar != refChar) {
                            if (readChar != 'N' && refChar != 'N' && (int)readQual[readIndex] >= min_base_qual) {
                                if (msr == -1) { msr = mer = refIndex; msd = med = readIndex; }
                                else if (refIndex - mer <= min_flank) { mer = refIndex; med = readIndex; }
                                else {
                                    out.put(msr + refSeqStart, mer - msr + 1, med - msd + 1, roff + msr, soff + msd);
                                    msr = mer = refIndex; msd = med = readIndex;
                                }
                            }
                        } else if (msr != -1 && refIndex - mer > min_flank) {
                            out.put(msr + refSeqStart, mer - msr + 1, med - msd + 1, roff + msr, soff + msd);
                            msr = mer = msd = med = -1;
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
    hipLaunchKernelGGL(plat::k_candidates, dim3((unsigned)((b.n_reads + 63) / 64)), dim3(64), 0, (hipStream_t)stream, b,
                       min_flank, min_base_qual, gen_snps, gen_indels, max_per_read, read_region, out_rec, out_count, out_status);
    PLAT_HIP(ctx, hipGetLastError());
    return PLAT_OK;
}
