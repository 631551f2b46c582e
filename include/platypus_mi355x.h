/*
 * platypus_mi355x.h -- C ABI of libplat_mi355x.so
 *
 * MI355X-native (gfx950, hand-written HIP) replacement for the read->haplotype likelihood and
 * local-assembly hot path of andyrimmer/Platypus v0.8.1.1.  Every entry point names the reference
 * interface it replaces (file:line relative to the reference tree).  Plain C, plain pointers and
 * sizes, no C++/torch types; intended to be bound with cgo / JNI / ctypes / Cython `cdef extern`
 * (see INTEGRATION.md for the Cython stub a Platypus maintainer would add).
 *
 * Conventions
 *  - every function returns int: 0 = PLAT_OK, negative = error (plat_strerror()).  No exceptions
 *    cross the boundary (the reference raises Python exceptions: chaplotype.pyx:325-334,585-586).
 *  - all *batch* entry points take DEVICE pointers (HBM-resident; obtain with plat_malloc or pass
 *    e.g. torch.Tensor.data_ptr()) and a `stream` (hipStream_t as void*; NULL = default stream).
 *    They enqueue work and return; call plat_stream_sync() (or synchronise the stream yourself)
 *    before reading results.  Exceptions are flagged "[syncs]".
 *  - byte blobs (sequences, qualities) are 7-bit ASCII exactly as the reference holds them in
 *    cAlignedRead.seq / .qual (raw phred, NOT +33; htslibWrapper.pyx:366-368) and
 *    Haplotype.cHaplotypeSequence.  Every blob must be followed by >= PLAT_BLOB_PAD readable bytes.
 *  - the caller owns all buffers it passes; the context owns only its internal scratch.
 *  - a context is bound to one GPU and is not thread-safe (the reference is single-threaded per
 *    process: SURVEY.md 8(b)); use one context per process/rank -- or, as libplat_caller.so does, one per
 *    worker thread.  That includes plat_stream_sync: it records and waits on ONE event owned by the
 *    context, so two threads syncing different streams of the same context at once may return early.
 */
#ifndef PLATYPUS_MI355X_H
#define PLATYPUS_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PLAT_ABI_VERSION 1
#define PLAT_BLOB_PAD 32

/* error codes */
#define PLAT_OK 0
#define PLAT_ERR_INVALID (-1)       /* bad argument (NULL pointer, negative size, ...)              */
#define PLAT_ERR_HIP (-2)           /* HIP runtime error; plat_last_hip_error() has the code        */
#define PLAT_ERR_NOMEM (-3)         /* device allocation failed ("Out of memory in cHaplotype.alignReads") */
#define PLAT_ERR_HAP_TOO_LONG (-4)  /* haplotype longer than 16384 (chaplotype.pyx:180-183)         */
#define PLAT_ERR_HAP_TOO_SHORT (-5) /* hapLen < readLen+15: the reference reads past the buffer here */
#define PLAT_ERR_UNSUPPORTED (-6)   /* option combination not implemented on the device path         */
#define PLAT_ERR_NO_DEVICE (-7)     /* no gfx950 device / HIP runtime unavailable                    */
#define PLAT_ERR_OVERFLOW (-8)      /* an output capacity given by the caller was too small          */
#define PLAT_ERR_BAD_INPUT (-9)     /* device-side input validation failed (non-ASCII byte, ...)     */
#define PLAT_ERR_BAD_HINTS (-10)    /* plat_batch_hints do not cover the batch (asynchronous entry point) */

typedef struct plat_ctx plat_ctx;

/* ---- context & memory ------------------------------------------------------------------------- */
int plat_abi_version(void);
const char* plat_strerror(int code);
int plat_device_count(int* out_count);
int plat_ctx_create(int device, plat_ctx** out_ctx);
int plat_ctx_destroy(plat_ctx* ctx);
int plat_last_hip_error(const plat_ctx* ctx);               /* hipError_t of the last PLAT_ERR_HIP   */
int plat_malloc(plat_ctx* ctx, size_t bytes, void** out_dev_ptr);
int plat_free(plat_ctx* ctx, void* dev_ptr);
int plat_memcpy_h2d(plat_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes, void* stream);
int plat_memcpy_d2h(plat_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes, void* stream);
int plat_memcpy_d2d(plat_ctx* ctx, void* dst_dev, const void* src_dev, size_t bytes, void* stream);
int plat_memset(plat_ctx* ctx, void* dst_dev, int value, size_t bytes, void* stream);
int plat_stream_sync(plat_ctx* ctx, void* stream);          /* [syncs] */
int plat_sync_poll_us(plat_ctx* ctx, int microseconds);     /* how often plat_stream_sync looks at its event (default 40; 0 = hipEventSynchronize) */
/* plat_stream_sync waits asleep: it polls an event every plat_sync_poll_us microseconds (default 40: a caller whose waits last a
 * millisecond or more asks for more -- every look is a system call and a runtime query, and with two dozen waiting threads they add up:
 * the native region loop uses 500) -- or every PLAT_SYNC_POLL_US microseconds when the environment says so (0 = the
 * runtime's blocking hipEventSynchronize; PLAT_SYNC_SPIN=1 = hipStreamSynchronize).  While it naps it lowers the CALLING thread's
 * timer slack (prctl PR_SET_TIMERSLACK) to 2 us and puts the previous value back before it returns. */
/* a HIP stream of the context's device (hipStream_t as void*), for callers without a HIP runtime binding of their own;
 * pinned (page-locked) host memory for staging buffers: copies from / to it are asynchronous and run at link speed */
int plat_stream_create(plat_ctx* ctx, void** out_stream);
int plat_stream_destroy(plat_ctx* ctx, void* stream);
int plat_host_alloc(plat_ctx* ctx, size_t bytes, void** out_host_ptr);
int plat_host_free(plat_ctx* ctx, void* host_ptr);

/* ---- live kernel timing (HIP events recorded on the caller's stream around each kernel) ----------
 * Used by bench.py for the roofline line; off by default (no overhead).  plat_profile_last [syncs]
 * on the recorded events and returns the durations of the kernels of the most recent
 * plat_align_window_batch / plat_genotype_window_batch call, plus the algorithmic byte count of
 * the DP launch: sum over launched DPs of (4*len2 + 34) bytes.                  */
typedef struct plat_profile {
    float ms_prepare;     /* validation + haplotype->window map (+ read-back)                     */
    float ms_seed;        /* gap-open annotation + 7-mer index + diagonal vote + candidate lists  */
    float ms_dp;          /* banded DP kernel (the dominant kernel)                               */
    float ms_finalize;    /* candidate selection + score -> log-likelihood                        */
    float ms_genotype;    /* genotype likelihood kernel                                           */
    float ms_seed_kernel; /* k_seed alone (the first and largest kernel of the seed stage)         */
    int64_t dp_jobs;      /* DPs in the DP launch                                                 */
    int64_t dp_alg_bytes; /* algorithmic bytes of the DP launch                                   */
    float ms_sweep;       /* k_sweep alone (round 4: the seeding stage is two kernels; 0 with PLAT_SEED_FUSED=1) */
    float ms_pairs;       /* k_pairs alone                                                         */
    float ms_unpack;      /* the last plat_unpack_reads_pieces launch on this context since the profile was switched on (0: none) */
    float ms_candidates;  /* the last plat_candidates_batch launch (k_candidates)                     */
} plat_profile;
int plat_profile_enable(plat_ctx* ctx, int on);
int plat_profile_last(plat_ctx* ctx, plat_profile* out);   /* [syncs] */

/* The same for EVERY kernel of the region loop's chunk (round 6): while the profile is on, each launch listed below is bracketed by
 * its own pair of HIP events on the launch stream; plat_kernel_times [syncs] resolves the pairs recorded since the last call and ADDS
 * each launch's duration (ms) and 1 to out_ms[id] / out_launches[id] (arrays of PLAT_KT_COUNT entries, the caller zeroes them).  A kernel
 * shares the chip only with what the caller lets run beside it: bench.py's counting pass runs one chunk at a time. */
enum {
    PLAT_KT_CANDIDATES = 0, PLAT_KT_CAND_MERGE, PLAT_KT_CAND_FILTER, PLAT_KT_UNPACK_PIECES, PLAT_KT_CONCAT_TABLES, PLAT_KT_COPY_PIECES,
    PLAT_KT_GATHER_READS, PLAT_KT_SB_VARIANTS, PLAT_KT_SB_WINDOWS, PLAT_KT_SB_HAPS_RANK, PLAT_KT_SB_PREFIX, PLAT_KT_SB_SCAN,
    PLAT_KT_SB_HAPS_WRITE, PLAT_KT_SB_READS, PLAT_KT_VALIDATE, PLAT_KT_TILE_SCAN, PLAT_KT_PREP_READS, PLAT_KT_SWEEP, PLAT_KT_PAIRS,
    PLAT_KT_SEED_SLOW, PLAT_KT_DP_JOBS, PLAT_KT_FINALIZE, PLAT_KT_GENOTYPE, PLAT_KT_HAPLOTYPE_SCORE, PLAT_KT_EM, PLAT_KT_VARIANT_POSTERIOR,
    PLAT_KT_VARIANT_READ_STATS, PLAT_KT_VARIANT_INFO, PLAT_KT_GENOTYPE_CALL, PLAT_KT_ASSEMBLE, PLAT_KT_READ_QC, PLAT_KT_OTHER,
    PLAT_KT_COUNT
};
const char* plat_kernel_timer_name(int id);                 /* "k_candidates", ...; NULL outside 0 .. PLAT_KT_COUNT-1 */
int plat_kernel_times(plat_ctx* ctx, double* out_ms, int64_t* out_launches);   /* [syncs] */
/* ONE kernel timed while the profile is off (id < 0: none): two events per launch of that kernel, nothing else changes -- how bench.py times the
 * line's roofline kernel INSIDE its timed region, next to whatever shares the chip with it there.                                              */
int plat_kernel_timer_only(plat_ctx* ctx, int id);

/* ---- a1: fastAlignmentRoutine, score only -------------------------------------------------------
 * Replaces  int fastAlignmentRoutine(seq1, seq2, qual2, len1, len2, gapextend, nucprior,
 *                                    localgapopen, aln1=NULL, aln2=NULL, firstpos)
 *           src/c/align.h:8-10, src/c/align.c:77-586  (called from calign.pyx:232,258)
 * for n independent DP instances in padded rows:
 *   hap_slices [n][lmax+15], reads [n][lmax], quals [n][lmax], gapopen [n][lmax+15], len2 [n]
 *   (7 <= len2[i] <= lmax; rows need no particular alignment).  out_score [n].
 * The returned score is bit-identical to the reference's (8 x int16 wrapping lanes).          */
int plat_dp_batch(plat_ctx* ctx, int n, int lmax,
                  const uint8_t* hap_slices, const uint8_t* reads, const uint8_t* quals,
                  const uint8_t* gapopen, const int32_t* len2, int gapextend, int nucprior,
                  int32_t* out_score, void* stream);

/* ---- a3..a10: Haplotype.alignReads / alignSingleRead for whole windows -------------------------
 * Replaces, for every haplotype of every window in the batch,
 *   cdef double* Haplotype.alignReads(individualIndex, start,end, badStart,badEnd, brokenStart,
 *        brokenEnd, useMapQualCap)                       chaplotype.pxd:44, chaplotype.pyx:306-377
 *   cdef double  Haplotype.alignSingleRead(read, useMapQualCap)            chaplotype.pyx:379-384
 * including hash_sequence_multihit / hashReadForMapping / mapAndAlignReadToHaplotype
 * (calign.pyx:94-272), annotateWithGapOpen (chaplotype.pyx:552-590) and the score -> log
 * likelihood transform (chaplotype.pyx:621-676, standard mode).
 *
 * Ragged (CSR) layout, all arrays in device memory:
 *   windows  w in [0,n_windows):  haplotypes  win_hap_begin[w]..win_hap_begin[w+1]
 *                                 reads       win_read_begin[w]..win_read_begin[w+1]
 *                                 win_start[w], win_end[w] = Haplotype.startPos / endPos
 *                                 win_flank[w]             = Haplotype.endBufferSize
 *                                 pair_off[w]  = offset of the window's output block (int64)
 *   haplotypes h: bytes hap_seq[hap_off[h] .. hap_off[h+1])   (Haplotype.cHaplotypeSequence)
 *   reads r (reference order good -> bad -> brokenMates per sample, chaplotype.pyx:341-373):
 *       seq/qual bytes read_seq|read_qual[read_off[r] .. read_off[r+1]),
 *       read_pos[r], read_end[r] (cAlignedRead.pos/.end), read_mapq[r], read_flags[r] (bitFlag),
 *       read_kind[r] 0 = reads, 1 = badReads, 2 = brokenMates.
 * Output: out_loglik[pair_off[w] + hl*R_w + rl] for local haplotype hl, local read rl -- exactly
 * the concatenation of the per-haplotype likelihoodCache arrays without the 999 terminator
 * (0.0 for QCFail / overlap<7 reads, chaplotype.pyx:343-346).  out_score (optional, may be NULL)
 * receives the integer alignment score (-1 for skipped reads).
 * Options: calc_flank_score (--calculateFlankScore, runner.py:559, default 0): when 1 every DP runs in
 * the reference's traceback mode (align.c:96,345-365,494-577) and calculateFlankScore (align.c:593-644)
 * is subtracted as calign.pyx:235-245,261-264 do; needs win_flank > 0 (the reference dereferences a NULL
 * alignment buffer otherwise) else PLAT_ERR_UNSUPPORTED.  use_mapq_cap (HLATyping mode) must be 0 in this
 * version (PLAT_ERR_UNSUPPORTED otherwise).
 * [syncs] once internally (job-count read-back).                                                 */
typedef struct plat_window_batch {
    int32_t n_windows, n_haps, n_reads, _pad;
    const int32_t* win_hap_begin;   /* [n_windows+1] */
    const int32_t* win_read_begin;  /* [n_windows+1] */
    const int32_t* win_start;       /* [n_windows] */
    const int32_t* win_end;         /* [n_windows] */
    const int32_t* win_flank;       /* [n_windows] */
    const int64_t* pair_off;        /* [n_windows+1] */
    const uint8_t* hap_seq;         /* blob */
    const int64_t* hap_off;         /* [n_haps+1] */
    const uint8_t* read_seq;        /* blob */
    const uint8_t* read_qual;       /* blob, same offsets */
    const int64_t* read_off;        /* [n_reads+1] */
    const int32_t* read_pos;        /* [n_reads] */
    const int32_t* read_end;        /* [n_reads] */
    const uint8_t* read_mapq;       /* [n_reads] */
    const int32_t* read_flags;      /* [n_reads] */
    const uint8_t* read_kind;       /* [n_reads] */
} plat_window_batch;

typedef struct plat_align_stats {
    int64_t n_pairs;          /* read x haplotype pairs in the batch                                  */
    int64_t n_pairs_aligned;  /* pairs not skipped by the QCFail / overlap<7 rule                     */
    int64_t n_dp_launched;    /* banded DPs the device actually ran                                    */
    int64_t n_dp_reference;   /* fastAlignmentRoutine calls the reference would have made             */
    int64_t cells_reference;  /* sum over those calls of 16*len2 (the GCUPS numerator)   */
    int64_t cells_launched;   /* sum over launched DPs of 16*len2                                      */
    int64_t n_seed_fallback;  /* pairs whose candidate diagonals needed the full vote (not provably unique)   */
    int64_t _reserved;
} plat_align_stats;

int plat_align_window_batch(plat_ctx* ctx, const plat_window_batch* batch, int calc_flank_score,
                            int use_mapq_cap, double* out_loglik, int32_t* out_score,
                            plat_align_stats* out_stats /* host, may be NULL */, void* stream);

/* Asynchronous variant: the same work, but NOTHING is read back and the call never waits for the GPU
 * (plat_align_window_batch reads two small counters blocks back: after validation, to size its scratch
 * buffers, and after seeding, to size the DP launch -- ~55 us of idle GPU per call).  The caller, who
 * built the offset arrays, states the sizes up front:
 *   max_hap_len / max_read_len / max_reads_per_window  upper bounds over the batch
 *   n_pairs = pair_off[n_windows] (exact); hap_blob_len >= hap_off[n_haps]; read_blob_len >= read_off[n_reads]
 *   extra_jobs_cap  capacity for candidate DPs beyond one per pair (0 = n_pairs/4 + 4096)
 * The device checks the hints against the batch; errors that the synchronous call returns directly
 * (PLAT_ERR_BAD_INPUT, _HAP_TOO_LONG, ..., plus PLAT_ERR_BAD_HINTS and PLAT_ERR_OVERFLOW when
 * extra_jobs_cap was too small) are returned by the next plat_stream_sync(ctx, stream) instead; outputs of
 * a refused batch are undefined.  Scratch buffers grow on the host as before (hipMalloc may synchronise
 * the first time a larger batch is seen).                                                          */
typedef struct plat_batch_hints {
    int32_t max_hap_len, max_read_len, max_reads_per_window, _pad;
    int64_t n_pairs, hap_blob_len, read_blob_len, extra_jobs_cap;
} plat_batch_hints;

int plat_align_window_batch_async(plat_ctx* ctx, const plat_window_batch* batch, const plat_batch_hints* hints,
                                  int calc_flank_score, int use_mapq_cap, double* out_loglik,
                                  int32_t* out_score, void* stream);

/* ---- a11/a12: DiploidGenotype.calculateDataLikelihood + Population.setup ------------------------
 * Replaces  cdef double DiploidGenotype.calculateDataLikelihood(...)   cgenotype.pxd:14, .pyx:131-189
 *           cdef void   Population.setup(...)  (likelihood part)     cpopulation.pyx:283-309
 * Consumes the array written by plat_align_window_batch.  Individuals: the reads of window w are
 * segmented per individual by seg_read_begin[(w*n_ind + i)] .. [+1] (absolute read indices, must
 * tile win_read_begin) and seg_n_good[w*n_ind+i] = number of `reads` (good) entries
 * (bamReadBuffer.reads.windowEnd - windowStart, cpopulation.pyx:286).
 * Genotypes of a window are all unordered haplotype pairs (i<=j) in the order of
 * generateAllGenotypesFromHaplotypeList (cgenotype.pyx:193-218); G_w = H_w(H_w+1)/2;
 * gl_off[w] = offset of the window's block in units of genotypes*individuals:
 *   out_gl  [gl_off[w] + i*G_w + g]  = Population.genotypeLikelihoods[i][g] (rescaled, max 1.0)
 *   out_logl[same index]             = raw log-likelihood from calculateDataLikelihood
 *   out_gof [gl_off[w] + g*n_ind + i]= Population.goodnessOfFitValues[g][i]
 * Sums run over reads in index order, in fp64, without FMA contraction.                          */
int plat_genotype_window_batch(plat_ctx* ctx, const plat_window_batch* batch, int n_ind,
                               const int32_t* seg_read_begin, const int32_t* seg_n_good,
                               const double* loglik, const int64_t* gl_off,
                               double* out_gl, double* out_logl, double* out_gof, void* stream);

/* ---- what follows the genotype likelihoods in a calling window -----------------
 * All arrays are device pointers.  Windows/haplotypes/genotypes are indexed as in
 * plat_genotype_window_batch: win_hap_begin[n_windows+1]; gl[gl_off[w] + i*G_w + g]; n_reads[w*n_ind+i]
 * = seg_n_good (Population.nReads, cpopulation.pyx:286-287).  max_haps_per_window bounds H_w (LDS size); a window with more haplotypes is REFUSED: its out_iters and its out_call entries are -1, nothing else of it is written, and the next plat_stream_sync on this context returns PLAT_ERR_INVALID (no silent partial result).
 *
 * plat_em_window_batch replaces  cdef void Population.call(maxIters, ...)   cpopulation.pyx:678-703
 *   (EMiteration :384-457, callGenotypes :623-676) for every window:
 *     out_freq[win_hap_begin[w] + h]   = Population.frequencies[h]
 *     out_em  [gl_off[w] + i*G_w + g]  = Population.EMLikelihoods[i][g]  (rows of individuals without
 *                                         reads are zero; the reference leaves stale values there)
 *     out_call[w*n_ind + i]            = index of the called genotype, -1 for "None" (no reads);
 *                                         use_em_likelihoods = --useEMLikelihoods (runner.py:557)
 *     out_iters[w] (optional)          = EM iterations run (max_iters = 100 at variantcaller.pyx:140)
 *
 * plat_variant_posterior_batch replaces  cdef double Population.calculatePosterior(var)  :459-594
 *   for n_vars variants: variant v lives in window var_window[v]; hap_has_var[var_mask_off[v] + h] != 0
 *   iff the variant is one of haplotype h's variants (`var in hap.variants`); prior[v] =
 *   var.calculatePrior(refFile) computed by the caller (variant.pyx:219-259), or 0.5 for flatPrior;
 *   freq = out_freq of the EM.  out_posterior[v] = the rounded phred value the reference returns.
 *
 * plat_genotype_call_batch replaces  cdef tuple computeGenotypeCallAndLikelihoods(...)  vcfutils.pyx:163-334
 *   for n_sites VCF positions x n_ind samples: site s lies in window site_window[s] and holds
 *   site_nvar[s] variants; var_in_hap[site_vih_off[s] + h*nvar + k] = varThisPosInHap[h][k];
 *   is_ref[site_ref_off[s] + h] = haplotypeIsRefAtThisPos[h]; gof as written by
 *   plat_genotype_window_batch ([g][ind]).  Outputs per (s, i), t = s*n_ind + i:
 *     out_phased[2t..]  = (phasedIndex1, phasedIndex2)
 *     out_lik[lik_off[s] + i*NL_s ..], NL_s = (nvar+1)(nvar+2)/2 marginal likelihoods in the
 *                         reference's (index1, index2 <= index1) order
 *     out4[4t..]        = genotype posterior, non-ref posterior, ref posterior, best goodness of fit
 * Sums run in the reference's order in fp64 without FMA contraction.                               */
int plat_em_window_batch(plat_ctx* ctx, int n_windows, int n_ind, int max_haps_per_window,
                         const int32_t* win_hap_begin, const int64_t* gl_off, const int32_t* n_reads,
                         const double* gl, int max_iters, int use_em_likelihoods, double* out_freq,
                         double* out_em, int32_t* out_call, int32_t* out_iters, void* stream);

int plat_variant_posterior_batch(plat_ctx* ctx, int n_vars, int n_ind, int max_haps_per_window,
                                 const int32_t* win_hap_begin, const int64_t* gl_off, const int32_t* n_reads,
                                 const double* gl, const double* freq, const int32_t* var_window,
                                 const int64_t* var_mask_off, const uint8_t* hap_has_var, const double* prior,
                                 double* out_posterior, void* stream);

int plat_genotype_call_batch(plat_ctx* ctx, int n_sites, int n_ind, const int32_t* win_hap_begin,
                             const int64_t* gl_off, const double* gl, const double* gof, const double* freq,
                             const int32_t* site_window, const int32_t* site_nvar, const int64_t* site_vih_off,
                             const int64_t* site_ref_off, const int32_t* var_in_hap, const int32_t* is_ref,
                             const int64_t* lik_off, int32_t* out_phased, double* out_lik, double* out4,
                             void* stream);

/* ---- the HapScore of the INFO field ----------------------------------------------
 * Replaces  cdef int computeHaplotypeScore(list genotypes)                 vcfutils.pyx:1076-1114
 * together with the state it reads: DiploidGenotype.hap1Like / hap2Like, which calculateDataLikelihood resets
 * and refills on every call (cgenotype.pyx:148-161), so that after Population.setup they hold, per haplotype,
 * the sum over the reads of the LAST individual with reads of log10E * likelihood (read order, fp64).
 * Arguments as for plat_genotype_window_batch (loglik = the array written by plat_align_window_batch);
 * max_haps_per_window bounds H_w (LDS size).
 *   out_hap_like[h]  (optional, n_haps) = that sum for haplotype h of the batch
 *   out_hap_score[w] = the value vcfINFO stores as INFO['HapScore'] for every variant of window w          */
int plat_haplotype_score_batch(plat_ctx* ctx, const plat_window_batch* batch, int n_ind, int max_haps_per_window,
                               const int32_t* seg_read_begin, const int32_t* seg_n_good, const double* loglik,
                               double* out_hap_like, int32_t* out_hap_score, void* stream);

/* ---- VariantCandidateGenerator ------------------------------------------------
 * Replaces  VariantCandidateGenerator.addCandidatesFromReads(readStart, readEnd)   variant.pyx:722-743
 *           (getVariantCandidatesFromSingleRead :614-720, getSnpCandidatesFromReadSegment :529-612)
 * for the reads of n_regions regions.  Region g: reference bytes ref_seq[ref_off[g]..ref_off[g+1]) =
 * contig[ref_seq_start[g] ...) (the generator's cached pyRefSeq, region +- 2000, variant.pyx:486-489),
 * contig_len[g] = FastaFile SeqLength (deletions read the contig through getSequence, which clamps to it);
 * read r belongs to region read_region[r]: bases/qualities at read_off[r]..read_off[r+1], read_pos,
 * read_flags (bitFlag: QCFail reads are skipped), CIGAR as (op, length) int16 pairs
 * cigar[2*cig_off[r] .. 2*cig_off[r+1]).  Options: minFlank, minBaseQual, genSNPs, genIndels
 * (runner.py: 10, 20, 1, 1).
 * Output: read r owns records out_rec[5*(r*max_per_read + k)], k < out_count[r], in the reference's
 * emission order: {refPos, nRemoved, nAdded, offset of the removed bases in ref_seq (or -1), offset of
 * the added bases in read_seq (or -1)}.  out_status[r] = 0, PLAT_ERR_OVERFLOW (more than max_per_read;
 * out_count[r] is the number needed) or PLAT_ERR_BAD_INPUT (the read reaches outside the reference
 * window that was handed over).  Merging equal variants (addVariantToList :499-527) is left to the caller. */
typedef struct plat_candidate_batch {
    int32_t n_regions, n_reads;
    const uint8_t* ref_seq;
    const int64_t* ref_off;          /* [n_regions+1] */
    const int32_t* ref_seq_start;    /* [n_regions] */
    const int32_t* contig_len;       /* [n_regions] */
    const uint8_t* read_seq;
    const uint8_t* read_qual;
    const int64_t* read_off;         /* [n_reads+1] */
    const int32_t* read_pos;         /* [n_reads] */
    const int32_t* read_flags;       /* [n_reads] */
    const int16_t* cigar;            /* (op, len) pairs */
    const int32_t* cig_off;          /* [n_reads+1], in pairs */
} plat_candidate_batch;

int plat_candidates_batch(plat_ctx* ctx, const plat_candidate_batch* batch, int min_flank, int min_base_qual,
                          int gen_snps, int gen_indels, int max_per_read, const int32_t* read_region,
                          int32_t* out_rec, int32_t* out_count, int32_t* out_status, void* stream);

/* The dictionary step behind the scan and the per-sample support filter, on the device:
 * Replaces  VariantCandidateGenerator.addVariantToList (variant.pyx:499-527: equal records merge, supporting reads count) and
 *           `computeVariantReadSupportFrac(v, buffer) >= minVarFreq or v.nAdded != v.nRemoved`  (variantcaller.pyx:456-467,
 *           variantFilter.pyx:359-373 over ReadArray.countReadsCoveringRegion, cwindow.pyx:176-206)
 * for n_scans scans (a scan = the `reads` of one sample of one region) of a batch whose records plat_candidates_batch wrote:
 * the reads of scan g are [scan_read_begin[g], scan_read_begin[g+1]) (sorted by position; read_end = cAlignedRead.end),
 * scan_longest[g] = ReadArray.getLengthOfLongestRead().  Output per scan (out_n[2g] counts only when out_n[2g+1] == 0): out_n[2g] candidates that pass, unordered, 8 ints each at
 * out_cand[8*(g*cap_per_scan + i)]: {id of the first record with this content (read index * max_per_read + k: sort by it for
 * the dictionary's order), reads showing it, reads covering its position, then the record's five fields}.  out_n[2g+1] = 0,
 * PLAT_ERR_BAD_INPUT (a read outside its reference window, or read pointers out of order: the reference raises),
 * PLAT_ERR_OVERFLOW (more than 6144 distinct records or more than cap_per_scan candidates: merge this scan on the host) or
 * -(2^20 + n) when a read has n > max_per_read records (scan again with room for n).                                        */
int plat_candidates_merge_batch(plat_ctx* ctx, const plat_candidate_batch* batch, const int32_t* read_end, int n_scans,
                                const int32_t* scan_read_begin, const int32_t* scan_longest, int max_per_read,
                                const int32_t* rec, const int32_t* count, const int32_t* status, double min_var_freq,
                                int cap_per_scan, int32_t* out_cand, int32_t* out_n, void* stream);

/* plat_unpack_reads for many pieces in ONE launch: piece k = n packed bytes at src (device memory: uploaded, or resident) expanded to
 * out_seq / out_qual [dst, dst + n).  `pieces` is device memory.  The exceptions of all pieces follow in one pass: exc_index holds byte
 * indices into out_seq / out_qual (ascending not required).                                                                        */
typedef struct plat_unpack_piece { const uint8_t* src; int64_t dst, n; } plat_unpack_piece;
int plat_unpack_reads_pieces(plat_ctx* ctx, int n_pieces, int64_t max_piece_bytes, const plat_unpack_piece* pieces, uint8_t* out_seq, uint8_t* out_qual,
                             int64_t total_bytes, int64_t n_exc, const int64_t* exc_index, const uint8_t* exc_base, const uint8_t* exc_qual, void* stream);

/* The scan on 2-bit base codes (round 6).  A code is (ASCII >> 1) & 3: A 0, C 1, T 2, G 3 (N counts as 3); base i of a blob sits at bits
 * 2 (i & 15) of dword i >> 4.  Equal bytes have equal codes, so the mismatch scan of plat_candidates_batch can compare 32 bases per 64-bit word
 * and look at bytes only where codes differ -- the same records as long as a position with equal codes and different bytes can only be one the
 * reference's loop ignores: true when the reads hold A, C, G, T, N only (THE CALLER'S PROMISE for the read blob; what a PLAT_READS_PACKED
 * table expands to unless an exception carries another byte) and for every reference region without other bytes (found here: the others
 * are scanned byte by byte).
 *   plat_unpack_reads_pieces_codes  = plat_unpack_reads_pieces + out_codes[(total_bytes + 15) / 16 + 8] (zeroed and filled by the call)
 *   plat_ref_codes                  codes of the reference blob (n_bytes = ref_off[n_regions]; out_codes[(n_bytes + 15) / 16 + 8]) and
 *                                   out_irregular[g] = 1 where region g holds a byte other than A, C, G, T, N
 *   plat_candidates_batch_codes     = plat_candidates_batch on them (both code buffers 8-byte aligned)                                       */
int plat_unpack_reads_pieces_codes(plat_ctx* ctx, int n_pieces, int64_t max_piece_bytes, const plat_unpack_piece* pieces, uint8_t* out_seq, uint8_t* out_qual,
                                   uint32_t* out_codes, int64_t total_bytes, int64_t n_exc, const int64_t* exc_index, const uint8_t* exc_base,
                                   const uint8_t* exc_qual, void* stream);
int plat_ref_codes(plat_ctx* ctx, int n_regions, const uint8_t* ref_seq, const int64_t* ref_off, int64_t n_bytes, uint32_t* out_codes,
                   int32_t* out_irregular, void* stream);
int plat_candidates_batch_codes(plat_ctx* ctx, const plat_candidate_batch* batch, const uint32_t* read_codes, const uint32_t* ref_codes,
                                const int32_t* ref_irregular, int min_flank, int min_base_qual, int gen_snps, int gen_indels, int max_per_read,
                                const int32_t* read_region, int32_t* out_rec, int32_t* out_count, int32_t* out_status, void* stream);

/* n pieces of device memory copied into one blob in ONE launch: piece k = n bytes at src -> dst_blob[dst, dst + n) (any alignment).
 * `pieces` is device memory (the struct of plat_unpack_reads_pieces).                                                              */
int plat_copy_pieces(plat_ctx* ctx, int n_pieces, int64_t max_piece_bytes, const plat_unpack_piece* pieces, uint8_t* dst_blob, void* stream);

/* ---- a chunk's read table from read tables that are resident on the device --------------------------------------------------------
 * Replaces  the loader's copy of a bamReadBuffer's reads into the arrays the kernels above take (no reference counterpart: the
 *           reference walks cAlignedRead pointers, cwindow.pyx:485-595).
 * n_tables tables, table t described by desc[t] (device memory): its seven per-read arrays on the device, n reads, and where it goes
 * in the destination: first read index, first byte of its bases in the chunk blob, first CIGAR pair, and the scan id written to
 * dst_region[] for its reads (or -1: none).  Writes dst_off (+ byte base), dst_pos, dst_end, dst_mapq, dst_flags, dst_cig_off (+ pair
 * base), dst_cigar, dst_region and the closing entries dst_off[N] = total_bytes, dst_cig_off[N] = total_pairs, one zero CIGAR pair.    */
typedef struct plat_table_desc {
    const int64_t* off; const int32_t* pos; const int32_t* end; const uint8_t* mapq; const int32_t* flags; const int16_t* cigar; const int32_t* cig_off;
    int32_t n, scan; int64_t first_read, first_byte, first_pair;
} plat_table_desc;
int plat_concat_read_tables(plat_ctx* ctx, int n_tables, int max_reads_per_table, const plat_table_desc* desc, int64_t* dst_off, int32_t* dst_pos,
                            int32_t* dst_end, uint8_t* dst_mapq, int32_t* dst_flags, int32_t* dst_cig_off, int16_t* dst_cigar,
                            int32_t* dst_region, int64_t n_total_reads, int64_t total_bytes, int64_t total_pairs, void* stream);

/* ---- candidates -> variants -> calling windows -> haplotypes -> the window batch, on the device ----------
 * Replaces, for regions with ONE sample, what callVariantsInRegion does between the candidate generator and Population.setup:
 *   sorted(varCandGen.getCandidates())                                             variantcaller.pyx:456-470, variant.pyx:282-363
 *   leftNormaliseIndel(v, refFile, maxReadLength)                                  platypusutils.pyx:806-931
 *   filterVariants(sorted(...), refFile, maxReadLength, minSupport, maxDiff, ...)  variantFilter.pyx:98-171
 *   WindowGenerator.getBunchesOfVariants / WindowsAndVariants                      window.py:49-127,140-238
 *   bamReadBuffer.setWindowPointers                                                cwindow.pyx:176-264,655-689
 *   getFilteredHaplotypes, the `nVars <= log2(maxHaplotypes)` branch: every combination of the window's variants that
 *   isHaplotypeValid accepts                                                       variantFilter.pyx:377-441, platypusutils.pyx:735-802
 *   Haplotype.__init__ / getMutatedSequence (the haplotypes' bytes)                chaplotype.pyx:127-191,397-449
 *   mergeHaplotypes' sorted(haplotypes) (the ORDER; two equal sequences are left to the caller)    variantcaller.pyx:325-383
 * Input: the candidates plat_candidates_merge_batch left on the device (cand / cand_n = its out_cand / out_n, cap_per_scan), the
 * reference windows and the read table the scan saw, and per region g its three read arrays inside that table
 * (tab_begin/tab_n/tab_longest[3g + k], k = 0 reads, 1 badReads, 2 brokenMates; broken_mate_pos[i] = mate position of read
 * tab_begin[3g+2] + i, indexed from broken_base).
 * Output, per region g (all device memory of the caller):
 *   hdr[8g..]   {status, n_variants, n_windows, candidate records, bytes used of the added-bases blob, why (status != 0: 1 capacity /
 *               an exception, 2 dictionary too large to replay, 4 windows, 5 the merge's own verdict, 6 cap_vars / cap_windows / cap_added too
 *               small for this region), dictionaries replayed, 0}; status 0, or
 *               PLAT_SB_HOST: this region needs the caller's own code (more candidates / variants / windows than the capacities, an
 *               indel at the edge of its reference window, an order that depends on a Python dictionary, an exception the reference
 *               would raise, ...) -- nothing else of the region is valid then.
 *   variants    [g*cap_vars + i]: var_pos, var_nrem, var_nadd, var_support (nSupportingReads), var_bam_min, var_bam_max
 *               (bamMinPos / bamMaxPos), var_rem_pos (contig coordinate of the removed bases), var_add_off (offset of the added
 *               bases in added[g*cap_added ..]); sorted as the reference's list is.
 *   windows     [g*cap_windows + k]: win_start, win_end, win_var_first, win_var_n (its variants = a run of the region's list),
 *               win_flags (PLAT_SBW_*), win_ptrs[6 * ..] = {reads begin, end, badReads begin, end, brokenMates begin, end} (indices
 *               inside the region's arrays), win_n_haps (reference haplotype included), win_batch (index in the window batch or -1).
 * and, for the windows with win_flags == 0, a complete window batch in the arrays of `wb` (the plat_window_batch the likelihood
 * kernels take: windows in (region, window) order, haplotypes sorted by sequence, reads good -> bad -> brokenMates as indices
 * read_src[] into the read table + read_kind[]; read_off from the table's lengths; seg_begin / seg_n_good / gl_off for one sample)
 * plus b_hap_mask[h] = which of its window's variants haplotype h carries (bit i = variant win_var_first + i).
 *   totals[16]  {windows, haplotypes, reads, pairs, genotype likelihoods, haplotype bytes, read bytes, longest haplotype, most reads
 *               of a window, most haplotypes of a window, overflow (a batch capacity was too small: nothing of the batch is valid), ...}
 * No host round trip inside; the caller reads hdr / totals back once. */
#define PLAT_SB_HOST 1
#define PLAT_SBW_SKIP 1          /* no reads / too many reads / skipDifficultWindows: the loop does not call this window */
#define PLAT_SBW_HOST 2          /* the caller prepares this window itself (greedy haplotype filter, filterVariantsByCoverage, an exception) */
#define PLAT_SBW_DUPLICATE 4     /* in the batch, but two of its haplotypes have the same sequence and an indel among their variants (or
                                  * agree on more bytes than the device looks at): mergeHaplotypes / the order is the caller's */
typedef struct plat_stage_b_options {
    int32_t minReads, maxSize, mergeClusteredVariants, maxVarDist, minVarDist, largeWindows, maxVariants, maxHaplotypes;
    int32_t filterVarsByCoverage, skipDifficultWindows;
    double maxReads;
} plat_stage_b_options;
typedef struct plat_stage_b_in {
    int32_t n_regions, cap_per_scan;
    const int32_t* cand; const int32_t* cand_n;
    /* for the replay of the candidate generator's Python-2 dictionaries where an order depends on them: the scan's records (out_rec of
     * plat_candidates_batch) and per region hash(refName) as CPython 2.7 computes it for the contig's name; either NULL: such a region is
     * flagged PLAT_SB_HOST instead.  (The distinct records themselves are read from the context's merge table: call this right behind
     * plat_candidates_merge_batch, on the same context and stream.) */
    const int32_t* cand_rec; const int64_t* region_name_hash;
    const uint8_t* ref_seq; const int64_t* ref_off; const int32_t* ref_seq_start; const int32_t* contig_len;
    const int32_t* region_start; const int32_t* region_end; const int32_t* region_rlen;    /* [n_regions]; rlen: options.rlen as the loop carries it */
    const uint8_t* read_seq; const int64_t* read_off; const int32_t* read_pos; const int32_t* read_end;
    const int32_t* tab_begin; const int32_t* tab_n; const int32_t* tab_longest;            /* [3 * n_regions] */
    const int32_t* broken_mate_pos; int32_t broken_base;
    /* capacities */
    int32_t cap_vars, cap_windows, cap_added;
    int32_t cap_batch_windows, cap_batch_haps, cap_batch_reads; int64_t cap_hap_bytes;
} plat_stage_b_in;
typedef struct plat_stage_b_out {
    int32_t* hdr;
    int32_t* var_pos; int32_t* var_nrem; int32_t* var_nadd; int32_t* var_support; int32_t* var_bam_min; int32_t* var_bam_max;
    int32_t* var_rem_pos; int32_t* var_add_off; uint8_t* added;
    int32_t* win_start; int32_t* win_end; int32_t* win_var_first; int32_t* win_var_n; int32_t* win_flags; int32_t* win_ptrs;
    int32_t* win_n_haps; int32_t* win_batch;
    /* the window batch */
    int32_t* b_hap_begin; int32_t* b_read_begin; int32_t* b_start; int32_t* b_end; int32_t* b_flank;       /* [cap_batch_windows (+1)] */
    int64_t* b_pair_off; int64_t* b_gl_off; int32_t* b_seg_begin; int32_t* b_n_good;                          /* [cap_batch_windows (+1)] */
    int64_t* b_hap_off; uint32_t* b_hap_mask; uint8_t* b_hap_seq;                                            /* [cap_batch_haps + 1], [cap_hap_bytes] */
    int64_t* b_read_off; int32_t* b_read_src; uint8_t* b_read_kind;                                           /* [cap_batch_reads (+1)] */
    int64_t* totals;                                                                                           /* [16] */
    int32_t* scratch;                                                                                          /* [56 * n_regions * cap_windows + 48 * n_regions + 64] */
} plat_stage_b_out;
int plat_stage_b_batch(plat_ctx* ctx, const plat_stage_b_in* batch, const plat_stage_b_options* options,
                       const plat_stage_b_out* out, void* stream);

/* ---- read QC / trimming ----------------------------------------------------------------------------
 * Replaces  cdef int checkAndTrimRead(theRead, theLastRead, ...)   cwindow.pyx:332-481
 * as driven by bamReadBuffer.addReadToBuffer (:560-595) for whole streams of reads: read r's
 * `theLastRead` is read r-1 when stream_of[r-1] == stream_of[r].  read_qual is modified in place (trimmed
 * bases get quality 0), read_flags get BAM_FQCFAIL (512) where the reference sets it.
 *   out_ok[r]      the function's return value (1: goes to `reads`, 0: goes to `badReads`)
 *   out_reason[r]  -1 accepted, else the filteredReadCountsByType slot that was bumped (cwindow.pyx:40-46:
 *                  0 LOW_QUAL_BASES, 1 UNMAPPED_READ, 2 MATE_UNMAPPED, 3 MATE_DISTANT, 4 SMALL_INSERT,
 *                  5 DUPLICATE, 6 LOW_MAP_QUAL) or 7 for a secondary alignment
 * filter_* = 0 switches a filter off (the reference uses a counter value of -1 for that).        */
typedef struct plat_readqc_batch {
    int32_t n_reads, _pad;
    uint8_t* read_qual;              /* raw phred, in/out */
    const int64_t* read_off;         /* [n_reads+1] */
    const int32_t* read_pos;
    const uint8_t* read_mapq;
    int32_t* read_flags;             /* bitFlag, in/out */
    const int16_t* chrom_id;
    const int16_t* mate_chrom_id;
    const int32_t* insert_size;
    const int32_t* mate_pos;
    const int16_t* cigar;            /* (op, len) pairs */
    const int32_t* cig_off;          /* [n_reads+1], in pairs */
    const int32_t* stream_of;        /* [n_reads] id of the read's stream (sample buffer) */
} plat_readqc_batch;

typedef struct plat_readqc_options {
    int32_t min_good_qual_bases, min_map_qual, min_base_qual;           /* minGoodQualBases 20, minMapQual 20, minBaseQual 20 */
    int32_t trim_overlapping, trim_adapter, trim_read_flank, trim_soft_clipped;   /* 1, 1, 0, 1 */
    int32_t filter_mate_unmapped, filter_mate_distant, filter_small_insert, filter_duplicates;
} plat_readqc_options;

int plat_read_qc_batch(plat_ctx* ctx, const plat_readqc_batch* batch, const plat_readqc_options* options,
                       int32_t* out_ok, int32_t* out_reason, void* stream);

/* ---- read statistics of the VCF INFO field ---------------------------------------
 * Replaces the per-variant loop over a window's reads in  cdef dict vcfINFO(...)   vcfutils.pyx:1300-1390
 * (readOverlapsVariant :901-913, readQualIsGoodVariantPosition :917-943, variantSupportedByRead :961-1072).
 * Variant v lies in window var_window[v]; sample i of window w owns the good reads
 * [good_begin[w*n_ind+i], good_end[..]) and the bad reads [bad_begin[..], bad_end[..]) of one read table
 * (bases, raw phred qualities, pos, end, mapq, bitFlag, CIGAR pairs); var_in_genotype[v*n_ind+i] =
 * `variant in genotypeCalls[i]`; the added bases of v at var_added[var_added_off[v] ..+var_n_added[v]).
 * Options: badReadsWindow (runner.py: 11), countOnlyExactIndelMatches (0).
 * Output: out_counts[16*v + k]: 0 TC, 1 TC_bad, 2 TR, 3 TC_ab, 4 TR_ab, 5 NR_sb, 6 NF_sb, 7 TCR, 8 TCF,
 * 9 TCR_sb, 10 TCF_sb, 11 NR, 12 NF, 13 nGoodReads, 14 nBadReads, 15 sum of mapq^2 (the reference keeps this
 * sum in a float: identical below 2^24); out_per_sample[(v*n_ind+i)*2 + {0,1}] = nReadsThisSample,
 * nVarReadsThisSample; out_minq[minq_off[v] + k], k < out_nminq[v]: the MMLQ window minima in read order
 * (room for the window's good reads).  The INFO values follow by arithmetic the caller keeps
 * (MQ = sqrt(sum/(TC+TC_bad)), BRF, MMLQ = median, ABPV / SbPval from the _ab / _sb counts).      */
typedef struct plat_infostats_batch {
    int32_t n_vars, n_ind;
    const int32_t* var_window;       /* [n_vars] */
    const int32_t* var_pos;          /* Variant.refPos */
    const int32_t* var_bam_min;      /* Variant.bamMinPos */
    const int32_t* var_bam_max;      /* Variant.bamMaxPos */
    const int32_t* var_n_added;
    const int32_t* var_n_removed;
    const uint8_t* var_added;
    const int64_t* var_added_off;    /* [n_vars] */
    const uint8_t* var_in_genotype;  /* [n_vars*n_ind] */
    const int64_t* minq_off;         /* [n_vars] */
    const int32_t* good_begin;       /* [n_windows*n_ind] */
    const int32_t* good_end;
    const int32_t* bad_begin;
    const int32_t* bad_end;
    const uint8_t* read_seq;
    const uint8_t* read_qual;
    const int64_t* read_off;         /* [n_reads+1] */
    const int32_t* read_pos;
    const int32_t* read_end;
    const uint8_t* read_mapq;
    const int32_t* read_flags;
    const int16_t* cigar;
    const int32_t* cig_off;          /* [n_reads+1] */
} plat_infostats_batch;

int plat_variant_read_stats_batch(plat_ctx* ctx, const plat_infostats_batch* batch, int bad_reads_window,
                                  int count_only_exact_indel_matches, int64_t* out_counts, int32_t* out_per_sample,
                                  int32_t* out_minq, int32_t* out_nminq, void* stream);

/* ---- the loops of two INFO fields, from the counts above ------------------------------------------------
 * Replaces, per variant, the work inside
 *     computeAlleleBiasPValue / computeStrandBiasPValue   vcfutils.pyx:1156-1222   (INFO ABPV -- used by the alleleBias filter -- and SbPval)
 *     betaBinomialCDF, threeFTwo, logBetaFunction          platypusutils.pyx:178-315
 *     sorted(minBaseQualsInWindow)[n // 2]                  vcfutils.pyx:1390-1399   (INFO MMLQ)
 * that is a LOOP: the hypergeometric series 3F2 (|k - n + 1| terms), the sums of log-factorials (a table of the reference's own
 * logFactorial, made with the HOST's libm when the context is created), the median.  What is left to the caller are the three libm
 * calls of a CDF on these terms, so that the value has the bits the host's code gives:
 *     cdf = max(1e-30, 1 - exp((terms[1] + log(terms[2])) - terms[3]))
 * counts / minq_off / minq / n_minq: the outputs of plat_variant_read_stats_batch, still on the device.
 * out_terms[8 v + 0..3] allele bias, [8 v + 4..7] strand bias: [0] = 0: the function returns [1] as it is (its early exits; the allele
 * bias value is then final, min(p, 1 - p) included); 1: [1..3] are the terms of betaBinomialCDF(k, n, alpha, beta) above (allele bias:
 * the caller still takes min(p, 1 - p)); 2: an argument of logFactorial beyond the table (4096): the caller computes the field itself.
 * out_mmlq[v] = the median as the reference takes it, 100 when the variant has no entries.                                       */
int plat_variant_info_batch(plat_ctx* ctx, int n_vars, const int64_t* counts, const int64_t* minq_off, const int32_t* minq,
                            const int32_t* n_minq, double* out_terms, int32_t* out_mmlq, void* stream);

/* ---- read tables that crossed the link at one byte per base --------------------------------------------
 * The loader's decode loop (htslibWrapper.pyx:330-370: 4-bit BAM code -> letter, quality byte copied, one pass over every base)
 * may write ONE byte per base instead of two: bits 0..1 = (letter >> 1) & 3 (A 0, C 1, T 2, G 3), bits 2..7 = quality 0..63
 * (PLAT_READS_PACKED of include/platypus_caller.h).  plat_unpack_reads expands n_bytes such bytes to the ASCII base and raw
 * quality arrays every other entry point takes (out_seq / out_qual [n_bytes]), then patches the n_exc exceptions (bases other
 * than A/C/G/T, qualities above 63): out_seq[exc_index[k]] = exc_base[k], out_qual[exc_index[k]] = exc_qual[k].  All device
 * pointers; exc_* may be NULL when n_exc == 0.                                                                          */
int plat_unpack_reads(plat_ctx* ctx, int64_t n_bytes, const uint8_t* packed, uint8_t* out_seq, uint8_t* out_qual,
                      int64_t n_exc, const int64_t* exc_index, const uint8_t* exc_base, const uint8_t* exc_qual, void* stream);

/* ---- window read slices out of a resident read table ---------------------------------------------
 * Replaces the per-window walk over  bamReadBuffer.reads / badReads / brokenMates  between the window
 * pointers (cwindow.pyx:208-264,655-689) that feeds Haplotype.alignReads: the reads of a whole region are
 * uploaded ONCE (one table: bases, qualities, offsets, pos, end, mapq, bitFlag) and the read arrays of a
 * plat_window_batch are gathered from it on the device.  Destination read d is source read src_index[d];
 * its bases / qualities go to dst_seq|dst_qual[dst_off[d] .. dst_off[d+1]) (dst_off is what the batch then
 * uses as read_off, computed by the caller from the read lengths), its pos / end / mapq / bitFlag to
 * dst_*[d].  All pointers are device pointers.                                                       */
int plat_gather_reads(plat_ctx* ctx, int64_t n_dst, const int32_t* src_index, const int64_t* dst_off,
                      const uint8_t* src_seq, const uint8_t* src_qual, const int64_t* src_off,
                      const int32_t* src_pos, const int32_t* src_end, const uint8_t* src_mapq,
                      const int32_t* src_flags, uint8_t* dst_seq, uint8_t* dst_qual, int32_t* dst_pos,
                      int32_t* dst_end, uint8_t* dst_mapq, int32_t* dst_flags, void* stream);

/* ---- a14..a18: assembleReadsAndDetectVariants ---------------------------------------------------
 * Replaces  cdef list assembleReadsAndDetectVariants(chrom, assemStart, assemEnd, refStart, refEnd,
 *                                                    readBuffers, refSeq, options)
 *           assembler.pxd:3, assembler.pyx:1429-1476
 * for n_regions independent assembly tiles.  Reads of a region are those loadBAMDataIntoGraph
 * (assembler.pyx:1391-1425) would load, in its order, QCFail reads already removed.
 *   region g: reference bytes ref_seq[ref_off[g]..ref_off[g+1]) starting at genome coordinate
 *             ref_start[g]; assembly interval [assem_start[g], assem_end[g]);
 *             reads reg_read_begin[g]..reg_read_begin[g+1].
 * Options: kmer_size (assemblerKmerSize, 15), min_qual (minBaseQual, 20), min_weight
 * (minReads*minBaseQual, 40), no_cycles (noCycles, 0).
 * Output (device): per region up to max_vars_per_region variants, in the reference's sorted()
 * order: var_count[g]; var_pos/var_nrem/var_nadd [g*max_vars+i]; removed||added bytes at
 * var_blob[g*blob_per_region + var_off[g*max_vars+i]].  status[g] = 0 or PLAT_ERR_OVERFLOW.     */
typedef struct plat_assembly_batch {
    int32_t n_regions, n_reads;
    const uint8_t* ref_seq;
    const int64_t* ref_off;          /* [n_regions+1] */
    const int32_t* ref_start;        /* [n_regions] */
    const int32_t* assem_start;      /* [n_regions] */
    const int32_t* assem_end;        /* [n_regions] */
    const int32_t* reg_read_begin;   /* [n_regions+1] */
    const uint8_t* read_seq;
    const uint8_t* read_qual;
    const int64_t* read_off;         /* [n_reads+1] */
} plat_assembly_batch;

int plat_assemble_batch(plat_ctx* ctx, const plat_assembly_batch* batch, int kmer_size, int min_qual,
                        int min_weight, int no_cycles, int max_vars_per_region, int blob_per_region,
                        int32_t* var_count, int32_t* var_pos, int32_t* var_nrem, int32_t* var_nadd,
                        int32_t* var_off, uint8_t* var_blob, int32_t* status, void* stream);
/* The same without the read-back that sizes the graphs: plat_assemble_batch learns the longest reference window, the most reads
 * and the most k-mer positions of a tile from the device and WAITS for them; a caller that built the batch knows them.  Nothing
 * is read back and the call never waits.  hints: max_ref_len >= ref_off[g+1] - ref_off[g], max_reads_per_region >=
 * reg_read_begin[g+1] - reg_read_begin[g], max_positions >= reference bytes + read bytes + 2 * reads + 2 of every tile g.  The device
 * checks them: hints that do not cover the batch give status[g] = PLAT_ERR_BAD_HINTS for EVERY tile (PLAT_ERR_BAD_INPUT for offsets
 * out of order) and no variants.                                                                     */
typedef struct plat_assembly_hints {
    int32_t max_ref_len, max_reads_per_region;
    int64_t max_positions;
} plat_assembly_hints;
int plat_assemble_batch_async(plat_ctx* ctx, const plat_assembly_batch* batch, const plat_assembly_hints* hints, int kmer_size,
                              int min_qual, int min_weight, int no_cycles, int max_vars_per_region, int blob_per_region,
                              int32_t* var_count, int32_t* var_pos, int32_t* var_nrem, int32_t* var_nadd,
                              int32_t* var_off, uint8_t* var_blob, int32_t* status, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PLATYPUS_MI355X_H */
