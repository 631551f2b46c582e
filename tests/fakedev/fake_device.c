/*
 * fake_device.c -- TEST INFRASTRUCTURE: the C ABI of include/platypus_mi355x.h implemented on the CPU with the
 * parity oracle (oracle/liborc.so), "device" memory = host memory.  It exists so that the HOST logic layered on the C ABI
 * (platypus_amd/caller.py and the native region pipeline, libplat_caller.so) can be exercised end to end by the CPU
 * test suite, where no GPU is present.  It is built by tests/conftest.py into tests/fakedev/, is never shipped with the
 * package and is never loaded by it: the product binds libplat_mi355x.so (HIP) and fails loudly without a GPU.
 *
 * Only the entry points those host layers use are functional; the rest return PLAT_ERR_UNSUPPORTED.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/platypus_mi355x.h"

#define API __attribute__((visibility("default")))

/* ---- the oracle (oracle/plat_oracle.c) ------------------------------------------------------------------------------ */
extern void orc_align_window(int nHaps, const char* hapBlob, const int* hapOff, const int* hapLen, int hapStartPos, int hapEndPos,
                             int endBuffer, int nReads, const char* seqBlob, const char* qualBlob, const int* readOff,
                             const int* readLen, const int* readPos, const int* readEnd, const unsigned char* mapq, const int* flags,
                             const unsigned char* kind, int doFlank, double* out_ll, int* out_score, long long* n_dp_total);
extern double orc_genotype_loglik(const double* arr1, const double* arr2, int same_hap, int totalReads, int nGoodReads, double* gof,
                                  double* hap1Like, double* hap2Like);
extern void orc_population_setup_ind(int nHaps, const double* ll, int totalReads, int nGoodReads, double* out_logl, double* out_gl,
                                     double* out_gof);
extern int orc_em_call(int nInd, int nHap, const int* nReads, const double* gl, int maxIters, int useEM, double* freq, double* em,
                       int* calls, double* maxChangeOut);
extern double orc_variant_posterior(int nInd, int nHap, const int* nReads, const double* gl, const double* freq,
                                    const unsigned char* hapHasVar, double prior);
extern void orc_genotype_call(int nHap, int nVar, int nIndividuals, const double* freq, const double* gl, const double* gof,
                              const int* varInHap, const int* isRef, int* phased, double* likelihoods, double* out4);
extern int orc_variant_candidates(const char* ref, int refLen, int refSeqStart, int contigLen, int nReads, const char* seq,
                                  const char* qual, const long long* read_off, const int* pos, const int* flags, const short* cigar,
                                  const int* cig_off, int minFlank, int minBaseQual, int genSNPs, int genIndels, int* rec, int maxRec);
extern void orc_variant_read_stats(int nVars, const int* varPos, const int* bamMin, const int* bamMax, const int* nAdded,
                                   const int* nRemoved, const char* addedBlob, const int* addedOff, int nInd, const int* good_begin,
                                   const int* good_end, const int* bad_begin, const int* bad_end, const unsigned char* varInGenotype,
                                   const char* seq, const char* qual, const long long* read_off, const int* pos, const int* end,
                                   const unsigned char* mapq, const int* flags, const short* cigar, const int* cig_off, int minBaseQual,
                                   int badReadsWindow, int exactIndel, long long* out, int* per_sample, int* minq, int maxq, int* nminq);
extern int orc_assemble(const char* ref, int refLen, int refStart, int assemStart, int assemEnd, int nReads, const char* seqBlob,
                        const char* qualBlob, const int* off, const int* len, int kmerSize, int minQual, int minWeightI, int noCycles,
                        int maxVars, int* out_pos, int* out_nrem, int* out_nadd, int* out_off, char* out_blob, int blobCap,
                        int* out_nnodes);
extern int orc_haplotype_score(int nHaps, const double* hapLike);

struct plat_ctx { int sticky; };

API int plat_abi_version(void) { return PLAT_ABI_VERSION; }
API const char* plat_strerror(int code) { (void)code; return "fake device (tests/fakedev): see the error code"; }
API int plat_device_count(int* n) { if (!n) return PLAT_ERR_INVALID; *n = 1; return PLAT_OK; }
API int plat_ctx_create(int device, plat_ctx** out) {
    (void)device;
    if (!out) return PLAT_ERR_INVALID;
    *out = (plat_ctx*)calloc(1, sizeof(plat_ctx));
    return PLAT_OK;
}
API int plat_ctx_destroy(plat_ctx* c) { free(c); return PLAT_OK; }
API int plat_last_hip_error(const plat_ctx* c) { (void)c; return 0; }
API int plat_malloc(plat_ctx* c, size_t n, void** out) { (void)c; *out = calloc(n + 64, 1); return *out ? PLAT_OK : PLAT_ERR_NOMEM; }
API int plat_free(plat_ctx* c, void* p) { (void)c; free(p); return PLAT_OK; }
API int plat_host_alloc(plat_ctx* c, size_t n, void** out) { return plat_malloc(c, n, out); }
API int plat_host_free(plat_ctx* c, void* p) { return plat_free(c, p); }
API int plat_memcpy_h2d(plat_ctx* c, void* d, const void* s, size_t n, void* st) { (void)c; (void)st; if (n) memcpy(d, s, n); return PLAT_OK; }
API int plat_memcpy_d2d(plat_ctx* c, void* d, const void* s, size_t n, void* st) { (void)c; (void)st; if (n) memcpy(d, s, n); return PLAT_OK; }
API int plat_memcpy_d2h(plat_ctx* c, void* d, const void* s, size_t n, void* st) { (void)c; (void)st; if (n) memcpy(d, s, n); return PLAT_OK; }
API int plat_memset(plat_ctx* c, void* d, int v, size_t n, void* st) { (void)c; (void)st; if (n) memset(d, v, n); return PLAT_OK; }
API int plat_stream_create(plat_ctx* c, void** out) { (void)c; *out = (void*)(uintptr_t)0x10; return PLAT_OK; }
API int plat_stream_destroy(plat_ctx* c, void* s) { (void)c; (void)s; return PLAT_OK; }
/* fault injection for the host-side error handling: PLAT_FAKE_FAIL_SYNC="<code>:<n>" makes the n-th plat_stream_sync of the process
 * (counted from 1, over all contexts) return <code> */
static int g_sync_calls = 0;
API int plat_stream_sync(plat_ctx* c, void* s) {
    (void)s;
    int e = c->sticky; c->sticky = 0;
    const char* f = getenv("PLAT_FAKE_FAIL_SYNC");
    const int k = __sync_add_and_fetch(&g_sync_calls, 1);
    if (f) { int code = 0, nth = 0; if (sscanf(f, "%d:%d", &code, &nth) == 2 && k == nth) return code; }
    return e;
}
API void plat_fake_reset_sync_count(void) { g_sync_calls = 0; }
API int plat_profile_enable(plat_ctx* c, int on) { (void)c; (void)on; return PLAT_OK; }
API int plat_profile_last(plat_ctx* c, plat_profile* p) { (void)c; memset(p, 0, sizeof(*p)); return PLAT_OK; }
API int plat_sync_poll_us(plat_ctx* c, int us) { (void)c; return us < 0 ? PLAT_ERR_INVALID : PLAT_OK; }
API const char* plat_kernel_timer_name(int id) { return id >= 0 && id < PLAT_KT_COUNT ? "fake" : NULL; }
API int plat_kernel_timer_only(plat_ctx* c, int id) { (void)c; return id >= PLAT_KT_COUNT ? PLAT_ERR_INVALID : PLAT_OK; }
API int plat_kernel_times(plat_ctx* c, double* ms, int64_t* n) { (void)c; (void)ms; (void)n; return PLAT_OK; }
API int plat_dp_batch(plat_ctx* c, int n, int lmax, const uint8_t* a, const uint8_t* b, const uint8_t* q, const uint8_t* g,
                      const int32_t* l, int ge, int np_, int32_t* o, void* st)
{ (void)c; (void)n; (void)lmax; (void)a; (void)b; (void)q; (void)g; (void)l; (void)ge; (void)np_; (void)o; (void)st; return PLAT_ERR_UNSUPPORTED; }
API int plat_read_qc_batch(plat_ctx* c, const plat_readqc_batch* b, const plat_readqc_options* o, int32_t* ok, int32_t* why, void* st)
{ (void)c; (void)b; (void)o; (void)ok; (void)why; (void)st; return PLAT_ERR_UNSUPPORTED; }

/* ---- Haplotype.alignReads for whole windows --------------------------------------------------------------------------- */
static int align_impl(const plat_window_batch* b, int calc_flank, double* out_ll, int32_t* out_score, plat_align_stats* st)
{
    long long ndp_total = 0, npairs = 0;
    for (int w = 0; w < b->n_windows; ++w) {
        const int h0 = b->win_hap_begin[w], nH = b->win_hap_begin[w + 1] - h0;
        const int r0 = b->win_read_begin[w], nR = b->win_read_begin[w + 1] - r0;
        if (nH <= 0 || nR < 0) continue;
        if (calc_flank && b->win_flank[w] <= 0) return PLAT_ERR_UNSUPPORTED;
        int* hapOff = (int*)malloc(sizeof(int) * (size_t)nH), *hapLen = (int*)malloc(sizeof(int) * (size_t)nH);
        const long long hbase = b->hap_off[h0];
        for (int h = 0; h < nH; ++h) {
            hapOff[h] = (int)(b->hap_off[h0 + h] - hbase);
            hapLen[h] = (int)(b->hap_off[h0 + h + 1] - b->hap_off[h0 + h]);
            if (hapLen[h] > 16384) { free(hapOff); free(hapLen); return PLAT_ERR_HAP_TOO_LONG; }
        }
        int* readOff = (int*)malloc(sizeof(int) * (size_t)(nR + 1)), *readLen = (int*)malloc(sizeof(int) * (size_t)(nR + 1));
        const long long rbase = nR ? b->read_off[r0] : 0;
        for (int r = 0; r < nR; ++r) {
            readOff[r] = (int)(b->read_off[r0 + r] - rbase);
            readLen[r] = (int)(b->read_off[r0 + r + 1] - b->read_off[r0 + r]);
            /* the device refuses what the reference would read past its buffers for */
            int skip = 0;
            if (b->read_kind[r0 + r] != 2) {
                const int os = b->win_start[w] > b->read_pos[r0 + r] ? b->win_start[w] : b->read_pos[r0 + r];
                const int oe = b->win_end[w] < b->read_end[r0 + r] ? b->win_end[w] : b->read_end[r0 + r];
                skip = (b->read_flags[r0 + r] & 512) || (oe > os ? oe - os : -1) < 7;
            }
            if (!skip && readLen[r] >= 7)
                for (int h = 0; h < nH; ++h)
                    if (hapLen[h] < readLen[r] + 15) { free(hapOff); free(hapLen); free(readOff); free(readLen); return PLAT_ERR_HAP_TOO_SHORT; }
        }
        long long ndp = 0;
        int* sc = (int*)malloc(sizeof(int) * (size_t)nH * (size_t)(nR + 1));
        orc_align_window(nH, (const char*)b->hap_seq + hbase, hapOff, hapLen, b->win_start[w], b->win_end[w], b->win_flank[w], nR,
                         (const char*)b->read_seq + rbase, (const char*)b->read_qual + rbase, readOff, readLen, b->read_pos + r0,
                         b->read_end + r0, b->read_mapq + r0, b->read_flags + r0, b->read_kind + r0, calc_flank,
                         out_ll + b->pair_off[w], sc, &ndp);
        if (out_score) memcpy(out_score + b->pair_off[w], sc, sizeof(int) * (size_t)nH * (size_t)nR);
        ndp_total += ndp; npairs += (long long)nH * nR;
        free(sc); free(hapOff); free(hapLen); free(readOff); free(readLen);
    }
    if (st) { memset(st, 0, sizeof(*st)); st->n_pairs = npairs; st->n_dp_reference = ndp_total; st->n_dp_launched = ndp_total; }
    return PLAT_OK;
}

API int plat_align_window_batch(plat_ctx* c, const plat_window_batch* b, int calc_flank, int use_mapq_cap, double* out_ll,
                                int32_t* out_score, plat_align_stats* st, void* stream)
{
    (void)stream;
    if (!c || !b) return PLAT_ERR_INVALID;
    if (use_mapq_cap) return PLAT_ERR_UNSUPPORTED;
    return align_impl(b, calc_flank != 0, out_ll, out_score, st);
}

API int plat_align_window_batch_async(plat_ctx* c, const plat_window_batch* b, const plat_batch_hints* hints, int calc_flank,
                                      int use_mapq_cap, double* out_ll, int32_t* out_score, void* stream)
{
    (void)stream;
    if (!c || !b || !hints) return PLAT_ERR_INVALID;
    if (use_mapq_cap) return PLAT_ERR_UNSUPPORTED;
    const int rc = align_impl(b, calc_flank != 0, out_ll, out_score, NULL);
    if (rc && !c->sticky) c->sticky = rc;                      /* reported by the next plat_stream_sync, as on the device */
    return PLAT_OK;
}

/* ---- Population.setup ------------------------------------------------------------------------------------------------- */
API int plat_genotype_window_batch(plat_ctx* c, const plat_window_batch* b, int n_ind, const int32_t* seg_read_begin,
                                   const int32_t* seg_n_good, const double* loglik, const int64_t* gl_off, double* out_gl,
                                   double* out_logl, double* out_gof, void* stream)
{
    (void)c; (void)stream;
    for (int w = 0; w < b->n_windows; ++w) {
        const int nH = b->win_hap_begin[w + 1] - b->win_hap_begin[w], G = nH * (nH + 1) / 2;
        const int r0 = b->win_read_begin[w], R = b->win_read_begin[w + 1] - r0;
        for (int i = 0; i < n_ind; ++i) {
            const long long s = (long long)w * n_ind + i;
            const int a = seg_read_begin[s] - r0, e = seg_read_begin[s + 1] - r0, tot = e - a;
            double* ll = (double*)malloc(sizeof(double) * (size_t)nH * (size_t)(tot + 1));
            for (int h = 0; h < nH; ++h) {
                memcpy(ll + (size_t)h * (tot + 1), loglik + b->pair_off[w] + (long long)h * R + a, sizeof(double) * (size_t)tot);
                ll[(size_t)h * (tot + 1) + tot] = 999.0;
            }
            double* logl = (double*)malloc(sizeof(double) * 3 * (size_t)(G + 1));
            double* gl = logl + G, *gof = gl + G;
            orc_population_setup_ind(nH, ll, tot, seg_n_good[s], logl, gl, gof);
            for (int g = 0; g < G; ++g) {
                out_gl[gl_off[w] + (long long)i * G + g] = gl[g];
                out_logl[gl_off[w] + (long long)i * G + g] = logl[g];
                out_gof[gl_off[w] + (long long)g * n_ind + i] = gof[g];
            }
            free(ll); free(logl);
        }
    }
    return PLAT_OK;
}

API int plat_haplotype_score_batch(plat_ctx* c, const plat_window_batch* b, int n_ind, int max_haps, const int32_t* seg_read_begin,
                                   const int32_t* seg_n_good, const double* loglik, double* out_like, int32_t* out_score, void* stream)
{
    (void)c; (void)stream; (void)max_haps;
    for (int w = 0; w < b->n_windows; ++w) {
        const int h0 = b->win_hap_begin[w], nH = b->win_hap_begin[w + 1] - h0;
        const int r0 = b->win_read_begin[w], R = b->win_read_begin[w + 1] - r0;
        int ind = -1;
        for (int i = 0; i < n_ind; ++i) if (seg_n_good[(long long)w * n_ind + i] != 0) ind = i;
        int a = 0, e = 0;
        if (ind >= 0) { a = seg_read_begin[(long long)w * n_ind + ind] - r0; e = seg_read_begin[(long long)w * n_ind + ind + 1] - r0; }
        double* like = (double*)malloc(sizeof(double) * (size_t)(nH + 1));
        double* arr = (double*)malloc(sizeof(double) * (size_t)(e - a + 1));
        for (int h = 0; h < nH; ++h) {
            memcpy(arr, loglik + b->pair_off[w] + (long long)h * R + a, sizeof(double) * (size_t)(e - a));
            arr[e - a] = 999.0;
            double h1 = 0;
            orc_genotype_loglik(arr, arr, 1, e - a, 1, NULL, &h1, NULL);
            like[h] = h1;
            if (out_like) out_like[h0 + h] = h1;
        }
        out_score[w] = nH == 0 ? 0 : orc_haplotype_score(nH, like);
        free(like); free(arr);
    }
    return PLAT_OK;
}

/* ---- EM, posteriors, genotype marginalisation -------------------------------------------------------------------------- */
API int plat_em_window_batch(plat_ctx* c, int n_windows, int n_ind, int max_haps, const int32_t* win_hap_begin, const int64_t* gl_off,
                             const int32_t* n_reads, const double* gl, int max_iters, int use_em, double* out_freq, double* out_em,
                             int32_t* out_call, int32_t* out_iters, void* stream)
{
    (void)c; (void)stream; (void)max_haps;
    for (int w = 0; w < n_windows; ++w) {
        const int nH = win_hap_begin[w + 1] - win_hap_begin[w], G = nH * (nH + 1) / 2;
        double mc = 0;
        const int it = orc_em_call(n_ind, nH, n_reads + (long long)w * n_ind, gl + gl_off[w], max_iters, use_em,
                                   out_freq + win_hap_begin[w], out_em + gl_off[w], out_call + (long long)w * n_ind, &mc);
        for (int i = 0; i < n_ind; ++i)
            if (n_reads[(long long)w * n_ind + i] == 0) memset(out_em + gl_off[w] + (long long)i * G, 0, sizeof(double) * (size_t)G);
        if (out_iters) out_iters[w] = it;
    }
    return PLAT_OK;
}

API int plat_variant_posterior_batch(plat_ctx* c, int n_vars, int n_ind, int max_haps, const int32_t* win_hap_begin,
                                     const int64_t* gl_off, const int32_t* n_reads, const double* gl, const double* freq,
                                     const int32_t* var_window, const int64_t* var_mask_off, const uint8_t* hap_has_var,
                                     const double* prior, double* out, void* stream)
{
    (void)c; (void)stream; (void)max_haps;
    for (int v = 0; v < n_vars; ++v) {
        const int w = var_window[v], nH = win_hap_begin[w + 1] - win_hap_begin[w];
        out[v] = orc_variant_posterior(n_ind, nH, n_reads + (long long)w * n_ind, gl + gl_off[w], freq + win_hap_begin[w],
                                       hap_has_var + var_mask_off[v], prior[v]);
    }
    return PLAT_OK;
}

API int plat_genotype_call_batch(plat_ctx* c, int n_sites, int n_ind, const int32_t* win_hap_begin, const int64_t* gl_off,
                                 const double* gl, const double* gof, const double* freq, const int32_t* site_window,
                                 const int32_t* site_nvar, const int64_t* site_vih_off, const int64_t* site_ref_off,
                                 const int32_t* var_in_hap, const int32_t* is_ref, const int64_t* lik_off, int32_t* out_phased,
                                 double* out_lik, double* out4, void* stream)
{
    (void)c; (void)stream;
    for (int s = 0; s < n_sites; ++s) {
        const int w = site_window[s], nH = win_hap_begin[w + 1] - win_hap_begin[w], G = nH * (nH + 1) / 2, nV = site_nvar[s];
        const int NL = (nV + 1) * (nV + 2) / 2;
        double* gofrow = (double*)malloc(sizeof(double) * (size_t)(G + 1));
        for (int i = 0; i < n_ind; ++i) {
            for (int g = 0; g < G; ++g) gofrow[g] = gof[gl_off[w] + (long long)g * n_ind + i];
            const long long t = (long long)s * n_ind + i;
            orc_genotype_call(nH, nV, n_ind, freq + win_hap_begin[w], gl + gl_off[w] + (long long)i * G, gofrow,
                              var_in_hap + site_vih_off[s], is_ref + site_ref_off[s], out_phased + 2 * t,
                              out_lik + lik_off[s] + (long long)i * NL, out4 + 4 * t);
        }
        free(gofrow);
    }
    return PLAT_OK;
}

/* ---- candidates, read slices, read statistics, assembler ---------------------------------------------------------------- */
API int plat_candidates_batch(plat_ctx* c, const plat_candidate_batch* b, int min_flank, int min_base_qual, int gen_snps, int gen_indels,
                              int max_per_read, const int32_t* read_region, int32_t* out_rec, int32_t* out_count, int32_t* out_status,
                              void* stream)
{
    (void)c; (void)stream;
    int cap = 4096;
    int* rec = (int*)malloc(sizeof(int) * 6 * (size_t)cap);
    for (int r = 0; r < b->n_reads; ++r) {
        const int g = read_region[r];
        const int refLen = (int)(b->ref_off[g + 1] - b->ref_off[g]);
        int n;
        for (;;) {
            n = orc_variant_candidates((const char*)b->ref_seq + b->ref_off[g], refLen, b->ref_seq_start[g], b->contig_len[g], 1,
                                       (const char*)b->read_seq, (const char*)b->read_qual, (const long long*)b->read_off + r,
                                       b->read_pos + r, b->read_flags + r, b->cigar, b->cig_off + r, min_flank, min_base_qual,
                                       gen_snps, gen_indels, rec, cap);
            if (n != -1) break;
            cap *= 4;
            rec = (int*)realloc(rec, sizeof(int) * 6 * (size_t)cap);
        }
        if (n < 0) { out_count[r] = 0; out_status[r] = PLAT_ERR_BAD_INPUT; continue; }
        out_count[r] = n;
        out_status[r] = n > max_per_read ? PLAT_ERR_OVERFLOW : 0;
        for (int k = 0; k < n && k < max_per_read; ++k) {
            int32_t* o = out_rec + 5 * ((long long)r * max_per_read + k);
            const int* q = rec + 6 * k;
            o[0] = q[0]; o[1] = q[1]; o[2] = q[2];
            o[3] = q[1] ? (int32_t)(b->ref_off[g] + q[3]) : -1;
            o[4] = q[2] ? q[4] : -1;
        }
    }
    free(rec);
    return PLAT_OK;
}

/* the dictionary step + support filter: a straightforward restatement (sort the scan's records by content, count runs) */
typedef struct { const int32_t* rec; int id; } fk_rec;
static __thread const plat_candidate_batch* fk_b;       /* (the caller library runs several worker threads) */
static int fk_cmp_content(const int32_t* x, const int32_t* y) {
    if (x[0] != y[0]) return x[0] < y[0] ? -1 : 1;
    if (x[1] != y[1]) return x[1] < y[1] ? -1 : 1;
    if (x[2] != y[2]) return x[2] < y[2] ? -1 : 1;
    int c = x[1] ? memcmp(fk_b->ref_seq + x[3], fk_b->ref_seq + y[3], (size_t)x[1]) : 0;
    if (c) return c;
    return x[2] ? memcmp(fk_b->read_seq + x[4], fk_b->read_seq + y[4], (size_t)x[2]) : 0;
}
static int fk_cmp(const void* a, const void* b) {
    const fk_rec* x = (const fk_rec*)a, *y = (const fk_rec*)b;
    const int c = fk_cmp_content(x->rec, y->rec);
    if (c) return c;
    return x->id < y->id ? -1 : (x->id > y->id ? 1 : 0);
}
API int plat_unpack_reads_pieces(plat_ctx* c, int n_pieces, int64_t max_piece_bytes, const plat_unpack_piece* pieces, uint8_t* out_seq, uint8_t* out_qual,
                                 int64_t total_bytes, int64_t n_exc, const int64_t* exc_index, const uint8_t* exc_base, const uint8_t* exc_qual, void* stream)
{
    (void)c; (void)stream; (void)max_piece_bytes;
    for (int k = 0; k < n_pieces; ++k)
        for (int64_t i = 0; i < pieces[k].n; ++i) {
            const unsigned b = pieces[k].src[i];
            out_seq[pieces[k].dst + i] = (uint8_t)"ACTG"[b & 3u];
            out_qual[pieces[k].dst + i] = (uint8_t)(b >> 2);
        }
    for (int64_t e = 0; e < n_exc; ++e)
        if (exc_index[e] >= 0 && exc_index[e] < total_bytes) { out_seq[exc_index[e]] = exc_base[e]; out_qual[exc_index[e]] = exc_qual[e]; }
    return PLAT_OK;
}

/* the scan on 2-bit codes: not in the fake device (the host's loop falls back to plat_unpack_reads_pieces + plat_candidates_batch) */
API int plat_unpack_reads_pieces_codes(plat_ctx* c, int n_pieces, int64_t max_piece_bytes, const plat_unpack_piece* pieces, uint8_t* out_seq, uint8_t* out_qual,
                                       uint32_t* out_codes, int64_t total_bytes, int64_t n_exc, const int64_t* exc_index, const uint8_t* exc_base,
                                       const uint8_t* exc_qual, void* stream)
{ (void)c; (void)n_pieces; (void)max_piece_bytes; (void)pieces; (void)out_seq; (void)out_qual; (void)out_codes; (void)total_bytes; (void)n_exc; (void)exc_index; (void)exc_base; (void)exc_qual; (void)stream; return PLAT_ERR_UNSUPPORTED; }
API int plat_ref_codes(plat_ctx* c, int n_regions, const uint8_t* ref_seq, const int64_t* ref_off, int64_t n_bytes, uint32_t* out_codes, int32_t* out_irregular, void* stream)
{ (void)c; (void)n_regions; (void)ref_seq; (void)ref_off; (void)n_bytes; (void)out_codes; (void)out_irregular; (void)stream; return PLAT_ERR_UNSUPPORTED; }
API int plat_candidates_batch_codes(plat_ctx* c, const plat_candidate_batch* b, const uint32_t* read_codes, const uint32_t* ref_codes, const int32_t* ref_irregular,
                                    int min_flank, int min_base_qual, int gen_snps, int gen_indels, int max_per_read, const int32_t* read_region, int32_t* out_rec,
                                    int32_t* out_count, int32_t* out_status, void* stream)
{ (void)c; (void)b; (void)read_codes; (void)ref_codes; (void)ref_irregular; (void)min_flank; (void)min_base_qual; (void)gen_snps; (void)gen_indels; (void)max_per_read; (void)read_region; (void)out_rec; (void)out_count; (void)out_status; (void)stream; return PLAT_ERR_UNSUPPORTED; }
API int plat_copy_pieces(plat_ctx* c, int n_pieces, int64_t max_piece_bytes, const plat_unpack_piece* pieces, uint8_t* dst_blob, void* stream)
{
    (void)c; (void)stream; (void)max_piece_bytes;
    for (int k = 0; k < n_pieces; ++k) memcpy(dst_blob + pieces[k].dst, pieces[k].src, (size_t)pieces[k].n);
    return PLAT_OK;
}

API int plat_concat_read_tables(plat_ctx* c, int n_tables, int max_reads_per_table, const plat_table_desc* desc, int64_t* dst_off, int32_t* dst_pos,
                                int32_t* dst_end, uint8_t* dst_mapq, int32_t* dst_flags, int32_t* dst_cig_off, int16_t* dst_cigar, int32_t* dst_region,
                                int64_t n_total, int64_t total_bytes, int64_t total_pairs, void* stream)
{
    (void)c; (void)stream; (void)max_reads_per_table;
    if (n_tables < 1 || !desc) return PLAT_ERR_INVALID;
    for (int t = 0; t < n_tables; ++t) {
        const plat_table_desc* d = &desc[t];
        for (int i = 0; i < d->n; ++i) {
            const int64_t r = d->first_read + i;
            dst_off[r] = d->first_byte + d->off[i];
            dst_pos[r] = d->pos[i]; dst_end[r] = d->end[i]; dst_mapq[r] = d->mapq[i]; dst_flags[r] = d->flags[i];
            dst_cig_off[r] = (int32_t)(d->first_pair + d->cig_off[i]);
            for (int q = d->cig_off[i]; q < d->cig_off[i + 1]; ++q) { dst_cigar[2 * (d->first_pair + q)] = d->cigar[2 * q]; dst_cigar[2 * (d->first_pair + q) + 1] = d->cigar[2 * q + 1]; }
            if (d->scan >= 0) dst_region[r] = d->scan;
        }
    }
    dst_off[n_total] = total_bytes; dst_cig_off[n_total] = (int32_t)total_pairs;
    dst_cigar[2 * total_pairs] = 0; dst_cigar[2 * total_pairs + 1] = 0;
    return PLAT_OK;
}

/* stage B on the device: this stand-in has none -- the native region loop then runs its own (host) stage B, which is what the CPU suite pins */
API int plat_stage_b_batch(plat_ctx* c, const plat_stage_b_in* b, const plat_stage_b_options* o, const plat_stage_b_out* out, void* stream)
{
    (void)c; (void)b; (void)o; (void)out; (void)stream;
    return PLAT_ERR_UNSUPPORTED;
}

API int plat_candidates_merge_batch(plat_ctx* c, const plat_candidate_batch* b, const int32_t* read_end, int n_scans,
                                    const int32_t* scan_read_begin, const int32_t* scan_longest, int max_per_read, const int32_t* rec,
                                    const int32_t* count, const int32_t* status, double min_var_freq, int cap, int32_t* out_cand,
                                    int32_t* out_n, void* stream)
{
    (void)c; (void)stream;
    fk_b = b;
    for (int g = 0; g < n_scans; ++g) {
        const int r0 = scan_read_begin[g], N = scan_read_begin[g + 1] - r0;
        int st = 0, need = 0, n = 0;
        for (int q = 0; q < N; ++q) {
            if (status[r0 + q] == PLAT_ERR_BAD_INPUT) st = PLAT_ERR_BAD_INPUT;
            if (count[r0 + q] > max_per_read) { if (count[r0 + q] > need) need = count[r0 + q]; } else n += count[r0 + q];
        }
        if (st || need) { out_n[2 * g] = 0; out_n[2 * g + 1] = st ? st : -(1 << 20) - need; continue; }
        fk_rec* v = (fk_rec*)malloc(sizeof(fk_rec) * (size_t)(n + 1));
        n = 0;
        for (int q = 0; q < N; ++q)
            for (int k = 0; k < count[r0 + q]; ++k) { const int id = (r0 + q) * max_per_read + k; v[n].rec = rec + 5ll * id; v[n].id = id; ++n; }
        qsort(v, (size_t)n, sizeof(fk_rec), fk_cmp);
        int nout = 0;
        const int32_t* pos = b->read_pos + r0; const int32_t* endp = read_end + r0;
        for (int i = 0; i < n && st == 0;) {
            int j = i + 1;
            while (j < n && fk_cmp_content(v[i].rec, v[j].rec) == 0) ++j;
            const int32_t* me = v[i].rec;
            const int start = me[0], cnt = j - i;
            int total = 0;
            if (N > 0) {
                long long key = (long long)start - scan_longest[g]; if (key < 1) key = 1;
                int s = 0; while (s < N && (long long)pos[s] < key) ++s;
                int e = 0; while (e < N && pos[e] < start + 1) ++e;
                while (s < N && endp[s] <= start) ++s;
                if (s > e) { st = PLAT_ERR_BAD_INPUT; break; }
                total = e - s;
            }
            const double frac = total == 0 ? 0.0 : (double)cnt / (double)total;
            if (frac >= min_var_freq || me[1] != me[2]) {
                if (nout < cap) {
                    int32_t* o = out_cand + 8ll * ((long long)g * cap + nout);
                    o[0] = v[i].id; o[1] = cnt; o[2] = total; memcpy(o + 3, me, 5 * sizeof(int32_t));
                }
                ++nout;
            }
            i = j;
        }
        free(v);
        out_n[2 * g] = (st || nout > cap) ? 0 : nout;
        out_n[2 * g + 1] = st ? st : (nout > cap ? PLAT_ERR_OVERFLOW : 0);
    }
    return PLAT_OK;
}

API int plat_gather_reads(plat_ctx* c, int64_t n_dst, const int32_t* src_index, const int64_t* dst_off, const uint8_t* src_seq,
                          const uint8_t* src_qual, const int64_t* src_off, const int32_t* src_pos, const int32_t* src_end,
                          const uint8_t* src_mapq, const int32_t* src_flags, uint8_t* dst_seq, uint8_t* dst_qual, int32_t* dst_pos,
                          int32_t* dst_end, uint8_t* dst_mapq, int32_t* dst_flags, void* stream)
{
    (void)c; (void)stream;
    for (int64_t d = 0; d < n_dst; ++d) {
        const int s = src_index[d];
        const int64_t n = src_off[s + 1] - src_off[s];
        memcpy(dst_seq + dst_off[d], src_seq + src_off[s], (size_t)n);
        memcpy(dst_qual + dst_off[d], src_qual + src_off[s], (size_t)n);
        dst_pos[d] = src_pos[s]; dst_end[d] = src_end[s]; dst_mapq[d] = src_mapq[s]; dst_flags[d] = src_flags[s];
    }
    return PLAT_OK;
}

API int plat_unpack_reads(plat_ctx* c, int64_t n_bytes, const uint8_t* packed, uint8_t* out_seq, uint8_t* out_qual, int64_t n_exc,
                          const int64_t* exc_index, const uint8_t* exc_base, const uint8_t* exc_qual, void* stream)
{
    (void)c; (void)stream;
    for (int64_t i = 0; i < n_bytes; ++i) { out_seq[i] = (uint8_t)"ACTG"[packed[i] & 3]; out_qual[i] = packed[i] >> 2; }
    for (int64_t k = 0; k < n_exc; ++k)
        if (exc_index[k] >= 0 && exc_index[k] < n_bytes) { out_seq[exc_index[k]] = exc_base[k]; out_qual[exc_index[k]] = exc_qual[k]; }
    return PLAT_OK;
}

API int plat_variant_read_stats_batch(plat_ctx* c, const plat_infostats_batch* b, int bad_reads_window, int exact, int64_t* out_counts,
                                      int32_t* out_per_sample, int32_t* out_minq, int32_t* out_nminq, void* stream)
{
    (void)c; (void)stream;
    const int nI = b->n_ind;
    for (int v = 0; v < b->n_vars; ++v) {
        const int w = b->var_window[v];
        int maxq = 0;
        for (int i = 0; i < nI; ++i) maxq += b->good_end[(long long)w * nI + i] - b->good_begin[(long long)w * nI + i];
        if (maxq < 1) maxq = 1;
        const int zero = 0;
        int* tmp = (int*)calloc((size_t)maxq + 1, sizeof(int));
        orc_variant_read_stats(1, b->var_pos + v, b->var_bam_min + v, b->var_bam_max + v, b->var_n_added + v, b->var_n_removed + v,
                               (const char*)b->var_added + b->var_added_off[v], &zero, nI, b->good_begin + (long long)w * nI,
                               b->good_end + (long long)w * nI, b->bad_begin + (long long)w * nI, b->bad_end + (long long)w * nI,
                               b->var_in_genotype + (long long)v * nI, (const char*)b->read_seq, (const char*)b->read_qual,
                               (const long long*)b->read_off, b->read_pos, b->read_end, b->read_mapq, b->read_flags, b->cigar,
                               b->cig_off, 20, bad_reads_window, exact, (long long*)out_counts + 16 * (long long)v,
                               out_per_sample + 2 * (long long)v * nI, tmp, maxq, out_nminq + v);
        memcpy(out_minq + b->minq_off[v], tmp, sizeof(int) * (size_t)out_nminq[v]);
        free(tmp);
    }
    return PLAT_OK;
}

API int plat_variant_info_batch(plat_ctx* c, int n, const int64_t* counts, const int64_t* mo, const int32_t* mq, const int32_t* nm, double* t, int32_t* m, void* st)
{ (void)c; (void)n; (void)counts; (void)mo; (void)mq; (void)nm; (void)t; (void)m; (void)st; return PLAT_ERR_UNSUPPORTED; }   /* the host's own loops then */

API int plat_assemble_batch(plat_ctx* c, const plat_assembly_batch* b, int kmer_size, int min_qual, int min_weight, int no_cycles,
                            int max_vars, int blob_per_region, int32_t* var_count, int32_t* var_pos, int32_t* var_nrem, int32_t* var_nadd,
                            int32_t* var_off, uint8_t* var_blob, int32_t* status, void* stream)
{
    (void)c; (void)stream;
    for (int g = 0; g < b->n_regions; ++g) {
        const int r0 = b->reg_read_begin[g], nR = b->reg_read_begin[g + 1] - r0;
        int* off = (int*)malloc(sizeof(int) * (size_t)(nR + 1)), *len = (int*)malloc(sizeof(int) * (size_t)(nR + 1));
        const long long base = nR ? b->read_off[r0] : 0;
        for (int r = 0; r < nR; ++r) { off[r] = (int)(b->read_off[r0 + r] - base); len[r] = (int)(b->read_off[r0 + r + 1] - b->read_off[r0 + r]); }
        int nn = 0;
        const int n = orc_assemble((const char*)b->ref_seq + b->ref_off[g], (int)(b->ref_off[g + 1] - b->ref_off[g]), b->ref_start[g],
                                   b->assem_start[g], b->assem_end[g], nR, (const char*)b->read_seq + base, (const char*)b->read_qual + base,
                                   off, len, kmer_size, min_qual, min_weight, no_cycles, max_vars, var_pos + (long long)g * max_vars,
                                   var_nrem + (long long)g * max_vars, var_nadd + (long long)g * max_vars, var_off + (long long)g * max_vars,
                                   (char*)var_blob + (long long)g * blob_per_region, blob_per_region, &nn);
        var_count[g] = n < 0 ? 0 : n;
        status[g] = n < 0 ? PLAT_ERR_OVERFLOW : 0;
        free(off); free(len);
    }
    return PLAT_OK;
}

API int plat_assemble_batch_async(plat_ctx* c, const plat_assembly_batch* b, const plat_assembly_hints* hints, int kmer_size, int min_qual, int min_weight,
                                  int no_cycles, int max_vars, int blob_per_region, int32_t* var_count, int32_t* var_pos, int32_t* var_nrem,
                                  int32_t* var_nadd, int32_t* var_off, uint8_t* var_blob, int32_t* status, void* stream)
{
    if (!hints) return PLAT_ERR_INVALID;
    for (int g = 0; g < b->n_regions; ++g) {                            /* the device's check of the caller's sizes */
        const long long rl = b->ref_off[g + 1] - b->ref_off[g];
        const int r0 = b->reg_read_begin[g], r1 = b->reg_read_begin[g + 1];
        const long long bytes = r1 > r0 ? b->read_off[r1] - b->read_off[r0] : 0;
        if (rl > hints->max_ref_len || r1 - r0 > hints->max_reads_per_region || rl + 2 + bytes + 2ll * (r1 - r0) > hints->max_positions) {
            for (int k = 0; k < b->n_regions; ++k) { status[k] = PLAT_ERR_BAD_HINTS; var_count[k] = 0; }
            return PLAT_OK;
        }
    }
    return plat_assemble_batch(c, b, kmer_size, min_qual, min_weight, no_cycles, max_vars, blob_per_region, var_count, var_pos, var_nrem, var_nadd, var_off,
                               var_blob, status, stream);
}
