# cplat.pxd -- Cython declarations of libplat_mi355x.so (include/platypus_mi355x.h), the file a Platypus maintainer adds next to
# src/cython/chaplotype.pxd to call the MI355X library from the reference's own Cython modules:
#
#     cimport cplat                         # in chaplotype.pyx / cgenotype.pyx / cpopulation.pyx / assembler.pyx / variantcaller.pyx
#
# It replaces, at the call sites listed in INTEGRATION.md section 2, the per-read loops behind
#     cdef double* Haplotype.alignReads(...)                         src/cython/chaplotype.pxd:44
#     cdef double  Haplotype.alignSingleRead(...)                    src/cython/chaplotype.pxd:45
#     cdef double  DiploidGenotype.calculateDataLikelihood(...)      src/cython/cgenotype.pxd:14
#     cdef void    Population.setup(...)                             src/cython/cpopulation.pxd:55
#     cdef list    assembleReadsAndDetectVariants(...)               src/cython/assembler.pxd:3
#     int fastAlignmentRoutine(...)                                  src/c/align.h:8-10
# Every declaration below is checked by the compiler against the header: bindings/plat_binding_check.pyx cimports this
# file and tests/test_binding_cpu.py builds it with Cython 3 and links it to the library.
from libc.stdint cimport int16_t, int32_t, int64_t, uint8_t, uint32_t

cdef extern from "platypus_mi355x.h":
    int PLAT_ABI_VERSION
    int PLAT_BLOB_PAD
    int PLAT_OK
    int PLAT_ERR_INVALID
    int PLAT_ERR_HIP
    int PLAT_ERR_NOMEM
    int PLAT_ERR_HAP_TOO_LONG
    int PLAT_ERR_HAP_TOO_SHORT
    int PLAT_ERR_UNSUPPORTED
    int PLAT_ERR_NO_DEVICE
    int PLAT_ERR_OVERFLOW
    int PLAT_ERR_BAD_INPUT
    int PLAT_ERR_BAD_HINTS

    ctypedef struct plat_ctx:
        pass

    # ---- context & memory
    int plat_abi_version() nogil
    const char* plat_strerror(int code) nogil
    int plat_device_count(int* out_count) nogil
    int plat_ctx_create(int device, plat_ctx** out_ctx) nogil
    int plat_ctx_destroy(plat_ctx* ctx) nogil
    int plat_last_hip_error(const plat_ctx* ctx) nogil
    int plat_malloc(plat_ctx* ctx, size_t nbytes, void** out_dev_ptr) nogil
    int plat_free(plat_ctx* ctx, void* dev_ptr) nogil
    int plat_memcpy_h2d(plat_ctx* ctx, void* dst_dev, const void* src_host, size_t nbytes, void* stream) nogil
    int plat_memcpy_d2d(plat_ctx* ctx, void* dst_dev, const void* src_dev, size_t nbytes, void* stream) nogil
    int plat_memcpy_d2h(plat_ctx* ctx, void* dst_host, const void* src_dev, size_t nbytes, void* stream) nogil
    int plat_memset(plat_ctx* ctx, void* dst_dev, int value, size_t nbytes, void* stream) nogil
    int plat_stream_sync(plat_ctx* ctx, void* stream) nogil
    int plat_stream_create(plat_ctx* ctx, void** out_stream) nogil
    int plat_stream_destroy(plat_ctx* ctx, void* stream) nogil
    int plat_host_alloc(plat_ctx* ctx, size_t nbytes, void** out_host_ptr) nogil
    int plat_host_free(plat_ctx* ctx, void* host_ptr) nogil

    # ---- live kernel timing
    ctypedef struct plat_profile:
        float ms_prepare
        float ms_seed
        float ms_dp
        float ms_finalize
        float ms_genotype
        float ms_seed_kernel
        int64_t dp_jobs
        int64_t dp_alg_bytes
        float ms_sweep
        float ms_pairs
        float ms_unpack
        float ms_candidates
    int plat_profile_enable(plat_ctx* ctx, int on) nogil
    int plat_profile_last(plat_ctx* ctx, plat_profile* out) nogil
    int plat_sync_poll_us(plat_ctx* ctx, int microseconds) nogil
    const char* plat_kernel_timer_name(int id) nogil
    int plat_kernel_times(plat_ctx* ctx, double* out_ms, int64_t* out_launches) nogil
    int plat_kernel_timer_only(plat_ctx* ctx, int id) nogil

    # ---- fastAlignmentRoutine, score only (src/c/align.h:8-10)
    int plat_dp_batch(plat_ctx* ctx, int n, int lmax, const uint8_t* hap_slices, const uint8_t* reads, const uint8_t* quals,
                      const uint8_t* gapopen, const int32_t* len2, int gapextend, int nucprior, int32_t* out_score, void* stream) nogil

    # ---- Haplotype.alignReads / alignSingleRead for whole windows (chaplotype.pxd:44-45)
    ctypedef struct plat_window_batch:
        int32_t n_windows
        int32_t n_haps
        int32_t n_reads
        int32_t _pad
        const int32_t* win_hap_begin
        const int32_t* win_read_begin
        const int32_t* win_start
        const int32_t* win_end
        const int32_t* win_flank
        const int64_t* pair_off
        const uint8_t* hap_seq
        const int64_t* hap_off
        const uint8_t* read_seq
        const uint8_t* read_qual
        const int64_t* read_off
        const int32_t* read_pos
        const int32_t* read_end
        const uint8_t* read_mapq
        const int32_t* read_flags
        const uint8_t* read_kind
    ctypedef struct plat_align_stats:
        int64_t n_pairs
        int64_t n_pairs_aligned
        int64_t n_dp_launched
        int64_t n_dp_reference
        int64_t cells_reference
        int64_t cells_launched
        int64_t n_seed_fallback
        int64_t _reserved
    ctypedef struct plat_batch_hints:
        int32_t max_hap_len
        int32_t max_read_len
        int32_t max_reads_per_window
        int32_t _pad
        int64_t n_pairs
        int64_t hap_blob_len
        int64_t read_blob_len
        int64_t extra_jobs_cap
    int plat_align_window_batch(plat_ctx* ctx, const plat_window_batch* batch, int calc_flank_score, int use_mapq_cap,
                                double* out_loglik, int32_t* out_score, plat_align_stats* out_stats, void* stream) nogil
    int plat_align_window_batch_async(plat_ctx* ctx, const plat_window_batch* batch, const plat_batch_hints* hints, int calc_flank_score,
                                      int use_mapq_cap, double* out_loglik, int32_t* out_score, void* stream) nogil

    # ---- DiploidGenotype.calculateDataLikelihood + Population.setup (cgenotype.pxd:14, cpopulation.pxd:55)
    int plat_genotype_window_batch(plat_ctx* ctx, const plat_window_batch* batch, int n_ind, const int32_t* seg_read_begin,
                                   const int32_t* seg_n_good, const double* loglik, const int64_t* gl_off, double* out_gl,
                                   double* out_logl, double* out_gof, void* stream) nogil

    # ---- Population.call, calculatePosterior, computeGenotypeCallAndLikelihoods (cpopulation.pyx:384-703, vcfutils.pyx:163-334)
    int plat_em_window_batch(plat_ctx* ctx, int n_windows, int n_ind, int max_haps_per_window, const int32_t* win_hap_begin,
                             const int64_t* gl_off, const int32_t* n_reads, const double* gl, int max_iters, int use_em_likelihoods,
                             double* out_freq, double* out_em, int32_t* out_call, int32_t* out_iters, void* stream) nogil
    int plat_variant_posterior_batch(plat_ctx* ctx, int n_vars, int n_ind, int max_haps_per_window, const int32_t* win_hap_begin,
                                     const int64_t* gl_off, const int32_t* n_reads, const double* gl, const double* freq,
                                     const int32_t* var_window, const int64_t* var_mask_off, const uint8_t* hap_has_var,
                                     const double* prior, double* out_posterior, void* stream) nogil
    int plat_genotype_call_batch(plat_ctx* ctx, int n_sites, int n_ind, const int32_t* win_hap_begin, const int64_t* gl_off,
                                 const double* gl, const double* gof, const double* freq, const int32_t* site_window,
                                 const int32_t* site_nvar, const int64_t* site_vih_off, const int64_t* site_ref_off,
                                 const int32_t* var_in_hap, const int32_t* is_ref, const int64_t* lik_off, int32_t* out_phased,
                                 double* out_lik, double* out4, void* stream) nogil

    # ---- computeHaplotypeScore (vcfutils.pyx:1076-1114)
    int plat_haplotype_score_batch(plat_ctx* ctx, const plat_window_batch* batch, int n_ind, int max_haps_per_window,
                                   const int32_t* seg_read_begin, const int32_t* seg_n_good, const double* loglik,
                                   double* out_hap_like, int32_t* out_hap_score, void* stream) nogil

    # ---- VariantCandidateGenerator.addCandidatesFromReads (variant.pyx:722-743)
    ctypedef struct plat_candidate_batch:
        int32_t n_regions
        int32_t n_reads
        const uint8_t* ref_seq
        const int64_t* ref_off
        const int32_t* ref_seq_start
        const int32_t* contig_len
        const uint8_t* read_seq
        const uint8_t* read_qual
        const int64_t* read_off
        const int32_t* read_pos
        const int32_t* read_flags
        const int16_t* cigar
        const int32_t* cig_off
    int plat_candidates_batch(plat_ctx* ctx, const plat_candidate_batch* batch, int min_flank, int min_base_qual, int gen_snps,
                              int gen_indels, int max_per_read, const int32_t* read_region, int32_t* out_rec, int32_t* out_count,
                              int32_t* out_status, void* stream) nogil

    # ---- addVariantToList + the per-sample support filter (variant.pyx:499-527, variantcaller.pyx:456-467)
    int plat_candidates_merge_batch(plat_ctx* ctx, const plat_candidate_batch* batch, const int32_t* read_end, int n_scans,
                                    const int32_t* scan_read_begin, const int32_t* scan_longest, int max_per_read, const int32_t* rec,
                                    const int32_t* count, const int32_t* status, double min_var_freq, int cap_per_scan,
                                    int32_t* out_cand, int32_t* out_n, void* stream) nogil

    # ---- candidates -> variants -> windows -> haplotypes -> the window batch, one sample per region (variantcaller.pyx:456-531,
    #      platypusutils.pyx:806-931, variantFilter.pyx:98-171,377-441, window.py:49-238, chaplotype.pyx:127-191,397-449)
    ctypedef struct plat_stage_b_options:
        int32_t minReads
        int32_t maxSize
        int32_t mergeClusteredVariants
        int32_t maxVarDist
        int32_t minVarDist
        int32_t largeWindows
        int32_t maxVariants
        int32_t maxHaplotypes
        int32_t filterVarsByCoverage
        int32_t skipDifficultWindows
        double maxReads
    ctypedef struct plat_stage_b_in:
        int32_t n_regions
        int32_t cap_per_scan
        const int32_t* cand
        const int32_t* cand_n
        const int32_t* cand_rec
        const int64_t* region_name_hash
        const uint8_t* ref_seq
        const int64_t* ref_off
        const int32_t* ref_seq_start
        const int32_t* contig_len
        const int32_t* region_start
        const int32_t* region_end
        const int32_t* region_rlen
        const uint8_t* read_seq
        const int64_t* read_off
        const int32_t* read_pos
        const int32_t* read_end
        const int32_t* tab_begin
        const int32_t* tab_n
        const int32_t* tab_longest
        const int32_t* broken_mate_pos
        int32_t broken_base
        int32_t cap_vars
        int32_t cap_windows
        int32_t cap_added
        int32_t cap_batch_windows
        int32_t cap_batch_haps
        int32_t cap_batch_reads
        int64_t cap_hap_bytes
    ctypedef struct plat_stage_b_out:
        int32_t* hdr
        int32_t* var_pos
        int32_t* var_nrem
        int32_t* var_nadd
        int32_t* var_support
        int32_t* var_bam_min
        int32_t* var_bam_max
        int32_t* var_rem_pos
        int32_t* var_add_off
        uint8_t* added
        int32_t* win_start
        int32_t* win_end
        int32_t* win_var_first
        int32_t* win_var_n
        int32_t* win_flags
        int32_t* win_ptrs
        int32_t* win_n_haps
        int32_t* win_batch
        int32_t* b_hap_begin
        int32_t* b_read_begin
        int32_t* b_start
        int32_t* b_end
        int32_t* b_flank
        int64_t* b_pair_off
        int64_t* b_gl_off
        int32_t* b_seg_begin
        int32_t* b_n_good
        int64_t* b_hap_off
        uint32_t* b_hap_mask
        uint8_t* b_hap_seq
        int64_t* b_read_off
        int32_t* b_read_src
        uint8_t* b_read_kind
        int64_t* totals
        int32_t* scratch
    int plat_stage_b_batch(plat_ctx* ctx, const plat_stage_b_in* batch, const plat_stage_b_options* options, const plat_stage_b_out* out,
                           void* stream) nogil

    ctypedef struct plat_unpack_piece:
        const uint8_t* src
        int64_t dst
        int64_t n
    int plat_unpack_reads_pieces(plat_ctx* ctx, int n_pieces, int64_t max_piece_bytes, const plat_unpack_piece* pieces, uint8_t* out_seq, uint8_t* out_qual,
                                 int64_t total_bytes, int64_t n_exc, const int64_t* exc_index, const uint8_t* exc_base, const uint8_t* exc_qual, void* stream) nogil

    int plat_unpack_reads_pieces_codes(plat_ctx* ctx, int n_pieces, int64_t max_piece_bytes, const plat_unpack_piece* pieces, uint8_t* out_seq, uint8_t* out_qual,
                                       uint32_t* out_codes, int64_t total_bytes, int64_t n_exc, const int64_t* exc_index, const uint8_t* exc_base,
                                       const uint8_t* exc_qual, void* stream) nogil
    int plat_ref_codes(plat_ctx* ctx, int n_regions, const uint8_t* ref_seq, const int64_t* ref_off, int64_t n_bytes, uint32_t* out_codes,
                       int32_t* out_irregular, void* stream) nogil
    int plat_candidates_batch_codes(plat_ctx* ctx, const plat_candidate_batch* batch, const uint32_t* read_codes, const uint32_t* ref_codes,
                                    const int32_t* ref_irregular, int min_flank, int min_base_qual, int gen_snps, int gen_indels, int max_per_read,
                                    const int32_t* read_region, int32_t* out_rec, int32_t* out_count, int32_t* out_status, void* stream) nogil
    int plat_copy_pieces(plat_ctx* ctx, int n_pieces, int64_t max_piece_bytes, const plat_unpack_piece* pieces, uint8_t* dst_blob, void* stream) nogil

    # ---- a chunk's read table from tables resident on the device
    ctypedef struct plat_table_desc:
        const int64_t* off
        const int32_t* pos
        const int32_t* end
        const uint8_t* mapq
        const int32_t* flags
        const int16_t* cigar
        const int32_t* cig_off
        int32_t n
        int32_t scan
        int64_t first_read
        int64_t first_byte
        int64_t first_pair
    int plat_concat_read_tables(plat_ctx* ctx, int n_tables, int max_reads_per_table, const plat_table_desc* desc, int64_t* dst_off, int32_t* dst_pos,
                                int32_t* dst_end, uint8_t* dst_mapq, int32_t* dst_flags, int32_t* dst_cig_off, int16_t* dst_cigar, int32_t* dst_region,
                                int64_t n_total_reads, int64_t total_bytes, int64_t total_pairs, void* stream) nogil

    # ---- checkAndTrimRead (cwindow.pyx:332-481)
    ctypedef struct plat_readqc_batch:
        int32_t n_reads
        int32_t _pad
        uint8_t* read_qual
        const int64_t* read_off
        const int32_t* read_pos
        const uint8_t* read_mapq
        int32_t* read_flags
        const int16_t* chrom_id
        const int16_t* mate_chrom_id
        const int32_t* insert_size
        const int32_t* mate_pos
        const int16_t* cigar
        const int32_t* cig_off
        const int32_t* stream_of
    ctypedef struct plat_readqc_options:
        int32_t min_good_qual_bases
        int32_t min_map_qual
        int32_t min_base_qual
        int32_t trim_overlapping
        int32_t trim_adapter
        int32_t trim_read_flank
        int32_t trim_soft_clipped
        int32_t filter_mate_unmapped
        int32_t filter_mate_distant
        int32_t filter_small_insert
        int32_t filter_duplicates
    int plat_read_qc_batch(plat_ctx* ctx, const plat_readqc_batch* batch, const plat_readqc_options* options, int32_t* out_ok,
                           int32_t* out_reason, void* stream) nogil

    # ---- window read slices out of a resident read table (cwindow.pyx:208-264,655-689)
    # the read table of a loader that wrote one byte per base (2-bit base | quality << 2) expanded to ASCII on the device
    int plat_unpack_reads(plat_ctx* ctx, int64_t n_bytes, const uint8_t* packed, uint8_t* out_seq, uint8_t* out_qual, int64_t n_exc,
                          const int64_t* exc_index, const uint8_t* exc_base, const uint8_t* exc_qual, void* stream) nogil
    int plat_gather_reads(plat_ctx* ctx, int64_t n_dst, const int32_t* src_index, const int64_t* dst_off, const uint8_t* src_seq,
                          const uint8_t* src_qual, const int64_t* src_off, const int32_t* src_pos, const int32_t* src_end,
                          const uint8_t* src_mapq, const int32_t* src_flags, uint8_t* dst_seq, uint8_t* dst_qual, int32_t* dst_pos,
                          int32_t* dst_end, uint8_t* dst_mapq, int32_t* dst_flags, void* stream) nogil

    # ---- the per-read loop of vcfINFO (vcfutils.pyx:1300-1390)
    ctypedef struct plat_infostats_batch:
        int32_t n_vars
        int32_t n_ind
        const int32_t* var_window
        const int32_t* var_pos
        const int32_t* var_bam_min
        const int32_t* var_bam_max
        const int32_t* var_n_added
        const int32_t* var_n_removed
        const uint8_t* var_added
        const int64_t* var_added_off
        const uint8_t* var_in_genotype
        const int64_t* minq_off
        const int32_t* good_begin
        const int32_t* good_end
        const int32_t* bad_begin
        const int32_t* bad_end
        const uint8_t* read_seq
        const uint8_t* read_qual
        const int64_t* read_off
        const int32_t* read_pos
        const int32_t* read_end
        const uint8_t* read_mapq
        const int32_t* read_flags
        const int16_t* cigar
        const int32_t* cig_off
    int plat_variant_read_stats_batch(plat_ctx* ctx, const plat_infostats_batch* batch, int bad_reads_window,
                                      int count_only_exact_indel_matches, int64_t* out_counts, int32_t* out_per_sample,
                                      int32_t* out_minq, int32_t* out_nminq, void* stream) nogil
    int plat_variant_info_batch(plat_ctx* ctx, int n_vars, const int64_t* counts, const int64_t* minq_off, const int32_t* minq,
                                const int32_t* n_minq, double* out_terms, int32_t* out_mmlq, void* stream) nogil

    # ---- assembleReadsAndDetectVariants (assembler.pxd:3)
    ctypedef struct plat_assembly_batch:
        int32_t n_regions
        int32_t n_reads
        const uint8_t* ref_seq
        const int64_t* ref_off
        const int32_t* ref_start
        const int32_t* assem_start
        const int32_t* assem_end
        const int32_t* reg_read_begin
        const uint8_t* read_seq
        const uint8_t* read_qual
        const int64_t* read_off
    int plat_assemble_batch(plat_ctx* ctx, const plat_assembly_batch* batch, int kmer_size, int min_qual, int min_weight, int no_cycles,
                            int max_vars_per_region, int blob_per_region, int32_t* var_count, int32_t* var_pos, int32_t* var_nrem,
                            int32_t* var_nadd, int32_t* var_off, uint8_t* var_blob, int32_t* status, void* stream) nogil
    ctypedef struct plat_assembly_hints:
        int32_t max_ref_len
        int32_t max_reads_per_region
        int64_t max_positions
    int plat_assemble_batch_async(plat_ctx* ctx, const plat_assembly_batch* batch, const plat_assembly_hints* hints, int kmer_size, int min_qual,
                                  int min_weight, int no_cycles, int max_vars_per_region, int blob_per_region, int32_t* var_count, int32_t* var_pos,
                                  int32_t* var_nrem, int32_t* var_nadd, int32_t* var_off, uint8_t* var_blob, int32_t* status, void* stream) nogil
