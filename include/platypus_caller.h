/*
 * platypus_caller.h -- C ABI of libplat_caller.so: the region loop around the device hot path, native.
 *
 * Replaces, for a list of regions whose reads are already in host memory (structure-of-arrays, as a BAM
 * loader would leave them), the reference's per-region driver
 *
 *     PlatypusSingleProcess.run -> callVariantsInRegion        src/cython/variantcaller.pyx:535-615, 935-1012
 *         generateVariantsInRegion                             src/cython/variantcaller.pyx:412-531
 *         WindowGenerator.WindowsAndVariants                   src/python/window.py:140-238
 *         callVariantsInWindow                                 src/cython/variantcaller.pyx:74-141
 *         outputCallToVCF / VCF.write_data                     src/cython/vcfutils.pyx:338-599, src/python/vcf.py:710-739
 *
 * and writes the same VCF record lines.  It is host code (C++, threads) on top of libplat_mi355x.so
 * (include/platypus_mi355x.h): every O(reads) stage -- candidate scan, window read slices, likelihoods,
 * genotype likelihoods, HapScore, EM, posteriors, read statistics, per-site genotype calls -- is one device
 * call per chunk of regions; the host keeps what the reference keeps in Python (merging candidates, indel
 * left-normalisation, windows, haplotype enumeration, INFO / FILTER arithmetic, text).  Regions are
 * processed in chunks by worker threads, each with its own plat_ctx and HIP stream, so uploads, kernels and
 * host work of different chunks overlap.  The Python layer platypus_amd/caller.py does the same job one
 * window or one batch at a time and is what the parity tests compare this library with.
 *
 * Plain C: pointers and sizes, int status (0 = ok, negative = error of platypus_mi355x.h), no exceptions.
 * All pointers here are HOST pointers.  Not built: source VCFs (PLAT_ERR_UNSUPPORTED).
 */
#ifndef PLATYPUS_CALLER_H
#define PLATYPUS_CALLER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* How the bases and qualities of a plat_read_table are stored.
 *   PLAT_READS_ASCII   seq = 7-bit ASCII bases, qual = raw phred bytes: what cAlignedRead holds after the loader's decode loop
 *                      (htslibWrapper.pyx:330-370: one pass over every base, 4-bit BAM code -> letter, quality byte copied): 2 bytes
 *                      per base cross the host-to-device link.
 *   PLAT_READS_PACKED  seq = ONE byte per base: bits 0..1 the base ((letter >> 1) & 3: A 0, C 1, T 2, G 3), bits 2..7 the quality
 *                      (0..63); `qual` is not read.  Bases other than A/C/G/T and qualities above 63 are listed as exceptions
 *                      (exc_index = byte index into seq, ascending; exc_base / exc_qual = the real letter and quality; the packed
 *                      byte under an exception is ignored).  Same cost for the loader -- its decode loop writes this byte instead of
 *                      two -- and half the bytes on the link; the device expands the table to ASCII once (plat_unpack_reads), so
 *                      every kernel downstream sees what it sees with PLAT_READS_ASCII and the records are the same. */
enum { PLAT_READS_ASCII = 0, PLAT_READS_PACKED = 1 };

/* One ReadArray (cwindow.pyx:92-236) as arrays: cAlignedRead fields (htslibWrapper.pxd:187-201) of n_reads reads.
 * seq / qual: byte blobs (see `encoding`), read r at [off[r], off[r+1]); both followed by >= 32
 * readable bytes.  cigar: (op, length) int16 pairs, read r owns pairs [cig_off[r], cig_off[r+1]). */
typedef struct plat_read_table {
    int32_t n_reads, encoding;  /* PLAT_READS_* */
    const uint8_t* seq;
    const uint8_t* qual;
    const int64_t* off;        /* [n_reads+1] */
    const int32_t* pos;        /* cAlignedRead.pos */
    const int32_t* end;        /* cAlignedRead.end */
    const uint8_t* mapq;
    const int32_t* flags;      /* bitFlag */
    const int32_t* mate_pos;   /* matePos (orders brokenMates) */
    const int16_t* cigar;
    const int32_t* cig_off;    /* [n_reads+1], in pairs */
    int64_t n_exceptions;      /* PLAT_READS_PACKED only */
    const int64_t* exc_index;
    const uint8_t* exc_base;
    const uint8_t* exc_qual;
    /* Optional: DEVICE copies of `seq` (and, PLAT_READS_ASCII, `qual`) for a table whose bytes are already resident in HBM -- a loader
     * that keeps the reads on the device, or a job that uploaded them earlier.  When set, the chunk table is built with device-to-device
     * copies and nothing of the blobs crosses the link; `seq` must still be valid host memory (the host reads the inserted bases of the
     * few candidates it keeps).  NULL (the default): the blobs are copied from `seq` / `qual`. */
    const uint8_t* dev_seq;
    const uint8_t* dev_qual;
    /* Optional, same idea for the per-read arrays: DEVICE copies of off / pos / end / mapq / flags / cigar / cig_off (all seven or none).
     * When every table of a chunk has them (and dev_seq), the chunk's read table is put together on the device (plat_concat_read_tables)
     * and no per-read array crosses the link; the host arrays must still be valid (window pointers of the host's own stages). */
    const int64_t* dev_off;
    const int32_t* dev_pos;
    const int32_t* dev_end;
    const uint8_t* dev_mapq;
    const int32_t* dev_flags;
    const int16_t* dev_cigar;
    const int32_t* dev_cig_off;
    /* Optional: what a loader knows of its table anyway -- ReadArray keeps the length of its longest read as reads are appended
     * (cwindow.pyx:124,173-174,272; variantcaller.pyx:476-488 sets options.rlen from it).  longest_read = max(end[r] - pos[r]),
     * most_bases = max(off[r+1] - off[r]) over the table's reads: EXACT values, or 0 = not known (the library then walks pos / end / off of every
     * read of the region once, 16 bytes per read of cold host memory: a tenth of the region loop's CPU time on the 30x WGS job).
     * PLAT_CALLER_CHECK_HINTS=1: every hint is checked against the arrays and a wrong one refused (PLAT_ERR_BAD_INPUT). */
    int32_t longest_read;
    int32_t most_bases;
} plat_read_table;

/* One bamReadBuffer (cwindow.pyx:485-513): reads and badReads sorted by pos, brokenMates sorted by mate_pos
 * (bamReadBuffer.sortReads / sortBrokenMates, cwindow.pyx:748-766). */
typedef struct plat_sample_reads {
    plat_read_table reads, bad_reads, broken_mates;
} plat_sample_reads;

/* One region of the region list (runner.py:454-474): chrom:start-end, the contig's sequence (upper case; what
 * FastaFile.getSequence reads from, fastafile.pyx:173-207) and one plat_sample_reads per sample. */
typedef struct plat_region {
    const char* chrom;
    int32_t start, end;
    const uint8_t* contig_seq;
    int64_t contig_len;
    const plat_sample_reads* samples;   /* [n_samples] */
    /* Optional: a DEVICE copy of contig_seq (a reference that is resident in HBM).  When every region of a chunk has one, the chunk's
     * reference windows are put together on the device (plat_copy_pieces) and no reference byte crosses the link; contig_seq must still
     * be valid host memory. */
    const uint8_t* dev_contig_seq;
} plat_region;

/* The callVariants options the region loop reads (names and defaults of runner.py:519-597). */
typedef struct plat_caller_options {
    int32_t rlen;                        /* maxReadLength 150; in/out: follows the longest read (variantcaller.pyx:476-488) */
    int32_t minReads;                    /* 2 */
    double maxReads;                     /* 5000000 */
    int32_t maxSize, largeWindows;       /* 1500, 0 */
    int32_t maxVariants, coverageSamplingLevel, maxHaplotypes, originalMaxHaplotypes, skipDifficultWindows;   /* 8, 30, 50, 50, 0 */
    int32_t getVariantsFromBAMs, genSNPs, genIndels, mergeClusteredVariants, minFlank;                        /* 1, 1, 1, 1, 10 */
    int32_t filterVarsByCoverage;        /* 1 */
    double filteredReadsFrac;            /* 0.7 */
    int32_t maxVarDist, minVarDist;      /* 15, 9 */
    int32_t useEMLikelihoods, countOnlyExactIndelMatches, calculateFlankScore;                                /* 0, 0, 0 */
    int32_t assemble, outputRefCalls;    /* 0, 0 */
    int32_t minMapQual, minBaseQual;     /* 20, 20 */
    int32_t minPosterior;                /* 5 */
    double sbThreshold, scThreshold, abThreshold, minVarFreq;                                                 /* 1e-3, 0.95, 1e-3, 0.05 */
    int32_t badReadsWindow, badReadsThreshold, rmsmqThreshold, qdThreshold, hapScoreThreshold;                /* 11, 15, 40, 10, 4 */
    int32_t refCallBlockSize;            /* 1000 (outputRefCalls=1: window.py:172-219, variantcaller.pyx:584-607) */
    /* assemble=1 (variantcaller.pyx:496-519, assembler.pyx:1391-1476): tiles of assemblyRegionSize every max(100, min(1000, size / 2)) bases */
    int32_t assemblyRegionSize, assembleAll, assembleBadReads, assembleBrokenPairs, assemblerKmerSize, noCycles;   /* 1500, 1, 1, 0, 15, 0 */
} plat_caller_options;

typedef struct plat_caller_stats {
    int64_t n_regions, n_reads, n_candidate_records, n_variants, n_windows, n_windows_called, n_records;
    int64_t n_windows_greedy;            /* windows whose haplotypes came from the greedy filter (variantFilter.pyx:440-506) */
    int64_t n_windows_failed;            /* windows skipped after an error, as the reference's try/except does (:568-615) */
    double seconds_total;                /* wall time of the call */
    double seconds_host;                 /* sum over worker threads of time spent in host stages */
    double seconds_device_wait;          /* sum over worker threads of time spent waiting for the device */
    /* sum over worker threads, per stage: 0 upload of the reads, 1 candidate scan, 2 candidates -> variants -> windows ->
     * haplotypes (host), 3 greedy haplotype filter rounds, 4 window batch (pack, likelihoods, EM), 5 posteriors,
     * 6 read statistics + genotype calls, 7 INFO / FILTER / text */
    double seconds_stage[8];
    /* plat_call_regions_stream: sum over loader threads of time inside the source's load function; sum over worker threads of time
     * spent waiting for a loaded chunk (a source slower than the callers shows up here); input bytes handed over (bases + qualities) */
    double seconds_load, seconds_source_wait;
    int64_t input_bytes;
    int64_t n_assembly_tiles, n_assembler_variants;   /* assemble=1: tiles assembled, variants they returned */
    int64_t n_refcall_records;                        /* outputRefCalls=1: REFCALL lines among n_records */
    double seconds_assemble;                          /* sum over worker threads: tiles -> device assembler -> variants */
    int64_t n_pairs;                                  /* (read, haplotype) pairs of the called windows: entries of the likelihood arrays */
    /* plat_caller_count_cells(c, 1) only (else 0): the plat_align_stats of every likelihood batch of the call, greedy rounds included,
     * summed -- fastAlignmentRoutine calls the reference would make for these windows and their band cells (16 * read length each:
     * the GCUPS numerator of SURVEY 8(d)), and the same for the DPs the device ran */
    int64_t n_dp_reference, cells_reference, n_dp_launched, cells_launched;
    /* ... and, same switch, the likelihood batches' shapes and live kernel times (plat_profile, HIP events on the worker's stream):
     * number of batches, their haplotype / read bytes, reads, algorithmic bytes of the DP launches (4 * len + 34 per DP), and the
     * summed durations of k_seed and k_dp_jobs -- what a roofline entry of the region pipeline's largest kernels is computed from */
    int64_t n_align_batches, align_hap_bytes, align_read_bytes, align_reads, align_dp_bytes;
    double seconds_kernel_seed, seconds_kernel_dp;
    double seconds_kernel_sweep, seconds_kernel_pairs;   /* the two kernels of the seeding stage on their own (seed = sweep + pairs) */
    /* stage B (candidates -> variants -> windows -> haplotypes -> window batch): regions the device did (plat_stage_b_batch), regions
     * and single windows it left to the host's code (cohorts, assembly and reference-call runs never go to the device: not counted) */
    int64_t n_regions_stage_b_device, n_regions_stage_b_host, n_windows_stage_b_host;
    int64_t n_regions_dict_replay_device;   /* of the device's regions: those whose order needed the Python-2 dictionaries replayed (on the device) */
    double seconds_worker_cpu;              /* sum over worker threads of the CPU time (CLOCK_THREAD_CPUTIME_ID) their chunks took: next to seconds_host (wall
                                             * time outside waits) it says whether the workers had their cores to themselves */
    /* plat_caller_count_cells(c, 1) only: the two widest kernels of a chunk's read table, live (HIP events on the worker's stream): summed
     * durations, launches and ALGORITHMIC bytes -- k_unpack_pieces: one packed byte in, a base and a quality out per base of the resident
     * tables expanded; k_candidates: the bases of the chunk's reads once (what the scan has to read) */
    double seconds_kernel_unpack, seconds_kernel_candidates;
    int64_t unpack_bytes, candidates_bytes, n_unpack_launches, n_candidates_launches;
    /* plat_caller_count_cells(c, 1) only (round 6): EVERY kernel of the chunks, live -- summed launch durations (ms) and launches per kernel id
     * (PLAT_KT_* of platypus_mi355x.h, names from plat_kernel_timer_name): the ranking a roofline entry's kernel is chosen by */
    double kernel_ms[32];
    int64_t kernel_launches[32];
} plat_caller_stats;

typedef struct plat_caller plat_caller;

void plat_caller_default_options(plat_caller_options* out);
/* n_workers worker threads (each owns a plat_ctx + stream on `device`); regions_per_chunk regions go through the device
 * stages together (0 = default). */
int plat_caller_create(int device, int n_workers, int regions_per_chunk, plat_caller** out);
int plat_caller_destroy(plat_caller* c);
/* Measurement switch: on = every likelihood batch goes through the synchronous plat_align_window_batch, whose statistics kernels count
 * the reference's DPs and band cells (same results, two small read-backs per batch: not for timed runs); the sums are left in
 * plat_caller_stats.  Off (default) = the asynchronous entry point, nothing counted. */
int plat_caller_count_cells(plat_caller* c, int on);
/* Measurement switch: the calls that follow time ONE kernel of the loop (id = PLAT_KT_* of platypus_mi355x.h; < 0: none) with HIP events around its
 * launches, inside the ordinary asynchronous calls: stats.kernel_ms[id] / kernel_launches[id] then hold its summed duration and launches. */
int plat_caller_time_kernel(plat_caller* c, int id);
/* Calls every region; the record lines of all regions, in region order, are returned as one malloc'ed,
 * NUL-terminated buffer (*out_text, *out_len without the NUL; free with plat_caller_free).  options->rlen is
 * left at the last region's value, as after the reference's last callVariantsInRegion. */
int plat_call_regions(plat_caller* c, const plat_region* regions, int n_regions, int n_samples,
                      const char* const* sample_names, plat_caller_options* options, char** out_text,
                      size_t* out_len, plat_caller_stats* stats /* may be NULL */);
/* The same call for a region list that is LOADED ON DEMAND -- the reference's own shape: PlatypusSingleProcess.run walks its share of the
 * region list and loads the reads of one region at a time (loadBAMData, variantcaller.pyx:935-1012; bufferSize = 100 kb) before calling
 * it.  `load` is the loader: called from n_loader_threads threads of the library, concurrently for different regions, it fills *out for
 * region `index` (0-based position in the list) with pointers into memory the SOURCE owns; `slot` in [0, n_slots) names the reusable
 * buffer the source should use -- the library hands a slot out again only when the chunk of the region that held it is finished, so
 * n_slots bounds the reads in flight (a bounded queue: loaders run ahead of the callers by at most the free slots).  Needs
 * n_slots >= regions_per_chunk * (n_workers + 1).  A load that returns non-zero ends the call with that code.  Regions are called in
 * chunks in list order; options->rlen follows the regions in list order exactly as in plat_call_regions; the text is the same. */
typedef int (*plat_region_load_fn)(void* user, int index, int slot, plat_region* out);
int plat_call_regions_stream(plat_caller* c, int n_regions, int n_samples, const char* const* sample_names,
                             plat_caller_options* options, plat_region_load_fn load, void* user, int n_slots,
                             int n_loader_threads, char** out_text, size_t* out_len, plat_caller_stats* stats /* may be NULL */);
/* The merge of the per-process record files (runner.py:301-352: a k-way heap merge of the workers' temporary VCFs by chromosome and
 * position; key of a chromosome name runner.py:47-50: int(name.upper().strip("CHR")) if that parses, else the name, integers first):
 * n texts of record lines, each already in that order, -> one text in that order (ties: the earlier text first).  For the job's one
 * exchange: every rank's record text gathered to rank 0, merged there.  *out_text is malloc'ed (plat_caller_free). */
int plat_merge_record_texts(const char* const* texts, const size_t* lengths, int n, char** out_text, size_t* out_len);
/* The same merge when the texts are made of whole REGIONS that do not interleave (disjoint regions, each one's records in order -- what
 * plat_call_regions returns): then the merged text is a permutation of the regions' blocks and no line has to be looked at.
 * plat_caller_region_text_lengths: bytes of every region's record text in the text the last call on `c` returned, in list order (the
 * blocks lie back to back).  plat_merge_region_blocks: n blocks src[i][0 .. len[i]) copied to offsets at[i] (ascending, not
 * overlapping, inside `total`) of a fresh block (*out_text, NUL-terminated, plat_caller_free) on up to 16 threads; the caller puts the
 * blocks in (chromosome key, start) order and falls back to plat_merge_record_texts where regions overlap. */
int plat_caller_region_text_lengths(const plat_caller* c, int64_t* out, int n);
int plat_merge_region_blocks(int n, const char* const* src, const size_t* len, const size_t* at, size_t total, char** out_text);
void plat_caller_free(void* p);
/* Human-readable message of the last error of a failing plat_call_regions on this caller. */
const char* plat_caller_last_error(const plat_caller* c);

#ifdef __cplusplus
}
#endif
#endif /* PLATYPUS_CALLER_H */
