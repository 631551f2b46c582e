"""ctypes binding of libplat_mi355x.so (include/platypus_mi355x.h).

The HIP library is the product path.  There is NO CPU fallback: if the shared object is missing it is
built with hipcc (gfx950); if that is impossible, or if no GPU is present when a device entry point is
called, an exception is raised.
"""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libplat_mi355x.so")
CSRC = os.path.join(HERE, "csrc")

PLAT_BLOB_PAD = 32
PLAT_ABI_VERSION = 1

ERRORS = {
    0: "PLAT_OK", -1: "PLAT_ERR_INVALID", -2: "PLAT_ERR_HIP", -3: "PLAT_ERR_NOMEM", -4: "PLAT_ERR_HAP_TOO_LONG",
    -5: "PLAT_ERR_HAP_TOO_SHORT", -6: "PLAT_ERR_UNSUPPORTED", -7: "PLAT_ERR_NO_DEVICE", -8: "PLAT_ERR_OVERFLOW",
    -9: "PLAT_ERR_BAD_INPUT", -10: "PLAT_ERR_BAD_HINTS",
}


class PlatypusDeviceError(RuntimeError):
    def __init__(self, code, msg, where=""):
        self.code = code
        super().__init__("%s%s (%d): %s" % (where + ": " if where else "", ERRORS.get(code, "?"), code, msg))


class WindowBatch(C.Structure):
    _fields_ = [("n_windows", C.c_int32), ("n_haps", C.c_int32), ("n_reads", C.c_int32), ("_pad", C.c_int32),
                ("win_hap_begin", C.c_void_p), ("win_read_begin", C.c_void_p), ("win_start", C.c_void_p),
                ("win_end", C.c_void_p), ("win_flank", C.c_void_p), ("pair_off", C.c_void_p),
                ("hap_seq", C.c_void_p), ("hap_off", C.c_void_p), ("read_seq", C.c_void_p),
                ("read_qual", C.c_void_p), ("read_off", C.c_void_p), ("read_pos", C.c_void_p),
                ("read_end", C.c_void_p), ("read_mapq", C.c_void_p), ("read_flags", C.c_void_p),
                ("read_kind", C.c_void_p)]


class AlignStats(C.Structure):
    _fields_ = [("n_pairs", C.c_int64), ("n_pairs_aligned", C.c_int64), ("n_dp_launched", C.c_int64),
                ("n_dp_reference", C.c_int64), ("cells_reference", C.c_int64), ("cells_launched", C.c_int64),
                ("n_seed_fallback", C.c_int64), ("_reserved", C.c_int64)]


class Profile(C.Structure):
    _fields_ = [("ms_prepare", C.c_float), ("ms_seed", C.c_float), ("ms_dp", C.c_float), ("ms_finalize", C.c_float),
                ("ms_genotype", C.c_float), ("ms_seed_kernel", C.c_float), ("dp_jobs", C.c_int64), ("dp_alg_bytes", C.c_int64),
                ("ms_sweep", C.c_float), ("ms_pairs", C.c_float), ("ms_unpack", C.c_float), ("ms_candidates", C.c_float)]


class BatchHints(C.Structure):
    _fields_ = [("max_hap_len", C.c_int32), ("max_read_len", C.c_int32), ("max_reads_per_window", C.c_int32),
                ("_pad", C.c_int32), ("n_pairs", C.c_int64), ("hap_blob_len", C.c_int64), ("read_blob_len", C.c_int64),
                ("extra_jobs_cap", C.c_int64)]


class AssemblyBatch(C.Structure):
    _fields_ = [("n_regions", C.c_int32), ("n_reads", C.c_int32), ("ref_seq", C.c_void_p), ("ref_off", C.c_void_p),
                ("ref_start", C.c_void_p), ("assem_start", C.c_void_p), ("assem_end", C.c_void_p),
                ("reg_read_begin", C.c_void_p), ("read_seq", C.c_void_p), ("read_qual", C.c_void_p),
                ("read_off", C.c_void_p)]


class AssemblyHints(C.Structure):
    _fields_ = [("max_ref_len", C.c_int32), ("max_reads_per_region", C.c_int32), ("max_positions", C.c_int64)]


class CandidateBatch(C.Structure):
    _fields_ = [("n_regions", C.c_int32), ("n_reads", C.c_int32), ("ref_seq", C.c_void_p), ("ref_off", C.c_void_p),
                ("ref_seq_start", C.c_void_p), ("contig_len", C.c_void_p), ("read_seq", C.c_void_p),
                ("read_qual", C.c_void_p), ("read_off", C.c_void_p), ("read_pos", C.c_void_p), ("read_flags", C.c_void_p),
                ("cigar", C.c_void_p), ("cig_off", C.c_void_p)]


class ReadQCBatch(C.Structure):
    _fields_ = [("n_reads", C.c_int32), ("_pad", C.c_int32), ("read_qual", C.c_void_p), ("read_off", C.c_void_p),
                ("read_pos", C.c_void_p), ("read_mapq", C.c_void_p), ("read_flags", C.c_void_p), ("chrom_id", C.c_void_p),
                ("mate_chrom_id", C.c_void_p), ("insert_size", C.c_void_p), ("mate_pos", C.c_void_p), ("cigar", C.c_void_p),
                ("cig_off", C.c_void_p), ("stream_of", C.c_void_p)]


class ReadQCOptions(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("min_good_qual_bases", "min_map_qual", "min_base_qual", "trim_overlapping", "trim_adapter",
                                         "trim_read_flank", "trim_soft_clipped", "filter_mate_unmapped", "filter_mate_distant",
                                         "filter_small_insert", "filter_duplicates")]


class InfoStatsBatch(C.Structure):
    _fields_ = [("n_vars", C.c_int32), ("n_ind", C.c_int32)] + [(k, C.c_void_p) for k in (
        "var_window", "var_pos", "var_bam_min", "var_bam_max", "var_n_added", "var_n_removed", "var_added", "var_added_off",
        "var_in_genotype", "minq_off", "good_begin", "good_end", "bad_begin", "bad_end", "read_seq", "read_qual", "read_off",
        "read_pos", "read_end", "read_mapq", "read_flags", "cigar", "cig_off")]


# symbol -> (restype, argtypes): exactly the declarations of include/platypus_mi355x.h
SIGNATURES = {
    "plat_abi_version": (C.c_int, []),
    "plat_strerror": (C.c_char_p, [C.c_int]),
    "plat_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "plat_ctx_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "plat_ctx_destroy": (C.c_int, [C.c_void_p]),
    "plat_last_hip_error": (C.c_int, [C.c_void_p]),
    "plat_malloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "plat_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "plat_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "plat_memcpy_d2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "plat_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "plat_memset": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]),
    "plat_stream_sync": (C.c_int, [C.c_void_p, C.c_void_p]),
    "plat_stream_create": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "plat_stream_destroy": (C.c_int, [C.c_void_p, C.c_void_p]),
    "plat_host_alloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "plat_host_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "plat_gather_reads": (C.c_int, [C.c_void_p, C.c_int64] + [C.c_void_p] * 16),
    "plat_unpack_reads": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "plat_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "plat_profile_last": (C.c_int, [C.c_void_p, C.POINTER(Profile)]),
    "plat_sync_poll_us": (C.c_int, [C.c_void_p, C.c_int]),
    "plat_kernel_timer_name": (C.c_char_p, [C.c_int]),
    "plat_kernel_times": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "plat_kernel_timer_only": (C.c_int, [C.c_void_p, C.c_int]),
    "plat_dp_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "plat_align_window_batch": (C.c_int, [C.c_void_p, C.POINTER(WindowBatch), C.c_int, C.c_int, C.c_void_p,
                                          C.c_void_p, C.POINTER(AlignStats), C.c_void_p]),
    "plat_align_window_batch_async": (C.c_int, [C.c_void_p, C.POINTER(WindowBatch), C.POINTER(BatchHints), C.c_int, C.c_int,
                                                C.c_void_p, C.c_void_p, C.c_void_p]),
    "plat_genotype_window_batch": (C.c_int, [C.c_void_p, C.POINTER(WindowBatch), C.c_int, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p]),
    "plat_em_window_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p]),
    "plat_variant_posterior_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 11),
    "plat_genotype_call_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 16),
    "plat_haplotype_score_batch": (C.c_int, [C.c_void_p, C.POINTER(WindowBatch), C.c_int, C.c_int] + [C.c_void_p] * 6),
    "plat_candidates_batch": (C.c_int, [C.c_void_p, C.POINTER(CandidateBatch), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "plat_candidates_merge_batch": (C.c_int, [C.c_void_p, C.POINTER(CandidateBatch), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "plat_stage_b_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "plat_unpack_reads_pieces": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p]),
    "plat_unpack_reads_pieces_codes": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p,
                                                 C.c_void_p, C.c_void_p, C.c_void_p]),
    "plat_ref_codes": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "plat_candidates_batch_codes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "plat_copy_pieces": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "plat_concat_read_tables": (C.c_int, [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 9 + [C.c_int64] * 3 + [C.c_void_p]),
    "plat_read_qc_batch": (C.c_int, [C.c_void_p, C.POINTER(ReadQCBatch), C.POINTER(ReadQCOptions), C.c_void_p, C.c_void_p, C.c_void_p]),
    "plat_variant_read_stats_batch": (C.c_int, [C.c_void_p, C.POINTER(InfoStatsBatch), C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_void_p, C.c_void_p]),
    "plat_variant_info_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "plat_assemble_batch": (C.c_int, [C.c_void_p, C.POINTER(AssemblyBatch), C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p]),
    "plat_assemble_batch_async": (C.c_int, [C.c_void_p, C.POINTER(AssemblyBatch), C.POINTER(AssemblyHints), C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p]),
}

_lib = None


def build(verbose=False):
    """Compile libplat_mi355x.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    r = subprocess.run(["make", "-C", CSRC, "-j8"], capture_output=True, text=True)
    if verbose:
        print(r.stdout)
    if r.returncode != 0:
        raise RuntimeError("building libplat_mi355x.so failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    return LIB_PATH


def load():
    """Load the HIP library (building it first if needed).  Never falls back to a CPU implementation."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    try:
        # torch is the device-memory/stream plumbing of the Python host and ships its own libamdhip64:
        # import it first so the process holds ONE HIP runtime (loading /opt/rocm's copy first and torch's
        # second leaves this library without a usable device).  A host without torch simply uses /opt/rocm's.
        import torch  # noqa: F401
    except ImportError:
        pass
    _lib = bind(C.CDLL(LIB_PATH))
    return _lib


def bind(lib):
    """Attach the prototypes of include/platypus_mi355x.h to a loaded library exporting that ABI."""
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError here == the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.plat_abi_version() != PLAT_ABI_VERSION:
        raise RuntimeError("libplat_mi355x.so ABI version mismatch")
    return lib


def check(code, where=""):
    if code != 0:
        msg = load().plat_strerror(code).decode()
        raise PlatypusDeviceError(code, msg, where)
