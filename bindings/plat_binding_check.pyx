# cython: language_level=3
# plat_binding_check.pyx -- compiles bindings/cplat.pxd against include/platypus_mi355x.h and links libplat_mi355x.so:
# what a Cython module of the reference does with the library before its first batch (tests/test_binding_cpu.py builds and
# imports this module; nothing here needs a GPU).
from libc.stdint cimport int32_t, int64_t, uint8_t
from libc.string cimport memset
cimport cplat


def abi_version():
    return cplat.plat_abi_version(), cplat.PLAT_ABI_VERSION, cplat.PLAT_BLOB_PAD


def strerror(int code):
    return cplat.plat_strerror(code).decode("ascii")


def device_count():
    cdef int n = -1
    cdef int rc = cplat.plat_device_count(&n)
    return rc, n


def describe_window_batch(int n_windows, int n_haps, int n_reads):
    """Fill a plat_window_batch the way Population.setup would (cpopulation.pyx:197-309): counts here, device pointers from
    plat_malloc in a real binding."""
    cdef cplat.plat_window_batch b
    cdef cplat.plat_batch_hints h
    cdef cplat.plat_align_stats st
    memset(&b, 0, sizeof(b))
    memset(&h, 0, sizeof(h))
    memset(&st, 0, sizeof(st))
    b.n_windows, b.n_haps, b.n_reads = n_windows, n_haps, n_reads
    h.n_pairs = <int64_t>n_haps * n_reads
    return dict(size=sizeof(b), hints=sizeof(h), stats=sizeof(st), n=(b.n_windows, b.n_haps, b.n_reads), pairs=h.n_pairs,
                null_batch_rc=cplat.plat_align_window_batch(NULL, &b, 0, 0, NULL, NULL, &st, NULL))


def create_context(int device=0):
    """plat_ctx_create: PLAT_ERR_NO_DEVICE on a host without a GPU (never a CPU fallback)."""
    cdef cplat.plat_ctx* ctx = NULL
    cdef int rc = cplat.plat_ctx_create(device, &ctx)
    if rc == cplat.PLAT_OK:
        cplat.plat_ctx_destroy(ctx)
    return rc, cplat.PLAT_ERR_NO_DEVICE
