// plat_ctx.hip -- context, memory helpers and error strings of libplat_mi355x.so.
#include <time.h>
#include <sys/prctl.h>
#include <math.h>

#include <algorithm>
#include "plat_internal.hpp"

PLAT_EXPORT int plat_abi_version(void) { return PLAT_ABI_VERSION; }

PLAT_EXPORT const char* plat_strerror(int code) {
    switch (code) {
        case PLAT_OK: return "ok";
        case PLAT_ERR_INVALID: return "invalid argument";
        case PLAT_ERR_HIP: return "HIP runtime error (see plat_last_hip_error)";
        case PLAT_ERR_NOMEM: return "out of device memory";
        case PLAT_ERR_HAP_TOO_LONG: return "haplotype is too long (max allowed length is 16384)";
        case PLAT_ERR_HAP_TOO_SHORT: return "haplotype shorter than read length + 15";
        case PLAT_ERR_UNSUPPORTED: return "option not supported by the device path";
        case PLAT_ERR_BAD_HINTS: return "plat_batch_hints do not cover the batch";
        case PLAT_ERR_NO_DEVICE: return "no usable gfx950 device";
        case PLAT_ERR_OVERFLOW: return "output capacity too small";
        case PLAT_ERR_BAD_INPUT: return "input failed device-side validation";
        default: return "unknown error";
    }
}

PLAT_EXPORT int plat_device_count(int* out_count) {
    if (!out_count) return PLAT_ERR_INVALID;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *out_count = 0; return PLAT_ERR_NO_DEVICE; }
    *out_count = n;
    return PLAT_OK;
}

PLAT_EXPORT int plat_ctx_create(int device, plat_ctx** out_ctx) {
    if (!out_ctx) return PLAT_ERR_INVALID;
    *out_ctx = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return PLAT_ERR_NO_DEVICE;
    plat_ctx* ctx = new plat_ctx();
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess) { delete ctx; return PLAT_ERR_NO_DEVICE; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { delete ctx; return PLAT_ERR_NO_DEVICE; }
    ctx->n_cu = prop.multiProcessorCount;
    // LDS one workgroup may ask for (160 KB on gfx950; the opt-in figure where the runtime reports a smaller default)
    ctx->lds_max = std::max<size_t>(prop.sharedMemPerBlock, std::max<size_t>(prop.sharedMemPerBlockOptin, prop.maxSharedMemoryPerMultiProcessor));
    // probMapRight table: host libm, identical to the reference's own log/exp (chaplotype.pyx:621)
    double lut[256];
    const double mLTOT = -0.23025850929940459;
    for (int q = 0; q < 256; ++q) lut[q] = log(1.0 - exp(mLTOT * q));
    hipError_t e = hipMalloc(&ctx->d_mapq_lut, sizeof(lut));
    if (e == hipSuccess) e = hipMemcpy(ctx->d_mapq_lut, lut, sizeof(lut), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipHostMalloc((void**)&ctx->h_readback, 64 * sizeof(int64_t));
    if (e == hipSuccess) e = hipHostMalloc((void**)&ctx->h_sticky, 8 * sizeof(int64_t), hipHostMallocMapped);
    if (e == hipSuccess) { ctx->h_sticky[0] = 0; e = hipHostGetDevicePointer(&ctx->d_sticky, ctx->h_sticky, 0); }
    if (e == hipSuccess) { hipEvent_t ev = nullptr; e = hipEventCreateWithFlags(&ev, hipEventBlockingSync | hipEventDisableTiming); ctx->sync_event = ev; }
    if (e != hipSuccess) { delete ctx; return PLAT_ERR_HIP; }
    *out_ctx = ctx;
    return PLAT_OK;
}

PLAT_EXPORT int plat_ctx_destroy(plat_ctx* ctx) {
    if (!ctx) return PLAT_ERR_INVALID;
    hipError_t e;
    plat_scratch* all[] = {&ctx->hapw, &ctx->tile, &ctx->codes, &ctx->rinfo, &ctx->hap_flags, &ctx->pair_rec,
                           &ctx->jobs, &ctx->job_score, &ctx->counters, &ctx->asm_scratch, &ctx->tb, &ctx->slow, &ctx->dense, &ctx->pop_scratch, &ctx->seedbase, &ctx->merge_tab, &ctx->seedmap, &ctx->seedstate, &ctx->asm_sig};
    for (plat_scratch* s : all)
        if (s->ptr) { e = hipFree(s->ptr); (void)e; }
    if (ctx->d_mapq_lut) { e = hipFree(ctx->d_mapq_lut); (void)e; }
    if (ctx->d_logfact) { e = hipFree(ctx->d_logfact); (void)e; }
    if (ctx->h_readback) { e = hipHostFree(ctx->h_readback); (void)e; }
    if (ctx->h_sticky) { e = hipHostFree(ctx->h_sticky); (void)e; }
    if (ctx->sync_event) { e = hipEventDestroy((hipEvent_t)ctx->sync_event); (void)e; }
    for (int i = 0; i < 9; ++i)
        if (ctx->ev[i]) { e = hipEventDestroy(ctx->ev[i]); (void)e; }
    for (int i = 0; i < 4; ++i)
        if (ctx->ev_tab[i]) { e = hipEventDestroy(ctx->ev_tab[i]); (void)e; }
    for (int i = 0; i < ctx->kt_pending_n; ++i) { e = hipEventDestroy(ctx->kt_pending[i].a); e = hipEventDestroy(ctx->kt_pending[i].b); (void)e; }
    for (int i = 0; i < ctx->kt_pool_n; ++i) { e = hipEventDestroy(ctx->kt_pool[i]); (void)e; }
    delete ctx;
    return PLAT_OK;
}

PLAT_EXPORT int plat_profile_enable(plat_ctx* ctx, int on) {
    if (!ctx) return PLAT_ERR_INVALID;
    if (on && !ctx->ev[0])
        for (int i = 0; i < 9; ++i) PLAT_HIP(ctx, hipEventCreate(&ctx->ev[i]));
    if (on && !ctx->ev_tab[0])
        for (int i = 0; i < 4; ++i) PLAT_HIP(ctx, hipEventCreate(&ctx->ev_tab[i]));
    ctx->profile = on ? 1 : 0;
    ctx->ev_valid_align = ctx->ev_valid_geno = ctx->ev_valid_unpack = ctx->ev_valid_cand = 0;
    return PLAT_OK;
}

PLAT_EXPORT int plat_profile_last(plat_ctx* ctx, plat_profile* out) {
    if (!ctx || !out) return PLAT_ERR_INVALID;
    memset(out, 0, sizeof(*out));
    if (!ctx->profile) return PLAT_ERR_INVALID;
    if (ctx->ev_valid_align) {
        PLAT_HIP(ctx, hipEventSynchronize(ctx->ev[4]));
        PLAT_HIP(ctx, hipEventElapsedTime(&out->ms_prepare, ctx->ev[0], ctx->ev[1]));
        PLAT_HIP(ctx, hipEventElapsedTime(&out->ms_seed, ctx->ev[1], ctx->ev[2]));
        PLAT_HIP(ctx, hipEventElapsedTime(&out->ms_seed_kernel, ctx->ev[1], ctx->ev[5]));
        if (ctx->ev_split) {
            PLAT_HIP(ctx, hipEventElapsedTime(&out->ms_sweep, ctx->ev[1], ctx->ev[8]));
            PLAT_HIP(ctx, hipEventElapsedTime(&out->ms_pairs, ctx->ev[8], ctx->ev[5]));
        }
        PLAT_HIP(ctx, hipEventElapsedTime(&out->ms_dp, ctx->ev[2], ctx->ev[3]));
        PLAT_HIP(ctx, hipEventElapsedTime(&out->ms_finalize, ctx->ev[3], ctx->ev[4]));
        out->dp_jobs = ctx->prof_dp_jobs;
        out->dp_alg_bytes = ctx->prof_dp_bytes;
    }
    if (ctx->ev_valid_geno) {
        PLAT_HIP(ctx, hipEventSynchronize(ctx->ev[7]));
        PLAT_HIP(ctx, hipEventElapsedTime(&out->ms_genotype, ctx->ev[6], ctx->ev[7]));
    }
    if (ctx->ev_valid_unpack) {
        PLAT_HIP(ctx, hipEventSynchronize(ctx->ev_tab[1]));
        PLAT_HIP(ctx, hipEventElapsedTime(&out->ms_unpack, ctx->ev_tab[0], ctx->ev_tab[1]));
    }
    if (ctx->ev_valid_cand) {
        PLAT_HIP(ctx, hipEventSynchronize(ctx->ev_tab[3]));
        PLAT_HIP(ctx, hipEventElapsedTime(&out->ms_candidates, ctx->ev_tab[2], ctx->ev_tab[3]));
    }
    return PLAT_OK;
}

PLAT_EXPORT const char* plat_kernel_timer_name(int id) {
    static const char* names[PLAT_KT_COUNT] = {
        "k_candidates", "k_candidates_merge", "k_candidates_filter", "k_unpack_pieces", "k_concat_tables", "k_copy_pieces", "k_gather_reads",
        "k_sb_variants", "k_sb_windows", "k_sb_haps_rank", "k_sb_prefix", "k_sb_scan", "k_sb_haps_write", "k_sb_reads", "k_validate", "k_tile_scan",
        "k_prep_reads", "k_sweep", "k_pairs", "k_seed_slow", "k_dp_jobs", "k_finalize", "k_genotype", "k_haplotype_score", "k_em",
        "k_variant_posterior", "k_variant_read_stats", "k_variant_info", "k_genotype_call", "k_assemble", "k_read_qc", "other"};
    return id >= 0 && id < PLAT_KT_COUNT ? names[id] : nullptr;
}

PLAT_EXPORT int plat_kernel_timer_only(plat_ctx* ctx, int id) {
    if (!ctx || id >= PLAT_KT_COUNT) return PLAT_ERR_INVALID;
    ctx->kt_single = id < 0 ? -1 : id;
    return PLAT_OK;
}

PLAT_EXPORT int plat_kernel_times(plat_ctx* ctx, double* out_ms, int64_t* out_launches) {
    if (!ctx || !out_ms || !out_launches) return PLAT_ERR_INVALID;
    for (int k = 0; k < ctx->kt_pending_n; ++k) {
        plat_ctx::KtPair& q = ctx->kt_pending[k];
        if (q.closed) {
            PLAT_HIP(ctx, hipEventSynchronize(q.b));
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, q.a, q.b) == hipSuccess) { out_ms[q.id] += (double)ms; out_launches[q.id] += 1; }
        }
        ctx->kt_pool[ctx->kt_pool_n++] = q.a; ctx->kt_pool[ctx->kt_pool_n++] = q.b;
    }
    ctx->kt_pending_n = 0; ctx->kt_open = -1;
    return PLAT_OK;
}

PLAT_EXPORT int plat_last_hip_error(const plat_ctx* ctx) { return ctx ? ctx->last_hip : 0; }

PLAT_EXPORT int plat_malloc(plat_ctx* ctx, size_t bytes, void** out) {
    if (!ctx || !out) return PLAT_ERR_INVALID;
    *out = nullptr;
    PLAT_HIP(ctx, hipSetDevice(ctx->device));
    PLAT_HIP(ctx, hipMalloc(out, bytes ? bytes : 1));
    return PLAT_OK;
}

PLAT_EXPORT int plat_free(plat_ctx* ctx, void* p) {
    if (!ctx) return PLAT_ERR_INVALID;
    if (p) PLAT_HIP(ctx, hipFree(p));
    return PLAT_OK;
}

PLAT_EXPORT int plat_memcpy_h2d(plat_ctx* ctx, void* dst, const void* src, size_t bytes, void* stream) {
    if (!ctx || (!dst && bytes) || (!src && bytes)) return PLAT_ERR_INVALID;
    if (bytes) PLAT_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return PLAT_OK;
}

PLAT_EXPORT int plat_memcpy_d2d(plat_ctx* ctx, void* dst, const void* src, size_t bytes, void* stream) {
    if (!ctx || (!dst && bytes) || (!src && bytes)) return PLAT_ERR_INVALID;
    if (bytes) PLAT_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return PLAT_OK;
}

PLAT_EXPORT int plat_memcpy_d2h(plat_ctx* ctx, void* dst, const void* src, size_t bytes, void* stream) {
    if (!ctx || (!dst && bytes) || (!src && bytes)) return PLAT_ERR_INVALID;
    if (bytes) PLAT_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return PLAT_OK;
}

PLAT_EXPORT int plat_memset(plat_ctx* ctx, void* dst, int value, size_t bytes, void* stream) {
    if (!ctx || (!dst && bytes)) return PLAT_ERR_INVALID;
    if (bytes) PLAT_HIP(ctx, hipMemsetAsync(dst, value, bytes, (hipStream_t)stream));
    return PLAT_OK;
}

PLAT_EXPORT int plat_stream_create(plat_ctx* ctx, void** out_stream) {
    if (!ctx || !out_stream) return PLAT_ERR_INVALID;
    *out_stream = nullptr;
    PLAT_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st;
    PLAT_HIP(ctx, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    *out_stream = (void*)st;
    return PLAT_OK;
}

PLAT_EXPORT int plat_stream_destroy(plat_ctx* ctx, void* stream) {
    if (!ctx) return PLAT_ERR_INVALID;
    if (stream) PLAT_HIP(ctx, hipStreamDestroy((hipStream_t)stream));
    return PLAT_OK;
}

PLAT_EXPORT int plat_host_alloc(plat_ctx* ctx, size_t bytes, void** out) {
    if (!ctx || !out) return PLAT_ERR_INVALID;
    *out = nullptr;
    PLAT_HIP(ctx, hipSetDevice(ctx->device));
    PLAT_HIP(ctx, hipHostMalloc(out, bytes ? bytes : 1));
    return PLAT_OK;
}

PLAT_EXPORT int plat_host_free(plat_ctx* ctx, void* p) {
    if (!ctx) return PLAT_ERR_INVALID;
    if (p) PLAT_HIP(ctx, hipHostFree(p));
    return PLAT_OK;
}

PLAT_EXPORT int plat_sync_poll_us(plat_ctx* ctx, int microseconds) {
    if (!ctx || microseconds < 0) return PLAT_ERR_INVALID;
    ctx->sync_poll_ns = (long)microseconds * 1000L;
    return PLAT_OK;
}

PLAT_EXPORT int plat_stream_sync(plat_ctx* ctx, void* stream) {
    if (!ctx) return PLAT_ERR_INVALID;
    // The waiting thread SLEEPS instead of spinning on the stream: a caller with many worker threads, most of them waiting for the
    // device at any time, would otherwise burn the cores its host stages need.  The runtime's own blocking wait (an event with
    // hipEventBlockingSync) still polls for a good while before it blocks -- measured in the region loop: 0.085 ms of CPU per region
    // inside 0.09 ms of waiting -- so the default is a poll of the event every PLAT_SYNC_POLL_US microseconds (40) with the thread
    // asleep in between; PLAT_SYNC_POLL_US=0: hipEventSynchronize; PLAT_SYNC_SPIN=1: hipStreamSynchronize, the runtime's default wait.
    static const bool spin = [] { const char* e = getenv("PLAT_SYNC_SPIN"); return e && e[0] == '1'; }();
    static const long env_poll_ns = [] { const char* e = getenv("PLAT_SYNC_POLL_US"); const long v = e ? atol(e) : -1; return v < 0 ? -1L : v * 1000L; }();
    const long poll_ns = env_poll_ns >= 0 ? env_poll_ns : ctx->sync_poll_ns;          // (the environment wins over plat_sync_poll_us: measurements)
    if (spin || !ctx->sync_event) PLAT_HIP(ctx, hipStreamSynchronize((hipStream_t)stream));
    else {
        PLAT_HIP(ctx, hipEventRecord((hipEvent_t)ctx->sync_event, (hipStream_t)stream));
        if (poll_ns == 0) PLAT_HIP(ctx, hipEventSynchronize((hipEvent_t)ctx->sync_event));
        else {
            // (the default timer slack of 50 us would double every nap: 2 us while THIS wait lasts; the calling thread may be the
            //  application's own -- worker 0 of plat_call_regions runs on the caller -- so its slack is put back before returning)
            const int slack0 = prctl(PR_GET_TIMERSLACK, 0, 0, 0, 0);
            bool napped = false;
            hipError_t bad = hipSuccess;
            for (;;) {
                const hipError_t q = hipEventQuery((hipEvent_t)ctx->sync_event);
                if (q == hipSuccess) break;
                if (q != hipErrorNotReady) { bad = q; break; }
                (void)hipGetLastError();                                     // (hipErrorNotReady is sticky in hipGetLastError otherwise)
                if (!napped) { if (slack0 > 2000) prctl(PR_SET_TIMERSLACK, 2000UL, 0, 0, 0); napped = true; }
                timespec ts{0, poll_ns};
                nanosleep(&ts, nullptr);
            }
            if (napped && slack0 > 2000) prctl(PR_SET_TIMERSLACK, (unsigned long)slack0, 0, 0, 0);
            if (bad != hipSuccess) PLAT_HIP(ctx, bad);
        }
    }
    if (ctx->h_sticky && ctx->h_sticky[0] != 0) {          // error recorded by an asynchronous call since the last sync
        const int rc = (int)ctx->h_sticky[0];
        ctx->h_sticky[0] = 0;
        return rc;
    }
    return PLAT_OK;
}
