// plat_internal.hpp -- context, error handling and scratch management shared by the .hip files.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/platypus_mi355x.h"

#define PLAT_EXPORT extern "C" __attribute__((visibility("default")))
#define PLAT_GRID_Y_MAX 65535          /* HIP's limit on gridDim.y: launches that put a list in y go in batches of this many */

struct plat_scratch {
    void* ptr = nullptr;
    size_t cap = 0;
};

struct plat_ctx {
    int device = 0;
    int last_hip = 0;
    int n_cu = 0;
    size_t lds_max = 0;
    double* d_mapq_lut = nullptr;       // log(1 - exp(mLTOT*mapq)), chaplotype.pyx:621, host libm
    double* d_logfact = nullptr;        // logFactorial(0 .. 4095) then log(1 .. 4096) (platypusutils.pyx:178-191), host libm: plat_variant_info_batch
    // device scratch (grow-only)
    plat_scratch hapw, tile, codes, rinfo, hap_flags, pair_rec, jobs, job_score, counters, asm_scratch, tb, slow, dense, pop_scratch, seedbase, merge_tab, seedmap, seedstate, asm_sig;
    unsigned long long asm_epoch = 0;   // counts the (re)allocations of asm_scratch AND the changes of the slice layout between launches: part of the signature k_assemble leaves in asm_sig
    unsigned long long asm_last_layout = 0;   // the previous launch's slice layout (sizes asm_carve is given)
    bool sb_attr_set = false;           // k_sb_variants' dynamic-LDS limit has been raised on this context's device
    bool asm_last_kept = false;         // ... and whether it maintained asm_sig (a launch without it dirties slices behind the signatures' back)
    // pinned host read-back area
    int64_t* h_readback = nullptr;
    // asynchronous entry points: first device-side error since the last plat_stream_sync (pinned, device-visible)
    int64_t* h_sticky = nullptr;
    void* d_sticky = nullptr;
    void* sync_event = nullptr;         // hipEvent_t with hipEventBlockingSync: plat_stream_sync sleeps on it
    long sync_poll_ns = 40000;          // ... polling it this often (plat_sync_poll_us)
    // optional live timing: events 0..5 bracket prepare|seed|dp|finalize, 6..7 bracket genotype
    int profile = 0;
    hipEvent_t ev[9] = {};          // + 8: between k_sweep and k_pairs
    int ev_valid_align = 0, ev_valid_geno = 0, ev_split = 0;
    hipEvent_t ev_tab[4] = {};      // around k_unpack_pieces (0, 1) and k_candidates (2, 3)
    int ev_valid_unpack = 0, ev_valid_cand = 0;
    int64_t prof_dp_jobs = 0, prof_dp_bytes = 0;
    // generic kernel timers (plat_kernel_times)
    struct KtPair { int id; hipEvent_t a, b; bool closed; };
    static constexpr int KT_PENDING = 512;
    KtPair kt_pending[KT_PENDING];
    hipEvent_t kt_pool[2 * KT_PENDING + 2];
    int kt_pending_n = 0, kt_pool_n = 0, kt_open = -1;
    int kt_single = -1;                  // plat_kernel_timer_only: this one kernel is timed even while the profile is off
    double kt_ms[PLAT_KT_COUNT] = {};
    int64_t kt_launches[PLAT_KT_COUNT] = {};
};

#define PLAT_HIP(ctx, call)                                   \
    do {                                                      \
        hipError_t _e = (call);                               \
        if (_e != hipSuccess) {                               \
            if (ctx) (ctx)->last_hip = (int)_e;               \
            return _e == hipErrorOutOfMemory ? PLAT_ERR_NOMEM : PLAT_ERR_HIP; \
        }                                                     \
    } while (0)

// ---- generic live kernel timers (plat_kernel_times): a pair of HIP events around a launch while the profile is on, resolved and summed per
// kernel id when the caller asks.  Off: one test of ctx->profile per launch.
static inline void plat_kt_mark(plat_ctx* ctx, int id, hipStream_t st, bool end) {
    if (id < 0 || id >= PLAT_KT_COUNT || !(ctx->profile || ctx->kt_single == id)) return;
    if (!end) {
        if (ctx->kt_pending_n >= plat_ctx::KT_PENDING) return;               // (too many unresolved pairs: this launch is not timed -- and takes no events)
        hipEvent_t a = nullptr, b = nullptr;
        if (ctx->kt_pool_n >= 2) { a = ctx->kt_pool[--ctx->kt_pool_n]; b = ctx->kt_pool[--ctx->kt_pool_n]; }
        else if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
        ctx->kt_pending[ctx->kt_pending_n] = {id, a, b, false};
        ctx->kt_open = ctx->kt_pending_n++;
        (void)hipEventRecord(a, st);
    } else if (ctx->kt_open >= 0 && ctx->kt_pending[ctx->kt_open].id == id) {
        (void)hipEventRecord(ctx->kt_pending[ctx->kt_open].b, st);
        ctx->kt_pending[ctx->kt_open].closed = true;
        ctx->kt_open = -1;
    }
}
#define PLAT_KT_BEGIN(ctx, id, st) plat_kt_mark(ctx, id, (hipStream_t)(st), false)
#define PLAT_KT_END(ctx, id, st) plat_kt_mark(ctx, id, (hipStream_t)(st), true)

#define PLAT_EV_TAB(ctx, i, st) do { if ((ctx)->profile) PLAT_HIP(ctx, hipEventRecord((ctx)->ev_tab[i], st)); } while (0)
#define PLAT_EV(ctx, i, st) do { if ((ctx)->profile) PLAT_HIP(ctx, hipEventRecord((ctx)->ev[i], st)); } while (0)

static inline int plat_reserve(plat_ctx* ctx, plat_scratch& s, size_t bytes) {
    if (bytes <= s.cap) return PLAT_OK;
    if (s.ptr) { hipError_t e = hipFree(s.ptr); (void)e; s.ptr = nullptr; s.cap = 0; }
    size_t want = bytes + bytes / 4 + 4096;
    hipError_t e = hipMalloc(&s.ptr, want);
    if (e != hipSuccess) { ctx->last_hip = (int)e; s.ptr = nullptr; return PLAT_ERR_NOMEM; }
    s.cap = want;
    return PLAT_OK;
}
