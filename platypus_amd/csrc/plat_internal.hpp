// plat_internal.hpp -- context, error handling and scratch management shared by the .hip files.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/platypus_mi355x.h"

#define PLAT_EXPORT extern "C" __attribute__((visibility("default")))

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
    // device scratch (grow-only)
    plat_scratch go_blob, pair_rec, jobs, job_score, counters;
    // pinned host read-back area
    int64_t* h_readback = nullptr;
};

#define PLAT_HIP(ctx, call)                                   \
    do {                                                      \
        hipError_t _e = (call);                               \
        if (_e != hipSuccess) {                               \
            if (ctx) (ctx)->last_hip = (int)_e;               \
            return _e == hipErrorOutOfMemory ? PLAT_ERR_NOMEM : PLAT_ERR_HIP; \
        }                                                     \
    } while (0)

static inline int plat_reserve(plat_ctx* ctx, plat_scratch& s, size_t bytes) {
    if (bytes <= s.cap) return PLAT_OK;
    if (s.ptr) { hipError_t e = hipFree(s.ptr); (void)e; s.ptr = nullptr; s.cap = 0; }
    size_t want = bytes + bytes / 4 + 4096;
    hipError_t e = hipMalloc(&s.ptr, want);
    if (e != hipSuccess) { ctx->last_hip = (int)e; s.ptr = nullptr; return PLAT_ERR_NOMEM; }
    s.cap = want;
    return PLAT_OK;
}
