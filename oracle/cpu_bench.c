/*
 * cpu_bench.c -- timing harness for bench.py's `cpu_baseline` leg (TEST/BENCH INFRASTRUCTURE ONLY).
 *
 * Times the UNMODIFIED reference kernel `fastAlignmentRoutine` (src/c/align.c:77, C ABI of align.h:8-10)
 * taken from oracle/_ref/libalign_ref.so via dlopen, on padded rows of DP instances, single thread.
 * traceback = 1 reproduces the reference's production behaviour (aln buffers are always allocated when
 * hapFlank > 0: calign.pyx:199-202), traceback = 0 is the score-only mode the GPU path runs.
 * Nothing here is linked into the product library.
 */
#define _POSIX_C_SOURCE 200809L
#include <dlfcn.h>
#include <stdlib.h>
#include <time.h>

typedef int (*fast_align_fn)(const char*, const char*, const char*, int, int, int, int, const char*, char*, char*, int*);

__attribute__((visibility("default")))
double cpu_time_reference_dp(const char* libpath, int n, int lmax, const char* haps, const char* reads,
                             const char* quals, const char* gos, const int* len2, int traceback, int reps,
                             long long* checksum, long long* cells)
{
    void* h = dlopen(libpath, RTLD_NOW);
    if (!h) return -1.0;
    fast_align_fn f = (fast_align_fn)dlsym(h, "fastAlignmentRoutine");
    if (!f) return -2.0;
    char* a1 = (char*)malloc(2 * (size_t)lmax + 32);
    char* a2 = (char*)malloc(2 * (size_t)lmax + 32);
    int fp = 0;
    long long cs = 0, c = 0;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int r = 0; r < reps; ++r)
        for (int j = 0; j < n; ++j) {
            const int L = len2[j];
            cs += f(haps + (size_t)j * (lmax + 15), reads + (size_t)j * lmax, quals + (size_t)j * lmax, L + 15, L,
                    3, 2, gos + (size_t)j * (lmax + 15), traceback ? a1 : NULL, traceback ? a2 : NULL, &fp);
            c += 16ll * L;
        }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    free(a1); free(a2);
    if (checksum) *checksum = cs;
    if (cells) *cells = c;
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
