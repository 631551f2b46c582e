// dp_mapping_a.hip -- the OTHER mapping of the banded DP onto a wave, measured against the one the library uses.
//
// SURVEY.md section 7 names two ways to put fastAlignmentRoutine (src/c/align.c:77-586) on a 64-wide wavefront:
//   A  "one band cell per lane, anti-diagonal wavefront, shuffles carry the M/I/D recurrence" (north_star's wording): the reference's
//      8 int16 SSE lanes become 8 GPU lanes, a wave holds 8 alignments, the reference's one-lane shifts (_mm_slli/_mm_srli_si128 by 2)
//      become DPP row shifts (v_mov_b32_dpp row_shr:1 / row_shl:1);
//   B  one GPU lane = one whole alignment, the 8 int16 lanes packed two to a VGPR (csrc/dp_core.hpp): what libplat_mi355x.so runs.
// This program IS mapping A, bit exact: it is compared score by score with mapping B through plat_dp_batch, the library's ROW entry point
// (k_dp_rows: padded rows, one lane per row with strided byte loads -- the config-1 plumbing path, not the tuned one).  The rate to set
// mapping A against is the job-list kernel's, k_dp_jobs with every reference DP executed: `gcups_all_dp` of the bench line (3 970 GCUPS
// on the same 150 bp reads).  It exists so that the choice of B is a measurement, not an assumption (profiles/HISTORY.md, round 5):
// mapping A as written here runs 640 GCUPS -- its loads are one byte per lane and step from L1, which an LDS stage would remove, but
// the instruction count below bounds it at a quarter of B whatever the loads cost.
//
// Why A cannot win here, in instruction counts (score-only mode): per step of 16 band cells a lane of mapping A issues ~52 vector
// instructions for ONE int16 lane of ONE alignment (2 cells): 8 DPP moves + 8 selects for the window shifts, the rest the same adds / mins
// B does.  B issues 101 instructions per step for 8 int16 lanes x 64 alignments.  Cells per wave-instruction: A 128 / 52 = 2.5,
// B 1024 / 101 = 10.1 -- the band is 8 wide, a wave is 64 wide, and a cross-lane move costs an issue slot like an add does.
// Packing two alignments per lane (lo / hi halves, v_pk_*) would double A's figure and still leave it 2 x behind.
//
// build + run (GPU box):   hipcc --offload-arch=gfx950 -O3 -o /tmp/dp_mapping_a tools/ubench/dp_mapping_a.hip -Lplatypus_amd -lplat_mi355x
//                          LD_LIBRARY_PATH=platypus_amd /tmp/dp_mapping_a [n_alignments] [read_len]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/platypus_mi355x.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); exit(2); } } while (0)

typedef short i16;
__device__ __forceinline__ i16 add16(i16 a, i16 b) { return (i16)(unsigned short)((unsigned)(unsigned short)a + (unsigned)(unsigned short)b); }
__device__ __forceinline__ i16 min16(i16 a, i16 b) { return a < b ? a : b; }
// lane k <- lane k - 1 inside the row (DPP row_shr:1); the group's lane 0 takes `fill`
__device__ __forceinline__ i16 up(i16 v, i16 fill, bool first) {
    const int x = __builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);
    return first ? fill : (i16)x;
}
// lane k <- lane k + 1 (DPP row_shl:1); the group's lane 7 takes `fill`
__device__ __forceinline__ i16 down(i16 v, i16 fill, bool last) {
    const int x = __builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xF, 0xF, false);
    return last ? fill : (i16)x;
}

// 8 lanes per alignment: lane k of a group is lane k of the reference's SSE registers
__global__ void __launch_bounds__(256)
k_dp_mapping_a(int n, int lmax, const uint8_t* __restrict__ haps, const uint8_t* __restrict__ reads, const uint8_t* __restrict__ quals,
               const uint8_t* __restrict__ gos, const int32_t* __restrict__ len2s, int gapextend, int nucprior, int32_t* __restrict__ out)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int a = t >> 3, k = t & 7;
    if (a >= n) return;
    const int len2 = len2s[a], len1 = len2 + 15;
    const uint8_t* seq1 = haps + (size_t)a * (lmax + 15);
    const uint8_t* go = gos + (size_t)a * (lmax + 15);
    const uint8_t* seq2 = reads + (size_t)a * lmax;
    const uint8_t* qual2 = quals + (size_t)a * lmax;
    const i16 INF = 0x7800, GE = (i16)(gapextend * 4), NP = (i16)(nucprior * 4);
    const bool first = k == 0, last = k == 7;
    i16 m1 = INF, i1 = INF, d1 = INF, m2 = INF, i2 = INF, d2 = INF;
    i16 s1w = seq1[k], s2w = INF, q2w = 64 * 4, gop = (i16)(4 * go[k]);
    i16 minscore = INF;
    for (int h = 0; h < len2 + 8; ++h) {
        // even half-step (align.c:218-335)
        const bool in = h < len2;
        s2w = up(s2w, in ? (i16)seq2[h] : (i16)'0', first);
        q2w = up(q2w, in ? (i16)(4 * qual2[h]) : (i16)(64 * 4), first);
        if (k == h) { m1 = (i16)-0x8000; m2 = (i16)-0x8000; }
        m1 = min16(m1, min16(i1, d1));
        if (k == h - len2) minscore = min16(minscore, m1);
        const i16 s1n = s1w == 'N' ? (i16)0 : INF;
        m1 = add16(m1, min16(s2w == s1w ? (i16)0 : q2w, s1n));
        const i16 gnext = down(gop, 0, last);
        const i16 tt = min16(add16(d2, GE), add16(min16(m2, i2), gnext));
        d1 = up(tt, INF, first);
        i1 = add16(min16(add16(i2, GE), add16(m2, gop)), NP);
        // odd half-step (:376-484)
        const int x = 8 + h;
        const i16 c = x < len1 ? (i16)seq1[x] : (i16)'N';
        const i16 g = (i16)(4 * go[x < len1 ? x : len1 - 1]);
        s1w = down(s1w, c, last);
        gop = down(gop, g, last);
        m2 = min16(m2, min16(i2, d2));
        if (k == h - len2) minscore = min16(minscore, m2);
        const i16 s1n2 = s1w == 'N' ? (i16)0 : INF;
        m2 = add16(m2, min16(s2w == s1w ? (i16)0 : q2w, s1n2));
        d2 = min16(add16(d1, GE), add16(min16(m1, i1), gop));
        const i16 i1n = down(i1, 0, false), m1n = down(m1, 0, false);
        i2 = last ? INF : add16(min16(add16(i1n, GE), add16(m1n, gop)), NP);
    }
    int ms = minscore;                                                  // min over the group's 8 lanes
    for (int d = 1; d < 8; d <<= 1) { const int o = __shfl_xor(ms, d, 8); ms = o < ms ? o : ms; }
    if (first) out[a] = (ms + 0x8000) >> 2;
}

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 400000, L = argc > 2 ? atoi(argv[2]) : 150;
    std::vector<uint8_t> hap((size_t)n * (L + 15)), rd((size_t)n * L), ql((size_t)n * L), go((size_t)n * (L + 15));
    std::vector<int32_t> len(n, L);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 11); };
    for (int a = 0; a < n; ++a) {
        uint8_t* h = &hap[(size_t)a * (L + 15)];
        for (int i = 0; i < L + 15; ++i) { h[i] = "ACGT"[rnd() & 3]; go[(size_t)a * (L + 15) + i] = (uint8_t)(1 + rnd() % 45); }
        if (rnd() % 50 == 0) h[rnd() % (L + 15)] = 'N';
        const int off = rnd() % 16;                                     // the read starts on any diagonal of the band
        for (int i = 0; i < L; ++i) {
            uint8_t b = h[(off + i) < L + 15 ? off + i : L + 14];
            if (rnd() % 100 == 0) b = "ACGT"[rnd() & 3];
            rd[(size_t)a * L + i] = b; ql[(size_t)a * L + i] = (uint8_t)(2 + rnd() % 40);
        }
        if (rnd() % 10 == 0) {                                           // a deletion in the read: the path has to leave its diagonal
            const int at = 20 + rnd() % (L - 40), w = 1 + rnd() % 3;
            memmove(&rd[(size_t)a * L + at], &rd[(size_t)a * L + at + w], (size_t)(L - at - w));
        }
    }
    uint8_t *dh, *dr, *dq, *dg; int32_t *dl, *oa, *ob;
    CK(hipMalloc(&dh, hap.size() + 64)); CK(hipMalloc(&dr, rd.size() + 64)); CK(hipMalloc(&dq, ql.size() + 64)); CK(hipMalloc(&dg, go.size() + 64));
    CK(hipMalloc(&dl, n * 4)); CK(hipMalloc(&oa, n * 4)); CK(hipMalloc(&ob, n * 4));
    CK(hipMemcpy(dh, hap.data(), hap.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(dr, rd.data(), rd.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dq, ql.data(), ql.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(dg, go.data(), go.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dl, len.data(), n * 4, hipMemcpyHostToDevice));
    plat_ctx* ctx = nullptr;
    if (plat_ctx_create(0, &ctx) != PLAT_OK) { fprintf(stderr, "plat_ctx_create failed\n"); return 2; }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 5;
    float msA = 0, msB = 0;
    const unsigned blocks = (unsigned)(((size_t)n * 8 + 255) / 256);
    for (int r = 0; r <= reps; ++r) {                                    // (first round untimed)
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_dp_mapping_a, dim3(blocks), dim3(256), 0, 0, n, L, dh, dr, dq, dg, dl, 3, 2, oa);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r) msA += ms;
        CK(hipEventRecord(e0, 0));
        if (plat_dp_batch(ctx, n, L, dh, dr, dq, dg, dl, 3, 2, ob, nullptr) != PLAT_OK) { fprintf(stderr, "plat_dp_batch failed\n"); return 2; }
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); if (r) msB += ms;
    }
    std::vector<int32_t> ha(n), hb(n);
    CK(hipMemcpy(ha.data(), oa, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), ob, n * 4, hipMemcpyDeviceToHost));
    long long diff = 0, nonzero = 0;
    for (int a = 0; a < n; ++a) { diff += ha[a] != hb[a]; nonzero += hb[a] != 0; }
    const double cells = 16.0 * L * n;
    printf("{\"tool\": \"tools/ubench/dp_mapping_a.hip\", \"alignments\": %d, \"read_len\": %d, \"scores_that_differ\": %lld, \"scores_nonzero\": %lld, "
           "\"mapping_a\": {\"what\": \"8 lanes per alignment (one band diagonal per lane), DPP row shifts\", \"ms\": %.4f, \"gcups\": %.1f}, "
           "\"mapping_b_row_entry_point\": {\"what\": \"plat_dp_batch = k_dp_rows (one lane per padded row, strided byte loads: the plumbing path, used here as the CHECKER; the tuned mapping B is k_dp_jobs: gcups_all_dp of bench.py)\", \"ms\": %.4f, \"gcups\": %.1f}, "
           "\"compare_mapping_a_with\": \"k_dp_jobs all-DP rate of the bench line (gcups_all_dp)\"}\n", n, L, diff, nonzero, msA / reps, cells / (msA / reps * 1e-3) / 1e9, msB / reps,
           cells / (msB / reps * 1e-3) / 1e9);
    return diff ? 1 : 0;
}
