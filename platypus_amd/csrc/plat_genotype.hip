// plat_genotype.hip -- DiploidGenotype.calculateDataLikelihood + the likelihood part of Population.setup
// (cgenotype.pyx:131-189, cpopulation.pyx:283-309) on the device.
//
// One quarter-wave (16 lanes) per (window, individual): most windows have 3..10 genotypes, a full wave per unit left
// three quarters of the fp64 lanes idle.  Lane g owns genotype g (looping when G > 16) and adds the terms of the
// individual's reads IN INDEX ORDER in fp64 (no FMA contraction: built with -ffp-contract=off), so the sums are the
// reference's sums.  The order of the ADDITIONS is fixed, the order in which the terms are COMPUTED is not: the rarely
// taken term log(0.5*(exp(l1)+exp(l2))) (a few per cent of the (genotype, read) pairs, but with 64 lanes some lane takes
// it at almost every read, and a wave pays for a branch any of its lanes takes) is queued in LDS per block of 8 reads
// and evaluated by all lanes of the wave together, before the block's additions.  Only that term and the final exp()
// rescale go through the device libm instead of glibc (difference <= a few ulp).
#include "plat_internal.hpp"

namespace plat {

constexpr int GENO_GROUP = 16;       // lanes per (window, individual) unit

constexpr int GENO_BLOCK = 8;        // reads per block of additions

// Exclusive prefix sum over the wave of a per-lane count <= 15, and the wave's total (ballot per bit + mbcnt).
__device__ __forceinline__ int geno_wave_prefix(int cnt, int& total) {
    int pre = 0;
    total = 0;
#pragma unroll
    for (int bit = 0; bit < 4; ++bit) {
        const unsigned long long m = __ballot((cnt >> bit) & 1);
        pre += (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u)) << bit;
        total += __popcll(m) << bit;
    }
    return pre;
}

__global__ void __launch_bounds__(64)
k_genotype(plat_window_batch b, int n_ind, long long n_units, const int32_t* __restrict__ seg_read_begin,
           const int32_t* __restrict__ seg_n_good, const double* __restrict__ loglik,
           const int64_t* __restrict__ gl_off, double* __restrict__ out_gl, double* __restrict__ out_logl,
           double* __restrict__ out_gof)
{
    __shared__ double s_q1[64 * GENO_BLOCK], s_q2[64 * GENO_BLOCK];    // queued (l1, l2); s_q1 takes the results
    const long long unit = (long long)blockIdx.x * (64 / GENO_GROUP) + threadIdx.x / GENO_GROUP;
    const bool live = unit < n_units;                                    // (the wave's queue needs every lane to stay)
    const int w = live ? (int)(unit / n_ind) : 0, ind = live ? (int)(unit % n_ind) : 0;
    const int lane = threadIdx.x % GENO_GROUP;
    const int H = live ? b.win_hap_begin[w + 1] - b.win_hap_begin[w] : 0;
    const int rb = b.win_read_begin[w], R = b.win_read_begin[w + 1] - rb;
    const int G = H * (H + 1) / 2;
    const long long seg = (long long)w * n_ind + ind;
    const int s0 = seg_read_begin[seg] - rb, s1 = live ? seg_read_begin[seg + 1] - rb : s0;
    const int nGood = seg_n_good[seg];
    const double* ll = loglik + b.pair_off[w];
    const long long gbase = gl_off[w];
    const double log10E = 0.43429448190325182;      // cgenotype.pyx:24
    const double logHalf = -0.69314718055994529;    // cgenotype.pyx:28

    double mymax = -1e7;                            // cpopulation.pyx:288
    int Gmax = G;                                   // rounds of 16 genotypes: as many as the unit of the wave with most
    for (int s = 32; s >= GENO_GROUP; s >>= 1) Gmax = max(Gmax, __shfl_xor(Gmax, s));
    for (int g0 = 0; g0 < Gmax; g0 += GENO_GROUP) {
        const int g = g0 + lane;
        const bool mine = g < G;
        // genotype g -> (a, b), a <= b, in the order of cgenotype.pyx:212-216
        int a = 0, rem = mine ? g : 0;
        while (rem >= H - a && a < H) { rem -= H - a; ++a; }
        const int bb = a + rem;
        const bool summing = mine && nGood != 0;    // cpopulation.pyx:293
        const double* arr1 = ll + (long long)a * R;
        const double* arr2 = ll + (long long)bb * R;
        double like = 0.0, gsum = 0.0;
        // reads in index order (cgenotype.pyx:151-180), a block of 8 at a time: loads issued together, the slow terms of the
        // whole wave evaluated together, then the additions in order
        for (int r0 = s0; __any(summing && r0 < s1); r0 += GENO_BLOCK) {
            double v1[GENO_BLOCK], v2[GENO_BLOCK];
            unsigned slow = 0;
#pragma unroll
            for (int k = 0; k < GENO_BLOCK; ++k) {
                const bool in = summing && r0 + k < s1;
                v1[k] = in ? arr1[r0 + k] : 0.0; v2[k] = in ? arr2[r0 + k] : 0.0;
                const double d = fabs(v1[k] - v2[k]);
                if (in && a != bb && !(d >= 3) && !(d <= 1e-3)) slow |= 1u << k;   // exactly the adds' last branch below (NaN included)
            }
            int total;
            const int pre = geno_wave_prefix(__popc(slow), total);
            if (total > 0) {                        // wave-uniform
                int j = pre;
#pragma unroll
                for (int k = 0; k < GENO_BLOCK; ++k)
                    if (slow >> k & 1) { s_q1[j] = v1[k]; s_q2[j] = v2[k]; ++j; }
                __syncthreads();
                for (int i = threadIdx.x; i < total; i += 64) s_q1[i] = log(0.5 * (exp(s_q1[i]) + exp(s_q2[i])));
                __syncthreads();
            }
            int j = pre;
#pragma unroll
            for (int k = 0; k < GENO_BLOCK; ++k) {
                if (!summing || r0 + k >= s1) continue;
                const double l1 = v1[k], l2 = v2[k];
                const double ll1 = log10E * l1, ll2 = log10E * l2;
                gsum += (ll1 > ll2 ? ll1 : ll2);
                if (a == bb) like += l1;
                else if (slow >> k & 1) like += s_q1[j++];                   // the queued term: the class is decided once, above
                else if (fabs(l1 - l2) >= 3) like += (logHalf + (l1 > l2 ? l1 : l2));
                else like += l1;
            }
            if (total > 0) __syncthreads();         // the queue is rewritten by the next block
        }
        if (mine) {
            double L = 1.0, gof = 0.0;
            if (nGood != 0) {
                L = like;
                gof = (-10 * gsum) / nGood;         // cgenotype.pyx:182-183
                if (L > mymax) mymax = L;
            }
            out_logl[gbase + (long long)ind * G + g] = L;
            out_gof[gbase + (long long)g * n_ind + ind] = gof;
        }
    }
    for (int s = GENO_GROUP / 2; s > 0; s >>= 1) {  // stays inside the unit's 16 lanes
        double o = __shfl_xor(mymax, s);
        mymax = o > mymax ? o : mymax;
    }
    for (int g = lane; g < G; g += GENO_GROUP) {    // cpopulation.pyx:304-309
        const long long o = gbase + (long long)ind * G + g;
        double v = 1.0;
        if (nGood != 0) {
            v = exp(out_logl[o] - mymax);
            v = v > 1e-300 ? v : 1e-300;
        }
        out_gl[o] = v;
    }
}

}  // namespace plat

PLAT_EXPORT int plat_genotype_window_batch(plat_ctx* ctx, const plat_window_batch* batch, int n_ind,
                                           const int32_t* seg_read_begin, const int32_t* seg_n_good,
                                           const double* loglik, const int64_t* gl_off, double* out_gl,
                                           double* out_logl, double* out_gof, void* stream)
{
    if (!ctx || !batch || n_ind <= 0) return PLAT_ERR_INVALID;
    if (batch->n_windows == 0) return PLAT_OK;
    if (!seg_read_begin || !seg_n_good || !loglik || !gl_off || !out_gl || !out_logl || !out_gof)
        return PLAT_ERR_INVALID;
    PLAT_HIP(ctx, hipSetDevice(ctx->device));
    const long long nunits = (long long)batch->n_windows * n_ind;
    const long long nblk = (nunits + 64 / plat::GENO_GROUP - 1) / (64 / plat::GENO_GROUP);
    if (nblk > 0x7FFFFFFFll) return PLAT_ERR_INVALID;
    ctx->ev_valid_geno = 0;
    PLAT_EV(ctx, 6, (hipStream_t)stream);
    { PLAT_KT_BEGIN(ctx, PLAT_KT_GENOTYPE, (hipStream_t)stream); hipLaunchKernelGGL(plat::k_genotype, dim3((unsigned)nblk), dim3(64), 0, (hipStream_t)stream, *batch, n_ind, nunits,
                       seg_read_begin, seg_n_good, loglik, gl_off, out_gl, out_logl, out_gof); PLAT_KT_END(ctx, PLAT_KT_GENOTYPE, (hipStream_t)stream); }
    PLAT_HIP(ctx, hipGetLastError());
    PLAT_EV(ctx, 7, (hipStream_t)stream);
    ctx->ev_valid_geno = ctx->profile;
    return PLAT_OK;
}
