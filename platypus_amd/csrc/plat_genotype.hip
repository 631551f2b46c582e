// plat_genotype.hip -- DiploidGenotype.calculateDataLikelihood + the likelihood part of Population.setup
// (cgenotype.pyx:131-189, cpopulation.pyx:283-309) on the device.
//
// One quarter-wave (16 lanes) per (window, individual): most windows have 3..10 genotypes, a full wave per unit left
// three quarters of the fp64 lanes idle.  Lane g owns genotype g (looping when G > 16) and walks the
// individual's reads IN INDEX ORDER in fp64 (no FMA contraction: built with -ffp-contract=off), so the
// sums are the reference's sums.  Only the rarely taken branch log(0.5*(exp(l1)+exp(l2))) and the
// final exp() rescale go through the device libm instead of glibc (difference <= a few ulp).
#include "plat_internal.hpp"

namespace plat {

constexpr int GENO_GROUP = 16;       // lanes per (window, individual) unit

__global__ void __launch_bounds__(64)
k_genotype(plat_window_batch b, int n_ind, long long n_units, const int32_t* __restrict__ seg_read_begin,
           const int32_t* __restrict__ seg_n_good, const double* __restrict__ loglik,
           const int64_t* __restrict__ gl_off, double* __restrict__ out_gl, double* __restrict__ out_logl,
           double* __restrict__ out_gof)
{
    const long long unit = (long long)blockIdx.x * (64 / GENO_GROUP) + threadIdx.x / GENO_GROUP;
    if (unit >= n_units) return;
    const int w = (int)(unit / n_ind), ind = (int)(unit % n_ind);
    const int lane = threadIdx.x % GENO_GROUP;
    const int H = b.win_hap_begin[w + 1] - b.win_hap_begin[w];
    const int rb = b.win_read_begin[w], R = b.win_read_begin[w + 1] - rb;
    const int G = H * (H + 1) / 2;
    if (G == 0) return;
    const long long seg = (long long)w * n_ind + ind;
    const int s0 = seg_read_begin[seg] - rb, s1 = seg_read_begin[seg + 1] - rb;
    const int nGood = seg_n_good[seg];
    const double* ll = loglik + b.pair_off[w];
    const long long gbase = gl_off[w];
    const double log10E = 0.43429448190325182;      // cgenotype.pyx:24
    const double logHalf = -0.69314718055994529;    // cgenotype.pyx:28

    double mymax = -1e7;                            // cpopulation.pyx:288
    for (int g0 = 0; g0 < G; g0 += GENO_GROUP) {
        const int g = g0 + lane;
        if (g < G) {
            // genotype g -> (a, b), a <= b, in the order of cgenotype.pyx:212-216
            int a = 0, rem = g;
            while (rem >= H - a) { rem -= H - a; ++a; }
            const int bb = a + rem;
            double L = 1.0, gof = 0.0;
            if (nGood != 0) {                       // cpopulation.pyx:293
                const double* arr1 = ll + (long long)a * R;
                const double* arr2 = ll + (long long)bb * R;
                double like = 0.0, gsum = 0.0;
                // reads in index order (cgenotype.pyx:151-180); the loads of 8 reads are issued together so that the
                // serial fp64 chain does not wait for a memory round trip per read
                for (int r0 = s0; r0 < s1; r0 += 8) {
                    double v1[8], v2[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int r = min(r0 + k, s1 - 1);
                        v1[k] = arr1[r]; v2[k] = arr2[r];
                    }
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        if (r0 + k >= s1) break;
                        const double l1 = v1[k], l2 = v2[k];
                        const double ll1 = log10E * l1, ll2 = log10E * l2;
                        gsum += (ll1 > ll2 ? ll1 : ll2);
                        if (a == bb) like += l1;
                        else if (fabs(l1 - l2) >= 3) like += (logHalf + (l1 > l2 ? l1 : l2));
                        else if (fabs(l1 - l2) <= 1e-3) like += l1;
                        else like += log(0.5 * (exp(l1) + exp(l2)));
                    }
                }
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
    hipLaunchKernelGGL(plat::k_genotype, dim3((unsigned)nblk), dim3(64), 0, (hipStream_t)stream, *batch, n_ind, nunits,
                       seg_read_begin, seg_n_good, loglik, gl_off, out_gl, out_logl, out_gof);
    PLAT_HIP(ctx, hipGetLastError());
    PLAT_EV(ctx, 7, (hipStream_t)stream);
    ctx->ev_valid_geno = ctx->profile;
    return PLAT_OK;
}
