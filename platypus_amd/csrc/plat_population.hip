// plat_population.hip -- what follows the genotype likelihoods in a calling window:
//   Population.call            cpopulation.pyx:678-703   EM for the haplotype frequencies (EMiteration :384-457)
//   Population.callGenotypes   cpopulation.pyx:623-676   arg-max genotype per individual
//   Population.calculatePosterior            :459-594   phred-scaled posterior of one variant
//   computeGenotypeCallAndLikelihoods  vcfutils.pyx:163-334   per-site genotype marginalisation of one sample
//
// Everything is fp64 and every sum runs in the reference's order (built with -ffp-contract=off), so frequencies, EM
// likelihoods, calls and marginal likelihoods are the reference's doubles bit for bit; only calculatePosterior goes
// through log/exp/log10 of the device libm before its final round().
//
// Genotype g <-> haplotype pair (a, b), a <= b, in the order of generateAllGenotypesFromHaplotypeList
// (cgenotype.pyx:193-218): g(a, b) = a*H - a*(a-1)/2 + (b - a).
#include "plat_internal.hpp"

namespace plat {

__device__ __forceinline__ int geno_index(int a, int b, int H) { return a * H - a * (a - 1) / 2 + (b - a); }

// k_em_wide: the same arithmetic with the window's genotype likelihoods and responsibilities in LDS and the E-step spread over
// (individual, genotype) pairs.  k_em's E-step is one lane per individual walking its G genotypes through global memory: with one
// wave per window that is a chain of memory round trips (187 us for 200 windows x 100 samples, 76 us for 10 000 windows x 1).
// Here the likelihoods are copied in once (coalesced); every product L * f_s * f_r * (1 + (r != s)) is one lane's (:421); the sums
// that have an order -- csrSum over a sample's genotypes (:422), the M-step's per-haplotype sum over samples and genotypes
// (:437-446) -- are still walked in that order by one lane each, from LDS.  Same expressions, same order: same doubles.
__global__ void __launch_bounds__(256)
k_em_wide(int n_ind, const int32_t* __restrict__ win_hap_begin, const int64_t* __restrict__ gl_off,
          const int32_t* __restrict__ n_reads, const double* __restrict__ gl, int max_iters, int use_em,
          double* __restrict__ out_freq, double* __restrict__ out_em, int32_t* __restrict__ out_call,
          int32_t* __restrict__ out_iters, int max_haps, int maxG, int use_streams, long long* sticky)
{
    extern __shared__ double s_freq[];                 // [max_haps] | Ls [n_ind][G] | rsp [n_ind][G] | csum [n_ind] | streams | nr, gs, gr
    __shared__ unsigned long long s_change;
    __shared__ int s_with;
    const int w = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
    const int h0 = win_hap_begin[w], H = win_hap_begin[w + 1] - h0;
    const int G = H * (H + 1) / 2;
    if (H <= 0 || H > max_haps) {
        // more haplotypes than the caller sized the LDS for: the window is refused -- calls and iterations -1, and the error is left
        // in the context's sticky word, so that the next plat_stream_sync fails (a caller that never reads out_iters sees it too)
        if (tid == 0 && out_iters) out_iters[w] = H <= 0 ? 0 : -1;
        if (H > max_haps) {
            for (int i = tid; i < n_ind; i += nthr) out_call[(long long)w * n_ind + i] = -1;
            if (tid == 0 && *sticky == 0) *sticky = PLAT_ERR_INVALID;
        }
        return;
    }
    const double* L = gl + gl_off[w];
    double* em = out_em + gl_off[w];
    double* Ls = s_freq + max_haps;
    double* rsp = Ls + (size_t)n_ind * maxG;
    double* csum = rsp + (size_t)n_ind * maxG;
    // M-step streams (use_streams): T[k][i][0..H] = the terms haplotype k's frequency sum takes from sample i, in the order it
    // takes them ((a, k) for a < k, (k, k) twice, (k, b) for b > k): the lane that owns k then only reads and adds
    double* T = csum + n_ind;
    const int SL = H + 1;                              // terms per (haplotype, sample)
    const int TS = (n_ind * SL + 7) & ~7;               // stream length per haplotype, padded with zeros to the M-step's batches of 8
    int* nr = (int*)(T + (use_streams ? (size_t)max_haps * ((n_ind * (max_haps + 1) + 7) & ~7) : 0));
    short* gs = (short*)(nr + n_ind);
    short* gr = gs + maxG;
    const int nP = n_ind * G;
    double eps = 1.0 / (n_ind * 2 * 2);                // :684
    if (1e-3 < eps) eps = 1e-3;
    const double uniformFreq = 1.0 / H;
    if (tid == 0) { s_with = 0; s_change = 0ull; }
    for (int k = tid; k < H; k += nthr) s_freq[k] = uniformFreq;
    for (int j = tid; j < G; j += nthr) {              // genotype j = (s, r), s <= r, cgenotype.pyx:212-216
        int a = 0, rem = j;
        while (rem >= H - a) { rem -= H - a; ++a; }
        gs[j] = (short)a; gr[j] = (short)(a + rem);
    }
    for (int i = tid; i < n_ind; i += nthr) nr[i] = n_reads[(long long)w * n_ind + i];
    {   // (four loads in flight per thread)
        int p = tid;
        for (; p + 3 * nthr < nP; p += 4 * nthr) {
            const double a0 = L[p], a1 = L[p + nthr], a2 = L[p + 2 * nthr], a3 = L[p + 3 * nthr];
            Ls[p] = a0; Ls[p + nthr] = a1; Ls[p + 2 * nthr] = a2; Ls[p + 3 * nthr] = a3;
        }
        for (; p < nP; p += nthr) Ls[p] = L[p];
    }
    __syncthreads();
    {
        int mine = 0;
        for (int i = tid; i < n_ind; i += nthr) mine += nr[i] != 0;
        if (mine) atomicAdd(&s_with, mine);
        for (int p = tid; p < nP; p += nthr)           // the reference leaves stale values for samples without reads and never reads them
            if (nr[p / G] == 0) em[p] = 0.0;
        if (use_streams)                               // samples without reads contribute nothing: their terms stay 0.0 (x + 0.0 == x)
            for (int t = tid; t < H * TS; t += nthr) T[t] = 0.0;
    }
    __syncthreads();
    const int nWithData = s_with;
    double maxChange = eps + 1;
    int iters = 0;
    while (maxChange > eps && iters < max_iters) {     // :700-702
        for (int p = tid; p < nP; p += nthr) {         // E-step: the products
            const int i = p / G, j = p - i * G;
            if (nr[i] == 0) continue;
            const int s2 = gs[j], r2 = gr[j];
            rsp[p] = Ls[p] * s_freq[s2] * s_freq[r2] * (1 + (r2 != s2));   // :421
        }
        __syncthreads();
        for (int i = tid; i < n_ind; i += nthr) {      // csrSum in genotype order
            if (nr[i] == 0) continue;
            double cs = 0.0;
            const double* row = rsp + i * G;
            int j = 0;
            for (; j + 4 <= G; j += 4) {               // (four reads in flight, additions in order)
                const double a0 = row[j], a1 = row[j + 1], a2 = row[j + 2], a3 = row[j + 3];
                cs += a0; cs += a1; cs += a2; cs += a3;
            }
            for (; j < G; ++j) cs += row[j];
            csum[i] = cs;
        }
        __syncthreads();
        for (int p = tid; p < nP; p += nthr) {         // normalise; the responsibilities are the EMLikelihoods output too
            const int i = p / G;
            if (nr[i] == 0) continue;
            double v = rsp[p];
            if (csum[i] > 0.0) v /= csum[i];
            rsp[p] = v;
            em[p] = v;
            if (use_streams) {
                const int j = p - i * G, s2 = gs[j], r2 = gr[j];
                if (s2 == r2) { T[s2 * TS + i * SL + s2] = v; T[s2 * TS + i * SL + s2 + 1] = v; }
                else {
                    T[s2 * TS + i * SL + r2 + 1] = v;               // stream s2, partner r2 > s2: behind the doubled diagonal
                    T[r2 * TS + i * SL + s2] = v;                   // stream r2, partner s2 < r2
                }
            }
        }
        if (tid == 0) s_change = 0ull;
        __syncthreads();
        // M-step: one lane per haplotype k adds the responsibilities of the genotypes that contain k in the order the reference's
        // double loop reaches them: (a, k) for a < k, (k, k) added as first and as second, (k, b) for b > k
        for (int k = tid; k < H; k += nthr) {
            double acc = 0.0;
            if (use_streams) {
                const double* tk = T + (size_t)k * TS;
                for (int t0 = 0; t0 < TS; t0 += 8) {   // eight LDS reads in flight; the padding adds 0.0
                    double v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = tk[t0 + u];
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc += v[u];
                }
            } else
            for (int i = 0; i < n_ind; ++i) {
                if (nr[i] == 0) continue;
                const double* csr = rsp + i * G;
                for (int a0 = 0; a0 < H; a0 += 8) {
                    double v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int a = min(a0 + u, H - 1);
                        v[u] = csr[a < k ? geno_index(a, k, H) : geno_index(k, a, H)];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int a = a0 + u;
                        if (a < H) { acc += v[u]; if (a == k) acc += v[u]; }
                    }
                }
            }
            const double nf = acc / (2 * nWithData);   // :449
            const double fc = fabs(s_freq[k] - nf);
            if (fc > 0.0) atomicMax(&s_change, (unsigned long long)__double_as_longlong(fc));   // (fc > 0: the bit patterns order like the values; a NaN never raises the maximum, as in `if fc > change`)
            s_freq[k] = nf;                            // each thread owns its k: nobody else reads freq in this phase
        }
        __syncthreads();
        maxChange = __longlong_as_double((long long)s_change);
        ++iters;
        __syncthreads();
    }
    for (int k = tid; k < H; k += nthr) out_freq[h0 + k] = s_freq[k];
    if (tid == 0 && out_iters) out_iters[w] = iters;
    // callGenotypes, :623-676
    for (int i = tid; i < n_ind; i += nthr) {
        int best = -1;
        if (nr[i] != 0) {
            const double* row = (use_em == 1 ? rsp : Ls) + i * G;
            double maxL = 0.0;
            for (int g = 0; g < G; ++g) {
                const double v = row[g];
                if (best == -1 || v > maxL) { maxL = v; best = g; }
            }
        }
        out_call[(long long)w * n_ind + i] = best;
    }
}

// One single-wave workgroup per window.  E-step: one lane per individual (the genotype loop is sequential, :411-429);
// M-step: one lane per haplotype k, which adds the responsibilities of the genotypes that contain k in exactly the
// order the reference's double loop reaches them (:437-446).
__global__ void __launch_bounds__(64)
k_em(int n_ind, const int32_t* __restrict__ win_hap_begin, const int64_t* __restrict__ gl_off,
     const int32_t* __restrict__ n_reads, const double* __restrict__ gl, int max_iters, int use_em,
     double* __restrict__ out_freq, double* __restrict__ out_em, int32_t* __restrict__ out_call,
     int32_t* __restrict__ out_iters, int max_haps, int csr_in_lds, long long* sticky)
{
    extern __shared__ double s_freq[];                 // [max_haps], then (csr_in_lds) the responsibilities [n_ind][G] of this window
    const int w = blockIdx.x, lane = threadIdx.x;
    const int h0 = win_hap_begin[w], H = win_hap_begin[w + 1] - h0;
    const int G = H * (H + 1) / 2;
    if (H <= 0) { if (lane == 0 && out_iters) out_iters[w] = 0; return; }
    if (H > max_haps) {                                // refused as k_em_wide refuses it (the frequencies would not fit the LDS carve)
        if (lane == 0 && out_iters) out_iters[w] = -1;
        for (int i = lane; i < n_ind; i += 64) out_call[(long long)w * n_ind + i] = -1;
        if (lane == 0 && *sticky == 0) *sticky = PLAT_ERR_INVALID;
        return;
    }
    const double* L = gl + gl_off[w];
    double* em = out_em + gl_off[w];
    // The M-step is a serial chain per haplotype over all individuals (the reference's order of additions); reading the
    // responsibilities back from global memory made it a chain of dependent L2 round trips (350 us for 100 samples x 8
    // haplotypes).  When they fit they are kept in LDS as well (the global copy is the EMLikelihoods output).
    double* rsp = csr_in_lds ? s_freq + max_haps : em;
    const int32_t* nr = n_reads + (long long)w * n_ind;

    double eps = 1.0 / (n_ind * 2 * 2);                // :684
    if (1e-3 < eps) eps = 1e-3;
    double maxChange = eps + 1;
    const double uniformFreq = 1.0 / H;
    for (int k = lane; k < H; k += 64) s_freq[k] = uniformFreq;
    int nWithData = 0;
    for (int i = lane; i < n_ind; i += 64) {
        nWithData += nr[i] != 0;
        if (nr[i] == 0)                                // the reference leaves stale values here and never reads them
            for (int j = 0; j < G; ++j) em[(long long)i * G + j] = 0.0;
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) nWithData += __shfl_xor(nWithData, s);
    __syncthreads();
    int iters = 0;
    while (maxChange > eps && iters < max_iters) {     // :700-702
        // E-step
        for (int i = lane; i < n_ind; i += 64) {
            if (nr[i] == 0) continue;
            const double* Li = L + (long long)i * G;
            double* csr = rsp + (long long)i * G;
            double csrSum = 0.0;
            int j = 0;
            for (int s = 0; s < H; ++s)
                for (int r = s; r < H; ++r, ++j) {
                    const double thisCSR = Li[j] * s_freq[s] * s_freq[r] * (1 + (r != s));   // :421
                    csr[j] = thisCSR;
                    csrSum += thisCSR;
                }
            if (csrSum > 0.0)
                for (j = 0; j < G; ++j) csr[j] /= csrSum;
            if (csr_in_lds)
                for (j = 0; j < G; ++j) em[(long long)i * G + j] = csr[j];
        }
        __syncthreads();                               // (one wave: orders the global em writes before the reads below)
        __threadfence_block();
        // M-step
        double change = 0.0;
        for (int k = lane; k < H; k += 64) {
            double acc = 0.0;
            for (int i = 0; i < n_ind; ++i) {
                if (nr[i] == 0) continue;
                const double* csr = rsp + (long long)i * G;
                // genotypes that contain k, in the order the reference's double loop reaches them: (a, k) for a < k (k is the
                // second haplotype), (k, k) added as first and as second, (k, b) for b > k.  Eight loads are issued
                // together so that the chain of additions does not wait for a memory round trip per term.
                for (int a0 = 0; a0 < H; a0 += 8) {
                    double v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int a = min(a0 + u, H - 1);
                        v[u] = csr[a < k ? geno_index(a, k, H) : geno_index(k, a, H)];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int a = a0 + u;
                        if (a < H) { acc += v[u]; if (a == k) acc += v[u]; }
                    }
                }
            }
            const double nf = acc / (2 * nWithData);   // :449
            const double fc = fabs(s_freq[k] - nf);
            if (fc > change) change = fc;
            s_freq[k] = nf;                            // each lane owns its k: no other lane reads freq in this phase
        }
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) {
            const double o = __shfl_xor(change, s);
            if (o > change) change = o;
        }
        maxChange = change;
        ++iters;
        __syncthreads();
    }
    for (int k = lane; k < H; k += 64) out_freq[h0 + k] = s_freq[k];
    if (lane == 0 && out_iters) out_iters[w] = iters;
    // callGenotypes, :623-676
    for (int i = lane; i < n_ind; i += 64) {
        int best = -1;
        if (nr[i] != 0) {
            const double* row = (use_em == 1 ? em : L) + (long long)i * G;
            double maxL = 0.0;
            for (int g = 0; g < G; ++g) {
                const double v = row[g];
                if (best == -1 || v > maxL) { maxL = v; best = g; }
            }
        }
        out_call[(long long)w * n_ind + i] = best;
    }
}

// One single-wave workgroup per variant: lanes over individuals, then lane 0 adds the per-individual logs in
// individual order (:576-584).
__global__ void __launch_bounds__(64)
k_variant_posterior(int n_ind, const int32_t* __restrict__ win_hap_begin, const int64_t* __restrict__ gl_off,
                    const int32_t* __restrict__ n_reads, const double* __restrict__ gl, const double* __restrict__ freq,
                    const int32_t* __restrict__ var_window, const int64_t* __restrict__ var_mask_off,
                    const uint8_t* __restrict__ hap_has_var, const double* __restrict__ prior,
                    double* __restrict__ scratch, double* __restrict__ out_post)
{
    extern __shared__ double s_f[];                    // freq[H] | freqsPrime[H]
    const int v = blockIdx.x, lane = threadIdx.x;
    const int w = var_window[v];
    const int h0 = win_hap_begin[w], H = win_hap_begin[w + 1] - h0;
    const int G = H * (H + 1) / 2;
    double* f = s_f;
    double* fp = s_f + H;
    const uint8_t* has = hap_has_var + var_mask_off[v];
    const double* L = gl + gl_off[w];
    const int32_t* nr = n_reads + (long long)w * n_ind;
    double* logs = scratch + (long long)v * 2 * n_ind;
    for (int k = lane; k < H; k += 64) f[k] = freq[h0 + k];
    __syncthreads();
    if (lane == 0) {                                   // :509-534
        double sumFreqs = 0.0;
        for (int i = 0; i < H; ++i) {
            if (!has[i]) { fp[i] = f[i]; sumFreqs += f[i]; }
            else fp[i] = 0.0;
        }
        if (sumFreqs > 0)
            for (int i = 0; i < H; ++i) fp[i] /= sumFreqs;
    }
    __syncthreads();
    for (int i = lane; i < n_ind; i += 64) {
        if (nr[i] == 0) continue;
        const double* Li = L + (long long)i * G;
        double sumVar = 0.0, sumNoVar = 0.0;
        int g = 0;
        for (int r = 0; r < H; ++r)
            for (int s = r; s < H; ++s, ++g) {
                const double factor = r != s ? 2.0 : 1.0;
                sumVar += (factor * f[r] * f[s] * Li[g]);            // :564
                sumNoVar += (factor * fp[r] * fp[s] * Li[g]);        // :569
            }
        logs[2 * i] = sumVar > 0 ? log(sumVar) : -708.0;             // :576-584
        logs[2 * i + 1] = sumNoVar > 0 ? log(sumNoVar) : -708.0;
    }
    __syncthreads();
    __threadfence_block();
    if (lane == 0) {
        double sv = 0.0, sn = 0.0;
        for (int i = 0; i < n_ind; ++i) {
            if (nr[i] == 0) continue;
            sv += logs[2 * i];
            sn += logs[2 * i + 1];
        }
        double ratio = exp(sn - sv);                                 // :586
        if (!(ratio > 1e-300)) ratio = 1e-300;
        const double p = prior[v];
        out_post[v] = round(-10.0 * (log10(ratio * (1.0 - p)) - log10(p + ratio * (1.0 - p))));   // :594
    }
}

// One lane per (site, individual).
__global__ void __launch_bounds__(64)
k_genotype_call(int n_sites, int n_ind, const int32_t* __restrict__ win_hap_begin,
                const int64_t* __restrict__ gl_off, const double* __restrict__ gl, const double* __restrict__ gof,
                const double* __restrict__ freq, const int32_t* __restrict__ site_window,
                const int32_t* __restrict__ site_nvar, const int64_t* __restrict__ site_vih_off,
                const int64_t* __restrict__ site_ref_off, const int32_t* __restrict__ var_in_hap,
                const int32_t* __restrict__ is_ref,
                const int64_t* __restrict__ lik_off, int32_t* __restrict__ out_phased, double* __restrict__ out_lik,
                double* __restrict__ out4)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)n_sites * n_ind) return;
    const int s = (int)(t / n_ind), ind = (int)(t % n_ind);
    const int w = site_window[s], nVar = site_nvar[s];
    const int h0 = win_hap_begin[w], H = win_hap_begin[w + 1] - h0;
    const int G = H * (H + 1) / 2;
    const double* L = gl + gl_off[w] + (long long)ind * G;
    const double* gf = gof + gl_off[w];                               // [g][ind]
    const double* f = freq + h0;
    const int32_t* vih = var_in_hap + site_vih_off[s];                // varThisPosInHap [H][nVar]
    const int32_t* ref = is_ref + site_ref_off[s];                    // haplotypeIsRefAtThisPos [H]
    const int NL = (nVar + 1) * (nVar + 2) / 2;
    double* lik = out_lik + lik_off[s] + (long long)ind * NL;

    double sumLikelihoods = 0.0, bestGof = 1e6, bestLikelihood = -1.0, nonRefPosterior = 0.0, refPosterior = 0.0;
    double phasedMaxLike = -1e6;
    int phasedIndex1 = -1, phasedIndex2 = -1, nl = 0;
    for (int index1 = 0; index1 <= nVar; ++index1)
        for (int index2 = 0; index2 <= index1; ++index2) {
            double marginal = 0.0;
            int g = 0;
            for (int a = 0; a < H; ++a)
                for (int bb = a; bb < H; ++bb, ++g) {
                    const int ref1 = ref[a], ref2 = ref[bb];
                    const double factor = a != bb ? 2.0 : 1.0;
                    int matching = 0, v1h1 = 0, v1h2 = 0, v2h1 = 0, v2h2 = 0;
                    if (index1 == 0 && index2 == 0) {
                        if (ref1 && ref2) matching = 1;
                    } else if (index2 == 0) {
                        v1h1 = vih[a * nVar + index1 - 1]; v1h2 = vih[bb * nVar + index1 - 1];
                        if ((ref2 && v1h1) || (ref1 && v1h2)) matching = 1;
                    } else {
                        v1h1 = vih[a * nVar + index1 - 1]; v1h2 = vih[bb * nVar + index1 - 1];
                        v2h1 = vih[a * nVar + index2 - 1]; v2h2 = vih[bb * nVar + index2 - 1];
                        if ((v1h1 && v2h2) || (v2h1 && v1h2)) matching = 1;
                    }
                    if (!matching) continue;
                    double cur;
                    if (n_ind > 25) cur = (factor * f[a] * f[bb] * L[g]);     // vcfutils.pyx:252-255
                    else cur = (factor * L[g]);
                    marginal += cur;
                    if (cur > phasedMaxLike) {                                                // :260-303
                        phasedMaxLike = cur;
                        if (index1 == 0 && index2 == 0) { phasedIndex1 = index1; phasedIndex2 = index2; }
                        else if (index2 == 0 && index1 != 0) {
                            if (v1h1) { phasedIndex1 = index1; phasedIndex2 = index2; }
                            else if (v1h2) { phasedIndex1 = index2; phasedIndex2 = index1; }
                        } else if (index2 == index1 && index1 > 0) { phasedIndex1 = index1; phasedIndex2 = index2; }
                        else if (index2 > 0 && index1 > 0 && index2 != index1) {
                            if (v1h1 && v2h2) { phasedIndex1 = index1; phasedIndex2 = index2; }
                            else if (v1h2 && v2h1) { phasedIndex1 = index2; phasedIndex2 = index1; }
                        }
                    }
                    const double gv = gf[(long long)g * n_ind + ind];
                    if (gv < bestGof) bestGof = gv;
                }
            if (marginal > bestLikelihood) bestLikelihood = marginal;
            if ((index1 == 1 && index2 == 0) || (index1 == 1 && index2 == 1)) nonRefPosterior += marginal;
            else if (index1 == 0 && index2 == 0) refPosterior += marginal;
            sumLikelihoods += marginal;
            lik[nl++] = marginal;
        }
    out_phased[2 * t] = phasedIndex1;
    out_phased[2 * t + 1] = phasedIndex2;
    out4[4 * t] = bestLikelihood / sumLikelihoods;
    out4[4 * t + 1] = nonRefPosterior / sumLikelihoods;
    out4[4 * t + 2] = refPosterior / sumLikelihoods;
    out4[4 * t + 3] = bestGof;
}


// ---- computeHaplotypeScore (vcfutils.pyx:1076-1114) ------------------------------------------------------------------------
// The reference reads DiploidGenotype.hap1Like / hap2Like, which calculateDataLikelihood (cgenotype.pyx:148-161) resets and
// refills on every call: after Population.setup they hold, for each haplotype, the sum over the reads of the LAST individual
// with reads of log10E * likelihood, in read order.  One quarter-wave per window: lane h sums haplotype h (fp64, read order,
// no FMA contraction), then lane 0 walks the negated sums in ascending order and sizes the first two clusters.
constexpr int HS_GROUP = 16;

__global__ void __launch_bounds__(64)
k_haplotype_score(plat_window_batch b, int n_ind, int max_haps, const int32_t* __restrict__ seg_read_begin,
                  const int32_t* __restrict__ seg_n_good, const double* __restrict__ loglik,
                  double* __restrict__ out_hap_like, int32_t* __restrict__ out_hap_score)
{
    extern __shared__ double s_hs[];
    const int grp = threadIdx.x / HS_GROUP, lane = threadIdx.x % HS_GROUP;
    const int w = blockIdx.x * (64 / HS_GROUP) + grp;
    const bool live = w < b.n_windows;
    double* mine = s_hs + (size_t)grp * max_haps;
    int H = 0;
    if (live) {
        const int hb = b.win_hap_begin[w];
        H = b.win_hap_begin[w + 1] - hb;
        const int rb = b.win_read_begin[w], R = b.win_read_begin[w + 1] - rb;
        int ind = -1;
        for (int i = 0; i < n_ind; ++i)
            if (seg_n_good[(long long)w * n_ind + i] != 0) ind = i;            // cpopulation.pyx:293: only these are aligned
        int s0 = 0, s1 = 0;
        if (ind >= 0) {
            s0 = seg_read_begin[(long long)w * n_ind + ind] - rb;
            s1 = seg_read_begin[(long long)w * n_ind + ind + 1] - rb;
        }
        const double log10E = 0.43429448190325182;                              // cgenotype.pyx:24
        const double* ll = loglik + b.pair_off[w];
        for (int h = lane; h < H; h += HS_GROUP) {
            const double* arr = ll + (long long)h * R;
            double sum = 0.0;
            for (int r0 = s0; r0 < s1; r0 += 8) {
                double v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = arr[min(r0 + k, s1 - 1)];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (r0 + k >= s1) break;
                    sum += log10E * v[k];
                }
            }
            if (out_hap_like) out_hap_like[hb + h] = sum;
            if (h < max_haps) mine[h] = -sum;                                   // hapScores[hap] = -hapLike
        }
    }
    __syncthreads();
    if (!live || lane != 0) return;
    if (H > max_haps || H == 0) { out_hap_score[w] = H == 0 ? 0 : -1; return; }
    // ascending walk without a sort: next = smallest (value, index) above the previous one
    double prev = 0.0;
    int prev_i = -1, n1 = 0, n2 = 0, clusters = 1;
    double dist = 0.0;
    for (int k = 0; k < H; ++k) {
        double best = 0.0;
        int bi = -1;
        for (int h = 0; h < H; ++h) {
            const double v = mine[h];
            const bool after = k == 0 || v > prev || (v == prev && h > prev_i);
            if (after && (bi < 0 || v < best)) { best = v; bi = h; }
        }
        if (k == 0) n1 = 1;
        else if (best - prev > 20) {                                            // vcfutils.pyx:1099-1104
            if (clusters == 1) dist = best - prev;
            if (clusters == 2) break;
            clusters = 2; n2 = 1;
        } else if (clusters == 1) ++n1;
        else ++n2;
        prev = best; prev_i = bi;
    }
    out_hap_score[w] = n1 + ((dist < 50 && dist > 0) ? n2 : 0);                 // :1109-1112
}

}  // namespace plat

using namespace plat;

PLAT_EXPORT int plat_em_window_batch(plat_ctx* ctx, int n_windows, int n_ind, int max_haps_per_window,
                                     const int32_t* win_hap_begin, const int64_t* gl_off, const int32_t* n_reads,
                                     const double* gl, int max_iters, int use_em_likelihoods, double* out_freq,
                                     double* out_em, int32_t* out_call, int32_t* out_iters, void* stream)
{
    if (!ctx || n_windows < 0 || n_ind < 1 || max_haps_per_window < 0 || max_iters < 0) return PLAT_ERR_INVALID;
    if (n_windows == 0) return PLAT_OK;
    if (!win_hap_begin || !gl_off || !n_reads || !gl || !out_freq || !out_em || !out_call) return PLAT_ERR_INVALID;
    size_t lds = (size_t)max_haps_per_window * sizeof(double) + 16;
    if (lds > 64 * 1024) return PLAT_ERR_INVALID;
    const size_t maxG = (size_t)max_haps_per_window * (max_haps_per_window + 1) / 2;
    const size_t csr_bytes = (size_t)n_ind * maxG * sizeof(double);
    PLAT_HIP(ctx, hipSetDevice(ctx->device));
    // likelihoods and responsibilities of one window in LDS when they fit: k_em_wide
    const size_t wide = (size_t)max_haps_per_window * 8 + 2 * csr_bytes + (size_t)n_ind * 12 + maxG * 4 + 64;
    const bool no_wide = getenv("PLAT_EM_NARROW") != nullptr;             // (read per call: the one-wave kernel, for measurements and the cross-check test)
    const size_t lds_dev = ctx->lds_max ? ctx->lds_max : 64 * 1024;     // what a workgroup of this device may ask for (160 KB on gfx950)
    if (wide <= 96 * 1024 && wide <= lds_dev && max_haps_per_window < 32768 && !no_wide) {
        const size_t pairs = (size_t)n_ind * maxG;
        const int threads = pairs > 128 ? 256 : (pairs > 64 ? 128 : 64);
        const size_t streams = (size_t)max_haps_per_window * (((size_t)n_ind * (max_haps_per_window + 1) + 7) & ~(size_t)7) * 8;
        const int use_streams = wide + streams <= 150 * 1024 && wide + streams <= lds_dev;
        const size_t lds_wide = wide + (use_streams ? streams : 0);
        if (lds_wide > 48 * 1024)
            PLAT_HIP(ctx, hipFuncSetAttribute((const void*)k_em_wide, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_wide));
        { PLAT_KT_BEGIN(ctx, PLAT_KT_EM, (hipStream_t)stream); hipLaunchKernelGGL(k_em_wide, dim3(n_windows), dim3(threads), lds_wide, (hipStream_t)stream, n_ind, win_hap_begin, gl_off, n_reads, gl,
                           max_iters, use_em_likelihoods, out_freq, out_em, out_call, out_iters, max_haps_per_window, (int)maxG, use_streams, (long long*)ctx->d_sticky); PLAT_KT_END(ctx, PLAT_KT_EM, (hipStream_t)stream); }
        PLAT_HIP(ctx, hipGetLastError());
        return PLAT_OK;
    }
    const int csr_in_lds = n_ind >= 8 && lds + csr_bytes <= 60 * 1024 && lds + csr_bytes <= lds_dev;   // responsibilities of one window next to the frequencies (pays with many samples)
    if (csr_in_lds) lds += csr_bytes;
    if (lds > 48 * 1024)
        PLAT_HIP(ctx, hipFuncSetAttribute((const void*)k_em, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    { PLAT_KT_BEGIN(ctx, PLAT_KT_EM, (hipStream_t)stream); hipLaunchKernelGGL(k_em, dim3(n_windows), dim3(64), lds, (hipStream_t)stream, n_ind, win_hap_begin, gl_off, n_reads, gl,
                       max_iters, use_em_likelihoods, out_freq, out_em, out_call, out_iters, max_haps_per_window, csr_in_lds, (long long*)ctx->d_sticky); PLAT_KT_END(ctx, PLAT_KT_EM, (hipStream_t)stream); }
    PLAT_HIP(ctx, hipGetLastError());
    return PLAT_OK;
}

PLAT_EXPORT int plat_variant_posterior_batch(plat_ctx* ctx, int n_vars, int n_ind, int max_haps_per_window,
                                             const int32_t* win_hap_begin, const int64_t* gl_off, const int32_t* n_reads,
                                             const double* gl, const double* freq, const int32_t* var_window,
                                             const int64_t* var_mask_off, const uint8_t* hap_has_var, const double* prior,
                                             double* out_posterior, void* stream)
{
    if (!ctx || n_vars < 0 || n_ind < 1 || max_haps_per_window < 0) return PLAT_ERR_INVALID;
    if (n_vars == 0) return PLAT_OK;
    if (!win_hap_begin || !gl_off || !n_reads || !gl || !freq || !var_window || !var_mask_off || !hap_has_var || !prior ||
        !out_posterior)
        return PLAT_ERR_INVALID;
    const size_t lds = (size_t)2 * max_haps_per_window * sizeof(double) + 16;
    if (lds > 64 * 1024) return PLAT_ERR_INVALID;
    PLAT_HIP(ctx, hipSetDevice(ctx->device));
    int rc = plat_reserve(ctx, ctx->pop_scratch, (size_t)n_vars * 2 * n_ind * sizeof(double));
    if (rc) return rc;
    { PLAT_KT_BEGIN(ctx, PLAT_KT_VARIANT_POSTERIOR, (hipStream_t)stream); hipLaunchKernelGGL(k_variant_posterior, dim3(n_vars), dim3(64), lds, (hipStream_t)stream, n_ind, win_hap_begin, gl_off,
                       n_reads, gl, freq, var_window, var_mask_off, hap_has_var, prior, (double*)ctx->pop_scratch.ptr,
                       out_posterior); PLAT_KT_END(ctx, PLAT_KT_VARIANT_POSTERIOR, (hipStream_t)stream); }
    PLAT_HIP(ctx, hipGetLastError());
    return PLAT_OK;
}

PLAT_EXPORT int plat_genotype_call_batch(plat_ctx* ctx, int n_sites, int n_ind, const int32_t* win_hap_begin,
                                         const int64_t* gl_off, const double* gl, const double* gof, const double* freq,
                                         const int32_t* site_window, const int32_t* site_nvar, const int64_t* site_vih_off,
                                         const int64_t* site_ref_off, const int32_t* var_in_hap, const int32_t* is_ref,
                                         const int64_t* lik_off,
                                         int32_t* out_phased, double* out_lik, double* out4, void* stream)
{
    if (!ctx || n_sites < 0 || n_ind < 1) return PLAT_ERR_INVALID;
    if (n_sites == 0) return PLAT_OK;
    if (!win_hap_begin || !gl_off || !gl || !gof || !freq || !site_window || !site_nvar || !site_vih_off || !site_ref_off || !var_in_hap ||
        !is_ref || !lik_off || !out_phased || !out_lik || !out4)
        return PLAT_ERR_INVALID;
    PLAT_HIP(ctx, hipSetDevice(ctx->device));
    const long long n = (long long)n_sites * n_ind;
    { PLAT_KT_BEGIN(ctx, PLAT_KT_GENOTYPE_CALL, (hipStream_t)stream); hipLaunchKernelGGL(k_genotype_call, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, (hipStream_t)stream, n_sites, n_ind,
                       win_hap_begin, gl_off, gl, gof, freq, site_window, site_nvar, site_vih_off, site_ref_off, var_in_hap,
                       is_ref, lik_off, out_phased, out_lik, out4); PLAT_KT_END(ctx, PLAT_KT_GENOTYPE_CALL, (hipStream_t)stream); }
    PLAT_HIP(ctx, hipGetLastError());
    return PLAT_OK;
}

PLAT_EXPORT int plat_haplotype_score_batch(plat_ctx* ctx, const plat_window_batch* batch, int n_ind, int max_haps_per_window,
                                           const int32_t* seg_read_begin, const int32_t* seg_n_good, const double* loglik,
                                           double* out_hap_like, int32_t* out_hap_score, void* stream)
{
    if (!ctx || !batch || n_ind < 1 || max_haps_per_window < 0) return PLAT_ERR_INVALID;
    if (batch->n_windows == 0) return PLAT_OK;
    if (!seg_read_begin || !seg_n_good || !loglik || !out_hap_score) return PLAT_ERR_INVALID;
    const size_t lds = (size_t)(64 / HS_GROUP) * max_haps_per_window * sizeof(double) + 16;
    if (lds > 64 * 1024) return PLAT_ERR_INVALID;
    PLAT_HIP(ctx, hipSetDevice(ctx->device));
    const unsigned nblk = (unsigned)((batch->n_windows + 64 / HS_GROUP - 1) / (64 / HS_GROUP));
    { PLAT_KT_BEGIN(ctx, PLAT_KT_HAPLOTYPE_SCORE, (hipStream_t)stream); hipLaunchKernelGGL(k_haplotype_score, dim3(nblk), dim3(64), lds, (hipStream_t)stream, *batch, n_ind, max_haps_per_window,
                       seg_read_begin, seg_n_good, loglik, out_hap_like, out_hap_score); PLAT_KT_END(ctx, PLAT_KT_HAPLOTYPE_SCORE, (hipStream_t)stream); }
    PLAT_HIP(ctx, hipGetLastError());
    return PLAT_OK;
}
