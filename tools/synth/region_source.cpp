// region_source.cpp -- libplat_synth.so: the synthetic WGS workload of BASELINE config 4 as a REGION SOURCE for plat_call_regions_stream
// (include/platypus_caller.h).  Benchmark tooling, not product: it stands where the reference's loader stands (loadBAMData for one
// 100 kb region at a time, variantcaller.pyx:935-1012) and produces, from seed (+) region index alone, what that loader would hand over --
// the region's reference and its reads as arrays (cAlignedRead fields) in the slot's pinned buffer, as ASCII or packed bytes.
//
// Recipe (SURVEY.md 8(d) cfg 4; the same as platypus_amd/synth.py::config4_region_arrays, drawn with its own generator so that a region
// costs a millisecond or two of one core instead of a quarter second): own contig r<index> = flank + region + flank of uniform ACGT;
// SNPs at snp_rate and 1..10 bp indels at indel_rate inside the region (never closer than 2 / 3+ bases); a diploid donor per sample,
// every variant on either haplotype with probability 1/2; depth x region_len / read_len reads of read_len bases from either haplotype,
// start uniform over the region, with the CIGAR an aligner would report; substitution errors at `err`; qualities ~ clipped N(35, 5)
// (rows of a shared tape of such draws); mapq 60; flags 3 | 16 at random; sorted by position.
#include <algorithm>
#include <chrono>
#include <immintrin.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <thread>
#include <atomic>

#include "../../include/platypus_caller.h"

#define SYNTH_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

struct Rng {                                                               // xoshiro256**, seeded through splitmix64
    uint64_t s[4];
    static uint64_t sm(uint64_t& x) { uint64_t z = (x += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
    Rng(uint64_t seed, uint64_t stream) { uint64_t x = seed ^ (stream * 0xD1342543DE82EF95ull + 0x2545F4914F6CDD1Dull); for (auto& v : s) v = sm(x); }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() { const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17; s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45); return r; }
    double uni() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    uint32_t below(uint32_t n) { return (uint32_t)(((next() >> 32) * (uint64_t)n) >> 32); }
    int poisson(double lam) { int k = 0; double t = -std::log(1.0 - uni()) / lam; while (t < 1.0) { ++k; t += -std::log(1.0 - uni()) / lam; } return k; }
    int geometric(double p) { return 1 + (int)std::floor(std::log(1.0 - uni()) / std::log(1.0 - p)); }
};

struct Var { int pos, kind, len; std::string bases; };                     // kind 0 SNP, 1 insertion after pos, 2 deletion of (pos, pos+len]

}  // namespace

struct plat_synth {
    uint64_t seed;
    int regionLen, flank, nSamples, depth, readLen, encoding;
    double snpRate, indelRate, err;
    std::vector<int32_t> index;                                            // job position -> region id
    uint8_t* mem; size_t slotBytes; int nSlots;
    std::vector<uint8_t> tape;                                             // quality tape
    std::vector<uint8_t> tape4;                                            // the same << 2: the quality bits of a packed byte
    std::vector<int32_t> gapTape;                                          // 64 K draws of the distance to the next substitution error (geometric)
    // variant model: indel length 1 + min(indelMax - 1, geometric(indelP) - 1); counts Poisson(rate x length) unless a [min, max] range is set
    int indelMax = 10, nIndelMin = -1, nIndelMax = -1, nSnpMin = -1, nSnpMax = -1;
    double indelP = 0.4;
    // resident mode (plat_synth_pregenerate): region k of the list lives in slot k for the life of the source; `resident` holds the filled
    // structs, devDelta the distance from a host address inside `mem` to its device mirror (0: none)
    std::vector<plat_region> resident;
    long long devDelta = 0;
    bool haveMirror = false;
    long long planted = 0, reads = 0;                                      // totals over the regions loaded (statistics)
    long long phaseNs[6] = {0, 0, 0, 0, 0, 0};                             // reference, variants, haplotypes, read starts + sort, reads, rest
};

static size_t align64(size_t x) { return (x + 63) & ~(size_t)63; }

static size_t slotBytesFor(int regionLen, int flank, int nSamples, int depth, int readLen, int encoding) {
    const size_t n = (size_t)regionLen + 2 * (size_t)flank, nr = (size_t)((double)depth * regionLen / readLen) + 1, nb = nr * (size_t)readLen;
    size_t perSample = align64(nb + 64) * (encoding == PLAT_READS_PACKED ? 1 : 2) + align64((nr + 1) * 8) + 4 * align64(nr * 4 + 4) + align64(nr + 4) +
                       align64((nr + 1) * 4) + align64((3 * nr + 64) * 4);
    return align64(n + 64) + 256 + align64((size_t)nSamples * sizeof(plat_sample_reads)) + (size_t)nSamples * perSample + 4096;
}

SYNTH_EXPORT size_t plat_synth_slot_bytes(int region_len, int flank, int n_samples, int depth, int read_len, int encoding) {
    return slotBytesFor(region_len, flank, n_samples, depth, read_len, encoding);
}

SYNTH_EXPORT int plat_synth_create(uint64_t seed, int region_len, int flank, int n_samples, int depth, int read_len, double snp_rate, double indel_rate,
                                   double err, int encoding, const int32_t* region_index, int n_regions, void* slot_memory, size_t slot_bytes, int n_slots,
                                   plat_synth** out)
{
    if (!out || region_len < 200 || flank < 0 || n_samples < 1 || depth < 1 || read_len < 20 || read_len > 10000 || !region_index || n_regions < 0 ||
        !slot_memory || n_slots < 1 || (encoding != PLAT_READS_ASCII && encoding != PLAT_READS_PACKED) || err < 0 || err >= 0.5)
        return -1;
    if (slot_bytes < slotBytesFor(region_len, flank, n_samples, depth, read_len, encoding)) return -1;
    plat_synth* g = new plat_synth();
    g->seed = seed; g->regionLen = region_len; g->flank = flank; g->nSamples = n_samples; g->depth = depth; g->readLen = read_len; g->encoding = encoding;
    g->snpRate = snp_rate; g->indelRate = indel_rate; g->err = err;
    g->index.assign(region_index, region_index + n_regions);
    g->mem = (uint8_t*)slot_memory; g->slotBytes = slot_bytes; g->nSlots = n_slots;
    // the quality tape: 64 K draws of clip(round(N(35, 5)), 2, 41) (+ one read of slack), Box-Muller; a read's qualities are a row of it
    const size_t T = (1u << 16) + (size_t)read_len + 64;
    g->tape.resize(T);
    Rng r(seed, 0xFFFFFFFFull);
    for (size_t i = 0; i < T; i += 2) {
        const double u1 = 1.0 - r.uni(), u2 = r.uni(), m = std::sqrt(-2.0 * std::log(u1));
        const double z[2] = {m * std::cos(6.283185307179586 * u2), m * std::sin(6.283185307179586 * u2)};
        for (int k = 0; k < 2 && i + k < T; ++k) g->tape[i + k] = (uint8_t)std::min(41.0, std::max(2.0, std::floor(35.0 + 5.0 * z[k] + 0.5)));
    }
    g->tape4.resize(g->tape.size());
    for (size_t i = 0; i < g->tape.size(); ++i) g->tape4[i] = (uint8_t)(g->tape[i] << 2);
    if (err > 0) {                                                           // no logarithm per error in the loaders: a draw picks an entry
        g->gapTape.resize(1u << 16);
        Rng rg(seed, 0xFFFFFFFDull);
        const double logKeep = std::log(1.0 - err);
        for (int32_t& x : g->gapTape) x = (int32_t)std::min(1e9, std::floor(std::log(1.0 - rg.uni()) / logKeep));
    }
    *out = g;
    return 0;
}

// BASELINE config 3's model: n_indel_min..max indels of 1..indel_max_len bases (geometric, p = indel_p) and n_snp_min..max
// SNPs per region instead of Poisson counts; lowq_frac of the bases of the quality tape redrawn uniformly below Q20.
SYNTH_EXPORT int plat_synth_set_model(plat_synth* g, int indel_max_len, double indel_p, int n_indel_min, int n_indel_max, int n_snp_min, int n_snp_max,
                                      double lowq_frac)
{
    if (!g || indel_max_len < 1 || indel_max_len > 1000 || indel_p <= 0 || indel_p > 1 || lowq_frac < 0 || lowq_frac > 1) return -1;
    g->indelMax = indel_max_len; g->indelP = indel_p;
    g->nIndelMin = n_indel_min; g->nIndelMax = n_indel_max; g->nSnpMin = n_snp_min; g->nSnpMax = n_snp_max;
    if (lowq_frac > 0) {
        Rng r(g->seed, 0xFFFFFFFEull);
        for (uint8_t& q : g->tape) if (r.uni() < lowq_frac) q = (uint8_t)(2 + r.below(18));
        for (size_t i = 0; i < g->tape.size(); ++i) g->tape4[i] = (uint8_t)(g->tape[i] << 2);
    }
    return 0;
}

SYNTH_EXPORT void plat_synth_destroy(plat_synth* g) { delete g; }
SYNTH_EXPORT long long plat_synth_planted(const plat_synth* g) { return g ? g->planted : 0; }
SYNTH_EXPORT void plat_synth_phase_seconds(const plat_synth* g, double* out6) { for (int k = 0; k < 6; ++k) out6[k] = g ? 1e-9 * (double)g->phaseNs[k] : 0.0; }

namespace {
struct Carve {
    uint8_t* p; uint8_t* end;
    template <class T> T* take(size_t n) { T* r = (T*)p; p += align64(n * sizeof(T)); return p <= end ? r : nullptr; }
};
struct Scratch {
    std::vector<uint8_t> hs[2];
    std::vector<uint8_t> hc[2];                                            // 2-bit codes of hs (+ slack), packed tables only
    std::vector<int32_t> h2r[2];
    std::vector<std::pair<int32_t, int32_t>> gaps[2];                      // per haplotype, ascending: hap positions [a, b) a read must not touch to be one plain match
    std::vector<int32_t> i0, posOf, order, tmpOrder;
    std::vector<uint64_t> draw, sdraw;
    std::vector<Var> vars;
    std::vector<int> spots, kinds;
};
}  // namespace

// The plat_region_load_fn: region `index` of the job into slot `slot`.
SYNTH_EXPORT int plat_synth_load(void* user, int index, int slot, plat_region* out)
{
    plat_synth* g = (plat_synth*)user;
    if (!g || !out || index < 0 || index >= (int)g->index.size() || slot < 0 || slot >= g->nSlots) return -1;
    static thread_local Scratch S;
    typedef std::chrono::steady_clock Clk;
    auto tmark = Clk::now();
    auto lap = [&](int k) { const auto now = Clk::now(); __atomic_add_fetch(&g->phaseNs[k], (long long)std::chrono::duration_cast<std::chrono::nanoseconds>(now - tmark).count(), __ATOMIC_RELAXED); tmark = now; };
    const int id = g->index[(size_t)index];
    Rng rng(g->seed, (uint64_t)(uint32_t)id);
    const int L = g->readLen, start = g->flank, end = g->flank + g->regionLen, n = g->regionLen + 2 * g->flank;
    Carve cv{g->mem + (size_t)slot * g->slotBytes, g->mem + (size_t)(slot + 1) * g->slotBytes};
    uint8_t* ref = cv.take<uint8_t>((size_t)n + 64);
    char* chrom = cv.take<char>(64);
    plat_sample_reads* samples = cv.take<plat_sample_reads>((size_t)g->nSamples);
    if (!ref || !chrom || !samples) return -3;
    {   // 32 bases per draw, four per table look-up (the slot has 64 bytes of slack behind the reference)
        static const struct Lut { uint32_t w[256]; Lut() { for (int b = 0; b < 256; ++b) { uint32_t v = 0; for (int k = 0; k < 4; ++k) v |= (uint32_t)"ACGT"[(b >> (2 * k)) & 3] << (8 * k); w[b] = v; } } } lut;
        for (int i = 0; i < n; i += 32) {
            uint64_t w = rng.next();
            for (int k = 0; k < 8; ++k, w >>= 8) memcpy(ref + i + 4 * k, &lut.w[w & 255], 4);
        }
    }
    memset(ref + n, 0, 64);
    lap(0);
    snprintf(chrom, 64, "r%d", id);
    // ---- planted variants
    std::vector<Var>& vars = S.vars;
    vars.clear();
    {
        const int nSnp = g->nSnpMax >= 0 ? g->nSnpMin + (int)rng.below((uint32_t)(g->nSnpMax - g->nSnpMin + 1)) : rng.poisson(g->regionLen * g->snpRate + 1e-12);
        const int nInd = g->nIndelMax >= 0 ? g->nIndelMin + (int)rng.below((uint32_t)(g->nIndelMax - g->nIndelMin + 1)) : rng.poisson(g->regionLen * g->indelRate + 1e-12);
        S.kinds.assign((size_t)nSnp, 0); S.kinds.insert(S.kinds.end(), (size_t)nInd, 1);
        for (size_t i = S.kinds.size(); i > 1; --i) std::swap(S.kinds[i - 1], S.kinds[rng.below((uint32_t)i)]);
        const int lo = start + 20, hi = end - 40 - (g->indelMax > 10 ? g->indelMax : 0);
        S.spots.clear();
        if (hi > lo) for (size_t i = 0; i < S.kinds.size(); ++i) S.spots.push_back(lo + (int)rng.below((uint32_t)(hi - lo)));
        std::sort(S.spots.begin(), S.spots.end());
        int last = start + 20;
        for (size_t i = 0; i < S.spots.size(); ++i) {
            const int p = S.spots[i];
            if (p < last) continue;
            if (S.kinds[i] == 0) {
                const char* B = "ACGT";
                const int cur = (int)(strchr(B, ref[p]) - B);
                vars.push_back(Var{p, 0, 1, std::string(1, B[(cur + 1 + (int)rng.below(3)) & 3])});
                last = p + 2;
            } else {
                const int k = 1 + std::min(g->indelMax - 1, rng.geometric(g->indelP) - 1);
                if (rng.uni() < 0.5) {
                    std::string b((size_t)k, 'A');
                    for (char& c : b) c = "ACGT"[rng.below(4)];
                    vars.push_back(Var{p, 1, k, b});
                    last = p + 3;
                } else if (p + 1 + k < end) {
                    vars.push_back(Var{p, 2, k, std::string()});
                    last = p + k + 3;
                }
            }
        }
    }
    lap(1);
    const int nReads = (int)((double)g->depth * g->regionLen / L);
    long long nReadsAll = 0;
    for (int si = 0; si < g->nSamples; ++si) {
        // ---- the donor's two haplotypes: bytes, hap position -> reference position (-1 = inserted), reference position -> hap position
        for (int h = 0; h < 2; ++h) {
            std::vector<uint8_t>& hs = S.hs[h];
            std::vector<int32_t>& h2r = S.h2r[h];
            hs.clear(); h2r.clear();
            std::vector<std::pair<int32_t, int32_t>>& gaps = S.gaps[h];
            gaps.clear();
            int cur = 0;
            auto copyTo = [&](int upto) {                                    // reference bases [cur, upto) as they are
                if (upto > cur) {
                    const size_t at = hs.size(), m = (size_t)(upto - cur);
                    hs.resize(at + m); h2r.resize(at + m);
                    memcpy(hs.data() + at, ref + cur, m);
                    for (size_t x = 0; x < m; ++x) h2r[at + x] = cur + (int32_t)x;
                    cur = upto;
                }
            };
            for (const Var& v : vars) {
                if (!(rng.next() >> 63)) continue;                          // this haplotype does not carry it
                if (v.kind == 0) { copyTo(v.pos); hs.push_back((uint8_t)v.bases[0]); h2r.push_back(v.pos); cur = v.pos + 1; }
                else if (v.kind == 1) {                                     // a read holding any inserted base is not a plain match
                    copyTo(v.pos + 1);
                    gaps.push_back({(int32_t)hs.size(), (int32_t)(hs.size() + v.bases.size())});
                    for (char c : v.bases) { hs.push_back((uint8_t)c); h2r.push_back(-1); }
                } else {                                                    // nor one holding the bases on both sides of a deletion
                    copyTo(v.pos + 1); cur = v.pos + 1 + v.len;
                    gaps.push_back({(int32_t)hs.size() - 1, (int32_t)hs.size() + 1});
                }
            }
            copyTo(n);
        }
        if ((int)S.hs[0].size() < L || (int)S.hs[1].size() < L) return -1;
        if (g->encoding == PLAT_READS_PACKED)
            for (int h = 0; h < 2; ++h) {                                   // a read without errors is then one OR of two byte rows
                const size_t m = S.hs[h].size();
                S.hc[h].resize(m + 64);
                const uint8_t* __restrict a = S.hs[h].data();
                uint8_t* __restrict c = S.hc[h].data();
                for (size_t i = 0; i < m; ++i) c[i] = (uint8_t)((a[i] >> 1) & 3);
                memset(c + m, 0, 64);
            }
        lap(2);
        // ---- read starts: ONE 64-bit draw per read (bit 0 haplotype, bit 1 strand, bits 2..17 quality row, bits 32..63 the start, uniform
        // over the haplotype's stretch of the region), then sorted by reference position (two stable 9-bit radix passes over pos - lo)
        S.i0.resize((size_t)nReads); S.posOf.resize((size_t)nReads); S.draw.resize((size_t)nReads); S.order.resize((size_t)nReads);
        S.tmpOrder.resize((size_t)nReads);
        int hLo[2], hHi[2];
        for (int h = 0; h < 2; ++h) {                                       // hap positions of the reference positions start-L+10 and end-10
            const std::vector<int32_t>& m = S.h2r[h];
            auto at = [&](int x) { int i = std::min<int>(std::max(x, 0), (int)m.size() - 1); while (i > 0 && (m[(size_t)i] < 0 || m[(size_t)i] > x)) --i;
                                   while (i + 1 < (int)m.size() && (m[(size_t)i] < 0 || m[(size_t)i] < x)) ++i; return i; };
            hLo[h] = at(start - L + 10); hHi[h] = std::max(hLo[h] + 1, at(end - 10));
        }
        const int lo = std::max(0, start - L);
        for (int r = 0; r < nReads; ++r) {
            const uint64_t w = rng.next();
            const int h = (int)(w & 1);
            int i0 = hLo[h] + (int)(((w >> 32) * (uint64_t)(hHi[h] - hLo[h])) >> 32);
            i0 = std::min(i0, (int)S.hs[h].size() - L);
            while (i0 > 0 && S.h2r[h][(size_t)i0] < 0) --i0;                // a read starts on a reference base
            S.draw[(size_t)r] = w; S.i0[(size_t)r] = i0; S.posOf[(size_t)r] = S.h2r[h][(size_t)i0];
        }
        {
            uint32_t hist[512];
            const int32_t* key = S.posOf.data();
            int32_t* src = S.tmpOrder.data();
            int32_t* dst = S.order.data();
            for (int r = 0; r < nReads; ++r) src[r] = r;
            for (int pass = 0; pass < 2; ++pass) {                          // positions span < 2^18
                memset(hist, 0, sizeof hist);
                const int sh = 9 * pass;
                for (int r = 0; r < nReads; ++r) ++hist[((uint32_t)(key[src[r]] - lo) >> sh) & 511];
                uint32_t run = 0;
                for (int b = 0; b < 512; ++b) { const uint32_t c = hist[b]; hist[b] = run; run += c; }
                for (int r = 0; r < nReads; ++r) dst[hist[((uint32_t)(key[src[r]] - lo) >> sh) & 511]++] = src[r];
                std::swap(src, dst);
            }
            if (src != S.order.data()) S.order.swap(S.tmpOrder);
        }
        lap(3);
        // ---- the table
        const size_t nb = (size_t)nReads * (size_t)L;
        uint8_t* seq = cv.take<uint8_t>(nb + 64);
        uint8_t* qual = g->encoding == PLAT_READS_PACKED ? nullptr : cv.take<uint8_t>(nb + 64);
        int64_t* off = cv.take<int64_t>((size_t)nReads + 1);
        int32_t* pos = cv.take<int32_t>((size_t)nReads + 1);
        int32_t* endp = cv.take<int32_t>((size_t)nReads + 1);
        int32_t* flags = cv.take<int32_t>((size_t)nReads + 1);
        int32_t* mate = cv.take<int32_t>((size_t)nReads + 1);
        uint8_t* mapq = cv.take<uint8_t>((size_t)nReads + 4);
        int32_t* cigoff = cv.take<int32_t>((size_t)nReads + 1);
        const size_t cigCap = 3 * (size_t)nReads + 64;
        int16_t* cigar = cv.take<int16_t>(2 * cigCap);
        if (!seq || (!qual && g->encoding != PLAT_READS_PACKED) || !off || !pos || !endp || !flags || !mate || !mapq || !cigoff || !cigar) return -3;
        size_t nc = 0;
        uint8_t tmpS[10064];
        const int32_t* gapTape = g->gapTape.data();
        auto nextGap = [&]() -> long long { return gapTape[rng.next() >> 48]; };
        long long toErr = g->err > 0 ? nextGap() : (1ll << 62);             // bases until the next substitution error
        // the per-read fields that need no thought, in tight loops; the draws and starts in sorted order (one gather instead of two
        // dependent look-ups per read in the big loop)
        S.tmpOrder.resize((size_t)nReads);
        int32_t* sI0 = S.tmpOrder.data();
        for (int k = 0; k < nReads; ++k) sI0[k] = S.i0[(size_t)S.order[(size_t)k]];
        S.sdraw.resize((size_t)nReads);
        for (int k = 0; k < nReads; ++k) S.sdraw[(size_t)k] = S.draw[(size_t)S.order[(size_t)k]];
        for (int k = 0; k < nReads; ++k) off[k] = (int64_t)k * L;
        for (int k = 0; k < nReads; ++k) { mate[k] = -1; flags[k] = 3 | ((S.sdraw[(size_t)k] & 2) ? 16 : 0); }
        memset(mapq, 60, (size_t)nReads);
        size_t gapAt[2] = {0, 0};                                           // reads come sorted: a haplotype's gaps are passed once
        for (int k = 0; k < nReads; ++k) {
            const int i0 = sI0[k];
            const uint64_t w = S.sdraw[(size_t)k];
            const int h = (int)(w & 1);
            const uint8_t* src = S.hs[h].data() + i0;
            const int32_t* map = S.h2r[h].data() + i0;
            pos[k] = map[0];
            cigoff[k] = (int32_t)nc;
            const std::vector<std::pair<int32_t, int32_t>>& gv = S.gaps[h];
            while (gapAt[h] < gv.size() && gv[gapAt[h]].second <= i0) ++gapAt[h];
            const bool plain = gapAt[h] >= gv.size() || gv[gapAt[h]].first >= i0 + L;
            if (plain) {
                cigar[2 * nc] = 0; cigar[2 * nc + 1] = (int16_t)L; ++nc;
                endp[k] = map[0] + L;
            } else {
                int lastRef = -1;
                const size_t first = nc;
                for (int i = 0; i < L; ++i) {
                    int op, ln = 1;
                    if (map[i] < 0) op = 1;
                    else {
                        if (lastRef >= 0 && map[i] - lastRef > 1) {
                            if (nc >= cigCap) return -8;
                            cigar[2 * nc] = 2; cigar[2 * nc + 1] = (int16_t)(map[i] - lastRef - 1); ++nc;
                        }
                        op = 0; lastRef = map[i];
                    }
                    if (nc > first && cigar[2 * (nc - 1)] == op) cigar[2 * (nc - 1) + 1] = (int16_t)(cigar[2 * (nc - 1) + 1] + ln);
                    else { if (nc >= cigCap) return -8; cigar[2 * nc] = (int16_t)op; cigar[2 * nc + 1] = (int16_t)ln; ++nc; }
                }
                endp[k] = lastRef + 1;
            }
            const uint8_t* q = g->tape.data() + ((w >> 2) & 0xFFFFu);
            uint8_t* ds = seq + (size_t)k * L;
            if (g->encoding == PLAT_READS_PACKED) {
                // one OR of two byte rows in blocks of 32 bytes: what runs past the read lands in the next read's row (written after this
                // one) or in the table's slack; then the substitution errors, on the codes (A C G T = 0 1 3 2)
                uint8_t* __restrict d = ds;
                const uint8_t* __restrict cb = S.hc[h].data() + i0;
                const uint8_t* __restrict qb = g->tape4.data() + ((w >> 2) & 0xFFFFu);
                for (int i = 0; i < L; i += 32)
                    _mm256_storeu_si256((__m256i*)(d + i), _mm256_or_si256(_mm256_loadu_si256((const __m256i*)(cb + i)), _mm256_loadu_si256((const __m256i*)(qb + i))));
                while (toErr < L) {                                         // (rare: one read in six at 0.1 %)
                    static const uint8_t idxOf[4] = {0, 1, 3, 2}, codeOf[4] = {0, 1, 3, 2};
                    const uint8_t c = cb[toErr];
                    ds[toErr] = (uint8_t)(codeOf[(idxOf[c] + 1 + (int)rng.below(3)) & 3] | qb[toErr]);
                    toErr += 1 + nextGap();
                }
            } else {
                const uint8_t* s = src;
                if (toErr < L) {
                    memcpy(tmpS, src, (size_t)L);
                    while (toErr < L) {
                        const char* B = "ACGT";
                        const char* at = strchr(B, tmpS[toErr]);
                        tmpS[toErr] = (uint8_t)B[((at ? (int)(at - B) : 0) + 1 + (int)rng.below(3)) & 3];
                        toErr += 1 + nextGap();
                    }
                    s = tmpS;
                }
                memcpy(ds, s, (size_t)L);
                memcpy(qual + (size_t)k * L, q, (size_t)L);
            }
            toErr -= L;
        }
        lap(4);
        off[nReads] = (int64_t)nb; cigoff[nReads] = (int32_t)nc;
        memset(seq + nb, 0, 64);
        if (qual) memset(qual + nb, 0, 64);
        plat_sample_reads& sr = samples[si];
        memset(&sr, 0, sizeof sr);
        plat_read_table& t = sr.reads;
        t.n_reads = nReads; t.encoding = g->encoding; t.seq = seq; t.qual = qual ? qual : seq; t.off = off; t.pos = pos; t.end = endp; t.mapq = mapq;
        t.flags = flags; t.mate_pos = mate; t.cigar = cigar; t.cig_off = cigoff;
        {   // what a loader knows of its reads as it appends them (ReadArray.__longestRead, cwindow.pyx:173-174): handed over with the table
            int lg = 0; int64_t mb = 0;
            for (int r = 0; r < nReads; ++r) { lg = std::max(lg, endp[r] - pos[r]); mb = std::max<int64_t>(mb, off[r + 1] - off[r]); }
            t.longest_read = lg; t.most_bases = (int32_t)mb;
        }
        static const int64_t zero64[1] = {0};
        static const int32_t zero32[1] = {0};
        for (plat_read_table* e : {&sr.bad_reads, &sr.broken_mates}) {
            e->n_reads = 0; e->encoding = PLAT_READS_ASCII; e->seq = seq; e->qual = seq; e->off = zero64; e->pos = zero32; e->end = zero32; e->mapq = mapq;
            e->flags = zero32; e->mate_pos = zero32; e->cigar = cigar; e->cig_off = zero32;
        }
        nReadsAll += nReads;
    }
    out->chrom = chrom; out->start = start; out->end = end; out->contig_seq = ref; out->contig_len = n; out->samples = samples;
    __atomic_add_fetch(&g->planted, (long long)vars.size(), __ATOMIC_RELAXED);
    __atomic_add_fetch(&g->reads, nReadsAll, __ATOMIC_RELAXED);
    return 0;
}

// address of the load function, for callers that pass it on as a plain pointer (ctypes)
SYNTH_EXPORT void* plat_synth_load_fn(void) { return (void*)&plat_synth_load; }

// ---- resident mode: "inputs already resident when the timed region starts" ------------------------------------------------------------
// Every region of the list is generated ONCE, region k into slot k (needs n_slots >= the number of regions), on n_threads threads; the
// caller may then upload the whole slot memory to the device in one copy and name the mirror (plat_synth_set_device_mirror): the resident
// loader hands out the stored structs with dev_seq / dev_qual pointing into it, and a timed run neither generates nor moves read bytes.
SYNTH_EXPORT int plat_synth_pregenerate(plat_synth* g, int n_threads) {
    if (!g || (size_t)g->nSlots < g->index.size()) return -1;
    const int n = (int)g->index.size();
    g->resident.assign((size_t)n, plat_region());
    std::atomic<int> next(0), err(0);
    auto work = [&] {
        for (int k = next.fetch_add(1); k < n; k = next.fetch_add(1)) {
            const int rc = plat_synth_load(g, k, k, &g->resident[(size_t)k]);
            if (rc != 0) err.store(rc);
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < std::max(1, n_threads); ++t) th.emplace_back(work);
    work();
    for (std::thread& t : th) t.join();
    return err.load();
}
SYNTH_EXPORT int plat_synth_set_device_mirror(plat_synth* g, const void* device_base) {
    if (!g) return -1;
    g->haveMirror = device_base != nullptr;
    g->devDelta = device_base ? (long long)((intptr_t)device_base - (intptr_t)g->mem) : 0;
    return 0;
}
static int plat_synth_load_resident(void* user, int index, int slot, plat_region* out) {
    (void)slot;
    plat_synth* g = (plat_synth*)user;
    if (!g || !out || index < 0 || (size_t)index >= g->resident.size()) return -2;
    *out = g->resident[(size_t)index];
    if (g->haveMirror) {
        out->dev_contig_seq = out->contig_seq + g->devDelta;            // (the slot holds the contig too: the reference is resident)
        // (the per-sample structs live in the slot and are shared by every hand-out: the pointers written are the same every time)
        plat_sample_reads* sm = const_cast<plat_sample_reads*>(out->samples);
        for (int i = 0; i < g->nSamples; ++i) {
            plat_read_table& t = sm[i].reads;
            t.dev_seq = t.seq + g->devDelta;
            t.dev_qual = t.encoding == PLAT_READS_ASCII ? t.qual + g->devDelta : nullptr;
            // (the whole slot is mirrored: the per-read arrays are resident too)
            auto dev = [g](const void* p) { return (const void*)((const uint8_t*)p + g->devDelta); };
            t.dev_off = (const int64_t*)dev(t.off); t.dev_pos = (const int32_t*)dev(t.pos); t.dev_end = (const int32_t*)dev(t.end);
            t.dev_mapq = (const uint8_t*)dev(t.mapq); t.dev_flags = (const int32_t*)dev(t.flags); t.dev_cigar = (const int16_t*)dev(t.cigar);
            t.dev_cig_off = (const int32_t*)dev(t.cig_off);
        }
    }
    return 0;
}
SYNTH_EXPORT void* plat_synth_load_resident_fn(void) { return (void*)&plat_synth_load_resident; }
