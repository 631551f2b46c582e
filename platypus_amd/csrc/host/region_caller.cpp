// region_caller.cpp -- libplat_caller.so: the region loop around the device hot path (include/platypus_caller.h).
//
//   generateVariantsInRegion      src/cython/variantcaller.pyx:412-531   (BAM candidates; assembler / source VCFs not built here)
//   callVariantsInRegion          src/cython/variantcaller.pyx:535-615
//   callVariantsInWindow          src/cython/variantcaller.pyx:74-141
//   ReadArray window pointers     src/cython/cwindow.pyx:176-264
//   Haplotype construction        src/cython/chaplotype.pyx:127-191,397-449
//   getFilteredHaplotypes, computeBestScoreForGenotype   src/cython/variantFilter.pyx:237-283,377-506
//   mergeHaplotypes               src/cython/variantcaller.pyx:325-383
//   Population.setup / call, vcfINFO, vcfFILTER, outputCallToVCF   (device stages + records.hpp)
//
// Host code only: every O(reads) stage is a call into libplat_mi355x.so (include/platypus_mi355x.h) on device pointers.
// Regions are processed in chunks; a chunk goes through
//   A  upload of its reads (one table) + candidate scan            plat_candidates_batch
//   B  host: merge / normalise / filter candidates, windows, window pointers, haplotypes (greedy rounds: plat_align_window_batch)
//   C  window read slices gathered on the device, likelihoods, genotype likelihoods, HapScore, EM
//      plat_gather_reads, plat_align_window_batch_async, plat_genotype_window_batch, plat_haplotype_score_batch, plat_em_window_batch
//   D  host: priors, variant masks -> posteriors (plat_variant_posterior_batch) -> INFO variants, call sites
//   E  read statistics + per-site genotype calls                   plat_variant_read_stats_batch, plat_genotype_call_batch
//   F  host: INFO / FILTER arithmetic, record text
// on one worker thread with its own plat_ctx and stream; several workers run side by side, so the uploads, kernels and host
// stages of different chunks overlap.  Same text as platypus_amd/caller.py::callVariantsInRegions (tests/test_native_caller_*.py).
#include <atomic>
#ifdef PLAT_HOSTPROF
#include <x86intrin.h>
#include <map>
#endif
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_map>

#include "../../../include/platypus_caller.h"
#include "../../../include/platypus_mi355x.h"
#include "records.hpp"
#include "variants.hpp"

#define CALLER_EXPORT extern "C" __attribute__((visibility("default")))

namespace plathost {

typedef std::chrono::steady_clock Clock;
static inline double secs(Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double>(b - a).count(); }
#ifdef PLAT_HOSTPROF                                                  // (local measurement builds only: cycle counts of named scopes)
static const char* g_profName[256];
static std::atomic<unsigned long long> g_profCyc[256], g_profCalls[256];
static std::atomic<int> g_profN{0};
static int profId(const char* n) { const int k = g_profN.fetch_add(1); g_profName[k] = n; return k; }
struct ProfScope { int k; unsigned long long t0; ProfScope(int k_) : k(k_), t0(__rdtsc()) {}
                   ~ProfScope() { g_profCyc[k].fetch_add(__rdtsc() - t0, std::memory_order_relaxed); g_profCalls[k].fetch_add(1, std::memory_order_relaxed); } };
#define PROF_CAT2(a, b) a##b
#define PROF_CAT(a, b) PROF_CAT2(a, b)
#define PROF(name) static const int PROF_CAT(profid_, __LINE__) = profId(name); ProfScope PROF_CAT(prof_, __LINE__)(PROF_CAT(profid_, __LINE__))
static void profDump(double n) {
    std::map<std::string, std::pair<unsigned long long, unsigned long long>> m;
    for (int k = 0; k < g_profN.load(); ++k) { m[g_profName[k]].first += g_profCyc[k].exchange(0); m[g_profName[k]].second += g_profCalls[k].exchange(0); }
    for (auto& kv : m) fprintf(stderr, "  [prof] %-28s %9.1f kcycles/region %8.1f calls/region\n", kv.first.c_str(), 1e-3 * (double)kv.second.first / n, (double)kv.second.second / n);
}
#else
#define PROF(name)
static void profDump(double) {}
#endif
// PLAT_CALLER_TRACE=1 (measurement): of every stage's seconds, the part spent waiting for the device; [8] host, [9] wait (under the stats mutex)
static double g_stageWait[10];
static void traceStages(const plat_caller_stats& st) {
    const char* e = getenv("PLAT_CALLER_TRACE");
    if (e && e[0] == '1') {
        static const char* names[8] = {"upload", "candidate_scan", "variants_windows_haplotypes", "greedy_rounds", "window_batch", "posteriors",
                                       "read_stats_calls", "text"};
        const double n = (double)std::max<int64_t>(1, st.n_regions);
        fprintf(stderr, "[plat_caller] per region, worker seconds (of which waiting for the device):");
        for (int k = 0; k < 8; ++k) fprintf(stderr, " %s %.1f us (%.1f)", names[k], 1e6 * st.seconds_stage[k] / n, 1e6 * g_stageWait[k] / n);
        fprintf(stderr, "; host %.1f us, wait %.1f us\n", 1e6 * g_stageWait[8] / n, 1e6 * g_stageWait[9] / n);
        profDump(n);
    }
    for (double& x : g_stageWait) x = 0;
}

struct DeviceError : std::runtime_error {
    int code;
    DeviceError(int c, const std::string& where) : std::runtime_error(where + ": device error " + std::to_string(c) + " (" + plat_strerror(c) + ")"), code(c) {}
};
static inline void ck(int rc, const char* where) { if (rc != PLAT_OK) throw DeviceError(rc, where); }
// errors one calling window's data can cause (retried window by window, the guilty window skipped) -- as opposed to the runtime's
static inline bool windowClassError(int code) {
    return code == PLAT_ERR_BAD_INPUT || code == PLAT_ERR_OVERFLOW || code == PLAT_ERR_INVALID || code == PLAT_ERR_HAP_TOO_LONG ||
           code == PLAT_ERR_HAP_TOO_SHORT || code == PLAT_ERR_BAD_HINTS;
}

// ---- grow-only buffers: pinned host + device mirror -----------------------------------------------------------------------------
struct Slot;                                                              // one worker's device context
template <class T> struct Staged {
    T* h = nullptr; T* d = nullptr; size_t hcap = 0, dcap = 0, n = 0;
    bool view = false;                                                     // h / d point into an arena (Layout): nothing owned
    // zeroStream != nullptr: a grown device buffer is zeroed once, on that stream (blob slack must hold 7-bit bytes for the kernels
    // that validate whole dwords; afterwards it only ever holds old, valid bytes)
    void reserve(plat_ctx* ctx, size_t want, bool host = true, bool dev = true, void* zeroStream = nullptr) {
        if (host && want > hcap) {
            const size_t ncap = want + want / 2 + 64;
            T* nh = nullptr;
            ck(plat_host_alloc(ctx, ncap * sizeof(T), (void**)&nh), "plat_host_alloc");
            if (h) { if (n) memcpy(nh, h, std::min(n, hcap) * sizeof(T)); plat_host_free(ctx, h); }
            h = nh; hcap = ncap;
        }
        if (dev && want > dcap) {
            const size_t ncap = want + want / 2 + 64;
            T* nd = nullptr;
            ck(plat_malloc(ctx, ncap * sizeof(T) + PLAT_BLOB_PAD, (void**)&nd), "plat_malloc");
            if (d) plat_free(ctx, d);                                      // (contents are rewritten by whoever grows a buffer)
            d = nd; dcap = ncap;
            if (zeroStream) ck(plat_memset(ctx, d, 0, ncap * sizeof(T) + PLAT_BLOB_PAD, zeroStream), "plat_memset");
        }
    }
    void release(plat_ctx* ctx) {
        if (!view) { if (h) plat_host_free(ctx, h); if (d) plat_free(ctx, d); }
        h = nullptr; d = nullptr; hcap = dcap = n = 0;
    }
};
typedef Staged<uint8_t> Arena;

struct Slot {
    plat_ctx* ctx = nullptr;
    void* stream = nullptr;
    bool countCells = false;                                               // plat_caller_count_cells: likelihood batches through the synchronous entry point
    int64_t nDpRef = 0, cellsRef = 0, nDpRun = 0, cellsRun = 0;            // ... and their plat_align_stats summed (this worker's share)
    int64_t nAlign = 0, alignHapBytes = 0, alignReadBytes = 0, alignReads = 0, alignDpBytes = 0;
    double secSeed = 0.0, secDp = 0.0, secSweep = 0.0, secPairs = 0.0;
    // chunk read table (device): bases, qualities, offsets, per-read fields, CIGARs; t_pack: the bytes of PLAT_READS_PACKED tables as
    // they crossed the link (expanded into t_seq / t_qual by plat_unpack_reads), t_exc*: their exceptions
    Staged<uint8_t> t_seq, t_qual, t_mapq, t_pack, t_excb, t_excq;
    Staged<int64_t> t_excidx;
    Staged<plat_table_desc> t_desc;
    Staged<plat_unpack_piece> t_pieces;
    Staged<int64_t> t_off;
    Staged<int32_t> t_pos, t_end, t_flags, t_cigoff, t_region;
    Staged<int16_t> t_cigar;
    // candidate scan
    Staged<uint8_t> c_ref;
    Staged<int64_t> c_refoff;
    Staged<int32_t> c_rss, c_clen, c_rec, c_cnt, c_status, c_scanbegin, c_scanlongest, m_cand, m_n;
    // window batch
    Staged<int32_t> w_hapbegin, w_readbegin, w_start, w_end, w_flank, w_segbegin, w_ngood, w_src, g_pos, g_end, g_flags, o_calls, o_iters, o_hapscore, o_score;
    Staged<int64_t> w_pairoff, w_hapoff, w_readoff, w_gloff;
    Staged<uint8_t> w_hapseq, w_kind, g_seq, g_qual, g_mapq;
    Staged<double> o_loglik, o_gl, o_logl, o_gof, o_freq, o_em;
    // posteriors / stats / calls
    Staged<int32_t> p_win, s_vw, s_pos, s_min, s_max, s_nadd, s_nrem, s_gb, s_ge, s_bb, s_be, s_ps, s_minq, s_nminq, k_win, k_nvar, k_vih, k_ref, k_ph;
    Staged<int64_t> p_off, s_aoff, s_moff, s_counts, k_vo, k_ro, k_lo;
    Staged<uint8_t> p_mask, s_added, s_vig;
    Staged<double> p_prior, p_post, k_lik, k_out4;
    // assembler tiles (assemble=1)
    Staged<uint8_t> as_ref, as_seq, as_qual, as_mapq, as_blob;
    Staged<int64_t> as_refoff, as_roff;
    Staged<int32_t> as_refstart, as_astart, as_aend, as_rbegin, as_src, as_pos, as_end, as_flags, as_cnt, as_status, as_vpos, as_nrem, as_nadd, as_off;
    // stage B on the device (plat_stage_b_batch): what it reads, what comes back, and the window batch it leaves on the device
    Staged<int32_t> sb_rstart, sb_rend, sb_rlen, sb_tabbegin, sb_tabn, sb_tablongest, sb_matepos;
    Staged<int32_t> sb_hdr, sb_vpos, sb_vnrem, sb_vnadd, sb_vsupp, sb_vbmin, sb_vbmax, sb_vrempos, sb_vaddoff, sb_wstart, sb_wend, sb_wvfirst, sb_wvn, sb_wflags,
                    sb_wptrs, sb_wnhaps, sb_wbatch;
    Staged<uint8_t> sb_added;
    Staged<uint32_t> sb_hapmask;
    Staged<int64_t> sb_totals;
    Staged<int32_t> d_hapbegin, d_readbegin, d_start, d_end, d_flank, d_segbegin, d_ngood, d_src, d_scratch;                 // device only
    Staged<int64_t> d_pairoff, d_gloff, d_hapoff, d_readoff;
    Staged<uint8_t> d_hapseq, d_kind;
    // many small arrays travel as ONE copy: they are views into these blocks (Layout)
    Arena a_tab, a_desc, a_cin, a_cout, a_mout, a_win, a_wout, a_pin, a_sin, a_sout, a_asin, a_asout, a_bin, a_bout;
    double t_host = 0, t_wait = 0;

    void sync(const char* where) {
        const auto t0 = Clock::now();
        const int rc = plat_stream_sync(ctx, stream);
        t_wait += secs(t0, Clock::now());
        ck(rc, where);
    }
    template <class T> void up(Staged<T>& s, size_t n) { if (n) ck(plat_memcpy_h2d(ctx, s.d, s.h, n * sizeof(T), stream), "plat_memcpy_h2d"); }
    template <class T> void down(Staged<T>& s, size_t n) { if (n) ck(plat_memcpy_d2h(ctx, s.h, s.d, n * sizeof(T), stream), "plat_memcpy_d2h"); }
};

// Arrays of one stage laid out back to back in one pinned block + one device block: one copy per stage and direction instead of one per
// array (a copy costs ~5 us of GPU time and as much host time however small it is).
struct Layout {
    struct Item { void** h; void** d; size_t bytes, off; };
    std::vector<Item> items;
    size_t total = 0;
    template <class T> void add(Staged<T>& st, size_t n) {
        st.view = true; st.n = n;
        items.push_back(Item{(void**)&st.h, (void**)&st.d, (n + 8) * sizeof(T), 0});
    }
    void commit(Slot& s, Arena& a) {
        total = 0;
        for (Item& it : items) { it.off = total; total += (it.bytes + 255) & ~(size_t)255; }
        a.reserve(s.ctx, total + PLAT_BLOB_PAD);
        for (Item& it : items) { *it.h = a.h + it.off; *it.d = a.d + it.off; }
    }
    void upload(Slot& s, Arena& a) { if (total) ck(plat_memcpy_h2d(s.ctx, a.d, a.h, total, s.stream), "plat_memcpy_h2d"); }
    void uploadFirst(Slot& s, Arena& a, size_t nItems) {
        const size_t bytes = nItems >= items.size() ? total : items[nItems].off;
        if (bytes) ck(plat_memcpy_h2d(s.ctx, a.d, a.h, bytes, s.stream), "plat_memcpy_h2d");
    }
    // only the first `nItems` arrays (they lie in the order they were added)
    void downloadFirst(Slot& s, Arena& a, size_t nItems) {
        const size_t bytes = nItems >= items.size() ? total : items[nItems].off;
        if (bytes) ck(plat_memcpy_d2h(s.ctx, a.h, a.d, bytes, s.stream), "plat_memcpy_d2h");
    }
    void download(Slot& s, Arena& a) { if (total) ck(plat_memcpy_d2h(s.ctx, a.h, a.d, total, s.stream), "plat_memcpy_d2h"); }
};

// ---- a read table of the caller as the region loop sees it (ReadArray, cwindow.pyx:92-236) --------------------------------------
struct TableView {
    const plat_read_table* t = nullptr;
    int64_t base = 0;                                                     // index of its first read in the chunk's device table
    int64_t blobBase = 0;                                                 // first byte of its bases in the chunk's blob
    int longest = 0;                                                      // getLengthOfLongestRead (:167-172)
    int maxLen = 0;                                                       // most bases of a read
    int n() const { return t->n_reads; }
    static int lowerBound(const int32_t* a, int n, int64_t key) { return (int)(std::lower_bound(a, a + n, key, [](int32_t x, int64_t k) { return (int64_t)x < k; }) - a); }
    // the same index, found by galloping away from `hint` (the loop's windows ascend: the last window's pointer is a few reads away)
    static int lowerBoundNear(const int32_t* a, int n, int64_t key, int hint) {
        int lo, hi;                                                       // answer in [lo, hi]
        hint = std::min(std::max(hint, 0), n);
        if (hint < n && (int64_t)a[hint] < key) {
            int step = 1; lo = hint + 1;
            while (lo + step <= n && lo + step - 1 < n && (int64_t)a[lo + step - 1] < key) { lo += step; step <<= 1; }
            hi = std::min(n, lo + step - 1);
        } else {
            int step = 1; hi = hint;
            while (hi - step >= 0 && (int64_t)a[hi - step] >= key) { hi -= step; step <<= 1; }
            lo = std::max(0, hi - step + 1);
        }
        return lo + lowerBound(a + lo, hi - lo, key);
    }
    // shared body of countReadsCoveringRegion (:176-206) and setWindowPointers (:208-234)
    void overlapRange(int start, int end, int& s, int& e, int hintS = -1, int hintE = -1) const {
        const int N = n();
        if (N == 0) { s = e = 0; return; }
        const int64_t keyS = std::max<int64_t>(1, (int64_t)start - longest);
        s = hintS >= 0 ? lowerBoundNear(t->pos, N, keyS, hintS) : lowerBound(t->pos, N, keyS);
        e = hintE >= 0 ? lowerBoundNear(t->pos, N, end, hintE) : lowerBound(t->pos, N, end);
        while (s < N && t->end[s] <= start) ++s;
        if (s > e) throw WindowError("This should never happen. Read start pointer > read end pointer!!");
        e = std::min(e, N);
    }
    void matePosRange(int start, int end, int& s, int& e) const {         // setWindowPointersBasedOnMatePos (:236-264)
        const int N = n();
        if (N == 0) { s = e = 0; return; }
        s = lowerBound(t->mate_pos, N, std::max<int64_t>(1, (int64_t)start - longest));
        e = lowerBound(t->mate_pos, N, end);
        if (s > e) throw WindowError("This should never happen. Read start pointer > read end pointer!!");
        e = std::min(e, N);
    }
    int rlen(int i) const { return (int)(t->off[i + 1] - t->off[i]); }
};
struct SampleView { TableView reads, bad, broken; };

// `n` bases of a read table from byte `at` of its blob, as letters (the host only ever needs the few inserted bases of candidates)
static std::string tableBases(const plat_read_table& t, int64_t at, int n) {
    std::string out((size_t)std::max(n, 0), 'A');
    if (n <= 0) return out;
    if (t.encoding != PLAT_READS_PACKED) { memcpy(&out[0], t.seq + at, (size_t)n); return out; }
    for (int i = 0; i < n; ++i) out[(size_t)i] = "ACTG"[t.seq[at + i] & 3];
    if (t.n_exceptions > 0) {
        const int64_t* e = std::lower_bound(t.exc_index, t.exc_index + t.n_exceptions, at);
        for (; e < t.exc_index + t.n_exceptions && *e < at + n; ++e) out[(size_t)(*e - at)] = (char)t.exc_base[e - t.exc_index];
    }
    return out;
}

static int longestRead(const plat_read_table& t) {
    int m = 0;
    for (int i = 0; i < t.n_reads; ++i) m = std::max(m, t.end[i] - t.pos[i]);
    return m;
}
static int mostBases(const plat_read_table& t) {
    int64_t m = 0;
    for (int i = 0; i < t.n_reads; ++i) m = std::max(m, t.off[i + 1] - t.off[i]);
    return (int)m;
}

// ---- haplotypes ---------------------------------------------------------------------------------------------------------------
struct Hap {
    VarList variants;
    std::string seq;
};

// chaplotype.pyx:127-191 + getMutatedSequence :397-449.  startPos / endPos already clamped as the constructor does.
static std::string haplotypeSequence(const Fasta& fa, int startPos, int endPos, int endBuf, const VarList& variants) {
    if (variants.empty()) return fa.getSequence((int64_t)startPos - endBuf, (int64_t)endPos + endBuf);
    std::string out;
    size_t extra = 0;
    for (const Variant* v : variants) extra += v->added.size();
    out.reserve((size_t)std::max(0, endPos - startPos) + 2 * (size_t)endBuf + extra + 16);
    fa.appendSequence(out, (int64_t)startPos - endBuf, startPos);
    int cur = startPos;
    const Variant* first = variants[0];
    if (first->refPos != cur) { fa.appendSequence(out, cur, first->refPos); cur = first->refPos; }
    for (const Variant* v : variants) {
        if (v->refPos > cur) { fa.appendSequence(out, cur, v->refPos); cur = v->refPos; }
        if (v->nAdded == v->nRemoved) { out += v->added; cur += v->nRemoved; }
        else {
            if (v->added.empty() || v->removed.empty()) {
                if (v->refPos == cur) { out += fa.getCharacter(v->refPos); cur += 1; }
            }
            cur += v->nRemoved;
            out += v->added;
        }
    }
    if (cur < endPos) fa.appendSequence(out, cur, endPos);
    fa.appendSequence(out, endPos, (int64_t)endPos + endBuf);
    return out;
}

// Python tuple comparison of (score, variants) as the heap of getFilteredHaplotypes orders them
struct ScoredHap { double score; VarList vs; };
static bool scoredLess(const ScoredHap& a, const ScoredHap& b) {
    if (a.score != b.score) return a.score < b.score;
    const size_t n = std::min(a.vs.size(), b.vs.size());
    for (size_t i = 0; i < n; ++i) {
        if (a.vs[i] == b.vs[i] || a.vs[i]->same(*b.vs[i])) continue;
        return variantLess(a.vs[i], b.vs[i]);
    }
    return a.vs.size() < b.vs.size();
}
// heapq (CPython): _siftdown / _siftup / heappush / heappushpop
static void heapSiftDown(std::vector<ScoredHap>& h, size_t startpos, size_t pos) {
    ScoredHap item = h[pos];
    while (pos > startpos) {
        const size_t parentpos = (pos - 1) >> 1;
        if (scoredLess(item, h[parentpos])) { h[pos] = h[parentpos]; pos = parentpos; continue; }
        break;
    }
    h[pos] = item;
}
static void heapSiftUp(std::vector<ScoredHap>& h, size_t pos) {
    const size_t endpos = h.size(), startpos = pos;
    ScoredHap item = h[pos];
    size_t childpos = 2 * pos + 1;
    while (childpos < endpos) {
        const size_t rightpos = childpos + 1;
        if (rightpos < endpos && !scoredLess(h[childpos], h[rightpos])) childpos = rightpos;
        h[pos] = h[childpos];
        pos = childpos;
        childpos = 2 * pos + 1;
    }
    h[pos] = item;
    heapSiftDown(h, startpos, pos);
}
static void heapPush(std::vector<ScoredHap>& h, const ScoredHap& item) { h.push_back(item); heapSiftDown(h, 0, h.size() - 1); }
static void heapPushPop(std::vector<ScoredHap>& h, ScoredHap item) {
    if (!h.empty() && scoredLess(h[0], item)) { std::swap(item, h[0]); heapSiftUp(h, 0); }
}

// ---- per-window and per-region working state ---------------------------------------------------------------------------------------
struct Ptrs { int gs, ge, bs, be, ks, ke; };
typedef SmallVec<Ptrs, 2> PtrList;                                        // one per sample                               // window pointers of one sample: reads, badReads, brokenMates

struct WindowWork {
    int region = 0, startPos = 0, endPos = 0;
    VarList vars;                                                          // window["variants"] (after filterVariantsByCoverage)
    VarList allVars;                                                       // the unfiltered list callVariantsInWindow keeps as `variants`
    PtrList ptrs;
    int nReads = 0;
    int hapStart = 0, hapEnd = 0, endBuf = 0;                              // Haplotype.startPos / endPos / endBufferSize
    std::string refSeq;                                                    // reference haplotype
    std::vector<Hap> haps;                                                 // merged, sorted (Population.haplotypes)
    bool live = false;                                                     // goes to the device
    // greedy filter state
    bool greedy = false;
    VarList byCoverage;
    size_t step = 0;
    std::vector<ScoredHap> heap;
    std::vector<VarList> cands;
    std::vector<int> sampledSeg;                                           // per sample: [begin, end) into `sampled`
    std::vector<std::pair<int, int>> sampled;                              // (sample, local index in reads table)
    // results
    int bw = -1;                                                           // window index in the device batch
    int hapBegin = 0;                                                      // index of its first haplotype there
    bool onDevice = false;                                                 // prepared by plat_stage_b_batch: its batch entries are on the device already
    VarList distinct;                                                      // _distinctVariants
    std::vector<double> posterior;                                         // aligned with distinct
    VarList called;                                                        // variantPosteriors keys, in insertion order
    std::vector<double> calledPost;
    std::vector<std::pair<int, VarList>> byPos;                            // varsByPos, insertion order
    std::vector<VarInfo> info;                                             // vcfInfo in getHaplotypeInfo order
    int firstStatVar = 0, firstSite = 0;
    bool failed = false;                                                   // raised while it was prepared: logged and skipped, no line of any kind
    std::string text;                                                      // its record lines (and, with outputRefCalls, the REFCALL lines that belong to it)
    int64_t nRecords = 0, nRefRecords = 0;
    int firstFlat = -1;                                                    // outputRefCalls: index of its first flat-prior posterior (one per variant of `vars`)
};

struct VariantPool {
    std::deque<Variant> store;
    Variant* make(int pos, const std::string& rem, const std::string& add, int nSupp, int source) {
        store.emplace_back(pos, rem, add, nSupp, source);
        return &store.back();
    }
};

// what the region loop writes, in the order it writes it: calling windows and (outputRefCalls=1) reference-call blocks
struct Item { int kind; int window; std::string text; int64_t nRef = 0; };     // kind 0: windows[window]; 1: a block whose line is already in `text`

struct RegionWork {
    const plat_region* in = nullptr;
    int index = 0;
    std::vector<Item> items;
    PtrList cur;                                                           // the samples' window pointers as the loop last left them (a REFCALL line's NR)
    VarList asmVariants;                                                   // assembler candidates, tile after tile (variantcaller.pyx:496-519)
    Fasta fa;
    int rlen = 0;
    std::vector<SampleView> samples;
    VariantPool pool;
    VarList variants;
    std::vector<WindowWork> windows;
    std::string text;
    int64_t nCandRecords = 0;
    // frees everything but the record text (called by the worker that finished the region, so that the cost of freeing thousands of
    // windows and haplotypes is spread over the workers instead of being paid serially at the end of plat_call_regions)
    void release() {
        std::vector<WindowWork>().swap(windows);
        std::vector<Item>().swap(items);
        VarList().swap(variants);
        VarList().swap(asmVariants);
        pool = VariantPool();
    }
};

struct Options : plat_caller_options {};

static void logWindowFailure(const char* chrom, int s, int e, const char* what) {
    fprintf(stderr, "platypus caller: problem calling variants in window %s:%d-%d, skipping it: %s\n", chrom, s, e, what);
}

// ---- the device window batch ---------------------------------------------------------------------------------------------------------
struct BatchBuilder {
    int nInd = 0;
    std::vector<int32_t> hapbegin{0}, readbegin{0}, start, end, flank, segbegin{0}, ngood, src;
    std::vector<int64_t> pairoff{0}, hapoff{0}, readoff{0}, gloff{0};
    std::vector<uint8_t> kind;
    std::string hapseq;
    int maxHap = 0, maxRead = 0, maxR = 0, maxH = 0;
    void beginWindow(int s, int e, int fl) { start.push_back(s); end.push_back(e); flank.push_back(fl); }
    void addHap(const std::string& seq) {
        hapseq += seq;
        hapoff.push_back((int64_t)hapseq.size());
        maxHap = std::max(maxHap, (int)seq.size());
    }
    void addRead(const TableView& tv, int i, int k) {
        src.push_back((int32_t)(tv.base + i));
        kind.push_back((uint8_t)k);
        const int L = tv.rlen(i);
        readoff.push_back(readoff.back() + L);
        maxRead = std::max(maxRead, L);
    }
    void addReads(const TableView& tv, int i0, int i1, int k) {          // reads [i0, i1) of one table: the same as addRead one by one
        if (i1 <= i0) return;
        const size_t n = (size_t)(i1 - i0), at = src.size();
        src.resize(at + n); kind.resize(at + n, (uint8_t)k); readoff.resize(at + n + 1);
        int32_t* sp = src.data() + at;
        int64_t* rp = readoff.data() + at;                                 // rp[0] = the running end so far
        const int64_t* off = tv.t->off;
        const int32_t base = (int32_t)tv.base;
        int64_t run = rp[0];
        int longest = maxRead;
        for (size_t j = 0; j < n; ++j) {
            const int i = i0 + (int)j;
            const int L = (int)(off[i + 1] - off[i]);
            sp[j] = base + i;
            run += L; rp[j + 1] = run;
            longest = std::max(longest, L);
        }
        maxRead = longest;
    }
    void endSegment(int nGood) { segbegin.push_back((int32_t)src.size()); ngood.push_back(nGood); }
    void endWindow() {
        const int H = (int)hapoff.size() - 1 - hapbegin.back(), R = (int)src.size() - readbegin.back();
        hapbegin.push_back((int32_t)hapoff.size() - 1);
        readbegin.push_back((int32_t)src.size());
        pairoff.push_back(pairoff.back() + (int64_t)H * R);
        gloff.push_back(gloff.back() + (int64_t)(H * (H + 1) / 2) * nInd);
        maxR = std::max(maxR, R); maxH = std::max(maxH, H);
    }
    // back to empty with the memory kept: the arrays of a chunk are megabytes, and growing them from nothing for every chunk is a
    // chain of mmap + page faults + copies
    void reset(int nInd_) {
        nInd = nInd_;
        hapbegin.assign(1, 0); readbegin.assign(1, 0); start.clear(); end.clear(); flank.clear(); segbegin.assign(1, 0); ngood.clear(); src.clear();
        pairoff.assign(1, 0); hapoff.assign(1, 0); readoff.assign(1, 0); gloff.assign(1, 0);
        kind.clear(); hapseq.clear();
        maxHap = maxRead = maxR = maxH = 0;
    }
    int nWindows() const { return (int)start.size(); }
    int nHaps() const { return (int)hapoff.size() - 1; }
    int nReads() const { return (int)src.size(); }
};

template <class T, class V> static void fill(Slot& s, Staged<T>& st, const V& v, bool dev = true) {
    if (!st.view) st.reserve(s.ctx, v.size() + 1, true, dev);
    for (size_t i = 0; i < v.size(); ++i) st.h[i] = (T)v[i];
    st.n = v.size();
}

struct DeviceBatch {                                                       // what stays valid on the device after runWindows
    plat_window_batch wb;
    int nWindows = 0, nHaps = 0, nReads = 0, nInd = 0, maxH = 0;
    int64_t nPairs = 0, nGl = 0;
    // per window: first haplotype, first genotype likelihood, good reads, first read of the (one) segment -- device arrays
    const int32_t* hapbegin = nullptr; const int64_t* gloff = nullptr; const int32_t* ngood = nullptr; const int32_t* segbegin = nullptr;
    const int32_t* src = nullptr; const int64_t* readoff = nullptr;
    int maxHap = 0, maxRead = 0, maxR = 0;
    int64_t hapBlob = 0, readBlob = 0;
};

// The likelihood part of a window batch whose arrays are on the device (db.wb all but the gathered reads): gather its reads from the chunk
// table and run Haplotype.alignReads for all of it; full = also Population.setup, HapScore and EM.  Results are copied to the pinned host
// mirrors; waits for them.
static void runBatch(Slot& s, DeviceBatch& db, const Options& o, bool full, bool wantLoglik) {
    const size_t blob = (size_t)db.readBlob;
    s.g_seq.reserve(s.ctx, blob + PLAT_BLOB_PAD, false, true, s.stream); s.g_qual.reserve(s.ctx, blob + PLAT_BLOB_PAD, false, true, s.stream);
    const size_t nR = (size_t)db.nReads;
    s.g_pos.reserve(s.ctx, nR + 1, false); s.g_end.reserve(s.ctx, nR + 1, false); s.g_flags.reserve(s.ctx, nR + 1, false); s.g_mapq.reserve(s.ctx, nR + 1, false);
    ck(plat_gather_reads(s.ctx, (int64_t)nR, db.src, db.readoff, s.t_seq.d, s.t_qual.d, s.t_off.d, s.t_pos.d, s.t_end.d, s.t_mapq.d,
                         s.t_flags.d, s.g_seq.d, s.g_qual.d, s.g_pos.d, s.g_end.d, s.g_mapq.d, s.g_flags.d, s.stream), "plat_gather_reads");
    plat_window_batch& wb = db.wb;
    wb.n_windows = db.nWindows; wb.n_haps = db.nHaps; wb.n_reads = db.nReads;
    wb.read_seq = s.g_seq.d; wb.read_qual = s.g_qual.d; wb.read_pos = s.g_pos.d; wb.read_end = s.g_end.d;
    wb.read_mapq = s.g_mapq.d; wb.read_flags = s.g_flags.d;
    s.o_loglik.reserve(s.ctx, (size_t)db.nPairs + 1, wantLoglik);
    plat_batch_hints h;
    memset(&h, 0, sizeof h);
    h.max_hap_len = db.maxHap; h.max_read_len = db.maxRead; h.max_reads_per_window = db.maxR;
    h.n_pairs = db.nPairs; h.hap_blob_len = db.hapBlob; h.read_blob_len = (int64_t)blob; h.extra_jobs_cap = 0;
    if (s.countCells) {
        plat_align_stats as;
        memset(&as, 0, sizeof as);
        ck(plat_profile_enable(s.ctx, 1), "plat_profile_enable");
        ck(plat_align_window_batch(s.ctx, &wb, o.calculateFlankScore ? 1 : 0, 0, s.o_loglik.d, nullptr, &as, s.stream), "plat_align_window_batch");
        s.nDpRef += as.n_dp_reference; s.cellsRef += as.cells_reference; s.nDpRun += as.n_dp_launched; s.cellsRun += as.cells_launched;
        plat_profile pf;
        memset(&pf, 0, sizeof pf);
        ck(plat_profile_last(s.ctx, &pf), "plat_profile_last");
        ck(plat_profile_enable(s.ctx, 0), "plat_profile_enable");
        s.nAlign += 1; s.alignHapBytes += db.hapBlob; s.alignReadBytes += (int64_t)blob; s.alignReads += db.nReads;
        s.alignDpBytes += pf.dp_alg_bytes; s.secSeed += 1e-3 * pf.ms_seed_kernel; s.secDp += 1e-3 * pf.ms_dp;
        s.secSweep += 1e-3 * pf.ms_sweep; s.secPairs += 1e-3 * pf.ms_pairs;
    } else
        ck(plat_align_window_batch_async(s.ctx, &wb, &h, o.calculateFlankScore ? 1 : 0, 0, s.o_loglik.d, nullptr, s.stream), "plat_align_window_batch_async");
    if (wantLoglik) s.down(s.o_loglik, (size_t)db.nPairs);
    if (full) {
        const size_t nG = (size_t)db.nGl + 1;
        s.o_gl.reserve(s.ctx, nG, false); s.o_logl.reserve(s.ctx, nG, false); s.o_gof.reserve(s.ctx, nG, false); s.o_em.reserve(s.ctx, nG, false);
        Layout LO;
        LO.add(s.o_freq, (size_t)db.nHaps); LO.add(s.o_calls, (size_t)db.nWindows * db.nInd); LO.add(s.o_hapscore, (size_t)db.nWindows);
        LO.commit(s, s.a_wout);
        s.o_iters.reserve(s.ctx, (size_t)db.nWindows + 1, false);
        ck(plat_genotype_window_batch(s.ctx, &wb, db.nInd, db.segbegin, db.ngood, s.o_loglik.d, db.gloff, s.o_gl.d, s.o_logl.d, s.o_gof.d,
                                      s.stream), "plat_genotype_window_batch");
        ck(plat_haplotype_score_batch(s.ctx, &wb, db.nInd, db.maxH, db.segbegin, db.ngood, s.o_loglik.d, nullptr, s.o_hapscore.d, s.stream),
           "plat_haplotype_score_batch");
        ck(plat_em_window_batch(s.ctx, db.nWindows, db.nInd, db.maxH, db.hapbegin, db.gloff, db.ngood, s.o_gl.d, 100, o.useEMLikelihoods,
                                s.o_freq.d, s.o_em.d, s.o_calls.d, s.o_iters.d, s.stream), "plat_em_window_batch");
        LO.download(s, s.a_wout);
    }
    s.sync("window batch");
}

// Upload a BatchBuilder and run it (runBatch)
static DeviceBatch runWindows(Slot& s, const BatchBuilder& b, const Options& o, bool full, bool wantLoglik) {
    DeviceBatch db;
    db.nWindows = b.nWindows(); db.nHaps = b.nHaps(); db.nReads = b.nReads(); db.nInd = b.nInd; db.maxH = b.maxH;
    db.nPairs = b.pairoff.back(); db.nGl = b.gloff.back();
    if (db.nWindows == 0) return db;
    {
        Layout L;
        L.add(s.w_hapbegin, b.hapbegin.size()); L.add(s.w_readbegin, b.readbegin.size()); L.add(s.w_start, b.start.size()); L.add(s.w_end, b.end.size());
        L.add(s.w_flank, b.flank.size()); L.add(s.w_pairoff, b.pairoff.size()); L.add(s.w_hapoff, b.hapoff.size()); L.add(s.w_readoff, b.readoff.size());
        L.add(s.w_gloff, b.gloff.size()); L.add(s.w_segbegin, b.segbegin.size()); L.add(s.w_ngood, b.ngood.size()); L.add(s.w_src, b.src.size());
        L.add(s.w_kind, b.kind.size()); L.add(s.w_hapseq, b.hapseq.size() + PLAT_BLOB_PAD);
        L.commit(s, s.a_win);
        fill(s, s.w_hapbegin, b.hapbegin); fill(s, s.w_readbegin, b.readbegin); fill(s, s.w_start, b.start); fill(s, s.w_end, b.end);
        fill(s, s.w_flank, b.flank); fill(s, s.w_pairoff, b.pairoff); fill(s, s.w_hapoff, b.hapoff); fill(s, s.w_readoff, b.readoff);
        fill(s, s.w_gloff, b.gloff); fill(s, s.w_segbegin, b.segbegin); fill(s, s.w_ngood, b.ngood); fill(s, s.w_src, b.src); fill(s, s.w_kind, b.kind);
        memcpy(s.w_hapseq.h, b.hapseq.data(), b.hapseq.size());
        memset(s.w_hapseq.h + b.hapseq.size(), 0, PLAT_BLOB_PAD);
        L.upload(s, s.a_win);
    }
    plat_window_batch& wb = db.wb;
    memset(&wb, 0, sizeof wb);
    wb.win_hap_begin = s.w_hapbegin.d; wb.win_read_begin = s.w_readbegin.d; wb.win_start = s.w_start.d; wb.win_end = s.w_end.d;
    wb.win_flank = s.w_flank.d; wb.pair_off = s.w_pairoff.d; wb.hap_seq = s.w_hapseq.d; wb.hap_off = s.w_hapoff.d;
    wb.read_off = s.w_readoff.d; wb.read_kind = s.w_kind.d;
    db.hapbegin = s.w_hapbegin.d; db.gloff = s.w_gloff.d; db.ngood = s.w_ngood.d; db.segbegin = s.w_segbegin.d; db.src = s.w_src.d; db.readoff = s.w_readoff.d;
    db.maxHap = b.maxHap; db.maxRead = b.maxRead; db.maxR = b.maxR; db.hapBlob = (int64_t)b.hapseq.size(); db.readBlob = b.readoff.back();
    runBatch(s, db, o, full, wantLoglik);
    return db;
}

// ---- the chunk pipeline ----------------------------------------------------------------------------------------------------------------
struct Chunk {
    Slot& s;
    const Options& o;
    int nInd;
    const char* const* names;
    std::vector<RegionWork*> regions;
    plat_caller_stats& st;
    std::mutex& stMutex;

    // -- A: one device table for every read of the chunk; layout: all `reads` of every (region, sample), then all badReads, then all brokenMates
    void uploadReads() {
        size_t nReads[3] = {0, 0, 0}, nBytes[3] = {0, 0, 0}, nCig[3] = {0, 0, 0}, nExc = 0;
        bool anyPacked = false;
        for (RegionWork* r : regions)
            for (SampleView& sv : r->samples) {
                TableView* tv[3] = {&sv.reads, &sv.bad, &sv.broken};
                for (int k = 0; k < 3; ++k) {
                    const plat_read_table& t = *tv[k]->t;
                    nReads[k] += (size_t)t.n_reads;
                    nBytes[k] += (size_t)t.off[t.n_reads];
                    nCig[k] += (size_t)t.cig_off[t.n_reads];
                    if (t.encoding == PLAT_READS_PACKED) { anyPacked = true; nExc += (size_t)std::max<int64_t>(t.n_exceptions, 0); }
                    else if (t.encoding != PLAT_READS_ASCII) throw DeviceError(PLAT_ERR_INVALID, "plat_read_table.encoding");
                }
            }
        // (tables lie back to back in the chunk blob: read i's bytes are [t_off[i], t_off[i + 1]) for every consumer.  A packed table that is
        //  resident on the device is expanded straight from there; plat_unpack_reads reads a source of another misalignment with unaligned loads)
        const size_t N = nReads[0] + nReads[1] + nReads[2], B = nBytes[0] + nBytes[1] + nBytes[2], Cg = nCig[0] + nCig[1] + nCig[2];
        if (N > 0x7FFFFFF0ull) throw DeviceError(PLAT_ERR_OVERFLOW, "chunk read table");
        Slot& z = s;
        z.t_seq.reserve(z.ctx, B + PLAT_BLOB_PAD, false, true, z.stream); z.t_qual.reserve(z.ctx, B + PLAT_BLOB_PAD, false, true, z.stream);
        if (anyPacked) z.t_pack.reserve(z.ctx, B + PLAT_BLOB_PAD, false, true, z.stream);
        // every table with its per-read arrays on the device already: the chunk table is put together there (plat_concat_read_tables)
        bool cols = true;
        for (RegionWork* r : regions)
            for (SampleView& sv : r->samples)
                for (const TableView* tv : {&sv.reads, &sv.bad, &sv.broken}) {
                    const plat_read_table& t = *tv->t;
                    if (t.n_reads && !(t.dev_off && t.dev_pos && t.dev_end && t.dev_mapq && t.dev_flags && t.dev_cigar && t.dev_cig_off && t.dev_seq)) cols = false;
                }
        Layout L;
        L.add(z.t_excidx, nExc + 1); L.add(z.t_excb, nExc + 1); L.add(z.t_excq, nExc + 1);
        L.add(z.t_off, N + 1); L.add(z.t_pos, N + 1); L.add(z.t_end, N + 1); L.add(z.t_flags, N + 1); L.add(z.t_mapq, N + 1); L.add(z.t_cigoff, N + 1);
        L.add(z.t_cigar, 2 * Cg + 2); L.add(z.t_region, nReads[0] + 1);
        L.commit(z, z.a_tab);
        Layout LD;
        size_t nDesc = 0;
        int mostPerTable = 0;
        const size_t nTables = 3 * regions.size() * (regions.empty() ? 0 : regions[0]->samples.size());
        LD.add(z.t_pieces, nTables + 1);
        if (cols) LD.add(z.t_desc, nTables + 1);
        LD.commit(z, z.a_desc);
        struct Pending { size_t bo, nb, e0, ne; const uint8_t* dev; };       // dev: expand from this device address instead of t_pack + bo
        std::vector<Pending> packed;
        size_t ri = 0, bo = 0, co = 0, eo = 0, inBytes = 0;
        int scan = 0;
        for (int k = 0; k < 3; ++k) {
            scan = 0;
            for (RegionWork* r : regions)
                for (SampleView& sv : r->samples) {
                    TableView& tv = k == 0 ? sv.reads : (k == 1 ? sv.bad : sv.broken);
                    const plat_read_table& t = *tv.t;
                    const int n = t.n_reads;
                    tv.base = (int64_t)ri; tv.blobBase = (int64_t)bo;
                    maxReadLen = std::max(maxReadLen, tv.maxLen);
                    const size_t nb = (size_t)t.off[n], nc = (size_t)t.cig_off[n];
                    if (nb && t.encoding == PLAT_READS_PACKED) {            // one byte per base crosses the link (or none: dev_seq); expanded below
                        if (!t.dev_seq) ck(plat_memcpy_h2d(z.ctx, z.t_pack.d + bo, t.seq, nb, z.stream), "plat_memcpy_h2d(packed)");
                        const size_t ne = (size_t)std::max<int64_t>(t.n_exceptions, 0);
                        // every packed table of the chunk is expanded by ONE launch (plat_unpack_reads_pieces): tables that follow each other in
                        // t_pack join into one piece; exceptions are indexed from the chunk blob's first byte
                        const bool joins = !t.dev_seq && !packed.empty() && !packed.back().dev && packed.back().bo + packed.back().nb == bo;
                        for (size_t e = 0; e < ne; ++e) { z.t_excidx.h[eo + e] = t.exc_index[e] + (int64_t)bo; z.t_excb.h[eo + e] = t.exc_base[e]; z.t_excq.h[eo + e] = t.exc_qual[e]; }
                        if (joins) { packed.back().nb += nb; packed.back().ne += ne; }
                        else packed.push_back(Pending{bo, nb, eo, ne, t.dev_seq});
                        eo += ne; inBytes += (t.dev_seq ? 0 : nb) + 10 * ne;
                    } else if (nb && t.dev_seq && t.dev_qual) {            // resident in HBM already
                        ck(plat_memcpy_d2d(z.ctx, z.t_seq.d + bo, t.dev_seq, nb, z.stream), "plat_memcpy_d2d(seq)");
                        ck(plat_memcpy_d2d(z.ctx, z.t_qual.d + bo, t.dev_qual, nb, z.stream), "plat_memcpy_d2d(qual)");
                    } else if (nb) {                                       // bases and qualities go straight from the caller's memory
                        ck(plat_memcpy_h2d(z.ctx, z.t_seq.d + bo, t.seq, nb, z.stream), "plat_memcpy_h2d(seq)");
                        ck(plat_memcpy_h2d(z.ctx, z.t_qual.d + bo, t.qual, nb, z.stream), "plat_memcpy_h2d(qual)");
                        inBytes += 2 * nb;
                    }
                    if (cols) {
                        if (n) {
                            plat_table_desc& d = z.t_desc.h[nDesc++];
                            d.off = t.dev_off; d.pos = t.dev_pos; d.end = t.dev_end; d.mapq = t.dev_mapq; d.flags = t.dev_flags; d.cigar = t.dev_cigar; d.cig_off = t.dev_cig_off;
                            d.n = n; d.scan = k == 0 ? scan : -1; d.first_read = (int64_t)ri; d.first_byte = (int64_t)bo; d.first_pair = (int64_t)co;
                            mostPerTable = std::max(mostPerTable, n);
                        }
                        ri += (size_t)n; bo += nb; co += nc;
                        ++scan;
                        continue;
                    }
                    for (int i = 0; i < n; ++i) {
                        z.t_off.h[ri + i] = (int64_t)bo + t.off[i];
                        z.t_cigoff.h[ri + i] = (int32_t)(co + (size_t)t.cig_off[i]);
                    }
                    if (n) {
                        memcpy(z.t_pos.h + ri, t.pos, sizeof(int32_t) * (size_t)n); memcpy(z.t_end.h + ri, t.end, sizeof(int32_t) * (size_t)n);
                        memcpy(z.t_flags.h + ri, t.flags, sizeof(int32_t) * (size_t)n); memcpy(z.t_mapq.h + ri, t.mapq, (size_t)n);
                        if (nc) memcpy(z.t_cigar.h + 2 * co, t.cigar, sizeof(int16_t) * 2 * nc);
                        if (k == 0) for (int i = 0; i < n; ++i) z.t_region.h[ri + i] = scan;
                    }
                    ri += (size_t)n; bo += nb; co += nc;
                    ++scan;
                }
        }
        if (cols) {
            L.uploadFirst(z, z.a_tab, 3);                                   // (the exceptions of packed tables; the per-read arrays are made on the device)
            if (nDesc) {
                LD.upload(z, z.a_desc);
                ck(plat_concat_read_tables(z.ctx, (int)nDesc, mostPerTable, z.t_desc.d, z.t_off.d, z.t_pos.d, z.t_end.d, z.t_mapq.d, z.t_flags.d, z.t_cigoff.d, z.t_cigar.d,
                                           z.t_region.d, (int64_t)N, (int64_t)bo, (int64_t)Cg, z.stream), "plat_concat_read_tables");
            }
        }
        if (!cols || !nDesc) {
            z.t_off.h[N] = (int64_t)bo; z.t_cigoff.h[N] = (int32_t)Cg;
            z.t_cigar.h[2 * Cg] = 0; z.t_cigar.h[2 * Cg + 1] = 0;
            L.upload(z, z.a_tab);
        }
        if (!packed.empty()) {
            size_t most = 0;
            for (size_t q = 0; q < packed.size(); ++q) {
                const Pending& p = packed[q];
                z.t_pieces.h[q] = plat_unpack_piece{p.dev ? p.dev : z.t_pack.d + p.bo, (int64_t)p.bo, (int64_t)p.nb};
                most = std::max(most, p.nb);
            }
            ck(plat_memcpy_h2d(z.ctx, z.t_pieces.d, z.t_pieces.h, packed.size() * sizeof(plat_unpack_piece), z.stream), "plat_memcpy_h2d(pieces)");
            ck(plat_unpack_reads_pieces(z.ctx, (int)packed.size(), (int64_t)most, z.t_pieces.d, z.t_seq.d, z.t_qual.d, (int64_t)bo, (int64_t)eo, z.t_excidx.d, z.t_excb.d,
                                        z.t_excq.d, z.stream), "plat_unpack_reads_pieces");
        }
        nGood = nReads[0]; nScan = scan; nBad = nReads[1]; nBroken = nReads[2];
        std::lock_guard<std::mutex> g(stMutex);
        st.n_reads += (int64_t)N;
        st.input_bytes += (int64_t)inBytes;
    }
    size_t nGood = 0, nBad = 0, nBroken = 0;
    int nScan = 0, maxReadLen = 0;
    int maxPerRead = 8;

    // -- A2: VariantCandidateGenerator.addCandidatesFromReads over the `reads` of every (region, sample) (variant.pyx:459-751)
    void scanCandidates() {
        Slot& z = s;
        std::vector<int64_t> refoff{0};
        std::vector<int32_t> rss, clen, scanbegin, scanlongest;
        std::string blob;
        for (RegionWork* r : regions)
            for (size_t i = 0; i < r->samples.size(); ++i) {
                scanbegin.push_back((int32_t)r->samples[i].reads.base); scanlongest.push_back(r->samples[i].reads.longest);
                const int64_t a = std::max<int64_t>(0, (int64_t)r->in->start - 2000);                   // variant.pyx:486-488
                const int64_t e = std::min<int64_t>((int64_t)r->in->end + 2000, r->fa.len - 1);
                blob += r->fa.getSequence(a, e);
                refoff.push_back((int64_t)blob.size());
                rss.push_back((int32_t)a); clen.push_back((int32_t)r->fa.len);
            }
        {
            Layout L;
            scanbegin.push_back((int32_t)nGood);
            L.add(z.c_ref, blob.size() + PLAT_BLOB_PAD); L.add(z.c_refoff, refoff.size()); L.add(z.c_rss, rss.size()); L.add(z.c_clen, clen.size());
            L.add(z.c_scanbegin, scanbegin.size()); L.add(z.c_scanlongest, scanlongest.size());
            L.commit(z, z.a_cin);
            memcpy(z.c_ref.h, blob.data(), blob.size()); memset(z.c_ref.h + blob.size(), 0, PLAT_BLOB_PAD);
            fill(z, z.c_refoff, refoff); fill(z, z.c_rss, rss); fill(z, z.c_clen, clen); fill(z, z.c_scanbegin, scanbegin); fill(z, z.c_scanlongest, scanlongest);
            L.upload(z, z.a_cin);
        }
        refBlob.swap(blob);
        if (nGood == 0) { hostTally = true; deviceB = false; return; }      // nothing to scan: the (empty) host tally
        plat_candidate_batch cb;
        memset(&cb, 0, sizeof cb);
        cb.n_regions = nScan; cb.n_reads = (int32_t)nGood;
        cb.ref_seq = z.c_ref.d; cb.ref_off = z.c_refoff.d; cb.ref_seq_start = z.c_rss.d; cb.contig_len = z.c_clen.d;
        cb.read_seq = z.t_seq.d; cb.read_qual = z.t_qual.d; cb.read_off = z.t_off.d; cb.read_pos = z.t_pos.d; cb.read_flags = z.t_flags.d;
        cb.cigar = z.t_cigar.d; cb.cig_off = z.t_cigoff.d;
        hostTally = getenv("PLAT_CALLER_HOST_TALLY") != nullptr;         // (measurements / tests: merge the records on the host)
        for (;;) {
            // records stay on the device when the merge kernel can take them: c_cnt / c_status / c_rec are laid out for a download
            // all the same (the host tally needs them when a scan overflows the kernel's table)
            Layout LO;
            LO.add(z.c_cnt, nGood); LO.add(z.c_status, nGood); LO.add(z.c_rec, nGood * (size_t)maxPerRead * 5);
            LO.commit(z, z.a_cout);
            recArenaBytes = LO.total; recordsOnHost = false;
            ck(plat_candidates_batch(z.ctx, &cb, o.minFlank, o.minBaseQual, o.genSNPs, o.genIndels, maxPerRead, z.t_region.d, z.c_rec.d, z.c_cnt.d,
                                     z.c_status.d, z.stream), "plat_candidates_batch");
            int need = 0;
            if (!hostTally) {
                // addVariantToList + the per-sample support filter on the device (variant.pyx:499-527, variantcaller.pyx:456-467)
                Layout LM;
                LM.add(z.m_n, (size_t)nScan * 2); LM.add(z.m_cand, (size_t)nScan * mergeCap * 8);
                LM.commit(z, z.a_mout);
                {
                    // (its table is 64 KB per scan in the context's scratch: a cohort too wide for it falls back to the host tally, it does not fail the call)
                    const int rcm = plat_candidates_merge_batch(z.ctx, &cb, z.t_end.d, nScan, z.c_scanbegin.d, z.c_scanlongest.d, maxPerRead, z.c_rec.d, z.c_cnt.d,
                                                                z.c_status.d, o.minVarFreq, mergeCap, z.m_cand.d, z.m_n.d, z.stream);
                    // (no room for the table: the records this scan has just written are merged on the host instead -- they are NOT scanned again;
                    //  a device that is really out of memory fails the next allocation of the chunk with the same code, loudly)
                    if (rcm == PLAT_ERR_NOMEM) hostTally = true;
                    else ck(rcm, "plat_candidates_merge_batch");
                }
                if (!hostTally) {
                    lmLayout = LM;
                    if (deviceB) launchStageB();
                    if (deviceB) LM.downloadFirst(z, z.a_mout, 1);                              // (only the counts: the candidates stay on the device)
                    else LM.download(z, z.a_mout);
                    z.sync("candidate scan");
                    for (int g = 0; g < nScan; ++g) {
                        const int st_ = z.m_n.h[2 * g + 1];
                        if (st_ == PLAT_ERR_BAD_INPUT) throw DeviceError(PLAT_ERR_BAD_INPUT, "a read reaches outside the reference window handed over, or read pointers out of order");
                        if (st_ <= -(1 << 20)) need = std::max(need, -st_ - (1 << 20));
                        else if (st_ != 0) hostTally = true;                // more distinct records / candidates than the kernel takes
                    }
                    if (!need && !hostTally) break;
                    if (need) { maxPerRead = need; continue; }
                }
            }
            LO.download(z, z.a_cout);
            z.sync("candidate scan");
            recordsOnHost = true;
            for (size_t i = 0; i < nGood; ++i) {
                if (z.c_status.h[i] == PLAT_ERR_BAD_INPUT) throw DeviceError(PLAT_ERR_BAD_INPUT, "a read reaches outside the reference window handed over");
                if (z.c_status.h[i] == PLAT_ERR_OVERFLOW) need = std::max(need, z.c_cnt.h[i]);
            }
            if (!need) break;
            maxPerRead = need;                                              // a read with more candidates than its slice: again with room for it
        }
        if (hostTally) deviceB = false;
    }
    bool hostTally = false;
    int mergeCap = 2048;
    std::string refBlob;
    Layout lmLayout;

    // -- B on the device (plat_stage_b_batch): regions with one sample, candidates from the reads alone, no reference-call blocks
    bool deviceB = false;
    DeviceBatch devBatch;
    int capV = 768, capW = 512, capA = 4096;
    Layout sbOut;
    bool eligibleDeviceB() const {
        if (nInd != 1 || o.assemble || o.outputRefCalls || !o.getVariantsFromBAMs || o.maxHaplotypes < 3 || regions.empty()) return false;
        const char* e = getenv("PLAT_CALLER_HOST_B");                       // (measurements / tests: stage B on the host)
        return !(e && e[0] == '1');
    }
    void launchStageB() {
        Slot& z = s;
        const size_t nR = regions.size();
        size_t nBr = 0;
        for (RegionWork* r : regions) nBr += (size_t)r->samples[0].broken.n();
        Layout LI;
        LI.add(z.sb_rstart, nR); LI.add(z.sb_rend, nR); LI.add(z.sb_rlen, nR); LI.add(z.sb_tabbegin, 3 * nR); LI.add(z.sb_tabn, 3 * nR); LI.add(z.sb_tablongest, 3 * nR);
        LI.add(z.sb_matepos, nBr + 1);
        LI.commit(z, z.a_bin);
        size_t mo = 0;
        for (size_t g = 0; g < nR; ++g) {
            RegionWork& r = *regions[g];
            SampleView& sv = r.samples[0];
            z.sb_rstart.h[g] = r.in->start; z.sb_rend.h[g] = r.in->end; z.sb_rlen.h[g] = r.rlen;
            const TableView* tv[3] = {&sv.reads, &sv.bad, &sv.broken};
            for (int k = 0; k < 3; ++k) { z.sb_tabbegin.h[3 * g + k] = (int32_t)tv[k]->base; z.sb_tabn.h[3 * g + k] = tv[k]->n(); z.sb_tablongest.h[3 * g + k] = tv[k]->longest; }
            if (sv.broken.n()) memcpy(z.sb_matepos.h + mo, sv.broken.t->mate_pos, sizeof(int32_t) * (size_t)sv.broken.n());
            mo += (size_t)sv.broken.n();
        }
        z.sb_matepos.h[mo] = 0;
        LI.upload(z, z.a_bin);
        const size_t capBW = nR * (size_t)capW, capBH = nR * 2048, capBR = std::max<size_t>(4 * (nGood + nBad + nBroken), 65536), capHB = capBH * 1280;
        Layout LO;
        LO.add(z.sb_hdr, 8 * nR); LO.add(z.sb_totals, 16);
        LO.add(z.sb_vpos, nR * capV); LO.add(z.sb_vnrem, nR * capV); LO.add(z.sb_vnadd, nR * capV); LO.add(z.sb_vsupp, nR * capV); LO.add(z.sb_vbmin, nR * capV);
        LO.add(z.sb_vbmax, nR * capV); LO.add(z.sb_vrempos, nR * capV); LO.add(z.sb_vaddoff, nR * capV); LO.add(z.sb_added, nR * capA);
        LO.add(z.sb_wstart, capBW); LO.add(z.sb_wend, capBW); LO.add(z.sb_wvfirst, capBW); LO.add(z.sb_wvn, capBW); LO.add(z.sb_wflags, capBW); LO.add(z.sb_wnhaps, capBW);
        LO.add(z.sb_wbatch, capBW); LO.add(z.sb_wptrs, 6 * capBW); LO.add(z.sb_hapmask, capBH);
        LO.commit(z, z.a_bout);
        sbOut = LO;
        z.d_hapbegin.reserve(z.ctx, capBW + 2, false); z.d_readbegin.reserve(z.ctx, capBW + 2, false); z.d_start.reserve(z.ctx, capBW + 2, false);
        z.d_end.reserve(z.ctx, capBW + 2, false); z.d_flank.reserve(z.ctx, capBW + 2, false); z.d_segbegin.reserve(z.ctx, capBW + 2, false);
        z.d_ngood.reserve(z.ctx, capBW + 2, false); z.d_pairoff.reserve(z.ctx, capBW + 2, false); z.d_gloff.reserve(z.ctx, capBW + 2, false);
        z.d_hapoff.reserve(z.ctx, capBH + 2, false); z.d_hapseq.reserve(z.ctx, capHB + PLAT_BLOB_PAD, false, true, z.stream);
        z.d_readoff.reserve(z.ctx, capBR + 2, false); z.d_src.reserve(z.ctx, capBR + 2, false); z.d_kind.reserve(z.ctx, capBR + 2, false);
        z.d_scratch.reserve(z.ctx, 24 * capBW + 48 * nR + 64, false);
        plat_stage_b_in in;
        memset(&in, 0, sizeof in);
        in.n_regions = (int32_t)nR; in.cap_per_scan = mergeCap; in.cand = z.m_cand.d; in.cand_n = z.m_n.d;
        in.ref_seq = z.c_ref.d; in.ref_off = z.c_refoff.d; in.ref_seq_start = z.c_rss.d; in.contig_len = z.c_clen.d;
        in.region_start = z.sb_rstart.d; in.region_end = z.sb_rend.d; in.region_rlen = z.sb_rlen.d;
        in.read_seq = z.t_seq.d; in.read_off = z.t_off.d; in.read_pos = z.t_pos.d; in.read_end = z.t_end.d;
        in.tab_begin = z.sb_tabbegin.d; in.tab_n = z.sb_tabn.d; in.tab_longest = z.sb_tablongest.d; in.broken_mate_pos = z.sb_matepos.d; in.broken_base = (int32_t)(nGood + nBad);
        in.cap_vars = capV; in.cap_windows = capW; in.cap_added = capA;
        in.cap_batch_windows = (int32_t)capBW; in.cap_batch_haps = (int32_t)capBH; in.cap_batch_reads = (int32_t)capBR; in.cap_hap_bytes = (int64_t)capHB;
        plat_stage_b_options so;
        memset(&so, 0, sizeof so);
        so.minReads = o.minReads; so.maxSize = o.maxSize; so.mergeClusteredVariants = o.mergeClusteredVariants; so.maxVarDist = o.maxVarDist; so.minVarDist = o.minVarDist;
        so.largeWindows = o.largeWindows; so.maxVariants = o.maxVariants; so.maxHaplotypes = o.maxHaplotypes; so.filterVarsByCoverage = o.filterVarsByCoverage;
        so.skipDifficultWindows = o.skipDifficultWindows; so.maxReads = o.maxReads;
        plat_stage_b_out ob;
        memset(&ob, 0, sizeof ob);
        ob.hdr = z.sb_hdr.d; ob.var_pos = z.sb_vpos.d; ob.var_nrem = z.sb_vnrem.d; ob.var_nadd = z.sb_vnadd.d; ob.var_support = z.sb_vsupp.d; ob.var_bam_min = z.sb_vbmin.d;
        ob.var_bam_max = z.sb_vbmax.d; ob.var_rem_pos = z.sb_vrempos.d; ob.var_add_off = z.sb_vaddoff.d; ob.added = z.sb_added.d;
        ob.win_start = z.sb_wstart.d; ob.win_end = z.sb_wend.d; ob.win_var_first = z.sb_wvfirst.d; ob.win_var_n = z.sb_wvn.d; ob.win_flags = z.sb_wflags.d;
        ob.win_ptrs = z.sb_wptrs.d; ob.win_n_haps = z.sb_wnhaps.d; ob.win_batch = z.sb_wbatch.d;
        ob.b_hap_begin = z.d_hapbegin.d; ob.b_read_begin = z.d_readbegin.d; ob.b_start = z.d_start.d; ob.b_end = z.d_end.d; ob.b_flank = z.d_flank.d;
        ob.b_pair_off = z.d_pairoff.d; ob.b_gl_off = z.d_gloff.d; ob.b_seg_begin = z.d_segbegin.d; ob.b_n_good = z.d_ngood.d;
        ob.b_hap_off = z.d_hapoff.d; ob.b_hap_mask = z.sb_hapmask.d; ob.b_hap_seq = z.d_hapseq.d;
        ob.b_read_off = z.d_readoff.d; ob.b_read_src = z.d_src.d; ob.b_read_kind = z.d_kind.d; ob.totals = z.sb_totals.d; ob.scratch = z.d_scratch.d;
        const int rc = plat_stage_b_batch(z.ctx, &in, &so, &ob, z.stream);
        if (rc == PLAT_ERR_UNSUPPORTED) { deviceB = false; return; }        // (a device library without this stage: the host's own code)
        ck(rc, "plat_stage_b_batch");
        LO.download(z, z.a_bout);
    }

    // what plat_stage_b_batch left: Variant / WindowWork objects for the stages behind it.  A region (or window) the device flagged goes
    // through the host's own regionVariants / regionWindows (prepareWindow).
    void stageBFromDevice() {
        Slot& z = s;
        const size_t nR = regions.size();
        if (z.sb_totals.h[10] != 0) {                                      // a batch capacity was too small: the whole chunk on the host
            deviceB = false;
            lmLayout.download(z, z.a_mout);
            z.sync("candidates");
            int scan0 = 0;
            for (RegionWork* r : regions) { regionVariants(*r, scan0); ++scan0; regionWindows(*r); }
            std::lock_guard<std::mutex> g(stMutex);
            st.n_regions_stage_b_host += (int64_t)nR;
            return;
        }
        bool rows = false;
        for (size_t g = 0; g < nR; ++g)
            if (z.sb_hdr.h[8 * g] != 0) {                                   // this region's candidates for the host's code
                const size_t at = g * (size_t)mergeCap * 8, n = (size_t)std::max(0, z.m_n.h[2 * g]) * 8;
                if (n) ck(plat_memcpy_d2h(z.ctx, z.m_cand.h + at, z.m_cand.d + at, n * sizeof(int32_t), z.stream), "plat_memcpy_d2h");
                rows = true;
            }
        if (rows) z.sync("candidates");
        int hapRun = 0;
        int64_t nHostRegions = 0, nHostWindows = 0;
        for (size_t g = 0; g < nR; ++g) {
            RegionWork& r = *regions[g];
            const int32_t* hdr = z.sb_hdr.h + 8 * g;
            if (hdr[0] != 0) { regionVariants(r, (int)g); regionWindows(r); ++nHostRegions; continue; }
            PROF("s2.fillRegion");
            const int nV = hdr[1], nW = hdr[2];
            r.nCandRecords += hdr[3];
            r.variants.clear();
            const uint8_t* blob = z.sb_added.h + g * (size_t)capA;
            for (int i = 0; i < nV; ++i) {
                PROF("s2.fill.variant");
                const size_t k = g * (size_t)capV + (size_t)i;
                const int nrem = z.sb_vnrem.h[k], nadd = z.sb_vnadd.h[k];
                Variant* v = r.pool.make(z.sb_vpos.h[k], std::string((const char*)r.fa.seq + z.sb_vrempos.h[k], (size_t)nrem),
                                         std::string((const char*)blob + z.sb_vaddoff.h[k], (size_t)nadd), z.sb_vsupp.h[k], PLATYPUS_VAR);
                v->bamMinPos = z.sb_vbmin.h[k]; v->bamMaxPos = z.sb_vbmax.h[k];
                r.variants.push_back(v);
            }
            if (r.cur.size() != r.samples.size()) r.cur.assign(r.samples.size(), Ptrs{0, 0, 0, 0, 0, 0});
            r.windows.reserve(r.windows.size() + (size_t)nW); r.items.reserve(r.items.size() + (size_t)nW);
            for (int q = 0; q < nW; ++q) {
                PROF("s2.fill.window");
                const size_t k = g * (size_t)capW + (size_t)q;
                r.items.push_back(Item{0, (int)r.windows.size(), std::string(), 0});
                r.windows.emplace_back();                                   // (filled in place: a WindowWork is two dozen containers to move otherwise)
                WindowWork& w = r.windows.back();
                w.region = r.index; w.startPos = z.sb_wstart.h[k]; w.endPos = z.sb_wend.h[k];
                const int vf = z.sb_wvfirst.h[k], vn = z.sb_wvn.h[k], flags = z.sb_wflags.h[k], nH = z.sb_wnhaps.h[k], bw = z.sb_wbatch.h[k];
                for (int i = 0; i < vn; ++i) w.vars.push_back(r.variants[(size_t)(vf + i)]);
                w.allVars = w.vars;
                const int hap0 = hapRun;
                if (bw >= 0) hapRun += nH;
                if (flags & (PLAT_SBW_HOST | PLAT_SBW_DUPLICATE)) {         // the greedy filter, filterVariantsByCoverage, mergeHaplotypes, an exception: the host's code
                    ++nHostWindows;
                    try { prepareWindow(r, w); }
                    catch (const WindowError& e) {
                        logWindowFailure(r.in->chrom, w.startPos, w.endPos, e.what());
                        std::lock_guard<std::mutex> gd(stMutex);
                        ++st.n_windows_failed;
                        w.live = false; w.greedy = false; w.failed = true;
                    }
                } else {
                    w.hapStart = std::max(0, w.startPos);
                    w.hapEnd = (int)std::min<int64_t>(w.endPos, r.fa.len - 1);
                    w.endBuf = std::min(2 * r.rlen, 500);
                    const int32_t* pp = z.sb_wptrs.h + 6 * k;
                    w.ptrs.resize(1);
                    w.ptrs[0] = Ptrs{pp[0], pp[1], pp[2], pp[3], pp[4], pp[5]};
                    w.nReads = pp[1] - pp[0];
                    r.cur = w.ptrs;
                    if (flags == 0) {
                        w.live = true; w.onDevice = true; w.bw = bw; w.hapBegin = hap0;
                        w.haps.resize((size_t)nH);
                        for (int h = 0; h < nH; ++h) {
                            const uint32_t m = z.sb_hapmask.h[hap0 + h];
                            for (int i = 0; i < vn; ++i) if (m >> i & 1u) w.haps[(size_t)h].variants.push_back(w.vars[(size_t)i]);
                        }
                    }
                }
            }
        }
        // the batch the device built
        DeviceBatch& db = devBatch;
        db = DeviceBatch();
        const int64_t* T = z.sb_totals.h;
        db.nWindows = (int)T[0]; db.nHaps = (int)T[1]; db.nReads = (int)T[2]; db.nPairs = T[3]; db.nGl = T[4]; db.hapBlob = T[5]; db.readBlob = T[6];
        db.maxHap = (int)T[7]; db.maxR = (int)T[8]; db.maxH = (int)T[9]; db.maxRead = maxReadLen; db.nInd = 1;
        memset(&db.wb, 0, sizeof db.wb);
        db.wb.win_hap_begin = z.d_hapbegin.d; db.wb.win_read_begin = z.d_readbegin.d; db.wb.win_start = z.d_start.d; db.wb.win_end = z.d_end.d;
        db.wb.win_flank = z.d_flank.d; db.wb.pair_off = z.d_pairoff.d; db.wb.hap_seq = z.d_hapseq.d; db.wb.hap_off = z.d_hapoff.d;
        db.wb.read_off = z.d_readoff.d; db.wb.read_kind = z.d_kind.d;
        db.hapbegin = z.d_hapbegin.d; db.gloff = z.d_gloff.d; db.ngood = z.d_ngood.d; db.segbegin = z.d_segbegin.d; db.src = z.d_src.d; db.readoff = z.d_readoff.d;
        std::lock_guard<std::mutex> g(stMutex);
        st.n_regions_stage_b_device += (int64_t)nR - nHostRegions; st.n_regions_stage_b_host += nHostRegions; st.n_windows_stage_b_host += nHostWindows;
    }

    // -- A3: the assembler part of generateVariantsInRegion (variantcaller.pyx:496-519): tiles of assemblyRegionSize every
    // max(100, min(1000, size / 2)) bases, doWeNeedToAssembleThisRegion (:276-321) per tile, the reads loadBAMDataIntoGraph would load
    // (assembler.pyx:1391-1425: good reads between the window pointers, badReads / brokenMates if the options say so, QCFail reads never)
    // gathered from the chunk's device table; ALL tiles of the chunk in one plat_assemble_batch
    struct Tile { int region, assemStart, assemEnd, refStart; };
    void assembleTiles() {
        if (!o.assemble) return;
        Slot& z = s;
        const auto t0 = Clock::now();
        const int size = o.assemblyRegionSize;
        if (size <= 0) throw DeviceError(PLAT_ERR_INVALID, "assemblyRegionSize");
        const int shift = std::max(100, std::min(1000, size / 2));
        std::vector<Tile> tiles;
        std::vector<int64_t> refoff{0}, roff{0};
        std::vector<int32_t> refstart, astart, aend, rbegin{0}, src;
        std::string blob;
        for (RegionWork* rp : regions) {
            RegionWork& r = *rp;
            r.cur.assign(r.samples.size(), Ptrs{0, 0, 0, 0, 0, 0});
            for (int64_t a0 = r.in->start; a0 < r.in->end; a0 += shift) {
                const int assemStart = (int)a0, assemEnd = (int)std::min<int64_t>(a0 + size, r.in->end);
                const int refStart = std::max(0, assemStart - size);
                const std::string refSeq = r.fa.getSequence(refStart, (int64_t)assemEnd + size);
                // doWeNeedToAssembleThisRegion: the window pointers move to the tile whatever the answer
                bool need = o.assembleAll != 0;
                for (size_t i = 0; i < r.samples.size(); ++i) {
                    Ptrs& p = r.cur[i];
                    r.samples[i].reads.overlapRange(assemStart, assemEnd, p.gs, p.ge);
                    r.samples[i].bad.overlapRange(assemStart, assemEnd, p.bs, p.be);
                    r.samples[i].broken.matePosRange(assemStart, assemEnd, p.ks, p.ke);
                }
                for (size_t i = 0; !need && i < r.samples.size(); ++i) {
                    const Ptrs& p = r.cur[i];
                    const double n = p.ge - p.gs, nBad = p.be - p.bs;
                    if (n == 0) continue;
                    double gaps = 0, improper = 0;                           // countAlignmentGaps / countImproperPairs (cwindow.pyx:598-647): reads + badReads
                    auto scan = [&](const TableView& tv, int b, int e) {
                        for (int q = b; q < e; ++q) {
                            for (int c = tv.t->cig_off[q]; c < tv.t->cig_off[q + 1]; ++c) { const int op = tv.t->cigar[2 * c]; gaps += op >= 1 && op <= 4; }
                            improper += !(tv.t->flags[q] & 2);
                        }
                    };
                    scan(r.samples[i].reads, p.gs, p.ge); scan(r.samples[i].bad, p.bs, p.be);
                    if (gaps / n > 2 || improper / (n + nBad) > 0.1) need = true;
                }
                if (!need) continue;
                tiles.push_back(Tile{regionSlot(r.index), assemStart, assemEnd, refStart});
                blob += refSeq;
                refoff.push_back((int64_t)blob.size());
                refstart.push_back(refStart); astart.push_back(assemStart); aend.push_back(assemEnd);
                for (size_t i = 0; i < r.samples.size(); ++i) {
                    const Ptrs& p = r.cur[i];
                    auto take = [&](const TableView& tv, int b, int e) {
                        for (int q = b; q < e; ++q) {
                            if (tv.t->flags[q] & 512) continue;               // Read_IsQCFail
                            src.push_back((int32_t)(tv.base + q));
                            roff.push_back(roff.back() + tv.rlen(q));
                        }
                    };
                    take(r.samples[i].reads, p.gs, p.ge);
                    if (o.assembleBadReads) take(r.samples[i].bad, p.bs, p.be);
                    if (o.assembleBrokenPairs) take(r.samples[i].broken, p.ks, p.ke);
                }
                rbegin.push_back((int32_t)src.size());
            }
        }
        const int nT = (int)tiles.size();
        if (nT > 0) {
            Layout L;
            L.add(z.as_ref, blob.size() + PLAT_BLOB_PAD); L.add(z.as_refoff, refoff.size()); L.add(z.as_refstart, refstart.size()); L.add(z.as_astart, astart.size());
            L.add(z.as_aend, aend.size()); L.add(z.as_rbegin, rbegin.size()); L.add(z.as_src, src.size()); L.add(z.as_roff, roff.size());
            L.commit(z, z.a_asin);
            memcpy(z.as_ref.h, blob.data(), blob.size()); memset(z.as_ref.h + blob.size(), 0, PLAT_BLOB_PAD);
            fill(z, z.as_refoff, refoff); fill(z, z.as_refstart, refstart); fill(z, z.as_astart, astart); fill(z, z.as_aend, aend); fill(z, z.as_rbegin, rbegin);
            fill(z, z.as_src, src); fill(z, z.as_roff, roff);
            L.upload(z, z.a_asin);
            const size_t nR = src.size(), nb = (size_t)roff.back();
            z.as_seq.reserve(z.ctx, nb + PLAT_BLOB_PAD, false, true, z.stream); z.as_qual.reserve(z.ctx, nb + PLAT_BLOB_PAD, false, true, z.stream);
            z.as_pos.reserve(z.ctx, nR + 1, false); z.as_end.reserve(z.ctx, nR + 1, false); z.as_flags.reserve(z.ctx, nR + 1, false); z.as_mapq.reserve(z.ctx, nR + 1, false);
            if (nR) ck(plat_gather_reads(z.ctx, (int64_t)nR, z.as_src.d, z.as_roff.d, z.t_seq.d, z.t_qual.d, z.t_off.d, z.t_pos.d, z.t_end.d, z.t_mapq.d, z.t_flags.d,
                                         z.as_seq.d, z.as_qual.d, z.as_pos.d, z.as_end.d, z.as_mapq.d, z.as_flags.d, z.stream), "plat_gather_reads(assembler)");
            plat_assembly_batch ab;
            memset(&ab, 0, sizeof ab);
            ab.n_regions = nT; ab.n_reads = (int32_t)nR;
            ab.ref_seq = z.as_ref.d; ab.ref_off = z.as_refoff.d; ab.ref_start = z.as_refstart.d; ab.assem_start = z.as_astart.d; ab.assem_end = z.as_aend.d;
            ab.reg_read_begin = z.as_rbegin.d; ab.read_seq = z.as_seq.d; ab.read_qual = z.as_qual.d; ab.read_off = z.as_roff.d;
            for (;;) {                                                       // room per tile grows until every tile's variants fit
                Layout LO;
                LO.add(z.as_cnt, (size_t)nT); LO.add(z.as_status, (size_t)nT); LO.add(z.as_vpos, (size_t)nT * asmMaxVars); LO.add(z.as_nrem, (size_t)nT * asmMaxVars);
                LO.add(z.as_nadd, (size_t)nT * asmMaxVars); LO.add(z.as_off, (size_t)nT * asmMaxVars); LO.add(z.as_blob, (size_t)nT * asmBlob);
                LO.commit(z, z.a_asout);
                ck(plat_assemble_batch(z.ctx, &ab, o.assemblerKmerSize, o.minBaseQual, o.minReads * o.minBaseQual, o.noCycles, asmMaxVars, asmBlob, z.as_cnt.d,
                                       z.as_vpos.d, z.as_nrem.d, z.as_nadd.d, z.as_off.d, z.as_blob.d, z.as_status.d, z.stream), "plat_assemble_batch");
                LO.download(z, z.a_asout);
                z.sync("assembler");
                bool over = false;
                for (int g = 0; g < nT; ++g) {
                    if (z.as_status.h[g] == PLAT_ERR_OVERFLOW) over = true;
                    else if (z.as_status.h[g] != 0) throw DeviceError(z.as_status.h[g], "plat_assemble_batch(tile)");
                }
                if (!over) break;
                if (asmMaxVars >= (1 << 14)) throw DeviceError(PLAT_ERR_OVERFLOW, "plat_assemble_batch(tile)");
                asmMaxVars *= 4; asmBlob *= 4;
            }
            int64_t nv = 0;
            for (int g = 0; g < nT; ++g) {                                   // per tile in the reference's sorted() order (the device's), tile after tile
                RegionWork& r = *regions[(size_t)tiles[(size_t)g].region];
                const uint8_t* raw = z.as_blob.h + (size_t)g * (size_t)asmBlob;
                for (int i = 0; i < z.as_cnt.h[g]; ++i) {
                    const size_t k = (size_t)g * (size_t)asmMaxVars + (size_t)i;
                    const int off = z.as_off.h[k], nrem = z.as_nrem.h[k], nadd = z.as_nadd.h[k];
                    r.asmVariants.push_back(r.pool.make(z.as_vpos.h[k], std::string((const char*)raw + off, (size_t)nrem),
                                                        std::string((const char*)raw + off + nrem, (size_t)nadd), 0, ASSEMBLER_VAR));
                    ++nv;
                }
            }
            std::lock_guard<std::mutex> g(stMutex);
            st.n_assembly_tiles += nT; st.n_assembler_variants += nv;
        }
        std::lock_guard<std::mutex> g(stMutex);
        st.seconds_assemble += secs(t0, Clock::now());
    }
    int asmMaxVars = 64, asmBlob = 4096;

    // -- B1: candidates of one region -> merged, per-sample support filter, left-normalised, filtered (variantcaller.pyx:439-531)
    // one sample's variantHeap: its distinct records in first-occurrence order with the number of reads showing each (addVariantToList),
    // from the scan's records on the host
    struct CandKey { int pos, nrem, nadd, count; const char* rem; const char* add; };
    void tallySample(const RegionWork& r, size_t i, std::vector<CandKey>& keys, std::deque<std::string>& addedStore, int64_t* nRecords) {
        Slot& z = s;
        const TableView& tv = r.samples[i].reads;
        keys.clear();
        std::vector<int32_t> table;                                         // open addressing over `keys` (index + 1, 0 = empty)
        size_t tmask = 4095;
        table.assign(tmask + 1, 0);
        auto hashKey = [](const CandKey& k) -> size_t {
            size_t h = (size_t)k.pos * 1000003u + (size_t)k.nrem * 131u + (size_t)k.nadd;
            for (int j = 0; j < k.nrem; ++j) h = h * 31u + (unsigned char)k.rem[j];
            for (int j = 0; j < k.nadd; ++j) h = h * 37u + (unsigned char)k.add[j];
            return h * 0x9E3779B97F4A7C15ull >> 20;
        };
        auto sameKey = [](const CandKey& a, const CandKey& b) {
            return a.pos == b.pos && a.nrem == b.nrem && a.nadd == b.nadd && memcmp(a.rem, b.rem, (size_t)a.nrem) == 0 && memcmp(a.add, b.add, (size_t)a.nadd) == 0;
        };
        const int64_t blobBase = tv.blobBase;
        for (int q = 0; q < tv.n(); ++q) {
            const size_t g = (size_t)(tv.base + q);
            const int cnt = z.c_cnt.h[g];
            for (int k = 0; k < cnt; ++k) {
                const int32_t* rec = z.c_rec.h + 5 * (g * (size_t)maxPerRead + (size_t)k);
                const char* addp = "";
                if (rec[2]) {
                    if (tv.t->encoding == PLAT_READS_ASCII) addp = (const char*)tv.t->seq + (rec[4] - blobBase);
                    else { addedStore.push_back(tableBases(*tv.t, rec[4] - blobBase, rec[2])); addp = addedStore.back().data(); }
                }
                CandKey key{std::max(0, rec[0]), rec[1], rec[2], 1, rec[1] ? refBlob.data() + rec[3] : "", addp};
                if (nRecords) ++*nRecords;
                size_t slot = hashKey(key) & tmask;
                while (table[slot] && !sameKey(keys[(size_t)table[slot] - 1], key)) slot = (slot + 1) & tmask;
                if (table[slot]) { ++keys[(size_t)table[slot] - 1].count; continue; }    // one more read showing it (addVariantToList)
                keys.push_back(key);
                table[slot] = (int32_t)keys.size();
                if (keys.size() * 2 > tmask) {                          // grow
                    tmask = tmask * 2 + 1;
                    table.assign(tmask + 1, 0);
                    for (size_t e = 0; e < keys.size(); ++e) { size_t s2 = hashKey(keys[e]) & tmask; while (table[s2]) s2 = (s2 + 1) & tmask; table[s2] = (int32_t)e + 1; }
                }
            }
        }
    }
    // :456-467: per-sample support, indels always
    bool passesSupport(const RegionWork& r, size_t i, const CandKey& k) const {
        int s0, e0;
        r.samples[i].reads.overlapRange(k.pos, k.pos + 1, s0, e0);
        const int total = e0 - s0;
        const double frac = total == 0 ? 0.0 : (double)k.count / total;
        return frac >= o.minVarFreq || k.nadd != k.nrem;
    }

    // -- B1: candidates of one region -> merged, per-sample support filter, left-normalised, filtered (variantcaller.pyx:439-531)
    void regionVariants(RegionWork& r, int scan0) {
        Slot& z = s;
        VarList everyone;                                                   // the all-samples generator's variantHeap, insertion order
        std::unordered_map<std::string, Variant*> everyoneIndex;
        // a candidate of one sample that passed the support filter joins the all-samples dictionary: equal variants of different
        // samples merge (addVariantToList, variant.pyx:499-527)
        const bool oneSample = r.samples.size() == 1;                       // a sample's candidates are distinct already: nothing to merge them with
        auto pass = [&](int pos, const char* rem, int nrem, const char* add, int nadd, int count) {
            if (oneSample) {
                everyone.push_back(r.pool.make(pos, std::string(rem, (size_t)nrem), std::string(add, (size_t)nadd), count, PLATYPUS_VAR));
                return;
            }
            std::string key = std::to_string(pos);
            key += '|'; key.append(rem, (size_t)nrem); key += '|'; key.append(add, (size_t)nadd);
            auto it = everyoneIndex.find(key);
            if (it != everyoneIndex.end()) {
                Variant tmp(pos, std::string(), std::string(), count, PLATYPUS_VAR);
                it->second->addVariant(tmp);
            } else {
                Variant* v = r.pool.make(pos, std::string(rem, (size_t)nrem), std::string(add, (size_t)nadd), count, PLATYPUS_VAR);
                everyoneIndex.emplace(std::move(key), v);
                everyone.push_back(v);
            }
        };
        if (!hostTally && o.getVariantsFromBAMs) {
            PROF("s2.rv.cands");
            // merged and filtered on the device (plat_candidates_merge_batch): the scan's candidates in the order of their first records
            for (size_t i = 0; i < r.samples.size(); ++i) {
                const TableView& tv = r.samples[i].reads;
                const int g = scan0 + (int)i, n = z.m_n.h[2 * g];
                const int64_t blobBase = tv.blobBase;
                std::vector<const int32_t*> cands((size_t)n);
                for (int k = 0; k < n; ++k) cands[(size_t)k] = z.m_cand.h + 8 * ((size_t)g * (size_t)mergeCap + (size_t)k);
                std::sort(cands.begin(), cands.end(), [](const int32_t* a, const int32_t* b) { return a[0] < b[0]; });
                for (const int32_t* c : cands) {
                    r.nCandRecords += c[1];
                    const std::string added = tableBases(*tv.t, c[7] - blobBase, c[5]);
                    pass(std::max(0, c[3]), c[4] ? refBlob.data() + c[6] : "", c[4], added.data(), c[5], c[1]);
                }
            }
        }
        std::vector<CandKey> keys;                                          // a sample's variantHeap: distinct records, first-occurrence order
        std::deque<std::string> addedStore;                                 // (letters of the added bases when the table is not ASCII)
        for (size_t i = 0; hostTally && o.getVariantsFromBAMs && i < r.samples.size(); ++i) {
            tallySample(r, i, keys, addedStore, &r.nCandRecords);
            // Only the candidates that pass become Variant objects (the sample's own heap is not looked at again).
            for (const CandKey& k : keys) if (passesSupport(r, i, k)) pass(k.pos, k.rem, k.nrem, k.add, k.nadd, k.count);
        }
        std::stable_sort(everyone.begin(), everyone.end(), variantLess);    // getCandidates(): sorted(values)
        // rawBamVariants + assemblerVariants (:521), left-normalised, sorted, filtered (:523-531)
        VarList norm;
        auto finish = [&](const VarList& raw) {
            VarList all(raw);
            all.insert(all.end(), r.asmVariants.begin(), r.asmVariants.end());
            norm.clear();
            for (Variant* v : all) norm.push_back(leftNormaliseIndel(v, r.fa, r.rlen, r.pool));
            std::stable_sort(norm.begin(), norm.end(), variantLess);
            r.variants = filterVariants(norm, o.minReads, o.minReads, o.maxSize);
        };
        std::vector<Variant> asmBackup;                                     // (filterVariants adds the support of equal neighbours up in place)
        for (const Variant* v : r.asmVariants) asmBackup.push_back(*v);
        finish(everyone);
        // `sorted` is stable: candidates that compare equal (two alleles of one type and length at one position) stay in the order the
        // all-samples dictionary yields them, a Python-2 dict keyed by Variant (hash of (refName, refPos, removed, added),
        // variant.pyx:270-280) that was filled while walking each sample's dictionary of the same kind (variantcaller.pyx:457).  Every
        // other order is decided by the keys.  That order can only reach the result where two of the variants that are KEPT compare
        // equal, or where a run of equal keys holds a variant twice (equal neighbours are merged by filterVariants: who is whose
        // neighbour then depends on it) next to a different one -- most regions hold such pairs only among the sequencing errors that
        // are dropped.  Only a region where it can matter pays for replaying the dictionaries.
        bool replay = false;
        for (size_t k = 1; k < r.variants.size() && !replay; ++k)
            replay = !variantLess(r.variants[k - 1], r.variants[k]) && !variantLess(r.variants[k], r.variants[k - 1]);
        for (size_t a = 0; a < norm.size() && !replay;) {
            size_t e = a + 1;
            while (e < norm.size() && !variantLess(norm[a], norm[e])) ++e;  // (sorted: not less = equal key)
            if (e - a >= 3) {
                bool twice = false, other = false;
                for (size_t x = a; x < e; ++x)
                    for (size_t y = x + 1; y < e; ++y) { if (norm[x]->same(*norm[y])) twice = true; else other = true; }
                replay = twice && other;
            }
            a = e;
        }
        if (replay && o.getVariantsFromBAMs && !getenv("PLAT_CALLER_FIRST_OCCURRENCE_ORDER")) {      // (the switch: tests only, to show the replay matters)
            if (getenv("PLAT_CALLER_TRACE")) fprintf(stderr, "[plat_caller] region %s: candidates that compare equal are kept, dictionaries replayed\n", r.in->chrom ? r.in->chrom : "?");
            PROF("s2.rv.replay");
            if (!hostTally && !recordsOnHost) {                             // the scan's records are still on the device: this region's reads' rows
                PROF("s2.rv.replay.d2h");
                for (const SampleView& sv : r.samples) {
                    const size_t b0 = (size_t)sv.reads.base, n = (size_t)sv.reads.n();
                    if (!n) continue;
                    ck(plat_memcpy_d2h(z.ctx, z.c_cnt.h + b0, z.c_cnt.d + b0, n * sizeof(int32_t), z.stream), "plat_memcpy_d2h");
                    const size_t row = (size_t)maxPerRead * 5;
                    ck(plat_memcpy_d2h(z.ctx, z.c_rec.h + b0 * row, z.c_rec.d + b0 * row, n * row * sizeof(int32_t), z.stream), "plat_memcpy_d2h");
                }
                z.sync("candidate records");
            }
            const uint64_t nameHash = py2_string_hash(r.in->chrom ? std::string(r.in->chrom) : std::string());
            VarList all;
            std::vector<uint64_t> allHash;
            std::unordered_map<std::string, size_t> allIndex;
            for (size_t i = 0; i < r.samples.size(); ++i) {
                { PROF("s2.rv.replay.tally"); tallySample(r, i, keys, addedStore, nullptr); }
                std::vector<uint64_t> hs(keys.size());
                PROF("s2.rv.replay.order");
                for (size_t k = 0; k < keys.size(); ++k) hs[k] = py2_variant_hash(nameHash, keys[k].pos, keys[k].rem, (size_t)keys[k].nrem, keys[k].add, (size_t)keys[k].nadd);
                for (int k : py2_dict_slot_order(hs)) {                     // varCandGen.variantHeap.iteritems()
                    const CandKey& c = keys[(size_t)k];
                    if (!passesSupport(r, i, c)) continue;
                    std::string key = std::to_string(c.pos);
                    key += '|'; key.append(c.rem, (size_t)c.nrem); key += '|'; key.append(c.add, (size_t)c.nadd);
                    auto it = allIndex.find(key);
                    if (it != allIndex.end()) {
                        Variant tmp(c.pos, std::string(), std::string(), c.count, PLATYPUS_VAR);
                        all[it->second]->addVariant(tmp);
                    } else {
                        allIndex.emplace(std::move(key), all.size());
                        all.push_back(r.pool.make(c.pos, std::string(c.rem, (size_t)c.nrem), std::string(c.add, (size_t)c.nadd), c.count, PLATYPUS_VAR));
                        allHash.push_back(hs[(size_t)k]);
                    }
                }
            }
            everyone.clear();
            for (int k : py2_dict_slot_order(allHash)) everyone.push_back(all[(size_t)k]);     // allSampleVarCandGen.variantHeap.values()
            std::stable_sort(everyone.begin(), everyone.end(), variantLess);
            for (size_t k = 0; k < asmBackup.size(); ++k) *r.asmVariants[k] = asmBackup[k];
            finish(everyone);
        }
    }
    bool recordsOnHost = false;
    size_t recArenaBytes = 0;

    // -- B2/B3: windows, window pointers, haplotype enumeration (callVariantsInWindow up to Population.setup)
    Hap makeHap(const RegionWork& r, const WindowWork& w, const VarList& vs) const {
        Hap h;
        h.variants = vs;
        h.seq = haplotypeSequence(r.fa, w.hapStart, w.hapEnd, w.endBuf, vs);
        if (h.seq.size() > 16384) throw WindowError("Haplotype is too long. Max allowed length is 16384");   // chaplotype.pyx:180-183
        return h;
    }

    void regionWindows(RegionWork& r) {
        WindowOptions wo{o.mergeClusteredVariants, o.maxVarDist, o.minVarDist, o.maxSize, o.largeWindows, r.rlen, o.maxVariants, o.outputRefCalls, o.refCallBlockSize};
        std::vector<Window> wins;
        { PROF("s2.windowsAndVariants"); wins = windowsAndVariants(r.in->start, r.in->end, r.fa.len - 1, r.variants, wo); }
        if (r.cur.size() != r.samples.size()) r.cur.assign(r.samples.size(), Ptrs{0, 0, 0, 0, 0, 0});
        r.windows.reserve(r.windows.size() + wins.size()); r.items.reserve(r.items.size() + wins.size());
        for (Window& win : wins) {
            if (win.variants.empty()) {                                      // a reference-call block between calling windows (:605-607)
                if (o.outputRefCalls) {
                    Item it{1, -1, std::string(), 0};
                    try {
                        if (refCallLine(r, it.text, win.startPos, win.endPos, snapshotNR(r.cur), false, 0.0)) it.nRef = 1;
                    } catch (const WindowError& e) { logWindowFailure(r.in->chrom, win.startPos, win.endPos, e.what()); }
                    r.items.push_back(std::move(it));
                }
                continue;
            }
            if (win.endPos - win.startPos > o.maxSize) continue;             // variantcaller.pyx:566-568
            WindowWork w;
            w.region = r.index; w.startPos = win.startPos; w.endPos = win.endPos;
            w.vars = win.variants; w.allVars = win.variants;
            try {
                PROF("s2.prepareWindow");
                prepareWindow(r, w);
            } catch (const WindowError& e) {
                logWindowFailure(r.in->chrom, w.startPos, w.endPos, e.what());
                std::lock_guard<std::mutex> g(stMutex);
                ++st.n_windows_failed;
                w.live = false; w.greedy = false; w.failed = true;
            }
            r.items.push_back(Item{0, (int)r.windows.size(), std::string(), 0});
            r.windows.push_back(std::move(w));
        }
    }
    static std::vector<int> snapshotNR(const PtrList& ptrs) {
        std::vector<int> nr;
        for (const Ptrs& p : ptrs) nr.push_back(p.ge - p.gs);
        return nr;
    }

    // outputRefCall (variantcaller.pyx:764-867) for [windowStart, windowEnd): QUAL 0 without coverage somewhere in the block; else the
    // phred-scaled beta-binomial p-value of seeing no variant read at the block's smallest coverage, capped -- when the block holds
    // candidates -- by the best candidate's posterior under a flat prior (maxPost).  nReads: the samples' reads between the window
    // pointers as the loop last left them (the reference does not move them for a block).  Returns false when the reference raises here
    // (an infinite QUAL: logged and skipped by its try/except).
    bool refCallLine(const RegionWork& r, std::string& out, int windowStart, int windowEnd, const std::vector<int>& nReads, bool hasVariants, double maxPost) const {
        long minCov = -1;
        for (const SampleView& sv : r.samples) {
            const TableView& tv = sv.reads;
            const int N = tv.n();
            for (int p = windowStart; p < windowEnd; ++p) {                  // countReadsCoveringRegion(p, p + 1), cwindow.pyx:176-206
                long c = 0;
                if (N > 0) {
                    int s0 = TableView::lowerBound(tv.t->pos, N, std::max<int64_t>(1, (int64_t)p - tv.longest));
                    const int e0 = TableView::lowerBound(tv.t->pos, N, (int64_t)p + 1);
                    while (s0 < N && tv.t->end[s0] <= p) ++s0;
                    if (s0 > e0) throw WindowError("This should never happen. Read start pointer > read end pointer!!");
                    c = std::min(e0, N) - s0;
                }
                minCov = minCov == -1 ? c : std::min(minCov, c);
            }
        }
        const int phredPValue = (int)(-10 * log10(betaBinomialCDF(0, minCov, 20, 20)));
        int qual;
        if (minCov == 0) qual = 0;
        else if (!hasVariants) qual = phredPValue;
        else {
            const double maxProbVar = 1.0 - pow(10.0, -0.1 * maxPost), probRef = 1.0 - maxProbVar;
            const double v = -10.0 * log10(1.0 - probRef);
            if (std::isinf(v) || std::isnan(v)) return false;               // int(round(inf)) raises there
            qual = std::min((int)py2_round0(v), phredPValue);
        }
        const std::string ref = r.fa.getSequence(windowStart, (int64_t)windowStart + 1);
        writeRefCallLine(out, r.in->chrom, windowStart, windowEnd, ref.empty() ? 'N' : ref[0], qual, nReads);
        return true;
    }

    void prepareWindow(RegionWork& r, WindowWork& w) {
        w.hapStart = std::max(0, w.startPos);
        w.hapEnd = (int)std::min<int64_t>(w.endPos, r.fa.len - 1);
        w.endBuf = std::min(2 * r.rlen, 500);                               // chaplotype.pyx:142
        { PROF("s2.pw.refseq"); w.refSeq = haplotypeSequence(r.fa, w.hapStart, w.hapEnd, w.endBuf, VarList()); }
        if (w.refSeq.size() > 16384) throw WindowError("Haplotype is too long. Max allowed length is 16384");
        w.ptrs.resize(r.samples.size());
        w.nReads = 0;
        PROF("s2.pw.rest");
        { PROF("s2.pw.ptrs");
        for (size_t i = 0; i < r.samples.size(); ++i) {                    // bamReadBuffer.setWindowPointers (cwindow.pyx:655-689)
            Ptrs& p = w.ptrs[i];
            r.samples[i].reads.overlapRange(w.startPos, w.endPos, p.gs, p.ge, r.cur[i].gs, r.cur[i].ge);
            r.samples[i].bad.overlapRange(w.startPos, w.endPos, p.bs, p.be);
            r.samples[i].broken.matePosRange(w.startPos, w.endPos, p.ks, p.ke);
            w.nReads += p.ge - p.gs;
        }
        }
        r.cur = w.ptrs;                                                     // (the buffers' window pointers now stand on this window)
        if (w.nReads == 0 || (double)w.nReads > o.maxReads) return;
        if ((int)w.vars.size() > o.maxVariants) {
            if (o.skipDifficultWindows) return;
            if (o.filterVarsByCoverage) w.vars = filterVariantsByCoverage(w.vars, o.maxVariants);
        }
        // getFilteredHaplotypes (variantFilter.pyx:377-506)
        const int maxHaplotypes = o.maxHaplotypes - 1;
        const int nVars = (int)w.vars.size();
        const double lg = log2((double)maxHaplotypes);
        if (nVars <= lg || (o.filterVarsByCoverage && o.maxVariants <= lg)) {
            static thread_local std::vector<Hap> haps;                      // (storage reused from window to window of this thread)
            haps.clear();
            SmallVec<int, 8> idx;
            for (int n = 1; n <= nVars; ++n) {                             // itertools.combinations order
                idx.resize((size_t)n);
                for (int i = 0; i < n; ++i) idx[(size_t)i] = i;
                for (;;) {
                    VarList vs;
                    for (int i : idx) vs.push_back(w.vars[(size_t)i]);
                    bool valid; { PROF("s2.pw.valid"); valid = isHaplotypeValid(vs); }
                    if (valid) { PROF("s2.pw.makeHap"); haps.push_back(makeHap(r, w, vs)); }
                    int i = n - 1;
                    while (i >= 0 && idx[(size_t)i] == i + nVars - n) --i;
                    if (i < 0) break;
                    ++idx[(size_t)i];
                    for (int j = i + 1; j < n; ++j) idx[(size_t)j] = idx[(size_t)j - 1] + 1;
                }
            }
            { PROF("s2.pw.finishHaps"); finishHaplotypes(r, w, haps); }
            return;
        }
        // greedy growth of the best haplotypes, one variant at a time (most supported first); the alignments of a step are
        // batched over every such window of the chunk (greedyRounds)
        w.greedy = true;
        w.byCoverage = w.vars;
        std::stable_sort(w.byCoverage.begin(), w.byCoverage.end(), [](const Variant* a, const Variant* b) { return a->nSupportingReads > b->nSupportingReads; });
        w.step = 0;
        // the sampled reads of computeBestScoreForGenotype (variantFilter.pyx:237-283)
        const int windowSize = w.endPos - w.startPos, target = o.coverageSamplingLevel;
        if (windowSize <= 0 || target <= 0) throw WindowError("integer division or modulo by zero");
        w.sampledSeg.assign(1, 0);
        for (size_t i = 0; i < r.samples.size(); ++i) {
            const Ptrs& p = w.ptrs[i];
            const int n = p.ge - p.gs;
            if (n > 0) {
                const int meanCoverage = r.samples[i].reads.rlen(p.gs) * n / windowSize;            // :264
                const int sampleRate = std::max(1, meanCoverage / target);
                for (int q = p.gs; q < p.ge; q += sampleRate) w.sampled.push_back({(int)i, q});
            }
            w.sampledSeg.push_back((int)w.sampled.size());
        }
    }

    // mergeHaplotypes (variantcaller.pyx:325-383) over [reference haplotype] + haps; a window with one haplotype is not called
    void finishHaplotypes(RegionWork& r, WindowWork& w, std::vector<Hap>& haps) {
        static thread_local std::vector<Hap> all;
        all.clear();
        all.reserve(haps.size() + 1);
        Hap ref;
        ref.seq = w.refSeq;
        all.push_back(std::move(ref));
        for (Hap& h : haps) all.push_back(std::move(h));
        SmallVec<size_t, 16> order;
        order.resize(all.size());
        for (size_t i = 0; i < order.size(); ++i) order[i] = i;
        // (a stable sort: insertion sort for the handful of haplotypes a window has -- same order, no scratch buffer)
        if (order.size() <= 16) {
            for (size_t i = 1; i < order.size(); ++i) {
                const size_t x = order[i];
                size_t j = i;
                while (j > 0 && all[x].seq < all[order[j - 1]].seq) { order[j] = order[j - 1]; --j; }
                order[j] = x;
            }
        } else std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return all[a].seq < all[b].seq; });
        std::vector<Hap> merged;
        merged.reserve(all.size());
        int last = -1;
        for (size_t k : order) {
            if (last < 0) { last = (int)k; continue; }
            if (all[k].seq == all[(size_t)last].seq) {
                double p1 = 1.0, p2 = 1.0;
                for (Variant* v : all[(size_t)last].variants) p1 *= calculatePrior(*v, r.fa);
                for (Variant* v : all[k].variants) p2 *= calculatePrior(*v, r.fa);
                if (p2 > p1) last = (int)k;
            } else {
                merged.push_back(std::move(all[(size_t)last]));
                last = (int)k;
            }
        }
        if (last >= 0) merged.push_back(std::move(all[(size_t)last]));
        w.greedy = false;
        if (merged.size() <= 1) { w.live = false; return; }
        w.haps.swap(merged);
        w.live = true;
    }

    // -- the greedy haplotype filter, one variant per round for every window that needs it (variantFilter.pyx:440-506)
    void greedyRounds() {
        std::vector<WindowWork*> todo;
        for (RegionWork* r : regions) for (WindowWork& w : r->windows) if (w.greedy) todo.push_back(&w);
        if (todo.empty()) return;
        { std::lock_guard<std::mutex> g(stMutex); st.n_windows_greedy += (int64_t)todo.size(); }
        const int originalMax = o.originalMaxHaplotypes - 1, maxHaplotypes = o.maxHaplotypes - 1;
        for (;;) {
            std::vector<WindowWork*> active;
            BatchBuilder b;
            b.nInd = 1;
            for (WindowWork* w : todo) {
                if (!w->greedy || w->step >= w->byCoverage.size()) continue;
                RegionWork& r = *regions[(size_t)regionSlot(w->region)];
                try {
                    Variant* tempVar = w->byCoverage[w->step];
                    std::vector<ScoredHap> old = w->heap;
                    std::stable_sort(old.begin(), old.end(), scoredLess);
                    w->cands.clear();
                    w->cands.push_back(VarList{tempVar});
                    for (const ScoredHap& sh : old) {
                        VarList both{tempVar};
                        both.insert(both.end(), sh.vs.begin(), sh.vs.end());
                        std::stable_sort(both.begin(), both.end(), variantLess);
                        if (isHaplotypeValid(both)) w->cands.push_back(both);
                    }
                    if (w->sampled.empty()) {                               // no reads sampled: every score is -1e20
                        for (const VarList& vs : w->cands) makeHap(r, *w, vs);
                        for (const VarList& vs : w->cands) pushScored(*w, ScoredHap{-1e20, vs}, originalMax);
                        ++w->step;
                        active.push_back(nullptr);                          // (keeps the loop going without a device window)
                        continue;
                    }
                    std::vector<std::string> seqs;
                    for (const VarList& vs : w->cands) seqs.push_back(makeHap(r, *w, vs).seq);
                    b.beginWindow(w->hapStart, w->hapEnd, w->endBuf);
                    b.addHap(w->refSeq);
                    for (const std::string& q : seqs) b.addHap(q);
                    for (auto& sq : w->sampled) b.addRead(r.samples[(size_t)sq.first].reads, sq.second, 2);   // alignSingleRead: never skipped
                    b.endSegment(0);
                    b.endWindow();
                    active.push_back(w);
                } catch (const WindowError& e) {
                    logWindowFailure(r.in->chrom, w->startPos, w->endPos, e.what());
                    { std::lock_guard<std::mutex> g(stMutex); ++st.n_windows_failed; }
                    w->greedy = false; w->live = false; w->failed = true;
                }
            }
            if (active.empty()) break;
            if (b.nWindows() > 0) {
                runWindows(s, b, o, false, true);
                int bw = 0;
                for (WindowWork* w : active) {
                    if (!w) continue;
                    const int nH = (int)w->cands.size(), n = (int)w->sampled.size();
                    const double* ll = s.o_loglik.h + b.pairoff[(size_t)bw];
                    for (int k = 0; k < nH; ++k) {
                        const double* row = ll + (size_t)(k + 1) * (size_t)n;
                        double best = -1e20;
                        for (size_t i = 0; i + 1 < w->sampledSeg.size(); ++i) {
                            if (w->sampledSeg[i] == w->sampledSeg[i + 1]) continue;                     // :261-262
                            double score = 0.0;
                            for (int q = w->sampledSeg[i]; q < w->sampledSeg[i + 1]; ++q) score += log(0.5 * (exp(ll[q]) + exp(row[q])));   // :270-272
                            best = std::max(best, score);
                        }
                        pushScored(*w, ScoredHap{best, w->cands[(size_t)k]}, originalMax);
                    }
                    ++w->step;
                    ++bw;
                }
            }
        }
        for (WindowWork* w : todo) {
            if (!w->greedy) continue;
            RegionWork& r = *regions[(size_t)regionSlot(w->region)];
            try {
                std::vector<ScoredHap> best = w->heap;                      // sorted(hapsByBestScore, reverse=True): descending, equal ones keep their order
                std::stable_sort(best.begin(), best.end(), [](const ScoredHap& a, const ScoredHap& b) { return scoredLess(b, a); });
                std::vector<Hap> haps;
                for (size_t i = 0; i < best.size() && (int)i < maxHaplotypes; ++i) haps.push_back(makeHap(r, *w, best[i].vs));
                finishHaplotypes(r, *w, haps);
            } catch (const WindowError& e) {
                logWindowFailure(r.in->chrom, w->startPos, w->endPos, e.what());
                { std::lock_guard<std::mutex> g(stMutex); ++st.n_windows_failed; }
                w->greedy = false; w->live = false; w->failed = true;
            }
        }
    }
    static void pushScored(WindowWork& w, const ScoredHap& item, int originalMax) {
        if ((int)w.heap.size() < originalMax) heapPush(w.heap, item); else heapPushPop(w.heap, item);
    }
    int regionSlot(int regionIndex) const { return regionIndex - regions[0]->index; }

    // -- C..F for a list of windows
    // fromDevice: the windows are those plat_stage_b_batch prepared -- their batch is on the device already (devBatch), w->bw / w->hapBegin are set
    void callWindows(std::vector<WindowWork*>& wins, bool fromDevice = false) {
        if (wins.empty()) return;
        Slot& z = s;
        static thread_local BatchBuilder callBatch;                      // (kept from chunk to chunk of this worker thread: see BatchBuilder::reset)
        BatchBuilder& b = callBatch;
        b.reset(nInd);
        for (WindowWork* w : wins) {
            if (fromDevice) break;
            PROF("s4.build");
            RegionWork& r = *regions[(size_t)regionSlot(w->region)];
            w->bw = b.nWindows();
            b.beginWindow(w->hapStart, w->hapEnd, w->endBuf);
            { PROF("s4.addHap"); for (const Hap& h : w->haps) b.addHap(h.seq); }
            PROF("s4.addReads");
            for (size_t i = 0; i < r.samples.size(); ++i) {                // good -> bad -> brokenMates (chaplotype.pyx:341-373)
                const Ptrs& p = w->ptrs[i];
                b.addReads(r.samples[i].reads, p.gs, p.ge, 0);
                b.addReads(r.samples[i].bad, p.bs, p.be, 1);
                b.addReads(r.samples[i].broken, p.ks, p.ke, 2);
                b.endSegment(p.ge - p.gs);
            }
            b.endWindow();
        }
        DeviceBatch db;
        if (fromDevice) { PROF("s4.runBatch"); db = devBatch; runBatch(z, db, o, true, false); }
        else {
            PROF("s4.runWindows");
            db = runWindows(z, b, o, true, false);
            for (WindowWork* w : wins) w->hapBegin = b.hapbegin[(size_t)w->bw];
        }
        {
            int64_t np = db.nPairs;
            if (fromDevice) { np = 0; for (WindowWork* w : wins) { int nr = 0; for (const Ptrs& p : w->ptrs) nr += (p.ge - p.gs) + (p.be - p.bs) + (p.ke - p.ks); np += (int64_t)w->haps.size() * nr; } }
            std::lock_guard<std::mutex> g(stMutex);
            st.n_pairs += np;                                               // (of the windows CALLED from this batch)
        }
        lap(4);

        // D: distinct variants, masks, priors -> posteriors (Population.computeVariantPosteriors, cpopulation.pyx:596-621)
        std::vector<int32_t> pwin;
        std::vector<int64_t> poff{0};
        std::vector<uint8_t> pmask;
        std::vector<double> pprior;
        for (WindowWork* w : wins) {
            PROF("s5.build");
            RegionWork& r = *regions[(size_t)regionSlot(w->region)];
            w->distinct.clear();
            for (const Hap& h : w->haps)
                for (Variant* v : h.variants) if (!contains(w->distinct, v)) w->distinct.push_back(v);
            for (Variant* v : w->distinct) {
                pwin.push_back(w->bw);
                for (const Hap& h : w->haps) pmask.push_back(contains(h.variants, v) ? 1 : 0);
                poff.push_back((int64_t)pmask.size());
                { PROF("s5.prior"); pprior.push_back(calculatePrior(*v, r.fa)); }
            }
            if (o.outputRefCalls)                                           // pop.calculatePosterior(v, 1) of outputRefCall: the window's candidates under a flat prior
                for (Variant* v : w->vars) {
                    pwin.push_back(w->bw);
                    for (const Hap& h : w->haps) pmask.push_back(contains(h.variants, v) ? 1 : 0);
                    poff.push_back((int64_t)pmask.size());
                    pprior.push_back(0.5);
                }
        }
        const size_t nV = pwin.size();
        if (nV) {
            Layout L;
            L.add(z.p_win, nV); L.add(z.p_off, nV + 1); L.add(z.p_mask, pmask.size()); L.add(z.p_prior, nV);
            L.commit(z, z.a_pin);
            fill(z, z.p_win, pwin); fill(z, z.p_off, poff); fill(z, z.p_mask, pmask); fill(z, z.p_prior, pprior);
            z.p_post.reserve(z.ctx, nV + 1);
            L.upload(z, z.a_pin);
            ck(plat_variant_posterior_batch(z.ctx, (int)nV, nInd, db.maxH, db.hapbegin, db.gloff, db.ngood, z.o_gl.d, z.o_freq.d, z.p_win.d,
                                            z.p_off.d, z.p_mask.d, z.p_prior.d, z.p_post.d, z.stream), "plat_variant_posterior_batch");
            z.down(z.p_post, nV);
            z.sync("posteriors");
        }
        lap(5);
        // varsByPos, INFO variants (getHaplotypeInfo order, vcfutils.pyx:1118-1152), read statistics and call sites of the live windows
        std::vector<int32_t> svw, spos, smin, smax, snadd, snrem, sgb, sge, sbb, sbe, kwin, knvar, kvih, kref;
        std::vector<int64_t> saoff, smoff, kvo{0}, kro{0}, klo{0};
        std::vector<uint8_t> svig;
        std::string sadded;
        int64_t mtot = 0;
        size_t at = 0;
        std::vector<WindowWork*> live;
        sgb.assign((size_t)db.nWindows * (size_t)nInd, 0); sge.assign(sgb.size(), 0); sbb.assign(sgb.size(), 0); sbe.assign(sgb.size(), 0);
        for (WindowWork* w : wins) {
            PROF("s6.build");
            RegionWork& r = *regions[(size_t)regionSlot(w->region)];
            w->called.clear(); w->calledPost.clear(); w->byPos.clear(); w->info.clear();
            w->text.clear(); w->nRecords = 0; w->nRefRecords = 0;
            w->firstFlat = o.outputRefCalls ? (int)(at + w->distinct.size()) : -1;
            for (size_t k = 0; k < w->distinct.size(); ++k, ++at) {
                const double p = z.p_post.h[at];
                if (p >= (double)o.minPosterior) {
                    Variant* v = w->distinct[k];
                    w->called.push_back(v); w->calledPost.push_back(p);
                    bool found = false;
                    for (auto& pv : w->byPos) if (pv.first == v->refPos) { pv.second.push_back(v); found = true; break; }
                    if (!found) w->byPos.push_back({v->refPos, VarList{v}});
                }
            }
            if (o.outputRefCalls) at += w->vars.size();
            // good / bad read ranges of every (window, sample) in the chunk table, for the statistics kernel
            // (indexed by the window's place in the BATCH: a batch the device built also holds windows that are not called from it)
            for (size_t i = 0; i < r.samples.size(); ++i) {
                const Ptrs& p = w->ptrs[i];
                const size_t seg = (size_t)w->bw * (size_t)nInd + i;
                sgb[seg] = (int32_t)(r.samples[i].reads.base + p.gs); sge[seg] = (int32_t)(r.samples[i].reads.base + p.ge);
                sbb[seg] = (int32_t)(r.samples[i].bad.base + p.bs); sbe[seg] = (int32_t)(r.samples[i].bad.base + p.be);
            }
            if (w->called.empty()) continue;
            live.push_back(w);
            const double* freq = z.o_freq.h + w->hapBegin;
            const int32_t* calls = z.o_calls.h + (size_t)w->bw * (size_t)nInd;
            for (size_t h = 0; h < w->haps.size(); ++h) {
                VarList seen;                                               // Haplotype.vcfINFO(): a dictionary over the haplotype's variants
                for (Variant* v : w->haps[h].variants) {
                    if (contains(seen, v)) continue;
                    seen.push_back(v);
                    int ci = -1;
                    for (size_t c = 0; c < w->called.size(); ++c) if (w->called[c]->same(*v)) { ci = (int)c; break; }
                    if (ci < 0) continue;
                    VarInfo* d = nullptr;
                    for (VarInfo& x : w->info) if (x.var->same(*v)) { d = &x; break; }
                    if (!d) {
                        VarInfo n;
                        n.var = v;
                        PROF("s6.hp_sc");
                        n.HP = homopolymerLengthForOneVariant(*v, r.fa);
                        n.SC = getSequenceContext(*v, r.fa);
                        n.PP.clear();
                        append_fixed(n.PP, w->calledPost[(size_t)ci], 0);                   // "%.0f"

                        n.FRsum = freq[h];
                        w->info.push_back(std::move(n));
                    } else d->FRsum += freq[h];
                }
            }
            int ngood = 0;
            for (const Ptrs& p : w->ptrs) ngood += p.ge - p.gs;
            w->firstStatVar = (int)svw.size();
            for (VarInfo& d : w->info) {
                const Variant* v = d.var;
                svw.push_back(w->bw); spos.push_back(v->refPos); smin.push_back(v->bamMinPos); smax.push_back(v->bamMaxPos);
                snadd.push_back(v->nAdded); snrem.push_back(v->nRemoved);
                saoff.push_back((int64_t)sadded.size());
                sadded += v->added;
                for (int i = 0; i < nInd; ++i) {                            // `variant in genotypeCalls[i]` (cgenotype.pyx:98-105)
                    const int g = calls[i];
                    bool in = false;
                    if (g >= 0) {
                        int a = 0, bq = 0, rowlen = (int)w->haps.size(), gg = g;
                        while (gg >= rowlen) { gg -= rowlen; --rowlen; ++a; }
                        bq = a + gg;
                        in = contains(w->haps[(size_t)a].variants, v) || contains(w->haps[(size_t)bq].variants, v);
                    }
                    svig.push_back(in ? 1 : 0);
                }
                smoff.push_back(mtot);
                mtot += std::max(ngood, 1);
            }
            // call sites: varThisPosInHap / haplotypeIsRefAtThisPos per VCF position (vcfutils.pyx:400-426)
            SmallVec<std::pair<int, VarList>*, 8> positions;
            for (auto& pv : w->byPos) positions.push_back(&pv);
            std::sort(positions.begin(), positions.end(), [](const std::pair<int, VarList>* a, const std::pair<int, VarList>* bb) { return a->first < bb->first; });
            w->firstSite = (int)kwin.size();
            for (auto* pv : positions) {
                const int POS = pv->first;
                const VarList& vars = pv->second;
                kwin.push_back(w->bw); knvar.push_back((int32_t)vars.size());
                for (const Hap& h : w->haps) for (Variant* v : vars) kvih.push_back(contains(h.variants, v) ? 1 : 0);
                for (const Hap& h : w->haps) {
                    bool any = false;
                    for (Variant* v : h.variants)
                        if ((contains(vars, v) || contains(w->allVars, v)) && v->minRefPos <= POS && POS <= v->maxRefPos) { any = true; break; }
                    kref.push_back(any ? 0 : 1);
                }
                kvo.push_back((int64_t)kvih.size()); kro.push_back((int64_t)kref.size());
                const int64_t NL = (int64_t)(vars.size() + 1) * (int64_t)(vars.size() + 2) / 2;
                klo.push_back(klo.back() + NL * nInd);
            }
        }
        const std::vector<double> flatPost(z.p_post.h, z.p_post.h + (nV ? nV : 0));     // (p_post's pinned mirror is reused by nothing below, copied for clarity)
        if (!live.empty()) {
        // E: read statistics + per-site genotype calls
        const size_t nSV = svw.size(), nSites = kwin.size();
        kvih.push_back(0);
        Layout L, LO;
        L.add(z.s_vw, nSV); L.add(z.s_pos, nSV); L.add(z.s_min, nSV); L.add(z.s_max, nSV); L.add(z.s_nadd, nSV); L.add(z.s_nrem, nSV); L.add(z.s_aoff, nSV);
        L.add(z.s_moff, nSV); L.add(z.s_vig, svig.size()); L.add(z.s_gb, sgb.size()); L.add(z.s_ge, sge.size()); L.add(z.s_bb, sbb.size()); L.add(z.s_be, sbe.size());
        L.add(z.s_added, sadded.size() + PLAT_BLOB_PAD);
        L.add(z.k_win, nSites); L.add(z.k_nvar, nSites); L.add(z.k_vo, nSites + 1); L.add(z.k_ro, nSites + 1); L.add(z.k_lo, nSites + 1); L.add(z.k_ref, kref.size());
        L.add(z.k_vih, kvih.size());
        L.commit(z, z.a_sin);
        fill(z, z.s_vw, svw); fill(z, z.s_pos, spos); fill(z, z.s_min, smin); fill(z, z.s_max, smax); fill(z, z.s_nadd, snadd); fill(z, z.s_nrem, snrem);
        fill(z, z.s_aoff, saoff); fill(z, z.s_moff, smoff); fill(z, z.s_vig, svig); fill(z, z.s_gb, sgb); fill(z, z.s_ge, sge); fill(z, z.s_bb, sbb); fill(z, z.s_be, sbe);
        memcpy(z.s_added.h, sadded.data(), sadded.size()); memset(z.s_added.h + sadded.size(), 0, PLAT_BLOB_PAD);
        fill(z, z.k_win, kwin); fill(z, z.k_nvar, knvar); fill(z, z.k_vo, kvo); fill(z, z.k_ro, kro); fill(z, z.k_lo, klo); fill(z, z.k_ref, kref); fill(z, z.k_vih, kvih);
        L.upload(z, z.a_sin);
        LO.add(z.s_counts, nSV * 16); LO.add(z.s_ps, nSV * (size_t)nInd * 2); LO.add(z.s_nminq, nSV); LO.add(z.s_minq, (size_t)mtot);
        LO.add(z.k_ph, nSites * (size_t)nInd * 2); LO.add(z.k_lik, (size_t)klo.back()); LO.add(z.k_out4, nSites * (size_t)nInd * 4);
        LO.commit(z, z.a_sout);
        plat_infostats_batch ib;
        memset(&ib, 0, sizeof ib);
        ib.n_vars = (int32_t)nSV; ib.n_ind = nInd;
        ib.var_window = z.s_vw.d; ib.var_pos = z.s_pos.d; ib.var_bam_min = z.s_min.d; ib.var_bam_max = z.s_max.d; ib.var_n_added = z.s_nadd.d;
        ib.var_n_removed = z.s_nrem.d; ib.var_added = z.s_added.d; ib.var_added_off = z.s_aoff.d; ib.var_in_genotype = z.s_vig.d; ib.minq_off = z.s_moff.d;
        ib.good_begin = z.s_gb.d; ib.good_end = z.s_ge.d; ib.bad_begin = z.s_bb.d; ib.bad_end = z.s_be.d;
        ib.read_seq = z.t_seq.d; ib.read_qual = z.t_qual.d; ib.read_off = z.t_off.d; ib.read_pos = z.t_pos.d; ib.read_end = z.t_end.d; ib.read_mapq = z.t_mapq.d;
        ib.read_flags = z.t_flags.d; ib.cigar = z.t_cigar.d; ib.cig_off = z.t_cigoff.d;
        ck(plat_variant_read_stats_batch(z.ctx, &ib, o.badReadsWindow, o.countOnlyExactIndelMatches, z.s_counts.d, z.s_ps.d, z.s_minq.d, z.s_nminq.d, z.stream),
           "plat_variant_read_stats_batch");
        ck(plat_genotype_call_batch(z.ctx, (int)nSites, nInd, db.hapbegin, db.gloff, z.o_gl.d, z.o_gof.d, z.o_freq.d, z.k_win.d, z.k_nvar.d, z.k_vo.d,
                                    z.k_ro.d, z.k_vih.d, z.k_ref.d, z.k_lo.d, z.k_ph.d, z.k_lik.d, z.k_out4.d, z.stream), "plat_genotype_call_batch");
        LO.download(z, z.a_sout);
        z.sync("read statistics / genotype calls");
        }
        lap(6);
        // F: INFO, FILTER, text -- and, with outputRefCalls, the REFCALL lines that belong to a calling window: the blocks between its
        // called positions (:584-603), or one line for the whole window when nothing in it was called (:605-607)
        for (WindowWork* w : wins) {
            RegionWork& r = *regions[(size_t)regionSlot(w->region)];
            try {
                if (!w->called.empty()) {
                    writeWindow(r, *w, klo);
                    if (o.outputRefCalls && w->byPos.size() > 1) refCallBlocksBetween(r, *w);
                } else if (o.outputRefCalls) {
                    double maxPost = 0.0;
                    for (size_t k = 0; k < w->vars.size(); ++k) { const double p = flatPost[(size_t)w->firstFlat + k]; maxPost = k ? std::max(maxPost, p) : p; }
                    if (refCallLine(r, w->text, w->startPos, w->endPos, snapshotNR(w->ptrs), !w->vars.empty(), maxPost)) { ++w->nRecords; ++w->nRefRecords; }
                    else throw WindowError("cannot convert float infinity to integer");
                }
            } catch (const WindowError& e) {
                logWindowFailure(r.in->chrom, w->startPos, w->endPos, e.what());
                std::lock_guard<std::mutex> g(stMutex);
                ++st.n_windows_failed;
            }
        }
        lap(7);
        countCalled(wins.size());                                           // (once per window: a batch that failed half way counted nothing)
    }
    // :584-603: reference-call blocks between the called positions of one window, walked in the order a Python-2 dictionary holds its
    // integer keys (pop.varsByPos.iteritems())
    void refCallBlocksBetween(RegionWork& r, WindowWork& w) {
        std::vector<int> keys;
        for (auto& pv : w.byPos) keys.push_back(pv.first);
        const std::vector<int> order = py2_int_dict_order(keys);
        const VarList* last = nullptr;
        if (o.refCallBlockSize <= 0) throw WindowError("range() arg 3 must not be zero");
        for (size_t index = 0; index < order.size(); ++index) {
            const VarList* these = nullptr;
            for (auto& pv : w.byPos) if (pv.first == order[index]) { these = &pv.second; break; }
            if (index > 0) {
                int lastVarPos = (*last)[0]->maxRefPos, nextVarPos = (*these)[0]->minRefPos;
                for (const Variant* v : *last) lastVarPos = std::max(lastVarPos, v->maxRefPos);
                for (const Variant* v : *these) nextVarPos = std::min(nextVarPos, v->minRefPos);
                nextVarPos += 1;
                if (nextVarPos - lastVarPos > 1)
                    for (int blockStart = lastVarPos + 1; blockStart < nextVarPos; blockStart += o.refCallBlockSize) {
                        const int blockEnd = std::min(blockStart + o.refCallBlockSize, nextVarPos - 1);
                        if (blockStart == blockEnd) continue;
                        try {
                            if (refCallLine(r, w.text, blockStart, blockEnd, snapshotNR(w.ptrs), false, 0.0)) { ++w.nRecords; ++w.nRefRecords; }
                        } catch (const WindowError& e) { logWindowFailure(r.in->chrom, blockStart, blockEnd, e.what()); }
                    }
            }
            last = these;
        }
    }
    void countCalled(size_t n) { std::lock_guard<std::mutex> g(stMutex); st.n_windows_called += (int64_t)n; }

    // vcfINFO (vcfutils.pyx:1226-1460), vcfFILTER (:1502-1627), outputCallToVCF (:338-599), VCF.write_data (vcf.py:710-739)
    void writeWindow(RegionWork& r, WindowWork& w, const std::vector<int64_t>& klo) {
        Slot& z = s;
        const int hapScore = z.o_hapscore.h[w.bw];
        for (size_t k = 0; k < w.info.size(); ++k) {
            VarInfo& d = w.info[k];
            const size_t sv = (size_t)w.firstStatVar + k;
            PROF("text.info");
            infoFieldsFromReadStats(d, z.s_counts.h + 16 * sv, z.s_ps.h + 2 * sv * (size_t)nInd, nInd, z.s_minq.h + z.s_moff.h[sv], z.s_nminq.h[sv]);
            if (d.TR > 0) {                                                // :1400-1409
                const double qual = strtod(d.PP.c_str(), nullptr);
                if (qual > 2500) d.QD = Num::I(o.qdThreshold + 10);
                else d.QD = Num::D((qual + (-10 * log10(calculatePrior(*d.var, r.fa)))) / (double)d.TR);
            } else d.QD = Num::I(0);
            d.FRtext.clear();
            append_fixed(d.FRtext, d.FRsum, 4);                                             // "%1.4f"
            d.HapScore = hapScore;
            d.Source.clear();
            if (d.var->varSource & PLATYPUS_VAR) d.Source.push_back("Platypus");
            if (d.var->varSource & ASSEMBLER_VAR) d.Source.push_back("Assembler");
            if (d.var->varSource & FILE_VAR) d.Source.push_back("File");
            d.filters.clear();
        }
        auto infoOf = [&](const Variant* v) -> VarInfo& {
            for (VarInfo& d : w.info) if (d.var->same(*v)) return d;
            throw WindowError("variant without INFO");
        };
        // vcfFILTER
        for (auto& pv : w.byPos) {
            PROF("text.filter");
            const VarList& varsAtPos = pv.second;
            const int n = (int)varsAtPos.size();
            const bool failsSC = computeSCValue(infoOf(varsAtPos[0]).SC) > o.scThreshold;
            int fQD = 0, fHap = 0, fMQ = 0, fSB = 0, fAB = 0, fMMLQ = 0, bestQual = 0;
            double BRF = 0.0;
            for (Variant* v : varsAtPos) {
                VarInfo& d = infoOf(v);
                d.filters.clear();
                if (failsSC) d.filters.push_back("SC");
                BRF = d.BRF.value();
                bestQual = std::max(bestQual, atoi(d.PP.c_str()));
                fMMLQ += d.MMLQ < o.badReadsThreshold;
                fQD += d.QD.value() < (double)o.qdThreshold;
                fHap += d.HapScore > o.hapScoreThreshold;
                fAB += d.TC > 0 && d.ABPV.value() < o.abThreshold;
                fSB += d.SbPval.value() < o.sbThreshold;
                fMQ += d.MQ.value() < (double)o.rmsmqThreshold;
            }
            for (Variant* v : varsAtPos) {                                  // BRF: of the last variant, as there
                VarInfo& d = infoOf(v);
                if (fQD == n) d.filters.push_back("QD");
                if (fHap == n) d.filters.push_back("HapScore");
                if (fMQ == n) d.filters.push_back("MQ");
                if (fSB == n) d.filters.push_back("strandBias");
                if (fAB == n) d.filters.push_back("alleleBias");
                if (fMMLQ == n || BRF >= o.filteredReadsFrac) d.filters.push_back("badReads");
                if (bestQual < 20) d.filters.push_back("Q20");
            }
        }
        // outputCallToVCF
        SmallVec<std::pair<int, VarList>*, 8> positions;
        for (auto& pv : w.byPos) positions.push_back(&pv);
        std::sort(positions.begin(), positions.end(), [](const std::pair<int, VarList>* a, const std::pair<int, VarList>* b) { return a->first < b->first; });
        std::string& out = w.text;
        out.reserve(out.size() + positions.size() * (size_t)(320 + 40 * nInd));                  // (a record line is ~300 characters: no regrowth on the way)
        for (size_t pi = 0; pi < positions.size(); ++pi) {
            PROF("text.record");
            int POS = positions[pi]->first;
            const VarList& variants = positions[pi]->second;
            const int nVariants = (int)variants.size();
            const size_t site = (size_t)w.firstSite + pi;
            // (the record's lists live from record to record of this thread: their storage is reused)
            static thread_local std::string ref;
            static thread_local std::vector<std::string> alt, linefilter, FR, PP, sampleCols;
            SmallVec<long long, 4> NF, NR, TR;
            linefilter.clear(); FR.clear(); PP.clear(); sampleCols.clear();
            { PROF("text.record.refalt"); refAndAlt(POS, variants, r.fa, ref, alt); }
            VarInfo& lead = infoOf(variants[0]);
            for (Variant* v : variants) {
                VarInfo& d = infoOf(v);
                for (const char* f : d.filters) linefilter.emplace_back(f);
                FR.push_back(d.FRtext); PP.push_back(d.PP); NF.push_back(d.NF); NR.push_back(d.NR); TR.push_back(d.TR);
            }
            int qual = 0;
            bool first = true;
            for (const std::string& pp : PP) { const int q = atoi(pp.c_str()); if (first || q > qual) qual = q; first = false; }
            // per-sample columns
            double maxGof = 0.0;
            int nNonRefCalls = 0;
            const int64_t NL = (int64_t)(nVariants + 1) * (nVariants + 2) / 2;
            for (int i = 0; i < nInd; ++i) {
                PROF("text.record.samplecol");
                const Ptrs& p = w.ptrs[(size_t)i];
                if (p.ge - p.gs == 0) { sampleCols.push_back("./.:0,0,0:0:0:0:0"); continue; }        // :498-500
                const size_t t = site * (size_t)nInd + (size_t)i;
                const int index1 = z.k_ph.h[2 * t], index2 = z.k_ph.h[2 * t + 1];
                const double* lik = z.k_lik.h + klo[site] + (int64_t)i * NL;
                const double gtPost = z.k_out4.h[4 * t], nonRefPost = z.k_out4.h[4 * t + 1], refPost = z.k_out4.h[4 * t + 2], gofValue = z.k_out4.h[4 * t + 3];
                if (!(index1 == 0 && index2 == 0)) ++nNonRefCalls;
                // GT : GL : GOF : GQ : NR : NV, written in place; format_formatdata(key=False) then drops the trailing entries made only
                // of "," and "." -- GT "./." can only be dropped when everything after it is, and the integers after it never are
                std::string col;
                const bool oneVar = nVariants == 1;
                bool noCall = false;
                if (oneVar) {                                               // :524-542, :550-553
                    if (phred(nonRefPost) < o.minPosterior) { if (phred(refPost) < o.minPosterior) noCall = true; else col = "0/0"; }
                    if (infoOf(variants[0]).nReadsPerSample[(size_t)i] < o.minReads) noCall = true;
                }
                if (noCall) col = "./.";
                else if (col.empty()) { append_int(col, index1); col += '/'; append_int(col, index2); }
                col += ':';
                if (oneVar) {
                    double top = lik[0];
                    for (int64_t q = 1; q < NL; ++q) top = std::max(top, lik[q]);
                    for (int64_t q = 0; q < NL; ++q) {                   // (FORMAT fields have no numeric missing value: -1.0 stays -1.0)
                        if (q) col += ',';
                        append_py2_str(col, py2_round2(log10(std::max(lik[q] / top, 1e-300))));
                    }
                } else col += "-1,-1,-1";
                col += ':'; append_int(col, (long long)gofValue);
                col += ':'; append_int(col, phred(gtPost));
                col += ':';
                for (int k = 0; k < nVariants; ++k) { if (k) col += ','; append_int(col, infoOf(variants[(size_t)k]).nReadsPerSample[(size_t)i]); }
                col += ':';
                for (int k = 0; k < nVariants; ++k) { if (k) col += ','; append_int(col, infoOf(variants[(size_t)k]).nVarReadsPerSample[(size_t)i]); }
                sampleCols.push_back(std::move(col));
                maxGof = std::max(maxGof, gofValue);
            }
            const long long MGOF = (long long)py2_round2(maxGof);
            if (!(nNonRefCalls > 0 || o.minPosterior == 0 || o.outputRefCalls == 1)) continue;
            trimLeftPadding(POS, ref, alt);
            bool plain = true;
            for (char c : ref) if (c != 'A' && c != 'C' && c != 'T' && c != 'G') { plain = false; break; }
            if (!plain) continue;                                           // :583-592
            // VCF.write_data
            PROF("text.record.write");
            // (written through a pointer into space reserved for the whole line: a bound on its length first)
            const size_t chromLen = strlen(r.in->chrom);
            size_t bound = chromLen + ref.size() + lead.SC.size() + 768 + 80 * (size_t)nVariants;          // literals 130, 15 numbers of at most 32, 3 counts per variant
            for (const std::string& a : alt) bound += a.size() + 1;
            for (const std::string& f : linefilter) bound += f.size() + 1;
            for (const std::string& c : sampleCols) bound += c.size() + 1;
            for (const std::string& t : FR) bound += t.size() + 1;
            for (const std::string& t : PP) bound += t.size() + 1;
            bound += 32;                                                       // Source: at most Platypus,Assembler,File
            const size_t at0 = out.size();
            out.resize(at0 + bound);
            char* p = &out[at0];
            p = put_chars(p, r.in->chrom, chromLen); *p++ = '\t';
            p = put_int(p, POS + 1); p = put_lit(p, "\t.\t"); p = put_str(p, ref); *p++ = '\t';
            if (alt.empty()) *p++ = '.'; else for (size_t q = 0; q < alt.size(); ++q) { if (q) *p++ = ','; p = put_str(p, alt[q]); }
            *p++ = '\t'; p = put_int(p, qual); *p++ = '\t';
            if (linefilter.empty()) p = put_lit(p, "PASS");
            else {
                std::vector<std::string> flt = py2_set_order(linefilter);
                for (size_t q = 0; q < flt.size(); ++q) { if (q) *p++ = ';'; p = put_str(p, flt[q]); }
            }
            *p++ = '\t';
            auto joinLL = [&p](const SmallVec<long long, 4>& v) { for (size_t q = 0; q < v.size(); ++q) { if (q) *p++ = ','; p = Num::I(v[q]).put(p); } };
            auto joinS = [&p](const std::vector<std::string>& v) { for (size_t q = 0; q < v.size(); ++q) { if (q) *p++ = ','; p = put_str(p, v[q]); } };
            // INFO keys in sorted order: BRF FR HP HapScore MGOF MMLQ MQ NF NR PP QD SC SbPval Source TC TCF TCR TR WE WS
            p = put_lit(p, "BRF="); p = lead.BRF.put(p);
            p = put_lit(p, ";FR="); joinS(FR);
            p = put_lit(p, ";HP="); p = Num::I(lead.HP).put(p);
            p = put_lit(p, ";HapScore="); p = Num::I(lead.HapScore).put(p);
            p = put_lit(p, ";MGOF="); p = Num::I(MGOF).put(p);
            p = put_lit(p, ";MMLQ="); p = Num::I(lead.MMLQ).put(p);
            p = put_lit(p, ";MQ="); p = lead.MQ.put(p);
            p = put_lit(p, ";NF="); joinLL(NF);
            p = put_lit(p, ";NR="); joinLL(NR);
            p = put_lit(p, ";PP="); joinS(PP);
            p = put_lit(p, ";QD="); p = lead.QD.put(p);
            p = put_lit(p, ";SC="); p = put_str(p, lead.SC);
            p = put_lit(p, ";SbPval="); p = lead.SbPval.put(p);
            p = put_lit(p, ";Source="); for (size_t q = 0; q < lead.Source.size(); ++q) { if (q) *p++ = ','; p = put_chars(p, lead.Source[q], strlen(lead.Source[q])); }
            p = put_lit(p, ";TC="); p = Num::I(lead.TC).put(p);
            p = put_lit(p, ";TCF="); p = Num::I(lead.TCF).put(p);
            p = put_lit(p, ";TCR="); p = Num::I(lead.TCR).put(p);
            p = put_lit(p, ";TR="); joinLL(TR);
            p = put_lit(p, ";WE="); p = Num::I(w.endPos).put(p);
            p = put_lit(p, ";WS="); p = Num::I(w.startPos).put(p);
            p = put_lit(p, "\tGT:GL:GOF:GQ:NR:NV");
            for (const std::string& c : sampleCols) { *p++ = '\t'; p = put_str(p, c); }
            *p++ = '\n';
            out.resize((size_t)(p - out.data()));
            ++w.nRecords;
        }
    }

    double stage[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stageWait[8] = {0, 0, 0, 0, 0, 0, 0, 0}, waitMark = 0;
    Clock::time_point mark;
    void lap(int k) { const auto now = Clock::now(); stage[k] += secs(mark, now); mark = now; stageWait[k] += s.t_wait - waitMark; waitMark = s.t_wait; }

    void run() {
        const auto t0 = Clock::now();
        double wait0 = s.t_wait;
        mark = t0; waitMark = wait0;
        uploadReads();
        lap(0);
        deviceB = eligibleDeviceB();
        if (o.getVariantsFromBAMs) scanCandidates();
        assembleTiles();
        lap(1);
        if (deviceB) stageBFromDevice();
        else {
            int scan0 = 0;
            for (RegionWork* r : regions) {
                { PROF("s2.regionVariants"); regionVariants(*r, scan0); }
                scan0 += (int)r->samples.size();
                PROF("s2.regionWindows");
                regionWindows(*r);
            }
        }
        lap(2);
        greedyRounds();
        lap(3);
        std::vector<WindowWork*> wins, devWins;
        int64_t nWin = 0, nVar = 0, nCand = 0;
        for (RegionWork* r : regions) {
            nVar += (int64_t)r->variants.size(); nCand += r->nCandRecords;
            for (WindowWork& w : r->windows) if (w.live) { ++nWin; (w.onDevice ? devWins : wins).push_back(&w); }
        }
        if (!devWins.empty()) {
            try {
                callWindows(devWins, true);
            } catch (const DeviceError& e) {
                // a window the device refuses takes the batch with it: the batch's windows are prepared again by the host's code and go
                // through the per-window retry below with the others
                if (!windowClassError(e.code)) throw;
                for (WindowWork* w : devWins) {
                    RegionWork& r = *regions[(size_t)regionSlot(w->region)];
                    w->text.clear(); w->nRecords = 0; w->nRefRecords = 0; w->onDevice = false; w->haps.clear(); w->live = false;
                    try { prepareWindow(r, *w); }
                    catch (const WindowError& e2) {
                        logWindowFailure(r.in->chrom, w->startPos, w->endPos, e2.what());
                        std::lock_guard<std::mutex> g(stMutex);
                        ++st.n_windows_failed;
                        w->live = false; w->greedy = false; w->failed = true;
                    }
                }
                greedyRounds();
                wins.clear();
                for (RegionWork* r : regions) for (WindowWork& w : r->windows) if (w.live && !w.onDevice) wins.push_back(&w);
            }
        }
        try {
            callWindows(wins);
        } catch (const DeviceError& e) {
            // Only what a single WINDOW can be guilty of is retried: one window the device refuses (bad input, a haplotype too long or
            // too short, a size that overflows) would take every other window of the chunk with it, so they are called one at a time and
            // only the failing ones are skipped (what the reference's per-window try/except does, variantcaller.pyx:568-615).  A failing
            // runtime, an exhausted device or a lost GPU is nobody's window: it ends plat_call_regions with that error.
            if (!windowClassError(e.code)) throw;
            for (WindowWork* w : wins) { w->text.clear(); w->nRecords = 0; w->nRefRecords = 0; }
            for (WindowWork* w : wins) {
                std::vector<WindowWork*> one{w};
                try { callWindows(one); }
                catch (const DeviceError& e2) {
                    if (!windowClassError(e2.code)) throw;
                    w->text.clear(); w->nRecords = 0; w->nRefRecords = 0;
                    logWindowFailure(regions[(size_t)regionSlot(w->region)]->in->chrom, w->startPos, w->endPos, e2.what());
                    std::lock_guard<std::mutex> g(stMutex);
                    ++st.n_windows_failed;
                }
            }
        }
        // the region's text: what the loop writes, in the order it writes it
        int64_t nRec = 0, nRef = 0;
        for (RegionWork* r : regions) {
            int nHapLast = 0;                                               // haplotypes of the last window set up in this region (Population.nHaplotypes)
            for (Item& it : r->items) {
                if (it.kind == 1) { r->text += it.text; nRec += it.nRef; nRef += it.nRef; continue; }
                WindowWork& w = r->windows[(size_t)it.window];
                if (w.failed) continue;
                if (w.live) { r->text += w.text; nRec += w.nRecords; nRef += w.nRefRecords; nHapLast = (int)w.haps.size(); continue; }
                if (!o.outputRefCalls) continue;
                // a window the loop left without calling (no reads, too many, one haplotype): outputRefCall on a Population that was reset
                // and not set up for it.  Its haplotype list is empty but it still holds the haplotype COUNT of the last window it was
                // set up for, so calculatePosterior's loop raises (logged, skipped) -- unless it never was set up in this region
                try {
                    bool ok;
                    if (w.vars.empty()) ok = refCallLine(*r, r->text, w.startPos, w.endPos, snapshotNR(w.ptrs), false, 0.0);
                    else {
                        // (minCov == 0 decides before the posterior is asked for)
                        if (nHapLast > 0 && !coverageHasAZero(*r, w.startPos, w.endPos)) throw WindowError("list index out of range");
                        const double prior = 0.5;
                        const double post = py2_round0(-10.0 * (log10(1.0 * (1.0 - prior)) - log10(prior + 1.0 * (1.0 - prior))));
                        ok = refCallLine(*r, r->text, w.startPos, w.endPos, snapshotNR(w.ptrs), true, post);
                    }
                    if (ok) { ++nRec; ++nRef; }
                    else throw WindowError("cannot convert float infinity to integer");
                } catch (const WindowError& e) {
                    logWindowFailure(r->in->chrom, w.startPos, w.endPos, e.what());
                    std::lock_guard<std::mutex> g(stMutex);
                    ++st.n_windows_failed;
                }
            }
            r->release();
        }
        const double total = secs(t0, Clock::now()), waited = s.t_wait - wait0;
        std::lock_guard<std::mutex> g(stMutex);
        st.n_windows += nWin; st.n_variants += nVar; st.n_candidate_records += nCand; st.n_records += nRec; st.n_refcall_records += nRef;
        st.seconds_host += total - waited; st.seconds_device_wait += waited;
        for (int k = 0; k < 8; ++k) { st.seconds_stage[k] += stage[k]; g_stageWait[k] += stageWait[k]; }
        g_stageWait[8] += total - waited; g_stageWait[9] += waited;
    }
    bool coverageHasAZero(const RegionWork& r, int windowStart, int windowEnd) const {
        for (const SampleView& sv : r.samples) {
            const TableView& tv = sv.reads;
            const int N = tv.n();
            if (N == 0) return windowStart < windowEnd;
            for (int p = windowStart; p < windowEnd; ++p) {
                int s0 = TableView::lowerBound(tv.t->pos, N, std::max<int64_t>(1, (int64_t)p - tv.longest));
                const int e0 = TableView::lowerBound(tv.t->pos, N, (int64_t)p + 1);
                while (s0 < N && tv.t->end[s0] <= p) ++s0;
                if (std::min(e0, N) - s0 <= 0) return true;
            }
        }
        return false;
    }
};

}  // namespace plathost

using namespace plathost;

struct plat_caller {
    int device = 0, nWorkers = 1, regionsPerChunk = 4;
    bool countCells = false;
    std::vector<std::unique_ptr<Slot>> slots;
    std::string lastError;
};

CALLER_EXPORT void plat_caller_default_options(plat_caller_options* o) {
    if (!o) return;
    memset(o, 0, sizeof *o);
    o->rlen = 150; o->minReads = 2; o->maxReads = 5000000; o->maxSize = 1500; o->largeWindows = 0; o->maxVariants = 8; o->coverageSamplingLevel = 30;
    o->maxHaplotypes = 50; o->originalMaxHaplotypes = 50; o->skipDifficultWindows = 0; o->getVariantsFromBAMs = 1; o->genSNPs = 1; o->genIndels = 1;
    o->mergeClusteredVariants = 1; o->minFlank = 10; o->filterVarsByCoverage = 1; o->filteredReadsFrac = 0.7; o->maxVarDist = 15; o->minVarDist = 9;
    o->useEMLikelihoods = 0; o->countOnlyExactIndelMatches = 0; o->calculateFlankScore = 0; o->assemble = 0; o->outputRefCalls = 0; o->minMapQual = 20;
    o->minBaseQual = 20; o->minPosterior = 5; o->sbThreshold = 1e-3; o->scThreshold = 0.95; o->abThreshold = 1e-3; o->minVarFreq = 0.05;
    o->badReadsWindow = 11; o->badReadsThreshold = 15; o->rmsmqThreshold = 40; o->qdThreshold = 10; o->hapScoreThreshold = 4;
    o->refCallBlockSize = 1000; o->assemblyRegionSize = 1500; o->assembleAll = 1; o->assembleBadReads = 1; o->assembleBrokenPairs = 0; o->assemblerKmerSize = 15;
    o->noCycles = 0;
}

CALLER_EXPORT int plat_caller_create(int device, int n_workers, int regions_per_chunk, plat_caller** out) {
    if (!out || n_workers < 1) return PLAT_ERR_INVALID;
    *out = nullptr;
    std::unique_ptr<plat_caller> c(new plat_caller());
    c->device = device; c->nWorkers = n_workers; c->regionsPerChunk = regions_per_chunk > 0 ? regions_per_chunk : 4;
    for (int i = 0; i < n_workers; ++i) {
        std::unique_ptr<Slot> s(new Slot());
        int rc = plat_ctx_create(device, &s->ctx);
        if (rc == PLAT_OK) rc = plat_stream_create(s->ctx, &s->stream);
        if (rc != PLAT_OK) {
            if (s->ctx) plat_ctx_destroy(s->ctx);
            for (auto& q : c->slots) { plat_stream_destroy(q->ctx, q->stream); plat_ctx_destroy(q->ctx); }
            return rc;
        }
        c->slots.push_back(std::move(s));
    }
    *out = c.release();
    return PLAT_OK;
}

template <class... S> static void releaseAll(plat_ctx* ctx, S&... s) { (void)std::initializer_list<int>{(s.release(ctx), 0)...}; }

CALLER_EXPORT int plat_caller_count_cells(plat_caller* c, int on) {
    if (!c) return PLAT_ERR_INVALID;
    c->countCells = on != 0;
    return PLAT_OK;
}

CALLER_EXPORT int plat_caller_destroy(plat_caller* c) {
    if (!c) return PLAT_ERR_INVALID;
    for (auto& q : c->slots) {
        Slot& z = *q;
        releaseAll(z.ctx, z.t_seq, z.t_qual, z.t_mapq, z.t_off, z.t_pos, z.t_end, z.t_flags, z.t_cigoff, z.t_region, z.t_cigar, z.c_ref, z.c_refoff, z.c_rss,
                   z.c_clen, z.c_rec, z.c_cnt, z.c_status, z.w_hapbegin, z.w_readbegin, z.w_start, z.w_end, z.w_flank, z.w_segbegin, z.w_ngood, z.w_src, z.g_pos,
                   z.g_end, z.g_flags, z.o_calls, z.o_iters, z.o_hapscore, z.o_score, z.w_pairoff, z.w_hapoff, z.w_readoff, z.w_gloff, z.w_hapseq, z.w_kind, z.g_seq,
                   z.g_qual, z.g_mapq, z.o_loglik, z.o_gl, z.o_logl, z.o_gof, z.o_freq, z.o_em, z.p_win, z.s_vw, z.s_pos, z.s_min, z.s_max, z.s_nadd, z.s_nrem,
                   z.s_gb, z.s_ge, z.s_bb, z.s_be, z.s_ps, z.s_minq, z.s_nminq, z.k_win, z.k_nvar, z.k_vih, z.k_ref, z.k_ph, z.p_off, z.s_aoff, z.s_moff, z.s_counts,
                   z.k_vo, z.k_ro, z.k_lo, z.p_mask, z.s_added, z.s_vig, z.p_prior, z.p_post, z.k_lik, z.k_out4, z.t_pack, z.as_seq, z.as_qual, z.as_mapq, z.as_pos, z.as_end, z.as_flags, z.a_asin, z.a_asout, z.a_tab, z.a_cin, z.a_cout, z.a_mout, z.c_scanbegin, z.c_scanlongest, z.m_cand, z.m_n, z.a_win, z.a_wout,
                   z.a_pin, z.a_sin, z.a_sout, z.a_bin, z.a_bout, z.a_desc, z.d_hapbegin, z.d_readbegin, z.d_start, z.d_end, z.d_flank, z.d_segbegin, z.d_ngood, z.d_src,
                   z.d_scratch, z.d_pairoff, z.d_gloff, z.d_hapoff, z.d_readoff, z.d_hapseq, z.d_kind);
        plat_stream_destroy(z.ctx, z.stream);
        plat_ctx_destroy(z.ctx);
    }
    delete c;
    return PLAT_OK;
}

CALLER_EXPORT const char* plat_caller_last_error(const plat_caller* c) { return c ? c->lastError.c_str() : ""; }
CALLER_EXPORT void plat_caller_free(void* p) { free(p); }

// ---- where the chunks of a call come from -----------------------------------------------------------------------------------------------
// (a worker asks for its next chunk of regions, calls it, and hands it back)
struct Feed {
    virtual bool next(std::vector<RegionWork*>& out) = 0;                  // false: no more chunks (or the call has failed)
    virtual void done(const std::vector<RegionWork*>& chunk) = 0;
    virtual ~Feed() {}
};

static std::unique_ptr<RegionWork> makeRegionWork(const plat_region* in, int index, int n_samples, int& longestOut) {
    std::unique_ptr<RegionWork> r(new RegionWork());
    r->in = in; r->index = index;
    r->fa.seq = in->contig_seq; r->fa.len = in->contig_len;
    r->samples.resize((size_t)n_samples);
    int longest = 0;
    for (int i = 0; i < n_samples; ++i) {
        const plat_sample_reads& sr = in->samples[i];
        SampleView& sv = r->samples[(size_t)i];
        sv.reads.t = &sr.reads; sv.bad.t = &sr.bad_reads; sv.broken.t = &sr.broken_mates;
        sv.reads.longest = longestRead(sr.reads); sv.bad.longest = longestRead(sr.bad_reads); sv.broken.longest = longestRead(sr.broken_mates);
        sv.reads.maxLen = mostBases(sr.reads); sv.bad.maxLen = mostBases(sr.bad_reads); sv.broken.maxLen = mostBases(sr.broken_mates);
        longest = std::max(longest, sv.reads.longest);
    }
    longestOut = longest;
    return r;
}
// options.rlen follows the longest read of each region and is kept from the region before when a region has no reads (variantcaller.pyx:476-488)
static inline int nextRlen(int rlen, int longest, int maxSize, int fromBams = 1) { return (fromBams && longest > 0) ? (longest >= maxSize ? maxSize : longest) : rlen; }

// every region already in memory (plat_call_regions)
struct MemoryFeed : Feed {
    std::vector<std::unique_ptr<RegionWork>>& work;
    int per, nChunks;
    std::atomic<int> nextChunk{0};
    std::atomic<bool>& failed;
    MemoryFeed(std::vector<std::unique_ptr<RegionWork>>& w, int per_, std::atomic<bool>& f) : work(w), per(per_), nChunks(((int)w.size() + per_ - 1) / per_), failed(f) {}
    bool next(std::vector<RegionWork*>& out) override {
        if (failed.load()) return false;
        const int ch = nextChunk.fetch_add(1);
        if (ch >= nChunks) return false;
        out.clear();
        for (int k = ch * per; k < std::min((int)work.size(), (ch + 1) * per); ++k) out.push_back(work[(size_t)k].get());
        return true;
    }
    void done(const std::vector<RegionWork*>&) override {}
};

// regions loaded on demand by loader threads into a bounded set of slots (plat_call_regions_stream)
struct StreamFeed : Feed {
    int n, nSamples, per, nChunks, maxSize, fromBams = 1;
    plat_region_load_fn load; void* user;
    std::mutex m;
    std::condition_variable cvLoaded, cvSlot;
    std::vector<int> freeSlots;
    std::vector<std::unique_ptr<RegionWork>> work;
    std::vector<plat_region> desc;
    std::vector<int> slotOf, longest;
    std::vector<char> loaded;
    int nextToLoad = 0, nextChunk = 0, rlen, error = PLAT_OK;
    std::string errText;
    std::atomic<bool>& failed;
    double tLoad = 0, tWait = 0;
    StreamFeed(int n_, int nS, int per_, int maxSize_, int rlen0, plat_region_load_fn l, void* u, int nSlots, std::atomic<bool>& f)
        : n(n_), nSamples(nS), per(per_), nChunks((n_ + per_ - 1) / per_), maxSize(maxSize_), load(l), user(u), work((size_t)n_), desc((size_t)n_),
          slotOf((size_t)n_, -1), longest((size_t)n_, 0), loaded((size_t)n_, 0), rlen(rlen0), failed(f) {
        for (int k = nSlots - 1; k >= 0; --k) freeSlots.push_back(k);
    }
    void fail(int code, const std::string& what) {
        std::lock_guard<std::mutex> g(m);
        if (error == PLAT_OK) { error = code; errText = what; }
        failed.store(true);
        cvLoaded.notify_all(); cvSlot.notify_all();
    }
    void loader() {
        for (;;) {
            int idx, slot;
            {
                std::unique_lock<std::mutex> g(m);
                cvSlot.wait(g, [&] { return !freeSlots.empty() || nextToLoad >= n || failed.load(); });
                if (nextToLoad >= n || failed.load()) return;
                slot = freeSlots.back(); freeSlots.pop_back();             // slot first, index second, under one lock: slots are held in index order
                idx = nextToLoad++;
            }
            const auto t0 = Clock::now();
            memset(&desc[(size_t)idx], 0, sizeof(plat_region));
            const int rc = load(user, idx, slot, &desc[(size_t)idx]);
            if (rc != PLAT_OK) { fail(rc, "the region source failed for region " + std::to_string(idx)); return; }
            int lg = 0;
            std::unique_ptr<RegionWork> r;
            try { r = makeRegionWork(&desc[(size_t)idx], idx, nSamples, lg); }
            catch (const std::exception& e) { fail(PLAT_ERR_BAD_INPUT, e.what()); return; }
            const double dt = secs(t0, Clock::now());
            std::lock_guard<std::mutex> g(m);
            work[(size_t)idx] = std::move(r); slotOf[(size_t)idx] = slot; longest[(size_t)idx] = lg; loaded[(size_t)idx] = 1;
            tLoad += dt;
            cvLoaded.notify_all();
        }
    }
    bool next(std::vector<RegionWork*>& out) override {
        const auto t0 = Clock::now();
        std::unique_lock<std::mutex> g(m);
        for (;;) {
            if (failed.load() || nextChunk >= nChunks) return false;
            const int ch = nextChunk, a = ch * per, b = std::min(n, (ch + 1) * per);
            bool all = true;
            for (int k = a; k < b; ++k) all = all && loaded[(size_t)k];
            if (all) {
                out.clear();
                for (int k = a; k < b; ++k) {                                // list order: rlen walks the regions as the reference's loop does
                    rlen = nextRlen(rlen, longest[(size_t)k], maxSize, fromBams);
                    work[(size_t)k]->rlen = rlen;
                    out.push_back(work[(size_t)k].get());
                }
                ++nextChunk;
                tWait += secs(t0, Clock::now());
                return true;
            }
            cvLoaded.wait(g);
        }
    }
    void done(const std::vector<RegionWork*>& chunk) override {
        std::lock_guard<std::mutex> g(m);
        for (RegionWork* r : chunk) { freeSlots.push_back(slotOf[(size_t)r->index]); slotOf[(size_t)r->index] = -1; }
        cvSlot.notify_all();
    }
};

// the workers of one call: every worker thread (own plat_ctx + stream) pulls chunks from the feed until it runs dry
static int runWorkers(plat_caller* c, Feed& feed, std::atomic<bool>& failed, const Options& o, int n_samples, const char* const* sample_names,
                      plat_caller_stats& st, int nThreads)
{
    std::mutex stMutex, errMutex;
    int firstError = PLAT_OK;
    std::string errText;
    auto worker = [&](Slot* slot) {
        std::vector<RegionWork*> regs;
        while (feed.next(regs)) {
            Chunk chunk{*slot, o, n_samples, sample_names, regs, st, stMutex};
            try {
                chunk.run();
            } catch (const DeviceError& e) {
                std::lock_guard<std::mutex> g(errMutex);
                if (firstError == PLAT_OK) { firstError = e.code; errText = e.what(); }
                failed.store(true);
            } catch (const std::exception& e) {
                std::lock_guard<std::mutex> g(errMutex);
                if (firstError == PLAT_OK) { firstError = PLAT_ERR_BAD_INPUT; errText = e.what(); }
                failed.store(true);
            }
            feed.done(regs);
        }
    };
    nThreads = std::max(1, std::min<int>((int)c->slots.size(), nThreads));
    for (auto& q : c->slots) {
        q->countCells = c->countCells; q->nDpRef = q->cellsRef = q->nDpRun = q->cellsRun = 0;
        q->nAlign = q->alignHapBytes = q->alignReadBytes = q->alignReads = q->alignDpBytes = 0; q->secSeed = q->secDp = q->secSweep = q->secPairs = 0.0;
    }
    std::vector<std::thread> threads;
    for (int i = 1; i < nThreads; ++i) threads.emplace_back(worker, c->slots[(size_t)i].get());
    worker(c->slots[0].get());
    for (std::thread& t : threads) t.join();
    for (auto& q : c->slots) {
        st.n_dp_reference += q->nDpRef; st.cells_reference += q->cellsRef; st.n_dp_launched += q->nDpRun; st.cells_launched += q->cellsRun;
        st.n_align_batches += q->nAlign; st.align_hap_bytes += q->alignHapBytes; st.align_read_bytes += q->alignReadBytes; st.align_reads += q->alignReads;
        st.align_dp_bytes += q->alignDpBytes; st.seconds_kernel_seed += q->secSeed; st.seconds_kernel_dp += q->secDp;
        st.seconds_kernel_sweep += q->secSweep; st.seconds_kernel_pairs += q->secPairs;
    }
    if (firstError != PLAT_OK) c->lastError = errText;
    return firstError;
}

// pieces[i] -> out + offset[i], on a few threads when there is enough to move (a whole-genome share is ~100 MB of record text: one thread
// would spend as long on first-touch page faults of the fresh block as on the copy)
static void copyPieces(char* out, const std::vector<const char*>& from, const std::vector<size_t>& len, const std::vector<size_t>& at) {
    size_t total = 0;
    for (size_t l : len) total += l;
    const size_t n = from.size();
    const int nT = total < ((size_t)8 << 20) ? 1 : (int)std::min<size_t>(8, std::max<size_t>(1, std::thread::hardware_concurrency()));
    auto part = [&](int t) {
        // thread t takes the pieces whose bytes start in its slice of the output
        const size_t span = n ? at[n - 1] + len[n - 1] + 1 : 0;            // (offsets may leave gaps between the pieces: they run over the output, not over the bytes copied)
        const size_t lo = span / (size_t)nT * (size_t)t, hi = t == nT - 1 ? span + 1 : span / (size_t)nT * (size_t)(t + 1);
        size_t i = (size_t)(std::lower_bound(at.begin(), at.end(), lo) - at.begin());
        for (; i < n && at[i] < hi; ++i) if (len[i]) memcpy(out + at[i], from[i], len[i]);
    };
    if (nT == 1) { part(0); return; }
    std::vector<std::thread> th;
    for (int t = 1; t < nT; ++t) th.emplace_back(part, t);
    part(0);
    for (std::thread& x : th) x.join();
}

static int finishText(std::vector<std::unique_ptr<RegionWork>>& work, char** out_text, size_t* out_len) {
    std::vector<const char*> from;
    std::vector<size_t> len, at;
    size_t total = 0;
    for (auto& r : work) if (r) { from.push_back(r->text.data()); len.push_back(r->text.size()); at.push_back(total); total += r->text.size(); }
    char* text = (char*)malloc(total + 1);
    if (!text) return PLAT_ERR_NOMEM;
    copyPieces(text, from, len, at);
    text[total] = 0;
    *out_text = text; *out_len = total;
    return PLAT_OK;
}

static int checkCallArgs(plat_caller* c, const plat_caller_options* options, char** out_text, size_t* out_len, int n_regions, int n_samples) {
    if (!c || !options || !out_text || !out_len || n_regions < 0 || n_samples < 1) return PLAT_ERR_INVALID;
    *out_text = nullptr; *out_len = 0;
    if (!options->getVariantsFromBAMs && !options->assemble) {
        // (the reference then has no candidates at all unless a source VCF is given, which is not built)
        c->lastError = "getVariantsFromBAMs=0 without assemble=1 leaves no candidate source (source VCFs are not built)";
        return PLAT_ERR_UNSUPPORTED;
    }
    return PLAT_OK;
}

CALLER_EXPORT int plat_call_regions(plat_caller* c, const plat_region* regions, int n_regions, int n_samples, const char* const* sample_names,
                                    plat_caller_options* options, char** out_text, size_t* out_len, plat_caller_stats* stats)
{
    int rc = checkCallArgs(c, options, out_text, out_len, n_regions, n_samples);
    if (rc != PLAT_OK) return rc;
    if (n_regions > 0 && !regions) return PLAT_ERR_INVALID;
    const auto t0 = Clock::now();
    plat_caller_stats st;
    memset(&st, 0, sizeof st);
    st.n_regions = n_regions;
    Options o;
    static_cast<plat_caller_options&>(o) = *options;
    std::vector<std::unique_ptr<RegionWork>> work;
    int rlen = options->rlen;
    for (int k = 0; k < n_regions; ++k) {
        int longest = 0;
        std::unique_ptr<RegionWork> r = makeRegionWork(&regions[k], k, n_samples, longest);
        rlen = nextRlen(rlen, longest, options->maxSize, options->getVariantsFromBAMs);
        r->rlen = rlen;
        work.push_back(std::move(r));
    }
    std::atomic<bool> failed(false);
    MemoryFeed feed(work, c->regionsPerChunk, failed);
    rc = runWorkers(c, feed, failed, o, n_samples, sample_names, st, std::max(1, feed.nChunks));
    if (rc != PLAT_OK) return rc;
    if ((rc = finishText(work, out_text, out_len)) != PLAT_OK) return rc;
    options->rlen = rlen;
    st.seconds_total = secs(t0, Clock::now());
    traceStages(st);
    if (stats) *stats = st;
    return PLAT_OK;
}

CALLER_EXPORT int plat_call_regions_stream(plat_caller* c, int n_regions, int n_samples, const char* const* sample_names, plat_caller_options* options,
                                           plat_region_load_fn load, void* user, int n_slots, int n_loader_threads, char** out_text, size_t* out_len,
                                           plat_caller_stats* stats)
{
    int rc = checkCallArgs(c, options, out_text, out_len, n_regions, n_samples);
    if (rc != PLAT_OK) return rc;
    const int per = c->regionsPerChunk, nWorkers = (int)c->slots.size();
    if (!load || n_loader_threads < 1) return PLAT_ERR_INVALID;
    if (n_slots < per * (std::min(nWorkers, std::max(1, (n_regions + per - 1) / per)) + 1) && n_slots < n_regions) {
        c->lastError = "plat_call_regions_stream: n_slots must be at least regions_per_chunk * (n_workers + 1)";
        return PLAT_ERR_INVALID;
    }
    const auto t0 = Clock::now();
    plat_caller_stats st;
    memset(&st, 0, sizeof st);
    st.n_regions = n_regions;
    Options o;
    static_cast<plat_caller_options&>(o) = *options;
    std::atomic<bool> failed(false);
    StreamFeed feed(n_regions, n_samples, per, options->maxSize, options->rlen, load, user, n_slots, failed);
    feed.fromBams = options->getVariantsFromBAMs;
    std::vector<std::thread> loaders;
    for (int i = 0; i < std::min(n_loader_threads, std::max(1, n_regions)); ++i) loaders.emplace_back([&feed] { feed.loader(); });
    rc = runWorkers(c, feed, failed, o, n_samples, sample_names, st, std::max(1, feed.nChunks));
    { std::lock_guard<std::mutex> g(feed.m); feed.cvSlot.notify_all(); }
    if (rc != PLAT_OK) feed.fail(rc, c->lastError);                       // (wakes loaders that wait for a slot)
    for (std::thread& t : loaders) t.join();
    if (rc == PLAT_OK && feed.error != PLAT_OK) { rc = feed.error; c->lastError = feed.errText; }
    if (rc != PLAT_OK) return rc;
    if ((rc = finishText(feed.work, out_text, out_len)) != PLAT_OK) return rc;
    options->rlen = feed.rlen;
    st.seconds_total = secs(t0, Clock::now());
    st.seconds_load = feed.tLoad; st.seconds_source_wait = feed.tWait;
    traceStages(st);
    if (stats) *stats = st;
    return PLAT_OK;
}

// ---- runner.py:301-352: the merge of the per-process record texts by (chromosome key, position) ------------------------------------------------
namespace plathost {
struct MergeKey { int kind; long long num; const char* name; size_t nameLen; long long pos; };
// runner.py:47-50: int(chrom.upper().strip("CHR")) if it is one, else the name itself; integers sort before names (Python 2 orders int < str)
static MergeKey mergeKeyOf(const char* line, const char* end) {
    const char* t1 = (const char*)memchr(line, '\t', (size_t)(end - line));
    MergeKey k{1, 0, line, t1 ? (size_t)(t1 - line) : (size_t)(end - line), 0};
    if (t1) {
        const char* p = t1 + 1;
        long long v = 0;
        while (p < end && *p >= '0' && *p <= '9') v = v * 10 + (*p++ - '0');
        k.pos = v - 1;                                                     // (record positions are 1-based in the text: runner.py:77-82 keys on int(cols[1]))
    }
    const char* a = line; const char* b = line + k.nameLen;
    auto strip = [](char c) { c = (char)toupper((unsigned char)c); return c == 'C' || c == 'H' || c == 'R'; };
    while (a < b && strip(*a)) ++a;
    while (b > a && strip(b[-1])) --b;
    if (a < b) {
        const char* p = a;
        bool neg = false;
        if (*p == '+' || *p == '-') { neg = *p == '-'; ++p; }
        bool digits = p < b;
        long long v = 0;
        for (const char* q = p; q < b; ++q) { if (*q < '0' || *q > '9') { digits = false; break; } v = v * 10 + (*q - '0'); }
        if (digits) { k.kind = 0; k.num = neg ? -v : v; }
    }
    return k;
}
static bool mergeLess(const MergeKey& a, const MergeKey& b) {
    if (a.kind != b.kind) return a.kind < b.kind;
    if (a.kind == 0) { if (a.num != b.num) return a.num < b.num; }
    else {
        const int c = memcmp(a.name, b.name, std::min(a.nameLen, b.nameLen));
        if (c != 0) return c < 0;
        if (a.nameLen != b.nameLen) return a.nameLen < b.nameLen;
    }
    return a.pos < b.pos;
}
}  // namespace plathost

// Three passes: (1) the record lines of every text and their keys, texts cut into slices at line ends and the slices scanned on a few
// threads; (2) the merge itself over the keys -- a run of lines of one text that stays in front of every other text's head is one block;
// ties between texts go to the text that comes first (heapq of (key, index) pairs there); (3) the blocks copied into place, again on a
// few threads.
CALLER_EXPORT int plat_merge_record_texts(const char* const* texts, const size_t* lengths, int n, char** out_text, size_t* out_len) {
    if (!out_text || !out_len || n < 0 || (n > 0 && (!texts || !lengths))) return PLAT_ERR_INVALID;
    struct Line { const char* p; const char* eol; MergeKey key; };
    size_t total = 0;
    for (int i = 0; i < n; ++i) total += lengths[i] + 1;
    // (1)
    struct Slice { int text; const char* a; const char* b; std::vector<Line> lines; };
    std::vector<Slice> slices;
    const size_t sliceBytes = (size_t)4 << 20;
    for (int i = 0; i < n; ++i) {
        const char* p = texts[i];
        const char* end = texts[i] + lengths[i];
        while (p < end) {
            const char* q = (size_t)(end - p) > sliceBytes ? p + sliceBytes : end;
            if (q < end) { const char* e = (const char*)memchr(q, '\n', (size_t)(end - q)); q = e ? e + 1 : end; }
            slices.push_back(Slice{i, p, q, {}});
            p = q;
        }
    }
    auto scan = [](Slice& sl) {
        const char* p = sl.a;
        while (p < sl.b) {
            const char* e = (const char*)memchr(p, '\n', (size_t)(sl.b - p));
            const char* eol = e ? e : sl.b;
            if (*p != '\n' && *p != '#') sl.lines.push_back(Line{p, eol, mergeKeyOf(p, eol)});   // empty and header lines do not take part
            p = e ? e + 1 : sl.b;
        }
    };
    {
        const int nT = (int)std::min<size_t>({(size_t)8, slices.size(), std::max<size_t>(1, std::thread::hardware_concurrency())});
        std::atomic<size_t> next(0);
        auto work = [&] { for (size_t k; (k = next.fetch_add(1)) < slices.size();) scan(slices[k]); };
        std::vector<std::thread> th;
        for (int t = 1; t < nT; ++t) th.emplace_back(work);
        work();
        for (std::thread& x : th) x.join();
    }
    std::vector<std::vector<Line>> lines((size_t)std::max(n, 0));
    for (Slice& sl : slices) {
        std::vector<Line>& L = lines[(size_t)sl.text];
        if (L.empty()) L.swap(sl.lines); else L.insert(L.end(), sl.lines.begin(), sl.lines.end());
    }
    // (2)
    std::vector<const char*> from;
    std::vector<size_t> len, at;
    size_t outAt = 0;
    std::vector<size_t> cur((size_t)std::max(n, 0), 0);
    auto live = [&](int i) { return cur[(size_t)i] < lines[(size_t)i].size(); };
    auto head = [&](int i) -> const MergeKey& { return lines[(size_t)i][cur[(size_t)i]].key; };
    for (;;) {
        int best = -1;
        for (int i = 0; i < n; ++i)                                        // (a handful of texts: a scan is as good as a heap)
            if (live(i) && (best < 0 || mergeLess(head(i), head(best)))) best = i;
        if (best < 0) break;
        // the other texts' smallest head bounds the run
        int other = -1;
        for (int i = 0; i < n; ++i) if (i != best && live(i) && (other < 0 || mergeLess(head(i), head(other)))) other = i;
        const std::vector<Line>& L = lines[(size_t)best];
        size_t k = cur[(size_t)best] + 1;
        if (other < 0) k = L.size();
        else
            while (k < L.size() && !(mergeLess(head(other), L[k].key) || (other < best && !mergeLess(L[k].key, head(other))))) ++k;
        // lines [cur, k) of this text: contiguous in it unless empty / header lines lie between them -- those are cut out
        size_t a = cur[(size_t)best];
        while (a < k) {
            size_t b = a + 1;
            while (b < k && L[b].p == L[b - 1].eol + 1) ++b;
            const size_t l = (size_t)(L[b - 1].eol - L[a].p) + 1;             // + the newline (written below when the text ends without one)
            from.push_back(L[a].p); len.push_back(l - 1); at.push_back(outAt);
            outAt += l;
            a = b;
        }
        cur[(size_t)best] = k;
    }
    char* out = (char*)malloc(total + 1);
    if (!out) return PLAT_ERR_NOMEM;
    // (3)
    copyPieces(out, from, len, at);
    for (size_t i = 0; i < at.size(); ++i) out[at[i] + len[i]] = '\n';
    out[outAt] = 0;
    *out_text = out; *out_len = outAt;
    return PLAT_OK;
}

// ---- probes of the Python-2 restatements (records.hpp), for tests/test_py2_semantics_cpu.py; not part of the caller's interface
CALLER_EXPORT double plat_caller_debug_round2(double x) { return py2_round2(x); }
CALLER_EXPORT double plat_caller_debug_round0(double x) { return py2_round0(x); }
CALLER_EXPORT unsigned long long plat_caller_debug_string_hash(const char* s) { return (unsigned long long)py2_string_hash(s ? s : ""); }
CALLER_EXPORT void plat_caller_debug_str(double x, char* out, size_t cap) {
    const std::string t = py2_str(x);
    snprintf(out, cap, "%s", t.c_str());
}
CALLER_EXPORT unsigned long long plat_caller_debug_tuple_hash(const unsigned long long* item_hashes, int n) {
    std::vector<uint64_t> h(item_hashes, item_hashes + n);
    return (unsigned long long)py2_tuple_hash(h.data(), n);
}
CALLER_EXPORT unsigned long long plat_caller_debug_variant_hash(const char* ref_name, long long ref_pos, const char* removed, const char* added) {
    return (unsigned long long)py2_variant_hash(py2_string_hash(ref_name ? ref_name : ""), ref_pos, removed ? removed : "", removed ? strlen(removed) : 0,
                                                added ? added : "", added ? strlen(added) : 0);
}
CALLER_EXPORT double plat_caller_debug_prior(const char* ref, long long ref_len, long long pos, const char* removed, const char* added) {
    Fasta fa;
    fa.seq = (const uint8_t*)ref; fa.len = ref_len;
    Variant v((int)pos, removed ? removed : "", added ? added : "", 1, PLATYPUS_VAR);
    return calculatePrior(v, fa);
}
CALLER_EXPORT void plat_caller_debug_fixed(double x, int decimals, char* out, size_t cap) {
    std::string t;
    append_fixed(t, x, decimals);
    snprintf(out, cap, "%s", t.c_str());
}
CALLER_EXPORT void plat_caller_debug_dict_slot_order(const unsigned long long* hashes, int n, int* out) {
    std::vector<uint64_t> h(hashes, hashes + n);
    const std::vector<int> order = py2_dict_slot_order(h);
    for (int i = 0; i < n; ++i) out[i] = order[(size_t)i];
}
// names: '\n'-separated, in insertion order; out: the iteration order of the set, '\n'-separated
CALLER_EXPORT void plat_caller_debug_set_order(const char* names, char* out, size_t cap) {
    std::vector<std::string> in;
    std::string cur;
    for (const char* p = names; p && *p; ++p) { if (*p == '\n') { in.push_back(cur); cur.clear(); } else cur += *p; }
    if (!cur.empty()) in.push_back(cur);
    std::string t;
    for (const std::string& k : py2_set_order(in)) { if (!t.empty()) t += '\n'; t += k; }
    snprintf(out, cap, "%s", t.c_str());
}
