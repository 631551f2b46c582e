// caller_common.hpp -- what every stage of the native region loop shares: grow-only pinned + device buffers (Slot), one-copy layouts,
// read tables as the loop sees them, windows / regions in flight, the window batch and the call that runs it on the device.
#pragma once
#include <atomic>
#ifdef PLAT_HOSTPROF
#include <x86intrin.h>
#include <map>
#endif
#include <chrono>
#include <time.h>
#include <condition_variable>
#include <cstdarg>
#include <deque>
#include <functional>
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
static inline double threadCpuSeconds() { timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
#ifdef PLAT_HOSTPROF                                                  // (local measurement builds only: cycle counts of named scopes)
// per-thread counters (a scope costs two rdtsc and two plain adds: shared atomics cost more than most of the scopes they measured); the
// threads' blocks are linked into a list and summed by profDump
static const char* g_profName[256];
static std::atomic<int> g_profN{0};
struct ProfBlock { unsigned long long cyc[256] = {}, calls[256] = {}; ProfBlock* next = nullptr; };
static std::atomic<ProfBlock*> g_profBlocks{nullptr};
static ProfBlock* profMine() {
    static thread_local ProfBlock* mine = nullptr;
    if (!mine) { mine = new ProfBlock(); ProfBlock* h = g_profBlocks.load(); do { mine->next = h; } while (!g_profBlocks.compare_exchange_weak(h, mine)); }
    return mine;
}
static int profId(const char* n) { const int k = g_profN.fetch_add(1); g_profName[k] = n; return k; }
struct ProfScope { int k; ProfBlock* b; unsigned long long t0; ProfScope(int k_) : k(k_), b(profMine()), t0(__rdtsc()) {}
                   ~ProfScope() { b->cyc[k] += __rdtsc() - t0; b->calls[k] += 1; } };
#define PROF_CAT2(a, b) a##b
#define PROF_CAT(a, b) PROF_CAT2(a, b)
#define PROF(name) static const int PROF_CAT(profid_, __LINE__) = profId(name); ProfScope PROF_CAT(prof_, __LINE__)(PROF_CAT(profid_, __LINE__))
static void profDump(double n) {
    std::map<std::string, std::pair<unsigned long long, unsigned long long>> m;
    for (ProfBlock* b = g_profBlocks.load(); b; b = b->next)
        for (int k = 0; k < g_profN.load(); ++k) { m[g_profName[k]].first += b->cyc[k]; m[g_profName[k]].second += b->calls[k]; b->cyc[k] = 0; b->calls[k] = 0; }
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

struct SparePools;                                                         // window / Variant storage a worker keeps between the chunks of a call (defined behind WindowWork)
struct Slot {
    SparePools* spare = nullptr;                                           // (made by the worker's first chunk of a call, freed when its chunks run out)
    plat_ctx* ctx = nullptr;
    void* stream = nullptr;
    bool countCells = false;                                               // plat_caller_count_cells: likelihood batches through the synchronous entry point
    int timeKernel = -1;                                                   // plat_caller_time_kernel: this kernel's launches are timed in the ordinary calls
    int64_t nDpRef = 0, cellsRef = 0, nDpRun = 0, cellsRun = 0;            // ... and their plat_align_stats summed (this worker's share)
    int64_t nAlign = 0, alignHapBytes = 0, alignReadBytes = 0, alignReads = 0, alignDpBytes = 0;
    double secSeed = 0.0, secDp = 0.0, secSweep = 0.0, secPairs = 0.0, secUnpack = 0.0, secCand = 0.0;
    int64_t unpackBytes = 0, candBytes = 0, nUnpack = 0, nCand = 0;
    double ktMs[PLAT_KT_COUNT] = {};                                       // plat_kernel_times of this worker's chunks (counting pass)
    int64_t ktLaunches[PLAT_KT_COUNT] = {};
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
    Staged<uint8_t> c_ref, c_refdev;
    Staged<uint32_t> t_codes, c_refcodes;                                   // 2-bit base codes of the chunk's read blob / reference blob (the scan on codes, round 6)
    Staged<int32_t> c_refirr;                                               // per scan: the reference window holds a byte other than A, C, G, T, N
    Staged<plat_unpack_piece> c_pieces;
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
    Staged<double> p_prior, p_post, k_lik, k_out4, s_terms;
    Staged<int32_t> s_mmlq;
    bool infoOnDevice = false;                                            // this chunk's s_terms / s_mmlq are valid (plat_variant_info_batch)
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
    Staged<int64_t> sb_totals, sb_namehash;
    Staged<int32_t> d_hapbegin, d_readbegin, d_start, d_end, d_flank, d_segbegin, d_ngood, d_src, d_scratch;                 // device only
    Staged<int64_t> d_pairoff, d_gloff, d_hapoff, d_readoff;
    Staged<uint8_t> d_hapseq, d_kind;
    // many small arrays travel as ONE copy: they are views into these blocks (Layout)
    Arena a_tab, a_desc, a_cin, a_cout, a_mout, a_win, a_wout, a_pin, a_sin, a_sout, a_asin, a_asout, a_bin, a_bout;
    // per-region capacities of plat_stage_b_batch's outputs (they are downloaded whole): doubled when a region does not fit
    int sbCapV = 320, sbCapW = 192, sbCapA = 2048;
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
    // A fresh window in a recycled object (WindowList): every member back to its initial value, the containers EMPTY BUT WITH THEIR STORAGE --
    // a window is five heap blocks (haplotypes, text, positions, INFO, posteriors) that were allocated and freed once per window and pass.
    void reset() {
        region = 0; startPos = 0; endPos = 0;
        vars.clear(); allVars.clear(); ptrs.clear();
        nReads = 0; hapStart = 0; hapEnd = 0; endBuf = 0;
        refSeq.clear(); haps.clear(); live = false;
        greedy = false; byCoverage.clear(); step = 0; heap.clear(); cands.clear(); sampledSeg.clear(); sampled.clear();
        bw = -1; hapBegin = 0; onDevice = false;
        distinct.clear(); posterior.clear(); called.clear(); calledPost.clear(); byPos.clear(); info.clear();
        firstStatVar = 0; firstSite = 0; failed = false; text.clear(); nRecords = 0; nRefRecords = 0; firstFlat = -1;
    }
};
// The windows of a region: a vector whose elements outlive the region -- a worker keeps the storage of the regions it has finished and hands
// it to the regions of its next chunk (Chunk::run), so that in the steady state a window costs no allocation at all.  The part of
// std::vector's interface the loop uses; [0, n) are the region's windows, the objects behind them are spares.
struct WindowList {
    std::vector<WindowWork> store;
    size_t n = 0;
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    void reserve(size_t k) { if (k > store.capacity()) store.reserve(std::max(k, 2 * store.capacity())); }
    WindowWork& emplace_back() {
        if (n < store.size()) store[n].reset(); else store.emplace_back();
        return store[n++];
    }
    void push_back(WindowWork&& w) { if (n < store.size()) store[n] = std::move(w); else store.push_back(std::move(w)); ++n; }
    WindowWork& back() { return store[n - 1]; }
    WindowWork& operator[](size_t i) { return store[i]; }
    const WindowWork& operator[](size_t i) const { return store[i]; }
    WindowWork* begin() { return store.data(); }
    WindowWork* end() { return store.data() + n; }
};

// The Variants of a region: objects with stable addresses in blocks of 64 that OUTLIVE the region -- a worker hands the blocks of the regions it
// has finished to the regions of its next chunk (Chunk::run), a recycled object is assigned to (its strings keep their storage).
struct SparePools {
    std::vector<std::vector<WindowWork>> windows;
    std::vector<std::unique_ptr<Variant[]>> variants;
};
struct VariantPool {
    static constexpr size_t BLOCK = 64;
    std::vector<std::unique_ptr<Variant[]>> blocks;
    size_t n = 0;
    Variant* next() {
        if (n == blocks.size() * BLOCK) blocks.emplace_back(new Variant[BLOCK]);
        Variant* v = &blocks[n / BLOCK][n % BLOCK];
        ++n;
        return v;
    }
    Variant* make(int pos, const std::string& rem, const std::string& add, int nSupp, int source) {
        Variant* v = next();
        v->assign(pos, rem.data(), rem.size(), add.data(), add.size(), nSupp, source);
        return v;
    }
    Variant* make(int pos, const char* rem, size_t nrem, const char* add, size_t nadd, int nSupp, int source) {
        Variant* v = next();
        v->assign(pos, rem, nrem, add, nadd, nSupp, source);
        return v;
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
    WindowList windows;
    std::string text;
    int64_t nCandRecords = 0;
    // frees everything but the record text (called by the worker that finished the region, so that the cost of freeing thousands of
    // windows and haplotypes is spread over the workers instead of being paid serially at the end of plat_call_regions)
    // (spare: where the worker keeps window storage for its next chunk's regions; nullptr: freed)
    void release(std::vector<std::vector<WindowWork>>* spare = nullptr, std::vector<std::unique_ptr<Variant[]>>* spareVariants = nullptr) {
        if (spare && spare->size() < 512 && !windows.store.empty()) { spare->emplace_back(); spare->back().swap(windows.store); }
        else std::vector<WindowWork>().swap(windows.store);
        windows.n = 0;
        if (spareVariants && spareVariants->size() < 2048) for (auto& b : pool.blocks) spareVariants->push_back(std::move(b));
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
// `tail`: what the caller wants launched behind the batch on the same stream before the ONE wait (the posteriors: their inputs do not depend on
// the batch's results, only their kernel does)
static void runBatch(Slot& s, DeviceBatch& db, const Options& o, bool full, bool wantLoglik, const std::function<void(DeviceBatch&)>* tail = nullptr) {
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
        if (getenv("PLAT_CALLER_TRACE")) fprintf(stderr, "[plat_caller] likelihood batch: %d windows, %d haplotypes (longest %d), %lld pairs, %lld for the exact vote, %lld DPs launched / %lld reference\n",
                                                 db.nWindows, db.nHaps, db.maxHap, (long long)db.nPairs, (long long)as.n_seed_fallback, (long long)as.n_dp_launched, (long long)as.n_dp_reference);
        plat_profile pf;
        memset(&pf, 0, sizeof pf);
        ck(plat_profile_last(s.ctx, &pf), "plat_profile_last");     // (the profile stays on: Chunk::run collects every kernel's timers at its end)
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
    if (tail) (*tail)(db);
    s.sync("window batch");
}

// Upload a BatchBuilder and run it (runBatch)
static DeviceBatch runWindows(Slot& s, const BatchBuilder& b, const Options& o, bool full, bool wantLoglik, const std::function<void(DeviceBatch&)>* tail = nullptr) {
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
    runBatch(s, db, o, full, wantLoglik, tail);
    return db;
}

}  // namespace plathost
