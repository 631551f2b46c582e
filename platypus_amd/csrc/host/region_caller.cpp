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
// Regions are processed in chunks; a chunk goes through the stages, one file each:
//   A  stage_a.hpp         the chunk's read table (uploaded, or put together on the device from resident tables: plat_concat_read_tables,
//                          plat_unpack_reads_pieces), the candidate scan + merge (plat_candidates_batch, plat_candidates_merge_batch),
//                          assembler tiles (plat_assemble_batch)
//   B  stage_b_device.hpp  regions with one sample: candidates -> variants -> windows -> window pointers -> haplotypes -> the window
//                          batch, on the device (plat_stage_b_batch), launched behind the merge with no round trip in between
//      stage_b_host.hpp    the same on the host: cohorts, assembly / reference-call runs, and every region or window the device flags
//                          (greedy haplotype filter: plat_align_window_batch per round)
//   C-E stage_cde.hpp      window read slices gathered on the device, likelihoods, genotype likelihoods, HapScore, EM (plat_gather_reads,
//                          plat_align_window_batch_async, plat_genotype_window_batch, plat_haplotype_score_batch, plat_em_window_batch);
//                          priors, variant masks -> posteriors (plat_variant_posterior_batch); read statistics + per-site genotype calls
//                          (plat_variant_read_stats_batch, plat_genotype_call_batch)
//   F  stage_f.hpp         INFO / FILTER arithmetic, record text (records.hpp)
// on one worker thread with its own plat_ctx and stream (chunk.hpp: Chunk::run); several workers run side by side, so the uploads,
// kernels and host stages of different chunks overlap.  This file: the feeds, the worker pool, the C entry points, the merge of
// record texts.  Same text as platypus_amd/caller.py::callVariantsInRegions (tests/test_native_caller_*.py).
#include <sys/mman.h>

#include "caller_common.hpp"
#include "chunk.hpp"
#include "stage_a.hpp"
#include "stage_b_device.hpp"
#include "stage_b_host.hpp"
#include "stage_cde.hpp"
#include "stage_f.hpp"

using namespace plathost;

struct plat_caller {
    int device = 0, nWorkers = 1, regionsPerChunk = 4;
    bool countCells = false;
    int timeKernel = -1;                                                    // plat_caller_time_kernel
    std::vector<std::unique_ptr<Slot>> slots;
    std::string lastError;
    std::vector<int64_t> lastLengths;                                       // bytes of record text of every region of the last call, in list order
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
        // a chunk's waits last milliseconds: the workers look at their events every 500 us instead of every 40 (measured on the whole-genome job,
        // 24 workers on 16 CPUs: 5.2 M windows/s at 40 us, 5.8 M at 250, 6.2 M at 500, 6.0 M at 1000, 5.5 M at 2000; PLAT_CALLER_POLL_US / PLAT_SYNC_POLL_US override)
        if (rc == PLAT_OK) { const char* e = getenv("PLAT_CALLER_POLL_US"); rc = plat_sync_poll_us(s->ctx, e && atoi(e) >= 0 ? atoi(e) : (c->regionsPerChunk >= 32 ? 500 : 100)); }
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

CALLER_EXPORT int plat_caller_time_kernel(plat_caller* c, int id) {
    if (!c || id >= PLAT_KT_COUNT) return PLAT_ERR_INVALID;
    c->timeKernel = id < 0 ? -1 : id;
    for (auto& q : c->slots) { q->timeKernel = c->timeKernel; plat_kernel_timer_only(q->ctx, c->timeKernel); }
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
                   z.d_scratch, z.d_pairoff, z.d_gloff, z.d_hapoff, z.d_readoff, z.d_hapseq, z.d_kind, z.c_refdev, z.t_codes, z.c_refcodes, z.c_refirr);
        plat_stream_destroy(z.ctx, z.stream);
        plat_ctx_destroy(z.ctx);
        delete z.spare; z.spare = nullptr;
    }
    delete c;
    return PLAT_OK;
}

CALLER_EXPORT const char* plat_caller_last_error(const plat_caller* c) { return c ? c->lastError.c_str() : ""; }
// Large text blocks are kept for the next call instead of going back to the system: a whole genome's record text is ~0.8 GB, and a fresh
// block of that size is 0.2 M first-touch page faults (or 400 huge ones) plus their release per call -- 30-60 ms of a 0.5 s pass.  At most
// two blocks are kept (the caller typically still holds the previous call's text while the next is written).
static std::mutex g_textMutex;
static std::vector<std::pair<void*, size_t>> g_textLive, g_textSpare;     // (big blocks handed out; blocks given back and kept), with capacities
static void* takeSpareText(size_t bytes) {
    std::lock_guard<std::mutex> g(g_textMutex);
    for (size_t i = 0; i < g_textSpare.size(); ++i)
        if (g_textSpare[i].second >= bytes && g_textSpare[i].second <= 2 * bytes + ((size_t)64 << 20)) {
            const std::pair<void*, size_t> b = g_textSpare[i];
            g_textSpare.erase(g_textSpare.begin() + (long)i);
            g_textLive.push_back(b);
            return b.first;
        }
    return nullptr;
}
CALLER_EXPORT void plat_caller_free(void* p) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> g(g_textMutex);
        for (size_t i = 0; i < g_textLive.size(); ++i)
            if (g_textLive[i].first == p) {
                const std::pair<void*, size_t> b = g_textLive[i];
                g_textLive.erase(g_textLive.begin() + (long)i);
                if (g_textSpare.size() < 2) { g_textSpare.push_back(b); return; }
                // (two spares already: the smallest of the three goes back to the system)
                size_t k = 0;
                for (size_t j = 1; j < g_textSpare.size(); ++j) if (g_textSpare[j].second < g_textSpare[k].second) k = j;
                if (g_textSpare[k].second < b.second) { p = g_textSpare[k].first; g_textSpare[k] = b; }
                break;
            }
    }
    free(p);
}

// ---- where the chunks of a call come from -----------------------------------------------------------------------------------------------
// (a worker asks for its next chunk of regions, calls it, and hands it back)
struct Feed {
    virtual bool next(std::vector<RegionWork*>& out) = 0;                  // false: no more chunks (or the call has failed)
    virtual void done(const std::vector<RegionWork*>& chunk) = 0;
    virtual ~Feed() {}
};

static bool checkHints() { const char* e = getenv("PLAT_CALLER_CHECK_HINTS"); return e && e[0] == '1'; }     // (read once per call of the library, not once per process)
static std::unique_ptr<RegionWork> makeRegionWork(const plat_region* in, int index, int n_samples, int& longestOut, bool check) {
    std::unique_ptr<RegionWork> r(new RegionWork());
    r->in = in; r->index = index;
    r->fa.seq = in->contig_seq; r->fa.len = in->contig_len;
    r->samples.resize((size_t)n_samples);
    int longest = 0;
    for (int i = 0; i < n_samples; ++i) {
        const plat_sample_reads& sr = in->samples[i];
        SampleView& sv = r->samples[(size_t)i];
        sv.reads.t = &sr.reads; sv.bad.t = &sr.bad_reads; sv.broken.t = &sr.broken_mates;
        // (the loader's own figures when it gives them: plat_read_table.longest_read / most_bases; walked here otherwise)
        const plat_read_table* tabs[3] = {&sr.reads, &sr.bad_reads, &sr.broken_mates};
        TableView* views[3] = {&sv.reads, &sv.bad, &sv.broken};
        for (int k = 0; k < 3; ++k) {
            const plat_read_table& t = *tabs[k];
            if (check && ((t.longest_read > 0 && t.longest_read != longestRead(t)) || (t.most_bases > 0 && t.most_bases != mostBases(t)) || t.longest_read < 0 || t.most_bases < 0))
                throw std::runtime_error("plat_read_table.longest_read / most_bases do not describe the table's reads");
            views[k]->longest = t.longest_read > 0 ? t.longest_read : longestRead(t);
            views[k]->maxLen = t.most_bases > 0 ? t.most_bases : mostBases(t);
        }
        longest = std::max(longest, sv.reads.longest);
    }
    longestOut = longest;
    return r;
}
// options.rlen follows the longest read of each region and is kept from the region before when a region has no reads (variantcaller.pyx:476-488)
static inline int nextRlen(int rlen, int longest, int maxSize, int fromBams = 1) { return (fromBams && longest > 0) ? (longest >= maxSize ? maxSize : longest) : rlen; }

// Where the chunks of a call begin: whole chunks of `per` regions while every worker gets the same number of them, and what is left of the
// list in one more round of EQUAL smaller chunks, one per worker -- with 61 chunks for 24 workers thirteen workers did three chunks while
// eleven did two and then watched; 48 whole chunks + 24 of half the size end together (PLAT_CALLER_EVEN_TAIL=0: chunks of `per` to the end).
static std::vector<int> chunkBounds(int n, int per, int workers) {
    std::vector<int> b{0};
    static const bool even = [] { const char* e = getenv("PLAT_CALLER_EVEN_TAIL"); return !(e && e[0] == '0'); }();
    int at = 0;
    if (even && workers > 1 && n > per * workers) {
        const int rounds = n / (per * workers);
        for (int k = 0; k < rounds * workers; ++k) { at += per; b.push_back(at); }
        const int rest = n - at;                                           // < per * workers: no chunk of the last round is larger than `per`
        if (rest >= 2 * workers)
            for (int w = 0; w < workers; ++w) { at += rest / workers + (w < rest % workers ? 1 : 0); b.push_back(at); }
    }
    while (at < n) { at = std::min(n, at + per); b.push_back(at); }
    return b;
}

// every region already in memory (plat_call_regions)
struct MemoryFeed : Feed {
    std::vector<std::unique_ptr<RegionWork>>& work;
    std::vector<int> bound;
    int nChunks;
    std::atomic<int> nextChunk{0};
    std::atomic<bool>& failed;
    MemoryFeed(std::vector<std::unique_ptr<RegionWork>>& w, int per_, int workers, std::atomic<bool>& f)
        : work(w), bound(chunkBounds((int)w.size(), per_, workers)), nChunks((int)bound.size() - 1), failed(f) {}
    bool next(std::vector<RegionWork*>& out) override {
        if (failed.load()) return false;
        const int ch = nextChunk.fetch_add(1);
        if (ch >= nChunks) return false;
        out.clear();
        for (int k = bound[(size_t)ch]; k < bound[(size_t)ch + 1]; ++k) out.push_back(work[(size_t)k].get());
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
    std::vector<int> chunkLoaded;                                          // regions of a chunk that are in: the workers are woken when a CHUNK is complete
    std::vector<int> bound, chunkOf;                                       // chunk c = regions [bound[c], bound[c + 1]) (chunkBounds)
    int nextToLoad = 0, nextChunk = 0, rlen, error = PLAT_OK;
    std::string errText;
    std::atomic<bool>& failed;
    double tLoad = 0, tWait = 0;
    const bool check = checkHints();
    StreamFeed(int n_, int nS, int per_, int workers, int maxSize_, int rlen0, plat_region_load_fn l, void* u, int nSlots, std::atomic<bool>& f)
        : n(n_), nSamples(nS), per(per_), nChunks(0), maxSize(maxSize_), load(l), user(u), work((size_t)n_), desc((size_t)n_),
          slotOf((size_t)n_, -1), longest((size_t)n_, 0), loaded((size_t)n_, 0), bound(chunkBounds(n_, per_, workers)), chunkOf((size_t)n_, 0), rlen(rlen0), failed(f) {
        nChunks = (int)bound.size() - 1;
        chunkLoaded.assign((size_t)nChunks, 0);
        for (int c = 0; c < nChunks; ++c) for (int k = bound[(size_t)c]; k < bound[(size_t)c + 1]; ++k) chunkOf[(size_t)k] = c;
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
            try { r = makeRegionWork(&desc[(size_t)idx], idx, nSamples, lg, check); }
            catch (const std::exception& e) { fail(PLAT_ERR_BAD_INPUT, e.what()); return; }
            const double dt = secs(t0, Clock::now());
            std::lock_guard<std::mutex> g(m);
            work[(size_t)idx] = std::move(r); slotOf[(size_t)idx] = slot; longest[(size_t)idx] = lg; loaded[(size_t)idx] = 1;
            tLoad += dt;
            // (one wake-up per chunk, not per region: a notify_all per region had every waiting worker re-check its chunk under this mutex
            //  thousands of times per call -- 5 ms of a 20 ms call with 20 workers)
            const int ch = chunkOf[(size_t)idx];
            if (++chunkLoaded[(size_t)ch] == bound[(size_t)ch + 1] - bound[(size_t)ch]) cvLoaded.notify_all();
        }
    }
    bool next(std::vector<RegionWork*>& out) override {
        const auto t0 = Clock::now();
        std::unique_lock<std::mutex> g(m);
        for (;;) {
            if (failed.load() || nextChunk >= nChunks) return false;
            const int ch = nextChunk, a = bound[(size_t)ch], b = bound[(size_t)ch + 1];
            if (chunkLoaded[(size_t)ch] == b - a) {
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
        // every worker frees its own spare storage when it runs out of chunks.  Keeping it for the next CALL (PLAT_CALLER_KEEP_SPARE=1) measured SLOWER on the
        // whole-genome job: 5.8-6.1 M windows/s against 6.5-6.6 M, 0.207 against 0.172 ms of worker CPU per region -- the next call's worker is a new thread,
        // often on the other NUMA node, and inherits ten thousand cold windows; freeing them all on one thread at the end of the call: 5.0 M
        static const bool keep = [] { const char* e = getenv("PLAT_CALLER_KEEP_SPARE"); return e && e[0] == '1'; }();
        if (!keep) { delete slot->spare; slot->spare = nullptr; }
    };
    nThreads = std::max(1, std::min<int>((int)c->slots.size(), nThreads));
    for (auto& q : c->slots) {
        q->countCells = c->countCells; q->nDpRef = q->cellsRef = q->nDpRun = q->cellsRun = 0;
        q->nAlign = q->alignHapBytes = q->alignReadBytes = q->alignReads = q->alignDpBytes = 0; q->secSeed = q->secDp = q->secSweep = q->secPairs = q->secUnpack = q->secCand = 0.0; q->unpackBytes = q->candBytes = q->nUnpack = q->nCand = 0;
        for (int k = 0; k < PLAT_KT_COUNT; ++k) { q->ktMs[k] = 0.0; q->ktLaunches[k] = 0; }
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
        st.seconds_kernel_unpack += q->secUnpack; st.seconds_kernel_candidates += q->secCand; st.unpack_bytes += q->unpackBytes; st.candidates_bytes += q->candBytes;
        st.n_unpack_launches += q->nUnpack; st.n_candidates_launches += q->nCand;
        for (int k = 0; k < PLAT_KT_COUNT && k < 32; ++k) { st.kernel_ms[k] += q->ktMs[k]; st.kernel_launches[k] += q->ktLaunches[k]; }
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
    const int nT = total < ((size_t)8 << 20) ? 1 : (int)std::min<size_t>(16, std::max<size_t>(1, std::thread::hardware_concurrency()));
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

// a block for ~100 MB of text: 2 MB aligned and advised for huge pages (a fresh block's first-touch faults are then hundreds, not tens of
// thousands); freed with free()
static char* allocText(size_t bytes) {
    const size_t big = (size_t)2 << 20;
    if (bytes < 4 * big) return (char*)malloc(bytes);
    if (void* spare = takeSpareText(bytes)) return (char*)spare;
    void* p = nullptr;
    const size_t cap = (bytes + bytes / 16 + big - 1) & ~(big - 1);       // (a little room: the next call's text is rarely the same size to the byte)
    if (posix_memalign(&p, big, cap) != 0) return nullptr;
#ifdef MADV_HUGEPAGE
    madvise(p, cap, MADV_HUGEPAGE);
#endif
    { std::lock_guard<std::mutex> g(g_textMutex); g_textLive.push_back({p, cap}); }
    return (char*)p;
}

static int finishText(plat_caller* c, std::vector<std::unique_ptr<RegionWork>>& work, char** out_text, size_t* out_len) {
    std::vector<const char*> from;
    std::vector<size_t> len, at;
    size_t total = 0;
    c->lastLengths.clear();
    for (auto& r : work) {
        c->lastLengths.push_back(r ? (int64_t)r->text.size() : 0);
        if (r) { from.push_back(r->text.data()); len.push_back(r->text.size()); at.push_back(total); total += r->text.size(); }
    }
    char* text = allocText(total + 1);
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
    const bool check = checkHints();
    for (int k = 0; k < n_regions; ++k) {
        int longest = 0;
        std::unique_ptr<RegionWork> r;
        try { r = makeRegionWork(&regions[k], k, n_samples, longest, check); }
        catch (const std::exception& e) { c->lastError = e.what(); return PLAT_ERR_BAD_INPUT; }          // (a hint that does not describe its table, PLAT_CALLER_CHECK_HINTS=1)
        rlen = nextRlen(rlen, longest, options->maxSize, options->getVariantsFromBAMs);
        r->rlen = rlen;
        work.push_back(std::move(r));
    }
    std::atomic<bool> failed(false);
    MemoryFeed feed(work, c->regionsPerChunk, (int)c->slots.size(), failed);
    rc = runWorkers(c, feed, failed, o, n_samples, sample_names, st, std::max(1, feed.nChunks));
    if (rc != PLAT_OK) return rc;
    if ((rc = finishText(c, work, out_text, out_len)) != PLAT_OK) return rc;
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
    StreamFeed feed(n_regions, n_samples, per, nWorkers, options->maxSize, options->rlen, load, user, n_slots, failed);
    feed.fromBams = options->getVariantsFromBAMs;
    std::vector<std::thread> loaders;
    for (int i = 0; i < std::min(n_loader_threads, std::max(1, n_regions)); ++i) loaders.emplace_back([&feed] { feed.loader(); });
    rc = runWorkers(c, feed, failed, o, n_samples, sample_names, st, std::max(1, feed.nChunks));
    const double tWorkers = secs(t0, Clock::now());
    { std::lock_guard<std::mutex> g(feed.m); feed.cvSlot.notify_all(); }
    if (rc != PLAT_OK) feed.fail(rc, c->lastError);                       // (wakes loaders that wait for a slot)
    for (std::thread& t : loaders) t.join();
    if (rc == PLAT_OK && feed.error != PLAT_OK) { rc = feed.error; c->lastError = feed.errText; }
    if (rc != PLAT_OK) return rc;
    if ((rc = finishText(c, feed.work, out_text, out_len)) != PLAT_OK) return rc;
    options->rlen = feed.rlen;
    st.seconds_total = secs(t0, Clock::now());
    st.seconds_load = feed.tLoad; st.seconds_source_wait = feed.tWait;
    if (getenv("PLAT_CALLER_TRACE")) fprintf(stderr, "[plat_caller] call: workers done after %.2f ms, text put together after %.2f ms\n", 1e3 * tWorkers, 1e3 * st.seconds_total);
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
CALLER_EXPORT int plat_caller_region_text_lengths(const plat_caller* c, int64_t* out, int n) {
    if (!c || !out || n < 0 || (size_t)n != c->lastLengths.size()) return PLAT_ERR_INVALID;
    for (int i = 0; i < n; ++i) out[i] = c->lastLengths[(size_t)i];
    return PLAT_OK;
}

// n blocks of text -> one block: src[i][0 .. len[i]) to out + at[i] (at ascending), on up to 16 threads
CALLER_EXPORT int plat_merge_region_blocks(int n, const char* const* src, const size_t* len, const size_t* at, size_t total, char** out_text) {
    if (n < 0 || !out_text || (n > 0 && (!src || !len || !at))) return PLAT_ERR_INVALID;
    for (int i = 0; i < n; ++i) if (at[i] + len[i] > total || (i > 0 && at[i] < at[i - 1] + len[i - 1])) return PLAT_ERR_INVALID;
    char* out = allocText(total + 1);
    if (!out) return PLAT_ERR_NOMEM;
    std::vector<const char*> from(src, src + n);
    std::vector<size_t> l(len, len + n), a(at, at + n);
    copyPieces(out, from, l, a);
    out[total] = 0;
    *out_text = out;
    return PLAT_OK;
}

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
