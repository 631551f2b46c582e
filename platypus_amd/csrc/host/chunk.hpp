// chunk.hpp -- one chunk of regions going through the stages on one worker thread
// (native region loop, libplat_caller.so: see region_caller.cpp for the stage map and the reference citations)
#pragma once
#include "caller_common.hpp"

namespace plathost {

// ---- the chunk pipeline ----------------------------------------------------------------------------------------------------------------
// One chunk of regions on one worker: members and the stages' entry points; the stages themselves are stage_a.hpp ... stage_f.hpp.
struct Chunk {
    Slot& s;
    const Options& o;
    int nInd;
    const char* const* names;
    std::vector<RegionWork*> regions;
    plat_caller_stats& st;
    std::mutex& stMutex;

    void uploadReads();
    size_t nGood = 0, nBad = 0, nBroken = 0;
    int nScan = 0, maxReadLen = 0;
    int maxPerRead = 8;

    void scanCandidates();
    bool hostTally = false;
    int mergeCap = 2048;
    std::string refBlob;                                                   // the chunk's reference windows on the host (hostRefBlob())
    const uint8_t* refDev = nullptr;                                        // ... and on the device
    const std::string& hostRefBlob() {
        if (refBlob.empty())
            for (RegionWork* r : regions)
                for (size_t i = 0; i < r->samples.size(); ++i) {
                    const int64_t a = std::max<int64_t>(0, (int64_t)r->in->start - 2000), e = std::min<int64_t>((int64_t)r->in->end + 2000, r->fa.len - 1);
                    refBlob.append((const char*)r->fa.seq + a, (size_t)(e - a));
                }
        return refBlob;
    }
    Layout lmLayout;

    // -- B on the device (plat_stage_b_batch): regions with one sample, candidates from the reads alone, no reference-call blocks
    bool deviceB = false;
    DeviceBatch devBatch;
    int capV = 0, capW = 0, capA = 0;                                       // (of this chunk: the worker's current ones, Slot::sbCap*)
    Layout sbOut;
    bool eligibleDeviceB() const;
    void launchStageB();

    void stageBFromDevice();
    // D's inputs (distinct variants, haplotype masks, priors: Population.computeVariantPosteriors, cpopulation.pyx:596-621) of a list of windows, in list order
    struct PosteriorInputs {
        std::vector<int32_t> pwin;
        std::vector<int64_t> poff{0};
        std::vector<uint8_t> pmask;
        std::vector<double> pprior;
        size_t windows = 0;
        void clear() { pwin.clear(); poff.assign(1, 0); pmask.clear(); pprior.clear(); windows = 0; }
        void add(RegionWork& r, WindowWork& w, bool refCalls) {
            ++windows;
            w.distinct.clear();
            for (const Hap& h : w.haps)
                for (Variant* v : h.variants) if (!contains(w.distinct, v)) w.distinct.push_back(v);
            for (Variant* v : w.distinct) {
                pwin.push_back(w.bw);
                for (const Hap& h : w.haps) pmask.push_back(contains(h.variants, v) ? 1 : 0);
                poff.push_back((int64_t)pmask.size());
                { PROF("s5.prior"); pprior.push_back(calculatePrior(*v, r.fa)); }
            }
            if (refCalls)                                               // pop.calculatePosterior(v, 1) of outputRefCall: the window's candidates under a flat prior
                for (Variant* v : w.vars) {
                    pwin.push_back(w.bw);
                    for (const Hap& h : w.haps) pmask.push_back(contains(h.variants, v) ? 1 : 0);
                    poff.push_back((int64_t)pmask.size());
                    pprior.push_back(0.5);
                }
        }
    };
    PosteriorInputs prePosterior;                                           // of the windows the device prepared, in the order callWindows(devWins) lists them

    // -- A3: the assembler part of generateVariantsInRegion (variantcaller.pyx:496-519): tiles of assemblyRegionSize every
    // max(100, min(1000, size / 2)) bases, doWeNeedToAssembleThisRegion (:276-321) per tile, the reads loadBAMDataIntoGraph would load
    // (assembler.pyx:1391-1425: good reads between the window pointers, badReads / brokenMates if the options say so, QCFail reads never)
    // gathered from the chunk's device table; ALL tiles of the chunk in one plat_assemble_batch
    struct Tile { int region, assemStart, assemEnd, refStart; };
    void assembleLaunch();
    void assembleEnqueue();
    void assembleCollect();
    std::vector<Tile> asmTiles;
    plat_assembly_batch asmBatch;
    plat_assembly_hints asmHints;
    int asmN = 0;
    int asmMaxVars = 64, asmBlob = 4096;

    // -- B1: candidates of one region -> merged, per-sample support filter, left-normalised, filtered (variantcaller.pyx:439-531)
    // one sample's variantHeap: its distinct records in first-occurrence order with the number of reads showing each (addVariantToList),
    // from the scan's records on the host
    struct CandKey { int pos, nrem, nadd, count; const char* rem; const char* add; };
    void tallySample(const RegionWork& r, size_t i, std::vector<CandKey>& keys, std::deque<std::string>& addedStore, int64_t* nRecords);
    bool passesSupport(const RegionWork& r, size_t i, const CandKey& k) const;

    void regionVariants(RegionWork& r, int scan0);
    bool recordsOnHost = false;
    bool readCodes = false;                                                 // the chunk's read blob has its 2-bit codes in s.t_codes (every table packed, exceptions A/C/G/T/N only)
    int64_t tabPackedBytes = 0, tabBlobBytes = 0;                          // this chunk's table: packed bytes expanded on the device, bytes of bases in all
    size_t recArenaBytes = 0;

    Hap makeHap(const RegionWork& r, const WindowWork& w, const VarList& vs) const;

    void regionWindows(RegionWork& r);
    static std::vector<int> snapshotNR(const PtrList& ptrs);

    bool refCallLine(const RegionWork& r, std::string& out, int windowStart, int windowEnd, const std::vector<int>& nReads, bool hasVariants, double maxPost) const;

    void prepareWindow(RegionWork& r, WindowWork& w);

    void finishHaplotypes(RegionWork& r, WindowWork& w, std::vector<Hap>& haps);

    void greedyRounds();
    static void pushScored(WindowWork& w, const ScoredHap& item, int originalMax);
    int regionSlot(int regionIndex) const { return regionIndex - regions[0]->index; }

    void callWindows(std::vector<WindowWork*>& wins, bool fromDevice = false);
    void refCallBlocksBetween(RegionWork& r, WindowWork& w);
    void countCalled(size_t n) { std::lock_guard<std::mutex> g(stMutex); st.n_windows_called += (int64_t)n; }

    void writeWindow(RegionWork& r, WindowWork& w, const std::vector<int64_t>& klo);

    double stage[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stageWait[8] = {0, 0, 0, 0, 0, 0, 0, 0}, waitMark = 0;
    Clock::time_point mark;
    void lap(int k) { const auto now = Clock::now(); stage[k] += secs(mark, now); mark = now; stageWait[k] += s.t_wait - waitMark; waitMark = s.t_wait; }

    // window storage of the regions this worker has finished, for the regions of its next chunks (WindowList) -- in the worker's Slot; the worker
    // frees it when the call's chunks run out (region_caller.cpp: keeping it for the next call measured slower)
    std::vector<std::vector<WindowWork>>& spareWindows() { if (!s.spare) s.spare = new SparePools(); return s.spare->windows; }

    std::vector<std::unique_ptr<Variant[]>>& spareVariants() { if (!s.spare) s.spare = new SparePools(); return s.spare->variants; }

    void run() {
        // plat_caller_count_cells (the untimed counting pass of a measurement): one chunk at a time, so that the live kernel timers of its
        // likelihood batch (HIP events on this worker's stream) time its kernels and not the other workers'
        static std::mutex countMutex;
        std::unique_lock<std::mutex> oneAtATime(countMutex, std::defer_lock);
        if (s.countCells) oneAtATime.lock();
        const auto t0 = Clock::now();
        const double cpu0 = threadCpuSeconds();
        double wait0 = s.t_wait;
        mark = t0; waitMark = wait0;
        if (s.countCells) ck(plat_profile_enable(s.ctx, 1), "plat_profile_enable");     // (the counting pass: live timers of the table kernels too)
        {   // every region starts with the window storage of a region this worker finished earlier, if there is one
            std::vector<std::vector<WindowWork>>& spare = spareWindows();
            for (RegionWork* r : regions)
                if (!spare.empty() && r->windows.store.empty()) { r->windows.store.swap(spare.back()); spare.pop_back(); r->windows.n = 0; }
            std::vector<std::unique_ptr<Variant[]>>& sv = spareVariants();
            for (RegionWork* r : regions)                               // ... and with two blocks of Variant objects (128: a region of the WGS job holds ~90)
                for (int k = 0; k < 2 && !sv.empty() && r->pool.blocks.size() < 2 && r->pool.n == 0; ++k) { r->pool.blocks.push_back(std::move(sv.back())); sv.pop_back(); }
        }
        { PROF("s0.uploadReads"); uploadReads(); }
        lap(0);
        deviceB = eligibleDeviceB();
        assembleLaunch();
        if (o.getVariantsFromBAMs) { PROF("s1.scanCandidates"); scanCandidates(); }
        if (s.countCells) {
            plat_profile pf;
            memset(&pf, 0, sizeof pf);
            ck(plat_profile_last(s.ctx, &pf), "plat_profile_last");
            if (getenv("PLAT_CALLER_TRACE")) fprintf(stderr, "[plat_caller] table kernels: unpack %.3f ms (%lld packed bytes), candidates %.3f ms (%lld bytes)\n", pf.ms_unpack, (long long)tabPackedBytes, pf.ms_candidates, (long long)tabBlobBytes);
            // one byte in, two out per base (+ a quarter: the 2-bit codes, when the chunk has them); the scan has to read the bases once: as 2-bit codes when it
            // runs on them (qualities and bytes only where codes differ), as bytes otherwise
            if (pf.ms_unpack > 0) { s.secUnpack += 1e-3 * pf.ms_unpack; s.unpackBytes += 3 * tabPackedBytes + (readCodes ? tabPackedBytes / 4 : 0); s.nUnpack += 1; }
            if (pf.ms_candidates > 0) { s.secCand += 1e-3 * pf.ms_candidates; s.candBytes += readCodes ? tabBlobBytes / 4 : tabBlobBytes; s.nCand += 1; }
        }
        assembleCollect();
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
        PROF("s7.assemble_text_release");
        int64_t nRec = 0, nRef = 0;
        for (RegionWork* r : regions) {
            int nHapLast = 0;                                               // haplotypes of the last window set up in this region (Population.nHaplotypes)
            {   // (the region's text in one allocation: it is the sum of its items' texts but for the rare lines written here)
                size_t need = r->text.size() + 256;
                for (const Item& it : r->items) need += it.kind == 1 ? it.text.size() : r->windows[(size_t)it.window].text.size();
                r->text.reserve(need);
            }
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
            { PROF("s7.release"); r->release(&spareWindows(), &spareVariants()); }
        }
        if (s.countCells) {                                                 // every kernel of this chunk, live (HIP events around each launch)
            ck(plat_kernel_times(s.ctx, s.ktMs, s.ktLaunches), "plat_kernel_times");
            ck(plat_profile_enable(s.ctx, 0), "plat_profile_enable");
        } else if (s.timeKernel >= 0) ck(plat_kernel_times(s.ctx, s.ktMs, s.ktLaunches), "plat_kernel_times");   // (one kernel, timed inside the ordinary run: its launches are over)
        const double total = secs(t0, Clock::now()), waited = s.t_wait - wait0;
        std::lock_guard<std::mutex> g(stMutex);
        st.n_windows += nWin; st.n_variants += nVar; st.n_candidate_records += nCand; st.n_records += nRec; st.n_refcall_records += nRef;
        st.seconds_host += total - waited; st.seconds_device_wait += waited; st.seconds_worker_cpu += threadCpuSeconds() - cpu0;
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
