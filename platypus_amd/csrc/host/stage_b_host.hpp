// stage_b_host.hpp -- B on the host: candidates -> variants -> windows -> haplotypes, the greedy haplotype filter (cohorts, assembly / reference-call runs, and whatever the device flags)
// (native region loop, libplat_caller.so: see region_caller.cpp for the stage map and the reference citations)
#pragma once
#include "chunk.hpp"

namespace plathost {

inline void Chunk::tallySample(const RegionWork& r, size_t i, std::vector<CandKey>& keys, std::deque<std::string>& addedStore, int64_t* nRecords) {
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
            CandKey key{std::max(0, rec[0]), rec[1], rec[2], 1, rec[1] ? hostRefBlob().data() + rec[3] : "", addp};
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
inline bool Chunk::passesSupport(const RegionWork& r, size_t i, const CandKey& k) const {
    int s0, e0;
    r.samples[i].reads.overlapRange(k.pos, k.pos + 1, s0, e0);
    const int total = e0 - s0;
    const double frac = total == 0 ? 0.0 : (double)k.count / total;
    return frac >= o.minVarFreq || k.nadd != k.nrem;
}

// -- B1: candidates of one region -> merged, per-sample support filter, left-normalised, filtered (variantcaller.pyx:439-531)
inline void Chunk::regionVariants(RegionWork& r, int scan0) {
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
                pass(std::max(0, c[3]), c[4] ? hostRefBlob().data() + c[6] : "", c[4], added.data(), c[5], c[1]);
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

// -- B2/B3: windows, window pointers, haplotype enumeration (callVariantsInWindow up to Population.setup)
inline Hap Chunk::makeHap(const RegionWork& r, const WindowWork& w, const VarList& vs) const {
    Hap h;
    h.variants = vs;
    h.seq = haplotypeSequence(r.fa, w.hapStart, w.hapEnd, w.endBuf, vs);
    if (h.seq.size() > 16384) throw WindowError("Haplotype is too long. Max allowed length is 16384");   // chaplotype.pyx:180-183
    return h;
}

inline void Chunk::regionWindows(RegionWork& r) {
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

inline std::vector<int> Chunk::snapshotNR(const PtrList& ptrs) {
    std::vector<int> nr;
    for (const Ptrs& p : ptrs) nr.push_back(p.ge - p.gs);
    return nr;
}

inline void Chunk::prepareWindow(RegionWork& r, WindowWork& w) {
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
inline void Chunk::finishHaplotypes(RegionWork& r, WindowWork& w, std::vector<Hap>& haps) {
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
inline void Chunk::greedyRounds() {
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

inline void Chunk::pushScored(WindowWork& w, const ScoredHap& item, int originalMax) {
    if ((int)w.heap.size() < originalMax) heapPush(w.heap, item); else heapPushPop(w.heap, item);
}

}  // namespace plathost
