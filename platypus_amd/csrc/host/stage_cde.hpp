// stage_cde.hpp -- C-E: the window batch (likelihoods, genotype likelihoods, HapScore, EM), posteriors, read statistics and genotype calls
// (native region loop, libplat_caller.so: see region_caller.cpp for the stage map and the reference citations)
#pragma once
#include "chunk.hpp"

namespace plathost {

// -- C..F for a list of windows
// fromDevice: the windows are those plat_stage_b_batch prepared -- their batch is on the device already (devBatch), w->bw / w->hapBegin are set
inline void Chunk::callWindows(std::vector<WindowWork*>& wins, bool fromDevice) {
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
    // D: distinct variants, masks, priors -> posteriors (Population.computeVariantPosteriors, cpopulation.pyx:596-621).  What the kernel is
    // given depends on the windows alone: it is put together and launched behind the batch (same stream), and the batch's one wait covers it
    size_t nV = 0;
    const std::function<void(DeviceBatch&)> posteriors = [&](DeviceBatch& dbb) {
        // (a device-built batch's inputs were put together window by window while stage B's output was turned into objects -- the windows were in
        //  the caches then; this pass over a chunk's ten thousand windows is one of four the host makes, and each one starts cold)
        PosteriorInputs local;
        const bool pre = fromDevice && prePosterior.windows == wins.size() && !wins.empty();
        PosteriorInputs& pi = pre ? prePosterior : local;
        if (!pre) for (WindowWork* w : wins) { PROF("s5.build"); pi.add(*regions[(size_t)regionSlot(w->region)], *w, o.outputRefCalls != 0); }
        std::vector<int32_t>& pwin = pi.pwin;
        std::vector<int64_t>& poff = pi.poff;
        std::vector<uint8_t>& pmask = pi.pmask;
        std::vector<double>& pprior = pi.pprior;
        nV = pwin.size();
        if (!nV) return;
        Layout L;
        L.add(z.p_win, nV); L.add(z.p_off, nV + 1); L.add(z.p_mask, pmask.size()); L.add(z.p_prior, nV);
        L.commit(z, z.a_pin);
        fill(z, z.p_win, pwin); fill(z, z.p_off, poff); fill(z, z.p_mask, pmask); fill(z, z.p_prior, pprior);
        z.p_post.reserve(z.ctx, nV + 1);
        L.upload(z, z.a_pin);
        ck(plat_variant_posterior_batch(z.ctx, (int)nV, nInd, dbb.maxH, dbb.hapbegin, dbb.gloff, dbb.ngood, z.o_gl.d, z.o_freq.d, z.p_win.d,
                                        z.p_off.d, z.p_mask.d, z.p_prior.d, z.p_post.d, z.stream), "plat_variant_posterior_batch");
        z.down(z.p_post, nV);
    };
    DeviceBatch db;
    if (fromDevice) { PROF("s4.runBatch"); db = devBatch; runBatch(z, db, o, true, false, &posteriors); }
    else {
        PROF("s4.runWindows");
        db = runWindows(z, b, o, true, false, &posteriors);
        for (WindowWork* w : wins) w->hapBegin = b.hapbegin[(size_t)w->bw];
    }
    {
        int64_t np = db.nPairs;
        if (fromDevice) { np = 0; for (WindowWork* w : wins) { int nr = 0; for (const Ptrs& p : w->ptrs) nr += (p.ge - p.gs) + (p.be - p.bs) + (p.ke - p.ks); np += (int64_t)w->haps.size() * nr; } }
        std::lock_guard<std::mutex> g(stMutex);
        st.n_pairs += np;                                               // (of the windows CALLED from this batch)
    }
    lap(4);
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
        w->info.reserve(w->called.size());
        const double* freq = z.o_freq.h + w->hapBegin;
        const int32_t* calls = z.o_calls.h + (size_t)w->bw * (size_t)nInd;
        for (size_t h = 0; h < w->haps.size(); ++h) {
            VarList seen;                                               // Haplotype.vcfINFO(): a dictionary over the haplotype's variants
            for (Variant* v : w->haps[h].variants) {
                if (contains(seen, v)) continue;
                seen.push_back(v);
                int ci = -1;
                for (size_t c = 0; c < w->called.size(); ++c) if (w->called[c] == v || w->called[c]->same(*v)) { ci = (int)c; break; }
                if (ci < 0) continue;
                VarInfo* d = nullptr;
                for (VarInfo& x : w->info) if (x.var == v || x.var->same(*v)) { d = &x; break; }
                if (!d) {
                    VarInfo n;
                    n.var = v;
                    PROF("s6.hp_sc");
                    n.HP = homopolymerLengthForOneVariant(*v, r.fa);
                    getSequenceContext(*v, r.fa, n.SC);
                    n.setPP(w->calledPost[(size_t)ci]);                                // "%.0f"

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
    PROF("s6.launch");
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
    LO.add(z.s_terms, nSV * 8); LO.add(z.s_mmlq, nSV);
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
    {   // the loops of ABPV / SbPval / MMLQ behind it, on the device (the host keeps the libm calls; a device library without it: the host's loops)
        static const bool hostInfo = getenv("PLAT_CALLER_HOST_INFO") != nullptr;
        const int rci = hostInfo ? PLAT_ERR_UNSUPPORTED
                                 : plat_variant_info_batch(z.ctx, (int)nSV, z.s_counts.d, z.s_moff.d, z.s_minq.d, z.s_nminq.d, z.s_terms.d, z.s_mmlq.d, z.stream);
        if (rci != PLAT_ERR_UNSUPPORTED) ck(rci, "plat_variant_info_batch");
        z.infoOnDevice = rci == PLAT_OK;
    }
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
inline void Chunk::refCallBlocksBetween(RegionWork& r, WindowWork& w) {
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

}  // namespace plathost
