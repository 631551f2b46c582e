// stage_b_device.hpp -- B on the device: plat_stage_b_batch and what the host makes of its output
// (native region loop, libplat_caller.so: see region_caller.cpp for the stage map and the reference citations)
#pragma once
#include "chunk.hpp"

namespace plathost {

inline bool Chunk::eligibleDeviceB() const {
    if (nInd != 1 || o.assemble || o.outputRefCalls || !o.getVariantsFromBAMs || o.maxHaplotypes < 3 || regions.empty()) return false;
    const char* e = getenv("PLAT_CALLER_HOST_B");                       // (measurements / tests: stage B on the host)
    return !(e && e[0] == '1');
}

inline void Chunk::launchStageB() {
    Slot& z = s;
    const size_t nR = regions.size();
    capV = z.sbCapV; capW = z.sbCapW; capA = z.sbCapA;
    for (RegionWork* r : regions) {                                     // (a long region gets room in proportion: ~1.5 variants / kb of the synthetic genome, x 2)
        const int len = std::max(0, r->in->end - r->in->start);
        capV = std::min(1024, std::max(capV, len / 300)); capW = std::min(1024, std::max(capW, len / 500));
    }
    size_t nBr = 0;
    for (RegionWork* r : regions) nBr += (size_t)r->samples[0].broken.n();
    Layout LI;
    LI.add(z.sb_rstart, nR); LI.add(z.sb_rend, nR); LI.add(z.sb_rlen, nR); LI.add(z.sb_tabbegin, 3 * nR); LI.add(z.sb_tabn, 3 * nR); LI.add(z.sb_tablongest, 3 * nR);
    LI.add(z.sb_matepos, nBr + 1); LI.add(z.sb_namehash, nR);
    LI.commit(z, z.a_bin);
    size_t mo = 0;
    for (size_t g = 0; g < nR; ++g) {
        RegionWork& r = *regions[g];
        SampleView& sv = r.samples[0];
        z.sb_rstart.h[g] = r.in->start; z.sb_rend.h[g] = r.in->end; z.sb_rlen.h[g] = r.rlen;
        z.sb_namehash.h[g] = (int64_t)py2_string_hash(r.in->chrom ? std::string(r.in->chrom) : std::string());      // (the dictionaries of Variants hash the contig's name)
        const TableView* tv[3] = {&sv.reads, &sv.bad, &sv.broken};
        for (int k = 0; k < 3; ++k) { z.sb_tabbegin.h[3 * g + k] = (int32_t)tv[k]->base; z.sb_tabn.h[3 * g + k] = tv[k]->n(); z.sb_tablongest.h[3 * g + k] = tv[k]->longest; }
        if (sv.broken.n()) memcpy(z.sb_matepos.h + mo, sv.broken.t->mate_pos, sizeof(int32_t) * (size_t)sv.broken.n());
        mo += (size_t)sv.broken.n();
    }
    z.sb_matepos.h[mo] = 0;
    LI.upload(z, z.a_bin);
    const size_t capBW = nR * (size_t)capW, capBH = nR * 1024, capBR = std::max<size_t>(4 * (nGood + nBad + nBroken), 65536), capHB = capBH * 1280;
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
    z.d_scratch.reserve(z.ctx, 56 * capBW + 48 * nR + 64, false);
    plat_stage_b_in in;
    memset(&in, 0, sizeof in);
    in.n_regions = (int32_t)nR; in.cap_per_scan = mergeCap; in.cand = z.m_cand.d; in.cand_n = z.m_n.d;
    in.cand_rec = getenv("PLAT_CALLER_NO_DEVICE_REPLAY") ? nullptr : z.c_rec.d; in.region_name_hash = z.sb_namehash.d;
    in.ref_seq = refDev; in.ref_off = z.c_refoff.d; in.ref_seq_start = z.c_rss.d; in.contig_len = z.c_clen.d;
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
inline void Chunk::stageBFromDevice() {
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
    prePosterior.clear();
    bool rows = false;
    for (size_t g = 0; g < nR; ++g)
        if (z.sb_hdr.h[8 * g] != 0) {                                   // this region's candidates for the host's code
            const size_t at = g * (size_t)mergeCap * 8, n = (size_t)std::max(0, z.m_n.h[2 * g]) * 8;
            if (n) ck(plat_memcpy_d2h(z.ctx, z.m_cand.h + at, z.m_cand.d + at, n * sizeof(int32_t), z.stream), "plat_memcpy_d2h");
            rows = true;
        }
    if (rows) z.sync("candidates");
    int hapRun = 0;
    int64_t nHostRegions = 0, nHostWindows = 0, nReplayed = 0;
    for (size_t g = 0; g < nR; ++g) {
        RegionWork& r = *regions[g];
        const int32_t* hdr = z.sb_hdr.h + 8 * g;
        if (hdr[0] != 0) {
            if (hdr[5] == 6) { z.sbCapV = std::min(2 * capV, 1024); z.sbCapW = std::min(2 * capW, 1024); z.sbCapA = std::min(2 * capA, 1 << 16); }   // more room from the next chunk on
            regionVariants(r, (int)g); regionWindows(r); ++nHostRegions; continue;
        }
        PROF("s2.fillRegion");
        const int nV = hdr[1], nW = hdr[2];
        nReplayed += hdr[6] != 0;
        r.nCandRecords += hdr[3];
        r.variants.clear();
        const uint8_t* blob = z.sb_added.h + g * (size_t)capA;
        for (int i = 0; i < nV; ++i) {
            PROF("s2.fill.variant");
            const size_t k = g * (size_t)capV + (size_t)i;
            const int nrem = z.sb_vnrem.h[k], nadd = z.sb_vnadd.h[k];
            Variant* v = r.pool.make(z.sb_vpos.h[k], (const char*)r.fa.seq + z.sb_vrempos.h[k], (size_t)nrem,
                                     (const char*)blob + z.sb_vaddoff.h[k], (size_t)nadd, z.sb_vsupp.h[k], PLATYPUS_VAR);
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
                    { PROF("s5.build"); prePosterior.add(r, w, false); }  // (the posterior kernel's inputs for this window, while it is in the caches)
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
    st.n_regions_stage_b_device += (int64_t)nR - nHostRegions; st.n_regions_stage_b_host += nHostRegions; st.n_windows_stage_b_host += nHostWindows; st.n_regions_dict_replay_device += nReplayed;
}

}  // namespace plathost
