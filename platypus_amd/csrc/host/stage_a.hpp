// stage_a.hpp -- A: the chunk's read table, the candidate scan (+ merge, + stage B on the device launched behind it), the assembler tiles
// (native region loop, libplat_caller.so: see region_caller.cpp for the stage map and the reference citations)
#pragma once
#include "chunk.hpp"

namespace plathost {

// -- A: one device table for every read of the chunk; layout: all `reads` of every (region, sample), then all badReads, then all brokenMates
inline void Chunk::uploadReads() {
    size_t nReads[3] = {0, 0, 0}, nBytes[3] = {0, 0, 0}, nCig[3] = {0, 0, 0}, nExc = 0;
    bool anyPacked = false, allPacked = true, excRegular = true;
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
                else if (t.n_reads) allPacked = false;
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
                    for (size_t e = 0; e < ne; ++e) {
                        z.t_excidx.h[eo + e] = t.exc_index[e] + (int64_t)bo; z.t_excb.h[eo + e] = t.exc_base[e]; z.t_excq.h[eo + e] = t.exc_qual[e];
                        const uint8_t eb = t.exc_base[e];
                        excRegular = excRegular && (eb == 'A' || eb == 'C' || eb == 'G' || eb == 'T' || eb == 'N');
                    }
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
        // the bases' 2-bit codes next to the bytes when every read of the chunk comes out of a packed table and no exception carries a byte other than
        // A, C, G, T, N (the promise plat_candidates_batch_codes asks for); a device library without the entry point: the byte scan
        static const bool noCodes = getenv("PLAT_CALLER_NO_CODES") != nullptr;         // (measurements / tests: the byte scan)
        int rcu = PLAT_ERR_UNSUPPORTED;
        if (allPacked && excRegular && !noCodes) {
            z.t_codes.reserve(z.ctx, (bo + 15) / 16 + 16, false);
            rcu = plat_unpack_reads_pieces_codes(z.ctx, (int)packed.size(), (int64_t)most, z.t_pieces.d, z.t_seq.d, z.t_qual.d, z.t_codes.d, (int64_t)bo, (int64_t)eo,
                                                 z.t_excidx.d, z.t_excb.d, z.t_excq.d, z.stream);
            if (rcu != PLAT_ERR_UNSUPPORTED) ck(rcu, "plat_unpack_reads_pieces_codes");
        }
        readCodes = rcu == PLAT_OK;
        if (!readCodes)
            ck(plat_unpack_reads_pieces(z.ctx, (int)packed.size(), (int64_t)most, z.t_pieces.d, z.t_seq.d, z.t_qual.d, (int64_t)bo, (int64_t)eo, z.t_excidx.d, z.t_excb.d,
                                        z.t_excq.d, z.stream), "plat_unpack_reads_pieces");
        for (const Pending& p : packed) tabPackedBytes += (int64_t)p.nb;
    }
    tabBlobBytes = (int64_t)bo;
    nGood = nReads[0]; nScan = scan; nBad = nReads[1]; nBroken = nReads[2];
    std::lock_guard<std::mutex> g(stMutex);
    st.n_reads += (int64_t)N;
    st.input_bytes += (int64_t)inBytes;
}

// -- A2: VariantCandidateGenerator.addCandidatesFromReads over the `reads` of every (region, sample) (variant.pyx:459-751)
inline void Chunk::scanCandidates() {
    Slot& z = s;
    std::vector<int64_t> refoff{0};
    std::vector<int32_t> rss, clen, scanbegin, scanlongest;
    // every region with its contig on the device already: the reference windows are put together there (plat_copy_pieces)
    bool refResident = !regions.empty();
    for (RegionWork* r : regions) refResident = refResident && r->in->dev_contig_seq != nullptr;
    std::vector<plat_unpack_piece> pieces;
    size_t blobLen = 0, mostRef = 0;
    for (RegionWork* r : regions)
        for (size_t i = 0; i < r->samples.size(); ++i) {
            scanbegin.push_back((int32_t)r->samples[i].reads.base); scanlongest.push_back(r->samples[i].reads.longest);
            const int64_t a = std::max<int64_t>(0, (int64_t)r->in->start - 2000);                   // variant.pyx:486-488
            const int64_t e = std::min<int64_t>((int64_t)r->in->end + 2000, r->fa.len - 1);
            if (e < a) throw WindowError("Cannot have beginPos > endPos in getSequence");
            if (refResident) pieces.push_back(plat_unpack_piece{r->in->dev_contig_seq + a, (int64_t)blobLen, e - a});
            blobLen += (size_t)(e - a); mostRef = std::max(mostRef, (size_t)(e - a));
            refoff.push_back((int64_t)blobLen);
            rss.push_back((int32_t)a); clen.push_back((int32_t)r->fa.len);
        }
    {
        Layout L;
        scanbegin.push_back((int32_t)nGood);
        L.add(z.c_refoff, refoff.size()); L.add(z.c_rss, rss.size()); L.add(z.c_clen, clen.size());
        L.add(z.c_scanbegin, scanbegin.size()); L.add(z.c_scanlongest, scanlongest.size());
        if (refResident) L.add(z.c_pieces, pieces.size() + 1); else L.add(z.c_ref, blobLen + PLAT_BLOB_PAD);
        L.commit(z, z.a_cin);
        fill(z, z.c_refoff, refoff); fill(z, z.c_rss, rss); fill(z, z.c_clen, clen); fill(z, z.c_scanbegin, scanbegin); fill(z, z.c_scanlongest, scanlongest);
        if (refResident) {
            for (size_t q = 0; q < pieces.size(); ++q) z.c_pieces.h[q] = pieces[q];
            z.c_refdev.reserve(z.ctx, blobLen + PLAT_BLOB_PAD, false, true, z.stream);
            refDev = z.c_refdev.d;
        } else {
            size_t at = 0;
            for (RegionWork* r : regions)
                for (size_t i = 0; i < r->samples.size(); ++i) {
                    const int64_t a = std::max<int64_t>(0, (int64_t)r->in->start - 2000), e = std::min<int64_t>((int64_t)r->in->end + 2000, r->fa.len - 1);
                    memcpy(z.c_ref.h + at, r->fa.seq + a, (size_t)(e - a));
                    at += (size_t)(e - a);
                }
            memset(z.c_ref.h + blobLen, 0, PLAT_BLOB_PAD);
            refDev = z.c_ref.d;
        }
        L.upload(z, z.a_cin);
        if (refResident) ck(plat_copy_pieces(z.ctx, (int)pieces.size(), (int64_t)mostRef, z.c_pieces.d, z.c_refdev.d, z.stream), "plat_copy_pieces");
    }
    refBlob.clear();                                                    // (the host's copy of the windows is made when a host stage asks for it: hostRefBlob)
    if (nGood == 0) { hostTally = true; deviceB = false; return; }      // nothing to scan: the (empty) host tally
    plat_candidate_batch cb;
    memset(&cb, 0, sizeof cb);
    cb.n_regions = nScan; cb.n_reads = (int32_t)nGood;
    cb.ref_seq = refDev; cb.ref_off = z.c_refoff.d; cb.ref_seq_start = z.c_rss.d; cb.contig_len = z.c_clen.d;
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
        int rcs = PLAT_ERR_UNSUPPORTED;
        if (readCodes) {                                                // the scan on 2-bit codes: the reference blob's codes first (a few MB per chunk)
            z.c_refcodes.reserve(z.ctx, (blobLen + 15) / 16 + 16, false); z.c_refirr.reserve(z.ctx, (size_t)nScan + 1, false);
            rcs = plat_ref_codes(z.ctx, nScan, refDev, z.c_refoff.d, (int64_t)blobLen, z.c_refcodes.d, z.c_refirr.d, z.stream);
            if (rcs == PLAT_OK)
                rcs = plat_candidates_batch_codes(z.ctx, &cb, z.t_codes.d, z.c_refcodes.d, z.c_refirr.d, o.minFlank, o.minBaseQual, o.genSNPs, o.genIndels, maxPerRead,
                                                  z.t_region.d, z.c_rec.d, z.c_cnt.d, z.c_status.d, z.stream);
            if (rcs != PLAT_ERR_UNSUPPORTED) ck(rcs, "plat_candidates_batch_codes");
        }
        if (rcs != PLAT_OK)
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
                // (host stage B may have to replay a region's dictionaries from the scan's records: a chunk of small regions brings them
                //  along now -- one copy behind the merge's -- instead of two copies and a wait per region that needs them later)
                const bool recordsAlong = !deviceB && LO.total <= ((size_t)4 << 20);
                if (deviceB) LM.downloadFirst(z, z.a_mout, 1);                              // (only the counts: the candidates stay on the device)
                else { LM.download(z, z.a_mout); if (recordsAlong) LO.download(z, z.a_cout); }
                z.sync("candidate scan");
                for (int g = 0; g < nScan; ++g) {
                    const int st_ = z.m_n.h[2 * g + 1];
                    if (st_ == PLAT_ERR_BAD_INPUT) throw DeviceError(PLAT_ERR_BAD_INPUT, "a read reaches outside the reference window handed over, or read pointers out of order");
                    if (st_ <= -(1 << 20)) need = std::max(need, -st_ - (1 << 20));
                    else if (st_ != 0) hostTally = true;                // more distinct records / candidates than the kernel takes
                }
                if (!need && !hostTally) { recordsOnHost = recordsAlong; break; }
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

// assemble=1: the tiles of every region of the chunk in ONE assembler launch (variantcaller.pyx:496-519).  Two halves: assembleLaunch() puts the
// batch together, sizes it itself (plat_assemble_batch_async: no read-back, no wait) and leaves launch + download on the stream BEFORE the
// candidate scan's work, so that the scan's one wait covers the assembler too; assembleCollect() turns the tuples into Variants.
inline void Chunk::assembleLaunch() {
    asmN = 0;
    if (!o.assemble) return;
    Slot& z = s;
    const auto t0 = Clock::now();
    const int size = o.assemblyRegionSize;
    if (size <= 0) throw DeviceError(PLAT_ERR_INVALID, "assemblyRegionSize");
    const int shift = std::max(100, std::min(1000, size / 2));
    std::vector<Tile>& tiles = asmTiles;
    tiles.clear();
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
        plat_assembly_batch& ab = asmBatch;
        memset(&ab, 0, sizeof ab);
        ab.n_regions = nT; ab.n_reads = (int32_t)nR;
        ab.ref_seq = z.as_ref.d; ab.ref_off = z.as_refoff.d; ab.ref_start = z.as_refstart.d; ab.assem_start = z.as_astart.d; ab.assem_end = z.as_aend.d;
        ab.reg_read_begin = z.as_rbegin.d; ab.read_seq = z.as_seq.d; ab.read_qual = z.as_qual.d; ab.read_off = z.as_roff.d;
        memset(&asmHints, 0, sizeof asmHints);
        for (int g = 0; g < nT; ++g) {
            const int64_t rl = refoff[(size_t)g + 1] - refoff[(size_t)g], nr = rbegin[(size_t)g + 1] - rbegin[(size_t)g];
            const int64_t bytes = roff[(size_t)rbegin[(size_t)g + 1]] - roff[(size_t)rbegin[(size_t)g]];
            asmHints.max_ref_len = std::max<int32_t>(asmHints.max_ref_len, (int32_t)rl);
            asmHints.max_reads_per_region = std::max<int32_t>(asmHints.max_reads_per_region, (int32_t)nr);
            asmHints.max_positions = std::max<int64_t>(asmHints.max_positions, rl + 2 + bytes + 2 * nr);
        }
        asmN = nT;
        assembleEnqueue();
    }
    std::lock_guard<std::mutex> g(stMutex);
    st.seconds_assemble += secs(t0, Clock::now());
}

inline void Chunk::assembleEnqueue() {
    Slot& z = s;
    const int nT = asmN;
    Layout LO;
    LO.add(z.as_cnt, (size_t)nT); LO.add(z.as_status, (size_t)nT); LO.add(z.as_vpos, (size_t)nT * asmMaxVars); LO.add(z.as_nrem, (size_t)nT * asmMaxVars);
    LO.add(z.as_nadd, (size_t)nT * asmMaxVars); LO.add(z.as_off, (size_t)nT * asmMaxVars); LO.add(z.as_blob, (size_t)nT * asmBlob);
    LO.commit(z, z.a_asout);
    ck(plat_assemble_batch_async(z.ctx, &asmBatch, &asmHints, o.assemblerKmerSize, o.minBaseQual, o.minReads * o.minBaseQual, o.noCycles, asmMaxVars, asmBlob,
                                 z.as_cnt.d, z.as_vpos.d, z.as_nrem.d, z.as_nadd.d, z.as_off.d, z.as_blob.d, z.as_status.d, z.stream), "plat_assemble_batch_async");
    LO.download(z, z.a_asout);
}

inline void Chunk::assembleCollect() {
    if (!o.assemble || asmN == 0) return;
    Slot& z = s;
    const auto t0 = Clock::now();
    const int nT = asmN;
    const std::vector<Tile>& tiles = asmTiles;
    {
        for (;;) {                                                       // room per tile grows until every tile's variants fit
            z.sync("assembler");                                         // (already over when the scan's wait covered it)
            bool over = false;
            for (int g = 0; g < nT; ++g) {
                if (z.as_status.h[g] == PLAT_ERR_OVERFLOW) over = true;
                else if (z.as_status.h[g] != 0) throw DeviceError(z.as_status.h[g], "plat_assemble_batch(tile)");
            }
            if (!over) break;
            if (asmMaxVars >= (1 << 14)) throw DeviceError(PLAT_ERR_OVERFLOW, "plat_assemble_batch(tile)");
            asmMaxVars *= 4; asmBlob *= 4;
            assembleEnqueue();
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

}  // namespace plathost
