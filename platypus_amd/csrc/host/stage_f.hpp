// stage_f.hpp -- F: INFO / FILTER arithmetic and record text
// (native region loop, libplat_caller.so: see region_caller.cpp for the stage map and the reference citations)
#pragma once
#include "chunk.hpp"

namespace plathost {

// outputRefCall (variantcaller.pyx:764-867) for [windowStart, windowEnd): QUAL 0 without coverage somewhere in the block; else the
// phred-scaled beta-binomial p-value of seeing no variant read at the block's smallest coverage, capped -- when the block holds
// candidates -- by the best candidate's posterior under a flat prior (maxPost).  nReads: the samples' reads between the window
// pointers as the loop last left them (the reference does not move them for a block).  Returns false when the reference raises here
// (an infinite QUAL: logged and skipped by its try/except).
inline bool Chunk::refCallLine(const RegionWork& r, std::string& out, int windowStart, int windowEnd, const std::vector<int>& nReads, bool hasVariants, double maxPost) const {
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

// vcfINFO (vcfutils.pyx:1226-1460), vcfFILTER (:1502-1627), outputCallToVCF (:338-599), VCF.write_data (vcf.py:710-739)
inline void Chunk::writeWindow(RegionWork& r, WindowWork& w, const std::vector<int64_t>& klo) {
    Slot& z = s;
    const int hapScore = z.o_hapscore.h[w.bw];
    for (size_t k = 0; k < w.info.size(); ++k) {
        VarInfo& d = w.info[k];
        const size_t sv = (size_t)w.firstStatVar + k;
        PROF("text.info");
        infoFieldsFromReadStats(d, z.s_counts.h + 16 * sv, z.s_ps.h + 2 * sv * (size_t)nInd, nInd, z.s_minq.h + z.s_moff.h[sv], z.s_nminq.h[sv],
                                z.infoOnDevice ? z.s_terms.h + 8 * sv : nullptr, z.infoOnDevice ? z.s_mmlq.h[sv] : -1);
        PROF("text.info.qd");
        if (d.TR > 0) {                                                // :1400-1409
            const double qual = d.PPnum;
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
        for (VarInfo& d : w.info) if (d.var == v) return d;                  // (nearly always the same object)
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
            bestQual = std::max(bestQual, d.PPint);
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
    // (the record's scratch lives from record to record of this thread: its storage is reused)
    static thread_local std::string ref;
    static thread_local std::vector<std::string> alt;
    static thread_local std::vector<char> line;
    struct SampleCall { int index1, index2, gq; long long gof; bool empty, noCall, refCall; };
    static thread_local std::vector<SampleCall> calls;
    for (size_t pi = 0; pi < positions.size(); ++pi) {
        PROF("text.record");
        int POS = positions[pi]->first;
        const VarList& variants = positions[pi]->second;
        const int nVariants = (int)variants.size();
        const size_t site = (size_t)w.firstSite + pi;
        { PROF("text.record.refalt"); refAndAlt(POS, variants, r.fa, ref, alt); }
        SmallVec<VarInfo*, 4> infos;                                       // vcfInfo[v] of the record's variants, in their order
        for (Variant* v : variants) infos.push_back(&infoOf(v));
        VarInfo& lead = *infos[0];
        int qual = 0;
        for (int k = 0; k < nVariants; ++k) { const int q = infos[(size_t)k]->PPint; if (k == 0 || q > qual) qual = q; }
        // per-sample calls first: MGOF (an INFO field) and the decision to write the record at all depend on them
        double maxGof = 0.0;
        int nNonRefCalls = 0;
        const int64_t NL = (int64_t)(nVariants + 1) * (nVariants + 2) / 2;
        const bool oneVar = nVariants == 1;
        calls.resize((size_t)nInd);
        for (int i = 0; i < nInd; ++i) {
            PROF("text.record.samplecol");
            SampleCall& c = calls[(size_t)i];
            const Ptrs& p = w.ptrs[(size_t)i];
            c.empty = p.ge - p.gs == 0;                                    // :498-500
            if (c.empty) continue;
            const size_t t = site * (size_t)nInd + (size_t)i;
            c.index1 = z.k_ph.h[2 * t]; c.index2 = z.k_ph.h[2 * t + 1];
            const double gtPost = z.k_out4.h[4 * t], nonRefPost = z.k_out4.h[4 * t + 1], refPost = z.k_out4.h[4 * t + 2], gofValue = z.k_out4.h[4 * t + 3];
            if (!(c.index1 == 0 && c.index2 == 0)) ++nNonRefCalls;
            c.noCall = false; c.refCall = false;
            if (oneVar) {                                                   // :524-542, :550-553
                if (phred(nonRefPost) < o.minPosterior) { if (phred(refPost) < o.minPosterior) c.noCall = true; else c.refCall = true; }
                if (lead.nReadsPerSample[(size_t)i] < o.minReads) c.noCall = true;
            }
            c.gof = (long long)gofValue; c.gq = phred(gtPost);
            maxGof = std::max(maxGof, gofValue);
        }
        const long long MGOF = (long long)py2_round2(maxGof);
        PROF("text.record.tail");
        if (!(nNonRefCalls > 0 || o.minPosterior == 0 || o.outputRefCalls == 1)) continue;
        trimLeftPadding(POS, ref, alt);
        bool plain = true;
        for (char c : ref) if (c != 'A' && c != 'C' && c != 'T' && c != 'G') { plain = false; break; }
        if (!plain) continue;                                           // :583-592
        // VCF.write_data
        PROF("text.record.write");
        // (written through a pointer into this thread's line buffer: a bound on the line's length first)
        const size_t chromLen = strlen(r.in->chrom);
        size_t bound = chromLen + ref.size() + lead.SC.size() + 800 + 96 * (size_t)nVariants;          // literals 130, 15 numbers of at most 32, 3 counts + FR + PP per variant
        for (const std::string& a : alt) bound += a.size() + 1;
        for (int k = 0; k < nVariants; ++k) bound += 12 * infos[(size_t)k]->filters.size() + infos[(size_t)k]->FRtext.size() + infos[(size_t)k]->PP.size();
        bound += (size_t)nInd * (64 + 33 * (size_t)NL + 24 * (size_t)nVariants);
        if (line.size() < bound) line.resize(bound + bound / 2);
        char* const line0 = line.data();
        char* p = line0;
        { PROF("tw.prefix");
        p = put_chars(p, r.in->chrom, chromLen); *p++ = '\t';
        p = put_int(p, POS + 1); p = put_lit(p, "\t.\t"); p = put_str(p, ref); *p++ = '\t';
        if (alt.empty()) *p++ = '.'; else for (size_t q = 0; q < alt.size(); ++q) { if (q) *p++ = ','; p = put_str(p, alt[q]); }
        *p++ = '\t'; p = put_int(p, qual); *p++ = '\t';
        {
            const char* names[64]; const char* ordered[64];
            int nf = 0;
            for (int k = 0; k < nVariants; ++k) for (const char* f : infos[(size_t)k]->filters) if (nf < 64) names[nf++] = f;
            if (nf == 0) p = put_lit(p, "PASS");
            else {
                const int m = py2_set_order_names(names, nf, ordered);
                for (int q = 0; q < m; ++q) { if (q) *p++ = ';'; p = put_chars(p, ordered[q], strlen(ordered[q])); }
            }
        }
        *p++ = '\t';
        }
        { PROF("tw.info");
        // INFO keys in sorted order: BRF FR HP HapScore MGOF MMLQ MQ NF NR PP QD SC SbPval Source TC TCF TCR TR WE WS
        p = put_lit(p, "BRF="); p = lead.BRF.put(p);
        p = put_lit(p, ";FR="); for (int k = 0; k < nVariants; ++k) { if (k) *p++ = ','; p = put_str(p, infos[(size_t)k]->FRtext); }
        p = put_lit(p, ";HP="); p = Num::I(lead.HP).put(p);
        p = put_lit(p, ";HapScore="); p = Num::I(lead.HapScore).put(p);
        p = put_lit(p, ";MGOF="); p = Num::I(MGOF).put(p);
        p = put_lit(p, ";MMLQ="); p = Num::I(lead.MMLQ).put(p);
        p = put_lit(p, ";MQ="); p = lead.MQ.put(p);
        p = put_lit(p, ";NF="); for (int k = 0; k < nVariants; ++k) { if (k) *p++ = ','; p = Num::I(infos[(size_t)k]->NF).put(p); }
        p = put_lit(p, ";NR="); for (int k = 0; k < nVariants; ++k) { if (k) *p++ = ','; p = Num::I(infos[(size_t)k]->NR).put(p); }
        p = put_lit(p, ";PP="); for (int k = 0; k < nVariants; ++k) { if (k) *p++ = ','; p = put_str(p, infos[(size_t)k]->PP); }
        p = put_lit(p, ";QD="); p = lead.QD.put(p);
        p = put_lit(p, ";SC="); p = put_chars(p, lead.SC.data(), lead.SC.size());
        p = put_lit(p, ";SbPval="); p = lead.SbPval.put(p);
        p = put_lit(p, ";Source="); for (size_t q = 0; q < lead.Source.size(); ++q) { if (q) *p++ = ','; p = put_chars(p, lead.Source[q], strlen(lead.Source[q])); }
        p = put_lit(p, ";TC="); p = Num::I(lead.TC).put(p);
        p = put_lit(p, ";TCF="); p = Num::I(lead.TCF).put(p);
        p = put_lit(p, ";TCR="); p = Num::I(lead.TCR).put(p);
        p = put_lit(p, ";TR="); for (int k = 0; k < nVariants; ++k) { if (k) *p++ = ','; p = Num::I(infos[(size_t)k]->TR).put(p); }
        p = put_lit(p, ";WE="); p = Num::I(w.endPos).put(p);
        p = put_lit(p, ";WS="); p = Num::I(w.startPos).put(p);
        p = put_lit(p, "\tGT:GL:GOF:GQ:NR:NV");
        }
        { PROF("tw.samples");
        // per-sample columns GT : GL : GOF : GQ : NR : NV, written in place; format_formatdata(key=False) then drops the trailing entries made only
        // of "," and "." -- GT "./." can only be dropped when everything after it is, and the integers after it never are
        for (int i = 0; i < nInd; ++i) {
            const SampleCall& c = calls[(size_t)i];
            *p++ = '\t';
            if (c.empty) { p = put_lit(p, "./.:0,0,0:0:0:0:0"); continue; }
            if (c.noCall) p = put_lit(p, "./.");
            else if (c.refCall) p = put_lit(p, "0/0");
            else { p = put_int(p, c.index1); *p++ = '/'; p = put_int(p, c.index2); }
            *p++ = ':';
            if (oneVar) {
                const double* lik = z.k_lik.h + klo[site] + (int64_t)i * NL;
                double top = lik[0];
                for (int64_t q = 1; q < NL; ++q) top = std::max(top, lik[q]);
                for (int64_t q = 0; q < NL; ++q) {                   // (FORMAT fields have no numeric missing value: -1.0 stays -1.0)
                    if (q) *p++ = ',';
                    p = put_py2_str(p, py2_round2(log10(std::max(lik[q] / top, 1e-300))));
                }
            } else p = put_lit(p, "-1,-1,-1");
            *p++ = ':'; p = put_int(p, c.gof);
            *p++ = ':'; p = put_int(p, c.gq);
            *p++ = ':';
            for (int k = 0; k < nVariants; ++k) { if (k) *p++ = ','; p = put_int(p, infos[(size_t)k]->nReadsPerSample[(size_t)i]); }
            *p++ = ':';
            for (int k = 0; k < nVariants; ++k) { if (k) *p++ = ','; p = put_int(p, infos[(size_t)k]->nVarReadsPerSample[(size_t)i]); }
        }
        }
        PROF("tw.append");
        *p++ = '\n';
        out.append(line0, (size_t)(p - line0));
        ++w.nRecords;
    }
}

}  // namespace plathost
