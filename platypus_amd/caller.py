"""The region loop around the hot path, on in-memory read buffers (BAM-free): candidates -> windows -> haplotypes ->
likelihoods / EM / posteriors -> VCF records.  Same names and argument meaning as the reference:

    callVariantsInWindow        src/cython/variantcaller.pyx:74-141
    generateVariantsInRegion    src/cython/variantcaller.pyx:412-531   (BAM candidates + assembler tiles; source VCFs are not built)
    doWeNeedToAssembleThisRegion                          :276-321
    callVariantsInRegion        src/cython/variantcaller.pyx:535-615   (loadBAMData replaced by the caller's read buffers)

in two shapes.  `callVariantsInRegion` walks the windows one at a time exactly as the reference does (a handful of device
calls per window).  `callVariantsInRegions` is the shape the device wants: the same windows, the same records, but every
device stage runs ONCE over all windows of all regions -- one candidate scan, one likelihood / genotype / HapScore / EM
pass, one posterior pass, one read-statistics pass, one genotype-marginalisation pass -- with the host logic (window
generation, haplotype enumeration, dictionary assembly, text) in between.  Both produce the same text
(tests/test_gpu_caller.py)."""
from types import SimpleNamespace

import numpy as np

from . import hostapi as H
from .regionprep import (WindowGenerator, computeVariantReadSupportFrac, filterVariants, filterVariantsByCoverage,
                         getHaplotypesInWindow, leftNormaliseIndel)
from .vcfrecords import (genotypeCallSites, genotypeCallTuples, getHaplotypeInfo, infoStatsWindow, outputCallToVCF, outputRefCall,
                         py2_dict_order, vcfFILTER, vcfINFO)


def _unsupported(options):
    if getattr(options, "sourceFile", None):
        raise NotImplementedError("candidates from a source VCF (variantutils.VariantCandidateReader) are not built")
    if getattr(options, "HLATyping", 0):
        raise NotImplementedError("HLA mode")


# ---- candidates -> filtered, left-normalised variants ---------------------------------------------------------------------

def _candidateRegion(gen, reads):
    return dict(ref=gen.pyRefSeq, ref_seq_start=gen.refSeqStart, contig_len=gen.refFile.refs[gen.rname].SeqLength,
                reads=[dict(seq=r.seq, qual=r.qual, pos=r.pos, flag=r.bitFlag, cigar=r.cigarOps) for r in reads])


def doWeNeedToAssembleThisRegion(readBuffers, chrom, start, end, options, refSeq=None):
    """variantcaller.pyx:276-321: window pointers to the tile, then: always (assembleAll, the default), or when a sample's reads
    carry more than two alignment gaps each or more than a tenth of them are improperly paired."""
    for b in readBuffers:
        b.setWindowPointers(start, end)
    if options.assembleAll:
        return True
    for b in readBuffers:
        n = float(b.reads.windowEnd - b.reads.windowStart)
        nBad = float(b.badReads.windowEnd - b.badReads.windowStart)
        if n == 0:
            continue
        if b.countAlignmentGaps() / n > 2 or b.countImproperPairs() / (n + nBad) > 0.1:
            return True
    return False


def _assemblerVariants(regions, refFile, options):
    """The assembler part of generateVariantsInRegion (:496-519) for every region: tiles of assemblyRegionSize starting
    every max(100, min(1000, size/2)) bases, ALL tiles of ALL regions assembled in one device batch."""
    size = options.assemblyRegionSize
    shift = max(100, min(1000, size // 2))
    tiles, owner, chroms = [], [], []
    for k, (chrom, start, end, buffers) in enumerate(regions):
        for assemStart in range(start, end, shift):
            assemEnd = min(assemStart + size, end)
            refStart, refEnd = max(0, assemStart - size), assemEnd + size
            refSeq = refFile.getSequence(chrom, refStart, refEnd)
            if doWeNeedToAssembleThisRegion(buffers, chrom, assemStart, assemEnd, options, refSeq):
                tiles.append(H.assemblyRegion(assemStart, assemEnd, refStart, refEnd, buffers, refSeq, options))
                owner.append(k); chroms.append(chrom)
    out = [[] for _ in regions]
    if tiles:
        for k, vs in zip(owner, H.assembleRegions(chroms, tiles, options)):
            out[k].extend(vs)
    return out


def _py2_heap_order(heap):
    """The values of a VariantCandidateGenerator.variantHeap in the order the reference's Python-2 dictionary would yield them: keys
    hash as (refName, refPos, removed, added) (variant.pyx:270-280), were inserted in first-occurrence order and never deleted."""
    from .vcfrecords import py2_dict_slot_order, py2_variant_hash
    vs = list(heap.values())
    return [vs[k] for k in py2_dict_slot_order([py2_variant_hash(v.refName, v.refPos, v.removed, v.added) for v in vs])]


def generateVariantsInRegions(regions, refFile, options):
    """generateVariantsInRegion for a list of (chrom, start, end, readBuffers): ONE candidate scan on the device for every
    sample of every region, then the reference's per-sample support filter, merge, left-normalisation and filterVariants.
    Returns (per-region variant lists, per-region rlen): options.rlen follows the longest read of EACH region as in the
    reference (:470-488: set per region before left-normalisation, kept from the region before when a region has no reads);
    the value a region was prepared with is the one its windows and haplotypes must be built with, so it is returned and
    options.rlen is left at the last region's value, as after the reference's last call."""
    mk = lambda chrom, start, end: H.VariantCandidateGenerator((chrom, start, end), refFile, options.minMapQual, options.minFlank,
                                                               options.minBaseQual, options.maxReads, options.rlen, options,
                                                               options.verbosity, options.genSNPs, options.genIndels)
    out = [[] for _ in regions]
    if not options.getVariantsFromBAMs:
        rlens = [options.rlen] * len(regions)
        if not options.assemble:
            return out, rlens
        merged = _assemblerVariants(regions, refFile, options)
        for k, cands in enumerate(merged):
            norm = sorted(leftNormaliseIndel(v, refFile, options.rlen) for v in cands)
            out[k] = filterVariants(norm, refFile, options.rlen, options.minReads, options.maxSize, options.verbosity, options)
        return out, rlens
    gens = [[mk(chrom, start, end) for _ in buffers] for chrom, start, end, buffers in regions]
    scans = [_candidateRegion(g, b.reads.array) for (_, _, _, buffers), gs in zip(regions, gens) for g, b in zip(gs, buffers)]
    found = iter(H.get_engine().candidates(scans, options.minFlank, options.minBaseQual, options.genSNPs, options.genIndels))
    heaps, rlens = [], []
    rlen = options.rlen
    for (chrom, start, end, buffers), gs in zip(regions, gens):
        longest = 0
        for g, b in zip(gs, buffers):
            longest = max(longest, b.reads.getLengthOfLongestRead())
            tally = {}                                                          # equal records merge into one variant with
            for pos, removed, added, _ in next(found):                          # the number of reads showing it (addVariantToList)
                key = (pos, removed, added)
                tally[key] = tally.get(key, 0) + 1
            for (pos, removed, added), n in tally.items():
                g.addVariantToList(H.Variant(chrom, pos, removed, added, n, H.PLATYPUS_VAR))
        heaps.append((gs, buffers, chrom, start, end))
        if longest > 0:                                                          # :476-488
            rlen = options.maxSize if longest >= options.maxSize else longest
        rlens.append(rlen)
    extras = _assemblerVariants(regions, refFile, options) if options.assemble else [[] for _ in regions]

    def clone(v):
        c = H.Variant(v.refName, v.refPos, v.removed, v.added, v.nSupportingReads, v.varSource)
        c.bamMinPos, c.bamMaxPos = v.bamMinPos, v.bamMaxPos
        return c

    def candidates(k, exact):
        """generateVariantsInRegion :456-531 for region k.  :456-467: per-sample support, indels always.  Both candidate dictionaries are
        walked in the order a Python-2 dict holds Variant keys (`variantHeap.iteritems()`, `sorted(variantHeap.values())`) when `exact`,
        in first-occurrence order otherwise; on copies, so that the second way starts from untouched variants."""
        gs, buffers, chrom, start, end = heaps[k]
        everyone = mk(chrom, start, end)
        for g, b in zip(gs, buffers):
            for v in (_py2_heap_order(g.variantHeap) if exact else g.variantHeap.values()):
                if computeVariantReadSupportFrac(v, b) >= options.minVarFreq or v.nAdded != v.nRemoved:
                    everyone.addVariantToList(clone(v))
        cands = sorted(_py2_heap_order(everyone.variantHeap) if exact else everyone.variantHeap.values())
        cands.extend(clone(v) for v in extras[k])                                # rawBamVariants + assemblerVariants (:521)
        norm = sorted(leftNormaliseIndel(v, refFile, rlens[k]) for v in cands)
        return norm, filterVariants(norm, refFile, rlens[k], options.minReads, options.maxSize, options.verbosity, options)

    def order_can_matter(norm, kept):
        """`sorted` is stable: candidates that compare equal (two alleles of one type and length at one position) stay in dictionary
        order; every other order is decided by the keys.  It reaches the result where two KEPT variants compare equal, or where a run
        of equal keys holds a variant twice (equal neighbours merge: who is whose neighbour depends on it) next to a different one."""
        tied = lambda a, c: not (a < c) and not (c < a)
        if any(tied(a, c) for a, c in zip(kept, kept[1:])):
            return True
        i = 0
        while i < len(norm):
            e = i + 1
            while e < len(norm) and not (norm[i] < norm[e]):
                e += 1
            if e - i >= 3:
                pairs = [(norm[x] == norm[y]) for x in range(i, e) for y in range(x + 1, e)]
                if any(pairs) and not all(pairs):
                    return True
            i = e
        return False

    for k in range(len(regions)):
        norm, kept = candidates(k, False)
        if order_can_matter(norm, kept):                                         # only such a region pays for replaying the dictionaries
            norm, kept = candidates(k, True)
        out[k] = kept
    if rlens:
        options.rlen = rlens[-1]
    return out, rlens


def generateVariantsInRegion(chrom, start, end, refFile, options, readBuffers):
    return generateVariantsInRegions([(chrom, start, end, readBuffers)], refFile, options)[0][0]


# ---- one window -----------------------------------------------------------------------------------------------------------

def _prepareWindow(window, options, refFile, readBuffers):
    """The host part of callVariantsInWindow up to Population.setup (:74-136).  Returns None for a window the reference
    leaves without calling, else (variants, haplotypes, genotypes)."""
    chrom, variants = window["chromosome"], window["variants"]
    windowStart, windowEnd = window["startPos"], window["endPos"]
    refHaplotype = H.Haplotype(chrom, windowStart, windowEnd, (), refFile, options.rlen, options)
    nReads = 0
    for b in readBuffers:
        b.setWindowPointers(windowStart, windowEnd)
        nReads += b.reads.windowEnd - b.reads.windowStart
    if nReads == 0 or nReads > options.maxReads:
        return None
    if len(variants) > options.maxVariants:
        if options.skipDifficultWindows:
            return None
        if options.filterVarsByCoverage:
            filterVariantsByCoverage(window, chrom, windowStart, windowEnd, refFile, options, variants, refHaplotype, readBuffers)
    haps = getHaplotypesInWindow(window, nReads, refFile, options.maxReads, options.minMapQual, options.minBaseQual, options.maxHaplotypes,
                                 options.maxVariants, options.rlen, options.verbosity, readBuffers, options)
    unique = H.mergeHaplotypes([refHaplotype] + haps, refFile)
    if len(unique) <= 1:
        return None
    return variants, unique, H.generateAllGenotypesFromHaplotypeList(unique)      # `variants`: the unfiltered list, as there


def callVariantsInWindow(window, options, refFile, readBuffers, pop):
    pop.reset()
    pop.refFile = refFile
    prep = _prepareWindow(window, options, refFile, readBuffers)
    if prep is None:
        return
    variants, haps, genotypes = prep
    pop.setup(variants, haps, genotypes, len(readBuffers), options.verbosity, readBuffers)
    pop.call(100, 1)


def _windowsOfRegion(chrom, start, end, refFile, options, variants, windowGenerator):
    maxContigPos = refFile.refs[chrom].SeqLength - 1
    for window in windowGenerator.WindowsAndVariants(chrom, start, end, maxContigPos, variants, options):
        if len(window["variants"]) == 0:
            if options.outputRefCalls:                                           # a reference-call block (:605-607)
                yield window
            continue
        if window["endPos"] - window["startPos"] > options.maxSize:              # :566-568
            continue
        yield window


def _windowFailed(window, exc):
    import logging
    logging.getLogger("Log").exception("Problem calling variants in window %s:%d-%d. Skipping it: %r", window["chromosome"],
                                       window["startPos"], window["endPos"], exc)


def _refCallBlocksBetween(varsByPos, chrom, options):
    """:584-603: reference-call blocks between the called positions of one window, walked in the order a Python-2 dictionary holds
    its integer keys (pop.varsByPos.iteritems())."""
    last = None
    for index, pos in enumerate(py2_dict_order(list(varsByPos.keys()))):
        these = varsByPos[pos]
        if index > 0:
            lastVarPos = max(v.maxRefPos for v in last)
            nextVarPos = min(v.minRefPos for v in these) + 1
            if nextVarPos - lastVarPos > 1:
                for blockStart in range(lastVarPos + 1, nextVarPos, options.refCallBlockSize):
                    blockEnd = min(blockStart + options.refCallBlockSize, nextVarPos - 1)
                    if blockStart != blockEnd:
                        yield dict(chromosome=chrom, startPos=blockStart, endPos=blockEnd, variants=[], nVar=0)
        last = these


def callVariantsInRegion(chrom, start, end, readBuffers, refFile, options, vcfFile, outputFile, windowGenerator=None, pop=None):
    """Window by window, as the reference."""
    _unsupported(options)
    windowGenerator = windowGenerator or WindowGenerator()
    pop = pop or H.Population(options)
    variants = generateVariantsInRegion(chrom, start, end, refFile, options, readBuffers)
    for windowIndex, window in enumerate(_windowsOfRegion(chrom, start, end, refFile, options, variants, windowGenerator)):
        try:                                                                     # :568-615: a failing window is logged and skipped
            if len(window["variants"]) > 0:
                callVariantsInWindow(window, options, refFile, readBuffers, pop)
            if len(window["variants"]) > 0 and len(pop.variantPosteriors) > 0:
                outputCallToVCF(pop.varsByPos, pop.vcfInfo, pop.vcfFilter, pop.haplotypes, pop.genotypes, pop.frequencies,
                                pop.genotypeLikelihoods, pop.goodnessOfFitValues, pop.haplotypeIndexes, pop.readBuffers, pop.nIndividuals,
                                vcfFile, refFile, outputFile, options, pop.variants, window["startPos"], window["endPos"], population=pop)
                if options.outputRefCalls and len(pop.varsByPos) > 1:
                    for block in _refCallBlocksBetween(pop.varsByPos, chrom, options):
                        outputRefCall(chrom, pop, vcfFile, refFile, outputFile, windowIndex, block, options, readBuffers)
            elif options.outputRefCalls:
                outputRefCall(chrom, pop, vcfFile, refFile, outputFile, windowIndex, window, options, readBuffers)
        except Exception as exc:
            _windowFailed(window, exc)


# ---- all windows of all regions at once -----------------------------------------------------------------------------------

def callWindowsBatched(specs, options, refFile):
    """Population.setup + call(computeVCFFields=1) for a list of windows in ONE pass per device stage.
    spec = dict(variants, haplotypes, genotypes, readBuffers (frozen per window)).  Returns one Population per spec with the
    same fields the per-window path fills, plus `_genotypeCalls` (the per-position 7-tuples of outputCallToVCF)."""
    if not specs:
        return []
    eng = H.get_engine()
    pops = []
    for sp in specs:
        p = H.Population(options)
        p.refFile = refFile
        p._bind(sp["variants"], sp["haplotypes"], sp["genotypes"], len(sp["readBuffers"]), sp["readBuffers"])
        pops.append(p)
    hb = H._pack_windows([([h.haplotypeSequence for h in p.haplotypes], p.haplotypes[0].startPos, p.haplotypes[0].endPos,
                           p.haplotypes[0].endBufferSize, p.readBuffers) for p in pops])
    db = eng.upload(hb)
    eng.call_windows(db, want_stats=False, calc_flank_score=int(options.calculateFlankScore))   # likelihoods + genotype likelihoods
    like, score = eng.haplotype_scores(db)
    eng.em(db, 100, int(options.useEMLikelihoods))
    eng.synchronize()
    setup_arrays = (db.gl.cpu().numpy(), db.logl.cpu().numpy(), db.gof.cpu().numpy(), db.loglik.cpu().numpy(), like, score)
    call_arrays = (db.freq.cpu().numpy(), db.em.cpu().numpy(), db.calls.cpu().numpy(), db.em_iters.cpu().numpy())
    var_w, masks, priors, owners = [], [], [], []
    for w, p in enumerate(pops):
        p._readSetup(db, w, *setup_arrays)
        p._readCall(*call_arrays)
        vs = p._distinctVariants()
        owners.append(len(vs))
        var_w += [w] * len(vs)
        masks += p._masks(vs)
        priors += [v.calculatePrior(refFile) for v in vs]
    post = eng.variant_posteriors(db, var_w, masks, priors) if var_w else np.zeros(0)
    at = 0
    for p, n in zip(pops, owners):
        p.computeVariantPosteriors(posteriors=post[at:at + n])
        p.vcfInfo, p.vcfFilter, p._genotypeCalls = {}, {}, []
        at += n
    live = [p for p in pops if len(p.variantPosteriors) > 0]
    if not live:
        return pops
    order = [list(getHaplotypeInfo(p.haplotypes, p.variantPosteriors, p.frequencies, p.nHaplotypes).keys()) for p in live]
    stats = eng.variant_read_stats([infoStatsWindow(vs, p.readBuffers, p.genotypeCalls) for p, vs in zip(live, order)],
                                   bad_reads_window=options.badReadsWindow, exact=options.countOnlyExactIndelMatches)
    sites, nsites = [], []
    for p in live:
        s_ = genotypeCallSites(p.varsByPos, p.haplotypes, p.variants, p._w)
        sites += s_
        nsites.append(len(s_))
    tuples = genotypeCallTuples(eng.genotype_calls(db, sites), hb.n_ind)
    at = 0
    for p, st, n in zip(live, stats, nsites):
        p.vcfInfo = vcfINFO(p.frequencies, p.variantPosteriors, p.genotypeCalls, p.genotypes, p.haplotypes, p.readBuffers,
                            p.nHaplotypes, options, refFile, hapScore=p.haplotypeScore, readStats=st)
        p.vcfFilter = vcfFILTER(p.genotypeCalls, p.haplotypes, p.vcfInfo, p.varsByPos, options)
        p._genotypeCalls = tuples[at:at + n]
        at += n
    return pops


class _RefCallBuffer:
    """What outputRefCall reads of a bamReadBuffer, with the window pointers as they stood when the reference would have written the
    block (it does not move them for a block without variants, so they are those of the window called before it)."""

    def __init__(self, buf):
        self.sample, self._buf = buf.sample, buf
        self.reads = SimpleNamespace(windowStart=buf.reads.windowStart, windowEnd=buf.reads.windowEnd)

    def countReadsCoveringRegion(self, start, end):
        return self._buf.countReadsCoveringRegion(start, end)


def callVariantsInRegions(regions, refFile, options, vcfFile, outputFile, windowGenerator=None):
    """regions: list of (chrom, start, end, readBuffers) with the same samples.  Records are written in region order, then
    window order, then position order -- the order callVariantsInRegion would write them one region after the other (reference-call
    blocks of --outputRefCalls=1 in their places between them)."""
    _unsupported(options)
    windowGenerator = windowGenerator or WindowGenerator()
    variants, rlens = generateVariantsInRegions(regions, refFile, options)
    specs, items = [], []                       # items: ("call", spec index) | ("ref", chrom, window, buffers as outputRefCall sees them, Population stand-in)
    for (chrom, start, end, buffers), vs, rlen in zip(regions, variants, rlens):
        options.rlen = rlen                                                      # the region's own value (see generateVariantsInRegions)
        nHapLast = 0                                                             # haplotypes of the last window set up in this region
        for window in _windowsOfRegion(chrom, start, end, refFile, options, vs, windowGenerator):
            if len(window["variants"]) == 0:                                     # reference-call block
                items.append(("ref", chrom, window, [_RefCallBuffer(b) for b in buffers], None))
                continue
            try:
                prep = _prepareWindow(window, options, refFile, buffers)         # (greedy haplotype filter: device calls of its own)
            except Exception as exc:                                             # :568-615: logged and skipped, as window by window
                _windowFailed(window, exc)
                continue
            if prep is not None:
                specs.append(dict(variants=prep[0], haplotypes=prep[1], genotypes=prep[2], window=window, chrom=chrom,
                                  readBuffers=[b.frozenWindow() for b in buffers], refBuffers=[_RefCallBuffer(b) for b in buffers]))
                items.append(("call", len(specs) - 1))
                nHapLast = len(prep[1])
            elif options.outputRefCalls:
                stub = H.Population(options)                                     # a Population that was reset and not set up for this window
                stub.nHaplotypes = nHapLast
                items.append(("ref", chrom, window, [_RefCallBuffer(b) for b in buffers], stub))
    try:
        pops = callWindowsBatched(specs, options, refFile)
    except Exception:
        # one window the device refuses would take every other window of the batch with it: call them one at a time instead,
        # so that only the failing ones are skipped
        pops = []
        for sp in specs:
            try:
                pops.append(callWindowsBatched([sp], options, refFile)[0])
            except Exception as exc:
                _windowFailed(sp["window"], exc)
                pops.append(None)
    for it in items:
        try:
            if it[0] == "ref":
                outputRefCall(it[1], it[4], vcfFile, refFile, outputFile, 0, it[2], options, it[3])
                continue
            sp, p = specs[it[1]], pops[it[1]]
            if p is None:
                continue
            window = sp["window"]
            if len(p.variantPosteriors) > 0:
                outputCallToVCF(p.varsByPos, p.vcfInfo, p.vcfFilter, p.haplotypes, p.genotypes, p.frequencies, p.genotypeLikelihoods,
                                p.goodnessOfFitValues, p.haplotypeIndexes, p.readBuffers, p.nIndividuals, vcfFile, refFile, outputFile,
                                options, p.variants, window["startPos"], window["endPos"], genotypeCalls=p._genotypeCalls)
                if options.outputRefCalls and len(p.varsByPos) > 1:
                    for block in _refCallBlocksBetween(p.varsByPos, sp["chrom"], options):
                        outputRefCall(sp["chrom"], p, vcfFile, refFile, outputFile, 0, block, options, sp["refBuffers"])
            elif options.outputRefCalls:
                outputRefCall(sp["chrom"], p, vcfFile, refFile, outputFile, 0, window, options, sp["refBuffers"])
        except Exception as exc:
            _windowFailed(it[2] if it[0] == "ref" else specs[it[1]]["window"], exc)
    return len(specs)
