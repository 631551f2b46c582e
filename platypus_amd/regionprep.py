"""Between the candidate scan and the calling windows ("... with window generation completes a BAM-free
region pipeline"): host logic, same names and argument meaning as the reference.

    leftNormaliseIndel                      src/cython/platypusutils.pyx:806-931
    filterVariants, filterVariantsByCoverage, computeVariantReadSupportFrac, getHaplotypesInWindow
                                            src/cython/variantFilter.pyx:98-171,359-373,571-622,626-650
    WindowGenerator                         src/python/window.py:18-238

Pinned by tests/golden/regionprep_cases.json.gz (outputs of the reference's own texts)."""
from .vcfrecords import ASSEMBLER_VAR, FILE_VAR, PLATYPUS_VAR


def leftNormaliseIndel(variant, refFile, maxReadLength):
    """Shift a pure insertion / deletion as far left as the surrounding sequence allows; anything else, and anything within
    100 bases of the contig start, is returned as it is.  bamMinPos / bamMaxPos of the new variant span the positions the
    indel can take (that is what the read statistics match reads against)."""
    from .hostapi import Variant
    nAdded, nRemoved = variant.nAdded, variant.nRemoved
    if nAdded == nRemoved or (nAdded > 0 and nRemoved > 0) or variant.refPos < 100:
        return variant
    window = max(nAdded, nRemoved) + maxReadLength
    seqMax = refFile.refs[variant.refName].SeqLength - 1
    windowMin, windowMax = max(1, variant.refPos - window), min(variant.refPos + window, seqMax)
    ref = refFile.getSequence(variant.refName, windowMin, windowMax)
    cut = variant.refPos - windowMin
    hap = ref[:cut + 1] + variant.added + ref[cut + nRemoved + 1:]
    if hap[-1] != ref[-1] and windowMax != seqMax:
        raise Exception("Variant %s not correctly normalised. \nRef = %s\nHap = %s" % (variant, ref, hap))
    n = min(len(ref), len(hap))
    fwd = next((i for i in range(n) if hap[i] != ref[i]), n)                      # rightmost placement: first mismatch from the left
    maxPos = windowMin + fwd + nRemoved
    for back in range(n):                                                          # leftmost: first mismatch from the right
        if hap[len(hap) - back - 1] == ref[len(ref) - back - 1]:
            continue
        newPos = windowMin + len(ref) - back - nRemoved - 1
        first = newPos - windowMin + 1
        newAdded = hap[first:first + nAdded] if nAdded > 0 else b""
        newRemoved = ref[first:first + nRemoved] if nRemoved > 0 else b""
        if len(newAdded) != nAdded or len(newRemoved) != nRemoved:
            raise Exception("Error in variant conversion to standard format")
        out = Variant(variant.refName, newPos, newRemoved, newAdded, variant.nSupportingReads, variant.varSource)
        out.bamMinPos, out.bamMaxPos = newPos, maxPos
        out.bamAdded, out.bamRemoved = variant.bamAdded, variant.bamRemoved
        if getattr(variant, "prior", None) is not None:
            out.prior = variant.prior
        return out
    return variant


def _onlyFromReads(source):
    return bool(source & PLATYPUS_VAR) and not (source & ASSEMBLER_VAR) and not (source & FILE_VAR)


def filterVariants(varList, refFile, maxReadLength, minSupport, maxDiff, verbosity, options):
    """Merge equal neighbours of a SORTED candidate list (supporting reads accumulate in the first of them) and drop weakly
    supported read-only candidates and oversized ones.  The last group is only tested against minSupport, as in the reference."""
    kept, last = [], None
    for v in varList:
        if last is None:
            last = v
        elif v == last:
            last.addVariant(v)
        else:
            size = max(last.nAdded, last.nRemoved)
            weak = _onlyFromReads(last.varSource) and ((last.nSupportingReads < minSupport and size < 15) or
                                                       (last.nSupportingReads < options.minReads and size >= 15))
            if not weak and size <= options.maxSize:
                kept.append(last)
            last = v
    if last is not None and not (last.nSupportingReads < minSupport and _onlyFromReads(last.varSource)):
        kept.append(last)
    return sorted(kept)


def computeVariantReadSupportFrac(variant, readBuffer):
    total = readBuffer.countReadsCoveringRegion(variant.refPos, variant.refPos + 1)
    return 0.0 if total == 0 else float(variant.nSupportingReads) / total


def filterVariantsByCoverage(thisWindow, chrom, windowStart, windowEnd, refFile, options, variants, refHaplotype, readBuffers):
    """Keep the options.maxVariants best supported variants of an over-full window (assembler-only variants first)."""
    top = max(v.nSupportingReads for v in variants)
    # (variants that neither precede nor follow each other -- two alleles at one position -- keep their relative order:
    #  a stable descending sort, exactly list.sort(reverse=True))
    ranked = sorted((((top + 1 if v.varSource == ASSEMBLER_VAR else v.nSupportingReads), v) for v in variants), reverse=True)
    thisWindow["variants"] = sorted(v for _, v in ranked[:options.maxVariants])


def getHaplotypesInWindow(window, nReads, refFile, maxCoverage, minMapQual, minBaseQual, maxHaplotypes, maxVariants, maxReadLength,
                          verbosity, readBuffers, options):
    from .hostapi import Haplotype, getFilteredHaplotypes
    chrom, start, end = window["chromosome"], window["startPos"], window["endPos"]
    refHaplotype = Haplotype(chrom, start, end, (), refFile, maxReadLength, options)
    if nReads == 0:
        return [refHaplotype]
    return getFilteredHaplotypes(chrom, start, end, refFile, options, window["variants"], refHaplotype, readBuffers)


class WindowGenerator:
    """window.py:18-238: group the sorted variants of a region into calling windows."""

    def getVariantsByPos(self, chromosome, start, end, sortedVariants):
        byPos = {}
        for v in sortedVariants:
            if v.refName == chromosome and start <= v.refPos < end:
                byPos.setdefault(v.refPos, []).append(v)
        return [byPos[p] for p in sorted(byPos)]

    def getBunchesOfInteractingVariants(self, varsByPos, options):
        bunches = []
        for group in varsByPos:
            if not bunches:
                bunches.append(group)
                continue
            lastMin, lastMax = min(v.minRefPos for v in bunches[-1]), max(v.maxRefPos for v in bunches[-1])
            thisMin, thisMax = min(v.minRefPos for v in group), max(v.maxRefPos for v in group)
            gap = thisMin - lastMax
            if lastMax >= thisMin:
                merge = True                                                       # overlapping variants always share a window
            elif not options.mergeClusteredVariants or gap >= options.maxVarDist:
                merge = False
            elif thisMax - lastMin > (options.maxSize if options.largeWindows == 1 else options.rlen):
                merge = False                                                      # would exceed one read length
            elif len(bunches[-1]) + len(group) <= options.maxVariants:
                merge = True
            else:
                merge = gap < options.minVarDist                                   # too many variants: split only at a wide gap
            if merge:
                bunches[-1].extend(group)
            else:
                bunches.append(group)
        return bunches

    def getWindowVariants(self, chromosome, start, end, sortedVariants, options):
        return self.getBunchesOfInteractingVariants(self.getVariantsByPos(chromosome, start, end, sortedVariants), options)

    @staticmethod
    def _refBlocks(chromosome, first, stop, step):
        for blockStart in range(first, stop, step):
            blockEnd = min(blockStart + step, stop - 1)
            if blockStart != blockEnd:
                yield dict(chromosome=chromosome, startPos=blockStart, endPos=blockEnd, variants=[], nVar=0)

    def WindowsAndVariants(self, chromosome, start, end, maxContigPos, sortedVariants, options):
        groups = self.getWindowVariants(chromosome, start, end, sortedVariants, options)
        for index, vs in enumerate(groups):
            lo, hi = min(v.minRefPos for v in vs), max(v.maxRefPos for v in vs)
            if options.outputRefCalls:                                             # reference-call blocks in the gaps (:172-219)
                if index == 0:
                    firstVarPos = max(lo + 1, start)
                    if firstVarPos - start >= 1:
                        yield from self._refBlocks(chromosome, start, firstVarPos, options.refCallBlockSize)
                else:
                    lastVarPos = max(v.maxRefPos for v in groups[index - 1])
                    if lo + 1 - lastVarPos > 1:
                        yield from self._refBlocks(chromosome, lastVarPos + 1, lo + 1, options.refCallBlockSize)
            yield dict(chromosome=chromosome, startPos=max(lo - options.minVarDist, start),
                       endPos=min(hi + options.minVarDist, maxContigPos), variants=vs, nVar=len(vs))
