"""Python 3 mirror of the reference's operator interface for the hot path: same class and
method names, argument meaning and error behaviour as the Cython classes it stands in for, with all scoring
done by the HIP library (never on the CPU).

    Variant                      src/cython/variant.pyx:100-454      (value type, ordering)
    AlignedRead                  src/cython/htslibWrapper.pxd:187-201 (cAlignedRead fields)
    ReadArray / bamReadBuffer    src/cython/cwindow.pyx:92-236,485-513 (window pointers)
    FastaFile (in-memory)        src/cython/fastafile.pyx:173-207     (getSequence semantics)
    Haplotype                    src/cython/chaplotype.pyx:127-191,306-384,397-449
    DiploidGenotype              src/cython/cgenotype.pyx:131-218
    Population.setup             src/cython/cpopulation.pyx:197-309
    assembleReadsAndDetectVariants  src/cython/assembler.pyx:1429-1476

The per-call methods (`alignReads`, `alignSingleRead`, `calculateDataLikelihood`) exist for drop-in compatibility
and launch tiny batches; throughput comes from `Population.setup` (all haplotypes x all individuals of a window in
one launch) and, beyond that, from batching many windows through `Engine.call_windows` directly.
"""
import bisect
from types import SimpleNamespace

import numpy as np

from .batch import BAM_FQCFAIL, HostBatch
from .options import default_options
from .regionprep import (WindowGenerator, computeVariantReadSupportFrac, filterVariants, filterVariantsByCoverage,  # noqa: F401
                         getHaplotypesInWindow, leftNormaliseIndel)
from .vcfrecords import (VCF, computeHaplotypeScore, computeSCValue, getHaplotypeInfo, outputCallToVCF, outputRefCall, py2_round,  # noqa: F401
                         refAndAlt, trimLeftPadding, vcfFILTER, vcfINFO)

PLATYPUS_VAR, FILE_VAR, ASSEMBLER_VAR = 1, 2, 4               # variant.pyx:43-45
SNP, MNP, INS, DEL, REP = 0, 1, 2, 3, 4                       # variant.pyx:49-53
hash_nucs, hash_size = 7, 16384                               # calign.pyx:25-26

_engine = None


def get_engine():
    """One Engine (one plat_ctx) per process, like the single Population object per worker (variantcaller.pyx:959)."""
    global _engine
    if _engine is None:
        from .engine import Engine
        import os
        import torch
        _engine = Engine(int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count()))
    return _engine


class Variant:
    """variant.pyx:100-146; ordering = Variant.__richcmp__ (variant.pyx:282-363)."""

    def __init__(self, refName, refPos, removed, added, nSupportingReads=0, varSource=PLATYPUS_VAR):
        refPos = max(0, refPos)
        self.refName, self.refPos = refName, refPos
        self.removed, self.added = bytes(removed), bytes(added)
        self.nRemoved, self.nAdded = len(self.removed), len(self.added)
        self.nSupportingReads, self.varSource = nSupportingReads, varSource
        self.minRefPos = refPos
        self.maxRefPos = max(refPos, refPos + self.nRemoved - 1)
        self.bamMinPos = self.bamMaxPos = refPos                                 # variant.pyx:125-126
        self.bamAdded, self.bamRemoved = self.added, self.removed               # :129-130
        if self.nRemoved == self.nAdded:
            self.varType = SNP if self.nAdded == 1 else MNP
        elif self.nRemoved == 0:
            self.varType = INS
        elif self.nAdded == 0:
            self.varType = DEL
        else:
            self.varType = REP

    def addVariant(self, other):                                                 # :261-268
        self.nSupportingReads += other.nSupportingReads
        self.varSource |= other.varSource
        self.bamMinPos = min(self.bamMinPos, other.bamMinPos)
        self.bamMaxPos = max(self.bamMaxPos, other.bamMaxPos)

    def _key(self):
        return (self.refName, self.refPos, self.varType, self.nRemoved)

    def __lt__(self, o):
        return self._key() < o._key()

    def __eq__(self, o):
        return (self.refName, self.refPos, self.added, self.removed) == (o.refName, o.refPos, o.added, o.removed)

    def __hash__(self):
        return hash((self.refName, self.refPos, self.removed, self.added))

    def __repr__(self):
        return "Variant(%s:%d %s->%s)" % (self.refName, self.refPos, self.removed.decode(), self.added.decode())

    def calculatePrior(self, refFile=None):
        """variant.pyx:219-259.  Indels: the tandem-repeat error model of variant.pyx:146-217 (platypus_amd/indelprior.py);
        an explicit `prior` attribute, when set, replaces the model (callers with their own priors)."""
        explicit = getattr(self, "prior", None)
        if explicit is not None:
            return max(float(explicit), 1e-10)
        if self.nAdded == 1 and self.nRemoved == 1:
            prior = 1e-3 / 3
        elif self.nAdded == self.nRemoved:
            nDiffs = len([1 for x, y in zip(self.added, self.removed) if x != y])
            prior = 5e-5 * (0.1 ** (nDiffs - 1)) * (1.0 - 0.1)
        elif self.nAdded > 0 and self.nRemoved == 0:
            prior = self.indelPrior(refFile, self.nAdded)
        elif self.nAdded == 0 and self.nRemoved > 0:
            prior = self.indelPrior(refFile, -self.nRemoved)
        else:
            prior = 5e-6
        return max(prior, 1e-10)

    def indelPrior(self, refFile, indel_length_and_type):                        # :146-217
        from .indelprior import indelPrior
        if refFile is None:
            raise ValueError("the indel prior needs the reference sequence around the variant: pass refFile (or set Variant.prior)")
        return indelPrior(self, refFile, indel_length_and_type)


class AlignedRead:
    """The fields of cAlignedRead the hot path reads (htslibWrapper.pxd:187-201).  qual is raw phred."""

    def __init__(self, seq, qual, pos, mapq=60, bitFlag=3, end=None, cigarOps=None, chromID=0, mateChromID=0, insertSize=0, matePos=-1):
        self.seq, self.qual = bytes(seq), bytes(qual)
        if len(self.seq) != len(self.qual):
            raise ValueError("seq and qual differ in length")
        self.rlen = len(self.seq)
        self.pos = int(pos)
        self.end = int(end) if end is not None else self.pos + self.rlen
        self.mapq, self.bitFlag = int(mapq), int(bitFlag)
        self.cigarOps = [tuple(c) for c in cigarOps] if cigarOps is not None else [(0, self.rlen)]   # (op, len) pairs; default = one match
        self.chromID, self.mateChromID, self.insertSize, self.matePos = chromID, mateChromID, insertSize, matePos

    def isQCFail(self):
        return (self.bitFlag & BAM_FQCFAIL) != 0


class ReadArray:
    """cwindow.pyx:92-236: reads sorted by position with [windowStart, windowEnd) pointers."""

    def __init__(self, reads=(), byMatePos=False):
        # bamReadBuffer.sortReads / sortBrokenMates (cwindow.pyx:748-766): brokenMates are ordered by the position of the mate
        self.array = sorted(reads, key=(lambda r: r.matePos) if byMatePos else (lambda r: r.pos))
        self._pos = [r.pos for r in self.array]
        self.longestRead = max([r.end - r.pos for r in self.array], default=0)   # :167-172
        self.windowStart = self.windowEnd = 0

    @classmethod
    def view(cls, reads):
        """The reads between another array's window pointers, frozen: window = everything, order kept."""
        a = cls.__new__(cls)
        a.array = list(reads)
        a._pos = [r.pos for r in a.array]
        a.longestRead = max([r.end - r.pos for r in a.array], default=0)
        a.windowStart, a.windowEnd = 0, len(a.array)
        return a

    def getSize(self):
        return len(self.array)

    def getLengthOfLongestRead(self):
        return self.longestRead

    def _overlapRange(self, start, end):                                          # shared body of :176-206 and :208-234
        s = bisect.bisect_left(self._pos, max(1, start - self.longestRead))
        e = bisect.bisect_left(self._pos, end)
        while s < len(self.array) and self.array[s].end <= start:
            s += 1
        if s > e:
            raise RuntimeError("This should never happen. Read start pointer > read end pointer!!")
        return s, min(e, len(self.array))

    def countReadsCoveringRegion(self, start, end):                               # :176-206
        if not self.array:
            return 0
        s, e = self._overlapRange(start, end)
        return e - s

    def setWindowPointersBasedOnMatePos(self, start, end):                        # :236-264 (no overlap trimming there)
        if not self.array:
            self.windowStart = self.windowEnd = 0
            return
        mates = [r.matePos for r in self.array]
        s = bisect.bisect_left(mates, max(1, start - self.longestRead))
        e = bisect.bisect_left(mates, end)
        if s > e:
            raise RuntimeError("This should never happen. Read start pointer > read end pointer!!")
        self.windowStart, self.windowEnd = s, min(e, len(self.array))

    def setWindowPointers(self, start, end):
        if not self.array:
            self.windowStart = self.windowEnd = 0
            return
        self.windowStart, self.windowEnd = self._overlapRange(start, end)          # cwindow.pyx:222-234

    def window(self):
        return self.array[self.windowStart:self.windowEnd]


class bamReadBuffer:
    """cwindow.pyx:485-513: per-sample reads / badReads / brokenMates."""

    def __init__(self, reads=(), badReads=(), brokenMates=(), sample="sample"):
        self.reads, self.badReads, self.brokenMates = ReadArray(reads), ReadArray(badReads), ReadArray(brokenMates, byMatePos=True)
        self.sample = sample

    def countReadsCoveringRegion(self, start, end):                               # cwindow.pyx:649-653
        return self.reads.countReadsCoveringRegion(start, end)

    def countAlignmentGaps(self):                                                 # cwindow.pyx:598-622
        return sum(1 for r in list(self.reads.window()) + list(self.badReads.window()) for op, _ in r.cigarOps if 1 <= op <= 4)

    def countImproperPairs(self):                                                 # cwindow.pyx:624-647
        return sum(1 for r in list(self.reads.window()) + list(self.badReads.window()) if not (r.bitFlag & 2))

    def frozenWindow(self):
        """A buffer holding exactly the reads between the current window pointers (batched calling keeps one per window,
        where the reference moves the pointers of the one buffer from window to window)."""
        b = bamReadBuffer.__new__(bamReadBuffer)
        b.sample = self.sample
        b.reads, b.badReads, b.brokenMates = (ReadArray.view(self.reads.window()), ReadArray.view(self.badReads.window()),
                                              ReadArray.view(self.brokenMates.window()))
        return b

    def setWindowPointers(self, start, end):                                      # cwindow.pyx:655-689
        self.reads.setWindowPointers(start, end)
        self.badReads.setWindowPointers(start, end)
        self.brokenMates.setWindowPointersBasedOnMatePos(start, end)

    def windowReads(self):
        """good -> bad -> brokenMates, the order Haplotype.alignReads walks them (chaplotype.pyx:341-373)."""
        return ([(r, 0) for r in self.reads.window()] + [(r, 1) for r in self.badReads.window()] +
                [(r, 2) for r in self.brokenMates.window()])


class FastaFile:
    """In-memory stand-in for fastafile.pyx: upper-cased sequence, half-open getSequence clamped to [0, len-1]."""

    def __init__(self, sequences):
        self._seq = {k: bytes(v).upper() for k, v in sequences.items()}
        self.refs = {k: SimpleNamespace(SeqLength=len(v)) for k, v in self._seq.items()}

    def getSequence(self, seqName, beginPos, endPos):                             # fastafile.pyx:173-207
        n = self.refs[seqName].SeqLength
        beginPos, endPos = max(0, beginPos), min(n - 1, endPos)
        if endPos < beginPos:
            raise IndexError("Cannot have beginPos = %s, endPos = %s" % (beginPos, endPos))
        return self._seq[seqName][beginPos:endPos]

    def getCharacter(self, seqName, pos):                                         # fastafile.pyx:120-132
        s = self._seq[seqName]
        return b"-" if (pos >= len(s) or pos < 0) else s[pos:pos + 1]


def _pack_windows(specs):
    """HostBatch of several calling windows; spec = (haplotype byte strings, startPos, endPos, endBufferSize, per-individual
    bamReadBuffers with their window pointers set).  All windows must have the same number of individuals."""
    haps, reads, whb, wrb, seg_b, seg_g, ws, we, wf = [], [], [0], [0], [0], [], [], [], []
    for hs, startPos, endPos, endBufferSize, buffers in specs:
        haps += hs
        for buf in buffers:
            reads += buf.windowReads()
            seg_b.append(len(reads))
            seg_g.append(buf.reads.windowEnd - buf.reads.windowStart)
        whb.append(len(haps)); wrb.append(len(reads))
        ws.append(startPos); we.append(endPos); wf.append(endBufferSize)
    n_ind = len(specs[0][4])
    assert all(len(sp[4]) == n_ind for sp in specs)
    hl = np.array([len(h) for h in haps], dtype=np.int64)
    rl = np.array([r.rlen for r, _ in reads], dtype=np.int64)
    cat = lambda parts: np.frombuffer(b"".join(parts), dtype=np.uint8)
    i32 = lambda a: np.array(a, dtype=np.int32)
    return HostBatch(
        n_ind=n_ind, win_hap_begin=i32(whb), win_read_begin=i32(wrb), win_start=i32(ws), win_end=i32(we), win_flank=i32(wf),
        hap_seq=cat(haps), hap_off=np.concatenate([[0], np.cumsum(hl)]).astype(np.int64), read_seq=cat([r.seq for r, _ in reads]),
        read_qual=cat([r.qual for r, _ in reads]), read_off=np.concatenate([[0], np.cumsum(rl)]).astype(np.int64),
        read_pos=i32([r.pos for r, _ in reads]), read_end=i32([r.end for r, _ in reads]),
        read_mapq=np.array([r.mapq for r, _ in reads], dtype=np.uint8), read_flags=i32([r.bitFlag for r, _ in reads]),
        read_kind=np.array([k for _, k in reads], dtype=np.uint8), seg_read_begin=i32(seg_b), seg_n_good=i32(seg_g))


def _pack_window(haps, startPos, endPos, endBufferSize, buffers):
    """One-window HostBatch from haplotype byte strings and per-individual bamReadBuffers."""
    return _pack_windows([(list(haps), startPos, endPos, endBufferSize, buffers)])


class Haplotype:
    """chaplotype.pyx:127-191 (construction), :306-384 (alignReads / alignSingleRead), :397-449 (mutated sequence)."""

    def __init__(self, refName, startPos, endPos, variants, refFile, maxReadLength, options=None):
        self.refName, self.refFile, self.variants = refName, refFile, tuple(variants)
        self.options = options if options is not None else default_options()
        self.startPos = max(0, startPos)
        self.endPos = min(endPos, refFile.refs[refName].SeqLength - 1)
        self.maxReadLength = maxReadLength
        self.endBufferSize = min(2 * maxReadLength, 500)                        # :142
        self.lastIndividualIndex = -1
        self.likelihoodCache = None
        if len(self.variants) > 0:                                               # :151-156
            self.minVarPos = min(v.minRefPos for v in self.variants)
            self.maxVarPos = max(v.maxRefPos for v in self.variants)
            if self.minVarPos == self.maxVarPos:
                self.maxVarPos += 1
        else:
            self.minVarPos, self.maxVarPos = self.startPos, self.endPos          # :159-160
        self.referenceSequence = refFile.getSequence(refName, self.startPos - self.endBufferSize, self.endPos + self.endBufferSize)
        if len(self.variants) == 0:
            self.haplotypeSequence = self.referenceSequence
        else:
            left = refFile.getSequence(refName, self.startPos - self.endBufferSize, self.startPos)
            right = refFile.getSequence(refName, self.endPos, self.endPos + self.endBufferSize)
            self.haplotypeSequence = left + self.getMutatedSequence() + right
        self.hapLen = len(self.haplotypeSequence)
        if self.hapLen > hash_size:                                              # :180-183
            raise Exception("Haplotype is too long. Max allowed length is %s" % hash_size)

    def getMutatedSequence(self):                                                # :397-449
        ref, name = self.refFile, self.refName
        cur = self.startPos
        bits = []
        first = self.variants[0]
        if first.refPos != cur:
            bits.append(ref.getSequence(name, cur, first.refPos))
            cur = first.refPos
        for v in self.variants:
            if v.refPos > cur:
                bits.append(ref.getSequence(name, cur, v.refPos))
                cur = v.refPos
            if v.nAdded == v.nRemoved:
                bits.append(v.added)
                cur += v.nRemoved
            else:
                if len(v.added) == 0 or len(v.removed) == 0:
                    if v.refPos == cur:
                        bits.append(ref.getCharacter(name, v.refPos))
                        cur += 1
                cur += v.nRemoved
                bits.append(v.added)
        if cur < self.endPos:
            bits.append(ref.getSequence(name, cur, self.endPos))
        return b"".join(bits)

    def homopolymerLengthForOneVariant(self, variant):                          # :462-498
        left = self.refFile.getSequence(variant.refName, variant.refPos - 20, variant.refPos)
        right = self.refFile.getSequence(variant.refName, variant.refPos + 1, variant.refPos + 21)
        if len(left) == 0 or len(right) == 0:
            return 0
        nl = len(left) - len(left.rstrip(left[-1:]))
        nr = len(right) - len(right.lstrip(right[:1]))
        return max(nl, nr) if left[-1] != right[0] else nl + nr

    def getSequenceContext(self, variant):                                       # :500-506
        return self.refFile.getSequence(variant.refName, variant.refPos - 10, variant.refPos + 11)

    def vcfINFO(self):                                                           # :508-530
        return {v: {"HP": [self.homopolymerLengthForOneVariant(v)], "SC": [self.getSequenceContext(v).decode("ascii")]}
                for v in self.variants}

    def __eq__(self, o):
        return (self.refName, self.startPos, self.endPos, self.haplotypeSequence) == (o.refName, o.startPos, o.endPos, o.haplotypeSequence)

    def __hash__(self):
        return hash((self.refName, self.startPos, self.endPos, self.haplotypeSequence))

    def alignReads(self, individualIndex, readBuffer, useMapQualCap=False):
        """-> numpy array of totalReads+1 natural-log likelihoods terminated by 999 (likelihoodCache, :306-377)."""
        if individualIndex != self.lastIndividualIndex or self.likelihoodCache is None:
            eng = get_engine()
            hb = _pack_window([self.haplotypeSequence], self.startPos, self.endPos, self.endBufferSize, [readBuffer])
            db = eng.upload(hb)
            eng.align(db, want_stats=False, calc_flank_score=int(self.options.calculateFlankScore),
                      use_mapq_cap=int(bool(useMapQualCap)))
            eng.synchronize()
            ll = db.loglik.cpu().numpy()[:hb.n_pairs]
            self.likelihoodCache = np.concatenate([ll, [999.0]])
            self.lastIndividualIndex = individualIndex
        return self.likelihoodCache

    def alignSingleRead(self, theRead, useMapQualCap=False):                     # :379-384 (never skipped: no overlap test)
        buf = bamReadBuffer(brokenMates=[theRead])
        buf.brokenMates.windowStart, buf.brokenMates.windowEnd = 0, 1
        eng = get_engine()
        hb = _pack_window([self.haplotypeSequence], self.startPos, self.endPos, self.endBufferSize, [buf])
        db = eng.upload(hb)
        eng.align(db, want_stats=False, calc_flank_score=int(self.options.calculateFlankScore),
                  use_mapq_cap=int(bool(useMapQualCap)))
        eng.synchronize()
        return float(db.loglik.cpu().numpy()[0])


class DiploidGenotype:
    """cgenotype.pyx:131-189."""

    def __init__(self, hap1, hap2):
        self.hap1, self.hap2 = hap1, hap2
        self.hap1Like = self.hap2Like = 0.0                                      # filled by Population.setup (device sums)

    def __contains__(self, v):                                                   # :98-105
        return v in self.hap1.variants or v in self.hap2.variants

    def calculateDataLikelihood(self, readBuffer, individualIndex, nIndividuals, gof=None, useMapQualCap=False):
        eng = get_engine()
        haps = [self.hap1.haplotypeSequence] if self.hap1 is self.hap2 else [self.hap1.haplotypeSequence, self.hap2.haplotypeSequence]
        hb = _pack_window(haps, self.hap1.startPos, self.hap1.endPos, self.hap1.endBufferSize, [readBuffer])
        db = eng.upload(hb)
        eng.call_windows(db, want_stats=False)
        eng.synchronize()
        logl = db.logl.cpu().numpy()
        g = 0 if self.hap1 is self.hap2 else 1                                   # genotypes of [h1,h2]: (0,0),(0,1),(1,1)
        if gof is not None:
            gof[individualIndex] = float(db.gof.cpu().numpy()[g])
        return float(logl[g])


def generateAllGenotypesFromHaplotypeList(haplotypes):                           # cgenotype.pyx:193-218
    return [DiploidGenotype(haplotypes[i], haplotypes[j]) for i in range(len(haplotypes)) for j in range(i, len(haplotypes))]


class Population:
    """cpopulation.pyx:197-309: the likelihood-array interface consumed by EM / VCF output."""

    def __init__(self, options=None):
        self.options = options if options is not None else default_options()
        self._w = 0
        self.reset()

    def _bind(self, variants, haplotypes, genotypes, nInd, readBuffers):
        if nInd != len(readBuffers):
            raise Exception("Error in cPopulation.setup")                        # :215-219
        self.variants, self.haplotypes, self.genotypes, self.readBuffers = variants, haplotypes, genotypes, readBuffers
        self.nGenotypes, self.nVariants, self.nHaplotypes, self.nIndividuals = len(genotypes), len(variants), len(haplotypes), nInd
        self._hapIndex = {id(h): i for i, h in enumerate(haplotypes)}
        self.haplotypeIndexes = np.array([[self._hapIndex[id(g.hap1)], self._hapIndex[id(g.hap2)]] for g in genotypes], dtype=np.int32)
        H = len(haplotypes)
        expected = [(i, j) for i in range(H) for j in range(i, H)]
        if [tuple(x) for x in self.haplotypeIndexes.tolist()] != expected:
            raise ValueError("genotypes must be generateAllGenotypesFromHaplotypeList(haplotypes)")

    def _readSetup(self, db, w, gl, logl, gof, loglik, like, score):
        """Take window w's slices of the batch results (host copies of db.gl / logl / gof / loglik, hap likes, HapScores)."""
        hb = db.host
        nInd, G, H = self.nIndividuals, self.nGenotypes, self.nHaplotypes
        o, h0 = int(hb.gl_off[w]), int(hb.win_hap_begin[w])
        self.nReads = np.array(hb.seg_n_good[w * nInd:(w + 1) * nInd], dtype=np.int32)               # :286-287
        self.genotypeLikelihoods = gl[o:o + nInd * G].reshape(nInd, G).copy()                        # [ind][genotype]
        self.genotypeLogLikelihoods = logl[o:o + nInd * G].reshape(nInd, G).copy()
        self.goodnessOfFitValues = gof[o:o + nInd * G].reshape(G, nInd).copy()                       # [genotype][ind]
        self.haplotypeLikelihoods = loglik[int(hb.pair_off[w]):int(hb.pair_off[w + 1])].reshape(H, -1).copy()
        self.haplotypeScore = int(score[w])
        for g in self.genotypes:                                                                     # cgenotype.pyx:148-161
            g.hap1Like, g.hap2Like = float(like[h0 + self._hapIndex[id(g.hap1)]]), float(like[h0 + self._hapIndex[id(g.hap2)]])
        self._db, self._w = db, w

    def _readCall(self, freq, em, calls, iters):
        db, w = self._db, self._w
        hb = db.host
        nInd, G = self.nIndividuals, self.nGenotypes
        o, h0 = int(hb.gl_off[w]), int(hb.win_hap_begin[w])
        self.frequencies = freq[h0:h0 + self.nHaplotypes].copy()
        self.EMLikelihoods = em[o:o + nInd * G].reshape(nInd, G).copy()
        self.genotypeCalls = [None if g < 0 else self.genotypes[g] for g in calls[w * nInd:(w + 1) * nInd]]   # :623-676
        self.emIterations = int(iters[w])

    def setup(self, variants, haplotypes, genotypes, nInd, verbosity, readBuffers):
        self._bind(variants, haplotypes, genotypes, nInd, readBuffers)
        h0 = haplotypes[0]
        eng = get_engine()
        hb = _pack_window([h.haplotypeSequence for h in haplotypes], h0.startPos, h0.endPos, h0.endBufferSize, readBuffers)
        db = eng.upload(hb)
        eng.call_windows(db, want_stats=False)
        like, score = eng.haplotype_scores(db)
        self._readSetup(db, 0, db.gl.cpu().numpy(), db.logl.cpu().numpy(), db.gof.cpu().numpy(), db.loglik.cpu().numpy(), like, score)
        return self

    # ---- SURVEY 8(f) rank 1 ----------------------------------------------------------------------------
    def call(self, maxIters=100, computeVCFFields=0):
        """cpopulation.pyx:678-720: EM frequencies, genotype calls, variant posteriors and, for computeVCFFields != 0, the
        INFO / FILTER dictionaries of the window."""
        eng = get_engine()
        db = self._db
        eng.em(db, maxIters, int(self.options.useEMLikelihoods))
        eng.synchronize()
        self._readCall(db.freq.cpu().numpy(), db.em.cpu().numpy(), db.calls.cpu().numpy(), db.em_iters.cpu().numpy())
        self.computeVariantPosteriors()
        self.vcfInfo, self.vcfFilter = {}, {}
        if computeVCFFields != 0 and len(self.variantPosteriors) > 0:                                # :718-720
            self.computeVariantINFO()
            self.computeVariantFILTER()
        return self

    def reset(self):                                                                                  # :166-195
        self.variantPosteriors, self.varsByPos, self.vcfInfo, self.vcfFilter = {}, {}, {}, {}
        self.variants, self.haplotypes, self.genotypes, self.genotypeCalls = [], [], [], []

    def computeVariantINFO(self):                                                                     # :155-158
        self.vcfInfo = vcfINFO(self.frequencies, self.variantPosteriors, self.genotypeCalls, self.genotypes, self.haplotypes,
                               self.readBuffers, self.nHaplotypes, self.options, getattr(self, "refFile", None),
                               hapScore=self.haplotypeScore)

    def computeVariantFILTER(self):                                                                   # :160-164
        self.vcfFilter = vcfFILTER(self.genotypeCalls, self.haplotypes, self.vcfInfo, self.varsByPos, self.options)

    def _masks(self, vs):
        return [np.array([v in h.variants for h in self.haplotypes], dtype=np.uint8) for v in vs]

    def calculatePosterior(self, var, flatPrior=0):                                                   # :459-594
        prior = 0.5 if flatPrior == 1 else var.calculatePrior(getattr(self, "refFile", None))
        if not self.haplotypes:
            # asked about a window it was not set up for (outputRefCall on a window whose calling stopped early): the reference's
            # object still holds the haplotype COUNT of the last window it was set up for and an empty haplotype list (reset(),
            # :166-195), so its loop over the haplotypes raises -- unless it never was set up, then every sum is empty
            if getattr(self, "nHaplotypes", 0) > 0:
                raise IndexError("list index out of range")
            import math
            return float(py2_round(-10.0 * (math.log10(1.0 * (1.0 - prior)) - math.log10(prior + 1.0 * (1.0 - prior)))))
        return float(get_engine().variant_posteriors(self._db, [self._w], self._masks([var]), [prior])[0])

    def _distinctVariants(self):
        vs, done = [], set()
        for h in self.haplotypes:
            for v in h.variants:
                if v not in done:
                    done.add(v); vs.append(v)
        return vs

    def computeVariantPosteriors(self, posteriors=None):                                              # :596-621
        vs = self._distinctVariants()
        self.variantPosteriors, self.varsByPos = {}, {}
        if not vs:
            return
        post = posteriors if posteriors is not None else get_engine().variant_posteriors(
            self._db, [self._w] * len(vs), self._masks(vs), [v.calculatePrior(getattr(self, "refFile", None)) for v in vs])
        for v, p in zip(vs, post):
            if p >= self.options.minPosterior:
                self.variantPosteriors[v] = float(p)
                self.varsByPos.setdefault(v.refPos, []).append(v)

    def computeGenotypeCallAndLikelihoods(self, sampleIndex, variantsThisPos, haplotypeIsRefAtThisPos):
        """vcfutils.pyx:163-334 for one sample and one VCF position.  Returns the reference's 7-tuple."""
        vih = np.array([[v in h.variants for v in variantsThisPos] for h in self.haplotypes], dtype=np.int32)
        ph, lik, out4 = get_engine().genotype_calls(self._db, [dict(window=self._w, var_in_hap=vih, is_ref=haplotypeIsRefAtThisPos)])[0]
        i = sampleIndex
        return (int(ph[i][0]), int(ph[i][1]), lik[i].tolist(), float(out4[i][0]), float(out4[i][1]), float(out4[i][2]), float(out4[i][3]))

# ---- read QC / trimming ----------------------------------------------------------------------------------------------------
def checkAndTrimReads(reads, options=None, enabled=(1, 1, 1, 1)):
    """checkAndTrimRead (cwindow.pyx:332-481) for a stream of AlignedRead objects in buffer order, as
    bamReadBuffer.addReadToBuffer (:560-595) applies it: qualities are trimmed and QCFail flags set IN PLACE; returns
    (ok list, filteredReadCountsByType-style counts [7]).  `enabled`: MATE_UNMAPPED, MATE_DISTANT, SMALL_INSERT, DUPLICATE filters
    (filterReadsWithUnmappedMates, filterReadsWithDistantMates, filterReadPairsWithSmallInserts, filterDuplicates)."""
    options = options if options is not None else default_options()
    st = [dict(qual=r.qual, pos=r.pos, mapq=r.mapq, flag=r.bitFlag, chromID=r.chromID, mateChromID=r.mateChromID,
               insertSize=r.insertSize, matePos=r.matePos, cigar=r.cigarOps) for r in reads]
    st = [dict(d, qual=list(d["qual"])) for d in st]
    ok, flags, quals, why = get_engine().read_qc([st], options.minGoodQualBases, options.minMapQual, options.minBaseQual,
                                                 options.trimOverlapping, options.trimAdapter, options.trimReadFlank,
                                                 options.trimSoftClipped, enabled)[0]
    for r, f, q in zip(reads, flags.tolist(), quals):
        r.bitFlag, r.qual = int(f), bytes(q.tolist())
    counts = [int((why == k).sum()) if (k < 2 or k == 6 or enabled[k - 2]) else -1 for k in range(7)]
    return [bool(x) for x in ok.tolist()], counts


# ---- variant candidates from the reads' CIGARs and mismatches ----------------------------------------
class VariantCandidateGenerator:
    """variant.pyx:459-751.  The per-read scan runs on the device (plat_candidates_batch); merging equal variants and
    counting their supporting reads (addVariantToList, :499-527) is the dictionary step below."""

    def __init__(self, region, referenceFile, minMapQual, minFlank, minBaseQual, maxCoverage, maxReadLength, options,
                 verbosity=2, genSNPs=1, genIndels=1):
        self.rname, self.rStart, self.rEnd = region
        self.refFile = referenceFile
        self.minFlank, self.minBaseQual, self.genSNPs, self.genIndels = minFlank, minBaseQual, genSNPs, genIndels
        self.refSeqStart = max(0, region[1] - 2000)                                                  # :486
        self.refSeqEnd = min(region[2] + 2000, referenceFile.refs[region[0]].SeqLength - 1)        # :487
        self.pyRefSeq = referenceFile.getSequence(self.rname, self.refSeqStart, self.refSeqEnd)     # :488
        self.variantHeap = {}

    def addVariantToList(self, var):                                                                # :499-527
        old = self.variantHeap.get(var)
        if old is None:
            self.variantHeap[var] = var
        else:
            old.addVariant(var)

    def addCandidatesFromReads(self, reads):                                                        # :722-743
        rs = [dict(seq=r.seq, qual=r.qual, pos=r.pos, flag=r.bitFlag, cigar=r.cigarOps) for r in reads]
        region = dict(ref=self.pyRefSeq, ref_seq_start=self.refSeqStart, contig_len=self.refFile.refs[self.rname].SeqLength, reads=rs)
        for pos, removed, added, _ in get_engine().candidates([region], self.minFlank, self.minBaseQual, self.genSNPs, self.genIndels)[0]:
            self.addVariantToList(Variant(self.rname, pos, removed, added, 1, PLATYPUS_VAR))

    def getCandidates(self, minReads=0):                                                            # :747-751
        return sorted(self.variantHeap.values())


# ---- haplotype enumeration and the greedy haplotype filter ------------------------------------------
def isHaplotypeValid(variants):
    """platypusutils.pyx:735-802: variants (sorted by co-ordinate) must not overlap."""
    n = len(variants)
    if n <= 1:
        return True
    for i in range(n - 1):
        a, b = variants[i], variants[i + 1]
        if a.minRefPos > b.minRefPos:
            raise Exception("Variants out of order in haplotype!")
        if a.maxRefPos > b.minRefPos:
            return False
        if a.maxRefPos == b.minRefPos:
            # a SNP/MNP at the labelled base of an indel is fine (the indel really starts one base later), :790-799
            if a.nAdded == a.nRemoved and b.nAdded != b.nRemoved:
                continue
            return False
    return True


def computeBestScoresForHaplotypes(readBuffers, refHaplotype, haplotypes, windowSize, targetCoverage):
    """computeBestScoreForGenotype (variantFilter.pyx:237-283) for the genotypes (refHaplotype, h), h in `haplotypes`,
    in ONE device batch: the sampled reads of every sample are aligned to the reference haplotype and to every candidate
    (alignSingleRead semantics: no overlap rule), the per-sample sums run on the host in the reference's order with the
    C library's log/exp."""
    assert targetCoverage > 0
    import math
    sampled, seg = [], [0]
    for buf in readBuffers:
        reads = buf.reads.window()
        if reads:
            meanCoverage = reads[0].rlen * len(reads) // windowSize                                # :264
            sampleRate = max(1, meanCoverage // targetCoverage)
            sampled += list(reads[::sampleRate])                                                   # :269-274
        seg.append(len(sampled))
    nH = len(haplotypes)
    if not sampled or nH == 0:
        return [-1e20] * nH
    buf = bamReadBuffer()
    buf.brokenMates.array = sampled                        # keep the per-sample order (ReadArray() would re-sort by position)
    buf.brokenMates.windowStart, buf.brokenMates.windowEnd = 0, len(sampled)
    eng = get_engine()
    seqs = [refHaplotype.haplotypeSequence] + [h.haplotypeSequence for h in haplotypes]
    hb = _pack_window(seqs, refHaplotype.startPos, refHaplotype.endPos, refHaplotype.endBufferSize, [buf])
    db = eng.upload(hb)
    eng.align(db, want_stats=False, calc_flank_score=int(refHaplotype.options.calculateFlankScore))
    eng.synchronize()
    ll = db.loglik.cpu().numpy()[:hb.n_pairs].reshape(nH + 1, len(sampled))
    ref = ll[0].tolist()
    out = []
    for k in range(nH):
        row = ll[k + 1].tolist()
        best = -1e20
        for i in range(len(readBuffers)):
            if seg[i] == seg[i + 1]:
                continue                                                                           # :261-262
            score = 0.0
            for r in range(seg[i], seg[i + 1]):
                score += math.log(0.5 * (math.exp(ref[r]) + math.exp(row[r])))                    # :270-272
            best = max(best, score)
        out.append(best)
    return out


def getFilteredHaplotypes(chrom, windowStart, windowEnd, refFile, options, variants, refHaplotype, readBuffers):
    """variantFilter.pyx:377-506.  Few variants: every valid combination.  Many: greedy growth of the best haplotypes, one
    variant at a time (most supported first); the candidates of one step are independent and go to the device together."""
    import math
    from heapq import heappush, heappushpop
    from itertools import combinations
    originalMaxHaplotypes = options.originalMaxHaplotypes - 1      # ref will be added later
    maxHaplotypes = options.maxHaplotypes - 1
    maxReadLength = options.rlen
    nVars = len(variants)
    windowSize = windowEnd - windowStart
    targetCoverage = options.coverageSamplingLevel
    mk = lambda vs: Haplotype(chrom, windowStart, windowEnd, vs, refFile, maxReadLength, options)
    if nVars <= math.log2(maxHaplotypes) or (options.filterVarsByCoverage and options.maxVariants <= math.log2(maxHaplotypes)):
        return [mk(vs) for n in range(1, nVars + 1) for vs in combinations(variants, n) if isHaplotypeValid(vs)]   # :411-438
    varsSortedByCoverage = sorted(variants, key=lambda v: v.nSupportingReads, reverse=True)
    hapsByBestScore = []

    def push(item):
        if len(hapsByBestScore) < originalMaxHaplotypes:
            heappush(hapsByBestScore, item)
        else:
            heappushpop(hapsByBestScore, item)
    for tempVar in varsSortedByCoverage:                                                           # :453-489
        tempOldHaps = sorted(hapsByBestScore)
        varThisHap = (tempVar,)
        cands = [varThisHap]
        for score, varsThisHap2 in tempOldHaps:
            both = tuple(sorted(varThisHap + varsThisHap2))
            if isHaplotypeValid(both):
                cands.append(both)
        scores = computeBestScoresForHaplotypes(readBuffers, refHaplotype, [mk(vs) for vs in cands], windowSize, targetCoverage)
        for sc, vs in zip(scores, cands):
            push((sc, vs))
    allHaps = []
    for index, (score, vs) in enumerate(sorted(hapsByBestScore, reverse=True)):                    # :499-504
        if index < maxHaplotypes:
            allHaps.append(mk(vs))
        else:
            break
    return allHaps


def mergeHaplotypes(haplotypes, refFile=None):
    """variantcaller.pyx:325-383: of haplotypes with the same sequence keep the one whose variants have the larger combined prior."""
    merged, last = [], None
    for hap in sorted(haplotypes, key=lambda h: (h.refName, h.startPos, h.haplotypeSequence)):
        if last is None:
            last = hap
        elif hap == last:
            p1 = p2 = 1.0
            for v in last.variants:
                p1 *= v.calculatePrior(refFile)
            for v in hap.variants:
                p2 *= v.calculatePrior(refFile)
            if p2 > p1:
                last = hap
        else:
            merged.append(last)
            last = hap
    if last is not None:
        merged.append(last)
    return merged

# ---- the INFO arithmetic on top of the device's read statistics ------------------------------------------
def logFactorial(x):                                                                                  # platypusutils.pyx:178-191
    import math
    if x < 15:
        ans = 0.0
        for i in range(1, x + 1):
            ans += math.log(i)
        return ans
    y = float(x)
    return (y * math.log(y) + math.log(2.0 * math.pi * y) / 2 - y + (pow(y, -1)) / 12 - (pow(y, -3)) / 360 + (pow(y, -5)) / 1260
            - (pow(y, -7)) / 1680 + (pow(y, -9)) / 1188)


def logBetaFunction(x, y):                                                                            # :213-218
    return (logFactorial(x - 1) + logFactorial(y - 1)) - logFactorial(x + y - 1)


def threeFTwo(k, n, alpha, beta):                                                                     # :267-295
    a_2, a_3, b_1, b_2 = alpha + k + 1.0, k - n + 1.0, k + 2.0, -beta - n + k + 2.0
    theSum = lastTerm = 1.0
    for i in range(1, abs(k - n + 1) + 1):
        newTerm = lastTerm * (a_2 + i - 1) * (a_3 + i - 1) / ((b_1 + i - 1) * (b_2 + i - 1))
        theSum += newTerm
        lastTerm = newTerm
    return theSum


def betaBinomialCDF(k, n, alpha, beta):                                                               # :306-315
    import math
    if k == n:
        return 1.0
    numerator = logBetaFunction(beta + n - k - 1, alpha + k + 1) + math.log(threeFTwo(k, n, alpha, beta))
    denominator = logBetaFunction(alpha, beta) + logBetaFunction(n - k, k + 2) + math.log(n + 1)
    return max(1e-30, 1.0 - math.exp(numerator - denominator))


def computeAlleleBiasPValue(totalReads, variantReads):                                                # vcfutils.pyx:1156-1173
    if totalReads > 0 and float(variantReads) / float(totalReads) >= 0.5:
        return 1.0
    if totalReads == 0:
        return 1.0
    pValue = betaBinomialCDF(variantReads, totalReads, 20, 20)
    return min(pValue, 1.0 - pValue)


def computeStrandBiasPValue(nFwdReads, nRevReads, nFwdVarReads, nRevVarReads):                        # vcfutils.pyx:1177-1222
    if nFwdReads == 0 or nRevReads == 0:
        return 1.0
    useForward = not (nFwdReads < nRevReads)
    if nFwdReads + nRevReads > 0 and nFwdVarReads + nRevVarReads > 0:
        freq = float(nFwdReads if useForward else nRevReads) / float(nFwdReads + nRevReads)
        if freq < 0.5:
            alpha = 20
            beta = int(float(alpha) / freq - alpha)
        elif freq > 0.5:
            beta = 20
            alpha = int(beta * freq / (1.0 - freq))
        else:
            alpha = beta = 20
        return betaBinomialCDF(nFwdVarReads if useForward else nRevVarReads, nFwdVarReads + nRevVarReads, alpha, beta)
    return 1.0


def _round2(x):
    """Python 2's round(x, 2) (the reference's interpreter): correctly rounded on the exact binary value, ties away from zero."""
    from decimal import Decimal, ROUND_HALF_UP
    return float(Decimal(x).quantize(Decimal("0.01"), rounding=ROUND_HALF_UP))


def infoFieldsFromReadStats(counts, nReadsPerSample, nVarReadsPerSample, minBaseQuals):
    """The INFO fields vcfINFO derives from its per-read loop (vcfutils.pyx:1392-1440), from the counters of
    plat_variant_read_stats_batch (Engine.variant_read_stats)."""
    import math
    TC, TC_bad, TR, TC_ab, TR_ab, NR_sb, NF_sb, TCR, TCF, TCR_sb, TCF_sb, NR, NF, nGood, nBad, sumsq = counts
    info = dict(ABPV=[_round2(computeAlleleBiasPValue(TC_ab, TR_ab))], SbPval=[_round2(computeStrandBiasPValue(TCF_sb, TCR_sb, NF_sb, NR_sb))],
                TR=[TR], NF=[NF], NR=[NR], BRF=[_round2(nBad / float(nGood + nBad))], TC=[TC], TCR=[TCR], TCF=[TCF],
                nReadsPerSample=list(nReadsPerSample), nVarReadsPerSample=list(nVarReadsPerSample))
    rms = np.float32(sumsq)                                                                           # `cdef float RMSMQ`: the quotient is
    info["MQ"] = [_round2(math.sqrt(float(rms / np.float32(TC + TC_bad))))] if (TC + TC_bad > 0 and rms > 0) else [0]   # a C float too
    q = sorted(minBaseQuals)
    info["MMLQ"] = [q[len(q) // 2]] if q else [100]
    return info


def assemblyRegion(assemStart, assemEnd, refStart, refEnd, readBuffers, refSeq, options):
    """One entry of Engine.assemble: the reads loadBAMDataIntoGraph (assembler.pyx:1391-1425) would feed the graph -- the good
    reads between the window pointers of every sample, bad reads and broken mates if the options say so, QCFail reads never."""
    seqs, quals = [], []
    for buf in readBuffers:
        sel = list(buf.reads.window())
        if options.assembleBadReads:
            sel += list(buf.badReads.window())
        if options.assembleBrokenPairs:
            sel += list(buf.brokenMates.window())
        for r in sel:
            if not r.isQCFail():
                seqs.append(r.seq); quals.append(r.qual)
    return dict(ref=bytes(refSeq), ref_start=refStart, assem_start=assemStart, assem_end=assemEnd, seqs=seqs, quals=quals)


def assembleRegions(chroms, regions, options):
    """assembleReadsAndDetectVariants for a list of assemblyRegion()s in one device batch -> per region the sorted variants."""
    out = get_engine().assemble(regions, kmer_size=options.assemblerKmerSize, min_qual=options.minBaseQual,
                                min_weight=options.minReads * options.minBaseQual, no_cycles=options.noCycles)
    return [sorted(Variant(c, p, r, a, 0, ASSEMBLER_VAR) for p, r, a in vs) for c, vs in zip(chroms, out)]


def assembleReadsAndDetectVariants(chrom, assemStart, assemEnd, refStart, refEnd, readBuffers, refSeq, options=None):
    """assembler.pyx:1429-1476; read selection as loadBAMDataIntoGraph (:1391-1425)."""
    options = options if options is not None else default_options()
    return assembleRegions([chrom], [assemblyRegion(assemStart, assemEnd, refStart, refEnd, readBuffers, refSeq, options)], options)[0]
