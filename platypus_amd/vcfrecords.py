"""the INFO / FILTER dictionaries of a calling window and the VCF record text, on top of the device
results (genotype likelihoods, EM frequencies, posteriors, per-position genotype calls, read statistics, HapScore).

Host-side mirror of the reference's record layer -- same names, argument meaning and outputs:

    computeHaplotypeScore, getHaplotypeInfo, vcfINFO, computeSCValue, vcfFILTER,
    refAndAlt, trimLeftPadding, outputCallToVCF                      src/cython/vcfutils.pyx:338-599,796-897,1076-1152,1226-1627
    VCF.write_data / format_formatdata                                src/python/vcf.py:297-329,710-739
    vcfInfoSignature / vcfFilterSignature / vcfFormatSignature        src/cython/vcfutils.pyx:72-123 (ids and missing values)

The O(reads) part of vcfINFO runs on the device (plat_variant_read_stats_batch), the per-position genotype marginalisation
too (plat_genotype_call_batch), HapScore too (plat_haplotype_score_batch); what is left here is dictionary assembly and text.

The reference runs under Python 2, and three of its behaviours reach the text: round() (ties away from zero, on the exact
binary value), str(float) ("%.12g") and the iteration order of a set of filter names (FILTER column).  They are restated
below (py2_round, py2_str, py2_set_order).  Pinned by tests/golden/vcf_cases.json.gz."""
import decimal
import math

import numpy as np

PLATYPUS_VAR, FILE_VAR, ASSEMBLER_VAR = 1, 2, 4                                  # variant.pyx:43-45

# ---- Python 2 semantics ---------------------------------------------------------------------------------------------------


def py2_round(x, ndigits=0):
    x = float(x)
    if math.isnan(x) or math.isinf(x):
        return x
    with decimal.localcontext() as ctx:
        ctx.prec = 800
        return float(decimal.Decimal(x).quantize(decimal.Decimal(1).scaleb(-ndigits), rounding=decimal.ROUND_HALF_UP))


def py2_str(x):
    if isinstance(x, float):
        t = "%.12g" % x
        return t + ".0" if t.lstrip("-").isdigit() else t
    if isinstance(x, bytes):
        return x.decode("ascii")
    return str(x)


def _py2_string_hash(key):
    """stringobject.c (2.7, 64-bit), as the unsigned value the set's table is indexed with."""
    b = key.encode("latin-1")
    if not b:
        return 0
    m = (1 << 64) - 1
    x = (b[0] << 7) & m
    for c in b:
        x = ((1000003 * x) & m) ^ c
    x ^= len(b)
    return m - 1 if x == m else x


def py2_tuple_hash(item_hashes):
    """hash(tuple) of CPython 2.7 (tupleobject.c) from the items' hashes, unsigned 64-bit."""
    m = (1 << 64) - 1
    x, mult, n = 0x345678, 1000003, len(item_hashes)
    for i, h in enumerate(item_hashes):
        left = n - 1 - i
        x = ((x ^ (h & m)) * mult) & m
        mult = (mult + 82520 + left + left) & m
    x = (x + 97531) & m
    return m - 1 if x == m else x


def py2_variant_hash(ref_name, ref_pos, removed, added):
    """hash(Variant) of the reference = hash((refName, refPos, removed, added)), variant.pyx:270-280, as the dictionary sees it: the
    value is kept in `public int hashValue` (variant.pxd:31) -- Cython's hash() is PyObject_Hash (Py_hash_t) and the assignment narrows it
    to a C int without a check -- so the table is probed with the sign-extended low 32 bits (a narrowed -1 leaves tp_hash as -2)."""
    as_str = lambda b: b.decode("latin-1") if isinstance(b, bytes) else b
    h = py2_tuple_hash([_py2_string_hash(as_str(ref_name)), (-2 if ref_pos == -1 else ref_pos) & ((1 << 64) - 1),
                        _py2_string_hash(as_str(removed)), _py2_string_hash(as_str(added))]) & 0xFFFFFFFF
    if h >= 1 << 31:
        h -= 1 << 32
    if h == -1:
        h = -2
    return h & ((1 << 64) - 1)


def py2_dict_slot_order(hashes):
    """Iteration order of a Python-2 dict into which DISTINCT keys with these (unsigned) hashes were inserted in this order and never
    deleted (dictobject.c; see py2_dict_order): indices into `hashes`."""
    m = (1 << 64) - 1

    def place(table, key):
        h = hashes[key]
        mask = len(table) - 1
        i, perturb = h & mask, h
        while table[i & mask] is not None:
            i = (5 * i + perturb + 1) & m
            perturb >>= 5
        table[i & mask] = key
    table, used = [None] * 8, 0
    for key in range(len(hashes)):
        place(table, key)
        used += 1
        if used * 3 >= len(table) * 2:
            size = 8
            while size <= used * (2 if used > 50000 else 4):
                size <<= 1
            grown = [None] * size
            for k in table:
                if k is not None:
                    place(grown, k)
            table = grown
    return [k for k in table if k is not None]


def py2_set_order(names):
    """list(set(names)) under CPython 2.7 (setobject.c): open addressing (i = 5i + perturb + 1, perturb >>= 5) in a table of
    8 slots that is rebuilt four times larger when two thirds full; iteration = slot order."""
    m = (1 << 64) - 1

    def slot(table, key, h):
        mask = len(table) - 1
        i, perturb = h & mask, h
        while table[i & mask] is not None and table[i & mask] != key:
            i = (5 * i + perturb + 1) & m
            perturb >>= 5
        return i & mask
    table, used = [None] * 8, 0
    for key in names:
        j = slot(table, key, _py2_string_hash(key))
        if table[j] is not None:
            continue
        table[j] = key
        used += 1
        if used * 3 >= len(table) * 2:
            size = 8
            while size <= used * (2 if used > 50000 else 4):
                size <<= 1
            grown = [None] * size
            for k in table:
                if k is not None:
                    grown[slot(grown, k, _py2_string_hash(k))] = k
            table = grown
    return [k for k in table if k is not None]


# ---- INFO -----------------------------------------------------------------------------------------------------------------

def computeHaplotypeScore(genotypes):
    """vcfutils.pyx:1076-1114 on the host, from DiploidGenotype.hap1Like / hap2Like (device sums, see Population.setup).  The
    batched path takes the same number from plat_haplotype_score_batch."""
    scores = {}
    for g in genotypes:
        scores[g.hap1] = -g.hap1Like
        scores[g.hap2] = -g.hap2Like
    v = sorted(scores.values())
    sizes, dist = [1], 0
    for a, b in zip(v, v[1:]):
        if b - a > 20:
            if len(sizes) == 1:
                dist = b - a
            if len(sizes) == 2:
                break
            sizes.append(1)
        else:
            sizes[-1] += 1
    return sizes[0] + (sizes[1] if 0 < dist < 50 else 0)


def getHaplotypeInfo(haplotypes, variantPosteriors, haplotypeFrequencies, nHaplotypes):              # :1118-1152
    info = {}
    for h in range(nHaplotypes):
        for var, value in haplotypes[h].vcfINFO().items():
            if var not in variantPosteriors:
                continue
            if var not in info:
                info[var] = dict(HP=value["HP"], PP=["%.0f" % variantPosteriors[var]], FR=[float(haplotypeFrequencies[h])], SC=value["SC"])
            else:
                info[var]["FR"][0] += float(haplotypeFrequencies[h])
    return info


def _read_dict(r):
    return dict(seq=r.seq, qual=r.qual, pos=r.pos, end=r.end, mapq=r.mapq, flag=r.bitFlag, cigar=r.cigarOps)


def infoStatsWindow(variants, readBuffers, genotypeCalls):
    """One window of Engine.variant_read_stats: the variants of the INFO dictionary (in its order), the good and bad reads
    between the window pointers of every sample, and `variant in genotypeCall` per (variant, sample)."""
    return dict(variants=[dict(pos=v.refPos, removed=v.removed, added=v.added, bam_min=v.bamMinPos, bam_max=v.bamMaxPos) for v in variants],
                samples=[dict(good=[_read_dict(r) for r in b.reads.window()], bad=[_read_dict(r) for r in b.badReads.window()])
                         for b in readBuffers],
                var_in_genotype=[[int(g is not None and v in g) for g in genotypeCalls] for v in variants])


def vcfINFO(haplotypeFrequencies, variantPosteriors, genotypeCalls, genotypes, haplotypes, readBuffers, nHaplotypes, options, refFile,
            hapScore=None, readStats=None):
    """vcfutils.pyx:1226-1460.  The loop over every read of every sample (:1300-1390) is plat_variant_read_stats_batch; pass
    `readStats` (a list aligned with the INFO dictionary's variants: (counts[16], nReadsPerSample, nVarReadsPerSample,
    minBaseQuals)) to supply the counters from elsewhere."""
    from . import hostapi
    if hapScore is None:
        hapScore = computeHaplotypeScore(genotypes)
    info = getHaplotypeInfo(haplotypes, variantPosteriors, haplotypeFrequencies, nHaplotypes)
    vs = list(info.keys())
    if readStats is None:
        readStats = hostapi.get_engine().variant_read_stats([infoStatsWindow(vs, readBuffers, genotypeCalls)],
                                                            bad_reads_window=options.badReadsWindow,
                                                            exact=options.countOnlyExactIndelMatches)[0]
    for v, (counts, nReads, nVarReads, minQuals) in zip(vs, readStats):
        d = info[v]
        d.update(hostapi.infoFieldsFromReadStats(counts, nReads, nVarReads, minQuals))
        TR = d["TR"][0]
        if TR > 0:                                                                                   # :1400-1409
            qual = float(d["PP"][0])
            d["QD"] = [options.qdThreshold + 10] if qual > 2500 else [(qual + (-10 * math.log10(v.calculatePrior(refFile)))) / TR]
        else:
            d["QD"] = [0]
        d["FR"][0] = "%1.4f" % d["FR"][0]
        d["HapScore"] = [hapScore]
        d["Source"] = [name for bit, name in ((PLATYPUS_VAR, "Platypus"), (ASSEMBLER_VAR, "Assembler"), (FILE_VAR, "File")) if v.varSource & bit]
    return info


# ---- FILTER ---------------------------------------------------------------------------------------------------------------

def computeSCValue(sequence):                                                                         # :1480-1498
    counts = sorted((sequence.count(c) for c in set(sequence)), reverse=True)
    return float(sum(counts[:2])) / float(len(sequence))


def vcfFILTER(genotypeCalls, haplotypes, vcfInfo, varsByPos, options):                                # :1502-1627
    out = {}
    for varsAtPos in varsByPos.values():
        n = len(varsAtPos)
        failsSC = computeSCValue(vcfInfo[varsAtPos[0]]["SC"][0]) > options.scThreshold
        fails = dict(QD=0, HapScore=0, MQ=0, strandBias=0, alleleBias=0, MMLQ=0)
        bestQual, BRF = 0, 0.0
        for v in varsAtPos:
            d = vcfInfo[v]
            out[v] = ["SC"] if failsSC else []
            BRF = float(d["BRF"][0])
            bestQual = max(bestQual, int(d.get("PP", [0])[0]))
            fails["MMLQ"] += int(d.get("MMLQ", [100])[0]) < options.badReadsThreshold
            fails["QD"] += float(d["QD"][0]) < options.qdThreshold
            fails["HapScore"] += int(d["HapScore"][0]) > options.hapScoreThreshold
            fails["alleleBias"] += int(d["TC"][0]) > 0 and float(d["ABPV"][0]) < options.abThreshold
            fails["strandBias"] += float(d["SbPval"][0]) < options.sbThreshold
            fails["MQ"] += float(d["MQ"][0]) < options.rmsmqThreshold
        for v in varsAtPos:                                                                          # BRF: of the last variant, as there
            for name in ("QD", "HapScore", "MQ", "strandBias", "alleleBias"):
                if fails[name] == n:
                    out[v].append(name)
            if fails["MMLQ"] == n or BRF >= options.filteredReadsFrac:
                out[v].append("badReads")
            if bestQual < 20:
                out[v].append("Q20")
    return out


# ---- REF / ALT ------------------------------------------------------------------------------------------------------------

def refAndAlt(chrom, POS, variants, refFile):                                                         # :843-897
    """-> (REF, [ALT...]) as native strings."""
    onlySnps = all(v.nRemoved == 1 and v.nAdded == 1 for v in variants)
    if onlySnps:
        return refFile.getCharacter(chrom, POS).decode("ascii"), [v.added.decode("ascii") for v in variants]
    indel = any(v.nRemoved != v.nAdded for v in variants)
    span = max(v.nRemoved for v in variants)
    REF = refFile.getSequence(chrom, POS, POS + span + (1 if indel else 0)).decode("ascii")
    ALT = []
    for v in variants:
        seq = list(REF)
        if v.nRemoved == v.nAdded:
            seq[0:v.nAdded] = v.added.decode("ascii")
        else:
            seq[1:1 + v.nRemoved] = v.added.decode("ascii")
        ALT.append("".join(seq))
    return REF, ALT


def trimLeftPadding(vcfDataLine):                                                                     # :796-839
    ref, alt = vcfDataLine["ref"], vcfDataLine["alt"]
    if alt:
        shortest = min([len(ref)] + [len(a) for a in alt])
        lengthsDiffer = any(len(a) != len(ref) for a in alt)
        for _ in range(1, shortest):
            first = set(a[0].upper() for a in alt)
            second = set(a[1].upper() for a in alt if len(a) > 1)
            if len(first) > 1 or ref[0].upper() not in first:
                break
            if lengthsDiffer and (len(second) > 1 or ref[1] not in second):
                break
            ref, alt = ref[1:], [a[1:] for a in alt]
            vcfDataLine["pos"] += 1
    vcfDataLine["ref"], vcfDataLine["alt"] = ref, alt


# ---- the writer -----------------------------------------------------------------------------------------------------------
# id -> missing value of the header definitions (vcfutils.pyx:72-123); a value equal to it is written as "."
INFO_MISSING = {k: -1 for k in ("FR PP TC WS WE TCR TCF TR NF NR MGOF SC HP BRF MMLQ QD Source START END Size HapScore MQ FS SbPval "
                                "ReadPosRankSum").split()}
FORMAT_MISSING = {k: "." for k in ("GT", "GL", "GQ", "GOF", "NR", "NV")}
FILTER_IDS = ("alleleBias", "strandBias", "badReads", "MQ", "Q20", "QualDepth", "HapScore", "GOF", "hp10", "REFCALL", "QD", "SC")


# header definitions (vcfutils.pyx:72-123): (id, Number, Type, Description).  The FILTER dictionary of the reference files the
# "HapScore" definition under the key "QualDepth" and vice versa; the header shows the ids, as there.
INFO_DEFS = [
    ("FR", ".", "Float", "Estimated population frequency of variant"),
    ("PP", ".", "Float", "Posterior probability (phred scaled) that this variant segregates"),
    ("TC", "1", "Integer", "Total coverage at this locus"),
    ("WS", "1", "Integer", "Starting position of calling window"),
    ("WE", "1", "Integer", "End position of calling window"),
    ("TCR", "1", "Integer", "Total reverse strand coverage at this locus"),
    ("TCF", "1", "Integer", "Total forward strand coverage at this locus"),
    ("TR", ".", "Integer", "Total number of reads containing this variant"),
    ("NF", ".", "Integer", "Total number of forward reads containing this variant"),
    ("NR", ".", "Integer", "Total number of reverse reads containing this variant"),
    ("MGOF", ".", "Integer", "Worst goodness-of-fit value reported across all samples"),
    ("SC", "1", "String", "Genomic sequence 10 bases either side of variant position"),
    ("HP", "1", "Integer", "Homopolymer run length around variant locus"),
    ("BRF", "1", "Float", "Fraction of reads around this variant that failed filters"),
    ("MMLQ", "1", "Float", "Median minimum base quality for bases around variant"),
    ("QD", "1", "Float", "Variant-quality/read-depth for this variant"),
    ("Source", ".", "String", "Was this variant suggested by Playtypus, Assembler, or from a VCF?"),
    ("START", ".", "Integer", "Start position of reference call block"),
    ("END", ".", "Integer", "End position of reference call block"),
    ("Size", ".", "Integer", "Size of reference call block"),
    ("HapScore", ".", "Integer", "Haplotype score measuring the number of haplotypes the variant is segregating into in a window"),
    ("MQ", ".", "Float", "Root mean square of mapping qualities of reads at the variant position"),
    ("FS", ".", "Float", "Fisher's exact test for strand bias (Phred scale)"),
    ("SbPval", ".", "Float", "Binomial P-value for strand bias test"),
    ("ReadPosRankSum", ".", "Float", "Mann-Whitney Rank sum test for difference between in positions of variants in reads from ref and alt"),
]
FILTER_DEFS = [
    ("alleleBias", "Variant frequency is lower than expected for het"),
    ("strandBias", "Variant fails strand-bias filter"),
    ("badReads", "Variant supported only by reads with low quality bases close to variant position, and not present on both strands."),
    ("MQ", "Root-mean-square mapping quality across calling region is low."),
    ("Q20", "Variant quality is below 20."),
    ("HapScore", "Too many haplotypes are supported by the data in this region."),
    ("QualDepth", "Variant quality/Read depth ratio is low."),
    ("GOF", "Variant fails goodness-of-fit test."),
    ("hp10", "Flanking sequence contains homopolymer of length 10 or greater"),
    ("REFCALL", "This line represents a homozygous reference call"),
    ("QD", "Variants fail quality/depth filter."),
    ("SC", "Variants fail sequence-context filter. Surrounding sequence is low-complexity"),
]
FORMAT_DEFS = [
    ("GT", "1", "String", "Unphased genotypes"),
    ("GL", ".", "Float", "Genotype log10-likelihoods for AA,AB and BB genotypes, where A = ref and B = variant. Only applicable for bi-allelic sites"),
    ("GQ", ".", "Integer", "Genotype quality as phred score"),
    ("GOF", ".", "Float", "Goodness of fit value"),
    ("NR", ".", "Integer", "Number of reads covering variant location in this sample"),
    ("NV", ".", "Integer", "Number of reads containing variant in this sample"),
]


class VCF:
    """The writing half of vcf.py's VCF class (write_data :710-739, format_formatdata :297-329, writeheader :371-409,844-848)."""

    def __init__(self, samples=()):
        self._samples = list(samples)
        self._version = 40
        self._header = []

    def setheader(self, header):
        self._header = list(header)

    def writeheader(self, stream):
        """##fileformat, the caller's key=value lines, the INFO / FILTER / FORMAT definitions and the #CHROM line.  The
        reference writes the definitions in the iteration order of three Python-2 dictionaries; here they come in the order of
        its source, which is the one difference from its header apart from the date and the option string."""
        stream.write("##fileformat=VCFv%s.%s\n" % (self._version // 10, self._version % 10))
        for key, value in self._header:
            stream.write("##%s=%s\n" % (key, value))
        for i, n, t, d in INFO_DEFS:
            stream.write('##INFO=<ID=%s,Number=%s,Type=%s,Description="%s">\n' % (i, n, t, d))
        for i, d in FILTER_DEFS:
            stream.write('##FILTER=<ID=%s,Description="%s">\n' % (i, d))
        for i, n, t, d in FORMAT_DEFS:
            stream.write('##FORMAT=<ID=%s,Number=%s,Type=%s,Description="%s">\n' % (i, n, t, d))
        stream.write("#" + "\t".join(["CHROM", "POS", "ID", "REF", "ALT", "QUAL", "FILTER", "INFO", "FORMAT"] + [py2_str(x) for x in self._samples]) + "\n")

    def setsamples(self, samples):
        self._samples = list(samples)

    def getfilter(self):
        return FILTER_IDS

    @staticmethod
    def format_formatdata(data, missing, key=True, value=True, separator=":"):
        if isinstance(data, list):                                                                   # the FORMAT column: keys only
            data = {k: [] for k in data}
        items = sorted((k, v) for k, v in data.items() if k != "GT")
        if "GT" in data:
            items.insert(0, ("GT", ["".join(py2_str(x) for x in gt) for gt in data["GT"]]))
        out = []
        for k, vals in items:
            # (an undefined key gets a definition without a usable missing value, vcf.py:280-294)
            vals = ["." if (k in missing and v == missing[k]) else v for v in vals]
            text = ",".join(py2_str(v) for v in vals) if vals else None
            if key and value:
                out.append(k if text is None else k + "=" + text)
            elif key:
                out.append(k)
            elif value:
                out.append("." if text is None else text)
        while len(out) > 1 and not out[-1].replace(",", "").replace(".", ""):                        # trailing missing data
            out.pop()
        return separator.join(out)

    def write_data(self, stream, data):
        for k in ["chrom", "pos", "id", "ref", "alt", "qual", "filter", "info", "format"] + self._samples:
            if k not in data:
                raise ValueError("Required key %s not found in data" % str(k))
        alt = "." if data["alt"] == [] else ",".join(data["alt"])
        if data["filter"] is None:
            flt = "."
        elif data["filter"] == []:
            flt = "0" if self._version == 33 else "PASS"
        else:
            flt = ";".join(data["filter"])
        if data["qual"] == -1:
            qual = "."
        else:
            qual = py2_str(data["qual"])
            if qual.endswith(".0"):
                qual = qual[:-2]
        cols = [py2_str(data["chrom"]), str(data["pos"] + 1), data["id"], data["ref"], alt, qual, flt,
                self.format_formatdata(data["info"], INFO_MISSING, separator=";"),
                self.format_formatdata(data["format"], FORMAT_MISSING, value=False)]
        cols += [self.format_formatdata(data[s_], FORMAT_MISSING, key=False) for s_ in self._samples]
        stream.write("\t".join(cols) + "\n")


# ---- records --------------------------------------------------------------------------------------------------------------

def _phred(p):
    return int(min(99, py2_round(-10.0 * math.log10(max(1e-10, 1.0 - p)))))


def genotypeCallSites(varsByPos, haplotypes, allVariants, window=0):
    """The inputs of computeGenotypeCallAndLikelihoods for every position of a window, as outputCallToVCF builds them
    (vcfutils.pyx:400-426): varThisPosInHap[h][k] and haplotypeIsRefAtThisPos[h]."""
    sites = []
    for POS in sorted(varsByPos.keys()):
        variants = varsByPos[POS]
        vih = np.array([[int(v in h.variants) for v in variants] for h in haplotypes], dtype=np.int32).reshape(len(haplotypes), len(variants))
        isRef = np.array([int(not any(v.minRefPos <= POS <= v.maxRefPos for v in h.variants if v in variants or v in allVariants))
                          for h in haplotypes], dtype=np.int32)
        sites.append(dict(window=window, var_in_hap=vih, is_ref=isRef))
    return sites


def genotypeCallTuples(results, nIndividuals):
    """Engine.genotype_calls output -> the reference's 7-tuples, [site][sample]."""
    return [[(int(ph[i][0]), int(ph[i][1]), lik[i].tolist(), float(o4[i][0]), float(o4[i][1]), float(o4[i][2]), float(o4[i][3]))
             for i in range(nIndividuals)] for ph, lik, o4 in results]


def _c_log10(x):
    """log10 of math.h (the reference's is the C function: -inf at zero instead of an exception)."""
    return -math.inf if x == 0 else math.log10(x)


def outputRefCall(chrom, pop, vcfFile, refFile, outputFile, windowIndex, window, options, readBuffers):
    """variantcaller.pyx:764-867: one REFCALL line for a block without called variants.  QUAL: 0 without coverage somewhere in the
    block; else the phred-scaled beta-binomial p-value of seeing no variant read at the block's smallest coverage, capped -- when the
    block holds candidates -- by the posterior of the best candidate under a flat prior (pop.calculatePosterior(v, 1)).
    Pinned by tests/golden/refcall_cases.json.gz (the reference's own text)."""
    from .hostapi import betaBinomialCDF
    windowStart, windowEnd = window["startPos"], window["endPos"]
    variants = window["variants"]
    minCov = -1
    for buf in readBuffers:
        for p in range(windowStart, windowEnd):
            c = buf.countReadsCoveringRegion(p, p + 1)
            minCov = c if minCov == -1 else min(minCov, c)
    phredPValue = int(-10 * _c_log10(betaBinomialCDF(0, minCov, 20, 20)))
    if minCov == 0:
        qual = 0
    elif len(variants) == 0:
        qual = phredPValue
    else:
        maxPost = max(pop.calculatePosterior(v, 1) for v in variants)
        maxProbVar = 1.0 - 10 ** (-0.1 * maxPost)
        probRef = 1.0 - maxProbVar
        qual = min(int(py2_round(-10.0 * _c_log10(1.0 - probRef))), phredPValue)       # (an infinite value raises here, as there)
    ref = py2_str(refFile.getSequence(chrom, windowStart, windowStart + 1))
    alt = ["T"] if ref == "N" else ["N"]                                                 # REF and ALT must differ
    lineinfo = {k: ["."] for k in ("FR", "MMLQ", "HP", "TCR", "WE", "WS", "Source", "FS", "START", "PP", "TR", "NF", "TCF", "NR", "TC", "MGOF",
                                   "SbPval", "ReadPosRankSum", "MQ", "QD", "SC", "BRF", "HapScore")}
    lineinfo["END"], lineinfo["Size"] = [windowEnd], [windowEnd - windowStart]
    line = dict(chrom=chrom, pos=windowStart, ref=ref, alt=alt, id=".", info=lineinfo, filter=["REFCALL"], qual=qual, format=["GT:GL:GOF:GQ:NR:NV"])
    for buf in readBuffers:
        n = buf.reads.windowEnd - buf.reads.windowStart
        line[buf.sample] = dict(GT=[[".", "/", "."]], GL=[-1, -1, -1], GQ=[-1], GOF=[-1], NR=[n], NV=[0])
    vcfFile.write_data(outputFile, line)


def py2_dict_order(keys):
    """Iteration order of a Python-2 dict holding these integer keys, inserted in this order (dictobject.c: the same open addressing
    and growth as the set of py2_set_order; hash(int) is the int, -1 -> -2)."""
    m = (1 << 64) - 1

    def slot(table, key):
        h = (-2 if key == -1 else key) & m
        mask = len(table) - 1
        i, perturb = h & mask, h
        while table[i & mask] is not None and table[i & mask] != key:
            i = (5 * i + perturb + 1) & m
            perturb >>= 5
        return i & mask
    table, used = [None] * 8, 0
    for key in keys:
        j = slot(table, key)
        if table[j] is not None:
            continue
        table[j] = key
        used += 1
        if used * 3 >= len(table) * 2:
            size = 8
            while size <= used * (2 if used > 50000 else 4):
                size <<= 1
            grown = [None] * size
            for k in table:
                if k is not None:
                    grown[slot(grown, k)] = k
            table = grown
    return [k for k in table if k is not None]


def outputCallToVCF(varsByPos, vcfInfo, vcfFilter, haplotypes, genotypes, haplotypeFrequencies, genotypeLikelihoods, gofValues,
                    haplotypeIndexes, readBuffers, nIndividuals, vcfFile, refFile, outputFile, options, allVariants, windowStart, windowEnd,
                    population=None, genotypeCalls=None):
    """vcfutils.pyx:338-599: one VCF line per position of varsByPos.  The per-sample marginalisation
    (computeGenotypeCallAndLikelihoods) runs on the device for all positions and samples of the window in one
    plat_genotype_call_batch: pass `population` (hostapi.Population after call()); `genotypeCalls` = precomputed 7-tuples
    [position index][sample] overrides it.  The array arguments the reference passes (frequencies, likelihoods, ...) are
    accepted for signature parity and not read: the device holds them."""
    positions = sorted(varsByPos.keys())
    if not positions:
        return
    if genotypeCalls is None:
        from . import hostapi
        sites = genotypeCallSites(varsByPos, haplotypes, allVariants, getattr(population, "_w", 0))
        genotypeCalls = genotypeCallTuples(hostapi.get_engine().genotype_calls(population._db, sites), nIndividuals)
    known = vcfFile.getfilter()
    for pi, POS in enumerate(positions):
        variants = varsByPos[POS]
        nVariants = len(variants)
        chrom = variants[0].refName
        ref, alt = refAndAlt(chrom, POS, variants, refFile)
        lineinfo = vcfInfo[variants[0]]
        linefilter = []
        merged = dict(FR=[], PP=[], NF=[], NR=[], TR=[])
        for var in variants:
            linefilter.extend(f for f in vcfFilter[var] if f in known)
            for k in merged:
                merged[k].extend(vcfInfo[var][k])
        lineinfo["WS"], lineinfo["WE"] = [windowStart], [windowEnd]
        lineinfo.update(merged)
        line = dict(chrom=chrom, pos=POS, ref=ref, alt=alt, id=".", info=lineinfo, filter=py2_set_order(linefilter),
                    qual=max(int(pp) for pp in lineinfo["PP"]), format=["GT:GL:GOF:GQ:NR:NV"])
        maxGof, nNonRefCalls = 0.0, 0
        for i in range(nIndividuals):
            buf = readBuffers[i]
            if buf.reads.windowEnd - buf.reads.windowStart == 0:                                     # :498-500
                line[buf.sample] = dict(GT=[[".", "/", "."]], GL=[0, 0, 0], GQ=[0], GOF=[0], NR=[0], NV=[0])
                continue
            index1, index2, likelihoods, gtPost, nonRefPost, refPost, gofValue = genotypeCalls[pi][i]
            if not (index1 == 0 and index2 == 0):
                nNonRefCalls += 1
            GT = [str(index1), "/", str(index2)]
            if nVariants == 1:                                                                       # :524-542
                if _phred(nonRefPost) < options.minPosterior:
                    GT = [".", "/", "."] if _phred(refPost) < options.minPosterior else ["0", "/", "0"]
                top = max(likelihoods)
                GL = [py2_round(math.log10(max(x / top, 1e-300)), 2) for x in likelihoods]
            else:
                GL = [-1, -1, -1]
            NR = [vcfInfo[v]["nReadsPerSample"][i] for v in variants]
            NV = [vcfInfo[v]["nVarReadsPerSample"][i] for v in variants]
            if nVariants == 1 and NR[0] < options.minReads:                                          # :550-553
                GT = [".", "/", "."]
            line[buf.sample] = dict(GT=[GT], GL=GL, GQ=[_phred(gtPost)], GOF=[int(gofValue)], NR=NR, NV=NV)
            maxGof = max(maxGof, gofValue)
        for k in ("nReadsPerSample", "nVarReadsPerSample", "ABPV"):                                  # temporary values
            lineinfo.pop(k)
        lineinfo["MGOF"] = [int(py2_round(maxGof, 2))]
        if nNonRefCalls > 0 or options.minPosterior == 0 or options.outputRefCalls == 1:
            trimLeftPadding(line)
            if all(c in "ACTG" for c in line["ref"]):                                                # :583-592
                vcfFile.write_data(outputFile, line)
