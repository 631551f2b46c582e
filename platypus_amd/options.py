"""The `callVariants` command-line surface of the reference (option names, dests and defaults of
src/python/runner.py:519-597), kept so that scripts driving Platypus keep working.

Only the options that reach the hot path are interpreted by this package (rlen/maxReadLength,
calculateFlankScore, HLATyping, the assembly and QC thresholds, nCPU -> number of GPU ranks); the others are
accepted, stored on the options object and ignored here (their consumers are outside the hot-path scope,
SURVEY.md 2.1)."""
import argparse

# (flag, dest, type, default)  -- facts of the reference's CLI, no help texts
CALL_VARIANTS_OPTIONS = [
    ("--output", "output", str, "AllVariants.vcf"), ("--refFile", "refFile", str, None),
    ("--regions", "regions", "list", None), ("--skipRegionsFile", "skipRegionsFile", str, None),
    ("--bamFiles", "bamFiles", "list", None), ("--bufferSize", "bufferSize", int, 100000),
    ("--minReads", "minReads", int, 2), ("--maxReads", "maxReads", float, 5000000),
    ("--verbosity", "verbosity", int, 2), ("--maxReadLength", "rlen", int, 150),
    ("--logFileName", "logFileName", str, "log.txt"), ("--source", "sourceFile", "list", None),
    ("--nCPU", "nCPU", int, 1), ("--parseNCBI", "parseNCBI", int, 0), ("--longHaps", "longHaps", int, 0),
    ("--alignScoreFile", "alignScoreFile", str, ""), ("--HLATyping", "HLATyping", int, 0),
    ("--compressReads", "compressReads", int, 0), ("--qualBinSize", "qualBinSize", int, 1),
    ("--fileCaching", "fileCaching", int, 0),
    ("--maxSize", "maxSize", int, 1500), ("--largeWindows", "largeWindows", int, 0),
    ("--maxVariants", "maxVariants", int, 8), ("--coverageSamplingLevel", "coverageSamplingLevel", int, 30),
    ("--maxHaplotypes", "maxHaplotypes", int, 50), ("--skipDifficultWindows", "skipDifficultWindows", int, 0),
    ("--getVariantsFromBAMs", "getVariantsFromBAMs", int, 1), ("--genSNPs", "genSNPs", int, 1),
    ("--genIndels", "genIndels", int, 1), ("--mergeClusteredVariants", "mergeClusteredVariants", int, 1),
    ("--minFlank", "minFlank", int, 10), ("--trimReadFlank", "trimReadFlank", int, 0),
    ("--filterVarsByCoverage", "filterVarsByCoverage", int, 1), ("--filteredReadsFrac", "filteredReadsFrac", float, 0.7),
    ("--maxVarDist", "maxVarDist", int, 15), ("--minVarDist", "minVarDist", int, 9),
    ("--useEMLikelihoods", "useEMLikelihoods", int, 0),
    ("--countOnlyExactIndelMatches", "countOnlyExactIndelMatches", int, 0),
    ("--calculateFlankScore", "calculateFlankScore", int, 0),
    ("--assemble", "assemble", int, 0), ("--assembleAll", "assembleAll", int, 1),
    ("--assemblyRegionSize", "assemblyRegionSize", int, 1500), ("--assembleBadReads", "assembleBadReads", int, 1),
    ("--assemblerKmerSize", "assemblerKmerSize", int, 15), ("--assembleBrokenPairs", "assembleBrokenPairs", int, 0),
    ("--noCycles", "noCycles", int, 0),
    ("--minMapQual", "minMapQual", int, 20), ("--minBaseQual", "minBaseQual", int, 20),
    ("--minGoodQualBases", "minGoodQualBases", int, 20), ("--filterDuplicates", "filterDuplicates", int, 1),
    ("--filterReadsWithUnmappedMates", "filterReadsWithUnmappedMates", int, 1),
    ("--filterReadsWithDistantMates", "filterReadsWithDistantMates", int, 1),
    ("--filterReadPairsWithSmallInserts", "filterReadPairsWithSmallInserts", int, 1),
    ("--trimOverlapping", "trimOverlapping", int, 1), ("--trimAdapter", "trimAdapter", int, 1),
    ("--trimSoftClipped", "trimSoftClipped", int, 1),
    ("--maxGOF", "maxGOF", int, 30), ("--minPosterior", "minPosterior", int, 5),
    ("--sbThreshold", "sbThreshold", float, 1e-3), ("--scThreshold", "scThreshold", float, 0.95),
    ("--abThreshold", "abThreshold", float, 1e-3), ("--minVarFreq", "minVarFreq", float, 0.05),
    ("--badReadsWindow", "badReadsWindow", int, 11), ("--badReadsThreshold", "badReadsThreshold", int, 15),
    ("--rmsmqThreshold", "rmsmqThreshold", int, 40), ("--qdThreshold", "qdThreshold", int, 10),
    ("--hapScoreThreshold", "hapScoreThreshold", int, 4),
    ("--outputRefCalls", "outputRefCalls", int, 0), ("--refCallBlockSize", "refCallBlockSize", int, 1000),
]


def _list(s):
    return s.split(",")


def build_parser():
    p = argparse.ArgumentParser(prog="Platypus.py callVariants", allow_abbrev=False)
    for flag, dest, typ, default in CALL_VARIANTS_OPTIONS:
        p.add_argument(flag, *(["-o"] if flag == "--output" else []), dest=dest,
                       type=_list if typ == "list" else typ, default=default)
    p.add_argument("--synthetic", dest="synthetic", type=str, default=None,
                   help="(this build) run the hot path on a synthetic BASELINE config instead of BAM input: config1|config2[:N]|config5[:N]")
    return p


def default_options(**overrides):
    """An options object carrying the reference's defaults (what `options` is inside the reference)."""
    ns = build_parser().parse_args([])
    for k, v in overrides.items():
        if not hasattr(ns, k):
            raise AttributeError("unknown Platypus option %r" % k)
        setattr(ns, k, v)
    ns.originalMaxHaplotypes = ns.maxHaplotypes          # set once per process by the reference, variantcaller.pyx:920
    return ns
