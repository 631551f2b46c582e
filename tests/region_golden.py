"""Shared by the CPU and GPU tests of the region loop against tests/golden/region_cases.json.gz: the REFERENCE's own
callVariantsInRegion text (variantcaller.pyx:535-615 over generateVariantsInRegion :412-531, doWeNeedToAssembleThisRegion :276-321,
callVariantsInWindow :74-141, mergeHaplotypes :325-390, WindowGenerator, the whole Population / Haplotype / bamReadBuffer classes,
the assembler and the record writer), run in the build container by tests/golden/gen_golden.py::gen_region.  A case is one process: a
region list called in order with one options object; inputs are the read buffers as the reference's loader left them."""
import gzip
import io
import json
import os

from platypus_amd import caller, fastcaller as F, hostapi as H
from platypus_amd.options import default_options
from platypus_amd.vcfrecords import VCF

HERE = os.path.dirname(os.path.abspath(__file__))


def load_cases():
    with gzip.open(os.path.join(HERE, "golden", "region_cases.json.gz"), "rt") as f:
        return json.load(f)


def _reads(lst):
    return [H.AlignedRead(x["seq"].encode(), bytes(ord(c) - 33 for c in x["qual"]), x["pos"], x["mapq"], x["flag"], end=x["end"],
                          cigarOps=[tuple(c) for c in x["cigar"]], matePos=x["matePos"]) for x in lst]


def case_work(case):
    """(fasta, [(chrom, start, end, buffers)], sample names) of the regions the reference's loader loaded."""
    fasta = H.FastaFile({"20": case["ref"].encode()})
    work = []
    for reg in case["regions"]:
        if not reg["loaded"]:                       # loadBAMData gave up on the region (maxReads, platypusutils.pyx:538-541): nothing is called
            continue
        bufs = [H.bamReadBuffer(_reads(s["reads"]), _reads(s["badReads"]), _reads(s["brokenMates"]), sample=s["sample"]) for s in reg["samples"]]
        work.append((reg["chrom"], reg["start"], reg["end"], bufs))
    return fasta, work, case["sample_names"]


def python_loop_text(case):
    fasta, work, names = case_work(case)
    opts = default_options(**case["options"])
    out = io.StringIO()
    caller.callVariantsInRegions(work, fasta, opts, VCF(names), out)
    return out.getvalue().split("\n")[:-1], opts.rlen


def window_by_window_text(case):
    fasta, work, names = case_work(case)
    opts = default_options(**case["options"])
    out = io.StringIO()
    vcf = VCF(names)
    for c, s, e, b in work:
        caller.callVariantsInRegion(c, s, e, b, fasta, opts, vcf, out)
    return out.getvalue().split("\n")[:-1], opts.rlen


def native_loop_text(case, lib=None, workers=2, per_chunk=2):
    fasta, work, names = case_work(case)
    opts = default_options(**case["options"])
    nc = F.NativeCaller(0, workers, per_chunk, lib=lib) if lib is not None else F.NativeCaller(0, workers, per_chunk)
    try:
        txt = nc.call_regions([F.RegionReads.from_buffers(c, s, e, fasta, b) for c, s, e, b in work], names, opts)
        failed = nc.stats["n_windows_failed"]
    finally:
        nc.close()
    return txt.split("\n")[:-1], opts.rlen, failed


def diff(got, want):
    """First differing record, for the assertion message."""
    for k, (a_, b_) in enumerate(zip(got, want)):
        if a_ != b_:
            fa, fb = a_.split("\t"), b_.split("\t")
            cols = [i for i in range(min(len(fa), len(fb))) if fa[i] != fb[i]]
            return "line %d differs in columns %s:\n got  %s\n want %s" % (k, cols, a_[:600], b_[:600])
    return "%d lines instead of %d; first extra / missing: %s" % (len(got), len(want), (got[len(want):] or want[len(got):])[0][:300])
