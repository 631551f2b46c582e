"""The region loop's HOST logic, end to end on the CPU: platypus_amd.caller (Python mirror of the reference's
callVariantsInRegion) and the native libplat_caller.so (platypus_amd/csrc/host) must write the same VCF record text.

No GPU here, so both run on tests/fakedev: the C ABI of include/platypus_mi355x.h implemented with the parity oracle (test
infrastructure only; the product binds the HIP library and has no CPU path).  The GPU suite repeats the comparison on the real
device (tests/test_gpu_caller.py) and checks the fake against it."""
import io

import numpy as np
import pytest

from platypus_amd import caller, fastcaller as F, hostapi as H, synth
from platypus_amd.options import default_options
from platypus_amd.vcfrecords import VCF


@pytest.fixture(scope="module")
def fake():
    from tests import fakedev
    old = H._engine
    H._engine = fakedev.fake_engine()
    yield fakedev.fake_caller_lib()
    H._engine = old


def _work(regs, names, extra=None):
    fasta = H.FastaFile({r["chrom"]: r["ref"] for r in regs})
    out = []
    for k, r in enumerate(regs):
        bufs = []
        for i, reads in enumerate(r["samples"]):
            rs = [H.AlignedRead(x["seq"], x["qual"], x["pos"], x["mapq"], x["flag"], end=x["end"], cigarOps=x["cigar"], matePos=x.get("matePos", -1))
                  for x in reads]
            good, bad, broken = rs, [], []
            if extra:
                good, bad, broken = extra(k, i, rs)
            bufs.append(H.bamReadBuffer(good, bad, broken, sample=names[i]))
        out.append((r["chrom"], r["start"], r["end"], bufs))
    return fasta, out


def _both(lib, regs, names, extra=None, workers=2, per_chunk=2, **opt):
    fasta, work = _work(regs, names, extra)
    o1 = default_options(**opt)
    py = io.StringIO()
    nw = caller.callVariantsInRegions(work, fasta, o1, VCF(names), py)
    nc = F.NativeCaller(0, workers, per_chunk, lib=lib)
    o2 = default_options(**opt)
    txt = nc.call_regions([F.RegionReads.from_buffers(c, s, e, fasta, b) for c, s, e, b in work], names, o2)
    st = nc.stats
    nc.close()
    assert txt.split("\n") == py.getvalue().split("\n")
    assert o1.rlen == o2.rlen and st["n_windows"] == nw and st["n_records"] == txt.count("\n")
    return txt, st


def test_native_equals_python_one_and_three_samples(fake):
    for ns, n, kw in ((1, 3, dict(region_len=4000, snp_rate=3e-3, indel_rate=1e-3, read_len=100, depth=30)),
                      (3, 2, dict(region_len=5000, snp_rate=4e-3, indel_rate=2e-3, read_len=150, depth=25))):
        regs = [synth.config4_region(10 * ns + i, n_samples=ns, **kw) for i in range(n)]
        txt, st = _both(fake, regs, ["S%d" % (i + 1) for i in range(ns)])
        assert st["n_records"] > 15 and st["n_windows_failed"] == 0


def test_native_equals_python_greedy_haplotype_filter(fake):
    """More than five variants in a window: getFilteredHaplotypes grows the best haplotypes greedily, one alignment batch per
    variant (variantFilter.pyx:440-506); coverage filter on and off, a small maxHaplotypes."""
    regs = [synth.config4_region(70 + i, n_samples=2, region_len=2500, snp_rate=8e-2, indel_rate=1e-2, read_len=100, depth=30) for i in range(2)]
    txt, st = _both(fake, regs, ["A", "B"])
    assert st["n_windows_greedy"] > 20
    txt, st = _both(fake, regs[:1], ["A", "B"], filterVarsByCoverage=0, maxVariants=12, maxHaplotypes=20)
    assert st["n_windows_greedy"] > 5


def test_native_equals_python_bad_reads_broken_mates_and_empty_samples(fake):
    """badReads and brokenMates take part in the likelihoods and the read statistics (chaplotype.pyx:341-373, vcfutils.pyx:1300-1390);
    a sample without reads in one region, a region without reads at all, regions of different read lengths (options.rlen)."""
    rng = np.random.default_rng(8)
    regs = [synth.config4_region(90 + i, n_samples=2, region_len=3000, snp_rate=4e-3, indel_rate=1.5e-3, read_len=[100, 76, 125][i], depth=40)
            for i in range(3)]
    regs.append(synth.config4_region(99, n_samples=2, region_len=1500, read_len=100, depth=20))
    regs[3]["samples"] = [[], []]

    def split(k, i, rs):
        if k == 1 and i == 1:
            return [], [], []
        good, bad, broken = [], [], []
        for r in rs:
            u = rng.random()
            if u < 0.08:
                r.mapq = int(rng.integers(0, 20)); r.bitFlag |= 512
                bad.append(r)
            elif u < 0.12:
                r.matePos = r.pos + int(rng.integers(-400, 400))
                broken.append(r)
            else:
                good.append(r)
        return good, bad, broken
    txt, st = _both(fake, regs, ["S1", "S2"], extra=split, workers=3, per_chunk=1)
    assert st["n_records"] > 20


@pytest.mark.parametrize("opt", [dict(minPosterior=0), dict(maxVariants=3, minPosterior=20), dict(mergeClusteredVariants=0, minReads=3),
                                 dict(largeWindows=1, maxVarDist=30, minVarDist=5), dict(skipDifficultWindows=1, maxVariants=4),
                                 dict(genIndels=0), dict(minVarFreq=0.2, badReadsThreshold=30, qdThreshold=30, hapScoreThreshold=1)])
def test_native_equals_python_option_variants(fake, opt):
    regs = [synth.config4_region(200 + i, n_samples=1, region_len=3000, snp_rate=1.2e-2, indel_rate=3e-3, read_len=100, depth=30) for i in range(2)]
    _both(fake, regs, ["S1"], **opt)


def test_native_many_chunks_per_worker(fake):
    """Several chunks per worker, one region per chunk, a region without reads in the middle: the same text whatever the split."""
    regs = [synth.config4_region(40 + i, n_samples=2, region_len=2500, snp_rate=4e-3, indel_rate=1e-3, read_len=100, depth=20) for i in range(7)]
    regs[3]["samples"] = [[], []]
    names = ["A", "B"]
    a, _ = _both(fake, regs, names, workers=1, per_chunk=1)
    b, _ = _both(fake, regs, names, workers=2, per_chunk=1)
    c, _ = _both(fake, regs, names, workers=2, per_chunk=3)
    assert a == b == c and a.count("\n") > 20


def test_native_refuses_what_it_does_not_build(fake):
    regs = [synth.config4_region(1, region_len=1000, read_len=100)]
    fasta, work = _work(regs, ["S1"])
    nc = F.NativeCaller(0, 1, 1, lib=fake)
    from platypus_amd._lib import PlatypusDeviceError
    for bad in (dict(getVariantsFromBAMs=0),):                              # no candidate source at all (source VCFs are not built)
        with pytest.raises(PlatypusDeviceError) as e:
            nc.call_regions([F.RegionReads.from_buffers(c, s, e_, fasta, b) for c, s, e_, b in work], ["S1"], default_options(**bad))
        assert e.value.code == -6


def _split_classes(rng):
    def split(k, i, rs):
        good, bad, broken = [], [], []
        for r in rs:
            u = rng.random()
            if u < 0.08:
                r.mapq = int(rng.integers(0, 20)); r.bitFlag |= 512 if u < 0.04 else 0
                bad.append(r)
            elif u < 0.12:
                r.matePos = r.pos + int(rng.integers(-400, 400))
                broken.append(r)
            else:
                good.append(r)
        return good, bad, broken
    return split


@pytest.mark.parametrize("opt", [dict(assemble=1), dict(assemble=1, getVariantsFromBAMs=0), dict(assemble=1, assemblyRegionSize=700, assembleBrokenPairs=1, assembleBadReads=0),
                                 dict(assemble=1, assembleAll=0), dict(assemble=1, noCycles=1, assemblerKmerSize=21)])
def test_native_equals_python_with_the_assembler(fake, opt):
    """--assemble=1 (variantcaller.pyx:496-519, 276-321; assembler.pyx:1391-1476): tiles of every region of a chunk assembled in one device
    batch, their variants join the BAM candidates (or stand alone), same text as the Python region loop -- tiling, read selection
    (badReads / brokenMates / QCFail), doWeNeedToAssembleThisRegion, sources in the INFO field."""
    rng = np.random.default_rng(12)
    regs = [synth.config4_region(700 + i, n_samples=2, region_len=2600, snp_rate=3e-3, indel_rate=2.5e-3, read_len=100, depth=30) for i in range(3)]
    regs[1]["samples"][1] = []
    txt, st = _both(fake, regs, ["S1", "S2"], extra=_split_classes(rng), workers=2, per_chunk=2, **opt)
    assert st["n_assembly_tiles"] >= (6 if opt.get("assembleAll", 1) else 0)
    if opt.get("assembleAll", 1):
        assert st["n_assembler_variants"] > 5 and "Source=Assembler" in txt or "Platypus,Assembler" in txt
    if not opt.get("getVariantsFromBAMs", 1):
        assert "Source=Platypus" not in txt and txt.count("\n") > 3


@pytest.mark.parametrize("opt", [dict(outputRefCalls=1), dict(outputRefCalls=1, refCallBlockSize=150, minPosterior=60),
                                 dict(outputRefCalls=1, refCallBlockSize=37, maxVariants=2, skipDifficultWindows=1), dict(outputRefCalls=1, assemble=1)])
def test_native_equals_python_with_reference_calls(fake, opt):
    """--outputRefCalls=1 (variantcaller.pyx:584-607,764-867; window.py:172-219): REFCALL lines for the blocks between calling windows,
    between the called positions of a window, for windows without a call and for windows the loop leaves uncalled -- the text of the
    Python region loop (itself pinned by refcall_cases / regionprep_cases and its window-by-window shape)."""
    regs = [synth.config4_region(400 + i, n_samples=2, region_len=2500, snp_rate=5e-3, indel_rate=1.5e-3, read_len=100, depth=[30, 4, 12][i % 3]) for i in range(4)]
    regs[2]["samples"][0] = []                                             # a sample without reads: blocks without coverage
    txt, st = _both(fake, regs, ["S1", "S2"], workers=2, per_chunk=2, **opt)
    assert st["n_refcall_records"] == txt.count("\tREFCALL\t") >= 20 and st["n_records"] - st["n_refcall_records"] >= 5


def test_native_equals_python_on_array_regions(fake):
    """synth.config4_region_arrays (the generator of the config-4 benchmark): arrays straight into the native loop, the same reads
    as objects into the Python loop."""
    regs = [synth.config4_region_arrays(300 + i, region_len=6000, snp_rate=3e-3, indel_rate=1e-3, n_samples=2, read_len=150, depth=30) for i in range(3)]
    names = ["S1", "S2"]
    fasta = H.FastaFile({r["chrom"]: r["ref"].tobytes() for r in regs})
    work = [(r["chrom"], r["start"], r["end"], [H.bamReadBuffer(F.aligned_reads_from_arrays(s), sample=names[i]) for i, s in enumerate(r["samples"])])
            for r in regs]
    o1, o2 = default_options(), default_options()
    py = io.StringIO()
    caller.callVariantsInRegions(work, fasta, o1, VCF(names), py)
    nc = F.NativeCaller(0, 2, 2, lib=fake)
    txt = nc.call_regions([F.region_from_arrays(r) for r in regs], names, o2)
    assert txt == py.getvalue() and txt.count("\n") > 30
    # the planted variants come back
    called = {(ln.split("\t")[0], int(ln.split("\t")[1])) for ln in txt.split("\n")[:-1]}
    snps = [(r["chrom"], p + 1) for r in regs for (p, rem, add), t0, t1 in zip(r["variants"], r["truth"][0], r["truth"][1]) if len(rem) == len(add) and t0 + t1 > 0]
    assert sum(k in called for k in snps) >= 0.9 * len(snps)


def test_reference_call_blocks_batched_equals_window_by_window(fake):
    """--outputRefCalls=1 (variantcaller.pyx:584-607,764-867): REFCALL lines for the blocks between calling windows, between the
    called positions of a window and for windows without a call; the batched shape of the Python region loop writes what the
    window-by-window shape writes."""
    regs = [synth.config4_region(400 + i, n_samples=2, region_len=2500, snp_rate=5e-3, indel_rate=1.5e-3, read_len=100, depth=[30, 4][i % 2]) for i in range(3)]
    names = ["S1", "S2"]
    fasta, work = _work(regs, names)
    for opt in (dict(outputRefCalls=1), dict(outputRefCalls=1, refCallBlockSize=150, minPosterior=60)):
        one = io.StringIO()
        for c, s_, e, bufs in _work(regs, names)[1]:
            caller.callVariantsInRegion(c, s_, e, bufs, fasta, default_options(**opt), VCF(names), one)
        many = io.StringIO()
        caller.callVariantsInRegions(_work(regs, names)[1], fasta, default_options(**opt), VCF(names), many)
        lines = many.getvalue().split("\n")[:-1]
        assert one.getvalue() == many.getvalue()
        ref = [ln for ln in lines if "\tREFCALL\t" in ln]
        var = [ln for ln in lines if "\tREFCALL\t" not in ln]
        assert len(ref) >= 20 and len(var) >= 5
        # the blocks tile the gaps: every REFCALL line carries END and Size = END - POS + 1 - 1 (0-based start, half-open end)
        for ln in ref:
            f = ln.split("\t")
            info = dict(kv.split("=") for kv in f[7].split(";"))
            assert int(info["Size"]) == int(info["END"]) - (int(f[1]) - 1) and f[4] in ("N", "T") and f[9].startswith("./.:-1,-1,-1:-1:-1:")


def test_runtime_errors_end_the_call_and_window_errors_skip_the_window(fake, monkeypatch):
    """A device error that no single window can be guilty of (HIP runtime, out of memory) in the window batch must end plat_call_regions
    with that code -- not be swallowed window by window with PLAT_OK and records missing; an input-class error is retried per window
    and only costs the windows that fail again (variantcaller.pyx:568-615)."""
    import ctypes as C
    from platypus_amd._lib import PlatypusDeviceError
    from tests import fakedev
    dev = C.CDLL(fakedev.FAKE_LIB)
    regs = [synth.config4_region(500, n_samples=1, region_len=3000, snp_rate=4e-3, indel_rate=1e-3, read_len=100, depth=30)]
    fasta, work = _work(regs, ["S1"])
    rr = [F.RegionReads.from_buffers(c, s, e, fasta, b) for c, s, e, b in work]
    nc = F.NativeCaller(0, 1, 1, lib=fake)
    good = nc.call_regions(rr, ["S1"], default_options())
    assert good.count("\n") > 5 and nc.stats["n_windows_called"] == nc.stats["n_windows"]
    # syncs of a chunk: 1 = candidate scan, 2 = window batch (no greedy rounds here)
    for code in (-2, -3):                                                    # PLAT_ERR_HIP, PLAT_ERR_NOMEM
        dev.plat_fake_reset_sync_count()
        monkeypatch.setenv("PLAT_FAKE_FAIL_SYNC", "%d:2" % code)
        with pytest.raises(PlatypusDeviceError) as e:
            nc.call_regions(rr, ["S1"], default_options())
        assert e.value.code == code
    dev.plat_fake_reset_sync_count()
    monkeypatch.setenv("PLAT_FAKE_FAIL_SYNC", "-9:2")                        # PLAT_ERR_BAD_INPUT once: the retry goes through
    again = nc.call_regions(rr, ["S1"], default_options())
    assert again == good and nc.stats["n_windows_failed"] == 0 and nc.stats["n_windows_called"] == nc.stats["n_windows"]
    monkeypatch.delenv("PLAT_FAKE_FAIL_SYNC")
    nc.close()


def two_allele_regions():
    """Regions with sites where the reads show TWO alternative SNP alleles, each in the same number of reads (a tie of type, length
    and support)."""

    rng = np.random.default_rng(21)
    regs = [synth.config4_region(900 + i, n_samples=2, region_len=3000, snp_rate=2e-3, indel_rate=5e-4, read_len=100, depth=40) for i in range(4)]
    n_sites = 0
    for r in regs[:3]:                                                      # (the fourth region stays as it is: no pair, no replay)
        taken = {v[0] for v in r["variants"]}
        sites = [p for p in rng.choice(np.arange(r["start"] + 150, r["end"] - 150), size=40, replace=False).tolist()
                 if all(abs(p - q) > 12 for q in taken)][:7]
        for p in sites:
            taken.add(p)
            ref_b = r["ref"][p]
            alts = [b for b in b"ACGT" if b != ref_b]
            a1, a2 = (alts[k] for k in rng.choice(3, size=2, replace=False))
            for reads in r["samples"]:                                      # the same number of reads for either allele: a tie of support too
                cover = [x for x in reads if x["pos"] + 12 <= p < x["pos"] + len(x["seq"]) - 12 and len(x["cigar"]) == 1]
                for k, x in enumerate(cover[:2 * min(14, len(cover) // 2)]):
                    s = bytearray(x["seq"])
                    s[p - x["pos"]] = a1 if k % 2 == 0 else a2
                    x["seq"] = bytes(s)
            n_sites += 1
    return regs, n_sites


def test_sites_with_two_alternative_alleles_follow_the_python2_dictionary_order(fake):
    """Two SNP alleles at one position compare equal under Variant.__richcmp__ (refPos, type, nRemoved): `sorted` keeps them in the
    order the candidate dictionaries yield them, Python-2 dicts keyed by hash((refName, refPos, removed, added)) (variant.pyx:270-280,
    :747-751; variantcaller.pyx:457).  Both region loops replay those dictionaries for a region that holds such a pair (and only
    then); the order they arrive at differs from first-occurrence order for some of the sites below, and decides ALT / PP / FR / NF / NR
    of the records."""
    regs, n_sites = two_allele_regions()
    txt, st = _both(fake, regs, ["A", "B"])
    multi = [ln.split("\t") for ln in txt.split("\n") if ln and "," in ln.split("\t")[4]]
    assert n_sites >= 15 and len(multi) >= 3, (n_sites, len(multi))
    # the dictionary order is not the order of first occurrence: for some of the pairs the candidate list holds the alleles the other way
    # round than the reads showed them (what reaches the record text of THESE windows is then ordered by haplotype sequence,
    # mergeHaplotypes, variantcaller.pyx:325-383 -- the candidate order decides where support ties are broken: greedy haplotype
    # filter, coverage filter)
    fasta, work = _work(regs, ["A", "B"])
    cands, _ = caller.generateVariantsInRegions(work, fasta, default_options())
    flipped = pairs = 0
    for r, vs in zip(regs, cands):
        first = {}
        for x in [x for reads in r["samples"][:1] for x in reads]:
            if len(x["cigar"]) == 1:
                for v in vs:
                    if v.nAdded == 1 and v.nRemoved == 1 and x["pos"] <= v.refPos < x["pos"] + len(x["seq"]) and x["seq"][v.refPos - x["pos"]] == v.added[0]:
                        first.setdefault((v.refPos, v.added), len(first))
        for a, c in zip(vs, vs[1:]):
            if a.refPos == c.refPos and a.nAdded == c.nAdded == 1 and a.nRemoved == c.nRemoved == 1 and (a.refPos, a.added) in first and (c.refPos, c.added) in first:
                pairs += 1
                flipped += first[(a.refPos, a.added)] > first[(c.refPos, c.added)]
    assert pairs >= 12 and 0 < flipped < pairs, (pairs, flipped)
    # ... and where support ties are broken it reaches the text: with one variant per window allowed (filterVariantsByCoverage keeps the
    # best supported, the first of equals) the surviving allele of such a pair is the dictionary's first.  Native = Python there too,
    # and the native loop without its replay (a switch for this test) calls other alleles.
    top, st = _both(fake, regs, ["A", "B"], maxVariants=1)
    import os
    os.environ["PLAT_CALLER_FIRST_OCCURRENCE_ORDER"] = "1"
    try:
        nc = F.NativeCaller(0, 2, 2, lib=fake)
        plain = nc.call_regions([F.RegionReads.from_buffers(c, s_, e, fasta, b) for c, s_, e, b in work], ["A", "B"], default_options(maxVariants=1))
        nc.close()
    finally:
        os.environ.pop("PLAT_CALLER_FIRST_OCCURRENCE_ORDER")
    assert plain != top and plain.count("\n") == top.count("\n")
    os.environ["PLAT_CALLER_HOST_TALLY"] = "1"                              # the candidate tally on the host takes the same way
    try:
        nc = F.NativeCaller(0, 2, 2, lib=fake)
        assert nc.call_regions([F.RegionReads.from_buffers(c, s_, e, fasta, b) for c, s_, e, b in work], ["A", "B"], default_options(maxVariants=1)) == top
        nc.close()
    finally:
        os.environ.pop("PLAT_CALLER_HOST_TALLY")


def test_the_loaders_own_figures_for_its_tables(fake, monkeypatch):
    """plat_read_table.longest_read / most_bases (optional: what ReadArray knows of itself as reads are appended, cwindow.pyx:173-174,272): the same
    text with the figures (the tables of fastcaller.ReadTable carry them), without them (0: the library walks the arrays) -- and, with
    PLAT_CALLER_CHECK_HINTS=1, a figure that does not describe its table is refused instead of believed."""
    monkeypatch.setenv("PLAT_CALLER_CHECK_HINTS", "1")
    regs = [synth.config4_region(60 + i, n_samples=2, region_len=2500, snp_rate=4e-3, indel_rate=2e-3, read_len=100, depth=20) for i in range(3)]
    names = ["A", "B"]
    fasta, work = _work(regs, names)

    def call(edit=None):
        rr = [F.RegionReads.from_buffers(c, s, e, fasta, b) for c, s, e, b in work]
        for r in rr:
            for tabs in r.samples:
                for t in tabs:
                    st = t.struct()
                    if t.n:
                        assert st.longest_read == int((t.end.astype(np.int64) - t.pos).max()) and st.most_bases == int(np.diff(t.off).max())
                    if edit:
                        edit(st)
        nc = F.NativeCaller(0, 2, 2, lib=fake)
        try:
            return nc.call_regions(rr, names, default_options())
        finally:
            nc.close()

    with_figures = call()
    assert with_figures.count("\n") > 10

    def forget(st):
        st.longest_read = 0; st.most_bases = 0
    assert call(forget) == with_figures

    def lie(st):
        if st.n_reads:
            st.longest_read += 1
    with pytest.raises(Exception):
        call(lie)
