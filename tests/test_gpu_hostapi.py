"""GPU tests of the drop-in interface: Haplotype.alignReads / alignSingleRead, DiploidGenotype, Population.setup and
assembleReadsAndDetectVariants through the reference-named classes, checked against the oracle."""
import numpy as np
import pytest

from platypus_amd import hostapi as H

pytestmark = pytest.mark.gpu


def make_window(seed=5, n_ind=2):
    rng = np.random.default_rng(seed)
    B = b"ACGT"
    ref = bytes(rng.choice(list(B), 3000).tolist())
    fasta = H.FastaFile({"20": ref})
    ws, we, L = 1500, 1560, 100
    snp = H.Variant("20", 1520, ref[1520:1521], b"A" if ref[1520:1521] != b"A" else b"C")
    dele = H.Variant("20", 1535, ref[1536:1539], b"")
    haps = [H.Haplotype("20", ws, we, v, fasta, L) for v in ((), (snp,), (dele,), (snp, dele))]
    donors = [h.haplotypeSequence for h in haps]
    buffers = []
    for i in range(n_ind):
        good, bad = [], []
        for _ in range(40):
            d = donors[int(rng.integers(0, 4))]
            off = int(rng.integers(120, 280))
            seq = bytearray(d[off:off + L])
            if rng.random() < 0.2:
                seq[int(rng.integers(0, L))] = B[int(rng.integers(0, 4))]
            q = bytes(np.clip(rng.normal(33, 6, L), 2, 41).astype(np.uint8).tolist())
            pos = ws - haps[0].endBufferSize + off
            r = H.AlignedRead(bytes(seq), q, pos, mapq=int(rng.choice([60, 60, 29, 0])))
            if rng.random() < 0.15:
                r.bitFlag |= 512 if rng.random() < 0.5 else 0
                bad.append(r)
            else:
                good.append(r)
        buffers.append(H.bamReadBuffer(good, bad))
        buffers[-1].setWindowPointers(ws, we)
    return fasta, haps, buffers, ws, we


def oracle_rows(oracle, haps, buf, ws, we, do_flank=0):
    rs = buf.windowReads()
    reads = dict(seq=[r.seq for r, _ in rs], qual=[r.qual for r, _ in rs], pos=[r.pos for r, _ in rs],
                 end=[r.end for r, _ in rs], mapq=[r.mapq for r, _ in rs], flags=[r.bitFlag for r, _ in rs],
                 kind=[k for _, k in rs])
    return oracle.align_window([h.haplotypeSequence for h in haps], ws, we, haps[0].endBufferSize, reads, do_flank=do_flank)[0]


def test_alignReads_cache_layout_and_values(oracle):
    fasta, haps, buffers, ws, we = make_window()
    exp = oracle_rows(oracle, haps, buffers[0], ws, we)
    for hi, h in enumerate(haps):
        arr = h.alignReads(0, buffers[0])
        assert arr[-1] == 999 and len(arr) == exp.shape[1] + 1          # chaplotype.pyx:375
        assert np.array_equal(arr[:-1], exp[hi])
        assert h.alignReads(0, buffers[0]) is arr                       # cached while individualIndex is unchanged (:320,339)
    r = buffers[0].reads.window()[3]
    one = haps[1].alignSingleRead(r)
    k = [x for x, _ in buffers[0].windowReads()].index(r)
    assert one == exp[1][k]


def test_population_setup_matches_oracle(oracle):
    fasta, haps, buffers, ws, we = make_window(n_ind=3)
    genotypes = H.generateAllGenotypesFromHaplotypeList(haps)
    pop = H.Population().setup([], haps, genotypes, 3, 0, buffers)
    assert pop.haplotypeIndexes.tolist()[:5] == [[0, 0], [0, 1], [0, 2], [0, 3], [1, 1]]
    for i, buf in enumerate(buffers):
        rows = oracle_rows(oracle, haps, buf, ws, we)
        logl, gl, gof = oracle.population_setup_ind(rows, buf.reads.windowEnd - buf.reads.windowStart)
        assert np.allclose(pop.genotypeLogLikelihoods[i], logl, rtol=1e-12, atol=0)      # device libm only in one rare branch
        assert np.allclose(pop.genotypeLikelihoods[i], gl, rtol=1e-10, atol=1e-300)
        assert np.allclose(pop.goodnessOfFitValues[:, i], gof, rtol=1e-13, atol=0)
        assert pop.nReads[i] == buf.reads.windowEnd - buf.reads.windowStart
    g = genotypes[1]
    gofv = np.zeros(3)
    L = g.calculateDataLikelihood(buffers[1], 1, 3, gofv)
    assert np.isclose(L, pop.genotypeLogLikelihoods[1][1], rtol=1e-12, atol=0) and gofv[1] != 0


def test_population_call_matches_oracle(oracle):
    """Population.call: EM frequencies, EM likelihoods, genotype calls, variant posteriors, per-site marginalisation."""
    fasta, haps, buffers, ws, we = make_window(n_ind=3)
    for h in haps:
        for v in h.variants:
            if v.nAdded != v.nRemoved:
                v.prior = 1e-4                                         # indel priors are host logic (variant.pyx:146-217)
    genotypes = H.generateAllGenotypesFromHaplotypeList(haps)
    pop = H.Population().setup([], haps, genotypes, 3, 0, buffers).call(100, 0)
    f, em, calls, iters, _ = oracle.em_call(pop.nReads, pop.genotypeLikelihoods, 100, 0)
    assert iters == pop.emIterations and np.array_equal(f, pop.frequencies)
    assert np.array_equal(em, pop.EMLikelihoods)
    assert [genotypes[c] if c >= 0 else None for c in calls] == pop.genotypeCalls
    variants = sorted({v for h in haps for v in h.variants})
    assert len(variants) == 2
    for v in variants:
        mask = [v in h.variants for h in haps]
        exp = oracle.variant_posterior(pop.nReads, pop.genotypeLikelihoods, f, mask, v.calculatePrior())
        assert pop.calculatePosterior(v) == exp
        assert (pop.variantPosteriors.get(v) == exp) if exp >= 5 else (v not in pop.variantPosteriors)
    v = variants[0]
    isref = np.array([v not in h.variants for h in haps], dtype=np.int32)
    got = pop.computeGenotypeCallAndLikelihoods(1, [v], isref)
    ph, lik, out4 = oracle.genotype_call(f, pop.genotypeLikelihoods[1], pop.goodnessOfFitValues[:, 1],
                                         np.array([[v in h.variants] for h in haps]), isref, 3)
    assert got[:2] == tuple(ph.tolist()) and got[2] == lik.tolist() and got[3:] == tuple(out4.tolist())


def test_getFilteredHaplotypes_matches_reference_golden(golden_dir):
    """haplotype enumeration / greedy haplotype filter (variantFilter.pyx:377-506 with
    computeBestScoreForGenotype :237-283) -- the surviving variant combinations, in the reference's order.  Every alignment
    runs on the device (one batch per greedy step); the per-sample sums use the C library's log/exp on the host."""
    import gzip, json, os
    from platypus_amd.options import default_options
    cases = json.load(gzip.open(os.path.join(golden_dir, "filter_cases.json.gz"), "rt"))["filter"]
    greedy = 0
    for c in cases:
        fasta = H.FastaFile({"20": c["ref"].encode()})
        variants = [H.Variant("20", v["pos"], v["removed"].encode(), v["added"].encode(), v["n_supporting"]) for v in c["variants"]]
        assert variants == sorted(variants)
        buffers = []
        for reads in c["samples"]:
            buf = H.bamReadBuffer([H.AlignedRead(r["seq"].encode(), bytes(r["qual"]), r["pos"], mapq=r["mapq"], bitFlag=r["flag"]) for r in reads])
            buf.reads.windowStart, buf.reads.windowEnd = 0, len(reads)
            buffers.append(buf)
        opt = default_options(rlen=c["rlen"], maxHaplotypes=c["max_haplotypes"], coverageSamplingLevel=c["coverage_sampling_level"])
        refHap = H.Haplotype("20", c["start"], c["end"], (), fasta, c["rlen"], opt)
        haps = H.getFilteredHaplotypes("20", c["start"], c["end"], fasta, opt, variants, refHap, buffers)
        got = [[variants.index(v) for v in h.variants] for h in haps]
        assert got == c["haplotypes"]
        greedy += len(variants) > 5
    assert greedy >= 8


def test_calculateFlankScore_option(oracle):
    fasta, haps, buffers, ws, we = make_window()
    exp0 = oracle_rows(oracle, haps, buffers[0], ws, we)
    exp1 = oracle_rows(oracle, haps, buffers[0], ws, we, do_flank=1)
    assert not np.array_equal(exp0, exp1)
    for h in haps:
        h.options.calculateFlankScore = 1                                  # runner.py:559
    for hi, h in enumerate(haps):
        assert np.array_equal(h.alignReads(5, buffers[0])[:-1], exp1[hi])


def test_unsupported_modes_raise():
    fasta, haps, buffers, ws, we = make_window()
    from platypus_amd._lib import PlatypusDeviceError
    with pytest.raises(PlatypusDeviceError):
        haps[0].alignSingleRead(buffers[0].reads.window()[0], useMapQualCap=True)


def test_assembleReadsAndDetectVariants(oracle):
    rng = np.random.default_rng(12)
    B = b"ACGT"
    ref = bytes(rng.choice(list(B), 3000).tolist())
    donor = bytearray(ref); donor[1500:1500] = b"GATTACAGATT"; del donor[1800:1806]; donor[1200] = ord("A") if donor[1200] != ord("A") else ord("C")
    donor = bytes(donor)
    good, bad = [], []
    for _ in range(360):
        p = int(rng.integers(0, len(donor) - 250))
        r = H.AlignedRead(donor[p:p + 250], bytes([35] * 250), 1000 + p)
        (bad if rng.random() < 0.1 else good).append(r)
    bad[0].bitFlag |= 512
    buf = H.bamReadBuffer(good, bad)
    buf.setWindowPointers(1000 + 700, 1000 + 2200)
    out = H.assembleReadsAndDetectVariants("20", 1000 + 750, 1000 + 2250, 1000, 4000, [buf], ref)
    seqs = [r.seq for r in buf.reads.window()] + [r.seq for r in buf.badReads.window() if not r.isQCFail()]
    quals = [r.qual for r in buf.reads.window()] + [r.qual for r in buf.badReads.window() if not r.isQCFail()]
    exp, _ = oracle.assemble(ref, 1000, 1750, 3250, seqs, quals)
    assert [(v.refPos, v.removed, v.added) for v in out] == exp
    assert len(out) >= 3 and all(v.varSource == H.ASSEMBLER_VAR for v in out)


def test_cli_synthetic_run_writes_records(tmp_path):
    """`python -m platypus_amd callVariants --synthetic ...`: alignReads + genotype likelihoods + EM for every window, one
    record per window in --output; BAM input is refused loudly (out of scope)."""
    import json, subprocess, sys
    out = tmp_path / "calls.txt"
    r = subprocess.run([sys.executable, "-m", "platypus_amd", "callVariants", "--synthetic", "config2:40", "--output", str(out),
                        "--calculateFlankScore", "1"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-500:]
    info = json.loads(r.stdout.strip().splitlines()[-1])
    lines = out.read_text().strip().split("\n")
    assert info["windows"] == 40 == len(lines)
    f = lines[0].split("\t")
    nh = int(f[2])
    assert len(f[3].split(",")) == nh * (nh + 1) // 2 and len(f[4].split(",")) == nh and abs(sum(map(float, f[4].split(","))) - 1) < 1e-3
    r = subprocess.run([sys.executable, "-m", "platypus_amd", "callVariants", "--bamFiles", "x.bam"], capture_output=True, text=True)
    assert r.returncode != 0 and "outside this build's scope" in r.stderr


def test_variant_candidates_match_reference_golden(golden_dir):
    """VariantCandidateGenerator (variant.pyx:459-751) -- device scan + host merge against outputs of the
    reference's own text: same candidates, same supporting-read counts, same getCandidates() order."""
    import gzip, json, os
    cases = json.load(gzip.open(os.path.join(golden_dir, "candidate_cases.json.gz"), "rt"))
    n = 0
    for c in cases:
        fasta = H.FastaFile({"20": c["ref"].encode()})
        reads = [H.AlignedRead(r["seq"].encode(), bytes(r["qual"]), r["pos"], bitFlag=r["flag"], cigarOps=r["cigar"]) for r in c["reads"]]
        gen = H.VariantCandidateGenerator(("20", c["start"], c["end"]), fasta, 20, c["min_flank"], c["min_base_qual"], 5000000, 150,
                                          None, 0, c["gen_snps"], c["gen_indels"])
        gen.addCandidatesFromReads(reads)
        got = [[v.refPos, v.removed.decode(), v.added.decode(), v.nSupportingReads] for v in gen.getCandidates(0)]
        assert got == c["sorted"]
        n += len(got)
    assert n > 5000


def test_check_and_trim_reads_match_reference_golden(golden_dir):
    """checkAndTrimRead (cwindow.pyx:332-481) on the device: accept/reject, QCFail flags, trimmed qualities, per-type counts."""
    import gzip, json, os
    from platypus_amd.options import default_options
    cases = json.load(gzip.open(os.path.join(golden_dir, "readqc_cases.json.gz"), "rt"))
    trimmed = 0
    for c in cases:
        o = c["options"]
        opt = default_options(minGoodQualBases=o["minGoodQualBases"], minMapQual=o["minMapQual"], minBaseQual=o["minBaseQual"],
                              trimOverlapping=o["trimOverlapping"], trimAdapter=o["trimAdapter"], trimReadFlank=o["trimReadFlank"],
                              trimSoftClipped=o["trimSoftClipped"])
        reads = [H.AlignedRead(r["seq"].encode(), bytes(r["qual"]), r["pos"], mapq=r["mapq"], bitFlag=r["flag"], cigarOps=r["cigar"],
                               chromID=r["chromID"], mateChromID=r["mateChromID"], insertSize=r["insertSize"], matePos=r["matePos"])
                 for r in c["reads"]]
        ok, counts = H.checkAndTrimReads(reads, opt, o["enabled"])
        assert [int(x) for x in ok] == c["ok"] and counts == c["counts"]
        assert [r.bitFlag for r in reads] == c["flag_out"]
        for r, src, exp in zip(reads, c["reads"], c["qual_out"]):
            assert list(r.qual) == (src["qual"] if exp is None else exp)
            trimmed += exp is not None
    assert trimmed > 500


def test_vcf_records_end_to_end_match_reference_golden(golden_dir, oracle):
    """reads + haplotypes -> Population.setup / call(computeVCFFields=1) on the device -> INFO, FILTER, VCF text: every value the
    reference's texts produced for the 48 windows of vcf_cases.json.gz (likelihoods, hapLikes, frequencies, posteriors, calls,
    HapScore, INFO dictionaries, filters, record lines)."""
    import gzip, io, json, os
    from test_hostapi_cpu import _vcf_case_objects, _info_equal
    from platypus_amd import vcfrecords as V
    from platypus_amd.options import default_options
    cases = json.load(gzip.open(os.path.join(golden_dir, "vcf_cases.json.gz"), "rt"))
    nlines = 0
    for c in cases:
        fasta, variants, haps, buffers, o = _vcf_case_objects(c)
        opts = default_options(**vars(o))
        opts.rlen = c["rlen"]
        genotypes = H.generateAllGenotypesFromHaplotypeList(haps)
        pop = H.Population(opts)
        pop.refFile = fasta
        pop.setup(variants, haps, genotypes, len(buffers), 0, buffers)
        nInd, nH = len(buffers), len(haps)
        for i in range(nInd):
            tot = len(c["samples"][i]["good"]) + len(c["samples"][i]["bad"])
            if tot:
                assert np.array_equal(pop.haplotypeLikelihoods[:, sum(len(s["good"]) + len(s["bad"]) for s in c["samples"][:i]):][:, :tot],
                                      np.array(c["loglik"][i]).reshape(nH, tot))
        assert [g.hap1Like for g in genotypes if g.hap1 is g.hap2] == c["hap_likes"]
        # exp() of the rescaling comes from the device libm (glibc in the reference): a few ulp, tolerance 1e-12 relative
        assert np.allclose(pop.genotypeLikelihoods, np.array(c["gl"]), rtol=1e-12, atol=0)
        pop.call(100, 1)
        assert np.allclose(pop.frequencies, np.array(c["freqs"]), rtol=1e-10, atol=1e-300)
        assert [(-1 if g is None else genotypes.index(g)) for g in pop.genotypeCalls] == c["calls"]
        for k, v in enumerate(variants):
            assert pop.calculatePosterior(v) == c["posteriors"][k]
        if c["info"] is None:
            assert not pop.variantPosteriors
            continue
        assert pop.haplotypeScore == oracle.haplotype_score(c["hap_likes"])
        assert sorted(variants.index(v) for v in pop.vcfInfo) == sorted(int(k) for k in c["info"])
        for v, d in pop.vcfInfo.items():
            _info_equal(d, c["info"][str(variants.index(v))])
        assert {str(variants.index(v)): f for v, f in pop.vcfFilter.items()} == c["filter"]
        out = io.StringIO()
        V.outputCallToVCF(pop.varsByPos, pop.vcfInfo, pop.vcfFilter, pop.haplotypes, pop.genotypes, pop.frequencies, pop.genotypeLikelihoods,
                          pop.goodnessOfFitValues, pop.haplotypeIndexes, pop.readBuffers, pop.nIndividuals,
                          V.VCF([s["name"] for s in c["samples"]]), fasta, out, opts, pop.variants, c["start"], c["end"], population=pop)
        assert out.getvalue().split("\n")[:-1] == c["lines"]
        nlines += len(c["lines"])
    assert nlines > 50


@pytest.mark.gpu
def test_every_kernel_of_a_call_is_timed_and_the_poll_interval_is_checked():
    """Round 6: plat_kernel_times brackets every launch of the calls made while the profile is on (bench.py's roofline kernel is chosen from these
    sums); plat_sync_poll_us takes an interval and refuses a negative one; a wait with a long interval still returns the batch's results."""
    import numpy as np
    from platypus_amd import synth, _lib
    from platypus_amd.engine import Engine
    eng = Engine(0)
    hb = synth.config2(300, seed=77)
    db = eng.upload(hb)
    eng.call_windows(db, want_stats=False)
    eng.synchronize()
    want = db.score.cpu().numpy()[:hb.n_pairs].copy()
    eng.profile_enable(True)
    for _ in range(3):
        eng.call_windows(db, want_stats=False, asynchronous=True)
    eng.synchronize()
    t = eng.kernel_times()
    eng.profile_enable(False)
    for k in ("k_prep_reads", "k_sweep", "k_pairs", "k_seed_slow", "k_dp_jobs", "k_finalize", "k_genotype"):
        assert t[k][1] == 3 and 0.0 < t[k][0] < 50.0, (k, t.get(k))
    assert eng.kernel_times() == {}                                             # resolved pairs are handed out once
    # ONE kernel timed while the profile is off (how bench.py times its roofline kernel inside the timed steps)
    assert eng.lib.plat_kernel_timer_only(eng.ctx, 20) == 0                     # PLAT_KT_DP_JOBS
    for _ in range(2):
        eng.call_windows(db, want_stats=False, asynchronous=True)
    eng.synchronize()
    t1 = eng.kernel_times()
    assert list(t1) == ["k_dp_jobs"] and t1["k_dp_jobs"][1] == 2 and t1["k_dp_jobs"][0] > 0
    assert eng.lib.plat_kernel_timer_only(eng.ctx, -1) == 0 and eng.lib.plat_kernel_timer_only(eng.ctx, 32) == -1
    eng.call_windows(db, want_stats=False, asynchronous=True)
    eng.synchronize()
    assert eng.kernel_times() == {}
    assert eng.lib.plat_sync_poll_us(eng.ctx, -1) == -1 and eng.lib.plat_sync_poll_us(eng.ctx, 2000) == 0
    eng.call_windows(db, want_stats=False, asynchronous=True)
    eng.synchronize()
    assert np.array_equal(db.score.cpu().numpy()[:hb.n_pairs], want)
    assert eng.lib.plat_sync_poll_us(eng.ctx, 40) == 0


def test_the_scan_on_two_bit_codes_gives_the_records_of_the_byte_scan(golden_dir):
    """Round 6: plat_candidates_batch_codes (32 bases per word on 2-bit codes, bytes looked at only where codes differ) = plat_candidates_batch, record
    for record, on the reference's golden reads and on generated regions with what could tell them apart: N in reads and reference (N shares G's
    code), low qualities, soft clips, insertions and deletions, reads hanging over the window's ends, and reference regions with other bytes
    (lowercase, IUPAC), which plat_ref_codes must flag so that they take the byte scan."""
    import gzip, json, os
    import numpy as np
    eng = H.get_engine()
    cases = json.load(gzip.open(os.path.join(golden_dir, "candidate_cases.json.gz"), "rt"))
    regions = []
    for c in cases:
        reads = [dict(seq=r["seq"].encode(), qual=bytes(r["qual"]), pos=r["pos"], flag=r["flag"], cigar=[tuple(x) for x in r["cigar"]]) for r in c["reads"]]
        if all(set(r["seq"]) <= set(b"ACGTN") for r in reads):
            regions.append(dict(ref=c["ref"].encode(), ref_seq_start=0, contig_len=len(c["ref"]), reads=reads))
    assert len(regions) > 20
    rng = np.random.default_rng(606)
    B = np.frombuffer(b"ACGT", dtype=np.uint8)
    for k in range(60):
        L = int(rng.integers(400, 1500))
        ref = B[rng.integers(0, 4, L)].copy()
        if k % 3 == 0:
            ref[rng.integers(0, L, 5)] = ord("N")
        if k % 5 == 1:
            ref[rng.integers(0, L, 3)] = np.frombuffer(b"aRy", dtype=np.uint8)          # an irregular region: bytes the codes cannot tell from A / C / T
        reads = []
        for _ in range(int(rng.integers(5, 60))):
            n = int(rng.integers(30, 151))
            p = int(rng.integers(0, L - n - 20))
            seq = ref[p:p + n].copy()
            seq[seq > 90] = ord("A")                                                    # (reads hold A, C, G, T, N only: the caller's promise)
            for _m in range(int(rng.integers(0, 6))):
                seq[int(rng.integers(0, n))] = B[int(rng.integers(0, 4))]
            if rng.random() < 0.3:
                seq[int(rng.integers(0, n))] = ord("N")
            qual = rng.integers(0, 41, n).astype(np.uint8)
            cig, t = [(0, n)], rng.random()
            if t < 0.2 and n > 60:
                a = int(rng.integers(15, n - 30)); d = int(rng.integers(1, 8))
                cig = [(0, a), (2, d), (0, n - a)]                                      # a deletion: the read's tail is compared d bases further on
            elif t < 0.4 and n > 60:
                a = int(rng.integers(15, n - 30)); i = int(rng.integers(1, 8))
                cig = [(0, a), (1, i), (0, n - a - i)]
            elif t < 0.5:
                sc = int(rng.integers(1, 12))
                cig = [(4, sc), (0, n - sc)]
            reads.append(dict(seq=bytes(seq), qual=bytes(qual), pos=p, flag=int(512 if rng.random() < 0.05 else 0), cigar=cig))
        regions.append(dict(ref=bytes(ref), ref_seq_start=0, contig_len=L, reads=reads))
    for lo in range(0, len(regions), 40):
        part = regions[lo:lo + 40]
        want = eng.candidates(part, 10, 20, 1, 1)
        got = eng.candidates(part, 10, 20, 1, 1, codes=True)
        assert got == want
        irr = eng.last_ref_irregular.cpu().numpy()[:len(part)]
        assert [int(x) for x in irr] == [int(any(ch not in b"ACGTN" for ch in g["ref"])) for g in part]
    assert sum(len(x) for x in want) > 0
