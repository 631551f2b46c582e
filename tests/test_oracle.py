"""CPU tests: pin the oracle (oracle/plat_oracle.c) to the reference.

 * against the committed golden vectors (generated from the reference's own code by
   tests/golden/gen_golden.py), and
 * directly against the unmodified reference align.c when oracle/_ref/libalign_ref.so is present.
"""
import gzip
import json
import math
import os

import numpy as np
import pytest

from oracle.oracle import RefAlign


def test_dp_golden_scores(oracle, golden_dir):
    g = np.load(os.path.join(golden_dir, "dp_cases.npz"))
    got = oracle.dp_batch(g["haps"], g["reads"], g["quals"], g["gos"], g["lens"])
    assert np.array_equal(got, g["score"])
    assert len(got) >= 2000


def test_dp_golden_traceback(oracle, golden_dir):
    g = np.load(os.path.join(golden_dir, "dp_cases.npz"))
    for n, j in enumerate(g["tb_idx"]):
        L = int(g["lens"][j])
        sc, a1, a2, fp = oracle.dp_align(g["haps"][j, :L + 15].tobytes(), g["reads"][j, :L].tobytes(),
                                         g["quals"][j, :L].tobytes(), g["gos"][j, :L + 15].tobytes())
        assert sc == g["score"][j]
        assert a1.decode() == str(g["tb_aln1"][n]) and a2.decode() == str(g["tb_aln2"][n])
        assert fp == g["tb_firstpos"][n]


def test_dp_known_answers(oracle):
    # exact match -> 0, one Q30 mismatch -> 30, 2-bp del -> 43, 2-bp ins -> 47, N in hap -> 0
    rng = np.random.default_rng(1)
    hap = bytes(rng.choice(list(b"ACGT"), 115).tolist())
    go = bytes([40] * 115)
    q = bytes([30] * 100)
    read = hap[8:108]
    assert oracle.dp_score(hap, read, q, go) == 0
    r2 = bytearray(read); r2[50] = ord("A") if r2[50] != ord("A") else ord("C")
    assert oracle.dp_score(hap, bytes(r2), q, go) == 30
    h2 = bytearray(hap); h2[58] = ord("N")
    assert oracle.dp_score(bytes(h2), bytes(r2), q, go) == 0


@pytest.mark.skipif(not RefAlign.available(), reason="oracle/_ref not built (reference tree absent)")
def test_dp_fuzz_against_reference_build(oracle):
    ref = RefAlign()
    rng = np.random.default_rng(99)
    B = b"ACGT"
    for it in range(1500):
        L = int(rng.choice([7, 9, 25, 64, 101, 150, 250]))
        hap = bytearray(rng.choice(list(B + b"N"), L + 15, p=[.24, .24, .24, .24, .04]).tolist())
        off = int(rng.integers(0, 16))
        read = bytearray(hap[off:off + L])
        for _ in range(int(rng.integers(0, 6))):
            read[int(rng.integers(0, L))] = B[int(rng.integers(0, 4))]
        if L > 30 and rng.random() < 0.5:
            p = int(rng.integers(5, L - 8)); k = int(rng.integers(1, 7))
            read = (read[:p] + read[p + k:] + bytearray(rng.choice(list(B), k).tolist()))[:L]
        q = bytes(rng.integers(0, 94, L).astype(np.uint8).tolist())
        go = bytes(rng.integers(1, 46, L + 15).astype(np.uint8).tolist())
        a = oracle.dp_align(bytes(hap), bytes(read), q, go)
        b = ref.dp_align(bytes(hap), bytes(read), q, go)
        assert a == b
        assert oracle.dp_score(bytes(hap), bytes(read), q, go) == ref.dp_score(bytes(hap), bytes(read), q, go) == a[0]
        if a[0] > 0:
            fa = oracle.flank_score(L + 15, 20, q, go, a[3], a[1], a[2])
            fb = ref.flank_score(L + 15, 20, q, go, b[3], b[1], b[2])
            assert fa == fb


def test_mapalign_golden(oracle, golden_dir):
    cases = json.load(gzip.open(os.path.join(golden_dir, "mapalign_cases.json.gz"), "rt"))
    assert len(cases) >= 500
    for c in cases:
        sc, _ = oracle.align_read_to_hap(c["read"].encode(), bytes(c["qual"]), c["readStart"], c["hap"].encode(),
                                         c["hapStart"], c["flank"], c["doFlank"])
        assert sc == c["score"]


def test_assembler_golden(oracle, golden_dir):
    cases = json.load(gzip.open(os.path.join(golden_dir, "assembler_cases.json.gz"), "rt"))
    assert len(cases) >= 100
    nvar = 0
    for c in cases:
        got, nn = oracle.assemble(c["ref"].encode(), c["refStart"], c["assemStart"], c["assemEnd"],
                                  [s.encode() for s in c["seqs"]], [q.encode("latin1") for q in c["quals"]],
                                  c["k"], c["minQual"], c["minWeight"], c["noCycles"])
        exp = [(p, r.encode(), a.encode()) for p, r, a in c["variants"]]
        assert got == exp
        assert nn == c["nNodes"]
        nvar += len(exp)
    assert nvar > 100


def test_gap_open_table_matches_reference_formula(oracle):
    # chaplotype.pyx:64-67: homopolq = chr(int(33.5 + 10*log((idx+1)*q)/log(0.1))) ; gap-open = char - '!'
    errs = [2.9e-5, 2.9e-5, 2.9e-5, 2.9e-5, 4.3e-5, 1.1e-4, 2.4e-4, 5.7e-4, 1.0e-3, 1.4e-3] + \
           [1.4e-3 + 4.3e-4 * (n - 10) for n in range(11, 50)]
    table = [int(33.5 + 10 * math.log((i + 1) * q) / math.log(0.1)) - 33 for i, q in enumerate(errs)]
    assert len(table) == 49
    # a homopolymer run of 60 A's exposes the whole table (run length to the right, capped at 48)
    go = oracle.gap_open(b"C" + b"A" * 60 + b"G")
    assert go[-1] == 0
    assert list(go[1:61]) == [table[min(59 - i, 48)] for i in range(60)]
    assert go[0] == table[0] and go[61] == table[0]
    # 'N' resets the run (chaplotype.pyx:588-590)
    go = oracle.gap_open(b"ANNNA")
    assert list(go[:5]) == [table[0]] * 5


def test_loglik_known_answers(oracle):
    assert oracle.loglik(0, 60) == math.log(1.0 - 1e-6) or abs(oracle.loglik(0, 60) - math.log(1 - 1e-6)) < 1e-15
    assert oracle.loglik(0, 0) == -300.0
    assert oracle.loglik(10, 60) == -0.23025850929940459 * 10 + math.log(1.0 - math.exp(-0.23025850929940459 * 60))
    assert oracle.loglik(5000, 60) == -300.0


def test_genotype_known_answers(oracle):
    a = np.array([-1.0, -2.0, -0.5, 999.0])
    b = np.array([-5.0, -2.0005, -1.5, 999.0])
    L, gof, h1, h2 = oracle.genotype_loglik(a, a, True, 3)
    assert L == (-1.0 + -2.0) + -0.5
    L, gof, h1, h2 = oracle.genotype_loglik(a, b, False, 3)
    exp = (math.log(0.5) + -1.0) + (-2.0) + math.log(0.5 * (math.exp(-0.5) + math.exp(-1.5)))
    assert abs(L - exp) < 1e-14
    log10e = 0.43429448190325182
    assert abs(gof - (-10 * (log10e * -1.0 + log10e * -2.0 + log10e * -0.5)) / 3) < 1e-14


# ---- a11/a12 + SURVEY 8(f) rank 1, pinned by the reference's own method texts (tests/golden/gen_golden.py) ------------
def _population_cases(golden_dir):
    import gzip, json
    return json.load(gzip.open(os.path.join(golden_dir, "population_cases.json.gz"), "rt"))


def test_genotype_likelihoods_match_reference_golden(oracle, golden_dir):
    """calculateDataLikelihood (cgenotype.pyx:131-189) + the rescaling of Population.setup (cpopulation.pyx:283-309)."""
    n = 0
    for c in _population_cases(golden_dir):
        for i, ind in enumerate(c["individuals"]):
            ll = np.array(ind["loglik"], dtype=np.float64).reshape(c["n_hap"], -1)
            logl, gl, gof = oracle.population_setup_ind(ll, ind["n_reads"])
            assert np.array_equal(gl, np.array(c["gl"][i]))
            if ind["n_reads"] > 0:
                assert np.array_equal(logl, np.array(c["logl"][i]))
                assert np.array_equal(gof, np.array(c["gof"][i]))
            n += 1
    assert n > 500


def test_em_calls_and_posteriors_match_reference_golden(oracle, golden_dir):
    """EMiteration / call / callGenotypes / calculatePosterior (cpopulation.pyx:384-703): bit-identical doubles."""
    long_runs = 0
    for c in _population_cases(golden_dir):
        nr = [ind["n_reads"] for ind in c["individuals"]]
        gl = np.array(c["gl"])
        freq, em, calls, iters, mc = oracle.em_call(nr, gl, 100, c["use_em"])
        assert iters == c["iters"] and mc == c["max_change"]
        assert np.array_equal(freq, np.array(c["freqs"]))
        live = np.array(nr) > 0
        assert np.array_equal(em[live], np.array(c["em"])[live])
        assert calls.tolist() == c["calls"]
        member = np.array(c["member"])
        for k, prior in enumerate(c["priors"]):
            assert oracle.variant_posterior(nr, gl, freq, member[:, k], prior) == c["posterior"][k]
            assert oracle.variant_posterior(nr, gl, freq, member[:, k], 0.5) == c["posterior_flat"][k]
        long_runs += iters > 5
    assert long_runs >= 5


def test_genotype_marginalisation_matches_reference_golden(oracle, golden_dir):
    """computeGenotypeCallAndLikelihoods (vcfutils.pyx:163-334)."""
    n = 0
    for c in _population_cases(golden_dir):
        member = np.array(c["member"])
        for gc in c["genotype_calls"]:
            rows = member[:, gc["vset"]]
            isref = (rows.sum(axis=1) == 0).astype(np.int32)
            ph, lik, out4 = oracle.genotype_call(c["freqs"], c["gl"][gc["ind"]], c["gof"][gc["ind"]], rows, isref, gc["n_individuals"])
            assert ph.tolist() == gc["phased"]
            assert np.array_equal(lik, np.array(gc["likelihoods"]))
            exp4 = np.array([gc["genotype_posterior"], gc["nonref_posterior"], gc["ref_posterior"], gc["gof"]])
            assert np.array_equal(out4, exp4, equal_nan=True)
            n += 1
    assert n > 900


# ---- a7-a10 pinned by the reference's own chaplotype.pyx texts (tests/golden/gen_golden.py: gen_haplotype) -----------
def _haplotype_cases(golden_dir):
    import gzip, json
    return json.load(gzip.open(os.path.join(golden_dir, "haplotype_cases.json.gz"), "rt"))


def _reads_dict(case, all_broken=False):
    r = case["reads"]
    return dict(seq=[x["seq"].encode() for x in r], qual=[bytes(x["qual"]) for x in r], pos=[x["pos"] for x in r],
                end=[x["end"] for x in r], mapq=[x["mapq"] for x in r], flags=[x["flag"] for x in r],
                kind=[2 if all_broken else x["kind"] for x in r])


def test_likelihood_cache_matches_reference_golden(oracle, golden_dir):
    """Haplotype.alignReads / alignSingleRead / annotateWithGapOpen (chaplotype.pyx:306-384,552-676): same doubles, same bytes."""
    n = flank = 0
    for c in _haplotype_cases(golden_dir):
        haps = [h.encode() for h in c["haps"]]
        ll, sc, _ = oracle.align_window(haps, c["start"], c["end"], c["buf"], _reads_dict(c), do_flank=c["calc_flank"])
        single, _, _ = oracle.align_window(haps, c["start"], c["end"], c["buf"], _reads_dict(c, True), do_flank=c["calc_flank"])
        for hi, h in enumerate(haps):
            assert c["cache"][hi][-1] == 999                                  # terminator, chaplotype.pyx:375
            assert np.array_equal(ll[hi], np.array(c["cache"][hi][:-1]))
            assert np.array_equal(single[hi], np.array(c["single"][hi]))
            assert list(oracle.gap_open(h)) == c["gapopen"][hi]
            n += len(c["single"][hi])
        flank += c["calc_flank"]
    assert n > 3000 and flank > 10


# ---- VariantCandidateGenerator, pinned by the reference's own text --------------------------------
def merge_candidates(records):
    """addVariantToList (variant.pyx:499-527): equal variants merge, supporting reads add up; dict order = first seen."""
    heap = {}
    for pos, rem, add, _ in records:
        heap[(pos, rem, add)] = heap.get((pos, rem, add), 0) + 1
    return [[p, r.decode(), a.decode(), c] for (p, r, a), c in heap.items()]


def variant_sort_key(v):
    """Variant.__richcmp__ ordering (variant.pyx:282-363): refPos, varType, nRemoved (refName is constant here)."""
    pos, rem, add, _ = v
    if len(rem) == len(add):
        vt = 0 if len(add) == 1 else 1
    elif len(rem) == 0:
        vt = 2
    elif len(add) == 0:
        vt = 3
    else:
        vt = 4
    return (pos, vt, len(rem))


def test_variant_candidates_match_reference_golden(oracle, golden_dir):
    import gzip, json
    cases = json.load(gzip.open(os.path.join(golden_dir, "candidate_cases.json.gz"), "rt"))
    n = 0
    for c in cases:
        ref = c["ref"].encode()
        rs = max(0, c["start"] - 2000)
        re_ = min(c["end"] + 2000, len(ref) - 1)                               # variant.pyx:486-488
        reads = [dict(seq=r["seq"].encode(), qual=bytes(r["qual"]), pos=r["pos"], flag=r["flag"], cigar=r["cigar"]) for r in c["reads"]]
        recs = oracle.variant_candidates(ref[rs:re_], rs, len(ref), reads, c["min_flank"], c["min_base_qual"], c["gen_snps"], c["gen_indels"])
        first_seen = merge_candidates(recs)
        assert first_seen == c["first_seen"]
        assert sorted(first_seen, key=variant_sort_key) == c["sorted"]          # sorted() is stable: ties keep first-seen order
        n += len(first_seen)
    assert n > 5000


# ---- read QC / trimming (checkAndTrimRead), pinned by the reference's own text ---------------------------------------------
def test_check_and_trim_matches_reference_golden(oracle, golden_dir):
    import gzip, json
    cases = json.load(gzip.open(os.path.join(golden_dir, "readqc_cases.json.gz"), "rt"))
    trimmed = 0
    for c in cases:
        ok, flags, quals, reason = oracle.check_and_trim(c["reads"], c["options"])
        assert ok.tolist() == c["ok"] and flags.tolist() == c["flag_out"]
        for r, q, exp in zip(c["reads"], quals, c["qual_out"]):
            assert q == (r["qual"] if exp is None else exp)
            trimmed += exp is not None
        counts = [int((reason == k).sum()) for k in range(7)]
        for k in range(7):
            if c["counts"][k] != -1:
                assert counts[k] == c["counts"][k]
    assert trimmed > 500


# ---- read statistics of the VCF INFO field, pinned by the reference's own leaf functions ------------------------------------
def infostats_inputs(c):
    variants = [dict(pos=v["pos"], removed=v["removed"].encode(), added=v["added"].encode(), bam_min=v["pos"], bam_max=v["pos"]) for v in c["variants"]]
    conv = lambda r: dict(seq=r["seq"].encode(), qual=bytes(r["qual"]), pos=r["pos"], end=r["end"], mapq=r["mapq"], flag=r["flag"], cigar=r["cigar"])
    samples = [dict(good=[conv(r) for r in s_["good"]], bad=[conv(r) for r in s_["bad"]]) for s_ in c["samples"]]
    return variants, samples


def test_variant_read_stats_match_reference_golden(oracle, golden_dir):
    import gzip, json
    cases = json.load(gzip.open(os.path.join(golden_dir, "infostats_cases.json.gz"), "rt"))
    nsup = 0
    for c in cases:
        variants, samples = infostats_inputs(c)
        res = oracle.variant_read_stats(variants, samples, c["var_in_genotype"], 20, c["bad_reads_window"], c["exact"])
        for (counts, nr, nvr, mq), exp in zip(res, c["results"]):
            assert counts == exp["counts"] and nr == exp["n_reads"] and nvr == exp["n_var_reads"] and mq == exp["min_quals"]
            nsup += counts[2]
    assert nsup > 400


def test_tandem_annotation_matches_unmodified_reference_build():
    """platypus_amd.indelprior.annotate against oracle/_ref/libtandem_ref.so = the UNMODIFIED src/c/tandem.c (built by
    oracle/Makefile from where it lies in the reference tree; the prebuilt file travels to the GPU box)."""
    import ctypes as C
    import os
    import numpy as np
    import pytest
    from platypus_amd.indelprior import annotate
    so = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libtandem_ref.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref/libtandem_ref.so not built (reference tree absent)")
    lib = C.CDLL(so)
    rng = np.random.default_rng(11)
    for it in range(400):
        n = int(rng.integers(1, 300))
        seq = bytearray(bytes(rng.choice(list(b"ACGT" if it % 3 else b"ACGTNacgtn"), n).astype(np.uint8)))
        for _ in range(int(rng.integers(0, 5))):
            p = int(rng.integers(0, n)); u = bytes(rng.choice(list(b"ACGT"), int(rng.integers(1, 13))).astype(np.uint8)); k = int(rng.integers(2, 90))
            seq[p:p + k] = (u * k)[:k]
        seq = bytes(seq[:n])
        for full in (True, False):
            s, d = C.create_string_buffer(n + 1), C.create_string_buffer(n + 1)
            lib.annotate(seq, s, d, -n if full else n)
            got = annotate(seq, full)
            assert (list(got[0]), list(got[1])) == (list(s.raw[:n]), list(d.raw[:n])), (seq, full)
