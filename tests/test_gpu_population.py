"""GPU parity of SURVEY 8(f) rank 1 (EM, genotype calls, variant posteriors, per-site genotype marginalisation) through
the C ABI, against golden vectors produced by the reference's own method texts (tests/golden/gen_golden.py) and against
the oracle on the likelihoods the device itself computed."""
import collections
import gzip
import json
import os

import numpy as np
import pytest

from platypus_amd import synth
from platypus_amd.engine import Engine

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = Engine(0)
    yield e
    e.close()


def _cases(golden_dir):
    return json.load(gzip.open(os.path.join(golden_dir, "population_cases.json.gz"), "rt"))


def _groups(cases):
    g = collections.defaultdict(list)
    for c in cases:
        g[(c["n_ind"], c["use_em"])].append(c)
    return g


def test_em_and_calls_bit_identical_to_reference_golden(eng, golden_dir):
    """Frequencies, EM likelihoods, genotype calls and iteration counts: only + * / in the reference's order -> equal doubles."""
    for (n_ind, use_em), cs in _groups(_cases(golden_dir)).items():
        lb = eng.upload_likelihoods(n_ind, [c["n_hap"] for c in cs], [[i["n_reads"] for i in c["individuals"]] for c in cs],
                                    [c["gl"] for c in cs], [np.array(c["gof"]).T for c in cs])
        eng.em(lb, 100, use_em)
        freq, em = lb.freq.cpu().numpy(), lb.em.cpu().numpy()
        calls, iters = lb.calls.cpu().numpy().reshape(len(cs), n_ind), lb.em_iters.cpu().numpy()
        for w, c in enumerate(cs):
            h0, h1 = lb.host.win_hap_begin[w], lb.host.win_hap_begin[w + 1]
            assert iters[w] == c["iters"]
            assert np.array_equal(freq[h0:h1], np.array(c["freqs"]))
            G = c["n_hap"] * (c["n_hap"] + 1) // 2
            got = em[lb.host.gl_off[w]:lb.host.gl_off[w] + n_ind * G].reshape(n_ind, G)
            live = np.array([i["n_reads"] for i in c["individuals"]]) > 0
            assert np.array_equal(got[live], np.array(c["em"])[live]) and not got[~live].any()
            assert calls[w].tolist() == c["calls"]


def test_variant_posteriors_match_reference_golden(eng, golden_dir):
    """calculatePosterior: log/exp/log10 come from the device libm, the result is a rounded phred value -> equal."""
    n = 0
    for (n_ind, use_em), cs in _groups(_cases(golden_dir)).items():
        lb = eng.upload_likelihoods(n_ind, [c["n_hap"] for c in cs], [[i["n_reads"] for i in c["individuals"]] for c in cs],
                                    [c["gl"] for c in cs])
        eng.em(lb, 100, use_em)
        vw, masks, priors, exp = [], [], [], []
        for w, c in enumerate(cs):
            member = np.array(c["member"])
            for k, p in enumerate(c["priors"]):
                vw += [w, w]; masks += [member[:, k], member[:, k]]; priors += [p, 0.5]
                exp += [c["posterior"][k], c["posterior_flat"][k]]
        got = eng.variant_posteriors(lb, vw, masks, priors)
        assert np.array_equal(got, np.array(exp))
        n += len(exp)
    assert n > 200


def test_genotype_marginalisation_bit_identical_to_reference_golden(eng, golden_dir):
    n = 0
    for (n_ind, use_em), cs in _groups(_cases(golden_dir)).items():
        lb = eng.upload_likelihoods(n_ind, [c["n_hap"] for c in cs], [[i["n_reads"] for i in c["individuals"]] for c in cs],
                                    [c["gl"] for c in cs], [np.array(c["gof"]).T for c in cs])
        eng.em(lb, 100, use_em)
        sites, expect = [], []
        for w, c in enumerate(cs):
            member = np.array(c["member"])
            seen = set()
            for gc in c["genotype_calls"]:
                if gc["n_individuals"] != n_ind:
                    continue                    # (those records exercise the > 25 rule on the oracle only)
                key = tuple(gc["vset"])
                if key not in seen:
                    seen.add(key)
                    rows = member[:, gc["vset"]]
                    sites.append(dict(window=w, var_in_hap=rows, is_ref=(rows.sum(axis=1) == 0).astype(np.int32)))
                expect.append((len(sites) - 1, gc))
        res = eng.genotype_calls(lb, sites)
        for s, gc in expect:
            ph, lik, out4 = res[s]
            i = gc["ind"]
            assert ph[i].tolist() == gc["phased"]
            assert np.array_equal(lik[i], np.array(gc["likelihoods"]))
            exp4 = np.array([gc["genotype_posterior"], gc["nonref_posterior"], gc["ref_posterior"], gc["gof"]])
            assert np.array_equal(out4[i], exp4, equal_nan=True)
            n += 1
    assert n > 400


def test_population_path_end_to_end_vs_oracle(eng, oracle):
    """align -> genotype likelihoods -> EM on the device (population config: 12 samples), EM checked against the oracle
    run on the device's own likelihoods."""
    hb = synth.config5(6, 12)
    db = eng.upload(hb)
    eng.call_windows(db, want_stats=False)
    eng.em(db, 100, 0)
    gl = db.gl.cpu().numpy()
    freq, em = db.freq.cpu().numpy(), db.em.cpu().numpy()
    calls, iters = db.calls.cpu().numpy().reshape(hb.n_windows, hb.n_ind), db.em_iters.cpu().numpy()
    for w in range(hb.n_windows):
        H = hb.win_hap_begin[w + 1] - hb.win_hap_begin[w]
        G = H * (H + 1) // 2
        nr = hb.seg_n_good[w * hb.n_ind:(w + 1) * hb.n_ind]
        rows = gl[hb.gl_off[w]:hb.gl_off[w] + hb.n_ind * G].reshape(hb.n_ind, G)
        f, e, c, it, mc = oracle.em_call(nr, rows, 100, 0)
        assert it == iters[w] and np.array_equal(f, freq[hb.win_hap_begin[w]:hb.win_hap_begin[w + 1]])
        assert c.tolist() == calls[w].tolist()
        live = np.asarray(nr) > 0
        assert np.array_equal(e[live], em[hb.gl_off[w]:hb.gl_off[w] + hb.n_ind * G].reshape(hb.n_ind, G)[live])


def test_em_iterates_on_weak_evidence_and_matches_the_oracle(eng, oracle):
    """100 samples at 1x with mostly low-quality bases: the EM (cpopulation.pyx:384-457) needs tens of iterations instead of the two of the 30x workloads; frequencies,
    EM likelihoods, calls and iteration counts of k_em_wide against the oracle's EM on the device's own genotype likelihoods."""
    hb = synth.config5_weak_evidence(40, 100)
    db = eng.upload(hb)
    eng.call_windows(db, want_stats=False)
    eng.em(db, 100, 0)
    eng.synchronize()
    gl = db.gl.cpu().numpy()
    freq, em = db.freq.cpu().numpy(), db.em.cpu().numpy()
    calls, iters = db.calls.cpu().numpy().reshape(hb.n_windows, hb.n_ind), db.em_iters.cpu().numpy()
    assert iters.mean() >= 10 and iters.max() >= 20, (iters.mean(), iters.max())
    for w in range(hb.n_windows):
        H = hb.win_hap_begin[w + 1] - hb.win_hap_begin[w]
        G = H * (H + 1) // 2
        nr = hb.seg_n_good[w * hb.n_ind:(w + 1) * hb.n_ind]
        rows = gl[hb.gl_off[w]:hb.gl_off[w] + hb.n_ind * G].reshape(hb.n_ind, G)
        f, e, c, it, mc = oracle.em_call(nr, rows, 100, 0)
        assert it == iters[w] and np.array_equal(f, freq[hb.win_hap_begin[w]:hb.win_hap_begin[w + 1]])
        assert c.tolist() == calls[w].tolist()
        live = np.asarray(nr) > 0
        assert np.array_equal(e[live], em[hb.gl_off[w]:hb.gl_off[w] + hb.n_ind * G].reshape(hb.n_ind, G)[live])


def test_em_wide_kernel_equals_the_one_wave_kernel(eng, monkeypatch):
    """k_em_wide (likelihoods, responsibilities and the M-step's term streams in LDS) against k_em (one wave per window; PLAT_EM_NARROW=1)
    on the same likelihoods: frequencies, EM likelihoods, calls and iteration counts bit for bit -- with and without useEMLikelihoods,
    at 1, 7 and 100 samples, with samples that have no reads, and with likelihood rows replaced by random ones (the synthetic windows
    converge in two iterations: random rows make the EM run longer)."""
    rng = np.random.default_rng(77)
    for hb in (synth.config2(300, seed=5), synth.config5(12, 7), synth.config5(30, 100)):
        db = eng.upload(hb)
        eng.call_windows(db, want_stats=False)
        for flavour in range(2):
            if flavour == 1:                                                    # random likelihoods in (0, 1], some rows of a sample all tiny
                g = rng.random(db.gl.numel()) ** 4 + 1e-300
                g[rng.random(g.size) < 0.02] = 1e-300
                db.gl.copy_(__import__("torch").from_numpy(g).to(db.gl.device))
            for use_em in (0, 1):
                res = []
                for narrow in (False, True):
                    if narrow:
                        monkeypatch.setenv("PLAT_EM_NARROW", "1")
                    else:
                        monkeypatch.delenv("PLAT_EM_NARROW", raising=False)
                    eng.em(db, 100, use_em)
                    eng.synchronize()
                    res.append((db.freq.cpu().numpy().copy(), db.em.cpu().numpy().copy(), db.calls.cpu().numpy().copy(), db.em_iters.cpu().numpy().copy()))
                monkeypatch.delenv("PLAT_EM_NARROW", raising=False)
                for a, b in zip(res[0], res[1]):
                    assert np.array_equal(a, b, equal_nan=True)
                if flavour == 1:
                    assert res[0][3][:hb.n_windows].max() > 2


def test_haplotype_scores_vs_oracle(eng, oracle):
    """plat_haplotype_score_batch on multi-window batches (1 and 12 samples, some samples without reads): per-haplotype sums =
    the oracle's hap1Like on the device's own likelihoods (same doubles), HapScore = the oracle's clustering of them."""
    for hb in (synth.config2(300, seed=77), synth.config5(6, 12)):
        hb.seg_n_good = np.array(hb.seg_n_good).copy()
        if hb.n_ind > 1:
            hb.seg_n_good[hb.n_ind - 1::hb.n_ind] = 0          # the last sample has no good reads: the one before it counts
        db = eng.upload(hb)
        eng.align(db, want_stats=False)
        like, score = eng.haplotype_scores(db)
        ll = db.loglik.cpu().numpy()
        for w in range(hb.n_windows):
            h0, h1 = hb.win_hap_begin[w], hb.win_hap_begin[w + 1]
            R = hb.win_read_begin[w + 1] - hb.win_read_begin[w]
            rows = ll[hb.pair_off[w]:hb.pair_off[w] + (h1 - h0) * R].reshape(h1 - h0, R)
            ind = max(i for i in range(hb.n_ind) if hb.seg_n_good[w * hb.n_ind + i] != 0)
            s0 = hb.seg_read_begin[w * hb.n_ind + ind] - hb.win_read_begin[w]
            s1 = hb.seg_read_begin[w * hb.n_ind + ind + 1] - hb.win_read_begin[w]
            exp = []
            for h in range(h1 - h0):
                arr = np.concatenate([rows[h, s0:s1], [999.0]])
                exp.append(oracle.genotype_loglik(arr, arr, True, 1)[2])
            assert like[h0:h1].tolist() == exp
            assert score[w] == oracle.haplotype_score(exp)


def test_variant_read_stats_match_reference_golden(eng, golden_dir):
    """SURVEY 8(f) rank 3 (INFO read statistics): coverage / support / strand counts, per-sample counts and the MMLQ window minima
    per variant, against the reference's own leaf functions driven as vcfINFO drives them."""
    cases = json.load(gzip.open(os.path.join(golden_dir, "infostats_cases.json.gz"), "rt"))
    conv = lambda r: dict(seq=r["seq"].encode(), qual=bytes(r["qual"]), pos=r["pos"], end=r["end"], mapq=r["mapq"], flag=r["flag"], cigar=r["cigar"])
    groups = collections.defaultdict(list)
    for c in cases:
        groups[(len(c["samples"]), c["bad_reads_window"], c["exact"])].append(c)
    nsup = 0
    for (nI, brw, exact), cs in groups.items():
        wins = [dict(variants=[dict(pos=v["pos"], removed=v["removed"].encode(), added=v["added"].encode(), bam_min=v["pos"], bam_max=v["pos"])
                               for v in c["variants"]],
                     samples=[dict(good=[conv(r) for r in s_["good"]], bad=[conv(r) for r in s_["bad"]]) for s_ in c["samples"]],
                     var_in_genotype=c["var_in_genotype"]) for c in cs]
        res = eng.variant_read_stats(wins, brw, exact)
        for c, rw in zip(cs, res):
            for (counts, nr, nvr, mq), exp in zip(rw, c["results"]):
                assert counts == exp["counts"] and nr == exp["n_reads"] and nvr == exp["n_var_reads"] and mq == exp["min_quals"]
                nsup += counts[2]
    assert nsup > 400
