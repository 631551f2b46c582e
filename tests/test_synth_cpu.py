"""CPU tests of the host logic: synthetic workloads (BASELINE configs) and HostBatch bookkeeping."""
import numpy as np

from platypus_amd import synth


def test_config1_shape():
    b = synth.config1()
    assert b.n_windows == 1 and b.n_haps == 4 and b.n_reads == 64 and b.n_pairs == 256
    haps = b.window_haps(0)
    assert all(len(h) == 1400 for h in haps)                      # W=1000 + 2*buf(200)
    assert sum(a != c for a, c in zip(haps[0], haps[1])) == 1     # single-SNP haplotypes


def test_config2_is_deterministic_and_consistent():
    a = synth.config2(64)
    b = synth.config2(64)
    assert np.array_equal(a.read_seq, b.read_seq) and np.array_equal(a.hap_seq, b.hap_seq)
    H = np.diff(a.win_hap_begin)
    assert set(H.tolist()) <= {2, 4, 8}
    assert np.all(a.win_flank == 300)
    assert np.array_equal(np.diff(a.hap_off), np.repeat(a.win_end - a.win_start + 600, H))
    # reads sorted: good by pos then bad by pos inside every window; QCFail == bad
    for w in range(a.n_windows):
        r = a.window_reads(w)
        k = np.asarray(r["kind"])
        assert np.all(np.diff(k.astype(int)) >= 0)
        for kk in (0, 1):
            assert np.all(np.diff(np.asarray(r["pos"])[k == kk]) >= 0)
        assert np.array_equal((np.asarray(r["flags"]) & 512) != 0, k == 1)
    assert a.seg_n_good.sum() == int((a.read_kind == 0).sum())
    assert a.pair_off[-1] == int((H * np.diff(a.win_read_begin)).sum())


def test_subset_roundtrip():
    a = synth.config2(32)
    s = a.subset([3, 7, 11])
    assert s.n_windows == 3
    assert s.window_haps(1) == a.window_haps(7)
    assert s.window_reads(2)["seq"] == a.window_reads(11)["seq"]
    assert np.array_equal(s.window_reads(0)["pos"], a.window_reads(3)["pos"])


def test_population_mode_segments():
    b = synth.config5(8, 5)
    assert b.n_ind == 5 and len(b.seg_n_good) == 40
    assert b.seg_read_begin[-1] == b.n_reads
    assert np.array_equal(b.seg_read_begin[::5], b.win_read_begin)


def test_config4_region_is_procedural_and_consistent():
    """Region i of config 4 depends on (seed, i) only; reads carry the CIGAR an aligner would give them and reproduce the
    reference outside their planted variants."""
    a = synth.config4_region(3, region_len=4000, n_samples=2, indel_rate=2e-3)
    b = synth.config4_region(3, region_len=4000, n_samples=2, indel_rate=2e-3)
    c = synth.config4_region(4, region_len=4000, n_samples=2, indel_rate=2e-3)
    assert a["ref"] == b["ref"] and a["variants"] == b["variants"] and a["samples"][1][7] == b["samples"][1][7]
    assert a["ref"] != c["ref"] and a["chrom"] == "r3" and c["chrom"] == "r4"
    assert len(a["truth"]) == 2 and all(len(t) == len(a["variants"]) and set(t) <= {0, 1, 2} for t in a["truth"])
    assert any(len(rem) != len(add) for _, rem, add in a["variants"])
    for r in a["samples"][0][:200]:
        span = sum(n for op, n in r["cigar"] if op in (0, 2))
        assert r["end"] - r["pos"] == span and sum(n for op, n in r["cigar"] if op in (0, 1)) == len(r["seq"]) == len(r["qual"])
        if r["cigar"] == [(0, len(r["seq"]))]:
            mism = sum(x != y for x, y in zip(r["seq"], a["ref"][r["pos"]:r["pos"] + len(r["seq"])]))
            assert mism <= 6                                        # planted SNPs + 0.1 % errors
