"""GPU parity tests (run with -m gpu on the MI355X box): the HIP path, called through the C ABI,
against the oracle on the same seeded inputs, against the committed golden vectors, and -- at
BASELINE.json's full size -- through size-independent properties.

Bar: bit-exact integer scores; log-likelihoods bit-identical except where the device libm is involved
(tolerance written at the assert)."""
import gzip
import json
import os

import numpy as np
import pytest

from platypus_amd import _lib, synth
from platypus_amd.engine import Engine
from platypus_amd.batch import HostBatch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from platypus_amd.engine import Engine
    return Engine(0)


def oracle_windows(oracle, hb, windows=None, do_flank=0):
    """Oracle results for a HostBatch: per-window (loglik [H,R], score [H,R], n_dp)."""
    out = []
    for w in (range(hb.n_windows) if windows is None else windows):
        out.append(oracle.align_window(hb.window_haps(w), int(hb.win_start[w]), int(hb.win_end[w]),
                                       int(hb.win_flank[w]), hb.window_reads(w), do_flank=do_flank))
    return out


def run_align(eng, hb, do_flank=0):
    db = eng.upload(hb)
    st = eng.align(db, calc_flank_score=do_flank)
    eng.synchronize()
    return db, st, db.loglik.cpu().numpy()[:hb.n_pairs], db.score.cpu().numpy()[:hb.n_pairs]


def check_against_oracle(oracle, hb, ll, sc, st=None, do_flank=0):
    ndp = 0
    for w, (oll, osc, n) in enumerate(oracle_windows(oracle, hb, do_flank=do_flank)):
        a, b = hb.pair_off[w], hb.pair_off[w + 1]
        assert np.array_equal(sc[a:b].reshape(osc.shape), osc), "scores differ in window %d" % w
        # score -> log-likelihood is one fp64 multiply + one add of a host-computed table entry: bit-identical
        assert np.array_equal(ll[a:b].reshape(oll.shape), oll), "log-likelihoods differ in window %d" % w
        ndp += n
    if st is not None:
        assert st.n_dp_reference == ndp
        assert st.n_pairs == hb.n_pairs


# ---- a1 ------------------------------------------------------------------------------------------
def test_dp_golden_vectors_bit_exact(eng, golden_dir):
    g = np.load(os.path.join(golden_dir, "dp_cases.npz"))
    got = eng.dp_batch(g["haps"], g["reads"], g["quals"], g["gos"], g["lens"])
    assert np.array_equal(got, g["score"])


def test_dp_fuzz_vs_oracle_all_lengths(eng, oracle):
    rng = np.random.default_rng(4242)
    n, lmax = 4096, 300
    lens = rng.integers(7, lmax + 1, n).astype(np.int32)
    lens[:64] = np.arange(7, 71)
    haps = rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), (n, lmax + 15), p=[.24, .24, .24, .24, .04])
    reads = np.empty((n, lmax), dtype=np.uint8)
    for j in range(n):
        off = int(rng.integers(0, 16))
        seg = haps[j, off:off + lmax]
        reads[j, :len(seg)] = seg
        reads[j, len(seg):] = ord("A")
    mut = rng.random((n, lmax)) < 0.03
    reads[mut] = rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), int(mut.sum()))
    for j in range(0, n, 3):          # indels
        L = int(lens[j]); p = int(rng.integers(1, max(2, L - 2))); k = int(rng.integers(1, 9))
        reads[j, p:lmax - k] = reads[j, p + k:lmax].copy()
    quals = rng.integers(0, 94, (n, lmax)).astype(np.uint8)
    quals[rng.random((n, lmax)) < 0.1] = 0
    gos = rng.integers(1, 46, (n, lmax + 15)).astype(np.uint8)
    got = eng.dp_batch(haps, reads, quals, gos, lens)
    exp = oracle.dp_batch(haps, reads, quals, gos, lens)
    assert np.array_equal(got, exp)


def test_dp_nonstandard_gap_parameters(eng, oracle):
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "dp_cases.npz"))
    sl = slice(0, 400)
    for ge, npri in ((1, 0), (5, 3), (3, 2)):
        got = eng.dp_batch(g["haps"][sl], g["reads"][sl], g["quals"][sl], g["gos"][sl], g["lens"][sl], ge, npri)
        exp = oracle.dp_batch(g["haps"][sl], g["reads"][sl], g["quals"][sl], g["gos"][sl], g["lens"][sl], ge, npri)
        assert np.array_equal(got, exp)


# ---- a3..a10 -------------------------------------------------------------------------------------
def test_config1_window(eng, oracle):
    hb = synth.config1()
    db, st, ll, sc = run_align(eng, hb)
    check_against_oracle(oracle, hb, ll, sc, st)
    assert st.n_pairs_aligned == 256


def test_config2_subset_vs_oracle(eng, oracle):
    hb = synth.config2(300)
    db, st, ll, sc = run_align(eng, hb)
    check_against_oracle(oracle, hb, ll, sc, st)
    assert st.cells_reference == st.n_dp_reference * 16 * 150


def single_pair_batch(cases):
    """One window per golden map-and-align case: 1 haplotype, 1 read (kind=brokenMate: never skipped)."""
    nW = len(cases)
    haps = [c["hap"].encode() for c in cases]
    reads = [c["read"].encode() for c in cases]
    quals = [bytes(c["qual"]) for c in cases]
    hl = np.array([len(h) for h in haps]); rl = np.array([len(r) for r in reads])
    pos = np.array([c["readStart"] for c in cases], dtype=np.int32)
    flank = np.array([c["flank"] for c in cases], dtype=np.int32)
    ws = np.array([c["hapStart"] + c["flank"] for c in cases], dtype=np.int32)
    return HostBatch(
        n_ind=1, win_hap_begin=np.arange(nW + 1, dtype=np.int32), win_read_begin=np.arange(nW + 1, dtype=np.int32),
        win_start=ws, win_end=ws + 10, win_flank=flank,
        hap_seq=np.frombuffer(b"".join(haps), dtype=np.uint8), hap_off=np.concatenate([[0], np.cumsum(hl)]).astype(np.int64),
        read_seq=np.frombuffer(b"".join(reads), dtype=np.uint8), read_qual=np.frombuffer(b"".join(quals), dtype=np.uint8),
        read_off=np.concatenate([[0], np.cumsum(rl)]).astype(np.int64), read_pos=pos, read_end=pos + rl.astype(np.int32),
        read_mapq=np.full(nW, 60, dtype=np.uint8), read_flags=np.zeros(nW, dtype=np.int32),
        read_kind=np.full(nW, 2, dtype=np.uint8), seg_read_begin=np.arange(nW + 1, dtype=np.int32),
        seg_n_good=np.zeros(nW, dtype=np.int32))


@pytest.mark.parametrize("do_flank", [0, 1])
def test_mapalign_golden_vectors(eng, golden_dir, do_flank):
    """Golden vectors from the reference's own calign.pyx: tandem repeats (hundreds of arg-max diagonals),
    N's, reads hanging off either haplotype end, wrong mapping hints; with and without
    doCalculateFlankScore (the latter exercises the device traceback + calculateFlankScore, a2)."""
    cases = [c for c in json.load(gzip.open(os.path.join(golden_dir, "mapalign_cases.json.gz"), "rt"))
             if c["doFlank"] == do_flank]
    assert len(cases) > (100 if do_flank else 300)
    hb = single_pair_batch(cases)
    db, st, ll, sc = run_align(eng, hb, do_flank)
    exp = np.array([c["score"] for c in cases], dtype=np.int32)
    assert np.array_equal(sc, exp)
    assert st.n_dp_launched >= len(cases)


def edge_batch():
    """Ragged / degenerate windows: reads of every length 5..260 (shorter than 7 -> score 0), QCFail and
    overlap<7 reads (-> 0.0), brokenMates far outside the window, N runs and tandem repeats in the
    haplotypes, a window with no reads, a window whose haplotypes are identical, mapq 0."""
    rng = np.random.default_rng(77)
    B = np.frombuffer(b"ACGT", dtype=np.uint8)
    wins = []
    for w in range(24):
        W = int(rng.integers(10, 120)); buf = 500
        ref = B[rng.integers(0, 4, W + 2 * buf + 400)]
        if w % 3 == 1:
            u = B[rng.integers(0, 4, int(rng.integers(1, 6)))]
            p = int(rng.integers(300, 500)); ref[p:p + 90] = np.resize(u, 90)
        if w % 4 == 2:
            ref[int(rng.integers(400, 600)):][:7] = ord("N")
        ws = 200 + buf
        haps = [ref[ws - buf:ws + W + buf].copy()]
        for k in range(int(rng.integers(0, 4))):
            h = haps[0].copy()
            t = int(rng.integers(0, 3)); p = buf + int(rng.integers(0, W))
            if t == 0:
                h[p] = B[(int(np.searchsorted(B, h[p])) + 1) % 4] if h[p] != ord("N") else ord("A")
            elif t == 1:
                h = np.concatenate([h[:p], B[rng.integers(0, 4, int(rng.integers(1, 20)))], h[p:]])
            else:
                h = np.concatenate([h[:p], h[p + int(rng.integers(1, 20)):]])
            haps.append(h)
        if w == 5:
            haps = [haps[0], haps[0].copy()]
        reads = []
        nR = 0 if w == 7 else int(rng.integers(1, 40))
        for r in range(nR):
            L = int(rng.integers(5, 261))
            src = haps[int(rng.integers(0, len(haps)))]
            p0 = int(rng.integers(max(0, buf - L + 1), min(len(src) - L, buf + W)))
            s = src[p0:p0 + L].copy()
            e = rng.random(L) < 0.02
            s[e] = B[rng.integers(0, 4, int(e.sum()))]
            q = rng.integers(0, 60, L).astype(np.uint8)
            kind = int(rng.choice([0, 0, 0, 1, 2]))
            flags = 512 if (kind == 1 and rng.random() < 0.5) else 0
            pos = ws - buf + p0 + int(rng.choice([0, 0, 0, 3, -40, 150]))
            if kind == 2:
                pos += int(rng.integers(-3000, 3000))
            mapq = int(rng.choice([0, 5, 29, 60, 60, 255]))
            reads.append((kind, pos, s, q, flags, mapq))
        reads.sort(key=lambda x: (x[0], x[1]))
        wins.append((ws, ws + W, buf, haps, reads))
    hap_seq = np.concatenate([h for w in wins for h in w[3]])
    hl = np.array([len(h) for w in wins for h in w[3]])
    allr = [r for w in wins for r in w[4]]
    rl = np.array([len(r[2]) for r in allr])
    return HostBatch(
        n_ind=1, win_hap_begin=np.concatenate([[0], np.cumsum([len(w[3]) for w in wins])]).astype(np.int32),
        win_read_begin=np.concatenate([[0], np.cumsum([len(w[4]) for w in wins])]).astype(np.int32),
        win_start=np.array([w[0] for w in wins], dtype=np.int32), win_end=np.array([w[1] for w in wins], dtype=np.int32),
        win_flank=np.array([w[2] for w in wins], dtype=np.int32), hap_seq=hap_seq,
        hap_off=np.concatenate([[0], np.cumsum(hl)]).astype(np.int64),
        read_seq=np.concatenate([r[2] for r in allr]), read_qual=np.concatenate([r[3] for r in allr]),
        read_off=np.concatenate([[0], np.cumsum(rl)]).astype(np.int64),
        read_pos=np.array([r[1] for r in allr], dtype=np.int32),
        read_end=np.array([r[1] + len(r[2]) for r in allr], dtype=np.int32),
        read_mapq=np.array([r[5] for r in allr], dtype=np.uint8), read_flags=np.array([r[4] for r in allr], dtype=np.int32),
        read_kind=np.array([r[0] for r in allr], dtype=np.uint8),
        seg_read_begin=np.concatenate([[0], np.cumsum([len(w[4]) for w in wins])]).astype(np.int32),
        seg_n_good=np.array([sum(1 for r in w[4] if r[0] == 0) for w in wins], dtype=np.int32))


@pytest.mark.parametrize("do_flank", [0, 1])
def test_likelihood_cache_golden_vectors(eng, golden_dir, do_flank):
    """a7-a10 end to end against outputs of the reference's own chaplotype.pyx texts (Haplotype.alignReads with its skip
    rules and read-class order, alignReadToHaplotype, annotateWithGapOpen) -- bit-identical log-likelihoods."""
    cases = [c for c in json.load(gzip.open(os.path.join(golden_dir, "haplotype_cases.json.gz"), "rt"))
             if c["calc_flank"] == do_flank]
    assert len(cases) >= 16
    u8 = lambda b: np.frombuffer(b, dtype=np.uint8)
    haps = [u8(h.encode()) for c in cases for h in c["haps"]]
    reads = [r for c in cases for r in c["reads"]]
    hl = np.array([len(h) for h in haps]); rl = np.array([len(r["seq"]) for r in reads])
    nh = [len(c["haps"]) for c in cases]; nr = [len(c["reads"]) for c in cases]
    hb = HostBatch(
        n_ind=1, win_hap_begin=np.concatenate([[0], np.cumsum(nh)]).astype(np.int32),
        win_read_begin=np.concatenate([[0], np.cumsum(nr)]).astype(np.int32),
        win_start=np.array([c["start"] for c in cases], dtype=np.int32), win_end=np.array([c["end"] for c in cases], dtype=np.int32),
        win_flank=np.array([c["buf"] for c in cases], dtype=np.int32), hap_seq=np.concatenate(haps),
        hap_off=np.concatenate([[0], np.cumsum(hl)]).astype(np.int64),
        read_seq=np.concatenate([u8(r["seq"].encode()) for r in reads]),
        read_qual=np.concatenate([np.array(r["qual"], dtype=np.uint8) for r in reads]),
        read_off=np.concatenate([[0], np.cumsum(rl)]).astype(np.int64),
        read_pos=np.array([r["pos"] for r in reads], dtype=np.int32), read_end=np.array([r["end"] for r in reads], dtype=np.int32),
        read_mapq=np.array([r["mapq"] for r in reads], dtype=np.uint8), read_flags=np.array([r["flag"] for r in reads], dtype=np.int32),
        read_kind=np.array([r["kind"] for r in reads], dtype=np.uint8),
        seg_read_begin=np.concatenate([[0], np.cumsum(nr)]).astype(np.int32),
        seg_n_good=np.array([sum(1 for r in c["reads"] if r["kind"] == 0) for c in cases], dtype=np.int32))
    db, st, ll, sc = run_align(eng, hb, do_flank)
    for w, c in enumerate(cases):
        exp = np.array([row[:-1] for row in c["cache"]])
        assert np.array_equal(ll[hb.pair_off[w]:hb.pair_off[w + 1]].reshape(exp.shape), exp), "window %d" % w
    # alignSingleRead (no skip rule): the same windows with every read passed as a brokenMate
    hb2 = HostBatch(**{**hb.__dict__, "read_kind": np.full(len(reads), 2, dtype=np.uint8), "pair_off": None, "gl_off": None})
    db, st, ll, sc = run_align(eng, hb2, do_flank)
    for w, c in enumerate(cases):
        exp = np.array(c["single"])
        assert np.array_equal(ll[hb.pair_off[w]:hb.pair_off[w + 1]].reshape(exp.shape), exp), "window %d" % w


def test_edge_cases_vs_oracle(eng, oracle):
    hb = edge_batch()
    db, st, ll, sc = run_align(eng, hb)
    check_against_oracle(oracle, hb, ll, sc, st)
    assert (sc == -1).any() and (ll == -300.0).any()          # skipped reads and mapq-0 reads are present


def test_empty_batch_and_null_checks(eng):
    import ctypes as C
    lib = _lib.load()
    z = _lib.WindowBatch()
    assert lib.plat_align_window_batch(eng.ctx, C.byref(z), 0, 0, None, None, None, None) == 0
    assert lib.plat_align_window_batch(eng.ctx, None, 0, 0, None, None, None, None) == -1
    assert lib.plat_dp_batch(eng.ctx, 0, 100, None, None, None, None, None, 3, 2, None, None) == 0


def test_async_entry_point_matches_sync_and_reports_errors(eng):
    """plat_align_window_batch_async: identical outputs without any read-back; bad hints / bad input / job overflow come
    back from plat_stream_sync."""
    for hb, flank in ((synth.config2(200), 0), (edge_batch(), 0), (synth.config2(60), 1)):
        db = eng.upload(hb)
        eng.align(db, want_stats=False, calc_flank_score=flank); eng.synchronize()
        ll0, sc0 = db.loglik.cpu().numpy().copy(), db.score.cpu().numpy().copy()
        db.loglik.zero_(); db.score.zero_()
        eng.align_async(db, calc_flank_score=flank); eng.synchronize()
        assert np.array_equal(db.loglik.cpu().numpy(), ll0) and np.array_equal(db.score.cpu().numpy(), sc0)
    hb = synth.config2(50)
    db = eng.upload(hb)
    for field, val in (("max_read_len", 100), ("max_hap_len", 300), ("n_pairs", hb.n_pairs - 1), ("max_reads_per_window", 3)):
        h = _lib.BatchHints.from_buffer_copy(db.hints)
        setattr(h, field, val)
        eng.align_async(db, hints=h)
        with pytest.raises(_lib.PlatypusDeviceError) as e:
            eng.synchronize()
        assert e.value.code == -10, field
    eng.synchronize()                                                       # the error is reported once
    # a tandem-repeat batch needs more extra job slots than a tiny explicit capacity: overflow is reported, not UB
    eb = edge_batch()
    db = eng.upload(eb)
    h = _lib.BatchHints.from_buffer_copy(db.hints)
    h.extra_jobs_cap = 1
    eng2 = Engine(0)                                                        # fresh context: its job buffer has no spare capacity yet
    db2 = eng2.upload(eb)
    eng2.align_async(db2, hints=h)
    with pytest.raises(_lib.PlatypusDeviceError) as e:
        eng2.synchronize()
    assert e.value.code == -8
    eng2.close()
    bad = HostBatch(**{**hb.__dict__, "read_seq": np.where(np.arange(len(hb.read_seq)) == 77, 200, hb.read_seq).astype(np.uint8),
                       "pair_off": None, "gl_off": None})
    db = eng.upload(bad)
    eng.align_async(db)
    with pytest.raises(_lib.PlatypusDeviceError) as e:
        eng.synchronize()
    assert e.value.code == -9


def test_unsupported_options_fail_loudly(eng):
    hb = synth.config1()
    db = eng.upload(hb)
    with pytest.raises(_lib.PlatypusDeviceError) as e:
        eng.align(db, use_mapq_cap=1)
    assert e.value.code == -6
    # --calculateFlankScore=1 without flanks makes the reference dereference a NULL alignment buffer: refused
    nf = HostBatch(**{**hb.__dict__, "win_flank": np.zeros_like(hb.win_flank), "pair_off": None, "gl_off": None})
    with pytest.raises(_lib.PlatypusDeviceError) as e:
        eng.align(eng.upload(nf), calc_flank_score=1)
    assert e.value.code == -6


def test_flank_score_mode_vs_oracle(eng, oracle):
    """a2: every DP in traceback mode, flank part of the alignment subtracted (calign.pyx:235-245,261-264)."""
    for hb in (synth.config2(120), edge_batch()):
        db, st, ll, sc = run_align(eng, hb, do_flank=1)
        check_against_oracle(oracle, hb, ll, sc, st, do_flank=1)
    # the option changes results (reads with errors in the flanks score lower), so the test is not vacuous
    hb = synth.config2(120)
    sc0 = run_align(eng, hb, 0)[3]
    sc1 = run_align(eng, hb, 1)[3]
    assert (sc1 <= sc0).all() and (sc1 < sc0).any()


def test_error_codes_from_device_validation(eng):
    hb = synth.config1()
    # haplotype longer than 16384 (chaplotype.pyx:180-183)
    big = HostBatch(**{**hb.__dict__, "hap_seq": np.full(4 * 17000, ord("A"), dtype=np.uint8),
                       "hap_off": np.arange(5, dtype=np.int64) * 17000, "pair_off": None, "gl_off": None})
    with pytest.raises(_lib.PlatypusDeviceError) as e:
        eng.align(eng.upload(big))
    assert e.value.code == -4
    # haplotype shorter than read + 15 (the reference reads past the buffer here)
    short = HostBatch(**{**hb.__dict__, "hap_seq": hb.hap_seq[:4 * 110].copy(),
                         "hap_off": np.arange(5, dtype=np.int64) * 110, "pair_off": None, "gl_off": None})
    with pytest.raises(_lib.PlatypusDeviceError) as e:
        eng.align(eng.upload(short))
    assert e.value.code == -5
    # non-ASCII byte
    bad = HostBatch(**{**hb.__dict__, "read_seq": np.where(np.arange(len(hb.read_seq)) == 5, 200, hb.read_seq).astype(np.uint8),
                       "pair_off": None, "gl_off": None})
    with pytest.raises(_lib.PlatypusDeviceError) as e:
        eng.align(eng.upload(bad))
    assert e.value.code == -9


def test_maximum_haplotype_length(eng, oracle):
    """hapLen = 16384 (the reference's hard cap) with a tandem-repeat stretch: the largest LDS configuration
    of the seeding kernel and a job list far larger than the initial capacity (re-run path)."""
    rng = np.random.default_rng(5)
    B = np.frombuffer(b"ACGT", dtype=np.uint8)
    hap = B[rng.integers(0, 4, 16384)]
    hap[8000:8400] = np.resize(np.frombuffer(b"AC", dtype=np.uint8), 400)
    reads, pos = [], []
    for p in (100, 7990, 8100, 8200, 16000, 16384 - 250):
        reads.append(hap[p:p + 250].copy()); pos.append(p)
    reads[2][100] = ord("G")
    nR = len(reads)
    hb = HostBatch(
        n_ind=1, win_hap_begin=np.array([0, 1], dtype=np.int32), win_read_begin=np.array([0, nR], dtype=np.int32),
        win_start=np.array([500], dtype=np.int32), win_end=np.array([15884], dtype=np.int32),
        win_flank=np.array([500], dtype=np.int32), hap_seq=hap, hap_off=np.array([0, 16384], dtype=np.int64),
        read_seq=np.concatenate(reads), read_qual=np.full(nR * 250, 30, dtype=np.uint8),
        read_off=np.arange(nR + 1, dtype=np.int64) * 250, read_pos=np.array(pos, dtype=np.int32),
        read_end=np.array(pos, dtype=np.int32) + 250, read_mapq=np.full(nR, 60, dtype=np.uint8),
        read_flags=np.zeros(nR, dtype=np.int32), read_kind=np.zeros(nR, dtype=np.uint8),
        seg_read_begin=np.array([0, nR], dtype=np.int32), seg_n_good=np.array([nR], dtype=np.int32))
    db, st, ll, sc = run_align(eng, hb)
    check_against_oracle(oracle, hb, ll, sc, st)


# ---- a11 / a12 -----------------------------------------------------------------------------------
def check_genotypes(oracle, hb, ll, logl, gl, gof):
    for w in range(hb.n_windows):
        H = hb.win_hap_begin[w + 1] - hb.win_hap_begin[w]
        R = hb.win_read_begin[w + 1] - hb.win_read_begin[w]
        G = H * (H + 1) // 2
        llw = ll[hb.pair_off[w]:hb.pair_off[w + 1]].reshape(H, R)
        for i in range(hb.n_ind):
            s = w * hb.n_ind + i
            a, b = hb.seg_read_begin[s] - hb.win_read_begin[w], hb.seg_read_begin[s + 1] - hb.win_read_begin[w]
            ol, og, of = oracle.population_setup_ind(llw[:, a:b], int(hb.seg_n_good[s]))
            o = hb.gl_off[w]
            got_l = logl[o + i * G:o + (i + 1) * G]
            # sums of fp64 terms in read order: identical, except for the log(0.5(e^a+e^b)) branch which
            # uses the device libm instead of glibc (<= few ulp per term): tolerance 1e-12 relative,
            # far inside north_star's 1e-4.
            assert np.allclose(got_l, ol, rtol=1e-12, atol=0), (w, i)
            assert np.allclose(gl[o + i * G:o + (i + 1) * G], og, rtol=1e-10, atol=1e-300)
            assert np.allclose(gof[o + np.arange(G) * hb.n_ind + i], of, rtol=1e-13, atol=0)


def test_genotype_likelihoods_vs_oracle(eng, oracle):
    for hb in (synth.config1(), synth.config2(150), edge_batch(), synth.config5(12, n_ind=7)):
        db = eng.upload(hb)
        eng.call_windows(db)
        eng.synchronize()
        ll = db.loglik.cpu().numpy()[:hb.n_pairs]
        check_genotypes(oracle, hb, ll, db.logl.cpu().numpy(), db.gl.cpu().numpy(), db.gof.cpu().numpy())


def test_population_mode_vs_oracle(eng, oracle):
    hb = synth.config5(6, n_ind=20)
    db, st, ll, sc = run_align(eng, hb)
    check_against_oracle(oracle, hb, ll, sc, st)


# ---- full BASELINE size: size-independent properties --------------------------------------------------
def test_config2_full_size_properties(eng, oracle):
    """10k windows (BASELINE config 2).  The oracle cannot do this in seconds, so check
    (1) a random sample of windows against the oracle, (2) invariance under window permutation,
    (3) identical haplotype rows / homozygous-genotype identity, (4) bounds."""
    hb = synth.config2(10000)
    db = eng.upload(hb)
    st = eng.call_windows(db)
    eng.synchronize()
    ll = db.loglik.cpu().numpy()[:hb.n_pairs]
    sc = db.score.cpu().numpy()[:hb.n_pairs]
    logl = db.logl.cpu().numpy()
    rng = np.random.default_rng(0)
    sample = sorted(rng.choice(hb.n_windows, 120, replace=False).tolist())
    for w, (oll, osc, n) in zip(sample, oracle_windows(oracle, hb, sample)):
        a, b = hb.pair_off[w], hb.pair_off[w + 1]
        assert np.array_equal(sc[a:b].reshape(osc.shape), osc)
        assert np.array_equal(ll[a:b].reshape(oll.shape), oll)
    assert (ll <= 0).all() and (ll >= -300).all() and (sc >= -1).all()
    assert st.n_pairs == hb.n_pairs and st.n_dp_reference >= st.n_pairs_aligned
    # homozygous genotype (a,a): logl == in-order sum of that haplotype's row
    for w in sample[:40]:
        H = hb.win_hap_begin[w + 1] - hb.win_hap_begin[w]; R = hb.win_read_begin[w + 1] - hb.win_read_begin[w]
        rows = ll[hb.pair_off[w]:hb.pair_off[w + 1]].reshape(H, R)
        g = 0
        for a in range(H):
            acc = 0.0
            for x in rows[a]:
                acc += x
            assert logl[hb.gl_off[w] + g] == acc
            g += H - a
    # permutation invariance: reversed window order gives the same per-window blocks
    perm = np.arange(hb.n_windows)[::-1]
    hb2 = hb.subset(perm[:2000])
    db2 = eng.upload(hb2)
    eng.call_windows(db2)
    eng.synchronize()
    ll2 = db2.loglik.cpu().numpy()[:hb2.n_pairs]
    for k in range(0, 2000, 7):
        w = perm[k]
        assert np.array_equal(ll2[hb2.pair_off[k]:hb2.pair_off[k + 1]], ll[hb.pair_off[w]:hb.pair_off[w + 1]])


def test_dp_add_flavours_agree_with_oracle(eng, oracle):
    """The DP adds both packed halves with one 32-bit add when the read's quality sum rules out a carry (dp_core.hpp),
    and with true 16-bit packed adds otherwise.  Windows whose reads straddle the threshold (quality sums 9 000 .. 24 000,
    many mismatches so that costs get large) must match the oracle either way."""
    rng = np.random.default_rng(31337)
    B = np.frombuffer(b"ACGT", dtype=np.uint8)
    wins = []
    for w in range(12):
        L = 250
        buf = 500
        W = 60
        ref = B[rng.integers(0, 4, W + 2 * buf + 200)]
        ws = 100 + buf
        hap0 = ref[ws - buf:ws + W + buf].copy()
        hap1 = hap0.copy(); hap1[buf + 20] = B[(int(np.searchsorted(B, hap1[buf + 20])) + 1) % 4]
        reads = []
        for r in range(48):
            off = int(rng.integers(buf - L + 10, buf + W - 10))
            seq = hap0[off:off + L].copy()
            nmm = int(rng.choice([0, 2, 30, 120, 250]))
            idx = rng.choice(L, nmm, replace=False)
            seq[idx] = B[rng.integers(0, 4, nmm)]
            qmean = float(rng.choice([36, 60, 75, 93]))                     # quality sums from ~9 000 to ~23 000
            q = np.clip(rng.normal(qmean, 3, L), 0, 93).astype(np.uint8)
            reads.append((0, ws - buf + off, seq, q, 0, 60))
        reads.sort(key=lambda x: (x[0], x[1]))
        wins.append((ws, ws + W, buf, [hap0, hap1], reads))
    hap_seq = np.concatenate([h for w in wins for h in w[3]])
    hl = np.array([len(h) for w in wins for h in w[3]])
    allr = [r for w in wins for r in w[4]]
    rl = np.array([len(r[2]) for r in allr])
    qsum = np.array([int(r[3].sum()) for r in allr])
    assert (qsum <= 15000).sum() > 50 and (qsum > 15000).sum() > 50
    hb = HostBatch(
        n_ind=1, win_hap_begin=np.concatenate([[0], np.cumsum([len(w[3]) for w in wins])]).astype(np.int32),
        win_read_begin=np.concatenate([[0], np.cumsum([len(w[4]) for w in wins])]).astype(np.int32),
        win_start=np.array([w[0] for w in wins], dtype=np.int32), win_end=np.array([w[1] for w in wins], dtype=np.int32),
        win_flank=np.array([w[2] for w in wins], dtype=np.int32), hap_seq=hap_seq,
        hap_off=np.concatenate([[0], np.cumsum(hl)]).astype(np.int64),
        read_seq=np.concatenate([r[2] for r in allr]), read_qual=np.concatenate([r[3] for r in allr]),
        read_off=np.concatenate([[0], np.cumsum(rl)]).astype(np.int64),
        read_pos=np.array([r[1] for r in allr], dtype=np.int32),
        read_end=np.array([r[1] + len(r[2]) for r in allr], dtype=np.int32),
        read_mapq=np.array([r[5] for r in allr], dtype=np.uint8), read_flags=np.array([r[4] for r in allr], dtype=np.int32),
        read_kind=np.array([r[0] for r in allr], dtype=np.uint8),
        seg_read_begin=np.concatenate([[0], np.cumsum([len(w[4]) for w in wins])]).astype(np.int32),
        seg_n_good=np.array([len(w[4]) for w in wins], dtype=np.int32))
    db, st, ll, sc = run_align(eng, hb)
    check_against_oracle(oracle, hb, ll, sc, st)
    assert sc.max() > 1000                                                   # costs far beyond anything config 2 produces


def _adversarial_batch(seed, n_windows=260, gapped=False, bigq=False):
    """Windows built to stress k_seed's ungapped-alignment proof: references with homopolymers and tandem repeats (cheap gaps,
    non-unique k-mers) and N runs, reads of 36..250 bp with 0..3 substitutions anywhere -- first and last bases included --, read
    N's, low-quality tails, quality minima down to 0, haplotypes differing by SNPs next to repeats.  gapped: a third of the reads
    also carry a 1..4 base insertion or deletion (often within 20 bases of an end, where the ungapped alignment shows only a
    few mismatches and a gapped one is cheaper), a third a tight cluster of 2..4 substitutions, and qualities reach 41 more often.
    bigq: the WRAP regime of align.c's int16 adds (align.c:81 "no overflow checks"): reads of 150 / 250 bp with qualities 60..93
    (quality sums 9 000 .. 23 000, most above dp_core.hpp's 15 000), homopolymers of 45..70 bases (gap-open penalties down to 1)."""
    from platypus_amd import hostapi as H
    rng = np.random.default_rng(seed)
    B = b"ACGT"
    rnd = lambda n: bytes(rng.choice(list(B), n).astype(np.uint8))
    specs = []
    for w in range(n_windows):
        L = int(rng.choice([150, 250, 250, 250])) if bigq else int(rng.choice([36, 76, 100, 150, 150, 250]))
        buf = min(2 * L, 500)
        W = int(rng.integers(20, 80))
        ref = bytearray(rnd(W + 2 * buf + 40))
        for _ in range(int(rng.integers(0, 6))):
            p = int(rng.integers(0, len(ref) - 60)); k = int(rng.integers(3, 40))
            u = rnd(int(rng.choice([1, 1, 1, 2, 3, 4, 6])))
            ref[p:p + k] = (u * k)[:k]
        if bigq and rng.random() < 0.5:                          # a long homopolymer near the window: gap-open penalties 1..3
            k = int(rng.integers(45, 70)); p = int(rng.integers(buf - 60, buf + W))
            ref[p:p + k] = rnd(1) * k
        if rng.random() < 0.12:                                  # haplotype N's (cost 0 in the DP): the proof must stand aside
            p = int(rng.integers(0, len(ref) - 8)); ref[p:p + int(rng.integers(1, 6))] = b"N" * 5
        ref = bytes(ref[:W + 2 * buf + 40])
        ws = 5000
        base = ref[:W + 2 * buf]
        haps = [base]
        for _ in range(int(rng.integers(1, 4))):
            h = bytearray(base)
            for _ in range(int(rng.integers(1, 3))):
                p = buf + int(rng.integers(0, W))
                h[p] = B[((B.index(h[p]) if h[p] in B else 0) + 1 + int(rng.integers(0, 3))) % 4]
            if bytes(h) not in haps:
                haps.append(bytes(h))
        reads = []
        for _ in range(int(rng.integers(20, 60))):
            src = haps[int(rng.integers(0, len(haps)))]
            off = int(rng.integers(max(0, buf - L + 7), min(len(src) - L, buf + W - 7) + 1))
            seq = bytearray(src[off:off + L])
            for _ in range(int(rng.choice([0, 0, 1, 1, 1, 2, 2, 3]))):
                p = int(rng.choice([0, 1, 2, L - 1, L - 2, L - 3, int(rng.integers(0, L)), int(rng.integers(0, L))]))
                seq[p] = B[((B.index(seq[p]) if seq[p] in B else 0) + 1 + int(rng.integers(0, 3))) % 4]
            if gapped:
                u = rng.random()
                if u < 0.35:
                    n = int(rng.integers(1, 5))
                    p = int(rng.choice([int(rng.integers(1, 20)), L - int(rng.integers(2, 20)), int(rng.integers(1, L - 1))]))
                    if rng.random() < 0.5:                       # bases inserted into the read, its tail drops off
                        seq[p:p] = rnd(n) if rng.random() < 0.5 else bytes(seq[max(0, p - 1):p] or b"A") * n
                        del seq[L:]
                    else:                                        # bases deleted from the read, the source fills the tail
                        del seq[p:p + n]
                        seq += src[off + L:off + L + (L - len(seq))]
                        seq += rnd(L - len(seq))
                elif u < 0.7:
                    c0 = int(rng.integers(0, L - 15)); span = int(rng.integers(3, 15))
                    for p in sorted(set(int(x) for x in rng.integers(c0, c0 + span, int(rng.integers(2, 5))))):
                        seq[p] = B[((B.index(seq[p]) if seq[p] in B else 0) + 1 + int(rng.integers(0, 3))) % 4]
                assert len(seq) == L
            if rng.random() < 0.05:
                seq[int(rng.integers(0, L))] = ord("N")          # read N: costs its quality against any base
            q = np.clip(rng.normal(36 if gapped else 33, 6, L), 1, 41 if gapped else 60).astype(np.uint8)
            if bigq:
                lo = int(rng.choice([60, 60, 70, 85]))
                q = rng.integers(lo, 94, L).astype(np.uint8)
            mode = rng.random() if not bigq else 0.2 + 0.8 * rng.random()
            if mode < 0.2:
                q[L - int(rng.integers(1, 30)):] = rng.integers(1, 12, 1)[0]
            elif mode < 0.3:
                q[:] = rng.integers(1, 6, L)
            elif mode < 0.35:
                q[int(rng.integers(0, L))] = 0
            reads.append(H.AlignedRead(bytes(seq), bytes(q.tolist()), ws - buf + off + int(rng.choice([0, 0, 0, 0, 1, -2])), 60, 3))
        specs.append((haps, ws, ws + W, buf, [H.bamReadBuffer(reads)]))
        for b_ in specs[-1][4]:
            b_.setWindowPointers(ws, ws + W)
    return H._pack_windows(specs)


def test_ungapped_shortcut_equals_the_dp_everywhere(eng, oracle):
    """k_seed finishes pairs whose read differs from the haplotype in one or two bases without a DP when it can prove that no
    other path of the band is cheaper.  Every score must equal what the DP gives for the same pair (PLAT_NO_UNGAPPED=1 sends
    all of them through the DP), on a stress batch and on BASELINE config 2; a sample is checked against the oracle too."""
    import os
    from platypus_amd import synth
    used = 0
    for hb in (_adversarial_batch(1), _adversarial_batch(2), _adversarial_batch(3, gapped=True), _adversarial_batch(4, gapped=True),
               synth.config2(1500, seed=9), synth.config2_hard(1200, seed=11)):
        res = {}
        # "0": both shortcuts; "1": the ungapped proof off; "all": the exact-match shortcut off too (every reference DP is run);
        # "nolow": the proof values its unique windows by the smallest quality only (no count of low-quality bases)
        for mode, env in (("0", {}), ("1", {"PLAT_NO_UNGAPPED": "1"}), ("all", {"PLAT_NO_UNGAPPED": "1", "PLAT_NO_EXACT": "1"}),
                          ("nolow", {"PLAT_NO_NLOW": "1"})):
            os.environ.update(env)
            try:
                db = eng.upload(hb)
                st = eng.align(db, want_stats=True)
                eng.synchronize()
                res[mode] = (db.score.cpu().numpy()[:hb.n_pairs].copy(), db.loglik.cpu().numpy()[:hb.n_pairs].copy(), int(st.n_dp_launched),
                             int(st.n_dp_reference))
            finally:
                for k in env:
                    os.environ.pop(k, None)
        for mode in ("1", "all", "nolow"):
            assert np.array_equal(res["0"][0], res[mode][0]), mode
            assert np.array_equal(res["0"][1], res[mode][1]), mode
        assert res["0"][2] < res["1"][2] < res["all"][2]       # each shortcut did take pairs away from the DP
        assert res["0"][2] <= res["nolow"][2]
        # with both off the device runs the reference's DPs: the counts differ only where the reference stops at a candidate that
        # scores 0 (the device has run the later ones too) or aligns the mapping position a second time (calign.pyx:252-267;
        # the device re-uses that candidate's score)
        assert res["all"][3] == res["0"][3] and res["all"][2] >= 0.98 * res["all"][3]
        used += res["1"][2] - res["0"][2]
    assert used > 10000
    # the wrap regime of the reference's int16 adds (quality sums above 15 000): the ungapped proof stands aside for these reads
    # (its cost model is exact arithmetic), the exact-match shortcut does not have to; the DP wraps as align.c does.  Same scores
    # with the shortcuts, without them, and from the oracle.
    for seed, gp in ((21, False), (22, True)):
        hb = _adversarial_batch(seed, 60, gapped=gp, bigq=True)
        assert (np.add.reduceat(hb.read_qual.astype(np.int64), hb.read_off[:-1]) > 15000).mean() > 0.4
        res = {}
        for mode, env in (("0", {}), ("all", {"PLAT_NO_UNGAPPED": "1", "PLAT_NO_EXACT": "1"})):
            os.environ.update(env)
            try:
                db = eng.upload(hb)
                st = eng.align(db, want_stats=True)
                eng.synchronize()
                res[mode] = (db.score.cpu().numpy()[:hb.n_pairs].copy(), db.loglik.cpu().numpy()[:hb.n_pairs].copy(), int(st.n_dp_launched))
            finally:
                for k in env:
                    os.environ.pop(k, None)
        assert np.array_equal(res["0"][0], res["all"][0]) and np.array_equal(res["0"][1], res["all"][1])
        assert res["0"][2] < res["all"][2]
        for w in range(0, hb.n_windows, 3):
            rd = hb.window_reads(w)
            exp = oracle.align_window(hb.window_haps(w), int(hb.win_start[w]), int(hb.win_end[w]), int(hb.win_flank[w]), rd)
            R, H_ = len(rd["seq"]), hb.win_hap_begin[w + 1] - hb.win_hap_begin[w]
            assert np.array_equal(res["0"][1][hb.pair_off[w]:hb.pair_off[w] + H_ * R].reshape(H_, R), np.asarray(exp[0]).reshape(H_, R))
    hb = _adversarial_batch(3, 40, gapped=True)
    db = eng.upload(hb)
    eng.align(db, want_stats=False)
    eng.synchronize()
    got = db.loglik.cpu().numpy()
    for w in range(hb.n_windows):
        rd = hb.window_reads(w)
        exp = oracle.align_window(hb.window_haps(w), int(hb.win_start[w]), int(hb.win_end[w]), int(hb.win_flank[w]), rd)
        R = len(rd["seq"])
        H_ = hb.win_hap_begin[w + 1] - hb.win_hap_begin[w]
        assert np.array_equal(got[hb.pair_off[w]:hb.pair_off[w] + H_ * R].reshape(H_, R), np.asarray(exp[0]).reshape(H_, R))


def test_config5_real_width(eng, oracle):
    """BASELINE config 5 at its real width: 100 samples per window.  The oracle on a sample of windows (per-read
    log-likelihoods bit for bit, genotype log-likelihoods to 1e-12), invariance under a permutation of the windows, and the EM /
    genotype calls against the oracle's on the device's own likelihoods."""
    from platypus_amd import synth
    hb = synth.config5(60, 100, seed=5005)
    assert hb.n_ind == 100 and hb.n_windows == 60
    db = eng.upload(hb)
    eng.call_windows(db, want_stats=False)
    eng.em(db, 100, 0)
    eng.synchronize()
    ll, logl, gl = db.loglik.cpu().numpy(), db.logl.cpu().numpy(), db.gl.cpu().numpy()
    freq, calls = db.freq.cpu().numpy(), db.calls.cpu().numpy().reshape(hb.n_windows, hb.n_ind)
    rng = np.random.default_rng(5)
    for w in rng.choice(hb.n_windows, 6, replace=False).tolist():
        rd = hb.window_reads(w)
        H_ = int(hb.win_hap_begin[w + 1] - hb.win_hap_begin[w])
        R = len(rd["seq"])
        exp = oracle.align_window(hb.window_haps(w), int(hb.win_start[w]), int(hb.win_end[w]), int(hb.win_flank[w]), rd)
        oll = np.asarray(exp[0]).reshape(H_, R)
        assert np.array_equal(ll[hb.pair_off[w]:hb.pair_off[w] + H_ * R].reshape(H_, R), oll)
        G = H_ * (H_ + 1) // 2
        r0 = int(hb.win_read_begin[w])
        for i in rng.choice(hb.n_ind, 8, replace=False).tolist():
            s = w * hb.n_ind + i
            a, b = int(hb.seg_read_begin[s]) - r0, int(hb.seg_read_begin[s + 1]) - r0
            ol, ogl, _ = oracle.population_setup_ind(np.ascontiguousarray(oll[:, a:b]), int(hb.seg_n_good[s]))
            o = int(hb.gl_off[w]) + i * G
            assert np.allclose(logl[o:o + G], ol, rtol=1e-12, atol=0)
            assert np.allclose(gl[o:o + G], ogl, rtol=1e-12, atol=0)
        # EM + calls on the device's own genotype likelihoods
        o = int(hb.gl_off[w])
        nreads = hb.seg_n_good[w * hb.n_ind:(w + 1) * hb.n_ind]
        of, _, oc, _, _ = oracle.em_call(nreads, gl[o:o + hb.n_ind * G].reshape(hb.n_ind, G), 100, 0)
        h0 = int(hb.win_hap_begin[w])
        assert np.array_equal(freq[h0:h0 + H_], of)
        assert np.array_equal(calls[w], oc)
    # the windows in another order give the same per-window results
    perm = rng.permutation(hb.n_windows).tolist()
    hp = hb.subset(perm)
    dp_ = eng.upload(hp)
    eng.call_windows(dp_, want_stats=False)
    eng.em(dp_, 100, 0)
    eng.synchronize()
    logl2, freq2 = dp_.logl.cpu().numpy(), dp_.freq.cpu().numpy()
    for k, w in enumerate(perm):
        n = int(hb.gl_off[w + 1] - hb.gl_off[w])
        assert np.array_equal(logl2[hp.gl_off[k]:hp.gl_off[k] + n], logl[hb.gl_off[w]:hb.gl_off[w] + n])
        H_ = int(hb.win_hap_begin[w + 1] - hb.win_hap_begin[w])
        assert np.array_equal(freq2[hp.win_hap_begin[k]:hp.win_hap_begin[k] + H_], freq[hb.win_hap_begin[w]:hb.win_hap_begin[w] + H_])


def test_rccl_process_group_of_one_rank_gathers_and_merges():
    """What one GPU allows of the multi-GPU exchange: an RCCL ("nccl") process group with one rank on cuda:0, the size
    all_gather of gather_records on device tensors, and the ordered merge (runner.py:301-352)."""
    import os, socket, subprocess, sys
    code = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from platypus_amd import sharding
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
recs = [("2", 7, "b"), ("1", 5, "a"), ("X", 1, "c")]
recs.sort(key=lambda r: (sharding.chrom_key(r[0]), r[1]))
got = sharding.gather_records(sharding.encode_records(recs), dist, device=torch.device("cuda", 0))
t = torch.ones(4, device="cuda"); dist.all_reduce(t); dist.barrier()
assert dist.get_backend() == "nccl" and len(got) == 1 and float(t.sum()) == 4.0
print("MERGED", ",".join(sharding.merge_record_streams([sharding.decode_records(p) for p in got])))
dist.destroy_process_group()
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
    assert "MERGED a,b,c" in r.stdout


def test_haplotypes_derived_from_the_windows_first_one_give_the_same_scores(eng):
    """PLAT_SEED_SHARE=1 (off by default): k_seed_base sweeps the first haplotype of every window once, k_seed derives the
    window's other haplotypes from it (patched planes, conservative uniqueness flags) instead of sweeping them.  Scores and likelihoods
    are those of the default path on config 2, the hard workload and the stress batches; only the number of DPs launched may grow."""
    import os
    from platypus_amd import synth
    for hb in (synth.config2(1500, seed=9), synth.config2_hard(800, seed=11), _adversarial_batch(5), _adversarial_batch(6, gapped=True), synth.config5(8, 20)):
        res = {}
        for mode in ("0", "1"):
            os.environ["PLAT_SEED_SHARE"] = mode
            try:
                db = eng.upload(hb)
                st = eng.align(db, want_stats=True)
                eng.synchronize()
                res[mode] = (db.score.cpu().numpy()[:hb.n_pairs].copy(), db.loglik.cpu().numpy()[:hb.n_pairs].copy(), int(st.n_dp_launched), int(st.n_dp_reference))
            finally:
                os.environ.pop("PLAT_SEED_SHARE", None)
        assert np.array_equal(res["0"][0], res["1"][0]) and np.array_equal(res["0"][1], res["1"][1])
        assert res["0"][3] == res["1"][3] and res["0"][2] <= res["1"][2] <= 1.02 * res["0"][2] + 50


def test_two_kernel_seeding_equals_the_fused_kernel(eng):
    """Round 4: the seeding stage is k_sweep (per haplotype) + k_pairs ((haplotype, read) pairs packed 64 to a wave, up to six haplotype
    records staged per wave, the k-mer index built per staged haplotype on demand); PLAT_SEED_FUSED=1 is rounds 1-3's single kernel.
    Same scores, likelihoods, reference DPs and launched DPs -- on config 2, the hard workload, the stress batches (tandem repeats: many
    index builds and slow-path pairs), a population batch (thousands of reads per window) and windows with few reads (R < 13: five whole
    haplotypes per wave), through both entry points."""
    import os
    from platypus_amd import synth
    few = synth.make_snp_windows(300, 21, read_len=150, depth=4)    # windows of ~5 reads
    for hb in (synth.config2(1500, seed=9), synth.config2_hard(800, seed=11), _adversarial_batch(5), _adversarial_batch(6, gapped=True),
               synth.config5(8, 20), few, synth.make_snp_windows(300, 22, read_len=150, depth=12)):      # (~15 reads: 64 pairs span five or six haplotypes)
        res = {}
        for mode in ("0", "1"):
            os.environ["PLAT_SEED_FUSED"] = mode
            try:
                db = eng.upload(hb)
                st = eng.align(db, want_stats=True)
                eng.synchronize()
                sync = (db.score.cpu().numpy()[:hb.n_pairs].copy(), db.loglik.cpu().numpy()[:hb.n_pairs].copy(), int(st.n_dp_launched), int(st.n_dp_reference))
                db2 = eng.upload(hb)
                eng.align_async(db2)
                eng.synchronize()
                res[mode] = sync + (db2.loglik.cpu().numpy()[:hb.n_pairs].copy(),)
            finally:
                os.environ.pop("PLAT_SEED_FUSED", None)
        assert np.array_equal(res["0"][0], res["1"][0]) and np.array_equal(res["0"][1], res["1"][1])
        assert res["0"][2:4] == res["1"][2:4]
        assert np.array_equal(res["0"][4], res["0"][1]) and np.array_equal(res["1"][4], res["1"][1])


def test_region_text_exchange_over_rccl_with_two_ranks():
    """The job's one exchange between REAL ranks (runner.py:301-352): two processes, one GPU each, backend "nccl" (= RCCL over
    xGMI): each rank's per-region text blocks travel device to device to rank 0 and are put in (chromosome key, start) order there by
    plat_copy_pieces; the merged text is what the line merge of the same texts gives.  Needs two visible GPUs: skipped on a one-GPU box,
    runs on the driver's multi-GPU node."""
    import os, socket, subprocess, sys
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL wants one device per rank)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
from platypus_amd import sharding, fastcaller as F
rank = int(os.environ["RANK"]); world = 2
torch.cuda.set_device(rank)
dev = torch.device("cuda", rank)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
regions = [("r%%d" %% g, 1000, 4000) for g in range(11)]                      # region g -> rank g %% 2 (runner.py:473-474)
per_rank = [regions[r::world] for r in range(world)]
def block(g):                                                                # a region's record lines (regions 3 and 8 call nothing)
    return b"" if g in (3, 8) else b"".join(b"r%%d\t%%d\t.\tA\tC\tline %%d of region %%d\n" %% (g, 1001 + 7 * k, k, g) for k in range(1 + g %% 4))
mine = [g for g in range(11) if g %% world == rank]
text = b"".join(block(g) for g in mine)
lens = np.array([len(block(g)) for g in mine], dtype=np.int64)
x = sharding.RegionTextExchange(per_rank, dist=dist, device=dev, device_index=rank)
for _ in range(2):                                                           # twice: the pinned block and the context are reused
    merged = x.exchange(text, lens)
    if rank == 0:
        want = F.text_bytes(F.merge_record_texts([b"".join(block(g) for g in range(11) if g %% world == r) for r in range(world)], raw=True))
        assert bytes(memoryview(merged)) == want == b"".join(block(g) for g in sorted(range(11), key=lambda g: sharding.chrom_key("r%%d" %% g)))
    else:
        assert merged is None
x.close()
dist.barrier()
if rank == 0:
    print("EXCHANGED over", dist.get_backend())
dist.destroy_process_group()
''' % root
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for rank in range(2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[1][-1500:] for o in outs)
    assert "EXCHANGED over nccl" in outs[0][0]
