"""Stage B on the device (plat_stage_b_batch: sorted candidates -> leftNormaliseIndel -> filterVariants -> calling windows -> window
pointers -> every valid combination of a window's variants -> haplotype bytes sorted -> the window batch) against the host's own
stage B (the code the reference goldens pin: tests/test_region_golden_cpu.py, tests/test_gpu_region_golden.py), through the native
region loop: the two must write the same record text, region by region, on SNP-only regions, indel-rich regions (normalisation,
equal candidates merging), dense regions (windows that go to the greedy filter = the host's code for that window), bad reads and
broken mates (the three window pointers), several regions per chunk, and option variants that change the rules."""
import io
import os

import numpy as np
import pytest

from platypus_amd import caller, fastcaller as F, hostapi as H, synth
from platypus_amd.options import default_options
from platypus_amd.vcfrecords import VCF

pytestmark = pytest.mark.gpu


def _work(n, seed, classes=False, **kw):
    rng = np.random.default_rng(seed)
    regs = [synth.config4_region(i, seed=seed, n_samples=1, **kw) for i in range(n)]
    fasta = H.FastaFile({r["chrom"]: r["ref"] for r in regs})
    work = []
    for r in regs:
        good, bad, broken = [], [], []
        for x in r["samples"][0]:
            a = H.AlignedRead(x["seq"], x["qual"], x["pos"], x["mapq"], x["flag"], end=x["end"], cigarOps=x["cigar"])
            u = rng.random() if classes else 1.0
            if u < 0.05:
                a.mapq = int(rng.integers(0, 20)); a.bitFlag |= 512; bad.append(a)
            elif u < 0.08:
                a.matePos = a.pos + int(rng.integers(-300, 300)); broken.append(a)
            else:
                good.append(a)
        work.append((r["chrom"], r["start"], r["end"], [H.bamReadBuffer(good, bad, broken, sample="S1")]))
    return fasta, work


def _native(work, fasta, per_chunk, host_b, packed=False, **over):
    old = os.environ.get("PLAT_CALLER_HOST_B")
    os.environ["PLAT_CALLER_HOST_B"] = "1" if host_b else "0"
    try:
        nc = F.NativeCaller(0, 2, per_chunk)
        txt = nc.call_regions([F.RegionReads.from_buffers(c, s, e, fasta, b, packed=packed) for c, s, e, b in work], ["S1"], default_options(**over))
        st = dict(nc.stats)
        nc.close()
    finally:
        if old is None:
            os.environ.pop("PLAT_CALLER_HOST_B", None)
        else:
            os.environ["PLAT_CALLER_HOST_B"] = old
    return txt, st


CASES = [
    # (regions, per chunk, region kwargs, read classes, options)
    (6, 4, dict(region_len=4000, snp_rate=3e-3, indel_rate=0.0, read_len=150, depth=30), False, {}),
    (6, 3, dict(region_len=4000, snp_rate=2e-3, indel_rate=3e-3, read_len=100, depth=35), True, {}),
    (5, 5, dict(region_len=3000, snp_rate=4e-2, indel_rate=6e-3, read_len=100, depth=20), False, {}),            # dense: greedy windows
    (5, 2, dict(region_len=2500, snp_rate=6e-3, indel_rate=2e-3, read_len=76, depth=50), True, dict(maxVariants=3, maxHaplotypes=12)),
    (4, 4, dict(region_len=2500, snp_rate=5e-3, indel_rate=1e-3, read_len=150, depth=25), False, dict(mergeClusteredVariants=0, minPosterior=0)),
    (4, 1, dict(region_len=6000, snp_rate=2e-3, indel_rate=2e-3, read_len=250, depth=30), True, dict(filterVarsByCoverage=0, maxVariants=12)),
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_stage_b_on_the_device_writes_the_text_of_stage_b_on_the_host(case):
    n, per, kw, classes, over = CASES[case]
    fasta, work = _work(n, 7100 + case, classes, **kw)
    dev, sd = _native(work, fasta, per, False, packed=bool(case % 2), **over)
    host, sh = _native(work, fasta, per, True, packed=bool(case % 2), **over)
    assert sd["n_regions_stage_b_device"] + sd["n_regions_stage_b_host"] == n and sd["n_regions_stage_b_device"] >= n - 1, sd
    assert sh["n_regions_stage_b_device"] == 0
    assert dev == host
    assert dev.count("\n") >= 8
    for k in ("n_windows", "n_variants", "n_candidate_records", "n_records", "n_windows_greedy", "n_windows_failed", "n_pairs"):
        assert sd[k] == sh[k], (k, sd[k], sh[k])
    assert sd["n_windows_stage_b_host"] >= sd["n_windows_greedy"]       # a window of the greedy filter is the host's


def test_windows_of_the_greedy_filter_stay_with_the_host_and_the_rest_of_their_region_on_the_device():
    fasta, work = _work(3, 7300, False, region_len=2500, snp_rate=9e-2, indel_rate=4e-3, read_len=100, depth=40)
    dev, sd = _native(work, fasta, 3, False)
    host, sh = _native(work, fasta, 3, True)
    assert dev == host and dev.count("\n") >= 8
    assert sd["n_windows_greedy"] == sh["n_windows_greedy"] > 0 and sd["n_windows_stage_b_host"] >= sd["n_windows_greedy"]
    assert sd["n_regions_stage_b_device"] >= 2


def test_stage_b_on_the_device_equals_the_python_region_loop():
    """... and the Python layer (window by window, hostapi's own Variant / WindowGenerator / Haplotype classes) writes it too."""
    fasta, work = _work(4, 7200, True, region_len=3000, snp_rate=4e-3, indel_rate=2e-3, read_len=100, depth=30)
    py = io.StringIO()
    caller.callVariantsInRegions(work, fasta, default_options(), VCF(["S1"]), py)
    dev, sd = _native(work, fasta, 4, False)
    assert sd["n_regions_stage_b_device"] == 4
    assert dev == py.getvalue()


def test_an_indel_in_a_repeat_is_moved_left_on_the_device_as_on_the_host():
    """leftNormaliseIndel: a deletion and an insertion inside a homopolymer / dinucleotide run, reported by the aligner at the run's RIGHT
    end, come out at its left end -- and reads that show the same indel at different places of the run merge into one candidate."""
    rng = np.random.default_rng(5)
    L = 1600
    ref = rng.choice(list(b"ACGT"), L).astype(np.uint8)
    ref[700:716] = ord("A")                                                 # A x 16
    ref[699], ref[716] = ord("C"), ord("G")
    ref[1000:1020] = np.frombuffer(b"CT" * 10, dtype=np.uint8)              # (CT) x 10
    ref[999], ref[1020] = ord("A"), ord("G")
    refb = ref.tobytes()
    reads = []
    for k in range(60):
        start = 560 + 9 * k
        rl = 150
        if start <= 690 and start + rl >= 730:                            # a one-base deletion of the A run, placed anywhere in it
            at = 700 + (k % 14)
            seq = refb[start:at] + refb[at + 1:start + rl + 1]
            cig = [(0, at - start), (2, 1), (0, rl - (at - start))]
            end = start + rl + 1
        elif start <= 990 and start + rl >= 1030:                         # a CT insertion into the CT run, placed at any unit
            at = 1000 + 2 * (k % 9)
            seq = refb[start:at] + b"CT" + refb[at:start + rl - 2]
            cig = [(0, at - start), (1, 2), (0, rl - 2 - (at - start))]
            end = start + rl - 2
        else:
            seq = refb[start:start + rl]
            cig = [(0, rl)]
            end = start + rl
        reads.append(H.AlignedRead(seq, bytes([35]) * len(seq), start, 60, 2 | 1 | (16 if k % 2 else 32), end=end, cigarOps=cig))
    fasta = H.FastaFile({"rr": refb})
    work = [("rr", 500, 1300, [H.bamReadBuffer(reads, sample="S1")])]
    dev, sd = _native(work, fasta, 1, False)
    host, sh = _native(work, fasta, 1, True)
    assert sd["n_regions_stage_b_device"] == 1
    assert dev == host
    pos = {int(ln.split("\t")[1]): ln.split("\t") for ln in dev.split("\n") if ln}
    assert 700 in pos and pos[700][3] == "CA" and pos[700][4] == "C", sorted(pos)          # VCF POS is 1-based: the base before the run
    assert 1000 in pos and pos[1000][3] == "A" and pos[1000][4] == "ACT", sorted(pos)


def _two_allele_work(seed, n_regions=4, n_sites=9, **kw):
    """One-sample regions with sites where the reads show TWO alternative SNP alleles in the same number of reads: a tie of position, type,
    length and support -- `sorted` keeps such a pair in the order the candidate generator's Python-2 dictionaries yield it."""
    rng = np.random.default_rng(seed)
    regs = [synth.config4_region(i, seed=seed, n_samples=1, **kw) for i in range(n_regions)]
    planted = 0
    for r in regs[:-1]:                                                     # (the last region stays as it is: no pair, no replay)
        taken = {v[0] for v in r["variants"]}
        sites = [p for p in rng.choice(np.arange(r["start"] + 150, r["end"] - 150), size=60, replace=False).tolist() if all(abs(p - q) > 12 for q in taken)][:n_sites]
        for p in sites:
            taken.add(p)
            alts = [b for b in b"ACGT" if b != r["ref"][p]]
            a1, a2 = (alts[k] for k in rng.choice(3, size=2, replace=False))
            reads = r["samples"][0]
            cover = [x for x in reads if x["pos"] + 12 <= p < x["pos"] + len(x["seq"]) - 12 and len(x["cigar"]) == 1]
            for k, x in enumerate(cover[:2 * min(14, len(cover) // 2)]):
                sq = bytearray(x["seq"])
                sq[p - x["pos"]] = a1 if k % 2 == 0 else a2
                x["seq"] = bytes(sq)
            planted += 1
    fasta = H.FastaFile({r["chrom"]: r["ref"] for r in regs})
    work = [(r["chrom"], r["start"], r["end"], [H.bamReadBuffer([H.AlignedRead(x["seq"], x["qual"], x["pos"], x["mapq"], x["flag"], end=x["end"], cigarOps=x["cigar"])
                                                                 for x in r["samples"][0]], sample="S1")]) for r in regs]
    return fasta, work, planted


@pytest.mark.parametrize("over", [dict(), dict(maxVariants=1), dict(maxVariants=2, minPosterior=0)])
def test_the_python2_dictionaries_are_replayed_on_the_device(over):
    """variantcaller.pyx:456-470: candidates that compare equal keep the order of two Python-2 dicts keyed by hash(Variant).  The device
    replays them (k_sb_variants: CPython's string / tuple hashes, the C-int narrowing of variant.pxd:31, open addressing with perturbation,
    the 4 x growth in slot order) for the regions that hold such a pair; the text is that of the host's own replay (the code the reference
    goldens pin) and of the Python loop, also where the pair's order decides what is kept (maxVariants = 1)."""
    fasta, work, planted = _two_allele_work(7400, region_len=3000, snp_rate=2e-3, indel_rate=5e-4, read_len=100, depth=40)
    assert planted >= 20
    dev, sd = _native(work, fasta, 4, False, **over)
    host, sh = _native(work, fasta, 4, True, **over)
    assert sd["n_regions_dict_replay_device"] == 3 and sd["n_regions_stage_b_device"] == 4, sd
    assert dev == host
    old = os.environ.get("PLAT_CALLER_NO_DEVICE_REPLAY")
    os.environ["PLAT_CALLER_NO_DEVICE_REPLAY"] = "1"                         # the same regions flagged for the host instead
    try:
        flagged, sf = _native(work, fasta, 4, False, **over)
    finally:
        if old is None:
            os.environ.pop("PLAT_CALLER_NO_DEVICE_REPLAY", None)
        else:
            os.environ["PLAT_CALLER_NO_DEVICE_REPLAY"] = old
    assert sf["n_regions_stage_b_host"] == 3 and sf["n_regions_dict_replay_device"] == 0 and flagged == dev
    py = io.StringIO()
    caller.callVariantsInRegions(work, fasta, default_options(**over), VCF(["S1"]), py)
    assert dev == py.getvalue()
    multi = [ln for ln in dev.split("\n") if ln and "," in ln.split("\t")[4]]
    assert len(multi) >= 3 or over
