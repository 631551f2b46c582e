"""GPU parity tests of the device assembler (a14-a18) against the golden vectors generated from the reference's
own assembler.pyx and against the oracle on fresh fuzz regions.  Bar: identical variant tuples in identical order."""
import gzip
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from platypus_amd.engine import Engine
    return Engine(0)


def to_region(c):
    return dict(ref=c["ref"].encode(), ref_start=c["refStart"], assem_start=c["assemStart"], assem_end=c["assemEnd"],
                seqs=[s.encode() for s in c["seqs"]], quals=[q.encode("latin1") for q in c["quals"]])


def test_assembler_golden_vectors(eng, golden_dir):
    cases = json.load(gzip.open(os.path.join(golden_dir, "assembler_cases.json.gz"), "rt"))
    for nc in (0, 1):
        sel = [c for c in cases if c["noCycles"] == nc]
        got = eng.assemble([to_region(c) for c in sel], kmer_size=15, min_qual=20, min_weight=40, no_cycles=nc)
        for c, g in zip(sel, got):
            exp = [(p, r.encode(), a.encode()) for p, r, a in c["variants"]]
            assert g == exp


def synth_region(rng, ref_len, nh, L, depth, nvar):
    B = b"ACGT"

    def rnd(n):
        return bytes(rng.choice(list(B), n).tolist())
    ref = bytearray(rnd(ref_len))
    if rng.random() < 0.4:
        p = int(rng.integers(100, ref_len - 200)); ln = int(rng.integers(20, 120)); u = rnd(int(rng.integers(1, 7)))
        ref[p:p + ln] = (u * (ln // len(u) + 1))[:ln]
    if rng.random() < 0.2:
        ref[int(rng.integers(0, ref_len))] = ord("N")
    ref = bytes(ref)
    ref_start = int(rng.integers(0, 100000))
    a0 = ref_start + int(rng.integers(0, ref_len // 3)); a1 = a0 + int(rng.integers(100, 1500))
    donors = []
    for _ in range(nh):
        d = bytearray(ref)
        for _ in range(int(rng.integers(0, nvar + 1))):
            lo, hi = max(50, a0 - ref_start), min(len(d) - 100, a1 - ref_start) + 1
            p = int(rng.integers(lo, hi)) if hi > lo else 60
            t = int(rng.integers(0, 3))
            if t == 0:
                d[p] = B[int(rng.integers(0, 4))]
            elif t == 1:
                d[p:p] = rnd(int(rng.integers(1, 41)))
            else:
                del d[p:p + int(rng.integers(1, 41))]
        donors.append(bytes(d))
    seqs, quals = [], []
    for _ in range(depth * ref_len // L):
        d = donors[int(rng.integers(0, nh))]
        if len(d) <= L:
            continue
        p = int(rng.integers(0, len(d) - L)); s = bytearray(d[p:p + L])
        q = np.clip(rng.normal(35, 5, L), 2, 41).astype(np.uint8)
        lo = rng.random(L) < 0.05; q[lo] = rng.integers(2, 20, int(lo.sum()))
        for e in np.nonzero(rng.random(L) < 0.002)[0]:
            s[e] = B[int(rng.integers(0, 4))]
        if rng.random() < 0.02:
            s[int(rng.integers(0, L))] = ord("N")
        seqs.append(bytes(s)); quals.append(bytes(q.tolist()))
    return dict(ref=ref, ref_start=ref_start, assem_start=a0, assem_end=a1, seqs=seqs, quals=quals)


def test_assembler_fuzz_vs_oracle(eng, oracle):
    rng = np.random.default_rng(31337)
    regions = [synth_region(rng, int(rng.integers(600, 4500)), int(rng.choice([1, 2, 2, 4])),
                            int(rng.choice([100, 150, 250])), int(rng.choice([15, 30])), int(rng.choice([0, 3, 6])))
               for _ in range(60)]
    regions.append(dict(ref=b"ACGT" * 30, ref_start=5, assem_start=0, assem_end=200, seqs=[], quals=[]))     # no reads
    regions.append(dict(ref=b"ACGTACG", ref_start=0, assem_start=0, assem_end=7, seqs=[b"ACGTACGTAC"], quals=[b"\x28" * 10]))  # ref shorter than k
    for k, nc in ((15, 0), (15, 1), (21, 0), (11, 0)):
        got = eng.assemble(regions, kmer_size=k, no_cycles=nc)
        nvar = 0
        for r, g in zip(regions, got):
            exp, _ = oracle.assemble(r["ref"], r["ref_start"], r["assem_start"], r["assem_end"], r["seqs"], r["quals"],
                                     k, 20, 40, nc)
            assert g == exp
            nvar += len(exp)
        assert nvar > 20


def test_assembler_fuzz_edge_paths(eng, oracle):
    """The paths of k_assemble the plain fuzz does not reach: reads longer than one 256-byte window, reference bytes other than
    A/C/G/T (N runs, an IUPAC letter, lower case: successor slots 4..7), quality bytes >= 128 (negative as the reference reads
    them), enough distinct k-mers to overflow the LDS table (global path), k > 31 (global path), a reference longer than the
    LDS cache, and reads that start a few bases apart (every dword alignment of the window loads)."""
    rng = np.random.default_rng(424242)
    regions = []
    for _ in range(10):                                   # long reads, odd reference bytes
        r = synth_region(rng, int(rng.integers(900, 3000)), 2, int(rng.choice([300, 400, 520])), 20, 4)
        ref = bytearray(r["ref"])
        for _ in range(3):
            q = int(rng.integers(0, len(ref) - 10))
            ref[q:q + int(rng.integers(1, 4))] = bytes(rng.choice([ord("N"), ord("R"), ord("a"), ord("n")], 1).tolist()) * 3
        r["ref"] = bytes(ref[:len(r["ref"])])
        regions.append(r)
    for _ in range(4):                                    # quality bytes >= 128 and 0
        r = synth_region(rng, 1500, 2, 150, 25, 3)
        qs = []
        for q in r["quals"]:
            q = bytearray(q)
            for _ in range(int(rng.integers(0, 3))):
                q[int(rng.integers(0, len(q)))] = int(rng.choice([0, 128, 200, 255]))
            qs.append(bytes(q))
        r["quals"] = qs
        regions.append(r)
    for _ in range(3):                                    # many distinct k-mers: reads from elsewhere (no bubble starts, but 40 k nodes)
        r = synth_region(rng, 4000, 2, 150, 30, 3)
        junk = [bytes(rng.choice(list(b"ACGT"), 150).tolist()) for _ in range(400)]
        at = sorted(int(x) for x in rng.integers(0, len(r["seqs"]), len(junk)))
        for j, sq in zip(reversed(at), junk):
            r["seqs"].insert(j, sq); r["quals"].insert(j, bytes([35]) * 150)
        regions.append(r)
    regions.append(synth_region(rng, 9000, 2, 150, 12, 4))     # reference beyond the LDS cache
    for k, nc in ((15, 0), (15, 1), (25, 0), (35, 0)):
        got = eng.assemble(regions, kmer_size=k, no_cycles=nc)
        for r, g in zip(regions, got):
            exp, _ = oracle.assemble(r["ref"], r["ref_start"], r["assem_start"], r["assem_end"], r["seqs"], r["quals"], k, 20, 40, nc)
            assert g == exp


def test_config3_shape_regions(eng, oracle):
    """BASELINE config 3 shape: 4.5 kb reference (1.5 kb tile +- 1.5 kb), 250 bp reads at 30x, 1-3 indels + SNPs."""
    rng = np.random.default_rng(3003)
    regions = [synth_region(rng, 4500, 2, 250, 30, 4) for _ in range(24)]
    got = eng.assemble(regions)
    tot = 0
    for r, g in zip(regions, got):
        exp, _ = oracle.assemble(r["ref"], r["ref_start"], r["assem_start"], r["assem_end"], r["seqs"], r["quals"])
        assert g == exp
        tot += len(exp)
    assert tot > 0


def test_fused_lds_pass_equals_the_three_pass_path(eng, golden_dir):
    """Round 4: on the LDS path the reads' k-mers and AddEdge events are ONE pass (reference nodes numbered first, ids for read-only nodes as
    they appear, first tickets kept as the events go); PLAT_ASM_FUSED=0 is rounds 2-3's insert / number / apply / collect-tickets sequence.
    The same variants in the same order on the reference's golden regions, on fuzz regions with repeats, N runs and low-quality stretches,
    and on config-3 shaped regions; several regions per workgroup, so the slot words a region leaves clean are met by the next one."""
    cases = json.load(gzip.open(os.path.join(golden_dir, "assembler_cases.json.gz"), "rt"))
    regs = [to_region(c) for c in cases if c["noCycles"] == 0] * 3
    rng = np.random.default_rng(4242)
    regs += [synth_region(rng, int(rng.integers(300, 2500)), 2, int(rng.choice([75, 100, 150, 250])), int(rng.integers(5, 40)), int(rng.integers(0, 6))) for _ in range(400)]
    regs += [synth_region(rng, 4500, 2, 250, 30, 4) for _ in range(48)]                      # config-3 shape
    out = {}
    for mode in ("1", "0"):
        os.environ["PLAT_ASM_FUSED"] = mode
        try:
            out[mode] = eng.assemble(regs, kmer_size=15, min_qual=20, min_weight=40, no_cycles=0)
        finally:
            os.environ.pop("PLAT_ASM_FUSED", None)
    assert out["1"] == out["0"] and sum(len(v) for v in out["1"]) > 300


def test_walks_in_the_slice_equal_walks_in_lds(eng, golden_dir):
    """Round 4: after the successors are picked the k-mer table's LDS holds an edge word per node, the bubble walks' stacks and their first
    path elements, and the variants are extracted one finished path per thread.  PLAT_ASM_DEBUG=4 puts the stacks and all but 24 path
    elements in the workgroup's slice of global memory (what a region with hundreds of bubble starts falls back to), 8 extracts the
    variants with one thread (the fallback when the paths' bytes do not fit): the same variants in the same order, on the reference's
    golden regions, fuzz regions and config-3 shaped regions with many sequencing-error branches."""
    cases = json.load(gzip.open(os.path.join(golden_dir, "assembler_cases.json.gz"), "rt"))
    regs = [to_region(c) for c in cases if c["noCycles"] == 0]
    rng = np.random.default_rng(777)
    regs += [synth_region(rng, int(rng.integers(300, 3000)), int(rng.choice([1, 2, 4])), int(rng.choice([100, 150, 250])), int(rng.integers(10, 60)), int(rng.integers(0, 7))) for _ in range(300)]
    regs += [synth_region(rng, 4500, 2, 250, 60, 6) for _ in range(32)]                      # config-3 shape at twice the depth: more error branches
    out = {}
    for mode in ("", "4", "8", "12"):
        if mode:
            os.environ["PLAT_ASM_DEBUG"] = mode
        try:
            out[mode] = eng.assemble(regs, kmer_size=15, min_qual=20, min_weight=40, no_cycles=0)
        finally:
            os.environ.pop("PLAT_ASM_DEBUG", None)
    assert out["4"] == out[""] and out["8"] == out[""] and out["12"] == out[""] and sum(len(v) for v in out[""]) > 300


def test_a_dozen_regions_per_workgroup_fused_equals_three_pass(eng):
    """The fused LDS path leaves the L1 invalidate out of the phase boundaries where only plainly stored words are read back (a workgroup
    barrier is enough inside a CU) and keeps it where words updated by L2 atomics are read by plain loads; the three-pass path keeps it at
    every boundary.  3 000 regions in one launch = a dozen regions after one another on every workgroup, whose slices and LDS still hold
    the region before: the same variants from both paths."""
    rng = np.random.default_rng(90210)
    regs = [synth_region(rng, int(rng.integers(400, 1300)), 2, int(rng.choice([100, 150])), int(rng.integers(8, 30)), int(rng.integers(0, 5))) for _ in range(3000)]
    out = {}
    for mode in ("1", "0"):
        os.environ["PLAT_ASM_FUSED"] = mode
        try:
            out[mode] = eng.assemble(regs, kmer_size=15, min_qual=20, min_weight=40, no_cycles=0)
        finally:
            os.environ.pop("PLAT_ASM_FUSED", None)
    assert out["1"] == out["0"] and sum(len(v) for v in out["1"]) > 2000


def test_async_entry_with_the_callers_sizes_equals_the_entry_that_reads_them_back(eng, golden_dir):
    """plat_assemble_batch_async (the region loop's entry: sizes from the caller, no read-back, no wait) = plat_assemble_batch on the
    reference's goldens; hints that do not cover the batch refuse EVERY tile (PLAT_ERR_BAD_HINTS), loudly."""
    from platypus_amd import _lib
    cases = json.load(gzip.open(os.path.join(golden_dir, "assembler_cases.json.gz"), "rt"))
    for nc in (0, 1):
        sel = [to_region(c) for c in cases if c["noCycles"] == nc]
        want = eng.assemble(sel, no_cycles=nc)
        assert eng.assemble(sel, no_cycles=nc, hints="exact") == want
        assert any(want)
    sel = [to_region(c) for c in cases[:6]]
    ref_len = max(len(r["ref"]) for r in sel)
    nreads = max(len(r["seqs"]) for r in sel)
    npos = max(len(r["ref"]) + 2 + sum(len(q) for q in r["seqs"]) + 2 * len(r["seqs"]) for r in sel)
    assert eng.assemble(sel, hints=(ref_len, nreads, npos)) == eng.assemble(sel)
    for bad in ((ref_len - 1, nreads, npos), (ref_len, nreads - 1, npos), (ref_len, nreads, npos - 1)):
        with pytest.raises(_lib.PlatypusDeviceError) as e:
            eng.assemble(sel, hints=bad)
        assert e.value.code == -10                      # PLAT_ERR_BAD_HINTS


def test_a_slice_kept_clean_across_launches_gives_the_variants_of_a_fresh_context(eng, monkeypatch):
    """Round 5: a workgroup that ends a launch on the fused path without an error leaves the signature of its slice of the scratch behind,
    and the next launch with the same layout skips the stores that establish the clean slot words.  A sequence of launches on ONE context
    that keeps, drops and re-establishes that promise -- fused (k = 15), again (kept), the global path (k = 21: dropped), fused again,
    a launch whose tiles overflow their output room (error: dropped), fused again, the three-pass path (PLAT_ASM_FUSED=0: dropped),
    fused -- gives launch by launch what a context without the promise (PLAT_ASM_NO_KEEP=1) gives."""
    from platypus_amd import _lib
    from platypus_amd.engine import Engine
    rng = np.random.default_rng(5005)
    sets = [[synth_region(rng, 4500, 2, 250, 30, 4) for _ in range(40)] for _ in range(3)]
    plan = [(0, 15, {}), (1, 15, {}), (0, 21, {}), (2, 15, {}), (1, 15, dict(max_vars=1)), (0, 15, {}), (2, 15, dict(env="PLAT_ASM_FUSED")), (1, 15, {}), (1, 15, {})]

    def run(e):
        out = []
        for which, k, kw in plan:
            if kw.get("env"):
                monkeypatch.setenv(kw["env"], "0")
            try:
                out.append(e.assemble(sets[which], kmer_size=k, max_vars=kw.get("max_vars", 512)))
            except _lib.PlatypusDeviceError as err:
                out.append(("error", err.code))
            if kw.get("env"):
                monkeypatch.delenv(kw["env"])
        return out
    got = run(Engine(0))
    monkeypatch.setenv("PLAT_ASM_NO_KEEP", "1")
    want = run(Engine(0))
    assert got == want
    assert got[4] == ("error", -8) and sum(len(v) for v in got[0]) > 20 and got[0] == got[5] and got[1] == got[7] == got[8]


def test_a_launch_with_another_layout_makes_every_kept_slice_stale(eng, monkeypatch):
    """Round 6: a slice sits at blockIdx.x * per_block and a launch only rewrites the signatures of the workgroups it
    runs.  A (40 tiles), B (10 larger tiles: a larger per_block, so B's slices 0..9 cover A's slices 10 and up), A again on ONE context:
    the third launch must not trust the signatures A left for workgroups 10..39.  Launch by launch the variants of a context without
    the promise (PLAT_ASM_NO_KEEP=1); several rounds, so that a layout that comes back is seen with stale, not absent, signatures."""
    from platypus_amd.engine import Engine
    rng = np.random.default_rng(6006)
    many = [synth_region(rng, 3000, 2, 150, 30, 4) for _ in range(40)]
    few = [synth_region(rng, 6000, 3, 250, 40, 5) for _ in range(10)]
    plan = [many, few, many, many, few, few, many]

    def run(e):
        return [e.assemble(s, kmer_size=15) for s in plan]
    got = run(Engine(0))
    monkeypatch.setenv("PLAT_ASM_NO_KEEP", "1")
    want = run(Engine(0))
    assert got == want
    assert got[0] == got[2] == got[3] == got[6] and got[1] == got[4] == got[5] and sum(len(v) for v in got[0]) > 20
