"""The region pipeline (candidates -> windows -> haplotypes -> likelihoods / EM / posteriors -> VCF records) on synthetic
config-4 regions: window by window as the reference walks them, and with every device stage batched over all windows of all
regions -- the two must write the same text -- and the planted variants must come back."""
import io

import numpy as np
import pytest

from platypus_amd import caller, hostapi as H, synth
from platypus_amd.options import default_options
from platypus_amd.vcfrecords import VCF

pytestmark = pytest.mark.gpu


def _regions(n, n_samples, **kw):
    regs = [synth.config4_region(i, n_samples=n_samples, **kw) for i in range(n)]
    fasta = H.FastaFile({r["chrom"]: r["ref"] for r in regs})
    names = ["S%d" % (i + 1) for i in range(n_samples)]

    def buffers(r):
        return [H.bamReadBuffer([H.AlignedRead(x["seq"], x["qual"], x["pos"], x["mapq"], x["flag"], end=x["end"], cigarOps=x["cigar"])
                                 for x in reads], sample=names[i]) for i, reads in enumerate(r["samples"])]
    return regs, fasta, names, buffers


def test_region_pipeline_batched_equals_window_by_window_and_finds_the_planted_variants():
    regs, fasta, names, buffers = _regions(4, 2, region_len=2500, snp_rate=4e-3, indel_rate=1.5e-3, read_len=100, depth=40)
    one = io.StringIO()
    for r in regs:
        opts = default_options()
        caller.callVariantsInRegion(r["chrom"], r["start"], r["end"], buffers(r), fasta, opts, VCF(names), one)
    opts = default_options()
    many = io.StringIO()
    n_windows = caller.callVariantsInRegions([(r["chrom"], r["start"], r["end"], buffers(r)) for r in regs], fasta, opts, VCF(names), many)
    assert one.getvalue() == many.getvalue()
    lines = many.getvalue().split("\n")[:-1]
    assert n_windows >= 20 and len(lines) >= 20
    called = {}
    for ln in lines:
        f = ln.split("\t")
        called[(f[0], int(f[1]))] = f
        assert len(f) == 9 + len(names) and f[8] == "GT:GL:GOF:GQ:NR:NV"
    planted = [(r["chrom"], p, rem, add) for r in regs for p, rem, add in r["variants"]]
    snps = [(c, p) for c, p, rem, add in planted if len(rem) == len(add)]
    found = sum((c, p + 1) in called for c, p in snps)
    # a planted variant is absent from both haplotypes of both samples with probability 1/16
    assert found >= 0.85 * len(snps), (found, len(snps))
    assert sum(f[6] == "PASS" for f in called.values()) >= 0.6 * len(called)
    indels = [(c, p) for c, p, rem, add in planted if len(rem) != len(add)]
    near = sum(any(cc == c and abs(pp - p) <= 12 and len(f[3]) != len(f[4].split(",")[0]) for (cc, pp), f in called.items()) for c, p in indels)
    assert near >= 0.6 * len(indels), (near, len(indels))


def test_batched_caller_handles_samples_without_reads_and_empty_regions():
    regs, fasta, names, buffers = _regions(2, 2, region_len=1200, snp_rate=5e-3, indel_rate=0, read_len=100, depth=30)
    bufs = [buffers(r) for r in regs]
    bufs[0][1] = H.bamReadBuffer([], sample=names[1])                        # sample 2 has no data in region 0
    out = io.StringIO()
    opts = default_options()
    caller.callVariantsInRegions([(r["chrom"], r["start"], r["end"], b) for r, b in zip(regs, bufs)] +
                                 [(regs[0]["chrom"], 10, 60, [H.bamReadBuffer([], sample=s) for s in names])], fasta, opts, VCF(names), out)
    lines = out.getvalue().split("\n")[:-1]
    assert lines and all(ln.split("\t")[10].startswith("./.") for ln in lines if ln.startswith(regs[0]["chrom"] + "\t"))


def test_cli_config4_two_ranks_equals_one_rank(tmp_path):
    """`callVariants --synthetic config4:N`: the same VCF records from one process and from two ranks (regions dealt round-robin,
    record lines gathered to rank 0 and merged in (chromosome, position) order) -- the two ranks share the one GPU here."""
    import json, os, socket, subprocess, sys
    one, two = tmp_path / "one.vcf", tmp_path / "two.vcf"
    base = ["callVariants", "--synthetic", "config4:5", "--bufferSize", "4000"]
    r = subprocess.run([sys.executable, "-m", "platypus_amd"] + base + ["--output", str(one)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-800:]
    info = json.loads(r.stdout.strip().splitlines()[-1])
    assert info["records"] > 10
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, PLAT_DIST_BACKEND="gloo")        # both ranks drive GPU 0; RCCL needs one device per rank
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), "-m", "platypus_amd"] + base + ["--output", str(two)], capture_output=True, text=True, env=env)
    assert r.returncode == 0, (r.stderr[-1500:], r.stdout[-500:])
    body = lambda t: [ln for ln in t.split("\n") if not ln.startswith("##")]        # (the header carries the output path and the date)
    a_, b_ = body(one.read_text()), body(two.read_text())
    assert a_ == b_ and a_[0].startswith("#CHROM\tPOS") and one.read_text().startswith("##fileformat=VCFv4.0\n")


def test_assembler_tiles_feed_the_region_pipeline():
    """--assemble=1: a 30-bp deletion whose reads were aligned WITHOUT a gap (plain M CIGARs, as a mapper that soft-clips or
    mis-places them would) is invisible to the CIGAR scan; the assembler tiles of generateVariantsInRegion find it and the
    record carries Source=Assembler.  Batched and window-by-window shapes agree here too."""
    rng = np.random.default_rng(77)
    B = b"ACGT"
    n = 4000
    ref = bytes(rng.choice(list(B), n).astype(np.uint8))
    start, end = 1000, 3000
    dp, dl = 2000, 30                                            # deletion of ref[dp+1 : dp+1+dl]
    alt = ref[:dp + 1] + ref[dp + 1 + dl:]
    reads = []
    for _ in range(500):
        src, shift = (alt, 1) if rng.random() < 0.5 else (ref, 0)
        L = 100
        p0 = int(rng.integers(start - 50, end - 60))
        seq = src[p0:p0 + L] if not shift or p0 + L <= dp else alt[p0 - (dl if p0 > dp else 0):p0 - (dl if p0 > dp else 0) + L]
        q = np.clip(rng.normal(35, 4, L), 20, 41).astype(np.uint8)
        reads.append(H.AlignedRead(seq, bytes(q.tolist()), p0, 60, 3 | (16 if rng.random() < 0.5 else 0)))   # default CIGAR: 100M
    fasta = H.FastaFile({"20": ref})
    texts = []
    for batched in (False, True):
        opts = default_options(assemble=1, getVariantsFromBAMs=0)
        buf = [H.bamReadBuffer(reads, sample="S1")]
        out = io.StringIO()
        if batched:
            caller.callVariantsInRegions([("20", start, end, buf)], fasta, opts, VCF(["S1"]), out)
        else:
            caller.callVariantsInRegion("20", start, end, buf, fasta, opts, VCF(["S1"]), out)
        texts.append(out.getvalue())
    assert texts[0] == texts[1] and texts[0]
    recs = [ln.split("\t") for ln in texts[0].split("\n")[:-1]]
    dels = [f for f in recs if len(f[3]) - len(f[4]) == dl]
    assert dels and all("Source=Assembler" in f[7] for f in dels) and abs(int(dels[0][1]) - (dp + 1)) <= 30


def test_called_genotypes_agree_with_the_planted_donors():
    """Beyond self-consistency: at 40x the genotype written for a planted SNP (0/1 vs 1/1, or no call / 0/0 when the donor does
    not carry it) equals the donor's number of alternative alleles for almost every (variant, sample)."""
    regs, fasta, names, buffers = _regions(3, 2, region_len=3000, snp_rate=3e-3, indel_rate=0, read_len=100, depth=40)
    out = io.StringIO()
    caller.callVariantsInRegions([(r["chrom"], r["start"], r["end"], buffers(r)) for r in regs], fasta, default_options(), VCF(names), out)
    recs = {}
    for ln in out.getvalue().split("\n")[:-1]:
        f = ln.split("\t")
        recs[(f[0], int(f[1]) - 1, f[3], f[4])] = f
    agree = total = 0
    for r in regs:
        for k, (p, rem, add) in enumerate(r["variants"]):
            f = recs.get((r["chrom"], p, rem.decode(), add.decode()))
            for i in range(len(names)):
                want = r["truth"][i][k]
                if f is None:
                    got = 0
                else:
                    gt = f[9 + i].split(":")[0]
                    got = 0 if "." in gt else sum(int(a) != 0 for a in gt.split("/"))
                total += 1
                agree += got == want
    assert total > 30 and agree >= 0.95 * total, (agree, total)


def test_batched_caller_uses_each_regions_own_read_length():
    """options.rlen follows the longest read of each region (variantcaller.pyx:476-488) and is what its indels are
    left-normalised with, its windows are merged with and its haplotypes are padded with; a region without reads keeps the value
    of the region before it.  Regions with 150 / 76 / no / 100 bp reads: the batched shape must write what the sequence of
    per-region calls with ONE options object writes."""
    specs = [dict(read_len=150), dict(read_len=76), None, dict(read_len=100)]
    regs, names = [], ["S1"]
    for i, kw in enumerate(specs):
        r = synth.config4_region(40 + i, n_samples=1, region_len=2200, snp_rate=5e-3, indel_rate=2e-3, depth=35, **(kw or dict(read_len=100)))
        if kw is None:
            r["samples"] = [[]]
        regs.append(r)
    fasta = H.FastaFile({r["chrom"]: r["ref"] for r in regs})
    buffers = lambda r: [H.bamReadBuffer([H.AlignedRead(x["seq"], x["qual"], x["pos"], x["mapq"], x["flag"], end=x["end"], cigarOps=x["cigar"])
                                          for x in r["samples"][0]], sample="S1")]
    opts = default_options()
    one = io.StringIO()
    seen = []
    for r in regs:
        caller.callVariantsInRegion(r["chrom"], r["start"], r["end"], buffers(r), fasta, opts, VCF(names), one)
        seen.append(opts.rlen)
    assert len(set(seen)) >= 3 and seen[2] == seen[1]                         # the empty region kept the value of the one before
    opts2 = default_options()
    many = io.StringIO()
    caller.callVariantsInRegions([(r["chrom"], r["start"], r["end"], buffers(r)) for r in regs], fasta, opts2, VCF(names), many)
    assert one.getvalue() == many.getvalue() and one.getvalue().count("\n") > 20
    assert opts2.rlen == seen[-1]


def _array_regions(n, ns, **kw):
    from platypus_amd import fastcaller as F
    regs = [synth.config4_region_arrays(500 + i, n_samples=ns, **kw) for i in range(n)]
    names = ["S%d" % (i + 1) for i in range(ns)]
    fasta = H.FastaFile({r["chrom"]: r["ref"].tobytes() for r in regs})
    work = [(r["chrom"], r["start"], r["end"], [H.bamReadBuffer(F.aligned_reads_from_arrays(s), sample=names[i]) for i, s in enumerate(r["samples"])])
            for r in regs]
    return regs, names, fasta, work


def test_native_region_loop_equals_python_region_loop_on_the_device():
    """libplat_caller.so (host threads + batched device stages) writes the text platypus_amd.caller writes: 8 regions x 20 kb, two
    samples, indels, several chunks in flight on several workers."""
    from platypus_amd import fastcaller as F
    regs, names, fasta, work = _array_regions(8, 2, region_len=20000, snp_rate=2e-3, indel_rate=4e-4, read_len=150, depth=30)
    o1, o2 = default_options(), default_options()
    py = io.StringIO()
    nw = caller.callVariantsInRegions(work, fasta, o1, VCF(names), py)
    nc = F.NativeCaller(0, 4, 2)
    txt = nc.call_regions([F.region_from_arrays(r) for r in regs], names, o2)
    assert txt == py.getvalue() and nc.stats["n_windows"] == nw and txt.count("\n") > 200
    # and again from the same caller object (scratch buffers re-used), one region per chunk
    nc2 = F.NativeCaller(0, 3, 1)
    assert nc2.call_regions([F.region_from_arrays(r) for r in regs], names, default_options()) == txt
    assert nc.call_regions([F.region_from_arrays(r) for r in regs[:3]], names, default_options()) == "".join(
        ln + "\n" for ln in txt.split("\n")[:-1] if ln.split("\t")[0] in {r["chrom"] for r in regs[:3]})


def test_native_region_loop_greedy_windows_and_read_classes_on_the_device():
    """Dense variants (greedy haplotype filter rounds on the device), badReads and brokenMates."""
    from platypus_amd import fastcaller as F
    rng = np.random.default_rng(4)
    regs = [synth.config4_region(600 + i, n_samples=2, region_len=2500, snp_rate=6e-2, indel_rate=8e-3, read_len=100, depth=30) for i in range(3)]
    names = ["A", "B"]
    fasta = H.FastaFile({r["chrom"]: r["ref"] for r in regs})
    work = []
    for r in regs:
        bufs = []
        for i, reads in enumerate(r["samples"]):
            good, bad, broken = [], [], []
            for x in reads:
                a = H.AlignedRead(x["seq"], x["qual"], x["pos"], x["mapq"], x["flag"], end=x["end"], cigarOps=x["cigar"])
                u = rng.random()
                if u < 0.06:
                    a.mapq = 5; a.bitFlag |= 512; bad.append(a)
                elif u < 0.1:
                    a.matePos = a.pos + int(rng.integers(-300, 300)); broken.append(a)
                else:
                    good.append(a)
            bufs.append(H.bamReadBuffer(good, bad, broken, sample=names[i]))
        work.append((r["chrom"], r["start"], r["end"], bufs))
    py = io.StringIO()
    caller.callVariantsInRegions(work, fasta, default_options(), VCF(names), py)
    nc = F.NativeCaller(0, 2, 2)
    txt = nc.call_regions([F.RegionReads.from_buffers(c, s, e, fasta, b) for c, s, e, b in work], names, default_options())
    assert txt == py.getvalue() and nc.stats["n_windows_greedy"] > 10


def test_fake_device_of_the_cpu_suite_agrees_with_the_device():
    """tests/fakedev (the C ABI implemented with the oracle, what the CPU suite runs the region loop on) gives the text the HIP
    library gives -- the whole device share of the region pipeline against the oracle, through the records."""
    import ctypes as C
    from platypus_amd import fastcaller as F
    from tests import fakedev
    regs, names, fasta, work = _array_regions(3, 2, region_len=5000, snp_rate=4e-3, indel_rate=1e-3, read_len=150, depth=30)
    nc = F.NativeCaller(0, 2, 2)
    real = nc.call_regions([F.region_from_arrays(r) for r in regs], names, default_options())
    nf = F.NativeCaller(0, 2, 2, lib=fakedev.fake_caller_lib())
    fake = nf.call_regions([F.region_from_arrays(r) for r in regs], names, default_options())
    assert real == fake and real.count("\n") > 30


def test_unpack_reads_on_the_device():
    """plat_unpack_reads: one byte per base (2-bit base | quality << 2) -> ASCII bases + raw qualities, exceptions patched; every
    alignment of source and destination (the fast path moves 16 bytes per lane)."""
    import torch
    from platypus_amd import _lib
    lib = _lib.load()
    eng = H.get_engine()
    rng = np.random.default_rng(5)
    for n, shift in ((1, 0), (15, 3), (4096, 0), (100003, 5), (1 << 20, 16)):
        packed = rng.integers(0, 256, n + 64, dtype=np.uint8)
        nex = min(n, 37)
        ex = np.sort(rng.choice(n, nex, replace=False)).astype(np.int64)
        eb, eq = rng.choice(np.frombuffer(b"NRYKM", dtype=np.uint8), nex), rng.integers(64, 94, nex, dtype=np.uint8)
        d_in = torch.from_numpy(packed).cuda()
        d_seq, d_qual = torch.zeros(n + 64, dtype=torch.uint8, device="cuda"), torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
        d_ex, d_eb, d_eq = torch.from_numpy(ex).cuda(), torch.from_numpy(eb).cuda(), torch.from_numpy(eq).cuda()
        rc = lib.plat_unpack_reads(eng.ctx, n, d_in.data_ptr() + shift, d_seq.data_ptr() + shift, d_qual.data_ptr() + shift, nex, d_ex.data_ptr(),
                                   d_eb.data_ptr(), d_eq.data_ptr(), None)
        assert rc == 0
        torch.cuda.synchronize()
        want_s = np.frombuffer(b"ACTG", dtype=np.uint8)[packed[shift:shift + n] & 3].copy()
        want_q = (packed[shift:shift + n] >> 2).copy()
        want_s[ex], want_q[ex] = eb, eq
        got_s, got_q = d_seq.cpu().numpy(), d_qual.cpu().numpy()
        assert np.array_equal(got_s[shift:shift + n], want_s) and np.array_equal(got_q[shift:shift + n], want_q)
        assert not got_s[:shift].any() and not got_s[shift + n:].any() and not got_q[shift + n:].any()      # nothing outside [0, n)


def test_streamed_packed_regions_equal_the_region_list_on_the_device():
    """The shape the config-4 benchmark runs: regions generated on demand by the native source (tools/synth) into pinned slots, one
    byte per base over the link, through plat_call_regions_stream -- the text of the same reads handed over as ASCII arrays in one list."""
    from platypus_amd import fastcaller as F
    from tools.synth import source
    kw = dict(region_len=20000, n_samples=2, depth=30, read_len=150, snp_rate=2e-3, indel_rate=4e-4)
    ids = list(range(40, 52))
    src = source.RegionSource(ids, 14, packed=True, pin=True, **kw)
    regs = []
    for k in range(len(ids)):
        a = F.arrays_from_region_struct(src.region(k, 0))
        regs.append(dict(chrom=a["chrom"], start=a["start"], end=a["end"], ref=a["ref"], samples=[x["reads"] for x in a["samples"](2)]))
    names = ["S1", "S2"]
    nc = F.NativeCaller(0, 4, 2)
    want = nc.call_regions([F.region_from_arrays(r) for r in regs], names, default_options())
    got = nc.call_stream(len(ids), src.load_fn, src.h, names, default_options(), n_slots=14, n_loaders=3)
    assert got == want and want.count("\n") > 300
    st = nc.stats
    assert st["input_bytes"] == st["n_reads"] * 150 and st["seconds_load"] > 0
    assert nc.call_regions([F.region_from_arrays(r, packed=True, pin=True) for r in regs], names, default_options()) == want


def test_native_assembler_and_reference_calls_equal_the_python_loop_on_the_device():
    """--assemble=1 and --outputRefCalls=1 in the native region loop (tiles of a chunk in one plat_assemble_batch on reads gathered from
    the chunk table; REFCALL blocks, flat-prior posteriors) write the text of the Python region loop; and BASELINE config 3's regions
    (250 bp reads, indels up to 60 bases) come through end to end with the planted indels called."""
    from platypus_amd import fastcaller as F
    regs, names, fasta, work = _array_regions(4, 2, region_len=6000, snp_rate=2e-3, indel_rate=1.5e-3, read_len=150, depth=30)
    for opt in (dict(assemble=1), dict(outputRefCalls=1, refCallBlockSize=400), dict(assemble=1, outputRefCalls=1, getVariantsFromBAMs=0)):
        o1, o2 = default_options(**opt), default_options(**opt)
        py = io.StringIO()
        caller.callVariantsInRegions(_array_regions(4, 2, region_len=6000, snp_rate=2e-3, indel_rate=1.5e-3, read_len=150, depth=30)[3], fasta, o1, VCF(names), py)
        nc = F.NativeCaller(0, 2, 2)
        txt = nc.call_regions([F.region_from_arrays(r, packed=True) for r in regs], names, o2)
        assert txt == py.getvalue() and txt.count("\n") > 30, opt
        st = nc.stats
        if opt.get("assemble"):
            assert st["n_assembly_tiles"] == 4 * 8 and st["n_assembler_variants"] > 20 and "Assembler" in txt
        if opt.get("outputRefCalls"):
            assert st["n_refcall_records"] == txt.count("\tREFCALL\t") > 30
    from tools import bench_other
    r = bench_other.config3_end_to_end(0, 64)
    lines = r["text"].split("\n")[:-1]
    assert r["tiles"] == 128 and r["assembler_variants"] > 100 and r["windows"] > 80 and r["pairs"] > 10000
    indels = [ln for ln in lines if len(ln.split("\t")[3]) != len(ln.split("\t")[4])]
    n_asm = sum("Assembler" in ln for ln in indels)
    longest = max(abs(len(ln.split("\t")[3]) - len(ln.split("\t")[4].split(",")[0])) for ln in indels)
    assert len(indels) > 40 and n_asm >= 15, (len(lines), len(indels), n_asm, longest)
    assert longest >= 20, (len(lines), len(indels), n_asm, longest)          # indels far beyond what a 250 bp read's CIGAR shows reliably


@pytest.mark.gpu
def test_sites_with_two_alleles_take_the_python2_dictionary_order_on_the_device():
    """The native loop fetches a scan's records from the device only for a region whose candidates hold a pair that compares equal
    (tests/test_native_caller_cpu.py builds such regions), replays the reference's Python-2 candidate dictionaries for it and ends
    with the text of the Python loop -- also where the pair's order reaches the text (maxVariants = 1)."""
    from platypus_amd import fastcaller as F
    from tests.test_native_caller_cpu import two_allele_regions, _work
    regs, n_sites = two_allele_regions()
    names = ["A", "B"]
    fasta, work = _work(regs, names)
    for kw in (dict(), dict(maxVariants=1)):
        py = io.StringIO()
        caller.callVariantsInRegions(work, fasta, default_options(**kw), VCF(names), py)
        nc = F.NativeCaller(0, 2, 2)
        txt = nc.call_regions([F.RegionReads.from_buffers(c, s, e, fasta, b) for c, s, e, b in work], names, default_options(**kw))
        nc.close()
        assert txt == py.getvalue() and txt.count("\n") > 20


@pytest.mark.gpu
def test_more_distinct_candidates_than_the_device_table_holds_fall_back_to_the_host_tally():
    """A scan with more distinct candidate records than k_candidates_merge's table takes (6 144: here ~15 000, reads with 2 % errors) is
    reported by the device and tallied on the host instead; reads with more candidates than their slice of the record array make the
    scan run again with room for them.  The text is the Python loop's either way."""
    from platypus_amd import fastcaller as F
    regs = [synth.config4_region(950 + i, n_samples=1, region_len=[25000, 3000][i], snp_rate=2e-3, indel_rate=4e-4, read_len=100, depth=30,
                                 err=[2e-2, 1e-3][i]) for i in range(2)]
    names = ["S1"]
    fasta = H.FastaFile({r["chrom"]: r["ref"] for r in regs})
    work = []
    for r in regs:
        rs = [H.AlignedRead(x["seq"], x["qual"], x["pos"], x["mapq"], x["flag"], end=x["end"], cigarOps=x["cigar"]) for x in r["samples"][0]]
        work.append((r["chrom"], r["start"], r["end"], [H.bamReadBuffer(rs, [], [], sample="S1")]))
    py = io.StringIO()
    caller.callVariantsInRegions(work, fasta, default_options(), VCF(names), py)
    nc = F.NativeCaller(0, 2, 2)
    txt = nc.call_regions([F.RegionReads.from_buffers(c, s, e, fasta, b) for c, s, e, b in work], names, default_options())
    assert nc.stats["n_candidate_records"] > 9000                          # (nearly all of them distinct: sequencing errors)
    nc.close()
    assert txt == py.getvalue() and txt.count("\n") > 30


def test_regions_resident_in_hbm_give_the_text_of_streamed_regions():
    """plat_read_table.dev_seq: a region list generated once and mirrored in HBM (tools/synth resident mode: the chunk tables are built
    with device-to-device copies, no read byte crosses the link) gives the record text of the same regions generated and uploaded on demand."""
    import torch
    from platypus_amd import fastcaller as F
    from tools.synth import source
    idx = list(range(12))
    texts, moved = [], []
    for resident in (False, True):
        src = source.RegionSource(idx, len(idx) if resident else 16, region_len=20000, pin=not resident)
        if resident:
            src.make_resident(torch.device("cuda", 0), threads=4)
        nc = F.NativeCaller(0, 3, 2)
        try:
            texts.append(nc.call_stream(len(idx), src.load_fn, src.h, ["S1"], default_options(), 16, 2))
            moved.append(nc.stats["input_bytes"])
        finally:
            nc.close()
            src.close()
    assert texts[0] == texts[1] and texts[0].count("\n") > 100
    assert moved[0] > 12 * 20000 * 30 * 0.9 and moved[1] == 0
