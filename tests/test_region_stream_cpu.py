"""plat_call_regions_stream (regions loaded on demand into a bounded set of slots -- the reference's own shape: loadBAMData for one region
at a time, variantcaller.pyx:935-1012), read tables handed over at one byte per base (PLAT_READS_PACKED) and the native region source of
the config-4 benchmark (tools/synth): whatever way the reads arrive, the record text is the text of plat_call_regions on the same reads,
which tests/test_native_caller_cpu.py pins against the Python region loop.  CPU: on tests/fakedev (test infrastructure)."""
import io

import numpy as np
import pytest

from platypus_amd import caller, fastcaller as F, hostapi as H, synth
from platypus_amd._lib import PlatypusDeviceError
from platypus_amd.options import default_options
from platypus_amd.vcfrecords import VCF


@pytest.fixture(scope="module")
def fake():
    from tests import fakedev
    old = H._engine
    H._engine = fakedev.fake_engine()
    yield fakedev.fake_caller_lib()
    H._engine = old


def _regions(n=7, **kw):
    kw = dict(dict(region_len=4000, snp_rate=3e-3, indel_rate=1e-3, n_samples=2, read_len=150, depth=25), **kw)
    return [synth.config4_region_arrays(300 + i, **kw) for i in range(n)]


def test_packed_read_tables_give_the_text_of_ascii_tables(fake):
    """One byte per base (2-bit base | quality << 2) + exceptions for N / IUPAC bases and qualities above 63: expanded on the device,
    every stage downstream sees the ASCII table; the host's own look at read bases (inserted bases of candidates) decodes them too."""
    regs = _regions()
    for r in regs[:4]:
        for s in r["samples"]:
            s["seq"][::997] = ord("N"); s["seq"][3::1409] = ord("R"); s["qual"][5::613] = 90; s["qual"][7::811] = 64
    names = ["S1", "S2"]
    nc = F.NativeCaller(0, 2, 2, lib=fake)
    a = nc.call_regions([F.region_from_arrays(r) for r in regs], names, default_options())
    bytes_ascii = nc.stats["input_bytes"]
    b = nc.call_regions([F.region_from_arrays(r, packed=True) for r in regs], names, default_options())
    assert a == b and a.count("\n") > 60
    assert nc.stats["input_bytes"] < 0.52 * bytes_ascii
    # the host tally (a scan whose records overflow the kernel's table takes it) reads the inserted bases on the host
    import os
    os.environ["PLAT_CALLER_HOST_TALLY"] = "1"
    try:
        c = nc.call_regions([F.region_from_arrays(r, packed=True) for r in regs], names, default_options())
    finally:
        os.environ.pop("PLAT_CALLER_HOST_TALLY")
    assert c == a
    nc.close()


def test_stream_of_regions_gives_the_text_of_the_region_list(fake):
    regs = _regions(9)
    regs[4]["samples"] = [dict(s, **{k: s[k][:0] for k in ("pos", "end", "mapq", "flags", "mate_pos")}, seq=s["seq"][:0], qual=s["qual"][:0],
                               off=s["off"][:1], cigar=s["cigar"][:0], cig_off=s["cig_off"][:1]) for s in regs[4]["samples"]]   # a region without reads
    regs[5] = synth.config4_region_arrays(305, region_len=4000, snp_rate=3e-3, indel_rate=1e-3, n_samples=2, read_len=100, depth=25)   # rlen changes
    names = ["S1", "S2"]
    rr = [F.region_from_arrays(r, packed=(i % 2 == 0)) for i, r in enumerate(regs)]
    o1, o2 = default_options(), default_options()
    for workers, per, slots, loaders in ((2, 2, 6, 2), (3, 1, 4, 3), (1, 4, 8, 1)):
        nc = F.NativeCaller(0, workers, per, lib=fake)
        want = nc.call_regions(rr, names, o1)
        order = []

        def load(index, slot, out):
            assert 0 <= slot < slots
            order.append(index)
            rr[index].fill(out, 2)
            return 0
        got = nc.call_stream(len(rr), load, None, names, o2, n_slots=slots, n_loaders=loaders)
        assert got == want and o1.rlen == o2.rlen and sorted(order) == list(range(len(rr)))
        assert nc.stats["n_regions"] == len(rr) and nc.stats["seconds_load"] > 0
        # too few slots for the workers is refused, not deadlocked
        if workers == 2:
            with pytest.raises(PlatypusDeviceError) as e:
                nc.call_stream(len(rr), load, None, names, default_options(), n_slots=3, n_loaders=1)
            assert e.value.code == -1
        nc.close()


def test_a_failing_region_source_ends_the_call(fake):
    regs = _regions(6)
    rr = [F.region_from_arrays(r) for r in regs]
    nc = F.NativeCaller(0, 2, 1, lib=fake)

    def load(index, slot, out):
        if index == 3:
            return -9
        rr[index].fill(out, 2)
        return 0
    with pytest.raises(PlatypusDeviceError) as e:
        nc.call_stream(len(rr), load, None, ["A", "B"], default_options(), n_slots=4, n_loaders=2)
    assert e.value.code == -9 and "region 3" in str(e.value)
    # and the caller is usable afterwards
    assert nc.call_stream(3, load, None, ["A", "B"], default_options(), n_slots=4, n_loaders=2).count("\n") > 10
    nc.close()


def test_native_region_source_of_the_benchmark(fake):
    """tools/synth: regions from seed (+) index.  A region does not depend on the slot, on the other regions of the list or on the encoding;
    streamed through the region loop it gives the text of the same reads handed over as arrays, and of the Python region loop."""
    from tools.synth import source
    kw = dict(region_len=3000, flank=600, n_samples=2, depth=25, read_len=100, snp_rate=3e-3, indel_rate=1.5e-3)
    ids = [11, 5, 8, 2, 9]
    packed, plain = source.RegionSource(ids, 7, packed=True, pin=False, **kw), source.RegionSource(ids[::-1], 2, packed=False, pin=False, **kw)
    regs = []
    for k, g in enumerate(ids):
        a = F.arrays_from_region_struct(packed.region(k, k % 7))
        b = F.arrays_from_region_struct(plain.region(len(ids) - 1 - k, k % 2))
        sa, sb = a["samples"](2), b["samples"](2)
        assert a["chrom"] == b["chrom"] == "r%d" % g and np.array_equal(a["ref"], b["ref"])
        for x, y in zip(sa, sb):
            assert all(np.array_equal(x["reads"][f], y["reads"][f]) for f in x["reads"])
            t = x["reads"]
            assert np.all(np.diff(t["pos"]) >= 0) and len(t["pos"]) == 750 and len(x["bad"]["pos"]) == 0
        regs.append(dict(chrom=a["chrom"], start=a["start"], end=a["end"], ref=a["ref"], samples=[x["reads"] for x in sa]))
    assert packed.planted > 20
    names = ["S1", "S2"]
    nc = F.NativeCaller(0, 2, 2, lib=fake)
    want = nc.call_regions([F.region_from_arrays(r) for r in regs], names, default_options())
    got = nc.call_stream(len(ids), packed.load_fn, packed.h, names, default_options(), n_slots=7, n_loaders=2)
    assert got == want and want.count("\n") > 25
    nc.close()
    # the Python region loop on the same reads
    fasta = H.FastaFile({r["chrom"]: r["ref"].tobytes() for r in regs})
    work = [(r["chrom"], r["start"], r["end"], [H.bamReadBuffer(F.aligned_reads_from_arrays(s), sample=names[i]) for i, s in enumerate(r["samples"])])
            for r in regs]
    py = io.StringIO()
    caller.callVariantsInRegions(work, fasta, default_options(), VCF(names), py)
    assert py.getvalue() == want
    # CIGARs are what an aligner would report: read bases against the reference along the CIGAR differ only at SNPs / errors
    r0, t = regs[0], regs[0]["samples"][0]
    mism = total = 0
    for k in range(len(t["pos"])):
        rp, qp = int(t["pos"][k]), int(t["off"][k])
        for op, ln in t["cigar"].reshape(-1, 2)[t["cig_off"][k]:t["cig_off"][k + 1]].tolist():
            if op == 0:
                mism += int((t["seq"][qp:qp + ln] != r0["ref"][rp:rp + ln]).sum()); total += ln; rp += ln; qp += ln
            elif op == 1:
                qp += ln
            else:
                rp += ln
        assert qp == t["off"][k + 1] and rp == t["end"][k]
    assert mism / total < 0.01


def test_the_last_round_of_smaller_equal_chunks_changes_nothing(fake):
    """chunkBounds (region_caller.cpp): whole chunks while every worker gets the same number, the rest of the list as ONE more round of
    equal smaller chunks -- 2 workers x 5 regions per chunk on 28 regions = 4 whole chunks + 2 chunks of 4 (not 5 and 3; never a chunk
    LARGER than the caller's regions_per_chunk: n_slots is sized by it).  The text, rlen and the statistics are those of one worker
    walking chunks of 5 to the end; list and stream entry."""
    regs = _regions(28)
    names = ["S1", "S2"]
    rr = [F.region_from_arrays(r, packed=(i % 3 == 0)) for i, r in enumerate(regs)]
    o0 = default_options()
    one = F.NativeCaller(0, 1, 5, lib=fake)
    want = one.call_regions(rr, names, o0)
    nwin = one.stats["n_windows"]
    one.close()
    assert want.count("\n") > 40
    nc = F.NativeCaller(0, 2, 5, lib=fake)
    o1, o2 = default_options(), default_options()
    assert nc.call_regions(rr, names, o1) == want and nc.stats["n_windows"] == nwin and o1.rlen == o0.rlen

    def load(index, slot, out):
        rr[index].fill(out, 2)
        return 0
    assert nc.call_stream(len(rr), load, None, names, o2, n_slots=15, n_loaders=2) == want and o2.rlen == o0.rlen
    nc.close()
