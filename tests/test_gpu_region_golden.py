"""The region loop ON THE DEVICE against the reference's own region loop: reads of 39 small processes (1-3 samples, --assemble 0/1,
windows with more variants than maxVariants, skipDifficultWindows, greedy haplotype rounds, empty samples and regions, option variants)
-> the record lines the reference's callVariantsInRegion text wrote (tests/golden/region_cases.json.gz, tests/region_golden.py).  Both
shapes of the product -- libplat_caller.so (plat_call_regions) and platypus_amd.caller -- through libplat_mi355x.so."""
import pytest

from tests import region_golden as R

pytestmark = pytest.mark.gpu
CASES = R.load_cases()


@pytest.mark.parametrize("ci", range(len(CASES)))
def test_plat_call_regions_writes_the_references_records(ci):
    case = CASES[ci]
    got, rlen, failed = R.native_loop_text(case)
    assert got == case["lines"], R.diff(got, case["lines"])
    assert failed == 0 and (rlen == case["rlen_after"] or not any(r["loaded"] for r in case["regions"]))


def test_python_region_loop_writes_the_references_records():
    for case in CASES:
        got, rlen = R.python_loop_text(case)
        assert got == case["lines"], (case["scenario"], case["options"], R.diff(got, case["lines"]))


def test_native_loop_in_one_call_over_many_workers():
    """All single-region default-option cases as ONE region list through 4 workers, 3 regions per chunk: the text is the concatenation."""
    import io
    from platypus_amd import fastcaller as F
    from platypus_amd.options import default_options
    sel = [c for c in CASES if not c["options"] and len(c["regions"]) == 1 and len(c["sample_names"]) == 1 and c["regions"][0]["samples"][0]["reads"]]
    assert len(sel) >= 3                       # (options.rlen follows the longest read of each region in list order: the cases do not interact)
    # (one contig name per case: the regions of one call share a FastaFile)
    from platypus_amd import hostapi as H
    fasta = H.FastaFile({"c%d" % k: c["ref"].encode() for k, c in enumerate(sel)})
    regs, want = [], []
    for k, c in enumerate(sel):
        _, work, _ = R.case_work(c)
        ch, s, e, b = work[0]
        regs.append(F.RegionReads.from_buffers("c%d" % k, s, e, fasta, b))
        want += [ln.replace("20\t", "c%d\t" % k, 1) for ln in c["lines"]]
    nc = F.NativeCaller(0, 4, 3)
    try:
        txt = nc.call_regions(regs, ["S1"], default_options())
    finally:
        nc.close()
    assert txt.split("\n")[:-1] == want
