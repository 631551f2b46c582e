"""The region loop against the REFERENCE's own region loop (tests/golden/region_cases.json.gz, see tests/region_golden.py): host logic
only -- no GPU here, so the device share comes from tests/fakedev (the C ABI implemented with the parity oracle, test infrastructure).
tests/test_gpu_region_golden.py repeats it on the device."""
import pytest

from platypus_amd import hostapi as H
from tests import region_golden as R

CASES = R.load_cases()


@pytest.fixture(scope="module")
def fake():
    from tests import fakedev
    old = H._engine
    H._engine = fakedev.fake_engine()
    yield fakedev.fake_caller_lib()
    H._engine = old


def test_fixture_covers_the_glue():
    opts = [c["options"] for c in CASES]
    assert len(CASES) >= 30 and sum(len(c["lines"]) for c in CASES) >= 200
    assert sum(o.get("assemble", 0) for o in opts) >= 5 and any(o.get("skipDifficultWindows") for o in opts) and any(o.get("maxVariants") == 3 for o in opts)
    assert any(len(c["sample_names"]) == 3 for c in CASES) and any(c["scenario"].get("empty") for c in CASES) and any(len(c["regions"]) == 2 for c in CASES)
    assert any("Source=Assembler" in ln for c in CASES for ln in c["lines"]) and any(not r["loaded"] for c in CASES for r in c["regions"])
    assert all(not c["skipped_windows"] and not c["errors"] for c in CASES)
    assert sum(c["haplotype_merges"] for c in CASES) >= 10          # mergeHaplotypes (variantcaller.pyx:325-390) merged equal sequences


@pytest.mark.parametrize("ci", range(len(CASES)))
def test_native_region_loop_writes_the_references_records(fake, ci):
    case = CASES[ci]
    got, rlen, failed = R.native_loop_text(case, fake)
    assert got == case["lines"], R.diff(got, case["lines"])
    assert failed == 0 and (rlen == case["rlen_after"] or not any(r["loaded"] for r in case["regions"]))


@pytest.mark.parametrize("ci", range(len(CASES)))
def test_python_region_loop_writes_the_references_records(fake, ci):
    case = CASES[ci]
    got, rlen = R.python_loop_text(case)
    assert got == case["lines"], R.diff(got, case["lines"])
    assert rlen == case["rlen_after"] or not any(r["loaded"] for r in case["regions"])
    if ci % 6 == 0:                                   # the window-by-window shape (one Population walked through the windows, as the reference does)
        got, rlen = R.window_by_window_text(case)
        assert got == case["lines"], R.diff(got, case["lines"])
