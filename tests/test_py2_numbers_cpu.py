"""The native record layer rounds and prints floats as Python 2 does (round(x, 2): ties away from zero on the exact binary value;
str(float): "%.12g").  Its fast paths (one fused multiply-add instead of a trip through text; digits written from the integer n when the
double is the one nearest to n/100) are checked here against the text-based definitions on a few million values."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fast_round2_and_str_equal_the_text_based_definitions(tmp_path):
    exe = str(tmp_path / "py2_numbers_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "platypus_amd", "csrc", "host"),
                    os.path.join(ROOT, "tests", "native", "py2_numbers_check.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe, "150000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "round2 differs 0, str differs 0" in r.stdout
