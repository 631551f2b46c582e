"""tools/native_soak.py on tests/fakedev (the C ABI implemented with the parity oracle) instead of the GPU: the host logic of the
native region loop against the Python region loop, on the CPU.  Lives under tests/ because the fake device calls the oracle.
usage: python tests/soak/native_soak_fake.py [seconds]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from platypus_amd import hostapi as H          # noqa: E402
from tests import fakedev                      # noqa: E402
import native_soak                              # noqa: E402

if __name__ == "__main__":
    H._engine = fakedev.fake_engine()
    native_soak.main(lib=fakedev.fake_caller_lib())
