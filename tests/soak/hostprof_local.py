"""TEST INFRASTRUCTURE (it runs on tests/fakedev, the oracle-backed stand-in: not under tools/).  Host cycle profile of the native region loop WITHOUT a GPU: libplat_caller's sources built with -DPLAT_HOSTPROF (per-thread cycle counters of
the named scopes, csrc/host/caller_common.hpp) against the CPU stand-in of the device library (tests/fakedev: the C ABI on the parity oracle), one
worker over N synthetic config-4 regions.  What it is good for: the host stages that do not depend on who computed the numbers -- INFO / FILTER
arithmetic, record text (text.*), the read-statistics / genotype-call inputs (s6.*), window and Variant objects -- in kcycles per region, before and
after a change to csrc/host; the device-side scopes (s1, s4.runWindows, s6.launch) are the stand-in's own CPU time and mean nothing here, and stage
B runs on the host (the stand-in has no plat_stage_b_batch).  The GPU box's profile of the real job is tools/hostprof.sh.

    python tests/soak/hostprof_local.py [regions=8]          # prints the [prof] lines of the last pass and a hash of the record text
"""
import ctypes as C
import hashlib
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


class OneRank:
    world, rank, cpus, dist, coll_device, dev_index = 1, 0, os.cpu_count() or 1, None, None, 0

    def barrier(self):
        pass

    def gather(self, t):
        return [t]

    def describe(self):
        return {}


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    os.environ["PLAT_CALLER_TRACE"] = "1"
    from tests import fakedev
    from platypus_amd import fastcaller as F
    import bench_other as B
    fakedev.build()
    out = os.path.join(tempfile.gettempdir(), "libplat_caller_hostprof.so")
    subprocess.run(["g++", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread", "-ffp-contract=off", "-fvisibility=hidden", "-DPLAT_HOSTPROF",
                    os.path.join(F.HOST_SRC, "region_caller.cpp"), "-o", out, "-L" + fakedev.HERE, "-lplat_fake", "-Wl,-rpath," + fakedev.HERE], check=True)
    lib = F._bind(C.CDLL(out))
    r = B.config4(0, range(n), 100000, 1, n, repeats=2, rk=OneRank(), lib=lib, pin=False, resident=False)
    print("regions %d windows %d records %d; record text sha256 %s" % (n, r["windows"], r["records"], hashlib.sha256(bytes(memoryview(r["text"]))).hexdigest()[:16]))


if __name__ == "__main__":
    main()
