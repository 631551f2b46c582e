"""TEST INFRASTRUCTURE: a CPU stand-in for the device library (the C ABI of include/platypus_mi355x.h implemented with the
parity oracle) so that host logic layered on the ABI can run in the CPU test suite.  Never imported by the package."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
FAKE_LIB = os.path.join(HERE, "libplat_fake.so")


def build():
    from oracle import oracle as orc
    orc.build()
    src = os.path.join(HERE, "fake_device.c")
    if not os.path.exists(FAKE_LIB) or os.path.getmtime(FAKE_LIB) < max(os.path.getmtime(src), os.path.getmtime(os.path.join(ROOT, "oracle", "liborc.so"))):
        subprocess.run(["gcc", "-O1", "-std=c11", "-fPIC", "-shared", "-fvisibility=hidden", src, "-o", FAKE_LIB,
                        "-L" + os.path.join(ROOT, "oracle"), "-lorc", "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-lm"], check=True)
    return FAKE_LIB


def fake_engine():
    """An Engine whose 'device' is host memory and whose library is the oracle-backed fake (torch CPU tensors as buffers)."""
    import torch
    from platypus_amd import _lib
    from platypus_amd.engine import Engine

    class FakeEngine(Engine):
        def __init__(self):
            self.lib = _lib.bind(C.CDLL(build()))
            self.device = torch.device("cpu")
            ctx = C.c_void_p()
            assert self.lib.plat_ctx_create(0, C.byref(ctx)) == 0
            self.ctx = ctx

        def _stream(self):
            return C.c_void_p(0)

        def _sync(self):
            pass
    return FakeEngine()


def fake_caller_lib():
    """libplat_caller (platypus_amd/csrc/host) linked against the fake device library instead of libplat_mi355x.so."""
    from platypus_amd import fastcaller as F
    build()
    out = os.path.join(HERE, "libplat_caller_fake.so")
    srcs = [os.path.join(F.HOST_SRC, f) for f in sorted(os.listdir(F.HOST_SRC)) if f.endswith((".cpp", ".hpp"))]
    if not os.path.exists(out) or os.path.getmtime(out) < max([os.path.getmtime(f) for f in srcs] + [os.path.getmtime(FAKE_LIB)]):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-ffp-contract=off", "-fvisibility=hidden", os.path.join(F.HOST_SRC, "region_caller.cpp"), "-o", out,
                        "-L" + HERE, "-lplat_fake", "-Wl,-rpath," + HERE], check=True)
    return F._bind(C.CDLL(out))
