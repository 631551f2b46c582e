"""The reference-side binding compiles: bindings/cplat.pxd (the Cython declarations a Platypus maintainer adds next to
chaplotype.pxd) is built with Cython 3 against include/platypus_mi355x.h, linked to libplat_mi355x.so and
imported; every function and struct of the header is declared in it."""
import importlib.util
import os
import re
import subprocess
import sys
import sysconfig

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pxd_declares_the_whole_header():
    hdr = open(os.path.join(ROOT, "include", "platypus_mi355x.h")).read()
    pxd = open(os.path.join(ROOT, "bindings", "cplat.pxd")).read()
    funcs = set(re.findall(r"\b(plat_[a-z0-9_]+)\s*\(", hdr)) - {"plat_ctx"}
    assert funcs and all(re.search(r"\b%s\(" % f, pxd) for f in funcs), [f for f in funcs if not re.search(r"\b%s\(" % f, pxd)]
    structs = set(re.findall(r"typedef struct (plat_[a-z_]+) \{", hdr))
    assert structs and all(("ctypedef struct %s:" % s) in pxd for s in structs)
    assert not any("..." in ln for ln in pxd.split("\n") if not ln.lstrip().startswith("#"))      # no elided signatures


def test_cython_binding_compiles_links_and_runs(tmp_path):
    pytest.importorskip("Cython")
    from platypus_amd import _lib
    _lib.build()
    bind = os.path.join(ROOT, "bindings")
    c_file = tmp_path / "plat_binding_check.c"
    subprocess.check_call([sys.executable, "-m", "cython", "-3", "-I", bind, os.path.join(bind, "plat_binding_check.pyx"), "-o", str(c_file)])
    ext = tmp_path / ("plat_binding_check" + sysconfig.get_config_var("EXT_SUFFIX"))
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-O1", "-I" + sysconfig.get_paths()["include"], "-I" + os.path.join(ROOT, "include"),
                           str(c_file), "-o", str(ext), "-L" + libdir, "-lplat_mi355x", "-Wl,-rpath," + libdir])
    _lib.load()                                    # (the HIP runtime torch ships, first: see _lib.load)
    spec = importlib.util.spec_from_file_location("plat_binding_check", str(ext))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.abi_version() == (_lib.PLAT_ABI_VERSION, _lib.PLAT_ABI_VERSION, _lib.PLAT_BLOB_PAD)
    assert mod.strerror(-4).startswith("haplotype is too long")
    import ctypes as C
    d = mod.describe_window_batch(3, 7, 11)
    assert d["size"] == C.sizeof(_lib.WindowBatch) and d["hints"] == C.sizeof(_lib.BatchHints) and d["stats"] == C.sizeof(_lib.AlignStats)
    assert d["n"] == (3, 7, 11) and d["pairs"] == 77 and d["null_batch_rc"] == -1          # PLAT_ERR_INVALID for a NULL context
    rc, n = mod.device_count()
    if rc != 0 or n == 0:
        assert mod.create_context(0) == (-7, -7)   # no GPU here: loud, never a CPU path
