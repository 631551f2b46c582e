"""CPU tests of the drop-in boundary: the C-ABI library builds, loads and exports every symbol the
header declares (no compute calls: there is no GPU here), and the ctypes mirrors match the header."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

from platypus_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "platypus_mi355x.h")


def test_library_builds_and_exports_every_declared_symbol():
    lib = _lib.load()
    text = open(HEADER).read()
    declared = set(re.findall(r"\b(plat_[a-z0-9_]+)\s*\(", text))
    declared -= {"plat_ctx"}
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.plat_abi_version() == _lib.PLAT_ABI_VERSION


def test_error_strings_and_no_device_is_loud():
    lib = _lib.load()
    for code in _lib.ERRORS:
        assert lib.plat_strerror(code)
    n = C.c_int(-1)
    rc = lib.plat_device_count(C.byref(n))
    if rc != 0 or n.value == 0:
        ctx = C.c_void_p()
        assert lib.plat_ctx_create(0, C.byref(ctx)) == -7          # PLAT_ERR_NO_DEVICE, never a CPU fallback
        with pytest.raises(_lib.PlatypusDeviceError):
            _lib.check(-7, "plat_ctx_create")


def test_struct_layout_matches_header(tmp_path):
    src = tmp_path / "lay.c"
    src.write_text(r'''
#include <stdio.h>
#include <stddef.h>
#include "platypus_mi355x.h"
int main(void){
  printf("%zu %zu %zu %zu\n", sizeof(plat_window_batch), offsetof(plat_window_batch, win_hap_begin),
         offsetof(plat_window_batch, hap_seq), offsetof(plat_window_batch, read_kind));
  printf("%zu %zu\n", sizeof(plat_align_stats), sizeof(plat_assembly_batch));
  printf("%d %d\n", PLAT_BLOB_PAD, PLAT_ABI_VERSION);
  return 0; }''')
    exe = tmp_path / "lay"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)], text=True).split()
    vals = list(map(int, out))
    WB = _lib.WindowBatch
    assert vals[0] == C.sizeof(WB) and vals[1] == WB.win_hap_begin.offset
    assert vals[2] == WB.hap_seq.offset and vals[3] == WB.read_kind.offset
    assert vals[4] == C.sizeof(_lib.AlignStats) and vals[5] == C.sizeof(_lib.AssemblyBatch)
    assert vals[6] == _lib.PLAT_BLOB_PAD and vals[7] == _lib.PLAT_ABI_VERSION


def test_product_package_never_imports_the_oracle():
    """The product path must not import, link or dlopen anything under oracle/ (no CPU fallback)."""
    pkg = os.path.join(ROOT, "platypus_amd")
    bad = re.compile(r"import\s+oracle|from\s+oracle|liborc|oracle\.oracle|oracle/|libalign_ref")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", "Makefile")):
                txt = open(os.path.join(dp, f)).read()
                assert not bad.search(txt), (dp, f)
