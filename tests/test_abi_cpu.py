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
    # tools/ and the headers neither: what calls the oracle lives under tests/ (tests/soak for the long runs)
    bad2 = re.compile(r"import\s+oracle|from\s+oracle|liborc|libalign_ref|fakedev")
    for dp in (os.path.join(ROOT, "tools"), os.path.join(ROOT, "include"), os.path.join(ROOT, "bindings")):
        for f in os.listdir(dp):
            if f.endswith((".py", ".sh", ".h", ".pxd", ".pyx")):
                assert not bad2.search(open(os.path.join(dp, f)).read()), (dp, f)


def test_caller_library_builds_and_matches_its_header(tmp_path):
    """libplat_caller.so (the native region loop, include/platypus_caller.h): builds, links against the device library only,
    exports every declared symbol, and the ctypes mirrors of its structs have the header's layout."""
    from platypus_amd import fastcaller as F
    F.build()
    lib = C.CDLL(F.LIB_PATH)
    text = open(os.path.join(ROOT, "include", "platypus_caller.h")).read()
    declared = set(re.findall(r"\b(plat_(?:call|merge)[a-z0-9_]*)\s*\(", text))
    assert declared == {"plat_caller_default_options", "plat_caller_create", "plat_caller_destroy", "plat_call_regions", "plat_call_regions_stream",
                        "plat_caller_free", "plat_caller_last_error", "plat_merge_record_texts", "plat_caller_count_cells", "plat_caller_time_kernel",
                            "plat_caller_region_text_lengths", "plat_merge_region_blocks"}
    for name in declared:
        assert hasattr(lib, name), name
    needed = subprocess.check_output(["readelf", "-d", F.LIB_PATH], text=True)
    assert "libplat_mi355x.so" in needed and "orc" not in needed and "fake" not in needed
    src = tmp_path / "lay.c"
    src.write_text(r'''
#include <stdio.h>
#include <stddef.h>
#include "platypus_caller.h"
int main(void){
  printf("%zu %zu %zu %zu %zu\n", sizeof(plat_read_table), sizeof(plat_sample_reads), sizeof(plat_region), sizeof(plat_caller_options), sizeof(plat_caller_stats));
  printf("%zu %zu %zu %zu\n", offsetof(plat_caller_options, maxReads), offsetof(plat_caller_options, filteredReadsFrac),
         offsetof(plat_caller_options, sbThreshold), offsetof(plat_caller_options, hapScoreThreshold));
  plat_caller_options o; plat_caller_default_options(&o);
  printf("%d %d %d %d\n", o.rlen, o.maxVariants, o.minPosterior, o.badReadsWindow);
  return 0; }''')
    exe = tmp_path / "lay"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe), "-L" + os.path.dirname(F.LIB_PATH), "-lplat_caller",
                           "-lplat_mi355x", "-Wl,-rpath," + os.path.dirname(F.LIB_PATH)])
    vals = list(map(int, subprocess.check_output([str(exe)], text=True).split()))
    O = F.CallerOptions
    assert vals[:5] == [C.sizeof(F._ReadTable), C.sizeof(F._SampleReads), C.sizeof(F._Region), C.sizeof(O), C.sizeof(F.CallerStats)]
    assert vals[5:9] == [O.maxReads.offset, O.filteredReadsFrac.offset, O.sbThreshold.offset, O.hapScoreThreshold.offset]
    assert vals[9:] == [150, 8, 5, 11]
    from platypus_amd.options import default_options
    o = O.from_options(default_options())
    d = O()
    lib.plat_caller_default_options.argtypes = [C.POINTER(O)]
    lib.plat_caller_default_options(C.byref(d))
    assert all(getattr(o, k) == getattr(d, k) for k, _ in O._fields_ if k != "_pad")
