"""ctypes front of tools/synth/region_source.cpp (libplat_synth.so): BASELINE config 4 as a region SOURCE for
plat_call_regions_stream -- regions generated from seed (+) region index inside the library's loader threads (no Python there), into a
bounded set of pinned slots.  Benchmark tooling: it stands where the reference's BAM loader stands."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libplat_synth.so")
SRC = os.path.join(HERE, "region_source.cpp")

_lib = None


def build():
    hdr = os.path.join(HERE, "..", "..", "include", "platypus_caller.h")
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(SRC), os.path.getmtime(hdr)):
        r = subprocess.run(["g++", "-O3", "-mavx2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-pthread", SRC, "-o", LIB], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("building libplat_synth.so failed:\n" + r.stderr[-3000:])
    return LIB


def load():
    global _lib
    if _lib is None:
        lib = C.CDLL(build())
        lib.plat_synth_slot_bytes.restype = C.c_size_t
        lib.plat_synth_slot_bytes.argtypes = [C.c_int] * 6
        lib.plat_synth_create.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.c_void_p,
                                          C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]
        lib.plat_synth_set_model.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double]
        lib.plat_synth_destroy.argtypes = [C.c_void_p]
        lib.plat_synth_planted.restype = C.c_longlong
        lib.plat_synth_planted.argtypes = [C.c_void_p]
        lib.plat_synth_phase_seconds.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        lib.plat_synth_load_fn.restype = C.c_void_p
        lib.plat_synth_load.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        lib.plat_synth_pregenerate.argtypes = [C.c_void_p, C.c_int]
        lib.plat_synth_set_device_mirror.argtypes = [C.c_void_p, C.c_void_p]
        lib.plat_synth_load_resident_fn.restype = C.c_void_p
        _lib = lib
    return _lib


class RegionSource:
    """The regions `indices` of the job's region list (region id -> contig r<id>), generated on demand.  n_slots pinned buffers
    (torch, page-locked when `pin`) bound the reads in flight."""

    def __init__(self, indices, n_slots, seed=4004, region_len=100000, flank=1000, n_samples=1, depth=30, read_len=150, snp_rate=1e-3, indel_rate=1e-4,
                 err=1e-3, packed=True, pin=True, model=None):
        """model: None = config 4 (Poisson counts, indels of 1..10 bases) or dict(indel_max_len, indel_p, n_indel=(min, max), n_snp=(min, max),
        lowq_frac) = config 3's assembly regions."""
        lib = load()
        self.lib = lib
        self.indices = np.ascontiguousarray(indices, dtype=np.int32)
        self.n_slots, self.n_samples, self.encoding = int(n_slots), n_samples, 1 if packed else 0
        self.slot_bytes = int(lib.plat_synth_slot_bytes(region_len, flank, n_samples, depth, read_len, self.encoding))
        import torch
        self.buf = torch.empty(self.n_slots * self.slot_bytes, dtype=torch.uint8)
        if pin:
            self.buf = self.buf.pin_memory()
        h = C.c_void_p()
        rc = lib.plat_synth_create(seed, region_len, flank, n_samples, depth, read_len, snp_rate, indel_rate, err, self.encoding, self.indices.ctypes.data,
                                   len(self.indices), self.buf.data_ptr(), self.slot_bytes, self.n_slots, C.byref(h))
        if rc != 0:
            raise ValueError("plat_synth_create refused the parameters (%d)" % rc)
        self.h = h
        if model:
            rc = lib.plat_synth_set_model(h, model.get("indel_max_len", 10), model.get("indel_p", 0.4), *model.get("n_indel", (-1, -1)), *model.get("n_snp", (-1, -1)),
                                          model.get("lowq_frac", 0.0))
            if rc != 0:
                raise ValueError("plat_synth_set_model refused the parameters")
        self.bytes_per_base = 1 if packed else 2

    @property
    def load_fn(self):
        return self.lib.plat_synth_load_resident_fn() if self.resident else self.lib.plat_synth_load_fn()

    resident = False
    mirror = None

    def make_resident(self, device=None, threads=8):
        """Generate every region of the list once (region k -> slot k: the source must have been created with n_slots >= len(indices)) and,
        with `device` (a torch device), upload the whole slot memory there in one copy: load_fn then hands out stored regions whose read
        bytes are already in HBM (plat_read_table.dev_seq) -- inputs resident when the timed region starts, as the bench contract has it."""
        if self.n_slots < len(self.indices):
            raise ValueError("resident mode keeps one slot per region")
        rc = self.lib.plat_synth_pregenerate(self.h, int(threads))
        if rc != 0:
            raise RuntimeError("plat_synth_pregenerate failed (%d)" % rc)
        if device is not None:
            self.mirror = self.buf.to(device)
            self.lib.plat_synth_set_device_mirror(self.h, self.mirror.data_ptr())
        self.resident = True

    @property
    def planted(self):
        return int(self.lib.plat_synth_planted(self.h))

    @property
    def phase_seconds(self):
        out = (C.c_double * 6)()
        self.lib.plat_synth_phase_seconds(self.h, out)
        return dict(zip(("reference", "variants", "haplotypes", "read_starts_and_sort", "reads", "rest"), list(out)))

    def region(self, index, slot=0):
        """Region `index` generated into `slot`, as a filled plat_region struct (tests: the same reads for the other paths)."""
        from platypus_amd import fastcaller as F
        reg = F._Region()
        rc = self.lib.plat_synth_load(self.h, index, slot, C.byref(reg))
        if rc != 0:
            raise RuntimeError("plat_synth_load failed (%d)" % rc)
        return reg

    def close(self):
        if getattr(self, "h", None):
            self.lib.plat_synth_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
