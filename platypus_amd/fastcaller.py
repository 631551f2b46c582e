"""ctypes binding of libplat_caller.so (include/platypus_caller.h): the native region loop.

    reads of a region in host memory (structure-of-arrays)  ->  VCF record lines

`ReadTable` / `RegionReads` are the array form of ReadArray / bamReadBuffer (cwindow.pyx:92-236,485-513);
`NativeCaller.call_regions` is callVariantsInRegion (variantcaller.pyx:535-615) for a list of regions and writes the
text platypus_amd.caller.callVariantsInRegions writes (tests/test_native_caller_*.py compare the two).  The library
is host code on top of libplat_mi355x.so; like the rest of the package it has no CPU fallback."""
import ctypes as C
import os
import subprocess

import numpy as np

from . import _lib

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libplat_caller.so")
HOST_SRC = os.path.join(HERE, "csrc", "host")


class _ReadTable(C.Structure):
    _fields_ = [("n_reads", C.c_int32), ("_pad", C.c_int32)] + [(k, C.c_void_p) for k in (
        "seq", "qual", "off", "pos", "end", "mapq", "flags", "mate_pos", "cigar", "cig_off")]


class _SampleReads(C.Structure):
    _fields_ = [("reads", _ReadTable), ("bad_reads", _ReadTable), ("broken_mates", _ReadTable)]


class _Region(C.Structure):
    _fields_ = [("chrom", C.c_char_p), ("start", C.c_int32), ("end", C.c_int32), ("contig_seq", C.c_void_p),
                ("contig_len", C.c_int64), ("samples", C.POINTER(_SampleReads))]


_OPT_FIELDS = [("rlen", C.c_int32), ("minReads", C.c_int32), ("maxReads", C.c_double), ("maxSize", C.c_int32), ("largeWindows", C.c_int32),
               ("maxVariants", C.c_int32), ("coverageSamplingLevel", C.c_int32), ("maxHaplotypes", C.c_int32),
               ("originalMaxHaplotypes", C.c_int32), ("skipDifficultWindows", C.c_int32), ("getVariantsFromBAMs", C.c_int32),
               ("genSNPs", C.c_int32), ("genIndels", C.c_int32), ("mergeClusteredVariants", C.c_int32), ("minFlank", C.c_int32),
               ("filterVarsByCoverage", C.c_int32), ("filteredReadsFrac", C.c_double), ("maxVarDist", C.c_int32), ("minVarDist", C.c_int32),
               ("useEMLikelihoods", C.c_int32), ("countOnlyExactIndelMatches", C.c_int32), ("calculateFlankScore", C.c_int32),
               ("assemble", C.c_int32), ("outputRefCalls", C.c_int32), ("minMapQual", C.c_int32), ("minBaseQual", C.c_int32),
               ("minPosterior", C.c_int32), ("sbThreshold", C.c_double), ("scThreshold", C.c_double), ("abThreshold", C.c_double),
               ("minVarFreq", C.c_double), ("badReadsWindow", C.c_int32), ("badReadsThreshold", C.c_int32), ("rmsmqThreshold", C.c_int32),
               ("qdThreshold", C.c_int32), ("hapScoreThreshold", C.c_int32), ("_pad", C.c_int32)]


class CallerOptions(C.Structure):
    _fields_ = _OPT_FIELDS

    @classmethod
    def from_options(cls, options):
        """From the reference's options object (platypus_amd.options.default_options())."""
        o = cls()
        for name, typ in _OPT_FIELDS:
            if name != "_pad":
                setattr(o, name, (float if typ is C.c_double else int)(getattr(options, name)))
        return o


class CallerStats(C.Structure):
    _fields_ = [(k, C.c_int64) for k in ("n_regions", "n_reads", "n_candidate_records", "n_variants", "n_windows", "n_windows_called",
                                         "n_records", "n_windows_greedy", "n_windows_failed")] + \
               [(k, C.c_double) for k in ("seconds_total", "seconds_host", "seconds_device_wait")] + [("seconds_stage", C.c_double * 8)]

    STAGES = ("upload", "candidate_scan", "variants_windows_haplotypes", "greedy_rounds", "window_batch", "posteriors", "read_stats_calls", "text")

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_ if k != "seconds_stage"}
        d["seconds_stage"] = dict(zip(self.STAGES, list(self.seconds_stage)))
        return d


class ReadTable:
    """One ReadArray as arrays (cAlignedRead fields, htslibWrapper.pxd:187-201).  `reads`: the order the ReadArray holds them in
    (sorted by pos; brokenMates by mate position)."""
    __slots__ = ("n", "seq", "qual", "off", "pos", "end", "mapq", "flags", "mate_pos", "cigar", "cig_off", "_pinned", "_struct")

    def __init__(self, seq, qual, off, pos, end, mapq, flags, mate_pos, cigar, cig_off, pin=False):
        """pin=True keeps the two byte blobs in page-locked memory (what a loader that decodes into pinned buffers hands over):
        their upload is then asynchronous and runs at link speed instead of going through the driver's staging copies."""
        self.n = len(pos)
        pad = np.zeros(_lib.PLAT_BLOB_PAD, dtype=np.uint8)
        c = lambda a, dt: np.ascontiguousarray(a, dtype=dt)
        self.seq, self.qual = np.concatenate([c(seq, np.uint8), pad]), np.concatenate([c(qual, np.uint8), pad])
        self._pinned = None
        if pin and self.n:
            import torch
            self._pinned = [torch.from_numpy(self.seq).pin_memory(), torch.from_numpy(self.qual).pin_memory()]
            self.seq, self.qual = self._pinned[0].numpy(), self._pinned[1].numpy()
        self.off, self.pos, self.end = c(off, np.int64), c(pos, np.int32), c(end, np.int32)
        self.mapq, self.flags, self.mate_pos = c(mapq, np.uint8), c(flags, np.int32), c(mate_pos, np.int32)
        self.cigar, self.cig_off = np.concatenate([c(cigar, np.int16).reshape(-1), np.zeros(2, dtype=np.int16)]), c(cig_off, np.int32)
        assert len(self.off) == self.n + 1 and len(self.cig_off) == self.n + 1 and (self.n == 0 or self.off[0] == 0)

    @classmethod
    def from_reads(cls, reads):
        """From hostapi.AlignedRead objects, in the given order."""
        lens = [r.rlen for r in reads]
        cig = [x for r in reads for c in r.cigarOps for x in c]
        return cls(np.frombuffer(b"".join(r.seq for r in reads), dtype=np.uint8), np.frombuffer(b"".join(r.qual for r in reads), dtype=np.uint8),
                   np.concatenate([[0], np.cumsum(lens)]), [r.pos for r in reads], [r.end for r in reads], [r.mapq for r in reads],
                   [r.bitFlag for r in reads], [r.matePos for r in reads], np.array(cig, dtype=np.int16),
                   np.concatenate([[0], np.cumsum([len(r.cigarOps) for r in reads])]))

    def struct(self):
        """The plat_read_table of these arrays (built once: the arrays are kept alive by, and never replaced on, this object)."""
        t = getattr(self, "_struct", None)
        if t is None:
            t = _ReadTable()
            t.n_reads = self.n
            for k in ("seq", "qual", "off", "pos", "end", "mapq", "flags", "mate_pos", "cigar", "cig_off"):
                setattr(t, k, getattr(self, k).ctypes.data)
            self._struct = t
        return t


class RegionReads:
    """One region: chrom:start-end, the contig's sequence and per sample (reads, badReads, brokenMates) as ReadTables."""

    def __init__(self, chrom, start, end, contig_seq, samples):
        self.chrom, self.start, self.end = chrom, int(start), int(end)
        self.contig = np.ascontiguousarray(np.frombuffer(contig_seq, dtype=np.uint8) if isinstance(contig_seq, (bytes, bytearray)) else contig_seq,
                                           dtype=np.uint8)
        self.samples = samples                                  # [(reads, bad, broken)]
        self._c = None

    def c_region(self):
        """(plat_region fields, the plat_sample_reads array they point to), built once per region."""
        if self._c is None:
            ss = (_SampleReads * len(self.samples))()
            for i, (a, b, c) in enumerate(self.samples):
                ss[i].reads, ss[i].bad_reads, ss[i].broken_mates = a.struct(), b.struct(), c.struct()
            self._c = (self.chrom.encode(), self.contig.ctypes.data, len(self.contig), ss)
        return self._c

    @classmethod
    def from_buffers(cls, chrom, start, end, fasta, buffers):
        """From hostapi.bamReadBuffer objects and a hostapi.FastaFile (the inputs of caller.callVariantsInRegions)."""
        return cls(chrom, start, end, fasta._seq[chrom],
                   [(ReadTable.from_reads(b.reads.array), ReadTable.from_reads(b.badReads.array), ReadTable.from_reads(b.brokenMates.array))
                    for b in buffers])


def region_from_arrays(reg, pin=False):
    """RegionReads of a synth.config4_region_arrays() region (every read in `reads`; no badReads / brokenMates)."""
    empty = ReadTable([], [], [0], [], [], [], [], [], [], [0])
    return RegionReads(reg["chrom"], reg["start"], reg["end"], reg["ref"],
                       [(ReadTable(s["seq"], s["qual"], s["off"], s["pos"], s["end"], s["mapq"], s["flags"], s["mate_pos"], s["cigar"], s["cig_off"],
                                   pin=pin), empty, empty) for s in reg["samples"]])


def aligned_reads_from_arrays(s):
    """hostapi.AlignedRead objects of one sample of a synth.config4_region_arrays() region (for the Python region loop)."""
    from .hostapi import AlignedRead
    seq, qual, cig = s["seq"].tobytes(), s["qual"].tobytes(), s["cigar"].reshape(-1, 2)
    off, co = s["off"], s["cig_off"]
    return [AlignedRead(seq[off[i]:off[i + 1]], qual[off[i]:off[i + 1]], int(s["pos"][i]), int(s["mapq"][i]), int(s["flags"][i]), end=int(s["end"][i]),
                        cigarOps=[tuple(x) for x in cig[co[i]:co[i + 1]].tolist()], matePos=int(s["mate_pos"][i])) for i in range(len(s["pos"]))]


def build(verbose=False):
    """g++ the host library and link it to libplat_mi355x.so (built first if needed)."""
    _lib.build()
    srcs = [os.path.join(HOST_SRC, f) for f in ("region_caller.cpp", "records.hpp", "variants.hpp")]
    if os.path.exists(LIB_PATH) and os.path.getmtime(LIB_PATH) >= max([os.path.getmtime(f) for f in srcs] + [os.path.getmtime(_lib.LIB_PATH)]):
        return LIB_PATH
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-ffp-contract=off", "-fvisibility=hidden", srcs[0], "-o", LIB_PATH,
           "-L" + HERE, "-lplat_mi355x", "-Wl,-rpath,$ORIGIN"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if verbose:
        print(" ".join(cmd), r.stdout, r.stderr)
    if r.returncode != 0:
        raise RuntimeError("building libplat_caller.so failed:\n" + r.stderr[-4000:])
    return LIB_PATH


def _bind(lib):
    lib.plat_caller_default_options.argtypes = [C.POINTER(CallerOptions)]
    lib.plat_caller_default_options.restype = None
    lib.plat_caller_create.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    lib.plat_caller_destroy.argtypes = [C.c_void_p]
    lib.plat_call_regions.argtypes = [C.c_void_p, C.POINTER(_Region), C.c_int, C.c_int, C.POINTER(C.c_char_p), C.POINTER(CallerOptions),
                                      C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(CallerStats)]
    lib.plat_caller_free.argtypes = [C.c_void_p]
    lib.plat_caller_free.restype = None
    lib.plat_caller_last_error.argtypes = [C.c_void_p]
    lib.plat_caller_last_error.restype = C.c_char_p
    return lib


_caller_lib = None


def load():
    global _caller_lib
    if _caller_lib is None:
        _lib.load()                                             # (torch's HIP runtime first, see _lib.load)
        if not os.path.exists(LIB_PATH):
            build()
        _caller_lib = _bind(C.CDLL(LIB_PATH))
    return _caller_lib


class NativeCaller:
    """plat_caller: `workers` threads, each with its own plat_ctx and stream on `device`; `regions_per_chunk` regions go through
    the device stages together."""

    def __init__(self, device=0, workers=4, regions_per_chunk=4, lib=None):
        self.lib = lib if lib is not None else load()
        h = C.c_void_p()
        rc = self.lib.plat_caller_create(device, workers, regions_per_chunk, C.byref(h))
        if rc != 0:
            raise _lib.PlatypusDeviceError(rc, "plat_caller_create failed (no GPU? the native region loop has no CPU fallback)", "plat_caller_create")
        self.h = h
        self.stats = None

    def close(self):
        if getattr(self, "h", None):
            self.lib.plat_caller_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def call_regions(self, regions, sample_names, options):
        """regions: list of RegionReads.  Returns the record lines of all regions (str), in region order; options.rlen is updated
        as the reference updates it."""
        n, nS = len(regions), len(sample_names)
        arr = (_Region * max(n, 1))()
        for k, r in enumerate(regions):
            assert len(r.samples) == nS
            a = arr[k]
            a.chrom, a.contig_seq, a.contig_len, ss = r.c_region()
            a.start, a.end = r.start, r.end
            a.samples = ss
        names = (C.c_char_p * nS)(*[s.encode() for s in sample_names])
        o = CallerOptions.from_options(options)
        text, length, st = C.c_void_p(), C.c_size_t(), CallerStats()
        rc = self.lib.plat_call_regions(self.h, arr, n, nS, names, C.byref(o), C.byref(text), C.byref(length), C.byref(st))
        if rc != 0:
            raise _lib.PlatypusDeviceError(rc, (self.lib.plat_caller_last_error(self.h) or b"").decode(), "plat_call_regions")
        try:
            out = C.string_at(text, length.value).decode("ascii")
        finally:
            self.lib.plat_caller_free(text)
        options.rlen = int(o.rlen)
        self.stats = st.as_dict()
        return out
