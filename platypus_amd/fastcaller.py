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


READS_ASCII, READS_PACKED = 0, 1


class _ReadTable(C.Structure):
    _fields_ = [("n_reads", C.c_int32), ("encoding", C.c_int32)] + [(k, C.c_void_p) for k in (
        "seq", "qual", "off", "pos", "end", "mapq", "flags", "mate_pos", "cigar", "cig_off")] + \
               [("n_exceptions", C.c_int64), ("exc_index", C.c_void_p), ("exc_base", C.c_void_p), ("exc_qual", C.c_void_p),
                ("dev_seq", C.c_void_p), ("dev_qual", C.c_void_p)] + [(k, C.c_void_p) for k in (
                    "dev_off", "dev_pos", "dev_end", "dev_mapq", "dev_flags", "dev_cigar", "dev_cig_off")] + \
               [("longest_read", C.c_int32), ("most_bases", C.c_int32)]       # optional: the loader's own figures (0 = not known)


class _SampleReads(C.Structure):
    _fields_ = [("reads", _ReadTable), ("bad_reads", _ReadTable), ("broken_mates", _ReadTable)]


class _Region(C.Structure):
    _fields_ = [("chrom", C.c_char_p), ("start", C.c_int32), ("end", C.c_int32), ("contig_seq", C.c_void_p),
                ("contig_len", C.c_int64), ("samples", C.POINTER(_SampleReads)), ("dev_contig_seq", C.c_void_p)]


_OPT_FIELDS = [("rlen", C.c_int32), ("minReads", C.c_int32), ("maxReads", C.c_double), ("maxSize", C.c_int32), ("largeWindows", C.c_int32),
               ("maxVariants", C.c_int32), ("coverageSamplingLevel", C.c_int32), ("maxHaplotypes", C.c_int32),
               ("originalMaxHaplotypes", C.c_int32), ("skipDifficultWindows", C.c_int32), ("getVariantsFromBAMs", C.c_int32),
               ("genSNPs", C.c_int32), ("genIndels", C.c_int32), ("mergeClusteredVariants", C.c_int32), ("minFlank", C.c_int32),
               ("filterVarsByCoverage", C.c_int32), ("filteredReadsFrac", C.c_double), ("maxVarDist", C.c_int32), ("minVarDist", C.c_int32),
               ("useEMLikelihoods", C.c_int32), ("countOnlyExactIndelMatches", C.c_int32), ("calculateFlankScore", C.c_int32),
               ("assemble", C.c_int32), ("outputRefCalls", C.c_int32), ("minMapQual", C.c_int32), ("minBaseQual", C.c_int32),
               ("minPosterior", C.c_int32), ("sbThreshold", C.c_double), ("scThreshold", C.c_double), ("abThreshold", C.c_double),
               ("minVarFreq", C.c_double), ("badReadsWindow", C.c_int32), ("badReadsThreshold", C.c_int32), ("rmsmqThreshold", C.c_int32),
               ("qdThreshold", C.c_int32), ("hapScoreThreshold", C.c_int32), ("refCallBlockSize", C.c_int32),
               ("assemblyRegionSize", C.c_int32), ("assembleAll", C.c_int32), ("assembleBadReads", C.c_int32), ("assembleBrokenPairs", C.c_int32),
               ("assemblerKmerSize", C.c_int32), ("noCycles", C.c_int32)]


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
               [(k, C.c_double) for k in ("seconds_total", "seconds_host", "seconds_device_wait")] + [("seconds_stage", C.c_double * 8)] + \
               [("seconds_load", C.c_double), ("seconds_source_wait", C.c_double), ("input_bytes", C.c_int64), ("n_assembly_tiles", C.c_int64),
                ("n_assembler_variants", C.c_int64), ("n_refcall_records", C.c_int64), ("seconds_assemble", C.c_double), ("n_pairs", C.c_int64),
                ("n_dp_reference", C.c_int64), ("cells_reference", C.c_int64), ("n_dp_launched", C.c_int64), ("cells_launched", C.c_int64),
                ("n_align_batches", C.c_int64), ("align_hap_bytes", C.c_int64), ("align_read_bytes", C.c_int64), ("align_reads", C.c_int64),
                ("align_dp_bytes", C.c_int64), ("seconds_kernel_seed", C.c_double), ("seconds_kernel_dp", C.c_double),
                ("seconds_kernel_sweep", C.c_double), ("seconds_kernel_pairs", C.c_double), ("n_regions_stage_b_device", C.c_int64),
                ("n_regions_stage_b_host", C.c_int64), ("n_windows_stage_b_host", C.c_int64), ("n_regions_dict_replay_device", C.c_int64),
                ("seconds_worker_cpu", C.c_double), ("seconds_kernel_unpack", C.c_double), ("seconds_kernel_candidates", C.c_double),
                ("unpack_bytes", C.c_int64), ("candidates_bytes", C.c_int64), ("n_unpack_launches", C.c_int64), ("n_candidates_launches", C.c_int64),
                ("kernel_ms", C.c_double * 32), ("kernel_launches", C.c_int64 * 32)]

    STAGES = ("upload", "candidate_scan", "variants_windows_haplotypes", "greedy_rounds", "window_batch", "posteriors", "read_stats_calls", "text")

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_ if k not in ("seconds_stage", "kernel_ms", "kernel_launches")}
        d["seconds_stage"] = dict(zip(self.STAGES, list(self.seconds_stage)))
        d["kernel_ms"], d["kernel_launches"] = list(self.kernel_ms), list(self.kernel_launches)
        return d


class ReadTable:
    """One ReadArray as arrays (cAlignedRead fields, htslibWrapper.pxd:187-201).  `reads`: the order the ReadArray holds them in
    (sorted by pos; brokenMates by mate position)."""
    __slots__ = ("n", "seq", "qual", "off", "pos", "end", "mapq", "flags", "mate_pos", "cigar", "cig_off", "_pinned", "_struct", "encoding", "exc")

    def __init__(self, seq, qual, off, pos, end, mapq, flags, mate_pos, cigar, cig_off, pin=False, packed=False):
        """pin=True keeps the two byte blobs in page-locked memory (what a loader that decodes into pinned buffers hands over):
        their upload is then asynchronous and runs at link speed instead of going through the driver's staging copies.
        packed=True hands the reads over as PLAT_READS_PACKED (one byte per base + exceptions) instead of ASCII bases + qualities."""
        self.n = len(pos)
        pad = np.zeros(_lib.PLAT_BLOB_PAD, dtype=np.uint8)
        c = lambda a, dt: np.ascontiguousarray(a, dtype=dt)
        self.encoding, self.exc = READS_ASCII, None
        if packed:
            seq, qual = c(seq, np.uint8), c(qual, np.uint8)
            plain = (seq == 65) | (seq == 67) | (seq == 71) | (seq == 84)
            ex = np.nonzero(~plain | (qual > 63))[0].astype(np.int64)
            self.exc = (ex, seq[ex].copy(), qual[ex].copy())
            seq = (((seq >> 1) & 3) | (np.minimum(qual, 63) << 2)).astype(np.uint8)
            qual = np.zeros(0, dtype=np.uint8)
            self.encoding = READS_PACKED
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
    def from_reads(cls, reads, packed=False):
        """From hostapi.AlignedRead objects, in the given order."""
        lens = [r.rlen for r in reads]
        cig = [x for r in reads for c in r.cigarOps for x in c]
        return cls(np.frombuffer(b"".join(r.seq for r in reads), dtype=np.uint8), np.frombuffer(b"".join(r.qual for r in reads), dtype=np.uint8),
                   np.concatenate([[0], np.cumsum(lens)]), [r.pos for r in reads], [r.end for r in reads], [r.mapq for r in reads],
                   [r.bitFlag for r in reads], [r.matePos for r in reads], np.array(cig, dtype=np.int16),
                   np.concatenate([[0], np.cumsum([len(r.cigarOps) for r in reads])]), packed=packed)

    def struct(self):
        """The plat_read_table of these arrays (built once: the arrays are kept alive by, and never replaced on, this object)."""
        t = getattr(self, "_struct", None)
        if t is None:
            t = _ReadTable()
            t.n_reads, t.encoding = self.n, self.encoding
            for k in ("seq", "qual", "off", "pos", "end", "mapq", "flags", "mate_pos", "cigar", "cig_off"):
                setattr(t, k, getattr(self, k).ctypes.data)
            if self.exc is not None:
                t.n_exceptions = len(self.exc[0])
                t.exc_index, t.exc_base, t.exc_qual = (a.ctypes.data for a in self.exc)
            if self.n:                                                       # what ReadArray knows of itself (cwindow.pyx:173-174): one numpy pass here, none in the library
                t.longest_read = max(0, int((self.end[:self.n].astype(np.int64) - self.pos[:self.n]).max()))
                t.most_bases = max(0, int(np.diff(self.off[:self.n + 1]).max()))
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
    def from_buffers(cls, chrom, start, end, fasta, buffers, packed=False):
        """From hostapi.bamReadBuffer objects and a hostapi.FastaFile (the inputs of caller.callVariantsInRegions)."""
        return cls(chrom, start, end, fasta._seq[chrom],
                   [(ReadTable.from_reads(b.reads.array, packed), ReadTable.from_reads(b.badReads.array, packed),
                     ReadTable.from_reads(b.brokenMates.array, packed)) for b in buffers])

    def fill(self, a, n_samples):
        """Write this region into the plat_region `a` (the arrays stay owned by, and alive with, this object)."""
        assert len(self.samples) == n_samples
        a.chrom, a.contig_seq, a.contig_len, ss = self.c_region()
        a.start, a.end = self.start, self.end
        a.samples = ss


def region_from_arrays(reg, pin=False, packed=False):
    """RegionReads of a synth.config4_region_arrays() region (every read in `reads`; no badReads / brokenMates)."""
    empty = ReadTable([], [], [0], [], [], [], [], [], [], [0])
    return RegionReads(reg["chrom"], reg["start"], reg["end"], reg["ref"],
                       [(ReadTable(s["seq"], s["qual"], s["off"], s["pos"], s["end"], s["mapq"], s["flags"], s["mate_pos"], s["cigar"], s["cig_off"],
                                   pin=pin, packed=packed), empty, empty) for s in reg["samples"]])


def arrays_from_region_struct(reg):
    """The arrays of a filled plat_region (copies), in the form of synth.config4_region_arrays(): what a region SOURCE handed over
    (tools/synth), for the parity tests that run the same reads through the Python region loop."""
    def arr(ptr, n, dt):
        if not ptr or n == 0:
            return np.zeros(0, dtype=dt)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(np.ctypeslib.as_ctypes_type(dt))), shape=(n,)).copy()

    def table(t):
        n = t.n_reads
        off = arr(t.off, n + 1, np.int64) if n else np.zeros(1, dtype=np.int64)
        nb = int(off[-1])
        seq = arr(t.seq, nb, np.uint8)
        if t.encoding == READS_PACKED:
            qual = (seq >> 2).astype(np.uint8)
            seq = np.frombuffer(b"ACTG", dtype=np.uint8)[seq & 3].copy()
            ne = int(t.n_exceptions)
            if ne:
                ix = arr(t.exc_index, ne, np.int64)
                seq[ix], qual[ix] = arr(t.exc_base, ne, np.uint8), arr(t.exc_qual, ne, np.uint8)
        else:
            qual = arr(t.qual, nb, np.uint8)
        co = arr(t.cig_off, n + 1, np.int32) if n else np.zeros(1, dtype=np.int32)
        return dict(seq=seq, qual=qual, off=off, pos=arr(t.pos, n, np.int32), end=arr(t.end, n, np.int32), mapq=arr(t.mapq, n, np.uint8),
                    flags=arr(t.flags, n, np.int32), mate_pos=arr(t.mate_pos, n, np.int32), cigar=arr(t.cigar, 2 * int(co[-1]), np.int16), cig_off=co)
    return dict(chrom=reg.chrom.decode(), start=int(reg.start), end=int(reg.end), ref=arr(reg.contig_seq, int(reg.contig_len), np.uint8),
                samples=lambda n: [dict(reads=table(reg.samples[i].reads), bad=table(reg.samples[i].bad_reads),
                                        broken=table(reg.samples[i].broken_mates)) for i in range(n)])


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
    srcs = [os.path.join(HOST_SRC, f) for f in sorted(os.listdir(HOST_SRC)) if f.endswith((".cpp", ".hpp"))]
    if os.path.exists(LIB_PATH) and os.path.getmtime(LIB_PATH) >= max([os.path.getmtime(f) for f in srcs] + [os.path.getmtime(_lib.LIB_PATH)]):
        return LIB_PATH
    # (-O3: the region loop is many small functions over small containers; +10 % windows/s over -O2 on the same box.  No -march: the library
    #  travels between machines; no fast-math: the records' numbers are the reference's)
    cmd = ["g++", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread", "-ffp-contract=off", "-fvisibility=hidden", os.path.join(HOST_SRC, "region_caller.cpp"), "-o", LIB_PATH,
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
    lib.plat_caller_count_cells.argtypes = [C.c_void_p, C.c_int]
    lib.plat_caller_time_kernel.argtypes = [C.c_void_p, C.c_int]
    lib.plat_call_regions.argtypes = [C.c_void_p, C.POINTER(_Region), C.c_int, C.c_int, C.POINTER(C.c_char_p), C.POINTER(CallerOptions),
                                      C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(CallerStats)]
    lib.plat_call_regions_stream.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_char_p), C.POINTER(CallerOptions), C.c_void_p, C.c_void_p,
                                             C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(CallerStats)]
    lib.plat_merge_record_texts.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    lib.plat_caller_region_text_lengths.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    lib.plat_merge_region_blocks.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
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


def _native_text(lib, ptr, length, raw):
    """What a call hands back for a malloc'ed text: str; bytes (raw=True); or, raw="view", the block itself as a ctypes char array (no
    copy; freed with the array) -- a whole-genome share is ~100 MB of record text and every copy of it is a pass through memory."""
    if raw == "view":
        import weakref
        arr = (C.c_char * length).from_address(ptr.value) if length else (C.c_char * 0)()
        if length:
            weakref.finalize(arr, lib.plat_caller_free, C.c_void_p(ptr.value))
        else:
            lib.plat_caller_free(ptr)
        return arr
    try:
        out = C.string_at(ptr, length)
        return out if raw else out.decode("ascii")
    finally:
        lib.plat_caller_free(ptr)


def text_bytes(x):
    """bytes of a text returned with raw=True / raw="view"."""
    return x if isinstance(x, bytes) else bytes(x)


def merge_record_texts(texts, lib=None, raw=False):
    """runner.py:301-352 natively: k-way merge of record texts (bytes, each sorted by (chromosome key, position)) -> one str (bytes with raw=True)."""
    lib = lib if lib is not None else load()
    n = len(texts)
    keep = [t if isinstance(t, (bytes, C.Array)) else bytes(t) for t in texts]
    arr = (C.c_void_p * max(n, 1))(*[C.cast(C.c_char_p(t), C.c_void_p).value if isinstance(t, bytes) else C.addressof(t) for t in keep])
    lens = (C.c_size_t * max(n, 1))(*[len(t) for t in keep])
    out, length = C.c_void_p(), C.c_size_t()
    rc = lib.plat_merge_record_texts(arr, lens, n, C.byref(out), C.byref(length))
    del keep
    if rc != 0:
        raise _lib.PlatypusDeviceError(rc, "merge failed", "plat_merge_record_texts")
    return _native_text(lib, out, length.value, raw)


def text_address(t):
    """(address, length) of a text held as bytes, a ctypes array (raw="view"), a numpy uint8 array or a CPU torch uint8 tensor."""
    if isinstance(t, bytes):
        return C.cast(C.c_char_p(t), C.c_void_p).value or 0, len(t)
    if isinstance(t, C.Array):
        return C.addressof(t), len(t)
    if hasattr(t, "data_ptr"):
        return int(t.data_ptr()), int(t.numel())
    return int(t.ctypes.data), int(t.size)


def merge_region_blocks(texts, lengths, keys, lib=None):
    """The job's merge when every text is made of whole regions that do not interleave (the native region loop's output): texts[r] = rank
    r's text, lengths[r] = bytes of each of its regions' records (plat_caller_region_text_lengths), keys[r] = (chromosome key, start, end) of
    each of its regions.  The regions' blocks are put in (chromosome key, start) order -- runner.py:301-352's order -- by block copies on
    up to 16 threads, no line is looked at.  Returns the merged text (raw view), or None when two regions overlap (the caller then merges
    line by line: merge_record_texts)."""
    import numpy as np
    lib = lib if lib is not None else load()
    blocks = []
    for r, (lens, ks) in enumerate(zip(lengths, keys)):
        base, _ = text_address(texts[r])
        off = 0
        for ln, k in zip(lens, ks):
            blocks.append((k[0], k[1], k[2], r, base + off, int(ln)))
            off += int(ln)
    blocks.sort(key=lambda b: (b[0], b[1], b[3]))
    for a, b in zip(blocks, blocks[1:]):
        if a[0] == b[0] and a[2] > b[1] and a[5] and b[5]:
            return None                                              # overlapping regions: their records may interleave
    n = len(blocks)
    src = np.array([b[4] for b in blocks], dtype=np.uint64)
    ln = np.array([b[5] for b in blocks], dtype=np.uint64)
    at = np.zeros(n, dtype=np.uint64)
    if n:
        at[1:] = np.cumsum(ln)[:-1]
    total = int(ln.sum())
    out = C.c_void_p()
    rc = lib.plat_merge_region_blocks(n, src.ctypes.data, ln.ctypes.data, at.ctypes.data, total, C.byref(out))
    if rc != 0:
        raise _lib.PlatypusDeviceError(rc, "merge failed", "plat_merge_region_blocks")
    return _native_text(lib, out, total, "view")


class BlockOrder:
    """merge_region_blocks with everything that depends only on the job's region list worked out ONCE (before a timed region): the order
    of all blocks.  plan = BlockOrder(keys) with keys[r] = [(chromosome key, start, end), ...] of rank r's regions; plan.merge(texts,
    lengths) then costs two small numpy passes and the block copies.  plan.ok is False when regions overlap."""

    def __init__(self, keys):
        import numpy as np
        flat = [(k[0], k[1], k[2], r, j) for r, ks in enumerate(keys) for j, k in enumerate(ks)]
        order = sorted(range(len(flat)), key=lambda i: (flat[i][0], flat[i][1], flat[i][3]))
        self.ok = all(not (flat[a][0] == flat[b][0] and flat[a][2] > flat[b][1]) for a, b in zip(order, order[1:]))
        self.rank = np.array([flat[i][3] for i in order], dtype=np.int64)
        self.index = np.array([flat[i][4] for i in order], dtype=np.int64)
        self.counts = [len(ks) for ks in keys]
        # one rank whose list is already in (chromosome key, start) order: its text IS the merged text (runner.py:301-352 merges per-process
        # files; a job of one process has one), nothing is copied
        self.identity = len(keys) == 1 and bool(np.array_equal(self.index, np.arange(len(self.index))))

    def merge(self, texts, lengths, lib=None):
        import numpy as np
        if self.identity and self.ok and len(texts) == 1:
            assert int(np.asarray(lengths[0], dtype=np.int64).sum()) == text_address(texts[0])[1], "region lengths do not add up to the text"
            return texts[0]
        lib = lib if lib is not None else load()
        starts = []
        for r, lens in enumerate(lengths):
            lens = np.asarray(lens, dtype=np.int64)
            assert len(lens) == self.counts[r]
            base, size = text_address(texts[r])
            st = np.zeros(len(lens), dtype=np.int64)
            if len(lens):
                st[1:] = np.cumsum(lens)[:-1]
            assert int(lens.sum()) == size, "region lengths do not add up to the text"
            starts.append((base + st, lens))
        n = len(self.rank)
        src = np.zeros(n, dtype=np.uint64)
        ln = np.zeros(n, dtype=np.uint64)
        for r, (st, lens) in enumerate(starts):
            m = self.rank == r
            src[m] = st[self.index[m]].astype(np.uint64)
            ln[m] = lens[self.index[m]].astype(np.uint64)
        at = np.zeros(n, dtype=np.uint64)
        if n:
            at[1:] = np.cumsum(ln)[:-1]
        total = int(ln.sum())
        out = C.c_void_p()
        rc = lib.plat_merge_region_blocks(n, src.ctypes.data, ln.ctypes.data, at.ctypes.data, total, C.byref(out))
        if rc != 0:
            raise _lib.PlatypusDeviceError(rc, "merge failed", "plat_merge_region_blocks")
        return _native_text(lib, out, total, "view")


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

    def time_kernel(self, kernel_id):
        """Measurement switch (plat_caller_time_kernel): the calls that follow time this one kernel (id of plat_kernel_timer_name; -1: none) inside the
        ordinary asynchronous runs; stats["kernel_ms"][id] / ["kernel_launches"][id] hold its summed duration and launches."""
        self.lib.plat_caller_time_kernel(self.h, int(kernel_id))

    def count_cells(self, on=True):
        """Measurement switch (plat_caller_count_cells): the calls that follow count the reference's DPs and band cells into stats
        (synchronous likelihood batches: not for timed runs)."""
        rc = self.lib.plat_caller_count_cells(self.h, 1 if on else 0)
        if rc != 0:
            raise _lib.PlatypusDeviceError(rc, "plat_caller_count_cells", "plat_caller_count_cells")

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
            r.fill(arr[k], nS)
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

    def region_text_lengths(self, n_regions):
        """Bytes of record text of every region of the last call, in list order (their blocks lie back to back in the text it returned)."""
        import numpy as np
        out = np.zeros(max(n_regions, 1), dtype=np.int64)
        rc = self.lib.plat_caller_region_text_lengths(self.h, out.ctypes.data, n_regions)
        if rc != 0:
            raise _lib.PlatypusDeviceError(rc, "no call yet, or another number of regions", "plat_caller_region_text_lengths")
        return out[:n_regions]

    def call_stream(self, n_regions, load, user, sample_names, options, n_slots, n_loaders=2, raw=False):
        """plat_call_regions_stream: regions loaded on demand.  `load`: a plat_region_load_fn -- the address of a native function (int,
        e.g. tools/synth's generator: no Python in the loader threads) or a Python callable (index, slot, region_struct) -> status (wrapped;
        tests).  Returns the record lines of all regions in region order: str, or bytes with raw=True (a whole-genome share is ~100 MB of
        text: every decode / encode of it is a pass through memory the caller may not want)."""
        nS = len(sample_names)
        keep = None
        if callable(load):
            fn = load

            def tramp(_user, index, slot, out):
                try:
                    return int(fn(index, slot, out.contents) or 0)
                except Exception:                                       # an exception cannot cross the C frames
                    import traceback
                    traceback.print_exc()
                    return -9
            keep = LOAD_FN(tramp)
            load = C.cast(keep, C.c_void_p)
        names = (C.c_char_p * nS)(*[s.encode() for s in sample_names])
        o = CallerOptions.from_options(options)
        text, length, st = C.c_void_p(), C.c_size_t(), CallerStats()
        rc = self.lib.plat_call_regions_stream(self.h, n_regions, nS, names, C.byref(o), load, user, n_slots, n_loaders, C.byref(text), C.byref(length),
                                               C.byref(st))
        del keep
        if rc != 0:
            raise _lib.PlatypusDeviceError(rc, (self.lib.plat_caller_last_error(self.h) or b"").decode(), "plat_call_regions_stream")
        out = _native_text(self.lib, text, length.value, raw)
        options.rlen = int(o.rlen)
        self.stats = st.as_dict()
        return out


LOAD_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(_Region))
