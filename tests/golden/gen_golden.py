#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE's own code.

Runs only in the build container (needs /root/reference, gcc, Cython).  Nothing from the
reference is written into the repository: the reference sources are compiled in a scratch
directory (default /tmp/platgold) and only INPUT/OUTPUT DATA is saved here.

Three reference builds are used (details + honest caveats in tests/golden/README.md):

 1. oracle/_ref/libalign_ref.so  -- src/c/align.c, UNMODIFIED (`make -C oracle ref`).
 2. calign                       -- src/cython/calign.pyx compiled with Cython 3 next to align.c.
                                    Build-level adaptations only: directive cpow=True (Cython-0.x
                                    integer `**`), the module-level `import htslibWrapper` line
                                    (unused at run time) neutralised, and a struct-only
                                    htslibWrapper.pxd = lines 187-201 of the original .pxd (the
                                    cAlignedRead struct), because htslib is not in this image.
 3. assembler core               -- src/cython/assembler.pyx lines 30-1389 textually included in a
                                    driver that feeds reads from Python lists (replaces the
                                    bamReadBuffer loop at :1391-1425) and mirrors the entry point
                                    :1429-1476.

Usage:  python tests/golden/gen_golden.py [--scratch DIR]
"""
import argparse
import gzip
import json
import os
import shutil
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("PLATYPUS_REF", "/root/reference")
sys.path.insert(0, ROOT)

CALIGN_DRV = r'''
cimport calign
from htslibWrapper cimport cAlignedRead
from libc.stdlib cimport malloc, free

def map_and_align(bytes read, bytes qual, int readStart, int hapStart, bytes hap, bytes gapopen, int hapFlank, int doFlank):
    cdef short* table = NULL
    cdef short* nxt = NULL
    cdef cAlignedRead r
    cdef int hapLen = len(hap)
    cdef int readLen = len(read)
    cdef int mcl = 2*(hapLen+readLen)
    cdef int* mc = <int*>malloc(mcl*sizeof(int))
    cdef char* cread = read
    cdef char* cqual = qual
    cdef char* chap = hap
    cdef char* cgo = gapopen
    r.seq = cread
    r.qual = cqual
    r.rlen = readLen
    r.hash = NULL
    calign.hash_sequence_multihit(chap, hapLen, &table, &nxt)
    if readLen >= 7:
        calign.hashReadForMapping(&r)
    cdef int sc = calign.mapAndAlignReadToHaplotype(cread, cqual, readStart, hapStart, readLen, hapLen, table, nxt, r.hash, chap, 3, 2, cgo, mc, mcl, hapFlank, doFlank)
    free(table); free(nxt); free(mc)
    if r.hash != NULL:
        free(r.hash)
    return sc
'''

ASM_DRV = r'''
cimport cython
import logging
logger = logging.getLogger("Log")
StandardError = Exception
cdef int ASSEMBLER_VAR = 4

ctypedef struct cAlignedRead:
    char* seq
    char* qual
    short* cigarOps
    short* hash
    short mateChromID
    short cigarLen
    short chromID
    short rlen
    int pos
    int end
    int insertSize
    int matePos
    int bitFlag
    unsigned char mapq

cdef class Variant:
    cdef public bytes refName
    cdef public int refPos
    cdef public bytes removed
    cdef public bytes added
    cdef public int varSource
    def __init__(self, bytes refName, int refPos, char* removed, char* added, int nSupportingReads, int varSource):
        self.refName = refName
        self.refPos = max(0, refPos)
        self.removed = removed
        self.added = added
        self.varSource = varSource

include "asm_core.pxi"

def assemble(bytes chrom, int assemStart, int assemEnd, int refStart, int refEnd, list seqs, list quals, bytes refSeq,
             int minQual, int minReads, int kmerSize, int noCycles):
    cdef int minWeight = minReads*minQual
    cdef int nBuckets = 5000
    cdef int verbosity = 0
    cdef list theVars = []
    cdef cAlignedRead r
    cdef DeBruijnGraph* theGraph = createDeBruijnGraph(kmerSize, nBuckets)
    cdef char* cref = refSeq
    cdef char* cchrom = chrom
    cdef bytes s
    cdef bytes q
    cdef int found = 1
    loadReferenceIntoGraph(theGraph, cref, refStart, kmerSize)
    for s, q in zip(seqs, quals):
        r.seq = s; r.qual = q; r.rlen = len(s); r.pos = 0
        loadReadIntoGraph(&r, theGraph, minQual, kmerSize)
    if noCycles:
        while detectCyclesInGraph_Recursive(theGraph, minWeight):
            if kmerSize > 50:
                found = 0
                break
            else:
                kmerSize += 5
                destroyDeBruijnGraph(theGraph)
                theGraph = createDeBruijnGraph(kmerSize, nBuckets)
                loadReferenceIntoGraph(theGraph, cref, refStart, kmerSize)
                for s, q in zip(seqs, quals):
                    r.seq = s; r.qual = q; r.rlen = len(s); r.pos = 0
                    loadReadIntoGraph(&r, theGraph, minQual, kmerSize)
    if found:
        theVars = findBubblesInGraph(theGraph, minWeight, cref, cchrom, refStart, refEnd, assemStart, assemEnd, verbosity)
    nNodes = theGraph.allNodes.top + 1
    destroyDeBruijnGraph(theGraph)
    return [(v.refPos, v.removed, v.added) for v in theVars], nNodes
'''

SETUP = r'''
from setuptools import setup, Extension
from Cython.Build import cythonize
exts = [Extension("calign", ["calign.pyx", "align.c"], include_dirs=["."]),
        Extension("calign_drv", ["calign_drv.pyx"], include_dirs=["."]),
        Extension("asm_drv", ["asm_drv.pyx"])]
setup(ext_modules=cythonize(exts, language_level=2,
      compiler_directives=dict(cdivision=True, cpow=True, legacy_implicit_noexcept=True)))
'''


def build_scratch(scratch):
    src = os.path.join(REF, "src")
    if os.path.isdir(scratch):
        shutil.rmtree(scratch)
    os.makedirs(scratch)
    for f in ("cython/calign.pyx", "cython/calign.pxd", "cython/cerrormodel.pxd", "c/align.c", "c/align.h"):
        shutil.copy(os.path.join(src, f), scratch)
    lines = open(os.path.join(src, "cython/htslibWrapper.pxd")).read().split("\n")
    open(os.path.join(scratch, "htslibWrapper.pxd"), "w").write("\n".join(lines[186:201]) + "\n")
    p = os.path.join(scratch, "calign.pyx")
    txt = open(p).read().replace("\nimport htslibWrapper\n", "\npass\n")
    open(p, "w").write(txt)
    asm = open(os.path.join(src, "cython/assembler.pyx")).read().split("\n")
    open(os.path.join(scratch, "asm_core.pxi"), "w").write("\n".join(asm[29:1389]) + "\n")
    open(os.path.join(scratch, "calign_drv.pyx"), "w").write(CALIGN_DRV)
    open(os.path.join(scratch, "asm_drv.pyx"), "w").write(ASM_DRV)
    open(os.path.join(scratch, "setup.py"), "w").write(SETUP)
    r = subprocess.run([sys.executable, "setup.py", "build_ext", "--inplace"], cwd=scratch,
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stdout[-3000:] + r.stderr[-3000:])
    sys.path.insert(0, scratch)


B = b"ACGT"


def rnd(rng, n):
    return bytes(rng.choice(list(B), n).tolist())


def tandem(rng, n):
    u = rnd(rng, int(rng.integers(1, 9)))
    return (u * (n // len(u) + 1))[:n]


def vtype(r, a):
    if len(r) == len(a):
        return 0 if len(a) == 1 else 1
    if len(r) == 0:
        return 2
    if len(a) == 0:
        return 3
    return 4


# ------------------------------------------------------------------------------------------------
def gen_dp(out):
    """>=2000 {hap slice, read, qual, gap-open -> score}; 500 of them with traceback strings."""
    from oracle.oracle import RefAlign
    ref = RefAlign()
    rng = np.random.default_rng(20260928)
    LMAX = 250
    n = 2400
    haps = np.full((n, LMAX + 15), ord("A"), dtype=np.uint8)
    reads = np.full((n, LMAX), ord("A"), dtype=np.uint8)
    quals = np.zeros((n, LMAX), dtype=np.uint8)
    gos = np.ones((n, LMAX + 15), dtype=np.uint8)
    lens = np.zeros(n, dtype=np.int32)
    score = np.zeros(n, dtype=np.int32)
    tb_idx, tb_a1, tb_a2, tb_fp = [], [], [], []
    for j in range(n):
        L = int(rng.choice([7, 8, 20, 36, 50, 100, 150, 250]))
        hap = bytearray(rnd(rng, L + 15))
        off = int(rng.integers(0, 16))
        read = bytearray(hap[off:off + L])
        mode = int(rng.integers(0, 7))
        for _ in range(int(rng.integers(0, 1 + (40 if mode == 5 else 4)))):
            read[int(rng.integers(0, L))] = B[int(rng.integers(0, 4))]
        if mode in (1, 2) and L > 20:
            p = int(rng.integers(5, L - 5))
            k = int(rng.integers(1, 7))
            if mode == 1:
                read = (read[:p] + bytearray(rnd(rng, k)) + read[p:])[:L]
            else:
                read = read[:p] + read[p + k:] + bytearray(rnd(rng, k))
        if mode == 3:
            for _ in range(3):
                hap[int(rng.integers(0, L + 15))] = ord("N")
        if mode == 4:
            read[int(rng.integers(0, L))] = ord("N")
        if mode == 6:   # homopolymer / repeat context
            hap = bytearray(tandem(rng, L + 15))
            read = bytearray(hap[off:off + L])
            if L > 20:
                p = int(rng.integers(5, L - 5))
                read = (read[:p] + read[p + 2:] + bytearray(rnd(rng, 2)))[:L]
        q = rng.integers(0, 94, L).astype(np.uint8)
        if rng.random() < 0.3:
            q[:] = int(rng.integers(0, 94))
        if rng.random() < 0.3:
            q[rng.random(L) < 0.2] = 0     # trimmed bases (cwindow.pyx:415-479 set qual 0)
        go = rng.integers(1, 46, L + 15).astype(np.uint8)
        haps[j, :L + 15] = np.frombuffer(bytes(hap), dtype=np.uint8)
        reads[j, :L] = np.frombuffer(bytes(read[:L]), dtype=np.uint8)
        quals[j, :L] = q
        gos[j, :L + 15] = go
        lens[j] = L
        hs, rs, qs, gs = bytes(hap), bytes(read[:L]), bytes(q.tolist()), bytes(go.tolist())
        score[j] = ref.dp_score(hs, rs, qs, gs)
        if j % 5 == 0:
            sc, a1, a2, fp = ref.dp_align(hs, rs, qs, gs)
            assert sc == score[j]
            tb_idx.append(j); tb_a1.append(a1.decode()); tb_a2.append(a2.decode()); tb_fp.append(fp)
    np.savez_compressed(os.path.join(out, "dp_cases.npz"), haps=haps, reads=reads, quals=quals, gos=gos,
                        lens=lens, score=score, tb_idx=np.array(tb_idx, dtype=np.int32),
                        tb_aln1=np.array(tb_a1), tb_aln2=np.array(tb_a2),
                        tb_firstpos=np.array(tb_fp, dtype=np.int32))
    print("dp_cases:", n, "cases;", len(tb_idx), "with traceback")


# ------------------------------------------------------------------------------------------------
def gen_mapalign(out):
    import calign_drv
    from oracle.oracle import Oracle
    o = Oracle()
    rng = np.random.default_rng(20260929)
    cases = []
    for it in range(640):
        hapLen = int(rng.integers(450, 1100))
        kind = int(rng.integers(0, 4))
        hap = bytearray(rnd(rng, hapLen))
        if kind == 1:
            p = int(rng.integers(50, hapLen - 200)); ln = int(rng.integers(20, 150)); hap[p:p + ln] = tandem(rng, ln)
        if kind == 2:
            for _ in range(5):
                hap[int(rng.integers(0, hapLen))] = ord("N")
        if kind == 3:
            hap = bytearray(tandem(rng, hapLen))
        L = int(rng.choice([20, 36, 50, 100, 150, 250]))
        start = int(rng.integers(-L // 2, hapLen - L // 2))
        read = bytearray()
        for p in range(start, start + L):
            read.append(hap[p] if 0 <= p < hapLen else B[int(rng.integers(0, 4))])
        for _ in range(int(rng.integers(0, 9))):
            read[int(rng.integers(0, L))] = B[int(rng.integers(0, 4))]
        m = int(rng.integers(0, 4))
        if m == 1 and L > 30:
            p = int(rng.integers(5, L - 5)); k = int(rng.integers(1, 13)); read = (read[:p] + bytearray(rnd(rng, k)) + read[p:])[:L]
        if m == 2 and L > 30:
            p = int(rng.integers(5, L - 15)); k = int(rng.integers(1, 13)); read = read[:p] + read[p + k:] + bytearray(rnd(rng, k))
        if rng.random() < 0.1:
            read[int(rng.integers(0, L))] = ord("N")
        read = bytes(read[:L])
        qual = bytes(rng.integers(0, 60, L).astype(np.uint8).tolist())
        hap = bytes(hap)
        go = o.gap_open(hap)       # validated separately against the formula; input to the reference here
        hapStart = 1000
        readStart = hapStart + start + (int(rng.integers(-200, 201)) if rng.random() < 0.5 else 0)
        flank = int(rng.choice([0, 200, 300]))
        doFlank = int(rng.random() < 0.3) if flank > 0 else 0   # reference dereferences NULL aln when flank==0
        sc = calign_drv.map_and_align(read, qual, readStart, hapStart, hap, go, flank, doFlank)
        cases.append(dict(read=read.decode(), qual=list(qual), readStart=readStart, hapStart=hapStart,
                          hap=hap.decode(), flank=flank, doFlank=doFlank, score=int(sc)))
    with gzip.open(os.path.join(out, "mapalign_cases.json.gz"), "wt") as f:
        json.dump(cases, f)
    print("mapalign_cases:", len(cases))


# ------------------------------------------------------------------------------------------------
def synth_region(rng, ref_len, nh, L, depth, nvar, in_window=True, low_q=0.05, err=0.002):
    ref = bytearray(rnd(rng, ref_len))
    if rng.random() < 0.3:
        p = int(rng.integers(100, ref_len - 200)); ln = int(rng.integers(20, 120)); ref[p:p + ln] = tandem(rng, ln)
    if rng.random() < 0.15:
        ref[int(rng.integers(0, ref_len))] = ord("N")
    ref = bytes(ref)
    refStart = int(rng.integers(0, 100000))
    a0 = refStart + int(rng.integers(0, ref_len // 3))
    a1 = a0 + int(rng.integers(100, 1500))
    donors = []
    for _ in range(nh):
        d = bytearray(ref)
        for _ in range(int(rng.integers(0, nvar + 1))):
            lo, hi = max(50, a0 - refStart), min(len(d) - 100, a1 - refStart) + 1
            p = int(rng.integers(lo, hi)) if (in_window and hi > lo) else int(rng.integers(50, len(d) - 100))
            t = int(rng.integers(0, 3))
            if t == 0:
                d[p] = B[int(rng.integers(0, 4))]
            elif t == 1:
                d[p:p] = rnd(rng, int(rng.integers(1, 41)))
            else:
                del d[p:p + int(rng.integers(1, 41))]
        donors.append(bytes(d))
    seqs, quals = [], []
    for _ in range(depth * ref_len // L):
        d = donors[int(rng.integers(0, nh))]
        if len(d) <= L:
            continue
        p = int(rng.integers(0, len(d) - L))
        s = bytearray(d[p:p + L])
        q = np.clip(rng.normal(35, 5, L), 2, 41).astype(np.uint8)
        lo = rng.random(L) < low_q
        q[lo] = rng.integers(2, 20, int(lo.sum()))
        for e in np.nonzero(rng.random(L) < err)[0]:
            s[e] = B[int(rng.integers(0, 4))]
        if rng.random() < 0.02:
            s[int(rng.integers(0, L))] = ord("N")
        seqs.append(bytes(s)); quals.append(bytes(q.tolist()))
    return ref, refStart, a0, a1, seqs, quals


def synth_combinatorial(rng):
    ref = rnd(rng, 700)
    refStart = int(rng.integers(0, 100000))
    sites = sorted(set(int(x) for x in rng.integers(300, 350, 6)))
    seqs, quals = [], []
    for _ in range(900):
        d = bytearray(ref)
        for s in sites:
            if rng.random() < 0.5:
                d[s] = B[(B.index(d[s]) + 1) % 4]
        p = int(rng.integers(150, 400))
        seqs.append(bytes(d[p:p + 100])); quals.append(bytes([35] * 100))
    return ref, refStart, refStart + 100, refStart + 600, seqs, quals


def gen_assembler(out):
    import asm_drv
    rng = np.random.default_rng(20260930)
    cases = []
    tot = 0
    for it in range(110):
        ref_len = int(rng.integers(600, 1800))
        nh = int(rng.choice([1, 2, 2, 2, 2, 4, 8]))
        L = int(rng.choice([100, 150, 250]))
        depth = int(rng.choice([15, 30, 30]))
        nvar = int(rng.choice([0, 3, 6]))
        ref, refStart, a0, a1, seqs, quals = synth_region(rng, ref_len, nh, L, depth, nvar)
        if it % 11 == 10:   # stress: 6 SNP sites within 50 bp, all allele combinations -> ">20 paths" aborts
            ref, refStart, a0, a1, seqs, quals = synth_combinatorial(rng)
        k = 15
        nc = int(it % 4 == 3)
        ev, nn = asm_drv.assemble(b"chr1", a0, a1, refStart, refStart + ref_len, seqs, quals, ref, 20, 2, k, nc)
        sv = sorted(ev, key=lambda v: (v[0], vtype(v[1], v[2]), len(v[1])))
        tot += len(sv)
        cases.append(dict(ref=ref.decode(), refStart=refStart, assemStart=a0, assemEnd=a1, k=k, minQual=20,
                          minWeight=40, noCycles=nc, seqs=[s.decode() for s in seqs],
                          quals=[q.decode("latin1") for q in quals], nNodes=int(nn),
                          emitted=[[int(p), r.decode(), a.decode()] for p, r, a in ev],
                          variants=[[int(p), r.decode(), a.decode()] for p, r, a in sv]))
    with gzip.open(os.path.join(out, "assembler_cases.json.gz"), "wt") as f:
        json.dump(cases, f)
    print("assembler_cases:", len(cases), "regions,", tot, "variants")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scratch", default="/tmp/platgold")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    if not os.path.isdir(REF):
        sys.exit("reference tree not found at %s: golden vectors can only be regenerated in the build container" % REF)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    build_scratch(a.scratch)
    todo = a.only.split(",") if a.only else ["dp", "mapalign", "assembler"]
    if "dp" in todo:
        gen_dp(HERE)
    if "mapalign" in todo:
        gen_mapalign(HERE)
    if "assembler" in todo:
        gen_assembler(HERE)


if __name__ == "__main__":
    main()
