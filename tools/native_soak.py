"""Soak of the native region loop (libplat_caller.so) against the Python region loop (platypus_amd.caller): text equality over
many synthetic regions, sample counts, read lengths, variant densities (greedy haplotype filter included), read classes and
option variants.  usage: python tools/native_soak.py [seconds]   (tests/soak/native_soak_fake.py runs the same loop without a GPU)"""
import io
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from platypus_amd import caller, fastcaller as F, hostapi as H, synth      # noqa: E402
from platypus_amd.options import default_options                           # noqa: E402
from platypus_amd.vcfrecords import VCF                                     # noqa: E402


def main(lib=None):
    """lib: a libplat_caller.so handle to use instead of the product one (tests/soak/native_soak_fake.py passes its own)."""
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    t0 = time.time()
    rounds = lines = windows = greedy = tiles = refcalls = 0
    seed = 90000
    nc = {}
    while time.time() - t0 < budget:
        rng = np.random.default_rng(seed)
        ns = 1 + seed % 3
        dense = seed % 5 == 0
        kw = dict(region_len=[1500, 2500, 4000, 8000][seed % 4], n_samples=ns, snp_rate=(4e-2 if dense else 3e-3 + 1e-3 * (seed % 3)),
                  indel_rate=(6e-3 if dense else 1e-3 * (seed % 4)), read_len=[100, 150, 76][seed % 3], depth=[20, 35, 50][seed % 3])
        regs = [synth.config4_region(i, seed=seed, **kw) for i in range(3)]
        if seed % 11 == 0:
            regs[1]["samples"] = [[] for _ in range(ns)]                      # a region without reads
        fasta = H.FastaFile({r["chrom"]: r["ref"] for r in regs})
        names = ["S%d" % (i + 1) for i in range(ns)]
        work = []
        for r in regs:
            bufs = []
            for i, rd in enumerate(r["samples"]):
                good, bad, broken = [], [], []
                for x in rd:
                    a = H.AlignedRead(x["seq"], x["qual"], x["pos"], x["mapq"], x["flag"], end=x["end"], cigarOps=x["cigar"])
                    u = rng.random() if seed % 2 else 1.0
                    if u < 0.05:
                        a.mapq = int(rng.integers(0, 20)); a.bitFlag |= 512; bad.append(a)
                    elif u < 0.08:
                        a.matePos = a.pos + int(rng.integers(-300, 300)); broken.append(a)
                    else:
                        good.append(a)
                bufs.append(H.bamReadBuffer(good, bad, broken, sample=names[i]))
            work.append((r["chrom"], r["start"], r["end"], bufs))
        over = dict(maxVariants=[8, 3, 8, 12][seed % 4], mergeClusteredVariants=int(seed % 5 != 1), minPosterior=[5, 0, 5, 20][seed % 4],
                    countOnlyExactIndelMatches=seed % 2, filterVarsByCoverage=int(seed % 6 != 0), maxHaplotypes=[50, 50, 12][seed % 3])
        if seed % 7 == 0:                                                     # round 3: the assembler, reference calls, packed read tables
            over.update(assemble=1, assemblyRegionSize=[1500, 700][seed % 2], assembleBrokenPairs=seed % 3 == 0, getVariantsFromBAMs=int(seed % 21 != 0))
        if seed % 9 == 0:
            over.update(outputRefCalls=1, refCallBlockSize=[1000, 150, 37][seed % 3])
        py = io.StringIO()
        o1 = default_options(**over)
        windows += caller.callVariantsInRegions(work, fasta, o1, VCF(names), py)
        key = (1 + seed % 4, 1 + seed % 3)
        if key not in nc:
            nc[key] = F.NativeCaller(0, key[0], key[1], lib=lib)
        o2 = default_options(**over)
        txt = nc[key].call_regions([F.RegionReads.from_buffers(c, s, e, fasta, b, packed=bool(seed % 2)) for c, s, e, b in work], names, o2)
        tiles += nc[key].stats["n_assembly_tiles"]; refcalls += nc[key].stats["n_refcall_records"]
        if txt != py.getvalue() or o1.rlen != o2.rlen:
            print("MISMATCH at seed", seed)
            a, b = py.getvalue().split("\n"), txt.split("\n")
            for x, y in zip(a, b):
                if x != y:
                    print("PY :", x); print("C++:", y); break
            sys.exit(1)
        greedy += nc[key].stats["n_windows_greedy"]
        lines += txt.count("\n")
        rounds += 1
        seed += 1
    print(json.dumps(dict(tool="tools/native_soak.py", rounds=rounds, regions=3 * rounds, windows=windows, greedy_windows=greedy, record_lines=lines, assembly_tiles=tiles, refcall_lines=refcalls,
                          identical=True, seconds=round(time.time() - t0, 1), device="caller library handed in" if lib else "MI355X")))


if __name__ == "__main__":
    main()
