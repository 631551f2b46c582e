"""Soak of the region pipeline: window-by-window vs batched text equality over many synthetic regions / option variants.
usage: python tools/caller_soak.py [seconds]"""
import io
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from platypus_amd import caller, hostapi as H, synth      # noqa: E402
from platypus_amd.options import default_options          # noqa: E402
from platypus_amd.vcfrecords import VCF                    # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    t0 = time.time()
    rounds = lines = windows = 0
    seed = 70000
    while time.time() - t0 < budget:
        ns = 1 + seed % 3
        kw = dict(region_len=1500 + 500 * (seed % 4), n_samples=ns, snp_rate=3e-3 + 1e-3 * (seed % 3), indel_rate=1e-3 * (seed % 4),
                  read_len=[100, 150, 76][seed % 3], depth=[20, 35, 50][seed % 3])
        regs = [synth.config4_region(i, seed=seed, **kw) for i in range(3)]
        fasta = H.FastaFile({r["chrom"]: r["ref"] for r in regs})
        names = ["S%d" % (i + 1) for i in range(ns)]
        mk = lambda r: [H.bamReadBuffer([H.AlignedRead(x["seq"], x["qual"], x["pos"], x["mapq"], x["flag"], end=x["end"], cigarOps=x["cigar"])
                                         for x in rd], sample=names[i]) for i, rd in enumerate(r["samples"])]
        over = dict(maxVariants=[8, 3, 8][seed % 3], mergeClusteredVariants=int(seed % 5 != 0), minPosterior=[5, 0, 5, 20][seed % 4],
                    assemble=int(seed % 7 == 0), countOnlyExactIndelMatches=seed % 2)
        one = io.StringIO()
        for r in regs:
            caller.callVariantsInRegion(r["chrom"], r["start"], r["end"], mk(r), fasta, default_options(**over), VCF(names), one)
        many = io.StringIO()
        windows += caller.callVariantsInRegions([(r["chrom"], r["start"], r["end"], mk(r)) for r in regs], fasta, default_options(**over), VCF(names), many)
        if one.getvalue() != many.getvalue():
            print("MISMATCH at seed", seed)
            sys.exit(1)
        lines += many.getvalue().count("\n")
        rounds += 1
        seed += 1
    print(json.dumps(dict(rounds=rounds, regions=3 * rounds, windows=windows, record_lines=lines, identical=True, seconds=round(time.time() - t0, 1))))


if __name__ == "__main__":
    main()
