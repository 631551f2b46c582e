"""Where the time of the batched region pipeline (platypus_amd.caller.callVariantsInRegions) goes: wall time per stage on
config-4 regions.  usage: python tools/region_pipeline_times.py [n_regions] [region_len]"""
import io
import json
import sys
import time

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from platypus_amd import caller, hostapi as H, synth, vcfrecords            # noqa: E402
from platypus_amd.options import default_options                            # noqa: E402

acc = {}


def timed(mod, name, label):
    f = getattr(mod, name)

    def g(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc[label] = acc.get(label, 0.0) + time.perf_counter() - t
    setattr(mod, name, g)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
    t = time.perf_counter()
    regs = [synth.config4_region(i, region_len=size) for i in range(n)]
    fasta = H.FastaFile({r["chrom"]: r["ref"] for r in regs})
    work = [(r["chrom"], r["start"], r["end"],
             [H.bamReadBuffer([H.AlignedRead(x["seq"], x["qual"], x["pos"], x["mapq"], x["flag"], end=x["end"], cigarOps=x["cigar"])
                               for x in r["samples"][0]], sample="S1")]) for r in regs]
    t_synth = time.perf_counter() - t
    H.get_engine()
    timed(caller, "generateVariantsInRegions", "1 candidates (device scan + host merge / normalise / filter)")
    timed(caller, "_prepareWindow", "2 windows: pointers, haplotype enumeration, merge (host)")
    timed(caller, "callWindowsBatched", "3 batched device stages + INFO/FILTER dictionaries")
    timed(caller, "outputCallToVCF", "4 record text (host)")
    eng = H.get_engine()
    for name in ("upload", "call_windows", "haplotype_scores", "em", "variant_posteriors", "variant_read_stats", "genotype_calls", "candidates"):
        timed(eng, name, "  engine." + name)
    for rep in range(2):                                                    # second pass: warm scratch buffers
        acc.clear()
        opts = default_options()
        out = io.StringIO()
        t = time.perf_counter()
        nw = caller.callVariantsInRegions(work, fasta, opts, vcfrecords.VCF(["S1"]), out)
        total = time.perf_counter() - t
    recs = out.getvalue().count("\n")
    print(json.dumps(dict(regions=n, region_len=size, reads=sum(len(r["samples"][0]) for r in regs), windows=nw, records=recs,
                          planted=sum(len(r["variants"]) for r in regs), synth_s=round(t_synth, 2), total_s=round(total, 3),
                          windows_per_s=round(nw / total, 1), stages={k: round(v, 3) for k, v in sorted(acc.items())}), indent=1))


if __name__ == "__main__":
    main()
