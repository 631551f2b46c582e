"""Exercise the kernels of the SURVEY 8(f) "next" rows at bench-like sizes so that `rocprofv3 --kernel-trace --stats` shows their
per-launch times:  EM / posteriors / genotype marginalisation / HapScore on BASELINE config 2 (10 000 windows) and config 5
(200 windows x 100 samples); candidate scan, read QC and INFO read statistics on config-4 regions (4 x 100 kb, 80 000 reads)."""
import io
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from platypus_amd import caller, hostapi as H, synth, vcfrecords     # noqa: E402
from platypus_amd.options import default_options                     # noqa: E402


def population_kernels(eng, hb, reps=3):
    db = eng.upload(hb)
    eng.call_windows(db, want_stats=False)
    nH = np.diff(hb.win_hap_begin)
    nvar = np.log2(nH).astype(int)
    var_w, masks, priors, sites = [], [], [], []
    for w in range(hb.n_windows):
        Hw = int(nH[w])
        for k in range(int(nvar[w])):
            var_w.append(w); masks.append(((np.arange(Hw) >> k) & 1).astype(np.uint8)); priors.append(1e-3 / 3)
        vih = ((np.arange(Hw)[:, None] >> np.arange(1)[None, :]) & 1).astype(np.int32)              # first SNP of the window
        sites.append(dict(window=w, var_in_hap=vih, is_ref=(vih[:, 0] == 0).astype(np.int32)))
    for _ in range(reps):
        eng.haplotype_scores(db)
        eng.em(db, 100, 0)
        eng.variant_posteriors(db, var_w, masks, priors)
        eng.genotype_calls(db, sites)
    eng.synchronize()
    return dict(windows=hb.n_windows, samples=hb.n_ind, haplotypes=hb.n_haps, variants=len(var_w), sites=len(sites))


def main():
    eng = H.get_engine()
    out = {"config2": population_kernels(eng, synth.config2(10000)), "config5": population_kernels(eng, synth.config5(200, 100))}
    regs = [synth.config4_region(i, region_len=100000) for i in range(4)]
    fasta = H.FastaFile({r["chrom"]: r["ref"] for r in regs})
    rd = lambda x: H.AlignedRead(x["seq"], x["qual"], x["pos"], x["mapq"], x["flag"], end=x["end"], cigarOps=x["cigar"])
    work = [(r["chrom"], r["start"], r["end"], [H.bamReadBuffer([rd(x) for x in r["samples"][0]], sample="S1")]) for r in regs]
    for _ in range(3):
        H.checkAndTrimReads([rd(x) for r in regs for x in r["samples"][0]], default_options())
        n = caller.callVariantsInRegions(work, fasta, default_options(), vcfrecords.VCF(["S1"]), io.StringIO())
    out["config4"] = dict(regions=4, reads=sum(len(r["samples"][0]) for r in regs), windows=n)
    print(out)


if __name__ == "__main__":
    main()
