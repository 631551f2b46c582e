"""`python -m platypus_amd callVariants ...` -- the reference's entry point name (Platypus.py:23-46).

BAM/CRAM/FASTA I/O (htslib) is outside the hot-path scope of this build; without
`--synthetic` the command explains that instead of silently doing something else."""
import json
import sys

from .options import build_parser


def call_variants(argv):
    opts = build_parser().parse_args(argv)
    if opts.HLATyping:
        sys.exit("platypus_amd: --HLATyping=1 (useMapQualCap) is not implemented on the device path")
    if opts.synthetic is None:
        sys.exit("platypus_amd: BAM/FASTA input needs the reference's htslib I/O layer, which is outside this build's "
                 "scope (see DESIGN.md, 'out of scope').  Use --synthetic=config2[:N] or the in-memory API "
                 "(platypus_amd.hostapi / include/platypus_mi355x.h).")
    from . import synth
    from .engine import Engine
    name, _, n = opts.synthetic.partition(":")
    hb = {"config1": lambda: synth.config1(), "config2": lambda: synth.config2(int(n or 10000)),
          "config5": lambda: synth.config5(int(n or 200), 100)}[name]()
    from . import sharding
    eng = Engine(0)
    db = eng.upload(hb)
    st = eng.align(db, calc_flank_score=int(opts.calculateFlankScore))      # Haplotype.alignReads for every haplotype
    eng.genotype(db)                                                         # Population.setup
    eng.em(db, 100, int(opts.useEMLikelihoods))                              # Population.call: EM + callGenotypes
    eng.synchronize()
    # per-window records (the VCF writer is outside this build's scope): chrom, window start, #haplotypes, genotype
    # log-likelihoods, then EM haplotype frequencies and the called genotype index per sample
    recs = sharding.format_window_records(hb, db.logl.cpu().numpy())
    freq, calls = db.freq.cpu().numpy(), db.calls.cpu().numpy().reshape(hb.n_windows, hb.n_ind)
    with open(opts.output, "w") as f:
        for w, (chrom, pos, line) in enumerate(recs):
            fr = ",".join("%.4f" % v for v in freq[hb.win_hap_begin[w]:hb.win_hap_begin[w + 1]])
            f.write("%s\t%s\t%s\n" % (line, fr, ",".join(str(int(c)) for c in calls[w])))
    print(json.dumps(dict(windows=hb.n_windows, pairs=int(st.n_pairs), dp_reference=int(st.n_dp_reference),
                          dp_launched=int(st.n_dp_launched), output=opts.output)))


def main():
    if len(sys.argv) < 2 or sys.argv[1] != "callVariants":
        sys.exit("usage: python -m platypus_amd callVariants [options]   (options as in Platypus.py callVariants)")
    call_variants(sys.argv[2:])


if __name__ == "__main__":
    main()
