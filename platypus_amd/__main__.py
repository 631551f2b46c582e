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
    if name == "config4":
        return call_variants_config4(opts, int(n or 8))
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


def call_variants_config4(opts, n_regions):
    """BASELINE config 4 in miniature: procedurally generated regions with reads -> candidates -> windows -> records, every
    device stage batched over all windows (platypus_amd.caller.callVariantsInRegions).  Under torch.distributed.run the regions
    are dealt round-robin to the ranks (runner.py:473-474), each rank drives its own GPU, and the record lines are gathered to
    rank 0 and merged in (chromosome, position) order (runner.py:301-352).  VCF records (no header) go to --output."""
    import io
    import os
    import time
    from . import caller, hostapi as H, sharding, synth
    from .vcfrecords import VCF
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        gpu = torch.cuda.is_available()
        # (PLAT_DIST_BACKEND=gloo: ranks sharing one GPU, as the single-GPU test box needs; RCCL wants one device per rank)
        dist.init_process_group(os.environ.get("PLAT_DIST_BACKEND", "nccl" if gpu else "gloo"), rank=rank, world_size=world)
        if gpu:
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank)) % torch.cuda.device_count())
    opts.originalMaxHaplotypes = opts.maxHaplotypes
    size = min(int(opts.bufferSize), 100000)
    mine = sharding.regions_for_rank(n_regions, rank, world)
    t0 = time.time()
    text = io.StringIO()
    if os.environ.get("PLAT_PYTHON_CALLER") == "1" or opts.assemble or not opts.getVariantsFromBAMs:
        # the Python region loop (platypus_amd.caller): window lists, haplotypes and records as Python objects
        regs = [synth.config4_region(i, region_len=size, n_samples=1, read_len=int(opts.rlen)) for i in mine]
        fasta = H.FastaFile({r["chrom"]: r["ref"] for r in regs})
        work = [(r["chrom"], r["start"], r["end"],
                 [H.bamReadBuffer([H.AlignedRead(x["seq"], x["qual"], x["pos"], x["mapq"], x["flag"], end=x["end"], cigarOps=x["cigar"])
                                   for x in r["samples"][0]], sample="S1")]) for r in regs]
        t0 = time.time()
        n_windows = caller.callVariantsInRegions(work, fasta, opts, VCF(["S1"]), text) if work else 0
    else:
        # the native region loop (libplat_caller.so): reads as arrays, host threads, every device stage batched per chunk
        from . import fastcaller as F
        regs = [synth.config4_region_arrays(i, region_len=size, n_samples=1, read_len=int(opts.rlen)) for i in mine]
        rr = [F.region_from_arrays(r) for r in regs]
        local = int(os.environ.get("LOCAL_RANK", rank))
        import torch
        nc = F.NativeCaller(local % max(1, torch.cuda.device_count()), int(os.environ.get("PLAT_CALLER_WORKERS", "8")),
                            int(os.environ.get("PLAT_CALLER_CHUNK", "4")))
        t0 = time.time()
        text.write(nc.call_regions(rr, ["S1"], opts) if rr else "")
        n_windows = nc.stats["n_windows"] if rr else 0
        nc.close()
    recs = sorted(sharding.records_from_vcf_text(text.getvalue()), key=lambda r: (sharding.chrom_key(r[0]), r[1]))
    device = None
    if dist is not None and dist.get_backend() == "nccl":
        import torch
        device = torch.device("cuda", torch.cuda.current_device())
    got = sharding.gather_records(sharding.encode_records(recs), dist, device)
    dt = time.time() - t0
    if rank == 0:
        merged = sharding.merge_record_streams([sharding.decode_records(p) for p in got])
        import datetime
        vcf = VCF(["S1"])
        vcf.setheader([("fileDate", datetime.date.today()), ("source", "platypus_amd (records as Platypus_Version_0.8.1.1 writes them)"),
                       ("platypusOptions", str(dict(sorted(vars(opts).items()))))])         # variantcaller.pyx:51,942
        with open(opts.output, "w") as f:
            vcf.writeheader(f)
            f.write("".join(line + "\n" for line in merged))
        print(json.dumps(dict(regions=n_regions, region_len=size, ranks=world, windows_rank0=n_windows, records=len(merged),
                              seconds=round(dt, 3), output=opts.output)))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    if len(sys.argv) < 2 or sys.argv[1] != "callVariants":
        sys.exit("usage: python -m platypus_amd callVariants [options]   (options as in Platypus.py callVariants)")
    call_variants(sys.argv[2:])


if __name__ == "__main__":
    main()
