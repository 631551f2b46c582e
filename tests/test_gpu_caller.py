"""The region pipeline (candidates -> windows -> haplotypes -> likelihoods / EM / posteriors -> VCF records) on synthetic
config-4 regions: window by window as the reference walks them, and with every device stage batched over all windows of all
regions -- the two must write the same text -- and the planted variants must come back."""
import io

import numpy as np
import pytest

from platypus_amd import caller, hostapi as H, synth
from platypus_amd.options import default_options
from platypus_amd.vcfrecords import VCF

pytestmark = pytest.mark.gpu


def _regions(n, n_samples, **kw):
    regs = [synth.config4_region(i, n_samples=n_samples, **kw) for i in range(n)]
    fasta = H.FastaFile({r["chrom"]: r["ref"] for r in regs})
    names = ["S%d" % (i + 1) for i in range(n_samples)]

    def buffers(r):
        return [H.bamReadBuffer([H.AlignedRead(x["seq"], x["qual"], x["pos"], x["mapq"], x["flag"], end=x["end"], cigarOps=x["cigar"])
                                 for x in reads], sample=names[i]) for i, reads in enumerate(r["samples"])]
    return regs, fasta, names, buffers


def test_region_pipeline_batched_equals_window_by_window_and_finds_the_planted_variants():
    regs, fasta, names, buffers = _regions(4, 2, region_len=2500, snp_rate=4e-3, indel_rate=1.5e-3, read_len=100, depth=40)
    one = io.StringIO()
    for r in regs:
        opts = default_options()
        caller.callVariantsInRegion(r["chrom"], r["start"], r["end"], buffers(r), fasta, opts, VCF(names), one)
    opts = default_options()
    many = io.StringIO()
    n_windows = caller.callVariantsInRegions([(r["chrom"], r["start"], r["end"], buffers(r)) for r in regs], fasta, opts, VCF(names), many)
    assert one.getvalue() == many.getvalue()
    lines = many.getvalue().split("\n")[:-1]
    assert n_windows >= 20 and len(lines) >= 20
    called = {}
    for ln in lines:
        f = ln.split("\t")
        called[(f[0], int(f[1]))] = f
        assert len(f) == 9 + len(names) and f[8] == "GT:GL:GOF:GQ:NR:NV"
    planted = [(r["chrom"], p, rem, add) for r in regs for p, rem, add in r["variants"]]
    snps = [(c, p) for c, p, rem, add in planted if len(rem) == len(add)]
    found = sum((c, p + 1) in called for c, p in snps)
    # a planted variant is absent from both haplotypes of both samples with probability 1/16
    assert found >= 0.85 * len(snps), (found, len(snps))
    assert sum(f[6] == "PASS" for f in called.values()) >= 0.6 * len(called)
    indels = [(c, p) for c, p, rem, add in planted if len(rem) != len(add)]
    near = sum(any(cc == c and abs(pp - p) <= 12 and len(f[3]) != len(f[4].split(",")[0]) for (cc, pp), f in called.items()) for c, p in indels)
    assert near >= 0.6 * len(indels), (near, len(indels))


def test_batched_caller_handles_samples_without_reads_and_empty_regions():
    regs, fasta, names, buffers = _regions(2, 2, region_len=1200, snp_rate=5e-3, indel_rate=0, read_len=100, depth=30)
    bufs = [buffers(r) for r in regs]
    bufs[0][1] = H.bamReadBuffer([], sample=names[1])                        # sample 2 has no data in region 0
    out = io.StringIO()
    opts = default_options()
    caller.callVariantsInRegions([(r["chrom"], r["start"], r["end"], b) for r, b in zip(regs, bufs)] +
                                 [(regs[0]["chrom"], 10, 60, [H.bamReadBuffer([], sample=s) for s in names])], fasta, opts, VCF(names), out)
    lines = out.getvalue().split("\n")[:-1]
    assert lines and all(ln.split("\t")[10].startswith("./.") for ln in lines if ln.startswith(regs[0]["chrom"] + "\t"))


def test_cli_config4_two_ranks_equals_one_rank(tmp_path):
    """`callVariants --synthetic config4:N`: the same VCF records from one process and from two ranks (regions dealt round-robin,
    record lines gathered to rank 0 and merged in (chromosome, position) order) -- the two ranks share the one GPU here."""
    import json, os, socket, subprocess, sys
    one, two = tmp_path / "one.vcf", tmp_path / "two.vcf"
    base = ["callVariants", "--synthetic", "config4:5", "--bufferSize", "4000"]
    r = subprocess.run([sys.executable, "-m", "platypus_amd"] + base + ["--output", str(one)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-800:]
    info = json.loads(r.stdout.strip().splitlines()[-1])
    assert info["records"] > 10
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, PLAT_DIST_BACKEND="gloo")        # both ranks drive GPU 0; RCCL needs one device per rank
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), "-m", "platypus_amd"] + base + ["--output", str(two)], capture_output=True, text=True, env=env)
    assert r.returncode == 0, (r.stderr[-1500:], r.stdout[-500:])
    assert one.read_text() == two.read_text()
