"""Multi-GPU sharding of the hot path.

The reference parallelises by giving every worker process a share of the sorted region list,
`regionsForEachProcess[index % nCPU]` (runner.py:473-474), and merging the per-worker temporary VCFs with a
k-way heap merge keyed by (chromosome, position) (runner.py:301-352, key: runner.py:47-50,77-82).  Here one
process drives one GPU; regions / windows are assigned the same way (index % world), there is NO data-path
collective, and the only exchange is ONE variable-length gather of the per-region record bytes to rank 0
(sizes by all_gather, payloads point to point; RCCL over xGMI on GPUs via backend "nccl", gloo on CPU for the tests)
followed by the same ordered merge.
"""
import heapq

import numpy as np


def regions_for_rank(n_regions, rank, world):
    """Indices of the regions owned by `rank` (runner.py:473-474: round-robin over the sorted region list)."""
    return list(range(rank, n_regions, world))


def chrom_key(chrom):
    """Sort key of a chromosome name as in runner.py:47-50: integer if it looks like one, else the string."""
    try:
        return (0, int(chrom.upper().strip("CHR")), "")
    except ValueError:
        return (1, 0, chrom)


def gather_records(payload: bytes, dist=None, device=None):
    """Gather one byte string per rank to rank 0.  Returns list[bytes] on rank 0, None elsewhere.

    Sizes travel in one all_gather(int64); each payload then goes point to point to rank 0 only (one send per rank, rank 0
    posts the matching receives at their exact sizes) -- no rank but 0 ever holds another rank's records, and nothing is padded
    to the largest payload.  With a process group of one rank the size exchange still runs (it is the whole collective
    then).  `device`: where the staging tensors live -- the GPU for backend "nccl" (RCCL), None = CPU for gloo."""
    import torch
    if dist is None or not dist.is_initialized():
        return [payload]
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = device if device is not None else torch.device("cpu")
    n = torch.tensor([len(payload)], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    mine = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(dev) if payload else torch.zeros(0, dtype=torch.uint8, device=dev)
    if rank != 0:
        if sizes[rank] > 0:
            dist.send(mine, dst=0)
        return None
    out = [payload]
    bufs = [torch.empty(sizes[r], dtype=torch.uint8, device=dev) for r in range(1, world)]
    reqs = [dist.irecv(b, src=r) for r, b in zip(range(1, world), bufs) if b.numel() > 0]
    for q in reqs:
        q.wait()
    out += [bytes(b.cpu().numpy().tobytes()) for b in bufs]
    return out


def merge_record_streams(streams):
    """k-way merge of per-rank record lists, each already sorted by (chrom key, pos): runner.py:301-352.
    A record is a tuple (chrom, pos, line)."""
    keyed = [[(chrom_key(c), p, i, line) for i, (c, p, line) in enumerate(s)] for s in streams]
    return [line for _, _, _, line in heapq.merge(*keyed)]


def format_window_records(hb, logl, windows=None, chrom="1"):
    """Deterministic per-window text records (chrom, window start, H, genotype log-likelihoods rounded to 2 dp as the
    VCF GL field is, vcfutils.pyx): the gather payload of the likelihood-only configurations (the region pipeline gathers real VCF
    record lines, records_from_vcf_text)."""
    out = []
    ws = range(hb.n_windows) if windows is None else windows
    for w in ws:
        H = int(hb.win_hap_begin[w + 1] - hb.win_hap_begin[w])
        G = H * (H + 1) // 2
        o = int(hb.gl_off[w])
        gls = ",".join("%.2f" % v for v in logl[o:o + G * hb.n_ind])
        out.append((chrom, int(hb.win_start[w]), "%s\t%d\t%d\t%s" % (chrom, int(hb.win_start[w]), H, gls)))
    return out


def records_from_vcf_text(text):
    """(chrom, 0-based position, line) per VCF record line: the unit the ordered merge works on (runner.py:77-82)."""
    out = []
    for line in text.split("\n"):
        if line and not line.startswith("#"):
            chrom, pos = line.split("\t", 2)[:2]
            out.append((chrom, int(pos) - 1, line))
    return out


def encode_records(records):
    return "\n".join("%s\x1f%d\x1f%s" % r for r in records).encode()


def decode_records(payload):
    if not payload:
        return []
    out = []
    for ln in payload.decode().split("\n"):
        c, p, line = ln.split("\x1f")
        out.append((c, int(p), line))
    return out
