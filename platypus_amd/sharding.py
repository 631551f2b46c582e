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


class RegionTextExchange:
    """The job's ONE exchange for the native region loop's output (runner.py:301-352): every rank's record text to rank 0,
    merged there by (chromosome key, position) -- as a permutation of whole REGION blocks.

    A rank's text is its regions' record lines back to back, each region's in order; regions of a job list do not overlap, so the merged
    text is the regions' blocks in (chromosome key, start) order and no line needs to be parsed.  That order depends on the region list
    alone: it is worked out once, here, before anything is timed.  Per call: the regions' byte counts travel in one all_gather, the texts
    point to point to rank 0 (RCCL: device to device over xGMI), and rank 0 puts the blocks in place --
      * backend "nccl": ONE scatter kernel over the texts where they arrived, in HBM (plat_copy_pieces), then one copy of the merged text
        to (cached) pinned host memory;
      * gloo / no process group: block copies on up to 16 host threads (plat_merge_region_blocks).
    When regions DO overlap the texts are merged line by line instead (fastcaller.merge_record_texts), as before.

    regions_of_rank[r] = [(chrom, start, end), ...] in rank r's list order (the same on every rank)."""

    def __init__(self, regions_of_rank, dist=None, device=None, lib=None, device_index=0):
        from . import fastcaller as F
        self.F, self.lib, self.dist = F, lib, dist if (dist is not None and dist.is_initialized()) else None
        self.world = self.dist.get_world_size() if self.dist else 1
        self.rank = self.dist.get_rank() if self.dist else 0
        self.device = device
        self.on_device = self.dist is not None and device is not None and getattr(device, "type", "cpu") == "cuda"
        self.counts = [len(x) for x in regions_of_rank]
        self.plan = F.BlockOrder([[(chrom_key(c), int(s), int(e)) for c, s, e in regs] for regs in regions_of_rank])
        self.ctx = None
        self._pinned = None                                              # this exchange's pinned host block for the merged text (grown, never shrunk)
        if self.on_device and self.rank == 0:
            import ctypes as C
            from . import _lib
            self._dl = _lib.load()
            ctx = C.c_void_p()
            if self._dl.plat_ctx_create(int(device_index), C.byref(ctx)) != 0:
                raise RuntimeError("plat_ctx_create failed")
            self.ctx = ctx

    def close(self):
        """Give the device context back (rank 0 of an "nccl" job holds one for the scatter kernel)."""
        ctx, self.ctx = self.ctx, None
        if ctx is not None:
            try:
                self._dl.plat_ctx_destroy(ctx)
            except Exception:                                            # pragma: no cover  (interpreter shutdown)
                pass
        self._pinned = None

    def __del__(self):
        self.close()

    def exchange(self, text, lengths):
        """text: this rank's record text (bytes / raw view); lengths: int64 bytes per region of this rank (NativeCaller.region_text_lengths).
        Returns the merged text on rank 0 (a buffer: bytes(x) / memoryview(x) give its bytes), None elsewhere.  Under "nccl" the buffer is
        a view of this object's pinned block: it holds the text until the NEXT exchange() of the same object (copy it to keep it longer)."""
        import torch
        F = self.F
        lengths = np.ascontiguousarray(lengths, dtype=np.int64)
        if self.dist is None:
            if not self.plan.ok:
                return F.merge_record_texts([text], lib=self.lib, raw="view")
            return self.plan.merge([text], [lengths], lib=self.lib)
        dev = self.device if self.on_device else torch.device("cpu")
        addr, size = F.text_address(text)
        mine = torch.from_numpy(np.ctypeslib.as_array((__import__("ctypes").c_uint8 * size).from_address(addr)) if size else np.zeros(0, dtype=np.uint8))
        lens_t = torch.from_numpy(lengths)
        if self.on_device:
            mine, lens_t = mine.to(dev), lens_t.to(dev)
        # sizes are implied by the regions' byte counts: one all_gather of them (padded to the longest list)
        most = max(self.counts) if self.counts else 0
        pad = torch.zeros(most, dtype=torch.int64, device=dev)
        pad[:len(lengths)] = lens_t
        allp = [torch.zeros(most, dtype=torch.int64, device=dev) for _ in range(self.world)]
        self.dist.all_gather(allp, pad)
        all_lens = [allp[r][:self.counts[r]].cpu().numpy() for r in range(self.world)]
        sizes = [int(x.sum()) for x in all_lens]
        if self.rank != 0:
            if sizes[self.rank] > 0:
                self.dist.send(mine, dst=0)
            return None
        bufs = [mine] + [torch.empty(sizes[r], dtype=torch.uint8, device=dev) for r in range(1, self.world)]
        reqs = [self.dist.irecv(bufs[r], src=r) for r in range(1, self.world) if sizes[r] > 0]
        for q in reqs:
            q.wait()
        if not self.plan.ok:
            return F.merge_record_texts([bytes(b.cpu().numpy().tobytes()) for b in bufs], lib=self.lib, raw="view")
        if not self.on_device:
            return self.plan.merge([b.numpy() for b in bufs], all_lens, lib=self.lib)
        return self._merge_on_device(bufs, all_lens)

    def _merge_on_device(self, bufs, all_lens):
        import ctypes as C
        import torch
        plan = self.plan
        n = len(plan.rank)
        src = np.zeros(n, dtype=np.int64)
        ln = np.zeros(n, dtype=np.int64)
        for r, lens in enumerate(all_lens):
            st = np.zeros(len(lens), dtype=np.int64)
            if len(lens):
                st[1:] = np.cumsum(lens)[:-1]
            m = plan.rank == r
            src[m] = int(bufs[r].data_ptr()) + st[plan.index[m]]
            ln[m] = lens[plan.index[m]]
        at = np.zeros(n, dtype=np.int64)
        if n:
            at[1:] = np.cumsum(ln)[:-1]
        total = int(ln.sum())
        keep = ln > 0
        pieces = np.stack([src[keep], at[keep], ln[keep]], axis=1).astype(np.int64)          # plat_unpack_piece {src, dst, n}
        out = torch.empty(total + 64, dtype=torch.uint8, device=self.device)
        if len(pieces):
            pd = torch.from_numpy(np.ascontiguousarray(pieces)).to(self.device)
            stream = torch.cuda.current_stream(self.device).cuda_stream
            rc = self._dl.plat_copy_pieces(self.ctx, int(len(pieces)), int(ln.max()), C.c_void_p(pd.data_ptr()), C.c_void_p(out.data_ptr()), C.c_void_p(stream))
            if rc != 0:
                raise RuntimeError("plat_copy_pieces failed (%d)" % rc)
        if self._pinned is None or self._pinned.numel() < total:
            self._pinned = torch.empty(int(total * 1.25) + 64, dtype=torch.uint8).pin_memory()
        host = self._pinned[:total]
        host.copy_(out[:total], non_blocking=True)
        torch.cuda.synchronize(self.device)
        return host.numpy()


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
