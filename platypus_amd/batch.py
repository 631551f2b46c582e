"""Host-side (numpy, structure-of-arrays) description of a batch of calling windows, and its HBM image.

A *window* is one `callVariantsInWindow` unit of the reference (variantcaller.pyx:74-141): a list of
haplotypes (byte strings built as in chaplotype.pyx:127-191) and, per individual, the reads of
`bamReadBuffer.reads/badReads/brokenMates` between the window pointers (cwindow.pyx:208-236), in the
order `Haplotype.alignReads` walks them (good -> bad -> brokenMates, chaplotype.pyx:341-373).

The arrays are exactly the fields of `plat_window_batch` in include/platypus_mi355x.h.
"""
from dataclasses import dataclass, field

import numpy as np

from ._lib import PLAT_BLOB_PAD

BAM_FQCFAIL = 512          # htslibWrapper.pxd:243
KIND_GOOD, KIND_BAD, KIND_BROKEN = 0, 1, 2


@dataclass
class HostBatch:
    n_ind: int
    win_hap_begin: np.ndarray      # int32 [nW+1]
    win_read_begin: np.ndarray     # int32 [nW+1]
    win_start: np.ndarray          # int32 [nW]   Haplotype.startPos
    win_end: np.ndarray            # int32 [nW]   Haplotype.endPos
    win_flank: np.ndarray          # int32 [nW]   Haplotype.endBufferSize
    hap_seq: np.ndarray            # uint8 blob
    hap_off: np.ndarray            # int64 [nH+1]
    read_seq: np.ndarray           # uint8 blob
    read_qual: np.ndarray          # uint8 blob (raw phred)
    read_off: np.ndarray           # int64 [nR+1]
    read_pos: np.ndarray           # int32 [nR]
    read_end: np.ndarray           # int32 [nR]
    read_mapq: np.ndarray          # uint8 [nR]
    read_flags: np.ndarray         # int32 [nR]
    read_kind: np.ndarray          # uint8 [nR]
    seg_read_begin: np.ndarray     # int32 [nW*n_ind+1]  per (window, individual) read ranges
    seg_n_good: np.ndarray         # int32 [nW*n_ind]    number of `reads` (good) entries
    pair_off: np.ndarray = field(default=None)   # int64 [nW+1]
    gl_off: np.ndarray = field(default=None)     # int64 [nW+1]
    meta: dict = field(default_factory=dict)

    def __post_init__(self):
        H = np.diff(self.win_hap_begin).astype(np.int64)
        R = np.diff(self.win_read_begin).astype(np.int64)
        self.pair_off = np.concatenate([[0], np.cumsum(H * R)]).astype(np.int64)
        G = H * (H + 1) // 2
        self.gl_off = np.concatenate([[0], np.cumsum(G * self.n_ind)]).astype(np.int64)

    # ---- sizes ---------------------------------------------------------------------------------
    @property
    def n_windows(self):
        return len(self.win_start)

    @property
    def n_haps(self):
        return len(self.hap_off) - 1

    @property
    def n_reads(self):
        return len(self.read_off) - 1

    @property
    def n_pairs(self):
        return int(self.pair_off[-1])

    def algorithmic_bytes(self):
        """Lower-bound HBM traffic of the likelihood path ('per window, dedup'd'):
        read seq+qual once, haplotype bytes twice (sequence + gap-open), 8 bytes per output."""
        return int(2 * self.read_off[-1] + 2 * self.hap_off[-1] + 8 * self.n_pairs)

    # ---- per-window views (used by the oracle-side of the tests) ----------------------------------
    def window_haps(self, w):
        hb = self.hap_seq.tobytes()
        return [hb[self.hap_off[h]:self.hap_off[h + 1]] for h in range(self.win_hap_begin[w], self.win_hap_begin[w + 1])]

    def window_reads(self, w):
        a, b = self.win_read_begin[w], self.win_read_begin[w + 1]
        sb, qb = self.read_seq.tobytes(), self.read_qual.tobytes()
        return dict(seq=[sb[self.read_off[r]:self.read_off[r + 1]] for r in range(a, b)],
                    qual=[qb[self.read_off[r]:self.read_off[r + 1]] for r in range(a, b)],
                    pos=self.read_pos[a:b], end=self.read_end[a:b], mapq=self.read_mapq[a:b],
                    flags=self.read_flags[a:b], kind=self.read_kind[a:b])

    def subset(self, windows):
        """A new HostBatch holding only the given windows (used to test at oracle-friendly sizes)."""
        windows = list(windows)
        hs, rs = [], []
        whb, wrb, seg_b, seg_g = [0], [0], [0], []
        for w in windows:
            h0, h1 = self.win_hap_begin[w], self.win_hap_begin[w + 1]
            r0, r1 = self.win_read_begin[w], self.win_read_begin[w + 1]
            hs.extend(range(h0, h1)); rs.extend(range(r0, r1))
            whb.append(len(hs)); wrb.append(len(rs))
            for i in range(self.n_ind):
                s = w * self.n_ind + i
                seg_b.append(seg_b[-1] + int(self.seg_read_begin[s + 1] - self.seg_read_begin[s]))
                seg_g.append(int(self.seg_n_good[s]))
        hs = np.array(hs, dtype=np.int64); rs = np.array(rs, dtype=np.int64)

        def gather_blob(blob, off, idx):
            lens = (off[idx + 1] - off[idx]) if len(idx) else np.zeros(0, dtype=np.int64)
            noff = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
            out = np.empty(int(noff[-1]), dtype=np.uint8)
            for k, i in enumerate(idx):
                out[noff[k]:noff[k + 1]] = blob[off[i]:off[i + 1]]
            return out, noff
        hseq, hoff = gather_blob(self.hap_seq, self.hap_off, hs)
        rseq, roff = gather_blob(self.read_seq, self.read_off, rs)
        rqual, _ = gather_blob(self.read_qual, self.read_off, rs)
        wi = np.array(windows, dtype=np.int64)
        return HostBatch(self.n_ind, np.array(whb, dtype=np.int32), np.array(wrb, dtype=np.int32),
                         self.win_start[wi], self.win_end[wi], self.win_flank[wi], hseq, hoff, rseq, rqual, roff,
                         self.read_pos[rs], self.read_end[rs], self.read_mapq[rs], self.read_flags[rs],
                         self.read_kind[rs], np.array(seg_b, dtype=np.int32), np.array(seg_g, dtype=np.int32),
                         meta=dict(self.meta))


def pad_blob(a):
    return np.concatenate([np.ascontiguousarray(a, dtype=np.uint8), np.zeros(PLAT_BLOB_PAD, dtype=np.uint8)])
