"""Python host for libplat_mi355x.so: device buffers (via torch, plumbing only) + the batched entry points.

`Engine` owns one `plat_ctx` (one per process / GPU, like one PlatypusSingleProcess per worker in the
reference, variantcaller.pyx:935-980).  All heavy lifting happens in the HIP library; nothing here
computes alignment scores or likelihoods on the CPU.
"""
import ctypes as C

import numpy as np

from . import _lib
from .batch import HostBatch, pad_blob


def _torch():
    import torch
    return torch


class DeviceBatch:
    """HBM image of a HostBatch (fields of plat_window_batch) + the ctypes struct pointing at it."""

    def __init__(self, hb: HostBatch, device):
        torch = _torch()
        self.host = hb
        self.device = device

        def up(a, dtype):
            return torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).to(device)
        self.t = dict(
            win_hap_begin=up(hb.win_hap_begin, np.int32), win_read_begin=up(hb.win_read_begin, np.int32),
            win_start=up(hb.win_start, np.int32), win_end=up(hb.win_end, np.int32),
            win_flank=up(hb.win_flank, np.int32), pair_off=up(hb.pair_off, np.int64),
            hap_seq=up(pad_blob(hb.hap_seq), np.uint8), hap_off=up(hb.hap_off, np.int64),
            read_seq=up(pad_blob(hb.read_seq), np.uint8), read_qual=up(pad_blob(hb.read_qual), np.uint8),
            read_off=up(hb.read_off, np.int64), read_pos=up(hb.read_pos, np.int32),
            read_end=up(hb.read_end, np.int32), read_mapq=up(hb.read_mapq, np.uint8),
            read_flags=up(hb.read_flags, np.int32), read_kind=up(hb.read_kind, np.uint8),
            seg_read_begin=up(hb.seg_read_begin, np.int32), seg_n_good=up(hb.seg_n_good, np.int32),
            gl_off=up(hb.gl_off, np.int64))
        s = _lib.WindowBatch()
        s.n_windows, s.n_haps, s.n_reads = hb.n_windows, hb.n_haps, hb.n_reads
        for name, _ in _lib.WindowBatch._fields_[4:]:
            setattr(s, name, self.t[name].data_ptr())
        self.struct = s
        self.loglik = torch.empty(max(hb.n_pairs, 1), dtype=torch.float64, device=device)
        self.score = torch.empty(max(hb.n_pairs, 1), dtype=torch.int32, device=device)
        ng = max(int(hb.gl_off[-1]), 1)
        self.gl = torch.empty(ng, dtype=torch.float64, device=device)
        self.logl = torch.empty(ng, dtype=torch.float64, device=device)
        self.gof = torch.empty(ng, dtype=torch.float64, device=device)


class Engine:
    def __init__(self, device_index=0):
        torch = _torch()
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.PlatypusDeviceError(-7, "torch reports no GPU; the HIP path has no CPU fallback", "Engine")
        self.device = torch.device("cuda", device_index)
        torch.cuda.set_device(self.device)
        ctx = C.c_void_p()
        _lib.check(self.lib.plat_ctx_create(device_index, C.byref(ctx)), "plat_ctx_create")
        self.ctx = ctx

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.plat_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(_torch().cuda.current_stream(self.device).cuda_stream)

    # ---- a1 --------------------------------------------------------------------------------------
    def dp_batch(self, haps, reads, quals, gos, lens, gapextend=3, nucprior=2):
        """Score-only fastAlignmentRoutine for padded rows (numpy in, numpy out)."""
        torch = _torch()
        n, lmax = reads.shape
        assert haps.shape == (n, lmax + 15) and gos.shape == (n, lmax + 15) and quals.shape == (n, lmax)
        d = [torch.from_numpy(pad_blob(a.reshape(-1))).to(self.device) for a in (haps, reads, quals, gos)]
        dl = torch.from_numpy(np.ascontiguousarray(lens, dtype=np.int32)).to(self.device)
        out = torch.empty(n, dtype=torch.int32, device=self.device)
        _lib.check(self.lib.plat_dp_batch(self.ctx, n, lmax, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(),
                                          d[3].data_ptr(), dl.data_ptr(), gapextend, nucprior, out.data_ptr(),
                                          self._stream()), "plat_dp_batch")
        torch.cuda.synchronize(self.device)
        return out.cpu().numpy()

    # ---- a3..a10 ---------------------------------------------------------------------------------
    def upload(self, hb: HostBatch) -> DeviceBatch:
        return DeviceBatch(hb, self.device)

    def align(self, db: DeviceBatch, want_stats=True, want_score=True, calc_flank_score=0, use_mapq_cap=0):
        """Haplotype.alignReads for every haplotype of every window.  Results stay in HBM (db.loglik)."""
        st = _lib.AlignStats()
        rc = self.lib.plat_align_window_batch(self.ctx, C.byref(db.struct), calc_flank_score, use_mapq_cap,
                                              db.loglik.data_ptr(), db.score.data_ptr() if want_score else None,
                                              C.byref(st) if want_stats else None, self._stream())
        _lib.check(rc, "plat_align_window_batch")
        return st

    # ---- a11/a12 ---------------------------------------------------------------------------------
    def genotype(self, db: DeviceBatch):
        rc = self.lib.plat_genotype_window_batch(self.ctx, C.byref(db.struct), db.host.n_ind,
                                                 db.t["seg_read_begin"].data_ptr(), db.t["seg_n_good"].data_ptr(),
                                                 db.loglik.data_ptr(), db.t["gl_off"].data_ptr(), db.gl.data_ptr(),
                                                 db.logl.data_ptr(), db.gof.data_ptr(), self._stream())
        _lib.check(rc, "plat_genotype_window_batch")

    def call_windows(self, db: DeviceBatch, want_stats=True):
        """One pass of the hot path: likelihood arrays, then genotype likelihoods (Population.setup)."""
        st = self.align(db, want_stats=want_stats)
        self.genotype(db)
        return st

    def profile_enable(self, on=True):
        _lib.check(self.lib.plat_profile_enable(self.ctx, int(on)), "plat_profile_enable")

    def profile_last(self):
        p = _lib.Profile()
        _lib.check(self.lib.plat_profile_last(self.ctx, C.byref(p)), "plat_profile_last")
        return p

    def synchronize(self):
        _torch().cuda.synchronize(self.device)
