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
        # what the host knows about its own batch: lets the asynchronous entry point skip the internal read-backs
        hl = np.diff(hb.hap_off)
        rl = np.diff(hb.read_off)
        h = _lib.BatchHints()
        h.max_hap_len = int(hl.max()) if len(hl) else 0
        h.max_read_len = int(rl.max()) if len(rl) else 0
        h.max_reads_per_window = int(np.diff(hb.win_read_begin).max()) if hb.n_windows else 0
        h.n_pairs, h.hap_blob_len, h.read_blob_len, h.extra_jobs_cap = int(hb.n_pairs), int(hb.hap_off[-1]), int(hb.read_off[-1]), 0
        self.hints = h
        self.loglik = torch.empty(max(hb.n_pairs, 1), dtype=torch.float64, device=device)
        self.score = torch.empty(max(hb.n_pairs, 1), dtype=torch.int32, device=device)
        ng = max(int(hb.gl_off[-1]), 1)
        self.gl = torch.empty(ng, dtype=torch.float64, device=device)
        self.logl = torch.empty(ng, dtype=torch.float64, device=device)
        self.gof = torch.empty(ng, dtype=torch.float64, device=device)


class LikelihoodBatch:
    """HBM image of genotype likelihoods that were computed elsewhere (e.g. handed over by a caller that kept
    Population.setup on its side): what em() / variant_posteriors() / genotype_calls() need of a DeviceBatch.

    hap_counts[w] = H_w; n_reads [nW][n_ind]; gl[w] = [n_ind][G_w]; gof[w] = [G_w][n_ind] (optional)."""

    class _Host:
        pass

    def __init__(self, n_ind, hap_counts, n_reads, gl, gof, device):
        torch = _torch()
        hb = self.host = LikelihoodBatch._Host()
        hb.n_ind, hb.n_windows = int(n_ind), len(hap_counts)
        hb.win_hap_begin = np.concatenate([[0], np.cumsum(hap_counts)]).astype(np.int32)
        hb.n_haps = int(hb.win_hap_begin[-1])
        G = np.asarray(hap_counts, dtype=np.int64) * (np.asarray(hap_counts, dtype=np.int64) + 1) // 2
        hb.gl_off = np.concatenate([[0], np.cumsum(G * n_ind)]).astype(np.int64)

        def up(a, dt):
            return torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(device)
        self.t = dict(win_hap_begin=up(hb.win_hap_begin, np.int32), gl_off=up(hb.gl_off, np.int64),
                      seg_n_good=up(np.asarray(n_reads).reshape(-1), np.int32))
        self.gl = up(np.concatenate([np.asarray(x, dtype=np.float64).reshape(-1) for x in gl] + [np.zeros(1)]), np.float64)
        if gof is None:
            gof = [np.zeros(int(g) * n_ind) for g in G]
        self.gof = up(np.concatenate([np.asarray(x, dtype=np.float64).reshape(-1) for x in gof] + [np.zeros(1)]), np.float64)


class AssemblyDeviceBatch:
    """HBM image of a plat_assembly_batch + the output buffers of plat_assemble_batch."""

    def __init__(self, ab, device, max_vars=512, blob_per_region=1 << 16):
        torch = _torch()

        def dev(a, dt):
            return torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(device)
        self.device, self.max_vars, self.blob_per_region = device, max_vars, blob_per_region
        nG = self.n_regions = int(ab["n_regions"])
        self.t = dict(ref_seq=dev(pad_blob(ab["ref_seq"]), np.uint8), ref_off=dev(ab["ref_off"], np.int64),
                      ref_start=dev(ab["ref_start"], np.int32), assem_start=dev(ab["assem_start"], np.int32),
                      assem_end=dev(ab["assem_end"], np.int32), reg_read_begin=dev(ab["reg_read_begin"], np.int32),
                      read_seq=dev(pad_blob(ab["read_seq"]), np.uint8), read_qual=dev(pad_blob(ab["read_qual"]), np.uint8),
                      read_off=dev(ab["read_off"], np.int64))
        s = _lib.AssemblyBatch()
        s.n_regions, s.n_reads = nG, int(ab["n_reads"])
        for name, _ in _lib.AssemblyBatch._fields_[2:]:
            setattr(s, name, self.t[name].data_ptr())
        self.struct = s
        i32 = dict(dtype=torch.int32, device=device)
        self.cnt = torch.zeros(nG, **i32); self.status = torch.zeros(nG, **i32)
        self.pos = torch.zeros(nG * max_vars, **i32); self.nrem = torch.zeros(nG * max_vars, **i32)
        self.nadd = torch.zeros(nG * max_vars, **i32); self.off = torch.zeros(nG * max_vars, **i32)
        self.blob = torch.zeros(nG * blob_per_region, dtype=torch.uint8, device=device)

    def results(self):
        """[syncs] per region the list of (pos, removed, added) in the reference's sorted() order."""
        if self.device.type == "cuda":
            _torch().cuda.synchronize(self.device)
        cnt, status, pos, nrem, nadd, off = (x.cpu().numpy() for x in (self.cnt, self.status, self.pos, self.nrem, self.nadd, self.off))
        blob = self.blob.cpu().numpy()
        out = []
        for g in range(self.n_regions):
            if status[g] != 0:
                _lib.check(int(status[g]), "plat_assemble_batch(region %d)" % g)
            vs = []
            if cnt[g]:
                raw = blob[g * self.blob_per_region:(g + 1) * self.blob_per_region].tobytes()
                for i in range(cnt[g]):
                    k = g * self.max_vars + i
                    o = off[k]
                    vs.append((int(pos[k]), raw[o:o + nrem[k]], raw[o + nrem[k]:o + nrem[k] + nadd[k]]))
            out.append(vs)
        return out


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

    def _sync(self):
        _torch().cuda.synchronize(self.device)

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
        self._sync()
        return out.cpu().numpy()

    # ---- a3..a10 ---------------------------------------------------------------------------------
    def upload(self, hb: HostBatch) -> DeviceBatch:
        return DeviceBatch(hb, self.device)

    def align_async(self, db: DeviceBatch, want_score=True, calc_flank_score=0, use_mapq_cap=0, hints=None):
        """Same as align(), enqueued without any internal read-back (sizes from db.hints).  Device-side errors are
        raised by the next synchronize()."""
        rc = self.lib.plat_align_window_batch_async(self.ctx, C.byref(db.struct), C.byref(hints if hints is not None else db.hints),
                                                    calc_flank_score, use_mapq_cap, db.loglik.data_ptr(),
                                                    db.score.data_ptr() if want_score else None, self._stream())
        _lib.check(rc, "plat_align_window_batch_async")

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

    def call_windows(self, db: DeviceBatch, want_stats=True, asynchronous=False, calc_flank_score=0):
        """One pass of the hot path: likelihood arrays, then genotype likelihoods (Population.setup).
        asynchronous=True: nothing is read back and nothing waits (no statistics; errors surface in synchronize()).
        calc_flank_score = options.calculateFlankScore (chaplotype.pyx:606-612)."""
        if asynchronous and not want_stats:
            self.align_async(db, calc_flank_score=calc_flank_score)
            st = None
        else:
            st = self.align(db, want_stats=want_stats, calc_flank_score=calc_flank_score)
        self.genotype(db)
        return st

    # ---- EM, genotype calls, posteriors, per-site marginalisation -------------------
    def upload_likelihoods(self, n_ind, hap_counts, n_reads, gl, gof=None) -> LikelihoodBatch:
        return LikelihoodBatch(n_ind, hap_counts, n_reads, gl, gof, self.device)

    def haplotype_scores(self, db: DeviceBatch):
        """computeHaplotypeScore (INFO['HapScore']) for every window of `db` from the likelihoods left in HBM by align().
        Returns (hap_like [n_haps] = what DiploidGenotype.hap1Like holds after Population.setup, hap_score [n_windows])."""
        torch = _torch()
        hb = db.host
        like = torch.empty(max(hb.n_haps, 1), dtype=torch.float64, device=self.device)
        score = torch.empty(max(hb.n_windows, 1), dtype=torch.int32, device=self.device)
        maxh = int(np.max(np.diff(hb.win_hap_begin))) if hb.n_windows else 0
        rc = self.lib.plat_haplotype_score_batch(self.ctx, C.byref(db.struct), hb.n_ind, maxh, db.t["seg_read_begin"].data_ptr(),
                                                 db.t["seg_n_good"].data_ptr(), db.loglik.data_ptr(), like.data_ptr(),
                                                 score.data_ptr(), self._stream())
        _lib.check(rc, "plat_haplotype_score_batch")
        self._sync()
        return like.cpu().numpy()[:hb.n_haps], score.cpu().numpy()[:hb.n_windows]

    def em(self, db, max_iters=100, use_em_likelihoods=0):
        """Population.call (EM + callGenotypes) for every window of `db`, on the genotype likelihoods left in HBM by
        genotype().  Results stay in HBM: db.freq [n_haps], db.em [like db.gl], db.calls [n_windows*n_ind], db.em_iters."""
        torch = _torch()
        hb = db.host
        db.freq = torch.empty(max(hb.n_haps, 1), dtype=torch.float64, device=self.device)
        db.em = torch.empty_like(db.gl)
        db.calls = torch.empty(max(hb.n_windows * hb.n_ind, 1), dtype=torch.int32, device=self.device)
        db.em_iters = torch.empty(max(hb.n_windows, 1), dtype=torch.int32, device=self.device)
        maxh = int(np.max(np.diff(hb.win_hap_begin))) if hb.n_windows else 0
        db.max_haps = maxh
        rc = self.lib.plat_em_window_batch(self.ctx, hb.n_windows, hb.n_ind, maxh, db.t["win_hap_begin"].data_ptr(),
                                           db.t["gl_off"].data_ptr(), db.t["seg_n_good"].data_ptr(), db.gl.data_ptr(),
                                           max_iters, use_em_likelihoods, db.freq.data_ptr(), db.em.data_ptr(),
                                           db.calls.data_ptr(), db.em_iters.data_ptr(), self._stream())
        _lib.check(rc, "plat_em_window_batch")

    def variant_posteriors(self, db, var_window, hap_masks, priors):
        """Population.calculatePosterior for a list of variants: var_window[v] = window, hap_masks[v] = 0/1 per haplotype
        of that window (`var in hap.variants`), priors[v].  Needs em().  Returns a numpy array of phred posteriors."""
        torch = _torch()
        n = len(var_window)
        if n == 0:
            return np.zeros(0)
        hb = db.host
        off = np.concatenate([[0], np.cumsum([len(m) for m in hap_masks])]).astype(np.int64)
        blob = np.concatenate([np.asarray(m, dtype=np.uint8) for m in hap_masks])

        def up(a, dt):
            return torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(self.device)
        d_w, d_off, d_blob, d_pr = up(var_window, np.int32), up(off, np.int64), up(blob, np.uint8), up(priors, np.float64)
        out = torch.empty(n, dtype=torch.float64, device=self.device)
        rc = self.lib.plat_variant_posterior_batch(self.ctx, n, hb.n_ind, db.max_haps, db.t["win_hap_begin"].data_ptr(),
                                                   db.t["gl_off"].data_ptr(), db.t["seg_n_good"].data_ptr(),
                                                   db.gl.data_ptr(), db.freq.data_ptr(), d_w.data_ptr(), d_off.data_ptr(),
                                                   d_blob.data_ptr(), d_pr.data_ptr(), out.data_ptr(), self._stream())
        _lib.check(rc, "plat_variant_posterior_batch")
        self._sync()
        return out.cpu().numpy()

    def genotype_calls(self, db, sites):
        """computeGenotypeCallAndLikelihoods for every (site, sample).  `sites`: list of dicts {window, var_in_hap
        [H][nVar], is_ref [H]}.  Needs em().  Returns per site (phased [n_ind][2], likelihoods [n_ind][NL], out4 [n_ind][4])."""
        torch = _torch()
        nS = len(sites)
        if nS == 0:
            return []
        hb = db.host
        nvar = np.array([np.asarray(s["var_in_hap"]).shape[1] for s in sites], dtype=np.int32)
        vih = [np.asarray(s["var_in_hap"], dtype=np.int32).reshape(-1) for s in sites]
        ref = [np.asarray(s["is_ref"], dtype=np.int32) for s in sites]
        vih_off = np.concatenate([[0], np.cumsum([len(v) for v in vih])]).astype(np.int64)
        ref_off = np.concatenate([[0], np.cumsum([len(r) for r in ref])]).astype(np.int64)
        NL = (nvar.astype(np.int64) + 1) * (nvar + 2) // 2
        lik_off = np.concatenate([[0], np.cumsum(NL * hb.n_ind)]).astype(np.int64)

        def up(a, dt):
            return torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(self.device)
        d_win = up([s["window"] for s in sites], np.int32)
        d_nvar, d_vo, d_ro, d_lo = up(nvar, np.int32), up(vih_off, np.int64), up(ref_off, np.int64), up(lik_off, np.int64)
        d_vih = up(np.concatenate(vih + [np.zeros(1, dtype=np.int32)]), np.int32)
        d_ref = up(np.concatenate(ref), np.int32)
        ph = torch.empty(nS * hb.n_ind * 2, dtype=torch.int32, device=self.device)
        lik = torch.empty(int(lik_off[-1]), dtype=torch.float64, device=self.device)
        out4 = torch.empty(nS * hb.n_ind * 4, dtype=torch.float64, device=self.device)
        rc = self.lib.plat_genotype_call_batch(self.ctx, nS, hb.n_ind, db.t["win_hap_begin"].data_ptr(),
                                               db.t["gl_off"].data_ptr(), db.gl.data_ptr(), db.gof.data_ptr(),
                                               db.freq.data_ptr(), d_win.data_ptr(), d_nvar.data_ptr(), d_vo.data_ptr(),
                                               d_ro.data_ptr(), d_vih.data_ptr(), d_ref.data_ptr(), d_lo.data_ptr(),
                                               ph.data_ptr(), lik.data_ptr(), out4.data_ptr(), self._stream())
        _lib.check(rc, "plat_genotype_call_batch")
        self._sync()
        ph, lik, out4 = ph.cpu().numpy().reshape(nS, hb.n_ind, 2), lik.cpu().numpy(), out4.cpu().numpy().reshape(nS, hb.n_ind, 4)
        return [(ph[s], lik[lik_off[s]:lik_off[s + 1]].reshape(hb.n_ind, int(NL[s])), out4[s]) for s in range(nS)]

    # ---- VariantCandidateGenerator ------------------------------------------------------
    @staticmethod
    def base_codes(blob):
        """The 2-bit codes of a byte blob as the device lays them out ((ASCII >> 1) & 3, base i at bits 2 (i & 15) of dword i >> 4), + 8 zero words."""
        a = np.frombuffer(bytes(blob), dtype=np.uint8)
        n = (len(a) + 15) // 16
        c = np.zeros(n * 16, dtype=np.uint32)
        c[:len(a)] = (a >> 1) & 3
        words = (c.reshape(n, 16) << (2 * np.arange(16, dtype=np.uint32))).sum(axis=1, dtype=np.uint64).astype(np.uint32)
        return np.concatenate([words, np.zeros(8, dtype=np.uint32)])

    def candidates(self, regions, min_flank=10, min_base_qual=20, gen_snps=1, gen_indels=1, max_per_read=64, codes=False):
        """VariantCandidateGenerator.addCandidatesFromReads for a list of regions.

        `regions`: list of dicts {ref: bytes (contig[ref_seq_start:...]), ref_seq_start, contig_len, reads: [dict(seq, qual,
        pos, flag, cigar [(op, len), ...])]}.  Returns per region the per-occurrence records [(refPos, removed, added,
        read index)] in the reference's emission order (merging equal variants is the caller's dictionary step)."""
        torch = _torch()
        nG = len(regions)
        reads = [r for g in regions for r in g["reads"]]
        nR = len(reads)
        if nR == 0:
            return [[] for _ in regions]

        def dev(a, dt):
            return torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(self.device)
        ref_blob = b"".join(bytes(g["ref"]) for g in regions)
        ref_off = np.concatenate([[0], np.cumsum([len(g["ref"]) for g in regions])]).astype(np.int64)
        seq_blob = b"".join(r["seq"] for r in reads)
        qual_blob = b"".join(r["qual"] for r in reads)
        read_off = np.concatenate([[0], np.cumsum([len(r["seq"]) for r in reads])]).astype(np.int64)
        cig = np.array([x for r in reads for c in r["cigar"] for x in c] + [0, 0], dtype=np.int16)
        cig_off = np.concatenate([[0], np.cumsum([len(r["cigar"]) for r in reads])]).astype(np.int32)
        region_of = np.repeat(np.arange(nG, dtype=np.int32), [len(g["reads"]) for g in regions])
        t = dict(ref=dev(pad_blob(np.frombuffer(ref_blob, dtype=np.uint8)), np.uint8), ref_off=dev(ref_off, np.int64),
                 rss=dev([g["ref_seq_start"] for g in regions], np.int32), clen=dev([g["contig_len"] for g in regions], np.int32),
                 seq=dev(pad_blob(np.frombuffer(seq_blob, dtype=np.uint8)), np.uint8),
                 qual=dev(pad_blob(np.frombuffer(qual_blob, dtype=np.uint8)), np.uint8), read_off=dev(read_off, np.int64),
                 pos=dev([r["pos"] for r in reads], np.int32), flags=dev([r["flag"] for r in reads], np.int32),
                 cig=dev(cig, np.int16), cig_off=dev(cig_off, np.int32), region_of=dev(region_of, np.int32))
        b = _lib.CandidateBatch()
        b.n_regions, b.n_reads = nG, nR
        b.ref_seq, b.ref_off, b.ref_seq_start, b.contig_len = t["ref"].data_ptr(), t["ref_off"].data_ptr(), t["rss"].data_ptr(), t["clen"].data_ptr()
        b.read_seq, b.read_qual, b.read_off = t["seq"].data_ptr(), t["qual"].data_ptr(), t["read_off"].data_ptr()
        b.read_pos, b.read_flags, b.cigar, b.cig_off = t["pos"].data_ptr(), t["flags"].data_ptr(), t["cig"].data_ptr(), t["cig_off"].data_ptr()
        while True:
            rec = torch.empty(nR * max_per_read * 5, dtype=torch.int32, device=self.device)
            cnt = torch.empty(nR, dtype=torch.int32, device=self.device)
            stt = torch.empty(nR, dtype=torch.int32, device=self.device)
            if codes:
                # the scan on 2-bit codes (plat_candidates_batch_codes): the reads' codes from the host here (the region loop gets them from the unpack
                # kernel), the reference's from plat_ref_codes; the caller promises reads of A, C, G, T, N only
                rc_t = dev(self.base_codes(seq_blob), np.uint32)
                fc_t = torch.zeros((len(ref_blob) + 15) // 16 + 16, dtype=torch.int32, device=self.device)
                irr = torch.zeros(nG + 1, dtype=torch.int32, device=self.device)
                _lib.check(self.lib.plat_ref_codes(self.ctx, nG, t["ref"].data_ptr(), t["ref_off"].data_ptr(), len(ref_blob), fc_t.data_ptr(), irr.data_ptr(),
                                                   self._stream()), "plat_ref_codes")
                rc = self.lib.plat_candidates_batch_codes(self.ctx, C.byref(b), rc_t.data_ptr(), fc_t.data_ptr(), irr.data_ptr(), min_flank, min_base_qual, gen_snps,
                                                          gen_indels, max_per_read, t["region_of"].data_ptr(), rec.data_ptr(), cnt.data_ptr(), stt.data_ptr(),
                                                          self._stream())
                self.last_ref_irregular = irr
            else:
                rc = self.lib.plat_candidates_batch(self.ctx, C.byref(b), min_flank, min_base_qual, gen_snps, gen_indels, max_per_read,
                                                    t["region_of"].data_ptr(), rec.data_ptr(), cnt.data_ptr(), stt.data_ptr(), self._stream())
            _lib.check(rc, "plat_candidates_batch")
            self._sync()
            cnt_h, st_h = cnt.cpu().numpy(), stt.cpu().numpy()
            if (st_h == -9).any():
                raise _lib.PlatypusDeviceError(-9, "a read reaches outside the reference window handed over", "plat_candidates_batch")
            if (st_h == -8).any():
                max_per_read = int(cnt_h.max())            # a read with more candidates than the slice: rerun with room for it
                continue
            break
        rec_h = rec.cpu().numpy().reshape(nR, max_per_read, 5)
        out = [[] for _ in regions]
        first = np.concatenate([[0], np.cumsum([len(g["reads"]) for g in regions])])
        for r in np.nonzero(cnt_h)[0].tolist():
            g = int(region_of[r])
            for p_, nrem, nadd, ro, ao in rec_h[r, :cnt_h[r]].tolist():
                out[g].append((p_, ref_blob[ro:ro + nrem] if nrem else b"", seq_blob[ao:ao + nadd] if nadd else b"", r - int(first[g])))
        return out

    # ---- read QC / trimming (checkAndTrimRead) ---------------------------------------------------------------
    def read_qc(self, streams, min_good_qual_bases=20, min_map_qual=20, min_base_qual=20, trim_overlapping=1, trim_adapter=1,
                trim_read_flank=0, trim_soft_clipped=1, enabled=(1, 1, 1, 1)):
        """checkAndTrimRead over whole streams of reads (one stream = the reads one bamReadBuffer sees, in order).
        `streams`: list of lists of dicts {qual, pos, mapq, flag, chromID, mateChromID, insertSize, matePos, cigar}.
        Returns per stream (ok [n], flags_out [n], quals_out [list of uint8 arrays], reason [n])."""
        torch = _torch()
        reads = [r for st in streams for r in st]
        n = len(reads)
        if n == 0:
            return [(np.zeros(0, np.int32), np.zeros(0, np.int32), [], np.zeros(0, np.int32)) for _ in streams]

        def dev(a, dt):
            return torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(self.device)
        off = np.concatenate([[0], np.cumsum([len(r["qual"]) for r in reads])]).astype(np.int64)
        qual = dev(pad_blob(np.concatenate([np.asarray(r["qual"], dtype=np.uint8) for r in reads])), np.uint8)
        t = dict(off=dev(off, np.int64), pos=dev([r["pos"] for r in reads], np.int32), mapq=dev([r["mapq"] for r in reads], np.uint8),
                 flags=dev([r["flag"] for r in reads], np.int32), cid=dev([r["chromID"] for r in reads], np.int16),
                 mcid=dev([r["mateChromID"] for r in reads], np.int16), ins=dev([r["insertSize"] for r in reads], np.int32),
                 mpos=dev([r["matePos"] for r in reads], np.int32),
                 cig=dev([x for r in reads for c in r["cigar"] for x in c] + [0, 0], np.int16),
                 coff=dev(np.concatenate([[0], np.cumsum([len(r["cigar"]) for r in reads])]), np.int32),
                 sof=dev(np.repeat(np.arange(len(streams)), [len(st) for st in streams]), np.int32))
        b = _lib.ReadQCBatch()
        b.n_reads = n
        b.read_qual, b.read_off, b.read_pos, b.read_mapq, b.read_flags = qual.data_ptr(), t["off"].data_ptr(), t["pos"].data_ptr(), t["mapq"].data_ptr(), t["flags"].data_ptr()
        b.chrom_id, b.mate_chrom_id, b.insert_size, b.mate_pos = t["cid"].data_ptr(), t["mcid"].data_ptr(), t["ins"].data_ptr(), t["mpos"].data_ptr()
        b.cigar, b.cig_off, b.stream_of = t["cig"].data_ptr(), t["coff"].data_ptr(), t["sof"].data_ptr()
        o = _lib.ReadQCOptions(min_good_qual_bases, min_map_qual, min_base_qual, trim_overlapping, trim_adapter, trim_read_flank,
                               trim_soft_clipped, *[int(x) for x in enabled])
        ok = torch.empty(n, dtype=torch.int32, device=self.device)
        why = torch.empty(n, dtype=torch.int32, device=self.device)
        _lib.check(self.lib.plat_read_qc_batch(self.ctx, C.byref(b), C.byref(o), ok.data_ptr(), why.data_ptr(), self._stream()),
                   "plat_read_qc_batch")
        self._sync()
        ok_h, why_h, fl_h, q_h = ok.cpu().numpy(), why.cpu().numpy(), t["flags"].cpu().numpy(), qual.cpu().numpy()
        out, a = [], 0
        for st in streams:
            e = a + len(st)
            out.append((ok_h[a:e], fl_h[a:e], [q_h[off[i]:off[i + 1]] for i in range(a, e)], why_h[a:e]))
            a = e
        return out

    # ---- read statistics of the VCF INFO field --------------------------------------------
    def variant_read_stats(self, windows, bad_reads_window=11, exact=0):
        """vcfINFO's per-read loop for a list of windows.  A window: dict {variants: [dict(pos, removed, added, bam_min,
        bam_max)], samples: [dict(good=[reads], bad=[reads])], var_in_genotype: [nVars][nInd]}; a read = dict(seq, qual, pos, end,
        mapq, flag, cigar).  All windows must have the same number of samples.  Returns per window a list over its variants of
        (counts[16], n_reads[nInd], n_var_reads[nInd], min_quals)."""
        torch = _torch()
        nI = len(windows[0]["samples"]) if windows else 0
        reads, gb, ge, bb, be, vw, vars_, vig, moff = [], [], [], [], [], [], [], [], []
        mtot = 0
        for w, win in enumerate(windows):
            assert len(win["samples"]) == nI
            ngood = 0
            for s_ in win["samples"]:
                gb.append(len(reads)); reads += s_["good"]; ge.append(len(reads)); ngood += len(s_["good"])
                bb.append(len(reads)); reads += s_["bad"]; be.append(len(reads))
            for k, v in enumerate(win["variants"]):
                vw.append(w); vars_.append(v); vig.append(win["var_in_genotype"][k]); moff.append(mtot); mtot += max(ngood, 1)
        nV = len(vars_)
        if nV == 0:
            return [[] for _ in windows]

        def dev(a, dt):
            return torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(self.device)
        blob = lambda parts: dev(pad_blob(np.frombuffer(b"".join(parts), dtype=np.uint8)), np.uint8)
        nadd = [len(v["added"]) for v in vars_]
        t = dict(vw=dev(vw, np.int32), vpos=dev([v["pos"] for v in vars_], np.int32), vmin=dev([v["bam_min"] for v in vars_], np.int32),
                 vmax=dev([v["bam_max"] for v in vars_], np.int32), nadd=dev(nadd, np.int32),
                 nrem=dev([len(v["removed"]) for v in vars_], np.int32), added=blob([v["added"] for v in vars_]),
                 aoff=dev(np.concatenate([[0], np.cumsum(nadd)[:-1]]), np.int64), vig=dev(np.asarray(vig, dtype=np.uint8).reshape(-1), np.uint8),
                 moff=dev(moff, np.int64), gb=dev(gb, np.int32), ge=dev(ge, np.int32), bb=dev(bb, np.int32), be=dev(be, np.int32),
                 seq=blob([r["seq"] for r in reads]), qual=blob([r["qual"] for r in reads]),
                 off=dev(np.concatenate([[0], np.cumsum([len(r["seq"]) for r in reads])]), np.int64),
                 pos=dev([r["pos"] for r in reads], np.int32), end=dev([r["end"] for r in reads], np.int32),
                 mapq=dev([r["mapq"] for r in reads], np.uint8), flags=dev([r["flag"] for r in reads], np.int32),
                 cig=dev([x for r in reads for c in r["cigar"] for x in c] + [0, 0], np.int16),
                 coff=dev(np.concatenate([[0], np.cumsum([len(r["cigar"]) for r in reads])]), np.int32))
        b = _lib.InfoStatsBatch()
        b.n_vars, b.n_ind = nV, nI
        for name, key in (("var_window", "vw"), ("var_pos", "vpos"), ("var_bam_min", "vmin"), ("var_bam_max", "vmax"), ("var_n_added", "nadd"),
                          ("var_n_removed", "nrem"), ("var_added", "added"), ("var_added_off", "aoff"), ("var_in_genotype", "vig"),
                          ("minq_off", "moff"), ("good_begin", "gb"), ("good_end", "ge"), ("bad_begin", "bb"), ("bad_end", "be"),
                          ("read_seq", "seq"), ("read_qual", "qual"), ("read_off", "off"), ("read_pos", "pos"), ("read_end", "end"),
                          ("read_mapq", "mapq"), ("read_flags", "flags"), ("cigar", "cig"), ("cig_off", "coff")):
            setattr(b, name, t[key].data_ptr())
        out = torch.empty(nV * 16, dtype=torch.int64, device=self.device)
        ps = torch.empty(nV * nI * 2, dtype=torch.int32, device=self.device)
        mq = torch.empty(max(mtot, 1), dtype=torch.int32, device=self.device)
        nmq = torch.empty(nV, dtype=torch.int32, device=self.device)
        _lib.check(self.lib.plat_variant_read_stats_batch(self.ctx, C.byref(b), bad_reads_window, exact, out.data_ptr(), ps.data_ptr(),
                                                          mq.data_ptr(), nmq.data_ptr(), self._stream()), "plat_variant_read_stats_batch")
        self._sync()
        out_h, ps_h, mq_h, nmq_h = out.cpu().numpy().reshape(nV, 16), ps.cpu().numpy().reshape(nV, nI, 2), mq.cpu().numpy(), nmq.cpu().numpy()
        res = [[] for _ in windows]
        for v in range(nV):
            res[vw[v]].append((out_h[v].tolist(), ps_h[v, :, 0].tolist(), ps_h[v, :, 1].tolist(), mq_h[moff[v]:moff[v] + nmq_h[v]].tolist()))
        return res

    # ---- a14..a18 ------------------------------------------------------------------------------------
    def upload_assembly(self, ab, max_vars=512, blob_per_region=1 << 16):
        """HBM image of the host arrays of plat_assembly_batch (dict as synth.config3 returns) + output buffers."""
        return AssemblyDeviceBatch(ab, self.device, max_vars, blob_per_region)

    def assemble_device(self, adb, kmer_size=15, min_qual=20, min_weight=40, no_cycles=0, hints=None):
        """plat_assemble_batch on a resident batch; enqueues only, results stay in HBM (adb.results() reads them).  hints = (longest
        reference window, most reads of a tile, most k-mer positions of a tile): plat_assemble_batch_async, which reads nothing back."""
        tail = (adb.max_vars, adb.blob_per_region, adb.cnt.data_ptr(), adb.pos.data_ptr(), adb.nrem.data_ptr(), adb.nadd.data_ptr(), adb.off.data_ptr(),
                adb.blob.data_ptr(), adb.status.data_ptr(), self._stream())
        if hints is None:
            rc = self.lib.plat_assemble_batch(self.ctx, C.byref(adb.struct), kmer_size, min_qual, min_weight, no_cycles, *tail)
        else:
            h = _lib.AssemblyHints(int(hints[0]), int(hints[1]), int(hints[2]))
            rc = self.lib.plat_assemble_batch_async(self.ctx, C.byref(adb.struct), C.byref(h), kmer_size, min_qual, min_weight, no_cycles, *tail)
        _lib.check(rc, "plat_assemble_batch")

    def assemble(self, regions, kmer_size=15, min_qual=20, min_weight=40, no_cycles=0, max_vars=512,
                 blob_per_region=1 << 16, hints=None):
        """assembleReadsAndDetectVariants for a list of regions.

        `regions`: list of dicts {ref: bytes, ref_start, assem_start, assem_end, seqs: [bytes], quals: [bytes]}
        (reads already in loadBAMDataIntoGraph order, QCFail reads removed).  Returns per region the list of
        (pos, removed, added) in the reference's sorted() order."""
        nG = len(regions)
        if nG == 0:
            return []
        ref_len = np.array([len(r["ref"]) for r in regions], dtype=np.int64)
        nreads = np.array([len(r["seqs"]) for r in regions], dtype=np.int64)
        rl = np.array([len(s) for r in regions for s in r["seqs"]], dtype=np.int64)
        ab = dict(n_regions=nG, n_reads=int(nreads.sum()),
                  ref_seq=np.frombuffer(b"".join(r["ref"] for r in regions), dtype=np.uint8),
                  ref_off=np.concatenate([[0], np.cumsum(ref_len)]),
                  ref_start=[r["ref_start"] for r in regions], assem_start=[r["assem_start"] for r in regions],
                  assem_end=[r["assem_end"] for r in regions], reg_read_begin=np.concatenate([[0], np.cumsum(nreads)]),
                  read_seq=np.frombuffer(b"".join(s for r in regions for s in r["seqs"]), dtype=np.uint8),
                  read_qual=np.frombuffer(b"".join(q for r in regions for q in r["quals"]), dtype=np.uint8),
                  read_off=np.concatenate([[0], np.cumsum(rl)]))
        adb = self.upload_assembly(ab, max_vars, blob_per_region)
        if hints == "exact":                                               # what a caller that built the batch knows
            per = [int(ref_len[g]) + 2 + sum(len(q) for q in regions[g]["seqs"]) + 2 * int(nreads[g]) for g in range(nG)]
            hints = (int(ref_len.max()), int(nreads.max()), max(per))
        self.assemble_device(adb, kmer_size, min_qual, min_weight, no_cycles, hints)
        return adb.results()

    def profile_enable(self, on=True):
        _lib.check(self.lib.plat_profile_enable(self.ctx, int(on)), "plat_profile_enable")

    def profile_last(self):
        p = _lib.Profile()
        _lib.check(self.lib.plat_profile_last(self.ctx, C.byref(p)), "plat_profile_last")
        return p

    def kernel_times(self):
        """{kernel name: (summed ms, launches)} of the launches bracketed since the profile was switched on / the last call (plat_kernel_times)."""
        ms = (C.c_double * 32)()
        n = (C.c_int64 * 32)()
        _lib.check(self.lib.plat_kernel_times(self.ctx, ms, n), "plat_kernel_times")
        return {(self.lib.plat_kernel_timer_name(i) or b"?").decode(): (float(ms[i]), int(n[i])) for i in range(32) if n[i]}

    def synchronize(self):
        """Waits for the stream and raises the first error an asynchronous call recorded since the last synchronize()."""
        _lib.check(self.lib.plat_stream_sync(self.ctx, self._stream()), "plat_stream_sync")
        self._sync()
