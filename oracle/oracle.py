"""ctypes bindings for the parity oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  The product package `platypus_amd` never does.

  * `Oracle`     -> oracle/liborc.so, our C restatement (oracle/plat_oracle.c)
  * `RefAlign`   -> oracle/_ref/libalign_ref.so, the UNMODIFIED reference src/c/align.c
                    (C ABI from the reference's src/c/align.h:8-12)
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIBORC = os.path.join(HERE, "liborc.so")
LIBREF = os.path.join(HERE, "_ref", "libalign_ref.so")


def build(quiet=True):
    """Compile liborc.so (and _ref/libalign_ref.so when /root/reference is present)."""
    out = subprocess.run(["make", "-C", HERE], capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + out.stdout + out.stderr)
    if not quiet:
        print(out.stdout)


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _bytes_arr(b):
    return np.frombuffer(bytes(b), dtype=np.int8).copy() if not isinstance(b, np.ndarray) else b


class Oracle:
    def __init__(self):
        if not os.path.exists(LIBORC):
            build()
        self.lib = L = C.CDLL(LIBORC)
        L.orc_dp_align.restype = C.c_int
        L.orc_dp_align.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int,
                                   C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_int)]
        L.orc_dp_score.restype = C.c_int
        L.orc_dp_score.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_char_p]
        L.orc_dp_batch.restype = None
        L.orc_dp_batch.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_flank_score.restype = C.c_int
        L.orc_flank_score.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int,
                                      C.c_char_p, C.c_char_p]
        L.orc_kmer_code.restype = C.c_uint
        L.orc_kmer_code.argtypes = [C.c_char_p]
        L.orc_gap_open.restype = None
        L.orc_gap_open.argtypes = [C.c_char_p, C.c_int, C.c_char_p]
        L.orc_loglik.restype = C.c_double
        L.orc_loglik.argtypes = [C.c_int, C.c_int]
        L.orc_align_read_to_hap.restype = C.c_int
        L.orc_align_read_to_hap.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_int,
                                            C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
        L.orc_align_window.restype = None
        L.orc_align_window.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                       C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                       C.c_void_p, C.POINTER(C.c_longlong)]
        L.orc_em_call.restype = C.c_int
        L.orc_em_call.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.POINTER(C.c_double)]
        L.orc_variant_posterior.restype = C.c_double
        L.orc_variant_posterior.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double]
        L.orc_genotype_call.restype = None
        L.orc_genotype_call.argtypes = [C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 8
        L.orc_variant_candidates.restype = C.c_int
        L.orc_variant_candidates.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 7 + [C.c_int] * 4 + [C.c_void_p, C.c_int]
        L.orc_check_and_trim.restype = None
        L.orc_check_and_trim.argtypes = [C.c_int] + [C.c_void_p] * 11 + [C.c_int] * 7 + [C.c_void_p] * 3
        L.orc_variant_read_stats.restype = None
        L.orc_variant_read_stats.argtypes = [C.c_int] + [C.c_void_p] * 7 + [C.c_int] + [C.c_void_p] * 14 + [C.c_int] * 3 + [C.c_void_p] * 3 + [C.c_int, C.c_void_p]
        L.orc_haplotype_score.restype = C.c_int
        L.orc_haplotype_score.argtypes = [C.c_int, C.c_void_p]
        L.orc_genotype_loglik.restype = C.c_double
        L.orc_genotype_loglik.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                          C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_population_setup_ind.restype = None
        L.orc_population_setup_ind.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                               C.c_void_p]
        L.orc_assemble.restype = C.c_int
        L.orc_assemble.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_int, C.POINTER(C.c_int)]

    # -- a1 ------------------------------------------------------------------------------------
    def dp_score(self, hap_slice, read, qual, go, gapextend=3, nucprior=2):
        assert len(hap_slice) >= len(read) + 15 and len(go) >= len(read) + 15
        return self.lib.orc_dp_score(bytes(hap_slice), bytes(read), bytes(qual), len(read), gapextend,
                                     nucprior, bytes(go))

    def dp_align(self, hap_slice, read, qual, go, gapextend=3, nucprior=2):
        n = len(read)
        a1 = C.create_string_buffer(2 * n + 16)
        a2 = C.create_string_buffer(2 * n + 16)
        fp = C.c_int(0)
        sc = self.lib.orc_dp_align(bytes(hap_slice), bytes(read), bytes(qual), n, gapextend, nucprior,
                                   bytes(go), a1, a2, C.byref(fp))
        return sc, a1.value, a2.value, fp.value

    def dp_batch(self, haps, reads, quals, gos, len2, gapextend=3, nucprior=2):
        n, lmax = reads.shape
        out = np.empty(n, dtype=np.int32)
        len2 = np.ascontiguousarray(len2, dtype=np.int32)
        self.lib.orc_dp_batch(n, lmax, haps.ctypes.data, reads.ctypes.data, quals.ctypes.data,
                              gos.ctypes.data, len2.ctypes.data, gapextend, nucprior, out.ctypes.data)
        return out

    def flank_score(self, hapLen, hapFlank, quals, go, firstpos, aln1, aln2, gapextend=3, nucprior=2):
        return self.lib.orc_flank_score(hapLen, hapFlank, bytes(quals), bytes(go), gapextend, nucprior,
                                        firstpos, bytes(aln1), bytes(aln2))

    # -- a3..a8 --------------------------------------------------------------------------------
    def kmer_code(self, seq7):
        return self.lib.orc_kmer_code(bytes(seq7))

    def gap_open(self, hap):
        out = C.create_string_buffer(len(hap) + 1)
        self.lib.orc_gap_open(bytes(hap), len(hap), out)
        return out.raw[:len(hap) + 1]

    def loglik(self, score, mapq):
        return self.lib.orc_loglik(int(score), int(mapq))

    def align_read_to_hap(self, read, qual, read_start, hap, hap_start, hap_flank, do_flank=0):
        nd = C.c_int(0)
        sc = self.lib.orc_align_read_to_hap(bytes(read), bytes(qual), len(read), read_start, bytes(hap),
                                            len(hap), hap_start, hap_flank, do_flank, C.byref(nd))
        return sc, nd.value

    # -- a9 ------------------------------------------------------------------------------------
    def align_window(self, haps, hap_start_pos, hap_end_pos, end_buffer, reads, do_flank=0):
        """haps: list[bytes]; reads: dict of arrays (seq list, qual list, pos, end, mapq, flags, kind)."""
        nH = len(haps)
        hap_blob = b"".join(haps)
        hap_len = np.array([len(h) for h in haps], dtype=np.int32)
        hap_off = np.concatenate([[0], np.cumsum(hap_len)[:-1]]).astype(np.int32)
        seqs, quals = reads["seq"], reads["qual"]
        nR = len(seqs)
        seq_blob = b"".join(seqs)
        qual_blob = b"".join(quals)
        rlen = np.array([len(s) for s in seqs], dtype=np.int32)
        roff = (np.concatenate([[0], np.cumsum(rlen)[:-1]]) if nR else np.zeros(0)).astype(np.int32)
        pos = np.ascontiguousarray(reads["pos"], dtype=np.int32)
        end = np.ascontiguousarray(reads["end"], dtype=np.int32)
        mapq = np.ascontiguousarray(reads["mapq"], dtype=np.uint8)
        flags = np.ascontiguousarray(reads["flags"], dtype=np.int32)
        kind = np.ascontiguousarray(reads["kind"], dtype=np.uint8)
        ll = np.zeros((nH, nR), dtype=np.float64)
        sc = np.zeros((nH, nR), dtype=np.int32)
        ndp = C.c_longlong(0)
        hb = np.frombuffer(hap_blob, dtype=np.uint8)
        sb = np.frombuffer(seq_blob + b"\0", dtype=np.uint8)
        qb = np.frombuffer(qual_blob + b"\0", dtype=np.uint8)
        self.lib.orc_align_window(nH, hb.ctypes.data, hap_off.ctypes.data, hap_len.ctypes.data,
                                  hap_start_pos, hap_end_pos, end_buffer, nR, sb.ctypes.data, qb.ctypes.data,
                                  roff.ctypes.data, rlen.ctypes.data, pos.ctypes.data, end.ctypes.data,
                                  mapq.ctypes.data, flags.ctypes.data, kind.ctypes.data, do_flank,
                                  ll.ctypes.data, sc.ctypes.data, C.byref(ndp))
        return ll, sc, ndp.value

    # -- a11 / a12 -----------------------------------------------------------------------------
    def genotype_loglik(self, arr1, arr2, same_hap, n_good):
        a1 = np.ascontiguousarray(arr1, dtype=np.float64)
        a2 = a1 if same_hap else np.ascontiguousarray(arr2, dtype=np.float64)
        gof = C.c_double(0)
        h1 = C.c_double(0)
        h2 = C.c_double(0)
        L = self.lib.orc_genotype_loglik(a1.ctypes.data, a2.ctypes.data, int(same_hap), len(a1) - 1,
                                         n_good, C.byref(gof), C.byref(h1), C.byref(h2))
        return L, gof.value, h1.value, h2.value

    def haplotype_score(self, hap_likes):
        a = np.ascontiguousarray(hap_likes, dtype=np.float64)
        return int(self.lib.orc_haplotype_score(len(a), a.ctypes.data))

    def population_setup_ind(self, ll_rows, n_good):
        """ll_rows: [nHaps][totalReads] (no sentinel).  Returns (logl, gl, gof) per genotype."""
        ll_rows = np.asarray(ll_rows, dtype=np.float64)
        nH, nR = ll_rows.shape
        withs = np.concatenate([ll_rows, np.full((nH, 1), 999.0)], axis=1).copy()
        nG = nH * (nH + 1) // 2
        logl = np.zeros(nG)
        gl = np.zeros(nG)
        gof = np.zeros(nG)
        self.lib.orc_population_setup_ind(nH, withs.ctypes.data, nR, n_good, logl.ctypes.data,
                                          gl.ctypes.data, gof.ctypes.data)
        return logl, gl, gof

    # -- EM, genotype calls, posteriors -----------------------------------------
    def em_call(self, n_reads, gl, max_iters=100, use_em=0):
        """gl: [nInd][nGen].  Returns (freqs[nHap], em[nInd][nGen], calls[nInd], iters, max_change)."""
        gl = np.ascontiguousarray(gl, dtype=np.float64)
        nInd, nGen = gl.shape
        nHap = int(round((np.sqrt(8 * nGen + 1) - 1) / 2))
        nr = np.ascontiguousarray(n_reads, dtype=np.int32)
        freq = np.zeros(nHap)
        em = np.zeros((nInd, nGen))
        calls = np.zeros(nInd, dtype=np.int32)
        mc = C.c_double(0)
        it = self.lib.orc_em_call(nInd, nHap, nr.ctypes.data, gl.ctypes.data, max_iters, use_em, freq.ctypes.data,
                                  em.ctypes.data, calls.ctypes.data, C.byref(mc))
        return freq, em, calls, it, mc.value

    def variant_posterior(self, n_reads, gl, freq, hap_has_var, prior):
        gl = np.ascontiguousarray(gl, dtype=np.float64)
        nInd = gl.shape[0]
        freq = np.ascontiguousarray(freq, dtype=np.float64)
        nr = np.ascontiguousarray(n_reads, dtype=np.int32)
        hv = np.ascontiguousarray(hap_has_var, dtype=np.uint8)
        return self.lib.orc_variant_posterior(nInd, len(freq), nr.ctypes.data, gl.ctypes.data, freq.ctypes.data,
                                              hv.ctypes.data, float(prior))

    def genotype_call(self, freq, gl_row, gof_row, var_in_hap, is_ref, n_individuals):
        """vcfutils.computeGenotypeCallAndLikelihoods for one sample.  Returns (phased, likelihoods, out4)."""
        freq = np.ascontiguousarray(freq, dtype=np.float64)
        gl_row = np.ascontiguousarray(gl_row, dtype=np.float64)
        gof_row = np.ascontiguousarray(gof_row, dtype=np.float64)
        vih = np.ascontiguousarray(var_in_hap, dtype=np.int32)
        nHap, nVar = vih.shape
        ir = np.ascontiguousarray(is_ref, dtype=np.int32)
        ph = np.zeros(2, dtype=np.int32)
        lik = np.zeros((nVar + 1) * (nVar + 2) // 2)
        out4 = np.zeros(4)
        self.lib.orc_genotype_call(nHap, nVar, n_individuals, freq.ctypes.data, gl_row.ctypes.data, gof_row.ctypes.data,
                                   vih.ctypes.data, ir.ctypes.data, ph.ctypes.data, lik.ctypes.data, out4.ctypes.data)
        return ph, lik, out4

    # -- VariantCandidateGenerator -----------------------------------------------
    def variant_candidates(self, ref, ref_seq_start, contig_len, reads, min_flank=10, min_base_qual=20, gen_snps=1, gen_indels=1):
        """reads: list of dicts {seq, qual (bytes), pos, flag, cigar [(op, len), ...]}.  Returns the per-occurrence records
        [(pos, removed, added, read index)] in the reference's emission order."""
        seqs = [r["seq"] for r in reads]
        off = np.concatenate([[0], np.cumsum([len(x) for x in seqs])]).astype(np.int64)
        sb = b"".join(seqs) + b"\0"
        qb = b"".join(r["qual"] for r in reads) + b"\0"
        pos = np.array([r["pos"] for r in reads], dtype=np.int32)
        flags = np.array([r["flag"] for r in reads], dtype=np.int32)
        cig = np.array([x for r in reads for c in r["cigar"] for x in c] + [0, 0], dtype=np.int16)
        coff = np.concatenate([[0], np.cumsum([len(r["cigar"]) for r in reads])]).astype(np.int32)
        cap = 64 * max(1, len(reads))
        while True:
            rec = np.zeros((cap, 6), dtype=np.int32)
            n = self.lib.orc_variant_candidates(bytes(ref), len(ref), ref_seq_start, contig_len, len(reads), sb, qb,
                                                off.ctypes.data, pos.ctypes.data, flags.ctypes.data, cig.ctypes.data,
                                                coff.ctypes.data, min_flank, min_base_qual, gen_snps, gen_indels,
                                                rec.ctypes.data, cap)
            if n != -1:
                break
            cap *= 4
        if n < 0:
            raise RuntimeError("orc_variant_candidates: %d" % n)
        out = []
        for p, nrem, nadd, ro, ao, ri in rec[:n].tolist():
            out.append((p, bytes(ref[ro:ro + nrem]) if nrem else b"", sb[ao:ao + nadd] if nadd else b"", ri))
        return out

    # -- read QC / trimming ---------------------------------------------------------------------------
    def check_and_trim(self, reads, opt):
        """reads: one stream, dicts {qual (list/bytes), pos, mapq, flag, chromID, mateChromID, insertSize, matePos, cigar}.
        Returns (ok, flags_out, quals_out(list of lists), reason)."""
        n = len(reads)
        off = np.concatenate([[0], np.cumsum([len(r["qual"]) for r in reads])]).astype(np.int64)
        qual = np.concatenate([np.asarray(r["qual"], dtype=np.int8) for r in reads] + [np.zeros(1, dtype=np.int8)])
        arr = lambda k, dt: np.array([r[k] for r in reads], dtype=dt)
        pos, mapq, flags = arr("pos", np.int32), arr("mapq", np.uint8), arr("flag", np.int32)
        cid, mcid, ins, mpos = arr("chromID", np.int16), arr("mateChromID", np.int16), arr("insertSize", np.int32), arr("matePos", np.int32)
        cig = np.array([x for r in reads for c in r["cigar"] for x in c] + [0, 0], dtype=np.int16)
        coff = np.concatenate([[0], np.cumsum([len(r["cigar"]) for r in reads])]).astype(np.int32)
        en = np.array(opt["enabled"], dtype=np.int32)
        ok = np.zeros(n, dtype=np.int32); reason = np.zeros(n, dtype=np.int32)
        self.lib.orc_check_and_trim(n, qual.ctypes.data, off.ctypes.data, pos.ctypes.data, mapq.ctypes.data, flags.ctypes.data,
                                    cid.ctypes.data, mcid.ctypes.data, ins.ctypes.data, mpos.ctypes.data, cig.ctypes.data,
                                    coff.ctypes.data, opt["minGoodQualBases"], opt["minMapQual"], opt["minBaseQual"],
                                    opt["trimOverlapping"], opt["trimAdapter"], opt["trimReadFlank"], opt["trimSoftClipped"],
                                    en.ctypes.data, ok.ctypes.data, reason.ctypes.data)
        return ok, flags, [qual[off[i]:off[i + 1]].tolist() for i in range(n)], reason

    # -- read statistics of the VCF INFO field --------------------------------------------------------
    def variant_read_stats(self, variants, samples, var_in_genotype, min_base_qual=20, bad_reads_window=11, exact=0):
        """variants: dicts {pos, removed, added (bytes), bam_min, bam_max}; samples: dicts {good: [reads], bad: [reads]}, a read =
        {seq, qual (bytes), pos, end, mapq, flag, cigar}.  Returns per variant (counts[16], n_reads[nInd], n_var_reads[nInd], min_quals)."""
        reads, gb, ge, bb, be = [], [], [], [], []
        for s_ in samples:
            gb.append(len(reads)); reads += s_["good"]; ge.append(len(reads))
            bb.append(len(reads)); reads += s_["bad"]; be.append(len(reads))
        nV, nI = len(variants), len(samples)
        off = np.concatenate([[0], np.cumsum([len(r["seq"]) for r in reads])]).astype(np.int64)
        sb = b"".join(r["seq"] for r in reads) + b"\0"
        qb = b"".join(r["qual"] for r in reads) + b"\0"
        arr = lambda k, dt: np.array([r[k] for r in reads], dtype=dt)
        pos, end, mapq, flags = arr("pos", np.int32), arr("end", np.int32), arr("mapq", np.uint8), arr("flag", np.int32)
        cig = np.array([x for r in reads for c in r["cigar"] for x in c] + [0, 0], dtype=np.int16)
        coff = np.concatenate([[0], np.cumsum([len(r["cigar"]) for r in reads])]).astype(np.int32)
        va = lambda k: np.array([v[k] for v in variants], dtype=np.int32)
        vpos, vmin, vmax = va("pos"), va("bam_min"), va("bam_max")
        nadd = np.array([len(v["added"]) for v in variants], dtype=np.int32)
        nrem = np.array([len(v["removed"]) for v in variants], dtype=np.int32)
        ablob = b"".join(v["added"] for v in variants) + b"\0"
        aoff = np.concatenate([[0], np.cumsum(nadd)[:-1]]).astype(np.int32) if nV else np.zeros(0, np.int32)
        vig = np.ascontiguousarray(var_in_genotype, dtype=np.uint8).reshape(nV, nI)
        maxq = max(1, len(reads))
        out = np.zeros((nV, 16), dtype=np.int64); ps = np.zeros((nV, nI, 2), dtype=np.int32)
        minq = np.zeros((nV, maxq), dtype=np.int32); nminq = np.zeros(nV, dtype=np.int32)
        i32 = lambda a: np.array(a, dtype=np.int32)
        gb, ge, bb, be = i32(gb), i32(ge), i32(bb), i32(be)
        self.lib.orc_variant_read_stats(nV, vpos.ctypes.data, vmin.ctypes.data, vmax.ctypes.data, nadd.ctypes.data, nrem.ctypes.data,
                                        ablob, aoff.ctypes.data, nI, gb.ctypes.data, ge.ctypes.data, bb.ctypes.data, be.ctypes.data,
                                        vig.ctypes.data, sb, qb, off.ctypes.data, pos.ctypes.data, end.ctypes.data, mapq.ctypes.data,
                                        flags.ctypes.data, cig.ctypes.data, coff.ctypes.data, min_base_qual, bad_reads_window, exact,
                                        out.ctypes.data, ps.ctypes.data, minq.ctypes.data, maxq, nminq.ctypes.data)
        return [(out[v].tolist(), ps[v, :, 0].tolist(), ps[v, :, 1].tolist(), minq[v, :nminq[v]].tolist()) for v in range(nV)]

    # -- a14..a18 ------------------------------------------------------------------------------
    def assemble(self, ref, ref_start, assem_start, assem_end, seqs, quals, k=15, min_qual=20,
                 min_weight=40, no_cycles=0):
        nR = len(seqs)
        rlen = np.array([len(s) for s in seqs], dtype=np.int32)
        roff = (np.concatenate([[0], np.cumsum(rlen)[:-1]]) if nR else np.zeros(0)).astype(np.int32)
        sb = np.frombuffer(b"".join(seqs) + b"\0", dtype=np.uint8)
        qb = np.frombuffer(b"".join(quals) + b"\0", dtype=np.uint8)
        cap, blobcap = 4096, 1 << 20
        pos = np.zeros(cap, dtype=np.int32)
        nrem = np.zeros(cap, dtype=np.int32)
        nadd = np.zeros(cap, dtype=np.int32)
        off = np.zeros(cap, dtype=np.int32)
        blob = np.zeros(blobcap, dtype=np.uint8)
        nn = C.c_int(0)
        n = self.lib.orc_assemble(bytes(ref), len(ref), ref_start, assem_start, assem_end, nR,
                                  sb.ctypes.data, qb.ctypes.data, roff.ctypes.data, rlen.ctypes.data, k,
                                  min_qual, min_weight, no_cycles, cap, pos.ctypes.data, nrem.ctypes.data,
                                  nadd.ctypes.data, off.ctypes.data, blob.ctypes.data, blobcap, C.byref(nn))
        if n < 0:
            raise RuntimeError("orc_assemble: output capacity too small")
        out = []
        raw = blob.tobytes()
        for i in range(n):
            o = off[i]
            out.append((int(pos[i]), raw[o:o + nrem[i]], raw[o + nrem[i]:o + nrem[i] + nadd[i]]))
        return out, nn.value


class RefAlign:
    """The unmodified reference align.c (fastAlignmentRoutine / calculateFlankScore)."""

    def __init__(self):
        if not os.path.exists(LIBREF):
            raise FileNotFoundError(LIBREF + " (build with `make -C oracle` where /root/reference exists)")
        self.lib = L = C.CDLL(LIBREF)
        L.fastAlignmentRoutine.restype = C.c_int
        L.fastAlignmentRoutine.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int,
                                           C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_int)]
        L.calculateFlankScore.restype = C.c_int
        L.calculateFlankScore.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int,
                                          C.c_char_p, C.c_char_p]

    @staticmethod
    def available():
        return os.path.exists(LIBREF)

    def dp_score(self, hap_slice, read, qual, go, gapextend=3, nucprior=2):
        n = len(read)
        return self.lib.fastAlignmentRoutine(bytes(hap_slice), bytes(read), bytes(qual), n + 15, n, gapextend,
                                             nucprior, bytes(go), None, None, None)

    def dp_align(self, hap_slice, read, qual, go, gapextend=3, nucprior=2):
        n = len(read)
        a1 = C.create_string_buffer(2 * n + 16)
        a2 = C.create_string_buffer(2 * n + 16)
        fp = C.c_int(0)
        sc = self.lib.fastAlignmentRoutine(bytes(hap_slice), bytes(read), bytes(qual), n + 15, n, gapextend,
                                           nucprior, bytes(go), a1, a2, C.byref(fp))
        return sc, a1.value, a2.value, fp.value

    def flank_score(self, hapLen, hapFlank, quals, go, firstpos, aln1, aln2, gapextend=3, nucprior=2):
        return self.lib.calculateFlankScore(hapLen, hapFlank, bytes(quals), bytes(go), gapextend, nucprior,
                                            firstpos, bytes(aln1), bytes(aln2))
