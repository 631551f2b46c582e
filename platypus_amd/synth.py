"""Synthetic workloads of BASELINE.json `configs`, generated in memory.

The generator reproduces what reaches the kernel in the reference: haplotype byte
strings `ref[start-buf:start) + mutated(start,end) + ref[end:end+buf)` with buf = min(2*rlen, 500)
(chaplotype.pyx:142,165-172), and per-window read slices in buffer order (reads sorted by position,
then badReads; cwindow.pyx:208-236,748-759).  All randomness is numpy PCG64 with fixed seeds.
"""
import numpy as np

from .batch import BAM_FQCFAIL, KIND_BAD, KIND_GOOD, HostBatch

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def _rand_bases(rng, shape):
    return ACGT[rng.integers(0, 4, size=shape)]


def _other_base(rng, base):
    """A uniformly random base different from `base` (uint8 arrays of ACGT)."""
    idx = np.searchsorted(ACGT, base)          # ACGT is sorted
    return ACGT[(idx + rng.integers(1, 4, size=base.shape)) % 4]


def make_snp_windows(n_windows, seed, read_len=150, depth=30, n_ind=1, ref_len=1200, max_snps=3,
                     cluster=40, min_var_dist=9, err=1e-3, frac_bad=0.02, hap_freq_beta=None, lowq_frac=0.0, q2_tail_reads=0.0,
                     read_indel_rate=0.0):
    """Config-2 style windows: 1..max_snps SNPs in a `cluster`-bp cluster, all 2^n haplotypes
    (the `nVars <= log2(maxHaplotypes-1)` branch, variantFilter.pyx:411-438), window = cluster +- 9
    (minVarDist), `read_len` reads at `depth`x per individual, diploid truth, 0.1% substitution errors,
    quals ~ clipped N(35,5) -> [2,41], `frac_bad` of reads with mapq < 20 (-> badReads, QCFail).
    Harder reads (config2_hard): `lowq_frac` of the bases get a quality uniform in [2, 19], `q2_tail_reads` of the reads end
    in a run of 5..40 bases of quality 2, and `read_indel_rate` per base of sequencing indels (one base inserted or skipped;
    the read keeps its length and its reported position)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    nW = n_windows
    buf = min(2 * read_len, 500)                                   # chaplotype.pyx:142
    ref = _rand_bases(rng, (nW, ref_len))
    nvar = rng.integers(1, max_snps + 1, size=nW)
    cs = rng.integers(ref_len // 2 - cluster, ref_len // 2, size=nW)
    # distinct sorted SNP offsets inside the cluster
    offs = np.sort(np.argsort(rng.random((nW, cluster)), axis=1)[:, :max_snps], axis=1)
    snp_pos = cs[:, None] + offs                                   # [nW, max_snps]; first nvar[w] are used
    # use the first nvar columns: re-sort so the used ones are ascending
    for w in np.nonzero(nvar < max_snps)[0]:
        snp_pos[w, :nvar[w]] = np.sort(snp_pos[w, :nvar[w]])
    snp_alt = _other_base(rng, np.take_along_axis(ref, snp_pos, axis=1))
    first = snp_pos[:, 0]
    last = snp_pos[np.arange(nW), nvar - 1]
    wstart = (first - min_var_dist).astype(np.int32)
    wend = (last + min_var_dist + 1).astype(np.int32)
    H = (1 << nvar).astype(np.int64)
    hap_len = (wend - wstart + 2 * buf).astype(np.int64)

    # ---- haplotypes: combination c (bit k set = SNP k alt), c = 0 is the reference haplotype
    nH = int(H.sum())
    win_hap_begin = np.concatenate([[0], np.cumsum(H)]).astype(np.int32)
    hap_w = np.repeat(np.arange(nW), H)
    hap_c = np.arange(nH) - win_hap_begin[hap_w]
    hlen = hap_len[hap_w]
    hap_off = np.concatenate([[0], np.cumsum(hlen)]).astype(np.int64)
    hap_seq = np.empty(int(hap_off[-1]), dtype=np.uint8)
    # gather ref[w, wstart-buf + j]
    pos_in = np.arange(int(hap_off[-1])) - np.repeat(hap_off[:-1], hlen)
    src = np.repeat((wstart[hap_w] - buf).astype(np.int64), hlen) + pos_in
    hap_seq[:] = ref[np.repeat(hap_w, hlen), src]
    for k in range(max_snps):
        sel = np.nonzero((k < nvar[hap_w]) & (((hap_c >> k) & 1) == 1))[0]
        p = hap_off[sel] + (snp_pos[hap_w[sel], k] - (wstart[hap_w[sel]] - buf))
        hap_seq[p] = snp_alt[hap_w[sel], k]

    # ---- truth genotypes per (window, individual): two haplotype combinations
    if hap_freq_beta is None:
        g1 = rng.integers(0, 1 << 30, size=(nW, n_ind)) % H[:, None]
        g2 = rng.integers(0, 1 << 30, size=(nW, n_ind)) % H[:, None]
    else:   # population mode: per-window haplotype frequencies ~ Beta(a, b) (config 5)
        f = rng.beta(hap_freq_beta[0], hap_freq_beta[1], size=(nW, 8)) + 1e-9
        f[np.arange(8)[None, :] >= H[:, None]] = 0
        cdf = np.cumsum(f / f.sum(axis=1, keepdims=True), axis=1)
        u1, u2 = rng.random((nW, n_ind)), rng.random((nW, n_ind))
        g1 = np.minimum((u1[:, :, None] > cdf[:, None, :]).sum(axis=2), H[:, None] - 1)
        g2 = np.minimum((u2[:, :, None] > cdf[:, None, :]).sum(axis=2), H[:, None] - 1)

    # ---- reads: starts uniform over every position with >= 1 bp overlap of the window
    span = (wend - wstart) + read_len - 1
    n_per = np.maximum(1, np.rint(depth * span / read_len)).astype(np.int64)       # per (window, individual)
    R_wi = np.repeat(n_per, n_ind).reshape(nW, n_ind)
    nR = int(R_wi.sum())
    seg_len = R_wi.reshape(-1)
    seg_read_begin = np.concatenate([[0], np.cumsum(seg_len)]).astype(np.int32)
    r_seg = np.repeat(np.arange(nW * n_ind), seg_len)
    r_w = r_seg // n_ind
    r_i = r_seg % n_ind
    pos = (wstart[r_w] - read_len + 1 + (rng.random(nR) * span[r_w]).astype(np.int64)).astype(np.int32)
    bad = rng.random(nR) < frac_bad
    mapq = np.where(bad, rng.integers(0, 20, size=nR), 60).astype(np.uint8)
    kind = np.where(bad, KIND_BAD, KIND_GOOD).astype(np.uint8)
    flags = np.where(bad, BAM_FQCFAIL | 3, 3).astype(np.int32)      # paired + proper pair (+ QCFail)
    # buffer order inside each (window, individual): good by pos, then bad by pos
    order = np.lexsort((pos, kind, r_seg))
    pos, mapq, kind, flags, r_w, r_i = pos[order], mapq[order], kind[order], flags[order], r_w[order], r_i[order]
    n_good = np.bincount(r_seg[order][kind == KIND_GOOD], minlength=nW * n_ind).astype(np.int32)
    allele = rng.integers(0, 2, size=nR)
    donor = np.where(allele == 0, g1[r_w, r_i], g2[r_w, r_i])
    cols = pos[:, None].astype(np.int64) + np.arange(read_len)[None, :]
    seq = ref[r_w[:, None], cols]
    for k in range(max_snps):
        has = (k < nvar[r_w]) & (((donor >> k) & 1) == 1)
        sp = snp_pos[r_w, k]
        inside = has & (sp >= pos) & (sp < pos + read_len)
        idx = np.nonzero(inside)[0]
        seq[idx, sp[idx] - pos[idx]] = snp_alt[r_w[idx], k]
    e = rng.random(seq.shape) < err
    seq[e] = _other_base(rng, seq[e])
    qual = np.clip(np.rint(rng.normal(35, 5, size=seq.shape)), 2, 41).astype(np.uint8)
    if read_indel_rate > 0:
        rows, cols_ = np.nonzero(rng.random(seq.shape) < read_indel_rate)
        ins = rng.random(len(rows)) < 0.5
        for r_, c_, i_ in zip(rows.tolist(), cols_.tolist(), ins.tolist()):
            if i_:                                                   # one base inserted at c_: the rest moves right
                seq[r_, c_ + 1:] = seq[r_, c_:-1].copy()
                seq[r_, c_] = ACGT[rng.integers(0, 4)]
            else:                                                    # the base at c_ is skipped: the rest moves left
                seq[r_, c_:-1] = seq[r_, c_ + 1:].copy()
    if lowq_frac > 0:
        low = rng.random(seq.shape) < lowq_frac
        qual[low] = rng.integers(2, 20, size=int(low.sum())).astype(np.uint8)
    if q2_tail_reads > 0:
        tails = np.nonzero(rng.random(nR) < q2_tail_reads)[0]
        tl = rng.integers(5, 41, size=len(tails))
        for r_, t_ in zip(tails.tolist(), tl.tolist()):
            qual[r_, read_len - t_:] = 2
    read_off = (np.arange(nR + 1, dtype=np.int64) * read_len)
    win_read_begin = np.concatenate([[0], np.cumsum(R_wi.sum(axis=1))]).astype(np.int32)
    return HostBatch(
        n_ind=n_ind, win_hap_begin=win_hap_begin, win_read_begin=win_read_begin, win_start=wstart, win_end=wend,
        win_flank=np.full(nW, buf, dtype=np.int32), hap_seq=hap_seq, hap_off=hap_off,
        read_seq=seq.reshape(-1), read_qual=qual.reshape(-1), read_off=read_off, read_pos=pos,
        read_end=(pos + read_len).astype(np.int32), read_mapq=mapq, read_flags=flags, read_kind=kind,
        seg_read_begin=seg_read_begin, seg_n_good=n_good,
        meta=dict(kind="snp_windows", seed=seed, read_len=read_len, depth=depth, n_windows=nW, n_ind=n_ind))


def config1(seed=1001):
    """BASELINE config 1: single 1-kb window, 64 synthetic 100 bp reads x 4 haplotypes (ref + 3 single-SNP
    haplotypes at 1250/1500/1750), reads from a het 0/1 genotype, mapq 60, buf = 200, hapLen = 1400."""
    rng = np.random.Generator(np.random.PCG64(seed))
    L, buf, ws, we = 100, 200, 1000, 2000
    ref = _rand_bases(rng, 3000)
    snps = [1250, 1500, 1750]
    haps = [ref[ws - buf:we + buf].copy()]
    for s in snps:
        h = ref[ws - buf:we + buf].copy()
        h[s - (ws - buf)] = _other_base(rng, ref[s:s + 1])[0]
        haps.append(h)
    donors = [ref.copy(), ref.copy()]
    donors[1][snps[0]] = haps[1][snps[0] - (ws - buf)]
    nR = 64
    pos = np.sort(rng.integers(ws - L + 7, we - 7, size=nR)).astype(np.int32)
    seq = np.stack([donors[int(rng.integers(0, 2))][p:p + L] for p in pos])
    e = rng.random(seq.shape) < 1e-3
    seq[e] = _other_base(rng, seq[e])
    qual = np.clip(np.rint(rng.normal(35, 5, size=seq.shape)), 2, 41).astype(np.uint8)
    hap_seq = np.concatenate(haps)
    return HostBatch(
        n_ind=1, win_hap_begin=np.array([0, 4], dtype=np.int32), win_read_begin=np.array([0, nR], dtype=np.int32),
        win_start=np.array([ws], dtype=np.int32), win_end=np.array([we], dtype=np.int32),
        win_flank=np.array([buf], dtype=np.int32), hap_seq=hap_seq,
        hap_off=(np.arange(5, dtype=np.int64) * (we - ws + 2 * buf)), read_seq=seq.reshape(-1),
        read_qual=qual.reshape(-1), read_off=np.arange(nR + 1, dtype=np.int64) * L, read_pos=pos,
        read_end=(pos + L).astype(np.int32), read_mapq=np.full(nR, 60, dtype=np.uint8),
        read_flags=np.full(nR, 3, dtype=np.int32), read_kind=np.zeros(nR, dtype=np.uint8),
        seg_read_begin=np.array([0, nR], dtype=np.int32), seg_n_good=np.array([nR], dtype=np.int32),
        meta=dict(kind="config1", seed=seed))


def config2(n_windows=10000, seed=2002):
    """BASELINE config 2: 10k windows, 150 bp reads, 30x, <= 8 haplotypes/window, SNP-only."""
    return make_snp_windows(n_windows, seed, read_len=150, depth=30)


def config2_hard(n_windows=10000, seed=2202):
    """Config-2 geometry with reads an ungapped-alignment proof likes less: 1 % substitution errors, 3.5 % of the bases below
    Q20 plus Q2 tails of 5..40 bases on a tenth of the reads (5 % of all bases below Q20), 1e-4 sequencing indels per base."""
    return make_snp_windows(n_windows, seed, read_len=150, depth=30, err=1e-2, lowq_frac=0.035, q2_tail_reads=0.1,
                            read_indel_rate=1e-4)


def config5(n_windows=2000, n_ind=100, seed=5005):
    """BASELINE config 5: population mode, 100 samples at 30x each, haplotype frequencies ~ Beta(0.5, 2)."""
    return make_snp_windows(n_windows, seed, read_len=150, depth=30, n_ind=n_ind, hap_freq_beta=(0.5, 2.0))


def config5_weak_evidence(n_windows=200, n_ind=100, seed=5105, depth=1, lowq_frac=0.8, err=2e-2):
    """Config 5's geometry with WEAK evidence: 100 samples at `depth`x (1x: a read or two per sample and window), four fifths of the
    bases below Q20, 2 % substitution errors.  At 30x the
    genotype likelihoods are so peaked that the EM of Population.call (cpopulation.pyx:384-457) stops after two iterations; here a
    sample's likelihoods barely separate its genotypes, the haplotype frequencies have to be learned from the whole population and
    the EM runs for tens of iterations: the workload that measures the EM kernel doing work."""
    return make_snp_windows(n_windows, seed, read_len=150, depth=depth, n_ind=n_ind, hap_freq_beta=(0.5, 2.0), lowq_frac=lowq_frac, err=err)


# ---- BASELINE config 4: regions with reads, for the region pipeline (candidates -> windows -> records) ---------------------

def _read_with_cigar(ref, p0, L, carried):
    """Sequence, CIGAR [(op, len)] and end position of a read starting at reference position p0 that carries the variants
    `carried` = sorted [(pos, removed, added)] (Platypus position convention: an indel sits after the base at pos)."""
    seq, cig, rp = bytearray(), [], p0

    def push(op, n):
        if n > 0:
            if cig and cig[-1][0] == op:
                cig[-1] = (op, cig[-1][1] + n)
            else:
                cig.append((op, n))
    todo = [v for v in carried if v[0] >= p0]
    while len(seq) < L:
        v = todo[0] if todo else None
        if v is None or v[0] - rp >= L - len(seq):
            k = L - len(seq); seq += ref[rp:rp + k]; push(0, k); rp += k
            break
        pos, rem, add = v
        if len(rem) == len(add):
            k = pos - rp; seq += ref[rp:rp + k] + add; push(0, k + len(add)); rp = pos + len(add)
        elif len(rem) == 0:
            k = pos - rp + 1; seq += ref[rp:rp + k]; push(0, k); seq += add; push(1, len(add)); rp = pos + 1
        else:
            k = pos - rp + 1; seq += ref[rp:rp + k]; push(0, k); push(2, len(rem)); rp = pos + 1 + len(rem)
        todo = [x for x in todo if x[0] >= rp]
    seq = seq[:L]
    out, used = [], 0
    for op, n in cig:
        if op in (0, 1):
            n = min(n, L - used); used += n
        if n > 0:
            out.append((op, n))
    while out and out[-1][0] == 2:
        out.pop()
    return bytes(seq), out, p0 + sum(n for op, n in out if op in (0, 2))


def config4_region(index, seed=4004, region_len=100000, n_samples=1, depth=30, read_len=150, snp_rate=1e-3, indel_rate=1e-4,
                   flank=1000, err=1e-3):
    """One region of BASELINE config 4, generated procedurally from (seed, index): its own contig `r<index>` (flank + region +
    flank), SNPs at snp_rate and 1..10 bp indels at indel_rate inside the region, a diploid donor per sample (each variant on
    either haplotype with probability 1/2, so hets and homs mix), `depth`x reads of read_len with the CIGAR an aligner
    would report, 0.1 % substitution errors, qualities ~ clipped N(35, 5), mapq 60.
    Returns dict(chrom, start, end, ref (bytes), variants [(pos, removed, added)], samples [list of read dicts], truth
    [sample][variant] = copies of the alternative allele in the donor)."""
    rng = np.random.Generator(np.random.PCG64([seed, index]))
    n = region_len + 2 * flank
    ref = bytes(_rand_bases(rng, n))
    start, end = flank, flank + region_len
    acgt = b"ACGT"
    variants, last = [], start + 20
    n_snp, n_indel = rng.poisson(region_len * snp_rate), rng.poisson(region_len * indel_rate)
    kinds = [0] * int(n_snp) + [1] * int(n_indel)
    spots = np.sort(rng.choice(np.arange(start + 20, end - 40), size=min(len(kinds), max(0, region_len - 60)), replace=False))
    rng.shuffle(kinds)
    for p, kind in zip(spots.tolist(), kinds):
        if p < last:
            continue
        if kind == 0:
            variants.append((p, ref[p:p + 1], bytes([acgt[(acgt.index(ref[p]) + 1 + int(rng.integers(0, 3))) % 4]])))
            last = p + 2
        else:
            k = 1 + int(min(9, rng.geometric(0.4) - 1))
            if rng.random() < 0.5:
                variants.append((p, b"", bytes(_rand_bases(rng, k))))
                last = p + 3
            else:
                variants.append((p, ref[p + 1:p + 1 + k], b""))
                last = p + k + 3
    samples, truth = [], []
    for _ in range(n_samples):
        carry = [[v for v in variants if rng.random() < 0.5] for _ in range(2)]
        truth.append([sum(v in c for c in carry) for v in variants])             # copies of the alternative allele: 0, 1 or 2
        reads = []
        for _ in range(int(depth * region_len / read_len)):
            h = carry[int(rng.integers(0, 2))]
            p0 = int(rng.integers(start - read_len + 10, end - 10))
            seq, cig, e = _read_with_cigar(ref, p0, read_len, [v for v in h if p0 <= v[0] < p0 + read_len + 12])
            s = np.frombuffer(seq, dtype=np.uint8).copy()
            bad = np.nonzero(rng.random(len(s)) < err)[0]
            if len(bad):
                s[bad] = _other_base(rng, s[bad])
            q = np.clip(np.rint(rng.normal(35, 5, size=len(s))), 2, 41).astype(np.uint8)
            reads.append(dict(seq=s.tobytes(), qual=q.tobytes(), pos=p0, end=e, mapq=60, flag=3 | (16 if rng.random() < 0.5 else 0), cigar=cig))
        reads.sort(key=lambda r: r["pos"])
        samples.append(reads)
    return dict(chrom="r%d" % index, start=start, end=end, ref=ref, variants=variants, samples=samples, truth=truth)


# ---- BASELINE config 3: assembly tiles ---------------------------------------------------------------------------------------

def config3(n_regions=2000, seed=3003, ref_len=4500, tile=1500, read_len=250, depth=30, lowq_frac=0.05, err=1e-3):
    """BASELINE config 3: `n_regions` assembly tiles.  Each: `ref_len` bp of random reference (the 1.5 kb tile +- 1.5 kb), a
    diploid donor carrying 1..3 indels (length geometric, 1..60 bp, half insertions) and 0..3 SNPs inside the tile, `read_len` bp
    reads at `depth`x from either donor haplotype, qualities ~ clipped N(35, 5) with `lowq_frac` of the bases below Q20,
    0.1 % substitution errors.  Returns the host arrays of plat_assembly_batch (dict: ref_seq, ref_off, ref_start, assem_start,
    assem_end, reg_read_begin, read_seq, read_qual, read_off, n_regions, n_reads) plus `truth` = planted variants per region."""
    rng = np.random.Generator(np.random.PCG64(seed))
    refs, seqs, ref_start, a0s, a1s, nreads, truth = [], [], [], [], [], [], []
    n_per = depth * ref_len // read_len
    for g in range(n_regions):
        ref = _rand_bases(rng, ref_len)
        rs = int(rng.integers(0, 1 << 20))
        lo = (ref_len - tile) // 2
        nind, nsnp = int(rng.integers(1, 4)), int(rng.integers(0, 4))
        spots = np.sort(rng.choice(np.arange(lo + 30, lo + tile - 100), size=nind + nsnp, replace=False)).tolist()
        kinds = [1] * nind + [0] * nsnp
        rng.shuffle(kinds)
        planted = []
        for p, kd in zip(spots, kinds):
            if kd == 0:
                planted.append((p, 0, _other_base(rng, ref[p:p + 1])))
            else:
                k = int(min(60, rng.geometric(0.15)))
                planted.append((p, -k, None) if rng.random() < 0.5 else (p, k, _rand_bases(rng, k)))
        donors = []
        for _ in range(2):
            parts, cur = [], 0
            for p, k, bases in planted:
                if p < cur or rng.random() < 0.5:
                    continue
                if k == 0:
                    parts += [ref[cur:p], bases]; cur = p + 1
                elif k > 0:
                    parts += [ref[cur:p + 1], bases]; cur = p + 1
                else:
                    parts += [ref[cur:p + 1]]; cur = p + 1 - k
            parts.append(ref[cur:])
            donors.append(np.concatenate(parts))
        which = rng.integers(0, 2, size=n_per)
        out = np.empty((n_per, read_len), dtype=np.uint8)
        for d in (0, 1):
            sel = np.nonzero(which == d)[0]
            st = rng.integers(0, len(donors[d]) - read_len, size=len(sel))
            out[sel] = donors[d][st[:, None] + np.arange(read_len)[None, :]]
        refs.append(ref); seqs.append(out); ref_start.append(rs); a0s.append(rs + lo); a1s.append(rs + lo + tile)
        nreads.append(n_per); truth.append([(rs + p, k) for p, k, _ in planted])
    seq = np.concatenate(seqs)
    e = rng.random(seq.shape, dtype=np.float32) < err
    seq[e] = _other_base(rng, seq[e])
    qual = rng.standard_normal(seq.shape, dtype=np.float32)
    qual *= 5.0
    qual += 35.5                                                         # floor(x + 0.5): round half up is as good as rint here
    np.clip(qual, 2, 41.5, out=qual)
    qual = qual.astype(np.uint8)
    low = rng.random(seq.shape, dtype=np.float32) < lowq_frac
    qual[low] = rng.integers(2, 20, size=int(low.sum()), dtype=np.uint8)
    nR = seq.shape[0]
    return dict(n_regions=n_regions, n_reads=nR, ref_seq=np.concatenate(refs), ref_off=np.arange(n_regions + 1, dtype=np.int64) * ref_len,
                ref_start=np.array(ref_start, dtype=np.int32), assem_start=np.array(a0s, dtype=np.int32),
                assem_end=np.array(a1s, dtype=np.int32), reg_read_begin=np.concatenate([[0], np.cumsum(nreads)]).astype(np.int32),
                read_seq=seq.reshape(-1), read_qual=qual.reshape(-1), read_off=np.arange(nR + 1, dtype=np.int64) * read_len, truth=truth)


def config4_region_arrays(index, seed=4004, region_len=100000, n_samples=1, depth=30, read_len=150, snp_rate=1e-3, indel_rate=1e-4,
                          flank=1000, err=1e-3):
    """One region of BASELINE config 4 as ARRAYS (the form a BAM loader leaves reads in, and what the native region loop takes):
    the same recipe as config4_region -- own contig `r<index>`, SNPs at snp_rate and 1..10 bp indels at indel_rate, a diploid donor
    per sample, `depth`x reads of read_len with the CIGAR an aligner would report, 0.1 % substitution errors, qualities ~ clipped
    N(35, 5), mapq 60 -- generated with array operations (a 100 kb region at 30x in a fraction of a second), so the draws differ
    from config4_region's.  Returns dict(chrom, start, end, ref (uint8 array), variants [(pos, removed, added)], samples = [dict(seq,
    qual, off, pos, end, mapq, flags, mate_pos, cigar, cig_off)] sorted by position, truth)."""
    rng = np.random.Generator(np.random.PCG64([seed, index, 7]))
    n = region_len + 2 * flank
    ref = _rand_bases(rng, n)
    start, end = flank, flank + region_len
    n_snp, n_indel = int(rng.poisson(region_len * snp_rate)), int(rng.poisson(region_len * indel_rate))
    kinds = np.array([0] * n_snp + [1] * n_indel)
    rng.shuffle(kinds)
    spots = np.sort(rng.choice(np.arange(start + 20, end - 40), size=min(len(kinds), max(0, region_len - 60)), replace=False))
    variants, last = [], start + 20                                              # (pos, kind, length, bases): kind 0 SNP, 1 insertion, 2 deletion
    for p, kd in zip(spots.tolist(), kinds.tolist()):
        if p < last:
            continue
        if kd == 0:
            variants.append((p, 0, 1, _other_base(rng, ref[p:p + 1])))
            last = p + 2
        else:
            k = 1 + int(min(9, rng.geometric(0.4) - 1))
            if rng.random() < 0.5:
                variants.append((p, 1, k, _rand_bases(rng, k)))
                last = p + 3
            else:
                variants.append((p, 2, k, None))
                last = p + k + 3
    out_vars = [(p, bytes(ref[p:p + 1]) if kd == 0 else (b"" if kd == 1 else bytes(ref[p + 1:p + 1 + k])),
                 bytes(b) if kd != 2 else b"") for p, kd, k, b in variants]
    samples, truth = [], []
    n_reads = int(depth * region_len / read_len)
    L = read_len
    for _ in range(n_samples):
        carry = rng.random((2, len(variants))) < 0.5
        truth.append(carry.sum(axis=0).astype(int).tolist())
        haps = []
        for h in range(2):
            parts, maps, cur = [], [], 0
            for (p, kd, k, b), c in zip(variants, carry[h].tolist()):
                if not c:
                    continue
                if kd == 0:
                    parts += [ref[cur:p], b]; maps += [np.arange(cur, p, dtype=np.int32), np.array([p], dtype=np.int32)]; cur = p + 1
                elif kd == 1:                                                    # insertion after the base at p
                    parts += [ref[cur:p + 1], b]; maps += [np.arange(cur, p + 1, dtype=np.int32), np.full(k, -1, dtype=np.int32)]; cur = p + 1
                else:                                                            # deletion of ref[p+1 : p+1+k]
                    parts += [ref[cur:p + 1]]; maps += [np.arange(cur, p + 1, dtype=np.int32)]; cur = p + 1 + k
            parts.append(ref[cur:]); maps.append(np.arange(cur, n, dtype=np.int32))
            hs, h2r = np.concatenate(parts), np.concatenate(maps)
            r2h = np.full(n + 1, len(hs), dtype=np.int64)
            m = np.nonzero(h2r >= 0)[0]
            r2h[h2r[m]] = m
            r2h = np.minimum.accumulate(r2h[::-1])[::-1]                         # a deleted reference base maps to the next base that exists
            haps.append((hs, h2r, r2h))
        which = rng.integers(0, 2, size=n_reads)
        p0 = rng.integers(start - L + 10, end - 10, size=n_reads)
        seq = np.empty((n_reads, L), dtype=np.uint8)
        refpos = np.empty((n_reads, L), dtype=np.int32)
        for h in range(2):
            hs, h2r, r2h = haps[h]
            sel = np.nonzero(which == h)[0]
            i0 = np.minimum(r2h[p0[sel]], len(hs) - L)
            idx = i0[:, None] + np.arange(L)[None, :]
            seq[sel] = hs[idx]
            refpos[sel] = h2r[idx]
        pos = refpos[:, 0].astype(np.int32)                                      # (the first base of a read is a reference base by construction)
        simple = (refpos[:, -1] - refpos[:, 0] == L - 1) & (refpos.min(axis=1) >= 0)
        endp = np.where(simple, pos + L, 0).astype(np.int32)
        cigs = [None] * n_reads
        for r_ in np.nonzero(~simple)[0].tolist():
            rp = refpos[r_]
            ops, lastref = [], None
            for x in rp.tolist():
                if x < 0:
                    op, ln = 1, 1
                else:
                    if lastref is not None and x - lastref > 1:
                        ops.append((2, x - lastref - 1))
                    op, ln, lastref = 0, 1, x
                if ops and ops[-1][0] == op:
                    ops[-1] = (op, ops[-1][1] + ln)
                else:
                    ops.append((op, ln))
            cigs[r_] = ops
            endp[r_] = lastref + 1
        e = rng.random(seq.shape, dtype=np.float32) < err
        seq[e] = _other_base(rng, seq[e])
        qual = rng.standard_normal(seq.shape, dtype=np.float32)
        qual *= 5.0
        qual += 35.5
        np.clip(qual, 2, 41.5, out=qual)
        qual = qual.astype(np.uint8)
        flags = (3 | (16 * (rng.random(n_reads) < 0.5))).astype(np.int32)
        order = np.argsort(pos, kind="stable")
        ncig = np.array([1 if c is None else len(c) for c in cigs], dtype=np.int64)[order]
        cig_off = np.concatenate([[0], np.cumsum(ncig)]).astype(np.int32)
        cigar = np.zeros((int(cig_off[-1]), 2), dtype=np.int16)
        cigar[cig_off[:-1], 1] = L                                               # default: one match of the whole read
        for k_, r_ in enumerate(order.tolist()):
            if cigs[r_] is not None:
                cigar[cig_off[k_]:cig_off[k_ + 1]] = np.array(cigs[r_], dtype=np.int16)
        samples.append(dict(seq=seq[order].reshape(-1), qual=qual[order].reshape(-1), off=np.arange(n_reads + 1, dtype=np.int64) * L,
                            pos=pos[order], end=endp[order], mapq=np.full(n_reads, 60, dtype=np.uint8), flags=flags[order],
                            mate_pos=np.full(n_reads, -1, dtype=np.int32), cigar=cigar.reshape(-1), cig_off=cig_off))
    return dict(chrom="r%d" % index, start=start, end=end, ref=ref, variants=out_vars, samples=samples, truth=truth)
