"""The kernels that put a chunk's read table and a window batch's read arrays together on the device, each against numpy through the C ABI:
plat_gather_reads (cwindow.pyx:208-264: the reads between a window's pointers), plat_unpack_reads_pieces (the packed tables of up to 64
regions in one launch), plat_copy_pieces, plat_concat_read_tables (the per-read columns of resident tables).  Integer / byte work: exact."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from platypus_amd.engine import Engine
    return Engine(0)


def _dev(eng, a, dt):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(eng.device)


PAD = 32


def test_gather_reads_every_length_and_alignment(eng):
    """Destination read d = source read src_index[d]: bases, qualities, pos, end, mapq, bitFlag; lengths 0..300 (the kernel moves 16-byte
    pieces and ends the last one AT the read's end: 0, 1, 15, 16, 17, 31, 32, 33 ... are the edges), sources and destinations at every
    byte alignment, reads taken several times and not at all."""
    import torch
    from platypus_amd import _lib
    rng = np.random.default_rng(11)
    lens = np.concatenate([np.arange(0, 70), [150, 151, 250, 299, 300, 16, 32, 48, 64], rng.integers(1, 301, 400)])
    rng.shuffle(lens)
    n = len(lens)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    seq = rng.integers(1, 255, off[-1] + PAD).astype(np.uint8)
    qual = rng.integers(0, 94, off[-1] + PAD).astype(np.uint8)
    pos, end = rng.integers(0, 1 << 30, n), rng.integers(0, 1 << 30, n)
    mapq, flags = rng.integers(0, 256, n), rng.integers(0, 1 << 12, n)
    src = np.concatenate([rng.integers(0, n, 3000), np.arange(n)[::-1], [7] * 40])
    doff = np.concatenate([[0], np.cumsum(lens[src])]).astype(np.int64)
    d = dict(src=_dev(eng, src, np.int32), doff=_dev(eng, doff, np.int64), seq=_dev(eng, seq, np.uint8), qual=_dev(eng, qual, np.uint8), off=_dev(eng, off, np.int64),
             pos=_dev(eng, pos, np.int32), end=_dev(eng, end, np.int32), mapq=_dev(eng, mapq, np.uint8), flags=_dev(eng, flags, np.int32))
    nd = len(src)
    o = dict(seq=torch.full((int(doff[-1]) + PAD,), 0xEE, dtype=torch.uint8, device=eng.device), qual=torch.full((int(doff[-1]) + PAD,), 0xEE, dtype=torch.uint8, device=eng.device),
             pos=torch.zeros(nd, dtype=torch.int32, device=eng.device), end=torch.zeros(nd, dtype=torch.int32, device=eng.device),
             mapq=torch.zeros(nd, dtype=torch.uint8, device=eng.device), flags=torch.zeros(nd, dtype=torch.int32, device=eng.device))
    _lib.check(eng.lib.plat_gather_reads(eng.ctx, nd, d["src"].data_ptr(), d["doff"].data_ptr(), d["seq"].data_ptr(), d["qual"].data_ptr(), d["off"].data_ptr(),
                                         d["pos"].data_ptr(), d["end"].data_ptr(), d["mapq"].data_ptr(), d["flags"].data_ptr(), o["seq"].data_ptr(), o["qual"].data_ptr(),
                                         o["pos"].data_ptr(), o["end"].data_ptr(), o["mapq"].data_ptr(), o["flags"].data_ptr(), eng._stream()), "plat_gather_reads")
    eng._sync()
    want_seq = np.concatenate([seq[off[s]:off[s + 1]] for s in src])
    want_qual = np.concatenate([qual[off[s]:off[s + 1]] for s in src])
    got_seq, got_qual = o["seq"].cpu().numpy(), o["qual"].cpu().numpy()
    assert np.array_equal(got_seq[:doff[-1]], want_seq) and np.array_equal(got_qual[:doff[-1]], want_qual)
    assert (got_seq[doff[-1]:] == 0xEE).all() and (got_qual[doff[-1]:] == 0xEE).all()          # nothing behind the last read is touched
    assert np.array_equal(o["pos"].cpu().numpy(), pos[src].astype(np.int32)) and np.array_equal(o["end"].cpu().numpy(), end[src].astype(np.int32))
    assert np.array_equal(o["mapq"].cpu().numpy(), mapq[src].astype(np.uint8)) and np.array_equal(o["flags"].cpu().numpy(), flags[src].astype(np.int32))


class Piece(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_int64), ("n", C.c_int64)]


def _pieces(eng, triples):
    arr = (Piece * len(triples))(*[Piece(s, d, n) for s, d, n in triples])
    return _dev(eng, np.frombuffer(bytes(arr), dtype=np.uint8).copy(), np.uint8)


def test_unpack_and_copy_pieces_at_any_alignment(eng):
    """plat_unpack_reads_pieces: packed byte = 2-bit base (A C T G) | quality << 2, pieces at every source and destination alignment,
    empty pieces, exceptions (N, qualities above 63) patched behind them; plat_copy_pieces: the same pieces moved as they are."""
    import torch
    from platypus_amd import _lib
    rng = np.random.default_rng(12)
    sizes = [0, 1, 7, 15, 16, 17, 33, 100, 1000, 4097, 0, 31, 250000, 5]
    shifts = [0, 1, 3, 8, 15, 5, 0, 9, 2, 13, 4, 6, 11, 7]
    blob = rng.integers(0, 256, sum(sizes) + 16 * len(sizes) + PAD).astype(np.uint8)
    dblob = _dev(eng, blob, np.uint8)
    triples, at, dst = [], 0, 3                                             # (the destination starts at 3: no alignment shared with the sources)
    want_seq, want_qual, want_raw = [], [], []
    for n, sh in zip(sizes, shifts):
        at += sh
        triples.append((dblob.data_ptr() + at, dst, n))
        p = blob[at:at + n]
        want_seq.append(np.frombuffer(b"ACTG", dtype=np.uint8)[p & 3]); want_qual.append(p >> 2); want_raw.append(p)
        at += n; dst += n + (n % 3)
    total = dst + PAD
    exc_i = np.array([triples[7][1] + 5, triples[12][1] + 77777, triples[8][1]], dtype=np.int64)
    exc_b, exc_q = np.frombuffer(b"NNa", dtype=np.uint8).copy(), np.array([0, 93, 70], dtype=np.uint8)
    pc = _pieces(eng, triples)
    oseq = torch.full((total,), 0xEE, dtype=torch.uint8, device=eng.device)
    oqual = torch.full((total,), 0xEE, dtype=torch.uint8, device=eng.device)
    di, db, dq = _dev(eng, exc_i, np.int64), _dev(eng, exc_b, np.uint8), _dev(eng, exc_q, np.uint8)
    _lib.check(eng.lib.plat_unpack_reads_pieces(eng.ctx, len(triples), max(sizes), pc.data_ptr(), oseq.data_ptr(), oqual.data_ptr(), total, len(exc_i), di.data_ptr(),
                                                db.data_ptr(), dq.data_ptr(), eng._stream()), "plat_unpack_reads_pieces")
    oraw = torch.full((total,), 0xEE, dtype=torch.uint8, device=eng.device)
    _lib.check(eng.lib.plat_copy_pieces(eng.ctx, len(triples), max(sizes), pc.data_ptr(), oraw.data_ptr(), eng._stream()), "plat_copy_pieces")
    eng._sync()
    ws, wq, wr = (np.full(total, 0xEE, dtype=np.uint8) for _ in range(3))
    for (s_, d_, n_), a, b_, c in zip(triples, want_seq, want_qual, want_raw):
        ws[d_:d_ + n_] = a; wq[d_:d_ + n_] = b_; wr[d_:d_ + n_] = c
    ws[exc_i] = exc_b; wq[exc_i] = exc_q
    assert np.array_equal(oseq.cpu().numpy(), ws) and np.array_equal(oqual.cpu().numpy(), wq)   # (incl. the bytes between and behind the pieces: untouched)
    assert np.array_equal(oraw.cpu().numpy(), wr)


class TableDesc(C.Structure):
    _fields_ = [("off", C.c_void_p), ("pos", C.c_void_p), ("end", C.c_void_p), ("mapq", C.c_void_p), ("flags", C.c_void_p), ("cigar", C.c_void_p), ("cig_off", C.c_void_p),
                ("n", C.c_int32), ("scan", C.c_int32), ("first_read", C.c_int64), ("first_byte", C.c_int64), ("first_pair", C.c_int64)]


def test_concat_read_tables(eng):
    """The per-read columns of resident tables (offsets from 0 each) as ONE chunk table: offsets and CIGAR offsets rebased, the scan id of
    every read, the closing entries; tables without reads, with one read, with 3 000."""
    import torch
    from platypus_amd import _lib
    rng = np.random.default_rng(13)
    ns = [5, 0, 1, 3000, 64, 0, 257]
    keep, descs = [], []
    first_read = first_byte = first_pair = 0
    want = dict(off=[], pos=[], end=[], mapq=[], flags=[], cig_off=[], cigar=[], region=[])
    for t, n in enumerate(ns):
        lens = rng.integers(1, 300, n)
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        nc = rng.integers(1, 5, n)
        coff = np.concatenate([[0], np.cumsum(nc)]).astype(np.int32)
        cig = rng.integers(0, 300, 2 * int(coff[-1])).astype(np.int16)
        cols = dict(off=off, pos=rng.integers(0, 1 << 30, n).astype(np.int32), end=rng.integers(0, 1 << 30, n).astype(np.int32),
                    mapq=rng.integers(0, 256, n).astype(np.uint8), flags=rng.integers(0, 4096, n).astype(np.int32), cigar=np.concatenate([cig, [0, 0]]).astype(np.int16), cig_off=coff)
        dev = {k: _dev(eng, v, v.dtype) for k, v in cols.items()}
        keep.append(dev)
        first_byte += int(rng.integers(0, 16))                              # (tables need not follow one another without a gap)
        scan = -1 if t == 4 else t
        descs.append(TableDesc(dev["off"].data_ptr(), dev["pos"].data_ptr(), dev["end"].data_ptr(), dev["mapq"].data_ptr(), dev["flags"].data_ptr(),
                               dev["cigar"].data_ptr(), dev["cig_off"].data_ptr(), n, scan, first_read, first_byte, first_pair))
        want["off"].append(off[:-1] + first_byte); want["cig_off"].append(coff[:-1] + first_pair); want["cigar"].append(cig)
        for k in ("pos", "end", "mapq", "flags"): want[k].append(cols[k])
        want["region"].append(np.full(n, scan if scan >= 0 else 77, dtype=np.int32))
        first_read += n; first_byte += int(off[-1]); first_pair += int(coff[-1])
    N, B, P = first_read, first_byte, first_pair
    dd = _dev(eng, np.frombuffer(bytes((TableDesc * len(descs))(*descs)), dtype=np.uint8).copy(), np.uint8)
    out = dict(off=torch.zeros(N + 1, dtype=torch.int64, device=eng.device), pos=torch.zeros(N + 1, dtype=torch.int32, device=eng.device),
               end=torch.zeros(N + 1, dtype=torch.int32, device=eng.device), mapq=torch.zeros(N + 1, dtype=torch.uint8, device=eng.device),
               flags=torch.zeros(N + 1, dtype=torch.int32, device=eng.device), cig_off=torch.zeros(N + 1, dtype=torch.int32, device=eng.device),
               cigar=torch.zeros(2 * P + 2, dtype=torch.int16, device=eng.device), region=torch.full((N + 1,), 77, dtype=torch.int32, device=eng.device))
    _lib.check(eng.lib.plat_concat_read_tables(eng.ctx, len(descs), max(ns), dd.data_ptr(), out["off"].data_ptr(), out["pos"].data_ptr(), out["end"].data_ptr(),
                                               out["mapq"].data_ptr(), out["flags"].data_ptr(), out["cig_off"].data_ptr(), out["cigar"].data_ptr(), out["region"].data_ptr(),
                                               N, B, P, eng._stream()), "plat_concat_read_tables")
    eng._sync()
    g = {k: v.cpu().numpy() for k, v in out.items()}
    assert np.array_equal(g["off"], np.concatenate(want["off"] + [[B]])) and np.array_equal(g["cig_off"], np.concatenate(want["cig_off"] + [[P]]))
    for k in ("pos", "end", "mapq", "flags"):
        assert np.array_equal(g[k][:N], np.concatenate(want[k]))
    assert np.array_equal(g["cigar"], np.concatenate(want["cigar"] + [[0, 0]])) and np.array_equal(g["region"][:N], np.concatenate(want["region"]))


def test_more_pieces_than_a_grid_dimension_holds(eng):
    """Round 6: the pieces of a launch sit in gridDim.y (at most 65 535); a job-wide exchange hands plat_copy_pieces one piece per
    region.  70 000 pieces of 0..40 bytes: copied and unpacked exactly, the bytes between them untouched."""
    import torch
    from platypus_amd import _lib
    rng = np.random.default_rng(66)
    n = 70000
    sizes = rng.integers(0, 41, n).astype(np.int64)
    gaps = rng.integers(0, 3, n).astype(np.int64)
    src_at = np.cumsum(sizes + 1) - (sizes + 1)
    dst_at = np.cumsum(sizes + gaps) - (sizes + gaps) + 5
    blob = rng.integers(0, 256, int(src_at[-1] + sizes[-1] + 64)).astype(np.uint8)
    dblob = _dev(eng, blob, np.uint8)
    pieces = np.stack([dblob.data_ptr() + src_at, dst_at, sizes], axis=1).astype(np.int64)
    pc = _dev(eng, pieces.reshape(-1), np.int64)
    total = int(dst_at[-1] + sizes[-1] + 64)
    oraw, oseq, oqual = (torch.full((total,), 0xEE, dtype=torch.uint8, device=eng.device) for _ in range(3))
    _lib.check(eng.lib.plat_copy_pieces(eng.ctx, n, int(sizes.max()), pc.data_ptr(), oraw.data_ptr(), eng._stream()), "plat_copy_pieces")
    _lib.check(eng.lib.plat_unpack_reads_pieces(eng.ctx, n, int(sizes.max()), pc.data_ptr(), oseq.data_ptr(), oqual.data_ptr(), total, 0, 0, 0, 0, eng._stream()),
               "plat_unpack_reads_pieces")
    eng._sync()
    wr, ws, wq = (np.full(total, 0xEE, dtype=np.uint8) for _ in range(3))
    idx = np.concatenate([np.arange(d, d + k) for d, k in zip(dst_at, sizes)])
    val = np.concatenate([blob[a:a + k] for a, k in zip(src_at, sizes)])
    wr[idx] = val; ws[idx] = np.frombuffer(b"ACTG", dtype=np.uint8)[val & 3]; wq[idx] = val >> 2
    assert np.array_equal(oraw.cpu().numpy(), wr) and np.array_equal(oseq.cpu().numpy(), ws) and np.array_equal(oqual.cpu().numpy(), wq)


def test_unpack_pieces_also_writes_the_two_bit_codes(eng):
    """plat_unpack_reads_pieces_codes = plat_unpack_reads_pieces + the bases' 2-bit codes ((ASCII >> 1) & 3, base i at bits 2 (i & 15) of dword i >> 4):
    pieces that start and end anywhere inside a 16-base line (their partial lines are OR-ed in, neighbours share a dword), exceptions patched behind them."""
    import torch
    from platypus_amd import _lib
    from platypus_amd.engine import Engine
    rng = np.random.default_rng(14)
    sizes = [0, 1, 7, 15, 16, 17, 33, 100, 1000, 4097, 0, 31, 50000, 5, 16, 48]
    blob = rng.integers(0, 256, sum(sizes) + 64).astype(np.uint8)
    dblob = _dev(eng, blob, np.uint8)
    triples, at, dst = [], 0, 0
    for n in sizes:
        triples.append((dblob.data_ptr() + at, dst, n))
        at += n; dst += n                                                    # back to back in the output, as a chunk's tables are
    total = dst
    exc_i = np.array([triples[8][1] + 3, triples[12][1] + 777, triples[12][1] + 778], dtype=np.int64)
    exc_b, exc_q = np.frombuffer(b"NNT", dtype=np.uint8).copy(), np.array([0, 2, 70], dtype=np.uint8)
    pc = _pieces(eng, triples)
    oseq = torch.zeros(total + PAD, dtype=torch.uint8, device=eng.device)
    oqual = torch.zeros(total + PAD, dtype=torch.uint8, device=eng.device)
    codes = torch.full(((total + 15) // 16 + 8,), -1, dtype=torch.int32, device=eng.device)
    di, db, dq = _dev(eng, exc_i, np.int64), _dev(eng, exc_b, np.uint8), _dev(eng, exc_q, np.uint8)
    _lib.check(eng.lib.plat_unpack_reads_pieces_codes(eng.ctx, len(triples), max(sizes), pc.data_ptr(), oseq.data_ptr(), oqual.data_ptr(), codes.data_ptr(), total,
                                                      len(exc_i), di.data_ptr(), db.data_ptr(), dq.data_ptr(), eng._stream()), "plat_unpack_reads_pieces_codes")
    eng._sync()
    seq = oseq.cpu().numpy()[:total]
    want = np.frombuffer(b"ACTG", dtype=np.uint8)[blob[:total] & 3].copy()
    want[exc_i] = exc_b
    assert np.array_equal(seq, want)
    assert np.array_equal(codes.cpu().numpy().view(np.uint32), Engine.base_codes(bytes(want)))
